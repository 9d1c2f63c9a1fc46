"""Executable statement of rule R3 of DESIGN.md 3.9: the state-table slot protocol of the speculative kernel under a memory
system that lands posted stores LATE.

Not a test of the HIP code (that is tests/test_zz_jitter.py on a GPU) but of the protocol it implements, in a model small
enough to enumerate adversarial delays: one query, one table, batches of look-ups.  Memory holds, per slot, the value that
has LANDED; a compare-and-swap acts on the landed value at once; an agent-scope load returns the landed value; a plain probe
may return any value the slot has held (a stale L1 line); a posted store lands after a delay the adversary picks -- possibly
several batches later.  The round-3 protocol ("an own-tag claim that is still a claim after the re-read was abandoned by a cut
unit: pass it") creates a state twice under such delays -- what round 4 saw on the device under a neighbour's hipMemset; the
round-4 protocol (every claim is resolved by its batch -- entry or DEAD --, a claim of an earlier batch is waited for) never does.
"""
import random

EMPTY, DEAD = "empty", "dead"


class Table:
    def __init__(self, n, rng, max_delay, resolved_claims):
        self.n, self.rng, self.max_delay, self.resolved = n, rng, max_delay, resolved_claims
        self.landed = [EMPTY] * n                 # what memory holds
        self.history = [[EMPTY] for _ in range(n)]  # every value a slot has held (what a stale plain probe may show)
        self.pending = []                          # (lands_at, slot, value): posted stores on their way
        self.now = 0
        self.waits = 0

    def _set(self, slot, value):
        self.landed[slot] = value
        self.history[slot].append(value)

    def tick(self):
        self.now += 1
        due = [p for p in self.pending if p[0] <= self.now]
        self.pending = [p for p in self.pending if p[0] > self.now]
        for _, slot, value in sorted(due):  # (stores to one slot land in program order)
            self._set(slot, value)

    def post(self, slot, value):
        self.pending.append((self.now + self.rng.randint(0, self.max_delay), slot, value))

    def cas_empty(self, slot, value):
        old = self.landed[slot]
        if old == EMPTY:
            self._set(slot, value)
        return old

    def probe_plain(self, slot):
        return self.rng.choice(self.history[slot])

    def load_agent(self, slot):
        return self.landed[slot]

    def lookup(self, key, batch):
        """-> ("found", id) | ("claimed", slot).  Entries are ("entry", key, id); claims ("claim", key, batch)."""
        pos = (key * 7) % 64  # (start positions crowd into 64 slots: long shared chains; the table itself never fills -- at most 40 entries + 8 dead slots per batch)
        while True:
            v = self.probe_plain(pos)
            if v == EMPTY:
                old = self.cas_empty(pos, ("claim", key, batch))
                if old == EMPTY:
                    return ("claimed", pos)
                v = old
            if v != DEAD and v[0] == "claim":        # own tag (one query in the model): look past the L1
                v = self.load_agent(pos)
                if self.resolved:
                    while v != DEAD and v[0] == "claim" and v[2] != batch:  # a store on its way: wait for it
                        self.tick()
                        self.waits += 1
                        v = self.load_agent(pos)
            if v != DEAD and v != EMPTY and v[0] == "entry" and v[1] == key:
                return ("found", v[2])
            pos = (pos + 1) % self.n                 # foreign key, dead slot, or a claim (current batch / round-3 rule): move on


def run(seed, resolved_claims, n_batches=60, keys=40, max_delay=3):
    rng = random.Random(seed)
    t = Table(1031, rng, max_delay, resolved_claims)
    created = {}   # key -> ids created for it
    next_id = 0
    for b in range(1, n_batches + 1):
        wanted = rng.sample(range(keys), rng.randint(1, 8))  # distinct keys of the batch (the batch table dedups lanes)
        claimed = {}
        for k in wanted:
            r = t.lookup(k, b)
            if r[0] == "claimed":
                claimed[k] = r[1]
            else:
                assert r[1] in created[k]
        for k, slot in claimed.items():              # the commit: some units are cut
            if rng.random() < 0.6:
                created.setdefault(k, []).append(next_id)
                t.post(slot, ("entry", k, next_id))
                next_id += 1
            elif resolved_claims:
                t.post(slot, DEAD)                   # (round 3 left the claim in place)
        t.tick()
    return created, t.waits


def test_round4_protocol_never_creates_a_state_twice_whatever_the_store_delays():
    waited = 0
    for seed in range(300):
        created, waits = run(seed, resolved_claims=True)
        assert all(len(ids) == 1 for ids in created.values()), (seed, created)
        waited += waits
    assert waited > 0  # (the delays really made look-ups meet claims of earlier batches)


def test_round3_protocol_does_under_delayed_stores_and_not_without():
    dup = sum(any(len(ids) > 1 for ids in run(seed, resolved_claims=False)[0].values()) for seed in range(300))
    assert dup > 0      # the failure round 4 found on the device, reproduced in the model
    quiet = sum(any(len(ids) > 1 for ids in run(seed, resolved_claims=False, max_delay=0)[0].values()) for seed in range(300))
    assert quiet == 0   # ... and why hundreds of quiet runs never showed it
