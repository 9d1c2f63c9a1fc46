"""Executable statement of rule R2 of DESIGN.md 3.9 for the look-ahead cache: a 16-byte record names a row of heuristics; the
helper writes the row first and the record after it, but nothing orders when the two BECOME VISIBLE to the leader on another XCD.

Model: every 8-byte word of the row and of the record is a posted store that lands at a time of the adversary's choosing; the
leader polls the record, and once it sees it reads the row.  Round 3 took the row at face value (its words may still hold what
the row held before: another state's heuristics, or fresh-memory garbage); round 4 checks the row's check word -- XOR of a term
per heuristic it reads, the voxel-read count, a salt of (key hash, query, epoch) -- and reads again until it matches.  The
arithmetic mirrors mplx_spec.h (cache_row_term / cache_row_salt) in 32-bit wrap-around.
"""
import random
import struct

M32 = 0xFFFFFFFF


def row_term(h, lu):
    b = struct.unpack("<Q", struct.pack("<d", h))[0]
    return ((((b & M32) ^ (b >> 32)) + lu) * 0x9E3779B1) & M32


def row_salt(khash, q, epoch, reads):
    return (khash ^ ((q * 0x85EBCA77) & M32) ^ ((epoch * 0xC2B2AE3D) & M32) ^ ((reads * 0x27D4EB2F) & M32) ^ 0xA5A5A5A5) & M32


def check_word(hs, act, khash, q, epoch, reads):
    cs = 0
    for lu, (h, a) in enumerate(zip(hs, act)):
        if a:
            cs ^= row_term(h, lu)
    return cs ^ row_salt(khash, q, epoch, reads)


def one_exchange(rng, checked):
    """A helper publishes one row; returns (what the leader consumed, what the helper wrote)."""
    n_u = 27
    act = [rng.random() < 0.45 for _ in range(n_u)]
    new = [rng.uniform(0.0, 300.0) if a else 0.0 for a in act]
    old = [rng.choice([float("nan"), rng.uniform(0.0, 300.0), 0.0]) for _ in range(n_u)]  # what the row's memory held before
    khash, q, epoch, reads = rng.getrandbits(32), rng.randrange(1024), rng.randrange(1, 1 << 20), rng.randrange(1, 4000)
    cw = check_word(new, act, khash, q, epoch, reads)
    # landing times: the record at time 0 (the leader has just seen it), every word of the row somewhere around it
    lands = [rng.choice([-1, -1, -1, rng.randint(0, 6)]) for _ in range(n_u + 1)]  # (-1: already there; last = the reads | check word slot)
    old_slot = (rng.randrange(1, 4000), rng.getrandbits(32))

    def read_row(t):
        hs = [new[i] if lands[i] <= t else old[i] for i in range(n_u)]
        rd, stored = (reads, cw) if lands[n_u] <= t else old_slot
        return hs, rd, stored

    t = 0
    while True:
        hs, rd, stored = read_row(t)
        if not checked or check_word(hs, act, khash, q, epoch, rd) == stored:
            return [h for h, a in zip(hs, act) if a], [h for h, a in zip(new, act) if a], rd, reads
        t += 1
        assert t < 50


def same(a, b):
    return struct.pack(f"<{len(a)}d", *a) == struct.pack(f"<{len(b)}d", *b)


def test_checked_rows_are_always_the_rows_the_helper_wrote():
    rng = random.Random(4)
    for _ in range(4000):
        got, want, rd, reads = one_exchange(rng, checked=True)
        assert same(got, want) and rd == reads


def test_unchecked_rows_are_sometimes_stale_when_stores_land_late():
    rng = random.Random(4)
    stale = sum(not same(*one_exchange(rng, checked=False)[:2]) for _ in range(4000))
    assert stale > 100  # what round 4 saw on the device as "another expansion order" / "OPEN runs dry"


def test_check_word_arithmetic_pins():
    # known answers of the 32-bit arithmetic (a change of the constants in mplx_spec.h must change these)
    assert row_term(1.0, 0) == ((0x3FF00000) * 0x9E3779B1) & M32
    assert row_term(0.0, 5) == (5 * 0x9E3779B1) & M32
    assert row_salt(0, 0, 0, 0) == 0xA5A5A5A5
    assert check_word([2.5, 0.0], [True, False], 0x12345678, 7, 3, 100) == row_term(2.5, 0) ^ row_salt(0x12345678, 7, 3, 100)
