mkdir -p gpurun_out/r06p
for f in 8 64 512 4096 32768; do
  MPLX_PLPA_BUCKET_FACTOR=$f timeout 100 python bench.py --config plpa --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/r06p/plpa_f$f.json 2>/dev/null
  MPLX_LPA_BUCKET_FACTOR=$f timeout 100 python bench.py --config lpa --map 256 --steps 2 --warmup 1 --cpu-seconds 1 > gpurun_out/r06p/lpa_f$f.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06p/plpa_f$f.json")); a=(round(d["value"],3), [round(r["lpa_kernel_ms"],3) for r in d["replans"]])
except Exception as e: a="failed %s"%e
try:
    d=json.load(open("gpurun_out/r06p/lpa_f$f.json")); b=([ (round(r["lpa_ms"],2), round(r.get("update_ms",0),2)) for r in d["cycle"]], d.get("parity_sample",{}).get("mismatches"))
except Exception as e: b="failed %s"%e
print("factor $f plpa", a, "lpa", b)
PY
done
