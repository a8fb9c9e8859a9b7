for u in 0 2; do
  BKCORE_RELAY_UNITS=$u timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null > gpurun_out/ab_bench_$u.json
done
python - <<'PY'
import json
for u in (0, 2):
    d = json.loads(open("gpurun_out/ab_bench_%d.json" % u).readline())
    print("units", u, round(d["ms_per_step"], 4), round(d["fwd_ms"], 4), round(d["bwd_ms"], 4))
PY
