# Development: forward time of the library's own choice against one workgroup per chunk (BKCORE_FORCE_CHUNKS=1).
# On the GPU box: bash tools/ab_default_vs_chunks.sh "<row counts>" "<dtypes>"
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/dvc
for dt in ${2:-f32 bf16}; do
  for rows in ${1:-600 800 900 1152 1300 1400 1536 2000}; do
    timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --n ${AB_N:-32768} --dtype $dt --tag default > $R/gpurun_out/dvc/d_${dt}_$rows.log 2>&1 || exit 1
    BKCORE_FORCE_CHUNKS=1 timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --n ${AB_N:-32768} --dtype $dt --tag chunks > $R/gpurun_out/dvc/c_${dt}_$rows.log 2>&1 || exit 1
    python3 - $R/gpurun_out/dvc/d_${dt}_$rows.log $R/gpurun_out/dvc/c_${dt}_$rows.log <<'P'
import json, sys
r = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
print(r[0]["dtype"], r[0]["rows"], "default", r[0]["fwd_ms_med"], "chunks", r[1]["fwd_ms_med"], "two outputs", r[0]["fwd2_ms_med"], r[1]["fwd2_ms_med"], flush=True)
P
  done
done
