# Development: forward time with whole rows (BKCORE_RELAY_UNITS=0) against one workgroup per chunk (BKCORE_FORCE_CHUNKS=1)
# at row counts where the library by itself would cut rows into units.  On the GPU box: bash tools/ab_whole_vs_chunks.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/wvc
for dt in f32 bf16; do
  for rows in 800 1024 1300 1500; do
    BKCORE_RELAY_UNITS=0 timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --dtype $dt --tag whole > $R/gpurun_out/wvc/w_${dt}_$rows.log 2>&1 || exit 1
    BKCORE_FORCE_CHUNKS=1 timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --dtype $dt --tag chunks > $R/gpurun_out/wvc/c_${dt}_$rows.log 2>&1 || exit 1
    python3 - $R/gpurun_out/wvc/w_${dt}_$rows.log $R/gpurun_out/wvc/c_${dt}_$rows.log <<'P'
import json, sys
r = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
print(r[0]["dtype"], r[0]["rows"], "whole", r[0]["fwd_ms_med"], "chunks", r[1]["fwd_ms_med"], "two outputs", r[0]["fwd2_ms_med"], r[1]["fwd2_ms_med"], flush=True)
P
  done
done
