# Development: 16-bit operands, whole rows against one workgroup per chunk over row counts around multiples of the 512
# places whole rows of that type have (two workgroups per CU).  On the GPU box: bash tools/ab_bf16_rows.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/bf16rows
for rows in 480 512 560 900 980 1024 1080 1300 1536 2048; do
  BKCORE_FORCE_CHUNKS=-1 timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --dtype bf16 --iters 50 --tag whole > $R/gpurun_out/bf16rows/w_$rows.log 2>&1 || exit 1
  BKCORE_FORCE_CHUNKS=1 timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --dtype bf16 --iters 50 --tag chunks > $R/gpurun_out/bf16rows/c_$rows.log 2>&1 || exit 1
  python3 - $R/gpurun_out/bf16rows/w_$rows.log $R/gpurun_out/bf16rows/c_$rows.log <<'P'
import json, sys
r = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
print(r[0]["dtype"], r[0]["rows"], "whole", r[0]["fwd_ms_med"], "chunks", r[1]["fwd_ms_med"], flush=True)
P
done
