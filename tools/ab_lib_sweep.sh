# Development: forward time of the in-tree library against another build over shapes.
# On the GPU box: bash tools/ab_lib_sweep.sh tools/ab/libbkcore_base.so "<N:rows:dtype ...>"
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/libsweep
for spec in ${2:-32768:1024:f32 32768:600:f32 32768:1300:f32 32768:2048:f32 32768:1024:bf16 32768:700:bf16 1048576:16:f32 131072:256:f32}; do
  n=${spec%%:*}; rest=${spec#*:}; rows=${rest%%:*}; dt=${rest#*:}
  timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --n $n --dtype $dt --lib $R/$1 --tag other > $R/gpurun_out/libsweep/o.log 2>&1 || exit 1
  timeout -k 10 120 python3 $R/tools/ab_forward.py --rows $rows --n $n --dtype $dt --tag tree > $R/gpurun_out/libsweep/t.log 2>&1 || exit 1
  python3 - $R/gpurun_out/libsweep/o.log $R/gpurun_out/libsweep/t.log <<'P'
import json, sys
r = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
print(r[0]["dtype"], r[0]["n"], r[0]["rows"], "other", r[0]["fwd_ms_med"], "in-tree", r[1]["fwd_ms_med"], "two outputs", r[0]["fwd2_ms_med"], r[1]["fwd2_ms_med"], "err", r[1].get("err_headline"), flush=True)
P
done
