# Development: HBM traffic of the analytic backward's kernels (FETCH_SIZE / WRITE_SIZE in separate passes).
# usage (GPU box): bash tools/pmc_ana.sh TAG  ->  gpurun_out/TAG/pmc_ana_summary.json
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $out/pa_f -o f --output-format csv -- python3 $R/tools/prof_ana.py 3 > $out/pa_f.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $out/pa_w -o w --output-format csv -- python3 $R/tools/prof_ana.py 3 > $out/pa_w.log 2>&1
cd $R
python3 tools/pmc_traffic.py $out/pmc_ana_summary.json 1024 32768 $(find $out/pa_f $out/pa_w -name "*counter_collection.csv") > /dev/null
python3 - <<PY
import json
d = json.load(open("$out/pmc_ana_summary.json"))["counters"]
n = 1024 * 32768
for k, v in d.items():
    f = 2.0 if k.endswith("FETCH_SIZE") else 1.0
    print(f"{k:50s} launches {v['launches']:3d}  {f * v['mean_KiB'] * 1024 / n:8.2f} B/elem")
PY
