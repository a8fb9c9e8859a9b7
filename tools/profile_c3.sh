# Development: rocprofv3 evidence for the long-context configuration (B = 16, N = 2^20), on the GPU box:
#   bash tools/profile_c3.sh r03_c3   ->  gpurun_out/r03_c3/{kernel_stats.csv, pmc_fetch_write_summary.json, c3.log}
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $R/tools/prof_c3.py --iters 200 > $out/c3.log 2>&1 &&
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $R/tools/prof_c3.py > $out/kt.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_f -o f --output-format csv -- python3 $R/tools/prof_c3.py --iters 5 > $out/pmc_f.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_w -o w --output-format csv -- python3 $R/tools/prof_c3.py --iters 5 > $out/pmc_w.log 2>&1
cd $R
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 tools/pmc_traffic.py $out/pmc_fetch_write_summary.json 16 1048576 $(find $out/pmc_f $out/pmc_w -name "*counter_collection.csv")
rm -f $out/traffic_latest.json $out/traffic_analytic_latest.json
grep "forward B" $out/c3.log
head -3 $out/kernel_stats.csv | cut -c1-60,260-400
