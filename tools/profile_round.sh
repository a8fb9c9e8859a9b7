# Development: the round's rocprofv3 evidence.  usage (on the GPU box): bash tools/profile_round.sh r02_v1
# Leaves under gpurun_out/$1: kernel_stats.csv (kernel trace of bench.py), bench.json, bench_under_rocprof.json,
# pmc_fetch_write_summary.json (FETCH_SIZE and WRITE_SIZE in separate passes over tools/prof_fwd.py and
# tools/prof_ana.py), kernel_stats_analytic.csv (kernel trace of the analytic backward on bench.py's inputs),
# kernel_stats_analytic_states.csv (the same, starting from the boundary states the forward kept).
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $out/bench.json 2> $out/bench.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 100 > $out/bench_under_rocprof.json 2> $out/kt.log &&
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_f -o f --output-format csv -- python3 $R/tools/prof_fwd.py > $out/pmc_f.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_w -o w --output-format csv -- python3 $R/tools/prof_fwd.py > $out/pmc_w.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $out/pa_f -o f --output-format csv -- python3 $R/tools/prof_ana.py 3 > $out/pa_f.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $out/pa_w -o w --output-format csv -- python3 $R/tools/prof_ana.py 3 > $out/pa_w.log 2>&1 &&
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/kt_ana -o kt --output-format csv -- python3 $R/tools/prof_ana.py 300 > $out/kt_ana.log 2>&1 &&
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/kt_ana_st -o kt --output-format csv -- python3 $R/tools/prof_ana.py 300 states > $out/kt_ana_st.log 2>&1
cd $R
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cp $(find $out/kt_ana -name "*kernel_stats.csv" | head -1) $out/kernel_stats_analytic.csv
cp $(find $out/kt_ana_st -name "*kernel_stats.csv" | head -1) $out/kernel_stats_analytic_states.csv
python3 tools/pmc_traffic.py $out/pmc_fetch_write_summary.json 1024 32768 $(find $out/pmc_f $out/pmc_w $out/pa_f $out/pa_w -name "*counter_collection.csv")
head -8 $out/kernel_stats.csv | cut -c1-200
