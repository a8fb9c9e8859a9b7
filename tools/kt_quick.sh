# Development: kernel-trace averages of a short python command, on the GPU box:  bash tools/kt_quick.sh <tag> <script and args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $R/"$@" > $out/kt.log 2>&1 || exit 1
cd $R
python3 - $out <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
P
