# Development: run tools/ab_forward.py over every library under tools/ab/ (one process each); lines to gpurun_out/$1.jsonl
mkdir -p gpurun_out/r2
out=gpurun_out/r2/$1.jsonl
: > $out
for lib in tools/ab/libbkcore_*.so; do
  tag=$(basename $lib .so | sed 's/libbkcore_//')
  timeout -k 10 120 python tools/ab_forward.py --lib $lib --tag $tag --iters 60 >> $out 2>> gpurun_out/r2/$1.err || { echo "$tag failed" >> $out; }
done
cat $out
