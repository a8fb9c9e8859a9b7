# Development: run tools/ab_analytic.py for the product library and every tools/ab/libbkcore_ana_*.so (one process each)
R=$GRAFT_REPO_ROOT
timeout -k 10 200 python3 $R/tools/ab_analytic.py || exit 1
for f in $R/tools/ab/libbkcore_ana_*.so; do
  [ -e "$f" ] || continue
  timeout -k 10 200 python3 $R/tools/ab_analytic.py --lib $f || exit 1
done
