"""Development: per-kernel means of rocprofv3 counter_collection.csv files (kernels named bk_*)."""
import csv, re, sys, collections
agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"(bk_\w+)", r["Kernel_Name"])
        if m and m.group(1) in ("bk_ratio_forward_kernel", "bk_adjoint_sweeps", "bk_backward_ref_kernel"):
            agg[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, k), v in sorted(agg.items()):
    print(f"{kn:26s} {k:24s} n={len(v)} mean={sum(v) / len(v):.5g}")
