import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "bk_forward" in r["Kernel_Name"] or "bk_backward" in r["Kernel_Name"]:
        nm = "fwd" if "bk_forward" in r["Kernel_Name"] else "bwd"
        agg[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n=%d" % len(v), "max=%.4g" % max(v), "mean=%.4g" % (sum(v) / len(v)))
