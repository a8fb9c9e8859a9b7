import csv, sys, collections
agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "bk_ratio" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n=%d" % len(v), "mean=%.5g" % (sum(v) / len(v)))
