"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv files given on the command
line) per kernel and writes the corrected HBM traffic of the dominant forward kernel.
FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads: doubled here.  Units: KiB in the CSV."""
import collections, csv, json, re, sys

out_path, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
agg = collections.defaultdict(list)
for path in sys.argv[4:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"(bk_\w+)", r["Kernel_Name"])
        if not m:
            continue
        name = m.group(1)
        if name == "bk_forward_kernel":
            name += "<second launch>" if "false" in r["Kernel_Name"].split("<")[1].split(",")[2] else "<scan first launch>"
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
summ = {}
for (name, ctr), v in sorted(agg.items()):
    summ[f"{name}|{ctr}"] = {"launches": len(v), "mean_KiB": sum(v) / len(v), "max_KiB": max(v)}
fetch = summ.get("bk_ratio_forward_kernel|FETCH_SIZE", {}).get("mean_KiB", 0.0)
write = summ.get("bk_ratio_forward_kernel|WRITE_SIZE", {}).get("mean_KiB", 0.0)
total = (2.0 * fetch + write) * 1024.0
afetch = summ.get("bk_adjoint_sweeps|FETCH_SIZE", {}).get("mean_KiB", 0.0)
awrite = summ.get("bk_adjoint_sweeps|WRITE_SIZE", {}).get("mean_KiB", 0.0)
atotal = (2.0 * afetch + awrite) * 1024.0
res = {"counters": summ,
       "note": "FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads; doubled. WRITE_SIZE is exact.",
       "forward_traffic_bytes_per_launch_corrected": total,
       "forward_traffic_bytes_per_elem_corrected": total / (B * N),
       "analytic_traffic_bytes_per_launch_corrected": atotal,
       "analytic_traffic_bytes_per_elem_corrected": atotal / (B * N)}
json.dump(res, open(out_path, "w"), indent=1)
# the two small files bench.py reads (copied to profiles/ by hand with the rest of the summary)
import os
d = os.path.dirname(out_path)
src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per the gfx950 correction"
json.dump({"rows": B, "seq_len": N, "bytes_per_launch": total, "bytes_per_elem": total / (B * N),
           "kernel": "bk_ratio_forward_kernel", "source": src}, open(os.path.join(d, "traffic_latest.json"), "w"), indent=1)
json.dump({"rows": B, "seq_len": N, "analytic_traffic_bytes_per_launch_corrected": atotal,
           "analytic_traffic_bytes_per_elem_corrected": atotal / (B * N), "kernel": "bk_adjoint_sweeps", "source": src},
          open(os.path.join(d, "traffic_analytic_latest.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("forward_traffic_bytes_per_launch_corrected", "forward_traffic_bytes_per_elem_corrected")}))
