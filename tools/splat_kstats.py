"""Per-kernel average durations from rocprofv3 --kernel-trace --stats CSVs (one line per run directory)."""
import csv
import glob
import sys

for d in sys.argv[1:]:
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    short = {}
    for r in rows:
        n = r["Name"]
        n = n.split("::")[1].split("(")[0] if "anonymous" in n else n.split("(")[0]
        short[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    a = [k for k in short if k.startswith("cells_pass_kernel<false")]
    b = [k for k in short if k.startswith("cells_pass_kernel<true")]
    keys = ["cells_seed_classify_kernel"] + a[:1] + ["cells_hiz_kernel", "cells_merge_hiz_kernel"] + b[:1] + ["splat_resolve_kernel"]
    tot = sum(short[k][1] for k in keys if k in short)
    print("%-28s " % d.rstrip("/").split("/")[-1] + "  ".join("%s %.1f" % (k.replace("cells_", "").replace("_kernel", "").replace("splat_", ""),
                                                                      short[k][1]) for k in keys if k in short) + "  | sum %.1f us" % tot)
