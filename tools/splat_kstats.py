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
    keys = ["cells_seed_classify_kernel", "cells_pass_kernel<false, false, false>", "cells_hiz_kernel", "cells_pass_kernel<true, false, false>",
            "splat_resolve_kernel"]
    tot = sum(short[k][1] for k in keys if k in short)
    print("%-28s " % d.rstrip("/").split("/")[-1] + "  ".join("%s %.1f" % (k.replace("cells_", "").replace("_kernel", "").replace("splat_", ""),
                                                                      short[k][1]) for k in keys if k in short) + "  | sum %.1f us" % tot)
