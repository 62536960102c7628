"""Per-kernel averages of whatever counters one rocprofv3 --pmc pass (csv) collected:  python tools/pmc_dump.py <dir> [match ...]"""
import collections
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
match = sys.argv[2:] or [""]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[:n.find("(")] if "(" in n else n
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": n, "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
for d in disp.values():
    if not any(m in d["name"] for m in match):
        continue
    g = agg.setdefault(d["name"], collections.defaultdict(float))
    g["n"] += 1
    for k, v in d.items():
        if k != "name":
            g[k] += v
for name, g in agg.items():
    n = g["n"]
    print("%-70s x%-4d %8.1f us  " % (name[:70], n, g["dur"] / n / 1e3) + "  ".join("%s %.3g" % (k, v / n) for k, v in g.items() if k not in ("n", "dur")))
