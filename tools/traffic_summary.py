"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).

    python tools/traffic_summary.py <dir_FETCH> <dir_WRITE> [--md out.md] [--json out.json]

FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B (rocprofv3 derived counters).  On gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced stream (MI355X_MICROARCH.md §HBM); the factor is
re-derived here from a kernel with a known byte count (texture_to_rows: reads and writes N*C*4 bytes).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def load(d):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]))
        rows[key] = rows.get(key, 0.0) + float(r["Counter_Value"])
        rows[(key, "dur")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--md", default="")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    F, Wr = load(a.fetch_dir), load(a.write_dir)
    agg = collections.OrderedDict()
    for src, col in ((F, "fetch"), (Wr, "write")):
        for k, v in src.items():
            if isinstance(k[0], tuple):
                continue
            _, name, grid = k
            g = agg.setdefault((short(name), grid), {"fetch": 0.0, "write": 0.0, "n_fetch": 0, "n_write": 0, "dur": 0})
            g[col] += v * 1024.0
            g["n_" + col] += 1
            if col == "fetch":
                g["dur"] += src[(k, "dur")]
    # calibration on texture_to_rows (reads = writes = grid * 8 channels * 4 B, grid ~ N points)
    cal_f = cal_w = None
    for (name, grid), g in agg.items():
        if name.startswith("texture_to_rows"):
            m = re.search(r"\d+", str(grid))
            true = float(grid) * 32.0
            cal_f = true / (g["fetch"] / g["n_fetch"])
            cal_w = true / (g["write"] / g["n_write"])
    cal_f = cal_f or 2.0
    cal_w = cal_w or 1.0
    lines = [f"FETCH_SIZE calibration factor {cal_f:.3f} (guide: 2.0), WRITE_SIZE factor {cal_w:.3f}; both applied below.\n",
             "| kernel | grid | launches | avg us | read MB/launch | write MB/launch | HBM-side GB/s |", "|---|---|---|---|---|---|---|"]
    out = {}
    for (name, grid), g in agg.items():
        if not g["n_fetch"] or not g["n_write"]:
            continue
        rd = g["fetch"] / g["n_fetch"] * cal_f
        wr = g["write"] / g["n_write"] * cal_w
        us = g["dur"] / g["n_fetch"] / 1e3
        lines.append(f"| `{name}` | {grid} | {g['n_fetch']} | {us:.1f} | {rd / 1e6:.2f} | {wr / 1e6:.2f} | {(rd + wr) / us / 1e3:.0f} |")
        out[f"{name}@{grid}"] = {"launches": g["n_fetch"], "avg_us": us, "read_bytes": rd, "write_bytes": wr}
    txt = "\n".join(lines)
    print(txt)
    if a.md:
        open(a.md, "w").write(txt + "\n")
    if a.json:
        json.dump({"fetch_factor": cal_f, "write_factor": cal_w, "kernels": out}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
