"""Per-kernel MFMA-pipe utilisation and VALU share from one rocprofv3 --pmc pass (csv output):

    python tools/pmc_summary.py <dir> [--md out.md]

Counters expected: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE (and optionally SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_ACTIVE_INST_VALU,
SQ_BUSY_CYCLES).  busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); SQ_ACTIVE_INST_VALU counts
quad-cycles (x4 = cycles, MI355X_MICROARCH.md).  Rows are (kernel, grid): the same kernel at different image levels stays apart.
"""
import argparse
import collections
import csv
import glob
import os


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--md", default="")
    ap.add_argument("--match", default="gated_conv")
    ap.add_argument("--chunk", type=int, default=0, help="label every CHUNK consecutive matching dispatches as one group (a tool "
                    "that launches shape after shape, e.g. tools/sweep_conv.py --main-only --iters 2: 4 launches per level)")
    ap.add_argument("--labels", default="", help="comma list of names for the chunks")
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]),
                                                    "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    labels = [x for x in a.labels.split(",") if x]
    seen = 0
    for d in disp.values():
        if a.match not in d["name"]:
            continue
        if a.chunk:
            ci = seen // a.chunk
            d["name"] = (labels[ci] if ci < len(labels) else f"chunk{ci}") + " " + d["name"]
            seen += 1
        g = agg.setdefault((d["name"], d["grid"]), collections.defaultdict(float))
        g["n"] += 1
        for k, v in d.items():
            if k not in ("name", "grid"):
                g[k] += v
    lines = ["| kernel | grid | launches | avg µs | MFMA pipe busy % | MFMA per launch | VALU per launch | VALU per MFMA | VALU-active share of SIMD time % |",
             "|---|---|---|---|---|---|---|---|---|"]
    for (name, grid), g in agg.items():
        n = g["n"]
        gui = g["GRBM_GUI_ACTIVE"] / 8.0
        busy = 100.0 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui) if gui else float("nan")
        mf, va = g.get("SQ_INSTS_MFMA", 0.0) / n, g.get("SQ_INSTS_VALU", 0.0) / n
        vact = 100.0 * 4.0 * g.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * gui) if gui else float("nan")
        lines.append(f"| `{name}` | {grid} | {int(n)} | {g['dur'] / n / 1e3:.1f} | {busy:.1f} | {mf:.0f} | {va - mf:.0f} | "
                     f"{(va - mf) / mf if mf else float('nan'):.2f} | {vact:.1f} |")
    txt = "\n".join(lines)
    print(txt)
    if a.md:
        open(a.md, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
