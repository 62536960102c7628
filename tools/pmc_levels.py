"""Per-level table of the dominant kernel from two rocprofv3 --pmc passes over `tools/sweep_conv.py --main-only` (csv output):

    python tools/pmc_levels.py <dir pass 1: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE> <dir pass 2: SQ_INSTS_VALU, SQ_INSTS_MFMA> [kernel substring]

The sweep runs the four C -> C shapes in level order, the same number of launches each: dispatches of the kernel are cut into four
equal runs.  busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) (MI355X_MICROARCH.md).  Prints the markdown
rows `bench.py` reads back (profiles/r6_pmc_mfma_busy.md)."""
import collections
import csv
import glob
import os
import sys


def load(d, sub):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            e = out.setdefault(int(r["Dispatch_Id"]), {"grid": int(r["Grid_Size"]), "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                        "name": r["Kernel_Name"]})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(out.values())


def main():
    sub = sys.argv[3] if len(sys.argv) > 3 else "gated_conv_wino4h_kernel"
    a, b = load(sys.argv[1], sub), load(sys.argv[2], sub)
    shapes = ["L0 C=32 352x1216", "L1 C=64 176x608", "L2 C=128 88x304", "L3 C=256 44x152"]
    n, m = len(a) // 4, len(b) // 4
    print("| kernel | grid | launches | avg µs | MFMA pipe busy % | MFMA per launch | VALU per launch | VALU per MFMA |")
    print("|---|---|---|---|---|---|---|---|")
    for i, sh in enumerate(shapes):
        ra, rb = a[i * n:(i + 1) * n], b[i * m:(i + 1) * m]
        us = sum(r["us"] for r in ra) / len(ra)
        busy = 100.0 * sum(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * r["GRBM_GUI_ACTIVE"] / 8.0) for r in ra) / len(ra)
        mf = sum(r["SQ_INSTS_MFMA"] for r in rb) / len(rb)
        va = sum(r["SQ_INSTS_VALU"] for r in rb) / len(rb)
        kn = ra[0]["name"].replace("(anonymous namespace)::", "").replace("void ", "")
        kn = kn[:kn.find("(")] if "(" in kn else kn
        print("| `%s %s` | %d | %d | %.1f | %.1f | %d | %d | %.2f |" % (sh, kn, ra[0]["grid"], len(ra), us, busy, mf, va, va / mf))


if __name__ == "__main__":
    main()
