"""Summarise a rocprofv3 SQLite (rocpd) result: per-kernel stats and, if present, PMC counters.

    python tools/rocpd_summary.py <results.db> [--md out.md]
"""
import argparse
import collections
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--md", default="")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    out = []
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    out.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for n, c, t, avg, pct in rows:
        out.append(f"| `{short(n)}` | {c} | {t / 1e6:.3f} | {avg / 1e3:.2f} | {pct:.2f} |")
    try:
        cc = con.execute("select kernel_name, dispatch_id, counter_name, value, duration, grid_size, workgroup_size "
                         "from counters_collection").fetchall()
    except sqlite3.Error:
        cc = []
    if cc:
        per = collections.OrderedDict()
        for kn, did, cn, v, dur, gs, ws in cc:
            d = per.setdefault((short(kn), did), {"dur": dur, "grid": gs, "wg": ws})
            d[cn] = d.get(cn, 0.0) + v
        agg = collections.OrderedDict()
        for (kn, did), d in per.items():
            g = agg.setdefault((kn, d["grid"]), collections.defaultdict(float))
            g["n"] += 1
            for k, v in d.items():
                if k not in ("grid", "wg"):
                    g[k] += v
        names = sorted({k for g in agg.values() for k in g if k not in ("n", "dur")})
        out.append("\n| kernel | grid | dispatches | avg us | " + " | ".join(names) + " |\n|---|---|---|---|" + "---|" * len(names))
        for (kn, gs), g in agg.items():
            n = g["n"]
            out.append(f"| `{kn}` | {gs} | {int(n)} | {g['dur'] / n / 1e3:.2f} | " +
                       " | ".join(f"{g[k] / n:.4g}" for k in names) + " |")
    txt = "\n".join(out)
    print(txt)
    if a.md:
        open(a.md, "w").write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
