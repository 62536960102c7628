"""GPU timeline of the rasteriser from a rocprofv3 --kernel-trace csv: per kernel the mean duration, per FRAME (one resolve kernel
= one frame) the busy time (sum of kernel durations), the span (first start -> last end), the idle time inside the frame and the
gap to the next frame — i.e. whether a measured ms/frame is the device's or the host's.

    python tools/splat_timeline.py <dir with *kernel_trace.csv> [--skip 20]
"""
import argparse
import collections
import csv
import glob
import os

import numpy as np


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.find("(")] if "(" in n else n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip", type=int, default=20, help="frames to skip at the start (warm-up)")
    ap.add_argument("--take", type=int, default=0, help="frames to use after the skipped ones (0: all)")
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows = sorted(r for r in rows if r[2].startswith(("cells_", "splat_")))
    frames, cur = [], []
    for r in rows:
        cur.append(r)
        if "resolve" in r[2]:
            frames.append(cur)
            cur = []
    frames = frames[a.skip:]
    if a.take:
        frames = frames[:a.take]
    # frames of the counter-collecting (read_tuning_set("splat_stats", 1)) laps of a tool run different kernels: not the product's
    frames = [fr for fr in frames if not any("<false, true" in n or "<true, true" in n for _, _, n in fr)]
    per = collections.defaultdict(list)
    busy, span, idle, gap, nk = [], [], [], [], []
    for i, fr in enumerate(frames):
        for s, e, n in fr:
            per[n].append((e - s) / 1e3)
        b = sum(e - s for s, e, _ in fr) / 1e3
        sp = (fr[-1][1] - fr[0][0]) / 1e3
        busy.append(b)
        span.append(sp)
        idle.append(sp - b)
        nk.append(len(fr))
        if i + 1 < len(frames):
            gap.append((frames[i + 1][0][0] - fr[-1][1]) / 1e3)
    print(f"{len(frames)} frames, {np.mean(nk):.2f} kernels per frame")
    for n, v in per.items():
        print(f"  {n:60s} {np.mean(v):7.2f} us  x{len(v) / len(frames):.2f} per frame")
    print(f"per frame: busy {np.mean(busy):.1f} us   span {np.mean(span):.1f} us   idle inside {np.mean(idle):.1f} us   "
          f"gap to the next frame {np.mean(gap):.1f} us (median {np.median(gap):.1f})   frame period {np.mean(span) + np.mean(gap):.1f} us")


if __name__ == "__main__":
    main()
