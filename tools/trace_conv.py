"""Per-workgroup timeline of one gated-conv launch (debug; run on the GPU box).

    python tools/trace_conv.py --shape L0 --config 0 --out gpurun_out/trace_L0_c0.npz
"""
import os
os.environ.setdefault("READ_HIP_DEBUG", "1")   # the probes live in libreadhip_debug.so only (python -m read_amd.build --debug)
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                                    # noqa: E402
from read_amd.gated_conv import PackedGatedConv, config_names, gated_conv   # noqa: E402

SHAPES = {"L0": (32, 352, 1216), "L1": (64, 176, 608), "L2": (128, 88, 304), "L3": (256, 44, 152)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="L0")
    ap.add_argument("--configs", default="0")
    ap.add_argument("--out", default="gpurun_out/trace")
    ap.add_argument("--tune", default="", help="key=value[,key=value] for read_tuning_set")
    a = ap.parse_args()
    C, H, W = SHAPES[a.shape]
    st = synthetic.make_unet_state([("L", C, C, 3)], 1)
    b = "L.block."
    pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                         st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"],
                         st[b + "norm.running_var"])
    x = torch.randn(H, W, C, device="cuda")
    res = torch.randn(H, W, C, device="cuda")
    out = torch.empty(H, W, C, device="cuda")
    L = _lib.lib()
    for kv in filter(None, a.tune.split(",")):
        _lib.check(L.read_tuning_set(kv.split("=")[0].encode(), int(kv.split("=")[1])))
    names = config_names()
    for ci in [names.index(c) if not c.isdigit() else int(c) for c in a.configs.split(",")]:
        for _ in range(3):
            gated_conv(pk, [(x, 0)], elu=True, residual=res, config=ci, out=out)
        torch.cuda.synchronize()
        buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")          # 8 MiB = 131072 workgroups
        _lib.check(L.read_debug_set_trace(buf.data_ptr(), buf.numel() * 8))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gated_conv(pk, [(x, 0)], elu=True, residual=res, config=ci, out=out)
        e1.record()
        torch.cuda.synchronize()
        _lib.check(L.read_debug_set_trace(None, 0))
        rec = buf.cpu().numpy().reshape(-1, 8)
        rec = rec[rec[:, 0] != 0]
        t0 = rec[:, 0].min()
        np.savez_compressed(f"{a.out}_{a.shape}_c{ci}.npz", rec=rec, name=names[ci], ms=e0.elapsed_time(e1))
        if "wino" in names[ci]:                  # persistent kernel: one record per wave, rec[2] = ticks in unit epilogues
            ep = rec[:, 2] * 0.01
            tot = (rec[:, 6] - rec[:, 0]) * 0.01
            pro = (rec[:, 1] - rec[:, 0]) * 0.01
            C_, H_, W_ = SHAPES[a.shape]
            units = -(-H_ // 8) * -(-W_ // 16) * (C_ // 32)
            per = units / (len(rec) / 4)
            print(f"{a.shape} {names[ci]}: {len(rec) // 4} workgroups, {units} units ({per:.2f}/workgroup), kernel "
                  f"{e0.elapsed_time(e1) * 1e3:.1f} us (event), first start -> last exit {(rec[:, 3].max() - t0) * 0.01:.1f} us")
            for nm, d in (("prologue", pro), ("epilogues", ep), ("loop", tot - pro - ep), ("total", tot)):
                print("    %-9s p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (nm, *np.percentile(d, [10, 50, 90]), d.max()))
            print("    epilogue per unit (medians, us): 3rd barrier passed at %.2f, LDS reads landed %.2f, loads landed %.2f, end %.2f" % (
                np.median(rec[:, 3]) * 0.01 / per, np.median(rec[:, 4]) * 0.01 / per, np.median(rec[:, 5]) * 0.01 / per,
                np.median(ep) / per))
            print("    per unit: loop %.2f us, epilogue %.2f us ; per chunk %.2f us" % (
                np.median(tot - pro - ep) / per, np.median(ep) / per, np.median(tot - pro - ep) / per / (C_ // 16)))
            continue
        tt = (rec[:, :4] - t0) * 0.01            # us (100 MHz)
        print(f"{a.shape} config {ci} {names[ci]}: {len(rec)} WGs, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (event), "
              f"last exit {tt[:, 3].max():.1f} us")
        print("  per-WG us: prologue %.2f  loop %.2f  epilogue %.2f  total %.2f (medians)" % (
            np.median(tt[:, 1] - tt[:, 0]), np.median(tt[:, 2] - tt[:, 1]), np.median(tt[:, 3] - tt[:, 2]),
            np.median(tt[:, 3] - tt[:, 0])))
        for nm, d in (("prologue", tt[:, 1] - tt[:, 0]), ("loop", tt[:, 2] - tt[:, 1]), ("epilogue", tt[:, 3] - tt[:, 2])):
            print("    %-8s p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (nm, *np.percentile(d, [10, 50, 90]), d.max()))
        # concurrency over time
        ev = np.concatenate([np.stack([tt[:, 0], np.ones(len(tt))], 1), np.stack([tt[:, 3], -np.ones(len(tt))], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        conc = np.cumsum(ev[:, 1])
        dur = np.diff(ev[:, 0], append=ev[-1, 0])
        print("  mean resident WGs %.1f ; time with <256 WGs resident: %.1f us ; start spread: first %.2f last-start %.2f us" % (
            (conc * dur).sum() / max(dur.sum(), 1e-9), dur[conc < 256].sum(), tt[:, 0].min(), tt[:, 0].max()))
        # per CU slot sequences: gaps between consecutive WGs on the same (xcc, hw cu/se/sh)
        key = (rec[:, 5] << 32) | (rec[:, 4] & 0xFF00)      # xcc + (cu_id, sh_id, se_id) bits 8..15
        gaps = []
        for k in np.unique(key):
            r = tt[key == k]
            r = r[np.argsort(r[:, 0])]
            # with 2+ WGs resident per CU the "gap" is between an exit and the next entry on that CU
            exits = np.sort(r[:, 3])
            starts = np.sort(r[:, 0])
            n_slots = int((starts < exits[0]).sum()) if len(exits) else 0
            for i in range(n_slots, len(starts)):
                gaps.append(starts[i] - exits[i - n_slots])
        if gaps:
            print("  CU turnover gap (exit -> next entry on that CU): median %.2f us, p90 %.2f us, n=%d" % (
                np.median(gaps), np.percentile(gaps, 90), len(gaps)))


if __name__ == "__main__":
    main()
