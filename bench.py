#!/usr/bin/env python
"""Headline benchmark: rendered frames/s at 1216x352 on a synthetic 30 M-point cloud.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one full frame of READ's render path on one GPU: rasterise 5 scales (one pass over the
cloud) -> gather 8-channel descriptors -> 99-conv gated UNet -> RGBA frame.  Inputs (xyz,
descriptors, packed weights) are resident in HBM before the timed region; each step uses the next
camera pose of the novel-view sweep.  With N ranks every rank renders its own
poses (weak scaling: K frames per GPU) and the finished frames are all-gathered over RCCL — the
only exchange the path has (§8e).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from read_amd import camera, synthetic          # noqa: E402
from read_amd.frame import FrameRenderer        # noqa: E402
from read_amd.unet import weight_spec           # noqa: E402

W, H = 1216, 352
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--points", type=int, default=30_000_000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=1)
    p.add_argument("--detail", type=str, default="", help="write per-launch timings to this JSON file")
    p.add_argument("--tune", type=str, default="", help="comma list of key=value for read_tuning_set (A/B runs)")
    return p.parse_args()


def hip_time_ms(fn, iters):
    """Average duration of fn() measured with HIP events on the current stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def profiled_traffic():
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes
    (profiles/r1_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE x2 as
    MI355X_MICROARCH.md prescribes for gfx950, factor re-derived there from a kernel of known byte count).
    Launch-weighted mean over every launch of the 3x3/s1 kernels (the 73 C->C launches of a frame plus the
    three small SCM 3x3 layers and the 32->3 output layer, which share the kernel name)."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    try:
        ks = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    n = tot = 0
    for name, v in ks.items():
        if name.startswith("gated_conv_wino_kernel") or (name.startswith("gated_conv") and "<3, 1, 16" in name):
            n += v["launches"]
            tot += (v["read_bytes"] + v["write_bytes"]) * v["launches"]
    return tot / n if n else None


def cpu_baseline(xyz, desc, state, proj, frames):
    """The oracle (CPU restatement of the reference path) on this box's host cores: bounded sample (one frame)."""
    import oracle
    from oracle import unet_torch
    # 32 threads: on the 256-thread GPU hosts torch's CPU convolutions are ~30x SLOWER with all threads than with 32
    # (the UNet is ~600 small ops; measured 148 s/frame at 256 threads), so the fair baseline caps the thread count.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    t_r = t_g = t_u = 0.0
    for k in range(frames):
        M = camera.total_matrix(proj, synthetic.sweep_pose(k))[0]
        t0 = time.perf_counter()
        idx, _ = oracle.raster_multiscale(xyz, M, W, H, 5, threads=cores)
        t1 = time.perf_counter()
        with torch.no_grad():
            feats = [unet_torch.point_texture_forward(desc[None], i[None]) for i in idx]
            t2 = time.perf_counter()
            unet_torch.unet_forward(state, *feats[:4])
        t3 = time.perf_counter()
        t_r, t_g, t_u = t_r + (t1 - t0), t_g + (t2 - t1), t_u + (t3 - t2)
    per = (t_r + t_g + t_u) / frames
    return {"value": 1.0 / per, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{frames} full frame(s) (1216x352, {xyz.shape[0]} pts) on {cores} threads of {os.cpu_count()}: oracle "
                      f"raster C/OpenMP + torch-CPU gather + torch-CPU fp32 UNet",
            "ms_raster": 1e3 * t_r / frames, "ms_gather": 1e3 * t_g / frames, "ms_unet": 1e3 * t_u / frames}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.tune:
        from read_amd import _lib
        for kv in a.tune.split(","):
            k, v = kv.split("=")
            _lib.check(_lib.lib().read_tuning_set(k.encode(), int(v)), "read_tuning_set")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    # ---- scene: identical on every rank.  Rank 0 builds it; the others receive it over RCCL/xGMI
    # (one-time broadcast of xyz + descriptors + packed weights, §8e).
    N = a.points
    state = synthetic.make_unet_state(weight_spec())
    if rank == 0:
        xyz = synthetic.make_cloud(N)
        desc = synthetic.make_descriptors(N)
        xyz_d, desc_d = torch.from_numpy(xyz).to(dev), torch.from_numpy(desc).to(dev)
    else:
        xyz = desc = None
        xyz_d = torch.empty((N, 3), dtype=torch.float32, device=dev)
        desc_d = torch.empty((8, N), dtype=torch.float32, device=dev)
    if world > 1:
        dist.broadcast(xyz_d, 0)
        dist.broadcast(desc_d, 0)
    proj = synthetic.make_proj(W, H)
    fr = FrameRenderer(xyz_d, desc_d, state, W, H, proj_matrix=proj, device=dev)
    del desc_d
    poses = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(256)]

    # double-buffered frames: the all-gather of frame i overlaps the rendering of frame i+1
    frames = [fr.rgba, torch.empty_like(fr.rgba)]
    gathered = [torch.empty((world, H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]

    def step(i):
        j = i & 1
        if pending[j] is not None:
            pending[j].wait()                      # frames[j] / gathered[j] are free again
            pending[j] = None
        fr.render_total(poses[(i * world + rank) % 256], out=frames[j])
        if world > 1:
            pending[j] = dist.all_gather_into_tensor(gathered[j], frames[j][None], async_op=True)

    for i in range(a.warmup):
        step(i)
    for p in pending:
        if p is not None:
            p.wait()
    pending[:] = [None, None]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    for p in pending:
        if p is not None:
            p.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel durations, live, with HIP events on the launch stream (rank 0)
    out = None
    if rank == 0:
        # the rasteriser warm-starts from the previous frame, so it is timed over consecutive poses of the sweep
        # (as in the timed loop), not over one repeated pose
        it = iter(range(1, 10 ** 6))
        fr.rasterize(poses[0])
        ms_splat = hip_time_ms(lambda: fr.rasterize(poses[next(it) % 256]), 16)
        ms_gather = hip_time_ms(lambda: fr.gather(), 10)
        ms_unet = hip_time_ms(lambda: fr.refine(), 5)
        f = fr.feat
        prof = None
        for _ in range(3):
            cur = fr.unet.profile(f[0][0], f[1][0], f[2][0], f[3][0], channels=4)
            prof = cur if prof is None else [(l, m0 + m1, fl, c) for (l, m0, fl, c), (_, m1, _, _) in zip(prof, cur)]
        prof = [(l, m / 3.0, fl, c) for (l, m, fl, c) in prof]
        c3_ms = sum(m for (_, m, _, c) in prof if c)
        c3_fl = sum(fl for (_, _, fl, c) in prof if c)
        n_c3 = sum(1 for (_, _, _, c) in prof if c)
        all_fl = sum(fl for (_, _, fl, _) in prof)
        achieved_tfs = c3_fl / (c3_ms * 1e-3) / 1e12
        # launches that ran the Winograd F(2x2,3x3) kernel execute 2.25x fewer MFMA flops than the algorithmic count
        wino_fl = sum(fl for (_, _, fl, c) in prof if c == 2)
        n_wino = sum(1 for (_, _, _, c) in prof if c == 2)
        executed_tfs = (c3_fl - wino_fl + wino_fl / 2.25) / (c3_ms * 1e-3) / 1e12
        all_exec = all_fl - wino_fl + wino_fl / 2.25
        splat_bytes = 12.0 * N + 8.0 * sum(w * h for (w, h) in camera.level_sizes(W, H, 5))
        gather_bytes = 68.0 * sum(w * h for (w, h) in camera.level_sizes(W, H, 5))
        out = {
            "metric": "rendered frames/sec @1216x352, 30M pts", "value": world * a.steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: synthetic {N}-point KITTI-like slab, 1216x352, 8-dim "
                                   "descriptors, 256-pose sweep, 5-scale raster + gather + 99-conv gated UNet "
                                   "(seeded random weights), RGBA out",
                       "points": N, "width": W, "height": H, "parallelism": f"pose-sharded x{world}"},
            "roofline": {
                "kernel": ("gated_conv_wino_kernel: 3x3/s1 C->C gated conv, Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32"
                           if n_wino else "gated_conv_kernel 3x3/s1 C->C (v_mfma_f32_32x32x2_f32)"), "bound": "mfma",
                "achieved": achieved_tfs, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                "frac": achieved_tfs / FP32_MFMA_PEAK_TFS, "traffic": profiled_traffic(),
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, profiles/r1_traffic.json)",
                "launches_per_frame": n_c3, "avg_launch_ms": c3_ms / max(n_c3, 1),
                "flops_per_frame": c3_fl, "winograd_launches": n_wino,
                "executed_mfma_TFLOPs": executed_tfs, "mfma_pipe_util": executed_tfs / FP32_MFMA_PEAK_TFS,
                "note": "achieved = algorithmic direct-convolution flops / time (SURVEY 8d); the Winograd launches "
                        "execute 1/2.25 of them on the MFMA pipe (mfma_pipe_util), so frac can exceed 1"},
            "stages": {
                "splat_ms": ms_splat, "splat_GBps": splat_bytes / (ms_splat * 1e-3) / 1e9,
                "splat_frac_hbm": splat_bytes / (ms_splat * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "gather_ms": ms_gather, "gather_GBps": gather_bytes / (ms_gather * 1e-3) / 1e9,
                "unet_ms": ms_unet, "unet_TFLOPs": all_fl / (ms_unet * 1e-3) / 1e12,
                "unet_frac_mfma": all_fl / (ms_unet * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFS,
                "unet_executed_TFLOPs": all_exec / (ms_unet * 1e-3) / 1e12},
        }
        if a.detail:
            os.makedirs(os.path.dirname(os.path.abspath(a.detail)), exist_ok=True)
            with open(a.detail, "w") as fh:
                json.dump([{"label": l, "ms": m, "gflop": fl / 1e9, "c3s1": c} for (l, m, fl, c) in prof], fh, indent=0)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(xyz, desc, state, proj, a.cpu_frames)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
