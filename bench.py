#!/usr/bin/env python
"""Headline benchmark: rendered frames/s at 1216x352 on a synthetic 30 M-point cloud.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
    python bench.py --config kitti6_like                 (second line: BASELINE configs[1] stand-in, see below)

A "step" is one full frame of READ's render path on one GPU: rasterise 5 scales (one pass over the
cloud) -> gather 8-channel descriptors -> 99-conv gated UNet -> RGBA frame.  Inputs (xyz, descriptors,
packed weights, cell-ordered cloud) are resident in HBM before the timed region; each step uses the next
camera pose of the novel-view sweep.  With N ranks every rank renders its own poses (weak
scaling: K frames per GPU); rank 0 builds the scene once and broadcasts it over RCCL, finished frames are
exchanged over RCCL — the only exchange the path has (§8e; read_amd/sweep.py is the one implementation of
that loop).  Rank 0 prints ONE JSON line.

Throughput mode (default --frames-in-flight 2, FrameRenderer(frames_in_flight=2)): the rasteriser + gather of the sweep run
on one stream in pose order, the UNet of consecutive poses on two streams with their own plans — the workgroups of one
frame's layer fill the CUs that the last, partly filled round of the other frame's layer leaves idle (a layer's units
rarely divide by the 512 persistent workgroups).  Every frame is still rendered completely inside the timed region;
--frames-in-flight 1 is the latency mode (one frame at a time).

After the timed loop rank 0 re-renders pose 0 through the SAME warm renderer and compares it with the CPU
oracle's frame (which the cpu_baseline leg computes anyway): raster index/depth bit-exact on all five levels,
RGB PSNR / max|diff| — "verified" in the JSON line; a mismatch exits non-zero.

--config kitti6_like: a seeded 10 M-point surface-like street scene (read_amd/synthetic.make_street_cloud; the
real kitti6 scan is a download) at the kitti6 viewport 1216x368 (downloads/kitti6.yaml:1), written to disk as a
scene directory (scene.yaml + PLY + Metashape camera.xml + reference-format checkpoints) and rendered through
load_scene_data -> setup_scene -> OGL.infer(), i.e. the reference's viewer API.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from read_amd import _lib, camera, synthetic, sweep          # noqa: E402
from read_amd.frame import FrameRenderer                      # noqa: E402
from read_amd.texture import gather_pyramid                   # noqa: E402
from read_amd.unet import pack_state, weight_spec             # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
N_POSES = 256
PSNR_FLOOR_DB = 120.0            # same guard as tests/test_gpu_unet.py (measured ~148 dB)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=256, help="timed frames per GPU (default: one full 256-pose sweep)")
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", choices=("slab30m", "kitti6_like", "train"), default="slab30m")
    p.add_argument("--points", type=int, default=0, help="override the cloud size of the config")
    p.add_argument("--frames-in-flight", type=int, default=2,
                   help="UNet plans / streams the frames of the sweep rotate through (FrameRenderer(frames_in_flight=...))")
    p.add_argument("--exchange", choices=("all", "root", "none"), default="all",
                   help="N>1: all-gather finished frames to every rank / gather to rank 0 / keep them local")
    p.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (then nothing is verified)")
    p.add_argument("--cpu-frames", type=int, default=3, help="timed CPU frames after one warm-up frame")
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
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this same command, FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950, factor re-derived
    there from a kernel of known byte count).  Launch-weighted mean over every launch of the 3x3/s1 kernels."""
    for name in ("r2_traffic.json", "r1_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            ks = json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        n = tot = 0
        for kname, v in ks.items():
            if kname.startswith("gated_conv_wino_kernel") or (kname.startswith("gated_conv") and "<3, 1, 16" in kname):
                n += v["launches"]
                tot += (v["read_bytes"] + v["write_bytes"]) * v["launches"]
        if n:
            return tot / n, name
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
class SlabWorkload:
    """BASELINE configs[2]: synthetic KITTI-like slab, FrameRenderer (three C calls per frame)."""
    name = "slab30m"
    W, H = 1216, 352

    def __init__(self, a, dev, rank):
        self.N = a.points or 30_000_000
        N = self.N
        self.state = synthetic.make_unet_state(weight_spec())
        self.xyz = self.desc = None

        def make():
            # rank 0 builds the scene; every tensor the renderer needs travels over the device collective
            from read_amd.raster import build_cells
            self.xyz = synthetic.make_cloud(N)
            self.desc = synthetic.make_descriptors(N)
            return [torch.from_numpy(self.xyz), torch.from_numpy(self.desc), torch.from_numpy(pack_state(self.state)),
                    torch.from_numpy(build_cells(self.xyz))]
        xyz_d, desc_d, packed_d, cells_d = sweep.broadcast_scene_from_rank0(make, dev)
        self.proj = synthetic.make_proj(self.W, self.H)
        self.fr = FrameRenderer(xyz_d, desc_d, packed_d, self.W, self.H, proj_matrix=self.proj, device=dev, cells=cells_d,
                                frames_in_flight=a.frames_in_flight)
        del desc_d
        self.total = [camera.total_matrix(self.proj, synthetic.sweep_pose(k)) for k in range(N_POSES)]
        self.describe = (f"BASELINE configs[2]: synthetic {N}-point KITTI-like slab, 1216x352, 8-dim descriptors, "
                         "256-pose sweep, 5-scale raster + gather + 99-conv gated UNet (seeded random weights), RGBA out")

    def render_into(self, k, out):
        self.fr.render_total(self.total[k], out=out)
        return self.fr.frame_done                  # None (one frame at a time) or the event of this frame's UNet stream

    def timed_frame(self, k):
        """Pose k through render_into() — the call the timed loop makes — preceded by its sweep predecessor so that both
        pipeline slots have been in flight; -> (index pyramid, depth pyramid, RGBA frame) of pose k."""
        bufs = [torch.empty((self.H, self.W, 4), dtype=torch.float32, device=self.fr.device) for _ in range(2)]
        self.render_into((k - 1) % N_POSES, bufs[0])
        self.render_into(k, bufs[1])
        self.fr.sync()
        torch.cuda.synchronize()
        return self.fr.idx, self.fr.depth, bufs[1]

    def rasterize(self, k):
        self.fr.sync()
        self.fr.rasterize(self.total[k])
        return self.fr.idx, self.fr.depth

    def gather(self):
        return self.fr.gather()

    def refine(self):
        return self.fr.refine()

    def profile(self):
        f = self.fr.feat
        return self.fr.unet.profile(f[0][0], f[1][0], f[2][0], f[3][0], channels=4)

    def oracle_inputs(self):
        return self.xyz, self.desc, self.state


class Kitti6LikeWorkload:
    """BASELINE configs[1] stand-in: scene directory -> load_scene_data -> setup_scene -> OGL.infer()."""
    name = "kitti6_like"
    W, H = 1216, 368

    def __init__(self, a, dev, rank):
        from read_amd import scene_io
        from read_amd.ogl import OGL
        from read_amd.pipeline import save_model
        from read_amd.render import Scene
        from read_amd.texture import PointTexture
        from read_amd.unet import UNet
        self.N = a.points or 10_000_000
        N, W, H = self.N, self.W, self.H
        fmt = "uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2, uv_1d_p1_ds3, uv_1d_p1_ds4"
        holder = [None]
        if rank == 0:
            d = tempfile.mkdtemp(prefix="read_kitti6_like_")
            xyz = synthetic.make_street_cloud(N)
            scene_io.write_ply(os.path.join(d, "pointcloud.ply"), xyz)
            cams = []
            for k in range(N_POSES):
                m = synthetic.sweep_pose(k).astype(np.float64).copy()
                m[:, 1:3] *= -1                           # Metashape convention on disk (READ/gl/utils.py:205)
                cams.append(f'<camera id="{k}" label="{k}"><transform>' + " ".join(repr(float(v)) for v in m.reshape(-1))
                            + "</transform></camera>")
            with open(os.path.join(d, "camera.xml"), "w") as fh:
                fh.write(f'<document><chunk><sensors><sensor id="0"><calibration><resolution width="{W}" height="{H}"/>'
                         f'<f>720.0</f></calibration></sensor></sensors><cameras>{"".join(cams)}</cameras></chunk></document>')
            ck = os.path.join(d, "run", "checkpoints")
            os.makedirs(ck)
            state = synthetic.make_unet_state(weight_spec())
            net = UNet()
            net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
            tex = PointTexture(8, N, init_method='zeros')
            with torch.no_grad():
                tex.texture_.copy_(torch.from_numpy(synthetic.make_descriptors(N))[None])
            args = dict(input_format=fmt, descriptor_size=8, texture_activation='none', n_points=N, supersampling=1,
                        pipeline='READ.pipelines.ogl.TexturePipeline', inference=False, lr=1e-4, texture_lr=1e-1)
            save_model(os.path.join(ck, "UNet_stage_0_epoch_1_net.pth"), net, args=args)
            save_model(os.path.join(ck, "PointTexture_stage_0_epoch_1.pth"), tex, args=args)
            with open(os.path.join(d, "scene.yaml"), "w") as fh:
                fh.write(f"viewport_size: [{W}, {H}]\nintrinsic_matrix: camera.xml\nview_matrix: camera.xml\n"
                         f"pointcloud: pointcloud.ply\nnet_path: {os.path.join(d, 'run')}\n"
                         "ckpt: UNet_stage_0_epoch_1_net.pth\ntexture_ckpt: PointTexture_stage_0_epoch_1.pth\n")
            holder[0] = d
        if dist.is_initialized():
            dist.broadcast_object_list(holder, src=0)          # one node: every rank reads the same directory
        self.dir = holder[0]
        sd = scene_io.load_scene_data(os.path.join(self.dir, "scene.yaml"))
        self.scene = Scene()
        scene_io.setup_scene(self.scene, sd)
        self.proj = camera.get_proj_matrix(sd["intrinsic_matrix"], sd["config"]["viewport_size"], 0.1, 1000.).astype(np.float32)
        self.scene.set_proj_matrix(self.proj)
        self.ogl = OGL(self.scene, sd, sd["config"]["viewport_size"], sd["net_ckpt"], sd["tex_ckpt"], out_buffer_location='torch')
        self.views = [np.asarray(v, np.float32) for v in sd["view_matrix"]]
        self.xyz = np.asarray(sd["pointcloud"]["xyz"], np.float32)
        self.texture = self.ogl.model._modules['0']
        self.net = self.ogl.model.net
        self.levels = 5
        self.idx = self.depth = self.feat = None
        if dist.is_initialized():
            dist.barrier()
        if rank == 0:
            self._cleanup = self.dir
        self.describe = (f"BASELINE configs[1] stand-in: seeded {N}-point surface-like street scene (road, facades, vehicles, "
                         "foliage), kitti6 viewport 1216x368, scene.yaml -> load_scene_data -> OGL.infer(), 256-pose sweep")

    def close(self):
        if getattr(self, "_cleanup", None):
            shutil.rmtree(self._cleanup, ignore_errors=True)

    def render_into(self, k, out):
        self.scene.set_camera_view(self.views[k])
        out.copy_(self.ogl.infer()["output"])

    def rasterize(self, k):
        self.scene.set_camera_view(self.views[k])
        self.idx, self.depth = self.scene.rasterizer().render(self.scene.total_matrix(), self.W, self.H, self.levels)
        return self.idx, self.depth

    def gather(self):
        self.feat = gather_pyramid(self.texture.rows(), self.idx, self.texture.activation)
        return self.feat

    def refine(self):
        f = self.feat
        return self.net.engine(self.H, self.W).forward(f[0][0], f[1][0], f[2][0], f[3][0], channels=4)

    def profile(self):
        f = self.feat
        return self.net.engine(self.H, self.W).profile(f[0][0], f[1][0], f[2][0], f[3][0], channels=4)

    def oracle_inputs(self):
        state = {k: v.detach().cpu().numpy() for k, v in self.net.state_dict().items()}
        return self.xyz, self.texture.texture_.detach().cpu().numpy()[0], state

    @property
    def total(self):
        class _T:
            def __getitem__(_, k):
                self.scene.set_camera_view(self.views[k])
                return self.scene.total_matrix()
        return _T()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: training iterations/s (train.py --crop_size 256x256, batch_size 2 x inner_batch 4 = 8 crops)
# ---------------------------------------------------------------------------------------------------------------------
def run_train(a, dev):
    """One iteration = what src/train.py:150-265 does per batch with the headless renderer: rasterise 8 cameras x 5 scales
    (MyRender), look the descriptors up, UNet forward, loss, backward, Adam on the net, RMSprop on the descriptors.  The
    criterion is the Huber term (x 1e4, train.py:549) only: the VGG term needs downloaded VGG weights (no network)."""
    from types import SimpleNamespace
    from read_amd.pipeline import TexturePipeline
    from read_amd.render import MyRender
    from read_amd.train import huber_loss
    N, B, S = a.points or 10_000_000, 8, 256
    xyz = synthetic.make_street_cloud(N)
    fmt = "uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2, uv_1d_p1_ds3, uv_1d_p1_ds4"

    class DS:
        id, name, input_format, tgt_sh = 0, "kitti6_like", fmt, (S, S)
        scene_data = {'pointcloud': {'xyz': xyz}}
        def load(self): pass
        def unload(self): pass

    class Crit(torch.nn.Module):
        def forward(self, out, target):
            return huber_loss(out, target)

    args = SimpleNamespace(inference=False, descriptor_size=8, texture_activation='none', use_mesh=False, supersampling=1,
                           lr=1e-4, texture_lr=1e-1, texture_ckpt=None, get_datasets=lambda _a: ([DS()], [DS()]),
                           criterion_module=Crit, criterion_args={}, pipeline='READ.pipelines.ogl.TexturePipeline')
    pipe = TexturePipeline()
    pipe.create(args)
    state = synthetic.make_unet_state(weight_spec())
    pipe.net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    with torch.no_grad():
        pipe.textures[0].texture_.copy_(torch.from_numpy(synthetic.make_descriptors(N))[None])
    model = pipe.model
    pipe.dataset_load([DS()])
    model.cuda().eval()                                           # eval_in_train: True (configs/train_example.yaml)
    extra = pipe.extra_optimizer([DS()])
    renderer = MyRender([DS()], device_outputs=True)
    rng = np.random.default_rng(2019)
    proj = synthetic.make_proj(S, S).astype(np.float32)
    targets = torch.from_numpy(rng.random((4, B, 3, S, S)).astype(np.float32)).to(dev)

    def step(i):
        views = np.stack([synthetic.sweep_pose(int(k)) for k in rng.integers(0, N_POSES, B)])
        data = {'input': {'id': torch.zeros(B, dtype=torch.long)}, 'view_matrix': torch.from_numpy(views),
                'proj_matrix': torch.from_numpy(np.repeat(proj[None], B, 0))}
        inputs, _ = renderer.render(data)
        out = model(inputs)
        loss = pipe.criterion(out, targets[i % 4]) * 1e4
        loss.backward()
        pipe.optimizer.step()
        pipe.optimizer.zero_grad()
        extra.step()
        extra.zero_grad()
        return loss

    steps, warm = (a.steps if a.steps != 256 else 10), max(a.warmup, 2)
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warm + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fwd_flops = 187.06e9 * B                                      # conv MACs x 2 at 256x256, measured on the reference module
    achieved = 3.0 * fwd_flops / (dt / steps) / 1e12               # forward + dgrad + wgrad
    out = {"metric": "training iterations/sec (8 crops of 256x256 per iteration)", "value": steps / dt, "unit": "iters/s",
           "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4] stand-in: TexturePipeline training step on a seeded {N}-point street scene, "
                                  "batch_size 2 x inner_batch 4 = 8 crops of 256x256: MyRender raster (8 cameras x 5 scales) + "
                                  "gather + UNet forward/backward (HIP autograd nodes) + Huber x 1e4 (no VGG term: weights are a "
                                  "download) + Adam(net) + sparse RMSprop(descriptors), BatchNorm in eval mode (eval_in_train)",
                      "points": N, "crop": S, "batch": B, "parallelism": "single GPU (DataParallel of the reference not rebuilt)"},
           "roofline": {"kernel": "whole step (MFMA convolutions forward + dgrad + wgrad)", "bound": "mfma",
                        "achieved": achieved, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFS,
                        "traffic": None, "flops_per_step": 3.0 * fwd_flops,
                        "note": "3 x the forward convolution FLOPs of 8 crops / wall time of a step (host-side autograd "
                                "bookkeeping included)"},
           "final_loss": float(loss.detach()), "tuning": _lib.tuning_state(), "cpu_baseline": None, "verified": None}
    if not a.no_cpu_baseline:
        from oracle import unet_torch
        import torch.nn.functional as Fnn
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        st_r = {k: torch.nn.Parameter(torch.from_numpy(np.asarray(v)).clone()) if (np.asarray(v).dtype == np.float32 and "running" not in k)
                else torch.from_numpy(np.asarray(v)) for k, v in state.items()}
        idx = [torch.from_numpy(rng.integers(0, N, (B, S >> l, S >> l))) for l in range(5)]
        tex_r = torch.nn.Parameter(torch.from_numpy(synthetic.make_descriptors(N))[None])
        t0 = time.perf_counter()
        outs = [unet_torch.unet_forward(st_r, *[tex_r[:, :, i[b]] for i in idx[:4]]) for b in range(B)]
        (Fnn.huber_loss(torch.cat(outs, 0), targets[0].cpu()) * 1e4).backward()
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "iters/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "1 iteration (8 crops): torch-CPU gather + UNet forward/backward + Huber through the oracle; "
                                         "rasterisation and optimizer steps not included"}
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# CPU leg: the oracle on this box's host cores (bounded sample) + the frame the GPU result is verified against
# ---------------------------------------------------------------------------------------------------------------------
def cpu_leg(wl, frames):
    import oracle
    from oracle import unet_torch
    xyz, desc, state = wl.oracle_inputs()
    W, H = wl.W, wl.H
    ncpu = os.cpu_count() or 1
    # thread count: torch's CPU convolutions get SLOWER with too many threads on the 256-thread GPU hosts (the UNet is
    # ~600 small ops; measured on a 128x128 frame: 0.18 s at 32 threads, 0.39 s at 64, 149 s at all 256 —
    # profiles/r2_bench.log), so probe {16, 32, 64, all if <= 128} on a small frame and run the sample with the best
    cands = sorted({min(16, ncpu), min(32, ncpu), min(64, ncpu)} | ({ncpu} if ncpu <= 128 else set()))
    xs = [torch.rand(1, 8, 128 >> l, 128 >> l) for l in range(4)]
    probe = {}
    with torch.no_grad():
        torch.set_num_threads(cands[0])
        unet_torch.unet_forward(state, *xs)
        for t in cands:
            torch.set_num_threads(t)
            t0 = time.perf_counter()
            unet_torch.unet_forward(state, *xs)
            probe[t] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    r_threads = min(ncpu, 64)
    t_r = t_g = t_u = 0.0
    first = None
    for k in range(frames + 1):                              # frame 0 = warm-up (and the verification frame)
        M = np.asarray(wl.total[k], np.float32).reshape(-1, 4, 4)[0]
        t0 = time.perf_counter()
        idx, dep = oracle.raster_multiscale(xyz, M, W, H, 5, threads=r_threads)
        t1 = time.perf_counter()
        with torch.no_grad():
            feats = [unet_torch.point_texture_forward(desc[None], i[None]) for i in idx]
            t2 = time.perf_counter()
            rgb = unet_torch.unet_forward(state, *feats[:4])
        t3 = time.perf_counter()
        if k == 0:
            first = (idx, dep, rgb[0])
        else:
            t_r, t_g, t_u = t_r + (t1 - t0), t_g + (t2 - t1), t_u + (t3 - t2)
    per = (t_r + t_g + t_u) / max(frames, 1)
    base = {"value": 1.0 / per, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 warm-up + {frames} timed full frames ({W}x{H}, {xyz.shape[0]} pts, sweep poses 1..{frames}): oracle "
                      f"raster C/OpenMP on {r_threads} threads + torch-CPU gather + torch-CPU fp32 UNet on {cores} threads "
                      f"(best of the probed thread counts) of {ncpu}",
            "thread_probe_s_128x128": {str(k): v for k, v in probe.items()},
            "ms_raster": 1e3 * t_r / max(frames, 1), "ms_gather": 1e3 * t_g / max(frames, 1),
            "ms_unet": 1e3 * t_u / max(frames, 1)}
    return base, first


def verify(wl, first):
    """Pose 0 through the warm renderer against the oracle's pose-0 frame."""
    from oracle import unet_torch
    idx_o, dep_o, rgb_o = first
    if hasattr(wl, "timed_frame"):
        # the frame comes out of the SAME call the timed loop makes (with frames in flight: two poses through the pipelined
        # path, the second one is compared), the index / depth pyramids are what that call's rasteriser left behind
        idx, depth, rgba = wl.timed_frame(0)
    else:
        idx, depth = wl.rasterize(0)
        wl.gather()
        rgba = wl.refine()
    torch.cuda.synchronize()
    exact = True
    for l in range(5):
        exact &= bool(np.array_equal(idx[l][0].cpu().numpy(), idx_o[l]))
        exact &= bool(np.array_equal(depth[l][0].cpu().numpy().view(np.uint32), dep_o[l].view(np.uint32)))
    got = rgba[:, :, :3].permute(2, 0, 1).cpu()
    diff = (got.double() - rgb_o.double())
    out = {"pose": 0, "raster_bit_exact": exact, "psnr_db": unet_torch.psnr(got, rgb_o),
           "max_abs_diff": float(diff.abs().max()),
           "rel_rms": float(diff.pow(2).mean().sqrt() / rgb_o.double().std()),
           "alpha_is_one": bool((rgba[:, :, 3] == 1).all())}
    out["ok"] = bool(exact and out["psnr_db"] >= PSNR_FLOOR_DB and out["alpha_is_one"])
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.tune:
        for kv in a.tune.split(","):
            k, v = kv.split("=")
            _lib.check(_lib.lib().read_tuning_set(k.encode(), int(v)), "read_tuning_set")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if a.config == "train":
        assert world == 1, "the training step is single-GPU (the reference's nn.DataParallel is not rebuilt)"
        run_train(a, dev)
        return
    wl = (SlabWorkload if a.config == "slab30m" else Kitti6LikeWorkload)(a, dev, rank)
    W, H, N = wl.W, wl.H, wl.N
    ex = sweep.FrameExchange((H, W, 4), dev, torch.float32, None if a.exchange == "none" else a.exchange)

    sweep.run_steps(wl.render_into, ex, 0, a.warmup, N_POSES)
    ex.drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep.run_steps(wl.render_into, ex, a.warmup, a.steps, N_POSES)
    ex.drain()
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
    rc = 0
    if rank == 0:
        # the rasteriser warm-starts from the previous frame, so it is timed over consecutive poses of the sweep
        # (as in the timed loop), not over one repeated pose
        it = iter(range(1, 10 ** 6))
        wl.rasterize(0)
        ms_splat = hip_time_ms(lambda: wl.rasterize(next(it) % N_POSES), 32)
        ms_gather = hip_time_ms(lambda: wl.gather(), 10)
        ms_unet = hip_time_ms(lambda: wl.refine(), 5)
        prof = None
        for _ in range(3):
            cur = wl.profile()
            prof = cur if prof is None else [(l, m0 + m1, fl, c) for (l, m0, fl, c), (_, m1, _, _) in zip(prof, cur)]
        prof = [(l, m / 3.0, fl, c) for (l, m, fl, c) in prof]
        c3_ms = sum(m for (_, m, _, c) in prof if c)
        c3_fl = sum(fl for (_, _, fl, c) in prof if c)
        n_c3 = sum(1 for (_, _, _, c) in prof if c)
        all_fl = sum(fl for (_, _, fl, _) in prof)
        algorithmic_tfs = c3_fl / (c3_ms * 1e-3) / 1e12
        # launches that ran the Winograd F(2x2,3x3) kernel execute 2.25x fewer MFMA flops than the algorithmic count
        wino_fl = sum(fl for (_, _, fl, c) in prof if c == 2)
        n_wino = sum(1 for (_, _, _, c) in prof if c == 2)
        c3_exec = c3_fl - wino_fl + wino_fl / 2.25
        executed_tfs = c3_exec / (c3_ms * 1e-3) / 1e12
        all_exec = all_fl - wino_fl + wino_fl / 2.25
        sizes = camera.level_sizes(W, H, 5)
        splat_bytes = 12.0 * N + 8.0 * sum(w * h for (w, h) in sizes)
        gather_bytes = 68.0 * sum(w * h for (w, h) in sizes)
        traffic, traffic_src = profiled_traffic() if a.config == "slab30m" else (None, None)
        out = {
            "metric": "rendered frames/sec @1216x352, 30M pts" if a.config == "slab30m"
                      else "rendered frames/sec @1216x368, kitti6-like 10M pts",
            "value": world * a.steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.describe, "points": N, "width": W, "height": H,
                       "parallelism": f"pose-sharded x{world}", "frame_exchange": ex.mode or "none",
                       "frames_in_flight": a.frames_in_flight},
            "roofline": {
                "kernel": ("gated_conv_wino_kernel: 3x3/s1 C->C gated conv, Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32"
                           if n_wino else "gated_conv_kernel 3x3/s1 C->C (v_mfma_f32_32x32x2_f32)"), "bound": "mfma",
                "achieved": executed_tfs, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                "frac": executed_tfs / FP32_MFMA_PEAK_TFS, "traffic": traffic,
                "traffic_unit": f"HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, profiles/{traffic_src})",
                "launches_per_frame": n_c3, "avg_launch_ms": c3_ms / max(n_c3, 1),
                "executed_flops_per_frame": c3_exec, "algorithmic_flops_per_frame": c3_fl,
                "algorithmic_TFLOPs": algorithmic_tfs, "winograd_launches": n_wino,
                "winograd_gain": c3_fl / c3_exec,
                "note": "achieved = MFMA flops the launches EXECUTE / time (a Winograd launch executes 1/2.25 of the "
                        "direct-convolution count); algorithmic_TFLOPs = SURVEY 8d's direct-convolution flops / time"},
            "stages": {
                "splat_ms": ms_splat, "splat_GBps": splat_bytes / (ms_splat * 1e-3) / 1e9,
                "splat_frac_hbm": splat_bytes / (ms_splat * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "splat_algorithmic_bytes": splat_bytes,
                "gather_ms": ms_gather, "gather_GBps": gather_bytes / (ms_gather * 1e-3) / 1e9,
                "unet_ms": ms_unet, "unet_executed_TFLOPs": all_exec / (ms_unet * 1e-3) / 1e12,
                "unet_frac_mfma": all_exec / (ms_unet * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFS,
                "unet_algorithmic_TFLOPs": all_fl / (ms_unet * 1e-3) / 1e12},
            "tuning": _lib.tuning_state(),
        }
        if a.detail:
            os.makedirs(os.path.dirname(os.path.abspath(a.detail)), exist_ok=True)
            with open(a.detail, "w") as fh:
                json.dump([{"label": l, "ms": m, "gflop": fl / 1e9, "c3s1": c} for (l, m, fl, c) in prof], fh, indent=0)
        out["cpu_baseline"] = None
        out["verified"] = None
        if not a.no_cpu_baseline:
            base, first = cpu_leg(wl, a.cpu_frames)
            if world == 1:
                out["cpu_baseline"] = base
            out["verified"] = verify(wl, first)
            if not out["verified"]["ok"]:
                rc = 3
        print(json.dumps(out), flush=True)
        if rc:
            print("bench.py: the timed configuration does NOT reproduce the oracle frame: %r" % (out["verified"],),
                  file=sys.stderr, flush=True)
    if hasattr(wl, "close"):
        wl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
