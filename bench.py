#!/usr/bin/env python
"""Headline benchmark: rendered frames/s at 1216x352 on a synthetic 30 M-point cloud.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
    python bench.py --config kitti6_like                 (second line: BASELINE configs[1] stand-in, see below)
    python bench.py --config train                       (third line: BASELINE configs[4], training iterations/s)

The default (headline) run at N = 1 also carries compact sub-records of the other configurations in the same JSON line —
``"also": {"latency_mode", "kitti6_like", "train"}`` — each measured and VERIFIED against the oracle inside this run (their
CPU timing legs are skipped, the verification frames / iteration are kept); ``--no-also`` drops them.

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
from read_amd.unet import default_layout, pack_state, weight_spec             # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
F16_MFMA_PEAK_TFS = 2500.0       # MI355X_MICROARCH.md: BF16 / F16 dense peak (~2.5 PF; the sparse figure is never used)
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
    p.add_argument("--no-also", action="store_true", help="headline run without the kitti6_like / train / latency sub-records")
    p.add_argument("--pose-layout", choices=sweep.LAYOUTS, default=sweep.DEFAULT_LAYOUT,
                   help="how the sweep's poses are split over the ranks (read_amd/sweep.py pose_of_step)")
    p.add_argument("--pose-stride", type=int, default=0,
                   help="single-GPU proxy of ONE rank's share of a STRIDE-rank sweep: this process renders exactly the poses rank "
                        "--pose-offset of STRIDE would (no exchange partner needed); 0 = off")
    p.add_argument("--pose-offset", type=int, default=0, help="the rank emulated by --pose-stride")
    p.add_argument("--train-mode", choices=("eval", "train"), default="eval",
                   help="--config train: BatchNorm in eval mode (eval_in_train: True, configs/train_example.yaml) or with "
                        "batch statistics (model.train(), the reference's default)")
    return p.parse_args()


def hip_time_ms(fn, iters, batches=5):
    """Duration of fn() measured with HIP events on the current stream: the MEDIAN over `batches` batches of `iters` calls each —
    an event pair around back-to-back launches also times the host when the host is late (a 57 ms hiccup right after the CPU
    baseline's OpenMP threads once read as a 5.7 ms gather); the median of five batches does not."""
    torch.cuda.synchronize()
    ts = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]


def mfma_sustained_tfs(dev):
    """What the fp32 matrix path delivers on THIS box when a wave does nothing but v_mfma_f32_16x16x4_f32 (one wave per SIMD, every CU;
    read_mfma_f32_rate_probe): measured live, best of three 5 ms launches.  The guide's 157 TF is quoted at the 2.4 GHz boost clock;
    under matrix load the chip settles lower, so `frac` (against the guide's peak, as the contract asks) has a ceiling below 1."""
    import ctypes as C
    L = _lib.lib()
    scratch = torch.zeros(256 * 1024, dtype=torch.float32, device=dev)
    fl = C.c_double(0.0)
    best = 0.0
    for _ in range(4):                                          # the first launch also warms the clock up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.read_mfma_f32_rate_probe(20000, scratch.data_ptr(), C.byref(fl), _lib.stream_ptr()), "read_mfma_f32_rate_probe")
        e1.record()
        e1.synchronize()
        best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def profiled_mfma_busy():
    """MFMA-pipe busy % of the dominant kernel per level from the committed PMC pass (profiles/r6_pmc_mfma_busy.md: rocprofv3
    SQ_VALU_MFMA_BUSY_CYCLES over the four C -> C shapes) — static, like the traffic figure: counters need rocprofv3 around the process."""
    path = os.path.join(ROOT, "profiles", "r6_pmc_mfma_busy.md")
    out = {}
    try:
        for line in open(path):
            if line.startswith("| `L") and "wino4" in line:
                cells = [c.strip() for c in line.strip().strip("|").split("|")]
                out[cells[0].strip("`").split(" gated")[0]] = {"avg_us": float(cells[3]), "mfma_busy_pct": float(cells[4]), "valu_per_mfma": float(cells[7])}
    except (OSError, ValueError, IndexError):
        return None
    return out or None


def profiled_traffic():
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs of this same command, FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950, factor re-derived
    there from a kernel of known byte count).  Launch-weighted mean over every launch of the 3x3/s1 kernels."""
    for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            ks = json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        n = tot = 0
        for kname, v in ks.items():
            if kname.startswith("gated_conv_wino4_kernel") and ", 0, 0>" not in kname:
                continue                                     # the training path's linear launches (LIN = 1, 2): not this family
            if kname.startswith("gated_conv_wino4h2"):
                continue                                     # debug library only
            if kname.startswith("gated_conv_wino") or (kname.startswith("gated_conv") and "<3, 1, 16" in kname):
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
    announces_next = True            # the sweep's next pose is known: render_into(k, out, next_k)

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
            return [torch.from_numpy(self.xyz), torch.from_numpy(self.desc),
                    torch.from_numpy(pack_state(self.state, layout=default_layout() if not a.tune else 0)),   # --tune: every order
                    torch.from_numpy(build_cells(self.xyz))]
        xyz_d, desc_d, packed_d, cells_d = sweep.broadcast_scene_from_rank0(make, dev)
        self.proj = synthetic.make_proj(self.W, self.H)
        self.fr = FrameRenderer(xyz_d, desc_d, packed_d, self.W, self.H, proj_matrix=self.proj, device=dev, cells=cells_d,
                                frames_in_flight=a.frames_in_flight)
        self.xyz_d = xyz_d
        del desc_d
        self.total = [camera.total_matrix(self.proj, synthetic.sweep_pose(k)) for k in range(N_POSES)]
        self.describe = (f"BASELINE configs[2]: synthetic {N}-point KITTI-like slab, 1216x352, 8-dim descriptors, "
                         "256-pose sweep, 5-scale raster + gather + 99-conv gated UNet (seeded random weights), RGBA out")

    def render_into(self, k, out, nxt=None):
        # nxt: the sweep's next pose on this rank (sweep.run_steps(announce_next=True)): the rasteriser prepares that frame's
        # chunk lists and depth seeds inside this frame's last launch (read_splat_hint_next_camera)
        self.fr.render_total(self.total[k], out=out, next_total=None if nxt is None else self.total[nxt])
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

    def rasterize(self, k, wait=True, nxt=None):
        if wait:
            self.fr.sync()                           # three host-side stream synchronisations: not inside a timed loop (wait=False)
        self.fr.rasterize(self.total[k], None if nxt is None else self.total[nxt])
        return self.fr.idx, self.fr.depth

    def bound_raster(self):
        """call(k, next_k): pose k through the rasteriser with every foreign-call argument built once (PointCloudRasterizer.bind)."""
        return self.fr.raster.bind(self.W, self.H, self.fr.levels, (self.fr.idx, self.fr.depth), self.total)

    def gather(self):
        return self.fr.gather()

    def refine(self):
        return self.fr.refine()

    def profile(self):
        f = self.fr.feat
        return self.fr.unet.profile(f[0][0], f[1][0], f[2][0], f[3][0], channels=4)

    def oracle_inputs(self):
        if self.xyz is None:                     # ranks > 0 received the scene over the collective: copy it back to the host
            self.xyz = self.xyz_d.cpu().numpy()
            self.desc = self.fr.rows.t().contiguous().cpu().numpy()
        return self.xyz, self.desc, self.state


class Kitti6LikeWorkload:
    """BASELINE configs[1] stand-in: scene directory -> load_scene_data -> setup_scene -> OGL.infer()."""
    name = "kitti6_like"
    W, H = 1216, 368

    def __init__(self, a, dev, rank, street=None):
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
            xyz = street if street is not None and len(street) == N else synthetic.make_street_cloud(N)
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

    announces_next = True            # a pose sweep through the viewer API: Scene.announce_next_camera_view

    def render_into(self, k, out, nxt=None):
        self.scene.set_camera_view(self.views[k])
        self.scene.announce_next_camera_view(None if nxt is None else self.views[nxt])
        out.copy_(self.ogl.infer()["output"])

    def timed_frame(self, k):
        """Pose k through render_into() — the call the timed loop makes (OGL.infer(), device-resident branch)."""
        out = torch.empty((self.H, self.W, 4), dtype=torch.float32, device=self.texture.texture_.device)
        self.render_into(k, out)
        torch.cuda.synchronize()
        assert self.ogl.last_path == 'fast', "OGL.infer() left its device-resident branch"
        idx, depth = self.rasterize(k)               # the same rasteriser call infer() made, with depth this time
        return idx, depth, out

    def rasterize(self, k, wait=True, nxt=None):
        self.scene.set_camera_view(self.views[k])
        self.scene.announce_next_camera_view(None if nxt is None else self.views[nxt])
        self.idx, self.depth = self.scene.rasterizer().render(self.scene.total_matrix(), self.W, self.H, self.levels,
                                                              next_total=self.scene.take_next_total_matrix())
        return self.idx, self.depth

    def bound_raster(self):
        total = self.total
        self.idx, self.depth = self.rasterize(0)
        return self.scene.rasterizer().bind(self.W, self.H, self.levels, (self.idx, self.depth), [total[k] for k in range(N_POSES)])

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
def _grad_err(got, ref):
    """(max|diff| / max|ref|,  smallest floor factor a such that |diff| <= a * max|ref| + 1e-3 * |ref| holds everywhere)."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = max(float(ref.abs().max()), 1e-30)
    diff = (got - ref).abs()
    return float(diff.max()) / scale, float((diff - 1e-3 * ref.abs()).clamp_min(0).max()) / scale


def _host_barrier(key="read_bench_verify_done", timeout_s=1800):
    """All ranks meet on the rendezvous store (TCP, host side): no collective is in flight while a rank is busy on its CPU."""
    import datetime
    store = dist.distributed_c10d._get_default_store()
    n, r = dist.get_world_size(), dist.get_rank()
    store.set(f"{key}_{r}", "1")
    store.wait([f"{key}_{i}" for i in range(n)], datetime.timedelta(seconds=timeout_s))


def run_train(a, dev, street=None, steps=None, warm=None, cpu_timing=True, do_verify=True):
    """One iteration = what src/train.py:150-265 does per batch with the headless renderer: rasterise 8 cameras x 5 scales
    (MyRender), look the descriptors up, UNet forward, loss, backward, Adam on the net, RMSprop on the descriptors.  The
    criterion is the Huber term (x 1e4, train.py:549) only: the VGG term needs downloaded VGG weights (no network).

    verified: BEFORE the timed loop one iteration's forward/backward runs on fixed cameras and is compared with the oracle
    on the host (oracle rasteriser -> torch-CPU gather + UNet + Huber under torch.autograd): index maps bit-exact, loss,
    every parameter gradient and the descriptor gradient rows."""
    from types import SimpleNamespace
    from read_amd.pipeline import TexturePipeline
    from read_amd.render import MyRender
    from read_amd.train import huber_loss
    N, B, S = a.points or 10_000_000, 8, 256
    xyz = street if street is not None and len(street) == N else synthetic.make_street_cloud(N)
    fmt = "uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2, uv_1d_p1_ds3, uv_1d_p1_ds4"
    bn_train = a.train_mode == "train"

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
    desc0 = synthetic.make_descriptors(N)
    with torch.no_grad():
        pipe.textures[0].texture_.copy_(torch.from_numpy(desc0)[None])
    model = pipe.model
    pipe.dataset_load([DS()])
    model.cuda()
    if bn_train:
        model.train()                                             # the reference's default (train.py:271-279, eval_in_train False)
    else:
        model.eval()                                              # eval_in_train: True (configs/train_example.yaml)
    extra = pipe.extra_optimizer([DS()])
    renderer = MyRender([DS()], device_outputs=True)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    # N > 1: one process per GPU, every rank its own 8 crops; per step ONE all-reduce of the flat gradient arena and ONE
    # all-gather of the sparse descriptor pairs over RCCL (read_amd/ddp.py) — the reference's nn.DataParallel, rebuilt
    ddp_step = None
    if world > 1:
        from read_amd.ddp import DataParallelStep
        ddp_step = DataParallelStep(pipe.net, pipe.textures)
    rng = np.random.default_rng(2019 + rank)
    proj = synthetic.make_proj(S, S).astype(np.float32)
    targets = torch.from_numpy(rng.random((4, B, 3, S, S)).astype(np.float32)).to(dev)
    tex = pipe.textures[0]

    phases = {"render": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "adam": 0.0, "rmsprop": 0.0}   # HOST time, no syncs

    def forward_backward(views, target):
        t0 = time.perf_counter()
        data = {'input': {'id': torch.zeros(B, dtype=torch.long)}, 'view_matrix': torch.from_numpy(views),
                'proj_matrix': torch.from_numpy(np.repeat(proj[None], B, 0))}
        inputs, _ = renderer.render(data)
        t1 = time.perf_counter()
        out = model(inputs)
        t2 = time.perf_counter()
        loss = pipe.criterion(out, target) * 1e4
        t3 = time.perf_counter()
        loss.backward()
        t4 = time.perf_counter()
        for k_, v_ in (("render", t1 - t0), ("forward", t2 - t1), ("loss", t3 - t2), ("backward", t4 - t3)):
            phases[k_] += v_
        return loss

    def step(i):
        views = np.stack([synthetic.sweep_pose(int(k)) for k in rng.integers(0, N_POSES, B)])
        loss = forward_backward(views, targets[i % 4])
        if ddp_step is not None:
            ddp_step.reduce()
        t0 = time.perf_counter()
        pipe.optimizer.step()
        pipe.optimizer.zero_grad()
        t1 = time.perf_counter()
        extra.step()
        extra.zero_grad()
        phases["adam"] += t1 - t0
        phases["rmsprop"] += time.perf_counter() - t1
        return loss

    # ---- verification iteration (no optimizer step: the weights the timed loop starts from are the seeded ones)
    verified, base = None, None
    if do_verify and not a.no_cpu_baseline:
        import oracle
        from oracle import unet_torch
        import torch.nn.functional as Fnn
        vviews = np.stack([synthetic.sweep_pose(int(k)) for k in (0, 37, 64, 101, 128, 165, 192, 229)])
        bn_before = {k: v.detach().clone() for k, v in pipe.net.state_dict().items() if "running_" in k} if bn_train else None
        loss_g = forward_backward(vviews, targets[0])
        idx_g = [t.cpu().numpy() for t in renderer.last_index]
        grads_g = {n: p.grad.detach().cpu() for n, p in pipe.net.named_parameters() if p.grad is not None}
        drows_g = tex.grad_rows().detach().cpu()                       # dense descriptor gradient rows of this iteration (N, 8)
        bn_after = {k: v.detach().cpu() for k, v in pipe.net.state_dict().items() if "running_" in k} if bn_train else None
        pipe.optimizer.zero_grad()
        tex.take_touched()
        tex.grad_rows().zero_()
        tex.null_grad()
        if bn_train:                                                    # undo the running-statistics update of this iteration
            pipe.net.load_state_dict({**pipe.net.state_dict(), **bn_before})
        torch.cuda.synchronize()
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        r_threads = min(os.cpu_count() or 1, 64)

        def oracle_iteration(views, target, st_r, tex_r):
            tm = camera.total_matrix(proj, views)
            maps = [[], [], [], [], []]
            for b in range(B):
                il, _ = oracle.raster_multiscale(xyz, tm[b], S, S, 5, threads=r_threads)
                for l in range(5):
                    maps[l].append(il[l])
            maps = [np.stack(m) for m in maps]
            feats = [tex_r[0][:, torch.from_numpy(m.astype(np.int64))].permute(1, 0, 2, 3) for m in maps]
            # .train(): the reference runs the net once per batch item (READ/models/compose.py:137-176): per-item statistics
            out = (unet_torch.unet_forward_per_item(st_r, *feats[:4], training=True) if bn_train
                   else unet_torch.unet_forward(st_r, *feats[:4]))
            loss = Fnn.huber_loss(out, target) * 1e4
            loss.backward()
            return maps, loss

        def fresh():
            st = {k: torch.nn.Parameter(torch.from_numpy(np.asarray(v)).clone())
                  if (np.asarray(v).dtype == np.float32 and "running" not in k) else torch.from_numpy(np.asarray(v)).clone()
                  for k, v in state.items()}
            return st, torch.nn.Parameter(torch.from_numpy(desc0.copy())[None])
        st_r, tex_r = fresh()
        t0 = time.perf_counter()
        maps_o, loss_o = oracle_iteration(vviews, targets[0].cpu(), st_r, tex_r)
        t_first = time.perf_counter() - t0
        exact = all(bool(np.array_equal(idx_g[l], maps_o[l])) for l in range(5))
        worst = worst_floor = 0.0
        n_par, worst_name, over = 0, "", []
        for n, g in grads_g.items():
            e, f = _grad_err(g, st_r[n].grad)
            if e > worst:
                worst_name = n
            if e > 1e-3:
                over.append((n, round(e, 4)))
            worst, worst_floor, n_par = max(worst, e), max(worst_floor, f), n_par + 1
        touched = torch.from_numpy(np.unique(np.concatenate([m.reshape(-1) for m in maps_o])).astype(np.int64))
        dref = tex_r.grad[0].t()[touched]
        e_desc, f_desc = _grad_err(drows_g[touched], dref)
        rest = torch.ones(drows_g.shape[0], dtype=torch.bool)
        rest[touched] = False
        untouched_zero = bool(float(drows_g[rest].abs().max()) == 0.0) if bool(rest.any()) else True
        loss_rel = abs(float(loss_g) - float(loss_o)) / max(abs(float(loss_o)), 1e-30)
        verified = {"raster_bit_exact": exact, "loss": float(loss_g), "loss_oracle": float(loss_o), "loss_rel_err": loss_rel,
                    "param_grads_compared": n_par, "worst_param_grad_err_of_max": worst, "worst_param": worst_name, "params_off_by_more_than_1e-3": over[:12],
                    "worst_param_grad_floor_with_1e-3_rel": worst_floor,
                    "descriptor_rows_compared": int(touched.numel()), "descriptor_grad_err_of_max": e_desc,
                    "descriptor_grad_floor_with_1e-3_rel": f_desc, "untouched_rows_zero": untouched_zero,
                    "batchnorm": "batch statistics" if bn_train else "eval (running statistics)"}
        if bn_train:
            e_bn = max(_grad_err(bn_after[k], st_r[k])[0] for k in bn_after)
            verified["running_stats_err_of_max"] = e_bn
        g_tol = 1e-3 if bn_train else 2e-4              # batch statistics couple every pixel of a channel: round-off is amplified
        verified["grad_tolerance_of_max"] = g_tol
        verified["ok"] = bool(exact and loss_rel <= 1e-4 and n_par >= 594 and worst <= g_tol and e_desc <= g_tol and untouched_zero
                              and (not bn_train or verified["running_stats_err_of_max"] <= 1e-4))
        if cpu_timing:
            # CPU baseline: the first oracle iteration above was the warm-up; time 2 more complete iterations (oracle
            # rasteriser for 8 cameras + gather + UNet forward/backward + Huber + Adam + dense RMSprop, as the reference does)
            opt_r = torch.optim.Adam([p_ for p_ in st_r.values() if isinstance(p_, torch.nn.Parameter)], lr=1e-4)
            ext_r = torch.optim.RMSprop([tex_r], lr=1e-1)
            opt_r.zero_grad(); ext_r.zero_grad()
            n_cpu = 2
            t0 = time.perf_counter()
            for i in range(n_cpu):
                views = np.stack([synthetic.sweep_pose(int(k)) for k in rng.integers(0, N_POSES, B)])
                oracle_iteration(views, targets[i % 4].cpu(), st_r, tex_r)
                opt_r.step(); opt_r.zero_grad(); ext_r.step(); ext_r.zero_grad()
            t_cpu = (time.perf_counter() - t0) / n_cpu
            base = {"value": 1.0 / t_cpu, "unit": "iters/s", "cores": os.cpu_count() or 1, "threads_used": torch.get_num_threads(),
                    "kind": "port",
                    "sample": f"1 warm-up ({t_first:.1f} s, also the verification iteration) + {n_cpu} timed iterations of 8 crops: "
                              f"oracle rasteriser (C/OpenMP, {r_threads} threads, 8 cameras x 5 scales over {N} points) + torch-CPU "
                              "gather + UNet forward/backward + Huber through the oracle + torch Adam + dense torch RMSprop"}

    steps = steps if steps is not None else (a.steps if a.steps != 256 else 10)
    warm = warm if warm is not None else max(a.warmup, 2)
    if world > 1:
        # rank 0 alone ran the verification iteration (8 crops through the CPU oracle: tens of seconds): the other ranks wait for it
        # HERE, on the host-side store, not inside the first warm-up step's all-reduce where the RCCL watchdog would be counting
        dist.monitored_barrier(timeout=__import__("datetime").timedelta(minutes=30)) if dist.get_backend() == "gloo" else _host_barrier()
    for i in range(warm):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    for k_ in phases:
        phases[k_] = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warm + i)
    dt_host = time.perf_counter() - t0                             # the host has enqueued everything; the device may still be busy
    host_phases = {k_: 1e3 * v_ / steps for k_, v_ in phases.items()}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t_ = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        dt = float(t_.item())
    fwd_flops = 187.06e9 * B                                      # conv MACs x 2 at 256x256, measured on the reference module
    algorithmic = 3.0 * fwd_flops / (dt / steps) / 1e12            # forward + dgrad + wgrad, direct-convolution count
    # the flops the step's launches EXECUTE (the convention of the headline's roofline.frac): one instrumented step logs every
    # convolution launch with the kernel family the library takes for it (read_conv_kernel_family) — F(4x4,3x3) executes 1/4
    # of the direct count, F(2x2,3x3) 1/2.25, the Winograd-domain wgrad 1/4, direct kernels all of it (padded channels and
    # dilated dgrads included)
    from read_amd import train as _train
    step_path = _train.LAST_STEP_PATH                              # 'graph': the timed steps replayed the captured HIP graphs
    _train.FLOP_LOG, graph_was = [], _train.GRAPH_TRAIN
    _train.GRAPH_TRAIN = False                                     # the instrumented step runs the per-layer Python (same launches)
    try:
        step(warm + steps)
        torch.cuda.synchronize()
        log = list(_train.FLOP_LOG)
    finally:
        _train.FLOP_LOG, _train.GRAPH_TRAIN = None, graph_was
    gain = {4: 4.0, 2: 2.25}
    executed_flops = sum(fl / gain.get(fam, 1.0) for (_, fl, fam) in log)
    achieved = executed_flops / (dt / steps) / 1e12 if log else algorithmic
    out = {"metric": "training iterations/sec (8 crops of 256x256 per iteration and GPU)", "value": world * steps / dt, "unit": "iters/s",
           "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4] stand-in: TexturePipeline training step on a seeded {N}-point street scene, "
                                  "batch_size 2 x inner_batch 4 = 8 crops of 256x256: MyRender raster (8 cameras x 5 scales) + "
                                  "gather + UNet forward/backward (HIP autograd nodes) + Huber x 1e4 (no VGG term: weights are a "
                                  "download) + Adam(net) + sparse RMSprop(descriptors), BatchNorm "
                                  + ("with batch statistics (model.train())" if bn_train else "in eval mode (eval_in_train)"),
                      "points": N, "crop": S, "batch": B, "parallelism": "single GPU" if world == 1 else f"data-parallel x{world}: one process per GPU, 8 crops each; per step one RCCL "
                                     "all-reduce of the flat gradient arena + one all-gather of the sparse descriptor pairs (read_amd/ddp.py)",
                      "gradient_arena_bytes": None if ddp_step is None else ddp_step.arena.nbytes},
           "roofline": {"kernel": "whole step (MFMA convolutions forward + dgrad + wgrad)", "bound": "mfma",
                        "achieved": achieved, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFS,
                        "traffic": None, "flops_per_step": executed_flops if log else 3.0 * fwd_flops,
                        "algorithmic_flops_per_step": 3.0 * fwd_flops, "algorithmic_TFLOPs": algorithmic,
                        "frac_algorithmic": algorithmic / FP32_MFMA_PEAK_TFS,
                        "launches_logged": {k_: sum(1 for (kk, _, _) in log if kk == k_) for k_ in ("conv", "wgrad", "dgrad_valu")},
                        "winograd_f4_launches": sum(1 for (_, _, f_) in log if f_ == 4),
                        "winograd_f2_launches": sum(1 for (_, _, f_) in log if f_ == 2),
                        "note": "achieved = MFMA flops the step's convolution launches EXECUTE (forward pre-activations and "
                                "dgrad on the Winograd kernels at 1/4 or 1/2.25 of the direct count; the 3x3/s1 weight gradients "
                                "in the Winograd F(4x4,3x3) domain at 1/4, the others direct) / wall time of a step, the "
                                "headline's convention; frac_algorithmic = 3 x SURVEY 8d's forward count / wall time"},
           # host time inside the step's calls WITHOUT a sync: includes waiting for room in the launch queue — the Python / launch
           # path itself costs 17.5 ms per step (tools/train_host_probe.py: the same at 32 x 32 crops), the step is device-bound
           "host_enqueue_ms_per_step": 1e3 * dt_host / steps,
           "bound": "device: tools/train_host_probe.py (profiles/r4_train_host_probe.log) measures the host path of a step at 17.5 ms "
                    "whatever the crop size; host_enqueue_ms_per_step includes the time blocked on the full launch queue",
           "step_path": step_path, "host_phases_ms_per_step": host_phases,
           "final_loss": float(loss.detach()), "tuning": _lib.tuning_state(), "cpu_baseline": base, "verified": verified}
    pipe.dataset_unload([DS()])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CPU leg: the oracle on this box's host cores (bounded sample) + the frame the GPU result is verified against
# ---------------------------------------------------------------------------------------------------------------------
def cpu_leg(wl, frames, pose0=0, probe_threads=True):
    """-> (cpu_baseline record or None when frames == 0, the oracle's frame of pose `pose0`)."""
    import oracle
    from oracle import unet_torch
    xyz, desc, state = wl.oracle_inputs()
    W, H = wl.W, wl.H
    ncpu = os.cpu_count() or 1
    # thread count: torch's CPU convolutions get SLOWER with too many threads on the 256-thread GPU hosts (the UNet is
    # ~600 small ops; measured on a 128x128 frame: 0.18 s at 32 threads, 0.39 s at 64, 149 s at all 256 —
    # profiles/r2_bench.log), so the count is probed — on the REAL frame size (round 5; rounds 2-4 probed a 128x128 frame and
    # applied its winner to 1216x352, where the work per op is 26x larger and more threads can pay): one full UNet frame per
    # candidate, ascending, stopping as soon as a candidate is slower than the best so far by 1.5x
    probe = {}
    cores = min(32, ncpu)
    if probe_threads and frames > 0:
        cands = sorted({min(16, ncpu), min(32, ncpu), min(64, ncpu), min(128, ncpu)})
        xs = [torch.rand(1, 8, H >> l, W >> l) for l in range(4)]
        with torch.no_grad():
            torch.set_num_threads(cands[0])
            unet_torch.unet_forward(state, *[torch.rand(1, 8, 64 >> l, 64 >> l) for l in range(4)])   # first-touch warm-up
            for t in cands:
                torch.set_num_threads(t)
                t0 = time.perf_counter()
                unet_torch.unet_forward(state, *xs)
                probe[t] = time.perf_counter() - t0
                if probe[t] > 1.5 * min(probe.values()):
                    break
        cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    r_threads = min(ncpu, 64)
    t_r = t_g = t_u = 0.0
    first = None
    for k in range(frames + 1):                              # frame 0 = warm-up (and the verification frame)
        M = np.asarray(wl.total[(pose0 + k) % N_POSES], np.float32).reshape(-1, 4, 4)[0]
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
    if frames == 0:
        return None, first
    per = (t_r + t_g + t_u) / frames
    base = {"value": 1.0 / per, "unit": "frames/s", "cores": ncpu, "threads_used": cores, "raster_threads_used": r_threads, "kind": "port",
            "sample": f"1 warm-up + {frames} timed full frames ({W}x{H}, {xyz.shape[0]} pts, sweep poses 1..{frames}): oracle "
                      f"raster C/OpenMP on {r_threads} threads + torch-CPU gather + torch-CPU fp32 UNet on {cores} threads "
                      f"(best of the probed thread counts) of {ncpu}",
            "thread_probe_s_full_frame": {str(k): v for k, v in probe.items()},
            "ms_raster": 1e3 * t_r / frames, "ms_gather": 1e3 * t_g / frames, "ms_unet": 1e3 * t_u / frames}
    return base, first


def verify(wl, first, pose=0):
    """Pose `pose` through the warm renderer against the oracle's frame of that pose."""
    from oracle import unet_torch
    idx_o, dep_o, rgb_o = first
    # the frame comes out of the SAME call the timed loop makes (with frames in flight: two poses through the pipelined
    # path, the second one is compared), the index / depth pyramids are what that call's rasteriser left behind
    idx, depth, rgba = wl.timed_frame(pose)
    torch.cuda.synchronize()
    exact = True
    for l in range(5):
        exact &= bool(np.array_equal(idx[l][0].cpu().numpy(), idx_o[l]))
        exact &= bool(np.array_equal(depth[l][0].cpu().numpy().view(np.uint32), dep_o[l].view(np.uint32)))
    got = rgba[:, :, :3].permute(2, 0, 1).cpu()
    diff = (got.double() - rgb_o.double())
    out = {"pose": pose, "raster_bit_exact": exact, "psnr_db": unet_torch.psnr(got, rgb_o),
           "max_abs_diff": float(diff.abs().max()),
           "rel_rms": float(diff.pow(2).mean().sqrt() / rgb_o.double().std()),
           "alpha_is_one": bool((rgba[:, :, 3] == 1).all())}
    out["ok"] = bool(exact and out["psnr_db"] >= PSNR_FLOOR_DB and out["alpha_is_one"])
    return out


def timed_sweep(wl, ex, warmup, steps, world, dev, layout=None, shard=None):
    """The timed region of the contract: W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks."""
    ann = bool(getattr(wl, "announces_next", False))
    sweep.run_steps(wl.render_into, ex, 0, warmup, N_POSES, layout, shard, ann)
    ex.drain()
    if hasattr(wl, "fr"):
        wl.fr.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep.run_steps(wl.render_into, ex, warmup, steps, N_POSES, layout, shard, ann)
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
    return dt


def frame_latencies(wl, ex, n, first_step):
    """One frame at a time, each frame timed on its own from the call to its completion (host clock around render + sync):
    what a viewer sees per frame.  -> percentiles in ms."""
    ts = []
    for i in range(first_step, first_step + n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sweep.run_steps(wl.render_into, ex, i, 1, N_POSES)
        if hasattr(wl, "fr"):
            wl.fr.sync()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    q = np.percentile(np.asarray(ts), [50, 90, 99])
    return {"p50_ms": float(q[0]), "p90_ms": float(q[1]), "p99_ms": float(q[2]), "min_ms": float(min(ts)), "max_ms": float(max(ts)),
            "frames": n}


STAGE_EXTRA = {}


def stage_times(wl):
    """Per-kernel durations, live, with HIP events on the launch stream."""
    # the rasteriser warm-starts from the previous frame, so it is timed over consecutive poses of the sweep
    # (as in the timed loop), not over one repeated pose
    import ctypes as C
    ann = bool(getattr(wl, "announces_next", False))
    if hasattr(wl, "fr"):
        wl.fr.sync()
    call = wl.bound_raster()

    def lap(announce=ann):
        for k in range(N_POSES):
            call(k, (k + 1) % N_POSES if announce else None)

    def lap_ms(announce=ann):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lap(announce)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / N_POSES

    # The rasteriser stage = ONE WHOLE LAP of the 256-pose sweep, the poses in order (the rasteriser warm-starts from the previous
    # frame), each frame announcing the next pose as the timed loop does (read_splat_hint_next_camera: 4 dependent launches per
    # frame), frames queued back to back through pre-bound calls (~5 us of host time per frame: the device never waits for the
    # host).  A frame's cost depends on the pose — the camera travels 76 m into the cloud — so a few dozen poses are not the sweep
    # (rounds 1-4 timed 160 consecutive poses from wherever the counter stood).  `splat_ms` = median of three laps;
    # `splat_ms_unannounced`: one lap without the announcement (5 launches: what a viewer with a free camera gets);
    # `splat_ms_host_paced`: 64 frames through FrameRenderer.rasterize() with a host synchronisation per frame (rounds 2-4's
    # `splat_ms` loop); `splat_kernels_ms`: HIP events around every launch, mean over a lap (read_splat_profile_last; each figure
    # carries ~2 us of event overhead), `splat_kernel_sum_ms` their sum.
    lap()
    lap()
    ms_splat = sorted(lap_ms() for _ in range(3))[1]
    STAGE_EXTRA["splat_ms_queued"] = ms_splat
    if ann:
        lap(False)
        STAGE_EXTRA["splat_ms_unannounced"] = lap_ms(False)
        lap()
    it = iter(range(1, 10 ** 6))

    def frame():
        k = next(it) % N_POSES
        wl.rasterize(k, wait=True, nxt=(k + 1) % N_POSES if ann else None)
    wl.rasterize(0, nxt=1 if ann else None)
    STAGE_EXTRA["splat_ms_host_paced"] = hip_time_ms(frame, 32, batches=2)
    try:
        lap()
        _lib.check(_lib.lib().read_tuning_set(b"splat_prof", 1))
        rows = []
        buf = (C.c_float * 5)()
        for k in range(N_POSES):
            call(k, (k + 1) % N_POSES if ann else None)
            _lib.check(_lib.lib().read_splat_profile_last(buf), "read_splat_profile_last")
            rows.append(list(buf))
        mean = [float(np.mean([r[i] for r in rows])) for i in range(5)]
        STAGE_EXTRA["splat_kernels_ms"] = dict(zip(("seed_classify", "pass_a", "merge_hiz", "pass_b", "resolve_and_next"), mean))
        STAGE_EXTRA["splat_kernel_sum_ms"] = float(sum(mean))
    except _lib.ReadHipError as e:                       # e.g. a cloud below the cell path's size: no per-kernel figures
        STAGE_EXTRA["splat_kernels_ms"] = str(e)
    finally:
        _lib.check(_lib.lib().read_tuning_set(b"splat_prof", 0))
    ms_gather = hip_time_ms(lambda: wl.gather(), 10)
    ms_unet = hip_time_ms(lambda: wl.refine(), 3)
    return ms_splat, ms_gather, ms_unet


PROXY_RANK, PROXY_WORLD = 3, 8


def shard_proxy(a, dev, wl, verify_it=True):
    """One GPU renders rank PROXY_RANK's share of a PROXY_WORLD-rank sweep (same renderer, same frames in flight as the headline),
    per pose layout: frames/s, the rasteriser's time over that walk, and rank's first frame verified against the oracle."""
    rec = {"rank": PROXY_RANK, "world": PROXY_WORLD,
           "what": "single-GPU proxy: exactly the poses one rank of an 8-rank sweep renders (bench.py --pose-stride 8 --pose-offset 3); "
                   "per-GPU rate of the N = 8 run up to the frame exchange"}
    sizes = camera.level_sizes(wl.W, wl.H, 5)
    splat_bytes = 12.0 * wl.N + 8.0 * sum(w * h for (w, h) in sizes)
    n = min(a.steps, 64)
    for layout in sweep.LAYOUTS:
        ex = sweep.FrameExchange((wl.H, wl.W, 4), dev, torch.float32, None)
        dt = timed_sweep(wl, ex, a.warmup, n, 1, dev, layout, (PROXY_RANK, PROXY_WORLD))
        # the rasteriser over this rank's share of ONE sweep (32 poses), each frame announcing its next pose; the first frame of a
        # pass follows the share's last pose (a jump that a real run does not have) and is left out of the timing
        call = wl.bound_raster()
        share = [sweep.pose_of_step(i, PROXY_RANK, PROXY_WORLD, N_POSES, layout) for i in range(sweep.sweep_steps(N_POSES, PROXY_WORLD))]
        ts = []
        for rep in range(4):
            call(share[0], share[1])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for j in range(1, len(share)):
                call(share[j], share[j + 1] if j + 1 < len(share) else None)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / (len(share) - 1))
        ms = sorted(ts[1:])[1]
        r = {"value": n / dt, "unit": "frames/s", "steps": n, "splat_ms": ms,
             "splat_frac_hbm": splat_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "verified": None}
        if verify_it and not a.no_cpu_baseline:
            p0 = sweep.pose_of_step(0, PROXY_RANK, PROXY_WORLD, N_POSES, layout)
            _, first = cpu_leg(wl, 0, pose0=p0, probe_threads=False)
            r["verified"] = verify(wl, first, pose=p0)
        rec[layout] = r
    rec["interleave_over_block"] = rec["interleave"]["value"] / rec["block"]["value"]
    rec["default_layout"] = sweep.DEFAULT_LAYOUT
    return rec


def also_records(a, dev, wl, first=None):
    """Compact records of the configurations the headline line is not quoted on, measured and verified inside this run."""
    import copy
    rec = {}
    # (1) latency mode: the same renderer, one frame at a time (the viewer's mode, OGL.infer / FrameRenderer.render)
    try:
        wl.fr.set_frames_in_flight(1)
        ex = sweep.FrameExchange((wl.H, wl.W, 4), dev, torch.float32, None)
        n = min(a.steps, 64)
        dt = timed_sweep(wl, ex, a.warmup, n, 1, dev)
        lat = frame_latencies(wl, ex, max(n, 100), a.warmup + n)
        rec["latency_mode"] = {"value": n / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / n, "steps": n, "frames_in_flight": 1,
                               "frame_latency": lat,
                               "what": "headline workload, one frame at a time (every frame complete before the next starts)",
                               # the oracle's frame of pose 0 (computed for the headline) against THIS mode's frame of pose 0
                               "verified": verify(wl, first, pose=0) if first is not None else None}
    finally:
        wl.fr.set_frames_in_flight(a.frames_in_flight)
    # (1a) the UNet stage with the 3x3/s1 family back on the fp32 matrix cores (read_tuning_set("conv_w4h", 0); a second plan from the
    # full blob, the headline plan's feature pyramid as input): the A/B of the round's kernel, and the fp32-path roofline line
    rec["fp32_mfma"] = fp32_mfma_record(wl, dev)
    # (1b) multi-GPU evidence that needs no second GPU: this GPU renders exactly the share rank PROXY_RANK of an 8-rank sweep
    # would, in both pose layouts (read_amd/sweep.py) — the rasteriser warm-starts from the previous frame, so a stride-8
    # walk of the trajectory costs it coherence that a contiguous block does not
    rec["shard_proxy"] = shard_proxy(a, dev, wl)
    street = synthetic.make_street_cloud(10_000_000)
    # (2) BASELINE configs[1] stand-in through the viewer API
    ak = copy.copy(a)
    ak.points = 0
    kw = Kitti6LikeWorkload(ak, dev, 0, street=street)
    try:
        ex = sweep.FrameExchange((kw.H, kw.W, 4), dev, torch.float32, None)
        n = min(a.steps, 64)
        dt = timed_sweep(kw, ex, a.warmup, n, 1, dev)
        ms_splat, ms_gather, ms_unet = stage_times(kw)
        r = {"value": n / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / n, "steps": n, "splat_ms": ms_splat,
             "splat_ms_unannounced": STAGE_EXTRA.get("splat_ms_unannounced"), "splat_ms_host_paced": STAGE_EXTRA.get("splat_ms_host_paced"),
             "splat_kernels_ms": STAGE_EXTRA.get("splat_kernels_ms"),
             "splat_kernel_sum_ms": STAGE_EXTRA.get("splat_kernel_sum_ms"),
             "gather_ms": ms_gather, "unet_ms": ms_unet, "infer_path": kw.ogl.last_path,
             "splat_frac_hbm": (12.0 * kw.N + 8.0 * sum(w * h for (w, h) in camera.level_sizes(kw.W, kw.H, 5)))
                               / (ms_splat * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "what": kw.describe, "verified": None}
        if not a.no_cpu_baseline:
            _, first = cpu_leg(kw, 0, probe_threads=False)
            r["verified"] = verify(kw, first)
        rec["kitti6_like"] = r
    finally:
        kw.close()
        del kw
    # (3) BASELINE configs[4]: the training step
    at = copy.copy(a)
    at.points = 0
    t = run_train(at, dev, street=street, steps=10, warm=2, cpu_timing=False)
    rec["train"] = {"value": t["value"], "unit": t["unit"], "ms_per_step": t["ms_per_step"], "steps": t["steps"],
                    "frac": t["roofline"]["frac"], "frac_algorithmic": t["roofline"]["frac_algorithmic"],
                    "frac_convention": "executed MFMA flops / wall time (as roofline.frac of the headline)",
                    "host_enqueue_ms_per_step": t["host_enqueue_ms_per_step"], "step_path": t.get("step_path"),
                    "host_phases_ms_per_step": t.get("host_phases_ms_per_step"),
                    "final_loss": t["final_loss"], "verified": t["verified"],
                    "what": t["config"]["workload"]}
    return rec


def fp32_mfma_record(wl, dev):
    from read_amd.unet import LAYOUT_FULL, UNetEngine
    L = _lib.lib()
    wl.fr.sync()
    torch.cuda.synchronize()
    f = wl.fr.feat
    x = [f[i][0] for i in range(4)]
    rgba_h = wl.fr.unet.forward(*x, channels=4).clone()            # the headline kernels' frame of these features
    _lib.check(L.read_tuning_set(b"conv_w4h", 0))
    _lib.check(L.read_tuning_set(b"conv_d3h_fam", 0))
    try:
        eng = UNetEngine(torch.from_numpy(pack_state(wl.state, layout=LAYOUT_FULL)).to(dev), wl.H, wl.W)
        rgba_f = eng.forward(*x, channels=4)
        ms = hip_time_ms(lambda: eng.forward(*x, channels=4), 10)
        passes = [eng.profile(*x, channels=4) for _ in range(5)]
        prof = [(l, float(np.median([p_[i][1] for p_ in passes])), fl, c) for i, (l, _, fl, c) in enumerate(passes[0])]
        torch.cuda.synchronize()
        del eng
        # ... and round 5's whole plan: every split-operand kernel of round 6 off (3x3/s1 family, FAM, stride-2 / 4x4, 1x1, the 8-channel heads)
        for k_ in (b"conv_d3h_s2", b"conv_pxh", b"conv_t3h"):
            _lib.check(L.read_tuning_set(k_, 0))
        eng5 = UNetEngine(torch.from_numpy(pack_state(wl.state, layout=LAYOUT_FULL)).to(dev), wl.H, wl.W)
        ms_r5 = hip_time_ms(lambda: eng5.forward(*x, channels=4), 10)
        prof5 = eng5.profile(*x, channels=4)
        torch.cuda.synchronize()
        del eng5
    finally:
        for k_, v_ in ((b"conv_w4h", 32), (b"conv_d3h_fam", 32), (b"conv_d3h_s2", 32), (b"conv_pxh", 16), (b"conv_t3h", 8)):
            _lib.check(L.read_tuning_set(k_, v_))
    fam_ms = sum(m for (_, m, _, c) in prof if c)
    fam_exec = sum(fl / {2: 2.25, 4: 4.0}.get(c, 1.0) for (_, _, fl, c) in prof if c)
    d = (rgba_h[:, :, :3] - rgba_f[:, :, :3]).double()
    peak = float(rgba_f[:, :, :3].abs().max())
    mse = float((d * d).mean())
    ms_h = hip_time_ms(lambda: wl.fr.unet.forward(*x, channels=4), 10)
    return {"unet_ms": ms, "unet_ms_headline_kernels": ms_h, "unet_ms_round5_plan": ms_r5,
            "other_launches_ms_round5_plan": sum(m for (_, m, _, c) in prof5 if not c),
            "family_ms": fam_ms, "family_launches": sum(1 for p_ in prof if p_[3]),
            "winograd_f4_launches": sum(1 for p_ in prof if p_[3] == 4),
            "achieved": fam_exec / (fam_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
            "frac": fam_exec / (fam_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFS,
            "psnr_db_between_the_two_paths": (10.0 * np.log10(peak * peak / mse)) if mse > 0 else float("inf"),
            "max_abs_diff_between_the_two_paths": float(d.abs().max()),
            "what": "the UNet stage alone (one plan, one stream), 3x3/s1 family on gated_conv_wino4_kernel (v_mfma_f32_16x16x4_f32) "
                    "instead of the split-operand kernel; frac = executed fp32 MFMA flops of the family / its launch time / 157.3 TF — "
                    "rounds 3-5's roofline.frac.  unet_ms_round5_plan: the same with EVERY split-operand kernel of round 6 off (stride-2 / 4x4, "
                    "1x1 and 8-channel-head layers back on the fp32 kernels too) = round 5's launch plan on this box"}


def conv_hip_sha16():
    import hashlib
    with open(os.path.join(ROOT, "read_amd", "csrc", "conv.hip"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def profile_staleness():
    """The counter figures the line quotes from profiles/ (HBM traffic, MFMA-pipe busy) are static: taken by rocprofv3 around this
    command and committed with the sha16 of the conv.hip they were taken with (profiles/r6_profile_sources.json).  -> (stale?, note)"""
    try:
        src = json.load(open(os.path.join(ROOT, "profiles", "r6_profile_sources.json")))
    except (OSError, ValueError):
        return True, "profiles/r6_profile_sources.json is missing: the counter figures cannot be tied to this build"
    now = conv_hip_sha16()
    stale = src.get("conv_hip_sha16") != now
    return stale, {"conv_hip_sha16_profiled": src.get("conv_hip_sha16"), "conv_hip_sha16_now": now, "files": src.get("files")}


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves — the same
    argv under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 and a free port — and hand their exit
    code on.  Rank 0's JSON line reaches stdout through the inherited descriptor; the launcher's own chatter goes to stderr.
    (The reference drives all GPUs from one process, train.py:138-139: its user never types a launcher either.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %s" % (a.gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def stub_main(a, world, rank):
    """READ_BENCH_STUB=1: the launch / rendezvous / timing / one-JSON-line skeleton of main() with a stub renderer on the gloo
    backend (no GPU): what tests/test_bench_launch.py runs through the self-launch path at world size 2."""
    dist.init_process_group("gloo")
    frame = torch.full((8, 8, 4), float(rank))
    gathered = [torch.empty_like(frame) for _ in range(world)]
    for _ in range(a.warmup):
        dist.all_gather(gathered, frame)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dist.all_gather(gathered, frame)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ok = all(float(g[0, 0, 0]) == float(r) for r, g in enumerate(gathered))
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        print(json.dumps({"metric": "stub frames/sec (launcher self-test, no renderer)", "value": a.steps * world / float(dt), "unit": "frames/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": float(dt) / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub",
                          "config": {"workload": "stub"}, "verified_ranks": flags}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch N ranks with --gpus N, or give --gpus N alone: bench.py starts them)"
    if os.environ.get("READ_BENCH_STUB") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return stub_main(a, world, rank)
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
        out = run_train(a, dev, cpu_timing=(world == 1), do_verify=(rank == 0))
        bad = out["verified"] is not None and not out["verified"]["ok"]
        if rank == 0:
            print(json.dumps(out), flush=True)
            if bad:
                print("bench.py: the training iteration does NOT reproduce the oracle: %r" % (out["verified"],), file=sys.stderr, flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if bad:
            sys.exit(3)
        return
    wl = (SlabWorkload if a.config == "slab30m" else Kitti6LikeWorkload)(a, dev, rank)
    W, H, N = wl.W, wl.H, wl.N
    ex = sweep.FrameExchange((H, W, 4), dev, torch.float32, None if a.exchange == "none" else a.exchange)
    shard = (a.pose_offset, a.pose_stride) if a.pose_stride else None
    assert shard is None or (world == 1 and 0 <= a.pose_offset < a.pose_stride), "--pose-stride is a single-GPU proxy"
    dt = timed_sweep(wl, ex, a.warmup, a.steps, world, dev, a.pose_layout, shard)

    # ---- every rank checks one of ITS OWN frames (the first pose it rendered) against the oracle on its host cores
    rc = 0
    my_verified = None
    base = None
    my_pose0 = sweep.pose_of_step(0, *(shard or (rank, world)), N_POSES, a.pose_layout)
    if not a.no_cpu_baseline:
        base, first = cpu_leg(wl, a.cpu_frames if (rank == 0 and world == 1) else 0, pose0=my_pose0)
        my_verified = verify(wl, first, pose=my_pose0)
    verified_ranks = sweep.gather_objects(None if my_verified is None else bool(my_verified["ok"]))

    if rank == 0:
        # Per-kernel figures AFTER the CPU leg, as in rounds 1-3: the device has idled for the ~15 s of the CPU baseline and runs
        # at the clocks the rocprofv3 tables in profiles/ were taken at.  Right after the sustained sweep the same rasteriser loop
        # reads 94-99 us instead of 85-90 (measured in round 4: the rasteriser is latency / issue bound and follows the clock; the
        # F(4x4) figure does not move).  Medians of five batches: a late host (the CPU leg's OpenMP threads) cannot leak into them.
        ms_splat, ms_gather, ms_unet = stage_times(wl)
        # per launch: the MEDIAN of five instrumented frames (event intervals around every launch) — one late launch in one pass
        # (a host hiccup between two event records) moved the mean of three by 5 % in one of the round's runs
        passes = [wl.profile() for _ in range(5)]
        prof = [(l, float(np.median([p_[i][1] for p_ in passes])), fl, c) for i, (l, _, fl, c) in enumerate(passes[0])]
        c3_ms = sum(m for (_, m, _, c) in prof if c)
        c3_fl = sum(fl for (_, _, fl, c) in prof if c)
        n_c3 = sum(1 for (_, _, _, c) in prof if c)
        all_fl = sum(fl for (_, _, fl, _) in prof)
        algorithmic_tfs = c3_fl / (c3_ms * 1e-3) / 1e12
        # a launch that ran a Winograd kernel executes fewer MFMA flops than the algorithmic (direct-convolution) count:
        # F(2x2,3x3) 1/2.25 (c == 2), F(4x4,3x3) 1/4 (c == 4)
        # ... and the split-operand F(4x4) kernel (c == 5) forms each of those products from THREE f16 piece pairs on the f16 matrix
        # cores: it executes 3/4 of the direct count there.  `gain` = direct count / fp32-product count (what an fp32 kernel would run)
        # the direct split-operand kernel (c == 6: FAM's launches) executes every product, three piece pairs each
        gain = {0: 1.0, 1: 1.0, 2: 2.25, 4: 4.0, 5: 4.0, 6: 1.0}
        c3_exec = sum(fl / gain.get(c, 1.0) for (_, _, fl, c) in prof if c)
        n_wino = sum(1 for (_, _, _, c) in prof if c == 2)
        n_wino4 = sum(1 for (_, _, _, c) in prof if c == 4)
        n_w4h = sum(1 for (_, _, _, c) in prof if c == 5)
        executed_tfs = c3_exec / (c3_ms * 1e-3) / 1e12          # fp32-product equivalent: comparable with rounds 3-5
        all_exec = sum(fl / gain.get(c, 1.0) for (_, _, fl, c) in prof)
        n_d3h = sum(1 for (_, _, _, c) in prof if c == 6)
        w4h_ms = sum(m for (_, m, _, c) in prof if c == 5)
        w4h_f16_flops = sum(3.0 * fl / 4.0 for (_, _, fl, c) in prof if c == 5)
        w4h_tfs = w4h_f16_flops / (w4h_ms * 1e-3) / 1e12 if n_w4h else None
        sizes = camera.level_sizes(W, H, 5)
        splat_bytes = 12.0 * N + 8.0 * sum(w * h for (w, h) in sizes)
        gather_bytes = 68.0 * sum(w * h for (w, h) in sizes)
        traffic, traffic_src = profiled_traffic() if a.config == "slab30m" else (None, None)
        sustained = mfma_sustained_tfs(dev)
        out = {
            "metric": "rendered frames/sec @1216x352, 30M pts" if a.config == "slab30m"
                      else "rendered frames/sec @1216x368, kitti6-like 10M pts",
            "value": world * a.steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (matrix-core products from two f16 pieces per operand, three piece pairs, fp32 accumulation)" if n_w4h else "f32",
            "data": "synthetic",
            "config": {"workload": wl.describe, "points": N, "width": W, "height": H,
                       "parallelism": f"pose-sharded x{world}", "pose_layout": a.pose_layout,
                       "shard_proxy_of": None if shard is None else {"rank": shard[0], "world": shard[1]},
                       "frame_exchange": ex.mode or "none",
                       "frames_in_flight": a.frames_in_flight},
            "roofline": {
                "kernel": ("gated_conv_wino4h_kernel: Winograd F(4x4,3x3) with split fp32 operands on the f16 matrix cores "
                           f"({n_w4h} of the {n_c3} launches of the 3x3/s1 C->C family; {n_d3h} (FAM's x1 * x2) on the direct split-operand kernel "
                           f"gated_conv_d3h_kernel, {n_wino4} on the fp32-matrix-core F(4x4) kernel, "
                           f"{n_wino} F(2x2), {n_c3 - n_wino - n_wino4 - n_w4h - n_d3h} direct fp32)") if n_w4h else
                          ("3x3/s1 C->C gated conv family: Winograd kernels on the fp32 matrix cores "
                           f"({n_wino} launches F(2x2,3x3), {n_wino4} launches F(4x4,3x3), {n_c3 - n_wino - n_wino4} direct)"),
                "bound": "mfma",
                "achieved": w4h_tfs if n_w4h else executed_tfs, "peak": F16_MFMA_PEAK_TFS if n_w4h else FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                "frac": (w4h_tfs / F16_MFMA_PEAK_TFS) if n_w4h else executed_tfs / FP32_MFMA_PEAK_TFS,
                "frac_note": ("executed f16 MFMA flops (3 piece pairs x 1/4 of the direct count) / launch time / the f16 dense peak.  The kernel "
                              "is NOT matrix-pipe bound on this path: its MFMAs alone take 9 us of a 44 us launch (profiles/r6_w4h_ablation.md); what "
                              "binds it is the operand stream — 288 KiB of weight fragments + 43 KiB of patch per unit and 32 channels from "
                              "L2 into one CU, with 48 KiB in flight (`operand_stream`)") if n_w4h else None,
                "fp32_equivalent": {"achieved": executed_tfs, "peak": FP32_MFMA_PEAK_TFS, "frac": executed_tfs / FP32_MFMA_PEAK_TFS,
                                    "note": "the fp32 products an fp32-matrix-core kernel would execute for the same launches (direct count / 4 "
                                            "for F(4x4)) / the family's time / the fp32 MFMA peak: the figure rounds 3-5 reported as frac (0.40)"},
                "traffic": traffic, "traffic_stale": profile_staleness()[0], "profile_sources": profile_staleness()[1],
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE)",
                "traffic_source": f"static: profiles/{traffic_src} — separate --pmc passes of this command, committed; NOT measured "
                                  "in this run (counters need rocprofv3 around the process)" if traffic_src else None,
                "mfma_busy": profiled_mfma_busy(),
                "mfma_busy_source": "static: profiles/r6_pmc_mfma_busy.md — rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES per level, committed; NOT "
                                    "measured in this run; it agrees with executed flops / time per level (same file)",
                "mfma_sustained": sustained, "frac_of_sustained": executed_tfs / sustained if sustained else None,
                "mfma_sustained_note": "TFLOP/s of back-to-back v_mfma_f32_16x16x4_f32, one wave per SIMD on every CU, measured live in this run "
                                       "(read_mfma_f32_rate_probe): the matrix pipe at the clock the chip sustains; `peak` is the guide's figure at 2.4 GHz",
                "frac_algorithmic": algorithmic_tfs / FP32_MFMA_PEAK_TFS,
                "launches_per_frame": n_c3, "avg_launch_ms": c3_ms / max(n_c3, 1),
                "executed_flops_per_frame": c3_exec, "algorithmic_flops_per_frame": c3_fl,
                "algorithmic_TFLOPs": algorithmic_tfs, "winograd_f2_launches": n_wino, "winograd_f4_launches": n_wino4,
                "winograd_gain": c3_fl / c3_exec,
                "note": "achieved = MFMA flops the launches EXECUTE / time (an F(2x2,3x3) launch executes 1/2.25 of the "
                        "direct-convolution count, an F(4x4,3x3) launch 1/4); algorithmic_TFLOPs = SURVEY 8d's "
                        "direct-convolution flops / time"},
            "stages": {
                "splat_ms": ms_splat, "splat_GBps": splat_bytes / (ms_splat * 1e-3) / 1e9,
                "splat_frac_hbm": splat_bytes / (ms_splat * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "splat_algorithmic_bytes": splat_bytes, "splat_ms_queued": STAGE_EXTRA.get("splat_ms_queued"),
                "splat_ms_unannounced": STAGE_EXTRA.get("splat_ms_unannounced"), "splat_ms_host_paced": STAGE_EXTRA.get("splat_ms_host_paced"),
                "splat_kernels_ms": STAGE_EXTRA.get("splat_kernels_ms"), "splat_kernel_sum_ms": STAGE_EXTRA.get("splat_kernel_sum_ms"),
                "splat_frac_hbm_kernels": (splat_bytes / (STAGE_EXTRA["splat_kernel_sum_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                           if STAGE_EXTRA.get("splat_kernel_sum_ms") else None),
                "splat_note": "splat_ms / splat_frac_hbm (= splat_ms_queued): mean per frame over ONE WHOLE LAP of the 256-pose sweep, next pose "
                              "announced as in the timed loop, frames queued back to back (median of 3 laps); _unannounced: a lap without "
                              "the announcement (5 launches per frame); _host_paced: rounds 2-4's loop (64 frames, a host sync per frame, "
                              "poses 1..64); _kernel_sum: HIP events around every launch, lap mean",
                "gather_ms": ms_gather, "gather_GBps": gather_bytes / (ms_gather * 1e-3) / 1e9,
                "gather_frac_hbm": gather_bytes / (ms_gather * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "unet_ms": ms_unet, "unet_launches": len(prof), "unet_family_ms": c3_ms, "unet_other_ms": sum(m for (_, m, _, c) in prof if not c),
                "unet_other_launches": sum(1 for (_, _, _, c) in prof if not c),
                "unet_executed_TFLOPs": all_exec / (ms_unet * 1e-3) / 1e12,
                "unet_frac_mfma": all_exec / (ms_unet * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFS,
                "unet_algorithmic_TFLOPs": all_fl / (ms_unet * 1e-3) / 1e12},
            "tuning": _lib.tuning_state(),
        }
        if a.detail:
            os.makedirs(os.path.dirname(os.path.abspath(a.detail)), exist_ok=True)
            with open(a.detail, "w") as fh:
                json.dump([{"label": l, "ms": m, "gflop": fl / 1e9, "c3s1": c} for (l, m, fl, c) in prof], fh, indent=0)
        out["cpu_baseline"] = base if world == 1 else None
        out["verified"] = my_verified
        out["verified_ranks"] = verified_ranks
        if my_verified is not None and not all(bool(v) for v in verified_ranks):
            rc = 3
        if a.config == "slab30m" and world == 1 and not a.no_also:
            out["also"] = also_records(a, dev, wl, first if not a.no_cpu_baseline else None)
            for k, r in out["also"].items():
                v = r.get("verified")
                if v is not None and not v["ok"]:
                    rc = 3
        print(json.dumps(out), flush=True)
        if rc:
            print("bench.py: the timed configuration does NOT reproduce the oracle: %r / ranks %r / also %r"
                  % (out["verified"], verified_ranks, {k: r.get("verified") for k, r in out.get("also", {}).items()}),
                  file=sys.stderr, flush=True)
    if hasattr(wl, "close"):
        wl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
