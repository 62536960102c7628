"""The fused per-frame path: camera -> point-index pyramids -> descriptor pyramids -> RGB.

This is the hot loop of the reference's viewer (viewer.py:263-285 -> READ/gl/nn.py:113-129 ->
READ/datasets/dynamic.py:66-99 -> READ/models/compose.py:125-181) with every stage resident on
one MI355X: the rasteriser's launches + 1 gather launch + the UNet plan's launches (counts: DESIGN.md §3), three C calls,
no host round trips, no per-frame allocation.
"""
import numpy as np
import torch

from . import _lib
from .camera import level_sizes, total_matrix
from .raster import PointCloudRasterizer
from .texture import gather_pyramid, texture_to_rows
from .unet import LAYOUT_FULL, UNetEngine, default_layout, layout_of, pack_state

LEVELS = 5          # the reference rasterises and gathers 5 scales; the UNet consumes 4 (unet.py:209-212)


class FrameRenderer:
    def __init__(self, xyz, texture_cn, unet_state, W, H, proj_matrix=None, device=None, levels=LEVELS, cells=True,
                 frames_in_flight=1):
        """xyz (N,3); texture_cn (C,N) descriptors (PointTexture.texture_[0]); unet_state: state dict (tensors or
        ndarrays) under the reference's names, or an already packed fp32 blob (1-D tensor, e.g. received from rank 0);
        W,H multiples of 16; cells: see PointCloudRasterizer.

        frames_in_flight > 1 (throughput mode for pose sweeps): ``render_total`` rasterises and gathers on one stream, in
        call order (the rasteriser warm-starts from the previous call), and runs the UNet of call i on stream i mod F with its
        own plan and feature buffers — the launches of consecutive frames overlap, so the workgroups of one frame fill the
        CUs that the last, partly filled round of the other frame's layer leaves idle.  The returned tensor is then
        complete on ``frame_done`` (an event; ``sync()`` waits for everything), not on the caller's stream."""
        self.device = device if device is not None else _lib.require_gpu()
        if W % 16 or H % 16:
            raise ValueError(f"set width {16 * (W // 16)} / height {16 * (H // 16)}")    # READ/gl/nn.py:107-109
        self.W, self.H, self.levels = W, H, levels
        self.raster = PointCloudRasterizer(xyz, self.device, cells=cells)
        if self.raster.n != int(torch.as_tensor(texture_cn).shape[-1]):
            raise ValueError(f"descriptor table has {int(torch.as_tensor(texture_cn).shape[-1])} columns for a cloud of "
                             f"{self.raster.n} points")
        tex = torch.as_tensor(texture_cn, dtype=torch.float32).to(self.device).contiguous()
        self.rows = texture_to_rows(tex)
        if torch.is_tensor(unet_state):
            self.packed = unet_state.to(self.device, torch.float32).contiguous()       # full or lean: read off its length
            self.unet = UNetEngine(self.packed, H, W)
        else:
            # the lean blob (451 of 952 MB: the F(4x4) layers carry their F(4x4) order only) unless the plan cannot be served by it
            self.packed = torch.from_numpy(pack_state(unet_state, layout=default_layout())).to(self.device)
            try:
                self.unet = UNetEngine(self.packed, H, W)
            except _lib.ReadHipError:
                if layout_of(self.packed) == LAYOUT_FULL:
                    raise
                self.packed = torch.from_numpy(pack_state(unet_state, layout=LAYOUT_FULL)).to(self.device)
                self.unet = UNetEngine(self.packed, H, W)
        self.proj = None if proj_matrix is None else np.asarray(proj_matrix, np.float32)
        sizes = level_sizes(W, H, levels)
        self.idx = [torch.empty((1, h, w), dtype=torch.int32, device=self.device) for (w, h) in sizes]
        self.depth = [torch.empty((1, h, w), dtype=torch.float32, device=self.device) for (w, h) in sizes]
        self.feat = [torch.empty((1, h, w, self.rows.shape[1]), dtype=torch.float32, device=self.device)
                     for (w, h) in sizes]
        self.rgba = torch.empty((H, W, 4), dtype=torch.float32, device=self.device)
        self.frame_done = None
        self._slots = []
        self._calls = 0
        self._raster_stream = None
        self.set_frames_in_flight(frames_in_flight)

    def set_frames_in_flight(self, frames_in_flight):
        """1 = one frame at a time on the caller's stream (the viewer's mode); F > 1 = F UNet plans / streams (see __init__).
        Waits for the frames in flight before switching."""
        self.sync()
        self._slots = []
        self._calls = 0
        self.frame_done = None
        if frames_in_flight > 1:
            if self._raster_stream is None:
                self._raster_stream = torch.cuda.Stream(self.device)
            for _ in range(int(frames_in_flight)):
                slot = {"feat": [torch.empty_like(f) for f in self.feat], "unet": UNetEngine(self.packed, self.H, self.W),
                        "stream": torch.cuda.Stream(self.device), "ready": torch.cuda.Event(), "done": torch.cuda.Event()}
                slot["done"].record(torch.cuda.current_stream(self.device))
                self._slots.append(slot)

    def rasterize(self, total_m, next_total=None):
        return self.raster.render(total_m, self.W, self.H, self.levels, out=(self.idx, self.depth), next_total=next_total)

    def gather(self):
        return gather_pyramid(self.rows, self.idx, out=self.feat)

    def refine(self, out=None, channels=4):
        f = self.feat
        return self.unet.forward(f[0][0], f[1][0], f[2][0], f[3][0], out=self.rgba if out is None else out,
                                 channels=channels)

    def render_total(self, total_m, out=None, channels=4, next_total=None):
        """total_m = proj @ inv(view) (4x4 fp32) -> (H,W,channels) fp32 frame on the device.
        next_total: the NEXT call's matrix when the caller knows it (PointCloudRasterizer.render): one launch less per frame."""
        if not self._slots:
            self.rasterize(total_m, next_total)
            self.gather()
            return self.refine(out, channels)
        slot = self._slots[self._calls % len(self._slots)]
        self._calls += 1
        if out is None:
            out = torch.empty((self.H, self.W, channels), dtype=torch.float32, device=self.device)
        # whatever the caller's stream has waited for (e.g. the exchange that last read `out`) the writer waits for too
        entered = torch.cuda.Event()
        entered.record(torch.cuda.current_stream(self.device))
        rs = self._raster_stream
        with torch.cuda.stream(rs):
            rs.wait_event(slot["done"])                  # this slot's features were last read by the frame F calls ago
            self.rasterize(total_m, next_total)
            gather_pyramid(self.rows, self.idx, out=slot["feat"])
            slot["ready"].record(rs)
        us = slot["stream"]
        with torch.cuda.stream(us):
            us.wait_event(slot["ready"])
            us.wait_event(entered)
            f = slot["feat"]
            slot["unet"].forward(f[0][0], f[1][0], f[2][0], f[3][0], out=out, channels=channels)
            slot["done"].record(us)
        out.record_stream(us)
        self.frame_done = slot["done"]
        return out

    def sync(self):
        """Wait (on the host) for every frame in flight."""
        if self._slots:
            self._raster_stream.synchronize()
            for slot in self._slots:
                slot["stream"].synchronize()

    def render(self, view_matrix, proj_matrix=None, out=None, channels=4):
        """view_matrix: camera->world 4x4 (the reference's convention); -> H x W x 4 RGBA (alpha = 1)."""
        proj = self.proj if proj_matrix is None else np.asarray(proj_matrix, np.float32)
        if proj is None:
            raise ValueError("no projection matrix set")
        return self.render_total(total_matrix(proj, view_matrix), out, channels)
