"""Single-layer front end of ``read_gated_conv_forward`` (READ's BasicConv, unet.py:22-53).

Used by the layer-level parity tests and the tile-configuration sweeps; the full network goes
through ``read_unet_forward`` instead (one C call per frame)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def kc_for(src_channels):
    """Input-channel chunk of the kernel: 16 when every concatenated source has C % 16 == 0, else 8."""
    return 8 if any(c % 16 for c in src_channels) else 16


class PackedGatedConv:
    """Weights of one BasicConv packed for the MFMA kernel and resident on the device."""

    def __init__(self, wf, bf, wm, bm, gamma, beta, mean, var, src_channels=None, eps=1e-5, device=None, kc=None):
        device = device if device is not None else _lib.require_gpu()
        f32 = lambda a: np.ascontiguousarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float32)
        wf, wm, bf, bm, gamma, beta, mean, var = map(f32, (wf, wm, bf, bm, gamma, beta, mean, var))
        self.cout, self.cin, self.k, _ = wf.shape
        self.kc = kc if kc is not None else kc_for(src_channels if src_channels is not None else [self.cin])
        L = _lib.lib()
        wp = np.empty(L.read_conv_packed_floats(self.cin, self.cout, self.k), np.float32)
        pp = np.empty(L.read_conv_param_floats(self.cout), np.float32)
        _lib.check(L.read_conv_pack_weights_host(self.cin, self.cout, self.k, self.kc, wf.ctypes.data, wm.ctypes.data,
                                                 wp.ctypes.data), "read_conv_pack_weights_host")
        _lib.check(L.read_conv_pack_params_host(self.cout, bf.ctypes.data, bm.ctypes.data, gamma.ctypes.data,
                                                beta.ctypes.data, mean.ctypes.data, var.ctypes.data, eps,
                                                pp.ctypes.data), "read_conv_pack_params_host")
        self.wpacked = torch.from_numpy(wp).to(device)
        self.params = torch.from_numpy(pp).to(device)
        self.wpacked_wino = self.wpacked_w16 = self.wpacked_w4 = self.wpacked_sc = self.wpacked_w4h = self.wpacked_d3h = self.wpacked_t3h = None
        if self.k == 3 and L.read_conv_t3h_floats(self.cin, self.cout):     # 8 - 32 input channels: the implicit-GEMM operand of the split-operand pixel-lane kernel
            t3 = np.empty(L.read_conv_t3h_floats(self.cin, self.cout), np.float32)
            _lib.check(L.read_conv_pack_t3h_host(self.cin, self.cout, wf.ctypes.data, wm.ctypes.data, t3.ctypes.data), "read_conv_pack_t3h_host")
            self.wpacked_t3h = torch.from_numpy(t3).to(device)
        if self.k == 3 and L.read_conv_sc_floats(self.cin, self.cout):      # small-Cout order for the vector-pipe kernel (Cout <= 4)
            sc = np.empty(L.read_conv_sc_floats(self.cin, self.cout), np.float32)
            _lib.check(L.read_conv_pack_sc_host(self.cin, self.cout, wf.ctypes.data, wm.ctypes.data, sc.ctypes.data),
                       "read_conv_pack_sc_host")
            self.wpacked_sc = torch.from_numpy(sc).to(device)
        if self.k == 3 and self.cin % 16 == 0:        # Winograd F(2x2,3x3) operand for the 3x3/s1 kernel variant
            ww = np.empty(L.read_conv_wino_floats(self.cin, self.cout), np.float32)
            _lib.check(L.read_conv_pack_wino_host(self.cin, self.cout, wf.ctypes.data, wm.ctypes.data, ww.ctypes.data),
                       "read_conv_pack_wino_host")
            self.wpacked_wino = torch.from_numpy(ww).to(device)
            w16 = np.empty(L.read_conv_wino_floats(self.cin, self.cout), np.float32)
            _lib.check(L.read_conv_pack_w16_host(self.cin, self.cout, wf.ctypes.data, wm.ctypes.data, w16.ctypes.data),
                       "read_conv_pack_w16_host")
            self.wpacked_w16 = torch.from_numpy(w16).to(device)
            if self.cout % 32 == 0 and self.cin >= 32:
                w4 = np.empty(L.read_conv_w4_floats(self.cin, self.cout), np.float32)
                _lib.check(L.read_conv_pack_w4_host(self.cin, self.cout, wf.ctypes.data, wm.ctypes.data, w4.ctypes.data),
                           "read_conv_pack_w4_host")
                self.wpacked_w4 = torch.from_numpy(w4).to(device)
                if self.cin % 32 == 0:            # ... and the same operand split into f16 piece pairs (the f16 matrix cores)
                    w4h = np.empty(L.read_conv_w4h_floats(self.cin, self.cout), np.float32)
                    _lib.check(L.read_conv_pack_w4h_host(self.cin, self.cout, wf.ctypes.data, wm.ctypes.data, w4h.ctypes.data),
                               "read_conv_pack_w4h_host")
                    self.wpacked_w4h = torch.from_numpy(w4h).to(device)
        if self.k in (1, 3, 4) and L.read_conv_dkh_floats(self.cin, self.cout, self.k):  # the plain weights as f16 piece pairs (direct split-operand kernels; 1x1: pixel-lane)
            dkh = np.empty(L.read_conv_dkh_floats(self.cin, self.cout, self.k), np.float32)
            _lib.check(L.read_conv_pack_dkh_host(self.cin, self.cout, self.k, wf.ctypes.data, wm.ctypes.data, dkh.ctypes.data),
                       "read_conv_pack_dkh_host")
            self.wpacked_d3h = torch.from_numpy(dkh).to(device)


def gated_conv(packed, sources, **kw):
    """One BasicConv launch (read_gated_conv_forward); arguments as conv_desc.  Returns the NHWC output (outH,outW,Cout)."""
    d, out = _desc(packed, sources, **kw)
    _lib.check(_lib.lib().read_gated_conv_forward(C.byref(d), _lib.stream_ptr()), "read_gated_conv_forward")
    return out


def conv_desc(packed, sources, **kw):
    """The filled read_conv_desc of a launch (for read_conv_kernel_family and friends); the tensors it points at stay alive with it."""
    d, out = _desc(packed, sources, **kw)
    d._keep = (packed, sources, kw, out)
    return d


def _desc(packed, sources, stride=1, elu=True, mul=None, residual=None, config=-1, out=None,
          out_channels=None, fill=None, linear=False, pre=None):
    """sources: list of (NHWC tensor (h,w,C), shift).

    linear: plain convolution, output channels [conv_f + b_f | conv_m + b_m] (2*Cout).
    pre: (NHWC tensor, f_off, m_off, shift[, bilinear]) pre-activation addend sampled at (y >> shift, x >> shift), or — with
    bilinear = True and shift 2 — as nn.Upsample(x4, bilinear, align_corners=False) of the tensor (include/read_hip.h)."""
    t0, s0 = sources[0]
    inH = (t0.shape[0] >> s0) if s0 >= 0 else (t0.shape[0] << -s0)
    inW = (t0.shape[1] >> s0) if s0 >= 0 else (t0.shape[1] << -s0)
    pad = (packed.k - 1) // 2
    outH = (inH + 2 * pad - packed.k) // stride + 1
    outW = (inW + 2 * pad - packed.k) // stride + 1
    cs = out_channels if out_channels is not None else packed.cout * (2 if linear else 1)
    if out is None:
        out = torch.empty((outH, outW, cs), dtype=torch.float32, device=t0.device)
    d = _lib.ConvDesc()
    d.n_src = len(sources)
    for i, (t, sh) in enumerate(sources):
        assert t.is_contiguous() and t.dtype == torch.float32
        d.src[i].data = t.data_ptr()
        d.src[i].C = t.shape[2]
        d.src[i].srcH, d.src[i].srcW = t.shape[0], t.shape[1]
        d.src[i].shift = sh
    d.mul = mul.data_ptr() if mul is not None else None
    d.inH, d.inW = inH, inW
    d.Cout, d.ksize, d.stride = packed.cout, packed.k, stride
    d.elu = 1 if elu else 0
    d.wpacked, d.params = packed.wpacked.data_ptr(), packed.params.data_ptr()
    d.residual = residual.data_ptr() if residual is not None else None
    d.out, d.out_cstride = out.data_ptr(), cs
    d.fill_pad = 0 if fill is None else 1
    d.out_fill = 0.0 if fill is None else float(fill)
    d.config = config
    d.wpacked_wino = packed.wpacked_wino.data_ptr() if packed.wpacked_wino is not None else None
    d.wpacked_w16 = packed.wpacked_w16.data_ptr() if packed.wpacked_w16 is not None else None
    d.wpacked_w4 = packed.wpacked_w4.data_ptr() if packed.wpacked_w4 is not None else None
    d.wpacked_sc = packed.wpacked_sc.data_ptr() if packed.wpacked_sc is not None else None
    d.wpacked_w4h = packed.wpacked_w4h.data_ptr() if packed.wpacked_w4h is not None else None
    d.wpacked_d3h = packed.wpacked_d3h.data_ptr() if packed.wpacked_d3h is not None else None
    d.wpacked_t3h = packed.wpacked_t3h.data_ptr() if getattr(packed, "wpacked_t3h", None) is not None else None
    d.linear = 1 if linear else 0
    if pre is not None:
        pt, f_off, m_off, psh = pre[:4]
        d.pre_bilinear = 1 if (len(pre) > 4 and pre[4]) else 0
        assert pt.is_contiguous() and pt.dtype == torch.float32
        d.pre, d.pre_cstride, d.pre_f_off, d.pre_m_off, d.pre_shift = pt.data_ptr(), pt.shape[2], f_off, m_off, psh
        d.preH, d.preW = pt.shape[0], pt.shape[1]
    return d, out


def bilinear_up4(x):
    """NHWC (h,w,C) -> (4h,4w,C), align_corners=False (unet.py:200)."""
    h, w, c = x.shape
    out = torch.empty((4 * h, 4 * w, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().read_bilinear_up4(x.data_ptr(), h, w, c, out.data_ptr(), _lib.stream_ptr()),
               "read_bilinear_up4")
    return out


def config_names():
    L = _lib.lib()
    return [L.read_conv_config_name(i).decode() for i in range(L.read_conv_config_count())]
