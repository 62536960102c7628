"""READ's refinement CNN behind the reference's ``UNet`` interface (READ/models/unet.py:121-285).

The module tree exists only to own parameters under the reference's state-dict names
(``<path>.block.{conv_f,conv_m}.{weight,bias}``, ``<path>.block.norm.*``) so that
checkpoints, ``.cuda()``, ``state_dict()`` and optimizers behave as with the reference.  It is
generated from the layer table exported by libreadhip.so — the single source of truth for the
architecture — and ``forward`` hands the whole frame to the HIP launch plan (``read_unet_forward``:
99 fused gated-conv kernels + 3 bilinear upsamples, one C call).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _alias, _lib

BN_EPS = 1e-5


def layer_table():
    """[(path, cin, cout, ksize, stride, elu)] for the 101 BasicConvs, from the C library."""
    L = _lib.lib()
    out = []
    for i in range(L.read_unet_layer_count()):
        path = C.c_char_p()
        cin, cout, k, s, elu = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(L.read_unet_layer_info(i, C.byref(path), C.byref(cin), C.byref(cout), C.byref(k), C.byref(s),
                                          C.byref(elu)))
        out.append((path.value.decode(), cin.value, cout.value, k.value, s.value, elu.value))
    return out


def weight_spec():
    """(path, cin, cout, k) per BasicConv — the input of synthetic.make_unet_state."""
    return [(p, cin, cout, k) for (p, cin, cout, k, _, _) in layer_table()]


_RAW_ORDER = ("block.conv_f.weight", "block.conv_f.bias", "block.conv_m.weight", "block.conv_m.bias",
              "block.norm.weight", "block.norm.bias", "block.norm.running_mean", "block.norm.running_var")


def raw_blob_from_state(state):
    """Flatten a state dict (tensors or ndarrays) into the raw fp32 blob ``read_unet_pack_host`` expects."""
    parts = []
    for (path, *_rest) in layer_table():
        for suffix in _RAW_ORDER:
            v = state[f"{path}.{suffix}"]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            parts.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
    blob = np.concatenate(parts)
    assert blob.size == _lib.lib().read_unet_raw_floats(), (blob.size, _lib.lib().read_unet_raw_floats())
    return blob


LAYOUT_FULL, LAYOUT_LEAN = 0, 1      # READ_UNET_LAYOUT_*: every fragment order of every layer (952 MB) / what the default plan reads (451 MB)


def default_layout():
    """LEAN unless READ_AMD_FULL_PACK=1 (tuning sessions that send F(4x4) layers to other kernels need the full blob)."""
    import os
    return LAYOUT_FULL if os.environ.get("READ_AMD_FULL_PACK") == "1" else LAYOUT_LEAN


def layout_of(packed):
    """The layout of a packed blob, read off its length."""
    L = _lib.lib()
    n = int(packed.numel() if torch.is_tensor(packed) else packed.size)
    for layout in (LAYOUT_FULL, LAYOUT_LEAN):
        if n == L.read_unet_packed_floats_layout(layout):
            return layout
    raise _lib.ReadHipError(f"a packed UNet blob has {L.read_unet_packed_floats_layout(LAYOUT_FULL)} (full) or "
                            f"{L.read_unet_packed_floats_layout(LAYOUT_LEAN)} (lean) floats, not {n}")


def pack_state(state, eps=BN_EPS, layout=LAYOUT_FULL):
    """state dict -> packed fp32 blob (host ndarray) in MFMA fragment order with folded BatchNorm."""
    raw = raw_blob_from_state(state)
    packed = np.zeros(_lib.lib().read_unet_packed_floats_layout(layout), np.float32)     # zeros: the blob has alignment gaps
    _lib.check(_lib.lib().read_unet_pack_host_layout(raw.ctypes.data, eps, packed.ctypes.data, layout), "read_unet_pack_host")
    return packed


class UNetEngine:
    """A launch plan bound to one resolution and one set of packed weights on one device."""

    def __init__(self, packed_dev, H, W):
        L = _lib.lib()
        self.H, self.W = H, W
        self.packed = packed_dev
        need = L.read_unet_workspace_bytes(H, W)
        if need == 0:
            raise _lib.ReadHipError(f"UNet viewport {W}x{H} is not a positive multiple of 16")
        self.ws = torch.empty(need, dtype=torch.uint8, device=packed_dev.device)
        h = C.c_void_p()
        _lib.check(L.read_unet_create_layout(C.byref(h), packed_dev.data_ptr(), H, W, self.ws.data_ptr(), need, layout_of(packed_dev)),
                   "read_unet_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().read_unet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, x0, x1, x2, x3, out=None, channels=3):
        """NHWC (h,w,8) fp32 pyramids -> (H,W,channels) fp32; channels=4 appends alpha=1."""
        if out is None:
            out = torch.empty((self.H, self.W, channels), dtype=torch.float32, device=x0.device)
        _lib.check(_lib.lib().read_unet_forward(self.handle, x0.data_ptr(), x1.data_ptr(), x2.data_ptr(),
                                                x3.data_ptr(), out.data_ptr(), channels, _lib.stream_ptr()),
                   "read_unet_forward")
        return out

    def launch_labels(self):
        """Labels of the plan's launches in order (layer paths; ``up4(..)`` for a separate bilinear pass)."""
        L = _lib.lib()
        return [L.read_unet_launch_label(self.handle, i).decode() for i in range(L.read_unet_launch_count(self.handle))]

    def profile(self, x0, x1, x2, x3, channels=3):
        """One instrumented frame: [(label, ms, flops, is_conv3x3_s1)] per launch (synchronises)."""
        L = _lib.lib()
        n = L.read_unet_launch_count(self.handle)
        ms = (C.c_float * n)()
        fl = (C.c_double * n)()
        c3 = (C.c_int * n)()
        out = torch.empty((self.H, self.W, channels), dtype=torch.float32, device=x0.device)
        _lib.check(L.read_unet_profile(self.handle, x0.data_ptr(), x1.data_ptr(), x2.data_ptr(), x3.data_ptr(),
                                       out.data_ptr(), channels, _lib.stream_ptr(), ms, fl, c3), "read_unet_profile")
        return [(L.read_unet_launch_label(self.handle, i).decode(), float(ms[i]), float(fl[i]), int(c3[i]))
                for i in range(n)]

    def debug_tensor(self, name):
        """Copy of an intermediate activation as an (H,W,C) tensor (tests only)."""
        H, W, Cc = C.c_int(), C.c_int(), C.c_int()
        p = _lib.lib().read_unet_debug_tensor(self.handle, name.encode(), C.byref(H), C.byref(W), C.byref(Cc))
        if not p:
            raise KeyError(name)
        off = p - self.ws.data_ptr()
        n = H.value * W.value * Cc.value
        return self.ws[off:off + 4 * n].view(torch.float32).view(H.value, W.value, Cc.value).clone()


class _GatedConvParams(nn.Module):
    """Parameter holder of one BasicConv (unet.py:22-53): block.{conv_f,conv_m,norm}."""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        pad = int((k - 1) / 2)
        self.block = nn.ModuleDict({
            'conv_f': nn.Conv2d(cin, cout, k, stride=stride, padding=pad),
            'conv_m': nn.Conv2d(cin, cout, k, stride=stride, padding=pad),
            'norm': nn.BatchNorm2d(cout),
        })


def _attach(root, path, module):
    parts = path.split('.')
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, nn.Module())
        node = node._modules[p]
    node.add_module(parts[-1], module)


class UNet(nn.Module):
    r""" Rendering network with UNet architecture and multi-scale input (drop-in for READ.models.unet.UNet).

    Args:
        num_input_channels: must be 8 (the descriptor size the reference's get_net() fixes).
        num_output_channels: must be 3.
        feature_scale, num_res: accepted for signature compatibility; the reference ignores
            feature_scale (unet.py:144-145) and READ always builds num_res=4.
    """

    def __init__(self, num_input_channels=8, num_output_channels=3, feature_scale=4, num_res=4):
        super().__init__()
        if num_input_channels != 8 or num_output_channels != 3 or num_res != 4:
            raise ValueError("the HIP UNet is built for READ's fixed configuration: 8 -> 3 channels, num_res=4")
        self.feature_scale = feature_scale
        # None: follow the tree behind the READ alias package (read_amd/_alias.py) — the reference's root tree returns the image
        # tensor (READ/models/unet.py:285), its src tree {'im_out': tensor} (src/READ/models/unet.py:280); 'tensor' / 'dict' pin it
        self.result_convention = None
        for (path, cin, cout, k, stride, _elu) in layer_table():
            _attach(self, path, _GatedConvParams(cin, cout, k, stride))
        self._engines = {}
        self._packed = None
        self._packed_key = None

    # ---- weights -> device blob --------------------------------------------------------------
    def _weights_key(self):
        # called once per frame (engine()): the flat tensor list is cached — walking the module tree for its ~700
        # parameters and buffers costs 1.4 ms of host time per call, reading their version counters 0.05 ms
        ts = self.__dict__.get('_flat_tensors')
        if ts is None:
            ts = self.__dict__['_flat_tensors'] = list(self.parameters()) + list(self.buffers())
        return (ts[0].device,) + tuple(t._version for t in ts)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_flat_tensors'] = None          # .cuda() / .to() may replace the tensors
        return super()._apply(fn, *args, **kwargs)

    def packed_weights(self):
        key = self._weights_key()
        if self._packed is None or self._packed_key != key:
            dev = key[0]
            if dev.type != 'cuda':
                raise _lib.ReadHipError("UNet.forward runs on the GPU: move the module with .cuda() "
                                        "(there is no CPU fallback)")
            self._packed = torch.from_numpy(pack_state(self.state_dict(), layout=self.__dict__.get('_layout', default_layout()))).to(dev)
            self._packed_key = key
            self._engines = {}
        return self._packed

    def invalidate(self):
        """Drop the packed weights (call after editing parameters through ``.data``, which does not bump ``_version``)."""
        self._packed = None
        self._engines = {}
        self.__dict__['_flat_tensors'] = None

    def launch_labels(self):
        """Launch labels of the most recently built plan (tests)."""
        e = list(self._engines.values())
        return e[-1].launch_labels() if e else []

    def load_state_dict(self, *args, **kwargs):
        self.invalidate()
        return super().load_state_dict(*args, **kwargs)

    def engine(self, H, W):
        packed = self.packed_weights()
        e = self._engines.get((H, W))
        if e is None:
            try:
                e = UNetEngine(packed, H, W)
            except _lib.ReadHipError:
                if layout_of(packed) != LAYOUT_LEAN:
                    raise
                # the lean blob cannot serve this plan (a tuning knob or a >= 2 GiB tensor keeps a layer off the F(4x4) kernel):
                # every fragment order, from now on
                self.__dict__['_layout'] = LAYOUT_FULL
                self._packed = None
                packed = self.packed_weights()
                e = UNetEngine(packed, H, W)
            self._engines[(H, W)] = e
        return e

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, *inputs, **kwargs):
        """inputs: x, x_2, x_4, x_8 [, x_16 ignored] as (B,8,h,w) -> (B,3,H,W)  (unet.py:202-285), wrapped as
        ``{'im_out': ...}`` under the src tree's convention (src/READ/models/unet.py:280)."""
        z = self._forward_image(*inputs, **kwargs)
        convention = self.result_convention or _alias.result_convention()
        return {'im_out': z} if convention == 'dict' else z

    def _forward_image(self, *inputs, **kwargs):
        inputs = list(inputs)
        if len(inputs) < 4:
            raise ValueError("UNet.forward needs the 1, 1/2, 1/4 and 1/8 scale inputs")
        _lib.require_gpu()
        dev = next(self.parameters()).device
        wants_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                  or any(torch.is_tensor(x) and x.requires_grad for x in inputs[:4]))
        if wants_grad or self.training:
            # .train(): batch-statistics BatchNorm (the reference's default training mode, train.py:271-279,450) lives in
            # the layer-by-layer graph, with or without gradients; the fused inference plan folds the RUNNING statistics
            # training step: every BasicConv is an autograd node backed by the HIP kernels of csrc/train.hip
            from .train import unet_forward_train_batch
            return unet_forward_train_batch(self, [x.to(dev, torch.float32) for x in inputs[:4]],
                                            per_item_statistics=bool(kwargs.get('per_item_statistics')))
        xs = [x.to(dev, torch.float32).permute(0, 2, 3, 1).contiguous() for x in inputs[:4]]
        B, H, W, _ = xs[0].shape
        eng = self.engine(H, W)
        out = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
        for b in range(B):
            eng.forward(xs[0][b], xs[1][b], xs[2][b], xs[3][b], out=out[b], channels=3)
        return out.permute(0, 3, 1, 2)
