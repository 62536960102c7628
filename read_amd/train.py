"""Native training step of READ's render path (BASELINE configs[4]).

What the reference gets from torch.autograd + cuDNN under ``src/train.py:132-203`` — backward of every ``BasicConv``
(READ/models/unet.py:22-53), of the ×4 bilinear upsample, of the descriptor lookup (READ/models/texture.py:61), the Huber
loss (src/READ/models/compose.py:35,38) and the descriptor optimizer (READ/pipelines/ogl.py:16,99-100) — runs here on the
HIP kernels of csrc/train.hip + csrc/conv.hip:

  GatedConvFn     forward  = MFMA convolution in linear mode (pre-activations f|m kept) + gate kernel
                  backward = gate backward (+ bias / BatchNorm-affine sums), dgrad (the same MFMA kernel over d[f|m] with
                             flipped, transposed weights; the six stride-2 layers as four stride-1 dgrads, one per pixel
                             parity), wgrad (3x3/s1: in the Winograd F(4x4,3x3) domain; the others direct MFMA)
  Up4Fn           bilinear x4 and its adjoint
  huber_loss      loss value + gradient in one launch
  SparseDescriptorRMSprop   RMSprop over the descriptor rows a step touched (the reference sweeps all N rows: 960 MB of
                             gradient + state at 30 M points every step)

The graph around the convolutions (torch.cat, nearest resampling, FAM's product, residual adds) is expressed with torch
tensor ops on the device so that autograd does the bookkeeping of the 99-layer graph; every FLOP-carrying node is HIP.
BatchNorm: with the net in ``.eval()`` (configs/train_example.yaml ``eval_in_train: True``; train.py:271-277) it is the
eval-mode affine map with trainable gamma/beta; with the net in ``.train()`` (the reference's default, train.py:450) every
layer normalises with the statistics of the batch and moves its running buffers (``read_bn_train_forward`` /
``read_gate_backward_bn``), exactly what nn.BatchNorm2d does in the reference's BasicConv (unet.py:40,51).
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1          # nn.BatchNorm2d default (unet.py:40)
BN_PER_ITEM = 2            # GatedConvFn's bn_train: 0 eval, 1 batch statistics over the stacked batch, 2 per stacked item


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _kc_for(cin):
    if cin % 8:
        raise ValueError(f"the HIP convolution needs input channels in multiples of 8 (got {cin})")
    return 16 if cin % 16 == 0 else 8


def _w4_fits(cin, cout):
    """Layers the Winograd F(4x4,3x3) kernel takes in linear mode (csrc/conv.hip conv_uses_w4): at least two 16-channel chunks,
    output channels in multiples of 8 (the dgrad of a 32-channel layer is a 64 -> 16 + 16 virtual layer: half a group)."""
    return USE_W4 and cin % 16 == 0 and cin >= 32 and cout % 8 == 0       # linear launches: a padded last group is masked


def _linear_conv(x, cin, wpacked, params, cout, k, stride, out, wino=None, gated=None, w4=False):
    """x (H,W,cin) NHWC -> out (Ho,Wo,2*cout): [conv_f + b_f | conv_m + b_m] through the MFMA kernel (linear epilogue);
    with Winograd fragments (3x3 / stride 1, cin % 16 == 0) through the Winograd F(2x2,3x3) kernel — which can store the
    layer's gated output in the same pass: gated = (y (Ho,Wo,cout), elu, block_h, valid_h)."""
    H, W = int(x.shape[0]), int(x.shape[1])
    d = _lib.ConvDesc()
    d.n_src = 1
    d.src[0].data = x.data_ptr()
    d.src[0].C, d.src[0].srcH, d.src[0].srcW, d.src[0].shift = cin, H, W, 0
    d.inH, d.inW, d.Cout, d.ksize, d.stride, d.elu = H, W, cout, k, stride, 0
    d.wpacked, d.params, d.out, d.out_cstride = wpacked.data_ptr(), params.data_ptr(), out.data_ptr(), 2 * cout
    d.config, d.linear = -1, 1
    if wino is not None:
        if w4:
            d.wpacked_w4 = wino.data_ptr()
        else:
            d.wpacked_wino = wino.data_ptr()
        if gated is not None:
            y, elu, bh, vh = gated
            d.out_gated, d.elu, d.block_h, d.valid_h = y.data_ptr(), int(elu), int(bh), int(vh)
    if FLOP_LOG is not None:                                 # bench.py: MFMA flops this launch EXECUTES (Winograd gains counted)
        pad = (k - 1) // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        FLOP_LOG.append(("conv", 2.0 * Ho * Wo * cin * 2 * cout * k * k, int(_lib.lib().read_conv_kernel_family(C.byref(d)))))
    _lib.check(_lib.lib().read_gated_conv_forward(C.byref(d), _lib.stream_ptr()), "read_gated_conv_forward(linear)")
    return out


# bench.py sets this to a list for one instrumented step: (kind, direct-convolution flops of the launch as issued — padded /
# dilated operands included —, kernel family 4 = F(4x4,3x3) executes 1/4 of them, 2 = F(2x2,3x3) 1/2.25, 0 = all)
FLOP_LOG = None


# packed copies of a layer's weights, reused by every image of a batch (weights only change at optimizer steps, which bump
# the parameters' _version): key = id of the conv_f weight -> (versions, params block, forward fragments, dgrad fragments)
_PACK_CACHE = {}
USE_W4 = True                # ... and through the F(4x4,3x3) kernel where its units fit (cin >= 32, cout % 32 == 0)
USE_WINOGRAD = True          # 3x3 / stride-1 layers with cin % 16 == 0: forward pre-activations and dgrad through the Winograd kernel


WGRAD_SIDE_STREAM = os.environ.get("READ_AMD_WGRAD_SIDE", "1") != "0"   # weight gradients on a side stream, joined at the end of the backward pass
_SIDE = {}                   # device -> [streams, join queued for the running backward pass, layer counter]


WGRAD_STREAMS = max(1, int(os.environ.get("READ_AMD_WGRAD_STREAMS", "1")))     # side streams the layers' weight gradients rotate over


def _side_stream(dev):
    """The side stream of the NEXT layer's weight gradient (round robin over WGRAD_STREAMS streams)."""
    e = _SIDE.get(dev)
    if e is None:
        e = _SIDE[dev] = [[torch.cuda.Stream(dev) for _ in range(WGRAD_STREAMS)], False, 0]
    e[2] += 1
    return e[0][e[2] % len(e[0])]


def _queue_join(dev):
    e = _SIDE[dev]
    if e[1]:
        return
    e[1] = True

    def join():
        e[1] = False
        cur = torch.cuda.current_stream(dev)
        for st_ in e[0]:
            cur.wait_stream(st_)
    torch.autograd.Variable._execution_engine.queue_callback(join)


_ZERO_PARAMS = {}


def _bump_version(*tensors):
    inc = getattr(torch._C, '_increment_version', None)
    if inc is not None:
        inc(list(tensors))
    else:
        for t in tensors:
            t.add_(0)


_CONST_VECS = {}


def _const_vec(n, value, dev):
    v = _CONST_VECS.get((n, value, dev))
    if v is None:
        v = _CONST_VECS[(n, value, dev)] = torch.full((n,), value, dtype=torch.float32, device=dev)
    return v


def _zero_params(n, dev):
    """An all-zero parameter block (bias / scale / shift of the dgrad's virtual layer): read-only, one per size and device."""
    z = _ZERO_PARAMS.get((n, dev))
    if z is None:
        z = _ZERO_PARAMS[(n, dev)] = torch.zeros(n, dtype=torch.float32, device=dev)
    return z


def _pack_version(wf, bf, wm, bm, gamma, beta, mean, var, identity_bn):
    if identity_bn:
        return tuple(t._version for t in (wf, bf, wm, bm)) + (wf.data_ptr(), wm.data_ptr(), 'identity')
    return tuple(t._version for t in (wf, bf, wm, bm, gamma, beta, mean, var)) + (wf.data_ptr(), wm.data_ptr())


def _packed_for(wf, bf, wm, bm, gamma, beta, mean, var, cin, cout, k, identity_bn=False, stride=1):
    """identity_bn (batch-statistics mode): the params block carries the two biases and scale 1 / shift 0 — the launch then
    stores g = act(f) * sigmoid(m) itself and read_bn_train_forward normalises it."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    ver = _pack_version(wf, bf, wm, bm, gamma, beta, mean, var, identity_bn)
    # While a HIP graph is being captured (GraphedTrainStep) every layer packs afresh: the packing launches must be PART of
    # the graph — it is replayed after every optimizer step — and a cache hit would freeze the fragments of the capture step in.
    capturing = torch.cuda.is_current_stream_capturing()
    hit = None if capturing else _PACK_CACHE.get(id(wf))
    if hit is not None and hit[6]() is not wf:              # the id was recycled by another tensor: not this layer's entry
        hit = None
    if hit is not None and hit[0] == ver:
        if hit[4] is not None:                              # None: the step's pack plan filled it on this stream, into buffers it keeps
            cur = torch.cuda.current_stream()
            cur.wait_event(hit[4])                          # packed on another item's stream
            hit[1].record_stream(cur)
            hit[2].record_stream(cur)
            if hit[5] is not None:
                hit[5].record_stream(cur)
        return hit
    dev = wf.device
    params = torch.empty(L.read_conv_param_floats(cout), dtype=torch.float32, device=dev)
    if identity_bn:
        one, zero = _const_vec(cout, 1.0, dev), _const_vec(cout, 0.0, dev)
        _lib.check(L.read_conv_pack_params_device(cout, _ptr(bf.detach()), _ptr(bm.detach()), one.data_ptr(), zero.data_ptr(),
                                                  zero.data_ptr(), one.data_ptr(), 0.0, params.data_ptr(), st))      # 1 / sqrt(1 + 0) = 1
    else:
        _lib.check(L.read_conv_pack_params_device(cout, _ptr(bf.detach()), _ptr(bm.detach()), _ptr(gamma.detach()),
                                                  _ptr(beta.detach()), _ptr(mean), _ptr(var), BN_EPS, params.data_ptr(), st))
    wf_c, wm_c = wf.detach().contiguous(), wm.detach().contiguous()
    # ONE fragment order per layer and step: the Winograd order where that kernel runs the layer (3x3 / stride 1), the direct
    # order everywhere else — packing both cost 64 us per layer and step for fragments nobody read
    wp = wino = None
    if k == 3 and stride == 1 and cin % 16 == 0 and USE_WINOGRAD:
        if _w4_fits(cin, cout):
            wino = torch.empty(L.read_conv_w4_floats(cin, cout), dtype=torch.float32, device=dev)
            _lib.check(L.read_conv_pack_w4_device(cin, cout, wf_c.data_ptr(), wm_c.data_ptr(), wino.data_ptr(), st))
        else:
            wino = torch.empty(L.read_conv_wino_floats(cin, cout), dtype=torch.float32, device=dev)
            _lib.check(L.read_conv_pack_wino_device(cin, cout, wf_c.data_ptr(), wm_c.data_ptr(), wino.data_ptr(), st))
        wp = wino                                             # read_conv_desc.wpacked must point somewhere; it is not read
    else:
        wp = torch.empty(L.read_conv_packed_floats(cin, cout, k), dtype=torch.float32, device=dev)
        _lib.check(L.read_conv_pack_weights_device(cin, cout, k, _kc_for(cin), wf_c.data_ptr(), wm_c.data_ptr(), wp.data_ptr(), st))
    ev = None
    if not capturing:             # an event recorded inside a capture must never be waited on from outside it (see backward)
        ev = torch.cuda.Event()
        ev.record()
    key = id(wf)
    old = _PACK_CACHE.get(key)
    if old is not None and old[6]() is wf:
        ref = old[6]                                          # same parameter, new version: keep its finalizer's handle
    else:
        ref = weakref.ref(wf)
        weakref.finalize(wf, _drop_pack, key, ref)            # the entry dies with its parameter
    entry = [ver, params, wp, None, ev, wino, ref]
    if not capturing:                                         # a captured entry lives in the graph's memory pool, not in the cache
        _PACK_CACHE[key] = entry
    return entry


def _drop_pack(key, ref):
    e = _PACK_CACHE.get(key)
    if e is not None and e[6] is ref:
        del _PACK_CACHE[key]


def _pack_dgrad(entry, wf, wm, cin, cout, k):
    """Flipped / transposed fragments of the layer for its dgrad (direct and, for 3x3, Winograd) on the current stream."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    dev = wf.device
    wf_c, wm_c = wf.detach().contiguous(), wm.detach().contiguous()
    wdw = None
    if k == 3 and USE_WINOGRAD:                                      # the virtual input has 2 * cp channels: always % 16
        if _w4_fits(2 * ((cout + 7) // 8 * 8), cin // 2):            # the virtual layer: 2 * cp -> cin / 2 gated channels
            wdw = torch.empty(L.read_conv_dgrad_w4_floats(cin, cout), dtype=torch.float32, device=dev)
            _lib.check(L.read_conv_pack_dgrad_w4_device(cin, cout, wf_c.data_ptr(), wm_c.data_ptr(), wdw.data_ptr(), st))
        else:
            wdw = torch.empty(L.read_conv_dgrad_wino_floats(cin, cout), dtype=torch.float32, device=dev)
            _lib.check(L.read_conv_pack_dgrad_wino_device(cin, cout, wf_c.data_ptr(), wm_c.data_ptr(), wdw.data_ptr(), st))
        wd = wdw                                                     # the stride-1 dgrad of a 3x3 layer always runs on that kernel
    else:
        wd = torch.empty(L.read_conv_dgrad_packed_floats(cin, cout, k), dtype=torch.float32, device=dev)
        _lib.check(L.read_conv_pack_dgrad_device(cin, cout, k, 16, wf_c.data_ptr(), wm_c.data_ptr(), wd.data_ptr(), st))
    ev = None
    if not torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()
        ev.record()
    entry[3] = (wd, ev, wdw)


# -----------------------------------------------------------------------------------------------------------------------
# dgrad of the 4x4 / stride-2 layers (feat_extract.7 / .3 / .4, READ/models/unet.py:198-200) on the Winograd kernel
# -----------------------------------------------------------------------------------------------------------------------
# y[o] = sum_a W[a] x[2 o + a - 1]  =>  dx[2 u + p] = sum_{t in 0..2} K_p[t] dy[u + t - 1] with K_0 = [W3, W1, 0], K_1 = [0, W2, W0]
# (per axis): each of the four pixel parities of dx is a 3x3 / stride-1 correlation over the HALF-resolution d[f|m] — i.e. the
# stride-1 dgrad of a 3x3 pseudo-layer whose weights are Wp = [0, W1, W3] (even positions) / [W0, W2, 0] (odd positions) along
# each axis.  Four launches of the F(4x4,3x3) kernel on a quarter of the pixels replace the generic vector kernel (690 us per
# layer, three layers on the critical path of the backward pass).
POLYPHASE_DGRAD = os.environ.get("READ_AMD_POLYPHASE", "1") != "0"
_POLY = {}                   # id(conv_f weight) -> (versions, fragments [4][n], w4 flag)
_POLY_SEL = {}               # device -> index tensors of the four parities


def _poly_fragments(wf, wm, cin, cout, k=4):
    """k = 3 (stride 2, pad 1) the same way: dx[2 u] = W1 dy[u], dx[2 u + 1] = W2 dy[u] + W0 dy[u + 1] — pseudo-weights
    [0, W1, 0] / [W0, W2, 0]; instead of the stride-1 dgrad over a zero-dilated d[f|m] (4x the pixels, 3/4 of them zeros)."""
    L = _lib.lib()
    key = id(wf)
    ver = (wf._version, wm._version, wf.data_ptr(), wm.data_ptr())
    capturing = torch.cuda.is_current_stream_capturing()
    hit = None if capturing else _POLY.get(key)
    if hit is not None and hit[0] == ver:
        cur = torch.cuda.current_stream(wf.device)
        if hit[3] is not None and hit[4] != cur.cuda_stream:      # packed on another stream (e.g. a graph capture's warm-up)
            cur.wait_event(hit[3])
            hit[1].record_stream(cur)
        return hit[:3]
    dev = wf.device
    sel = _POLY_SEL.get((dev, k))
    if sel is None:                                                            # parity -> source tap of the pseudo taps (k: the zero tap)
        taps = torch.tensor([[4, 1, 3], [0, 2, 4]] if k == 4 else [[3, 1, 3], [0, 2, 3]], device=dev)
        sel = _POLY_SEL[(dev, k)] = (taps[[0, 0, 1, 1]][:, :, None], taps[[0, 1, 0, 1]][:, None, :])
    w5 = torch.nn.functional.pad(torch.stack((wf.detach(), wm.detach())), (0, 1, 0, 1))      # (2, cout, cin, k + 1, k + 1)
    wp = w5[:, :, :, sel[0], sel[1]].permute(3, 0, 1, 2, 4, 5).contiguous()                    # (parity 2 py + px, f|m, cout, cin, 3, 3)
    w4 = _w4_fits(2 * ((cout + 7) // 8 * 8), cin // 2)
    n = L.read_conv_dgrad_w4_floats(cin, cout) if w4 else L.read_conv_dgrad_wino_floats(cin, cout)
    buf = torch.empty((4, n), dtype=torch.float32, device=dev)
    pack = L.read_conv_pack_dgrad_w4_device if w4 else L.read_conv_pack_dgrad_wino_device
    for par in range(4):
        _lib.check(pack(cin, cout, wp[par, 0].data_ptr(), wp[par, 1].data_ptr(), buf[par].data_ptr(), _lib.stream_ptr()))
    if not capturing:
        ev = torch.cuda.Event()
        ev.record()
        if key not in _POLY:
            weakref.finalize(wf, _POLY.pop, key, None)
        _POLY[key] = (ver, buf, w4, ev, torch.cuda.current_stream(dev).cuda_stream)
    return (ver, buf, w4)


# -----------------------------------------------------------------------------------------------------------------------
# Every packing job of a step in one launch
# -----------------------------------------------------------------------------------------------------------------------
# The optimizer changes every weight every step, so a step re-packs the parameter block, the forward fragments and the dgrad
# fragments of all 99 layers.  Layer by layer that was 297 launches of 2 .. 5 us kernels per step, each with its allocation, its
# event and its record_stream calls on the host and ~8 us of launch gap on the device (bench.py, round 4 before the wgrad work: ~1200 launches per step).  The net's tensors keep their addresses across optimizer steps, so the jobs are tabulated
# ONCE per (net, BatchNorm mode) — outputs in buffers the plan keeps — and a step is one read_conv_pack_batch launch at the head of
# the forward pass; the per-layer cache then hits for every layer.
PACK_BATCH = os.environ.get("READ_AMD_PACK_BATCH", "1") != "0"


class _NotBatchPackable(Exception):
    pass


class _PackPlan:
    """Assumptions a caller may rely on: (1) the fragments are packed on the stream of the refresh() that saw the weights change; a
    forward on ANOTHER stream waits for that launch (the event below) even when its own refresh() returns early; (2) the buffers are
    overwritten in place by the next refresh, so a backward pass uses the fragments of the weights AS THEY ARE when it runs —
    i.e. between a forward and its backward the weights must not be stepped (the usual order: backward, then optimizer.step())."""

    def __init__(self, net, identity_bn):
        from .unet import layer_table
        L = _lib.lib()
        dev = next(net.parameters()).device
        self.identity_bn = bool(identity_bn)
        self.layers = []            # (tensors of the layer, cache entry pieces)
        jobs = []

        def job(kind, mode, cin, cout, k, kc, out, wf=None, wm=None, par=None):
            j = _lib.PackJob()
            j.kind, j.mode, j.Cin, j.Cout, j.ksize, j.kc, j.eps = kind, mode, cin, cout, k, kc, BN_EPS
            j.out = out.data_ptr()
            if wf is not None:
                j.wf, j.wm = wf.data_ptr(), wm.data_ptr()
            if par is not None:
                j.bf, j.bm, j.gamma, j.beta, j.mean, j.var = (t.data_ptr() if t is not None else None for t in par)
            _lib.check(L.read_conv_pack_job_prepare(C.byref(j)), "read_conv_pack_job_prepare")
            jobs.append(j)

        f32 = dict(dtype=torch.float32, device=dev)
        for (path, cin, cout, k, stride, _elu) in layer_table():
            if path.startswith("ConvsOut."):                 # in the state dict, never executed (unet.py:181-186)
                continue
            node = net
            for p_ in path.split('.'):
                node = node._modules[p_]
            b = node.block
            n = b['norm']
            wf, bf, wm, bm = b['conv_f'].weight, b['conv_f'].bias, b['conv_m'].weight, b['conv_m'].bias
            if not (wf.is_contiguous() and wm.is_contiguous()):
                raise _NotBatchPackable("non-contiguous weights")
            # parameter block
            params = torch.empty(L.read_conv_param_floats(cout), **f32)
            if self.identity_bn:
                one, zero = _const_vec(cout, 1.0, dev), _const_vec(cout, 0.0, dev)
                par = (bf, bm, one, zero, zero, one)
            else:
                par = (bf, bm, n.weight, n.bias, n.running_mean, n.running_var)
            job(_lib.PACK_PARAMS, 0, cin, cout, k, 0, params, par=par)
            if self.identity_bn:
                jobs[-1].eps = 0.0                           # scale = 1 / sqrt(1 + 0)
            # forward fragments: ONE order per layer, the one the launch takes (as _packed_for)
            wino = None
            if k == 3 and stride == 1 and cin % 16 == 0 and USE_WINOGRAD:
                if _w4_fits(cin, cout):
                    wino = torch.empty(L.read_conv_w4_floats(cin, cout), **f32)
                    job(_lib.PACK_W4, 0, cin, cout, 3, 16, wino, wf, wm)
                else:
                    wino = torch.empty(L.read_conv_wino_floats(cin, cout), **f32)
                    job(_lib.PACK_WINO, 0, cin, cout, 3, 16, wino, wf, wm)
                wp = wino
            else:
                wp = torch.empty(L.read_conv_packed_floats(cin, cout, k), **f32)
                job(_lib.PACK_DIRECT, 0, cin, cout, k, _kc_for(cin), wp, wf, wm)
            # dgrad fragments (as _pack_dgrad) for the layers whose dgrad is a convolution launch
            dg = None
            if stride == 1 or (stride == 2 and k == 3):
                cp = (cout + 7) // 8 * 8
                if k == 3 and USE_WINOGRAD:
                    if _w4_fits(2 * cp, cin // 2):
                        wdw = torch.empty(L.read_conv_dgrad_w4_floats(cin, cout), **f32)
                        job(_lib.PACK_W4, 1, cin, cout, 3, 16, wdw, wf, wm)
                    else:
                        wdw = torch.empty(L.read_conv_dgrad_wino_floats(cin, cout), **f32)
                        job(_lib.PACK_WINO, 1, cin, cout, 3, 16, wdw, wf, wm)
                    dg = (wdw, None, wdw)
                else:
                    wd = torch.empty(L.read_conv_dgrad_packed_floats(cin, cout, k), **f32)
                    job(_lib.PACK_DIRECT, 1, cin, cout, k, 16, wd, wf, wm)
                    dg = (wd, None, None)
            self.layers.append(((wf, bf, wm, bm, n.weight, n.bias, n.running_mean, n.running_var), params, wp, dg, wino,
                                (cin, cout, k, stride)))
        first = 0
        for j in jobs:
            j.first_block = first
            first += j.nblocks
        self.total_blocks, self.njobs = first, len(jobs)
        table = (_lib.PackJob * len(jobs))(*jobs)
        self.table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
        self.signature = None
        self.stream = self.event = None

    def refresh(self):
        """Pack every layer with its current weights — one launch — unless nothing changed since the last one."""
        sig = tuple(t._version for ts in self.layers for t in ts[0])
        cur = torch.cuda.current_stream()
        if sig == self.signature:
            if self.event is not None and cur != self.stream and not torch.cuda.is_current_stream_capturing():
                cur.wait_event(self.event)                # packed on another stream: order this one behind that launch
            return
        _lib.check(_lib.lib().read_conv_pack_batch(self.table.data_ptr(), self.njobs, self.total_blocks, _lib.stream_ptr()),
                   "read_conv_pack_batch")
        self.signature = sig
        self.stream, self.event = cur, None
        if not torch.cuda.is_current_stream_capturing():
            self.event = torch.cuda.Event()
            self.event.record(cur)
        for (ts, params, wp, dg, wino, _shape) in self.layers:
            wf = ts[0]
            key = id(wf)
            old = _PACK_CACHE.get(key)
            if old is not None and old[6]() is wf:
                ref = old[6]
            else:
                ref = weakref.ref(wf)
                weakref.finalize(wf, _drop_pack, key, ref)
            _PACK_CACHE[key] = [_pack_version(*ts, self.identity_bn), params, wp, dg, None, wino, ref]


def _pack_plan(net, identity_bn):
    """The net's plan for this BatchNorm mode; rebuilt when a tensor has moved (.cuda() / .to() / a new parameter object)."""
    ts = net.__dict__.get('_flat_tensors')
    if ts is None:
        ts = net.__dict__['_flat_tensors'] = list(net.parameters()) + list(net.buffers())
    key = (bool(identity_bn), hash(tuple(t.data_ptr() for t in ts)), USE_W4, USE_WINOGRAD)
    plans = net.__dict__.setdefault('_pack_plans', {})
    if key in plans:
        return plans[key]
    if len(plans) >= 4:
        plans.clear()
    try:
        plan = _PackPlan(net, identity_bn)
    except _NotBatchPackable:
        plan = None                # e.g. channels_last weights: every layer packs itself (_packed_for calls .contiguous())
    plans[key] = plan
    return plan


class _GradArena:
    """The bias / BatchNorm-parameter gradients of every layer of ONE step live in one tensor, zero-filled by ONE launch when the
    step's first backward node asks for its slice (read_bn_param_grads accumulates): 99 `torch.zeros((4, cout))` per step otherwise."""

    def __init__(self):
        self.size, self.buf, self.taken = 0, None, set()

    def reserve(self, n):
        off = self.size
        self.size += (n + 63) // 64 * 64                    # 256-byte slices
        return off

    def take(self, off, n, dev):
        if off in self.taken:                               # a second backward over the same graph (retain_graph): fresh zeros
            return torch.zeros(n, dtype=torch.float32, device=dev)
        self.taken.add(off)
        if self.buf is None:
            self.buf = torch.zeros(self.size, dtype=torch.float32, device=dev)
        return self.buf[off:off + n]


class GatedConvFn(torch.autograd.Function):
    """y = BN_eval(act(conv_f(x) + b_f) * sigmoid(conv_m(x) + b_m)) for ONE image, x (H,W,Cin) NHWC -> (Ho,Wo,Cout)."""

    @staticmethod
    def forward(ctx, x, wf, bf, wm, bm, gamma, beta, mean, var, k, stride, elu, nb=1, v_num=1, v_den=1, bn_train=False, arena=None):
        L = _lib.lib()
        st = _lib.stream_ptr()
        x = x.contiguous()
        H, W, cin = (int(v) for v in x.shape)
        cout = int(wf.shape[0])
        pad = (k - 1) // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        dev = x.device
        wf_c, wm_c = wf.detach().contiguous(), wm.detach().contiguous()
        entry = _packed_for(wf, bf, wm, bm, gamma, beta, mean, var, cin, cout, k, identity_bn=bn_train, stride=stride)
        params, wp = entry[1], entry[2]
        ctx.pack = entry
        # the dgrad's fragments too, now: in the backward pass the wgrad kernels of the layers above fill the chip from their side
        # stream and a small packing launch between two dgrads waits 200 us for its turn (10 us here)
        # stride-2 layers: dgrad as four stride-1 dgrads, one per pixel parity (_poly_fragments) — not inside a captured step, whose
        # retained backward graph is walked again after the weights have changed (the dilated / generic paths stay there)
        poly = (POLYPHASE_DGRAD and USE_WINOGRAD and stride == 2 and k in (3, 4) and H % 2 == 0 and W % 2 == 0 and cin % 16 == 0
                and not torch.cuda.is_current_stream_capturing())
        ctx.poly = poly
        # a captured step's backward (walked eagerly over the retained graph, on the capture's stream) keeps its weight gradients on
        # that one stream: the side-stream hand-over is only exercised, and only verified, in the per-layer path
        ctx.side = WGRAD_SIDE_STREAM and not torch.cuda.is_current_stream_capturing()
        if entry[3] is None and x.requires_grad and (stride == 1 or (stride == 2 and k == 3 and H % 2 == 0 and W % 2 == 0 and not poly)):
            _pack_dgrad(entry, wf, wm, cin, cout, k)
        if poly and x.requires_grad:
            _poly_fragments(wf, wm, cin, cout, k)            # now, for the same reason
        side = _SIDE.get(dev)
        if side is not None:
            side[1] = False          # a backward pass that died before its join callback ran must not mute the next one's
        fm = torch.empty((Ho, Wo, 2 * cout), dtype=torch.float32, device=dev)
        y = torch.empty((Ho, Wo, cout), dtype=torch.float32, device=dev)
        # a batch is one tall image of nb stacked items; separator rows (block geometry at THIS layer's output scale) stay zero
        bh = Ho // nb if nb > 1 else 0
        vh = bh * v_num // v_den
        wino = entry[5]
        # the Winograd kernel writes the gated output next to the pre-activations; the other kernels leave it to the gate pass
        _linear_conv(x, cin, wp, params, cout, k, stride, fm, wino=wino, gated=(y, elu, bh, vh), w4=_w4_fits(cin, cout))
        if wino is None:
            _lib.check(L.read_gate_forward(fm.data_ptr(), Ho * Wo, cout, params.data_ptr(), int(elu), None, y.data_ptr(), Wo, bh, vh, st))
        stat, groups = None, 1
        if bn_train:
            # y holds g = act(f) * sigmoid(m) so far: normalise it with the batch statistics, in place; the module's running
            # buffers (mean, var ARE the buffers) move as nn.BatchNorm2d's do — no autograd through them.
            # bn_train == 2 (BN_PER_ITEM): every stacked item is its own BatchNorm batch (the net called once per item,
            # READ/models/compose.py:137-176) — per-item statistics, the buffers move nb times in item order
            groups = nb if (int(bn_train) == BN_PER_ITEM and nb > 1) else 1
            stat = torch.empty((groups, 2, cout), dtype=torch.float32, device=dev)
            scale_shift = torch.empty((groups, 2, (cout + 31) // 32 * 32), dtype=torch.float32, device=dev)
            scratch = torch.empty(groups * 2 * cout, dtype=torch.float64, device=dev)
            _lib.check(L.read_bn_train_forward(y.data_ptr(), Ho * Wo, cout, Wo, bh, vh, groups, gamma.detach().data_ptr(),
                                               beta.detach().data_ptr(), BN_EPS, BN_MOMENTUM, mean.data_ptr(), var.data_ptr(),
                                               stat.data_ptr(), scale_shift.data_ptr(), scratch.data_ptr(), st), "read_bn_train_forward")
            _bump_version(mean, var)                 # written through raw pointers: caches keyed on ._version must see it
            mean = var = stat                        # the backward pass needs the groups' {mean, biased var}, not the buffers
        ctx.bn_train = bool(bn_train)
        ctx.bn_groups = groups
        ctx.arena = (arena, arena.reserve(8 * cout)) if arena is not None else None      # eval mode: sums + gradients; .train(): gradients
        # the weights ride on the context, not in save_for_backward: a captured step's autograd graph is RETAINED and walked again
        # after every optimizer step (_HybridStepFn) — saved tensors are version-checked, and the optimizer's in-place update of
        # a parameter would read as "modified by an inplace operation"; a step's backward wants the weights of its own forward,
        # which are the current ones
        ctx.weights = (wf_c, wm_c, gamma.detach() if bn_train else None, mean, var)      # mean / var: running buffers or batch statistics
        ctx.save_for_backward(x, fm, params)
        ctx.cfg = (k, stride, int(elu), H, W, cin, cout, Ho, Wo, bh, vh)
        ctx.wrefs = (weakref.ref(wf), weakref.ref(wm))
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        st = _lib.stream_ptr()
        x, fm, params = ctx.saved_tensors
        wf, wm, gamma_d, mean, var = ctx.weights
        k, stride, elu, H, W, cin, cout, Ho, Wo, bh, vh = ctx.cfg
        dev = x.device
        dy = dy.contiguous()
        cp = (cout + 7) // 8 * 8
        dfm = torch.empty((Ho, Wo, 2 * cp), dtype=torch.float32, device=dev)
        groups = ctx.bn_groups
        # eval-mode BatchNorm: sums[4][cout] and the four parameter-gradient rows are one zero-filled slice [8][cout] — of the step's
        # arena (ONE fill for all layers) or of a tensor of its own; read_gate_backward / read_bn_param_grads accumulate into it
        zeroed = None
        if not ctx.bn_train:
            zeroed = (ctx.arena[0].take(ctx.arena[1], 8 * cout, dev) if ctx.arena is not None
                      else torch.zeros(8 * cout, dtype=torch.float32, device=dev)).view(8, cout)
        sums = zeroed[:4] if zeroed is not None else torch.empty((groups, 4, cout), dtype=torch.float32, device=dev)
        if ctx.bn_train:
            stat = mean                                               # [groups][2][cout]: batch mean / biased variance of the forward pass
            abc = torch.empty((groups, 3, cout), dtype=torch.float32, device=dev)
            _lib.check(L.read_gate_backward_bn(dy.data_ptr(), fm.data_ptr(), Ho * Wo, cout, params.data_ptr(), elu, dfm.data_ptr(),
                                               sums.data_ptr(), Wo, bh, vh, groups, stat.data_ptr(), gamma_d.data_ptr(), BN_EPS,
                                               abc.data_ptr(), st), "read_gate_backward_bn")
        else:
            _lib.check(L.read_gate_backward(dy.data_ptr(), fm.data_ptr(), Ho * Wo, cout, params.data_ptr(), elu, dfm.data_ptr(),
                                            sums.data_ptr(), Wo, bh, vh, st))
        # The side stream's outputs are allocated BEFORE the event it waits for: a block the caching allocator hands out here was
        # freed by main-stream work that precedes the event.  Allocated after the dgrad below, dwf could be the block of a dgrad
        # temporary whose kernels are still queued on the main stream — the wgrad, ordered only behind ev_dfm, then wrote into memory
        # the main stream was still using (seen as a garbage dW_f of feat_extract.7 once the dgrad had freed tensors of that size).
        want_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        dwf = dwm = scratch = None
        if want_w:
            dwf, dwm = torch.empty_like(wf), torch.empty_like(wm)
            n_scr = L.read_conv_wgrad_scratch_floats(cin, cout, k, Ho)
            scratch = torch.empty(n_scr, dtype=torch.float32, device=dev)
        ev_dfm = None
        if ctx.side:
            ev_dfm = torch.cuda.Event()
            ev_dfm.record(torch.cuda.current_stream(dev))              # d[f|m] (and everything before it) is complete here
        if zeroed is not None:
            dbf, dbm, dgamma, dbeta = zeroed[4:].unbind(0)
        elif ctx.arena is not None:                     # the step's ONE zero-filled tensor for every layer's bias / BatchNorm gradients
            dbf, dbm, dgamma, dbeta = ctx.arena[0].take(ctx.arena[1], 4 * cout, dev).view(4, cout).unbind(0)
        else:
            dbf, dbm, dgamma, dbeta = torch.zeros((4, cout), dtype=torch.float32, device=dev).unbind(0)     # one fill, four rows
        if ctx.bn_train:
            _lib.check(L.read_bn_param_grads_groups(cout, groups, sums.data_ptr(), stat.data_ptr(), BN_EPS, dbf.data_ptr(),
                                                    dbm.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), st))
        else:
            _lib.check(L.read_bn_param_grads(cout, sums.data_ptr(), mean.data_ptr(), var.data_ptr(), BN_EPS, dbf.data_ptr(),
                                             dbm.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), st))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((H, W, cin), dtype=torch.float32, device=dev)
            poly = ctx.poly and ctx.wrefs[0]() is not None and ctx.wrefs[1]() is not None
            dilated = not poly and stride == 2 and k == 3 and H % 2 == 0 and W % 2 == 0
            if poly:
                # four 3x3 / stride-1 dgrads over the half-resolution d[f|m], one per pixel parity of dx (_poly_fragments)
                _, frags, w4 = _poly_fragments(ctx.wrefs[0](), ctx.wrefs[1](), cin, cout, k)  # of the CURRENT weights (cache: packed in forward)
                zero = _zero_params(L.read_conv_param_floats(cin // 2), dev)
                for par in range(4):
                    dxp = torch.empty((Ho, Wo, cin), dtype=torch.float32, device=dev)
                    _linear_conv(dfm, 2 * cp, frags[par], zero, cin // 2, 3, 1, dxp, wino=frags[par], w4=w4)
                    dx[par >> 1::2, par & 1::2] = dxp
            elif stride == 1 or dilated:
                # dgrad = the same MFMA convolution over d[f|m] with flipped, transposed weights; the two "gate halves" of
                # the kernel's output tile are simply the two halves of the input channels.  A 3x3 / stride-2 layer is the
                # stride-1 layer sampled at the even positions, so its dgrad is the stride-1 dgrad of d[f|m] spread onto the
                # even positions of a zero image — 4x the necessary MFMAs, still 7x faster than the generic VALU kernel
                # (1.34 ms per layer) because it runs on the Winograd kernel.
                d_in = dfm
                if dilated:
                    d_in = torch.zeros((H, W, 2 * cp), dtype=torch.float32, device=dev)
                    d_in[::2, ::2] = dfm
                if ctx.pack[3] is None:
                    _pack_dgrad(ctx.pack, wf, wm, cin, cout, k)
                wd, ev, wdw = ctx.pack[3]
                if ev is not None:                   # None: packed by the step's forward HIP graph, ordered by the replay itself
                    torch.cuda.current_stream().wait_event(ev)
                    wd.record_stream(torch.cuda.current_stream())
                    if wdw is not None:
                        wdw.record_stream(torch.cuda.current_stream())
                zero = _zero_params(L.read_conv_param_floats(cin // 2), dev)
                _linear_conv(d_in, 2 * cp, wd, zero, cin // 2, k, 1, dx, wino=wdw, w4=_w4_fits(2 * cp, cin // 2))
            else:
                if FLOP_LOG is not None:
                    FLOP_LOG.append(("dgrad_valu", 2.0 * Ho * Wo * cin * 2 * cout * k * k, 0))
                ws = torch.empty(L.read_conv_dgrad_generic_floats(cin, cout, k), dtype=torch.float32, device=dev)
                _lib.check(L.read_conv_dgrad_generic(dfm.data_ptr(), Ho, Wo, cin, cout, k, stride, wf.data_ptr(), wm.data_ptr(),
                                                     ws.data_ptr(), H, W, dx.data_ptr(), st))
        if not want_w:                                                      # frozen net: nobody asked for weight gradients
            return dx, None, dbf, None, dbm, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None
        if FLOP_LOG is not None:
            FLOP_LOG.append(("wgrad", 2.0 * Ho * Wo * cin * 2 * cout * k * k, int(L.read_conv_wgrad_family(cin, k, stride, H, W))))
        if ctx.side:
            # nothing downstream of this layer needs its weight gradient before the optimizer step, so it is computed on a
            # side stream while the main stream goes on with the dgrad chain of the layers below; the end of the backward
            # pass joins the streams (_join_side_stream, queued once per pass on the autograd engine)
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_event(ev_dfm)
            with torch.cuda.stream(side):
                _lib.check(L.read_conv_wgrad(x.data_ptr(), H, W, cin, dfm.data_ptr(), cout, k, stride, dwf.data_ptr(), dwm.data_ptr(),
                                             0, scratch.data_ptr(), n_scr, _lib.stream_ptr()))
            for t in (x, dfm, dwf, dwm, scratch):
                t.record_stream(side)
            # Returning dwf / dwm before the side stream has produced them is only safe when AccumulateGrad STEALS the
            # tensors (``.grad`` is None, no hooks): otherwise ``grad += dwf`` runs on the main stream right away — gradient
            # accumulation over several backward() calls, zero_grad(set_to_none=False), a layer used twice in one graph, a
            # tensor hook.  Then the main stream waits for this layer's wgrad here.
            stealable = True
            for r in ctx.wrefs:
                p = r()
                if p is None or p.grad is not None or getattr(p, '_backward_hooks', None) or getattr(p, '_post_accumulate_grad_hooks', None):
                    stealable = False
            if stealable:
                _queue_join(dev)
            else:
                ev_w = torch.cuda.Event()
                ev_w.record(side)
                main.wait_event(ev_w)
        else:
            _lib.check(L.read_conv_wgrad(x.data_ptr(), H, W, cin, dfm.data_ptr(), cout, k, stride, dwf.data_ptr(), dwm.data_ptr(), 0,
                                         scratch.data_ptr(), n_scr, st))
        return dx, dwf, dbf, dwm, dbm, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


class Up4Fn(torch.autograd.Function):
    """nn.Upsample(scale_factor=4, mode='bilinear') (unet.py:200) on an (H,W,C) NHWC image — or on nb vertically stacked
    items, each interpolated on its own — and its adjoint."""

    @staticmethod
    def forward(ctx, x, nb=1, v_num=1, v_den=1):
        x = x.contiguous()
        H, W, c = (int(v) for v in x.shape)
        bh = H // nb if nb > 1 else 0
        vh = bh * v_num // v_den
        out = torch.empty((4 * H, 4 * W, c), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().read_bilinear_up4_blocks(x.data_ptr(), H, W, c, out.data_ptr(), bh, vh, _lib.stream_ptr()))
        ctx.shape = (H, W, c, bh, vh)
        return out

    @staticmethod
    def backward(ctx, dout):
        H, W, c, bh, vh = ctx.shape
        dout = dout.contiguous()
        din = torch.empty((H, W, c), dtype=torch.float32, device=dout.device)
        _lib.check(_lib.lib().read_bilinear_up4_backward(dout.data_ptr(), H, W, c, din.data_ptr(), bh, vh, _lib.stream_ptr()))
        return din, None, None, None


class _HuberFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target):
        out, target = out.contiguous(), target.contiguous().to(out.device, torch.float32)
        n = out.numel()
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out)
        _lib.check(_lib.lib().read_huber_loss(out.data_ptr(), target.data_ptr(), n, 1.0 / n, loss.data_ptr(), grad.data_ptr(),
                                              _lib.stream_ptr()))
        ctx.save_for_backward(grad)
        return loss[0] / n

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def huber_loss(out, target):
    """``F.huber_loss(out, target)`` (delta 1, mean) with value and gradient from one HIP launch."""
    return _HuberFn.apply(out, target)


# -----------------------------------------------------------------------------------------------------------------------
# the UNet graph in training mode (READ/models/unet.py:202-285), one image, NHWC
# -----------------------------------------------------------------------------------------------------------------------
def _bc(net, path, x, k, stride=1, elu=True, blk=(1, 1, 1)):
    node = net
    for p in path.split('.'):
        node = node._modules[p]
    b = node.block
    n = b['norm']
    bn_train = int(bool(net.training))
    if bn_train and net.__dict__.get('_bn_per_item') and blk[0] > 1:
        bn_train = BN_PER_ITEM
    if bn_train and n.num_batches_tracked is not None:
        # nn.BatchNorm2d bookkeeping (momentum is fixed, so only a counter): one forward call per statistic group
        n.num_batches_tracked += blk[0] if bn_train == BN_PER_ITEM else 1
    return GatedConvFn.apply(x, b['conv_f'].weight, b['conv_f'].bias, b['conv_m'].weight, b['conv_m'].bias, n.weight, n.bias,
                             n.running_mean, n.running_var, k, stride, elu, *blk, bn_train, net.__dict__.get('_grad_arena'))


def _res_blocks(net, prefix, x, blk):
    for j in range(4):
        p = f"{prefix}.layers.{j}.main."
        x = _bc(net, p + "1", _bc(net, p + "0", x, 3, blk=blk), 3, elu=False, blk=blk) + x
    return x


def _scm(net, name, x, blk):
    y = _bc(net, name + ".main.0", x, 3, blk=blk)
    y = _bc(net, name + ".main.1", y, 1, blk=blk)
    y = _bc(net, name + ".main.2", y, 3, blk=blk)
    y = _bc(net, name + ".main.3", y, 1, blk=blk)
    return _bc(net, name + ".conv", torch.cat([x, y], -1), 1, elu=False, blk=blk)


def _down(x, s):                       # F.interpolate(scale_factor=1/s), nearest: source index = dst * s
    return x[::s, ::s]


def _up(x, s):                         # nearest up: source index = dst // s
    return x.repeat_interleave(s, 0).repeat_interleave(s, 1)


def unet_forward_train(net, x, x2, x4, x8, blk=(1, 1, 1)):
    """(h,w,8) NHWC pyramids -> (H,W,3); every tensor carries autograd history.  blk = (nb, v_num, v_den): the tensors hold
    nb items stacked vertically, v_num of every v_den rows of an item's block are valid (the rest: zero separator rows;
    ``stack_batch``).  Nearest resampling, concatenation, products and sums keep the separators zero by themselves."""
    capturing = torch.cuda.is_current_stream_capturing()
    if PACK_BATCH and not capturing:
        plan = _pack_plan(net, bool(net.training))         # every layer's parameter block and fragments: one launch
        if plan is not None:                               # None: a layout the batch packer does not take -> the per-layer packers
            plan.refresh()
    # a captured step's autograd graph is walked again every step: its nodes must not share one zero-filled tensor across steps
    net.__dict__['_grad_arena'] = None if capturing else _GradArena()
    z2, z4, z8 = _scm(net, "SCM2", x2, blk), _scm(net, "SCM1", x4, blk), _scm(net, "SCM0", x8, blk)
    res1 = _res_blocks(net, "Encoder.0", _bc(net, "feat_extract.0", x, 3, blk=blk), blk)
    z = _bc(net, "feat_extract.1", res1, 3, stride=2, blk=blk)
    z = z + _bc(net, "FAM2.merge", z * z2, 3, elu=False, blk=blk)
    res2 = _res_blocks(net, "Encoder.1", z, blk)
    z = _bc(net, "feat_extract.2", res2, 3, stride=2, blk=blk)
    z = z + _bc(net, "FAM1.merge", z * z4, 3, elu=False, blk=blk)
    res3 = _res_blocks(net, "Encoder.2", z, blk)
    z = _bc(net, "feat_extract.6", res3, 3, stride=2, blk=blk)
    z = z + _bc(net, "FAM0.merge", z * z8, 3, elu=False, blk=blk)
    z = _res_blocks(net, "Encoder.3", z, blk)
    z12, z13 = _down(res1, 2), _down(res1, 4)
    z21, z23 = _up(res2, 2), _down(res2, 2)
    z32, z31 = _up(res3, 2), _up(res3, 4)
    z43 = _up(z, 2)
    z42 = _up(z43, 2)
    z41 = _up(z42, 2)

    def aff(name, xs):
        return _bc(net, name + ".conv.1", _bc(net, name + ".conv.0", torch.cat(xs, -1), 1, blk=blk), 3, elu=False, blk=blk)
    r1, r2, r3 = aff("AFFs.0", [res1, z21, z31, z41]), aff("AFFs.1", [z12, res2, z32, z42]), aff("AFFs.2", [z13, z23, res3, z43])
    z = _res_blocks(net, "Decoder.0", z, blk)
    z = Up4Fn.apply(_bc(net, "feat_extract.7", z, 4, stride=2, blk=blk), *blk)
    z = _res_blocks(net, "Decoder.1", _bc(net, "Convs.0", torch.cat([z, r3], -1), 1, blk=blk), blk)
    z = Up4Fn.apply(_bc(net, "feat_extract.3", z, 4, stride=2, blk=blk), *blk)
    z = _res_blocks(net, "Decoder.2", _bc(net, "Convs.1", torch.cat([z, r2], -1), 1, blk=blk), blk)
    z = Up4Fn.apply(_bc(net, "feat_extract.4", z, 4, stride=2, blk=blk), *blk)
    z = _res_blocks(net, "Decoder.3", _bc(net, "Convs.2", torch.cat([z, r1], -1), 1, blk=blk), blk)
    return _bc(net, "feat_extract.5", z, 3, elu=False, blk=blk)


SEPARATOR_ROWS = 16        # zero rows between the items of a stacked batch at full resolution (1 row at the 1/16 scale)


def stack_batch(x_nchw, level):
    """(B,C,h,w) -> (B * (h + gap), w, C) NHWC: the items of a batch stacked vertically with SEPARATOR_ROWS >> level zero
    rows after each — the zero padding every convolution expects between neighbours, so one launch covers the batch."""
    B, c, h, w = x_nchw.shape
    x = x_nchw.permute(0, 2, 3, 1)
    if B > 1:
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, SEPARATOR_ROWS >> level))
    return x.reshape(-1, w, c).contiguous()


# -----------------------------------------------------------------------------------------------------------------------
# The training forward of the UNet from a HIP graph, its backward over the autograd graph of the capture
# -----------------------------------------------------------------------------------------------------------------------
# Round 3 read the training step as host-bound: 48 of 58 ms of host time inside the step's calls — 28 ms in the forward's 99 autograd
# nodes, 15 ms walking them backward (bench.py host_phases_ms_per_step).  (Round 4's tools/train_host_probe.py showed most of that to
# be time blocked on the full launch queue: the Python / launch path costs 17.5 ms per step at any crop size, the step is
# device-bound.  The capture below stays as an option; it is not what made the step faster.)
# The forward's launch sequence is the same every iteration — same shapes, same weights at the same addresses — so it is
# captured ONCE per batch geometry into a HIP graph (hipGraph through torch.cuda.CUDAGraph) and replayed: one graph launch
# (5 ms of host time) instead of ~600 Python-driven launches.  The capture runs the per-layer Python with autograd on, so it also
# leaves the forward's autograd graph behind, its saved activations living in the HIP graph's memory pool; every replay rewrites
# them in place, and the step's backward walks that SAME retained autograd graph (``torch.autograd.grad(..., retain_graph=True)``)
# eagerly, on the capture's stream (weight gradients included: ctx.side is off for captured nodes).
# Why not the backward from a graph as well (it was built first, with torch.cuda.make_graphed_callables, and measured,
# profiles/README.md): a replayed hipGraph ran its branches back to back on this ROCm — kernel-trace overlap factor 1.06 against
# 1.51 for the eager two-stream backward — so the step became GPU-bound at 63 ms instead of host-bound at 58.
# Inside the forward graph: packing the weights' fragment orders (they change with every optimizer step), batch-statistics
# BatchNorm with its running-buffer updates.  What the scheme cannot tolerate falls back to the per-layer path on its own: a second
# forward before the first one's backward (the retained graph holds ONE set of activations), parameters moved to other addresses
# (.cuda() / .to(): the key changes and a new graph is captured), hooks, anomaly mode, a failed capture.
# Measured (profiles/README.md, round 4): the forward's host time falls from 28 to 1.3 ms, but the step does not get shorter —
# 63 ms against 58: a step is ~1200 launches, the device spends ~8 us of gap on each however they are enqueued, and with the host
# out of the way the device side is the bound.  So the capture is an OPTION (READ_AMD_GRAPH_TRAIN=1, read_amd.train.GRAPH_TRAIN),
# not the default; what shortens the step is fewer launches (the pack plan above).
GRAPH_TRAIN = os.environ.get("READ_AMD_GRAPH_TRAIN", "0") == "1"
_GRAPH_CACHE_MAX = 4         # (geometry, mode) entries kept per net: an entry holds every activation of its step


class _HybridStepFn(torch.autograd.Function):
    """forward: copy the inputs into the capture's static buffers, replay the HIP graph; backward: the retained autograd graph."""

    @staticmethod
    def forward(ctx, step, n_in, *tensors):
        for s_in, x in zip(step.static_in, tensors[:n_in]):
            if s_in.data_ptr() != x.data_ptr():
                s_in.data.copy_(x)                   # .data: no version bump on a tensor the retained graph may have saved
        step.graph.replay()
        step.generation += 1                         # the static activations now belong to THIS forward
        ctx.step = step
        ctx.generation = step.generation
        ctx.n_tensors = len(tensors)
        return step.out.detach()

    @staticmethod
    def backward(ctx, gout):
        step = ctx.step
        if ctx.generation != step.generation:
            # a later forward has replayed the graph over the activations this backward would read (the `pending` latch was
            # released because the parameters moved in between: a delayed backward after an optimizer step, load_state_dict
            # or broadcast_parameters).  Eager autograd raises a version error here; returning gradients of the wrong
            # activations silently would be worse.
            raise RuntimeError("read_amd.train: backward of a captured forward whose buffers a later forward has replayed "
                               f"(generation {ctx.generation}, now {step.generation}); run backward before the next forward, "
                               "or set READ_AMD_GRAPH_TRAIN=0")
        step.pending = False
        grads = torch.autograd.grad([step.out], step.targets, [gout.contiguous()], retain_graph=True, allow_unused=True)
        res = [None] * ctx.n_tensors
        for slot, g in zip(step.target_slots, grads):
            res[slot] = g
        return (None, None, *res)


_WARNED_PENDING = False


class _GraphedStep:
    def __init__(self, graph, static_in, out, params):
        self.graph, self.static_in, self.out, self.params = graph, static_in, out, params
        # gradient targets inside the retained graph: the static inputs that require grad, then the parameters — and where each
        # one's gradient goes in _HybridStepFn's argument list (inputs first, parameters after them)
        self.targets, self.target_slots = [], []
        for i, t in enumerate(static_in):
            if t.requires_grad:
                self.targets.append(t)
                self.target_slots.append(i)
        for j, p_ in enumerate(params):
            self.targets.append(p_)
            self.target_slots.append(len(static_in) + j)
        self.pending = False         # a forward whose backward has not run yet: its saved activations are still needed
        self.pending_version = None  # ... and the parameter versions it saw (_graphed_step releases a stale latch)
        self.generation = 0          # replays so far: a backward checks that its forward was the LAST one (_HybridStepFn.backward)

    def __call__(self, xs):
        self.pending = True
        self.pending_version = tuple(p_._version for p_ in self.params)
        return _HybridStepFn.apply(self, len(xs), *xs, *self.params)


def _graph_key(net, xs, per_item):
    ts = net.__dict__.get('_flat_tensors')
    if ts is None:
        ts = net.__dict__['_flat_tensors'] = list(net.parameters()) + list(net.buffers())
    return (tuple(tuple(x.shape) for x in xs), tuple(bool(x.requires_grad) for x in xs), bool(net.training), bool(per_item),
            hash(tuple(t.data_ptr() for t in ts)), hash(tuple(bool(t.requires_grad) for t in ts)))


def _graphed_step(net, xs, per_item):
    """-> the captured step for this geometry, or None (then the caller runs the per-layer path)."""
    if not (GRAPH_TRAIN and torch.is_grad_enabled() and all(x.is_cuda and x.dtype == torch.float32 for x in xs)):
        return None
    if torch.cuda.is_current_stream_capturing() or torch.is_anomaly_enabled():
        return None
    if not any(p.requires_grad for p in net.parameters()) and not any(x.requires_grad for x in xs):
        return None
    if net._forward_hooks or net._backward_hooks or net._forward_pre_hooks:
        return None
    cache = net.__dict__.setdefault('_train_graphs', {})
    key = _graph_key(net, xs, per_item)
    g = cache.get(key)
    if g is None:
        if len(cache) >= _GRAPH_CACHE_MAX:
            cache.pop(next(iter(cache)))
        g = cache[key] = _capture_step(net, xs, per_item)
    if g is False:
        return None
    if g.pending:
        # A forward whose backward never ran (a validation pass under .train() with gradients on, an exception before backward())
        # would latch `pending` for good.  Its saved activations can only still be wanted while the weights are the ones it saw:
        # once an optimizer step has moved them, that forward's graph is spent — release the latch.
        if tuple(p_._version for p_ in g.params) != g.pending_version:
            g.pending = False
        else:
            global _WARNED_PENDING
            if not _WARNED_PENDING:
                _WARNED_PENDING = True
                import warnings
                warnings.warn("read_amd.train: a second forward before the first one's backward — this step runs on the per-layer "
                              "path (READ_AMD_GRAPH_TRAIN=1 replays ONE step's buffers)")
            return None                                  # a second forward before the first one's backward: that one keeps the buffers
    return g


def _capture_step(net, xs, per_item):
    params = [p_ for p_ in net.parameters() if p_.requires_grad]
    # the eager warm-up below would move the BatchNorm running buffers and num_batches_tracked in .train(): restored afterwards
    # (the capture itself executes nothing)
    saved = [b.detach().clone() for b in net.buffers()] if net.training else None
    cur = torch.cuda.current_stream()
    try:
        static_in = [x.detach().clone().requires_grad_(x.requires_grad) for x in xs]
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                        # warm-up: lazy initialisation, allocator, caches — off the capture
            out = _unet_forward_train_batch_eager(net, static_in, per_item)
            tg = [t for t in static_in if t.requires_grad] + params
            if tg:
                torch.autograd.grad([out], tg, [torch.ones_like(out)], allow_unused=True)
            del out
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if saved is not None:
            with torch.no_grad():
                for b, v in zip(net.buffers(), saved):
                    b.copy_(v)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):                        # autograd is on: the capture leaves the forward's autograd graph behind
            out = _unet_forward_train_batch_eager(net, static_in, per_item)
    except Exception as e:                                   # capture refused (an API the graph cannot hold, out of memory ...)
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        if saved is not None:
            with torch.no_grad():
                for b, v in zip(net.buffers(), saved):
                    b.copy_(v)
        import warnings
        warnings.warn(f"read_amd: HIP-graph capture of the UNet training forward failed ({type(e).__name__}: {e}); "
                      "this geometry runs layer by layer")
        return False
    return _GraphedStep(graph, static_in, out, params)


def unet_forward_train_batch(net, xs, per_item_statistics=False):
    """xs: four (B,8,h,w) pyramids -> (B,3,H,W) through ONE stacked image.  In ``.train()`` the BatchNorm layers normalise with
    the statistics of the whole batch — ``nn.BatchNorm2d`` on a (B,C,h,w) tensor — unless ``per_item_statistics``: then every item
    is its own batch of one and the running buffers move B times in item order, which is what B separate calls of the net do
    (``NetAndTexture.forward``, READ/models/compose.py:137-176).  The forward's launches come out of a HIP graph when the step
    can be captured (``_graphed_step``; its backward then walks the capture's retained autograd graph), else from the per-layer
    autograd nodes directly; ``LAST_STEP_PATH`` says which ('graph' / 'eager')."""
    global LAST_STEP_PATH
    g = _graphed_step(net, xs, per_item_statistics)
    if g is not None:
        LAST_STEP_PATH = 'graph'
        return g(xs)
    LAST_STEP_PATH = 'eager'
    return _unet_forward_train_batch_eager(net, xs, per_item_statistics)


LAST_STEP_PATH = None


def _unet_forward_train_batch_eager(net, xs, per_item_statistics=False):
    B, _, H, W = xs[0].shape
    if H % 16 or W % 16:
        raise ValueError(f"training crops must be multiples of 16, got {W}x{H}")
    blk = (B, H, H + SEPARATOR_ROWS) if B > 1 else (1, 1, 1)
    net.__dict__['_bn_per_item'] = bool(per_item_statistics)
    try:
        out = unet_forward_train(net, *[stack_batch(x, l) for l, x in enumerate(xs)], blk=blk)
    finally:
        net.__dict__['_bn_per_item'] = False
    out = out.reshape(B, -1, W, 3)[:, :H]
    return out.permute(0, 3, 1, 2)


# -----------------------------------------------------------------------------------------------------------------------
# descriptor optimizer
# -----------------------------------------------------------------------------------------------------------------------
class SparseDescriptorRMSprop:
    """RMSprop (torch defaults, READ/pipelines/ogl.py:16: alpha 0.99, eps 1e-8, no momentum) over the descriptor ROWS that
    the step's index maps touched.  The dense optimizer of the reference reads and writes all N rows of gradient, state
    and parameter every step; rows without a pixel have zero gradient, so only their second-moment decay is owed — it is
    applied lazily when the row is touched again (csrc/train.hip), which makes every row's trajectory equal the dense one.

    Duck-types what ``train.py`` uses of a torch optimizer: ``step()``, ``zero_grad()``, ``param_groups`` (lr),
    ``state_dict()`` / ``load_state_dict()``."""

    def __init__(self, textures, lr=1e-1, alpha=0.99, eps=1e-8):
        self.textures = list(textures)
        self.param_groups = [{'params': [t.texture_ for t in self.textures], 'lr': lr, 'alpha': alpha, 'eps': eps}]
        self.state = {}

    def _state(self, tex):
        s = self.state.get(id(tex))
        if s is None:
            rows = tex.training_rows()
            s = self.state[id(tex)] = {'step': 0, 'sq': torch.zeros_like(rows),
                                       'stamp': torch.zeros(rows.shape[0], dtype=torch.int32, device=rows.device)}
        return s

    def step(self, closure=None):
        g = self.param_groups[0]
        L = _lib.lib()
        for tex in self.textures:
            if tex._touched:                      # someone asked for the dense gradient rows: everything goes through them
                tex.grad_rows()
            pend = tex.take_pending()
            ids = tex.take_touched()
            if pend is None and ids is None:
                continue
            s = self._state(tex)
            s['step'] += 1
            rows = tex.training_rows()
            if pend is not None:
                # pairs sorted by id, runs summed in order, each row updated once (no gradient table, deterministic)
                pids, pg = pend
                sorted_ids, perm = torch.sort(pids, stable=True)
                if 'scratch' not in s:
                    s['scratch'] = torch.zeros(int(L.read_rmsprop_sorted_scratch_ints()), dtype=torch.int32, device=rows.device)
                _lib.check(L.read_rmsprop_sorted(rows.data_ptr(), s['sq'].data_ptr(), s['stamp'].data_ptr(), int(rows.shape[1]),
                                                 int(rows.shape[0]), sorted_ids.data_ptr(), perm.data_ptr(), pg.data_ptr(),
                                                 int(pids.numel()), s['step'], float(g['lr']), float(g['alpha']), float(g['eps']),
                                                 s['scratch'].data_ptr(), _lib.stream_ptr()), "read_rmsprop_sorted")
            else:
                grad = tex.grad_rows()
                _lib.check(L.read_rmsprop_sparse(rows.data_ptr(), s['sq'].data_ptr(), grad.data_ptr(), s['stamp'].data_ptr(),
                                                 int(rows.shape[1]), int(rows.shape[0]), ids.data_ptr(), ids.numel(),
                                                 s['step'], float(g['lr']), float(g['alpha']), float(g['eps']),
                                                 _lib.stream_ptr()), "read_rmsprop_sparse")
            tex.rows_changed()

    def zero_grad(self, set_to_none=True):
        pass                                      # step() leaves every touched gradient row zero again

    def state_dict(self):
        return {'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}],
                'state': [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self._state(t).items() if k != 'scratch'}
                          for t in self.textures]}

    def load_state_dict(self, sd):
        self.param_groups[0].update(sd['param_groups'][0])
        for t, s in zip(self.textures, sd['state']):
            cur = self._state(t)
            cur['step'] = s['step']
            cur['sq'].copy_(s['sq'])
            cur['stamp'].copy_(s['stamp'])
