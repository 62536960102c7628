"""ctypes binding of libreadhip.so — the only road from Python into the render path.

There is deliberately NO fallback: if the library is missing or a call fails the caller gets a
``RuntimeError`` carrying ``read_last_error()``.  Nothing here touches ``oracle/``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# READ_HIP_DEBUG=1: the -DREAD_DEBUG_KNOBS build (attribution probes for tools/; python -m read_amd.build --debug)
LIB_PATH = os.path.join(_HERE, "libreadhip_debug.so" if os.environ.get("READ_HIP_DEBUG") else "libreadhip.so")
if os.environ.get("READ_HIP_VARIANT"):            # tools only: an A/B build of the same sources (tools/w4h_ab.py)
    LIB_PATH = os.path.join(_HERE, "libreadhip_v_%s.so" % os.environ["READ_HIP_VARIANT"])

READ_MAX_LEVELS = 5
READ_CONV_MAX_SRC = 4
DESC_CHANNELS = 8


class ReadHipError(RuntimeError):
    pass


class ConvSrc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("C", C.c_int), ("srcH", C.c_int), ("srcW", C.c_int), ("shift", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("n_src", C.c_int), ("src", ConvSrc * READ_CONV_MAX_SRC), ("mul", C.c_void_p),
        ("inH", C.c_int), ("inW", C.c_int), ("Cout", C.c_int), ("ksize", C.c_int), ("stride", C.c_int),
        ("elu", C.c_int), ("wpacked", C.c_void_p), ("params", C.c_void_p), ("residual", C.c_void_p),
        ("out", C.c_void_p), ("out_cstride", C.c_int), ("out_fill", C.c_float), ("fill_pad", C.c_int),
        ("config", C.c_int), ("wpacked_wino", C.c_void_p), ("wpacked_w16", C.c_void_p), ("wpacked_w4", C.c_void_p),
        ("linear", C.c_int),
        ("pre", C.c_void_p), ("pre_cstride", C.c_int), ("pre_f_off", C.c_int), ("pre_m_off", C.c_int),
        ("pre_shift", C.c_int), ("preH", C.c_int), ("preW", C.c_int),
        ("out_gated", C.c_void_p), ("block_h", C.c_int), ("valid_h", C.c_int),
        ("wpacked_sc", C.c_void_p), ("pre_bilinear", C.c_int),
        ("wpacked_w4h", C.c_void_p), ("wpacked_d3h", C.c_void_p), ("wpacked_t3h", C.c_void_p),
    ]


class PackJob(C.Structure):
    """read_pack_job (include/read_hip.h): one entry of read_conv_pack_batch's table."""
    _fields_ = [("kind", C.c_int), ("mode", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("ksize", C.c_int), ("kc", C.c_int),
                ("Cp", C.c_int), ("first_block", C.c_int), ("nblocks", C.c_int), ("eps", C.c_float),
                ("wf", C.c_void_p), ("wm", C.c_void_p), ("bf", C.c_void_p), ("bm", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p), ("out", C.c_void_p), ("total", C.c_longlong)]


PACK_PARAMS, PACK_DIRECT, PACK_WINO, PACK_W4 = 0, 1, 2, 3


class SplatGlOpts(C.Structure):
    _fields_ = [("point_size", C.c_float), ("relative", C.c_int), ("min_point_size", C.c_float),
                ("discard", C.c_void_p), ("drop_threshold", C.c_uint32), ("drop_seed", C.c_uint32),
                ("perturb", C.c_void_p), ("perturb_amp", C.c_float), ("perturb_seed", C.c_uint32),
                ("point_sizes", C.c_void_p)]


_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
_pp = C.POINTER(C.c_void_p)

# entry points of the debug build only (include/read_hip_debug.h; READ_HIP_DEBUG=1 loads libreadhip_debug.so)
DEBUG_SIGNATURES = {
    "read_debug_set_trace": (_i, [_vp, _sz]),
    "read_debug_mfma_probe": (_i, [_i, _i, _i, _vp, _vp]),
    "read_debug_operand_probe": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "read_debug_valu_probe": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "read_debug_issue_probe": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "read_debug_chain_probe": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
}

# name -> (restype, argtypes): every symbol include/read_hip.h declares
SIGNATURES = {
    "read_last_error": (C.c_char_p, []),
    "read_abi_version": (_i, []),
    "read_device_arch": (_i, [C.c_char_p, _i]),
    "read_tuning_set": (_i, [C.c_char_p, _i]),
    "read_tuning_get": (_i, [C.c_char_p, C.POINTER(_i)]),
    "read_tuning_key": (C.c_char_p, [_i]),
    "read_splat_workspace_bytes": (_sz, [_i, _i, _i]),
    "read_splat_workspace_init": (_i, [_vp, _sz, _vp]),
    "read_splat_forward": (_i, [_vp, _i64, C.POINTER(_f), _i, _i, _i, _i, _pp, _pp, _vp, _sz, _vp]),
    "read_splat_cells_bytes": (_sz, [_i64]),
    "read_splat_cells_build_host": (_i, [_vp, _i64, _vp, _sz]),
    "read_splat_cells_invalidate": (_i, [_vp, _i64]),
    "read_splat_forward_cells": (_i, [_vp, _vp, _i64, C.POINTER(_f), _i, _i, _i, _i, _pp, _pp, _vp, _sz, _vp]),
    "read_splat_hint_next_camera": (_i, [_vp, C.POINTER(_f)]),
    "read_splat_profile_last": (_i, [C.POINTER(_f)]),
    "read_mfma_f32_rate_probe": (_i, [_i, _vp, C.POINTER(C.c_double), _vp]),
    "read_splat_forward_gl": (_i, [_vp, _i64, C.POINTER(_f), _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "read_index_to_float": (_i, [_vp, _i64, _vp, _vp]),
    "read_splat_project_points": (_i, [_vp, _i64, C.POINTER(_f), _i, _i, _vp, _vp, _vp]),
    "read_texture_to_rows": (_i, [_vp, _i64, _i, _vp, _vp]),
    "read_rows_to_texture": (_i, [_vp, _i64, _i, _vp, _vp]),
    "read_gather_forward": (_i, [_vp, _i64, _i, _i, _pp, C.POINTER(_i64), _pp, _i, _vp]),
    "read_gather_forward_ss": (_i, [_vp, _i64, _i, _i, _i, _pp, C.POINTER(_i), C.POINTER(_i), _i, _pp, _i, _vp]),
    "read_gather_backward": (_i, [_vp, _i64, _i, _i, _pp, C.POINTER(_i64), _pp, _vp]),
    "read_bilinear_down": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "read_bilinear_down_backward": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "read_conv_packed_floats": (_sz, [_i, _i, _i]),
    "read_conv_param_floats": (_sz, [_i]),
    "read_conv_pack_weights_host": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "read_conv_pack_params_host": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "read_conv_wino_floats": (_sz, [_i, _i]),
    "read_conv_pack_wino_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_pack_w16_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_w4_floats": (_sz, [_i, _i]),
    "read_conv_pack_w4_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_w4h_floats": (_sz, [_i, _i]),
    "read_conv_pack_w4h_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_d3h_floats": (_sz, [_i, _i]),
    "read_conv_pack_d3h_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_t3h_floats": (_sz, [_i, _i]),
    "read_conv_pack_t3h_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_dkh_floats": (_sz, [_i, _i, _i]),
    "read_conv_pack_dkh_host": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "read_gated_conv_forward": (_i, [C.POINTER(ConvDesc), _vp]),
    "read_conv_kernel_family": (_i, [_vp]),
    "read_conv_sc_floats": (_sz, [_i, _i]),
    "read_conv_pack_sc_host": (_i, [_i, _i, _vp, _vp, _vp]),
    "read_conv_pack_job_prepare": (_i, [C.POINTER(PackJob)]),
    "read_conv_pack_batch": (_i, [_vp, _i, _i, _vp]),
    "read_conv_config_count": (_i, []),
    "read_conv_config_name": (C.c_char_p, [_i]),
    "read_bilinear_up4": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "read_conv_pack_params_device": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "read_conv_pack_weights_device": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "read_conv_pack_wino_device": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "read_conv_dgrad_wino_floats": (_sz, [_i, _i]),
    "read_conv_pack_dgrad_wino_device": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "read_conv_pack_w4_device": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "read_conv_dgrad_w4_floats": (_sz, [_i, _i]),
    "read_conv_pack_dgrad_w4_device": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "read_conv_dgrad_packed_floats": (_sz, [_i, _i, _i]),
    "read_conv_pack_dgrad_device": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "read_gate_forward": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "read_gate_backward": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "read_bn_param_grads": (_i, [_i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "read_bn_train_forward": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "read_gate_backward_bn": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp]),
    "read_bn_param_grads_groups": (_i, [_i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "read_conv_dgrad_generic_floats": (_sz, [_i, _i, _i]),
    "read_conv_dgrad_generic": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "read_conv_wgrad_scratch_floats": (_sz, [_i, _i, _i, _i]),
    "read_conv_wgrad": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "read_conv_wgrad_family": (_i, [_i, _i, _i, _i, _i]),
    "read_bilinear_up4_blocks": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "read_bilinear_up4_backward": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "read_huber_loss": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp]),
    "read_rmsprop_sparse": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _i64, _i, _f, _f, _f, _vp]),
    "read_rmsprop_sorted_scratch_ints": (_sz, []),
    "read_rmsprop_sorted": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _i64, _i, _f, _f, _f, _vp, _vp]),
    "read_unet_layer_count": (_i, []),
    "read_unet_layer_info": (_i, [_i, C.POINTER(C.c_char_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                  C.POINTER(_i), C.POINTER(_i)]),
    "read_unet_raw_floats": (_sz, []),
    "read_unet_packed_floats": (_sz, []),
    "read_unet_pack_host": (_i, [_vp, _f, _vp]),
    "read_unet_workspace_bytes": (_sz, [_i, _i]),
    "read_unet_create": (_i, [_pp, _vp, _i, _i, _vp, _sz]),
    "read_unet_create_layout": (_i, [_pp, _vp, _i, _i, _vp, _sz, _i]),
    "read_unet_packed_floats_layout": (_sz, [_i]),
    "read_unet_pack_host_layout": (_i, [_vp, _f, _vp, _i]),
    "read_unet_destroy": (None, [_vp]),
    "read_unet_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "read_unet_launch_count": (_i, [_vp]),
    "read_unet_launch_label": (C.c_char_p, [_vp, _i]),
    "read_unet_launch_info": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                   C.POINTER(_i), C.POINTER(_i)]),
    "read_unet_profile": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, C.POINTER(_f), C.POINTER(C.c_double),
                               C.POINTER(_i)]),
    "read_unet_debug_tensor": (_vp, [_vp, C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
}

_LIB = None


def lib():
    """The loaded library; raises ReadHipError when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ReadHipError(
                f"{LIB_PATH} is missing: the HIP extension is required (there is no CPU fallback). "
                "Build it with `python -m read_amd.build` (or __graft_entry__.build()).")
        # torch first: its wheel bundles a HIP runtime of its own, and the process must end up with ONE — loaded in this order
        # libreadhip.so binds to the runtime torch brought (same soname); the other way round (this library first, through its
        # RUNPATH to /opt/rocm, then torch's copy) the second runtime reports "no ROCm-capable device" at the first launch — seen
        # when __graft_entry__.build() and smoke() ran in one process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        table = dict(SIGNATURES)
        if os.environ.get("READ_HIP_DEBUG"):
            table.update(DEBUG_SIGNATURES)
        for name, (res, args) in table.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("READ_CONV_WAVE"):          # A/B switch for tests: wave-autonomous conv kernels
            L.read_tuning_set(b"conv_wave", int(os.environ["READ_CONV_WAVE"]))
        for kv in filter(None, os.environ.get("READ_TUNE", "").split(",")):   # A/B switches: "key=value,key=value"
            k, v = kv.split("=")
            if L.read_tuning_set(k.encode(), int(v)) != 0:
                raise ReadHipError(f"READ_TUNE: {L.read_last_error().decode()}")
        _LIB = L
    return _LIB


def tuning_state():
    """{key: value} of every read_tuning_set knob of the loaded library (recorded by bench.py)."""
    L = lib()
    out, i = {}, 0
    while True:
        k = L.read_tuning_key(i)
        if not k:
            return out
        v = C.c_int()
        check(L.read_tuning_get(k, C.byref(v)), "read_tuning_get")
        out[k.decode()] = v.value
        i += 1


def check(rc, what=""):
    if rc != 0:
        msg = lib().read_last_error().decode("utf-8", "replace")
        raise ReadHipError(f"{what + ': ' if what else ''}libreadhip error {rc}: {msg}")


def ptr_array(ptrs):
    """Host array of device pointers (None -> NULL)."""
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise ReadHipError("no HIP device visible: READ's render path runs on the MI355X only "
                           "(there is no CPU fallback in read_amd)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
