"""CPU (hipcc cross-compiles): properties of the generated gfx950 code that round 3's measurements depend on and that a source
change can silently lose — found in the ISA, not in any test result (profiles/README.md):

  * the Winograd F(4x4,3x3) inference kernels keep their cross-unit pipeline: no `s_waitcnt vmcnt(0)` at the head of the unit
    loop (a second path through the epilogue once put one there: -1.8 % frames/s), no scratch, one wave per SIMD's registers;
  * the wgrad kernel's ring of five is counted by the compiler: `vmcnt(40)` in front of its MFMA groups, not a drain;
  * the whole-quad 1x1 pixel-lane kernel waits for `vmcnt(3)` in front of its MFMA groups;
  * the small-Cout vector-pipe kernel keeps its software-pipelined scalar weight loads: 1728 v_fmac_f32 with SGPR multipliers per
    pixel, 144 s_load_dwordx16 issued one group ahead, no scratch;
  * the Winograd-domain wgrad kernel fits two waves per SIMD (<= 256 registers, no scratch) and issues its 36 MFMAs per iteration;
  * the rasteriser's pass A keeps five waves per SIMD (<= 96 registers, no scratch): round 5's pipelined variant needed 149 - 175 and
    lost 8 - 19 us per frame to the occupancy it gave up (profiles/r5_pass_a_pipe_ab.md).
"""
import os
import re
import subprocess

import pytest

from read_amd import build as hip_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asm(src, tmp_path_factory):
    out = tmp_path_factory.getbasetemp() / (src + ".s")
    if not out.exists():
        cmd = [hip_build._hipcc()] + hip_build.FLAGS + hip_build.PER_FILE.get(src, []) + \
              ["-S", "--cuda-device-only", os.path.join(hip_build.CSRC, src), "-o", str(out)]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out.read_text()


def _function(asm, mangled_part):
    m = re.search(r"^(_Z\S*" + re.escape(mangled_part) + r"\S*):\s*;", asm, flags=re.M)
    assert m, mangled_part
    body = asm[m.start():]
    return m.group(1), body[:body.index(".Lfunc_end")]


def _meta(asm, name, key):
    m = re.search(r"\.set " + re.escape(name) + r"\." + key + r", (\d+)", asm)
    assert m, (name, key)
    return int(m.group(1))


@pytest.fixture(scope="module")
def conv_asm(tmp_path_factory):
    return _asm("conv.hip", tmp_path_factory)


def test_f4_inference_kernels_keep_the_cross_unit_pipeline(conv_asm):
    for variant in ("gated_conv_wino4_kernelILb0ELi0ELi0E", "gated_conv_wino4_kernelILb1ELi0ELi0E"):
        name, body = _function(conv_asm, variant)
        assert _meta(conv_asm, name, "private_seg_size") == 0, "scratch in the F(4x4) kernel"
        assert _meta(conv_asm, name, "num_vgpr") == 256 and _meta(conv_asm, name, "num_agpr") >= 144
        lines = body.split("\n")
        heads = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
        assert heads, "unit loop not found"
        head = "\n".join(lines[heads[0]:heads[0] + 12])
        assert "v_mfma_f32_16x16x4_f32" in head                                  # the loop starts with the first stage's MFMAs
        assert "vmcnt(0)" not in head, "the unit loop drains the previous unit's stores and loads:\n" + head
        assert body.count("v_mfma_f32_16x16x4_f32") == 288                        # first + steady stage, 144 each, nothing duplicated


def test_split_operand_f4_kernel_keeps_its_registers_and_pipeline(conv_asm):
    """gated_conv_wino4h_kernel (f16 matrix cores, round 6): one wave per SIMD with the accumulators in the accumulation file, no
    scratch (a denser transform cadence once flipped hipcc's allocation to 256 + 22 registers and 1.4 KB of scratch: 720 us per
    launch instead of 60), 108 MFMAs per stage instance, the patch loads and both operand rings inside the stage, a unit loop
    that does not drain the memory pipeline — and the specialised-wave variant (measured slower) is not in the product."""
    for variant in ("gated_conv_wino4h_kernelILi0E",):
        name, body = _function(conv_asm, variant)
        assert _meta(conv_asm, name, "private_seg_size") == 0, "scratch in the split-operand F(4x4) kernel"
        assert _meta(conv_asm, name, "num_vgpr") == 256 and _meta(conv_asm, name, "num_agpr") >= 144
        assert body.count("v_mfma_f32_16x16x32_f16") == 216                      # first + steady stage
        assert body.count("v_cvt_pk_f16_f32") == 3 * 72                          # prologue + two stage instances: hi and lo of 36 frequencies
        assert "v_mfma_f32_16x16x4_f32" not in body
        lines = body.split("\n")
        heads = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
        assert heads, "unit loop not found"
        head = "\n".join(lines[heads[0]:heads[0] + 12])
        assert "vmcnt(0)" not in head, "the unit loop drains the previous unit's stores and loads:\n" + head
    assert "gated_conv_wino4h2_kernel" not in conv_asm


def test_direct_split_operand_kernel_fits_two_waves_per_simd(conv_asm):
    """gated_conv_d3h_kernel (round 6): eight waves per workgroup = two per SIMD, so at most 256 registers per wave, no scratch; 216 MFMAs
    per stage (9 taps x 4 pixel blocks x 2 row blocks x 3 piece pairs); the weight loads stay ahead of their use (hipcc sinks an
    unpinned load to its first use: the first version waited vmcnt(0) behind every tap's loads)."""
    for variant in ("gated_conv_d3h_kernelILb0ELi0E", "gated_conv_d3h_kernelILb1ELi0E"):
        name, body = _function(conv_asm, variant)
        assert _meta(conv_asm, name, "private_seg_size") == 0, "scratch in the direct split-operand kernel"
        assert _meta(conv_asm, name, "num_vgpr") + _meta(conv_asm, name, "num_agpr") <= 256
        assert body.count("v_mfma_f32_16x16x32_f16") == 216
        lines = [l.strip() for l in body.split("\n")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma_f32_16x16x32_f16")]
        loop = lines[mf[0]:mf[-1]]
        assert sum(1 for l in loop if l.startswith("s_waitcnt") and "vmcnt(0)" in l) <= 2, "the stage drains its weight loads"


def test_training_kernels_count_their_loads(conv_asm, tmp_path_factory):
    for variant in ("gated_conv_wino4_kernelILb0ELi0ELi1E", "gated_conv_wino4_kernelILb0ELi0ELi2E"):
        name, body = _function(conv_asm, variant)
        lines = body.split("\n")
        head = "\n".join(lines[[i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l][0]:][:12])
        assert "vmcnt(0)" not in head and _meta(conv_asm, name, "private_seg_size") == 0
    train = _asm("train.hip", tmp_path_factory)
    _, body = _function(train, "wgrad_mfma_kernelILi9ELb0E")
    assert len(re.findall(r"s_waitcnt vmcnt\(40\)\s*\n\s*v_mfma_f32_32x32x2_f32", body)) >= 5, \
        "the wgrad ring is no longer four steps of ten loads ahead of its MFMAs"


def test_pixel_lane_kernel_prefetches(conv_asm):
    _, body = _function(conv_asm, "gated_conv_px_kernelILi1ELi2ELb1E")
    assert len(re.findall(r"s_waitcnt vmcnt\(3\)[^\n]*\n\s*v_mfma_f32_32x32x2_f32", body)) >= 4


def test_small_cout_kernel_streams_weights_through_sgprs(conv_asm):
    name, body = _function(conv_asm, "gated_conv_smallc_kernelILi32ELi8ELi1ELi3E")
    assert _meta(conv_asm, name, "private_seg_size") == 0
    assert _meta(conv_asm, name, "num_vgpr") <= 64                          # eight waves per SIMD
    fmacs = re.findall(r"v_fmac_f32 v\d+, s\d+, v\d+", body)
    assert len(fmacs) == 9 * 32 * 6, len(fmacs)
    assert body.count("s_load_dwordx16") == 144
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    # a weight request is followed by the FMAs of the PREVIOUS group before anything waits for it — except the first group of
    # each of the four LDS phases
    exposed = 0
    for i, l in enumerate(lines):
        if l.startswith("s_load_dwordx16"):
            nxt = [x for x in lines[i + 1:i + 8] if not x.startswith(("s_movk_i32", "s_mov_b32", "s_load_dwordx16"))]
            exposed += nxt[0].startswith("s_waitcnt lgkmcnt")
    assert exposed <= 4 + 2, exposed                                        # + the kernel-argument loads at entry


@pytest.fixture(scope="module")
def train_asm(tmp_path_factory):
    return _asm("train.hip", tmp_path_factory)


def test_winograd_wgrad_kernel_fits_two_waves_per_simd(train_asm):
    name, body = _function(train_asm, "wgrad_wino4_kernel")
    assert _meta(train_asm, name, "private_seg_size") == 0
    assert _meta(train_asm, name, "num_vgpr") + _meta(train_asm, name, "num_agpr") <= 256
    assert body.count("v_mfma_f32_32x32x2_f32") == 36                     # 9 frequencies x 4 tile pairs per iteration, nothing duplicated
    assert body.count("buffer_load_dword ") + body.count("buffer_load_dword\t") >= 52 or body.count("buffer_load_dword") >= 52


def test_two_waves_per_simd_f4_kernel_is_not_in_the_product(conv_asm):
    """Round 5 ran the frequency-split two-waves-per-SIMD F(4x4) kernel: results equal, 3-9 % slower at every level
    (profiles/r5_w4x2_ab.json, DESIGN.md 12.1 d).  It lives on in the debug library only (-DREAD_DEBUG_KNOBS) as the record of
    the experiment; the product's device code does not contain it and the release library does not know its knob."""
    assert "gated_conv_wino4x2_kernel" not in conv_asm
    from read_amd import _lib
    if not os.environ.get("READ_HIP_DEBUG"):
        L = _lib.lib()
        assert L.read_tuning_set(b"conv_w4x2", 1) != 0
        keys, i = [], 0
        while L.read_tuning_key(i):
            keys.append(L.read_tuning_key(i).decode())
            i += 1
        assert "conv_w4x2" not in keys and "conv_w4" in keys


def test_rasteriser_pass_a_keeps_five_waves_per_simd(tmp_path_factory):
    """cells_pass_kernel<A> of the product (no statistics, bound image through L1, LDS table, binned candidates): what hides its chain
    of dependent round trips is the number of resident waves (profiles/r5_pass_a_pipe_ab.md) — 512 registers / 5 waves = 102, the
    allocation granule is 8."""
    asm = _asm("splat.hip", tmp_path_factory)
    name, body = _function(asm, "cells_pass_kernelILb0ELb0ELb0ELb1ELb1E")
    assert _meta(asm, name, "private_seg_size") == 0, "scratch in pass A"
    assert _meta(asm, name, "num_vgpr") + _meta(asm, name, "num_agpr") <= 96, "pass A no longer fits five waves per SIMD"
    assert "global_atomic_add" in body and "ds_write" in body           # the bin reservation and the candidate queue are in this kernel
