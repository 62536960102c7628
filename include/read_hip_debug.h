/* read_hip_debug.h — entry points of the DEBUG build of the library only (libreadhip_debug.so: python -m read_amd.build --debug,
 * -DREAD_DEBUG_KNOBS).  Measurement probes and the kernel timeline used by tools/ (issue_probe.py, operand_probe.py,
 * mfma_probe.py, trace_conv.py); none of them is exported by libreadhip.so, the product. */
#ifndef READ_HIP_DEBUG_H
#define READ_HIP_DEBUG_H
#include "read_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Debug timeline of the following gated-conv launches: 64 bytes per workgroup (direct kernels: s_memrealtime at
 * entry / after prologue / after the k-loop / at exit, HW_ID, XCC_ID, blockIdx.x/y) or per wave (Winograd kernel:
 * entry, end of prologue, ticks spent in unit epilogues split three ways, exit, HW_ID) in `buf` (device);
 * read by tools/trace_conv.py.  NULL switches tracing off (the default). */
int read_debug_set_trace(void *buf, size_t bytes);
/* Debug probe: `blocks` workgroups x 4 waves, each wave issues iters*nacc*4 v_mfma_f32_32x32x2_f32
 * (4096 FLOP each) from registers — the sustained matrix-core ceiling for the conv kernels.
 * scratch: >= blocks*256 floats (never written in practice). */
int read_debug_mfma_probe(int blocks, int iters, int nacc, float *scratch, void *stream);
/* Issue-model probe (debug): every wave runs `iters` rounds of 16 fp32 MFMAs (kind 0: v_mfma_f32_32x32x2_f32, 1:
 * v_mfma_f32_16x16x4_f32), each followed by K filler instructions of type `filler` (0 independent v_add_f32, 1 ds_read_b128,
 * 2 s_add_u32, 3 dependent v_add_f32 chain, 4 global_load_dwordx4 from gsrc); cycles[blocks * 4] = s_memtime span per wave. */
/* Operand probe (debug): v_mfma_f32_16x16x4_f32 rate by operand register pattern (mode 0 one A/B pair, 1 a pair per MFMA, 2 the
 * Winograd kernel's float4-component pattern, 3 as 2 with the B operands re-read from LDS every round). */
int read_debug_operand_probe(int mode, int blocks, int iters, float *scratch, unsigned long long *cycles, void *stream);
int read_debug_issue_probe(int kind, int filler, int K, int blocks, int iters, float *scratch, unsigned long long *cycles,
                           const float *gsrc, void *stream);

/* fp32 VALU rate probe: every wave issues iters * 64 instructions of one form over eight accumulators (mode 0 v_fma_f32, 1
 * v_fmac_f32 with an SGPR multiplier, 2 v_pk_fma_f32, 3 v_pk_fma_f32 with an SGPR pair and a broadcast op_sel, 4 v_pk_add_f32, 5
 * v_pk_mul_f32, 6 v_pk_fma_f32 with the F(4x4) transform's half-selects, 7 v_add_f32); cycles[blocks * 4] = s_memtime span per wave. */
int read_debug_valu_probe(int mode, int blocks, int iters, float *scratch, unsigned long long *cycles, void *stream);

/* Kernel boundary against grid barrier: `phases` dependent phases over `blocks` workgroups, each touching floats_per_block floats
 * of its slice of buf — mode 0 as dependent launches, mode 1 as one persistent launch with grid barriers (counter_and_flag: two
 * unsigned; [1] is set when a barrier gave up waiting). */
int read_debug_chain_probe(int mode, int blocks, int phases, int floats_per_block, float *buf, unsigned *counter_and_flag,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
