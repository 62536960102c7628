// Measurement probes (debug): what does the chip sustain on the instruction the conv kernels are built on?
#include "common.h"

using namespace readhip;
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {
// NACC independent accumulators per wave, `iters` rounds of NACC*4 MFMAs each; operands live in registers
// and are perturbed every round so the multiplier inputs are not constant.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_probe_kernel(float *out, int iters, float seed)
{
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float a0 = seed + 0.001f * threadIdx.x, a1 = a0 * 0.5f + 0.1f, a2 = a0 - 0.3f, a3 = 0.7f - a0;
    float b0 = 1.0f - 0.002f * threadIdx.x, b1 = b0 * 0.25f, b2 = b0 + 0.2f, b3 = 0.3f * b0 - 0.1f;
    if (seed < 0.0f) {          // "random" mode: per-lane hashed operands with full mantissa entropy
        unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
        auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (float)(int)h * (1.0f / 2147483648.0f); };
        a0 = rnd(); a1 = rnd(); a2 = rnd(); a3 = rnd(); b0 = rnd(); b1 = rnd(); b2 = rnd(); b3 = rnd();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[i], 0, 0, 0);
        }
        if (seed < 0.0f) {      // decorrelate every round (values stay in [-1,1))
            a0 = a0 * 1.7f; a0 -= (float)(int)a0; a1 = -a1 * 1.3f; a1 -= (float)(int)a1; a2 = a2 * 1.9f; a2 -= (float)(int)a2;
            a3 = -a3 * 1.1f; a3 -= (float)(int)a3; b0 = b0 * 1.5f; b0 -= (float)(int)b0; b1 = -b1 * 1.21f; b1 -= (float)(int)b1;
            b2 = b2 * 1.83f; b2 -= (float)(int)b2; b3 = -b3 * 1.37f; b3 -= (float)(int)b3;
        } else {
            a0 = a0 * 0.999f + 0.0007f; a1 = a1 * 1.0003f - 0.0002f; a2 = -a2; a3 = a3 * 0.9995f;
            b0 = b0 * 1.0001f - 0.0001f; b1 = -b1; b2 = b2 * 0.9997f + 0.0001f; b3 = b3 * 1.0002f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1234.5678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;   // keeps everything live
}

// ---- issue-model probe: what does ONE filler instruction in the shadow of an fp32 MFMA cost, alone on a SIMD and beside a
// second wave?  Each wave runs `iters` rounds of 16 MFMAs on independent accumulators; after every MFMA come K fillers
// (pinned with sched_barrier): FT 0 independent v_add_f32, 1 ds_read_b128 (waited once per round), 2 s_add_u32, 3 v_add_f32 in
// one dependent chain, 4 global_load_dwordx4 (L2-resident line, waited once per round).  KIND 0: v_mfma_f32_32x32x2_f32
// (8 accumulators, 2 MFMAs each per round), 1: v_mfma_f32_16x16x4_f32 (16 accumulators), 2: v_mfma_f32_16x16x32_bf16 (16
// accumulators; round 5, DESIGN.md 12.1 c: does a vector instruction beside a bf16 MFMA cost what it costs beside an fp32 one?).
// Lane 0 of every wave stores its s_memtime span.
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int KIND, int FT, int K>
__global__ __launch_bounds__(256, 2) void issue_probe_kernel(float *out, unsigned long long *cycles, int iters, const float4 *gsrc)
{
    __shared__ float4 lbuf[256];
    lbuf[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    floatx16 acc32[KIND == 0 ? 8 : 1];
    floatx4 acc16[KIND >= 1 ? 16 : 1];
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ha[i] = (__bf16)(0.25f + 0.001f * (threadIdx.x & 31) + 0.01f * i);
        hb[i] = (__bf16)(1.0f - 0.002f * (threadIdx.x & 15) - 0.01f * i);
    }
#pragma unroll
    for (int i = 0; i < (KIND == 0 ? 8 : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < (KIND >= 1 ? 16 : 1); ++i) acc16[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a0 = 0.37f + 0.001f * threadIdx.x, b0 = 1.0f - 0.002f * threadIdx.x;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.01f * (i + 1) + threadIdx.x;
    floatx4 ld[4] = {};
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    floatx2 pk[4], pkb = {0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = floatx2{0.1f * i, 1.0f + threadIdx.x};
    unsigned sacc = 0;
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const float4 *gp = gsrc + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (KIND == 0) acc32[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc32[m & 7], 0, 0, 0);
            else if (KIND == 1) acc16[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc16[m], 0, 0, 0);
            else acc16[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc16[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int q = (m * K + k) & 7;
                if (FT == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[q]) : "v"(b0));
                else if (FT == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[q & 3]) : "v"(laddr) : "memory");
                else if (FT == 2) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
                else if (FT == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b0));
                else if (FT == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[q & 3]) : "v"(gp) : "memory");
                else if (FT == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[q & 3]) : "v"(pkb));
                else if (FT == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x[q]));
                else if (FT == 7) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(ld[q & 3]) : "memory");
                else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[q]) : "v"(b0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FT == 1 || FT == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (FT == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a0 = a0 * 0.999f + 0.0007f;
        b0 = b0 * 1.0001f - 0.0001f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float ssum = (float)sacc;
#pragma unroll
    for (int i = 0; i < 8; ++i) ssum += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) ssum += ld[i][0] + ld[i][3] + pk[i][0] + pk[i][1];
#pragma unroll
    for (int i = 0; i < (KIND == 0 ? 8 : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ssum += acc32[i][r];
#pragma unroll
    for (int i = 0; i < (KIND >= 1 ? 16 : 1); ++i) ssum += acc16[i][0] + acc16[i][3];
    if (ssum == 1234.5678f) out[blockIdx.x * blockDim.x + threadIdx.x] = ssum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- operand probe: does the fp32 MFMA rate depend on WHICH registers feed it?  MODE 0: every MFMA of a round reads the same A / B
// register; 1: 16 distinct A and 16 distinct B registers (one pair per MFMA); 2: the kernel's pattern — A = component e of four
// float4 "weight" registers, B = component e of eight float4 "operand" registers (order frequency, k-step, block); 3: as 2 with
// the operand registers rewritten by ds_read_b128 every round (values identical) and waited with s_waitcnt lgkmcnt(0) once per round.
template <int MODE>
__global__ __launch_bounds__(256, 2) void operand_probe_kernel(float *out, unsigned long long *cycles, int iters)
{
    __shared__ float4 lbuf[512];
    lbuf[threadIdx.x] = make_float4(0.001f * threadIdx.x, 0.5f, 0.25f, 0.125f);
    lbuf[256 + threadIdx.x] = make_float4(0.002f * threadIdx.x, 0.3f, 0.2f, 0.1f);
    __syncthreads();
    floatx4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = 0.37f + 0.001f * threadIdx.x + 0.01f * i;
        b[i] = 1.0f - 0.002f * threadIdx.x - 0.02f * i;
    }
    floatx4 W4[4], V4[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) W4[i] = floatx4{a[i], a[i + 4], a[i + 8], a[i + 12]};
#pragma unroll
    for (int i = 0; i < 8; ++i) V4[i] = floatx4{b[i], b[i + 8], b[(i + 3) & 15], b[(i + 5) & 15]};
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(V4[i]) : "v"(laddr), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float av, bv;
            if (MODE == 0) { av = a[0]; bv = b[0]; }
            else if (MODE == 1) { av = a[m]; bv = b[m]; }
            else {
                const int fl = m >> 3, e = (m >> 1) & 3, blk = m & 1;          // (frequency pair, k-step, block)
                av = W4[fl][e];
                bv = V4[fl * 2 + blk][e];
            }
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float ssum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ssum += acc[i][0] + acc[i][3];
    if (ssum == 1234.5678f) out[blockIdx.x * blockDim.x + threadIdx.x] = ssum;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int FT>
int issue_probe_launch(int K, int blocks, float *out, unsigned long long *cycles, int iters, const float4 *gsrc, hipStream_t s)
{
#define IP_CASE(k) case k: hipLaunchKernelGGL((issue_probe_kernel<KIND, FT, k>), dim3(blocks), dim3(256), 0, s, out, cycles, iters, gsrc); return 0;
    switch (K) {
        IP_CASE(0) IP_CASE(1) IP_CASE(2) IP_CASE(3) IP_CASE(4) IP_CASE(6) IP_CASE(8) IP_CASE(12)
    default: return 1;
    }
#undef IP_CASE
}
}  // namespace

extern "C" int read_debug_operand_probe(int mode, int blocks, int iters, float *scratch, unsigned long long *cycles, void *stream)
{
    READ_CHECK_ARG(blocks > 0 && iters > 0 && scratch && cycles && mode >= 0 && mode <= 3, "read_debug_operand_probe: bad arguments");
    hipStream_t s = as_stream(stream);
    if (mode == 0) hipLaunchKernelGGL(operand_probe_kernel<0>, dim3(blocks), dim3(256), 0, s, scratch, cycles, iters);
    else if (mode == 1) hipLaunchKernelGGL(operand_probe_kernel<1>, dim3(blocks), dim3(256), 0, s, scratch, cycles, iters);
    else if (mode == 2) hipLaunchKernelGGL(operand_probe_kernel<2>, dim3(blocks), dim3(256), 0, s, scratch, cycles, iters);
    else hipLaunchKernelGGL(operand_probe_kernel<3>, dim3(blocks), dim3(256), 0, s, scratch, cycles, iters);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// kind 0 / 1 / 2 = v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_bf16; filler type 0..4 (see issue_probe_kernel); K fillers per MFMA in
// {0,1,2,3,4,6,8,12}; `blocks` workgroups of 4 waves; cycles[blocks * 4] receives every wave's s_memtime span; gsrc: 1 KiB.
extern "C" int read_debug_issue_probe(int kind, int filler, int K, int blocks, int iters, float *scratch, unsigned long long *cycles,
                                      const float *gsrc, void *stream)
{
    READ_CHECK_ARG(blocks > 0 && iters > 0 && scratch && cycles && gsrc, "read_debug_issue_probe: bad arguments");
    const float4 *g = reinterpret_cast<const float4 *>(gsrc);
    hipStream_t s = as_stream(stream);
    int rc = 1;
    if (kind == 0) {
        if (filler == 0) rc = issue_probe_launch<0, 0>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 1) rc = issue_probe_launch<0, 1>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 2) rc = issue_probe_launch<0, 2>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 3) rc = issue_probe_launch<0, 3>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 4) rc = issue_probe_launch<0, 4>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 5) rc = issue_probe_launch<0, 5>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 6) rc = issue_probe_launch<0, 6>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 7) rc = issue_probe_launch<0, 7>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 8) rc = issue_probe_launch<0, 8>(K, blocks, scratch, cycles, iters, g, s);
    } else if (kind == 1) {
        if (filler == 0) rc = issue_probe_launch<1, 0>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 1) rc = issue_probe_launch<1, 1>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 2) rc = issue_probe_launch<1, 2>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 3) rc = issue_probe_launch<1, 3>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 4) rc = issue_probe_launch<1, 4>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 5) rc = issue_probe_launch<1, 5>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 6) rc = issue_probe_launch<1, 6>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 7) rc = issue_probe_launch<1, 7>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 8) rc = issue_probe_launch<1, 8>(K, blocks, scratch, cycles, iters, g, s);
    }
    else if (kind == 2) {
        if (filler == 0) rc = issue_probe_launch<2, 0>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 1) rc = issue_probe_launch<2, 1>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 2) rc = issue_probe_launch<2, 2>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 4) rc = issue_probe_launch<2, 4>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 5) rc = issue_probe_launch<2, 5>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 7) rc = issue_probe_launch<2, 7>(K, blocks, scratch, cycles, iters, g, s);
        else if (filler == 8) rc = issue_probe_launch<2, 8>(K, blocks, scratch, cycles, iters, g, s);
    }
    if (rc) { set_error("read_debug_issue_probe: unsupported kind / filler / K"); return READ_EINVAL; }
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// Launches `blocks` workgroups of 4 waves, each wave issuing iters*nacc*4 MFMAs (4096 FLOP each).
extern "C" int read_debug_mfma_probe(int blocks, int iters, int nacc, float *scratch, void *stream)
{
    READ_CHECK_ARG(blocks > 0 && iters > 0 && scratch, "read_debug_mfma_probe: bad arguments");
    const float seed = nacc < 0 ? -1.0f : 0.37f;      // nacc < 0: random per-lane operands
    nacc = nacc < 0 ? -nacc : nacc;
    if (nacc == 4) hipLaunchKernelGGL(mfma_f32_probe_kernel<4>, dim3(blocks), dim3(256), 0, as_stream(stream), scratch, iters, seed);
    else if (nacc == 2) hipLaunchKernelGGL(mfma_f32_probe_kernel<2>, dim3(blocks), dim3(256), 0, as_stream(stream), scratch, iters, seed);
    else { set_error("read_debug_mfma_probe: nacc must be 2 or 4"); return READ_EINVAL; }
    READ_CHECK_LAUNCH();
    return READ_OK;
}


namespace {
// ---- fp32 VALU rate probe: what does ONE vector instruction of each form cost a wave?  Eight independent accumulators, 64
// instructions per round, nothing else in the loop.  MODE 0 v_fma_f32 (registers) · 1 v_fmac_f32 with an SGPR multiplier · 2
// v_pk_fma_f32 (register pairs, default selects) · 3 v_pk_fma_f32 with an SGPR pair and a broadcast op_sel (the first form of the
// small-Cout kernel) · 4 v_pk_add_f32 · 5 v_pk_mul_f32 · 6 v_pk_fma_f32 with the half-selects of the F(4x4) input transform ·
// 7 v_add_f32.
typedef float floatx2p __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void valu_probe_kernel(float *out, unsigned long long *cycles, int iters, float sw0, float sw1)
{
    floatx2p acc[8], xv = {0.001f * threadIdx.x, 1.0f - 0.002f * threadIdx.x}, wv = {0.999f, 1.001f};
    const floatx2p ws = {sw0, sw1};                                   // kernel arguments: wave-uniform, live in SGPRs
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = floatx2p{0.01f * i, 0.02f * i + threadIdx.x};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            floatx2p &A = acc[m & 7];
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A.x) : "v"(xv.x), "v"(wv.x));
            else if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(A.x) : "s"(sw0), "v"(xv.x));
            else if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(xv), "v"(wv));
            else if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(A) : "v"(xv), "s"(ws));
            else if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(A) : "v"(wv));
            else if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(A) : "v"(wv));
            else if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(A) : "v"(xv), "v"(wv));
            else asm volatile("v_add_f32 %0, %0, %1" : "+v"(A.x) : "v"(wv.x));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float ssum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ssum += acc[i].x + acc[i].y;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (ssum == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = ssum;
}
}  // namespace

extern "C" int read_debug_valu_probe(int mode, int blocks, int iters, float *scratch, unsigned long long *cycles, void *stream)
{
    READ_CHECK_ARG(blocks > 0 && iters > 0 && scratch && cycles && mode >= 0 && mode <= 7, "read_debug_valu_probe: bad arguments");
    hipStream_t s = as_stream(stream);
#define VP_CASE(m) case m: hipLaunchKernelGGL(valu_probe_kernel<m>, dim3(blocks), dim3(256), 0, s, scratch, cycles, iters, 0.9995f, 1.0005f); break;
    switch (mode) { VP_CASE(0) VP_CASE(1) VP_CASE(2) VP_CASE(3) VP_CASE(4) VP_CASE(5) VP_CASE(6) VP_CASE(7) }
#undef VP_CASE
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// What a kernel boundary costs against a grid-wide barrier inside one persistent kernel (the rasteriser's five dependent
// launches: are they worth merging?).  Every "phase" touches `bytes_per_block` of its workgroup's slice of a buffer (read-modify-
// write, so a phase depends on the previous one through memory).
//   mode 0: `phases` dependent launches of `blocks` workgroups.
//   mode 1: ONE launch; the phases are separated by grid barriers (release fence, one atomic add per workgroup on a counter, spin
//           on an agent-scope load, acquire fence).  `blocks` must not exceed what the device keeps resident.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void chain_phase(float *buf, int floats_per_block, int phase)
{
    float *p = buf + (size_t)blockIdx.x * floats_per_block;
    for (int i = threadIdx.x; i < floats_per_block; i += blockDim.x) p[i] = p[i] * 1.0001f + (float)phase;
}

__global__ __launch_bounds__(256) void chain_launch_kernel(float *buf, int floats_per_block, int phase)
{
    chain_phase(buf, floats_per_block, phase);
}

__global__ __launch_bounds__(256) void chain_barrier_kernel(float *buf, int floats_per_block, int phases, unsigned *counter, unsigned *timeout_flag)
{
    for (int ph = 0; ph < phases; ++ph) {
        chain_phase(buf, floats_per_block, ph);
        if (ph + 1 == phases) break;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                                   // release: this workgroup's stores
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(ph + 1) * gridDim.x;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {                                    // ~seconds: never hang the box
                    *timeout_flag = 1u;
                    break;
                }
            }
            __threadfence();                                                   // acquire: the other workgroups' stores
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int read_debug_chain_probe(int mode, int blocks, int phases, int floats_per_block, float *buf, unsigned *counter_and_flag,
                                      void *stream)
{
    READ_CHECK_ARG(blocks > 0 && phases > 0 && floats_per_block > 0 && buf && counter_and_flag, "read_debug_chain_probe: bad arguments");
    hipStream_t s = as_stream(stream);
    if (mode == 0) {
        for (int ph = 0; ph < phases; ++ph)
            hipLaunchKernelGGL(chain_launch_kernel, dim3(blocks), dim3(256), 0, s, buf, floats_per_block, ph);
    } else {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        READ_CHECK_HIP(hipGetDevice(&dev));
        READ_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        READ_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_barrier_kernel, 256, 0));
        READ_CHECK_ARG(blocks <= per_cu * prop.multiProcessorCount, "read_debug_chain_probe: %d workgroups are not co-resident (%d x %d)",
                       blocks, per_cu, prop.multiProcessorCount);
        READ_CHECK_HIP(hipMemsetAsync(counter_and_flag, 0, 2 * sizeof(unsigned), s));
        hipLaunchKernelGGL(chain_barrier_kernel, dim3(blocks), dim3(256), 0, s, buf, floats_per_block, phases, counter_and_flag,
                           counter_and_flag + 1);
    }
    READ_CHECK_LAUNCH();
    return READ_OK;
}
