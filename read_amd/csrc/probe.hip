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
}  // namespace

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
