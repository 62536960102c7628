// Round 6 probe: what does the f16 / bf16 matrix path do with SPLIT fp32 operands?
//   hipcc --offload-arch=gfx950 -O2 -o f16split_probe f16split_probe.hip && ./f16split_probe
// One wave computes C[16][16] = sum_k A[16][K] B[K][16] four ways and the host compares with fp64:
//   f32      v_mfma_f32_16x16x4_f32 chain (the product kernel's arithmetic)
//   f16x3    x = xh + 2^-11 xl (both f16, xl scaled), products ah*bh + ah*bl' + al*bh' with the 2^-11 folded into the weight side
//   bf16x6   three bf16 pieces each, six largest pairs
// It also pins the A/B fragment layout of v_mfma_f32_16x16x32_{f16,bf16}: lane l holds A[l & 15][8 (l >> 4) .. + 7].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *A, const float *B, int K, float *Cf32, float *Cf16, float *Cbf, float ascale)
{
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], c0, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        h8 ah, al, ahs, bh, bl;
        b8 a3[3], b3[3];
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + 8 * q + i;
            const float a = A[r * K + k] * ascale, b = B[k * 16 + r];
            const _Float16 a_h = (_Float16)a;
            ah[i] = a_h;
            al[i] = (_Float16)(a - (float)a_h);                       // true low piece of the (pre-scaled) weight
            ahs[i] = (_Float16)((float)a_h * 0x1p-11f);               // high piece times 2^-11 (meets the scaled low piece of b)
            const _Float16 b_h = (_Float16)b;
            bh[i] = b_h;
            bl[i] = (_Float16)((b - (float)b_h) * 2048.0f);
            float ra = A[r * K + k], rb = b;
            for (int p = 0; p < 3; ++p) {
                a3[p][i] = (__bf16)ra; ra -= (float)a3[p][i];
                b3[p][i] = (__bf16)rb; rb -= (float)b3[p][i];
            }
        }
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahs, bl, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c1, 0, 0, 0);
        const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
        for (int p = 0; p < 6; ++p) c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[pa[p]], b3[pb[p]], c2, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) {
        Cf32[(4 * q + i) * 16 + r] = c0[i];
        Cf16[(4 * q + i) * 16 + r] = c1[i] / ascale;
        Cbf[(4 * q + i) * 16 + r] = c2[i];
    }
}

int main()
{
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (float amp : {1.0f, 1e-3f, 1e-6f, 300.0f}) {
        for (int K : {32, 64, 256, 1024}) {
            std::vector<float> A(16 * K), B(K * 16);
            for (auto &v : A) v = nd(rng) * 0.05f;
            for (auto &v : B) v = nd(rng) * 10.0f * amp;
            float amax = 0;
            for (auto v : A) amax = std::fmax(amax, std::fabs(v));
            const float ascale = std::exp2(std::floor(std::log2(16384.0f / amax)));
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 3 * 256 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            probe<<<1, 64>>>(dA, dB, K, dC, dC + 256, dC + 512, ascale);
            std::vector<float> C(3 * 256);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double e[3] = {0, 0, 0}, ref2 = 0, refabs = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 16 + j]; sa += std::fabs((double)A[i * K + k] * B[k * 16 + j]); }
                    ref2 += s * s; refabs += sa;
                    for (int m = 0; m < 3; ++m) { const double d = C[m * 256 + i * 16 + j] - s; e[m] += d * d; }
                }
            printf("amp %-7g K %4d  rel rms error: f32 %.3e  f16x3 %.3e  bf16x6 %.3e   (|sum| rms %.3e, scale 2^%d)\n", amp, K,
                   std::sqrt(e[0] / ref2), std::sqrt(e[1] / ref2), std::sqrt(e[2] / ref2), std::sqrt(ref2 / 256), (int)std::log2(ascale));
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    return 0;
}
