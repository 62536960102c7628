// Training step of READ's gated convolutions on gfx950: what torch.autograd + cuDNN do for
// READ/models/unet.py:22-53 under src/train.py:132-203, plus the loss of src/READ/models/compose.py:29-40 (Huber part)
// and the descriptor optimizer of READ/pipelines/ogl.py:16,99-100 (RMSprop) restricted to the rows a step touched.
//
// One BasicConv, y = BN_eval( act(f) * sigmoid(m) ), f|m = conv_{f|m}(x) + b:
//   forward (training)  read_gated_conv_forward(linear = 1)  -> pre-activations [f | m] (kept for the backward pass)
//                       gate_forward_kernel                   -> y
//   backward            gate_backward_kernel   dy, [f|m] -> d[f|m] (channel-padded) + per-channel sums for db_f, db_m, dgamma, dbeta
//                       dgrad                  d[f|m] -> dx: for stride-1 layers the SAME MFMA convolution kernel with flipped,
//                                              transposed weights (pack_dgrad_weights_kernel + linear = 1); stride 2: dgrad_generic_kernel
//                       wgrad_mfma_kernel      x, d[f|m] -> dW_f, dW_m on the matrix cores (pixels are the reduction dimension)
// BatchNorm is the eval-mode affine map (running statistics frozen, gamma/beta trained) — the configuration the reference
// trains with (configs/train_example.yaml: eval_in_train: True, train.py:271-277).
// Weights change every optimizer step, so the MFMA fragment orders are produced on the device (pack_*_kernel).
#include "common.h"

using namespace readhip;

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

int grid_for(long long items, int per_block = 256, int cap = 256 * 16)
{
    long long b = (items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---------------------------------------------------------------------------------------------------------------------
// parameter / weight packing on the device
// ---------------------------------------------------------------------------------------------------------------------
// params block of the conv kernels: 4 x CoutPad = bias_f, bias_m, bn_scale, bn_shift (read_conv_pack_params_host)
__device__ __forceinline__ void pack_params_body(int c, int Cout, int CoutPad, const float *bf, const float *bm, const float *gamma,
                                                 const float *beta, const float *mean, const float *var, float eps, float *out)
{
    if (c >= CoutPad) return;
    const bool ok = c < Cout;
    const float sc = ok ? gamma[c] / sqrtf(var[c] + eps) : 0.0f;
    out[c] = ok && bf ? bf[c] : 0.0f;
    out[CoutPad + c] = ok && bm ? bm[c] : 0.0f;
    out[2 * CoutPad + c] = sc;
    out[3 * CoutPad + c] = ok ? beta[c] - mean[c] * sc : 0.0f;
}
__global__ void pack_params_kernel(int Cout, int CoutPad, const float *bf, const float *bm, const float *gamma,
                                   const float *beta, const float *mean, const float *var, float eps, float *out)
{
    pack_params_body(blockIdx.x * blockDim.x + threadIdx.x, Cout, CoutPad, bf, bm, gamma, beta, mean, var, eps, out);
}

// Direct-kernel fragment order of read_conv_pack_weights_host: [chunk][tap][k8][tile(f,m)][lane][4].
// mode 0: the layer's own weights  W{f|m}[cout][cin][tap]                                    (forward)
// mode 1: dgrad as a convolution over d[f|m] (2*Cp channels: df then dm, Cp = padded Cout of the layer):
//         virtual conv with Cin_v = 2*Cp input channels and Cout_v = Cin/2 gated output channels per half;
//         W_v{half}[co_v][ci_v][tap] = W{ci_v < Cp ? f : m}[ci_v % Cp][half * Cin/2 + co_v][k*k - 1 - tap]   (flipped, transposed)
__device__ __forceinline__ void pack_weights_body(int bid, int nblk, int mode, int Cin, int Cout, int ksize, int kc, int Cp, const float *wf,
                                                  const float *wm, float *out, long long total)
{
    const int taps = ksize * ksize;
    const int CinV = mode ? 2 * Cp : Cin, CoutV = mode ? Cin / 2 : Cout;
    const int CoutPad = (CoutV + 31) / 32 * 32, NT = CoutPad / 16, KK = kc / 8;
    for (long long o = (long long)bid * blockDim.x + threadIdx.x; o < total; o += (long long)nblk * blockDim.x) {
        long long r = o;
        const int j = (int)(r & 3); r >>= 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int nt = (int)(r % NT); r /= NT;
        const int kk = (int)(r % KK); r /= KK;
        const int tap = (int)(r % taps);
        const int chunk = (int)(r / taps);
        const int cout = (nt >> 1) * 32 + (lane & 31);
        const int cin = chunk * kc + kk * 8 + 4 * (lane >> 5) + j;
        float v = 0.0f;
        if (cout < CoutV && cin < CinV) {
            if (!mode) {
                v = ((nt & 1) ? wm : wf)[((size_t)cout * Cin + cin) * taps + tap];
            } else {
                const int layer_co = cin % Cp, layer_ci = (nt & 1) * (Cin / 2) + cout;
                if (layer_co < Cout) v = (cin < Cp ? wf : wm)[((size_t)layer_co * Cin + layer_ci) * taps + (taps - 1 - tap)];
            }
        }
        out[o] = v;
    }
}
__global__ void pack_weights_kernel(int mode, int Cin, int Cout, int ksize, int kc, int Cp, const float *wf, const float *wm,
                                    float *out, long long total)
{
    pack_weights_body((int)blockIdx.x, (int)gridDim.x, mode, Cin, Cout, ksize, kc, Cp, wf, wm, out, total);
}

// The 3x3 kernel of the (virtual) layer's (cout, cin) pair, or null where the pair does not exist.  mode 0: the layer's own weights;
// mode 1: dgrad as a convolution over d[f|m] (pack_weights_kernel): flipped, transposed.
__device__ __forceinline__ const float *w3x3_of(int mode, int Cin, int Cout, int Cp, const float *wf, const float *wm, int fm, int co,
                                                int ci, int CoutV)
{
    if (co >= CoutV) return nullptr;
    if (!mode) return (fm ? wm : wf) + ((size_t)co * Cin + ci) * 9;
    const int layer_co = ci % Cp, layer_ci = fm * (Cin / 2) + co;
    return layer_co < Cout ? (ci < Cp ? wf : wm) + ((size_t)layer_co * Cin + layer_ci) * 9 : nullptr;
}

// Winograd F(2x2,3x3) fragment order of read_conv_pack_wino_host: U = G g G^T per (cout, cin) pair, packed
// [group][k8 step][row i][j][f|m][lane][4], row 2 negated.  One workgroup per (group, k8 step): thread (lane, e) forms the 4 x 4
// transform of its pair ONCE (constant indices: the first version looked G up per output element and spent 114 us per layer in
// scratch traffic) and writes its 2 x 16 outputs, 1 KiB per instruction and workgroup.
__device__ __forceinline__ void pack_wino_body(int bid, int mode, int Cin, int Cout, int Cp, const float *wf, const float *wm, float *out)
{
    const int CinV = mode ? 2 * Cp : Cin, CoutV = mode ? Cin / 2 : Cout;
    const int nsteps = CinV / 8;
    const int st = bid % nsteps, g = bid / nsteps;
    const int lane = threadIdx.x >> 2, e = threadIdx.x & 3;
    const int co = g * 32 + (lane & 31), ci = 8 * st + 4 * (lane >> 5) + e;
    float *o = out + (size_t)bid * (16 * 2 * 256) + threadIdx.x;
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const float *k = w3x3_of(mode, Cin, Cout, Cp, wf, wm, fm, co, ci, CoutV);
        float w[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = k ? k[mode ? 8 - t : t] : 0.0f;
        float h[4][3];                                         // G g: rows g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) {
            h[0][b2] = w[0][b2];
            h[1][b2] = 0.5f * w[0][b2] + 0.5f * w[1][b2] + 0.5f * w[2][b2];
            h[2][b2] = 0.5f * w[0][b2] - 0.5f * w[1][b2] + 0.5f * w[2][b2];
            h[3][b2] = w[2][b2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u[4] = {h[i][0], 0.5f * h[i][0] + 0.5f * h[i][1] + 0.5f * h[i][2], 0.5f * h[i][0] - 0.5f * h[i][1] + 0.5f * h[i][2],
                                h[i][2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) o[((i * 4 + j) * 2 + fm) * 256] = i == 2 ? -u[j] : u[j];
        }
    }
}
__global__ __launch_bounds__(256) void pack_wino_kernel(int mode, int Cin, int Cout, int Cp, const float *wf, const float *wm, float *out)
{
    pack_wino_body((int)blockIdx.x, mode, Cin, Cout, Cp, wf, wm, out);
}

// Winograd F(4x4,3x3) fragment order of read_conv_pack_w4_host: U = G g G^T (6 x 6, evaluated in double, rounded once) per
// (cout, cin) pair, [group][wave 4][chunk of 16 cin][frequency 6 xi + nu][lane][e]; lane (i = lane & 15, kl = lane >> 4) =
// U_{i < 8 ? f : m}[xi][nu][cin = 16 chunk + 4 kl + e][cout = 32 group + 8 wave + (i & 7)].  One workgroup per (group, wave, chunk):
// thread (lane, e) forms the 36 values of its pair and writes them 1 KiB per instruction and workgroup.
__device__ __forceinline__ void pack_w4_body(int bid, int mode, int Cin, int Cout, int Cp, const float *wf, const float *wm, float *out)
{
    const int CinV = mode ? 2 * Cp : Cin, CoutV = mode ? Cin / 2 : Cout;
    const int nchunks = CinV / 16;
    const int chunk = bid % nchunks, w = (bid / nchunks) & 3, g = bid / (4 * nchunks);
    const int lane = threadIdx.x >> 2, e = threadIdx.x & 3;
    const int slot = lane & 15, fm = slot >> 3;
    const int co = g * 32 + w * 8 + (slot & 7), ci = 16 * chunk + 4 * (lane >> 4) + e;
    const float *k = w3x3_of(mode, Cin, Cout, Cp, wf, wm, fm, co, ci, CoutV);
    double wk[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t / 3][t % 3] = k ? (double)k[mode ? 8 - t : t] : 0.0;
    // G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], the same sums in the same order as the host packer
    auto gmul = [](double x0, double x1, double x2, double (&r)[6]) {
        r[0] = 0.25 * x0 + 0.0 * x1 + 0.0 * x2;
        r[1] = (-1.0 / 6) * x0 + (-1.0 / 6) * x1 + (-1.0 / 6) * x2;
        r[2] = (-1.0 / 6) * x0 + (1.0 / 6) * x1 + (-1.0 / 6) * x2;
        r[3] = (1.0 / 24) * x0 + (1.0 / 12) * x1 + (1.0 / 6) * x2;
        r[4] = (1.0 / 24) * x0 + (-1.0 / 12) * x1 + (1.0 / 6) * x2;
        r[5] = 0.0 * x0 + 0.0 * x1 + 1.0 * x2;
    };
    double h[3][6];                                            // h[b][xi] = sum_a G[xi][a] g[a][b]
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) gmul(wk[0][b2], wk[1][b2], wk[2][b2], h[b2]);
    float *o = out + (size_t)bid * (36 * 256) + threadIdx.x;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
        double u[6];
        gmul(h[0][xi], h[1][xi], h[2][xi], u);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) o[(xi * 6 + nu) * 256] = (float)u[nu];
    }
}
__global__ __launch_bounds__(256) void pack_w4_kernel(int mode, int Cin, int Cout, int Cp, const float *wf, const float *wm, float *out)
{
    pack_w4_body((int)blockIdx.x, mode, Cin, Cout, Cp, wf, wm, out);
}

// Every packing job of a training step in ONE launch (read_conv_pack_batch): a step re-packs the parameter block, the forward
// fragments and the dgrad fragments of all 99 layers because the optimizer has just changed them — 297 launches of 2 .. 5 us
// kernels, each with its own allocation and event on the host side and ~8 us of launch gap on the device side.  A workgroup finds
// its job in the table (first_block is the running sum of the jobs' block counts) and runs that job's body.
__global__ __launch_bounds__(256) void pack_batch_kernel(const read_pack_job *__restrict__ jobs, int njobs)
{
    __shared__ int s_job;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;                            // last job with first_block <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        s_job = lo;
    }
    __syncthreads();
    const read_pack_job j = jobs[s_job];
    const int bid = (int)blockIdx.x - j.first_block;
    if (bid >= j.nblocks) return;
    switch (j.kind) {
    case READ_PACK_PARAMS:
        pack_params_body(bid * 256 + (int)threadIdx.x, j.Cout, (j.Cout + 31) / 32 * 32, j.bf, j.bm, j.gamma, j.beta, j.mean, j.var, j.eps,
                         j.out);
        break;
    case READ_PACK_DIRECT:
        pack_weights_body(bid, j.nblocks, j.mode, j.Cin, j.Cout, j.ksize, j.kc, j.Cp, j.wf, j.wm, j.out, j.total);
        break;
    case READ_PACK_WINO:
        pack_wino_body(bid, j.mode, j.Cin, j.Cout, j.Cp, j.wf, j.wm, j.out);
        break;
    case READ_PACK_W4:
        pack_w4_body(bid, j.mode, j.Cin, j.Cout, j.Cp, j.wf, j.wm, j.out);
        break;
    default:
        break;
    }
}

// weights for dgrad_generic_kernel: [tap][co' in 2*Cp][ci]  (ci contiguous)
__global__ void pack_dgrad_generic_kernel(int Cin, int Cout, int taps, int Cp, const float *wf, const float *wm, float *out,
                                          long long total)
{
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(o % Cin);
        const int cp = (int)((o / Cin) % (2 * Cp));
        const int tap = (int)(o / ((long long)Cin * 2 * Cp));
        const int co = cp % Cp;
        out[o] = co < Cout ? (cp < Cp ? wf : wm)[((size_t)co * Cin + ci) * taps + tap] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gate forward / backward (elementwise over pixels x channels)
// ---------------------------------------------------------------------------------------------------------------------
// fm: [pixels][2*Cout] = f (bias included) | m;  y = (act(f) * sigmoid(m)) * scale + shift (+ residual)
// Batches are stacked vertically into one tall image, `block_h` rows per item of which the first `valid_h` are the item and
// the rest a separator that must stay ZERO in every activation (it is the zero padding between neighbours): rows with
// (row % block_h) >= valid_h are written as 0 here and get zero gradient in gate_backward_kernel.  block_h = 0: no blocks.
__device__ __forceinline__ bool separator_row(long long pixel, int W, int block_h, int valid_h)
{
    return block_h > 0 && (int)((pixel / W) % block_h) >= valid_h;
}

__global__ __launch_bounds__(256) void gate_forward_kernel(const float *__restrict__ fm, long long pixels, int Cout, int CoutPad,
                                                           const float *__restrict__ params, int elu,
                                                           const float *__restrict__ residual, float *__restrict__ y, int W,
                                                           int block_h, int valid_h)
{
    const long long total = pixels * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / Cout;
        const int c = (int)(i - p * Cout);
        if (separator_row(p, W, block_h, valid_h)) {
            y[i] = 0.0f;
            continue;
        }
        float f = fm[p * 2 * Cout + c];
        const float m = fm[p * 2 * Cout + Cout + c];
        if (elu) f = f > 0.0f ? f : fast_exp(f) - 1.0f;
        const float s = __builtin_amdgcn_rcpf(1.0f + fast_exp(-m));
        float v = (f * s) * params[2 * CoutPad + c] + params[3 * CoutPad + c];
        if (residual) v += residual[i];
        y[i] = v;
    }
}

// dy [pixels][Cout], fm [pixels][2*Cout]  ->  dfm [pixels][2*Cp] (df | dm, channels >= Cout zero) and
// sums[4][Cout] += { sum df, sum dm, sum dy, sum dy * g }  with g = act(f) * sigmoid(m)   (db_f, db_m, dbeta, dgamma pieces)
// One workgroup = 64 pixels x all channels (thread t: channel t % CW ... ), block-level partial sums in LDS, one
// atomicAdd per channel and workgroup.
// mode 0: BatchNorm as the eval-mode affine map, dg = dy * scale.
// Batch-statistics BatchNorm (nn.BatchNorm2d in .train(), unet.py:40,51) needs the per-channel sums of dy and dy * g BEFORE any
// dg can be formed (dg = gamma r (dy - mean(dy) - xhat mean(dy xhat))), so its backward is two passes of this kernel:
// mode 1: only the sums {., ., sum dy, sum dy * g} (no dfm written);  mode 2: dg = A dy + B + C g with the per-channel constants
// abc[3][Cout] that bn_bwd_coeff_kernel derives from the mode-1 sums.
// Statistic groups (gridDim.y > 1, modes 1 and 2): one group per item of the stacked batch — the reference runs its net once per
// batch item (READ/models/compose.py:137-176), so nn.BatchNorm2d sees N = 1 there; group j covers the rows of block j and uses
// sums + j * 4 * Cout, abc + j * 3 * Cout.
template <int CW>
__global__ __launch_bounds__(256) void gate_backward_kernel(const float *__restrict__ dy, const float *__restrict__ fm,
                                                            long long pixels, int Cout, int CoutPad, int Cp,
                                                            const float *__restrict__ params, int elu,
                                                            float *__restrict__ dfm, float *__restrict__ sums, int W, int block_h,
                                                            int valid_h, int mode, const float *__restrict__ abc)
{
    constexpr int ROWS = 256 / CW;                       // pixels handled concurrently by a workgroup
    __shared__ float red[4][256];
    const int c0 = threadIdx.x % CW, r = threadIdx.x / CW;
    long long p_begin = 0, p_end = pixels;
    if (gridDim.y > 1) {
        p_begin = (long long)blockIdx.y * block_h * W;
        p_end = p_begin + (long long)block_h * W < pixels ? p_begin + (long long)block_h * W : pixels;
        sums += (size_t)blockIdx.y * 4 * Cout;
        if (abc) abc += (size_t)blockIdx.y * 3 * Cout;
    }
    for (int cb = 0; cb < Cp; cb += CW) {                // channel blocks of CW
        const int c = cb + c0;
        const bool ok = c < Cout;
        const float sc = ok ? params[2 * CoutPad + c] : 0.0f;
        const float cA = (ok && mode == 2) ? abc[c] : 0.0f, cB = (ok && mode == 2) ? abc[Cout + c] : 0.0f,
                    cC = (ok && mode == 2) ? abc[2 * Cout + c] : 0.0f;
        float s_df = 0.f, s_dm = 0.f, s_dy = 0.f, s_dyg = 0.f;
        for (long long p = p_begin + (long long)blockIdx.x * ROWS + r; p < p_end; p += (long long)gridDim.x * ROWS) {
            float df = 0.f, dm = 0.f;
            if (ok && !separator_row(p, W, block_h, valid_h)) {
                const float f = fm[p * 2 * Cout + c], m = fm[p * 2 * Cout + Cout + c];
                const float g = dy[p * Cout + c];
                const float a = elu ? (f > 0.0f ? f : fast_exp(f) - 1.0f) : f;
                const float da = elu ? (f > 0.0f ? 1.0f : a + 1.0f) : 1.0f;          // ELU'(f) = exp(f) = a + 1 for f <= 0
                const float s = __builtin_amdgcn_rcpf(1.0f + fast_exp(-m));
                const float gs = mode == 2 ? cA * g + cB + cC * (a * s) : g * sc;   // through the BatchNorm (eval: its scale)
                df = gs * s * da;
                dm = gs * a * s * (1.0f - s);
                s_df += df;
                s_dm += dm;
                s_dy += g;
                s_dyg += g * (a * s);
            }
            if (c < Cp && mode != 1) {
                dfm[p * 2 * Cp + c] = df;
                dfm[p * 2 * Cp + Cp + c] = dm;
            }
        }
        red[0][threadIdx.x] = s_df;
        red[1][threadIdx.x] = s_dm;
        red[2][threadIdx.x] = s_dy;
        red[3][threadIdx.x] = s_dyg;
        __syncthreads();
        if (threadIdx.x < CW && ok) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < ROWS; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] += red[q][k * CW + threadIdx.x];
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(sums + q * Cout + c, t[q]);
        }
        __syncthreads();
    }
}

// The same for layers whose channel count is a multiple of 32 (every C -> C layer of the UNet): thread = (pixel, FOUR channels), all
// channels of a pixel in one pass, 128-bit loads and stores (the scalar form above moves 256 bytes per wave instruction and walks
// the tensor once per 32-channel block: 92 us per layer against the 355 MB it touches).
__global__ __launch_bounds__(256) void gate_backward4_kernel(const float *__restrict__ dy, const float *__restrict__ fm,
                                                             long long pixels, int Cout, int CoutPad,
                                                             const float *__restrict__ params, int elu,
                                                             float *__restrict__ dfm, float *__restrict__ sums, int W, int block_h,
                                                             int valid_h, int mode, const float *__restrict__ abc)
{
    __shared__ float red[16][256];
    const int QW = Cout >> 2, ROWS = 256 / QW;            // QW = 8, 16, 32, 64: float4 quads per pixel; pixels per workgroup pass
    const int q = threadIdx.x % QW, r = threadIdx.x / QW, c = 4 * q;
    long long p_begin = 0, p_end = pixels;
    if (gridDim.y > 1) {
        p_begin = (long long)blockIdx.y * block_h * W;
        p_end = p_begin + (long long)block_h * W < pixels ? p_begin + (long long)block_h * W : pixels;
        sums += (size_t)blockIdx.y * 4 * Cout;
        if (abc) abc += (size_t)blockIdx.y * 3 * Cout;
    }
    const float4 sc = *reinterpret_cast<const float4 *>(params + 2 * CoutPad + c);
    float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA, cC = cA;
    if (mode == 2) {
        cA = *reinterpret_cast<const float4 *>(abc + c);
        cB = *reinterpret_cast<const float4 *>(abc + Cout + c);
        cC = *reinterpret_cast<const float4 *>(abc + 2 * Cout + c);
    }
    float acc[4][4];                                       // [sum kind][channel of the quad]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    for (long long p = p_begin + (long long)blockIdx.x * ROWS + r; p < p_end; p += (long long)gridDim.x * ROWS) {
        float df[4] = {0.f, 0.f, 0.f, 0.f}, dm[4] = {0.f, 0.f, 0.f, 0.f};
        if (!separator_row(p, W, block_h, valid_h)) {
            const float4 f4 = *reinterpret_cast<const float4 *>(fm + p * 2 * Cout + c);
            const float4 m4 = *reinterpret_cast<const float4 *>(fm + p * 2 * Cout + Cout + c);
            const float4 g4 = *reinterpret_cast<const float4 *>(dy + p * Cout + c);
            const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, av[4] = {cA.x, cA.y, cA.z, cA.w}, bv[4] = {cB.x, cB.y, cB.z, cB.w},
                        cv[4] = {cC.x, cC.y, cC.z, cC.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f = fv[j], g = gv[j];
                const float a = elu ? (f > 0.0f ? f : fast_exp(f) - 1.0f) : f;
                const float da = elu ? (f > 0.0f ? 1.0f : a + 1.0f) : 1.0f;
                const float sg = __builtin_amdgcn_rcpf(1.0f + fast_exp(-mv[j]));
                const float gs = mode == 2 ? av[j] * g + bv[j] + cv[j] * (a * sg) : g * scv[j];
                df[j] = gs * sg * da;
                dm[j] = gs * a * sg * (1.0f - sg);
                acc[0][j] += df[j];
                acc[1][j] += dm[j];
                acc[2][j] += g;
                acc[3][j] += g * (a * sg);
            }
        }
        if (mode != 1) {
            *reinterpret_cast<float4 *>(dfm + p * 2 * Cout + c) = make_float4(df[0], df[1], df[2], df[3]);
            *reinterpret_cast<float4 *>(dfm + p * 2 * Cout + Cout + c) = make_float4(dm[0], dm[1], dm[2], dm[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[k * 4 + j][threadIdx.x] = acc[k][j];
    __syncthreads();
    if ((int)threadIdx.x < Cout) {                         // one thread per channel: its quad's column over the ROWS pixel slots
        const int ch = threadIdx.x, qq = ch >> 2, j = ch & 3;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ROWS; ++k)
#pragma unroll
            for (int kind = 0; kind < 4; ++kind) t[kind] += red[kind * 4 + j][k * QW + qq];
#pragma unroll
        for (int kind = 0; kind < 4; ++kind) atomicAdd(sums + kind * Cout + ch, t[kind]);
    }
}

// dbf = S0, dbm = S1, dbeta = S2, dgamma = (S3 - mean * S2) / sqrt(var + eps)      (y = g * gamma * r + beta - mean * gamma * r)
__global__ void bn_grads_kernel(int Cout, const float *sums, const float *mean, const float *var, float eps, float *dbf,
                                float *dbm, float *dgamma, float *dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cout) return;
    const float rstd = 1.0f / sqrtf(var[c] + eps);
    if (dbf) dbf[c] += sums[c];
    if (dbm) dbm[c] += sums[Cout + c];
    if (dbeta) dbeta[c] += sums[2 * Cout + c];
    if (dgamma) dgamma[c] += (sums[3 * Cout + c] - mean[c] * sums[2 * Cout + c]) * rstd;
}

// The same over statistic groups (one per batch item): sums[groups][4][Cout], stat[groups][2][Cout] = {mean, biased var} of each
// group; every parameter gradient is the sum of the groups' contributions.
__global__ void bn_grads_groups_kernel(int Cout, int groups, const float *sums, const float *stat, float eps, float *dbf, float *dbm,
                                       float *dgamma, float *dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cout) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < groups; ++j) {
        const float *S = sums + (size_t)j * 4 * Cout, *st = stat + (size_t)j * 2 * Cout;
        const float rstd = 1.0f / sqrtf(st[Cout + c] + eps);
        a0 += S[c];
        a1 += S[Cout + c];
        a2 += S[2 * Cout + c];
        a3 += (S[3 * Cout + c] - st[c] * S[2 * Cout + c]) * rstd;
    }
    if (dbf) dbf[c] += a0;
    if (dbm) dbm[c] += a1;
    if (dbeta) dbeta[c] += a2;
    if (dgamma) dgamma[c] += a3;
}

// ---------------------------------------------------------------------------------------------------------------------
// Batch-statistics BatchNorm (model.train(): the reference's default, train.py:271-279,450 -> nn.BatchNorm2d of
// unet.py:40,51 normalises with the statistics of the batch and moves its running buffers).
//   forward : g = act(f) * sigmoid(m) is produced by the linear launch / gate pass with an identity BatchNorm (scale 1, shift 0);
//             bn_stats_kernel   per-channel sum g, sum g^2 over the valid pixels, accumulated in fp64
//             bn_finalize_kernel mean, biased variance -> scale = gamma r, shift = beta - mean scale; running buffers:
//                               rm = (1 - mom) rm + mom mean, rv = (1 - mom) rv + mom var n / (n - 1)   (torch's update)
//             bn_apply_kernel   y = g scale + shift in place (separator rows of a stacked batch stay zero)
//   backward: gate_backward_kernel mode 1 (sums) -> bn_bwd_coeff_kernel -> gate_backward_kernel mode 2
// Statistic groups: 1 = the whole stacked batch is one nn.BatchNorm2d batch (UNet.forward on a (B,8,h,w) tensor); B = every item
// is its own batch of one and the running buffers move B times, in item order (NetAndTexture.forward calls the net once per item,
// READ/models/compose.py:137-176).  Group j = block j of the stacked image: sums[j][2][C], stat[j][2][C], scale_shift[j][2][CoutPad].
// ---------------------------------------------------------------------------------------------------------------------
template <int CW>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ g, long long pixels, int C, double *__restrict__ sums,
                                                       int W, int block_h, int valid_h)
{
    constexpr int ROWS = 256 / CW;
    __shared__ double red[2][256];
    const int c0 = threadIdx.x % CW, r = threadIdx.x / CW;
    long long p_begin = 0, p_end = pixels;
    if (gridDim.y > 1) {
        p_begin = (long long)blockIdx.y * block_h * W;
        p_end = p_begin + (long long)valid_h * W < pixels ? p_begin + (long long)valid_h * W : pixels;
        sums += (size_t)blockIdx.y * 2 * C;
    }
    for (int cb = 0; cb < C; cb += CW) {
        const int c = cb + c0;
        const bool ok = c < C;
        double s1 = 0.0, s2 = 0.0;
        if (ok)
            for (long long p = p_begin + (long long)blockIdx.x * ROWS + r; p < p_end; p += (long long)gridDim.x * ROWS) {
                if (separator_row(p, W, block_h, valid_h)) continue;
                const double v = (double)g[p * C + c];
                s1 += v;
                s2 += v * v;
            }
        red[0][threadIdx.x] = s1;
        red[1][threadIdx.x] = s2;
        __syncthreads();
        if (threadIdx.x < CW && ok) {
            double t0 = 0.0, t1 = 0.0;
            for (int k = 0; k < ROWS; ++k) {
                t0 += red[0][k * CW + threadIdx.x];
                t1 += red[1][k * CW + threadIdx.x];
            }
            atomicAdd(sums + c, t0);
            atomicAdd(sums + C + c, t1);
        }
        __syncthreads();
    }
}

__global__ void bn_finalize_kernel(int C, int CoutPad, int groups, const double *sums, double count, const float *gamma,
                                   const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                   float *stat, float *scale_shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
    for (int j = 0; j < groups; ++j) {                   // the running buffers move once per group, in item order
        const double mean = sums[(size_t)j * 2 * C + c] / count;
        double var = sums[(size_t)j * 2 * C + C + c] / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float mf = (float)mean, vf = (float)var;
        stat[(size_t)j * 2 * C + c] = mf;
        stat[(size_t)j * 2 * C + C + c] = vf;
        const float sc = gamma[c] / sqrtf(vf + eps);
        scale_shift[(size_t)j * 2 * CoutPad + c] = sc;
        scale_shift[(size_t)j * 2 * CoutPad + CoutPad + c] = beta[c] - mf * sc;
        rm = (1.0f - momentum) * rm + momentum * mf;
        rv = (1.0f - momentum) * rv + momentum * (float)(count > 1.0 ? var * count / (count - 1.0) : var);
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(float *__restrict__ y, long long pixels, int C, int CoutPad, int groups,
                                                       const float *__restrict__ scale_shift, int W, int block_h, int valid_h)
{
    const long long total = pixels * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / C;
        const int c = (int)(i - p * C);
        const float *ss = scale_shift + (groups > 1 ? (size_t)((p / W) / block_h) * 2 * CoutPad : 0);
        y[i] = separator_row(p, W, block_h, valid_h) ? 0.0f : y[i] * ss[c] + ss[CoutPad + c];
    }
}

// abc[3][C]: dg = A dy + B + C' g with A = gamma r, C' = -A r dgamma / n, B = -A dbeta / n - C' mean,
// dbeta = sum dy, dgamma = (sum dy g - mean sum dy) r      (sums rows 2, 3 of gate_backward_kernel mode 1)
__global__ void bn_bwd_coeff_kernel(int C, const float *sums, const float *stat, const float *gamma, float eps, float count, float *abc)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    sums += (size_t)blockIdx.y * 4 * C;                  // statistic group
    stat += (size_t)blockIdx.y * 2 * C;
    abc += (size_t)blockIdx.y * 3 * C;
    const float mean = stat[c], r = 1.0f / sqrtf(stat[C + c] + eps);
    const float dbeta = sums[2 * C + c], dgamma = (sums[3 * C + c] - mean * dbeta) * r;
    const float A = gamma[c] * r, Cc = -A * r * dgamma / count;
    abc[c] = A;
    abc[C + c] = -A * dbeta / count - Cc * mean;
    abc[2 * C + c] = Cc;
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad, generic form (any ksize / stride): dx[p][ci] = sum over taps, co' of dfm[q][co'] * W[tap][co'][ci] with
// q * stride + tap - pad == p.  Thread = (input pixel, 4 input channels).  Used for the six stride-2 layers.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dgrad_generic_kernel(const float *__restrict__ dfm, int outH, int outW, int C2,
                                                            const float *__restrict__ wt, int Cin, int ksize, int stride,
                                                            int inH, int inW, float *__restrict__ dx)
{
    const int pad = (ksize - 1) / 2, q4 = Cin >> 2;
    const long long total = (long long)inH * inW * q4;
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(item % q4);
        const long long pix = item / q4;
        const int x = (int)(pix % inW), y = (int)(pix / inW);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < ksize; ++ky) {
            const int ty = y + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= outH) continue;
            for (int kx = 0; kx < ksize; ++kx) {
                const int tx = x + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= outW) continue;
                const float *g = dfm + ((long long)oy * outW + ox) * C2;
                const float *w = wt + ((long long)(ky * ksize + kx) * C2) * Cin + 4 * q;
                for (int c = 0; c < C2; ++c) {
                    const float gv = g[c];
                    const float4 wv = *reinterpret_cast<const float4 *>(w + (long long)c * Cin);
                    acc.x += gv * wv.x;
                    acc.y += gv * wv.y;
                    acc.z += gv * wv.z;
                    acc.w += gv * wv.w;
                }
            }
        }
        *reinterpret_cast<float4 *>(dx + pix * Cin + 4 * q) = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad on the matrix cores: dW_tap[ci][co'] = sum over output pixels of x[in(pixel, tap)][ci] * dfm[pixel][co'].
// v_mfma_f32_32x32x2_f32 with M = 32 input channels, N = 32 channels of d[f|m], K = 2 output pixels: lane l supplies
// A[i = l & 31][k = l >> 5] = x[pixel + k][ci0 + i] and B[k][j = l & 31] = dfm[pixel + k][co0 + j] — with NHWC tensors both
// are two coalesced 128-byte rows per instruction, no transposition anywhere.  A wave owns one (ci tile, co' tile) and
// accumulates NT taps at once (144 accumulator registers for a 3x3 layer), B is loaded once per pixel pair and shared by
// the taps.  The pixel range is split over grid.z (split-K); partial tiles go to a scratch buffer and
// wgrad_reduce_kernel sums them into dW_f / dW_m in the PyTorch layout (Cout, Cin, k, k).
// ---------------------------------------------------------------------------------------------------------------------
// CIB (1x1 layers): the NT accumulators are NT consecutive 32-channel tiles of the INPUT channels instead of NT taps, so the
// d[f|m] operand is loaded once for NT MFMAs (with NT = 1 a 1x1 layer issues two loads per MFMA: 362 us per layer on average).
template <int NT, bool CIB>
__global__ __launch_bounds__(64) void wgrad_mfma_kernel(const float *__restrict__ x, int inH, int inW, int Cin,
                                                        const float *__restrict__ dfm, int outH, int outW, int C2, int ksize,
                                                        int stride, int tiles_co, int rows_per_split,
                                                        float *__restrict__ partial)
{
    const int lane = threadIdx.x;
    const int ci0 = ((int)blockIdx.x / tiles_co) * 32 * (CIB ? NT : 1), co0 = ((int)blockIdx.x % tiles_co) * 32;
    const int pad = (ksize - 1) / 2;
    const int i = lane & 31, kk = lane >> 5;
    const bool ci_ok = ci0 + i < Cin, co_ok = co0 + i < C2;
    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int y_begin = (int)blockIdx.z * rows_per_split;
    const int y_end = min(outH, y_begin + rows_per_split);
    const int taps = ksize * ksize;
    // One k-step = two output pixels (px = 2 p + kk) of one row.  Operands run two steps ahead of the MFMAs (a ring of three
    // register sets: one step of NT MFMAs, ~0.25 us, does not cover an L2 round trip).  The fetch cursor (row, pixel pair) is
    // wave-uniform and advanced with scalar arithmetic; for a pair whose taps all fall inside the image — all but the first
    // and last pair of a row and the first / last rows — a lane's NT loads share ONE 32-bit offset, the tap offsets are scalar
    // (buffer_load ... offen with an SGPR offset), i.e. one VALU add per step instead of a bounds test and an address per tap.  (The first version
    // derived (row, pair) from a 64-bit step index by division and tested every tap: the wave spent more issue slots on
    // addresses than on MFMAs, 18 TF.)
    const int n_pairs = (outW + 1) >> 1;
    const int cil = ci_ok ? ci0 + i : Cin - 1, col = co_ok ? co0 + i : C2 - 1;      // padded rows / columns are never stored
    int tky[NT], tkx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = CIB ? 0 : (int)blockIdx.y * NT + t;
        tky[t] = tap / ksize;
        tkx[t] = tap - tky[t] * ksize;
    }
    const unsigned lane_x = (unsigned)(kk * stride * Cin + cil), lane_d = (unsigned)(kk * C2 + col);
    int f_oy = y_begin, f_p = 0;
    // Both operands through buffer descriptors, and ONE set of load instructions for interior and border steps alike: the two
    // cases only differ in the offsets they compute (a border lane that has nothing to read carries an out-of-range offset and
    // gets zeros).  With the loads inside the two branches hipcc's waitcnt pass could not count them across the merge and drained
    // every load in flight once per RING steps (s_waitcnt vmcnt(0) in front of the first MFMA group of each unrolled iteration,
    // right behind the fetch it had just issued) — the ring hid one step of latency in five.
    constexpr unsigned OOR = 0x80000000u;                                          // tensors are below 2 GiB (read_conv_wgrad)
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (unsigned)(inH * inW * Cin) * 4u, 0x00020000);
    const auto drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dfm), 0, (unsigned)(outH * outW * C2) * 4u, 0x00020000);
    int tapoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tapoff[t] = ((tky[t] * inW + tkx[t]) * Cin + (CIB ? 32 * t : 0)) * 4;
    auto fetch = [&](float (&a)[NT], float &b) {
        const bool live = f_oy < y_end;
        const int oy = live ? f_oy : y_begin, p = live ? f_p : 0;
        const int iy0 = oy * stride - pad, ix0 = 2 * p * stride - pad;             // tap (0,0) of the pair's first pixel
        const bool inside = live && iy0 >= 0 && iy0 + ksize - 1 < inH && ix0 >= 0 && ix0 + stride + ksize - 1 < inW &&
                            2 * p + 1 < outW && (CIB || taps == NT * (int)gridDim.y);
        unsigned vo[NT], vob;
        int so[NT];
        if (inside) {                                                            // wave-uniform: one lane offset, scalar tap offsets
            const unsigned off = ((unsigned)((iy0 * inW + ix0) * Cin) + lane_x) * 4u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                vo[t] = off;
                so[t] = tapoff[t];
            }
            vob = ((unsigned)((oy * outW + 2 * p) * C2) + lane_d) * 4u;
        } else {
            const int px = 2 * p + kk;
            const bool p_ok = live && px < outW;
            vob = (p_ok && co_ok) ? (unsigned)((oy * outW + px) * C2 + col) * 4u : OOR;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int iy = iy0 + tky[t], ix = px * stride - pad + tkx[t];
                const bool ok = p_ok && ci_ok && (CIB || (int)blockIdx.y * NT + t < taps) && iy >= 0 && iy < inH && ix >= 0 && ix < inW;
                vo[t] = ok ? (unsigned)((iy * inW + ix) * Cin + cil + (CIB ? 32 * t : 0)) * 4u : OOR;
                so[t] = 0;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo[t], so[t], 0));
        b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, vob, 0, 0));
        if (++f_p == n_pairs) {
            f_p = 0;
            ++f_oy;
        }
    };
    // Operands RING - 1 steps ahead of the MFMAs in a ring of RING register sets with STATIC roles (the loop is unrolled RING
    // times): one step of NT MFMAs is ~0.27 us per wave, and every step touches lines nobody has read yet (2 new pixels of x per
    // tap row, 2 of d[f|m]) — an HBM round trip of ~2 us sits in front of every step's in-order vmcnt.  (A ring rotated
    // through register moves was measured: two steps ahead 290 us per 3x3 layer, four steps ahead 373 — the moves cost more
    // than the latency they hid.)
    constexpr int RING = NT >= 8 ? 5 : 6;
    const long long total_steps = (long long)(y_end - y_begin) * n_pairs;
    float a[RING][NT], b[RING];
#pragma unroll
    for (int r = 0; r + 1 < RING; ++r) fetch(a[r], b[r]);
    long long step = 0;
    for (; step + RING <= total_steps; step += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            fetch(a[(r + RING - 1) % RING], b[(r + RING - 1) % RING]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][t], b[r], acc[t], 0, 0, 0);
        }
    }
    // tail: at most RING - 1 steps left, already fetched (fetch() returns zeros past the end)
#pragma unroll
    for (int r = 0; r + 1 < RING; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][t], b[r], acc[t], 0, 0, 0);
    // D[i][j] sits in lane (j + 32 * ((i >> 2) & 1)), register (i & 3) + 4 * (i >> 3): row i = ci, column j = co'
    float *dst = partial + (((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NT * 1024);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            dst[(t * 32 + row) * 32 + i] = acc[t][r];                    // [tap][ci][co'] within the tile
        }
}

// dW{f|m}[co][ci][tap] (+)= sum over splits of partial[split][tapgroup][tile][t][ci][co'].  Threads walk the PARTIAL layout
// (co' fastest), so the `splits` reads of a thread's element are coalesced across the wave; the one scattered access is the
// final store.  (Walking the output layout instead cost 94 us per layer: every read of every split was a 4-byte gather.)
__global__ void wgrad_reduce_kernel(const float *__restrict__ partial, int splits, int tap_groups, int NT, int tiles_ci,
                                    int tiles_co, int Cin, int Cout, int Cp, int taps, float *dwf, float *dwm, int accumulate,
                                    int cib)
{
    const long long tile_stride = (long long)NT * 1024;
    const long long tiles = (long long)tiles_ci * tiles_co;
    const long long per_split = (long long)tap_groups * tiles * tile_stride;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < per_split; e += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(e & 31), row = (int)((e >> 5) & 31);
        const int t = (int)((e >> 10) % NT);
        const long long tt = e / tile_stride;                      // tapgroup * tiles + tile
        const int tile = (int)(tt % tiles), tg = (int)(tt / tiles);
        const int tap = cib ? 0 : tg * NT + t;                       // cib: t = tile of input channels inside the block
        const int ci = cib ? ((tile / tiles_co) * NT + t) * 32 + row : (tile / tiles_co) * 32 + row, cp = (tile % tiles_co) * 32 + col;
        const int half = cp >= Cp ? 1 : 0, co = cp - half * Cp;
        if (tap >= taps || ci >= Cin || co >= Cout || cp >= 2 * Cp) continue;
        float s = 0.0f;
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            const float a0 = partial[(long long)k * per_split + e], a1 = partial[(long long)(k + 1) * per_split + e];
            const float a2 = partial[(long long)(k + 2) * per_split + e], a3 = partial[(long long)(k + 3) * per_split + e];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; k < splits; ++k) s += partial[(long long)k * per_split + e];
        float *dst = (half ? dwm : dwf) + ((long long)co * Cin + ci) * taps + tap;
        *dst = accumulate ? *dst + s : s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad of the 3x3 / stride-1 layers in the Winograd F(4x4,3x3) domain: 4x fewer multiplications than wgrad_mfma_kernel<9>.
//
//   forward:  Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A      per 4 x 4 output tile, d = its 6 x 6 input patch
//   hence     dg = G^T [ sum_tiles (B^T d B) . (A dY A^T) ] G   — the elementwise products of two 6 x 6 transforms, summed over
//   the tiles: per frequency (xi, nu) a matrix product  dU[ci][co'] = sum_tiles V[tile][ci] * M[tile][co']  with the TILES as the
//   reduction dimension — 36 MACs per (tile, ci, co') against 16 pixels x 9 taps = 144 of the direct form.
// Workgroup = (32 input channels) x (32 channels of d[f|m]) x a range of tile rows; per iteration eight tiles of one tile row:
//   * thread (tile t = tid >> 5, channel c = tid & 31) loads its channel's 6 x 6 patch of x and its channel's 4 x 4 tile of d[f|m]
//     straight from the NHWC tensors (a wave's load = two 128-byte rows; the bounds tests are per row and for the first / last
//     column of the image only: buffer loads with an out-of-range offset return the zero padding), one iteration AHEAD of their
//     use, transforms both (144 + 90 FMAs / adds) and writes V[36][8][32] and M[36][8][32] to LDS;
//   * wave w owns frequencies 9w .. 9w + 8: v_mfma_f32_32x32x2_f32 with A[i][k] = V[f][2s + k][i], B[k][j] = M[f][2s + k][j] — both
//     operands are conflict-free ds_read_b32 — 36 MFMAs per iteration, 144 accumulator registers;
//   * 74 KiB of LDS per workgroup (two fit a CU; one per CU over the chip measured best: fewer split-K partials).
// Partials [split][tile pair][36][32][32] go to the scratch buffer; wgrad_wino4_reduce_kernel sums the splits and applies G^T . G.
// ---------------------------------------------------------------------------------------------------------------------
struct Wg4 {
    static constexpr int TL = 8;                               // tiles per iteration
    static constexpr int FS = TL * 32;                         // floats per frequency plane
    static constexpr unsigned OOR = 0x80000000u;
};

__device__ __forceinline__ void wg4_bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                        float &t0, float &t1, float &t2, float &t3, float &t4, float &t5)
{   // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
    const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3), c = d4 - d2, e = d3 - d1;
    t0 = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    t1 = a + b;
    t2 = a - b;
    t3 = fmaf(2.0f, e, c);
    t4 = fmaf(-2.0f, e, c);
    t5 = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
}

__device__ __forceinline__ void wg4_a4(const float y0, const float y1, const float y2, const float y3,
                                       float &o0, float &o1, float &o2, float &o3, float &o4, float &o5)
{   // A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]
    const float s02 = y0 + y2, s13 = y1 + y3, p = fmaf(4.0f, y2, y0), q = fmaf(8.0f, y3, 2.0f * y1);
    o0 = y0;
    o1 = s02 + s13;
    o2 = s02 - s13;
    o3 = p + q;
    o4 = p - q;
    o5 = y3;
}

__global__ __launch_bounds__(256, 2) void wgrad_wino4_kernel(const float *__restrict__ x, int H, int W, int Cin,
                                                            const float *__restrict__ dfm, int C2, int tiles_co,
                                                            int rows_per_split, float *__restrict__ partial)
{
    __shared__ float Vs[36 * Wg4::FS], Ms[36 * Wg4::FS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = tid >> 5, c = tid & 31;
    const int ci0 = ((int)blockIdx.x / tiles_co) * 32, co0 = ((int)blockIdx.x % tiles_co) * 32;
    const int tiles_x = W >> 2, tiles_y = H >> 2, groups_x = (tiles_x + Wg4::TL - 1) / Wg4::TL;
    const int ty_begin = (int)blockIdx.z * rows_per_split, ty_end = min(tiles_y, ty_begin + rows_per_split);
    const int n_it = (ty_end - ty_begin) * groups_x;
    const bool co_ok = co0 + c < C2;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (unsigned)(H * W * Cin) * 4u, 0x00020000);
    const auto drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dfm), 0, (unsigned)(H * W * C2) * 4u, 0x00020000);
    int xcol[6], dcol[4];                                        // wave-uniform column offsets (bytes): SGPRs
#pragma unroll
    for (int j = 0; j < 6; ++j) xcol[j] = j * Cin * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dcol[j] = j * C2 * 4;

    float xr[6][6], dr[4][4];
    int f_ty = ty_begin, f_g = 0;
    auto fetch = [&]() {
        const int tx = f_g * Wg4::TL + t;
        const bool tile_ok = tx < tiles_x && f_ty < ty_end;
        const int y0 = 4 * f_ty - 1, x0 = 4 * tx - 1;
        unsigned base[6], base_l[6], base_r[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int yy = y0 + r;
            const bool ok = tile_ok && yy >= 0 && yy < H;
            base[r] = ok ? (unsigned)((yy * W + x0 + 1) * Cin + ci0 + c) * 4u : Wg4::OOR;    // column x0 + 1 >= 0: no negative offsets
            base_l[r] = (ok && x0 >= 0) ? base[r] - (unsigned)Cin * 4u : Wg4::OOR;
            base_r[r] = x0 + 5 < W ? base[r] : Wg4::OOR;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                xr[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, j == 0 ? base_l[r] : j == 5 ? base_r[r] : base[r],
                                                                                          j == 0 ? 0 : xcol[j - 1], 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned db = (tile_ok && co_ok) ? (unsigned)(((4 * f_ty + r) * W + 4 * tx) * C2 + co0 + c) * 4u : Wg4::OOR;
#pragma unroll
            for (int j = 0; j < 4; ++j) dr[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, db, dcol[j], 0));
        }
        if (++f_g == groups_x) {
            f_g = 0;
            ++f_ty;
        }
    };

    floatx16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;
    const int i32 = lane & 31, kk = lane >> 5;
    fetch();
    for (int it = 0; it < n_it; ++it) {
        {   // V = B^T d B of this thread's (tile, input channel); M = A dY A^T of its (tile, d[f|m] channel)
            float u[6][6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                wg4_bt6(xr[0][j], xr[1][j], xr[2][j], xr[3][j], xr[4][j], xr[5][j], u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j]);
            float *vp = Vs + t * 32 + c;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float v0, v1, v2, v3, v4, v5;
                wg4_bt6(u[r][0], u[r][1], u[r][2], u[r][3], u[r][4], u[r][5], v0, v1, v2, v3, v4, v5);
                vp[(6 * r + 0) * Wg4::FS] = v0;
                vp[(6 * r + 1) * Wg4::FS] = v1;
                vp[(6 * r + 2) * Wg4::FS] = v2;
                vp[(6 * r + 3) * Wg4::FS] = v3;
                vp[(6 * r + 4) * Wg4::FS] = v4;
                vp[(6 * r + 5) * Wg4::FS] = v5;
            }
            float w[6][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wg4_a4(dr[0][j], dr[1][j], dr[2][j], dr[3][j], w[0][j], w[1][j], w[2][j], w[3][j], w[4][j], w[5][j]);
            float *mp = Ms + t * 32 + c;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float m0, m1, m2, m3, m4, m5;
                wg4_a4(w[r][0], w[r][1], w[r][2], w[r][3], m0, m1, m2, m3, m4, m5);
                mp[(6 * r + 0) * Wg4::FS] = m0;
                mp[(6 * r + 1) * Wg4::FS] = m1;
                mp[(6 * r + 2) * Wg4::FS] = m2;
                mp[(6 * r + 3) * Wg4::FS] = m3;
                mp[(6 * r + 4) * Wg4::FS] = m4;
                mp[(6 * r + 5) * Wg4::FS] = m5;
            }
        }
        fetch();                                                 // the next iteration's patches travel under the MFMAs (zeros past the end)
        __syncthreads();
        const float *va = Vs + (wv * 9) * Wg4::FS + kk * 32 + i32, *mb = Ms + (wv * 9) * Wg4::FS + kk * 32 + i32;
#pragma unroll
        for (int f = 0; f < 9; ++f)
#pragma unroll
            for (int sidx = 0; sidx < Wg4::TL / 2; ++sidx)
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[f * Wg4::FS + sidx * 64], mb[f * Wg4::FS + sidx * 64], acc[f], 0, 0, 0);
        __syncthreads();
    }
    // D[i][j] sits in lane (j + 32 * ((i >> 2) & 1)), register (i & 3) + 4 * (i >> 3): row i = ci, column j = co'
    float *dst = partial + ((long long)blockIdx.z * gridDim.x + blockIdx.x) * (36 * 1024) + (long long)(wv * 9) * 1024;
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            dst[(f * 32 + row) * 32 + i32] = acc[f][r];
        }
}

// Step 1 of the reduction: partial[0][e] = sum over splits of partial[split][e] for every element of the [tile pair][36][ci][co']
// block — thread = element, so the `splits` reads of a wave are coalesced and there are 36 x 1024 threads per tile pair (one
// thread per (ci, co') walking all 36 frequencies and all splits was 6500 dependent loads long: 700 us per layer).
__global__ __launch_bounds__(256) void wgrad_wino4_sum_kernel(float *__restrict__ partial, int splits, long long per_split)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < per_split; e += (long long)gridDim.x * blockDim.x) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            s0 += partial[(long long)k * per_split + e];
            s1 += partial[(long long)(k + 1) * per_split + e];
            s2 += partial[(long long)(k + 2) * per_split + e];
            s3 += partial[(long long)(k + 3) * per_split + e];
        }
        for (; k < splits; ++k) s0 += partial[(long long)k * per_split + e];
        partial[e] = (s0 + s1) + (s2 + s3);
    }
}

// Step 2: dW{f|m}[co][ci][3][3] (+)= G^T U G with U = the summed [36][ci][co'] block of a tile pair,
// G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1].  Thread = (tile pair, ci, co'), co' fastest.
__global__ __launch_bounds__(256) void wgrad_wino4_reduce_kernel(const float *__restrict__ summed, int tiles_ci, int tiles_co, int Cin,
                                                                 int Cout, int Cp, float *dwf, float *dwm, int accumulate)
{
    const long long tiles = (long long)tiles_ci * tiles_co;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tiles * 1024; e += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(e & 31), row = (int)((e >> 5) & 31), tile = (int)(e >> 10);
        const int ci = (tile / tiles_co) * 32 + row, cp = (tile % tiles_co) * 32 + col;
        const int half = cp >= Cp ? 1 : 0, co = cp - half * Cp;
        if (ci >= Cin || co >= Cout || cp >= 2 * Cp) continue;
        const float *src = summed + (long long)tile * 36 * 1024 + row * 32 + col;
        float u[6][6];
#pragma unroll
        for (int f = 0; f < 36; ++f) u[f / 6][f % 6] = src[f * 1024];
        // rows: G^T u (3 x 6), then columns: (G^T u) G (3 x 3)
        float gtu[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float s12 = u[1][j] + u[2][j], d12 = u[2][j] - u[1][j], s34 = u[3][j] + u[4][j], d34 = u[3][j] - u[4][j];
            gtu[0][j] = 0.25f * u[0][j] - s12 * (1.0f / 6.0f) + s34 * (1.0f / 24.0f);
            gtu[1][j] = d12 * (1.0f / 6.0f) + d34 * (1.0f / 12.0f);
            gtu[2][j] = -s12 * (1.0f / 6.0f) + s34 * (1.0f / 6.0f) + u[5][j];
        }
        float *dst = (half ? dwm : dwf) + ((long long)co * Cin + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float *g = gtu[a];
            const float s12 = g[1] + g[2], d12 = g[2] - g[1], s34 = g[3] + g[4], d34 = g[3] - g[4];
            const float o0 = 0.25f * g[0] - s12 * (1.0f / 6.0f) + s34 * (1.0f / 24.0f);
            const float o1 = d12 * (1.0f / 6.0f) + d34 * (1.0f / 12.0f);
            const float o2 = -s12 * (1.0f / 6.0f) + s34 * (1.0f / 6.0f) + g[5];
            dst[3 * a + 0] = accumulate ? dst[3 * a + 0] + o0 : o0;
            dst[3 * a + 1] = accumulate ? dst[3 * a + 1] + o1 : o1;
            dst[3 * a + 2] = accumulate ? dst[3 * a + 2] + o2 : o2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bilinear x4 upsample, backward (adjoint of bilinear_up4_kernel in conv.hip): thread = (input pixel, 4 channels)
// gathers the up-to-6x6 output pixels that read it.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up4_src(int o, int n_in, int &i0, int &i1, float &l1)
{
    float s = 0.25f * ((float)o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// Vertically stacked batch: rows are interpolated INSIDE an item (block_h input rows per item, the first valid_h valid),
// exactly as nn.Upsample treats separate images; separator rows of the output are zero.  block_h = 0: one image.
__global__ __launch_bounds__(256) void bilinear_up4_blocks_kernel(const float *__restrict__ in, int inH, int inW, int C,
                                                                  float *__restrict__ out, int block_h, int valid_h)
{
    const int outW = inW * 4, q4 = C >> 2;
    const int bh = block_h > 0 ? block_h : inH, vh = block_h > 0 ? valid_h : inH;
    const long long total = (long long)inH * 4 * outW * q4;
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(item % q4);
        const long long pix = item / q4;
        const int ox = (int)(pix % outW), oy = (int)(pix / outW);
        const int blk = oy / (4 * bh), oyl = oy - blk * 4 * bh;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oyl < 4 * vh) {
            int y0, y1, x0, x1;
            float ly1, lx1;
            up4_src(oyl, vh, y0, y1, ly1);
            up4_src(ox, inW, x0, x1, lx1);
            const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const float *base = in + (long long)blk * bh * inW * C + 4 * q;
            const float4 v00 = *reinterpret_cast<const float4 *>(base + ((long long)y0 * inW + x0) * C);
            const float4 v01 = *reinterpret_cast<const float4 *>(base + ((long long)y0 * inW + x1) * C);
            const float4 v10 = *reinterpret_cast<const float4 *>(base + ((long long)y1 * inW + x0) * C);
            const float4 v11 = *reinterpret_cast<const float4 *>(base + ((long long)y1 * inW + x1) * C);
            o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
            o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
            o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
            o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        }
        *reinterpret_cast<float4 *>(out + pix * C + 4 * q) = o;
    }
}

__global__ __launch_bounds__(256) void bilinear_up4_backward_kernel(const float *__restrict__ dout, int inH, int inW, int C,
                                                                    float *__restrict__ din, int block_h, int valid_h)
{
    const int outW = inW * 4, q4 = C >> 2;
    const int bh = block_h > 0 ? block_h : inH, vh = block_h > 0 ? valid_h : inH;
    const long long total = (long long)inH * inW * q4;
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(item % q4);
        const long long pix = item / q4;
        const int x = (int)(pix % inW), yg = (int)(pix / inW);
        const int blk = yg / bh, y = yg - blk * bh;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < vh) {
            const float *base = dout + (long long)blk * 4 * bh * outW * C + 4 * q;
            // output rows whose source interval [i0, i1] contains y lie within 4y-4 .. 4y+5 (borders: clamped sources)
            for (int oy = max(0, 4 * y - 4); oy <= min(4 * vh - 1, 4 * y + 5); ++oy) {
                int y0, y1;
                float ly1;
                up4_src(oy, vh, y0, y1, ly1);
                const float wy = (y0 == y ? 1.f - ly1 : 0.f) + (y1 == y ? ly1 : 0.f);
                if (wy == 0.f) continue;
                for (int ox = max(0, 4 * x - 4); ox <= min(outW - 1, 4 * x + 5); ++ox) {
                    int x0, x1;
                    float lx1;
                    up4_src(ox, inW, x0, x1, lx1);
                    const float wx = (x0 == x ? 1.f - lx1 : 0.f) + (x1 == x ? lx1 : 0.f);
                    if (wx == 0.f) continue;
                    const float4 g = *reinterpret_cast<const float4 *>(base + ((long long)oy * outW + ox) * C);
                    const float w = wy * wx;
                    acc.x += w * g.x;
                    acc.y += w * g.y;
                    acc.z += w * g.z;
                    acc.w += w * g.w;
                }
            }
        }
        *reinterpret_cast<float4 *>(din + pix * C + 4 * q) = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Huber loss (F.huber_loss, delta = 1, mean reduction; src/READ/models/compose.py:35,38) and its gradient
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void huber_kernel(const float *__restrict__ out, const float *__restrict__ target,
                                                    long long n, float scale, float *__restrict__ loss_sum,
                                                    float *__restrict__ grad)
{
    __shared__ float red[256];
    float s = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = out[i] - target[i];
        const float ad = fabsf(d);
        s += ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
        if (grad) grad[i] = scale * (ad < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, red[0]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Sparse RMSprop over the descriptor rows a step touched (torch.optim.RMSprop defaults of READ/pipelines/ogl.py:16:
// alpha 0.99, eps 1e-8, no momentum, not centered):  sq = alpha sq + (1 - alpha) g^2;  p -= lr g / (sqrt(sq) + eps).
// ids are the int32 index maps of the step (all levels, all batch items); a per-row epoch stamp makes every touched row
// update exactly once however many pixels hit it.  Rows that no pixel touched have zero gradient; the dense optimizer
// would still decay their sq by alpha each step — `lazy_decay` applies alpha^(steps missed) when a row is touched again, so
// the trajectory of every row equals the dense one (tests/test_gpu_train.py).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsprop_sparse_kernel(float *__restrict__ rows, float *__restrict__ sq,
                                                             float *__restrict__ grad, int *__restrict__ stamp, int C,
                                                             long long n_rows, const int32_t *__restrict__ ids,
                                                             long long n_ids, int step, float lr, float alpha, float eps)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_ids; i += (long long)gridDim.x * blockDim.x) {
        long long id = ids[i];
        if (id < 0 || id >= n_rows) continue;
        const int last = atomicExch(stamp + id, step);                 // first thread to see an old stamp owns the row
        if (last == step) continue;
        const float decay = powf(alpha, (float)(step - 1 - last));      // steps in which the dense optimizer saw g = 0
        for (int c = 0; c < C; ++c) {
            const float g = grad[id * C + c];
            const float v = alpha * (sq[id * C + c] * decay) + (1.0f - alpha) * g * g;
            sq[id * C + c] = v;
            rows[id * C + c] -= lr * g / (sqrtf(v) + eps);
            grad[id * C + c] = 0.0f;                                   // leave the gradient buffer clean for the next step
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int read_conv_pack_params_device(int Cout, const float *bf, const float *bm, const float *gamma, const float *beta,
                                            const float *mean, const float *var, float eps, float *params, void *stream)
{
    READ_CHECK_ARG(Cout >= 1 && gamma && beta && mean && var && params, "read_conv_pack_params_device: null pointer");
    const int CoutPad = (Cout + 31) / 32 * 32;
    hipLaunchKernelGGL(pack_params_kernel, dim3(ceil_div(CoutPad, 64)), dim3(64), 0, as_stream(stream), Cout, CoutPad, bf, bm,
                       gamma, beta, mean, var, eps, params);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_conv_pack_weights_device(int Cin, int Cout, int ksize, int kc, const float *wf, const float *wm,
                                             float *wpacked, void *stream)
{
    READ_CHECK_ARG(wf && wm && wpacked, "read_conv_pack_weights_device: null pointer");
    READ_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 4, "read_conv_pack_weights_device: ksize must be 1, 3 or 4");
    READ_CHECK_ARG((kc == 8 || kc == 16 || (kc == 32 && ksize == 1)) && Cin >= kc && Cin % kc == 0,
                   "read_conv_pack_weights_device: Cin=%d is not a multiple of kc=%d", Cin, kc);
    const long long total = (long long)read_conv_packed_floats(Cin, Cout, ksize);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), 0, Cin, Cout, ksize, kc, 0,
                       wf, wm, wpacked, total);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_conv_pack_wino_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_wino, void *stream)
{
    READ_CHECK_ARG(wf && wm && wpacked_wino, "read_conv_pack_wino_device: null pointer");
    READ_CHECK_ARG(Cin >= 16 && Cin % 16 == 0 && Cout >= 1, "read_conv_pack_wino_device: needs Cin %% 16 == 0 (got %d)", Cin);
    const long long total = (long long)read_conv_wino_floats(Cin, Cout);
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)(total / (32 * 256))), dim3(256), 0, as_stream(stream), 0, Cin, Cout, 0, wf, wm,
                       wpacked_wino);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" size_t read_conv_dgrad_wino_floats(int Cin, int Cout)
{
    if (Cin < 2 || Cin % 2 || Cout < 1) return 0;
    return read_conv_wino_floats(2 * ((Cout + 7) / 8 * 8), Cin / 2);
}

extern "C" int read_conv_pack_dgrad_wino_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_wino,
                                                void *stream)
{
    READ_CHECK_ARG(wf && wm && wpacked_wino, "read_conv_pack_dgrad_wino_device: null pointer");
    READ_CHECK_ARG(Cin >= 2 && Cin % 2 == 0 && Cout >= 1, "read_conv_pack_dgrad_wino_device: Cin must be even");
    const int Cp = (Cout + 7) / 8 * 8;
    const long long total = (long long)read_conv_dgrad_wino_floats(Cin, Cout);
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)(total / (32 * 256))), dim3(256), 0, as_stream(stream), 1, Cin, Cout, Cp, wf, wm,
                       wpacked_wino);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_conv_pack_w4_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_w4, void *stream)
{
    READ_CHECK_ARG(wf && wm && wpacked_w4, "read_conv_pack_w4_device: null pointer");
    READ_CHECK_ARG(Cin >= 16 && Cin % 16 == 0 && Cout >= 1, "read_conv_pack_w4_device: needs Cin %% 16 == 0 (got %d)", Cin);
    const long long total = (long long)read_conv_w4_floats(Cin, Cout);
    hipLaunchKernelGGL(pack_w4_kernel, dim3((unsigned)(total / (36 * 256))), dim3(256), 0, as_stream(stream), 0, Cin, Cout, 0, wf, wm, wpacked_w4);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" size_t read_conv_dgrad_w4_floats(int Cin, int Cout)
{
    if (Cin < 2 || Cin % 2 || Cout < 1) return 0;
    return read_conv_w4_floats(2 * ((Cout + 7) / 8 * 8), Cin / 2);
}

extern "C" int read_conv_pack_dgrad_w4_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_w4, void *stream)
{
    READ_CHECK_ARG(wf && wm && wpacked_w4, "read_conv_pack_dgrad_w4_device: null pointer");
    READ_CHECK_ARG(Cin >= 2 && Cin % 2 == 0 && Cout >= 1, "read_conv_pack_dgrad_w4_device: Cin must be even");
    const int Cp = (Cout + 7) / 8 * 8;
    const long long total = (long long)read_conv_dgrad_w4_floats(Cin, Cout);
    hipLaunchKernelGGL(pack_w4_kernel, dim3((unsigned)(total / (36 * 256))), dim3(256), 0, as_stream(stream), 1, Cin, Cout, Cp, wf, wm, wpacked_w4);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" size_t read_conv_dgrad_packed_floats(int Cin, int Cout, int ksize)
{
    if (Cin < 2 || Cin % 2 || Cout < 1) return 0;
    const int Cp = (Cout + 7) / 8 * 8;
    return read_conv_packed_floats(2 * Cp, Cin / 2, ksize);
}

extern "C" int read_conv_pack_dgrad_device(int Cin, int Cout, int ksize, int kc, const float *wf, const float *wm,
                                           float *wpacked, void *stream)
{
    READ_CHECK_ARG(wf && wm && wpacked, "read_conv_pack_dgrad_device: null pointer");
    READ_CHECK_ARG(Cin >= 2 && Cin % 2 == 0, "read_conv_pack_dgrad_device: Cin must be even");
    const int Cp = (Cout + 7) / 8 * 8;
    READ_CHECK_ARG((kc == 8 || kc == 16) && (2 * Cp) % kc == 0, "read_conv_pack_dgrad_device: 2*pad8(Cout)=%d is not a multiple of kc=%d",
                   2 * Cp, kc);
    const long long total = (long long)read_conv_dgrad_packed_floats(Cin, Cout, ksize);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), 1, Cin, Cout, ksize, kc, Cp,
                       wf, wm, wpacked, total);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// Fills the derived fields of a packing job (Cp, total, nblocks) from its kind / mode / shape; first_block stays the caller's.
extern "C" int read_conv_pack_job_prepare(read_pack_job *job)
{
    READ_CHECK_ARG(job, "read_conv_pack_job_prepare: null job");
    read_pack_job &j = *job;
    READ_CHECK_ARG(j.Cout >= 1 && j.out, "read_conv_pack_job_prepare: bad job (Cout %d)", j.Cout);
    j.Cp = (j.Cout + 7) / 8 * 8;
    if (j.kind == READ_PACK_PARAMS) {
        READ_CHECK_ARG(j.gamma && j.beta && j.mean && j.var, "read_conv_pack_job_prepare: params job needs gamma / beta / mean / var");
        j.total = 4ll * ((j.Cout + 31) / 32 * 32);
        j.nblocks = ((j.Cout + 31) / 32 * 32 + 255) / 256;
        return READ_OK;
    }
    READ_CHECK_ARG(j.wf && j.wm && j.Cin >= 2 && (j.mode == 0 || j.Cin % 2 == 0), "read_conv_pack_job_prepare: bad weights job");
    const int CinV = j.mode ? 2 * j.Cp : j.Cin, CoutV = j.mode ? j.Cin / 2 : j.Cout;
    if (j.kind == READ_PACK_DIRECT) {
        READ_CHECK_ARG(j.ksize == 1 || j.ksize == 3 || j.ksize == 4, "read_conv_pack_job_prepare: ksize must be 1, 3 or 4");
        READ_CHECK_ARG((j.kc == 8 || j.kc == 16 || (j.kc == 32 && j.ksize == 1)) && CinV % j.kc == 0,
                       "read_conv_pack_job_prepare: %d input channels are not a multiple of kc=%d", CinV, j.kc);
        j.total = (long long)read_conv_packed_floats(CinV, CoutV, j.ksize);
        const long long b = (j.total + 256 * 8 - 1) / (256 * 8);
        j.nblocks = (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
    } else if (j.kind == READ_PACK_WINO) {
        READ_CHECK_ARG(CinV % 16 == 0, "read_conv_pack_job_prepare: Winograd fragments need Cin %% 16 == 0 (got %d)", CinV);
        j.total = (long long)read_conv_wino_floats(CinV, CoutV);
        j.nblocks = (int)(j.total / (32 * 256));
    } else if (j.kind == READ_PACK_W4) {
        READ_CHECK_ARG(CinV % 16 == 0, "read_conv_pack_job_prepare: Winograd fragments need Cin %% 16 == 0 (got %d)", CinV);
        j.total = (long long)read_conv_w4_floats(CinV, CoutV);
        j.nblocks = (int)(j.total / (36 * 256));
    } else {
        READ_CHECK_ARG(false, "read_conv_pack_job_prepare: unknown kind %d", j.kind);
    }
    return READ_OK;
}

extern "C" int read_conv_pack_batch(const read_pack_job *jobs_dev, int njobs, int total_blocks, void *stream)
{
    READ_CHECK_ARG(jobs_dev && njobs >= 1 && total_blocks >= 1, "read_conv_pack_batch: empty job table");
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), jobs_dev, njobs);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_gate_forward(const float *fm, int64_t pixels, int Cout, const float *params, int elu,
                                 const float *residual, float *y, int W, int block_h, int valid_h, void *stream)
{
    READ_CHECK_ARG(fm && params && y && pixels >= 1 && Cout >= 1, "read_gate_forward: null pointer or empty tensor");
    READ_CHECK_ARG(block_h == 0 || (W >= 1 && valid_h >= 1 && valid_h <= block_h), "read_gate_forward: bad block geometry");
    hipLaunchKernelGGL(gate_forward_kernel, dim3(grid_for(pixels * Cout)), dim3(256), 0, as_stream(stream), fm,
                       (long long)pixels, Cout, (Cout + 31) / 32 * 32, params, elu, residual, y, W > 0 ? W : 1, block_h, valid_h);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_gate_backward(const float *dy, const float *fm, int64_t pixels, int Cout, const float *params, int elu,
                                  float *dfm, float *sums, int W, int block_h, int valid_h, void *stream)
{
    READ_CHECK_ARG(dy && fm && params && dfm && sums && pixels >= 1 && Cout >= 1, "read_gate_backward: null pointer or empty tensor");
    READ_CHECK_ARG(block_h == 0 || (W >= 1 && valid_h >= 1 && valid_h <= block_h), "read_gate_backward: bad block geometry");
    if (W < 1) W = 1;
    const int Cp = (Cout + 7) / 8 * 8, CoutPad = (Cout + 31) / 32 * 32;
    // sums is ACCUMULATED into: the caller hands it over zero-filled (a step's host zero-fills ONE tensor for all of its layers'
    // sums and parameter gradients; a memset per layer here was one more ~3 us launch in a chain that is bound by its launches)
    if (Cout % 32 == 0 && Cout <= 256) {
        const int blocks = grid_for(pixels, 1024 / Cout, 2048);
        hipLaunchKernelGGL(gate_backward4_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, fm, (long long)pixels, Cout,
                           CoutPad, params, elu, dfm, sums, W, block_h, valid_h, 0, (const float *)nullptr);
    } else if (Cp <= 8) {
        const int blocks = grid_for(pixels, 32, 2048);
        hipLaunchKernelGGL(gate_backward_kernel<8>, dim3(blocks), dim3(256), 0, as_stream(stream), dy, fm, (long long)pixels, Cout,
                           CoutPad, Cp, params, elu, dfm, sums, W, block_h, valid_h, 0, (const float *)nullptr);
    } else {
        const int blocks = grid_for(pixels, 8, 2048);
        hipLaunchKernelGGL(gate_backward_kernel<32>, dim3(blocks), dim3(256), 0, as_stream(stream), dy, fm, (long long)pixels, Cout,
                           CoutPad, Cp, params, elu, dfm, sums, W, block_h, valid_h, 0, (const float *)nullptr);
    }
    READ_CHECK_LAUNCH();
    return READ_OK;
}

static long long valid_pixels(int64_t pixels, int W, int block_h, int valid_h)
{
    if (block_h <= 0) return pixels;
    const long long rows = pixels / W, full = rows / block_h, rem = rows % block_h;
    return (full * valid_h + (rem < valid_h ? rem : valid_h)) * W;
}

// groups: 1, or the number of stacked items (then pixels == groups * block_h * W and every item has valid_h * W valid pixels)
static bool groups_fit(int groups, int64_t pixels, int W, int block_h)
{
    return groups == 1 || (groups > 1 && block_h > 0 && pixels == (int64_t)groups * block_h * W);
}

extern "C" int read_bn_train_forward(float *g_to_y, int64_t pixels, int C, int W, int block_h, int valid_h, int groups,
                                     const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                     float *running_var, float *stat, float *scale_shift, double *scratch, void *stream)
{
    READ_CHECK_ARG(g_to_y && gamma && beta && stat && scale_shift && scratch && pixels >= 1 && C >= 1,
                   "read_bn_train_forward: null pointer or empty tensor");
    READ_CHECK_ARG(block_h == 0 || (W >= 1 && valid_h >= 1 && valid_h <= block_h), "read_bn_train_forward: bad block geometry");
    if (W < 1) W = 1;
    READ_CHECK_ARG(groups_fit(groups, pixels, W, block_h), "read_bn_train_forward: statistic groups must be 1 or the number of stacked blocks");
    const long long n = groups > 1 ? (long long)valid_h * W : valid_pixels(pixels, W, block_h, valid_h);
    READ_CHECK_ARG(n >= 1, "read_bn_train_forward: no valid pixel");
    const int CoutPad = (C + 31) / 32 * 32;
    const long long span = groups > 1 ? (long long)valid_h * W : (long long)pixels;      // pixels one statistic group walks
    READ_CHECK_HIP(hipMemsetAsync(scratch, 0, sizeof(double) * 2 * (size_t)C * groups, as_stream(stream)));
    if (C <= 8)
        hipLaunchKernelGGL(bn_stats_kernel<8>, dim3(grid_for(span, 32, 1024), groups), dim3(256), 0, as_stream(stream), g_to_y,
                           (long long)pixels, C, scratch, W, block_h, valid_h);
    else
        hipLaunchKernelGGL(bn_stats_kernel<32>, dim3(grid_for(span, 8, 1024), groups), dim3(256), 0, as_stream(stream), g_to_y,
                           (long long)pixels, C, scratch, W, block_h, valid_h);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, as_stream(stream), C, CoutPad, groups, scratch,
                       (double)n, gamma, beta, eps, momentum, running_mean, running_var, stat, scale_shift);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(pixels * C)), dim3(256), 0, as_stream(stream), g_to_y, (long long)pixels, C,
                       CoutPad, groups, scale_shift, W, block_h, valid_h);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_gate_backward_bn(const float *dy, const float *fm, int64_t pixels, int Cout, const float *params, int elu,
                                     float *dfm, float *sums, int W, int block_h, int valid_h, int groups, const float *stat,
                                     const float *gamma, float eps, float *abc, void *stream)
{
    READ_CHECK_ARG(dy && fm && params && dfm && sums && stat && gamma && abc && pixels >= 1 && Cout >= 1,
                   "read_gate_backward_bn: null pointer or empty tensor");
    READ_CHECK_ARG(block_h == 0 || (W >= 1 && valid_h >= 1 && valid_h <= block_h), "read_gate_backward_bn: bad block geometry");
    if (W < 1) W = 1;
    READ_CHECK_ARG(groups_fit(groups, pixels, W, block_h), "read_gate_backward_bn: statistic groups must be 1 or the number of stacked blocks");
    const int Cp = (Cout + 7) / 8 * 8, CoutPad = (Cout + 31) / 32 * 32;
    const long long n = groups > 1 ? (long long)valid_h * W : valid_pixels(pixels, W, block_h, valid_h);
    const long long span = groups > 1 ? (long long)block_h * W : (long long)pixels;
    for (int mode = 1; mode <= 2; ++mode) {
        READ_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(float) * 4 * (size_t)Cout * groups, as_stream(stream)));
        if (Cout % 32 == 0 && Cout <= 256)
            hipLaunchKernelGGL(gate_backward4_kernel, dim3(grid_for(span, 1024 / Cout, 2048), groups), dim3(256), 0, as_stream(stream), dy,
                               fm, (long long)pixels, Cout, CoutPad, params, elu, dfm, sums, W, block_h, valid_h, mode, (const float *)abc);
        else if (Cp <= 8)
            hipLaunchKernelGGL(gate_backward_kernel<8>, dim3(grid_for(span, 32, 2048), groups), dim3(256), 0, as_stream(stream), dy, fm,
                               (long long)pixels, Cout, CoutPad, Cp, params, elu, dfm, sums, W, block_h, valid_h, mode, (const float *)abc);
        else
            hipLaunchKernelGGL(gate_backward_kernel<32>, dim3(grid_for(span, 8, 2048), groups), dim3(256), 0, as_stream(stream), dy, fm,
                               (long long)pixels, Cout, CoutPad, Cp, params, elu, dfm, sums, W, block_h, valid_h, mode, (const float *)abc);
        if (mode == 1)
            hipLaunchKernelGGL(bn_bwd_coeff_kernel, dim3(ceil_div(Cout, 64), groups), dim3(64), 0, as_stream(stream), Cout, sums, stat,
                               gamma, eps, (float)n, abc);
    }
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_bn_param_grads_groups(int Cout, int groups, const float *sums, const float *stat, float eps, float *dbf,
                                          float *dbm, float *dgamma, float *dbeta, void *stream)
{
    READ_CHECK_ARG(Cout >= 1 && groups >= 1 && sums && stat, "read_bn_param_grads_groups: null pointer");
    hipLaunchKernelGGL(bn_grads_groups_kernel, dim3(ceil_div(Cout, 64)), dim3(64), 0, as_stream(stream), Cout, groups, sums, stat,
                       eps, dbf, dbm, dgamma, dbeta);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_bn_param_grads(int Cout, const float *sums, const float *mean, const float *var, float eps, float *dbf,
                                   float *dbm, float *dgamma, float *dbeta, void *stream)
{
    READ_CHECK_ARG(Cout >= 1 && sums && mean && var, "read_bn_param_grads: null pointer");
    hipLaunchKernelGGL(bn_grads_kernel, dim3(ceil_div(Cout, 64)), dim3(64), 0, as_stream(stream), Cout, sums, mean, var, eps, dbf,
                       dbm, dgamma, dbeta);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" size_t read_conv_dgrad_generic_floats(int Cin, int Cout, int ksize)
{
    const int Cp = (Cout + 7) / 8 * 8;
    return (size_t)ksize * ksize * 2 * Cp * Cin;
}

extern "C" int read_conv_dgrad_generic(const float *dfm, int outH, int outW, int Cin, int Cout, int ksize, int stride,
                                       const float *wf, const float *wm, float *wscratch, int inH, int inW, float *dx,
                                       void *stream)
{
    READ_CHECK_ARG(dfm && wf && wm && wscratch && dx, "read_conv_dgrad_generic: null pointer");
    READ_CHECK_ARG(Cin >= 4 && Cin % 4 == 0 && Cout >= 1, "read_conv_dgrad_generic: Cin must be a multiple of 4");
    READ_CHECK_ARG(ksize >= 1 && ksize <= 4 && (stride == 1 || stride == 2), "read_conv_dgrad_generic: bad ksize / stride");
    const int Cp = (Cout + 7) / 8 * 8, pad = (ksize - 1) / 2;
    READ_CHECK_ARG(outH == (inH + 2 * pad - ksize) / stride + 1 && outW == (inW + 2 * pad - ksize) / stride + 1,
                   "read_conv_dgrad_generic: %dx%d is not the output size of a %dx%d input", outH, outW, inH, inW);
    const long long wtotal = (long long)read_conv_dgrad_generic_floats(Cin, Cout, ksize);
    hipLaunchKernelGGL(pack_dgrad_generic_kernel, dim3(grid_for(wtotal)), dim3(256), 0, as_stream(stream), Cin, Cout,
                       ksize * ksize, Cp, wf, wm, wscratch, wtotal);
    READ_CHECK_LAUNCH();
    hipLaunchKernelGGL(dgrad_generic_kernel, dim3(grid_for((long long)inH * inW * (Cin / 4))), dim3(256), 0, as_stream(stream),
                       dfm, outH, outW, 2 * Cp, (const float *)wscratch, Cin, ksize, stride, inH, inW, dx);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

namespace readhip {
int g_wgrad_wino = 1;     // read_tuning_set("wgrad_wino", 0): 3x3 / stride-1 weight gradients back on the direct kernel
void train_set_wgrad_wino(int v) { g_wgrad_wino = v; }
int train_get(const char *key, int *value)
{
    if (!strcmp(key, "wgrad_wino")) { *value = g_wgrad_wino; return 1; }
    return 0;
}
}  // namespace readhip

namespace {
// Winograd-domain wgrad: 3x3 / stride 1, whole 32-channel tiles of input channels, whole 4 x 4 tiles of pixels
bool wgrad_uses_wino4(int Cin, int ksize, int stride, int H, int W)
{
    return readhip::g_wgrad_wino && ksize == 3 && stride == 1 && Cin % 32 == 0 && H % 4 == 0 && W % 4 == 0 && H >= 4 && W >= 4;
}
struct Wgrad4Plan {
    int tiles_ci, tiles_co, splits, rows_per_split;
    size_t partial_floats;
};
Wgrad4Plan wgrad4_plan(int Cin, int Cout, int H)
{
    Wgrad4Plan p;
    const int Cp = (Cout + 7) / 8 * 8;
    p.tiles_ci = Cin / 32;
    p.tiles_co = (2 * Cp + 31) / 32;
    const int wgs = p.tiles_ci * p.tiles_co, tiles_y = H / 4;
    // one workgroup per CU over the chip (measured, training it/s: 256 workgroups 25.7, 512: 25.5, 768: 24.9, 1024: 24.8, 2048: 23.8 —
    // fewer splits mean fewer partials to write and sum); knob values > 1: workgroups aimed at / 128
    const int target = readhip::g_wgrad_wino > 1 ? 128 * readhip::g_wgrad_wino : 256;
    int splits = (target + wgs - 1) / wgs;
    if (splits > tiles_y) splits = tiles_y;
    if (splits < 1) splits = 1;
    p.rows_per_split = (tiles_y + splits - 1) / splits;
    p.splits = (tiles_y + p.rows_per_split - 1) / p.rows_per_split;
    p.partial_floats = (size_t)p.splits * wgs * 36 * 1024;
    return p;
}

struct WgradPlan {
    int NT, tap_groups, tiles_ci, tiles_co, splits, rows_per_split;
    int cib;                   // 1x1 layers: NT tiles of input channels per wave (tiles_ci then counts blocks of NT tiles)
    size_t partial_floats;
};
WgradPlan wgrad_plan(int Cin, int Cout, int ksize, int outH)
{
    WgradPlan p;
    const int Cp = (Cout + 7) / 8 * 8, taps = ksize * ksize;
    p.NT = taps == 1 ? 1 : (taps == 9 ? 9 : 8);
    p.tap_groups = (taps + p.NT - 1) / p.NT;
    p.tiles_ci = (Cin + 31) / 32;
    p.cib = 0;
    if (taps == 1 && Cin % 32 == 0 && p.tiles_ci > 1) {
        p.NT = p.tiles_ci % 5 == 0 ? 5 : (p.tiles_ci % 4 == 0 ? 4 : (p.tiles_ci % 2 == 0 ? 2 : 1));
        p.cib = p.NT > 1;
        p.tiles_ci /= p.NT;
    }
    p.tiles_co = (2 * Cp + 31) / 32;
    const int waves = p.tiles_ci * p.tiles_co * p.tap_groups;
    int splits = (2048 + waves - 1) / waves;                    // ~2 waves per SIMD over the chip
    if (splits > outH) splits = outH;
    if (splits < 1) splits = 1;
    p.rows_per_split = (outH + splits - 1) / splits;
    p.splits = (outH + p.rows_per_split - 1) / p.rows_per_split;
    p.partial_floats = (size_t)p.splits * waves * p.NT * 1024;
    return p;
}
}  // namespace

extern "C" size_t read_conv_wgrad_scratch_floats(int Cin, int Cout, int ksize, int outH)
{
    if (Cin < 1 || Cout < 1 || outH < 1) return 0;
    // the larger of the two plans a 3x3 layer may take (the caller does not pass the stride or the width)
    const size_t direct = wgrad_plan(Cin, Cout, ksize, outH).partial_floats;
    const size_t wino = (ksize == 3 && Cin % 32 == 0 && outH % 4 == 0) ? wgrad4_plan(Cin, Cout, outH).partial_floats : 0;
    return direct > wino ? direct : wino;
}

extern "C" int read_conv_wgrad_family(int Cin, int ksize, int stride, int inH, int inW)
{
    return wgrad_uses_wino4(Cin, ksize, stride, inH, inW) ? 4 : 0;
}

extern "C" int read_conv_wgrad(const float *x, int inH, int inW, int Cin, const float *dfm, int Cout, int ksize, int stride,
                               float *dwf, float *dwm, int accumulate, float *scratch, size_t scratch_floats, void *stream)
{
    READ_CHECK_ARG(x && dfm && dwf && dwm && scratch, "read_conv_wgrad: null pointer");
    READ_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 4, "read_conv_wgrad: ksize must be 1, 3 or 4");
    READ_CHECK_ARG(stride == 1 || stride == 2, "read_conv_wgrad: stride must be 1 or 2");
    const int pad = (ksize - 1) / 2, Cp = (Cout + 7) / 8 * 8;
    const int outH = (inH + 2 * pad - ksize) / stride + 1, outW = (inW + 2 * pad - ksize) / stride + 1;
    READ_CHECK_ARG(outH >= 1 && outW >= 1, "read_conv_wgrad: empty output");
    // the kernel addresses both tensors with 32-bit byte offsets (plus the largest tap offset)
    READ_CHECK_ARG(((long long)inH + ksize) * inW * Cin * 4 < (1ll << 31) && (long long)outH * outW * 2 * Cp * 4 < (1ll << 31),
                   "read_conv_wgrad: tensors of 2 GiB and more are not supported");
    if (wgrad_uses_wino4(Cin, ksize, stride, inH, inW)) {
        const Wgrad4Plan q = wgrad4_plan(Cin, Cout, inH);
        READ_CHECK_ARG(scratch_floats >= q.partial_floats, "read_conv_wgrad: scratch %zu < %zu floats", scratch_floats, q.partial_floats);
        hipLaunchKernelGGL(wgrad_wino4_kernel, dim3((unsigned)(q.tiles_ci * q.tiles_co), 1, (unsigned)q.splits), dim3(256), 0,
                           as_stream(stream), x, inH, inW, Cin, dfm, 2 * Cp, q.tiles_co, q.rows_per_split, scratch);
        READ_CHECK_LAUNCH();
        const long long per_split = (long long)q.tiles_ci * q.tiles_co * 36 * 1024;
        if (q.splits > 1) {
            hipLaunchKernelGGL(wgrad_wino4_sum_kernel, dim3(grid_for(per_split)), dim3(256), 0, as_stream(stream), scratch, q.splits, per_split);
            READ_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(wgrad_wino4_reduce_kernel, dim3(grid_for((long long)q.tiles_ci * q.tiles_co * 1024)), dim3(256), 0,
                           as_stream(stream), (const float *)scratch, q.tiles_ci, q.tiles_co, Cin, Cout, Cp, dwf, dwm, accumulate);
        READ_CHECK_LAUNCH();
        return READ_OK;
    }
    const WgradPlan p = wgrad_plan(Cin, Cout, ksize, outH);
    READ_CHECK_ARG(scratch_floats >= p.partial_floats, "read_conv_wgrad: scratch %zu < %zu floats", scratch_floats, p.partial_floats);
    const dim3 grid((unsigned)(p.tiles_ci * p.tiles_co), (unsigned)p.tap_groups, (unsigned)p.splits);
    auto kern = p.cib ? (p.NT == 5 ? wgrad_mfma_kernel<5, true> : (p.NT == 4 ? wgrad_mfma_kernel<4, true> : wgrad_mfma_kernel<2, true>))
                      : (p.NT == 9 ? wgrad_mfma_kernel<9, false> : (p.NT == 8 ? wgrad_mfma_kernel<8, false> : wgrad_mfma_kernel<1, false>));
    hipLaunchKernelGGL(kern, grid, dim3(64), 0, as_stream(stream), x, inH, inW, Cin, dfm, outH, outW, 2 * Cp, ksize, stride,
                       p.tiles_co, p.rows_per_split, scratch);
    READ_CHECK_LAUNCH();
    const long long total = (long long)p.tap_groups * p.tiles_ci * p.tiles_co * p.NT * 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float *)scratch,
                       p.splits, p.tap_groups, p.NT, p.tiles_ci, p.tiles_co, Cin, Cout, Cp, ksize * ksize, dwf, dwm, accumulate, p.cib);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_bilinear_up4_blocks(const float *in, int inH, int inW, int C, float *out, int block_h, int valid_h,
                                        void *stream)
{
    READ_CHECK_ARG(in && out && inH >= 1 && inW >= 1, "read_bilinear_up4_blocks: null pointer or empty input");
    READ_CHECK_ARG(C >= 4 && C % 4 == 0, "read_bilinear_up4_blocks: C must be a multiple of 4");
    READ_CHECK_ARG(block_h == 0 || (valid_h >= 1 && valid_h <= block_h && inH % block_h == 0), "read_bilinear_up4_blocks: bad block geometry");
    hipLaunchKernelGGL(bilinear_up4_blocks_kernel, dim3(grid_for((long long)inH * 4 * inW * 4 * (C / 4))), dim3(256), 0,
                       as_stream(stream), in, inH, inW, C, out, block_h, valid_h);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_bilinear_up4_backward(const float *dout, int inH, int inW, int C, float *din, int block_h, int valid_h,
                                          void *stream)
{
    READ_CHECK_ARG(dout && din && inH >= 1 && inW >= 1, "read_bilinear_up4_backward: null pointer or empty input");
    READ_CHECK_ARG(C >= 4 && C % 4 == 0, "read_bilinear_up4_backward: C must be a multiple of 4");
    READ_CHECK_ARG(block_h == 0 || (valid_h >= 1 && valid_h <= block_h && inH % block_h == 0), "read_bilinear_up4_backward: bad block geometry");
    hipLaunchKernelGGL(bilinear_up4_backward_kernel, dim3(grid_for((long long)inH * inW * (C / 4))), dim3(256), 0,
                       as_stream(stream), dout, inH, inW, C, din, block_h, valid_h);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_huber_loss(const float *out, const float *target, int64_t n, float grad_scale, float *loss_sum, float *grad,
                               void *stream)
{
    READ_CHECK_ARG(out && target && n >= 1 && (loss_sum || grad), "read_huber_loss: null pointer or empty tensor");
    if (loss_sum) READ_CHECK_HIP(hipMemsetAsync(loss_sum, 0, sizeof(float), as_stream(stream)));
    hipLaunchKernelGGL(huber_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, as_stream(stream), out, target, (long long)n,
                       grad_scale, loss_sum, grad);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// The same update from the step's (pixel id, gradient row) pairs SORTED by id (torch.sort): the head of every run of equal ids
// sums the run's rows in sorted order — deterministic, no atomics, no N x C gradient table — and applies the update.  The run
// length comes from a binary search; runs above LONG_RUN pairs (the background id 0 collects every empty pixel) are queued
// for rmsprop_long_kernel, one workgroup each.  (Scatter-adding into the dense table first cost 10.5 ms per iteration:
// 5.6 M random fp32 atomics into 320 MB.)
constexpr int LONG_RUN = 512, MAX_LONG = 64;

__device__ __forceinline__ void rmsprop_row(float *rows, float *sq, int *stamp, int C, long long id, const float *g, int step,
                                            float lr, float alpha, float eps)
{
    const int last = stamp[id];
    stamp[id] = step;
    const float decay = powf(alpha, (float)(step - 1 - last));      // steps in which the dense optimizer saw g = 0
    for (int c = 0; c < C; ++c) {
        const float v = alpha * (sq[id * C + c] * decay) + (1.0f - alpha) * g[c] * g[c];
        sq[id * C + c] = v;
        rows[id * C + c] -= lr * g[c] / (sqrtf(v) + eps);
    }
}

constexpr int RMS_MAX_C = 16;

__global__ __launch_bounds__(256) void rmsprop_sorted_kernel(float *__restrict__ rows, float *__restrict__ sq, int *__restrict__ stamp,
                                                             int C, long long n_rows, const int32_t *__restrict__ sorted_ids,
                                                             const long long *__restrict__ perm, const float *__restrict__ g,
                                                             long long n, int step, float lr, float alpha, float eps,
                                                             int *__restrict__ long_list)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long id = sorted_ids[i];
    if (i > 0 && sorted_ids[i - 1] == id) return;                   // not the head of its run
    if (id < 0 || id >= n_rows) return;
    long long lo = i + 1, hi = n;                                   // first index past the run
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (sorted_ids[mid] == id) lo = mid + 1;
        else hi = mid;
    }
    const long long len = lo - i;
    if (len > LONG_RUN) {
        const int slot = atomicAdd(long_list, 1);
        if (slot < MAX_LONG) {
            long_list[1 + 3 * slot] = (int)id;
            long_list[2 + 3 * slot] = (int)i;
            long_list[3 + 3 * slot] = (int)len;
            return;
        }
    }
    float acc[RMS_MAX_C];
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (long long j = i; j < lo; ++j) {
        const float *row = g + perm[j] * C;
        for (int c = 0; c < C; ++c) acc[c] += row[c];
    }
    rmsprop_row(rows, sq, stamp, C, id, acc, step, lr, alpha, eps);
}

// long runs: LONG_SUB workgroups each sum a slice of the run (whole rows, fixed tree), the final kernel adds the slices in order
constexpr int LONG_SUB = 64;

__global__ __launch_bounds__(256) void rmsprop_long_partial_kernel(int C, const long long *__restrict__ perm, const float *__restrict__ g,
                                                                   const int *__restrict__ long_list, float *__restrict__ partial)
{
    __shared__ float red[256];
    const int count = long_list[0] < MAX_LONG ? long_list[0] : MAX_LONG;
    const int seg = blockIdx.y, sub = blockIdx.x;
    if (seg >= count) return;
    const long long start = long_list[2 + 3 * seg], len = long_list[3 + 3 * seg];
    const long long per = (len + LONG_SUB - 1) / LONG_SUB, lo = sub * per, hi = lo + per < len ? lo + per : len;
    float acc[RMS_MAX_C];
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (long long j = lo + threadIdx.x; j < hi; j += 256) {
        const float *row = g + perm[start + j] * C;
        for (int c = 0; c < C; ++c) acc[c] += row[c];
    }
    for (int c = 0; c < C; ++c) {
        red[threadIdx.x] = acc[c];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[((size_t)seg * LONG_SUB + sub) * RMS_MAX_C + c] = red[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void rmsprop_long_final_kernel(float *__restrict__ rows, float *__restrict__ sq, int *__restrict__ stamp,
                                                                int C, int step, float lr, float alpha, float eps,
                                                                const int *__restrict__ long_list, const float *__restrict__ partial)
{
    __shared__ float total[RMS_MAX_C];
    const int count = long_list[0] < MAX_LONG ? long_list[0] : MAX_LONG;
    if ((int)blockIdx.x >= count) return;
    if ((int)threadIdx.x < C) {
        float s = 0.0f;
        for (int k = 0; k < LONG_SUB; ++k) s += partial[((size_t)blockIdx.x * LONG_SUB + k) * RMS_MAX_C + threadIdx.x];
        total[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) rmsprop_row(rows, sq, stamp, C, long_list[1 + 3 * blockIdx.x], total, step, lr, alpha, eps);
}

// scratch = [count, MAX_LONG x (id, start, length)] as int32, padded to 256 words, then MAX_LONG x LONG_SUB x RMS_MAX_C floats
constexpr size_t LONG_LIST_WORDS = 256;
static_assert(1 + 3 * MAX_LONG <= LONG_LIST_WORDS, "long-run list does not fit its scratch region");
extern "C" size_t read_rmsprop_sorted_scratch_ints(void) { return LONG_LIST_WORDS + (size_t)MAX_LONG * LONG_SUB * RMS_MAX_C; }

extern "C" int read_rmsprop_sorted(float *rows, float *sq, int32_t *stamp, int C, int64_t n_rows, const int32_t *sorted_ids,
                                   const int64_t *perm, const float *g, int64_t n, int step, float lr, float alpha, float eps,
                                   int32_t *scratch, void *stream)
{
    READ_CHECK_ARG(rows && sq && stamp && sorted_ids && perm && g && scratch, "read_rmsprop_sorted: null pointer");
    READ_CHECK_ARG(C >= 1 && C <= RMS_MAX_C && n_rows >= 1 && n >= 0 && step >= 1, "read_rmsprop_sorted: bad sizes / step");
    if (n == 0) return READ_OK;
    READ_CHECK_HIP(hipMemsetAsync(scratch, 0, sizeof(int), as_stream(stream)));
    hipLaunchKernelGGL(rmsprop_sorted_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream), rows, sq, stamp, C,
                       (long long)n_rows, sorted_ids, (const long long *)perm, g, (long long)n, step, lr, alpha, eps, scratch);
    READ_CHECK_LAUNCH();
    float *partial = reinterpret_cast<float *>(scratch + LONG_LIST_WORDS);
    hipLaunchKernelGGL(rmsprop_long_partial_kernel, dim3(LONG_SUB, MAX_LONG), dim3(256), 0, as_stream(stream), C,
                       (const long long *)perm, g, (const int *)scratch, partial);
    READ_CHECK_LAUNCH();
    hipLaunchKernelGGL(rmsprop_long_final_kernel, dim3(MAX_LONG), dim3(64), 0, as_stream(stream), rows, sq, stamp, C, step, lr, alpha,
                       eps, (const int *)scratch, (const float *)partial);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_rmsprop_sparse(float *rows, float *sq, float *grad, int32_t *stamp, int C, int64_t n_rows,
                                   const int32_t *ids, int64_t n_ids, int step, float lr, float alpha, float eps, void *stream)
{
    READ_CHECK_ARG(rows && sq && grad && stamp && ids, "read_rmsprop_sparse: null pointer");
    READ_CHECK_ARG(C >= 1 && n_rows >= 1 && n_ids >= 0 && step >= 1, "read_rmsprop_sparse: bad sizes / step");
    if (n_ids == 0) return READ_OK;
    hipLaunchKernelGGL(rmsprop_sparse_kernel, dim3(grid_for(n_ids)), dim3(256), 0, as_stream(stream), rows, sq, grad, stamp, C,
                       (long long)n_rows, ids, (long long)n_ids, step, lr, alpha, eps);
    READ_CHECK_LAUNCH();
    return READ_OK;
}
