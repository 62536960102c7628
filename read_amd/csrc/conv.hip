// Gated convolution (READ's BasicConv, READ/models/unet.py:22-53) as ONE fused implicit-GEMM
// kernel on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
//   out = BN_eval( act(conv_f(x) + b_f) * sigmoid(conv_m(x) + b_m) ) [+ residual]
//
// GEMM view per output tile:  D[pixel][cout] = sum_k A[pixel][k] * B[k][cout],  k = (tap, cin).
//   * rows (M) = 32 consecutive output pixels of one image row  -> MFMA rows
//   * cols (N) = 32 output channels; conv_f and conv_m of the SAME 32 channels are two
//     accumulators with identical lane mapping, so the gate product is lane-local
//   * A is staged per cin-chunk (KC channels) as an NHWC halo tile in LDS, padded to KC+4 floats per
//     pixel: ds_read_b128 of 32 consecutive pixels is then bank-conflict free
//   * B (weights) is pre-packed on the host in exact fragment order, so a wave's B fragment is one
//     contiguous 1 KiB global_load_dwordx4 (L2 resident: weights are shared by every workgroup)
//   * torch.cat, nearest F.interpolate and FAM's x1*x2 are folded into the A-tile loader
//     (multi-source addressing with a per-source power-of-two shift); the ResBlock / FAM
//     residual add, both biases, ELU, sigmoid, gate and eval-mode BatchNorm live in the epilogue.
//
// MFMA 32x32x2 f32 layouts (cdna_hip_programming.md §3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; D[i][j] sits in lane (j + 32*((i>>2)&1)), register (i&3) + 4*(i>>3).
// A k-step of 8 is four MFMAs fed from ONE float4 per operand: lanes <32 carry cin 0..3 of the
// 8-block, lanes >=32 carry cin 4..7 (the k order inside a step is free as long as A and B agree).
#include "common.h"

using namespace readhip;

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr long long OOB_LIMIT = 0x7ffffff0ll;   // byte offsets below this are in range for the epilogue's buffer ops

struct SrcDev {
    const float *p;
    int C, H, W;
    int sl, sr;          // source coordinate = (dst << sl) >> sr
};

struct ConvKArgs {
    SrcDev src[READ_CONV_MAX_SRC];
    const float *mul;
    const float *wp;
    const float *wp_wino;          // Winograd-transformed weights (read_conv_pack_wino_host) or null
    const float *wp_w16;           // the same weights in the order of the wave-autonomous kernel (read_conv_pack_w16_host) or null
    const float *wp_w4;            // Winograd F(4x4,3x3) weights (read_conv_pack_w4_host) or null
    const void *wp_w4h;            // ... split into f16 piece pairs + row scales (read_conv_pack_w4h_host) or null
    const void *wp_d3h;            // the plain 3x3 weights as f16 piece pairs + row scales (read_conv_pack_d3h_host) or null
    const float *wp_sc;            // [tap][cin][f0..f3 | m0..m3] of a layer with at most four output channels (read_conv_pack_sc_host) or null
    const float *params;
    const float *residual;
    float *out;
    int inH, inW, outH, outW;
    int Cout, CoutPad, out_cstride;
    int nchunks, tiles_x, n_src;
    int tiles_y, n_units;          // wave-autonomous kernel: units = (group set, x tile, y tile)
    int ablate;                    // debug ablation bits: 1 no epilogue loads/stores, 2 no A restaging, 4 B loads pinned to
                                   //   step 0, 8 no MFMAs (results invalid; tools/ablate_conv.py)
    int stagger_ticks;             //   start delay (10 ns ticks) of waves in odd hardware slots: de-phases SIMD partners
    int n_full, col_split;         //   units [0,n_full) are P-row units of columns [0,col_split); the rest are
                                   //   1-row units of the remaining (group set, x tile) columns (balanced tail)
    int wino_dby, wino_dbx;        // Winograd kernel: (tile row, tile column) step between a workgroup's units
    int elu, fill_pad;
    int linear;                    // 1: plain convolution — store conv_f + b_f at channel c and conv_m + b_m at channel Cout + c
                                   //    (no gate, no BatchNorm, no residual): the training path's pre-activations and dgrad
    float out_fill;
    // optional pre-activation addend (read_conv_desc.pre): conv_f += pre[(oy >> s, ox >> s)][pre_foff + c], conv_m likewise
    // with pre_moff — partial sums of the same layer computed at a coarser level (nearest up-sampling commutes with a 1x1 conv)
    const float *pre;
    int pre_cstride, pre_foff, pre_moff, pre_shift, pre_W, pre_bytes;
    int pre_bil, pre_H;            // 1: the addend is the 4x BILINEAR up-sampling of `pre` (read_conv_desc.pre_bilinear; pixel-lane kernel)
    // linear launches of the training path (Winograd kernel): also store the gated output BN(act(f) * sigmoid(m)) here, zero on
    // the separator rows of a stacked batch (rows r with r % blk_h >= blk_valid)
    float *out_gated;
    int blk_h, blk_valid;
    unsigned long long *trace;     // optional timeline: 8 x u64 per workgroup (read_debug_set_trace)
};

// Epilogue transcendentals on the hardware v_exp_f32 / v_rcp_f32 (about 1 ulp each): the gate and
// the ELU tail are O(1) quantities, so the absolute error stays ~1e-7 — far inside the CNN tolerance
// (tests/test_gpu_conv.py) — while the epilogue VALU cost drops ~4x against the ocml expf/expm1f/div.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : fast_exp(x) - 1.0f; }
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }


// ------------------------------------------------------------------------------------------
// The k-steps of one input-channel chunk for one wave, hand-interleaved.
//
// A wave issues in order and an MFMA blocks it until the matrix pipe accepts it, so whatever follows a
// block of back-to-back MFMAs runs while the pipe drains and then idles it.  The first versions of this
// loop did exactly that ([prefetch loads][16 MFMAs]); an ablation on MI355X (profiles/README.md) showed
// kernel time = MFMA time + time of the MFMA-free skeleton, i.e. zero overlap, and co-resident waves
// phase-lock through round-robin arbitration so they do not cover for each other.  Here every
// prefetch (B fragments of step s+PF from L2, A fragments of step s+1 from LDS) is issued in the
// shadow of an MFMA of step s: one item right after each of the first T+P MFMAs, pinned with
// sched_barrier so hipcc cannot sink them.  Register rings are indexed statically (the loop is fully
// unrolled), so there are no rotation moves.
//   bq[RING][T]: B ring, slot (step % RING); requires SPC % RING == 0 so slots line up across chunks
//   bnext(ls)  : wave-uniform pointer to tile 0 of the step that is PF ahead of local step ls
template <int KS, int S, int IW, int PS, int KK, int P, int T, int PF, int SPC, typename BNext>
__device__ __forceinline__ void chunk_steps(const float *buf, int abase, int tap0, int lane, floatx16 (&acc)[P][T],
                                            float4 (&bq)[PF + 1][T], BNext bnext)
{
    constexpr int RING = PF + 1;
    static_assert(SPC % RING == 0, "B ring slots must line up across chunks");
    float4 aq[2][P];
    auto a_addr = [&](int ls, int p) {
        const int tap = tap0 + ls / KK, kk = ls % KK;
        const int ky = tap / KS, kx = tap % KS;
        return reinterpret_cast<const float4 *>(buf + abase + ((p * S + ky) * IW + kx) * PS + kk * 8);
    };
#pragma unroll
    for (int p = 0; p < P; ++p) aq[0][p] = *a_addr(0, p);
    const float4 *nb_next = bnext(0) + lane;
#pragma unroll
    for (int ls = 0; ls < SPC; ++ls) {
        const int cur = ls % RING, nxt = (ls + PF) % RING, ac = ls & 1, an = ac ^ 1;
        const float4 *nb = nb_next;
        if (PF == 0) {
#pragma unroll
            for (int t = 0; t < T; ++t) bq[0][t] = nb[t * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float av = j == 0 ? aq[ac][p].x : j == 1 ? aq[ac][p].y : j == 2 ? aq[ac][p].z : aq[ac][p].w;
                    const float bv = j == 0 ? bq[cur][t].x : j == 1 ? bq[cur][t].y : j == 2 ? bq[cur][t].z : bq[cur][t].w;
                    acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p][t], 0, 0, 0);
                    const int m = (j * P + p) * T + t;          // one prefetch item in the shadow of MFMA m
                    if (PF > 0 && m < T) bq[nxt][m] = nb[m * 64];
                    else if (m >= T && m < T + P && ls + 1 < SPC) aq[an][m - T] = *a_addr(ls + 1, m - T);
                    else if (m == T + P && ls + 1 < SPC) nb_next = bnext(ls + 1) + lane;   // address math in the shadow too
                    __builtin_amdgcn_sched_barrier(0);
                }
    }
}

template <int KS, int S, int KC, int P, int QG, int WM, int WN>
struct Tile {
    static constexpr int WK = 4 / (WM * WN);          // waves that split the taps of ONE output tile (split-K)
    static constexpr int TPW = KS * KS / WK;          // taps per wave
    static constexpr int NR = 16 / WK;                // accumulator registers a wave finishes in the epilogue
    static constexpr int TH = WM * P;                 // output rows per workgroup
    static constexpr int TW = 32;                     // output columns per workgroup (one MFMA M)
    static constexpr int IH = (TH - 1) * S + KS;      // input halo tile
    static constexpr int IW = (TW - 1) * S + KS;
    static constexpr int PS = KC + 4;                 // padded pixel stride (floats) in LDS
    static constexpr int BUF = IH * IW * PS;          // floats per LDS buffer
    static constexpr int Q4 = KC / 4;                 // float4 per pixel per chunk
    static constexpr int NE = IH * IW * Q4;           // float4 elements per chunk tile
    static constexpr int NI = (NE + 255) / 256;       // staging registers (float4) per thread
    static constexpr int KK = KC / 8;                 // k-steps of 8 per (chunk, tap)
    static constexpr int T = 2 * QG;                  // B tiles (f,m interleaved) per wave
    static constexpr int PAD = (KS - 1) / 2;          // int(dilation*(k-1)/2), unet.py:30
};

template <int KS, int S, int KC, int P, int QG, int WM, int WN, bool MUL, int PF, int NBUF>
__global__ __launch_bounds__(256) void gated_conv_kernel(const ConvKArgs a)
{
    using TL = Tile<KS, S, KC, P, QG, WM, WN>;
    static_assert(WM * WN * TL::WK == 4 && (KS * KS) % TL::WK == 0, "4 waves per workgroup; taps split evenly");
    static_assert(TL::WK == 1 || 4 * P * TL::T * 16 * 64 <= NBUF * TL::BUF, "split-K reduction must fit the LDS tile");
    __shared__ __attribute__((aligned(16))) float lds[NBUF * TL::BUF];

    unsigned long long t_trace[4];
    if (a.trace) t_trace[0] = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN);                  // 0 unless split-K
    const int wm = (wave / WN) % WM, wn = wave % WN;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int ox0 = tx * TL::TW, oy0 = ty * TL::TH;
    const int ix0 = ox0 * S - TL::PAD, iy0 = oy0 * S - TL::PAD;
    const int NT = a.CoutPad >> 4;                                  // 2 tiles per 32 channels
    const int nt0 = ((int)blockIdx.y * WN + wn) * TL::T;

    // Staging registers of the A tile.  The (pixel, quad) a thread stages and its in-image test do
    // not depend on the chunk; nothing consumes a staged value before lwrite(), so the global
    // loads of chunk c+1 stay in flight under the MFMAs of chunk c.
    float4 st[TL::NI];
    float4 sm[MUL ? TL::NI : 1];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < TL::NI; ++i) {
        const int e = tid + i * 256;
        const int pix = e / TL::Q4;
        const int gy = iy0 + pix / TL::IW, gx = ix0 + pix % TL::IW;
        const bool ok = e < TL::NE && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
        okmask |= (ok ? 1u : 0u) << i;
    }

    // (source, channel offset) of the NEXT chunk to stage, advanced with scalar arithmetic.  A table
    // indexed by the chunk number in the kernarg segment turns into a vector load + readfirstlane in
    // front of every chunk's loads (seen in the ISA): ~1 us of exposed latency per chunk.
    int nsrc_i = 0, ncoff = 0;
    SrcDev ns = a.src[0];
    auto gload = [&]() {
        const SrcDev s = ns;
        const int coff = ncoff;
#pragma unroll
        for (int i = 0; i < TL::NI; ++i) {
            const int e = tid + i * 256;
            const int q = e % TL::Q4;
            const int pix = e / TL::Q4;
            const int gy = iy0 + pix / TL::IW, gx = ix0 + pix % TL::IW;
            const int sy = (gy << s.sl) >> s.sr, sx = (gx << s.sl) >> s.sr;
            // out-of-image elements read element 0 (always mapped) and are zeroed in lwrite()
            const int off = ((okmask >> i) & 1u) ? (sy * s.W + sx) * s.C + coff + 4 * q : 0;
            st[i] = *reinterpret_cast<const float4 *>(s.p + off);
            if (MUL) sm[i] = *reinterpret_cast<const float4 *>(a.mul + off);
        }
        ncoff += KC;
        if (ncoff >= s.C && nsrc_i + 1 < a.n_src) {       // uniform: next chunk comes from the next source
            ++nsrc_i;
            ncoff = 0;
            ns = a.src[nsrc_i];
        }
    };
    auto lwrite = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < TL::NI; ++i) {
            const int e = tid + i * 256;
            if (e < TL::NE) {
                const int q = e % TL::Q4;
                const int pix = e / TL::Q4;
                float4 v = st[i];
                if (MUL) { v.x *= sm[i].x; v.y *= sm[i].y; v.z *= sm[i].z; v.w *= sm[i].w; }
                if (!((okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(buf + pix * TL::PS + 4 * q) = v;
            }
        }
    };

    floatx16 acc[P][TL::T];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int t = 0; t < TL::T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.0f;

    // B fragments: float4 index (step*NT + nt)*64 + lane, addressed as (wave-uniform pointer) + lane so the
    // loads take the scalar-base form.  This wave walks local steps ls = (chunk, tap in its TPW-tap share,
    // kk); gstep() maps them to the packed order.  A static ring keeps PF steps in flight.
    const float4 *wtile = reinterpret_cast<const float4 *>(a.wp) + (size_t)nt0 * 64;
    constexpr int SPC = TL::TPW * TL::KK;             // local steps per chunk
    const int total_steps = a.nchunks * SPC;
    auto gstep = [&](int ls) {
        ls = ls < total_steps ? ls : total_steps - 1;
        const int chunk = ls / SPC, rem = ls % SPC;
        return (chunk * KS * KS + wk * TL::TPW + rem / TL::KK) * TL::KK + rem % TL::KK;
    };
    float4 bq[PF + 1][TL::T];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const float4 *bp = wtile + (size_t)gstep(d) * NT * 64 + lane;
#pragma unroll
        for (int t = 0; t < TL::T; ++t) bq[d][t] = bp[t * 64];
    }

    gload();
    lwrite(lds);
    __syncthreads();
    if (a.trace) t_trace[1] = __builtin_amdgcn_s_memrealtime();

    // A fragment base inside a buffer: row (wm*P + p)*S, column (lane&31)*S, cin 4*(lane>>5)
    const int abase = ((wm * P) * S * TL::IW + (lane & 31) * S) * TL::PS + 4 * (lane >> 5);

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const bool more = chunk + 1 < a.nchunks && !(a.ablate & 2);
        if (more) gload();
        const float *buf = lds + (NBUF == 2 ? (chunk & 1) * TL::BUF : 0);
        const int step0 = chunk * SPC;
        chunk_steps<KS, S, TL::IW, TL::PS, TL::KK, P, TL::T, PF, SPC>(
            buf, abase, TL::WK == 1 ? 0 : wk * TL::TPW, lane, acc, bq,
            [&](int ls) { return wtile + (size_t)gstep(step0 + ls + PF) * NT * 64; });
        if (NBUF == 2) {
            if (more) lwrite(lds + ((chunk + 1) & 1) * TL::BUF);
            __syncthreads();
        } else {
            __syncthreads();                      // everyone is done reading the single buffer
            if (more) {
                lwrite(lds);
                __syncthreads();
            }
        }
    }

    if (a.trace) t_trace[2] = __builtin_amdgcn_s_memrealtime();

    // ---------------- split-K: the WK waves of a tile exchange partial sums through LDS; wave wk then
    // finishes accumulator registers [wk*NR, wk*NR+NR) (every tile, every group).
    if (TL::WK > 1) {
        // the chunk loop ended with a barrier: the A tiles are dead, reuse the LDS
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < TL::T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    lds[(((wave * P + p) * TL::T + t) * 16 + r) * 64 + lane] = acc[p][t][r];
        __syncthreads();
    }

    // ---------------- epilogue: bias, ELU, sigmoid gate, BatchNorm(eval), residual, store.
    // Raw buffer loads/stores with the hardware range check do the masking (partial tiles, padded
    // channels): no per-element branches, all residual loads of a tile in flight together, all
    // stores issued back to back.  (The first version branched per element and hipcc put an
    // s_waitcnt vmcnt(0) in front of every store — one memory round trip per output value.)
    constexpr int OOB = 0x7ffffff0;
    const int hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.out, 0, a.outH * a.outW * a.out_cstride * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.residual ? a.residual : a.out), 0, a.residual ? a.outH * a.outW * a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.pre ? a.pre : a.out), 0, a.pre ? a.pre_bytes : 0, 0x00020000);
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int c = ((nt0 >> 1) + g) * 32 + (lane & 31);
        const float bf = a.params[c];
        const float bm = a.params[a.CoutPad + c];
        const float sc = a.params[2 * a.CoutPad + c];
        const float sh = a.params[3 * a.CoutPad + c];
        const bool c_ok = c < a.Cout;
        const bool c_st = c_ok || (a.fill_pad && c < a.out_cstride);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int oy = oy0 + wm * P + p;
            int ooff[TL::NR];
            float rv[TL::NR], pf[TL::NR], pm[TL::NR];
#pragma unroll
            for (int rr = 0; rr < TL::NR; ++rr) {
                const int r = wk * TL::NR + rr;
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool in = (oy < a.outH) & (ox < a.outW);
                const int opix = oy * a.outW + ox;
                ooff[rr] = (in & c_st) ? (opix * a.out_cstride + c) * 4 : OOB;
                const int roff = (in & c_ok) ? (opix * a.Cout + c) * 4 : OOB;
                rv[rr] = (a.ablate & 1) ? 0.0f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, 0, 0));
                if ((a.ablate & 1) && !(in && oy == 0 && ox == 0)) ooff[rr] = OOB;      // one store per tile keeps the math live
                pf[rr] = pm[rr] = 0.0f;
                if (a.pre) {
                    const int poff = (in & c_ok) ? (((oy >> a.pre_shift) * a.pre_W + (ox >> a.pre_shift)) * a.pre_cstride + c) * 4 : OOB;
                    pf[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, poff, a.pre_foff * 4, 0));
                    pm[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, poff, a.pre_moff * 4, 0));
                }
            }
#pragma unroll
            for (int rr = 0; rr < TL::NR; ++rr) {
                const int r = wk * TL::NR + rr;
                float f, m;
                if (TL::WK > 1) {
                    f = m = 0.0f;
#pragma unroll
                    for (int w = 0; w < TL::WK; ++w) {
                        f += lds[(((w * P + p) * TL::T + 2 * g) * 16 + r) * 64 + lane];
                        m += lds[(((w * P + p) * TL::T + 2 * g + 1) * 16 + r) * 64 + lane];
                    }
                } else {
                    f = acc[p][2 * g][rr];            // WK == 1: r == rr
                    m = acc[p][2 * g + 1][rr];
                }
                f += bf + pf[rr];
                m += bm + pm[rr];
                if (a.linear) {
                    // ooff addresses channel c of the first half; the second half starts Cout channels later
                    const int o2 = (ooff[rr] != OOB && c_ok) ? ooff[rr] + a.Cout * 4 : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f), orsrc, c_ok ? ooff[rr] : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), orsrc, o2, 0, 0);
                    continue;
                }
                if (a.elu) f = elu1(f);
                float v = (f * sigmoidf(m)) * sc + sh + rv[rr];
                v = c_ok ? v : a.out_fill;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, ooff[rr], 0, 0);
            }
        }
    }
    if (a.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // include this wave's stores
        t_trace[3] = __builtin_amdgcn_s_memrealtime();
        unsigned long long *rec = a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        rec[0] = t_trace[0];
        rec[1] = t_trace[1];
        rec[2] = t_trace[2];
        rec[3] = t_trace[3];
        rec[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID (wave/simd/cu/sh/se)
        rec[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID
        rec[6] = blockIdx.x;
        rec[7] = blockIdx.y;
    }
}


// ------------------------------------------------------------------------------------------
// Wave-autonomous persistent variant (the default for 3x3/s1 and 1x1 layers).
//
// The workgroup-tiled kernel above runs its k-loop at ~95 % MFMA occupancy, but a timeline trace
// (tools/trace_conv.py, profiles/README.md) showed a whole launch at only 60 %: every workgroup
// exposes its prologue (first tile load) and epilogue, the barrier couples its four waves, and the
// last round of workgroups leaves most CUs idle (work is quantised in 4- or 8-unit workgroups).
// Here every WAVE is an independent worker:
//   * unit of work = P image rows x 32 pixels x QG channel groups; a wave owns a private LDS tile
//     (no s_barrier anywhere) and walks units  u = wave_id, wave_id + n_waves, ...  (persistent grid);
//   * the stream of (unit, chunk) items is software-pipelined ACROSS units: while chunk c is being
//     multiplied, chunk c+1 (possibly the next unit's first) is in registers on its way from L2, and
//     the B-fragment ring also runs ahead into the next unit — a unit's epilogue stores and the next
//     unit's first loads overlap the MFMAs of the co-resident waves;
//   * consecutive wave ids take vertically adjacent tiles of the same channel group, so their halos
//     and weight fragments meet in L1/L2.
template <int KS, int S, int KC, int P, int QG, int PF>
__global__ __launch_bounds__(256, (P * QG >= 2 ? 2 : 3)) void gated_conv_wave_kernel(const ConvKArgs a)
{
    constexpr int IH = (P - 1) * S + KS, IW = 31 * S + KS, PS = KC + 4, BUF = IH * IW * PS;
    constexpr int Q4 = KC / 4, NE = IH * IW * Q4, NI = (NE + 63) / 64, KK = KC / 8, T = 2 * QG;
    constexpr int PAD = (KS - 1) / 2, SPC = KS * KS * KK;
    __shared__ __attribute__((aligned(16))) float lds_all[4 * BUF];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *lds = lds_all + wave * BUF;
    const int gw = (int)blockIdx.x * 4 + wave;
    const int nw = (int)gridDim.x * 4;
    const int NT = a.CoutPad >> 4;
    const int total_steps = a.nchunks * SPC;
    if (gw >= a.n_units) return;
    if (a.stagger_ticks > 0) {
        // Waves sharing a SIMD do identical work and would stay in lockstep for the whole launch: both
        // in their epilogue (MFMA pipe idle) at the same moments.  Delay the odd hardware wave slot once;
        // its partner runs alone at full MFMA rate meanwhile, so nothing is lost.
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID, WAVE_ID = bits 3:0
        if (hw & 1u) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.stagger_ticks) __builtin_amdgcn_s_sleep(20);
        }
    }

    // unit -> tile origin, channel-group set and number of valid rows (P for full units, 1 for tail units)
    auto unit_coords = [&](int u, int &ox0, int &oy0, int &gs, int &nrows) {
        int col;
        if (u < a.n_full) {
            col = u / a.tiles_y;
            oy0 = (u % a.tiles_y) * P;
            nrows = P;
        } else {
            const int v = u - a.n_full;
            col = a.col_split + v / a.outH;
            oy0 = v % a.outH;
            nrows = 1;
        }
        ox0 = (col % a.tiles_x) * 32;
        gs = col / a.tiles_x;
    };

    // ---- staging cursor: which (unit, chunk) goes into the staging registers next
    int lu = gw, lc = 0, nsrc_i = 0, ncoff = 0;
    SrcDev ns = a.src[0];
    float4 st[NI];
    unsigned okmask = 0;
    bool staged = false;
    auto gload = [&]() {
        int ox0, oy0, gs, nr;
        unit_coords(lu, ox0, oy0, gs, nr);
        const int ix0 = ox0 * S - PAD, iy0 = oy0 * S - PAD;
        const SrcDev s = ns;
        const int coff = ncoff;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = lane + i * 64;
            const int q = e % Q4, pix = e / Q4;
            const int gy = iy0 + pix / IW, gx = ix0 + pix % IW;
            const bool ok = e < NE && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
            okmask |= (ok ? 1u : 0u) << i;
            const int sy = (gy << s.sl) >> s.sr, sx = (gx << s.sl) >> s.sr;
            const int off = ok ? (sy * s.W + sx) * s.C + coff + 4 * q : 0;
            st[i] = *reinterpret_cast<const float4 *>(s.p + off);
        }
        staged = true;
        ncoff += KC;
        if (ncoff >= s.C && nsrc_i + 1 < a.n_src) {
            ++nsrc_i;
            ncoff = 0;
            ns = a.src[nsrc_i];
        }
        if (++lc == a.nchunks) {        // next item belongs to this wave's next unit
            lc = 0;
            lu += nw;
            nsrc_i = 0;
            ncoff = 0;
            ns = a.src[0];
        }
    };
    auto lwrite = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = lane + i * 64;
            if (e < NE) {
                float4 v = st[i];
                if (!((okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(lds + (e / Q4) * PS + 4 * (e % Q4)) = v;
            }
        }
        staged = false;
    };

    const float4 *wp4base = reinterpret_cast<const float4 *>(a.wp);
    const int abase = ((lane & 31) * S) * PS + 4 * (lane >> 5);
    constexpr int OOB = 0x7ffffff0;
    const int hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.out, 0, a.outH * a.outW * a.out_cstride * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.residual ? a.residual : a.out), 0, a.residual ? a.outH * a.outW * a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.pre ? a.pre : a.out), 0, a.pre ? a.pre_bytes : 0, 0x00020000);

    // ---- prologue: first item into LDS, second into registers, B ring of the first unit
    gload();
    lwrite();
    if (lu < a.n_units) gload();
    float4 bq[PF + 1][T];
    {
        int ox0, oy0, gs, nr;
        unit_coords(gw, ox0, oy0, gs, nr);
        const float4 *wl = wp4base + (size_t)gs * T * 64;
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int sidx = d < total_steps ? d : total_steps - 1;
#pragma unroll
            for (int t = 0; t < T; ++t) bq[d][t] = wl[((size_t)sidx * NT + t) * 64 + lane];
        }
    }

    for (int u = gw; u < a.n_units; u += nw) {
        int ox0, oy0, gs, nrows, nox0, noy0, ngs, nnr;
        unit_coords(u, ox0, oy0, gs, nrows);
        unit_coords(u + nw < a.n_units ? u + nw : u, nox0, noy0, ngs, nnr);
        const float4 *wl = wp4base + (size_t)gs * T * 64;
        const float4 *wl_next = wp4base + (size_t)ngs * T * 64;   // the B ring runs ahead into the next unit

        floatx16 acc[P][T];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.0f;

        for (int chunk = 0; chunk < a.nchunks; ++chunk) {
            const int step0 = chunk * SPC;
            chunk_steps<KS, S, IW, PS, KK, P, T, PF, SPC>(lds, abase, 0, lane, acc, bq, [&](int ls) {
                int nstep = step0 + ls + PF;
                const float4 *base = wl;
                if (nstep >= total_steps) {          // ring runs ahead into this wave's next unit
                    nstep -= total_steps;
                    base = wl_next;
                }
                return base + (size_t)nstep * NT * 64;
            });
            // this wave's reads of the tile are done (their data fed the MFMAs above): restage
            if (staged) {
                lwrite();
                if (lu < a.n_units) gload();
            }
        }

        // ---- epilogue of this unit (same math as the workgroup kernel; the next unit's first chunk is
        // already in LDS and its second on the way, so these loads/stores overlap real work)
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const int c = (gs * QG + g) * 32 + (lane & 31);
            const float bf = a.params[c];
            const float bm = a.params[a.CoutPad + c];
            const float sc = a.params[2 * a.CoutPad + c];
            const float sh = a.params[3 * a.CoutPad + c];
            const bool c_ok = c < a.Cout;
            const bool c_st = c_ok || (a.fill_pad && c < a.out_cstride);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int oy = oy0 + p;
                if (p > 0 && p >= nrows) continue;
                int ooff[16];
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool in = (oy < a.outH) & (ox < a.outW);
                    const int opix = oy * a.outW + ox;
                    ooff[r] = (in & c_st) ? (opix * a.out_cstride + c) * 4 : OOB;
                    const int roff = (in & c_ok) ? (opix * a.Cout + c) * 4 : OOB;
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, 0, 0));
                }
                if (a.pre) {                               // pre-activation addend: straight into the accumulators
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool in = (oy < a.outH) & (ox < a.outW) & c_ok;
                        const int poff = in ? (((oy >> a.pre_shift) * a.pre_W + (ox >> a.pre_shift)) * a.pre_cstride + c) * 4 : OOB;
                        acc[p][2 * g][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, poff, a.pre_foff * 4, 0));
                        acc[p][2 * g + 1][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prsrc, poff, a.pre_moff * 4, 0));
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float f = acc[p][2 * g][r] + bf;
                    const float m = acc[p][2 * g + 1][r] + bm;
                    if (a.elu) f = elu1(f);
                    float v = (f * sigmoidf(m)) * sc + sh + rv[r];
                    v = c_ok ? v : a.out_fill;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, ooff[r], 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // one 16-value batch at a time: bounds the live registers
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) variant of the 3x3 / stride-1 gated conv (single source, Cin % 16 == 0).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A      per 2x2 output tile, summed over input channels before A^T..A:
//   16 independent [tiles x Cin] x [Cin x Cout] contractions instead of 9 taps -> 2.25x fewer MFMAs, still
//   exact-fp32 MFMA arithmetic (the transforms only add/subtract; G g G^T is folded into the packed weights).
//
// Workgroup = 4 waves = the 4 frequency ROWS i of one block of 32 tiles (4 x 8 tiles = 8 x 16 output pixels)
// and one 32-channel group; MFMA row = tile, MFMA column = output channel, k = input channel.
//   * the 10 x 18-pixel input patch of a 16-channel chunk is staged in LDS for all four waves, three buffers,
//     fetched two chunks ahead (so the transform of the next chunk's first k-step can run under this chunk's
//     MFMAs and the in-order vmcnt never makes a B wait cover a fresh patch load);
//   * wave i builds its A fragments on the fly: t[c] = d[ra][c] +- d[rb][c] (row pair of B^T for its i; row 2 is
//     taken negated, the sign lives in the packed weights), then the four column combinations V[i][0..3] —
//     8 ds_read_b128 + 32 VALU per 8-channel k-step, which feeds 32 MFMAs (4 frequencies x {conv_f, conv_m} x 4);
//   * B = transformed weights, packed [group][k8 step][row i][j][f|m][lane][4]: a wave reads 8 KiB per k-step,
//     contiguous; prefetched half a k-step (4 fragments) ahead;
//   * every non-MFMA instruction of the loop is an "item" issued right after one MFMA and pinned there
//     (see chunk_steps above for why): B loads after MFMAs 0-3 of each half-step, patch loads, LDS reads,
//     transform arithmetic and LDS writes after the others;
//   * output: each wave reduces its row over j with A^T (R_b = sum_j A^T[b][j] M[i][j]), the four rows meet in LDS
//     (64 KiB, aliasing the dead input buffers), wave w finishes accumulator registers 4w..4w+3: Y[a][b] =
//     sum_i A^T[a][i] R_b(i), then the usual gated epilogue and 2x2-pixel stores.
// Index maps are mirrored in tests/wino_ref.py (NumPy model checked against conv2d on the CPU).
struct WinoGeom {
    static constexpr int TR = 4, TC = 8;                 // tiles per block (rows, cols) -> 32 = one MFMA M
    static constexpr int IH = 2 * TR + 2, IW = 2 * TC + 2;   // 10 x 18 input pixels
    static constexpr int KC = 16, PS = KC + 4, BUF = IH * IW * PS;
    static constexpr int NE = IH * IW * (KC / 4), NI = (NE + 255) / 256;
    static constexpr int RED = 4 * 2 * 16 * 64;          // floats of the cross-wave reduction of one of {f, m} (32 KiB)
    static constexpr int LDS_FLOATS = 3 * BUF + 4 + RED; // three patch buffers, a dummy float4 slot, the reduction
};

// (scalar base + 32-bit lane offset) load: hipcc selects the saddr form global_load_dwordx4 v, voff, s[base:base+1],
// no per-load 64-bit VALU address math.  (raw_buffer_load_b64/b128 builtins of this ROCm load a single dword.)
__device__ __forceinline__ float4 load_f4(const char *sbase, unsigned voff)
{
    asm volatile("" : "+v"(voff));      // opaque: keeps hipcc from folding the lane offset into hoisted 64-bit VGPR bases
    return *reinterpret_cast<const float4 *>(sbase + voff);
}

template <bool TRACE, bool MUL>
__global__ __launch_bounds__(256, 2) void gated_conv_wino_kernel(const ConvKArgs a)
{
    using WG = WinoGeom;
    __shared__ __attribute__((aligned(16))) float lds[WG::LDS_FLOATS];
    // TRACE: rec = {start, end of prologue, ticks in epilogues: total, to 3rd barrier passed, to last LDS read landed,
    //        to residual/parameter loads landed, end, hw id} (10 ns ticks; tools/trace_conv.py)
    unsigned long long t_trace[2] = {0, 0}, t_ep[4] = {0, 0, 0, 0};
    if (TRACE) t_trace[0] = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int row = __builtin_amdgcn_readfirstlane(tid >> 6);          // frequency row i of this wave
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;                  // G % groups == 0: g is fixed per workgroup
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;

    // ---- units: u = blockIdx.x + k G  ->  tile u / groups = (by, bx), advanced by (wino_dby, wino_dbx) per step
    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // prefetch cursor
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    // ---- input patch staging.  Lane constants: LDS slot and byte offset of its float4s relative to the patch
    // origin; per unit only the scalar origin and the inside-the-image mask change.  Outside pixels load the
    // patch's (1,1) pixel (always inside) and are zeroed on the way into LDS.
    int loff[WG::NI];
    unsigned rel[WG::NI], aoff[WG::NI];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) {
        const int e = tid + i * 256, q = e % 4, pix = e / 4;
        loff[i] = e < WG::NE ? pix * WG::PS + 4 * q : -1;
        rel[i] = (unsigned)(((pix / WG::IW) * s.W + pix % WG::IW) * s.C + 4 * q) * 4u;
    }
    const unsigned safe_rel = (unsigned)((s.W + 1) * s.C) * 4u;
    const char *pbase = nullptr;                                        // patch origin of the cursor unit (+ chunk)
    long pdelta = 0;                                                    // MUL: byte distance source -> multiplier tensor
    if constexpr (MUL) pdelta = reinterpret_cast<const char *>(a.mul) - reinterpret_cast<const char *>(s.p);
    auto set_patch = [&]() {
        const int y0 = pby * (2 * WG::TR) - 1, x0 = pbx * (2 * WG::TC) - 1;
        pbase = reinterpret_cast<const char *>(s.p) + ((long)y0 * s.W + x0) * (long)(s.C * 4);
        okmask = 0;
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) {
            const int e = tid + i * 256, pix = e / 4, ppy = pix / WG::IW, ppx = pix % WG::IW;
            const bool ok = (e < WG::NE) & (ppy >= -y0) & (ppy < a.inH - y0) & (ppx >= -x0) & (ppx < a.inW - x0);
            okmask |= (ok ? 1u : 0u) << i;
            aoff[i] = ok ? rel[i] : safe_rel;
        }
    };
    // the cursor runs two chunks ahead of the MFMAs and crosses unit boundaries (past the last unit it repeats it)
    auto advance = [&]() {
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    float4 st[WG::NI], stm[MUL ? WG::NI : 1];                           // MUL (FAM): the conv input is src * mul
    auto gload1 = [&](int i) {
        st[i] = load_f4(pbase + pchunk * (WG::KC * 4), aoff[i]);
        if constexpr (MUL) stm[i] = load_f4(pbase + pdelta + pchunk * (WG::KC * 4), aoff[i]);
    };
    auto staged = [&](int i, unsigned mask, const float4 &x, const float4 *y) {
        float4 v = x;
        if constexpr (MUL) v = make_float4(x.x * y[i].x, x.y * y[i].y, x.z * y[i].z, x.w * y[i].w);
        return ((mask >> i) & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto lwrite1 = [&](int i, int obuf) {                               // lanes past the patch write a dummy slot
        const float4 v = staged(i, okmask, st[i], stm);
        *reinterpret_cast<float4 *>(lds + (loff[i] >= 0 ? obuf + loff[i] : 3 * WG::BUF)) = v;
    };

    // ---- per-wave constants of the input transform: rows (ra, rb) of the 4x4 patch, t = d[ra] + rs * d[rb]
    // (row 0: d0 - d2, row 1: d1 + d2, row 2: d1 - d2 = -(B^T d)[2], row 3: d1 - d3)
    const int ra = row == 0 ? 0 : 1, rb = row == 3 ? 3 : 2;
    const float rs = row == 1 ? 1.0f : -1.0f;
    const int t = lane & 31, tr = t >> 3, tc = t & 7, half = lane >> 5;
    const int abase_a = ((2 * tr + ra) * WG::IW + 2 * tc) * WG::PS + 4 * half;
    const int abase_b = ((2 * tr + rb) * WG::IW + 2 * tc) * WG::PS + 4 * half;
    float4 da[4], db[4];                                               // transform scratch (t[c] ends up in da[c])
    auto rd1 = [&](const float *buf, int kk, int r) {                  // LDS read r = 2c + {0: row ra, 1: row rb}
        const int c = r >> 1;
        if (r & 1) db[c] = *reinterpret_cast<const float4 *>(buf + abase_b + c * WG::PS + kk * 8);
        else da[c] = *reinterpret_cast<const float4 *>(buf + abase_a + c * WG::PS + kk * 8);
    };
    auto tt1 = [&](int c) {
        da[c].x = __builtin_fmaf(db[c].x, rs, da[c].x);
        da[c].y = __builtin_fmaf(db[c].y, rs, da[c].y);
        da[c].z = __builtin_fmaf(db[c].z, rs, da[c].z);
        da[c].w = __builtin_fmaf(db[c].w, rs, da[c].w);
    };
    auto sub4 = [](const float4 &x, const float4 &y) { return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w); };
    auto add4 = [](const float4 &x, const float4 &y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); };
    auto vv1 = [&](float4(&O)[4], int j) {                             // column combination j of B^T d B
        O[j] = j == 0 ? sub4(da[0], da[2]) : j == 1 ? add4(da[1], da[2]) : j == 2 ? sub4(da[2], da[1]) : sub4(da[1], da[3]);
    };

    floatx16 acc[4][2];                                             // written (C = 0) by the first chunk of every unit

    // ---- B fragments: wave (g, row) reads 8 tiles per k8 step, [j][f|m]; half-step = 4 tiles (two frequencies)
    const int nhs = n * 4;                                          // half-steps of one unit
    const char *const bbase = reinterpret_cast<const char *>(a.wp_wino) +
                              ((size_t)g * (n * 2) * 4 + row) * (8 * 64 * 16);           // wave-uniform
    const unsigned bvoff = lane * 16;
    float4 bq[2][4];
    auto bload1 = [&](int slot, int q, int hs) {                    // half-step hs = 2*step + h of the unit
        bq[slot][q] = load_f4(bbase + (size_t)(hs >> 1) * (4 * 8 * 64 * 16) + (hs & 1) * (4 * 64 * 16) + q * 1024, bvoff);
    };

    float *const red = lds + 3 * WG::BUF + 4;                       // [row][b][register][lane], one of {f, m} at a time

    // ---- prologue: the first two chunks of the stream into LDS, first B half-step, A fragments of k-step 0
    set_patch();
    {
        float4 st1[WG::NI], stm1[MUL ? WG::NI : 1];
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) gload1(i);
        const unsigned ok0 = okmask;
        advance();
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) {
            st1[i] = load_f4(pbase + pchunk * (WG::KC * 4), aoff[i]);
            if constexpr (MUL) stm1[i] = load_f4(pbase + pdelta + pchunk * (WG::KC * 4), aoff[i]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) bload1(0, q, 0);
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) {
            const float4 v0 = staged(i, ok0, st[i], stm);
            const float4 v1 = staged(i, okmask, st1[i], stm1);
            if (loff[i] >= 0) {
                *reinterpret_cast<float4 *>(lds + loff[i]) = v0;
                *reinterpret_cast<float4 *>(lds + WG::BUF + loff[i]) = v1;
            }
        }
        advance();
    }
    __syncthreads();
    float4 V[4], Vn[4];                                             // A fragments of k-step 0 / 1 of the running chunk
#pragma unroll
    for (int r = 0; r < 8; ++r) rd1(lds, 0, r);
#pragma unroll
    for (int c = 0; c < 4; ++c) tt1(c);
#pragma unroll
    for (int j = 0; j < 4; ++j) vv1(V, j);
    if (TRACE) t_trace[1] = __builtin_amdgcn_s_memrealtime();

    int o_cur = 0, o_nxt = WG::BUF, o_nn = 2 * WG::BUF;             // LDS buffers of chunk, chunk+1, chunk+2 of the stream

    // One chunk = 4 half-steps x 16 MFMAs, every other instruction an item in an MFMA's shadow.  FIRST: the unit's
    // first chunk starts the accumulators from C = 0 (no zeroing pass).
    auto chunk_body = [&](auto first_tag, int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const float *buf = lds + o_cur, *bufn = lds + o_nxt;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {                             // half-steps: (k-step q4 >> 1, frequency pair q4 & 1)
            int hs1 = chunk * 4 + q4 + 1;                            // next half-step; wraps into the next unit (same g)
            if (q4 == 3) hs1 = hs1 == nhs ? 0 : hs1;
            const int cur = q4 & 1, nxt = cur ^ 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int fm = 0; fm < 2; ++fm) {
                        const int j = 2 * (q4 & 1) + jj, m = e * 4 + jj * 2 + fm;
                        const float4 av = q4 < 2 ? V[j] : Vn[j];
                        const float4 bv = bq[cur][jj * 2 + fm];
                        const float ae = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
                        const float be = e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w;
                        if (FIRST && q4 < 2 && e == 0) {
                            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[j][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, zero, 0, 0, 0);
                        } else
                            acc[j][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc[j][fm], 0, 0, 0);
                        // ---- one item in the shadow of MFMA m.  Even half-step: patch load / LDS write, then LDS
                        // reads 0..6 of the next k-step's transform with t[0], t[1] as soon as their rows are in;
                        // odd half-step: read 7, t[2], t[3], the four column combinations.
                        const float *tb = q4 < 2 ? buf : bufn;               // k-step 1 of this chunk / k-step 0 of the next
                        const int tk = q4 < 2 ? 1 : 0;
                        if (m < 4) bload1(nxt, m, hs1);
                        else if ((q4 & 1) == 0) {
                            if (m - 4 < WG::NI) {
                                if (q4 == 0) gload1(m - 4);                              // patch at the cursor -> registers
                                else lwrite1(m - 4, o_nn);                               // ... -> LDS, half a chunk later
                            } else if (m >= 7 && m <= 10) rd1(tb, tk, m - 7);
                            else if (m == 11) tt1(0);
                            else if (m == 12 || m == 13) rd1(tb, tk, m - 8);
                            else if (m == 14) tt1(1);
                            else if (m == 15) rd1(tb, tk, 6);
                        } else {
                            if (m == 4) rd1(tb, tk, 7);
                            else if (m == 5) tt1(2);
                            else if (m == 7) tt1(3);
                            else if (m >= 8 && m < 12) { if (q4 == 1) vv1(Vn, m - 8); else vv1(V, m - 8); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        advance();
        __syncthreads();
        const int o = o_cur;
        o_cur = o_nxt;
        o_nxt = o_nn;
        o_nn = o;
    };

    for (int u = blockIdx.x; u < a.n_units; u += G) {
        chunk_body(std::true_type{}, 0);
        for (int chunk = 1; chunk < n; ++chunk) chunk_body(std::false_type{}, chunk);

        // ================= unit epilogue =================
        // MFMA D layout: register 4 row' + rr of lane (half, cout) is tile (row', rr + 4 half), i.e. output pixels
        // (oy0 + 2 row' + a, ox0 + 8 half + 2 rr + b).  Wave `row` finishes the registers with row' = row.  The data goes
        // through LDS anyway (cross-wave sum over the frequency rows), so it comes back TRANSPOSED: lane = (slot, channel
        // quad) with slot = (rr, half), 4 consecutive channels per lane -> 128-bit residual loads and output stores
        // (4 + 4 memory instructions per unit instead of 16 + 16; dword stores were the most expensive part of this phase).
        unsigned long long t_e0 = 0;
        if (TRACE) t_e0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_setprio(3);        // the co-resident workgroup's MFMA stream otherwise starves this phase
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int cq = lane & 7, e_rr = (lane >> 3) & 3, e_hf = lane >> 5;
        const int c0 = g * 32 + 4 * cq;                                                       // first channel of this lane
        const int oy = by * (2 * WG::TR) + 2 * row, ox = bx * (2 * WG::TC) + 8 * e_hf + 2 * e_rr;      // (a, b) = (0, 0)
        const int c_lim = a.fill_pad ? a.out_cstride : a.Cout;                              // channels stored per pixel
        const bool quad_st = c0 + 3 < c_lim && (a.out_cstride & 3) == 0;                     // whole-quad 128-bit stores
        const bool quad_ld = a.residual && c0 + 3 < a.Cout && (a.Cout & 3) == 0;
        const f32x4 bf = *reinterpret_cast<const f32x4 *>(a.params + c0);
        const f32x4 bm = *reinterpret_cast<const f32x4 *>(a.params + a.CoutPad + c0);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.params + 2 * a.CoutPad + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.params + 3 * a.CoutPad + c0);
        bool pix_in[2][2];
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int b = 0; b < 2; ++b) pix_in[aa][b] = (oy + aa < a.outH) & (ox + b < a.outW);
        // residual loads first, so the memory latency runs under the cross-wave reduction
        f32x4 rv[2][2];
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                rv[aa][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float *rp = a.residual + ((size_t)(oy + aa) * a.outW + ox + b) * a.Cout + c0;
                if (pix_in[aa][b] && quad_ld) rv[aa][b] = *reinterpret_cast<const f32x4 *>(rp);
                else if (pix_in[aa][b] && a.residual) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c0 + k < a.Cout) rv[aa][b][k] = rp[k];
                }
            }
        // output transform: in-wave over j (R_b = sum_j A^T[b][j] M[row][j]) into red[(row * 2 + b) * 16 + reg][lane],
        // then across the four row-waves (Y[a][b] = sum_i A^T[a][i] R_b(i)), conv_f then conv_m through the same 32 KiB
        f32x4 Y[2][2][2];                                            // [f|m][a][b], component = channel of the quad
        const int rsrc_lane = ((row * 4 + e_rr) * 64 + e_hf * 32 + 4 * cq);                  // (register, source lane) to read
#pragma unroll
        for (int fm = 0; fm < 2; ++fm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float r0 = acc[0][fm][r] + acc[1][fm][r] + acc[2][fm][r];
                const float r1 = acc[1][fm][r] - acc[2][fm][r] - acc[3][fm][r];
                red[((row * 2 + 0) * 16 + r) * 64 + lane] = r0;
                red[((row * 2 + 1) * 16 + r) * 64 + lane] = r1;
            }
            __syncthreads();
            if (TRACE && fm == 1) t_ep[1] += __builtin_amdgcn_s_memrealtime() - t_e0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 R[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) R[i] = *reinterpret_cast<const f32x4 *>(red + (i * 2 + b) * 16 * 64 + rsrc_lane);
                Y[fm][0][b] = R[0] + R[1] + R[2];
                Y[fm][1][b] = R[1] - R[2] - R[3];
            }
            if (fm == 0) __syncthreads();      // (after conv_m the next writer is a whole chunk of barriers away)
        }
        if (TRACE) t_ep[2] += __builtin_amdgcn_s_memrealtime() - t_e0;
        if (TRACE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_ep[3] += __builtin_amdgcn_s_memrealtime() - t_e0;
        }
        {
            constexpr float LOG2E = 1.44269504088896341f;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 f = Y[0][aa][b] + bf;
                    if (a.linear) {
                        // plain convolution (training path): conv_f + b_f at channel c, conv_m + b_m at channel Cout + c
                        const f32x4 m = Y[1][aa][b] + bm;
                        float *op = a.out + ((size_t)(oy + aa) * a.outW + ox + b) * a.out_cstride + c0;
                        if (pix_in[aa][b]) {
                            if (c0 + 3 < a.Cout && ((a.out_cstride | a.Cout) & 3) == 0) {
                                *reinterpret_cast<f32x4 *>(op) = f;
                                *reinterpret_cast<f32x4 *>(op + a.Cout) = m;
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (c0 + k < a.Cout) {
                                        op[k] = f[k];
                                        op[a.Cout + k] = m[k];
                                    }
                            }
                        }
                        if (a.out_gated && pix_in[aa][b]) {          // ... and the layer's output in the same pass (gate_forward_kernel)
                            constexpr float L2E = 1.44269504088896341f;
                            f32x4 g = f;
                            if (a.elu) {
                                const f32x4 fe = f * L2E;
#pragma unroll
                                for (int k = 0; k < 4; ++k) g[k] = f[k] > 0.0f ? f[k] : __builtin_amdgcn_exp2f(fe[k]) - 1.0f;
                            }
                            const f32x4 mn = m * -L2E;
                            f32x4 v;
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = (g[k] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mn[k]))) * sc[k] + sh[k];
                            if (a.blk_h > 0 && (oy + aa) % a.blk_h >= a.blk_valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
                            float *gp = a.out_gated + ((size_t)(oy + aa) * a.outW + ox + b) * a.Cout + c0;
                            if (c0 + 3 < a.Cout && (a.Cout & 3) == 0) *reinterpret_cast<f32x4 *>(gp) = v;
                            else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (c0 + k < a.Cout) gp[k] = v[k];
                            }
                        }
                        continue;
                    }
                    const f32x4 mm = (Y[1][aa][b] + bm) * -LOG2E;
                    if (a.elu) {                                     // x > 0 ? x : exp(x) - 1
                        const f32x4 fe = f * LOG2E;
#pragma unroll
                        for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : __builtin_amdgcn_exp2f(fe[k]) - 1.0f;
                    }
                    f32x4 sg;                                        // sigmoid(m) = 1 / (1 + exp(-m))
#pragma unroll
                    for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mm[k]));
                    f32x4 v = (f * sg) * sc + sh + rv[aa][b];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = c0 + k < a.Cout ? v[k] : a.out_fill;
                    float *op = a.out + ((size_t)(oy + aa) * a.outW + ox + b) * a.out_cstride + c0;
                    if (pix_in[aa][b]) {
                        if (quad_st) *reinterpret_cast<f32x4 *>(op) = v;
                        else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (c0 + k < c_lim) op[k] = v[k];
                        }
                    }
                }
        }
        step_tile(by, bx);
        __builtin_amdgcn_s_setprio(0);
        if (TRACE) t_ep[0] += __builtin_amdgcn_s_memrealtime() - t_e0;
    }
    if (TRACE && a.trace && lane == 0) {                        // one record per wave: 4 per workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *rec = a.trace + ((size_t)blockIdx.x * 4 + row) * 8;
        rec[0] = t_trace[0];
        rec[1] = t_trace[1];
        rec[2] = t_ep[0];
        rec[3] = t_ep[1];
        rec[4] = t_ep[2];
        rec[5] = t_ep[3];
        rec[6] = __builtin_amdgcn_s_memrealtime();
        rec[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID (wave/simd/cu/sh/se)
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 subtraction: hipcc scalarises `a - b` on float vectors (two v_sub_f32 per pair); the packed add with both halves of
// the second operand negated is one instruction
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)      // (... and now and then an addition too)
{
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_sub(a.lo, b.lo), hi = pk_sub(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// ------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3), wave-autonomous form on v_mfma_f32_16x16x4_f32 (round 3; index maps mirrored in tests/wino16_ref.py).
//
// The kernel above gives a wave one frequency ROW of a 32-tile x 32-channel unit, so the output transform Y = A^T M A needs the
// other three waves: two passes through 32 KiB of LDS and three barriers per unit.  Here the unit is cut the other way:
//   * wave w owns output channels 8w .. 8w+7 of the 32-channel group and ALL 16 frequencies of all 32 tiles: 16x16x4 MFMAs with
//     A operand = transformed weights (rows 0..7 = conv_f, rows 8..15 = conv_m of those 8 channels, straight from L2,
//     [group][wave][chunk][row a][j][lane][4 k-steps]), B operand = transformed input of 16 tiles (block b = tile rows 2b, 2b+1);
//     16 frequencies x 2 blocks x 4 registers = 128 accumulators; a lane ends up with every frequency of its (tile, 4 channels),
//     so A^T M A is lane-local; one v_permlane32_swap per register pair brings conv_f (lanes 0..31) and conv_m (lanes 32..63)
//     together, the gate is lane-local and a lane loads / stores 4 consecutive channels of a pixel;
//   * the INPUT TRANSFORM is shared by the four waves through LDS.  tools/issue_probe.py (profiles/r3_issue_probe.json): the fp32
//     MFMA runs on the FP32 vector pipe itself — a v_add_f32 behind a v_mfma_f32_* costs its full issue time whether it is "in the
//     shadow" or not (LDS reads, SALU and global loads DO overlap) — so a first version in which every wave formed B^T d B of all
//     32 tiles for itself spent 19 % of its loop there (profiles/r3_w16_ablation.md; that kernel is in the history, 9e4c1b0).
//     A chunk is processed in two PARTS (frequency rows 2p, 2p + 1): wave w forms row a = 2p + (w & 1) of block tb = w >> 1 (lane =
//     tile x 4 input channels): 8 ds_read_b128 of the raw patch, 32 VALU, 4 ds_write_b128 into Vbuf[p][8 frequencies][32 tiles]
//     [16 channels] (16 KiB, XOR-swizzled float4 slots: stores and loads are bank-conflict free);
//   * B operands are read from Vbuf with one ds_read_b128 per (frequency, block) = 4 k-steps; 64 MFMAs per part, order (frequency,
//     k-step, block) so that an accumulator is touched every second MFMA;
//   * pipeline: during the MFMAs of part s the wave transforms part s + 1 into the other V buffer, fetches the weights three
//     frequencies ahead and stages the raw patch of the next chunk (two raw buffers); one barrier per part.
// Measured equal to or slower than the row-per-wave kernel on the full C -> C layers (profiles/README.md), so it is the default only
// where most of a 32-channel group is padding (the 32 -> 3 output layer: waves without a real channel skip their MFMAs).
// LDS: 2 x 15 KiB raw + 3 x 16 KiB V.
struct Wino16sGeom {
    static constexpr int IH = 10, IW = 18, KC = 16, PS = KC + 4;
    static constexpr int RS = IW * PS + 24;                    // floats per raw patch row (384)
    static constexpr int BUF = IH * RS;                        // raw patch buffer (3840 floats)
    static constexpr int VBUF = 8 * 32 * 16;                   // one part of the transformed chunk (4096 floats)
    static constexpr int NE = IH * IW * (KC / 4), NI = (NE + 255) / 256;
    static constexpr int V0 = 2 * BUF;                         // float offset of Vbuf[0]
    static constexpr int LDS_FLOATS = 2 * BUF + 3 * VBUF + 4;  // two raw buffers, three V buffers, a dummy float4 slot
};

template <bool MUL, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gated_conv_wino16s_kernel(const ConvKArgs a)
{
    using WG = Wino16sGeom;
    __shared__ __attribute__((aligned(16))) float lds[WG::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;

    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // prefetch cursor (raw patches)
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    // ---- raw patch staging (global -> registers -> LDS), two buffers
    int loff[WG::NI];
    unsigned rel[WG::NI], aoff[WG::NI];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) {
        const int e = tid + i * 256, q = e % 4, pix = e / 4;
        loff[i] = e < WG::NE ? (pix / WG::IW) * WG::RS + (pix % WG::IW) * WG::PS + 4 * q : -1;
        rel[i] = (unsigned)(((pix / WG::IW) * s.W + pix % WG::IW) * s.C + 4 * q) * 4u;
    }
    const unsigned safe_rel = (unsigned)((s.W + 1) * s.C) * 4u;
    const char *pbase = nullptr;
    long pdelta = 0;
    if constexpr (MUL) pdelta = reinterpret_cast<const char *>(a.mul) - reinterpret_cast<const char *>(s.p);
    auto set_patch = [&]() {
        const int y0 = pby * 8 - 1, x0 = pbx * 16 - 1;
        pbase = reinterpret_cast<const char *>(s.p) + ((long)y0 * s.W + x0) * (long)(s.C * 4);
        okmask = 0;
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) {
            const int e = tid + i * 256, pix = e / 4, ppy = pix / WG::IW, ppx = pix % WG::IW;
            const bool ok = (e < WG::NE) & (ppy >= -y0) & (ppy < a.inH - y0) & (ppx >= -x0) & (ppx < a.inW - x0);
            okmask |= (ok ? 1u : 0u) << i;
            aoff[i] = ok ? rel[i] : safe_rel;
        }
    };
    auto advance = [&]() {                                   // the cursor crosses unit boundaries (past the last unit it repeats it)
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    float4 st[WG::NI], stm[MUL ? WG::NI : 1];
    unsigned st_ok = 0;                                      // in-image mask of the chunk held in st[]
    auto gload1 = [&](int i) {
        st[i] = load_f4(pbase + pchunk * (WG::KC * 4), aoff[i]);
        if constexpr (MUL) stm[i] = load_f4(pbase + pdelta + pchunk * (WG::KC * 4), aoff[i]);
    };
    auto lwrite1 = [&](int i, int obuf) {
        float4 v = st[i];
        if constexpr (MUL) v = make_float4(v.x * stm[i].x, v.y * stm[i].y, v.z * stm[i].z, v.w * stm[i].w);
        if (!((st_ok >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(__builtin_assume_aligned(lds + (loff[i] >= 0 ? obuf + loff[i] : WG::LDS_FLOATS - 4), 16)) = v;
    };

    // ---- lane constants.  t16 = tile within a 16-tile block (MFMA column / transform item), kl = channel quad
    const int t16 = lane & 15, kl = lane >> 4;
    const int lbase = (2 * (t16 >> 3)) * WG::RS + (2 * (t16 & 7)) * WG::PS + 4 * kl;               // raw patch (floats)
    const int vlane = t16 * 16 + 4 * (kl ^ ((t16 >> 1) & 3));                                       // swizzled float4 slot of a V row
    const int t_blk = wv >> 1, t_row = wv & 1;                                                      // transform role of this wave
    const int vwbase = WG::V0 + ((t_row * 4) * 32 + 16 * t_blk) * 16 + vlane;                       // + V buffer + j * 512
    const int rbase = lbase + (4 * t_blk) * WG::RS;                                                 // + raw buffer

    // This wave's share of B^T d B for the next stage as 18 small steps (columns in the order 0, 2, 1, 3 so that every V[j] is
    // stored as soon as it exists: at most five float4 are live).  T(c) = d[ra][c] +- d[rb][c];  V0 = T0 - T2, V1 = T1 + T2,
    // V2 = T2 - T1, V3 = T1 - T3.
    float4 T0, T1, T2, T3, tq;
    auto sub4 = [](const float4 &x, const float4 &y) { return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w); };
    auto add4 = [](const float4 &x, const float4 &y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); };
    auto t_step = [&](const float *raw, int arow, int vb, int k) {
        const int ra = arow == 0 ? 0 : arow == 2 ? 2 : 1, rb = arow == 0 ? 2 : arow == 1 ? 2 : arow == 2 ? 1 : 3;
        auto rdA = [&](int c) { return *reinterpret_cast<const float4 *>(__builtin_assume_aligned(raw + rbase + c * WG::PS + ra * WG::RS, 16)); };
        auto rdB = [&](int c) { return *reinterpret_cast<const float4 *>(__builtin_assume_aligned(raw + rbase + c * WG::PS + rb * WG::RS, 16)); };
        // d[ra] + sgn d[rb] as one FMA per component: arow depends on the wave (t_row), a select between add and sub would be
        // a BRANCH in the middle of the MFMA stream (seen in the ISA of the first build)
        const float sgn = arow == 1 ? 1.0f : -1.0f;
        auto row = [&](const float4 &x, const float4 &y) {
            return make_float4(__builtin_fmaf(y.x, sgn, x.x), __builtin_fmaf(y.y, sgn, x.y), __builtin_fmaf(y.z, sgn, x.z),
                               __builtin_fmaf(y.w, sgn, x.w));
        };
        auto wr = [&](int j, const float4 &v) {
            *reinterpret_cast<float4 *>(__builtin_assume_aligned(lds + vwbase + vb + j * 512, 16)) = v;
        };
        switch (k) {
        case 0: T0 = rdA(0); break;
        case 1: tq = rdB(0); break;
        case 2: T2 = rdA(2); break;
        case 3: T1 = rdB(2); break;                 // (T1 is free until column 1 arrives)
        case 4: T0 = row(T0, tq); break;
        case 5: T2 = row(T2, T1); break;
        case 6: T1 = rdA(1); break;
        case 7: tq = rdB(1); break;
        case 8: T0 = sub4(T0, T2); break;           // V0
        case 9: wr(0, T0); break;
        case 10: T3 = rdA(3); break;
        case 11: T0 = rdB(3); break;                // (T0 is free after its store)
        case 12: T1 = row(T1, tq); break;
        case 13: tq = add4(T1, T2); break;          // V1
        case 14: T2 = sub4(T2, T1); break;          // V2
        case 15: wr(1, tq); break;
        case 16: wr(2, T2); break;
        case 17: T3 = row(T3, T0); break;
        case 18: T3 = sub4(T1, T3); break;          // V3
        case 19: wr(3, T3); break;
        default: break;
        }
    };
    constexpr int T_STEPS = 20;
    // ---- A operand (weights): [group][wave][stage = 2 chunk + part][fl][lane][4]; ring of 4 frequencies, fetched 3 ahead
    const char *const wbase = reinterpret_cast<const char *>(a.wp_w16) + ((size_t)(g * 4 + wv) * n) * (16 * 1024);
    const unsigned wvoff = lane * 16;
    float4 Wq[4];
    auto wload1 = [&](int slot, int stage, int fl) { Wq[slot] = load_f4(wbase + (size_t)(stage * 8 + fl) * 1024, wvoff); };
    // ---- B operand: Vbuf[part][fl][16 b + t16][slot]
    float4 Bq[4][2];                                             // ring of four frequencies: fetched two ahead
    auto bload1 = [&](int slot, int vb, int fl, int b) {
        Bq[slot][b] = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(lds + WG::V0 + vb + (fl * 32 + 16 * b) * 16 + vlane, 16));
    };

    f32x4 acc[2][4][4];
    // the wave's 8 channels exist, rows are quad aligned, no padded-channel fill: the epilogue of interior units needs no masks
    const bool chan_full = (g * 32 + wv * 8 + 8 <= a.Cout) & ((a.out_cstride & 3) == 0) & ((a.Cout & 3) == 0) & !a.fill_pad;

    // ---- prologue: raw(0) -> LDS, raw(1) -> registers, V(0, part 0), the first three weight fragments
    const int nstages = 2 * n;
    set_patch();
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) gload1(i);
    st_ok = okmask;
#pragma unroll
    for (int j = 0; j < 3; ++j) wload1(j, 0, j);
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) lwrite1(i, 0);
    advance();
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) gload1(i);
    st_ok = okmask;
    advance();
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < T_STEPS; ++k2) t_step(lds, t_row, 0, k2);
    __syncthreads();
    bload1(0, 0, 0, 0);
    bload1(0, 0, 0, 1);
    bload1(1, 0, 1, 0);
    bload1(1, 0, 1, 1);

    int raw_cur = 0, raw_nxt = WG::BUF;                        // raw buffers of this stage's chunk / the next chunk
    int v_cur = 0, v_nxt = WG::VBUF, v_nn = 2 * WG::VBUF;       // V buffers (float offsets): this stage, the next, the free one

    // One stage = part P of a chunk: 64 MFMAs; shadow items prepare the NEXT stage.  The barrier sits in the MIDDLE of a stage,
    // right after this wave's share of the next stage's V has been stored: the second half of the stage can then already
    // fetch the next stage's first B operands, so no wave waits on LDS behind a barrier.  Three V buffers make that safe
    // (the buffer written in stage s was last read in stage s - 2, which every wave has left before the barrier of s - 1).
    auto stage_body = [&](auto first_tag, auto part_tag, int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int P = decltype(part_tag)::value;
        const int stage = 2 * chunk + P;
        int nstage = stage + 1;                                       // wraps into the next unit (same weights)
        nstage = nstage == nstages ? 0 : nstage;
        const float *traw = lds + (P == 0 ? raw_cur : raw_nxt);       // raw patch of the next stage's chunk
        const int narow = 2 * (1 - P) + t_row;                        // frequency row this wave forms for the next stage
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fl = 0; fl < 8; ++fl) {
            const int arow = 2 * P + (fl >> 2), j = fl & 3;
            const int bs = fl & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int m = fl * 8 + e * 2 + b;
                    const float4 wv4 = Wq[(ABL & 2) ? 0 : (fl & 3)], vv4 = Bq[(ABL & 4) ? 0 : bs][(ABL & 4) ? 0 : b];
                    const float we = e == 0 ? wv4.x : e == 1 ? wv4.y : e == 2 ? wv4.z : wv4.w;
                    const float ve = e == 0 ? vv4.x : e == 1 ? vv4.y : e == 2 ? vv4.z : vv4.w;
                    if (FIRST && e == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[b][arow][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(we, ve, zero, 0, 0, 0);
                    } else
                        acc[b][arow][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(we, ve, acc[b][arow][j], 0, 0, 0);
                    // ---- shadow items
                    const int mm = e * 2 + b;                                        // position inside this frequency's 8 MFMAs
                    if (!(ABL & 4) && mm < 2) {                                      // B operands two frequencies ahead
                        if (fl + 2 < 8) bload1((fl + 2) & 3, v_cur, fl + 2, mm);
                        else bload1((fl + 2) & 3, v_nxt, fl + 2 - 8, mm);           // next stage (stored before this stage's barrier)
                    }
                    if (!(ABL & 2) && mm == 2) {                                     // weights three frequencies ahead
                        if (fl + 3 < 8) wload1((fl + 3) & 3, stage, fl + 3);
                        else wload1((fl + 3) & 3, nstage, fl + 3 - 8);
                    }
                    if (!(ABL & 32)) {                                               // the next stage's share of B^T d B
                        // program order = step order (the steps recycle registers); every arithmetic step sits at least 8 MFMAs
                        // behind the LDS reads it consumes, everything is stored before the barrier at m = 39
                        constexpr int at[T_STEPS] = {2, 3, 4, 5, 12, 13, 14, 15, 15, 16, 16, 17, 23, 24, 24, 25, 26, 27, 28, 29};
#pragma unroll
                        for (int k2 = 0; k2 < T_STEPS; ++k2)
                            if (at[k2] == m) t_step(traw, narow, v_nxt, k2);
                    }
                    if (!(ABL & 8)) {
                        if (P == 0 && m >= 28 && m - 28 < WG::NI) lwrite1(m - 28, raw_nxt);        // raw(chunk + 1): registers -> LDS
                        if (P == 1 && m >= 50 && m - 50 < WG::NI) gload1(m - 50);                  // raw(chunk + 2) -> registers
                    }
                    if (m == 39 && !(ABL & 16)) __syncthreads();                     // V of the next stage (and raw(chunk + 1)) complete
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (P == 1) {
            if (!(ABL & 8)) st_ok = okmask;
            advance();
            const int o = raw_cur;
            raw_cur = raw_nxt;
            raw_nxt = o;
        }
        const int v = v_cur;
        v_cur = v_nxt;
        v_nxt = v_nn;
        v_nn = v;
    };

    if (g * 32 + wv * 8 >= a.Cout) {
        // a wave whose 8 channels are all padding (Cout = 3) takes part in staging and transforming only
        for (int u = blockIdx.x; u < a.n_units; u += G)
            for (int chunk = 0; chunk < n; ++chunk)
                for (int P = 0; P < 2; ++P) {
                    const float *traw = lds + (P == 0 ? raw_cur : raw_nxt);
                    const int narow = 2 * (1 - P) + t_row;
#pragma unroll
                    for (int k2 = 0; k2 < T_STEPS; ++k2) t_step(traw, narow, v_nxt, k2);
                    if (P == 0) {
#pragma unroll
                        for (int i = 0; i < WG::NI; ++i) lwrite1(i, raw_nxt);
                    }
                    __syncthreads();
                    if (P == 1) {
#pragma unroll
                        for (int i = 0; i < WG::NI; ++i) gload1(i);
                        st_ok = okmask;
                        advance();
                        const int o = raw_cur;
                        raw_cur = raw_nxt;
                        raw_nxt = o;
                    }
                    const int v = v_cur;
                    v_cur = v_nxt;
                    v_nxt = v_nn;
                    v_nn = v;
                }
        return;
    }

    for (int u = blockIdx.x; u < a.n_units; u += G) {
        stage_body(std::true_type{}, std::integral_constant<int, 0>{}, 0);
        stage_body(std::true_type{}, std::integral_constant<int, 1>{}, 0);
        for (int chunk = 1; chunk < n; ++chunk) {
            stage_body(std::false_type{}, std::integral_constant<int, 0>{}, chunk);
            stage_body(std::false_type{}, std::integral_constant<int, 1>{}, chunk);
        }
        if (ABL & 1) {                                                 // keep the accumulators live with one store per wave
            f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ssum += acc[b][aa][j];
            if (ssum[0] + ssum[1] + ssum[2] + ssum[3] == 12345.678f) a.out[lane] = ssum[0];
            step_tile(by, bx);
            continue;
        }
        // ================= unit epilogue (lane-local) =================
        // D layout of v_mfma_f32_16x16x4_f32: lane (t = lane & 15, q = lane >> 4), register r = MFMA row 4q + r:
        // q = 0, 1 -> conv_f of channels 4q + r; q = 2, 3 -> conv_m of channels 4 (q - 2) + r; column = tile t of block b.
        __builtin_amdgcn_s_setprio(1);
        const int cq = (lane >> 4) & 1, eb = lane >> 5;                                    // channel quad, block finished by this lane
        const int c0 = g * 32 + wv * 8 + 4 * cq;
        const int oy = by * 8 + 2 * (2 * eb + (t16 >> 3)), ox = bx * 16 + 2 * (t16 & 7);
        // interior unit with 8 real channels and quad-aligned tensors: no masks at all (every unit of a C -> C layer but the
        // ragged bottom / right blocks)
        const bool full = (by * 8 + 8 <= a.outH) & (bx * 16 + 16 <= a.outW) & chan_full;
        const f32x4 bf = *reinterpret_cast<const f32x4 *>(a.params + c0);
        const f32x4 bm = *reinterpret_cast<const f32x4 *>(a.params + a.CoutPad + c0);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.params + 2 * a.CoutPad + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.params + 3 * a.CoutPad + c0);
        const int c_lim = a.fill_pad ? a.out_cstride : a.Cout;
        const bool quad_st = c0 + 3 < c_lim && (a.out_cstride & 3) == 0;
        const bool quad_ld = a.residual && c0 + 3 < a.Cout && (a.Cout & 3) == 0;
        bool pix_in[2][2];
        f32x4 rv[2][2];
        const unsigned ro = (unsigned)((oy * a.outW + ox) * a.Cout + c0) * 4u;              // byte offsets of pixel (0, 0)
        const unsigned oo = (unsigned)((oy * a.outW + ox) * a.out_cstride + c0) * 4u;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                rv[pa][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (full) {
                    if (a.residual)
                        rv[pa][pb] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(a.residual) + ro +
                                                                      (unsigned)((pa * a.outW + pb) * a.Cout) * 4u);
                    continue;
                }
                pix_in[pa][pb] = (oy + pa < a.outH) & (ox + pb < a.outW);
                const float *rp = a.residual + ((size_t)(oy + pa) * a.outW + ox + pb) * a.Cout + c0;
                if (pix_in[pa][pb] && quad_ld) rv[pa][pb] = *reinterpret_cast<const f32x4 *>(rp);
                else if (pix_in[pa][pb] && a.residual) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c0 + k < a.Cout) rv[pa][pb][k] = rp[k];
                }
            }
        // Y[pa][pb] = sum_a sum_j A^T[pa][a] M[a][j] A^T[pb][j],  A^T = [1 1 1 0; 0 1 -1 -1]: first over j (R[a][pb], 4 additions per
        // row), then over a (4 per column) = 24 vector additions per block
        f32x4 Yf[2][2], Ym[2][2];
        {
            f32x4 yb[2][2][2];                                         // [block][pa][pb]
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 R[4][2];
#pragma unroll
                for (int ar = 0; ar < 4; ++ar) {
                    const f32x4 s12 = acc[b][ar][1] + acc[b][ar][2], d12 = acc[b][ar][1] - acc[b][ar][2];
                    R[ar][0] = acc[b][ar][0] + s12;
                    R[ar][1] = d12 - acc[b][ar][3];
                }
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const f32x4 s12 = R[1][pb] + R[2][pb], d12 = R[1][pb] - R[2][pb];
                    yb[b][0][pb] = R[0][pb] + s12;
                    yb[b][1][pb] = d12 - R[3][pb];
                }
            }
            // lanes 0..31 hold conv_f, lanes 32..63 conv_m of (block 0 | block 1): after the half exchange the lower half-wave
            // owns block 0 and the upper half block 1, f in one register and m in the other (whole-vector bit casts: with
            // __builtin_bit_cast of single vector ELEMENTS this hipcc folds the four swaps into one — seen in the ISA)
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    u32x4 u0 = __builtin_bit_cast(u32x4, yb[0][pa][pb]), u1 = __builtin_bit_cast(u32x4, yb[1][pa][pb]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                        u0[k] = sw[0];
                        u1[k] = sw[1];
                    }
                    Yf[pa][pb] = __builtin_bit_cast(f32x4, u0);
                    Ym[pa][pb] = __builtin_bit_cast(f32x4, u1);
                }
        }
        {
            constexpr float LOG2E = 1.44269504088896341f;
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    f32x4 f = Yf[pa][pb] + bf;
                    const f32x4 mm = (Ym[pa][pb] + bm) * -LOG2E;
                    if (a.elu) {
                        const f32x4 fe = f * LOG2E;
#pragma unroll
                        for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : __builtin_amdgcn_exp2f(fe[k]) - 1.0f;
                    }
                    f32x4 sg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mm[k]));
                    f32x4 v = (f * sg) * sc + sh + rv[pa][pb];
                    if (full) {
                        *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(a.out) + oo + (unsigned)((pa * a.outW + pb) * a.out_cstride) * 4u) = v;
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = c0 + k < a.Cout ? v[k] : a.out_fill;
                    float *op = a.out + ((size_t)(oy + pa) * a.outW + ox + pb) * a.out_cstride + c0;
                    if (pix_in[pa][pb]) {
                        if (quad_st) *reinterpret_cast<f32x4 *>(op) = v;
                        else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (c0 + k < c_lim) op[k] = v[k];
                        }
                    }
                }
        }
        step_tile(by, bx);
        __builtin_amdgcn_s_setprio(0);
    }
}

// ------------------------------------------------------------------------------------------
// Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32: every 3x3 / stride-1 layer with Cin >= 32 (round 3; index maps: tests/wino4_ref.py).
//
// 4x fewer multiplications than the direct form (F(2x2,3x3): 2.25x), paid with more additions — and on this machine every VALU
// instruction costs FP-pipe time next to the fp32 MFMAs (tools/issue_probe.py), so the design minimises VALU work per MFMA:
//   * unit = 2 x 8 tiles of 4 x 4 output pixels (8 x 32 pixels) x 32 output channels; a workgroup of four waves, ONE per SIMD
//     (144 accumulators + deep operand rings per wave; one wave per SIMD also keeps the weight stream — one 1 KiB fragment per
//     four MFMAs — at half of what a SIMD's vector-memory path issues);
//   * the input transform B^T d B (36 frequencies of a 6 x 6 patch) is computed ONCE per (tile, input channel): thread = (channel
//     of the 16-channel chunk, tile), 36 ds_read_b32 of the raw patch, 95 VALU in packed fp32 (columns in pairs, the row pass with
//     op_sel half-selects), 18 ds_write2st64_b32 into a V buffer [frequency][tile][swizzled channel] that all four waves read
//     their B operands from (one ds_read_b128 = 4 k-steps);
//   * no per-lane address arithmetic or masks beside the MFMAs: raw patches, weights, residual and output go through buffer
//     descriptors (SGPR offsets; out-of-image lanes carry an out-of-range offset: loads return zeros, stores are dropped);
//   * wave w owns output channels 8w .. 8w+7: A operand = G g G^T of conv_f (rows 0..7) and conv_m (rows 8..15), straight from L2
//     ([group][wave][chunk][frequency][lane][4 k-steps], ten frequencies ahead); a lane ends up with all 36 frequencies
//     of its (tile, 4 channels): the output transform A^T M A is lane-local, one v_permlane32_swap per register pair brings conv_f
//     and conv_m of half of the pixels together, then the usual gate / BatchNorm / residual epilogue with 128-bit accesses;
//   * per 16-channel chunk: 144 MFMAs (frequencies in pairs, so an accumulator is touched every second MFMA), beside them the
//     transform of the NEXT chunk into the other V buffer and the staging of the raw patch two chunks ahead; one barrier, eight
//     MFMAs before the end of the chunk.
struct Wino4Geom {
    static constexpr int IH = 10, IW = 34, KC = 16, PS = KC + 4;
    static constexpr int RS = IW * PS;                         // floats per raw patch row (680)
    static constexpr int BUF = IH * RS;                        // raw patch buffer (6800 floats)
    static constexpr int VBUF = 36 * 16 * 16;                  // transformed chunk (9216 floats)
    static constexpr int NE = IH * IW * (KC / 4), NI = (NE + 255) / 256;    // 1360 float4 -> 6 per thread
    static constexpr int V0 = 2 * BUF;
    static constexpr int LDS_FLOATS = 2 * BUF + 2 * VBUF + 4;  // two raw buffers, two V buffers, a dummy float4 slot (128 KiB)
};

// ABL (attribution probes, results invalid; -DREAD_DEBUG_KNOBS builds only, read_tuning_set("conv_abl")): 1 no transform
// arithmetic, 2 no transform LDS reads, 4 no transform LDS writes, 8 no raw-patch global loads, 16 no raw-patch LDS writes,
// 32 weights loaded once, 64 B operands loaded once, 128 no epilogue, 256 no barrier
// LIN: the training path's linear launches — 1: pre-activations + gated output (forward), 2: pre-activations only (dgrad) —
// separate instantiations, so that every kernel's unit loop keeps ONE path through its epilogue (the waitcnt bookkeeping of hipcc merges every path at the loop
// header: with the dgrad's early `continue` in the same function every unit started with s_waitcnt vmcnt(0))
template <bool MUL, int ABL = 0, int LIN = 0>
__global__ __launch_bounds__(256, 1) void gated_conv_wino4_kernel(const ConvKArgs a)
{
    using WG = Wino4Geom;
    __shared__ __attribute__((aligned(16))) float lds[WG::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;

    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit (8 x 32 pixel block)
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // prefetch cursor (raw patches)
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    // ---- raw patch staging (global -> registers -> LDS), two buffers.  Buffer loads: a lane whose pixel lies outside the image
    // (or whose element does not exist: 1360 float4 over 6 x 256 lanes) carries an out-of-range offset and the hardware returns
    // zeros — no masks, no selects beside the MFMAs (every VALU instruction there costs FP-pipe time)
    int loff[WG::NI];
    unsigned rel[WG::NI], aoff[WG::NI];
    constexpr unsigned OOR = 0x80000000u;                      // >= the tensor's size (conv_uses_w4), no 32-bit wrap with the chunk offset
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) {
        const int e = tid + i * 256, q = e % 4, pix = e / 4;
        loff[i] = e < WG::NE ? (pix / WG::IW) * WG::RS + (pix % WG::IW) * WG::PS + 4 * q : WG::KC;   // else: pixel 0's pad floats
        rel[i] = (unsigned)(((pix / WG::IW) * s.W + pix % WG::IW) * s.C + 4 * q) * 4u;
    }
    const unsigned src_bytes = (unsigned)(a.inH * s.W * s.C) * 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, src_bytes, 0x00020000);
    const auto rsrc_mul = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MUL ? a.mul : s.p), 0, src_bytes, 0x00020000);
    auto set_patch = [&]() {
        const int y0 = pby * 8 - 1, x0 = pbx * 32 - 1;
        const unsigned base = (unsigned)((y0 * s.W + x0) * s.C) * 4u;          // may wrap: only lanes inside the image use it
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) {
            const int e = tid + i * 256, pix = e / 4, ppy = pix / WG::IW, ppx = pix % WG::IW;
            const bool ok = (e < WG::NE) & (ppy >= -y0) & (ppy < a.inH - y0) & (ppx >= -x0) & (ppx < a.inW - x0);
            aoff[i] = ok ? base + rel[i] : OOR;
        }
    };
    auto advance = [&]() {
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    float4 st[WG::NI], stm[MUL ? WG::NI : 1];
    auto gload1 = [&](int i) {
        st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, aoff[i], pchunk * (WG::KC * 4), 0));
        if constexpr (MUL) stm[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_mul, aoff[i], pchunk * (WG::KC * 4), 0));
    };
    auto lwrite1 = [&](int i, int obuf) {
        float4 v = st[i];
        if constexpr (MUL) v = make_float4(v.x * stm[i].x, v.y * stm[i].y, v.z * stm[i].z, v.w * stm[i].w);
        *reinterpret_cast<float4 *>(__builtin_assume_aligned(lds + obuf + loff[i], 16)) = v;
    };

    // ---- transform role: thread = (input channel c16 of the chunk, tile tl): one 6 x 6 patch -> 36 frequencies
    const int c16 = tid & 15, tl = tid >> 4;
    const int rbase = ((4 * (tl >> 3)) * WG::IW + 4 * (tl & 7)) * WG::PS + c16;                    // raw patch (floats)
    const int vwoff = WG::V0 + tl * 16 + ((c16 >> 2) ^ ((tl >> 1) & 3)) * 4 + (c16 & 3);           // + V buffer + frequency * 256
    // The patch lives in registers as column PAIRS d2[r][cp] = (d[r][2cp], d[r][2cp + 1]): packed fp32 instructions cost the issue
    // time of scalar ones beside the MFMAs (tools/issue_probe.py), so the transform is written for v_pk_*: 96 instructions instead
    // of 168 on the pipe the MFMAs run on.
    f32x2 d2[6][3];
    if (ABL) {
#pragma unroll
        for (int i = 0; i < 18; ++i) d2[i / 3][i % 3] = f32x2{1.0f + tid, 2.0f + tid};
    }
    // 1-D transform with B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]:
    // down the rows, two columns at a time: 14 packed operations
    auto bt6v = [](f32x2 &x0, f32x2 &x1, f32x2 &x2, f32x2 &x3, f32x2 &x4, f32x2 &x5) {
        const f32x2 p = pk_add(x3, x4), q = pk_add(x1, x2), r = pk_sub(x4, x3), u = pk_sub(x1, x2), f = pk_sub(x3, x1), h = pk_sub(x4, x2);
        const f32x2 y0 = __builtin_elementwise_fma(x2, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x0, f32x2{4.0f, 4.0f}, x4));
        const f32x2 y5 = __builtin_elementwise_fma(x3, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x1, f32x2{4.0f, 4.0f}, x5));
        x0 = y0;
        x1 = __builtin_elementwise_fma(q, f32x2{-4.0f, -4.0f}, p);
        x2 = __builtin_elementwise_fma(u, f32x2{4.0f, 4.0f}, r);
        x3 = __builtin_elementwise_fma(f, f32x2{2.0f, 2.0f}, h);
        x4 = __builtin_elementwise_fma(f, f32x2{-2.0f, -2.0f}, h);
        x5 = y5;
    };
    // ... and along a row whose six values x0..x5 sit in three pairs P0 = (x0, x1), P1 = (x2, x3), P2 = (x4, x5): the half selects
    // (op_sel) and per-half negations of the packed instructions do the shuffling, 9 instructions; results
    // P0 = (y0, y5), P1 = (y1, y2), P2 = (y3, y4)
    const f32x2 KA = {4.0f, -5.0f}, KB = {2.0f, 0.0f};
    auto bt6row = [&](f32x2 &P0, f32x2 &P1, f32x2 &P2) {
        f32x2 T, Y05, QU, PR, Y12, Y34;
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(T) : "v"(P0), "v"(KA), "v"(P2));                  // 4 x0 + x4 | 4 x1 + x5
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(Y05) : "v"(P1), "v"(KA), "v"(T));                 // - 5 x2 | - 5 x3
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(QU) : "v"(P0), "v"(P1));                     // x1 + x2 | x1 - x2
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(PR) : "v"(P2), "v"(P1));                     // x4 + x3 | x4 - x3
        f32x2 F2, H2;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(F2) : "v"(P1), "v"(P0));       // x3 - x1 | x1 - x3
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(H2) : "v"(P2), "v"(P1));       // x4 - x2 | x4 - x2
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(Y12) : "v"(QU), "v"(KA), "v"(PR));  // p - 4 q | r + 4 u
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(Y34) : "v"(F2), "v"(KB), "v"(H2));                // h + 2 f | h - 2 f
        P0 = Y05;
        P1 = Y12;
        P2 = Y34;
    };
    // step k of the next chunk's transform: -36..-1 reads (one ds_read_b32 with an immediate offset each — in pairs they would
    // become ds_read2_b32, whose 8-bit offsets need a VALU add per pair), 18..20 columns, 21..26 rows, 27..44 stores (two each)
    auto t_step = [&](const float *raw, int vb, int k) {
        if (k < 0) {
            const int e = k + 36, r = e / 6, c = e % 6;
            if (!(ABL & 2)) d2[r][c >> 1][c & 1] = raw[rbase + (r * WG::IW + c) * WG::PS];
        } else if (k < 21) {
            const int c = k - 18;
            if (!(ABL & 1)) bt6v(d2[0][c], d2[1][c], d2[2][c], d2[3][c], d2[4][c], d2[5][c]);
        } else if (k < 27) {
            const int r = k - 21;
            if (!(ABL & 1)) bt6row(d2[r][0], d2[r][1], d2[r][2]);
        } else {
            const int r = (k - 27) / 3, j = (k - 27) % 3;             // pair j of row r holds frequencies (0, 5), (1, 2), (3, 4) of that row
            const int f0 = r * 6 + (j == 0 ? 0 : j == 1 ? 1 : 3), f1 = r * 6 + (j == 0 ? 5 : j == 1 ? 2 : 4);
            if (!(ABL & 4)) {
                lds[vwoff + vb + f0 * 256] = d2[r][j].x;
                lds[vwoff + vb + f1 * 256] = d2[r][j].y;
            }
        }
    };
    constexpr int T_FIRST = -36, T_STEPS = 45;
    constexpr int BAR_M = 135;                                 // MFMA slot of the per-stage barrier (stage_body); 143 = after the last MFMA

    // ---- A operand (weights): [group][wave][chunk][frequency][lane][4]; ring of 12 frequencies, fetched 10 ahead, one fragment per
    // four MFMAs.  Buffer loads: the lane offset is one constant VGPR and the fragment offset an SGPR — no address arithmetic on
    // the vector pipe (the global_load form needed a v_mov per load to stay in its saddr form, 12 cycles each beside the MFMAs)
    const float *const wbase = a.wp_w4 + ((size_t)(g * 4 + wv) * n) * (36 * 256);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wbase), 0, (unsigned)n * (36 * 1024), 0x00020000);
    const unsigned wvoff = lane * 16;
    float4 Wq[12];
    constexpr int WLEAD = (ABL & 512) ? 6 : 10;                // frequencies ahead (probe 512: the sensitivity to that distance)
    auto wload1 = [&](int slot, int chunk, int fq) {
        Wq[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (chunk * 36 + fq) * 1024, 0));
    };
    // ---- B operand: Vbuf[frequency][tile t16][slot]; ring of 6 frequencies, fetched 4 ahead
    const int t16 = lane & 15, kl = lane >> 4;
    const int vlane = t16 * 16 + 4 * (kl ^ ((t16 >> 1) & 3));
    float4 Bq[6];
    auto bload1 = [&](int slot, int vb, int fq) {
        Bq[slot] = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(lds + WG::V0 + vb + fq * 256 + vlane, 16));
    };

    f32x4 acc[36];
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(a.outH * a.outW * a.out_cstride) * 4u, 0x00020000);
    // the second [outH][outW][Cout] tensor of a launch: the residual (read) or, for the training path's linear launches, the
    // gated output (written)
    float *const aux = LIN ? a.out_gated : const_cast<float *>(a.residual);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(aux ? aux : a.out, 0, (unsigned)(a.outH * a.outW * a.Cout) * 4u, 0x00020000);

    // ---- prologue: raw(0), raw(1) -> LDS, raw(2) -> registers, V(0), the first ten weight fragments, the first four B operands
    set_patch();
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) gload1(i);
#pragma unroll
    for (int j = 0; j < WLEAD; ++j) wload1(j, 0, j);
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) lwrite1(i, 0);
    advance();
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) gload1(i);
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) lwrite1(i, WG::BUF);
    advance();
#pragma unroll
    for (int i = 0; i < WG::NI; ++i) gload1(i);
    advance();
    __syncthreads();
#pragma unroll
    for (int k = T_FIRST; k < T_STEPS; ++k)
        if (k < 0 || k >= 18) t_step(lds, 0, k);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) bload1(j, 0, j);

    int raw_cur = 0, raw_nxt = WG::BUF;                        // raw buffers: this chunk's (already transformed: free) / the next chunk's
    int v_cur = 0, v_nxt = WG::VBUF;

    // One stage = one 16-channel chunk: 144 MFMAs; beside them: the transform of chunk + 1 (raw_nxt -> v_nxt), raw(chunk + 2)
    // registers -> LDS (into raw_cur, whose chunk was transformed during the previous stage) and raw(chunk + 3) -> registers.
    auto stage_body = [&](auto first_tag, int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int nchunk = chunk + 1;                                       // wraps into the next unit (same weights)
        nchunk = nchunk == n ? 0 : nchunk;
        const float *traw = lds + raw_nxt;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 18; ++pr)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    const int fq = 2 * pr + sidx, m = pr * 8 + e * 2 + sidx;
                    const float4 wv4 = Wq[fq % 12], vv4 = Bq[fq % 6];
                    const float we = e == 0 ? wv4.x : e == 1 ? wv4.y : e == 2 ? wv4.z : wv4.w;
                    const float ve = e == 0 ? vv4.x : e == 1 ? vv4.y : e == 2 ? vv4.z : vv4.w;
                    if (FIRST && e == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[fq] = __builtin_amdgcn_mfma_f32_16x16x4f32(we, ve, zero, 0, 0, 0);
                    } else
                        acc[fq] = __builtin_amdgcn_mfma_f32_16x16x4f32(we, ve, acc[fq], 0, 0, 0);
                    // ---- shadow items
                    const int mm = e * 2 + sidx;                                     // position inside this pair's 8 MFMAs
                    if (!(ABL & 64) && mm < 2 && 2 * pr + 4 + mm < 36) bload1((2 * pr + 4 + mm) % 6, v_cur, 2 * pr + 4 + mm);   // B operands 4 ahead
                    if (!(ABL & 32) && (mm == 2 || mm == 6)) {                       // weights 10 frequencies ahead
                        const int wf = 2 * pr + WLEAD + (mm == 6);
                        if (wf < 36) wload1(wf % 12, chunk, wf);
                        else wload1(wf % 12, nchunk, wf - 36);
                    }
                    // the next chunk's transform: reads first (one per MFMA), arithmetic, stores; then the raw patch traffic
                    if (m < 36) t_step(traw, v_nxt, m - 36);
                    if (m >= 44 && m < 53) t_step(traw, v_nxt, 18 + (m - 44));
                    if (m >= 58 && m < 94 && !(m & 1)) t_step(traw, v_nxt, 27 + ((m - 58) >> 1));
                    if (!(ABL & 16) && m >= 96 && m - 96 < WG::NI) lwrite1(m - 96, raw_cur);        // raw(chunk + 2): registers -> LDS
                    if (!(ABL & 8) && m >= 104 && m - 104 < WG::NI) gload1(m - 104);               // raw(chunk + 3) -> registers
                    // The stage's barrier, eight MFMAs BEFORE its end: V(chunk + 1) and raw(chunk + 2) are complete in every
                    // wave (last LDS write at m = 101), every wave has fetched its last B operand of this chunk (m = 121), and
                    // the first B operands of the next chunk travel under the remaining MFMAs (frequencies 34, 35: ring slots
                    // 4, 5) instead of behind the barrier
                    if (m == BAR_M) {
                        if (!(ABL & 256)) __syncthreads();
                        if (!(ABL & 64)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) bload1(j, v_nxt, j);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            advance();
        const int o = raw_cur;
        raw_cur = raw_nxt;
        raw_nxt = o;
        const int v = v_cur;
        v_cur = v_nxt;
        v_nxt = v;
    };

    for (int u = blockIdx.x; u < a.n_units; u += G) {
        stage_body(std::true_type{}, 0);
        for (int chunk = 1; chunk < n; ++chunk) stage_body(std::false_type{}, chunk);

        // ================= unit epilogue (lane-local) =================
        // D layout: lane (t16, q = lane >> 4), register r = MFMA row 4q + r: q = 0, 1 -> conv_f of channels 4q + r, q = 2, 3 -> conv_m
        // of channels 4 (q - 2) + r; column = tile t16.  Y = A^T M A with A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
        __builtin_amdgcn_s_setprio(1);
        const int cq = (lane >> 4) & 1, hf = lane >> 5;
        const int c0 = g * 32 + wv * 8 + 4 * cq;
        const int oy = by * 8 + 4 * (t16 >> 3) + 2 * hf, ox = bx * 32 + 4 * (t16 & 7);           // this lane finishes rows oy, oy + 1
        if (ABL & 128) {                                               // keep the accumulators live with one store per lane
            f32x4 sum = acc[0];
#pragma unroll
            for (int i = 1; i < 36; ++i) sum += acc[i];
            if (oy < a.outH && ox < a.outW) *reinterpret_cast<f32x4 *>(a.out + ((size_t)oy * a.outW + ox) * a.out_cstride + c0) = sum;
            step_tile(by, bx);
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        const f32x4 bf = *reinterpret_cast<const f32x4 *>(a.params + c0);
        const f32x4 bm = *reinterpret_cast<const f32x4 *>(a.params + a.CoutPad + c0);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.params + 2 * a.CoutPad + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.params + 3 * a.CoutPad + c0);
        // residual loads and output stores through buffer descriptors: a 32-bit lane offset per pixel, out of range for pixels
        // outside the image (loads return zeros, stores are dropped): no 64-bit address arithmetic, no masks, no separate path for
        // partial blocks or a padded last channel group
        unsigned rvoff[2][4], ovoff[2][4];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const bool in = (oy + py < a.outH) & (ox + px < a.outW) & (c0 < a.Cout);      // Cout % 8 == 0: a lane's quad is real or padding
                const int pix = (oy + py) * a.outW + ox + px;
                rvoff[py][px] = in ? (unsigned)((pix * a.Cout + c0) * 4) : OOR;
                ovoff[py][px] = in ? (unsigned)((pix * a.out_cstride + c0) * 4) : OOR;
            }
        f32x4 rv[2][4];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                rv[py][px] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.residual)
                    rv[py][px] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvoff[py][px], 0, 0));
            }
        // rows: R[p][nu] from M[0..5][nu]; then columns: Y[p][0..3] from R[p][0..5]
        f32x4 Y[4][4];
        {
            f32x4 R[4][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const f32x4 s1 = acc[6 + nu] + acc[12 + nu], d1 = pk_sub4(acc[6 + nu], acc[12 + nu]);
                const f32x4 s2 = acc[18 + nu] + acc[24 + nu], d2 = pk_sub4(acc[18 + nu], acc[24 + nu]);
                R[0][nu] = acc[nu] + s1 + s2;
                R[1][nu] = d1 + 2.0f * d2;
                R[2][nu] = s1 + 4.0f * s2;
                R[3][nu] = d1 + 8.0f * d2 + acc[30 + nu];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4 s1 = R[p][1] + R[p][2], d1 = pk_sub4(R[p][1], R[p][2]);
                const f32x4 s2 = R[p][3] + R[p][4], d2 = pk_sub4(R[p][3], R[p][4]);
                Y[p][0] = R[p][0] + s1 + s2;
                Y[p][1] = d1 + 2.0f * d2;
                Y[p][2] = s1 + 4.0f * s2;
                Y[p][3] = d1 + 8.0f * d2 + R[p][5];
            }
        }
        if constexpr (LIN) {
            // training path: the pre-activations conv_f + b_f (channel c) and conv_m + b_m (channel Cout + c) of all four rows of
            // the tile — lanes 0..31 hold f, lanes 32..63 m — before anything is gated
            const f32x4 bb = hf ? bm : bf;
            const int oyt = by * 8 + 4 * (t16 >> 3), chan = (hf ? a.Cout : 0) + c0;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const bool in = (oyt + p < a.outH) & (ox + px < a.outW) & (c0 < a.Cout);
                    const unsigned vo = in ? (unsigned)((((oyt + p) * a.outW + ox + px) * a.out_cstride + chan) * 4) : OOR;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Y[p][px] + bb), out_rsrc, vo, 0, 0);
                }
        }
        if constexpr (LIN == 2) {                                      // dgrad: nothing to gate
            step_tile(by, bx);
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        // lanes 0..31 hold conv_f, lanes 32..63 conv_m: exchange rows (py, py + 2) so that the lower half-wave owns rows 0, 1 and
        // the upper half rows 2, 3 of the tile, f in one register and m in the other
        f32x4 Yf[2][4], Ym[2][4];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                u32x4 u0 = __builtin_bit_cast(u32x4, Y[py][px]), u1 = __builtin_bit_cast(u32x4, Y[py + 2][px]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                    u0[k] = sw[0];
                    u1[k] = sw[1];
                }
                Yf[py][px] = __builtin_bit_cast(f32x4, u0);
                Ym[py][px] = __builtin_bit_cast(f32x4, u1);
            }
        {
            constexpr float LOG2E = 1.44269504088896341f;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    f32x4 f = Yf[py][px] + bf;
                    const f32x4 mm = (Ym[py][px] + bm) * -LOG2E;
                    if (a.elu) {                                   // vector forms wherever an operation has one: they become v_pk_*
                        const f32x4 fe = f * LOG2E;
                        f32x4 e;
#pragma unroll
                        for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f(fe[k]);
                        e = e + f32x4{-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                        for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : e[k];
                    }
                    f32x4 sg, t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_exp2f(mm[k]);
                    t = t + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(t[k]);
                    f32x4 v = (f * sg) * sc + sh + rv[py][px];
                    if constexpr (LIN) {                               // the gated output, zero on the separator rows of a stacked batch
                        if (a.blk_h > 0 && (oy + py) % a.blk_h >= a.blk_valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), res_rsrc, rvoff[py][px], 0, 0);
                        continue;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, ovoff[py][px], 0, 0);
                }
        }
        step_tile(by, bx);
        __builtin_amdgcn_s_setprio(0);
    }
    if (ABL && a.nchunks == -12345) {                                  // never true: keeps the probes' dead values alive
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i) sink += d2[i / 3][i % 3].x + d2[i / 3][i % 3].y;
#pragma unroll
        for (int i = 0; i < WG::NI; ++i) sink += st[i].x + st[i].y + st[i].z + st[i].w;
        a.out[tid] = sink;
    }
}

// ------------------------------------------------------------------------------------------
// Winograd F(4x4,3x3) with SPLIT fp32 operands on the f16 matrix cores (round 6): v_mfma_f32_16x16x32_f16, fp32 accumulation.
//
// Why: the kernel above runs on v_mfma_f32_16x16x4_f32, the 157 TF path that shares the FP32 vector pipe (40 % of its peak); the
// f16 / bf16 path is 16x faster per instruction-flop and overlaps with plain vector instructions (profiles/r5_issue_probe_bf16.md).
// How the fp32 values get through an 11-bit significand: every operand is split into TWO f16 pieces
//     V = Vh + 2^-11 Vl      Vh = f16(V),  Vl = f16((V - Vh) 2^11)        (input transform, per launch)
//     U' = Uh + Ul           Uh = f16(U s), Ul = f16(U s - Uh)            (host packer; s = a power of two per output row that
//                                                                          puts max |U s| in [2^14, 2^15): Ul stays a normal number)
// and the product is the THREE largest piece pairs, summed by three MFMAs into one fp32 accumulator:
//     U' V ~= (2^-11 Uh) Vl + Ul Vh + Uh Vh        (2^-11 Uh is formed in registers: v_pk_mul_f16, exact)
// hi + lo carries 22 significant bits (2^-22 relative, RTN), the dropped pair is 2^-22 too; the matrix core sums a 32-channel block
// before it rounds to fp32 once — measured ON the device against fp64 (tools/probes/f16split_probe.hip, profiles/r6_f16split_probe.txt):
// relative rms error 0.9e-7 / 1.8e-7 / 3.1e-7 at K = 32 / 256 / 1024 against 1.1e-7 / 2.8e-7 / 5.9e-7 for the fp32 MFMA chain —
// the split product is MORE accurate than the fp32 instruction it replaces, from |V| ~ 1e-4 up to the f16 overflow at |V| = 65504,
// i.e. activations up to ~650 (B^T d B amplifies by at most 100); below ~1e-4 the absolute error floors at ~1e-11.
// The scaled low piece is what makes it two pieces instead of three (bf16 x 3: six MFMAs, 1.5x the weight bytes).
//
// Shape: the unit, grid walk, weight ownership (wave w = output channels 8w .. 8w+7, rows 0..7 conv_f, 8..15 conv_m) and the
// lane-local output transform / gate epilogue are the fp32 kernel's.  What differs:
//   * chunk = 32 input channels = ONE MFMA k-block: 36 frequencies x 3 MFMAs per chunk and wave (fp32 kernel: 288 for 32 channels);
//   * A operand: [group][wave][chunk][frequency][piece Uh | Ul][lane][8 halfs], lane (row = lane & 15, k = 8 (lane >> 4) ..+7):
//     two 1 KiB buffer loads per frequency, eight frequencies ahead, ring of twelve; then 2 CoutPad floats 1 / s (f rows, m rows);
//   * B operand: V[buffer 2][frequency 36][piece 2][tile 16][slot 4][8 halfs] in LDS = 147,456 bytes — ALL of the LDS budget, so
//     the raw patches are not staged through LDS: transform thread (tile, channel PAIR) loads its 6 x 6 x 2 patch straight from
//     L2 (36 buffer_load_dwordx2 with a per-lane offset table; pixels outside the image carry an out-of-range offset and arrive
//     as zeros), one chunk ahead of its transform, into the registers the transform works in;
//   * transform in packed fp32 over the channel pair (168 v_pk_*), split 5 instructions per frequency (v_cvt_pk_f16_f32,
//     2 v_fma_mix_f32, v_pk_mul_f32, v_cvt_pk_f16_f32), one ds_write2st64_b32 per frequency (both pieces);
//   * FAM's multiply (x1 * x2 in the loader) is NOT taken: those three launches stay on the fp32 kernel (conv_uses_w4h);
//   * slot swizzle slot = q ^ ((-(tile >> 2)) & 3): the four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) each cover
//     all 64 banks, and the 32-lane groups of the transform's stores cover the 32 store banks.
struct Wino4hGeom {
    static constexpr int VFREQ = 512;                          // dwords per frequency: 2 pieces x 16 tiles x 64 bytes
    static constexpr int VBUF = 36 * VFREQ;                    // dwords per V buffer (73,728 bytes)
    static constexpr int LDS_DWORDS = 2 * VBUF;                // 147,456 bytes
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ABL (attribution probes, results invalid; -DREAD_DEBUG_KNOBS builds only, read_tuning_set("conv_abl")): 1 no transform arithmetic,
// 2 no split arithmetic, 4 no V stores, 8 no patch loads, 32 weights loaded once, 64 B operands loaded once, 128 no epilogue,
// 256 no barrier, 512 no 2^-11 Uh products, 1024 no MFMAs, 2048 / 4096 patch / weight loads from one cache-resident address,
// epilogue: 8192 no residual loads, 16384 one store instead of eight, 32768 no exp / rcp, 65536 no output transform
template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void gated_conv_wino4h_kernel(const ConvKArgs a)
{
    using WG = Wino4hGeom;
    __shared__ __attribute__((aligned(16))) unsigned lds[WG::LDS_DWORDS];
    __shared__ __attribute__((aligned(16))) float epar[6][32];  // the group's epilogue parameters: b_f, -log2e b_m, BN scale, BN shift, 1 / s_f, -log2e / s_m
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;                                   // 32-channel chunks

    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit (8 x 32 pixel block)
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // prefetch cursor (raw patches)
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    // ---- transform role: thread = (tile tl, channel pair cp of the 32-channel chunk); wave w = tiles 4w .. 4w+3 (one tile row of
    // the unit per wave: the patch ROWS are wave-uniform — their offsets travel in SGPRs — the patch COLUMNS are per lane)
    const int cp = lane & 15, tl = wv * 4 + (lane >> 4);
    constexpr unsigned OOR = 0x80000000u;
    unsigned xoff[6];                                          // byte offset of patch column c, channel pair cp (a pixel row, chunk 0);
                                                               // columns outside the image: out of range, the load returns zeros
    int rowoff[6];                                             // byte offset of patch row r, clamped into the image ...
    float rmask[6], rmask_d[6];                                // ... and 0 where it was clamped: rows of the cursor's unit / of the patch in d2
    bool rclamp = false, rclamp_d = false;                     // any row clamped (top / bottom units only)
    const unsigned src_bytes = (unsigned)(a.inH * s.W * s.C) * 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, src_bytes, 0x00020000);
    auto set_patch = [&]() {
        const int y0 = pby * 8 + 4 * (wv >> 1) - 1, x0 = pbx * 32 + 4 * (tl & 7) - 1;
        rclamp = false;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int yy = y0 + r;
            const bool ok = (unsigned)yy < (unsigned)a.inH;
            const int yc = yy < 0 ? 0 : yy >= a.inH ? a.inH - 1 : yy;
            rowoff[r] = yc * s.W * s.C * 4;
            rmask[r] = ok ? 1.0f : 0.0f;
            rclamp = rclamp || !ok;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) xoff[c] = (unsigned)(x0 + c) < (unsigned)a.inW ? (unsigned)((x0 + c) * s.C + 2 * cp) * 4u : OOR;
    };
    auto advance = [&]() {
#pragma unroll
        for (int r = 0; r < 6; ++r) rmask_d[r] = rmask[r];     // the patch just loaded is the next one to be transformed
        rclamp_d = rclamp;
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    f32x2 d2[6][6];                                            // the patch of the chunk under transform: (channel 2 cp, 2 cp + 1)
    auto gload = [&](int r, int c) {
        if (ABL & 8) return;
        if (ABL & 2048) {                                      // every patch load from the tensor's first 128 bytes: the instruction without its miss
            d2[r][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, cp * 8u, 0, 0));
            return;
        }
        d2[r][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, xoff[c], rowoff[r] + pchunk * 128, 0));
    };
    // 1-D transform with B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1] on six
    // channel pairs: 14 packed operations
    auto bt6 = [](f32x2 &x0, f32x2 &x1, f32x2 &x2, f32x2 &x3, f32x2 &x4, f32x2 &x5) {
        const f32x2 p = pk_add(x3, x4), q = pk_add(x1, x2), r = pk_sub(x4, x3), u = pk_sub(x1, x2), f = pk_sub(x3, x1), h = pk_sub(x4, x2);
        const f32x2 y0 = __builtin_elementwise_fma(x2, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x0, f32x2{4.0f, 4.0f}, x4));
        const f32x2 y5 = __builtin_elementwise_fma(x3, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x1, f32x2{4.0f, 4.0f}, x5));
        x0 = y0;
        x1 = __builtin_elementwise_fma(q, f32x2{-4.0f, -4.0f}, p);
        x2 = __builtin_elementwise_fma(u, f32x2{4.0f, 4.0f}, r);
        x3 = __builtin_elementwise_fma(f, f32x2{2.0f, 2.0f}, h);
        x4 = __builtin_elementwise_fma(f, f32x2{-2.0f, -2.0f}, h);
        x5 = y5;
    };
    // V store address (dwords) of this thread: + buffer + frequency * 512 (+ 256: the low piece)
    const int vwoff = tl * 16 + (((cp >> 2) ^ ((-wv) & 3)) << 2) + (cp & 3);
    auto split_store = [&](const f32x2 x, int vb, int fq) {
        unsigned hi, lo;
        float r0, r1;
        if (ABL & 2) {
            if (!(ABL & 4)) {
                lds[vwoff + vb + fq * WG::VFREQ] = __builtin_bit_cast(unsigned, x.x);
                lds[vwoff + vb + fq * WG::VFREQ + 256] = __builtin_bit_cast(unsigned, x.y);
            }
            return;
        }
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x.x), "v"(x.y));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x.x));                    // x - f32(hi), exact
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x.y));
        const f32x2 rs = f32x2{r0, r1} * f32x2{2048.0f, 2048.0f};
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(rs.x), "v"(rs.y));
        if (ABL & 4) {
            asm volatile("" :: "v"(hi), "v"(lo));
            return;
        }
        lds[vwoff + vb + fq * WG::VFREQ] = hi;
        lds[vwoff + vb + fq * WG::VFREQ + 256] = lo;
    };
    // step k of a chunk's transform: 0..5 the column pass of column k; then per row r seven steps: the row pass, six
    // frequencies (split + store); the row's registers are then refilled with the patch of the chunk after
    constexpr int T_STEPS = 48;
    auto t_step = [&](int vb, int k, bool reload) {
        if (k < 6) {
            if (rclamp_d) {                                    // wave-uniform: top / bottom units zero the rows outside the image
#pragma unroll
                for (int r = 0; r < 6; ++r) d2[r][k] = d2[r][k] * f32x2{rmask_d[r], rmask_d[r]};
            }
            if (!(ABL & 1)) bt6(d2[0][k], d2[1][k], d2[2][k], d2[3][k], d2[4][k], d2[5][k]);
        } else {
            const int r = (k - 6) / 7, j = (k - 6) % 7;
            if (j == 0) {
                if (!(ABL & 1)) bt6(d2[r][0], d2[r][1], d2[r][2], d2[r][3], d2[r][4], d2[r][5]);
            }
            else {
                split_store(d2[r][j - 1], vb, r * 6 + j - 1);
                if (j == 6 && reload) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) gload(r, c);
                }
            }
        }
    };

    // ---- A operand (weights)
    const char *const wbase = reinterpret_cast<const char *>(a.wp_w4h) + ((size_t)(g * 4 + wv) * n) * (36 * 2048);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, (unsigned)n * (36 * 2048), 0x00020000);
    const unsigned wvoff = lane * 16;
    constexpr int RW = 9, WLEAD = 6;                           // ring of nine frequencies, fetched six ahead (36 % RW == 0: static slots)
    u32x4 Wh[RW], Wl[RW];
    f16x8 Ws[4];                                               // 2^-11 Uh, one frequency pair ahead
    auto wload = [&](int slot, int chunk, int fq) {
        if (ABL & 4096) {                                      // every weight load from one 2 KiB fragment: the instruction without its L2 traffic
            Wh[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, 0, 0);
            Wl[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, 1024, 0);
            return;
        }
        Wh[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (chunk * 36 + fq) * 2048, 0);
        Wl[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (chunk * 36 + fq) * 2048 + 1024, 0);
    };
    auto wscale = [&](int fq) {
        const _Float16 k = (_Float16)0x1p-11f;
        Ws[fq % 4] = __builtin_bit_cast(f16x8, Wh[fq % RW]) * f16x8{k, k, k, k, k, k, k, k};
    };
    // ---- B operand
    const int t16 = lane & 15, kl = lane >> 4;
    const int vrd = t16 * 16 + ((kl ^ ((-(t16 >> 2)) & 3)) << 2);
    constexpr int RB = 4;                                      // ring of four frequencies, fetched one pair ahead
    u32x4 Bh[RB], Bl[RB];
    auto bload = [&](int slot, int vb, int fq) {
        Bh[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + vb + fq * WG::VFREQ + vrd, 16));
        Bl[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + vb + fq * WG::VFREQ + 256 + vrd, 16));
    };

    f32x4 acc[36];
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(a.outH * a.outW * a.out_cstride) * 4u, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.residual ? a.residual : a.out), 0,
                                                            (unsigned)(a.outH * a.outW * a.Cout) * 4u, 0x00020000);
    const float *const wsc = reinterpret_cast<const float *>(a.wp_w4h) + (size_t)n * 2304 * a.CoutPad;       // 1 / s: [f | m][CoutPad]
    // the group is fixed per workgroup: its epilogue parameters go to LDS once.  (Loaded from global memory in every unit's epilogue
    // they queued behind the next unit's weight and patch loads, already in flight, and the epilogue waited for those: loads return in
    // order.  Measured on one box, old / new library alternating: 64.9 / 55.3 / 49.2 / 47.3 -> 64.0 / 52.4 / 48.5 / 47.3 us at C = 32 .. 256.)
    if (tid < 192) {
        constexpr float L2E = 1.44269504088896341f;
        const int arr = tid >> 5, c = g * 32 + (tid & 31);
        const float v = arr < 4 ? a.params[arr * a.CoutPad + c] : wsc[(arr - 4) * a.CoutPad + c];
        epar[arr][tid & 31] = (arr == 1 || arr == 5) ? v * -L2E : v;
    }

    // ---- prologue: patch(0) -> registers -> V(0); patch(1) -> registers; the first weight fragments and B operands
    set_patch();
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) gload(r, c);
#pragma unroll
    for (int j = 0; j < WLEAD; ++j) wload(j, 0, j);
    advance();
#pragma unroll
    for (int k = 0; k < T_STEPS; ++k) t_step(0, k, true);
    advance();
    wscale(0);
    wscale(1);
    __syncthreads();
    bload(0, 0, 0);
    bload(1, 0, 1);

    int v_cur = 0, v_nxt = WG::VBUF;
    constexpr int BAR_M = 102;                                 // the stage's barrier: in front of the first MFMA of frequency pair 17

    // One stage = one 32-channel chunk: 108 MFMAs (18 frequency pairs x 3 piece pairs x 2); beside them the transform of
    // chunk + 1 (registers -> v_nxt) and the loads of patch(chunk + 2) into the registers the transform leaves behind.
    auto stage_body = [&](auto first_tag, int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int nchunk = chunk + 1;                                       // wraps into the next unit (same weights)
        nchunk = nchunk == n ? 0 : nchunk;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 18; ++pr)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int fq = 2 * pr + (j & 1), pc = j >> 1, m = pr * 6 + j;
                if (m == BAR_M && !(ABL & 256)) {
                    // V(chunk + 1) is complete in every wave (last store at slot 94) and every wave has fetched its last B
                    // operand of this chunk (slot 97); the first B operands of the next chunk travel under the last six MFMAs
                    __syncthreads();
                }
                const f16x8 av = pc == 0 ? Ws[fq % 4] : __builtin_bit_cast(f16x8, pc == 1 ? Wl[fq % RW] : Wh[fq % RW]);
                const f16x8 bv = __builtin_bit_cast(f16x8, pc == 0 ? Bl[fq % RB] : Bh[fq % RB]);
                if (ABL & 1024) {
                    if (FIRST && pc == 0) acc[fq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (pc == 2) asm volatile("" :: "v"(av), "v"(bv));
                } else if (FIRST && pc == 0) {
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    acc[fq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, zero, 0, 0, 0);
                } else
                    acc[fq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[fq], 0, 0, 0);
                // ---- shadow items
                if (j < 2) {                                                         // B operands one frequency pair ahead
                    const int bf = 2 * pr + 2 + j;
                    if (ABL & 64) {
                    } else if (bf < 36) bload(bf % RB, v_cur, bf);
                    else bload(bf % RB, v_nxt, bf - 36);
                } else if (j < 4) {                                                  // weights WLEAD frequencies ahead
                    const int wf = 2 * pr + WLEAD + (j - 2);
                    if (ABL & 32) {
                    } else if (wf < 36) wload(wf % RW, chunk, wf);
                    else wload(wf % RW, nchunk, wf - 36);
                } else if (!(ABL & 512))
                    wscale((2 * pr + 2 + (j - 4)) % 36);                              // 2^-11 Uh of the next frequency pair
                if (!(m & 1) && (m >> 1) < T_STEPS) t_step(v_nxt, m >> 1, true);
                __builtin_amdgcn_sched_barrier(0);
            }
        advance();
        const int v = v_cur;
        v_cur = v_nxt;
        v_nxt = v;
    };

    for (int u = blockIdx.x; u < a.n_units; u += G) {
        stage_body(std::true_type{}, 0);
        for (int chunk = 1; chunk < n; ++chunk) stage_body(std::false_type{}, chunk);

        // ================= unit epilogue (lane-local; the fp32 kernel's, with the rows' 1 / s folded into the bias FMAs) =================
        __builtin_amdgcn_s_setprio(1);
        const int cq = (lane >> 4) & 1, hf = lane >> 5;
        const int c0 = g * 32 + wv * 8 + 4 * cq;
        const int oy = by * 8 + 4 * (t16 >> 3) + 2 * hf, ox = bx * 32 + 4 * (t16 & 7);           // this lane finishes rows oy, oy + 1
        if (ABL & 128) {                                               // keep the accumulators live with one store per lane
            f32x4 sum = acc[0];
#pragma unroll
            for (int i = 1; i < 36; ++i) sum += acc[i];
            if (oy < a.outH && ox < a.outW) *reinterpret_cast<f32x4 *>(a.out + ((size_t)oy * a.outW + ox) * a.out_cstride + c0) = sum;
            step_tile(by, bx);
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        const int cl = wv * 8 + 4 * cq;
        const f32x4 bf = *reinterpret_cast<const f32x4 *>(&epar[0][cl]), bml = *reinterpret_cast<const f32x4 *>(&epar[1][cl]);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(&epar[2][cl]), sh = *reinterpret_cast<const f32x4 *>(&epar[3][cl]);
        const f32x4 isf = *reinterpret_cast<const f32x4 *>(&epar[4][cl]), ism = *reinterpret_cast<const f32x4 *>(&epar[5][cl]);
        constexpr float LOG2E = 1.44269504088896341f;
        unsigned rvoff[2][4], ovoff[2][4];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const bool in = (oy + py < a.outH) & (ox + px < a.outW) & (c0 < a.Cout);
                const int pix = (oy + py) * a.outW + ox + px;
                rvoff[py][px] = in ? (unsigned)((pix * a.Cout + c0) * 4) : OOR;
                ovoff[py][px] = in ? (unsigned)((pix * a.out_cstride + c0) * 4) : OOR;
            }
        f32x4 rv[2][4];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                rv[py][px] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.residual && !(ABL & 8192))
                    rv[py][px] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvoff[py][px], 0, 0));
            }
        f32x4 Y[4][4];
        if (ABL & 65536) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Y[i >> 2][i & 3] = acc[i] + acc[16 + i] + acc[20 + i];
        } else {
            f32x4 R[4][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const f32x4 s1 = acc[6 + nu] + acc[12 + nu], d1 = pk_sub4(acc[6 + nu], acc[12 + nu]);
                const f32x4 s2 = acc[18 + nu] + acc[24 + nu], dd2 = pk_sub4(acc[18 + nu], acc[24 + nu]);
                R[0][nu] = acc[nu] + s1 + s2;
                R[1][nu] = d1 + 2.0f * dd2;
                R[2][nu] = s1 + 4.0f * s2;
                R[3][nu] = d1 + 8.0f * dd2 + acc[30 + nu];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4 s1 = R[p][1] + R[p][2], d1 = pk_sub4(R[p][1], R[p][2]);
                const f32x4 s2 = R[p][3] + R[p][4], dd2 = pk_sub4(R[p][3], R[p][4]);
                Y[p][0] = R[p][0] + s1 + s2;
                Y[p][1] = d1 + 2.0f * dd2;
                Y[p][2] = s1 + 4.0f * s2;
                Y[p][3] = d1 + 8.0f * dd2 + R[p][5];
            }
        }
        f32x4 Yf[2][4], Ym[2][4];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                u32x4 u0 = __builtin_bit_cast(u32x4, Y[py][px]), u1 = __builtin_bit_cast(u32x4, Y[py + 2][px]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                    u0[k] = sw[0];
                    u1[k] = sw[1];
                }
                Yf[py][px] = __builtin_bit_cast(f32x4, u0);
                Ym[py][px] = __builtin_bit_cast(f32x4, u1);
            }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                f32x4 f = __builtin_elementwise_fma(Yf[py][px], isf, bf);
                const f32x4 mm = __builtin_elementwise_fma(Ym[py][px], ism, bml);
                if (a.elu) {
                    const f32x4 fe = f * LOG2E;
                    f32x4 e;
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = (ABL & 32768) ? fe[k] * 0.5f : __builtin_amdgcn_exp2f(fe[k]);
                    e = e + f32x4{-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : e[k];
                }
                f32x4 sg, t;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = (ABL & 32768) ? mm[k] * 0.25f : __builtin_amdgcn_exp2f(mm[k]);
                t = t + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) sg[k] = (ABL & 32768) ? t[k] * 0.125f : __builtin_amdgcn_rcpf(t[k]);
                const f32x4 v = (f * sg) * sc + sh + rv[py][px];
                if ((ABL & 16384) && (py | px)) {
                    asm volatile("" :: "v"(v));
                    continue;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, ovoff[py][px], 0, 0);
            }
        step_tile(by, bx);
        __builtin_amdgcn_s_setprio(0);
    }
    if (ABL && a.nchunks == -12345) {                                  // never true: keeps the probes' dead values alive
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 36; ++i) sink += d2[i / 6][i % 6].x + d2[i / 6][i % 6].y;
        a.out[tid] = sink;
    }
}

// ------------------------------------------------------------------------------------------
// DIRECT 3x3 / stride-1 gated convolution with split fp32 operands on the f16 matrix cores (round 6, second kernel of the round).
//
// Why: the attribution of the split-operand Winograd kernel above (profiles/r6_w4h_ablation.md) — its MFMAs are a fifth of its launch,
// the rest is what Winograd costs around them: an input transform of 350 vector instructions per (tile, channel pair), 36 / 16
// times the operand volume of the plain product (288 KiB of weight fragments per 32 channels and unit through one CU's load path),
// a 144-register accumulator set that pins one wave per SIMD, and an output transform in the epilogue.  Winograd trades 4x fewer
// multiplications for all that; on v_mfma_f32_16x16x32_f16 a multiplication costs a sixteenth of what it costs on the fp32 cores,
// so the trade runs the other way: this kernel executes ALL 9 taps (4x the MFMAs of F(4x4): 22.9 us per launch at the measured
// 8.1 ns per MFMA, the same at every level) and needs none of the above —
//   * operands as in the Winograd kernel: x = xh + 2^-11 xl (f16 pieces, formed ONCE per input element while the patch is staged into
//     LDS: 10 vector instructions per float4), w s = wh + wl (host packer, power-of-two row scale), products = (2^-11 wh) xl + wl xh
//     + wh xh, three MFMAs into one fp32 accumulator; no transform multiplies the input by up to 100, so the f16 range now covers
//     activations up to 65504 (Winograd: ~650);
//   * unit = 8 x 32 output pixels x 32 output channels (the Winograd kernels' unit, same persistent walk); EIGHT waves, two per SIMD:
//     wave (rh = w & 1, pq = w >> 1) owns 32 MFMA rows (channels 16 rh .. 16 rh + 15: two row blocks of 8 conv_f + 8 conv_m
//     channels) x image rows 2 pq, 2 pq + 1 of the unit (four blocks of 16 consecutive pixels): 32 accumulators;
//   * per 32 input channels: the 10 x 34 patch (fp32, 1360 float4: three coalesced 16-byte loads per thread) -> f16 pieces ->
//     X[buffer 2][pixel 340][slot 8 = (piece, channel octet) ^ (x & 7)][8 halfs] = 43.5 KB per buffer; a tap is a shifted read of it
//     (immediate offsets; ds_read_b128, conflict-free by the swizzle), one B fragment feeds both row blocks;
//   * weights [group][rh][chunk][tap][row block][wh | wl][lane][8 halfs]: 4 KiB per (chunk, tap) and wave for 48 MFMAs (the Winograd
//     kernel: 2 KiB per 3), straight from L2 into registers two taps ahead; the four waves of a row half ask for the same lines.
// FAM's x1 * x2 is a multiplication at staging time.  Epilogue: one v_permlane32_swap per register pair brings conv_f and conv_m of
// a pixel block together (as everywhere), gate / BatchNorm / residual, 16-byte stores — no output transform.
struct D3hGeom {
    static constexpr int IH = 10, IW = 34, NPIX = IH * IW;     // patch of a unit
    static constexpr int XBUF = NPIX * 32;                     // dwords per buffer: 128 bytes per pixel (hi + lo of 32 channels)
    static constexpr int NE = NPIX * 8, NI = (NE + 511) / 512; // float4 of a patch chunk: 2720 -> 6 per thread?  (8 float4 per pixel)
};

// ABL (attribution probes, results invalid; debug library): 1 no patch staging, 2 weights loaded once, 4 B operands read once, 8 no epilogue,
// 16 no barrier, 32 no 2^-11 wh products
template <bool MUL, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gated_conv_d3h_kernel(const ConvKArgs a)
{
    using DG = D3hGeom;
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * DG::XBUF];
    __shared__ __attribute__((aligned(16))) float epar[6][32];  // the group's epilogue parameters: b_f, -log2e b_m, BN scale, BN shift, 1 / s_f, -log2e / s_m
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), rh = wv & 1, pq = wv >> 1;
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;                                   // 32-channel chunks
    constexpr unsigned OOR = 0x80000000u;

    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // staging cursor
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    // ---- staging: thread = float4 e = tid + 512 i of the patch chunk (pixel e >> 3, channel quad e & 7)
    int soff[DG::NI];                                          // LDS dword offset of the hi piece (+ buffer); lo = the same ^ 16 dwords
    unsigned rel[DG::NI], aoff[DG::NI];
#pragma unroll
    for (int i = 0; i < DG::NI; ++i) {
        const int e = tid + i * 512, pix = e >> 3, q4 = e & 7, ppx = pix % DG::IW;
        soff[i] = pix * 32 + ((((q4 >> 1)) ^ (ppx & 7)) << 2) + ((q4 & 1) << 1);
        rel[i] = (unsigned)(((pix / DG::IW) * s.W + ppx) * s.C + 4 * q4) * 4u;
    }
    const unsigned src_bytes = (unsigned)(a.inH * s.W * s.C) * 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, src_bytes, 0x00020000);
    const auto rsrc_mul = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MUL ? a.mul : s.p), 0, src_bytes, 0x00020000);
    auto set_patch = [&]() {
        const int y0 = pby * 8 - 1, x0 = pbx * 32 - 1;
        const unsigned base = (unsigned)((y0 * s.W + x0) * s.C) * 4u;          // may wrap: only pixels inside the image use it
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) {
            const int e = tid + i * 512, pix = e >> 3, ppy = pix / DG::IW, ppx = pix % DG::IW;
            const bool ok = (e < DG::NE) & (ppy >= -y0) & (ppy < a.inH - y0) & (ppx >= -x0) & (ppx < a.inW - x0);
            aoff[i] = ok ? base + rel[i] : OOR;
        }
    };
    auto advance = [&]() {
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    float4 st[DG::NI], stm[MUL ? DG::NI : 1];
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) {
            st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, aoff[i], pchunk * 128, 0));
            if constexpr (MUL) stm[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_mul, aoff[i], pchunk * 128, 0));
        }
    };
    auto split2 = [](float x, float y, unsigned &hi, unsigned &lo) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x));                    // x - f32(hi), exact
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(y));
        const f32x2 rs = f32x2{r0, r1} * f32x2{2048.0f, 2048.0f};
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(rs.x), "v"(rs.y));
    };
    auto lwrite = [&](int xb) {                                 // registers -> f16 pieces -> X[xb]
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) {
            if ((DG::NI - 1) * 512 + 511 >= DG::NE && i == DG::NI - 1 && tid + i * 512 >= DG::NE) continue;      // the last round is partial
            float4 v = st[i];
            if constexpr (MUL) v = make_float4(v.x * stm[i].x, v.y * stm[i].y, v.z * stm[i].z, v.w * stm[i].w);
            unsigned h0, l0, h1, l1;
            split2(v.x, v.y, h0, l0);
            split2(v.z, v.w, h1, l1);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + soff[i], 8)) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + (soff[i] ^ 16), 8)) = u32x2{l0, l1};
        }
    };

    // ---- A operand (weights): [group][rh][chunk][tap][row block 2][piece 2][lane][8 halfs]; ring of three taps, two ahead
    const char *const wbase = reinterpret_cast<const char *>(a.wp_d3h) + ((size_t)(g * 2 + rh) * n) * (9 * 4096);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, (unsigned)n * (9 * 4096), 0x00020000);
    const unsigned wvoff = lane * 16;
    u32x4 Wh[3][2], Wl[3][2];
    auto wload = [&](int slot, int chunk, int tap) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            Wh[slot][rb] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, ((chunk * 9 + tap) * 4 + rb * 2) * 1024, 0);
            Wl[slot][rb] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, ((chunk * 9 + tap) * 4 + rb * 2 + 1) * 1024, 0);
        }
    };
    // ---- B operand: lane (pixel nn of the 16-pixel block, channel octet kq); per tap column dx the swizzled slot differs
    const int nn = lane & 15, kq = lane >> 4;
    int bho[3], blo[3];                                         // dword offsets inside a pixel row segment, hi / lo piece
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        bho[dx] = nn * 32 + ((kq ^ ((nn + dx) & 7)) << 2);
        blo[dx] = nn * 32 + (((4 + kq) ^ ((nn + dx) & 7)) << 2);
    }

    f32x4 acc[2][4];
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(a.outH * a.outW * a.out_cstride) * 4u, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.residual ? a.residual : a.out), 0,
                                                            (unsigned)(a.outH * a.outW * a.Cout) * 4u, 0x00020000);
    const float *const wsc = reinterpret_cast<const float *>(a.wp_d3h) + (size_t)n * 576 * a.CoutPad;       // 1 / s: [f | m][CoutPad]

    // ---- prologue: patch(0) -> X[0]; patch(1) on its way; the first two taps' weights; the epilogue's parameters into LDS (a load
    // from global memory inside a stage would queue behind, and wait for, the weight stream)
    if (tid < 192) {
        constexpr float L2E = 1.44269504088896341f;
        const int arr = tid >> 5, c = g * 32 + (tid & 31);
        const float v = arr < 4 ? a.params[arr * a.CoutPad + c] : wsc[(arr - 4) * a.CoutPad + c];
        epar[arr][tid & 31] = (arr == 1 || arr == 5) ? v * -L2E : v;
    }
    set_patch();
    gload();
    wload(0, 0, 0);
    wload(1, 0, 1);
    lwrite(0);
    advance();
    gload();
    __syncthreads();
    int x_cur = 0, x_nxt = DG::XBUF;

    // B operands of (tap, pixel block): ring of two, fetched one block ahead; A operands: weights two taps ahead (Wh / Wl ring of
    // three), 2^-11 wh of the next tap formed during this one.  hipcc sinks a load to its first use unless the order is pinned:
    // every group of six MFMAs ends in a sched_barrier, the loads sit between the groups.
    u32x4 Bh[2], Bl[2];
    auto bload = [&](int slot, int xb, int tap, int pb) {
        const int dy = tap / 3, dx = tap % 3;
        const int base = xb + ((2 * pq + (pb >> 1) + dy) * DG::IW + 16 * (pb & 1) + dx) * 32;
        Bh[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + base + bho[dx], 16));
        Bl[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + base + blo[dx], 16));
    };
    f16x8 As[2][2];                                            // [tap & 1][row block]
    auto ascale = [&](int tap) {
        const _Float16 k11 = (_Float16)0x1p-11f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) As[tap & 1][rb] = __builtin_bit_cast(f16x8, Wh[tap % 3][rb]) * f16x8{k11, k11, k11, k11, k11, k11, k11, k11};
    };
    auto lwrite1 = [&](int i, int xb) {
        if ((DG::NI - 1) * 512 + 511 >= DG::NE && i == DG::NI - 1 && tid + i * 512 >= DG::NE) return;
        float4 v = st[i];
        if constexpr (MUL) v = make_float4(v.x * stm[i].x, v.y * stm[i].y, v.z * stm[i].z, v.w * stm[i].w);
        unsigned h0, l0, h1, l1;
        split2(v.x, v.y, h0, l0);
        split2(v.z, v.w, h1, l1);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + soff[i], 8)) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + (soff[i] ^ 16), 8)) = u32x2{l0, l1};
    };
    ascale(0);
    bload(0, x_cur, 0, 0);

    // ---- epilogue of a unit: acc[rb][pb], lane (nn, kq), register r = MFMA row 4 kq + r: kq = 0, 1 -> conv_f of channels 4 kq + r of the
    // row block, kq = 2, 3 -> conv_m of channels 4 (kq - 2) + r; column nn = pixel 16 (pb & 1) + nn of unit row 2 pq + (pb >> 1).
    // (Running it as shadow work inside the next unit's first stage — accumulators copied, residual requested at once, one output
    // behind every fourth MFMA group — was built and measured level: 67.9 / 56.3 / 55.3 / 51.0 us against 64.4 / 58.0 / 52.5 / 49.5; what the
    // "no epilogue" probe removes at level 0 is the residual's and the output's 110 MB, not instructions.)
    const int cq = kq & 1, hf = lane >> 5;
    constexpr float LOG2E = 1.44269504088896341f;
    auto epilogue = [&]() {
        f32x4 erv[4];
        unsigned eovo[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int rb = o >> 1, rs = o & 1, c0 = g * 32 + rh * 16 + rb * 8 + 4 * cq;
            const int oy = by * 8 + 2 * pq + rs, ox = bx * 32 + 16 * hf + nn;
            const bool in = (oy < a.outH) & (ox < a.outW);
            const int pix = oy * a.outW + ox;
            const unsigned rvo = in ? (unsigned)((pix * a.Cout + c0) * 4) : OOR;
            eovo[o] = in ? (unsigned)((pix * a.out_cstride + c0) * 4) : OOR;
            erv[o] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.residual) erv[o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvo, 0, 0));
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int rb = o >> 1, rs = o & 1, cl = rh * 16 + rb * 8 + 4 * cq;
            const f32x4 bf = *reinterpret_cast<const f32x4 *>(&epar[0][cl]), bml = *reinterpret_cast<const f32x4 *>(&epar[1][cl]);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(&epar[2][cl]), sh = *reinterpret_cast<const f32x4 *>(&epar[3][cl]);
            const f32x4 isf = *reinterpret_cast<const f32x4 *>(&epar[4][cl]), ism = *reinterpret_cast<const f32x4 *>(&epar[5][cl]);
            // lanes 0..31 hold conv_f, lanes 32..63 conv_m: after the exchange the lower half-wave owns pixel block (rs, 0), the upper
            // half block (rs, 1), f in one register and m in the other
            u32x4 u0 = __builtin_bit_cast(u32x4, acc[rb][2 * rs]), u1 = __builtin_bit_cast(u32x4, acc[rb][2 * rs + 1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                u0[k] = sw[0];
                u1[k] = sw[1];
            }
            f32x4 f = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, u0), isf, bf);
            const f32x4 mm = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, u1), ism, bml);
            if (a.elu) {
                const f32x4 fe = f * LOG2E;
                f32x4 e;
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f(fe[k]);
                e = e + f32x4{-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : e[k];
            }
            f32x4 sg, t;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_exp2f(mm[k]);
            t = t + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(t[k]);
            const f32x4 v = (f * sg) * sc + sh + erv[o];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, eovo[o], 0, 0);
        }
    };

    auto stage = [&](int chunk) {
        const int nchunk = chunk + 1 == n ? 0 : chunk + 1;      // wraps into the next unit (same weights)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int m = tap * 4 + pb, cur = m & 1;
                // ---- loads for later: B operands of the next block (the first block of the next stage waits for the barrier)
                if (m + 1 < 36 && !(ABL & 4)) bload(cur ^ 1, x_cur, (m + 1) >> 2, (m + 1) & 3);
                if (pb == 0 && !(ABL & 2)) {                                            // weights two taps ahead
                    if (tap + 2 < 9) wload((tap + 2) % 3, chunk, tap + 2);
                    else wload((tap + 2) % 3, nchunk, tap + 2 - 9);
                }
                // the next chunk's patch: registers -> pieces -> the other buffer, one float4 per block from tap 2 on
                if (m >= 8 && m - 8 < DG::NI && !(ABL & 1)) lwrite1(m - 8, x_nxt);
                if (pb == 3 && tap < 8 && !(ABL & 32)) ascale(tap + 1);                 // its wh arrived a tap ago (slot (tap + 1) % 3)
                const f16x8 bh = __builtin_bit_cast(f16x8, Bh[cur]), bl = __builtin_bit_cast(f16x8, Bl[cur]);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[rb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(As[tap & 1][rb], bl, acc[rb][pb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Wl[tap % 3][rb]), bh, acc[rb][pb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Wh[tap % 3][rb]), bh, acc[rb][pb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(ABL & 32)) ascale(0);                             // tap 0 of the next stage (nine taps: the parity ring restarts)
        advance();
        if (!(ABL & 1)) gload();                                // the patch after next: a whole stage to land
        if (!(ABL & 16)) __syncthreads();                       // X[x_nxt] complete, X[x_cur] free
        const int t_ = x_cur;
        x_cur = x_nxt;
        x_nxt = t_;
        if (!(ABL & 4)) bload(0, x_cur, 0, 0);                  // first block of the next stage (the last stage of all reads a valid buffer)
    };

    for (int u = blockIdx.x; u < a.n_units; u += G) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < n; ++chunk) stage(chunk);
        if (ABL & 8) {
            f32x4 sum = acc[0][0];
#pragma unroll
            for (int i = 1; i < 8; ++i) sum += acc[i >> 2][i & 3];
            if (sum[0] == 12345.678f) a.out[tid] = sum[1] + sum[2] + sum[3];
        } else
            epilogue();
        step_tile(by, bx);
    }
}

// ------------------------------------------------------------------------------------------
// The direct split-operand kernel at STRIDE 2 (the encoder's three down-sampling layers, feat_extract.1 / .2 / .6: 291 us of the plan on
// the fp32 direct kernels).  Same operands, weight blob, waves and stage as gated_conv_d3h_kernel; what differs is the geometry:
//   * unit = 8 x 16 output pixels x 64 output channels (a PAIR of 32-channel groups): wave (rq = w & 3, ph = w >> 2) owns row quarter rq
//     of the pair (32 MFMA rows) x output rows 4 ph .. 4 ph + 3 (four blocks of 16 pixels = one image row each);
//   * the 17 x 33 input patch of a unit is stored as its four PARITY PLANES X[plane (y & 1, x & 1)][9][17][slot][8 halfs] (78 KB per
//     buffer): tap (dy, dx) of output (r, c) reads plane (dy & 1, dx & 1) at (r + (dy >> 1), c + (dx >> 1)) — consecutive output pixels
//     are consecutive plane pixels, so a B fragment is the same conflict-free ds_read_b128 as at stride 1.
template <int KS>
struct D3hS2Geom {
    static constexpr int NT = KS * KS, RING = NT % 3 == 0 ? 3 : 4;   // taps; weight ring (NT % RING == 0: static slots across stages)
    static constexpr int IH = 14 + KS, IW = 30 + KS, NPIX = IH * IW;   // input patch of a unit: 17 x 33 (3 x 3), 18 x 34 (4 x 4)
    static constexpr int PH = 9, PW = 17;                      // a parity plane (the odd planes use 8 rows / 16 columns of it)
    static constexpr int XBUF = 4 * PH * PW * 32;              // dwords per buffer (78,336 bytes)
    static constexpr int NE = NPIX * 8, NI = (NE + 511) / 512; // 4488 / 4896 float4 of a patch chunk: 9 / 10 per thread
};

// ABL: as gated_conv_d3h_kernel
// KS = 4: the decoder's 4 x 4 / stride-2 layers (pad 1): sixteen taps, the same four planes (row / column offsets 0, 1)
template <int KS = 3, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gated_conv_d3h_s2_kernel(const ConvKArgs a)
{
    using DG = D3hS2Geom<KS>;
    constexpr int NT = DG::NT, RING = DG::RING;
    constexpr bool MUL = false;
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * DG::XBUF];
    __shared__ __attribute__((aligned(16))) float epar[6][64];  // the group PAIR's epilogue parameters: b_f, -log2e b_m, BN scale, BN shift, 1 / s_f, -log2e / s_m
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), rq = wv & 3, rh = rq & 1, ph = wv >> 2;    // row quarter of the 64-channel pair, pixel half
    const SrcDev s = a.src[0];
    const int groups = (a.CoutPad + 63) >> 6, G = gridDim.x;       // group PAIRS (64 output channels per unit; the last pair may be half)
    const int gp = blockIdx.x % groups, g = gp * 2 + (rq >> 1);
    const int n = a.nchunks;                                   // 32-channel chunks
    constexpr unsigned OOR = 0x80000000u;

    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // staging cursor
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    // ---- staging: thread = float4 e = tid + 512 i of the patch chunk (pixel e >> 3, channel quad e & 7)
    int soff[DG::NI];                                          // LDS dword offset of the hi piece (+ buffer); lo = the same ^ 16 dwords
    unsigned rel[DG::NI], aoff[DG::NI];
#pragma unroll
    for (int i = 0; i < DG::NI; ++i) {
        const int e = tid + i * 512, pix = e >> 3, q4 = e & 7, iy = pix / DG::IW, ix = pix % DG::IW;
        const int plane = (iy & 1) * 2 + (ix & 1), prow = iy >> 1, pcol = ix >> 1;         // the four parity planes of the patch
        soff[i] = ((plane * DG::PH + prow) * DG::PW + pcol) * 32 + ((((q4 >> 1)) ^ (pcol & 7)) << 2) + ((q4 & 1) << 1);
        rel[i] = (unsigned)((iy * s.W + ix) * s.C + 4 * q4) * 4u;
    }
    const unsigned src_bytes = (unsigned)(a.inH * s.W * s.C) * 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, src_bytes, 0x00020000);
    const auto rsrc_mul = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MUL ? a.mul : s.p), 0, src_bytes, 0x00020000);
    auto set_patch = [&]() {
        const int y0 = pby * 16 - 1, x0 = pbx * 32 - 1;                         // input origin of the 8 x 16 output unit at stride 2
        const unsigned base = (unsigned)((y0 * s.W + x0) * s.C) * 4u;          // may wrap: only pixels inside the image use it
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) {
            const int e = tid + i * 512, pix = e >> 3, ppy = pix / DG::IW, ppx = pix % DG::IW;
            const bool ok = (e < DG::NE) & (ppy >= -y0) & (ppy < a.inH - y0) & (ppx >= -x0) & (ppx < a.inW - x0);
            aoff[i] = ok ? base + rel[i] : OOR;
        }
    };
    auto advance = [&]() {
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    float4 st[DG::NI], stm[MUL ? DG::NI : 1];
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) {
            st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, aoff[i], pchunk * 128, 0));
            if constexpr (MUL) stm[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_mul, aoff[i], pchunk * 128, 0));
        }
    };
    auto split2 = [](float x, float y, unsigned &hi, unsigned &lo) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x));                    // x - f32(hi), exact
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(y));
        const f32x2 rs = f32x2{r0, r1} * f32x2{2048.0f, 2048.0f};
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(rs.x), "v"(rs.y));
    };
    auto lwrite = [&](int xb) {                                 // registers -> f16 pieces -> X[xb]
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) {
            if ((DG::NI - 1) * 512 + 511 >= DG::NE && i == DG::NI - 1 && tid + i * 512 >= DG::NE) continue;      // the last round is partial
            float4 v = st[i];
            if constexpr (MUL) v = make_float4(v.x * stm[i].x, v.y * stm[i].y, v.z * stm[i].z, v.w * stm[i].w);
            unsigned h0, l0, h1, l1;
            split2(v.x, v.y, h0, l0);
            split2(v.z, v.w, h1, l1);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + soff[i], 8)) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + (soff[i] ^ 16), 8)) = u32x2{l0, l1};
        }
    };

    // ---- A operand (weights): [group][rh][chunk][tap][row block 2][piece 2][lane][8 halfs]; ring of three taps, two ahead
    const char *const wbase = reinterpret_cast<const char *>(a.wp_d3h) + ((size_t)(g * 2 + rh) * n) * (NT * 4096);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, g < (a.CoutPad >> 5) ? (unsigned)n * (NT * 4096) : 0u, 0x00020000);   // a missing group: every load out of range
    const unsigned wvoff = lane * 16;
    u32x4 Wh[RING][2], Wl[RING][2];
    auto wload = [&](int slot, int chunk, int tap) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            Wh[slot][rb] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, ((chunk * NT + tap) * 4 + rb * 2) * 1024, 0);
            Wl[slot][rb] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, ((chunk * NT + tap) * 4 + rb * 2 + 1) * 1024, 0);
        }
    };
    // ---- B operand: lane (pixel nn of the 16-pixel block, channel octet kq); per tap column dx the swizzled slot differs
    const int nn = lane & 15, kq = lane >> 4;
    int bho[KS], blo[KS];                                         // dword offsets inside a pixel row segment, hi / lo piece
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
        bho[dx] = nn * 32 + ((kq ^ ((nn + (dx >> 1)) & 7)) << 2);
        blo[dx] = nn * 32 + (((4 + kq) ^ ((nn + (dx >> 1)) & 7)) << 2);
    }

    f32x4 acc[2][4];
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(a.outH * a.outW * a.out_cstride) * 4u, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.residual ? a.residual : a.out), 0,
                                                            (unsigned)(a.outH * a.outW * a.Cout) * 4u, 0x00020000);
    const float *const wsc = reinterpret_cast<const float *>(a.wp_d3h) + (size_t)n * (64 * NT) * a.CoutPad;       // 1 / s: [f | m][CoutPad]

    // ---- prologue: patch(0) -> X[0]; patch(1) on its way; the first two taps' weights; the epilogue's parameters into LDS (a load
    // from global memory inside a stage would queue behind, and wait for, the weight stream)
    if (tid < 384) {
        constexpr float L2E = 1.44269504088896341f;
        const int arr = tid >> 6, c = gp * 64 + (tid & 63);
        const float v = c >= a.CoutPad ? 0.0f : arr < 4 ? a.params[arr * a.CoutPad + c] : wsc[(arr - 4) * a.CoutPad + c];
        epar[arr][tid & 63] = (arr == 1 || arr == 5) ? v * -L2E : v;
    }
    set_patch();
    gload();
    wload(0, 0, 0);
    wload(1, 0, 1);
    lwrite(0);
    advance();
    gload();
    __syncthreads();
    int x_cur = 0, x_nxt = DG::XBUF;

    // B operands of (tap, pixel block): ring of two, fetched one block ahead; A operands: weights two taps ahead (Wh / Wl ring of
    // three), 2^-11 wh of the next tap formed during this one.  hipcc sinks a load to its first use unless the order is pinned:
    // every group of six MFMAs ends in a sched_barrier, the loads sit between the groups.
    u32x4 Bh[2], Bl[2];
    auto bload = [&](int slot, int xb, int tap, int pb) {
        const int dy = tap / KS, dx = tap % KS;                 // output row 4 ph + pb reads plane (dy & 1, dx & 1) at row + (dy >> 1), column + (dx >> 1)
        const int base = xb + ((((dy & 1) * 2 + (dx & 1)) * DG::PH + 4 * ph + pb + (dy >> 1)) * DG::PW + (dx >> 1)) * 32;
        Bh[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + base + bho[dx], 16));
        Bl[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + base + blo[dx], 16));
    };
    f16x8 As[2][2];                                            // [tap & 1][row block]
    auto ascale = [&](int tap) {
        const _Float16 k11 = (_Float16)0x1p-11f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) As[tap & 1][rb] = __builtin_bit_cast(f16x8, Wh[tap % RING][rb]) * f16x8{k11, k11, k11, k11, k11, k11, k11, k11};
    };
    auto lwrite1 = [&](int i, int xb) {
        if ((DG::NI - 1) * 512 + 511 >= DG::NE && i == DG::NI - 1 && tid + i * 512 >= DG::NE) return;
        float4 v = st[i];
        if constexpr (MUL) v = make_float4(v.x * stm[i].x, v.y * stm[i].y, v.z * stm[i].z, v.w * stm[i].w);
        unsigned h0, l0, h1, l1;
        split2(v.x, v.y, h0, l0);
        split2(v.z, v.w, h1, l1);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + soff[i], 8)) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(__builtin_assume_aligned(lds + xb + (soff[i] ^ 16), 8)) = u32x2{l0, l1};
    };
    ascale(0);
    bload(0, x_cur, 0, 0);

    // ---- epilogue of a unit: acc[rb][pb], lane (nn, kq), register r = MFMA row 4 kq + r: kq = 0, 1 -> conv_f of channels 4 kq + r of the
    // row block, kq = 2, 3 -> conv_m of channels 4 (kq - 2) + r; column nn = pixel nn of unit row 4 ph + pb.
    const int cq = kq & 1, hf = lane >> 5;
    constexpr float LOG2E = 1.44269504088896341f;
    auto epilogue = [&]() {
        f32x4 erv[4];
        unsigned eovo[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int rb = o >> 1, rs = o & 1, c0 = gp * 64 + rq * 16 + rb * 8 + 4 * cq;
            const int oy = by * 8 + 4 * ph + 2 * rs + hf, ox = bx * 16 + nn;          // blocks (2 rs, 2 rs + 1) = two image rows
            const bool in = (oy < a.outH) & (ox < a.outW);
            const int pix = oy * a.outW + ox;
            const unsigned rvo = in ? (unsigned)((pix * a.Cout + c0) * 4) : OOR;
            eovo[o] = in ? (unsigned)((pix * a.out_cstride + c0) * 4) : OOR;
            erv[o] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.residual) erv[o] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvo, 0, 0));
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int rb = o >> 1, rs = o & 1, cl = rq * 16 + rb * 8 + 4 * cq;
            const f32x4 bf = *reinterpret_cast<const f32x4 *>(&epar[0][cl]), bml = *reinterpret_cast<const f32x4 *>(&epar[1][cl]);
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(&epar[2][cl]), sh = *reinterpret_cast<const f32x4 *>(&epar[3][cl]);
            const f32x4 isf = *reinterpret_cast<const f32x4 *>(&epar[4][cl]), ism = *reinterpret_cast<const f32x4 *>(&epar[5][cl]);
            // lanes 0..31 hold conv_f, lanes 32..63 conv_m: after the exchange the lower half-wave owns image row 2 rs of the wave's four, the
            // upper half row 2 rs + 1, f in one register and m in the other
            u32x4 u0 = __builtin_bit_cast(u32x4, acc[rb][2 * rs]), u1 = __builtin_bit_cast(u32x4, acc[rb][2 * rs + 1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                u0[k] = sw[0];
                u1[k] = sw[1];
            }
            f32x4 f = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, u0), isf, bf);
            const f32x4 mm = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, u1), ism, bml);
            if (a.elu) {
                const f32x4 fe = f * LOG2E;
                f32x4 e;
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f(fe[k]);
                e = e + f32x4{-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : e[k];
            }
            f32x4 sg, t;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_exp2f(mm[k]);
            t = t + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(t[k]);
            const f32x4 v = (f * sg) * sc + sh + erv[o];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, eovo[o], 0, 0);
        }
    };

    auto stage = [&](int chunk) {
        const int nchunk = chunk + 1 == n ? 0 : chunk + 1;      // wraps into the next unit (same weights)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int m = tap * 4 + pb, cur = m & 1;
                // ---- loads for later: B operands of the next block (the first block of the next stage waits for the barrier)
                if (m + 1 < 4 * NT && !(ABL & 4)) bload(cur ^ 1, x_cur, (m + 1) >> 2, (m + 1) & 3);
                if (pb == 0 && !(ABL & 2)) {                                            // weights two taps ahead
                    if (tap + 2 < NT) wload((tap + 2) % RING, chunk, tap + 2);
                    else wload((tap + 2) % RING, nchunk, tap + 2 - NT);
                }
                // the next chunk's patch: registers -> pieces -> the other buffer, one float4 per block from tap 2 on
                if (m >= 8 && m - 8 < DG::NI && !(ABL & 1)) lwrite1(m - 8, x_nxt);
                if (pb == 3 && tap < NT - 1 && !(ABL & 32)) ascale(tap + 1);                 // its wh arrived a tap ago (slot (tap + 1) % 3)
                const f16x8 bh = __builtin_bit_cast(f16x8, Bh[cur]), bl = __builtin_bit_cast(f16x8, Bl[cur]);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[rb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(As[tap & 1][rb], bl, acc[rb][pb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Wl[tap % RING][rb]), bh, acc[rb][pb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Wh[tap % RING][rb]), bh, acc[rb][pb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(ABL & 32)) ascale(0);                             // tap 0 of the next stage (nine taps: the parity ring restarts)
        advance();
        if (!(ABL & 1)) gload();                                // the patch after next: a whole stage to land
        if (!(ABL & 16)) __syncthreads();                       // X[x_nxt] complete, X[x_cur] free
        const int t_ = x_cur;
        x_cur = x_nxt;
        x_nxt = t_;
        if (!(ABL & 4)) bload(0, x_cur, 0, 0);                  // first block of the next stage (the last stage of all reads a valid buffer)
    };

    // a layer with an odd number of 32-channel groups (feat_extract.4: 64 -> 32): the waves of the pair's missing half stage patches and
    // keep the barriers, nothing else
    const bool live = g < (a.CoutPad >> 5);
    auto stage_idle = [&]() {
#pragma unroll
        for (int i = 0; i < DG::NI; ++i) lwrite1(i, x_nxt);
        advance();
        gload();
        __syncthreads();
        const int t_ = x_cur;
        x_cur = x_nxt;
        x_nxt = t_;
    };
    for (int u = blockIdx.x; u < a.n_units; u += G) {
        if (!live) {
            for (int chunk = 0; chunk < n; ++chunk) stage_idle();
            step_tile(by, bx);
            continue;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < n; ++chunk) stage(chunk);
        if (ABL & 8) {
            f32x4 sum = acc[0][0];
#pragma unroll
            for (int i = 1; i < 8; ++i) sum += acc[i >> 2][i & 3];
            if (sum[0] == 12345.678f) a.out[tid] = sum[1] + sum[2] + sum[3];
        } else
            epilogue();
        step_tile(by, bx);
    }
}


// ------------------------------------------------------------------------------------------
// NEGATIVE RESULT, kept in the DEBUG library only (-DREAD_DEBUG_KNOBS, read_tuning_set("conv_w4h_waves", 8)): the split-operand
// kernel above cut into SPECIALISED waves — eight per workgroup, two per SIMD: waves 0..3 multiply (MFMA stream, weight ring,
// B operands, epilogue), waves 4..7 produce (patch loads, input transform, split, V stores).  Same arithmetic, operands, LDS layout
// and unit walk; results bit-identical to the four-wave kernel (tests/test_gpu_conv.py runs both when the debug library is loaded).
// Why it was tried (profiles/r6_w4h_ablation.md): in the four-wave kernel the MFMAs alone take 9 us of a 44 us launch at C = 256 and
// everything else ADDS to them — a wave alone on its SIMD issues in order, its loads return in order (a weight fragment requested
// behind a patch load waits for that load's miss), the transform's 350 vector instructions sit between its MFMAs.  Two waves with
// different jobs on a SIMD overlap by construction and have separate vmcnt.
// MEASURED (round 6, tools/w4h_ab.py): 65.6 / 70.3 / 54.7 / 48.5 us per launch at C = 32 / 64 / 128 / 256 against 59.8 / 52.9 / 47.4 / 44.2
// for the four-wave kernel — SLOWER at every level.  The probes say why: the multiplying waves ALONE (producers switched off) take
// 35.5 us at C = 256.  At two waves per SIMD a wave has 256 registers; 144 are accumulators, which leaves a six-frequency weight
// ring fetched four ahead: 8 KiB in flight per wave, 32 KiB per CU, and an L2 hit under this load takes ~0.5 us — the weight stream
// (288 KiB per 32 input channels and unit) then runs at 65 GB/s per CU = 4.4 us per stage, where the four-wave kernel (nine-frequency
// ring, six ahead, 48 KiB in flight per CU) is not weight-bound.  Specialisation trades the issue serialisation for a shallower
// prefetch, and on this machine the prefetch depth is worth more.  It is therefore not part of libreadhip.so.
// ------------------------------------------------------------------------------------------
#ifdef READ_DEBUG_KNOBS
template <int ABL = 0>
__global__ __launch_bounds__(512, 1) void gated_conv_wino4h2_kernel(const ConvKArgs a)
{
    using WG = Wino4hGeom;
    __shared__ __attribute__((aligned(16))) unsigned lds[WG::LDS_DWORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;                                   // 32-channel chunks
    constexpr unsigned OOR = 0x80000000u;
    constexpr int BAR_M = 102;                                 // multiplying waves: the stage's barrier in front of frequency pair 17

    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };

    if (wv >= 4) {
        // =============================== producing waves: patch -> registers -> B^T d B -> f16 pieces -> V ===============================
        const int tw = wv - 4;
        const int cp = lane & 15, tl = tw * 4 + (lane >> 4);
        int pby = (blockIdx.x / groups) / a.tiles_x, pbx = (blockIdx.x / groups) % a.tiles_x, pu = blockIdx.x, pchunk = 0;
        unsigned xoff[6];
        int rowoff[6];
        float rmask[6], rmask_d[6];
        bool rclamp = false, rclamp_d = false;
        const unsigned src_bytes = (unsigned)(a.inH * s.W * s.C) * 4u;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, src_bytes, 0x00020000);
        auto set_patch = [&]() {
            const int y0 = pby * 8 + 4 * (tw >> 1) - 1, x0 = pbx * 32 + 4 * (tl & 7) - 1;
            rclamp = false;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int yy = y0 + r;
                const bool ok = (unsigned)yy < (unsigned)a.inH;
                const int yc = yy < 0 ? 0 : yy >= a.inH ? a.inH - 1 : yy;
                rowoff[r] = yc * s.W * s.C * 4;
                rmask[r] = ok ? 1.0f : 0.0f;
                rclamp = rclamp || !ok;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) xoff[c] = (unsigned)(x0 + c) < (unsigned)a.inW ? (unsigned)((x0 + c) * s.C + 2 * cp) * 4u : OOR;
        };
        auto advance = [&]() {
#pragma unroll
            for (int r = 0; r < 6; ++r) rmask_d[r] = rmask[r];
            rclamp_d = rclamp;
            if (++pchunk == n) {
                pchunk = 0;
                if (pu + G < a.n_units) {
                    pu += G;
                    step_tile(pby, pbx);
                }
                set_patch();
            }
        };
        f32x2 d2[6][6];
        auto gload_all = [&]() {
            if (ABL & 8) return;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    d2[r][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, xoff[c], rowoff[r] + pchunk * 128, 0));
                }
        };
        auto bt6 = [](f32x2 &x0, f32x2 &x1, f32x2 &x2, f32x2 &x3, f32x2 &x4, f32x2 &x5) {
            const f32x2 p = pk_add(x3, x4), q = pk_add(x1, x2), r = pk_sub(x4, x3), u = pk_sub(x1, x2), f = pk_sub(x3, x1), h = pk_sub(x4, x2);
            const f32x2 y0 = __builtin_elementwise_fma(x2, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x0, f32x2{4.0f, 4.0f}, x4));
            const f32x2 y5 = __builtin_elementwise_fma(x3, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x1, f32x2{4.0f, 4.0f}, x5));
            x0 = y0;
            x1 = __builtin_elementwise_fma(q, f32x2{-4.0f, -4.0f}, p);
            x2 = __builtin_elementwise_fma(u, f32x2{4.0f, 4.0f}, r);
            x3 = __builtin_elementwise_fma(f, f32x2{2.0f, 2.0f}, h);
            x4 = __builtin_elementwise_fma(f, f32x2{-2.0f, -2.0f}, h);
            x5 = y5;
        };
        const int vwoff = tl * 16 + (((cp >> 2) ^ ((-tw) & 3)) << 2) + (cp & 3);
        auto split_store = [&](const f32x2 x, int vb, int fq) {
            unsigned hi, lo;
            float r0, r1;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x.x), "v"(x.y));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x.x));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x.y));
            const f32x2 rs = f32x2{r0, r1} * f32x2{2048.0f, 2048.0f};
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(rs.x), "v"(rs.y));
            lds[vwoff + vb + fq * WG::VFREQ] = hi;
            lds[vwoff + vb + fq * WG::VFREQ + 256] = lo;
        };
        auto transform = [&](int vb) {
            if (ABL & 1) return;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (rclamp_d) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) d2[r][c] = d2[r][c] * f32x2{rmask_d[r], rmask_d[r]};
                }
                bt6(d2[0][c], d2[1][c], d2[2][c], d2[3][c], d2[4][c], d2[5][c]);
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                bt6(d2[r][0], d2[r][1], d2[r][2], d2[r][3], d2[r][4], d2[r][5]);
#pragma unroll
                for (int c = 0; c < 6; ++c) split_store(d2[r][c], vb, r * 6 + c);
            }
        };
        // prologue: patch(0) -> V(0); patch(1) on its way
        set_patch();
        gload_all();
        advance();
        transform(0);
        gload_all();
        advance();
        __syncthreads();                                       // P: V(0) complete
        int v_nxt = WG::VBUF;
        for (int u = blockIdx.x; u < a.n_units; u += G)
            for (int chunk = 0; chunk < n; ++chunk) {
                transform(v_nxt);                              // chunk + 1 (of this or of the next unit) -> the other buffer
                gload_all();                                   // chunk + 2: a whole stage to land
                advance();
                v_nxt = WG::VBUF - v_nxt;
                if (!(ABL & 256)) __syncthreads();             // S: V(chunk + 1) complete; V(chunk) free
            }
        return;
    }

    // =============================== multiplying waves ===============================
    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;
    const char *const wbase = reinterpret_cast<const char *>(a.wp_w4h) + ((size_t)(g * 4 + wv) * n) * (36 * 2048);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, (unsigned)n * (36 * 2048), 0x00020000);
    const unsigned wvoff = lane * 16;
    constexpr int RW = 6, WLEAD = 4;                           // ring of six frequencies, fetched four ahead
    u32x4 Wh[RW], Wl[RW];
    f16x8 Ws[2];                                               // 2^-11 Uh of the next frequency pair
    auto wload = [&](int slot, int chunk, int fq) {
        Wh[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (chunk * 36 + fq) * 2048, 0);
        Wl[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (chunk * 36 + fq) * 2048 + 1024, 0);
    };
    auto wscale = [&](int fq) {
        const _Float16 k = (_Float16)0x1p-11f;
        Ws[fq % 2] = __builtin_bit_cast(f16x8, Wh[fq % RW]) * f16x8{k, k, k, k, k, k, k, k};
    };
    const int t16 = lane & 15, kl = lane >> 4;
    const int vrd = t16 * 16 + ((kl ^ ((-(t16 >> 2)) & 3)) << 2);
    constexpr int RB = 4;
    u32x4 Bh[RB], Bl[RB];
    auto bload = [&](int slot, int vb, int fq) {
        Bh[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + vb + fq * WG::VFREQ + vrd, 16));
        Bl[slot] = *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(lds + vb + fq * WG::VFREQ + 256 + vrd, 16));
    };
    f32x4 acc[36];
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(a.outH * a.outW * a.out_cstride) * 4u, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.residual ? a.residual : a.out), 0,
                                                            (unsigned)(a.outH * a.outW * a.Cout) * 4u, 0x00020000);
    const float *const wsc = reinterpret_cast<const float *>(a.wp_w4h) + (size_t)n * 2304 * a.CoutPad;
    int v_cur = 0, v_nxt = WG::VBUF;

    // One stage = one 32-channel chunk = 108 MFMAs.  LAST: the unit's last chunk — nothing of the next unit is prefetched (its
    // rings would have to live through the epilogue: no registers for that at two waves per SIMD)
    auto stage_body = [&](auto first_tag, auto last_tag, int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 18; ++pr)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int fq = 2 * pr + (j & 1), pc = j >> 1, m = pr * 6 + j;
                if (m == BAR_M && !(ABL & 256)) __syncthreads();
                const f16x8 av = pc == 0 ? Ws[fq % 2] : __builtin_bit_cast(f16x8, pc == 1 ? Wl[fq % RW] : Wh[fq % RW]);
                const f16x8 bv = __builtin_bit_cast(f16x8, pc == 0 ? Bl[fq % RB] : Bh[fq % RB]);
                if (ABL & 1024) {
                    if (FIRST && pc == 0) acc[fq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (pc == 2) asm volatile("" :: "v"(av), "v"(bv));
                } else if (FIRST && pc == 0) {
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    acc[fq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, zero, 0, 0, 0);
                } else
                    acc[fq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[fq], 0, 0, 0);
                if (j < 2) {                                                         // B operands one frequency pair ahead
                    const int bf = 2 * pr + 2 + j;
                    if (bf < 36) bload(bf % RB, v_cur, bf);
                    else if (!LAST) bload(bf % RB, v_nxt, bf - 36);
                } else if (j < 4) {                                                  // weights WLEAD frequencies ahead
                    const int wf = 2 * pr + WLEAD + (j - 2);
                    if (wf < 36) wload(wf % RW, chunk, wf);
                    else if (!LAST) wload(wf % RW, chunk + 1, wf - 36);
                } else {                                                             // 2^-11 Uh of the next frequency pair
                    const int sf = 2 * pr + 2 + (j - 4);
                    if (sf < 36 || !LAST) wscale(sf % 36);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        const int v = v_cur;
        v_cur = v_nxt;
        v_nxt = v;
    };

    __syncthreads();                                           // P
    for (int u = blockIdx.x; u < a.n_units; u += G) {
        // prime the rings (first unit: behind the prologue barrier; later units: behind the epilogue)
#pragma unroll
        for (int j = 0; j < WLEAD; ++j) wload(j, 0, j);
        bload(0, v_cur, 0);
        bload(1, v_cur, 1);
        wscale(0);
        wscale(1);
        if (n == 1) stage_body(std::true_type{}, std::true_type{}, 0);
        else {
            stage_body(std::true_type{}, std::false_type{}, 0);
            for (int chunk = 1; chunk < n - 1; ++chunk) stage_body(std::false_type{}, std::false_type{}, chunk);
            stage_body(std::false_type{}, std::true_type{}, n - 1);
        }

        // ================= unit epilogue: two tile rows (p, p + 2) at a time =================
        __builtin_amdgcn_s_setprio(1);
        const int cq = (lane >> 4) & 1, hf = lane >> 5;
        const int c0 = g * 32 + wv * 8 + 4 * cq;
        const int oy = by * 8 + 4 * (t16 >> 3) + 2 * hf, ox = bx * 32 + 4 * (t16 & 7);
        if (ABL & 128) {
            f32x4 sum = acc[0];
#pragma unroll
            for (int i = 1; i < 36; ++i) sum += acc[i];
            if (oy < a.outH && ox < a.outW) *reinterpret_cast<f32x4 *>(a.out + ((size_t)oy * a.outW + ox) * a.out_cstride + c0) = sum;
            step_tile(by, bx);
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        const f32x4 bf = *reinterpret_cast<const f32x4 *>(a.params + c0);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.params + 2 * a.CoutPad + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.params + 3 * a.CoutPad + c0);
        const f32x4 isf = *reinterpret_cast<const f32x4 *>(wsc + c0);
        constexpr float LOG2E = 1.44269504088896341f;
        const f32x4 ism = *reinterpret_cast<const f32x4 *>(wsc + a.CoutPad + c0) * -LOG2E;
        const f32x4 bml = *reinterpret_cast<const f32x4 *>(a.params + a.CoutPad + c0) * -LOG2E;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            unsigned rvoff[4], ovoff[4];
            f32x4 rv[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const bool in = (oy + py < a.outH) & (ox + px < a.outW) & (c0 < a.Cout);
                const int pix = (oy + py) * a.outW + ox + px;
                rvoff[px] = in ? (unsigned)((pix * a.Cout + c0) * 4) : OOR;
                ovoff[px] = in ? (unsigned)((pix * a.out_cstride + c0) * 4) : OOR;
                rv[px] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.residual) rv[px] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvoff[px], 0, 0));
            }
            // rows p = py and p + 2 of A^T M:  p = 0: a0 + s1 + s2 | p = 2: s1 + 4 s2   (s = sums of frequency rows 1,2 / 3,4)
            //                                  p = 1: d1 + 2 d2    | p = 3: d1 + 8 d2 + a5  (d = differences)
            f32x4 Ra[6], Rb[6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                if (py == 0) {
                    const f32x4 s1 = acc[6 + nu] + acc[12 + nu], s2 = acc[18 + nu] + acc[24 + nu];
                    Ra[nu] = acc[nu] + s1 + s2;
                    Rb[nu] = s1 + 4.0f * s2;
                } else {
                    const f32x4 d1 = pk_sub4(acc[6 + nu], acc[12 + nu]), dd2 = pk_sub4(acc[18 + nu], acc[24 + nu]);
                    Ra[nu] = d1 + 2.0f * dd2;
                    Rb[nu] = d1 + 8.0f * dd2 + acc[30 + nu];
                }
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                // column px of (A^T M) A for both rows
                f32x4 ya, yb;
                if (px == 0) {
                    ya = Ra[0] + (Ra[1] + Ra[2]) + (Ra[3] + Ra[4]);
                    yb = Rb[0] + (Rb[1] + Rb[2]) + (Rb[3] + Rb[4]);
                } else if (px == 1) {
                    ya = pk_sub4(Ra[1], Ra[2]) + 2.0f * pk_sub4(Ra[3], Ra[4]);
                    yb = pk_sub4(Rb[1], Rb[2]) + 2.0f * pk_sub4(Rb[3], Rb[4]);
                } else if (px == 2) {
                    ya = (Ra[1] + Ra[2]) + 4.0f * (Ra[3] + Ra[4]);
                    yb = (Rb[1] + Rb[2]) + 4.0f * (Rb[3] + Rb[4]);
                } else {
                    ya = pk_sub4(Ra[1], Ra[2]) + 8.0f * pk_sub4(Ra[3], Ra[4]) + Ra[5];
                    yb = pk_sub4(Rb[1], Rb[2]) + 8.0f * pk_sub4(Rb[3], Rb[4]) + Rb[5];
                }
                // lanes 0..31 hold conv_f, lanes 32..63 conv_m: after the exchange the lower half-wave owns tile row py, the upper
                // half row py + 2, f in one register and m in the other
                u32x4 u0 = __builtin_bit_cast(u32x4, ya), u1 = __builtin_bit_cast(u32x4, yb);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                    u0[k] = sw[0];
                    u1[k] = sw[1];
                }
                f32x4 f = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, u0), isf, bf);
                const f32x4 mm = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, u1), ism, bml);
                if (a.elu) {
                    const f32x4 fe = f * LOG2E;
                    f32x4 e;
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f(fe[k]);
                    e = e + f32x4{-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : e[k];
                }
                f32x4 sg, t;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_exp2f(mm[k]);
                t = t + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(t[k]);
                const f32x4 v = (f * sg) * sc + sh + rv[px];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, ovoff[px], 0, 0);
            }
        }
        step_tile(by, bx);
        __builtin_amdgcn_s_setprio(0);
    }
}

#endif  // READ_DEBUG_KNOBS

// ------------------------------------------------------------------------------------------
// NEGATIVE RESULT, kept in the DEBUG library only (-DREAD_DEBUG_KNOBS, read_tuning_set("conv_w4x2", 1)): the F(4x4,3x3) kernel
// above cut for TWO waves per SIMD.  Why it was tried (DESIGN.md 12.1 d): one wave per SIMD issues a vector instruction every
// 5.2 cycles, two waves one every 2.6 (tools/valu_probe.py), and a VALU instruction behind an fp32 MFMA costs 4 - 11.5 cycles with
// one wave per SIMD against 1.4 - 2.5 with two (tools/issue_probe.py).  The accumulators do not shrink with the tile (36
// frequencies x a 16 x 16 MFMA block), so the cut is over FREQUENCIES: eight waves per workgroup, wave (co = w & 3, fh = w >> 2) owns
// output channels 8 co .. 8 co + 7 and frequency rows 3 fh .. 3 fh + 2 of the 6 x 6 grid — 18 frequencies, 72 accumulators, the SAME
// weight blob and the same V buffer; input transform by halves (each half reads the whole 6 x 6 patch), output transform by halves
// with a hand-over of partial row sums between the wave pair through LDS (message counters, no workgroup barrier).
// MEASURED in round 5 (profiles/r5_w4x2_ab.json, tools/w4x2_ab.py; results equal to the kernel above within 7e-6, parity test
// green): 73.2 / 67.7 / 62.5 / 61.1 us per launch at C = 32 / 64 / 128 / 256 against 71.2 / 66.1 / 59.4 / 56.0 for the one-wave
// kernel, 214.6 against 220.8 frames/s in bench.py — SLOWER at every level.  The reason is structural, not a tuning state: the
// split halves the cost of a vector instruction and doubles their number (each wave still reads the whole 6 x 6 patch, runs its own
// operand rings and address bookkeeping: ~345 non-MFMA instructions per 72 MFMAs and wave, i.e. 690 per SIMD and stage against ~300),
// and the hand-over adds a dependent LDS round trip per unit.  It is therefore not part of libreadhip.so.
// ------------------------------------------------------------------------------------------
#ifdef READ_DEBUG_KNOBS
__global__ __launch_bounds__(512, 1) void gated_conv_wino4x2_kernel(const ConvKArgs a)
{
    using WG = Wino4Geom;
    constexpr int NI = (WG::NE + 511) / 512;                   // 1360 float4 of a raw chunk over 512 threads: 3 each
    constexpr int XCH = WG::LDS_FLOATS, XW = 4 * 64 * 4;       // exchange area: 4 float4 slots x 64 lanes per wave (4 KiB)
    __shared__ __attribute__((aligned(16))) float lds[WG::LDS_FLOATS + 8 * XW];
    __shared__ int xflag[16];                                  // [w]: messages written by wave w; [8 + w]: messages of its partner wave w has read
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), co = wv & 3, fh = wv >> 2;
    const SrcDev s = a.src[0];
    const int groups = a.CoutPad >> 5, G = gridDim.x;
    const int g = blockIdx.x % groups;
    const int n = a.nchunks;
    if (tid < 16) xflag[tid] = 0;

    int by = (blockIdx.x / groups) / a.tiles_x, bx = (blockIdx.x / groups) % a.tiles_x;      // running unit (8 x 32 pixel block)
    int pby = by, pbx = bx, pu = blockIdx.x, pchunk = 0;                                     // prefetch cursor (raw patches)
    auto step_tile = [&](int &ty_, int &tx_) {
        ty_ += a.wino_dby;
        tx_ += a.wino_dbx;
        if (tx_ >= a.tiles_x) {
            tx_ -= a.tiles_x;
            ++ty_;
        }
    };
    int loff[NI];
    unsigned rel[NI], aoff[NI];
    constexpr unsigned OOR = 0x80000000u;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 512, q = e % 4, pix = e / 4;
        loff[i] = e < WG::NE ? (pix / WG::IW) * WG::RS + (pix % WG::IW) * WG::PS + 4 * q : WG::KC;   // else: pixel 0's pad floats
        rel[i] = (unsigned)(((pix / WG::IW) * s.W + pix % WG::IW) * s.C + 4 * q) * 4u;
    }
    const unsigned src_bytes = (unsigned)(a.inH * s.W * s.C) * 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, src_bytes, 0x00020000);
    auto set_patch = [&]() {
        const int y0 = pby * 8 - 1, x0 = pbx * 32 - 1;
        const unsigned base = (unsigned)((y0 * s.W + x0) * s.C) * 4u;          // may wrap: only lanes inside the image use it
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + i * 512, pix = e / 4, ppy = pix / WG::IW, ppx = pix % WG::IW;
            const bool ok = (e < WG::NE) & (ppy >= -y0) & (ppy < a.inH - y0) & (ppx >= -x0) & (ppx < a.inW - x0);
            aoff[i] = ok ? base + rel[i] : OOR;
        }
    };
    auto advance = [&]() {
        if (++pchunk == n) {
            pchunk = 0;
            if (pu + G < a.n_units) {
                pu += G;
                step_tile(pby, pbx);
            }
            set_patch();
        }
    };
    float4 st[NI];
    auto gload1 = [&](int i) {
        st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, aoff[i], pchunk * (WG::KC * 4), 0));
    };
    auto lwrite1 = [&](int i, int obuf) {
        *reinterpret_cast<float4 *>(__builtin_assume_aligned(lds + obuf + loff[i], 16)) = st[i];
    };

    // ---- transform role: thread = (input channel c16 of the chunk, tile tl, frequency-row half = fh of its wave)
    const int c16 = tid & 15, tl = (tid >> 4) & 15;
    const int rbase = ((4 * (tl >> 3)) * WG::IW + 4 * (tl & 7)) * WG::PS + c16;
    const int vwoff = WG::V0 + tl * 16 + ((c16 >> 2) ^ ((tl >> 1) & 3)) * 4 + (c16 & 3);
    f32x2 d2[6][3];                                            // the patch as column pairs; after the column step rows 0..2 hold B^T d rows 3 fh ..
    // the half-specific arithmetic takes the half as a compile-time tag: the caller branches ONCE per stage (wave-uniform), so
    // that no branch sits between the MFMAs
    auto bt3v = [&](auto half, int c) {                        // rows 3 fh .. 3 fh + 2 of B^T applied down column pair c
        const f32x2 x0 = d2[0][c], x1 = d2[1][c], x2 = d2[2][c], x3 = d2[3][c], x4 = d2[4][c], x5 = d2[5][c];
        if constexpr (decltype(half)::value == 0) {
            const f32x2 p = pk_add(x3, x4), q = pk_add(x1, x2), r = pk_sub(x4, x3), u = pk_sub(x1, x2);
            d2[0][c] = __builtin_elementwise_fma(x2, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x0, f32x2{4.0f, 4.0f}, x4));
            d2[1][c] = __builtin_elementwise_fma(q, f32x2{-4.0f, -4.0f}, p);
            d2[2][c] = __builtin_elementwise_fma(u, f32x2{4.0f, 4.0f}, r);
        } else {
            const f32x2 f = pk_sub(x3, x1), h = pk_sub(x4, x2);
            d2[0][c] = __builtin_elementwise_fma(f, f32x2{2.0f, 2.0f}, h);
            d2[1][c] = __builtin_elementwise_fma(f, f32x2{-2.0f, -2.0f}, h);
            d2[2][c] = __builtin_elementwise_fma(x3, f32x2{-5.0f, -5.0f}, __builtin_elementwise_fma(x1, f32x2{4.0f, 4.0f}, x5));
        }
    };
    const f32x2 KA = {4.0f, -5.0f}, KB = {2.0f, 0.0f};
    auto bt6row = [&](f32x2 &P0, f32x2 &P1, f32x2 &P2) {        // as in the kernel above: (x0,x1)(x2,x3)(x4,x5) -> (y0,y5)(y1,y2)(y3,y4)
        f32x2 T, Y05, QU, PR, Y12, Y34, F2, H2;
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(T) : "v"(P0), "v"(KA), "v"(P2));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(Y05) : "v"(P1), "v"(KA), "v"(T));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(QU) : "v"(P0), "v"(P1));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(PR) : "v"(P2), "v"(P1));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(F2) : "v"(P1), "v"(P0));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(H2) : "v"(P2), "v"(P1));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(Y12) : "v"(QU), "v"(KA), "v"(PR));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(Y34) : "v"(F2), "v"(KB), "v"(H2));
        P0 = Y05;
        P1 = Y12;
        P2 = Y34;
    };
    // step k of the next chunk's transform: -36..-1 reads, 18..20 column pairs, 21..23 rows, 24..32 stores (two each)
    auto t_step = [&](auto half, const float *raw, int vb, int k) {
        if (k < 0) {
            const int e = k + 36, r = e / 6, c = e % 6;
            d2[r][c >> 1][c & 1] = raw[rbase + (r * WG::IW + c) * WG::PS];
        } else if (k < 21) {
            bt3v(half, k - 18);
        } else if (k < 24) {
            const int r = k - 21;
            bt6row(d2[r][0], d2[r][1], d2[r][2]);
        } else {
            const int r = (k - 24) / 3, j = (k - 24) % 3;             // pair j of local row r holds frequencies (0, 5), (1, 2), (3, 4) of row 3 fh + r
            const int f0 = (3 * fh + r) * 6 + (j == 0 ? 0 : j == 1 ? 1 : 3), f1 = (3 * fh + r) * 6 + (j == 0 ? 5 : j == 1 ? 2 : 4);
            lds[vwoff + vb + f0 * 256] = d2[r][j].x;
            lds[vwoff + vb + f1 * 256] = d2[r][j].y;
        }
    };
    constexpr int T_FIRST = -36, T_STEPS = 33;
    constexpr int BAR_M = 63;                                  // MFMA slot of the per-stage barrier (of 72)

    // ---- A operand: the same blob as the kernel above; this wave's half of its octet's fragments.  Ring of 6 (18 % 6 == 0: a
    // frequency keeps its slot across chunks), fetched 4 ahead (a ring of 9 / 7 ahead spilled: 256 registers per wave at 2 waves per SIMD)
    const float *const wbase = a.wp_w4 + ((size_t)(g * 4 + co) * n) * (36 * 256);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wbase), 0, (unsigned)n * (36 * 1024), 0x00020000);
    const unsigned wvoff = lane * 16;
    float4 Wq[6];
    constexpr int WLEAD = 4;
    auto wload1 = [&](int slot, int chunk, int fql) {
        Wq[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (chunk * 36 + 18 * fh + fql) * 1024, 0));
    };
    // ---- B operand: Vbuf[frequency][tile t16][slot]; ring of 6, fetched 4 ahead
    const int t16 = lane & 15, kl = lane >> 4;
    const int vlane = t16 * 16 + 4 * (kl ^ ((t16 >> 1) & 3));
    float4 Bq[6];
    auto bload1 = [&](int slot, int vb, int fql) {
        Bq[slot] = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(lds + WG::V0 + vb + (18 * fh + fql) * 256 + vlane, 16));
    };

    f32x4 acc[18];
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)(a.outH * a.outW * a.out_cstride) * 4u, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.residual ? a.residual : a.out), 0,
                                                            (unsigned)(a.outH * a.outW * a.Cout) * 4u, 0x00020000);

    // ---- prologue: raw(0), raw(1) -> LDS, raw(2) -> registers, this half's V(0), the first weight fragments and B operands
    set_patch();
#pragma unroll
    for (int i = 0; i < NI; ++i) gload1(i);
#pragma unroll
    for (int j = 0; j < WLEAD; ++j) wload1(j, 0, j);
#pragma unroll
    for (int i = 0; i < NI; ++i) lwrite1(i, 0);
    advance();
#pragma unroll
    for (int i = 0; i < NI; ++i) gload1(i);
#pragma unroll
    for (int i = 0; i < NI; ++i) lwrite1(i, WG::BUF);
    advance();
#pragma unroll
    for (int i = 0; i < NI; ++i) gload1(i);
    advance();
    __syncthreads();
    if (fh == 0) {
#pragma unroll
        for (int k = T_FIRST; k < T_STEPS; ++k)
            if (k < 0 || k >= 18) t_step(std::integral_constant<int, 0>{}, lds, 0, k);
    } else {
#pragma unroll
        for (int k = T_FIRST; k < T_STEPS; ++k)
            if (k < 0 || k >= 18) t_step(std::integral_constant<int, 1>{}, lds, 0, k);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) bload1(j, 0, j);

    int raw_cur = 0, raw_nxt = WG::BUF;
    int v_cur = 0, v_nxt = WG::VBUF;
    int n_msg = 0;                                             // messages this wave has sent (two per unit)

    auto stage_body = [&](auto first_tag, auto half, int chunk) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int nchunk = chunk + 1;
        nchunk = nchunk == n ? 0 : nchunk;
        const float *traw = lds + raw_nxt;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 9; ++pr)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    const int fq = 2 * pr + sidx, m = pr * 8 + e * 2 + sidx;
                    const float4 wv4 = Wq[fq % 6], vv4 = Bq[fq % 6];
                    const float we = e == 0 ? wv4.x : e == 1 ? wv4.y : e == 2 ? wv4.z : wv4.w;
                    const float ve = e == 0 ? vv4.x : e == 1 ? vv4.y : e == 2 ? vv4.z : vv4.w;
                    if (FIRST && e == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[fq] = __builtin_amdgcn_mfma_f32_16x16x4f32(we, ve, zero, 0, 0, 0);
                    } else
                        acc[fq] = __builtin_amdgcn_mfma_f32_16x16x4f32(we, ve, acc[fq], 0, 0, 0);
                    // ---- shadow items
                    const int mm = e * 2 + sidx;
                    if (mm < 2 && 2 * pr + 4 + mm < 18) bload1((2 * pr + 4 + mm) % 6, v_cur, 2 * pr + 4 + mm);       // B operands 4 ahead
                    if (mm == 2 || mm == 6) {                                                                        // weights 4 frequencies ahead
                        const int wf = 2 * pr + WLEAD + (mm == 6);
                        if (wf < 18) wload1(wf % 6, chunk, wf);
                        else wload1(wf % 6, nchunk, wf - 18);
                    }
                    if (m < 36) t_step(half, traw, v_nxt, m - 36);                                    // the next chunk's transform: reads ...
                    if (m >= 38 && m < 44 && !(m & 1)) t_step(half, traw, v_nxt, 18 + ((m - 38) >> 1));   // ... column pairs ...
                    if (m >= 44 && m < 50 && !(m & 1)) t_step(half, traw, v_nxt, 21 + ((m - 44) >> 1));   // ... rows ...
                    if (m >= 50 && m < 59) t_step(half, traw, v_nxt, 24 + (m - 50));                  // ... stores
                    if (m >= 59 && m - 59 < NI) lwrite1(m - 59, raw_cur);                        // raw(chunk + 2): registers -> LDS
                    if (m > BAR_M && m - (BAR_M + 1) < NI) gload1(m - (BAR_M + 1));              // raw(chunk + 3) -> registers
                    if (m == BAR_M) {                                                            // V(chunk + 1) and raw(chunk + 2) complete in every wave
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < 4; ++j) bload1(j, v_nxt, j);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        advance();
        const int o = raw_cur;
        raw_cur = raw_nxt;
        raw_nxt = o;
        const int v = v_cur;
        v_cur = v_nxt;
        v_nxt = v;
    };

    float *const xme = lds + XCH + wv * XW + lane * 4;
    const float *const xpa = lds + XCH + (wv ^ 4) * XW + lane * 4;
    for (int u = blockIdx.x; u < a.n_units; u += G) {
        if (fh == 0) {
            stage_body(std::true_type{}, std::integral_constant<int, 0>{}, 0);
            for (int chunk = 1; chunk < n; ++chunk) stage_body(std::false_type{}, std::integral_constant<int, 0>{}, chunk);
        } else {
            stage_body(std::true_type{}, std::integral_constant<int, 1>{}, 0);
            for (int chunk = 1; chunk < n; ++chunk) stage_body(std::false_type{}, std::integral_constant<int, 1>{}, chunk);
        }

        // ================= unit epilogue =================
        const int cq = (lane >> 4) & 1, hf = lane >> 5;
        const int c0 = g * 32 + co * 8 + 4 * cq;
        const int oy = by * 8 + 4 * (t16 >> 3) + 2 * fh + hf, ox = bx * 32 + 4 * (t16 & 7);     // this lane finishes ONE row of its tile
        const f32x4 bf = *reinterpret_cast<const f32x4 *>(a.params + c0);
        const f32x4 bm = *reinterpret_cast<const f32x4 *>(a.params + a.CoutPad + c0);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.params + 2 * a.CoutPad + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.params + 3 * a.CoutPad + c0);
        unsigned rvoff[4], ovoff[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const bool in = (oy < a.outH) & (ox + px < a.outW) & (c0 < a.Cout);
            const int pix = oy * a.outW + ox + px;
            rvoff[px] = in ? (unsigned)((pix * a.Cout + c0) * 4) : OOR;
            ovoff[px] = in ? (unsigned)((pix * a.out_cstride + c0) * 4) : OOR;
        }
        f32x4 rv[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            rv[px] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.residual) rv[px] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvoff[px], 0, 0));
        }
        // column pass, local to a frequency row: T[r][q] from M[r][0..5] (acc[6 r + nu])
        f32x4 T[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const f32x4 s1 = acc[6 * r + 1] + acc[6 * r + 2], d1 = pk_sub4(acc[6 * r + 1], acc[6 * r + 2]);
            const f32x4 s2 = acc[6 * r + 3] + acc[6 * r + 4], dd = pk_sub4(acc[6 * r + 3], acc[6 * r + 4]);
            T[r][0] = acc[6 * r] + s1 + s2;
            T[r][1] = d1 + 2.0f * dd;
            T[r][2] = s1 + 4.0f * s2;
            T[r][3] = d1 + 8.0f * dd + acc[6 * r + 5];
        }
        // row pass, this half's three rows of A^T = [1 1 1 | 1 1 0; 0 1 -1 | 2 -2 0; 0 1 1 | 4 4 0; 0 1 -1 | 8 -8 1]: partial sums of
        // all four output rows; `keep` = rows 2 fh, 2 fh + 1, `give` = the partner's rows
        f32x4 keep[2][4], give[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (fh == 0) {
                const f32x4 sm = T[1][q] + T[2][q], df = pk_sub4(T[1][q], T[2][q]);
                keep[0][q] = T[0][q] + sm;
                keep[1][q] = df;
                give[0][q] = sm;
                give[1][q] = df;
            } else {
                const f32x4 sm = T[0][q] + T[1][q], df = pk_sub4(T[0][q], T[1][q]);
                give[0][q] = sm;
                give[1][q] = 2.0f * df;
                keep[0][q] = 4.0f * sm;
                keep[1][q] = 8.0f * df + T[2][q];
            }
        }
        // exchange with the partner wave, two rounds of four float4 per lane (columns 0, 1 then 2, 3)
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const int seq = n_msg + round + 1;
            while (__hip_atomic_load(&xflag[8 + (wv ^ 4)], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq - 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4 *>(__builtin_assume_aligned(xme + (o * 2 + j) * 256, 16)) = give[o][2 * round + j];
            __hip_atomic_store(&xflag[wv], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&xflag[wv ^ 4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    keep[o][2 * round + j] += *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(xpa + (o * 2 + j) * 256, 16));
            __hip_atomic_store(&xflag[8 + wv], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        n_msg += 2;
        __builtin_amdgcn_s_setprio(1);                          // only now: a wave polling its partner's counter must not outrank the partner on their SIMD
        // lanes 0..31 hold conv_f, lanes 32..63 conv_m: exchange the two rows so that the lower half-wave owns row 2 fh and the upper
        // half row 2 fh + 1, f in one register and m in the other
        {
            constexpr float LOG2E = 1.44269504088896341f;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                u32x4 u0 = __builtin_bit_cast(u32x4, keep[0][px]), u1 = __builtin_bit_cast(u32x4, keep[1][px]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(u0[k], u1[k], false, false);
                    u0[k] = sw[0];
                    u1[k] = sw[1];
                }
                f32x4 f = __builtin_bit_cast(f32x4, u0) + bf;
                const f32x4 mm = (__builtin_bit_cast(f32x4, u1) + bm) * -LOG2E;
                if (a.elu) {
                    const f32x4 fe = f * LOG2E;
                    f32x4 e;
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f(fe[k]);
                    e = e + f32x4{-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : e[k];
                }
                f32x4 sg, t;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_exp2f(mm[k]);
                t = t + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(t[k]);
                const f32x4 v = (f * sg) * sc + sh + rv[px];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, ovoff[px], 0, 0);
            }
        }
        step_tile(by, bx);
        __builtin_amdgcn_s_setprio(0);
    }
}
#endif  // READ_DEBUG_KNOBS

// ------------------------------------------------------------------------------------------
// 3x3 / stride-1 layers with at most FOUR output channels on the vector pipe (READ's output layer, feat_extract.5: 32 -> 3).
//
// The matrix cores have no shape for this layer: conv_f and conv_m together have 6 output channels — 6 rows of a 16- or 32-row
// MFMA (the F(2x2) kernel ran it at 62 us with three of four waves skipping their MFMAs; a 16x16x4 tiling would spend 25 us
// multiplying 10 rows of zeros).  On the vector pipe nothing is padded: thread = output pixel, COUT + COUT accumulators, and per
// (tap, input channel) 2 COUT v_fmac_f32 whose multiplier sits in an SGPR — the weights are wave-uniform, stream through the
// scalar cache ([tap][cin][f0..f3 m0..m3], two s_load_dwordx16 per four channels) and cost no vector register or LDS read.
// 9 x 32 x 6 = 1728 FMAs per pixel at Cout = 3.
//   * plain v_fmac_f32, not v_pk_fma_f32: tools/valu_probe.py measures 1.8 ns per wave-instruction per SIMD for the former and
//     2.2 ns for the latter at 4 waves per SIMD — a packed FMA buys 1.16x the flops of a plain one, not 2x, and the fourth
//     (padding) channel of a pair costs more than that;
//   * workgroup = 8 x 32 output pixels; the 10 x 34 halo tile goes through LDS CPH input channels at a time (buffer loads: pixels
//     outside the image carry an out-of-range offset and arrive as zeros — the zero padding), padded to CPH + 4 floats per pixel
//     so that the 64 lanes' ds_read_b128 of a (tap, channel quad) hit all banks; 16 KiB at CPH = 8: all 1672 workgroups of a
//     1216 x 352 frame are resident at once, and the next phase's loads are in flight under this phase's FMAs;
//   * epilogue as everywhere (bias, ELU, sigmoid gate, BatchNorm scale / shift), RGBA-padded store with the fill value.
// Measured (profiles/README.md, round 4): 45 us for the frame against 62 on the F(2x2) kernel.  Taking the parts out one at a time
// (results invalid, timing only): scalar weight loads 12 us (61 MB through scalar caches shared between CUs), the tile's global
// loads 17 us (73 MB with the halo), FMAs 10 us, LDS reads 3 us, and they add rather than overlap; two pixels per thread (half the
// scalar traffic, PPT = 2: measured, not instantiated) and larger phases (CPH = 16, 32) were slower (50 .. 53 us and 48, 52 us).
// ------------------------------------------------------------------------------------------
template <int CIN, int CPH, int PPT, int COUT>
__global__ __launch_bounds__(256) void gated_conv_smallc_kernel(const ConvKArgs a)
{
    // CPH = input channels per LDS phase; PPT = output pixels per thread (rows py + 8 r of an 8 PPT x 32 tile: the weights of a
    // group are fetched once per wave and used PPT times); COUT = 3 or 4 accumulated channels per branch.
    constexpr int TH = 8 * PPT, IH = TH + 2, IW = 34, PS = CPH + 4, Q4 = CPH / 4, NPH = CIN / CPH, QT = CIN / 4;
    constexpr int NE = IH * IW * Q4, NI = (NE + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[IH * IW * PS];
    const int tid = threadIdx.x;
    const SrcDev s = a.src[0];
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int y0 = ty * TH - 1, x0 = tx * 32 - 1;
    constexpr unsigned OOR = 0x80000000u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.p), 0, (unsigned)(a.inH * s.W * s.C) * 4u, 0x00020000);
    unsigned goff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256, q = e % Q4, pix = e / Q4, ppy = pix / IW, ppx = pix % IW;
        const bool ok = (e < NE) & (y0 + ppy >= 0) & (y0 + ppy < a.inH) & (x0 + ppx >= 0) & (x0 + ppx < a.inW);
        goff[i] = ok ? (unsigned)(((y0 + ppy) * s.W + x0 + ppx) * s.C + 4 * q) * 4u : OOR;
    }
    // The WHOLE patch is requested here, all phases of a pixel back to back (round 6): fetched phase by phase, 32 bytes of a pixel's
    // 128-byte line at a time and a phase apart, every line came from memory four times — the counters read 223 MB for a 55 MB
    // tensor, and 223 MB are the launch's 44 us (profiles/r6_hbm_traffic_per_kernel.md)
    float4 st[NPH][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)                        // OOR + anything stays out of range
            st[ph][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i], ph * CPH * 4, 0));
    const int py = tid >> 5, px = tid & 31;
    const float *tp = tile + (py * IW + px) * PS;
    // Weights: 32 floats per (tap, channel quad) = two s_load_dwordx16, requested one group AHEAD together with the next
    // ds_read_b128 of pixel values.  Scalar loads return out of order, so the only usable wait is lgkmcnt(0): it sits at the head
    // of a group and covers requests that have had the previous group's packed FMAs (and the other waves' turns) to land.
    // hipcc does not pipeline scalar loads itself (it put every s_load directly in front of its first use, 288 exposed round
    // trips per pixel), hence the asm; the loaded values pass through the wait's operand list and every statement of the loop is
    // volatile, so the order below is the order issued.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int NG = 9 * Q4;                                 // groups of a phase: (tap, quad of the phase)
    const float *wbase = a.wp_sc;
    float fa[PPT][4], ma[PPT][4];
#pragma unroll
    for (int r = 0; r < PPT; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) fa[r][c] = ma[r][c] = 0.f;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph) __syncthreads();                               // the previous phase's reads of the tile are done
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + i * 256, q = e % Q4, pix = e / Q4;
            if (e < NE) *reinterpret_cast<float4 *>(tile + pix * PS + 4 * q) = st[ph][i];
        }
        f32x16 wa, wb;
        asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(wa) : "s"(wbase), "s"((ph * Q4) * 128));
        asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(wb) : "s"(wbase), "s"((ph * Q4) * 128 + 64));
        __syncthreads();
        f32x4 x[PPT];
#pragma unroll
        for (int r = 0; r < PPT; ++r) x[r] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(tp + r * 8 * IW * PS, 16));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (PPT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wa), "+s"(wb), "+v"(x[0]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wa), "+s"(wb), "+v"(x[0]), "+v"(x[PPT - 1]));
            f32x16 na, nb;
            f32x4 nx[PPT];
            if (g + 1 < NG) {
                const int tap = (g + 1) / Q4, q = (g + 1) % Q4;
                const int wo = (tap * QT + ph * Q4 + q) * 128;
                asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(na) : "s"(wbase), "s"(wo));
                asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(nb) : "s"(wbase), "s"(wo + 64));
#pragma unroll
                for (int r = 0; r < PPT; ++r)
                    nx[r] = *reinterpret_cast<const f32x4 *>(
                        __builtin_assume_aligned(tp + ((r * 8 + tap / 3) * IW + tap % 3) * PS + 4 * q, 16));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {                           // input channel of the quad; weights [f0 f1 f2 f3 m0 m1 m2 m3] each
#pragma unroll
                for (int j = 0; j < COUT; ++j) {
                    const float wf = c < 2 ? wa[8 * (c & 1) + j] : wb[8 * (c & 1) + j];
                    const float wm = c < 2 ? wa[8 * (c & 1) + 4 + j] : wb[8 * (c & 1) + 4 + j];
#pragma unroll
                    for (int r = 0; r < PPT; ++r) {
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(fa[r][j]) : "s"(wf), "v"(x[r][c]));
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ma[r][j]) : "s"(wm), "v"(x[r][c]));
                    }
                }
            }
            if (g + 1 < NG) {
                wa = na;
                wb = nb;
#pragma unroll
                for (int r = 0; r < PPT; ++r) x[r] = nx[r];
            }
        }
    }
    const int ox = tx * 32 + px;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int oy = ty * TH + py + 8 * r;
        if (oy >= a.outH || ox >= a.outW) continue;
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float f = fa[r][c] + a.params[c];
            const float m = ma[r][c] + a.params[a.CoutPad + c];
            if (a.elu) f = elu1(f);
            o[c] = c < a.Cout ? (f * sigmoidf(m)) * a.params[2 * a.CoutPad + c] + a.params[3 * a.CoutPad + c] : a.out_fill;
        }
        float *op = a.out + ((size_t)oy * a.outW + ox) * a.out_cstride;
        if (a.out_cstride == 4 && (a.fill_pad || a.Cout == 4))
            *reinterpret_cast<float4 *>(op) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.Cout || (a.fill_pad && c < a.out_cstride)) op[c] = o[c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// 1x1 layers in the "pixel-lane" orientation: weights are the MFMA A operand, activations the B operand.
//
//   D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]        (v_mfma_f32_32x32x2_f32: lane = pixel, registers = channels)
//
// With NHWC activations both sides of the product are then plain 128-bit global accesses and nothing is transposed:
//   * B operand: lane (h = lane >> 5, j = lane & 31) supplies k = h (+2, +4, +6 inside a float4 step): the float4 at channel
//     8s + 4h of pixel j — one global_load_dwordx4 per lane and k8 step, straight from the tensor (no LDS tile);
//   * D: lane (h, j) ends up with channels 8q + 4h + {0..3} (q = register quad) of pixel j, for conv_f and conv_m alike:
//     the epilogue is lane-local and loads / stores 4 consecutive channels at a time (residual, pre-activation addend,
//     output), 4 + 4 memory instructions per 32x32 tile instead of the 16 + 16 dword accesses of the other orientation;
//   * A operand = the layer's packed weights (same fragment order as everywhere else), copied ONCE per persistent
//     workgroup into LDS and read from there with conflict-free ds_read_b128 — a wave then streams pixel tiles with no
//     barrier and no LDS writes in its loop.
// A wave's unit = PT tiles of 32 consecutive pixels (flattened y*W + x) x GW 32-channel groups; the activation ring holds
// four k8 steps per tile and is refilled four steps ahead of the MFMAs, across unit boundaries.  Sources of other levels
// (nearest resampling, unet.py:239-254) are addressed per lane.  Measured next to the LDS-tiled kernels in
// profiles/README.md (the 1x1 layers are short-K: their time is the epilogue and HBM traffic, not MFMAs).
// FULLQ: Cin % 32 == 0 (whole quads of k8 steps): the ring refill and the MFMA group of a step then sit in straight-line code.
// With the `step < nsteps` tests in the loop hipcc's waitcnt pass loses count of the loads in flight and puts s_waitcnt vmcnt(0)
// in front of every MFMA group AND behind every refill (seen in the ISA): the four-steps-ahead ring then prefetches nothing.
template <int PT, int GW, bool FULLQ = false>
__global__ __launch_bounds__(256, (PT * GW >= 4 ? 2 : 3)) void gated_conv_px_kernel(const ConvKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float4 wl[];      // [k8 step][2 GW tiles (f, m per group)][lane]
    constexpr int T = 2 * GW;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, lp = lane & 31;
    const int nsteps = a.nchunks;                                    // k8 steps = Cin / 8
    const int nquads = (nsteps + 3) >> 2;                            // the ring advances in quads; steps >= nsteps are empty
    const int gsets = (a.CoutPad >> 5) / GW;
    const int gs = blockIdx.x % gsets;                               // channel-group set of this workgroup
    {
        const int NT = a.CoutPad >> 4;
        const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp);
        for (int i = threadIdx.x; i < nsteps * T * 64; i += 256) {
            const int l = i & 63, t = (i >> 6) % T, st = (i >> 6) / T;
            wl[i] = wp4[((size_t)st * NT + gs * T + t) * 64 + l];
        }
    }
    __syncthreads();
    const int npix = a.outH * a.outW;
    const int wslots = (gridDim.x / gsets) * 4, w0 = (blockIdx.x / gsets) * 4 + wave;
    if (w0 >= a.n_units) return;

    // ---- load cursor: (unit, step) of the next activation fragment, four steps ahead of the MFMAs
    int lu = w0, lstep = 0, lsrc = 0, lcoff = 0;
    SrcDev lsd = a.src[0];
    int ly[PT], lx[PT];
    auto set_load_unit = [&]() {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            int p = (lu * PT + pt) * 32 + lp;
            p = p < npix ? p : npix - 1;                             // past the image (and past the last unit): a valid pixel
            ly[pt] = p / a.outW;
            lx[pt] = p - ly[pt] * a.outW;
        }
    };
    float4 ring[4][PT];
    auto load_step = [&](int slot) {
        if (FULLQ || lstep < nsteps) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int sy = (ly[pt] << lsd.sl) >> lsd.sr, sx = (lx[pt] << lsd.sl) >> lsd.sr;
                ring[slot][pt] = *reinterpret_cast<const float4 *>(lsd.p + (sy * lsd.W + sx) * lsd.C + lcoff + 4 * half);
            }
            lcoff += 8;
            if (lcoff >= lsd.C && lsrc + 1 < a.n_src) {
                ++lsrc;
                lcoff = 0;
                lsd = a.src[lsrc];
            }
        }
        if (++lstep == 4 * nquads) {
            lstep = 0;
            lu += wslots;
            lsrc = 0;
            lcoff = 0;
            lsd = a.src[0];
            set_load_unit();
        }
    };
    set_load_unit();
#pragma unroll
    for (int e = 0; e < 4; ++e) load_step(e);

    const bool quad_res = a.residual != nullptr;
    for (int u = w0; u < a.n_units; u += wslots) {
        floatx16 acc[PT][GW][2];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int g = 0; g < GW; ++g)
#pragma unroll
                for (int fm = 0; fm < 2; ++fm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[pt][g][fm][r] = 0.0f;

        float4 w[2][T];
#pragma unroll
        for (int t = 0; t < T; ++t) w[0][t] = wl[t * 64 + lane];
        for (int q = 0; q < nquads; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int st = 4 * q + e;
                if (FULLQ || st < nsteps) {
                    const int sn = st + 1 < nsteps ? st + 1 : 0;     // next step's weights (the unit's first again at the end)
#pragma unroll
                    for (int t = 0; t < T; ++t) w[(e + 1) & 1][t] = wl[(sn * T + t) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                            for (int g = 0; g < GW; ++g)
#pragma unroll
                                for (int fm = 0; fm < 2; ++fm) {
                                    const float4 wv = w[e & 1][2 * g + fm], xv = ring[e][pt];
                                    const float av = j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w;
                                    const float bv = j == 0 ? xv.x : j == 1 ? xv.y : j == 2 ? xv.z : xv.w;
                                    acc[pt][g][fm] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[pt][g][fm], 0, 0, 0);
                                }
                }
                load_step(e);                                        // slot e now takes step st + 4 of the stream
            }
        }
        // (nsteps odd multiples of 1..3 leave w[] parity off by the skipped steps: reload at the next unit's start)

        // ---- epilogue: lane = (pixel, channel quad), everything 128 bits wide
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = (u * PT + pt) * 32 + lp;
            const bool p_ok = p < npix;
            const int y = p / a.outW, x = p - y * a.outW;
            const size_t pre_pix = a.pre ? ((size_t)(y >> a.pre_shift) * a.pre_W + (x >> a.pre_shift)) * a.pre_cstride : 0;
            // bilinear addend (pre_bil): the four neighbours and weights of nn.Upsample(x4, bilinear, align_corners=False) at (y, x),
            // computed exactly as bilinear_up4_kernel does (area_pixel_compute_source_index: 0.25 (dst + 0.5) - 0.5 clamped at 0)
            size_t bp00 = 0, bp01 = 0, bp10 = 0, bp11 = 0;
            float bly0 = 0.f, bly1 = 0.f, blx0 = 0.f, blx1 = 0.f;
            if (a.pre_bil) {
                float sy = 0.25f * ((float)y + 0.5f) - 0.5f, sx = 0.25f * ((float)x + 0.5f) - 0.5f;
                sy = sy < 0.f ? 0.f : sy;
                sx = sx < 0.f ? 0.f : sx;
                const int y0 = (int)sy, x0 = (int)sx;
                const int y1 = y0 + (y0 < a.pre_H - 1 ? 1 : 0), x1 = x0 + (x0 < a.pre_W - 1 ? 1 : 0);
                bly1 = sy - (float)y0;
                blx1 = sx - (float)x0;
                bly0 = 1.f - bly1;
                blx0 = 1.f - blx1;
                bp00 = ((size_t)y0 * a.pre_W + x0) * a.pre_cstride;
                bp01 = ((size_t)y0 * a.pre_W + x1) * a.pre_cstride;
                bp10 = ((size_t)y1 * a.pre_W + x0) * a.pre_cstride;
                bp11 = ((size_t)y1 * a.pre_W + x1) * a.pre_cstride;
            }
#pragma unroll
            for (int g = 0; g < GW; ++g)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int c0 = (gs * GW + g) * 32 + 8 * qd + 4 * half;
                    if (!p_ok || c0 >= a.Cout) continue;
                    f32x4 f, m;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f[k] = acc[pt][g][0][4 * qd + k];
                        m[k] = acc[pt][g][1][4 * qd + k];
                    }
                    f += *reinterpret_cast<const f32x4 *>(a.params + c0);
                    m += *reinterpret_cast<const f32x4 *>(a.params + a.CoutPad + c0);
                    if (a.pre_bil) {
                        const float *pf = a.pre + a.pre_foff + c0, *pm = a.pre + a.pre_moff + c0;
                        const f32x4 f00 = *reinterpret_cast<const f32x4 *>(pf + bp00), f01 = *reinterpret_cast<const f32x4 *>(pf + bp01);
                        const f32x4 f10 = *reinterpret_cast<const f32x4 *>(pf + bp10), f11 = *reinterpret_cast<const f32x4 *>(pf + bp11);
                        const f32x4 m00 = *reinterpret_cast<const f32x4 *>(pm + bp00), m01 = *reinterpret_cast<const f32x4 *>(pm + bp01);
                        const f32x4 m10 = *reinterpret_cast<const f32x4 *>(pm + bp10), m11 = *reinterpret_cast<const f32x4 *>(pm + bp11);
                        f += bly0 * (blx0 * f00 + blx1 * f01) + bly1 * (blx0 * f10 + blx1 * f11);
                        m += bly0 * (blx0 * m00 + blx1 * m01) + bly1 * (blx0 * m10 + blx1 * m11);
                    } else if (a.pre) {
                        f += *reinterpret_cast<const f32x4 *>(a.pre + pre_pix + a.pre_foff + c0);
                        m += *reinterpret_cast<const f32x4 *>(a.pre + pre_pix + a.pre_moff + c0);
                    }
                    float *op = a.out + (size_t)p * a.out_cstride + c0;
                    if (a.linear) {
                        *reinterpret_cast<f32x4 *>(op) = f;
                        *reinterpret_cast<f32x4 *>(op + a.Cout) = m;
                        continue;
                    }
                    constexpr float LOG2E = 1.44269504088896341f;
                    if (a.elu) {
                        const f32x4 fe = f * LOG2E;
#pragma unroll
                        for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : __builtin_amdgcn_exp2f(fe[k]) - 1.0f;
                    }
                    const f32x4 mm = m * -LOG2E;
                    f32x4 sg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mm[k]));
                    f32x4 v = (f * sg) * *reinterpret_cast<const f32x4 *>(a.params + 2 * a.CoutPad + c0) +
                              *reinterpret_cast<const f32x4 *>(a.params + 3 * a.CoutPad + c0);
                    if (quad_res) v += *reinterpret_cast<const f32x4 *>(a.residual + (size_t)p * a.Cout + c0);
                    *reinterpret_cast<f32x4 *>(op) = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 1x1 layers with SPLIT fp32 operands on the f16 matrix cores, pixel-lane orientation (round 6: the 18 launches outside the
// 3x3 family that are 1x1 convolutions — SCM tails, AFF, Convs.k — took 635 us per frame on the fp32 matrix cores, 35 - 60 TF).
//
//   D[cout][pixel] = sum_k W[cout][k] X[k][pixel]     v_mfma_f32_32x32x16_f16, lane (h = lane >> 5, j = lane & 31):
//                                                     A[row j][k = 8 h + e], B[k = 8 h + e][pixel j], D as in gated_conv_px_kernel
//
// The arithmetic is the direct split-operand kernel's (gated_conv_d3h_kernel): x = xh + 2^-11 xl (two f16 pieces formed from the
// fp32 activation in registers), w s = wh + wl (host packer, power-of-two row scale s), products (2^-11 wh) xl + wl xh + wh xh into
// one fp32 accumulator, 1 / s in the epilogue.  The structure is the pixel-lane kernel's: the group set's weight fragments are
// copied once per persistent workgroup into LDS ([k16 step][tile][wh | wl][lane] x 16 bytes), a wave streams tiles of 32 pixels
// with no barrier; lane (h, j) loads the 8 consecutive channels 16 step + 8 h .. + 7 of pixel j (two float4) four steps ahead of
// the MFMAs, from whichever concatenated source holds them (sources are multiples of 8 channels: the two half-waves may read
// different sources — cat[x(8), main(P - 8)] of SCM.conv; UNI = every source a multiple of 16, one cursor).
// Three 32-cycle MFMAs replace eight 64-cycle ones per 16 channels and tile pair; what is left is the activation stream.
//
// MODE 2 (TAPS): the same kernel as an implicit GEMM for 3x3 / stride-1 layers over ONE source of C = 8, 16 or 32 channels (the first layer
// and the SCM heads read the 8-channel descriptor pyramid; 67 TF on the fp32 direct kernel): k = tap C + channel, 9 C padded to whole
// k16 steps (read_conv_pack_t3h_host), lane (h, j) loads the 8 channels of ITS tap at pixel j + (dy, dx) through a buffer descriptor
// (a tap outside the image, or tap 9 of the padding, reads zeros); neighbouring taps hit the same lines in L1 / L2.
template <int PT, int GW, bool FULLQ, int MODE>
__global__ __launch_bounds__(256, 2) void gated_conv_pxh_kernel(const ConvKArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32x4 wlh[];      // [k16 step][T tiles (f, m per group)][wh | wl][lane]
    constexpr bool UNI = MODE != 0, TAPS = MODE == 2;
    constexpr int T = 2 * GW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, lp = lane & 31;
    const int nsteps = a.nchunks;                                    // k16 steps = Cin / 16
    const int nquads = (nsteps + 3) >> 2;                            // the ring advances in quads; steps >= nsteps are empty
    const int gsets = (a.CoutPad >> 5) / GW;
    // workgroup -> (group set, index inside the set), XCD-aware: consecutive workgroup ids go to the eight XCDs in turn, so the group sets
    // that read the SAME pixels are ids xcd + 8 (gsets k + gs) — they share an XCD and its L2: the pixels come from memory once per
    // XCD instead of once per group set (gridDim.x = per_set x gsets with per_set a multiple of 8)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gs = slot % gsets, wgi = (slot / gsets) * 8 + xcd;     // channel-group set of this workgroup; its index among the set's workgroups
#if defined(PXH_ABL)                 // attribution probes (results invalid): 1 no epilogue memory traffic, 2 activation loads from one resident line per lane,
    constexpr int abl = PXH_ABL;     //   4 no MFMAs, 16 no weight copy.  Compile-time in variant builds (tools/pxh_probe.py), run-time in the debug library
#elif defined(READ_DEBUG_KNOBS)
    const int abl = a.ablate;
#else
    constexpr int abl = 0;
#endif
    if (!(abl & 16)) {
        // the set's fragments: T consecutive 2 KiB blocks per k16 step.  Eight 16-byte loads per thread in flight, then their stores
        // (one load, one store at a time cost a memory round trip per KiB: 16 round trips for 64 KiB)
        const int NT = a.CoutPad >> 4;                               // 32-row tiles of the layer: (f, m) per group
        const u32x4 *wp4 = reinterpret_cast<const u32x4 *>(a.wp_d3h);
        const int total = nsteps * T * 128;
        for (int i0 = threadIdx.x; i0 < total; i0 += 8 * 256) {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * 256;
                const int ic = i < total ? i : total - 1;
                v[k] = wp4[((size_t)((ic >> 7) / T) * NT + gs * T + (ic >> 7) % T) * 128 + (ic & 127)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + k * 256 < total) wlh[i0 + k * 256] = v[k];
        }
    }
    // the set's epilogue parameters behind the fragments: b_f, b_m, BN scale, BN shift, 1 / s_f, 1 / s_m  [6][EP = 32 GW]
    constexpr int EP = 32 * GW;
    float *const epar = reinterpret_cast<float *>(wlh + nsteps * T * 128);
    for (int i = threadIdx.x; i < 6 * EP; i += 256) {
        const float *const wsc = reinterpret_cast<const float *>(a.wp_d3h) + (size_t)nsteps * 32 * a.CoutPad;     // 1 / s: [f | m][CoutPad]
        const int arr = i / EP, c = gs * EP + i % EP;
        epar[i] = arr < 4 ? a.params[arr * a.CoutPad + c] : wsc[(arr - 4) * a.CoutPad + c];
    }
    __syncthreads();
    const int npix = a.outH * a.outW;
    const int wslots = (gridDim.x / gsets) * 4, w0 = wgi * 4 + wave;
    if (w0 >= a.n_units) return;

    // ---- load cursor: (unit, step) of the next activation fragment, four steps ahead of the MFMAs.  Cursor A = channels 16 step ..
    // + 7 (lower half-wave), cursor B = 16 step + 8 .. + 15 (upper half-wave); both wave-uniform (scalar registers)
    int lu = w0, lstep = 0;
    int srcA = 0, coffA = 0, srcB = 0, coffB = 0;
    SrcDev sdA = a.src[0], sdB = a.src[0];
    auto reset_cursors = [&]() {
        srcA = 0;
        coffA = 0;
        sdA = a.src[0];
        if (!UNI) {
            srcB = 0;
            coffB = 8;
            if (a.n_src > 1 && a.src[0].C == 8) {
                srcB = 1;
                coffB = 0;
            }
            sdB = a.src[srcB];
        }
    };
    auto advance16 = [&](int &src, int &coff, SrcDev &sd) {
        coff += 16;
        while (src + 1 < a.n_src && coff >= sd.C) {
            coff -= sd.C;
            ++src;
            sd = a.src[src];
        }
    };
    int ly[PT], lx[PT];
    auto set_load_unit = [&]() {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            int p = (lu * PT + pt) * 32 + lp;
            p = p < npix ? p : npix - 1;                             // past the image (and past the last unit): a valid pixel
            ly[pt] = p / a.outW;
            lx[pt] = p - ly[pt] * a.outW;
        }
    };
    float4 ring[4][PT][2];
    const auto trsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.src[0].p), 0, (unsigned)(a.inH * a.inW * a.src[0].C) * 4u, 0x00020000);
    const int cshift = a.tiles_y;                                    // TAPS: log2 of the source's channels
    auto load_step = [&](int slot) {
        if (TAPS) {
            if (lstep < nsteps) {
                const int k0 = 16 * lstep + 8 * half;
                const int tap = k0 >> cshift, coff = k0 & ((1 << cshift) - 1);
                const int ty = (tap * 11) >> 5, dy = ty - 1, dx = tap - 3 * ty - 1;          // tap / 3 for tap <= 10
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int yy = ly[pt] + dy, xx = lx[pt] + dx;
                    const bool ok = (tap < 9) & ((unsigned)yy < (unsigned)a.inH) & ((unsigned)xx < (unsigned)a.inW);
                    const unsigned voff = ok ? (unsigned)((((yy * a.inW + xx) << cshift) + coff) * 4) : 0x80000000u;
                    ring[slot][pt][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(trsrc, voff, 0, 0));
                    ring[slot][pt][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(trsrc, voff, 16, 0));
                }
            }
        } else if (FULLQ || lstep < nsteps) {
            const float *p;
            int sW, sC, sl, sr, coff;
            if (UNI) {
                p = sdA.p;
                sW = sdA.W;
                sC = sdA.C;
                sl = sdA.sl;
                sr = sdA.sr;
                coff = coffA + 8 * half;
            } else {
                p = half ? sdB.p : sdA.p;
                sW = half ? sdB.W : sdA.W;
                sC = half ? sdB.C : sdA.C;
                sl = half ? sdB.sl : sdA.sl;
                sr = half ? sdB.sr : sdA.sr;
                coff = half ? coffB : coffA;
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int sy = (ly[pt] << sl) >> sr, sx = (lx[pt] << sl) >> sr;
                const float4 *q = reinterpret_cast<const float4 *>(p + (sy * sW + sx) * sC + coff);
                if (abl & 2) q = reinterpret_cast<const float4 *>(p + lane * 8);
                ring[slot][pt][0] = q[0];
                ring[slot][pt][1] = q[1];
            }
            advance16(srcA, coffA, sdA);
            if (!UNI) advance16(srcB, coffB, sdB);
        }
        if (++lstep == 4 * nquads) {
            lstep = 0;
            lu += wslots;
            reset_cursors();
            set_load_unit();
        }
    };
    reset_cursors();
    set_load_unit();
#pragma unroll
    for (int e = 0; e < 4; ++e) load_step(e);

    auto split2 = [](float x, float y, unsigned &hi, unsigned &lo) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x));                    // x - f32(hi), exact
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(y));
        const f32x2 rs = f32x2{r0, r1} * f32x2{2048.0f, 2048.0f};
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(rs.x), "v"(rs.y));
    };

    const bool quad_res = a.residual != nullptr;
    for (int u = w0; u < a.n_units; u += wslots) {
        floatx16 acc[PT][T];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][t][r] = 0.0f;

        for (int q = 0; q < nquads; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int st = 4 * q + e;
                if (FULLQ || st < nsteps) {
                    u32x4 wh[T], wl[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        wh[t] = wlh[((st * T + t) * 2) * 64 + lane];
                        wl[t] = wlh[((st * T + t) * 2 + 1) * 64 + lane];
                    }
                    u32x4 bh[PT], bl[PT];
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        const float4 v0 = ring[e][pt][0], v1 = ring[e][pt][1];
                        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                        split2(v0.x, v0.y, h0, l0);
                        split2(v0.z, v0.w, h1, l1);
                        split2(v1.x, v1.y, h2, l2);
                        split2(v1.z, v1.w, h3, l3);
                        bh[pt] = u32x4{h0, h1, h2, h3};
                        bl[pt] = u32x4{l0, l1, l2, l3};
                    }
                    const _Float16 k11 = (_Float16)0x1p-11f;
                    f16x8 as[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) as[t] = __builtin_bit_cast(f16x8, wh[t]) * f16x8{k11, k11, k11, k11, k11, k11, k11, k11};
                    if (abl & 4) {
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                acc[pt][t][0] += __builtin_bit_cast(float, bl[pt][0] ^ bh[pt][1] ^ wl[t][0]);
                                acc[pt][t][1] += __builtin_bit_cast(float, bl[pt][2] ^ bh[pt][3] ^ __builtin_bit_cast(u32x4, as[t])[1]);
                                acc[pt][t][2] += __builtin_bit_cast(float, bl[pt][1] ^ bh[pt][0] ^ wh[t][2]);
                                acc[pt][t][3] += __builtin_bit_cast(float, bl[pt][3] ^ bh[pt][2] ^ wh[t][3]);
                            }
                        load_step(e);
                        continue;
                    }
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[t], __builtin_bit_cast(f16x8, bl[pt]), acc[pt][t], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl[t]), __builtin_bit_cast(f16x8, bh[pt]), acc[pt][t], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh[t]), __builtin_bit_cast(f16x8, bh[pt]), acc[pt][t], 0, 0, 0);
                }
                load_step(e);                                        // slot e now takes step st + 4 of the stream
            }
        }

        // ---- epilogue: lane = (pixel, channel quad), everything 128 bits wide.  Parameters come from LDS (a global load here would
        // queue behind the next unit's activation loads, which are already in flight, and wait for them: loads return in order);
        // the addends of a channel group are requested together, in front of the arithmetic.
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = (u * PT + pt) * 32 + lp;
            const bool p_ok = p < npix;
            const int pc = p_ok ? p : npix - 1;
            const int y = pc / a.outW, x = pc - y * a.outW;
            const float *const pp = a.pre ? a.pre + ((size_t)(y >> a.pre_shift) * a.pre_W + (x >> a.pre_shift)) * a.pre_cstride : nullptr;
#pragma unroll
            for (int g = 0; g < GW; ++g) {
                f32x4 af[4], am[4];
                if (a.pre) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int c0 = (gs * GW + g) * 32 + 8 * qd + 4 * half, cc = c0 < a.Cout ? c0 : a.Cout - 4;
                        af[qd] = *reinterpret_cast<const f32x4 *>(pp + a.pre_foff + cc);
                        am[qd] = *reinterpret_cast<const f32x4 *>(pp + a.pre_moff + cc);
                    }
                }
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int cl = g * 32 + 8 * qd + 4 * half, c0 = gs * GW * 32 + cl;
                    f32x4 f, m;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f[k] = acc[pt][2 * g][4 * qd + k];
                        m[k] = acc[pt][2 * g + 1][4 * qd + k];
                    }
                    f = __builtin_elementwise_fma(f, *reinterpret_cast<const f32x4 *>(&epar[4 * EP + cl]), *reinterpret_cast<const f32x4 *>(&epar[cl]));
                    m = __builtin_elementwise_fma(m, *reinterpret_cast<const f32x4 *>(&epar[5 * EP + cl]), *reinterpret_cast<const f32x4 *>(&epar[EP + cl]));
                    const int cc = c0 < a.Cout ? c0 : a.Cout - 4;
                    if (a.pre) {
                        f += af[qd];
                        m += am[qd];
                    }
                    const bool ok = p_ok && c0 < a.Cout;
                    float *op = a.out + (size_t)pc * a.out_cstride + (c0 < a.Cout ? c0 : 0);
                    if (abl & 1) op = a.out + lane * 4;
                    if (a.linear) {
                        if (ok) {
                            *reinterpret_cast<f32x4 *>(op) = f;
                            *reinterpret_cast<f32x4 *>(op + a.Cout) = m;
                        }
                        continue;
                    }
                    constexpr float LOG2E = 1.44269504088896341f;
                    if (a.elu) {
                        const f32x4 fe = f * LOG2E;
#pragma unroll
                        for (int k = 0; k < 4; ++k) f[k] = f[k] > 0.0f ? f[k] : __builtin_amdgcn_exp2f(fe[k]) - 1.0f;
                    }
                    const f32x4 mm = m * -LOG2E;
                    f32x4 sg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sg[k] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(mm[k]));
                    f32x4 v = (f * sg) * *reinterpret_cast<const f32x4 *>(&epar[2 * EP + cl]) + *reinterpret_cast<const f32x4 *>(&epar[3 * EP + cl]);
                    if (quad_res) v += *reinterpret_cast<const f32x4 *>(a.residual + (abl & 1 ? (size_t)lane * 4 : (size_t)pc * a.Cout + cc));
                    if (ok) *reinterpret_cast<f32x4 *>(op) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// configuration table
// ------------------------------------------------------------------------------------------
typedef void (*conv_fn)(const ConvKArgs);

struct ConvConfig {
    const char *name;
    int KS, S, KC, P, QG, WM, WN, PF, NB;
    conv_fn fn;
    conv_fn fn_mul;      // variant whose loader multiplies by a second tensor (FAM), or null
    int wave;            // 1: wave-autonomous persistent kernel (WM = WN = 1 means "one wave per unit")
    int wg_per_cu;       // wave kernel: persistent workgroups per CU (LDS / register limited)
    int wino;            // 1: Winograd F(2x2,3x3) kernel (needs desc->wpacked_wino)
};

#define CFGN(KS, S, KC, P, QG, WM, WN, PF, NB)                                                        \
    {"k" #KS "s" #S "c" #KC "_p" #P "q" #QG "m" #WM "n" #WN "f" #PF "b" #NB, KS, S, KC, P, QG, WM, WN, PF, NB, \
     gated_conv_kernel<KS, S, KC, P, QG, WM, WN, false, PF, NB>, nullptr}
#define CFGNM(KS, S, KC, P, QG, WM, WN, PF, NB)                                                       \
    {"k" #KS "s" #S "c" #KC "_p" #P "q" #QG "m" #WM "n" #WN "f" #PF "b" #NB, KS, S, KC, P, QG, WM, WN, PF, NB, \
     gated_conv_kernel<KS, S, KC, P, QG, WM, WN, false, PF, NB>,                                      \
     gated_conv_kernel<KS, S, KC, P, QG, WM, WN, true, PF, NB>}
#define CFG(KS, S, KC, P, QG, WM, WN, PF) CFGN(KS, S, KC, P, QG, WM, WN, PF, 2)
#define CFGM(KS, S, KC, P, QG, WM, WN, PF)                                                           \
    {"k" #KS "s" #S "c" #KC "_p" #P "q" #QG "m" #WM "n" #WN "f" #PF "b2", KS, S, KC, P, QG, WM, WN, PF, 2, \
     gated_conv_kernel<KS, S, KC, P, QG, WM, WN, false, PF, 2>,                                     \
     gated_conv_kernel<KS, S, KC, P, QG, WM, WN, true, PF, 2>}

#define CFGW(KS, S, KC, P, QG, PF, WGCU)                                                      \
    {"k" #KS "s" #S "c" #KC "_wave_p" #P "q" #QG "f" #PF, KS, S, KC, P, QG, 1, 1, PF, 1,        \
     gated_conv_wave_kernel<KS, S, KC, P, QG, PF>, nullptr, 1, WGCU}

// Entries are addressed by name / parameters (find_config), never by position.
// Order matters: pick_config() takes the first entry whose (ksize, stride, chunk) match and whose
// channel-group coverage divides the layer's groups; remaining groups go to grid.y.  The order below
// follows the measured sweep on MI355X (profiles/r1_sweep_conv.md): small tiles + group splitting over
// grid.y beat wide per-wave tiles at every level (more workgroups in flight, 3-5 waves/SIMD).
const ConvConfig g_configs[] = {
    // 3x3 stride 1, 16-channel chunks (ResBlocks, FAM, AFF second conv, SCM third conv, fe5)
    CFGM(3, 1, 16, 2, 1, 4, 1, 1),  // 8x32 px, one 32-channel group per workgroup
    CFGM(3, 1, 16, 1, 1, 4, 1, 1),  // 4x32 px
    CFGM(3, 1, 16, 2, 1, 2, 2, 1),  // 4x32 px, two groups (B split over waves)
    CFGM(3, 1, 16, 2, 2, 4, 1, 1),  // 8x32 px, two groups per wave
    CFGM(3, 1, 16, 2, 2, 2, 2, 1),  // 4x32 px, four groups
    CFGM(3, 1, 16, 1, 2, 1, 4, 1),  // 1x32 px, eight groups
    CFGM(3, 1, 16, 2, 1, 4, 1, 2),  // 8x32 px, one group, B prefetch depth 2
    CFGM(3, 1, 16, 1, 1, 4, 1, 2),  // 4x32 px, depth 2
    CFGM(3, 1, 16, 2, 1, 2, 2, 2),  // 4x32 px, two groups, depth 2
    CFGN(3, 1, 16, 2, 1, 4, 1, 2, 1),   // single LDS buffer, depth 2
    CFGNM(3, 1, 16, 1, 1, 4, 1, 2, 1),  // 4x32 px, single buffer: best at <= 2 channel groups
    CFGNM(3, 1, 16, 2, 1, 2, 2, 2, 1),  // 4x32 px x 2 groups, single buffer: best at 8 groups
    // 3x3 stride 1, 8-channel chunks (inputs straight from the 8-channel pyramid)
    CFG(3, 1, 8, 2, 1, 4, 1, 2),
    CFG(3, 1, 8, 1, 1, 4, 1, 2),
    // 1x1, 16-channel chunks (SCM, AFF first conv, Convs)
    CFG(1, 1, 16, 2, 1, 4, 1, 1),
    CFG(1, 1, 16, 1, 1, 4, 1, 1),
    CFG(1, 1, 16, 2, 2, 2, 2, 1),
    // 1x1, 32-channel chunks (AFF first conv, Convs: every source a multiple of 32 channels): twice the MFMAs per
    // staged tile and per barrier of the 16-channel chunks
    CFG(1, 1, 32, 2, 1, 4, 1, 1),
    CFG(1, 1, 32, 1, 1, 4, 1, 1),
    CFG(1, 1, 32, 2, 2, 2, 2, 1),
    // 1x1, 8-channel chunks (SCM tail: cat[x(8), main(P-8)])
    CFG(1, 1, 8, 2, 1, 4, 1, 0),
    CFG(1, 1, 8, 1, 1, 4, 1, 0),
    // 3x3 stride 2 (encoder downsampling)
    CFG(3, 2, 16, 1, 1, 4, 1, 1),
    CFG(3, 2, 16, 1, 2, 2, 2, 1),
    CFG(3, 2, 16, 1, 2, 4, 1, 1),
    // wave-autonomous persistent kernels
    CFGW(3, 1, 16, 2, 1, 1, 2),
    CFGW(3, 1, 16, 2, 1, 2, 2),
    CFGW(3, 1, 16, 1, 1, 2, 3),
    CFGW(1, 1, 16, 2, 1, 1, 2),
    CFGW(1, 1, 16, 1, 1, 1, 4),
    CFGW(1, 1, 32, 2, 1, 1, 2),
    CFGW(1, 1, 32, 1, 1, 1, 4),
    CFGW(3, 1, 8, 2, 1, 2, 2),
    {"k3s1c16_p1q1_wino", 3, 1, 16, 1, 1, 4, 1, 1, 2, gated_conv_wino_kernel<false, false>, gated_conv_wino_kernel<false, true>, 0, 0, 1},
    // 4x4 stride 2 (decoder, before the bilinear x4): outputs are 1/4 .. 1/16 scale, so the 16 taps of
    // ONE 1x32-pixel tile are split over the four waves (split-K, WM*WN == 1) to fill the chip
    CFG(4, 2, 16, 1, 1, 1, 1, 1),   // split-K, one group
    CFG(4, 2, 16, 1, 1, 4, 1, 1),
    CFG(4, 2, 16, 1, 2, 2, 2, 1),
};
constexpr int N_CONFIGS = sizeof(g_configs) / sizeof(g_configs[0]);

int find_config(int ks, int s, int kc, int P, int QG, int WM, int WN, int PF, int NB)
{
    for (int i = 0; i < N_CONFIGS; ++i) {
        const ConvConfig &c = g_configs[i];
        if (c.KS == ks && c.S == s && c.KC == kc && c.P == P && c.QG == QG && c.WM == WM && c.WN == WN && c.PF == PF &&
            c.NB == NB)
            return i;
    }
    return -1;
}

int g_prefer_wave = 1;   // read_tuning_set("conv_wave", 0): workgroup-tiled kernels only
int g_wino_wgs = 2;        // read_tuning_set("conv_wino_wgs", 1): one persistent Winograd workgroup per CU (A/B with frames in flight)
int g_conv_px = 1;         // read_tuning_set("conv_px", v): pixel-lane kernel for 1x1 layers — 0 off; 1 / 2 where it measured faster
                           // (64 / 128 accumulator registers per wave); 3 / 4 every layer it fits
int g_kc32 = 1;            // 32-channel chunks for 1x1 layers whose sources are all multiples of 32 (read_tuning_set("conv_kc32", 0): 16)
int g_use_wino = 1 << 30;  // read_tuning_set("conv_wino", max Cin): Winograd kernel for eligible 3x3 layers (0 = off)
int g_w4_grid = 0;        // read_tuning_set("conv_w4_grid", 1): F(4x4) launches with the same number of units per workgroup (measured: see profiles)
int g_w4x2 = 0;            // debug library only: read_tuning_set("conv_w4x2", 1) = the two-waves-per-SIMD F(4x4) kernel (measured slower, round 5)
int g_w4 = 32;             // read_tuning_set("conv_w4", min Cin): layers with at least this many channels take the Winograd F(4x4,3x3)
int g_w4h = 32;            // read_tuning_set("conv_w4h", min Cin): F(4x4) layers with Cin % 32 == 0 and at least this many channels take the split-operand
                           // kernel on the f16 matrix cores when its operand was supplied (0 = never: the fp32 kernel)
int g_w4h_waves = 4;       // debug library only: read_tuning_set("conv_w4h_waves", 8) = the split-operand kernel with specialised waves (measured slower, round 6)
int g_d3h = 0;             // read_tuning_set("conv_d3h", min Cin): gated 3x3 / stride-1 layers with whole 32-channel chunks and at least this many channels take the
                           // DIRECT split-operand kernel (f16 matrix cores, all nine taps) when its operand was supplied.  Default 0 (never): on plain
                           // launches it measured 5 - 15 % slower than the Winograd split-operand kernel (profiles/r6_d3h_ab.md) ...
int g_d3h_s2 = 32;         // read_tuning_set("conv_d3h_s2", min Cin): 3x3 / STRIDE-2 layers on the direct split-operand kernel (0 = never: the fp32 direct kernels)
int g_d3h_fam = 32;        // read_tuning_set("conv_d3h_fam", min Cin): ... and 11 us FASTER per launch than the fp32 kernel on FAM's x1 * x2 launches, which the
                           // Winograd split-operand kernel does not take: those run on it (0 = never)
int g_pxh = 16;            // read_tuning_set("conv_pxh", min Cin): 1x1 / stride-1 layers with Cin % 16 == 0, Cin <= 256 and at least this many input channels take the
                           // split-operand pixel-lane kernel (f16 matrix cores) when their operand (wpacked_d3h of a 1x1 layer) was supplied (0 = never)
int g_t3h = 8;             // read_tuning_set("conv_t3h", max Cin): 3x3 / stride-1 layers over one source of at most this many channels (8, 16 or 32) take the
                           // split-operand implicit-GEMM form of that kernel when wpacked_t3h was supplied (0 = never)
int g_sc = 8;              // read_tuning_set("conv_sc", 0): the output layer (Cout <= 4) back on the F(2x2) MFMA kernel instead of the vector pipe; other values: conv_set_sc
                           // kernel when its weights were supplied (0 = never)
int g_abl = 0;             // read_tuning_set("conv_abl", bits): attribution probes of the 16x16x4 Winograd kernels (results invalid); -DREAD_DEBUG_KNOBS builds only
int g_w16 = 0;             // read_tuning_set("conv_w16", v): F(2x2,3x3) launches: 0 the row-per-wave kernel (default: measured equal or faster),
                           // 1 the wave-autonomous kernel with the shared input transform
int g_stagger_ticks = 0;   // read_tuning_set("conv_stagger", ticks of 10 ns)
int g_ablate = 0;          // read_tuning_set("conv_ablate", bits): attribution probe, results invalid; -DREAD_DEBUG_KNOBS builds only

int find_wave_config(int ks, int s, int kc, int P, int QG)
{
    int best = -1;
    for (int i = 0; i < N_CONFIGS; ++i) {
        const ConvConfig &c = g_configs[i];
        if (c.wave && c.KS == ks && c.S == s && c.KC == kc && c.P == P && c.QG == QG) best = i;   // last = deepest prefetch
    }
    return best;
}

// Automatic choice = the measured-best entry per layer family on MI355X at 1216x352
// (profiles/README.md, sweep r1k).  read_tuning_set("conv_wave", 0) restricts it to the workgroup-tiled
// kernels (A/B runs).
int pick_config(int ks, int s, int kc, int groups, int outH, int outW)
{
    // round 6 (tools/sweep_small.py): the sweep behind the table below ran at 1216 x 352; on the SCM chains' quarter- and eighth-
    // resolution images two entries start too few workgroups to fill the chip
    const long pixels = (long)outH * outW;
    int c = -1;
    if (ks == 3 && s == 1 && kc == 16) {
        if (groups % 4 == 0 && groups % 8 != 0) c = find_config(3, 1, 16, 2, 2, 2, 2, 1, 2);     // 128 ch: 108.8 TF
        else if (g_prefer_wave) c = find_wave_config(3, 1, 16, 1, 1);                               // 104-117 TF
        else if (groups % 8 == 0) c = find_config(3, 1, 16, 2, 1, 2, 2, 2, 1);
        else c = find_config(3, 1, 16, 1, 1, 4, 1, 2, 1);
    } else if (ks == 3 && s == 1 && kc == 8) {
        // small images: 4 x 32-pixel workgroup tiles (SCM1/0.main.0 at 88 x 304 and 44 x 152: 12.1 -> 9.5, 12.0 -> 9.1 us)
        c = (g_prefer_wave && pixels > 30000) ? find_wave_config(3, 1, 8, 2, 1) : find_config(3, 1, 8, 1, 1, 4, 1, 2, 2);
    } else if (ks == 1 && s == 1 && kc == 32) {
        // four groups per workgroup only where that still leaves >= 128 workgroups (SCM0.main.1, 64 -> 128 at 44 x 152: 55
        // workgroups took 20.0 us, one group per wave-autonomous unit 9.0)
        const long wgs4 = (long)ceil_div(outW, 32) * ceil_div(outH, 4) * (groups / 4);
        if (groups % 4 == 0 && wgs4 >= 128) c = find_config(1, 1, 32, 2, 2, 2, 2, 1, 2);      // 91 TF at 128 channels
        else c = find_wave_config(1, 1, 32, 1, 1);                            // 99-103 TF (16-channel chunks: 93-97)
    } else if (ks == 1 && s == 1 && kc == 16) {
        if (groups % 4 == 0) c = find_config(1, 1, 16, 2, 2, 2, 2, 1, 2);
        else if (g_prefer_wave) c = find_wave_config(1, 1, 16, groups == 1 ? 2 : 1, 1);
        else c = groups == 1 ? find_config(1, 1, 16, 2, 1, 4, 1, 1, 2) : find_config(1, 1, 16, 1, 1, 4, 1, 1, 2);
    } else if (ks == 3 && s == 2) {
        c = groups % 4 == 0 ? find_config(3, 2, 16, 1, 2, 2, 2, 1, 2) : find_config(3, 2, 16, 1, 2, 4, 1, 1, 2);
    } else if (ks == 4 && s == 2) {
        c = groups == 1 ? find_config(4, 2, 16, 1, 1, 4, 1, 1, 2) : find_config(4, 2, 16, 1, 1, 1, 1, 1, 2);
    }
    if (c >= 0 && groups % (g_configs[c].WN * g_configs[c].QG) == 0) return c;
    for (int i = 0; i < N_CONFIGS; ++i) {
        const ConvConfig &k = g_configs[i];
        if (k.KS == ks && k.S == s && k.KC == kc && groups % (k.WN * k.QG) == 0) return i;
    }
    return -1;
}

int pad32(int c) { return (c + 31) / 32 * 32; }

// ------------------------------------------------------------------------------------------
// bilinear x4 upsample, align_corners=False (nn.Upsample, unet.py:200)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bilinear_up4_kernel(const float *__restrict__ in, int inH, int inW, int C,
                                                           float *__restrict__ out)
{
    const int outH = inH * 4, outW = inW * 4, q4 = C >> 2;
    const long long total = (long long)outH * outW * q4;
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(item % q4);
        const long long pix = item / q4;
        const int ox = (int)(pix % outW), oy = (int)(pix / outW);
        // area_pixel_compute_source_index: src = 0.25*(dst+0.5)-0.5, clamped at 0
        float sy = 0.25f * ((float)oy + 0.5f) - 0.5f;
        float sx = 0.25f * ((float)ox + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < inH - 1 ? 1 : 0), x1 = x0 + (x0 < inW - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float4 v00 = *reinterpret_cast<const float4 *>(in + ((long long)y0 * inW + x0) * C + 4 * q);
        const float4 v01 = *reinterpret_cast<const float4 *>(in + ((long long)y0 * inW + x1) * C + 4 * q);
        const float4 v10 = *reinterpret_cast<const float4 *>(in + ((long long)y1 * inW + x0) * C + 4 * q);
        const float4 v11 = *reinterpret_cast<const float4 *>(in + ((long long)y1 * inW + x1) * C + 4 * q);
        float4 o;
        o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        *reinterpret_cast<float4 *>(out + pix * C + 4 * q) = o;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int read_conv_config_count(void) { return N_CONFIGS; }
extern "C" const char *read_conv_config_name(int config)
{
    return (config >= 0 && config < N_CONFIGS) ? g_configs[config].name : "";
}

extern "C" size_t read_conv_packed_floats(int Cin, int Cout, int ksize)
{
    if (Cin < 8 || Cin % 8 || Cout < 1 || (ksize != 1 && ksize != 3 && ksize != 4)) return 0;
    return (size_t)Cin * ksize * ksize * 2 * pad32(Cout);
}
extern "C" size_t read_conv_param_floats(int Cout) { return Cout < 1 ? 0 : (size_t)4 * pad32(Cout); }

extern "C" int read_conv_pack_weights_host(int Cin, int Cout, int ksize, int kc, const float *wf, const float *wm,
                                           float *wpacked_host)
{
    READ_CHECK_ARG(wf && wm && wpacked_host, "read_conv_pack_weights_host: null pointer");
    READ_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 4, "read_conv_pack_weights_host: ksize must be 1, 3 or 4");
    // (for 1x1 layers the fragment order does not depend on kc: k8 steps are simply consecutive, so 16- and 32-channel
    //  chunk kernels read the same blob)
    READ_CHECK_ARG((kc == 8 || kc == 16 || (kc == 32 && ksize == 1)) && Cin >= kc && Cin % kc == 0,
                   "read_conv_pack_weights_host: Cin=%d is not a multiple of kc=%d", Cin, kc);
    READ_CHECK_ARG(Cout >= 1, "read_conv_pack_weights_host: Cout < 1");
    const int CoutPad = pad32(Cout), NT = CoutPad / 16, KK = kc / 8, taps = ksize * ksize;
    const int nchunks = Cin / kc;
    size_t o = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk)
        for (int tap = 0; tap < taps; ++tap)
            for (int kk = 0; kk < KK; ++kk)
                for (int nt = 0; nt < NT; ++nt) {
                    const float *w = (nt & 1) ? wm : wf;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j, ++o) {
                            const int cout = (nt >> 1) * 32 + (lane & 31);
                            const int cin = chunk * kc + kk * 8 + 4 * (lane >> 5) + j;
                            wpacked_host[o] = cout < Cout ? w[((size_t)cout * Cin + cin) * taps + tap] : 0.0f;
                        }
                }
    return READ_OK;
}

extern "C" size_t read_conv_wino_floats(int Cin, int Cout)
{
    if (Cin < 16 || Cin % 16 || Cout < 1) return 0;
    return (size_t)Cin * 16 * 2 * pad32(Cout);
}

// U = G g G^T per (cout, cin) pair, packed [group][k8 step][row i][j][f|m][lane][4] (tests/wino_ref.py).
extern "C" int read_conv_pack_wino_host(int Cin, int Cout, const float *wf, const float *wm, float *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_wino_host: null pointer");
    READ_CHECK_ARG(Cin >= 16 && Cin % 16 == 0 && Cout >= 1, "read_conv_pack_wino_host: needs Cin %% 16 == 0 (got %d)", Cin);
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    const int CoutPad = pad32(Cout), groups = CoutPad / 32, nsteps = Cin / 8;
    size_t o = 0;
    for (int g = 0; g < groups; ++g)
        for (int s = 0; s < nsteps; ++s)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    for (int fm = 0; fm < 2; ++fm) {
                        const float *w = fm ? wm : wf;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e, ++o) {
                                const int co = g * 32 + (lane & 31), ci = 8 * s + 4 * (lane >> 5) + e;
                                float u = 0.0f;
                                if (co < Cout) {
                                    const float *k = w + ((size_t)co * Cin + ci) * 9;
                                    for (int a = 0; a < 3; ++a)
                                        for (int b = 0; b < 3; ++b) u += G[i][a] * k[a * 3 + b] * G[j][b];
                                }
                                out[o] = i == 2 ? -u : u;      // the kernel builds row 2 of B^T d negated
                            }
                    }
    return READ_OK;
}

// The wave-autonomous Winograd kernel's order (tests/wino16_ref.py): [group][wave 4][chunk of 16 cin][a][j][lane][e]; lane
// (i = lane & 15, kl = lane >> 4) holds U_{i < 8 ? f : m}[a][j][cin = 16 chunk + 4 kl + e][cout = 32 group + 8 wave + (i & 7)].
extern "C" int read_conv_pack_w16_host(int Cin, int Cout, const float *wf, const float *wm, float *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_w16_host: null pointer");
    READ_CHECK_ARG(Cin >= 16 && Cin % 16 == 0 && Cout >= 1, "read_conv_pack_w16_host: needs Cin %% 16 == 0 (got %d)", Cin);
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    const int CoutPad = pad32(Cout), groups = CoutPad / 32, nchunks = Cin / 16;
    size_t o = 0;
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < 4; ++w)
            for (int c = 0; c < nchunks; ++c)
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e, ++o) {
                                const int slot = lane & 15, co = g * 32 + w * 8 + (slot & 7), ci = 16 * c + 4 * (lane >> 4) + e;
                                float u = 0.0f;
                                if (co < Cout) {
                                    const float *k = ((slot >> 3) ? wm : wf) + ((size_t)co * Cin + ci) * 9;
                                    for (int a = 0; a < 3; ++a)
                                        for (int b = 0; b < 3; ++b) u += G[i][a] * k[a * 3 + b] * G[j][b];
                                }
                                out[o] = u;
                            }
    return READ_OK;
}

extern "C" size_t read_conv_w4_floats(int Cin, int Cout)
{
    if (Cin < 16 || Cin % 16 || Cout < 1) return 0;
    return (size_t)Cin * 36 * 2 * pad32(Cout);
}

// F(4x4,3x3): U = G g G^T (6 x 6) per (cout, cin) pair, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1],
// evaluated in double and rounded once; order [group][wave 4][chunk of 16 cin][frequency 6 xi + nu][lane][e] with lane
// (i = lane & 15, kl = lane >> 4) = U_{i < 8 ? f : m}[xi][nu][cin = 16 chunk + 4 kl + e][cout = 32 group + 8 wave + (i & 7)]   (tests/wino4_ref.py)
extern "C" int read_conv_pack_w4_host(int Cin, int Cout, const float *wf, const float *wm, float *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_w4_host: null pointer");
    READ_CHECK_ARG(Cin >= 16 && Cin % 16 == 0 && Cout >= 1, "read_conv_pack_w4_host: needs Cin %% 16 == 0 (got %d)", Cin);
    static const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int CoutPad = pad32(Cout), groups = CoutPad / 32, nchunks = Cin / 16;
    size_t o = 0;
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < 4; ++w)
            for (int c = 0; c < nchunks; ++c)
                for (int fq = 0; fq < 36; ++fq)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e, ++o) {
                            const int slot = lane & 15, co = g * 32 + w * 8 + (slot & 7), ci = 16 * c + 4 * (lane >> 4) + e;
                            double u = 0.0;
                            if (co < Cout) {
                                const float *k = ((slot >> 3) ? wm : wf) + ((size_t)co * Cin + ci) * 9;
                                for (int a = 0; a < 3; ++a)
                                    for (int b = 0; b < 3; ++b) u += G[fq / 6][a] * (double)k[a * 3 + b] * G[fq % 6][b];
                            }
                            out[o] = (float)u;
                        }
    return READ_OK;
}

// F(4x4,3x3) operand of the split-operand kernel (gated_conv_wino4h_kernel): U = G g G^T in double, scaled per output ROW by the
// power of two s that puts max |U s| of the row in [2^14, 2^15), cut into Uh = f16(U s) and Ul = f16(U s - Uh) (round to nearest
// even, both); order [group][wave 4][chunk of 32 cin][frequency 36][piece Uh | Ul][lane][8 halfs], lane (i = lane & 15, kq = lane >> 4)
// = rows as in read_conv_pack_w4_host, cin = 32 chunk + 8 kq + e; then 2 * CoutPad floats 1 / s ([conv_f rows | conv_m rows]).
// Sized in floats like every block of a packed blob: Cin * 36 * 2 * CoutPad (the halfs) + 2 * CoutPad.   (tests/wino4h_ref.py)
extern "C" size_t read_conv_w4h_floats(int Cin, int Cout)
{
    if (Cin < 32 || Cin % 32 || Cout < 1) return 0;
    return (size_t)Cin * 36 * 2 * pad32(Cout) + 2 * (size_t)pad32(Cout);
}

namespace {
// double -> IEEE binary16, round to nearest even, subnormals kept (the values are far inside the range: |x| < 2^15)
unsigned short f16_bits_rtn(double x)
{
    const unsigned short sign = std::signbit(x) ? 0x8000u : 0u;
    double ax = std::fabs(x);
    if (ax == 0.0) return sign;
    if (ax >= 65520.0) return (unsigned short)(sign | 0x7c00u);
    int e;
    (void)std::frexp(ax, &e);                                  // ax = f 2^e, f in [0.5, 1)
    int ex = e - 1;                                            // ax = 1.m x 2^ex
    if (ex < -14) ex = -14;                                    // subnormal: fixed quantum 2^-24
    const double q = std::ldexp(1.0, ex - 10);                 // quantum of the 11-bit significand
    const double r = std::nearbyint(ax / q);                   // ties to even (default rounding mode)
    const double v = r * q;                                    // may have carried into the next binade: recompute the fields
    if (v >= 65520.0) return (unsigned short)(sign | 0x7c00u);
    if (v < std::ldexp(1.0, -14)) return (unsigned short)(sign | (unsigned short)std::lrint(v / std::ldexp(1.0, -24)));
    int e2;
    (void)std::frexp(v, &e2);
    const int ex2 = e2 - 1;
    const unsigned mant = (unsigned)std::lrint(v / std::ldexp(1.0, ex2 - 10)) - 1024u;
    return (unsigned short)(sign | (unsigned)((ex2 + 15) << 10) | mant);
}
double f16_value(unsigned short h)
{
    const int e = (h >> 10) & 31, m = h & 1023;
    const double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(1024 + m), e - 25);
    return (h & 0x8000u) ? -v : v;
}
}  // namespace

extern "C" int read_conv_pack_w4h_host(int Cin, int Cout, const float *wf, const float *wm, void *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_w4h_host: null pointer");
    READ_CHECK_ARG(Cin >= 32 && Cin % 32 == 0 && Cout >= 1, "read_conv_pack_w4h_host: needs Cin %% 32 == 0 (got %d)", Cin);
    static const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int CoutPad = pad32(Cout), nchunks = Cin / 32;
    unsigned short *h = static_cast<unsigned short *>(out);
    float *inv = reinterpret_cast<float *>(out) + (size_t)Cin * 36 * 2 * CoutPad;
    std::vector<double> U((size_t)36 * Cin);
    for (int row = 0; row < 2 * CoutPad; ++row) {               // row = [f | m] x padded output channel
        const int fm = row / CoutPad, co = row % CoutPad;
        double mx = 0.0;
        if (co < Cout)
            for (int ci = 0; ci < Cin; ++ci) {
                const float *k = (fm ? wm : wf) + ((size_t)co * Cin + ci) * 9;
                for (int fq = 0; fq < 36; ++fq) {
                    double u = 0.0;
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) u += G[fq / 6][a] * (double)k[a * 3 + b] * G[fq % 6][b];
                    U[(size_t)fq * Cin + ci] = u;
                    mx = std::fmax(mx, std::fabs(u));
                }
            }
        else
            std::fill(U.begin(), U.end(), 0.0);
        int ex = 0;                                            // s = 2^ex: max |U| s in [2^14, 2^15)
        if (mx > 0.0 && std::isfinite(mx)) {
            int e;
            (void)std::frexp(mx, &e);                          // mx in [2^(e-1), 2^e)
            ex = 15 - e;
            if (ex > 60) ex = 60;                              // a row of denormal weights: keep 1 / s an fp32 normal
            if (ex < -60) ex = -60;
        }
        inv[row] = (float)std::ldexp(1.0, -ex);
        const int g = co / 32, w = (co % 32) / 8, i = (co % 8) + 8 * fm;
        for (int c = 0; c < nchunks; ++c)
            for (int fq = 0; fq < 36; ++fq)
                for (int kq = 0; kq < 4; ++kq)
                    for (int e = 0; e < 8; ++e) {
                        const double us = std::ldexp(U[(size_t)fq * Cin + 32 * c + 8 * kq + e], ex);
                        const unsigned short hi = f16_bits_rtn(us), lo = f16_bits_rtn(us - f16_value(hi));
                        const size_t frag = ((((size_t)(g * 4 + w) * nchunks + c) * 36 + fq) * 2) * 512;   // halfs; 512 per piece
                        const int lane = i + 16 * kq;
                        h[frag + (size_t)lane * 8 + e] = hi;
                        h[frag + 512 + (size_t)lane * 8 + e] = lo;
                    }
    }
    return READ_OK;
}

// Operand of the direct split-operand kernel (gated_conv_d3h_kernel): the 3x3 weights themselves, scaled per output ROW by the power
// of two s that puts the row's largest |w s| in [2^14, 2^15), cut into wh = f16(w s) and wl = f16(w s - wh); order
// [group][row half rh 2][chunk of 32 cin][tap 9 = 3 ky + kx][row block rb 2][piece wh | wl][lane][8 halfs], lane (i = lane & 15,
// kq = lane >> 4) = row i of the block (i < 8: conv_f of channel 32 g + 16 rh + 8 rb + i, else conv_m of channel ... + i - 8), cin = 32
// chunk + 8 kq + e; then 2 * CoutPad floats 1 / s ([conv_f rows | conv_m rows]).   (tests/d3h_ref.py)
extern "C" size_t read_conv_dkh_floats(int Cin, int Cout, int ksize)
{
    if (ksize == 1) return (Cin < 16 || Cin % 16 || Cout < 1) ? 0 : (size_t)Cin * 2 * pad32(Cout) + 2 * (size_t)pad32(Cout);
    if (Cin < 32 || Cin % 32 || Cout < 1 || (ksize != 3 && ksize != 4)) return 0;
    return (size_t)Cin * 2 * ksize * ksize * pad32(Cout) + 2 * (size_t)pad32(Cout);
}
extern "C" size_t read_conv_d3h_floats(int Cin, int Cout) { return read_conv_dkh_floats(Cin, Cout, 3); }
extern "C" int read_conv_pack_dkh_host(int Cin, int Cout, int ksize, const float *wf, const float *wm, void *out);
extern "C" int read_conv_pack_d3h_host(int Cin, int Cout, const float *wf, const float *wm, void *out)
{
    return read_conv_pack_dkh_host(Cin, Cout, 3, wf, wm, out);
}

// ... the same for a k x k kernel, k = 3 or 4 (tap = k ky + kx): the 4 x 4 / stride-2 layers run on the stride-2 kernel too
extern "C" int read_conv_pack_dkh_host(int Cin, int Cout, int ksize, const float *wf, const float *wm, void *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_dkh_host: null pointer");
    if (ksize == 1) {
        // 1x1 layers (gated_conv_pxh_kernel, v_mfma_f32_32x32x16_f16 with the weights as the A operand): the same rows, scales and
        // pieces in the order [k16 step][tile t = 2 group + (f | m)][wh | wl][lane][8 halfs], lane (i = lane & 31, h = lane >> 5) =
        // row i of the tile (conv_f / conv_m of channel 32 group + i), cin = 16 step + 8 h + e; then the 2 * CoutPad floats 1 / s.
        READ_CHECK_ARG(Cin >= 16 && Cin % 16 == 0 && Cout >= 1, "read_conv_pack_dkh_host: a 1x1 layer needs Cin %% 16 == 0 (got %d)", Cin);
        const int CoutPad = pad32(Cout), nsteps = Cin / 16, NTL = CoutPad / 16;
        unsigned short *h = static_cast<unsigned short *>(out);
        float *inv = reinterpret_cast<float *>(out) + (size_t)Cin * 2 * CoutPad;
        for (int row = 0; row < 2 * CoutPad; ++row) {
            const int fm = row / CoutPad, co = row % CoutPad;
            const float *k = co < Cout ? (fm ? wm : wf) + (size_t)co * Cin : nullptr;
            double mx = 0.0;
            if (k)
                for (int i = 0; i < Cin; ++i) mx = std::fmax(mx, std::fabs((double)k[i]));
            int ex = 0;
            if (mx > 0.0 && std::isfinite(mx)) {
                int e;
                (void)std::frexp(mx, &e);
                ex = 15 - e;
                if (ex > 60) ex = 60;
                if (ex < -60) ex = -60;
            }
            inv[row] = (float)std::ldexp(1.0, -ex);
            const int t = 2 * (co / 32) + fm, i = co % 32;
            for (int st = 0; st < nsteps; ++st)
                for (int hh = 0; hh < 2; ++hh)
                    for (int e = 0; e < 8; ++e) {
                        const double ws = k ? std::ldexp((double)k[16 * st + 8 * hh + e], ex) : 0.0;
                        const unsigned short hi = f16_bits_rtn(ws), lo = f16_bits_rtn(ws - f16_value(hi));
                        const size_t frag = (((size_t)st * NTL + t) * 2) * 512;                   // halfs; 512 per piece
                        const int lane = i + 32 * hh;
                        h[frag + (size_t)lane * 8 + e] = hi;
                        h[frag + 512 + (size_t)lane * 8 + e] = lo;
                    }
        }
        return READ_OK;
    }
    READ_CHECK_ARG(ksize == 3 || ksize == 4, "read_conv_pack_dkh_host: ksize must be 1, 3 or 4");
    READ_CHECK_ARG(Cin >= 32 && Cin % 32 == 0 && Cout >= 1, "read_conv_pack_dkh_host: needs Cin %% 32 == 0 (got %d)", Cin);
    const int CoutPad = pad32(Cout), nchunks = Cin / 32, NT = ksize * ksize;
    unsigned short *h = static_cast<unsigned short *>(out);
    float *inv = reinterpret_cast<float *>(out) + (size_t)Cin * 2 * NT * CoutPad;
    for (int row = 0; row < 2 * CoutPad; ++row) {               // row = [f | m] x padded output channel
        const int fm = row / CoutPad, co = row % CoutPad;
        const float *k = co < Cout ? (fm ? wm : wf) + (size_t)co * Cin * NT : nullptr;
        double mx = 0.0;
        if (k)
            for (size_t i = 0; i < (size_t)Cin * NT; ++i) mx = std::fmax(mx, std::fabs((double)k[i]));
        int ex = 0;
        if (mx > 0.0 && std::isfinite(mx)) {
            int e;
            (void)std::frexp(mx, &e);
            ex = 15 - e;
            if (ex > 60) ex = 60;
            if (ex < -60) ex = -60;
        }
        inv[row] = (float)std::ldexp(1.0, -ex);
        const int g = co / 32, rh = (co % 32) / 16, rb = (co % 16) / 8, i = (co % 8) + 8 * fm;
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < NT; ++tap)
                for (int kq = 0; kq < 4; ++kq)
                    for (int e = 0; e < 8; ++e) {
                        const double ws = k ? std::ldexp((double)k[(size_t)(32 * c + 8 * kq + e) * NT + tap], ex) : 0.0;
                        const unsigned short hi = f16_bits_rtn(ws), lo = f16_bits_rtn(ws - f16_value(hi));
                        const size_t frag = (((((size_t)(g * 2 + rh) * nchunks + c) * NT + tap) * 2 + rb) * 2) * 512;    // halfs; 512 per piece
                        const int lane = i + 16 * kq;
                        h[frag + (size_t)lane * 8 + e] = hi;
                        h[frag + 512 + (size_t)lane * 8 + e] = lo;
                    }
    }
    return READ_OK;
}

// 3x3 weights of a layer with 8, 16 or 32 input channels as the implicit-GEMM operand of the split-operand pixel-lane kernel: the matrix
// W'[cout][k = tap Cin + ci] (tap = 3 ky + kx), k padded with zeros to whole k16 steps, in the 1x1 operand's order (read_conv_pack_dkh_host, ksize 1)
extern "C" size_t read_conv_t3h_floats(int Cin, int Cout)
{
    return (Cin == 8 || Cin == 16 || Cin == 32) && Cout >= 1 ? read_conv_dkh_floats((9 * Cin + 15) / 16 * 16, Cout, 1) : 0;
}

extern "C" int read_conv_pack_t3h_host(int Cin, int Cout, const float *wf, const float *wm, void *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_t3h_host: null pointer");
    READ_CHECK_ARG(read_conv_t3h_floats(Cin, Cout) > 0, "read_conv_pack_t3h_host: needs Cin = 8, 16 or 32 (got %d)", Cin);
    const int K = (9 * Cin + 15) / 16 * 16;
    std::vector<float> f((size_t)Cout * K, 0.0f), m((size_t)Cout * K, 0.0f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                f[(size_t)co * K + tap * Cin + ci] = wf[((size_t)co * Cin + ci) * 9 + tap];
                m[(size_t)co * K + tap * Cin + ci] = wm[((size_t)co * Cin + ci) * 9 + tap];
            }
    return read_conv_pack_dkh_host(K, Cout, 1, f.data(), m.data(), out);
}

// Small-Cout order (gated_conv_smallc_kernel): [tap][cin][f0 f1 f2 f3 | m0 m1 m2 m3], channels >= Cout zero.
extern "C" size_t read_conv_sc_floats(int Cin, int Cout)
{
    return (Cin == 32 && Cout >= 1 && Cout <= 4) ? (size_t)9 * Cin * 8 : 0;      // the kernel is instantiated for 32 input channels
}

extern "C" int read_conv_pack_sc_host(int Cin, int Cout, const float *wf, const float *wm, float *out)
{
    READ_CHECK_ARG(wf && wm && out, "read_conv_pack_sc_host: null pointer");
    READ_CHECK_ARG(read_conv_sc_floats(Cin, Cout) > 0, "read_conv_pack_sc_host: needs Cin == 32 and 1 <= Cout <= 4 (got %d, %d)", Cin, Cout);
    for (int tap = 0; tap < 9; ++tap)
        for (int ci = 0; ci < Cin; ++ci)
            for (int j = 0; j < 8; ++j) {
                const int co = j & 3;
                out[((size_t)tap * Cin + ci) * 8 + j] = co < Cout ? ((j >> 2) ? wm : wf)[((size_t)co * Cin + ci) * 9 + tap] : 0.0f;
            }
    return READ_OK;
}

extern "C" int read_conv_pack_params_host(int Cout, const float *bf, const float *bm, const float *gamma,
                                          const float *beta, const float *mean, const float *var, float eps,
                                          float *params_host)
{
    READ_CHECK_ARG(bf && bm && gamma && beta && mean && var && params_host, "read_conv_pack_params_host: null pointer");
    READ_CHECK_ARG(Cout >= 1, "read_conv_pack_params_host: Cout < 1");
    const int CoutPad = pad32(Cout);
    for (int c = 0; c < CoutPad; ++c) {
        const bool ok = c < Cout;
        // eval-mode BatchNorm folded to y = x*scale + shift (applied AFTER the gate product)
        const float scale = ok ? gamma[c] / sqrtf(var[c] + eps) : 0.0f;
        params_host[c] = ok ? bf[c] : 0.0f;
        params_host[CoutPad + c] = ok ? bm[c] : 0.0f;
        params_host[2 * CoutPad + c] = scale;
        params_host[3 * CoutPad + c] = ok ? beta[c] - mean[c] * scale : 0.0f;
    }
    return READ_OK;
}

namespace readhip {

void conv_set_prefer_wave(int v) { g_prefer_wave = v; }
void conv_set_stagger(int ticks) { g_stagger_ticks = ticks < 0 ? 0 : ticks; }
void conv_set_ablate(int bits) { g_ablate = bits; }
void conv_set_wino(int max_cin) { g_use_wino = max_cin; }
void conv_set_kc32(int v) { g_kc32 = v; }
void conv_set_w16(int v) { g_w16 = v != 0; }
void conv_set_abl(int v) { g_abl = v; }
void conv_set_w4(int v) { g_w4 = v < 0 ? 0 : v; }
void conv_set_w4h(int v) { g_w4h = v < 0 ? 0 : v; }
void conv_set_d3h(int v) { g_d3h = v < 0 ? 0 : v; }
void conv_set_d3h_fam(int v) { g_d3h_fam = v < 0 ? 0 : v; }
void conv_set_d3h_s2(int v) { g_d3h_s2 = v < 0 ? 0 : v; }
void conv_set_pxh(int v) { g_pxh = v < 0 ? 0 : v; }
void conv_set_t3h(int v) { g_t3h = v < 0 ? 0 : v; }
void conv_set_w4h_waves(int v) { g_w4h_waves = v == 4 ? 4 : 8; }
void conv_set_w4_grid(int v) { g_w4_grid = v != 0; }
void conv_set_wino_wgs(int v) { g_wino_wgs = v <= 1 ? 1 : 2; }
void conv_set_px(int v) { g_conv_px = v < 0 ? 0 : v > 4 ? 4 : v; }
void conv_set_w4x2(int v) { g_w4x2 = v != 0; }
void conv_set_sc(int v) { g_sc = v; }           // 0 off; 8 / 16 / 32 = input channels per LDS phase
int conv_get(const char *key, int *value)
{
    if (!strcmp(key, "conv_wave")) *value = g_prefer_wave;
    else if (!strcmp(key, "conv_stagger")) *value = g_stagger_ticks;
    else if (!strcmp(key, "conv_kc32")) *value = g_kc32;
    else if (!strcmp(key, "conv_px")) *value = g_conv_px;
    else if (!strcmp(key, "conv_sc")) *value = g_sc;
#ifdef READ_DEBUG_KNOBS
    else if (!strcmp(key, "conv_w4x2")) *value = g_w4x2;
#endif
    else if (!strcmp(key, "conv_wino_wgs")) *value = g_wino_wgs;
    else if (!strcmp(key, "conv_wino")) *value = g_use_wino;
    else if (!strcmp(key, "conv_w16")) *value = g_w16;
    else if (!strcmp(key, "conv_w4")) *value = g_w4;
    else if (!strcmp(key, "conv_w4h")) *value = g_w4h;
    else if (!strcmp(key, "conv_d3h")) *value = g_d3h;
    else if (!strcmp(key, "conv_d3h_fam")) *value = g_d3h_fam;
    else if (!strcmp(key, "conv_d3h_s2")) *value = g_d3h_s2;
    else if (!strcmp(key, "conv_pxh")) *value = g_pxh;
    else if (!strcmp(key, "conv_t3h")) *value = g_t3h;
#ifdef READ_DEBUG_KNOBS
    else if (!strcmp(key, "conv_w4h_waves")) *value = g_w4h_waves;
#endif
    else if (!strcmp(key, "conv_w4_grid")) *value = g_w4_grid;
#ifdef READ_DEBUG_KNOBS
    else if (!strcmp(key, "conv_ablate")) *value = g_ablate;
    else if (!strcmp(key, "conv_abl")) *value = g_abl;
#endif
    else return 0;
    return 1;
}

static unsigned long long *g_trace = nullptr;
static size_t g_trace_records = 0;
void conv_set_trace(void *buf, size_t bytes)
{
    g_trace = (unsigned long long *)buf;
    g_trace_records = buf ? bytes / 64 : 0;
}

// Validates a descriptor, builds kernel arguments and launches.  Shared by the single-layer
// entry point and the UNet executor.
int conv_uses_wino(const read_conv_desc *d);
int conv_uses_w4(const read_conv_desc *d);
int conv_uses_w4h(const read_conv_desc *d);
int conv_uses_d3h(const read_conv_desc *d);
int conv_uses_d3h_s2(const read_conv_desc *d);
int conv_uses_sc(const read_conv_desc *d);
int conv_uses_pxh(const read_conv_desc *d);
int conv_uses_t3h(const read_conv_desc *d);

int launch_gated_conv(const read_conv_desc *d, hipStream_t stream)
{
    READ_CHECK_ARG(d, "read_gated_conv_forward: null descriptor");
    READ_CHECK_ARG(d->n_src >= 1 && d->n_src <= READ_CONV_MAX_SRC, "read_gated_conv_forward: n_src must be 1..%d",
                   READ_CONV_MAX_SRC);
    READ_CHECK_ARG(d->ksize == 1 || d->ksize == 3 || d->ksize == 4, "read_gated_conv_forward: ksize must be 1,3,4");
    READ_CHECK_ARG(d->stride == 1 || d->stride == 2, "read_gated_conv_forward: stride must be 1 or 2");
    READ_CHECK_ARG(d->inH >= 1 && d->inW >= 1 && d->Cout >= 1, "read_gated_conv_forward: bad sizes");
    READ_CHECK_ARG((d->wpacked || d->wpacked_w4 || d->wpacked_w4h || d->wpacked_d3h || d->wpacked_wino || d->wpacked_t3h) && d->params && d->out, "read_gated_conv_forward: null weights/params/out");
    READ_CHECK_ARG(d->out_cstride >= (d->linear ? 2 : 1) * d->Cout, "read_gated_conv_forward: out_cstride too small");
    READ_CHECK_ARG(!d->linear || (!d->residual && !d->fill_pad), "read_gated_conv_forward: linear mode takes no residual / fill");
    READ_CHECK_ARG(!d->mul || d->n_src == 1, "read_gated_conv_forward: mul needs a single source");
    READ_CHECK_ARG((uintptr_t)d->wpacked % 16 == 0, "read_gated_conv_forward: packed weights misaligned");
    {
        // One fragment order per layer is enough for a host that asked read_conv_kernel_family first; a host that aliases
        // wpacked to its Winograd fragments (training: read_amd/train.py packs ONE order per layer and step) or leaves it NULL
        // (the lean UNet blob) must never reach a kernel that reads wpacked as the direct order — a tuning knob changed on a
        // live engine, a 2 GiB tensor or an odd out_cstride can decline the Winograd kernels after the host has packed for them.
        // Checked HERE, for both entry points (read_gated_conv_forward and the UNet executor's direct call).
        const int family = conv_uses_t3h(d) ? 8 : conv_uses_sc(d) ? 1 : conv_uses_pxh(d) ? 7 : (conv_uses_d3h(d) || conv_uses_d3h_s2(d)) ? 6 : conv_uses_w4h(d) ? 5 : conv_uses_w4(d) ? 4 : conv_uses_wino(d) ? 2 : 0;
        const bool cfg_wino = d->config >= 0 && d->config < N_CONFIGS && g_configs[d->config].wino;   // forced F(2x2) configs read wpacked_wino
        const bool w16_forced = d->config == -3;
        READ_CHECK_ARG(d->wpacked || family != 0 || cfg_wino || w16_forced,
                       "read_gated_conv_forward: this launch takes a direct kernel and wpacked is NULL (fragment order not packed)");
        READ_CHECK_ARG(!d->wpacked || (!((const void *)d->wpacked == (const void *)d->wpacked_w4 && family != 4 && family != 5 && family != 6 && family != 7 && family != 8) &&
                                       !((const void *)d->wpacked == (const void *)d->wpacked_wino && family != 2 && !cfg_wino)),
                       "read_gated_conv_forward: wpacked aliases Winograd fragments but the launch takes kernel family %d "
                       "(ask read_conv_kernel_family before packing)", family);
        READ_CHECK_ARG(family != 4 || d->wpacked_w4, "read_gated_conv_forward: F(4x4) launch without wpacked_w4");
        READ_CHECK_ARG(family != 2 || d->wpacked_wino, "read_gated_conv_forward: F(2x2) launch without wpacked_wino");
        READ_CHECK_ARG(family != 1 || d->wpacked_sc, "read_gated_conv_forward: small-Cout launch without wpacked_sc");
    }

    ConvKArgs a;
    memset(&a, 0, sizeof(a));
    int Cin = 0, kc = 16;
    for (int i = 0; i < d->n_src; ++i) {
        const read_conv_src &s = d->src[i];
        READ_CHECK_ARG(s.data && (uintptr_t)s.data % 16 == 0, "read_gated_conv_forward: source %d null or misaligned", i);
        READ_CHECK_ARG(s.C >= 8 && s.C % 8 == 0, "read_gated_conv_forward: source %d has C=%d (need a multiple of 8)", i, s.C);
        READ_CHECK_ARG(s.shift >= -4 && s.shift <= 4, "read_gated_conv_forward: shift out of range");
        READ_CHECK_ARG((long long)s.srcH * s.srcW * s.C < (1ll << 31), "read_gated_conv_forward: source too large");
        const int needH = s.shift >= 0 ? ((d->inH - 1) << s.shift) + 1 : ((d->inH - 1) >> -s.shift) + 1;
        const int needW = s.shift >= 0 ? ((d->inW - 1) << s.shift) + 1 : ((d->inW - 1) >> -s.shift) + 1;
        READ_CHECK_ARG(s.srcH >= needH && s.srcW >= needW, "read_gated_conv_forward: source %d (%dx%d) too small for %dx%d at shift %d",
                       i, s.srcH, s.srcW, d->inH, d->inW, s.shift);
        if (s.C % 16) kc = 8;
        Cin += s.C;
        a.src[i].p = s.data;
        a.src[i].C = s.C;
        a.src[i].H = s.srcH;
        a.src[i].W = s.srcW;
        a.src[i].sl = s.shift > 0 ? s.shift : 0;
        a.src[i].sr = s.shift < 0 ? -s.shift : 0;
    }
    READ_CHECK_ARG(!d->mul || d->src[0].shift == 0, "read_gated_conv_forward: mul needs shift 0");
    {   // 1x1 layers whose sources are all multiples of 32 channels run 32-channel chunks (weights packed with kc = 32)
        bool all32 = d->ksize == 1;
        for (int i = 0; i < d->n_src; ++i) all32 = all32 && d->src[i].C % 32 == 0;
        if (all32 && (d->config >= 0 ? (d->config < N_CONFIGS && g_configs[d->config].KC == 32) : g_kc32 != 0)) kc = 32;
    }
    const int nchunks = Cin / kc;
    a.n_src = d->n_src;
    const int pad = (d->ksize - 1) / 2;
    const int outH = (d->inH + 2 * pad - d->ksize) / d->stride + 1;
    const int outW = (d->inW + 2 * pad - d->ksize) / d->stride + 1;
    READ_CHECK_ARG(outH >= 1 && outW >= 1, "read_gated_conv_forward: empty output");
    READ_CHECK_ARG((long long)outH * outW * d->out_cstride * 4 < OOB_LIMIT, "read_gated_conv_forward: output too large");
    const int CoutPad = pad32(d->Cout), groups = CoutPad / 32;
    int cfg = d->config;
    if (d->pre) {
        READ_CHECK_ARG((uintptr_t)d->pre % 4 == 0 && d->pre_shift >= 0 && d->pre_shift <= 4 && d->pre_f_off >= 0 && d->pre_m_off >= 0 &&
                           d->pre_cstride >= d->pre_f_off + d->Cout && d->pre_cstride >= d->pre_m_off + d->Cout,
                       "read_gated_conv_forward: bad pre-activation addend layout");
        READ_CHECK_ARG(d->preH >= ((outH - 1) >> d->pre_shift) + 1 && d->preW >= ((outW - 1) >> d->pre_shift) + 1 &&
                           (long long)d->preH * d->preW * d->pre_cstride * 4 < OOB_LIMIT,
                       "read_gated_conv_forward: pre-activation addend %dx%d does not cover the %dx%d output at shift %d", d->preH,
                       d->preW, outH, outW, d->pre_shift);
        READ_CHECK_ARG(!d->pre_bilinear || (d->pre_shift == 2 && d->ksize == 1 && d->stride == 1),
                       "read_gated_conv_forward: pre_bilinear needs pre_shift 2 on a 1x1 / stride-1 layer");
        a.pre = d->pre;
        a.pre_bil = d->pre_bilinear ? 1 : 0;
        a.pre_H = d->preH;
        a.pre_cstride = d->pre_cstride;
        a.pre_foff = d->pre_f_off;
        a.pre_moff = d->pre_m_off;
        a.pre_shift = d->pre_shift;
        a.pre_W = d->preW;
        a.pre_bytes = d->preH * d->preW * d->pre_cstride * 4;
    }
    a.mul = d->mul;
    a.wp = d->wpacked;
    a.wp_wino = d->wpacked_wino;
    a.wp_w16 = d->wpacked_w16;
    a.wp_w4 = d->wpacked_w4;
    a.params = d->params;
    a.residual = d->residual;
    a.out = d->out;
    a.inH = d->inH;
    a.inW = d->inW;
    a.outH = outH;
    a.outW = outW;
    a.Cout = d->Cout;
    a.CoutPad = CoutPad;
    a.out_cstride = d->out_cstride;
    a.elu = d->elu;
    a.linear = d->linear;
    a.ablate = g_ablate;
    a.fill_pad = d->fill_pad;
    a.out_fill = d->out_fill;

    // ---- at most four output channels, 3x3 / stride 1: the vector-pipe kernel (config -6 forces it)
    a.wp_sc = d->wpacked_sc;
    READ_CHECK_ARG(d->config != -6 || conv_uses_sc(d), "read_gated_conv_forward: the small-Cout kernel takes gated 3x3/s1 layers with "
                   "Cin = 32, Cout <= 4, one unshifted source and wpacked_sc");
    if (conv_uses_sc(d)) {
        READ_CHECK_ARG((uintptr_t)d->wpacked_sc % 64 == 0 && (uintptr_t)d->src[0].data % 16 == 0, "read_gated_conv_forward: wpacked_sc / source misaligned");
        a.tiles_x = ceil_div(outW, 32);
        const int cph = g_sc & 63;                                 // knob: channels per LDS phase
        const dim3 grid((unsigned)(a.tiles_x * ceil_div(outH, 8)));
        const bool c3 = d->Cout <= 3;
#define SC_LAUNCH(CPH, PPT) \
        do { if (c3) hipLaunchKernelGGL((gated_conv_smallc_kernel<32, CPH, PPT, 3>), grid, dim3(256), 0, stream, a); \
             else hipLaunchKernelGGL((gated_conv_smallc_kernel<32, CPH, PPT, 4>), grid, dim3(256), 0, stream, a); } while (0)
        if (cph == 32) SC_LAUNCH(32, 1);
        else if (cph == 16) SC_LAUNCH(16, 1);
        else SC_LAUNCH(8, 1);
#undef SC_LAUNCH
        READ_CHECK_LAUNCH();
        return READ_OK;
    }

    // ---- 1x1 layers on the f16 matrix cores: the split-operand pixel-lane kernel (config -10 forces it)
    READ_CHECK_ARG(d->config != -10 || conv_uses_pxh(d), "read_gated_conv_forward: the split-operand pixel-lane kernel takes 1x1/s1 layers with "
                   "Cin %% 16 == 0, Cin <= 256, Cout %% 4 == 0, 16-byte aligned tensors and wpacked_d3h");
    READ_CHECK_ARG(d->config != -11 || conv_uses_t3h(d), "read_gated_conv_forward: the split-operand implicit-GEMM kernel takes 3x3/s1 layers over one "
                   "unshifted source of 8, 16 or 32 channels with Cout %% 4 == 0, 16-byte aligned tensors and wpacked_t3h");
    const bool taps = conv_uses_t3h(d);
    if (taps || conv_uses_pxh(d)) {
        READ_CHECK_ARG((uintptr_t)(taps ? d->wpacked_t3h : d->wpacked_d3h) % 16 == 0, "read_gated_conv_forward: wpacked_d3h / wpacked_t3h misaligned");
        static int n_cu_h = 0;
        if (!n_cu_h) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu_h = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                      prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        a.wp_d3h = taps ? d->wpacked_t3h : d->wpacked_d3h;
        const int nsteps = taps ? (9 * Cin + 15) / 16 : Cin / 16;
        if (taps) a.tiles_y = Cin == 8 ? 3 : Cin == 16 ? 4 : 5;     // log2 of the source's channels
        const int gw = (groups % 2 == 0 && nsteps <= 8) ? 2 : 1;
        const size_t lds = (size_t)nsteps * 2 * gw * 2048 + 6 * 32 * gw * sizeof(float);     // <= 64 KiB of fragments + parameters: two workgroups per CU
        const int gsets = groups / gw;
        const int per_cu = lds > 78 * 1024 ? 1 : 2;
        // two pixel tiles per wave as soon as one tile per wave would need a second round of units: a wave's time per unit is the
        // latency of its activation stream (ring of four steps), the same for one tile or two
        const long slots = (long)n_cu_h * per_cu / gsets * 4;
        const int pt = gw == 2 ? 1 : ((long)ceil_div(outH * outW, 32) > (slots > 0 ? slots : 1) ? 2 : 1);
        a.nchunks = nsteps;
        a.n_units = ceil_div(outH * outW, 32 * pt);
        int per_set = (n_cu_h * per_cu) / gsets;
        const int want = ceil_div(a.n_units, 4);
        per_set = per_set < 1 ? 1 : per_set;
        per_set = per_set < want ? per_set : want;
        per_set = (per_set + 7) / 8 * 8;                             // the kernel's XCD-aware workgroup map
        const bool fullq = nsteps % 4 == 0;
        bool uni = true;
        for (int i = 0; i < d->n_src; ++i) uni = uni && d->src[i].C % 16 == 0;
        const int shape = gw == 2 ? 0 : pt == 2 ? 1 : 2, vi = taps ? 12 + shape : shape * 4 + (fullq ? 2 : 0) + (uni ? 1 : 0);
        static const conv_fn fns[15] = {
            gated_conv_pxh_kernel<1, 2, false, 0>, gated_conv_pxh_kernel<1, 2, false, 1>, gated_conv_pxh_kernel<1, 2, true, 0>, gated_conv_pxh_kernel<1, 2, true, 1>,
            gated_conv_pxh_kernel<2, 1, false, 0>, gated_conv_pxh_kernel<2, 1, false, 1>, gated_conv_pxh_kernel<2, 1, true, 0>, gated_conv_pxh_kernel<2, 1, true, 1>,
            gated_conv_pxh_kernel<1, 1, false, 0>, gated_conv_pxh_kernel<1, 1, false, 1>, gated_conv_pxh_kernel<1, 1, true, 0>, gated_conv_pxh_kernel<1, 1, true, 1>,
            gated_conv_pxh_kernel<1, 2, false, 2>, gated_conv_pxh_kernel<2, 1, false, 2>, gated_conv_pxh_kernel<1, 1, false, 2>};
        static bool attr_set_h[15] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false, false};
        if (!attr_set_h[vi]) {                                       // 64 KiB of dynamic LDS at Cin = 256 (74 KiB: 3x3 over 32 channels)
            READ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fns[vi]), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            attr_set_h[vi] = true;
        }
        hipLaunchKernelGGL(fns[vi], dim3((unsigned)(per_set * gsets)), dim3(256), lds, stream, a);
        READ_CHECK_LAUNCH();
        return READ_OK;
    }

    // ---- 1x1 layers: the pixel-lane kernel (config -2 forces it, -1 takes it whenever the layer qualifies)
    {
        const int nsteps = Cin / 8;
        const int gw = (groups % 2 == 0 && nsteps <= 16) ? 2 : 1;
        const bool fits = d->ksize == 1 && d->stride == 1 && !d->mul && !d->fill_pad && d->Cout % 4 == 0 && d->out_cstride % 4 == 0 &&
                          (uintptr_t)d->out % 16 == 0 && (uintptr_t)d->params % 16 == 0 && nsteps <= 32 &&
                          (!d->residual || (uintptr_t)d->residual % 16 == 0) &&
                          (!d->pre || ((uintptr_t)d->pre % 16 == 0 && d->pre_cstride % 4 == 0 && d->pre_f_off % 4 == 0 && d->pre_m_off % 4 == 0));
        READ_CHECK_ARG(d->config != -2 || fits, "read_gated_conv_forward: the pixel-lane kernel takes 1x1/s1 layers with Cin <= 256, "
                       "Cout %% 4 == 0 and 16-byte aligned tensors");
        // Measured per layer at 1216x352 (profiles/README.md): the pixel-lane kernel wins where the LDS-tiled kernels fall to
        // 8-channel chunks (SCM tails, cat[x(8), main]) or pad the last channel group (Cout = 56 / 120 / 248), 3-7 us per
        // layer; on the other 1x1 layers it is level or up to 10 us slower (its four 32-byte segments per pixel line
        // arrive as separate instructions) — conv_px = 1 takes it only for the former, 3 / 4 for every layer it fits.
        bool px_pick = false;
        for (int i = 0; i < d->n_src; ++i) px_pick = px_pick || d->src[i].C % 16 != 0;
        px_pick = px_pick || d->Cout % 32 != 0 || g_conv_px >= 3;
        const bool bil = d->pre && d->pre_bilinear;                // only this kernel samples the addend bilinearly
        READ_CHECK_ARG(!bil || (fits && d->config < 0), "read_gated_conv_forward: pre_bilinear needs the pixel-lane kernel (1x1/s1, Cin <= 256, "
                       "Cout %% 4 == 0, 16-byte aligned tensors, automatic config)");
        if (d->config == -2 || bil || (d->config == -1 && g_conv_px && fits && px_pick)) {
            static int n_cu_p = 0;
            if (!n_cu_p) {
                int dev = 0;
                hipDeviceProp_t prop;
                n_cu_p = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                          prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
            }
            const bool wide = g_conv_px == 2 || g_conv_px == 4;     // 128 accumulator registers per wave instead of 64
            const int pt = (gw == 2 ? 1 : 2) * (wide ? 2 : 1);
            const size_t lds = (size_t)nsteps * 2 * gw * 1024;
            int per_cu = (int)((160 * 1024) / (lds + 256));
            const int reg_cap = wide ? 2 : 3;
            per_cu = per_cu < 1 ? 1 : per_cu > reg_cap ? reg_cap : per_cu;
            const int gsets = groups / gw;
            a.nchunks = nsteps;
            a.n_units = ceil_div(outH * outW, 32 * pt);
            int per_set = (n_cu_p * per_cu) / gsets;
            const int want = ceil_div(a.n_units, 4);
            per_set = per_set < 1 ? 1 : per_set;
            per_set = per_set < want ? per_set : want;
            const bool fullq = nsteps % 4 == 0;
            conv_fn fn = gw == 2 ? (wide ? (fullq ? gated_conv_px_kernel<2, 2, true> : gated_conv_px_kernel<2, 2>)
                                         : (fullq ? gated_conv_px_kernel<1, 2, true> : gated_conv_px_kernel<1, 2>))
                                 : (wide ? (fullq ? gated_conv_px_kernel<4, 1, true> : gated_conv_px_kernel<4, 1>)
                                         : (fullq ? gated_conv_px_kernel<2, 1, true> : gated_conv_px_kernel<2, 1>));
            static bool attr_set[8] = {false, false, false, false, false, false, false, false};
            const int vi = (gw == 2 ? 2 : 0) + (wide ? 1 : 0) + (fullq ? 4 : 0);
            if (!attr_set[vi]) {                                    // 64 KiB of dynamic LDS at Cin = 256
                READ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                attr_set[vi] = true;
            }
            hipLaunchKernelGGL(fn, dim3((unsigned)(per_set * gsets)), dim3(256), lds, stream, a);
            READ_CHECK_LAUNCH();
            return READ_OK;
        }
    }
    if (d->linear) {
        // the plain-convolution epilogue exists in the workgroup-tiled kernels and in the Winograd kernel
        READ_CHECK_ARG(cfg < 0 || (cfg < N_CONFIGS && !g_configs[cfg].wave),
                       "read_gated_conv_forward: linear mode needs a workgroup-tiled or Winograd config");
        if (cfg < 0 && conv_uses_wino(d))
            for (int i = N_CONFIGS - 1; i >= 0; --i)
                if (g_configs[i].wino) cfg = i;
        for (int i = 0; cfg < 0 && i < N_CONFIGS; ++i) {
            const ConvConfig &k = g_configs[i];
            if (!k.wave && !k.wino && k.KS == d->ksize && k.S == d->stride && k.KC == kc && groups % (k.WN * k.QG) == 0 &&
                (!d->mul || k.fn_mul))
                cfg = i;
        }
    }
    if (cfg < 0 && conv_uses_wino(d))
        for (int i = N_CONFIGS - 1; i >= 0; --i)
            if (g_configs[i].wino) cfg = i;       // first Winograd entry = the product kernel
    if (cfg < 0) cfg = pick_config(d->ksize, d->stride, kc, groups, outH, outW);
    if (d->config < 0 && d->mul && cfg >= 0 && !g_configs[cfg].fn_mul)
        for (int i = 0; i < N_CONFIGS; ++i) {
            const ConvConfig &k = g_configs[i];
            if (k.fn_mul && k.KS == d->ksize && k.S == d->stride && k.KC == kc && groups % (k.WN * k.QG) == 0) {
                cfg = i;
                break;
            }
        }
    READ_CHECK_ARG(cfg >= 0 && cfg < N_CONFIGS, "read_gated_conv_forward: no kernel for k=%d s=%d kc=%d groups=%d",
                   d->ksize, d->stride, kc, groups);
    const ConvConfig &c = g_configs[cfg];
    READ_CHECK_ARG(c.KS == d->ksize && c.S == d->stride && c.KC == kc && groups % (c.WN * c.QG) == 0,
                   "read_gated_conv_forward: config %s does not fit k=%d s=%d kc=%d groups=%d", c.name, d->ksize,
                   d->stride, kc, groups);
    a.nchunks = nchunks;
    a.tiles_x = ceil_div(outW, 32);
    READ_CHECK_ARG(!d->pre || !c.wino, "read_gated_conv_forward: the Winograd kernel takes no pre-activation addend");
    READ_CHECK_ARG(!d->out_gated || ((c.wino || conv_uses_w4(d)) && d->linear && (uintptr_t)d->out_gated % 16 == 0 && d->block_h >= 0 &&
                                     d->valid_h <= d->block_h),
                   "read_gated_conv_forward: out_gated needs a linear launch on the Winograd kernel");
    a.out_gated = d->out_gated;
    a.blk_h = d->block_h;
    a.blk_valid = d->valid_h;
    const int tiles_y = ceil_div(outH, c.WM * c.P);
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)(groups / (c.WN * c.QG)));
    a.trace = ((size_t)grid.x * grid.y <= g_trace_records) ? g_trace : nullptr;
    if (c.wino) {
        READ_CHECK_ARG(d->wpacked_wino && (uintptr_t)d->wpacked_wino % 16 == 0, "read_gated_conv_forward: config %s needs wpacked_wino", c.name);
        READ_CHECK_ARG(d->n_src == 1 && d->src[0].shift == 0 && d->src[0].C % 16 == 0,
                       "read_gated_conv_forward: the Winograd kernel takes one un-resampled source with C %% 16 == 0");
        a.tiles_x = ceil_div(outW, 16);
        a.n_units = a.tiles_x * ceil_div(outH, 8) * groups;       // unit u = (tile u / groups, group u % groups)
        static int n_cu_w = 0;
        if (!n_cu_w) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu_w = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                      prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        int nwg = a.n_units < g_wino_wgs * n_cu_w ? a.n_units : g_wino_wgs * n_cu_w;    // persistent: two workgroups per CU
        nwg -= nwg % groups;                                          // keeps the group fixed per workgroup
        if (nwg < groups) nwg = groups;
        grid = dim3((unsigned)nwg, 1);
        a.wino_dby = (nwg / groups) / a.tiles_x;
        a.wino_dbx = (nwg / groups) % a.tiles_x;
        a.trace = ((size_t)grid.x * 4 <= g_trace_records) ? g_trace : nullptr;
    }
    if (c.wave) {
        // persistent grid: every wave walks units u = wave, wave + n_waves, ...
        a.tiles_y = ceil_div(outH, c.P);
        a.trace = nullptr;
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                    prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        // Balanced schedule: whole rounds of P-row units over all wave slots, the remainder as 1-row
        // units, so the last round costs about half a unit instead of a full one (the trace of the
        // workgroup-tiled kernel showed 20 % of a launch spent in a quarter-filled last round).
        const int ncol = a.tiles_x * (groups / c.QG);
        const long slots = (long)n_cu * c.wg_per_cu * 4;
        const long unitsP = (long)ncol * a.tiles_y;
        // (a balanced tail of 1-row units was tried and measured neutral-to-negative; all units are P rows)
        (void)slots;
        (void)unitsP;
        const int col_split = ncol;
        a.stagger_ticks = g_stagger_ticks;
        a.col_split = col_split;
        a.n_full = col_split * a.tiles_y;
        a.n_units = a.n_full + (ncol - col_split) * outH;
        const int want = ceil_div(a.n_units, 4), cap = n_cu * c.wg_per_cu;
        grid = dim3((unsigned)(want < cap ? want : cap), 1);
    }
    conv_fn fn = c.fn;
    // 3x3 / stride 2 on the direct split-operand kernel: units of 8 x 16 output pixels x 64 channels, one persistent workgroup per CU
    if (conv_uses_d3h_s2(d)) {
        READ_CHECK_ARG((uintptr_t)d->wpacked_d3h % 16 == 0, "read_gated_conv_forward: wpacked_d3h misaligned");
        a.wp_d3h = d->wpacked_d3h;
        a.nchunks = Cin / 32;
        a.tiles_x = ceil_div(outW, 16);
        const int pairs = (CoutPad + 63) / 64;
        a.n_units = a.tiles_x * ceil_div(outH, 8) * pairs;
        int dev = 0, n_cu_s = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu_s = prop.multiProcessorCount;
        int nwg = a.n_units < n_cu_s ? a.n_units : n_cu_s;
        nwg -= nwg % pairs;
        if (nwg < pairs) nwg = pairs;
        a.wino_dby = (nwg / pairs) / a.tiles_x;
        a.wino_dbx = (nwg / pairs) % a.tiles_x;
        a.trace = nullptr;
        if (d->ksize == 4) hipLaunchKernelGGL((gated_conv_d3h_s2_kernel<4>), dim3((unsigned)nwg), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((gated_conv_d3h_s2_kernel<3>), dim3((unsigned)nwg), dim3(512), 0, stream, a);
        READ_CHECK_LAUNCH();
        return READ_OK;
    }
    // Winograd F(4x4,3x3): units of 8 x 32 pixels x 32 channels, one persistent workgroup per CU
    const bool d3h = conv_uses_d3h(d), w4h = !d3h && conv_uses_w4h(d);
    if (d3h || w4h || conv_uses_w4(d)) {
        READ_CHECK_ARG((uintptr_t)(d3h ? d->wpacked_d3h : w4h ? d->wpacked_w4h : (const void *)d->wpacked_w4) % 16 == 0,
                       "read_gated_conv_forward: wpacked_w4 / wpacked_w4h / wpacked_d3h misaligned");
        a.wp_d3h = d->wpacked_d3h;
        READ_CHECK_ARG(!d->mul || (uintptr_t)d->mul % 16 == 0, "read_gated_conv_forward: mul misaligned");
        a.wp_w4h = d->wpacked_w4h;
        if (w4h || d3h) a.nchunks = Cin / 32;
        a.tiles_x = ceil_div(outW, 32);
        a.n_units = a.tiles_x * ceil_div(outH, 8) * groups;
        static int n_cu_4 = 0;
        if (!n_cu_4) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu_4 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                      prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        int nwg = a.n_units < n_cu_4 ? a.n_units : n_cu_4;
        nwg -= nwg % groups;
        if (g_w4_grid) {                                               // A/B: every workgroup the same number of units
            const int cap = n_cu_4 - n_cu_4 % groups > 0 ? n_cu_4 - n_cu_4 % groups : groups;
            nwg = ceil_div(ceil_div(a.n_units, ceil_div(a.n_units, cap)), groups) * groups;
            if (nwg > n_cu_4) nwg -= groups;
        }
        if (nwg < groups) nwg = groups;
        a.wino_dby = (nwg / groups) / a.tiles_x;
        a.wino_dbx = (nwg / groups) % a.tiles_x;
        a.trace = nullptr;
        conv_fn fn4 = d->linear ? (d->out_gated ? gated_conv_wino4_kernel<false, 0, 1> : gated_conv_wino4_kernel<false, 0, 2>) : d->mul ? gated_conv_wino4_kernel<true> : gated_conv_wino4_kernel<false>;
        READ_CHECK_ARG(!d->linear || !d->mul, "read_gated_conv_forward: linear launches take no multiplier");
#ifdef READ_DEBUG_KNOBS
        if (!d->mul && g_abl) {
            switch (g_abl) {
#define READ_ABL_CASE(n) case n: fn4 = gated_conv_wino4_kernel<false, n>; break;
            READ_ABL_CASE(1) READ_ABL_CASE(7) READ_ABL_CASE(8) READ_ABL_CASE(24) READ_ABL_CASE(31) READ_ABL_CASE(32) READ_ABL_CASE(64)
            READ_ABL_CASE(128) READ_ABL_CASE(256) READ_ABL_CASE(511) READ_ABL_CASE(383) READ_ABL_CASE(512)
#undef READ_ABL_CASE
            default: break;
            }
        }
#endif
#ifdef READ_DEBUG_KNOBS
        if (g_w4x2 && !d->linear && !d->mul) {   // negative result (see the kernel): eight waves per workgroup, frequencies split over wave pairs
            hipLaunchKernelGGL(gated_conv_wino4x2_kernel, dim3((unsigned)nwg), dim3(512), 0, stream, a);
            READ_CHECK_LAUNCH();
            return READ_OK;
        }
#endif
        if (d3h) {
            conv_fn fnd = d->mul ? gated_conv_d3h_kernel<true> : gated_conv_d3h_kernel<false>;
#ifdef READ_DEBUG_KNOBS
            if (!d->mul && g_abl) {
                switch (g_abl) {
#define READ_ABL_CASE(n) case n: fnd = gated_conv_d3h_kernel<false, n>; break;
                READ_ABL_CASE(1) READ_ABL_CASE(2) READ_ABL_CASE(4) READ_ABL_CASE(8) READ_ABL_CASE(16) READ_ABL_CASE(32) READ_ABL_CASE(7) READ_ABL_CASE(63) READ_ABL_CASE(55)
#undef READ_ABL_CASE
                default: break;
                }
            }
#endif
            hipLaunchKernelGGL(fnd, dim3((unsigned)nwg), dim3(512), 0, stream, a);
            READ_CHECK_LAUNCH();
            return READ_OK;
        }
        if (w4h) fn4 = gated_conv_wino4h_kernel<>;
#ifdef READ_DEBUG_KNOBS
        if (w4h && g_abl) {
            switch (g_abl) {
#define READ_ABL_CASE(n) case n: fn4 = gated_conv_wino4h_kernel<n>; break;
            READ_ABL_CASE(1) READ_ABL_CASE(2) READ_ABL_CASE(3) READ_ABL_CASE(4) READ_ABL_CASE(7) READ_ABL_CASE(8) READ_ABL_CASE(15) READ_ABL_CASE(32) READ_ABL_CASE(64)
            READ_ABL_CASE(128) READ_ABL_CASE(256) READ_ABL_CASE(512) READ_ABL_CASE(1024) READ_ABL_CASE(1007) READ_ABL_CASE(2047 - 1024) READ_ABL_CASE(544) READ_ABL_CASE(2048) READ_ABL_CASE(4096) READ_ABL_CASE(6144) READ_ABL_CASE(40) READ_ABL_CASE(8192) READ_ABL_CASE(16384) READ_ABL_CASE(24576) READ_ABL_CASE(32768) READ_ABL_CASE(65536) READ_ABL_CASE(122880)
#undef READ_ABL_CASE
            default: break;
            }
        }
#endif
#ifdef READ_DEBUG_KNOBS
        if (w4h && g_w4h_waves == 8) {
            conv_fn fn8 = gated_conv_wino4h2_kernel<>;
            if (g_abl) {
                switch (g_abl) {
#define READ_ABL_CASE(n) case n: fn8 = gated_conv_wino4h2_kernel<n>; break;
                READ_ABL_CASE(1) READ_ABL_CASE(8) READ_ABL_CASE(9) READ_ABL_CASE(128) READ_ABL_CASE(256) READ_ABL_CASE(1024) READ_ABL_CASE(1033)
#undef READ_ABL_CASE
                default: break;
                }
            }
            hipLaunchKernelGGL(fn8, dim3((unsigned)nwg), dim3(512), 0, stream, a);
            READ_CHECK_LAUNCH();
            return READ_OK;
        }
#endif
        hipLaunchKernelGGL(fn4, dim3((unsigned)nwg), dim3(256), 0, stream, a);
        READ_CHECK_LAUNCH();
        return READ_OK;
    }
    if (c.wino && a.trace && !d->mul) fn = gated_conv_wino_kernel<true, false>;
    if (d->mul) {
        READ_CHECK_ARG(c.fn_mul, "read_gated_conv_forward: config %s has no multiply variant", c.name);
        READ_CHECK_ARG((uintptr_t)d->mul % 16 == 0, "read_gated_conv_forward: mul misaligned");
        fn = c.fn_mul;
    }
    // the wave-autonomous Winograd kernel (same units, same grid) whenever its weight order was supplied; linear launches
    // (training path) stay on the row-per-wave kernel, which carries the plain-convolution epilogue
    // ... and, by default, for layers whose last channel group is mostly padding (the 32 -> 3 output layer): its waves without a real
    // channel skip their MFMAs, the row-per-wave kernel pays for all 32 padded channels
    const bool mostly_padding = d->Cout <= 8;
    if (c.wino && !d->linear && !a.trace && d->wpacked_w16 && (d->config == -3 || (d->config < 0 && (g_w16 || mostly_padding)))) {
        READ_CHECK_ARG((uintptr_t)d->wpacked_w16 % 16 == 0, "read_gated_conv_forward: wpacked_w16 misaligned");
        fn = d->mul ? gated_conv_wino16s_kernel<true> : gated_conv_wino16s_kernel<false>;
#ifdef READ_DEBUG_KNOBS
        if (!d->mul && g_abl) {
            switch (g_abl) {
#define READ_ABL_CASE(n) case n: fn = gated_conv_wino16s_kernel<false, n>; break;
            READ_ABL_CASE(1) READ_ABL_CASE(2) READ_ABL_CASE(4) READ_ABL_CASE(8) READ_ABL_CASE(16) READ_ABL_CASE(32) READ_ABL_CASE(63)
#undef READ_ABL_CASE
            default: break;
            }
        }
#endif
    }
    hipLaunchKernelGGL(fn, grid, dim3(256), 0, stream, a);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// the automatic choice takes the Winograd kernel for every layer it can run (measured faster on all four levels)
int conv_uses_wino(const read_conv_desc *d)
{
    return d->config < 0 && g_use_wino && !d->pre && d->ksize == 3 && d->stride == 1 && d->n_src == 1 &&
           d->src[0].shift == 0 && d->src[0].C % 16 == 0 && d->wpacked_wino && d->src[0].C <= g_use_wino;
}

// F(4x4,3x3): non-linear 3x3 / stride-1 launches with full 32-channel groups and at least conv_w4 input channels (config -5 forces it)
int conv_uses_w4(const read_conv_desc *d)
{
    const bool shape = !d->pre && (!d->linear || !d->residual) && d->ksize == 3 && d->stride == 1 && d->n_src == 1 && d->src[0].shift == 0 &&
                       d->src[0].C % 16 == 0 && d->src[0].C >= 32 && (d->Cout % 32 == 0 || (d->linear && d->Cout % 8 == 0)) && !d->fill_pad && d->wpacked_w4 &&
                       d->out_cstride % 4 == 0 &&                                                     // 128-bit stores
                       (long long)d->src[0].srcH * d->src[0].srcW * d->src[0].C * 4 < (1ll << 31) &&  // 32-bit buffer offsets
                       (long long)d->inH * d->inW * d->out_cstride * 4 < (1ll << 31);
    return shape && (d->config == -5 || (d->config == -1 && g_w4 > 0 && d->src[0].C >= g_w4));
}

// ... on the f16 matrix cores with split operands: the gated (non-linear) launches of that family with whole 32-channel chunks whose
// split operand was supplied (config -7 forces it; the training path's linear launches stay on the fp32 kernel)
int conv_uses_w4h(const read_conv_desc *d)
{
    // (FAM's x1 * x2 stays on the fp32 kernel: the second patch costs the transform thread another 72 registers, and the
    //  variant with a shallower weight ring measured SLOWER than the fp32 kernel — 101 / 79 / 68 us against 77 / 69 / 64 at C = 64 / 128 / 256)
    if (!d->wpacked_w4h || d->linear || d->mul || d->src[0].C % 32 != 0 || d->Cout % 32 != 0) return 0;
    if (d->config == -7) {
        read_conv_desc t = *d;
        t.config = -5;
        t.wpacked_w4 = reinterpret_cast<const float *>(d->wpacked_w4h);    // the shape test of the family (any non-null operand)
        return conv_uses_w4(&t);
    }
    if (d->config != -1 || g_w4h <= 0 || d->src[0].C < g_w4h) return 0;
    read_conv_desc t = *d;
    t.wpacked_w4 = reinterpret_cast<const float *>(d->wpacked_w4h);
    return conv_uses_w4(&t);
}

// the DIRECT split-operand kernel: the same launches (FAM's x1 * x2 included: a multiplication at staging time); config -8 forces it
int conv_uses_d3h(const read_conv_desc *d)
{
    if (!d->wpacked_d3h || d->linear || d->src[0].C % 32 != 0 || d->Cout % 32 != 0) return 0;
    const int min_c = d->mul ? g_d3h_fam : g_d3h;
    if (!(d->config == -8 || (d->config == -1 && min_c > 0 && d->src[0].C >= min_c))) return 0;
    read_conv_desc t = *d;
    t.config = -5;
    t.wpacked_w4 = reinterpret_cast<const float *>(d->wpacked_d3h);        // the shape test of the family (any non-null operand)
    return conv_uses_w4(&t);
}

// ... and at stride 2 (the encoder's down-sampling layers): gated 3x3 / stride-2 single-source launches with whole 32-channel chunks in and
// whole 64-channel pairs out, no multiplier / addend / fill (config -9 forces it; read_tuning_set("conv_d3h_s2", 0) switches it off)
int conv_uses_d3h_s2(const read_conv_desc *d)
{
    const bool shape = d->wpacked_d3h && !d->linear && !d->mul && !d->pre && !d->fill_pad && (d->ksize == 3 || d->ksize == 4) && d->stride == 2 && d->n_src == 1 &&
                       d->src[0].shift == 0 && d->src[0].C % 32 == 0 && d->Cout % 32 == 0 && d->out_cstride % 4 == 0 &&
                       (long long)d->src[0].srcH * d->src[0].srcW * d->src[0].C * 4 < (1ll << 31) &&
                       (long long)d->inH * d->inW * d->out_cstride * 4 < (1ll << 31);
    return shape && (d->config == -9 || (d->config == -1 && g_d3h_s2 > 0 && d->src[0].C >= g_d3h_s2));
}

// 1x1 / stride-1 layers on the split-operand pixel-lane kernel (gated or linear, any number of sources, residual, nearest pre-activation addend):
// whole k16 steps, at most 64 KiB of weight fragments per group set, everything 16-byte aligned (config -10 forces it;
// read_tuning_set("conv_pxh", 0) switches it off)
int conv_uses_pxh(const read_conv_desc *d)
{
    if (!d->wpacked_d3h || d->ksize != 1 || d->stride != 1 || d->mul || d->fill_pad || d->n_src < 1 || d->n_src > READ_CONV_MAX_SRC) return 0;
    if (d->pre && d->pre_bilinear) return 0;                          // the bilinear addend (an option of the plan, off) stays on the fp32 pixel-lane kernel
    int Cin = 0;
    for (int i = 0; i < d->n_src; ++i) {
        if (d->src[i].C < 8 || d->src[i].C % 8 != 0 || (uintptr_t)d->src[i].data % 16 != 0) return 0;
        Cin += d->src[i].C;
    }
    const bool shape = Cin % 16 == 0 && Cin <= 256 && d->Cout % 4 == 0 && d->out_cstride % 4 == 0 && (uintptr_t)d->out % 16 == 0 &&
                       (uintptr_t)d->params % 16 == 0 && (!d->residual || (uintptr_t)d->residual % 16 == 0) &&
                       (!d->pre || ((uintptr_t)d->pre % 16 == 0 && d->pre_cstride % 4 == 0 && d->pre_f_off % 4 == 0 && d->pre_m_off % 4 == 0));
    return shape && (d->config == -10 || (d->config == -1 && g_pxh > 0 && Cin >= g_pxh));
}

// 3x3 / stride-1 layers over one unshifted source of 8, 16 or 32 channels as an implicit GEMM on the same kernel (k = tap C + channel;
// read_conv_pack_t3h_host): the layers that read the 8-channel descriptor pyramid by default (read_tuning_set("conv_t3h", max Cin), 0 = never;
// config -11 forces it)
int conv_uses_t3h(const read_conv_desc *d)
{
    if (!d->wpacked_t3h || d->ksize != 3 || d->stride != 1 || d->mul || d->fill_pad || d->pre || d->n_src != 1 || d->src[0].shift != 0) return 0;
    const int C = d->src[0].C;
    const bool shape = (C == 8 || C == 16 || C == 32) && d->Cout % 4 == 0 && d->out_cstride % 4 == 0 && (uintptr_t)d->out % 16 == 0 &&
                       (uintptr_t)d->src[0].data % 16 == 0 && (uintptr_t)d->params % 16 == 0 && (!d->residual || (uintptr_t)d->residual % 16 == 0) &&
                       (long long)d->inH * d->inW * C * 4 < (1ll << 31);
    // (automatic choice from 16 K pixels on: at 44 x 152 the fp32 direct kernel measured 11.5 us against 14.2)
    return shape && (d->config == -11 || (d->config == -1 && g_t3h > 0 && C <= g_t3h && (long long)d->inH * d->inW >= 16384));
}

// gated 3x3 / stride-1 layers with at most four output channels and 32 input channels (READ's output layer)
int conv_uses_sc(const read_conv_desc *d)
{
    const bool shape = d->ksize == 3 && d->stride == 1 && d->n_src == 1 && d->src[0].shift == 0 && d->src[0].C == 32 && d->Cout >= 1 &&
                       d->Cout <= 4 && !d->linear && !d->residual && !d->pre && !d->mul && d->wpacked_sc &&
                       (d->out_cstride != 4 || (uintptr_t)d->out % 16 == 0) &&
                       (long long)d->src[0].srcH * d->src[0].srcW * d->src[0].C * 4 < (1ll << 31);
    return shape && (d->config == -6 || (d->config == -1 && g_sc));
}

int conv_kc_for(const read_conv_desc *d)
{
    int kc = 16;
    for (int i = 0; i < d->n_src; ++i)
        if (d->src[i].C % 16) kc = 8;
    return kc;
}

}  // namespace readhip

extern "C" int read_conv_kernel_family(const read_conv_desc *desc)
{
    if (!desc) return -1;
    if (readhip::conv_uses_t3h(desc)) return 8;
    if (readhip::conv_uses_sc(desc)) return 1;
    if (readhip::conv_uses_pxh(desc)) return 7;
    if (readhip::conv_uses_d3h(desc) || readhip::conv_uses_d3h_s2(desc)) return 6;
    if (readhip::conv_uses_w4h(desc)) return 5;
    if (readhip::conv_uses_w4(desc)) return 4;
    if (readhip::conv_uses_wino(desc)) return 2;
    return 0;
}

extern "C" int read_gated_conv_forward(const read_conv_desc *desc, void *stream)
{
    // the fragment-order checks (a NULL or aliased wpacked against the kernel family this launch takes) live in
    // launch_gated_conv: the UNet executor (unet.cpp) calls that directly and must get the same refusal
    return readhip::launch_gated_conv(desc, as_stream(stream));
}

// ---- measurement aid (bench.py: roofline.mfma_sustained): the rate the fp32 matrix path delivers when a wave does nothing else.
// One workgroup of four waves per CU (one wave per SIMD, as the F(4x4) kernel runs), every wave `iters` rounds of 16 independent
// v_mfma_f32_16x16x4_f32.  The guide's 157 TF peak is quoted at the 2.4 GHz boost clock; under matrix load the chip settles lower,
// and the fraction of THAT rate is what says how much of the pipe a kernel leaves idle.
namespace {
__global__ __launch_bounds__(256, 1) void mfma_f32_rate_kernel(float *out, int iters)
{
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a0 = 0.37f + 0.001f * threadIdx.x, b0 = 1.0f - 0.002f * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[m], 0, 0, 0);
        a0 = a0 * 0.999f + 0.0007f;
        b0 = b0 * 1.0001f - 0.0001f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1234.5678f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the accumulators live
}
}  // namespace

extern "C" int read_mfma_f32_rate_probe(int iters, float *scratch, double *flops, void *stream)
{
    READ_CHECK_ARG(iters > 0 && scratch && flops, "read_mfma_f32_rate_probe: bad arguments");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        readhip::set_error("read_mfma_f32_rate_probe: cannot query the device");
        return READ_EHIP;
    }
    hipLaunchKernelGGL(mfma_f32_rate_kernel, dim3((unsigned)cus), dim3(256), 0, as_stream(stream), scratch, iters);
    READ_CHECK_LAUNCH();
    *flops = (double)cus * 4.0 * (double)iters * 16.0 * 2048.0;            // 16 x 16 x 4 x 2 flops per MFMA
    return READ_OK;
}

extern "C" int read_bilinear_up4(const float *in, int inH, int inW, int C, float *out, void *stream)
{
    READ_CHECK_ARG(in && out && inH >= 1 && inW >= 1, "read_bilinear_up4: null pointer or empty input");
    READ_CHECK_ARG(C >= 4 && C % 4 == 0, "read_bilinear_up4: C must be a multiple of 4");
    READ_CHECK_ARG((uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0, "read_bilinear_up4: misaligned pointer");
    const long long total = (long long)inH * 4 * inW * 4 * (C / 4);
    long long blocks = ceil_div64(total, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(bilinear_up4_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in, inH, inW, C, out);
    READ_CHECK_LAUNCH();
    return READ_OK;
}
