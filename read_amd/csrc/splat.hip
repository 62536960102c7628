// Z-buffered point splat for gfx950.
//
// Replaces the reference's per-scale, per-camera DepthProject launches
// (MyRender/CloudProjection/point_render.cu:125-200; GL twin READ/gl/render.py:52-85).  Every accepted point becomes
// a packed 64-bit key (fp32 depth bits << 32 | point id) folded into a level-0 key image with an unsigned 64-bit
// atomic min — min depth, ties -> min id, order independent.  One resolve pass derives levels
// 1..4 by 2x2 key-min (exactly the reference's five rasterisations, App. A.4), unpacks (id, depth) and leaves the key
// image EMPTY for the next frame.
//
// Two paths produce the key image (bit-identical results):
//   * plain path (any B, any size, small clouds): one pass over the cloud for up to 8 cameras, agent-scope atomics;
//     for one camera with a warm start from the previous frame's winners and an LDS-resident hierarchical-Z.
//   * cell-ordered, XCD-striped path (the per-frame render path: one camera, >= 2^20 points, W % 16 == 0):
//     the cloud is kept Morton-sorted in chunks of 1024 points with bounding boxes, so whole chunks outside the
//     frustum or behind the far bound of every 4x4 pixel block they can touch are never read.  The early-z test does
//     not read the key image at all: atomics on gfx950 execute memory-side and DROP the line from the issuing XCD's
//     L2, so in round 1 every early-z read after an atomic re-fetched 128 bytes for 8 (2.7x the algorithmic HBM
//     traffic, profiles/r1_hbm_traffic_per_kernel.md).  Instead a 4-byte-per-pixel image of depth UPPER BOUNDS
//     ("zimg") is kept with plain loads and stores: whoever issues an atomic for depth d also stores d.  Stores race,
//     but every value ever stored is the depth of a real point of the cloud at that pixel, hence >= the final depth
//     — a stale or lost update only lets a few more points through to the (exact) atomic.  The warm start needs no
//     atomics either: last frame's front points are re-projected and only their depths are stored as bounds; the
//     points themselves come by in the passes and pass the test (ties pass).  The SCREEN is cut into 8 column strips,
//     one per XCD, and a workgroup prefers the work of the strip its XCD owns, so that readers and writers of a zimg
//     line share an L2 (the per-XCD L2s are not coherent inside a launch); this is an optimisation only — any
//     workgroup may process any strip's work, and does once its own list is empty.
//
// HBM-bound: algorithmic bytes per frame = 12*N (xyz read once) + 8*sum_l(h_l*w_l) (id + depth).
// The arithmetic that decides which PIXEL a point lands in is bit-exact fp32: no FMA contraction, IEEE division,
// left-to-right dot products (helper_math.h:1252-1255).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)

using namespace readhip;

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int MAX_CAMS = 8;          // cameras folded into one pass over the points
constexpr int PTS_PER_THREAD = 4;    // 3 x float4 = 4 points
constexpr int MAX_STRIPS = 8;        // one per XCD

struct CamSet {
    float m[MAX_CAMS][16];
};
struct Cam1 {
    float m[16];
};

// The three IEEE divisions of point_render.cu:119 (c0 / c3, c1 / c3, c2 / c3) with ONE reciprocal.
// hipcc expands a correctly rounded fp32 a / b into  sb = v_div_scale(b), sa = v_div_scale(a), r = v_rcp(sb), e = fma(-sb, r, 1),
// r = fma(e, r, r), q = sa r, t = fma(-sb, q, sa), q = fma(t, r, q), t = fma(-sb, q, sa), v_div_fmas(t, r, q), v_div_fixup — 11
// instructions, 33 for a point (a quarter of pass A's vector instructions, and pass A is issue-bound).  v_div_scale only scales
// when an exponent is extreme: for 2^-40 <= |b| <= 2^40 it returns b itself, and it returns a itself unless |a| < 2^-103 (then
// |a / b| < 2^-63: n + 1 rounds to 1 whichever way the quotient was rounded — pixel and depth come out the same) or
// |a| >= 2^56 |b| (then |a / b| > 1 on both paths, or inf / NaN: the point is rejected either way).  With nothing scaled
// v_div_fmas is a plain fma and v_div_fixup returns its input, so the reciprocal and its Newton step (they depend on b alone)
// can be shared and each quotient is the same five instructions on the same operands as in the compiler's expansion — the
// same bits.  Outside the window the whole wave takes the compiler's divisions.  18 instead of 33 instructions per point;
// tests/test_gpu_splat.py::test_shared_reciprocal_projection_is_ieee_division compares > 10^8 device points with the host's
// IEEE divisions (random, window-edge, tiny, huge, zero and non-finite operands).
__device__ __forceinline__ void div3_ieee(float a0, float a1, float a2, float b, float &q0, float &q1, float &q2)
{
    const float ab = fabsf(b);
    const bool window = (ab >= 0x1p-40f) & (ab <= 0x1p40f);
    if (__builtin_expect(__ballot(!window) == 0ull, 1)) {
        float r = __builtin_amdgcn_rcpf(b);
        const float e = __builtin_fmaf(-b, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        float t;
        q0 = a0 * r;
        q1 = a1 * r;
        q2 = a2 * r;
        t = __builtin_fmaf(-b, q0, a0);
        q0 = __builtin_fmaf(t, r, q0);
        t = __builtin_fmaf(-b, q1, a1);
        q1 = __builtin_fmaf(t, r, q1);
        t = __builtin_fmaf(-b, q2, a2);
        q2 = __builtin_fmaf(t, r, q2);
        t = __builtin_fmaf(-b, q0, a0);
        q0 = __builtin_fmaf(t, r, q0);
        t = __builtin_fmaf(-b, q1, a1);
        q1 = __builtin_fmaf(t, r, q1);
        t = __builtin_fmaf(-b, q2, a2);
        q2 = __builtin_fmaf(t, r, q2);
    } else {
        q0 = a0 / b;
        q1 = a1 / b;
        q2 = a2 / b;
    }
}

// point_render.cu:135-147 for one point and one camera; returns the pixel or -1.
__device__ __forceinline__ int project_one(float x, float y, float z, const float *M, int W, int H,
                                           float &depth, int &xx_out, int &yy_out)
{
    const float c0 = M[0] * x + M[1] * y + M[2] * z + M[3] * 1.0f;
    const float c1 = M[4] * x + M[5] * y + M[6] * z + M[7] * 1.0f;
    const float c2 = M[8] * x + M[9] * y + M[10] * z + M[11] * 1.0f;
    const float c3 = M[12] * x + M[13] * y + M[14] * z + M[15] * 1.0f;
    float nx, ny, nz;
    div3_ieee(c0, c1, c2, c3, nx, ny, nz);
    // NaN compares false everywhere: written so that NaN is rejected (canonical semantics).
    const bool inside = (nx >= -1.0f) & (nx <= 1.0f) & (ny >= -1.0f) & (ny <= 1.0f) &
                        (nz >= -1.0f) & (nz <= 1.0f);
    const float u = ((float)W * (nx + 1.0f)) * 0.5f;
    const float v = ((float)H * (1.0f - ny)) * 0.5f;
    depth = (nz + 1.0f) * 0.5f;
    const int xx = (int)u, yy = (int)v;
    const bool ok = inside & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H);
    xx_out = xx;
    yy_out = yy;
    return ok ? yy * W + xx : -1;
}

// Plain-path policies (read_tuning_set("splat_mode", m)):
//   MODE_AGENT one key image per camera, agent-scope atomics (memory-side), early-z through the L2.
//   MODE_HIZ   (default) MODE_AGENT plus, for a single camera, a temporal warm start and a hierarchical-Z reject in
//              LDS: the previous frame's per-pixel winners are re-projected with the new camera and folded in, a
//              conservative far bound per 4x4-pixel block (max of the current depths, +inf if any pixel is empty) is
//              built, and every workgroup copies that bound image into LDS.  A point whose depth exceeds its block's
//              bound cannot win (keys only decrease), so it is dropped without touching global memory.
//              Exact: seeds are real points of this cloud, bounds are upper bounds of the final depths.
enum { MODE_AGENT = 1, MODE_HIZ = 7 };

// relaxed agent-scope load = global_load_dwordx2 sc1: served by L2, never by the CU's stale L1
__device__ __forceinline__ unsigned long long peek_key_agent(const unsigned long long *k)
{
    return __hip_atomic_load(k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fold_key_agent(unsigned long long *k, unsigned long long key)
{
    __hip_atomic_fetch_min(k, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Where pixel p's key lives in the key image of the striped path.  Mode 0: slot p.  Mode 1: slot 8 p (one key per 64 bytes).
// Mode 2: slot (p * odd) mod 2^k, a bijection that scatters neighbouring pixels over the whole image — the memory-side atomic
// units serialise atomics that hit the same line / channel, and the candidates of a round are Morton neighbours, i.e.
// pixels of a few adjacent 128-byte lines (read_tuning_set("splat_kslot", m); measured in profiles/README.md).
struct KeySlots {
    int mode;
    unsigned mask;           // 2^k - 1 >= W*H - 1 (mode 2)
};
__device__ __forceinline__ unsigned key_slot(const KeySlots ks, unsigned pix)
{
    return ks.mode == 0 ? pix : (ks.mode == 1 ? pix << 3 : (pix * 0x9E3779B1u) & ks.mask);
}

// Far bounds are stored as 16 bits: the upper half (bfloat16, truncated = rounded DOWN) of e = fl(1 - d_max).  Depth
// is d = 1 - O(znear / z), so e keeps 8 mantissa bits of the DISTANCE (0.4 %) where a half-precision d would resolve
// only ~5 m at 30 m.  Reject iff fl(1 - d) < bound: rounding is monotonic, so fl(1 - d) < fl(1 - d_max) implies
// d > d_max strictly (ties pass), and truncation only lowers the bound.  Empty block -> -1 (never rejects).
__device__ __forceinline__ unsigned short hiz_encode(unsigned depth_bits_max)
{
    if (depth_bits_max > 0x7f800000u) return 0xbf80;                  // a pixel of the block is still EMPTY: e = -1
    return (unsigned short)(__float_as_uint(1.0f - __uint_as_float(depth_bits_max)) >> 16);
}
__device__ __forceinline__ bool hiz_reject(unsigned short bound, float d)
{
    return (1.0f - d) < __uint_as_float((unsigned)bound << 16);
}

// NP points of one thread against one camera (plain path): all projections first, then all early-z reads in flight
// together, then the (few) atomics — no dependent memory round trip per point.
template <bool HIZ, int NP>
__device__ __forceinline__ void splat_points(const float (&px)[NP], const float (&py)[NP], const float (&pz)[NP],
                                             unsigned id0, int nvalid, const float *M, int W, int H,
                                             unsigned long long *keys, const unsigned short *hiz = nullptr, int nbx = 0,
                                             unsigned *stat = nullptr)
{
    int pix[NP];
    unsigned long long key[NP], seen[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        float d;
        int xx, yy;
        pix[k] = project_one(px[k], py[k], pz[k], M, W, H, d, xx, yy);
        if (k >= nvalid) pix[k] = -1;
        if (stat && pix[k] >= 0) stat[0]++;                 // visible
        if (HIZ) {
            // LDS-resident far bound of the point's 4x4 block; strictly greater cannot win (ties must pass)
            if (hiz_reject(hiz[pix[k] >= 0 ? (yy >> 2) * nbx + (xx >> 2) : 0], d)) pix[k] = -1;
        }
        if (stat && pix[k] >= 0) stat[1]++;                 // survived the LDS hi-z (or no hi-z)
        key[k] = ((unsigned long long)__float_as_uint(d) << 32) | (id0 + k);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) seen[k] = pix[k] >= 0 ? peek_key_agent(keys + pix[k]) : 0ull;
    // keys only ever decrease, so "not smaller than what I can see" is final
#pragma unroll
    for (int k = 0; k < NP; ++k)
        if (key[k] < seen[k]) {
            fold_key_agent(keys + pix[k], key[k]);
            if (stat) stat[2]++;                            // atomics issued
        }
}

// Point groups (4 points) are split into chunks of 256 groups (1024 points).  The bootstrap pass of MODE_HIZ
// takes every sub-th chunk, the main pass the others.  Threads walk a COMPACT index t over the groups of
// their pass (a strided walk over all groups with a skip test aliases with the power-of-two grid stride:
// chunk % sub never changes along a thread's walk and 1/sub of the threads would do all the work).
__device__ __forceinline__ long long pass_groups(long long groups, int sub, int sel)
{
    if (sub <= 0 || sel == 0) return groups;
    const long long chunks = (groups + 255) >> 8;
    const long long boot = (chunks + sub - 1) / sub;                 // chunks 0, sub, 2 sub, ...
    return (sel == 1 ? boot : chunks - boot) << 8;                   // upper bound; map_group() may return >= groups
}
__device__ __forceinline__ long long map_group(long long t, int sub, int sel)
{
    if (sub <= 0 || sel == 0) return t;
    const long long c = t >> 8;
    const long long chunk = sel == 1 ? c * sub : (c / (sub - 1)) * sub + (c % (sub - 1)) + 1;
    return (chunk << 8) | (t & 255);
}

__global__ __launch_bounds__(256) void splat_project_kernel(const float *__restrict__ xyz, long long n,
                                                            CamSet cams, int B, int W, int H,
                                                            unsigned long long *__restrict__ keys,
                                                            int vec_ok, int sub_mod, unsigned long long *stats)
{
    unsigned st_local[3] = {0, 0, 0};
    unsigned *stp = stats ? st_local : nullptr;
    // sub_mod > 0: bootstrap pass of MODE_HIZ — only every sub_mod-th chunk of 256 point groups (1024 points)
    const long long npx = (long long)W * H;
    const long long groups = n / PTS_PER_THREAD;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;

    if (vec_ok) {
        const float4 *xyz4 = reinterpret_cast<const float4 *>(xyz);
        const long long npass = pass_groups(groups, sub_mod, sub_mod > 0 ? 1 : 0);
        for (long long t = tid0; t < npass; t += nthreads) {
            const long long g = map_group(t, sub_mod, sub_mod > 0 ? 1 : 0);
            if (g >= groups) continue;
            // 48 contiguous bytes per lane = 4 points
            const float4 a = xyz4[3 * g + 0];
            const float4 b = xyz4[3 * g + 1];
            const float4 c = xyz4[3 * g + 2];
            const float px[4] = {a.x, a.w, b.z, c.y};
            const float py[4] = {a.y, b.x, b.w, c.z};
            const float pz[4] = {a.z, b.y, c.x, c.w};
            for (int cam = 0; cam < B; ++cam)
                splat_points<false, 4>(px, py, pz, (unsigned)(g * PTS_PER_THREAD), 4, cams.m[cam], W, H,
                                       keys + (long long)cam * npx, nullptr, 0, stp);
        }
    }
    // tail (n % 4 points), or everything when the pointer is not 16-byte aligned
    const long long first = vec_ok ? groups * PTS_PER_THREAD : 0;
    for (long long i = first + tid0; i < n && sub_mod <= 0; i += nthreads) {     // (the bootstrap pass leaves the tail to the main pass)
        const float px[1] = {xyz[3 * i + 0]}, py[1] = {xyz[3 * i + 1]}, pz[1] = {xyz[3 * i + 2]};
        for (int cam = 0; cam < B; ++cam)
            splat_points<false, 1>(px, py, pz, (unsigned)i, 1, cams.m[cam], W, H, keys + (long long)cam * npx);
    }
    if (stats)
        for (int i = 0; i < 3; ++i) atomicAdd(stats + i, (unsigned long long)st_local[i]);
}


// ---- workspace header ------------------------------------------------------------------------------------------
struct SplatHeader {          // first bytes of the workspace
    int valid;                // 0: no previous frame; 1: prev[0] holds the winners' point ids (plain path);
                              // 2: prev[parity] holds positions in the cell-ordered cloud (striped path)
    int W, H;
    int parity;               // (rounds 2-4: which seed image the next cell-path frame reads; now the host's frame counter decides)
};
constexpr int A_BANDS = 8;    // list A of a strip is kept in depth bands, nearest first
struct StripCounters {        // 256 bytes per strip, at HEADER_STRIPS_OFFSET + 256 * strip
    int nA[A_BANDS], nB;      // list lengths, written by the classification blocks of the seed launch (agent-scope atomics)
    int pad0[64 - A_BANDS - 1];
};
constexpr size_t HEADER_BYTES = 8192;
constexpr size_t HEADER_STATS_OFFSET = 64;      // 16 x u64 debug counters (read_tuning_set("splat_stats", 1))
constexpr size_t HEADER_STRIPS_OFFSET = 256;    // 2 sets x MAX_STRIPS x StripCounters
// Two SETS of list counters (and two bound images, WsLayout::zimg): frame k of a workspace works in set k & 1.  A frame's
// resolve leaves ITS set clean (counters zero, bounds "none"), so the OTHER set is clean while a frame runs — which is what lets
// the resolve launch of frame k already classify the chunks and seed the bounds of frame k + 1 (cells_resolve_next_kernel) when
// the caller has announced the next camera (read_splat_hint_next_camera).
constexpr size_t COUNTER_SET_BYTES = MAX_STRIPS * 256;
static_assert(sizeof(StripCounters) == 256, "StripCounters must be two 128-byte lines");
static_assert(HEADER_STRIPS_OFFSET + 2 * COUNTER_SET_BYTES <= HEADER_BYTES, "header too small");

__device__ __forceinline__ StripCounters *strip_counters(void *hdr, int set, int s)
{
    return reinterpret_cast<StripCounters *>((char *)hdr + HEADER_STRIPS_OFFSET + (size_t)set * COUNTER_SET_BYTES) + s;
}

// bound[block] = max over the block's pixels of the current depth, +inf if any pixel is still empty.
__global__ __launch_bounds__(256) void splat_hiz_kernel(const unsigned long long *__restrict__ keys, int W, int H,
                                                        int nbx, int nby, unsigned short *__restrict__ hiz)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbx * nby) return;
    const int bx = b % nbx, by = b / nbx;
    unsigned m = 0;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int x = bx * 4 + dx, y = by * 4 + dy;
            if (x < W && y < H) {
                const unsigned bits = (unsigned)(keys[(long long)y * W + x] >> 32);    // EMPTY -> 0xffffffff
                m = bits > m ? bits : m;
            }
        }
    hiz[b] = hiz_encode(m);
}

// The point pass of MODE_HIZ (plain path): 1024-thread workgroups (persistent, grid-stride) with the whole bound
// image in LDS.
__global__ __launch_bounds__(1024) void splat_project_hiz_kernel(const float *__restrict__ xyz, long long n, CamSet cams,
                                                                 int W, int H, unsigned long long *__restrict__ keys,
                                                                 int vec_ok, const unsigned short *__restrict__ hiz_g, int nbx,
                                                                 int nblocks, int sub_mod, unsigned long long *stats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short hiz[];
    unsigned st_local[3] = {0, 0, 0};
    unsigned *stp = stats ? st_local : nullptr;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) hiz[i] = hiz_g[i];
    __syncthreads();
    const long long groups = n / PTS_PER_THREAD;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    if (vec_ok) {
        const float4 *xyz4 = reinterpret_cast<const float4 *>(xyz);
        const long long npass = pass_groups(groups, sub_mod, sub_mod > 0 ? 2 : 0);
        for (long long t = tid0; t < npass; t += nthreads) {
            const long long g = map_group(t, sub_mod, sub_mod > 0 ? 2 : 0);   // bootstrap chunks are already folded in
            if (g >= groups) continue;
            const float4 a = xyz4[3 * g + 0];
            const float4 b = xyz4[3 * g + 1];
            const float4 c = xyz4[3 * g + 2];
            const float px[4] = {a.x, a.w, b.z, c.y};
            const float py[4] = {a.y, b.x, b.w, c.z};
            const float pz[4] = {a.z, b.y, c.x, c.w};
            splat_points<true, 4>(px, py, pz, (unsigned)(g * PTS_PER_THREAD), 4, cams.m[0], W, H, keys, hiz, nbx, stp);
        }
    }
    const long long first = vec_ok ? groups * PTS_PER_THREAD : 0;
    for (long long i = first + tid0; i < n; i += nthreads) {
        const float px[1] = {xyz[3 * i + 0]}, py[1] = {xyz[3 * i + 1]}, pz[1] = {xyz[3 * i + 2]};
        splat_points<true, 1>(px, py, pz, (unsigned)i, 1, cams.m[0], W, H, keys, hiz, nbx, stp);
    }
    if (stats)
        for (int i = 0; i < 3; ++i) atomicAdd(stats + 4 + i, (unsigned long long)st_local[i]);
}

// Plain-path warm start: re-project the previous frame's winners (one per level-0 pixel) with the new camera.
__global__ __launch_bounds__(256) void splat_seed_kernel(const float *__restrict__ xyz, long long n, CamSet cams,
                                                         int W, int H, unsigned long long *__restrict__ keys,
                                                         const SplatHeader *hdr, const int *__restrict__ prev_idx)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H || !prev_idx) return;
    if (!(hdr->valid == 1 && hdr->W == W && hdr->H == H)) return;
    const int id = prev_idx[p];
    if (id < 0 || id >= n) return;
    float d;
    int xx, yy;
    const int pix = project_one(xyz[3ll * id], xyz[3ll * id + 1], xyz[3ll * id + 2], cams.m[0], W, H, d, xx, yy);
    if (pix >= 0) fold_key_agent(keys + pix, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id);
}


// ---- cell-ordered cloud, XCD-striped passes -------------------------------------------------------------------------
// read_splat_cells_build_host() sorts the cloud once along a Morton curve and cuts it into chunks of 1024 points with
// their bounding boxes; every record is (x, y, z, original id), so keys carry the ORIGINAL id and the result is
// bit-identical to the unsorted pass (atomic min does not care about order).  Per frame, five launches — FOUR when the caller
// announced this frame's camera one frame ahead (read_splat_hint_next_camera): the previous frame's resolve launch then already
// did this frame's seeds and classification (cells_resolve_next_kernel), in the OTHER of the two sets of per-frame state:
//   cells_seed_classify_kernel
//       seed blocks      re-project last frame's front points (their POSITIONS in the sorted cloud were left in a
//                        per-pixel image by the threads that issued atomics — any real point is a valid seed, so that
//                        image may be written racily) and store their depths into zimg.  No atomics.
//       classify blocks  one thread per chunk, from the 8 projected corners of its box: outside the frustum — or wholly behind
//                        the camera plane, round 5 — -> dropped (no point of it is read); nearest corner closer than w_split, a
//                        chunk that held a front point lately (sticky), or every sub-th chunk on a first frame -> list A;
//                        the rest -> list B with its screen rectangle and depth threshold.  A chunk is appended to the
//                        list of the strip that holds the centre column of its rectangle (block-aggregated appends).
//   cells_pass_kernel<A> workgroup b works on strip b % ns and walks that strip's list A statically: zimg early-z, LDS
//                        table for dense chunks, then the candidate goes to its screen tile's bin (emit_binned; or an atomic
//                        min on the key with splat_bins = 0) + plain stores of the new bound and of the point's position
//                        (next frame's seed).
//   cells_merge_hiz_kernel  one workgroup per 32x32 tile: bins folded into the tile's keys in LDS, written back; far bound
//                        per 4x4 block from the (now exact) keys; zimg := exact current depths.  (cells_hiz_kernel without bins.)
//   cells_pass_kernel<B> same walk over list B: a chunk is skipped when its nearest possible depth is behind the bound
//                        of EVERY block of its rectangle, otherwise its points run as in pass A.
//   splat_resolve_kernel levels, keys back to EMPTY, this frame's zimg back to "no bound", its counters to zero
//                        (cells_resolve_next_kernel: plus the next frame's classify and seed blocks, for the announced camera).
// Conservative arithmetic: the fp32 projection of a point and of the box corners differ by rounding; with
// S_k = sum_j |M_kj| max|box_j| + |M_k3| every computed clip coordinate is within gamma S_k of the exact one, so ndc
// errors are bounded by gamma (S_k + S_3) / w_min + ulp; the rectangle is widened and the depth test tightened by that
// much (gamma = 1e-6, >= 4x the worst case of a 4-term fp32 dot product).  A box with a corner at or behind the camera
// plane is never culled (list A, all strips).
struct CellHeader {            // first 256 bytes of the cell-ordered cloud (device and host)
    long long n;
    int nchunks, version;
    float bbox[6];
    float density;             // points per unit volume of the bounding box
};
constexpr int CELL_CHUNK = 1024;

constexpr int CELL_VERSION = 2;
constexpr size_t CELL_HEADER_BYTES = 256;

struct CellEntryB {            // 16 bytes per list-B chunk
    int chunk;
    unsigned bx;               // bx0 << 16 | bx1 (4x4-pixel block columns, inclusive)
    unsigned by;
    float e_thr;               // cull iff e_thr < min over the rectangle of (1 - far bound)
};

struct CellCloud {             // device pointers into the blob
    const CellHeader *hdr;
    const float4 *pts;         // nchunks * 1024 records (x, y, z, bits of the original id), Morton order, tail padded
                               // with copies of the last point
    const float *aabb;         // nchunks * 8: min xyz, max xyz, 2 pad
    int *list_a;               // scratch: MAX_STRIPS x A_BANDS x nchunks chunk ids of this frame's lists A (by depth band)
    CellEntryB *list_b;        // scratch: MAX_STRIPS x nchunks
    unsigned char *sticky;     // scratch: per chunk, frames for which a chunk that produced a candidate stays in list A
    int nchunks;
    int sticky_frames;         // what a chunk's counter is set to when pass B finds it in front of the bounds (splat_sticky; 0: never promoted)
    int mark_candidates;       // splat_mark: ANY chunk one of whose points reaches a bound gets its counter set (not only pass B's survivors)
    long long hdr_n;           // points of the cloud (host bookkeeping: a prediction belongs to one cloud)
};

struct StripInfo {
    int ns;                    // strips in use (<= MAX_STRIPS)
    int xb[MAX_STRIPS + 1];    // strip s = pixel columns [xb[s], xb[s+1]), multiples of 16; XCD x starts with strip x % ns
};

__device__ __forceinline__ int strip_of_column(const StripInfo &si, int x)
{
    int s = 0;
#pragma unroll
    for (int k = 1; k < MAX_STRIPS; ++k)
        if (k < si.ns && x >= si.xb[k]) s = k;
    return s;
}

// class of one chunk for camera M: 0 dropped, 1 list A, 2 list B (then e fills in); [cx0, cx1] = pixel columns
__device__ __forceinline__ int classify_chunk(const float *bb, const float *M, int W, int H, float w_split, bool boot,
                                              CellEntryB &e, int &cx0, int &cx1, float &wmin_out, int &area_out)
{
    constexpr float GAMMA = 1e-6f;
    const float mn[3] = {bb[0], bb[1], bb[2]}, mx[3] = {bb[3], bb[4], bb[5]};
    const float ax = fmaxf(fabsf(mn[0]), fabsf(mx[0])), ay = fmaxf(fabsf(mn[1]), fabsf(mx[1])),
                az = fmaxf(fabsf(mn[2]), fabsf(mx[2]));
    float S[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        S[k] = fabsf(M[4 * k]) * ax + fabsf(M[4 * k + 1]) * ay + fabsf(M[4 * k + 2]) * az + fabsf(M[4 * k + 3]);
    float c[8][4];
    float wmin = 3.0e38f, wmax = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float cx = (i & 1) ? mx[0] : mn[0], cy = (i & 2) ? mx[1] : mn[1], cz = (i & 4) ? mx[2] : mn[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[i][k] = M[4 * k] * cx + M[4 * k + 1] * cy + M[4 * k + 2] * cz + M[4 * k + 3] * 1.0f;
        wmin = fminf(wmin, c[i][3]);
        wmax = fmaxf(wmax, c[i][3]);
    }
    cx0 = 0;
    cx1 = W - 1;
    area_out = W * H;
    wmin_out = wmin > 0.0f ? wmin : 0.0f;
    // A box wholly on ONE side of the plane w = 0 is mapped by x -> clip(x) / w(x) onto the convex hull of its corners' images,
    // whichever side it is: the tests below only use the ratios and |w|.  Round 5: the boxes BEHIND the camera plane (every
    // corner w < 0) therefore take the same path with |w| = -w — with READ's projection their z / w is > 1 and they are dropped
    // here.  Rounds 2-4 sent every box with a corner at w <= 0 to list A unexamined: with the camera INSIDE the cloud that is
    // everything behind it (at the end of the benchmark sweep 64 % of the cloud, walked point by point by pass A for nothing).
    // Only a box that STRADDLES the plane is never culled.
    const float w_eps = fmaxf(1e-3f, 1e-5f * S[3]);
    if (wmax < -w_eps) {
        wmin = -wmax;                                             // the smallest |w| of the box
        wmin_out = wmin;
    } else if (!(wmin > w_eps))
        return 1;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float inv = 1.0f / c[i][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = c[i][k] * inv;
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
    }
    const float rw = GAMMA / wmin;
    const float ex = rw * (S[0] + S[3]) + 4e-7f, ey = rw * (S[1] + S[3]) + 4e-7f, ez = rw * (S[2] + S[3]) + 4e-7f;
    if (hi[0] < -1.0f - ex || lo[0] > 1.0f + ex || hi[1] < -1.0f - ey || lo[1] > 1.0f + ey || hi[2] < -1.0f - ez ||
        lo[2] > 1.0f + ez)
        return 0;
    const float px0 = (float)W * (lo[0] + 1.0f) * 0.5f - ((float)W * 0.5f * ex + 1.0f);
    const float px1 = (float)W * (hi[0] + 1.0f) * 0.5f + ((float)W * 0.5f * ex + 1.0f);
    const float py0 = (float)H * (1.0f - hi[1]) * 0.5f - ((float)H * 0.5f * ey + 1.0f);
    const float py1 = (float)H * (1.0f - lo[1]) * 0.5f + ((float)H * 0.5f * ey + 1.0f);
    const unsigned ux0 = (unsigned)fminf(fmaxf(px0, 0.0f), (float)(W - 1));
    const unsigned ux1 = (unsigned)fminf(fmaxf(px1, 0.0f), (float)(W - 1));
    const unsigned uy0 = (unsigned)fminf(fmaxf(py0, 0.0f), (float)(H - 1));
    const unsigned uy1 = (unsigned)fminf(fmaxf(py1, 0.0f), (float)(H - 1));
    cx0 = (int)ux0;
    cx1 = (int)ux1;
    area_out = (int)((ux1 - ux0 + 1u) * (uy1 - uy0 + 1u));
    if (wmin < w_split || boot) return 1;
    const float dmin = (fmaxf(lo[2], -1.0f) + 1.0f) * 0.5f;
    e.e_thr = (1.0f - dmin) + 2.0f * (0.5f * ez + 2e-7f);
    e.bx = (ux0 >> 2) << 16 | (ux1 >> 2);
    e.by = (uy0 >> 2) << 16 | (uy1 >> 2);
    return 2;
}

// The classification blocks of cells_seed_classify_kernel: one thread per chunk, block-aggregated appends (one atomic
// per block, list and strip: a per-wave append measured ~8 us of same-address atomics on the critical path).
__device__ __forceinline__ void classify_block(const CellCloud &cc, const float *M, int W, int H, int sub, float near_count,
                                               int block, void *hdr, int cset, const StripInfo &si)
{
    constexpr int PER_STRIP = A_BANDS + 1, LISTS = MAX_STRIPS * PER_STRIP;      // per strip: the bands of list A, then list B
    __shared__ int s_cnt[LISTS];
    __shared__ int s_base[LISTS];
    if (threadIdx.x < LISTS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int chunk = block * 256 + threadIdx.x;
    // list A takes the chunks nearer than the distance within which a pixel expects `near_count` points
    const float focal = sqrtf(M[0] * M[0] + M[1] * M[1] + M[2] * M[2]) * (float)W * 0.5f;
    const float w_split = cbrtf(3.0f * near_count * focal * focal / fmaxf(cc.hdr->density, 1e-20f));
    CellEntryB e;
    e.chunk = chunk;
    e.bx = e.by = 0;
    e.e_thr = 0.0f;
    int cls = 0, cx0 = 0, cx1 = 0, area = 0;
    float wmin = 0.0f;
    const bool boot = sub > 0 && chunk % sub == 0;
    if (chunk < cc.nchunks) cls = classify_chunk(cc.aabb + (size_t)chunk * 8, M, W, H, w_split, boot, e, cx0, cx1, wmin, area);
    // Dense chunk: more points than pixels in its rectangle — its points share pixels, so the passes fold them in the LDS
    // table first (strip_points); a sparse chunk's candidates each own a pixel and go to memory directly.  Bit 31 of the entry.
    const int dense_bit = area < CELL_CHUNK ? (int)0x80000000u : 0;
    // A chunk belongs to ONE strip — the one that holds the centre column of its rectangle — and all of its points are
    // processed there; the few that fall into a neighbouring strip read and write that strip's part of zimg from the "wrong"
    // XCD (a staler bound, never a wrong result).  Listing a chunk in every strip it touches kept the bounds exact but read
    // near chunks twice (pass A fetched 287 MB per frame for 93 MB of records).
    const int strip = strip_of_column(si, (cx0 + cx1) >> 1);
    // Depth band of a list-A chunk: the waves of a strip walk the bands nearest first, so a pixel's nearest candidates
    // arrive first and the farther ones fail the bound instead of each costing a memory-side atomic (pass A's time is the
    // number of atomics).  Chunk counts grow with the cube of the distance: equal-population bands at t^3.
    // A list-B chunk that survived the bound test is processed by ONE wave at the tail of pass B (the 26 survivors of the
    // benchmark scene cost ~8 us that way).  Survivors are stable from frame to frame, so pass B marks them and the next
    // splat_sticky classifications put them into list A, where they are spread over the whole grid; then they are tested again.
    if (cls == 2 && cc.sticky[chunk]) {
        cc.sticky[chunk] -= 1;
        cls = 1;
    }
    const float t = fminf(wmin / fmaxf(w_split, 1e-20f), 1.0f);
    const int band = wmin >= w_split ? A_BANDS - 1 : min(A_BANDS - 1, (int)((float)A_BANDS * t * t * t));
    const int l = strip * PER_STRIP + (cls == 1 ? band : A_BANDS);
    const int mine = cls ? atomicAdd(&s_cnt[l], 1) : 0;            // position inside the block's share (LDS atomic)
    __syncthreads();
    if (threadIdx.x < LISTS) {
        const int s = threadIdx.x / PER_STRIP, k = threadIdx.x % PER_STRIP;
        const int tot = s_cnt[threadIdx.x];
        StripCounters *sc = strip_counters(hdr, cset, s);
        s_base[threadIdx.x] = tot ? atomicAdd(k < A_BANDS ? &sc->nA[k] : &sc->nB, tot) : 0;     // one atomic per block and list
    }
    __syncthreads();
    e.chunk = chunk | dense_bit;
    if (cls == 1)
        cc.list_a[((size_t)strip * A_BANDS + band) * cc.nchunks + s_base[l] + mine] = chunk | dense_bit;
    else if (cls == 2)
        cc.list_b[(size_t)strip * cc.nchunks + s_base[l] + mine] = e;
}

// One seed block: 256 pixels of the seed image `pos` (positions, in the cell-ordered cloud, of the front points a previous frame's
// passes found) are re-projected with THIS camera and their depths stored as bounds.
__device__ __forceinline__ void seed_block(const CellCloud &cc, const float *M, int W, int H, unsigned *zimg, const int *pos_img,
                                           int block)
{
    const int p = block * 256 + (int)threadIdx.x;
    if (p >= W * H) return;
    const int pos = pos_img[p];
    if ((unsigned)pos >= (unsigned)cc.nchunks * CELL_CHUNK) return;
    const float4 q = cc.pts[pos];
    float d;
    int xx, yy;
    const int pix = project_one(q.x, q.y, q.z, M, W, H, d, xx, yy);
    // the depth of a real point at this pixel bounds the final depth from above; the point itself is folded in when its
    // chunk comes by (ties pass every test).  Colliding seeds race; either value is a valid bound.  (Also storing the
    // seed's KEY with a plain store, so that the seed point can skip its atomic after reading its own key back, was
    // measured slower: the dependent key reads cost more than the ~200 K atomics they saved.)
    if (pix >= 0) zimg[pix] = __float_as_uint(d);
}

__global__ __launch_bounds__(256) void cells_seed_classify_kernel(CellCloud cc, Cam1 cam, int W, int H,
                                                                  unsigned *zimg, void *hdr_v, const int *pos_img, int cset,
                                                                  StripInfo si, int seed_blocks,
                                                                  int sub, float near_count, int use_seeds)
{
    if ((int)blockIdx.x >= seed_blocks) {
        classify_block(cc, cam.m, W, H, sub, near_count, (int)blockIdx.x - seed_blocks, hdr_v, cset, si);
        return;
    }
    const SplatHeader *hdr = (const SplatHeader *)hdr_v;
    if (!use_seeds || !(hdr->valid == 2 && hdr->W == W && hdr->H == H)) return;
    seed_block(cc, cam.m, W, H, zimg, pos_img, (int)blockIdx.x);
}

// `rounds` x 256 consecutive points of one chunk for one wave and one strip: zimg early-z, then atomic min on the key +
// plain stores of the new bound and of the point's position (next frame's seed).  The records of round r+1 are loaded
// before round r is processed (one HBM round trip per chunk instead of one per round on the critical path).
//
// LDS (template flag): the candidates of a round are first folded into a 256-slot hash table in LDS that belongs to the
// wave (slot = hash(pixel); ds_cmpst claims it, ds_min_u64 keeps the smallest key), and only the table's survivors go to
// the memory-side atomics.  The 256 points of a round are Morton neighbours: on a densely sampled surface they share a
// handful of pixels and differ in depth by the sampling noise, so most of them beat the bound — and each other — and
// without the table every one of them costs a 15 G/s memory-side atomic (measured on the street scene: 9 atomics per
// covered pixel).  On a sparse cloud nearly every candidate keeps its own slot and the table is a few LDS operations of
// overhead.  Candidates that find their probe slots taken by other pixels go to memory directly.
constexpr int LDS_SLOTS = 256;       // per wave

// ---- pass A without data-path atomics: candidates are binned by 32x32-pixel screen tile ------------------------------
// Pass A was bound by the NUMBER of memory-side atomics (0.75 M x ~65 ps = 49 of its 54 us, profiles/README.md).  With
// bins a candidate (a point that beat the bound image) becomes a 16-byte record (pixel inside the tile, key) appended
// to its tile's bin; cells_merge_hiz_kernel — one workgroup per tile, the tile's keys in LDS — then takes the minimum
// per pixel with LDS atomics and writes the tile back with plain stores.  min() does not care about order or grouping, so
// the key image after the merge is the one the atomics would have produced.  Slots are reserved per wave and tile: the
// candidates of a wave's 256-point item fall into a handful of tiles, each (tile, count) pair costs ONE returning
// atomic on the tile's counter, all of them issued together (one memory round trip per item).  A full bin (or more than
// 64 distinct (pass, tile) pairs in one item) falls back to the atomic on the key image, which the merge reads first.
// Counters are kept MINUS ONE: read_splat_workspace_init() fills the workspace with ones, which then means "empty".
constexpr int BIN_TILE = 32;
constexpr int BIN_SUB = 32;    // sub-bins per tile, chosen by the wave's index: same-address atomics serialise at ~30 ns each
                               // (one counter per tile: pass A 96 us — near chunks of the same depth band hammer a few tiles)
static_assert(BIN_SUB * 8 == 256, "cells_merge_hiz_kernel: 8 threads per sub-bin");
struct BinInfo {
    uint4 *recs;               // tiles x BIN_SUB x cap records {pixel inside the tile, 0, key lo, key hi}; null = pass A with atomics
    unsigned *count;           // per (tile, sub-bin), minus one
    int cap, tiles_x;
    float inv_w;               // 1 / W (exact row of a pixel index: (pix + 0.5) * inv_w, W * H <= 2^20)
    int compact;               // splat_compact: a round's candidates are compacted into dense lanes before they are binned
};

// NC candidates per lane (valid bit k of `valid`): records into the bins; pos[k] = position of the point (next frame's seed)
template <int NC>
__device__ __forceinline__ void emit_binned(const BinInfo &bi, const int (&pix)[NC], const int (&px)[NC], const int (&py)[NC],
                                            const unsigned long long (&key)[NC], unsigned valid, unsigned long long *keys,
                                            int lane, unsigned *wl_tile, unsigned *wl_cnt, int sub)
{
    int tile[NC], inpix[NC], entry[NC], rank[NC];
    int nd = 0;                                                     // wave-uniform: (pass, tile) pairs so far
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // The candidates of an item are Morton neighbours: usually ONE tile.  Then the ranks are prefix counts over the NC
    // ballots and the item costs a single reservation, with no lists.
    {
        unsigned long long mk[NC];
        int t0 = -1;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const bool act = (valid >> k) & 1u;
            tile[k] = act ? ((py[k] >> 5) * bi.tiles_x + (px[k] >> 5)) * BIN_SUB + sub : -1;    // (tile, sub-bin of this wave)
            inpix[k] = (py[k] & 31) * BIN_TILE + (px[k] & 31);
            mk[k] = __ballot(act);
            if (t0 < 0 && mk[k]) t0 = __builtin_amdgcn_readlane(tile[k], __builtin_ctzll(mk[k]));
        }
        bool same = true;
#pragma unroll
        for (int k = 0; k < NC; ++k) same = same && __ballot(tile[k] >= 0 && tile[k] != t0) == 0ull;
        if (same) {
            int total = 0;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                rank[k] = total + __builtin_popcountll(mk[k] & lt_mask);
                total += __builtin_popcountll(mk[k]);
            }
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(bi.count + t0, (unsigned)total) + 1u;        // counters are kept minus one
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (!((valid >> k) & 1u)) continue;
                const unsigned slot = base + (unsigned)rank[k];
                if (slot < (unsigned)bi.cap)
                    bi.recs[(size_t)t0 * bi.cap + slot] = make_uint4((unsigned)inpix[k], 0u, (unsigned)key[k], (unsigned)(key[k] >> 32));
                else
                    __hip_atomic_fetch_min(keys + pix[k], key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const bool act = (valid >> k) & 1u;
        entry[k] = -1;
        rank[k] = 0;
        unsigned long long todo = __ballot(act);
        while (todo && nd < 64) {
            const int leader = __builtin_ctzll(todo);
            const int t = __builtin_amdgcn_readlane(tile[k], leader);
            const unsigned long long m = __ballot(act && tile[k] == t);
            if (act && tile[k] == t) {
                entry[k] = nd;
                rank[k] = __builtin_popcountll(m & lt_mask);
            }
            if (lane == 0) {
                wl_tile[nd] = (unsigned)t;
                wl_cnt[nd] = (unsigned)__builtin_popcountll(m);
            }
            ++nd;
            todo &= ~m;
        }
    }
    if (nd == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");          // the wave's own lists: LDS operations complete in order
    if (lane < nd) {
        const unsigned base = atomicAdd(bi.count + wl_tile[lane], wl_cnt[lane]) + 1u;     // counters are kept minus one
        wl_cnt[lane] = base;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (!((valid >> k) & 1u)) continue;
        const unsigned slot = entry[k] >= 0 ? wl_cnt[entry[k]] + (unsigned)rank[k] : 0xffffffffu;
        if (slot < (unsigned)bi.cap)
            bi.recs[(size_t)tile[k] * bi.cap + slot] = make_uint4((unsigned)inpix[k], 0u, (unsigned)key[k], (unsigned)(key[k] >> 32));
        else
            __hip_atomic_fetch_min(keys + pix[k], key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");          // the lists are reused by the next batch
}

template <bool STATS, bool ZL2, bool LDS, bool BIN>
__device__ __forceinline__ void strip_points(const CellCloud &cc, const float *M, int W, int H, int xlo, int xhi,
                                             unsigned long long *keys, unsigned *zimg, int *next, int first, int rounds,
                                             int lane, unsigned &st_in, unsigned &st_atomics, unsigned *tag,
                                             unsigned long long *hkey, int *hpos, const KeySlots ks, bool use_lds,
                                             const BinInfo &bi, unsigned *wl_tile, unsigned *wl_cnt, int sub,
                                             float4 (&q)[4], int next_first, uint4 *cq = nullptr)
{
    // q holds the first 256 records of this call (loaded by the caller: point_records); on return it holds the first 256 of
    // the caller's NEXT call (next_first, -1 = none) — their loads run under this call's bound reads and slot reservations
    float4 qn[4];
    for (int r = 0; r < rounds; ++r) {
        // 256 points: lane l takes records base + l + 64 k (each load instruction = 1 KiB contiguous)
        const int base = first + r * 256 + lane;
        if (r + 1 < rounds) {
#pragma unroll
            for (int k = 0; k < 4; ++k) qn[k] = cc.pts[base + 256 + 64 * k];
        } else if (next_first >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) qn[k] = cc.pts[next_first + lane + 64 * k];
        }
        int pix[4], px[4], py[4];
        unsigned dbits[4], bound[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d;
            pix[k] = project_one(q[k].x, q[k].y, q[k].z, M, W, H, d, px[k], py[k]);
            if (px[k] < xlo || px[k] >= xhi) pix[k] = -1;
            dbits[k] = __float_as_uint(d);
            if (STATS && pix[k] >= 0) st_in++;
        }
        // early-z against the bound image (L1 / this XCD's L2; a stale bound is only ever LARGER)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bound[k] = pix[k] < 0 ? 0u
                       : ZL2 ? __hip_atomic_load(zimg + pix[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   // sc1: L2, not L1
                             : zimg[pix[k]];
        // Candidates = points at or in front of their pixel's bound (ties pass: the key's id part breaks them).  Kept as
        // bit masks over the four points of a lane and handled in straight-line stages — the first version walked a
        // per-point if-chain through all of it and the compiler shuffled the four 64-bit keys between branch arms
        // (385 of the loop's 1370 VALU instructions were moves; the pass is VALU-issue bound).
        unsigned long long key[4];
        unsigned cand = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            key[k] = ((unsigned long long)dbits[k] << 32) | __float_as_uint(q[k].w);
            cand |= (pix[k] >= 0 && dbits[k] <= bound[k] ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (((cand >> k) & 1u) && dbits[k] < bound[k]) zimg[pix[k]] = dbits[k];
        // A chunk one of whose points reached a bound holds front points.  Beyond the near split (list B) such a chunk survives the
        // bound test of EVERY frame and is run by ONE workgroup at pass B's tail; marked, the next splat_sticky classifications list
        // it in A, where it is banded, binned and spread over the grid (surface scenes: far facades hold front points)
        if (cc.mark_candidates && __ballot(cand != 0u) && lane == 0) cc.sticky[(unsigned)(first + r * 256) / CELL_CHUNK] = (unsigned char)cc.sticky_frames;
        unsigned direct = cand;                                    // candidates that go to memory (bins / atomics) themselves
        if (LDS && use_lds) {                                      // wave-uniform: dense chunk, fold into the wave's table first
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((cand >> k) & 1u)) continue;
                unsigned h = ((unsigned)pix[k] * 2654435761u) >> 24;
                bool placed = false;
#pragma unroll
                for (int probe = 0; probe < 2 && !placed; ++probe, h = (h + 1) & (LDS_SLOTS - 1)) {
                    const unsigned old = atomicCAS(tag + h, 0u, (unsigned)pix[k] + 1u);
                    if (old == 0u || old == (unsigned)pix[k] + 1u) {
                        if (key[k] < atomicMin(hkey + h, key[k])) hpos[h] = base + 64 * k;
                        placed = true;
                    }
                }
                if (placed) direct &= ~(1u << k);
            }
        }
        // Round 5: the ~12 % of a round's 256 points that reach a bound are COMPACTED into dense lanes before they are binned (BIN):
        // the code behind the bound test then runs once for up to 64 candidates instead of four times, masked, over the four point
        // slots of every lane (pass A is bound by the number of vector instructions it issues).  Ranks are prefix
        // counts over the four ballots; the candidates travel through a wave-private queue in LDS (pixel, position, key).
        bool compacted = false;
        if (BIN && cq) {
            unsigned long long mk[4];
            int rank[4], tot = 0;
            const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mk[k] = __ballot((direct >> k) & 1u);
                rank[k] = tot + __builtin_popcountll(mk[k] & lt_mask);
                tot += __builtin_popcountll(mk[k]);
            }
            if (tot <= 64) {                                           // wave-uniform
                compacted = true;
                if (tot) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((direct >> k) & 1u)
                            cq[rank[k]] = make_uint4((unsigned)pix[k], (unsigned)(base + 64 * k), (unsigned)key[k], (unsigned)(key[k] >> 32));
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // the wave's own queue: LDS operations complete in order
                    const unsigned v = lane < tot ? 1u : 0u;
                    const uint4 c = cq[lane < tot ? lane : 0];
                    const int p1[1] = {(int)c.x};
                    int y1[1], x1[1];
                    y1[0] = (int)(((float)p1[0] + 0.5f) * bi.inv_w);           // exact for W * H <= 2^20 (as in the table flush below)
                    x1[0] = p1[0] - y1[0] * W;
                    const unsigned long long k1[1] = {((unsigned long long)c.w << 32) | c.z};
                    if (v) next[p1[0]] = (int)c.y;                             // a front point of this pixel: next frame's seed
                    if (STATS) st_atomics += v;
                    emit_binned<1>(bi, p1, x1, y1, k1, v, keys, lane, wl_tile, wl_cnt, sub);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // the queue is free again before the next round writes it
                }
            }
        }
        if (!compacted) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((direct >> k) & 1u) {
                    next[pix[k]] = base + 64 * k;                      // a front point of this pixel: next frame's seed
                    if (!BIN) __hip_atomic_fetch_min(keys + key_slot(ks, (unsigned)pix[k]), key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (STATS) st_atomics++;
                }
            if (BIN && __ballot(direct != 0u)) emit_binned<4>(bi, pix, px, py, key, direct, keys, lane, wl_tile, wl_cnt, sub);
        }
        if (LDS && use_lds) {
            // the wave's own table: its LDS operations complete in program order, no barrier needed
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            unsigned flush = 0;
#pragma unroll
            for (int j = 0; j < LDS_SLOTS / 64; ++j) {
                const int sl = lane + 64 * j;
                const unsigned t = tag[sl];
                pix[j] = (int)t - 1;
                key[j] = hkey[sl];
                py[j] = (int)(((float)pix[j] + 0.5f) * bi.inv_w);  // exact for W * H <= 2^20 (BIN only; unused otherwise)
                px[j] = pix[j] - py[j] * W;
                if (t) {
                    flush |= 1u << j;
                    if (!BIN) __hip_atomic_fetch_min(keys + key_slot(ks, t - 1u), key[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    next[t - 1u] = hpos[sl];
                    tag[sl] = 0u;
                    hkey[sl] = ~0ull;
                    if (STATS) st_atomics++;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (BIN && __ballot(flush != 0u)) emit_binned<4>(bi, pix, px, py, key, flush, keys, lane, wl_tile, wl_cnt, sub);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = qn[k];
    }
}

// Pass A: an item = 1024 / sub_items consecutive points of one list-A chunk, for one wave and one strip.
// Pass B: four list-B entries in flight per wave: the hi-Z bounds of their rectangles (inside the strip) are loaded together,
// then reduced; chunks that survive are processed like pass-A chunks.
template <bool PASS_B, bool STATS, bool ZL2, bool LDS, bool BIN>
__global__ __launch_bounds__(256) void cells_pass_kernel(CellCloud cc, Cam1 cam, int W, int H,
                                                         unsigned long long *keys, unsigned *zimg,
                                                         const unsigned short *__restrict__ hiz_g, int nbx, void *hdr_v,
                                                         int *next, int cset, StripInfo si, int sub_items,
                                                         unsigned long long *stats, KeySlots ks, BinInfo bi)
{
    __shared__ unsigned s_tag[LDS ? 4 * LDS_SLOTS : 1];
    __shared__ unsigned long long s_key[LDS ? 4 * LDS_SLOTS : 1];
    __shared__ int s_pos[LDS ? 4 * LDS_SLOTS : 1];
    __shared__ unsigned s_wl[BIN ? 4 * 128 : 1];                     // per wave: 64 tiles + 64 counts / bases (emit_binned)
    __shared__ uint4 s_cq[BIN ? 4 * 64 : 1];                         // per wave: the compacted candidates of a round (strip_points)
    __shared__ int s_surv[PASS_B ? 2 * 24 : 1];                      // pass B: the workgroup's surviving list entries of one round
    __shared__ int s_nsurv[2];
    unsigned *wl_tile = s_wl + (BIN ? (threadIdx.x >> 6) * 128 : 0), *wl_cnt = wl_tile + (BIN ? 64 : 0);
    uint4 *cq = (BIN && bi.compact) ? s_cq + (threadIdx.x >> 6) * 64 : nullptr;
    const float *M = cam.m;                                         // (`next`: the seed image this frame's front points go to)
    const int lane = threadIdx.x & 63;
    unsigned *tag = s_tag + (LDS ? (threadIdx.x >> 6) * LDS_SLOTS : 0);
    unsigned long long *hkey = s_key + (LDS ? (threadIdx.x >> 6) * LDS_SLOTS : 0);
    int *hpos = s_pos + (LDS ? (threadIdx.x >> 6) * LDS_SLOTS : 0);
    if (LDS) {
        for (int j = lane; j < LDS_SLOTS; j += 64) {
            tag[j] = 0u;
            hkey[j] = ~0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    unsigned st_in = 0, st_atomics = 0, n_run = 0, n_cull = 0;
    // Strip = blockIdx % ns: with the round-robin dispatch of workgroups over the XCDs (block b -> XCD b % 8, observed, not
    // promised) all work of a strip runs on the same XCDs and shares their L2 view of zimg; any other placement only makes
    // bounds staler.  Lists are walked statically — wave w of the strip takes entries w, w + n_waves, ... — because a
    // ticket counter per strip costs ~30 ns per draw (same-address atomics serialise memory-side: 20 K draws = 75 us).
    const int s = (int)(blockIdx.x % (unsigned)si.ns);
    const int wg_in_strip = (int)(blockIdx.x / (unsigned)si.ns), n_wg = (int)(gridDim.x / (unsigned)si.ns);
    if (wg_in_strip >= n_wg) return;
    const int n_waves = n_wg * (int)(blockDim.x >> 6);
    const int wave = __builtin_amdgcn_readfirstlane(wg_in_strip * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
    const StripCounters *sc = strip_counters(hdr_v, cset, s);
    const int xlo = 0, xhi = W;                                     // a chunk is processed whole by the strip that lists it
    float4 q[4];
    if (!PASS_B) {
        const int rounds = 4 / sub_items;
        int n_items = 0;
        for (int b = 0; b < A_BANDS; ++b) n_items += sc->nA[b] * sub_items;
        // Items are pipelined over three iterations of the walk: the list entry of item i+2 and the point records of item
        // i+1 are in flight while item i is tested — a wave's item is otherwise a chain of four dependent memory round trips
        // (entry, records, bounds, slot reservation) with nothing of its own to overlap them.
        int band = 0, band_first = 0, band_items = sc->nA[0] * sub_items;      // items [band_first, band_first + band_items)
        auto entry_of = [&](int t, int &part) {                                  // t only grows: the cursor moves forward
            while (t >= band_first + band_items) {
                band_first += band_items;
                ++band;
                band_items = sc->nA[band] * sub_items;
            }
            const int tl = t - band_first, li = tl / sub_items;
            part = tl - li * sub_items;
            return cc.list_a[((size_t)s * A_BANDS + band) * cc.nchunks + li];
        };
        int part0 = 0, part1 = 0, part2 = 0, e0 = 0, e1 = 0, e2 = 0;
        if (wave < n_items) e0 = entry_of(wave, part0);
        if (wave + n_waves < n_items) e1 = entry_of(wave + n_waves, part1);
        if (wave < n_items) {
            const int first0 = (__builtin_amdgcn_readfirstlane(e0) & 0x7fffffff) * CELL_CHUNK + part0 * rounds * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = cc.pts[first0 + lane + 64 * k];
        }
        for (int t = wave; t < n_items; t += n_waves) {
            if (t + 2 * n_waves < n_items) e2 = entry_of(t + 2 * n_waves, part2);
            const int entry = __builtin_amdgcn_readfirstlane(e0);
            const int chunk = entry & 0x7fffffff;
            const int next_first = t + n_waves < n_items
                                       ? (__builtin_amdgcn_readfirstlane(e1) & 0x7fffffff) * CELL_CHUNK + part1 * rounds * 256 : -1;
            ++n_run;
            strip_points<STATS, ZL2, LDS, BIN>(cc, M, W, H, xlo, xhi, keys, zimg, next, chunk * CELL_CHUNK + part0 * rounds * 256,
                                               rounds, lane, st_in, st_atomics, tag, hkey, hpos, ks, entry < 0, bi, wl_tile, wl_cnt,
                                               wave & (BIN_SUB - 1), q, next_first, cq);
            e0 = e1;
            part0 = part1;
            e1 = e2;
            part1 = part2;
        }
    } else {
        const CellEntryB *list_b = cc.list_b + (size_t)s * cc.nchunks;
        const int n_list = sc->nB;
        // Round = every wave tests up to three entries (2.3 on the benchmark scene): their records are fetched together, then each
        // rectangle's bounds (one 2-byte load per lane for rectangles of <= 64 blocks).  The survivors of a round (a few per
        // FRAME on the benchmark scene, none in most workgroups) go to a list of the workgroup and are then run by its four waves
        // TOGETHER, one 256-point quarter each: a survivor processed by the one wave that found it was four dependent rounds of
        // load -> project -> bound -> reserve -> store, ~8 us at the tail of a 13 us kernel.  The number of rounds is the same for
        // every wave of the grid (the barriers are uniform); the two survivor lists alternate so that a round's reset cannot race
        // with the next round's appends.
        const int wv = (int)(threadIdx.x >> 6);
        if (threadIdx.x < 2) s_nsurv[threadIdx.x] = 0;
        __syncthreads();
        // EPW entries per wave and round, ALL of them in flight together: the entries first, then the first 64 block bounds of
        // every rectangle (one 2-byte load per lane and entry; bigger rectangles loop on), then the reductions — a round is one
        // chain of two dependent memory round trips however many entries it holds, and the benchmark scene (3.6 entries per
        // wave) needs ONE round instead of the two it took at three entries per round.
        constexpr int EPW = 6;
        const int n_rounds = (n_list + EPW * n_waves - 1) / (EPW * n_waves);
        for (int it = 0; it < n_rounds; ++it) {
            const int t0 = wave + it * EPW * n_waves, par = it & 1;
            if (t0 < n_list) {
                CellEntryB e[EPW];
                bool have[EPW];
#pragma unroll
                for (int j = 0; j < EPW; ++j) {
                    const int t = t0 + j * n_waves;
                    have[j] = t < n_list;
                    e[j] = list_b[have[j] ? t : t0];
                }
                int bx0[EPW], by0[EPW], rw[EPW], nblk[EPW];
                float inv_rw[EPW], emin[EPW];
                bool run[EPW];
#pragma unroll
                for (int j = 0; j < EPW; ++j) {
                    bx0[j] = (int)(e[j].bx >> 16);
                    by0[j] = (int)(e[j].by >> 16);
                    rw[j] = (int)(e[j].bx & 0xffffu) - bx0[j] + 1;
                    nblk[j] = rw[j] * ((int)(e[j].by & 0xffffu) - by0[j] + 1);
                    run[j] = have[j] && rw[j] > 0;
                    inv_rw[j] = 1.0f / (float)(rw[j] > 0 ? rw[j] : 1);
                    emin[j] = 3.0e38f;                                 // min over the rectangle of (1 - far bound)
                    if (run[j] && nblk[j] <= 4096 && lane < nblk[j]) {
                        const int ry = (int)(((float)lane + 0.5f) * inv_rw[j]);   // i / rw for i < 4096 (exact: |error| << 0.5 / rw)
                        emin[j] = __uint_as_float((unsigned)hiz_g[(by0[j] + ry) * nbx + bx0[j] + lane - ry * rw[j]] << 16);
                    }
                }
                unsigned todo = 0;
#pragma unroll
                for (int j = 0; j < EPW; ++j) {
                    if (!have[j]) continue;
                    if (run[j] && nblk[j] <= 4096) {
                        for (int i = lane + 64; i < nblk[j]; i += 64) {
                            const int ry = (int)(((float)i + 0.5f) * inv_rw[j]);
                            emin[j] = fminf(emin[j], __uint_as_float((unsigned)hiz_g[(by0[j] + ry) * nbx + bx0[j] + i - ry * rw[j]] << 16));
                        }
                        float m = emin[j];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o));
                        run[j] = !(e[j].e_thr < m);                    // cull iff every point of the box is behind every bound
                    }
                    if (run[j]) todo |= 1u << j;
                    else ++n_cull;
                }
                todo = __builtin_amdgcn_readfirstlane(todo);
#pragma unroll
                for (int j = 0; j < EPW; ++j) {
                    if (!((todo >> j) & 1u)) continue;
                    const int entry = __builtin_amdgcn_readfirstlane(e[j].chunk);
                    if (lane == 0) {
                        cc.sticky[entry & 0x7fffffff] = (unsigned char)cc.sticky_frames;
                        s_surv[par * (4 * EPW) + atomicAdd(&s_nsurv[par], 1)] = entry;      // <= EPW per wave and round
                    }
                    ++n_run;
                }
            }
            __syncthreads();
            const int ns = s_nsurv[par];
            if (threadIdx.x == 0) s_nsurv[par ^ 1] = 0;                // the next round's list (nobody touches it before the barrier below)
            for (int item = wv; item < 4 * ns; item += 4) {
                const int entry = s_surv[par * 24 + (item >> 2)];
                const int first = (entry & 0x7fffffff) * CELL_CHUNK + (item & 3) * 256;
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = cc.pts[first + lane + 64 * k];
                strip_points<STATS, ZL2, LDS, BIN>(cc, M, W, H, xlo, xhi, keys, zimg, next, first, 1, lane, st_in, st_atomics, tag, hkey,
                                                   hpos, ks, entry < 0, bi, wl_tile, wl_cnt, wave & (BIN_SUB - 1), q, -1);
            }
            __syncthreads();
        }
    }
    if (STATS) {
        unsigned v[4] = {st_in, st_atomics, lane == 0 ? n_run : 0u, lane == 0 ? n_cull : 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
        if (lane == 0) {
            if (v[0]) atomicAdd(stats + (PASS_B ? 4 : 0), (unsigned long long)v[0]);
            if (v[1]) atomicAdd(stats + (PASS_B ? 6 : 2), (unsigned long long)v[1]);
            atomicAdd(stats + (PASS_B ? 11 : 8), (unsigned long long)v[2]);
            if (PASS_B) atomicAdd(stats + 9, (unsigned long long)v[3]);
        }
    }
}

// After pass A: bound[block] = max over the block's pixels of the current depth (from the exact key image), "none" if a
// pixel is still empty; and zimg is set to the exact current depths (the racy stores of pass A may have left a larger
// value than the minimum), so pass B's early-z is exact.
__global__ __launch_bounds__(256) void cells_hiz_kernel(const unsigned long long *__restrict__ keys, unsigned *__restrict__ zimg,
                                                        int W, int H, int nbx, int nby, unsigned short *__restrict__ hiz,
                                                        KeySlots ks)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbx * nby) return;
    const int bx = b % nbx, by = b / nbx;
    unsigned m = 0;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
        if (by * 4 + dy < H) {                                  // W % 16 == 0: the block's 4 columns exist
            const long long off = (long long)(by * 4 + dy) * W + bx * 4;
            uint4 z;
            if (ks.mode == 0) {
                const ulonglong2 k01 = *reinterpret_cast<const ulonglong2 *>(keys + off);
                const ulonglong2 k23 = *reinterpret_cast<const ulonglong2 *>(keys + off + 2);
                z.x = (unsigned)(k01.x >> 32);                  // EMPTY -> 0xffffffff = "none"
                z.y = (unsigned)(k01.y >> 32);
                z.z = (unsigned)(k23.x >> 32);
                z.w = (unsigned)(k23.y >> 32);
            } else {
                z.x = (unsigned)(keys[key_slot(ks, (unsigned)off)] >> 32);
                z.y = (unsigned)(keys[key_slot(ks, (unsigned)off + 1)] >> 32);
                z.z = (unsigned)(keys[key_slot(ks, (unsigned)off + 2)] >> 32);
                z.w = (unsigned)(keys[key_slot(ks, (unsigned)off + 3)] >> 32);
            }
            *reinterpret_cast<uint4 *>(zimg + off) = z;
            m = max(max(m, z.x), max(max(z.y, z.z), z.w));
        }
    hiz[b] = hiz_encode(m);
}

// Binned pass A (emit_binned): one workgroup per 32x32-pixel tile folds the tile's bin into its keys — LDS atomic min, then
// plain coalesced stores — and does cells_hiz_kernel's work for the tile's 8x8 blocks on the way out (zimg := exact depths,
// far bound per 4x4 block).  The tile's counter goes back to "empty" (minus one).
__global__ __launch_bounds__(256) void cells_merge_hiz_kernel(unsigned long long *__restrict__ keys, unsigned *__restrict__ zimg,
                                                              int W, int H, int nbx, unsigned short *__restrict__ hiz, BinInfo bi)
{
    __shared__ unsigned long long lk[BIN_TILE * BIN_TILE];
    const int t = threadIdx.x, tile = blockIdx.x;
    const int tx = tile % bi.tiles_x, ty = tile / bi.tiles_x;
    const int r = t >> 3, c4 = (t & 7) * 4;                         // thread = 4 consecutive pixels of tile row r
    const int y = ty * BIN_TILE + r, x = tx * BIN_TILE + c4;
    const bool in = y < H && x < W;                                 // W % 16 == 0: the four columns exist together
    const long long off = (long long)y * W + x;
    ulonglong2 k01 = make_ulonglong2(EMPTY_KEY, EMPTY_KEY), k23 = k01;
    if (in) {
        k01 = *reinterpret_cast<const ulonglong2 *>(keys + off);
        k23 = *reinterpret_cast<const ulonglong2 *>(keys + off + 2);
    }
    lk[r * BIN_TILE + c4] = k01.x;
    lk[r * BIN_TILE + c4 + 1] = k01.y;
    lk[r * BIN_TILE + c4 + 2] = k23.x;
    lk[r * BIN_TILE + c4 + 3] = k23.y;
    // 8 threads per sub-bin; the first four records of every thread are fetched before the tile is in LDS
    const int sb = tile * BIN_SUB + (t >> 3);
    unsigned n = bi.count[sb] + 1u;                                 // kept minus one
    n = n < (unsigned)bi.cap ? n : (unsigned)bi.cap;
    const uint4 *recs = bi.recs + (size_t)sb * bi.cap;
    uint4 rec[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned i = (t & 7) + 8 * j;
        rec[j] = recs[i < n ? i : 0];
    }
    __syncthreads();
    for (unsigned i0 = t & 7; i0 < n; i0 += 32) {
        uint4 nxt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = i0 + 32 + 8 * j;
            nxt[j] = recs[i < n ? i : 0];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + 8 * j < n) atomicMin(lk + rec[j].x, ((unsigned long long)rec[j].w << 32) | rec[j].z);
#pragma unroll
        for (int j = 0; j < 4; ++j) rec[j] = nxt[j];
    }
    __syncthreads();
    if ((t & 7) == 0) bi.count[sb] = 0xffffffffu;
    uint4 z;
    {
        const unsigned long long a = lk[r * BIN_TILE + c4], b = lk[r * BIN_TILE + c4 + 1], c = lk[r * BIN_TILE + c4 + 2],
                                 d = lk[r * BIN_TILE + c4 + 3];
        if (in) {
            *reinterpret_cast<ulonglong2 *>(keys + off) = make_ulonglong2(a, b);
            *reinterpret_cast<ulonglong2 *>(keys + off + 2) = make_ulonglong2(c, d);
        }
        z = make_uint4((unsigned)(a >> 32), (unsigned)(b >> 32), (unsigned)(c >> 32), (unsigned)(d >> 32));   // EMPTY -> "none"
    }
    if (in) *reinterpret_cast<uint4 *>(zimg + off) = z;
    // 4x4 block = this thread's 4 pixels in 4 consecutive rows: rows r, r^1, r^2, r^3 are lanes t, t^8, t^16, t^24
    unsigned m = in ? max(max(z.x, z.y), max(z.z, z.w)) : 0u;       // rows below the image do not count
    m = max(m, (unsigned)__shfl_xor((int)m, 8));
    m = max(m, (unsigned)__shfl_xor((int)m, 16));
    if ((r & 3) == 0 && in) hiz[((ty * BIN_TILE + r) >> 2) * nbx + ((tx * BIN_TILE + c4) >> 2)] = hiz_encode(m);
}

// ---- GL twin features: point sizes, "ps" splats, discard, clip-space perturbation ------------------------------------
// READ/gl/programs.py:121-198 (vertex shader) + READ/gl/render.py:52-85, canonical semantics as restated in
// oracle/raster.c (oracle_raster_level_gl): perturbed clip coordinates, centre clipping, a square of side
// s = point_size (or max(min_point_size, point_size / clip.z) for "ps" tokens, never below 1) covering pixel columns
// floor(u - (s-1)/2) .. floor(u + (s-1)/2) (rows likewise) of THE LEVEL IT IS DRAWN INTO — so every level is rasterised
// by its own pass (the 2x2 key-min pyramid only holds for 1-px points).  Augmentation path of the datasets
// (READ/datasets/dynamic.py:235-239), not the per-frame viewer path: one plain pass over the cloud per level.
struct GlOpts {
    float point_size;
    int relative;
    float min_point_size;
    const unsigned char *discard;
    unsigned drop_threshold, drop_seed;
    const float *perturb;
    float perturb_amp;
    unsigned perturb_seed;
    const float *point_sizes;    // per-point sizes (a_point_size), used when point_size < 1 (programs.py:183-187)
};

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned rnd32(unsigned i, unsigned seed, unsigned k)
{
    return hash32(i ^ hash32(seed + 0x9e3779b9u * (k + 1u)));
}

__global__ __launch_bounds__(256) void splat_project_gl_kernel(const float *__restrict__ xyz, long long n, Cam1 cam,
                                                               int W, int H, unsigned long long *__restrict__ keys, GlOpts o)
{
    const float *M = cam.m;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (o.discard && o.discard[i]) continue;
        if (o.drop_threshold && rnd32((unsigned)i, o.drop_seed, 0) < o.drop_threshold) continue;
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        float c0 = M[0] * x + M[1] * y + M[2] * z + M[3] * 1.0f;
        float c1 = M[4] * x + M[5] * y + M[6] * z + M[7] * 1.0f;
        const float c2 = M[8] * x + M[9] * y + M[10] * z + M[11] * 1.0f;
        const float c3 = M[12] * x + M[13] * y + M[14] * z + M[15] * 1.0f;
        if (o.perturb) {
            c0 = c0 + o.perturb[2 * i];
            c1 = c1 + o.perturb[2 * i + 1];
        }
        if (o.perturb_amp != 0.0f) {
            const float ux = (float)(rnd32((unsigned)i, o.perturb_seed, 1) >> 8) * (1.0f / 16777216.0f);
            const float uy = (float)(rnd32((unsigned)i, o.perturb_seed, 2) >> 8) * (1.0f / 16777216.0f);
            c0 = c0 + o.perturb_amp * (ux - 0.5f);
            c1 = c1 + o.perturb_amp * (uy - 0.5f);
        }
        const float nx = c0 / c3, ny = c1 / c3, nz = c2 / c3;
        const bool inside = (nx >= -1.0f) & (nx <= 1.0f) & (ny >= -1.0f) & (ny <= 1.0f) & (nz >= -1.0f) & (nz <= 1.0f);
        if (!inside) continue;
        const float u = ((float)W * (nx + 1.0f)) * 0.5f;
        const float v = ((float)H * (1.0f - ny)) * 0.5f;
        const float d = (nz + 1.0f) * 0.5f;
        if ((int)u < 0 || (int)u >= W || (int)v < 0 || (int)v >= H) continue;
        float sz = o.point_size;
        if (sz < 1.0f && o.point_sizes) sz = o.point_sizes[i];      // global_point_size 0 -> the vertex attribute
        if (o.relative) {
            sz = sz / c2;
            if (!(sz > o.min_point_size)) sz = o.min_point_size;
        }
        if (!(sz > 1.0f)) sz = 1.0f;
        if (sz > 4096.0f) sz = 4096.0f;
        const float half = 0.5f * (sz - 1.0f);
        const int x0 = max((int)floorf(u - half), 0), x1 = min((int)floorf(u + half), W - 1);
        const int y0 = max((int)floorf(v - half), 0), y1 = min((int)floorf(v + half), H - 1);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
                unsigned long long *k = keys + (long long)yy * W + xx;
                if (key < peek_key_agent(k)) fold_key_agent(k, key);
            }
    }
}

struct ResolveOut {
    int32_t *idx[READ_MAX_LEVELS];
    float *depth[READ_MAX_LEVELS];
};

__device__ __forceinline__ void emit(const ResolveOut &o, int level, long long off, unsigned long long key)
{
    const bool empty = key == EMPTY_KEY;
    if (o.idx[level]) o.idx[level][off] = empty ? 0 : (int32_t)(unsigned)(key & 0xffffffffull);
    if (o.depth[level]) o.depth[level][off] = empty ? 0.0f : __uint_as_float((unsigned)(key >> 32));
}

__device__ __forceinline__ unsigned long long kmin(unsigned long long a, unsigned long long b)
{
    return a < b ? a : b;
}

// One 256-thread block = a 32x32 tile of level 0; thread (qx,qy) owns a 2x2 quad.
// Levels 2..4 are reduced through LDS (16x16 -> 8x8 -> 4x4 -> 2x2 keys).
// keep: 0 nothing, 1 record the winners' ids (plain-path warm start), 2 striped frame (reset zimg and the strip
// counters, flip the seed-image parity).
__device__ __forceinline__ void resolve_tile(unsigned long long *__restrict__ keys, int W, int H,
                                             int levels, const ResolveOut &out, int tiles_x,
                                             int *__restrict__ prev_idx,
                                             void *hdr_v, int keep, unsigned *__restrict__ zimg, int cset, const KeySlots &ks,
                                             const int blk, const int cam)
{
    __shared__ unsigned long long s1[256], s2[64], s3[16];
    const int tx = blk % tiles_x, ty = blk / tiles_x;
    const int t = threadIdx.x;
    const int qx = t & 15, qy = t >> 4;
    const int x0 = tx * 32 + qx * 2, y0 = ty * 32 + qy * 2;
    const long long npx0 = (long long)W * H;
    unsigned long long *kc = keys + (long long)cam * npx0;

    unsigned long long k[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = x0 + dx, y = y0 + dy;
            unsigned long long v = EMPTY_KEY;
            if (x < W && y < H) {
                const long long off = (long long)y * W + x;
                const long long slot = keep == 2 ? (long long)key_slot(ks, (unsigned)off) : off;
                v = kc[slot];
                kc[slot] = EMPTY_KEY;                // leave the workspace clean for the next frame
                emit(out, 0, cam * npx0 + off, v);
                if (keep == 1) prev_idx[off] = v == EMPTY_KEY ? -1 : (int)(unsigned)(v & 0xffffffffull);   // next frame's seeds
                if (keep == 2) zimg[off] = 0xffffffffu;       // "no bound"
            }
            k[dy][dx] = v;
        }
    if (keep && blk == 0 && cam == 0) {
        SplatHeader *hdr = (SplatHeader *)hdr_v;
        if (keep == 2 && t < MAX_STRIPS) {
            StripCounters *sc = strip_counters(hdr_v, cset, t);       // this frame's set: clean again for the frame after next
            for (int b = 0; b < A_BANDS; ++b) sc->nA[b] = 0;
            sc->nB = 0;
        }
        if (t == 0) {
            // any integer below the padded point count is the position of a real point, i.e. a valid seed, so the two
            // seed images need no initialisation discipline: which of them a frame reads and which it writes follows the
            // host's frame counter (cells_frame), and reading the "wrong" one would only give staler seeds
            hdr->valid = keep;
            hdr->W = W;
            hdr->H = H;
        }
    }
    if (levels < 2) return;
    const unsigned long long k1 = kmin(kmin(k[0][0], k[0][1]), kmin(k[1][0], k[1][1]));
    {
        const int W1 = W >> 1, H1 = H >> 1, x = x0 >> 1, y = y0 >> 1;
        if (x < W1 && y < H1) emit(out, 1, (long long)cam * W1 * H1 + (long long)y * W1 + x, k1);
    }
    if (levels < 3) return;
    s1[t] = k1;
    __syncthreads();
    if (t < 64) {
        const int x2 = t & 7, y2 = t >> 3;
        const unsigned long long *r = s1 + (2 * y2) * 16 + 2 * x2;
        const unsigned long long k2 = kmin(kmin(r[0], r[1]), kmin(r[16], r[17]));
        s2[t] = k2;
        const int W2 = W >> 2, H2 = H >> 2, x = tx * 8 + x2, y = ty * 8 + y2;
        if (x < W2 && y < H2) emit(out, 2, (long long)cam * W2 * H2 + (long long)y * W2 + x, k2);
    }
    if (levels < 4) return;
    __syncthreads();
    if (t < 16) {
        const int x3 = t & 3, y3 = t >> 2;
        const unsigned long long *r = s2 + (2 * y3) * 8 + 2 * x3;
        const unsigned long long k3 = kmin(kmin(r[0], r[1]), kmin(r[8], r[9]));
        s3[t] = k3;
        const int W3 = W >> 3, H3 = H >> 3, x = tx * 4 + x3, y = ty * 4 + y3;
        if (x < W3 && y < H3) emit(out, 3, (long long)cam * W3 * H3 + (long long)y * W3 + x, k3);
    }
    if (levels < 5) return;
    __syncthreads();
    if (t < 4) {
        const int x4 = t & 1, y4 = t >> 1;
        const unsigned long long *r = s3 + (2 * y4) * 4 + 2 * x4;
        const unsigned long long k4 = kmin(kmin(r[0], r[1]), kmin(r[4], r[5]));
        const int W4 = W >> 4, H4 = H >> 4, x = tx * 2 + x4, y = ty * 2 + y4;
        if (x < W4 && y < H4) emit(out, 4, (long long)cam * W4 * H4 + (long long)y * W4 + x, k4);
    }
}


__global__ __launch_bounds__(256) void splat_resolve_kernel(unsigned long long *__restrict__ keys, int W, int H,
                                                            int levels, ResolveOut out, int tiles_x,
                                                            int tiles_y, int *__restrict__ prev_idx,
                                                            void *hdr_v, int keep, unsigned *__restrict__ zimg, int cset, KeySlots ks)
{
    resolve_tile(keys, W, H, levels, out, tiles_x, prev_idx, hdr_v, keep, zimg, cset, ks, (int)blockIdx.x, (int)blockIdx.y);
}

// The resolve launch of a cell-path frame whose caller has announced the NEXT camera (read_splat_hint_next_camera): next to the
// tiles of this frame (blocks [0, res_blocks)) the launch carries the classification blocks and the seed blocks of the next
// frame — cells_seed_classify_kernel's work, which depends on nothing this frame still has to produce: the chunk boxes are
// static, the seed image was completed by this frame's passes, and the next frame's counter set and bound image are the clean
// OTHER set (strip_counters).  One dependent launch (2.9 us of chain, tools/chain_probe.py) and ~5 us of a near-empty kernel
// less per frame; the 115 classification workgroups run beside the 418 tile workgroups instead of alone on the chip.
struct NextFrame {
    Cam1 cam;
    unsigned *zimg;          // the next frame's bound image (clean)
    const int *pos_img;      // the seed image this frame's passes wrote
    int cset, sub, use_seeds, class_blocks;
    float near_count;
};
__global__ __launch_bounds__(256) void cells_resolve_next_kernel(unsigned long long *__restrict__ keys, int W, int H,
                                                                 int levels, ResolveOut out, int tiles_x, int res_blocks,
                                                                 void *hdr_v, unsigned *__restrict__ zimg, int cset, KeySlots ks,
                                                                 CellCloud cc, NextFrame nx, StripInfo si)
{
    // dispatch order = block order: the classification blocks first — theirs is the longest dependent chain of the launch (box ->
    // class -> LDS count -> one atomic per list -> list write) —, then the tiles, then the seeds
    const int b = (int)blockIdx.x;
    if (b < nx.class_blocks) {
        classify_block(cc, nx.cam.m, W, H, nx.sub, nx.near_count, b, hdr_v, nx.cset, si);
        return;
    }
    if (b < nx.class_blocks + res_blocks) {
        resolve_tile(keys, W, H, levels, out, tiles_x, nullptr, hdr_v, 2, zimg, cset, ks, b - nx.class_blocks, 0);
        return;
    }
    if (nx.use_seeds) seed_block(cc, nx.cam.m, W, H, nx.zimg, nx.pos_img, b - res_blocks - nx.class_blocks);
}

__global__ __launch_bounds__(256) void fill_keys_kernel(unsigned long long *keys, long long count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) keys[i] = EMPTY_KEY;
}

__global__ __launch_bounds__(256) void index_to_float_kernel(const int32_t *__restrict__ idx, long long count,
                                                             float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (float)idx[i];
}

int level_dim(int v, int l)
{
    // int(v * 0.5**l)  (myrender.py:33-34); exact for the power-of-two scales used here
    return (int)((double)v * (1.0 / (double)(1 << l)));
}

int g_splat_mode = MODE_HIZ;
int g_splat_subset = 8;        // plain MODE_HIZ bootstrap pass over every g_splat_subset-th 1024-point chunk (0 = seeds only).
                               // Measured at 30 M points: 0.307 ms seeds only, 0.252 / 0.233 / 0.235 / 0.260 ms at 4 / 8 / 16 / 32
int g_splat_stats = 0;         // debug: accumulate counters in the workspace header (u64 at byte 64: pass A visible /
                               // early-z reads / atomics / -, pass B visible / after hi-z / atomics / -, A items run,
                               // B chunks culled, -, B items run)
int g_splat_near = 12;         // cell path: pass A takes chunks nearer than the depth at which a pixel expects this many points
int g_splat_cells = 1;         // 0: ignore the cell-ordered copy (A/B)
int g_splat_cells_sub = 0;     // list A also takes every n-th chunk (a first bound where nothing is near); 0: only on a workspace's first frame (every 32nd)
int g_splat_seeds = 1;         // 0: no warm start from the previous frame's front points (A/B)
int g_splat_items = 4;         // work items per chunk in the striped passes (1, 2 or 4): 0.101 / 0.101 / 0.097 ms per frame
int g_splat_zl2 = 0;            // 1: early-z loads bypass the L1 (sc1); measured slower (0.107 vs 0.101 ms)
int g_splat_kslot = 0;          // key-image layout of the striped path: 0 linear, 1 one key per 64 B, 2 scattered (key_slot):
                                // pass A 60.7 / 66.0 / 61.9 us — the atomics do not serialise on neighbouring lines, their NUMBER is the cost
int g_splat_lds = 1;            // 1: per-wave LDS hash table in front of the memory-side atomics (strip_points)
int g_splat_wgs = 4;            // workgroups per CU of the striped passes: 0.0996 / 0.0936 / 0.0893 / 0.0927 / 0.0923 ms at 2 / 3 / 4 / 6 / 8
                                // (fewer waves = more rounds per wave = finer front-to-back order over the depth bands)
int g_splat_compact = 1;        // 1: pass A compacts a round's candidates before binning them (strip_points); 0: four masked point slots per lane
int g_splat_cells_batch = 1;    // 1: a batch of cameras runs as B cell-path frames; 0: the plain pass over the whole cloud (rounds 1-4)
int g_splat_wgs_b = 0;          // workgroups per CU of pass B (0: as pass A, splat_wgs)
int g_splat_mark = 1;           // 1: every chunk one of whose points reaches a depth bound is listed in A for the next splat_sticky classifications
                                // (strip_points); 0: only the chunks pass B found in front of the bounds (rounds 2-4)
int g_splat_sticky = 1;         // classifications for which a marked list-B chunk is listed in A (0: never; rounds 2-4: 8, pass-B survivors only)
int g_splat_ahead = 1;          // 1: with an announced next camera (read_splat_hint_next_camera) a frame's resolve launch also classifies and
                                // seeds the next frame (cells_resolve_next_kernel): 4 dependent launches per frame instead of 5; 0: A/B
int g_splat_bins = 1;           // 1: pass A appends its candidates to per-tile bins, merged in LDS (emit_binned); 0: one memory-side atomic each
int g_splat_strips = 1;         // column strips of the striped passes (1, 2, 4 or 8).  8 was best while a strip's list was walked in
                               // Morton order (pass A 61.5 / 75.5 / 70 us at 8 / 2 / 1: fewer atomics with exact bounds); with the
                               // depth bands ONE global list wins — global front-to-back order and perfect balance: bench
                               // 0.0866 / 0.0867 / 0.0951 ms at 1 / 2 / 8 strips, street scene 0.099 vs 0.138 ms at 1 vs 8

// Workspace layout (fixed by the (B, W, H) it was sized for; one workspace serves one such triple):
//   [header 4096 B][key images: min(B,8) x W*H x 8 B][hi-z bounds: ceil(W/4)*ceil(H/4) x 4 B][seed image 0: W*H x 4 B]
//   [seed image 1: W*H x 4 B][zimg 0, zimg 1: W*H x 4 B each, depth upper bounds of the cell path, 0xffffffff = none]
struct WsLayout {
    void *hdr;
    unsigned long long *keys;
    unsigned short *hiz;       // 16-bit far bounds (the region keeps its 4 bytes per block)
    int *prev[2];              // plain path: prev[0] = winners' ids; striped path: positions, double buffered
    unsigned *zimg[2];         // cell path: depth upper bounds, one image per frame parity (set k & 1, see strip_counters)
    unsigned *bin_count;       // striped path, binned pass A: per 32x32 tile, minus one
    uint4 *bin_recs;           // tiles x bin_cap records
    int bin_cap, bin_tiles_x, bin_tiles;
    size_t key_slots;
    int nbx, nby;
    size_t total;
};

WsLayout ws_layout(void *ws, int B, int W, int H)
{
    WsLayout L;
    const int nb = B < MAX_CAMS ? B : MAX_CAMS;
    char *p = (char *)ws;
    L.hdr = p;
    size_t off = HEADER_BYTES;
    L.keys = (unsigned long long *)(p + off);
    L.key_slots = (size_t)(nb < 8 ? 8 : nb) * W * H;            // >= 8 W H: room for the scattered key layouts of the striped path
    off += L.key_slots * sizeof(unsigned long long);
    off = (off + 255) / 256 * 256;
    L.nbx = ceil_div(W, 4);
    L.nby = ceil_div(H, 4);
    L.hiz = (unsigned short *)(p + off);
    off += ((size_t)L.nbx * L.nby * sizeof(float) + 255) / 256 * 256;
    for (int i = 0; i < 2; ++i) {
        L.prev[i] = (int *)(p + off);
        off += ((size_t)W * H * sizeof(int) + 255) / 256 * 256;
    }
    for (int i = 0; i < 2; ++i) {
        L.zimg[i] = (unsigned *)(p + off);
        off += ((size_t)W * H * sizeof(unsigned) + 255) / 256 * 256;
    }
    // bins of the striped path: BIN_SUB sub-bins per tile, <= 64 MiB of records, 32..256 per sub-bin (a full one falls back
    // to the atomics)
    L.bin_tiles_x = ceil_div(W, BIN_TILE);
    L.bin_tiles = L.bin_tiles_x * ceil_div(H, BIN_TILE);
    const size_t fit = ((size_t)64 << 20) / sizeof(uint4) / ((size_t)L.bin_tiles * BIN_SUB);
    L.bin_cap = fit >= 256 ? 256 : (fit >= 32 ? (int)fit : 32);
    L.bin_count = (unsigned *)(p + off);
    off += ((size_t)L.bin_tiles * BIN_SUB * sizeof(unsigned) + 255) / 256 * 256;
    L.bin_recs = (uint4 *)(p + off);
    off += (size_t)L.bin_tiles * BIN_SUB * L.bin_cap * sizeof(uint4);
    L.total = off;
    return L;
}

constexpr size_t HIZ_LDS_LIMIT = 150 * 1024;   // of the 160 KiB per CU

int device_cus()
{
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n_cu;
}

ResolveOut resolve_out(int b0, int W, int H, int levels, int32_t *const *idx_levels, float *const *depth_levels, int level_base)
{
    ResolveOut out;
    memset(&out, 0, sizeof(out));
    for (int l = 0; l < levels; ++l) {
        const size_t lpx = (size_t)level_dim(W, l) * level_dim(H, l);
        if (idx_levels && idx_levels[level_base + l]) out.idx[l] = idx_levels[level_base + l] + lpx * b0;
        if (depth_levels && depth_levels[level_base + l]) out.depth[l] = depth_levels[level_base + l] + lpx * b0;
    }
    return out;
}

// cset: the frame's counter set / bound image (cell path, keep == 2)
int resolve_launch(unsigned long long *keys, int nb, int b0, int W, int H, int levels, int32_t *const *idx_levels,
                   float *const *depth_levels, int level_base, const WsLayout &ws, int keep, hipStream_t stream,
                   KeySlots ks = KeySlots{0, 0}, int cset = 0)
{
    const ResolveOut out = resolve_out(b0, W, H, levels, idx_levels, depth_levels, level_base);
    const int tiles_x = ceil_div(W, 32), tiles_y = ceil_div(H, 32);
    hipLaunchKernelGGL(splat_resolve_kernel, dim3(tiles_x * tiles_y, nb), dim3(256), 0, stream, keys, W, H,
                       levels, out, tiles_x, tiles_y, ws.prev[0], ws.hdr, keep, ws.zimg[cset], cset, ks);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

// ---- host-side frame state of a workspace -----------------------------------------------------------------------------
// Which counter set / bound image / seed image a cell-path frame uses follows a frame counter kept HERE, per workspace address
// (the workspace itself stays plain caller-owned device memory).  Nothing on the device depends on the parity being "right":
// both sets are clean between frames (a frame's resolve cleans its own), and either seed image holds positions of real points.
// The one state that must not be lost is a PREDICTION: cells_resolve_next_kernel has filled the other set for an announced
// camera — the next call on that workspace either consumes it (same camera, size and knobs) or wipes that set first.
struct WsHost {
    unsigned long long frame = 0;
    bool hinted = false;                 // read_splat_hint_next_camera since the last frame
    float hint[16];
    bool pred = false;                   // the set (frame & 1) holds the classification + seeds of camera `pm`
    float pm[16];
    int pW = 0, pH = 0, p_sub = 0, p_near = 0, p_ns = 0, p_seeds = 0;
    void *p_counters = nullptr, *p_zimg = nullptr;
    size_t p_zimg_bytes = 0;
    const void *p_cells = nullptr;       // ... for the lists in this cell blob, as of its frame count p_cells_frame
    unsigned long long p_cells_frame = 0;
    long long p_n = 0;                   // ... of a cloud of this many points
};
std::mutex g_ws_mutex;
std::unordered_map<void *, WsHost> g_ws_host;
// The chunk lists a classification fills live in the CELL BLOB (one blob serves one stream at a time), not in the workspace: a
// prediction is only good while no other frame — of any workspace — has run over the same blob.  Frames per blob, counted here.
// The count is ALSO what read_splat_cells_invalidate bumps: a blob that was rewritten, copied over or re-allocated at the address of
// an old one carries wiped or foreign lists — its owner says so, and every prediction made against the old contents stops matching.
// Counts never restart (an erased entry would let a stale prediction match frame count 0 again): g_cells_epoch seeds new entries.
std::unordered_map<const void *, unsigned long long> g_cells_frames;
unsigned long long g_cells_epoch = 1;
unsigned long long &cells_frames_of(const void *pts)
{
    auto it = g_cells_frames.find(pts);
    if (it == g_cells_frames.end()) it = g_cells_frames.emplace(pts, (g_cells_epoch++) << 40).first;
    return it->second;
}

// a pending prediction nobody will consume: wipe the set it dirtied (stream order puts this after the launch that filled it)
int ws_drop_prediction(WsHost &h, hipStream_t stream)
{
    if (!h.pred) return READ_OK;
    h.pred = false;
    READ_CHECK_HIP(hipMemsetAsync(h.p_counters, 0, COUNTER_SET_BYTES, stream));
    READ_CHECK_HIP(hipMemsetAsync(h.p_zimg, 0xff, h.p_zimg_bytes, stream));
    return READ_OK;
}

// ---- plain path ------------------------------------------------------------------------------------------------------
int project_and_resolve(const float *xyz, int64_t n, const float *M_host, int B, int W, int H, int levels,
                        int32_t *const *idx_levels, float *const *depth_levels, int level_base,
                        const WsLayout &ws, bool allow_hiz, hipStream_t stream)
{
    unsigned long long *keys = ws.keys;
    const size_t hiz_bytes = (((size_t)ws.nbx * ws.nby * sizeof(unsigned short)) + 15) & ~(size_t)15;
    const bool use_hiz = g_splat_mode == MODE_HIZ && allow_hiz && B == 1 && n > 0 && hiz_bytes <= HIZ_LDS_LIMIT &&
                         ((uintptr_t)xyz % 16) == 0;
    for (int b0 = 0; b0 < B; b0 += MAX_CAMS) {
        const int nb = (B - b0) < MAX_CAMS ? (B - b0) : MAX_CAMS;
        CamSet cams;
        memset(&cams, 0, sizeof(cams));
        memcpy(cams.m, M_host + 16 * (size_t)b0, sizeof(float) * 16 * (size_t)nb);
        const int vec_ok = ((uintptr_t)xyz % 16) == 0;
        unsigned long long *stats = g_splat_stats ? (unsigned long long *)((char *)ws.hdr + HEADER_STATS_OFFSET) : nullptr;
        if (use_hiz) {
            // bootstrap: a strided 1/sub of the cloud (only when the vector path is usable) on top of the seeds,
            // so that nearly every pixel is covered before the bounds are taken
            const int sub = (vec_ok && g_splat_subset > 1 && n >= (1 << 20)) ? g_splat_subset : 0;
            hipLaunchKernelGGL(splat_seed_kernel, dim3(ceil_div(W * H, 256)), dim3(256), 0, stream, xyz, (long long)n, cams,
                               W, H, keys, (const SplatHeader *)ws.hdr, g_splat_seeds ? ws.prev[0] : (int *)nullptr);
            READ_CHECK_LAUNCH();
            if (sub) {
                int64_t blocks = ceil_div64(ceil_div64(n, PTS_PER_THREAD), 256);
                if (blocks > 256 * 8) blocks = 256 * 8;
                hipLaunchKernelGGL(splat_project_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, xyz,
                                   (long long)n, cams, 1, W, H, keys, vec_ok, sub, stats);
                READ_CHECK_LAUNCH();
            }
            hipLaunchKernelGGL(splat_hiz_kernel, dim3(ceil_div(ws.nbx * ws.nby, 256)), dim3(256), 0, stream, keys, W, H,
                               ws.nbx, ws.nby, ws.hiz);
            READ_CHECK_LAUNCH();
            static bool attr_set = false;
            if (!attr_set) {
                READ_CHECK_HIP(hipFuncSetAttribute((const void *)splat_project_hiz_kernel,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIZ_LDS_LIMIT));
                attr_set = true;
            }
            const int n_cu = device_cus();
            int64_t blocks = ceil_div64(ceil_div64(n, PTS_PER_THREAD), 1024);
            const int per_cu = 2 * hiz_bytes <= 150 * 1024 ? 2 : 1;     // two workgroups per CU when their bounds fit
            if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
            hipLaunchKernelGGL(splat_project_hiz_kernel, dim3((unsigned)blocks), dim3(1024), hiz_bytes, stream, xyz,
                               (long long)n, cams, W, H, keys, vec_ok, ws.hiz, ws.nbx, ws.nbx * ws.nby, sub, stats);
            READ_CHECK_LAUNCH();
        } else if (n > 0) {
            const int64_t work = ceil_div64(n, PTS_PER_THREAD);
            // HBM-bound stream: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
            int64_t blocks = ceil_div64(work, 256);
            if (blocks > 256 * 8) blocks = 256 * 8;
            hipLaunchKernelGGL(splat_project_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, xyz, (long long)n, cams,
                               nb, W, H, keys, vec_ok, 0, stats);
            READ_CHECK_LAUNCH();
        }
        // the winners of a single-camera frame seed the next frame rendered through this workspace
        const bool keep = g_splat_mode == MODE_HIZ && allow_hiz && B == 1;
        const int rc = resolve_launch(keys, nb, b0, W, H, levels, idx_levels, depth_levels, level_base, ws, keep ? 1 : 0,
                                      stream);
        if (rc != READ_OK) return rc;
    }
    return READ_OK;
}

// ---- striped path ----------------------------------------------------------------------------------------------------
StripInfo make_strips(int W)
{
    StripInfo si;
    memset(&si, 0, sizeof(si));
    const int cols = W / 16;                                   // 128-byte lines of keys per image row
    si.ns = g_splat_strips < cols ? g_splat_strips : cols;
    for (int s = 0; s <= si.ns; ++s) si.xb[s] = 16 * (int)((long long)cols * s / si.ns);
    for (int s = si.ns + 1; s <= MAX_STRIPS; ++s) si.xb[s] = W;
    return si;
}

// ---- per-kernel durations of the LAST cell-path frame (read_tuning_set("splat_prof", 1) + read_splat_profile_last): HIP events
// on the launch stream around every launch of the frame.  Slots: 0 seeds + classification (0 when the previous frame's resolve
// launch did that work), 1 pass A, 2 bin merge + bounds, 3 pass B, 4 resolve (+ the next frame's seeds / classification).
int g_splat_prof = 0;
hipEvent_t g_prof_ev[6];
bool g_prof_made = false, g_prof_valid = false, g_prof_slot0 = false;
int prof_mark(int i, hipStream_t stream)
{
    if (!g_splat_prof) return READ_OK;
    if (!g_prof_made) {
        for (auto &e : g_prof_ev) READ_CHECK_HIP(hipEventCreate(&e));
        g_prof_made = true;
    }
    READ_CHECK_HIP(hipEventRecord(g_prof_ev[i], stream));
    return READ_OK;
}

int cells_frame(const CellCloud &cc, const float *M_host, int W, int H, int levels, int32_t *const *idx_levels,
                float *const *depth_levels, const WsLayout &ws, hipStream_t stream, int b0 = 0)
{
    const StripInfo si = make_strips(W);
    Cam1 cam;
    memcpy(cam.m, M_host, sizeof(cam.m));
    unsigned long long *stats = g_splat_stats ? (unsigned long long *)((char *)ws.hdr + HEADER_STATS_OFFSET) : nullptr;
    const int seed_blocks = ceil_div(W * H, 256), class_blocks = ceil_div(cc.nchunks, 256);
    // ---- this frame's set; was it prepared by the previous frame's resolve launch?
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    WsHost &h = g_ws_host[ws.hdr];
    const int fp = (int)(h.frame & 1);
    unsigned long long &blob_frames = cells_frames_of((const void *)cc.pts);
    // every n-th chunk joins list A as a first bound where nothing is near: on a workspace's FIRST frame (no seeds yet) when the knob
    // leaves it to us (splat_cells_sub 0), never afterwards — the seeds are that bound (lap 75.8 -> 75.1 us with splat_sticky 1)
    const int sub_now = g_splat_cells_sub > 0 ? g_splat_cells_sub : (h.frame == 0 ? 32 : 0);
    const int sub_next = g_splat_cells_sub > 0 ? g_splat_cells_sub : 0;
    const bool prepared = h.pred && h.pW == W && h.pH == H && memcmp(h.pm, M_host, sizeof(h.pm)) == 0 && h.p_sub == sub_now &&
                          h.p_near == g_splat_near && h.p_ns == si.ns && h.p_seeds == g_splat_seeds && h.p_zimg == (void *)ws.zimg[fp] &&
                          h.p_cells == (const void *)cc.pts && h.p_cells_frame == blob_frames && h.p_n == cc.hdr_n;
    blob_frames += 1;
    g_prof_valid = false;
    g_prof_slot0 = !prepared;
    if (prepared) h.pred = false;                     // consumed
    else {
        const int rc = ws_drop_prediction(h, stream);
        if (rc != READ_OK) return rc;
        if (prof_mark(0, stream) != READ_OK) return READ_EHIP;
        hipLaunchKernelGGL(cells_seed_classify_kernel, dim3((unsigned)(seed_blocks + class_blocks)), dim3(256), 0,
                           stream, cc, cam, W, H, ws.zimg[fp], ws.hdr, (const int *)ws.prev[fp], fp, si,
                           seed_blocks, sub_now, (float)g_splat_near, g_splat_seeds);
        READ_CHECK_LAUNCH();
    }
    int *const next_pos = ws.prev[fp ^ 1];            // the seed image this frame's passes write: the next frame's (set fp ^ 1) seeds
    const unsigned grid = (unsigned)(device_cus() * g_splat_wgs);
    const int items = g_splat_items;
    KeySlots ks;
    ks.mode = g_splat_kslot;
    ks.mask = 1;
    while (ks.mask < (unsigned)(W * H)) ks.mask <<= 1;
    ks.mask -= 1;
    if (ks.mode == 2 && (size_t)ks.mask + 1 > ws.key_slots) ks.mode = 0;      // (a workspace of this size always has 8 W H slots)
    // bins need the linear key layout and exact pixel rows from (pix + 0.5) / W in fp32
    const bool bins = g_splat_bins && ks.mode == 0 && (long long)W * H <= (1ll << 20);
    BinInfo bi;
    memset(&bi, 0, sizeof(bi));
    if (bins) {
        bi.recs = ws.bin_recs;
        bi.count = ws.bin_count;
        bi.cap = ws.bin_cap;
        bi.tiles_x = ws.bin_tiles_x;
        bi.inv_w = 1.0f / (float)W;
        bi.compact = g_splat_compact;
    }
    auto pass_a = bins ? (stats ? (g_splat_lds ? cells_pass_kernel<false, true, false, true, true> : cells_pass_kernel<false, true, false, false, true>)
                          : g_splat_zl2 ? cells_pass_kernel<false, false, true, false, true>
                          : g_splat_lds ? cells_pass_kernel<false, false, false, true, true> : cells_pass_kernel<false, false, false, false, true>)
                  : stats ? (g_splat_lds ? cells_pass_kernel<false, true, false, true, false> : cells_pass_kernel<false, true, false, false, false>)
                  : g_splat_zl2 ? cells_pass_kernel<false, false, true, false, false>
                  : g_splat_lds ? cells_pass_kernel<false, false, false, true, false> : cells_pass_kernel<false, false, false, false, false>;
    auto pass_b = stats ? (g_splat_lds ? cells_pass_kernel<true, true, false, true, false> : cells_pass_kernel<true, true, false, false, false>)
                  : g_splat_zl2 ? cells_pass_kernel<true, false, true, false, false>
                  : g_splat_lds ? cells_pass_kernel<true, false, false, true, false> : cells_pass_kernel<true, false, false, false, false>;
    if (prof_mark(1, stream) != READ_OK) return READ_EHIP;
    hipLaunchKernelGGL(pass_a, dim3(grid), dim3(256), 0, stream, cc, cam, W, H, ws.keys, ws.zimg[fp],
                       (const unsigned short *)ws.hiz, ws.nbx, ws.hdr, next_pos, fp, si, items, stats, ks, bi);
    READ_CHECK_LAUNCH();
    if (prof_mark(2, stream) != READ_OK) return READ_EHIP;
    if (bins)
        hipLaunchKernelGGL(cells_merge_hiz_kernel, dim3((unsigned)ws.bin_tiles), dim3(256), 0, stream, ws.keys, ws.zimg[fp], W, H,
                           ws.nbx, ws.hiz, bi);
    else
        hipLaunchKernelGGL(cells_hiz_kernel, dim3(ceil_div(ws.nbx * ws.nby, 256)), dim3(256), 0, stream,
                           (const unsigned long long *)ws.keys, ws.zimg[fp], W, H, ws.nbx, ws.nby, ws.hiz, ks);
    READ_CHECK_LAUNCH();
    bi.recs = nullptr;
    if (prof_mark(3, stream) != READ_OK) return READ_EHIP;
    hipLaunchKernelGGL(pass_b, dim3((unsigned)(device_cus() * (g_splat_wgs_b > 0 ? g_splat_wgs_b : g_splat_wgs))), dim3(256), 0, stream, cc, cam, W, H, ws.keys, ws.zimg[fp],
                       (const unsigned short *)ws.hiz, ws.nbx, ws.hdr, next_pos, fp, si, items, stats, ks, bi);
    READ_CHECK_LAUNCH();
    // ---- resolve; with an announced next camera the same launch prepares the next frame's set
    const bool ahead = h.hinted && g_splat_ahead;
    h.hinted = false;
    h.frame += 1;
    if (prof_mark(4, stream) != READ_OK) return READ_EHIP;
    if (!ahead) {
        const int rc = resolve_launch(ws.keys, 1, b0, W, H, levels, idx_levels, depth_levels, 0, ws, 2, stream, ks, fp);
        if (rc == READ_OK && prof_mark(5, stream) == READ_OK) g_prof_valid = g_splat_prof != 0;
        return rc;
    }
    NextFrame nx;
    memcpy(nx.cam.m, h.hint, sizeof(nx.cam.m));
    nx.zimg = ws.zimg[fp ^ 1];
    nx.pos_img = next_pos;
    nx.cset = fp ^ 1;
    nx.sub = sub_next;
    nx.use_seeds = g_splat_seeds;
    nx.class_blocks = class_blocks;
    nx.near_count = (float)g_splat_near;
    const ResolveOut out = resolve_out(b0, W, H, levels, idx_levels, depth_levels, 0);
    const int tiles_x = ceil_div(W, 32), res_blocks = tiles_x * ceil_div(H, 32);
    hipLaunchKernelGGL(cells_resolve_next_kernel, dim3((unsigned)(res_blocks + class_blocks + (g_splat_seeds ? seed_blocks : 0))), dim3(256),
                       0, stream, ws.keys, W, H, levels, out, tiles_x, res_blocks, ws.hdr, ws.zimg[fp], fp, ks, cc, nx, si);
    READ_CHECK_LAUNCH();
    h.pred = true;
    memcpy(h.pm, h.hint, sizeof(h.pm));
    h.pW = W;
    h.pH = H;
    h.p_sub = sub_next;
    h.p_near = g_splat_near;
    h.p_ns = si.ns;
    h.p_seeds = g_splat_seeds;
    h.p_counters = (char *)ws.hdr + HEADER_STRIPS_OFFSET + (size_t)(fp ^ 1) * COUNTER_SET_BYTES;
    h.p_zimg = ws.zimg[fp ^ 1];
    h.p_zimg_bytes = (size_t)W * H * sizeof(unsigned);
    h.p_cells = (const void *)cc.pts;
    h.p_cells_frame = blob_frames;
    h.p_n = cc.hdr_n;
    if (prof_mark(5, stream) == READ_OK) g_prof_valid = g_splat_prof != 0;
    return READ_OK;
}

}  // namespace

extern "C" size_t read_splat_workspace_bytes(int B, int W, int H)
{
    if (B < 1 || W < 1 || H < 1) return 0;
    return ws_layout(nullptr, B, W, H).total;
}

namespace readhip {
void splat_set_subset(int v) { g_splat_subset = v < 0 ? 0 : v; }
void splat_set_stats(int v) { g_splat_stats = v; }
int splat_set_mode(int m)
{
    if (m != MODE_AGENT && m != MODE_HIZ) return READ_EINVAL;
    g_splat_mode = m;
    return READ_OK;
}
void splat_set_near(int v) { g_splat_near = v < 1 ? 1 : v; }
void splat_set_cells(int v) { g_splat_cells = v; }
void splat_set_seeds(int v) { g_splat_seeds = v; }
void splat_set_cells_sub(int v) { g_splat_cells_sub = v < 0 ? 0 : v; }
void splat_set_items(int v) { g_splat_items = v >= 4 ? 4 : (v >= 2 ? 2 : 1); }
void splat_set_zl2(int v) { g_splat_zl2 = v != 0; }
void splat_set_lds(int v) { g_splat_lds = v != 0; }
void splat_set_bins(int v) { g_splat_bins = v != 0; }
void splat_set_ahead(int v) { g_splat_ahead = v != 0; }
void splat_set_mark(int v) { g_splat_mark = v != 0; }
void splat_set_cells_batch(int v) { g_splat_cells_batch = v != 0; }
void splat_set_compact(int v) { g_splat_compact = v != 0; }
void splat_set_wgs_b(int v) { g_splat_wgs_b = v < 0 ? 0 : (v > 16 ? 16 : v); }
void splat_set_sticky(int v) { g_splat_sticky = v < 0 ? 0 : (v > 200 ? 200 : v); }
void splat_set_prof(int v) { g_splat_prof = v != 0; }
void splat_set_kslot(int v) { g_splat_kslot = v < 0 ? 0 : (v > 2 ? 2 : v); }
void splat_set_wgs(int v) { g_splat_wgs = v < 1 ? 1 : (v > 16 ? 16 : v); }
void splat_set_strips(int v) { g_splat_strips = v >= 8 ? 8 : (v >= 4 ? 4 : (v >= 2 ? 2 : 1)); }
int splat_get(const char *key, int *value)
{
    if (!strcmp(key, "splat_mode")) *value = g_splat_mode;
    else if (!strcmp(key, "splat_subset")) *value = g_splat_subset;
    else if (!strcmp(key, "splat_stats")) *value = g_splat_stats;
    else if (!strcmp(key, "splat_near")) *value = g_splat_near;
    else if (!strcmp(key, "splat_cells")) *value = g_splat_cells;
    else if (!strcmp(key, "splat_seeds")) *value = g_splat_seeds;
    else if (!strcmp(key, "splat_cells_sub")) *value = g_splat_cells_sub;
    else if (!strcmp(key, "splat_items")) *value = g_splat_items;
    else if (!strcmp(key, "splat_strips")) *value = g_splat_strips;
    else if (!strcmp(key, "splat_wgs")) *value = g_splat_wgs;
    else if (!strcmp(key, "splat_zl2")) *value = g_splat_zl2;
    else if (!strcmp(key, "splat_lds")) *value = g_splat_lds;
    else if (!strcmp(key, "splat_bins")) *value = g_splat_bins;
    else if (!strcmp(key, "splat_ahead")) *value = g_splat_ahead;
    else if (!strcmp(key, "splat_mark")) *value = g_splat_mark;
    else if (!strcmp(key, "splat_cells_batch")) *value = g_splat_cells_batch;
    else if (!strcmp(key, "splat_compact")) *value = g_splat_compact;
    else if (!strcmp(key, "splat_wgs_b")) *value = g_splat_wgs_b;
    else if (!strcmp(key, "splat_sticky")) *value = g_splat_sticky;
    else if (!strcmp(key, "splat_prof")) *value = g_splat_prof;
    else if (!strcmp(key, "splat_kslot")) *value = g_splat_kslot;
    else return 0;
    return 1;
}
}

__global__ __launch_bounds__(256) void splat_header_clear_kernel(int *hdr)
{
    for (int i = threadIdx.x; i < (int)(HEADER_BYTES / 4); i += blockDim.x) hdr[i] = 0;    // "no previous frame"
}

extern "C" int read_splat_workspace_init(void *ws, size_t ws_bytes, void *stream)
{
    READ_CHECK_ARG(ws && ws_bytes % 8 == 0, "read_splat_workspace_init: workspace null or not a multiple of 8 bytes");
    // everything becomes EMPTY (all ones); the header is then zeroed: "no previous frame"
    READ_CHECK_ARG((uintptr_t)ws % 256 == 0, "read_splat_workspace_init: workspace must be 256-byte aligned");
    const long long count = (long long)(ws_bytes / 8);
    if (count == 0) return READ_OK;
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        g_ws_host.erase(ws);                                       // frame counter, hint and prediction of whatever lived here before
    }
    hipLaunchKernelGGL(fill_keys_kernel, dim3((unsigned)ceil_div64(count, 256)), dim3(256), 0, as_stream(stream),
                       (unsigned long long *)ws, count);
    READ_CHECK_LAUNCH();
    if (ws_bytes >= HEADER_BYTES) {
        hipLaunchKernelGGL(splat_header_clear_kernel, dim3(1), dim3(256), 0, as_stream(stream), (int *)ws);
        READ_CHECK_LAUNCH();
    }
    return READ_OK;
}

extern "C" int read_splat_forward(const float *xyz, int64_t n, const float *M_host, int B, int W, int H,
                                  int levels, int32_t *const *idx_levels, float *const *depth_levels,
                                  void *ws, size_t ws_bytes, void *stream)
{
    READ_CHECK_ARG(n >= 0 && (n == 0 || xyz), "read_splat_forward: xyz is null");
    READ_CHECK_ARG(n <= 0xFFFFFFFEll, "read_splat_forward: point ids must fit 32 bits");
    READ_CHECK_ARG(M_host, "read_splat_forward: M_host is null");
    READ_CHECK_ARG(B >= 1 && W >= 1 && H >= 1, "read_splat_forward: bad B/W/H (%d,%d,%d)", B, W, H);
    READ_CHECK_ARG((long long)W * H < (1ll << 31), "read_splat_forward: image too large");
    READ_CHECK_ARG(levels >= 1 && levels <= READ_MAX_LEVELS, "read_splat_forward: levels must be 1..%d",
                   READ_MAX_LEVELS);
    READ_CHECK_ARG(idx_levels || depth_levels, "read_splat_forward: no outputs requested");
    READ_CHECK_ARG(level_dim(W, levels - 1) >= 1 && level_dim(H, levels - 1) >= 1,
                   "read_splat_forward: coarsest level is empty");
    READ_CHECK_ARG(ws && (uintptr_t)ws % 256 == 0, "read_splat_forward: workspace null or not 256-byte aligned");
    if (ws_bytes < read_splat_workspace_bytes(B, W, H)) {
        set_error("read_splat_forward: workspace %zu < %zu bytes", ws_bytes, read_splat_workspace_bytes(B, W, H));
        return READ_ENOMEM;
    }
    const WsLayout L = ws_layout(ws, B, W, H);
    hipStream_t s = as_stream(stream);
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);             // a cell-path prediction pending on this workspace is not for this call
        auto it = g_ws_host.find(ws);
        if (it != g_ws_host.end()) {
            it->second.hinted = false;
            const int rc = ws_drop_prediction(it->second, s);
            if (rc != READ_OK) return rc;
        }
    }
    const int mask = (1 << (levels - 1)) - 1;
    if (((W | H) & mask) == 0) {
        // pyramid identity holds (App. A.4): one pass over the points feeds every level
        return project_and_resolve(xyz, n, M_host, B, W, H, levels, idx_levels, depth_levels, 0, L, true, s);
    }
    // generic sizes: rasterise each level directly, as the reference does (no warm start: the key image
    // of every level lives at the front of the same region)
    for (int l = 0; l < levels; ++l) {
        WsLayout Ll = L;
        Ll.nbx = ceil_div(level_dim(W, l), 4);
        Ll.nby = ceil_div(level_dim(H, l), 4);
        const int rc = project_and_resolve(xyz, n, M_host, B, level_dim(W, l), level_dim(H, l), 1, idx_levels,
                                           depth_levels, l, Ll, false, s);
        if (rc != READ_OK) return rc;
    }
    return READ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// cell-ordered cloud: host build
// ---------------------------------------------------------------------------------------------------------------
namespace {

size_t cells_chunks(int64_t n) { return (size_t)((n + CELL_CHUNK - 1) / CELL_CHUNK); }

struct CellOffsets {
    size_t pts, aabb, list_a, list_b, sticky, total;
};
CellOffsets cell_offsets(int64_t n)
{
    const size_t nc = cells_chunks(n);
    CellOffsets o;
    o.pts = CELL_HEADER_BYTES;
    o.aabb = o.pts + nc * CELL_CHUNK * sizeof(float4);
    o.list_a = o.aabb + nc * 8 * sizeof(float);                      // per-frame scratch (written by the passes)
    o.list_b = o.list_a + ((MAX_STRIPS * A_BANDS * nc * sizeof(int) + 255) & ~(size_t)255);
    o.sticky = o.list_b + MAX_STRIPS * nc * sizeof(CellEntryB);
    o.total = o.sticky + ((nc + 255) & ~(size_t)255);
    return o;
}

inline uint32_t spread10(uint32_t v)       // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x30000ffu;
    v = (v | (v << 8)) & 0x300f00fu;
    v = (v | (v << 4)) & 0x30c30c3u;
    v = (v | (v << 2)) & 0x9249249u;
    return v;
}

// run fn(t, begin, end) over [0, n) split into T contiguous ranges on T host threads
template <class F>
void parallel_ranges(size_t n, int T, F fn)
{
    if (T <= 1 || n < (size_t)T * 4096) {
        fn(0, (size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" size_t read_splat_cells_bytes(int64_t n)
{
    if (n < 1 || n > 0xFFFFFFFEll) return 0;
    return cell_offsets(n).total;
}

// Host-side build (once per cloud, multi-threaded): Morton order over a 1024^3 grid of the bounding box (stable LSD
// radix sort, so equal codes keep ascending ids), chunks of 1024 consecutive points with their exact bounding boxes.
extern "C" int read_splat_cells_build_host(const float *xyz, int64_t n, void *blob, size_t blob_bytes)
{
    READ_CHECK_ARG(xyz && blob, "read_splat_cells_build_host: null pointer");
    READ_CHECK_ARG(n >= 1 && n <= 0xFFFFFFFEll, "read_splat_cells_build_host: n out of range");
    const CellOffsets o = cell_offsets(n);
    READ_CHECK_ARG(blob_bytes >= o.total, "read_splat_cells_build_host: buffer %zu < %zu bytes", blob_bytes, o.total);
    const size_t N = (size_t)n;
    int T = (int)std::thread::hardware_concurrency();
    T = T < 1 ? 1 : (T > 32 ? 32 : T);
    if (N < (size_t)T * 4096) T = 1;

    // 1. bounding box + finiteness
    std::vector<float> tlo((size_t)T * 3, 3.0e38f), thi((size_t)T * 3, -3.0e38f);
    std::vector<long long> tbad((size_t)T, -1);
    parallel_ranges(N, T, [&](int t, size_t b, size_t e) {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        long long bad = -1;
        for (size_t i = b; i < e; ++i)
            for (int k = 0; k < 3; ++k) {
                const float v = xyz[3 * i + k];
                if (!(v == v && v - v == 0.0f) && bad < 0) bad = (long long)i;
                lo[k] = v < lo[k] ? v : lo[k];
                hi[k] = v > hi[k] ? v : hi[k];
            }
        for (int k = 0; k < 3; ++k) {
            tlo[(size_t)t * 3 + k] = lo[k];
            thi[(size_t)t * 3 + k] = hi[k];
        }
        tbad[(size_t)t] = bad;
    });
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int t = 0; t < T; ++t) {
        READ_CHECK_ARG(tbad[(size_t)t] < 0, "read_splat_cells_build_host: point %lld is not finite", tbad[(size_t)t]);
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], tlo[(size_t)t * 3 + k]);
            hi[k] = std::max(hi[k], thi[(size_t)t * 3 + k]);
        }
    }
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) ext = (hi[k] - lo[k]) > ext ? (hi[k] - lo[k]) : ext;
    const float scale = ext > 0.0f ? 1023.999f / ext : 0.0f;

    // 2. Morton codes (30 bits) with the id in the low word
    std::vector<uint64_t> a(N), b(N);
    parallel_ranges(N, T, [&](int, size_t bgn, size_t end) {
        for (size_t i = bgn; i < end; ++i) {
            uint32_t q[3];
            for (int k = 0; k < 3; ++k) {
                const float t = (xyz[3 * i + k] - lo[k]) * scale;
                q[k] = t <= 0.0f ? 0u : (t >= 1023.0f ? 1023u : (uint32_t)t);
            }
            const uint64_t code = spread10(q[0]) | ((uint64_t)spread10(q[1]) << 1) | ((uint64_t)spread10(q[2]) << 2);
            a[i] = (code << 32) | (uint64_t)(uint32_t)i;
        }
    });

    // 3. stable LSD radix sort, 10 bits per pass: per-thread histograms over contiguous ranges keep the order stable
    std::vector<size_t> hist((size_t)T * 1024);
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = 32 + 10 * pass;
        std::fill(hist.begin(), hist.end(), (size_t)0);
        parallel_ranges(N, T, [&](int t, size_t bgn, size_t end) {
            size_t *h = hist.data() + (size_t)t * 1024;
            for (size_t i = bgn; i < end; ++i) ++h[(a[i] >> shift) & 1023u];
        });
        size_t run = 0;
        for (int d = 0; d < 1024; ++d)
            for (int t = 0; t < T; ++t) {
                const size_t c = hist[(size_t)t * 1024 + d];
                hist[(size_t)t * 1024 + d] = run;
                run += c;
            }
        parallel_ranges(N, T, [&](int t, size_t bgn, size_t end) {
            size_t *h = hist.data() + (size_t)t * 1024;
            for (size_t i = bgn; i < end; ++i) b[h[(a[i] >> shift) & 1023u]++] = a[i];
        });
        a.swap(b);
    }

    // 4. records + boxes
    char *base = (char *)blob;
    memset(base, 0, CELL_HEADER_BYTES);
    memset(base + o.sticky, 0, o.total - o.sticky);
    float *rec = (float *)(base + o.pts);
    float *bb = (float *)(base + o.aabb);
    const size_t nc = cells_chunks(n);
    parallel_ranges(nc, T, [&](int, size_t cb, size_t ce) {
        for (size_t c = cb; c < ce; ++c) {
            float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            for (size_t i = c * CELL_CHUNK; i < (c + 1) * CELL_CHUNK; ++i) {
                const uint32_t id = (uint32_t)(a[i < N ? i : N - 1] & 0xffffffffu);   // tail: copies of the last point
                float *r = rec + 4 * i;
                for (int k = 0; k < 3; ++k) {
                    const float v = xyz[3 * (size_t)id + k];
                    r[k] = v;
                    mn[k] = v < mn[k] ? v : mn[k];
                    mx[k] = v > mx[k] ? v : mx[k];
                }
                memcpy(r + 3, &id, sizeof(id));
            }
            float *r = bb + 8 * c;
            r[0] = mn[0]; r[1] = mn[1]; r[2] = mn[2]; r[3] = mx[0]; r[4] = mx[1]; r[5] = mx[2]; r[6] = r[7] = 0.0f;
        }
    });
    CellHeader *h = (CellHeader *)base;
    h->n = n;
    h->nchunks = (int)nc;
    h->version = CELL_VERSION;
    double vol = 1.0;
    for (int k = 0; k < 3; ++k) {
        h->bbox[k] = lo[k];
        h->bbox[3 + k] = hi[k];
        const double e = (double)hi[k] - lo[k];
        vol *= e > 1e-6 * ext ? e : (ext > 0 ? 1e-6 * ext : 1.0);   // a flat cloud still gets a finite density
    }
    h->density = (float)((double)n / (vol > 0 ? vol : 1.0));
    return READ_OK;
}

extern "C" int read_splat_profile_last(float *ms5)
{
    READ_CHECK_ARG(ms5, "read_splat_profile_last: null pointer");
    READ_CHECK_ARG(g_prof_valid, "read_splat_profile_last: no profiled cell-path frame (read_tuning_set(\"splat_prof\", 1), then a frame)");
    READ_CHECK_HIP(hipEventSynchronize(g_prof_ev[5]));
    for (int i = 0; i < 5; ++i) {
        ms5[i] = 0.0f;
        if (i == 0 && !g_prof_slot0) continue;
        READ_CHECK_HIP(hipEventElapsedTime(&ms5[i], g_prof_ev[i], g_prof_ev[i + 1]));
    }
    return READ_OK;
}

extern "C" int read_splat_hint_next_camera(void *ws, const float *M_next_host)
{
    READ_CHECK_ARG(ws, "read_splat_hint_next_camera: null workspace");
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    WsHost &h = g_ws_host[ws];
    h.hinted = M_next_host != nullptr;
    if (M_next_host) memcpy(h.hint, M_next_host, sizeof(h.hint));
    return READ_OK;
}

extern "C" int read_splat_cells_invalidate(const void *cells, int64_t n)
{
    READ_CHECK_ARG(cells && n >= 1, "read_splat_cells_invalidate: null blob or n < 1");
    const CellOffsets o = cell_offsets(n);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    g_cells_frames.erase((const char *)cells + o.pts);           // the next frame over this address starts a new epoch: no prediction matches it
    return READ_OK;
}

extern "C" int read_splat_forward_cells(const float *xyz, void *cells, int64_t n, const float *M_host, int B,
                                        int W, int H, int levels, int32_t *const *idx_levels,
                                        float *const *depth_levels, void *ws, size_t ws_bytes, void *stream)
{
    const int mask = levels >= 1 && levels <= READ_MAX_LEVELS ? (1 << (levels - 1)) - 1 : 0;
    // the striped passes serve the single-camera, pyramid-identity case with 128-byte-aligned key rows (the per-frame
    // render path); everything else goes through the plain pass
    if (!cells || !g_splat_cells || g_splat_mode != MODE_HIZ || B < 1 || (B > 1 && !g_splat_cells_batch) || n < (1 << 20) || ((W | H) & mask) != 0 ||
        (W & 15) != 0 || !xyz || !M_host || !ws || W < 1 || H < 1)
        return read_splat_forward(xyz, n, M_host, B, W, H, levels, idx_levels, depth_levels, ws, ws_bytes, stream);
    READ_CHECK_ARG(n <= 0xFFFFFFFEll, "read_splat_forward_cells: point ids must fit 32 bits");
    READ_CHECK_ARG(levels >= 1 && levels <= READ_MAX_LEVELS, "read_splat_forward_cells: levels must be 1..%d",
                   READ_MAX_LEVELS);
    READ_CHECK_ARG(idx_levels || depth_levels, "read_splat_forward_cells: no outputs requested");
    READ_CHECK_ARG((long long)W * H < (1ll << 31), "read_splat_forward_cells: image too large");
    READ_CHECK_ARG((uintptr_t)ws % 256 == 0 && (uintptr_t)cells % 256 == 0,
                   "read_splat_forward_cells: workspace and cells must be 256-byte aligned");
    if (ws_bytes < read_splat_workspace_bytes(B, W, H)) {
        set_error("read_splat_forward_cells: workspace %zu < %zu bytes", ws_bytes, read_splat_workspace_bytes(B, W, H));
        return READ_ENOMEM;
    }
    const CellOffsets o = cell_offsets(n);
    CellCloud cc;
    cc.hdr = (const CellHeader *)cells;
    cc.pts = (const float4 *)((const char *)cells + o.pts);
    cc.aabb = (const float *)((const char *)cells + o.aabb);
    cc.list_a = (int *)((char *)cells + o.list_a);
    cc.list_b = (CellEntryB *)((char *)cells + o.list_b);
    cc.sticky = (unsigned char *)cells + o.sticky;
    cc.nchunks = (int)cells_chunks(n);
    cc.hdr_n = n;
    cc.sticky_frames = g_splat_sticky;
    cc.mark_candidates = g_splat_mark && g_splat_sticky > 0;
    const WsLayout L = ws_layout(ws, B, W, H);
    // A batch of cameras (the training step's 8 crops, MyRender) = B cell-path frames, one after the other through the same workspace
    // state: each reads only what its chunk lists keep (the plain pass read the whole cloud once per 8 cameras AND projected every
    // point 8 times: 777 MB and 729 us for 8 crops of a 10 M-point cloud, profiles/r4_hbm_traffic_per_kernel.md).  The seeds a
    // camera inherits from its predecessor are bounds from real points under ITS matrix — valid, if rarely useful across crops.
    for (int b = 0; b < B; ++b) {
        const int rc = cells_frame(cc, M_host + 16 * b, W, H, levels, idx_levels, depth_levels, L, as_stream(stream), b);
        if (rc != READ_OK) return rc;
    }
    return READ_OK;
}

extern "C" int read_splat_forward_gl(const float *xyz, int64_t n, const float *M_host, int W, int H,
                                     const read_splat_gl_opts *opts, int32_t *idx, float *depth, void *ws,
                                     size_t ws_bytes, void *stream)
{
    READ_CHECK_ARG(n >= 0 && (n == 0 || xyz), "read_splat_forward_gl: xyz is null");
    READ_CHECK_ARG(n <= 0xFFFFFFFEll, "read_splat_forward_gl: point ids must fit 32 bits");
    READ_CHECK_ARG(M_host && opts, "read_splat_forward_gl: M_host / opts is null");
    READ_CHECK_ARG(W >= 1 && H >= 1 && (long long)W * H < (1ll << 31), "read_splat_forward_gl: bad size (%d,%d)", W, H);
    READ_CHECK_ARG(idx || depth, "read_splat_forward_gl: no outputs requested");
    READ_CHECK_ARG((opts->point_size >= 1.0f || (opts->point_size == 0.0f && opts->point_sizes)) && opts->min_point_size >= 0.0f,
                   "read_splat_forward_gl: point_size must be >= 1, or 0 with a per-point size array");
    READ_CHECK_ARG(ws && (uintptr_t)ws % 256 == 0, "read_splat_forward_gl: workspace null or not 256-byte aligned");
    if (ws_bytes < read_splat_workspace_bytes(1, W, H)) {
        set_error("read_splat_forward_gl: workspace %zu < %zu bytes", ws_bytes, read_splat_workspace_bytes(1, W, H));
        return READ_ENOMEM;
    }
    const WsLayout L = ws_layout(ws, 1, W, H);
    hipStream_t s = as_stream(stream);
    if (n > 0) {
        Cam1 cam;
        memcpy(cam.m, M_host, sizeof(cam.m));
        GlOpts o;
        o.point_size = opts->point_size;
        o.relative = opts->relative;
        o.min_point_size = opts->min_point_size;
        o.discard = opts->discard;
        o.drop_threshold = opts->drop_threshold;
        o.drop_seed = opts->drop_seed;
        o.perturb = opts->perturb;
        o.perturb_amp = opts->perturb_amp;
        o.perturb_seed = opts->perturb_seed;
        o.point_sizes = opts->point_sizes;
        int64_t blocks = ceil_div64(n, 256);
        if (blocks > (int64_t)device_cus() * 8) blocks = (int64_t)device_cus() * 8;
        hipLaunchKernelGGL(splat_project_gl_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xyz, (long long)n, cam, W, H,
                           L.keys, o);
        READ_CHECK_LAUNCH();
    }
    int32_t *idx_l[1] = {idx};
    float *dep_l[1] = {depth};
    return resolve_launch(L.keys, 1, 0, W, H, 1, idx ? idx_l : nullptr, depth ? dep_l : nullptr, 0, L, 0, s);
}

namespace {
// One thread per point: the pixel it projects to (or -1) and its depth — project_one as every pass uses it, without the z-test.
__global__ __launch_bounds__(256) void project_points_kernel(const float *__restrict__ xyz, long long n, Cam1 cam, int W, int H,
                                                             int32_t *__restrict__ pixel, float *__restrict__ depth)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float d;
        int xx, yy;
        const int pix = project_one(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cam.m, W, H, d, xx, yy);
        pixel[i] = pix;
        if (depth) depth[i] = d;
    }
}
}  // namespace

extern "C" int read_splat_project_points(const float *xyz, int64_t n, const float *M_host, int W, int H, int32_t *pixel,
                                         float *depth, void *stream)
{
    READ_CHECK_ARG(n >= 0 && (n == 0 || (xyz && pixel)), "read_splat_project_points: null pointer");
    READ_CHECK_ARG(M_host, "read_splat_project_points: M_host is null");
    READ_CHECK_ARG(W >= 1 && H >= 1 && (long long)W * H < (1ll << 31), "read_splat_project_points: bad W/H (%d,%d)", W, H);
    if (n == 0) return READ_OK;
    Cam1 cam;
    for (int i = 0; i < 16; ++i) cam.m[i] = M_host[i];
    long long blocks = ceil_div64(n, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(project_points_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), xyz, (long long)n, cam, W,
                       H, pixel, depth);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_index_to_float(const int32_t *idx, int64_t count, float *out, void *stream)
{
    READ_CHECK_ARG(count >= 0 && (count == 0 || (idx && out)), "read_index_to_float: null pointer");
    if (count == 0) return READ_OK;
    hipLaunchKernelGGL(index_to_float_kernel, dim3((unsigned)ceil_div64(count, 256)), dim3(256), 0,
                       as_stream(stream), idx, (long long)count, out);
    READ_CHECK_LAUNCH();
    return READ_OK;
}
