// Z-buffered point splat for gfx950.
//
// Replaces the reference's per-scale, per-camera DepthProject launches
// (MyRender/CloudProjection/point_render.cu:125-200; GL twin READ/gl/render.py:52-85) with
//   1. splat_project : ONE pass over the cloud for all B cameras.  Each accepted point becomes a
//      packed 64-bit key (fp32 depth bits << 32 | point id) and is folded into a level-0 key
//      image with an unsigned 64-bit atomic min — min depth, ties -> min id, order independent
//      (SURVEY.md App. A.3).  A relaxed L1-bypassing read of the current key filters out the
//      points that cannot win before they cost an atomic (keys only ever decrease, so a stale
//      read is merely conservative).  Each XCD folds into its own image with L2-local atomics.
//   2. splat_resolve : one small pass over the key images that takes the min over the XCDs, derives
//      levels 1..4 by 2x2 key-min (exactly the reference's five rasterisations, App. A.4), unpacks
//      (id, depth) for every level and resets the key images to EMPTY for the next frame.
//
// HBM-bound: algorithmic bytes per frame = 12*N (xyz read once) + 8*sum_l(h_l*w_l) (id + depth).
// The arithmetic that decides which PIXEL a point lands in is bit-exact fp32: no FMA
// contraction, IEEE division, left-to-right dot products (helper_math.h:1252-1255).
#include "common.h"

#pragma clang fp contract(off)

using namespace readhip;

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int MAX_CAMS = 8;          // cameras folded into one pass over the points
constexpr int PTS_PER_THREAD = 4;    // 3 x float4 = 4 points

struct CamSet {
    float m[MAX_CAMS][16];
};

typedef unsigned long long __attribute__((address_space(1))) *gkey_ptr;

// point_render.cu:135-147 for one point and one camera; returns the pixel or -1.
__device__ __forceinline__ int project_one(float x, float y, float z, const float *M, int W, int H,
                                           float &depth, int &xx_out, int &yy_out)
{
    const float c0 = M[0] * x + M[1] * y + M[2] * z + M[3] * 1.0f;
    const float c1 = M[4] * x + M[5] * y + M[6] * z + M[7] * 1.0f;
    const float c2 = M[8] * x + M[9] * y + M[10] * z + M[11] * 1.0f;
    const float c3 = M[12] * x + M[13] * y + M[14] * z + M[15] * 1.0f;
    const float nx = c0 / c3, ny = c1 / c3, nz = c2 / c3;
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

// Key-image update policies (read_tuning_set("splat_mode", m)):
//   MODE_XCD   one private key image per XCD.  Workgroups read HW_REG_XCC_ID and fold into
//              "their" image with WORKGROUP-scope atomics, which execute in that XCD's L2 and never
//              cross the fabric; the 3.4 MB image stays L2 resident, and the early-z read (sc1: L2,
//              not the CU's L1) sees every earlier fold of the same XCD.  The resolve pass takes the
//              min over the 8 images.  Correctness needs only that all accesses to image x come
//              from XCD x inside the launch, plus ordinary kernel-boundary visibility.
//   MODE_AGENT (default) one shared image, agent-scope atomics (memory-side), early-z through a possibly
//              stale L2 copy (conservative, but filters less).
//   MODE_NOZ   projection only, no z-buffer traffic: timing floor for tuning, results are invalid.
//   MODE_SYS   one shared image, agent-scope atomics, early-z read at SYSTEM scope (sc0 sc1: bypasses
//              the XCD L2 too, so the filter is exact but every read crosses the fabric).
//   MODE_PEEK / MODE_PEEK_L1 / MODE_ATOM  attribution probes (invalid results): early-z reads only (sc1 /
//              plain L1-cached), and atomics only (no early-z filter).
//   MODE_HIZ   (default) MODE_AGENT plus a temporal warm start and a hierarchical-Z reject that lives in LDS:
//              before the point pass the previous frame's per-pixel winners are re-projected with the new
//              camera and folded in, a conservative far bound per 4x4-pixel block (max of the current depths,
//              +inf if any pixel is empty) is built, and every workgroup copies that bound image (107 KB at
//              1216x352) into LDS.  A point whose depth exceeds its block's bound cannot win (keys only
//              decrease), so it is dropped without touching global memory — ~90 % of the points of a frame.
//              Exact: seeds are real points of this cloud, bounds are upper bounds of the final depths.
//   MODE_AGENT_L1 / MODE_HIZ_L1 (cell-ordered passes, read_tuning_set("splat_l1", 1)): as MODE_AGENT / MODE_HIZ with
//              the early-z read served by the CU's L1.  A stale copy is only ever LARGER than the live key (keys
//              decrease), so the filter stays conservative; the chunks are spatially compact, so their reads share
//              cache lines, and the seeds of the previous launch are visible (L1 is invalidated at kernel start).
enum { MODE_XCD = 0, MODE_AGENT = 1, MODE_NOZ = 2, MODE_SYS = 3, MODE_PEEK = 4, MODE_PEEK_L1 = 5, MODE_ATOM = 6,
       MODE_HIZ = 7, MODE_AGENT_L1 = 8, MODE_HIZ_L1 = 9 };
constexpr int XCD_COPIES = 8;

__device__ __forceinline__ unsigned xcc_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg((3 << 11) | 20) & (XCD_COPIES - 1);
}

template <int MODE>
__device__ __forceinline__ unsigned long long peek_key(const unsigned long long *k)
{
    // relaxed agent-scope load = global_load_dwordx2 sc1: served by L2, never by the CU's stale L1
    if (MODE == MODE_SYS) return __hip_atomic_load(k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == MODE_PEEK_L1 || MODE == MODE_AGENT_L1 || MODE == MODE_HIZ_L1) return *k;
    return __hip_atomic_load(k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__device__ __forceinline__ void fold_key(unsigned long long *k, unsigned long long key)
{
    if (MODE == MODE_XCD)
        __hip_atomic_fetch_min(k, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        __hip_atomic_fetch_min(k, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Far bounds are stored as 16 bits: the upper half (bfloat16, truncated = rounded DOWN) of e = fl(1 - d_max).  Depth
// is d = 1 - O(znear / z), so e keeps 8 mantissa bits of the DISTANCE (0.4 %) where a half-precision d would resolve
// only ~5 m at 30 m; and 16-bit bounds let two 1024-thread workgroups share a CU's LDS (8 waves per SIMD instead of
// 4 for this latency-bound pass).  Reject iff fl(1 - d) < bound: rounding is monotonic, so fl(1 - d) < fl(1 - d_max)
// implies d > d_max strictly (ties pass), and truncation only lowers the bound.  Empty block -> -1 (never rejects).
__device__ __forceinline__ unsigned short hiz_encode(unsigned depth_bits_max)
{
    if (depth_bits_max > 0x7f800000u) return 0xbf80;                  // a pixel of the block is still EMPTY: e = -1
    return (unsigned short)(__float_as_uint(1.0f - __uint_as_float(depth_bits_max)) >> 16);
}
__device__ __forceinline__ bool hiz_reject(unsigned short bound, float d)
{
    return (1.0f - d) < __uint_as_float((unsigned)bound << 16);
}

// NP points of one thread against one camera: all projections first, then all early-z reads in
// flight together, then the (few) atomics — no dependent memory round trip per point.
template <int MODE, int NP>
__device__ __forceinline__ void splat_points_ids(const float (&px)[NP], const float (&py)[NP], const float (&pz)[NP],
                                                 const unsigned (&ids)[NP], int nvalid, const float *M, int W, int H,
                                                 unsigned long long *keys, unsigned &sink,
                                                 const unsigned short *hiz = nullptr, int nbx = 0, unsigned *stat = nullptr)
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
        if (MODE == MODE_HIZ || MODE == MODE_HIZ_L1) {
            // LDS-resident far bound of the point's 4x4 block; strictly greater cannot win (ties must pass)
            if (hiz_reject(hiz[pix[k] >= 0 ? (yy >> 2) * nbx + (xx >> 2) : 0], d)) pix[k] = -1;
        }
        if (stat && pix[k] >= 0) stat[1]++;                 // survived the LDS hi-z (or no hi-z)
        key[k] = ((unsigned long long)__float_as_uint(d) << 32) | ids[k];
    }
    if (MODE == MODE_NOZ) {
#pragma unroll
        for (int k = 0; k < NP; ++k) sink += (unsigned)pix[k];
        return;
    }
    if (MODE == MODE_ATOM) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (pix[k] >= 0) fold_key<MODE>(keys + pix[k], key[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) seen[k] = pix[k] >= 0 ? peek_key<MODE>(keys + pix[k]) : 0ull;
    if (MODE == MODE_PEEK || MODE == MODE_PEEK_L1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) sink += (unsigned)(seen[k] >> 32) + (unsigned)pix[k];
        return;
    }
    // keys only ever decrease, so "not smaller than what I can see" is final
#pragma unroll
    for (int k = 0; k < NP; ++k)
        if (key[k] < seen[k]) {
            fold_key<MODE>(keys + pix[k], key[k]);
            if (stat) stat[2]++;                            // atomics issued
        }
}

template <int MODE, int NP>
__device__ __forceinline__ void splat_points(const float (&px)[NP], const float (&py)[NP], const float (&pz)[NP],
                                             unsigned id0, int nvalid, const float *M, int W, int H,
                                             unsigned long long *keys, unsigned &sink, const unsigned short *hiz = nullptr,
                                             int nbx = 0, unsigned *stat = nullptr)
{
    unsigned ids[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) ids[k] = id0 + k;
    splat_points_ids<MODE, NP>(px, py, pz, ids, nvalid, M, W, H, keys, sink, hiz, nbx, stat);
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

template <int MODE>
__global__ __launch_bounds__(256) void splat_project_kernel(const float *__restrict__ xyz, long long n,
                                                            CamSet cams, int B, int W, int H,
                                                            unsigned long long *__restrict__ keys,
                                                            int vec_ok, unsigned *sink_out, int sub_mod,
                                                            unsigned long long *stats)
{
    unsigned st_local[3] = {0, 0, 0};
    unsigned *stp = stats ? st_local : nullptr;
    // sub_mod > 0: bootstrap pass of MODE_HIZ — only every sub_mod-th chunk of 256 point groups (1024 points)
    const long long npx = (long long)W * H;
    const long long groups = n / PTS_PER_THREAD;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    // image layout: [camera][copy][pixel]; a workgroup only ever touches its own XCD's copy
    const int copies = MODE == MODE_XCD ? XCD_COPIES : 1;
    unsigned long long *kbase = keys + (MODE == MODE_XCD ? (long long)xcc_id() * npx : 0);
    unsigned sink = 0;

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
                splat_points<MODE, 4>(px, py, pz, (unsigned)(g * PTS_PER_THREAD), 4, cams.m[cam], W, H,
                                      kbase + (long long)cam * copies * npx, sink, nullptr, 0, stp);
        }
    }
    // tail (n % 4 points), or everything when the pointer is not 16-byte aligned
    const long long first = vec_ok ? groups * PTS_PER_THREAD : 0;
    for (long long i = first + tid0; i < n && sub_mod <= 0; i += nthreads) {     // (the bootstrap pass leaves the tail to the main pass)
        const float px[1] = {xyz[3 * i + 0]}, py[1] = {xyz[3 * i + 1]}, pz[1] = {xyz[3 * i + 2]};
        for (int cam = 0; cam < B; ++cam)
            splat_points<MODE, 1>(px, py, pz, (unsigned)i, 1, cams.m[cam], W, H,
                                  kbase + (long long)cam * copies * npx, sink);
    }
    if ((MODE == MODE_NOZ || MODE == MODE_PEEK || MODE == MODE_PEEK_L1) && sink == 0x7fffffffu && sink_out)
        *sink_out = sink;                                                       // keeps the probes live
    if (stats)
        for (int i = 0; i < 3; ++i) atomicAdd(stats + i, (unsigned long long)st_local[i]);
}


// ---- MODE_HIZ pieces ------------------------------------------------------------------------------
struct SplatHeader {          // first 256 bytes of the workspace
    int valid, W, H, pad;
};
constexpr size_t HEADER_BYTES = 256;

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

// The point pass of MODE_HIZ: one 1024-thread workgroup per CU (persistent, grid-stride) with the whole
// bound image in LDS.
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
    unsigned sink = 0;
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
            splat_points<MODE_HIZ, 4>(px, py, pz, (unsigned)(g * PTS_PER_THREAD), 4, cams.m[0], W, H, keys, sink, hiz, nbx, stp);
        }
    }
    const long long first = vec_ok ? groups * PTS_PER_THREAD : 0;
    for (long long i = first + tid0; i < n; i += nthreads) {
        const float px[1] = {xyz[3 * i + 0]}, py[1] = {xyz[3 * i + 1]}, pz[1] = {xyz[3 * i + 2]};
        splat_points<MODE_HIZ, 1>(px, py, pz, (unsigned)i, 1, cams.m[0], W, H, keys, sink, hiz, nbx, stp);
    }
    if (stats)
        for (int i = 0; i < 3; ++i) atomicAdd(stats + 4 + i, (unsigned long long)st_local[i]);
}


// ---- cell-ordered cloud: chunk-level frustum / occlusion culling ---------------------------------------------
// read_splat_cells_build_host() sorts the cloud once along a Morton curve and cuts it into chunks of 1024 points with
// their bounding boxes; the original point ids travel with the points (keys carry the ORIGINAL id, so the result is
// bit-identical to the unsorted pass — atomic min does not care about order).  Per frame:
//   splat_seed_kernel      (extra blocks) classifies every chunk from the 8 projected corners of its box: outside the
//                          frustum -> dropped (no point of it is read); nearest corner closer than w_split, or every
//                          sub-th chunk -> list A; the rest -> list B with its screen rectangle and depth threshold.
//                          Lists are compacted with one atomic per wave.
//   splat_cells_kernel<A>  one wave per list-A chunk: plain early-z + atomic min (these chunks set the first bounds).
//   splat_hiz_kernel       far bound per 4x4 block.
//   splat_cells_kernel<B>  one wave per list-B chunk: skipped when its nearest possible depth is behind the bound of
//                          EVERY block its rectangle touches, otherwise the LDS hi-z point pass.
// Conservative arithmetic: the fp32 projection of a point and of the box corners differ by rounding; with
// S_k = sum_j |M_kj| max|box_j| + |M_k3| every computed clip coordinate is within gamma S_k of the exact one, so ndc
// errors are bounded by gamma (S_k + S_3) / w_min + ulp; the rectangle is widened and the depth test tightened by that
// much (gamma = 1e-6, >= 4x the worst case of a 4-term fp32 dot product).  A box with a corner at or behind the camera
// plane is never culled (list A).  A single dynamic chunk counter was tried first: 37 K same-address atomics cost
// 0.65 ms per pass (~17 ns each) — hence classification + static walks over compact lists.
struct CellHeader {            // first 256 bytes of the cell-ordered cloud (device and host)
    long long n;
    int nchunks, version;
    float bbox[6];
    float density;             // points per unit volume of the bounding box
};
constexpr int CELL_CHUNK = 1024;
constexpr size_t CELL_HEADER_BYTES = 256;
constexpr size_t CELL_COUNTER_OFFSET = 192;   // two ints in the WORKSPACE header: lengths of list A / list B

struct CellEntryB {            // 16 bytes per list-B chunk
    int chunk;
    unsigned bx;               // bx0 << 16 | bx1 (4x4-pixel block columns, inclusive)
    unsigned by;
    float e_thr;               // cull iff e_thr < min over the rectangle of (1 - far bound)
};

struct CellCloud {             // device pointers into the blob
    const CellHeader *hdr;
    const float *xyz;          // nchunks * 1024 * 3, Morton order, tail padded with copies of the last point
    const unsigned *ids;       // original point id of every sorted point
    const float *aabb;         // nchunks * 8: min xyz, max xyz, 2 pad
    int *list_a;               // scratch: chunk ids of this frame's list A
    CellEntryB *list_b;        // scratch: list B
    int nchunks;
};

// class of one chunk for camera M: 0 dropped, 1 list A, 2 list B (then e fills in)
__device__ __forceinline__ int classify_chunk(const float *bb, const float *M, int W, int H, float w_split, bool boot,
                                              CellEntryB &e)
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
    float wmin = 3.0e38f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float cx = (i & 1) ? mx[0] : mn[0], cy = (i & 2) ? mx[1] : mn[1], cz = (i & 4) ? mx[2] : mn[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[i][k] = M[4 * k] * cx + M[4 * k + 1] * cy + M[4 * k + 2] * cz + M[4 * k + 3] * 1.0f;
        wmin = fminf(wmin, c[i][3]);
    }
    if (!(wmin > fmaxf(1e-3f, 1e-5f * S[3]))) return 1;           // a corner at / behind the camera plane: never culled
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
    if (wmin < w_split || boot) return 1;
    const float dmin = (fmaxf(lo[2], -1.0f) + 1.0f) * 0.5f;
    e.e_thr = (1.0f - dmin) + 2.0f * (0.5f * ez + 2e-7f);
    const float px0 = (float)W * (lo[0] + 1.0f) * 0.5f - ((float)W * 0.5f * ex + 1.0f);
    const float px1 = (float)W * (hi[0] + 1.0f) * 0.5f + ((float)W * 0.5f * ex + 1.0f);
    const float py0 = (float)H * (1.0f - hi[1]) * 0.5f - ((float)H * 0.5f * ey + 1.0f);
    const float py1 = (float)H * (1.0f - lo[1]) * 0.5f + ((float)H * 0.5f * ey + 1.0f);
    const unsigned bx0 = (unsigned)fminf(fmaxf(px0, 0.0f), (float)(W - 1)) >> 2;
    const unsigned bx1 = (unsigned)fminf(fmaxf(px1, 0.0f), (float)(W - 1)) >> 2;
    const unsigned by0 = (unsigned)fminf(fmaxf(py0, 0.0f), (float)(H - 1)) >> 2;
    const unsigned by1 = (unsigned)fminf(fmaxf(py1, 0.0f), (float)(H - 1)) >> 2;
    e.bx = bx0 << 16 | bx1;
    e.by = by0 << 16 | by1;
    return 2;
}

// The classification blocks of splat_seed_kernel (blockIdx >= pix_blocks): one thread per chunk.
__device__ __forceinline__ void classify_block(const CellCloud &cc, const float *M, int W, int H, int sub, float near_count,
                                               int block, int *counts)
{
    const int chunk = block * 256 + threadIdx.x;
    // list A takes the chunks nearer than the distance within which a pixel expects `near_count` points
    const float focal = sqrtf(M[0] * M[0] + M[1] * M[1] + M[2] * M[2]) * (float)W * 0.5f;
    const float w_split = cbrtf(3.0f * near_count * focal * focal / fmaxf(cc.hdr->density, 1e-20f));
    CellEntryB e;
    e.chunk = chunk;
    e.bx = e.by = 0;
    e.e_thr = 0.0f;
    int cls = 0;
    if (chunk < cc.nchunks)
        cls = classify_chunk(cc.aabb + (size_t)chunk * 8, M, W, H, w_split, sub > 0 && chunk % sub == 0, e);
    // wave-aggregated append: one atomic per wave and list
    const int lane = threadIdx.x & 63;
    const unsigned long long ma = __ballot(cls == 1), mb = __ballot(cls == 2);
    int base_a = 0, base_b = 0;
    if (lane == 0) {
        if (ma) base_a = atomicAdd(counts + 0, __popcll(ma));
        if (mb) base_b = atomicAdd(counts + 1, __popcll(mb));
    }
    base_a = __shfl(base_a, 0);
    base_b = __shfl(base_b, 0);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (cls == 1) cc.list_a[base_a + __popcll(ma & below)] = chunk;
    if (cls == 2) cc.list_b[base_b + __popcll(mb & below)] = e;
}

// Re-project the previous frame's winners (one per level-0 pixel) with the new camera and fold them in.
__global__ __launch_bounds__(256) void splat_seed_kernel(const float *__restrict__ xyz, long long n, CamSet cams,
                                                         int W, int H, unsigned long long *__restrict__ keys,
                                                         SplatHeader *hdr, const int *__restrict__ prev_idx,
                                                         CellCloud cc, int pix_blocks, int sub, float near_count)
{
    if ((int)blockIdx.x >= pix_blocks) {           // the extra blocks classify the chunks of the cell-ordered cloud
        classify_block(cc, cams.m[0], W, H, sub, near_count, (int)blockIdx.x - pix_blocks,
                       (int *)((char *)hdr + CELL_COUNTER_OFFSET));
        return;
    }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H || !prev_idx) return;
    if (!(hdr->valid == 1 && hdr->W == W && hdr->H == H)) return;
    const int id = prev_idx[p];
    if (id < 0 || id >= n) return;
    float d;
    int xx, yy;
    const int pix = project_one(xyz[3ll * id], xyz[3ll * id + 1], xyz[3ll * id + 2], cams.m[0], W, H, d, xx, yy);
    if (pix >= 0) fold_key<MODE_AGENT>(keys + pix, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id);
}

template <bool PASS_B, bool L1>
__global__ __launch_bounds__(PASS_B ? 1024 : 256) void splat_cells_kernel(CellCloud cc, CamSet cams, int W, int H,
                                                                         unsigned long long *__restrict__ keys,
                                                                         const unsigned short *__restrict__ hiz_g, int nbx,
                                                                         int nby, const int *counts,
                                                                         unsigned long long *stats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short hiz[];
    if (PASS_B) {
        for (int i = threadIdx.x; i < nbx * nby; i += blockDim.x) hiz[i] = hiz_g[i];
        __syncthreads();
    }
    const float *M = cams.m[0];
    const int lane = threadIdx.x & 63;
    unsigned st_local[3] = {0, 0, 0};
    unsigned *stp = stats ? st_local : nullptr;
    unsigned n_proc = 0, n_cull = 0;
    const float4 *xyz4 = reinterpret_cast<const float4 *>(cc.xyz);
    const uint4 *ids4 = reinterpret_cast<const uint4 *>(cc.ids);
    const int n_list = counts[PASS_B ? 1 : 0];
    // one wave per chunk, lists walked in (roughly Morton) order by neighbouring waves.  Measured alternatives: a team
    // of four waves per chunk (one quarter each) 76 / 48 us for pass A / B instead of 78 / 38; a transposed walk that
    // keeps concurrent waves far apart 94 / 50 (locality of the key image matters more than contention).
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    unsigned sink = 0;
    for (int i = wave; i < n_list; i += n_waves) {
        int chunk;
        if (PASS_B) {
            const CellEntryB e = cc.list_b[i];
            chunk = e.chunk;
            const int bx0 = (int)(e.bx >> 16), bx1 = (int)(e.bx & 0xffffu), by0 = (int)(e.by >> 16), by1 = (int)(e.by & 0xffffu);
            if ((bx1 - bx0 + 1) * (by1 - by0 + 1) <= 8192) {
                float emin = 3.0e38f;                                  // min over the rectangle of (1 - far bound)
                for (int ry = by0; ry <= by1; ++ry)
                    for (int rx = bx0 + lane; rx <= bx1; rx += 64)
                        emin = fminf(emin, __uint_as_float((unsigned)hiz[ry * nbx + rx] << 16));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) emin = fminf(emin, __shfl_xor(emin, o));
                if (e.e_thr < emin) {                                  // every point of the box is behind every bound
                    ++n_cull;
                    continue;
                }
            }
        } else {
            chunk = cc.list_a[i];
        }
        chunk = __builtin_amdgcn_readfirstlane(chunk);
        ++n_proc;
        // ---- the chunk's 1024 points: 4 rounds of 4 points per lane
#pragma unroll 2
        for (int it = 0; it < 4; ++it) {
            const long long g = (long long)chunk * 256 + it * 64 + lane;        // group of 4 points
            const float4 a = xyz4[3 * g + 0];
            const float4 b = xyz4[3 * g + 1];
            const float4 c = xyz4[3 * g + 2];
            const uint4 id = ids4[g];
            const float px[4] = {a.x, a.w, b.z, c.y};
            const float py[4] = {a.y, b.x, b.w, c.z};
            const float pz[4] = {a.z, b.y, c.x, c.w};
            const unsigned ids[4] = {id.x, id.y, id.z, id.w};
            if (PASS_B)
                splat_points_ids<L1 ? MODE_HIZ_L1 : MODE_HIZ, 4>(px, py, pz, ids, 4, M, W, H, keys, sink, hiz, nbx, stp);
            else
                splat_points_ids<L1 ? MODE_AGENT_L1 : MODE_AGENT, 4>(px, py, pz, ids, 4, M, W, H, keys, sink, nullptr, 0, stp);
        }
    }
    if (stats && lane == 0) {
        for (int i = 0; i < 3; ++i) atomicAdd(stats + (PASS_B ? 4 : 0) + i, (unsigned long long)st_local[i]);
        atomicAdd(stats + (PASS_B ? 11 : 8), (unsigned long long)n_proc);
        if (PASS_B) atomicAdd(stats + 9, (unsigned long long)n_cull);
    }
}


// ---- software-pipelined point pass (MODE_AGENT and MODE_HIZ) ------------------------------------------
// vmcnt retires in order and counts atomics, so in the straightforward loop every wave-iteration that
// issued an atomic waits out its memory-side round trip before it may consume the next group's loads
// (measured: 1.9 M filtered atomics cost as much as 200 us, ~5x their throughput cost).  Here an iteration
//   1. issues the loads of the NEXT point group,
//   2. projects the current group and issues its early-z reads,
//   3. issues the atomics that the PREVIOUS iteration decided on (younger than 1./2., so waiting for the
//      reads does not wait for them; they have a whole iteration to complete),
//   4. consumes the early-z reads and records this iteration's survivors as pending.
// sub_sel: 0 all point chunks, 1 only chunks with (chunk % sub_mod) == 0 (bootstrap), 2 only the others.
template <bool HIZ>
__global__ __launch_bounds__(HIZ ? 1024 : 256) void splat_pipe_kernel(const float *__restrict__ xyz, long long n,
                                                                      CamSet cams, int B, int W, int H,
                                                                      unsigned long long *__restrict__ keys,
                                                                      const unsigned short *__restrict__ hiz_g, int nbx,
                                                                      int nblocks, int sub_mod, int sub_sel,
                                                                      unsigned long long *stats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short hiz[];
    if (HIZ) {
        for (int i = threadIdx.x; i < nblocks; i += blockDim.x) hiz[i] = hiz_g[i];
        __syncthreads();
    }
    const long long npx = (long long)W * H;
    const long long groups = n / PTS_PER_THREAD;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const float4 *xyz4 = reinterpret_cast<const float4 *>(xyz);
    unsigned st_local[3] = {0, 0, 0};

    auto wanted = [&](long long g) {
        if (sub_sel == 0) return true;
        const bool boot = ((g >> 8) % sub_mod) == 0;
        return sub_sel == 1 ? boot : !boot;
    };
    auto advance = [&](long long g) {            // next group of this thread that belongs to the pass
        while (g < groups && !wanted(g)) g += nthreads;
        return g;
    };

    long long pidx[4] = {-1, -1, -1, -1};        // pending atomics: index into keys (camera offset included)
    unsigned long long pkey[4] = {0, 0, 0, 0};
    long long g = advance(tid0);
    float4 a, b, c;
    if (g < groups) {
        a = xyz4[3 * g + 0];
        b = xyz4[3 * g + 1];
        c = xyz4[3 * g + 2];
    }
    while (g < groups) {
        const long long gn = advance(g + nthreads);
        float4 an = a, bn = b, cn = c;
        if (gn < groups) {                        // 1. next group's loads first
            an = xyz4[3 * gn + 0];
            bn = xyz4[3 * gn + 1];
            cn = xyz4[3 * gn + 2];
        }
        const float px[4] = {a.x, a.w, b.z, c.y};
        const float py[4] = {a.y, b.x, b.w, c.z};
        const float pz[4] = {a.z, b.y, c.x, c.w};
        const unsigned id0 = (unsigned)(g * PTS_PER_THREAD);
        for (int cam = 0; cam < B; ++cam) {
            int pix[4];
            unsigned long long key[4], seen[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {         // 2. project + early-z reads
                float d;
                int xx, yy;
                pix[k] = project_one(px[k], py[k], pz[k], cams.m[cam], W, H, d, xx, yy);
                if (stats && pix[k] >= 0) st_local[0]++;
                if (HIZ) {
                    if (hiz_reject(hiz[pix[k] >= 0 ? (yy >> 2) * nbx + (xx >> 2) : 0], d)) pix[k] = -1;
                }
                if (stats && pix[k] >= 0) st_local[1]++;
                key[k] = ((unsigned long long)__float_as_uint(d) << 32) | (id0 + k);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                seen[k] = pix[k] >= 0 ? peek_key<MODE_AGENT>(keys + cam * npx + pix[k]) : 0ull;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)           // 3. the previous step's atomics
                if (pidx[k] >= 0) fold_key<MODE_AGENT>(keys + pidx[k], pkey[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {         // 4. decide; keys only decrease, so a stale read is conservative
                const bool win = key[k] < seen[k];
                pidx[k] = win ? cam * npx + pix[k] : -1;
                pkey[k] = key[k];
                if (stats && win) st_local[2]++;
            }
        }
        a = an;
        b = bn;
        c = cn;
        g = gn;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (pidx[k] >= 0) fold_key<MODE_AGENT>(keys + pidx[k], pkey[k]);
    // tail (n % 4 points): left to the pass that takes "the others"
    if (sub_sel != 1) {
        unsigned sink = 0;
        for (long long i = groups * PTS_PER_THREAD + tid0; i < n; i += nthreads) {
            const float px[1] = {xyz[3 * i + 0]}, py[1] = {xyz[3 * i + 1]}, pz[1] = {xyz[3 * i + 2]};
            for (int cam = 0; cam < B; ++cam)
                splat_points<MODE_AGENT, 1>(px, py, pz, (unsigned)i, 1, cams.m[cam], W, H, keys + cam * npx, sink);
        }
    }
    if (stats)
        for (int i = 0; i < 3; ++i) atomicAdd(stats + (HIZ ? 4 : 0) + i, (unsigned long long)st_local[i]);
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
__global__ __launch_bounds__(256) void splat_resolve_kernel(unsigned long long *__restrict__ keys, int W, int H,
                                                            int levels, ResolveOut out, int tiles_x,
                                                            int tiles_y, int copies, int *__restrict__ prev_idx,
                                                            SplatHeader *hdr)
{
    __shared__ unsigned long long s1[256], s2[64], s3[16];
    const int cam = blockIdx.y;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int t = threadIdx.x;
    const int qx = t & 15, qy = t >> 4;
    const int x0 = tx * 32 + qx * 2, y0 = ty * 32 + qy * 2;
    const long long npx0 = (long long)W * H;
    unsigned long long *kc = keys + (long long)cam * copies * npx0;

    unsigned long long k[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = x0 + dx, y = y0 + dy;
            unsigned long long v = EMPTY_KEY;
            if (x < W && y < H) {
                const long long off = (long long)y * W + x;
                for (int c = 0; c < copies; ++c) {   // min over the per-XCD images
                    v = kmin(v, kc[c * npx0 + off]);
                    kc[c * npx0 + off] = EMPTY_KEY;  // leave the workspace clean for the next frame
                }
                emit(out, 0, cam * npx0 + off, v);
                if (prev_idx) prev_idx[off] = v == EMPTY_KEY ? -1 : (int)(unsigned)(v & 0xffffffffull);   // next frame's seeds
            }
            k[dy][dx] = v;
        }
    if (hdr && blockIdx.x == 0 && blockIdx.y == 0 && t == 0) {
        hdr->valid = 1;
        hdr->W = W;
        hdr->H = H;
        ((int *)((char *)hdr + 192))[0] = 0;       // list lengths of the cell-ordered passes (CELL_COUNTER_OFFSET)
        ((int *)((char *)hdr + 192))[1] = 0;
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
int g_splat_subset = 8;        // MODE_HIZ bootstrap pass over every g_splat_subset-th 1024-point chunk (0 = seeds only).
                               // Measured at 30 M points: 0.307 ms seeds only, 0.252 / 0.233 / 0.235 / 0.260 ms at 4 / 8 / 16 / 32
int g_splat_pipe = 0;          // 1: software-pipelined point pass (measured slower: 0.43 vs 0.39 ms in MODE_AGENT)
int g_splat_stats = 0;         // debug: accumulate counters in the workspace header (u64 at byte 64: pass A visible /
                               // survivors / atomics / -, pass B visible / survivors / atomics)

// Workspace layout (fixed by the (B, W, H) it was sized for; one workspace serves one such triple):
//   [header 256 B][key images: min(B,8) x 8 x W*H x 8 B][hi-z bounds: ceil(W/4)*ceil(H/4) x 4 B][previous winners: W*H x 4 B]
struct WsLayout {
    SplatHeader *hdr;
    unsigned long long *keys;
    unsigned short *hiz;       // 16-bit far bounds (the region keeps its 4 bytes per block)
    int *prev;
    int nbx, nby;
    size_t total;
};

WsLayout ws_layout(void *ws, int B, int W, int H)
{
    WsLayout L;
    const int nb = B < MAX_CAMS ? B : MAX_CAMS;
    char *p = (char *)ws;
    L.hdr = (SplatHeader *)p;
    size_t off = HEADER_BYTES;
    L.keys = (unsigned long long *)(p + off);
    off += (size_t)nb * XCD_COPIES * W * H * sizeof(unsigned long long);
    L.nbx = ceil_div(W, 4);
    L.nby = ceil_div(H, 4);
    L.hiz = (unsigned short *)(p + off);
    off += ((size_t)L.nbx * L.nby * sizeof(float) + 255) / 256 * 256;
    L.prev = (int *)(p + off);
    off += (size_t)W * H * sizeof(int);
    L.total = (off + 255) / 256 * 256;
    return L;
}

constexpr size_t HIZ_LDS_LIMIT = 150 * 1024;   // of the 160 KiB per CU

int g_splat_near = 12;         // cell path: pass A takes chunks nearer than the depth at which a pixel expects this many points
int g_splat_cells = 1;         // 0: ignore the cell-ordered copy (A/B)
int g_splat_l1 = 1;            // early-z reads of the cell-ordered passes through the L1 (0: L2, A/B)
int g_splat_cells_sub = 32;    // list A also takes every n-th chunk (0: none): a first bound where nothing is near
int g_splat_seeds = 1;         // 0: no warm start from the previous frame's winners (A/B)

int project_and_resolve(const float *xyz, int64_t n, const float *M_host, int B, int W, int H, int levels,
                        int32_t *const *idx_levels, float *const *depth_levels, int level_base,
                        const WsLayout &ws, bool allow_hiz, hipStream_t stream, const CellCloud *cells = nullptr)
{
    unsigned long long *keys = ws.keys;
    const size_t hiz_bytes = (((size_t)ws.nbx * ws.nby * sizeof(unsigned short)) + 15) & ~(size_t)15;
    const bool use_hiz = g_splat_mode == MODE_HIZ && allow_hiz && B == 1 && n > 0 && hiz_bytes <= HIZ_LDS_LIMIT &&
                         ((uintptr_t)xyz % 16) == 0;
    const int mode = g_splat_mode == MODE_HIZ ? MODE_AGENT : g_splat_mode;
    const int copies = mode == MODE_XCD ? XCD_COPIES : 1;
    for (int b0 = 0; b0 < B; b0 += MAX_CAMS) {
        const int nb = (B - b0) < MAX_CAMS ? (B - b0) : MAX_CAMS;
        CamSet cams;
        memset(&cams, 0, sizeof(cams));
        memcpy(cams.m, M_host + 16 * (size_t)b0, sizeof(float) * 16 * (size_t)nb);
        const int vec_ok = ((uintptr_t)xyz % 16) == 0;
        unsigned long long *stats = g_splat_stats ? (unsigned long long *)((char *)ws.hdr + 64) : nullptr;
        if (use_hiz) {
            // bootstrap: a strided 1/sub of the cloud (only when the vector path is usable) on top of the seeds,
            // so that nearly every pixel is covered before the bounds are taken
            const int sub = (vec_ok && g_splat_subset > 1 && n >= (1 << 20)) ? g_splat_subset : 0;
            static int n_cu_c = 0;
            if (!n_cu_c) {
                int dev = 0;
                hipDeviceProp_t prop;
                n_cu_c = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                          prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
            }
            const bool use_cells = cells && g_splat_cells;
            const int *counts = (const int *)((const char *)ws.hdr + CELL_COUNTER_OFFSET);
            CellCloud cc_none;
            memset(&cc_none, 0, sizeof(cc_none));
            const int pix_blocks = ceil_div(W * H, 256);
            // seeds (previous winners re-projected) + classification of the chunks of the cell-ordered cloud
            hipLaunchKernelGGL(splat_seed_kernel, dim3(pix_blocks + (use_cells ? ceil_div(cells->nchunks, 256) : 0)), dim3(256),
                               0, stream, xyz, (long long)n, cams, W, H, keys, ws.hdr, g_splat_seeds ? ws.prev : (int *)nullptr,
                               use_cells ? *cells : cc_none,
                               pix_blocks, g_splat_cells_sub, (float)g_splat_near);
            READ_CHECK_LAUNCH();
            if (use_cells) {
                // pass A: near chunks + every (2 sub)-th chunk, straight early-z splat, one wave per chunk
                auto kern_a = g_splat_l1 ? splat_cells_kernel<false, true> : splat_cells_kernel<false, false>;
                hipLaunchKernelGGL(kern_a, dim3((unsigned)(n_cu_c * 8)), dim3(256), 0, stream, *cells, cams,
                                   W, H, keys, (const unsigned short *)nullptr, ws.nbx, ws.nby, counts, stats);
                READ_CHECK_LAUNCH();
            } else if (sub) {
                int64_t blocks = ceil_div64(ceil_div64(n, PTS_PER_THREAD), 256);
                if (blocks > 256 * 8) blocks = 256 * 8;
                hipLaunchKernelGGL(splat_project_kernel<MODE_AGENT>, dim3((unsigned)blocks), dim3(256), 0, stream, xyz,
                                   (long long)n, cams, 1, W, H, keys, vec_ok, (unsigned *)nullptr, sub, stats);
                READ_CHECK_LAUNCH();
            }
            hipLaunchKernelGGL(splat_hiz_kernel, dim3(ceil_div(ws.nbx * ws.nby, 256)), dim3(256), 0, stream, keys, W, H,
                               ws.nbx, ws.nby, ws.hiz);
            READ_CHECK_LAUNCH();
            static bool attr_set = false;
            if (!attr_set) {
                READ_CHECK_HIP(hipFuncSetAttribute((const void *)splat_pipe_kernel<true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIZ_LDS_LIMIT));
                READ_CHECK_HIP(hipFuncSetAttribute((const void *)splat_project_hiz_kernel,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)HIZ_LDS_LIMIT));
                attr_set = true;
            }
            static int n_cu = 0;
            if (!n_cu) {
                int dev = 0;
                hipDeviceProp_t prop;
                n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                        prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
            }
            int64_t blocks = ceil_div64(ceil_div64(n, PTS_PER_THREAD), 1024);
            const int per_cu = 2 * hiz_bytes <= 150 * 1024 ? 2 : 1;     // two workgroups per CU when their bounds fit
            if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
            if (use_cells) {
                static bool attr_c = false;
                if (!attr_c) {
                    auto kb0 = splat_cells_kernel<true, false>;
                    auto kb1 = splat_cells_kernel<true, true>;
                    READ_CHECK_HIP(hipFuncSetAttribute((const void *)kb0, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)HIZ_LDS_LIMIT));
                    READ_CHECK_HIP(hipFuncSetAttribute((const void *)kb1, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)HIZ_LDS_LIMIT));
                    attr_c = true;
                }
                auto kern_b = g_splat_l1 ? splat_cells_kernel<true, true> : splat_cells_kernel<true, false>;
                hipLaunchKernelGGL(kern_b, dim3((unsigned)(n_cu * per_cu)), dim3(1024), hiz_bytes, stream,
                                   *cells, cams, W, H, keys, ws.hiz, ws.nbx, ws.nby, counts, stats);
            } else if (g_splat_pipe && !sub)
                hipLaunchKernelGGL(splat_pipe_kernel<true>, dim3((unsigned)blocks), dim3(1024), hiz_bytes, stream, xyz,
                                   (long long)n, cams, 1, W, H, keys, ws.hiz, ws.nbx, ws.nbx * ws.nby, sub, sub ? 2 : 0, stats);
            else
                hipLaunchKernelGGL(splat_project_hiz_kernel, dim3((unsigned)blocks), dim3(1024), hiz_bytes, stream, xyz,
                                   (long long)n, cams, W, H, keys, vec_ok, ws.hiz, ws.nbx, ws.nbx * ws.nby, sub, stats);
            READ_CHECK_LAUNCH();
        } else if (n > 0) {
            const int64_t work = ceil_div64(n, PTS_PER_THREAD);
            // HBM-bound stream: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
            int64_t blocks = ceil_div64(work, 256);
            if (blocks > 256 * 8) blocks = 256 * 8;
            if (mode == MODE_AGENT && vec_ok && g_splat_pipe) {
                hipLaunchKernelGGL(splat_pipe_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, xyz, (long long)n,
                                   cams, nb, W, H, keys, (const unsigned short *)nullptr, 0, 0, 0, 0, stats);
                READ_CHECK_LAUNCH();
            } else {
            auto kern = mode == MODE_XCD ? splat_project_kernel<MODE_XCD>
                        : mode == MODE_AGENT ? splat_project_kernel<MODE_AGENT>
                        : mode == MODE_SYS ? splat_project_kernel<MODE_SYS>
                        : mode == MODE_PEEK ? splat_project_kernel<MODE_PEEK>
                        : mode == MODE_PEEK_L1 ? splat_project_kernel<MODE_PEEK_L1>
                        : mode == MODE_ATOM ? splat_project_kernel<MODE_ATOM> : splat_project_kernel<MODE_NOZ>;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, xyz, (long long)n, cams, nb, W, H,
                               keys, vec_ok, (unsigned *)nullptr, 0, stats);
            READ_CHECK_LAUNCH();
            }
        }
        ResolveOut out;
        memset(&out, 0, sizeof(out));
        for (int l = 0; l < levels; ++l) {
            const size_t lpx = (size_t)level_dim(W, l) * level_dim(H, l);
            if (idx_levels && idx_levels[level_base + l]) out.idx[l] = idx_levels[level_base + l] + lpx * b0;
            if (depth_levels && depth_levels[level_base + l])
                out.depth[l] = depth_levels[level_base + l] + lpx * b0;
        }
        const int tiles_x = ceil_div(W, 32), tiles_y = ceil_div(H, 32);
        // the winners of a single-camera frame seed the next frame rendered through this workspace
        const bool keep = g_splat_mode == MODE_HIZ && allow_hiz && B == 1;
        hipLaunchKernelGGL(splat_resolve_kernel, dim3(tiles_x * tiles_y, nb), dim3(256), 0, stream, keys, W, H,
                           levels, out, tiles_x, tiles_y, copies, keep ? ws.prev : (int *)nullptr,
                           keep ? ws.hdr : (SplatHeader *)nullptr);
        READ_CHECK_LAUNCH();
    }
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
void splat_set_pipe(int v) { g_splat_pipe = v; }
int splat_set_mode(int m)
{
    if (m < MODE_XCD || m > MODE_HIZ) return READ_EINVAL;
    g_splat_mode = m;
    return READ_OK;
}
}

__global__ __launch_bounds__(64) void splat_header_clear_kernel(int *hdr)
{
    hdr[threadIdx.x] = 0;       // 64 ints = the 256-byte header: no previous frame
}

extern "C" int read_splat_workspace_init(void *ws, size_t ws_bytes, void *stream)
{
    READ_CHECK_ARG(ws && ws_bytes % 8 == 0, "read_splat_workspace_init: workspace null or not a multiple of 8 bytes");
    // everything becomes EMPTY (all ones); the header is then zeroed: "no previous frame"
    READ_CHECK_ARG((uintptr_t)ws % 16 == 0, "read_splat_workspace_init: workspace must be 16-byte aligned");
    const long long count = (long long)(ws_bytes / 8);
    if (count == 0) return READ_OK;
    hipLaunchKernelGGL(fill_keys_kernel, dim3((unsigned)ceil_div64(count, 256)), dim3(256), 0, as_stream(stream),
                       (unsigned long long *)ws, count);
    READ_CHECK_LAUNCH();
    if (ws_bytes >= HEADER_BYTES) {
        hipLaunchKernelGGL(splat_header_clear_kernel, dim3(1), dim3(64), 0, as_stream(stream), (int *)ws);
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
    READ_CHECK_ARG(ws && (uintptr_t)ws % 16 == 0, "read_splat_forward: workspace null or misaligned");
    if (ws_bytes < read_splat_workspace_bytes(B, W, H)) {
        set_error("read_splat_forward: workspace %zu < %zu bytes", ws_bytes, read_splat_workspace_bytes(B, W, H));
        return READ_ENOMEM;
    }
    const WsLayout L = ws_layout(ws, B, W, H);
    hipStream_t s = as_stream(stream);
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
// cell-ordered cloud
// ---------------------------------------------------------------------------------------------------------------
namespace {

size_t cells_chunks(int64_t n) { return (size_t)((n + CELL_CHUNK - 1) / CELL_CHUNK); }

struct CellOffsets {
    size_t xyz, ids, aabb, list_a, list_b, total;
};
CellOffsets cell_offsets(int64_t n)
{
    const size_t nc = cells_chunks(n);
    CellOffsets o;
    o.xyz = CELL_HEADER_BYTES;
    o.ids = o.xyz + nc * CELL_CHUNK * 3 * sizeof(float);
    o.aabb = o.ids + nc * CELL_CHUNK * sizeof(unsigned);
    o.list_a = o.aabb + nc * 8 * sizeof(float);                      // per-frame scratch (written by the passes)
    o.list_b = o.list_a + ((nc * sizeof(int) + 15) & ~(size_t)15);
    o.total = o.list_b + nc * sizeof(CellEntryB);
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

}  // namespace

extern "C" size_t read_splat_cells_bytes(int64_t n)
{
    if (n < 1 || n > 0xFFFFFFFEll) return 0;
    return cell_offsets(n).total;
}

// Host-side build (once per cloud): Morton order over a 1024^3 grid of the bounding box (stable LSD radix sort, so equal
// codes keep ascending ids), chunks of 1024 consecutive points with their exact bounding boxes.
extern "C" int read_splat_cells_build_host(const float *xyz, int64_t n, void *blob, size_t blob_bytes)
{
    READ_CHECK_ARG(xyz && blob, "read_splat_cells_build_host: null pointer");
    READ_CHECK_ARG(n >= 1 && n <= 0xFFFFFFFEll, "read_splat_cells_build_host: n out of range");
    const CellOffsets o = cell_offsets(n);
    READ_CHECK_ARG(blob_bytes >= o.total, "read_splat_cells_build_host: buffer %zu < %zu bytes", blob_bytes, o.total);
    float lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * i + k];
            READ_CHECK_ARG(v == v && v - v == 0.0f, "read_splat_cells_build_host: point %lld is not finite", (long long)i);
            lo[k] = v < lo[k] ? v : lo[k];
            hi[k] = v > hi[k] ? v : hi[k];
        }
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) ext = (hi[k] - lo[k]) > ext ? (hi[k] - lo[k]) : ext;
    const float scale = ext > 0.0f ? 1023.999f / ext : 0.0f;
    std::vector<uint64_t> a((size_t)n), b((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) {
            float t = (xyz[3 * i + k] - lo[k]) * scale;
            q[k] = t <= 0.0f ? 0u : (t >= 1023.0f ? 1023u : (uint32_t)t);
        }
        const uint64_t code = spread10(q[0]) | ((uint64_t)spread10(q[1]) << 1) | ((uint64_t)spread10(q[2]) << 2);
        a[(size_t)i] = (code << 32) | (uint64_t)(uint32_t)i;
    }
    for (int pass = 0; pass < 3; ++pass) {                      // 30 code bits, 10 per pass
        const int shift = 32 + 10 * pass;
        size_t hist[1025] = {0};
        for (size_t i = 0; i < (size_t)n; ++i) ++hist[((a[i] >> shift) & 1023u) + 1];
        for (int k = 0; k < 1024; ++k) hist[k + 1] += hist[k];
        for (size_t i = 0; i < (size_t)n; ++i) b[hist[(a[i] >> shift) & 1023u]++] = a[i];
        a.swap(b);
    }
    char *base = (char *)blob;
    memset(base, 0, CELL_HEADER_BYTES);
    float *xs = (float *)(base + o.xyz);
    unsigned *ids = (unsigned *)(base + o.ids);
    float *bb = (float *)(base + o.aabb);
    const size_t nc = cells_chunks(n), padded = nc * CELL_CHUNK;
    for (size_t i = 0; i < padded; ++i) {
        const uint32_t id = (uint32_t)(a[i < (size_t)n ? i : (size_t)n - 1] & 0xffffffffu);   // tail: copies of the last point
        ids[i] = id;
        xs[3 * i + 0] = xyz[3 * (size_t)id + 0];
        xs[3 * i + 1] = xyz[3 * (size_t)id + 1];
        xs[3 * i + 2] = xyz[3 * (size_t)id + 2];
    }
    for (size_t c = 0; c < nc; ++c) {
        float mn[3], mx[3];
        for (int k = 0; k < 3; ++k) mn[k] = mx[k] = xs[3 * c * CELL_CHUNK + k];
        for (size_t i = c * CELL_CHUNK; i < (c + 1) * CELL_CHUNK; ++i)
            for (int k = 0; k < 3; ++k) {
                const float v = xs[3 * i + k];
                mn[k] = v < mn[k] ? v : mn[k];
                mx[k] = v > mx[k] ? v : mx[k];
            }
        float *r = bb + 8 * c;
        r[0] = mn[0]; r[1] = mn[1]; r[2] = mn[2]; r[3] = mx[0]; r[4] = mx[1]; r[5] = mx[2]; r[6] = r[7] = 0.0f;
    }
    CellHeader *h = (CellHeader *)base;
    h->n = n;
    h->nchunks = (int)nc;
    h->version = 1;
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

extern "C" int read_splat_forward_cells(const float *xyz, void *cells, int64_t n, const float *M_host, int B,
                                        int W, int H, int levels, int32_t *const *idx_levels,
                                        float *const *depth_levels, void *ws, size_t ws_bytes, void *stream)
{
    const int mask = levels >= 1 && levels <= READ_MAX_LEVELS ? (1 << (levels - 1)) - 1 : 0;
    // the cell-ordered passes serve the single-camera, pyramid-identity case (the per-frame render path); everything
    // else goes through the plain pass
    if (!cells || B != 1 || n < (1 << 20) || ((W | H) & mask) != 0 || !xyz || !M_host || !ws || W < 1 || H < 1)
        return read_splat_forward(xyz, n, M_host, B, W, H, levels, idx_levels, depth_levels, ws, ws_bytes, stream);
    READ_CHECK_ARG(n <= 0xFFFFFFFEll, "read_splat_forward_cells: point ids must fit 32 bits");
    READ_CHECK_ARG(levels >= 1 && levels <= READ_MAX_LEVELS, "read_splat_forward_cells: levels must be 1..%d",
                   READ_MAX_LEVELS);
    READ_CHECK_ARG(idx_levels || depth_levels, "read_splat_forward_cells: no outputs requested");
    READ_CHECK_ARG((long long)W * H < (1ll << 31), "read_splat_forward_cells: image too large");
    READ_CHECK_ARG((uintptr_t)ws % 16 == 0 && (uintptr_t)cells % 16 == 0, "read_splat_forward_cells: misaligned pointer");
    if (ws_bytes < read_splat_workspace_bytes(B, W, H)) {
        set_error("read_splat_forward_cells: workspace %zu < %zu bytes", ws_bytes, read_splat_workspace_bytes(B, W, H));
        return READ_ENOMEM;
    }
    const CellOffsets o = cell_offsets(n);
    CellCloud cc;
    cc.hdr = (const CellHeader *)cells;
    cc.xyz = (const float *)((const char *)cells + o.xyz);
    cc.ids = (const unsigned *)((const char *)cells + o.ids);
    cc.aabb = (const float *)((const char *)cells + o.aabb);
    cc.list_a = (int *)((char *)cells + o.list_a);
    cc.list_b = (CellEntryB *)((char *)cells + o.list_b);
    cc.nchunks = (int)cells_chunks(n);
    const WsLayout L = ws_layout(ws, B, W, H);
    return project_and_resolve(xyz, n, M_host, B, W, H, levels, idx_levels, depth_levels, 0, L, true, as_stream(stream), &cc);
}

namespace readhip {
void splat_set_near(int v) { g_splat_near = v < 1 ? 1 : v; }
void splat_set_cells(int v) { g_splat_cells = v; }
void splat_set_seeds(int v) { g_splat_seeds = v; }
void splat_set_l1(int v) { g_splat_l1 = v; }
void splat_set_cells_sub(int v) { g_splat_cells_sub = v < 0 ? 0 : v; }
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
