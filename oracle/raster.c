/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (read_amd/).  Allowed users: tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg.
 *
 * CPU restatement of READ's z-buffered point projector, in plain C, fp32,
 * compiled with -ffp-contract=off (no FMA) and IEEE division.
 *
 * Follows (reference @ /root/reference):
 *   MyRender/CloudProjection/point_render.cu:110-121  math::MatrixMul  (4 dot products, divide by w)
 *   MyRender/CloudProjection/helper_math.h:1252-1255   dot(float4,float4) = a.x*b.x + a.y*b.y + a.z*b.z + a.w*b.w
 *   MyRender/CloudProjection/helper_math.h:1023        float4 / float
 *   MyRender/CloudProjection/point_render.cu:125-167  DepthProject (clip, pixel, z-test)
 *   MyRender/CloudProjection/point_render.cu:169-200  GPU_PCPR     (zero init, per-batch offset)
 *   src/READ/gl/myrender.py:32-40                      5 scales, w=int(W*0.5^i), h=int(H*0.5^i)
 *
 * Canonical deterministic semantics: the reference resolves the
 * per-pixel z-test with a (defective) spin lock whose outcome depends on thread
 * arrival order.  This oracle defines the unique order-independent outcome that the
 * reference produces when threads arrive in point order and nothing is dropped:
 *   per pixel: minimum fp32 depth d, ties -> minimum point index; empty -> (0, 0).
 * Realised exactly as the reference's critical section executed serially in index
 * order:  if (depth[ind] > d || depth[ind] == 0) { depth[ind] = d; index[ind] = i; }
 * with one documented departure: a point whose d is exactly 0.0f (on the near
 * plane; measure-zero) is kept as a normal minimum instead of being treated as
 * "empty" by later points, and non-finite NDC coordinates are rejected.
 *
 * Parity pin: the reference repo holds no golden vectors for this path
 * (SURVEY.md §4, §8c).  This restatement is pinned instead against the reference's
 * own kernel source compiled for the CPU and executed serially (oracle/_ref,
 * built by oracle/build_ref.sh; tests/test_oracle_raster.py).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* One point through point_render.cu:135-147.  Returns 1 and fills (*pix, *depth)
 * when accepted. */
static inline int project_point(const float *p, const float *M, int W, int H,
                                int *pix, float *depth)
{
    const float x = p[0], y = p[1], z = p[2];
    /* dot(row_k, (x,y,z,1)) left to right, helper_math.h:1252 */
    const float c0 = M[0] * x + M[1] * y + M[2] * z + M[3] * 1.0f;
    const float c1 = M[4] * x + M[5] * y + M[6] * z + M[7] * 1.0f;
    const float c2 = M[8] * x + M[9] * y + M[10] * z + M[11] * 1.0f;
    const float c3 = M[12] * x + M[13] * y + M[14] * z + M[15] * 1.0f;
    /* ans / ans.w, point_render.cu:119 */
    const float nx = c0 / c3, ny = c1 / c3, nz = c2 / c3;
    if (!(nx == nx) || !(ny == ny) || !(nz == nz)) return 0;      /* NaN: rejected (canonical) */
    if (nx < -1.0f || nx > 1.0f || ny < -1.0f || ny > 1.0f || nz < -1.0f || nz > 1.0f) return 0; /* :139 */
    /* :141-143.  The literal 0.5 is a double in the reference; x*0.5 is exact in
     * either precision, so fp32 evaluation is bit-identical. */
    const float u = ((float)W * (nx + 1.0f)) * 0.5f;
    const float v = ((float)H * (1.0f - ny)) * 0.5f;
    const float d = (nz + 1.0f) * 0.5f;
    const int xx = (int)u, yy = (int)v;                            /* :145-146 truncation */
    if (xx < 0 || xx >= W || yy < 0 || yy >= H) return 0;          /* :147 */
    *pix = yy * W + xx;
    *depth = d;
    return 1;
}

/* The projection alone for every point: pixel[i] = yy * W + xx or -1 (rejected), depth[i] (accepted points only; 0 otherwise).
 * The checker of read_splat_project_points — in particular of the shared-reciprocal form of the three divisions on the device. */
void oracle_project_points(const float *xyz, int64_t n, const float *M, int W, int H, int32_t *pixel, float *depth, int threads)
{
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) {
        int pix = -1;
        float d = 0.0f;
        if (!project_point(xyz + 3 * i, M, W, H, &pix, &d)) {
            pix = -1;
            d = 0.0f;
        }
        pixel[i] = pix;
        depth[i] = d;
    }
}

/* Single level, single camera; serial in point order == canonical semantics.
 * out_index is int32 (the reference stores float(i); see raster_index_to_float). */
void oracle_raster_level(const float *xyz, int64_t n, const float *M, int W, int H,
                         int32_t *out_index, float *out_depth)
{
    const size_t npx = (size_t)W * (size_t)H;
    uint8_t *set = (uint8_t *)__builtin_malloc(npx);
    memset(set, 0, npx);
    memset(out_index, 0, npx * sizeof(int32_t));                   /* torch::zeros, :176-177 */
    memset(out_depth, 0, npx * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        int pix; float d;
        if (!project_point(xyz + 3 * i, M, W, H, &pix, &d)) continue;
        if (!set[pix] || out_depth[pix] > d) {                     /* :155 with the d==0 departure */
            set[pix] = 1;
            out_depth[pix] = d;
            out_index[pix] = (int32_t)i;
        }
    }
    __builtin_free(set);
}

/* Packed-key form used to merge per-thread partial images: key = depth_bits<<32 | idx;
 * for non-negative floats the unsigned order of the bits is the float order, so
 * min(key) == (min depth, then min index). */
static inline uint64_t make_key(float d, int64_t i)
{
    uint32_t b; memcpy(&b, &d, 4);
    return ((uint64_t)b << 32) | (uint32_t)i;
}

/* Multi-threaded variant (OpenMP when compiled with -fopenmp; otherwise serial).
 * Bit-identical to oracle_raster_level; used as the CPU baseline. */
void oracle_raster_level_mt(const float *xyz, int64_t n, const float *M, int W, int H,
                            int32_t *out_index, float *out_depth, int nthreads)
{
    const size_t npx = (size_t)W * (size_t)H;
    if (nthreads < 1) nthreads = 1;
    uint64_t *keys = (uint64_t *)__builtin_malloc(npx * sizeof(uint64_t) * (size_t)nthreads);
    memset(keys, 0xFF, npx * sizeof(uint64_t) * (size_t)nthreads);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int t = 0, nt = 1;
#endif
        uint64_t *k = keys + (size_t)t * npx;
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t i = lo; i < hi; ++i) {
            int pix; float d;
            if (!project_point(xyz + 3 * i, M, W, H, &pix, &d)) continue;
            const uint64_t key = make_key(d, i);
            if (key < k[pix]) k[pix] = key;
        }
    }
    for (size_t p = 0; p < npx; ++p) {
        uint64_t best = keys[p];
        for (int t = 1; t < nthreads; ++t) {
            const uint64_t c = keys[(size_t)t * npx + p];
            if (c < best) best = c;
        }
        if (best == ~(uint64_t)0) { out_index[p] = 0; out_depth[p] = 0.0f; }
        else {
            const uint32_t b = (uint32_t)(best >> 32);
            memcpy(&out_depth[p], &b, 4);
            out_index[p] = (int32_t)(uint32_t)best;
        }
    }
    __builtin_free(keys);
}

/* All `levels` scales of one camera, each rasterised DIRECTLY at its own size
 * (myrender.py:32-40: w=int(W*0.5^i), h=int(H*0.5^i)) — deliberately not via the
 * 2x2 key-min pyramid the HIP path uses, so that the pyramid identity
 *  is itself under test.  Outputs are concatenated level by level. */
void oracle_raster_multiscale(const float *xyz, int64_t n, const float *M, int W, int H,
                              int levels, int32_t *out_index, float *out_depth, int nthreads)
{
    size_t off = 0;
    for (int l = 0; l < levels; ++l) {
        const int w = (int)((double)W * pow(0.5, (double)l));
        const int h = (int)((double)H * pow(0.5, (double)l));
        if (nthreads > 1) oracle_raster_level_mt(xyz, n, M, w, h, out_index + off, out_depth + off, nthreads);
        else oracle_raster_level(xyz, n, M, w, h, out_index + off, out_depth + off);
        off += (size_t)w * (size_t)h;
    }
}

/* ------------------------------------------------------------------------------------------------
 * GL twin features (READ/gl/programs.py:121-198 vertex shader, READ/gl/render.py:52-85): point sizes,
 * perspective ("ps") splats, point discard and clip-space perturbation.  The reference implements these
 * in OpenGL, which cannot run here and whose point rasterisation is implementation-defined at pixel
 * boundaries; this is the canonical restatement the HIP kernel is held to (parity unpinned against GL):
 *   - clip = M * (x,y,z,1); clip.x += perturb.x; clip.y += perturb.y           (programs.py:125-128)
 *   - the point is clipped by its CENTRE (same tests as the 1-px rule above)
 *   - size s = point_size, or max(min_point_size, point_size / clip.z) for "ps" tokens (:183-192),
 *     never below 1; the point covers pixel columns floor(u - (s-1)/2) .. floor(u + (s-1)/2) and rows
 *     floor(v - (s-1)/2) .. floor(v + (s-1)/2) of the level it is drawn into (s = 1: exactly the 1-px
 *     rule), clipped to the viewport; every covered pixel gets the point's depth and id (flat shading)
 *   - discarded points (a_discard == 1, :102,:236) are not drawn
 *   - z-test: min depth, ties -> min id (GL_LESS in draw order)
 * Seeded variants of the two augmentations (READ/datasets/dynamic.py:235-239 draws them from numpy's
 * global RNG): point i is dropped iff rnd(i, seed, 0) < drop_threshold; perturb = amp * (u01 - 0.5)
 * with u01 = (rnd(i, seed, 1 | 2) >> 8) * 2^-24.
 */
static inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static inline uint32_t rnd32(uint32_t i, uint32_t seed, uint32_t k)
{
    return hash32(i ^ hash32(seed + 0x9e3779b9u * (k + 1u)));
}

typedef struct oracle_gl_opts {
    float point_size;            /* pN / psN */
    int relative;                /* 1: "ps" token (size / clip.z) */
    float min_point_size;
    const uint8_t *discard;      /* optional N bytes, 1 = not drawn */
    uint32_t drop_threshold;     /* seeded drop: rnd < threshold (0 = off) */
    uint32_t drop_seed;
    const float *perturb;        /* optional N x 2 clip-space offsets */
    float perturb_amp;           /* seeded perturbation amplitude (0 = off) */
    uint32_t perturb_seed;
    const float *point_sizes;    /* optional per-point sizes (a_point_size), used when point_size < 1 (programs.py:183-187) */
} oracle_gl_opts;

void oracle_raster_level_gl(const float *xyz, int64_t n, const float *M, int W, int H, const oracle_gl_opts *o,
                            int32_t *out_index, float *out_depth)
{
    const size_t npx = (size_t)W * (size_t)H;
    uint64_t *keys = (uint64_t *)__builtin_malloc(npx * sizeof(uint64_t));
    memset(keys, 0xFF, npx * sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i) {
        if (o->discard && o->discard[i]) continue;
        if (o->drop_threshold && rnd32((uint32_t)i, o->drop_seed, 0) < o->drop_threshold) continue;
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        float c0 = M[0] * x + M[1] * y + M[2] * z + M[3] * 1.0f;
        float c1 = M[4] * x + M[5] * y + M[6] * z + M[7] * 1.0f;
        const float c2 = M[8] * x + M[9] * y + M[10] * z + M[11] * 1.0f;
        const float c3 = M[12] * x + M[13] * y + M[14] * z + M[15] * 1.0f;
        if (o->perturb) { c0 = c0 + o->perturb[2 * i]; c1 = c1 + o->perturb[2 * i + 1]; }
        if (o->perturb_amp != 0.0f) {
            const float ux = (float)(rnd32((uint32_t)i, o->perturb_seed, 1) >> 8) * (1.0f / 16777216.0f);
            const float uy = (float)(rnd32((uint32_t)i, o->perturb_seed, 2) >> 8) * (1.0f / 16777216.0f);
            c0 = c0 + o->perturb_amp * (ux - 0.5f);
            c1 = c1 + o->perturb_amp * (uy - 0.5f);
        }
        const float nx = c0 / c3, ny = c1 / c3, nz = c2 / c3;
        if (!(nx == nx) || !(ny == ny) || !(nz == nz)) continue;
        if (nx < -1.0f || nx > 1.0f || ny < -1.0f || ny > 1.0f || nz < -1.0f || nz > 1.0f) continue;
        const float u = ((float)W * (nx + 1.0f)) * 0.5f;
        const float v = ((float)H * (1.0f - ny)) * 0.5f;
        const float d = (nz + 1.0f) * 0.5f;
        if ((int)u < 0 || (int)u >= W || (int)v < 0 || (int)v >= H) continue;
        float sz = o->point_size;
        if (sz < 1.0f && o->point_sizes) sz = o->point_sizes[i];
        if (o->relative) { sz = sz / c2; if (!(sz > o->min_point_size)) sz = o->min_point_size; }
        if (!(sz > 1.0f)) sz = 1.0f;
        if (sz > 4096.0f) sz = 4096.0f;
        const float half = 0.5f * (sz - 1.0f);
        int x0 = (int)floorf(u - half), x1 = (int)floorf(u + half);
        int y0 = (int)floorf(v - half), y1 = (int)floorf(v + half);
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (x1 > W - 1) x1 = W - 1;
        if (y1 > H - 1) y1 = H - 1;
        const uint64_t key = make_key(d, i);
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
                uint64_t *k = keys + (size_t)yy * W + xx;
                if (key < *k) *k = key;
            }
    }
    for (size_t p = 0; p < npx; ++p) {
        if (keys[p] == ~(uint64_t)0) { out_index[p] = 0; out_depth[p] = 0.0f; }
        else {
            const uint32_t b = (uint32_t)(keys[p] >> 32);
            memcpy(&out_depth[p], &b, 4);
            out_index[p] = (int32_t)(uint32_t)keys[p];
        }
    }
    __builtin_free(keys);
}

/* The seeded augmentation streams themselves (for tests that pass explicit arrays). */
void oracle_drop_mask(int64_t n, uint32_t threshold, uint32_t seed, uint8_t *mask)
{
    for (int64_t i = 0; i < n; ++i) mask[i] = rnd32((uint32_t)i, seed, 0) < threshold;
}
void oracle_perturb_array(int64_t n, float amp, uint32_t seed, float *out_n2)
{
    for (int64_t i = 0; i < n; ++i) {
        const float ux = (float)(rnd32((uint32_t)i, seed, 1) >> 8) * (1.0f / 16777216.0f);
        const float uy = (float)(rnd32((uint32_t)i, seed, 2) >> 8) * (1.0f / 16777216.0f);
        out_n2[2 * i] = amp * (ux - 0.5f);
        out_n2[2 * i + 1] = amp * (uy - 0.5f);
    }
}

/* The reference stores the point id in a float (point_render.cu:158): ids >= 2^24 round. */
void oracle_index_to_float(const int32_t *idx, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = (float)idx[i];
}

/* Descriptor gather, READ/models/texture.py:55-63:
 *   feat[c][y][x] = texture[c][idx[y][x]]   (texture is (C, N) channel-major; output CHW). */
void oracle_gather_chw(const float *texture_cn, int64_t n, int C, const int32_t *idx, size_t npx,
                       float *out_chw)
{
    for (int c = 0; c < C; ++c)
        for (size_t p = 0; p < npx; ++p)
            out_chw[(size_t)c * npx + p] = texture_cn[(size_t)c * (size_t)n + (size_t)idx[p]];
}

/* Backward of the gather (autograd of texture.py:61 = index_add):
 *   grad_texture[c][idx[p]] += grad_out[c][p], accumulated in pixel order. */
void oracle_gather_backward_chw(const float *grad_chw, const int32_t *idx, size_t npx, int C,
                                int64_t n, float *grad_texture_cn)
{
    for (int c = 0; c < C; ++c)
        for (size_t p = 0; p < npx; ++p)
            grad_texture_cn[(size_t)c * (size_t)n + (size_t)idx[p]] += grad_chw[(size_t)c * npx + p];
}
