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
 * built by oracle/build_ref.sh; tests/test_oracle_ref.py).
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
