// Descriptor gather / scatter-add for gfx950.
//
// Replaces PointTexture.forward (READ/models/texture.py:42-70: index_select over a (C, B*N)
// channel-major table, i.e. C scattered 4-byte reads per pixel at stride 4*N bytes) with a
// row-major N x C table (32-byte rows at C = 8): one 16-byte load + one 16-byte NHWC store per
// lane, all pyramid levels in a single launch.  Background pixels carry id 0 and therefore
// receive descriptor[0], exactly as the reference (SURVEY.md "Empty pixels sample point 0").
// Backward = scatter-add of dL/dfeat into the rows (autograd of texture.py:61).
// HBM-bound: algorithmic bytes = sum_l px_l * (4 + 4C + 4C).
#include "common.h"

using namespace readhip;

namespace {

struct LevelTable {
    const int32_t *idx[READ_MAX_LEVELS];
    float *feat[READ_MAX_LEVELS];
    long long end[READ_MAX_LEVELS];   // exclusive prefix end, in units of (pixel, quad) items
    int levels;
};

__device__ __forceinline__ float act_apply(float v, int activation)
{
    if (activation == 1) return 1.0f / (1.0f + expf(-v));
    if (activation == 2) return tanhf(v);
    return v;
}

// item = (pixel, quad of 4 channels); consecutive lanes -> consecutive quads of consecutive pixels,
// so the NHWC store is fully coalesced.
__global__ __launch_bounds__(256) void gather_forward_kernel(const float *__restrict__ rows, long long n, int C,
                                                             LevelTable tab, int activation)
{
    const int qpp = C >> 2;
    const long long total = tab.end[tab.levels - 1];
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        int l = 0;
        long long base = 0;
#pragma unroll
        for (int k = 0; k < READ_MAX_LEVELS - 1; ++k)
            if (k < tab.levels - 1 && item >= tab.end[k]) { l = k + 1; base = tab.end[k]; }
        const long long local = item - base;
        const long long pix = local / qpp;
        const int q = (int)(local - pix * qpp);
        long long id = tab.idx[l][pix];
        id = id < 0 ? 0 : (id >= n ? n - 1 : id);   // defensive clamp; ids come from the rasteriser
        float4 v = *reinterpret_cast<const float4 *>(rows + id * C + 4 * q);
        if (activation) {
            v.x = act_apply(v.x, activation);
            v.y = act_apply(v.y, activation);
            v.z = act_apply(v.z, activation);
            v.w = act_apply(v.w, activation);
        }
        *reinterpret_cast<float4 *>(tab.feat[l] + pix * C + 4 * q) = v;
    }
}

// Supersampled lookup: index maps at ss x the output size; out = bilinear downscale (align_corners = False) of the activated
// samples — torch's F.interpolate(scale_factor = 1 / ss) as READ/models/compose.py:162-163 applies it to the texture samples.
struct LevelTableSS {
    const int32_t *idx[READ_MAX_LEVELS];
    float *feat[READ_MAX_LEVELS];
    long long end[READ_MAX_LEVELS];   // exclusive prefix end in (pixel, quad) items over all B images of a level
    int h[READ_MAX_LEVELS], w[READ_MAX_LEVELS];
    int levels;
};

__device__ __forceinline__ float4 act4(float4 v, int activation)
{
    if (activation) {
        v.x = act_apply(v.x, activation);
        v.y = act_apply(v.y, activation);
        v.z = act_apply(v.z, activation);
        v.w = act_apply(v.w, activation);
    }
    return v;
}

__global__ __launch_bounds__(256) void gather_forward_ss_kernel(const float *__restrict__ rows, long long n, int C,
                                                                LevelTableSS tab, int ss, int activation)
{
    const int qpp = C >> 2;
    const long long total = tab.end[tab.levels - 1];
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        int l = 0;
        long long base = 0;
#pragma unroll
        for (int k = 0; k < READ_MAX_LEVELS - 1; ++k)
            if (k < tab.levels - 1 && item >= tab.end[k]) { l = k + 1; base = tab.end[k]; }
        const long long local = item - base;
        const long long pix = local / qpp;
        const int q = (int)(local - pix * qpp);
        const int h = tab.h[l], w = tab.w[l], sh = h * ss, sw = w * ss;
        const int ox = (int)(pix % w), oy = (int)((pix / w) % h);
        const long long b = pix / ((long long)w * h);
        // source coordinate (o + 0.5) * ss - 0.5 >= 0 for ss >= 1
        const float sy = ((float)oy + 0.5f) * (float)ss - 0.5f, sx = ((float)ox + 0.5f) * (float)ss - 0.5f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const int32_t *im = tab.idx[l] + b * (long long)sh * sw;
        float4 v[4];
        const int yy[4] = {y0, y0, y1, y1}, xx[4] = {x0, x1, x0, x1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            long long id = im[(long long)yy[k] * sw + xx[k]];
            id = id < 0 ? 0 : (id >= n ? n - 1 : id);
            v[k] = act4(*reinterpret_cast<const float4 *>(rows + id * C + 4 * q), activation);
        }
        const float wy0 = 1.0f - ly, wx0 = 1.0f - lx;
        float4 o;
        o.x = wy0 * (wx0 * v[0].x + lx * v[1].x) + ly * (wx0 * v[2].x + lx * v[3].x);
        o.y = wy0 * (wx0 * v[0].y + lx * v[1].y) + ly * (wx0 * v[2].y + lx * v[3].y);
        o.z = wy0 * (wx0 * v[0].z + lx * v[1].z) + ly * (wx0 * v[2].z + lx * v[3].z);
        o.w = wy0 * (wx0 * v[0].w + lx * v[1].w) + ly * (wx0 * v[2].w + lx * v[3].w);
        *reinterpret_cast<float4 *>(tab.feat[l] + pix * C + 4 * q) = o;
    }
}

// Bilinear down-scale by an integer factor of NCHW planes: torch's F.interpolate(scale_factor = 1 / ss, mode = 'bilinear',
// align_corners = False) as READ/models/compose.py:162-163 applies it to the concatenated (non-uv extras + texture sample)
// network inputs.  out[p][oy][ox] = blend of the 4 samples around ((o + 0.5) * ss - 0.5); lane = output pixel of a plane.
__global__ __launch_bounds__(256) void bilinear_down_kernel(const float *__restrict__ in, long long planes, int h, int w, int ss,
                                                            float *__restrict__ out)
{
    const long long total = planes * h * w;
    const int sh = h * ss, sw = w * ss;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % w), oy = (int)((i / w) % h);
        const long long p = i / ((long long)w * h);
        const float sy = ((float)oy + 0.5f) * (float)ss - 0.5f, sx = ((float)ox + 0.5f) * (float)ss - 0.5f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, wy0 = 1.0f - ly, wx0 = 1.0f - lx;
        const float *im = in + p * (long long)sh * sw;
        out[i] = wy0 * (wx0 * im[(long long)y0 * sw + x0] + lx * im[(long long)y0 * sw + x1]) +
                 ly * (wx0 * im[(long long)y1 * sw + x0] + lx * im[(long long)y1 * sw + x1]);
    }
}

// Its adjoint.  For ss >= 2 the 2x2 footprints of different outputs are disjoint, so the gradient of an input sample comes
// from at most one output: a gather, no atomics.  lane = input sample.
__global__ __launch_bounds__(256) void bilinear_down_backward_kernel(const float *__restrict__ dout, long long planes, int h, int w,
                                                                     int ss, float *__restrict__ din)
{
    const int sh = h * ss, sw = w * ss;
    const long long total = planes * sh * sw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % sw), y = (int)((i / sw) % sh);
        const long long p = i / ((long long)sw * sh);
        const int oy = y / ss, ox = x / ss;
        const float sy = ((float)oy + 0.5f) * (float)ss - 0.5f, sx = ((float)ox + 0.5f) * (float)ss - 0.5f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        float wy = 0.0f, wx = 0.0f;
        if (y == y0) wy += 1.0f - ly;
        if (y == y1) wy += ly;
        if (x == x0) wx += 1.0f - lx;
        if (x == x1) wx += lx;
        din[i] = wy * wx * dout[(p * h + oy) * w + ox];
    }
}

struct LevelTableBwd {
    const int32_t *idx[READ_MAX_LEVELS];
    const float *dfeat[READ_MAX_LEVELS];
    long long end[READ_MAX_LEVELS];
    int levels;
};

__global__ __launch_bounds__(256) void gather_backward_kernel(float *__restrict__ drows, long long n, int C,
                                                              LevelTableBwd tab)
{
    const int qpp = C >> 2;
    const long long total = tab.end[tab.levels - 1];
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < total;
         item += (long long)gridDim.x * blockDim.x) {
        int l = 0;
        long long base = 0;
#pragma unroll
        for (int k = 0; k < READ_MAX_LEVELS - 1; ++k)
            if (k < tab.levels - 1 && item >= tab.end[k]) { l = k + 1; base = tab.end[k]; }
        const long long local = item - base;
        const long long pix = local / qpp;
        const int q = (int)(local - pix * qpp);
        long long id = tab.idx[l][pix];
        id = id < 0 ? 0 : (id >= n ? n - 1 : id);
        const float4 g = *reinterpret_cast<const float4 *>(tab.dfeat[l] + pix * C + 4 * q);
        float *dst = drows + id * C + 4 * q;
        atomicAdd(dst + 0, g.x);
        atomicAdd(dst + 1, g.y);
        atomicAdd(dst + 2, g.z);
        atomicAdd(dst + 3, g.w);
    }
}

// (C, N) -> N x C: lane = point, C strided-but-coalesced loads, one contiguous row store.
__global__ __launch_bounds__(256) void texture_to_rows_kernel(const float *__restrict__ tex, long long n, int C,
                                                              float *__restrict__ rows)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < C; c += 4) {
        float4 v;
        v.x = tex[(long long)(c + 0) * n + i];
        v.y = tex[(long long)(c + 1) * n + i];
        v.z = tex[(long long)(c + 2) * n + i];
        v.w = tex[(long long)(c + 3) * n + i];
        *reinterpret_cast<float4 *>(rows + i * C + c) = v;
    }
}

__global__ __launch_bounds__(256) void rows_to_texture_kernel(const float *__restrict__ rows, long long n, int C,
                                                              float *__restrict__ tex)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(rows + i * C + c);
        tex[(long long)(c + 0) * n + i] = v.x;
        tex[(long long)(c + 1) * n + i] = v.y;
        tex[(long long)(c + 2) * n + i] = v.z;
        tex[(long long)(c + 3) * n + i] = v.w;
    }
}

int check_levels(const char *who, int64_t n, int C, int levels, const void *idx, const int64_t *count, const void *ptrs)
{
    READ_CHECK_ARG(n >= 1, "%s: empty descriptor table", who);
    READ_CHECK_ARG(C >= 4 && C % 4 == 0 && C <= 64, "%s: C must be a multiple of 4 in [4,64] (got %d)", who, C);
    READ_CHECK_ARG(levels >= 1 && levels <= READ_MAX_LEVELS, "%s: levels must be 1..%d", who, READ_MAX_LEVELS);
    READ_CHECK_ARG(idx && count && ptrs, "%s: null level table", who);
    return READ_OK;
}

}  // namespace

extern "C" int read_texture_to_rows(const float *tex_cn, int64_t n, int C, float *rows_nc, void *stream)
{
    READ_CHECK_ARG(tex_cn && rows_nc && n >= 1, "read_texture_to_rows: null pointer or empty table");
    READ_CHECK_ARG(C >= 4 && C % 4 == 0, "read_texture_to_rows: C must be a multiple of 4");
    READ_CHECK_ARG((uintptr_t)rows_nc % 16 == 0, "read_texture_to_rows: rows must be 16-byte aligned");
    hipLaunchKernelGGL(texture_to_rows_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream),
                       tex_cn, (long long)n, C, rows_nc);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_rows_to_texture(const float *rows_nc, int64_t n, int C, float *tex_cn, void *stream)
{
    READ_CHECK_ARG(tex_cn && rows_nc && n >= 1, "read_rows_to_texture: null pointer or empty table");
    READ_CHECK_ARG(C >= 4 && C % 4 == 0, "read_rows_to_texture: C must be a multiple of 4");
    READ_CHECK_ARG((uintptr_t)rows_nc % 16 == 0, "read_rows_to_texture: rows must be 16-byte aligned");
    hipLaunchKernelGGL(rows_to_texture_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, as_stream(stream),
                       rows_nc, (long long)n, C, tex_cn);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_gather_forward(const float *rows_nc, int64_t n, int C, int levels,
                                   const int32_t *const *idx_levels, const int64_t *count_levels,
                                   float *const *feat_levels, int activation, void *stream)
{
    int rc = check_levels("read_gather_forward", n, C, levels, idx_levels, count_levels, feat_levels);
    if (rc) return rc;
    READ_CHECK_ARG(rows_nc && (uintptr_t)rows_nc % 16 == 0, "read_gather_forward: rows null or misaligned");
    READ_CHECK_ARG(activation >= 0 && activation <= 2, "read_gather_forward: activation must be 0,1,2");
    LevelTable tab;
    memset(&tab, 0, sizeof(tab));
    long long acc = 0;
    const int qpp = C / 4;
    for (int l = 0; l < levels; ++l) {
        READ_CHECK_ARG(count_levels[l] >= 0, "read_gather_forward: negative pixel count");
        READ_CHECK_ARG(count_levels[l] == 0 || (idx_levels[l] && feat_levels[l]), "read_gather_forward: null level %d", l);
        READ_CHECK_ARG((uintptr_t)feat_levels[l] % 16 == 0, "read_gather_forward: feat level %d misaligned", l);
        tab.idx[l] = idx_levels[l];
        tab.feat[l] = feat_levels[l];
        acc += count_levels[l] * qpp;
        tab.end[l] = acc;
    }
    tab.levels = levels;
    if (acc == 0) return READ_OK;
    int64_t blocks = ceil_div64(acc, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(gather_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), rows_nc,
                       (long long)n, C, tab, activation);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_gather_forward_ss(const float *rows_nc, int64_t n, int C, int levels, int B,
                                      const int32_t *const *idx_levels, const int *h_levels, const int *w_levels, int ss,
                                      float *const *feat_levels, int activation, void *stream)
{
    int rc = check_levels("read_gather_forward_ss", n, C, levels, idx_levels, (const int64_t *)h_levels, feat_levels);
    if (rc) return rc;
    READ_CHECK_ARG(rows_nc && (uintptr_t)rows_nc % 16 == 0, "read_gather_forward_ss: rows null or misaligned");
    READ_CHECK_ARG(w_levels && B >= 1 && ss >= 1 && ss <= 8, "read_gather_forward_ss: bad B / ss / sizes");
    READ_CHECK_ARG(activation >= 0 && activation <= 2, "read_gather_forward_ss: activation must be 0,1,2");
    LevelTableSS tab;
    memset(&tab, 0, sizeof(tab));
    long long acc = 0;
    const int qpp = C / 4;
    for (int l = 0; l < levels; ++l) {
        READ_CHECK_ARG(h_levels[l] >= 1 && w_levels[l] >= 1 && idx_levels[l] && feat_levels[l],
                       "read_gather_forward_ss: bad level %d", l);
        READ_CHECK_ARG((uintptr_t)feat_levels[l] % 16 == 0, "read_gather_forward_ss: feat level %d misaligned", l);
        tab.idx[l] = idx_levels[l];
        tab.feat[l] = feat_levels[l];
        tab.h[l] = h_levels[l];
        tab.w[l] = w_levels[l];
        acc += (long long)B * h_levels[l] * w_levels[l] * qpp;
        tab.end[l] = acc;
    }
    tab.levels = levels;
    int64_t blocks = ceil_div64(acc, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(gather_forward_ss_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), rows_nc,
                       (long long)n, C, tab, ss, activation);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_gather_backward(float *drows_nc, int64_t n, int C, int levels,
                                    const int32_t *const *idx_levels, const int64_t *count_levels,
                                    const float *const *dfeat_levels, void *stream)
{
    int rc = check_levels("read_gather_backward", n, C, levels, idx_levels, count_levels, dfeat_levels);
    if (rc) return rc;
    READ_CHECK_ARG(drows_nc, "read_gather_backward: drows is null");
    LevelTableBwd tab;
    memset(&tab, 0, sizeof(tab));
    long long acc = 0;
    const int qpp = C / 4;
    for (int l = 0; l < levels; ++l) {
        READ_CHECK_ARG(count_levels[l] >= 0, "read_gather_backward: negative pixel count");
        READ_CHECK_ARG(count_levels[l] == 0 || (idx_levels[l] && dfeat_levels[l]), "read_gather_backward: null level %d", l);
        READ_CHECK_ARG((uintptr_t)dfeat_levels[l] % 16 == 0, "read_gather_backward: dfeat level %d misaligned", l);
        tab.idx[l] = idx_levels[l];
        tab.dfeat[l] = dfeat_levels[l];
        acc += count_levels[l] * qpp;
        tab.end[l] = acc;
    }
    tab.levels = levels;
    if (acc == 0) return READ_OK;
    int64_t blocks = ceil_div64(acc, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(gather_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), drows_nc,
                       (long long)n, C, tab);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_bilinear_down(const float *in, int64_t planes, int h, int w, int ss, float *out, void *stream)
{
    READ_CHECK_ARG(in && out && planes >= 1 && h >= 1 && w >= 1, "read_bilinear_down: null pointer or empty tensor");
    READ_CHECK_ARG(ss >= 2 && ss <= 8, "read_bilinear_down: the factor must be 2..8 (got %d)", ss);
    int64_t blocks = ceil_div64(planes * h * w, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(bilinear_down_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in, (long long)planes, h, w,
                       ss, out);
    READ_CHECK_LAUNCH();
    return READ_OK;
}

extern "C" int read_bilinear_down_backward(const float *dout, int64_t planes, int h, int w, int ss, float *din, void *stream)
{
    READ_CHECK_ARG(dout && din && planes >= 1 && h >= 1 && w >= 1, "read_bilinear_down_backward: null pointer or empty tensor");
    READ_CHECK_ARG(ss >= 2 && ss <= 8, "read_bilinear_down_backward: the factor must be 2..8 (got %d)", ss);
    int64_t blocks = ceil_div64(planes * h * w * ss * ss, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(bilinear_down_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dout,
                       (long long)planes, h, w, ss, din);
    READ_CHECK_LAUNCH();
    return READ_OK;
}
