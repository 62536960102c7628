/*
 * read_hip.h — C ABI of libreadhip.so: READ's per-frame render path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  Everything above it (the Python mirror of READ's
 * pcpr / MyRender / PointTexture / UNet / NetAndTexture / TexturePipeline / OGL API in
 * read_amd/) and any other host (C++, cgo, JNI, ctypes) binds exactly these symbols.
 *
 * Conventions
 *   - every function returns 0 (READ_OK) or a negative READ_E* code and never throws;
 *     read_last_error() gives the message of the calling thread's last failure;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     enqueued on it, nothing synchronises, nothing allocates after *_create / workspace setup;
 *   - pointers are DEVICE pointers owned by the caller unless the name ends in `_host`;
 *   - images are row-major, row 0 = image top; activations are NHWC fp32;
 *   - matrices are row-major 4x4 applied as M * [x y z 1]^T (point_render.cu:96-121).
 *
 * Reference interfaces replaced (file:line under the reference repo):
 *   read_splat_forward        MyRender/CloudProjection/pcpr_cuda.cpp:23-42 (pcpr.forward),
 *                             point_render.cu:125-200 (DepthProject + GPU_PCPR), called 5x per
 *                             iteration by src/READ/gl/myrender.py:32-40; GL twin:
 *                             READ/gl/render.py:52-85 + READ/gl/programs.py:121-125,164-167
 *   read_index_to_float       point_render.cu:158 (`out_index[ind] = (float)ids`)
 *   read_gather_forward       READ/models/texture.py:42-70 (PointTexture.forward = index_select)
 *   read_gather_backward      autograd of texture.py:61 (index_add into texture_.grad)
 *   read_texture_to_rows      layout change of PointTexture.texture_ (1,C,N) -> N x C rows
 *   read_gated_conv_forward   READ/models/unet.py:22-53 (BasicConv) incl. the ResBlock add (:19-20),
 *                             FAM multiply/add (:115-117), torch.cat (:88,:104,:262,:270,:278) and
 *                             F.interpolate nearest (:239-250) folded into its prologue/epilogue
 *   read_bilinear_up4         READ/models/unet.py:200 (nn.Upsample(scale_factor=4, 'bilinear'))
 *   read_unet_*               READ/models/unet.py:121-285 (UNet.__init__/forward)
 */
#ifndef READ_HIP_H
#define READ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define READ_OK        0
#define READ_EINVAL  (-22)   /* bad argument (shape, alignment, null pointer) */
#define READ_ENOMEM  (-12)   /* workspace too small */
#define READ_EHIP     (-5)   /* a HIP runtime call failed; see read_last_error() */

#define READ_MAX_LEVELS      5
#define READ_DESC_CHANNELS   8   /* descriptor size the UNet consumes (READ/pipelines/ogl.py:19-25) */

const char *read_last_error(void);
int read_abi_version(void);
/* Fills name[0..len) with the gfx arch of the current device ("gfx950"); READ_EHIP without a GPU. */
int read_device_arch(char *name, int len);

/* Measurement knobs (A/B runs on the GPU box).  Every knob of the release library selects between implementations
 * that produce the SAME results (the attribution probes with invalid results exist only in -DREAD_DEBUG_KNOBS builds):
 *   "splat_mode"      plain path: 7 (default) warm start + LDS hierarchical-Z, 1 = agent-scope atomics + early-z only
 *   "splat_cells"     0: ignore the cell-ordered copy (plain path everywhere)
 *   "splat_seeds"     0: no warm start from the previous frame
 *   "splat_near"      striped path: expected points per pixel in front of the pass-A split distance (default 12)
 *   "splat_cells_sub" cell path: every n-th chunk joins pass A (default 0 = only on a workspace's first frame, every 32nd; rounds 2-4: 32)
 *   "splat_items"     striped path: work items per 1024-point chunk (1, 2, 4)
 *   "splat_subset"    plain path: bootstrap pass over every n-th chunk (default 8)
 *   "splat_stats"     debug counters in the workspace header
 *   "conv_wino"       largest Cin that takes the Winograd F(2x2,3x3) kernel (0 = direct implicit-GEMM kernels everywhere)
 *   "conv_sc"         8 (default): gated 3x3/s1 layers with Cin = 32, Cout <= 4 on the vector-pipe kernel, 8 input channels per LDS
 *                     phase (16, 32: larger phases); 0: on the MFMA kernels
 *   "splat_mark"      1 (default): a chunk one of whose points reaches a depth bound is listed in pass A for the next "splat_sticky"
 *                     (default 1) classifications, wherever it lies — on surface-like scenes the chunks beyond the near split that hold
 *                     front points then run banded and binned in pass A instead of surviving pass B's bound test every frame
 *                     (street scene 81.7 -> 70.9 us per frame; volumetric slab unchanged); 0: only pass B's survivors are promoted
 *   "splat_compact"   1 (default): pass A compacts the candidates of a 256-point round into dense lanes before binning them; 0: rounds 2-4
 *   "splat_cells_batch" 1 (default): a batch of cameras runs as B cell-path frames; 0: the plain pass over the whole cloud
 *   "splat_prof"      1: HIP events around every launch of a cell-path frame (read_splat_profile_last); 0 (default)
 *   "splat_ahead"     1 (default): with an announced next camera (read_splat_hint_next_camera) a cell-path frame's resolve launch
 *                     also classifies / seeds the next frame — 4 dependent launches per frame instead of 5; 0: always 5
 *   "unet_up_fold"    0 (default): nn.Upsample(x4, bilinear) as a pass of its own; 1: folded into the Convs.k launches
 *                     (read_conv_desc.pre_bilinear; measured level-to-slower, profiles/r5_up_fold_ab.md); plans created afterwards
 *   "wgrad_wino"      1 (default): 3x3/s1 weight gradients in the Winograd F(4x4,3x3) domain; 0: direct MFMA kernel; v > 1: the same
 *                     with 128 v workgroups aimed at (default 256)
 *   "conv_wave", "conv_kc32", "conv_stagger", "unet_streams": see csrc/conv.hip, csrc/unet.cpp.
 * read_tuning_key(i) enumerates the keys (NULL past the end); read_tuning_get reads the current value, so that a
 * benchmark can record the state it ran with. */
int read_tuning_set(const char *key, int value);
int read_tuning_get(const char *key, int *value);
const char *read_tuning_key(int i);
/* Debug entry points (timeline trace of the conv kernels, issue / operand / MFMA-ceiling probes) are NOT part of this ABI: they
 * exist only in libreadhip_debug.so (python -m read_amd.build --debug, -DREAD_DEBUG_KNOBS) and are declared in read_hip_debug.h. */

/* ---------------------------------------------------------------- rasteriser (z-buffer splat) */

/* Bytes of the persistent rasteriser state for ONE (B, W, H): an 8192-byte header (frame state, two sets of chunk-list counters),
 * the key images (min(B,8) cameras x W*H x 8 B, depth_bits<<32 | point_id), the hierarchical-Z bound image, two seed images (the
 * warm start of the next frame), the two depth-bound images of the cell path (frames alternate between them) and its bins.  A workspace serves the (B, W, H) it
 * was sized for; re-run read_splat_workspace_init before using it with another size.  256-byte aligned. */
size_t read_splat_workspace_bytes(int B, int W, int H);
/* Must be called once on a fresh workspace (sets every key to EMPTY).  read_splat_forward
 * leaves the workspace EMPTY again, so consecutive frames need no further clears. */
int read_splat_workspace_init(void *ws, size_t ws_bytes, void *stream);

/*
 * One pass over the cloud for B cameras and `levels` scales.
 *   xyz          N x 3 fp32 (device)
 *   M_host       B x 16 fp32 on the HOST: proj @ inv(view) per camera (myrender.py:28-30)
 *   W,H          level-0 size; level l is int(W*0.5^l) x int(H*0.5^l) (myrender.py:33-34)
 *   idx_levels   host array of `levels` device pointers, level l = int32 [B][h_l][w_l]
 *   depth_levels same shape, fp32; either array (or single entries) may be NULL to skip
 * Semantics (bit-exact, deterministic): per pixel the accepted point of minimum fp32 depth,
 * ties -> minimum point id; empty pixels (0, 0.0f).  When W or H is not a multiple of
 * 2^(levels-1) each level is rasterised by its own pass over the points.
 */
int read_splat_forward(const float *xyz, int64_t n, const float *M_host, int B, int W, int H,
                       int levels, int32_t *const *idx_levels, float *const *depth_levels,
                       void *ws, size_t ws_bytes, void *stream);

/* Cell-ordered copy of a cloud (optional accelerator of the single-camera path; results are bit-identical).
 * read_splat_cells_build_host() sorts the points once along a Morton curve into chunks of 1024 records
 * (x, y, z, original id) with their bounding boxes (host arrays in, host blob of read_splat_cells_bytes(n) out,
 * multi-threaded); the caller uploads the blob (256-byte aligned) and passes it to read_splat_forward_cells().
 * Whole chunks outside the frustum, or behind the far depth bound of every 4x4 pixel block they can touch, are then
 * skipped without reading their points, and the z-test runs XCD-striped against an L2-resident bound image
 * (csrc/splat.hip).  The tail of the blob is per-frame scratch (chunk lists), so one blob serves one stream at a
 * time.  With cells == NULL, B > 1, n < 2^20, W % 16 != 0 or sizes that are not multiples of 2^(levels-1) the call is
 * exactly read_splat_forward(). */
/* read_splat_hint_next_camera(workspace, M_next_host): optional, before a read_splat_forward_cells call — "the NEXT call on this
 * workspace will use camera M_next" (16 floats on the host; NULL withdraws the hint).  A sweep, a trajectory replay or a viewer
 * that extrapolates its camera knows that matrix one frame ahead.  The hinted frame's last launch then also does the next frame's
 * first one (chunk classification for M_next, seeds re-projected with M_next into the other bound image): a frame is 4 dependent
 * launches instead of 5.  Purely a performance device with identical results: if the next call comes with another matrix, size or
 * tuning state, the prepared state is wiped (two small memsets) and the frame runs as without a hint.  The hint is consumed by
 * the next forward call on the workspace; the frame counter behind it lives on the host, keyed by the workspace address. */
int read_splat_hint_next_camera(void *workspace, const float *M_next_host);
/* Per-kernel durations (ms) of the LAST cell-path frame, HIP events on the launch stream around every launch; needs
 * read_tuning_set("splat_prof", 1) before the frame.  ms5[0] seeds + classification (0 when the previous frame's resolve launch did
 * that work), [1] pass A, [2] bin merge + bounds, [3] pass B, [4] resolve (+ the next frame's seeds / classification).  Synchronises. */
int read_splat_profile_last(float *ms5);
size_t read_splat_cells_bytes(int64_t n);
int read_splat_cells_build_host(const float *xyz_host, int64_t n, void *cells_host, size_t cells_bytes);
/* The library keeps host-side bookkeeping per cell blob ADDRESS (which frames ran over its chunk lists; a frame prepared for an
 * announced camera is only consumed while nothing else touched them).  Whoever rewrites a blob in place, copies another cloud's
 * blob over it, or frees it (the allocator may hand the address out again) calls this: pending preparations made against the old
 * contents then no longer match and are wiped instead of consumed, and the entry is forgotten.  Cheap; no device work. */
int read_splat_cells_invalidate(const void *cells, int64_t n);
int read_splat_forward_cells(const float *xyz, void *cells, int64_t n, const float *M_host, int B, int W, int H,
                             int levels, int32_t *const *idx_levels, float *const *depth_levels,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Measurement aid for bench.py (roofline.mfma_sustained): one workgroup of four waves per CU, every wave `iters` rounds of 16 independent
 * v_mfma_f32_16x16x4_f32 and nothing else.  scratch: >= 256 floats per CU on the device (never written); *flops receives the number of
 * floating-point operations the launch executes — the caller times the launch on `stream`.  Not on the render path. */
int read_mfma_f32_rate_probe(int iters, float *scratch, double *flops, void *stream);

/* GL twin features of the rasteriser (READ/gl/programs.py:121-198, READ/gl/render.py:52-85, READ/gl/dataset.py:39-82,
 * READ/datasets/dynamic.py:235-239): ONE level of ONE camera rasterised at its own size W x H with
 *   point_size / relative / min_point_size   "pN" tokens: a square of N pixels; "psN" tokens (relative = 1): side
 *                                             max(min_point_size, N / clip.z); never below one pixel
 *   discard                                   optional device array of N bytes, 1 = point not drawn (set_point_discard)
 *   drop_threshold, drop_seed                 seeded drop: point i is dropped iff rnd(i, seed) < threshold (0 = off);
 *                                             threshold = p * (2^32 - 1)
 *   perturb                                   optional device array N x 2 added to clip.xy (set_point_perturb)
 *   perturb_amp, perturb_seed                 seeded perturbation amp * (u01(i, seed) - 0.5) per axis (0 = off)
 * Coverage rule = OpenGL's basic point rasterisation with GL_PROGRAM_POINT_SIZE (READ/gl/render.py:55; GL 4.6 core spec
 * section 14.4.1: "a fragment for each framebuffer pixel whose center lies inside a square centered at the point's (xw, yw), with
 * side length equal to the current point size"): pixel column i is covered iff xw - s/2 <= i + 0.5 < xw + s/2, i.e.
 * i in [floor(u - (s-1)/2), floor(u + (s-1)/2)] away from exact ties — for odd AND even s; points are clipped by their CENTRE
 * (spec 13.7: a point outside the clip volume is discarded whole), sizes clamp to [1, 4096] (the aliased point size range).
 * Exact semantics: csrc/splat.hip (splat_project_gl_kernel) == oracle/raster.c (oracle_raster_level_gl).  With the default
 * options {1, 0, 1, NULL, 0, 0, NULL, 0, 0, NULL} the result equals read_splat_forward(levels = 1).  The workspace is the one of
 * read_splat_workspace_bytes(1, W, H); it is left EMPTY. */
typedef struct read_splat_gl_opts {
    float point_size;
    int relative;
    float min_point_size;
    const unsigned char *discard;
    uint32_t drop_threshold, drop_seed;
    const float *perturb;
    float perturb_amp;
    uint32_t perturb_seed;
    const float *point_sizes;    /* optional device array of N per-point sizes (NNScene.set_point_sizes, the a_point_size vertex
                                  * attribute): used instead of point_size when point_size == 0 (the shader's
                                  * "if (point_size < 1) point_size = a_point_size", READ/gl/programs.py:183-187, :339-345) */
} read_splat_gl_opts;
int read_splat_forward_gl(const float *xyz, int64_t n, const float *M_host, int W, int H,
                          const read_splat_gl_opts *opts, int32_t *idx, float *depth, void *workspace,
                          size_t workspace_bytes, void *stream);

/* out[i] = (float)idx[i] — the reference's index image dtype (ids >= 2^24 round). */
int read_index_to_float(const int32_t *idx, int64_t count, float *out, void *stream);
/* Projection alone (point_render.cu:110-147 without the z-test of :149-166): pixel[i] = yy * W + xx of point i under the
 * row-major 4x4 matrix M_host, or -1 when the point is clipped or falls outside the image; depth[i] (may be NULL) = (nz + 1) / 2
 * as the rasteriser computes it, also for rejected points.  The same device function as every rasteriser pass — per-point
 * visibility for a host, and the entry the division test drives (tests/test_gpu_splat.py). */
int read_splat_project_points(const float *xyz, int64_t n, const float *M_host, int W, int H, int32_t *pixel, float *depth,
                              void *stream);

/* ---------------------------------------------------------------- descriptor gather / scatter */

/* (C, N) channel-major texture  ->  N x C row-major rows (and back, for gradients). */
int read_texture_to_rows(const float *tex_cn, int64_t n, int C, float *rows_nc, void *stream);
int read_rows_to_texture(const float *rows_nc, int64_t n, int C, float *tex_cn, void *stream);

/*
 * feat_l[p][c] = rows[idx_l[p]][c]  for every level l and pixel p (NHWC, C % 4 == 0).
 *   count_levels[l] = number of pixels of level l (B*h_l*w_l); activation: 0 none, 1 sigmoid, 2 tanh
 */
int read_gather_forward(const float *rows_nc, int64_t n, int C, int levels,
                        const int32_t *const *idx_levels, const int64_t *count_levels,
                        float *const *feat_levels, int activation, void *stream);
/* Supersampled lookup (READ/gl/nn.py:100-101 + READ/models/compose.py:162-163): index maps rendered at ss x the feature
 * size; feat_l = bilinear-downscale-by-ss (align_corners = False, torch's F.interpolate(scale_factor = 1/ss)) of the
 * activated samples, fused: each output pixel blends the four samples around source coordinate (o + 0.5) * ss - 0.5.
 *   h_levels[l], w_levels[l] = OUTPUT size of level l; idx_l is [B][ss*h][ss*w]; feat_l is [B][h][w][C]. */
int read_gather_forward_ss(const float *rows_nc, int64_t n, int C, int levels, int B,
                           const int32_t *const *idx_levels, const int *h_levels, const int *w_levels, int ss,
                           float *const *feat_levels, int activation, void *stream);
/* drows[idx_l[p]][c] += dfeat_l[p][c]   (fp32 atomics; drows must be zeroed by the caller). */
int read_gather_backward(float *drows_nc, int64_t n, int C, int levels,
                         const int32_t *const *idx_levels, const int64_t *count_levels,
                         const float *const *dfeat_levels, void *stream);

/* Bilinear down-scale by an integer factor ss (2..8) of `planes` NCHW planes [ss*h][ss*w] -> [h][w]: torch's
 * F.interpolate(scale_factor = 1/ss, mode = 'bilinear', align_corners = False) as READ/models/compose.py:162-163 applies it to
 * network inputs that mix non-uv tokens with texture samples (the all-uv case is fused into read_gather_forward_ss), and its
 * adjoint (din is written, not accumulated). */
int read_bilinear_down(const float *in, int64_t planes, int h, int w, int ss, float *out, void *stream);
int read_bilinear_down_backward(const float *dout, int64_t planes, int h, int w, int ss, float *din, void *stream);

/* ---------------------------------------------------------------- gated conv (BasicConv) */

#define READ_CONV_MAX_SRC 4

typedef struct read_conv_src {
    const float *data;   /* NHWC fp32 [srcH][srcW][C] */
    int C;               /* channels taken from this source (all of them) */
    int srcH, srcW;
    int shift;           /* nearest resample folded into addressing: >0: source is 2^shift finer
                            (src = dst << shift), <0: coarser (src = dst >> -shift), 0: same size */
} read_conv_src;

typedef struct read_conv_desc {
    int n_src;                              /* inputs concatenated along C (torch.cat order) */
    read_conv_src src[READ_CONV_MAX_SRC];
    const float *mul;                       /* optional: input := src[0] * mul (FAM), same shape */
    int inH, inW;                           /* logical input size after resampling */
    int Cout, ksize, stride;                /* ksize in {1,3,4}, stride in {1,2}; pad = (ksize-1)/2 */
    int elu;                                /* 1: ELU on the feature branch (relu=True) */
    const float *wpacked;                   /* read_conv_pack_weights() output (device) */
    const float *params;                    /* 4 x CoutPad: bias_f, bias_m, bn_scale, bn_shift */
    const float *residual;                  /* optional NHWC [outH][outW][Cout], added after BN */
    float *out;                             /* NHWC [outH][outW][out_cstride], first Cout written */
    int out_cstride;                        /* >= Cout */
    float out_fill;                         /* value for channels Cout..out_cstride-1 when fill_pad */
    int fill_pad;
    int config;                             /* tile configuration id, -1 = pick automatically */
    const float *wpacked_wino;              /* optional: read_conv_pack_wino_host() output (device); enables the
                                               Winograd F(2x2,3x3) kernel for 3x3/s1 single-source layers */
    const float *wpacked_w16;               /* optional: read_conv_pack_w16_host() output (device): the same Winograd operand in the
                                               order of the wave-autonomous F(2x2) kernel (all 16 frequencies of a tile in one wave,
                                               v_mfma_f32_16x16x4_f32); taken by non-linear launches with Cout <= 8 (the 32 -> 3
                                               output layer), with read_tuning_set("conv_w16", 1), or with config = -3 */
    const float *wpacked_w4;                /* optional: read_conv_pack_w4_host() output (device): Winograd F(4x4,3x3) operand; taken by
                                               3x3 / stride-1 single-source launches with Cin % 16 == 0, Cin >= read_tuning("conv_w4")
                                               (default 32) and Cout % 32 == 0 (linear launches: Cout % 8 == 0); config = -5
                                               forces it where the shape fits */
    int linear;                             /* 1: plain convolution (training path): out[..][c] = conv_f + b_f,
                                               out[..][Cout + c] = conv_m + b_m, out_cstride >= 2 * Cout; no gate /
                                               BatchNorm / residual; workgroup-tiled or Winograd kernel */
    /* optional pre-activation addend, NHWC [preH][preW][pre_cstride]: before bias / activation / gate,
     *   conv_f(y,x,c) += pre[y >> pre_shift][x >> pre_shift][pre_f_off + c],  conv_m likewise with pre_m_off.
     * A 1x1 convolution commutes with nearest up-sampling, so the share of torch.cat([.., F.interpolate(coarse)]) -> 1x1 conv
     * that comes from a coarser tensor is computed at the coarse resolution by a `linear` launch and enters here
     * (READ/models/unet.py:239-254, the AFF inputs).  Not available with the Winograd kernel. */
    const float *pre;
    int pre_cstride, pre_f_off, pre_m_off, pre_shift, preH, preW;
    /* optional, linear launches that run on the Winograd kernel (3x3 / stride 1, Cin % 16 == 0): ALSO store the layer's output
     * BN_eval(act(f) * sigmoid(m)), NHWC [outH][outW][Cout], in the same pass (what read_gate_forward would compute from
     * `out`); rows r of a stacked batch with r % block_h >= valid_h are written as zeros (block_h = 0: one image). */
    float *out_gated;
    int block_h, valid_h;
    const float *wpacked_sc;                /* optional: read_conv_pack_sc_host() output (device, 64-byte aligned): gated 3x3 / stride-1 launches
                                               with Cin = 32, Cout <= 4 and one unshifted source (READ's output layer 32 -> 3) run on the
                                               vector pipe — thread = pixel, weights streamed through SGPRs — instead of padding six output
                                               rows to an MFMA shape; read_tuning_set("conv_sc", 0) switches it off, config = -6 forces it */
    int pre_bilinear;                       /* 1: the pre-activation addend is sampled BILINEARLY at 4x (pre_shift must be 2): the value
                                               nn.Upsample(scale_factor=4, mode='bilinear', align_corners=False) of `pre` has at (y, x)
                                               (source index 0.25 (dst + 0.5) - 0.5 clamped at 0, neighbours clamped at the edge).  A
                                               1x1 convolution commutes with bilinear up-sampling as it does with nearest, so the share
                                               of cat[Upsample4(coarse), fine] -> 1x1 conv that comes from the coarse tensor is computed at
                                               1/16 of the pixels by a `linear` launch and enters here: no up-sampled tensor is ever
                                               written (READ/models/unet.py:261-262,269-270,277-278, the Convs.k inputs).  1x1 / stride-1
                                               layers on the pixel-lane kernel only (16-byte aligned tensors, Cin <= 256, Cout % 4 == 0) */
    const void *wpacked_w4h;                /* optional: read_conv_pack_w4h_host() output (device): the Winograd F(4x4,3x3) operand split into
                                               two f16 pieces per weight + one power-of-two scale per output row.  Gated (non-linear)
                                               launches of the F(4x4) family with Cin % 32 == 0, Cin >= read_tuning("conv_w4h") (default
                                               32, 0 = never) and Cout % 32 == 0 then run on the f16 matrix cores (v_mfma_f32_16x16x32_f16,
                                               fp32 accumulation, three piece pairs per product: fp32-level results — DESIGN.md 3.3 (a+),
                                               profiles/r6_f16split_probe.txt; transformed inputs must stay below 65504 in magnitude,
                                               i.e. activations below ~650); config = -7 forces it where the shape fits */
    const void *wpacked_d3h;                /* optional: read_conv_pack_d3h_host() output (device): the 3x3 weights themselves as two f16 pieces
                                               per weight + a power-of-two scale per output row.  Gated 3x3 / stride-1 single-source launches
                                               with Cin % 32 == 0, Cin >= read_tuning("conv_d3h") (default 32, 0 = never) and Cout % 32 == 0
                                               (FAM's mul included) then run as a DIRECT convolution on the f16 matrix cores — all nine
                                               taps, three piece pairs per product, fp32 accumulation: no Winograd transform on either
                                               side, fp32-level results (DESIGN.md 3.3 (a++)); inputs must stay below 65504 in magnitude;
                                               config = -8 forces it where the shape fits.  Takes precedence over wpacked_w4h / wpacked_w4.
                                               For a 1x1 / stride-1 layer the same field takes read_conv_pack_dkh_host(Cin, Cout, 1, ...): launches
                                               (gated or linear, concatenated sources, residual, pre-activation addend) with Cin % 16 == 0,
                                               read_tuning("conv_pxh") (default 16, 0 = never) <= Cin <= 256, Cout % 4 == 0 and 16-byte aligned
                                               tensors then run on the split-operand pixel-lane kernel (v_mfma_f32_32x32x16_f16, three piece
                                               pairs per product, fp32 accumulation); config = -10 forces it where the shape fits */
    const void *wpacked_t3h;                /* optional: read_conv_pack_t3h_host() output (device): the 3x3 weights of a layer with 8, 16 or 32 input
                                               channels as an implicit-GEMM operand (k = tap * Cin + channel, zero-padded to whole steps of 16) in
                                               f16 piece pairs.  3x3 / stride-1 launches (gated or linear, residual) over ONE unshifted source with
                                               C <= read_tuning("conv_t3h") (default 8: the layers that read the 8-channel descriptor pyramid; 0 =
                                               never), Cout % 4 == 0 and 16-byte aligned tensors then run on the split-operand pixel-lane kernel
                                               (zero padding at the image border, as nn.Conv2d); config = -11 forces it where the shape fits */
} read_conv_desc;

/* Sizes (in floats) of the packed weight / parameter blocks of one BasicConv. */
size_t read_conv_packed_floats(int Cin, int Cout, int ksize);
size_t read_conv_param_floats(int Cout);
/* Host-side packing from the PyTorch layout (all host pointers):
 *   wf, wm (Cout,Cin,k,k); bf, bm (Cout); BN gamma, beta, running_mean, running_var (Cout), eps.
 *   kc = input-channel chunk the kernel stages per step: 16 when every concatenated source has
 *   C % 16 == 0, else 8 (the packing order depends on it). */
int read_conv_pack_weights_host(int Cin, int Cout, int ksize, int kc, const float *wf,
                                const float *wm, float *wpacked_host);
/* Winograd F(2x2,3x3): transformed weights G g G^T in fragment order, Cin % 16 == 0 (16/9 the plain size). */
size_t read_conv_wino_floats(int Cin, int Cout);
int read_conv_pack_wino_host(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_wino_host);
/* same size (read_conv_wino_floats), order of the wave-autonomous Winograd kernel: [group][wave][chunk][row][col][lane][4] */
int read_conv_pack_w16_host(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_w16_host);
/* Winograd F(4x4,3x3) operand: read_conv_w4_floats(Cin, Cout) = Cin * 36 * 2 * pad32(Cout) floats,
 * [group][wave][chunk of 16 cin][frequency 36][lane][4] */
size_t read_conv_w4_floats(int Cin, int Cout);
int read_conv_pack_w4_host(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_w4_host);
/* Split-operand F(4x4,3x3) operand (desc.wpacked_w4h): read_conv_w4h_floats(Cin, Cout) = Cin * 36 * 2 * pad32(Cout) + 2 * pad32(Cout)
 * floats (0: Cin % 32 != 0) — [group][wave][chunk of 32 cin][frequency 36][piece hi | lo][lane][8 halfs], then 1 / scale per output row */
size_t read_conv_w4h_floats(int Cin, int Cout);
int read_conv_pack_w4h_host(int Cin, int Cout, const float *wf, const float *wm, void *wpacked_w4h_host);
/* Direct split-operand 3x3 operand (desc.wpacked_d3h): read_conv_d3h_floats(Cin, Cout) = Cin * 18 * pad32(Cout) + 2 * pad32(Cout) floats
 * (0: Cin % 32 != 0) — [group][row half][chunk of 32 cin][tap 9][row block 2][piece hi | lo][lane][8 halfs], then 1 / scale per output row */
size_t read_conv_d3h_floats(int Cin, int Cout);
int read_conv_pack_d3h_host(int Cin, int Cout, const float *wf, const float *wm, void *wpacked_d3h_host);
/* ... for a k x k kernel, k = 3 or 4 ([tap k * k] in the order above; Cin * 2 * k * k * pad32(Cout) + 2 * pad32(Cout) floats): with stride 2, Cin % 32 == 0
 * and Cout % 32 == 0 (the encoder's 3x3 / stride-2 and the decoder's 4x4 / stride-2 layers) desc.wpacked_d3h sends the launch to the stride-2 form
 * of the direct split-operand kernel (read_tuning("conv_d3h_s2"), default 32, 0 = never; config = -9 forces it).
 * ksize = 1 (Cin % 16 == 0): the 1x1 layers' operand for the split-operand pixel-lane kernel, Cin * 2 * pad32(Cout) + 2 * pad32(Cout) floats,
 * [k16 step][tile = 2 group + (f | m)][piece hi | lo][lane][8 halfs] (lane = row (lane & 31) of the tile, cin = 16 step + 8 (lane >> 5) + e), then 1 / scale */
size_t read_conv_dkh_floats(int Cin, int Cout, int ksize);
int read_conv_pack_dkh_host(int Cin, int Cout, int ksize, const float *wf, const float *wm, void *wpacked_host);
/* Implicit-GEMM operand of the 3x3 layers with 8, 16 or 32 input channels (desc.wpacked_t3h): the ksize-1 order above for the matrix
 * W'[cout][tap * Cin + ci] padded to K = pad16(9 * Cin) columns: K * 2 * pad32(Cout) + 2 * pad32(Cout) floats (0: another Cin) */
size_t read_conv_t3h_floats(int Cin, int Cout);
int read_conv_pack_t3h_host(int Cin, int Cout, const float *wf, const float *wm, void *wpacked_t3h_host);
/* Small-Cout order [tap][cin][f0 f1 f2 f3 | m0 m1 m2 m3] (9 * Cin * 8 floats; 0 = the shape has no such order: only Cin = 32,
 * Cout <= 4 has a kernel). */
size_t read_conv_sc_floats(int Cin, int Cout);
int read_conv_pack_sc_host(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_sc_host);
int read_conv_pack_params_host(int Cout, const float *bf, const float *bm, const float *gamma,
                               const float *beta, const float *mean, const float *var, float eps,
                               float *params_host);
int read_gated_conv_forward(const read_conv_desc *desc, void *stream);
/* Which kernel family read_gated_conv_forward takes for this (filled) descriptor under the current tuning knobs: 8 = the same kernel as an
 * implicit GEMM over a 3x3 layer with 8 - 32 input channels (reads wpacked_t3h), 7 = 1x1 pixel-lane kernel
 * with split operands on the f16 matrix cores (reads wpacked_d3h of a 1x1 layer), 6 = direct 3x3 with
 * split operands on the f16 matrix cores (reads wpacked_d3h), 5 = Winograd
 * F(4x4,3x3) with split operands on the f16 matrix cores (reads wpacked_w4h), 4 = Winograd F(4x4,3x3) on the fp32 matrix cores
 * (reads wpacked_w4), 2 = Winograd F(2x2,3x3) (reads wpacked_wino), 1 = vector-pipe small-Cout kernel (reads
 * wpacked_sc), 0 = direct implicit GEMM (reads wpacked); -1 = NULL.  A host that packs ONE fragment order per layer asks this before packing (set the pointer it intends to fill to any
 * non-NULL value); a launch whose wpacked aliases Winograd fragments it would not read is refused with READ_EINVAL. */
int read_conv_kernel_family(const read_conv_desc *desc);
/* Number of compiled tile configurations and a printable name for each (for tuning sweeps). */
int read_conv_config_count(void);
const char *read_conv_config_name(int config);

/* out[y][x][c] = bilinear x4 (align_corners=False) of in, NHWC, C % 4 == 0. */
int read_bilinear_up4(const float *in, int inH, int inW, int C, float *out, void *stream);

/* ---------------------------------------------------------------- training step (csrc/train.hip)
 * Backward of one BasicConv, y = BN_eval(act(f) * sigmoid(m)) with [f | m] = conv_{f|m}(x) + b (READ/models/unet.py:44-53 under
 * torch.autograd in src/train.py:132-203); BatchNorm as the eval-mode affine map (configs/train_example.yaml eval_in_train).
 *   forward (training)  read_conv_pack_params_device + read_conv_pack_weights_device, read_gated_conv_forward with
 *                       desc.linear = 1 -> pre-activations [pixels][2*Cout], read_gate_forward -> y
 *   backward            read_gate_backward: dy, [f|m] -> dfm [pixels][2*Cp] (Cp = Cout rounded up to 8; df | dm) and the
 *                       per-channel sums S[4][Cout] += {sum df, sum dm, sum dy, sum dy*g} (ACCUMULATED: the caller zero-fills S —
 *                       one fill for all layers of a step; read_gate_backward_bn clears its own); read_bn_param_grads turns them
 *                       into db_f, db_m, dgamma, dbeta (accumulating);
 *                       dgrad: stride 1 -> read_conv_pack_dgrad_device + read_gated_conv_forward(linear = 1) over dfm with
 *                       Cout := Cin/2, zero biases (the same MFMA kernel with flipped, transposed weights);
 *                       stride 2 -> read_conv_dgrad_generic;
 *                       read_conv_wgrad: x, dfm -> dW_f, dW_m in the PyTorch layout (Cout, Cin, k, k), MFMA, split over
 *                       pixel rows with a scratch of read_conv_wgrad_scratch_floats().  3x3 / stride-1 layers with Cin % 32 == 0 on
 *                       images of whole 4 x 4 tiles are summed in the Winograd F(4x4,3x3) domain (dg = G^T [sum over tiles of
 *                       (B^T d B) . (A dY A^T)] G: a quarter of the multiplications; read_tuning_set("wgrad_wino", 0): direct
 *                       kernel); read_conv_wgrad_family says which: 4 or 0. */
int read_conv_pack_params_device(int Cout, const float *bf, const float *bm, const float *gamma, const float *beta,
                                 const float *mean, const float *var, float eps, float *params, void *stream);
int read_conv_pack_weights_device(int Cin, int Cout, int ksize, int kc, const float *wf, const float *wm, float *wpacked,
                                  void *stream);
/* Winograd fragments (3x3 / stride 1, Cin % 16 == 0) of the layer's own weights and of its dgrad weights, produced on the
 * device: with desc.wpacked_wino set, linear-mode launches of eligible layers take the Winograd kernel too. */
int read_conv_pack_wino_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_wino, void *stream);
size_t read_conv_dgrad_wino_floats(int Cin, int Cout);
int read_conv_pack_dgrad_wino_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_wino, void *stream);
/* ... and the Winograd F(4x4,3x3) fragments (read_conv_pack_w4_host's order; desc.wpacked_w4): linear-mode launches of layers
 * with Cin >= 32 and Cout % 32 == 0 (the dgrad's virtual layer: 2 * pad8(Cout) input and Cin / 2 gated output channels) then run
 * on the F(4x4) kernel, which stores the pre-activations and — with desc.out_gated — the gated output in the same pass. */
int read_conv_pack_w4_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_w4, void *stream);
size_t read_conv_dgrad_w4_floats(int Cin, int Cout);
int read_conv_pack_dgrad_w4_device(int Cin, int Cout, const float *wf, const float *wm, float *wpacked_w4, void *stream);
size_t read_conv_dgrad_packed_floats(int Cin, int Cout, int ksize);
int read_conv_pack_dgrad_device(int Cin, int Cout, int ksize, int kc, const float *wf, const float *wm, float *wpacked,
                                void *stream);
/* All packing jobs of a training step in ONE launch.  The optimizer changes every weight every step, so a step re-packs the
 * parameter block, the forward fragments and the dgrad fragments of every layer: the entry points above are one launch each
 * (297 per step for READ's UNet).  A host that knows its layers builds a table of jobs ONCE (the tensors keep their addresses
 * across optimizer steps), uploads it, and calls read_conv_pack_batch once per step.
 *   kind READ_PACK_PARAMS: out[4 * pad32(Cout)] = bias_f, bias_m, bn scale, bn shift   (read_conv_pack_params_device)
 *        READ_PACK_DIRECT / _WINO / _W4: the fragment order of read_conv_pack_weights_device / _wino_device / _w4_device;
 *        mode 1 = the layer's dgrad fragments (read_conv_pack_dgrad_device / _dgrad_wino_device / _dgrad_w4_device).
 * read_conv_pack_job_prepare (host) validates a job and fills Cp / total / nblocks; the caller then sets first_block to the
 * running sum of nblocks and passes the grand total as total_blocks. */
enum { READ_PACK_PARAMS = 0, READ_PACK_DIRECT = 1, READ_PACK_WINO = 2, READ_PACK_W4 = 3 };
typedef struct read_pack_job {
    int kind, mode;                         /* mode: 0 the layer's own weights, 1 its dgrad */
    int Cin, Cout, ksize, kc;               /* kc: channel chunk of the direct order (8 / 16 / 32) */
    int Cp, first_block, nblocks;           /* derived (prepare) / prefix sum (caller) */
    float eps;                              /* params job: BatchNorm epsilon */
    const float *wf, *wm;                   /* (Cout, Cin, k, k) conv_f / conv_m weights */
    const float *bf, *bm, *gamma, *beta, *mean, *var;   /* params job (bf / bm may be NULL) */
    float *out;
    long long total;                        /* floats written (derived) */
} read_pack_job;
int read_conv_pack_job_prepare(read_pack_job *job);
int read_conv_pack_batch(const read_pack_job *jobs_dev, int njobs, int total_blocks, void *stream);
/* A training batch is ONE tall image: the items stacked vertically, block_h rows per item of which the first valid_h are
 * the item and the rest a separator that stays zero in every activation — it is the zero padding between neighbours, so the
 * convolutions need no batch dimension and one launch covers the batch.  The gate kernels keep the separators zero (forward)
 * and give them zero gradient (backward); W = row length in pixels; block_h = 0 means a single image. */
int read_gate_forward(const float *fm, int64_t pixels, int Cout, const float *params, int elu, const float *residual,
                      float *y, int W, int block_h, int valid_h, void *stream);
int read_gate_backward(const float *dy, const float *fm, int64_t pixels, int Cout, const float *params, int elu, float *dfm,
                       float *sums, int W, int block_h, int valid_h, void *stream);
int read_bn_param_grads(int Cout, const float *sums, const float *mean, const float *var, float eps, float *dbf, float *dbm,
                        float *dgamma, float *dbeta, void *stream);
/* Batch-statistics BatchNorm — nn.BatchNorm2d in .train() (READ/models/unet.py:40,51; the reference's default training mode,
 * train.py:271-279,450).  read_bn_train_forward: g = act(f) * sigmoid(m) [pixels][C] (produced with an identity BatchNorm in the
 * params block: scale 1, shift 0) is normalised IN PLACE with the per-channel mean / biased variance of the valid pixels
 * (fp64 accumulation): y = (g - mean) / sqrt(var + eps) * gamma + beta; separator rows of a stacked batch stay zero and do
 * not count.
 * `groups` = number of statistic groups: 1 — the whole stacked batch is ONE BatchNorm batch (what nn.BatchNorm2d does with a
 * (B,C,h,w) tensor: UNet.forward on a batch); B = the number of stacked items — every item is normalised with ITS OWN statistics
 * and the running buffers move B times, in item order: the reference's NetAndTexture.forward calls the net once per batch item
 * (READ/models/compose.py:137-176), so its BatchNorm layers see N = 1.  stat[groups][2][C] receives {mean, biased var} per group
 * (kept for the backward pass), scale_shift (groups * 2 * pad32(C) floats) the scale / shift rows, running_mean / running_var
 * (may be NULL) move by `momentum` once per group (variance unbiased, n / (n - 1)), scratch = groups * 2 * C doubles.
 * read_gate_backward_bn: read_gate_backward through that BatchNorm: two passes (sums of dy and dy * g, then
 * dg = gamma r (dy - mean(dy) - xhat mean(dy xhat))) per group; sums[groups][4][Cout] as read_gate_backward per group, so
 * read_bn_param_grads_groups(Cout, groups, sums, stat, ...) gives db_f, db_m, dgamma, dbeta (accumulating, summed over the
 * groups); abc = groups * 3 * Cout floats of scratch. */
int read_bn_train_forward(float *g_to_y, int64_t pixels, int C, int W, int block_h, int valid_h, int groups, const float *gamma,
                          const float *beta, float eps, float momentum, float *running_mean, float *running_var, float *stat,
                          float *scale_shift, double *scratch, void *stream);
int read_gate_backward_bn(const float *dy, const float *fm, int64_t pixels, int Cout, const float *params, int elu, float *dfm,
                          float *sums, int W, int block_h, int valid_h, int groups, const float *stat, const float *gamma,
                          float eps, float *abc, void *stream);
int read_bn_param_grads_groups(int Cout, int groups, const float *sums, const float *stat, float eps, float *dbf, float *dbm,
                               float *dgamma, float *dbeta, void *stream);
size_t read_conv_dgrad_generic_floats(int Cin, int Cout, int ksize);
int read_conv_dgrad_generic(const float *dfm, int outH, int outW, int Cin, int Cout, int ksize, int stride, const float *wf,
                            const float *wm, float *wscratch, int inH, int inW, float *dx, void *stream);
size_t read_conv_wgrad_scratch_floats(int Cin, int Cout, int ksize, int outH);
int read_conv_wgrad(const float *x, int inH, int inW, int Cin, const float *dfm, int Cout, int ksize, int stride, float *dwf,
                    float *dwm, int accumulate, float *scratch, size_t scratch_floats, void *stream);
int read_conv_wgrad_family(int Cin, int ksize, int stride, int inH, int inW);
/* read_bilinear_up4 for a vertically stacked batch (rows interpolate inside an item only) and the adjoint of both:
 * dout [4*inH][4*inW][C] -> din [inH][inW][C] */
int read_bilinear_up4_blocks(const float *in, int inH, int inW, int C, float *out, int block_h, int valid_h, void *stream);
int read_bilinear_up4_backward(const float *dout, int inH, int inW, int C, float *din, int block_h, int valid_h, void *stream);
/* F.huber_loss(out, target) (delta 1, mean; src/READ/models/compose.py:35,38): *loss_sum = sum of the per-element losses
 * (divide by n), grad[i] = grad_scale * clip(out - target, -1, 1).  Either output may be NULL. */
int read_huber_loss(const float *out, const float *target, int64_t n, float grad_scale, float *loss_sum, float *grad,
                    void *stream);
/* RMSprop (READ/pipelines/ogl.py:16,99-100 defaults: alpha 0.99, eps 1e-8) over the descriptor ROWS the step touched only:
 * ids = the step's int32 index maps; every touched row is updated once, its gradient row is zeroed again, and the decay
 * of the steps in which it was not touched (gradient 0 in the dense optimizer) is applied lazily from `stamp`
 * (int32 per row, zero-initialised; step counts from 1).  rows / sq / grad are N x C. */
int read_rmsprop_sparse(float *rows, float *sq, float *grad, int32_t *stamp, int C, int64_t n_rows, const int32_t *ids,
                        int64_t n_ids, int step, float lr, float alpha, float eps, void *stream);
/* The same update without the N x C gradient table: the step's (id, gradient row) pairs sorted by id (sorted_ids ascending,
 * perm[i] = row of g that sorted_ids[i] came from, e.g. torch.sort); runs of equal ids are summed in sorted order
 * (deterministic) and applied once.  scratch: read_rmsprop_sorted_scratch_ints() int32 of device memory. */
size_t read_rmsprop_sorted_scratch_ints(void);
int read_rmsprop_sorted(float *rows, float *sq, int32_t *stamp, int C, int64_t n_rows, const int32_t *sorted_ids,
                        const int64_t *perm, const float *g, int64_t n, int step, float lr, float alpha, float eps,
                        int32_t *scratch, void *stream);

/* ---------------------------------------------------------------- UNet */

typedef struct read_unet read_unet_t;

/* The 101 BasicConvs of READ's UNet in canonical order (99 execute; ConvsOut.* are packed but
 * never run, unet.py:181-186).  For layer i: state-dict path prefix, Cin, Cout, ksize. */
int read_unet_layer_count(void);
int read_unet_layer_info(int i, const char **path, int *cin, int *cout, int *ksize, int *stride,
                         int *elu);
/* Total floats of the raw parameter blob: per layer, in order,
 *   conv_f.weight, conv_f.bias, conv_m.weight, conv_m.bias, norm.weight, norm.bias,
 *   norm.running_mean, norm.running_var. */
size_t read_unet_raw_floats(void);
size_t read_unet_packed_floats(void);
int read_unet_pack_host(const float *raw_host, float bn_eps, float *packed_host);
/* Layout of the packed blob.  FULL (the two functions above): every fragment order of every layer — direct, both F(2x2) orders,
 * F(4x4) where it exists —, so any tuning knob can send a layer to any of its kernels: 952 MB for READ's 121 MB of weights.
 * LEAN: what the default plan reads — a layer the F(4x4) kernel runs (73 of the 105 launches) carries its F(4x4) order only,
 * layers that no launch executes (ConvsOut) carry nothing: 451 MB.  read_unet_create_layout refuses a lean blob (READ_EINVAL)
 * when, under the current tuning state or at a size whose tensors reach 2 GiB, one of those layers would not run on the F(4x4)
 * kernel.  The raw blob is the same for both layouts. */
enum { READ_UNET_LAYOUT_FULL = 0, READ_UNET_LAYOUT_LEAN = 1 };
size_t read_unet_packed_floats_layout(int layout);
int read_unet_pack_host_layout(const float *raw_host, float bn_eps, float *packed_host, int layout);
size_t read_unet_workspace_bytes(int H, int W);
/* H, W multiples of 16 (READ/gl/nn.py:107-109).  `packed` and `ws` are device memory that must
 * outlive the handle. */
int read_unet_create(read_unet_t **out, const float *packed, int H, int W, void *ws, size_t ws_bytes);
int read_unet_create_layout(read_unet_t **out, const float *packed, int H, int W, void *ws, size_t ws_bytes, int layout);
void read_unet_destroy(read_unet_t *u);
/* x0..x3: NHWC [H>>l][W>>l][8] feature pyramids; rgb: NHWC [H][W][rgb_cstride], channels 0..2
 * written (channel 3 set to 1.0f when rgb_cstride == 4, the viewer's RGBA frame, nn.py:123-124). */
int read_unet_forward(read_unet_t *u, const float *x0, const float *x1, const float *x2,
                      const float *x3, float *rgb, int rgb_cstride, void *stream);
/* Per-layer timing of the last instrumented forward: runs one forward with hipEvents around
 * every launch and fills ms[0..n) (n = read_unet_launch_count()).  Synchronises. */
int read_unet_launch_count(read_unet_t *u);
const char *read_unet_launch_label(read_unet_t *u, int i);
/* Static facts about launch i of the plan: algorithmic FLOPs (2*MAC, both gate convs), whether it
 * is one of the dominant 3x3/s1 C->C gated convs, output size and channel counts. */
int read_unet_launch_info(read_unet_t *u, int i, double *flops, int *is_conv3x3_s1, int *outH,
                          int *outW, int *cin, int *cout);
int read_unet_profile(read_unet_t *u, const float *x0, const float *x1, const float *x2,
                      const float *x3, float *rgb, int rgb_cstride, void *stream, float *ms,
                      double *flops, int *is_conv3x3_s1);
/* Device pointer of an intermediate activation by name ("res1", "z8", ...), for layer-level tests. */
const float *read_unet_debug_tensor(read_unet_t *u, const char *name, int *H, int *W, int *C);

#ifdef __cplusplus
}
#endif
#endif /* READ_HIP_H */
