// Error plumbing + device query of the C ABI.
#include <stdarg.h>

#include "common.h"

namespace readhip {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace readhip

extern "C" const char *read_last_error(void) { return readhip::g_err; }
extern "C" int read_abi_version(void) { return 3; }   // 2: read_conv_desc.wpacked_w4h / wpacked_d3h; 3: wpacked_t3h at the end of the struct (round 6)

extern "C" int read_device_arch(char *name, int len)
{
    READ_CHECK_ARG(name && len > 0, "read_device_arch: bad buffer");
    int dev = 0;
    READ_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    READ_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(name, (size_t)len, "%s", p.gcnArchName);
    char *colon = strchr(name, ':');
    if (colon) *colon = 0;
    return READ_OK;
}

namespace readhip {
int splat_set_mode(int m);
void splat_set_subset(int v);
void splat_set_stats(int v);
void splat_set_near(int v);
void splat_set_cells(int v);
void splat_set_seeds(int v);
void splat_set_cells_sub(int v);
void splat_set_items(int v);
void splat_set_strips(int v);
void splat_set_wgs(int v);
void splat_set_zl2(int v);
void splat_set_lds(int v);
void splat_set_bins(int v);
void splat_set_ahead(int v);
void splat_set_prof(int v);
void splat_set_mark(int v);
void splat_set_cells_batch(int v);
void splat_set_compact(int v);
void splat_set_wgs_b(int v);
void splat_set_sticky(int v);
void splat_set_kslot(int v);
int splat_get(const char *key, int *value);
void conv_set_trace(void *buf, size_t bytes);
void conv_set_prefer_wave(int v);
void conv_set_stagger(int ticks);
void conv_set_ablate(int bits);
void conv_set_wino(int max_cin);
void conv_set_kc32(int v);
void conv_set_w16(int v);
void conv_set_abl(int v);
void conv_set_w4_grid(int v);
void conv_set_w4(int v);
void conv_set_w4h(int v);
void conv_set_d3h(int v);
void conv_set_d3h_fam(int v);
void conv_set_d3h_s2(int v);
void conv_set_pxh(int v);
void conv_set_t3h(int v);
void conv_set_w4h_waves(int v);
void conv_set_px(int v);
void conv_set_sc(int v);
void conv_set_w4x2(int v);
void conv_set_wino_wgs(int v);
int conv_get(const char *key, int *value);
void unet_set_streams(int v);
void train_set_wgrad_wino(int v);
int train_get(const char *key, int *value);
void unet_set_aff_split(int v);
void unet_set_up_fold(int v);
int unet_get(const char *key, int *value);
}

// Debug: per-workgroup timeline of the next gated-conv launches.  buf = device memory, 64 bytes per
// workgroup: s_memrealtime (100 MHz) at kernel entry / after the prologue / after the k-loop / at exit,
// HW_ID, XCC_ID, blockIdx.x, blockIdx.y.  NULL switches it off.
#ifdef READ_DEBUG_KNOBS
extern "C" int read_debug_set_trace(void *buf, size_t bytes)
{
    readhip::conv_set_trace(buf, bytes);
    return READ_OK;
}
#endif

// Tuning knobs for A/B measurements on the GPU box (not needed in production).  Every knob of the release library
// selects between implementations that produce the SAME results; the attribution probes whose results are invalid
// ("conv_ablate") exist only in builds with -DREAD_DEBUG_KNOBS.
static const char *const k_tuning_keys[] = {"splat_mode", "splat_stats", "splat_subset", "splat_near", "splat_cells",
                                            "splat_cells_sub", "splat_seeds", "splat_items", "splat_strips", "splat_wgs", "splat_zl2", "splat_lds", "splat_bins", "splat_ahead", "splat_prof", "splat_mark", "splat_cells_batch", "splat_compact", "splat_sticky", "splat_wgs_b", "splat_kslot", "unet_streams", "unet_aff_split", "unet_up_fold", "conv_kc32", "conv_px", "conv_sc", "conv_wino_wgs",
                                            "conv_wino", "conv_w16", "conv_w4", "conv_w4h", "conv_d3h", "conv_d3h_fam", "conv_d3h_s2", "conv_pxh", "conv_t3h", "conv_w4_grid", "conv_stagger", "conv_wave", "wgrad_wino",
#ifdef READ_DEBUG_KNOBS
                                            "conv_ablate", "conv_abl", "conv_w4x2", "conv_w4h_waves",
#endif
                                            nullptr};

extern "C" int read_tuning_set(const char *key, int value)
{
    READ_CHECK_ARG(key, "read_tuning_set: null key");
    if (!strcmp(key, "splat_mode")) {
        const int rc = readhip::splat_set_mode(value);
        if (rc) readhip::set_error("read_tuning_set: splat_mode must be 1 (agent atomics) or 7 (warm start + hi-z)");
        return rc;
    }
    if (!strcmp(key, "splat_stats")) { readhip::splat_set_stats(value); return READ_OK; }
    if (!strcmp(key, "splat_subset")) { readhip::splat_set_subset(value); return READ_OK; }
    // cell path: expected points per pixel in front of the pass-A split distance
    if (!strcmp(key, "splat_near")) { readhip::splat_set_near(value); return READ_OK; }
    if (!strcmp(key, "splat_cells_sub")) { readhip::splat_set_cells_sub(value); return READ_OK; }
    if (!strcmp(key, "splat_seeds")) { readhip::splat_set_seeds(value); return READ_OK; }     // 0: no warm start
    if (!strcmp(key, "splat_cells")) { readhip::splat_set_cells(value); return READ_OK; }     // 0: ignore the cell-ordered copy
    if (!strcmp(key, "splat_items")) { readhip::splat_set_items(value); return READ_OK; }     // work items per chunk: 1, 2, 4
    if (!strcmp(key, "splat_kslot")) { readhip::splat_set_kslot(value); return READ_OK; }     // key-image layout: 0 linear, 1 strided, 2 scattered
    if (!strcmp(key, "splat_lds")) { readhip::splat_set_lds(value); return READ_OK; }         // 0: no LDS table in front of the atomics
    if (!strcmp(key, "splat_bins")) { readhip::splat_set_bins(value); return READ_OK; }       // 0: pass A with one atomic per candidate
    if (!strcmp(key, "splat_ahead")) { readhip::splat_set_ahead(value); return READ_OK; }     // 0: never fold the next frame's first launch into this frame's last
    if (!strcmp(key, "splat_wgs_b")) { readhip::splat_set_wgs_b(value); return READ_OK; }     // workgroups per CU of pass B (0: as pass A)
    if (!strcmp(key, "splat_cells_batch")) { readhip::splat_set_cells_batch(value); return READ_OK; }   // 0: camera batches on the plain pass
    if (!strcmp(key, "splat_compact")) { readhip::splat_set_compact(value); return READ_OK; }  // 0: pass A bins its candidates from four masked slots per lane
    if (!strcmp(key, "splat_mark")) { readhip::splat_set_mark(value); return READ_OK; }       // 0: only pass-B survivors are promoted into list A
    if (!strcmp(key, "splat_sticky")) { readhip::splat_set_sticky(value); return READ_OK; }   // frames a front chunk stays in list A
    if (!strcmp(key, "splat_prof")) { readhip::splat_set_prof(value); return READ_OK; }       // events around the cell path's launches
    if (!strcmp(key, "splat_zl2")) { readhip::splat_set_zl2(value); return READ_OK; }         // 1: early-z loads bypass the L1
    if (!strcmp(key, "splat_wgs")) { readhip::splat_set_wgs(value); return READ_OK; }         // workgroups per CU of the passes
    if (!strcmp(key, "splat_strips")) { readhip::splat_set_strips(value); return READ_OK; }   // column strips: 1, 2, 4, 8
    if (!strcmp(key, "unet_streams")) { readhip::unet_set_streams(value); return READ_OK; }   // 0: SCM chains on the caller's stream
    // 0: AFF first convs as single 480-channel launches (takes effect for plans created afterwards)
    if (!strcmp(key, "unet_aff_split")) { readhip::unet_set_aff_split(value != 0); return READ_OK; }
    // 0: Upsample4(bilinear) as a separate pass and Convs.k over the concat (takes effect for plans created afterwards)
    if (!strcmp(key, "unet_up_fold")) { readhip::unet_set_up_fold(value != 0); return READ_OK; }
    if (!strcmp(key, "conv_wino_wgs")) { readhip::conv_set_wino_wgs(value); return READ_OK; } // persistent Winograd workgroups per CU: 1 or 2
    if (!strcmp(key, "conv_px")) { readhip::conv_set_px(value); return READ_OK; }             // pixel-lane kernel for 1x1 layers
    if (!strcmp(key, "conv_sc")) { readhip::conv_set_sc(value); return READ_OK; }             // vector-pipe kernel for Cout <= 4
    if (!strcmp(key, "conv_kc32")) { readhip::conv_set_kc32(value); return READ_OK; }
    if (!strcmp(key, "conv_w4_grid")) { readhip::conv_set_w4_grid(value); return READ_OK; }   // F(4x4): equal units per workgroup
#ifdef READ_DEBUG_KNOBS
    if (!strcmp(key, "conv_w4h_waves")) { readhip::conv_set_w4h_waves(value); return READ_OK; }   // 8: specialised waves (measured slower); 4: the product kernel
#endif
    if (!strcmp(key, "conv_t3h")) { readhip::conv_set_t3h(value); return READ_OK; }           // max Cin of 3x3 layers on the split-operand implicit-GEMM kernel (0 = off)
    if (!strcmp(key, "conv_pxh")) { readhip::conv_set_pxh(value); return READ_OK; }           // min Cin of 1x1 layers on the split-operand pixel-lane kernel (0 = off)
    if (!strcmp(key, "conv_d3h_s2")) { readhip::conv_set_d3h_s2(value); return READ_OK; }     // min Cin of 3x3 / stride-2 layers on the direct split-operand kernel (0 = off)
    if (!strcmp(key, "conv_d3h_fam")) { readhip::conv_set_d3h_fam(value); return READ_OK; }   // min Cin of FAM (x1 * x2) launches on the direct split-operand kernel (0 = off)
    if (!strcmp(key, "conv_d3h")) { readhip::conv_set_d3h(value); return READ_OK; }           // min Cin on the direct split-operand 3x3 kernel (f16 matrix cores; 0 = off)
    if (!strcmp(key, "conv_w4h")) { readhip::conv_set_w4h(value); return READ_OK; }           // min Cin on the split-operand F(4x4) kernel (f16 matrix cores; 0 = off)
    if (!strcmp(key, "conv_w4")) { readhip::conv_set_w4(value); return READ_OK; }             // min Cin on the Winograd F(4x4,3x3) kernel (0 = off)
    if (!strcmp(key, "conv_w16")) { readhip::conv_set_w16(value); return READ_OK; }           // wave-autonomous Winograd kernel (0 = row-per-wave)
    if (!strcmp(key, "conv_wino")) { readhip::conv_set_wino(value); return READ_OK; }         // largest Cin on the Winograd kernel (0 = off)
    if (!strcmp(key, "conv_stagger")) { readhip::conv_set_stagger(value); return READ_OK; }
    if (!strcmp(key, "conv_wave")) { readhip::conv_set_prefer_wave(value != 0); return READ_OK; }
    if (!strcmp(key, "wgrad_wino")) { readhip::train_set_wgrad_wino(value); return READ_OK; }   // 0: 3x3 weight gradients on the direct kernel
#ifdef READ_DEBUG_KNOBS
    if (!strcmp(key, "conv_ablate")) { readhip::conv_set_ablate(value); return READ_OK; }
    if (!strcmp(key, "conv_abl")) { readhip::conv_set_abl(value); return READ_OK; }          // probes of the 16x16x4 Winograd kernels
    if (!strcmp(key, "conv_w4x2")) { readhip::conv_set_w4x2(value); return READ_OK; }        // the two-waves-per-SIMD F(4x4) kernel (measured slower)
#endif
    readhip::set_error("read_tuning_set: unknown key '%s'", key);
    return READ_EINVAL;
}

extern "C" int read_tuning_get(const char *key, int *value)
{
    READ_CHECK_ARG(key && value, "read_tuning_get: null pointer");
    if (readhip::splat_get(key, value) || readhip::conv_get(key, value) || readhip::unet_get(key, value) || readhip::train_get(key, value)) return READ_OK;
    readhip::set_error("read_tuning_get: unknown key '%s'", key);
    return READ_EINVAL;
}

extern "C" const char *read_tuning_key(int i)
{
    int n = 0;
    while (k_tuning_keys[n]) ++n;
    return (i >= 0 && i < n) ? k_tuning_keys[i] : nullptr;
}
