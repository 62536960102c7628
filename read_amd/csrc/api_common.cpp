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
extern "C" int read_abi_version(void) { return 1; }

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
void splat_set_pipe(int v);
void conv_set_trace(void *buf, size_t bytes);
void conv_set_prefer_wave(int v);
void conv_set_stagger(int ticks);
void conv_set_ablate(int bits);
void conv_set_wino(int max_cin);
void conv_set_kc32(int v);
void unet_set_streams(int v);
void splat_set_near(int v);
void splat_set_cells(int v);
void splat_set_seeds(int v);
void splat_set_l1(int v);
void splat_set_cells_sub(int v);
}

// Debug: per-workgroup timeline of the next gated-conv launches.  buf = device memory, 64 bytes per
// workgroup: s_memrealtime (100 MHz) at kernel entry / after the prologue / after the k-loop / at exit,
// HW_ID, XCC_ID, blockIdx.x, blockIdx.y.  NULL switches it off.
extern "C" int read_debug_set_trace(void *buf, size_t bytes)
{
    readhip::conv_set_trace(buf, bytes);
    return READ_OK;
}

// Tuning knobs for A/B measurements on the GPU box (not needed in production):
//   "splat_mode": 0 = per-XCD key images + L2-local atomics (default), 1 = one image + agent-scope
//                 atomics, 2 = projection only (timing floor; results invalid), 3 = one image,
//                 agent-scope atomics, system-scope (L2-bypassing) early-z reads.
extern "C" int read_tuning_set(const char *key, int value)
{
    READ_CHECK_ARG(key, "read_tuning_set: null key");
    if (!strcmp(key, "splat_mode")) {
        const int rc = readhip::splat_set_mode(value);
        if (rc) readhip::set_error("read_tuning_set: splat_mode must be 0..7");
        return rc;
    }
    if (!strcmp(key, "splat_pipe")) {
        readhip::splat_set_pipe(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_stats")) {
        readhip::splat_set_stats(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_subset")) {
        readhip::splat_set_subset(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_near")) {      // cell path: expected points per pixel in front of the pass-A split distance
        readhip::splat_set_near(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_l1")) {
        readhip::splat_set_l1(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_cells_sub")) {
        readhip::splat_set_cells_sub(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_seeds")) {     // 0: no warm start from the previous frame
        readhip::splat_set_seeds(value);
        return READ_OK;
    }
    if (!strcmp(key, "splat_cells")) {     // 0: ignore the cell-ordered copy
        readhip::splat_set_cells(value);
        return READ_OK;
    }
    if (!strcmp(key, "unet_streams")) {    // 0: the SCM chains stay on the caller's stream
        readhip::unet_set_streams(value);
        return READ_OK;
    }
    if (!strcmp(key, "conv_kc32")) {
        readhip::conv_set_kc32(value);
        return READ_OK;
    }
    if (!strcmp(key, "conv_wino")) {       // value = largest Cin that takes the Winograd kernel (0 = off)
        readhip::conv_set_wino(value);
        return READ_OK;
    }
    if (!strcmp(key, "conv_ablate")) {
        readhip::conv_set_ablate(value);
        return READ_OK;
    }
    if (!strcmp(key, "conv_stagger")) {
        readhip::conv_set_stagger(value);
        return READ_OK;
    }
    if (!strcmp(key, "conv_wave")) {
        readhip::conv_set_prefer_wave(value != 0);
        return READ_OK;
    }
    readhip::set_error("read_tuning_set: unknown key '%s'", key);
    return READ_EINVAL;
}
