/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * Minimal CUDA-on-CPU environment so that the reference's DepthProject kernel source
 * (MyRender/CloudProjection/point_render.cu:1-167, compiled IN PLACE from /root/reference by
 * oracle/build_ref.sh — never copied) builds with g++ and can be executed serially, one
 * emulated thread per point, as the reference's launch does (point_render.cu:179-192).
 * Nothing here is derived from CUDA headers: it declares just the names that
 * helper_math.h and point_render.cu use. */
#ifndef ORACLE_SHIM_CUDA_RUNTIME_H
#define ORACLE_SHIM_CUDA_RUNTIME_H
#include <math.h>
#include <stdlib.h>

#define __CUDACC__ 1            /* skip helper_math.h's host re-definitions of fminf/fmaxf */
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

#define ORACLE_VEC(T, N2, N3, N4)                                                       \
    struct N2 { T x, y; };                                                              \
    struct N3 { T x, y, z; };                                                           \
    struct N4 { T x, y, z, w; };                                                        \
    static inline N2 make_##N2(T x, T y) { N2 r; r.x = x; r.y = y; return r; }          \
    static inline N3 make_##N3(T x, T y, T z) { N3 r; r.x = x; r.y = y; r.z = z; return r; } \
    static inline N4 make_##N4(T x, T y, T z, T w) { N4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
ORACLE_VEC(float, float2, float3, float4)
ORACLE_VEC(int, int2, int3, int4)
ORACLE_VEC(unsigned int, uint2, uint3, uint4)
#undef ORACLE_VEC

static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

/* Serial execution: one emulated thread at a time, so the CAS always succeeds when the
 * lock is free — semantics of CUDA's atomicCAS (returns the old value). */
static inline int atomicCAS(int *addr, int compare, int val)
{
    int old = *addr;
    if (old == compare) *addr = val;
    return old;
}
#endif
