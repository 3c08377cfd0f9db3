// Shared host-side helpers for libsyn3r_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/syn3r_hip.h"

namespace syn3r {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SYN3R_E_HIP;
    }
    return SYN3R_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remembered per (kernel, device), so a host that
// drives several GPUs from one process gets the large-LDS attribute on each of them.
struct DevOnce { std::atomic<unsigned long long> done{0}; };
inline int set_max_lds(DevOnce& once, const void* fn, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;       // devices beyond 63: set on every launch
    if (bit && (once.done.load(std::memory_order_acquire) & bit)) return SYN3R_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return check_hip(e, what);
    once.done.fetch_or(bit, std::memory_order_release);
    return SYN3R_OK;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Optional per-kernel timing with HIP events on the launch stream (syn3r_trace_* in the ABI).
bool trace_on();
bool trace_detail();   // syn3r_trace_enable(2): contraction launches carry their shape in the name
bool trace_open(const char* name, hipEvent_t* start, hipEvent_t* stop);   // false: this kernel is filtered out

// 4x4 / 3x3 matrices travel to kernels by value.
struct Mat4f { float m[16]; };
struct Mat3f { float m[9]; };
struct Mat4d { double m[16]; };
struct Mat3d { double m[9]; };

}  // namespace syn3r

// Dispatch switches for A/B measurements exist only in developer builds (-DSYN3R_TUNING through SYN3R_EXTRA_HIPCC_FLAGS,
// syn3r_amd/build.py): the shipped library reads no environment variable and holds no mutable process-wide state
// (include/syn3r_hip.h; tests/test_abi_cpu.py checks the sources and the built library for getenv).
#ifdef SYN3R_TUNING
#include <stdlib.h>
inline int tune_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int tune_env(const char*, int dflt) { return dflt; }
#endif

// Upper bound of every int size argument of the ABI (rows, channels, image sides, Gaussians ...): checked FIRST, so the
// size arithmetic behind the other checks (products, round-ups) stays inside int / long long whatever the caller passes
// (tests/test_abi_sanitize.py drives every entry point with INT_MAX and friends under UBSan).
#define SYN3R_DIM_MAX (1 << 24)
#define SYN3R_DIM_OK(x) ((x) > 0 && (x) <= SYN3R_DIM_MAX)
#define SYN3R_SIDE_MAX (1 << 15)                 /* image height / width */
#define SYN3R_SIDE_OK(x) ((x) > 0 && (x) <= SYN3R_SIDE_MAX)

#define SYN3R_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            syn3r::set_error(__VA_ARGS__);       \
            return SYN3R_E_INVALID;              \
        }                                        \
    } while (0)

// Kernel launch the tracer can time.  A traced launch goes through hipExtLaunchKernelGGL, which attaches the
// start / stop events to the dispatch packet itself (the kernel's own begin / end timestamps): no marker
// packets before and after the kernel, so tracing costs a fraction of two hipEventRecord calls per launch.
#define SYN3R_LAUNCH_NAMED(name, kernel, grid, block, shmem, stream, ...)                                  \
    do {                                                                                                   \
        hipEvent_t ea__ = nullptr, eb__ = nullptr;                                                         \
        if (syn3r::trace_on() && syn3r::trace_open(name, &ea__, &eb__))                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ea__, eb__, 0, __VA_ARGS__);          \
        else                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
    } while (0)
#define SYN3R_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    SYN3R_LAUNCH_NAMED(#kernel, kernel, grid, block, shmem, stream, __VA_ARGS__)

#define SYN3R_LAUNCH_CHECK(name)                                        \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) return syn3r::check_hip(e__, name);      \
    } while (0)

// Workgroups are dealt to the 8 XCDs round-robin by blockIdx; this bijection hands every XCD one CONTIGUOUS
// chunk of logical ids, so neighbouring tiles (which share operands) meet in the same 4 MB L2.
__device__ __forceinline__ unsigned xcd_chunk_remap(unsigned bid, unsigned nblk) {
    unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// Wave-level reductions (64 lanes).
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): ~14 VALU operations instead of libm's ~40.
// Used by the GELU of the GEGLU gate, whose result is rounded to fp16 (relative step 4.9e-4).
__device__ __forceinline__ float fast_erff(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.0f + fast_erff(g * 0.70710678118654752f)); }

// Exact (erf) GELU of TWO values on packed fp32 arithmetic:  gelu(g) = g * Phi(g),  Phi(g) = 1/2 + g * P(g^2)  with P a
// degree-12 polynomial (Chebyshev fit of (Phi(g) - 1/2) / g on |g| <= 5, evaluated by Horner in t = 2 g^2 / 25 - 1) and g
// clamped to [-5, 5] inside Phi (beyond it Phi is 0 / 1 to 3e-7).  |error| <= 2.3e-6 on gelu (fp32 Horner, checked
// against scipy erf over [-7, 7]): 25x below the fp16 rounding step of the gate output it feeds.  No transcendental, no
// reciprocal: 15 v_pk_fma_f32 / v_pk_mul_f32 per pair, i.e. ~7 issue slots per value against ~26 for gelu_erf.
typedef float syn3r_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ syn3r_f2 gelu_pk(syn3r_f2 g) {
    const syn3r_f2 gc = (syn3r_f2){__builtin_amdgcn_fmed3f(g.x, -5.0f, 5.0f), __builtin_amdgcn_fmed3f(g.y, -5.0f, 5.0f)};
    const syn3r_f2 t = gc * gc * 0.08f - 1.0f;
    syn3r_f2 a = t * 7.695898276e-04f + -1.685841001e-03f;
    a = a * t + 1.275028536e-03f;
    a = a * t + -2.503029935e-03f;
    a = a * t + 6.874790886e-03f;
    a = a * t + -1.132975735e-02f;
    a = a * t + 1.618207803e-02f;
    a = a * t + -2.320382084e-02f;
    a = a * t + 3.148886876e-02f;
    a = a * t + -4.045282865e-02f;
    a = a * t + 5.151694415e-02f;
    a = a * t + -7.029583794e-02f;
    a = a * t + 1.413638313e-01f;
    return g * (gc * a + 0.5f);
}
