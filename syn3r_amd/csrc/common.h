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
// polynomial (fit of (Phi(g) - 1/2) / g on |g| <= G, evaluated by Horner in t = 2 g^2 / G^2 - 1) and g clamped to [-G, G]
// inside Phi.  No transcendental, no reciprocal: DEG + 3 v_pk_fma_f32 / v_pk_mul_f32 per pair against ~26 issue slots per
// VALUE for gelu_erf.  The gate is bound by the SIMD's vector EXECUTION (a wave64 v_pk_fma_f32 is four passes of the 32-lane
// unit: two cycles per value and Horner step, whoever issues it), so its cost is its degree:
//   SYN3R_GELU_DEG 12  (rounds 2-5)  G = 5    |error| <= 2.3e-6 on gelu, 1.2-1.7 % of the fp16-rounded gate outputs differ by one
//                                             ulp from the correctly rounded erf form
//   SYN3R_GELU_DEG 10                G = 4.5  |error| <= 2.6e-5 (at the clamp: gelu(-4.5) = -1.5e-5 -> 0), rms 1.6e-6, 2.9 % one-ulp
//   SYN3R_GELU_DEG 9                 G = 4.5  |error| <= 3.7e-5, rms 7.6e-6, 8.4 % one-ulp
//   SYN3R_GELU_DEG 8   (round 6)     G = 4.25 |error| <= 6.3e-5 (8.5e-5 at |g| = 8: 1e-5 relative), rms 1.7e-5, 15.9 % one-ulp
// (weighted minimax fits, the weight = the tighter of fp16's half-ulp of gelu(+-g) and 1.5e-5; checked against scipy erf over
// [-8, 8] in fp32 Horner arithmetic; rms over N(0, 1.5) gates, whose fp16 rounding noise is ~1e-4 rms: the lower degrees add
// 0.01-3 % to the noise the fp16 output carries anyway.  The parity decision rests on the reference fixtures)
#ifndef SYN3R_GELU_DEG
#define SYN3R_GELU_DEG 8
#endif
typedef float syn3r_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ syn3r_f2 gelu_pk(syn3r_f2 g) {
#if SYN3R_GELU_DEG == 12
    constexpr float G = 5.0f, SC = 0.08f;
    constexpr float c[13] = {1.413638313e-01f, -7.029583794e-02f, 5.151694415e-02f, -4.045282865e-02f, 3.148886876e-02f,
                             -2.320382084e-02f, 1.618207803e-02f, -1.132975735e-02f, 6.874790886e-03f, -2.503029935e-03f,
                             1.275028536e-03f, -1.685841001e-03f, 7.695898276e-04f};
#elif SYN3R_GELU_DEG == 10
    constexpr float G = 4.5f, SC = 0.09876543209876543f;
    constexpr float c[11] = {1.569051000e-01f, -7.719027244e-02f, 5.468925326e-02f, -4.017315754e-02f, 2.841398309e-02f,
                             -1.872261455e-02f, 1.095662898e-02f, -5.753943520e-03f, 3.429505493e-03f, -2.064591753e-03f,
                             6.205179385e-04f};
#elif SYN3R_GELU_DEG == 9
    constexpr float G = 4.5f, SC = 0.09876543209876543f;
    constexpr float c[10] = {1.569051204e-01f, -7.718434032e-02f, 5.468310931e-02f, -4.028949651e-02f, 2.852618328e-02f,
                             -1.812994075e-02f, 1.038848351e-02f, -6.836305825e-03f, 4.474478846e-03f, -1.427198998e-03f};
#elif SYN3R_GELU_DEG == 8
    constexpr float G = 4.25f, SC = 0.11072664359861592f;
    constexpr float c[9] = {1.659352130e-01f, -8.078260179e-02f, 5.572614317e-02f, -3.908646575e-02f, 2.530956491e-02f,
                            -1.488701428e-02f, 9.392296479e-03f, -5.790942058e-03f, 1.829005353e-03f};
#else
#error "SYN3R_GELU_DEG must be 8, 9, 10 or 12"
#endif
    const syn3r_f2 gc = (syn3r_f2){__builtin_amdgcn_fmed3f(g.x, -G, G), __builtin_amdgcn_fmed3f(g.y, -G, G)};
    const syn3r_f2 t = gc * gc * SC - 1.0f;
    syn3r_f2 a = t * c[SYN3R_GELU_DEG] + c[SYN3R_GELU_DEG - 1];
#pragma unroll
    for (int k = SYN3R_GELU_DEG - 2; k >= 0; --k) a = a * t + c[k];
    return g * (gc * a + 0.5f);
}

// The published 3DGS parameter activations (GaussianModel.get_scaling / get_rotation / get_opacity: exp, normalize, sigmoid) and
// their chain rule, as torch writes them (normalize = x / max(|x|_2, 1e-12); d/dx = (g - xhat (xhat . g)) / max(|x|, eps);
// sigmoid' = s (1 - s); exp' = e).  ONE arithmetic for k_activate[_bwd] (train.hip, built without fma contraction) and for the
// raw-parameter rasteriser entries (raster_fwd.hip / raster_bwd.hip, built with it): contraction is off inside these bodies, so
// both routes leave the same bits.
__device__ __forceinline__ float act_exp(float x) { return expf(x); }
__device__ __forceinline__ float act_sigmoid(float x) {
#pragma clang fp contract(off)
    return 1.0f / (1.0f + expf(-x));
}
__device__ __forceinline__ float act_quat_inv_norm(float4 q) {
#pragma clang fp contract(off)
    return 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
}
__device__ __forceinline__ float4 act_quat(float4 q, float inv) { return make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv); }
__device__ __forceinline__ float4 act_quat_bwd(float4 h /*normalised*/, float4 g, float inv) {
#pragma clang fp contract(off)
    const float dot = h.x * g.x + h.y * g.y + h.z * g.z + h.w * g.w;
    return make_float4((g.x - h.x * dot) * inv, (g.y - h.y * dot) * inv, (g.z - h.z * dot) * inv, (g.w - h.w * dot) * inv);
}
__device__ __forceinline__ float act_sigmoid_bwd(float s, float g) {
#pragma clang fp contract(off)
    return g * s * (1.0f - s);
}
