// Micro-benchmark (developer tool): issue rate of v_exp_f32 / v_fma_f32 / v_pk_fma_f32 / MFMA mixes on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    half8 x; for (int i = 0; i < 8; ++i) x[i] = (_Float16)1.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 1) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 2) {   // 4 exp + 4 fma interleaved
            asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) {   // 1 MFMA 32x32x16 + 8 fma
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 4) {   // 1 MFMA only
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc, 0, 0, 0);
        } else if (MODE == 5) {   // 1 MFMA + 8 exp
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    if (MODE == 9 || MODE == 10) {   // packed fp32: 4 independent v_pk_fma_f32 / 2 v_pk_fma_f32 + 2 v_fma_f32 (round 6: what a packed instruction costs the SIMD)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
        for (int it = 0; it < iters; ++it) {
            if (MODE == 9)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            else
                asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        }
        a0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    }
    if (MODE >= 6 && MODE <= 8) {
        // phase-structured like attention: 8 MFMAs (two dependent chains of 4) -> NV dependent VALU ops -> 8 MFMAs
        float16v b0 = acc, b1 = acc, c0 = acc, c1 = acc;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, b0, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, b1, 0, 0, 0);
            }
            a0 += b0[0]; a1 += b1[0];
            constexpr int NV = MODE == 6 ? 12 : (MODE == 7 ? 12 : 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (MODE == 6)
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
                else
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
            half8 y = x; y[0] = (_Float16)a0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) a0 += b0[i] + b1[i] + c0[i] + c1[i];
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float ops_per_iter) {
    int iters = (MODE >= 6 && MODE <= 8) ? 2000 : 20000;
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves = 1 per SIMD) per block
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 100);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: waves_per_simd waves x iters x ops
    double ns_per_wave_iter = ms * 1e6 / ((double)iters * waves_per_simd);
    printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f ns per wave-iteration (%.0f instr) -> %.2f ns/instr\n", name, waves_per_simd, ms,
           ns_per_wave_iter, ops_per_iter, ns_per_wave_iter / ops_per_iter);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("8 x v_exp_f32", w, 8);
        run<1>("8 x v_fma_f32", w, 8);
        run<2>("4 exp + 4 fma", w, 8);
        run<4>("1 mfma 32x32x16", w, 1);
        run<3>("1 mfma + 8 fma", w, 9);
        run<5>("1 mfma + 8 exp", w, 9);
        run<9>("4 x v_pk_fma_f32", w, 4);
        run<10>("2 v_pk_mul + 2 v_pk_add f32", w, 4);
    }
    for (int w : {1, 2, 3, 4}) {
        run<8>("phases: 16 mfma only", w, 16);
        run<6>("phases: 16 mfma + 96 fma", w, 112);
        run<7>("phases: 16 mfma+72fma+24exp", w, 112);
    }
    return 0;
}
