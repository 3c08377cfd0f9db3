// Microbenchmark (developer tool): the tile loop of a persistent contraction kernel reduced to its memory streams - per tile NK
// "k-tiles", each = [every wavefront requests 9 KiB of operands (global_load_lds, 16 B per lane, as the kernels' LDS-DMA) |
// ALU-only wait of SPIN x 10 ns | s_waitcnt vmcnt(0) + barrier], then the tile's output: 20 stores of 1 KiB per wavefront.
// Question: what do the stores cost the NEXT tile's operand stream (vmcnt retires in order and counts stores; the HBM sees a write
// burst), and which arrangement avoids it?
//   mode 0: loads only (no stores)            mode 1: stores by every wavefront, loads by every wavefront (the kernels today)
//   mode 2: stores by wavefronts 4-7 only (they store twice as much), loads by wavefronts 0-3 only (twice as many)
//   mode 3: as 1, but the stores are issued 2 per k-tile inside the NEXT tile's k-loop instead of as one burst
//   mode 4: stores only
//   mode 5: as 3 with a COUNTED wait (vmcnt(2): the k-tile's two stores, issued behind its loads, may stay in flight)
//   mode 6: as 1, and the burst is drained (vmcnt(0)) before the next tile's first request
//   mode 7: as 1, and during the tile's last PF k-tiles every wavefront touches one line per lane of the NEXT tile's A rows
//           (k-tiles 0 .. PFK-1: plain loads into a dummy register) so that the requests behind the store burst are served by the L2
// A operands: rows of a [M, K] fp16 matrix read once per tile column band (4 tiles share a row block); W: a small resident panel.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_load_mix.hip -o tools/ubench/store_load_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void spin10ns(int n) {
    if (n <= 0) return;
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)n;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(1);
}

#ifndef PFK
#define PFK 4
#endif
template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const char* A, const char* W, char* out, long long lda, long long ldo, int tiles_n, int ntiles, int nk, int spin,
                                            unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    const bool loader = MODE != 2 || wv < 4, storer = MODE != 2 || wv >= 4;
    const int nload = MODE == 2 ? 18 : 9, nstore = MODE == 2 ? 40 : 20;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int pending = 0;                                  // mode 3: stores of the previous tile still to issue
    char* pbase = out;
    unsigned sink_acc = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        const char* arow = A + (size_t)tm * 256 * lda;
        for (int kt = 0; kt < nk; ++kt) {
            if (MODE != 4 && loader) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    if (i >= nload) break;
                    const int piece = (wv * nload + i) % 72;
                    const char* src = piece < 32 ? arow + (size_t)(piece * 8 + (lane >> 3)) * lda + (size_t)kt * 128 + (lane & 7) * 16
                                                 : W + (size_t)(tn % 4) * 40960 * 16 + (size_t)((piece - 32) * 1024 + lane * 16) + (size_t)(kt % 16) * 40960;
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + (kt & 1) * 73728 + piece * 1024), 16, 0, 0);
                }
            }
            if ((MODE == 3 || MODE == 5) && pending > 0 && storer) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int it = 20 - pending;
                    const int q = lane + it * 64, row = q / 20, ch = q % 20;
                    *(u32x4*)(pbase + (size_t)((wv >> 1) * 64 + row) * ldo + (wv & 1) * 320 + ch * 16) = v;
                    --pending;
                }
            }
            if (MODE == 7 && kt >= nk - 2 && t + (int)gridDim.x < ntiles) {
                // next tile's A rows: 256 rows x PFK k-tiles x 128 B = 256 * PFK lines; 16 wave-loads of 64 lines per k-tile here -> 2 k-tiles cover PFK = 8
                const int tn2 = t + (int)gridDim.x;
                const char* arow2 = A + (size_t)(tn2 / tiles_n) * 256 * lda;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int line = ((kt - (nk - 2)) * 16 + wv * 2 + i) * 64 + lane;         // 0 .. 2047
                    if (line < 256 * PFK) {
                        unsigned x;
                        asm volatile("global_load_dword %0, %1, off" : "=v"(x) : "v"(arow2 + (size_t)(line % 256) * lda + (size_t)(line / 256) * 128));
                        sink_acc += 0;      // (the value is never consumed: the request only has to reach the L2)
                    }
                }
            }
            spin10ns(spin);
            if (MODE == 5 && pending >= 0 && pending < 20) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        char* base = out + (size_t)tm * 256 * ldo + (size_t)tn * 640;
        if (MODE == 3 || MODE == 5) { pending = 20; pbase = base; }
        else if (MODE != 0 && storer) {
#pragma unroll
            for (int it = 0; it < 40; ++it) {
                if (it >= nstore) break;
                const int w2 = MODE == 2 ? (wv - 4) * 2 + it / 20 : wv, q = lane + (it % 20) * 64, row = q / 20, ch = q % 20;
                *(u32x4*)(base + (size_t)((w2 >> 1) * 64 + row) * ldo + (w2 & 1) * 320 + ch * 16) = v;
            }
        }
        if (MODE == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

template <int MODE>
void run(const char* label, const char* A, const char* W, char* out, long long lda, long long ldo, int tiles_m, int tiles_n, int nk, int spin, unsigned long long* cyc) {
    const int ntiles = tiles_m * tiles_n, G = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(G), dim3(512), 163840, 0, A, W, out, lda, ldo, tiles_n, ntiles, nk, spin, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double per_tile = best * 1e3 / ((double)ntiles / G);
    printf("%-58s nk=%2d spin=%4d : %8.1f us  %7.2f us per tile (k-loop ideal %6.2f)\n", label, nk, spin, best * 1e3, per_tile, nk * spin * 0.01);
}

int main() {
    const int tiles_m = 252, tiles_n = 16;                       // [64512, 5120] output, K = 640 (nk = 10) or 1280
    const long long ldo = 5120 * 2;
    char *A, *W, *out; unsigned long long* cyc;
    hipMalloc(&A, (size_t)64512 * 2560 * 2); hipMemset(A, 0, (size_t)64512 * 2560 * 2);
    hipMalloc(&W, (size_t)64 << 20); hipMemset(W, 0, (size_t)64 << 20);
    hipMalloc(&out, (size_t)tiles_m * 256 * ldo + 4096);
    hipMalloc(&cyc, 4096 * 8);
    for (int nk : {10, 20}) {
        const long long lda = (long long)nk * 128;
        for (int spin : {140, 0}) {
            run<0>("loads only", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<4>("stores only", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<1>("stores burst + loads, every wavefront (today)", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<2>("stores by wavefronts 4-7, loads by wavefronts 0-3", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<3>("stores 2 per k-tile inside the next tile's k-loop", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<5>("  ... with a counted wait (the 2 stores stay in flight)", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<6>("stores burst drained before the next tile's loads", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
            run<7>("today + next tile's first A k-tiles touched into the L2 ahead", A, W, out, lda, ldo, tiles_m, tiles_n, nk, spin, cyc);
        }
    }
    return 0;
}
