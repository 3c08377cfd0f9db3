// Microbenchmark (developer tool): what the output stream of a persistent 256 x 320 contraction tile costs with nothing else
// running - G blocks of 512 threads each write T tiles of 256 rows x 640 bytes (row stride LD bytes) and, between tiles, spin for
// C cycles ("main loop" without memory instructions).  Answers: (1) the aggregate store rate per pattern / cache policy /
// number of CUs storing at once, (2) whether a CU's stores drain under its own ALU work (time = max(C, drain)) or not (sum),
// (3) what a dependent load behind the stores (vmcnt retires in order and counts stores) does to that.
//   pattern 0: as lean_store on the 256 x 320 tile (wave = 64 rows x 320 B; a store instruction covers 3.2 rows of 320 B)
//   pattern 1: a wave's store instruction covers 1.6 whole 640-byte tile rows
//   pattern 2: the tile is one contiguous 160 KB run (a store instruction = 1 KiB contiguous)
//   pattern 3: as lean_store is really called on that tile: two passes of 80 columns, a store instruction covers 6.4 rows of 160 B
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_rate.hip -o tools/ubench/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT, bool DEPLOAD>
__global__ void __launch_bounds__(512, 2) k(char* out, const char* src, long long ld, int tiles_n, int ntiles, int spin, int stagger,
                                            unsigned long long* cyc, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    unsigned acc = 0;
    if (stagger > 0) {                                   // de-phase the blocks of an XCD (8 phases)
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(((blockIdx.x >> 3) & 7) * stagger);
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(2);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        char* base = PAT == 2 ? out + (size_t)t * 163840 : out + (size_t)tm * 256 * ld + (size_t)tn * 640;
#pragma unroll
        for (int it = 0; it < 20; ++it) {
            const int q = lane + it * 64;
            char* dst;
            if (PAT == 0) { const int row = q / 20, ch = q % 20; dst = base + (size_t)(wm * 64 + row) * ld + wn * 320 + ch * 16; }
            else if (PAT == 1) { const int row = q / 40, ch = q % 40; dst = base + (size_t)(wv * 32 + row) * ld + ch * 16; }
            else if (PAT == 3) { const int pass = it / 10, q2 = lane + (it % 10) * 64, row = q2 / 10, ch = q2 % 10; dst = base + (size_t)(wm * 64 + row) * ld + wn * 320 + pass * 160 + ch * 16; }
            else dst = base + wv * 20480 + q * 16;
            if (NT) __builtin_nontemporal_store(v, (u32x4*)dst); else *(u32x4*)dst = v;
        }
        if (DEPLOAD) {                                   // a load behind the stores, waited for: vmcnt(0) = every store acknowledged
            unsigned x = *(const volatile unsigned*)(src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 4);
            acc += x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (spin > 0) {
            const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)spin;
            while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT, bool NT, bool DEP>
void run(const char* label, char* out, const char* src, long long ld, int tiles_m, int tiles_n, int G, int spin, unsigned long long* cyc, unsigned* sink, int stagger = 0) {
    const int ntiles = tiles_m * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, NT, DEP>), dim3(G), dim3(512), 0, 0, out, src, ld, tiles_n, ntiles, spin, stagger, cyc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= G;
    const double bytes = (double)ntiles * 163840.0, per_tile = avg / ((double)ntiles / G);
    printf("%-44s G=%3d spin=%6d stagger=%5d : %8.1f us  %5.2f TB/s  %8.2f us per tile per block  %6.1f GB/s per CU\n", label, G, spin, stagger, best * 1e3,
           bytes / best / 1e9, per_tile / 100.0, 163840.0 / (per_tile * 10.0));
}

int main(int argc, char** argv) {
    const int tiles_m = 252, tiles_n = 16;             // [64512, 5120] fp16 = 660 MB
    const long long ld = 5120 * 2;
    char* out; hipMalloc(&out, (size_t)tiles_m * 256 * ld + 4096);
    char* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    unsigned long long* cyc; hipMalloc(&cyc, 4096 * 8);
    unsigned* sink; hipMalloc(&sink, 64);
    printf("output [64512, 5120] fp16 (660 MB), tiles of 256 rows x 640 B; times from s_memrealtime (100 MHz): spin / stagger / per-tile figures in units of 10 ns\n");
    for (int G : {256, 128, 64, 32}) {
        run<0, false, false>("pattern 0 (lean_store)", out, src, ld, tiles_m, tiles_n, G, 0, cyc, sink);
        run<1, false, false>("pattern 1 (whole tile rows)", out, src, ld, tiles_m, tiles_n, G, 0, cyc, sink);
        run<2, false, false>("pattern 2 (contiguous tile)", out, src, ld, tiles_m, tiles_n, G, 0, cyc, sink);
        run<3, false, false>("pattern 3 (two passes of 160-byte pieces)", out, src, ld, tiles_m, tiles_n, G, 0, cyc, sink);
        run<0, true, false>("pattern 0, non-temporal", out, src, ld, tiles_m, tiles_n, G, 0, cyc, sink);
        run<2, true, false>("pattern 2, non-temporal", out, src, ld, tiles_m, tiles_n, G, 0, cyc, sink);
    }
    for (int spin : {250, 500, 1000, 2000}) {
        run<0, false, false>("pattern 0 + spin (stores drain under ALU?)", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink);
        run<0, false, true>("pattern 0 + dependent load + spin", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink);
        run<0, true, false>("pattern 0 nt + spin", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink);
    }
    for (int spin : {250, 500, 1000, 2000}) {
        run<3, false, false>("pattern 3 + spin", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink);
        run<1, false, false>("pattern 1 + spin", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink);
        run<2, false, false>("pattern 2 + spin", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink);
    }
    for (int spin : {500, 1000, 2800}) {
        for (int stg : {0, spin / 8, spin / 4}) {
            run<0, false, false>("pattern 0 + spin, staggered", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink, stg);
            run<0, false, true>("pattern 0 + dependent load + spin, staggered", out, src, ld, tiles_m, tiles_n, 256, spin, cyc, sink, stg);
        }
    }
    return 0;
}
