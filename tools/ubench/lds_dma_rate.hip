// Microbenchmark (developer tool): what one CU's LDS sustains when LDS-DMA fills (global_load_lds_dwordx4, 1 KiB per
// wave-instruction, cache-resident source) and ds_read_b128 fragment reads run together - the two LDS streams of the
// contraction main loop - in cycles per "k-tile" of 72 DMA pieces + 224 reads per CU (8 wavefronts: 9 + 28 each).
//   modes: 0 = DMA only, 1 = reads only, 2 = both, 3 = register staging (global_load_dwordx4 + ds_write_b128) + reads,
//          4 = register staging only
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_rate.hip -o tools/ubench/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const char* src, unsigned long long* out, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned ra = lds0 + (unsigned)((wv * 32 + fr) * 128 + ((fq ^ (fr & 7)) << 4));
    const char* g = src + (size_t)(blockIdx.x % 8) * 4096 + lane * 16;        // a few KiB per XCD: L1 / L2 resident
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        char* slot = smem + (it & 1) * 73728;
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 9; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + i * 1024), (lds_void_t*)(slot + (wv * 9 + i) * 1024), 16, 0, 0);
        }
        u32x4 st[9];
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 9; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st[i]) : "v"(g + i * 1024));
        }
        if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 28; ++i) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(ra), "i"((i % 16) * 2048));
                if ((i & 3) == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                acc ^= v;
            }
        }
        if (MODE == 3 || MODE == 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 9; ++i)
                asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + (unsigned)((it & 1) * 73728 + (wv * 9 + i) * 1024 + lane * 16)), "v"(st[i]) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}

template <int MODE>
void run(const char* name, const char* src, unsigned long long* out, unsigned* sink) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 163840, 0, src, out, iters, sink);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 163840, 0, src, out, iters, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-44s %8.0f cycles per k-tile (72 KiB filled, 224 KiB read per CU)\n", name, s / blocks / iters);
}

int main() {
    char* src; unsigned long long* out; unsigned* sink;
    hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20);
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64);
    run<0>("LDS-DMA fill only (72 pieces)", src, out, sink);
    run<1>("ds_read_b128 only (224 reads)", src, out, sink);
    run<2>("LDS-DMA fill + reads", src, out, sink);
    run<4>("register staging only (load + ds_write_b128)", src, out, sink);
    run<3>("register staging + reads", src, out, sink);
    return 0;
}
