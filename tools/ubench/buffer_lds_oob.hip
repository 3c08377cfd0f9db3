// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (gfx950 LDS-DMA through a buffer resource) write ZEROS to its LDS
// slot, or leave the slot untouched?  (developer probe for the implicit-GEMM convolutions' padding: round 6)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/buffer_lds_oob.hip -o /tmp/buffer_lds_oob && /tmp/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;

__global__ void k(const float* a, unsigned bytes, float* out) {
    extern __shared__ char smem[];
    float* s = (float*)smem;
    for (int i = threadIdx.x; i < 64 * 4; i += 64) s[i] = -7.0f;             // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, bytes, 0x00020000);
    // even lanes: in range (16 bytes at lane * 16); odd lanes: far out of range
    const unsigned off = (threadIdx.x & 1) ? 0xFFFFFF00u : threadIdx.x * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)smem, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = s[i];
}

int main() {
    const int n = 64 * 4;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
    float *a, *o;
    hipMalloc(&a, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, (unsigned)(n * 4), o);
    std::vector<float> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int zeros = 0, kept = 0, ok = 0, other = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const float v = r[l * 4 + e];
            if (l & 1) { if (v == 0.0f) ++zeros; else if (v == -7.0f) ++kept; else ++other; }
            else { if (v == h[l * 4 + e]) ++ok; else ++other; }
        }
    printf("in-range values correct: %d / 128; out-of-range slots: %d zero, %d untouched (sentinel), %d other\n", ok, zeros, kept, other);
    printf("lane 1 slot: %g %g %g %g\n", r[4], r[5], r[6], r[7]);
    return 0;
}
