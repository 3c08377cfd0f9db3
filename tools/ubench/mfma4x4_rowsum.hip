#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(float4v* o) {
    int l = threadIdx.x;
    half4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    half4 b = {(_Float16)(l), (_Float16)(0.25f), (_Float16)(0.5f), (_Float16)(1000.f)};
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, b, c, 0, 0, 0);
    o[l] = c;
}
int main() {
    float4v* d; hipMalloc(&d, 64 * 16);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (h[4 * l + i] != l + 1000.75f) ++bad;
    printf("lane-own-sum in every register: %s (bad=%d)  lane5: %g %g %g %g\n", bad ? "NO" : "YES", bad, h[20], h[21], h[22], h[23]);
    return 0;
}
