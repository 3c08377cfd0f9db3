// Trainer-loop pieces adjacent to the rasteriser: photometric L1 loss with its gradient,
// and the Adam update of the Gaussian parameters.
//
// The reference runs these inside FSGS' gsTrainer.training()/finetune() (call sites model/diffusionGS.py:139,
// 1640; un-vendored) as chains of torch elementwise kernels: the published 3DGS step is
// `Ll1 = |render - gt|.mean()` followed by torch.optim.Adam(eps=1e-15).  Here each is one pass over memory:
//   L1 forward   reads image + target once, deterministic two-level sum (no float atomics);
//   L1 backward  reads image + target once, writes w * go / n * sign(image - target) (go read on the device);
//   Adam         one kernel per parameter tensor, torch.optim.Adam's operation order (lerp / addcmul / addcdiv)
//                so the result matches the torch optimiser to rounding.
// HBM-bound: 8 B/element (forward), 12 B/element (backward), 28 B/element (Adam).
#include "common.h"

using namespace syn3r;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

__global__ void __launch_bounds__(kThreads) k_l1_partial(const float* __restrict__ a, const float* __restrict__ b,
                                                        long long n, float* __restrict__ partial) {
    float s = 0.0f;
    const long long n4 = n >> 2;
    const float4* a4 = (const float4*)a;
    const float4* b4 = (const float4*)b;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        float4 x = a4[i], y = b4[i];
        s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
    }
    if (blockIdx.x == 0) {
        long long i = (n4 << 2) + threadIdx.x;
        if (i < n) s += fabsf(a[i] - b[i]);
    }
    s = wave_sum(s);
    __shared__ float ws[kThreads / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// one block: fixed-order sum of the block partials (bitwise reproducible run to run)
__global__ void __launch_bounds__(kThreads) k_l1_final(const float* __restrict__ partial, int nblocks, float scale,
                                                      float* __restrict__ loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kThreads) s += (double)partial[i];
    s = wave_sum_d(s);
    __shared__ double ws[kThreads / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)(((ws[0] + ws[1]) + (ws[2] + ws[3])) * (double)scale);
}

__global__ void __launch_bounds__(kThreads) k_l1_grad(const float* __restrict__ a, const float* __restrict__ b,
                                                     long long n, float scale, const float* __restrict__ go,
                                                     float* __restrict__ grad) {
    const float g = scale * (go ? *go : 1.0f);
    auto sgn = [g](float d) { return d > 0.0f ? g : (d < 0.0f ? -g : 0.0f); };   // torch.sign: 0 at 0
    const long long n4 = n >> 2;
    const float4* a4 = (const float4*)a;
    const float4* b4 = (const float4*)b;
    float4* g4 = (float4*)grad;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        float4 x = a4[i], y = b4[i];
        g4[i] = make_float4(sgn(x.x - y.x), sgn(x.y - y.y), sgn(x.z - y.z), sgn(x.w - y.w));
    }
    if (blockIdx.x == 0) {
        long long i = (n4 << 2) + threadIdx.x;
        if (i < n) grad[i] = sgn(a[i] - b[i]);
    }
}

// torch.optim.Adam (_single_tensor_adam / _multi_tensor_adam, no weight decay, no amsgrad, not maximize):
//   exp_avg.lerp_(grad, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;  param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
__global__ void __launch_bounds__(kThreads) k_adam(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, long long n,
                                                  float w1 /*1-beta1*/, float beta2, float w2 /*1-beta2*/,
                                                  float bc2_sqrt, float eps, float neg_step) {
    long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float mi = m[i], vi = v[i];
    mi = mi + w1 * (gi - mi);                    // lerp with weight < 0.5
    vi = vi * beta2 + (w2 * gi) * gi;            // addcmul: value * t1 * t2
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] + neg_step * (mi / denom);       // addcdiv: value * (t1 / t2)
    m[i] = mi;
    v[i] = vi;
}

int l1_blocks(long long n) {
    long long b = (n / 4 + kThreads * 8 - 1) / (kThreads * 8);
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

}  // namespace

extern "C" size_t syn3r_l1_loss_workspace_bytes(long long n) { return n > 0 ? (size_t)kMaxBlocks * sizeof(float) : 0; }

extern "C" int syn3r_l1_loss(const float* image, const float* target, long long n, float weight, float* loss,
                             void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SYN3R_REQUIRE(n > 0, "l1_loss: n must be positive");
    SYN3R_REQUIRE(image && target && loss && ws, "l1_loss: null pointer");
    SYN3R_REQUIRE(ws_bytes >= syn3r_l1_loss_workspace_bytes(n), "l1_loss: workspace too small");
    SYN3R_REQUIRE((((uintptr_t)image | (uintptr_t)target) & 15) == 0, "l1_loss: image/target must be 16-byte aligned");
    const int nb = l1_blocks(n);
    SYN3R_LAUNCH(k_l1_partial, dim3(nb), dim3(kThreads), 0, stream, image, target, n, (float*)ws);
    SYN3R_LAUNCH(k_l1_final, dim3(1), dim3(kThreads), 0, stream, (const float*)ws, nb, weight / (float)n, loss);
    SYN3R_LAUNCH_CHECK("l1_loss launch");
    return SYN3R_OK;
}

extern "C" int syn3r_l1_loss_backward(const float* image, const float* target, long long n, float weight,
                                      const float* grad_loss, float* grad_image, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SYN3R_REQUIRE(n > 0, "l1_loss_backward: n must be positive");
    SYN3R_REQUIRE(image && target && grad_image, "l1_loss_backward: null pointer");
    SYN3R_REQUIRE((((uintptr_t)image | (uintptr_t)target | (uintptr_t)grad_image) & 15) == 0,
                  "l1_loss_backward: buffers must be 16-byte aligned");
    long long b = (n / 4 + kThreads * 4 - 1) / (kThreads * 4);
    if (b < 1) b = 1;
    if (b > 4 * kMaxBlocks) b = 4 * kMaxBlocks;
    SYN3R_LAUNCH(k_l1_grad, dim3((unsigned)b), dim3(kThreads), 0, stream, image, target, n, weight / (float)n,
                 grad_loss, grad_image);
    SYN3R_LAUNCH_CHECK("l1_loss_backward launch");
    return SYN3R_OK;
}

extern "C" int syn3r_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                               float lr, float beta1, float beta2, float eps, int step, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SYN3R_REQUIRE(n > 0, "adam_step: n must be positive");
    SYN3R_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
    SYN3R_REQUIRE(step >= 1, "adam_step: step is 1-based");
    SYN3R_REQUIRE(beta1 >= 0.5f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "adam_step: betas out of range");
    // the scalar factors are computed as torch does, in double on the host
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float neg_step = (float)(-((double)lr / bc1));
    const float bc2_sqrt = (float)sqrt(bc2);
    long long b = (n + kThreads - 1) / kThreads;
    SYN3R_REQUIRE(b < (1ll << 31), "adam_step: tensor too large");
    SYN3R_LAUNCH(k_adam, dim3((unsigned)b), dim3(kThreads), 0, stream, param, grad, exp_avg, exp_avg_sq, n,
                 (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2), bc2_sqrt, eps, neg_step);
    SYN3R_LAUNCH_CHECK("adam_step launch");
    return SYN3R_OK;
}
