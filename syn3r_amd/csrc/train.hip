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

// SQ = false: sum |a - b| (L1); SQ = true: sum (a - b)^2 (the MSE behind the PSNR of the metric record)
template <bool SQ>
__device__ __forceinline__ float dist_term(float d) { return SQ ? d * d : fabsf(d); }

template <bool SQ>
__global__ void __launch_bounds__(kThreads) k_l1_partial(const float* __restrict__ a, const float* __restrict__ b,
                                                        long long n, float* __restrict__ partial) {
    float s = 0.0f;
    const long long n4 = n >> 2;
    const float4* a4 = (const float4*)a;
    const float4* b4 = (const float4*)b;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        float4 x = a4[i], y = b4[i];
        s += (dist_term<SQ>(x.x - y.x) + dist_term<SQ>(x.y - y.y)) + (dist_term<SQ>(x.z - y.z) + dist_term<SQ>(x.w - y.w));
    }
    if (blockIdx.x == 0) {
        long long i = (n4 << 2) + threadIdx.x;
        if (i < n) s += dist_term<SQ>(a[i] - b[i]);
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

// The same update for up to kAdamMulti tensors in ONE launch (round 6): a descriptor table travels as the kernel
// argument, a block finds its tensor by the table's block offsets.  Element for element the arithmetic of k_adam (bit-identical);
// what it saves is the launches (the trainer's five / six parameter groups: 60 us of HBM-bound work in 5 launches - the bytes
// stay, 330 MB per step at 200 000 Gaussians).
constexpr int kAdamMulti = 8;
struct AdamTable {
    float* p[kAdamMulti]; const float* g[kAdamMulti]; float* m[kAdamMulti]; float* v[kAdamMulti];
    long long n[kAdamMulti];
    float neg_step[kAdamMulti], eps[kAdamMulti], bc2_sqrt[kAdamMulti];
    unsigned blk0[kAdamMulti + 1];
    int count;
    float w1, beta2, w2;
};
__global__ void __launch_bounds__(kThreads) k_adam_multi(AdamTable t) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < kAdamMulti; ++q) k += (q < t.count && blockIdx.x >= t.blk0[q]) ? 1 : 0;
    const long long i = (long long)(blockIdx.x - t.blk0[k]) * kThreads + threadIdx.x;
    if (i >= t.n[k]) return;
    float* __restrict__ p = t.p[k]; const float* __restrict__ g = t.g[k]; float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k];
    const float gi = g[i];
    float mi = m[i], vi = v[i];
    mi = mi + t.w1 * (gi - mi);
    vi = vi * t.beta2 + (t.w2 * gi) * gi;
    const float denom = sqrtf(vi) / t.bc2_sqrt[k] + t.eps[k];
    p[i] = p[i] + t.neg_step[k] * (mi / denom);
    m[i] = mi;
    v[i] = vi;
}

// ---------------------------------------------------------------------------------------------
// The published 3DGS parameter activations (GaussianModel.get_scaling / get_rotation / get_opacity: exp, normalize, sigmoid;
// FSGS' trainer behind gsTrainer.training() / finetune(), model/diffusionGS.py:139,1640) as ONE launch forward and ONE launch
// for the chain rule backward, instead of three torch operators + their ~10 autograd kernels per training iteration.
// torch's formulas: normalize = x / max(|x|_2, 1e-12); d/dx = (g - xhat (xhat . g)) / max(|x|, eps);
// sigmoid' = s (1 - s); exp' = e.
__global__ void __launch_bounds__(kThreads) k_activate(int N, const float* __restrict__ log_scale, const float* __restrict__ rot,
                                                       const float* __restrict__ logit, float* __restrict__ scale,
                                                       float* __restrict__ rot_n, float* __restrict__ opacity) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) scale[3 * i + c] = act_exp(log_scale[3 * i + c]);
    const float4 q = *(const float4*)(rot + 4 * i);
    *(float4*)(rot_n + 4 * i) = act_quat(q, act_quat_inv_norm(q));
    opacity[i] = act_sigmoid(logit[i]);
}

__global__ void __launch_bounds__(kThreads) k_activate_bwd(int N, const float* __restrict__ rot, const float* __restrict__ scale,
                                                           const float* __restrict__ rot_n, const float* __restrict__ opacity,
                                                           const float* __restrict__ d_scale, const float* __restrict__ d_rot_n,
                                                           const float* __restrict__ d_opacity, float* __restrict__ d_log_scale,
                                                           float* __restrict__ d_rot, float* __restrict__ d_logit) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_log_scale[3 * i + c] = d_scale[3 * i + c] * scale[3 * i + c];
    const float4 q = *(const float4*)(rot + 4 * i), h = *(const float4*)(rot_n + 4 * i), g = *(const float4*)(d_rot_n + 4 * i);
    *(float4*)(d_rot + 4 * i) = act_quat_bwd(h, g, act_quat_inv_norm(q));
    d_logit[i] = act_sigmoid_bwd(opacity[i], d_opacity[i]);
}

// GaussianModel.add_densification_stats of the published trainer for the visible Gaussians (radii > 0): torch.norm(grad[:, :2]) is
// sqrt(gx*gx + gy*gy) in fp32 (this file compiles without fma contraction)
__global__ void __launch_bounds__(kThreads) k_densify_stats(int N, const int* __restrict__ radii, const float* __restrict__ vgrad,
                                                           float* __restrict__ accum, float* __restrict__ denom,
                                                           float* __restrict__ max_radii) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = vgrad[3 * i], gy = vgrad[3 * i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
    max_radii[i] = fmaxf(max_radii[i], (float)r);
}

// ---------------------------------------------------------------------------------------------
// Photometric loss of the published 3DGS trainer, fused:  L = w * [(1 - lambda) * mean|I - G| + lambda * (1 - SSIM(I, G))]
// with SSIM as published (11x11 Gaussian window, sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over
// all channels and pixels).  Forward: one pass over 32x32 tiles, separable window (horizontal then vertical) on the five
// moments; it also stores the three per-pixel derivative maps (d ssim / d mu1 [total], / d sigma1^2, / d sigma12) the
// backward needs.  Backward: the same separable window on those three maps, combined with the L1 sign term, one write of
// the image gradient.
// Round 5 form (the 16x16-tile, one-output-per-thread kernels ran 96 + 67 us at 1080p, half of it vector issue): a thread
// owns FOUR adjacent outputs in each pass - a row run in the horizontal pass, whose 14 inputs come straight from global
// memory as aligned 16-byte loads (no staging tile, one barrier less), a column run in the vertical pass - so the inputs'
// squares and products are formed once per input instead of once per tap, and the moments travel in pairs on packed fp32
// FMAs: (mu1, mu2) and (E[p^2], E[q^2]) are one v_pk_fma_f32 per tap each, E[pq] a v_fma_f32 (3 instructions per tap and
// output where the unpacked, uncontracted form issued 10).  The taps of an output are accumulated in ascending order.
constexpr int kWin = 11, kHalo = 5, kTile = 32, kReg = kTile + 2 * kHalo;   // 42
constexpr int kRun = 4, kSpan = kRun + kWin - 1;   // 4 outputs per thread and pass, 14 inputs
constexpr int kLead = 8 - kHalo;                   // a row run is loaded from 8 columns left of its first output: 16-byte aligned
constexpr int kPhotoItems = kReg * (kTile / kRun);   // 336 horizontal items per tile
constexpr int kPhotoThreads = 256;                 // the vertical pass' 32 columns x 8 row runs (384 threads = one round of horizontal items: +17 % forward)
struct SsimWindow { float g[kWin]; };
// Tile of block b: blocks are dealt to the 8 XCDs round-robin, so XCD x takes the x-th contiguous eighth of the row-major tile list -
// neighbouring tiles (which share 10 of 42 halo rows and 16 of 48 loaded columns) meet in one L2 instead of being fetched by two.
struct PhotoTile { int c, x0, y0; unsigned id; };
__device__ __forceinline__ PhotoTile photo_tile(int gx, int gy) {
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    PhotoTile o;
    o.id = t;
    o.c = (int)(t / (unsigned)(gx * gy));
    const unsigned rem = t - (unsigned)o.c * (unsigned)(gx * gy);
    o.y0 = (int)(rem / (unsigned)gx) * kTile;
    o.x0 = (int)(rem % (unsigned)gx) * kTile;
    return o;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ float block_sum_photo(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

// 20 consecutive values of image row y from column x (x % 4 == 0; zero outside the image).  VEC: W % 4 == 0 and the plane is
// 16-byte aligned, so a group of four is inside the image or outside it as a whole.  Every load is UNCONDITIONAL from an address
// clamped into the plane and the zero is a select afterwards: a load behind its own `inside ? load : 0` branch waits for the one
// before it (hipcc does not move loads across the exec-mask branches: the ten loads of a row run were a chain of ten latencies).
template <bool VEC>
__device__ __forceinline__ void photo_load20(float (&v)[20], const float* __restrict__ plane, int y, int x, int H, int W) {
    const bool row = y >= 0 && y < H;
    const float* src = plane + (size_t)min(max(y, 0), H - 1) * W;
    if constexpr (VEC) {
        float4 t[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) t[q] = *(const float4*)(src + min(max(x + 4 * q, 0), W - 4));
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const bool in = row && x + 4 * q >= 0 && x + 4 * q < W;
            v[4 * q] = in ? t[q].x : 0.0f; v[4 * q + 1] = in ? t[q].y : 0.0f; v[4 * q + 2] = in ? t[q].z : 0.0f; v[4 * q + 3] = in ? t[q].w : 0.0f;
        }
    } else {
        float t[20];
#pragma unroll
        for (int e = 0; e < 20; ++e) t[e] = src[min(max(x + e, 0), W - 1)];
#pragma unroll
        for (int e = 0; e < 20; ++e) v[e] = (row && x + e >= 0 && x + e < W) ? t[e] : 0.0f;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(kPhotoThreads) k_photo_fwd(const float* __restrict__ img, const float* __restrict__ gt, int H,
                                                   int W, int C, SsimWindow win, float* __restrict__ maps,
                                                   float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) f2 hA[kReg][kTile], hB[kReg][kTile];   // (mu1, mu2), (E[p^2], E[q^2]) after the horizontal pass
    __shared__ __attribute__((aligned(16))) float hC[kReg][kTile];                 // E[pq]
    __shared__ float red[kPhotoThreads / 64];
    const PhotoTile tl_ = photo_tile((W + kTile - 1) / kTile, (H + kTile - 1) / kTile);
    const int c = tl_.c, x0 = tl_.x0, y0 = tl_.y0;
    const size_t plane = (size_t)H * W;
    const float* a = img + c * plane;
    const float* b = gt + c * plane;
    float l1_sum = 0.f;
    // horizontal pass: item = (halo row, run of 4 columns)
    for (int i = threadIdx.x; i < kPhotoItems; i += kPhotoThreads) {
        const int ry = i / (kTile / kRun), tx = (i - ry * (kTile / kRun)) * kRun;
        const int y = y0 + ry - kHalo;
        float p[20], q[20];
        photo_load20<VEC>(p, a, y, x0 + tx - 8, H, W);
        photo_load20<VEC>(q, b, y, x0 + tx - 8, H, W);
        f2 pq[kSpan], sq[kSpan];
        float pr[kSpan];
#pragma unroll
        for (int t = 0; t < kSpan; ++t) {
            pq[t] = (f2){p[kLead + t], q[kLead + t]};
            sq[t] = pq[t] * pq[t];
            pr[t] = pq[t].x * pq[t].y;
        }
        f2 mA[kRun], mB[kRun];
        float mC[kRun];
#pragma unroll
        for (int j = 0; j < kRun; ++j) { mA[j] = (f2){0.f, 0.f}; mB[j] = (f2){0.f, 0.f}; mC[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const f2 w2 = (f2){win.g[k], win.g[k]};
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                mA[j] = fma2(w2, pq[j + k], mA[j]);
                mB[j] = fma2(w2, sq[j + k], mB[j]);
                mC[j] = __builtin_fmaf(win.g[k], pr[j + k], mC[j]);
            }
        }
        *(float4*)&hA[ry][tx] = make_float4(mA[0].x, mA[0].y, mA[1].x, mA[1].y);
        *(float4*)&hA[ry][tx + 2] = make_float4(mA[2].x, mA[2].y, mA[3].x, mA[3].y);
        *(float4*)&hB[ry][tx] = make_float4(mB[0].x, mB[0].y, mB[1].x, mB[1].y);
        *(float4*)&hB[ry][tx + 2] = make_float4(mB[2].x, mB[2].y, mB[3].x, mB[3].y);
        *(float4*)&hC[ry][tx] = make_float4(mC[0], mC[1], mC[2], mC[3]);
        if (ry >= kHalo && ry < kHalo + kTile && y < H) {      // an output row: its L1 terms (the run's own pixels are inputs kHalo .. kHalo + 3)
#pragma unroll
            for (int j = 0; j < kRun; ++j)
                if (x0 + tx + j < W) l1_sum += fabsf(pq[kHalo + j].x - pq[kHalo + j].y);
        }
    }
    __syncthreads();
    // vertical pass: thread = (column, run of 4 rows)
    const int tx = threadIdx.x & (kTile - 1), ty = (threadIdx.x >> 5) * kRun;
    const int x = x0 + tx;
    f2 mA[kRun], mB[kRun];
    float mC[kRun];
    {
        f2 vA[kSpan], vB[kSpan];
        float vC[kSpan];
#pragma unroll
        for (int t = 0; t < kSpan; ++t) { vA[t] = hA[ty + t][tx]; vB[t] = hB[ty + t][tx]; vC[t] = hC[ty + t][tx]; }
#pragma unroll
        for (int j = 0; j < kRun; ++j) { mA[j] = (f2){0.f, 0.f}; mB[j] = (f2){0.f, 0.f}; mC[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const f2 w2 = (f2){win.g[k], win.g[k]};
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                mA[j] = fma2(w2, vA[j + k], mA[j]);
                mB[j] = fma2(w2, vB[j + k], mB[j]);
                mC[j] = __builtin_fmaf(win.g[k], vC[j + k], mC[j]);
            }
        }
    }
    float ssim_sum = 0.f;
    const size_t n = (size_t)C * plane;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
        const int y = y0 + ty + j;
        if (x < W && y < H) {
            const float mu1 = mA[j].x, mu2 = mA[j].y, e11 = mB[j].x, e22 = mB[j].y, e12 = mC[j];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float sg1 = e11 - mu1s, sg2 = e22 - mu2s, sg12 = e12 - mu12;
            const float A = 2.f * mu12 + C1, B = 2.f * sg12 + C2, Cc = mu1s + mu2s + C1, D = sg1 + sg2 + C2;
            // Cc >= C1, D >= C2 - rounding: both positive and far from the denormals; v_rcp_f32 is a 1-ulp reciprocal
            // (an IEEE division is ~10 instructions, three of them were a third of this pass)
            const float invC = __builtin_amdgcn_rcpf(Cc), invD = __builtin_amdgcn_rcpf(D);
            const float inv = invC * invD;
            const float ssim = A * B * inv;
            // partial derivatives of the map value (gt is constant)
            const float d_sg1 = -ssim * invD;              // d/d sigma1^2
            const float d_sg12 = 2.f * A * inv;            // d/d sigma12
            const float d_mu1 = (2.f * mu2 * B * Cc - 2.f * mu1 * A * B) * inv * invC    // explicit
                                - 2.f * mu1 * d_sg1 - mu2 * d_sg12;                      // through sigma1^2, sigma12
            const size_t o = (size_t)c * plane + (size_t)y * W + x;
            maps[o] = d_mu1; maps[n + o] = d_sg1; maps[2 * n + o] = d_sg12;
            ssim_sum += ssim;
        }
    }
    const float ts = block_sum_photo(ssim_sum, red);
    const float tl = block_sum_photo(l1_sum, red);
    if (threadIdx.x == 0) {
        const size_t bid = tl_.id;
        partial[2 * bid] = tl;
        partial[2 * bid + 1] = ts;
    }
}

// loss[0] = w*((1-lam)*L1 + lam*(1-SSIM)), loss[1] = L1, loss[2] = SSIM  (fixed-order double sums; a block of 256 threads)
struct PhotoFinal { const float* partial; long long nblocks; double inv_n; float lam, weight; float* loss; };
__device__ __forceinline__ void photo_final(const PhotoFinal f) {
    static_assert(kPhotoThreads == 256, "photo_final: one sum per thread of a 256-thread block");
    double sl = 0.0, ss = 0.0;
    for (long long i = threadIdx.x; i < f.nblocks; i += 256) { sl += (double)f.partial[2 * i]; ss += (double)f.partial[2 * i + 1]; }
    sl = wave_sum_d(sl); ss = wave_sum_d(ss);
    __shared__ double w1[4], w2[4];
    if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = sl; w2[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double l1 = ((w1[0] + w1[1]) + (w1[2] + w1[3])) * f.inv_n, ssim = ((w2[0] + w2[1]) + (w2[2] + w2[3])) * f.inv_n;
        f.loss[0] = (float)((double)f.weight * ((1.0 - (double)f.lam) * l1 + (double)f.lam * (1.0 - ssim)));
        f.loss[1] = (float)l1;
        f.loss[2] = (float)ssim;
    }
}
__global__ void __launch_bounds__(256) k_photo_final(PhotoFinal f) { photo_final(f); }

template <bool VEC>
__global__ void __launch_bounds__(kPhotoThreads) k_photo_bwd(const float* __restrict__ img, const float* __restrict__ gt, int H,
                                                   int W, int C, SsimWindow win, const float* __restrict__ maps, float c_l1,
                                                   float c_ssim, const float* __restrict__ go,
                                                   float* __restrict__ grad, PhotoFinal fin) {
    __shared__ __attribute__((aligned(16))) f2 hA[kReg][kTile];      // windowed (d_mu1, d_sigma1^2) after the horizontal pass
    // syn3r_photo_loss_step: the forward's per-tile sums become loss3 HERE (k_photo_final's fixed-order double sums, by the block
    // that is dispatched first) instead of in a single-block launch between the two passes: the gradient does not read the loss
    if (fin.partial && blockIdx.x == 0) photo_final(fin);
    __shared__ __attribute__((aligned(16))) float hC[kReg][kTile];   // windowed d_sigma12
    const PhotoTile tl_ = photo_tile((W + kTile - 1) / kTile, (H + kTile - 1) / kTile);
    const int c = tl_.c, x0 = tl_.x0, y0 = tl_.y0;
    const size_t plane = (size_t)H * W, n = (size_t)C * plane;
    const float* m0 = maps + c * plane;
    // the vertical pass' own pixels (column tx, rows ty .. ty + 3), requested before the horizontal pass
    const int vtx = threadIdx.x & (kTile - 1), vty = (threadIdx.x >> 5) * kRun;
    float pv[kRun], qv[kRun];
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
        const size_t o = (size_t)c * plane + (size_t)min(y0 + vty + j, H - 1) * W + min(x0 + vtx, W - 1);     // (clamped: rows / columns past the image are not written)
        pv[j] = img[o];
        qv[j] = gt[o];
    }
    for (int i = threadIdx.x; i < kPhotoItems; i += kPhotoThreads) {
        const int ry = i / (kTile / kRun), tx = (i - ry * (kTile / kRun)) * kRun;
        const int y = y0 + ry - kHalo;
        float u0[20], u1[20], u2[20];
        photo_load20<VEC>(u0, m0, y, x0 + tx - 8, H, W);
        photo_load20<VEC>(u1, m0 + n, y, x0 + tx - 8, H, W);
        photo_load20<VEC>(u2, m0 + 2 * n, y, x0 + tx - 8, H, W);
        f2 mA[kRun];
        float mC[kRun];
#pragma unroll
        for (int j = 0; j < kRun; ++j) { mA[j] = (f2){0.f, 0.f}; mC[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const f2 w2 = (f2){win.g[k], win.g[k]};
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                mA[j] = fma2(w2, (f2){u0[kLead + j + k], u1[kLead + j + k]}, mA[j]);
                mC[j] = __builtin_fmaf(win.g[k], u2[kLead + j + k], mC[j]);
            }
        }
        *(float4*)&hA[ry][tx] = make_float4(mA[0].x, mA[0].y, mA[1].x, mA[1].y);
        *(float4*)&hA[ry][tx + 2] = make_float4(mA[2].x, mA[2].y, mA[3].x, mA[3].y);
        *(float4*)&hC[ry][tx] = make_float4(mC[0], mC[1], mC[2], mC[3]);
    }
    __syncthreads();
    const int tx = vtx, ty = vty;
    const int x = x0 + tx;
    f2 gA[kRun];
    float gC[kRun];
    {
        f2 vA[kSpan];
        float vC[kSpan];
#pragma unroll
        for (int t = 0; t < kSpan; ++t) { vA[t] = hA[ty + t][tx]; vC[t] = hC[ty + t][tx]; }
#pragma unroll
        for (int j = 0; j < kRun; ++j) { gA[j] = (f2){0.f, 0.f}; gC[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const f2 w2 = (f2){win.g[k], win.g[k]};
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                gA[j] = fma2(w2, vA[j + k], gA[j]);
                gC[j] = __builtin_fmaf(win.g[k], vC[j + k], gC[j]);
            }
        }
    }
    const float up = go ? *go : 1.0f;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
        const int y = y0 + ty + j;
        if (x >= W || y >= H) continue;
        const size_t o = (size_t)c * plane + (size_t)y * W + x;
        const float p = pv[j], q = qv[j], d = p - q;
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        grad[o] = up * (c_l1 * sgn - c_ssim * (gA[j].x + 2.0f * p * gA[j].y + q * gC[j]));
    }
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
    SYN3R_LAUNCH(k_l1_partial<false>, dim3(nb), dim3(kThreads), 0, stream, image, target, n, (float*)ws);
    SYN3R_LAUNCH(k_l1_final, dim3(1), dim3(kThreads), 0, stream, (const float*)ws, nb, weight / (float)n, loss);
    SYN3R_LAUNCH_CHECK("l1_loss launch");
    return SYN3R_OK;
}

extern "C" int syn3r_image_mse(const float* image, const float* target, long long n, float* mse, void* ws,
                               size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SYN3R_REQUIRE(n > 0, "image_mse: n must be positive");
    SYN3R_REQUIRE(image && target && mse && ws, "image_mse: null pointer");
    SYN3R_REQUIRE(ws_bytes >= syn3r_l1_loss_workspace_bytes(n), "image_mse: workspace too small");
    SYN3R_REQUIRE((((uintptr_t)image | (uintptr_t)target) & 15) == 0, "image_mse: image/target must be 16-byte aligned");
    const int nb = l1_blocks(n);
    SYN3R_LAUNCH(k_l1_partial<true>, dim3(nb), dim3(kThreads), 0, stream, image, target, n, (float*)ws);
    SYN3R_LAUNCH(k_l1_final, dim3(1), dim3(kThreads), 0, stream, (const float*)ws, nb, 1.0f / (float)n, mse);
    SYN3R_LAUNCH_CHECK("image_mse launch");
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

extern "C" int syn3r_adam_step_multi(int count, float* const* params, const float* const* grads, float* const* exp_avgs,
                                     float* const* exp_avg_sqs, const long long* numels, const float* lrs, float beta1, float beta2,
                                     const float* epss, const int* steps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SYN3R_REQUIRE(count >= 1 && count <= kAdamMulti, "adam_step_multi: count=%d must be 1..%d", count, kAdamMulti);
    SYN3R_REQUIRE(params && grads && exp_avgs && exp_avg_sqs && numels && lrs && epss && steps, "adam_step_multi: null table");
    SYN3R_REQUIRE(beta1 >= 0.5f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "adam_step_multi: betas out of range");
    AdamTable t{};
    t.count = count;
    t.w1 = (float)(1.0 - (double)beta1); t.beta2 = beta2; t.w2 = (float)(1.0 - (double)beta2);
    long long blocks = 0;
    for (int k = 0; k < count; ++k) {
        SYN3R_REQUIRE(params[k] && grads[k] && exp_avgs[k] && exp_avg_sqs[k], "adam_step_multi: null pointer in tensor %d", k);
        SYN3R_REQUIRE(numels[k] > 0 && steps[k] >= 1, "adam_step_multi: tensor %d: n must be positive, step 1-based", k);
        // the scalar factors are computed as torch does, in double on the host (syn3r_adam_step)
        const double bc1 = 1.0 - pow((double)beta1, (double)steps[k]);
        const double bc2 = 1.0 - pow((double)beta2, (double)steps[k]);
        t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avgs[k]; t.v[k] = exp_avg_sqs[k]; t.n[k] = numels[k];
        t.neg_step[k] = (float)(-((double)lrs[k] / bc1)); t.bc2_sqrt[k] = (float)sqrt(bc2); t.eps[k] = epss[k];
        t.blk0[k] = (unsigned)blocks;
        blocks += (numels[k] + kThreads - 1) / kThreads;
        SYN3R_REQUIRE(blocks < (1ll << 31), "adam_step_multi: tensors too large");
    }
    for (int k = count; k <= kAdamMulti; ++k) t.blk0[k] = (unsigned)blocks;
    SYN3R_LAUNCH(k_adam_multi, dim3((unsigned)blocks), dim3(kThreads), 0, stream, t);
    SYN3R_LAUNCH_CHECK("adam_step_multi launch");
    return SYN3R_OK;
}

extern "C" int syn3r_gaussian_activate(int N, const float* log_scales, const float* rotations, const float* opacity_logits,
                                       float* scales, float* rotations_n, float* opacities, void* stream_) {
    SYN3R_REQUIRE(SYN3R_DIM_OK(N), "gaussian_activate: bad N=%d", N);
    SYN3R_REQUIRE(log_scales && rotations && opacity_logits && scales && rotations_n && opacities, "gaussian_activate: null pointer");
    SYN3R_REQUIRE((((uintptr_t)rotations | (uintptr_t)rotations_n) % 16) == 0, "gaussian_activate: quaternions must be 16-byte aligned");
    SYN3R_LAUNCH(k_activate, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream_, N, log_scales,
                 rotations, opacity_logits, scales, rotations_n, opacities);
    SYN3R_LAUNCH_CHECK("gaussian_activate launch");
    return SYN3R_OK;
}

extern "C" int syn3r_gaussian_activate_backward(int N, const float* rotations, const float* scales, const float* rotations_n,
                                                const float* opacities, const float* d_scales, const float* d_rotations_n,
                                                const float* d_opacities, float* d_log_scales, float* d_rotations,
                                                float* d_opacity_logits, void* stream_) {
    SYN3R_REQUIRE(SYN3R_DIM_OK(N), "gaussian_activate_backward: bad N=%d", N);
    SYN3R_REQUIRE(rotations && scales && rotations_n && opacities && d_scales && d_rotations_n && d_opacities && d_log_scales &&
                  d_rotations && d_opacity_logits, "gaussian_activate_backward: null pointer");
    SYN3R_REQUIRE((((uintptr_t)rotations | (uintptr_t)rotations_n | (uintptr_t)d_rotations_n | (uintptr_t)d_rotations) % 16) == 0,
                  "gaussian_activate_backward: quaternion tensors must be 16-byte aligned");
    SYN3R_LAUNCH(k_activate_bwd, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream_, N, rotations,
                 scales, rotations_n, opacities, d_scales, d_rotations_n, d_opacities, d_log_scales, d_rotations, d_opacity_logits);
    SYN3R_LAUNCH_CHECK("gaussian_activate_backward launch");
    return SYN3R_OK;
}

extern "C" int syn3r_densification_stats(int N, const int* radii, const float* viewspace_grad, float* grad_accum, float* denom,
                                         float* max_radii, void* stream_) {
    SYN3R_REQUIRE(SYN3R_DIM_OK(N), "densification_stats: bad N=%d", N);
    SYN3R_REQUIRE(radii && viewspace_grad && grad_accum && denom && max_radii, "densification_stats: null pointer");
    SYN3R_LAUNCH(k_densify_stats, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream_, N, radii,
                 viewspace_grad, grad_accum, denom, max_radii);
    SYN3R_LAUNCH_CHECK("densification_stats launch");
    return SYN3R_OK;
}

static SsimWindow make_window() {
    SsimWindow w;
    double g[kWin], sum = 0.0;
    for (int i = 0; i < kWin; ++i) { g[i] = exp(-(double)((i - kWin / 2) * (i - kWin / 2)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < kWin; ++i) w.g[i] = (float)(g[i] / sum);
    return w;
}

extern "C" size_t syn3r_photo_loss_workspace_bytes(int C, int H, int W) {
    if (C <= 0 || C > 65535 || !SYN3R_SIDE_OK(H) || !SYN3R_SIDE_OK(W)) return 0;
    const size_t n = (size_t)C * H * W;
    const size_t blocks = (size_t)C * ((H + kTile - 1) / kTile) * ((W + kTile - 1) / kTile);
    return 3 * n * sizeof(float) + ((2 * blocks * sizeof(float) + 255) / 256) * 256;
}

// forward pass (k_photo_fwd) of both entries; `final_launch`: loss3 by k_photo_final (syn3r_photo_loss) or left to the backward
static int photo_forward(const char* who, const float* image, const float* target, int C, int H, int W, float lambda_dssim,
                         float weight, float* loss3, void* ws, size_t ws_bytes, hipStream_t stream, bool final_launch, PhotoFinal* fin) {
    SYN3R_REQUIRE(C > 0 && C <= 65535 && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W), "%s: bad sizes C=%d H=%d W=%d", who, C, H, W);
    SYN3R_REQUIRE(image && target && loss3 && ws, "%s: null pointer", who);
    SYN3R_REQUIRE(lambda_dssim >= 0.0f && lambda_dssim <= 1.0f, "%s: lambda_dssim must be in [0, 1]", who);
    SYN3R_REQUIRE(ws_bytes >= syn3r_photo_loss_workspace_bytes(C, H, W), "%s: workspace too small", who);
    const size_t n = (size_t)C * H * W;
    float* maps = (float*)ws;
    float* partial = maps + 3 * n;
    const long long tiles = (long long)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile) * C;
    SYN3R_REQUIRE(tiles < (1ll << 31), "%s: %lld tiles exceed the grid limit", who, tiles);
    const dim3 grid((unsigned)tiles);
    const bool vec = W % 4 == 0 && (((uintptr_t)image | (uintptr_t)target) & 15) == 0;      // aligned 16-byte row runs
    if (vec) SYN3R_LAUNCH(k_photo_fwd<true>, grid, dim3(kPhotoThreads), 0, stream, image, target, H, W, C, make_window(), maps, partial);
    else SYN3R_LAUNCH(k_photo_fwd<false>, grid, dim3(kPhotoThreads), 0, stream, image, target, H, W, C, make_window(), maps, partial);
    const PhotoFinal f{partial, (long long)grid.x, 1.0 / (double)n, lambda_dssim, weight, loss3};
    if (final_launch) SYN3R_LAUNCH(k_photo_final, dim3(1), dim3(256), 0, stream, f);
    if (fin) *fin = f;
    return SYN3R_OK;
}

static int photo_backward(const char* who, const float* image, const float* target, int C, int H, int W, float lambda_dssim,
                          float weight, const float* grad_loss, const void* ws, float* grad_image, hipStream_t stream, PhotoFinal fin) {
    SYN3R_REQUIRE(C > 0 && C <= 65535 && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W), "%s: bad sizes", who);
    SYN3R_REQUIRE(image && target && ws && grad_image, "%s: null pointer", who);
    const double n = (double)C * H * W;
    const long long tiles = (long long)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile) * C;
    SYN3R_REQUIRE(tiles < (1ll << 31), "%s: %lld tiles exceed the grid limit", who, tiles);
    const dim3 grid((unsigned)tiles);
    const bool vec = W % 4 == 0 && ((uintptr_t)ws & 15) == 0;
    const float c_l1 = (float)((double)weight * (1.0 - (double)lambda_dssim) / n), c_ssim = (float)((double)weight * (double)lambda_dssim / n);
    if (vec) SYN3R_LAUNCH(k_photo_bwd<true>, grid, dim3(kPhotoThreads), 0, stream, image, target, H, W, C, make_window(), (const float*)ws,
                          c_l1, c_ssim, grad_loss, grad_image, fin);
    else SYN3R_LAUNCH(k_photo_bwd<false>, grid, dim3(kPhotoThreads), 0, stream, image, target, H, W, C, make_window(), (const float*)ws,
                      c_l1, c_ssim, grad_loss, grad_image, fin);
    return SYN3R_OK;
}

extern "C" int syn3r_photo_loss(const float* image, const float* target, int C, int H, int W, float lambda_dssim,
                                float weight, float* loss3, void* ws, size_t ws_bytes, void* stream_) {
    const int rc = photo_forward("photo_loss", image, target, C, H, W, lambda_dssim, weight, loss3, ws, ws_bytes, (hipStream_t)stream_,
                                 true, nullptr);
    if (rc) return rc;
    SYN3R_LAUNCH_CHECK("photo_loss launch");
    return SYN3R_OK;
}

extern "C" int syn3r_photo_loss_backward(const float* image, const float* target, int C, int H, int W,
                                         float lambda_dssim, float weight, const float* grad_loss, const void* ws,
                                         float* grad_image, void* stream_) {
    const int rc = photo_backward("photo_loss_backward", image, target, C, H, W, lambda_dssim, weight, grad_loss, ws, grad_image,
                                  (hipStream_t)stream_, PhotoFinal{});
    if (rc) return rc;
    SYN3R_LAUNCH_CHECK("photo_loss_backward launch");
    return SYN3R_OK;
}

extern "C" int syn3r_photo_loss_step(const float* image, const float* target, int C, int H, int W, float lambda_dssim,
                                     float weight, const float* grad_loss, float* loss3, float* grad_image, void* ws,
                                     size_t ws_bytes, void* stream_) {
    PhotoFinal fin{};
    int rc = photo_forward("photo_loss_step", image, target, C, H, W, lambda_dssim, weight, loss3, ws, ws_bytes, (hipStream_t)stream_,
                           false, &fin);
    if (rc) return rc;
    rc = photo_backward("photo_loss_step", image, target, C, H, W, lambda_dssim, weight, grad_loss, ws, grad_image,
                        (hipStream_t)stream_, fin);
    if (rc) return rc;
    SYN3R_LAUNCH_CHECK("photo_loss_step launch");
    return SYN3R_OK;
}
