// LPIPS (VGG16) perceptual term of the trainer's loss: the element-wise / per-pixel pieces around the
// 3x3 convolutions (which run on the implicit-GEMM kernel of gemm.hip, syn3r_conv2d3x3_act_f16).
//
// The reference toggles `gsTrainer.opt.use_lpips_loss` for every refine (model/diffusionGS.py:1690,1697); the loss itself
// lives in FSGS (un-vendored) and calls the `lpips` package, which is not in /root/reference either.  What is restated is
// the PUBLISHED definition (Zhang et al. 2018, `lpips.LPIPS(net='vgg')` v0.1): inputs scaled to [-1,1], the fixed
// per-channel shift / scale, VGG16 features after relu1_2, 2_2, 3_3, 4_3, 5_3, each normalised to unit length over
// channels (eps 1e-10 added to the norm), squared difference, a learned non-negative 1x1 weighting, spatial mean, summed
// over the five layers.  Weights are the caller's (no checkpoint is reachable offline): PARITY UNPINNED, oracle =
// oracle/lpips_oracle.py (torch fp32).
//
// Layout: activations are channels-last fp16 [H*W, C] (the UNet's token-matrix layout); the image enters as [3,H,W] fp32
// in [0,1] and its gradient leaves in the same form.  Gradients travel through the fp16 convolutions multiplied by a loss
// scale (they are ~1e-9 unscaled: below fp16's subnormal range) which k_lpips_image_bwd divides out.
// All kernels are HBM-bound single passes.
#include "common.h"

using namespace syn3r;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__constant__ float kShift[3] = {-0.030f, -0.088f, -0.188f};
__constant__ float kScale[3] = {0.458f, 0.448f, 0.450f};

// [3,H,W] fp32 in [0,1] -> [H*W, 64] fp16: channel c = ((2x - 1) - shift_c) / scale_c, channels 3..63 zero (the convolution
// kernel wants Cin % 64 == 0)
__global__ void __launch_bounds__(256) k_lpips_image(const float* __restrict__ img, long long hw, __half* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per (pixel, 8-channel chunk)
    if (i >= hw * 8) return;
    const long long p = i >> 3;
    const int ch = (int)(i & 7);
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
    if (ch == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (_Float16)(((2.0f * img[c * hw + p] - 1.0f) - kShift[c]) / kScale[c]);
    }
    *(half8*)(out + p * 64 + ch * 8) = v;
}

// gradient wrt the [H*W, 64] input of conv1_1 (scaled by loss_scale) -> d loss / d image [3,H,W] fp32
__global__ void __launch_bounds__(256) k_lpips_image_bwd(const __half* __restrict__ g, long long hw, float inv_scale, float* __restrict__ d_img) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_img[c * hw + p] = __half2float(g[p * 64 + c]) * (2.0f / kScale[c]) * inv_scale;
}

// 2x2 / stride-2 max pooling on [H,W,C] -> [H/2,W/2,C] (floor, as nn.MaxPool2d(2,2))
__global__ void __launch_bounds__(256) k_maxpool2(const __half* __restrict__ x, int H, int W, int C, __half* __restrict__ y) {
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Ho * Wo * C8) return;
    const int ch = (int)(i % C8);
    const long long q = i / C8;
    const int xo = (int)(q % Wo), yo = (int)(q / Wo);
    const __half* s = x + ((long long)(2 * yo) * W + 2 * xo) * C + ch * 8;
    const half8 a = *(const half8*)s, b = *(const half8*)(s + C), c = *(const half8*)(s + (long long)W * C), d = *(const half8*)(s + (long long)W * C + C);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m = fmaxf(fmaxf((float)a[e], (float)b[e]), fmaxf((float)c[e], (float)d[e]));
        o[e] = (_Float16)m;
    }
    *(half8*)(y + q * C + ch * 8) = o;
}

// its backward: the gradient of an output cell goes to the FIRST maximum of its window (row-major order, as torch); cells of
// an odd last row / column receive zero.  gx is written in full (no accumulation: a pooled activation feeds nothing else).
__global__ void __launch_bounds__(256) k_maxpool2_bwd(const __half* __restrict__ x, const __half* __restrict__ gy, int H, int W, int C,
                                                     __half* __restrict__ gx) {
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * W * C8) return;
    const int ch = (int)(i % C8);
    const long long q = i / C8;
    const int xi = (int)(q % W), yi = (int)(q / W);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)0.0f;
    const int yo = yi >> 1, xo = xi >> 1;
    if (yo < Ho && xo < Wo) {
        const __half* s = x + ((long long)(2 * yo) * W + 2 * xo) * C + ch * 8;
        const half8 a = *(const half8*)s, b = *(const half8*)(s + C), c = *(const half8*)(s + (long long)W * C), d = *(const half8*)(s + (long long)W * C + C);
        const half8 g = *(const half8*)(gy + ((long long)yo * Wo + xo) * C + ch * 8);
        const int me = (yi & 1) * 2 + (xi & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float va = (float)a[e], vb = (float)b[e], vc = (float)c[e], vd = (float)d[e];
            int arg = 0;
            float m = va;
            if (vb > m) { m = vb; arg = 1; }
            if (vc > m) { m = vc; arg = 2; }
            if (vd > m) { m = vd; arg = 3; }
            if (arg == me) o[e] = g[e];
        }
    }
    *(half8*)(gx + q * C + ch * 8) = o;
}

// One LPIPS layer on two feature maps a, b [P, C] fp16 (C = 64 .. 512, a multiple of 64; LPR = C / 8 lanes per pixel):
//   value_p = sum_c w_c (a_c / (|a| + eps) - b_c / (|b| + eps))^2 ;  the layer's term is mean_p value_p.
// Forward: per-block partial sums (fixed order) -> k_lpips_finish adds them up.
template <int LPR>
__global__ void __launch_bounds__(256) k_lpips_layer(const __half* __restrict__ a, const __half* __restrict__ b, const float* __restrict__ w,
                                                    long long P, float* __restrict__ partial) {
    constexpr int C = LPR * 8, PPB = 256 / LPR;       // pixels per block
    const int sub = threadIdx.x % LPR;
    const long long p = (long long)blockIdx.x * PPB + threadIdx.x / LPR;
    float val = 0.0f;
    if (p < P) {
        const half8 av = *(const half8*)(a + p * C + sub * 8), bv = *(const half8*)(b + p * C + sub * 8);
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa += (float)av[e] * (float)av[e]; sb += (float)bv[e] * (float)bv[e]; }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
        const float ra = 1.0f / (sqrtf(sa) + 1e-10f), rb = 1.0f / (sqrtf(sb) + 1e-10f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)av[e] * ra - (float)bv[e] * rb;
            val += w[sub * 8 + e] * d * d;
        }
    }
    // block sum in a fixed order: wave shuffles, then the 4 wave sums
    for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// out[0] (+)= sum(partial) / P, one block, fixed order
__global__ void __launch_bounds__(1024) k_lpips_finish(const float* __restrict__ partial, int n, float inv_P, int accumulate, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + red[0] * inv_P;
}

// Backward of one layer wrt a: with n = a / r, r = |a| + eps, e_c = 2 w_c (n_c - b_c / (|b| + eps)) / P,
//   d value / d a_k = e_k / r - (sum_c e_c a_c) a_k / (r^2 |a|)
// written (accumulate = 0) or added (1) to ga [P, C] fp16, multiplied by `gscale` (upstream gradient x loss scale); the sum
// is then masked by a > 0 (the feature map is the output of a ReLU: what leaves is the gradient wrt its pre-activation).
template <int LPR>
__global__ void __launch_bounds__(256) k_lpips_layer_bwd(const __half* __restrict__ a, const __half* __restrict__ b, const float* __restrict__ w,
                                                        long long P, float gscale, int accumulate, __half* __restrict__ ga) {
    constexpr int C = LPR * 8, PPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long long p = (long long)blockIdx.x * PPB + threadIdx.x / LPR;
    if (p >= P) return;                                  // (whole LPR-lane groups leave together: the shuffles below stay inside a group)
    const half8 av = *(const half8*)(a + p * C + sub * 8), bv = *(const half8*)(b + p * C + sub * 8);
    float sa = 0.0f, sb = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa += (float)av[e] * (float)av[e]; sb += (float)bv[e] * (float)bv[e]; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    const float na = sqrtf(sa);
    const float ra = 1.0f / (na + 1e-10f), rb = 1.0f / (sqrtf(sb) + 1e-10f);
    float ev[8], dot = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ev[e] = 2.0f * w[sub * 8 + e] * ((float)av[e] * ra - (float)bv[e] * rb) * gscale;
        dot += ev[e] * (float)av[e];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    const float k2 = na > 0.0f ? dot * ra * ra / na : 0.0f;
    half8 o;
    if (accumulate) o = *(const half8*)(ga + p * C + sub * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float g = ev[e] * ra - k2 * (float)av[e];
        // `a` is a post-ReLU activation: the total gradient passes its ReLU here (zero where the activation is zero).
        // The gradient carries the loss scale (1024 H W / P_k) and a factor 1 / |a|: a pixel whose feature norm is tiny would
        // overflow fp16 to inf and the backward convolutions would spread inf / NaN into the image gradient - saturate.
        const float t = fminf(fmaxf((accumulate ? (float)o[e] : 0.0f) + g, -65504.0f), 65504.0f);
        o[e] = (float)av[e] > 0.0f ? (_Float16)t : (_Float16)0.0f;
    }
    *(half8*)(ga + p * C + sub * 8) = o;
}

template <int LPR>
int launch_layer(const __half* a, const __half* b, const float* w, long long P, float* partial, hipStream_t stream) {
    const int blocks = (int)((P + 256 / LPR - 1) / (256 / LPR));
    SYN3R_LAUNCH_NAMED("k_lpips_layer", k_lpips_layer<LPR>, dim3(blocks), dim3(256), 0, stream, a, b, w, P, partial);
    return blocks;
}
template <int LPR>
void launch_layer_bwd(const __half* a, const __half* b, const float* w, long long P, float gscale, int acc, __half* ga, hipStream_t stream) {
    const int blocks = (int)((P + 256 / LPR - 1) / (256 / LPR));
    SYN3R_LAUNCH_NAMED("k_lpips_layer_bwd", k_lpips_layer_bwd<LPR>, dim3(blocks), dim3(256), 0, stream, a, b, w, P, gscale, acc, ga);
}

}  // namespace

extern "C" int syn3r_lpips_image_f16(const float* img, int H, int W, void* out, void* stream_) {
    SYN3R_REQUIRE(img && out && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W), "lpips_image: bad arguments H=%d W=%d", H, W);
    const long long hw = (long long)H * W;
    SYN3R_LAUNCH(k_lpips_image, dim3((unsigned)((hw * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, img, hw, (__half*)out);
    SYN3R_LAUNCH_CHECK("lpips_image launch");
    return SYN3R_OK;
}

extern "C" int syn3r_lpips_image_bwd(const void* grad64, int H, int W, float loss_scale, float* d_img, void* stream_) {
    SYN3R_REQUIRE(grad64 && d_img && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && loss_scale > 0.0f, "lpips_image_bwd: bad arguments");
    const long long hw = (long long)H * W;
    SYN3R_LAUNCH(k_lpips_image_bwd, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (const __half*)grad64, hw,
                 1.0f / loss_scale, d_img);
    SYN3R_LAUNCH_CHECK("lpips_image_bwd launch");
    return SYN3R_OK;
}

extern "C" int syn3r_maxpool2_f16(const void* x, int H, int W, int C, void* y, void* stream_) {
    SYN3R_REQUIRE(x && y && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && H >= 2 && W >= 2 && SYN3R_DIM_OK(C) && C % 8 == 0, "maxpool2: bad arguments H=%d W=%d C=%d", H, W, C);
    const long long n = (long long)(H / 2) * (W / 2) * (C / 8);
    SYN3R_LAUNCH(k_maxpool2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (const __half*)x, H, W, C, (__half*)y);
    SYN3R_LAUNCH_CHECK("maxpool2 launch");
    return SYN3R_OK;
}

extern "C" int syn3r_maxpool2_bwd_f16(const void* x, const void* gy, int H, int W, int C, void* gx, void* stream_) {
    SYN3R_REQUIRE(x && gy && gx && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && H >= 2 && W >= 2 && SYN3R_DIM_OK(C) && C % 8 == 0, "maxpool2_bwd: bad arguments");
    const long long n = (long long)H * W * (C / 8);
    SYN3R_LAUNCH(k_maxpool2_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, (const __half*)x, (const __half*)gy, H, W, C,
                 (__half*)gx);
    SYN3R_LAUNCH_CHECK("maxpool2_bwd launch");
    return SYN3R_OK;
}

extern "C" size_t syn3r_lpips_layer_workspace_bytes(long long P, int C) {
    if (P <= 0 || C < 64 || C > 512 || C % 64) return 0;
    const long long ppb = 256 / (C / 8);
    return (size_t)((P + ppb - 1) / ppb) * 4 + 256;
}

// value[0] (+)= mean_p sum_c w_c (a_c/(|a|+eps) - b_c/(|b|+eps))^2     a, b [P, C] fp16; w [C] fp32; value fp32 (device)
extern "C" int syn3r_lpips_layer_f16(const void* a, const void* b, const float* w, long long P, int C, int accumulate, float* value,
                                     void* ws, size_t ws_bytes, void* stream_) {
    SYN3R_REQUIRE(a && b && w && value && ws, "lpips_layer: null argument");
    SYN3R_REQUIRE(P > 0 && P < (1ll << 31) && (C == 64 || C == 128 || C == 256 || C == 512), "lpips_layer: P=%lld, C=%d (64 / 128 / 256 / 512)", P, C);
    SYN3R_REQUIRE(ws_bytes >= syn3r_lpips_layer_workspace_bytes(P, C), "lpips_layer: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    float* partial = (float*)ws;
    int blocks = 0;
    const __half *ah = (const __half*)a, *bh = (const __half*)b;
    if (C == 64) blocks = launch_layer<8>(ah, bh, w, P, partial, stream);
    else if (C == 128) blocks = launch_layer<16>(ah, bh, w, P, partial, stream);
    else if (C == 256) blocks = launch_layer<32>(ah, bh, w, P, partial, stream);
    else blocks = launch_layer<64>(ah, bh, w, P, partial, stream);
    SYN3R_LAUNCH(k_lpips_finish, dim3(1), dim3(1024), 0, stream, (const float*)partial, blocks, 1.0f / (float)P, accumulate, value);
    SYN3R_LAUNCH_CHECK("lpips_layer launch");
    return SYN3R_OK;
}

// grad_a [P, C] fp16 (=, or += with accumulate) d(layer term)/d a * gscale
extern "C" int syn3r_lpips_layer_bwd_f16(const void* a, const void* b, const float* w, long long P, int C, float gscale, int accumulate,
                                         void* grad_a, void* stream_) {
    SYN3R_REQUIRE(a && b && w && grad_a, "lpips_layer_bwd: null argument");
    SYN3R_REQUIRE(P > 0 && P < (1ll << 31) && (C == 64 || C == 128 || C == 256 || C == 512), "lpips_layer_bwd: P=%lld, C=%d", P, C);
    hipStream_t stream = (hipStream_t)stream_;
    const __half *ah = (const __half*)a, *bh = (const __half*)b;
    __half* g = (__half*)grad_a;
    const float gs = gscale / (float)P;
    if (C == 64) launch_layer_bwd<8>(ah, bh, w, P, gs, accumulate, g, stream);
    else if (C == 128) launch_layer_bwd<16>(ah, bh, w, P, gs, accumulate, g, stream);
    else if (C == 256) launch_layer_bwd<32>(ah, bh, w, P, gs, accumulate, g, stream);
    else launch_layer_bwd<64>(ah, bh, w, P, gs, accumulate, g, stream);
    SYN3R_LAUNCH_CHECK("lpips_layer_bwd launch");
    return SYN3R_OK;
}
