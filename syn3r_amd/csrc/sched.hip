// Fused modified-Euler scheduler steps.
//
// Reference behaviour restated from
//   thirdparty/diffusers/src/diffusers/schedulers/scheduling_euler_discrete.py
//     :633-814   step_interp                (guidance-gradient variant)
//     :1343-1515 step_interp_prob_uncertain (soft-replacement variant)
//
// The reference sorts 4*h*w values per frame with torch.sort and syncs to the
// host once per frame to index the sorted array.  Here one workgroup per frame
// finds the same order statistic with a 4-pass 8-bit radix select on the fp32
// bit patterns (non-negative floats order as unsigned ints), entirely on
// device: LDS histograms, wavefront shuffles for the digit scan.  HBM-bound;
// algorithmic bytes are listed in DESIGN.md.
#include "common.h"

using namespace syn3r;

namespace {

constexpr int kSel = 1024;  // threads per frame for the select kernel
constexpr int kEw = 256;

template <int DT>
__device__ __forceinline__ float ld(const void* p, size_t i) {
    if constexpr (DT == SYN3R_F16) return __half2float(((const __half*)p)[i]);
    else return ((const float*)p)[i];
}
template <int DT>
__device__ __forceinline__ void st(void* p, size_t i, float v) {
    if constexpr (DT == SYN3R_F16) ((__half*)p)[i] = __float2half_rn(v);
    else ((float*)p)[i] = v;
}

// v-prediction x0 (:728).  With a half model_output the product with the 0-dim
// fp32 scalar is rounded to half by torch's type promotion before the fp32 add.
template <int VDT>
__device__ __forceinline__ float pred_x0(float v, float x, float c_out, float denom) {
    float t = v * c_out;
    if constexpr (VDT == SYN3R_F16) t = __half2float(__float2half_rn(t));
    return t + x / denom;
}

constexpr int kMaxFrames = 64;

struct StepParams {
    float sigma, dt, c_out, denom, sqrt_sigma, lr;
    int F, C, h, w;
};

// lambda_ts[step_i] travels by value (host pointer in the ABI, no async host copy)
struct LambdaRow { double v[kMaxFrames]; };

// valid-pixel test: mean over channels of ((1-mask) > 0.5) > 0.5   (:742,:752-753)
__device__ __forceinline__ float mean_valid(const float* __restrict__ mask_f, int C, int hw, int pix) {
    int cnt = 0;
    for (int c = 0; c < C; ++c) cnt += ((1.0f - mask_f[(size_t)c * hw + pix]) > 0.5f) ? 1 : 0;
    return (float)cnt / (float)C;
}

// One workgroup per interior frame tau = blockIdx.x + 1.
// Writes cutoff[tau] (the value sorted_diff[k-1]) and stores x0 (optional) and |d| bits.
template <int VDT, int SDT>
__global__ void __launch_bounds__(kSel) k_select(StepParams p, const void* __restrict__ v_,
                                                 const void* __restrict__ x_, const float* __restrict__ cond,
                                                 const float* __restrict__ mask, LambdaRow lam,
                                                 unsigned* __restrict__ dbits, float* __restrict__ cutoff) {
    const int tau = blockIdx.x + 1;
    const int hw = p.h * p.w;
    const int n = p.C * hw;
    const size_t base = (size_t)tau * n;
    const float* mask_f = mask + (size_t)(tau - 1) * n;
    unsigned* db = dbits + base;

    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_n0;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_n0 = 0;
    __syncthreads();

    // pass 0: compute |d| bits, count masked pixels, histogram of the top byte
    int n0_local = 0;
    for (int i = threadIdx.x; i < n; i += kSel) {
        int pix = i % hw;
        bool mt = mean_valid(mask_f, p.C, hw, pix) > 0.5f;
        if (i < hw && !mt) ++n0_local;
        float x0 = pred_x0<VDT>(ld<VDT>(v_, base + i), ld<SDT>(x_, base + i), p.c_out, p.denom);
        float mf = mt ? 1.0f : 0.0f;
        float d = fabsf(x0 * mf - cond[base + i] * mf);
        unsigned u = __float_as_uint(d);
        db[i] = u;
        atomicAdd(&hist[u >> 24], 1u);
    }
    n0_local = wave_sum_i(n0_local);
    if ((threadIdx.x & 63) == 0 && n0_local) atomicAdd(&s_n0, (unsigned)n0_local);
    __syncthreads();

    // cutoff index (:768-770): k = int(clamp(lambda,0.4,1) * (n - n0)) + n0 ; value = sorted[k-1]
    if (threadIdx.x == 0) {
        double wgt = lam.v[tau];
        wgt = wgt < 0.4 ? 0.4 : (wgt > 1.0 ? 1.0 : wgt);
        long long n0 = (long long)s_n0;
        long long k = (long long)(wgt * (double)((long long)n - n0)) + n0;
        long long idx = k - 1;
        if (idx < 0) idx += n;  // python negative index
        if (idx > n - 1) idx = n - 1;
        s_rank = (unsigned)idx;  // 0-based rank of the wanted element
        s_prefix = 0;
    }
    __syncthreads();

    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (pass > 0) {
            if (threadIdx.x < 256) hist[threadIdx.x] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const unsigned himask = 0xFFFFFFFFu << (shift + 8);
            for (int i = threadIdx.x; i < n; i += kSel) {
                unsigned u = db[i];
                if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
            __syncthreads();
        }
        // wave 0 scans the 256 bins (4 per lane) and picks the bin holding s_rank
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            unsigned tot = c0 + c1 + c2 + c3;
            unsigned incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                unsigned t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            unsigned excl = incl - tot;
            unsigned rank = s_rank;
            if (rank >= excl && rank < incl) {
                unsigned r = rank - excl;
                unsigned digit;
                if (r < c0) { digit = 4 * lane; }
                else if (r < c0 + c1) { digit = 4 * lane + 1; r -= c0; }
                else if (r < c0 + c1 + c2) { digit = 4 * lane + 2; r -= c0 + c1; }
                else { digit = 4 * lane + 3; r -= c0 + c1 + c2; }
                s_rank = r;
                s_prefix = s_prefix | (digit << shift);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) cutoff[tau] = __uint_as_float(s_prefix);
}

// Elementwise tail of step_interp: top mask, g = (x0 - cond) * top, sums for std, Euler update.
template <int VDT, int SDT>
__global__ void __launch_bounds__(kEw) k_interp_tail(StepParams p, const void* __restrict__ v_,
                                                     const void* __restrict__ x_, const float* __restrict__ cond,
                                                     const float* __restrict__ mask,
                                                     const float* __restrict__ cutoff, int compute_grad,
                                                     void* __restrict__ prev, float* __restrict__ x0_out,
                                                     float* __restrict__ grad, double* __restrict__ sums) {
    const int hw = p.h * p.w;
    const int n = p.C * hw;
    const size_t total = (size_t)p.F * n;
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * kEw + threadIdx.x; i < total; i += (size_t)gridDim.x * kEw) {
        int f = (int)(i / n);
        int j = (int)(i - (size_t)f * n);
        float x = ld<SDT>(x_, i);
        float x0 = pred_x0<VDT>(ld<VDT>(v_, i), x, p.c_out, p.denom);
        if (x0_out) x0_out[i] = x0;
        if (compute_grad) {
            float c = cond[i];
            bool top = true;  // first and last frame: ones (:777-779)
            if (f > 0 && f < p.F - 1) {
                int pix = j % hw;
                bool mt = mean_valid(mask + (size_t)(f - 1) * n, p.C, hw, pix) > 0.5f;
                float mf = mt ? 1.0f : 0.0f;
                float d = fabsf(x0 * mf - c * mf);
                top = (d <= cutoff[f]) && mt;
            }
            float g = top ? (x0 - c) : 0.0f;
            grad[i] = g;
            s1 += (double)g;
            s2 += (double)g * (double)g;
        }
        // Euler update (:798-804)
        float deriv = (x - x0) / p.sigma;
        st<VDT>(prev, i, x + deriv * p.dt);
    }
    if (compute_grad) {
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        __shared__ double r1[kEw / 64], r2[kEw / 64];
        int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) { r1[wv] = s1; r2[wv] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 1; k < kEw / 64; ++k) { s1 += r1[k]; s2 += r2[k]; }
            unsafeAtomicAdd(&sums[0], s1);
            unsafeAtomicAdd(&sums[1], s2);
        }
    }
}

// grad = lr * (g / std(g) * sigma^0.5), std unbiased over all elements (:789-795)
__global__ void __launch_bounds__(kEw) k_grad_scale(StepParams p, float* __restrict__ grad,
                                                    const double* __restrict__ sums) {
    const size_t total = (size_t)p.F * p.C * p.h * p.w;
    const double N = (double)total;
    const double mean = sums[0] / N;
    double var = (sums[1] - N * mean * mean) / (N - 1.0);
    const float sd = (float)sqrt(var > 0.0 ? var : 0.0);
    for (size_t i = (size_t)blockIdx.x * kEw + threadIdx.x; i < total; i += (size_t)gridDim.x * kEw) {
        grad[i] = p.lr * (grad[i] / sd * p.sqrt_sigma);
    }
}

// Elementwise tail of step_interp_prob_uncertain (:1482-1505)
template <int VDT, int SDT>
__global__ void __launch_bounds__(kEw) k_replace_tail(StepParams p, const void* __restrict__ v_,
                                                      const void* __restrict__ x_, const float* __restrict__ cond,
                                                      const float* __restrict__ mask,
                                                      const float* __restrict__ cutoff, void* __restrict__ prev,
                                                      float* __restrict__ x0_out) {
    const int hw = p.h * p.w;
    const int n = p.C * hw;
    const size_t total = (size_t)p.F * n;
    for (size_t i = (size_t)blockIdx.x * kEw + threadIdx.x; i < total; i += (size_t)gridDim.x * kEw) {
        int f = (int)(i / n);
        int j = (int)(i - (size_t)f * n);
        float x = ld<SDT>(x_, i);
        float x0 = pred_x0<VDT>(ld<VDT>(v_, i), x, p.c_out, p.denom);
        float c = cond[i];
        if (f == 0 || f == p.F - 1) {
            x0 = c;
        } else {
            int pix = j % hw;
            float mbar = mean_valid(mask + (size_t)(f - 1) * n, p.C, hw, pix);
            float mf = (mbar > 0.5f) ? 1.0f : 0.0f;
            float d = fabsf(x0 * mf - c * mf);
            float t = 1.0f / (1.0f - mbar + 1e-6f);
            float wgt = t / (1.0f + t);
            wgt = (wgt >= 0.51f) ? wgt : 0.0f;
            wgt = ((d <= cutoff[f]) ? 1.0f : 0.0f) * wgt;
            x0 = (1.0f - wgt) * x0 + wgt * c;
        }
        if (x0_out) x0_out[i] = x0;
        float deriv = (x - x0) / p.sigma;
        st<VDT>(prev, i, x + deriv * p.dt);
    }
}

struct Ws {
    double* sums;      // 2 doubles (+pad)
    float* cutoff;     // F floats
    unsigned* dbits;   // F*C*h*w
};

size_t ws_bytes(int F, int C, int h, int w) {
    return 16 + (((size_t)F * 4 + 15) / 16) * 16 + (size_t)F * C * h * w * 4;
}

Ws carve(void* ws, int F) {
    Ws r;
    r.sums = (double*)ws;
    r.cutoff = (float*)((char*)ws + 16);
    r.dbits = (unsigned*)((char*)ws + 16 + (((size_t)F * 4 + 15) / 16) * 16);
    return r;
}

int ew_grid(size_t total) {
    size_t g = (total + kEw - 1) / kEw;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

template <int VDT, int SDT>
int run_step(bool replace, const void* v, const void* x, const float* cond, const float* mask,
             const double* lambda_row, const StepParams& p, int compute_grad, void* prev, float* x0, float* grad,
             void* workspace, hipStream_t stream) {
    Ws ws = carve(workspace, p.F);
    const size_t total = (size_t)p.F * p.C * p.h * p.w;
    const bool need_select = replace || compute_grad;
    if (need_select && p.F > 2) {
        LambdaRow lam;
        for (int f = 0; f < p.F; ++f) lam.v[f] = lambda_row[f];
        SYN3R_LAUNCH((k_select<VDT, SDT>), dim3(p.F - 2), dim3(kSel), 0, stream, p, v, x, cond, mask, lam,
                           ws.dbits, ws.cutoff);
    }
    if (replace) {
        SYN3R_LAUNCH((k_replace_tail<VDT, SDT>), dim3(ew_grid(total)), dim3(kEw), 0, stream, p, v, x, cond, mask,
                           ws.cutoff, prev, x0);
    } else {
        if (compute_grad) {
            int rc = check_hip(hipMemsetAsync(ws.sums, 0, 16, stream), "memset");
            if (rc) return rc;
        }
        SYN3R_LAUNCH((k_interp_tail<VDT, SDT>), dim3(ew_grid(total)), dim3(kEw), 0, stream, p, v, x, cond, mask,
                           ws.cutoff, compute_grad, prev, x0, grad, ws.sums);
        if (compute_grad)
            SYN3R_LAUNCH(k_grad_scale, dim3(ew_grid(total)), dim3(kEw), 0, stream, p, grad, ws.sums);
    }
    SYN3R_LAUNCH_CHECK("scheduler step launch");
    return SYN3R_OK;
}

int dispatch(bool replace, const void* v, int vdt, const void* x, int sdt, const float* cond, const float* mask,
             const double* lambda_row, const StepParams& p, int compute_grad, void* prev, float* x0, float* grad,
             void* workspace, size_t workspace_bytes, void* stream_) {
    SYN3R_REQUIRE(v && x && prev, "scheduler step: null tensor");
    SYN3R_REQUIRE(p.F >= 1 && p.C >= 1 && p.h >= 1 && p.w >= 1, "scheduler step: bad shape F=%d C=%d h=%d w=%d", p.F,
                  p.C, p.h, p.w);
    SYN3R_REQUIRE((vdt == SYN3R_F16 || vdt == SYN3R_F32) && (sdt == SYN3R_F16 || sdt == SYN3R_F32),
                  "scheduler step: unsupported dtype");
    const bool need_select = replace || compute_grad;
    if (need_select) {
        SYN3R_REQUIRE(cond && mask && lambda_row, "scheduler step: conditioning tensors required");
        SYN3R_REQUIRE(p.F >= 3 && p.F <= kMaxFrames, "scheduler step: frames must be in [3,%d], got %d", kMaxFrames, p.F);
        SYN3R_REQUIRE((long long)p.C * p.h * p.w < (1ll << 31), "scheduler step: frame too large");
    }
    SYN3R_REQUIRE(replace || !compute_grad || grad, "step_interp: grad output required when compute_grad");
    size_t need = ws_bytes(p.F, p.C, p.h, p.w);
    if (!workspace || workspace_bytes < need) {
        set_error("scheduler step: workspace %zu < %zu", workspace_bytes, need);
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream_;
    if (vdt == SYN3R_F16 && sdt == SYN3R_F16)
        return run_step<SYN3R_F16, SYN3R_F16>(replace, v, x, cond, mask, lambda_row, p, compute_grad, prev, x0, grad, workspace, s);
    if (vdt == SYN3R_F16 && sdt == SYN3R_F32)
        return run_step<SYN3R_F16, SYN3R_F32>(replace, v, x, cond, mask, lambda_row, p, compute_grad, prev, x0, grad, workspace, s);
    if (vdt == SYN3R_F32 && sdt == SYN3R_F16)
        return run_step<SYN3R_F32, SYN3R_F16>(replace, v, x, cond, mask, lambda_row, p, compute_grad, prev, x0, grad, workspace, s);
    return run_step<SYN3R_F32, SYN3R_F32>(replace, v, x, cond, mask, lambda_row, p, compute_grad, prev, x0, grad, workspace, s);
}

}  // namespace

extern "C" size_t syn3r_step_workspace_bytes(int F, int C, int h, int w) {
    if (F <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    return ws_bytes(F, C, h, w);
}

extern "C" int syn3r_step_interp(const void* model_output, int vdtype, const void* sample, int sdtype,
                                 const float* cond, const float* mask, const double* lambda_row, float sigma,
                                 float dt, float c_out, float denom, float sqrt_sigma, float lr, int compute_grad,
                                 void* prev_sample, float* pred_x0, float* grad, int F, int C, int h, int w,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    StepParams p{sigma, dt, c_out, denom, sqrt_sigma, lr, F, C, h, w};
    return dispatch(false, model_output, vdtype, sample, sdtype, cond, mask, lambda_row, p, compute_grad, prev_sample,
                    pred_x0, grad, workspace, workspace_bytes, stream);
}

extern "C" int syn3r_step_replace(const void* model_output, int vdtype, const void* sample, int sdtype,
                                  const float* cond, const float* mask, const double* lambda_row, float sigma,
                                  float dt, float c_out, float denom, void* prev_sample, float* pred_x0, int F,
                                  int C, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
    StepParams p{sigma, dt, c_out, denom, 0.0f, 0.0f, F, C, h, w};
    return dispatch(true, model_output, vdtype, sample, sdtype, cond, mask, lambda_row, p, 0, prev_sample, pred_x0,
                    nullptr, workspace, workspace_bytes, stream);
}
