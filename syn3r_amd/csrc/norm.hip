// HBM-bound normalisation / activation kernels of the SVD UNet on channels-last fp16 tensors
// (statistics and arithmetic in fp32, 16-byte vector accesses).
//
//   GroupNorm(32) [+ SiLU]   resnet.py:272,286,574,588 ; transformer_temporal.py:235 ; unet_...:243-244
//                            2D form: one sample = one frame (R = h*w rows)
//                            3D form: one sample = one batch item (R = F*h*w rows, contiguous)
//   LayerNorm                attention.py:195,225,253,430,440,453,465 — optionally fused with the
//                            broadcast add of the frame-position embedding (transformer_temporal.py:355)
//   GEGLU                    activations.py GEGLU.forward: hidden * gelu(gate), exact erf GELU
#include "common.h"

using namespace syn3r;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- GroupNorm
// thread t owns channel vector cv = t % cvec (8 channels) of rows (t / cvec) + k * rpb
// Two-source input: channels [0, C1) of a row come from x (row stride C1), channels [C1, C) from x2 (row stride C - C1):
// the concatenation [x | x2] along the channels that the up blocks normalise (unet_3d_blocks.py: torch.cat of the
// hidden state and the skip) is never written.  x2 = null (C1 = C): one tensor.
__device__ __forceinline__ const __half* gn_src(const __half* x, const __half* x2, int C, int C1, int cv, size_t sample_rows,
                                                int& stride) {
    if (cv * 8 < C1) { stride = C1; return x + sample_rows * (size_t)C1 + (size_t)cv * 8; }
    stride = C - C1;
    return x2 + sample_rows * (size_t)(C - C1) + (size_t)(cv * 8 - C1);
}

__global__ void k_gn_stats(const __half* __restrict__ x, const __half* __restrict__ x2, int C1, int R, int C, int rows_per_block,
                           float* __restrict__ stats) {
    extern __shared__ float red[];   // [threads][16]
    const int cvec = C >> 3;
    const int rpb = blockDim.x / cvec;
    const int cv = threadIdx.x % cvec, rl = threadIdx.x / cvec;
    const int sample = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(R, r_begin + rows_per_block);
    int stride;
    const __half* base = gn_src(x, x2, C, C1, cv, (size_t)sample * R, stride);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    // four rows of a thread are requested together (one load per trip left the pass waiting out a memory latency per row); the
    // sums take the rows in the same order
    constexpr int UN = 4;
    int r = r_begin + rl;
    for (; r + (UN - 1) * rpb < r_end; r += UN * rpb) {
        half8 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = *(const half8*)(base + (size_t)(r + u * rpb) * stride);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[u][e];
                s[e] += f;
                q[e] += f * f;
            }
        }
    }
    for (; r < r_end; r += rpb) {
        half8 v = *(const half8*)(base + (size_t)r * stride);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e];
            s[e] += f;
            q[e] += f * f;
        }
    }
    float* my = red + threadIdx.x * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { my[e] = s[e]; my[8 + e] = q[e]; }
    __syncthreads();
    // threads 0..31: one group each; sum its channels over the rpb row-lanes
    if (threadIdx.x < 32) {
        const int g = threadIdx.x, cpg = C / 32;
        float gs = 0.f, gq = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            int v = c >> 3, e = c & 7;
            for (int k = 0; k < rpb; ++k) {
                const float* o = red + (k * cvec + v) * 16;
                gs += o[e];
                gq += o[8 + e];
            }
        }
        // per-chunk partial (summed in a fixed order by k_gn_apply: bitwise reproducible, no atomics)
        float* dst = stats + (((size_t)sample * gridDim.x + blockIdx.x) * 32 + g) * 2;
        dst[0] = gs;
        dst[1] = gq;
    }
}

// Fixed-order reduction of the per-chunk partials -> (mean, rstd) per (sample, group).
// One 1024-thread block per sample: 32 threads per group stride over the chunks, then a fixed
// 5-step shuffle tree (bitwise reproducible).
__global__ void __launch_bounds__(1024) k_gn_finalize(const float* __restrict__ partial, int chunks, float inv_n, float eps,
                                                     float* __restrict__ mr) {
    const int sample = blockIdx.x;
    const int g = threadIdx.x >> 5, sub = threadIdx.x & 31;
    float s1 = 0.f, s2 = 0.f;
    for (int ch = sub; ch < chunks; ch += 32) {
        const float* st = partial + (((size_t)sample * chunks + ch) * 32 + g) * 2;
        s1 += st[0];
        s2 += st[1];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, 32);
        s2 += __shfl_down(s2, o, 32);
    }
    if (sub == 0) {
        float mean = s1 * inv_n;
        float var = fmaxf(s2 * inv_n - mean * mean, 0.0f);
        mr[((size_t)sample * 32 + g) * 2] = mean;
        mr[((size_t)sample * 32 + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

// (mean, rstd) per (sample, group) from the partial sums the PRODUCER of the activation left behind (GemmParams::gn_part, the lean
// epilogue of the persistent contraction kernels; round 6): part[((m / 32) * 2 + q) * U + n / 10] over 32-row blocks and 10-column
// units.  Two-source input: units [0, U1) come from part1, the rest from part2 (a group may straddle the two tensors: 1280 + 640
// channels are groups of 60).  One block per (group, sample); thread t folds items t, t + 256, ... of the group's
// (row block, unit) list in order, then a fixed tree: bitwise reproducible.  No pass over the activation.
__global__ void __launch_bounds__(256) k_gn_finalize_parts(const float* __restrict__ part1, int U1, const float* __restrict__ part2, int U2,
                                                           int rb_per_sample, int upg, float inv_n, float eps, float* __restrict__ mr) {
    const int g = blockIdx.x, sample = blockIdx.y;
    const int items = rb_per_sample * upg;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int rbl = i / upg, u = g * upg + (i - rbl * upg);
        const size_t rb = (size_t)sample * rb_per_sample + rbl;
        const float* P = u < U1 ? part1 + rb * 2 * (size_t)U1 + u : part2 + rb * 2 * (size_t)U2 + (u - U1);
        const int U = u < U1 ? U1 : U2;
        s1 += P[0];
        s2 += P[U];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, 64);
        s2 += __shfl_down(s2, o, 64);
    }
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = s1; red[(threadIdx.x >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = (red[0] + red[2]) + (red[4] + red[6]);
        s2 = (red[1] + red[3]) + (red[5] + red[7]);
        const float mean = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - mean * mean, 0.0f);
        mr[((size_t)sample * 32 + g) * 2] = mean;
        mr[((size_t)sample * 32 + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

template <bool SILU>
__global__ void k_gn_apply(const __half* __restrict__ x, const __half* __restrict__ x2, int C1, __half* __restrict__ y, int R,
                           int C, int rows_per_block,
                           const float* __restrict__ stats, const __half* __restrict__ gamma,
                           const __half* __restrict__ beta, float eps) {
    const int cvec = C >> 3;
    const int rpb = blockDim.x / cvec;
    const int cv = threadIdx.x % cvec, rl = threadIdx.x / cvec;
    const int sample = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(R, r_begin + rows_per_block);
    const int cpg = C / 32;
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int c = cv * 8 + e;
        int g = c / cpg;
        const float* st = stats + ((size_t)sample * 32 + g) * 2;   // (mean, rstd) from k_gn_finalize
        float mean = st[0], rstd = st[1];
        float ga = (float)((const _Float16*)gamma)[c], be = (float)((const _Float16*)beta)[c];
        a[e] = rstd * ga;
        b[e] = be - mean * rstd * ga;
    }
    const size_t off = (size_t)sample * R * C + (size_t)cv * 8;
    int stride;
    const __half* src = gn_src(x, x2, C, C1, cv, (size_t)sample * R, stride);
    auto apply = [&](half8 v) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * a[e] + b[e];
            if (SILU) f = f * __builtin_amdgcn_rcpf(1.0f + __expf(-f));     // v_rcp_f32 (1 ulp; the result is rounded to fp16): the IEEE division was ~10 of the element's ~16 vector instructions, and the pass is not far from vector-bound
            o[e] = (_Float16)f;
        }
        return o;
    };
    // four rows of a thread are requested together (one load per trip: a memory latency per row)
    constexpr int UN = 4;
    int r = r_begin + rl;
    for (; r + (UN - 1) * rpb < r_end; r += UN * rpb) {
        half8 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = *(const half8*)(src + (size_t)(r + u * rpb) * stride);
#pragma unroll
        for (int u = 0; u < UN; ++u) *(half8*)(y + off + (size_t)(r + u * rpb) * C) = apply(v[u]);
    }
    for (; r < r_end; r += rpb) *(half8*)(y + off + (size_t)r * C) = apply(*(const half8*)(src + (size_t)r * stride));
}

// ---------------------------------------------------------------- LayerNorm
// LPR lanes cooperate on one row (64/LPR rows per wavefront), each lane holding up to LN_MAXV vectors of
// 8 channels: C = 320 -> 8 lanes x 5 vectors, every lane busy (one wavefront per row would idle 24 of 64).
constexpr int LN_MAXV = 8;

// NV > 0: the row is EXACTLY NV vectors per lane (C = 8 * LPR * NV: 640 / 1280 / 320 of the UNet are 5 x 16 / 32 / 8) - no per-vector
// guard, so a lane's loads are issued together; behind `if (cv < cvec)` branches each load waits for the one before it (hipcc keeps
// loads inside their exec-mask regions: five latencies in a row per pass).  NV == 0: any C, guarded.  Same arithmetic, same order.
template <int LPR, int NV>
__global__ void __launch_bounds__(256) k_layernorm(const __half* __restrict__ x, __half* __restrict__ y,
                                                   __half* __restrict__ xsum, const __half* __restrict__ addvec,
                                                   int rows_per_vec, long long M, int C,
                                                   const __half* __restrict__ gamma, const __half* __restrict__ beta,
                                                   float eps) {
    constexpr int RPW = 64 / LPR;
    constexpr bool EXACT = NV > 0;
    constexpr int KV = EXACT ? NV : LN_MAXV;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < M;
    if (!live) row = M - 1;                     // keep the lanes in the shuffles
    const int cvec = C >> 3;
    const __half* src = x + row * C;
    const __half* add = addvec ? addvec + (row / rows_per_vec) * C : nullptr;
    float v[KV][8];
    float s = 0.f;
    if constexpr (EXACT) {
        half8 h[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) h[k] = *(const half8*)(src + (sub + LPR * k) * 8);
        if (add) {
            half8 a[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) a[k] = *(const half8*)(add + (sub + LPR * k) * 8);
#pragma unroll
            for (int k = 0; k < KV; ++k) {
#pragma unroll
                for (int e = 0; e < 8; ++e) h[k][e] = h[k][e] + a[k][e];   // fp16 add, as the reference's tensor add
            }
            if (xsum && live) {
#pragma unroll
                for (int k = 0; k < KV; ++k) *(half8*)(xsum + row * C + (sub + LPR * k) * 8) = h[k];
            }
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[k][e] = (float)h[k][e]; s += v[k][e]; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            int cv = sub + LPR * k;
            if (cv < cvec) {
                half8 h = *(const half8*)(src + cv * 8);
                if (add) {
                    half8 a = *(const half8*)(add + cv * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = h[e] + a[e];   // fp16 add, as the reference's tensor add
                    if (xsum && live) *(half8*)(xsum + row * C + cv * 8) = h;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[k][e] = (float)h[e]; s += v[k][e]; }
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        int cv = sub + LPR * k;
        if (EXACT || cv < cvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = v[k][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    if constexpr (EXACT) {
        half8 g[KV], b[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) { g[k] = *(const half8*)(gamma + (sub + LPR * k) * 8); b[k] = *(const half8*)(beta + (sub + LPR * k) * 8); }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[k][e] - mean) * rstd * (float)g[k][e] + (float)b[k][e]);
            *(half8*)(y + row * C + (sub + LPR * k) * 8) = o;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            int cv = sub + LPR * k;
            if (cv < cvec) {
                half8 g = *(const half8*)(gamma + cv * 8), b = *(const half8*)(beta + cv * 8), o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[k][e] - mean) * rstd * (float)g[e] + (float)b[e]);
                *(half8*)(y + row * C + cv * 8) = o;
            }
        }
    }
}

// ---------------------------------------------------------------- GEGLU
__global__ void __launch_bounds__(256) k_geglu(const __half* __restrict__ x, __half* __restrict__ y, long long M, int D) {
    const int dvec = D >> 3;
    const long long total = M * dvec;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long row = i / dvec;
        int cv = (int)(i - row * dvec);
        const __half* src = x + row * 2 * D + cv * 8;
        half8 hv = *(const half8*)src, gv = *(const half8*)(src + D), o;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {       // the same packed GELU as the contraction kernels' fused gates (common.h)
            const syn3r_f2 r = (syn3r_f2){(float)hv[e], (float)hv[e + 1]} * gelu_pk((syn3r_f2){(float)gv[e], (float)gv[e + 1]});
            o[e] = (_Float16)r.x; o[e + 1] = (_Float16)r.y;
        }
        *(half8*)(y + row * D + cv * 8) = o;
    }
}

int gn_threads(int C) {
    int cvec = C / 8;
    if (cvec > 1024) return -1;
    int rpb = 256 / cvec;
    if (rpb < 1) rpb = 1;
    return rpb * cvec;
}

}  // namespace

namespace {
// launch geometry shared by the workspace query and the launcher: ~2048 blocks in total
void gn_geometry(int samples, int rows, int& chunks, int& rows_per_block) {
    chunks = (2048 + samples - 1) / samples;
    rows_per_block = (rows + chunks - 1) / chunks;
    if (rows_per_block < 8) rows_per_block = 8;
    chunks = (rows + rows_per_block - 1) / rows_per_block;
}
}  // namespace

extern "C" size_t syn3r_groupnorm_workspace_bytes(int samples, int rows) {
    if (!SYN3R_DIM_OK(samples) || !SYN3R_DIM_OK(rows)) return 0;
    int chunks, rpb;
    gn_geometry(samples, rows, chunks, rpb);
    return (size_t)samples * chunks * 32 * 2 * sizeof(float) + (size_t)samples * 32 * 2 * sizeof(float);
}

namespace {
int groupnorm_launch(const void* x, const void* x2, int C1, void* y, int samples, int rows, int C, const void* gamma,
                     const void* beta, float eps, int silu, void* workspace, size_t workspace_bytes, void* stream_,
                     const float* part1 = nullptr, const float* part2 = nullptr) {
    SYN3R_REQUIRE(x && y && gamma && beta, "groupnorm: null tensor");
    SYN3R_REQUIRE(SYN3R_DIM_OK(samples) && SYN3R_DIM_OK(rows) && SYN3R_DIM_OK(C) && C % 32 == 0 && C % 8 == 0, "groupnorm: bad sizes samples=%d rows=%d C=%d",
                  samples, rows, C);
    SYN3R_REQUIRE(x2 ? (C1 > 0 && C1 < C && C1 % 8 == 0) : C1 == C, "groupnorm: bad channel split C1=%d of C=%d", C1, C);
    int threads = gn_threads(C);
    SYN3R_REQUIRE(threads > 0, "groupnorm: C=%d too large", C);
    size_t need = syn3r_groupnorm_workspace_bytes(samples, rows);
    if (!workspace || workspace_bytes < need) {
        set_error("groupnorm: workspace %zu < %zu", workspace_bytes, need);
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    int chunks, rows_per_block;
    gn_geometry(samples, rows, chunks, rows_per_block);
    dim3 grid(chunks, samples);
    size_t lds = (size_t)threads * 16 * sizeof(float);
    float* partial = (float*)workspace;
    float* meanrstd = partial + (size_t)samples * chunks * 64;
    if (part1) {          // the producers' partial sums: no statistics pass
        const int cpg = C / 32;
        SYN3R_REQUIRE(rows % 32 == 0 && cpg % 10 == 0 && C1 % 10 == 0 && (!x2 || part2), "groupnorm_pre: needs rows %% 32 == 0 and whole 10-channel units (rows=%d C=%d C1=%d)", rows, C, C1);
        SYN3R_REQUIRE(((uintptr_t)part1 | (uintptr_t)part2) % 4 == 0, "groupnorm_pre: misaligned partial sums");
        SYN3R_LAUNCH(k_gn_finalize_parts, dim3(32, samples), dim3(256), 0, stream, part1, C1 / 10, part2, (C - C1) / 10, rows / 32, cpg / 10,
                     1.0f / ((float)rows * (float)cpg), eps, meanrstd);
    } else {
        SYN3R_LAUNCH(k_gn_stats, grid, dim3(threads), lds, stream, (const __half*)x, (const __half*)x2, C1, rows, C, rows_per_block,
                     partial);
        SYN3R_LAUNCH(k_gn_finalize, dim3(samples), dim3(1024), 0, stream, (const float*)partial, chunks,
                     1.0f / ((float)rows * (float)(C / 32)), eps, meanrstd);
    }
    if (silu)
        SYN3R_LAUNCH(k_gn_apply<true>, grid, dim3(threads), 0, stream, (const __half*)x, (const __half*)x2, C1, (__half*)y, rows,
                     C, rows_per_block, (const float*)meanrstd, (const __half*)gamma, (const __half*)beta, eps);
    else
        SYN3R_LAUNCH(k_gn_apply<false>, grid, dim3(threads), 0, stream, (const __half*)x, (const __half*)x2, C1, (__half*)y, rows,
                     C, rows_per_block, (const float*)meanrstd, (const __half*)gamma, (const __half*)beta, eps);
    SYN3R_LAUNCH_CHECK("groupnorm launch");
    return SYN3R_OK;
}
}  // namespace

extern "C" int syn3r_groupnorm_f16(const void* x, void* y, int samples, int rows, int C, const void* gamma,
                                   const void* beta, float eps, int silu, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
    return groupnorm_launch(x, nullptr, C, y, samples, rows, C, gamma, beta, eps, silu, workspace, workspace_bytes, stream_);
}

extern "C" int syn3r_groupnorm_2src_f16(const void* x1, int C1, const void* x2, int C2, void* y, int samples, int rows,
                                        const void* gamma, const void* beta, float eps, int silu, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
    SYN3R_REQUIRE(x2 != nullptr && SYN3R_DIM_OK(C1) && SYN3R_DIM_OK(C2), "groupnorm_2src: second source missing or bad widths C1=%d C2=%d", C1, C2);
    return groupnorm_launch(x1, x2, C1, y, samples, rows, C1 + C2, gamma, beta, eps, silu, workspace, workspace_bytes, stream_);
}

extern "C" int syn3r_groupnorm_pre_f16(const void* x1, int C1, const void* part1, const void* x2, int C2, const void* part2, void* y,
                                       int samples, int rows, const void* gamma, const void* beta, float eps, int silu,
                                       void* workspace, size_t workspace_bytes, void* stream_) {
    SYN3R_REQUIRE(part1 != nullptr && SYN3R_DIM_OK(C1) && (x2 == nullptr ? C2 == 0 : (SYN3R_DIM_OK(C2) && part2 != nullptr)),
                  "groupnorm_pre: partial sums missing or bad widths C1=%d C2=%d", C1, C2);
    return groupnorm_launch(x1, x2, C1, y, samples, rows, C1 + C2, gamma, beta, eps, silu, workspace, workspace_bytes, stream_,
                            (const float*)part1, (const float*)part2);
}

extern "C" int syn3r_layernorm_f16(const void* x, void* y, void* xsum, const void* addvec, int rows_per_vec,
                                   long long M, int C, const void* gamma, const void* beta, float eps, void* stream_) {
    SYN3R_REQUIRE(x && y && gamma && beta, "layernorm: null tensor");
    SYN3R_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAXV, "layernorm: bad sizes M=%lld C=%d", M, C);
    SYN3R_REQUIRE(!addvec || rows_per_vec > 0, "layernorm: rows_per_vec required with addvec");
    const int cvec = C / 8;
    int lpr = 8;
    while (lpr < 64 && (cvec + lpr - 1) / lpr > 5) lpr *= 2;       // <= 5 vectors per lane where possible
    long long rows_per_block = 4 * (64 / lpr);
    long long blocks = (M + rows_per_block - 1) / rows_per_block;
    SYN3R_REQUIRE(blocks < (1ll << 31), "layernorm: too many rows");
    const bool exact5 = cvec == 5 * lpr;                           // 320 / 640 / 1280 / 2560: unguarded, batched loads
#define LN_LAUNCH(L_, NV_)                                                                                              \
    SYN3R_LAUNCH((k_layernorm<L_, NV_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const __half*)x,   \
                 (__half*)y, (__half*)xsum, (const __half*)addvec, rows_per_vec, M, C, (const __half*)gamma,              \
                 (const __half*)beta, eps)
#define LN_PICK(L_) do { if (exact5) LN_LAUNCH(L_, 5); else LN_LAUNCH(L_, 0); } while (0)
    if (lpr == 8) LN_PICK(8);
    else if (lpr == 16) LN_PICK(16);
    else if (lpr == 32) LN_PICK(32);
    else LN_PICK(64);
#undef LN_PICK
#undef LN_LAUNCH
    SYN3R_LAUNCH_CHECK("layernorm launch");
    return SYN3R_OK;
}

extern "C" int syn3r_geglu_f16(const void* x, void* y, long long M, int D, void* stream_) {
    SYN3R_REQUIRE(x && y, "geglu: null tensor");
    SYN3R_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "geglu: bad sizes M=%lld D=%d", M, D);
    long long total = M * (D / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SYN3R_LAUNCH(k_geglu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const __half*)x, (__half*)y, M, D);
    SYN3R_LAUNCH_CHECK("geglu launch");
    return SYN3R_OK;
}


// ---------------------------------------------------------------------------------------------
// Row softmax (fp16 storage, fp32 arithmetic) and the temporal decoder's 3-channel output convolution:
// the two element-wise pieces of the temporal-decoder VAE that the UNet kernels lack.
namespace {

__global__ void __launch_bounds__(256) k_softmax_rows(const __half* __restrict__ x, __half* __restrict__ y, int N,
                                                      long long ld, float scale) {
    const long long row = blockIdx.x;
    const __half* src = x + row * ld;
    __half* dst = y + row * ld;
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = -3.0e38f;
    for (int i = threadIdx.x * 8; i < N; i += 256 * 8) {
        if (i + 8 <= N) {
            half8 v = *(const half8*)(src + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[e]);
        } else {
            for (int e = 0; i + e < N; ++e) mx = fmaxf(mx, __half2float(src[i + e]));
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;   // scale > 0
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x * 8; i < N; i += 256 * 8) {
        const int n = min(8, N - i);
        for (int e = 0; e < n; ++e) sum += __expf(__half2float(src[i + e]) * scale - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x * 8; i < N; i += 256 * 8) {
        if (i + 8 <= N) {
            half8 v = *(const half8*)(src + i), o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)(__expf((float)v[e] * scale - mx) * inv);
            *(half8*)(dst + i) = o;
        } else {
            for (int e = 0; i + e < N; ++e) dst[i + e] = __float2half(__expf(__half2float(src[i + e]) * scale - mx) * inv);
        }
    }
}

struct TimeConvW { float w[27]; float b[3]; };

__global__ void __launch_bounds__(256) k_time_conv_out(const __half* __restrict__ x, long long ldx, TimeConvW tw,
                                                       float* __restrict__ out, int F, long long HW, long long total) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (b*F + f)*HW + p
    if (i >= total) return;
    const long long bf = i / HW, pix = i - bf * HW;
    const int f = (int)(bf % F);
    float acc[3] = {tw.b[0], tw.b[1], tw.b[2]};
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int ff = f + dt - 1;
        if (ff < 0 || ff >= F) continue;
        const __half* src = x + (i + (long long)(dt - 1) * HW) * ldx;
        const float c0 = __half2float(src[0]), c1 = __half2float(src[1]), c2 = __half2float(src[2]);
#pragma unroll
        for (int co = 0; co < 3; ++co)
            acc[co] += tw.w[(co * 3 + 0) * 3 + dt] * c0 + tw.w[(co * 3 + 1) * 3 + dt] * c1 + tw.w[(co * 3 + 2) * 3 + dt] * c2;
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) out[(bf * 3 + co) * HW + pix] = acc[co];
}

}  // namespace

extern "C" int syn3r_softmax_rows_f16(const void* x, void* y, long long M, int N, long long ld, float scale, void* stream) {
    SYN3R_REQUIRE(M > 0 && N > 0 && ld >= N, "softmax_rows: bad sizes M=%lld N=%d ld=%lld", M, N, ld);
    SYN3R_REQUIRE(x && y, "softmax_rows: null pointer");
    SYN3R_REQUIRE(scale > 0.0f, "softmax_rows: scale must be positive");
    SYN3R_REQUIRE(ld % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "softmax_rows: rows must be 16-byte aligned");
    SYN3R_REQUIRE(M < (1ll << 31), "softmax_rows: too many rows");
    SYN3R_LAUNCH(k_softmax_rows, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, (const __half*)x, (__half*)y, N, ld,
                 scale);
    SYN3R_LAUNCH_CHECK("softmax_rows launch");
    return SYN3R_OK;
}

extern "C" int syn3r_time_conv_out(const void* x, long long ldx, const float* w, const float* bias, float* out, int B,
                                   int F, long long HW, void* stream) {
    SYN3R_REQUIRE(SYN3R_DIM_OK(B) && SYN3R_DIM_OK(F) && HW > 0 && HW <= SYN3R_DIM_MAX && ldx >= 3, "time_conv_out: bad sizes");
    SYN3R_REQUIRE(x && w && bias && out, "time_conv_out: null pointer");
    TimeConvW tw;
    for (int i = 0; i < 27; ++i) tw.w[i] = w[i];     // host pointers: 30 floats travel as kernel arguments
    for (int i = 0; i < 3; ++i) tw.b[i] = bias[i];
    const long long total = (long long)B * F * HW;
    SYN3R_REQUIRE((total + 255) / 256 < (1ll << 31), "time_conv_out: too many pixels");
    SYN3R_LAUNCH(k_time_conv_out, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                 (const __half*)x, ldx, tw, out, F, HW, total);
    SYN3R_LAUNCH_CHECK("time_conv_out launch");
    return SYN3R_OK;
}
