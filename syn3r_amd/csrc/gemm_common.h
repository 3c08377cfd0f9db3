// Shared pieces of the contraction kernels: per-device launch attributes, tile constants, GemmParams and the staged epilogue.
// Included by gemm.hip inside its anonymous namespace (one translation unit; the kernels share GemmParams, the epilogues and the
// LDS-DMA typedefs of gemm_common.h / gemm_dma.h).

// (DevOnce / set_max_lds - the per-device MaxDynamicSharedMemorySize attribute - live in common.h)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // 16-byte staging register (native vector: stays in VGPRs)

constexpr int BN = 160, BK = 64;
constexpr int WM = 64, WN = 80;            // per-wavefront output tile
constexpr int TM = WM / 16, TN = WN / 16;  // 4 x 5 MFMA tiles
constexpr int B_TILE = BN * BK;            // halfs
// Two block shapes: BM = 256 (512 threads, 1 block/CU) and BM = 128 (256 threads, 2 independent
// blocks/CU whose barrier phases de-synchronise: while one block stages / waits, the other issues MFMAs).
constexpr int EPI_LD = 88;                 // padded row stride (halfs) of the epilogue staging tile

enum { MODE_DENSE = 0, MODE_CONV2D = 1, MODE_TCONV = 2 };

// Which row of the per-sample vector table row m adds.  rows_per_vec > 0: one vector per block of rows.
// rows_per_vec = -P: vector m mod P (the batch-interleaved context of the temporal cross-attention,
// transformer_temporal.py:310-317, for a batch of P); with rv_group = G > 0 the rows come in groups of G that each
// emulate a SEPARATE batch-of-P call: group g = m / G reads vectors g * P + m mod P (two CFG passes in one launch).
__device__ __forceinline__ int rowvec_index(int m, int rows_per_vec, int rv_group) {
    if (rows_per_vec > 0) return m / rows_per_vec;
    const int P = -rows_per_vec;
    return (rv_group > 0 ? (m / rv_group) * P : 0) + m % P;
}

struct GemmParams {
    const __half* A; long long lda;       // dense: row stride; conv: unused (NHWC dense)
    const __half* W;                      // [N][K], K contiguous
    __half* out; long long ldc;
    const __half* bias;                   // [N] or null
    const __half* rowvec; long long ldrv; int rows_per_vec;   // [M/rows_per_vec][ldrv] or null
    int rv_group;                         // rows_per_vec < 0 only: rows per context group (0 = one group), see rowvec_index
    const __half* residual; long long ldr;
    const __half* aux; long long ldaux;
    float s_acc, s_res, s_aux;
    int M, N, K;
    // conv geometry (NHWC): output Ho x Wo, input Hi x Wi, Cin channels (K = taps * Cin)
    int Ho, Wo, Hi, Wi, Cin, stride, ups, pad;   // pad: zero rows/cols before the first pixel (1, or 0 for the (0,1,0,1) pad)
    // temporal conv: F frames of HW rows each (row = (b*F + f)*HW + p)
    int F, HW;
    // k_gemm_dmap<MODE_TCONV>: tile order with the FRAME index minor (tc_pb = HW / 256 pixel blocks per frame, tc_nf = M / HW
    // frames; 0 = rows in memory order).  The three taps of a row tile read the same 256 pixels of frames f - 1, f, f + 1: in
    // memory order those are 36 tiles apart at level 0 and every tap streams its rows from beyond the L2 (counted HBM bytes
    // 2.14x the algorithmic ones, profiles/r04/traffic.json); with the frames of one pixel block consecutive, the 16 row tiles an
    // XCD holds at a time are 16 frames of that block and two of a tile's three A taps are another tile's rows (L2 hits).
    int tc_pb, tc_nf;
    // GEGLU epilogue: W rows are packed per 160-row tile as [80 hidden | 80 gate]; out has geglu_D columns
    int geglu_D;
    // A-tiled layout of a [M, D] matrix (the feed-forward's gated hidden activation, written by the GEGLU kernel and
    // read once as the A operand of the second projection): [ceil(M/128)][D/64][128 rows][64 columns], i.e. the
    // 16 KB image of every (128-row block, 64-wide k-tile) is one contiguous run: the second projection streams its
    // A operand as whole tile images (2..5 % faster inside the UNet than from rows at a 2.5-10 KB pitch), and the
    // writes of a wavefront stay inside two 8 KB windows.
    int out_tiled;                        // the kernel writes `out` in that layout (ldc unused)
    int out_nt;                           // non-temporal output stores (see OUT_STORE)
    int a_tiled;                          // the kernel reads A in that layout (lda unused; dense mode only)
    // Two-source A (k_gemm_widep only): columns [0, K1) of a row come from A (stride lda), columns [K1, K) from A2
    // (stride lda2) - the channel concatenation [A | A2] the up blocks' shortcut projection reads is never written.
    const __half* A2; long long lda2; int K1;     // A2 = null: one source
    // k_ffn320r only: the residual operand is residual + res_add[row / res_add_rpv] (an fp16 tensor add, rounded as such)
    const __half* res_add; int res_add_rpv;
    // VGG-style activation options of the GENERAL epilogue (gemm_epilogue; the convolution kernels use it):
    int band;                             // persistent 256 x 320 kernels: tile columns per band of the tile order (band_width())
    // split-K (k_gemm_dma<MODE, 256>, implicit-GEMM convolutions whose tile grid leaves most CUs idle): the K range is cut into
    // `ksplit` equal parts, one block per (tile, part) writes its fp32 partial tile to split_ws
    // [ksplit][M][N]; k_splitk_finish sums the parts in order and applies the epilogue (launch_dma)
    int ksplit; float* split_ws;
    int relu;                             // result = max(result, 0)
    const __half* relu_mask;              // [M][ldc]: result zeroed where mask <= 0 (ReLU backward: grad * (activation > 0))
    // GroupNorm partial sums of the OUTPUT (round 6; the lean epilogue of the persistent kernels, gn_tile_stats in gemm_wide.h):
    // gn_part[((m / 32) * 2 + q) * gn_units + n / 10], q = 0: sum of x, q = 1: sum of x^2 over rows [32 rb, 32 rb + 32) and columns
    // [10 u, 10 u + 10) of the fp16 results as stored.  Every GroupNorm(32) of the UNet has 10, 20, 30, 40, 60 or 80 channels per
    // group (C = 320 .. 2560), so any consumer folds whole units; the statistics pass over the activation (k_gn_stats) is not
    // launched (resnet.py:272,286,574,588, transformer_temporal.py:235).  null: not written.  Needs M % 32 == 0 and N % 80 == 0.
    float* gn_part; int gn_units;
    // k_gemm_z convolution modes (round 6): bytes of the input tensor, the extent of the buffer resource the A pieces are fetched
    // through - a padded chunk is an out-of-range offset and the LDS-DMA writes zeros (tools/ubench/buffer_lds_oob.hip)
    unsigned a_bytes;
};

// element offset of (row m, column d) in the A-tiled layout of a matrix with D columns
__device__ __forceinline__ long long tiled_off(int m, int d, int D) {
    return ((long long)(m >> 7) * (D >> 6) + (d >> 6)) * 8192 + (m & 127) * 64 + (d & 63);
}

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 3); }

// Shared epilogue: acc (+bias +rowvec) -> fp16 through LDS -> row-contiguous 16-byte stores (+residual, +aux),
// or the GEGLU gate.  Must be entered by every wavefront of the block after the last LDS tile read.
// PREFETCH_RES: 1 = residual rows requested before the accumulators are staged (most latency hidden);
// 2 = requested after the staging writes, when the accumulators are dead (wide tile: registers are short)
// Output stores.  nt: non-temporal (streamed past the L2): for the feed-forward's gated hidden activation when it is
// larger than the memory-side cache (-4..6 % on that projection inside the UNet); on outputs that the next kernel
// reads back at once (qkv, proj_in) non-temporal stores cost the PRODUCER 8..25 %, so it is opt-in per call.
#define OUT_STORE(ptr, val) do { if (p.out_nt) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)

template <int PREFETCH_RES = 1, bool RES_ADD = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, float4v (&acc)[TM][TN], char* smem_raw, int lane,
                                              int wv, int wm, int wn, int m0, int n0, int tile_n) {
    const int fr = lane & 15, fq = lane >> 4;
    __half* st = (__half*)smem_raw + wv * (WM * EPI_LD);
    const int gm0 = m0 + wm * WM, gn0 = n0 + wn * WN;
    // The MFMAs are issued with the weight fragment as the A operand, so acc[i][j][r] is
    // C[row i*16 + (lane&15)][col j*16 + (lane>>4)*4 + r]: four CONSECUTIVE output columns per lane ->
    // one 8-byte LDS store per accumulator tile (20 per lane).
    typedef _Float16 half4e __attribute__((ext_vector_type(4)));
    // The residual rows this lane will add in the store loop are requested NOW, so their HBM latency hides
    // behind the accumulator -> LDS staging and the block barrier (10 x 16 B per lane; the MFMA fragments
    // are dead here, so the registers are free).
    half8 res[WM * (WN / 8) / 64];
    half8 radd[RES_ADD ? WM * (WN / 8) / 64 : 1];      // RES_ADD: the vector the residual gets added first (requested with it, added where it is used)
    auto prefetch_residual = [&]() {
#pragma unroll
        for (int it = 0; it < WM * (WN / 8) / 64; ++it) {
            const int q = lane + it * 64;
            int row = q / (WN / 8), ch = q - row * (WN / 8);
            int m = gm0 + row, n = gn0 + ch * 8;
            if (m < p.M && n + 8 <= p.N) {
                res[it] = *(const half8*)(p.residual + (long long)m * p.ldr + n);
                if constexpr (RES_ADD) { if (p.res_add) radd[it] = *(const half8*)(p.res_add + (long long)(m / p.res_add_rpv) * p.N + n); }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) res[it][e] = (m < p.M && n + e < p.N) ? ((const _Float16*)p.residual)[(long long)m * p.ldr + n + e] : (_Float16)0.f;
            }
        }
    };
    if (PREFETCH_RES == 1 && p.residual && p.geglu_D <= 0) prefetch_residual();
    float bias4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = gn0 + j * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[j][r] = 0.0f;
        if (p.bias) {
            if (n + 4 <= p.N) {
                half4e b = *(const half4e*)(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) bias4[j][r] = (float)b[r];
            } else {
                for (int r = 0; r < 4; ++r) if (n + r < p.N) bias4[j][r] = __half2float(p.bias[n + r]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = i * 16 + fr;
        const int m = gm0 + row;
        const __half* rv = nullptr;
        if (p.rowvec && m < p.M) {
            const int vi = rowvec_index(m, p.rows_per_vec, p.rv_group);
            rv = p.rowvec + (long long)vi * p.ldrv;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = j * 16 + fq * 4;
            const int n = gn0 + col;
            float add[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) add[r] = bias4[j][r];
            if (rv) {
                if (n + 4 <= p.N) {
                    half4e t = *(const half4e*)(rv + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) add[r] += (float)t[r];
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) add[r] += __half2float(rv[n + r]);
                }
            }
            half4e o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = (acc[i][j][r] + add[r]) * p.s_acc;
                if (p.relu) v = fmaxf(v, 0.0f);
                o[r] = (_Float16)v;
            }
            *(half4e*)(st + row * EPI_LD + col) = o;
        }
    }
    if (PREFETCH_RES == 2 && p.residual && p.geglu_D <= 0) prefetch_residual();
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (p.geglu_D > 0) {
        // activations.py GEGLU.forward: hidden * gelu(gate).  Wave (wm,0) staged the 80 hidden columns and
        // (wm,1) the 80 gate columns of the same 64 rows (fp16, as the reference's projection output);
        // each of the two waves finishes 32 of those rows.
        const __half* hs = (const __half*)smem_raw + (wm * 2 + 0) * (WM * EPI_LD);
        const __half* gs = (const __half*)smem_raw + (wm * 2 + 1) * (WM * EPI_LD);
        const int nout0 = tile_n * WN;
        for (int q = lane; q < 32 * (WN / 8); q += 64) {
            int row = wn * 32 + q / (WN / 8), ch = q % (WN / 8);
            int m = gm0 + row, n = nout0 + ch * 8;
            if (m >= p.M || n >= p.geglu_D) continue;
            half8 hv = *(const half8*)(hs + row * EPI_LD + ch * 8);
            half8 gv = *(const half8*)(gs + row * EPI_LD + ch * 8);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const syn3r_f2 y = (syn3r_f2){(float)hv[e], (float)hv[e + 1]} * gelu_pk((syn3r_f2){(float)gv[e], (float)gv[e + 1]});
                o[e] = (_Float16)y.x; o[e + 1] = (_Float16)y.y;
            }
            if (p.out_tiled) {
                OUT_STORE((half8*)(p.out + tiled_off(m, n, p.geglu_D)), o);
            } else if (n + 8 <= p.geglu_D) {
                OUT_STORE((half8*)(p.out + (long long)m * p.ldc + n), o);
            } else {
                for (int e = 0; e < 8 && n + e < p.geglu_D; ++e) ((_Float16*)p.out)[(long long)m * p.ldc + n + e] = o[e];
            }
        }
        return;
    }
    // 64 rows x 10 chunks of 8 halfs per wavefront
    constexpr int NQ = WM * (WN / 8) / 64;   // 10 stores per lane
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
        const int q = lane + it * 64;
        int row = q / (WN / 8), ch = q - row * (WN / 8);
        int m = gm0 + row, n = gn0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        half8 v = *(const half8*)(st + row * EPI_LD + ch * 8);
        if (p.residual || p.aux) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
            if (p.residual) {
                half8 rr;
                if constexpr (PREFETCH_RES != 0) rr = res[it];
                else rr = *(const half8*)(p.residual + (long long)m * p.ldr + n);
                if constexpr (RES_ADD) { if (p.res_add) rr = rr + radd[it]; }          // fp16 tensor add
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += p.s_res * (float)rr[e];
            }
            if (p.aux) {
                half8 av = *(const half8*)(p.aux + (long long)m * p.ldaux + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += p.s_aux * (float)av[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)f[e];
        }
        if (p.relu_mask && n + 8 <= p.N) {
            const half8 mk = *(const half8*)(p.relu_mask + (long long)m * p.ldc + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.0f ? v[e] : (_Float16)0.0f;
        }
        if (p.out_tiled) {
            OUT_STORE((half8*)(p.out + tiled_off(m, n, p.N)), v);
        } else if (n + 8 <= p.N) {
            OUT_STORE((half8*)(p.out + (long long)m * p.ldc + n), v);
        } else {
            for (int e = 0; e < 8 && n + e < p.N; ++e) ((_Float16*)p.out)[(long long)m * p.ldc + n + e] = v[e];
        }
    }
}
