// fp16 MFMA GEMM / implicit-GEMM convolution for the SVD spatio-temporal UNet.
//
// One kernel serves every dense contraction of the UNet forward
// (thirdparty/diffusers/src/diffusers/models/unets/unet_spatio_temporal_condition.py:356-489):
//   MODE_DENSE   out[M,N] = A[M,K] . W[N,K]^T            nn.Linear, 1x1 Conv2d/Conv3d shortcuts
//   MODE_CONV2D  3x3 Conv2d on NHWC activations (stride 1/2, optional fused nearest-2x
//                upsample of the input)                   resnet.py:274,290, downsampling.py:116-148, upsampling.py:172-183
//   MODE_TCONV   (3,1,1) Conv3d over the frame axis       resnet.py:571-597
// with a fused epilogue
//   out = s_acc * (acc + bias[n] + rowvec[m / rows_per_vec, n]) + s_res * residual[m,n] + s_aux * aux[m,n]
// which covers bias, the time-embedding add (resnet.py:352), residual adds, the Sk=1
// cross-attention broadcast and the AlphaBlender mix (resnet.py:789-802).
//
// CDNA4 mapping.  All kernels: v_mfma_f32_16x16x32_f16 (f32 accumulate) with the weight fragment as the A operand, LDS
// images of XOR-swizzled 16-byte chunks (conflict-free ds_read_b128 fragment reads), output tiles through LDS so that
// stores are 16 bytes per lane and row-contiguous, blocks dealt to XCDs in contiguous chunks.  Kernel families, chosen
// per shape by launch_dma (DESIGN.md section 4 has the measurements behind every rule):
//   k_gemm_widep   persistent 256 x 320 tile (dense contractions whose tiles fill the CUs): LDS-DMA 2-stage ring,
//                  cross-tile prefetch, scalar addressing, lean epilogue, GEGLU gate in registers, two-source A
//   k_gemm_z       the same tile with a software-pipelined main loop (gemm_z.h): gated projections, 3x3 convolutions
//   k_ffn320r      FeedForward (GEGLU) for C = 320 in one kernel, hidden activation never leaves the CU
//   k_lnlin320     LayerNorm + stacked q / k / v projection for C = 320 in one kernel
//   k_gemm_dma     BM x 160 tile, LDS-DMA 3-stage ring (BM = 256, wavefronts 4-7 staggered; every convolution /
//                  temporal convolution and the K = 320 residual projections) or 2-stage (BM = 128, small grids)
//   k_gemm_skinny  M <= 16 rows (time embedding, folded cross-attention context)
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <type_traits>

using namespace syn3r;

namespace {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remembered per (kernel, device), so a host that
// drives several GPUs from one process gets the 160 KB LDS attribute on each of them.
struct DevOnce { std::atomic<unsigned long long> done{0}; };
inline int set_max_lds(DevOnce& once, const void* fn, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;       // devices beyond 63: set on every launch
    if (bit && (once.done.load(std::memory_order_acquire) & bit)) return SYN3R_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return check_hip(e, what);
    once.done.fetch_or(bit, std::memory_order_release);
    return SYN3R_OK;
}

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
    // k_ffn320 only: the residual operand is residual + res_add[row / res_add_rpv] (an fp16 tensor add, rounded as such)
    const __half* res_add; int res_add_rpv;
    // VGG-style activation options of the GENERAL epilogue (gemm_epilogue; the convolution kernels use it):
    int band;                             // persistent 256 x 320 kernels: tile columns per band of the tile order (band_width())
    // split-K (k_gemm_dma<MODE, 256>, implicit-GEMM convolutions whose tile grid leaves most CUs idle): the K range is cut into
    // `ksplit` equal parts, one block per (tile, part) writes its fp32 partial tile to split_ws
    // [ksplit][M][N]; k_splitk_finish sums the parts in order and applies the epilogue (launch_dma)
    int ksplit; float* split_ws;
    int relu;                             // result = max(result, 0)
    const __half* relu_mask;              // [M][ldc]: result zeroed where mask <= 0 (ReLU backward: grad * (activation > 0))
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

// ---------------------------------------------------------------------------------------------
// LDS-DMA pipelined variant (default).  global_load_lds (16 B per lane, per-lane source address = an
// im2col gather for the convolutions, a zero page for padding / out-of-range rows) writes straight into a
// 3-stage LDS ring; a counted s_waitcnt vmcnt leaves the next stage's DMA in flight across ONE raw
// s_barrier per k-tile, so two k-tiles (104 KB per CU) of loads are always outstanding and no VGPRs or
// ds_write instructions are spent on staging.  The LDS image is lane-linear per wave-instruction (8 rows x
// 128 B), so the XOR swizzle is applied to the per-lane SOURCE chunk and undone by the fragment reads.
// hipcc would put s_waitcnt vmcnt(0) in front of any ds_read it can see while a DMA is pending, so the
// fragment reads are inline asm (ds_read_b128 + counted lgkmcnt, operands tied through "+v").
__device__ __half g_zero_page[64];   // zero-initialised: source of padded chunks

constexpr int DMA_B_BYTES = BN * BK * 2;                  // 20480
// BM = 256: 512 threads, 3-stage ring (156 KB, one block per CU, two k-tiles of DMA in flight).
// BM = 128: 256 threads, 2-stage ring (72 KB, TWO blocks per CU): a block's prologue DMA latency and its
//           40-80 KB store tail (store-issue bound at ~10 B/clk/CU) are hidden behind the other block's MFMAs
//           instead of idling the CU; costs 1.4x the L2->LDS bytes per output row (B tile per 128 rows).

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

#define DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))

template <int MODE, int BM>
__global__ void __launch_bounds__(BM * 2, 2) k_gemm_dma(GemmParams p) {
    constexpr int DMA_STAGES = BM == 256 ? 3 : 2;
    constexpr int NWAVES = BM / 32;                          // 8 or 4
    constexpr int DMA_A_BYTES = BM * BK * 2;
    constexpr int DMA_STAGE_BYTES = DMA_A_BYTES + DMA_B_BYTES;
    constexpr int NB_MAX = (20 + NWAVES - 1) / NWAVES;       // B pieces per wavefront: 3 (8 waves) or 5 (4 waves)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const unsigned ntile = (unsigned)(tiles_m * tiles_n);
    // split-K: which part of the K range.  An XCD takes ONE K part (blocks b and b + 8 share an XCD: part = (b % 8) % S) and a
    // contiguous chunk of that part's tiles, so that it streams 1 / S of the weight panel and 1 / (8 / S) of the rows instead of the
    // whole panel (round 4's order gave every XCD both parts of its tiles: counted HBM bytes 4.1x the algorithmic ones on the
    // M = 4 032 launches, profiles/r04/traffic.json).  Speed only: any placement computes the same partial tiles.
    int sp = 0;
    unsigned bid;
    if (p.ksplit > 1 && (ntile * (unsigned)p.ksplit) % 8 == 0 && 8 % p.ksplit == 0) {
        const unsigned xcd = blockIdx.x % 8, k = blockIdx.x / 8, S = (unsigned)p.ksplit;
        sp = (int)(xcd % S);
        bid = (xcd / S) * (ntile * S / 8) + k;
    } else {
        sp = p.ksplit > 1 ? (int)(blockIdx.x / ntile) : 0;
        bid = xcd_remap(blockIdx.x - (unsigned)sp * ntile, ntile);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- DMA assignment: lane -> (row within an 8-row piece, destination slot); source chunk un-swizzled
    const int prow = lane >> 3;
    const int csrc = (lane & 7) ^ prow;                 // source 16-byte chunk that lands in slot (lane & 7)
    const __half* zero = g_zero_page;
    const __half* a_base[4];
    int a_n[4], a_y[4], a_x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + wv * 32 + i * 8 + prow;
        bool ok = m < p.M;
        int mc = ok ? m : p.M - 1;
        if constexpr (MODE == MODE_DENSE) {
            // A-tiled: rows past M exist in the padded last row block (never stored); the k advance is one tile image
            const int last_rb = (p.M + 127) >> 7, rb = (m >> 7) < last_rb ? (m >> 7) : last_rb - 1;
            a_base[i] = p.a_tiled ? p.A + (long long)rb * (p.K >> 6) * 8192 + (m & 127) * 64 + csrc * 8
                                  : p.A + (long long)mc * p.lda + csrc * 8;
            // two sources (split-K launches only, parts never straddle K1): the parts from k-tile K1 / BK on read A2
            if (p.A2 && p.ksplit > 1 && sp * (p.K / BK / p.ksplit) >= p.K1 / BK) a_base[i] = p.A2 + (long long)mc * p.lda2 + csrc * 8;
            a_n[i] = a_y[i] = a_x[i] = 0;
        } else if constexpr (MODE == MODE_CONV2D) {
            int hw = p.Ho * p.Wo;
            a_n[i] = mc / hw;
            int r = mc - a_n[i] * hw;
            a_y[i] = r / p.Wo;
            a_x[i] = r - a_y[i] * p.Wo;
            a_base[i] = p.A + csrc * 8;
        } else {
            a_y[i] = (mc / p.HW) % p.F;
            a_n[i] = a_x[i] = 0;
            a_base[i] = p.A + (long long)mc * p.Cin + csrc * 8;
        }
    }
    // B pieces issued by this wavefront (20 in total): 8 waves -> 3,3,3,3,2,2,2,2 ; 4 waves -> 5 each
    const int nb = NWAVES == 8 ? (wv < 4 ? 3 : 2) : 5;
    const int b_first = NWAVES == 8 ? (wv < 4 ? wv * 3 : 12 + (wv - 4) * 2) : wv * 5;
    const __half* b_base[NB_MAX];
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
        int n = n0 + (b_first + j) * 8 + prow;
        b_base[j] = (j < nb && n < p.N) ? p.W + (long long)n * p.K + csrc * 8 : nullptr;
    }
    const int cpb = (MODE == MODE_DENSE) ? 1 : p.Cin / BK;

    // Source addresses advance incrementally: stages are issued in k order, a k-tile inside one filter tap is
    // +128 bytes on every live row, and the full im2col arithmetic (64-bit multiplies, bounds tests) runs only
    // when the tap changes (every Cin/64 k-tiles).  Padded rows point at the zero page and do not advance.
    const __half* a_cur[4];
    int a_inc[4];                         // halfs per k-tile: BK for live rows, 0 for zero-page rows
    const __half* b_cur[NB_MAX];
    int b_inc[NB_MAX];
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) { b_cur[j] = b_base[j] ? b_base[j] : zero; b_inc[j] = b_base[j] ? BK : 0; }
    int tap_next = 0, c_left = 0;         // wave-uniform: next tap to set up, k-tiles left in the current tap
    auto setup_tap = [&](int tap) {
        if constexpr (MODE == MODE_CONV2D) {
            const int dy = tap / 3 - p.pad, dx = tap % 3 - p.pad;
            const int Hg = p.ups ? p.Hi * 2 : p.Hi, Wg = p.ups ? p.Wi * 2 : p.Wi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int yy = a_y[i] * p.stride + dy, xx = a_x[i] * p.stride + dx;
                bool ok = yy >= 0 && yy < Hg && xx >= 0 && xx < Wg;
                if (p.ups) { yy >>= 1; xx >>= 1; }
                long long off = (((long long)a_n[i] * p.Hi + yy) * p.Wi + xx) * p.Cin;
                a_cur[i] = ok ? a_base[i] + off : zero;
                a_inc[i] = ok ? BK : 0;
            }
        } else if constexpr (MODE == MODE_TCONV) {
            const int df = tap - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int ff = a_y[i] + df;
                bool ok = ff >= 0 && ff < p.F;
                a_cur[i] = ok ? a_base[i] + (long long)df * p.HW * p.Cin : zero;
                a_inc[i] = ok ? BK : 0;
            }
        }
    };
    if constexpr (MODE == MODE_DENSE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_cur[i] = a_base[i]; a_inc[i] = p.a_tiled ? 8192 : BK; }
    }

    auto issue_stage = [&](int kt, int buf) {     // must be called with kt = 0, 1, 2, ... in order
        char* st = smem_raw + buf * DMA_STAGE_BYTES;
        if constexpr (MODE != MODE_DENSE) {
            if (c_left == 0) { setup_tap(tap_next); ++tap_next; c_left = cpb; }
            --c_left;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)a_cur[i], (lds_void_t*)(st + (wv * 4 + i) * 1024), 16, 0, 0);
            a_cur[i] += a_inc[i];
        }
#pragma unroll
        for (int j = 0; j < NB_MAX; ++j) {
            if (j < nb) {
                __builtin_amdgcn_global_load_lds((gbl_void_t*)b_cur[j], (lds_void_t*)(st + DMA_A_BYTES + (b_first + j) * 1024), 16, 0, 0);
                b_cur[j] += b_inc[j];
            }
        }
    };

    float4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int nkt = p.ksplit > 1 ? p.K / BK / p.ksplit : p.K / BK;
    if (p.ksplit > 1) {                  // this block's part starts at k-tile sp * nkt, possibly inside a filter tap
        const int kt0 = sp * nkt;
#pragma unroll
        for (int j = 0; j < NB_MAX; ++j) b_cur[j] += (long long)kt0 * b_inc[j];
        if constexpr (MODE == MODE_DENSE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a_cur[i] += (long long)(kt0 - ((p.A2 && kt0 >= p.K1 / BK) ? p.K1 / BK : 0)) * a_inc[i];
        } else {
            const int rem = kt0 % cpb;   // k-tiles of the tap already behind this part
            tap_next = kt0 / cpb;
            setup_tap(tap_next);
            ++tap_next;
            c_left = cpb - rem;
#pragma unroll
            for (int i = 0; i < 4; ++i) a_cur[i] += (long long)rem * a_inc[i];
        }
    }
    issue_stage(0, 0);
    if (nkt > 1) issue_stage(1, 1);

    // fragment addressing (byte offsets inside a stage)
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const unsigned a_row = (unsigned)((wm * WM + fr) * 128);
    const unsigned b_row = (unsigned)(DMA_A_BYTES + (wn * WN + fr) * 128);
    const unsigned sw0 = (unsigned)(((0 + fq) ^ (fr & 7)) << 4), sw1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);

    // BM = 256 (two wavefronts per SIMD behind one barrier): wavefronts 4-7 defer every stage's second MFMA group past
    // the next barrier (stagger: DESIGN.md section 4, round 2): they multiply while their SIMD partners issue DMA and read fragments.
    half8 a0[TM], b0[TN], a1[TM], b1[TN];
    const bool defer = (BM == 256) && wv >= 4;
    auto mma1 = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (DMA_STAGES == 3) {
            // stage kt has landed once at most one later stage (6..7 loads of this wavefront) is still in flight
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // two slots: only in the first iteration is a younger stage (9 loads) already in flight
            if (kt == 0 && nkt > 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (defer && kt > 0) mma1();             // second k-half of stage kt-1 (fragments read before the barrier)
        if constexpr (DMA_STAGES == 3) {
            if (kt + 2 < nkt) {
                int nbuf = buf + 2; if (nbuf >= DMA_STAGES) nbuf -= DMA_STAGES;
                issue_stage(kt + 2, nbuf);      // overwrites the stage read in iteration kt-1 (all waves are past it)
            }
        } else {
            if (kt >= 1 && kt + 1 < nkt) issue_stage(kt + 1, buf ^ 1);   // the slot read in iteration kt-1
        }
        const unsigned sb = lds0 + (unsigned)buf * DMA_STAGE_BYTES;
        {
            const unsigned aa = sb + a_row + sw0, ba = sb + b_row + sw0;
            DS_READ128(a0[0], aa, 0); DS_READ128(a0[1], aa, 2048); DS_READ128(a0[2], aa, 4096); DS_READ128(a0[3], aa, 6144);
            DS_READ128(b0[0], ba, 0); DS_READ128(b0[1], ba, 2048); DS_READ128(b0[2], ba, 4096); DS_READ128(b0[3], ba, 6144);
            DS_READ128(b0[4], ba, 8192);
        }
        {
            const unsigned aa = sb + a_row + sw1, ba = sb + b_row + sw1;
            DS_READ128(a1[0], aa, 0); DS_READ128(a1[1], aa, 2048); DS_READ128(a1[2], aa, 4096); DS_READ128(a1[3], aa, 6144);
            DS_READ128(b1[0], ba, 0); DS_READ128(b1[1], ba, 2048); DS_READ128(b1[2], ba, 4096); DS_READ128(b1[3], ba, 6144);
            DS_READ128(b1[4], ba, 8192);
        }
        asm volatile("s_waitcnt lgkmcnt(9)"
                     : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b0[4]));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[j], a0[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the second wait behind the first MFMA group
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(b1[4]));
        if (!defer) mma1();
        if (++buf == DMA_STAGES) buf = 0;
    }
    if (defer) mma1();
    if (p.ksplit > 1) {                  // fp32 partial tile of this K part: acc[i][j] = rows i*16 + fr, four columns j*16 + fq*4 ..
        float* ws = p.split_ws + (size_t)sp * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WM + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + fq * 4;
                if (m < p.M && n < p.N) *(float4v*)(ws + (size_t)m * p.N + n) = acc[i][j];      // (N % 8 == 0: whole quads)
            }
        }
        return;
    }
    __syncthreads();   // every wavefront is done reading the ring before the epilogue reuses it
    gemm_epilogue(p, acc, smem_raw, lane, wv, wm, wn, m0, n0, tile_n);
}

// The second half of a split-K contraction: out = epilogue(sum over the K parts, in order) with gemm_epilogue's arithmetic
// (bias and row vector added in fp32, scaled, rounded to fp16; then the residual / aux blend on the rounded value).
// One thread per 8 output columns.
__global__ void __launch_bounds__(256) k_splitk_finish(GemmParams p) {
    const int nch = p.N / 8;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long long)p.M * nch) return;
    const int m = (int)(q / nch), n = (int)(q - (long long)m * nch) * 8;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    for (int s = 0; s < p.ksplit; ++s) {
        const float4v* src = (const float4v*)(p.split_ws + ((size_t)s * p.M + m) * p.N + n);
        const float4v a = src[0], b = src[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] += a[e]; f[4 + e] += b[e]; }
    }
    if (p.bias) {
        const half8 b = *(const half8*)(p.bias + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)b[e];
    }
    if (p.rowvec) {
        const half8 t = *(const half8*)(p.rowvec + (long long)rowvec_index(m, p.rows_per_vec, p.rv_group) * p.ldrv + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)t[e];
    }
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(f[e] * p.s_acc);
    if (p.residual || p.aux) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
        if (p.residual) {
            const half8 r = *(const half8*)(p.residual + (long long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += p.s_res * (float)r[e];
        }
        if (p.aux) {
            const half8 a = *(const half8*)(p.aux + (long long)m * p.ldaux + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += p.s_aux * (float)a[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)f[e];
    }
    *(half8*)(p.out + (long long)m * p.ldc + n) = v;
}

// ---------------------------------------------------------------------------------------------
// Wide tile: 256 x 320 output tile, 512 threads, ONE block per CU, wavefront tile 64 x 160
// (4 x 10 MFMA tiles = 160 accumulator registers).  Against the 128 x 160 blocks it halves the L2->LDS bytes
// per FLOP (N = 320 is one tile: A is read exactly once) and issues 0.35 instead of 0.45 fragment reads per
// MFMA; the price is one wave-pair per SIMD and no second block to hide a tile's prologue and epilogue, so it
// is selected per shape (launch_dma).  2-stage LDS ring of 73,728-byte stages, one barrier per k-tile, the
// fragments of a k-half are read into the SAME registers after the 40 MFMAs of the previous half have issued
// (the partner wavefront on the SIMD covers the read latency).  The GEGLU pair [80 hidden | 80 gate] of a
// 160-column group lives in one wavefront, so the gate is applied in registers.  (Rounds 1-2 also carried a
// one-tile-per-block form of this tile, k_gemm_wide; the persistent form below superseded it on every shape and the
// shapes it does not admit go to the 160-column kernels.)
constexpr int WBM = 256, WBN = 320, WTN = 10;
constexpr int W_A_BYTES = WBM * BK * 2;                      // 32,768
constexpr int W_B_BYTES = WBN * BK * 2;                      // 40,960
constexpr int W_STAGE = W_A_BYTES + W_B_BYTES;               // 73,728

#ifdef SYN3R_TIMING
__device__ unsigned long long g_wide_timing[64];
#endif

// ---------------------------------------------------------------------------------------------
// PERSISTENT form of the 256 x 320 tile (dense contractions with M and N multiples of 8: every UNet projection).
// One block per CU (launch_widep caps the grid at the CU count); block b lives on XCD b % 8 and walks that XCD's
// contiguous chunk of the tile list with the stride of the XCD's block count, so in every round the 32 CUs of an XCD
// hold 32 neighbouring tiles - the order the one-tile-per-block grid has.  What the loop buys (measured on the
// one-tile kernel: a fixed ~10 us per tile next to ~17 us per 640 of K): stage 0 of the NEXT tile is requested during
// the last k-tile of this one, so its L2 / HBM latency and the block hand-over hide behind the epilogue, and the
// epilogue's stores drain under the next tile's k-loop instead of holding the CU until the block retires.
// With the tile loop around it the kernel has no register to spare for addressing (160 accumulators + 56 fragment
// registers): a staged piece is 8 whole rows, so with M, N multiples of 8 its row clamp is wave-uniform and the
// source address of a piece is an SGPR base (advanced on the scalar ALU) plus ONE per-lane byte offset that never
// changes (rows past the matrix re-read its last 8 rows; their products land in accumulator rows / columns the
// epilogue never stores).
// Lean epilogue of the persistent kernels for NI x 16 rows x 80 columns of a wavefront's accumulators (NI = 4: the whole
// 64-row tile, 11,264 B of staging; NI = 2: half of it, 5,632 B): (+bias +row vector) * s_acc -> fp16 through the
// wavefront's own LDS staging area `st` -> row-contiguous 16-byte stores (+ s_res * residual + s_aux * aux), row-major or
// A-tiled.  N is the logical column count (a multiple of 8, so a 16-byte chunk is inside the matrix or outside it as a whole).
template <int NI>
__device__ __forceinline__ void lean_store(const GemmParams& p, float4v (*acc)[TN], __half* st, int lane, int gm0,
                                           int gn0, int N, const __half* bias, const __half* residual, const __half* aux, bool full) {
    typedef _Float16 half4e __attribute__((ext_vector_type(4)));
    const int fr = lane & 15, fq = lane >> 4;
    // per-sample row vector (time embedding / folded cross-attention): row m takes rowvec[m / rows_per_vec] (or
    // rowvec[m mod |rows_per_vec|]); rows past M read vector 0 (their results are never stored)
    const __half* rv[NI];
    if (p.rowvec) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = gm0 + i * 16 + fr;
            const int vi = m < p.M ? rowvec_index(m, p.rows_per_vec, p.rv_group) : 0;
            rv[i] = p.rowvec + (long long)vi * p.ldrv;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = gn0 + j * 16 + fq * 4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) {
            const half4e b = *(const half4e*)(bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = (float)b[r];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float a4[4] = {b4[0], b4[1], b4[2], b4[3]};
            if (p.rowvec && n < N) {
                const half4e t = *(const half4e*)(rv[i] + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) a4[r] += (float)t[r];
            }
            half4e o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (_Float16)((acc[i][j][r] + a4[r]) * p.s_acc);
            *(half4e*)(st + (i * 16 + fr) * EPI_LD + j * 16 + fq * 4) = o;
        }
    }
    __builtin_amdgcn_wave_barrier();      // the staging area is the wavefront's own: program order is enough
    static_assert(NI % 2 == 0, "lean_store: an even number of row tiles");
    constexpr int NQ = NI * 16 * (WN / 8) / 64;   // 10 (NI = 4) or 5 (NI = 2) chunks of 16 bytes per lane
    auto put = [&](int m, int n, const half8& v) {
        if (!(full || (m < p.M && n < N))) return;
        if (p.out_tiled) OUT_STORE((half8*)(p.out + tiled_off(m, n, N)), v);
        else OUT_STORE((half8*)(p.out + (long long)m * p.ldc + n), v);
    };
    if (residual && aux) {                // (temporal blend: out = s_acc * y + s_res * residual + s_aux * aux)
        constexpr int RB = 5;             // rounds of five chunks: 2 x 20 registers of prefetched operands
#pragma unroll
        for (int h5 = 0; h5 < NQ; h5 += RB) {
            half8 res[RB], ax[RB];
#pragma unroll
            for (int it = 0; it < RB; ++it) {
                const int q = lane + (h5 + it) * 64, row = q / (WN / 8), ch = q - row * (WN / 8);
                int m = gm0 + row, n = gn0 + ch * 8;
                m = m < p.M ? m : p.M - 1; n = n < N ? n : N - 8;
                res[it] = *(const half8*)(residual + (long long)m * p.ldr + n);
                ax[it] = *(const half8*)(aux + (long long)m * p.ldaux + n);
            }
#pragma unroll
            for (int it = 0; it < RB; ++it) {
                const int q = lane + (h5 + it) * 64, row = q / (WN / 8), ch = q - row * (WN / 8);
                half8 v = *(const half8*)(st + row * EPI_LD + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {         // same order of additions as gemm_epilogue
                    float f = (float)v[e];
                    f += p.s_res * (float)res[it][e];
                    f += p.s_aux * (float)ax[it][e];
                    v[e] = (_Float16)f;
                }
                put(gm0 + row, gn0 + ch * 8, v);
            }
        }
    } else if (residual) {
        half8 res[NQ];                    // all ten requests first (clamped addresses: unconditional loads): one latency
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int q = lane + it * 64, row = q / (WN / 8), ch = q - row * (WN / 8);
            int m = gm0 + row, n = gn0 + ch * 8;
            m = m < p.M ? m : p.M - 1; n = n < N ? n : N - 8;
            res[it] = *(const half8*)(residual + (long long)m * p.ldr + n);
        }
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int q = lane + it * 64, row = q / (WN / 8), ch = q - row * (WN / 8);
            half8 v = *(const half8*)(st + row * EPI_LD + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + p.s_res * (float)res[it][e]);
            put(gm0 + row, gn0 + ch * 8, v);
        }
    } else {
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int q = lane + it * 64, row = q / (WN / 8), ch = q - row * (WN / 8);
            const half8 v = *(const half8*)(st + row * EPI_LD + ch * 8);
            put(gm0 + row, gn0 + ch * 8, v);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void widep_store(const GemmParams& p, float4v (&acc)[TM][TN], char* epi, int lane, int wv, int gm0,
                                            int gn0, int N, const __half* bias, const __half* residual, const __half* aux, bool full) {
    lean_store<TM>(p, acc, (__half*)epi + wv * (WM * EPI_LD), lane, gm0, gn0, N, bias, residual, aux, full);
}

__global__ void __launch_bounds__(512, 2) k_gemm_widep(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tiles_n = (p.N + WBN - 1) / WBN;
    const int tiles_m = (p.M + WBM - 1) / WBM;
    const unsigned nblk = (unsigned)(tiles_m * tiles_n);
    const unsigned xcd = blockIdx.x % 8, q8 = nblk / 8, r8 = nblk % 8;
    const unsigned t_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned t_len = q8 + (xcd < r8 ? 1u : 0u);
    const unsigned t_stride = (gridDim.x - xcd + 7) / 8;

    // wave-uniform BYTE offsets (from p.A / p.W; launch_widep checks that both operands span < 4 GB) of the 4 A and
    // 5 B pieces this wavefront stages: 9 scalar registers, advanced on the scalar ALU
    unsigned oa[4], ob[5];
    int m0 = 0, n0 = 0, tile_n = 0;
    const unsigned a_step = p.a_tiled ? 16384u : 2u * BK;      // bytes per k-tile
    const char* abase = (const char*)p.A;                      // the A source of the next stage (two-source A: see GemmParams)
    const int kt_switch = p.A2 ? p.K1 / BK : 0x7fffffff;       // first k-tile read from A2
    int ks = 0;                                                // k-tile index of the next stage of the tile being staged
    // Tile order: bands of 4 tile columns, row-major inside a band.  The 32 tiles an XCD holds at one time are then
    // 8 rows x 4 columns: per k-tile they pull 8 A slabs (32 KB) + 4 B slabs (40 KB) = 416 KB through that XCD's L2
    // for 2.4 MB of LDS fill, and the band's weight panel (4 x 320 rows x K) is what the XCD keeps re-reading round
    // after round.  Plain row-major order made that 2 x 16 (N = 5120: 704 KB) or 1 x 32 (N = 10240: 1.3 MB per
    // k-tile, ~6 TB/s chip-wide from beyond the L2) with a weight panel that no L2 holds.
    const unsigned bw0 = p.band > 0 ? (unsigned)p.band : 4u;
    const unsigned bw = (unsigned)tiles_n >= bw0 ? bw0 : (unsigned)tiles_n;
    const unsigned band_sz = (unsigned)tiles_m * bw, full_bands = (unsigned)tiles_n / bw;
    auto setup_tile = [&](unsigned tile) {
        unsigned b = tile / band_sz, w = bw, t2 = tile - b * band_sz;
        if (b >= full_bands) { b = full_bands; t2 = tile - full_bands * band_sz; w = (unsigned)tiles_n - full_bands * bw; }
        const int tile_m = (int)(t2 / w);
        tile_n = (int)(b * bw + t2 % w);
        m0 = tile_m * WBM; n0 = tile_n * WBN;
        ks = 0; abase = (const char*)p.A;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = m0 + wv * 32 + i * 8;
            r = r < p.M - 8 ? r : p.M - 8;
            if (p.a_tiled) oa[i] = 2u * ((unsigned)(r >> 7) * (unsigned)(p.K >> 6) * 8192u + (unsigned)(r & 127) * 64u);
            else oa[i] = 2u * (unsigned)r * (unsigned)p.lda;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int n = n0 + (wv * 5 + j) * 8;
            n = n < p.N - 8 ? n : p.N - 8;
            ob[j] = 2u * (unsigned)n * (unsigned)p.K;
        }
    };
    unsigned voff_a = 0, voff_a2 = 0, voff_b = 0;      // per-lane byte offset inside a piece (row lane >> 3, swizzled 16-byte chunk)
    auto issue_stage = [&](int buf, auto LOAD) {      // stages are issued in k order; LOAD = false only advances the offsets
        constexpr bool load = decltype(LOAD)::value;
        char* st = smem_raw + buf * W_STAGE;
        if (ks == kt_switch) {            // wave-uniform: from here on the A columns come from the second source
            abase = (const char*)p.A2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = m0 + wv * 32 + i * 8;
                r = r < p.M - 8 ? r : p.M - 8;
                oa[i] = 2u * (unsigned)r * (unsigned)p.lda2;
            }
        }
        const unsigned va = ks >= kt_switch ? voff_a2 : voff_a;
        ++ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (load) __builtin_amdgcn_global_load_lds((gbl_void_t*)(abase + (size_t)(oa[i] + va)), (lds_void_t*)(st + (wv * 4 + i) * 1024), 16, 0, 0);
            oa[i] += a_step;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if constexpr (load) __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)p.W + (size_t)(ob[j] + voff_b)), (lds_void_t*)(st + W_A_BYTES + (wv * 5 + j) * 1024), 16, 0, 0);
            ob[j] += 2u * BK;
        }
    };

    float4v acc[2][TM][TN];               // [column half][row tile][column tile]: halves are 80 columns each
    const int nkt = p.K / BK;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    unsigned a_row = 0, b_row = 0, swz[2] = {0, 0};
    const bool defer = wv >= 4;           // stagger of the SIMD partners, as in k_gemm_dma
    // Cross-tile prefetch: every tile starts in ring slot 0, so with an even k-tile count the last k-tile sits in
    // slot 1 and slot 0 is free for the next tile's stage 0, while slot 1 plus the 16 KB behind the ring are exactly
    // the 90,112 bytes the epilogue stages the accumulators through.  Odd counts issue after the epilogue.
    const bool xpf = (nkt & 1) == 0;
#ifdef SYN3R_TIMING     // tools/wide_timing.py: s_memtime ticks of one block's tile phases + both clocks around the tile loop
    unsigned long long tph[3] = {0, 0, 0}, tkt[2] = {0, 0}, tgate = 0, ntile = 0, t_a = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_a, r_begin = __builtin_amdgcn_s_memrealtime();
#define PSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - t_a; t_a = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PSTAMP(i)
#endif
    bool staged = false;                  // stage 0 of the tile about to start is already in flight
    for (unsigned tl = blockIdx.x / 8; tl < t_len; tl += t_stride) {
        {   // everything derived from the lane id is rebuilt per tile behind an opaque copy: hoisted out of the tile
            // loop it would be carried through the epilogue in registers the 160 accumulators do not leave
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int prow = lo >> 3, csrc = (lo & 7) ^ prow;
            voff_a = p.a_tiled ? (unsigned)((prow * 64 + csrc * 8) * 2) : (unsigned)(prow * (int)p.lda + csrc * 8) * 2u;
            voff_a2 = (unsigned)(prow * (int)p.lda2 + csrc * 8) * 2u;
            voff_b = (unsigned)(prow * p.K + csrc * 8) * 2u;
            const int fr = lo & 15, fq = lo >> 4;
            a_row = (unsigned)((wm * WM + fr) * 128);
            b_row = (unsigned)(W_A_BYTES + (wn * 160 + fr) * 128);
            swz[0] = (unsigned)(((0 + fq) ^ (fr & 7)) << 4);
            swz[1] = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
        }
        setup_tile(t_start + tl);
        if (staged) {
            issue_stage(0, std::false_type{});      // stage 0 was requested during the previous tile's last k-tile
        } else {
            __syncthreads();                        // the previous tile's epilogue is done with the ring
            issue_stage(0, std::true_type{});
        }
        staged = false;
        const int em0 = m0, en0 = n0, etn = tile_n;     // this tile's origin (setup_tile moves on during the last k-tile)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[hh][i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
        // (declared per tile: the fragment registers are read-modify-write operands of the asm reads, at function scope
        // they would stay live - 56 registers - through the epilogue)
        half8 af[TM], bf[WTN];
        auto read_half = [&](unsigned sbase, int kh) {
            const unsigned aa = sbase + a_row + swz[kh], ba = sbase + b_row + swz[kh];
            DS_READ128(af[0], aa, 0); DS_READ128(af[1], aa, 2048); DS_READ128(af[2], aa, 4096); DS_READ128(af[3], aa, 6144);
            DS_READ128(bf[0], ba, 0); DS_READ128(bf[1], ba, 2048); DS_READ128(bf[2], ba, 4096); DS_READ128(bf[3], ba, 6144);
            DS_READ128(bf[4], ba, 8192); DS_READ128(bf[5], ba, 10240); DS_READ128(bf[6], ba, 12288); DS_READ128(bf[7], ba, 14336);
            DS_READ128(bf[8], ba, 16384); DS_READ128(bf[9], ba, 18432);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]),
                           "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), "+v"(bf[7]), "+v"(bf[8]), "+v"(bf[9]));
        };
        auto mma = [&]() {
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[j / TN][i][j % TN] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[j / TN][i][j % TN], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // the next reads reuse af / bf: keep them behind these MFMAs
        };
        // The fragment registers are read-modify-write operands of the asm reads: (re)define them here with an empty
        // output-only asm, or all 56 stay live from one tile's last read through the epilogue to the next tile's first.
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(af[i]));
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("" : "=v"(bf[j]));
        PSTAMP(0);
#ifdef SYN3R_TIMING
        unsigned long long tkt_prev = 0;
#endif
        int buf = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage kt has landed (and, at kt = 0, the previous tile's stores)
            __builtin_amdgcn_s_barrier();
            if (defer && kt > 0) mma();                          // second k-half of stage kt-1 (deferred wavefronts)
            if (kt + 1 < nkt) issue_stage(buf ^ 1, std::true_type{});   // the slot every wavefront finished reading in iteration kt-1
            else if (xpf && tl + t_stride < t_len) {
                setup_tile(t_start + tl + t_stride);             // (rebuilt at the top of the next tile: nothing stays live)
                issue_stage(0, std::true_type{});
                staged = true;
            }
            const unsigned sbase = lds0 + (unsigned)buf * W_STAGE;
            read_half(sbase, 0);
            mma();
            read_half(sbase, 1);
            if (!defer) mma();
            buf ^= 1;
#ifdef SYN3R_TIMING
            if (kt < 2) { __builtin_amdgcn_sched_barrier(0); tkt[kt] += __builtin_amdgcn_s_memtime() - t_a - (kt ? tkt_prev : 0); tkt_prev = __builtin_amdgcn_s_memtime() - t_a; }
#endif
        }
        if (defer) mma();
        PSTAMP(1);
        __syncthreads();   // every wavefront is done reading the ring before the epilogue reuses it
        char* epi = staged ? smem_raw + W_STAGE : smem_raw;
        int le = lane;                    // opaque per tile, as above: the epilogue's lane-derived indices stay inside the tile
        asm volatile("" : "+v"(le));
        const int gm0 = em0 + wm * WM;
        if (p.geglu_D > 0) {
            // GEGLU.forward: hidden * gelu(gate) on the fp16-rounded projection output (activations.py); the wavefront's
            // 160 columns are one packed group [80 hidden | 80 gate]
            const int fq = le >> 4;
            const int gn = en0 + wn * 160;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = gn + j * 16 + fq * 4;
                float bh[4], bg[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bh[r] = (p.bias && n + r < p.N) ? __half2float(p.bias[n + r]) : 0.f;
                    bg[r] = (p.bias && n + 80 + r < p.N) ? __half2float(p.bias[n + 80 + r]) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {       // pairs: the gate is packed fp32 arithmetic (gelu_pk, common.h)
                        const syn3r_f2 hv = (syn3r_f2){(float)(_Float16)(acc[0][i][j][r] + bh[r]), (float)(_Float16)(acc[0][i][j][r + 1] + bh[r + 1])};
                        const syn3r_f2 gv = (syn3r_f2){(float)(_Float16)(acc[1][i][j][r] + bg[r]), (float)(_Float16)(acc[1][i][j][r + 1] + bg[r + 1])};
                        const syn3r_f2 y = hv * gelu_pk(gv);
                        acc[0][i][j][r] = y.x; acc[0][i][j][r + 1] = y.y;       // (launch_widep: s_acc == 1 with a gate)
                    }
            }
#ifdef SYN3R_TIMING
            { __builtin_amdgcn_sched_barrier(0); tgate += __builtin_amdgcn_s_memtime() - t_a; __builtin_amdgcn_sched_barrier(0); }
#endif
            const int go0 = etn * 160 + wn * WN;
            const bool full = gm0 + WM <= p.M && go0 + WN <= p.geglu_D;
            widep_store(p, acc[0], epi, le, wv, gm0, go0, p.geglu_D, nullptr, nullptr, nullptr, full);
        } else {
            const int gn0 = en0 + wn * 160;
            const bool full = gm0 + WM <= p.M && gn0 + 160 <= p.N;
            widep_store(p, acc[0], epi, le, wv, gm0, gn0, p.N, p.bias, p.residual, p.aux, full);
            widep_store(p, acc[1], epi, le, wv, gm0, gn0 + WN, p.N, p.bias, p.residual, p.aux, full);
        }
        PSTAMP(2);
#ifdef SYN3R_TIMING
        ++ntile;
#endif
    }
#ifdef SYN3R_TIMING
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
        g_wide_timing[wv * 8 + 0] = tph[0]; g_wide_timing[wv * 8 + 1] = tph[1]; g_wide_timing[wv * 8 + 2] = tph[2];
        g_wide_timing[wv * 8 + 3] = ntile;
        g_wide_timing[wv * 8 + 4] = __builtin_amdgcn_s_memtime() - t_begin;
        g_wide_timing[wv * 8 + 5] = __builtin_amdgcn_s_memrealtime() - r_begin;
        g_wide_timing[wv * 8 + 6] = tkt[0]; g_wide_timing[wv * 8 + 7] = tgate ? tgate : tkt[1];     // (gated tiles: the gate's share of the epilogue)
    }
#endif
#undef PSTAMP
}

#include "gemm_z.h"

// ---------------------------------------------------------------------------------------------
// PERSISTENT form of the 256-row LDS-DMA kernel (k_gemm_dma<MODE, 256>), used for the temporal convolutions and the dense
// contractions that stay on the 256 x 160 tile (the 3x3 convolutions measured 1-3 % slower with it and keep one tile per block).  Same tile, ring (3 slots of 53,248 B), staggered wavefronts
// and per-lane im2col addressing; what changes is what happens at a tile boundary:
//   - the block walks its XCD's share of the tiles (one block per CU, as k_gemm_widep);
//   - the DMA ISSUE CURSOR runs on across tile boundaries: stages are numbered through the block's whole tile list, the
//     ring slot of stage g is g mod 3, and during the last two k-tiles of a tile the cursor already requests the first
//     two stages of the next one - their L2 / HBM latency hides behind this tile's epilogue;
//   - the epilogue therefore has ONE slot (the last one read) instead of the whole ring: the accumulators go through it
//     in two passes of 32 rows per wavefront (lean_store<2>: 8 x 5,632 B = 45 KB), stores drain under the next k-loop.
// vmcnt bookkeeping: a stage wait is "all but the one younger stage" (vmcnt(6), as k_gemm_dma) except for the first
// k-tile after an epilogue, where the epilogue's loads and stores sit between the two prefetched stages: vmcnt(0)
// (both stages were requested a whole epilogue earlier).
template <int MODE>
__global__ void __launch_bounds__(512, 2) k_gemm_dmap(GemmParams p) {
    constexpr int BM = 256;
    constexpr int DMA_A_BYTES = BM * BK * 2;                  // 32,768
    constexpr int STAGE = DMA_A_BYTES + DMA_B_BYTES;          // 53,248
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const unsigned nblk = (unsigned)(tiles_m * tiles_n);
    const unsigned xcd = blockIdx.x % 8, q8 = nblk / 8, r8 = nblk % 8;
    const unsigned t_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned t_len = q8 + (xcd < r8 ? 1u : 0u);
    const unsigned t_stride = (gridDim.x - xcd + 7) / 8;
    const int nkt = p.K / BK;
    const int cpb = (MODE == MODE_DENSE) ? 1 : p.Cin / BK;

    // ---- issue cursor: per-lane source state of the tile whose stages are being requested
    const int prow = lane >> 3;
    const int csrc = (lane & 7) ^ prow;                 // source 16-byte chunk that lands in slot (lane & 7)
    const __half* zero = g_zero_page;
    const int nb = wv < 4 ? 3 : 2;                      // B pieces of this wavefront (20 in total: 3,3,3,3,2,2,2,2)
    const int b_first = wv < 4 ? wv * 3 : 12 + (wv - 4) * 2;
    const __half* a_base[4];
    int a_n[4], a_y[4], a_x[4];
    const __half* a_cur[4];
    int a_inc[4];
    const __half* b_cur[3];
    int b_inc[3];
    int tap_next = 0, c_left = 0;
    unsigned itl = blockIdx.x / 8;                      // the cursor's position in this block's tile list ...
    int ikt = 0, islot = 0;                             // ... k-tile inside that tile, ring slot of the next stage
    // row tile of position `rt` in the tile order (GemmParams::tc_pb: temporal convolutions walk the frames of a pixel block first)
    auto row_tile = [&](unsigned rt) -> int {
        if constexpr (MODE == MODE_TCONV) {
            if (p.tc_pb > 0) { const unsigned pb = rt / (unsigned)p.tc_nf, fr_ = rt - pb * (unsigned)p.tc_nf; return (int)(fr_ * (unsigned)p.tc_pb + pb); }
        }
        return (int)rt;
    };
    auto setup_issue_tile = [&](unsigned tile) {
        const int m0 = row_tile(tile / (unsigned)tiles_n) * BM, n0 = (int)(tile % (unsigned)tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wv * 32 + i * 8 + prow;
            const int mc = m < p.M ? m : p.M - 1;
            if constexpr (MODE == MODE_DENSE) {
                const int last_rb = (p.M + 127) >> 7, rb = (m >> 7) < last_rb ? (m >> 7) : last_rb - 1;
                a_base[i] = p.a_tiled ? p.A + (long long)rb * (p.K >> 6) * 8192 + (m & 127) * 64 + csrc * 8
                                      : p.A + (long long)mc * p.lda + csrc * 8;
                a_n[i] = a_y[i] = a_x[i] = 0;
                a_cur[i] = a_base[i]; a_inc[i] = p.a_tiled ? 8192 : BK;
            } else if constexpr (MODE == MODE_CONV2D) {
                const int hw = p.Ho * p.Wo;
                a_n[i] = mc / hw;
                const int r = mc - a_n[i] * hw;
                a_y[i] = r / p.Wo;
                a_x[i] = r - a_y[i] * p.Wo;
                a_base[i] = p.A + csrc * 8;
            } else {
                a_y[i] = (mc / p.HW) % p.F;
                a_n[i] = a_x[i] = 0;
                a_base[i] = p.A + (long long)mc * p.Cin + csrc * 8;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + (b_first + j) * 8 + prow;
            const bool ok = j < nb && n < p.N;
            b_cur[j] = ok ? p.W + (long long)n * p.K + csrc * 8 : zero;
            b_inc[j] = ok ? BK : 0;
        }
        tap_next = 0; c_left = 0;
    };
    auto setup_tap = [&](int tap) {
        if constexpr (MODE == MODE_CONV2D) {
            const int dy = tap / 3 - p.pad, dx = tap % 3 - p.pad;
            const int Hg = p.ups ? p.Hi * 2 : p.Hi, Wg = p.ups ? p.Wi * 2 : p.Wi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int yy = a_y[i] * p.stride + dy, xx = a_x[i] * p.stride + dx;
                const bool ok = yy >= 0 && yy < Hg && xx >= 0 && xx < Wg;
                if (p.ups) { yy >>= 1; xx >>= 1; }
                const long long off = (((long long)a_n[i] * p.Hi + yy) * p.Wi + xx) * p.Cin;
                a_cur[i] = ok ? a_base[i] + off : zero;
                a_inc[i] = ok ? BK : 0;
            }
        } else if constexpr (MODE == MODE_TCONV) {
            const int df = tap - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ff = a_y[i] + df;
                const bool ok = ff >= 0 && ff < p.F;
                a_cur[i] = ok ? a_base[i] + (long long)df * p.HW * p.Cin : zero;
                a_inc[i] = ok ? BK : 0;
            }
        }
    };
    auto issue_next = [&]() -> bool {     // request the next stage of the block's stage sequence; false: none left
        if (itl >= t_len) return false;
        if (ikt == 0) setup_issue_tile(t_start + itl);
        char* st = smem_raw + islot * STAGE;
        if constexpr (MODE != MODE_DENSE) {
            if (c_left == 0) { setup_tap(tap_next); ++tap_next; c_left = cpb; }
            --c_left;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)a_cur[i], (lds_void_t*)(st + (wv * 4 + i) * 1024), 16, 0, 0);
            a_cur[i] += a_inc[i];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < nb) {
                __builtin_amdgcn_global_load_lds((gbl_void_t*)b_cur[j], (lds_void_t*)(st + DMA_A_BYTES + (b_first + j) * 1024), 16, 0, 0);
                b_cur[j] += b_inc[j];
            }
        }
        if (++ikt == nkt) { ikt = 0; itl += t_stride; }
        if (++islot == 3) islot = 0;
        return true;
    };

    // fragment addressing (byte offsets inside a stage)
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const unsigned a_row = (unsigned)((wm * WM + fr) * 128);
    const unsigned b_row = (unsigned)(DMA_A_BYTES + (wn * WN + fr) * 128);
    const unsigned sw0 = (unsigned)(((0 + fq) ^ (fr & 7)) << 4), sw1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    const bool defer = wv >= 4;           // stagger of the SIMD partners (see k_gemm_widep)

    int issued = 0, consumed = 0;         // stages requested / stages whose k-tile has been multiplied (wave-uniform)
    if (issue_next()) ++issued;
    if (issue_next()) ++issued;
    int cslot = 0;
    bool first_tile = true;
    for (unsigned tl = blockIdx.x / 8; tl < t_len; tl += t_stride) {
        const unsigned tile = t_start + tl;
        const int m0 = row_tile(tile / (unsigned)tiles_n) * BM, n0 = (int)(tile % (unsigned)tiles_n) * BN;
        float4v acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
        half8 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { asm volatile("" : "=v"(a0[i])); asm volatile("" : "=v"(a1[i])); }   // (not live across tiles)
#pragma unroll
        for (int j = 0; j < TN; ++j) { asm volatile("" : "=v"(b0[j])); asm volatile("" : "=v"(b1[j])); }
        auto mma1 = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a1[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt < nkt; ++kt) {
            // the stage of this k-tile has landed once only the ONE younger stage (6..7 loads of this wavefront) is in flight
            if ((kt == 0 && !first_tile) || issued - consumed < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (defer && kt > 0) mma1();             // second k-half of the previous stage (fragments read before the barrier)
            if (issue_next()) ++issued;              // overwrites the slot read one iteration ago (all wavefronts are past it)
            const unsigned sb = lds0 + (unsigned)cslot * STAGE;
            {
                const unsigned aa = sb + a_row + sw0, ba = sb + b_row + sw0;
                DS_READ128(a0[0], aa, 0); DS_READ128(a0[1], aa, 2048); DS_READ128(a0[2], aa, 4096); DS_READ128(a0[3], aa, 6144);
                DS_READ128(b0[0], ba, 0); DS_READ128(b0[1], ba, 2048); DS_READ128(b0[2], ba, 4096); DS_READ128(b0[3], ba, 6144);
                DS_READ128(b0[4], ba, 8192);
            }
            {
                const unsigned aa = sb + a_row + sw1, ba = sb + b_row + sw1;
                DS_READ128(a1[0], aa, 0); DS_READ128(a1[1], aa, 2048); DS_READ128(a1[2], aa, 4096); DS_READ128(a1[3], aa, 6144);
                DS_READ128(b1[0], ba, 0); DS_READ128(b1[1], ba, 2048); DS_READ128(b1[2], ba, 4096); DS_READ128(b1[3], ba, 6144);
                DS_READ128(b1[4], ba, 8192);
            }
            asm volatile("s_waitcnt lgkmcnt(9)"
                         : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(b0[4]));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[j], a0[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep the second wait behind the first MFMA group
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(b1[4]));
            if (!defer) mma1();
            ++consumed;
            if (++cslot == 3) cslot = 0;
        }
        if (defer) mma1();
        first_tile = false;
        __syncthreads();   // every wavefront is done reading the last stage: its slot is the epilogue's staging area
        {
            const int last = cslot == 0 ? 2 : cslot - 1;
            int le = lane;                    // opaque per tile: the epilogue's lane-derived indices stay inside the tile loop
            asm volatile("" : "+v"(le));
            __half* st = (__half*)(smem_raw + last * STAGE) + wv * (32 * EPI_LD);
            const int gm0 = m0 + wm * WM, gn0 = n0 + wn * WN;
            const bool full = gm0 + WM <= p.M && gn0 + WN <= p.N;
            lean_store<2>(p, acc, st, le, gm0, gn0, p.N, p.bias, p.residual, p.aux, full);
            lean_store<2>(p, acc + 2, st, le, gm0 + 32, gn0, p.N, p.bias, p.residual, p.aux, full);
        }
        // (the next tile's first barrier orders these staging reads before the DMA that reuses the slot)
    }
}

// 16-byte chunk swizzle of LDS images with 64-byte rows (4 chunks): slot = chunk ^ s(row >> 2 & 3) with s = (0,2,3,1) keeps every
// 16-lane group of a ds_read_b128 fragment read on 16 different 16-byte bank units (k_lnlin320's weight stages).
__device__ __forceinline__ int h_swz(int row_in_16) { return (0x78 >> (2 * (row_in_16 >> 2))) & 3; }

// ---------------------------------------------------------------------------------------------
// Fused feed-forward for C = 320 (the level-0 transformer blocks: FeedForward.forward, attention.py:608-665, with the
// GEGLU of activations.py):   out = epilogue( geglu(x . W1^T + b1) . W2^T )   in ONE kernel.
// The two-kernel path writes the gated hidden activation ([M, 1280] fp16 = 660 MB at M = 258 048) and reads it back;
// round 1 measured the first projection at half its matrix rate because of that output stream.  Here a
// block owns 128 rows: its x tile (80 KB) stays in LDS, the hidden dimension is walked in chunks of 64 —
//     phase 1   S[128, 128]  = x . W1_j^T            (K = 320, five 64-wide k-tiles of the chunk's 128 packed rows)
//     gate      h[128, 64]   = (S_h + b) * gelu(S_g + b)   in registers, fp16-rounded as the reference's projection output
//     phase 2   out[128,320] += h . W2[:, j]^T       (K = 64)
// and the [128, 320] fp32 result lives in registers for the whole kernel: EIGHT wavefronts (2 x 4), each 64 rows x 80
// output columns (80 accumulators) + its 64 x 32 slice of S (32), so a wavefront stays under 256 registers and every
// SIMD holds TWO: one wavefront's gate arithmetic, LDS-DMA issue (≈60-100 cycles per 1 KiB piece, MI355X_MICROARCH.md)
// and barrier waits overlap the other's MFMAs (a one-wavefront-per-SIMD build of this kernel ran 2.1x slower: 590
// TFLOP/s, those phases serialise).  Nothing but x and out touches HBM; the weights (2.4 MB, L2-resident) stream
// through a 3-slot LDS ring by LDS-DMA with counted vmcnt and ONE barrier per stage.
// LDS: x 80 KB | ring 3 x 20 KB | h 16 KB | per-wavefront bias lines 4 KB = 163,840 B (all of it).
// W1 rows are packed per 64-wide chunk as 4 x [16 hidden | 16 gate] (wavefront column wn owns one group, so a lane
// holds a hidden value and its gate in matching accumulator tiles).
constexpr int F_C = 320, F_HC = 64, F_BM = 128;
constexpr int F_X_BYTES = F_BM * F_C * 2;            // 81,920
constexpr int F_SLOT = 160 * BK * 2;                 // 20,480: a W2 half-chunk [160 x 64]; W1 k-tiles [128 x 64] use 16,384 of it
constexpr int F_RING = F_X_BYTES;                    // ring offset
constexpr int F_H = F_RING + 3 * F_SLOT;             // 143,360
constexpr int F_BIAS = F_H + F_BM * F_HC * 2;        // 159,744
constexpr int F_LDS = F_BIAS + 8 * 512;              // 163,840

struct FfnParams {
    GemmParams e;            // A = x, lda; W = w2 [320, D]; out / ldc; bias = b2; residual / aux / scales; M; N = 320
    const __half* w1;        // [D/64][128][320] packed rows
    const __half* b1;        // [D/64][128] packed
    int D;                   // hidden width (multiple of 64)
    const __half* ln_g;      // non-null: x is LayerNorm'ed (gamma, beta, eps over the 320 channels) inside the kernel first
    const __half* ln_b;
    float ln_eps;
    const __half* ln_add;    // non-null: x + ln_add[row / ln_add_rpv] (fp16 tensor add) is what gets normalised ([rows, 320], 16-byte aligned)
    int ln_add_rpv;
};

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
#define DS_READ64(dst, addr, OFF) asm volatile("ds_read_b64 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define DS_WRITE64(addr, val) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(val) : "memory")

// LayerNorm of a resident [128 x 320] x tile (five [128 x 64] k-tile images, 16-byte chunk index XOR-swizzled by the row) in
// place, by all 512 threads of the block; waits for the tile's DMA first.  Shared by k_ffn320 and k_lnlin320.
__device__ __forceinline__ void ln_tile320(char* smem_raw, int tid, int m0, int M, float cf, const __half* ln_g, const __half* ln_b,
                                           float ln_eps, const __half* add, int add_rpv) {
    // LayerNorm of the resident x tile (attention.py:430-453: norm3 in front of ff), so that the normalised activation is never
    // written to / re-read from HBM.  Same arithmetic, same order of additions as k_layernorm<8> (norm.hip): 8 partial sums per
    // row over the 16-byte chunks c, c + 8, ..., combined by the xor tree 4, 2, 1 - here four threads per row hold two of the
    // eight each.  The tile is five [128 x 64] k-tile images with the chunk index XOR-swizzled by the row.
    {
        const int r = tid >> 2, part = tid & 3;
        half8 xv[5][2], addv[5][2];
        if (add) {       // requested before the wait for the x tile: one latency, not two
            int m = m0 + r;
            m = m < M ? m : M - 1;
            const __half* av = add + (long long)(m / add_rpv) * F_C;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int e = 0; e < 2; ++e) addv[kt][e] = *(const half8*)(av + (kt * 8 + part * 2 + e) * 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                xv[kt][e] = *(const half8*)(smem_raw + kt * 16384 + r * 128 + (((part * 2 + e) ^ (r & 7)) << 4));
        if (add) {       // norm_in of the temporal block normalises hidden + frame-position embedding (attention.py:500-507)
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int e = 0; e < 2; ++e) xv[kt][e] = xv[kt][e] + addv[kt][e];   // fp16 add, as k_layernorm
        }
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int i = 0; i < 8; ++i) { sa += (float)xv[kt][0][i]; sb += (float)xv[kt][1][i]; }
        sa += __shfl_xor(sa, 2, 64); sb += __shfl_xor(sb, 2, 64);
        sa += __shfl_xor(sa, 1, 64); sb += __shfl_xor(sb, 1, 64);
        // cf = 320 as a run-time value: the same division k_layernorm compiles to
        const float mean = (sa + sb) / cf;
        float qa = 0.f, qb = 0.f;
        {
#pragma clang fp contract(off)      // k_layernorm's squares are a packed multiply followed by adds, not an fma: the same bits here
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float da = (float)xv[kt][0][i] - mean, db = (float)xv[kt][1][i] - mean;
                    const float da2 = da * da, db2 = db * db;
                    qa += da2; qb += db2;
                }
        }
        qa += __shfl_xor(qa, 2, 64); qb += __shfl_xor(qb, 2, 64);
        qa += __shfl_xor(qa, 1, 64); qb += __shfl_xor(qb, 1, 64);
        const float rstd = rsqrtf((qa + qb) / cf + ln_eps);
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cv = kt * 8 + part * 2 + e;
                const half8 g = *(const half8*)(ln_g + cv * 8), b = *(const half8*)(ln_b + cv * 8);
                half8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (_Float16)(((float)xv[kt][e][i] - mean) * rstd * (float)g[i] + (float)b[i]);
                *(half8*)(smem_raw + kt * 16384 + r * 128 + (((part * 2 + e) ^ (r & 7)) << 4)) = o;
            }
        __syncthreads();
    }

}

// ---------------------------------------------------------------------------------------------
// k_ffn320r: the fused feed-forward with the x tile in REGISTERS (round 4).  In k_ffn320 the resident x tile is half of the LDS:
// the weight ring has three slots (two stages of look-ahead), every stage re-reads the x fragments from LDS (320 of the 624 KB
// a chunk reads), and the block is alone on its CU.  Here the eight wavefronts are 4 (row groups of 32) x 2 (column halves): a
// wavefront keeps ITS 32 rows of x as MFMA fragments (2 x 10 x 16 B per lane = 80 registers, normalised in registers with the
// arithmetic and summation order of k_layernorm<8>), which frees 80 KB: the ring has SEVEN 20 KB slots (a chunk's five W1 k-tiles
// and two W2 halves), so a chunk needs THREE barriers instead of six (k-tiles 0-2 | k-tiles 3-4, gate | W2 + h) with every stage
// issued two barrier intervals ahead, and inside an interval the weight fragments of k-step t + 1 are read under the MFMAs of
// k-step t (two 4-fragment buffers); phase 1 reads only weight fragments (8 instead of 12 ds_read_b128 per k-tile and wavefront),
// phase 2 reads the wavefront's own W2 half.  Same arithmetic, same accumulation order as k_ffn320: bit-identical output.
// Measured inside the unit, same box: 11.9 ms against 12.6-13.0 (15 launches at M = 258 048); with one barrier per stage and no
// read-ahead the same kernel ran 14.0 ms, with 13 spilled registers (scratch reloads drain the DMA queue) 16.8 ms.
// Registers: x 80 + out 32 x 160 (80) + S 32 x 64 (32) + fragments.  LDS: ring 7 x 20 KB | h 16 KB | bias 4 KB = 163,840 B.
constexpr int R_SLOTS = 7;
constexpr int R_H = R_SLOTS * F_SLOT;                // 143,360
constexpr int R_BIAS = R_H + F_BM * F_HC * 2;        // 159,744
constexpr int R_LDS = R_BIAS + 8 * 512;              // 163,840

__global__ void __launch_bounds__(512, 2) k_ffn320r(FfnParams q) {
    const GemmParams& p = q.e;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;              // 4 row groups of 32 rows x 2 column halves
    const int tiles_m = (p.M + F_BM - 1) / F_BM;
    const int m0 = (int)xcd_remap(blockIdx.x, (unsigned)tiles_m) * F_BM;
    const int nchunks = q.D / F_HC;
    const long long D = q.D;
    const int fr = lane & 15, fq = lane >> 4;
    typedef _Float16 half4e __attribute__((ext_vector_type(4)));

    // ---- weight DMA (as k_ffn320): W1 k-tile = 16 pieces of 8 rows x 128 B (2 per wavefront), W2 half = 20 pieces (3 / 2)
    const int prow = lane >> 3;
    const int csrc = (lane & 7) ^ prow;
    const int nbw = wv < 4 ? 3 : 2;
    const int b_first = wv < 4 ? wv * 3 : 12 + (wv - 4) * 2;
    // per-lane 32-bit byte offsets; the stage's base stays a scalar (opaque to the optimiser, as in k_attn_spatial), so the copies
    // take the scalar-base + lane-offset form and no 64-bit per-lane pointer lives across the chunk loop
    unsigned ow1 = (unsigned)((((wv * 2) * 8 + prow) * F_C + csrc * 8) * 2);
    unsigned ow2 = (unsigned)(((long long)(b_first * 8 + prow) * D + csrc * 8) * 2);
    unsigned ob1 = (unsigned)(lane * 4);
    char* const bias_line = smem_raw + R_BIAS + wv * 512;
    // Ring: slots 0..4 = the chunk's five W1 k-tiles, slots 5, 6 = its two W2 halves.  THREE barriers per chunk (k-tiles 0-2, k-tiles
    // 3-4, W2 + h; k_ffn320: six); behind each one the stages whose slots the barrier just released are issued:
    //   I1(j): W2(j)          I2(j): W1(j+1, 0..2)          I3(j): W1(j+1, 3), W1(j+1, 4)
    // i.e. every stage is issued two barrier intervals before it is needed.  Per-wavefront DMA instructions: W1 k-tile 2 (+ 1 bias
    // line with k-tile 0), W2 2 * nbw.
    auto issue_w1 = [&](int ij, int ir) {
        asm volatile("" : "+v"(ow1), "+v"(ob1));
        long long soff = ((long long)ij * (128 * F_C) + ir * BK) * 2;
        asm volatile("" : "+s"(soff));
        const char* src = (const char*)q.w1 + soff;
        char* slot = smem_raw + ir * F_SLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + i * (8 * F_C * 2) + (size_t)ow1), (lds_void_t*)(slot + (wv * 2 + i) * 1024), 16, 0, 0);
        if (ir == 0) {
            long long boff = (long long)ij * 256;
            asm volatile("" : "+s"(boff));
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)q.b1 + boff + (size_t)ob1), (lds_void_t*)(bias_line + (ij & 1) * 256), 4, 0, 0);
        }
    };
    auto issue_w2 = [&](int ij) {
        asm volatile("" : "+v"(ow2));
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            long long soff = ((long long)hh * 160 * D + (long long)ij * F_HC) * 2;
            asm volatile("" : "+s"(soff));
            const char* src = (const char*)p.W + soff;
            char* slot = smem_raw + (5 + hh) * F_SLOT;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nbw)
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (long long)i * 8 * D * 2 + (size_t)ow2), (lds_void_t*)(slot + (b_first + i) * 1024), 16, 0, 0);
        }
    };
    auto wait_vm = [&](int n) {
        switch (n) {
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };

    // ---- x fragments: lane (fr, fq) holds, for row tile i and k-step ks (32 wide), x[row i*16 + fr][ks*32 + fq*8 .. +8]
    half8 xf[2][10];
    {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + wm * 32 + i * 16 + fr;
            m = m < p.M ? m : p.M - 1;
            const __half* xr = p.A + (long long)m * p.lda + fq * 8;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) xf[i][ks] = *(const half8*)(xr + ks * 32);
        }
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) issue_w1(0, s);      // chunk 0's W1 k-tiles; its W2 halves follow behind the first barrier
    if (q.ln_g) {
        // LayerNorm in registers: the 16-byte chunk c = 4 ks + fq of a row belongs to k_layernorm<8>'s lane sub = c % 8, i.e. this
        // lane holds sub = fq (even ks) and sub = fq + 4 (odd ks), each in k_layernorm's order; its xor tree 4, 2, 1 is
        // (own pair) , lane ^ 32 , lane ^ 16 here.  Same expressions as ln_tile320 / k_layernorm: the same bits.
        const float cf = (float)p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (q.ln_add) {
                int m = m0 + wm * 32 + i * 16 + fr;
                m = m < p.M ? m : p.M - 1;
                const __half* av = q.ln_add + (long long)(m / q.ln_add_rpv) * F_C + fq * 8;
#pragma unroll
                for (int ks = 0; ks < 10; ++ks) xf[i][ks] = xf[i][ks] + *(const half8*)(av + ks * 32);   // fp16 add, as k_layernorm
            }
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) { sa += (float)xf[i][2 * k][e]; sb += (float)xf[i][2 * k + 1][e]; }
            float s = sa + sb;
            s += __shfl_xor(s, 32, 64);
            s += __shfl_xor(s, 16, 64);
            const float mean = s / cf;
            float qa = 0.f, qb = 0.f;
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float da = (float)xf[i][2 * k][e] - mean, db = (float)xf[i][2 * k + 1][e] - mean;
                        const float da2 = da * da, db2 = db * db;
                        qa += da2; qb += db2;
                    }
            }
            float qq = qa + qb;
            qq += __shfl_xor(qq, 32, 64);
            qq += __shfl_xor(qq, 16, 64);
            const float rstd = rsqrtf(qq / cf + q.ln_eps);
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const half8 g = *(const half8*)(q.ln_g + ks * 32 + fq * 8), b = *(const half8*)(q.ln_b + ks * 32 + fq * 8);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)xf[i][ks][e] - mean) * rstd * (float)g[e] + (float)b[e]);
                xf[i][ks] = o;
            }
        }
    }

    float4v acc[2][10];                   // out: 32 rows x 160 columns of this wavefront
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const unsigned sw0 = (unsigned)(((0 + fq) ^ (fr & 7)) << 4), sw1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    const unsigned w1_row = lds0 + (unsigned)((wn * 64 + fr) * 128);              // + slot * F_SLOT + t * 2048 (t: h0, g0, h1, g1) + sw
    const unsigned w2_row = lds0 + (unsigned)((5 + wn) * F_SLOT + fr * 128);      // + jt * 2048 + sw
    const unsigned h_rd = lds0 + R_H + (unsigned)((wm * 32 + fr) * 128);          // + i * 2048 + sw
    const unsigned bias_rd = lds0 + R_BIAS + (unsigned)(wv * 512 + (wn * 64 + fq * 4) * 2);   // + u * 64 ; gate at + 32 ; + (j & 1) * 256
    unsigned h_wr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int hc = (wn * 2 + u) * 16 + fq * 4;
        h_wr[u] = lds0 + R_H + (unsigned)((wm * 32 + fr) * 128) + (unsigned)((((hc >> 3) ^ (fr & 7)) << 4) + (hc & 7) * 2);   // + i * 2048
    }

    for (int j = 0; j < nchunks; ++j) {
        float4v S[2][4];                  // [row tile][h0, g0, h1, g1]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) S[i][t] = (float4v){0.f, 0.f, 0.f, 0.f};
        half4e bh[2], bg[2];
        const bool more = j + 1 < nchunks;
        half8 b[2][4];
#define R_RD(BUF, T)                                                                                                     \
        {                                                                                                                \
            const unsigned wa_ = w1_row + (unsigned)(((T) >> 1) * F_SLOT) + (((T) & 1) ? sw1 : sw0);                     \
            DS_READ128(b[BUF][0], wa_, 0); DS_READ128(b[BUF][1], wa_, 2048); DS_READ128(b[BUF][2], wa_, 4096); DS_READ128(b[BUF][3], wa_, 6144); \
        }
#define R_MF(BUF, T, CNT)                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(b[BUF][0]), "+v"(b[BUF][1]), "+v"(b[BUF][2]), "+v"(b[BUF][3]));    \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) S[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[BUF][t], xf[i][T], S[i][t], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
        // I1: k-tiles 0-2 have landed once only k-tiles 3, 4 (issued after them) may still be in flight
        wait_vm(4);
        __builtin_amdgcn_s_barrier();
        R_RD(0, 0) R_RD(1, 1)
        issue_w2(j);
        R_MF(0, 0, 4) R_RD(0, 2) R_MF(1, 1, 4) R_RD(1, 3) R_MF(0, 2, 4) R_RD(0, 4) R_MF(1, 3, 4) R_RD(1, 5) R_MF(0, 4, 4) R_MF(1, 5, 0)
        // I2: k-tiles 3, 4: behind them this chunk's W2
        wait_vm(2 * nbw);
        __builtin_amdgcn_s_barrier();
        R_RD(0, 6) R_RD(1, 7)
        if (more) { issue_w1(j + 1, 0); issue_w1(j + 1, 1); issue_w1(j + 1, 2); }
        R_MF(0, 6, 4) R_RD(0, 8) R_MF(1, 7, 4) R_RD(1, 9) R_MF(0, 8, 4) R_MF(1, 9, 0)
#undef R_RD
#undef R_MF
        // ---- gate (GEGLU.forward), as k_ffn320: fp16-rounded projection outputs, packed fp32 GELU, h as the k-tile image of phase 2
        {   // the chunk's bias values (landed with its first stage), read only now: no registers held across the five stages
            const unsigned ba = bias_rd + (unsigned)((j & 1) * 256);
            DS_READ64(bh[0], ba, 0); DS_READ64(bg[0], ba, 32); DS_READ64(bh[1], ba, 64); DS_READ64(bg[1], ba, 96);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bg[0]), "+v"(bh[1]), "+v"(bg[1]));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half4e o;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const syn3r_f2 hv = (syn3r_f2){(float)(_Float16)(S[i][2 * u][r] + (float)bh[u][r]), (float)(_Float16)(S[i][2 * u][r + 1] + (float)bh[u][r + 1])};
                    const syn3r_f2 gv = (syn3r_f2){(float)(_Float16)(S[i][2 * u + 1][r] + (float)bg[u][r]), (float)(_Float16)(S[i][2 * u + 1][r + 1] + (float)bg[u][r + 1])};
                    const syn3r_f2 y = hv * gelu_pk(gv);
                    o[r] = (_Float16)y.x; o[r + 1] = (_Float16)y.y;
                }
                DS_WRITE64(h_wr[u] + (unsigned)(i * 2048), o);
            }
        wait_vm(more ? 7 : 0);            // I3: both W2 halves have landed: behind them the next chunk's k-tiles 0-2 (and its bias line)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // h is written
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: out[32 x 160] += h[32 x 64] . W2half[160 x 64]^T
        {
            half8 af[2][2], bf[2][5];
            DS_READ128(af[0][0], h_rd + sw0, 0); DS_READ128(af[1][0], h_rd + sw0, 2048);
            DS_READ128(bf[0][0], w2_row + sw0, 0); DS_READ128(bf[0][1], w2_row + sw0, 2048); DS_READ128(bf[0][2], w2_row + sw0, 4096);
            DS_READ128(bf[0][3], w2_row + sw0, 6144); DS_READ128(bf[0][4], w2_row + sw0, 8192);
            DS_READ128(bf[1][0], w2_row + sw0, 10240); DS_READ128(bf[1][1], w2_row + sw0, 12288); DS_READ128(bf[1][2], w2_row + sw0, 14336);
            DS_READ128(bf[1][3], w2_row + sw0, 16384); DS_READ128(bf[1][4], w2_row + sw0, 18432);
            if (more) { issue_w1(j + 1, 3); issue_w1(j + 1, 4); }   // into the slots of this chunk's k-tiles 3, 4
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(af[0][0]), "+v"(af[1][0]), "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[0][3]), "+v"(bf[0][4]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jt = 0; jt < 5; ++jt) acc[i][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[0][jt], af[i][0], acc[i][jt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            DS_READ128(af[0][1], h_rd + sw1, 0); DS_READ128(af[1][1], h_rd + sw1, 2048);
            half8 bg0[5];
            DS_READ128(bg0[0], w2_row + sw1, 0); DS_READ128(bg0[1], w2_row + sw1, 2048); DS_READ128(bg0[2], w2_row + sw1, 4096);
            DS_READ128(bg0[3], w2_row + sw1, 6144); DS_READ128(bg0[4], w2_row + sw1, 8192);
            asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2]), "+v"(bf[1][3]), "+v"(bf[1][4]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jt = 0; jt < 5; ++jt) acc[i][5 + jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[1][jt], af[i][0], acc[i][5 + jt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            half8 bg1[5];
            DS_READ128(bg1[0], w2_row + sw1, 10240); DS_READ128(bg1[1], w2_row + sw1, 12288); DS_READ128(bg1[2], w2_row + sw1, 14336);
            DS_READ128(bg1[3], w2_row + sw1, 16384); DS_READ128(bg1[4], w2_row + sw1, 18432);
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(af[0][1]), "+v"(af[1][1]), "+v"(bg0[0]), "+v"(bg0[1]), "+v"(bg0[2]), "+v"(bg0[3]), "+v"(bg0[4]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jt = 0; jt < 5; ++jt) acc[i][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bg0[jt], af[i][1], acc[i][jt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bg1[0]), "+v"(bg1[1]), "+v"(bg1[2]), "+v"(bg1[3]), "+v"(bg1[4]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jt = 0; jt < 5; ++jt) acc[i][5 + jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bg1[jt], af[i][1], acc[i][5 + jt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every wavefront is done with the ring before the epilogue stages through it

    // ---- epilogue (the arithmetic of gemm_epilogue): fp16((acc + bias) * s_acc) staged per wavefront [32 rows x 160 columns], then
    // + s_res * residual (+ res_add first, an fp16 add) + s_aux * aux on whole 16-byte chunks
    {
        const int gm0 = m0 + wm * 32, gn0 = wn * 160;
        __half* st = (__half*)smem_raw + wv * (32 * 168);            // 168-half rows (padded): 10,752 B per wavefront
        half8 res[10], radd[10];
        if (p.residual) {
#pragma unroll
            for (int it = 0; it < 10; ++it) {
                const int qi = lane + it * 64;
                const int row = qi / 20, ch = qi - row * 20;
                const int m = gm0 + row, n = gn0 + ch * 8;
                if (m < p.M) {
                    res[it] = *(const half8*)(p.residual + (long long)m * p.ldr + n);
                    if (p.res_add) radd[it] = *(const half8*)(p.res_add + (long long)(m / p.res_add_rpv) * p.N + n);
                } else res[it] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int jt = 0; jt < 10; ++jt) {
            const int n = gn0 + jt * 16 + fq * 4;
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                const half4e b = *(const half4e*)(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = (float)b[r];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                half4e o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)((acc[i][jt][r] + b4[r]) * p.s_acc);
                *(half4e*)(st + (i * 16 + fr) * 168 + jt * 16 + fq * 4) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int qi = lane + it * 64;
            const int row = qi / 20, ch = qi - row * 20;
            const int m = gm0 + row, n = gn0 + ch * 8;
            if (m >= p.M) continue;
            half8 v = *(const half8*)(st + row * 168 + ch * 8);
            if (p.residual || p.aux) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
                if (p.residual) {
                    half8 rr = res[it];
                    if (p.res_add) rr = rr + radd[it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += p.s_res * (float)rr[e];
                }
                if (p.aux) {
                    const half8 av = *(const half8*)(p.aux + (long long)m * p.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += p.s_aux * (float)av[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)f[e];
            }
            *(half8*)(p.out + (long long)m * p.ldc + n) = v;
        }
    }
}

int launch_ffn320(const FfnParams& q, hipStream_t stream) {
    static DevOnce once;
    int rc = set_max_lds(once, (const void*)k_ffn320r, R_LDS, "hipFuncSetAttribute(ffn320r)");
    if (rc) return rc;
    const int tiles = (q.e.M + F_BM - 1) / F_BM;
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_ffn320[M%d,D%d,e%d]", q.e.M, q.D, q.e.residual != nullptr);
        else snprintf(name, sizeof(name), "k_gemm_ffn320");
    }
    SYN3R_LAUNCH_NAMED(name, k_ffn320r, dim3(tiles), dim3(512), R_LDS, stream, q);
    SYN3R_LAUNCH_CHECK("ffn320r launch");
    return SYN3R_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm + bias-free projection for C = 320 in ONE kernel: `norm1(hidden_states)` -> `attn1.to_q / to_k / to_v` of the level-0
// transformer blocks (attention.py:340-352, 509-512; the three projections are stored as one [960, 320] matrix).  The two-launch
// path writes the normalised activation ([M, 320] fp16) and reads it back, and its contraction (K = 320: five k-tiles per tile,
// k_gemm_w128) runs at a quarter of the matrix peak.  Here, as in k_ffn320, a block owns 128 rows: the x tile (80 KB) is DMA'd
// into LDS once, normalised in place (ln_tile320: the arithmetic of k_layernorm<8>), and the output columns are walked in chunks
// of 320 - the weight chunk streams through a 3-slot ring in [320 x 32] stages (20 KB; 20 MFMAs per wavefront and barrier),
// eight wavefronts (2 x 4) of 64 rows x 80 columns each.  A chunk's [128 x 320] result goes out through a per-wavefront staging
// buffer (16 rows at a time, wavefront-local synchronisation only: whole 160-byte row runs per store) while the next chunk's
// first stages are already in flight.
// LDS: x 80 KB | ring 3 x 20 KB | staging 8 x 2.5 KB = 163,840 B.
// (Tried and measured slower on the same shapes, profiles/r04/lnqkv_ab.txt: a 4-slot ring with the fragments double-buffered in
// registers and an LDS-free epilogue by v_permlane16_swap - its 64-byte row segments cost more than the k-loop gained; the chunk's
// stores interleaved into the next chunk's k-loop.)
constexpr int Q_SLOT = F_C * 32 * 2;                  // 20,480: [320 rows x 32 k] of the weight chunk, 64-byte rows
constexpr int Q_RING = F_X_BYTES;
constexpr int Q_ST = Q_RING + 3 * Q_SLOT;             // 143,360
constexpr int Q_ST_WAVE = 16 * WN * 2;                // 2,560: 16 rows x 80 columns
constexpr int Q_LDS = Q_ST + 8 * Q_ST_WAVE;           // 163,840

struct LnLinParams {
    const __half* x; long long ldx;      // [M, 320]
    const __half* W;                     // [N, 320], N a multiple of 320
    __half* out; long long ldc;          // [M, N]
    int M, N, C;                         // C = 320 (run-time copy: the LayerNorm's divisor)
    const __half* ln_g; const __half* ln_b; float ln_eps;
};

__global__ void __launch_bounds__(512, 2) k_lnlin320(LnLinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;
    const int tiles_m = (p.M + F_BM - 1) / F_BM;
    const int m0 = (int)xcd_remap(blockIdx.x, (unsigned)tiles_m) * F_BM;
    const int nstage = (p.N / F_C) * 10;
    const bool full = m0 + F_BM <= p.M;               // every output store of the block is issued: exact vmcnt bookkeeping

    // weight stage DMA: a wave-instruction moves 16 rows x 64 B; lane -> (row, 16-byte slot), source chunk swizzled (h_swz)
    const int wprow = lane >> 2;
    const int wcsrc = (lane & 3) ^ h_swz(wprow);
    const int nbw = wv < 4 ? 3 : 2;
    const int b_first = wv < 4 ? wv * 3 : 12 + (wv - 4) * 2;
    const __half* w_lane = p.W + (long long)(b_first * 16 + wprow) * F_C + wcsrc * 8;
    char* const ring = smem_raw + Q_RING;
    int ig = 0, islot = 0;
    auto issue_next = [&]() {
        if (ig >= nstage) return;
        const int c = ig / 10, ks = ig - c * 10;
        const __half* src = w_lane + (long long)c * (F_C * F_C) + ks * 32;
        char* slot = ring + islot * Q_SLOT;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < nbw)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + i * 16 * F_C), (lds_void_t*)(slot + (b_first + i) * 1024), 16, 0, 0);
        ++ig;
        if (++islot == 3) islot = 0;
    };

    // ---- prologue: the x tile (as k_ffn320) and the first two weight stages, then the LayerNorm in place
    {
        const int prow = lane >> 3;
        const int csrc = (lane & 7) ^ prow;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = m0 + (wv * 2 + i) * 8 + prow;
                m = m < p.M ? m : p.M - 1;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(p.x + (long long)m * p.ldx + kt * BK + csrc * 8),
                                                 (lds_void_t*)(smem_raw + kt * 16384 + (wv * 2 + i) * 1024), 16, 0, 0);
            }
    }
    issue_next();
    issue_next();
    ln_tile320(smem_raw, tid, m0, p.M, (float)p.C, p.ln_g, p.ln_b, p.ln_eps, nullptr, 1);   // waits for every DMA above, ends on a barrier

    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const unsigned sw0 = (unsigned)(((0 + fq) ^ (fr & 7)) << 4), sw1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    const unsigned x_row = lds0 + (unsigned)((wm * 64 + fr) * 128);                       // + kt * 16384 + i * 2048 + sw
    const unsigned w_row = (unsigned)((wn * 80 + fr) * 64) + (unsigned)((fq ^ h_swz(fr)) << 4);   // inside a slot, + j * 1024
    const unsigned st_base = lds0 + Q_ST + (unsigned)(wv * Q_ST_WAVE);
    const unsigned st_wr = st_base + (unsigned)(fr * (WN * 2) + fq * 8);                  // + j * 32
    typedef _Float16 half4e __attribute__((ext_vector_type(4)));

    int cslot = 0, g = 0;
    const int nchunks = p.N / F_C;
    for (int c = 0; c < nchunks; ++c) {
        float4v acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 10; ++ks, ++g) {
            // Stage g has landed once only what was issued AFTER its DMA is still in flight (vmcnt retires in order): the next
            // stage's DMA (3 or 2 instructions per wavefront) and, in the first two stages after a chunk's stores (12 per wavefront;
            // a block with rows past M may skip store instructions and counts none: it then waits for the stores too), those.
            if (g + 1 >= nstage) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (ks < 2 && c > 0 && full) {
                if (wv < 4) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            } else {
                if (wv < 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            const unsigned xa = x_row + (unsigned)((ks >> 1) * 16384) + ((ks & 1) ? sw1 : sw0);
            const unsigned wa = lds0 + Q_RING + (unsigned)(cslot * Q_SLOT) + w_row;
            half8 a[TM], b[TN];
            DS_READ128(a[0], xa, 0); DS_READ128(a[1], xa, 2048); DS_READ128(a[2], xa, 4096); DS_READ128(a[3], xa, 6144);
            DS_READ128(b[0], wa, 0); DS_READ128(b[1], wa, 1024); DS_READ128(b[2], wa, 2048); DS_READ128(b[3], wa, 3072);
            DS_READ128(b[4], wa, 4096);
            issue_next();                 // stage g + 2 into the slot every wavefront finished reading before this barrier
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]));
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[0], a[i], acc[i][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b[1]));
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[1], a[i], acc[i][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b[2]));
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[2], a[i], acc[i][2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[3]));
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[3], a[i], acc[i][3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[4]));
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[4], a[i], acc[i][4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (++cslot == 3) cslot = 0;
        }
        // ---- the chunk's 64 x 80 block of this wavefront: 16 rows at a time through its own staging buffer (no block barrier;
        // acc[i][j][r] = C[row i*16 + (lane & 15)][col j*16 + (lane >> 4)*4 + r], see gemm_epilogue); every lane executes every LDS
        // instruction (an inline-asm output written under a divergent branch would be merged before its data has arrived)
        __half* const orow = p.out + (long long)c * F_C + wn * WN;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                half4e o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)acc[i][j][r];
                DS_WRITE64(st_wr + (unsigned)(j * 32), o);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            half8 v[3];
            DS_READ128(v[0], st_base + (unsigned)(lane * 16), 0);
            DS_READ128(v[1], st_base + (unsigned)(lane * 16), 1024);
            DS_READ128(v[2], st_base + (unsigned)((lane & 31) * 16), 2048);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int q = lane + it * 64;
                const int row = q / (WN / 8), ch = q - row * (WN / 8);
                const int m = m0 + wm * WM + i * 16 + row;
                if (q < 16 * (WN / 8) && m < p.M) *(half8*)(orow + (long long)m * p.ldc + ch * 8) = v[it];
            }
        }
    }
}

int launch_lnlin320(const LnLinParams& p, hipStream_t stream) {
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_lnlin320, (int)(Q_LDS), "hipFuncSetAttribute(lnlin320)")) return rc;
    const int tiles = (p.M + F_BM - 1) / F_BM;
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_lnlin320[M%d,N%d]", p.M, p.N);
        else snprintf(name, sizeof(name), "k_gemm_lnlin320");
    }
    SYN3R_LAUNCH_NAMED(name, k_lnlin320, dim3(tiles), dim3(512), Q_LDS, stream, p);
    SYN3R_LAUNCH_CHECK("lnlin320 launch");
    return SYN3R_OK;
}

// grid of the persistent kernels: the CU count of the CURRENT device (queried once per device; SYN3R_PERSISTENT_BLOCKS overrides
// in tuning builds)
int persistent_blocks() {
    static std::atomic<int> cus_of[64];
    static const int forced = tune_env("SYN3R_PERSISTENT_BLOCKS", 0);
    if (forced > 0) return forced < 8 ? 8 : forced;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cus_of[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        if (n < 8) n = 8;
        cus_of[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Tile columns per band of the persistent 256 x 320 kernels' tile order (bands of `band` tile columns, row-major inside a band: the
// 32 tiles an XCD holds at a time are ~32 / band rows x band columns).  A narrower band re-reads A more often from beyond the L2
// (once per band) but keeps the band's weight panel (band x 320 x K x 2 B) well inside the 4 MB L2.  Measured
// (tools/band_ab.sh and tools/gemm_ab.py, profiles/r04/band_width.txt): with more than four tile columns and K = 640, 3 beats 4 by
// 4-7 % both isolated and inside the unit ([64512,5120,640]: 564 -> 541 us isolated, 7.35 -> 6.85 ms per unit over its 15
// calls); at K = 1280 the isolated gain ([16128,10240,1280] 432 -> 392 us) does not survive inside the unit (+0.5 %, and +4 % on
// [4032,10240,1280]), and with at most four tile columns one band (A read once) is best.  SYN3R_Z_BAND overrides (tuning).
int band_width(const GemmParams& p) {
    static const int band_env = tune_env("SYN3R_Z_BAND", 0);
    if (band_env > 0) return band_env;
    return ((p.N + WBN - 1) / WBN > 4 && p.K <= 640) ? 3 : 4;
}

int launch_widep(const GemmParams& p, hipStream_t stream) {
    constexpr size_t lds = (size_t)2 * W_STAGE + 16384;   // 163,840 B: the ring + the tail the epilogue staging runs into
    static_assert(8 * WM * EPI_LD * sizeof(__half) <= lds - W_STAGE, "epilogue staging must fit behind ring slot 0");
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_gemm_widep, (int)(lds), "hipFuncSetAttribute(gemm_widep)")) return rc;
    int tiles = ((p.M + WBM - 1) / WBM) * ((p.N + WBN - 1) / WBN);
    const int blocks = std::min(tiles, persistent_blocks());
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_widep[M%d,N%d,K%d,e%d]", p.M, p.N, p.K, p.geglu_D > 0 ? 2 : (p.residual != nullptr));
        else snprintf(name, sizeof(name), "k_gemm_widep");
    }
    GemmParams q = p;
    q.band = band_width(p);
    SYN3R_LAUNCH_NAMED(name, k_gemm_widep, dim3(blocks), dim3(512), lds, stream, q);
    SYN3R_LAUNCH_CHECK("gemm_widep launch");
    return SYN3R_OK;
}

template <int MODE = MODE_DENSE>
int launch_z(const GemmParams& p, hipStream_t stream) {
    static DevOnce once, once2;
    if (int rc = set_max_lds(once, (const void*)k_gemm_z<false, MODE>, Z_LDS, "hipFuncSetAttribute(gemm_z)")) return rc;
    if constexpr (MODE == MODE_DENSE) { if (int rc = set_max_lds(once2, (const void*)k_gemm_z<true, MODE_DENSE>, Z_LDS, "hipFuncSetAttribute(gemm_z)")) return rc; }
    if constexpr (MODE == MODE_CONV2D) { if (int rc = set_max_lds(once2, (const void*)k_gemm_z<false, MODE_CONV2D, true>, Z_LDS, "hipFuncSetAttribute(gemm_z)")) return rc; }
    int tiles = ((p.M + WBM - 1) / WBM) * ((p.N + WBN - 1) / WBN);
    const int blocks = std::min(tiles, persistent_blocks());
    GemmParams q = p;
    q.band = band_width(p);
    const GemmParams& p_ = q;
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_z<%d>[M%d,N%d,K%d,e%d]", MODE, p.M, p.N, p.K, p.geglu_D > 0 ? 2 : (p.residual != nullptr));
        else snprintf(name, sizeof(name), "k_gemm_z<%d>", MODE);
    }
    if constexpr (MODE == MODE_DENSE) {
        if (p.A2) { SYN3R_LAUNCH_NAMED(name, (k_gemm_z<true, MODE_DENSE>), dim3(blocks), dim3(512), Z_LDS, stream, p_); SYN3R_LAUNCH_CHECK("gemm_z launch"); return SYN3R_OK; }
    }
    if constexpr (MODE == MODE_CONV2D) {
        if (p.ups) { SYN3R_LAUNCH_NAMED(name, (k_gemm_z<false, MODE_CONV2D, true>), dim3(blocks), dim3(512), Z_LDS, stream, p_); SYN3R_LAUNCH_CHECK("gemm_z launch"); return SYN3R_OK; }
    }
    SYN3R_LAUNCH_NAMED(name, (k_gemm_z<false, MODE>), dim3(blocks), dim3(512), Z_LDS, stream, p_);
    SYN3R_LAUNCH_CHECK("gemm_z launch");
    return SYN3R_OK;
}

// Kernel family forced by the CALLING THREAD (syn3r_gemm_set_tile, see launch_dma); thread_local: no state shared between host threads
thread_local int g_dma_bm = 0;
// Split-K scratch of the CALLING THREAD (syn3r_gemm_set_splitk_workspace): null = no split-K
thread_local void* g_splitk_ws = nullptr;
thread_local size_t g_splitk_bytes = 0;

// Which of the two persistent 256 x 320 kernels: measured inside the UNet unit on one box (tools/gemm_ab.py SYN3R_GEMM_Z 0 1,
// profiles/r04/gemm_z_ab.txt) the software-pipelined k_gemm_z is 1.6..3.5 % faster on the gated projections and 1..5 %
// slower on the residual-add ones, a wash on their sum (74.33 against 74.23 ms): it takes the gated shapes.
// SYN3R_GEMM_Z overrides (tuning): 0 = never, 1 = every shape the 256 x 320 tile is chosen for.
int wide_launch(const GemmParams& p, hipStream_t stream) {
    static const int z_env = tune_env("SYN3R_GEMM_Z", -1);
    const bool z = g_dma_bm == -322 ? true : (g_dma_bm == -320 ? false : (z_env < 0 ? p.geglu_D > 0 : z_env != 0));
    return z ? launch_z<MODE_DENSE>(p, stream) : launch_widep(p, stream);
}

// Does the persistent 256 x 320 kernel take this contraction?  Lean epilogue (no row vector together with a gate, aux only
// with a residual, a gate of whole 16-byte chunks and s_acc = 1), 32-bit byte offsets inside the operands, M and N
// multiples of 8.  Everything else goes to the 160-column kernels, whose epilogue is general.
bool widep_admits(const GemmParams& p) {
    const bool lean = (!p.rowvec || p.geglu_D <= 0) && (!p.aux || p.residual) && (p.geglu_D <= 0 || (p.geglu_D % 8 == 0 && p.s_acc == 1.0f));
    const bool small = (p.a_tiled ? (long long)((p.M + 127) / 128) * 128 * p.K : (long long)p.M * p.lda) < (1ll << 31) &&
                       (long long)p.N * p.K < (1ll << 31) && (!p.A2 || (long long)p.M * p.lda2 < (1ll << 31));
    return lean && small && p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8;
}

// Kernel family forced by the CALLING THREAD (syn3r_gemm_set_tile; tests and tuning tools): 0 = by shape, 128 / 256 = the
// 160-column LDS-DMA kernel of that block height, -320 = the persistent 256 x 320 kernel k_gemm_widep wherever it admits the
// shape, -322 = the software-pipelined 256 x 320 kernel k_gemm_z (dense, two-source, convolutions).  thread_local: no state shared between host threads.

template <int MODE>
int launch_dmap(const GemmParams& p, hipStream_t stream) {
    constexpr size_t lds = (size_t)3 * (256 * BK * 2 + DMA_B_BYTES);   // 159,744 B
    static_assert(8 * 32 * EPI_LD * sizeof(__half) <= 256 * BK * 2 + DMA_B_BYTES, "the two-pass epilogue staging must fit in one ring slot");
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_gemm_dmap<MODE>, (int)(lds), "hipFuncSetAttribute(gemm_dmap)")) return rc;
    const int tiles = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
    const int blocks = std::min(tiles, persistent_blocks());
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_dmap<%d>[M%d,N%d,K%d,e%d]", MODE, p.M, p.N, p.K, p.residual != nullptr);
        else snprintf(name, sizeof(name), "k_gemm_dmap<%d>", MODE);
    }
    GemmParams q = p;
    q.tc_pb = q.tc_nf = 0;
    if constexpr (MODE == MODE_TCONV) {
        static const int tc_env = tune_env("SYN3R_TCONV_ORDER", 1);       // 0: rows in memory order (tuning builds)
        if (tc_env != 0 && p.HW % 256 == 0 && p.M % p.HW == 0 && p.M / p.HW > 1) { q.tc_pb = p.HW / 256; q.tc_nf = p.M / p.HW; }
    }
    SYN3R_LAUNCH_NAMED(name, (k_gemm_dmap<MODE>), dim3(blocks), dim3(512), lds, stream, q);
    SYN3R_LAUNCH_CHECK("gemm_dmap launch");
    return SYN3R_OK;
}

template <int MODE, int BM>
int launch_dma_bm(const GemmParams& p, hipStream_t stream) {
    if constexpr (BM == 256) {
        static const int pers_env = tune_env("SYN3R_DMA_PERSISTENT", 1);       // 0: the one-tile-per-block kernel (tuning builds)
        // the lean epilogue: no GEGLU gate, aux only together with a residual, whole 16-byte chunks of columns
        // measured inside the UNet unit, same box (SYN3R_DMA_PERSISTENT=0 against the default): dense K = 320 residual
        // projections -11 %, temporal convolutions -1..-5 %, 3x3 convolutions +1..+3 % (their k-loops are 45-360 k-tiles
        // long: nothing to hide at a tile boundary, and the cursor's bookkeeping is in the loop) -> those keep one tile per block
        if (pers_env != 0 && MODE != MODE_CONV2D && p.geglu_D <= 0 && (!p.aux || p.residual) && p.N % 8 == 0 && p.N >= 8)
            return launch_dmap<MODE>(p, stream);
    }
    constexpr size_t lds = (size_t)(BM == 256 ? 3 : 2) * (BM * BK * 2 + DMA_B_BYTES);   // 159,744 B / 73,728 B
    static_assert((BM / 32) * WM * EPI_LD * sizeof(__half) <= lds, "epilogue staging must fit in the ring");
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_gemm_dma<MODE, BM>, (int)lds, "hipFuncSetAttribute(gemm_dma)")) return rc;
    int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_dma<%d,%d>[M%d,N%d,K%d,e%d]", MODE, BM, p.M, p.N, p.K, p.geglu_D > 0 ? 2 : (p.residual != nullptr));
        else snprintf(name, sizeof(name), "k_gemm_dma<%d,%d>", MODE, BM);
    }
    SYN3R_LAUNCH_NAMED(name, (k_gemm_dma<MODE, BM>), dim3(tiles), dim3(BM * 2), lds, stream, p);
    SYN3R_LAUNCH_CHECK("gemm_dma launch");
    return SYN3R_OK;
}

// Split-K for the contractions whose grid leaves most of the chip idle (level 3 of the UNet at F = 14: M = 4 032 rows = 128
// tiles of 256 x 160 on 256 CUs, and doubling the rows costs such a launch only +16 % time): S = 2 or 4 equal K parts so that
// tiles x S fills one round of the chip, fp32 partial tiles in the caller's workspace (syn3r_gemm_set_splitk_workspace, per
// calling thread), summed in order by k_splitk_finish.  Without a workspace, or when it is too small: *done stays false and the
// caller launches one pass.  Two-source A: only when no part straddles the boundary between the sources.
template <int MODE>
int launch_splitk(const GemmParams& p, hipStream_t stream, bool* done) {
    const long long tiles256 = (long long)((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
    const int nkt_all = p.K / BK;
    const int S = (tiles256 * 4 <= 256 && nkt_all % 4 == 0 && nkt_all >= 16) ? 4 : ((tiles256 * 2 <= 256 && nkt_all % 2 == 0 && nkt_all >= 8) ? 2 : 1);
    const size_t need = (size_t)S * p.M * p.N * sizeof(float);
    if (S == 1 || g_dma_bm != 0 || !g_splitk_ws || need > g_splitk_bytes || p.N % 8 != 0 || p.M < 8 || p.relu || p.relu_mask ||
        p.geglu_D > 0 || p.out_tiled || (p.A2 && (MODE != MODE_DENSE || p.a_tiled || (p.K1 / BK) % (nkt_all / S) != 0)))
        return SYN3R_OK;
    constexpr size_t lds = (size_t)3 * (256 * BK * 2 + DMA_B_BYTES);
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_gemm_dma<MODE, 256>, (int)lds, "hipFuncSetAttribute(gemm_dma)")) return rc;
    GemmParams q = p;
    q.ksplit = S; q.split_ws = (float*)g_splitk_ws;
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_dma<%d,256>/%d[M%d,N%d,K%d,e%d]", MODE, S, p.M, p.N, p.K, p.residual != nullptr);
        else snprintf(name, sizeof(name), "k_gemm_dma<%d,256>/k", MODE);
    }
    SYN3R_LAUNCH_NAMED(name, (k_gemm_dma<MODE, 256>), dim3((unsigned)(tiles256 * S)), dim3(512), lds, stream, q);
    const long long chunks = (long long)p.M * (p.N / 8);
    SYN3R_LAUNCH(k_splitk_finish, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, q);
    SYN3R_LAUNCH_CHECK("gemm split-K launch");
    *done = true;
    return SYN3R_OK;
}

template <int MODE>
int launch_dma(const GemmParams& p, hipStream_t stream) {
    static const int wide_env = tune_env("SYN3R_GEMM_WIDE", -1);       // -1 = by shape, 0 = never, 1 = always (tuning builds)
    const bool wide_ok = MODE == MODE_DENSE && widep_admits(p);
    if (g_dma_bm == -320 && wide_ok) return launch_widep(p, stream);                              // syn3r_gemm_set_tile(-320)
    if (g_dma_bm == -322 && wide_ok) return launch_z<MODE_DENSE>(p, stream);                      // syn3r_gemm_set_tile(-322)
    if (g_dma_bm == 0 && wide_env != 0 && wide_ok) {
        // measured on MI355X inside the UNet (tools/gemm_ab.py, same box): the 256 x 320 tile is 7..14 % faster on
        // the dense contractions whenever its tiles fill the 256 CUs (last round >= 80 % full), except the
        // residual-add projections with K <= 320, whose time is their epilogue;
        // the implicit-GEMM convolutions are within 3 % either way and keep the 160-column kernel
        const long long tiles = (long long)((p.M + WBM - 1) / WBM) * ((p.N + WBN - 1) / WBN);
        const long long rounds = (tiles + 255) / 256;
        const bool fills = tiles * 10 >= rounds * 256 * 8;
        // (round 2: with the persistent kernel's epilogue - row vector, all ten residual requests in flight at once - the
        // K = 640 / 1280 residual projections are 3..4 % faster on the wide tile; K = 320 stays 4 % slower there)
        const bool short_residual = p.residual != nullptr && p.K <= 320;
        if (wide_env == 1 || (fills && !short_residual)) {
            // (the 128 x 320 two-blocks-per-CU variant that took the K <= 320, N > 640 shapes until round 4 is gone: the level-0
            // feed-forward and q / k / v projections it was built for run in k_ffn320r / k_lnlin320)
            return wide_launch(p, stream);
        }
    }
    // Convolutions on the 256 x 320 tile (round 4, k_gemm_z<MODE>): lean epilogue (no ReLU options, aux only with a residual),
    // whole 16-byte column chunks, 32-bit byte offsets into the input, and a grid that fills the chip as the
    // dense rule above asks (M = 4 032 at level 3 gives 64 tiles: +110 % there - those stay on the 160-column tile, 128 blocks).
    // Measured inside the UNet unit, same box (tools/gemm_ab.py SYN3R_CONV_Z 0 1, profiles/r04/conv_z_ab.txt), with the filter
    // taps innermost in K: 3x3 convolutions -4..-12 % (isolated +8..+16 %, 1 165-1 352 TFLOP/s), the temporal convolutions +-2 %
    // without and +12 % with a residual (they stay), the 8-column output convolution +64 % (one 320-wide tile column for 8
    // columns: stays).  SYN3R_CONV_Z=0: never; 1: every admissible shape (tests, tuning).
    if constexpr (MODE != MODE_DENSE) {
        static const int cz_env = tune_env("SYN3R_CONV_Z", -1);
        const long long tiles = (long long)((p.M + WBM - 1) / WBM) * ((p.N + WBN - 1) / WBN);
        const long long rounds = (tiles + 255) / 256;
        const bool fills = tiles * 10 >= rounds * 256 * 8;
        const long long in_bytes = MODE == MODE_CONV2D ? (long long)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Cin * 2 : (long long)p.M * p.Cin * 2;
        static const int czu_env = tune_env("SYN3R_CONV_Z_UPS", 1);       // 0: the fused-upsample convolutions stay on the 160-column kernel (tuning builds)
        const bool lean = !p.relu && !p.relu_mask && (!p.aux || p.residual) && p.geglu_D <= 0 && !p.A2 && !p.a_tiled &&
                          !(p.ups && (p.stride != 1 || p.pad != 1 || (czu_env == 0 && g_dma_bm != -322)));
        // (p.M < 2^24: the pixel-index division of gemm_z.h is a float-reciprocal multiply with a +-1 correction, exact below that)
        const bool ok = lean && p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8 && p.M < (1 << 24) && in_bytes < (1ll << 32) - (1 << 20) &&
                        (long long)p.N * p.K < (1ll << 31) && p.Cin % BK == 0;
        const bool pays = MODE == MODE_CONV2D && p.N >= 320;
        if (g_dma_bm == -322 && ok) return launch_z<MODE>(p, stream);                             // syn3r_gemm_set_tile(-322)
        if (g_dma_bm == 0 && cz_env != 0 && ok && (cz_env == 1 || (fills && pays))) return launch_z<MODE>(p, stream);
    }
    // 256-row blocks (eight wavefronts, wavefronts 4-7 staggered by half a k-tile against their SIMD partners) against
    // two independent 128-row blocks per CU, measured inside the UNet unit on MI355X (tools/unet_breakdown.py with
    // SYN3R_SET_TILE=-256, round 2): every implicit-GEMM convolution and temporal convolution -1..-10 % (-3.7 % on their
    // sum), the residual-add projections -5..-6 %; only grids that leave CUs without a 256-row block (dense, M = 4032)
    // stay with the 128-row blocks.  (Before the stagger the 128-row pairs won everywhere but N >= 5120.)
    const long long tiles256 = (long long)((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
    bool split = false;
    int rc = launch_splitk<MODE>(p, stream, &split);
    if (split || rc) return rc;
    int bm = g_dma_bm > 0 ? g_dma_bm : ((MODE != MODE_DENSE || tiles256 >= 256) ? 256 : 128);
    return bm == 128 ? launch_dma_bm<MODE, 128>(p, stream) : launch_dma_bm<MODE, 256>(p, stream);
}

template <int MODE>
int launch(const GemmParams& p, hipStream_t stream) { return launch_dma<MODE>(p, stream); }

int check_common(const GemmParams& p, const char* who) {
    SYN3R_REQUIRE(p.A && p.W && p.out, "%s: null operand", who);
    SYN3R_REQUIRE(SYN3R_DIM_OK(p.M) && SYN3R_DIM_OK(p.N) && SYN3R_DIM_OK(p.K), "%s: bad sizes M=%d N=%d K=%d", who, p.M, p.N, p.K);
    SYN3R_REQUIRE(p.K % BK == 0, "%s: K=%d must be a multiple of %d", who, p.K, BK);
    SYN3R_REQUIRE(p.ldc % 8 == 0 && p.ldc >= p.N, "%s: ldc=%lld must be >= N and a multiple of 8", who, p.ldc);
    SYN3R_REQUIRE(!p.residual || (p.ldr % 8 == 0 && p.ldr >= p.N), "%s: bad residual stride", who);
    SYN3R_REQUIRE(!p.aux || (p.ldaux % 8 == 0 && p.ldaux >= p.N), "%s: bad aux stride", who);
    SYN3R_REQUIRE(!p.rowvec || (p.rows_per_vec != 0 && p.ldrv >= p.N), "%s: bad rowvec arguments", who);
    SYN3R_REQUIRE(p.rv_group >= 0 && (p.rv_group == 0 || p.rows_per_vec < 0), "%s: rv_group_rows needs rows_per_vec < 0", who);
    SYN3R_REQUIRE(((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out | (uintptr_t)p.residual | (uintptr_t)p.aux) % 16 == 0,
                  "%s: operands must be 16-byte aligned", who);
    return SYN3R_OK;
}

}  // namespace

extern "C" int syn3r_gemm_set_splitk_workspace(void* workspace, size_t bytes) {
    SYN3R_REQUIRE((workspace == nullptr) == (bytes == 0), "gemm_set_splitk_workspace: pointer and size must both be given or both be zero");
    SYN3R_REQUIRE(((uintptr_t)workspace % 16) == 0, "gemm_set_splitk_workspace: the workspace must be 16-byte aligned");
    g_splitk_ws = workspace;
    g_splitk_bytes = bytes;
    return SYN3R_OK;
}

extern "C" int syn3r_gemm_set_tile(int bm) {
    SYN3R_REQUIRE(bm == 0 || bm == -128 || bm == -256 || bm == -320 || bm == -322, "gemm_set_tile: bm must be 0, -128, -256, -320 or -322");
    g_dma_bm = (bm == -128 || bm == -256) ? -bm : bm;        // this thread's launches only (thread_local)
    return SYN3R_OK;
}

extern "C" int syn3r_gemm_2src_supported(int M, int N, int K1, int K2, long long lda1, long long lda2) {
    if (M < 8 || N < 8 || M % 8 || N % 8 || K1 <= 0 || K2 <= 0 || K1 % BK || K2 % BK) return 0;
    if ((long long)M * lda1 >= (1ll << 31) || (long long)M * lda2 >= (1ll << 31) || (long long)N * (K1 + K2) >= (1ll << 31)) return 0;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Skinny contraction, M <= 16 rows (time-embedding projections [B,1280], the folded Sk = 1 cross-attention
// context k/v [B,1024], the frame-position MLP [F,C]): ~100 launches per UNet forward whose whole cost is
// streaming the weight matrix once.  The tile kernels would put all of it on N/160 CUs; here a wavefront owns
// 2 output columns and splits K across its lanes (16-byte weight loads, x rows from L1/L2), so the weight
// stream is spread over N/8 workgroups.  HBM-bound: N*K*2 bytes.
constexpr int SKINNY_MAX_M = 16;
template <int MR>
__global__ void __launch_bounds__(256) k_gemm_skinny(GemmParams p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wv) * 2;
    if (n0 >= p.N) return;
    const bool two = n0 + 1 < p.N;
    const __half* w0 = p.W + (long long)n0 * p.K;
    const __half* w1 = p.W + (long long)(two ? n0 + 1 : n0) * p.K;
    float acc0[MR], acc1[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
    for (int k = lane * 8; k < p.K; k += 512) {
        const half8 a = *(const half8*)(w0 + k), b = *(const half8*)(w1 + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < p.M) {
                const half8 x = *(const half8*)(p.A + (long long)m * p.lda + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc0[m] += (float)x[e] * (float)a[e];
                    acc1[m] += (float)x[e] * (float)b[e];
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m < p.M) {
            float s0 = wave_sum(acc0[m]), s1 = wave_sum(acc1[m]);
            if (lane == 0) {
                float b0 = p.bias ? __half2float(p.bias[n0]) : 0.f;
                p.out[(long long)m * p.ldc + n0] = __float2half((s0 + b0) * p.s_acc);
                if (two) {
                    float b1 = p.bias ? __half2float(p.bias[n0 + 1]) : 0.f;
                    p.out[(long long)m * p.ldc + n0 + 1] = __float2half((s1 + b1) * p.s_acc);
                }
            }
        }
    }
}

int launch_skinny(const GemmParams& p, hipStream_t stream) {
    const int blocks = (p.N + 7) / 8;
    if (p.M <= 2) SYN3R_LAUNCH_NAMED("k_gemm_skinny<2>", k_gemm_skinny<2>, dim3(blocks), dim3(256), 0, stream, p);
    else if (p.M <= 8) SYN3R_LAUNCH_NAMED("k_gemm_skinny<8>", k_gemm_skinny<8>, dim3(blocks), dim3(256), 0, stream, p);
    else SYN3R_LAUNCH_NAMED("k_gemm_skinny<16>", k_gemm_skinny<16>, dim3(blocks), dim3(256), 0, stream, p);
    SYN3R_LAUNCH_CHECK("gemm_skinny launch");
    return SYN3R_OK;
}

extern "C" int syn3r_gemm_f16(const void* A, long long lda, const void* W, void* out, long long ldc, const void* bias,
                              const void* rowvec, long long ldrv, int rows_per_vec, int rv_group_rows, const void* residual,
                              long long ldr, const void* aux, long long ldaux, float s_acc, float s_res, float s_aux,
                              int M, int N, int K, void* stream) {
    GemmParams p{};
    p.A = (const __half*)A; p.lda = lda; p.W = (const __half*)W; p.out = (__half*)out; p.ldc = ldc;
    p.bias = (const __half*)bias; p.rowvec = (const __half*)rowvec; p.ldrv = ldrv; p.rows_per_vec = rows_per_vec;
    p.rv_group = rv_group_rows;
    p.residual = (const __half*)residual; p.ldr = ldr; p.aux = (const __half*)aux; p.ldaux = ldaux;
    p.s_acc = s_acc; p.s_res = s_res; p.s_aux = s_aux; p.M = M; p.N = N; p.K = K;
    int rc = check_common(p, "gemm_f16");
    if (rc) return rc;
    SYN3R_REQUIRE(lda % 8 == 0 && lda >= K, "gemm_f16: lda=%lld must be >= K and a multiple of 8", lda);
    if (M <= SKINNY_MAX_M && !p.rowvec && !p.residual && !p.aux && g_dma_bm == 0)
        return launch_skinny(p, (hipStream_t)stream);
    return launch<MODE_DENSE>(p, (hipStream_t)stream);
}

extern "C" int syn3r_gemm_2src_f16(const void* A1, long long lda1, int K1, const void* A2, long long lda2, int K2, const void* W,
                                   void* out, long long ldc, const void* bias, int M, int N, void* stream) {
    GemmParams p{};
    SYN3R_REQUIRE(A2 != nullptr && SYN3R_DIM_OK(K1) && SYN3R_DIM_OK(K2) && K1 % BK == 0 && K2 % BK == 0, "gemm_2src: K1=%d, K2=%d must be positive multiples of %d", K1, K2, BK);
    p.A = (const __half*)A1; p.lda = lda1; p.A2 = (const __half*)A2; p.lda2 = lda2; p.K1 = K1;
    p.W = (const __half*)W; p.out = (__half*)out; p.ldc = ldc; p.bias = (const __half*)bias;
    p.s_acc = 1.0f; p.s_res = 1.0f; p.s_aux = 1.0f; p.M = M; p.N = N; p.K = K1 + K2;
    int rc = check_common(p, "gemm_2src_f16");
    if (rc) return rc;
    SYN3R_REQUIRE(lda1 % 8 == 0 && lda1 >= K1 && lda2 % 8 == 0 && lda2 >= K2 && ((uintptr_t)A2 % 16) == 0, "gemm_2src: bad strides / alignment");
    // only the persistent 256 x 320 kernel reads two sources; syn3r_gemm_2src_supported() is its admission test
    SYN3R_REQUIRE(syn3r_gemm_2src_supported(M, N, K1, K2, lda1, lda2) != 0,
                  "gemm_2src: shape M=%d N=%d not served by the two-source kernel (concatenate and call syn3r_gemm_f16)", M, N);
    bool split = false;
    rc = launch_splitk<MODE_DENSE>(p, (hipStream_t)stream, &split);     // a grid of a quarter of the chip: the two sources as K parts
    if (split || rc) return rc;
    return wide_launch(p, (hipStream_t)stream);
}

extern "C" int syn3r_gemm_geglu_f16(const void* A, long long lda, const void* Wpacked, const void* bias_packed, void* out,
                                    long long ldc, int M, int D, int K, void* stream) {
    GemmParams p{};
    SYN3R_REQUIRE(SYN3R_DIM_OK(D), "gemm_geglu_f16: bad D=%d", D);
    const int tiles = (D + WN - 1) / WN;
    p.A = (const __half*)A; p.lda = lda; p.W = (const __half*)Wpacked; p.out = (__half*)out; p.ldc = ldc;
    p.bias = (const __half*)bias_packed; p.s_acc = 1.0f; p.M = M; p.N = tiles * BN; p.K = K; p.geglu_D = D;
    SYN3R_REQUIRE(ldc >= D && ldc % 8 == 0, "gemm_geglu_f16: ldc=%lld must be >= D and a multiple of 8", ldc);
    long long save = p.ldc;
    p.ldc = ((long long)p.N + 7) / 8 * 8 > p.ldc ? (long long)p.N : p.ldc;   // check_common compares ldc with the packed N
    int rc = check_common(p, "gemm_geglu_f16");
    if (rc) return rc;
    p.ldc = save;
    SYN3R_REQUIRE(lda % 8 == 0 && lda >= K, "gemm_geglu_f16: lda=%lld must be >= K and a multiple of 8", lda);
    return launch<MODE_DENSE>(p, (hipStream_t)stream);
}

extern "C" size_t syn3r_feedforward_workspace_bytes(int M, int D) {
    if (!SYN3R_DIM_OK(M) || !SYN3R_DIM_OK(D)) return 0;
    return (size_t)((M + 127) / 128) * 128 * (size_t)D * sizeof(__half);
}

extern "C" int syn3r_feedforward_f16(const void* x, long long ldx, const void* w1_packed, const void* b1_packed, int D,
                                     const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                                     long long ldr, const void* aux, long long ldaux, float s_acc, float s_res,
                                     float s_aux, int M, int C_in, int C_out, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    SYN3R_REQUIRE(x && w1_packed && b1_packed && w2 && out, "feedforward_f16: null operand");
    SYN3R_REQUIRE(M > 0 && D > 0 && D % BK == 0 && C_in > 0 && C_out > 0, "feedforward_f16: bad sizes M=%d D=%d C_in=%d C_out=%d (D must be a multiple of %d)",
                  M, D, C_in, C_out, BK);
    const size_t need = syn3r_feedforward_workspace_bytes(M, D);
    if (!workspace || workspace_bytes < need) {
        set_error("feedforward_f16: workspace %zu < %zu", workspace_bytes, need);
        return SYN3R_E_WORKSPACE;
    }
    SYN3R_REQUIRE((uintptr_t)workspace % 16 == 0, "feedforward_f16: workspace must be 16-byte aligned");
    // net.0 (GEGLU projection) -> the gated hidden activation in the A-tiled layout ...
    GemmParams p{};
    const int tiles = (D + WN - 1) / WN;
    p.A = (const __half*)x; p.lda = ldx; p.W = (const __half*)w1_packed; p.out = (__half*)workspace; p.ldc = (long long)tiles * BN;
    p.bias = (const __half*)b1_packed; p.s_acc = 1.0f; p.M = M; p.N = tiles * BN; p.K = C_in; p.geglu_D = D; p.out_tiled = 1;
    p.out_nt = need >= ((size_t)256 << 20);     // larger than the 256 MB memory-side cache: nothing of it would be re-read from there
    int rc = check_common(p, "feedforward_f16(net.0)");
    if (rc) return rc;
    SYN3R_REQUIRE(ldx % 8 == 0 && ldx >= C_in, "feedforward_f16: ldx=%lld must be >= C_in and a multiple of 8", ldx);
    rc = launch<MODE_DENSE>(p, (hipStream_t)stream);
    if (rc) return rc;
    // ... which net.2 reads as its A operand
    GemmParams q{};
    q.A = (const __half*)workspace; q.lda = D; q.a_tiled = 1; q.W = (const __half*)w2; q.out = (__half*)out; q.ldc = ldc;
    q.bias = (const __half*)b2; q.residual = (const __half*)residual; q.ldr = ldr; q.aux = (const __half*)aux; q.ldaux = ldaux;
    q.s_acc = s_acc; q.s_res = s_res; q.s_aux = s_aux; q.M = M; q.N = C_out; q.K = D;
    rc = check_common(q, "feedforward_f16(net.2)");
    if (rc) return rc;
    return launch<MODE_DENSE>(q, (hipStream_t)stream);
}

namespace {
int feedforward_fused(const void* x, long long ldx, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* w1_chunked,
                      const void* b1_chunked, int D, const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                      long long ldr, const void* aux, long long ldaux, float s_acc, float s_res, float s_aux, int M, int C, void* stream,
                      const void* addvec = nullptr, int rows_per_vec = 0);
}
extern "C" int syn3r_feedforward_fused_f16(const void* x, long long ldx, const void* w1_chunked, const void* b1_chunked,
                                           int D, const void* w2, const void* b2, void* out, long long ldc,
                                           const void* residual, long long ldr, const void* aux, long long ldaux,
                                           float s_acc, float s_res, float s_aux, int M, int C, void* stream) {
    return feedforward_fused(x, ldx, nullptr, nullptr, 0.f, w1_chunked, b1_chunked, D, w2, b2, out, ldc, residual, ldr, aux, ldaux,
                             s_acc, s_res, s_aux, M, C, stream);
}
extern "C" int syn3r_feedforward_fused_ln_f16(const void* x, long long ldx, const void* ln_gamma, const void* ln_beta, float ln_eps,
                                              const void* w1_chunked, const void* b1_chunked, int D, const void* w2, const void* b2,
                                              void* out, long long ldc, const void* residual, long long ldr, const void* aux,
                                              long long ldaux, float s_acc, float s_res, float s_aux, int M, int C, void* stream) {
    SYN3R_REQUIRE(ln_gamma && ln_beta, "feedforward_fused_ln_f16: null LayerNorm parameters");
    SYN3R_REQUIRE(((uintptr_t)ln_gamma | (uintptr_t)ln_beta) % 16 == 0, "feedforward_fused_ln_f16: LayerNorm parameters must be 16-byte aligned");
    SYN3R_REQUIRE(ln_eps > 0.f, "feedforward_fused_ln_f16: eps must be positive");
    return feedforward_fused(x, ldx, ln_gamma, ln_beta, ln_eps, w1_chunked, b1_chunked, D, w2, b2, out, ldc, residual, ldr, aux, ldaux,
                             s_acc, s_res, s_aux, M, C, stream);
}
extern "C" int syn3r_feedforward_fused_addln_f16(const void* x, long long ldx, const void* addvec, int rows_per_vec, const void* ln_gamma,
                                                 const void* ln_beta, float ln_eps, const void* w1_chunked, const void* b1_chunked, int D,
                                                 const void* w2, const void* b2, void* out, long long ldc, const void* aux, long long ldaux,
                                                 float s_acc, float s_res, float s_aux, int M, int C, void* stream) {
    SYN3R_REQUIRE(ln_gamma && ln_beta && addvec, "feedforward_fused_addln_f16: null LayerNorm parameters / add vector");
    SYN3R_REQUIRE(((uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)addvec) % 16 == 0, "feedforward_fused_addln_f16: LayerNorm parameters and the add vector must be 16-byte aligned");
    SYN3R_REQUIRE(ln_eps > 0.f, "feedforward_fused_addln_f16: eps must be positive");
    SYN3R_REQUIRE(rows_per_vec > 0, "feedforward_fused_addln_f16: rows_per_vec=%d must be positive", rows_per_vec);
    // the residual of this entry IS x + addvec (the tensor the reference keeps as `residual` before norm_in)
    return feedforward_fused(x, ldx, ln_gamma, ln_beta, ln_eps, w1_chunked, b1_chunked, D, w2, b2, out, ldc, x, ldx, aux, ldaux,
                             s_acc, s_res, s_aux, M, C, stream, addvec, rows_per_vec);
}
namespace {
int feedforward_fused(const void* x, long long ldx, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* w1_chunked,
                      const void* b1_chunked, int D, const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                      long long ldr, const void* aux, long long ldaux, float s_acc, float s_res, float s_aux, int M, int C, void* stream,
                      const void* addvec, int rows_per_vec) {
    SYN3R_REQUIRE(x && w1_chunked && b1_chunked && w2 && out, "feedforward_fused_f16: null operand");
    SYN3R_REQUIRE(C == F_C, "feedforward_fused_f16: the fused kernel is built for C = %d channels (got %d): use syn3r_feedforward_f16", F_C, C);
    SYN3R_REQUIRE(M > 0 && D >= F_HC && D % F_HC == 0, "feedforward_fused_f16: bad sizes M=%d D=%d (D must be a multiple of %d)", M, D, F_HC);
    FfnParams q{};
    GemmParams& p = q.e;
    p.A = (const __half*)x; p.lda = ldx; p.W = (const __half*)w2; p.out = (__half*)out; p.ldc = ldc;
    p.bias = (const __half*)b2; p.residual = (const __half*)residual; p.ldr = ldr; p.aux = (const __half*)aux; p.ldaux = ldaux;
    p.s_acc = s_acc; p.s_res = s_res; p.s_aux = s_aux; p.M = M; p.N = F_C; p.K = D;
    q.w1 = (const __half*)w1_chunked; q.b1 = (const __half*)b1_chunked; q.D = D;
    q.ln_g = (const __half*)ln_gamma; q.ln_b = (const __half*)ln_beta; q.ln_eps = ln_eps;
    q.ln_add = (const __half*)addvec; q.ln_add_rpv = rows_per_vec; p.res_add = (const __half*)addvec; p.res_add_rpv = rows_per_vec;
    int rc = check_common(p, "feedforward_fused_f16");
    if (rc) return rc;
    SYN3R_REQUIRE(ldx % 8 == 0 && ldx >= C, "feedforward_fused_f16: ldx=%lld must be >= C and a multiple of 8", ldx);
    SYN3R_REQUIRE(((uintptr_t)w1_chunked | (uintptr_t)b1_chunked) % 16 == 0, "feedforward_fused_f16: weights must be 16-byte aligned");
    return launch_ffn320(q, (hipStream_t)stream);
}
}  // namespace

extern "C" int syn3r_layernorm_linear320_f16(const void* x, long long ldx, const void* ln_gamma, const void* ln_beta, float ln_eps,
                                             const void* W, void* out, long long ldc, int M, int N, int C, void* stream) {
    SYN3R_REQUIRE(x && ln_gamma && ln_beta && W && out, "layernorm_linear320_f16: null operand");
    SYN3R_REQUIRE(C == F_C, "layernorm_linear320_f16: the kernel is built for C = %d channels (got %d): use syn3r_layernorm_f16 + syn3r_gemm_f16", F_C, C);
    SYN3R_REQUIRE(SYN3R_DIM_OK(M) && SYN3R_DIM_OK(N) && N % F_C == 0, "layernorm_linear320_f16: bad sizes M=%d N=%d (N must be a multiple of %d)", M, N, F_C);
    SYN3R_REQUIRE(ldx % 8 == 0 && ldx >= C && ldc % 8 == 0 && ldc >= N, "layernorm_linear320_f16: bad strides ldx=%lld ldc=%lld", ldx, ldc);
    SYN3R_REQUIRE(((uintptr_t)x | (uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)W | (uintptr_t)out) % 16 == 0, "layernorm_linear320_f16: operands must be 16-byte aligned");
    SYN3R_REQUIRE(ln_eps > 0.f, "layernorm_linear320_f16: eps must be positive");
    SYN3R_REQUIRE((long long)M * ldx < (1ll << 40) && (long long)M * ldc < (1ll << 40), "layernorm_linear320_f16: operand too large");
    LnLinParams p{};
    p.x = (const __half*)x; p.ldx = ldx; p.W = (const __half*)W; p.out = (__half*)out; p.ldc = ldc; p.M = M; p.N = N; p.C = C;
    p.ln_g = (const __half*)ln_gamma; p.ln_b = (const __half*)ln_beta; p.ln_eps = ln_eps;
    return launch_lnlin320(p, (hipStream_t)stream);
}

extern "C" int syn3r_conv2d3x3_f16(const void* X, const void* W, void* out, long long ldc, const void* bias,
                                   const void* rowvec, long long ldrv, int rows_per_vec, const void* residual,
                                   long long ldr, float s_acc, float s_res, int NB, int Hi, int Wi, int Cin, int Cout,
                                   int stride, int upsample, int pad_lo, void* stream) {
    SYN3R_REQUIRE(NB > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv2d3x3: bad sizes");
    SYN3R_REQUIRE(pad_lo == 0 || pad_lo == 1, "conv2d3x3: pad_lo must be 0 or 1");
    SYN3R_REQUIRE(stride == 1 || stride == 2, "conv2d3x3: stride must be 1 or 2");
    SYN3R_REQUIRE(!(upsample && stride != 1), "conv2d3x3: upsample requires stride 1");
    SYN3R_REQUIRE(Cin % BK == 0, "conv2d3x3: Cin=%d must be a multiple of %d (pad the input channels)", Cin, BK);
    GemmParams p{};
    p.A = (const __half*)X; p.W = (const __half*)W; p.out = (__half*)out; p.ldc = ldc; p.bias = (const __half*)bias;
    p.rowvec = (const __half*)rowvec; p.ldrv = ldrv; p.rows_per_vec = rows_per_vec;
    p.residual = (const __half*)residual; p.ldr = ldr; p.s_acc = s_acc; p.s_res = s_res; p.s_aux = 0.f;
    p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.stride = stride; p.ups = upsample ? 1 : 0; p.pad = pad_lo;
    int Hg = upsample ? 2 * Hi : Hi, Wg = upsample ? 2 * Wi : Wi;
    p.Ho = (Hg + pad_lo + 1 - 3) / stride + 1;      // one zero row/col always follows the last pixel
    p.Wo = (Wg + pad_lo + 1 - 3) / stride + 1;
    SYN3R_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d3x3: input too small");
    long long M = (long long)NB * p.Ho * p.Wo;
    SYN3R_REQUIRE(M < (1ll << 31), "conv2d3x3: too many output pixels");
    p.M = (int)M; p.N = Cout; p.K = 9 * Cin;
    int rc = check_common(p, "conv2d3x3");
    if (rc) return rc;
    return launch<MODE_CONV2D>(p, (hipStream_t)stream);
}

extern "C" int syn3r_conv2d3x3_act_f16(const void* X, const void* W, void* out, const void* bias, int relu, const void* relu_mask,
                                       int NB, int Hi, int Wi, int Cin, int Cout, void* stream) {
    SYN3R_REQUIRE(NB > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv2d3x3_act: bad sizes");
    SYN3R_REQUIRE(Cin % BK == 0 && Cout % 8 == 0, "conv2d3x3_act: Cin=%d must be a multiple of %d, Cout=%d of 8", Cin, BK, Cout);
    GemmParams p{};
    p.A = (const __half*)X; p.W = (const __half*)W; p.out = (__half*)out; p.ldc = Cout; p.bias = (const __half*)bias;
    p.s_acc = 1.0f; p.s_res = 0.f; p.s_aux = 0.f;
    p.relu = relu ? 1 : 0; p.relu_mask = (const __half*)relu_mask;
    p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.stride = 1; p.ups = 0; p.pad = 1; p.Ho = Hi; p.Wo = Wi;
    long long M = (long long)NB * Hi * Wi;
    SYN3R_REQUIRE(M < (1ll << 31), "conv2d3x3_act: too many output pixels");
    p.M = (int)M; p.N = Cout; p.K = 9 * Cin;
    int rc = check_common(p, "conv2d3x3_act");
    if (rc) return rc;
    SYN3R_REQUIRE(((uintptr_t)relu_mask % 16) == 0, "conv2d3x3_act: mask must be 16-byte aligned");
    // the persistent kernels have their own (lean) epilogue: the convolution modes never use them (launch_dma_bm)
    return launch<MODE_CONV2D>(p, (hipStream_t)stream);
}

extern "C" int syn3r_tconv3_f16(const void* X, const void* W, void* out, long long ldc, const void* bias,
                                const void* rowvec, long long ldrv, int rows_per_vec, const void* residual,
                                long long ldr, float s_acc, float s_res, int B, int F, int HW, int Cin, int Cout,
                                void* stream) {
    SYN3R_REQUIRE(B > 0 && F > 0 && HW > 0 && Cin > 0 && Cout > 0, "tconv3: bad sizes");
    SYN3R_REQUIRE(Cin % BK == 0, "tconv3: Cin=%d must be a multiple of %d", Cin, BK);
    GemmParams p{};
    p.A = (const __half*)X; p.W = (const __half*)W; p.out = (__half*)out; p.ldc = ldc; p.bias = (const __half*)bias;
    p.rowvec = (const __half*)rowvec; p.ldrv = ldrv; p.rows_per_vec = rows_per_vec;
    p.residual = (const __half*)residual; p.ldr = ldr; p.s_acc = s_acc; p.s_res = s_res; p.s_aux = 0.f;
    p.F = F; p.HW = HW; p.Cin = Cin;
    long long M = (long long)B * F * HW;
    SYN3R_REQUIRE(M < (1ll << 31), "tconv3: too many rows");
    p.M = (int)M; p.N = Cout; p.K = 3 * Cin;
    int rc = check_common(p, "tconv3");
    if (rc) return rc;
    return launch<MODE_TCONV>(p, (hipStream_t)stream);
}

#ifdef SYN3R_TIMING
// developer hook (tools/wide_timing.py): the segment sums the last timed launch left behind
extern "C" __attribute__((visibility("default"))) int syn3r_debug_wide_timing(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_wide_timing), sizeof(unsigned long long) * 64);
}
#endif
