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
//   k_gemm_dmapd   persistent 256 x 160 tile with a deferred, LDS-free epilogue (gemm_dmapd.h): residual projections, 640 <= K <= 1280
//   k_gemm_skinny  M <= 16 rows (time embedding, folded cross-attention context)
// This file is the dispatch: the shape rules (launch_dma), the launchers and the C-ABI entry points; the kernel families live in
// gemm_common.h / gemm_dma.h / gemm_wide.h / gemm_z.h / gemm_dmap.h / gemm_ffn.h, included below into ONE translation unit.
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <type_traits>

using namespace syn3r;

namespace {

#include "gemm_common.h"
#include "gemm_dma.h"
#include "gemm_wide.h"
#include "gemm_z.h"
#include "gemm_dmap.h"
#include "gemm_dmapd.h"
#include "gemm_g256.h"
#include "gemm_ffn.h"

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

// GroupNorm partial sums (GemmParams::gn_part) of the CALLING THREAD's next contraction (syn3r_gemm_set_gn_partials): the entry
// points move the pending buffer into the launch's parameters (gn_take), the launchers of the kernels with the lean epilogue
// report that they wrote it (syn3r_gemm_gn_partials_written); every other kernel leaves it untouched and the caller runs the
// statistics pass.
thread_local void* g_gn_pending = nullptr;
thread_local size_t g_gn_pending_bytes = 0;
thread_local bool g_gn_written = false;
size_t gn_partials_bytes(long long M, long long N) { return (size_t)(M / 32) * 2 * (size_t)(N / 10) * sizeof(float); }
// entry points whose kernels never write partial sums: a request left pending by the caller must not reach a later, unrelated launch
void gn_drop() { g_gn_pending = nullptr; g_gn_pending_bytes = 0; g_gn_written = false; }
void gn_take(GemmParams& p) {
    g_gn_written = false;
    void* buf = g_gn_pending;
    const size_t bytes = g_gn_pending_bytes;
    g_gn_pending = nullptr; g_gn_pending_bytes = 0;
    if (!buf || p.M % 32 != 0 || p.N % 80 != 0 || p.geglu_D > 0 || p.out_tiled || bytes < gn_partials_bytes(p.M, p.N)) return;
    p.gn_part = (float*)buf; p.gn_units = p.N / 10;
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
    g_gn_written = p.gn_part != nullptr;
    return SYN3R_OK;
}

// k_gemm_g256 (gemm_g256.h): the gated projection on whole 256 x 256 tiles into the A-tiled hidden activation
bool g256_admits(const GemmParams& p) {
    return p.geglu_D > 0 && p.bias != nullptr && p.out_tiled && !p.a_tiled && !p.A2 && !p.rowvec && !p.residual && !p.aux && p.s_acc == 1.0f && p.M % 128 == 0 && p.M >= 256 &&
           p.geglu_D % 128 == 0 && p.N == 2 * p.geglu_D && p.K >= 2 * BK && p.lda % 8 == 0 && (long long)p.M * p.lda < (1ll << 31) &&
           (long long)p.N * p.K < (1ll << 31);
}
int launch_g256(const GemmParams& p, hipStream_t stream) {
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_gemm_g256, G_LDS, "hipFuncSetAttribute(gemm_g256)")) return rc;
    const int tiles = ((p.M + 255) / 256) * (p.N / 256);
    const int blocks = std::min(tiles, persistent_blocks());
    GemmParams q = p;
    static const int band_env = tune_env("SYN3R_G256_BAND", 0);
    q.band = band_env > 0 ? band_env : 4;
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_g256[M%d,N%d,K%d,e2]", p.M, p.N, p.K);
        else snprintf(name, sizeof(name), "k_gemm_g256");
    }
    SYN3R_LAUNCH_NAMED(name, k_gemm_g256, dim3(blocks), dim3(512), G_LDS, stream, q);
    SYN3R_LAUNCH_CHECK("gemm_g256 launch");
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
    if constexpr (MODE == MODE_CONV2D) q.a_bytes = (unsigned)((long long)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Cin * 2);      // (launch_dma: < 2^32 - 2^20)
    if constexpr (MODE == MODE_TCONV) q.a_bytes = (unsigned)((long long)p.M * p.Cin * 2);
    const GemmParams& p_ = q;
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_z<%d>[M%d,N%d,K%d,e%d]", MODE, p.M, p.N, p.K, p.geglu_D > 0 ? 2 : (p.residual != nullptr));
        else snprintf(name, sizeof(name), "k_gemm_z<%d>", MODE);
    }
    if constexpr (MODE == MODE_DENSE) {
        if (p.A2) { SYN3R_LAUNCH_NAMED(name, (k_gemm_z<true, MODE_DENSE>), dim3(blocks), dim3(512), Z_LDS, stream, p_); SYN3R_LAUNCH_CHECK("gemm_z launch"); g_gn_written = p.gn_part != nullptr; return SYN3R_OK; }
    }
    if constexpr (MODE == MODE_CONV2D) {
        if (p.ups) { SYN3R_LAUNCH_NAMED(name, (k_gemm_z<false, MODE_CONV2D, true>), dim3(blocks), dim3(512), Z_LDS, stream, p_); SYN3R_LAUNCH_CHECK("gemm_z launch"); g_gn_written = p.gn_part != nullptr; return SYN3R_OK; }
    }
    SYN3R_LAUNCH_NAMED(name, (k_gemm_z<false, MODE>), dim3(blocks), dim3(512), Z_LDS, stream, p_);
    SYN3R_LAUNCH_CHECK("gemm_z launch");
    g_gn_written = p.gn_part != nullptr;
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

int launch_dmapd(const GemmParams& p, hipStream_t stream) {
    constexpr size_t lds = (size_t)3 * (256 * BK * 2 + DMA_B_BYTES);   // 159,744 B
    static DevOnce once;
    if (int rc = set_max_lds(once, (const void*)k_gemm_dmapd, (int)lds, "hipFuncSetAttribute(gemm_dmapd)")) return rc;
    const int tiles = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
    const int blocks = std::min(tiles, persistent_blocks());
    char name[96];
    if (trace_on()) {
        if (trace_detail()) snprintf(name, sizeof(name), "k_gemm_dmapd[M%d,N%d,K%d,e%d]", p.M, p.N, p.K, p.residual != nullptr);
        else snprintf(name, sizeof(name), "k_gemm_dmapd");
    }
    SYN3R_LAUNCH_NAMED(name, k_gemm_dmapd, dim3(blocks), dim3(512), lds, stream, p);
    SYN3R_LAUNCH_CHECK("gemm_dmapd launch");
    return SYN3R_OK;
}

// Does the deferred-epilogue kernel take this contraction?  Dense, row-major output, WHOLE 256 x 160 tiles (no row / column clamps in
// the kernel: what the UNet launches at full frames), at least four k-tiles (a k-tile carries at most one of the parked tile's four
// units), a residual only with 16-byte aligned rows, no aux blend.
bool dmapd_admits(const GemmParams& p) {
    return p.geglu_D <= 0 && !p.A2 && !p.a_tiled && !p.out_tiled && !p.out_nt && !p.relu && !p.relu_mask && p.ksplit <= 1 && !p.aux &&
           !p.res_add && p.M % 256 == 0 && p.N % BN == 0 && p.K / BK >= 4 && p.ldc % 8 == 0 && (!p.residual || p.ldr % 8 == 0) &&
           (!p.rowvec || p.ldrv % 4 == 0);
}

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
    g_gn_written = p.gn_part != nullptr;
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
    if constexpr (MODE == MODE_DENSE) {
        // Round 6: the residual projections with 640 <= K <= 1280 (attn1 / attn2 to_out, proj_out at levels 1-2: [64512,640,640],
        // [16128,1280,1280]) spend as long in their epilogue as in their k-loop on the 256 x 320 tile; k_gemm_dmapd parks the result
        // and runs the residual loads / adds / stores inside the next tile's k-loop: -8 % on exactly these shapes inside the unit, same
        // box (profiles/r05/dmapd_deferred_epilogue.txt; K = 320 +-0 and the projections WITHOUT a residual 0..+7 % slower: they stay;
        // [16128,1280,1280] without a residual -11 %: taken).  SYN3R_DMAPD_WIDE=0: never; 2: every admissible K <= 1280 shape.
        static const int ddw_env = tune_env("SYN3R_DMAPD_WIDE", 1);
        const bool won = p.residual ? (p.K >= 640 && p.K <= 1280 && p.N <= 1280) : (p.K == 1280 && p.N == 1280);
        // (a launch asked for GroupNorm partial sums - proj_out, 5 of these 15 launches per level - keeps the lean epilogue, which
        // writes them: the statistics pass it saves costs more than the deferred epilogue gains)
        if (ddw_env != 0 && g_dma_bm == 0 && !p.gn_part && (ddw_env == 2 ? p.K <= 1280 : won) && dmapd_admits(p) &&
            (long long)((p.M + 255) / 256) * ((p.N + BN - 1) / BN) >= 256)
            return launch_dmapd(p, stream);
    }
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

extern "C" size_t syn3r_gn_partials_bytes(int M, int N) {
    if (!SYN3R_DIM_OK(M) || !SYN3R_DIM_OK(N) || M % 32 != 0 || N % 80 != 0) return 0;
    return gn_partials_bytes(M, N);
}

extern "C" int syn3r_gemm_set_gn_partials(void* partials, size_t bytes) {
    SYN3R_REQUIRE((partials == nullptr) == (bytes == 0), "gemm_set_gn_partials: pointer and size must both be given or both be zero");
    SYN3R_REQUIRE(((uintptr_t)partials % 16) == 0, "gemm_set_gn_partials: the buffer must be 16-byte aligned");
    g_gn_pending = partials;
    g_gn_pending_bytes = bytes;
    return SYN3R_OK;
}

extern "C" int syn3r_gemm_gn_partials_written(void) { return g_gn_written ? 1 : 0; }

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
    gn_take(p);
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
    gn_take(p);
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
    gn_drop();
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
    gn_drop();
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

extern "C" int syn3r_feedforward_p64_supported(int M, int D, int C_in) {
    return (SYN3R_DIM_OK(M) && SYN3R_DIM_OK(D) && SYN3R_DIM_OK(C_in) && M % 128 == 0 && M >= 256 && D % 128 == 0 && C_in % BK == 0 && C_in >= 2 * BK &&
            (long long)M * C_in < (1ll << 31) && 2ll * D * C_in < (1ll << 31)) ? 1 : 0;
}

extern "C" int syn3r_feedforward_p64_f16(const void* x, long long ldx, const void* w1_packed64, const void* b1_packed64, int D,
                                         const void* w2, const void* b2, void* out, long long ldc, const void* residual,
                                         long long ldr, const void* aux, long long ldaux, float s_acc, float s_res,
                                         float s_aux, int M, int C_in, int C_out, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    gn_drop();
    SYN3R_REQUIRE(x && w1_packed64 && b1_packed64 && w2 && out, "feedforward_p64_f16: null operand");
    SYN3R_REQUIRE(syn3r_feedforward_p64_supported(M, D, C_in) != 0 && C_out > 0 && ldx == C_in,
                  "feedforward_p64_f16: shape M=%d D=%d C_in=%d not served (syn3r_feedforward_p64_supported; x must be dense rows)", M, D, C_in);
    const size_t need = syn3r_feedforward_workspace_bytes(M, D);
    if (!workspace || workspace_bytes < need) {
        set_error("feedforward_p64_f16: workspace %zu < %zu", workspace_bytes, need);
        return SYN3R_E_WORKSPACE;
    }
    SYN3R_REQUIRE((uintptr_t)workspace % 16 == 0, "feedforward_p64_f16: workspace must be 16-byte aligned");
    GemmParams p{};
    p.A = (const __half*)x; p.lda = ldx; p.W = (const __half*)w1_packed64; p.out = (__half*)workspace; p.ldc = 2ll * D;
    p.bias = (const __half*)b1_packed64; p.s_acc = 1.0f; p.M = M; p.N = 2 * D; p.K = C_in; p.geglu_D = D; p.out_tiled = 1;
    p.out_nt = need >= ((size_t)256 << 20);
    int rc = check_common(p, "feedforward_p64_f16(net.0)");
    if (rc) return rc;
    SYN3R_REQUIRE(g256_admits(p), "feedforward_p64_f16: shape not admitted by the 256 x 256 kernel");
    rc = launch_g256(p, (hipStream_t)stream);
    if (rc) return rc;
    GemmParams q{};
    q.A = (const __half*)workspace; q.lda = D; q.a_tiled = 1; q.W = (const __half*)w2; q.out = (__half*)out; q.ldc = ldc;
    q.bias = (const __half*)b2; q.residual = (const __half*)residual; q.ldr = ldr; q.aux = (const __half*)aux; q.ldaux = ldaux;
    q.s_acc = s_acc; q.s_res = s_res; q.s_aux = s_aux; q.M = M; q.N = C_out; q.K = D;
    rc = check_common(q, "feedforward_p64_f16(net.2)");
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
    gn_drop();
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
    gn_drop();
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
    gn_take(p);
    int rc = check_common(p, "conv2d3x3");
    if (rc) return rc;
    return launch<MODE_CONV2D>(p, (hipStream_t)stream);
}

extern "C" int syn3r_conv2d3x3_act_f16(const void* X, const void* W, void* out, const void* bias, int relu, const void* relu_mask,
                                       int NB, int Hi, int Wi, int Cin, int Cout, void* stream) {
    gn_drop();
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
    gn_take(p);
    int rc = check_common(p, "tconv3");
    if (rc) return rc;
    return launch<MODE_TCONV>(p, (hipStream_t)stream);
}

#ifdef SYN3R_TIMING
// developer hook (tools/wide_timing.py): the segment sums the last timed launch left behind
extern "C" __attribute__((visibility("default"))) int syn3r_debug_wide_timing(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_wide_timing), sizeof(unsigned long long) * 64);
}
extern "C" __attribute__((visibility("default"))) int syn3r_debug_g256_timing(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_g256_timing), sizeof(unsigned long long) * 64);
}
#endif
