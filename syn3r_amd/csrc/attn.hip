// Fused attention (QK^T / softmax / .V) for the SVD spatio-temporal UNet, head dim 64, fp16 in,
// fp32 softmax and accumulation.
//
// Replaces F.scaled_dot_product_attention as called from AttnProcessor2_0
// (thirdparty/diffusers/src/diffusers/models/attention_processor.py:1222-1299; scale 1/8, no mask):
//   k_attn_spatial   self-attention over the S = h*w tokens of one frame (attention.py:329),
//                    S up to 9216, flash-style online softmax, never materialises S x S.
//   k_attn_temporal  self-attention over the F <= 32 frames of one pixel (attention.py:491-508),
//                    one wavefront per (batch, pixel, head); tokens are addressed with the frame
//                    stride, so the reference's permutes (attention.py:487-489,527-529) never happen.
//
// CDNA4 mapping (both): scores are computed transposed, S^T = K.Q^T, so a lane owns ONE query column and its keys sit in
// the accumulator registers; the probability tile is used directly from the accumulator as the B operand of
// O^T += V^T.P (the k order of an accumulator-sourced fragment is a permutation of the keys, applied to V^T's fragment
// as well).  V stays row-major [key][d] in LDS and its transposed A fragments come from gfx950's ds_read_b64_tr_b16 (each
// 16-lane group fetches a 4-key x 16-d block column-major).  k_attn_spatial uses v_mfma_f32_16x16x32_f16 (round 5, below),
// k_attn_temporal v_mfma_f32_32x32x16_f16 (one 32 x 32 score tile IS its whole problem: element j of lane half h is key
// 16s + 8(j>>2) + 4h + (j&3), two transposed reads - keys r0..r0+3 and r0+8..r0+11 - make the permuted 8-key fragment).
#include "common.h"
#include <type_traits>

using namespace syn3r;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// LDS bank rules (MI355X_MICROARCH.md, LDS): ds_read_b64_tr_b16 is served per 32-lane half on 64 banks
// (a 256-byte bank row = two 128-byte tile rows); ds_read_b128 per 16-lane group {0-3,12-15,20-27}, ...
//
// V tile [keys][64 d] row-major.  A 32-lane half of a transposed read fetches 4 consecutive keys x 64 bytes:
// keys r and r+1 sit in opposite halves of the bank row, keys r and r+2 in the same half, so the 64-byte
// window of a key is flipped by bit 1 of the key (chunk ^ 4): 4 x 64 B cover all 64 banks once.
// 8-byte pieces stay intact.
__device__ __forceinline__ int v_off(int key, int d) { return key * 64 + ((((d >> 3) ^ (((key >> 1) & 1) << 2)) << 3) | (d & 7)); }
// K tile [keys][64 d] row-major, read as ds_read_b128 with row = lane & 31 and chunk fixed per lane half:
// the rows of one 16-lane group are {0-3,12-15,20-27} or {4-11,16-19,28-31}; (key >> 1) & 7 is distinct over
// the 8 even and over the 8 odd rows of either set, so chunk ^ ((key >> 1) & 7) hits 16 different 16-byte slots.
__device__ __forceinline__ int k_off(int key, int chunk) { return key * 64 + ((chunk ^ ((key >> 1) & 7)) << 3); }

// A fragment of O^T += V^T.P for d = d0 + (lane&31): keys r0 + {0..3} and r0 + 8 + {0..3} (r0 includes 4*h)
__device__ __forceinline__ half8 v_frag_tr(const __half* vs, int r0, int d0, int lane) {
    const int i = lane & 15, q = i >> 2, pc = i & 3;
    const int c = d0 + ((lane >> 4) & 1) * 16 + 4 * pc;
    typedef __attribute__((address_space(3))) fp16x4 lds4;
    fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4*)(vs + v_off(r0 + q, c)));
    fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4*)(vs + v_off(r0 + 8 + q, c)));
    half8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (_Float16)a[e]; r[4 + e] = (_Float16)b[e]; }
    return r;
}

constexpr float kNegBig = -1.0e30f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLazy = 8.0f;       // log2 units: how far a score may exceed the softmax reference before it is moved

__device__ __forceinline__ half8 pack8(const float16v& p, int s) {
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (_Float16)p[8 * s + j];
    return r;
}

struct AttnParams {
    const __half* q; const __half* k; const __half* v;   // column offsets already applied; head hd at +64*hd
    long long ld;                                         // row stride (halfs) of q/k/v
    __half* o; long long ldo;
    int S;        // tokens per sequence
    int nseq;     // sequences (spatial: B*F ; temporal: B*HW handled through strides)
    int heads;
    // temporal addressing: token (b, f, pix) lives at row (b*F + f)*HW + pix
    int F, HW;
};

constexpr int BQ = 128, BKV = 64, ATHREADS = 256;

typedef float float4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// k_attn_spatial (round 5: v_mfma_f32_16x16x32_f16).  Under this kernel the chip is clock-throttled with the matrix pipe
// exposed (profiles/r04/pmc: MFMA busy 0.68 at 1.4-1.5 GHz), and MI355X_MICROARCH.md (DVFS give-back, item 7) measures
// 1.12-1.15x the FLOP/s for the 16x16x32 shape in exactly that regime at equal cycles per FLOP.  One 32-query x 64-key score
// tile costs a wavefront 36 MFMAs of 16 cycles (16 QK^T, 16 PV, 4 row sums), 32 v_exp_f32, 16 v_max3_f32, 16 v_cvt_pk and
// 8 + 16 LDS fragment reads.  Everything the algorithm does not need is kept off the vector port:
//   * Q is prescaled by log2(e)/8 (one fp32 multiply, one rounding to fp16);
//   * a score tile S^T[16 keys][16 queries] leaves the pipe with the lane owning query (lane & 15) and keys 4 (lane >> 4) + r,
//     r = 0..3: a wavefront's 32 queries x 64 keys are 4 key groups x 2 query groups = 8 accumulator quads (32 registers);
//   * the softmax reference m is subtracted BY THE MATRIX PIPE through the C operand: the first k-step of every score tile
//     takes the 4-register tuple {-m, -m, -m, -m} of the lane's query as C, so the accumulator leaves the pipe as
//     s*log2e - m, ready for v_exp_f32 (hipcc emits the three-address MFMA form: no tuple copy; round 4's 32 x 32 kernel
//     needed an extra k-step against a ones fragment for this, two 32-cycle MFMAs per tile);
//   * m is moved lazily, and the test needs no cross-lane traffic: "some score exceeds m by more than 2^8" is a ballot over
//     per-lane partial maxima; the exact per-query maximum (two v_permlane swaps per query group) is only formed on the
//     path that moves the reference;
//   * P as the B operand of O^T += V^T.P: k-slot 8 (lane >> 4) + j of the 32-key group G is key 32 G + 4 (lane >> 4) + j for
//     j < 4 and 32 G + 16 + 4 (lane >> 4) + (j - 4) otherwise - the two score quads (2G, 2G+1) of the lane as they stand; the
//     matching A fragment of V^T comes from two ds_read_b64_tr_b16 (keys 32 G + 4 (lane >> 4) + 0..3 and + 16);
//   * the row sums come from the matrix pipe too: a 17th "d" tile of V^T that is all ones (a constant register fragment)
//     gives every lane its query's COMPLETE sum over the 32 keys - one 16-cycle MFMA per (key group, query group), no
//     cross-lane step at the end, and the normaliser sums exactly the fp16 probabilities the O^T MFMAs use (the per-lane
//     v_mfma_f32_4x4x4_16b_f16 sums of round 4 cost twice the issue slots: +2.7 % on this kernel, same box);
//   * K and V tiles go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane; the bank swizzles are applied on the
//     per-lane SOURCE address, both are XOR involutions of the 16-byte chunk index): no staging registers, no ds_write, and
//     the copy is issued after the tile's last fragment read so that hipcc's conservative vmcnt(0) in front of LDS reads
//     never waits on a copy in flight.
// LDS images: K (k_off): ds_read_b128 rows lane & 15, 16-byte chunk 4 ks + (lane >> 4), conflict-free in all four lane
// groups; V (v16_off): the 32-byte granule of a row XORed with (key >> 1) & 3 - a 32-lane half of a transposed read covers
// 8 consecutive keys x 32 bytes = all 64 banks once.
// Measured (profiles/r05/attention_v3.txt): 3.33-3.34 -> 3.02-3.12 ms at the level-0 shape (28 x 5 heads x 9216).
#ifndef ATTN_WAVES
#define ATTN_WAVES 3
#endif
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int v16_off(int key, int d) { return key * 64 + ((((d >> 4) ^ ((key >> 1) & 3)) << 4) | (d & 15)); }

// A fragment of O^T += V^T.P for d = d0 + (lane & 15): keys r0 + {0..3} and r0 + 16 + {0..3}, r0 = 32 G + 4 (lane >> 4)
__device__ __forceinline__ half8 v16_frag_tr(const __half* vs, int r0, int d0, int lane) {
    const int i = lane & 15, q = i >> 2, pc = i & 3;
    const int c = d0 + 4 * pc;
    typedef __attribute__((address_space(3))) fp16x4 lds4;
    fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4*)(vs + v16_off(r0 + q, c)));
    fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4*)(vs + v16_off(r0 + 16 + q, c)));
    half8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (_Float16)a[e]; r[4 + e] = (_Float16)b[e]; }
    return r;
}

__global__ void __launch_bounds__(ATHREADS, ATTN_WAVES) k_attn_spatial(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) __half Ks[2][BKV * 64];
    __shared__ __attribute__((aligned(1024))) __half Vs[2][BKV * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA targets in SGPRs
    const int l16 = lane & 15, g4 = lane >> 4;
    const int qblocks = (p.S + BQ - 1) / BQ;
    // all query blocks of one (sequence, head) share its K/V (2.4 MB at S = 9216): keep them on one XCD's L2
    int bid = (int)xcd_chunk_remap(blockIdx.x, gridDim.x);
    const int qb = bid % qblocks; bid /= qblocks;
    const int hd = bid % p.heads;
    const int seq = bid / p.heads;
    const long long row0 = (long long)seq * p.S;
    const int q0 = qb * BQ + wv * 32;
    const __half* qp = p.q + hd * 64;
    const __half* kp = p.k + hd * 64 + row0 * p.ld;
    const __half* vp = p.v + hd * 64 + row0 * p.ld;

    // Q fragments (B operand: query qg 16 + (lane & 15), d = 32 ks + 8 (lane >> 4) + j), prescaled by log2(e) / 8
    half8 qf[2][2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
        int qi = q0 + qg * 16 + l16;
        if (qi > p.S - 1) qi = p.S - 1;
        const __half* src = qp + (row0 + qi) * p.ld;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 t = *(const half8*)(src + ks * 32 + 8 * g4);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (_Float16)((float)t[j] * (0.125f * kLog2e));
            qf[qg][ks] = t;
        }
    }

    // LDS-DMA staging: a wavefront copies 1 KiB (8 keys x 128 B) per instruction, lane -> LDS slot (key = lane / 8,
    // stored chunk = lane % 8); wavefront wv owns keys 16 wv .. 16 wv + 15 of K and of V (two pieces each).
    unsigned ok[2], ov[2];            // per-lane byte offsets of the two source chunks inside a tile
    const int s_key = wv * 16 + (lane >> 3), s_c = lane & 7;
    const unsigned ldb = (unsigned)p.ld * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = s_key + 8 * i;
        ok[i] = (unsigned)key * ldb + (unsigned)((s_c ^ ((key >> 1) & 7)) << 4);
        ov[i] = (unsigned)key * ldb + (unsigned)((s_c ^ (((key >> 1) & 3) << 1)) << 4);
    }
    auto dma_tile = [&](int kv0, int buf) {
        // the tile's base stays a SCALAR (opaque to loop strength reduction, which would otherwise carry six per-lane
        // 64-bit pointers through the loop): the copies take the scalar-base + 32-bit lane-offset form
        long long tile_off = (long long)kv0 * p.ld * 2;
        asm volatile("" : "+s"(tile_off));
        const char* kb = (const char*)kp + tile_off;
        const char* vb = (const char*)vp + tile_off;
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(ok[i]), "+v"(ov[i]));   // keep the lane offsets 32-bit registers
        if (kv0 + BKV <= p.S) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(kb + (size_t)ok[i]), (lds_void_t*)(&Ks[buf][(wv * 2 + i) * 512]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(vb + (size_t)ov[i]), (lds_void_t*)(&Vs[buf][(wv * 2 + i) * 512]), 16, 0, 0);
            }
        } else {
            // the sequence's last, partial tile: keys beyond it are masked out of the scores below, their K / V rows only
            // have to be finite, so they re-read the sequence's last row
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = s_key + 8 * i;
                const int back = kv0 + key > p.S - 1 ? kv0 + key - (p.S - 1) : 0;
                const unsigned d = (unsigned)back * ldb;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(kb + (size_t)(ok[i] - d)), (lds_void_t*)(&Ks[buf][(wv * 2 + i) * 512]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(vb + (size_t)(ov[i] - d)), (lds_void_t*)(&Vs[buf][(wv * 2 + i) * 512]), 16, 0, 0);
            }
        }
    };

    float4v ot[4][2];                 // O^T: d = 16 dt + 4 (lane >> 4) + r, query 16 qg + (lane & 15)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) ot[dt][qg] = (float4v){0.f, 0.f, 0.f, 0.f};
    float4v lsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4v negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // C operand of a score tile's first k-step: -m of the lane's query
    half8 ones8;                      // A operand of the row sums: a 17th "d" tile of V^T that is all ones
#pragma unroll
    for (int j = 0; j < 8; ++j) ones8[j] = (_Float16)1.f;

    const int ntiles = (p.S + BKV - 1) / BKV;
    // one KV tile; CUR (the LDS buffer) is a compile-time constant so every fragment address is base + immediate
    auto tile = [&](int t, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        // Fragment reads are issued in batches AHEAD of the MFMAs that consume them: the four K fragments of a k-step before
        // its eight MFMAs, and all eight V fragments before the softmax, whose VALU work then covers their latency.
        float4v st[4][2];
        half8 kf[4];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) kf[kg] = *(const half8*)(&Ks[cur][k_off(kg * 16 + l16, g4)]);
        __builtin_amdgcn_sched_barrier(0);
        // The QK^T MFMAs issue at raised priority: VALU arbitration between the waves of a SIMD is by priority, then
        // age, and at equal priority the other waves' softmax VALU starves this wave's matrix issue (+5 %).
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) st[kg][qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kg], qf[qg][0], negm[qg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) kf[kg] = *(const half8*)(&Ks[cur][k_off(kg * 16 + l16, 4 + g4)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) st[kg][qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kg], qf[qg][1], st[kg][qg], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        half8 vf[2][4];
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vf[G][dt] = v16_frag_tr(Vs[cur], G * 32 + 4 * g4, dt * 16, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntiles) dma_tile((t + 1) * BKV, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const int kv0 = t * BKV;
        if (kv0 + BKV > p.S) {   // wave-uniform: mask keys beyond the sequence
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + kg * 16 + 4 * g4 + r;
                    if (key >= p.S) { st[kg][0][r] = kNegBig; st[kg][1][r] = kNegBig; }
                }
        }
        // per-lane partial maxima of the two queries (relative to the reference the matrix pipe subtracted)
        float mq[2];
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) {
            // v_max3_f32 written out: fmaxf() on matrix-pipe results makes hipcc canonicalise every operand first (42 v_max
            // per tile instead of 16)
            float a = max3(st[0][qg][0], st[0][qg][1], st[0][qg][2]), b = max3(st[1][qg][0], st[1][qg][1], st[1][qg][2]);
            float c = max3(st[2][qg][0], st[2][qg][1], st[2][qg][2]), d = max3(st[3][qg][0], st[3][qg][1], st[3][qg][2]);
            a = max3(a, st[0][qg][3], st[1][qg][3]);
            c = max3(c, st[2][qg][3], st[3][qg][3]);
            mq[qg] = max3(max3(a, b, c), d, d);
        }
        // m does not have to be the exact running maximum: softmax is invariant to it, it only has to keep exp2(s - m) inside
        // fp16 (P <= 2^kLazy) and the sums inside fp32.  It is set by the first tile and moved (scores shifted, running sums
        // rescaled, the C tuple rewritten) only when some query of the wavefront exceeds it by more than kLazy = 8.
        const bool first = t == 0;
        if (first || __ballot(fmaxf(mq[0], mq[1]) > kLazy) != 0ull) {
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) {
                float mx = mq[qg];
                {
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(mx), __float_as_int(mx), false, false);
                    mx = fmaxf(__int_as_float(sw[0]), __int_as_float(sw[1]));
                    auto sx = __builtin_amdgcn_permlane16_swap(__float_as_int(mx), __float_as_int(mx), false, false);
                    mx = fmaxf(__int_as_float(sx[0]), __int_as_float(sx[1]));
                }
                const float delta = first ? mx : fmaxf(mx, 0.f);     // the reference moves up by delta
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ot[dt][qg][r] *= alpha;
#pragma unroll
                    for (int r = 0; r < 4; ++r) lsum[qg][r] *= alpha;     // (only row 0 is read at the end)
                }
#pragma unroll
                for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[kg][qg][r] -= delta;
#pragma unroll
                for (int r = 0; r < 4; ++r) negm[qg][r] -= delta;
            }
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kg][qg][r] = __builtin_amdgcn_exp2f(st[kg][qg][r]);
        // O^T += V^T . P, row sums += 1 . P
#pragma unroll
        for (int G = 0; G < 2; ++G)
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pf[j] = (_Float16)st[2 * G][qg][j]; pf[4 + j] = (_Float16)st[2 * G + 1][qg][j]; }
                lsum[qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf, lsum[qg], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) ot[dt][qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[G][dt], pf, ot[dt][qg], 0, 0, 0);
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile's K / V have landed
        __syncthreads();
    };
    dma_tile(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int t = 0; t < ntiles; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
        // a query's normaliser: every row of the ones tile is the complete sum over the keys
        const float inv = 1.0f / lsum[qg][0];
        const int qi = q0 + qg * 16 + l16;
        if (qi < p.S) {
            __half* dst = p.o + (row0 + qi) * p.ldo + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                half4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (_Float16)(ot[dt][qg][e] * inv);
                *(half4*)(dst + dt * 16 + 4 * g4) = o4;
            }
        }
    }
}

// One wavefront per (batch, pixel, head); 4 wavefronts per block.
__global__ void __launch_bounds__(ATHREADS, 2) k_attn_temporal(AttnParams p, long long nitems) {
    __shared__ __attribute__((aligned(16))) __half Vsm[4][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lq = lane & 31, h = lane >> 5;
    long long item = (long long)blockIdx.x * 4 + wv;
    const bool live = item < nitems;
    if (!live) item = nitems - 1;   // keep the wavefront in lock-step for the barrier
    const int hd = (int)(item % p.heads);
    long long bp = item / p.heads;
    const int pix = (int)(bp % p.HW);
    const int b = (int)(bp / p.HW);
    const int F = p.F;
    auto row_of = [&](int f) { return ((long long)b * F + f) * p.HW + pix; };

    const int fq = lq < F ? lq : F - 1;
    const __half* qsrc = p.q + row_of(fq) * p.ld + hd * 64;
    const __half* ksrc = p.k + row_of(fq) * p.ld + hd * 64;
    half8 qf[4], kf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        half8 t = *(const half8*)(qsrc + ks * 16 + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = t[j] * (_Float16)0.125f;
        qf[ks] = t;
        kf[ks] = *(const half8*)(ksrc + ks * 16 + 8 * h);
    }
    // V row-major [32 keys][64 d] into this wavefront's LDS slice: 256 chunks of 16 bytes, 4 per lane
    __half* vs = Vsm[wv];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int q = lane + 64 * i;
        int key = q >> 3, ch = q & 7;
        uint4 rv = make_uint4(0, 0, 0, 0);
        if (key < F) rv = *(const uint4*)(p.v + row_of(key) * p.ld + hd * 64 + ch * 8);
        *(uint4*)(vs + v_off(key, ch * 8)) = rv;
    }
    float16v st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], st, 0, 0, 0);
    float mx = kNegBig;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int key = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (key >= F) st[r] = kNegBig;
        mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float e = __expf(st[r] - mx);
        st[r] = e;
        ls += e;
    }
    ls += __shfl_xor(ls, 32, 64);
    __syncthreads();   // V^T visible
    float16v ot[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        half8 pf = pack8(st, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            half8 vf = v_frag_tr(vs, s * 16 + 4 * h, dt * 32, lane);
            ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[dt], 0, 0, 0);
        }
    }
    if (live && lq < F) {
        const float inv = 1.0f / ls;
        __half* dst = p.o + row_of(lq) * p.ldo + hd * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (_Float16)(ot[dt][4 * g + e] * inv);
                *(half4*)(dst + dt * 32 + 8 * g + 4 * h) = o4;
            }
    }
}

}  // namespace

extern "C" int syn3r_attention_f16(const void* q, const void* k, const void* v, long long ld, void* out, long long ldo,
                                   int nseq, int S, int heads, void* stream) {
    SYN3R_REQUIRE(q && k && v && out, "attention: null tensor");
    SYN3R_REQUIRE(SYN3R_DIM_OK(nseq) && SYN3R_DIM_OK(S) && heads > 0 && heads <= 4096, "attention: bad sizes nseq=%d S=%d heads=%d", nseq, S, heads);
    SYN3R_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ld >= 64 * heads && ldo >= 64 * heads, "attention: bad strides");
    SYN3R_REQUIRE(ld < (1ll << 24), "attention: row stride %lld too large (the K / V tile offsets 64 * ld * 2 are 32-bit)", ld);
    SYN3R_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "attention: misaligned tensor");
    AttnParams p{};
    p.q = (const __half*)q; p.k = (const __half*)k; p.v = (const __half*)v; p.ld = ld;
    p.o = (__half*)out; p.ldo = ldo; p.S = S; p.nseq = nseq; p.heads = heads;
    long long blocks = (long long)nseq * heads * ((S + BQ - 1) / BQ);
    SYN3R_REQUIRE(blocks < (1ll << 31), "attention: grid too large");
    SYN3R_LAUNCH(k_attn_spatial, dim3((unsigned)blocks), dim3(ATHREADS), 0, (hipStream_t)stream, p);
    SYN3R_LAUNCH_CHECK("attention launch");
    return SYN3R_OK;
}

extern "C" int syn3r_attention_temporal_f16(const void* q, const void* k, const void* v, long long ld, void* out,
                                            long long ldo, int B, int F, int HW, int heads, void* stream) {
    SYN3R_REQUIRE(q && k && v && out, "attention_temporal: null tensor");
    SYN3R_REQUIRE(SYN3R_DIM_OK(B) && SYN3R_DIM_OK(HW) && heads > 0 && heads <= 4096 && F >= 1 && F <= 32, "attention_temporal: bad sizes B=%d F=%d HW=%d heads=%d (F <= 32)",
                  B, F, HW, heads);
    SYN3R_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ld >= 64 * heads && ldo >= 64 * heads, "attention_temporal: bad strides");
    SYN3R_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "attention_temporal: misaligned tensor");
    AttnParams p{};
    p.q = (const __half*)q; p.k = (const __half*)k; p.v = (const __half*)v; p.ld = ld;
    p.o = (__half*)out; p.ldo = ldo; p.S = F; p.heads = heads; p.F = F; p.HW = HW;
    long long nitems = (long long)B * HW * heads;
    long long blocks = (nitems + 3) / 4;
    SYN3R_REQUIRE(blocks < (1ll << 31), "attention_temporal: grid too large");
    SYN3R_LAUNCH(k_attn_temporal, dim3((unsigned)blocks), dim3(ATHREADS), 0, (hipStream_t)stream, p, nitems);
    SYN3R_LAUNCH_CHECK("attention_temporal launch");
    return SYN3R_OK;
}
