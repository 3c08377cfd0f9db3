// The two fused level-0 kernels (C = 320) and their launchers: k_ffn320r (FeedForward in one kernel) and k_lnlin320 (LayerNorm + q / k / v).
// Included by gemm.hip inside its anonymous namespace (one translation unit; the kernels share GemmParams, the epilogues and the
// LDS-DMA typedefs of gemm_common.h / gemm_dma.h).

// ---------------------------------------------------------------------------------------------
// Fused feed-forward for C = 320 (the level-0 transformer blocks: FeedForward.forward, attention.py:608-665, with the
// GEGLU of activations.py):   out = epilogue( geglu(x . W1^T + b1) . W2^T )   in ONE kernel.
// The two-kernel path writes the gated hidden activation ([M, 1280] fp16 = 660 MB at M = 258 048) and reads it back;
// round 1 measured the first projection at half its matrix rate because of that output stream.  Here a
// block owns 128 rows and walks the hidden dimension in chunks of 64 -
//     phase 1   S[128, 128]  = x . W1_j^T            (K = 320, five 64-wide k-tiles of the chunk's 128 packed rows)
//     gate      h[128, 64]   = (S_h + b) * gelu(S_g + b)   in registers, fp16-rounded as the reference's projection output
//     phase 2   out[128,320] += h . W2[:, j]^T       (K = 64)
// and the [128, 320] fp32 result lives in registers for the whole kernel; every SIMD holds TWO wavefronts, so one's gate
// arithmetic, LDS-DMA issue and barrier waits overlap the other's MFMAs (a one-wavefront-per-SIMD build ran 2.1x slower).
// Nothing but x and out touches HBM; the weights (2.4 MB, L2-resident) stream through an LDS ring by LDS-DMA with counted
// vmcnt.  W1 rows are packed per 64-wide chunk as 4 x [16 hidden | 16 gate] (a lane holds a hidden value and its gate in
// matching accumulator tiles).  Rounds 1-3 kept the x tile in LDS (k_ffn320: x 80 KB | ring 3 x 20 KB | h | bias lines; removed in
// round 5); k_ffn320r below keeps it in registers.  The tile constants are shared with k_lnlin320, whose x tile IS LDS-resident.
constexpr int F_C = 320, F_HC = 64, F_BM = 128;
constexpr int F_X_BYTES = F_BM * F_C * 2;            // 81,920
constexpr int F_SLOT = 160 * BK * 2;                 // 20,480: a W2 half-chunk [160 x 64]; W1 k-tiles [128 x 64] use 16,384 of it

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
// place, by all 512 threads of the block; waits for the tile's DMA first (k_lnlin320; k_ffn320r normalises its register tile with the
// same expressions).
__device__ __forceinline__ void ln_tile320(char* smem_raw, int tid, int m0, int M, float cf, const __half* ln_g, const __half* ln_b,
                                           float ln_eps, const __half* add, int add_rpv) {
    // LayerNorm of the resident x tile (attention.py:430-453: norm3 in front of ff), so that the normalised activation is never
    // written to / re-read from HBM.  Same arithmetic, same order of additions as k_layernorm<8> (norm.hip): 8 partial sums per
    // row over the 16-byte chunks c, c + 8, ..., combined by the xor tree 4, 2, 1 - here four threads per row hold two of the
    // eight each: sub-lanes part and part + 4 (round 6; rounds 4-5 held 2 part and 2 part + 1, whose four threads per row read the
    // EVEN chunks of a k-tile in one ds_read_b128 - two of the four rows a 16-lane group covers then share their banks whatever the
    // row swizzle: the 2-way conflict of profiles/r05/pmc; chunks part / part + 4 put a group's rows on disjoint bank quarters).
    {
        const int r = tid >> 2, part = tid & 3;
        // Which of its two chunks a thread touches FIRST alternates with bit 1 of the row (f): the 16 lanes of a ds_read_b128 /
        // ds_write_b128 pass are four consecutive rows x four parts, a row's pass covers one 64-byte quarter of the 256-byte bank
        // space - quarter 2 (r & 1) + (e ^ (r >> 2 & 1)) with chunks part / part + 4 in a fixed order, i.e. rows r and r + 2 on the
        // SAME quarter (the 2-way conflict that was left: 0.25 of the kernel's LDS cycles, profiles/r06/pmc); with slot e reading chunk
        // part + 4 (e ^ f) the four rows take the four quarters.  The partial sums keep their identity (sa: chunks part, sb: chunks
        // part + 4, each summed over the k-tiles in order), so the bits are k_layernorm<8>'s as before.
        const int f = (r >> 1) & 1;
        half8 xv[5][2], addv[5][2];        // [k-tile][slot]: slot e holds chunk part + 4 (e ^ f)
        if (add) {       // requested before the wait for the x tile: one latency, not two
            int m = m0 + r;
            m = m < M ? m : M - 1;
            const __half* av = add + (long long)(m / add_rpv) * F_C;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int e = 0; e < 2; ++e) addv[kt][e] = *(const half8*)(av + (kt * 8 + part + 4 * (e ^ f)) * 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                xv[kt][e] = *(const half8*)(smem_raw + kt * 16384 + r * 128 + (((part + 4 * (e ^ f)) ^ (r & 7)) << 4));
        if (add) {       // norm_in of the temporal block normalises hidden + frame-position embedding (attention.py:500-507)
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int e = 0; e < 2; ++e) xv[kt][e] = xv[kt][e] + addv[kt][e];   // fp16 add, as k_layernorm
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int i = 0; i < 8; ++i) { s0 += (float)xv[kt][0][i]; s1 += (float)xv[kt][1][i]; }
        const float sa = f ? s1 : s0, sb = f ? s0 : s1;      // sa: the chunks part, sb: the chunks part + 4
        float s = sa + sb;                                   // xor 4: sub-lanes part and part + 4 live in this thread
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        // cf = 320 as a run-time value: the same division k_layernorm compiles to
        const float mean = s / cf;
        float q0 = 0.f, q1 = 0.f;
        {
#pragma clang fp contract(off)      // k_layernorm's squares are a packed multiply followed by adds, not an fma: the same bits here
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float da = (float)xv[kt][0][i] - mean, db = (float)xv[kt][1][i] - mean;
                    const float da2 = da * da, db2 = db * db;
                    q0 += da2; q1 += db2;
                }
        }
        const float qa = f ? q1 : q0, qb = f ? q0 : q1;
        float qq = qa + qb;
        qq += __shfl_xor(qq, 2, 64);
        qq += __shfl_xor(qq, 1, 64);
        const float rstd = rsqrtf(qq / cf + ln_eps);
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cv = kt * 8 + part + 4 * (e ^ f);
                const half8 g = *(const half8*)(ln_g + cv * 8), b = *(const half8*)(ln_b + cv * 8);
                half8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (_Float16)(((float)xv[kt][e][i] - mean) * rstd * (float)g[i] + (float)b[i]);
                *(half8*)(smem_raw + kt * 16384 + r * 128 + (((part + 4 * (e ^ f)) ^ (r & 7)) << 4)) = o;
            }
        __syncthreads();
    }

}

// ---------------------------------------------------------------------------------------------
// k_ffn320r: the fused feed-forward with the x tile in REGISTERS (round 4).  In its predecessor k_ffn320 (removed) the resident x
// tile was half of the LDS: the weight ring had three slots (two stages of look-ahead), every stage re-reads the x fragments from LDS (320 of the 624 KB
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
                // five requests at a time: written load-add-load-add, hipcc gave every load its own s_waitcnt vmcnt(0) - ten serialised
                // round trips per row tile (round 6; all ten at once spills)
#pragma unroll
                for (int k5 = 0; k5 < 10; k5 += 5) {
                    half8 addv[5];
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) addv[ks] = *(const half8*)(av + (k5 + ks) * 32);
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) xf[i][k5 + ks] = xf[i][k5 + ks] + addv[ks];   // fp16 add, as k_layernorm
                }
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
                // packed arithmetic throughout (round 6): bias add as v_pk_add_f32, the projection's rounding to fp16 as ONE
                // v_cvt_pk_f16_f32 per pair - the values and roundings of the scalar form, ~3.5 instructions fewer per pair
                typedef _Float16 half2e __attribute__((ext_vector_type(2)));
                typedef unsigned u2e __attribute__((ext_vector_type(2)));
                u2e o;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const syn3r_f2 bhf = __builtin_convertvector((half2e){bh[u][r], bh[u][r + 1]}, syn3r_f2);
                    const syn3r_f2 bgf = __builtin_convertvector((half2e){bg[u][r], bg[u][r + 1]}, syn3r_f2);
                    const half2e hh = __builtin_convertvector((syn3r_f2){S[i][2 * u][r], S[i][2 * u][r + 1]} + bhf, half2e);
                    const half2e gh = __builtin_convertvector((syn3r_f2){S[i][2 * u + 1][r], S[i][2 * u + 1][r + 1]} + bgf, half2e);
                    const syn3r_f2 y = __builtin_convertvector(hh, syn3r_f2) * gelu_pk(__builtin_convertvector(gh, syn3r_f2));
                    o[r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(y, half2e));
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
        // the ten bias requests first (round 6): written load-use per column tile, hipcc gave each its own s_waitcnt vmcnt(0) - ten
        // serialised round trips per block
        half4e bq[10];
#pragma unroll
        for (int jt = 0; jt < 10; ++jt) {
            bq[jt] = (half4e){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (p.bias) bq[jt] = *(const half4e*)(p.bias + gn0 + jt * 16 + fq * 4);
        }
#pragma unroll
        for (int jt = 0; jt < 10; ++jt) {
            float b4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = (float)bq[jt][r];
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
// path writes the normalised activation ([M, 320] fp16) and reads it back, and its contraction (K = 320: five k-tiles per tile)
// runs at a quarter of the matrix peak.  Here a block owns 128 rows: the x tile (80 KB) is DMA'd
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
    // Staging writes (round 6): a 16-lane group of a ds_write_b64 is 16 rows of ONE column position; at the 160-byte row pitch they
    // fall on four bank pairs - the 4-way conflict that was 12.3 M of this kernel's 38.3 M LDS cycles (profiles/r05/pmc).  Row r
    // keeps its ten 16-byte chunks ROTATED by r positions ((chunk + r) mod 10): eight bank groups per 16 rows, 2-way (an 8-byte
    // store cannot do better at a 16-byte-aligned pitch); the read-back takes whole rows as before and un-rotates the column.
    unsigned st_wr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st_wr[j] = st_base + (unsigned)(fr * (WN * 2) + ((j * 2 + (fq >> 1) + fr) % 10) * 16 + (fq & 1) * 8);
    int st_col[3];                       // column (halfs) of the chunk this lane reads back in round it: position q = lane + 64 it
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int q = lane + it * 64, row = q / (WN / 8), chp = q - row * (WN / 8);
        st_col[it] = ((chp + 20 - row) % 10) * 8;
    }
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
                DS_WRITE64(st_wr[j], o);
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
                const int row = q / (WN / 8);
                const int m = m0 + wm * WM + i * 16 + row;
                if (q < 16 * (WN / 8) && m < p.M) *(half8*)(orow + (long long)m * p.ldc + st_col[it]) = v[it];
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
