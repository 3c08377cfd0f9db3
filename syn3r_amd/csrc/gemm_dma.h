// k_gemm_dma (BM x 160 tile on an LDS-DMA ring, one tile per block) and k_splitk_finish.
// Included by gemm.hip inside its anonymous namespace (one translation unit; the kernels share GemmParams, the epilogues and the
// LDS-DMA typedefs of gemm_common.h / gemm_dma.h).

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
