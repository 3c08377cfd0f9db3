// k_gemm_widep: the persistent 256 x 320 tile, and the lean epilogue the persistent kernels share.
// Included by gemm.hip inside its anonymous namespace (one translation unit; the kernels share GemmParams, the epilogues and the
// LDS-DMA typedefs of gemm_common.h / gemm_dma.h).

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
// GroupNorm partial sums of a wavefront's staged output tile (GemmParams::gn_part): NI x 16 rows x 80 columns of FINAL fp16 values
// in `st` (row stride EPI_LD).  A lane takes one row (NI = 4: row = lane; NI = 2: row = lane & 31 and the column half lane >> 5) and
// reads it as 16-byte chunks (row stride 176 B: the 16 lanes of a read pass land on 16 different 16-byte bank groups), sums pairs of
// neighbouring columns with v_dot2c_f32_f16 (x . (1, 1) and x . x: fp16 products are exact in fp32) into the 10-column unit the pair
// belongs to (compile-time: pair p of the 40 pairs of a row is unit p / 5), then the 32 lanes of a half are folded: one
// v_permlane16_swap + add per unit leaves the sums of x on the even 16-lane rows and the sums of x^2 on the odd ones, four DPP
// row rotations finish each.  Fixed order throughout: bitwise reproducible.  ~140 (NI = 4) / ~70 vector instructions per call.
template <int NI>
__device__ __forceinline__ void gn_tile_stats(const GemmParams& p, const __half* st, int lane, int gm0, int gn0, int N) {
    typedef _Float16 half2e __attribute__((ext_vector_type(2)));
    constexpr int NU = NI == 4 ? 8 : 4, NC = NI == 4 ? 10 : 5;
    const int row = NI == 4 ? lane : (lane & 31);
    const int half_ = NI == 4 ? 0 : (lane >> 5);
    float s[NU], q[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { s[u] = 0.f; q[u] = 0.f; }
    const half2e one = {(_Float16)1.0f, (_Float16)1.0f};
    const __half* src = st + row * EPI_LD + half_ * 40;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const half8 v = *(const half8*)(src + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half2e pr = {v[2 * k], v[2 * k + 1]};
            s[(4 * c + k) / 5] = __builtin_amdgcn_fdot2(pr, one, s[(4 * c + k) / 5], false);
            q[(4 * c + k) / 5] = __builtin_amdgcn_fdot2(pr, pr, q[(4 * c + k) / 5], false);
        }
    }
    float w[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {        // odd 16-lane rows of s <-> even rows of q: even rows own the sums, odd rows the sums of squares
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(s[u]), __float_as_int(q[u]), false, false);
        w[u] = __int_as_float(r[0]) + __int_as_float(r[1]);
        w[u] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[u]), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
        w[u] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[u]), 0x124 /* row_ror:4 */, 0xF, 0xF, false));
        w[u] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[u]), 0x122 /* row_ror:2 */, 0xF, 0xF, false));
        w[u] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[u]), 0x121 /* row_ror:1 */, 0xF, 0xF, false));
    }
    if ((lane & 15) == 0) {
        const int rowp = lane >> 4;                                   // 0 / 2: sums, 1 / 3: sums of squares; 2, 3: the upper 32 lanes
        const int m = gm0 + (NI == 4 ? 32 * (rowp >> 1) : 0);
        const int n = gn0 + (NI == 4 ? 0 : 40 * (rowp >> 1));
        if (m < p.M && n < N) {
            float* dst = p.gn_part + ((size_t)(m >> 5) * 2 + (rowp & 1)) * (size_t)p.gn_units + n / 10;
#pragma unroll
            for (int u = 0; u < NU; u += 4) *(float4v*)(dst + u) = (float4v){w[u], w[u + 1], w[u + 2], w[u + 3]};
        }
    }
}

template <int NI, bool HOIST = false>
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
    if constexpr (HOIST) {
    // Every bias / row-vector value of the call is REQUESTED before the first is used (round 6, HOIST; k_gemm_dmap): written
        // load-use-load-use, the compiler gave each of the up to 25 loads its own s_waitcnt vmcnt(0) - 5 to 25 serialised L2 round trips inside the epilogue.
        half4e bq[TN], rq[NI][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = gn0 + j * 16 + fq * 4;
            bq[j] = (half4e){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (bias && n < N) bq[j] = *(const half4e*)(bias + n);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                rq[i][j] = (half4e){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                if (p.rowvec && n < N) rq[i][j] = *(const half4e*)(rv[i] + n);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float b4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = (float)bq[j][r];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float a4[4] = {b4[0], b4[1], b4[2], b4[3]};
                if (p.rowvec) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a4[r] += (float)rq[i][j][r];
                }
                half4e o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)((acc[i][j][r] + a4[r]) * p.s_acc);
                *(half4e*)(st + (i * 16 + fr) * EPI_LD + j * 16 + fq * 4) = o;
            }
        }
    } else {
        // (the kernels whose accumulators leave no registers for that - k_gemm_widep, k_gemm_z: 160 accumulators - load where they use)
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
                if (p.gn_part) *(half8*)(st + row * EPI_LD + ch * 8) = v;      // the statistics pass reads the values as stored
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
            if (p.gn_part) *(half8*)(st + row * EPI_LD + ch * 8) = v;          // the statistics pass reads the values as stored
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
    if (p.gn_part) {                      // (wave-uniform)
        gn_tile_stats<NI>(p, st, lane, gm0, gn0, N);
        __builtin_amdgcn_wave_barrier();
    }
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
