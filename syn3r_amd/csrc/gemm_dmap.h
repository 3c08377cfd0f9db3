// k_gemm_dmap: the persistent form of the 256 x 160 LDS-DMA kernel.
// Included by gemm.hip inside its anonymous namespace (one translation unit; the kernels share GemmParams, the epilogues and the
// LDS-DMA typedefs of gemm_common.h / gemm_dma.h).

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
            lean_store<2, true>(p, acc, st, le, gm0, gn0, p.N, p.bias, p.residual, p.aux, full);
            lean_store<2, true>(p, acc + 2, st, le, gm0 + 32, gn0, p.N, p.bias, p.residual, p.aux, full);
        }
        // (the next tile's first barrier orders these staging reads before the DMA that reuses the slot)
    }
}

// 16-byte chunk swizzle of LDS images with 64-byte rows (4 chunks): slot = chunk ^ s(row >> 2 & 3) with s = (0,2,3,1) keeps every
// 16-lane group of a ds_read_b128 fragment read on 16 different 16-byte bank units (k_lnlin320's weight stages).
__device__ __forceinline__ int h_swz(int row_in_16) { return (0x78 >> (2 * (row_in_16 >> 2))) & 3; }
