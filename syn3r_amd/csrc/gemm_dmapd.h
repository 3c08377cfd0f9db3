// k_gemm_dmapd: the persistent 256 x 160 dense kernel with a deferred, LDS-free epilogue (built in round 5, dispatched in round 6
// for the shapes it won on: profiles/r05/dmapd_deferred_epilogue.txt).
// Included by gemm.hip inside its anonymous namespace (one translation unit; the kernels share GemmParams, the epilogues and the
// LDS-DMA typedefs of gemm_common.h / gemm_dma.h).

// ---------------------------------------------------------------------------------------------
// k_gemm_dmapd (round 5): the persistent 256 x 160 dense kernel with a DEFERRED, LDS-free epilogue.  The K <= 1280 projections
// with a residual (attn1.to_out, proj_out: attention_processor.py:187-202, transformer_temporal.py:371-379) spend as long in
// their epilogue as in their k-loop - a CU drains ~9 B of stores per clock and its matrix pipe idles meanwhile (DESIGN.md section
// 4) - and on the 256 x 320 tile the registers to overlap the two do not exist (160 accumulators; hipcc spilled every
// formulation).  This tile has 80: at the end of a tile's k-loop the result is PARKED as fp16((acc + bias + row vector) * s_acc) in
// 40 registers - the value the one-pass epilogue rounds first - and the tile's residual loads, adds and stores are issued as
// four UNITS (one per 16-row tile) spread over the NEXT tile's k-tiles: per lane two 16-byte pieces (two v_permlane16_swap per
// pair of column tiles leave every lane with eight consecutive columns: 64-byte row segments, no LDS) and one 8-byte piece.
// Loads go out behind the k-tile's barrier, the adds and stores behind its MFMAs.  vmcnt bookkeeping: at the top of k-tile k + 1 the
// stage of k-tile k + 1 (requested during k-tile k - 1) must have landed; the counted wait lets only the youngest operations stay in
// flight - the stores of k-tile k and six pieces of the stage requested during k-tile k - which also retires that iteration's unit
// loads (consumed already) and the stores of the iteration before: a superset of what is needed, never less.
// Same arithmetic per output element as lean_store (same roundings in the same order): identical bits.
struct DmapdParked { int gm0, gn0, valid; };

__global__ void __launch_bounds__(512, 2) k_gemm_dmapd(GemmParams p) {
    constexpr int BM = 256;
    constexpr int DMA_A_BYTES = BM * BK * 2;                  // 32,768
    constexpr int STAGE = DMA_A_BYTES + DMA_B_BYTES;          // 53,248
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef _Float16 half4e __attribute__((ext_vector_type(4)));
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

    // ---- issue cursor (as k_gemm_dmap, dense mode).  Whole tiles only (M % 256 == 0, N % 160 == 0: dmapd_admits), so the four A
    // pieces and the B pieces of a wavefront are ONE per-lane pointer each plus wave-uniform strides: 4 registers of source state
    const int prow = lane >> 3;
    const int csrc = (lane & 7) ^ prow;
    const int nb = wv < 4 ? 3 : 2;
    const int b_first = wv < 4 ? wv * 3 : 12 + (wv - 4) * 2;
    const __half* a_cur;
    const __half* b_cur;
    const long long a_piece = p.a_tiled ? 512 : 8 * p.lda;            // halfs between the pieces (8 rows apart) of a wavefront
    const int a_inc = p.a_tiled ? 8192 : BK;
    const long long b_piece = 8ll * p.K;
    unsigned itl = blockIdx.x / 8;
    int ikt = 0, islot = 0;
    auto setup_issue_tile = [&](unsigned tile) {
        const int m0 = (int)(tile / (unsigned)tiles_n) * BM, n0 = (int)(tile % (unsigned)tiles_n) * BN;
        const int m = m0 + wv * 32 + prow;
        a_cur = p.a_tiled ? p.A + (long long)(m >> 7) * (p.K >> 6) * 8192 + (m & 127) * 64 + csrc * 8
                          : p.A + (long long)m * p.lda + csrc * 8;
        b_cur = p.W + (long long)(n0 + b_first * 8 + prow) * p.K + csrc * 8;
    };
    auto issue_next = [&]() -> bool {
        if (itl >= t_len) return false;
        if (ikt == 0) setup_issue_tile(t_start + itl);
        char* st = smem_raw + islot * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_cur + i * a_piece), (lds_void_t*)(st + (wv * 4 + i) * 1024), 16, 0, 0);
        a_cur += a_inc;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nb) __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_cur + j * b_piece), (lds_void_t*)(st + DMA_A_BYTES + (b_first + j) * 1024), 16, 0, 0);
        b_cur += BK;
        if (++ikt == nkt) { ikt = 0; itl += t_stride; }
        if (++islot == 3) islot = 0;
        return true;
    };

    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const unsigned a_row = (unsigned)((wm * WM + fr) * 128);
    const unsigned b_row = (unsigned)(DMA_A_BYTES + (wn * WN + fr) * 128);
    const unsigned sw0 = (unsigned)(((0 + fq) ^ (fr & 7)) << 4), sw1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    const bool defer = wv >= 4;           // stagger of the SIMD partners (see k_gemm_widep)

    // ---- the parked tile and its four units (one per 16-row tile: two 16-byte pieces + one 8-byte piece per lane)
    half4e pk[TM][TN];                    // fp16((acc + bias + rowvec) * s_acc) of the tile whose epilogue is in progress
    DmapdParked park{0, 0, 0};
    // column (inside the wavefront's 80) of this lane's piece: pairs give 8 halfs from column 32 c + (fq even ? 4 fq : 16 + 4 (fq - 1)),
    // the fifth column tile 4 halfs from column 64 + 4 fq
    const int pair_col = (fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq;
    half8 lres[2];                        // residual pieces of the unit of the current k-tile: two 16-byte ...
    half4e lres4;                         // ... and one 8-byte piece per lane (aux blends keep the one-pass kernel: registers)
    // The loads are inline asm: hipcc cannot count the LDS-DMA instructions issued behind them (they sit under wave-uniform
    // branches), so a visible load would be waited for with vmcnt(0) - which drains the DMA stream every k-tile.  Issued BEFORE the
    // k-tile's DMA, the 6-7 DMA instructions are all that is younger: the counted wait in front of their use is vmcnt(6).
    auto unit_load = [&](int i) __attribute__((always_inline)) {
        const int m = park.gm0 + i * 16 + fr;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n = park.gn0 + (c < 2 ? 32 * c + pair_col : 64 + 4 * fq);
            if (p.residual) {
                const __half* src = p.residual + (long long)m * p.ldr + n;
                if (c < 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lres[c < 2 ? c : 0]) : "v"(src) : "memory");
                else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(lres4) : "v"(src) : "memory");
            }
        }
    };
    // one unit's adds + stores.  The parked tile is a QUEUE of row tiles: the unit always works on pk[0] and then moves the other
    // three up (30 v_mov per unit) - one copy of this code with static register indices instead of a four-way switch.
    auto unit_store_rt = [&](int i) __attribute__((always_inline)) {
        if (p.residual)                   // the unit's loads have landed (younger: this k-tile's DMA only)
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(lres[0]), "+v"(lres[1]), "+v"(lres4) :: "memory");
        const int m = park.gm0 + i * 16 + fr;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            half8 v;
            if (c < 2) {
                const half4e x = pk[0][2 * c], y = pk[0][2 * c + 1];
                unsigned xr[2], yr[2];
                __builtin_memcpy(xr, &x, 8); __builtin_memcpy(yr, &y, 8);
                unsigned o[4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    auto r = __builtin_amdgcn_permlane16_swap(xr[k], yr[k], false, false);
                    o[k] = r[0]; o[2 + k] = r[1];
                }
                __builtin_memcpy(&v, o, 16);
            } else {
                const half4e x = pk[0][4];
                v = (half8){x[0], x[1], x[2], x[3], 0, 0, 0, 0};
            }
            const int W = c < 2 ? 8 : 4;
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 8; ++e)           // the same rounding as lean_store: fp16(float(v) + s_res * residual)
                    if (e < W) v[e] = (_Float16)((float)v[e] + p.s_res * (c < 2 ? (float)lres[c < 2 ? c : 0][e] : (float)lres4[e & 3]));
            }
            const int n = park.gn0 + (c < 2 ? 32 * c + pair_col : 64 + 4 * fq);
            if (c < 2) *(half8*)(p.out + (long long)m * p.ldc + n) = v;
            else *(half4e*)(p.out + (long long)m * p.ldc + n) = (half4e){v[0], v[1], v[2], v[3]};
        }
#pragma unroll
        for (int r = 0; r + 1 < TM; ++r)
#pragma unroll
            for (int j = 0; j < TN; ++j) pk[r][j] = pk[r + 1][j];
    };

    int issued = 0, consumed = 0;
    if (issue_next()) ++issued;
    if (issue_next()) ++issued;
    int cslot = 0;
    bool first_tile = true;
    int prev_stores = 0;                  // stores this wavefront issued in the previous k-tile iteration (wave-uniform)
    for (unsigned tl = blockIdx.x / 8; tl < t_len; tl += t_stride) {
        const unsigned tile = t_start + tl;
        const int m0 = (int)(tile / (unsigned)tiles_n) * BM, n0 = (int)(tile % (unsigned)tiles_n) * BN;
        float4v acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
        half8 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { asm volatile("" : "=v"(a0[i])); asm volatile("" : "=v"(a1[i])); }
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
            // the unit of the parked tile that belongs to this k-tile: unit u runs in k-tile (u nkt) / 4 (nkt >= 4: at most one per k-tile)
            int unit = -1;
            if (park.valid) {
#pragma unroll
                for (int u = 0; u < TM; ++u) if ((u * nkt) / TM == kt) unit = u;
            }
            if ((kt == 0 && !first_tile) || issued - consumed < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (prev_stores) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // the previous k-tile's three stores may stay in flight
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (defer && kt > 0) mma1();
            if (unit >= 0) unit_load(unit);                                // (before the DMA in program order: see the vmcnt note)
            if (issue_next()) ++issued;
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
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(b1[4]));
            if (!defer) mma1();
            prev_stores = 0;
            if (unit >= 0) {                                               // behind this k-tile's MFMAs
                unit_store_rt(unit);
                prev_stores = 1;                                           // three store instructions (whole tiles: every lane stores)
            }
            __builtin_amdgcn_sched_barrier(0);
            ++consumed;
            if (++cslot == 3) cslot = 0;
        }
        if (defer) mma1();
        first_tile = false;
        // ---- park this tile: fp16((acc + bias + row vector) * s_acc), the one-pass epilogue's first rounding.  Every bias / row-vector
        // piece is requested before the first one is used (one exposed latency per tile, not twenty-five)
        {
            const int gm0 = m0 + wm * WM, gn0 = n0 + wn * WN;
            half4e bv[TN];
            const half4e z4 = {0, 0, 0, 0};
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *(const half4e*)(p.bias + gn0 + j * 16 + fq * 4);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = z4;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {              // a row tile's five row-vector pieces together
                half4e rvv[TN];
                if (p.rowvec) {
                    const __half* rv = p.rowvec + (long long)rowvec_index(gm0 + i * 16 + fr, p.rows_per_vec, p.rv_group) * p.ldrv + gn0 + fq * 4;
#pragma unroll
                    for (int j = 0; j < TN; ++j) rvv[j] = *(const half4e*)(rv + j * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = (float)bv[j][r];                         // the order of lean_store: bias, + row vector, + accumulator, scale
                        if (p.rowvec) a += (float)rvv[j][r];
                        pk[i][j][r] = (_Float16)((acc[i][j][r] + a) * p.s_acc);
                    }
            }
            park.gm0 = gm0; park.gn0 = gn0; park.valid = 1;
        }
    }
    // ---- the last tile's epilogue: its four units
    if (park.valid) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            unit_load(u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (nothing younger than the loads here: the counted wait inside would not cover them)
            unit_store_rt(u);
        }
    }
}

