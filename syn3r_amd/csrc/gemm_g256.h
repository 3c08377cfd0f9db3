// k_gemm_g256: the GEGLU projection on a persistent 256 x 256 tile with a SECOND look-ahead stage for the A operand (round 6;
// VERDICT r05 item 1c).  Included by gemm.hip inside its anonymous namespace.
//
// What it is for.  The gated projections of the feed-forward at levels 1-2 ([64512, 5120, 640] and [16128, 10240, 1280]) lose
// 25-45 % of their time where a tile's output stream meets the next tile's operand stream: the 256 x 320 kernels keep ONE k-tile
// of look-ahead (two 73,728-byte stages fill the LDS), and the wait in front of the next tile's first k-tiles also waits for the
// tile's stores to be acknowledged.  Here:
//   tile      256 rows x 256 packed columns = [64 hidden | 64 gate] per 128-column group and wavefront (weights packed per 128
//             rows by ops.pack_geglu(..., group=64)), 8 wavefronts = 4 row groups x 2 column groups, 128 accumulators each;
//   LDS       THREE 32 KB slots for A and TWO for B = 163,840 B: the A pieces of k-tile g + 2 and the B pieces of k-tile g + 1 are
//             requested during k-tile g (A streams from HBM / the memory-side cache, the band's weight panel sits in the L2), the
//             stage cursors run on across tile boundaries;
//   epilogue  no LDS: the gate runs in registers (activations.py GEGLU.forward on the fp16-rounded projection, gelu_pk, packed fp32
//             throughout) and the result goes out straight from the MFMA layout - two v_permlane16_swap per pair of column tiles give
//             every lane a 16-byte piece; the A-tiled layout of the hidden activation (GemmParams::out_tiled: [128-row block][64-column
//             tile][128][64]) makes the wavefront's 64 x 64 result ONE contiguous 8 KB run, a store instruction covers 16 rows x 64 bytes;
//   waits     at a tile boundary (one extra barrier) the next tile's stage 1 of B and stage 2 of A are requested into the slots the last
//             k-tile released, the gate runs while they land, and only then - everything the next tile's k-tiles 0 and 1 read has
//             landed - are the stores issued: those two k-tiles need no vmcnt wait, and the first counted wait that includes the
//             stores' acknowledgements comes two k-tiles after their issue (ablation: the same store instructions with ONE active
//             lane cost 70 % of what the full stores cost - it is their acknowledgement inside a counted wait, not their bytes);
//   bias      the epilogue's bias values are requested at the START of the tile by inline asm (left to the compiler: four load +
//             s_waitcnt vmcnt(0) round trips inside the epilogue).
// Arithmetic: the same MFMA order per accumulator as k_gemm_widep / k_gemm_z (k ascending, v_mfma_f32_16x16x32_f16 with the weight
// fragment as the A operand), the same gate: results are bit-identical to those kernels' (tests/test_unet_ops_gpu.py).
// Reference semantics: attention.py:608-665 (FeedForward), activations.py GEGLU.
// (The ablation / variant builds behind profiles/r06/g256_ablations.txt - no stores, one-lane stores, cache-resident stores, 8-byte stores,
// parked and spread stores, de-phased CUs - are profiles/r06/g256_ablation_variants.patch against this file.)

constexpr int G_SLOT = 256 * BK * 2;             // 32,768: one operand stage (256 rows x 64 halfs)
constexpr int G_B0 = 3 * G_SLOT;                 // A slots at 0 / 32 K / 64 K, B slots at 96 K / 128 K
constexpr int G_LDS = 5 * G_SLOT;                // 163,840
#ifdef SYN3R_TIMING
__device__ unsigned long long g_g256_timing[64];
#endif

__global__ void __launch_bounds__(512, 2) k_gemm_g256(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;  // (launch_g256: M a multiple of 128 - the last row tile may be a half: its rows 128-255 re-read valid rows and are never stored)
    const unsigned nblk = (unsigned)(tiles_m * tiles_n);
    const unsigned xcd = blockIdx.x % 8, q8 = nblk / 8, r8 = nblk % 8;
    const unsigned t_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned t_len = q8 + (xcd < r8 ? 1u : 0u);
    const unsigned t_stride = (gridDim.x - xcd + 7) / 8;
    if (blockIdx.x / 8 >= t_len) return;                          // (block-uniform, before any barrier)
    const int nkt = p.K / BK;
    // tile order: bands of `band` tile columns, row-major inside a band (k_gemm_widep's: an XCD's 32 tiles share the band's weight panel)
    const unsigned bw0 = p.band > 0 ? (unsigned)p.band : 4u;
    const unsigned bw = (unsigned)tiles_n >= bw0 ? bw0 : (unsigned)tiles_n;
    const unsigned band_sz = (unsigned)tiles_m * bw, full_bands = (unsigned)tiles_n / bw;
    auto tile_origin = [&](unsigned tile, int& m0, int& tn) {
        unsigned b = tile / band_sz, w = bw, t2 = tile - b * band_sz;
        if (b >= full_bands) { b = full_bands; t2 = tile - full_bands * band_sz; w = (unsigned)tiles_n - full_bands * bw; }
        tn = (int)(b * bw + t2 % w);
        m0 = (int)(t2 / w) * 256;
    };

    // ---- the two stage cursors (scalar state).  Each numbers its operand's k-tile stages through the block's whole tile list and
    // writes ring slot (stage mod 3) / (stage mod 2); once the list is exhausted a cursor re-requests its last stage (valid
    // addresses, into slots nobody reads again), so the main loop carries no "is there a next stage" branch around its DMAs.
    unsigned oa[4] = {0u, 0u, 0u, 0u}, ob[4] = {0u, 0u, 0u, 0u};   // byte offsets (from p.A / p.W) of this wavefront's four pieces
    unsigned ca_tl = blockIdx.x / 8, cb_tl = blockIdx.x / 8;
    int ca_ks = 0, cb_ks = 0;
    unsigned ca_slot = 0, cb_slot = 0;
    unsigned voff_a, voff_b;
    {
        const int prow = lane >> 3, csrc = (lane & 7) ^ prow;
        voff_a = (unsigned)(prow * (int)p.lda + csrc * 8) * 2u;
        voff_b = (unsigned)(prow * p.K + csrc * 8) * 2u;
    }
    auto issue_a = [&]() {
        if (ca_tl < t_len) {
            if (ca_ks == 0) {
                int m0_, tn_;
                tile_origin(t_start + ca_tl, m0_, tn_);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int r = m0_ + wv * 32 + i * 8;
                    r = r < p.M - 8 ? r : p.M - 8;
                    oa[i] = 2u * (unsigned)r * (unsigned)p.lda;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) oa[i] += 2u * BK;
            }
            if (++ca_ks == nkt) { ca_ks = 0; ca_tl += t_stride; }
        }
        char* st = smem_raw + ca_slot * G_SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)p.A + (size_t)(oa[i] + voff_a)), (lds_void_t*)(st + (wv * 4 + i) * 1024), 16, 0, 0);
        ca_slot = ca_slot == 2 ? 0u : ca_slot + 1;
    };
    auto issue_b = [&]() {
        if (cb_tl < t_len) {
            if (cb_ks == 0) {
                int m0_, tn_;
                tile_origin(t_start + cb_tl, m0_, tn_);
#pragma unroll
                for (int j = 0; j < 4; ++j) ob[j] = 2u * (unsigned)(tn_ * 256 + wv * 32 + j * 8) * (unsigned)p.K;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) ob[j] += 2u * BK;
            }
            if (++cb_ks == nkt) { cb_ks = 0; cb_tl += t_stride; }
        }
        char* st = smem_raw + G_B0 + cb_slot * G_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)p.W + (size_t)(ob[j] + voff_b)), (lds_void_t*)(st + (wv * 4 + j) * 1024), 16, 0, 0);
        cb_slot ^= 1u;
    };

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    const bool defer = wv >= 4;                 // SIMD partners staggered by half a k-tile (k_gemm_dma / k_gemm_widep)
    unsigned ra = 0, rb = 0;                    // ring slots of the k-tile being read
    float4v acc[8][4];                          // [column tile: 0-3 hidden, 4-7 gate][row tile]

    // kernel prologue: the first tile's stages 0 and 1 of both operands and stage 2 of A - what every later tile finds requested
    // (and landed) when it starts, see the tile boundary below
    issue_a(); issue_b(); issue_a(); issue_b(); issue_a();
    bool first_tile = true;
#ifdef SYN3R_TIMING     // tools/g256_timing.py: s_memtime ticks per segment of a tile, summed over one block's tiles, per wavefront
    unsigned long long gt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g_a = __builtin_amdgcn_s_memtime();
#define GSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gt[i] += t_ - g_a; g_a = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define GSTAMP(i)
#endif
    typedef _Float16 half4e __attribute__((ext_vector_type(4)));
    for (unsigned tl = blockIdx.x / 8; tl < t_len; tl += t_stride) {
        int m0, tile_n;
        tile_origin(t_start + tl, m0, tile_n);
        unsigned a_row, b_row, swz[2];
        {   // lane-derived indices behind an opaque copy of the lane id: rebuilt per tile instead of carried through the epilogue
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, fq = lo >> 4;
            a_row = (unsigned)((wm * 64 + fr) * 128);
            b_row = (unsigned)(G_B0 + (wn * 128 + fr) * 128);
            swz[0] = (unsigned)(((0 + fq) ^ (fr & 7)) << 4);
            swz[1] = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
        }
        // the epilogue's bias values, requested NOW (inline asm: left to the compiler they are four load + s_waitcnt vmcnt(0) round trips
        // inside the epilogue); they are older than every stage request of this tile, so the waits of the k-loop cover them
        half4e bh[4], bg[4];
        {
            const __half* bptr = p.bias + tile_n * 256 + wn * 128 + (lane >> 4) * 4;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(bh[0]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:32" : "=v"(bh[1]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:64" : "=v"(bh[2]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:96" : "=v"(bh[3]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:128" : "=v"(bg[0]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:160" : "=v"(bg[1]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:192" : "=v"(bg[2]) : "v"(bptr));
            asm volatile("global_load_dwordx2 %0, %1, off offset:224" : "=v"(bg[3]) : "v"(bptr));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (float4v){0.f, 0.f, 0.f, 0.f};
        half8 af[4], bf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(af[i]));
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "=v"(bf[j]));
        // A k-half: twelve fragment reads (the four A fragments first), then the MFMAs column tile by column tile, each group of
        // four behind a COUNTED wait for its own weight fragment - the first group issues when five of the twelve reads have arrived
        auto read_half = [&](int kh) {
            const unsigned aa = lds0 + ra * G_SLOT + a_row + swz[kh], ba = lds0 + rb * G_SLOT + b_row + swz[kh];
            DS_READ128(af[0], aa, 0); DS_READ128(af[1], aa, 2048); DS_READ128(af[2], aa, 4096); DS_READ128(af[3], aa, 6144);
            DS_READ128(bf[0], ba, 0); DS_READ128(bf[1], ba, 2048); DS_READ128(bf[2], ba, 4096); DS_READ128(bf[3], ba, 6144);
            DS_READ128(bf[4], ba, 8192); DS_READ128(bf[5], ba, 10240); DS_READ128(bf[6], ba, 12288); DS_READ128(bf[7], ba, 14336);
        };
#define G256_GROUP(J) do { \
            if constexpr ((J) == 0) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bf[0])); \
            else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bf[J]) : "i"(7 - (J))); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[J][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[J], af[i], acc[J][i], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); } while (0)
        auto mma = [&]() {
            G256_GROUP(0); G256_GROUP(1); G256_GROUP(2); G256_GROUP(3); G256_GROUP(4); G256_GROUP(5); G256_GROUP(6); G256_GROUP(7);
        };
        for (int kt = 0; kt < nkt; ++kt) {
            // stage kt of both operands has landed once only the youngest four requests (the A pieces of stage kt + 1) are in flight.
            // k-tiles 0 and 1 find their stages landed: they were requested, and waited for, in front of the previous tile's stores
            // (the kernel's first tile: requested by the prologue) - the first counted wait that includes those stores' acknowledgements
            // is k-tile 2's, two k-tiles after they were issued
            if (kt >= 2) {
                GSTAMP(4);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#ifdef SYN3R_TIMING
                if (kt == 2) GSTAMP(1) else GSTAMP(2)
#endif
            }
            else if (first_tile && kt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt >= 2) { GSTAMP(3); }
            if (defer && kt > 0) mma();                          // second k-half of stage kt - 1 (fragments read before the barrier)
            if (kt > 0) {
            issue_b();                                           // stage kt + 1 of B: the slot every wavefront finished reading in iteration kt - 1
            issue_a();                                           // stage kt + 2 of A: likewise
            }
            read_half(0);
            mma();
            read_half(1);
            // (drained in every wavefront: the deferred ones multiply behind the next barrier, whose DMA refills the slot these reads come from)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]),
                           "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), "+v"(bf[7]));
            if (!defer) mma();
            ra = ra == 2 ? 0u : ra + 1;
            rb ^= 1u;
            if (kt == 1) { GSTAMP(0); }
        }
        if (defer) mma();
        first_tile = false;
        GSTAMP(4);
        // tile boundary: every wavefront is done with the last k-tile's slots - the next tile's stage 1 of B and stage 2 of A go out
        // now and land under the gate
        __builtin_amdgcn_s_barrier();
        issue_b(); issue_a();

        // ---- epilogue: gate in registers, 16-byte pieces (v_permlane16_swap) stored straight into the A-tiled hidden activation
        int le = lane;
        asm volatile("" : "+v"(le));
        const int fr = le & 15, fq = le >> 4;
        const int gn = tile_n * 256 + wn * 128;                  // packed column origin of this wavefront: [64 hidden | 64 gate]
        const int gm0 = m0 + wm * 64, go0 = tile_n * 128 + wn * 64;
        unsigned o2[4][4][2];                                    // [row tile][column tile][column pair]: the gated result as packed fp16 pairs, 32 registers
        // (the bias loads are older than everything this wait could leave in flight)
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]), "+v"(bg[0]), "+v"(bg[1]), "+v"(bg[2]), "+v"(bg[3]) :: "memory");
        typedef _Float16 half2e __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // packed arithmetic throughout: bias add (v_pk_add_f32), rounding of the projection to fp16 and back (v_cvt_pk_f16_f32 + two
            // v_cvt_f32_f16), gate, rounding of the result - the values and roundings of the scalar form (activations.py GEGLU on fp16)
            const syn3r_f2 bhf[2] = {__builtin_convertvector((half2e){bh[j][0], bh[j][1]}, syn3r_f2), __builtin_convertvector((half2e){bh[j][2], bh[j][3]}, syn3r_f2)};
            const syn3r_f2 bgf[2] = {__builtin_convertvector((half2e){bg[j][0], bg[j][1]}, syn3r_f2), __builtin_convertvector((half2e){bg[j][2], bg[j][3]}, syn3r_f2)};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const half2e hh = __builtin_convertvector((syn3r_f2){acc[j][i][r], acc[j][i][r + 1]} + bhf[r >> 1], half2e);
                    const half2e gh = __builtin_convertvector((syn3r_f2){acc[j + 4][i][r], acc[j + 4][i][r + 1]} + bgf[r >> 1], half2e);
                    const syn3r_f2 hv = __builtin_convertvector(hh, syn3r_f2), gv = __builtin_convertvector(gh, syn3r_f2);
                    const syn3r_f2 y = hv * gelu_pk(gv);
                    o2[i][j][r >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(y, half2e));
                }
        }
        // every stage the next tile's k-tiles 0 and 1 read has landed (requested at least a gate's length ago): nothing but this tile's
        // stores is in flight when that tile starts
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GSTAMP(5);
        const bool rows_in = gm0 < p.M;                          // (wave-uniform: a half last tile's wavefronts 4-7 have nothing to store)
        {
            // lanes of 16-lane row q hold columns 4 q .. 4 q + 3 of a column tile; after the swaps rows 0 / 2 hold columns 0-7 / 8-15 of tile
            // j and rows 1 / 3 those of tile j + 1: piece (row q) = tile j + (q & 1), columns 8 (q >> 1) .. + 7
            __half* pb = p.out + tiled_off(gm0, go0, p.geglu_D) + fr * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    auto r0 = __builtin_amdgcn_permlane16_swap(o2[i][j][0], o2[i][j + 1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(o2[i][j][1], o2[i][j + 1][1], false, false);
                    const u32x4 v = (u32x4){(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                    u32x4* dst = (u32x4*)(pb + i * 16 * 64 + j * 16);
                    if (rows_in) { if (p.out_nt) __builtin_nontemporal_store(v, dst); else *dst = v; }
                }
        }
        GSTAMP(6);
#ifdef SYN3R_TIMING
        ++gt[7];
#endif
    }
#ifdef SYN3R_TIMING
    if (blockIdx.x == gridDim.x / 2 && lane == 0)
        for (int i = 0; i < 8; ++i) g_g256_timing[wv * 8 + i] = gt[i];
#endif
#undef GSTAMP
#undef G256_GROUP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the exhausted cursors' last requests)
}
