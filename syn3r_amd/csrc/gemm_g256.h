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
//   epilogue  no LDS: the gate runs in registers (activations.py GEGLU.forward on the fp16-rounded projection, gelu_pk) and the
//             result goes out as 8-byte stores straight from the MFMA layout - the A-tiled layout of the hidden activation
//             (GemmParams::out_tiled: [128-row block][64-column tile][128][64]) makes the wavefront's 64 x 64 result ONE contiguous
//             8 KB run, a store instruction covers 16 rows x 32 bytes and four of them complete the rows' 128-byte lines;
//   waits     the next tile's first stages are waited for BEFORE the stores are issued (they were requested one and two k-tiles
//             earlier), so that tile's first k-tile needs no vmcnt wait at all and the stores have a whole k-tile to be
//             acknowledged before a counted wait includes them.
// Arithmetic: the same MFMA order per accumulator as k_gemm_widep / k_gemm_z (k ascending, v_mfma_f32_16x16x32_f16 with the weight
// fragment as the A operand), the same gate: results are bit-identical to those kernels' (tests/test_unet_ops_gpu.py).
// Reference semantics: attention.py:608-665 (FeedForward), activations.py GEGLU.

constexpr int G_SLOT = 256 * BK * 2;             // 32,768: one operand stage (256 rows x 64 halfs)
constexpr int G_B0 = 3 * G_SLOT;                 // A slots at 0 / 32 K / 64 K, B slots at 96 K / 128 K
constexpr int G_LDS = 5 * G_SLOT;                // 163,840

__global__ void __launch_bounds__(512, 2) k_gemm_g256(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int tiles_n = p.N / 256, tiles_m = p.M / 256;          // (launch_g256: whole tiles only)
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
                for (int i = 0; i < 4; ++i) oa[i] = 2u * (unsigned)(m0_ + wv * 32 + i * 8) * (unsigned)p.lda;
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

    // kernel prologue, in the steady state's order (B of stage g + 1 before A of stage g + 2): A0, B0, A1
    issue_a(); issue_b(); issue_a();
    bool first_tile = true;
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
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (float4v){0.f, 0.f, 0.f, 0.f};
        half8 af[4], bf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(af[i]));
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "=v"(bf[j]));
        auto read_half = [&](int kh) {
            const unsigned aa = lds0 + ra * G_SLOT + a_row + swz[kh], ba = lds0 + rb * G_SLOT + b_row + swz[kh];
            DS_READ128(af[0], aa, 0); DS_READ128(af[1], aa, 2048); DS_READ128(af[2], aa, 4096); DS_READ128(af[3], aa, 6144);
            DS_READ128(bf[0], ba, 0); DS_READ128(bf[1], ba, 2048); DS_READ128(bf[2], ba, 4096); DS_READ128(bf[3], ba, 6144);
            DS_READ128(bf[4], ba, 8192); DS_READ128(bf[5], ba, 10240); DS_READ128(bf[6], ba, 12288); DS_READ128(bf[7], ba, 14336);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]),
                           "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), "+v"(bf[7]));
        };
        auto mma = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // the next reads reuse af / bf: keep them behind these MFMAs
        };
        for (int kt = 0; kt < nkt; ++kt) {
            // stage kt of both operands has landed once only the youngest four requests (the A pieces of stage kt + 1) are in flight;
            // a later tile's first stages were waited for in front of the previous tile's stores
            if (kt > 0 || first_tile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (defer && kt > 0) mma();                          // second k-half of stage kt - 1 (fragments read before the barrier)
            issue_b();                                           // stage kt + 1 of B: the slot every wavefront finished reading in iteration kt - 1
            issue_a();                                           // stage kt + 2 of A: likewise
            read_half(0);
            mma();
            read_half(1);
            if (!defer) mma();
            ra = ra == 2 ? 0u : ra + 1;
            rb ^= 1u;
        }
        if (defer) mma();
        first_tile = false;
        // the next tile's stage 0 (both operands) has landed; its A stage 1 may still be in flight.  No store is outstanding here.
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");

        // ---- epilogue: gate in registers, 8-byte stores from the MFMA layout into the A-tiled hidden activation
        int le = lane;
        asm volatile("" : "+v"(le));
        const int fr = le & 15, fq = le >> 4;
        const int gn = tile_n * 256 + wn * 128;                  // packed column origin of this wavefront: [64 hidden | 64 gate]
        const int gm0 = m0 + wm * 64, go0 = tile_n * 128 + wn * 64;
        typedef _Float16 half4e __attribute__((ext_vector_type(4)));
        __half* obase = p.out + tiled_off(gm0, go0, p.geglu_D) + fr * 64 + fq * 4;
        half4e o[4][4];                                          // [row tile][column tile]: the gated result, 32 registers
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = gn + j * 16 + fq * 4;
            float bh[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                const half4e b0 = *(const half4e*)(p.bias + n), b1 = *(const half4e*)(p.bias + n + 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) { bh[r] = (float)b0[r]; bg[r] = (float)b1[r]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const syn3r_f2 hv = (syn3r_f2){(float)(_Float16)(acc[j][i][r] + bh[r]), (float)(_Float16)(acc[j][i][r + 1] + bh[r + 1])};
                    const syn3r_f2 gv = (syn3r_f2){(float)(_Float16)(acc[j + 4][i][r] + bg[r]), (float)(_Float16)(acc[j + 4][i][r + 1] + bg[r + 1])};
                    const syn3r_f2 y = hv * gelu_pk(gv);
                    o[i][j][r] = (_Float16)y.x; o[i][j][r + 1] = (_Float16)y.y;
                }
        }
        // four consecutive stores complete the 128-byte lines of 16 rows
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half4e* dst = (half4e*)(obase + i * 16 * 64 + j * 16);
                if (p.out_nt) __builtin_nontemporal_store(o[i][j], dst); else *dst = o[i][j];
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the exhausted cursors' last requests)
}
