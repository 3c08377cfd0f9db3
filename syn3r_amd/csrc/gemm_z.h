// k_gemm_z: the persistent 256 x 320 dense contraction with a SOFTWARE-PIPELINED main loop (round 4).  Included by gemm.hip
// (inside its anonymous namespace, after GemmParams / rowvec_index / tiled_off / OUT_STORE / the LDS-DMA typedefs).
//
// What it replaces and why: k_gemm_widep opened every k-tile with
// `s_waitcnt vmcnt(0)` + barrier and only THEN read its fragments - all eight wavefronts at once, 14 x ds_read_b128 drained by
// lgkmcnt(0) in front of each block of 40 MFMAs - so a wavefront never overlapped its own LDS reads with its own MFMAs
// (4 100 cycles per k-tile for 2 560 of matrix work, MFMA busy 0.39).  Here the same tile, ring and DMA pieces run as three
// streams per wavefront that only meet at ONE barrier per k-tile:
//   read stream   the B (weight) fragment of MFMA group Q + 3 is requested before group Q's four MFMAs issue (ring of four
//                 fragment registers, counted lgkmcnt), the four A fragments of the next k-half land in a second register set
//                 while the current k-half multiplies: 160 accumulators + 32 + 16 fragment registers;
//   MFMA stream   20 groups of 4 per k-tile (weight fragment j against the four row fragments), never waiting for more than
//                 the one fragment it is about to use;
//   DMA stream    the nine LDS-DMA pieces of the NEXT k-tile go out one per group in groups 0-8, into the ring slot the
//                 previous barrier released, and have until group 17 to land.
// The barrier sits where the READ stream crosses from k-tile G to G + 1 (group 17): in front of it `lgkmcnt(0)` (this
// wavefront has no read of slot G left: WAR for the DMA that refills it) and `vmcnt(0)` (its pieces of k-tile G + 1 have
// landed - they were requested >= 8 groups earlier; behind the barrier everybody's have: RAW); behind it the first reads of
// k-tile G + 1, while the MFMAs of groups 18-19 still run from registers.  Nothing is drained in front of matrix work.
// The DMA cursor runs on across tile boundaries (stage 0 of the next tile is requested during this tile's last k-tile and is
// landed and visible when the epilogue ends); the epilogue (k_gemm_widep's lean_store) stages through the ring slot that was
// read last plus a 16 KB gap between the slots, and the output stores are acknowledged under the next tile's k-loop (their
// vmcnt is first waited for at that tile's group 17).
// Reference semantics: nn.Linear / 1x1 convolutions / GEGLU of diffusers attention.py:608-665, resnet.py:316,
// activations.py GEGLU.forward - identical arithmetic to k_gemm_widep (same MFMA order per accumulator: k ascending).

// LDS: [slot 0: 73,728 B][gap 16,384 B][slot 1: 73,728 B] = 163,840 B.  The epilogue stages a tile's accumulators through the
// slot whose k-tile was read last plus the gap (8 x 11,264 B = 90,112 B contiguous either way: slot 0 + gap or gap + slot 1).
constexpr int Z_SLOT1 = W_STAGE + 16384;          // 90,112: byte offset of ring slot 1
constexpr int Z_LDS = Z_SLOT1 + W_STAGE;          // 163,840

#define Z_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))

// Pieces of the main loop as function templates (inline asm in a generic lambda does not capture its operands with this
// compiler): fragment reads, the MFMA group, the counted wait.  Everything is forced inline; the arrays are the kernel's.
template <int KH, int NAF>    // NAF = 2: one register set per k-half (the next half's reads overlap this half's MFMAs); 1: one set
__device__ __forceinline__ void z_read_a(half8 (&af)[NAF][TM], const unsigned (&ra)[2]) {
    Z_DS_READ128(af[KH % NAF][0], ra[KH], 0); Z_DS_READ128(af[KH % NAF][1], ra[KH], 2048);
    Z_DS_READ128(af[KH % NAF][2], ra[KH], 4096); Z_DS_READ128(af[KH % NAF][3], ra[KH], 6144);
}
template <int Q, int RB>      // weight fragment of group Q (0..19) of the k-tile the read cursor is in
__device__ __forceinline__ void z_read_b(half8 (&bf)[RB], const unsigned (&rb)[2]) {
    Z_DS_READ128(bf[Q % RB], rb[Q / 10], (Q % 10) * 2048);
}
template <int Q, int RB, int NAF>
__device__ __forceinline__ void z_mma(float4v (&acc)[2][TM][TN], half8 (&af)[NAF][TM], half8 (&bf)[RB]) {
    constexpr int kh = (Q / 10) % NAF, j = Q % 10;
#pragma unroll
    for (int i = 0; i < TM; ++i)
        acc[j / TN][i][j % TN] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[Q % RB], af[kh][i], acc[j / TN][i][j % TN], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// wait until at most N LDS reads are outstanding; the group's fragment registers are then valid (ties its MFMAs behind the wait)
template <int Q, int RB, int N, int NAF>
__device__ __forceinline__ void z_wait(half8 (&af)[NAF][TM], half8 (&bf)[RB]) {
    constexpr int kh = (Q / 10) % NAF, j = Q % 10;
    if constexpr (j == 0)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(bf[Q % RB]), "+v"(af[kh][0]), "+v"(af[kh][1]), "+v"(af[kh][2]), "+v"(af[kh][3]) : "i"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bf[Q % RB]) : "i"(N));
}

// TWOSRC: the A columns from K1 on come from a second tensor (GemmParams::A2; the up blocks' shortcut projections)
// MODE (round 4): MODE_CONV2D / MODE_TCONV run the implicit-GEMM convolutions on this tile too (they lived on the 256 x 160 tile,
// whose k-loop is LDS-bound: 52 DMA pieces + 144 fragment reads per 1 280 MFMA cycles against 72 + 224 per 2 560 here,
// tools/ubench/lds_dma_rate.hip).  The A pieces then take a PER-LANE source: the im2col gather of resnet.py:274,290 /
// downsampling.py:116-148 / upsampling.py:172-183 / the frame shift of resnet.py:571-597, one 32-bit byte offset per piece
// (or the zero page for padding), re-derived when the filter tap changes and advanced by 128 bytes per k-tile inside a tap.
// UPS (MODE_CONV2D only): the input is read through a nearest-2x upsample (upsampling.py:172-183: F.interpolate(scale 2) then the
// 3 x 3 convolution; the upsampled image is never written).  Tap (ty, tx) of output pixel (y, x) reads input pixel
// ((y + ty - 1) >> 1, (x + tx - 1) >> 1): from the pixel of tap (0, 0) that is (ty + 1 - (y & 1)) >> 1 rows and the same in x
// further - taps 0 and 2 move by 0 / 1 for every pixel, the middle tap by 1 only for EVEN output rows / columns.  So the
// k-tile offset stays wave-uniform except for two lane-dependent additions, steered by two parity bits per piece that share
// the validity register (4 x 6 validity bits + 4 x 2 parity bits = 32).
template <bool TWOSRC, int MODE, bool UPS = false>
__global__ void __launch_bounds__(512, 2) k_gemm_z(GemmParams p) {
    static_assert(!UPS || MODE == MODE_CONV2D, "UPS is a conv2d option");
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
    const int nkt = p.K / BK;
    // tile order: bands of 4 tile columns, row-major inside a band (as k_gemm_widep: the 32 tiles an XCD holds at a time are
    // 8 rows x 4 columns and the band's weight panel is what that XCD's L2 keeps)
    const unsigned bw0 = p.band > 0 ? (unsigned)p.band : 4u;
    const unsigned bw = (unsigned)tiles_n >= bw0 ? bw0 : (unsigned)tiles_n;
    const unsigned band_sz = (unsigned)tiles_m * bw, full_bands = (unsigned)tiles_n / bw;
    auto tile_origin = [&](unsigned tile, int& m0, int& n0, int& tn) {
        unsigned b = tile / band_sz, w = bw, t2 = tile - b * band_sz;
        if (b >= full_bands) { b = full_bands; t2 = tile - full_bands * band_sz; w = (unsigned)tiles_n - full_bands * bw; }
        tn = (int)(b * bw + t2 % w);
        m0 = (int)(t2 / w) * WBM; n0 = tn * WBN;
    };

    // ---- DMA cursor (scalar state): the k-tile stage that is requested next, numbered through the block's whole tile list.
    // The cursor never runs dry inside the loop: once the list is exhausted (only the block's very last k-tile iteration gets
    // there) it keeps re-requesting the last stage - valid addresses, into the slot nobody reads again - so that the main loop
    // has no "is there a next stage" branch around its nine DMA instructions.
    unsigned oa[4], ob[5];                                     // byte offsets of this wavefront's 4 A and 5 B pieces
    const unsigned a_step = p.a_tiled ? 16384u : 2u * BK;
    const char* abase = (const char*)p.A;
    const int kt_switch = TWOSRC ? p.K1 / BK : 0x7fffffff;
    unsigned c_tl = blockIdx.x / 8;                            // cursor: position in the tile list ...
    int c_ks = 0, c_m0 = 0;                                    // ... k-tile inside that tile, its row origin
    unsigned c_slot = 0;                                       // ring slot (byte offset) of the stage being requested
    unsigned voff_a, voff_a2 = 0, voff_b;                      // per-lane byte offset inside a piece
    {
        const int prow = lane >> 3, csrc = (lane & 7) ^ prow;
        voff_a = p.a_tiled ? (unsigned)((prow * 64 + csrc * 8) * 2) : (unsigned)(prow * (int)p.lda + csrc * 8) * 2u;
        if constexpr (TWOSRC) voff_a2 = (unsigned)(prow * (int)p.lda2 + csrc * 8) * 2u;
        voff_b = (unsigned)(prow * p.K + csrc * 8) * 2u;
    }
    bool second = false;                                       // the stage being requested reads the second source
    // convolution modes: per lane and A piece, where the lane's output pixel sits and the byte offset (from p.A) of the 16-byte
    // chunk it fetches for the current k-tile; kPad = the chunk is padding (source: the zero page)
    constexpr unsigned kPad = 0xffffffffu;
    constexpr int NCV = MODE == MODE_DENSE ? 1 : 4;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(MODE == MODE_DENSE ? 0u : p.a_bytes), 0x00020000);
    unsigned cv_off[NCV];
    // MODE_CONV2D walks K as (64-channel chunk, filter tap) with the TAPS INNERMOST - W's [Cout][tap][Cin] rows are read at
    // (tap Cin + 64 chunk) - so the nine shifted reads of one channel slice of an XCD's band of image rows (64 rows x 128 pixels
    // x 64 channels = 1 MB) follow each other and hit that XCD's L2; with the taps outermost every tap re-streams the whole
    // band (5 MB at C = 320) through a 4 MB L2: the PMC passes counted 5.2x the algorithmic bytes (profiles/r04).  The pixel of
    // each piece (y << 16 | x, byte offset of its image) is kept per lane and tile; a k-tile's source offset is a few integer
    // operations from it.  MODE_TCONV keeps taps outermost (3 taps, whole rows: nothing to gain).
    // conv2d per-lane state (no nearest-2x upsample: launch_dma): cv_base[i] = byte offset of the lane's 16-byte chunk at the
    // pixel its output pixel maps to for the CENTRE-less tap (dy = dx = 0 before the padding shift) - every tap and channel chunk
    // is that plus a wave-uniform offset - and ONE register of validity bits, six per piece: tap row 0-2 in range, tap column
    // 0-2 in range.
    unsigned cv_base[MODE == MODE_CONV2D ? 4 : 1], cv_flags = 0;
    int c_tap = 0, c_left = 0, c_chunk = 0;                    // filter tap / k-tiles left in it (tconv) / channel chunk (conv2d)
    unsigned kb = 0;                                           // conv2d: byte offset of the k-tile inside a weight row
    int tap_off = 0;                                           // conv2d: wave-uniform byte offset of k-tile (chunk, tap) from cv_base
    unsigned tap_mask = 0;                                     // conv2d: the two validity bits of the current tap (piece 0's position)
    unsigned add_y = 0, add_x = 0;                             // UPS: what the middle tap adds for even output rows / columns (wave-uniform)
    const int cpb = MODE == MODE_DENSE ? 1 : p.Cin / BK;       // k-tiles per filter tap
    // m / d for m < 2^24 (launch_dma checks M) by one float multiply and a correction step
    auto udiv = [](unsigned m, unsigned d, float rcp, unsigned& rem) -> unsigned {
        unsigned q = (unsigned)((float)m * rcp);
        int r = (int)(m - q * d);
        if (r < 0) { --q; r += (int)d; }
        if (r >= (int)d) { ++q; r -= (int)d; }
        rem = (unsigned)r;
        return q;
    };
    const unsigned d_hw = MODE == MODE_CONV2D ? (unsigned)(p.Ho * p.Wo) : (unsigned)(MODE == MODE_TCONV ? p.HW : 1);
    const unsigned d_w = MODE == MODE_CONV2D ? (unsigned)p.Wo : (unsigned)(MODE == MODE_TCONV ? p.F : 1);
    const float r_hw = 1.0f / (float)d_hw, r_w = 1.0f / (float)d_w;
    auto conv_tile = [&]() {                                   // conv2d: this lane's four output pixels of the tile at c_m0
        if constexpr (MODE == MODE_CONV2D) {
            int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(l));
            const unsigned lanechunk = (unsigned)(((l & 7) ^ (l >> 3)) * 16);
            cv_flags = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned m = (unsigned)(c_m0 + wv * 32 + i * 8 + (l >> 3));
                m = m < (unsigned)p.M ? m : (unsigned)p.M - 1u;
                unsigned r, x;
                const unsigned n = udiv(m, d_hw, r_hw, r), y = udiv(r, d_w, r_w, x);
                unsigned f = 0;
                if constexpr (UPS) {
                    const int yu = (int)y - 1, xu = (int)x - 1;          // tap (0, 0) on the upsampled grid (stride 1, pad 1)
                    const int y0 = yu >> 1, x0 = xu >> 1;                // its input pixel (-1 for the padding row / column)
                    cv_base[i] = (unsigned)(((int)(n * (unsigned)p.Hi) + y0) * p.Wi + x0) * (unsigned)p.Cin * 2u + lanechunk;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        f |= (unsigned)(yu + t >= 0 && yu + t < 2 * p.Hi) << t;
                        f |= (unsigned)(xu + t >= 0 && xu + t < 2 * p.Wi) << (3 + t);
                    }
                    cv_flags |= ((y & 1u) << (24 + 2 * i)) | ((x & 1u) << (25 + 2 * i));
                } else {
                    const int y0 = (int)y * p.stride - p.pad, x0 = (int)x * p.stride - p.pad;   // input pixel of tap (0, 0)
                    cv_base[i] = (unsigned)(((int)(n * (unsigned)p.Hi) + y0) * p.Wi + x0) * (unsigned)p.Cin * 2u + lanechunk;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        f |= (unsigned)(y0 + t >= 0 && y0 + t < p.Hi) << t;
                        f |= (unsigned)(x0 + t >= 0 && x0 + t < p.Wi) << (3 + t);
                    }
                }
                cv_flags |= f << (6 * i);
            }
        }
    };
    auto conv_tap = [&](int tap) {                             // tconv: the chunk offsets of frame tap `tap` for the tile at c_m0
        if constexpr (MODE == MODE_TCONV) {
            int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(l));
            const unsigned chunk = (unsigned)(((l & 7) ^ (l >> 3)) * 16);
#pragma unroll
            for (int i = 0; i < NCV; ++i) {
                unsigned m = (unsigned)(c_m0 + wv * 32 + i * 8 + (l >> 3));
                m = m < (unsigned)p.M ? m : (unsigned)p.M - 1u;
                unsigned r, f;
                const unsigned bf_ = udiv(m, d_hw, r_hw, r);           // m = (b F + f) HW + pixel
                (void)udiv(bf_, d_w, r_w, f);
                const int ff = (int)f + tap - 1;
                cv_off[i] = (ff >= 0 && ff < p.F) ? (unsigned)((int)m + (tap - 1) * p.HW) * (unsigned)p.Cin * 2u + chunk : kPad;
            }
        }
    };
    auto cursor_begin = [&]() {
        if (c_tl >= t_len) return;                             // exhausted: the previous stage's offsets again
        if (c_ks == 0) {
            int n0_, tn_;
            tile_origin(t_start + c_tl, c_m0, n0_, tn_);
            abase = (const char*)p.A;
            second = false;
            if constexpr (MODE != MODE_DENSE) { c_tap = 0; c_left = 0; c_chunk = 0; }
            conv_tile();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = c_m0 + wv * 32 + i * 8;
                r = r < p.M - 8 ? r : p.M - 8;
                if (p.a_tiled) oa[i] = 2u * ((unsigned)(r >> 7) * (unsigned)(p.K >> 6) * 8192u + (unsigned)(r & 127) * 64u);
                else oa[i] = 2u * (unsigned)r * (unsigned)p.lda;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                int n = n0_ + (wv * 5 + j) * 8;
                n = n < p.N - 8 ? n : p.N - 8;
                ob[j] = 2u * (unsigned)n * (unsigned)p.K;
            }
            // (convolution modes: whatever vector-memory operation this once-per-tile branch left pending - hipcc spills inside
            // conv_tile - is drained HERE: left to the merge point, the compiler's scoreboard put an s_waitcnt vmcnt(0) behind the
            // first DMA request of EVERY k-tile)
            if constexpr (MODE != MODE_DENSE) __builtin_amdgcn_s_waitcnt(0x0F70);
        } else if (TWOSRC && c_ks == kt_switch) {              // from here on the A columns come from the second source
            abase = (const char*)p.A2;
            second = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = c_m0 + wv * 32 + i * 8;
                r = r < p.M - 8 ? r : p.M - 8;
                oa[i] = 2u * (unsigned)r * (unsigned)p.lda2;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) ob[j] += 2u * BK;
        } else if constexpr (MODE != MODE_CONV2D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) oa[i] += a_step;
#pragma unroll
            for (int j = 0; j < 5; ++j) ob[j] += 2u * BK;
        }
        if constexpr (MODE == MODE_CONV2D) {                   // k-tile (c_chunk, c_tap), taps innermost; ob[] stay the row offsets
            const int ty = c_tap / 3, tx = c_tap - 3 * ty;
            if constexpr (UPS) {
                tap_off = (((ty == 2 ? p.Wi : 0) + (tx == 2 ? 1 : 0)) * p.Cin + c_chunk * BK) * 2;
                add_y = ty == 1 ? (unsigned)(p.Wi * p.Cin * 2) : 0u;
                add_x = tx == 1 ? (unsigned)(p.Cin * 2) : 0u;
            } else {
                tap_off = ((ty * p.Wi + tx) * p.Cin + c_chunk * BK) * 2;
            }
            tap_mask = (1u << ty) | (8u << tx);
            kb = (unsigned)(c_tap * p.Cin + c_chunk * BK) * 2u;
            if (++c_tap == 9) { c_tap = 0; ++c_chunk; }
        } else if constexpr (MODE == MODE_TCONV) {
            if (c_left == 0) { conv_tap(c_tap); ++c_tap; c_left = cpb; }
            else {
#pragma unroll
                for (int i = 0; i < NCV; ++i) cv_off[i] = cv_off[i] == kPad ? kPad : cv_off[i] + 2u * BK;
            }
            --c_left;
        }
        if (++c_ks == nkt) { c_ks = 0; c_tl += t_stride; }
    };
    auto cursor_end = [&]() { c_slot = c_slot ? 0u : (unsigned)Z_SLOT1; };
    // piece IDX of the stage the cursor points at: 0-3 = A, 4-8 = B
    auto piece = [&](auto IDX) {
        constexpr int idx = decltype(IDX)::value;
        char* st = smem_raw + c_slot;
        // Convolution modes (round 6): the A pieces go through a BUFFER RESOURCE over the input tensor - a padded chunk is an
        // out-of-range offset, for which the LDS-DMA writes zeros (tools/ubench/buffer_lds_oob.hip): one 32-bit offset per lane and
        // piece instead of a 64-bit address selected against a zero page.  (With the 64-bit form hipcc kept the per-lane gather state
        // in scratch and reloaded it - scratch_load + s_waitcnt vmcnt(0), three times per k-tile, in front of the DMA requests of
        // groups 0-2 - and re-fetched the zero page's address with an s_load + lgkmcnt(0) inside the k-loop.)
        if constexpr (idx < 4 && MODE == MODE_CONV2D) {
            const unsigned mk = tap_mask << (6 * idx);
            const bool ok = (cv_flags & mk) == mk;
            unsigned o = cv_base[idx < 4 ? idx : 0] + (unsigned)tap_off;
            if constexpr (UPS) {
                o += (cv_flags & (1u << (24 + 2 * (idx & 3)))) ? 0u : add_y;
                o += (cv_flags & (2u << (24 + 2 * (idx & 3)))) ? 0u : add_x;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void_t*)(st + (wv * 4 + idx) * 1024), 16, (int)(ok ? o : kPad), 0, 0, 0);
        } else if constexpr (idx < 4 && MODE != MODE_DENSE) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void_t*)(st + (wv * 4 + idx) * 1024), 16, (int)cv_off[idx < NCV ? idx : 0], 0, 0, 0);
        } else if constexpr (idx < 4) {
            const unsigned va = (TWOSRC && second) ? voff_a2 : voff_a;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(abase + (size_t)(oa[idx] + va)), (lds_void_t*)(st + (wv * 4 + idx) * 1024), 16, 0, 0);
        } else {
            constexpr int j = idx - 4;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)p.W + (size_t)(ob[j] + (MODE == MODE_CONV2D ? kb : 0u) + voff_b)), (lds_void_t*)(st + W_A_BYTES + (wv * 5 + j) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing: [k-half] byte addresses inside the slot being READ
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    unsigned ra[2], rb[2];
    {
        const int fr = lane & 15, fq = lane >> 4;
        const unsigned a_row = (unsigned)((wm * WM + fr) * 128), b_row = (unsigned)(W_A_BYTES + (wn * 160 + fr) * 128);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const unsigned sw = (unsigned)(((kh * 4 + fq) ^ (fr & 7)) << 4);
            ra[kh] = lds0 + a_row + sw; rb[kh] = lds0 + b_row + sw;
        }
    }
    unsigned rslot = 0;                                        // 0 / Z_SLOT1: the slot ra / rb point into
    auto flip_read_slot = [&]() {
        const unsigned d = rslot ? (unsigned)(-Z_SLOT1) : (unsigned)Z_SLOT1;
        ra[0] += d; ra[1] += d; rb[0] += d; rb[1] += d;
        rslot = rslot ? 0u : (unsigned)Z_SLOT1;
    };

    // ---- the three streams of a k-tile, as compile-time schedules (Z_LA: how many weight fragments the read stream runs ahead)
#ifndef Z_LA
#define Z_LA 3
#endif
    constexpr int LA = Z_LA, RB = LA + 1;             // B-fragment ring; 20 % RB == 0 keeps the ring index a function of the group
    static_assert(20 % RB == 0 && LA >= 3 && LA <= 4, "ring of 4 or 5 weight fragments");
    float4v acc[2][TM][TN];               // [column half][row tile][column tile]: halves are 80 columns each
    // A fragments: one register set per k-half (the next half's four reads overlap this half's MFMAs), or - convolution modes, whose
    // per-lane gather state needs the registers - ONE set (the next half's reads issue behind the last MFMA group of this one)
    constexpr int NAF = MODE == MODE_DENSE ? 2 : 1;
    half8 af[NAF][TM], bf[RB];            // A fragments; ring of weight fragments

    // ---- kernel prologue: stage 0 of the first tile, complete and visible before the first read
    cursor_begin();
    piece(std::integral_constant<int, 0>{}); piece(std::integral_constant<int, 1>{}); piece(std::integral_constant<int, 2>{});
    piece(std::integral_constant<int, 3>{}); piece(std::integral_constant<int, 4>{}); piece(std::integral_constant<int, 5>{});
    piece(std::integral_constant<int, 6>{}); piece(std::integral_constant<int, 7>{}); piece(std::integral_constant<int, 8>{});
    cursor_end();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // group q of the steady stream (q <= 16): [DMA piece] [the A fragments of k-half 1 at q = 7] the weight fragment LA groups
    // ahead, the counted wait for this group's own fragment, four MFMAs.  Outstanding reads allowed at the wait = the reads
    // issued after the one the group needs: LA (fewer at the tail), + 4 while the A reads of group 7 are among them.
    auto group_piece = [&](auto QC) {
        constexpr int q = decltype(QC)::value;
        constexpr int order[9] = {0, 4, 1, 5, 2, 6, 3, 7, 8};     // A and B pieces alternate
        if constexpr (q < 9) piece(std::integral_constant<int, order[q < 9 ? q : 0]>{});
        if constexpr (q == 8) cursor_end();
    };
#define Z_YOUNGER(q) ((19 - (q) < LA ? 19 - (q) : LA) + ((NAF == 2 && (q) >= 7 && (q) < 7 + LA && (q) != 10) ? 4 : 0))
// (group 10 also needs the A reads of k-half 1: with two register sets they were issued at group 7 and the allowed count is the
// weight reads issued behind them; with ONE set they follow group 9's MFMAs: only group 10's own weight read is behind them.
// Group 0, one set: the A reads follow group 19's MFMAs, with two weight reads behind them.)
#define Z_NWAIT(q) (NAF == 1 ? ((q) == 10 ? 1 : ((q) == 0 ? 2 : Z_YOUNGER(q))) \
                             : ((q) == 10 ? (LA < 4 ? LA : ((10 + LA < 19 ? 10 + LA : 19) - (7 + LA) + 1)) : Z_YOUNGER(q)))
#define Z_GROUP(q) do { \
        group_piece(ZQ(q)); \
        if constexpr (NAF == 2 && (q) == 7) z_read_a<1, NAF>(af, ra); \
        if constexpr ((q) + LA <= 19) z_read_b<((q) + LA <= 19 ? (q) + LA : 0), RB>(bf, rb); \
        z_wait<(q), RB, Z_NWAIT(q), NAF>(af, bf); \
        z_mma<(q), RB, NAF>(acc, af, bf); \
        if constexpr (NAF == 1 && (q) == 9) z_read_a<1, NAF>(af, ra); } while (0)
#define ZQ(q) std::integral_constant<int, q>{}

#ifdef SYN3R_TIMING         // tools/z_timing.py: s_memtime ticks per segment of the k-tile loop, summed per wavefront of one block
    unsigned long long zt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, z_a = __builtin_amdgcn_s_memtime();
#define ZSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); zt[i] += t_ - z_a; z_a = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define ZSTAMP(i)
#endif
    for (unsigned tl = blockIdx.x / 8; tl < t_len; tl += t_stride) {
        int m0, n0, tile_n;
        tile_origin(t_start + tl, m0, n0, tile_n);
        // Convolution modes (round 6): the per-lane gather state of THIS tile is rebuilt here (it was built once already, one k-tile
        // before the previous tile's epilogue, for the cross-tile request of stage 0).  Carried across the epilogue instead, hipcc
        // kept it in scratch for the whole kernel and reloaded it inside the k-loop - scratch_load + s_waitcnt vmcnt(0) in front of
        // the DMA requests, i.e. every k-tile waited for the pieces it had just requested.  ~100 instructions per tile of >= 9 k-tiles.
        if constexpr (MODE != MODE_DENSE) {
            if (c_ks == 1 && c_m0 == m0) {               // (the cursor stands one stage into this tile; exhausted or K = 64: keep what it has)
                conv_tile();
                if constexpr (MODE == MODE_TCONV) conv_tap(0);
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[hh][i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
        // (the fragment registers are outputs of asm reads: redefined per tile, or they all stay live through the epilogue)
#pragma unroll
        for (int i = 0; i < TM; ++i) { asm volatile("" : "=v"(af[0][i])); asm volatile("" : "=v"(af[NAF - 1][i])); }
#pragma unroll
        for (int j = 0; j < RB; ++j) asm volatile("" : "=v"(bf[j]));
        // read prologue of the tile: the first k-half's A fragments and the first LA weight fragments (landed and visible);
        // with one A set in the order the steady stream leaves them in (two weight reads, the A reads, one weight read)
        if constexpr (NAF == 1) {
            z_read_b<0, RB>(bf, rb); z_read_b<1, RB>(bf, rb);
            z_read_a<0, NAF>(af, ra);
            z_read_b<2, RB>(bf, rb);
        } else {
            z_read_a<0, NAF>(af, ra);
            z_read_b<0, RB>(bf, rb); z_read_b<1, RB>(bf, rb); z_read_b<2, RB>(bf, rb);
            if constexpr (LA >= 4) z_read_b<3, RB>(bf, rb);
        }
        // (every per-lane constant of the k-loop is "used" here: where hipcc reloads one of them from scratch behind the epilogue, the
        // reload's s_waitcnt lands in front of the loop - left to the first use INSIDE the loop, the scoreboard's merge of the entry
        // path and the back edge keeps an s_waitcnt vmcnt(0) behind the first DMA request of every k-tile)
        asm volatile("" :: "v"(voff_a), "v"(voff_a2), "v"(voff_b), "v"(ra[0]), "v"(ra[1]), "v"(rb[0]), "v"(rb[1]));
        if constexpr (MODE == MODE_CONV2D) asm volatile("" :: "v"(cv_base[0]), "v"(cv_base[1]), "v"(cv_base[2]), "v"(cv_base[3]), "v"(cv_flags));
        if constexpr (MODE == MODE_TCONV) asm volatile("" :: "v"(cv_off[0]), "v"(cv_off[1]), "v"(cv_off[2]), "v"(cv_off[3]));
        for (int kt = 0; kt < nkt; ++kt) {
            // groups 0-8 carry the nine DMA pieces of the NEXT stage (into the slot the previous barrier released)
            ZSTAMP(5);
            cursor_begin();
            // (convolution modes: where hipcc keeps the gather state in scratch, this use makes it reload ALL of it here - no DMA
            // request of this k-tile is in flight yet, so the reload's s_waitcnt vmcnt(0) waits for the scratch loads alone - instead
            // of piece by piece behind the requests of groups 0-2)
            if constexpr (MODE == MODE_CONV2D) asm volatile("" :: "v"(cv_base[0]), "v"(cv_base[1]), "v"(cv_base[2]), "v"(cv_base[3]), "v"(cv_flags));
            if constexpr (MODE == MODE_TCONV) asm volatile("" :: "v"(cv_off[0]), "v"(cv_off[1]), "v"(cv_off[2]), "v"(cv_off[3]));
            Z_GROUP(0); Z_GROUP(1); Z_GROUP(2); Z_GROUP(3); Z_GROUP(4); Z_GROUP(5); Z_GROUP(6); Z_GROUP(7);
            Z_GROUP(8);
            ZSTAMP(0);
            Z_GROUP(9); Z_GROUP(10); Z_GROUP(11); Z_GROUP(12); Z_GROUP(13); Z_GROUP(14); Z_GROUP(15);
            Z_GROUP(16);
            ZSTAMP(1);
            // group 17: the read stream leaves this k-tile.  Everything this wavefront read from the slot has arrived
            // (lgkmcnt(0): the slot may be refilled), its pieces of the next stage have landed (vmcnt(0)); barrier.
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[17 % RB]), "+v"(bf[18 % RB]), "+v"(bf[19 % RB]));
            z_mma<17, RB, NAF>(acc, af, bf);
            ZSTAMP(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ZSTAMP(3);
            __builtin_amdgcn_s_barrier();
            ZSTAMP(4);
            flip_read_slot();
#ifdef SYN3R_TIMING
            ++zt[7];
#endif
            // the first reads of the next k-tile, while groups 18 and 19 still multiply from registers (not behind a tile's
            // last k-tile: the epilogue wants the registers; the next tile starts with the read prologue above)
            if (kt + 1 < nkt) {
                if constexpr (NAF == 1) {
                    z_read_b<0, RB>(bf, rb);
                    z_mma<18, RB, NAF>(acc, af, bf);
                    z_read_b<1, RB>(bf, rb);
                    z_mma<19, RB, NAF>(acc, af, bf);
                    z_read_a<0, NAF>(af, ra);
                    z_read_b<2, RB>(bf, rb);
                } else {
                    z_read_a<0, NAF>(af, ra);
                    z_read_b<0, RB>(bf, rb);
                    z_mma<18, RB, NAF>(acc, af, bf);
                    z_read_b<1, RB>(bf, rb);
                    z_mma<19, RB, NAF>(acc, af, bf);
                    z_read_b<2, RB>(bf, rb);
                    if constexpr (LA == 4) z_read_b<3, RB>(bf, rb);
                }
            } else {
                z_mma<18, RB, NAF>(acc, af, bf);
                z_mma<19, RB, NAF>(acc, af, bf);
            }
        }
        ZSTAMP(5);
        // ---- epilogue.  Staging = the slot read last (the read cursor has already moved on to the other one) + the gap.
        // Lane-derived indices behind an opaque copy of the lane id: hoisted out of the tile loop they would be carried
        // through the k-loop in registers the accumulators do not leave.
        int le = lane;
        asm volatile("" : "+v"(le));
        char* const epi = smem_raw + (rslot ? 0 : W_STAGE);
        const int gm0 = m0 + wm * WM;
        if (p.geglu_D > 0) {
            // GEGLU.forward: hidden * gelu(gate) on the fp16-rounded projection output; the wavefront's 160 columns are one
            // packed group [80 hidden | 80 gate]
            const int fq = le >> 4;
            const int gn = n0 + wn * 160;
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
                    for (int r = 0; r < 4; r += 2) {
                        const syn3r_f2 hv = (syn3r_f2){(float)(_Float16)(acc[0][i][j][r] + bh[r]), (float)(_Float16)(acc[0][i][j][r + 1] + bh[r + 1])};
                        const syn3r_f2 gv = (syn3r_f2){(float)(_Float16)(acc[1][i][j][r] + bg[r]), (float)(_Float16)(acc[1][i][j][r + 1] + bg[r + 1])};
                        const syn3r_f2 y = hv * gelu_pk(gv);
                        acc[0][i][j][r] = y.x; acc[0][i][j][r + 1] = y.y;
                    }
            }
            const int go0 = tile_n * 160 + wn * WN;
            const bool full = gm0 + WM <= p.M && go0 + WN <= p.geglu_D;
            widep_store(p, acc[0], epi, le, wv, gm0, go0, p.geglu_D, nullptr, nullptr, nullptr, full);
        } else {
            const int gn0 = n0 + wn * 160;
            const bool full = gm0 + WM <= p.M && gn0 + 160 <= p.N;
            widep_store(p, acc[0], epi, le, wv, gm0, gn0, p.N, p.bias, p.residual, p.aux, full);
            widep_store(p, acc[1], epi, le, wv, gm0, gn0 + WN, p.N, p.bias, p.residual, p.aux, full);
        }
        // every wavefront is done with the staging area before the next tile's first DMA piece refills that slot
        __builtin_amdgcn_s_barrier();
        ZSTAMP(6);
    }
#ifdef SYN3R_TIMING
    if (blockIdx.x == gridDim.x / 2 && lane == 0)
        for (int i = 0; i < 8; ++i) g_wide_timing[wv * 8 + i] = zt[i];
#endif
#undef ZSTAMP
#undef ZQ
#undef Z_GROUP
#undef Z_NWAIT
#undef Z_YOUNGER
}
