// Device-wide exclusive scan and stable LSD radix sort of (u64 key, u32 value)
// pairs, hand-written for 64-lane wavefronts.
//
// Role in the hot path: the reference rasteriser (diff-gaussian-rasterization-
// confidence, un-vendored) sorts (tile<<32 | depth-bits) keys with
// cub::DeviceRadixSort::SortPairs; a stable LSD radix sort gives the identical
// order (ties keep duplication order = Gaussian index order).
//
// Structure per BITS-wide pass (templated on the key type, u32 or u64, and on the digit width):
//   k_hist    per-block digit histogram (LDS atomics)        -> hist[digit][block]
//   scan      exclusive scan over hist (digit-major)          -> global bases  (ONE single-block kernel up to
//             16 k counters, the 3-kernel hierarchy above that)
//   k_scatter keys stay in registers; per-wave digit histograms, cross-wave
//             prefix, then per-round wave-level match (BITS ballots) gives each key
//             its stable rank; scatter to base + rank.
// HBM-bound: (key+4) B read + (key+4) B written per pair per pass, + one key read in k_hist.
//
// The rasteriser uses two instantiations (raster_fwd.hip): Gaussians by depth bits (u32 keys, 8-bit digits,
// 4 passes over N elements) and (tile, Gaussian) pairs by tile id (u32 keys, 7-bit digits, 2 passes over P
// pairs, emitted in depth order) - together the same order as one 45-bit sort of (tile << 32 | depth) keys.
#include "common.h"
#include "raster_common.h"

using namespace syn3r;

namespace syn3r {

// ---------------------------------------------------------------- scan (u32, exclusive)
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 4;                      // per thread
constexpr int kScanChunk = kScanThreads * kScanItems;  // 4096 per block

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* smem /*[16+1]*/, unsigned& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) smem[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        unsigned s = lane < (int)(blockDim.x >> 6) ? smem[lane] : 0;
        unsigned si = s;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            unsigned t = __shfl_up(si, o, 64);
            if (lane >= o) si += t;
        }
        if (lane < 16) smem[lane] = si - s;   // exclusive wave offsets
        if (lane == 15) smem[16] = si;        // block total
    }
    __syncthreads();
    unsigned res = incl - v + smem[wv];
    total = smem[16];
    __syncthreads();
    return res;
}

// phase 1: per-chunk sums
__global__ void __launch_bounds__(kScanThreads) k_scan_sums(const unsigned* __restrict__ in,
                                                           const unsigned* __restrict__ perm, size_t n,
                                                           unsigned* __restrict__ sums) {
    __shared__ unsigned smem[17];
    size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanItems;
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += (base + k < n) ? (perm ? in[perm[base + k]] : in[base + k]) : 0;
    unsigned total;
    block_exclusive_scan(s, smem, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// phase 2: one block scans the chunk sums in place (exclusive); writes the grand total to *total_out
__global__ void __launch_bounds__(kScanThreads) k_scan_top(unsigned* __restrict__ sums, int nchunks,
                                                          unsigned* __restrict__ total_out) {
    __shared__ unsigned smem[17];
    unsigned carry = 0;
    for (int start = 0; start < nchunks; start += kScanThreads) {
        int i = start + threadIdx.x;
        unsigned v = i < nchunks ? sums[i] : 0;
        unsigned total;
        unsigned ex = block_exclusive_scan(v, smem, total);
        if (i < nchunks) sums[i] = ex + carry;
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// phase 3: per-chunk exclusive scan + chunk offset
__global__ void __launch_bounds__(kScanThreads) k_scan_final(const unsigned* __restrict__ in,
                                                            const unsigned* __restrict__ perm, size_t n,
                                                            const unsigned* __restrict__ sums,
                                                            unsigned* __restrict__ out) {
    __shared__ unsigned smem[17];
    size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanItems;
    unsigned v[kScanItems];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? (perm ? in[perm[base + k]] : in[base + k]) : 0;
        s += v[k];
    }
    unsigned total;
    unsigned ex = block_exclusive_scan(s, smem, total) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

// n <= kScanThreads * ITEMS (a sort pass's digit table, the rasteriser's binning counters): a single 1024-thread block, ITEMS
// consecutive elements per thread with 16-byte loads and stores - one launch instead of three.
constexpr int kSmallItems = 16, kMidItems = 64;
constexpr size_t kScanSmallMax = (size_t)kScanThreads * kSmallItems, kScanMidMax = (size_t)kScanThreads * kMidItems;
template <int ITEMS>
__global__ void __launch_bounds__(kScanThreads) k_scan_small(const unsigned* __restrict__ in, size_t n,
                                                            unsigned* __restrict__ out,
                                                            unsigned* __restrict__ total_out) {
    __shared__ unsigned smem[17];
    const size_t base = (size_t)threadIdx.x * ITEMS;
    unsigned v[ITEMS];
    unsigned s = 0;
    const bool full = base + ITEMS <= n;
    if (full) {
        const uint4* src = (const uint4*)(in + base);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; ++k) {
            uint4 q = src[k];
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) v[k] = (base + k < n) ? in[base + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) s += v[k];
    unsigned total;
    unsigned ex = block_exclusive_scan(s, smem, total);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        unsigned t = v[k];
        v[k] = ex;
        ex += t;
    }
    if (full) {
        uint4* dst = (uint4*)(out + base);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; ++k) dst[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k)
            if (base + k < n) out[base + k] = v[k];
    }
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

size_t scan_scratch_bytes(size_t n) {
    size_t chunks = (n + kScanChunk - 1) / kScanChunk;
    return ((chunks * 4 + 255) / 256) * 256 + 256;
}

// out[i] = sum(in[0..i-1]); *total_out = sum of all (device pointer, may be null). in may alias out.
int exclusive_scan_u32(const unsigned* in, unsigned* out, size_t n, unsigned* total_out, void* scratch,
                       hipStream_t stream, const unsigned* perm) {
    if (n == 0) {
        if (total_out) return check_hip(hipMemsetAsync(total_out, 0, 4, stream), "memset");
        return SYN3R_OK;
    }
    if (!perm && n <= kScanMidMax && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0) {
        if (n <= kScanSmallMax) SYN3R_LAUNCH(k_scan_small<kSmallItems>, dim3(1), dim3(kScanThreads), 0, stream, in, n, out, total_out);
        else SYN3R_LAUNCH(k_scan_small<kMidItems>, dim3(1), dim3(kScanThreads), 0, stream, in, n, out, total_out);
        return SYN3R_OK;
    }
    size_t chunks = (n + kScanChunk - 1) / kScanChunk;
    if (chunks > (1u << 30)) { set_error("scan: too many elements"); return SYN3R_E_INVALID; }
    unsigned* sums = (unsigned*)scratch;
    SYN3R_LAUNCH(k_scan_sums, dim3((unsigned)chunks), dim3(kScanThreads), 0, stream, in, perm, n, sums);
    SYN3R_LAUNCH(k_scan_top, dim3(1), dim3(kScanThreads), 0, stream, sums, (int)chunks, total_out);
    SYN3R_LAUNCH(k_scan_final, dim3((unsigned)chunks), dim3(kScanThreads), 0, stream, in, perm, n, sums, out);
    return SYN3R_OK;
}

// ---------------------------------------------------------------- radix sort
constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortRounds = 16;                               // keys per thread
constexpr int kSortChunk = kSortThreads * kSortRounds;        // 4096 keys per block
constexpr int kWaveChunk = 64 * kSortRounds;                  // 1024 keys per wave, contiguous

// n_dev != nullptr: the element count lives on the device (clamped to the capacity n)
__device__ __forceinline__ size_t live_count(size_t n, const unsigned* __restrict__ n_dev) {
    if (!n_dev) return n;
    size_t d = *n_dev;
    return d < n ? d : n;
}

template <typename K, int BITS>
__global__ void __launch_bounds__(kSortThreads) k_hist(const K* __restrict__ keys, size_t n_cap,
                                                      const unsigned* __restrict__ n_dev, int shift,
                                                      unsigned* __restrict__ hist, unsigned nblocks, int block_major) {
    constexpr int BINS = 1 << BITS;
    const size_t n = live_count(n_cap, n_dev);
    __shared__ unsigned h[BINS];
    for (int d = threadIdx.x; d < BINS; d += kSortThreads) h[d] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * kSortChunk;
#pragma unroll 4
    for (int r = 0; r < kSortRounds; ++r) {
        size_t i = base + (size_t)r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & (BINS - 1)], 1u);
    }
    __syncthreads();
    // digit-major for the scan; block-major for k_scatter<FUSED>, whose threads (one per digit) then read it coalesced
    for (int d = threadIdx.x; d < BINS; d += kSortThreads) hist[block_major ? (size_t)blockIdx.x * BINS + d : (size_t)d * nblocks + blockIdx.x] = h[d];
}

// vals_in == nullptr: the value of element i is i (first pass of an argsort)
// The block first orders its 4096 elements by digit in LDS (stable: per-wave digit counts -> block-local starts ->
// wave-level match ranks), then writes them out in that order: consecutive threads write consecutive addresses inside a
// digit's run, instead of 64 lanes scattering 4-byte writes over up to 2^BITS destinations (round 1: 1 TB/s on the
// (tile, Gaussian) passes).
// FUSED (small sorts: at most kFusedMaxBlocks blocks, one thread per digit): `bases` is the RAW table of k_hist (block-major) -
// every block forms its own bases from it (a row sum and a prefix per digit, then a block scan over the digits: the table is a
// few hundred KB of L2 reads in total), so a pass needs no scan launch: the depth argsort of 200 000 Gaussians (49 blocks on
// 256 CUs, launch-latency bound) is 8 launches instead of 12.  (Blocks of 1 024 keys instead of 4 096 - four ranking rounds, 196
// blocks - were measured too: 15 instead of 16 us per scatter launch whatever the prologue's unroll; not kept.)  (Also counting the NEXT pass's digits while writing out - integer
// atomics into that pass's table, 5 launches - was built and measured: depth keys share their upper digits, so a pass's atomics
// all land on a few dozen counters and serialise, 1.5 ms per iteration instead of 0.09.)
template <typename K, int BITS, bool FUSED = false>
__global__ void __launch_bounds__(kSortThreads) k_scatter(const K* __restrict__ keys_in,
                                                         const unsigned* __restrict__ vals_in,
                                                         K* __restrict__ keys_out, unsigned* __restrict__ vals_out,
                                                         size_t n_cap, const unsigned* __restrict__ n_dev, int shift,
                                                         const unsigned* __restrict__ bases, unsigned nblocks) {
    constexpr int BINS = 1 << BITS;
    static_assert(!FUSED || BINS == kSortThreads, "the fused form gives every digit a thread");
    const size_t n = live_count(n_cap, n_dev);
    __shared__ unsigned fbase[FUSED ? BINS : 1];   // FUSED: what the scanned table would hold for (digit, this block)
    if constexpr (FUSED) {
        __shared__ unsigned ssm[17];
        const unsigned* col = bases + threadIdx.x;        // block-major table: entry (block b, digit d) at b * BINS + d
        unsigned tot = 0, pre = 0;
        constexpr int UN = 16;                             // loads in flight per thread: the loop is L2 latency, not bandwidth
        for (unsigned b0 = 0; b0 < nblocks; b0 += UN) {
            unsigned c[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) c[u] = b0 + u < nblocks ? col[(size_t)(b0 + u) * BINS] : 0u;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                pre += b0 + u < blockIdx.x ? c[u] : 0u;
                tot += c[u];
            }
        }
        unsigned total;
        fbase[threadIdx.x] = block_exclusive_scan(tot, ssm, total) + pre;
    }
    __shared__ unsigned wh[kSortWaves][BINS];   // per-wave digit counts, then running block-local offsets
    __shared__ unsigned gbase[BINS];            // global position of local position 0 of the digit's run (may wrap: unsigned arithmetic)
    __shared__ K lk[kSortChunk];
    __shared__ unsigned lv[kSortChunk];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kSortWaves * BINS; i += kSortThreads) (&wh[0][0])[i] = 0;
    __syncthreads();

    // wave w owns keys [w*1024, (w+1)*1024) of the block's chunk; round r covers 64 consecutive keys
    const size_t bbase = (size_t)blockIdx.x * kSortChunk;
    const size_t wbase = bbase + (size_t)wv * kWaveChunk;
    K key[kSortRounds];
    unsigned val[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        size_t i = wbase + (size_t)r * 64 + lane;
        bool ok = i < n;
        key[r] = ok ? keys_in[i] : (K)~(K)0;
        val[r] = ok ? (vals_in ? vals_in[i] : (unsigned)i) : 0u;
        if (ok) atomicAdd(&wh[wv][(unsigned)(key[r] >> shift) & (BINS - 1)], 1u);
    }
    __syncthreads();
    // digit d: block total -> exclusive scan over the digits (one wavefront, BINS / 64 digits per lane) -> per-wave
    // starting offsets inside the block, and the global position of the run
    if (wv == 0) {
        constexpr int PER = (BINS + 63) / 64;
        unsigned tot[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int d = lane * PER + k;
            unsigned t = 0;
            if (d < BINS) {
#pragma unroll
                for (int w = 0; w < kSortWaves; ++w) t += wh[w][d];
            }
            tot[k] = t;
            sum += t;
        }
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        unsigned run = incl - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int d = lane * PER + k;
            if (d < BINS) {
                gbase[d] = (FUSED ? fbase[d] : bases[(size_t)d * nblocks + blockIdx.x]) - run;
                unsigned off = run;
#pragma unroll
                for (int w = 0; w < kSortWaves; ++w) {
                    unsigned c = wh[w][d];
                    wh[w][d] = off;
                    off += c;
                }
                run += tot[k];
            }
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        size_t i = wbase + (size_t)r * 64 + lane;
        bool ok = i < n;
        unsigned d = (unsigned)(key[r] >> shift) & (BINS - 1);
        // lanes of this wave with the same digit (invalid lanes form their own group)
        unsigned long long peers = __ballot(ok);
        if (!ok) peers = ~peers;
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        unsigned rank = __popcll(peers & lt);
        unsigned cnt = __popcll(peers);
        int leader = __ffsll((long long)peers) - 1;
        unsigned base = 0;
        if (ok && lane == leader) {
            base = wh[wv][d];
            wh[wv][d] = base + cnt;
        }
        base = __shfl(base, leader, 64);
        if (ok) {
            lk[base + rank] = key[r];
            lv[base + rank] = val[r];
        }
        // a wave only ever touches its own wh[wv][*] row from here on: wave-level ordering is enough
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const size_t left = n > bbase ? n - bbase : 0;
    const unsigned cnt_block = left < (size_t)kSortChunk ? (unsigned)left : (unsigned)kSortChunk;
#pragma unroll 4
    for (unsigned i = threadIdx.x; i < cnt_block; i += kSortThreads) {
        const K k = lk[i];
        const unsigned pos = gbase[(unsigned)(k >> shift) & (BINS - 1)] + i;
        keys_out[pos] = k;
        vals_out[pos] = lv[i];
    }
}

constexpr int kMaxBins = 256;   // widest digit instantiated below

constexpr size_t kFusedMaxBlocks = 128;   // fused small sort: up to 524 288 elements

size_t sort_scratch_bytes(size_t n) {
    size_t nblocks = (n + kSortChunk - 1) / kSortChunk;
    if (nblocks == 0) nblocks = 1;
    size_t hist = (((size_t)kMaxBins * nblocks * 4 + 255) / 256) * 256;
    return hist + scan_scratch_bytes((size_t)kMaxBins * nblocks);
}

// Sorts bits [0, nbits) of the keys, BITS at a time (nbits is rounded up to a multiple of BITS).
// Ping-pongs between (keys_a, vals_a) and (keys_b, vals_b); returns which buffer holds the result.
// iota_vals: vals_a is not read, element i carries the value i (argsort).
template <typename K, int BITS>
int radix_sort_t(K* keys_a, unsigned* vals_a, K* keys_b, unsigned* vals_b, size_t n, int nbits, void* scratch,
                 hipStream_t stream, int* result_in_b, const unsigned* n_dev, bool iota_vals) {
    constexpr int BINS = 1 << BITS;
    *result_in_b = 0;
    if (n == 0) return SYN3R_OK;
    size_t nblocks = (n + kSortChunk - 1) / kSortChunk;
    if (nblocks > (1u << 22)) { set_error("sort: too many pairs (%zu)", n); return SYN3R_E_INVALID; }
    unsigned* hist = (unsigned*)scratch;
    size_t hist_bytes = (((size_t)kMaxBins * nblocks * 4 + 255) / 256) * 256;
    void* scan_scratch = (char*)scratch + hist_bytes;
    int passes = (nbits + BITS - 1) / BITS;
    K* kin = keys_a; unsigned* vin = vals_a;
    K* kout = keys_b; unsigned* vout = vals_b;
    bool fused = false;                      // small sort: no scan launch, k_scatter<FUSED> forms its bases from the raw table
    if constexpr (BINS == kSortThreads) fused = nblocks <= kFusedMaxBlocks;
    for (int p = 0; p < passes; ++p) {
        int shift = BITS * p;
        SYN3R_LAUNCH_NAMED("k_hist", (k_hist<K, BITS>), dim3((unsigned)nblocks), dim3(kSortThreads), 0, stream, kin, n,
                           n_dev, shift, hist, (unsigned)nblocks, fused ? 1 : 0);
        if (fused) {
            if constexpr (BINS == kSortThreads)
                SYN3R_LAUNCH_NAMED("k_scatter", (k_scatter<K, BITS, true>), dim3((unsigned)nblocks), dim3(kSortThreads), 0, stream,
                                   (const K*)kin, (const unsigned*)((p == 0 && iota_vals) ? nullptr : vin), kout, vout, n,
                                   n_dev, shift, (const unsigned*)hist, (unsigned)nblocks);
        } else {
            int rc = exclusive_scan_u32(hist, hist, (size_t)BINS * nblocks, nullptr, scan_scratch, stream);
            if (rc) return rc;
            SYN3R_LAUNCH_NAMED("k_scatter", (k_scatter<K, BITS>), dim3((unsigned)nblocks), dim3(kSortThreads), 0, stream,
                               (const K*)kin, (const unsigned*)((p == 0 && iota_vals) ? nullptr : vin), kout, vout, n,
                               n_dev, shift, (const unsigned*)hist, (unsigned)nblocks);
        }
        K* tk = kin; kin = kout; kout = tk;
        unsigned* tv = vin; vin = vout; vout = tv;
    }
    *result_in_b = (passes & 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return check_hip(e, "radix sort launch");
    return SYN3R_OK;
}

int radix_sort_pairs(unsigned long long* keys_a, unsigned* vals_a, unsigned long long* keys_b, unsigned* vals_b,
                     size_t n, int nbits, void* scratch, hipStream_t stream, int* result_in_b, const unsigned* n_dev) {
    return radix_sort_t<unsigned long long, 8>(keys_a, vals_a, keys_b, vals_b, n, nbits, scratch, stream, result_in_b,
                                               n_dev, false);
}

int argsort_depth_u32(unsigned* keys_a, unsigned* vals_a, unsigned* keys_b, unsigned* vals_b, size_t n, void* scratch,
                      hipStream_t stream, int* result_in_b) {
    return radix_sort_t<unsigned, 8>(keys_a, vals_a, keys_b, vals_b, n, 32, scratch, stream, result_in_b, nullptr, true);
}

int sort_pairs_by_tile_u32(unsigned* keys_a, unsigned* vals_a, unsigned* keys_b, unsigned* vals_b, size_t n, int nbits,
                           void* scratch, hipStream_t stream, int* result_in_b, const unsigned* n_dev) {
    return radix_sort_t<unsigned, 7>(keys_a, vals_a, keys_b, vals_b, n, nbits, scratch, stream, result_in_b, n_dev,
                                     false);
}

}  // namespace syn3r

// ---------------------------------------------------------------- C-ABI (exposed for tests and for callers that bin their own keys)
extern "C" size_t syn3r_sort_pairs_workspace_bytes(long long n) {
    if (n < 0) return 0;
    return sort_scratch_bytes((size_t)n) + 256;
}

extern "C" int syn3r_sort_pairs(unsigned long long* keys, unsigned* vals, unsigned long long* keys_tmp,
                                unsigned* vals_tmp, long long n, int nbits, void* workspace, size_t workspace_bytes,
                                int* result_in_tmp, void* stream) {
    SYN3R_REQUIRE(n >= 0 && nbits >= 1 && nbits <= 64, "sort_pairs: bad n=%lld nbits=%d", n, nbits);
    SYN3R_REQUIRE(result_in_tmp, "sort_pairs: result_in_tmp required");
    if (n == 0) { *result_in_tmp = 0; return SYN3R_OK; }
    SYN3R_REQUIRE(keys && vals && keys_tmp && vals_tmp, "sort_pairs: null buffer");
    if (!workspace || workspace_bytes < syn3r_sort_pairs_workspace_bytes(n)) {
        set_error("sort_pairs: workspace %zu < %zu", workspace_bytes, syn3r_sort_pairs_workspace_bytes(n));
        return SYN3R_E_WORKSPACE;
    }
    return radix_sort_pairs(keys, vals, keys_tmp, vals_tmp, (size_t)n, nbits, workspace, (hipStream_t)stream,
                            result_in_tmp, nullptr);
}
