// Mean squared distance of every point to its 3 nearest neighbours.
//
// The reference initialises the Gaussian scales from this quantity inside FSGS' GaussianModel.create_from_pcd
// (`distCUDA2` of the `simple-knn` CUDA extension; reached from gsTrainer construction and from
// reset_gaussians_from_pcd, call site model/diffusionGS.py:1685-1687).  The extension is NOT in /root/reference
// (un-vendored submodule): what it returns is well defined - for point i the mean of the three smallest squared
// Euclidean distances to OTHER points of the cloud - and any exact search returns the same three distances, so this
// file restates the published approach (points ordered along a Morton curve, boxes of consecutive points pruned by
// their bounding boxes) and is checked against brute force (bit-exact, same fp32 operation order) and a k-d tree.
//
// gfx950 mapping: a lane owns one point, lanes of a wavefront hold 64 CONSECUTIVE points of the Morton order, so
// they agree on which boxes survive the pruning test (one ballot per box, no divergent scans) and the scanned
// points are wave-uniform addresses (one 16-byte request per point for the whole wavefront, served by the scalar /
// L1 path).  Work: N * (boxes tested + 1024 * boxes scanned); with a Morton-local cloud 2-10 boxes survive.
// Compiled with -ffp-contract=off (syn3r_amd/build.py STRICT_FP): d2 = (dx*dx + dy*dy) + dz*dz, no fused multiply-add.
#include "common.h"
#include "raster_common.h"

using namespace syn3r;

namespace {

constexpr int kBox = 1024;           // consecutive Morton-ordered points per bounding box
constexpr int kThreads = 256;

struct Bounds { float lo[3], hi[3]; };

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// AABB of points [first, first + count) with stride `per` points per block: out[blockIdx.x]
__global__ void __launch_bounds__(kThreads) k_aabb(const float* __restrict__ pts, int stride_floats, int n, int per,
                                                  Bounds* __restrict__ out) {
    const int first = blockIdx.x * per, last = min(n, first + per);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = first + threadIdx.x; i < last; i += kThreads) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[(size_t)i * stride_floats + c];
            lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v);
        }
    }
    __shared__ float s[kThreads / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min(lo[c]); hi[c] = wave_max(hi[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = lo[c]; s[threadIdx.x >> 6][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        Bounds b;
        for (int c = 0; c < 3; ++c) {
            b.lo[c] = fminf(fminf(s[0][c], s[1][c]), fminf(s[2][c], s[3][c]));
            b.hi[c] = fmaxf(fmaxf(s[0][3 + c], s[1][3 + c]), fmaxf(s[2][3 + c], s[3][3 + c]));
        }
        out[blockIdx.x] = b;
    }
}

// one block: AABB of the per-block AABBs
__global__ void __launch_bounds__(kThreads) k_aabb_final(const Bounds* __restrict__ part, int nparts, Bounds* __restrict__ out) {
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < nparts; i += kThreads)
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], part[i].lo[c]); hi[c] = fmaxf(hi[c], part[i].hi[c]); }
    __shared__ float s[kThreads / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min(lo[c]); hi[c] = wave_max(hi[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = lo[c]; s[threadIdx.x >> 6][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        Bounds b;
        for (int c = 0; c < 3; ++c) {
            b.lo[c] = fminf(fminf(s[0][c], s[1][c]), fminf(s[2][c], s[3][c]));
            b.hi[c] = fmaxf(fmaxf(s[0][3 + c], s[1][3 + c]), fmaxf(s[2][3 + c], s[3][3 + c]));
        }
        *out = b;
    }
}

__device__ __forceinline__ unsigned spread10(unsigned v) {     // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// 30-bit Morton code of the point's cell in a 1024^3 grid over the cloud's bounding box (the ORDER only steers the
// pruning; the distances found do not depend on it)
__global__ void __launch_bounds__(kThreads) k_morton(const float* __restrict__ pts, int n, const Bounds* __restrict__ bb,
                                                    unsigned* __restrict__ codes) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    unsigned q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ext = bb->hi[c] - bb->lo[c];
        float t = ext > 0.0f ? (pts[(size_t)i * 3 + c] - bb->lo[c]) / ext : 0.0f;
        t = fminf(fmaxf(t * 1023.0f, 0.0f), 1023.0f);
        q[c] = (unsigned)t;
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
}

__global__ void __launch_bounds__(kThreads) k_gather(const float* __restrict__ pts, const unsigned* __restrict__ order, int n,
                                                    float4* __restrict__ sorted) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const unsigned s = order[i];
    sorted[i] = make_float4(pts[(size_t)s * 3], pts[(size_t)s * 3 + 1], pts[(size_t)s * 3 + 2], 0.0f);
}

__device__ __forceinline__ void keep3(float d, float& b0, float& b1, float& b2) {     // b0 <= b1 <= b2
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
        } else b2 = d;
    }
}

__device__ __forceinline__ float dist2(const float4& p, const float4& q) {
    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ float box_dist2(const Bounds& b, const float4& p) {
    const float dx = fmaxf(fmaxf(b.lo[0] - p.x, p.x - b.hi[0]), 0.0f);
    const float dy = fmaxf(fmaxf(b.lo[1] - p.y, p.y - b.hi[1]), 0.0f);
    const float dz = fmaxf(fmaxf(b.lo[2] - p.z, p.z - b.hi[2]), 0.0f);
    return (dx * dx + dy * dy) + dz * dz;
}

__global__ void __launch_bounds__(kThreads) k_knn3(const float4* __restrict__ sorted, const unsigned* __restrict__ order,
                                                  const Bounds* __restrict__ boxes, int n, int nboxes, float* __restrict__ out) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const bool live = i < n;
    const float4 p = sorted[live ? i : n - 1];
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;
    // the point's own box first: after it the third-best distance is tight and most other boxes fail the test
    const int own = (live ? i : n - 1) / kBox;
    auto scan = [&](int b) {
        const int first = b * kBox, last = min(n, first + kBox);
        for (int j = first; j < last; ++j) {
            const float d = dist2(p, sorted[j]);          // wave-uniform address
            if (j != i) keep3(d, b0, b1, b2);
        }
    };
    // Boxes are visited by a whole wavefront together: its 64 points are consecutive on the curve and inside ONE box
    // (64 divides 1024), so `own` is wave-uniform; another box is scanned if ANY lane still needs it - lanes that do
    // not only spend comparisons that change nothing.
    const int own_u = __builtin_amdgcn_readfirstlane(own);
    scan(own_u);
    for (int b = 0; b < nboxes; ++b) {
        if (b == own_u) continue;
        const float bd = box_dist2(boxes[b], p);
        // a box whose nearest face is farther than the current third-best cannot change the result; the slack keeps
        // every box that the rounded arithmetic could place just inside
        const bool need = live && bd <= b2 * 1.0001f;
        if (__ballot(need) == 0ull) continue;
        scan(b);
    }
    if (live) out[order[i]] = ((b0 + b1) + b2) / 3.0f;
}

// ----------------------------------------------------------------------------------------------------------------
// Statistical outlier removal of a point cloud: what the reference runs on the dust3r cloud,
// `down_pcd.remove_statistical_outlier(nb_neighbors=20, std_ratio=3.0)` (model/diffusionGS.py:321, open3d 0.17.0 - not in
// /root/reference).  Published algorithm (Open3D `PointCloud::RemoveStatisticalOutliers`), restated:
//   avg_i  = mean of the Euclidean distances to the nb_neighbors nearest points of the cloud, THE POINT ITSELF INCLUDED
//            (its k-d tree query returns the query point at distance 0), summed nearest first, in float64;
//   mean, std (n - 1 in the denominator) of avg over the cloud;   keep_i = 0 < avg_i < mean + std_ratio * std.
// Same search structure as the 3-NN above (Morton order, 1024-point boxes, wave-uniform scans), in float64 throughout:
// open3d holds its points as doubles, and the keep decision thresholds a float64 statistic.  A lane keeps its K best
// squared distances sorted in registers (K = 20: 40 VGPRs).
struct BoundsD { double lo[3], hi[3]; };

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ void __launch_bounds__(kThreads) k_aabb64(const double* __restrict__ pts, int stride, int n, int per,
                                                    BoundsD* __restrict__ out) {
    const int first = blockIdx.x * per, last = min(n, first + per);
    double lo[3] = {1.0e300, 1.0e300, 1.0e300}, hi[3] = {-1.0e300, -1.0e300, -1.0e300};
    for (int i = first + threadIdx.x; i < last; i += kThreads) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = pts[(size_t)i * stride + c];
            lo[c] = fmin(lo[c], v); hi[c] = fmax(hi[c], v);
        }
    }
    __shared__ double s[kThreads / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min_d(lo[c]); hi[c] = wave_max_d(hi[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = lo[c]; s[threadIdx.x >> 6][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        BoundsD b;
        for (int c = 0; c < 3; ++c) {
            b.lo[c] = fmin(fmin(s[0][c], s[1][c]), fmin(s[2][c], s[3][c]));
            b.hi[c] = fmax(fmax(s[0][3 + c], s[1][3 + c]), fmax(s[2][3 + c], s[3][3 + c]));
        }
        out[blockIdx.x] = b;
    }
}

__global__ void __launch_bounds__(kThreads) k_aabb64_final(const BoundsD* __restrict__ part, int nparts, BoundsD* __restrict__ out) {
    double lo[3] = {1.0e300, 1.0e300, 1.0e300}, hi[3] = {-1.0e300, -1.0e300, -1.0e300};
    for (int i = threadIdx.x; i < nparts; i += kThreads)
        for (int c = 0; c < 3; ++c) { lo[c] = fmin(lo[c], part[i].lo[c]); hi[c] = fmax(hi[c], part[i].hi[c]); }
    __shared__ double s[kThreads / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min_d(lo[c]); hi[c] = wave_max_d(hi[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { s[threadIdx.x >> 6][c] = lo[c]; s[threadIdx.x >> 6][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        BoundsD b;
        for (int c = 0; c < 3; ++c) {
            b.lo[c] = fmin(fmin(s[0][c], s[1][c]), fmin(s[2][c], s[3][c]));
            b.hi[c] = fmax(fmax(s[0][3 + c], s[1][3 + c]), fmax(s[2][3 + c], s[3][3 + c]));
        }
        *out = b;
    }
}

__global__ void __launch_bounds__(kThreads) k_morton64(const double* __restrict__ pts, int n, const BoundsD* __restrict__ bb,
                                                      unsigned* __restrict__ codes) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    unsigned q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double ext = bb->hi[c] - bb->lo[c];
        double t = ext > 0.0 ? (pts[(size_t)i * 3 + c] - bb->lo[c]) / ext : 0.0;
        t = fmin(fmax(t * 1023.0, 0.0), 1023.0);
        q[c] = (unsigned)t;
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
}

__global__ void __launch_bounds__(kThreads) k_gather64(const double* __restrict__ pts, const unsigned* __restrict__ order, int n,
                                                      double4* __restrict__ sorted) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const unsigned s = order[i];
    sorted[i] = make_double4(pts[(size_t)s * 3], pts[(size_t)s * 3 + 1], pts[(size_t)s * 3 + 2], 0.0);
}

__device__ __forceinline__ double dist2d(const double4& p, const double4& q) {
    const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    return (dx * dx + dy * dy) + dz * dz;                 // the k-d tree's accumulation order for 3 dimensions
}

__device__ __forceinline__ double box_dist2d(const BoundsD& b, const double4& p) {
    const double dx = fmax(fmax(b.lo[0] - p.x, p.x - b.hi[0]), 0.0);
    const double dy = fmax(fmax(b.lo[1] - p.y, p.y - b.hi[1]), 0.0);
    const double dz = fmax(fmax(b.lo[2] - p.z, p.z - b.hi[2]), 0.0);
    return (dx * dx + dy * dy) + dz * dz;
}

template <int K>
__global__ void __launch_bounds__(kThreads) k_knn_mean64(const double4* __restrict__ sorted, const unsigned* __restrict__ order,
                                                        const BoundsD* __restrict__ boxes, int n, int nboxes,
                                                        double* __restrict__ out) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const bool live = i < n;
    const double4 p = sorted[live ? i : n - 1];
    double best[K];                                        // ascending
#pragma unroll
    for (int t = 0; t < K; ++t) best[t] = 1.0e300;
    auto scan = [&](int b) {
        const int first = b * kBox, last = min(n, first + kBox);
        for (int j = first; j < last; ++j) {
            const double d = dist2d(p, sorted[j]);         // wave-uniform address; the point itself is a candidate (d = 0)
            if (d < best[K - 1]) {
#pragma unroll
                for (int t = K - 1; t > 0; --t) best[t] = d < best[t - 1] ? best[t - 1] : fmin(best[t], d);
                best[0] = fmin(best[0], d);
            }
        }
    };
    const int own_u = __builtin_amdgcn_readfirstlane((live ? i : n - 1) / kBox);
    scan(own_u);
    for (int b = 0; b < nboxes; ++b) {
        if (b == own_u) continue;
        const double bd = box_dist2d(boxes[b], p);
        const bool need = live && bd <= best[K - 1] * 1.000001;
        if (__ballot(need) == 0ull) continue;
        scan(b);
    }
    if (live) {
        const int cnt = n < K ? n : K;                     // a cloud smaller than K: the mean runs over what the query returns
        double sum = 0.0;
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < cnt) sum += sqrt(best[t]);
        out[order[i]] = sum / (double)cnt;
    }
}

// ONE block: mean and standard deviation as open3d's PointCloud::RemoveStatisticalOutliers takes them, threshold = mean +
// ratio * std.  The SUMS run over the positive entries only, the DIVISORS are `valid_distances` = the number of points whose
// neighbour query returned anything - every point here (a query returns at least the point itself) - and that count - 1:
// a cluster of more than nb_neighbors coincident points (average distance 0) lowers the mean instead of dropping out of it.
// Thread-strided partial sums + a fixed tree: the result does not depend on scheduling.  stats = {mean, std, threshold, n}
__global__ void __launch_bounds__(1024) k_outlier_stats(const double* __restrict__ avg, int n, double ratio, double* __restrict__ stats) {
    __shared__ double red[1024];
    __shared__ double mean_s;
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double v = avg[i];
        if (v > 0.0) { s += v; c += 1.0; }
    }
    auto reduce = [&](double v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    const double total = reduce(s);
    (void)c;
    const double valid = (double)n;
    if (threadIdx.x == 0) mean_s = total / valid;
    __syncthreads();
    const double mean = mean_s;
    double q = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double v = avg[i];
        if (v > 0.0) q += (v - mean) * (v - mean);
    }
    const double sq = reduce(q);
    if (threadIdx.x == 0) {
        const double sd = sqrt(sq / (valid - 1.0));
        stats[0] = mean; stats[1] = sd; stats[2] = mean + ratio * sd; stats[3] = valid;
    }
}

__global__ void __launch_bounds__(kThreads) k_outlier_keep(const double* __restrict__ avg, int n, const double* __restrict__ stats,
                                                          unsigned char* __restrict__ keep) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const double v = avg[i];
    keep[i] = (v > 0.0 && v < stats[2]) ? 1 : 0;
}

}  // namespace

extern "C" size_t syn3r_pcd_outlier_workspace_bytes(int n) {
    if (!SYN3R_DIM_OK(n)) return 0;
    const size_t nn = (size_t)n;
    const size_t nbox = (nn + kBox - 1) / kBox;
    return align256(nn * 4) * 4 + align256(nn * 32) + align256((nbox + 1) * sizeof(BoundsD)) * 2 + sort_scratch_bytes(nn) + 256;
}

extern "C" int syn3r_pcd_statistical_outlier(const double* points, int n, int nb_neighbors, double std_ratio, double* avg_dist,
                                             unsigned char* keep, double* stats, void* ws, size_t ws_bytes, void* stream_) {
    SYN3R_REQUIRE(points && avg_dist && keep && stats && ws, "pcd_statistical_outlier: null pointer");
    SYN3R_REQUIRE(n >= 2 && n <= SYN3R_DIM_MAX, "pcd_statistical_outlier: needs 2 .. %d points, got %d", SYN3R_DIM_MAX, n);
    SYN3R_REQUIRE(nb_neighbors == 20, "pcd_statistical_outlier: built for nb_neighbors = 20 (model/diffusionGS.py:321), got %d",
                  nb_neighbors);
    SYN3R_REQUIRE(std_ratio > 0.0, "pcd_statistical_outlier: std_ratio must be positive");
    SYN3R_REQUIRE(ws_bytes >= syn3r_pcd_outlier_workspace_bytes(n), "pcd_statistical_outlier: workspace too small (%zu < %zu)",
                  ws_bytes, syn3r_pcd_outlier_workspace_bytes(n));
    SYN3R_REQUIRE(((uintptr_t)ws & 255) == 0, "pcd_statistical_outlier: workspace must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t nn = (size_t)n;
    const int nbox = (int)((nn + kBox - 1) / kBox);
    char* w = (char*)ws;
    unsigned* codes_a = (unsigned*)w; w += align256(nn * 4);
    unsigned* codes_b = (unsigned*)w; w += align256(nn * 4);
    unsigned* order_a = (unsigned*)w; w += align256(nn * 4);
    unsigned* order_b = (unsigned*)w; w += align256(nn * 4);
    double4* sorted = (double4*)w; w += align256(nn * 32);
    BoundsD* part = (BoundsD*)w; w += align256((size_t)(nbox + 1) * sizeof(BoundsD));
    BoundsD* boxes = (BoundsD*)w; w += align256((size_t)(nbox + 1) * sizeof(BoundsD));
    void* sort_ws = w;
    SYN3R_LAUNCH(k_aabb64, dim3(nbox), dim3(kThreads), 0, stream, points, 3, n, kBox, part);
    SYN3R_LAUNCH(k_aabb64_final, dim3(1), dim3(kThreads), 0, stream, part, nbox, boxes + nbox);
    const int blocks = (n + kThreads - 1) / kThreads;
    SYN3R_LAUNCH(k_morton64, dim3(blocks), dim3(kThreads), 0, stream, points, n, boxes + nbox, codes_a);
    int in_b = 0;
    int rc = argsort_depth_u32(codes_a, order_a, codes_b, order_b, nn, sort_ws, stream, &in_b);
    if (rc != SYN3R_OK) return rc;
    const unsigned* order = in_b ? order_b : order_a;
    SYN3R_LAUNCH(k_gather64, dim3(blocks), dim3(kThreads), 0, stream, points, order, n, sorted);
    SYN3R_LAUNCH(k_aabb64, dim3(nbox), dim3(kThreads), 0, stream, (const double*)sorted, 4, n, kBox, boxes);
    SYN3R_LAUNCH(k_knn_mean64<20>, dim3(blocks), dim3(kThreads), 0, stream, sorted, order, boxes, n, nbox, avg_dist);
    SYN3R_LAUNCH(k_outlier_stats, dim3(1), dim3(1024), 0, stream, avg_dist, n, std_ratio, stats);
    SYN3R_LAUNCH(k_outlier_keep, dim3(blocks), dim3(kThreads), 0, stream, avg_dist, n, stats, keep);
    SYN3R_LAUNCH_CHECK("pcd_statistical_outlier");
    return SYN3R_OK;
}

extern "C" size_t syn3r_knn3_workspace_bytes(int n) {
    if (!SYN3R_DIM_OK(n)) return 0;
    const size_t nn = (size_t)n;
    const size_t nbox = (nn + kBox - 1) / kBox;
    return align256(nn * 4) * 4                      // Morton codes / order, ping and pong
           + align256(nn * 16)                       // points in Morton order
           + align256((nbox + 1) * sizeof(Bounds)) * 2   // per-box bounds (input order, then sorted order) + the cloud's
           + sort_scratch_bytes(nn) + 256;
}

extern "C" int syn3r_knn3_mean_dist2(const float* points, int n, float* out, void* ws, size_t ws_bytes, void* stream_) {
    SYN3R_REQUIRE(points && out && ws, "knn3: null pointer");
    SYN3R_REQUIRE(n >= 4 && n <= SYN3R_DIM_MAX, "knn3: needs 4 .. %d points (3 neighbours), got %d", SYN3R_DIM_MAX, n);
    SYN3R_REQUIRE(ws_bytes >= syn3r_knn3_workspace_bytes(n), "knn3: workspace too small (%zu < %zu)", ws_bytes,
                  syn3r_knn3_workspace_bytes(n));
    SYN3R_REQUIRE(((uintptr_t)ws & 255) == 0, "knn3: workspace must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t nn = (size_t)n;
    const int nbox = (int)((nn + kBox - 1) / kBox);
    char* w = (char*)ws;
    unsigned* codes_a = (unsigned*)w; w += align256(nn * 4);
    unsigned* codes_b = (unsigned*)w; w += align256(nn * 4);
    unsigned* order_a = (unsigned*)w; w += align256(nn * 4);
    unsigned* order_b = (unsigned*)w; w += align256(nn * 4);
    float4* sorted = (float4*)w; w += align256(nn * 16);
    Bounds* part = (Bounds*)w; w += align256((size_t)(nbox + 1) * sizeof(Bounds));
    Bounds* boxes = (Bounds*)w; w += align256((size_t)(nbox + 1) * sizeof(Bounds));
    void* sort_ws = w;

    // 1. bounding box of the cloud (two levels, fixed order)
    SYN3R_LAUNCH(k_aabb, dim3(nbox), dim3(kThreads), 0, stream, points, 3, n, kBox, part);
    SYN3R_LAUNCH(k_aabb_final, dim3(1), dim3(kThreads), 0, stream, part, nbox, boxes + nbox);
    // 2. Morton order (stable argsort of the 30-bit codes)
    const int blocks = (n + kThreads - 1) / kThreads;
    SYN3R_LAUNCH(k_morton, dim3(blocks), dim3(kThreads), 0, stream, points, n, boxes + nbox, codes_a);
    int in_b = 0;
    int rc = argsort_depth_u32(codes_a, order_a, codes_b, order_b, nn, sort_ws, stream, &in_b);
    if (rc != SYN3R_OK) return rc;
    const unsigned* order = in_b ? order_b : order_a;
    // 3. points in that order, bounds of every 1024 of them
    SYN3R_LAUNCH(k_gather, dim3(blocks), dim3(kThreads), 0, stream, points, order, n, sorted);
    SYN3R_LAUNCH(k_aabb, dim3(nbox), dim3(kThreads), 0, stream, (const float*)sorted, 4, n, kBox, boxes);
    // 4. search
    SYN3R_LAUNCH(k_knn3, dim3(blocks), dim3(kThreads), 0, stream, sorted, order, boxes, n, nbox, out);
    SYN3R_LAUNCH_CHECK("knn3");
    return SYN3R_OK;
}
