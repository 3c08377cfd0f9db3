// Gaussian rasteriser, forward: projection (EWA), tile binning, depth sort, front-to-back
// alpha blend with colour / depth / alpha outputs and a per-Gaussian confidence factor.
//
// Replaces the reference's `gsTrainer.render_view(cam)` hot kernels (call sites
// model/diffusionGS.py:154,166); the CUDA source is an un-vendored submodule
// (SURVEY.md §8c), so this follows the published 3DGS algorithm — see
// raster_common.h.  PARITY UNPINNED against the CUDA build; pinned against
// oracle/raster_oracle.py.
//
// MI355X mapping: one 16x16 tile = one 128-thread workgroup = 2 wavefronts, a wavefront covers a 16 x 8 half with two
// pixels per lane on packed fp32 arithmetic (k_render below).  Sorted splats are staged through LDS in batches as 48-byte
// records (3 x 16-byte loads per lane, broadcast reads in the blend loop).  The per-tile lists come from a hierarchical
// filter of the depth-ordered Gaussians, not from a pair sort, and the blend kernels take the tiles longest list first
// ("Hierarchical binning" below).
#include "common.h"
#include "raster_common.h"

using namespace syn3r;

namespace syn3r {

size_t geom_bytes(int N) {
    size_t n = (size_t)N;
    size_t b = 256;                         // header
    b += align256(n * 4);                   // depths
    b += align256(n * 8);                   // means2D
    b += align256(n * 24);                  // cov3D
    b += align256(n * 16);                  // conic_opacity
    b += align256(n * 12);                  // rgb
    b += align256(n * 4);                   // clamped
    b += align256(n * 4);                   // tiles_touched
    b += align256(n * 4);                   // point_offsets
    b += align256(n * sizeof(Splat));       // splats
    b += 4 * align256(n * 4);               // depth keys / order, ping + pong
    b += align256(sort_scratch_bytes(n));   // argsort histograms
    b += scan_scratch_bytes(n);
    return b;
}

GeomState carve_geom(void* buf, int N) {
    size_t n = (size_t)N;
    char* p = (char*)buf;
    GeomState g;
    g.header = (unsigned*)p; p += 256;
    g.depths = (float*)p; p += align256(n * 4);
    g.means2D = (float*)p; p += align256(n * 8);
    g.cov3D = (float*)p; p += align256(n * 24);
    g.conic_opacity = (float*)p; p += align256(n * 16);
    g.rgb = (float*)p; p += align256(n * 12);
    g.clamped = (unsigned*)p; p += align256(n * 4);
    g.tiles_touched = (unsigned*)p; p += align256(n * 4);
    g.point_offsets = (unsigned*)p; p += align256(n * 4);
    g.splats = (Splat*)p; p += align256(n * sizeof(Splat));
    g.dkeys_a = (unsigned*)p; p += align256(n * 4);
    g.dkeys_b = (unsigned*)p; p += align256(n * 4);
    g.order_a = (unsigned*)p; p += align256(n * 4);
    g.order_b = (unsigned*)p; p += align256(n * 4);
    g.order = g.order_a;                    // 32 bits in 8-bit digits = 4 passes: the result is back in the ping buffer
    g.sort_scratch = p; p += align256(sort_scratch_bytes(n));
    g.scan_scratch = p;
    return g;
}

size_t image_bytes(int H, int W) {
    size_t tiles = (size_t)((W + kTileX - 1) / kTileX) * ((H + kTileY - 1) / kTileY);
    size_t px = (size_t)H * W;
    return align256(tiles * 8) + align256(px * 4) + align256(px * 4) + align256(tiles * 16) + align256((kBinCounters + kMaxSuperSlots) * 4) + align256(tiles * 4);
}

ImageState carve_image(void* buf, int H, int W) {
    size_t tiles = (size_t)((W + kTileX - 1) / kTileX) * ((H + kTileY - 1) / kTileY);
    size_t px = (size_t)H * W;
    char* p = (char*)buf;
    ImageState s;
    s.ranges = (uint2*)p; p += align256(tiles * 8);
    s.n_contrib = (unsigned*)p; p += align256(px * 4);
    s.final_T = (float*)p; p += align256(px * 4);
    s.tile_counts = (unsigned*)p; p += align256(tiles * 16);
    s.bin_counters = (unsigned*)p; p += align256((kBinCounters + kMaxSuperSlots) * 4);
    s.tile_order = (unsigned*)p;
    return s;
}

size_t binning_bytes(long long P) {
    size_t n = (size_t)(P > 0 ? P : 1);
    return 2 * align256(n * 8) + 2 * align256(n * 4) + sort_scratch_bytes(n) + 256;
}

BinningState carve_binning(void* buf, long long P) {
    size_t n = (size_t)(P > 0 ? P : 1);
    char* p = (char*)buf;
    BinningState b;
    b.keys_a = (unsigned long long*)p; p += align256(n * 8);
    b.keys_b = (unsigned long long*)p; p += align256(n * 8);
    b.vals_a = (unsigned*)p; p += align256(n * 4);
    b.vals_b = (unsigned*)p; p += align256(n * 4);
    b.sort_scratch = p;
    return b;
}

}  // namespace syn3r

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

__device__ __forceinline__ float3 xf43(const float* m, float3 p) {
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xf44(const float* m, float3 p) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1) {
    x0 = min(gx, max(0, (int)((px - radius) / kTileX)));
    y0 = min(gy, max(0, (int)((py - radius) / kTileY)));
    x1 = min(gx, max(0, (int)((px + radius + kTileX - 1) / kTileX)));
    y1 = min(gy, max(0, (int)((py + radius + kTileY - 1) / kTileY)));
}

// view-dependent colour from spherical harmonics (degree D <= 3), +0.5, clamped at 0
__device__ float3 sh_to_rgb(int D, int M, float3 pos, const float* campos, const float* __restrict__ sh,
                            unsigned& clamped) {
    float3 dir = make_float3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
    float inv = 1.0f / sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x *= inv; dir.y *= inv; dir.z *= inv;
    auto c = [&](int k) { return make_float3(sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]); };
    auto mad = [](float3& a, float s, float3 b) { a.x += s * b.x; a.y += s * b.y; a.z += s * b.z; };
    float3 res = make_float3(0, 0, 0);
    mad(res, SH_C0, c(0));
    if (D > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        mad(res, -SH_C1 * y, c(1));
        mad(res, SH_C1 * z, c(2));
        mad(res, -SH_C1 * x, c(3));
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            mad(res, SH_C2[0] * xy, c(4));
            mad(res, SH_C2[1] * yz, c(5));
            mad(res, SH_C2[2] * (2.0f * zz - xx - yy), c(6));
            mad(res, SH_C2[3] * xz, c(7));
            mad(res, SH_C2[4] * (xx - yy), c(8));
            if (D > 2) {
                mad(res, SH_C3[0] * y * (3.0f * xx - yy), c(9));
                mad(res, SH_C3[1] * xy * z, c(10));
                mad(res, SH_C3[2] * y * (4.0f * zz - xx - yy), c(11));
                mad(res, SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), c(12));
                mad(res, SH_C3[4] * x * (4.0f * zz - xx - yy), c(13));
                mad(res, SH_C3[5] * z * (xx - yy), c(14));
                mad(res, SH_C3[6] * x * (xx - 3.0f * yy), c(15));
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamped = (res.x < 0.0f ? 1u : 0u) | (res.y < 0.0f ? 2u : 0u) | (res.z < 0.0f ? 4u : 0u);
    return make_float3(fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f));
}

__global__ void __launch_bounds__(256) k_preprocess(int N, int D, int M, const float* __restrict__ means3D,
                                                    const float* __restrict__ scales,
                                                    const float* __restrict__ rots,
                                                    const float* __restrict__ opacities,
                                                    const float* __restrict__ shs, const float* __restrict__ conf,
                                                    float scale_mod, Camera cam, int* __restrict__ radii,
                                                    GeomState g, int raw) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (i < 4) g.header[i] = 0u;            // pair count (written by the scan that follows) and overflow flag
    radii[i] = 0;
    g.tiles_touched[i] = 0;
    g.dkeys_a[i] = 0xFFFFFFFFu;             // culled Gaussians sort behind every visible one
    float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    float3 t = xf43(cam.view, p);
    if (t.z <= kNearClip) return;
    float4 ph = xf44(cam.proj, p);
    float pw = 1.0f / (ph.w + 0.0000001f);
    float ndcx = ph.x * pw, ndcy = ph.y * pw;

    // 3D covariance  Sigma = R S^2 R^T  (q = (r, x, y, z), not renormalised here)
    // raw (syn3r_raster_preprocess_raw): the tensors are the trainer's PARAMETERS (log-scales, unnormalised quaternions, opacity
    // logits) and the published activations are applied here, in k_activate's arithmetic (common.h: the same bits)
    float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    float4 q4 = make_float4(rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]);
    if (raw) {
        s0 = act_exp(s0); s1 = act_exp(s1); s2 = act_exp(s2);
        q4 = act_quat(q4, act_quat_inv_norm(q4));
    }
    float sx = scale_mod * s0, sy = scale_mod * s1, sz = scale_mod * s2;
    float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
    float R00 = 1.f - 2.f * (qy * qy + qz * qz), R01 = 2.f * (qx * qy - qr * qz), R02 = 2.f * (qx * qz + qr * qy);
    float R10 = 2.f * (qx * qy + qr * qz), R11 = 1.f - 2.f * (qx * qx + qz * qz), R12 = 2.f * (qy * qz - qr * qx);
    float R20 = 2.f * (qx * qz - qr * qy), R21 = 2.f * (qy * qz + qr * qx), R22 = 1.f - 2.f * (qx * qx + qy * qy);
    float m00 = R00 * sx, m01 = R01 * sy, m02 = R02 * sz;   // M = R S
    float m10 = R10 * sx, m11 = R11 * sy, m12 = R12 * sz;
    float m20 = R20 * sx, m21 = R21 * sy, m22 = R22 * sz;
    float c0 = m00 * m00 + m01 * m01 + m02 * m02;   // Sigma = M M^T
    float c1 = m00 * m10 + m01 * m11 + m02 * m12;
    float c2 = m00 * m20 + m01 * m21 + m02 * m22;
    float c3 = m10 * m10 + m11 * m11 + m12 * m12;
    float c4 = m10 * m20 + m11 * m21 + m12 * m22;
    float c5 = m20 * m20 + m21 * m21 + m22 * m22;
    float* cv = g.cov3D + 6 * (size_t)i;
    cv[0] = c0; cv[1] = c1; cv[2] = c2; cv[3] = c3; cv[4] = c4; cv[5] = c5;

    // EWA: cov2D = (J W) Sigma (J W)^T, W = rotation part of the view matrix
    float limx = kFovGuard * cam.tanfovx, limy = kFovGuard * cam.tanfovy;
    float tx = fminf(limx, fmaxf(-limx, t.x / t.z)) * t.z;
    float ty = fminf(limy, fmaxf(-limy, t.y / t.z)) * t.z;
    float J00 = cam.focal_x / t.z, J02 = -(cam.focal_x * tx) / (t.z * t.z);
    float J11 = cam.focal_y / t.z, J12 = -(cam.focal_y * ty) / (t.z * t.z);
    const float* v = cam.view;   // W[r][c] = v[c*4+r]
    float T00 = J00 * v[0] + J02 * v[2], T01 = J00 * v[4] + J02 * v[6], T02 = J00 * v[8] + J02 * v[10];
    float T10 = J11 * v[1] + J12 * v[2], T11 = J11 * v[5] + J12 * v[6], T12 = J11 * v[9] + J12 * v[10];
    float a0 = c0 * T00 + c1 * T01 + c2 * T02, a1 = c1 * T00 + c3 * T01 + c4 * T02, a2 = c2 * T00 + c4 * T01 + c5 * T02;
    float b0 = c0 * T10 + c1 * T11 + c2 * T12, b1 = c1 * T10 + c3 * T11 + c4 * T12, b2 = c2 * T10 + c4 * T11 + c5 * T12;
    float cxx = T00 * a0 + T01 * a1 + T02 * a2 + kLowPass;
    float cxy = T00 * b0 + T01 * b1 + T02 * b2;
    float cyy = T10 * b0 + T11 * b1 + T12 * b2 + kLowPass;

    float det = cxx * cyy - cxy * cxy;
    if (det == 0.0f) return;
    float det_inv = 1.0f / det;
    float conx = cyy * det_inv, cony = -cxy * det_inv, conz = cxx * det_inv;
    float mid = 0.5f * (cxx + cyy);
    float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda1 = mid + sq, lambda2 = mid - sq;
    int radius = (int)ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
    float px = ndc2pix(ndcx, cam.W), py = ndc2pix(ndcy, cam.H);
    int x0, y0, x1, y1;
    tile_rect(px, py, radius, cam.grid_x, cam.grid_y, x0, y0, x1, y1);
    int area = (x1 - x0) * (y1 - y0);
    if (area == 0) return;

    unsigned cl;
    float3 rgb = sh_to_rgb(D, M, p, cam.campos, shs + (size_t)i * M * 3, cl);
    float op = raw ? act_sigmoid(opacities[i]) : opacities[i];
    float cf = conf ? conf[i] : 1.0f;
    g.depths[i] = t.z;
    g.dkeys_a[i] = __float_as_uint(t.z);    // t.z > kNearClip > 0: the bit pattern orders like the float
    radii[i] = radius;
    g.means2D[2 * (size_t)i] = px;
    g.means2D[2 * (size_t)i + 1] = py;
    float* co = g.conic_opacity + 4 * (size_t)i;
    co[0] = conx; co[1] = cony; co[2] = conz; co[3] = op;
    g.rgb[3 * (size_t)i] = rgb.x; g.rgb[3 * (size_t)i + 1] = rgb.y; g.rgb[3 * (size_t)i + 2] = rgb.z;
    g.clamped[i] = cl;
    g.tiles_touched[i] = (unsigned)area;
    Splat s;
    s.x = px; s.y = py; s.cxx = conx; s.cxy = cony; s.cyy = conz; s.opacity = op * cf;
    s.r = rgb.x; s.g = rgb.y; s.b = rgb.z; s.depth = t.z; s.pad0 = 0.f; s.pad1 = 0.f;
    g.splats[i] = s;
}

// (tile id, Gaussian id) pairs, emitted in DEPTH order (position i of `order`), wave-cooperatively: the 64
// Gaussians of a wavefront own one contiguous output range; lane l of every 64-pair slice finds its owner by a
// 6-step search over the wavefront's exclusive tile counts (kept in registers, read by cross-lane shuffles) and
// writes one pair -> fully coalesced stores and no per-lane loops over differently sized tile rectangles.
__global__ void __launch_bounds__(256) k_dup_tiles(int N, const unsigned* __restrict__ order,
                                                   const float* __restrict__ means2D, const int* __restrict__ radii,
                                                   const unsigned* __restrict__ offsets, int gx, int gy,
                                                   unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                   unsigned cap, unsigned* __restrict__ header,
                                                   uint2* __restrict__ ranges, int tiles) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    for (int t = i; t < tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0u, 0u);   // k_tile_ranges fills the non-empty ones
    const int lane = threadIdx.x & 63;
    const bool valid = i < N;
    unsigned id = valid ? order[i] : 0u;
    int r = valid ? radii[id] : 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (r > 0) tile_rect(means2D[2 * (size_t)id], means2D[2 * (size_t)id + 1], r, gx, gy, x0, y0, x1, y1);
    const int w = x1 - x0;
    const unsigned cnt = (unsigned)(w * (y1 - y0));
    // wave-exclusive prefix of the counts; the output range starts at the first lane's global offset
    unsigned incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const unsigned local = incl - cnt;
    const unsigned total = __shfl(incl, 63, 64);
    const int first = blockIdx.x * 256 + (threadIdx.x & ~63);       // < N for every launched wavefront with work
    const unsigned base = first < N ? offsets[first] : 0u;
    const float inv_w = w > 0 ? 1.0f / (float)w : 0.0f;
    for (unsigned t = 0; t < total; t += 64) {
        const unsigned p = t + lane;
        int j = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {                   // largest j with local_j <= p
            unsigned lc = __shfl(local, j + step, 64);
            if (lc <= p) j += step;
        }
        const unsigned k = p - __shfl(local, j, 64);
        const int xj = __shfl(x0, j, 64), yj = __shfl(y0, j, 64), wj = __shfl(w, j, 64);
        const float iwj = __shfl(inv_w, j, 64);
        const unsigned idj = __shfl(id, j, 64);
        if (p < total) {
            // k / wj for k < 2^21: (k + 0.5) / wj is never within rounding distance of an integer
            const int ty = (int)(((float)k + 0.5f) * iwj);
            int tyc = ty;
            if ((unsigned)(tyc * wj) > k) --tyc;                      // reciprocal-multiply guard (at most one off)
            else if ((unsigned)((tyc + 1) * wj) <= k) ++tyc;
            const int tx = (int)k - tyc * wj;
            const unsigned o = base + p;
            if (o < cap) {
                keys[o] = (unsigned)((yj + tyc) * gx + xj + tx);
                vals[o] = idj;
            } else {
                header[1] = 1u;   // capacity overflow (asynchronous mode only): the caller re-renders
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_tile_ranges(long long P_cap, const unsigned* __restrict__ header,
                                                     const unsigned* __restrict__ keys,
                                                     uint2* __restrict__ ranges) {
    long long P = (long long)header[0];
    if (P > P_cap) P = P_cap;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    unsigned cur = keys[i];
    if (i == 0) ranges[cur].x = 0;
    else {
        unsigned prev = keys[i - 1];
        if (cur != prev) {
            ranges[prev].y = (unsigned)i;
            ranges[cur].x = (unsigned)i;
        }
    }
    if (i == P - 1) ranges[cur].y = (unsigned)P;
}

// ---------------------------------------------------------------------------------------------
// Hierarchical binning (round 5): the per-tile lists WITHOUT a pair sort and WITHOUT a global depth sort.
// The published pipeline duplicates every Gaussian into (tile, depth) keys and sorts them; rounds 1-4 of this build argsorted the
// Gaussians by depth (4 passes, 12 launches of 49 blocks: launch latency), emitted the pairs in that order and stably sorted them by
// tile id (two 7-bit passes over 2.6 M pairs) - 260 of the 950 us of a 1080p iteration.  A tile's list is the set of Gaussians whose
// tile rectangle holds the tile, ordered by (depth bits, index); so the lists are built by filtering and ordered where they are short:
//   super-tiles of 4 x 4 tiles (510 at 1080p; 8 x 8 where that would be more than 512: up to ~4K);
//   k_super_count / k_super_append  the Gaussians in chunks (index order): per (chunk, super-tile) the number of Gaussians whose
//                                   rectangle meets the super-tile (one LDS atomic each); every append block sums the count table's
//                                   columns itself (no scan launch) and appends its (depth bits << 32 | index) keys to the
//                                   super-tiles' lists, unordered inside the chunk;
//   k_super_sort                    a block per super-tile sorts its list (1 015 keys on average, 3 201 at most at 200 000 Gaussians)
//                                   in LDS - a bitonic network on 8-byte keys; longer lists are sorted in LDS-sized pieces and merged
//                                   through global memory - and leaves, in order, the ids and the rectangles clipped to the
//                                   super-tile (4 x 4 bits); its wavefronts then count, per tile of the super-tile, the entries
//                                   whose rectangle holds the tile (round 6: until then a launch of its own, a block per tile);
//   k_tile_offsets / k_tile_write   one single-block scan over the tiles (ranges, pair count, overflow flag, the blend kernels' tile
//                                   order), then a block per tile (four wavefronts, a quarter of the super-tile's list each,
//                                   coalesced, 2 B per entry) writes the entries whose rectangle holds the tile.
// Same lists as the (tile << 32 | depth) key sort, entry for entry - equal depths in index order, as a stable sort leaves them
// (tests/test_raster_gpu.py, test_raster_full_gpu.py against the oracle's sort); five launches where there were twenty-seven.
// Shapes it does not take (more than 512 super-tiles: images beyond ~4K) keep the argsort and the pair sort.
constexpr int kBinThreads = 512;          // Gaussians per round of a binning block
constexpr int kOffThreads = 1024;         // k_tile_offsets: one block
#ifndef SYN3R_MAX_SUPER
#define SYN3R_MAX_SUPER 512
#endif
constexpr int kMaxSuper = SYN3R_MAX_SUPER;     // (developer builds: -DSYN3R_MAX_SUPER=2048 takes 1080p to 2 x 2-tile super-tiles, profiles/r06/binning_supertiles.txt)

// exclusive scan of one value per thread over a 1024-thread block (sort.hip's block_exclusive_scan, local to this file)
__device__ __forceinline__ unsigned block_scan_1024(unsigned v, unsigned* smem /*[17]*/, unsigned& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) smem[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        const unsigned s0 = lane < 16 ? smem[lane] : 0u;
        unsigned si = s0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const unsigned t = __shfl_up(si, o, 64);
            if (lane >= o) si += t;
        }
        if (lane < 16) smem[lane] = si - s0;
        if (lane == 15) smem[16] = si;
    }
    __syncthreads();
    const unsigned res = incl - v + smem[wv];
    total = smem[16];
    __syncthreads();
    return res;
}

// Super-tiles of the SMALLEST side (1, 2, 4, 8 tiles) that gives at most 512 of them: 4 x 4 at 1080p (510 super-tiles, a block per
// list sorts ~1 000 keys), 8 x 8 up to ~4K; 2 x 2 at the LLFF / DTU image sizes the reference trains on (504 x 378: 192 super-tiles -
// with 4 x 4 its 48 lists of a 200 000-Gaussian model were ~4x longer than the LDS piece and k_super_sort's single-block global-
// memory merge took 370 of the iteration's 957 us, profiles/r06/raster_breakdown_504x378.txt), single tiles below 512
// tiles; `ss` = log2 of the side.  The Gaussians go through the count / append kernels in chunks of
// rounds x 512, as many chunks as keep the (chunk, super-tile) count table within kBinCounters.
struct BinPlan { int ss, sgx, sgy, nsuper, rounds, nchunks; bool ok; };
inline BinPlan bin_plan(int N, int gx, int gy) {
    BinPlan b;
    for (b.ss = 0; b.ss <= 3; ++b.ss) {
        const int side = 1 << b.ss;
        b.sgx = (gx + side - 1) / side; b.sgy = (gy + side - 1) / side;
        b.nsuper = b.sgx * b.sgy;
        if (b.nsuper <= kMaxSuper) break;
    }
    if (b.ss > 3) b.ss = 3;
    b.ok = b.nsuper <= kMaxSuper;
    const int max_chunks = b.ok ? (int)(kBinCounters / (size_t)b.nsuper) : 1;
    const int units = (N + kBinThreads - 1) / kBinThreads;
    b.rounds = (units + max_chunks - 1) / max_chunks;
    if (b.rounds < 1) b.rounds = 1;
    b.nchunks = (units + b.rounds - 1) / b.rounds;
    return b;
}

// does this shape take the hierarchical binning?  (SYN3R_BIN_HIER=0 in tuning builds: the pair sort, for A/B runs)
inline bool hier_binning(int N, int gx, int gy) {
    static const int on = tune_env("SYN3R_BIN_HIER", 1);
    return on != 0 && bin_plan(N, gx, gy).ok;
}

// the tile rectangle of Gaussian i (empty for culled ones)
struct BinRect { int x0, y0, x1, y1; };
__device__ __forceinline__ BinRect bin_rect(int i, int N, const float* __restrict__ means2D, const int* __restrict__ radii, int gx, int gy) {
    BinRect r; r.x0 = r.y0 = r.x1 = r.y1 = 0;
    if (i < N) {
        const int rad = radii[i];
        if (rad > 0) tile_rect(means2D[2 * (size_t)i], means2D[2 * (size_t)i + 1], rad, gx, gy, r.x0, r.y0, r.x1, r.y1);
    }
    return r;
}
__device__ __forceinline__ unsigned short clip_rect(const BinRect& q, int sx, int sy, int ss) {   // x0 | x1 << 4 | y0 << 8 | y1 << 12, each 0..8
    const int bx = sx << ss, by = sy << ss, side = 1 << ss;
    const int rx0 = max(q.x0 - bx, 0), rx1 = min(q.x1 - bx, side), ry0 = max(q.y0 - by, 0), ry1 = min(q.y1 - by, side);
    return (unsigned short)(rx0 | (rx1 << 4) | (ry0 << 8) | (ry1 << 12));
}

__global__ void __launch_bounds__(kBinThreads) k_super_count(int N, const float* __restrict__ means2D, const int* __restrict__ radii,
                                                             int gx, int gy, int ss, int sgx, int nsuper, int rounds, int nchunks,
                                                             unsigned* __restrict__ counters, unsigned* __restrict__ header) {
    __shared__ unsigned tot[kMaxSuper];
    for (int s = threadIdx.x; s < nsuper; s += kBinThreads) tot[s] = 0u;
    __syncthreads();
    int pairs = 0;            // the EXACT number of (Gaussian, tile) pairs of this block's Gaussians: header[3] (see k_tile_offsets)
    for (int r = 0; r < rounds; ++r) {
        const int i = (blockIdx.x * rounds + r) * kBinThreads + threadIdx.x;
        const BinRect q = bin_rect(i, N, means2D, radii, gx, gy);
        if (q.x1 > q.x0 && q.y1 > q.y0) {
            pairs += (q.x1 - q.x0) * (q.y1 - q.y0);
            const int side1 = (1 << ss) - 1, sx0 = q.x0 >> ss, sx1 = (q.x1 + side1) >> ss, sy0 = q.y0 >> ss, sy1 = (q.y1 + side1) >> ss;
            for (int sy = sy0; sy < sy1; ++sy)
                for (int sx = sx0; sx < sx1; ++sx) atomicAdd(&tot[sy * sgx + sx], 1u);
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nsuper; s += kBinThreads) counters[(size_t)blockIdx.x * nsuper + s] = tot[s];   // chunk-major: k_super_append reads it coalesced
    // ONE global atomic per block (integer: any order gives the same sum): the wavefronts meet in LDS first - eight atomics per block
    // on one address cost the kernel 8 of its 14 us
    __shared__ unsigned pairs_s;
    if (threadIdx.x == 0) pairs_s = 0u;
    __syncthreads();
    pairs = wave_sum_i(pairs);
    if ((threadIdx.x & 63) == 0 && pairs) atomicAdd(&pairs_s, (unsigned)pairs);
    __syncthreads();
    if (threadIdx.x == 0 && pairs_s) atomicAdd(&header[3], pairs_s);
}

// Appends the keys (depth bits << 32 | index) of this chunk's Gaussians to the lists of the super-tiles their rectangles meet.
// Where the chunk's entries of a super-tile start: the (chunk, super-tile) counts are not scanned by a launch of their own - every
// block sums the table's columns (a few parts per super-tile, coalesced over the super-tiles; a few hundred KB of L2 reads per
// block), a wavefront scans the totals, block 0 publishes the list starts for the kernels that follow.
__global__ void __launch_bounds__(kBinThreads) k_super_append(int N, const float* __restrict__ depths, const float* __restrict__ means2D,
                                                              const int* __restrict__ radii, int gx, int gy, int ss, int sgx, int nsuper,
                                                              int rounds, int nchunks, const unsigned* __restrict__ counters,
                                                              unsigned* __restrict__ sstart, unsigned cap,
                                                              unsigned long long* __restrict__ skeys, unsigned* __restrict__ header) {
    __shared__ unsigned run[kMaxSuper];            // next free position of (super-tile, this chunk)
    __shared__ unsigned tot_s[kMaxSuper], pre_s[kMaxSuper];
    __shared__ unsigned part_tot[kMaxSuper > kBinThreads ? kMaxSuper : kBinThreads], part_pre[kMaxSuper > kBinThreads ? kMaxSuper : kBinThreads];   // nsuper * PARTS <= max(kBinThreads, nsuper)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        const int PARTS = max(1, min(8, kBinThreads / nsuper));  // (super-tile, part) items: one per thread, one pass
        const int per = (nchunks + PARTS - 1) / PARTS;
        for (int w = threadIdx.x; w < nsuper * PARTS; w += kBinThreads) {
            const int s_ = w % nsuper, part = w / nsuper;
            const int c0 = part * per, c1 = min(nchunks, c0 + per);
            unsigned tot = 0, pre = 0;
            constexpr int UN = 32;
            for (int cb = c0; cb < c1; cb += UN) {
                unsigned c[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) c[u] = cb + u < c1 ? counters[(size_t)(cb + u) * nsuper + s_] : 0u;
#pragma unroll
                for (int u = 0; u < UN; ++u) { tot += c[u]; pre += cb + u < (int)blockIdx.x ? c[u] : 0u; }
            }
            part_tot[w] = tot; part_pre[w] = pre;
        }
        __syncthreads();
        for (int s_ = threadIdx.x; s_ < nsuper; s_ += kBinThreads) {
            unsigned tot = 0, pre = 0;
            for (int k = 0; k < PARTS; ++k) { tot += part_tot[k * nsuper + s_]; pre += part_pre[k * nsuper + s_]; }
            tot_s[s_] = tot; pre_s[s_] = pre;
        }
        __syncthreads();
        if (wv == 0) {
            constexpr int PER = kMaxSuper / 64;
            unsigned c[PER], sum = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int s_ = lane * PER + k; c[k] = s_ < nsuper ? tot_s[s_] : 0u; sum += c[k]; }
            unsigned incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            unsigned ex = incl - sum;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int s_ = lane * PER + k;
                if (s_ < nsuper) {
                    run[s_] = ex + pre_s[s_];
                    if (blockIdx.x == 0) sstart[s_] = ex;
                }
                ex += c[k];
            }
            if (blockIdx.x == 0 && lane == 63) { sstart[nsuper] = incl; header[2] = incl; }
        }
        __syncthreads();
    }
    for (int r = 0; r < rounds; ++r) {
        const int i = (blockIdx.x * rounds + r) * kBinThreads + threadIdx.x;
        const BinRect q = bin_rect(i, N, means2D, radii, gx, gy);
        if (q.x1 > q.x0 && q.y1 > q.y0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(depths[i]) << 32) | (unsigned)i;
            const int side1 = (1 << ss) - 1, sx0 = q.x0 >> ss, sx1 = (q.x1 + side1) >> ss, sy0 = q.y0 >> ss, sy1 = (q.y1 + side1) >> ss;
            for (int sy = sy0; sy < sy1; ++sy)
                for (int sx = sx0; sx < sx1; ++sx) {
                    const unsigned o = atomicAdd(&run[sy * sgx + sx], 1u);
                    if (o < cap) skeys[o] = key;
                    else header[1] = 1u;
                }
        }
    }
}

// the list of super-tile s: [sstart[s], sstart[s + 1]) clipped to the capacity
__device__ __forceinline__ void super_range(int s, const unsigned* __restrict__ sstart, unsigned cap, unsigned& b, unsigned& e) {
    b = min(sstart[s], cap); e = min(sstart[s + 1], cap);
}

// A block per super-tile: its list of keys into ascending order (depth bits, then index: what a stable sort by depth leaves), then
// the ids and clipped rectangles of the entries.  Bitonic network in the form whose every comparison puts the smaller key at the
// lower index (first step of a merge: partner l ^ (k - 1); later steps: l + j), so the virtual +inf padding up to a power of two
// never moves and any length sorts.  Up to kSortLds keys (every list of the 200 000-Gaussian benchmark: 3 201 at most) the whole
// network runs in LDS; beyond, LDS-sized pieces are sorted,
// and for the larger merges the steps whose distance reaches across pieces run on global memory (one block: __syncthreads orders
// them), the rest again in LDS.
constexpr int kSortLds = 4096;                    // keys per LDS piece (32 KB: four blocks per CU, every super-tile's block resident at once)
constexpr int kSuperSortThreads = 1024;
__device__ __forceinline__ void cmpx(unsigned long long& a, unsigned long long& b) { if (a > b) { const unsigned long long t = a; a = b; b = t; } }
// pair of comparison t in a step of distance j (a power of two): l = t with a zero inserted at bit log2(j), r = l + j
__device__ __forceinline__ int pair_lo(int t, int j) { return (t << 1) - (t & (j - 1)); }
// The steps of the network on `buf[0, m)` (m a power of two, LDS) are taken TWO per pass: a thread owns the four keys that two
// consecutive steps connect, exchanges them in registers and stores them - half the barriers and half the LDS round trips of a
// step per pass (the network is bound by those, not by its comparisons: 78 steps -> 42 passes at 4 096 keys).
// distances j, j/2, .. 1 of a merge (pairs (l, l + j)), two distances per pass, a last single one if their number is odd
__device__ __forceinline__ void bitonic_dists(unsigned long long* buf, int m, int j) {
    for (; j >= 2; j >>= 2) {
        const int jh = j >> 1;
        for (int t = threadIdx.x; t < (m >> 2); t += kSuperSortThreads) {
            const int e0 = pair_lo(pair_lo(t, jh), j);
            unsigned long long a = buf[e0], b = buf[e0 + jh], c = buf[e0 + j], d = buf[e0 + j + jh];
            cmpx(a, c); cmpx(b, d);          // distance j
            cmpx(a, b); cmpx(c, d);          // distance j / 2
            buf[e0] = a; buf[e0 + jh] = b; buf[e0 + j] = c; buf[e0 + j + jh] = d;
        }
        __syncthreads();
    }
    if (j == 1) {
        for (int t = threadIdx.x; t < (m >> 1); t += kSuperSortThreads) cmpx(buf[2 * t], buf[2 * t + 1]);
        __syncthreads();
    }
}
// all steps of merge size k: the flip step (pairs (l, l ^ (k - 1)) inside blocks of k) together with distance k / 4, then the rest
__device__ __forceinline__ void bitonic_merge(unsigned long long* buf, int m, int k) {
    if (k == 2) {
        for (int t = threadIdx.x; t < (m >> 1); t += kSuperSortThreads) cmpx(buf[2 * t], buf[2 * t + 1]);
        __syncthreads();
        return;
    }
    const int q = k >> 2;
    for (int t = threadIdx.x; t < (m >> 2); t += kSuperSortThreads) {
        const int off = t & (q - 1), blk = (t - off) << 2;           // (t / q) * k
        const int ia = blk + off, ib = ia + q, ird = blk + (k - 1 - off), irc = ird - q;
        unsigned long long a = buf[ia], b = buf[ib], c = buf[irc], d = buf[ird];
        cmpx(a, d); cmpx(b, c);              // flip
        cmpx(a, b); cmpx(c, d);              // distance k / 4
        buf[ia] = a; buf[ib] = b; buf[irc] = c; buf[ird] = d;
    }
    __syncthreads();
    bitonic_dists(buf, m, q >> 1);
}

__device__ __forceinline__ bool rect_has(unsigned r, int rx, int ry) {
    return rx >= (int)(r & 15u) && rx < (int)((r >> 4) & 15u) && ry >= (int)((r >> 8) & 15u) && ry < (int)(r >> 12);
}

constexpr int kBinUnroll = 8;
// k_tile_write: a block per tile, its four wavefronts a quarter of the super-tile's list each (the longest list sets the kernel's
// time: a wavefront per tile left the tiles of the busiest super-tile walking ~8 000 entries while the rest of the chip had finished).
__device__ __forceinline__ void tile_quarter(unsigned b, unsigned e, int q, unsigned& qb, unsigned& qe) {
    const unsigned len = (e - b + 3u) / 4u;
    qb = min(e, b + (unsigned)q * len); qe = min(e, qb + len);
}

// The per-tile, per-quarter entry counts of super-tile (sx, sy) (tcount[4 t + q], what k_tile_offsets scans and k_tile_write starts
// from) by the block that has just ordered the list: wavefront w takes tiles w, w + 16, ... of the super-tile and walks the clipped
// rectangles (2 B per entry: in LDS where the list was sorted there, else as written a moment ago by this block - visible after
// the barrier: one workgroup, one CU), quarter by quarter.  Until round 6 a launch of its own (k_tile_count, a block per tile:
// 10 us + a launch at 1080p).
__device__ __forceinline__ void super_tile_counts(int sx, int sy, int ss, int gx, int gy, unsigned n, const unsigned short* srect_b,
                                                  unsigned* __restrict__ tcount) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, side = 1 << ss;
    const unsigned len = (n + 3u) / 4u;                        // tile_quarter's split of [0, n)
    // (tile, quarter) items over the block's 16 wavefronts: 2 x 2-tile super-tiles (the LLFF sizes, lists of ~10 000) keep all of
    // them busy, 4 x 4 gives a wavefront the four quarters of one tile.  Per-lane counts on the vector unit and one cross-lane sum
    // per item (the scalar unit is shared by the CU's wavefronts: per-chunk ballot masks doubled this kernel's time); four chunks
    // of 64 entries are requested before the first is tested (the long lists are read from global memory).
    for (int it = wv; it < side * side * 4; it += kSuperSortThreads / 64) {
        const int ti = it >> 2, q = it & 3;
        const int rx = ti & (side - 1), ry = ti >> ss;
        const int tx = (sx << ss) + rx, ty = (sy << ss) + ry;
        if (tx >= gx || ty >= gy) continue;                    // (wave-uniform)
        const unsigned lo = min(n, (unsigned)q * len), hi = min(n, lo + len);
        unsigned cnt = 0;
        for (unsigned i0 = lo; i0 < hi; i0 += 256) {
            unsigned short r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = srect_b[min(i0 + u * 64 + lane, hi - 1u)];
#pragma unroll
            for (int u = 0; u < 4; ++u) cnt += (i0 + u * 64 + lane < hi && rect_has(r[u], rx, ry)) ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) tcount[4 * ((size_t)ty * gx + tx) + q] = cnt;
    }
}

// (8 wavefronts per SIMD = two blocks per CU: 510 blocks at 1080p are resident at once - at 72 registers only one block fits and the
// kernel runs its blocks in two rounds, 36 -> 70 us)
__global__ void __launch_bounds__(kSuperSortThreads, 8) k_super_sort(int N, const float* __restrict__ means2D, const int* __restrict__ radii,
                                                                  int gx, int gy, int ss, int sgx, const unsigned* __restrict__ sstart, unsigned cap,
                                                                  unsigned long long* __restrict__ skeys, unsigned* __restrict__ sid,
                                                                  unsigned short* __restrict__ srect, unsigned* __restrict__ tcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];   // kSortLds keys
    const int s_ = blockIdx.x, sx = s_ % sgx, sy = s_ / sgx;
    unsigned b, e;
    super_range(s_, sstart, cap, b, e);
    const int n = (int)(e - b);
    if (n == 0) { super_tile_counts(sx, sy, ss, gx, gy, 0u, srect, tcount); return; }     // (zeros for its tiles)
    unsigned long long* gk = skeys + b;
    const unsigned long long INF = ~0ull;
    int m = 1;
    while (m < n) m <<= 1;
    const unsigned long long* sorted;              // where the sorted keys end up
    if (m <= kSortLds) {
        for (int i = threadIdx.x; i < m; i += kSuperSortThreads) sk[i] = i < n ? gk[i] : INF;
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1) bitonic_merge(sk, m, k);
        sorted = sk;
    } else {
        // pieces of kSortLds keys, each through the whole network in LDS
        for (int p0 = 0; p0 < n; p0 += kSortLds) {
            for (int i = threadIdx.x; i < kSortLds; i += kSuperSortThreads) sk[i] = p0 + i < n ? gk[p0 + i] : INF;
            __syncthreads();
            for (int k = 2; k <= kSortLds; k <<= 1) bitonic_merge(sk, kSortLds, k);
            for (int i = threadIdx.x; i < kSortLds; i += kSuperSortThreads) if (p0 + i < n) gk[p0 + i] = sk[i];
            __syncthreads();
        }
        // merges across pieces: the steps of distance >= kSortLds on global memory (a partner beyond n is +inf: nothing to do)
        for (int k = kSortLds << 1; k <= m; k <<= 1) {
            const int h = k >> 1;
            for (int t = threadIdx.x; t < (m >> 1); t += kSuperSortThreads) {
                const int off = t & (h - 1), l = pair_lo(t, h), r = l - off + (k - 1 - off);
                if (r < n) { unsigned long long a = gk[l], c = gk[r]; if (a > c) { gk[l] = c; gk[r] = a; } }
            }
            __syncthreads();
            for (int j = k >> 2; j >= kSortLds; j >>= 1) {
                for (int t = threadIdx.x; t < (m >> 1); t += kSuperSortThreads) {
                    const int l = pair_lo(t, j), r = l + j;
                    if (r < n) { unsigned long long a = gk[l], c = gk[r]; if (a > c) { gk[l] = c; gk[r] = a; } }
                }
                __syncthreads();
            }
            for (int p0 = 0; p0 < n; p0 += kSortLds) {            // distances below a piece: in LDS
                for (int i = threadIdx.x; i < kSortLds; i += kSuperSortThreads) sk[i] = p0 + i < n ? gk[p0 + i] : INF;
                __syncthreads();
                bitonic_dists(sk, kSortLds, kSortLds >> 1);
                for (int i = threadIdx.x; i < kSortLds; i += kSuperSortThreads) if (p0 + i < n) gk[p0 + i] = sk[i];
                __syncthreads();
            }
        }
        sorted = gk;
    }
    if (sorted == sk) {
        // (n <= kSortLds = 4 x kSuperSortThreads) the clipped rectangles also go to LDS - over the keys, once every thread has read
        // its ids - where the tile counts read them: 2 B per entry and tile from global memory, one dependent load per 64 entries,
        // made the longest list's block the kernel's time (36 -> 74 us)
        unsigned short clip[kSortLds / kSuperSortThreads];
#pragma unroll
        for (int u = 0; u < kSortLds / kSuperSortThreads; ++u) {
            const int i = threadIdx.x + u * kSuperSortThreads;
            clip[u] = 0;
            if (i < n) {
                const unsigned id = (unsigned)sk[i];
                const BinRect q = bin_rect((int)id, N, means2D, radii, gx, gy);
                clip[u] = clip_rect(q, sx, sy, ss);
                sid[b + i] = id;
                srect[b + i] = clip[u];
            }
        }
        __syncthreads();
        unsigned short* sr = (unsigned short*)sk;
#pragma unroll
        for (int u = 0; u < kSortLds / kSuperSortThreads; ++u) {
            const int i = threadIdx.x + u * kSuperSortThreads;
            if (i < n) sr[i] = clip[u];
        }
        __syncthreads();
        super_tile_counts(sx, sy, ss, gx, gy, (unsigned)n, sr, tcount);
        return;
    }
    for (int i = threadIdx.x; i < n; i += kSuperSortThreads) {
        const unsigned id = (unsigned)sorted[i];
        const BinRect q = bin_rect((int)id, N, means2D, radii, gx, gy);
        sid[b + i] = id;
        srect[b + i] = clip_rect(q, sx, sy, ss);
    }
    __syncthreads();                 // the block's rectangles are in memory for the block
    super_tile_counts(sx, sy, ss, gx, gy, (unsigned)n, srect + b, tcount);
}

// ranges[t] = (start, end) of tile t's list, clipped to the capacity; header[0] = the pair count, header[1] = 1 if it (or the
// super-tile lists) did not fit.  One block, any tile count.
__global__ void __launch_bounds__(kOffThreads) k_tile_offsets(int tiles, const unsigned* __restrict__ tcount, unsigned cap,
                                                              uint2* __restrict__ ranges, unsigned* __restrict__ header,
                                                              unsigned* __restrict__ tile_order) {
    __shared__ unsigned smem[17];
    __shared__ unsigned bh[256];
    if (tile_order) { for (int k = threadIdx.x; k < 256; k += kOffThreads) bh[k] = 0u; __syncthreads(); }
    constexpr int PER = 8;                                       // consecutive tiles per thread: 8 192 tiles per round
    unsigned carry = 0;
    for (int start = 0; start < tiles; start += kOffThreads * PER) {
        const int t0 = start + threadIdx.x * PER;
        unsigned c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            c[k] = 0u;
            if (t0 + k < tiles) { const uint4 q4 = ((const uint4*)tcount)[t0 + k]; c[k] = (q4.x + q4.y) + (q4.z + q4.w); }
            sum += c[k];
        }
        unsigned total;
        unsigned ex = block_scan_1024(sum, smem, total) + carry;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            // an empty tile is (0, 0), as k_tile_ranges leaves it
            if (t0 + k < tiles) ranges[t0 + k] = c[k] ? make_uint2(min(ex, cap), min(ex + c[k], cap)) : make_uint2(0u, 0u);
            if (tile_order && t0 + k < tiles) atomicAdd(&bh[255u - min(255u, c[k] >> 5)], 1u);
            ex += c[k];
        }
        carry += total;
    }
    if (tile_order) {        // tiles by descending list length (buckets of 32 entries): the blend kernels take the long ones first
        __syncthreads();
        if (threadIdx.x < 64) {                                   // exclusive scan of the 256 bucket counts: 4 per lane
            const int l = threadIdx.x;
            unsigned c[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[k] = bh[4 * l + k]; sum += c[k]; }
            unsigned incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (l >= o) incl += t; }
            unsigned ex = incl - sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) { bh[4 * l + k] = ex; ex += c[k]; }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < tiles; t += kOffThreads) {
            const uint4 q4 = ((const uint4*)tcount)[t];
            const unsigned c = (q4.x + q4.y) + (q4.z + q4.w);
            tile_order[atomicAdd(&bh[255u - min(255u, c >> 5)], 1u)] = (unsigned)t;
        }
    }
    if (threadIdx.x == 0) {
        // `carry` sums the per-tile counts taken from super-tile lists that are CLIPPED to the capacity: once a render overflows it
        // under-counts (every key past `cap` is dropped and uncounted), and a caller that sizes its next buffer from header[0] would
        // at best double per round.  header[3] is the exact pair count (sum of the tile rectangles' areas, k_super_count): reported
        // whenever the lists did not fit, so ONE re-render recovers (tests/test_raster_gpu.py).
        const bool over = carry > cap || header[2] > cap;
        header[0] = over ? max(carry, max(header[3], header[2])) : carry;
        if (over) header[1] = 1u;
    }
}

__global__ void __launch_bounds__(256) k_tile_write(int gx, int tiles, int ss, int sgx, const unsigned* __restrict__ sstart, unsigned cap,
                                                    const unsigned* __restrict__ sid, const unsigned short* __restrict__ srect,
                                                    const uint2* __restrict__ ranges, const unsigned* __restrict__ tcount,
                                                    unsigned* __restrict__ point_list) {
    const int t = blockIdx.x, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = t % gx, ty = t / gx;
    unsigned b, e;
    super_range((ty >> ss) * sgx + (tx >> ss), sstart, cap, b, e);
    tile_quarter(b, e, q, b, e);
    const int rx = tx & ((1 << ss) - 1), ry = ty & ((1 << ss) - 1);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint4 q4 = ((const uint4*)tcount)[t];
    unsigned pos = ranges[t].x + (q > 0 ? q4.x : 0u) + (q > 1 ? q4.y : 0u) + (q > 2 ? q4.z : 0u);
    for (unsigned i0 = b; i0 < e; i0 += 64 * kBinUnroll) {       // wave-uniform trip count
        unsigned short r[kBinUnroll];
        unsigned id[kBinUnroll];
#pragma unroll
        for (int u = 0; u < kBinUnroll; ++u) {
            const unsigned i = i0 + u * 64 + lane;
            r[u] = i < e ? srect[i] : (unsigned short)0;
            id[u] = i < e ? sid[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kBinUnroll; ++u) {
            const bool hit = rect_has(r[u], rx, ry);
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const unsigned o = pos + (unsigned)__popcll(m & lt);
                if (o < cap) point_list[o] = id[u];
            }
            pos += (unsigned)__popcll(m);
        }
    }
}

// contiguous-chunk-per-XCD remap of a 1-D block id (bijective for any block count)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// Two pixels per lane (see k_render_bwd in raster_bwd.hip): a 16 x 16 tile is a block of TWO wavefronts, wavefront w
// owns the 16 x 8 half (rows 8w .. 8w+7) and lane l the pixels (l & 15, 8w + (l >> 4)) and (.., + 4).  The
// quadratic form, the exponent argument, the weights and the colour / depth accumulation are float2 arithmetic
// (v_pk_fma_f32 / v_pk_mul_f32: two pixels per VALU issue); exp, min and the compares stay one per pixel.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kFwdThreads = 128;
#ifdef SYN3R_RASTER_STATS      // developer build: [0] lane tests, [1] wavefront visits, [2] visits with an active pixel, [3] active pixels
__device__ unsigned long long g_fwd_stats[4];
#define FSTAT(i, n) do { if (lane == 0) atomicAdd(&g_fwd_stats[i], (unsigned long long)(n)); } while (0)
#else
#define FSTAT(i, n)
#endif

__global__ void __launch_bounds__(kFwdThreads) k_render(int H, int W, int gx, int gy, const uint2* __restrict__ ranges,
                                                        const unsigned* __restrict__ point_list,
                                                        const Splat* __restrict__ splats, float bg0, float bg1,
                                                        float bg2, unsigned* __restrict__ n_contrib,
                                                        float* __restrict__ final_T, float* __restrict__ out_color,
                                                        float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                        const unsigned* __restrict__ tile_order) {
    __shared__ float4 sm[kFwdThreads * 3];
    const unsigned tile = tile_order ? tile_order[blockIdx.x] : xcd_remap(blockIdx.x, (unsigned)(gx * gy));
    const int tx = tile % gx, ty = tile / gx;
    const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = lane & 15, ly = wq * 8 + (lane >> 4);
    const int px = tx * kTileX + lx, py0 = ty * kTileY + ly, py1 = py0 + 4;
    const bool in0 = px < W && py0 < H, in1 = px < W && py1 < H;
    const float fx = (float)px;
    const f2 fy = (f2){(float)py0, (float)py1};
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + kFwdThreads - 1) / kFwdThreads;

    bool done0 = !in0, done1 = !in1;
    f2 T = (f2){1.0f, 1.0f}, Cr = (f2){0.f, 0.f}, Cg = Cr, Cb = Cr, Dp = Cr;
    unsigned last0 = 0, last1 = 0;
    int todo = total;
    const float sx0 = (float)(tx * kTileX), sx1 = sx0 + 15.0f;
    const float sy0 = (float)(ty * kTileY + wq * 8), sy1 = sy0 + 7.0f;
    // The records of round rd + 1 are requested (list entry, then the 48-byte record: two dependent global loads)
    // BEFORE round rd is blended and land in registers meanwhile: the gather latency is off the critical path.
    float4 n0, n1, n2;
    bool have = false;
    auto fetch = [&](int rd) {
        const int idx = rd * kFwdThreads + threadIdx.x;
        have = idx < total;
        if (have) {
            const float4* src = (const float4*)(splats + point_list[range.x + idx]);
            n0 = src[0]; n1 = src[1]; n2 = src[2];
        }
    };
    fetch(0);
    for (int rd = 0; rd < rounds; ++rd, todo -= kFwdThreads) {
        if (__syncthreads_count(done0 && done1) == kFwdThreads) break;
        if (have) {
            sm[threadIdx.x * 3 + 0] = n0;
            sm[threadIdx.x * 3 + 1] = n1;
            sm[threadIdx.x * 3 + 2] = n2;
        }
        fetch(rd + 1);
        __syncthreads();
        const int cnt = min(kFwdThreads, todo);
        // Visit list: each lane tests ONE staged splat against the wavefront's 16 x 8 half (splat_reaches_rect); the
        // ballot is the list, walked in order with scalar bit operations.  Most (wavefront, splat) visits of the
        // 3-sigma tile lists never reach alpha >= 1/255 on the half and are skipped for the price of one lane-test
        // instead of a 128-pixel evaluation.
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            if (__ballot(!(done0 && done1)) == 0ull) break;
            bool hit = false;
            if (c0 + lane < cnt) {
                const float4 a = sm[(c0 + lane) * 3], b = sm[(c0 + lane) * 3 + 1];
                hit = splat_reaches_rect(a.x, a.y, a.z, a.w, b.x, b.y, sx0, sx1, sy0, sy1);
            }
            unsigned long long m = __ballot(hit);
            FSTAT(0, min(64, cnt - c0));
            FSTAT(1, __popcll(m));
            while (m) {
                const int j = c0 + (int)__builtin_ctzll(m);
                m &= m - 1;
                const float4 a = sm[j * 3], b = sm[j * 3 + 1], c = sm[j * 3 + 2];
                // a = (x, y, cxx, cxy)  b = (cyy, opacity, r, g)  c = (b, depth, -, -)
                const float dx = a.x - fx;
                const f2 dy = (f2){a.y, a.y} - fy;
                const float hxx = -0.5f * a.z * dx * dx, bxy = a.w * dx;
                const f2 power = (-0.5f * b.x) * dy * dy - bxy * dy + hxx;
                const f2 araw = b.y * (f2){__expf(power.x), __expf(power.y)};
                const float al0 = fminf(kAlphaMax, araw.x), al1 = fminf(kAlphaMax, araw.y);
                const f2 test_T = T * (1.0f - (f2){al0, al1});
                const bool c0_ = !done0 && power.x <= 0.0f && al0 >= kAlphaMin;
                const bool c1_ = !done1 && power.y <= 0.0f && al1 >= kAlphaMin;
                if (c0_ && test_T.x < kTransmittanceMin) done0 = true;
                if (c1_ && test_T.y < kTransmittanceMin) done1 = true;
                const bool t0 = c0_ && !done0, t1 = c1_ && !done1;
#ifdef SYN3R_RASTER_STATS
                { unsigned long long b0 = __ballot(t0), b1 = __ballot(t1); FSTAT(2, (b0 | b1) != 0ull); FSTAT(3, __popcll(b0) + __popcll(b1)); }
#endif
                // a pixel that does not take the splat adds a zero weight and keeps its transmittance: branch-free
                const f2 w = (f2){t0 ? al0 : 0.0f, t1 ? al1 : 0.0f} * T;
                Cr += b.z * w; Cg += b.w * w; Cb += c.x * w; Dp += c.y * w;
                T = (f2){t0 ? test_T.x : T.x, t1 ? test_T.y : T.y};
                const unsigned here = (unsigned)(rd * kFwdThreads + j + 1);
                last0 = t0 ? here : last0;
                last1 = t1 ? here : last1;
            }
        }
    }
    const size_t hw = (size_t)H * W;
    if (in0) {
        const size_t pix = (size_t)py0 * W + px;
        final_T[pix] = T.x;
        n_contrib[pix] = last0;
        out_color[pix] = Cr.x + T.x * bg0;
        out_color[hw + pix] = Cg.x + T.x * bg1;
        out_color[2 * hw + pix] = Cb.x + T.x * bg2;
        out_depth[pix] = Dp.x;
        out_alpha[pix] = 1.0f - T.x;
    }
    if (in1) {
        const size_t pix = (size_t)py1 * W + px;
        final_T[pix] = T.y;
        n_contrib[pix] = last1;
        out_color[pix] = Cr.y + T.y * bg0;
        out_color[hw + pix] = Cg.y + T.y * bg1;
        out_color[2 * hw + pix] = Cb.y + T.y * bg2;
        out_depth[pix] = Dp.y;
        out_alpha[pix] = 1.0f - T.y;
    }
}

int bits_for(unsigned v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

void fill_camera(Camera& cam, const float* view, const float* proj, const float* campos, float tanfovx, float tanfovy,
                 int H, int W) {
    for (int i = 0; i < 16; ++i) { cam.view[i] = view[i]; cam.proj[i] = proj[i]; }
    for (int i = 0; i < 3; ++i) cam.campos[i] = campos[i];
    cam.tanfovx = tanfovx; cam.tanfovy = tanfovy;
    cam.focal_x = W / (2.0f * tanfovx);
    cam.focal_y = H / (2.0f * tanfovy);
    cam.H = H; cam.W = W;
    cam.grid_x = (W + kTileX - 1) / kTileX;
    cam.grid_y = (H + kTileY - 1) / kTileY;
}

}  // namespace

namespace syn3r {
void raster_fill_camera(Camera& cam, const float* view, const float* proj, const float* campos, float tanfovx,
                        float tanfovy, int H, int W) {
    fill_camera(cam, view, proj, campos, tanfovx, tanfovy, H, W);
}
bool raster_tiles_ordered(int N, int gx, int gy) {
    static const int order_env = tune_env("SYN3R_TILE_ORDER", 1);
    return order_env != 0 && hier_binning(N, gx, gy);
}
}  // namespace syn3r

extern "C" size_t syn3r_raster_geom_bytes(int N) { return SYN3R_DIM_OK(N) ? geom_bytes(N) : 0; }
extern "C" size_t syn3r_raster_image_bytes(int H, int W) { return (SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W)) ? image_bytes(H, W) : 0; }
extern "C" size_t syn3r_raster_binning_bytes(long long P) { return P >= 0 ? binning_bytes(P) : 0; }

static int raster_preprocess(int raw, int N, int sh_degree, int sh_coeffs, const float* means3D,
                             const float* scales, const float* rotations, const float* opacities,
                             const float* shs, const float* confidence, float scale_modifier,
                             const float* viewmatrix, const float* projmatrix, const float* campos,
                             float tanfovx, float tanfovy, int H, int W, int* radii, void* geom,
                             size_t geom_bytes_, long long* num_rendered_host, void* stream_) {
    SYN3R_REQUIRE(SYN3R_DIM_OK(N) && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W), "raster_preprocess: bad sizes N=%d H=%d W=%d", N, H, W);
    SYN3R_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "raster_preprocess: sh_degree %d not in 0..3", sh_degree);
    SYN3R_REQUIRE(sh_coeffs >= (sh_degree + 1) * (sh_degree + 1) && sh_coeffs <= 1024,
                  "raster_preprocess: sh_degree %d needs >= %d coefficients, got %d", sh_degree,
                  (sh_degree + 1) * (sh_degree + 1), sh_coeffs);
    SYN3R_REQUIRE(means3D && scales && rotations && opacities && shs && viewmatrix && projmatrix && campos && radii,
                  "raster_preprocess: null argument");
    SYN3R_REQUIRE(tanfovx > 0 && tanfovy > 0, "raster_preprocess: bad field of view");
    if (!geom || geom_bytes_ < geom_bytes(N)) {
        set_error("raster_preprocess: geometry buffer %zu < %zu", geom_bytes_, geom_bytes(N));
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GeomState g = carve_geom(geom, N);
    Camera cam;
    fill_camera(cam, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W);
    SYN3R_LAUNCH(k_preprocess, dim3(ceil_div(N, 256)), dim3(256), 0, stream, N, sh_degree, sh_coeffs, means3D,
                       scales, rotations, opacities, shs, confidence, scale_modifier, cam, radii, g, raw);
    int rc = SYN3R_OK;
    const bool hier = hier_binning(N, cam.grid_x, cam.grid_y);
    if (!hier) {
        // the pair-sort path: Gaussians by ascending depth (stable: equal depths keep index order), then the tile counts scanned in
        // that order: pairs emitted along it and stably sorted by tile id end up ordered exactly like the
        // published (tile << 32 | depth bits) key sort, for 8 B instead of 72 B of sort traffic per pair
        int in_b = 0;
        rc = argsort_depth_u32(g.dkeys_a, g.order_a, g.dkeys_b, g.order_b, (size_t)N, g.sort_scratch, stream, &in_b);
        if (rc) return rc;
        if ((in_b ? g.order_b : g.order_a) != g.order) { set_error("raster_preprocess: unexpected argsort parity"); return SYN3R_E_INVALID; }
        rc = exclusive_scan_u32(g.tiles_touched, g.point_offsets, (size_t)N, g.header, g.scan_scratch, stream, g.order);
        if (rc) return rc;
    } else if (num_rendered_host) {
        // The hierarchical binning of syn3r_raster_render orders and counts on its own (k_super_sort, k_tile_offsets); a caller who
        // asks for the exact pair count before sizing the binning buffer gets the sum of the tile counts
        rc = exclusive_scan_u32(g.tiles_touched, g.point_offsets, (size_t)N, g.header, g.scan_scratch, stream);
        if (rc) return rc;
    }
    SYN3R_LAUNCH_CHECK("raster_preprocess launch");
    if (num_rendered_host) {
        unsigned total = 0;
        rc = check_hip(hipMemcpyAsync(&total, g.header, 4, hipMemcpyDeviceToHost, stream), "num_rendered copy");
        if (rc) return rc;
        rc = check_hip(hipStreamSynchronize(stream), "num_rendered sync");
        if (rc) return rc;
        *num_rendered_host = (long long)total;
    }
    return SYN3R_OK;
}

extern "C" int syn3r_raster_preprocess(int N, int sh_degree, int sh_coeffs, const float* means3D,
                                       const float* scales, const float* rotations, const float* opacities,
                                       const float* shs, const float* confidence, float scale_modifier,
                                       const float* viewmatrix, const float* projmatrix, const float* campos,
                                       float tanfovx, float tanfovy, int H, int W, int* radii, void* geom,
                                       size_t geom_bytes_, long long* num_rendered_host, void* stream_) {
    return raster_preprocess(0, N, sh_degree, sh_coeffs, means3D, scales, rotations, opacities, shs, confidence, scale_modifier,
                             viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, radii, geom, geom_bytes_, num_rendered_host,
                             stream_);
}

extern "C" int syn3r_raster_preprocess_raw(int N, int sh_degree, int sh_coeffs, const float* means3D,
                                           const float* log_scales, const float* raw_rotations, const float* opacity_logits,
                                           const float* shs, const float* confidence, float scale_modifier,
                                           const float* viewmatrix, const float* projmatrix, const float* campos,
                                           float tanfovx, float tanfovy, int H, int W, int* radii, void* geom,
                                           size_t geom_bytes_, long long* num_rendered_host, void* stream_) {
    return raster_preprocess(1, N, sh_degree, sh_coeffs, means3D, log_scales, raw_rotations, opacity_logits, shs, confidence,
                             scale_modifier, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, radii, geom, geom_bytes_,
                             num_rendered_host, stream_);
}

extern "C" int syn3r_raster_render(int N, int H, int W, const float* bg, const int* radii, void* geom,
                                   size_t geom_bytes_, void* binning, size_t binning_bytes_, void* image,
                                   size_t image_bytes_, long long P, float* out_color, float* out_depth,
                                   float* out_alpha, unsigned** point_list_out, void* stream_) {
    SYN3R_REQUIRE(N > 0 && H > 0 && W > 0 && P >= 0 && P < (1ll << 31), "raster_render: bad sizes N=%d H=%d W=%d P=%lld",
                  N, H, W, P);
    SYN3R_REQUIRE(bg && radii && out_color && out_depth && out_alpha, "raster_render: null argument");
    if (!geom || geom_bytes_ < geom_bytes(N) || !image || image_bytes_ < image_bytes(H, W) || !binning ||
        binning_bytes_ < binning_bytes(P)) {
        set_error("raster_render: state buffer too small (geom %zu/%zu image %zu/%zu binning %zu/%zu)", geom_bytes_,
                  geom_bytes(N), image_bytes_, image_bytes(H, W), binning_bytes_, binning_bytes(P));
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GeomState g = carve_geom(geom, N);
    ImageState im = carve_image(image, H, W);
    BinningState bn = carve_binning(binning, P);
    const int gx = (W + kTileX - 1) / kTileX, gy = (H + kTileY - 1) / kTileY;
    const size_t tiles = (size_t)gx * gy;
    int rc = SYN3R_OK;
    if (P == 0) {
        rc = check_hip(hipMemsetAsync(im.ranges, 0, tiles * 8, stream), "memset ranges");
        if (rc) return rc;
    }
    unsigned* point_list = bn.vals_a;
    unsigned* tile_order = nullptr;
    if (P > 0 && hier_binning(N, gx, gy)) {
        SYN3R_REQUIRE(P < (1ll << 30), "raster_render: pair capacity %lld: the super-tile sort indexes a list with 31-bit integers (2^30 pairs at most)", P);
        const BinPlan bp = bin_plan(N, gx, gy);
        unsigned* counters = im.bin_counters;
        unsigned* sid = bn.vals_b;
        unsigned short* srect = (unsigned short*)bn.keys_a;
        unsigned* sstart = counters + kBinCounters;              // [nsuper + 1] list starts
        unsigned long long* skeys = bn.keys_b;
        static DevOnce once;
        if (int rc2 = set_max_lds(once, (const void*)k_super_sort, kSortLds * 8, "hipFuncSetAttribute(super_sort)")) return rc2;
        SYN3R_LAUNCH(k_super_count, dim3(bp.nchunks), dim3(kBinThreads), 0, stream, N, (const float*)g.means2D, radii, gx, gy, bp.ss, bp.sgx,
                     bp.nsuper, bp.rounds, bp.nchunks, counters, g.header);
        SYN3R_LAUNCH(k_super_append, dim3(bp.nchunks), dim3(kBinThreads), 0, stream, N, (const float*)g.depths, (const float*)g.means2D, radii,
                     gx, gy, bp.ss, bp.sgx, bp.nsuper, bp.rounds, bp.nchunks, (const unsigned*)counters, sstart, (unsigned)P, skeys, g.header);
        SYN3R_LAUNCH(k_super_sort, dim3(bp.nsuper), dim3(kSuperSortThreads), kSortLds * 8, stream, N, (const float*)g.means2D, radii, gx, gy,
                     bp.ss, bp.sgx, (const unsigned*)sstart, (unsigned)P, skeys, sid, srect, im.tile_counts);
        // the blend kernels take the tiles longest list first: with one block per tile in image order they ended on the few long
        // tiles of the last dispatch round (k_render 170 -> 130 us, k_render_bwd 463 -> 380 us at 200 000 Gaussians / 1080p;
        // SYN3R_TILE_ORDER=0 in tuning builds restores the image order)
        static const int order_env = tune_env("SYN3R_TILE_ORDER", 1);
        tile_order = order_env ? im.tile_order : nullptr;
        SYN3R_LAUNCH(k_tile_offsets, dim3(1), dim3(kOffThreads), 0, stream, (int)tiles, (const unsigned*)im.tile_counts, (unsigned)P,
                     im.ranges, g.header, tile_order);
        SYN3R_LAUNCH(k_tile_write, dim3((unsigned)tiles), dim3(256), 0, stream, gx, (int)tiles, bp.ss, bp.sgx,
                     (const unsigned*)sstart, (unsigned)P, (const unsigned*)sid, (const unsigned short*)srect,
                     (const uint2*)im.ranges, (const unsigned*)im.tile_counts, point_list);
    } else if (P > 0) {
        // P is the pair CAPACITY of the binning buffer; the live count is read from the geometry header on
        // the device, so the caller may pass an estimate and skip the device->host read of the exact count
        unsigned* tk_a = (unsigned*)bn.keys_a;
        unsigned* tk_b = (unsigned*)bn.keys_b;
        SYN3R_LAUNCH(k_dup_tiles, dim3(ceil_div(N, 256)), dim3(256), 0, stream, N, g.order, g.means2D, radii,
                           g.point_offsets, gx, gy, tk_a, bn.vals_a, (unsigned)P, g.header, im.ranges, (int)tiles);
        int in_b = 0;
        rc = sort_pairs_by_tile_u32(tk_a, bn.vals_a, tk_b, bn.vals_b, (size_t)P, bits_for((unsigned)tiles - 1),
                                    bn.sort_scratch, stream, &in_b, g.header);
        if (rc) return rc;
        const unsigned* keys = in_b ? tk_b : tk_a;
        point_list = in_b ? bn.vals_b : bn.vals_a;
        SYN3R_LAUNCH(k_tile_ranges, dim3(ceil_div(P, 256)), dim3(256), 0, stream, P, g.header, keys, im.ranges);
    }
    SYN3R_LAUNCH(k_render, dim3((unsigned)tiles), dim3(kFwdThreads), 0, stream, H, W, gx, gy, im.ranges, point_list,
                       g.splats, bg[0], bg[1], bg[2], im.n_contrib, im.final_T, out_color, out_depth, out_alpha, (const unsigned*)tile_order);
    SYN3R_LAUNCH_CHECK("raster_render launch");
    if (point_list_out) *point_list_out = point_list;
    return SYN3R_OK;
}

#ifdef SYN3R_RASTER_STATS
extern "C" __attribute__((visibility("default"))) int syn3r_debug_fwd_stats(unsigned long long* out4, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_fwd_stats), sizeof(unsigned long long) * 4);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stats), z, sizeof(z)); }
    return rc;
}
#endif
