// Shared declarations of the Gaussian rasteriser (forward, backward, sort).
//
// The reference's rasteriser (diff-gaussian-rasterization-confidence inside the
// un-vendored thirdparty/FSGS submodule; call sites model/diffusionGS.py:154,166
// and :139,1640) is NOT in /root/reference: this implementation
// restates the published 3DGS algorithm (Kerbl et al. 2023, "3D Gaussian
// Splatting for Real-Time Radiance Field Rendering", §4-§6 and appendix) with
// the depth / alpha outputs and the per-Gaussian confidence factor the call
// sites require.  Constants the published implementation fixes are named here so
// they can be matched against the CUDA build if it is ever supplied.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace syn3r {

constexpr int kTileX = 16;
constexpr int kTileY = 16;
constexpr int kTilePix = kTileX * kTileY;       // 256 threads = 4 wavefronts
constexpr float kNearClip = 0.2f;               // view-space z cull
constexpr float kFovGuard = 1.3f;               // clamp of t.x/t.z in the EWA Jacobian
constexpr float kLowPass = 0.3f;                // added to the 2D covariance diagonal
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTransmittanceMin = 1e-4f;

// Per-Gaussian screen-space record gathered by the blend kernels: 3 x 16-byte loads.
struct alignas(16) Splat {
    float x, y;            // pixel-space mean
    float cxx, cxy, cyy;   // conic (inverse 2D covariance)
    float opacity;         // opacity * confidence
    float r, g, b;         // view-dependent colour (SH evaluated, clamped)
    float depth;           // view-space z
    float pad0, pad1;
};
static_assert(sizeof(Splat) == 48, "Splat must be 48 bytes");

// Geometry state carved from the caller's buffer (all arrays 256-byte aligned).
struct GeomState {
    unsigned* header;        // [0] = number of (Gaussian, tile) pairs (the EXACT count, [3], when the lists were clipped), [1] = 1 if a render ran out of pair capacity, [2] = entries of the super-tile lists, [3] = sum of the tile rectangles' areas (hierarchical binning)
    float* depths;           // [N]
    float* means2D;          // [N,2]
    float* cov3D;            // [N,6]
    float* conic_opacity;    // [N,4] conic + raw opacity (without confidence)
    float* rgb;              // [N,3]
    unsigned* clamped;       // [N] bit c set if colour channel c was clamped at 0
    unsigned* tiles_touched; // [N]
    unsigned* point_offsets; // [N] exclusive scan of tiles_touched IN DEPTH ORDER (entry i belongs to order[i])
    Splat* splats;           // [N]
    unsigned* dkeys_a;       // [N] depth bits of the visible Gaussians (0xFFFFFFFF otherwise), sort ping
    unsigned* dkeys_b;       // [N] sort pong
    unsigned* order_a;       // [N] Gaussian ids by ascending depth (stable), sort ping
    unsigned* order_b;       // [N] sort pong
    unsigned* order;         // whichever of order_a / order_b holds the result (4 passes: order_a)
    void* sort_scratch;      // argsort histograms
    void* scan_scratch;
};

#ifndef SYN3R_BIN_COUNTERS
#define SYN3R_BIN_COUNTERS 65536
#endif
constexpr size_t kBinCounters = SYN3R_BIN_COUNTERS;          // (chunk, super-tile) counters of the hierarchical binning
constexpr size_t kMaxSuperSlots = 4096;         // ... followed by the super-tile list starts (at most kMaxSuper + 1)
struct ImageState {
    uint2* ranges;           // [tiles] (start, end) into the sorted pair list
    unsigned* n_contrib;     // [H*W] index (1-based, within the tile list) of the last contributor
    float* final_T;          // [H*W]
    unsigned* tile_counts;   // [tiles][4] list length per tile and quarter of its super-tile's list (hierarchical binning)
    unsigned* bin_counters;  // [kBinCounters + kMaxSuperSlots] per (chunk, super-tile) counts, then the list starts
    unsigned* tile_order;    // [tiles] tiles by descending list length
};

struct BinningState {
    unsigned long long* keys_a;   // the rasteriser stores u32 tile ids here (half of each array is used)
    unsigned long long* keys_b;
    unsigned* vals_a;
    unsigned* vals_b;
    void* sort_scratch;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t scan_scratch_bytes(size_t n);
size_t sort_scratch_bytes(size_t n);
int exclusive_scan_u32(const unsigned* in, unsigned* out, size_t n, unsigned* total_out, void* scratch,
                       hipStream_t stream, const unsigned* perm = nullptr);   // perm: scan in[perm[i]]
// stable argsort of n u32 keys (vals_a is NOT read: element i carries i); 4 passes of 8 bits
int argsort_depth_u32(unsigned* keys_a, unsigned* vals_a, unsigned* keys_b, unsigned* vals_b, size_t n, void* scratch,
                      hipStream_t stream, int* result_in_b);
// stable sort of (u32 tile id, u32 Gaussian id) pairs on the low nbits of the key; 7-bit digits
int sort_pairs_by_tile_u32(unsigned* keys_a, unsigned* vals_a, unsigned* keys_b, unsigned* vals_b, size_t n, int nbits,
                           void* scratch, hipStream_t stream, int* result_in_b, const unsigned* n_dev);
int radix_sort_pairs(unsigned long long* keys_a, unsigned* vals_a, unsigned long long* keys_b, unsigned* vals_b,
                     size_t n, int nbits, void* scratch, hipStream_t stream, int* result_in_b,
                     const unsigned* n_dev = nullptr);

size_t geom_bytes(int N);
bool raster_tiles_ordered(int N, int gx, int gy);   // did syn3r_raster_render leave ImageState::tile_order for this shape?
size_t image_bytes(int H, int W);
size_t binning_bytes(long long P);
GeomState carve_geom(void* buf, int N);
ImageState carve_image(void* buf, int H, int W);
BinningState carve_binning(void* buf, long long P);

#ifdef __HIPCC__
// Conservative reach test used by the blend kernels to build per-wavefront visit lists: can the splat
// (pixel mean (mx,my), conic (A,B,C), opacity op) reach alpha >= kAlphaMin at ANY pixel centre of the
// rectangle [x0,x1] x [y0,y1]?  alpha = op * exp(-q/2) with q the conic's quadratic form, so the question is
// whether min q over the rectangle is <= 2 ln(255 op).  q is convex with its minimum at the mean: inside the
// rectangle the minimum is 0, otherwise it lies on an edge facing the mean, where q is a 1-D parabola whose
// clamped vertex is closed-form.  A margin keeps every pair the per-pixel test could accept despite rounding;
// pairs rejected here contribute exactly nothing in the unculled traversal, so results are unchanged.
__device__ __forceinline__ bool splat_reaches_rect(float mx, float my, float A, float B, float C, float op, float x0,
                                                   float x1, float y0, float y1) {
    if (!(A > 0.0f && C > 0.0f)) return true;              // not a proper conic: let the per-pixel test decide
    const float thr = 2.0f * __logf(255.0f * op);          // q must not exceed this (NaN / negative: unreachable)
    const float dx0 = x0 - mx, dx1 = x1 - mx, dy0 = y0 - my, dy1 = y1 - my;
    const bool inx = dx0 <= 0.0f && dx1 >= 0.0f, iny = dy0 <= 0.0f && dy1 >= 0.0f;
    float qmin = 0.0f;
    if (!(inx && iny)) {
        qmin = 3.0e38f;
        if (!inx) {                                        // the vertical edge facing the mean
            const float dx = dx0 > 0.0f ? dx0 : dx1;
            const float dy = fminf(fmaxf(-B * dx / C, dy0), dy1);
            qmin = A * dx * dx + 2.0f * B * dx * dy + C * dy * dy;
        }
        if (!iny) {                                        // the horizontal edge facing the mean
            const float dy = dy0 > 0.0f ? dy0 : dy1;
            const float dx = fminf(fmaxf(-B * dy / A, dx0), dx1);
            qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
        }
    }
    return qmin <= thr + (1.0e-3f + 1.0e-4f * thr);
}
#endif

// camera passed by value to kernels
struct Camera {
    float view[16];     // world->view, column-major (x' = v[0]x + v[4]y + v[8]z + v[12])
    float proj[16];     // world->clip, column-major
    float campos[3];
    float tanfovx, tanfovy, focal_x, focal_y;
    int H, W, grid_x, grid_y;
};

}  // namespace syn3r
