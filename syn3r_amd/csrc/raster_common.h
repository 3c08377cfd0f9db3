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
    unsigned* header;        // [0] = number of (Gaussian, tile) pairs, [1] = 1 if a render ran out of pair capacity
    float* depths;           // [N]
    float* means2D;          // [N,2]
    float* cov3D;            // [N,6]
    float* conic_opacity;    // [N,4] conic + raw opacity (without confidence)
    float* rgb;              // [N,3]
    unsigned* clamped;       // [N] bit c set if colour channel c was clamped at 0
    unsigned* tiles_touched; // [N]
    unsigned* point_offsets; // [N] exclusive scan of tiles_touched
    Splat* splats;           // [N]
    void* scan_scratch;
};

struct ImageState {
    uint2* ranges;           // [tiles] (start, end) into the sorted pair list
    unsigned* n_contrib;     // [H*W] index (1-based, within the tile list) of the last contributor
    float* final_T;          // [H*W]
};

struct BinningState {
    unsigned long long* keys_a;
    unsigned long long* keys_b;
    unsigned* vals_a;
    unsigned* vals_b;
    void* sort_scratch;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t scan_scratch_bytes(size_t n);
size_t sort_scratch_bytes(size_t n);
int exclusive_scan_u32(const unsigned* in, unsigned* out, size_t n, unsigned* total_out, void* scratch,
                       hipStream_t stream);
int radix_sort_pairs(unsigned long long* keys_a, unsigned* vals_a, unsigned long long* keys_b, unsigned* vals_b,
                     size_t n, int nbits, void* scratch, hipStream_t stream, int* result_in_b,
                     const unsigned* n_dev = nullptr);

size_t geom_bytes(int N);
size_t image_bytes(int H, int W);
size_t binning_bytes(long long P);
GeomState carve_geom(void* buf, int N);
ImageState carve_image(void* buf, int H, int W);
BinningState carve_binning(void* buf, long long P);

// camera passed by value to kernels
struct Camera {
    float view[16];     // world->view, column-major (x' = v[0]x + v[4]y + v[8]z + v[12])
    float proj[16];     // world->clip, column-major
    float campos[3];
    float tanfovx, tanfovy, focal_x, focal_y;
    int H, W, grid_x, grid_y;
};

}  // namespace syn3r
