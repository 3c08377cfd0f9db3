// Geometry kernels: fused inverse-warp + reprojection consistency, reprojection
// error alone, and the depth-weighted forward bilinear splat.
//
// Reference behaviour restated (not copied) from
//   solver_utils/forward_warp.py:187-279  inverse_warp
//   solver_utils/consistency.py:6-91      consistency_check_with_depth
//   solver_utils/forward_warp.py:7-182    forward_warp / bilinear_splatting
//
// All three are HBM/latency-bound gathers and scatters: one lane per pixel,
// coalesced row-major reads of the per-pixel inputs, gathers served by L2.
// Arithmetic follows the reference's fp32 (inverse warp) / fp64 (forward
// splat) operation order; this file is compiled with -ffp-contract=off so the
// compiler does not fuse what torch's elementwise kernels keep separate.
#include "common.h"

using namespace syn3r;

namespace {

constexpr int kBlock = 256;

// ordered-uint encoding of a float so that unsigned compare == float compare
__device__ __forceinline__ unsigned enc_f32(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float dec_f32(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// y = M[0:3,:] * (p,1) with an fma chain in k order (what a BLAS micro-kernel does)
__device__ __forceinline__ void xform4(const Mat4f& M, float x, float y, float z, float w, float& ox,
                                       float& oy, float& oz, float& ow) {
    const float* m = M.m;
    ox = fmaf(m[3], w, fmaf(m[2], z, fmaf(m[1], y, m[0] * x)));
    oy = fmaf(m[7], w, fmaf(m[6], z, fmaf(m[5], y, m[4] * x)));
    oz = fmaf(m[11], w, fmaf(m[10], z, fmaf(m[9], y, m[8] * x)));
    ow = fmaf(m[15], w, fmaf(m[14], z, fmaf(m[13], y, m[12] * x)));
}
__device__ __forceinline__ void xform3(const Mat3f& M, float x, float y, float z, float& ox, float& oy,
                                       float& oz) {
    const float* m = M.m;
    ox = fmaf(m[2], z, fmaf(m[1], y, m[0] * x));
    oy = fmaf(m[5], z, fmaf(m[4], y, m[3] * x));
    oz = fmaf(m[8], z, fmaf(m[7], y, m[6] * x));
}

// torch grid_sampler_unnormalize, align_corners=False
__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.0f) * (float)size - 1.0f) / 2.0f; }

// bilinear tap with zeros padding
__device__ __forceinline__ float tap(const float* __restrict__ img, int H, int W, int y, int x) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

// consistency.py:44-91 for one pixel. depth1 value d1 at (r,c); returns error.
__device__ __forceinline__ float reproj_pixel(float d1, int r, int c, const float* __restrict__ depth2, int H,
                                              int W, const Mat4f& T12, const Mat4f& T21, const Mat3f& K1,
                                              const Mat3f& K1inv, const Mat3f& K2) {
    // get_points_from_depth (:6-24): inv(K) @ (x, y, 1) * depth
    float px, py, pz;
    xform3(K1inv, (float)c, (float)r, 1.0f, px, py, pz);
    px *= d1; py *= d1; pz *= d1;
    // transform_points (:26-42)
    float ax, ay, az, aw;
    xform4(T12, px, py, pz, 1.0f, ax, ay, az, aw);
    ax /= aw; ay /= aw; az /= aw;
    // project with intrinsics2 (:62-63)
    float ix, iy, iz;
    xform3(K2, ax, ay, az, ix, iy, iz);
    ix /= iz; iy /= iz;
    // the reference's own normalisation (:66-68) fed to the default grid_sample
    float gx = ix / ((float)(W - 1) / 2.0f) - 1.0f;
    float gy = iy / ((float)(H - 1) / 2.0f) - 1.0f;
    float sx = unnorm(gx, W), sy = unnorm(gy, H);
    float fx0 = floorf(sx), fy0 = floorf(sy);
    int x0 = (int)fx0, y0 = (int)fy0;
    float wx1 = sx - fx0, wy1 = sy - fy0;
    float wx0 = (fx0 + 1.0f) - sx, wy0 = (fy0 + 1.0f) - sy;
    float d12 = 0.0f;
    if (isfinite(sx) && isfinite(sy)) {
        d12 += tap(depth2, H, W, y0, x0) * (wx0 * wy0);
        d12 += tap(depth2, H, W, y0, x0 + 1) * (wx1 * wy0);
        d12 += tap(depth2, H, W, y0 + 1, x0) * (wx0 * wy1);
        d12 += tap(depth2, H, W, y0 + 1, x0 + 1) * (wx1 * wy1);
    } else {
        d12 = 0.0f;
    }
    // lift back with the sampled depth (:74-75)
    float bx = ax / az * d12, by = ay / az * d12, bz = az / az * d12;
    float cx, cy, cz, cw;
    xform4(T21, bx, by, bz, 1.0f, cx, cy, cz, cw);
    cx /= cw; cy /= cw; cz /= cw;
    float jx, jy, jz;
    xform3(K1, cx, cy, cz, jx, jy, jz);
    jx /= jz; jy /= jz;
    float ex = jx - (float)c, ey = jy - (float)r;
    return sqrtf(ex * ex + ey * ey);
}

struct IwParams {
    Mat4f pose12, pose21;
    Mat3f K, Kinv;
    float fx, fy, cx, cy, bandwidth;
    int H, W;
};

// forward_warp.py:202-224: where target pixel (r,c) lands in the source view
__device__ __forceinline__ void iw_project(const IwParams& p, float z, int r, int c, float& x2, float& y2) {
    float x = ((float)c - p.cx) / p.fx;
    float y = ((float)r - p.cy) / p.fy;
    float X, Y, Z, Wd;
    xform4(p.pose12, x * z, y * z, 1.0f * z, 1.0f, X, Y, Z, Wd);
    x2 = p.fx * X / Z + p.cx;
    y2 = p.fy * Y / Z + p.cy;
}

// nearest grid_sample index (forward_warp.py:225-228); -1 if outside
__device__ __forceinline__ int iw_nearest(const IwParams& p, float x2, float y2) {
    float gx = 2.0f * x2 / (float)p.W - 1.0f;
    float gy = 2.0f * y2 / (float)p.H - 1.0f;
    float sx = nearbyintf(unnorm(gx, p.W));
    float sy = nearbyintf(unnorm(gy, p.H));
    if (!(sx >= 0.0f && sx < (float)p.W && sy >= 0.0f && sy < (float)p.H)) return -1;
    return (int)sy * p.W + (int)sx;
}

// pass 0: global max of warped depth and min of its positive part (1e4 sentinel), forward_warp.py:236-240
__global__ void __launch_bounds__(kBlock) k_iw_minmax(IwParams p, const float* __restrict__ depth,
                                                      const float* __restrict__ depth_pseudo,
                                                      unsigned* __restrict__ mm) {
    const int n = p.H * p.W;
    float vmax = -INFINITY, vmin = INFINITY;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int r = i / p.W, c = i - r * p.W;
        float x2, y2;
        iw_project(p, depth_pseudo[i], r, c, x2, y2);
        int src = iw_nearest(p, x2, y2);
        float wd = src >= 0 ? depth[src] : 0.0f;
        vmax = fmaxf(vmax, wd);
        vmin = fminf(vmin, wd > 0.0f ? wd : 1e4f);
    }
    vmax = wave_max(vmax);
    vmin = wave_min(vmin);
    __shared__ float smax[kBlock / 64], smin[kBlock / 64];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { smax[wv] = vmax; smin[wv] = vmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) { vmax = fmaxf(vmax, smax[k]); vmin = fminf(vmin, smin[k]); }
        atomicMax(&mm[0], enc_f32(vmax));
        atomicMin(&mm[1], enc_f32(vmin));
    }
}

__global__ void __launch_bounds__(kBlock) k_iw_main(
    IwParams p, const float* __restrict__ img, const float* __restrict__ depth,
    const float* __restrict__ depth_pseudo, const unsigned* __restrict__ mm, float* __restrict__ warped_img,
    float* __restrict__ warped_depth, uint8_t* __restrict__ mask_warp, uint8_t* __restrict__ mask_depth,
    uint8_t* __restrict__ mask, float* __restrict__ warped_masked_img, uint8_t* __restrict__ mask_inv,
    uint8_t* __restrict__ mask_depth_strict, uint8_t* __restrict__ mask_reproj,
    float* __restrict__ soft_mask_reproj, float* __restrict__ reproj_error) {
    const int n = p.H * p.W;
    const float dmax = dec_f32(mm[0]);
    const float dmin = dec_f32(mm[1]);
    const float range = dmax - dmin;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int r = i / p.W, c = i - r * p.W;
        float z = depth_pseudo[i];
        float x2, y2;
        iw_project(p, z, r, c, x2, y2);
        int src = iw_nearest(p, x2, y2);
        float wr = 0.f, wg = 0.f, wb = 0.f, wd = 0.f;
        if (src >= 0) {
            wr = img[src];
            wg = img[(size_t)n + src];
            wb = img[2 * (size_t)n + src];
            wd = depth[src];
        }
        bool mw = (x2 >= 0.0f) && (x2 < (float)p.W) && (y2 >= 0.0f) && (y2 < (float)p.H);
        bool pos = wd > 0.0f;
        // forward_warp.py:241-251
        float nwd = pos ? (wd - dmin) / range : 0.0f;
        float wd_out = pos ? wd : 0.0f;
        float npd = (z - dmin) / range;
        float ad = fabsf(nwd - npd);
        bool md = ad < 0.3f;
        bool mds = ad < 0.1f;
        bool m = mw && md;
        // forward_warp.py:257-266 (depth1 = depth_pseudo viewed from pose2, depth2 = source depth)
        float err = reproj_pixel(z, r, c, depth, p.H, p.W, p.pose12, p.pose21, p.K, p.Kinv, p.K);
        bool mr = (err < p.bandwidth) && mw;
        float q = err / p.bandwidth;
        float soft = expf(-(q * q * q));

        warped_img[i] = wr;
        warped_img[(size_t)n + i] = wg;
        warped_img[2 * (size_t)n + i] = wb;
        warped_depth[i] = wd_out;
        mask_warp[i] = mw;
        mask_depth[i] = md;
        mask[i] = m;
        float mf = m ? 1.0f : 0.0f;
        warped_masked_img[i] = wr * mf;
        warped_masked_img[(size_t)n + i] = wg * mf;
        warped_masked_img[2 * (size_t)n + i] = wb * mf;
        mask_inv[i] = !m;
        mask_depth_strict[i] = mds;
        mask_reproj[i] = mr;
        soft_mask_reproj[i] = soft;
        if (reproj_error) reproj_error[i] = err;
    }
}

struct RpParams {
    Mat4f T12, T21;
    Mat3f K1, K1inv, K2;
    int H, W;
};

__global__ void __launch_bounds__(kBlock) k_reproj(RpParams p, const float* __restrict__ depth1,
                                                   const float* __restrict__ depth2, float* __restrict__ err) {
    const int n = p.H * p.W;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int r = i / p.W, c = i - r * p.W;
        err[i] = reproj_pixel(depth1[i], r, c, depth2, p.H, p.W, p.T12, p.T21, p.K1, p.K1inv, p.K2);
    }
}

// ---------------------------------------------------------------- forward splat (fp64)

struct FwParams {
    Mat4d T;
    Mat3d K1inv, K2;
    int H, W;
};

// compute_transformed_points (forward_warp.py:7-38) for one pixel
__device__ __forceinline__ void fw_point(const FwParams& p, double d, int r, int c, double& u, double& v,
                                         double& tz) {
    const double* ki = p.K1inv.m;
    double x = (double)c, y = (double)r;
    double ux = ki[0] * x + ki[1] * y + ki[2] * 1.0;
    double uy = ki[3] * x + ki[4] * y + ki[5] * 1.0;
    double uz = ki[6] * x + ki[7] * y + ki[8] * 1.0;
    double wx = d * ux, wy = d * uy, wz = d * uz;
    const double* t = p.T.m;
    double tx = t[0] * wx + t[1] * wy + t[2] * wz + t[3] * 1.0;
    double ty = t[4] * wx + t[5] * wy + t[6] * wz + t[7] * 1.0;
    double tzz = t[8] * wx + t[9] * wy + t[10] * wz + t[11] * 1.0;
    const double* k2 = p.K2.m;
    double nx = k2[0] * tx + k2[1] * ty + k2[2] * tzz;
    double ny = k2[3] * tx + k2[4] * ty + k2[5] * tzz;
    double nz = k2[6] * tx + k2[7] * ty + k2[8] * tzz;
    u = nx / nz;
    v = ny / nz;
    tz = nz;
}

__device__ __forceinline__ double clipd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// pass 0: max over the image of clip(trans_depth, 0, 5000); log(1+.) is monotone so
// max(log(1+s)) == log(1+max s) (forward_warp.py:83-85)
__global__ void __launch_bounds__(kBlock) k_fw_maxdepth(FwParams p, const double* __restrict__ depth1,
                                                        unsigned long long* __restrict__ maxbits) {
    const int n = p.H * p.W;
    double vmax = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int r = i / p.W, c = i - r * p.W;
        double u, v, tz;
        fw_point(p, depth1[i], r, c, u, v, tz);
        vmax = fmax(vmax, clipd(tz, 0.0, 5000.0));
    }
    vmax = wave_max_d(vmax);
    __shared__ double smax[kBlock / 64];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) smax[wv] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) vmax = fmax(vmax, smax[k]);
        atomicMax(maxbits, (unsigned long long)__double_as_longlong(vmax));  // non-negative doubles order as u64
    }
}

__device__ __forceinline__ void splat4(double* __restrict__ acc, int W2, int y, int x, double w, double r,
                                       double g, double b) {
    double* a = acc + ((size_t)y * W2 + x) * 4;
    unsafeAtomicAdd(a + 0, r * w);
    unsafeAtomicAdd(a + 1, g * w);
    unsafeAtomicAdd(a + 2, b * w);
    unsafeAtomicAdd(a + 3, w);
}

__global__ void __launch_bounds__(kBlock) k_fw_splat(FwParams p, const double* __restrict__ frame1,
                                                     const uint8_t* __restrict__ mask1,
                                                     const double* __restrict__ depth1,
                                                     const unsigned long long* __restrict__ maxbits,
                                                     double* __restrict__ acc, double* __restrict__ flow12) {
    const int n = p.H * p.W;
    const int W2 = p.W + 2;
    const double logmax = log(1.0 + __longlong_as_double((long long)*maxbits));
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int r = i / p.W, c = i - r * p.W;
        double u, v, tz;
        fw_point(p, depth1[i], r, c, u, v, tz);
        double fx = u - (double)c, fy = v - (double)r;  // flow12 (:176-177)
        flow12[2 * (size_t)i] = fx;
        flow12[2 * (size_t)i + 1] = fy;
        // bilinear_splatting (:58-81)
        double ox = (fx + (double)c) + 1.0, oy = (fy + (double)r) + 1.0;
        double flx = floor(ox), fly = floor(oy), clx = ceil(ox), cly = ceil(oy);
        // astype(int) of a non-finite value is INT_MIN in numpy; clip sends it to 0
        int x0 = isfinite(ox) ? (int)clipd(flx, 0.0, (double)(p.W + 1)) : 0;
        int y0 = isfinite(oy) ? (int)clipd(fly, 0.0, (double)(p.H + 1)) : 0;
        int x1 = isfinite(ox) ? (int)clipd(clx, 0.0, (double)(p.W + 1)) : 0;
        int y1 = isfinite(oy) ? (int)clipd(cly, 0.0, (double)(p.H + 1)) : 0;
        ox = clipd(ox, 0.0, (double)(p.W + 1));
        oy = clipd(oy, 0.0, (double)(p.H + 1));
        double pnw = (1.0 - (oy - (double)y0)) * (1.0 - (ox - (double)x0));
        double psw = (1.0 - ((double)y1 - oy)) * (1.0 - (ox - (double)x0));
        double pne = (1.0 - (oy - (double)y0)) * (1.0 - ((double)x1 - ox));
        double pse = (1.0 - ((double)y1 - oy)) * (1.0 - ((double)x1 - ox));
        double sat = clipd(tz, 0.0, 5000.0);
        double dw = exp(log(1.0 + sat) / logmax * 50.0);
        double m = mask1 ? (double)(mask1[i] != 0) : 1.0;
        double wnw = pnw * m * 1.0 / dw, wsw = psw * m * 1.0 / dw;
        double wne = pne * m * 1.0 / dw, wse = pse * m * 1.0 / dw;
        double cr = frame1[3 * (size_t)i], cg = frame1[3 * (size_t)i + 1], cb = frame1[3 * (size_t)i + 2];
        splat4(acc, W2, y0, x0, wnw, cr, cg, cb);
        splat4(acc, W2, y1, x0, wsw, cr, cg, cb);
        splat4(acc, W2, y0, x1, wne, cr, cg, cb);
        splat4(acc, W2, y1, x1, wse, cr, cg, cb);
    }
}

// crop, normalise, clip, round-half-even, cast (forward_warp.py:109-126)
__global__ void __launch_bounds__(kBlock) k_fw_finish(int H, int W, const double* __restrict__ acc,
                                                      uint8_t* __restrict__ warped, uint8_t* __restrict__ mask2) {
    const int n = H * W;
    const int W2 = W + 2;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int r = i / W, c = i - r * W;
        const double* a = acc + ((size_t)(r + 1) * W2 + (c + 1)) * 4;
        double w = a[3];
        bool m = w > 0.0;
        mask2[i] = m;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double v = m ? a[k] / w : 0.0;
            v = clipd(v, 0.0, 255.0);
            warped[3 * (size_t)i + k] = (uint8_t)rint(v);
        }
    }
}

inline int grid_for(int n) {
    int g = ceil_div(n, kBlock);
    return g > 4096 ? 4096 : (g < 1 ? 1 : g);
}

void load3(Mat3f& d, const float* s) { for (int i = 0; i < 9; ++i) d.m[i] = s[i]; }
void load4(Mat4f& d, const float* s) { for (int i = 0; i < 16; ++i) d.m[i] = s[i]; }


// ------------------------------------------------------------------------------------------------
// Orchestrator post-processing on the device (diffusionGS.py:1447-1483, 821-862): what
// `warp_images_bw` and the uncertainty fusion do per frame with numpy / cv2 on the host, for all frames at once.

// per pixel: hard mask = 5x5 dilate of (1 - mask_reproj >= 0.5) (cv2.dilate, default border: the window's
// in-image part), cond = uint8(warped * (1 - ero)) / 255 (the reference's uint8 round trip), cond_ori = warped / 255,
// soft = 1 - soft_mask_reproj
__global__ void k_warp_post(const uint8_t* __restrict__ mask_reproj, const float* __restrict__ warped,
                            const float* __restrict__ soft_in, int H, int W, uint8_t* __restrict__ ero,
                            float* __restrict__ cond, float* __restrict__ cond_ori, float* __restrict__ soft) {
    __shared__ uint8_t tile[20][20];
    const int f = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
    const long long plane = (long long)H * W;
    const uint8_t* mr = mask_reproj + f * plane;
    for (int i = threadIdx.x; i < 400; i += 256) {
        int ty = i / 20, tx = i - ty * 20, y = y0 + ty - 2, x = x0 + tx - 2;
        tile[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? (uint8_t)(mr[(long long)y * W + x] == 0) : (uint8_t)0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    unsigned m = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) m |= tile[ty + dy][tx + dx];
    const long long pix = (long long)y * W + x;
    ero[f * plane + pix] = (uint8_t)m;
    const float keep = m ? 0.0f : 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = warped[((long long)f * 3 + c) * plane + pix];
        cond_ori[(f * plane + pix) * 3 + c] = v / 255.0f;
        float t = v * keep;                                   // np.uint8(): truncation (values are 0..255)
        t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
        cond[(f * plane + pix) * 3 + c] = (float)(unsigned)t / 255.0f;
    }
    soft[f * plane + pix] = 1.0f - soft_in[f * plane + pix];
}

// numpy's float32 mean over a contiguous run of fy*fx values (pairwise sum: 8 running sums below 128 elements)
__device__ __forceinline__ float np_block_mean(const float* __restrict__ src, int W, int fy, int fx) {
    const int n = fy * fx;
    float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float res;
    if (n >= 8 && n <= 128) {
        int i = 0;
        for (int k = 0; k < 8; ++k) r[k] = src[(long long)(k / fx) * W + k % fx];
        for (i = 8; i + 8 <= n; i += 8)
            for (int k = 0; k < 8; ++k) r[k] += src[(long long)((i + k) / fx) * W + (i + k) % fx];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += src[(long long)(i / fx) * W + i % fx];
    } else {
        res = 0.f;
        for (int i = 0; i < n; ++i) res += src[(long long)(i / fx) * W + i % fx];
    }
    return res / (float)n;
}

// per (h, w) cell of fy x fx pixels: mask = mean(ero) >= 0.2, soft_pool = mean(soft)
__global__ void k_warp_pool(const uint8_t* __restrict__ ero, const float* __restrict__ soft, int H, int W, int h, int w,
                            long long cells, float* __restrict__ masks, float* __restrict__ soft_pool) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const int fy = H / h, fx = W / w;
    const int cx = (int)(i % w), cy = (int)((i / w) % h);
    const long long f = i / ((long long)w * h);
    const long long base = f * H * W + (long long)cy * fy * W + (long long)cx * fx;
    if (ero) {
        int cnt = 0;
        for (int dy = 0; dy < fy; ++dy)
            for (int dx = 0; dx < fx; ++dx) cnt += ero[base + (long long)dy * W + dx];
        masks[i] = (double)cnt / (double)(fy * fx) >= 0.2 ? 1.0f : 0.0f;
    }
    if (soft) soft_pool[i] = np_block_mean(soft + base, W, fy, fx);
}

// diffusionGS.py:821-862: intensity confidence exp(-(|warp - gs|_2 / 0.5)^3) * (warp != 0), uncertainty
// u = 1 - conf * (1 - soft), cond = clip(u > 0.5 ? gs : warp, 0, 1)
__global__ void k_fuse_uncertainty(const float* __restrict__ cond_ori, const float* __restrict__ gs,
                                   const float* __restrict__ soft, long long npix, float* __restrict__ unc,
                                   float* __restrict__ cond) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float a0 = cond_ori[i * 3], a1 = cond_ori[i * 3 + 1], a2 = cond_ori[i * 3 + 2];
    const float g0 = gs[i * 3], g1 = gs[i * 3 + 1], g2 = gs[i * 3 + 2];
    const float known = (a0 + a1) + a2 > 0.0f ? 1.0f : 0.0f;
    const float d0 = a0 - g0, d1 = a1 - g1, d2 = a2 - g2;
    const float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    const float q = nrm / 0.5f;
    const float conf = expf(-(q * q * q)) * known;
    const float u = 1.0f - conf * (1.0f - soft[i]);
    unc[i] = u;
    const bool use_gs = u > 0.5f;
    const float c0 = use_gs ? g0 : a0, c1 = use_gs ? g1 : a1, c2 = use_gs ? g2 : a2;
    cond[i * 3] = fminf(fmaxf(c0, 0.f), 1.f);
    cond[i * 3 + 1] = fminf(fmaxf(c1, 0.f), 1.f);
    cond[i * 3 + 2] = fminf(fmaxf(c2, 0.f), 1.f);
}

// Forward / backward optical-flow consistency mask: the test behind the reference's
// `gsTrainer.generate_corresp_mask(gs_renderings, svd_outputs, dist_thresh=3, desc_only=False)` (model/diffusionGS.py:377).
// FSGS' wrapper and GMFlow are absent from /root/reference; the quantity is the published cycle check of a flow pair
// (GMFlow `forward_backward_consistency_check`, with the fixed pixel threshold the call site passes): a pixel x of image A is
// a correspondence when following the A->B flow and then the B->A flow SAMPLED at the landing point (bilinear, pixel
// coordinates) returns to within `thresh` pixels of x, and the landing point lies inside the image.
// flows [n,2,H,W] (channel 0 = dx, 1 = dy).  mask [n,H,W] in {0,1}; dist [n,H,W] (optional) = the cycle error, +inf outside.
__global__ void __launch_bounds__(256) k_flow_cycle_mask(const float* __restrict__ fw, const float* __restrict__ bw, int H, int W,
                                                        long long npix, float thresh, float* __restrict__ mask,
                                                        float* __restrict__ dist) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int hw = H * W;
    const int b = (int)(i / hw), r = (int)(i % hw), y = r / W, x = r % W;
    const float* f = fw + (size_t)b * 2 * hw;
    const float* g = bw + (size_t)b * 2 * hw;
    const float fx = f[r], fy = f[hw + r];
    const float tx = (float)x + fx, ty = (float)y + fy;
    float d = __builtin_huge_valf();
    if (tx >= 0.0f && tx <= (float)(W - 1) && ty >= 0.0f && ty <= (float)(H - 1)) {
        const int x0 = min((int)tx, W - 2 < 0 ? 0 : W - 2), y0 = min((int)ty, H - 2 < 0 ? 0 : H - 2);
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float ax = tx - (float)x0, ay = ty - (float)y0;
        const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
        const float gx = ((g[y0 * W + x0] * w00 + g[y0 * W + x1] * w01) + g[y1 * W + x0] * w10) + g[y1 * W + x1] * w11;
        const float gy = ((g[hw + y0 * W + x0] * w00 + g[hw + y0 * W + x1] * w01) + g[hw + y1 * W + x0] * w10) + g[hw + y1 * W + x1] * w11;
        const float ex = fx + gx, ey = fy + gy;
        d = sqrtf(ex * ex + ey * ey);
    }
    mask[i] = d < thresh ? 1.0f : 0.0f;
    if (dist) dist[i] = d;
}

}  // namespace

extern "C" size_t syn3r_inverse_warp_workspace_bytes(int nb) { return (size_t)(SYN3R_SIDE_OK(nb) ? nb : 1) * 16; }

extern "C" int syn3r_inverse_warp(const float* img, const float* depth, const float* depth_pseudo,
                                  const float* pose12, const float* pose21, const float* K, const float* Kinv,
                                  float bandwidth, int nb, int H, int W, float* warped_img, float* warped_depth,
                                  uint8_t* mask_warp, uint8_t* mask_depth, uint8_t* mask,
                                  float* warped_masked_img, uint8_t* mask_inv, uint8_t* mask_depth_strict,
                                  uint8_t* mask_reproj, float* soft_mask_reproj, float* reproj_error,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    SYN3R_REQUIRE(img && depth && depth_pseudo && pose12 && pose21 && K && Kinv, "inverse_warp: null input");
    SYN3R_REQUIRE(warped_img && warped_depth && mask_warp && mask_depth && mask && warped_masked_img &&
                      mask_inv && mask_depth_strict && mask_reproj && soft_mask_reproj,
                  "inverse_warp: null output");
    SYN3R_REQUIRE(SYN3R_SIDE_OK(nb) && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W), "inverse_warp: bad shape nb=%d H=%d W=%d", nb, H, W);
    SYN3R_REQUIRE((long long)H * W < (1ll << 30), "inverse_warp: image too large");
    if (!workspace || workspace_bytes < syn3r_inverse_warp_workspace_bytes(nb)) {
        set_error("inverse_warp: workspace %zu < %zu", workspace_bytes, syn3r_inverse_warp_workspace_bytes(nb));
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t n = (size_t)H * W;
    unsigned* mm = (unsigned*)workspace;
    // slot 2b = max (init 0 = below every encoded float), slot 2b+1 = min (init all-ones);
    // 16-byte stride per target so that both memsets are aligned
    int rc = check_hip(hipMemsetAsync(mm, 0x00, (size_t)nb * 16, stream), "memset");
    if (rc) return rc;
    for (int b = 0; b < nb; ++b) {
        rc = check_hip(hipMemsetAsync(mm + 4 * b + 1, 0xFF, 4, stream), "memset");
        if (rc) return rc;
    }
    for (int b = 0; b < nb; ++b) {
        IwParams p;
        load4(p.pose12, pose12 + 16 * b);
        load4(p.pose21, pose21 + 16 * b);
        load3(p.K, K);
        load3(p.Kinv, Kinv);
        p.fx = K[0]; p.fy = K[4]; p.cx = K[2]; p.cy = K[5];
        p.bandwidth = bandwidth; p.H = H; p.W = W;
        const float* dp = depth_pseudo + b * n;
        int g = grid_for((int)n);
        // one same-address atomic pair per block serialises in L2 (~20 ns each): keep the reduction grid small
        int g_mm = g > 128 ? 128 : g;
        SYN3R_LAUNCH(k_iw_minmax, dim3(g_mm), dim3(kBlock), 0, stream, p, depth, dp, mm + 4 * b);
        SYN3R_LAUNCH(k_iw_main, dim3(g), dim3(kBlock), 0, stream, p, img, depth, dp, mm + 4 * b,
                           warped_img + 3 * b * n, warped_depth + b * n, mask_warp + b * n, mask_depth + b * n,
                           mask + b * n, warped_masked_img + 3 * b * n, mask_inv + b * n,
                           mask_depth_strict + b * n, mask_reproj + b * n, soft_mask_reproj + b * n,
                           reproj_error ? reproj_error + b * n : nullptr);
    }
    SYN3R_LAUNCH_CHECK("inverse_warp launch");
    return SYN3R_OK;
}

extern "C" int syn3r_reproj_error(const float* depth1, const float* depth2, const float* T12, const float* T21,
                                  const float* K1, const float* K1inv, const float* K2, int H, int W, float* err,
                                  void* stream_) {
    SYN3R_REQUIRE(depth1 && depth2 && T12 && T21 && K1 && K1inv && K2 && err, "reproj_error: null argument");
    SYN3R_REQUIRE(SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W), "reproj_error: bad shape H=%d W=%d", H, W);
    RpParams p;
    load4(p.T12, T12); load4(p.T21, T21);
    load3(p.K1, K1); load3(p.K1inv, K1inv); load3(p.K2, K2);
    p.H = H; p.W = W;
    SYN3R_LAUNCH(k_reproj, dim3(grid_for(H * W)), dim3(kBlock), 0, (hipStream_t)stream_, p, depth1, depth2,
                       err);
    SYN3R_LAUNCH_CHECK("reproj_error launch");
    return SYN3R_OK;
}

extern "C" size_t syn3r_forward_warp_workspace_bytes(int H, int W) {
    if (!SYN3R_SIDE_OK(H) || !SYN3R_SIDE_OK(W)) return 0;
    return 16 + (size_t)(H + 2) * (W + 2) * 4 * sizeof(double);
}

extern "C" int syn3r_forward_warp(const double* frame1, const uint8_t* mask1, const double* depth1,
                                  const double* T, const double* K1inv, const double* K2, int H, int W,
                                  uint8_t* warped, uint8_t* mask2, double* flow12, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    SYN3R_REQUIRE(frame1 && depth1 && T && K1inv && K2 && warped && mask2 && flow12, "forward_warp: null argument");
    SYN3R_REQUIRE(SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && (long long)(H + 2) * (W + 2) < (1ll << 28), "forward_warp: bad shape H=%d W=%d",
                  H, W);
    size_t need = syn3r_forward_warp_workspace_bytes(H, W);
    if (!workspace || workspace_bytes < need) {
        set_error("forward_warp: workspace %zu < %zu", workspace_bytes, need);
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    FwParams p;
    for (int i = 0; i < 16; ++i) p.T.m[i] = T[i];
    for (int i = 0; i < 9; ++i) { p.K1inv.m[i] = K1inv[i]; p.K2.m[i] = K2[i]; }
    p.H = H; p.W = W;
    unsigned long long* maxbits = (unsigned long long*)workspace;
    double* acc = (double*)((char*)workspace + 16);
    int rc = check_hip(hipMemsetAsync(workspace, 0, need, stream), "memset");
    if (rc) return rc;
    int g = grid_for(H * W);
    SYN3R_LAUNCH(k_fw_maxdepth, dim3(g), dim3(kBlock), 0, stream, p, depth1, maxbits);
    SYN3R_LAUNCH(k_fw_splat, dim3(g), dim3(kBlock), 0, stream, p, frame1, mask1, depth1, maxbits, acc, flow12);
    SYN3R_LAUNCH(k_fw_finish, dim3(g), dim3(kBlock), 0, stream, H, W, acc, warped, mask2);
    SYN3R_LAUNCH_CHECK("forward_warp launch");
    return SYN3R_OK;
}

extern "C" int syn3r_warp_post(const uint8_t* mask_reproj, const float* warped_img, const float* soft_mask_reproj,
                               int n, int H, int W, int h, int w, uint8_t* ero, float* cond_image, float* cond_ori,
                               float* soft, float* masks, float* soft_pool, void* stream_) {
    SYN3R_REQUIRE(mask_reproj && warped_img && soft_mask_reproj && ero && cond_image && cond_ori && soft && masks && soft_pool,
                  "warp_post: null argument");
    SYN3R_REQUIRE(SYN3R_SIDE_OK(n) && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && h > 0 && w > 0 && H % h == 0 && W % w == 0,
                  "warp_post: bad shape n=%d H=%d W=%d pooled %dx%d", n, H, W, h, w);
    SYN3R_REQUIRE(n <= 65535, "warp_post: too many frames");
    hipStream_t stream = (hipStream_t)stream_;
    SYN3R_LAUNCH(k_warp_post, dim3((W + 15) / 16, (H + 15) / 16, n), dim3(256), 0, stream, mask_reproj, warped_img,
                 soft_mask_reproj, H, W, ero, cond_image, cond_ori, soft);
    const long long cells = (long long)n * h * w;
    SYN3R_LAUNCH(k_warp_pool, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)ero,
                 (const float*)soft, H, W, h, w, cells, masks, soft_pool);
    SYN3R_LAUNCH_CHECK("warp_post launch");
    return SYN3R_OK;
}

extern "C" int syn3r_fuse_uncertainty(const float* cond_ori, const float* gs_images, const float* soft, int n, int H,
                                      int W, int h, int w, float* uncertainty, float* cond_image, float* masks,
                                      void* stream_) {
    SYN3R_REQUIRE(cond_ori && gs_images && soft && uncertainty && cond_image && masks, "fuse_uncertainty: null argument");
    SYN3R_REQUIRE(SYN3R_SIDE_OK(n) && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && h > 0 && w > 0 && H % h == 0 && W % w == 0,
                  "fuse_uncertainty: bad shape n=%d H=%d W=%d pooled %dx%d", n, H, W, h, w);
    hipStream_t stream = (hipStream_t)stream_;
    const long long npix = (long long)n * H * W;
    SYN3R_LAUNCH(k_fuse_uncertainty, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, cond_ori, gs_images,
                 soft, npix, uncertainty, cond_image);
    const long long cells = (long long)n * h * w;
    SYN3R_LAUNCH(k_warp_pool, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)nullptr,
                 (const float*)uncertainty, H, W, h, w, cells, (float*)nullptr, masks);
    SYN3R_LAUNCH_CHECK("fuse_uncertainty launch");
    return SYN3R_OK;
}

extern "C" int syn3r_flow_cycle_mask(const float* flow_fw, const float* flow_bw, int n, int H, int W, float thresh, float* mask,
                                     float* dist, void* stream_) {
    SYN3R_REQUIRE(flow_fw && flow_bw && mask, "flow_cycle_mask: null argument");
    SYN3R_REQUIRE(SYN3R_SIDE_OK(n) && SYN3R_SIDE_OK(H) && SYN3R_SIDE_OK(W) && (long long)n * H * W < (1ll << 40),
                  "flow_cycle_mask: bad shape n=%d H=%d W=%d", n, H, W);
    SYN3R_REQUIRE(thresh > 0.0f, "flow_cycle_mask: threshold must be positive");
    hipStream_t stream = (hipStream_t)stream_;
    const long long npix = (long long)n * H * W;
    SYN3R_REQUIRE((npix + 255) / 256 < (1ll << 31), "flow_cycle_mask: too many pixels");
    SYN3R_LAUNCH(k_flow_cycle_mask, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, flow_fw, flow_bw, H, W, npix, thresh,
                 mask, dist);
    SYN3R_LAUNCH_CHECK("flow_cycle_mask launch");
    return SYN3R_OK;
}
