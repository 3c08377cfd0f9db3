// Gaussian rasteriser, backward: per-tile back-to-front traversal of the blend, then the
// per-Gaussian chain rule through EWA projection, SH colour and covariance construction.
//
// Replaces the rasteriser backward inside gsTrainer.training()/finetune() (call sites
// model/diffusionGS.py:139,1640); restates the published 3DGS backward (see raster_common.h;
// reference CUDA source absent).  Gradients are validated against autograd
// through oracle/raster_oracle.py.
//
// MI355X mapping: each wavefront owns a 16 x 8 pixel half of the tile, two pixels per lane on packed fp32
// arithmetic, and walks only the splats that can reach alpha >= 1/255 on that half (splat_reaches_rect,
// raster_common.h).  Per visited splat the up to 128 pixel contributions to 10 quantities are summed by
// reduce10 (two half/row swap levels + DPP row rotates,
// 30 VALU instructions), accumulated per (tile, splat) in LDS across the two wavefronts, and flushed with
// ONE atomic per record slot onto a contiguous 64-byte gradient record (MI355X float atomics want
// contiguous segments, MI355X_MICROARCH.md "Global float atomics").
#include "common.h"
#include "raster_common.h"

using namespace syn3r;

namespace syn3r {
void raster_fill_camera(Camera& cam, const float* view, const float* proj, const float* campos, float tanfovx,
                        float tanfovy, int H, int W);
}

namespace {

// 64-byte per-Gaussian gradient record written by the blend backward
constexpr int kGradSlots = 16;
enum { G_R = 0, G_G, G_B, G_DEPTH, G_MX, G_MY, G_CXX, G_CXY, G_CYY, G_OP, G_USED = 10 };

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// Sum each of the 10 per-lane gradient terms over the 64 lanes.  Returns, on lane l with column c = l & 15 and 16-lane row q = l >> 4,
// the total of value base(q) + c for c < 3 (even rows) / c < 2 (odd rows), base = 0, 3, 5, 8: row 0 owns values 0-2, row 1 values 3-4,
// row 2 values 5-7, row 3 values 8-9 (reduce_slot below).
// Two transposing levels use gfx950's half / row swaps (v_permlane32_swap, v_permlane16_swap): one VALU instruction moves BOTH
// directions of the exchange, so a level costs a swap and an add per surviving value and halves the number of live registers
// (10 -> 5 -> 3; the odd one of the second level is swapped against a copy of itself).  The last four levels stay inside a 16-lane
// row: v_add_f32 with a DPP row rotate, written as inline asm - through __builtin_amdgcn_update_dpp hipcc split the row_ror:1 step into
// v_mov 0 + v_mov_dpp + v_add, and the 12-value form of rounds 2-5 swapped two zero pads through both levels: 38 instructions, now 29
// (round 6; the kernel is bound by vector issue, 128 -> 119 instructions per visit).  No LDS-pipe ds_bpermute anywhere.
__device__ __forceinline__ float row_sum16(float x) {
    float y;
    asm("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
    asm("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y));
    asm("v_add_f32_dpp %0, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
    asm("v_add_f32_dpp %0, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(y));
    return x;
}
__device__ __forceinline__ float reduce10(float (&v)[10], int lane) {
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {   // lanes 32-63 of v[k] <-> lanes 0-31 of v[k+5]: the lower half owns k, the upper k + 5
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v[k]), __float_as_int(v[k + 5]), false, false);
        a[k] = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
    float b[3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {   // odd 16-lane rows of a[k] <-> even rows of a[k+3]: even rows own k, odd rows k + 3
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a[k]), __float_as_int(a[k + 3]), false, false);
        b[k] = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
    {                                // a[2] against itself: both rows of a pair get the pair's total (the odd rows' copy is not stored)
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a[2]), __float_as_int(a[2]), false, false);
        b[2] = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
    b[0] = row_sum16(b[0]); b[1] = row_sum16(b[1]); b[2] = row_sum16(b[2]);
    const int c = lane & 15;
    return c == 0 ? b[0] : (c == 1 ? b[1] : b[2]);
}
// the gradient slot reduce10's return value belongs to on this lane, or -1
__device__ __forceinline__ int reduce_slot(int lane) {
    const int c = lane & 15, q = lane >> 4;
    return c < ((q & 1) ? 2 : 3) ? ((5 * q + 1) >> 1) + c : -1;
}

// ---------------------------------------------------------------------------------------------
// Blend backward, two pixels per lane.
//
// A 16 x 16 tile is a block of TWO wavefronts; wavefront w owns the 16 x 8 half (rows 8w .. 8w+7) and lane l the
// pixels (l & 15, 8w + (l >> 4)) and (l & 15, 8w + (l >> 4) + 4).  Every per-pixel quantity is a float2 and the
// arithmetic is written on float2 so that it compiles to gfx950's packed fp32 instructions (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two pixels per VALU issue); only exp, rcp, min and the compares stay one per pixel.
// The kernel is VALU-bound (rocprofv3 round 1: 78 % VALU issue): against one pixel per lane this halves the issue
// slots of the chain-rule arithmetic and halves the number of cross-lane reductions per pixel (the two pixels of a
// lane are added before reduce10).  The visit list is per wavefront, i.e. per 16 x 8 half: coarser than the former
// 8 x 8 quadrant (more visits pass the reach test), but a visit now carries 128 pixels for ~0.6 of the issue cost
// of two 64-pixel visits.  Splats are staged 128 at a time (one per thread).
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kBwdThreads = 128;
#ifdef SYN3R_RASTER_STATS      // developer build: [0] lane tests, [1] wavefront visits, [2] visits with an active pixel, [3] active pixels
__device__ unsigned long long g_bwd_stats[4];
#define BSTAT(i, n) do { if (lane == 0) atomicAdd(&g_bwd_stats[i], (unsigned long long)(n)); } while (0)
#else
#define BSTAT(i, n)
#endif

__device__ __forceinline__ f2 splat2(float s) { return (f2){s, s}; }

__global__ void __launch_bounds__(kBwdThreads) k_render_bwd(
    int H, int W, int gx, int gy, const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list,
    const Splat* __restrict__ splats, float bg0, float bg1, float bg2, const unsigned* __restrict__ n_contrib,
    const float* __restrict__ final_T, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dalpha_out, float* __restrict__ grad_rec, const unsigned* __restrict__ tile_order) {
    __shared__ float4 sm[kBwdThreads * 3];
    __shared__ unsigned sid[kBwdThreads];
    __shared__ float sacc[kBwdThreads * kGradSlots];   // per-round gradient records: the 2 wavefronts meet here first
    const unsigned tile = tile_order ? tile_order[blockIdx.x] : xcd_remap(blockIdx.x, (unsigned)(gx * gy));
    const int tx = tile % gx, ty = tile / gx;
    const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = lane & 15, ly = wq * 8 + (lane >> 4);
    const int px = tx * kTileX + lx, py0 = ty * kTileY + ly, py1 = py0 + 4;
    const bool in0 = px < W && py0 < H, in1 = px < W && py1 < H;
    const float fx = (float)px;
    const f2 fy = (f2){(float)py0, (float)py1};
    const uint2 range = ranges[tile];
    const size_t hw = (size_t)H * W, pix0 = (size_t)py0 * W + px, pix1 = (size_t)py1 * W + px;

    const f2 T_final = (f2){in0 ? final_T[pix0] : 0.0f, in1 ? final_T[pix1] : 0.0f};
    f2 T = T_final;
    const int lc0 = in0 ? (int)n_contrib[pix0] : 0, lc1 = in1 ? (int)n_contrib[pix1] : 0;
    // The forward pass stops a tile once every pixel is saturated; the backward walks back from the LAST splat
    // any pixel of the tile took (max of n_contrib), not from the end of the tile's list.
    __shared__ int s_live;
    int wave_live;                        // the same bound for this wavefront's half alone
    if (threadIdx.x == 0) s_live = 0;
    __syncthreads();
    {
        int mc = max(lc0, lc1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mc = max(mc, __shfl_xor(mc, o, 64));
        if (lane == 0) atomicMax(&s_live, mc);
        wave_live = mc;
    }
    __syncthreads();
    const int total = min(s_live, (int)(range.y - range.x));
    const int rounds = (total + kBwdThreads - 1) / kBwdThreads;
    const float sx0 = (float)(tx * kTileX), sx1 = sx0 + 15.0f;
    const float sy0 = (float)(ty * kTileY + wq * 8), sy1 = sy0 + 7.0f;
    f2 gr = splat2(0.f), gg = splat2(0.f), gb = splat2(0.f), gD = splat2(0.f), gA = splat2(0.f);
    if (in0) {
        gr.x = dL_dcolor[pix0]; gg.x = dL_dcolor[hw + pix0]; gb.x = dL_dcolor[2 * hw + pix0];
        gD.x = dL_ddepth ? dL_ddepth[pix0] : 0.0f;
        gA.x = dL_dalpha_out ? dL_dalpha_out[pix0] : 0.0f;
    }
    if (in1) {
        gr.y = dL_dcolor[pix1]; gg.y = dL_dcolor[hw + pix1]; gb.y = dL_dcolor[2 * hw + pix1];
        gD.y = dL_ddepth ? dL_ddepth[pix1] : 0.0f;
        gA.y = dL_dalpha_out ? dL_dalpha_out[pix1] : 0.0f;
    }
    // output terms that do not depend on the splat: background of the colour output and A = 1 - T_final
    const f2 tail = T_final * (gA - (bg0 * gr + bg1 * gg + bg2 * gb));
    f2 acc_r = splat2(0.f), acc_g = splat2(0.f), acc_b = splat2(0.f), acc_d = splat2(0.f), last_alpha = splat2(0.f);
    float last_r = 0.f, last_g = 0.f, last_b = 0.f, last_d = 0.f;      // colour of the last visited splat: wave-uniform
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    const int rslot = reduce_slot(lane);
    int todo = total;
    // The records of round rd + 1 are requested (list entry, then the 48-byte record: two dependent global loads)
    // BEFORE round rd is processed and land in registers meanwhile: the gather latency is off the critical path.
    float4 n0, n1, n2;
    unsigned ngid = 0;
    bool have = false;
    auto fetch = [&](int rd) {
        const int progress = rd * kBwdThreads + threadIdx.x;
        have = progress < total;
        if (have) {
            ngid = point_list[range.x + total - 1 - progress];   // back to front
            const float4* src = (const float4*)(splats + ngid);
            n0 = src[0]; n1 = src[1]; n2 = src[2];
        }
    };
    fetch(0);
    for (int rd = 0; rd < rounds; ++rd, todo -= kBwdThreads) {
        __syncthreads();
        if (have) {
            sm[threadIdx.x * 3 + 0] = n0;
            sm[threadIdx.x * 3 + 1] = n1;
            sm[threadIdx.x * 3 + 2] = n2;
            sid[threadIdx.x] = ngid;
        }
        fetch(rd + 1);
#pragma unroll
        for (int k = 0; k < kGradSlots / 4; ++k)
            ((float4*)sacc)[threadIdx.x * (kGradSlots / 4) + k] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const int cnt = min(kBwdThreads, todo);
        // visit list of this wavefront's half (see k_render): one lane-test per staged splat, then a scalar walk
        // over the ballot; splats that cannot reach alpha >= 1/255 on the half are never evaluated
        for (int c0 = 0; c0 < cnt; c0 += 64) {
          bool hit = false;
          if (c0 + lane < cnt) {
              const float4 ta = sm[(c0 + lane) * 3], tb = sm[(c0 + lane) * 3 + 1];
              hit = splat_reaches_rect(ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, sx0, sx1, sy0, sy1) &&
                    (total - 1 - (rd * kBwdThreads + c0 + lane)) < wave_live;
          }
          unsigned long long vm = __ballot(hit);
          BSTAT(0, min(64, cnt - c0));
          BSTAT(1, __popcll(vm));
          while (vm) {
            const int j = c0 + (int)__builtin_ctzll(vm);
            vm &= vm - 1;
            const int contributor = total - 1 - (rd * kBwdThreads + j);
            const float4 a = sm[j * 3], b = sm[j * 3 + 1], c = sm[j * 3 + 2];
            // a = (x, y, cxx, cxy)  b = (cyy, opacity, r, g)  c = (b, depth, -, -)
            const float dx = a.x - fx;
            const f2 dy = splat2(a.y) - fy;
            const float hxx = -0.5f * a.z * dx * dx, bxy = a.w * dx;
            const f2 power = (-0.5f * b.x) * dy * dy - bxy * dy + hxx;
            f2 G = (f2){__expf(power.x), __expf(power.y)};
            const f2 araw = b.y * G;
            const bool act0 = (contributor < lc0) && (power.x <= 0.0f) && (fminf(kAlphaMax, araw.x) >= kAlphaMin);
            const bool act1 = (contributor < lc1) && (power.y <= 0.0f) && (fminf(kAlphaMax, araw.y) >= kAlphaMin);
            if (__ballot(act0 || act1) == 0ull) continue;   // wave-uniform
#ifdef SYN3R_RASTER_STATS
            BSTAT(2, 1); BSTAT(3, __popcll(__ballot(act0)) + __popcll(__ballot(act1)));
#endif
            // Branch-free: a pixel that does not take this splat blends it with alpha = 0 and G = 0, which is an exact
            // no-op on its running state (T * rcp(1) = T, the colour recursion absorbs a zero-weight layer exactly)
            // and makes every gradient term an exact zero - no EXEC-masked region.
            const f2 a_eff = (f2){act0 ? fminf(kAlphaMax, araw.x) : 0.0f, act1 ? fminf(kAlphaMax, araw.y) : 0.0f};
            G = (f2){act0 ? G.x : 0.0f, act1 ? G.y : 0.0f};
            const f2 one_m = 1.0f - a_eff;
            const f2 inv1ma = (f2){__builtin_amdgcn_rcpf(one_m.x), __builtin_amdgcn_rcpf(one_m.y)};   // 1 ulp reciprocal
            T = T * inv1ma;
            const f2 wgt = a_eff * T;
            // colour / depth recursion of the contribution behind this splat
            const f2 one_la = 1.0f - last_alpha;
            acc_r = last_alpha * last_r + one_la * acc_r;
            acc_g = last_alpha * last_g + one_la * acc_g;
            acc_b = last_alpha * last_b + one_la * acc_b;
            acc_d = last_alpha * last_d + one_la * acc_d;
            last_r = b.z; last_g = b.w; last_b = c.x; last_d = c.y;
            last_alpha = a_eff;
            f2 dL_da = (b.z - acc_r) * gr + (b.w - acc_g) * gg + (c.x - acc_b) * gb + (c.y - acc_d) * gD;
            dL_da = dL_da * T + tail * inv1ma;
            const f2 dL_dG = b.y * dL_da;
            const f2 gdx = G * dx, gdy = G * dy;
            const f2 dG_ddelx = -(gdx * a.z) - gdy * a.w;
            const f2 dG_ddely = -(gdy * b.x) - gdx * a.w;
            f2 w[10];
            w[G_R] = wgt * gr; w[G_G] = wgt * gg; w[G_B] = wgt * gb; w[G_DEPTH] = wgt * gD;
            w[G_MX] = dL_dG * dG_ddelx * ddelx_dx;
            w[G_MY] = dL_dG * dG_ddely * ddely_dy;
            w[G_CXX] = (-0.5f * dx) * gdx * dL_dG;
            w[G_CXY] = -(gdx * dy) * dL_dG;
            w[G_CYY] = -0.5f * (gdy * dy) * dL_dG;
            w[G_OP] = G * dL_da;
            float v[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) v[k] = w[k].x + w[k].y;
            const float s = reduce10(v, lane);
            if (rslot >= 0) atomicAdd(&sacc[j * kGradSlots + rslot], s);   // LDS, 10 banks
          }
        }
        // one global atomic per (tile, splat) instead of one per (wavefront, splat): 16 lanes per record, so a
        // wave-instruction covers four contiguous 64-byte records
        __syncthreads();
        {
            const int slot = threadIdx.x & 15;
            for (int q = threadIdx.x >> 4; q < cnt; q += kBwdThreads / 16) {
                float val = sacc[q * kGradSlots + slot];
                if (slot < G_USED && val != 0.0f) unsafeAtomicAdd(grad_rec + (size_t)sid[q] * kGradSlots + slot, val);
            }
        }
    }
}

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__constant__ float B_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
__constant__ float B_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                              -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// SH colour backward: writes dL_dsh (M x 3) and returns dL_dmean through the view direction
// PRELOAD (M == 16): every coefficient is read into registers before the first output is written, so `dL_dsh` may be the row
// `sh` itself (k_preprocess_bwd's LDS-staged rows)
template <bool PRELOAD>
__device__ __forceinline__ F3 sh_backward(int D, int M, F3 pos, const float* campos, const float* sh, unsigned clamped,
                                          F3 dL_dRGB, float* dL_dsh) {
    F3 dir_o = f3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
    float len2 = dot(dir_o, dir_o);
    float inv = 1.0f / sqrtf(len2);
    float x = dir_o.x * inv, y = dir_o.y * inv, z = dir_o.z * inv;
    if (clamped & 1u) dL_dRGB.x = 0.0f;
    if (clamped & 2u) dL_dRGB.y = 0.0f;
    if (clamped & 4u) dL_dRGB.z = 0.0f;
    float pre[PRELOAD ? 48 : 1];
    if constexpr (PRELOAD) {
#pragma unroll
        for (int k = 0; k < 48; ++k) pre[k] = sh[k];
    }
    auto c = [&](int k) { return PRELOAD ? f3(pre[3 * k], pre[3 * k + 1], pre[3 * k + 2]) : f3(sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]); };
    auto put = [&](int k, float s) {
        dL_dsh[3 * k] = s * dL_dRGB.x; dL_dsh[3 * k + 1] = s * dL_dRGB.y; dL_dsh[3 * k + 2] = s * dL_dRGB.z;
    };
    F3 dx = f3(0, 0, 0), dy = f3(0, 0, 0), dz = f3(0, 0, 0);
    put(0, SH_C0);
    if (D > 0) {
        put(1, -SH_C1 * y); put(2, SH_C1 * z); put(3, -SH_C1 * x);
        dx = -SH_C1 * c(3); dy = -SH_C1 * c(1); dz = SH_C1 * c(2);
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            put(4, B_C2[0] * xy); put(5, B_C2[1] * yz); put(6, B_C2[2] * (2.f * zz - xx - yy));
            put(7, B_C2[3] * xz); put(8, B_C2[4] * (xx - yy));
            dx = dx + (B_C2[0] * y) * c(4) + (B_C2[2] * 2.f * -x) * c(6) + (B_C2[3] * z) * c(7) + (B_C2[4] * 2.f * x) * c(8);
            dy = dy + (B_C2[0] * x) * c(4) + (B_C2[1] * z) * c(5) + (B_C2[2] * 2.f * -y) * c(6) + (B_C2[4] * 2.f * -y) * c(8);
            dz = dz + (B_C2[1] * y) * c(5) + (B_C2[2] * 2.f * 2.f * z) * c(6) + (B_C2[3] * x) * c(7);
            if (D > 2) {
                put(9, B_C3[0] * y * (3.f * xx - yy)); put(10, B_C3[1] * xy * z);
                put(11, B_C3[2] * y * (4.f * zz - xx - yy)); put(12, B_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                put(13, B_C3[4] * x * (4.f * zz - xx - yy)); put(14, B_C3[5] * z * (xx - yy));
                put(15, B_C3[6] * x * (xx - 3.f * yy));
                dx = dx + (B_C3[0] * 3.f * 2.f * xy) * c(9) + (B_C3[1] * yz) * c(10) + (B_C3[2] * -2.f * xy) * c(11) +
                     (B_C3[3] * -3.f * 2.f * xz) * c(12) + (B_C3[4] * (-3.f * xx + 4.f * zz - yy)) * c(13) +
                     (B_C3[5] * 2.f * xz) * c(14) + (B_C3[6] * 3.f * (xx - yy)) * c(15);
                dy = dy + (B_C3[0] * 3.f * (xx - yy)) * c(9) + (B_C3[1] * xz) * c(10) +
                     (B_C3[2] * (-3.f * yy + 4.f * zz - xx)) * c(11) + (B_C3[3] * -3.f * 2.f * yz) * c(12) +
                     (B_C3[4] * -2.f * xy) * c(13) + (B_C3[5] * -2.f * yz) * c(14) + (B_C3[6] * -3.f * 2.f * xy) * c(15);
                dz = dz + (B_C3[1] * xy) * c(10) + (B_C3[2] * 4.f * 2.f * yz) * c(11) +
                     (B_C3[3] * 3.f * (2.f * zz - xx - yy)) * c(12) + (B_C3[4] * 4.f * 2.f * xz) * c(13) +
                     (B_C3[5] * (xx - yy)) * c(14);
            }
        }
    }
    for (int k = (D + 1) * (D + 1); k < M; ++k) put(k, 0.0f);
    F3 dL_ddir = f3(dot(dx, dL_dRGB), dot(dy, dL_dRGB), dot(dz, dL_dRGB));
    // d(v/|v|): (g - vhat (vhat . g)) / |v|
    F3 vh = f3(x, y, z);
    float vg = dot(vh, dL_ddir);
    return f3((dL_ddir.x - x * vg) * inv, (dL_ddir.y - y * vg) * inv, (dL_ddir.z - z * vg) * inv);
}

// STAGED (sh_coeffs == 16, 16-byte aligned tensors): a block's spherical-harmonics rows (256 x 192 B, contiguous in memory) come in
// and its gradient rows go out through LDS with coalesced 16-byte accesses; a thread reading and writing its own 192-byte row in
// global memory touches 64 different cache lines per wave instruction (32 of the kernel's 42 us at 200 000 Gaussians).
constexpr int kShLd = 49;      // LDS row stride (floats): odd, so the 64 rows of a wavefront fall into 64 banks
template <bool STAGED>
__global__ void __launch_bounds__(256) k_preprocess_bwd(
    int N, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ conf, float scale_mod, Camera cam, const int* __restrict__ radii, GeomState g,
    const float* __restrict__ grad_rec, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drots, float* __restrict__ dL_dopacity, float* __restrict__ dL_dshs,
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dconf, int raw) {
    __shared__ float shl[STAGED ? 256 * kShLd : 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t blk0 = (size_t)blockIdx.x * 256 * 48;                // first float of the block's rows
    const int rows = min(256, N - (int)blockIdx.x * 256);
    if constexpr (STAGED) {
        // the twelve loads of a thread are issued TOGETHER (a full block: every block but the last); behind a per-load `row < rows`
        // branch each would wait for the one before it (hipcc keeps a load inside its exec-mask region)
        if (rows == 256) {
            float4 v4[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v4[k] = *(const float4*)(shs + blk0 + (threadIdx.x + 256 * k) * 4);
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int e = (threadIdx.x + 256 * k) * 4, row = e / 48, col = e - row * 48;
                float* d = &shl[row * kShLd + col];
                d[0] = v4[k].x; d[1] = v4[k].y; d[2] = v4[k].z; d[3] = v4[k].w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int e = (threadIdx.x + 256 * k) * 4, row = e / 48, col = e - row * 48;
                if (row < rows) {
                    const float4 v4 = *(const float4*)(shs + blk0 + e);
                    float* d = &shl[row * kShLd + col];
                    d[0] = v4.x; d[1] = v4.y; d[2] = v4.z; d[3] = v4.w;
                }
            }
        }
        __syncthreads();
    }
    if (i < N) do {
    float* osh = STAGED ? &shl[threadIdx.x * kShLd] : dL_dshs + (size_t)i * M * 3;
    if (radii[i] <= 0) {
        for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = 0.f; dL_dscales[3 * i + k] = 0.f; dL_dmeans2D[3 * i + k] = 0.f; }
        for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0.f;
        dL_dopacity[i] = 0.f;
        if (dL_dconf) dL_dconf[i] = 0.f;
        for (int k = 0; k < 3 * M; ++k) osh[k] = 0.f;
        break;
    }
    const float* gr = grad_rec + (size_t)i * kGradSlots;
    F3 p = f3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    const float* v = cam.view;
    F3 t = f3(v[0] * p.x + v[4] * p.y + v[8] * p.z + v[12], v[1] * p.x + v[5] * p.y + v[9] * p.z + v[13],
              v[2] * p.x + v[6] * p.y + v[10] * p.z + v[14]);

    // ---- conic -> 2D covariance
    const float* cv = g.cov3D + 6 * (size_t)i;
    float c0 = cv[0], c1 = cv[1], c2 = cv[2], c3 = cv[3], c4 = cv[4], c5 = cv[5];
    float limx = kFovGuard * cam.tanfovx, limy = kFovGuard * cam.tanfovy;
    float txtz = t.x / t.z, tytz = t.y / t.z;
    float tx = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    float ty = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    float x_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    float y_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    float J00 = cam.focal_x / t.z, J02 = -(cam.focal_x * tx) / (t.z * t.z);
    float J11 = cam.focal_y / t.z, J12 = -(cam.focal_y * ty) / (t.z * t.z);
    float W00 = v[0], W01 = v[4], W02 = v[8], W10 = v[1], W11 = v[5], W12 = v[9], W20 = v[2], W21 = v[6], W22 = v[10];
    float T00 = J00 * W00 + J02 * W20, T01 = J00 * W01 + J02 * W21, T02 = J00 * W02 + J02 * W22;
    float T10 = J11 * W10 + J12 * W20, T11 = J11 * W11 + J12 * W21, T12 = J11 * W12 + J12 * W22;
    float a0 = c0 * T00 + c1 * T01 + c2 * T02, a1 = c1 * T00 + c3 * T01 + c4 * T02, a2 = c2 * T00 + c4 * T01 + c5 * T02;
    float b0 = c0 * T10 + c1 * T11 + c2 * T12, b1 = c1 * T10 + c3 * T11 + c4 * T12, b2 = c2 * T10 + c4 * T11 + c5 * T12;
    float A = T00 * a0 + T01 * a1 + T02 * a2 + kLowPass;
    float B = T00 * b0 + T01 * b1 + T02 * b2;
    float C = T10 * b0 + T11 * b1 + T12 * b2 + kLowPass;
    float den = A * C - B * B;
    float den2inv = 1.0f / (den * den + 0.0000001f);
    float gxx = gr[G_CXX], gxy = gr[G_CXY], gyy = gr[G_CYY];
    float dL_dA = 0.f, dL_dB = 0.f, dL_dC = 0.f;
    if (den2inv != 0.0f) {
        dL_dA = den2inv * (-C * C * gxx + B * C * gxy + (den - A * C) * gyy);
        dL_dC = den2inv * ((den - A * C) * gxx + A * B * gxy - A * A * gyy);
        dL_dB = den2inv * (2.f * B * C * gxx - (den + 2.f * B * B) * gxy + 2.f * A * B * gyy);
    }
    // dL/dSigma (stored upper triangle; off-diagonals count both symmetric entries)
    float dS0 = T00 * T00 * dL_dA + T00 * T10 * dL_dB + T10 * T10 * dL_dC;
    float dS3 = T01 * T01 * dL_dA + T01 * T11 * dL_dB + T11 * T11 * dL_dC;
    float dS5 = T02 * T02 * dL_dA + T02 * T12 * dL_dB + T12 * T12 * dL_dC;
    float dS1 = 2.f * T00 * T01 * dL_dA + (T00 * T11 + T01 * T10) * dL_dB + 2.f * T10 * T11 * dL_dC;
    float dS2 = 2.f * T00 * T02 * dL_dA + (T00 * T12 + T02 * T10) * dL_dB + 2.f * T10 * T12 * dL_dC;
    float dS4 = 2.f * T02 * T01 * dL_dA + (T01 * T12 + T02 * T11) * dL_dB + 2.f * T11 * T12 * dL_dC;
    // dL/dT, T = J W (2x3)
    float dT00 = 2.f * a0 * dL_dA + b0 * dL_dB, dT01 = 2.f * a1 * dL_dA + b1 * dL_dB, dT02 = 2.f * a2 * dL_dA + b2 * dL_dB;
    float dT10 = 2.f * b0 * dL_dC + a0 * dL_dB, dT11 = 2.f * b1 * dL_dC + a1 * dL_dB, dT12 = 2.f * b2 * dL_dC + a2 * dL_dB;
    float dJ00 = W00 * dT00 + W01 * dT01 + W02 * dT02;
    float dJ02 = W20 * dT00 + W21 * dT01 + W22 * dT02;
    float dJ11 = W10 * dT10 + W11 * dT11 + W12 * dT12;
    float dJ12 = W20 * dT10 + W21 * dT11 + W22 * dT12;
    float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = x_mul * -cam.focal_x * tz2 * dJ02;
    float dty = y_mul * -cam.focal_y * tz2 * dJ12;
    float dtz = -cam.focal_x * tz2 * dJ00 - cam.focal_y * tz2 * dJ11 + (2.f * cam.focal_x * tx) * tz3 * dJ02 +
                (2.f * cam.focal_y * ty) * tz3 * dJ12;
    // depth output: d(view z)/d(mean)
    dtz += gr[G_DEPTH];
    F3 dmean = f3(W00 * dtx + W10 * dty + W20 * dtz, W01 * dtx + W11 * dty + W21 * dtz, W02 * dtx + W12 * dty + W22 * dtz);

    // ---- screen-space mean (NDC) -> mean
    const float* pj = cam.proj;
    float hx = pj[0] * p.x + pj[4] * p.y + pj[8] * p.z + pj[12];
    float hy = pj[1] * p.x + pj[5] * p.y + pj[9] * p.z + pj[13];
    float hw = pj[3] * p.x + pj[7] * p.y + pj[11] * p.z + pj[15];
    float mw = 1.0f / (hw + 0.0000001f);
    float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
    float g2x = gr[G_MX], g2y = gr[G_MY];
    dmean.x += (pj[0] * mw - pj[3] * mul1) * g2x + (pj[1] * mw - pj[3] * mul2) * g2y;
    dmean.y += (pj[4] * mw - pj[7] * mul1) * g2x + (pj[5] * mw - pj[7] * mul2) * g2y;
    dmean.z += (pj[8] * mw - pj[11] * mul1) * g2x + (pj[9] * mw - pj[11] * mul2) * g2y;
    dL_dmeans2D[3 * i] = g2x; dL_dmeans2D[3 * i + 1] = g2y; dL_dmeans2D[3 * i + 2] = 0.f;

    // ---- colour -> SH and mean
    F3 dm_sh = sh_backward<STAGED>(D, M, p, cam.campos, STAGED ? (const float*)osh : shs + (size_t)i * M * 3, g.clamped[i],
                                   f3(gr[G_R], gr[G_G], gr[G_B]), osh);
    dmean = dmean + dm_sh;
    dL_dmeans3D[3 * i] = dmean.x; dL_dmeans3D[3 * i + 1] = dmean.y; dL_dmeans3D[3 * i + 2] = dmean.z;

    // ---- opacity / confidence (blend used opacity * confidence)
    // raw (syn3r_raster_backward_raw): `scales` / `rots` / `opacities` are the trainer's PARAMETERS; the activations are formed
    // again (k_activate's arithmetic, common.h) and the gradients leave through their chain rule (k_activate_bwd's): the same bits
    // as the two-launch route
    float cf = conf ? conf[i] : 1.0f;
    const float op_a = raw ? act_sigmoid(opacities[i]) : opacities[i];
    dL_dopacity[i] = raw ? act_sigmoid_bwd(op_a, gr[G_OP] * cf) : gr[G_OP] * cf;
    if (dL_dconf) dL_dconf[i] = gr[G_OP] * op_a;

    // ---- Sigma = M M^T, M = R S  -> scale, rotation
    float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    float4 q4 = make_float4(rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]);
    float q_inv = 1.0f;
    if (raw) {
        s0 = act_exp(s0); s1 = act_exp(s1); s2 = act_exp(s2);
        q_inv = act_quat_inv_norm(q4);
        q4 = act_quat(q4, q_inv);
    }
    float sx = scale_mod * s0, sy = scale_mod * s1, sz = scale_mod * s2;
    float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
    float R00 = 1.f - 2.f * (qy * qy + qz * qz), R01 = 2.f * (qx * qy - qr * qz), R02 = 2.f * (qx * qz + qr * qy);
    float R10 = 2.f * (qx * qy + qr * qz), R11 = 1.f - 2.f * (qx * qx + qz * qz), R12 = 2.f * (qy * qz - qr * qx);
    float R20 = 2.f * (qx * qz - qr * qy), R21 = 2.f * (qy * qz + qr * qx), R22 = 1.f - 2.f * (qx * qx + qy * qy);
    float m00 = R00 * sx, m01 = R01 * sy, m02 = R02 * sz;
    float m10 = R10 * sx, m11 = R11 * sy, m12 = R12 * sz;
    float m20 = R20 * sx, m21 = R21 * sy, m22 = R22 * sz;
    // full symmetric gradient G (off-diagonals halved); dL/dM = 2 G M
    float G00 = dS0, G11 = dS3, G22 = dS5, G01 = 0.5f * dS1, G02 = 0.5f * dS2, G12 = 0.5f * dS4;
    float dM00 = 2.f * (G00 * m00 + G01 * m10 + G02 * m20), dM01 = 2.f * (G00 * m01 + G01 * m11 + G02 * m21),
          dM02 = 2.f * (G00 * m02 + G01 * m12 + G02 * m22);
    float dM10 = 2.f * (G01 * m00 + G11 * m10 + G12 * m20), dM11 = 2.f * (G01 * m01 + G11 * m11 + G12 * m21),
          dM12 = 2.f * (G01 * m02 + G11 * m12 + G12 * m22);
    float dM20 = 2.f * (G02 * m00 + G12 * m10 + G22 * m20), dM21 = 2.f * (G02 * m01 + G12 * m11 + G22 * m21),
          dM22 = 2.f * (G02 * m02 + G12 * m12 + G22 * m22);
    float ds0 = scale_mod * (dM00 * R00 + dM10 * R10 + dM20 * R20);
    float ds1 = scale_mod * (dM01 * R01 + dM11 * R11 + dM21 * R21);
    float ds2 = scale_mod * (dM02 * R02 + dM12 * R12 + dM22 * R22);
    float dR00 = dM00 * sx, dR01 = dM01 * sy, dR02 = dM02 * sz;
    float dR10 = dM10 * sx, dR11 = dM11 * sy, dR12 = dM12 * sz;
    float dR20 = dM20 * sx, dR21 = dM21 * sy, dR22 = dM22 * sz;
    float4 dq;
    dq.x = 2.f * (-qz * dR01 + qy * dR02 + qz * dR10 - qx * dR12 - qy * dR20 + qx * dR21);
    dq.y = 2.f * (qy * dR01 + qz * dR02 + qy * dR10 - 2.f * qx * dR11 - qr * dR12 + qz * dR20 + qr * dR21 -
                  2.f * qx * dR22);
    dq.z = 2.f * (-2.f * qy * dR00 + qx * dR01 + qr * dR02 + qx * dR10 + qz * dR12 - qr * dR20 + qz * dR21 -
                  2.f * qy * dR22);
    dq.w = 2.f * (-2.f * qz * dR00 - qr * dR01 + qx * dR02 + qr * dR10 - 2.f * qz * dR11 + qy * dR12 +
                  qx * dR20 + qy * dR21);
    if (raw) {
        ds0 *= s0; ds1 *= s1; ds2 *= s2;
        dq = act_quat_bwd(q4, dq, q_inv);
    }
    dL_dscales[3 * i] = ds0; dL_dscales[3 * i + 1] = ds1; dL_dscales[3 * i + 2] = ds2;
    dL_drots[4 * i] = dq.x; dL_drots[4 * i + 1] = dq.y; dL_drots[4 * i + 2] = dq.z; dL_drots[4 * i + 3] = dq.w;
    } while (0);
    if constexpr (STAGED) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int e = (threadIdx.x + 256 * k) * 4, row = e / 48, col = e - row * 48;
            if (row < rows) {
                const float* d = &shl[row * kShLd + col];
                *(float4*)(dL_dshs + blk0 + e) = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

}  // namespace

extern "C" size_t syn3r_raster_backward_workspace_bytes(int N) {
    return N > 0 ? align256((size_t)N * kGradSlots * sizeof(float)) : 0;
}

static int raster_backward(int raw, int N, int sh_degree, int sh_coeffs, long long P, const float* means3D,
                                     const float* scales, const float* rotations, const float* opacities,
                                     const float* shs, const float* confidence, float scale_modifier,
                                     const float* viewmatrix, const float* projmatrix, const float* campos,
                                     float tanfovx, float tanfovy, int H, int W, const float* bg, const int* radii,
                                     void* geom, size_t geom_bytes_, const unsigned* point_list, void* image,
                                     size_t image_bytes_, const float* dL_dcolor, const float* dL_ddepth,
                                     const float* dL_dalpha, float* dL_dmeans3D, float* dL_dscales,
                                     float* dL_drotations, float* dL_dopacities, float* dL_dshs, float* dL_dmeans2D,
                                     float* dL_dconfidence, void* workspace, size_t workspace_bytes, void* stream_) {
    SYN3R_REQUIRE(N > 0 && H > 0 && W > 0 && P >= 0, "raster_backward: bad sizes N=%d H=%d W=%d P=%lld", N, H, W, P);
    SYN3R_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && sh_coeffs >= (sh_degree + 1) * (sh_degree + 1),
                  "raster_backward: bad SH configuration");
    SYN3R_REQUIRE(means3D && scales && rotations && opacities && shs && viewmatrix && projmatrix && campos && bg && radii,
                  "raster_backward: null input");
    SYN3R_REQUIRE(dL_dcolor && dL_dmeans3D && dL_dscales && dL_drotations && dL_dopacities && dL_dshs && dL_dmeans2D,
                  "raster_backward: null gradient buffer");
    SYN3R_REQUIRE(P == 0 || point_list, "raster_backward: point list required");
    size_t need = syn3r_raster_backward_workspace_bytes(N);
    if (!geom || geom_bytes_ < geom_bytes(N) || !image || image_bytes_ < image_bytes(H, W) || !workspace ||
        workspace_bytes < need) {
        set_error("raster_backward: state/workspace buffer too small");
        return SYN3R_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    GeomState g = carve_geom(geom, N);
    ImageState im = carve_image(image, H, W);
    Camera cam;
    raster_fill_camera(cam, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W);
    float* grad_rec = (float*)workspace;
    int rc = check_hip(hipMemsetAsync(grad_rec, 0, need, stream), "memset grads");
    if (rc) return rc;
    const unsigned tiles = (unsigned)(cam.grid_x * cam.grid_y);
    if (P > 0)
        SYN3R_LAUNCH(k_render_bwd, dim3(tiles), dim3(kBwdThreads), 0, stream, H, W, cam.grid_x, cam.grid_y, im.ranges,
                           point_list, g.splats, bg[0], bg[1], bg[2], im.n_contrib, im.final_T, dL_dcolor, dL_ddepth,
                           dL_dalpha, grad_rec, (const unsigned*)(raster_tiles_ordered(N, cam.grid_x, cam.grid_y) ? im.tile_order : nullptr));
    const bool staged = sh_coeffs == 16 && ((((uintptr_t)shs) | ((uintptr_t)dL_dshs)) & 15) == 0;
    if (staged)
        SYN3R_LAUNCH(k_preprocess_bwd<true>, dim3(ceil_div(N, 256)), dim3(256), 0, stream, N, sh_degree, sh_coeffs, means3D,
                       scales, rotations, opacities, shs, confidence, scale_modifier, cam, radii, g, grad_rec,
                       dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacities, dL_dshs, dL_dmeans2D, dL_dconfidence, raw);
    else
    SYN3R_LAUNCH(k_preprocess_bwd<false>, dim3(ceil_div(N, 256)), dim3(256), 0, stream, N, sh_degree, sh_coeffs, means3D,
                       scales, rotations, opacities, shs, confidence, scale_modifier, cam, radii, g, grad_rec,
                       dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacities, dL_dshs, dL_dmeans2D, dL_dconfidence, raw);
    SYN3R_LAUNCH_CHECK("raster_backward launch");
    return SYN3R_OK;
}

extern "C" int syn3r_raster_backward(int N, int sh_degree, int sh_coeffs, long long P, const float* means3D,
                                     const float* scales, const float* rotations, const float* opacities,
                                     const float* shs, const float* confidence, float scale_modifier,
                                     const float* viewmatrix, const float* projmatrix, const float* campos,
                                     float tanfovx, float tanfovy, int H, int W, const float* bg, const int* radii,
                                     void* geom, size_t geom_bytes_, const unsigned* point_list, void* image,
                                     size_t image_bytes_, const float* dL_dcolor, const float* dL_ddepth,
                                     const float* dL_dalpha, float* dL_dmeans3D, float* dL_dscales,
                                     float* dL_drotations, float* dL_dopacities, float* dL_dshs, float* dL_dmeans2D,
                                     float* dL_dconfidence, void* workspace, size_t workspace_bytes, void* stream_) {
    return raster_backward(0, N, sh_degree, sh_coeffs, P, means3D, scales, rotations, opacities, shs, confidence, scale_modifier,
                           viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, bg, radii, geom, geom_bytes_, point_list, image,
                           image_bytes_, dL_dcolor, dL_ddepth, dL_dalpha, dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacities,
                           dL_dshs, dL_dmeans2D, dL_dconfidence, workspace, workspace_bytes, stream_);
}

extern "C" int syn3r_raster_backward_raw(int N, int sh_degree, int sh_coeffs, long long P, const float* means3D,
                                         const float* log_scales, const float* raw_rotations, const float* opacity_logits,
                                         const float* shs, const float* confidence, float scale_modifier,
                                         const float* viewmatrix, const float* projmatrix, const float* campos,
                                         float tanfovx, float tanfovy, int H, int W, const float* bg, const int* radii,
                                         void* geom, size_t geom_bytes_, const unsigned* point_list, void* image,
                                         size_t image_bytes_, const float* dL_dcolor, const float* dL_ddepth,
                                         const float* dL_dalpha, float* dL_dmeans3D, float* dL_dlog_scales,
                                         float* dL_draw_rotations, float* dL_dopacity_logits, float* dL_dshs, float* dL_dmeans2D,
                                         float* dL_dconfidence, void* workspace, size_t workspace_bytes, void* stream_) {
    return raster_backward(1, N, sh_degree, sh_coeffs, P, means3D, log_scales, raw_rotations, opacity_logits, shs, confidence,
                           scale_modifier, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, bg, radii, geom, geom_bytes_,
                           point_list, image, image_bytes_, dL_dcolor, dL_ddepth, dL_dalpha, dL_dmeans3D, dL_dlog_scales,
                           dL_draw_rotations, dL_dopacity_logits, dL_dshs, dL_dmeans2D, dL_dconfidence, workspace, workspace_bytes,
                           stream_);
}

#ifdef SYN3R_RASTER_STATS
extern "C" __attribute__((visibility("default"))) int syn3r_debug_bwd_stats(unsigned long long* out4, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_bwd_stats), sizeof(unsigned long long) * 4);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), z, sizeof(z)); }
    return rc;
}
#endif
