// apd_device.h -- device-side data model and arithmetic of the MI355X PatchMatch path.
//
// Written for gfx950 only (wave64, no texture unit: bilinear sampling is explicit global loads).
// Build flags that are part of the contract: -ffp-contract=off, no fast-math (the algorithm relies
// on NaN comparisons being false, APD.cu:149 / SURVEY Appendix A #9).  fmaf() appears exactly where
// DESIGN.md's arithmetic contract says; everything else is literal IEEE binary32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/apd_mi355x.h"
#include "apd_tuning.h"
#include "apd_lab.h"

namespace apd {

// ------------------------------------------------------------------------------------------------
// data model
// ------------------------------------------------------------------------------------------------

// One fetched dword = the four byte taps of one bilinear fetch (layouts below); bit-identical to the float sampler (same taps,
// same three fmaf).
typedef uint32_t quad_t;
constexpr int kQuadShift = 2;
constexpr unsigned kQuadBytes = 1u << kQuadShift;

// Row-major copy of the 8-bit source images ("texel quads").  Default: 2-byte COLUMN PAIRS -- entry (t, u) = (qx + 1, qy + 1)
// holds {I(qx, qy), I(qx, qy + 1)} (clamped coordinates) and the dword at byte 2 * (u * (W + 2) + t) is therefore
// {I(qx,qy), I(qx,qy+1), I(qx+1,qy), I(qx+1,qy+1)}: all four taps of a bilinear fetch in ONE gather, as with 4-byte quads, at
// half the footprint (twice the pixels per 128-byte line, per L1, per L2).  Half of these gathers are not dword aligned,
// which costs nothing on gfx950 (tools/unaligned_gather.hip: 18.67 vs 18.85 ms for 1.07e9 lines).  -DAPD_QUAD4 builds the
// round-1 layout (4-byte entries {I(qx,qy), I(qx+1,qy), I(qx,qy+1), I(qx+1,qy+1)}, pitch W + 1) for A/B runs.
// The tiled copy (quad_tiled_offset_tu) is laid out so that its dwords have the byte order of the row-major dword, so there
// is one decode: kPair2: bytes {t00, t01, t10, t11}, else {t00, t10, t01, t11}.
#ifndef APD_QUAD4
constexpr bool kPair2 = true;
#else
constexpr bool kPair2 = false;
#endif
constexpr int kRowEntryShift = kPair2 ? 1 : 2;
constexpr unsigned kRowEntryBytes = 1u << kRowEntryShift;
__host__ __device__ __forceinline__ unsigned quad_row_pitch_bytes(int W) { return kPair2 ? 2u * (unsigned)(W + 2) : 4u * (unsigned)(W + 1); }
// + 4: the dword of the last entry reads two bytes past it
__host__ __device__ __forceinline__ size_t quad_image_bytes(int W, int H) { return (size_t)quad_row_pitch_bytes(W) * (size_t)(H + 1) + 4u; }

// One texel pair: the texel and the (rounded) difference to its right neighbour -- the two numbers the horizontal
// lerp fmaf(a, t10 - t00, t00) needs, as binary32.  A float texel quad is the pair of a texel and the pair of the texel
// below it: same taps, same three fused multiply-adds as a four-tap float sampler, one 16-byte gather instead of four
// 4-byte ones and no subtractions in the sample loop.
typedef float pair_t __attribute__((ext_vector_type(2)));
typedef float fquad_t __attribute__((ext_vector_type(4)));

// Per source view constants.  Rr/tr are the plane-independent part of ComputeHomography
// (APD.cu:305-331) hoisted to the host once per (reference, source) pair; the kernels read them
// through wave-uniform (scalar) loads.
struct ViewConst {
    float Rr[9];       // R_src * R_ref^T
    float tr[3];       // R_src * (C_ref - C_src)
    float k0, k2, k4, k5, k8;  // source intrinsics used by :354-362
    float wf, hf;      // (float)width, (float)height of the source camera (centre test :546)
    float K[9];        // full source camera for the geometric term (:740-750)
    float R[9];
    float t[3];
    float c[3];
    const float *img;    // W*H floats
    const float *depth;  // W*H floats or nullptr
    // Texel-quad image (only when every pixel of every view is an integer 0..255, i.e. 8-bit input at
    // scale 1): entry (qx, qy), qx in [-1, W-1], qy in [-1, H-1], holds the four clamped taps
    // {I(qx,qy), I(qx+1,qy), I(qx,qy+1), I(qx+1,qy+1)} of one bilinear fetch (quad_t below).
    const quad_t *quad;  // (H+1)*(W+1) entries or nullptr
    const quad_t *quad_tiled;  // the same texels in 128-byte tiles (quad_tiled_offset_tu) or nullptr
    // Float texel-quad image (every other input: float grey values, e.g. the resampled images of the coarse pyramid
    // levels, APD.cpp:474): entry (qx, qy), qx in [-1, W-1], qy in [-1, H-1], holds the two texel pairs
    // {I(qx,qy), I(qx+1,qy) - I(qx,qy), I(qx,qy+1), I(qx+1,qy+1) - I(qx,qy+1)} with clamped coordinates (fquad_t below):
    // one 16-byte gather per bilinear fetch.
    const fquad_t *fquad;  // (H+1)*(W+1) entries or nullptr
};

struct FrameArgs {
    int W, H;
    int num_src;       // num_images - 1
    int half_rows;     // rows reachable by the reference HALF launch: 2*ceil((H/2)/16)*16 (APD.cu:2402)
    int use_quads;     // 1: source images are also available as texel quads (ViewConst::quad)
    int have_tiled;    // 1: ... and as the tiled copy (ViewConst::quad_tiled)
    int approx_rcp;    // 1: tolerance mode APD_OPT_FAST_RCP (K6/K7 sample loops stop at v_rcp_f32; results are NOT the oracle's bits)
    // host-side dispatch switches (apd_set_option): which kernels launch_kernel picks; same results
    int tiled_mode;    // APD_OPT_TILED_COPY
    int k67_windows, k1415_windows;
    // params (main.h:75-94)
    int top_k;
    float depth_min, depth_max;
    int geom_consistency;
    int weak_peak_radius;
    int rotate_time;
    float ransac_threshold;
    float geom_factor;
    int state;
    unsigned long long seed;
    // K3 constants evaluated on the host in double (APD.cu:1791-1795)
    float k3_cos_angle, k3_sin_angle, k3_cone;
    int k3_shift_range;
    // K3's inlier test `dist / (depth_max - depth_min) < ransac_threshold` as `dist < k3_dist_cut` (exact; ransac_distance_cut,
    // apd_capi.hip); k3_cut_valid = 0: no such cut exists for these parameters, the kernel divides
    float k3_dist_cut;
    int k3_cut_valid;
    // reference camera
    float K[9], R[9], t[3], c[3];
    float ifx, ify;    // correctly rounded 1/K[0], 1/K[4]
    // state
    const float *ref_img;
    const ViewConst *views;  // [num_src], view v == source image v+1
    float4 *planes;
    float4 *fit_planes;
    float *costs;
    uint32_t *rng;           // 6 words per pixel
    uint32_t *selected_views;
    uint8_t *view_weight;    // 32 per pixel
    uint8_t *weak_info;
    uint8_t *weak_reliable;
    short2 *nearest_strong;
    int8_t *column_nearest;  // K2 scratch: dy of the nearest STRONG pixel in the same column, 127 = none within 100
    const int *neighbours_map;
    short2 *neighbours;      // 9 per weak pixel
    int early_out;           // 1 (default): the exact early-outs of the refinement loops, K14 and K15; APD_OPT_EARLY_OUT = 0
                             // evaluates every NCC the reference evaluates (A/B runs and the parity test; same results)
};

// The per-view constants are read-only for every kernel and the view index is wave-uniform wherever it is used: reading the
// table through the constant address space lets the compiler use scalar loads (s_load_dword*) into SGPRs.  Through the
// generic pointer of FrameArgs it cannot prove that the kernel's own stores leave the table alone, and every NCC re-read its
// 20-odd fields with vector loads (one per lane) and waited for them (round 2: 2.6e6 scalar against 3.8e8 vector loads
// per K6/K7 launch).
typedef const __attribute__((address_space(4))) ViewConst *const_view_ptr;
__device__ __forceinline__ const ViewConst &view_const(const FrameArgs &fa, int v)
{
    return *(const ViewConst *)((const_view_ptr)(uintptr_t)fa.views + v);
}


// ------------------------------------------------------------------------------------------------
// polynomial kernels (contract C5) -- same coefficients and operation order as the oracle
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ float sin_poly(float x)
{
    const float z = x * x;
    float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p * z, x, x);
}

__device__ __forceinline__ float cos_poly(float x)
{
    const float z = x * x;
    float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}

__device__ __forceinline__ float exp_poly(float x)
{
    if (!(x > -87.0f)) {
        return (x != x) ? x : 0.0f;
    }
    if (x > 88.0f) {
        return __builtin_inff();
    }
    const float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, r * r, r) + 1.0f;
    return p * __uint_as_float((uint32_t)((int)n + 127) << 23);
}

__device__ __forceinline__ float rsqrt_c4(float x) { return 1.0f / sqrtf(x); }

// Correctly rounded reciprocal (contract C3) without the IEEE division sequence: v_rcp_f32 (<= 1 ulp)
// followed by one FMA Newton step equals RN(1/z) bit for bit for every z whose biased exponent lies in
// 27..227 -- checked over all 2^32 inputs on gfx950 by tools/valu_rates.hip.  Outside that range (tiny,
// huge, zero, Inf, NaN) the IEEE division is taken; `ok` is meant to be hoisted out of sample loops.
__device__ __forceinline__ bool recip_fast_ok(float z) { return fabsf(z) >= 0x1p-100f && fabsf(z) <= 0x1p100f; }

__device__ __forceinline__ float recip_fast(float z)
{
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = fmaf(-z, r, 1.0f);
    return fmaf(e, r, r);
}

__device__ __forceinline__ float recip_rn(float z, bool ok) { return __builtin_expect(ok, 1) ? recip_fast(z) : 1.0f / z; }
__device__ __forceinline__ float recip_rn(float z) { return recip_rn(z, recip_fast_ok(z)); }

// The two IEEE operations every NCC ends with (contract C1 / C4: sqrtf correctly rounded, `/` an IEEE division), without the
// range scaling the compiler's general sequences carry (27 instead of 38 instructions per NCC epilogue; K9/K10 runs nine of them
// per NCCNew on two waves per SIMD, where every instruction of the chain costs its full latency):
//   sqrt_rn_mid(x)   = v_sqrt_f32 (<= 1 ulp) and the two one-ulp probes of the compiler's own sequence, whose range scaling only
//                      matters below 2^-96; equal to sqrtf for every binary32 x with biased exponent 31..223 (all of them checked)
//                      and for NaN;
//   div_rn_mid(a, b) = q0 = a * y with y = RN(1 / b) (recip_fast), then twice q <- q + (a - b q) y with the residual exact (fma): the
//                      first step makes q a faithful rounding of a / b, and from a faithful q and the correctly rounded reciprocal
//                      the second gives the correctly rounded quotient (Markstein 1990) as long as nothing over- or underflows --
//                      the steps of the compiler's own sequence without its range scaling; 2^32 random pairs over the operand
//                      ranges of an NCC checked against `/`, zero included.
// tools/valu_rates.hip --check runs both comparisons on the device (tests/test_gpu_edge_cases.py::test_isa_contract_exhaustive).
// Callers guarantee the ranges: x = var_ref * var_src with both in [1e-5, 1.7e4], b = sqrt of that, |a| <= 6.6e4.
__device__ __forceinline__ float sqrt_rn_mid(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = fmaf(-below, s, x), r_above = fmaf(-above, s, x);
    float r = (r_below <= 0.0f) ? below : s;
    r = (r_above > 0.0f) ? above : r;
    return r;
#else
    return sqrtf(x);
#endif
}

__device__ __forceinline__ float div_rn_mid(float a, float b)
{
    const float y = recip_fast(b);
    const float q0 = a * y;                      // within 1.5 ulp of a / b
    const float q1 = fmaf(fmaf(-b, q0, a), y, q0);  // a faithful rounding of a / b (the residual is exact)
    return fmaf(fmaf(-b, q1, a), y, q1);         // Markstein: faithful q1 and y = RN(1 / b) -> the correctly rounded quotient
}

// Tail of every NCC (APD.cu:585-613 and the sub-patch terms of :461-505): 1 - covariance / sqrt(var_ref * var_src), clamped to [0, 2].
// The callers have tested both variances against 1e-5 already.
__device__ __forceinline__ float ncc_cost_from_moments(float var_r, float var_s, float covar)
{
    const float denom = sqrt_rn_mid(var_r * var_s);
    return fmaxf(0.0f, fminf(2.0f, 1.0f - div_rn_mid(covar, denom)));
}

// ------------------------------------------------------------------------------------------------
// XORWOW (contract C8): state kept in six registers; AoS of 6 words per pixel in HBM
// ------------------------------------------------------------------------------------------------

struct Rng {
    uint32_t x0, x1, x2, x3, x4, d;
};

__device__ __forceinline__ Rng rng_load(const uint32_t *base, int center)
{
    const uint2 *p = reinterpret_cast<const uint2 *>(base + 6 * (size_t)center);
    const uint2 a = p[0], b = p[1], c = p[2];
    return Rng{a.x, a.y, b.x, b.y, c.x, c.y};
}

__device__ __forceinline__ void rng_store(uint32_t *base, int center, const Rng &r)
{
    uint2 *p = reinterpret_cast<uint2 *>(base + 6 * (size_t)center);
    p[0] = make_uint2(r.x0, r.x1);
    p[1] = make_uint2(r.x2, r.x3);
    p[2] = make_uint2(r.x4, r.d);
}

__device__ __forceinline__ uint32_t rng_next(Rng &r)
{
    const uint32_t t = r.x0 ^ (r.x0 >> 2);
    r.x0 = r.x1;
    r.x1 = r.x2;
    r.x2 = r.x3;
    r.x3 = r.x4;
    r.x4 = (r.x4 ^ (r.x4 << 4)) ^ (t ^ (t << 1));
    r.d += 362437u;
    return r.d + r.x4;
}

__device__ __forceinline__ float rng_uniform(Rng &r)
{
    const float v = (float)rng_next(r);
    return 2.3283064e-10f + (v * 2.3283064e-10f);
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int bit_test(uint32_t v, unsigned n) { return (int)((v >> n) & 1u); }
// APD.cu:47-50: clears bit n and every lower bit (reference behaviour, kept).
__device__ __forceinline__ uint32_t bit_unset_quirk(uint32_t v, unsigned n) { return v & (uint32_t)(0xFFFFFFFEu << n); }

__device__ __forceinline__ bool inside(const FrameArgs &fa, int x, int y)
{
    return x >= 0 && y >= 0 && x < fa.W && y < fa.H;
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
    const float n2 = x * x + y * y + z * z;
    const float inv = rsqrt_c4(n2);
    x *= inv;
    y *= inv;
    z *= inv;
}

__device__ __forceinline__ void normalize2(float &x, float &y)
{
    const float n2 = x * x + y * y;
    const float inv = rsqrt_c4(n2);
    x *= inv;
    y *= inv;
}

// Get3DPoint (APD.cu:159-171) in the reference camera
__device__ __forceinline__ void point3d(const FrameArgs &fa, int px, int py, float depth, float &X, float &Y, float &Z)
{
    X = depth * ((float)px - fa.K[2]) / fa.K[0];
    Y = depth * ((float)py - fa.K[5]) / fa.K[4];
    Z = depth;
}

// GetViewDirection (APD.cu:173-185)
__device__ __forceinline__ void view_direction(const FrameArgs &fa, int px, int py, float depth, float &vx, float &vy, float &vz)
{
    float X, Y, Z;
    point3d(fa, px, py, depth, X, Y, Z);
    const float norm = sqrtf(X * X + Y * Y + Z * Z);
    vx = X / norm;
    vy = Y / norm;
    vz = Z / norm;
}

// GetDistance2Origin (APD.cu:187-192)
__device__ __forceinline__ float distance_to_origin(const FrameArgs &fa, int px, int py, float depth, float nx, float ny, float nz)
{
    float X, Y, Z;
    point3d(fa, px, py, depth, X, Y, Z);
    return -(nx * X + ny * Y + nz * Z);
}

// ComputeDepthfromPlaneHypothesis (APD.cu:206-209)
__device__ __forceinline__ float depth_from_plane(const FrameArgs &fa, const float4 pl, int px, int py)
{
    return -pl.w * fa.K[0] /
           (((float)px - fa.K[2]) * pl.x + (fa.K[0] / fa.K[4]) * ((float)py - fa.K[5]) * pl.y + fa.K[0] * pl.z);
}

// TransformNormal (APD.cu:374-382): camera -> world
__device__ __forceinline__ float4 normal_cam_to_world(const FrameArgs &fa, const float4 p)
{
    float4 o;
    o.x = fa.R[0] * p.x + fa.R[3] * p.y + fa.R[6] * p.z;
    o.y = fa.R[1] * p.x + fa.R[4] * p.y + fa.R[7] * p.z;
    o.z = fa.R[2] * p.x + fa.R[5] * p.y + fa.R[8] * p.z;
    o.w = p.w;
    return o;
}

// TransformNormal2RefCam (APD.cu:384-392): world -> camera
__device__ __forceinline__ float4 normal_world_to_cam(const FrameArgs &fa, const float4 p)
{
    float4 o;
    o.x = fa.R[0] * p.x + fa.R[1] * p.y + fa.R[2] * p.z;
    o.y = fa.R[3] * p.x + fa.R[4] * p.y + fa.R[5] * p.z;
    o.z = fa.R[6] * p.x + fa.R[7] * p.y + fa.R[8] * p.z;
    o.w = p.w;
    return o;
}

// ------------------------------------------------------------------------------------------------
// hypothesis generation (APD.cu:211-282)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ float4 random_normal(const FrameArgs &fa, int px, int py, Rng &rng, float depth)
{
    float q1 = 1.0f, q2 = 1.0f, s = 2.0f;
    while (s >= 1.0f) {
        q1 = 2.0f * rng_uniform(rng) - 1.0f;
        q2 = 2.0f * rng_uniform(rng) - 1.0f;
        s = q1 * q1 + q2 * q2;
    }
    const float sq = sqrtf(1.0f - s);
    float nx = 2.0f * q1 * sq;
    float ny = 2.0f * q2 * sq;
    float nz = 1.0f - 2.0f * s;
    float vx, vy, vz;
    view_direction(fa, px, py, depth, vx, vy, vz);
    const float dot = nx * vx + ny * vy + nz * vz;
    if (dot > 0.0f) {
        nx = -nx;
        ny = -ny;
        nz = -nz;
    }
    normalize3(nx, ny, nz);
    return make_float4(nx, ny, nz, 0.0f);
}

__device__ __forceinline__ float4 perturbed_normal(const FrameArgs &fa, int px, int py, const float4 normal, Rng &rng, float perturbation)
{
    float vx, vy, vz;
    view_direction(fa, px, py, 1.0f, vx, vy, vz);
    const float a1 = (rng_uniform(rng) - 0.5f) * perturbation;
    const float a2 = (rng_uniform(rng) - 0.5f) * perturbation;
    const float a3 = (rng_uniform(rng) - 0.5f) * perturbation;
    const float s1 = sin_poly(a1), s2 = sin_poly(a2), s3 = sin_poly(a3);
    const float c1 = cos_poly(a1), c2 = cos_poly(a2), c3 = cos_poly(a3);
    const float R0 = c2 * c3;
    const float R1 = c3 * s1 * s2 - c1 * s3;
    const float R2 = s1 * s3 + c1 * c3 * s2;
    const float R3 = c2 * s3;
    const float R4 = c1 * c3 + s1 * s2 * s3;
    const float R5 = c1 * s2 * s3 - c3 * s1;
    const float R6 = -s2;
    const float R7 = c2 * s1;
    const float R8 = c1 * c2;
    float4 p;
    p.x = R0 * normal.x + R1 * normal.y + R2 * normal.z;
    p.y = R3 * normal.x + R4 * normal.y + R5 * normal.z;
    p.z = R6 * normal.x + R7 * normal.y + R8 * normal.z;
    p.w = 0.0f;
    if (p.x * vx + p.y * vy + p.z * vz >= 0.0f) {
        p = normal;
    }
    normalize3(p.x, p.y, p.z);
    return p;
}

__device__ __forceinline__ float4 random_plane(const FrameArgs &fa, int px, int py, Rng &rng)
{
    const float depth = rng_uniform(rng) * (fa.depth_max - fa.depth_min) + fa.depth_min;
    float4 pl = random_normal(fa, px, py, rng, depth);
    pl.w = distance_to_origin(fa, px, py, depth, pl.x, pl.y, pl.z);
    return pl;
}

// ------------------------------------------------------------------------------------------------
// homography (APD.cu:333-362 on top of the hoisted relative pose), contract C2/C3
// ------------------------------------------------------------------------------------------------

struct Homography {
    float h[9];
};

// q = n / d of the plane hypothesis (three multiplies by a correctly rounded 1/d)
__device__ __forceinline__ void plane_q(const float4 pl, float &qx, float &qy, float &qz)
{
    const float inv_w = recip_rn(pl.w);
    qx = pl.x * inv_w;
    qy = pl.y * inv_w;
    qz = pl.z * inv_w;
}

__device__ __forceinline__ Homography make_homography(const FrameArgs &fa, const ViewConst &vc, float qx, float qy, float qz)
{
    Homography o;
    float T[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float m0 = fmaf(-vc.tr[r], qx, vc.Rr[3 * r + 0]);
        const float m1 = fmaf(-vc.tr[r], qy, vc.Rr[3 * r + 1]);
        const float m2 = fmaf(-vc.tr[r], qz, vc.Rr[3 * r + 2]);
        T[3 * r + 0] = m0 * fa.ifx;
        T[3 * r + 1] = m1 * fa.ify;
        T[3 * r + 2] = fmaf(-T[3 * r + 1], fa.K[5], fmaf(-T[3 * r + 0], fa.K[2], m2));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.h[0 + c] = fmaf(vc.k2, T[6 + c], vc.k0 * T[0 + c]);
        o.h[3 + c] = fmaf(vc.k5, T[6 + c], vc.k4 * T[3 + c]);
        o.h[6 + c] = vc.k8 * T[6 + c];
    }
    return o;
}

// ComputeCorrespondingPoint (APD.cu:365-372)
__device__ __forceinline__ void correspond(const Homography &H, float xf, float yf, float &ox, float &oy)
{
    const float X = fmaf(H.h[1], yf, fmaf(H.h[0], xf, H.h[2]));
    const float Y = fmaf(H.h[4], yf, fmaf(H.h[3], xf, H.h[5]));
    const float Z = fmaf(H.h[7], yf, fmaf(H.h[6], xf, H.h[8]));
    const float inv = recip_rn(Z);
    ox = X * inv;
    oy = Y * inv;
}

// ------------------------------------------------------------------------------------------------
// sampler (contract C7): software replacement of tex2D(linear, clamp); gfx950 has no texture path
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ float fetch_texel(const float *__restrict__ img, int W, int H, int x, int y)
{
    return img[(unsigned)(clampi(y, 0, H - 1) * W + clampi(x, 0, W - 1))];
}

template <typename Ptr>
__device__ __forceinline__ float sample_bilinear(Ptr img, int W, int H, float sx, float sy)
{
    const float fx = floorf(sx), fy = floorf(sy);
    const float a = sx - fx, b = sy - fy;
    const int x0 = (int)fminf(fmaxf(fx, -1.0f), (float)W);
    const int y0 = (int)fminf(fmaxf(fy, -1.0f), (float)H);
    const int xa = clampi(x0, 0, W - 1), xb = min(x0 + 1, W - 1);
    const int ya = clampi(y0, 0, H - 1), yb = min(y0 + 1, H - 1);
    const unsigned ra = (unsigned)__mul24(ya, W), rb = (unsigned)__mul24(yb, W);
    const float t00 = img[ra + (unsigned)xa], t10 = img[ra + (unsigned)xb];
    const float t01 = img[rb + (unsigned)xa], t11 = img[rb + (unsigned)xb];
    const float top = fmaf(a, t10 - t00, t00);
    const float bot = fmaf(a, t11 - t01, t01);
    return fmaf(b, bot - top, top);
}

// Pointers into HBM as the compiler should see them: global address space (a pointer loaded from a
// struct would otherwise be "generic" and cost flat_load + 64-bit address arithmetic per gather).
typedef const __attribute__((address_space(1))) quad_t *global_quad_ptr;
typedef const __attribute__((address_space(1))) float *global_f32_ptr;
typedef const __attribute__((address_space(1))) pair_t *global_pair_ptr;
typedef const __attribute__((address_space(1))) fquad_t *global_fquad_ptr;

__device__ __forceinline__ fquad_t fquad_fetch(global_fquad_ptr fq, unsigned off)
{
    return *(global_fquad_ptr)((const __attribute__((address_space(1))) char *)fq + off);
}

// Same fetch from the texel-quad image: one gather instead of four.  Bit-identical to
// sample_bilinear on 8-bit data (the taps are the same floats, the lerp is the same three fmaf).
// A NaN/Inf coordinate gives a NaN weight, so the sample is NaN whatever texel is read and the index clamp only
// has to keep the address in range.  Fetch and interpolation are separate so a caller can put several gathers
// in flight before consuming the first one (quad_row_issue / quad_row_lerp, subpatch_cost_quad).
typedef quad_t quad_unaligned_t __attribute__((aligned(2)));
__device__ __forceinline__ quad_t quad_fetch(global_quad_ptr quad, unsigned off)
{
    // one global_load_dword; with 2-byte column pairs the address is only 2-byte aligned
    return *(const __attribute__((address_space(1))) quad_unaligned_t *)((const __attribute__((address_space(1))) char *)quad + off);
}

// fmaf(a, (float)hi16(p), (float)lo16(p)) in one instruction: p = {binary16 base, binary16 delta}
__device__ __forceinline__ float lerp_f16_pair(float a, uint32_t p)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(p));
    return r;
}

__device__ __forceinline__ float quad_lerp(quad_t t, float a, float b)
{
    float t00, t10, t01, t11;  // four v_cvt_f32_ubyte<k>; the differences below stay binary32 subtractions (see quad_row_lerp)
    asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t00) : "v"(t));
    asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(t11) : "v"(t));
    if (kPair2) {
        asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t01) : "v"(t));
        asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t10) : "v"(t));
    } else {
        asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t10) : "v"(t));
        asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t01) : "v"(t));
    }
    const float top = fmaf(a, t10 - t00, t00);
    const float bot = fmaf(a, t11 - t01, t01);
    return fmaf(b, bot - top, top);
}

// Bilinear tap position for the texel-quad image, three VALU instructions per axis:
//   weight  = v_fract_f32(s)        == s - floor(s) for every s >= 0; for s < 0 it can differ in the last bit
//                                      (clamped below 1), but there both taps are the same clamped edge texel,
//                                      so the sample is bit-identical; Inf/NaN give NaN like the subtraction
//   index   = v_cvt_flr_i32_f32(s)  == saturating (int)floor(s) (NaN -> INT_MAX; the sample is NaN anyway)
//   clamped = v_med3_i32(index, -1, size - 1)
// (all three checked over all 2^32 inputs by tools/valu_rates.hip)
__device__ __forceinline__ int cvt_floor_i32(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ int med3_i32(int x, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

// Second, TILED copy of the 8-bit source images for gathers that land anywhere in the source image (K5 and the first iteration
// of a FIRST_INIT pass: every lane warps its patch with another random plane).  One tile = one 128-byte line:
//   kPair2 (default): 7 columns x 8 rows of 2-byte column pairs, each tile row padded with the first pair of the next tile
//     (8 entries = 16 bytes per row), so that the dword at any entry still holds entries t and t + 1, i.e. the four taps;
//     an 11 x 11 px warped patch touches ~5.5 tiles (row-major pairs: ~7 lines of 64 x 1 px; round 1's row-major 4-byte quads: ~14);
//   APD_QUAD4: 8 x 4 four-byte quads (~7.9 tiles per patch; measured against the row-major quads in profiles/r02/tiled_vs_rowmajor.txt:
//     L1 -> L2 requests per gather 16.5 -> 9.7, first black launch of configs[1] 129 -> 94 ms).
// Rows are read better from the row-major copy (window staging, the 3 x 3 stride-5 sub-patches of the weak sweep), which stays.
// Entry (t, u) = (qx + 1, qy + 1).  Same texels, same arithmetic: same bits.
constexpr unsigned kTileCols = kPair2 ? 7u : 8u, kTileRows = kPair2 ? 8u : 4u;
__host__ __device__ __forceinline__ unsigned quad_tiles_x(int W)
{
    return kPair2 ? ((unsigned)(W + 2) + kTileCols - 1u) / kTileCols : ((unsigned)(W + 1) + 7u) >> 3;
}
__host__ __device__ __forceinline__ unsigned quad_tiles_y(int H) { return ((unsigned)(H + 1) + kTileRows - 1u) / kTileRows; }
__host__ __device__ __forceinline__ size_t quad_tiled_bytes(int W, int H) { return (size_t)quad_tiles_x(W) * quad_tiles_y(H) * 128u + 4u; }
// byte offset of entry (t, u); t / 7 as a multiply-shift: (t * 37450) >> 18 is exact for t < 43,690 and both factors fit the
// 24-bit multiplier for the 16,384-px widest image apd_create accepts
__host__ __device__ __forceinline__ unsigned quad_tiled_offset_tu(unsigned t, unsigned u, unsigned tiles_x)
{
    if (kPair2) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(APD_TILE_OFFSET_PLAIN)
        // the same value in nine instructions, five of the 4-cycle class (24-bit multiply-adds, shift-adds; every factor is
        // below 2^24), where the compiler's choice for the plain expression is twelve with seven of that class
        unsigned q, t2, r2, uq, ur, tile, off;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(q) : "v"(t), "v"(37450u));
        asm("v_lshrrev_b32 %0, 18, %1" : "=v"(q) : "v"(q));                            // t / 7
        asm("v_add_u32 %0, %1, %1" : "=v"(t2) : "v"(t));
        asm("v_mad_i32_i24 %0, %1, -14, %2" : "=v"(r2) : "v"(q), "v"(t2));              // 2 * (t - 7 q)
        asm("v_lshrrev_b32 %0, 3, %1" : "=v"(uq) : "v"(u));
        asm("v_and_b32 %0, 7, %1" : "=v"(ur) : "v"(u));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(tile) : "v"(uq), "v"(tiles_x), "v"(q));
        asm("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(off) : "v"(tile), "v"(r2));
        asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(off) : "v"(ur), "v"(off));
        return off;
#else
        const unsigned q = (t * 37450u) >> 18;  // t / 7
        const unsigned r = t - 7u * q;
        return (((u >> 3) * tiles_x + q) << 7) | ((u & 7u) << 4) | (r << 1);
#endif
    }
    return ((((u >> 2) * tiles_x + (t >> 3)) << 5) | ((u & 3u) << 3) | (t & 7u)) << 2;
}
__device__ __forceinline__ unsigned quad_tiled_byte_offset(int qx, int qy, unsigned tiles_x)
{
    return quad_tiled_offset_tu((unsigned)(qx + 1), (unsigned)(qy + 1), tiles_x);
}

// byte offset of quad entry (qx, qy), qx in [-1, W-1], qy in [-1, H-1]: qy*pitch + (pitch + entry) + entry*qx with
// pitch = (W+1)*entry bytes, two instructions
__device__ __forceinline__ unsigned quad_byte_offset(int qx, int qy, int pitch, int origin)
{
    int row, off;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row) : "v"(qy), "v"(pitch), "v"(origin));
    asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(off) : "v"(qx), "v"(row), "n"(kRowEntryShift));
    return (unsigned)off;
}

// ------------------------------------------------------------------------------------------------
// fixed 6x6 patch (strong_radius 5, strong_increment 2): the hot NCC of APD.cu:530-614
// ------------------------------------------------------------------------------------------------

constexpr int kPatchN = 6;       // samples per axis: offsets -5,-3,-1,1,3,5
constexpr int kPatchRadius = 5;
constexpr int kPatchStep = 2;

// Reference side of the patch: depends on the pixel only, reused by every hypothesis and view.
struct RefPatch {
    float v[kPatchN * kPatchN];  // v[i*6+j]: x offset index i (outer loop), y offset index j
    float mean;                  // sum_ref / 36
    float var;                   // sum_ref_ref/36 - mean^2
    __device__ __forceinline__ float at(int i, int j) const { return v[i * kPatchN + j]; }
    static constexpr bool kRuntimeIndex = false;  // a register array: indices must be compile-time constants
};

// Same interface with the 36 texels left in the workgroup's LDS tile (clamp-to-edge already applied when the
// tile was staged): frees 36 VGPRs per lane for one ds_read per sample.
template <int kPitch>
struct RefPatchLds {
    const float *base;  // &tile[(ly)*kPitch + lx], i.e. the texel at offset (-radius, -radius)
    float mean, var;
    __device__ __forceinline__ float at(int i, int j) const { return base[(kPatchStep * j) * kPitch + kPatchStep * i]; }
    static constexpr bool kRuntimeIndex = true;   // LDS: any index
};

__device__ __forceinline__ void ref_patch_finish(RefPatch &rp)
{
    float sum_r = 0.0f, sum_rr = 0.0f;
#pragma unroll
    for (int i = 0; i < kPatchN; ++i) {
        float row_r = 0.0f, row_rr = 0.0f;
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            const float r = rp.v[i * kPatchN + j];
            row_r += r;
            row_rr = fmaf(r, r, row_rr);
        }
        sum_r += row_r;
        sum_rr += row_rr;
    }
    const float inv_w = 1.0f / 36.0f;
    sum_r *= inv_w;
    sum_rr *= inv_w;
    rp.mean = sum_r;
    rp.var = fmaf(-sum_r, sum_r, sum_rr);
}

__device__ __forceinline__ void ref_patch_from_global(RefPatch &rp, const float *__restrict__ img, int W, int H, int px, int py)
{
#pragma unroll
    for (int i = 0; i < kPatchN; ++i) {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            rp.v[i * kPatchN + j] = fetch_texel(img, W, H, px + kPatchStep * i - kPatchRadius, py + kPatchStep * j - kPatchRadius);
        }
    }
    ref_patch_finish(rp);
}

// A dependent VALU chain issues one instruction per ~12 cycles per wave on gfx950 and independent ones one per
// ~5 (tools/valu_rates.hip), and only 2-3 waves fit a SIMD here, so the six samples of a patch row are
// computed in lock step: every stage below is six independent instructions, and the scheduler is not
// allowed to re-serialise the chains to save registers.
#define APD_STAGE() __builtin_amdgcn_sched_barrier(0)

// byte offset of a 16-byte float quad entry (qx, qy): qy*pitch + (pitch + 16) + 16*qx with pitch = (W+1)*16 bytes
__device__ __forceinline__ unsigned fquad_byte_offset(int qx, int qy, int pitch, int origin)
{
    int row, off;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row) : "v"(qy), "v"(pitch), "v"(origin));
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(off) : "v"(qx), "v"(row));
    return (unsigned)off;
}

// Sample positions of one patch row (fixed x, six y) -> bilinear weights + texel-quad gathers in flight.
// Reciprocal of the sample loops: kRecipIeee -- IEEE division (any denominator); kRecipExact -- v_rcp_f32 + one Newton step,
// the correctly rounded reciprocal on the range denominators_fast() checks (the default path); kRecipApprox -- the bare
// v_rcp_f32 (<= 1 ulp), what the reference's own --use_fast_math build does (CMakeLists.txt:20): the optional tolerance
// mode APD_OPT_FAST_RCP, NOT bit-identical to the oracle (tests/test_gpu_fast_rcp.py states what it keeps).
enum { kRecipIeee = 0, kRecipExact = 1, kRecipApprox = 2 };

template <int kRecip, bool kTiled = false>
__device__ __forceinline__ void quad_row_issue(const Homography &H, float bx, float by, float bz, const float (&yf)[kPatchN],
                                               global_quad_ptr srcq, unsigned pitch, int wm1, int hm1,
                                               float (&a)[kPatchN], float (&b)[kPatchN], quad_t (&t)[kPatchN])
{
    float z[kPatchN], X[kPatchN], Y[kPatchN], r[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        z[j] = fmaf(H.h[7], yf[j], bz);
        X[j] = fmaf(H.h[1], yf[j], bx);
        Y[j] = fmaf(H.h[4], yf[j], by);
    }
    APD_STAGE();
    if (kRecip == kRecipApprox) {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = __builtin_amdgcn_rcpf(z[j]);
        }
    } else if (kRecip == kRecipExact) {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = __builtin_amdgcn_rcpf(z[j]);
        }
        APD_STAGE();
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            z[j] = fmaf(-z[j], r[j], 1.0f);
        }
        APD_STAGE();
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = fmaf(z[j], r[j], r[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = 1.0f / z[j];
        }
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        X[j] *= r[j];
        Y[j] *= r[j];
    }
    APD_STAGE();
    int qx[kPatchN], qy[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        a[j] = __builtin_amdgcn_fractf(X[j]);
        b[j] = __builtin_amdgcn_fractf(Y[j]);
        qx[j] = cvt_floor_i32(X[j]);
        qy[j] = cvt_floor_i32(Y[j]);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        qx[j] = med3_i32(qx[j], -1, wm1);
        qy[j] = med3_i32(qy[j], -1, hm1);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        if constexpr (kTiled) {
            qx[j] = (int)quad_tiled_byte_offset(qx[j], qy[j], pitch);  // `pitch` = tiles per tile row here
        } else {
            qx[j] = (int)quad_byte_offset(qx[j], qy[j], (int)pitch, (int)(pitch + kRowEntryBytes));
        }
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        t[j] = quad_fetch(srcq, (unsigned)qx[j]);
    }
}

// Gathered quads + weights of one row -> six bilinear values (same three fmaf per sample as quad_lerp).
__device__ __forceinline__ void quad_row_lerp(const quad_t (&t)[kPatchN], const float (&a)[kPatchN], const float (&b)[kPatchN],
                                              float (&v)[kPatchN])
{
    float t00[kPatchN], t01[kPatchN];
    // byte k -> float with v_cvt_f32_ubyte<k> (4 cycles), the two differences as binary32 subtractions (2 cycles): left to
    // itself the compiler subtracts the bytes as integers (SDWA) and converts the difference, two 4-cycle instructions each
    float d0[kPatchN], d1[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t00[j]) : "v"(t[j]));
        asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(d1[j]) : "v"(t[j]));
        if (kPair2) {
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t01[j]) : "v"(t[j]));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(d0[j]) : "v"(t[j]));
        } else {
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(d0[j]) : "v"(t[j]));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(t01[j]) : "v"(t[j]));
        }
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        d0[j] -= t00[j];
        d1[j] -= t01[j];
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        t00[j] = fmaf(a[j], d0[j], t00[j]);
        t01[j] = fmaf(a[j], d1[j], t01[j]);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        t01[j] -= t00[j];
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        v[j] = fmaf(b[j], t01[j], t00[j]);
    }
}

// The same two stages for float texel-quad images (float grey values).
template <int kRecip>
__device__ __forceinline__ void fquad_row_issue(const Homography &H, float bx, float by, float bz, const float (&yf)[kPatchN],
                                                global_fquad_ptr fq, unsigned pitch, int wm1, int hm1, float (&a)[kPatchN],
                                                float (&b)[kPatchN], fquad_t (&t)[kPatchN])
{
    float z[kPatchN], X[kPatchN], Y[kPatchN], r[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        z[j] = fmaf(H.h[7], yf[j], bz);
        X[j] = fmaf(H.h[1], yf[j], bx);
        Y[j] = fmaf(H.h[4], yf[j], by);
    }
    APD_STAGE();
    if (kRecip == kRecipApprox) {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = __builtin_amdgcn_rcpf(z[j]);
        }
    } else if (kRecip == kRecipExact) {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = __builtin_amdgcn_rcpf(z[j]);
        }
        APD_STAGE();
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            z[j] = fmaf(-z[j], r[j], 1.0f);
        }
        APD_STAGE();
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = fmaf(z[j], r[j], r[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            r[j] = 1.0f / z[j];
        }
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        X[j] *= r[j];
        Y[j] *= r[j];
    }
    APD_STAGE();
    int qx[kPatchN], qy[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        a[j] = __builtin_amdgcn_fractf(X[j]);
        b[j] = __builtin_amdgcn_fractf(Y[j]);
        qx[j] = cvt_floor_i32(X[j]);
        qy[j] = cvt_floor_i32(Y[j]);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        qx[j] = med3_i32(qx[j], -1, wm1);
        qy[j] = med3_i32(qy[j], -1, hm1);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        qx[j] = (int)fquad_byte_offset(qx[j], qy[j], (int)pitch, (int)(pitch + 16u));
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        t[j] = fquad_fetch(fq, (unsigned)qx[j]);
    }
}

__device__ __forceinline__ void fquad_row_lerp(const fquad_t (&t)[kPatchN], const float (&a)[kPatchN], const float (&b)[kPatchN],
                                               float (&v)[kPatchN])
{
    float top[kPatchN], bot[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        top[j] = fmaf(a[j], t[j].y, t[j].x);
        bot[j] = fmaf(a[j], t[j].w, t[j].z);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        bot[j] -= top[j];
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        v[j] = fmaf(b[j], bot[j], top[j]);
    }
}

// The 36 warped source samples of one fixed patch and their three moments (APD.cu:561-583), summed in the
// reference's order (row partial sums, then total).  kRecip (see quad_row_issue); kRecipExact: every denominator is known to be in the
// range where recip_fast is the correctly rounded reciprocal.
template <bool kQuad, int kRecip, bool kTiled, typename Ref>
__device__ __forceinline__ void ncc_fixed_moments(const FrameArgs &fa, const ViewConst &vc, const Ref &rp, const Homography &H_in,
                                                  int px_in, int py_in, float &sum_s, float &sum_ss, float &sum_rs)
{
    const Homography &H = H_in;
    const int px = px_in, py = py_in;
    const global_quad_ptr srcq = (global_quad_ptr)(kTiled ? vc.quad_tiled : vc.quad);
    const int W = fa.W, Hh = fa.H;
    const unsigned qpitch = kTiled ? quad_tiles_x(W) : quad_row_pitch_bytes(W);
    const unsigned fpitch = 16u * (unsigned)(W + 1);
    const global_fquad_ptr srcf = (global_fquad_ptr)vc.fquad;
    const int wm1 = W - 1, hm1 = Hh - 1;
    float yf[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        yf[j] = (float)(py + kPatchStep * j - kPatchRadius);
    }
    sum_s = 0.0f;
    sum_ss = 0.0f;
    sum_rs = 0.0f;
    // software pipeline over the six rows: the gathers of the next APD_ROW_PREFETCH rows are in flight while row i
    // is reduced
    constexpr int kDepth = APD_ROW_PREFETCH, kBuf = kDepth + 1;
    float a[kBuf][kPatchN], b[kBuf][kPatchN];
    quad_t t[kQuad ? kBuf : 1][kPatchN];
    fquad_t tf[kQuad ? 1 : kBuf][kPatchN];
#pragma unroll
    for (int r = 0; r < kDepth; ++r) {
        const float xf = (float)(px + kPatchStep * r - kPatchRadius);
        const float bx = fmaf(H.h[0], xf, H.h[2]), by = fmaf(H.h[3], xf, H.h[5]), bz = fmaf(H.h[6], xf, H.h[8]);
        if constexpr (kQuad) {
            quad_row_issue<kRecip, kTiled>(H, bx, by, bz, yf, srcq, qpitch, wm1, hm1, a[r], b[r], t[r]);
        } else {
            fquad_row_issue<kRecip>(H, bx, by, bz, yf, srcf, fpitch, wm1, hm1, a[r], b[r], tf[r]);
        }
    }
#pragma unroll
    for (int i = 0; i < kPatchN; ++i) {
        float v[kPatchN];
        if (i + kDepth < kPatchN) {
            const int n = (i + kDepth) % kBuf;
            const float xf = (float)(px + kPatchStep * (i + kDepth) - kPatchRadius);
            const float bx = fmaf(H.h[0], xf, H.h[2]), by = fmaf(H.h[3], xf, H.h[5]), bz = fmaf(H.h[6], xf, H.h[8]);
            if constexpr (kQuad) {
                quad_row_issue<kRecip, kTiled>(H, bx, by, bz, yf, srcq, qpitch, wm1, hm1, a[n], b[n], t[n]);
            } else {
                fquad_row_issue<kRecip>(H, bx, by, bz, yf, srcf, fpitch, wm1, hm1, a[n], b[n], tf[n]);
            }
        }
        APD_STAGE();
        if constexpr (kQuad) {
            quad_row_lerp(t[i % kBuf], a[i % kBuf], b[i % kBuf], v);
        } else {
            fquad_row_lerp(tf[i % kBuf], a[i % kBuf], b[i % kBuf], v);
        }
        float ref[kPatchN];
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            ref[j] = rp.at(i, j);
        }
        float row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f;
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            row_s += v[j];
            row_ss = fmaf(v[j], v[j], row_ss);
            row_rs = fmaf(ref[j], v[j], row_rs);
        }
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
    }
}

// The kRecipIeee body again, one sample at a time in two rolled loops: the same instructions on the same operands in the same
// per-lane order (row partial sums, then the total), hence the same bits, for a fraction of the registers and of the code.  The
// lock-step body keeps six IEEE division sequences in flight -- the register peak of every kernel that inlines it, which the
// allocator pays for with spills around the (rare) path; this one is slower per call and is only taken when a denominator of the
// patch leaves the range of the exact fast reciprocal.  Needs a reference patch that can be indexed at run time (LDS).
template <bool kQuad, bool kTiled, typename Ref>
__device__ __forceinline__ void ncc_fixed_moments_ieee_rolled(const FrameArgs &fa, const ViewConst &vc, const Ref &rp, const Homography &H,
                                                              int px, int py, float &sum_s, float &sum_ss, float &sum_rs)
{
    const global_quad_ptr srcq = (global_quad_ptr)(kTiled ? vc.quad_tiled : vc.quad);
    const int W = fa.W, Hh = fa.H;
    const unsigned qpitch = kTiled ? quad_tiles_x(W) : quad_row_pitch_bytes(W);
    const unsigned fpitch = 16u * (unsigned)(W + 1);
    const global_fquad_ptr srcf = (global_fquad_ptr)vc.fquad;
    const int wm1 = W - 1, hm1 = Hh - 1;
    sum_s = 0.0f;
    sum_ss = 0.0f;
    sum_rs = 0.0f;
#pragma unroll 1
    for (int i = 0; i < kPatchN; ++i) {
        const float xf = (float)(px + kPatchStep * i - kPatchRadius);
        const float bx = fmaf(H.h[0], xf, H.h[2]), by = fmaf(H.h[3], xf, H.h[5]), bz = fmaf(H.h[6], xf, H.h[8]);
        float row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f;
#pragma unroll 1
        for (int j = 0; j < kPatchN; ++j) {
            const float yf = (float)(py + kPatchStep * j - kPatchRadius);
            const float z = fmaf(H.h[7], yf, bz);
            const float r = 1.0f / z;
            const float X = fmaf(H.h[1], yf, bx) * r, Y = fmaf(H.h[4], yf, by) * r;
            const float a = __builtin_amdgcn_fractf(X), b = __builtin_amdgcn_fractf(Y);
            const int qx = med3_i32(cvt_floor_i32(X), -1, wm1), qy = med3_i32(cvt_floor_i32(Y), -1, hm1);
            float v;
            if constexpr (kQuad) {
                const unsigned off = kTiled ? quad_tiled_byte_offset(qx, qy, qpitch) : quad_byte_offset(qx, qy, (int)qpitch, (int)(qpitch + kRowEntryBytes));
                v = quad_lerp(quad_fetch(srcq, off), a, b);
            } else {
                const fquad_t t = fquad_fetch(srcf, fquad_byte_offset(qx, qy, (int)fpitch, (int)(fpitch + 16u)));
                const float top = fmaf(a, t.y, t.x), bot = fmaf(a, t.w, t.z);
                v = fmaf(b, bot - top, top);
            }
            const float ref = rp.at(i, j);
            row_s += v;
            row_ss = fmaf(v, v, row_ss);
            row_rs = fmaf(ref, v, row_rs);
        }
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
    }
}

// The IEEE-division body a kernel should inline: rolled where the reference patch allows it.
template <bool kQuad, bool kTiled, typename Ref>
__device__ __forceinline__ void ncc_fixed_moments_ieee(const FrameArgs &fa, const ViewConst &vc, const Ref &rp, const Homography &H, int px,
                                                       int py, float &sum_s, float &sum_ss, float &sum_rs)
{
    if constexpr (APD_IEEE_COMPACT != 0 && Ref::kRuntimeIndex) {
        ncc_fixed_moments_ieee_rolled<kQuad, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    } else {
        ncc_fixed_moments<kQuad, kRecipIeee, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    }
}

// True when h6*x + h7*y + h8 is, for every (x, y) of the grid [x0, x1] x [y0, y1], in the range where
// recip_fast is the correctly rounded reciprocal.  The denominator is evaluated with monotone (rounded)
// fma, so over the grid it stays between its values at the four corners: if those share a sign and lie
// in the fast range, every sample does.
__device__ __forceinline__ bool denominators_fast(const Homography &H, float x0, float x1, float y0, float y1)
{
    const float b0 = fmaf(H.h[6], x0, H.h[8]), b1 = fmaf(H.h[6], x1, H.h[8]);
    const float z00 = fmaf(H.h[7], y0, b0), z01 = fmaf(H.h[7], y1, b0);
    const float z10 = fmaf(H.h[7], y0, b1), z11 = fmaf(H.h[7], y1, b1);
    const float lo = fminf(fminf(z00, z01), fminf(z10, z11)), hi = fmaxf(fmaxf(z00, z01), fmaxf(z10, z11));
    return (lo >= 0x1p-100f && hi <= 0x1p100f) || (hi <= -0x1p-100f && lo >= -0x1p100f);
}

// The fixed-patch cost for an already projected centre (the caller has done the bounds test of APD.cu:546).
template <bool kQuad, typename Ref, bool kTiled = false>
__device__ __forceinline__ float ncc_fixed_from_h(const FrameArgs &fa, const ViewConst &vc, const Ref &rp, const Homography &H,
                                                  int px, int py)
{
    const float kMinVar = 1e-5f;
    if (rp.var < kMinVar) {
        return 2.0f;  // the reference tests this after sampling; the result is the same
    }
    bool fast_recip = denominators_fast(H, (float)(px - kPatchRadius), (float)(px + kPatchRadius), (float)(py - kPatchRadius),
                                        (float)(py + kPatchRadius));
    // One 36-sample body per wave and NCC.  The IEEE division gives the bits of the fast reciprocal wherever that one is
    // valid, so when a single lane of the wave needs it, every lane takes it (random normals produce such lanes: a wave
    // with one of them used to run both bodies in turn).
    fast_recip = __builtin_amdgcn_ballot_w64(!fast_recip) == 0;
    float sum_s, sum_ss, sum_rs;
    if (__builtin_expect(fast_recip, 1)) {
        ncc_fixed_moments<kQuad, kRecipExact, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    } else {
        ncc_fixed_moments_ieee<kQuad, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    }
    const float inv_w = 1.0f / 36.0f;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    if (var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-rp.mean, sum_s, sum_rs);
    return ncc_cost_from_moments(rp.var, var_s, covar);
}

// ComputeBilateralNCCOld for plane q = n/d against source view vc.  kQuad selects the texel-quad image.
template <bool kQuad, typename Ref, bool kTiled = false>
__device__ __forceinline__ float ncc_fixed(const FrameArgs &fa, const ViewConst &vc, const Ref &rp, int px, int py,
                                           float qx, float qy, float qz)
{
    const Homography H = make_homography(fa, vc, qx, qy, qz);
    float cx, cy;
    correspond(H, (float)px, (float)py, cx, cy);
    if (cx >= vc.wf || cx < 0.0f || cy >= vc.hf || cy < 0.0f) {
        return 2.0f;
    }
    return ncc_fixed_from_h<kQuad, Ref, kTiled>(fa, vc, rp, H, px, py);
}

// ------------------------------------------------------------------------------------------------
// 3x3 sub-patch (weak_radius 5, weak_increment 5) around a reliable neighbour: the k >= 1 terms of
// ComputeBilateralNCCNew (APD.cu:461-505) on texel-quad images.  The reference side (nine texels and
// their moments) depends on the neighbour only and is prepared once per pixel by the caller.
// ------------------------------------------------------------------------------------------------

constexpr int kSubN = 3;       // offsets -5, 0, 5
constexpr int kSubStep = 5;

// Nine warped samples in lock step (same stages as quad_row_issue), reduced in the reference's order.
// ref_rows[i] packs the three reference texels of x offset i (y offset j in byte j).
// kRecip: kRecipExact (every denominator in the fast range) or kRecipIeee (any denominator; same bits where both are valid).
// Two halves, so that a caller can have the nine gathers of the next sub-patch in flight while it reduces this one
// (K9/K10, two waves per SIMD: latency is what is left once the traffic is halved).
template <int kRecip = kRecipExact>
__device__ __forceinline__ void subpatch_issue_quad(const Homography &H, global_quad_ptr srcq, unsigned qpitch, int wm1, int hm1, int cx,
                                                    int cy, float (&a)[kSubN * kSubN], float (&b)[kSubN * kSubN],
                                                    quad_t (&t)[kSubN * kSubN])
{
    constexpr int N = kSubN * kSubN;
    float z[N], X[N], Y[N], r[N];
#pragma unroll
    for (int i = 0; i < kSubN; ++i) {
        const float xf = (float)(cx + kSubStep * (i - 1));
        const float bx = fmaf(H.h[0], xf, H.h[2]);
        const float by = fmaf(H.h[3], xf, H.h[5]);
        const float bz = fmaf(H.h[6], xf, H.h[8]);
#pragma unroll
        for (int j = 0; j < kSubN; ++j) {
            const float yf = (float)(cy + kSubStep * (j - 1));
            z[i * kSubN + j] = fmaf(H.h[7], yf, bz);
            X[i * kSubN + j] = fmaf(H.h[1], yf, bx);
            Y[i * kSubN + j] = fmaf(H.h[4], yf, by);
        }
    }
    APD_STAGE();
    if constexpr (kRecip == kRecipExact) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            r[k] = __builtin_amdgcn_rcpf(z[k]);
        }
        APD_STAGE();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            z[k] = fmaf(-z[k], r[k], 1.0f);
        }
        APD_STAGE();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            r[k] = fmaf(z[k], r[k], r[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            r[k] = 1.0f / z[k];
        }
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        X[k] *= r[k];
        Y[k] *= r[k];
    }
    APD_STAGE();
    int qx[N], qy[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        a[k] = __builtin_amdgcn_fractf(X[k]);
        b[k] = __builtin_amdgcn_fractf(Y[k]);
        qx[k] = cvt_floor_i32(X[k]);
        qy[k] = cvt_floor_i32(Y[k]);
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        qx[k] = med3_i32(qx[k], -1, wm1);
        qy[k] = med3_i32(qy[k], -1, hm1);
    }
    APD_LAB_SUBPATCH_ROWS(qx, qy);
    // (Round 4 measured taking taps 0 and 1 of a row with ONE 16-byte load where they share a row segment -- 98.9 % of the rows do:
    // L1 tag accesses per launch -24 %, launch time 49.2 -> 56.7 ms; the tag count was a proxy, a 16-byte gather keeps the address /
    // data path of the L1 busy four times as long as a dword.  profiles/r04/ab_k910_row_segments.txt.  Nine dword gathers it is.)
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
#if APD_K910_SUBPATCH_TILED   // lab build: `srcq` is the tiled copy and `qpitch` its tiles per tile row (profiles/r05/ab_k910_tiled.txt)
        qx[k] = (int)quad_tiled_byte_offset(qx[k], qy[k], qpitch);
#else
        qx[k] = (int)quad_byte_offset(qx[k], qy[k], (int)qpitch, (int)(qpitch + kRowEntryBytes));
#endif
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        t[k] = quad_fetch(srcq, (unsigned)qx[k]);
    }
}

__device__ __forceinline__ float subpatch_finish_quad(const quad_t (&t)[kSubN * kSubN], const float (&a)[kSubN * kSubN],
                                                      const float (&b)[kSubN * kSubN], const uint32_t (&ref_rows)[kSubN], float mean_r,
                                                      float var_r)
{
    constexpr int N = kSubN * kSubN;
    float v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        v[k] = quad_lerp(t[k], a[k], b[k]);
    }
    float sum_s = 0.0f, sum_ss = 0.0f, sum_rs = 0.0f;
#pragma unroll
    for (int i = 0; i < kSubN; ++i) {
        float row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f;
#pragma unroll
        for (int j = 0; j < kSubN; ++j) {
            const float val = v[i * kSubN + j];
            const float ref = (float)((ref_rows[i] >> (8 * j)) & 0xFFu);
            row_s += val;
            row_ss = fmaf(val, val, row_ss);
            row_rs = fmaf(ref, val, row_rs);
        }
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
    }
    const float inv_w = 1.0f / 9.0f;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    const float kMinVar = 1e-5f;
    if (var_r < kMinVar || var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-mean_r, sum_s, sum_rs);
    return ncc_cost_from_moments(var_r, var_s, covar);
}

template <int kRecip = kRecipExact>
__device__ __forceinline__ float subpatch_cost_quad(const Homography &H, global_quad_ptr srcq, unsigned qpitch, int wm1, int hm1,
                                                    int cx, int cy, const uint32_t (&ref_rows)[kSubN], float mean_r, float var_r)
{
    float a[kSubN * kSubN], b[kSubN * kSubN];
    quad_t t[kSubN * kSubN];
    subpatch_issue_quad<kRecip>(H, srcq, qpitch, wm1, hm1, cx, cy, a, b, t);
    APD_STAGE();
    return subpatch_finish_quad(t, a, b, ref_rows, mean_r, var_r);
}

// The same sub-patch on a float texel-quad image (float grey values); ref[i * 3 + j] is read with stride `ref_stride` floats.
template <int kRecip = kRecipExact>
__device__ __forceinline__ float subpatch_cost_fquad(const Homography &H, global_fquad_ptr fq, unsigned fpitch, int wm1, int hm1,
                                                     int cx, int cy, const float *ref, int ref_stride, float mean_r, float var_r)
{
    constexpr int N = kSubN * kSubN;
    float z[N], X[N], Y[N], r[N];
#pragma unroll
    for (int i = 0; i < kSubN; ++i) {
        const float xf = (float)(cx + kSubStep * (i - 1));
        const float bx = fmaf(H.h[0], xf, H.h[2]);
        const float by = fmaf(H.h[3], xf, H.h[5]);
        const float bz = fmaf(H.h[6], xf, H.h[8]);
#pragma unroll
        for (int j = 0; j < kSubN; ++j) {
            const float yf = (float)(cy + kSubStep * (j - 1));
            z[i * kSubN + j] = fmaf(H.h[7], yf, bz);
            X[i * kSubN + j] = fmaf(H.h[1], yf, bx);
            Y[i * kSubN + j] = fmaf(H.h[4], yf, by);
        }
    }
    APD_STAGE();
    if constexpr (kRecip == kRecipExact) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            r[k] = __builtin_amdgcn_rcpf(z[k]);
        }
        APD_STAGE();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            z[k] = fmaf(-z[k], r[k], 1.0f);
        }
        APD_STAGE();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            r[k] = fmaf(z[k], r[k], r[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            r[k] = 1.0f / z[k];
        }
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        X[k] *= r[k];
        Y[k] *= r[k];
    }
    APD_STAGE();
    float a[N], b[N];
    int qx[N], qy[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        a[k] = __builtin_amdgcn_fractf(X[k]);
        b[k] = __builtin_amdgcn_fractf(Y[k]);
        qx[k] = cvt_floor_i32(X[k]);
        qy[k] = cvt_floor_i32(Y[k]);
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        qx[k] = med3_i32(qx[k], -1, wm1);
        qy[k] = med3_i32(qy[k], -1, hm1);
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        qx[k] = (int)fquad_byte_offset(qx[k], qy[k], (int)fpitch, (int)(fpitch + 16u));
    }
    APD_STAGE();
    fquad_t t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        t[k] = fquad_fetch(fq, (unsigned)qx[k]);
    }
    APD_STAGE();
    float v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float top = fmaf(a[k], t[k].y, t[k].x);
        const float bot = fmaf(a[k], t[k].w, t[k].z);
        v[k] = fmaf(b[k], bot - top, top);
    }
    float sum_s = 0.0f, sum_ss = 0.0f, sum_rs = 0.0f;
#pragma unroll
    for (int i = 0; i < kSubN; ++i) {
        float row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f;
#pragma unroll
        for (int j = 0; j < kSubN; ++j) {
            const float val = v[i * kSubN + j];
            const float rf = ref[(i * kSubN + j) * ref_stride];
            row_s += val;
            row_ss = fmaf(val, val, row_ss);
            row_rs = fmaf(rf, val, row_rs);
        }
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
    }
    const float inv_w = 1.0f / 9.0f;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    const float kMinVar = 1e-5f;
    if (var_r < kMinVar || var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-mean_r, sum_s, sum_rs);
    return ncc_cost_from_moments(var_r, var_s, covar);
}

// Generic patch (any centre / radius / increment): sub-patches of ComputeBilateralNCCNew
// (APD.cu:461-505).  Not hot: only WEAK pixels take this path.
__device__ __forceinline__ float patch_cost_generic(const FrameArgs &fa, const ViewConst &vc, const Homography &H, int cx, int cy,
                                                    int radius, int increment)
{
    const float *__restrict__ ref = fa.ref_img;
    const float *__restrict__ src = vc.img;
    const int W = fa.W, Hh = fa.H;
    float sum_r = 0.0f, sum_rr = 0.0f, sum_s = 0.0f, sum_ss = 0.0f, sum_rs = 0.0f, wsum = 0.0f;
    for (int i = -radius; i <= radius; i += increment) {
        float row_r = 0.0f, row_rr = 0.0f, row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f, row_w = 0.0f;
        const float xf = (float)(cx + i);
        const float bx = fmaf(H.h[0], xf, H.h[2]);
        const float by = fmaf(H.h[3], xf, H.h[5]);
        const float bz = fmaf(H.h[6], xf, H.h[8]);
        for (int j = -radius; j <= radius; j += increment) {
            const float r = fetch_texel(ref, W, Hh, cx + i, cy + j);
            const float yf = (float)(cy + j);
            const float inv = recip_rn(fmaf(H.h[7], yf, bz));
            const float sx = fmaf(H.h[1], yf, bx) * inv;
            const float sy = fmaf(H.h[4], yf, by) * inv;
            const float v = sample_bilinear(src, W, Hh, sx, sy);
            row_r += r;
            row_rr = fmaf(r, r, row_rr);
            row_s += v;
            row_ss = fmaf(v, v, row_ss);
            row_rs = fmaf(r, v, row_rs);
            row_w += 1.0f;
        }
        sum_r += row_r;
        sum_rr += row_rr;
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
        wsum += row_w;
    }
    const float inv_w = 1.0f / wsum;
    sum_r *= inv_w;
    sum_rr *= inv_w;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_r = fmaf(-sum_r, sum_r, sum_rr);
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    const float kMinVar = 1e-5f;
    if (var_r < kMinVar || var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-sum_r, sum_s, sum_rs);
    return ncc_cost_from_moments(var_r, var_s, covar);
}

// ------------------------------------------------------------------------------------------------
// geometric consistency (APD.cu:718-789)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ void backproject_world(float x, float y, float depth, const float *K, const float *R, const float *c,
                                                  float &Px, float &Py, float &Pz)
{
    const float X = depth * (x - K[2]) / K[0];
    const float Y = depth * (y - K[5]) / K[4];
    const float Z = depth;
    const float tx = R[0] * X + R[3] * Y + R[6] * Z;
    const float ty = R[1] * X + R[4] * Y + R[7] * Z;
    const float tz = R[2] * X + R[5] * Y + R[8] * Z;
    Px = tx + c[0];
    Py = ty + c[1];
    Pz = tz + c[2];
}

__device__ __forceinline__ void project_camera(float Px, float Py, float Pz, const float *K, const float *R, const float *t,
                                               float &u, float &v, float &depth)
{
    const float tx = R[0] * Px + R[1] * Py + R[2] * Pz + t[0];
    const float ty = R[3] * Px + R[4] * Py + R[5] * Pz + t[1];
    const float tz = R[6] * Px + R[7] * Py + R[8] * Pz + t[2];
    depth = K[6] * tx + K[7] * ty + K[8] * tz;
    u = (K[0] * tx + K[1] * ty + K[2] * tz) / depth;
    v = (K[3] * tx + K[4] * ty + K[5] * tz) / depth;
}

__device__ __forceinline__ float geom_cost(const FrameArgs &fa, const ViewConst &vc, int px, int py, const float4 pl)
{
    const float max_cost = 3.0f;
    const float depth = depth_from_plane(fa, pl, px, py);
    float Fx, Fy, Fz;
    backproject_world((float)px, (float)py, depth, fa.K, fa.R, fa.c, Fx, Fy, Fz);
    float su, sv, sd;
    project_camera(Fx, Fy, Fz, vc.K, vc.R, vc.t, su, sv, sd);
    const int ix = (int)fminf(fmaxf(su, -1.0f), (float)fa.W);
    const int iy = (int)fminf(fmaxf(sv, -1.0f), (float)fa.H);
    const float src_depth = fetch_texel(vc.depth, fa.W, fa.H, ix, iy);
    if (src_depth == 0.0f) {
        return max_cost;
    }
    float Px, Py, Pz;
    backproject_world(su, sv, src_depth, vc.K, vc.R, vc.c, Px, Py, Pz);
    float bu, bv, bd;
    project_camera(Px, Py, Pz, fa.K, fa.R, fa.t, bu, bv, bd);
    const float dc = (float)px - bu;
    const float dr = (float)py - bv;
    return fminf(max_cost, sqrtf(dc * dc + dr * dr));
}

// ------------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------------

// XCD-aware remap of a linear workgroup id: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md),
// so give each XCD one contiguous band of tiles and its private L2 one band of every image.
__device__ __forceinline__ int xcd_band_tile(int b, int ntiles)
{
    const int xcd = b & 7, slot = b >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

}  // namespace apd
