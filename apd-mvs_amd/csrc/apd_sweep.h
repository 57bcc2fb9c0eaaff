// apd_sweep.h -- pieces shared by the strong and weak checkerboard kernels.
#pragma once

#include <float.h>

#include "apd_device.h"

namespace apd {

__device__ __forceinline__ void sort_ascending(float *d, int n)  // APD.cu:3-12
{
    for (int i = 1; i < n; ++i) {
        const float v = d[i];
        int j = i;
        while (j >= 1 && v < d[j - 1]) {
            d[j] = d[j - 1];
            --j;
        }
        d[j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// checkerboard tiling shared by K6/K7, K9/K10, K12/K13
// ------------------------------------------------------------------------------------------------

// A wave64 owns the 64 same-colour pixels of a (128 / APD_CB_ROWS) x APD_CB_ROWS footprint; four waves make the
// workgroup tile.  The footprint shape trades two things: a wide, flat footprint makes one wave-level gather touch
// fewer 128-B lines of the row-major source image (the L1 looks lines up one per cycle), a square one keeps the
// patch halo, i.e. the L1 working set, small.
constexpr int kWaveH = APD_CB_ROWS, kWaveLanesX = 64 / kWaveH, kWaveW = 2 * kWaveLanesX;
constexpr int kWavesX = (kWaveH == 8) ? 2 : 1, kWavesY = 4 / kWavesX;
constexpr int kTileW = kWaveW * kWavesX, kTileH = kWaveH * kWavesY, kHalo = kPatchRadius;  // 32x16 (rows 8, 4) or 64x8 (rows 2)
constexpr int kLdsW = kTileW + 2 * kHalo;
constexpr int kLdsH = kTileH + 2 * kHalo;
constexpr int kLdsPitch = kLdsW | 1;           // odd pitch spreads the column reads over banks

struct TilePixel {
    int tx0, ty0;  // tile origin
    int lx, ly;    // pixel inside the tile
    int px, py;
};

// colour 0 = black ((x + y) even), 1 = red; wave w covers sub-tile (w % kWavesX, w / kWavesX).
__device__ __forceinline__ TilePixel checkerboard_pixel(const FrameArgs &fa, int colour)
{
    const int tiles_x = (fa.W + kTileW - 1) / kTileW;
    const int tiles_y = (fa.H + kTileH - 1) / kTileH;
    const int tile = xcd_band_tile(blockIdx.x, tiles_x * tiles_y);
    TilePixel t;
    t.ty0 = (tile / tiles_x) * kTileH;
    t.tx0 = (tile - (tile / tiles_x) * tiles_x) * kTileW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    t.ly = (wave / kWavesX) * kWaveH + lane / kWaveLanesX;
    t.lx = (wave % kWavesX) * kWaveW + 2 * (lane % kWaveLanesX) + ((t.ly + colour) & 1);
    t.px = t.tx0 + t.lx;
    t.py = t.ty0 + t.ly;
    return t;
}

__device__ __forceinline__ bool checkerboard_active(const FrameArgs &fa, const TilePixel &t)
{
    // rows beyond half_rows are never visited by the reference's HALF launch (APD.cu:2402)
    return t.px < fa.W && t.py < fa.H && t.py < fa.half_rows;
}

// ------------------------------------------------------------------------------------------------
// view weights: counts 0..15 (15 draws, APD.cu:1249) kept as nibbles in registers; the reference's
// uchar[32] per pixel (view_weight_cuda) is still what is stored in HBM.
// ------------------------------------------------------------------------------------------------

template <int NMAX>
struct ViewWeights {
    static constexpr int kWords = (NMAX + 7) / 8;
    uint32_t w[kWords];

    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < kWords; ++i) {
            w[i] = 0;
        }
    }
    __device__ __forceinline__ uint32_t word(int v) const
    {
        uint32_t x = w[0];
#pragma unroll
        for (int i = 1; i < kWords; ++i) {
            x = ((v >> 3) == i) ? w[i] : x;
        }
        return x;
    }
    __device__ __forceinline__ uint32_t get(int v) const { return (word(v) >> ((v & 7) * 4)) & 15u; }
    __device__ __forceinline__ void inc(int v)
    {
        const uint32_t one = 1u << ((v & 7) * 4);
#pragma unroll
        for (int i = 0; i < kWords; ++i) {
            w[i] += ((v >> 3) == i) ? one : 0u;
        }
    }
    // view_weight_cuda layout: one byte per view, 32 bytes per pixel
    __device__ __forceinline__ void store(const FrameArgs &fa, int center) const
    {
        uint32_t b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t nib = (i / 2 < kWords) ? (w[i / 2 < kWords ? i / 2 : 0] >> ((i & 1) * 16)) & 0xFFFFu : 0u;
            b[i] = (nib & 15u) | (((nib >> 4) & 15u) << 8) | (((nib >> 8) & 15u) << 16) | (((nib >> 12) & 15u) << 24);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(fa.view_weight + (size_t)center * APD_MAX_IMAGES);
        dst[0] = make_uint4(b[0], b[1], b[2], b[3]);
        dst[1] = make_uint4(b[4], b[5], b[6], b[7]);
    }
    __device__ __forceinline__ void load(const FrameArgs &fa, int center)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(fa.view_weight + (size_t)center * APD_MAX_IMAGES);
        const uint4 a = src[0], c = src[1];
        const uint32_t b[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < kWords; ++i) {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t d = b[2 * i + k];
                const uint32_t nib = (d & 15u) | (((d >> 8) & 15u) << 4) | (((d >> 16) & 15u) << 8) | (((d >> 24) & 15u) << 12);
                x |= nib << (16 * k);
            }
            w[i] = x;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// view selection shared by strong (:1203-1271) and weak (:1365-1434) propagation
// ------------------------------------------------------------------------------------------------

template <int NMAX>
__device__ __forceinline__ void select_views(const FrameArgs &fa, int iter, const float (*cost_array)[NMAX], const float *priors,
                                             Rng &rng, ViewWeights<NMAX> &vw, uint32_t &sel_out, float &weight_norm_out)
{
    const int nsrc = fa.num_src;
    float probs[NMAX];
    const float thr = (float)(0.8 * (double)exp_poly((float)(iter * iter) / (-90.0f)));
    for (int i = 0; i < nsrc; ++i) {
        float count = 0;
        int count_false = 0;
        float tmpw = 0;
        for (int j = 0; j < 8; ++j) {
            const float c = cost_array[j][i];
            if (c < thr) {
                tmpw += exp_poly(c * c / (-0.18f));
                count++;
            }
            if (c > 1.2f) {
                count_false++;
            }
        }
        float p = 0.0f;
        if (count > 2 && count_false < 3) {
            p = tmpw / count;
        } else if (count_false < 3) {
            p = exp_poly(thr * thr / (-0.32f));
        }
        probs[i] = p * priors[i];
    }
    // TransformPDFToCDF, APD.cu:143-157
    float sum = 0.0f;
    for (int i = 0; i < nsrc; ++i) {
        sum += probs[i];
    }
    const float inv = 1.0f / sum;
    float acc = 0.0f;
    for (int i = 0; i < nsrc; ++i) {
        acc += probs[i] * inv;
        probs[i] = acc;
    }
    for (int sample = 0; sample < 15; ++sample) {
        const float rp = rng_uniform(rng) - FLT_EPSILON;
        for (int v = 0; v < nsrc; ++v) {
            if (probs[v] > rp) {
                vw.inc(v);
                break;
            }
        }
    }
    uint32_t sel = 0;
    float wn = 0;
    for (int i = 0; i < nsrc; ++i) {
        const uint32_t wi = vw.get(i);
        if (wi > 0) {
            sel |= 1u << i;
            wn += (float)wi;
        }
    }
    sel_out = sel;
    weight_norm_out = wn;
}

// Early-out bound of the refinement loops (K6/K7, K9/K10).  A refinement hypothesis is accepted iff
// fl(sum / weight_norm) < cost (APD.cu:884, :932, :974) with sum = the weighted costs added view by view.  Every term is
// >= 0, so the partial sums and their quotients never decrease; if a partial sum has reached a value t with
// t / weight_norm >= cost in real arithmetic, the final quotient is >= cost as well and the hypothesis is rejected
// whatever the remaining views cost.  Returns such a t: fl(cost * weight_norm) is within 2^-24 (relative) of the
// product, two roundings and the factor 1 + 2^-22 put the result strictly above it.  Tiny, zero-weight and non-finite
// inputs return +inf or NaN (never "lost": the loops then score every view, as the reference does).
__device__ __forceinline__ float refinement_lost_bound(const FrameArgs &fa, float cost, float weight_norm)
{
    const float p = cost * weight_norm;
    if (!APD_REFINE_EARLY_OUT || !fa.early_out || !(p >= 0x1p-100f) || !(weight_norm > 0.0f)) {
        return __builtin_inff();
    }
    return p * (1.0f + 0x1p-22f);
}

// The five refinement hypotheses of APD.cu:855-867 / :939-951 (RNG order: depth, normal, depth, 3 angles).
__device__ __forceinline__ void make_refinement_set(const FrameArgs &fa, int px, int py, Rng &rng, const float4 plane, float depth,
                                                    float *depths, float4 *normals)
{
    const float depth_perturbation = 0.02f, normal_perturbation = 0.02f;
    const float depth_rand = rng_uniform(rng) * (fa.depth_max - fa.depth_min) + fa.depth_min;
    const float4 n_rand = random_normal(fa, px, py, rng, depth);
    const float lo = (1 - depth_perturbation) * depth;
    const float hi = (1 + depth_perturbation) * depth;
    const float depth_pert = rng_uniform(rng) * (hi - lo) + lo;  // the reference's do-while never loops
    const float4 n_pert = perturbed_normal(fa, px, py, plane, rng, (float)((double)normal_perturbation * 3.14159265358979323846));
    depths[0] = depth_rand;
    depths[1] = depth;
    depths[2] = depth_rand;
    depths[3] = depth;
    depths[4] = depth_pert;
    normals[0] = plane;
    normals[1] = n_rand;
    normals[2] = n_rand;
    normals[3] = n_pert;
    normals[4] = plane;
}

}  // namespace apd
