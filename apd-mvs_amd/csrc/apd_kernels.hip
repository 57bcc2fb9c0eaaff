// apd_kernels.hip -- hand-written gfx950 kernels of the PatchMatch path (K1..K15 of SURVEY.md 2.1).
//
// Thread mapping: one lane per pixel.  Checkerboard kernels give a wave64 the 64 same-colour pixels
// of a 16x8 footprint (not the reference's 32x2 strip, APD.cu:1512-1519) and a 256-thread workgroup
// a 32x16 tile whose reference-image patch halo is staged in LDS.  Workgroups are dealt to XCDs in
// contiguous bands of tiles so every XCD's L2 holds one band of each image.
#include "apd_device.h"
#include "apd_sweep.h"

#include <float.h>
#include <stdlib.h>
#include <rocrand/rocrand_xorwow.h>

namespace apd {

// ------------------------------------------------------------------------------------------------
// K1  InitRandomStates (APD.cu:791-804): state(y, x) = xorwow(seed, subsequence = y, offset = x)
// ------------------------------------------------------------------------------------------------

// rocRAND keeps its state protected; this only exposes the six words after rocrand_init.
struct XorwowPeek : public rocrand_device::xorwow_engine {
    __device__ XorwowPeek(unsigned long long seed, unsigned long long subsequence, unsigned long long offset)
        : rocrand_device::xorwow_engine(seed, subsequence, offset)
    {
    }
    __device__ Rng words() const { return Rng{m_state.x[0], m_state.x[1], m_state.x[2], m_state.x[3], m_state.x[4], m_state.d}; }
};

constexpr int kRngSegment = 64;  // pixels stepped sequentially after one skip-ahead

// One lane jumps to (row y, column x0) with rocRAND's skip-ahead matrices, then walks kRngSegment
// columns with plain xorwow steps: identical to rocrand_init(seed, y, x) per pixel, ~64x cheaper.
__global__ __launch_bounds__(64) void k1_init_random_states(FrameArgs fa)
{
    const int segs_per_row = (fa.W + kRngSegment - 1) / kRngSegment;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= segs_per_row * fa.H) {
        return;
    }
    const int y = gid / segs_per_row;
    const int x0 = (gid - y * segs_per_row) * kRngSegment;
    XorwowPeek eng(fa.seed, (unsigned long long)y, (unsigned long long)x0);
    Rng r = eng.words();
    const int x1 = min(x0 + kRngSegment, fa.W);
    for (int x = x0; x < x1; ++x) {
        rng_store(fa.rng, y * fa.W + x, r);
        rng_next(r);
    }
}

// ------------------------------------------------------------------------------------------------
// K5  RandomInitialization (APD.cu:806-835) with the initial costs of :616-693
// ------------------------------------------------------------------------------------------------

// Full-frame kernels (K5, K14, K15): a wave64 covers a (64 / APD_FF_ROWS) x APD_FF_ROWS block of pixels, four waves a
// workgroup tile (same trade-off as APD_CB_ROWS in apd_sweep.h).
constexpr int kFfWaveH = APD_FF_ROWS, kFfWaveW = 64 / kFfWaveH;
constexpr int kFfWavesX = (kFfWaveH == 8) ? 2 : 1, kFfWavesY = 4 / kFfWavesX;
constexpr int kFullTileW = kFfWaveW * kFfWavesX, kFullTileH = kFfWaveH * kFfWavesY;  // 16x16 (rows 8, 4) or 32x8 (rows 2)
constexpr int kFullLdsW = kFullTileW + 2 * kPatchRadius, kFullLdsH = kFullTileH + 2 * kPatchRadius;
constexpr int kFullPitch = kFullLdsW | 1;

__device__ __forceinline__ void full_frame_pixel(int &px, int &py)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    px = blockIdx.x * kFullTileW + (wave % kFfWavesX) * kFfWaveW + lane % kFfWaveW;
    py = blockIdx.y * kFullTileH + (wave / kFfWavesX) * kFfWaveH + lane / kFfWaveW;
}

// Stages the workgroup's reference tile + 5 px halo (clamp-to-edge) and returns this lane's patch accessor; the 36
// texels stay in LDS, only their two moments live in registers.  Every thread of the block must call it.
__device__ __forceinline__ RefPatchLds<kFullPitch> stage_full_frame_ref(const FrameArgs &fa, float *tile, int px, int py)
{
    const int x0 = blockIdx.x * kFullTileW - kPatchRadius, y0 = blockIdx.y * kFullTileH - kPatchRadius;
    for (int idx = threadIdx.x; idx < kFullLdsW * kFullLdsH; idx += 256) {
        const int r = idx / kFullLdsW, c = idx - r * kFullLdsW;
        tile[r * kFullPitch + c] = fetch_texel(fa.ref_img, fa.W, fa.H, x0 + c, y0 + r);
    }
    __syncthreads();
    RefPatchLds<kFullPitch> rp;
    rp.base = &tile[(py - y0 - kPatchRadius) * kFullPitch + (px - x0 - kPatchRadius)];
    RefPatch tmp;
#pragma unroll
    for (int i = 0; i < kPatchN; ++i) {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            tmp.v[i * kPatchN + j] = rp.at(i, j);
        }
    }
    ref_patch_finish(tmp);
    rp.mean = tmp.mean;
    rp.var = tmp.var;
    return rp;
}

// kTiled: the NCCs of the random planes (FIRST_INIT) gather from the tiled copy of the quad image
template <bool kQuad, bool kTiled>
__global__ __launch_bounds__(256) void k5_random_initialization(FrameArgs fa)
{
    __shared__ float tile[kFullLdsH * kFullPitch];
    int px, py;
    full_frame_pixel(px, py);
    const RefPatchLds<kFullPitch> rp = stage_full_frame_ref(fa, tile, px, py);
    if (px >= fa.W || py >= fa.H) {
        return;
    }
    const int center = py * fa.W + px;
    if (fa.state == APD_FIRST_INIT) {
        Rng rng = rng_load(fa.rng, center);
        const float4 pl = random_plane(fa, px, py, rng);
        rng_store(fa.rng, center, rng);
        fa.planes[center] = pl;
        // ComputeMultiViewInitialCostandSelectedViews, :616-662
        float qx, qy, qz;
        plane_q(pl, qx, qy, qz);
        float sorted[32], orig[32];
        int valid = 0;
#pragma unroll 1
        for (int v = 0; v < fa.num_src; ++v) {
            const float c = ncc_fixed<kQuad, RefPatchLds<kFullPitch>, kTiled>(fa, view_const(fa, v), rp, px, py, qx, qy, qz);
            sorted[v] = c;
            orig[v] = c;
            if (c < 2.0f) {
                valid++;
            }
        }
        sort_ascending(sorted, fa.num_src);
        uint32_t sel = 0;
        float cost = 2.0f;
        const int top_k = min(valid, fa.top_k);
        if (top_k > 0) {
            float acc = 0.0f;
            for (int i = 0; i < top_k; ++i) {
                acc += sorted[i];
            }
            const float thr = sorted[top_k - 1];
            for (int i = 0; i < fa.num_src; ++i) {
                if (orig[i] <= thr) {
                    sel |= 1u << i;
                }
            }
            cost = acc / (float)top_k;
        }
        fa.selected_views[center] = sel;
        fa.costs[center] = cost;
    } else {
        // loaded (world normal, depth) -> (camera normal, distance), :827-832
        float4 pl = normal_world_to_cam(fa, fa.planes[center]);
        const float depth = pl.w;
        pl.w = distance_to_origin(fa, px, py, depth, pl.x, pl.y, pl.z);
        fa.planes[center] = pl;
        // ComputeMultiViewInitialCost, :664-693
        float qx, qy, qz;
        plane_q(pl, qx, qy, qz);
        uint32_t sel = fa.selected_views[center];
        int count = 0;
        float cost = 0.0f;
#pragma unroll 1
        for (int v = 0; v < fa.num_src; ++v) {
            if (bit_test(sel, (unsigned)v)) {
                const float c = ncc_fixed<kQuad>(fa, view_const(fa, v), rp, px, py, qx, qy, qz);
                if (c < 2.0f) {
                    count++;
                    cost += c;
                } else {
                    sel = bit_unset_quirk(sel, (unsigned)v);
                }
            }
        }
        fa.selected_views[center] = sel;
        fa.costs[center] = (count == 0) ? 2.0f : cost / (float)count;
    }
}

// ------------------------------------------------------------------------------------------------
// K6/K7  Black/RedPixelUpdateStrong -> CheckerboardPropagationStrong (APD.cu:982-1321, 837-890)
// ------------------------------------------------------------------------------------------------

// Cheapest candidate of propagation arm `arm` (order of APD.cu:1020: near/far x up,down,left,right).
__device__ __forceinline__ bool arm_candidate(const FrameArgs &fa, int px, int py, int arm, int &pos)
{
    const int d = arm >> 1;
    const int dx = (d == 2) ? -1 : (d == 3 ? 1 : 0);
    const int dy = (d == 0) ? -1 : (d == 1 ? 1 : 0);
    const float *__restrict__ costs = fa.costs;
    const int W = fa.W;
    if (arm & 1) {  // far: +-3, then ten more at stride 2 (:1021-1095)
        if (!inside(fa, px + 3 * dx, py + 3 * dy)) {
            return false;
        }
        int best = (px + 3 * dx) + (py + 3 * dy) * W;
        float cmin = costs[best];
        for (int i = 1; i < 11; ++i) {
            const int qx = px + (3 + 2 * i) * dx, qy = py + (3 + 2 * i) * dy;
            if (inside(fa, qx, qy)) {
                const int q = qx + qy * W;
                const float c = costs[q];
                if (c < cmin) {
                    cmin = c;
                    best = q;
                }
            }
        }
        pos = best;
        return true;
    }
    // near: +-1, then three V-shaped pairs, negative side first (:1097-1199)
    if (!inside(fa, px + dx, py + dy)) {
        return false;
    }
    const int ex = dy != 0 ? 1 : 0, ey = dx != 0 ? 1 : 0;
    int best = (px + dx) + (py + dy) * W;
    float cmin = costs[best];
    for (int i = 0; i < 3; ++i) {
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const int qx = px + (2 + i) * dx + sgn * (1 + i) * ex;
            const int qy = py + (2 + i) * dy + sgn * (1 + i) * ey;
            if (inside(fa, qx, qy)) {
                const int q = qx + qy * W;
                const float c = costs[q];
                if (c < cmin) {
                    cmin = c;
                    best = q;
                }
            }
        }
    }
    pos = best;
    return true;
}

// One launch = one colour.  Hypotheses 0..7 are the propagation arms, 8 the current plane,
// 9..13 the refinement set; a single loop keeps one inlined copy of the 36-sample NCC.
template <int NMAX, bool kQuad>
__global__ __launch_bounds__(256, APD_K67_WAVES) void k67_update_strong(FrameArgs fa, int colour, int iter)
{
    __shared__ float tile[kLdsH * kLdsPitch];
    const TilePixel t = checkerboard_pixel(fa, colour);
    // stage the reference tile + 5 px halo (clamp-to-edge, as the texture unit would)
    for (int idx = threadIdx.x; idx < kLdsW * kLdsH; idx += 256) {
        const int r = idx / kLdsW, c = idx - r * kLdsW;
        tile[r * kLdsPitch + c] = fetch_texel(fa.ref_img, fa.W, fa.H, t.tx0 + c - kHalo, t.ty0 + r - kHalo);
    }
    __syncthreads();
    if (!checkerboard_active(fa, t)) {
        return;
    }
    const int px = t.px, py = t.py;
    const int center = py * fa.W + px;
    if (fa.weak_info[center] == APD_WEAK) {
        return;
    }
    // the 36 reference texels stay in the LDS tile (one ds_read per sample); only their moments live in registers
    RefPatchLds<kLdsPitch> rp;
    rp.base = &tile[t.ly * kLdsPitch + t.lx];
    {
        RefPatch tmp;
#pragma unroll
        for (int i = 0; i < kPatchN; ++i) {
#pragma unroll
            for (int j = 0; j < kPatchN; ++j) {
                tmp.v[i * kPatchN + j] = rp.at(i, j);
            }
        }
        ref_patch_finish(tmp);
        rp.mean = tmp.mean;
        rp.var = tmp.var;
    }

    const int nsrc = fa.num_src;
    Rng rng = rng_load(fa.rng, center);
    float cost_array[9][NMAX];  // [8] = current plane
    for (int h = 0; h < 9; ++h) {
        for (int v = 0; v < NMAX; ++v) {
            cost_array[h][v] = 0.0f;
        }
    }
    cost_array[0][0] = 2.0f;  // "= { 2.0f }" sets only the first element (APD.cu:1004)
    int positions[8];
    unsigned flags = 0;
    ViewWeights<NMAX> vw;
    vw.clear();
    float weight_norm = 0.0f;
    uint32_t sel = 0;
    float4 plane_now = fa.planes[center];
    float depth_now = 0.0f, cost_now = 0.0f, cost_committed = 0.0f;
    float ref_depths[5];
    float4 ref_normals[5];

#pragma unroll 1
    for (int h = 0; h < 14; ++h) {
        if (h == 9) {
            // ---- joint view selection (:1203-1271) ----
            float priors[NMAX];
            for (int j = 0; j < NMAX; ++j) {
                priors[j] = 0.0f;
            }
            const int nb_pos[4] = {center - fa.W, center + fa.W, center - 1, center + 1};
            for (int i = 0; i < 4; ++i) {
                if (flags & (1u << (2 * i))) {
                    const uint32_t sv = fa.selected_views[nb_pos[i]];
                    for (int j = 0; j < nsrc; ++j) {
                        priors[j] += bit_test(sv, (unsigned)j) == 1 ? 0.9f : 0.1f;
                    }
                }
            }
            select_views<NMAX>(fa, iter, cost_array, priors, rng, vw, sel, weight_norm);
            vw.store(fa, center);
            float final_costs[8];
            for (int i = 0; i < 8; ++i) {
                float f = 0.0f;
                for (int j = 0; j < nsrc; ++j) {
                    const uint32_t wj = vw.get(j);
                    if (wj > 0) {
                        f += (float)wj * cost_array[i][j];
                    }
                }
                final_costs[i] = f / weight_norm;
            }
            int best = 0;  // FindMinCostIndex: "<=" -> last minimum wins (:29-40)
            float best_c = final_costs[0];
            for (int i = 1; i < 8; ++i) {
                if (final_costs[i] <= best_c) {
                    best_c = final_costs[i];
                    best = i;
                }
            }
            cost_now = 0.0f;
            for (int i = 0; i < nsrc; ++i) {
                cost_now += (float)vw.get(i) * cost_array[8][i];
            }
            cost_now /= weight_norm;
            cost_committed = cost_now;  // costs[center] = cost_now (:1295)
            depth_now = depth_from_plane(fa, plane_now, px, py);
            if (flags & (1u << best)) {
                const float4 cand = fa.planes[positions[best]];
                const float d = depth_from_plane(fa, cand, px, py);
                if (d >= fa.depth_min && d <= fa.depth_max && final_costs[best] < cost_now) {
                    depth_now = d;
                    plane_now = cand;
                    cost_now = final_costs[best];
                    fa.selected_views[center] = sel;
                }
            }
            make_refinement_set(fa, px, py, rng, plane_now, depth_now, ref_depths, ref_normals);
        }
        float4 pl;
        if (h < 8) {
            int pos;
            if (!arm_candidate(fa, px, py, h, pos)) {
                continue;
            }
            positions[h] = pos;
            flags |= 1u << h;
            pl = fa.planes[pos];
        } else if (h == 8) {
            pl = plane_now;
        } else {
            pl = ref_normals[h - 9];
            pl.w = distance_to_origin(fa, px, py, ref_depths[h - 9], pl.x, pl.y, pl.z);
        }
        float qx, qy, qz;
        plane_q(pl, qx, qy, qz);
        float tc = 0.0f;
#pragma unroll 1
        for (int v = 0; v < nsrc; ++v) {
            const uint32_t wv = vw.get(v);
            if (h >= 9 && wv == 0) {
                continue;  // a refinement hypothesis only ever uses the costs of the selected views (:876-880)
            }
            const float c = ncc_fixed<kQuad>(fa, view_const(fa, v), rp, px, py, qx, qy, qz);
            if (h < 9) {
                cost_array[h][v] = c;
            } else {
                tc += (float)wv * c;
            }
        }
        if (h >= 9) {  // PlaneHypothesisRefinementStrong accept test (:881-888)
            tc /= weight_norm;
            const float d = depth_from_plane(fa, pl, px, py);
            if (d >= fa.depth_min && d <= fa.depth_max && tc < cost_now) {
                depth_now = d;
                plane_now = pl;
                cost_now = tc;
            }
        }
    }
    rng_store(fa.rng, center, rng);
    if (fa.state == APD_REFINE_INIT) {  // :1311-1316, double comparison
        if ((double)cost_now < (double)cost_committed - 0.1) {
            fa.costs[center] = cost_now;
            fa.planes[center] = plane_now;
        } else {
            fa.costs[center] = cost_committed;
        }
    } else {
        fa.costs[center] = cost_now;
        fa.planes[center] = plane_now;
    }
}

// ------------------------------------------------------------------------------------------------
// K11  GetDepthandNormal (APD.cu:1587-1602)
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k11_depth_and_normal(FrameArgs fa)
{
    const int center = blockIdx.x * 256 + threadIdx.x;
    if (center >= fa.W * fa.H) {
        return;
    }
    const int py = center / fa.W, px = center - py * fa.W;
    float4 pl = fa.planes[center];
    pl.w = depth_from_plane(fa, pl, px, py);
    fa.planes[center] = normal_cam_to_world(fa, pl);
}

// ------------------------------------------------------------------------------------------------
// K12/K13  Black/RedPixelFilterStrong (APD.cu:1604-1748)
// ------------------------------------------------------------------------------------------------

// {dx, dy, extra top margin}: every reference condition is "tap inside the image" except the two
// (+-1,-2) taps, which additionally need p.y > 2 (:1691, :1695).
__constant__ int8_t k_filter_taps[20][3] = {
    {0, -1, 0}, {0, -3, 0}, {0, -5, 0},  {0, 1, 0},  {0, 3, 0},   {0, 5, 0},   {-1, 0, 0},
    {-3, 0, 0}, {-5, 0, 0}, {1, 0, 0},   {3, 0, 0},  {5, 0, 0},   {2, -1, 0},  {2, 1, 0},
    {-2, -1, 0}, {-2, 1, 0}, {-1, -2, 1}, {1, -2, 1}, {-1, 2, 0}, {1, 2, 0}};

__global__ __launch_bounds__(256) void k1213_filter_strong(FrameArgs fa, int colour)
{
    const TilePixel t = checkerboard_pixel(fa, colour);
    if (!checkerboard_active(fa, t)) {
        return;
    }
    const int px = t.px, py = t.py, W = fa.W;
    const int center = py * W + px;
    if (fa.weak_info[center] == APD_WEAK) {
        return;
    }
    if (fa.costs[center] < 0.001f) {
        return;
    }
    float f[21];
    int n = 0;
    f[n++] = fa.planes[center].w;
    for (int k = 0; k < 20; ++k) {
        const int qx = px + k_filter_taps[k][0], qy = py + k_filter_taps[k][1];
        if (inside(fa, qx, qy - k_filter_taps[k][2]) && fa.weak_info[qx + qy * W] == APD_STRONG) {
            f[n++] = fa.planes[qx + qy * W].w;
        }
    }
    sort_ascending(f, n);
    const int m = n / 2;
    fa.planes[center].w = (n % 2 == 0) ? (f[m - 1] + f[m]) / 2 : f[m];
}

// ------------------------------------------------------------------------------------------------
// K14 DepthToWeak (APD.cu:1990-2144) and K15 LocalRefine (:2146-2232)
// ------------------------------------------------------------------------------------------------

// Weighted cost of one depth sample along the pixel's ray over the selected views.
//   kLocalRefine == false: sum_sel (ncc + gf*geom) * w           (:2070-2080, :2031-2035)
//   kLocalRefine == true : sum_sel ncc*w (+ gf*geom*w)            (:2217-2220)
template <bool kLocalRefine, bool kQuad, typename Ref>
__device__ __forceinline__ float disparity_sample_cost(const FrameArgs &fa, const Ref &rp, int px, int py, const float4 origin,
                                                       float depth, uint32_t sel, const ViewWeights<32> &vw)
{
    float4 pl = origin;
    pl.w = distance_to_origin(fa, px, py, depth, pl.x, pl.y, pl.z);
    float qx, qy, qz;
    plane_q(pl, qx, qy, qz);
    float acc = 0.0f;
#pragma unroll 1
    for (int v = 0; v < fa.num_src; ++v) {
        if (bit_test(sel, (unsigned)v)) {
            const float c = ncc_fixed<kQuad>(fa, view_const(fa, v), rp, px, py, qx, qy, qz);
            if (kLocalRefine) {
                const float wv = (float)vw.get(v);
                acc += c * wv;
                if (fa.geom_consistency) {
                    acc += fa.geom_factor * geom_cost(fa, view_const(fa, v), px, py, pl) * wv;
                }
            } else {
                float tc = 0.0f;
                tc += c;
                if (fa.geom_consistency) {
                    tc += fa.geom_factor * geom_cost(fa, view_const(fa, v), px, py, pl);
                }
                acc += tc * (float)vw.get(v);
            }
        }
    }
    return acc;
}

// baseline + weight sum over the selected views (:2036-2044); no image access
__device__ __forceinline__ int baseline_and_weight(const FrameArgs &fa, uint32_t sel, const ViewWeights<32> &vw, float &base_line, float &weight_normal)
{
    float bl = 0, wn = 0.0f;
    int valid = 0;
    for (int v = 0; v < fa.num_src; ++v) {
        if (bit_test(sel, (unsigned)v)) {
            const ViewConst &vc = view_const(fa, v);
            wn += (float)vw.get(v);
            const float d0 = fa.c[0] - vc.c[0];
            const float d1 = fa.c[1] - vc.c[1];
            const float d2 = fa.c[2] - vc.c[2];
            const double tv = (double)(d0 * d0 + d1 * d1 + d2 * d2);
            bl += sqrtf((float)tv);
            valid++;
        }
    }
    base_line = bl;
    weight_normal = wn;
    return valid;
}

template <bool kQuad>
__global__ __launch_bounds__(256) void k14_depth_to_weak(FrameArgs fa)
{
    __shared__ float tile[kFullLdsH * kFullPitch];
    int px, py;
    full_frame_pixel(px, py);
    const RefPatchLds<kFullPitch> rp = stage_full_frame_ref(fa, tile, px, py);
    if (px >= fa.W || py >= fa.H) {
        return;
    }
    const int W = fa.W, H = fa.H;
    const int min_margin = 6;
    const int center = px + py * W;
    if (px < min_margin || py < min_margin || px >= W - min_margin || py >= H - min_margin) {
        fa.weak_info[center] = APD_UNKNOWN;
        return;
    }
    const float4 origin = normal_world_to_cam(fa, fa.planes[center]);
    const float origin_depth = origin.w;
    if (origin_depth == 0) {
        fa.weak_info[center] = APD_UNKNOWN;
        return;
    }
    const uint32_t sel = fa.selected_views[center];
    ViewWeights<32> vw;
    vw.load(fa, center);
    float base_line, weight_normal;
    const int valid = baseline_and_weight(fa, sel, vw, base_line, weight_normal);
    if (valid == 0) {
        fa.weak_info[center] = APD_UNKNOWN;
        return;
    }
    // cost_now of :2022-2051 is computed by the reference but never used by K14's classification
    base_line /= (float)valid;
    const float disp = fa.K[0] * base_line / origin_depth;
    constexpr int RADIUS = 30, NP = 2 * RADIUS + 1;
    float pc[NP];
#pragma unroll 1
    for (int pd = -RADIUS; pd <= RADIUS; ++pd) {
        const float p_depth = fa.K[0] * base_line / (disp + (float)pd);
        if (p_depth < fa.depth_min || p_depth > fa.depth_max) {
            pc[pd + RADIUS] = 2.0f;
            continue;
        }
        float p_cost = disparity_sample_cost<false, kQuad>(fa, rp, px, py, origin, p_depth, sel, vw);
        p_cost /= weight_normal;
        pc[pd + RADIUS] = (2.0f > p_cost) ? p_cost : 2.0f;  // MIN(2.0f, p_cost): NaN -> 2
    }
    uint64_t peaks = 0;
    int peak_count = 0, min_peak = 0;
    float min_cost = 2.0f;
    for (int i = 2; i < NP - 2; ++i) {
        if (pc[i - 1] > pc[i] && pc[i + 1] > pc[i]) {
            peaks |= 1ull << i;
            peak_count++;
            if (pc[i] < min_cost) {
                min_peak = i;
                min_cost = pc[i];
            }
        }
    }
    if (abs(min_peak - RADIUS) > fa.weak_peak_radius || pc[min_peak] > 0.5f) {
        fa.weak_info[center] = APD_WEAK;
        return;
    }
    if (peak_count == 1) {
        fa.weak_info[center] = (pc[min_peak] <= 0.15f) ? APD_STRONG : APD_WEAK;
        return;
    }
    float var = 0.0f;
    for (int i = 2; i < NP - 2; ++i) {
        if (((peaks >> i) & 1ull) && i != min_peak) {
            const float dist = pc[i] - min_cost;
            var += dist * dist;
        }
    }
    var = sqrtf(var);
    var /= (float)(peak_count - 1);
    fa.weak_info[center] = (var > 0.2f) ? APD_STRONG : APD_WEAK;
}

template <bool kQuad>
__global__ __launch_bounds__(256) void k15_local_refine(FrameArgs fa)
{
    __shared__ float tile[kFullLdsH * kFullPitch];
    int px, py;
    full_frame_pixel(px, py);
    const RefPatchLds<kFullPitch> rp = stage_full_frame_ref(fa, tile, px, py);
    if (px >= fa.W || py >= fa.H) {
        return;
    }
    const int W = fa.W;
    const int center = px + py * W;
    const float4 origin = normal_world_to_cam(fa, fa.planes[center]);
    const float origin_depth = origin.w;
    if (origin_depth == 0) {
        return;
    }
    const uint32_t sel = fa.selected_views[center];
    ViewWeights<32> vw;
    vw.load(fa, center);
    float base_line, weight_normal;
    const int valid = baseline_and_weight(fa, sel, vw, base_line, weight_normal);
    if (weight_normal == 0 || valid == 0) {
        return;
    }
    base_line /= (float)valid;
    const float disp = fa.K[0] * base_line / origin_depth;
    const int radius = 5;
    float cost_now = 0.0f;
    float min_cost = 2.0f;
    float best_depth = origin_depth;
    // pd == -radius-1 evaluates the current depth with K14's cost form (:2173-2183)
#pragma unroll 1
    for (int pd = -radius - 1; pd <= radius; ++pd) {
        if (pd == -radius - 1) {
            cost_now = disparity_sample_cost<false, kQuad>(fa, rp, px, py, origin, origin_depth, sel, vw) / weight_normal;
            continue;
        }
        const float p_depth = fa.K[0] * base_line / (disp + (float)pd);
        if (p_depth < fa.depth_min || p_depth > fa.depth_max) {
            continue;
        }
        const float tc = disparity_sample_cost<true, kQuad>(fa, rp, px, py, origin, p_depth, sel, vw) / weight_normal;
        if (tc < min_cost) {
            min_cost = tc;
            best_depth = p_depth;
        }
    }
    if ((double)(cost_now - min_cost) > 0.1) {
        fa.planes[center].w = best_depth;
    }
}

// ------------------------------------------------------------------------------------------------
// texel-quad images (built once per upload)
// ------------------------------------------------------------------------------------------------

// *flag stays 1 only if every pixel is an integer in [0, 255] (8-bit input at scale 1)
__global__ __launch_bounds__(256) void k_check_u8(const float *__restrict__ img, int n, int *flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = img[i];
        if (!(v >= 0.0f && v <= 255.0f && v == floorf(v))) {
            *flag = 0;
        }
    }
}

// round 1's row-major 4-byte quads (-DAPD_QUAD4): entry (qx, qy) = {I(qx-1,qy-1), I(qx,qy-1), I(qx-1,qy), I(qx,qy)}, clamped
__global__ __launch_bounds__(256) void k_pack_quads(const float *__restrict__ img, int W, int H, quad_t *__restrict__ quad)
{
    const int qx = blockIdx.x * 32 + (threadIdx.x & 31);  // 0..W  <-> image x = qx - 1
    const int qy = blockIdx.y * 8 + (threadIdx.x >> 5);   // 0..H
    if (qx > W || qy > H) {
        return;
    }
    const uint32_t t00 = (uint32_t)fetch_texel(img, W, H, qx - 1, qy - 1), t10 = (uint32_t)fetch_texel(img, W, H, qx, qy - 1);
    const uint32_t t01 = (uint32_t)fetch_texel(img, W, H, qx - 1, qy), t11 = (uint32_t)fetch_texel(img, W, H, qx, qy);
    quad[(size_t)qy * (W + 1) + qx] = t00 | (t10 << 8) | (t01 << 16) | (t11 << 24);
}

// 2-byte column pairs (the default row-major copy, apd_device.h): entry (t, u), t in [0, W + 1], u in [0, H], =
// {I(t - 1, u - 1), I(t - 1, u)} with clamped coordinates
__global__ __launch_bounds__(256) void k_pack_pairs(const float *__restrict__ img, int W, int H, uint16_t *__restrict__ pairs)
{
    const int t = blockIdx.x * 32 + (threadIdx.x & 31);
    const int u = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (t > W + 1 || u > H) {
        return;
    }
    const uint32_t top = (uint32_t)fetch_texel(img, W, H, t - 1, u - 1), bot = (uint32_t)fetch_texel(img, W, H, t - 1, u);
    pairs[(size_t)u * (W + 2) + t] = (uint16_t)(top | (bot << 8));
}

// the tiled copy (apd_device.h: quad_tiled_offset_tu).  kPair2: one thread per (tile row, slot): slots 0..6 are the tile's own
// columns, slot 7 repeats the first column of the next tile so that every dword of the tile holds two consecutive pairs
__global__ __launch_bounds__(256) void k_pack_quads_tiled(const float *__restrict__ img, int W, int H, quad_t *__restrict__ quad)
{
    if (kPair2) {
        const unsigned tiles_x = quad_tiles_x(W), tiles_y = quad_tiles_y(H);
        const unsigned gid = blockIdx.x * 256u + threadIdx.x;  // (tile, row in tile, slot)
        const unsigned slot = gid & 7u, iy = (gid >> 3) & 7u, tile = gid >> 6;
        if (tile >= tiles_x * tiles_y) {
            return;
        }
        const unsigned ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int t = (int)(tx * 7u + slot), u = (int)(ty * 8u + iy);
        const uint32_t top = (uint32_t)fetch_texel(img, W, H, t - 1, u - 1), bot = (uint32_t)fetch_texel(img, W, H, t - 1, u);
        reinterpret_cast<uint16_t *>(quad)[(size_t)tile * 64u + iy * 8u + slot] = (uint16_t)(top | (bot << 8));
        return;
    }
    const int qx = blockIdx.x * 256 + threadIdx.x;  // flat over (W + 1) x (H + 1) entries
    const int n = (W + 1) * (H + 1);
    if (qx >= n) {
        return;
    }
    const int ex = qx % (W + 1), ey = qx / (W + 1);
    const uint32_t t00 = (uint32_t)fetch_texel(img, W, H, ex - 1, ey - 1), t10 = (uint32_t)fetch_texel(img, W, H, ex, ey - 1);
    const uint32_t t01 = (uint32_t)fetch_texel(img, W, H, ex - 1, ey), t11 = (uint32_t)fetch_texel(img, W, H, ex, ey);
    *reinterpret_cast<quad_t *>(reinterpret_cast<char *>(quad) + quad_tiled_offset_tu((unsigned)ex, (unsigned)ey, quad_tiles_x(W))) =
        t00 | (t10 << 8) | (t01 << 16) | (t11 << 24);
}

// float texel quads of a float image: entry (qx, qy), qx in [-1, W-1], qy in [-1, H-1] (clamped coordinates)
__global__ __launch_bounds__(256) void k_pack_fquads(const float *__restrict__ img, int W, int H, fquad_t *__restrict__ fq)
{
    const int qx = blockIdx.x * 32 + (threadIdx.x & 31);  // 0..W  <-> image x = qx - 1
    const int qy = blockIdx.y * 8 + (threadIdx.x >> 5);   // 0..H  <-> image y = qy - 1
    if (qx > W || qy > H) {
        return;
    }
    const float t00 = fetch_texel(img, W, H, qx - 1, qy - 1), t10 = fetch_texel(img, W, H, qx, qy - 1);
    const float t01 = fetch_texel(img, W, H, qx - 1, qy), t11 = fetch_texel(img, W, H, qx, qy);
    fq[(size_t)qy * (W + 1) + qx] = fquad_t{t00, t10 - t00, t01, t11 - t01};
}

hipError_t launch_pack_fquads(const float *img, int W, int H, fquad_t *fq, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_fquads, dim3((W + 1 + 31) / 32, (H + 1 + 7) / 8), dim3(256), 0, s, img, W, H, fq);
    return hipGetLastError();
}

hipError_t launch_check_u8(const float *img, int n, int *flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_check_u8, dim3((n + 255) / 256), dim3(256), 0, s, img, n, flag);
    return hipGetLastError();
}

hipError_t launch_pack_quads(const float *img, int W, int H, quad_t *quad, hipStream_t s)
{
    if (kPair2) {
        hipLaunchKernelGGL(k_pack_pairs, dim3((W + 2 + 31) / 32, (H + 1 + 7) / 8), dim3(256), 0, s, img, W, H, reinterpret_cast<uint16_t *>(quad));
    } else {
        hipLaunchKernelGGL(k_pack_quads, dim3((W + 1 + 31) / 32, (H + 1 + 7) / 8), dim3(256), 0, s, img, W, H, quad);
    }
    return hipGetLastError();
}

hipError_t launch_pack_quads_tiled(const float *img, int W, int H, quad_t *quad, hipStream_t s)
{
    const size_t threads = kPair2 ? (size_t)quad_tiles_x(W) * quad_tiles_y(H) * 64u : (size_t)(W + 1) * (H + 1);
    hipLaunchKernelGGL(k_pack_quads_tiled, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, img, W, H, quad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------

static inline dim3 grid_full_frame(const FrameArgs &fa) { return dim3((fa.W + kFullTileW - 1) / kFullTileW, (fa.H + kFullTileH - 1) / kFullTileH); }
static inline int checkerboard_tiles(const FrameArgs &fa) { return ((fa.W + kTileW - 1) / kTileW) * ((fa.H + kTileH - 1) / kTileH); }

hipError_t launch_k67_windowed(const FrameArgs &fa, int colour, int iter, hipStream_t s);  // apd_kernels_k67w.hip

// APD_OPT_K67_WINDOWS = 0 selects the kernel without the LDS source windows (A/B timing and the window-vs-global parity
// test; same results)

hipError_t launch_k14_windowed(const FrameArgs &fa, hipStream_t s);  // apd_kernels_k1415w.hip
hipError_t launch_k15_windowed(const FrameArgs &fa, hipStream_t s);

// APD_OPT_K1415_WINDOWS = 0: K14/K15 without LDS source windows (same results)

template <int NMAX>
static void launch_k67(const FrameArgs &fa, int colour, int iter, hipStream_t s)
{
    if (fa.use_quads) {
        hipLaunchKernelGGL((k67_update_strong<NMAX, true>), dim3(checkerboard_tiles(fa)), dim3(256), 0, s, fa, colour, iter);
    } else {
        hipLaunchKernelGGL((k67_update_strong<NMAX, false>), dim3(checkerboard_tiles(fa)), dim3(256), 0, s, fa, colour, iter);
    }
}

hipError_t launch_kernel(const FrameArgs &fa, int kernel_id, int iter, hipStream_t s)
{
    switch (kernel_id) {
    case APD_K1_INIT_RANDOM_STATES: {
        const int segs = ((fa.W + kRngSegment - 1) / kRngSegment) * fa.H;
        hipLaunchKernelGGL(k1_init_random_states, dim3((segs + 63) / 64), dim3(64), 0, s, fa);
        break;
    }
    case APD_K5_RANDOM_INITIALIZATION:
        if (fa.use_quads) {
            if (fa.have_tiled && fa.state == APD_FIRST_INIT) {
                hipLaunchKernelGGL((k5_random_initialization<true, true>), grid_full_frame(fa), dim3(256), 0, s, fa);
            } else {
                hipLaunchKernelGGL((k5_random_initialization<true, false>), grid_full_frame(fa), dim3(256), 0, s, fa);
            }
        } else {
            hipLaunchKernelGGL((k5_random_initialization<false, false>), grid_full_frame(fa), dim3(256), 0, s, fa);
        }
        break;
    case APD_K6_BLACK_UPDATE_STRONG:
    case APD_K7_RED_UPDATE_STRONG: {
        const int colour = (kernel_id == APD_K6_BLACK_UPDATE_STRONG) ? 0 : 1;
        if (fa.k67_windows) {
            return launch_k67_windowed(fa, colour, iter, s);
        }
        if (fa.num_src <= 8) {
            launch_k67<8>(fa, colour, iter, s);
        } else if (fa.num_src <= 12) {
            launch_k67<12>(fa, colour, iter, s);
        } else if (fa.num_src <= 16) {
            launch_k67<16>(fa, colour, iter, s);
        } else {
            launch_k67<32>(fa, colour, iter, s);
        }
        break;
    }
    case APD_K11_GET_DEPTH_NORMAL:
        hipLaunchKernelGGL(k11_depth_and_normal, dim3((fa.W * fa.H + 255) / 256), dim3(256), 0, s, fa);
        break;
    case APD_K12_BLACK_FILTER:
    case APD_K13_RED_FILTER:
        hipLaunchKernelGGL(k1213_filter_strong, dim3(checkerboard_tiles(fa)), dim3(256), 0, s, fa,
                           (kernel_id == APD_K12_BLACK_FILTER) ? 0 : 1);
        break;
    case APD_K14_DEPTH_TO_WEAK:
        if (fa.k1415_windows) {
            return launch_k14_windowed(fa, s);
        }
        if (fa.use_quads) {
            hipLaunchKernelGGL(k14_depth_to_weak<true>, grid_full_frame(fa), dim3(256), 0, s, fa);
        } else {
            hipLaunchKernelGGL(k14_depth_to_weak<false>, grid_full_frame(fa), dim3(256), 0, s, fa);
        }
        break;
    case APD_K15_LOCAL_REFINE:
        if (fa.k1415_windows) {
            return launch_k15_windowed(fa, s);
        }
        if (fa.use_quads) {
            hipLaunchKernelGGL(k15_local_refine<true>, grid_full_frame(fa), dim3(256), 0, s, fa);
        } else {
            hipLaunchKernelGGL(k15_local_refine<false>, grid_full_frame(fa), dim3(256), 0, s, fa);
        }
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace apd
