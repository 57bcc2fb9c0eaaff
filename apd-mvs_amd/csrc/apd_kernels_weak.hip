// apd_kernels_weak.hip -- adaptive-patch-deformation kernels (the "APD" of APD-MVS) for gfx950:
// K2 FindNearestStrongPoint, K3 GenNeighbours, K4 NeigbourUpdate, K8 RANSACToGetFitPlane,
// K9/K10 Black/RedPixelUpdateWeak, plus the depth/normal export used before the RCCL all-gather.
#include "apd_device.h"
#include "apd_sweep.h"
#include "apd_window.h"

#include <float.h>

#include <type_traits>

namespace apd {


// ------------------------------------------------------------------------------------------------
// K2  FindNearestStrongPoint (APD.cu:2234-2270)
// ------------------------------------------------------------------------------------------------

// The reference probes the 201x201 window column by column (x outer, y inner) and keeps the first
// strictly smaller distance, i.e. the minimum of d2 = dx^2 + dy^2 with ties going to the smallest
// (dx, dy) in lexicographic order.  That minimum separates: inside one column the winner is the
// smallest |dy| (negative dy first), so pass A stores that dy per pixel and pass B scans the 201
// columns of a WEAK pixel's row -- 402 probes instead of 40,401, same result bit for bit.
constexpr int kNearestRadius = 100;
constexpr int8_t kNoStrongInColumn = 127;

__global__ __launch_bounds__(256) void k2a_column_nearest(FrameArgs fa)
{
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = fa.W, H = fa.H;
    if (px >= W || py >= H) {
        return;
    }
    const uint8_t *__restrict__ wi = fa.weak_info;
    int8_t best = kNoStrongInColumn;
    for (int k = 0; k <= kNearestRadius; ++k) {
        if (py - k >= 0 && wi[px + (py - k) * W] == APD_STRONG) {
            best = (int8_t)(-k);
            break;
        }
        if (py + k < H && wi[px + (py + k) * W] == APD_STRONG) {
            best = (int8_t)k;
            break;
        }
    }
    fa.column_nearest[px + py * W] = best;
}

__global__ __launch_bounds__(256) void k2b_row_search(FrameArgs fa)
{
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = fa.W;
    if (px >= W || py >= fa.H) {
        return;
    }
    const int center = px + py * W;
    short2 out = make_short2(-1, -1);
    if (fa.weak_info[center] == APD_WEAK) {
        // sqrtf is monotone and the reference compares sqrt values: two different integer d2 can round
        // to the same float only above 2^24, far beyond 2*100^2, so comparing d2 is equivalent; its
        // initial min_dist = 255.0f exceeds every distance in the window.
        const int8_t *__restrict__ row = fa.column_nearest + py * W;
        int best_d2 = 0x7fffffff;
        const int x_lo = max(px - kNearestRadius, 0), x_hi = min(px + kNearestRadius, W - 1);
        for (int qx = x_lo; qx <= x_hi; ++qx) {
            const int dy = row[qx];
            if (dy != kNoStrongInColumn) {
                const int dx = qx - px;
                const int d2 = dx * dx + dy * dy;
                if (d2 < best_d2) {
                    best_d2 = d2;
                    out = make_short2((short)qx, (short)(py + dy));
                }
            }
        }
    }
    fa.nearest_strong[center] = out;
}

// ------------------------------------------------------------------------------------------------
// K3  GenNeighbours (APD.cu:1750-1969)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool point_in_triangle(short2 A, short2 B, short2 C, int px, int py)  // APD.cu:91-112
{
    const float ABx = (float)(B.x - A.x), ABy = (float)(B.y - A.y);
    const float BCx = (float)(C.x - B.x), BCy = (float)(C.y - B.y);
    const float CAx = (float)(A.x - C.x), CAy = (float)(A.y - C.y);
    const float ab = sqrtf(ABx * ABx + ABy * ABy);
    const float bc = sqrtf(BCx * BCx + BCy * BCy);
    const float ca = sqrtf(CAx * CAx + CAy * CAy);
    if (ab <= 2 || bc <= 2 || ca <= 2) {
        return false;
    }
    if (!(ab + bc > ca && bc + ca > ab && ab + ca > bc)) {
        return false;
    }
    const float PAx = (float)(A.x - px), PAy = (float)(A.y - py);
    const float PBx = (float)(B.x - px), PBy = (float)(B.y - py);
    const float PCx = (float)(C.x - px), PCy = (float)(C.y - py);
    const float t1 = PAx * PBy - PAy * PBx;
    const float t2 = PBx * PCy - PBy * PCx;
    const float t3 = PCx * PAy - PCy * PAx;
    return t1 * t2 >= 0 && t1 * t3 >= 0;
}

// (curand()%2==0 ? 1 : -1) * curand() % range in unsigned arithmetic; sign draw first (:1813-1814)
__device__ __forceinline__ int jitter_shift(Rng &rng, int range)
{
    const uint32_t sign = (rng_next(rng) % 2u == 0u) ? 1u : 0xFFFFFFFFu;
    const uint32_t mag = rng_next(rng);
    return (int)((sign * mag) % (uint32_t)range);
}

__device__ __forceinline__ void sort_points_by_weight(short2 *pts, float *w, int n)  // APD.cu:14-27
{
    for (int i = 1; i < n; ++i) {
        const short2 p = pts[i];
        const float v = w[i];
        int j = i;
        while (j >= 1 && v < w[j - 1]) {
            pts[j] = pts[j - 1];
            w[j] = w[j - 1];
            --j;
        }
        pts[j] = p;
        w[j] = v;
    }
}

// One lane per entry of a compacted WEAK list (build_weak_lists below; K3 visits both colours): a per-pixel launch fills
// 18 % of its lanes on a typical frame and the ray search of those diverges against nothing.  In list order the lanes of a
// wave are image neighbours, which walk their rays in step.
//
// kCut: `dist / depth_diff < ransac_threshold` (:1911, :1946) is evaluated as `dist < fa.k3_dist_cut`.  x -> RN(x / d) is
// monotone for d > 0, so the set of non-negative floats that pass the test is an initial segment; the host finds its end
// with IEEE divisions (ransac_distance_cut, apd_capi.hip) -- 1,600 divisions per WEAK pixel become comparisons, same bits.
// kNoJitter: with the reference's rotate_time = 4 the jitter range (int)(tan(5.625 deg) * 20) is 1, every `% range` is 0 and
// the four attempts of a (slot, radius) probe the same pixel: one probe, and the twelve draws of the three repeats are
// skipped over when it fails.
// The candidate points live twice: compacted in scratch memory for the three randomly indexed reads of a RANSAC draw, and
// slot-indexed in registers for the inlier count, which visits all of them in a fixed order (an integer count has no order).
template <bool kCut, bool kNoJitter>
__global__ __launch_bounds__(64) void k3_gen_neighbours(FrameArgs fa, const int *__restrict__ list, int count)
{
    const int gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= count) {
        return;
    }
    const int W = fa.W, H = fa.H;
    const int center = list[gid];
    const int py = center / W, px = center - py * W;
    const int min_margin = 6;
    const float depth_diff = fa.depth_max - fa.depth_min;
    Rng rng = rng_load(fa.rng, center);
    short2 *nb = &fa.neighbours[(size_t)fa.neighbours_map[center] * APD_NEIGHBOUR_NUM];
    for (int i = 0; i < APD_NEIGHBOUR_NUM; ++i) {
        nb[i] = make_short2(-1, -1);
    }
    nb[0] = make_short2((short)px, (short)py);
    short2 strong_pts[32];
    unsigned dir_valid = 0;
    for (int i = 0; i < 32; ++i) {
        strong_pts[i] = make_short2(-1, -1);
    }
    int dir_base = -1, found = 0;
    const int rotate_time = fa.rotate_time;
    const int shift_range = fa.k3_shift_range;
    for (int ox = -1; ox <= 1; ++ox) {
        for (int oy = -1; oy <= 1; ++oy) {
            if (ox == 0 && oy == 0) {
                continue;
            }
            float odx = (float)ox, ody = (float)oy;
            normalize2(odx, ody);
            dir_base++;
            for (int rot = 0; rot < rotate_time; ++rot) {
                const int slot = dir_base * 4 + rot;
                for (int radius = 2; radius <= APD_MAX_SEARCH_RADIUS; radius = min(radius * 2, radius + 25)) {
                    const float tx = (float)px + odx * (float)radius;
                    const float ty = (float)py + ody * (float)radius;
                    if (tx < 0 || ty < 0 || tx >= (float)W || ty >= (float)H) {
                        break;
                    }
                    for (int attempt = 0; attempt < (kNoJitter ? 1 : 4); ++attempt) {
                        int sx = 0, sy = 0;
                        if constexpr (kNoJitter) {
                            rng_next(rng);
                            rng_next(rng);
                            rng_next(rng);
                            rng_next(rng);
                        } else {
                            sx = jitter_shift(rng, shift_range);
                            sy = jitter_shift(rng, shift_range);
                        }
                        float dirx = odx * 20 + (float)sx, diry = ody * 20 + (float)sy;
                        normalize2(dirx, diry);
                        short2 q = make_short2((short)((float)px + dirx * (float)radius), (short)((float)py + diry * (float)radius));
                        if (q.x < min_margin || q.y < min_margin || q.x >= W - min_margin || q.y >= H - min_margin) {
                            continue;
                        }
                        int qc = q.x + q.y * W;
                        if (fa.weak_info[qc] != APD_STRONG) {
                            q = fa.nearest_strong[qc];
                            if (q.x == -1 || q.y == -1) {
                                continue;
                            }
                            qc = q.x + q.y * W;
                        }
                        float tdx = (float)(q.x - px), tdy = (float)(q.y - py);
                        normalize2(tdx, tdy);
                        const float ca = tdx * odx + tdy * ody;
                        if (ca > fa.k3_cone) {
                            strong_pts[slot] = q;
                            dir_valid |= 1u << slot;
                            found++;
                            break;
                        }
                    }
                    if (dir_valid & (1u << slot)) {
                        break;
                    }
                    if constexpr (kNoJitter) {  // the three repeats of the failed probe: four draws each
#pragma unroll
                        for (int k = 0; k < 12; ++k) {
                            rng_next(rng);
                        }
                    }
                }
                {
                    float rdx = odx * fa.k3_cos_angle - ody * fa.k3_sin_angle;
                    float rdy = odx * fa.k3_sin_angle + ody * fa.k3_cos_angle;
                    normalize2(rdx, rdy);
                    odx = rdx;
                    ody = rdy;
                }
            }
        }
    }
    if (found <= 3) {
        fa.weak_reliable[center] = 0;
        rng_store(fa.rng, center, rng);
        return;
    }
    float4 best_plane = make_float4(0, 0, 0, 0);
    int use_a = -1, use_b = -1, use_c = -1;
    bool has_plane = false;
    short2 pts[32];
    float3 pts3d[32];
    float rx[32], ry[32], rz[32];  // slot-indexed copy, static indices only: registers
    int valid = 0;
    float Xc, Yc, Zc;
    point3d(fa, px, py, fa.planes[center].w, Xc, Yc, Zc);  // .w still holds the DEPTH before K5 (:1866)
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        pts[i] = make_short2(-1, -1);
        rx[i] = ry[i] = rz[i] = 0.0f;
        if (dir_valid & (1u << i)) {
            const short2 sp = strong_pts[i];
            pts[valid] = sp;
            float X, Y, Z;
            point3d(fa, sp.x, sp.y, fa.planes[sp.x + sp.y * W].w, X, Y, Z);
            pts3d[valid] = make_float3(X, Y, Z);
            rx[i] = X;
            ry[i] = Y;
            rz[i] = Z;
            valid++;
        }
    }
    {
        int iteration = 50;
        float min_cost = FLT_MAX;
        int max_count = 3;
        while (iteration--) {
            const int a = (int)(rng_next(rng) % (uint32_t)valid);
            const int b = (int)(rng_next(rng) % (uint32_t)valid);
            const int c = (int)(rng_next(rng) % (uint32_t)valid);
            if (a == b || b == c || a == c) {
                continue;
            }
            if (!point_in_triangle(pts[a], pts[b], pts[c], px, py)) {
                continue;
            }
            const float3 A = pts3d[a], B = pts3d[b], C = pts3d[c];
            const float ACx = A.x - C.x, ACy = A.y - C.y, ACz = A.z - C.z;
            const float BCx = B.x - C.x, BCy = B.y - C.y, BCz = B.z - C.z;
            float nx = ACy * BCz - BCy * ACz;
            float ny = -(ACx * BCz - BCx * ACz);
            float nz = ACx * BCy - BCx * ACy;
            if ((nx == 0 && ny == 0 && nz == 0) || nx != nx || ny != ny || nz != nz) {
                continue;
            }
            normalize3(nx, ny, nz);
            const float nw = -(nx * A.x + ny * A.y + nz * A.z);
            int count = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float dist = fabsf(nx * rx[k] + ny * ry[k] + nz * rz[k] + nw);
                const bool inlier = kCut ? dist < fa.k3_dist_cut : dist / depth_diff < fa.ransac_threshold;
                count += (((dir_valid >> k) & 1u) != 0u && inlier) ? 1 : 0;
            }
            if (count < 6) {
                continue;
            }
            const float cdist = fabsf(nx * Xc + ny * Yc + nz * Zc + nw);
            if (count > max_count) {
                max_count = count;
                min_cost = cdist;
                best_plane = make_float4(nx, ny, nz, nw);
                has_plane = true;
                use_a = a;
                use_b = b;
                use_c = c;
            } else if (count == max_count) {
                if (cdist < min_cost) {
                    min_cost = cdist;
                    best_plane = make_float4(nx, ny, nz, nw);
                    use_a = a;
                    use_b = b;
                    use_c = c;
                }
            }
        }
    }
    rng_store(fa.rng, center, rng);
    if (!has_plane) {
        fa.weak_reliable[center] = 0;
        return;
    }
    float weight[32];
    for (int i = 0; i < valid; ++i) {
        const float3 P = pts3d[i];
        float dist = fabsf(best_plane.x * P.x + best_plane.y * P.y + best_plane.z * P.z + best_plane.w);
        if (kCut ? dist >= fa.k3_dist_cut : dist / depth_diff >= fa.ransac_threshold) {
            pts[i] = make_short2(-1, -1);
            weight[i] = FLT_MAX;
            continue;
        }
        if (i == use_a || i == use_b || i == use_c) {
            dist -= 1;
        }
        weight[i] = dist;
    }
    sort_points_by_weight(pts, weight, valid);
    for (int i = 1; i < APD_NEIGHBOUR_NUM; ++i) {
        nb[i] = pts[i - 1];
    }
    fa.weak_reliable[center] = 1;
}

// ------------------------------------------------------------------------------------------------
// K4  NeigbourUpdate (APD.cu:1971-1987)
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k4_neighbour_update(FrameArgs fa)
{
    const int center = blockIdx.x * 256 + threadIdx.x;
    if (center >= fa.W * fa.H) {
        return;
    }
    if (fa.weak_info[center] == APD_WEAK && fa.weak_reliable[center] != 1) {
        fa.weak_info[center] = APD_UNKNOWN;
    }
}

// ------------------------------------------------------------------------------------------------
// K8  RANSACToGetFitPlane (APD.cu:2272-2384)
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k8_ransac_fit_plane(FrameArgs fa)
{
    // The up to eight neighbour points of a lane are indexed by random draws: kept in LDS ([slot][lane], conflict free)
    // instead of private arrays, which the compiler can only place in scratch memory (50 draws x ~11 dependent scratch
    // loads per WEAK pixel made this kernel latency bound: 9.0 -> 2.6 ms per launch at 4096x3072, 18 % WEAK).  The same change
    // makes K3 slower (15 -> 33 ms): its 32 candidate slots need 40 KB per wave, four waves per CU.
    __shared__ float lds_x[8][256], lds_y[8][256], lds_z[8][256];
    __shared__ int lds_p[8][256];
    const int lane = threadIdx.x;
    const int center = blockIdx.x * 256 + threadIdx.x;
    const int W = fa.W;
    if (center >= W * fa.H) {
        return;
    }
    const float4 pl = fa.planes[center];
    if (fa.weak_info[center] != APD_WEAK) {
        fa.fit_planes[center] = pl;
        return;
    }
    const int py = center / W, px = center - py * W;
    Rng rng = rng_load(fa.rng, center);
    const short2 *nb = &fa.neighbours[(size_t)fa.neighbours_map[center] * APD_NEIGHBOUR_NUM];
    int count = 0;
    for (int i = 1; i < APD_NEIGHBOUR_NUM; ++i) {
        const short2 q = nb[i];
        if (q.x == -1 || q.y == -1) {
            continue;
        }
        lds_p[count][lane] = (int)(unsigned short)q.x | ((int)q.y << 16);
        const float depth = depth_from_plane(fa, fa.planes[q.x + q.y * W], q.x, q.y);
        float X, Y, Z;
        point3d(fa, q.x, q.y, depth, X, Y, Z);
        lds_x[count][lane] = X;
        lds_y[count][lane] = Y;
        lds_z[count][lane] = Z;
        count++;
    }
    if (count < 3) {
        fa.fit_planes[center] = pl;
        return;
    }
    int iteration = 50;
    float min_cost = FLT_MAX;
    float4 best = make_float4(0, 0, 0, 0);
    bool has_best = false;
    while (iteration--) {
        const int a = (int)(rng_next(rng) % (uint32_t)count);
        const int b = (int)(rng_next(rng) % (uint32_t)count);
        const int c = (int)(rng_next(rng) % (uint32_t)count);
        if (a == b || b == c || a == c) {
            continue;
        }
        const int pa = lds_p[a][lane], pb = lds_p[b][lane], pc = lds_p[c][lane];
        if (!point_in_triangle(make_short2((short)(pa & 0xFFFF), (short)(pa >> 16)), make_short2((short)(pb & 0xFFFF), (short)(pb >> 16)),
                               make_short2((short)(pc & 0xFFFF), (short)(pc >> 16)), px, py)) {
            continue;
        }
        const float3 A = make_float3(lds_x[a][lane], lds_y[a][lane], lds_z[a][lane]);
        const float3 B = make_float3(lds_x[b][lane], lds_y[b][lane], lds_z[b][lane]);
        const float3 C = make_float3(lds_x[c][lane], lds_y[c][lane], lds_z[c][lane]);
        const float ACx = A.x - C.x, ACy = A.y - C.y, ACz = A.z - C.z;
        const float BCx = B.x - C.x, BCy = B.y - C.y, BCz = B.z - C.z;
        float nx = ACy * BCz - BCy * ACz;
        float ny = -(ACx * BCz - BCx * ACz);
        float nz = ACx * BCy - BCx * ACy;
        if ((nx == 0 && ny == 0 && nz == 0) || nx != nx || ny != ny || nz != nz) {
            continue;
        }
        normalize3(nx, ny, nz);
        const float nw = -(nx * A.x + ny * A.y + nz * A.z);
        float tc = 0.0f;
        for (int k = 0; k < count; ++k) {
            if (k == a || k == b || k == c) {
                continue;
            }
            tc += fabsf(nx * lds_x[k][lane] + ny * lds_y[k][lane] + nz * lds_z[k][lane] + nw);
        }
        if (tc < min_cost) {
            min_cost = tc;
            best = make_float4(nx, ny, nz, nw);
            has_best = true;
        }
        if (min_cost == 0) {
            break;
        }
    }
    rng_store(fa.rng, center, rng);
    if (has_best) {
        const float depth = depth_from_plane(fa, pl, px, py);
        float vx, vy, vz;
        view_direction(fa, px, py, depth, vx, vy, vz);
        const float dot = best.x * vx + best.y * vy + best.z * vz;
        if (dot > 0) {
            best = make_float4(-best.x, -best.y, -best.z, -best.w);
        }
        fa.fit_planes[center] = best;
    } else {
        fa.fit_planes[center] = make_float4(0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// K9/K10  Black/RedPixelUpdateWeak -> CheckerboardPropagationWeak (APD.cu:1323-1508, 892-980)
// ------------------------------------------------------------------------------------------------

// ComputeBilateralNCCNew (APD.cu:400-528): centre 6x6 patch + up to eight 3x3 sub-patches around the
// reliable neighbours, all warped by the same homography.
//
// Per-pixel, hypothesis-independent data of the eight neighbours lives in LDS ([slot][lane], so a wave
// reads consecutive banks): position, the nine reference texels of the 3x3 sub-patch (texel-quad mode:
// bytes, three per dword) and their mean / variance in the reference's summation order.
template <bool kQuad>
struct WeakLdsT {
    int nb[8][64];            // x | y << 16, -1 = empty slot
    // reference texels of the 3x3 sub-patches: texel-quad mode three per dword (bytes), else nine floats
    typename std::conditional<kQuad, uint32_t[8][kSubN][64], float[8][kSubN * kSubN][64]>::type ref;
    // Moments of the sub-patches.  Float images: mean and variance as computed by weak_prepare_neighbours.  Texel-quad mode: every
    // texel is an integer 0..255, so the sub-patch's texel sum S (< 2^12) and sum of squares Q (< 2^20) are exact integers in
    // binary32 and mean = S * (1/9), var = fma(-mean, mean, Q * (1/9)) are formed again from them, with the same bits, where they
    // are used: Q rides in the unused top bytes of the three row dwords (byte 3 of row i = bits 8i .. 8i+7), S is the byte sum of
    // the nine texels (v_sad_u8).  4 KB less LDS per workgroup, which the centre-patch window (below) gets.
    float mean[kQuad ? 1 : 8][kQuad ? 1 : 64];
    float var[kQuad ? 1 : 8][kQuad ? 1 : 64];
    uint32_t centre[kQuad ? kPatchN * kPatchN / 4 : 1][64];  // texel-quad mode: the pixel's own 36 reference texels as bytes

    // sum_r, sum_rr: texel sum and sum of squares of sub-patch k before the division by nine
    __device__ __forceinline__ void store_sub(int k, int lane, const uint32_t (&rows)[kSubN], float sum_r, float sum_rr)
    {
        if constexpr (kQuad) {
            const uint32_t q = (uint32_t)sum_rr;
#pragma unroll
            for (int i = 0; i < kSubN; ++i) {
                ref[k][i][lane] = rows[i] | (((q >> (8 * i)) & 0xFFu) << 24);
            }
        } else {
            const float inv_w = 1.0f / 9.0f;
            sum_r *= inv_w;
            sum_rr *= inv_w;
            mean[k][lane] = sum_r;
            var[k][lane] = fmaf(-sum_r, sum_r, sum_rr);
        }
    }
    __device__ __forceinline__ void load_sub(int k, int lane, uint32_t (&rows)[kSubN], float &mean_r, float &var_r) const
    {
        if constexpr (kQuad) {
            uint32_t s = 0, q = 0;
#pragma unroll
            for (int i = 0; i < kSubN; ++i) {
                rows[i] = ref[k][i][lane];
                s = __builtin_amdgcn_sad_u8(rows[i] & 0x00FFFFFFu, 0u, s);
                q |= (rows[i] >> 24) << (8 * i);
            }
            const float inv_w = 1.0f / 9.0f;
            mean_r = (float)s * inv_w;
            var_r = fmaf(-mean_r, mean_r, (float)q * inv_w);
        } else {
            mean_r = mean[k][lane];
            var_r = var[k][lane];
        }
    }
};
typedef WeakLdsT<true> WeakLds;

// The pixel's own 6x6 reference patch kept as bytes in LDS (texel-quad mode: every texel is an integer 0..255):
// frees 36 VGPRs per lane in the register-hungry weak kernel.  Same interface as RefPatch / RefPatchLds.
struct RefPatchBytes {
    const uint32_t *base;  // &lds.centre[0][lane]
    float mean, var;
    __device__ __forceinline__ float at(int i, int j) const
    {
        const int idx = i * kPatchN + j;
        return (float)((base[(idx >> 2) * 64] >> (8 * (idx & 3))) & 0xFFu);
    }
    static constexpr bool kRuntimeIndex = true;
};

__device__ __forceinline__ RefPatchBytes ref_patch_to_lds(const RefPatch &rp, WeakLds &lds, int lane)
{
#pragma unroll
    for (int w = 0; w < kPatchN * kPatchN / 4; ++w) {
        lds.centre[w][lane] = (uint32_t)rp.v[4 * w] | ((uint32_t)rp.v[4 * w + 1] << 8) | ((uint32_t)rp.v[4 * w + 2] << 16) |
                              ((uint32_t)rp.v[4 * w + 3] << 24);
    }
    RefPatchBytes out;
    out.base = &lds.centre[0][lane];
    out.mean = rp.mean;
    out.var = rp.var;
    return out;
}

template <bool kQuad>
__device__ __forceinline__ void weak_prepare_neighbours(const FrameArgs &fa, const short2 *nb, WeakLdsT<kQuad> &lds, int lane)
{
    const int W = fa.W, H = fa.H;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const short2 q = nb[k + 1];
        const bool valid = !(q.x == -1 || q.y == -1);
        lds.nb[k][lane] = valid ? ((int)(unsigned short)q.x | ((int)q.y << 16)) : -1;
        if (!valid) {
            continue;
        }
        float sum_r = 0.0f, sum_rr = 0.0f;
        uint32_t rows[kSubN] = {0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < kSubN; ++i) {
            float row_r = 0.0f, row_rr = 0.0f;
#pragma unroll
            for (int j = 0; j < kSubN; ++j) {
                const float r = fetch_texel(fa.ref_img, W, H, q.x + kSubStep * (i - 1), q.y + kSubStep * (j - 1));
                row_r += r;
                row_rr = fmaf(r, r, row_rr);
                if constexpr (kQuad) {
                    rows[i] |= (uint32_t)r << (8 * j);
                } else {
                    lds.ref[k][i * kSubN + j][lane] = r;
                }
            }
            sum_r += row_r;
            sum_rr += row_rr;
        }
        lds.store_sub(k, lane, rows, sum_r, sum_rr);
    }
}

// ComputeBilateralNCCNew in three pieces, so that K9/K10 can evaluate the sub-patch part of a (pixel, hypothesis) pair on another lane than
// the centre part (see the propagation phase of k910_update_weak); ncc_deformed below is the three in one.

// k == 0: the pixel itself with the strong geometry.  The caller has done the bounds test of APD.cu:546 on the projected centre.
// w: this wave's window of view v for the centre patch (texel-quad mode; valid = 0: none staged)
template <bool kQuad, typename Ref>
__device__ __forceinline__ float deformed_centre(const FrameArgs &fa, const ViewConst &vc, const Ref &rp, const Homography &H, int px, int py,
                                                 const SrcWindow &w)
{
    // Lanes whose 36 samples fall inside the wave's window read LDS, the others gather: only the latter cost L1 tag accesses, which bound this kernel.
    if constexpr (kQuad) {
        return ncc_fixed_windowed_from_h<true, kWinW, false, false, APD_K910_WIN_DIVERGENT != 0>(fa, vc, w, rp, H, px, py);
    } else {
        return ncc_fixed_from_h<kQuad>(fa, vc, rp, H, px, py);
    }
}

// k = 1 .. 8: the 3 x 3 sub-patches around the reliable neighbours of the pixel whose data sits in column `owner` of the wave's LDS
// tables, warped by H; summed in slot order (APD.cu:461-520).
template <bool kQuad>
__device__ __forceinline__ void deformed_strong(const FrameArgs &fa, const ViewConst &vc, int v, const WeakLdsT<kQuad> &lds, int owner,
                                                const Homography &H, float &strong_cost, int &strong_count)
{
#if APD_K910_SUBPATCH_TILED
    const global_quad_ptr srcq = (global_quad_ptr)vc.quad_tiled;   // needs --opt tiled_copy=2 (the copy is built for every pass)
    const unsigned qpitch = quad_tiles_x(fa.W);
#else
    const global_quad_ptr srcq = (global_quad_ptr)vc.quad;
    const unsigned qpitch = quad_row_pitch_bytes(fa.W);
#endif
    const unsigned fpitch = 16u * (unsigned)(fa.W + 1);
    const global_fquad_ptr srcf = (global_fquad_ptr)vc.fquad;
    const int wm1 = fa.W - 1, hm1 = fa.H - 1;
    strong_cost = 0.0f;
    strong_count = 0;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const int packed = lds.nb[k][owner];
        if (packed == -1) {
            continue;
        }
        const int nbx = (int)(short)(packed & 0xFFFF), nby = packed >> 16;
        float nx, ny;
        correspond(H, (float)nbx, (float)nby, nx, ny);
        if (nx < 0 || ny < 0 || nx >= (float)fa.W || ny >= (float)fa.H) {
            const uint32_t vi = fa.selected_views[nbx + nby * fa.W];
            if (bit_test(vi, (unsigned)v)) {
                strong_cost += 2.0f;
                strong_count++;
            }
            continue;
        }
        float c;
        APD_WEAK_COUNT(4, 1);
        APD_WEAK_COUNT_WAVE(5);
        // one nine-sample body per wave and sub-patch: the IEEE division gives the bits of the fast reciprocal wherever that
        // one is valid, so if one lane needs it (a sign change or an extreme denominator under a random normal) all take it
        const bool fast = denominators_fast(H, (float)(nbx - kSubStep), (float)(nbx + kSubStep), (float)(nby - kSubStep), (float)(nby + kSubStep));
        uint32_t ref_rows[kSubN] = {0u, 0u, 0u};
        float mean_r, var_r;
        lds.load_sub(k, owner, ref_rows, mean_r, var_r);
        if (__builtin_amdgcn_ballot_w64(!fast) == 0) {
            if constexpr (kQuad) {
                c = subpatch_cost_quad<kRecipExact>(H, srcq, qpitch, wm1, hm1, nbx, nby, ref_rows, mean_r, var_r);
            } else {
                c = subpatch_cost_fquad<kRecipExact>(H, srcf, fpitch, wm1, hm1, nbx, nby, &lds.ref[k][0][owner], 64, mean_r, var_r);
            }
        } else {
            if constexpr (kQuad) {
                c = subpatch_cost_quad<kRecipIeee>(H, srcq, qpitch, wm1, hm1, nbx, nby, ref_rows, mean_r, var_r);
            } else {
                c = subpatch_cost_fquad<kRecipIeee>(H, srcf, fpitch, wm1, hm1, nbx, nby, &lds.ref[k][0][owner], 64, mean_r, var_r);
            }
        }
        strong_cost += c;
        strong_count++;
    }
}

// APD.cu:505-527: the mean of the sub-patch costs, clamped, mixed 3 : 1 with the centre cost in double
__device__ __forceinline__ float deformed_combine(float center_cost, float strong_cost, int strong_count)
{
    if (strong_count == 0) {
        return center_cost;
    }
    strong_cost /= (float)strong_count;
    strong_cost = (strong_cost > 2.0f) ? 2.0f : strong_cost;
    return (float)(0.25 * (double)center_cost + 0.75 * (double)strong_cost);
}

template <bool kQuad, typename Ref>
__device__ __forceinline__ float ncc_deformed(const FrameArgs &fa, const ViewConst &vc, int v, const Ref &rp, const WeakLdsT<kQuad> &lds,
                                              int lane, int px, int py, const float4 pl, const SrcWindow &w)
{
    float qx, qy, qz;
    plane_q(pl, qx, qy, qz);
    const Homography H = make_homography(fa, vc, qx, qy, qz);
    float cx, cy;
    correspond(H, (float)px, (float)py, cx, cy);
    if (cx >= vc.wf || cx < 0.0f || cy >= vc.hf || cy < 0.0f) {
        return 2.0f;
    }
    const float center_cost = deformed_centre<kQuad>(fa, vc, rp, H, px, py, w);
    float strong_cost;
    int strong_count;
    deformed_strong<kQuad>(fa, vc, v, lds, lane, H, strong_cost, strong_count);
    return deformed_combine(center_cost, strong_cost, strong_count);
}

// WEAK pixels are sparse and clustered: the ones of each colour are compacted into a list and the update kernels run on
// full waves.  The list order decides what the caches see, so it is built deterministically (counts -> scan -> scatter,
// no atomics) in an order chosen for locality: 16 x 8 px tiles (one wave's worth of one colour), tiles raster-ordered
// inside supertiles of 16 x 16 tiles (256 x 128 px), supertiles in raster order.  Consecutive list chunks are therefore
// image neighbours in both directions, and an XCD that walks one contiguous eighth of the list (weak_chunk_of_block)
// keeps a compact 2-D region of every source image in its private L2 instead of a full-width strip shared with the
// seven other XCDs.  WEAK pixels only change in K4 and K14, so the two lists are built once per pass (apd_capi.hip).
constexpr int kSuperShift = APD_WEAK_SUPER_SHIFT, kSuperTiles = 1 << kSuperShift;  // supertile edge in tiles
constexpr int kListTileW = 16, kListTileH = 8;                  // 64 pixels of one colour
constexpr int kCountTilesPerBlock = 64;

struct TileOrder {
    int tiles_x, tiles_y, supers_x, total;  // total = number of ordered tile slots (supertiles are padded to 256 tiles)
};

static TileOrder tile_order(int W, int H)
{
    TileOrder o;
    o.tiles_x = (W + kListTileW - 1) / kListTileW;
    o.tiles_y = (H + kListTileH - 1) / kListTileH;
    o.supers_x = (o.tiles_x + kSuperTiles - 1) >> kSuperShift;
    const int supers_y = (o.tiles_y + kSuperTiles - 1) >> kSuperShift;
    o.total = o.supers_x * supers_y * kSuperTiles * kSuperTiles;
    return o;
}

// lane -> pixel of ordered tile slot s; false for padding slots and pixels outside the image / at or below row_limit (the
// image height for K3's full launch, half_rows for K9/K10: rows the reference's HALF launch never visits, APD.cu:2402)
__device__ __forceinline__ bool weak_pixel_of_lane(const FrameArgs &fa, const TileOrder o, int colour, int row_limit, int s, int lane, int &center)
{
    const int super = s >> (2 * kSuperShift), within = s & (kSuperTiles * kSuperTiles - 1);
    const int sy = super / o.supers_x, sx = super - sy * o.supers_x;
    const int tx = (sx << kSuperShift) + (within & (kSuperTiles - 1)), ty = (sy << kSuperShift) + (within >> kSuperShift);
    const int py = ty * kListTileH + (lane >> 3);
    const int px = tx * kListTileW + 2 * (lane & 7) + ((py + colour) & 1);
    center = py * fa.W + px;
    return tx < o.tiles_x && ty < o.tiles_y && px < fa.W && py < row_limit && fa.weak_info[center] == APD_WEAK;
}

// pass 1: WEAK pixels of `colour` per ordered tile -> exclusive offsets inside a block of 64 tiles + the block's total
__global__ __launch_bounds__(256) void k_weak_tile_counts(FrameArgs fa, TileOrder o, int colour, int row_limit, int *__restrict__ local_off,
                                                          int *__restrict__ block_total)
{
    __shared__ int cnt[kCountTilesPerBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = blockIdx.x * kCountTilesPerBlock;
    for (int t = 0; t < kCountTilesPerBlock / 4; ++t) {
        const int s = s0 + wave * (kCountTilesPerBlock / 4) + t;
        int center;
        const bool weak = s < o.total && weak_pixel_of_lane(fa, o, colour, row_limit, s, lane, center);
        const unsigned long long mask = __ballot(weak);
        if (lane == 0) {
            cnt[wave * (kCountTilesPerBlock / 4) + t] = __popcll(mask);
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int v = cnt[lane];
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) {
                incl += up;
            }
        }
        if (s0 + lane < o.total) {
            local_off[s0 + lane] = incl - v;
        }
        if (lane == 63) {
            block_total[blockIdx.x] = incl;
        }
    }
}

// pass 3 (pass 2 = k_weak_block_offsets over the block totals): write the pixel indices
__global__ __launch_bounds__(256) void k_weak_tile_scatter(FrameArgs fa, TileOrder o, int colour, int row_limit, const int *__restrict__ local_off,
                                                           const int *__restrict__ block_off, int *__restrict__ list)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = blockIdx.x * kCountTilesPerBlock;
    const int base0 = block_off[blockIdx.x];
    for (int t = 0; t < kCountTilesPerBlock / 4; ++t) {
        const int s = s0 + wave * (kCountTilesPerBlock / 4) + t;
        if (s >= o.total) {
            break;
        }
        int center;
        const bool weak = weak_pixel_of_lane(fa, o, colour, row_limit, s, lane, center);
        const unsigned long long mask = __ballot(weak);
        if (weak) {
            list[base0 + local_off[s] + __popcll(mask & ((1ull << lane) - 1ull))] = center;
        }
    }
}

// Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md): XCD x walks list chunks [x * per_xcd, (x + 1) * per_xcd).
__device__ __forceinline__ int weak_chunk_of_block(int b, int per_xcd)
{
    return (b & 7) * per_xcd + (b >> 3);
}

// Plane of reliable neighbour h of a WEAK pixel (slot h + 1 of its neighbour table): re-read where it is needed -- the planes
// of STRONG pixels do not change during a weak launch -- instead of eight float4 held in registers through the whole kernel.
__device__ __forceinline__ float4 candidate_plane(const FrameArgs &fa, const short2 *nb, int h)
{
    const short2 q = nb[h + 1];
    return fa.planes[q.x + q.y * fa.W];
}

// Rows of fetch positions of K9/K10's centre-patch window: a wave's pixels come from one or two 16 x 8 px tiles, the patch adds
// five rows either side, the rest is slack for hypotheses that move the patch along a slanted epipolar line.  With the packed
// sub-patch moments the workgroup's LDS stays below 20 KB, i.e. eight workgroups per CU: the kernel loses 17 % with seven
// (profiles/r03/ab_k910_occupancy.txt).
constexpr int kK910WinH = APD_K910_WIN_H;

// Window of view vc around where the wave's pixels land under their current planes.  Every lane of the wave calls this.
__device__ __forceinline__ SrcWindow weak_stage_window(const FrameArgs &fa, const ViewConst &vc, uint32_t *win, int px, int py, const float4 plane)
{
    float qx, qy, qz;
    plane_q(plane, qx, qy, qz);
    const Homography H = make_homography(fa, vc, qx, qy, qz);
    float cx, cy;
    correspond(H, (float)px, (float)py, cx, cy);
    const bool ok = cx >= 0.0f && cx < vc.wf && cy >= 0.0f && cy < vc.hf;  // false for NaN
    return stage_window_around<true, kK910WinH>(fa, vc, win, ok, cx, cy);
}

__device__ __forceinline__ SrcWindow no_window()
{
    SrcWindow none;
    none.valid = 0;
    none.wx0 = none.wy0 = none.addr0 = 0;
    none.lo_x = none.lo_y = 3.0e38f;
    none.hi_x = none.hi_y = -3.0e38f;
    return none;
}

// Hypotheses: 0..7 the eight reliable neighbours' planes, 8 the current plane, 9 the RANSAC fit
// plane, 10..14 the refinement set, 15 the final fixed-patch re-score.
template <int NMAX, bool kQuad>
__global__ __launch_bounds__(64, APD_K910_WAVES) void k910_update_weak(FrameArgs fa, int iter, const int *__restrict__ list, int count, int per_xcd)
{
    __shared__ WeakLdsT<kQuad> lds;
    // texel-quad mode: hypotheses 9..14 walk a compacted table of open (lane, hypothesis) pairs (see below)
    constexpr bool kCompact = kQuad && APD_K910_COMPACT_REFINE != 0;
    // Two phases, one LDS region.  Propagation (APD_K910_REMAP): prop_cost[h][pixel] -- the centre cost of (pixel, hypothesis h) on the way
    // in, its NCCNew cost on the way out -- and prop_live[pixel], the hypotheses whose sub-patches are to be scored.  Refinement
    // (kCompact): the table of open (lane, hypothesis) pairs and their costs.
    constexpr bool kRemap = APD_K910_REMAP != 0;
    constexpr int kPropWords = kRemap ? 8 * 64 + 64 : 0, kRefineWords = kCompact ? 5 * 64 + 5 * 64 / 2 : 0;
    __shared__ uint32_t phase_lds[(kPropWords > kRefineWords ? kPropWords : kRefineWords) > 0 ? (kPropWords > kRefineWords ? kPropWords : kRefineWords) : 1];
    float (*const prop_cost)[64] = reinterpret_cast<float (*)[64]>(phase_lds);
    uint32_t *const prop_live = phase_lds + 8 * 64;
    float (*const refine_cost)[64] = reinterpret_cast<float (*)[64]>(phase_lds);
    uint16_t *const refine_items = reinterpret_cast<uint16_t *>(phase_lds + 5 * 64);
    // the wave's window of the current source view for the centre patches (texel-quad mode)
    __shared__ uint32_t centre_window[kQuad ? window_dwords(true, kK910WinH) : 1];
    const int lane = threadIdx.x;
    const int first = weak_chunk_of_block(blockIdx.x, per_xcd) * 64;
    if (first >= count) {
        return;
    }
    // The window staging and the compacted stages are wave-wide operations: the lanes past the end of the list (last chunk only)
    // repeat its last pixel -- same inputs, same stores of the same values -- instead of leaving the wave.
    const int gid = min(first + lane, count - 1);
    const int W = fa.W;
    const int center = list[gid];
    const int py = center / W, px = center - py * W;
    const int nsrc = fa.num_src;
    const short2 *nb = &fa.neighbours[(size_t)fa.neighbours_map[center] * APD_NEIGHBOUR_NUM];
    weak_prepare_neighbours<kQuad>(fa, nb, lds, lane);
    RefPatch rp_regs;
    ref_patch_from_global(rp_regs, fa.ref_img, W, fa.H, px, py);
    // texel-quad mode: the 36 texels move to LDS as bytes; otherwise (float images) they stay in registers
    typename std::conditional<kQuad, RefPatchBytes, RefPatch>::type rp;
    if constexpr (kQuad) {
        rp = ref_patch_to_lds(rp_regs, lds, lane);
    } else {
        rp = rp_regs;
    }
    Rng rng = rng_load(fa.rng, center);

    float cost_array[9][NMAX];
    for (int h = 0; h < 9; ++h) {
        for (int v = 0; v < NMAX; ++v) {
            cost_array[h][v] = 0.0f;
        }
    }
    cost_array[0][0] = 2.0f;  // APD.cu:1345
    unsigned flags = 0;
    ViewWeights<NMAX> vw;
    vw.clear();
    float weight_norm = 0.0f;
    uint32_t sel = 0;
    float4 plane_now = fa.planes[center];
    float4 plane_final = plane_now;
    float depth_now = 0.0f, cost_now = 0.0f, cost_committed = 0.0f;
    float ref_depths[5];
    float4 ref_normals[5];
    bool skip_refine = false;

    // ---- candidates: the eight reliable neighbours' planes (must still be STRONG, :1354) + the current plane ----
    for (int h = 0; h < 8; ++h) {
        const short2 q = nb[h + 1];
        if (q.x == -1 || q.y == -1 || fa.weak_info[q.x + q.y * W] != APD_STRONG) {
            continue;
        }
        flags |= 1u << h;
    }
    // Costs of hypotheses 0..8.  View-major order: every hypothesis of a pixel projects a neighbour's sub-patch
    // to nearly the same place in one source image, so the nine evaluations per view reuse the lines the
    // first one brought in (the reference's hypothesis-major order cycles through all N images in between).
#pragma unroll 1
    for (int v = 0; v < nsrc; ++v) {
        const ViewConst &vc = view_const(fa, v);
        SrcWindow w = no_window();
        if constexpr (kQuad && APD_K910_WINDOW != 0) {
            w = weak_stage_window(fa, vc, centre_window, px, py, plane_now);
        }
        if constexpr (!kRemap) {
#pragma unroll 1
            for (int h = 0; h < 9; ++h) {
                if (h < 8 && !(flags & (1u << h))) {
                    continue;
                }
                float4 pl = plane_now;
                if (h < 8) {  // the neighbour's position is already in LDS (weak_prepare_neighbours): one global load instead of two dependent ones
                    const int packed = lds.nb[h][lane];
                    pl = fa.planes[(int)(short)(packed & 0xFFFF) + (packed >> 16) * W];
                }
                APD_WEAK_COUNT(0, 1);
                APD_WEAK_COUNT_WAVE(1);
                cost_array[h][v] = ncc_deformed<kQuad>(fa, vc, v, rp, lds, lane, px, py, pl, w);
            }
        } else {
            // Lane = pixel for the centre patches (their 36 samples lie around the pixel: the wave's window serves them) and for the own-plane
            // hypothesis; lane = (pixel, hypothesis) for the sub-patches of the eight neighbour hypotheses.  A sub-patch sits where its ANCHOR
            // is, a median of 27 px from the pixel in a direction of its own (tools/nb_cluster.py), so with lane = pixel the 64 lanes of a
            // sub-patch tap read 64 unrelated places: one L1 tag access per lane, the bound of this kernel through round 5 (46 of 49 ms).
            // With eight consecutive lanes on the eight hypotheses of ONE pixel and slot, the eight read the same anchor's neighbourhood
            // under eight nearly equal planes -- texels a few bytes apart, which the L1 serves with one tag access per aligned 16 bytes
            // (tools/tcp_patterns.hip) -- and a wave-level tap touches eight places instead of sixty-four.  Same operands, same operations,
            // the sub-patch costs of a (pixel, hypothesis) pair summed by one lane in slot order: same bits.
            unsigned live = 0;   // neighbour hypotheses whose centre projects into the view: their sub-patches are scored below
#pragma unroll 1
            for (int h = 0; h < 9; ++h) {
                if (h < 8 && !(flags & (1u << h))) {
                    continue;
                }
                float4 pl = plane_now;
                if (h < 8) {
                    const int packed = lds.nb[h][lane];
                    pl = fa.planes[(int)(short)(packed & 0xFFFF) + (packed >> 16) * W];
                }
                APD_WEAK_COUNT(0, 1);
                APD_WEAK_COUNT_WAVE(1);
                float qx, qy, qz;
                plane_q(pl, qx, qy, qz);
                const Homography H = make_homography(fa, vc, qx, qy, qz);
                float cx, cy;
                correspond(H, (float)px, (float)py, cx, cy);
                if (cx >= vc.wf || cx < 0.0f || cy >= vc.hf || cy < 0.0f) {
                    cost_array[h][v] = 2.0f;
                    continue;
                }
                const float center_cost = deformed_centre<kQuad>(fa, vc, rp, H, px, py, w);
                if (h == 8) {
                    float strong_cost;
                    int strong_count;
                    deformed_strong<kQuad>(fa, vc, v, lds, lane, H, strong_cost, strong_count);
                    cost_array[8][v] = deformed_combine(center_cost, strong_cost, strong_count);
                } else {
                    prop_cost[h][lane] = center_cost;
                    live |= 1u << h;
                }
            }
            prop_live[lane] = live;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int hyp = lane & 7;
#pragma unroll 1
            for (int g = 0; g < 8; ++g) {
                const int owner = g * 8 + (lane >> 3);
                const bool active = ((prop_live[owner] >> hyp) & 1u) != 0;
                if (active) {
                    const int packed = lds.nb[hyp][owner];
                    const float4 pl = fa.planes[(int)(short)(packed & 0xFFFF) + (packed >> 16) * W];
                    float qx, qy, qz;
                    plane_q(pl, qx, qy, qz);
                    const Homography H = make_homography(fa, vc, qx, qy, qz);
                    float strong_cost;
                    int strong_count;
                    deformed_strong<kQuad>(fa, vc, v, lds, owner, H, strong_cost, strong_count);
                    prop_cost[hyp][owner] = deformed_combine(prop_cost[hyp][owner], strong_cost, strong_count);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                if (live & (1u << h)) {
                    cost_array[h][v] = prop_cost[h][lane];
                }
            }
        }
        if constexpr (kQuad || kRemap) {  // the window / the cost table is rewritten for the next view
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

#pragma unroll 1
    for (int h = 9; h < 16; ++h) {
        if constexpr (kCompact) {
            if (h >= 10 && h <= 14) {
                continue;  // scored by the compacted stages at h == 9
            }
        }
        float4 pl;
        if (h == 9) {
            // ---- joint view selection (:1365-1434), adopt (:1436-1485) ----
            float priors[NMAX];
            for (int j = 0; j < NMAX; ++j) {
                priors[j] = 0.0f;
            }
            for (int i = 0; i < 8; ++i) {
                const short2 q = nb[i + 1];
                if (q.x == -1 || q.y == -1) {
                    continue;
                }
                const uint32_t sv = fa.selected_views[q.x + q.y * W];
                for (int j = 0; j < nsrc; ++j) {
                    priors[j] += bit_test(sv, (unsigned)j) == 1 ? 0.9f : 0.1f;
                }
            }
            select_views<NMAX>(fa, iter, cost_array, priors, rng, vw, sel, weight_norm);
            vw.store(fa, center);
            float final_costs[8];
            for (int i = 0; i < 8; ++i) {
                float f = 0.0f;
                for (int j = 0; j < nsrc; ++j) {
                    if (vw.get(j) > 0) {
                        if (fa.geom_consistency) {
                            if (flags & (1u << i)) {
                                f += (float)vw.get(j) * (cost_array[i][j] + fa.geom_factor * geom_cost(fa, view_const(fa, j), px, py, candidate_plane(fa, nb, i)));
                            } else {
                                f += (float)vw.get(j) * (cost_array[i][j] + fa.geom_factor * 3.0f);
                            }
                        } else {
                            f += (float)vw.get(j) * cost_array[i][j];
                        }
                    }
                }
                final_costs[i] = f / weight_norm;
            }
            int best = 0;
            float best_c = final_costs[0];
            for (int i = 1; i < 8; ++i) {
                if (final_costs[i] <= best_c) {
                    best_c = final_costs[i];
                    best = i;
                }
            }
            cost_now = 0.0f;
            for (int i = 0; i < nsrc; ++i) {
                if (fa.geom_consistency) {
                    cost_now += (float)vw.get(i) * (cost_array[8][i] + fa.geom_factor * geom_cost(fa, view_const(fa, i), px, py, plane_now));
                } else {
                    cost_now += (float)vw.get(i) * cost_array[8][i];
                }
            }
            cost_now /= weight_norm;
            cost_committed = cost_now;
            depth_now = depth_from_plane(fa, plane_now, px, py);
            if (flags & (1u << best)) {
                const float4 cand_best = candidate_plane(fa, nb, best);
                const float d = depth_from_plane(fa, cand_best, px, py);
                if (d >= fa.depth_min && d <= fa.depth_max && final_costs[best] < cost_now) {
                    depth_now = d;
                    plane_now = cand_best;
                    cost_now = final_costs[best];
                    fa.selected_views[center] = sel;
                }
            }
            // PlaneHypothesisRefinementWeak: fit plane first (:910-936)
            pl = fa.fit_planes[center];
            if (pl.x == 0 && pl.y == 0 && pl.z == 0) {
                skip_refine = true;  // returns before the random refinement too (:912-914)
            }
        } else if (h == 15) {
            // commit (:1488-1497), then re-score with the fixed patch (:1499-1507)
            if (fa.state == APD_REFINE_INIT) {
                if ((double)cost_now < (double)cost_committed - 0.1) {
                    plane_final = plane_now;
                }
            } else {
                plane_final = plane_now;
            }
            pl = plane_final;
        } else {
            if (h == 10 && !skip_refine) {
                make_refinement_set(fa, px, py, rng, plane_now, depth_now, ref_depths, ref_normals);
            }
            pl = ref_normals[h - 10];
            pl.w = distance_to_origin(fa, px, py, ref_depths[h - 10], pl.x, pl.y, pl.z);
        }
        if constexpr (kCompact) {
            if (h == 9) {
                // Hypotheses 9..14 in two stages -- the fit plane, then the five random refinements around whatever it left
                // (:910-980) -- each evaluated view by view over a compacted table: which (lane, hypothesis) pairs are still
                // open differs from lane to lane (unselected views, pixels without a fit plane, partial sums that have
                // already lost), and a wave runs an NCC whenever ANY lane needs it: 33.7 needed against 56.5 executed per
                // WEAK pixel (tools/weak_stats.py).  Every open pair of a view is entered into a per-wave table, the wave walks
                // it one entry per lane, and the lane that gets (owner, hypothesis) scores it from the owner's position with
                // the owner's neighbour data (LDS columns) and hypothesis (ds_bpermute) and leaves the cost in LDS; the owner
                // adds it to its sum in view order, as before.  All five refinements are bounded by the cost the stage
                // starts from (the hypothesis-major loop tightened the bound after every accepted one: a superset of the
                // NCCs, identical sums for every hypothesis that can still win, identical decisions).
                const unsigned long long live = __builtin_amdgcn_ballot_w64(true);
                const int wid = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
                const int nworkers = __builtin_popcountll(live);
                float4 hyp_n[5];
                float hyp_w[5], tcs[5];
#pragma unroll 1
                for (int stage = 0; stage < 2; ++stage) {
                    const int nh = stage == 0 ? 1 : 5;
                    if (stage == 0) {
                        hyp_n[0] = pl;
                        hyp_w[0] = pl.w;
                    } else if (!skip_refine) {
                        make_refinement_set(fa, px, py, rng, plane_now, depth_now, ref_depths, ref_normals);
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            hyp_n[k] = ref_normals[k];
                            hyp_w[k] = distance_to_origin(fa, px, py, ref_depths[k], ref_normals[k].x, ref_normals[k].y, ref_normals[k].z);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        tcs[k] = 0.0f;
                    }
                    const float lost = !(fa.geom_factor < 0.0f) ? refinement_lost_bound(fa, cost_now, weight_norm) : __builtin_inff();
#pragma unroll 1
                    for (int v = 0; v < nsrc; ++v) {
                        const uint32_t wv = skip_refine ? 0u : vw.get(v);
                        unsigned open = 0;
                        if (wv > 0) {
#pragma unroll
                            for (int k = 0; k < 5; ++k) {
                                open |= (k < nh && !(tcs[k] >= lost)) ? (1u << k) : 0u;
                            }
                        }
                        int off[6];
                        off[0] = 0;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const unsigned long long m = __builtin_amdgcn_ballot_w64((open >> k) & 1u);
                            if ((open >> k) & 1u) {
                                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                refine_items[off[k] + rank] = (uint16_t)((k << 6) | lane);
                            }
                            off[k + 1] = off[k] + __builtin_popcountll(m);
                        }
                        const int total = off[5];
                        if (total == 0) {
                            continue;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const ViewConst &vc = view_const(fa, v);
                        const SrcWindow w = APD_K910_WINDOW != 0 ? weak_stage_window(fa, vc, centre_window, px, py, plane_now) : no_window();
                        const int slot_step = nworkers, slot_end = total;
#pragma unroll 1
                        for (int first = 0; first < slot_end; first += slot_step) {
                            const int idx = first + wid;
                            const bool valid = idx < total;
                            const unsigned item = valid ? (unsigned)refine_items[idx] : (unsigned)lane;
                            const int owner = (int)(item & 63u), hyp = (int)(item >> 6);
                            const int last = min(total, first + nworkers) - 1;
                            int k_lo = 0, k_hi = 0;
#pragma unroll
                            for (int k = 1; k < 5; ++k) {
                                k_lo += off[k] <= first ? 1 : 0;
                                k_hi += off[k] <= last ? 1 : 0;
                            }
                            float4 hp = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
#pragma unroll 1
                            for (int k = k_lo; k <= k_hi; ++k) {
                                const float nx = __shfl(hyp_n[k].x, owner), ny = __shfl(hyp_n[k].y, owner), nz = __shfl(hyp_n[k].z, owner);
                                const float nw = __shfl(hyp_w[k], owner);
                                if (hyp == k) {
                                    hp = make_float4(nx, ny, nz, nw);
                                }
                            }
                            RefPatchBytes orp;
                            orp.base = &lds.centre[0][owner];
                            orp.mean = __shfl(rp.mean, owner);
                            orp.var = __shfl(rp.var, owner);
                            const int opx = __shfl(px, owner), opy = __shfl(py, owner);
                            if (valid) {
                                APD_WEAK_COUNT(2, 1);
                                APD_WEAK_COUNT_WAVE(3);
                                float c = ncc_deformed<kQuad>(fa, vc, v, orp, lds, owner, opx, opy, hp, w);
                                if (fa.geom_consistency) {
                                    c = c + fa.geom_factor * geom_cost(fa, vc, opx, opy, hp);
                                }
                                refine_cost[hyp][owner] = c;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            if ((open >> k) & 1u) {
                                tcs[k] += (float)wv * refine_cost[k][lane];
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (!skip_refine) {
#pragma unroll 1
                        for (int k = 0; k < nh; ++k) {  // accept tests in the reference's order (:932, :974)
                            float4 hp = hyp_n[k];
                            hp.w = hyp_w[k];
                            const float c = tcs[k] / weight_norm;
                            const float d = depth_from_plane(fa, hp, px, py);
                            if (d >= fa.depth_min && d <= fa.depth_max && c < cost_now) {
                                depth_now = d;
                                plane_now = hp;
                                cost_now = c;
                            }
                        }
                    }
                }
                continue;
            }
        }
        if (h <= 14 && skip_refine) {
            continue;
        }
        float tc = 0.0f;
        // hypotheses 9..14 are only compared with the running cost (:932, :974): a partial sum that has reached `lost`
        // cannot win any more and the remaining views are skipped (refinement_lost_bound, apd_sweep.h; the geometric term
        // is >= 0 for the non-negative geom_factor the bound needs)
        const float lost = (h <= 14 && !(fa.geom_factor < 0.0f)) ? refinement_lost_bound(fa, cost_now, weight_norm) : __builtin_inff();
#pragma unroll 1
        for (int v = 0; v < nsrc; ++v) {
            const ViewConst &vc = view_const(fa, v);
            if (vw.get(v) == 0) {
                // the cost of an unselected view is never used (:966-972); in the re-score it is multiplied by a zero
                // weight (:1503) and, being a finite value in [0, 2], adds exactly +0
                continue;
            }
            if (tc >= lost) {
                break;
            }
            if (h == 15) {
                float qx, qy, qz;
                plane_q(pl, qx, qy, qz);
                tc += (float)vw.get(v) * ncc_fixed<kQuad>(fa, vc, rp, px, py, qx, qy, qz);
            } else {
                APD_WEAK_COUNT(2, 1);
                APD_WEAK_COUNT_WAVE(3);
                const float c = ncc_deformed<kQuad>(fa, vc, v, rp, lds, lane, px, py, pl, no_window());
                {
                    if (fa.geom_consistency) {
                        tc += (float)vw.get(v) * (c + fa.geom_factor * geom_cost(fa, vc, px, py, pl));
                    } else {
                        tc += (float)vw.get(v) * c;
                    }
                }
            }
        }
        if (h <= 14) {
            tc /= weight_norm;
            const float d = depth_from_plane(fa, pl, px, py);
            if (d >= fa.depth_min && d <= fa.depth_max && tc < cost_now) {
                depth_now = d;
                plane_now = pl;
                cost_now = tc;
            }
        } else {
            fa.costs[center] = tc / weight_norm;
            fa.planes[center] = plane_final;
        }
    }
    rng_store(fa.rng, center, rng);
}

// ------------------------------------------------------------------------------------------------
// export of main.cpp:105-115 on the device (depth with out-of-range -> 0, normal)
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_export_depth_normal(FrameArgs fa, float *depth, float *normal)
{
    const int center = blockIdx.x * 256 + threadIdx.x;
    if (center >= fa.W * fa.H) {
        return;
    }
    const float4 pl = fa.planes[center];
    float d = pl.w;
    if (d < fa.depth_min || d > fa.depth_max) {
        d = 0.0f;
    }
    depth[center] = d;
    if (normal) {
        normal[3 * (size_t)center + 0] = pl.x;
        normal[3 * (size_t)center + 1] = pl.y;
        normal[3 * (size_t)center + 2] = pl.z;
    }
}

// ------------------------------------------------------------------------------------------------
// weak index map of APD.cpp:526-537 on the device: map[i] = number of WEAK pixels before pixel i in row-major order
// for a WEAK pixel, 0 for any other.  Three launches: per-block counts (4096 pixels per block, 16 consecutive pixels per
// lane), one block that turns the counts into offsets (+ the total), per-block exclusive scan + write.
// ------------------------------------------------------------------------------------------------

constexpr int kMapItems = 16, kMapBlock = 256, kMapChunk = kMapItems * kMapBlock;

__device__ __forceinline__ int weak_bits_of_lane(const uint8_t *__restrict__ weak, size_t n, size_t first)
{
    int bits = 0;
#pragma unroll
    for (int k = 0; k < kMapItems; ++k) {
        const size_t i = first + k;
        if (i < n && weak[i] == APD_WEAK) {
            bits |= 1 << k;
        }
    }
    return bits;
}

// exclusive prefix of `v` over the 256 lanes of the block (lane order); *total = block sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds4, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) {
            incl += o;
        }
    }
    if (lane == 63) {
        lds4[wave] = incl;
    }
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kMapBlock / 64; ++w) {
        const int c = lds4[w];
        before += (w < wave) ? c : 0;
        all += c;
    }
    *total = all;
    return before + incl - v;
}

__global__ __launch_bounds__(kMapBlock) void k_weak_block_counts(const uint8_t *__restrict__ weak, size_t n, int *__restrict__ counts)
{
    __shared__ int lds4[kMapBlock / 64];
    const size_t first = (size_t)blockIdx.x * kMapChunk + (size_t)threadIdx.x * kMapItems;
    int total;
    block_exclusive_scan(__popc(weak_bits_of_lane(weak, n, first)), lds4, &total);
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = total;
    }
}

// counts[0..nblocks) -> exclusive offsets in place, counts[nblocks] = total (one block, sequential over chunks of 256)
__global__ __launch_bounds__(kMapBlock) void k_weak_block_offsets(int *__restrict__ counts, int nblocks)
{
    __shared__ int lds4[kMapBlock / 64];
    int running = 0;
    for (int base = 0; base < nblocks; base += kMapBlock) {
        const int i = base + threadIdx.x;
        const int v = (i < nblocks) ? counts[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, lds4, &total);
        if (i < nblocks) {
            counts[i] = running + ex;
        }
        running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[nblocks] = running;
    }
}

__global__ __launch_bounds__(kMapBlock) void k_weak_index_map(const uint8_t *__restrict__ weak, size_t n, const int *__restrict__ offsets,
                                                              int *__restrict__ map)
{
    __shared__ int lds4[kMapBlock / 64];
    const size_t first = (size_t)blockIdx.x * kMapChunk + (size_t)threadIdx.x * kMapItems;
    const int bits = weak_bits_of_lane(weak, n, first);
    int total;
    int idx = offsets[blockIdx.x] + block_exclusive_scan(__popc(bits), lds4, &total);
#pragma unroll
    for (int k = 0; k < kMapItems; ++k) {
        const size_t i = first + k;
        if (i < n) {
            const bool w = (bits >> k) & 1;
            map[i] = w ? idx : 0;
            idx += w ? 1 : 0;
        }
    }
}

// scratch: at least ceil(n / 4096) + 1 ints; the total is left in scratch[ceil(n / 4096)]
hipError_t launch_weak_index_map(const uint8_t *weak, size_t n, int *map, int *scratch, hipStream_t s)
{
    const int nblocks = (int)((n + kMapChunk - 1) / kMapChunk);
    hipLaunchKernelGGL(k_weak_block_counts, dim3(nblocks), dim3(kMapBlock), 0, s, weak, n, scratch);
    hipLaunchKernelGGL(k_weak_block_offsets, dim3(1), dim3(kMapBlock), 0, s, scratch, nblocks);
    hipLaunchKernelGGL(k_weak_index_map, dim3(nblocks), dim3(kMapBlock), 0, s, weak, n, (const int *)scratch, map);
    return hipGetLastError();
}

hipError_t launch_export_depth_normal(const FrameArgs &fa, float *depth, float *normal, hipStream_t s)
{
    hipLaunchKernelGGL(k_export_depth_normal, dim3((fa.W * fa.H + 255) / 256), dim3(256), 0, s, fa, depth, normal);
    return hipGetLastError();
}

// scratch ints build_weak_lists needs for a W x H frame
size_t weak_list_scratch_ints(int W, int H)
{
    const TileOrder o = tile_order(W, H);
    return (size_t)o.total + (size_t)(o.total / kCountTilesPerBlock) + 1;
}

// The compacted WEAK pixels of both colours (list[0] black, list[1] red; each with room for every WEAK pixel) and their
// lengths, which come back to the host: the update kernels are launched with exact grids.  Synchronises the stream.
// all_rows: every row of the image (K3 is a full-frame launch in the reference); otherwise the rows its HALF launches reach.
hipError_t build_weak_lists(const FrameArgs &fa, bool all_rows, int *const list[2], int *scratch, int counts[2], hipStream_t s)
{
    const int row_limit = all_rows ? fa.H : (fa.half_rows < fa.H ? fa.half_rows : fa.H);
    const TileOrder o = tile_order(fa.W, fa.H);
    const int nblocks = o.total / kCountTilesPerBlock;  // o.total is a multiple of 256
    int *local_off = scratch, *block_off = scratch + o.total;
    for (int colour = 0; colour < 2; ++colour) {
        hipLaunchKernelGGL(k_weak_tile_counts, dim3(nblocks), dim3(256), 0, s, fa, o, colour, row_limit, local_off, block_off);
        hipLaunchKernelGGL(k_weak_block_offsets, dim3(1), dim3(kMapBlock), 0, s, block_off, nblocks);
        hipLaunchKernelGGL(k_weak_tile_scatter, dim3(nblocks), dim3(256), 0, s, fa, o, colour, row_limit, (const int *)local_off,
                           (const int *)block_off, list[colour]);
        hipError_t e = hipMemcpyAsync(&counts[colour], block_off + nblocks, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) {
            return e;
        }
        e = hipStreamSynchronize(s);  // the scratch is reused by the other colour
        if (e != hipSuccess) {
            return e;
        }
    }
    return hipGetLastError();
}

template <int NMAX>
static void launch_k910(const FrameArgs &fa, const int *list, int count, int iter, hipStream_t s)
{
    if (count <= 0) {
        return;
    }
    const int chunks = (count + 63) / 64, per_xcd = (chunks + 7) / 8;
    if (fa.use_quads) {
        hipLaunchKernelGGL((k910_update_weak<NMAX, true>), dim3(8 * per_xcd), dim3(64), 0, s, fa, iter, list, count, per_xcd);
    } else {
        hipLaunchKernelGGL((k910_update_weak<NMAX, false>), dim3(8 * per_xcd), dim3(64), 0, s, fa, iter, list, count, per_xcd);
    }
}

hipError_t launch_weak_kernel(const FrameArgs &fa, int kernel_id, int iter, hipStream_t s, const int *const weak_list[2], const int weak_count[2])
{
    const int n = fa.W * fa.H;
    switch (kernel_id) {
    case APD_K2_FIND_NEAREST_STRONG: {
        const dim3 grid((fa.W + 63) / 64, (fa.H + 3) / 4);
        hipLaunchKernelGGL(k2a_column_nearest, grid, dim3(256), 0, s, fa);
        hipLaunchKernelGGL(k2b_row_search, grid, dim3(256), 0, s, fa);
        break;
    }
    case APD_K3_GEN_NEIGHBOURS:
        for (int colour = 0; colour < 2 && weak_list; ++colour) {  // the lists only split the pixels by colour; K3 has no colour
            if (weak_list[colour] && weak_count[colour] > 0) {
                const dim3 grid((weak_count[colour] + 63) / 64);
                const bool cut = fa.k3_cut_valid != 0, no_jitter = fa.k3_shift_range == 1;
                if (cut && no_jitter) {
                    hipLaunchKernelGGL((k3_gen_neighbours<true, true>), grid, dim3(64), 0, s, fa, weak_list[colour], weak_count[colour]);
                } else if (cut) {
                    hipLaunchKernelGGL((k3_gen_neighbours<true, false>), grid, dim3(64), 0, s, fa, weak_list[colour], weak_count[colour]);
                } else if (no_jitter) {
                    hipLaunchKernelGGL((k3_gen_neighbours<false, true>), grid, dim3(64), 0, s, fa, weak_list[colour], weak_count[colour]);
                } else {
                    hipLaunchKernelGGL((k3_gen_neighbours<false, false>), grid, dim3(64), 0, s, fa, weak_list[colour], weak_count[colour]);
                }
            }
        }
        break;
    case APD_K4_NEIGHBOUR_UPDATE:
        hipLaunchKernelGGL(k4_neighbour_update, dim3((n + 255) / 256), dim3(256), 0, s, fa);
        break;
    case APD_K8_RANSAC_FIT_PLANE:
        hipLaunchKernelGGL(k8_ransac_fit_plane, dim3((n + 255) / 256), dim3(256), 0, s, fa);
        break;
    case APD_K9_BLACK_UPDATE_WEAK:
    case APD_K10_RED_UPDATE_WEAK: {
        const int colour = (kernel_id == APD_K9_BLACK_UPDATE_WEAK) ? 0 : 1;
        if (!weak_list || !weak_list[colour]) {
            break;  // no prior state was uploaded: every pixel is STRONG (APD.cpp:541-547)
        }
        if (fa.num_src <= 8) {
            launch_k910<8>(fa, weak_list[colour], weak_count[colour], iter, s);
        } else if (fa.num_src <= 12) {
            launch_k910<12>(fa, weak_list[colour], weak_count[colour], iter, s);
        } else if (fa.num_src <= 16) {
            launch_k910<16>(fa, weak_list[colour], weak_count[colour], iter, s);
        } else {
            launch_k910<32>(fa, weak_list[colour], weak_count[colour], iter, s);
        }
        break;
    }
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace apd


APD_LAB_WEAK_STATS_ACCESSOR
