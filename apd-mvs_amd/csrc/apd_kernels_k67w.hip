// apd_kernels_k67w.hip -- K6/K7 (Black/RedPixelUpdateStrong, APD.cu:982-1321, 837-890) with the source-image
// gathers served from LDS.
//
// Why: a wave-level gather from the row-major texel-quad image costs the L1 (TCP) about 30 tag accesses, one per
// cycle; 4032 gathers per pixel and iteration keep the TCP of a CU busy for as long as the whole launch takes
// (profiles/r01/tuning/pmc_k67_*: TCP_TOTAL_CACHE_ACCESSES / 256 CUs == launch cycles).  LDS serves the same 64-lane
// gather in two to four cycles.  After the first iteration almost every hypothesis of a pixel (its neighbours' planes,
// its own plane, the small perturbations) projects the patch to within a few texels of where the current plane puts
// it, so each wave copies, once per source view and phase, the (kWinW x kWinH)-entry window of the quad image around
// the projection of its 32x4 footprint into a private LDS region, and every NCC whose 36 samples provably fall inside
// that window reads LDS.  Everything else (random hypotheses, first iteration, occlusion edges) takes the global path
// of apd_device.h.  Both paths read the same entries and do the same arithmetic in the same order: bit-identical.
//
// To stage one window per (view, phase) instead of one per (hypothesis, view) the evaluation order is view-major
// (the reference's is hypothesis-major, APD.cu:1203 / :869): per-view costs do not depend on one another, the
// weighted sums over views are still accumulated in view order, and the five refinement hypotheses are all fixed
// before the first of them is tested (:855-867), so the order of evaluation is not observable.
#include "apd_device.h"
#include "apd_sweep.h"
#include "apd_window.h"


namespace apd {

constexpr int kWinH = APD_WIN_H;

// Places the window around the projections of the live pixels' centres under their current planes and stages it.
// Every lane of the wave calls this (no divergence).
// `trusted`: the pixel's plane already explains the images (low cost).  While the planes are still converging the
// bounding box of all projections is useless (one wild plane moves its centre anywhere), so the box is taken over the
// trusted pixels when there are any.
template <bool kQuad>
__device__ __forceinline__ SrcWindow stage_window(const FrameArgs &fa, const ViewConst &vc, uint32_t *win, bool alive, int px, int py,
                                                  const float4 plane, bool trusted, bool enabled)
{
    if (!enabled) {  // wave-uniform: the planes are still random (first iteration of a FIRST_INIT pass), nothing to stage
        SrcWindow none;
        none.valid = 0;
        none.wx0 = none.wy0 = none.addr0 = 0;
        none.lo_x = none.lo_y = 3.0e38f;
        none.hi_x = none.hi_y = -3.0e38f;
        return none;
    }
    float cx = 0.0f, cy = 0.0f;
    bool ok = false;
    if (alive) {
        float qx, qy, qz;
        plane_q(plane, qx, qy, qz);
        const Homography H = make_homography(fa, vc, qx, qy, qz);
        correspond(H, (float)px, (float)py, cx, cy);
        ok = cx >= 0.0f && cx < vc.wf && cy >= 0.0f && cy < vc.hf;  // false for NaN
    }
    if (__builtin_amdgcn_ballot_w64(ok && trusted) != 0) {
        ok = ok && trusted;
    }
    return stage_window_around<kQuad, kWinH>(fa, vc, win, ok, cx, cy);
}

// Cheapest candidate of propagation arm `arm` (order of APD.cu:1020: near/far x up,down,left,right).
__device__ __forceinline__ bool arm_pos(const FrameArgs &fa, int px, int py, int arm, int &pos)
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

constexpr float kTrustedCost = APD_WIN_TRUST;

// kTiled: NCCs that miss the window (all of them while the windows are off) gather from the tiled copy of the quad image
// kApprox: tolerance mode APD_OPT_FAST_RCP (bare v_rcp_f32 in the sample loops; not bit-identical to the oracle)
template <int NMAX, bool kQuad, bool kTiled, bool kApprox>
__global__ __launch_bounds__(256, kQuad ? APD_K67W_WAVES : APD_K67W_WAVES_F32) void k67w_update_strong(FrameArgs fa, int colour, int iter)
{
    __shared__ float tile[kLdsH * kLdsPitch];
    __shared__ uint32_t windows[4][window_dwords(kQuad, kWinH)];
    __shared__ uint16_t refine_items[4][5 * 64];  // per wave: (hypothesis << 6) | owner lane of every open (lane, hypothesis)
    __shared__ float refine_cost[4][5][64];       // per wave: the cost a worker lane computed for (hypothesis, owner lane)
    const TilePixel t = checkerboard_pixel(fa, colour);
    // stage the reference tile + 5 px halo (clamp-to-edge, as the texture unit would)
    for (int idx = threadIdx.x; idx < kLdsW * kLdsH; idx += 256) {
        const int r = idx / kLdsW, c = idx - r * kLdsW;
        tile[r * kLdsPitch + c] = fetch_texel(fa.ref_img, fa.W, fa.H, t.tx0 + c - kHalo, t.ty0 + r - kHalo);
    }
    __syncthreads();
    uint32_t *win = windows[threadIdx.x >> 6];
    const int px = t.px, py = t.py;
    const int center = py * fa.W + px;
    // lanes without a pixel to update stay in the wave: every lane helps to stage the windows
    const bool alive = checkerboard_active(fa, t) && fa.weak_info[center] != APD_WEAK;
    if (__builtin_amdgcn_ballot_w64(alive) == 0) {
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
    Rng rng = Rng{0, 0, 0, 0, 0, 0};
    // [8] = current plane.  The reference's "= { 2.0f }" (APD.cu:1004) sets element [0][0] to 2 and the rest to 0, and an arm that
    // falls outside the image keeps those values: the loop below writes them where it skips the NCC instead of clearing the
    // whole table first -- it lives in scratch memory, and the 9 x NMAX clearing stores per pixel were a third of the launch's
    // write-back (3.7 of 11.3 GB on configs[1]); columns >= nsrc are never read.
    float cost_array[9][NMAX];
    int positions[8];
    unsigned flags = 0;
    ViewWeights<NMAX> vw;
    vw.clear();
    float weight_norm = 0.0f;
    uint32_t sel = 0;
    float4 plane_now = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
    float depth_now = 0.0f, cost_now = 0.0f, cost_committed = 0.0f;
    float ref_w[5];  // plane distances of the five refinement hypotheses
    float4 ref_normals[5];
    float tc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

    // After K5's random initialisation nearly every hypothesis of the first iteration lands somewhere else in the
    // source image (10 % of the NCCs could read a window, 60 % of the waves are mixed): no windows in that iteration.
    const bool use_windows = !(fa.state == APD_FIRST_INIT && iter < APD_WIN_FROM_ITER);
    bool trusted = false;  // window placement only: the plane from the previous update has a low cost
    if (alive) {
        plane_now = fa.planes[center];
        trusted = fa.costs[center] < kTrustedCost;
#pragma unroll 1
        for (int h = 0; h < 8; ++h) {
            int pos = 0;
            if (arm_pos(fa, px, py, h, pos)) {
                flags |= 1u << h;
            }
            positions[h] = pos;
        }
    }

    // ---- costs of the eight propagation candidates and of the current plane, view by view (:1203-1209) ----
#pragma unroll 1
    for (int v = 0; v < nsrc; ++v) {
        const ViewConst &vc = view_const(fa, v);
        const SrcWindow w = stage_window<kQuad>(fa, vc, win, alive, px, py, plane_now, trusted, use_windows);
        if (alive) {
            // The candidate planes are re-read per view: eight float4 do not fit the register budget.  Hiding that read was
            // tried twice in round 2 and is not worth it: loading candidate h + 1 into registers before candidate h is scored
            // costs more in spills than the wait it hides (59.6 -> 61.8 ms per iteration on configs[1]), and sending it to a
            // per-wave LDS slot with global_load_lds_dwordx4 changes nothing (59.5 against 60.0).
#pragma unroll 1
            for (int h = 0; h < 9; ++h) {
                if (h < 8 && !(flags & (1u << h))) {
                    cost_array[h][v] = (h == 0 && v == 0) ? 2.0f : 0.0f;
                    continue;
                }
                const float4 pl = (h < 8) ? fa.planes[positions[h]] : plane_now;
                float qx, qy, qz;
                plane_q(pl, qx, qy, qz);
                cost_array[h][v] = ncc_fixed_windowed<kQuad, kWinW, kTiled, kApprox>(fa, vc, w, rp, px, py, qx, qy, qz);
            }
        }
    }

    if (alive) {
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
        // the random stream is only drawn from between here and make_refinement_set: six registers less through both NCC phases
        rng = rng_load(fa.rng, center);
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
        float ref_depths[5];
        trusted = cost_now < kTrustedCost;
        make_refinement_set(fa, px, py, rng, plane_now, depth_now, ref_depths, ref_normals);
        rng_store(fa.rng, center, rng);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            ref_w[k] = distance_to_origin(fa, px, py, ref_depths[k], ref_normals[k].x, ref_normals[k].y, ref_normals[k].z);
        }
    }

    // ---- PlaneHypothesisRefinementStrong (:837-890): the five hypotheses against every selected view ----
    // A hypothesis is only ever compared with the running cost (`temp_cost < *cost`, :884), which never exceeds the cost
    // the refinement starts from, and its weighted sum grows monotonically view by view (weights > 0, costs in [0, 2],
    // round-to-nearest addition and division are monotone).  Once the partial sum reaches `lost` -- a float just above
    // cost_now * weight_norm -- the quotient can no longer be below cost_now: the remaining views of that hypothesis are
    // skipped and the accept test below rejects it exactly as it would reject the full sum.  In converged iterations
    // that removes most NCCs of the two random-depth hypotheses, which are also the ones that miss the windows.
    const float lost = refinement_lost_bound(fa, cost_now, weight_norm);
    // Which (lane, hypothesis) pairs are still open differs from lane to lane -- a view a pixel did not select, a hypothesis
    // that has already lost -- and a wave runs an NCC whenever ANY of its lanes needs it: in converged iterations the lanes
    // need 11 refinement NCCs per pixel and the waves execute 25 (tools/win_stats.py).  So the open pairs of a view are
    // compacted: every lane enters its pairs into a per-wave table (hypothesis-major, ranks from ballots), the wave walks
    // the table 64 entries at a time, and the lane that gets entry (owner, hypothesis) fetches the owner's hypothesis and
    // reference moments through ds_bpermute, scores it from the owner's pixel position (same window, same reference tile:
    // both belong to the wave) and leaves the cost in LDS for the owner, who adds it to its running sum in view order
    // exactly as before.  Same NCCs on the same operands: same bits; ceil(pairs / 64) wave-level NCCs per view instead
    // of one per hypothesis anybody still has open.
    const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
    uint16_t *items = refine_items[wave_id];
    float(*costs_out)[64] = refine_cost[wave_id];
    const int wave_lx0 = (wave_id % kWavesX) * kWaveW, wave_ly0 = (wave_id / kWavesX) * kWaveH;
#pragma unroll 1
    for (int v = 0; v < nsrc; ++v) {
        const uint32_t wv = alive ? vw.get(v) : 0u;
        unsigned open = 0;  // hypotheses of this lane that can still win
        if (wv > 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                open |= (tc[k] >= lost) ? 0u : (1u << k);
            }
        }
        // table order: the hypotheses that keep the pixel's depth first (random normal, perturbed normal, perturbed depth: their
        // patches land where the window is), the two random-depth ones last -- a slot takes the global path as soon as one of its
        // entries misses the window, so the misses are collected in the last, partly filled slot
        constexpr int kOrder[5] = {1, 3, 4, 0, 2};
        int off[6];
        off[0] = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = kOrder[j];
            const unsigned long long m = __builtin_amdgcn_ballot_w64((open >> k) & 1u);
            if ((open >> k) & 1u) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                items[off[j] + rank] = (uint16_t)((k << 6) | lane);
            }
            off[j + 1] = off[j] + __builtin_popcountll(m);
        }
        const int total = off[5];
        if (total == 0) {
            continue;  // nobody in the wave has anything left to score in this view
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the table is read by other lanes of the wave
        __builtin_amdgcn_wave_barrier();
        const ViewConst &vc = view_const(fa, v);
        const SrcWindow w = stage_window<kQuad>(fa, vc, win, alive, px, py, plane_now, trusted, use_windows);
#pragma unroll 1
        for (int first = 0; first < total; first += 64) {
            const int idx = first + lane;
            const bool valid = idx < total;
            const unsigned item = valid ? (unsigned)items[idx] : 0u;
            const int owner = (int)(item & 63u), hyp = (int)(item >> 6);
            // the entries of one slot span few hypotheses (the table is hypothesis-major): fetch only those
            const int last = min(total, first + 64) - 1;
            int j_lo = 0, j_hi = 0;
#pragma unroll
            for (int j = 1; j < 5; ++j) {
                j_lo += off[j] <= first ? 1 : 0;
                j_hi += off[j] <= last ? 1 : 0;
            }
            float4 pl = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
#pragma unroll 1
            for (int j = j_lo; j <= j_hi; ++j) {
                const int k = (0x20431 >> (4 * j)) & 7;  // kOrder[j]
                const float nx = __shfl(ref_normals[k].x, owner), ny = __shfl(ref_normals[k].y, owner), nz = __shfl(ref_normals[k].z, owner);
                const float nw = __shfl(ref_w[k], owner);
                if (hyp == k) {
                    pl = make_float4(nx, ny, nz, nw);
                }
            }
            RefPatchLds<kLdsPitch> orp;
            orp.mean = __shfl(rp.mean, owner);
            orp.var = __shfl(rp.var, owner);
            const int oly = wave_ly0 + owner / kWaveLanesX;
            const int olx = wave_lx0 + 2 * (owner % kWaveLanesX) + ((oly + colour) & 1);
            orp.base = &tile[oly * kLdsPitch + olx];
            if (valid) {
                float qx, qy, qz;
                plane_q(pl, qx, qy, qz);
                costs_out[hyp][owner] = ncc_fixed_windowed<kQuad, kWinW, kTiled, kApprox>(fa, vc, w, orp, t.tx0 + olx, t.ty0 + oly, qx, qy, qz);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if ((open >> k) & 1u) {
                tc[k] += (float)wv * costs_out[k][lane];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the table and the costs are rewritten for the next view
        __builtin_amdgcn_wave_barrier();
    }

    if (!alive) {
        return;
    }
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {  // accept tests in the reference's order (:881-888)
        float4 pl = ref_normals[k];
        pl.w = ref_w[k];
        const float c = tc[k] / weight_norm;
        const float d = depth_from_plane(fa, pl, px, py);
        if (d >= fa.depth_min && d <= fa.depth_max && c < cost_now) {
            depth_now = d;
            plane_now = pl;
            cost_now = c;
        }
    }
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


template <bool kQuad, bool kTiled, bool kApprox>
static void launch_k67w_n(const FrameArgs &fa, int tiles, int colour, int iter, hipStream_t s)
{
    if (fa.num_src <= 8) {
        hipLaunchKernelGGL((k67w_update_strong<8, kQuad, kTiled, kApprox>), dim3(tiles), dim3(256), 0, s, fa, colour, iter);
    } else if (fa.num_src <= 12) {  // ten sources is the common MVS count: do not pay scratch for sixteen columns
        hipLaunchKernelGGL((k67w_update_strong<12, kQuad, kTiled, kApprox>), dim3(tiles), dim3(256), 0, s, fa, colour, iter);
    } else if (fa.num_src <= 16) {
        hipLaunchKernelGGL((k67w_update_strong<16, kQuad, kTiled, kApprox>), dim3(tiles), dim3(256), 0, s, fa, colour, iter);
    } else {
        hipLaunchKernelGGL((k67w_update_strong<32, kQuad, kTiled, kApprox>), dim3(tiles), dim3(256), 0, s, fa, colour, iter);
    }
}

template <bool kQuad, bool kTiled>
static void launch_k67w(const FrameArgs &fa, int tiles, int colour, int iter, hipStream_t s)
{
    if (fa.approx_rcp) {
        launch_k67w_n<kQuad, kTiled, true>(fa, tiles, colour, iter, s);
    } else {
        launch_k67w_n<kQuad, kTiled, false>(fa, tiles, colour, iter, s);
    }
}

// Which launches gather from the tiled copy (APD_OPT_TILED_COPY; same results either way):
//   0 never, 1 (default) while the windows are off, i.e. the random first iteration of a FIRST_INIT pass, 2 always.
// Measured on configs[1] (profiles/r02/tiled_vs_rowmajor.txt): the first black launch 129 -> 94 ms; from the second
// iteration on the row-major copy is faster (three extra address instructions per sample on a VALU-bound kernel).
hipError_t launch_k67_windowed(const FrameArgs &fa, int colour, int iter, hipStream_t s)
{
    const int tiles = ((fa.W + kTileW - 1) / kTileW) * ((fa.H + kTileH - 1) / kTileH);
    if (fa.use_quads) {
        const int mode = fa.tiled_mode;
        const bool windows_off = fa.state == APD_FIRST_INIT && iter < APD_WIN_FROM_ITER;
        if (fa.have_tiled && (mode == 2 || (mode == 1 && windows_off))) {
            launch_k67w<true, true>(fa, tiles, colour, iter, s);
        } else {
            launch_k67w<true, false>(fa, tiles, colour, iter, s);
        }
    } else {
        launch_k67w<false, false>(fa, tiles, colour, iter, s);
    }
    return hipGetLastError();
}

}  // namespace apd


APD_WIN_STATS_ACCESSOR(apd_debug_win_stats)
