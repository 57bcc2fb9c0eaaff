// apd_kernels_k1415w.hip -- K14 DepthToWeak (APD.cu:1990-2144) and K15 LocalRefine (:2146-2232) with the source-image
// gathers served from per-wave LDS windows (apd_window.h).
//
// Both kernels score depth samples along the pixel's ray, one pixel of disparity apart (61 for K14, 12 for K15),
// against every selected view: under one view the patch slides along the epipolar line by about b_view / b_mean texels
// per step.  The evaluation is therefore view-major (the reference's is sample-major, :2056 / :2196): per view, the
// samples are walked in chunks, the wave stages one window around the projections of its 8x8 pixels at the middle
// sample of the chunk, and every NCC whose 36 samples provably fall inside reads LDS with the four-instruction lerp;
// anything else takes the global path.  The per-sample costs are accumulated over the views in view order, exactly as
// the sample-major loop does, so the result is bit-identical.
#include "apd_device.h"
#include "apd_sweep.h"
#include "apd_window.h"


namespace apd {

constexpr int kFwTile = 16;                                // workgroup tile: 16x16 pixels, wave64 = 8x8
constexpr int kFwLds = kFwTile + 2 * kPatchRadius;         // 26
constexpr int kFwPitch = APD_FW_TILE_PITCH;                // 27
constexpr int kFwWinPitch = APD_K1415_WIN_PITCH;
static_assert(kFwWinPitch >= kWinW && kFwWinPitch <= 127, "window pitch: at least the wave width; two-address LDS reads need offset1 < 256 dwords");
template <bool kQuad> constexpr int k14_win_h() { return kQuad ? APD_K14_WIN_H : APD_K14_WIN_H_F32; }

__device__ __forceinline__ void fw_pixel(int &px, int &py)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    px = blockIdx.x * kFwTile + (wave & 1) * 8 + (lane & 7);
    py = blockIdx.y * kFwTile + (wave >> 1) * 8 + (lane >> 3);
}

// Stages the workgroup's reference tile + 5 px halo (clamp-to-edge); the 36 texels of a lane's patch stay in LDS.
__device__ __forceinline__ RefPatchLds<kFwPitch> fw_stage_ref(const FrameArgs &fa, float *tile, int px, int py)
{
    const int x0 = blockIdx.x * kFwTile - kPatchRadius, y0 = blockIdx.y * kFwTile - kPatchRadius;
    for (int idx = threadIdx.x; idx < kFwLds * kFwLds; idx += 256) {
        const int r = idx / kFwLds, c = idx - r * kFwLds;
        tile[r * kFwPitch + c] = fetch_texel(fa.ref_img, fa.W, fa.H, x0 + c, y0 + r);
    }
    __syncthreads();
    RefPatchLds<kFwPitch> rp;
    rp.base = &tile[(py - y0 - kPatchRadius) * kFwPitch + (px - x0 - kPatchRadius)];
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

// baseline + weight sum over the selected views (:2036-2044); no image access
__device__ __forceinline__ int fw_baseline_and_weight(const FrameArgs &fa, uint32_t sel, const ViewWeights<32> &vw, float &base_line,
                                                      float &weight_normal)
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

// Window of view vc around where the pixels of the wave land when their planes are (origin normal, distance w).
template <bool kQuad>
__device__ __forceinline__ SrcWindow fw_stage(const FrameArgs &fa, const ViewConst &vc, uint32_t *win, bool use, int px, int py,
                                              const float4 origin, float w)
{
    float cx = 0.0f, cy = 0.0f;
    bool ok = false;
    if (use) {
        float4 pl = origin;
        pl.w = w;
        float qx, qy, qz;
        plane_q(pl, qx, qy, qz);
        const Homography H = make_homography(fa, vc, qx, qy, qz);
        correspond(H, (float)px, (float)py, cx, cy);
        ok = cx >= 0.0f && cx < vc.wf && cy >= 0.0f && cy < vc.hf;  // false for NaN
    }
    return stage_window_around<kQuad, k14_win_h<kQuad>(), kFwWinPitch>(fa, vc, win, ok, cx, cy);
}

// ------------------------------------------------------------------------------------------------
// K14
// ------------------------------------------------------------------------------------------------

// Loop order: chunk of depth samples outermost, views inside.  The window count is the same in either order (one per view and
// chunk); with the views innermost the accumulated costs of the chunk's samples live in registers (a vector indexed by the
// wave-uniform sample counter: v_movrel, no scratch) and every finished cost -- divided by the weight sum and clamped, :2093 -- is
// written ONCE to the per-lane profile the peak search reads.  Rounds 2-4 ran the views outermost and read-modified-wrote the
// 61-entry profile in scratch memory once per view: 158 / 218 GB of write-back per launch at 6200x4130 with 10 sources for 25.6 MB
// of output (profiles/r04/pmc_pass_*).  Every sample still adds its views in view order: same bits.
//
// kPairs: the kernel may walk the (sample, lane) pairs of a chunk instead of its samples (see the view loop).  Carrying that
// second loop costs the sample loop registers (4096x3072, 8 sources: 119.6 -> 124.0 ms; float images 34.5 -> 40.1 at 2048x1536),
// so the launcher picks it where views are selected sparsely enough for it to pay: from ten sources on (15 draws over N views;
// K14 ms at 2048x1536, photometric / geometric pass: N = 10 38.0 / 47.4 -> 38.4 / 44.3, N = 12 46.3 / 58.3 -> 44.3 / 51.2,
// N = 16 61.6 / 76.6 -> 51.9 / 59.8).
static_assert(APD_K14_CHUNK == 4 || APD_K14_CHUNK == 8 || APD_K14_CHUNK == 16, "the chunk's accumulators are a register vector");
typedef float k14_chunk_t __attribute__((ext_vector_type(APD_K14_CHUNK)));

template <bool kQuad, bool kPairs>
__global__ __launch_bounds__(256, kQuad ? APD_K14W_WAVES : APD_K1415W_WAVES_F32) void k14w_depth_to_weak(FrameArgs fa)
{
    __shared__ float tile[kFwLds * kFwPitch];
    __shared__ uint32_t windows[4][window_dwords(kQuad, k14_win_h<kQuad>(), kFwWinPitch)];
    int px, py;
    fw_pixel(px, py);
    const RefPatchLds<kFwPitch> rp = fw_stage_ref(fa, tile, px, py);
    uint32_t *win = windows[threadIdx.x >> 6];
    const int wave_id = threadIdx.x >> 6;
    const int W = fa.W, H = fa.H;
    const int min_margin = 6;
    const int center = px + py * W;
    constexpr int RADIUS = 30, NP = 2 * RADIUS + 1;
    constexpr int NP_PAD = ((NP + APD_K14_CHUNK - 1) / APD_K14_CHUNK) * APD_K14_CHUNK;

    // lanes without depth samples to score stay in the wave: every lane helps to stage the windows
    bool alive = px < W && py < H;
    float4 origin = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
    uint32_t sel = 0;
    ViewWeights<32> vw;
    vw.clear();
    float weight_normal = 1.0f;
    float base_line = 0.0f, disp = 0.0f;
    float pc[NP_PAD];   // finished cost of depth sample i: MIN(2, sum / weight_normal), 2 outside [depth_min, depth_max]; written once
    uint64_t in_range = 0;
    if (alive) {
        if (px < min_margin || py < min_margin || px >= W - min_margin || py >= H - min_margin) {
            fa.weak_info[center] = APD_UNKNOWN;
            alive = false;
        }
    }
    if (alive) {
        origin = normal_world_to_cam(fa, fa.planes[center]);
        if (origin.w == 0) {
            fa.weak_info[center] = APD_UNKNOWN;
            alive = false;
        }
    }
    if (alive) {
        sel = fa.selected_views[center];
        vw.load(fa, center);
        const int valid = fw_baseline_and_weight(fa, sel, vw, base_line, weight_normal);
        if (valid == 0) {
            fa.weak_info[center] = APD_UNKNOWN;
            alive = false;
        } else {
            // cost_now of :2022-2051 is computed by the reference but never used by K14's classification
            base_line /= (float)valid;
            disp = fa.K[0] * base_line / origin.w;
            // Samples 0 and NP - 1 are never read: the peak search covers i = 2 .. NP - 3 and looks at i - 1 and i + 1 (:2104-2115),
            // and pc[min_peak] of :2120 can only be pc[0] when no peak was found, where abs(0 - 30) > weak_peak_radius has already
            // decided (for every radius below 30; with a larger one the two samples are scored like the rest).  Two of 61 NCC
            // rounds per selected view less, no state bit changes.
            const bool skip_ends = fa.early_out && fa.weak_peak_radius < RADIUS;
#pragma unroll 1
            for (int i = 0; i < NP; ++i) {
                const float p_depth = fa.K[0] * base_line / (disp + (float)(i - RADIUS));
                if (skip_ends && (i == 0 || i == NP - 1)) {
                    continue;
                }
                if (!(p_depth < fa.depth_min || p_depth > fa.depth_max)) {
                    in_range |= 1ull << i;
                }
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(alive && in_range != 0) == 0) {
        if (alive) {  // every sample out of range: all costs are 2, no minimum
            fa.weak_info[center] = APD_WEAK;  // abs(0 - 30) > weak_peak_radius or pc[0] = 2 > 0.5 (:2120)
        }
        return;
    }

    // Two phases.  The classification below is WEAK whenever the lowest local minimum lies more than weak_peak_radius
    // samples from the centre or costs more than 0.5 (:2120).  So a pixel without a local minimum of at most 0.5 WITHIN the
    // radius is WEAK whatever the other samples cost: either its lowest minimum lies outside the radius (or there is none:
    // min_peak = 0), or it lies inside and costs more than 0.5.  Phase 0 therefore scores the chunks that cover
    // [RADIUS - weak_peak_radius, RADIUS + weak_peak_radius] against every selected view; lanes that are WEAK by that
    // argument write their result and drop out, and phase 1 scores the remaining chunks for the others (a wave of a
    // textureless region ends after phase 0).
    const int wr = min(max(fa.weak_peak_radius, 0), RADIUS);
    const int centre_lo = ((RADIUS - wr) / APD_K14_CHUNK) * APD_K14_CHUNK;                        // first sample of the first centre chunk
    const int centre_hi = (APD_K14_CENTRE_FIRST && fa.early_out) ? ((RADIUS + wr) / APD_K14_CHUNK + 1) * APD_K14_CHUNK : 0;  // one past the last centre chunk
#pragma unroll 1
    for (int phase = (APD_K14_CENTRE_FIRST && fa.early_out) ? 0 : 1; phase < 2; ++phase) {
#pragma unroll 1
        for (int c0 = 0; c0 < NP; c0 += APD_K14_CHUNK) {
            const bool centre_chunk = c0 >= centre_lo && c0 < centre_hi;
            if (centre_chunk != (phase == 0)) {
                continue;
            }
            const int c1 = min(c0 + APD_K14_CHUNK, NP);
            const int n = c1 - c0;
            const unsigned chunk_bits = (unsigned)((in_range >> c0) & ((1ull << n) - 1ull));
            k14_chunk_t acc = 0.0f, pwc = 1.0f;   // weighted cost sums and plane distances of samples c0 .. c1 - 1
            if (__builtin_amdgcn_ballot_w64(alive && chunk_bits != 0) != 0) {   // else nobody has a sample to score in this chunk
#pragma unroll
                for (int j = 0; j < APD_K14_CHUNK; ++j) {
                    if ((chunk_bits >> j) & 1u) {   // the depth of the first loop, computed again: same operands, same bits
                        const float p_depth = fa.K[0] * base_line / (disp + (float)(c0 + j - RADIUS));
                        pwc[j] = distance_to_origin(fa, px, py, p_depth, origin.x, origin.y, origin.z);
                    }
                }
                const int mid = (c0 + c1) >> 1;
#pragma unroll 1
                for (int v = 0; v < fa.num_src; ++v) {
                    const bool use = alive && bit_test(sel, (unsigned)v) != 0;
                    if (__builtin_amdgcn_ballot_w64(use && chunk_bits != 0) == 0) {
                        continue;
                    }
                    const ViewConst &vc = view_const(fa, v);
                    const float wv = (float)vw.get(v);
                    const SrcWindow w = fw_stage<kQuad>(fa, vc, win, use && ((chunk_bits >> (mid - c0)) & 1u) != 0, px, py, origin, pwc[mid - c0]);
                    // A lane scores a view only if its pixel selected it (6.5 of 8 views on the synthetic 8-source scenes, 8.8 of 16 on the
                    // 16-source one, fewer on real ones), and a wave runs an NCC for a depth sample if ANY lane does.  When every lane of
                    // the view has every sample of the chunk (the regular case: all in [depth_min, depth_max]) the n x U (sample, lane)
                    // pairs of the chunk can be walked 64 at a time, sample-major: pair idx is sample idx / U of the lane of rank idx % U
                    // (rank -> lane through one ds_permute), its worker fetches that lane's ray, plane distance and reference
                    // moments through ds_bpermute, scores the pair from the owner's pixel position against the same window and
                    // reference tile (both belong to the wave), and the owner pulls the cost back with another ds_bpermute and
                    // adds it to its sum in view order as before: ceil(n U / 64) wave-level NCCs per chunk instead of n, same
                    // operands, same bits.  A slot of pairs costs about a tenth more than a slot of one sample (the fetches, the
                    // pull), so this path is taken when it saves two slots of the chunk, or one when every
                    // NCC also pays the geometric term.  K14 ms at 2048x1536, photometric / geometric pass: 16 sources (U = 35 of 64)
                    // 61.6 / 76.6 -> 51.9 / 59.8; with 8 sources (U = 52) a chunk saves one slot at best and stays on the sample loop.
                    const unsigned long long um = __builtin_amdgcn_ballot_w64(use);
                    const int U = __builtin_popcountll(um);
                    bool regular = false;
                    if constexpr (kPairs) {
                        regular = ((n * U + 63) >> 6) + (fa.geom_consistency ? 1 : APD_K14_PAIRS_MIN_SAVE) <= n &&
                                  __builtin_amdgcn_ballot_w64(use && chunk_bits != ((1u << n) - 1u)) == 0;
                    }
                    if (kPairs && regular) {
                        const int lane_id = threadIdx.x & 63;
                        const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                        int rank_to_lane;
                        {  // a full permutation: owners take ranks 0..U-1, the other lanes U..63
                            const unsigned long long nm = ~um;
                            const int other_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
                            const int target = use ? my_rank : U + other_rank;
                            rank_to_lane = __builtin_amdgcn_ds_permute(target << 2, lane_id);
                        }
                        const int total = n * U;
#pragma unroll 1
                        for (int first = 0; first < total; first += 64) {
                            const int idx = first + lane_id;
                            const bool valid = idx < total;
                            int j = 0;
#pragma unroll
                            for (int k = 1; k < APD_K14_CHUNK; ++k) {
                                j += (k < n && k * U <= idx) ? 1 : 0;
                            }
                            const int owner = __shfl(rank_to_lane, valid ? idx - j * U : 0);
                            const int j_lo = first / U, j_hi = (min(total, first + 64) - 1) / U;
                            float w_item = 1.0f;
#pragma unroll 1
                            for (int jj = j_lo; jj <= j_hi; ++jj) {
                                const float v_ = __shfl(pwc[jj], owner);
                                if (j == jj) {
                                    w_item = v_;
                                }
                            }
                            const float4 pl = make_float4(__shfl(origin.x, owner), __shfl(origin.y, owner), __shfl(origin.z, owner), w_item);
                            RefPatchLds<kFwPitch> orp;
                            orp.mean = __shfl(rp.mean, owner);
                            orp.var = __shfl(rp.var, owner);
                            const int olx = (wave_id & 1) * 8 + (owner & 7), oly = (wave_id >> 1) * 8 + (owner >> 3);
                            orp.base = &tile[oly * kFwPitch + olx];
                            const int opx = blockIdx.x * kFwTile + olx, opy = blockIdx.y * kFwTile + oly;
                            float tc = 0.0f;
                            if (valid) {
                                float qx, qy, qz;
                                plane_q(pl, qx, qy, qz);
                                tc += ncc_fixed_windowed<kQuad, kFwWinPitch, false, false, true>(fa, vc, w, orp, opx, opy, qx, qy, qz);
                                if (fa.geom_consistency) {
                                    tc += fa.geom_factor * geom_cost(fa, vc, opx, opy, pl);
                                }
                            }
                            // owners collect: pair (jj, my_rank) sits in slot (jj U + my_rank) / 64 at lane (jj U + my_rank) % 64
#pragma unroll 1
                            for (int jj = j_lo; jj <= j_hi; ++jj) {
                                const int at = jj * U + my_rank;
                                const float c = __shfl(tc, at & 63);
                                const float a = acc[jj];
                                acc[jj] = (use && (at >> 6) == (first >> 6)) ? a + c * wv : a;
                            }
                        }
                    } else {
#pragma unroll 1
                        for (int j = 0; j < n; ++j) {
                            const float pw_j = pwc[j], a = acc[j];
                            if (use && ((chunk_bits >> j) & 1u)) {
                                float4 pl = origin;
                                pl.w = pw_j;
                                float qx, qy, qz;
                                plane_q(pl, qx, qy, qz);
                                float tc = 0.0f;
                                tc += ncc_fixed_windowed<kQuad, kFwWinPitch, false, false, true>(fa, vc, w, rp, px, py, qx, qy, qz);
                                if (fa.geom_consistency) {
                                    tc += fa.geom_factor * geom_cost(fa, vc, px, py, pl);
                                }
                                acc[j] = a + tc * wv;
                            }
                        }
                    }
                }
            }
            // the chunk's samples are complete: :2093
#pragma unroll
            for (int j = 0; j < APD_K14_CHUNK; ++j) {
                const float p_cost = acc[j] / weight_normal;
                pc[c0 + j] = ((chunk_bits >> j) & 1u) ? ((2.0f > p_cost) ? p_cost : 2.0f) : 2.0f;  // MIN(2.0f, p_cost): NaN -> 2
            }
        }
        if (phase == 0) {
            if (alive && wr < RADIUS) {
                // Is there a sample within the radius that costs at most 0.5 and can still be a local minimum?  A neighbour that
                // phase 0 has not scored (one step outside the centre chunks) counts as "may be higher".
                bool candidate = false;
                float below = 2.0f, here = 2.0f;
                bool below_known = false;
                const int first = max(RADIUS - wr - 1, centre_lo);
                const int last = min(RADIUS + wr + 1, min(centre_hi, NP) - 1);
#pragma unroll 1
                for (int i = first; i <= last + 1; ++i) {
                    // `above` = cost of sample i, `here` = i - 1, `below` = i - 2
                    const bool above_known = i <= last;
                    const float above = above_known ? pc[i] : 2.0f;
                    const int j = i - 1;  // the sample under test
                    if (j >= max(RADIUS - wr, 2) && j <= min(RADIUS + wr, NP - 3) && j >= first) {  // the peak search covers 2 .. NP - 3 (:2104)
                        const bool lower_ok = !below_known || below > here;
                        const bool upper_ok = !above_known || above > here;
                        candidate = candidate || (lower_ok && upper_ok && !(here > 0.5f));
                    }
                    below = here;
                    below_known = j >= first;
                    here = above;
                }
                if (!candidate) {
                    fa.weak_info[center] = APD_WEAK;
                    alive = false;
                }
            }
            if (__builtin_amdgcn_ballot_w64(alive) == 0) {
                return;
            }
        }
    }
    if (!alive) {
        return;
    }
    uint64_t peaks = 0;
    int peak_count = 0, min_peak = 0;
    float min_cost = 2.0f;
    {
        float lower = pc[1], here = pc[2];
#pragma unroll 1
        for (int i = 2; i < NP - 2; ++i) {
            const float upper = pc[i + 1];
            if (lower > here && upper > here) {
                peaks |= 1ull << i;
                peak_count++;
                if (here < min_cost) {
                    min_peak = i;
                    min_cost = here;
                }
            }
            lower = here;
            here = upper;
        }
    }
    // pc[min_peak] is min_cost when a peak was found; without one min_peak = 0 is more than weak_peak_radius away from RADIUS for
    // every radius below 30, and with a larger radius sample 0 was scored like the rest
    if (abs(min_peak - RADIUS) > fa.weak_peak_radius || pc[min_peak] > 0.5f) {
        fa.weak_info[center] = APD_WEAK;
        return;
    }
    if (peak_count == 1) {
        fa.weak_info[center] = (pc[min_peak] <= 0.15f) ? APD_STRONG : APD_WEAK;
        return;
    }
    float var = 0.0f;
#pragma unroll 1
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

// ------------------------------------------------------------------------------------------------
// K15
// ------------------------------------------------------------------------------------------------

template <bool kQuad>
__global__ __launch_bounds__(256, kQuad ? APD_K15W_WAVES : APD_K1415W_WAVES_F32) void k15w_local_refine(FrameArgs fa)
{
    __shared__ float tile[kFwLds * kFwPitch];
    __shared__ uint32_t windows[4][window_dwords(kQuad, k14_win_h<kQuad>(), kFwWinPitch)];
    int px, py;
    fw_pixel(px, py);
    const RefPatchLds<kFwPitch> rp = fw_stage_ref(fa, tile, px, py);
    uint32_t *win = windows[threadIdx.x >> 6];
    const int W = fa.W, H = fa.H;
    const int center = px + py * W;
    constexpr int RADIUS = 5, NP = 2 * RADIUS + 1;

    bool alive = px < W && py < H;
    float4 origin = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
    uint32_t sel = 0;
    ViewWeights<32> vw;
    vw.clear();
    float weight_normal = 1.0f, w_now = 1.0f, acc_now = 0.0f;
    float pw[NP], pd_depth[NP], acc[NP];
    uint32_t in_range = 0;
    if (alive) {
        origin = normal_world_to_cam(fa, fa.planes[center]);
        if (origin.w == 0) {
            alive = false;
        }
    }
    if (alive) {
        sel = fa.selected_views[center];
        vw.load(fa, center);
        float base_line;
        const int valid = fw_baseline_and_weight(fa, sel, vw, base_line, weight_normal);
        if (weight_normal == 0 || valid == 0) {
            alive = false;
        } else {
            base_line /= (float)valid;
            const float disp = fa.K[0] * base_line / origin.w;
            w_now = distance_to_origin(fa, px, py, origin.w, origin.x, origin.y, origin.z);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float p_depth = fa.K[0] * base_line / (disp + (float)(i - RADIUS));
                acc[i] = 0.0f;
                pw[i] = 1.0f;
                pd_depth[i] = p_depth;
                if (!(p_depth < fa.depth_min || p_depth > fa.depth_max)) {
                    in_range |= 1u << i;
                    pw[i] = distance_to_origin(fa, px, py, p_depth, origin.x, origin.y, origin.z);
                }
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(alive) == 0) {
        return;
    }
    // Two passes over the views.  First the current depth with K14's cost form (:2173-2183), which gives cost_now.  A
    // depth sample only matters if it is the minimum and fl(cost_now - its cost) exceeds 0.1 (:2227); its weighted sum
    // never decreases from view to view (weights > 0, costs and geometric terms >= 0, monotone rounding), so once the
    // partial sum has reached `lost` -- chosen such that sum >= lost implies fl(cost_now - fl(sum / weight_normal)) <=
    // 0.0999 -- the sample can neither be adopted nor hide an adoptable one, and its remaining views are skipped.  With
    // cost_now below 0.0999 that is every sample from the start: the kernel then costs one NCC per selected view.
#pragma unroll 1
    for (int v = 0; v < fa.num_src; ++v) {
        const bool use = alive && bit_test(sel, (unsigned)v) != 0;
        if (__builtin_amdgcn_ballot_w64(use) == 0) {
            continue;
        }
        const ViewConst &vc = view_const(fa, v);
        const float wv = (float)vw.get(v);
        const SrcWindow w = fw_stage<kQuad>(fa, vc, win, use, px, py, origin, w_now);
        if (use) {
            float4 pl = origin;
            pl.w = w_now;
            float qx, qy, qz;
            plane_q(pl, qx, qy, qz);
            float tc = 0.0f;
            tc += ncc_fixed_windowed<kQuad, kFwWinPitch, false, false, true>(fa, vc, w, rp, px, py, qx, qy, qz);
            if (fa.geom_consistency) {
                tc += fa.geom_factor * geom_cost(fa, vc, px, py, pl);
            }
            acc_now += tc * wv;
        }
    }
    float lost = __builtin_inff();
    if (APD_K15_EARLY_OUT && fa.early_out && alive && !(fa.geom_factor < 0.0f)) {
        // bound >= cost_now - 0.0999 in real arithmetic (|cost_now| <= 2 + 3 * geom_factor: two roundings are far below 1e-6)
        const float bound = (acc_now / weight_normal - 0.0999f) + 1e-6f;
        if (bound <= 0.0f) {
            lost = 0.0f;
        } else if (bound < 16.0f) {  // false for NaN
            lost = (bound * weight_normal) * (1.0f + 0x1p-22f);  // sum >= lost => sum / weight_normal >= bound
        }
    }
#pragma unroll 1
    for (int v = 0; v < fa.num_src; ++v) {
        const bool use = alive && bit_test(sel, (unsigned)v) != 0;
        uint32_t open = 0;  // depth samples of this lane that can still be adopted
        if (use) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                open |= (((in_range >> i) & 1u) && !(acc[i] >= lost)) ? (1u << i) : 0u;
            }
        }
        if (__builtin_amdgcn_ballot_w64(open != 0) == 0) {
            continue;
        }
        const ViewConst &vc = view_const(fa, v);
        const float wv = (float)vw.get(v);
        const SrcWindow w = fw_stage<kQuad>(fa, vc, win, use, px, py, origin, w_now);
#pragma unroll 1
        for (int i = 0; i < NP; ++i) {  // LocalRefine's cost form (:2217-2220)
            if ((open >> i) & 1u) {
                float4 pl = origin;
                pl.w = pw[i];
                float qx, qy, qz;
                plane_q(pl, qx, qy, qz);
                const float c = ncc_fixed_windowed<kQuad, kFwWinPitch, false, false, true>(fa, vc, w, rp, px, py, qx, qy, qz);
                float a = acc[i];
                a += c * wv;
                if (fa.geom_consistency) {
                    a += fa.geom_factor * geom_cost(fa, vc, px, py, pl) * wv;
                }
                acc[i] = a;
            }
        }
    }
    if (!alive) {
        return;
    }
    const float cost_now = acc_now / weight_normal;
    float min_cost = 2.0f;
    float best_depth = origin.w;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if ((in_range >> i) & 1u) {
            const float tc = acc[i] / weight_normal;
            if (tc < min_cost) {
                min_cost = tc;
                best_depth = pd_depth[i];
            }
        }
    }
    if ((double)(cost_now - min_cost) > 0.1) {
        fa.planes[center].w = best_depth;
    }
}

hipError_t launch_k14_windowed(const FrameArgs &fa, hipStream_t s)
{
    const dim3 grid((fa.W + kFwTile - 1) / kFwTile, (fa.H + kFwTile - 1) / kFwTile);
    // the (sample, lane)-pair variant pays from fewer views on in geometric passes (a saved wave-level NCC also saves its geometric term,
    // and the walk is taken when it saves ONE): 8-bit input, K14 ms at 6200 x 4130 without / with it -- N = 8: 256.8 / 216.7 geometric,
    // 240.0 / 238.7 photometric; N = 6: 194.4 / 176.0, 182.0 / 185.9; N = 4: 121.8 / 112.6, 122.8 / 126.2; N = 2: 41.3 / 40.2, 65.4 / 67.7
    // (profiles/r06/ab_k14_pairs_by_pass_kind.txt).  Float images: 3100 x 2065, N = 8: 78.2 / 77.5, 66.3 / 68.7 -- unchanged rule.
    const int pairs_from = !fa.use_quads ? APD_K14_PAIRS_FROM_N : (fa.geom_consistency ? APD_K14_PAIRS_FROM_N_GEOM : APD_K14_PAIRS_FROM_N_PHOTO);
    const bool pairs = APD_K14_COMPACT && fa.num_src >= pairs_from;
    if (fa.use_quads) {
        if (pairs) {
            hipLaunchKernelGGL((k14w_depth_to_weak<true, true>), grid, dim3(256), 0, s, fa);
        } else {
            hipLaunchKernelGGL((k14w_depth_to_weak<true, false>), grid, dim3(256), 0, s, fa);
        }
    } else if (pairs) {
        hipLaunchKernelGGL((k14w_depth_to_weak<false, true>), grid, dim3(256), 0, s, fa);
    } else {
        hipLaunchKernelGGL((k14w_depth_to_weak<false, false>), grid, dim3(256), 0, s, fa);
    }
    return hipGetLastError();
}

hipError_t launch_k15_windowed(const FrameArgs &fa, hipStream_t s)
{
    const dim3 grid((fa.W + kFwTile - 1) / kFwTile, (fa.H + kFwTile - 1) / kFwTile);
    if (fa.use_quads) {
        hipLaunchKernelGGL(k15w_local_refine<true>, grid, dim3(256), 0, s, fa);
    } else {
        hipLaunchKernelGGL(k15w_local_refine<false>, grid, dim3(256), 0, s, fa);
    }
    return hipGetLastError();
}

}  // namespace apd


APD_WIN_STATS_ACCESSOR(apd_debug_win_stats_k1415)
