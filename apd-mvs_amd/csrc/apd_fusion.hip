// apd_fusion.hip -- depth-map fusion (RunFusion, APD.cpp:826-977) on the device: apd_fuse_views of include/apd_mi355x.h.
//
// The reference fuses on the host: views in problem order, pixels in raster order, and a source pixel that supported an
// accepted point is consumed (`masks`), so later pixels of the same view can no longer use it -- pixel order is part of
// the result.  Views stay sequential here.  Within one view the per-pixel geometry (one thread per reference pixel: lift,
// project into every source view, back-project, thresholds -- apd_fusion_math.h, shared with the host build) has no
// order at all; only the consumption has, and it is resolved exactly by a fixed-point iteration:
//
//   * every pixel that is still undecided or accepted "claims" the source pixels of its (still possible) votes with
//     atomicMin(epoch-stamped raster index): claim(s) = first such pixel in raster order;
//   * an undecided pixel p looks at each of its votes: claim == p  -> nobody earlier can take it: the vote counts;
//     claim == q < p, q accepted -> consumed by q: the vote is lost for good; q undecided -> wait for the next round;
//   * when no vote is waiting, p is decided exactly as the sequential loop would decide it (sum in source order).
//
// The first undecided pixel of a round always decides (everything before it is decided), so the iteration ends, and a
// pixel's decision only ever depends on decisions of earlier pixels: same result as the raster-order loop, bit for bit.
// Accepted pixels then consume their supports, and a block scan compacts the points in raster order.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/apd_mi355x.h"
#include "apd_fusion_math.h"

namespace {

using apd_fusion::View;

constexpr int kMaxSrc = APD_MAX_IMAGES;  // sources of one reference view (main.h: MAX_IMAGES 32 includes the reference)

struct DevView {
    View geo;
    const float *image;   // rows*cols*channels, 0..255; channels 1 (grey) or 3 (blue, green, red)
    const float *depth;   // <= 0: no estimate
    const float *normal;  // 3 per pixel, world frame
    const uint8_t *weak;  // PixelState
    const uint8_t *block; // optional `blocks/mask_<id>.jpg` (APD.cpp:849-853): pixels < 128 are not fused as reference pixels
    uint8_t *consumed;    // the reference's `masks`
    unsigned long long *claim;  // epoch-stamped first claimant of this pixel in the view being fused
};

struct RefTask {
    int ref;                 // index of the reference view
    int num_src;
    int src[kMaxSrc];
    int *vote_idx;           // [pixel][num_src]: source pixel index, -1 = no (more) vote
    float *vote_w;           // exp(-score) of that vote
    uint8_t *state;          // 0 inactive, 1 undecided, 2 accepted, 3 rejected
    int *flags;              // [0] undecided pixels left after this round
    int channels;            // of the images
};

enum : uint8_t { kInactive = 0, kUndecided = 1, kAccepted = 2, kRejected = 3 };

__device__ __forceinline__ unsigned long long stamp(unsigned epoch, unsigned p) { return ((unsigned long long)(0xFFFFFFFFu - epoch) << 32) | p; }

// Votes of every reference pixel, ignoring consumption inside this view (APD.cpp:882-926).
__global__ __launch_bounds__(256) void k_fusion_votes(const DevView *__restrict__ views, RefTask task)
{
    const DevView &rv = views[task.ref];
    const int n = rv.geo.rows * rv.geo.cols;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) {
        return;
    }
    uint8_t st = kInactive;
    const float ref_depth = rv.depth[p];
    if (!(rv.block && rv.block[p] < 128) && rv.consumed[p] != 1 && !(ref_depth <= 0.0f)) {
        const int r = p / rv.geo.cols, c = p - r * rv.geo.cols;
        const float ref_n[3] = {rv.normal[3 * (size_t)p], rv.normal[3 * (size_t)p + 1], rv.normal[3 * (size_t)p + 2]};
        float P[3];
        apd_fusion::lift(rv.geo, c, r, ref_depth, P);
        int votes = 0;
        for (int j = 0; j < task.num_src; ++j) {
            const DevView &sv = views[task.src[j]];
            int idx = -1;
            float w = 0.0f;
            int sc, sr;
            if (apd_fusion::vote_target(sv.geo, P, sc, sr)) {
                const int s = sr * sv.geo.cols + sc;
                const float src_depth = sv.depth[s];
                if (sv.consumed[s] != 1 && !(src_depth <= 0.0f)) {
                    const float src_n[3] = {sv.normal[3 * (size_t)s], sv.normal[3 * (size_t)s + 1], sv.normal[3 * (size_t)s + 2]};
                    if (apd_fusion::vote_check(rv.geo, sv.geo, c, r, ref_depth, ref_n, sc, sr, src_depth, src_n, w)) {
                        idx = s;
                        votes++;
                    }
                }
            }
            task.vote_idx[(size_t)p * task.num_src + j] = idx;
            task.vote_w[(size_t)p * task.num_src + j] = w;
        }
        st = votes > 0 ? kUndecided : kRejected;  // no vote: num_consistent == 0, never a point
    }
    task.state[p] = st;
}

__global__ __launch_bounds__(256) void k_fusion_claim(const DevView *__restrict__ views, RefTask task, unsigned epoch)
{
    const DevView &rv = views[task.ref];
    const int n = rv.geo.rows * rv.geo.cols;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) {
        return;
    }
    const uint8_t st = task.state[p];
    if (st != kUndecided && st != kAccepted) {
        return;
    }
    for (int j = 0; j < task.num_src; ++j) {
        const int s = task.vote_idx[(size_t)p * task.num_src + j];
        if (s >= 0) {
            atomicMin(&views[task.src[j]].claim[s], stamp(epoch, (unsigned)p));
        }
    }
}

__global__ __launch_bounds__(256) void k_fusion_decide(const DevView *__restrict__ views, RefTask task, unsigned epoch)
{
    const DevView &rv = views[task.ref];
    const int n = rv.geo.rows * rv.geo.cols;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n || task.state[p] != kUndecided) {
        return;
    }
    bool waiting = false;
    for (int j = 0; j < task.num_src; ++j) {
        const int s = task.vote_idx[(size_t)p * task.num_src + j];
        if (s < 0) {
            continue;
        }
        const unsigned first = (unsigned)(views[task.src[j]].claim[s] & 0xFFFFFFFFull);  // stamped this round: p itself claimed
        if (first == (unsigned)p) {
            continue;  // every earlier claimant is rejected or lost this vote: it counts
        }
        // first < p.  Its state may change while this kernel runs; a stale "undecided" only costs a round.
        const uint8_t fs = reinterpret_cast<volatile uint8_t *>(task.state)[first];
        if (fs == kAccepted) {
            task.vote_idx[(size_t)p * task.num_src + j] = -1;  // consumed by an earlier point of this view
        } else {
            waiting = true;  // undecided, or rejected a moment ago (then the next claim round names its successor)
        }
    }
    if (waiting) {
        atomicAdd(&task.flags[0], 1);
        return;
    }
    int agreeing = 0;
    float consistency = 0.0f;
    for (int j = 0; j < task.num_src; ++j) {
        if (task.vote_idx[(size_t)p * task.num_src + j] >= 0) {
            consistency += task.vote_w[(size_t)p * task.num_src + j];
            agreeing++;
        }
    }
    const bool ok = apd_fusion::accept_point(agreeing, consistency, (int)rv.weak[p]);
    __threadfence();
    reinterpret_cast<volatile uint8_t *>(task.state)[p] = ok ? kAccepted : kRejected;
}

// Accepted pixels consume their supports and produce their point (APD.cpp:939-960); per-block counts for the scan.
__global__ __launch_bounds__(256) void k_fusion_emit(const DevView *__restrict__ views, RefTask task, float *__restrict__ xyz_sparse,
                                                      uint8_t *__restrict__ bgr_sparse, int *__restrict__ block_counts)
{
    const DevView &rv = views[task.ref];
    const int n = rv.geo.rows * rv.geo.cols;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool acc = p < n && task.state[p] == kAccepted;
    if (acc) {
        const int r = p / rv.geo.cols, c = p - r * rv.geo.cols;
        float P[3];
        apd_fusion::lift(rv.geo, c, r, rv.depth[p], P);
        const int nc = task.channels;
        float colour[3];
        for (int k = 0; k < 3; ++k) {
            colour[k] = rv.image[(size_t)p * nc + (nc == 3 ? k : 0)];
        }
        int agreeing = 0;
        for (int j = 0; j < task.num_src; ++j) {
            const int s = task.vote_idx[(size_t)p * task.num_src + j];
            if (s >= 0) {
                const DevView &sv = views[task.src[j]];
                sv.consumed[s] = 1;
                for (int k = 0; k < 3; ++k) {
                    colour[k] += sv.image[(size_t)s * nc + (nc == 3 ? k : 0)];
                }
                agreeing++;
            }
        }
        xyz_sparse[3 * (size_t)p + 0] = P[0];
        xyz_sparse[3 * (size_t)p + 1] = P[1];
        xyz_sparse[3 * (size_t)p + 2] = P[2];
        for (int k = 0; k < 3; ++k) {
            bgr_sparse[3 * (size_t)p + k] = static_cast<uint8_t>(colour[k] / (agreeing + 1));
        }
    }
    const unsigned long long m = __ballot(acc);
    __shared__ int wave_counts[4];
    if ((threadIdx.x & 63) == 0) {
        wave_counts[threadIdx.x >> 6] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_counts[blockIdx.x] = wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
    }
}

// exclusive scan of the block counts (one workgroup; a view has at most a few hundred thousand blocks)
__global__ __launch_bounds__(1024) void k_fusion_scan(int *__restrict__ counts, int nblocks, int *__restrict__ total)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = t * per, b1 = min(b0 + per, nblocks);
    int sum = 0;
    for (int b = b0; b < b1; ++b) {
        sum += counts[b];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int b = b0; b < b1; ++b) {
        const int c = counts[b];
        counts[b] = run;
        run += c;
    }
    if (t == 1023) {
        *total = part[1023];
    }
}

// Packs the accepted points of a view in raster order as the 15-byte records of the PLY body (x y z float, diffuse_blue /
// green / red uchar, APD.cpp:214-254): one download per view straight into the file image, no per-point loop on the host.
__global__ __launch_bounds__(256) void k_fusion_compact(RefTask task, int n, const float *__restrict__ xyz_sparse,
                                                         const uint8_t *__restrict__ bgr_sparse, const int *__restrict__ block_offsets,
                                                         uint8_t *__restrict__ records)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool acc = p < n && task.state[p] == kAccepted;
    const unsigned long long m = __ballot(acc);
    __shared__ int wave_counts[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        wave_counts[wave] = __popcll(m);
    }
    __syncthreads();
    if (acc) {
        int pos = block_offsets[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) {
            pos += wave_counts[w];
        }
        uint8_t *rec = records + (size_t)pos * 15;
        for (int k = 0; k < 3; ++k) {
            const uint32_t bits = __float_as_uint(xyz_sparse[3 * (size_t)p + k]);  // little endian, as the host's memcpy wrote them
            rec[4 * k + 0] = (uint8_t)(bits & 0xFFu);
            rec[4 * k + 1] = (uint8_t)((bits >> 8) & 0xFFu);
            rec[4 * k + 2] = (uint8_t)((bits >> 16) & 0xFFu);
            rec[4 * k + 3] = (uint8_t)(bits >> 24);
        }
        rec[12] = bgr_sparse[3 * (size_t)p + 0];
        rec[13] = bgr_sparse[3 * (size_t)p + 1];
        rec[14] = bgr_sparse[3 * (size_t)p + 2];
    }
}

thread_local std::string g_fusion_error;
thread_local double g_fusion_ms[3] = {0.0, 0.0, 0.0};  // last apd_fuse_views: set-up (allocations, uploads), views (kernels + point downloads), PLY file

int fusion_fail(int code, const char *what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "apd_fuse_views: %s: %s", what, hipGetErrorString(e));
    g_fusion_error = buf;
    return code;
}

#define FUS_TRY(expr)                                        \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) {                              \
            cleanup();                                       \
            return fusion_fail(APD_ERR_HIP, #expr, e_);      \
        }                                                    \
    } while (0)

}  // namespace

extern "C" const char *apd_fusion_last_error(void) { return g_fusion_error.c_str(); }

extern "C" int apd_fusion_last_timing(double *setup_ms, double *views_ms, double *file_ms)
{
    if (setup_ms) {
        *setup_ms = g_fusion_ms[0];
    }
    if (views_ms) {
        *views_ms = g_fusion_ms[1];
    }
    if (file_ms) {
        *file_ms = g_fusion_ms[2];
    }
    return APD_OK;
}

extern "C" int apd_fuse_views(int device, int num_views, const apd_camera *cameras, const float *const *images, int image_channels,
                              const float *const *depths, const float *const *normals, const uint8_t *const *weaks,
                              const uint8_t *const *blocks, const int *rows, const int *cols, const int *pair_offsets, const int *pair_indices, int maps_on_device,
                              const char *ply_path, long long *num_points)
{
    g_fusion_error.clear();
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    if (num_views <= 0 || !cameras || !images || !depths || !normals || !weaks || !rows || !cols || !pair_offsets || !pair_indices ||
        !ply_path || !num_points) {
        g_fusion_error = "apd_fuse_views: null argument";
        return APD_ERR_INVALID;
    }
    if (image_channels != 1 && image_channels != 3) {
        g_fusion_error = "apd_fuse_views: images have 1 (grey) or 3 (blue, green, red) channels";
        return APD_ERR_INVALID;
    }
    for (int i = 0; i < num_views; ++i) {
        const int ns = pair_offsets[i + 1] - pair_offsets[i];
        if (ns < 0 || ns > kMaxSrc) {
            g_fusion_error = "apd_fuse_views: a view has more than APD_MAX_IMAGES sources";
            return APD_ERR_INVALID;
        }
        for (int k = pair_offsets[i]; k < pair_offsets[i + 1]; ++k) {
            if (pair_indices[k] < 0 || pair_indices[k] >= num_views) {
                g_fusion_error = "apd_fuse_views: source index out of range";
                return APD_ERR_INVALID;
            }
            if (pair_indices[k] == i) {  // the consumption of a view's own pixels would be order dependent inside the vote kernel
                g_fusion_error = "apd_fuse_views: a view lists itself as a source (use the host fusion)";
                return APD_ERR_INVALID;
            }
        }
    }
    std::vector<void *> owned;
    std::vector<DevView> hv(num_views);
    DevView *dviews = nullptr;
    void *staging = nullptr;  // page-locked buffer of the point downloads
    auto cleanup = [&]() {
        for (void *p : owned) {
            hipFree(p);
        }
        owned.clear();
        if (staging) {
            hipHostFree(staging);
            staging = nullptr;
        }
    };
    auto dev_alloc = [&](size_t bytes, void **out) -> hipError_t {
        hipError_t e = hipMalloc(out, bytes > 0 ? bytes : 1);
        if (e == hipSuccess) {
            owned.push_back(*out);
        }
        return e;
    };
    FUS_TRY(hipSetDevice(device));
    size_t max_px = 0;
    for (int i = 0; i < num_views; ++i) {
        const size_t n = (size_t)rows[i] * cols[i];
        max_px = n > max_px ? n : max_px;
        DevView &v = hv[i];
        const apd_camera &c = cameras[i];
        memcpy(v.geo.K, c.K, sizeof(v.geo.K));
        memcpy(v.geo.R, c.R, sizeof(v.geo.R));
        memcpy(v.geo.t, c.t, sizeof(v.geo.t));
        // -R^T t in float, term order of Get3DPointonWorld (APD.cpp:795-798)
        v.geo.centre[0] = -(c.R[0] * c.t[0] + c.R[3] * c.t[1] + c.R[6] * c.t[2]);
        v.geo.centre[1] = -(c.R[1] * c.t[0] + c.R[4] * c.t[1] + c.R[7] * c.t[2]);
        v.geo.centre[2] = -(c.R[2] * c.t[0] + c.R[5] * c.t[1] + c.R[8] * c.t[2]);
        v.geo.rows = rows[i];
        v.geo.cols = cols[i];
        if (maps_on_device) {
            v.image = images[i];
            v.depth = depths[i];
            v.normal = normals[i];
            v.weak = weaks[i];
            v.block = blocks ? blocks[i] : nullptr;
        } else {
            void *g, *d, *nm, *w;
            FUS_TRY(dev_alloc(n * 4 * image_channels, &g));
            FUS_TRY(dev_alloc(n * 4, &d));
            FUS_TRY(dev_alloc(n * 12, &nm));
            FUS_TRY(dev_alloc(n, &w));
            FUS_TRY(hipMemcpy(g, images[i], n * 4 * image_channels, hipMemcpyHostToDevice));
            FUS_TRY(hipMemcpy(d, depths[i], n * 4, hipMemcpyHostToDevice));
            FUS_TRY(hipMemcpy(nm, normals[i], n * 12, hipMemcpyHostToDevice));
            FUS_TRY(hipMemcpy(w, weaks[i], n, hipMemcpyHostToDevice));
            v.image = (const float *)g;
            v.depth = (const float *)d;
            v.normal = (const float *)nm;
            v.weak = (const uint8_t *)w;
            v.block = nullptr;
            if (blocks && blocks[i]) {
                void *b;
                FUS_TRY(dev_alloc(n, &b));
                FUS_TRY(hipMemcpy(b, blocks[i], n, hipMemcpyHostToDevice));
                v.block = (const uint8_t *)b;
            }
        }
        void *cons, *claim;
        FUS_TRY(dev_alloc(n, &cons));
        FUS_TRY(dev_alloc(n * 8, &claim));
        FUS_TRY(hipMemset(cons, 0, n));
        FUS_TRY(hipMemset(claim, 0xFF, n * 8));
        v.consumed = (uint8_t *)cons;
        v.claim = (unsigned long long *)claim;
    }
    {
        void *p;
        FUS_TRY(dev_alloc(sizeof(DevView) * num_views, &p));
        dviews = (DevView *)p;
        FUS_TRY(hipMemcpy(dviews, hv.data(), sizeof(DevView) * num_views, hipMemcpyHostToDevice));
    }
    int max_src = 1;
    for (int i = 0; i < num_views; ++i) {
        max_src = std::max(max_src, pair_offsets[i + 1] - pair_offsets[i]);
    }
    const int max_blocks = (int)((max_px + 255) / 256);
    void *vote_idx, *vote_w, *state, *flags, *xyz_sparse, *grey_sparse, *block_counts, *total, *records;
    FUS_TRY(dev_alloc(max_px * max_src * 4, &vote_idx));
    FUS_TRY(dev_alloc(max_px * max_src * 4, &vote_w));
    FUS_TRY(dev_alloc(max_px, &state));
    FUS_TRY(dev_alloc(sizeof(int), &flags));
    FUS_TRY(dev_alloc(max_px * 12, &xyz_sparse));
    FUS_TRY(dev_alloc(max_px * 3, &grey_sparse));
    FUS_TRY(dev_alloc((size_t)max_blocks * 4, &block_counts));
    FUS_TRY(dev_alloc(sizeof(int), &total));
    FUS_TRY(dev_alloc(max_px * 15, &records));

    // PLY records: x y z float + diffuse_blue/green/red uchar (APD.cpp:214-254), one buffer per view (one growing vector re-allocates and
    // copies hundreds of megabytes at Tanks&Temples scale), downloaded through one page-locked staging buffer
    std::vector<std::vector<uint8_t>> body((size_t)num_views);
    if (hipHostMalloc(&staging, max_px * 15 > 0 ? max_px * 15 : 1, hipHostMallocDefault) != hipSuccess) {
        staging = nullptr;  // pageable downloads then
    }
    g_fusion_ms[0] = ms_since(t_begin);
    const auto t_views = std::chrono::steady_clock::now();
    long long count = 0;
    unsigned epoch = 0;
    for (int i = 0; i < num_views; ++i) {
        const int n = rows[i] * cols[i];
        const int blocks = (n + 255) / 256;
        RefTask task;
        task.ref = i;
        task.num_src = pair_offsets[i + 1] - pair_offsets[i];
        for (int j = 0; j < task.num_src; ++j) {
            task.src[j] = pair_indices[pair_offsets[i] + j];
        }
        task.vote_idx = (int *)vote_idx;
        task.vote_w = (float *)vote_w;
        task.state = (uint8_t *)state;
        task.flags = (int *)flags;
        task.channels = image_channels;
        if (n == 0) {
            continue;
        }
        hipLaunchKernelGGL(k_fusion_votes, dim3(blocks), dim3(256), 0, 0, dviews, task);
        FUS_TRY(hipGetLastError());
        int rounds = 0;
        for (;;) {
            ++epoch;
            ++rounds;
            FUS_TRY(hipMemsetAsync(flags, 0, sizeof(int), 0));
            hipLaunchKernelGGL(k_fusion_claim, dim3(blocks), dim3(256), 0, 0, dviews, task, epoch);
            hipLaunchKernelGGL(k_fusion_decide, dim3(blocks), dim3(256), 0, 0, dviews, task, epoch);
            FUS_TRY(hipGetLastError());
            int undecided = 0;
            FUS_TRY(hipMemcpy(&undecided, flags, sizeof(int), hipMemcpyDeviceToHost));
            if (undecided == 0) {
                break;
            }
            if (rounds > n) {  // cannot happen: the first undecided pixel decides in every round
                cleanup();
                g_fusion_error = "apd_fuse_views: consumption rounds did not converge";
                return APD_ERR_STATE;
            }
        }
        hipLaunchKernelGGL(k_fusion_emit, dim3(blocks), dim3(256), 0, 0, dviews, task, (float *)xyz_sparse, (uint8_t *)grey_sparse,
                           (int *)block_counts);
        hipLaunchKernelGGL(k_fusion_scan, dim3(1), dim3(1024), 0, 0, (int *)block_counts, blocks, (int *)total);
        hipLaunchKernelGGL(k_fusion_compact, dim3(blocks), dim3(256), 0, 0, task, n, (const float *)xyz_sparse,
                           (const uint8_t *)grey_sparse, (const int *)block_counts, (uint8_t *)records);
        FUS_TRY(hipGetLastError());
        int npts = 0;
        FUS_TRY(hipMemcpy(&npts, total, sizeof(int), hipMemcpyDeviceToHost));
        (void)rounds;
        if (npts > 0) {
            body[i].resize((size_t)npts * 15);
            if (staging) {
                FUS_TRY(hipMemcpy(staging, records, (size_t)npts * 15, hipMemcpyDeviceToHost));
                memcpy(body[i].data(), staging, (size_t)npts * 15);
            } else {
                FUS_TRY(hipMemcpy(body[i].data(), records, (size_t)npts * 15, hipMemcpyDeviceToHost));
            }
            count += npts;
        }
    }
    g_fusion_ms[1] = ms_since(t_views);
    const auto t_file = std::chrono::steady_clock::now();
    cleanup();
    FILE *f = fopen(ply_path, "wb");
    if (!f) {
        g_fusion_error = std::string("apd_fuse_views: cannot write ") + ply_path;
        return APD_ERR_IO;
    }
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
               "property uchar diffuse_blue\nproperty uchar diffuse_green\nproperty uchar diffuse_red\nend_header\n", (int)count);
    bool ok = true;
    for (const std::vector<uint8_t> &part : body) {
        ok = ok && (part.empty() || fwrite(part.data(), 1, part.size(), f) == part.size());
    }
    if (fclose(f) != 0 || !ok) {
        g_fusion_error = std::string("apd_fuse_views: short write to ") + ply_path;
        return APD_ERR_IO;
    }
    *num_points = count;
    g_fusion_ms[2] = ms_since(t_file);
    return APD_OK;
}
