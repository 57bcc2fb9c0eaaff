// apd_capi.hip -- host side of the C ABI (include/apd_mi355x.h): one apd_context == one reference
// `APD` object (APD.h:67-145).  Owns every device allocation, the stream and the per-kernel timers.
#include <hip/hip_runtime.h>

#include <math.h>
#include <cmath>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "apd_device.h"

namespace apd {
hipError_t launch_kernel(const FrameArgs &fa, int kernel_id, int iter, hipStream_t s);
hipError_t launch_weak_kernel(const FrameArgs &fa, int kernel_id, int iter, hipStream_t s, const int *const weak_list[2], const int weak_count[2]);
size_t weak_list_scratch_ints(int W, int H);
hipError_t build_weak_lists(const FrameArgs &fa, bool all_rows, int *const list[2], int *scratch, int counts[2], hipStream_t s);
hipError_t launch_export_depth_normal(const FrameArgs &fa, float *depth, float *normal, hipStream_t s);
hipError_t launch_export_state(const FrameArgs &fa, float4 *planes4, uint8_t *weak, uint32_t *views, float *depth, hipStream_t s);
hipError_t launch_check_u8(const float *img, int n, int *flag, hipStream_t s);
hipError_t launch_weak_index_map(const uint8_t *weak, size_t n, int *map, int *scratch, hipStream_t s);
hipError_t launch_pack_quads(const float *img, int W, int H, quad_t *quad, hipStream_t s);
hipError_t launch_pack_quads_tiled(const float *img, int W, int H, quad_t *quad, hipStream_t s);
hipError_t launch_pack_fquads(const float *img, int W, int H, fquad_t *fq, hipStream_t s);
}  // namespace apd

using apd::FrameArgs;
using apd::ViewConst;

static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            return fail(APD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                    \
    } while (0)

struct apd_context {
    int device = 0;
    int W = 0, H = 0;
    apd_params params{};
    int num_images = 0;
    bool views_uploaded = false;
    bool depths_pending = false;   // apd_upload_views_split on a geometric pass: the depth maps follow with apd_upload_depths
    bool prior_uploaded = false;
    int weak_count = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device memory
    std::vector<float *> images;   // the handle's own copies (apd_upload_views); views of shared images (apd_upload_views_shared) are not kept here
    const float *ref_img = nullptr;  // reference image the kernels read: images[0] or a shared image's
    std::vector<float *> depths;
    std::vector<apd::quad_t *> quads;
    std::vector<apd::quad_t *> quads_tiled;  // second copy in 8 x 4 tiles (FIRST_INIT passes: random first iteration)
    std::vector<apd::fquad_t *> fquads;
    int *flag_dev = nullptr;
    bool use_quads = false;
    bool have_tiled = false;
    ViewConst *views_dev = nullptr;
    float4 *planes = nullptr, *fit_planes = nullptr;
    float *costs = nullptr;
    uint32_t *rng = nullptr, *selected_views = nullptr;
    uint8_t *view_weight = nullptr, *weak_info = nullptr, *weak_reliable = nullptr;
    short2 *nearest_strong = nullptr, *neighbours = nullptr;
    int8_t *column_nearest = nullptr;
    // K9/K10: compacted WEAK pixels per checkerboard colour, rebuilt when weak_info changes (upload, K4, K14)
    int *weak_list[2] = {nullptr, nullptr};
    size_t weak_list_cap = 0;  // entries per list
    int *weak_list_scratch = nullptr;
    int weak_list_count[2] = {0, 0};
    bool weak_lists_valid = false;
    bool weak_lists_all_rows = false;  // the valid lists were built for K3 (every row) / for K9, K10 (rows of the HALF launches)
    // weak_info was rewritten (K14, apd_upload_state) after the upload that sized `neighbours`, the index map and the lists: the
    // kernels that walk them (K3, K8, K9, K10) are refused until apd_upload_prior / apd_reset -- the reference builds all three
    // once per object from the map it loads (APD.cpp:526-537) and never runs a second pass on it
    bool weak_map_stale = false;
    bool first_half_done = false;  // apd_run_before_depths ran on this upload (cleared by reset / upload): apd_run_after_depths needs it
    int options[APD_OPT_COUNT] = {0, 1, 1, 1, 1, 1};  // defaults of include/apd_mi355x.h
    int *neighbours_map = nullptr;
    size_t neighbours_cap = 0;
    FrameArgs fa{};
    // profiling
    bool profiling = false;
    double prof_ms[APD_KERNEL_COUNT] = {0};
    int prof_launches[APD_KERNEL_COUNT] = {0};
    struct PendingEvent {
        int kernel;
        hipEvent_t start, stop;
    };
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t export_event = nullptr;   // recorded behind the kernel of the last export (apd_export_event)
};

static int record_export(apd_context *c)
{
    if (!c->export_event) {
        HIP_TRY(hipEventCreateWithFlags(&c->export_event, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(c->export_event, c->stream));
    return APD_OK;
}

// ---------------------------------------------------------------------------------------------
// contract C2/C3 on the host: plane-independent part of ComputeHomography (APD.cu:305-331).
// Built with -ffp-contract=off; fmaf exactly where the oracle has it.
// ---------------------------------------------------------------------------------------------
static inline float dot3_fma(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

static void relative_pose(const apd_camera &ref, const apd_camera &src, float Rr[9], float tr[3])
{
    float refC[3], srcC[3], Cr[3];
    for (int j = 0; j < 3; ++j) {
        refC[j] = -dot3_fma(ref.R[0 + j], ref.t[0], ref.R[3 + j], ref.t[1], ref.R[6 + j], ref.t[2]);
        srcC[j] = -dot3_fma(src.R[0 + j], src.t[0], src.R[3 + j], src.t[1], src.R[6 + j], src.t[2]);
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            Rr[3 * r + c] = dot3_fma(src.R[3 * r + 0], ref.R[3 * c + 0], src.R[3 * r + 1], ref.R[3 * c + 1], src.R[3 * r + 2],
                                     ref.R[3 * c + 2]);
        }
    }
    for (int j = 0; j < 3; ++j) {
        Cr[j] = refC[j] - srcC[j];
    }
    for (int r = 0; r < 3; ++r) {
        tr[r] = dot3_fma(src.R[3 * r + 0], Cr[0], src.R[3 * r + 1], Cr[1], src.R[3 * r + 2], Cr[2]);
    }
}

// K3 tests `dist / dd < thr` 1,600 times per WEAK pixel (APD.cu:1911), dist >= 0, dd = depth_max - depth_min.  For dd > 0 the
// map x -> RN(x / dd) is monotone non-decreasing, so the non-negative floats that pass form an initial segment [0, cut): the
// kernel compares with `cut` instead of dividing.  The end of the segment is found here with the same IEEE binary32
// divisions (this translation unit is compiled without fast-math): start at RN(thr * dd) and walk single floats until
// cut fails the test and its predecessor passes it.  Returns false when the parameters leave no such cut.
static bool ransac_distance_cut(float dd, float thr, float *cut)
{
    if (!(dd > 0.0f) || !std::isfinite(dd) || !std::isfinite(thr)) {
        return false;
    }
    if (!(thr > 0.0f)) {  // x / dd >= 0 is never below a threshold <= 0
        *cut = 0.0f;
        return true;
    }
    volatile float g = thr * dd;
    int steps = 0;
    while (std::isfinite(g) && (float)(g / dd) < thr) {
        g = std::nextafterf(g, INFINITY);
        if (++steps > 256) {
            return false;
        }
    }
    if (!std::isfinite(g)) {
        return false;
    }
    while (g > 0.0f) {
        const volatile float below = std::nextafterf(g, 0.0f);
        if ((float)(below / dd) < thr) {
            break;
        }
        g = below;
        if (++steps > 512) {
            return false;
        }
    }
    *cut = g;
    return true;
}

static void refresh_frame_args(apd_context *c)
{
    FrameArgs &fa = c->fa;
    const apd_params &p = c->params;
    fa.W = c->W;
    fa.H = c->H;
    fa.num_src = c->num_images > 0 ? c->num_images - 1 : 0;
    fa.half_rows = 2 * (((c->H / 2) + 15) / 16) * 16;
    fa.use_quads = c->use_quads ? 1 : 0;
    fa.have_tiled = c->have_tiled ? 1 : 0;
    fa.approx_rcp = c->options[APD_OPT_FAST_RCP];  // tolerance mode, default off: the parity target is the exact mode
    fa.tiled_mode = c->options[APD_OPT_TILED_COPY];
    fa.k67_windows = c->options[APD_OPT_K67_WINDOWS];
    fa.k1415_windows = c->options[APD_OPT_K1415_WINDOWS];
    fa.top_k = p.top_k;
    fa.depth_min = p.depth_min;
    fa.depth_max = p.depth_max;
    fa.geom_consistency = p.geom_consistency;
    fa.weak_peak_radius = p.weak_peak_radius;
    fa.rotate_time = p.rotate_time;
    fa.ransac_threshold = p.ransac_threshold;
    fa.geom_factor = p.geom_factor;
    fa.state = p.state;
    fa.seed = p.seed;
    // constants of GenNeighbours evaluated in double on the host (APD.cu:1791-1795)
    const float angle = 45.0f / (float)(p.rotate_time > 0 ? p.rotate_time : 1);
    fa.k3_cos_angle = (float)cos((double)angle * M_PI / (double)180.f);
    fa.k3_sin_angle = (float)sin((double)angle * M_PI / (double)180.f);
    fa.k3_cone = (float)cos((double)(angle / 2.0f) * M_PI / (double)180.0f);
    int shift = (int)(tan((double)(angle / 2.0f) * M_PI / (double)180.0f) * 20);
    fa.k3_shift_range = shift < 1 ? 1 : shift;
    fa.k3_dist_cut = 0.0f;
    fa.k3_cut_valid = ransac_distance_cut(p.depth_max - p.depth_min, p.ransac_threshold, &fa.k3_dist_cut) ? 1 : 0;
    fa.ref_img = c->ref_img;
    fa.views = c->views_dev;
    fa.planes = c->planes;
    fa.fit_planes = c->fit_planes;
    fa.costs = c->costs;
    fa.rng = c->rng;
    fa.selected_views = c->selected_views;
    fa.view_weight = c->view_weight;
    fa.weak_info = c->weak_info;
    fa.weak_reliable = c->weak_reliable;
    fa.nearest_strong = c->nearest_strong;
    fa.column_nearest = c->column_nearest;
    fa.neighbours_map = c->neighbours_map;
    fa.neighbours = c->neighbours;
    fa.early_out = c->options[APD_OPT_EARLY_OUT];
}

extern "C" {

void apd_default_params(apd_params *p)
{
    if (!p) {
        return;
    }
    memset(p, 0, sizeof(*p));
    p->max_iterations = 3;
    p->num_images = 5;
    p->sigma_spatial = 5.0f;
    p->sigma_color = 3.0f;
    p->top_k = 4;
    p->depth_min = 0.0f;
    p->depth_max = 1.0f;
    p->geom_consistency = 0;
    p->strong_radius = 5;
    p->strong_increment = 2;
    p->weak_radius = 5;
    p->weak_increment = 5;
    p->use_APD = 1;
    p->weak_peak_radius = 2;
    p->rotate_time = 4;
    p->ransac_threshold = 0.005f;
    p->geom_factor = 0.2f;
    p->state = APD_FIRST_INIT;
    p->seed = 12345ull;
}

const char *apd_last_error(void) { return g_last_error.c_str(); }
int apd_version(void) { return 106; }

// Digest of the HIP sources, headers and compiler flags this library was built from (apd-mvs_amd/build.py writes it next to the
// objects before compiling this file): what apd_mvs_amd.build.expected_build_id() returns for the same tree.
const char *apd_build_id(void)
{
    return
#include "apd_build_id.inc"
        ;
}

int apd_ransac_distance_cut(float depth_min, float depth_max, float ransac_threshold, float *cut)
{
    float c = 0.0f;
    const bool ok = ransac_distance_cut(depth_max - depth_min, ransac_threshold, &c);
    if (cut) {
        *cut = c;
    }
    return ok ? 1 : 0;
}

int apd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

// State a freshly constructed APD object starts from (CudaSpaceInitialization, APD.cpp:636-666); also what apd_reset
// restores, so a recycled handle behaves exactly like a new one.
static int initial_state(apd_context *c)
{
    const size_t n = (size_t)c->W * c->H;
    HIP_TRY(hipMemsetAsync(c->costs, 0, n * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->rng, 0, n * 6 * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->selected_views, 0, n * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->view_weight, 0, n * APD_MAX_IMAGES, c->stream));  // uninitialised in the reference
    HIP_TRY(hipMemsetAsync(c->planes, 0, n * sizeof(float4), c->stream));
    HIP_TRY(hipMemsetAsync(c->fit_planes, 0, n * sizeof(float4), c->stream));    // APD.cpp:651
    HIP_TRY(hipMemsetAsync(c->weak_info, APD_STRONG, n, c->stream));             // APD.cpp:541-547
    HIP_TRY(hipMemsetAsync(c->weak_reliable, 0, n, c->stream));
    HIP_TRY(hipMemsetAsync(c->nearest_strong, 0, n * sizeof(short2), c->stream));
    HIP_TRY(hipMemsetAsync(c->neighbours_map, 0, n * sizeof(int), c->stream));
    HIP_TRY(hipMemsetAsync(c->neighbours, 0, c->neighbours_cap * APD_NEIGHBOUR_NUM * sizeof(short2), c->stream));
    return APD_OK;
}

static int create_buffers(apd_context *c, int width, int height, const apd_params *params);

int apd_create(apd_handle *out, int device, int width, int height, const apd_params *params)
{
    if (!out || !params || width <= 0 || height <= 0) {
        return fail(APD_ERR_INVALID, "apd_create: bad argument");
    }
    if (width > 16384 || height > 16384) {  // pixel coordinates travel as short2 and through 24-bit multiply-adds
        return fail(APD_ERR_UNSUPPORTED, "apd_create: image larger than 16384 x 16384 px");
    }
    if (params->strong_radius != 5 || params->strong_increment != 2 || params->weak_radius != 5 || params->weak_increment != 5) {
        // the reference never changes these (main.h:84-87); the kernels are specialised for them
        return fail(APD_ERR_UNSUPPORTED, "apd_create: only strong 5/2 and weak 5/5 patch geometry is built");
    }
    if (device >= 0) {
        HIP_TRY(hipSetDevice(device));
    }
    apd_context *c = new apd_context();
    const int st = create_buffers(c, width, height, params);
    if (st != APD_OK) {
        apd_destroy(c);  // frees whatever was allocated before the failure (null pointers are skipped), the stream, the context
        return st;
    }
    *out = c;
    return APD_OK;
}

// Allocations of CudaSpaceInitialization (APD.cpp:636-666).  On failure the caller (apd_create) destroys the context.
static int create_buffers(apd_context *c, int width, int height, const apd_params *params)
{
    HIP_TRY(hipGetDevice(&c->device));
    c->W = width;
    c->H = height;
    c->params = *params;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    const size_t n = (size_t)width * height;
    // allocations of CudaSpaceInitialization (APD.cpp:636-666)
    HIP_TRY(hipMalloc(&c->costs, n * sizeof(float)));
    HIP_TRY(hipMalloc(&c->rng, n * 6 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->selected_views, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&c->view_weight, n * APD_MAX_IMAGES));
    HIP_TRY(hipMalloc(&c->planes, n * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->fit_planes, n * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->weak_info, n));
    HIP_TRY(hipMalloc(&c->weak_reliable, n));
    HIP_TRY(hipMalloc(&c->nearest_strong, n * sizeof(short2)));
    HIP_TRY(hipMalloc(&c->column_nearest, n));
    HIP_TRY(hipMalloc(&c->neighbours_map, n * sizeof(int)));
    HIP_TRY(hipMalloc(&c->views_dev, APD_MAX_IMAGES * sizeof(ViewConst)));
    HIP_TRY(hipMalloc(&c->neighbours, APD_NEIGHBOUR_NUM * sizeof(short2)));
    c->neighbours_cap = 1;
    const int st = initial_state(c);
    if (st != APD_OK) {
        return st;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    refresh_frame_args(c);
    return APD_OK;
}

int apd_reset(apd_handle c, const apd_params *params)
{
    if (!c || !params) {
        return fail(APD_ERR_INVALID, "apd_reset: bad argument");
    }
    if (params->strong_radius != 5 || params->strong_increment != 2 || params->weak_radius != 5 || params->weak_increment != 5) {
        return fail(APD_ERR_UNSUPPORTED, "apd_reset: only strong 5/2 and weak 5/5 patch geometry is built");
    }
    HIP_TRY(hipSetDevice(c->device));
    c->params = *params;
    c->views_uploaded = false;
    c->depths_pending = false;
    c->prior_uploaded = false;
    c->weak_count = 0;
    c->weak_lists_valid = false;
    c->weak_map_stale = false;
    c->first_half_done = false;
    const int st = initial_state(c);
    if (st != APD_OK) {
        return st;
    }
    refresh_frame_args(c);
    return APD_OK;
}

int apd_destroy(apd_handle c)
{
    if (!c) {
        return APD_OK;
    }
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (float *p : c->images) {
        hipFree(p);
    }
    for (float *p : c->depths) {
        hipFree(p);
    }
    for (apd::quad_t *p : c->quads) {
        hipFree(p);
    }
    for (apd::quad_t *p : c->quads_tiled) {
        hipFree(p);
    }
    for (apd::fquad_t *p : c->fquads) {
        hipFree(p);
    }
    hipFree(c->flag_dev);
    hipFree(c->views_dev);
    hipFree(c->planes);
    hipFree(c->fit_planes);
    hipFree(c->costs);
    hipFree(c->rng);
    hipFree(c->selected_views);
    hipFree(c->view_weight);
    hipFree(c->weak_info);
    hipFree(c->weak_reliable);
    hipFree(c->nearest_strong);
    hipFree(c->column_nearest);
    hipFree(c->weak_list[0]);
    hipFree(c->weak_list[1]);
    hipFree(c->weak_list_scratch);
    hipFree(c->neighbours_map);
    hipFree(c->neighbours);
    if (c->export_event) {
        hipEventDestroy(c->export_event);
    }
    for (auto &pe : c->pending) {
        hipEventDestroy(pe.start);
        hipEventDestroy(pe.stop);
    }
    for (hipEvent_t e : c->event_pool) {
        hipEventDestroy(e);
    }
    if (c->own_stream) {
        hipStreamDestroy(c->stream);
    }
    delete c;
    return APD_OK;
}

int apd_set_stream(apd_handle c, void *hip_stream)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_set_stream: null handle");
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_stream) {
        hipStreamDestroy(c->stream);
    }
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return APD_OK;
}

// Cameras, per-view constants and the image pointers the kernels read (own copies or shared images): the end of every upload.
static int finish_upload(apd_context *c, int num_images, const apd_camera *cameras, const float *const *img, const apd::quad_t *const *quad,
                         const apd::quad_t *const *tiled, const apd::fquad_t *const *fquad, bool want_depths, bool defer_depths)
{
    c->num_images = num_images;
    c->params.num_images = num_images;
    c->ref_img = img[0];
    const apd_camera &ref = cameras[0];
    FrameArgs &fa = c->fa;
    memcpy(fa.K, ref.K, sizeof(fa.K));
    memcpy(fa.R, ref.R, sizeof(fa.R));
    memcpy(fa.t, ref.t, sizeof(fa.t));
    memcpy(fa.c, ref.c, sizeof(fa.c));
    fa.ifx = 1.0f / ref.K[0];
    fa.ify = 1.0f / ref.K[4];
    std::vector<ViewConst> vcs(num_images - 1);
    for (int v = 0; v < num_images - 1; ++v) {
        const apd_camera &src = cameras[v + 1];
        ViewConst &vc = vcs[v];
        memset(&vc, 0, sizeof(vc));
        relative_pose(ref, src, vc.Rr, vc.tr);
        vc.k0 = src.K[0];
        vc.k2 = src.K[2];
        vc.k4 = src.K[4];
        vc.k5 = src.K[5];
        vc.k8 = src.K[8];
        vc.wf = (float)src.width;
        vc.hf = (float)src.height;
        memcpy(vc.K, src.K, sizeof(vc.K));
        memcpy(vc.R, src.R, sizeof(vc.R));
        memcpy(vc.t, src.t, sizeof(vc.t));
        memcpy(vc.c, src.c, sizeof(vc.c));
        vc.img = img[v + 1];
        vc.depth = want_depths ? c->depths[v + 1] : nullptr;
        vc.quad = quad[v + 1];
        vc.quad_tiled = tiled[v + 1];
        vc.fquad = fquad[v + 1];
    }
    HIP_TRY(hipMemcpyAsync(c->views_dev, vcs.data(), vcs.size() * sizeof(ViewConst), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->views_uploaded = true;
    c->depths_pending = defer_depths;
    c->first_half_done = false;
    refresh_frame_args(c);
    return APD_OK;
}

// Image / camera upload shared by apd_upload_views and apd_upload_views_split.  defer_depths: a geometric pass whose depth
// maps arrive later (apd_upload_depths): their buffers are allocated here, so that the per-view constants can point at them.
static int upload_views_impl(apd_context *c, int num_images, const apd_camera *cameras, const float *const *images, const float *const *depths,
                             bool defer_depths)
{
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->W * c->H;
    const bool want_depths = depths != nullptr || defer_depths;
    // A recycled handle (apd_reset) keeps every buffer it ever allocated -- image planes, depth planes and the derived
    // texel-pair / tiled / float-quad copies, all of one size per handle -- and only allocates what it lacks: hipFree
    // synchronises the whole device, i.e. every other handle's stream too, and a scheduler with several views in flight on
    // one device (host/multi_device.cpp) calls this once per (view, pass).  Which copies are valid is decided per upload.
    auto grow = [](auto &vec, size_t count) {
        if (vec.size() < count) {
            vec.resize(count, nullptr);
        }
    };
    grow(c->images, (size_t)num_images);
    grow(c->depths, (size_t)num_images);
    grow(c->quads, (size_t)num_images);
    grow(c->quads_tiled, (size_t)num_images);
    grow(c->fquads, (size_t)num_images);
    for (int i = 0; i < num_images; ++i) {
        if (cameras[i].width != c->W || cameras[i].height != c->H) {
            return fail(APD_ERR_INVALID, "apd_upload_views: camera %d is %dx%d, handle is %dx%d", i, cameras[i].width, cameras[i].height,
                        c->W, c->H);
        }
        if (!c->images[i]) {
            HIP_TRY(hipMalloc(&c->images[i], n * sizeof(float)));
        }
        HIP_TRY(hipMemcpyAsync(c->images[i], images[i], n * sizeof(float), hipMemcpyDefault, c->stream));
        if (want_depths) {
            if (!c->depths[i]) {
                HIP_TRY(hipMalloc(&c->depths[i], n * sizeof(float)));
            }
            if (depths) {
                HIP_TRY(hipMemcpyAsync(c->depths[i], depths[i], n * sizeof(float), hipMemcpyDefault, c->stream));
            }
        }
    }
    // 8-bit input (integers 0..255 in every view)?  Then also keep the source views as texel quads.
    if (!c->flag_dev) {
        HIP_TRY(hipMalloc(&c->flag_dev, sizeof(int)));
    }
    const int one = 1;
    HIP_TRY(hipMemcpyAsync(c->flag_dev, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
    for (int i = 0; i < num_images; ++i) {  // the reference view too: K9/K10 keep its sub-patch texels as bytes
        hipError_t e = apd::launch_check_u8(c->images[i], (int)n, c->flag_dev, c->stream);
        if (e != hipSuccess) {
            return fail(APD_ERR_HIP, "k_check_u8 failed: %s", hipGetErrorString(e));
        }
    }
    int all_u8 = 0;
    HIP_TRY(hipMemcpyAsync(&all_u8, c->flag_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->use_quads = all_u8 != 0 && c->options[APD_OPT_SOURCE_QUADS] != 0;
    if (!c->use_quads) {  // float grey values (e.g. a resampled pyramid level): float texel quads of the source views
        const size_t fn = (size_t)(c->W + 1) * (c->H + 1);
        for (int i = 1; i < num_images; ++i) {
            if (!c->fquads[i]) {
                HIP_TRY(hipMalloc(&c->fquads[i], fn * sizeof(apd::fquad_t)));
            }
            hipError_t e = apd::launch_pack_fquads(c->images[i], c->W, c->H, c->fquads[i], c->stream);
            if (e != hipSuccess) {
                return fail(APD_ERR_HIP, "k_pack_fquads failed: %s", hipGetErrorString(e));
            }
        }
    }
    if (c->use_quads) {
        const size_t qbytes = apd::quad_image_bytes(c->W, c->H);
        for (int i = 1; i < num_images; ++i) {
            if (!c->quads[i]) {
                HIP_TRY(hipMalloc(&c->quads[i], qbytes));
            }
            hipError_t e = apd::launch_pack_quads(c->images[i], c->W, c->H, c->quads[i], c->stream);
            if (e != hipSuccess) {
                return fail(APD_ERR_HIP, "k_pack_quads failed: %s", hipGetErrorString(e));
            }
        }
        // the tiled copy serves the gathers of planes that are still random: the first iteration of a FIRST_INIT pass
        // (APD_OPT_TILED_COPY = 2 uses it in every pass; 0 never builds it)
        const int tiled_mode = c->options[APD_OPT_TILED_COPY];
        c->have_tiled = tiled_mode == 2 || (tiled_mode == 1 && c->params.state == APD_FIRST_INIT);
        if (c->have_tiled) {
            const size_t tbytes = apd::quad_tiled_bytes(c->W, c->H);
            for (int i = 1; i < num_images; ++i) {
                if (!c->quads_tiled[i]) {
                    HIP_TRY(hipMalloc(&c->quads_tiled[i], tbytes));
                }
                hipError_t e = apd::launch_pack_quads_tiled(c->images[i], c->W, c->H, c->quads_tiled[i], c->stream);
                if (e != hipSuccess) {
                    return fail(APD_ERR_HIP, "k_pack_quads_tiled failed: %s", hipGetErrorString(e));
                }
            }
        }
    } else {
        c->have_tiled = false;
    }
    std::vector<const float *> img(num_images);
    std::vector<const apd::quad_t *> quad(num_images, nullptr), tiled(num_images, nullptr);
    std::vector<const apd::fquad_t *> fquad(num_images, nullptr);
    for (int i = 0; i < num_images; ++i) {
        img[i] = c->images[i];
        quad[i] = c->use_quads ? c->quads[i] : nullptr;
        tiled[i] = c->have_tiled ? c->quads_tiled[i] : nullptr;
        fquad[i] = c->use_quads ? nullptr : c->fquads[i];
    }
    return finish_upload(c, num_images, cameras, img.data(), quad.data(), tiled.data(), fquad.data(), want_depths, defer_depths);
}

static int check_upload_args(apd_context *c, int num_images, const apd_camera *cameras, const float *const *images, const char *who)
{
    if (!c || !cameras || !images || num_images < 2) {
        return fail(APD_ERR_INVALID, "%s: bad argument", who);
    }
    if (num_images > APD_MAX_IMAGES) {
        return fail(APD_ERR_TOO_MANY, "Can't process so much images: %d", num_images);  // APD.cpp:428-431
    }
    return APD_OK;
}

int apd_upload_views(apd_handle c, int num_images, const apd_camera *cameras, const float *const *images, const float *const *depths)
{
    const int rc = check_upload_args(c, num_images, cameras, images, "apd_upload_views");
    if (rc) {
        return rc;
    }
    if (c->params.geom_consistency && !depths) {
        return fail(APD_ERR_INVALID, "apd_upload_views: geom_consistency needs depth maps");
    }
    return upload_views_impl(c, num_images, cameras, images, depths, false);
}

int apd_upload_views_split(apd_handle c, int num_images, const apd_camera *cameras, const float *const *images)
{
    const int rc = check_upload_args(c, num_images, cameras, images, "apd_upload_views_split");
    if (rc) {
        return rc;
    }
    return upload_views_impl(c, num_images, cameras, images, nullptr, c->params.geom_consistency != 0);
}

// ---------------------------------------------------------------------------------------------
// Shared images: a level image of a view is the reference image of one (view, pass) and a source of ten others, pass after pass.
// apd_upload_views copies every image into the handle and packs the sources again each time (8 MB + a pack kernel + a range
// check per image and (view, pass) at 1920 x 1080); a scheduler that keeps the level images on the device creates each of them
// ONCE here -- float plane, 8-bit test, 2-byte column pairs or float texel quads, the tiled copy on first demand -- and hands
// the handles pointers.  Read-only once created; the lazy copies are made under the image's mutex and finished before it is
// released.
// ---------------------------------------------------------------------------------------------
struct apd_image {
    int device = 0, W = 0, H = 0;
    float *img = nullptr;
    apd::quad_t *pairs = nullptr, *tiled = nullptr;
    apd::fquad_t *fquads = nullptr;
    bool is_u8 = false;
    std::mutex m;
};

static int image_ensure(apd_image *im, int what /* 0 pairs, 1 tiled, 2 float quads */, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(im->m);
    hipError_t e = hipSuccess;
    if (what == 0 && !im->pairs) {
        HIP_TRY(hipMalloc(&im->pairs, apd::quad_image_bytes(im->W, im->H)));
        e = apd::launch_pack_quads(im->img, im->W, im->H, im->pairs, s);
    } else if (what == 1 && !im->tiled) {
        HIP_TRY(hipMalloc(&im->tiled, apd::quad_tiled_bytes(im->W, im->H)));
        e = apd::launch_pack_quads_tiled(im->img, im->W, im->H, im->tiled, s);
    } else if (what == 2 && !im->fquads) {
        HIP_TRY(hipMalloc(&im->fquads, (size_t)(im->W + 1) * (im->H + 1) * sizeof(apd::fquad_t)));
        e = apd::launch_pack_fquads(im->img, im->W, im->H, im->fquads, s);
    } else {
        return APD_OK;
    }
    e = e != hipSuccess ? e : hipStreamSynchronize(s);  // other handles may read the copy as soon as the mutex is free
    if (e != hipSuccess) {  // a copy that was not made must not look ready to the next caller
        void **made = what == 0 ? (void **)&im->pairs : what == 1 ? (void **)&im->tiled : (void **)&im->fquads;
        hipFree(*made);
        *made = nullptr;
        return fail(APD_ERR_HIP, "packing a shared image failed: %s", hipGetErrorString(e));
    }
    return APD_OK;
}

int apd_image_create(apd_image_t *out, int device, int width, int height, const float *pixels)
{
    if (!out || !pixels || width <= 0 || height <= 0 || width > 16384 || height > 16384) {
        return fail(APD_ERR_INVALID, "apd_image_create: bad argument");
    }
    if (device >= 0) {
        HIP_TRY(hipSetDevice(device));
    }
    apd_image *im = new apd_image();
    *out = nullptr;
    hipGetDevice(&im->device);
    im->W = width;
    im->H = height;
    const size_t n = (size_t)width * height;
    int *flag = nullptr;
    int all_u8 = 1;
    hipError_t e = hipMalloc(&im->img, n * sizeof(float));
    e = e != hipSuccess ? e : hipMemcpy(im->img, pixels, n * sizeof(float), hipMemcpyDefault);
    e = e != hipSuccess ? e : hipMalloc(&flag, sizeof(int));
    e = e != hipSuccess ? e : hipMemcpy(flag, &all_u8, sizeof(int), hipMemcpyHostToDevice);
    e = e != hipSuccess ? e : apd::launch_check_u8(im->img, (int)n, flag, nullptr);
    e = e != hipSuccess ? e : hipMemcpy(&all_u8, flag, sizeof(int), hipMemcpyDeviceToHost);
    hipFree(flag);
    if (e != hipSuccess) {
        apd_image_destroy(im);
        return fail(APD_ERR_HIP, "apd_image_create: %s", hipGetErrorString(e));
    }
    im->is_u8 = all_u8 != 0;
    const int rc = image_ensure(im, im->is_u8 ? 0 : 2, nullptr);
    if (rc != APD_OK) {
        apd_image_destroy(im);
        return rc;
    }
    *out = im;
    return APD_OK;
}

const float *apd_image_pixels(apd_image_t im) { return im ? im->img : nullptr; }

int apd_image_destroy(apd_image_t im)
{
    if (!im) {
        return APD_OK;
    }
    hipSetDevice(im->device);
    hipFree(im->img);
    hipFree(im->pairs);
    hipFree(im->tiled);
    hipFree(im->fquads);
    delete im;
    return APD_OK;
}

int apd_upload_views_shared(apd_handle c, int num_images, const apd_camera *cameras, const apd_image_t *images)
{
    if (!c || !cameras || !images || num_images < 2) {
        return fail(APD_ERR_INVALID, "apd_upload_views_shared: bad argument");
    }
    if (num_images > APD_MAX_IMAGES) {
        return fail(APD_ERR_TOO_MANY, "Can't process so much images: %d", num_images);  // APD.cpp:428-431
    }
    HIP_TRY(hipSetDevice(c->device));
    bool all_u8 = true;
    for (int i = 0; i < num_images; ++i) {
        if (!images[i] || images[i]->W != c->W || images[i]->H != c->H || images[i]->device != c->device) {
            return fail(APD_ERR_INVALID, "apd_upload_views_shared: image %d is missing, of another size or on another device than the handle", i);
        }
        if (cameras[i].width != c->W || cameras[i].height != c->H) {
            return fail(APD_ERR_INVALID, "apd_upload_views_shared: camera %d is %dx%d, handle is %dx%d", i, cameras[i].width, cameras[i].height, c->W, c->H);
        }
        all_u8 = all_u8 && images[i]->is_u8;
    }
    c->use_quads = all_u8 && c->options[APD_OPT_SOURCE_QUADS] != 0;
    const int tiled_mode = c->options[APD_OPT_TILED_COPY];
    c->have_tiled = c->use_quads && (tiled_mode == 2 || (tiled_mode == 1 && c->params.state == APD_FIRST_INIT));
    const bool geom = c->params.geom_consistency != 0;
    const size_t n = (size_t)c->W * c->H;
    if (geom) {  // the depth maps follow with apd_upload_depths: their buffers are the handle's own
        if (c->depths.size() < (size_t)num_images) {
            c->depths.resize((size_t)num_images, nullptr);
        }
        for (int i = 0; i < num_images; ++i) {
            if (!c->depths[i]) {
                HIP_TRY(hipMalloc(&c->depths[i], n * sizeof(float)));
            }
        }
    }
    std::vector<const float *> img(num_images);
    std::vector<const apd::quad_t *> quad(num_images, nullptr), tiled(num_images, nullptr);
    std::vector<const apd::fquad_t *> fquad(num_images, nullptr);
    for (int i = 0; i < num_images; ++i) {
        img[i] = images[i]->img;
        if (i == 0) {
            continue;
        }
        int rc = APD_OK;
        if (c->use_quads) {
            rc = image_ensure(images[i], 0, c->stream);
            quad[i] = images[i]->pairs;
            if (rc == APD_OK && c->have_tiled) {
                rc = image_ensure(images[i], 1, c->stream);
                tiled[i] = images[i]->tiled;
            }
        } else {
            rc = image_ensure(images[i], 2, c->stream);
            fquad[i] = images[i]->fquads;
        }
        if (rc != APD_OK) {
            return rc;
        }
    }
    return finish_upload(c, num_images, cameras, img.data(), quad.data(), tiled.data(), fquad.data(), geom, geom);
}

int apd_upload_depths(apd_handle c, int num_images, const float *const *depths)
{
    if (!c || !depths) {
        return fail(APD_ERR_INVALID, "apd_upload_depths: bad argument");
    }
    if (!c->views_uploaded || !c->depths_pending) {
        return fail(APD_ERR_STATE, "apd_upload_depths: no apd_upload_views_split of a geometric pass is waiting for depth maps");
    }
    if (num_images != c->num_images) {
        return fail(APD_ERR_INVALID, "apd_upload_depths: %d depth maps for %d views", num_images, c->num_images);
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->W * c->H;
    for (int i = 0; i < num_images; ++i) {  // on the handle's stream: ordered after the kernels already launched, before the next ones
        HIP_TRY(hipMemcpyAsync(c->depths[i], depths[i], n * sizeof(float), hipMemcpyDefault, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // the sources may be overwritten by their owners as soon as this returns
    c->depths_pending = false;
    return APD_OK;
}

int apd_upload_prior(apd_handle c, const float *planes4, const uint32_t *selected_views, const uint8_t *weak_info)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_upload_prior: null handle");
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->W * c->H;
    if (planes4) {
        HIP_TRY(hipMemcpyAsync(c->planes, planes4, n * sizeof(float4), hipMemcpyDefault, c->stream));
    } else {
        HIP_TRY(hipMemsetAsync(c->planes, 0, n * sizeof(float4), c->stream));
    }
    if (selected_views) {
        HIP_TRY(hipMemcpyAsync(c->selected_views, selected_views, n * sizeof(uint32_t), hipMemcpyDefault, c->stream));
    } else {
        HIP_TRY(hipMemsetAsync(c->selected_views, 0, n * sizeof(uint32_t), c->stream));
    }
    c->weak_count = 0;
    if (weak_info) {
        // weak index map of APD.cpp:526-537 (row-major running count of WEAK pixels), scanned on the device; fit_planes
        // (zeroed below, on the same stream) lends the scratch for the block sums
        HIP_TRY(hipMemcpyAsync(c->weak_info, weak_info, n, hipMemcpyDefault, c->stream));
        int *scratch = reinterpret_cast<int *>(c->fit_planes);
        HIP_TRY(apd::launch_weak_index_map(c->weak_info, n, c->neighbours_map, scratch, c->stream));
        int count = 0;
        HIP_TRY(hipMemcpyAsync(&count, scratch + (n + 4095) / 4096, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->weak_count = count;
    } else {
        HIP_TRY(hipMemsetAsync(c->weak_info, APD_STRONG, n, c->stream));
        HIP_TRY(hipMemsetAsync(c->neighbours_map, 0, n * sizeof(int), c->stream));
    }
    // + 1: a pixel behind the last WEAK one maps to index weak_count, and K8 forms (never follows) that address
    const size_t need = (size_t)c->weak_count + 1;
    if (need > c->neighbours_cap) {
        hipFree(c->neighbours);
        c->neighbours = nullptr;  // a failing re-allocation must not leave a dangling pointer for apd_destroy
        c->neighbours_cap = 0;
        HIP_TRY(hipMalloc(&c->neighbours, need * APD_NEIGHBOUR_NUM * sizeof(short2)));
        c->neighbours_cap = need;
    }
    HIP_TRY(hipMemsetAsync(c->neighbours, 0, c->neighbours_cap * APD_NEIGHBOUR_NUM * sizeof(short2), c->stream));
    c->weak_lists_valid = false;
    c->weak_map_stale = false;
    c->first_half_done = false;
    if (need > c->weak_list_cap) {  // one colour holds at most every WEAK pixel of the map uploaded above (K4 only removes some)
        for (int k = 0; k < 2; ++k) {
            hipFree(c->weak_list[k]);
            c->weak_list[k] = nullptr;
        }
        c->weak_list_cap = 0;
        HIP_TRY(hipMalloc(&c->weak_list[0], need * sizeof(int)));
        HIP_TRY(hipMalloc(&c->weak_list[1], need * sizeof(int)));
        c->weak_list_cap = need;
    }
    if (!c->weak_list_scratch) {
        HIP_TRY(hipMalloc(&c->weak_list_scratch, apd::weak_list_scratch_ints(c->W, c->H) * sizeof(int)));
    }
    HIP_TRY(hipMemsetAsync(c->fit_planes, 0, n * sizeof(float4), c->stream));
    HIP_TRY(hipMemsetAsync(c->view_weight, 0, n * APD_MAX_IMAGES, c->stream));
    HIP_TRY(hipMemsetAsync(c->weak_reliable, 0, n, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->prior_uploaded = true;
    refresh_frame_args(c);
    return APD_OK;
}

static hipEvent_t take_event(apd_context *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

static void drain_profile(apd_context *c)
{
    for (auto &pe : c->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
            c->prof_ms[pe.kernel] += (double)ms;
            c->prof_launches[pe.kernel] += 1;
        }
        c->event_pool.push_back(pe.start);
        c->event_pool.push_back(pe.stop);
    }
    c->pending.clear();
}

static int launch_one(apd_context *c, int kernel_id, int iter)
{
    // refusals first: a refused launch must leave nothing behind (no event taken from the pool, nothing recorded on the stream)
    if (c->weak_map_stale && (kernel_id == APD_K3_GEN_NEIGHBOURS || kernel_id == APD_K8_RANSAC_FIT_PLANE ||
                              kernel_id == APD_K9_BLACK_UPDATE_WEAK || kernel_id == APD_K10_RED_UPDATE_WEAK)) {
        return fail(APD_ERR_STATE, "kernel %d walks the WEAK lists / neighbour table of the last apd_upload_prior, but weak_info has been "
                                   "rewritten since (K14 or apd_upload_state): call apd_upload_prior or apd_reset first", kernel_id);
    }
    if (c->depths_pending && (kernel_id == APD_K9_BLACK_UPDATE_WEAK || kernel_id == APD_K10_RED_UPDATE_WEAK ||
                              kernel_id == APD_K14_DEPTH_TO_WEAK || kernel_id == APD_K15_LOCAL_REFINE)) {
        return fail(APD_ERR_STATE, "kernel %d reads the sources' depth maps (geometric term): call apd_upload_depths first", kernel_id);
    }
    apd_context::PendingEvent pe{kernel_id, nullptr, nullptr};
    if (c->profiling) {
        pe.start = take_event(c);
        pe.stop = take_event(c);
        if (hipEventRecord(pe.start, c->stream) != hipSuccess) {
            c->event_pool.push_back(pe.start);
            c->event_pool.push_back(pe.stop);
            return fail(APD_ERR_HIP, "hipEventRecord failed before kernel %d", kernel_id);
        }
    }
    hipError_t e;
    switch (kernel_id) {
    case APD_K3_GEN_NEIGHBOURS:
    case APD_K9_BLACK_UPDATE_WEAK:
    case APD_K10_RED_UPDATE_WEAK:
        if (c->weak_list[0] && (!c->weak_lists_valid || c->weak_lists_all_rows != (kernel_id == APD_K3_GEN_NEIGHBOURS))) {
            c->weak_lists_all_rows = kernel_id == APD_K3_GEN_NEIGHBOURS;
            e = apd::build_weak_lists(c->fa, c->weak_lists_all_rows, c->weak_list, c->weak_list_scratch, c->weak_list_count, c->stream);
            if (e != hipSuccess) {
                return fail(APD_ERR_HIP, "building the WEAK pixel lists failed: %s", hipGetErrorString(e));
            }
            c->weak_lists_valid = true;
        }
        e = apd::launch_weak_kernel(c->fa, kernel_id, iter, c->stream, c->weak_list, c->weak_list_count);
        break;
    case APD_K4_NEIGHBOUR_UPDATE:  // WEAK -> UNKNOWN: the lists are stale
        c->weak_lists_valid = false;
        e = apd::launch_weak_kernel(c->fa, kernel_id, iter, c->stream, nullptr, nullptr);
        break;
    case APD_K2_FIND_NEAREST_STRONG:
    case APD_K8_RANSAC_FIT_PLANE:
        e = apd::launch_weak_kernel(c->fa, kernel_id, iter, c->stream, nullptr, nullptr);
        break;
    case APD_K14_DEPTH_TO_WEAK:  // rewrites weak_info, usually with MORE WEAK pixels than the lists and the table have room for
        c->weak_lists_valid = false;
        c->weak_map_stale = true;
        e = apd::launch_kernel(c->fa, kernel_id, iter, c->stream);
        break;
    default:
        e = apd::launch_kernel(c->fa, kernel_id, iter, c->stream);
        break;
    }
    if (e != hipSuccess) {
        return fail(APD_ERR_HIP, "launch of kernel %d failed: %s", kernel_id, hipGetErrorString(e));
    }
    if (c->profiling) {
        HIP_TRY(hipEventRecord(pe.stop, c->stream));
        c->pending.push_back(pe);
    }
    return APD_OK;
}

static int check_ready(apd_context *c, const char *who)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "%s: null handle", who);
    }
    if (!c->views_uploaded) {
        return fail(APD_ERR_STATE, "%s: apd_upload_views has not been called", who);
    }
    if (c->params.state != APD_FIRST_INIT && !c->prior_uploaded) {
        return fail(APD_ERR_STATE, "%s: state != FIRST_INIT needs apd_upload_prior", who);
    }
    if (hipSetDevice(c->device) != hipSuccess) {
        return fail(APD_ERR_HIP, "%s: hipSetDevice failed", who);
    }
    return APD_OK;
}

int apd_run_kernel(apd_handle c, int kernel_id, int iter)
{
    int rc = check_ready(c, "apd_run_kernel");
    if (rc) {
        return rc;
    }
    if (kernel_id < 1 || kernel_id >= APD_KERNEL_COUNT) {
        return fail(APD_ERR_INVALID, "apd_run_kernel: unknown kernel %d", kernel_id);
    }
    return launch_one(c, kernel_id, iter);
}

int apd_run_sweeps(apd_handle c, int first_iter, int iters)
{
    int rc = check_ready(c, "apd_run_sweeps");
    if (rc) {
        return rc;
    }
    for (int i = first_iter; i < first_iter + iters; ++i) {  // APD.cu:2443-2457
        if ((rc = launch_one(c, APD_K6_BLACK_UPDATE_STRONG, i))) return rc;
        if ((rc = launch_one(c, APD_K7_RED_UPDATE_STRONG, i))) return rc;
        if ((rc = launch_one(c, APD_K8_RANSAC_FIT_PLANE, i))) return rc;
        if (c->weak_count > 0) {  // no WEAK pixel -> K9/K10 would retire every lane at once
            if ((rc = launch_one(c, APD_K9_BLACK_UPDATE_WEAK, i))) return rc;
            if ((rc = launch_one(c, APD_K10_RED_UPDATE_WEAK, i))) return rc;
        }
    }
    return APD_OK;
}

// The schedule of APD::RunPatchMatch (APD.cu:2409-2471) in two halves around the first kernel that reads a source's depth map.
// The geometric term (ComputeGeomConsistencyCost, APD.cu:752) is only evaluated by the weak update (K9/K10), K14 and K15: the
// strong sweep, K8 and everything before the loop never touch a depth map.  `first`: K1..K5, iteration 0 of K6..K8 and, while
// no WEAK pixel exists (no K9/K10), the remaining iterations and K11..K13; `second`: the rest.  Without the geometric term
// the first half is the whole pass.
static int run_schedule(apd_context *c, bool first, bool second)
{
    int rc;
    const bool geom = c->params.geom_consistency != 0;
    const bool weak = c->weak_count > 0;
    const int iters = c->params.max_iterations;
    // position of the split: number of complete iterations the first half may run, and whether it reaches past K13
    const int split_iter = !geom ? iters : (weak ? 0 : iters);      // first half runs iterations [0, split_iter) completely
    const bool head_of_split = geom && weak && iters > 0;             // ... and K6..K8 of iteration split_iter
    if (first) {
        if ((rc = launch_one(c, APD_K1_INIT_RANDOM_STATES, 0))) return rc;
        if ((rc = launch_one(c, APD_K2_FIND_NEAREST_STRONG, 0))) return rc;
        if (weak) {
            if ((rc = launch_one(c, APD_K3_GEN_NEIGHBOURS, 0))) return rc;
            if ((rc = launch_one(c, APD_K4_NEIGHBOUR_UPDATE, 0))) return rc;
        }
        if ((rc = launch_one(c, APD_K5_RANDOM_INITIALIZATION, 0))) return rc;
        if ((rc = apd_run_sweeps(c, 0, split_iter))) return rc;
        if (head_of_split) {
            if ((rc = launch_one(c, APD_K6_BLACK_UPDATE_STRONG, split_iter))) return rc;
            if ((rc = launch_one(c, APD_K7_RED_UPDATE_STRONG, split_iter))) return rc;
            if ((rc = launch_one(c, APD_K8_RANSAC_FIT_PLANE, split_iter))) return rc;
        }
    }
    if (first && (!geom || !weak)) {  // no depth map is read before K14
        if ((rc = launch_one(c, APD_K11_GET_DEPTH_NORMAL, 0))) return rc;
        if ((rc = launch_one(c, APD_K12_BLACK_FILTER, 0))) return rc;
        if ((rc = launch_one(c, APD_K13_RED_FILTER, 0))) return rc;
        if (!geom) {
            if ((rc = launch_one(c, APD_K14_DEPTH_TO_WEAK, 0))) return rc;
            if ((rc = launch_one(c, APD_K15_LOCAL_REFINE, 0))) return rc;
        }
    }
    if (second && geom) {
        if (weak) {
            if (head_of_split) {
                if ((rc = launch_one(c, APD_K9_BLACK_UPDATE_WEAK, split_iter))) return rc;
                if ((rc = launch_one(c, APD_K10_RED_UPDATE_WEAK, split_iter))) return rc;
                if ((rc = apd_run_sweeps(c, split_iter + 1, iters - split_iter - 1))) return rc;
            }
            if ((rc = launch_one(c, APD_K11_GET_DEPTH_NORMAL, 0))) return rc;
            if ((rc = launch_one(c, APD_K12_BLACK_FILTER, 0))) return rc;
            if ((rc = launch_one(c, APD_K13_RED_FILTER, 0))) return rc;
        }
        if ((rc = launch_one(c, APD_K14_DEPTH_TO_WEAK, 0))) return rc;
        if ((rc = launch_one(c, APD_K15_LOCAL_REFINE, 0))) return rc;
    }
    return APD_OK;
}

static int check_run(apd_context *c, const char *who)
{
    int rc = check_ready(c, who);
    if (rc) {
        return rc;
    }
    if (c->weak_map_stale) {
        return fail(APD_ERR_STATE, "%s: this handle already ran a pass (K14 rewrote weak_info): one handle is one (view, pass) like "
                                   "one APD object; call apd_reset or apd_upload_prior first", who);
    }
    return APD_OK;
}

int apd_run(apd_handle c)
{
    int rc = check_run(c, "apd_run");
    if (rc) {
        return rc;
    }
    if (c->depths_pending) {   // refused up front: a whole pass would otherwise run K1..K8 and fail at the first kernel that reads a depth map
        return fail(APD_ERR_STATE, "apd_run: the depth maps of this geometric pass are still pending (apd_upload_views_split / _shared): "
                                   "call apd_upload_depths first, or drive the pass with apd_run_before_depths / apd_run_after_depths");
    }
    if (c->first_half_done) {
        return fail(APD_ERR_STATE, "apd_run: apd_run_before_depths already ran on this upload; finish the pass with apd_run_after_depths");
    }
    return run_schedule(c, true, true);
}

int apd_run_before_depths(apd_handle c)
{
    int rc = check_run(c, "apd_run_before_depths");
    if (rc) {
        return rc;
    }
    if (c->first_half_done) {
        return fail(APD_ERR_STATE, "apd_run_before_depths: already ran on this upload");
    }
    rc = run_schedule(c, true, false);
    c->first_half_done = rc == APD_OK;
    return rc;
}

int apd_run_after_depths(apd_handle c)
{
    int rc = check_ready(c, "apd_run_after_depths");
    if (rc) {
        return rc;
    }
    if (!c->first_half_done) {   // K9 / K10 / K14 / K15 on planes and random states nobody initialised
        return fail(APD_ERR_STATE, "apd_run_after_depths: apd_run_before_depths has not run on this upload");
    }
    if (c->depths_pending) {
        return fail(APD_ERR_STATE, "apd_run_after_depths: the depth maps of this geometric pass have not been uploaded (apd_upload_depths)");
    }
    rc = run_schedule(c, false, true);
    c->first_half_done = false;   // the pass is complete (or failed): a second call is refused
    return rc;
}

int apd_synchronize(apd_handle c)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_synchronize: null handle");
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drain_profile(c);
    return APD_OK;
}

int apd_download(apd_handle c, float *planes4, uint8_t *weak_info, uint32_t *selected_views)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_download: null handle");
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->W * c->H;
    // APD.cu:2490-2492
    if (planes4) {
        HIP_TRY(hipMemcpyAsync(planes4, c->planes, n * sizeof(float4), hipMemcpyDefault, c->stream));
    }
    if (weak_info) {
        HIP_TRY(hipMemcpyAsync(weak_info, c->weak_info, n, hipMemcpyDefault, c->stream));
    }
    if (selected_views) {
        HIP_TRY(hipMemcpyAsync(selected_views, c->selected_views, n * sizeof(uint32_t), hipMemcpyDefault, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    drain_profile(c);
    return APD_OK;
}

static void *state_ptr(apd_context *c, int which, size_t *bytes)
{
    const size_t n = (size_t)c->W * c->H;
    switch (which) {
    case APD_STATE_PLANES: *bytes = n * 16; return c->planes;
    case APD_STATE_FIT_PLANES: *bytes = n * 16; return c->fit_planes;
    case APD_STATE_COSTS: *bytes = n * 4; return c->costs;
    case APD_STATE_RNG: *bytes = n * 24; return c->rng;
    case APD_STATE_SELECTED_VIEWS: *bytes = n * 4; return c->selected_views;
    case APD_STATE_VIEW_WEIGHT: *bytes = n * 32; return c->view_weight;
    case APD_STATE_WEAK_INFO: *bytes = n; return c->weak_info;
    case APD_STATE_WEAK_RELIABLE: *bytes = n; return c->weak_reliable;
    case APD_STATE_NEAREST_STRONG: *bytes = n * 4; return c->nearest_strong;
    case APD_STATE_NEIGHBOURS_MAP: *bytes = n * 4; return c->neighbours_map;
    case APD_STATE_NEIGHBOURS: *bytes = (size_t)(c->weak_count > 0 ? c->weak_count : 1) * APD_NEIGHBOUR_NUM * 4; return c->neighbours;
    default: *bytes = 0; return nullptr;
    }
}

size_t apd_state_bytes(apd_handle c, int which)
{
    size_t b = 0;
    if (c) {
        state_ptr(c, which, &b);
    }
    return b;
}

int apd_download_state(apd_handle c, int which, void *dst, size_t bytes)
{
    if (!c || !dst) {
        return fail(APD_ERR_INVALID, "apd_download_state: bad argument");
    }
    size_t cap = 0;
    void *src = state_ptr(c, which, &cap);
    if (!src || bytes > cap) {
        return fail(APD_ERR_INVALID, "apd_download_state: state %d has %zu bytes, asked for %zu", which, cap, bytes);
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    drain_profile(c);
    return APD_OK;
}

int apd_upload_state(apd_handle c, int which, const void *src, size_t bytes)
{
    if (!c || !src) {
        return fail(APD_ERR_INVALID, "apd_upload_state: bad argument");
    }
    size_t cap = 0;
    void *dst = state_ptr(c, which, &cap);
    if (!dst || bytes > cap) {
        return fail(APD_ERR_INVALID, "apd_upload_state: state %d has %zu bytes, got %zu", which, cap, bytes);
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (which == APD_STATE_WEAK_INFO) {
        c->weak_lists_valid = false;
        c->weak_map_stale = true;
    }
    return APD_OK;
}

int apd_export_depth_normal_device(apd_handle c, float *depth_dev, float *normal_dev)
{
    if (!c || !depth_dev) {
        return fail(APD_ERR_INVALID, "apd_export_depth_normal_device: bad argument");
    }
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = apd::launch_export_depth_normal(c->fa, depth_dev, normal_dev, c->stream);
    if (e != hipSuccess) {
        return fail(APD_ERR_HIP, "export kernel failed: %s", hipGetErrorString(e));
    }
    if (int rc = record_export(c)) {
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return APD_OK;
}

int apd_export_state_device(apd_handle c, float *planes4_dev, uint8_t *weak_dev, uint32_t *views_dev, float *depth_dev)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_export_state_device: null handle");
    }
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = apd::launch_export_state(c->fa, reinterpret_cast<float4 *>(planes4_dev), weak_dev, views_dev, depth_dev, c->stream);
    if (e != hipSuccess) {
        return fail(APD_ERR_HIP, "export kernel failed: %s", hipGetErrorString(e));
    }
    if (int rc = record_export(c)) {
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    drain_profile(c);
    return APD_OK;
}

int apd_export_event(apd_handle c, void **hip_event)
{
    if (!c || !hip_event) {
        return fail(APD_ERR_INVALID, "apd_export_event: bad argument");
    }
    *hip_event = (void *)c->export_event;
    return APD_OK;
}

int apd_set_option(apd_handle c, int option, int value)
{
    if (!c || option < 0 || option >= APD_OPT_COUNT) {
        return fail(APD_ERR_INVALID, "apd_set_option: bad handle or option %d", option);
    }
    const int hi = option == APD_OPT_TILED_COPY ? 2 : 1;
    if (value < 0 || value > hi) {
        return fail(APD_ERR_INVALID, "apd_set_option: option %d takes 0..%d, got %d", option, hi, value);
    }
    c->options[option] = value;
    refresh_frame_args(c);
    return APD_OK;
}

int apd_get_option(apd_handle c, int option, int *value)
{
    if (!c || !value || option < 0 || option >= APD_OPT_COUNT) {
        return fail(APD_ERR_INVALID, "apd_get_option: bad argument");
    }
    *value = c->options[option];
    return APD_OK;
}

int apd_get_stream(apd_handle c, void **hip_stream)
{
    if (!c || !hip_stream) {
        return fail(APD_ERR_INVALID, "apd_get_stream: bad argument");
    }
    *hip_stream = (void *)c->stream;
    return APD_OK;
}

int apd_width(apd_handle c) { return c ? c->W : 0; }
int apd_height(apd_handle c) { return c ? c->H : 0; }
float apd_depth_min(apd_handle c) { return c ? c->params.depth_min : 0.0f; }
float apd_depth_max(apd_handle c) { return c ? c->params.depth_max : 0.0f; }
int apd_weak_count(apd_handle c) { return c ? c->weak_count : 0; }

int apd_profile_enable(apd_handle c, int on)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_profile_enable: null handle");
    }
    c->profiling = on != 0;
    return APD_OK;
}

int apd_profile_reset(apd_handle c)
{
    if (!c) {
        return fail(APD_ERR_INVALID, "apd_profile_reset: null handle");
    }
    hipStreamSynchronize(c->stream);
    drain_profile(c);
    memset(c->prof_ms, 0, sizeof(c->prof_ms));
    memset(c->prof_launches, 0, sizeof(c->prof_launches));
    return APD_OK;
}

int apd_profile_get(apd_handle c, int kernel_id, double *total_ms, int *launches)
{
    if (!c || kernel_id < 0 || kernel_id >= APD_KERNEL_COUNT) {
        return fail(APD_ERR_INVALID, "apd_profile_get: bad argument");
    }
    if (total_ms) {
        *total_ms = c->prof_ms[kernel_id];
    }
    if (launches) {
        *launches = c->prof_launches[kernel_id];
    }
    return APD_OK;
}

}  // extern "C"
