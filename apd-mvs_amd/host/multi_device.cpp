// multi_device.cpp -- the pass table of the reference driver (main.cpp:168-215) in memory, on one or several devices of one
// node, with several views in flight per device.
//
// The reference takes one device index (main.cpp:149-153), processes the views one after the other and hands state from
// pass to pass through four files per view; in a geometric pass a view reads its sources' depths.dmb as they are at that
// moment.  Here:
//   * rank r of the device list owns the reference views r, r + G, r + 2G, ... (round-robin: neighbouring views are usually
//     each other's sources).  A rank runs `lanes` views at a time, each on its own host thread, handle and stream: one
//     view's launches leave a 256-CU device partly idle (a 960 x 540 level is 1.3 rounds of workgroups); the lanes of a
//     rank share its image and depth buffers;
//   * planes, weak map and selected views of a view stay on its rank's device from pass to pass (apd_export_state_device ->
//     apd_upload_prior, device to device); the nearest-neighbour resampling between pyramid levels runs on the device too;
//   * after every pass the depth maps of all views are all-gathered (apd_exchange_allgather: RCCL over xGMI, or direct peer
//     copies), which replaces the exchange through depths.dmb; after the last pass the planes (normal + depth) and weak
//     maps are all-gathered view by view and rank 0's copy is fused where it lies (apd_fuse_views on device pointers);
//   * order of views.  `APD folder gpu` (one rank, the default of the drop-in): the REFERENCE's order -- a view of a
//     geometric pass reads this pass's depth map of every source that precedes it in pair.txt and the previous pass's of
//     the others (Gauss-Seidel; what the files hold at that moment, main.cpp:117-124, APD.cpp:497-500), so the bytes are
//     those of the file-based driver.  Views still overlap: a photometric pass has no dependency between views at all
//     (main.cpp:182), and in a geometric pass only the weak update, K14 and K15 read depth maps (APD.cu:752), so a lane runs
//     the first half of its view (apd_run_before_depths), waits until its earlier sources have published, uploads the maps
//     (apd_upload_depths) and runs the rest.  Device lists and --jacobi: every view reads the PREVIOUS pass's maps (Jacobi),
//     so the result does not depend on the number of ranks or lanes: `APD folder 0 --jacobi` and `APD folder 0,0,0` write
//     the same bytes (tests/test_gpu_dropin_binary.py), which differ slightly, by construction, from the reference order.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "APD.h"
#include "schedule.h"

namespace {

// A failing device call ends the run like the reference's CudaSafeCall (APD.cpp:315-323), but not from inside a worker
// thread whose siblings are mid-launch: the worker records the message and stops, the others stop at their next step, and
// the thread that joins them prints it and returns the exit code.
struct Failure {
    std::atomic<bool> failed{false};
    std::mutex m;
    std::string what;
    void set(const std::string &msg)
    {
        std::lock_guard<std::mutex> lock(m);
        if (!failed.load()) {
            what = msg;
            failed.store(true);
        }
    }
};

void Check(int rc, const char *what)
{
    if (rc != APD_OK) {
        const char *a = apd_last_error(), *b = apd_exchange_last_error();
        throw std::runtime_error(std::string(what) + " failed: " + (a ? a : "") + ((b && b[0]) ? " / " : "") + ((b && b[0]) ? b : ""));
    }
}

struct DeviceBuffer {
    int device = 0;
    void *p = nullptr;
    size_t bytes = 0;
    void alloc(int dev, size_t n)
    {
        release();
        device = dev;
        bytes = n;
        Check(apd_device_malloc(dev, n, &p), "apd_device_malloc");
    }
    void release()
    {
        if (p) {
            apd_device_free(device, p);
            p = nullptr;
        }
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// state of one reference view between passes, on its rank's device
struct ResidentView {
    DeviceBuffer planes, weak, views;  // float4, uint8, uint32 at (W, H)
    int W = 0, H = 0;
    bool valid = false;
};

// one view in flight: host thread + handle + stream
struct Lane {
    apd_handle handle = nullptr;
    int handle_w = 0, handle_h = 0;
    void *export_event = nullptr;   // hipEvent_t behind the handle's newest export (apd_export_event); written and read under done_m
    DeviceBuffer scratch_planes, scratch_weak, scratch_views;  // resampling targets (swapped with a view's buffers)
};

struct Rank {
    int device = 0;
    std::vector<int> own;                // view indices, ascending
    std::vector<Lane> lanes;
    std::vector<char> needs;             // image i is the reference or a source of one of this rank's views
    std::vector<apd_image_t> images;     // level image of every needed view, shared by the rank's handles (apd_image_create: uploaded, tested, packed once)
    DeviceBuffer send, recv;             // depth blocks of the all-gather (float): this pass's own maps / every view's of the pass before
    DeviceBuffer zero_depth;             // a source-only view has no estimate
    std::unordered_map<int, ResidentView> state;
    std::atomic<int> next{0};            // next entry of `own` to hand to a lane
};

apd_params ToAbi(const PatchMatchParams &q)
{
    apd_params p;
    apd_default_params(&p);
    p.max_iterations = q.max_iterations;
    p.num_images = q.num_images;
    p.sigma_spatial = q.sigma_spatial;
    p.sigma_color = q.sigma_color;
    p.top_k = q.top_k;
    p.depth_min = q.depth_min;
    p.depth_max = q.depth_max;
    p.geom_consistency = q.geom_consistency ? 1 : 0;
    p.strong_radius = q.strong_radius;
    p.strong_increment = q.strong_increment;
    p.weak_radius = q.weak_radius;
    p.weak_increment = q.weak_increment;
    p.use_APD = q.use_APD ? 1 : 0;
    p.weak_peak_radius = q.weak_peak_radius;
    p.rotate_time = q.rotate_time;
    p.ransac_threshold = q.ransac_threshold;
    p.geom_factor = q.geom_factor;
    p.state = (int)q.state;
    p.seed = q.seed;
    return p;
}

struct StageClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    long long lap()
    {
        const auto now = std::chrono::steady_clock::now();
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(now - t).count();
        t = now;
        return ms;
    }
};

// Views in flight per device by level size (profiles/r04/ab_lanes_tt24.txt, 24 views of 1920 x 1080, passes in s: one 9.08, two 8.15,
// three 8.29, four 7.92, six 8.00, eight 8.08 in the reference's order; 2 % at 6200 x 4130).
}  // namespace

// Views in flight per rank by level size.  Above 12 Mpix one view fills the device: two in flight at 6200 x 4130 run the passes in 15.71 s
// instead of 15.80 (profiles/r04/ab_lanes_25mpix.txt) for twice the handle memory.
#ifndef APD_GS_LANES_PER_PASS
#define APD_GS_LANES_PER_PASS 2  // lanes that may work on one geometric pass in the reference's order (its second halves form a chain); 24 x 1080p passes: 1: 8.16 s, 2: 7.95, 3: 7.89, 9: 7.99 (profiles/r04/ab_gs_lanes_per_pass_tt24.txt)
#endif
#ifndef APD_LANES_ABOVE_12MPIX
#define APD_LANES_ABOVE_12MPIX 1
#endif
int DefaultLanes(size_t pixels)
{
    return pixels <= ((size_t)1 << 20) ? 6 : (pixels <= ((size_t)4 << 20) ? 4 : (pixels <= ((size_t)12 << 20) ? 2 : APD_LANES_ABOVE_12MPIX));
}

// Device bytes an in-memory run keeps resident on its busiest device (rank 0 also fuses): the shared level images (float plane +
// packed copies: 8.5 B/px each at the finest level), the two sets of depth maps, every owned view's state, per lane the
// resampling scratch and a handle's own arrays (state 107 B/px, WEAK lists and neighbour table, the depth maps of a geometric
// pass) -- and at the end the gathered final maps, their depth / normal split and the fusion's buffers.  Upper bound per pixel of
// the finest level; main() compares it with the free memory before choosing this scheduler and every in-memory mode checks it
// again here.
// fusion_prefetch: the colour images (3 floats per pixel and view) and block masks (1 B) of the fusion are uploaded to the first
// device WHILE the passes run (StartFusionInputs), so they count towards the passes' footprint, not only the final stage's.
double InMemoryBytesPerPixel(int num_images, int num_views, int num_ranks, int lanes, int max_sources, double *passes_out, double *final_out,
                             bool fusion_prefetch)
{
    const double slots = (double)((num_views + num_ranks - 1) / num_ranks);
    const double m = (double)max_sources;
    const double passes = 8.5 * num_images + 4.0 * slots * (1.0 + num_ranks) + 4.0 + 21.0 * slots + lanes * (21.0 + 107.0 + 25.0 + 4.0 * (m + 1.0)) +
                          (fusion_prefetch ? 13.0 * num_views : 0.0);
    const double final_stage = 21.0 * slots + (num_ranks > 1 ? 17.0 * slots * num_ranks : 0.0) + 16.0 * num_views + 21.0 * num_views + 8.0 * m + 40.0;
    if (passes_out) {
        *passes_out = passes;
    }
    if (final_out) {
        *final_out = final_stage;
    }
    return std::max(passes, final_stage);
}

// Views in flight per rank of a run over `num_views` views on `num_ranks` ranks whose finest level is width x height: the larger of
// what the finest and the coarsest pyramid level ask for (a coarse level's launches are small, more of its views fit the device side
// by side), never more than a rank owns.  ONE function for main()'s choice of scheduler and for RunMultiDevice's own fit test: the
// two used to count differently (6 against 4 at 1920 x 1080) and a folder main() had accepted could be refused here.
int InMemoryLanes(const Options &opt, int width, int height, int num_views, int num_ranks, bool distinct_devices)
{
    const int round_num = opt.single_level ? 1 : RoundNum(width, height);
    const int coarsest = opt.single_level ? 1 : 1 << (round_num - 1);
    const int own = std::max(1, (num_views + num_ranks - 1) / num_ranks);
    auto lanes_at = [&](size_t level_pixels) {
        const int want = opt.ranks_per_device > 0 ? opt.ranks_per_device : (distinct_devices ? DefaultLanes(level_pixels) : 1);
        return std::max(1, std::min(want, own));
    };
    return std::max(lanes_at((size_t)width * height),
                    lanes_at((size_t)std::lround(width / (double)coarsest) * (size_t)std::lround(height / (double)coarsest)));
}

InMemoryFit TestInMemoryFit(const Options &opt, int device, int width, int height, int num_images, int num_views, int num_ranks, int lanes, int max_sources)
{
    InMemoryFit fit;
    size_t free_bytes = 0, total_bytes = 0;
    double per_px_passes = 0, per_px_final = 0;
    const double pixels = (double)width * (double)height;
    const bool prefetch = !(opt.no_fusion || opt.late_fusion_inputs);
    fit.need_bytes = pixels * InMemoryBytesPerPixel(num_images, num_views, num_ranks, lanes, max_sources, &per_px_passes, &per_px_final, prefetch);
    fit.have_memory = apd_device_memory(device, &free_bytes, &total_bytes) == APD_OK;
    if (fit.have_memory && opt.scheduler_free_gb > 0) {
        free_bytes = std::min(free_bytes, (size_t)(opt.scheduler_free_gb * 1e9));
    }
    fit.free_bytes = (double)free_bytes;
    // room for the passes' buffers AND the final maps at once: the handles, level images and depth sets are then left to the end of
    // the process instead of being released one hipFree (= one device synchronisation) at a time before the fusion
    fit.release_before_fusion = !fit.have_memory || pixels * (per_px_passes + per_px_final) > 0.8 * (double)free_bytes;
    fit.fits = !fit.have_memory || fit.need_bytes <= 0.9 * (double)free_bytes;
    return fit;
}

int RunMultiDevice(const Options &opt, std::vector<Problem> &problems)
{
    StageClock stage;
    long long ms_load = 0, ms_setup = 0, ms_upload = 0, ms_passes = 0, ms_gather = 0, ms_fusion = 0;
    std::vector<int> devices = opt.devices;
    const int V = (int)problems.size();
    if (devices.empty() || V < 1) {
        fprintf(stderr, "nothing to do\n");
        return EXIT_FAILURE;
    }
    // ---- views: the reference views in pair.txt order, then the source-only images (main.cpp) ----
    std::vector<int> ids;
    std::unordered_map<int, int> index_of_id;
    size_t max_src = 1;
    for (const Problem &p : problems) {
        index_of_id.emplace(p.ref_image_id, (int)ids.size());
        ids.push_back(p.ref_image_id);
        max_src = std::max(max_src, p.src_image_ids.size());
    }
    for (const Problem &p : problems) {
        for (int s : p.src_image_ids) {
            if (index_of_id.emplace(s, (int)ids.size()).second) {
                ids.push_back(s);
            }
        }
    }
    if (max_src + 1 > MAX_IMAGES) {
        fprintf(stderr, "Can't process so much images: %zu\n", max_src + 1);  // APD.cpp:428-431
        return EXIT_FAILURE;
    }
    const int N = (int)ids.size();
    std::vector<Mat> full(N);
    std::vector<Camera> cams0(N);
    std::vector<int> failed(N, 0);
    ParallelFor((size_t)N, [&](size_t i) {
        memset(&cams0[i], 0, sizeof(Camera));
        if (!ReadGrayImageShared(opt.dense_folder / "images" / ToFormatIndex(ids[i]), full[i]) ||
            !ReadCamera(opt.dense_folder / "cams" / (ToFormatIndex(ids[i]) + "_cam.txt"), cams0[i])) {
            failed[i] = 1;
        }
    });
    for (int i = 0; i < N; ++i) {
        if (failed[i] || full[i].cols != full[0].cols || full[i].rows != full[0].rows) {
            fprintf(stderr, "Images may error, check it! (image %d)\n", ids[i]);  // main.cpp:158
            return EXIT_FAILURE;
        }
    }
    ms_load = stage.lap();
    const int W0 = full[0].cols, H0 = full[0].rows;
    const size_t pix0 = (size_t)W0 * H0;
    const int G = (int)devices.size();
    // Views in flight per rank.  A list that names a device more than once (0,0,0: the rank-count test of a one-GPU box) is
    // taken as given, one view per rank; --ranks N: exactly N per rank.
    bool distinct = true;
    for (size_t i = 0; i < devices.size(); ++i) {
        for (size_t j = 0; j < i; ++j) {
            distinct = distinct && devices[i] != devices[j];
        }
    }
    const int round_num = opt.single_level ? 1 : RoundNum(W0, H0);
    // per pyramid level: a coarse level's launches are small, more of its views fit the device side by side
    auto lanes_at = [&](size_t level_pixels) {
        const int want = opt.ranks_per_device > 0 ? opt.ranks_per_device : (distinct ? DefaultLanes(level_pixels) : 1);
        return std::max(1, std::min(want, (V + G - 1) / G));
    };
    const int lanes = InMemoryLanes(opt, W0, H0, V, G, distinct);
    const bool gauss_seidel = opt.in_memory;  // the reference's order of views (one rank); otherwise Jacobi over views
    bool release_before_fusion = true;
    printf("There are %d problems needed to be processed on %d rank(s), up to %d view(s) in flight per rank!\nRound nums: %d\n", V, G, lanes, round_num);
    {
        const InMemoryFit fit = TestInMemoryFit(opt, devices[0], W0, H0, N, V, G, lanes, (int)max_src);
        release_before_fusion = fit.release_before_fusion;
        if (fit.have_memory && !fit.fits) {
            fprintf(stderr, "%.1f GB of resident state against %.1f GB free on device %d: this folder does not fit the in-memory scheduler "
                            "(use --files, more devices or fewer views in flight: --ranks 1)\n", fit.need_bytes / 1e9, fit.free_bytes / 1e9, devices[0]);
            return kExitDoesNotFit;   // main() falls back to the file-based loop when it had chosen this scheduler by itself
        }
    }

    // colour images, cameras and masks of the fusion: decoded and uploaded to rank 0's device behind the passes
    // (four decode threads: the lanes' host threads must stay prompt with their launches)
    FusionPrefetch *fusion_inputs = (opt.no_fusion || opt.late_fusion_inputs) ? nullptr : StartFusionInputs(opt.dense_folder, problems, devices[0], W0, H0, 4);
    Failure failure;
    std::vector<Rank> ranks(G);
    apd_exchange_t exchange = nullptr;
    std::vector<FinalMaps> maps;  // host copies, only with --keep-maps
    // final maps on the fusion device (rank 0's): per view depth, normal (3 floats), weak
    std::vector<DeviceBuffer> fuse_depth(V), fuse_normal(V);
    std::vector<const uint8_t *> fuse_weak(V, nullptr);
    DeviceBuffer final_planes0, final_weak0;
    int LW = 0, LH = 0;
    try {
        // ---- ranks ----
        const int slots = (V + G - 1) / G;  // views per rank, padded
        for (int r = 0; r < G; ++r) {
            Rank &k = ranks[r];
            k.device = devices[r];
            k.needs.assign(N, 0);
            for (int v = r; v < V; v += G) {
                k.own.push_back(v);
                k.needs[v] = 1;
                for (int s : problems[v].src_image_ids) {
                    k.needs[index_of_id.at(s)] = 1;
                }
            }
            k.images.assign(N, nullptr);
            k.send.alloc(k.device, (size_t)slots * pix0 * sizeof(float));
            k.recv.alloc(k.device, (size_t)G * slots * pix0 * sizeof(float));
            k.zero_depth.alloc(k.device, pix0 * sizeof(float));
            Check(apd_device_memset(k.device, k.zero_depth.p, 0, pix0 * sizeof(float)), "apd_device_memset");
            k.lanes.resize(lanes);
            for (Lane &l : k.lanes) {
                l.scratch_planes.alloc(k.device, pix0 * 16);
                l.scratch_weak.alloc(k.device, pix0);
                l.scratch_views.alloc(k.device, pix0 * 4);
            }
            for (int v : k.own) {
                ResidentView &s = k.state[v];
                s.planes.alloc(k.device, pix0 * 16);
                s.weak.alloc(k.device, pix0);
                s.views.alloc(k.device, pix0 * 4);
            }
        }
        const long long ms_alloc = stage.lap();
        // RCCL's set-up costs seconds (dlopen of librccl 5.0 s from a cold page cache / 1.0 s warm, ncclCommInitAll 0.65 s for one device:
        // profiles/r05/rccl_init_time.txt) and runs here, before the first pass (moving it behind the passes lost: csrc/apd_exchange.hip).
        // Ranks that share one device have nothing to send through xGMI and do without RCCL unless --rccl.
        const bool want_rccl = WantsRccl(opt);
        Check(apd_exchange_create(&exchange, G, devices.data(), want_rccl ? 1 : 0), "apd_exchange_create");
        printf("Device buffers: %lld ms, exchange set-up: %lld ms\n", ms_alloc, stage.lap());
        printf("Exchange of depth maps between passes: %s\n", apd_exchange_backend(exchange));

        auto gathered_depth = [&](const Rank &k, int v, size_t pix) {  // view v inside a gathered block (the pass before)
            return k.recv.as<float>() + ((size_t)(v % G) * slots + (size_t)(v / G)) * pix;
        };

        ms_setup = ms_alloc + stage.lap();
        int level_scale = 0;
        std::vector<Camera> cams(N);
        // which pass has published view v's depth map (Gauss-Seidel order only)
        std::vector<int> done_pass(V, -1);
        std::mutex done_m;
        std::condition_variable done_cv;
        double exchange_ms = 0.0;   // wall time of the per-pass exchanges (under done_m)
        const auto t_all = std::chrono::steady_clock::now();
        // ---- level inputs (APD.cpp:464-488), once per level: resampled on the host (one thread per image), uploaded by one thread per
        // rank, only the images a rank's views reference ----
        auto prepare_level = [&](const Pass &pass) {
            StageClock up;
            level_scale = pass.scale_size;
            const float factor = 1.0f / (float)level_scale;
            LW = level_scale == 1 ? W0 : (int)std::round(W0 * factor);
            LH = level_scale == 1 ? H0 : (int)std::round(H0 * factor);
            const float sx = LW / static_cast<float>(W0), sy = LH / static_cast<float>(H0);
            std::vector<Mat> level(N);
            const size_t level_bytes = (size_t)LW * LH * sizeof(float);
            std::vector<char> pinned(N, 0);
            ParallelFor((size_t)N, [&](size_t i) {
                if (level_scale == 1) {
                    level[i] = full[i];
                } else {
                    ResizeLinear(full[i], level[i], LW, LH);
                }
                cams[i] = cams0[i];
                if (level_scale != 1) {
                    cams[i].K[0] *= sx;
                    cams[i].K[2] *= sx;
                    cams[i].K[4] *= sy;
                    cams[i].K[5] *= sy;
                }
                cams[i].width = LW;
                cams[i].height = LH;
                if (G > 1) {  // several devices read the same host buffer: page-lock it once (a single upload gains nothing)
                    pinned[i] = apd_host_register(level[i].data(), level_bytes) == APD_OK ? 1 : 0;
                }
            });
            std::vector<std::thread> uploaders;
            for (int r = 0; r < G; ++r) {
                uploaders.emplace_back([&, r]() {
                    try {
                        Rank &k = ranks[r];
                        for (int i = 0; i < N; ++i) {
                            if (k.needs[i]) {
                                apd_image_destroy(k.images[i]);  // the level before
                                k.images[i] = nullptr;
                                Check(apd_image_create(&k.images[i], k.device, LW, LH, level[i].ptr<float>()), "apd_image_create");
                            }
                        }
                    } catch (const std::exception &e) {
                        failure.set(e.what());
                    }
                });
            }
            for (std::thread &t : uploaders) {
                t.join();
            }
            for (int i = 0; i < N; ++i) {
                if (pinned[i]) {
                    apd_host_unregister(level[i].data());
                }
            }
            if (failure.failed) {
                throw std::runtime_error(failure.what);
            }
            printf("Image size: %d * %d, %d view(s) in flight per rank\n", LW, LH, lanes_at((size_t)LW * LH));
            ms_upload += up.lap();
        };

        // ---- one (view, pass) on a lane.  depth_of(a, j): where view j's depth map for this view's slot a lies (it may block until the
        // map is published; nullptr: the run has failed); depth_out: where this view's new depth map goes; before_export(): may block
        // until nobody reads what the export overwrites any more ----
        auto wait_done = [&](int view, int at_least) {  // false: the run has failed
            std::unique_lock<std::mutex> lock(done_m);
            done_cv.wait(lock, [&]() { return done_pass[view] >= at_least || failure.failed.load(); });
            return !failure.failed.load();
        };
        auto run_view = [&](int r, Lane &lane, const Pass &pass, int v, const std::function<const float *(size_t, int)> &depth_of, float *depth_out,
                            const std::function<bool()> &before_export) {
            Rank &k = ranks[r];
            Problem &problem = problems[v];  // one lane per (view, pass), and a view's passes follow one another: nobody else touches it
            Configure(problem, pass, opt);
            PatchMatchParams q = problem.params;
            q.depth_min = cams0[v].depth_min * 0.6f;   // APD.cpp:454-455
            q.depth_max = cams0[v].depth_max * 1.2f;
            std::vector<int> order{v};
            for (int s : problem.src_image_ids) {
                order.push_back(index_of_id.at(s));
            }
            q.num_images = (int)order.size();
            const apd_params p = ToAbi(q);
            if (!lane.handle || lane.handle_w != LW || lane.handle_h != LH) {
                if (lane.handle) {
                    {
                        std::lock_guard<std::mutex> lock(done_m);
                        lane.export_event = nullptr;   // the event goes with the handle
                    }
                    apd_destroy(lane.handle);
                    lane.handle = nullptr;
                }
                Check(apd_create(&lane.handle, k.device, LW, LH, &p), "apd_create");
                lane.handle_w = LW;
                lane.handle_h = LH;
            } else {
                Check(apd_reset(lane.handle, &p), "apd_reset");
            }
            void *stream = nullptr;
            Check(apd_get_stream(lane.handle, &stream), "apd_get_stream");
            std::vector<Camera> vc;
            std::vector<apd_image_t> img;
            for (int j : order) {
                vc.push_back(cams[j]);
                img.push_back(k.images[j]);
            }
            if (opt.copy_images) {  // A/B: every handle copies, tests and packs its images itself
                std::vector<const float *> raw;
                for (apd_image_t im : img) {
                    raw.push_back(apd_image_pixels(im));
                }
                Check(apd_upload_views_split(lane.handle, (int)order.size(), vc.data(), raw.data()), "apd_upload_views_split");
            } else {
                Check(apd_upload_views_shared(lane.handle, (int)order.size(), vc.data(), img.data()), "apd_upload_views_shared");
            }
            ResidentView &s = k.state[v];
            if (pass.state != FIRST_INIT) {  // prior state of the previous pass (APD.cpp:552-581), resampled if the level changed
                if (!s.valid) {
                    throw std::runtime_error("view " + std::to_string(problem.ref_image_id) + " has no state of a previous pass");
                }
                if (s.W != LW || s.H != LH) {  // on the lane's stream, ahead of the upload that reads the result
                    Check(apd_rescale_nearest_async(k.device, stream, s.planes.p, s.W, s.H, lane.scratch_planes.p, LW, LH, 16), "rescale planes");
                    Check(apd_rescale_nearest_async(k.device, stream, s.weak.p, s.W, s.H, lane.scratch_weak.p, LW, LH, 1), "rescale weak");
                    Check(apd_rescale_nearest_async(k.device, stream, s.views.p, s.W, s.H, lane.scratch_views.p, LW, LH, 4), "rescale views");
                    std::swap(s.planes.p, lane.scratch_planes.p);
                    std::swap(s.weak.p, lane.scratch_weak.p);
                    std::swap(s.views.p, lane.scratch_views.p);
                    s.W = LW;
                    s.H = LH;
                }
                Check(apd_upload_prior(lane.handle, s.planes.as<float>(), s.views.as<uint32_t>(), pass.use_APD ? s.weak.as<uint8_t>() : nullptr),
                      "apd_upload_prior");
            }
            Check(apd_run_before_depths(lane.handle), "apd_run_before_depths");
            if (pass.geom_consistency) {
                std::vector<const float *> dep;
                for (size_t a = 0; a < order.size(); ++a) {
                    const float *d = order[a] >= V ? k.zero_depth.as<float>() : depth_of(a, order[a]);
                    if (!d) {
                        return;  // the run has failed elsewhere
                    }
                    dep.push_back(d);
                }
                Check(apd_upload_depths(lane.handle, (int)dep.size(), dep.data()), "apd_upload_depths");
            }
            Check(apd_run_after_depths(lane.handle), "apd_run_after_depths");
            if (!before_export()) {
                return;
            }
            Check(apd_export_state_device(lane.handle, s.planes.as<float>(), s.weak.as<uint8_t>(), s.views.as<uint32_t>(), depth_out),
                  "apd_export_state_device");
            s.W = LW;
            s.H = LH;
            s.valid = true;
            void *exported = nullptr;
            Check(apd_export_event(lane.handle, &exported), "apd_export_event");
            {
                std::lock_guard<std::mutex> lock(done_m);
                lane.export_event = exported;
                done_pass[v] = pass.iteration;
                printf("pass %d (round %d, scale %d) view %08d done on rank %d (device %d)\n", pass.iteration, pass.level, pass.scale_size,
                       problem.ref_image_id, r, k.device);
            }
            done_cv.notify_all();
        };

        const std::vector<Pass> plan = BuildSchedule(round_num, opt.single_level);
        // One rank: no collective to wait for, so the passes of a level need no barrier between them either.  A (view, pass) task
        // needs exactly what it reads: its own previous pass; in a geometric pass the maps of its sources -- this pass's for the
        // sources that precede it and the previous pass's for the others in the reference's order, the previous pass's for all
        // with --jacobi; and nobody may still read the two-passes-old depth map it overwrites.  View 0 of pass p + 1 starts while
        // the last views of pass p are still in their second halves: the chains of consecutive passes overlap.  Depth maps live in
        // two versions by pass parity.
        const bool wavefront = G == 1 && !opt.force_rccl;
        std::vector<std::vector<int>> readers(V);  // readers[u]: the views that list u as a source
        for (int v = 0; v < V; ++v) {
            for (int s_id : problems[v].src_image_ids) {
                const int u = index_of_id.at(s_id);
                if (u < V) {
                    readers[u].push_back(v);
                }
            }
        }
        for (size_t first = 0; first < plan.size();) {
            size_t last = first;  // passes [first, last) run at one level
            while (last < plan.size() && plan[last].scale_size == plan[first].scale_size) {
                ++last;
            }
            {   // the fusion's inputs are prepared behind the passes: if that has failed, stop now, not after every pass has run
                std::string why;
                if (FusionInputsFailed(fusion_inputs, &why)) {
                    throw std::runtime_error(why.empty() ? "fusion inputs could not be prepared" : why);
                }
            }
            if (plan[first].scale_size != level_scale) {
                prepare_level(plan[first]);
            }
            const size_t pix = (size_t)LW * LH;
            const int level_lanes = lanes_at(pix);
            if (wavefront) {
                Rank &k = ranks[0];
                float *const version[2] = {k.send.as<float>(), k.recv.as<float>()};
                const int P = (int)(last - first);
                const int first_iteration = plan[first].iteration;
                std::vector<int> remaining(P, V), frontier(P, 0), active(P, 0);
                // Which task a free lane takes (under done_m).  Within a pass the views go out in order; the earliest pass that has
                // an eligible view wins.  Eligible: the view's own previous pass is done, the maps of the previous pass it will read
                // are published and the last readers of the map it will overwrite are done (so that a running task only ever waits for
                // views of its OWN pass that went out before it: no wait can point at a task nobody holds), and -- in the reference's
                // order, where the second halves of a geometric pass form a chain -- at most two lanes work on one pass: a third would
                // only queue up behind the chain, while the next pass can already start its first views.  The smallest unfinished
                // (pass, view) is always eligible or running, so the level drains.
                auto eligible = [&](int pi, int v) {
                    const Pass &pass = plan[first + (size_t)pi];
                    const int it = pass.iteration;
                    if (pi > 0 && done_pass[v] < it - 1) {
                        return false;
                    }
                    if (it - 2 >= first_iteration) {  // the export will overwrite the view's map of pass it - 2: its last readers must be done
                        for (int w : readers[v]) {
                            if (done_pass[w] < ((gauss_seidel && w > v) ? it - 2 : it - 1)) {
                                return false;
                            }
                        }
                    }
                    if (pass.geom_consistency) {
                        if (gauss_seidel && active[pi] >= APD_GS_LANES_PER_PASS) {
                            return false;
                        }
                        for (int s_id : problems[v].src_image_ids) {
                            const int u = index_of_id.at(s_id);
                            if (u < V && !(gauss_seidel && u < v) && done_pass[u] < it - 1) {
                                return false;
                            }
                        }
                    }
                    return true;
                };
                auto take_task = [&](int &pi_out, int &v_out) {  // false: nothing left (or the run has failed)
                    std::unique_lock<std::mutex> lock(done_m);
                    for (;;) {
                        if (failure.failed.load()) {
                            return false;
                        }
                        bool any_left = false;
                        for (int pi = 0; pi < P; ++pi) {
                            if (frontier[pi] >= V) {
                                continue;
                            }
                            any_left = true;
                            if (eligible(pi, frontier[pi])) {
                                pi_out = pi;
                                v_out = frontier[pi]++;
                                ++active[pi];
                                return true;
                            }
                        }
                        if (!any_left) {
                            return false;
                        }
                        done_cv.wait(lock);
                    }
                };
                std::vector<std::thread> workers;
                for (int li = 0; li < level_lanes; ++li) {
                    workers.emplace_back([&, li]() {
                        Lane &lane = k.lanes[li];
                        try {
                            int pi = 0, v = 0;
                            while (take_task(pi, v)) {
                                const Pass &pass = plan[first + (size_t)pi];
                                const int it = pass.iteration;
                                auto depth_of = [&](size_t a, int j) -> const float * {
                                    const bool this_pass = gauss_seidel && a > 0 && j < v;
                                    const int need = this_pass ? it : it - 1;
                                    if (!wait_done(j, need)) {
                                        return nullptr;
                                    }
                                    return version[need & 1] + (size_t)j * pix;
                                };
                                const auto before_export = []() { return true; };  // `eligible` has seen the old map's last readers finish
                                run_view(0, lane, pass, v, depth_of, version[it & 1] + (size_t)v * pix, before_export);
                                {
                                    std::lock_guard<std::mutex> lock(done_m);
                                    --active[pi];
                                    if (done_pass[v] == it && --remaining[pi] == 0 && it % 4 == 3) {
                                        printf("Round: %d done\n", pass.level);
                                    }
                                }
                                done_cv.notify_all();
                            }
                        } catch (const std::exception &e) {
                            failure.set(e.what());
                            done_cv.notify_all();
                        }
                    });
                }
                for (std::thread &t : workers) {
                    t.join();
                }
                if (failure.failed) {
                    throw std::runtime_error(failure.what);
                }
                fflush(stdout);
                first = last;
                continue;
            }
            // Several ranks (or --rccl): a pass ends with the all-gather of its depth maps (the reference: depths.dmb files, APD.cpp:497-500),
            // but only the halves that READ depth maps wait for it.  A rank's (pass, view) tasks of the level go out in order to its
            // `lanes` host threads (handle + stream each); a lane that finds pass p handed out starts the first views of pass p + 1 --
            // their first halves need nothing but the view's own state -- while the last views of pass p finish and their maps are
            // exchanged; the thread that finishes the last view of a pass (over all ranks) runs the exchange.  No wait can point at a task
            // nobody holds: whoever holds a task of pass p + 1 knows every task of pass p of its rank to be handed out, and a task of
            // pass p waits only for the exchange of pass p - 1, whose views were all handed out earlier still.
            {
                const int P = (int)(last - first);
                const int first_iteration = plan[first].iteration;
                std::vector<int> remaining(P, V);
                int exchanged = first_iteration - 1;  // the newest iteration whose maps every rank holds (under done_m)
                auto wait_exchange = [&](int iteration) {  // false: the run has failed
                    std::unique_lock<std::mutex> lock(done_m);
                    done_cv.wait(lock, [&]() { return exchanged >= iteration || failure.failed.load(); });
                    return !failure.failed.load();
                };
                std::vector<std::thread> workers;
                for (int r = 0; r < G; ++r) {
                    ranks[r].next.store(0);
                    for (int li = 0; li < level_lanes; ++li) {
                        workers.emplace_back([&, r, li]() {
                            Rank &k = ranks[r];
                            Lane &lane = k.lanes[li];
                            const int own = (int)k.own.size();
                            try {
                                for (;;) {
                                    const int at = k.next.fetch_add(1);
                                    if (at >= P * own || failure.failed) {
                                        break;
                                    }
                                    const int pi = at / own, v = k.own[at % own];
                                    const Pass &pass = plan[first + (size_t)pi];
                                    const int it = pass.iteration;
                                    if (pi > 0 && !wait_done(v, it - 1)) {  // the view's own previous pass (another lane may still hold it)
                                        break;
                                    }
                                    // The sources' depth maps: of this pass for the sources that precede the view in the reference's order
                                    // (they must have published; one rank only), of the pass before -- once exchanged -- for the others and
                                    // for the view itself.
                                    auto depth_of = [&](size_t a, int j) -> const float * {
                                        if (gauss_seidel && a > 0 && j < v) {
                                            return wait_done(j, it) ? k.send.as<float>() + (size_t)(j / G) * pix : nullptr;
                                        }
                                        return wait_exchange(it - 1) ? gathered_depth(k, j, pix) : nullptr;
                                    };
                                    // the export overwrites the view's block of `send`: the exchange of the pass before must have read it
                                    const auto before_export = [&]() { return pi == 0 || wait_exchange(it - 1); };
                                    run_view(r, lane, pass, v, depth_of, k.send.as<float>() + (size_t)(v / G) * pix, before_export);
                                    bool run_exchange = false;
                                    {
                                        std::lock_guard<std::mutex> lock(done_m);
                                        run_exchange = done_pass[v] == it && --remaining[pi] == 0;
                                    }
                                    if (run_exchange) {  // every view of this pass, on every rank, has exported
                                        std::vector<const void *> send(G);
                                        std::vector<void *> recv(G);
                                        for (int q = 0; q < G; ++q) {
                                            send[q] = ranks[q].send.p;
                                            recv[q] = ranks[q].recv.p;
                                        }
                                        // every view of the pass has exported; the exchange's streams wait for the export event of every
                                        // lane's handle (each marks the newest export on that lane's stream: at least this pass's) and for
                                        // nothing else -- no device-wide synchronisation, the first halves other lanes have queued for the
                                        // next pass keep running beside the exchange.  The readers of the old gathered maps were views of
                                        // this pass.
                                        std::vector<void *> exported;
                                        {
                                            std::lock_guard<std::mutex> lock(done_m);
                                            for (Rank &q : ranks) {
                                                for (Lane &l : q.lanes) {
                                                    if (l.export_event) {
                                                        exported.push_back(l.export_event);
                                                    }
                                                }
                                            }
                                        }
                                        const auto t_x = std::chrono::steady_clock::now();
                                        Check(opt.exchange_device_sync
                                                  ? apd_exchange_allgather(exchange, send.data(), recv.data(), (size_t)slots * pix * sizeof(float))
                                                  : apd_exchange_allgather_after(exchange, send.data(), recv.data(), (size_t)slots * pix * sizeof(float),
                                                                                 (int)exported.size(), exported.data()),
                                              "apd_exchange_allgather");
                                        const double x_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_x).count();
                                        {
                                            std::lock_guard<std::mutex> lock(done_m);
                                            exchange_ms += x_ms;
                                            exchanged = it;
                                            if (it % 4 == 3) {
                                                printf("Round: %d done\n", pass.level);
                                            }
                                        }
                                        done_cv.notify_all();
                                    }
                                }
                            } catch (const std::exception &e) {
                                failure.set(e.what());
                                done_cv.notify_all();
                            }
                        });
                    }
                }
                for (std::thread &t : workers) {
                    t.join();
                }
                if (failure.failed) {
                    throw std::runtime_error(failure.what);
                }
                fflush(stdout);
            }
            first = last;
        }
        ms_passes = stage.lap() - ms_upload;

        // ---- before fusion: planes (world normal + depth) and weak maps of all views on every rank, view by view ----
        const size_t pix = (size_t)LW * LH;
        if (release_before_fusion) {
            for (Rank &k : ranks) {
                for (Lane &l : k.lanes) {  // handles, level images and depth blocks are no longer needed: room for the final maps
                    if (l.handle) {
                        apd_destroy(l.handle);
                        l.handle = nullptr;
                        l.export_event = nullptr;
                    }
                    l.scratch_planes.release();
                    l.scratch_weak.release();
                    l.scratch_views.release();
                }
                for (apd_image_t &im : k.images) {
                    apd_image_destroy(im);
                    im = nullptr;
                }
                k.send.release();
                k.recv.release();
                k.zero_depth.release();
            }
        }
        std::vector<const float *> planes_of(V, nullptr);  // on rank 0's device
        if (G == 1) {  // one rank: everything is where the fusion runs
            for (int v = 0; v < V; ++v) {
                planes_of[v] = ranks[0].state[v].planes.as<float>();
                fuse_weak[v] = ranks[0].state[v].weak.as<uint8_t>();
            }
        } else {
            // One all-gather per slot straight out of the views' state buffers into [slot][rank] blocks, i.e. view order
            // (view v = slot * G + rank); a rank without a view in the last slot sends its first view's buffer as padding.
            std::vector<DeviceBuffer> all_planes(G), all_weak(G), padding(G);
            for (int r = 0; r < G; ++r) {
                all_planes[r].alloc(ranks[r].device, (size_t)slots * G * pix * 16);
                all_weak[r].alloc(ranks[r].device, (size_t)slots * G * pix);
                if ((int)ranks[r].own.size() < slots) {
                    padding[r].alloc(ranks[r].device, pix * 16);
                }
            }
            std::vector<const void *> send(G);
            std::vector<void *> recv(G);
            for (int sl = 0; sl < slots; ++sl) {
                for (int pass_kind = 0; pass_kind < 2; ++pass_kind) {
                    const size_t elem = pass_kind == 0 ? 16 : 1;
                    for (int r = 0; r < G; ++r) {
                        Rank &k = ranks[r];
                        if ((size_t)sl < k.own.size()) {
                            const ResidentView &s = k.state[k.own[sl]];
                            send[r] = pass_kind == 0 ? s.planes.p : s.weak.p;
                        } else {
                            send[r] = padding[r].p;
                        }
                        recv[r] = (pass_kind == 0 ? all_planes[r].as<char>() : all_weak[r].as<char>()) + (size_t)sl * G * pix * elem;
                    }
                    Check(apd_exchange_allgather(exchange, send.data(), recv.data(), pix * elem), "apd_exchange_allgather (final maps)");
                }
            }
            for (int v = 0; v < V; ++v) {
                planes_of[v] = all_planes[0].as<float>() + (size_t)v * pix * 4;
                fuse_weak[v] = all_weak[0].as<uint8_t>() + (size_t)v * pix;
            }
            final_planes0 = all_planes[0];
            final_weak0 = all_weak[0];
            all_planes[0].p = all_weak[0].p = nullptr;  // kept for the fusion; the other ranks' copies are done with
            for (int r = 0; r < G; ++r) {
                all_planes[r].release();
                all_weak[r].release();
                padding[r].release();
            }
        }
        for (int v = 0; v < V; ++v) {  // depth + normal maps as the fusion takes them (main.cpp:105-124 writes the same split)
            fuse_depth[v].alloc(ranks[0].device, pix * 4);
            fuse_normal[v].alloc(ranks[0].device, pix * 12);
            Check(apd_split_planes_async(ranks[0].device, nullptr, planes_of[v], pix, fuse_depth[v].as<float>(), fuse_normal[v].as<float>()), "apd_split_planes_async");
        }
        Check(apd_stream_synchronize(ranks[0].device, nullptr), "apd_stream_synchronize");
        if (opt.keep_maps) {  // the four files of ProcessProblem (main.cpp:117-124)
            maps.resize(V);
            for (int v = 0; v < V; ++v) {
                Rank &k = ranks[v % G];
                FinalMaps &m = maps[v];
                m.depth.create(LH, LW, MAT_32FC1);
                m.normal.create(LH, LW, MAT_32FC3);
                m.weak.create(LH, LW, MAT_8UC1);
                Mat views(LH, LW, MAT_32SC1);
                Check(apd_device_memcpy(ranks[0].device, m.depth.data(), fuse_depth[v].p, pix * 4), "download depth");
                Check(apd_device_memcpy(ranks[0].device, m.normal.data(), fuse_normal[v].p, pix * 12), "download normals");
                Check(apd_device_memcpy(ranks[0].device, m.weak.data(), fuse_weak[v], pix), "download weak");
                Check(apd_device_memcpy(k.device, views.data(), k.state[v].views.p, pix * 4), "download views");
                std::filesystem::create_directories(problems[v].result_folder);
                const Mat *out[4] = {&m.depth, &m.normal, &m.weak, &views};
                for (int f = 0; f < 4; ++f) {
                    if (!WriteBinMat(problems[v].result_folder / kStateFiles[f], *out[f])) {
                        throw std::runtime_error("cannot write " + (problems[v].result_folder / kStateFiles[f]).string());
                    }
                }
            }
        }
        {
            int with_rccl = 0, with_copies = 0;
            apd_exchange_counts(exchange, &with_rccl, &with_copies);
            double dl = 0, init = 0;
            apd_exchange_setup_times(exchange, &dl, &init);
            printf("Exchanges: %d through RCCL, %d through direct copies, the per-pass ones took %.0f ms in all; RCCL set-up: dlopen %.0f ms, communicators "
                   "%.0f ms (incl. the dlopen when this exchange was the process's first); backend %s\n", with_rccl, with_copies, exchange_ms, dl, init,
                   apd_exchange_backend(exchange));
        }
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_all).count();
        printf("All passes done: %lld ms\n", (long long)ms);
        ms_gather = stage.lap();
        if (!opt.no_fusion) {
            std::filesystem::create_directories(opt.dense_folder / "APD");
            std::vector<const float *> d(V), n(V);
            for (int v = 0; v < V; ++v) {
                d[v] = fuse_depth[v].as<float>();
                n[v] = fuse_normal[v].as<float>();
            }
            if (LW != W0 || LH != H0) {
                throw std::runtime_error("the last pass did not run at the full resolution");  // BuildSchedule ends at scale 1
            }
            if (!fusion_inputs) {
                fusion_inputs = StartFusionInputs(opt.dense_folder, problems, devices[0], W0, H0, 0);
            }
            RunFusionOnDevice(fusion_inputs, d, n, fuse_weak);
            fusion_inputs = nullptr;
        }
        ms_fusion = stage.lap();
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        fflush(stderr);
        CancelFusionInputs(fusion_inputs);
        // device memory goes with the process (the reference exits at the failing call, APD.cpp:315-323)
        return EXIT_FAILURE;
    }
    // The views' state, the final maps and (when there was room to keep them) the handles and level images stay allocated: this is the
    // program's last act, main() leaves through _Exit, and the driver reclaims a process's device memory in one step -- released one
    // hipFree at a time (each a device synchronisation) 152 views cost a third of a second.
    apd_exchange_destroy(exchange);
    printf("Stages: images + cameras %lld ms, device set-up %lld ms, level images (resample + upload) %lld ms, passes %lld ms, final gather + maps %lld ms, "
           "fusion %lld ms\n", ms_load, ms_setup, ms_upload, ms_passes, ms_gather, ms_fusion);
    printf("All done\n");
    return EXIT_SUCCESS;
}
