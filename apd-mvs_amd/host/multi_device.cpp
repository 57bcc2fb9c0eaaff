// multi_device.cpp -- `APD dense_folder 0,1,2,3`: the pass table of the reference driver (main.cpp:168-215) on several
// devices of one node, in memory.
//
// The reference takes one device index (main.cpp:149-153), processes the views one after the other and hands state from
// pass to pass through four files per view; in a geometric pass a view reads its sources' depths.dmb as they are at that
// moment.  Here:
//   * rank r of the device list owns the reference views r, r + G, r + 2G, ... (round-robin: neighbouring views are usually
//     each other's sources) and processes them on its own host thread, through its own handle and stream;
//   * planes, weak map and selected views of a view stay on its rank's device from pass to pass (apd_export_state_device ->
//     apd_upload_prior, device to device); the nearest-neighbour resampling between pyramid levels runs on the device too;
//   * after every pass the depth maps of all views are all-gathered (apd_exchange_allgather: RCCL over xGMI, or direct peer
//     copies), which replaces the exchange through depths.dmb; after the last pass the planes (normal + depth) and weak
//     maps are all-gathered the same way and rank 0's copy goes to the fusion;
//   * every view of a pass reads the depth maps of the PREVIOUS pass (Jacobi over views; the file-based driver is
//     Gauss-Seidel), so the result does not depend on the number of ranks: `APD folder 0 --jacobi` and `APD folder 0,0,0`
//     write the same bytes (tests/test_gpu_dropin_binary.py), while it differs slightly, by construction, from the
//     single-device file-based order.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

#include "APD.h"
#include "schedule.h"

namespace {

void Check(int rc, const char *what)
{
    if (rc != APD_OK) {  // reference: CudaSafeCall -> print + exit (APD.cpp:315-323)
        const char *a = apd_last_error(), *b = apd_exchange_last_error();
        fprintf(stderr, "%s failed: %s%s%s\n", what, a ? a : "", (b && b[0]) ? " / " : "", (b && b[0]) ? b : "");
        exit(EXIT_FAILURE);
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

struct Rank {
    int device = 0;
    std::vector<int> own;            // view indices, ascending
    apd_handle handle = nullptr;
    int handle_w = 0, handle_h = 0;
    std::vector<DeviceBuffer> images;    // level image of every loaded view
    DeviceBuffer send, recv;         // depth blocks of the all-gather (float)
    DeviceBuffer depth_level;        // every view's depth map resampled to the current level, when the gathered ones are coarser
    DeviceBuffer zero_depth;         // a source-only view has no estimate
    DeviceBuffer scratch_planes, scratch_weak, scratch_views;  // resampling targets
    std::unordered_map<int, ResidentView> state;
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

}  // namespace

namespace {
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
}  // namespace

int RunMultiDevice(const Options &opt, std::vector<Problem> &problems)
{
    StageClock stage;
    long long ms_load = 0, ms_setup = 0, ms_passes = 0, ms_gather = 0, ms_fusion = 0;
    std::vector<int> devices = opt.devices;
    const int V = (int)problems.size();
    if (devices.empty() || V < 1) {
        fprintf(stderr, "nothing to do\n");
        return EXIT_FAILURE;
    }
    // ---- views: the reference views in pair.txt order, then the source-only images (main.cpp) ----
    std::vector<int> ids;
    std::unordered_map<int, int> index_of_id;
    for (const Problem &p : problems) {
        index_of_id.emplace(p.ref_image_id, (int)ids.size());
        ids.push_back(p.ref_image_id);
    }
    for (const Problem &p : problems) {
        for (int s : p.src_image_ids) {
            if (index_of_id.emplace(s, (int)ids.size()).second) {
                ids.push_back(s);
            }
        }
    }
    const int N = (int)ids.size();
    std::vector<Mat> full(N);
    std::vector<Camera> cams0(N);
    std::vector<int> failed(N, 0);
    ParallelFor((size_t)N, [&](size_t i) {
        memset(&cams0[i], 0, sizeof(Camera));
        if (!ReadGrayImage(opt.dense_folder / "images" / ToFormatIndex(ids[i]), full[i]) ||
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
    // A device takes several scheduler ranks when the frames are small: one view's launches leave a 256-CU device partly idle (a
    // 960 x 540 level is 1.3 rounds of workgroups), two or three views in flight fill it -- 12 % less wall time on a 12-view 1080p
    // folder, 2 % at 6200 x 4130 (profiles/r03/e2e_timing.txt), same bytes (Jacobi over views).  --ranks N: exactly N per device.
    // The device list is repeated as a whole ("0,1,2,3" -> "0,1,2,3,0,1,2,3"): views stay round-robin over the devices, and the
    // exchange runs RCCL between one leader rank per device and copies inside the devices (csrc/apd_exchange.hip).  A list that
    // already names a device twice is taken as given.
    if (!opt.in_memory) {
        bool distinct = true;
        for (size_t i = 0; i < devices.size(); ++i) {
            for (size_t j = 0; j < i; ++j) {
                distinct = distinct && devices[i] != devices[j];
            }
        }
        int k = opt.ranks_per_device;
        if (k <= 0) {
            k = pix0 <= ((size_t)4 << 20) ? 3 : (pix0 <= ((size_t)12 << 20) ? 2 : 1);
        }
        k = std::max(1, std::min(k, V / (int)devices.size()));
        if (distinct && k > 1) {
            const std::vector<int> once = devices;
            for (int rep = 1; rep < k; ++rep) {
                devices.insert(devices.end(), once.begin(), once.end());
            }
        }
    }
    const int G = (int)devices.size();
    const int round_num = opt.single_level ? 1 : RoundNum(W0, H0);
    printf("There are %d problems needed to be processed on %d rank(s)!\nRound nums: %d\n", V, G, round_num);

    // ---- ranks ----
    const int slots = (V + G - 1) / G;  // views per rank, padded
    std::vector<Rank> ranks(G);
    for (int r = 0; r < G; ++r) {
        Rank &k = ranks[r];
        k.device = devices[r];
        for (int v = r; v < V; v += G) {
            k.own.push_back(v);
        }
        k.images.resize(N);
        for (int i = 0; i < N; ++i) {
            k.images[i].alloc(k.device, pix0 * sizeof(float));
        }
        k.send.alloc(k.device, (size_t)slots * pix0 * sizeof(float));
        k.recv.alloc(k.device, (size_t)G * slots * pix0 * sizeof(float));
        k.depth_level.alloc(k.device, (size_t)V * pix0 * sizeof(float));
        k.zero_depth.alloc(k.device, pix0 * sizeof(float));
        Check(apd_device_memset(k.device, k.zero_depth.p, 0, pix0 * sizeof(float)), "apd_device_memset");
        k.scratch_planes.alloc(k.device, pix0 * 16);
        k.scratch_weak.alloc(k.device, pix0);
        k.scratch_views.alloc(k.device, pix0 * 4);
        for (int v : k.own) {
            ResidentView &s = k.state[v];
            s.planes.alloc(k.device, pix0 * 16);
            s.weak.alloc(k.device, pix0);
            s.views.alloc(k.device, pix0 * 4);
        }
    }
    const long long ms_alloc = stage.lap();
    apd_exchange_t exchange = nullptr;
    // RCCL's set-up costs seconds (5.6 s for one device on the MI355X box, against 4.0 s for all eight passes of a 12-view 1080p
    // folder) and cannot be overlapped with the passes: ranks that share one device have nothing to send through xGMI and do without
    // unless --rccl
    int physical = 0;  // distinct devices of the list
    for (int i = 0; i < G; ++i) {
        bool seen = false;
        for (int j = 0; j < i; ++j) {
            seen = seen || devices[j] == devices[i];
        }
        physical += seen ? 0 : 1;
    }
    Check(apd_exchange_create(&exchange, G, devices.data(), (opt.use_rccl && (physical > 1 || opt.force_rccl)) ? 1 : 0), "apd_exchange_create");
    printf("Device buffers: %lld ms, exchange set-up: %lld ms\n", ms_alloc, stage.lap());
    printf("Exchange of depth maps between passes: %s\n", apd_exchange_backend(exchange));

    auto block_of_view = [&](const Rank &k, int v, size_t pix) {  // view v inside a gathered block
        return k.recv.as<float>() + ((size_t)(v % G) * slots + (size_t)(v / G)) * pix;
    };

    ms_setup = ms_alloc + stage.lap();
    int level_scale = 0, LW = 0, LH = 0;   // current level
    int gathered_w = 0, gathered_h = 0;    // size of the depth maps in `recv`
    std::vector<Camera> cams(N);
    const auto t_all = std::chrono::steady_clock::now();
    for (const Pass &pass : BuildSchedule(round_num, opt.single_level)) {
        // ---- level inputs (APD.cpp:464-488), once per level: resampled on the host, uploaded to every rank ----
        if (pass.scale_size != level_scale) {
            level_scale = pass.scale_size;
            const float factor = 1.0f / (float)level_scale;
            LW = level_scale == 1 ? W0 : (int)std::round(W0 * factor);
            LH = level_scale == 1 ? H0 : (int)std::round(H0 * factor);
            const float sx = LW / static_cast<float>(W0), sy = LH / static_cast<float>(H0);
            std::vector<Mat> level(N);
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
            });
            for (Rank &k : ranks) {
                for (int i = 0; i < N; ++i) {
                    Check(apd_device_memcpy(k.device, k.images[i].p, level[i].ptr<float>(), (size_t)LW * LH * sizeof(float)), "image upload");
                }
            }
            printf("Image size: %d * %d\n", LW, LH);
        }
        const size_t pix = (size_t)LW * LH;
        const bool resample_depths = pass.geom_consistency && (gathered_w != LW || gathered_h != LH);

        // ---- one host thread per rank ----
        std::vector<std::thread> workers;
        for (int r = 0; r < G; ++r) {
            workers.emplace_back([&, r]() {
                Rank &k = ranks[r];
                if (resample_depths) {  // the gathered maps are one level coarser (RescaleMatToTargetSize, APD.cpp:503-507)
                    for (int v = 0; v < V; ++v) {
                        Check(apd_rescale_nearest_device(k.device, block_of_view(k, v, (size_t)gathered_w * gathered_h), gathered_w, gathered_h,
                                                         k.depth_level.as<float>() + (size_t)v * pix, LW, LH, 4),
                              "apd_rescale_nearest_device");
                    }
                }
                for (int v : k.own) {
                    Problem &problem = problems[v];
                    Configure(problem, pass, opt);
                    PatchMatchParams q = problem.params;
                    q.depth_min = cams0[v].depth_min * 0.6f;   // APD.cpp:454-455
                    q.depth_max = cams0[v].depth_max * 1.2f;
                    std::vector<int> order{v};
                    for (int s : problem.src_image_ids) {
                        order.push_back(index_of_id.at(s));
                    }
                    if (order.size() > MAX_IMAGES) {
                        fprintf(stderr, "Can't process so much images: %zu\n", order.size());
                        exit(EXIT_FAILURE);
                    }
                    q.num_images = (int)order.size();
                    const apd_params p = ToAbi(q);
                    if (!k.handle || k.handle_w != LW || k.handle_h != LH) {
                        if (k.handle) {
                            apd_destroy(k.handle);
                        }
                        Check(apd_create(&k.handle, k.device, LW, LH, &p), "apd_create");
                        k.handle_w = LW;
                        k.handle_h = LH;
                    } else {
                        Check(apd_reset(k.handle, &p), "apd_reset");
                    }
                    std::vector<Camera> vc;
                    std::vector<const float *> img, dep;
                    for (int j : order) {
                        vc.push_back(cams[j]);
                        img.push_back(k.images[j].as<float>());
                        if (pass.geom_consistency) {
                            if (j >= V) {
                                dep.push_back(k.zero_depth.as<float>());
                            } else if (resample_depths) {
                                dep.push_back(k.depth_level.as<float>() + (size_t)j * pix);
                            } else {
                                dep.push_back(block_of_view(k, j, pix));
                            }
                        }
                    }
                    Check(apd_upload_views(k.handle, (int)order.size(), vc.data(), img.data(), pass.geom_consistency ? dep.data() : nullptr),
                          "apd_upload_views");
                    ResidentView &s = k.state[v];
                    if (pass.state != FIRST_INIT) {  // prior state of the previous pass (APD.cpp:552-581), resampled if the level changed
                        if (!s.valid) {
                            fprintf(stderr, "view %d has no state of a previous pass\n", problem.ref_image_id);
                            exit(EXIT_FAILURE);
                        }
                        if (s.W != LW || s.H != LH) {
                            Check(apd_rescale_nearest_device(k.device, s.planes.p, s.W, s.H, k.scratch_planes.p, LW, LH, 16), "rescale planes");
                            Check(apd_rescale_nearest_device(k.device, s.weak.p, s.W, s.H, k.scratch_weak.p, LW, LH, 1), "rescale weak");
                            Check(apd_rescale_nearest_device(k.device, s.views.p, s.W, s.H, k.scratch_views.p, LW, LH, 4), "rescale views");
                            std::swap(s.planes.p, k.scratch_planes.p);
                            std::swap(s.weak.p, k.scratch_weak.p);
                            std::swap(s.views.p, k.scratch_views.p);
                            s.W = LW;
                            s.H = LH;
                        }
                        Check(apd_upload_prior(k.handle, s.planes.as<float>(), s.views.as<uint32_t>(), pass.use_APD ? s.weak.as<uint8_t>() : nullptr),
                              "apd_upload_prior");
                    }
                    Check(apd_run(k.handle), "apd_run");
                    const size_t slot = (size_t)(v / G);
                    Check(apd_export_state_device(k.handle, s.planes.as<float>(), s.weak.as<uint8_t>(), s.views.as<uint32_t>(),
                                                  k.send.as<float>() + slot * pix),
                          "apd_export_state_device");
                    s.W = LW;
                    s.H = LH;
                    s.valid = true;
                    if (opt.in_memory) {
                        // the reference's order (Gauss-Seidel over views): ProcessProblem writes depths.dmb before the next view of
                        // the pass reads its sources' (main.cpp:117-124, APD.cpp:497-500) -- the view's new map replaces the
                        // gathered one at once (one rank: G == 1, slot == v)
                        Check(apd_device_memcpy(k.device, block_of_view(k, v, pix), k.send.as<float>() + slot * pix, pix * sizeof(float)),
                              "publish depth");
                    }
                    printf("pass %d (round %d, scale %d) view %08d done on rank %d (device %d)\n", pass.iteration, pass.level, pass.scale_size,
                           problem.ref_image_id, r, k.device);
                }
            });
        }
        for (std::thread &t : workers) {
            t.join();
        }
        // ---- every rank gets every view's depth map (the reference: depths.dmb files, APD.cpp:497-500) ----
        std::vector<const void *> send(G);
        std::vector<void *> recv(G);
        for (int r = 0; r < G; ++r) {
            send[r] = ranks[r].send.p;
            recv[r] = ranks[r].recv.p;
        }
        Check(apd_exchange_allgather(exchange, send.data(), recv.data(), (size_t)slots * pix * sizeof(float)), "apd_exchange_allgather");
        gathered_w = LW;
        gathered_h = LH;
        if (pass.iteration % 4 == 3) {
            printf("Round: %d done\n", pass.level);
        }
        fflush(stdout);
    }

    ms_passes = stage.lap();
    // ---- before fusion: planes (world normal + depth) and weak maps of all views on every rank ----
    const size_t pix = (size_t)LW * LH;
    std::vector<FinalMaps> maps(V);
    {
        std::vector<DeviceBuffer> send_pl(G), recv_pl(G), send_wk(G), recv_wk(G);
        for (int r = 0; r < G; ++r) {
            Rank &k = ranks[r];
            for (DeviceBuffer &b : k.images) {
                b.release();
            }
            k.depth_level.release();
            k.send.release();
            k.recv.release();
            send_pl[r].alloc(k.device, (size_t)slots * pix * 16);
            recv_pl[r].alloc(k.device, (size_t)G * slots * pix * 16);
            send_wk[r].alloc(k.device, (size_t)slots * pix);
            recv_wk[r].alloc(k.device, (size_t)G * slots * pix);
            for (int v : k.own) {
                const ResidentView &s = k.state[v];
                Check(apd_device_memcpy(k.device, send_pl[r].as<char>() + (size_t)(v / G) * pix * 16, s.planes.p, pix * 16), "pack planes");
                Check(apd_device_memcpy(k.device, send_wk[r].as<char>() + (size_t)(v / G) * pix, s.weak.p, pix), "pack weak");
            }
        }
        std::vector<const void *> send(G);
        std::vector<void *> recv(G);
        for (int r = 0; r < G; ++r) {
            send[r] = send_pl[r].p;
            recv[r] = recv_pl[r].p;
        }
        Check(apd_exchange_allgather(exchange, send.data(), recv.data(), (size_t)slots * pix * 16), "apd_exchange_allgather (planes)");
        for (int r = 0; r < G; ++r) {
            send[r] = send_wk[r].p;
            recv[r] = recv_wk[r].p;
        }
        Check(apd_exchange_allgather(exchange, send.data(), recv.data(), (size_t)slots * pix), "apd_exchange_allgather (weak)");
        // rank 0's copy goes to the host: depth / normal / weak as ProcessProblem writes them (main.cpp:105-124)
        std::vector<float> planes(pix * 4);
        for (int v = 0; v < V; ++v) {
            const size_t at = (size_t)(v % G) * slots + (size_t)(v / G);
            Check(apd_device_memcpy(ranks[0].device, planes.data(), recv_pl[0].as<char>() + at * pix * 16, pix * 16), "download planes");
            FinalMaps &m = maps[v];
            m.depth.create(LH, LW, MAT_32FC1);
            m.normal.create(LH, LW, MAT_32FC3);
            m.weak.create(LH, LW, MAT_8UC1);
            Check(apd_device_memcpy(ranks[0].device, m.weak.data(), recv_wk[0].as<char>() + at * pix, pix), "download weak");
            ParallelFor((size_t)LH, [&](size_t row) {
                float *d = m.depth.ptr<float>((int)row);
                Vec3f *n = m.normal.ptr<Vec3f>((int)row);
                const float *src = planes.data() + row * (size_t)LW * 4;
                for (int c = 0; c < LW; ++c) {
                    n[c] = Vec3f{{src[4 * c], src[4 * c + 1], src[4 * c + 2]}};
                    d[c] = src[4 * c + 3];
                }
            });
        }
        for (int r = 0; r < G; ++r) {
            send_pl[r].release();
            recv_pl[r].release();
            send_wk[r].release();
            recv_wk[r].release();
        }
    }
    if (opt.keep_maps) {  // the four files of ProcessProblem (main.cpp:117-124)
        for (int v = 0; v < V; ++v) {
            Rank &k = ranks[v % G];
            Mat views(LH, LW, MAT_32SC1);
            Check(apd_device_memcpy(k.device, views.data(), k.state[v].views.p, pix * 4), "download views");
            std::filesystem::create_directories(problems[v].result_folder);
            const Mat *out[4] = {&maps[v].depth, &maps[v].normal, &maps[v].weak, &views};
            for (int f = 0; f < 4; ++f) {
                if (!WriteBinMat(problems[v].result_folder / kStateFiles[f], *out[f])) {
                    fprintf(stderr, "cannot write %s\n", (problems[v].result_folder / kStateFiles[f]).string().c_str());
                    return EXIT_FAILURE;
                }
            }
        }
    }
    for (Rank &k : ranks) {
        if (k.handle) {
            apd_destroy(k.handle);
        }
        for (auto &kv : k.state) {
            kv.second.planes.release();
            kv.second.weak.release();
            kv.second.views.release();
        }
        k.zero_depth.release();
        k.scratch_planes.release();
        k.scratch_weak.release();
        k.scratch_views.release();
    }
    {
        int with_rccl = 0, with_copies = 0;
        apd_exchange_counts(exchange, &with_rccl, &with_copies);
        printf("Exchanges: %d through RCCL, %d through direct copies\n", with_rccl, with_copies);
    }
    apd_exchange_destroy(exchange);
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_all).count();
    printf("All passes done: %lld ms\n", (long long)ms);
    ms_gather = stage.lap();
    if (!opt.no_fusion) {
        std::filesystem::create_directories(opt.dense_folder / "APD");
        RunFusionWithMaps(opt.dense_folder, problems, &maps);
    }
    ms_fusion = stage.lap();
    printf("Stages: images + cameras %lld ms, device set-up %lld ms, passes %lld ms, final gather + maps %lld ms, fusion %lld ms\n", ms_load,
           ms_setup, ms_passes, ms_gather, ms_fusion);
    printf("All done\n");
    return EXIT_SUCCESS;
}
