// fusion.cpp -- depth-map fusion of the drop-in host: what the reference's RunFusion (ETH variant, APD.cpp:826-977) and
// ExportPointCloud (APD.cpp:214-254) produce, i.e. <dense>/APD/APD.ply.
//
// The fusion itself runs on the GPU (apd_fuse_views, csrc/apd_fusion.hip); this file is the drop-in entry point
// RunFusion, which reads the maps the way the reference does and hands them over.  There is no host fallback: a failing
// device fusion ends the program like any other device error.  (The reference's sequential loop lives in
// oracle/fusion_oracle.cpp as the checker of the device fusion.)
//
// Outside the PatchMatch path proper.  Like the reference, the images are re-read in colour for the
// point colours (cv::imread(IMREAD_COLOR), APD.cpp:859; host/jpeg_gray.cpp decodes to the same BGR bytes as libjpeg).
#include <atomic>
#include <chrono>
#include <sstream>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <thread>
#include <unordered_map>

#include "APD.h"
#include "schedule.h"

namespace {

int g_fusion_device = 0;

struct FusionView {
    Camera cam;
    Mat image;     // float 0..255: 3 channels (blue, green, red, cv::imread(IMREAD_COLOR), APD.cpp:859) or 1 (grey)
    Mat depth;     // float, <= 0: no estimate
    Mat normal;    // 3 x float, world frame
    Mat weak;      // uint8 PixelState
    Mat block;     // optional uint8 mask of <dense>/blocks (APD.cpp:849-853); empty = none
};

// Device fusion through the C ABI (host pointers).
long long fuse_dispatch(std::vector<FusionView> &views, const std::vector<std::vector<int>> &sources, const path &ply_path)
{
    const int V = (int)views.size();
    std::vector<apd_camera> cams(V);
    std::vector<const float *> imgs(V), deps(V), nors(V);
    std::vector<const uint8_t *> weaks(V), blocks(V, nullptr);
    bool any_block = false;
    std::vector<int> rows(V), cols(V), offs(V + 1, 0), idx;
    for (int i = 0; i < V; ++i) {
        cams[i] = views[i].cam;
        imgs[i] = views[i].image.ptr<float>();
        deps[i] = views[i].depth.ptr<float>();
        nors[i] = views[i].normal.ptr<float>();
        weaks[i] = views[i].weak.ptr<uint8_t>();
        if (!views[i].block.empty()) {
            blocks[i] = views[i].block.ptr<uint8_t>();
            any_block = true;
        }
        rows[i] = views[i].depth.rows;
        cols[i] = views[i].depth.cols;
        idx.insert(idx.end(), sources[i].begin(), sources[i].end());
        offs[i + 1] = (int)idx.size();
    }
    if (idx.empty()) {
        idx.push_back(0);
    }
    long long n = 0;
    const int channels = (V > 0 && views[0].image.type == MAT_32FC3) ? 3 : 1;
    const int st = apd_fuse_views(g_fusion_device, V, cams.data(), imgs.data(), channels, deps.data(), nors.data(), weaks.data(),
                                  any_block ? blocks.data() : nullptr, rows.data(), cols.data(), offs.data(), idx.data(), 0, ply_path.string().c_str(), &n);
    if (st != APD_OK) {
        std::cerr << apd_fusion_last_error() << std::endl;
        return -1;
    }
    return n;
}

}  // namespace

void SetFusionDevice(int device) { g_fusion_device = device; }

// Reads every view's final maps from <dense>/APD/<id>/ and fuses them into APD/APD.ply (APD.cpp:826-977).
void RunFusion(const path &dense_folder, const std::vector<Problem> &problems) { RunFusionWithMaps(dense_folder, problems, nullptr); }

// Inputs of the fusion besides the maps: colour image, camera, optional block mask of every view, resampled to the size of
// the maps when that differs from the image's (RescaleImageAndCamera, APD.cpp:729-750), and the sources of every view as view
// indices.  maps: host maps (RunFusionWithMaps) or nullptr with map_cols x map_rows > 0 (the maps are on the device) or
// nullptr with 0 x 0 (read the files).
namespace {

// log: where the reference's progress lines go ("Reading image ...", "Fusing image ...", APD.cpp:855, :899); the prefetch worker
// collects them and RunFusionOnDevice prints them where the reference would, after the passes' own lines.
bool prepare_fusion_inputs(const path &dense_folder, const std::vector<Problem> &problems, const std::vector<FinalMaps> *maps, int map_cols,
                           int map_rows, std::vector<FusionView> &views, std::vector<std::vector<int>> &sources, unsigned threads = 0,
                           std::ostream *log = nullptr)
{
    std::ostream &out = log ? *log : std::cout;
    const bool on_device = !maps && map_cols > 0 && map_rows > 0;
    views.assign(problems.size(), FusionView());
    std::unordered_map<int, int> index_of_id;
    const path block_folder = dense_folder / path("blocks");
    const bool use_block = std::filesystem::exists(block_folder);  // APD.cpp:849-853
    for (size_t i = 0; i < problems.size(); ++i) {
        out << "Reading image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
        index_of_id.emplace(problems[i].ref_image_id, (int)i);
    }
    // the views are independent here: decode, read and rescale them on several host threads
    std::vector<int> failed(problems.size(), 0);
    ParallelFor(problems.size(), [&](size_t i) {
        const Problem &problem = problems[i];
        FusionView &v = views[i];
        if (!ReadColorImage(dense_folder / path("images") / path(ToFormatIndex(problem.ref_image_id)), v.image)) {
            failed[i] = 1;
            return;
        }
        memset(&v.cam, 0, sizeof(v.cam));
        ReadCamera(dense_folder / path("cams") / path(ToFormatIndex(problem.ref_image_id) + "_cam.txt"), v.cam);
        int cols = map_cols, rows = map_rows;
        if (!on_device) {
            if (maps) {
                v.depth = (*maps)[i].depth;
                v.normal = (*maps)[i].normal;
                v.weak = (*maps)[i].weak.clone();  // resampled in place below
            } else {
                ReadBinMat(problem.result_folder / path("depths.dmb"), v.depth);
                ReadBinMat(problem.result_folder / path("normals.dmb"), v.normal);
                ReadBinMat(problem.result_folder / path("weak.bin"), v.weak);
            }
            if (v.depth.empty() || v.normal.empty() || v.weak.empty()) {
                std::cerr << "Missing maps of view " << problem.ref_image_id << " in " << problem.result_folder << std::endl;
                failed[i] = 1;
                return;
            }
            cols = v.depth.cols;
            rows = v.depth.rows;
        }
        if (cols != v.image.cols || rows != v.image.rows) {  // RescaleImageAndCamera, APD.cpp:729-750
            const float scale_x = cols / static_cast<float>(v.image.cols);
            const float scale_y = rows / static_cast<float>(v.image.rows);
            // the reference resizes the 8-bit colour image: each channel resampled, then rounded back to 8 bit
            const size_t n_in = (size_t)v.image.rows * v.image.cols, n_out = (size_t)rows * cols;
            Mat out(rows, cols, MAT_32FC3), plane(v.image.rows, v.image.cols, MAT_32FC1), scaled;
            for (int ch = 0; ch < 3; ++ch) {
                for (size_t k = 0; k < n_in; ++k) {
                    plane.ptr<float>()[k] = v.image.ptr<float>()[3 * k + ch];
                }
                ResizeLinear(plane, scaled, cols, rows);
                for (size_t k = 0; k < n_out; ++k) {
                    out.ptr<float>()[3 * k + ch] = std::nearbyint(scaled.ptr<float>()[k]);
                }
            }
            v.image = out;
            v.cam.K[0] *= scale_x;
            v.cam.K[2] *= scale_x;
            v.cam.K[4] *= scale_y;
            v.cam.K[5] *= scale_y;
        }
        v.cam.width = cols;
        v.cam.height = rows;
        if (!on_device) {  // device maps: the weak map already has the size of the depth map
            RescaleMatToTargetSize<uint8_t>(v.weak, v.weak, cols, rows);
        }
        if (use_block) {  // blocks/mask_<id>.jpg, read as grey (APD.cpp:871-875); must have the size of the depth map
            Mat grey;
            if (ReadGrayImage(block_folder / path("mask_" + std::to_string(problem.ref_image_id)), grey) && grey.rows == rows && grey.cols == cols) {
                v.block.create(grey.rows, grey.cols, MAT_8UC1);
                for (size_t k = 0; k < (size_t)grey.rows * grey.cols; ++k) {
                    v.block.ptr<uint8_t>()[k] = (uint8_t)grey.ptr<float>()[k];
                }
            }
        }
    }, threads);
    for (int f : failed) {
        if (f) {
            return false;
        }
    }
    sources.assign(problems.size(), std::vector<int>());
    for (size_t i = 0; i < problems.size(); ++i) {
        out << "Fusing image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
        for (int id : problems[i].src_image_ids) {
            // A source without a problem of its own has no maps to check against and is skipped.  (The reference's
            // imageIdToindexMap[id] default-inserts 0 for it, APD.cpp:919, i.e. silently checks against view 0 instead.)
            const auto it = index_of_id.find(id);
            if (it != index_of_id.end()) {
                sources[i].push_back(it->second);
            }
        }
    }
    return true;
}

}  // namespace

// The same with the final maps already in memory (maps[i] belongs to problems[i]; nullptr reads the files).
void RunFusionWithMaps(const path &dense_folder, const std::vector<Problem> &problems, const std::vector<FinalMaps> *maps)
{
    const auto t_inputs = std::chrono::steady_clock::now();
    std::vector<FusionView> views;
    std::vector<std::vector<int>> sources;
    if (!prepare_fusion_inputs(dense_folder, problems, maps, 0, 0, views, sources)) {
        exit(EXIT_FAILURE);
    }
    const path ply_path = dense_folder / path("APD") / path("APD.ply");
    const auto t_fuse = std::chrono::steady_clock::now();
    std::cout << "Fusion inputs ready: " << std::chrono::duration_cast<std::chrono::milliseconds>(t_fuse - t_inputs).count() << " ms" << std::endl;
    const long long n = fuse_dispatch(views, sources, ply_path);
    std::cout << "Fusion + PLY: " << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_fuse).count() << " ms" << std::endl;
    if (n < 0) {
        exit(EXIT_FAILURE);  // like every other device error of the reference (CudaSafeCall, APD.cpp:315-323)
    }
    std::cout << "Fused " << n << " points into " << ply_path << std::endl;
}

// The final maps are on `device` already (host/multi_device.cpp keeps every view's state there and gathers the other devices'
// into it): only the colour images (and block masks) go up; nothing comes down but the points.  The inputs that do not depend
// on the passes -- colour decode (cv::imread(IMREAD_COLOR), APD.cpp:859), cameras, masks, their upload -- are prepared on a
// background thread WHILE the passes run (StartFusionInputs right after the grey images are loaded): at 152 views of 1920 x 1080
// they take 4 s, as long as half of an 8-device run's passes.
struct FusionPrefetch {
    std::thread worker;
    path dense_folder;
    std::vector<Problem> problems;
    int device = 0, cols = 0, rows = 0;
    std::vector<FusionView> views;
    std::vector<std::vector<int>> sources;
    std::vector<void *> owned;
    std::vector<const float *> imgs;
    std::vector<const uint8_t *> blocks;
    bool any_block = false, ok = false;
    std::atomic<bool> failed{false};  // set by the worker as soon as it gives up: RunMultiDevice polls it between levels (FusionInputsFailed)
    int channels = 3;
    long long prepare_ms = 0;
    std::string error;
    std::ostringstream log;           // the reference's progress lines, printed by RunFusionOnDevice
};

FusionPrefetch *StartFusionInputs(const path &dense_folder, const std::vector<Problem> &problems, int device, int cols, int rows, unsigned threads)
{
    FusionPrefetch *f = new FusionPrefetch();
    f->dense_folder = dense_folder;
    f->problems = problems;
    f->device = device;
    f->cols = cols;
    f->rows = rows;
    f->worker = std::thread([f, threads]() {
        const auto t0 = std::chrono::steady_clock::now();
        // every way out of this worker releases the upload stream and the staging buffer, and a failure is published at once
        void *stream = nullptr, *staging = nullptr;
        struct Guard {
            FusionPrefetch *f;
            void *&stream, *&staging;
            ~Guard()
            {
                if (stream) {
                    apd_stream_destroy(f->device, stream);
                }
                if (staging) {
                    apd_host_free(staging);
                }
                if (!f->ok) {
                    f->failed.store(true);
                }
            }
        } guard{f, stream, staging};
        if (!prepare_fusion_inputs(f->dense_folder, f->problems, nullptr, f->cols, f->rows, f->views, f->sources, threads, &f->log)) {
            f->error = "fusion inputs could not be read";
            return;
        }
        const int V = (int)f->views.size();
        const size_t n = (size_t)f->rows * f->cols;
        f->channels = (V > 0 && f->views[0].image.type == MAT_32FC3) ? 3 : 1;
        f->imgs.assign(V, nullptr);
        f->blocks.assign(V, nullptr);
        auto upload = [&](const void *host, size_t bytes) -> void * {
            void *p = nullptr;
            if (apd_device_malloc(f->device, bytes, &p) != APD_OK) {
                f->error = std::string("fusion: device allocation failed: ") + apd_exchange_last_error();
                return nullptr;
            }
            f->owned.push_back(p);
            if (apd_device_memcpy(f->device, p, host, bytes) != APD_OK) {
                f->error = std::string("fusion: device upload failed: ") + apd_exchange_last_error();
                return nullptr;
            }
            return p;
        };
        // one allocation for every view's colour image: device allocations are not free while other threads launch kernels
        const size_t image_bytes = n * 4 * (size_t)f->channels;
        void *all_images = nullptr;
        if (apd_device_malloc(f->device, image_bytes * (size_t)std::max(V, 1), &all_images) != APD_OK) {
            f->error = std::string("fusion: device allocation failed: ") + apd_exchange_last_error();
            return;
        }
        f->owned.push_back(all_images);
        // Uploads on a stream of their own through ONE page-locked staging buffer.  Measured on 24 views of 1920 x 1080 with the
        // passes running beside it: a plain hipMemcpy of the pageable image, and just as much hipHostRegister + asynchronous copy +
        // unregister per image, cost the passes what the early start saved (8.2 -> 8.6 s): every map / unmap of host pages holds
        // the device's queues up; one mapping for the whole run does not.
        if (apd_stream_create(f->device, &stream) != APD_OK) {
            stream = nullptr;
        } else if (apd_host_alloc(image_bytes, &staging) != APD_OK) {
            staging = nullptr;
        }
        for (int i = 0; i < V; ++i) {
            void *dst = (char *)all_images + (size_t)i * image_bytes;
            const void *src = f->views[i].image.data();
            int rc;
            if (stream && staging) {
                memcpy(staging, src, image_bytes);
                rc = apd_device_memcpy_async(f->device, stream, dst, staging, image_bytes);
                rc = rc != APD_OK ? rc : apd_stream_synchronize(f->device, stream);
            } else {
                rc = apd_device_memcpy(f->device, dst, src, image_bytes);
            }
            if (rc != APD_OK) {
                f->error = std::string("fusion: device upload failed: ") + apd_exchange_last_error();
                return;
            }
            f->imgs[i] = (const float *)dst;
            f->views[i].image = Mat();  // the host copy is done with
            if (!f->views[i].block.empty()) {
                f->blocks[i] = (const uint8_t *)upload(f->views[i].block.data(), n);
                if (!f->blocks[i]) {
                    return;
                }
                f->any_block = true;
            }
        }
        f->prepare_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        f->ok = true;
    });
    return f;
}

// true as soon as the prefetch worker has given up (unreadable colour image, device out of memory): the caller stops before the
// next level instead of learning it after every pass has run
bool FusionInputsFailed(const FusionPrefetch *f, std::string *why)
{
    if (!f || !f->failed.load()) {
        return false;
    }
    if (why) {
        *why = f->error;   // written before the flag was set
    }
    return true;
}

void CancelFusionInputs(FusionPrefetch *f)
{
    if (!f) {
        return;
    }
    if (f->worker.joinable()) {
        f->worker.join();
    }
    for (void *p : f->owned) {
        apd_device_free(f->device, p);
    }
    delete f;
}

void RunFusionOnDevice(FusionPrefetch *f, const std::vector<const float *> &depths, const std::vector<const float *> &normals,
                       const std::vector<const uint8_t *> &weaks)
{
    const auto t_wait = std::chrono::steady_clock::now();
    f->worker.join();
    if (!f->ok) {
        std::cerr << f->error << std::endl;
        exit(EXIT_FAILURE);
    }
    const int V = (int)f->views.size();
    std::vector<apd_camera> cams(V);
    std::vector<int> rws(V, f->rows), cls(V, f->cols), offs(V + 1, 0), idx;
    for (int i = 0; i < V; ++i) {
        cams[i] = f->views[i].cam;
        idx.insert(idx.end(), f->sources[i].begin(), f->sources[i].end());
        offs[i + 1] = (int)idx.size();
    }
    if (idx.empty()) {
        idx.push_back(0);
    }
    const path ply_path = f->dense_folder / path("APD") / path("APD.ply");
    const auto t_fuse = std::chrono::steady_clock::now();
    std::cout << f->log.str();   // "Reading image ..." / "Fusing image ...": here, where the reference prints them (after the passes)
    std::cout << "Fusion inputs ready: prepared in " << f->prepare_ms << " ms behind the passes, waited "
              << std::chrono::duration_cast<std::chrono::milliseconds>(t_fuse - t_wait).count() << " ms" << std::endl;
    long long count = 0;
    const int st = apd_fuse_views(f->device, V, cams.data(), f->imgs.data(), f->channels, depths.data(), normals.data(), weaks.data(),
                                  f->any_block ? f->blocks.data() : nullptr, rws.data(), cls.data(), offs.data(), idx.data(), 1, ply_path.string().c_str(), &count);
    double ms_setup = 0, ms_views = 0, ms_file = 0;
    apd_fusion_last_timing(&ms_setup, &ms_views, &ms_file);
    std::cout << "Fusion + PLY: " << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_fuse).count() << " ms (set-up "
              << (long long)ms_setup << ", views " << (long long)ms_views << ", file " << (long long)ms_file << ")" << std::endl;
    const std::string err = st != APD_OK ? apd_fusion_last_error() : "";
    CancelFusionInputs(f);
    if (st != APD_OK) {
        std::cerr << err << std::endl;
        exit(EXIT_FAILURE);
    }
    std::cout << "Fused " << count << " points into " << ply_path << std::endl;
}

extern "C" {

// Flat entry for the Python pipeline (maps already in memory after the all-gather): per-view pointers, all maps of view
// i are rows[i] x cols[i]; sources of view i are pair_indices[pair_offsets[i] .. pair_offsets[i+1]).  Returns the
// number of points written to `ply_path`.
long long apdhost_fuse(int num_views, const apd_camera *cameras, const float *const *images, int image_channels,
                       const float *const *depths, const float *const *normals, const uint8_t *const *weaks,
                       const uint8_t *const *blocks, const int *rows, const int *cols, const int *pair_offsets,
                       const int *pair_indices, const char *ply_path)
{
    std::vector<FusionView> views(num_views);
    std::vector<std::vector<int>> sources(num_views);
    for (int i = 0; i < num_views; ++i) {
        const size_t n = (size_t)rows[i] * cols[i];
        FusionView &v = views[i];
        v.cam = cameras[i];
        v.cam.width = cols[i];
        v.cam.height = rows[i];
        v.image.create(rows[i], cols[i], image_channels == 3 ? MAT_32FC3 : MAT_32FC1);
        v.depth.create(rows[i], cols[i], MAT_32FC1);
        v.normal.create(rows[i], cols[i], MAT_32FC3);
        v.weak.create(rows[i], cols[i], MAT_8UC1);
        memcpy(v.image.data(), images[i], n * 4 * (image_channels == 3 ? 3 : 1));
        memcpy(v.depth.data(), depths[i], n * 4);
        memcpy(v.normal.data(), normals[i], n * 12);
        memcpy(v.weak.data(), weaks[i], n);
        if (blocks && blocks[i]) {
            v.block.create(rows[i], cols[i], MAT_8UC1);
            memcpy(v.block.data(), blocks[i], n);
        }
        sources[i].assign(pair_indices + pair_offsets[i], pair_indices + pair_offsets[i + 1]);
    }
    return fuse_dispatch(views, sources, path(ply_path));  // -1: the device fusion failed (message on stderr)
}

}  // extern "C"
