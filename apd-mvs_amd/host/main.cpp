// main.cpp -- drop-in driver of the PatchMatch path.
//
//   APD dense_folder [gpu_index | gpu,gpu,...] [--seed S] [--iters K] [--single-level] [--max-src N] [--keep-maps] [--no-fusion]
//       [--files | --in-memory] [--jacobi] [--ranks N] [--no-rccl] [--rccl]
//
// One device index: the reference's driver.  A device LIST (or --jacobi): host/multi_device.cpp -- views sharded over the
// devices, state resident on them, depth maps all-gathered after every pass (RCCL when there is more than one rank -- its set-up takes seconds
// -- direct copies for a single rank; --rccl: RCCL even then; --no-rccl: direct copies only).  With --jacobi a single device runs
// one to three scheduler ranks, by frame size (--ranks N: exactly N).  --in-memory: the same scheduler with one rank in the
// reference's own order (a view of a geometric pass sees the depth maps its sources have at that moment): the bytes of the
// file-based driver without the files.
//
// Same command line, files and results as the reference driver (main.cpp:140-233), organised differently:
//   * pair.txt is read as one token stream with diagnostics (the reference never notices a missing or short file,
//     main.cpp:10);
//   * the passes are rows of one table (BuildSchedule) -- the same table apd-mvs_amd/pipeline.py::pass_schedule builds,
//     so the file-based and the in-memory scheduler cannot drift apart;
//   * every image is decoded once per process instead of once per (view, pass).
// --files: state moves between passes through depths.dmb / normals.dmb / weak.bin / selected_views.bin in <dense>/APD/<%08d>/,
// as in the reference; RunFusion then writes APD/APD.ply and the four state files are removed (--keep-maps keeps them).
// Default for one device: the same order of views and the same bytes with the state resident on the device (host/multi_device.cpp,
// --in-memory) when the folder fits it -- the files of the last pass are written with --keep-maps.
// Not built: the debug JPEGs of show_medium_result.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>

#include "APD.h"
#include "schedule.h"

namespace {

bool ParseOptions(int argc, char **argv, Options &o)
{
    if (argc < 2) {
        return false;
    }
    o.dense_folder = path(argv[1]);
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](uint64_t &dst) {
            if (i + 1 >= argc) {
                return false;
            }
            dst = strtoull(argv[++i], nullptr, 10);
            return true;
        };
        uint64_t v = 0;
        if (a == "--seed") {
            if (!value(o.seed)) return false;
        } else if (a == "--iters") {
            if (!value(v)) return false;
            o.iters = (int)v;
        } else if (a == "--max-src") {
            if (!value(v)) return false;
            o.max_src = (int)v;
        } else if (a == "--single-level") {
            o.single_level = true;
        } else if (a == "--keep-maps") {
            o.keep_maps = true;
        } else if (a == "--no-fusion") {
            o.no_fusion = true;
        } else if (a == "--late-fusion-inputs") {
            o.late_fusion_inputs = true;
        } else if (a == "--scheduler-free-gb" && i + 1 < argc) {
            o.scheduler_free_gb = atof(argv[++i]);
        } else if (a == "--clean-exit") {
            o.clean_exit = true;
        } else if (a == "--copy-images") {
            o.copy_images = true;
        } else if (a == "--jacobi") {
            o.jacobi = true;
        } else if (a == "--in-memory") {
            o.in_memory = true;
        } else if (a == "--files") {
            o.files = true;
        } else if (a == "--ranks") {
            if (!value(v)) return false;
            o.ranks_per_device = (int)v;
        } else if (a == "--no-rccl") {
            o.use_rccl = false;
        } else if (a == "--exchange-device-sync") {
            o.exchange_device_sync = true;
        } else if (a == "--rccl") {
            o.force_rccl = true;
        } else if (i == 2 && a.size() && a[0] != '-') {
            // positional, as in the reference (main.cpp:149-153); additive: a comma-separated list = one scheduler rank per entry
            o.devices.clear();
            size_t pos = 0;
            while (pos <= a.size()) {
                const size_t comma = a.find(',', pos);
                const std::string item = a.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                char *end = nullptr;
                const long d = strtol(item.c_str(), &end, 10);
                if (item.empty() || *end != '\0') {
                    fprintf(stderr, "bad device list '%s'\n", a.c_str());
                    return false;
                }
                o.devices.push_back((int)d);
                if (comma == std::string::npos) {
                    break;
                }
                pos = comma + 1;
            }
            o.gpu_index = o.devices[0];
        } else {
            fprintf(stderr, "unknown argument '%s'\n", a.c_str());
            return false;
        }
    }
    return true;
}

// pair.txt (written by colmap2mvsnet.py:449-456): <num views>, then per view <ref id> and <n> followed by n
// (<source id> <score>) pairs.  Sources with score <= 0 are not used (main.cpp:43).  Returns an empty string or what is
// wrong with the file.
std::string ReadPairFile(const path &file, const path &dense_folder, std::vector<Problem> &problems)
{
    problems.clear();
    std::ifstream in(file);
    if (!in) {
        return "cannot open " + file.string();
    }
    const std::vector<std::string> tok{std::istream_iterator<std::string>(in), std::istream_iterator<std::string>()};
    size_t at = 0;
    auto number = [&](double &out) {
        if (at >= tok.size()) {
            return false;
        }
        char *end = nullptr;
        out = strtod(tok[at].c_str(), &end);
        const bool ok = end != tok[at].c_str() && *end == '\0';
        ++at;
        return ok;
    };
    double count = 0;
    if (!number(count) || count < 0) {
        return file.string() + ": no view count";
    }
    for (int i = 0; i < (int)count; ++i) {
        double id = 0, n = 0;
        if (!number(id) || !number(n) || n < 0) {
            return file.string() + ": entry " + std::to_string(i) + " is incomplete (token " + std::to_string(at) + ")";
        }
        Problem p;
        p.index = i;
        p.ref_image_id = (int)id;
        p.dense_folder = dense_folder;
        p.result_folder = dense_folder / "APD" / ToFormatIndex(p.ref_image_id);
        for (int j = 0; j < (int)n; ++j) {
            double src = 0, score = 0;
            if (!number(src) || !number(score)) {
                return file.string() + ": view " + std::to_string(p.ref_image_id) + " lists " + std::to_string((int)n) +
                       " sources but holds fewer";
            }
            if ((float)score > 0.0f) {
                p.src_image_ids.push_back((int)src);
            }
        }
        problems.push_back(std::move(p));
    }
    return std::string();
}

// main.cpp:51-70: every reference image must exist and have the size of the first one.  The sizes of the source images
// and of the prior maps are checked where they are loaded (APD::InuputInitialization).
bool CheckImages(const std::vector<Problem> &problems, int &width, int &height)
{
    width = height = 0;
    for (const Problem &p : problems) {
        Mat image;
        if (!ReadGrayImageShared(p.dense_folder / "images" / ToFormatIndex(p.ref_image_id), image)) {
            return false;
        }
        if (width == 0) {
            width = image.cols;
            height = image.rows;
        } else if (image.cols != width || image.rows != height) {
            return false;
        }
    }
    return !problems.empty();
}

}  // namespace

// One (view, pass): the reference's ProcessProblem (main.cpp:91-138) -- run the path, post-process (depth outside the
// search range -> 0 and UNKNOWN, main.cpp:109-112), write the four state files.
void ProcessProblem(const Problem &problem)
{
    printf("Processing image: %08d...\n", problem.ref_image_id);
    const auto t0 = std::chrono::steady_clock::now();
    APD apd(problem);
    apd.InuputInitialization();
    apd.CudaSpaceInitialization();
    apd.SetDataPassHelperInCuda();
    apd.RunPatchMatch();
    const int W = apd.GetWidth(), H = apd.GetHeight();
    const float lo = apd.GetDepthMin(), hi = apd.GetDepthMax();
    Mat depth(H, W, MAT_32FC1), normal(H, W, MAT_32FC3), states = apd.GetPixelStates();
    ParallelFor((size_t)H, [&](size_t row) {
        float *d = depth.ptr<float>((int)row);
        Vec3f *n = normal.ptr<Vec3f>((int)row);
        uint8_t *s = states.ptr<uint8_t>((int)row);
        for (int c = 0; c < W; ++c) {
            const float4 h = apd.GetPlaneHypothesis((int)row, c);
            const bool out_of_range = h.w < lo || h.w > hi;  // false for NaN, as in the reference
            d[c] = out_of_range ? 0.0f : h.w;
            if (out_of_range) {
                s[c] = UNKNOWN;
            }
            n[c] = Vec3f{{h.x, h.y, h.z}};
        }
    });
    const Mat *maps[4] = {&depth, &normal, &states, nullptr};
    const Mat views = apd.GetSelectedViews();
    maps[3] = &views;
    for (int k = 0; k < 4; ++k) {
        if (!WriteBinMat(problem.result_folder / kStateFiles[k], *maps[k])) {
            fprintf(stderr, "cannot write %s\n", (problem.result_folder / kStateFiles[k]).string().c_str());
            exit(EXIT_FAILURE);
        }
    }
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    printf("Processing image: %08d done!\nCost time: %lld ms\n", problem.ref_image_id, (long long)ms);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const auto t_start = std::chrono::steady_clock::now();
    // Views in flight run on their own HIP streams, and the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
    // two streams that land on one queue take turns.  Eight queues: 24 x 1080p passes 7.70 -> 7.58 s with six views in flight, and the
    // 1.3 s round 5 charged to the peer-copy exchange of `0,0 --no-rccl` (8.82 -> 7.54 s: its two lanes shared a queue; loading librccl
    // happened to shift the mapping) -- profiles/r06/ab_hw_queues.txt.  Read by the HIP runtime when it starts, i.e. at the first HIP
    // call below; a value the caller has set is kept.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    Options opt;
    if (!ParseOptions(argc, argv, opt)) {
        fprintf(stderr, "USAGE: APD dense_folder [gpu_index | gpu,gpu,...] [--seed S] [--iters K] [--single-level] [--max-src N] [--keep-maps] [--no-fusion] [--files | --in-memory] [--jacobi] [--ranks N] [--no-rccl] [--rccl] [--exchange-device-sync] [--late-fusion-inputs] [--copy-images] [--clean-exit]\n");
        return EXIT_FAILURE;
    }
    if (opt.devices.empty()) {
        opt.devices.push_back(opt.gpu_index);
    }
    for (int d : opt.devices) {
        if (d < 0 || d >= apd_device_count()) {
            fprintf(stderr, "Requested GPU %d, found %d device(s)\n", d, apd_device_count());
            return EXIT_FAILURE;
        }
    }
    APD::SetDevice(opt.gpu_index);
    SetFusionDevice(opt.gpu_index);

    std::vector<Problem> problems;
    const std::string why = ReadPairFile(opt.dense_folder / "pair.txt", opt.dense_folder, problems);
    if (!why.empty()) {
        fprintf(stderr, "%s\n", why.c_str());
        return EXIT_FAILURE;
    }
    std::vector<int> ids;  // every image any pass will read
    for (Problem &p : problems) {
        if (opt.max_src > 0 && (int)p.src_image_ids.size() > opt.max_src) {
            p.src_image_ids.resize((size_t)opt.max_src);
        }
        std::filesystem::create_directories(p.result_folder);
        ids.push_back(p.ref_image_id);
        ids.insert(ids.end(), p.src_image_ids.begin(), p.src_image_ids.end());
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    // Sources without an entry of their own (a subset of the views is reconstructed) are loaded like any other image
    // (APD.cpp:419-452): they contribute no depth map to the geometric term and take no part in the fusion (host/APD.cpp,
    // host/fusion.cpp; apd-mvs_amd/pipeline.py does the same).  A view that lists itself is refused: the fusion would count it
    // as its own witness.
    for (const Problem &p : problems) {
        for (int s : p.src_image_ids) {
            if (s == p.ref_image_id) {
                fprintf(stderr, "pair.txt: view %d lists itself as a source\n", s);
                return EXIT_FAILURE;
            }
        }
    }
    {
        std::vector<int> refs;
        for (const Problem &p : problems) {
            refs.push_back(p.ref_image_id);
        }
        APD::SetReconstructedViews(refs);
    }
    PrefetchGrayImages(opt.dense_folder / "images", ids);  // decoded once, on several host threads
    if (opt.files && (opt.in_memory || opt.jacobi || opt.devices.size() > 1)) {
        fprintf(stderr, "--files is the single-device driver (no device list, --jacobi or --in-memory)\n");
        return EXIT_FAILURE;
    }
    // `APD dense_folder [gpu]`, the reference's command line: its order of views and its bytes, in memory when the folder fits the
    // device (InMemoryBytesPerPixel: level images, two sets of depth maps, every view's state, the handles of the views in flight, the
    // final maps and the fusion's buffers), through the files otherwise or with --files.
    bool auto_in_memory = false;   // the in-memory scheduler was this function's choice, not the command line's
    if (!opt.files && !opt.in_memory && !opt.jacobi && opt.devices.size() == 1) {
        int w = 0, h = 0;
        if (CheckImages(problems, w, h)) {
            size_t max_src = 1;
            for (const Problem &p : problems) {
                max_src = std::max(max_src, p.src_image_ids.size());
            }
            // the scheduler's own test (same function, same free-memory figure and --scheduler-free-gb cap), BEFORE it loads a single
            // full-size image or prints its header: a folder that does not fit goes to the file loop at once
            const int lanes = InMemoryLanes(opt, w, h, (int)problems.size(), 1, true);   // the count RunMultiDevice will use
            const InMemoryFit fit = TestInMemoryFit(opt, opt.gpu_index, w, h, (int)ids.size(), (int)problems.size(), 1, lanes, (int)max_src);
            opt.in_memory = fit.have_memory && fit.fits;
            auto_in_memory = opt.in_memory;
            if (fit.have_memory && !fit.fits) {
                printf("%.1f GB of resident state against %.1f GB free on device %d: this folder does not fit the in-memory scheduler, passing state through files\n",
                       fit.need_bytes / 1e9, fit.free_bytes / 1e9, opt.gpu_index);
            }
        }
    }
    if (opt.in_memory && (opt.devices.size() > 1 || opt.jacobi)) {
        fprintf(stderr, "--in-memory keeps the reference's order of views: one device, one rank (no device list, no --jacobi)\n");
        return EXIT_FAILURE;
    }
    if (opt.devices.size() > 1 || opt.jacobi || opt.in_memory) {
        printf("Start-up (pair.txt, image decode on several threads, device query): %lld ms\n",
               (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count());
        const int rc = RunMultiDevice(opt, problems);  // host/multi_device.cpp: in memory, views sharded over the device list
        if (rc == kExitDoesNotFit && auto_in_memory) {
            // the plain `APD folder gpu` command line must process every folder the file driver can (free memory may have shrunk
            // between the estimate above and the scheduler's own): nothing has been run or written yet, take the files
            printf("in-memory scheduler refused the folder: passing state through files\n");
            opt.in_memory = false;
        } else {
            // Every file is written and closed.  Leaving through _Exit skips the one-by-one release of tens of gigabytes of device memory
            // and the runtime's own teardown (half a second at 152 views): the driver reclaims a process's memory in one step.
            // (--clean-exit returns normally: a profiler that writes its output from an exit handler -- rocprofv3 -- needs that.)
            const int code = rc == kExitDoesNotFit ? EXIT_FAILURE : rc;
            fflush(stdout);
            fflush(stderr);
            if (!opt.clean_exit) {
                std::_Exit(code);
            }
            return code;
        }
    }
    int width = 0, height = 0;
    if (!CheckImages(problems, width, height)) {
        fprintf(stderr, "Images may error, check it!\n");  // the reference's message (main.cpp:158)
        return EXIT_FAILURE;
    }
    printf("There are %zu problems needed to be processed!\n", problems.size());
    const int round_num = opt.single_level ? 1 : RoundNum(width, height);
    printf("Round nums: %d\n", round_num);

    for (const Pass &pass : BuildSchedule(round_num, opt.single_level)) {
        for (Problem &p : problems) {
            Configure(p, pass, opt);
            ProcessProblem(p);
        }
        if (pass.iteration % 4 == 3) {
            printf("Round: %d done\n", pass.level);
        }
    }
    if (!opt.no_fusion) {
        RunFusion(opt.dense_folder, problems);  // main.cpp:219
        if (!opt.keep_maps) {                   // main.cpp:220-230
            for (const Problem &p : problems) {
                for (const char *name : kStateFiles) {
                    std::filesystem::remove(p.result_folder / name);
                }
            }
        }
    }
    printf("All done\n");
    return EXIT_SUCCESS;
}
