// main.cpp -- drop-in driver of the PatchMatch path.
//
//   APD dense_folder [gpu_index] [--seed S] [--iters K] [--single-level] [--max-src N] [--keep-maps] [--no-fusion]
//
// Same command line, files and results as the reference driver (main.cpp:140-233), organised differently:
//   * pair.txt is read as one token stream with diagnostics (the reference never notices a missing or short file,
//     main.cpp:10);
//   * the passes are rows of one table (BuildSchedule) -- the same table apd-mvs_amd/pipeline.py::pass_schedule builds,
//     so the file-based and the in-memory scheduler cannot drift apart;
//   * every image is decoded once per process instead of once per (view, pass).
// State moves between passes through depths.dmb / normals.dmb / weak.bin / selected_views.bin in <dense>/APD/<%08d>/,
// as in the reference; RunFusion then writes APD/APD.ply and the four state files are removed (--keep-maps keeps them).
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

namespace {

struct Options {
    path dense_folder;
    int gpu_index = 0;
    uint64_t seed = 12345;
    int iters = 3;          // PatchMatchParams::max_iterations of every pass (reference: 3)
    int max_src = 0;        // > 0: keep only the first N sources of each pair.txt entry (they are sorted by score)
    bool single_level = false, keep_maps = false, no_fusion = false;
};

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
        } else if (i == 2 && a.size() && a[0] != '-') {
            o.gpu_index = atoi(a.c_str());  // positional, as in the reference (main.cpp:149-153)
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
        if (!ReadGrayImage(p.dense_folder / "images" / ToFormatIndex(p.ref_image_id), image)) {
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

// One pass over all views.  round_num pyramid levels, coarse to fine; per level one photometric pass and three
// geometric ones (main.cpp:168-215).
struct Pass {
    int level = 0;             // i of main.cpp:168
    int iteration = 0;         // Problem::iteration, counts passes
    int scale_size = 1;        // 2^(round_num - 1 - level)
    RunState state = FIRST_INIT;
    bool geom_consistency = false, use_APD = false;
    int weak_peak_radius = 6;
    float ransac_threshold = 0.005f;  // only read when use_APD (the struct default otherwise, main.h:92)
    int rotate_time = 4;
};

int RoundNum(int width, int height)  // main.cpp:72-88: halve until the longer side is <= 1000
{
    int rounds = 1;
    for (int longest = std::max(width, height); longest > 1000; longest /= 2) {
        ++rounds;
    }
    return rounds;
}

std::vector<Pass> BuildSchedule(int round_num, bool single_level)
{
    std::vector<Pass> plan;
    for (int level = 0; level < round_num; ++level) {
        for (int k = 0; k < 4; ++k) {  // k = 0: photometric, k = 1..3: geometric with j = k - 1
            Pass p;
            p.level = level;
            p.iteration = (int)plan.size();
            p.scale_size = single_level ? 1 : 1 << (round_num - 1 - level);
            p.state = k > 0 ? REFINE_ITER : (level == 0 ? FIRST_INIT : REFINE_INIT);
            p.geom_consistency = k > 0;
            p.weak_peak_radius = k == 0 ? 6 : std::max(4 - 2 * (k - 1), 2);
            p.use_APD = level > 0;
            if (p.use_APD) {
                p.ransac_threshold = (float)(0.01 - level * 0.00125);  // double arithmetic, then float, as main.cpp:180
                p.rotate_time = std::min(1 << level, 4);
            }
            plan.push_back(p);
        }
    }
    return plan;
}

// The reference keeps one PatchMatchParams per problem alive across passes and only overwrites some fields, so
// ransac_threshold / rotate_time of level 0 are the struct defaults: same here.
void Configure(Problem &problem, const Pass &pass, const Options &o)
{
    PatchMatchParams &q = problem.params;
    q.state = pass.state;
    q.use_APD = pass.use_APD;
    if (pass.use_APD) {
        q.ransac_threshold = pass.ransac_threshold;
        q.rotate_time = pass.rotate_time;
    }
    q.geom_consistency = pass.geom_consistency;
    q.max_iterations = o.iters;
    q.weak_peak_radius = pass.weak_peak_radius;
    q.seed = o.seed + (uint64_t)pass.iteration * 7919u + (uint64_t)problem.index;  // the reference seeds with clock64()
    problem.iteration = pass.iteration;
    problem.show_medium_result = true;
    problem.scale_size = pass.scale_size;
}

const char *const kStateFiles[4] = {"depths.dmb", "normals.dmb", "weak.bin", "selected_views.bin"};

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
    Options opt;
    if (!ParseOptions(argc, argv, opt)) {
        fprintf(stderr, "USAGE: APD dense_folder [gpu_index] [--seed S] [--iters K] [--single-level] [--max-src N] [--keep-maps] [--no-fusion]\n");
        return EXIT_FAILURE;
    }
    if (opt.gpu_index < 0 || opt.gpu_index >= apd_device_count()) {
        fprintf(stderr, "Requested GPU %d, found %d device(s)\n", opt.gpu_index, apd_device_count());
        return EXIT_FAILURE;
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
    PrefetchGrayImages(opt.dense_folder / "images", ids);  // decoded once, on several host threads
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
