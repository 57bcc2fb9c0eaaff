// main.cpp -- drop-in driver for the PatchMatch path: `APD dense_folder [gpu_index] [--seed S]
// [--iters K] [--single-level] [--max-src N] [--keep-maps] [--no-fusion]`.
//
// Mirrors the reference's CLI and per-pass parameter schedule (main.cpp:140-233): pair.txt ->
// Problems -> round_num pyramid levels x (1 photometric + 3 geometric) passes, state exchanged through
// depths.dmb / normals.dmb / weak.bin / selected_views.bin in <dense>/APD/<%08d>/, then RunFusion ->
// APD/APD.ply and removal of the four state files (main.cpp:219-230; --keep-maps leaves them in place).
// Not built: the debug JPEGs of show_medium_result.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "APD.h"

// pair.txt -> problems (main.cpp:6-49)
void GenerateSampleList(const path &dense_folder, std::vector<Problem> &problems)
{
    const path cluster_list_path = dense_folder / path("pair.txt");
    problems.clear();
    std::ifstream file(cluster_list_path);
    std::stringstream iss;
    std::string line;
    int num_images = 0;
    std::getline(file, line);
    iss.str(line);
    iss >> num_images;
    for (int i = 0; i < num_images; ++i) {
        Problem problem;
        problem.index = i;
        iss.clear();
        std::getline(file, line);
        iss.str(line);
        iss >> problem.ref_image_id;
        problem.dense_folder = dense_folder;
        problem.result_folder = dense_folder / path("APD") / path(ToFormatIndex(problem.ref_image_id));
        std::filesystem::create_directories(problem.result_folder);
        int num_src_images = 0;
        iss.clear();
        std::getline(file, line);
        iss.str(line);
        iss >> num_src_images;
        for (int j = 0; j < num_src_images; ++j) {
            int id;
            float score;
            iss >> id >> score;
            if (score <= 0.0f) {
                continue;
            }
            problem.src_image_ids.push_back(id);
        }
        problems.push_back(problem);
    }
}

// main.cpp:72-88
int ComputeRoundNum(const std::vector<Problem> &problems)
{
    if (problems.empty()) {
        return 0;
    }
    Mat image;
    if (!ReadGrayImage(problems[0].dense_folder / path("images") / path(ToFormatIndex(problems[0].ref_image_id)), image)) {
        return 0;
    }
    int max_size = image.cols > image.rows ? image.cols : image.rows;
    int round_num = 1;
    while (max_size > 1000) {
        max_size /= 2;
        round_num++;
    }
    return round_num;
}

// main.cpp:91-138
void ProcessProblem(const Problem &problem)
{
    std::cout << "Processing image: " << std::setw(8) << std::setfill('0') << problem.ref_image_id << "..." << std::endl;
    const auto start = std::chrono::steady_clock::now();
    APD apd(problem);
    apd.InuputInitialization();
    apd.CudaSpaceInitialization();
    apd.SetDataPassHelperInCuda();
    apd.RunPatchMatch();
    const int width = apd.GetWidth(), height = apd.GetHeight();
    Mat depth(height, width, MAT_32FC1);
    Mat normal(height, width, MAT_32FC3);
    Mat pixel_states = apd.GetPixelStates();
    const float depth_min = apd.GetDepthMin(), depth_max = apd.GetDepthMax();
    ParallelFor((size_t)height, [&](size_t row) {  // main.cpp:105-115, rows on several host threads
        const int r = (int)row;
        for (int c = 0; c < width; ++c) {
            const float4 plane_hypothesis = apd.GetPlaneHypothesis(r, c);
            depth.at<float>(r, c) = plane_hypothesis.w;
            if (depth.at<float>(r, c) < depth_min || depth.at<float>(r, c) > depth_max) {
                depth.at<float>(r, c) = 0;
                pixel_states.at<uint8_t>(r, c) = UNKNOWN;
            }
            normal.at<Vec3f>(r, c) = Vec3f{{plane_hypothesis.x, plane_hypothesis.y, plane_hypothesis.z}};
        }
    });
    WriteBinMat(problem.result_folder / path("depths.dmb"), depth);
    WriteBinMat(problem.result_folder / path("normals.dmb"), normal);
    WriteBinMat(problem.result_folder / path("weak.bin"), pixel_states);
    WriteBinMat(problem.result_folder / path("selected_views.bin"), apd.GetSelectedViews());
    const auto end = std::chrono::steady_clock::now();
    std::cout << "Processing image: " << std::setw(8) << std::setfill('0') << problem.ref_image_id << " done!" << std::endl;
    std::cout << "Cost time: " << std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count() << " ms" << std::endl;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "USAGE: APD dense_folder [gpu_index] [--seed S] [--iters K] [--single-level] [--max-src N] [--keep-maps] [--no-fusion]\n";
        return EXIT_FAILURE;
    }
    const path dense_folder(argv[1]);
    std::filesystem::create_directories(dense_folder / path("APD"));
    int gpu_index = 0;
    uint64_t seed = 12345;
    int iters = 3;
    bool single_level = false, keep_maps = false, no_fusion = false;
    int max_src = 0;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) {
            seed = strtoull(argv[++i], nullptr, 10);
        } else if (!strcmp(argv[i], "--iters") && i + 1 < argc) {
            iters = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--single-level")) {
            single_level = true;
        } else if (!strcmp(argv[i], "--max-src") && i + 1 < argc) {
            max_src = atoi(argv[++i]);  // additive knob: keep only the first N sources of pair.txt (best scores first)
        } else if (!strcmp(argv[i], "--keep-maps")) {
            keep_maps = true;
        } else if (!strcmp(argv[i], "--no-fusion")) {
            no_fusion = true;
        } else if (i == 2) {
            gpu_index = atoi(argv[i]);  // main.cpp:149-153
        }
    }
    if (gpu_index >= apd_device_count()) {
        std::cerr << "Requested GPU " << gpu_index << ", found " << apd_device_count() << " device(s)\n";
        return EXIT_FAILURE;
    }
    APD::SetDevice(gpu_index);
    SetFusionDevice(gpu_index);
    std::vector<Problem> problems;
    GenerateSampleList(dense_folder, problems);
    if (problems.empty()) {
        std::cerr << "Images may error, check it!\n";
        return EXIT_FAILURE;
    }
    if (max_src > 0) {
        for (auto &problem : problems) {
            if ((int)problem.src_image_ids.size() > max_src) {
                problem.src_image_ids.resize(max_src);
            }
        }
    }
    std::cout << "There are " << problems.size() << " problems needed to be processed!" << std::endl;
    {   // every image the passes will read, decoded once, on several host threads (each pass then copies from the cache)
        std::vector<int> ids;
        for (const auto &problem : problems) {
            ids.push_back(problem.ref_image_id);
            ids.insert(ids.end(), problem.src_image_ids.begin(), problem.src_image_ids.end());
        }
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        PrefetchGrayImages(dense_folder / path("images"), ids);
    }
    const int round_num = single_level ? 1 : ComputeRoundNum(problems);
    std::cout << "Round nums: " << round_num << std::endl;
    int iteration_index = 0;
    for (int i = 0; i < round_num; ++i) {
        for (auto &problem : problems) {
            auto &params = problem.params;
            if (i == 0) {
                params.state = FIRST_INIT;
                params.use_APD = false;
            } else {
                params.state = REFINE_INIT;
                params.use_APD = true;
                params.ransac_threshold = 0.01 - i * 0.00125;
                params.rotate_time = std::min(static_cast<int>(std::pow(2, i)), 4);
            }
            params.geom_consistency = false;
            params.max_iterations = iters;
            params.weak_peak_radius = 6;
            params.seed = seed + (uint64_t)iteration_index * 7919u + (uint64_t)problem.index;
            problem.iteration = iteration_index;
            problem.show_medium_result = true;
            problem.scale_size = single_level ? 1 : static_cast<int>(std::pow(2, round_num - 1 - i));
            ProcessProblem(problem);
        }
        iteration_index++;
        for (int j = 0; j < 3; ++j) {
            for (auto &problem : problems) {
                auto &params = problem.params;
                params.state = REFINE_ITER;
                if (i == 0) {
                    params.use_APD = false;
                } else {
                    params.use_APD = true;
                    params.ransac_threshold = 0.01 - i * 0.00125;
                    params.rotate_time = std::min(static_cast<int>(std::pow(2, i)), 4);
                }
                params.geom_consistency = true;
                params.max_iterations = iters;
                params.weak_peak_radius = std::max(4 - 2 * j, 2);
                params.seed = seed + (uint64_t)iteration_index * 7919u + (uint64_t)problem.index;
                problem.iteration = iteration_index;
                problem.show_medium_result = true;
                problem.scale_size = single_level ? 1 : static_cast<int>(std::pow(2, round_num - 1 - i));
                ProcessProblem(problem);
            }
            iteration_index++;
        }
        std::cout << "Round: " << i << " done\n";
    }
    if (!no_fusion) {
        RunFusion(dense_folder, problems);  // main.cpp:219
    }
    if (!keep_maps && !no_fusion) {  // main.cpp:220-230
        for (const auto &problem : problems) {
            std::filesystem::remove(problem.result_folder / path("weak.bin"));
            std::filesystem::remove(problem.result_folder / path("depths.dmb"));
            std::filesystem::remove(problem.result_folder / path("normals.dmb"));
            std::filesystem::remove(problem.result_folder / path("selected_views.bin"));
        }
    }
    std::cout << "All done\n";
    return EXIT_SUCCESS;
}
