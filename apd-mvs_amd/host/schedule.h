// schedule.h -- what the two schedulers of the drop-in host share: the command-line options, the pass table of the
// reference driver (main.cpp:168-215) and the per-pass parameters of a problem.  host/main.cpp runs the table view by view
// through files like the reference (Gauss-Seidel over views); host/multi_device.cpp runs it in memory on several devices.
#ifndef APD_MI355X_HOST_SCHEDULE_H_
#define APD_MI355X_HOST_SCHEDULE_H_

#include <algorithm>
#include <cstdint>
#include <vector>

#include "APD.h"

struct Options {
    path dense_folder;
    int gpu_index = 0;
    std::vector<int> devices;   // "0,1,2,3": one scheduler rank per entry (an entry may repeat: two ranks on one device)
    bool jacobi = false;        // the in-memory scheduler of host/multi_device.cpp even with a single device
    int ranks_per_device = 0;   // --ranks N: scheduler ranks on a single device (0: by frame size, host/multi_device.cpp)
    bool files = false;         // --files: state moves between passes through the four files per view, as in the reference
    bool in_memory = false;     // --in-memory: the in-memory scheduler on one device in the reference's own order (one rank, a view reads the
                                // depth maps its sources have at that moment): the file-based driver's bytes without the files
    bool use_rccl = true;       // --no-rccl: exchange maps with direct copies
    bool exchange_device_sync = false;  // --exchange-device-sync: the per-pass exchange synchronises every device first, as in rounds 2-4 (A/B measurements)
    bool force_rccl = false;    // --rccl: RCCL even for a single rank (which has nothing to exchange between devices and uses direct copies otherwise)
    uint64_t seed = 12345;
    int iters = 3;          // PatchMatchParams::max_iterations of every pass (reference: 3)
    int max_src = 0;        // > 0: keep only the first N sources of each pair.txt entry (they are sorted by score)
    bool single_level = false, keep_maps = false, no_fusion = false;
    bool copy_images = false;         // --copy-images: handles copy and pack their images per (view, pass) instead of sharing the level images (A/B)
    bool clean_exit = false;          // --clean-exit: return from main() instead of _Exit (exit handlers run: profilers)
    double scheduler_free_gb = 0;     // --scheduler-free-gb X: RunMultiDevice's own fit test counts at most X GB of free device memory (main()'s
                                      // choice of scheduler still uses the device's figure: this is how a test reaches the fall-back to files)
    bool late_fusion_inputs = false;  // --late-fusion-inputs: colour decode + upload after the passes instead of behind them (A/B measurements)
};


// One pass over all views.  round_num pyramid levels, coarse to fine; per level one photometric pass and three
// geometric ones (main.cpp:168-215).
// true when a run over this device list sets RCCL up (several physical devices, or --rccl); the set-up blocks before the first pass
// (round 5's --async-rccl measured slower and is gone: profiles/r05/ab_rccl_async_tt24.txt)
inline bool WantsRccl(const Options &opt)
{
    bool several = false;
    for (size_t i = 1; i < opt.devices.size(); ++i) {
        several = several || opt.devices[i] != opt.devices[0];
    }
    return opt.use_rccl && (several || opt.force_rccl);
}

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

inline int RoundNum(int width, int height)  // main.cpp:72-88: halve until the longer side is <= 1000
{
    int rounds = 1;
    for (int longest = std::max(width, height); longest > 1000; longest /= 2) {
        ++rounds;
    }
    return rounds;
}

inline std::vector<Pass> BuildSchedule(int round_num, bool single_level)
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
inline void Configure(Problem &problem, const Pass &pass, const Options &o)
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

static const char *const kStateFiles[4] = {"depths.dmb", "normals.dmb", "weak.bin", "selected_views.bin"};


// host/multi_device.cpp: every pass on all listed devices (views round-robin), state resident on the devices, depth maps
// all-gathered after every pass.  Returns the process exit code.
int RunMultiDevice(const Options &opt, std::vector<Problem> &problems);

// Views in flight per device by frame size, and the device bytes per pixel (finest level) the in-memory scheduler keeps resident
// on its busiest device (host/multi_device.cpp).
int DefaultLanes(size_t pixels);
int InMemoryLanes(const Options &opt, int width, int height, int num_views, int num_ranks, bool distinct_devices);
double InMemoryBytesPerPixel(int num_images, int num_views, int num_ranks, int lanes, int max_sources, double *passes_out = nullptr,
                             double *final_out = nullptr, bool fusion_prefetch = false);
// The scheduler's fit test, ONE function for main()'s choice of scheduler and for RunMultiDevice's own check: bytes the run keeps resident on
// `device` (its busiest one) against 90 % of the free device memory, the latter capped by --scheduler-free-gb.
struct InMemoryFit {
    bool have_memory = false;            // hipMemGetInfo answered
    bool fits = true;                    // need <= 0.9 * free (true when the device did not answer: the allocations will tell)
    bool release_before_fusion = true;   // no room for the passes' buffers and the final maps at once
    double need_bytes = 0, free_bytes = 0;
};
InMemoryFit TestInMemoryFit(const Options &opt, int device, int width, int height, int num_images, int num_views, int num_ranks, int lanes, int max_sources);
constexpr int kExitDoesNotFit = 75;  // RunMultiDevice: the folder does not fit the in-memory scheduler (nothing has been run or written)

// APD.h:34 with the maps already in memory (index = problem index); RunFusion reads them from the result folders instead
struct FinalMaps {
    Mat depth, normal, weak;
};
void RunFusionWithMaps(const path &dense_folder, const std::vector<Problem> &problems, const std::vector<FinalMaps> *maps);
// ... and with the final maps resident on `device` (cols x rows each; depth, normal = 3 floats per pixel, weak): nothing is
// downloaded.  StartFusionInputs decodes the colour images, reads cameras and masks and uploads them on a background thread
// (call it before the passes); RunFusionOnDevice joins it and fuses; CancelFusionInputs joins and frees without fusing.
struct FusionPrefetch;
FusionPrefetch *StartFusionInputs(const path &dense_folder, const std::vector<Problem> &problems, int device, int cols, int rows, unsigned threads);
void RunFusionOnDevice(FusionPrefetch *inputs, const std::vector<const float *> &depths, const std::vector<const float *> &normals,
                       const std::vector<const uint8_t *> &weaks);
void CancelFusionInputs(FusionPrefetch *inputs);
bool FusionInputsFailed(const FusionPrefetch *inputs, std::string *why);

#endif  // APD_MI355X_HOST_SCHEDULE_H_
