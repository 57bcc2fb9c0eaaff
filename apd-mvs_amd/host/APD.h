// APD.h -- C++ drop-in host for the MI355X PatchMatch path.
//
// Mirrors the reference's host boundary for this path: `Problem` / `PatchMatchParams` / `Camera`
// (main.h:47-106) and `class APD` with the same public member names, argument meaning and call
// order (APD.h:67-145, driven by ProcessProblem, main.cpp:91-138).  OpenCV and Boost are not
// available in this image, so `cv::Mat` becomes the minimal `Mat` below and
// `boost::filesystem::path` becomes `std::filesystem::path`; everything else keeps its name.
// All device work goes through the C ABI in include/apd_mi355x.h.
#ifndef APD_MI355X_HOST_APD_H_
#define APD_MI355X_HOST_APD_H_

#include <cstdint>
#include <cstring>
#include <filesystem>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/apd_mi355x.h"

using path = std::filesystem::path;

#define MAX_IMAGES APD_MAX_IMAGES
#define NEIGHBOUR_NUM APD_NEIGHBOUR_NUM

struct float4 {
    float x, y, z, w;
};

typedef apd_camera Camera;  // main.h:47-56, same layout

enum RunState { FIRST_INIT = APD_FIRST_INIT, REFINE_INIT = APD_REFINE_INIT, REFINE_ITER = APD_REFINE_ITER };
enum PixelState { WEAK = APD_WEAK, STRONG = APD_STRONG, UNKNOWN = APD_UNKNOWN };

// main.h:75-94 (defaults included) + the additive seed knob
struct PatchMatchParams {
    int max_iterations = 3;
    int num_images = 5;
    float sigma_spatial = 5.0f;
    float sigma_color = 3.0f;
    int top_k = 4;
    float depth_min = 0.0f;
    float depth_max = 1.0f;
    bool geom_consistency = false;
    int strong_radius = 5;
    int strong_increment = 2;
    int weak_radius = 5;
    int weak_increment = 5;
    bool use_APD = true;
    int weak_peak_radius = 2;
    int rotate_time = 4;
    float ransac_threshold = 0.005f;
    float geom_factor = 0.2f;
    RunState state = FIRST_INIT;
    uint64_t seed = 12345;  // the reference seeds cuRAND with clock64() (APD.cu:803)
};

// main.h:96-106
struct Problem {
    int index = 0;
    int ref_image_id = 0;
    std::vector<int> src_image_ids;
    path dense_folder;
    path result_folder;
    int scale_size = 1;
    PatchMatchParams params;
    bool show_medium_result = false;
    int iteration = 0;
};

// OpenCV type codes used by the .dmb/.bin files (APD.cpp:3-49)
enum MatType { MAT_8UC1 = 0, MAT_32SC1 = 4, MAT_32FC1 = 5, MAT_32FC3 = 21 };

// Minimal dense row-major matrix standing in for cv::Mat (no padding: step == cols * elemSize).
struct Mat {
    int rows = 0, cols = 0, type = MAT_8UC1;
    std::shared_ptr<std::vector<uint8_t>> buf;

    Mat() = default;
    Mat(int r, int c, int t) { create(r, c, t); }
    static size_t elemSizeOf(int t)
    {
        switch (t) {
        case MAT_8UC1: return 1;
        case MAT_32SC1: return 4;
        case MAT_32FC1: return 4;
        case MAT_32FC3: return 12;
        default: return 0;
        }
    }
    void create(int r, int c, int t)
    {
        rows = r;
        cols = c;
        type = t;
        buf = std::make_shared<std::vector<uint8_t>>((size_t)r * c * elemSizeOf(t), 0);
    }
    size_t elemSize() const { return elemSizeOf(type); }
    size_t step() const { return (size_t)cols * elemSize(); }
    bool empty() const { return !buf || buf->empty(); }
    uint8_t *data() { return buf ? buf->data() : nullptr; }
    const uint8_t *data() const { return buf ? buf->data() : nullptr; }
    template <typename T> T &at(int r, int c) { return reinterpret_cast<T *>(buf->data())[(size_t)r * cols + c]; }
    template <typename T> const T &at(int r, int c) const { return reinterpret_cast<const T *>(buf->data())[(size_t)r * cols + c]; }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(buf->data()) + (size_t)r * cols; }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(buf->data()) + (size_t)r * cols; }
    Mat clone() const
    {
        Mat m;
        m.rows = rows;
        m.cols = cols;
        m.type = type;
        m.buf = buf ? std::make_shared<std::vector<uint8_t>>(*buf) : nullptr;
        return m;
    }
};

struct Vec3f {
    float v[3];
    float &operator[](int i) { return v[i]; }
    const float &operator[](int i) const { return v[i]; }
};

// free functions of APD.h:15-34 that belong to this path
bool ReadBinMat(const path &mat_path, Mat &mat);          // APD.cpp:3-28
bool WriteBinMat(const path &mat_path, const Mat &mat);   // APD.cpp:30-49
bool ReadCamera(const path &cam_path, Camera &cam);       // APD.cpp:51-92 (TAT & ETH variant)
std::string ToFormatIndex(int index);                     // APD.cpp:350-354
template <typename TYPE> void RescaleMatToTargetSize(const Mat &src, Mat &dst, int target_width, int target_height);  // APD.cpp:752-774

// image input: `images/%08d.jpg` decoded to 8-bit grey (APD.cpp:410-413); `.pgm` / `.pfm` accepted too
bool ReadGrayImage(const path &image_path_without_ext, Mat &image_float);
bool ReadGrayImageShared(const path &image_path_without_ext, Mat &image_float);  // the process cache's own matrix: read only
// the same file as cv::imread(IMREAD_COLOR) returns it (fusion colours, APD.cpp:859): MAT_32FC3, blue first
bool ReadColorImage(const path &image_path_without_ext, Mat &image_bgr);
// host-side helpers of the drop-in (not in the reference): a small thread pool and a parallel warm-up of the image cache
void ParallelFor(size_t count, const std::function<void(size_t)> &job, unsigned max_threads = 0);
void PrefetchGrayImages(const path &image_folder, const std::vector<int> &ids);
// cv::resize(float, INTER_LINEAR) restated (APD.cpp:474)
void ResizeLinear(const Mat &src, Mat &dst, int new_cols, int new_rows);

// APD.h:34 -- consistency check + merge of the final depth maps into <dense>/APD/APD.ply (device fusion apd_fuse_views)
void RunFusion(const path &dense_folder, const std::vector<Problem> &problems);
void SetFusionDevice(int device);  // additive: HIP device of the fusion (default 0)

class APD {
public:
    APD(const Problem &problem);
    ~APD();

    void InuputInitialization();      // (sic) APD.cpp:399-583
    void CudaSpaceInitialization();   // APD.cpp:585-671
    void SetDataPassHelperInCuda();   // APD.cpp:673-699 (nothing left to do: the C ABI owns the device struct)
    void RunPatchMatch();             // APD.cu:2386-2495
    float4 GetPlaneHypothesis(int r, int c);
    Mat GetPixelStates();
    Mat GetSelectedViews();
    int GetWidth();
    int GetHeight();
    float GetDepthMin();
    float GetDepthMax();

    // additive: which HIP device this object uses (reference: process-global cudaSetDevice, main.cpp:153)
    static void SetDevice(int device);
    // additive: the ids of the views this run reconstructs (the reference ids of pair.txt).  A source outside the set is a
    // source-only view: the geometric term gets a zero depth map for it, whatever an earlier run left in <dense>/APD/.
    static void SetReconstructedViews(const std::vector<int> &ref_image_ids);

private:
    // the steps of InuputInitialization
    std::vector<int> LoadViewSet();
    void ApplyPyramidLevel(const std::vector<int> &ids);
    void LoadGeometricDepths();
    void LoadWeakMap();
    void LoadPriorState();

    int num_images = 0;
    int width = 0;
    int height = 0;
    Problem problem;
    std::vector<Mat> images;
    std::vector<Mat> depths;
    std::vector<Camera> cameras;
    int weak_count = 0;
    Mat weak_info_host;
    Mat neighbours_map_host;
    std::vector<float4> plane_hypotheses_host;
    PatchMatchParams params_host;
    Mat selected_views_host;
    bool has_prior = false;
    apd_handle handle = nullptr;
};

#endif  // APD_MI355X_HOST_APD_H_
