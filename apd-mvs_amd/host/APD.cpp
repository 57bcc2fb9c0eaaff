// APD.cpp -- host side of the drop-in `APD` class over the C ABI (see APD.h).
#include "APD.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

bool DecodeJpegGray(const uint8_t *data, size_t size, std::vector<uint8_t> &gray, int &width, int &height);
bool DecodeJpegBGR(const uint8_t *data, size_t size, std::vector<uint8_t> &bgr, int &width, int &height);

// ---------------------------------------------------------------------------------------------
// file formats (APD.cpp:3-92, 350-354)
// ---------------------------------------------------------------------------------------------

bool ReadBinMat(const path &mat_path, Mat &mat)
{
    std::ifstream in(mat_path, std::ios_base::binary);
    if (!in) {  // the reference tests in.bad(), which a failed open does not set; here it is an error
        std::cerr << "Error opening file: " << mat_path << std::endl;
        return false;
    }
    int32_t version = 0, rows = 0, cols = 0, type = 0;
    in.read((char *)&version, sizeof(int32_t));
    in.read((char *)&rows, sizeof(int32_t));
    in.read((char *)&cols, sizeof(int32_t));
    in.read((char *)&type, sizeof(int32_t));
    if (version != 1) {
        std::cerr << "Version error: " << mat_path << std::endl;
        return false;
    }
    if (rows < 0 || cols < 0 || Mat::elemSizeOf(type) == 0) {
        std::cerr << "Unsupported matrix header in " << mat_path << std::endl;
        return false;
    }
    mat.create(rows, cols, type);
    in.read((char *)mat.data(), (std::streamsize)(mat.step() * (size_t)mat.rows));
    return (bool)in;
}

bool WriteBinMat(const path &mat_path, const Mat &mat)
{
    std::ofstream out(mat_path, std::ios_base::binary);
    if (!out) {
        std::cout << "Error opening file: " << mat_path << std::endl;
        return false;
    }
    const int32_t version = 1, rows = mat.rows, cols = mat.cols, type = mat.type;
    out.write((const char *)&version, sizeof(int32_t));
    out.write((const char *)&rows, sizeof(int32_t));
    out.write((const char *)&cols, sizeof(int32_t));
    out.write((const char *)&type, sizeof(int32_t));
    out.write((const char *)mat.data(), (std::streamsize)(mat.step() * (size_t)mat.rows));
    return (bool)out;
}

bool ReadCamera(const path &cam_path, Camera &cam)
{
    std::ifstream in(cam_path);
    if (!in) {
        return false;
    }
    std::string token;
    in >> token;  // "extrinsic"
    for (int i = 0; i < 3; ++i) {
        in >> cam.R[3 * i + 0] >> cam.R[3 * i + 1] >> cam.R[3 * i + 2] >> cam.t[i];
    }
    float last_row[4];
    in >> last_row[0] >> last_row[1] >> last_row[2] >> last_row[3];
    in >> token;  // "intrinsic"
    for (int i = 0; i < 3; ++i) {
        in >> cam.K[3 * i + 0] >> cam.K[3 * i + 1] >> cam.K[3 * i + 2];
    }
    // camera centre in world coordinates, evaluated in double (APD.cpp:73-77)
    for (int j = 0; j < 3; ++j) {
        cam.c[j] = -float(double(cam.R[0 + j]) * double(cam.t[0]) + double(cam.R[3 + j]) * double(cam.t[1]) +
                          double(cam.R[6 + j]) * double(cam.t[2]));
    }
    // TAT & ETH layout: depth_min interval depth_num depth_max (APD.cpp:80-82)
    float depth_num = 0, interval = 0;
    in >> cam.depth_min >> interval >> depth_num >> cam.depth_max;
    return !in.fail();
}

std::string ToFormatIndex(int index)
{
    std::stringstream ss;
    ss << std::setw(8) << std::setfill('0') << index;
    return ss.str();
}

// Nearest-neighbour resampling of prior state; keeps the reference's swapped scale factors
// (row / scale_x, column / scale_y; APD.cpp:766-767).
template <typename TYPE> void RescaleMatToTargetSize(const Mat &src, Mat &dst, int target_width, int target_height)
{
    if (src.cols == target_width && src.rows == target_height) {
        return;
    }
    const float scale_x = target_width / static_cast<float>(src.cols);
    const float scale_y = target_height / static_cast<float>(src.rows);
    const Mat src_clone = src.clone();
    Mat out(target_height, target_width, src.type);
    for (int r = 0; r < target_height; ++r) {
        for (int c = 0; c < target_width; ++c) {
            const int o_r = static_cast<int>(r / scale_x);
            const int o_c = static_cast<int>(c / scale_y);
            if (o_r < 0 || o_c < 0 || o_r >= src_clone.rows || o_c >= src_clone.cols) {
                continue;
            }
            out.at<TYPE>(r, c) = src_clone.at<TYPE>(o_r, o_c);
        }
    }
    dst = out;
}
template void RescaleMatToTargetSize<float>(const Mat &, Mat &, int, int);
template void RescaleMatToTargetSize<uint8_t>(const Mat &, Mat &, int, int);
template void RescaleMatToTargetSize<uint32_t>(const Mat &, Mat &, int, int);
template void RescaleMatToTargetSize<Vec3f>(const Mat &, Mat &, int, int);

// ---------------------------------------------------------------------------------------------
// images
// ---------------------------------------------------------------------------------------------

static bool read_file(const path &p, std::vector<uint8_t> &bytes)
{
    std::ifstream in(p, std::ios_base::binary);
    if (!in) {
        return false;
    }
    in.seekg(0, std::ios_base::end);
    const std::streamoff n = in.tellg();
    in.seekg(0, std::ios_base::beg);
    bytes.resize((size_t)n);
    in.read((char *)bytes.data(), n);
    return (bool)in;
}

static bool read_pgm(const std::vector<uint8_t> &b, Mat &out)
{
    // binary PGM "P5 <w> <h> <maxval>\n<data>", 8 bit
    size_t pos = 0;
    auto token = [&]() {
        std::string t;
        while (pos < b.size()) {
            if (b[pos] == '#') {
                while (pos < b.size() && b[pos] != '\n') {
                    ++pos;
                }
            } else if (isspace(b[pos])) {
                ++pos;
            } else {
                break;
            }
        }
        while (pos < b.size() && !isspace(b[pos])) {
            t.push_back((char)b[pos++]);
        }
        return t;
    };
    if (token() != "P5") {
        return false;
    }
    const int w = atoi(token().c_str()), h = atoi(token().c_str()), maxval = atoi(token().c_str());
    ++pos;  // single whitespace after maxval
    if (w <= 0 || h <= 0 || maxval != 255 || pos + (size_t)w * h > b.size()) {
        return false;
    }
    out.create(h, w, MAT_32FC1);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        out.ptr<float>()[i] = (float)b[pos + i];
    }
    return true;
}

// Decoded grey images of this process, by path.  The reference decodes every image again for every (view, pass)
// (APD.cpp:410-427): 4 * round_num * (1 + sources) decodes per view.  With the PatchMatch pass itself down to a fraction
// of a second that would be most of the run time, so each file is decoded once and handed out as a copy.  The cache
// holds at most APD_IMAGE_CACHE_MB megabytes (default 16384; 0 disables it); images are not expected to change on
// disk while the program runs.
static bool read_gray_image_uncached(const path &stem, Mat &image_float);

static bool read_gray_image_cached(const path &stem, Mat &image_float, bool share);

bool ReadGrayImage(const path &stem, Mat &image_float) { return read_gray_image_cached(stem, image_float, false); }

// The cached matrix itself, not a copy, for callers that only read it (the in-memory scheduler keeps every full-resolution
// image for the whole run and CheckImages only looks at the size): saves one 4 B/px copy per image and caller.
bool ReadGrayImageShared(const path &stem, Mat &image_float) { return read_gray_image_cached(stem, image_float, true); }

static bool read_gray_image_cached(const path &stem, Mat &image_float, bool share)
{
    static std::unordered_map<std::string, Mat> cache;
    static size_t cached_bytes = 0;
    static const size_t cap_bytes = [] {
        const char *e = getenv("APD_IMAGE_CACHE_MB");
        return (size_t)(e ? atoll(e) : 16384) << 20;
    }();
    static std::mutex cache_mutex;  // PrefetchGrayImages and RunFusion decode on several threads
    const std::string key = stem.string();
    {
        std::lock_guard<std::mutex> lock(cache_mutex);
        auto it = cache.find(key);
        if (it != cache.end()) {
            image_float = share ? it->second : it->second.clone();
            return true;
        }
    }
    if (!read_gray_image_uncached(stem, image_float)) {
        return false;
    }
    const size_t bytes = (size_t)image_float.rows * image_float.step();
    std::lock_guard<std::mutex> lock(cache_mutex);
    if (cached_bytes + bytes <= cap_bytes && cache.find(key) == cache.end()) {
        cache.emplace(key, share ? image_float : image_float.clone());
        cached_bytes += bytes;
    }
    return true;
}

// Runs job(0) .. job(count - 1) on up to `max_threads` host threads (0: one per core, at most 32).
void ParallelFor(size_t count, const std::function<void(size_t)> &job, unsigned max_threads)
{
    unsigned n = max_threads ? max_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    n = (unsigned)std::min<size_t>(n, count);
    if (n <= 1) {
        for (size_t i = 0; i < count; ++i) {
            job(i);
        }
        return;
    }
    std::vector<std::thread> pool;
    std::mutex next_mutex;
    size_t next = 0;
    for (unsigned t = 0; t < n; ++t) {
        pool.emplace_back([&] {
            for (;;) {
                size_t i;
                {
                    std::lock_guard<std::mutex> lock(next_mutex);
                    if (next >= count) {
                        return;
                    }
                    i = next++;
                }
                job(i);
            }
        });
    }
    for (auto &th : pool) {
        th.join();
    }
}

// Decodes the given images into the process cache on several threads, so that the first pass does not wait for one
// decode after another (the reference decodes each of them serially, once per view and pass).
void PrefetchGrayImages(const path &image_folder, const std::vector<int> &ids)
{
    ParallelFor(ids.size(), [&](size_t i) {
        Mat scratch;
        ReadGrayImage(image_folder / path(ToFormatIndex(ids[i])), scratch);
    }, 0);
}

static bool read_gray_image_uncached(const path &stem, Mat &image_float)
{
    std::vector<uint8_t> bytes;
    path p = stem;
    p += ".jpg";
    if (read_file(p, bytes)) {
        std::vector<uint8_t> gray;
        int w = 0, h = 0;
        if (!DecodeJpegGray(bytes.data(), bytes.size(), gray, w, h)) {
            std::cerr << "Unsupported JPEG (only baseline sequential Huffman is built): " << p << std::endl;
            return false;
        }
        image_float.create(h, w, MAT_32FC1);  // image_uint.convertTo(image_float, CV_32FC1), APD.cpp:411-413
        for (size_t i = 0; i < gray.size(); ++i) {
            image_float.ptr<float>()[i] = (float)gray[i];
        }
        return true;
    }
    p = stem;
    p += ".pgm";
    if (read_file(p, bytes)) {
        return read_pgm(bytes, image_float);
    }
    std::cerr << "Can't read image: " << stem << ".{jpg,pgm}" << std::endl;
    return false;
}

// cv::imread(IMREAD_COLOR) of `images/%08d.jpg` as the fusion reads it (APD.cpp:859): 3 x float per pixel, blue first.
// `.pgm` (grey: B = G = R) and binary `.ppm` (P6) are accepted too.
bool ReadColorImage(const path &stem, Mat &image_bgr)
{
    std::vector<uint8_t> bytes;
    path p = stem;
    p += ".jpg";
    if (read_file(p, bytes)) {
        std::vector<uint8_t> bgr;
        int w = 0, h = 0;
        if (!DecodeJpegBGR(bytes.data(), bytes.size(), bgr, w, h)) {
            std::cerr << "Unsupported JPEG (only baseline sequential Huffman is built): " << p << std::endl;
            return false;
        }
        image_bgr.create(h, w, MAT_32FC3);
        for (size_t i = 0; i < bgr.size(); ++i) {
            image_bgr.ptr<float>()[i] = (float)bgr[i];
        }
        return true;
    }
    p = stem;
    p += ".ppm";
    if (read_file(p, bytes)) {  // "P6 <w> <h> 255\n" + RGB triples
        size_t pos = 0;
        auto token = [&]() {
            std::string t;
            while (pos < bytes.size() && (isspace(bytes[pos]) || bytes[pos] == '#')) {
                if (bytes[pos] == '#') {
                    while (pos < bytes.size() && bytes[pos] != '\n') {
                        ++pos;
                    }
                } else {
                    ++pos;
                }
            }
            while (pos < bytes.size() && !isspace(bytes[pos])) {
                t.push_back((char)bytes[pos++]);
            }
            return t;
        };
        if (token() != "P6") {
            return false;
        }
        const int w = atoi(token().c_str()), h = atoi(token().c_str()), maxval = atoi(token().c_str());
        ++pos;
        if (w <= 0 || h <= 0 || maxval != 255 || pos + (size_t)w * h * 3 > bytes.size()) {
            return false;
        }
        image_bgr.create(h, w, MAT_32FC3);
        float *o = image_bgr.ptr<float>();
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            o[3 * i + 0] = (float)bytes[pos + 3 * i + 2];
            o[3 * i + 1] = (float)bytes[pos + 3 * i + 1];
            o[3 * i + 2] = (float)bytes[pos + 3 * i + 0];
        }
        return true;
    }
    Mat grey;
    p = stem;
    p += ".pgm";
    if (read_file(p, bytes) && read_pgm(bytes, grey)) {
        image_bgr.create(grey.rows, grey.cols, MAT_32FC3);
        for (size_t i = 0; i < (size_t)grey.rows * grey.cols; ++i) {
            image_bgr.ptr<float>()[3 * i] = image_bgr.ptr<float>()[3 * i + 1] = image_bgr.ptr<float>()[3 * i + 2] = grey.ptr<float>()[i];
        }
        return true;
    }
    std::cerr << "Can't read image: " << stem << ".{jpg,ppm,pgm}" << std::endl;
    return false;
}

// cv::resize(src, dst, Size(new_cols, new_rows), 0, 0, INTER_LINEAR) on a float image:
// fx = (dx + 0.5) * (src/dst) - 0.5, taps floor(fx) and floor(fx)+1 clamped, float weights.
void ResizeLinear(const Mat &src, Mat &dst, int new_cols, int new_rows)
{
    Mat out(new_rows, new_cols, MAT_32FC1);
    const double sx = (double)src.cols / new_cols, sy = (double)src.rows / new_rows;
    std::vector<int> x0(new_cols), x1(new_cols);
    std::vector<float> ax(new_cols);
    for (int dx = 0; dx < new_cols; ++dx) {
        float fx = (float)((dx + 0.5) * sx - 0.5);
        int ix = (int)std::floor(fx);
        fx -= ix;
        if (ix < 0) {
            ix = 0;
            fx = 0;
        }
        if (ix >= src.cols - 1) {
            ix = src.cols - 1;
            fx = 0;
        }
        x0[dx] = ix;
        x1[dx] = ix + 1 < src.cols ? ix + 1 : ix;
        ax[dx] = fx;
    }
    for (int dy = 0; dy < new_rows; ++dy) {
        float fy = (float)((dy + 0.5) * sy - 0.5);
        int iy = (int)std::floor(fy);
        fy -= iy;
        if (iy < 0) {
            iy = 0;
            fy = 0;
        }
        if (iy >= src.rows - 1) {
            iy = src.rows - 1;
            fy = 0;
        }
        const float *r0 = src.ptr<float>(iy);
        const float *r1 = src.ptr<float>(iy + 1 < src.rows ? iy + 1 : iy);
        float *o = out.ptr<float>(dy);
        for (int dx = 0; dx < new_cols; ++dx) {
            const float a = ax[dx];
            const float top = r0[x0[dx]] * (1.f - a) + r0[x1[dx]] * a;
            const float bot = r1[x0[dx]] * (1.f - a) + r1[x1[dx]] * a;
            o[dx] = top * (1.f - fy) + bot * fy;
        }
    }
    dst = out;
}

// ---------------------------------------------------------------------------------------------
// class APD
// ---------------------------------------------------------------------------------------------

static int g_device = -1;
void APD::SetDevice(int device) { g_device = device; }

static std::unordered_set<int> g_reconstructed;
static bool g_have_reconstructed = false;
void APD::SetReconstructedViews(const std::vector<int> &ref_image_ids)
{
    g_reconstructed.clear();
    g_reconstructed.insert(ref_image_ids.begin(), ref_image_ids.end());
    g_have_reconstructed = true;
}

static void ApdSafeCall(int rc, const char *what)
{
    if (rc != APD_OK) {  // reference: CudaSafeCall -> print + exit (APD.cpp:315-323)
        std::cerr << what << " failed: " << apd_last_error() << std::endl;
        exit(EXIT_FAILURE);
    }
}

APD::APD(const Problem &problem)
{
    params_host = problem.params;
    this->problem = problem;
}

APD::~APD()
{
    if (handle) {
        apd_destroy(handle);
    }
}

// InuputInitialization (sic, APD.cpp:399-583) in five steps; the log lines are the reference's.
void APD::InuputInitialization()
{
    const std::vector<int> ids = LoadViewSet();
    ApplyPyramidLevel(ids);
    std::cout << "Image size: " << width << " * " << height << std::endl;
    // The device path keeps every view as width x height floats (apd_upload_views copies exactly that much from every
    // pointer): a source image of another size than the reference image is refused here, with the message the reference's
    // CheckImages prints for mismatching reference images (main.cpp:51-70, :158).
    for (int i = 0; i < num_images; ++i) {
        const Mat &im = images[i];
        if (im.cols != width || im.rows != height || im.type != MAT_32FC1) {
            std::cerr << "Images may error, check it! (image " << ids[i] << " is " << im.cols << " * " << im.rows << ", expected " << width
                      << " * " << height << ")\n";
            exit(EXIT_FAILURE);
        }
    }
    LoadGeometricDepths();
    LoadWeakMap();
    LoadPriorState();
}

// Reference image first, then the sources in pair.txt order, with their cameras (APD.cpp:409-461); the depth search range
// is [0.6 depth_min, 1.2 depth_max] of the reference camera.
std::vector<int> APD::LoadViewSet()
{
    std::vector<int> ids{problem.ref_image_id};
    ids.insert(ids.end(), problem.src_image_ids.begin(), problem.src_image_ids.end());
    if (ids.size() > MAX_IMAGES) {
        std::cerr << "Can't process so much images: " << ids.size() << std::endl;
        exit(EXIT_FAILURE);
    }
    const path image_folder = problem.dense_folder / path("images");
    const path cam_folder = problem.dense_folder / path("cams");
    images.assign(ids.size(), Mat());
    cameras.assign(ids.size(), Camera());
    for (size_t i = 0; i < ids.size(); ++i) {
        if (!ReadGrayImage(image_folder / path(ToFormatIndex(ids[i])), images[i])) {
            exit(EXIT_FAILURE);
        }
    }
    width = images[0].cols;
    height = images[0].rows;
    for (size_t i = 0; i < ids.size(); ++i) {
        Camera &cam = cameras[i];
        memset(&cam, 0, sizeof(cam));
        if (!ReadCamera(cam_folder / path(ToFormatIndex(ids[i]) + "_cam.txt"), cam)) {
            std::cerr << "Can't read camera " << ids[i] << std::endl;
            exit(EXIT_FAILURE);
        }
        cam.width = width;
        cam.height = height;
    }
    num_images = (int)ids.size();
    params_host.num_images = num_images;
    params_host.depth_min = cameras[0].depth_min * 0.6f;
    params_host.depth_max = cameras[0].depth_max * 1.2f;
    std::cout << "Read images and camera done\n";
    std::cout << "Depth range: " << params_host.depth_min << " " << params_host.depth_max << std::endl;
    std::cout << "Num images: " << params_host.num_images << std::endl;
    return ids;
}

// Images and intrinsics of pyramid level `scale_size` (APD.cpp:464-488): new size = round(old / scale), cv::resize
// INTER_LINEAR on the float image, K scaled with the actual ratios.  The resampled image of a file is the same for every
// (view, pass) of the run: computed once per process (see ReadGrayImage for the rationale) and handed out as a copy.
void APD::ApplyPyramidLevel(const std::vector<int> &ids)
{
    if (problem.scale_size == 1) {
        return;
    }
    static std::unordered_map<std::string, Mat> level_cache;
    static const bool cache_on = getenv("APD_IMAGE_CACHE_MB") == nullptr || atoll(getenv("APD_IMAGE_CACHE_MB")) > 0;
    const path image_folder = problem.dense_folder / path("images");
    const float factor = 1.0f / (float)(problem.scale_size);
    for (int i = 0; i < num_images; ++i) {
        const int old_cols = images[i].cols, old_rows = images[i].rows;
        const int new_cols = (int)std::round(old_cols * factor);
        const int new_rows = (int)std::round(old_rows * factor);
        const std::string key = (image_folder / path(ToFormatIndex(ids[i]))).string() + "@" + std::to_string(new_cols) + "x" + std::to_string(new_rows);
        const auto hit = level_cache.find(key);
        if (hit != level_cache.end()) {
            images[i] = hit->second.clone();
        } else {
            Mat scaled;
            ResizeLinear(images[i], scaled, new_cols, new_rows);
            if (cache_on) {
                level_cache.emplace(key, scaled.clone());
            }
            images[i] = scaled;
        }
        const float scale_x = new_cols / static_cast<float>(old_cols);
        const float scale_y = new_rows / static_cast<float>(old_rows);
        Camera &cam = cameras[i];
        cam.K[0] *= scale_x;
        cam.K[2] *= scale_x;
        cam.K[4] *= scale_y;
        cam.K[5] *= scale_y;
        cam.width = width = new_cols;
        cam.height = height = new_rows;
    }
    std::cout << "Scale images and cameras done\n";
}

// Depth maps of the previous pass for the geometric term (APD.cpp:492-510): the view's own, then one per source, resampled
// to this level with the reference's nearest-neighbour rule.
void APD::LoadGeometricDepths()
{
    depths.clear();
    if (!params_host.geom_consistency) {
        return;
    }
    auto load = [&](const path &folder, bool reconstructed) {
        Mat depth;
        if (!reconstructed) {
            // A source that is not reconstructed itself (no pair.txt entry of its own): the reference reads a file that is not
            // there and goes on with an unspecified matrix (APD.cpp:497-506); here the view has no estimate anywhere (depth 0),
            // which the geometric term prices like any pixel without a depth.  Decided by membership in the set of
            // reference views (SetReconstructedViews), never by what an earlier run left on the disk: main() creates a
            // result folder for every problem, so a stale folder -- or, after --keep-maps, a stale depths.dmb -- of a
            // previous, larger run must not turn a source-only view into one with a depth map (the in-memory scheduler
            // and pipeline.py decide by membership too, and all three must write the same bytes).
            depth.create(height, width, MAT_32FC1);  // zero-filled
            return depth;
        }
        ReadBinMat(folder / path("depths.dmb"), depth);
        if (depth.empty()) {
            std::cerr << "Missing depth map of a previous pass\n";
            exit(EXIT_FAILURE);
        }
        if (depth.type != MAT_32FC1) {
            std::cerr << "depths.dmb of a previous pass is not a float map\n";
            exit(EXIT_FAILURE);
        }
        if (depth.cols != width || depth.rows != height) {
            RescaleMatToTargetSize<float>(depth, depth, width, height);
        }
        return depth;
    };
    depths.push_back(load(problem.result_folder, true));
    for (int src_idx : problem.src_image_ids) {
        // without a set (a caller that drives the class directly) every source is expected to have a map, as in the reference
        const bool reconstructed = !g_have_reconstructed || g_reconstructed.count(src_idx) != 0;
        depths.push_back(load(problem.dense_folder / path("APD") / path(ToFormatIndex(src_idx)), reconstructed));
    }
}

// weak.bin of the previous pass when the adaptive patches are on (APD.cpp:513-548); every pixel STRONG otherwise.
void APD::LoadWeakMap()
{
    weak_count = 0;
    if (!params_host.use_APD) {
        weak_info_host.create(height, width, MAT_8UC1);
        memset(weak_info_host.data(), STRONG, (size_t)width * height);
        return;
    }
    const path weak_info_path = problem.result_folder / path("weak.bin");
    if (!std::filesystem::exists(weak_info_path)) {
        std::cerr << "Can't find weak info file: " << weak_info_path.string() << std::endl;
        exit(EXIT_FAILURE);
    }
    ReadBinMat(weak_info_path, weak_info_host);
    if (weak_info_host.empty() || weak_info_host.type != MAT_8UC1) {
        std::cerr << "Unreadable or mistyped weak info file: " << weak_info_path.string() << std::endl;
        exit(EXIT_FAILURE);
    }
    if (weak_info_host.cols != width || weak_info_host.rows != height) {
        std::cerr << "Weak info doesn't match the images' size!\n";
        RescaleMatToTargetSize<uint8_t>(weak_info_host, weak_info_host, width, height);
        std::cout << "Scale done\n";
    }
    const uint8_t *w = weak_info_host.ptr<uint8_t>();
    for (size_t k = 0, n = (size_t)width * height; k < n; ++k) {
        weak_count += w[k] == WEAK;
    }
    std::cout << "Weak count: " << weak_count << " / " << width * height << " = " << (float)weak_count / (float)(width * height) * 100 << "%"
              << std::endl;
}

// (world normal, depth) planes and selected views of the previous pass (APD.cpp:552-581), resampled to this level.
void APD::LoadPriorState()
{
    plane_hypotheses_host.assign((size_t)width * height, float4{0, 0, 0, 0});
    selected_views_host.create(height, width, MAT_32SC1);
    has_prior = params_host.state != FIRST_INIT;
    if (!has_prior) {
        return;
    }
    Mat depth, normal;
    ReadBinMat(problem.result_folder / path("depths.dmb"), depth);
    ReadBinMat(problem.result_folder / path("normals.dmb"), normal);
    if (depth.empty() || normal.empty() || depth.type != MAT_32FC1 || normal.type != MAT_32FC3) {
        std::cerr << "Missing or mistyped depths.dmb / normals.dmb of a previous pass\n";
        exit(EXIT_FAILURE);
    }
    if (depth.cols != width || depth.rows != height || normal.cols != width || normal.rows != height) {
        std::cerr << "Depth and Normal doesn't match the images' size!\n";
        RescaleMatToTargetSize<float>(depth, depth, width, height);
        RescaleMatToTargetSize<Vec3f>(normal, normal, width, height);
    }
    ParallelFor((size_t)height, [&](size_t r) {
        const float *d = depth.ptr<float>((int)r);
        const Vec3f *n = normal.ptr<Vec3f>((int)r);
        float4 *p = &plane_hypotheses_host[r * (size_t)width];
        for (int col = 0; col < width; ++col) {
            p[col] = float4{n[col][0], n[col][1], n[col][2], d[col]};
        }
    }, 0);
    ReadBinMat(problem.result_folder / path("selected_views.bin"), selected_views_host);
    if (selected_views_host.empty() || selected_views_host.type != MAT_32SC1) {
        std::cerr << "Missing or mistyped selected_views.bin of a previous pass\n";
        exit(EXIT_FAILURE);
    }
    if (selected_views_host.cols != width || selected_views_host.rows != height) {
        std::cerr << "Select view doesn't match the images' size!\n";
        RescaleMatToTargetSize<uint32_t>(selected_views_host, selected_views_host, width, height);
    }
}

void APD::CudaSpaceInitialization()
{
    apd_params p;
    apd_default_params(&p);
    p.max_iterations = params_host.max_iterations;
    p.num_images = params_host.num_images;
    p.sigma_spatial = params_host.sigma_spatial;
    p.sigma_color = params_host.sigma_color;
    p.top_k = params_host.top_k;
    p.depth_min = params_host.depth_min;
    p.depth_max = params_host.depth_max;
    p.geom_consistency = params_host.geom_consistency ? 1 : 0;
    p.strong_radius = params_host.strong_radius;
    p.strong_increment = params_host.strong_increment;
    p.weak_radius = params_host.weak_radius;
    p.weak_increment = params_host.weak_increment;
    p.use_APD = params_host.use_APD ? 1 : 0;
    p.weak_peak_radius = params_host.weak_peak_radius;
    p.rotate_time = params_host.rotate_time;
    p.ransac_threshold = params_host.ransac_threshold;
    p.geom_factor = params_host.geom_factor;
    p.state = (int)params_host.state;
    p.seed = params_host.seed;
    ApdSafeCall(apd_create(&handle, g_device, width, height, &p), "apd_create");
    std::vector<const float *> img_ptrs, depth_ptrs;
    for (auto &im : images) {
        img_ptrs.push_back(im.ptr<float>());
    }
    for (auto &d : depths) {
        depth_ptrs.push_back(d.ptr<float>());
    }
    ApdSafeCall(apd_upload_views(handle, num_images, cameras.data(), img_ptrs.data(), depths.empty() ? nullptr : depth_ptrs.data()),
                "apd_upload_views");
    if (has_prior || params_host.use_APD) {
        ApdSafeCall(apd_upload_prior(handle, has_prior ? &plane_hypotheses_host[0].x : nullptr,
                                     has_prior ? selected_views_host.ptr<uint32_t>() : nullptr,
                                     params_host.use_APD ? weak_info_host.ptr<uint8_t>() : nullptr),
                    "apd_upload_prior");
    }
}

void APD::SetDataPassHelperInCuda()
{
    // The reference fills a DataPassHelper of raw device pointers here (APD.cpp:673-699); the C ABI
    // builds its kernel argument block inside apd_upload_views / apd_upload_prior.
}

void APD::RunPatchMatch()
{
    ApdSafeCall(apd_run(handle), "apd_run");
    // APD.cu:2490-2492
    ApdSafeCall(apd_download(handle, &plane_hypotheses_host[0].x, weak_info_host.ptr<uint8_t>(), selected_views_host.ptr<uint32_t>()),
                "apd_download");
}

float4 APD::GetPlaneHypothesis(int r, int c) { return plane_hypotheses_host[(size_t)c + (size_t)r * width]; }
Mat APD::GetPixelStates() { return weak_info_host; }
Mat APD::GetSelectedViews() { return selected_views_host; }
int APD::GetWidth() { return width; }
int APD::GetHeight() { return height; }
float APD::GetDepthMin() { return params_host.depth_min; }
float APD::GetDepthMax() { return params_host.depth_max; }
