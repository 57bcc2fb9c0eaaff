// fusion.cpp -- depth-map fusion of the drop-in host: what the reference's RunFusion (ETH variant, APD.cpp:826-977) and
// ExportPointCloud (APD.cpp:214-254) produce, i.e. <dense>/APD/APD.ply.
//
// The product path fuses on the GPU (apd_fuse_views, csrc/apd_fusion.hip).  This file holds the drop-in entry point
// RunFusion (reads the maps, calls the device fusion) and the reference's sequential host loop, kept as the checker of
// the device fusion and selectable with APD_FUSION=cpu.  Fusion is order dependent by definition: views in problem order,
// pixels in raster order, and a source pixel that supported an accepted point is consumed (never a reference pixel,
// never a supporter again).  The per-pixel arithmetic (csrc/apd_fusion_math.h, compiled into both) keeps the
// reference's evaluation order and types (float geometry, double pow/sqrt for the reprojection error, float exp), so
// identical depth maps give an identical point list on either side.
//
// Outside the PatchMatch path proper.  One deviation: colours.  The reference re-reads the images in
// colour (cv::imread(IMREAD_COLOR), APD.cpp:859); the only decoder in this build returns the luma plane, so blue, green
// and red of a point all carry the grey value (identical to the reference for grey input images).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <unordered_map>

#include "APD.h"
#include "../csrc/apd_fusion_math.h"

namespace {

int g_fusion_device = 0;

struct V3 {
    float x, y, z;
};

struct FusionView {
    Camera cam;
    apd_fusion::View geo;  // the camera as the shared per-pixel arithmetic wants it (csrc/apd_fusion_math.h)
    Mat grey;      // float, 0..255
    Mat depth;     // float, <= 0: no estimate
    Mat normal;    // 3 x float, world frame
    Mat weak;      // uint8 PixelState
    Mat consumed;  // uint8, 1 = already merged into a point (the reference's `masks`)
};

void set_geometry(FusionView &v)
{
    const Camera &c = v.cam;
    memcpy(v.geo.K, c.K, sizeof(v.geo.K));
    memcpy(v.geo.R, c.R, sizeof(v.geo.R));
    memcpy(v.geo.t, c.t, sizeof(v.geo.t));
    // -R^T t in float, as Get3DPointonWorld recomputes it per call (APD.cpp:795-798)
    v.geo.centre[0] = -(c.R[0] * c.t[0] + c.R[3] * c.t[1] + c.R[6] * c.t[2]);
    v.geo.centre[1] = -(c.R[1] * c.t[0] + c.R[4] * c.t[1] + c.R[7] * c.t[2]);
    v.geo.centre[2] = -(c.R[2] * c.t[0] + c.R[5] * c.t[1] + c.R[8] * c.t[2]);
    v.geo.rows = v.depth.rows;
    v.geo.cols = v.depth.cols;
}

struct Support {
    int col = -1, row = -1;  // pixel of the source view that agrees with the reference pixel
};

// One source view's vote for reference pixel (c, r) with world point P: forward projection, nearest source pixel,
// backward reprojection; thresholds 2 px, 1 % depth, 10 degrees (APD.cpp:896-925).  `weight` is exp(-score).
bool vote(const FusionView &ref, const FusionView &src, int c, int r, float ref_depth, const Vec3f &ref_normal, const float P[3],
          Support &s, float &weight)
{
    int sc, sr;
    if (!apd_fusion::vote_target(src.geo, P, sc, sr)) {
        return false;
    }
    if (src.consumed.at<uint8_t>(sr, sc) == 1) {
        return false;
    }
    const float src_depth = src.depth.at<float>(sr, sc);
    if (src_depth <= 0.0) {
        return false;
    }
    const Vec3f &sn = src.normal.at<Vec3f>(sr, sc);
    if (!apd_fusion::vote_check(ref.geo, src.geo, c, r, ref_depth, ref_normal.v, sc, sr, src_depth, sn.v, weight)) {
        return false;
    }
    s.col = sc;
    s.row = sr;
    return true;
}

struct PlyWriter {  // APD.cpp:214-254: binary little-endian, x y z float + diffuse_blue/green/red uchar
    std::vector<uint8_t> body;
    size_t count = 0;
    void add(const V3 &p, const float bgr[3])
    {
        uint8_t rec[15];
        memcpy(rec + 0, &p.x, 4);
        memcpy(rec + 4, &p.y, 4);
        memcpy(rec + 8, &p.z, 4);
        for (int k = 0; k < 3; ++k) {
            rec[12 + k] = static_cast<uint8_t>(bgr[k]);
        }
        body.insert(body.end(), rec, rec + 15);
        ++count;
    }
    bool save(const path &p) const
    {
        FILE *f = fopen(p.string().c_str(), "wb");
        if (!f) {
            return false;
        }
        fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                   "property uchar diffuse_blue\nproperty uchar diffuse_green\nproperty uchar diffuse_red\nend_header\n", (int)count);
        const bool ok = body.empty() || fwrite(body.data(), 1, body.size(), f) == body.size();
        return fclose(f) == 0 && ok;
    }
};

size_t fuse(std::vector<FusionView> &views, const std::vector<std::vector<int>> &sources, const path &ply_path)
{
    PlyWriter ply;
    for (auto &v : views) {
        v.consumed.create(v.depth.rows, v.depth.cols, MAT_8UC1);
        set_geometry(v);
    }
    std::vector<Support> support;
    for (size_t i = 0; i < views.size(); ++i) {
        FusionView &ref = views[i];
        const std::vector<int> &ngb = sources[i];
        support.resize(ngb.size());
        for (int r = 0; r < ref.depth.rows; ++r) {
            for (int c = 0; c < ref.depth.cols; ++c) {
                if (ref.consumed.at<uint8_t>(r, c) == 1) {
                    continue;
                }
                const float ref_depth = ref.depth.at<float>(r, c);
                if (ref_depth <= 0.0) {
                    continue;
                }
                const Vec3f ref_normal = ref.normal.at<Vec3f>(r, c);
                float P[3];
                apd_fusion::lift(ref.geo, c, r, ref_depth, P);
                int agreeing = 0;
                float consistency = 0.0f;
                for (size_t j = 0; j < ngb.size(); ++j) {
                    support[j] = Support();
                    float weight = 0.0f;
                    if (vote(ref, views[ngb[j]], c, r, ref_depth, ref_normal, P, support[j], weight)) {
                        consistency += weight;
                        agreeing++;
                    }
                }
                if (!apd_fusion::accept_point(agreeing, consistency, (int)ref.weak.at<uint8_t>(r, c))) {
                    continue;
                }
                const float g = ref.grey.at<float>(r, c);
                float colour[3] = {g, g, g};
                for (size_t j = 0; j < ngb.size(); ++j) {
                    if (support[j].col == -1) {
                        continue;
                    }
                    FusionView &src = views[ngb[j]];
                    src.consumed.at<uint8_t>(support[j].row, support[j].col) = 1;
                    const float sg = src.grey.at<float>(support[j].row, support[j].col);
                    for (float &ch : colour) {
                        ch += sg;
                    }
                }
                for (float &ch : colour) {
                    ch /= (agreeing + 1);
                }
                ply.add(V3{P[0], P[1], P[2]}, colour);
            }
        }
    }
    if (!ply.save(ply_path)) {
        std::cerr << "Can't write " << ply_path << std::endl;
        return 0;
    }
    return ply.count;
}

// GPU fusion through the C ABI (host pointers); APD_FUSION=cpu selects the sequential host loop above.
size_t fuse_dispatch(std::vector<FusionView> &views, const std::vector<std::vector<int>> &sources, const path &ply_path)
{
    const char *mode = getenv("APD_FUSION");
    if (mode && strcmp(mode, "cpu") == 0) {
        return fuse(views, sources, ply_path);
    }
    const int V = (int)views.size();
    std::vector<apd_camera> cams(V);
    std::vector<const float *> imgs(V), deps(V), nors(V);
    std::vector<const uint8_t *> weaks(V);
    std::vector<int> rows(V), cols(V), offs(V + 1, 0), idx;
    for (int i = 0; i < V; ++i) {
        cams[i] = views[i].cam;
        imgs[i] = views[i].grey.ptr<float>();
        deps[i] = views[i].depth.ptr<float>();
        nors[i] = views[i].normal.ptr<float>();
        weaks[i] = views[i].weak.ptr<uint8_t>();
        rows[i] = views[i].depth.rows;
        cols[i] = views[i].depth.cols;
        idx.insert(idx.end(), sources[i].begin(), sources[i].end());
        offs[i + 1] = (int)idx.size();
    }
    if (idx.empty()) {
        idx.push_back(0);
    }
    long long n = 0;
    const int st = apd_fuse_views(g_fusion_device, V, cams.data(), imgs.data(), deps.data(), nors.data(), weaks.data(), rows.data(),
                                  cols.data(), offs.data(), idx.data(), 0, ply_path.string().c_str(), &n);
    if (st != APD_OK) {
        std::cerr << apd_fusion_last_error() << std::endl;
        exit(EXIT_FAILURE);
    }
    return (size_t)n;
}

}  // namespace

void SetFusionDevice(int device) { g_fusion_device = device; }

// Reads every view's final maps from <dense>/APD/<id>/ and fuses them into APD/APD.ply (APD.cpp:826-977).
void RunFusion(const path &dense_folder, const std::vector<Problem> &problems)
{
    std::vector<FusionView> views(problems.size());
    std::unordered_map<int, int> index_of_id;
    for (size_t i = 0; i < problems.size(); ++i) {
        const Problem &problem = problems[i];
        FusionView &v = views[i];
        std::cout << "Reading image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
        index_of_id.emplace(problem.ref_image_id, (int)i);
        if (!ReadGrayImage(dense_folder / path("images") / path(ToFormatIndex(problem.ref_image_id)), v.grey)) {
            exit(EXIT_FAILURE);
        }
        memset(&v.cam, 0, sizeof(v.cam));
        ReadCamera(dense_folder / path("cams") / path(ToFormatIndex(problem.ref_image_id) + "_cam.txt"), v.cam);
        ReadBinMat(problem.result_folder / path("depths.dmb"), v.depth);
        ReadBinMat(problem.result_folder / path("normals.dmb"), v.normal);
        ReadBinMat(problem.result_folder / path("weak.bin"), v.weak);
        if (v.depth.empty() || v.normal.empty() || v.weak.empty()) {
            std::cerr << "Missing maps of view " << problem.ref_image_id << " in " << problem.result_folder << std::endl;
            exit(EXIT_FAILURE);
        }
        if (v.depth.cols != v.grey.cols || v.depth.rows != v.grey.rows) {  // RescaleImageAndCamera, APD.cpp:729-750
            const float scale_x = v.depth.cols / static_cast<float>(v.grey.cols);
            const float scale_y = v.depth.rows / static_cast<float>(v.grey.rows);
            Mat scaled;
            ResizeLinear(v.grey, scaled, v.depth.cols, v.depth.rows);
            for (size_t k = 0; k < (size_t)scaled.rows * scaled.cols; ++k) {
                scaled.ptr<float>()[k] = std::nearbyint(scaled.ptr<float>()[k]);  // the reference resizes an 8-bit image
            }
            v.grey = scaled;
            v.cam.K[0] *= scale_x;
            v.cam.K[2] *= scale_x;
            v.cam.K[4] *= scale_y;
            v.cam.K[5] *= scale_y;
        }
        v.cam.width = v.depth.cols;
        v.cam.height = v.depth.rows;
        RescaleMatToTargetSize<uint8_t>(v.weak, v.weak, v.depth.cols, v.depth.rows);
    }
    std::vector<std::vector<int>> sources(problems.size());
    for (size_t i = 0; i < problems.size(); ++i) {
        std::cout << "Fusing image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
        for (int id : problems[i].src_image_ids) {
            sources[i].push_back(index_of_id[id]);  // operator[]: an id without a problem maps to view 0, as in the reference
        }
    }
    const path ply_path = dense_folder / path("APD") / path("APD.ply");
    const size_t n = fuse_dispatch(views, sources, ply_path);
    std::cout << "Fused " << n << " points into " << ply_path << std::endl;
}

extern "C" {

// Flat entry for the Python pipeline (maps already in memory after the all-gather): per-view pointers, all maps of view
// i are rows[i] x cols[i]; sources of view i are pair_indices[pair_offsets[i] .. pair_offsets[i+1]).  Returns the
// number of points written to `ply_path`.
long long apdhost_fuse(int num_views, const apd_camera *cameras, const float *const *images, const float *const *depths,
                       const float *const *normals, const uint8_t *const *weaks, const int *rows, const int *cols,
                       const int *pair_offsets, const int *pair_indices, const char *ply_path)
{
    std::vector<FusionView> views(num_views);
    std::vector<std::vector<int>> sources(num_views);
    for (int i = 0; i < num_views; ++i) {
        const size_t n = (size_t)rows[i] * cols[i];
        FusionView &v = views[i];
        v.cam = cameras[i];
        v.cam.width = cols[i];
        v.cam.height = rows[i];
        v.grey.create(rows[i], cols[i], MAT_32FC1);
        v.depth.create(rows[i], cols[i], MAT_32FC1);
        v.normal.create(rows[i], cols[i], MAT_32FC3);
        v.weak.create(rows[i], cols[i], MAT_8UC1);
        memcpy(v.grey.data(), images[i], n * 4);
        memcpy(v.depth.data(), depths[i], n * 4);
        memcpy(v.normal.data(), normals[i], n * 12);
        memcpy(v.weak.data(), weaks[i], n);
        sources[i].assign(pair_indices + pair_offsets[i], pair_indices + pair_offsets[i + 1]);
    }
    return (long long)fuse_dispatch(views, sources, path(ply_path));
}

}  // extern "C"
