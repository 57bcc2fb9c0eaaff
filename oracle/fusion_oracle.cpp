// fusion_oracle.cpp -- the reference's depth-map fusion as it is written: one sequential loop over views, rows,
// columns and source views with the consumption mask (RunFusion, ETH variant, APD.cpp:826-977) and the binary PLY of
// ExportPointCloud (APD.cpp:214-254).
//
// TEST INFRASTRUCTURE ONLY: the checker of the device fusion (apd_fuse_views, apd-mvs_amd/csrc/apd_fusion.hip).  Nothing
// in the product path calls it.  Parity with the reference is unpinned like the rest of the oracle: the
// reference ships no fusion fixture and cannot be built here.  The per-pixel arithmetic (lift / project / thresholds,
// APD.cpp:776-824, :896-925) is the header the device build compiles too (csrc/apd_fusion_math.h, contract C9); what
// this file pins is the order-dependent part: raster-order consumption of source pixels.
//
// Colours: images with 3 channels are blue, green, red as cv::imread(IMREAD_COLOR) gives them (APD.cpp:859); grey images
// (1 channel) give blue = green = red.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../apd-mvs_amd/csrc/apd_fusion_math.h"

namespace {

struct Cam {  // == apd_camera (include/apd_mi355x.h), main.h:47-56
    float K[9], R[9], t[3], c[3];
    int height, width;
    float depth_min, depth_max;
};

}  // namespace

// Same flat arguments as apd_fuse_views with host pointers.  Returns the number of points written, -1 on I/O failure.
extern "C" long long orc_fuse(int num_views, const void *cameras_v, const float *const *images, int image_channels,
                              const float *const *depths,
                              const float *const *normals, const uint8_t *const *weaks, const uint8_t *const *blocks,
                              const int *rows, const int *cols,
                              const int *pair_offsets, const int *pair_indices, const char *ply_path)
{
    const Cam *cameras = static_cast<const Cam *>(cameras_v);
    std::vector<apd_fusion::View> geo(num_views);
    std::vector<std::vector<uint8_t>> masks(num_views);  // APD.cpp:881-882
    for (int i = 0; i < num_views; ++i) {
        const Cam &c = cameras[i];
        memcpy(geo[i].K, c.K, sizeof(c.K));
        memcpy(geo[i].R, c.R, sizeof(c.R));
        memcpy(geo[i].t, c.t, sizeof(c.t));
        // -R^T t in float, as Get3DPointonWorld recomputes it per call (APD.cpp:795-798)
        geo[i].centre[0] = -(c.R[0] * c.t[0] + c.R[3] * c.t[1] + c.R[6] * c.t[2]);
        geo[i].centre[1] = -(c.R[1] * c.t[0] + c.R[4] * c.t[1] + c.R[7] * c.t[2]);
        geo[i].centre[2] = -(c.R[2] * c.t[0] + c.R[5] * c.t[1] + c.R[8] * c.t[2]);
        geo[i].rows = rows[i];
        geo[i].cols = cols[i];
        masks[i].assign((size_t)rows[i] * cols[i], 0);
    }
    std::vector<uint8_t> body;
    long long count = 0;
    std::vector<int> used;  // used_list, APD.cpp:914
    for (int i = 0; i < num_views; ++i) {  // APD.cpp:893
        const int num_ngb = pair_offsets[i + 1] - pair_offsets[i];
        const int *ngb = pair_indices + pair_offsets[i];
        used.assign(num_ngb, -1);
        for (int r = 0; r < rows[i]; ++r) {
            for (int c = 0; c < cols[i]; ++c) {
                const size_t p = (size_t)r * cols[i] + c;
                if (blocks && blocks[i] && blocks[i][p] < 128) {  // use_block, :898-900
                    continue;
                }
                if (masks[i][p] == 1) {  // :905
                    continue;
                }
                const float ref_depth = depths[i][p];
                if (ref_depth <= 0.0) {  // :909
                    continue;
                }
                const float *ref_n = normals[i] + 3 * p;
                float P[3];
                apd_fusion::lift(geo[i], c, r, ref_depth, P);  // :912
                int num_consistent = 0;
                float dynamic_consistency = 0.0f;
                for (int j = 0; j < num_ngb; ++j) {  // :916-932
                    used[j] = -1;
                    const int s_view = ngb[j];
                    int sc, sr;
                    if (!apd_fusion::vote_target(geo[s_view], P, sc, sr)) {
                        continue;
                    }
                    const size_t s = (size_t)sr * cols[s_view] + sc;
                    if (masks[s_view][s] == 1) {
                        continue;
                    }
                    const float src_depth = depths[s_view][s];
                    if (src_depth <= 0.0) {
                        continue;
                    }
                    float weight;
                    if (apd_fusion::vote_check(geo[i], geo[s_view], c, r, ref_depth, ref_n, sc, sr, src_depth, normals[s_view] + 3 * s,
                                               weight)) {
                        used[j] = (int)s;
                        dynamic_consistency += weight;
                        num_consistent++;
                    }
                }
                if (!apd_fusion::accept_point(num_consistent, dynamic_consistency, (int)weaks[i][p])) {  // :933-934
                    continue;
                }
                const int nc = image_channels;
                float colour[3];
                for (int k = 0; k < 3; ++k) {
                    colour[k] = images[i][p * nc + (nc == 3 ? k : 0)];
                }
                for (int j = 0; j < num_ngb; ++j) {  // :939-950
                    if (used[j] < 0) {
                        continue;
                    }
                    masks[ngb[j]][used[j]] = 1;
                    for (int k = 0; k < 3; ++k) {
                        colour[k] += images[ngb[j]][(size_t)used[j] * nc + (nc == 3 ? k : 0)];
                    }
                }
                uint8_t rec[15];
                memcpy(rec, P, 12);
                for (int k = 0; k < 3; ++k) {
                    rec[12 + k] = static_cast<uint8_t>(colour[k] / (num_consistent + 1));
                }
                body.insert(body.end(), rec, rec + 15);
                ++count;
            }
        }
    }
    FILE *f = fopen(ply_path, "wb");  // ExportPointCloud, APD.cpp:214-254
    if (!f) {
        return -1;
    }
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
               "property uchar diffuse_blue\nproperty uchar diffuse_green\nproperty uchar diffuse_red\nend_header\n", (int)count);
    const bool ok = body.empty() || fwrite(body.data(), 1, body.size(), f) == body.size();
    return (fclose(f) == 0 && ok) ? count : -1;
}
