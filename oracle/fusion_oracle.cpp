// fusion_oracle.cpp -- the reference's depth-map fusion as it is written: one sequential loop over views, rows, columns
// and source views with the consumption mask (RunFusion, ETH variant, APD.cpp:826-977), its three helpers
// Get3DPointonWorld / ProjectCamera / GetAngle (APD.cpp:776-824) and the binary PLY of ExportPointCloud (APD.cpp:214-254).
//
// TEST INFRASTRUCTURE ONLY: the checker of the device fusion (apd_fuse_views, apd-mvs_amd/csrc/apd_fusion.hip).  Nothing
// in the product path calls it, and it includes nothing from the product: every line below is an own restatement of the
// reference lines it cites, so a transcription error on either side shows up as a different APD.ply.
//
// PARITY UNPINNED like the rest of the oracle: the reference ships no fusion fixture and cannot be built
// here.  Third-party arithmetic the reference calls and how it is restated (arithmetic contract C9):
//   acosf (glibc libm, unpinned)  -> fdlibm's e_acosf.c algorithm (rational approximation on |x| < 0.5, sqrt reductions
//                                    elsewhere), evaluated in binary32 without contraction;
//   exp   (glibc libm, unpinned)  -> the Cephes expf polynomial of contract C5 on the float argument;
//   sqrt, pow(., 2) in double     -> IEEE double multiply / add / sqrt.
// tests/test_fusion_oracle.py checks both kernels against libm (a few ulp) and, bit for bit, against the kernels the
// product compiles (apd-mvs_amd/csrc/apd_fusion_math.h through libapd_host.so): the two were written separately.
//
// Colours: images with 3 channels are blue, green, red as cv::imread(IMREAD_COLOR) gives them (APD.cpp:859); grey images
// (1 channel) give blue = green = red.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct Camera {  // main.h:47-56 (== apd_camera of include/apd_mi355x.h)
    float K[9], R[9], t[3], c[3];
    int height, width;
    float depth_min, depth_max;
};

struct float2 {
    float x, y;
};
struct float3 {
    float x, y, z;
};

uint32_t word_of(float v)
{
    uint32_t u;
    memcpy(&u, &v, sizeof(u));
    return u;
}

float float_of(uint32_t u)
{
    float v;
    memcpy(&v, &u, sizeof(v));
    return v;
}

// e_acosf.c (fdlibm / FreeBSD msun): acos(x) = pi/2 - asin(x) for |x| < 0.5, pi - 2 asin(sqrt((1+x)/2)) for x < -0.5,
// 2 asin(sqrt((1-x)/2)) with a correction term for x > 0.5; asin through the rational function P(z)/Q(z).
float orc_acosf(float x)
{
    static const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    static const float P[6] = {1.6666667163e-01f, -3.2556581497e-01f, 2.0121252537e-01f, -4.0055535734e-02f, 7.9153501429e-04f,
                               3.4793309169e-05f};
    static const float Q[5] = {1.0f, -2.4033949375e+00f, 2.0209457874e+00f, -6.8828397989e-01f, 7.7038154006e-02f};
    const uint32_t hx = word_of(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const bool negative = (hx & 0x80000000u) != 0;
    if (ix == 0x3f800000u) {  // |x| == 1
        return negative ? pi + 2.0f * pio2_lo : 0.0f;
    }
    if (ix > 0x3f800000u) {  // |x| > 1 or NaN
        return std::nanf("");
    }
    auto ratio = [&](float z) {
        float p = P[5];
        for (int k = 4; k >= 0; --k) {
            p = P[k] + z * p;
        }
        p = z * p;
        float q = Q[4];
        for (int k = 3; k >= 1; --k) {
            q = Q[k] + z * q;
        }
        q = one + z * q;
        return p / q;
    };
    if (ix < 0x3f000000u) {  // |x| < 0.5
        if (ix <= 0x32800000u) {  // |x| < 2^-26
            return pio2_hi + pio2_lo;
        }
        const float r = ratio(x * x);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (negative) {  // x < -0.5
        const float z = (one + x) * 0.5f;
        const float s = std::sqrt(z);
        const float w = ratio(z) * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;  // x > 0.5
    const float s = std::sqrt(z);
    const float df = float_of(word_of(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float w = ratio(z) * s + c;
    return 2.0f * (df + w);
}

// Cephes expf: x = n ln2 + r (ln2 split in two parts), e^r by a degree-5 polynomial in r on top of 1 + r + r^2/2.
// Written with explicit fmaf, the way contract C5 fixes the device's exp.
float orc_expf_c5(float x)
{
    if (!(x > -87.0f)) {
        return x != x ? x : 0.0f;
    }
    if (x > 88.0f) {
        return HUGE_VALF;
    }
    const float n = std::floor(std::fmaf(x, 1.44269504088896341f, 0.5f));
    const float r = std::fmaf(n, 2.12194440e-4f, std::fmaf(n, -0.693359375f, x));
    static const float C[6] = {1.9875691500e-4f, 1.3981999507e-3f, 8.3334519073e-3f, 4.1665795894e-2f, 1.6666665459e-1f,
                               5.0000001201e-1f};
    float p = C[0];
    for (int k = 1; k < 6; ++k) {
        p = std::fmaf(p, r, C[k]);
    }
    const float e_r = std::fmaf(p, r * r, r) + 1.0f;
    return e_r * float_of((uint32_t)((int)n + 127) << 23);
}

// APD.cpp:776-803
float3 Get3DPointonWorld(int x, int y, float depth, const Camera &camera)
{
    float3 pointX, tmpX, C;
    pointX.x = depth * (x - camera.K[2]) / camera.K[0];
    pointX.y = depth * (y - camera.K[5]) / camera.K[4];
    pointX.z = depth;
    tmpX.x = camera.R[0] * pointX.x + camera.R[3] * pointX.y + camera.R[6] * pointX.z;
    tmpX.y = camera.R[1] * pointX.x + camera.R[4] * pointX.y + camera.R[7] * pointX.z;
    tmpX.z = camera.R[2] * pointX.x + camera.R[5] * pointX.y + camera.R[8] * pointX.z;
    C.x = -(camera.R[0] * camera.t[0] + camera.R[3] * camera.t[1] + camera.R[6] * camera.t[2]);
    C.y = -(camera.R[1] * camera.t[0] + camera.R[4] * camera.t[1] + camera.R[7] * camera.t[2]);
    C.z = -(camera.R[2] * camera.t[0] + camera.R[5] * camera.t[1] + camera.R[8] * camera.t[2]);
    return float3{tmpX.x + C.x, tmpX.y + C.y, tmpX.z + C.z};
}

// APD.cpp:805-815
void ProjectCamera(const float3 &PointX, const Camera &camera, float2 &point, float &depth)
{
    float3 tmp;
    tmp.x = camera.R[0] * PointX.x + camera.R[1] * PointX.y + camera.R[2] * PointX.z + camera.t[0];
    tmp.y = camera.R[3] * PointX.x + camera.R[4] * PointX.y + camera.R[5] * PointX.z + camera.t[1];
    tmp.z = camera.R[6] * PointX.x + camera.R[7] * PointX.y + camera.R[8] * PointX.z + camera.t[2];
    depth = camera.K[6] * tmp.x + camera.K[7] * tmp.y + camera.K[8] * tmp.z;
    point.x = (camera.K[0] * tmp.x + camera.K[1] * tmp.y + camera.K[2] * tmp.z) / depth;
    point.y = (camera.K[3] * tmp.x + camera.K[4] * tmp.y + camera.K[5] * tmp.z) / depth;
}

// APD.cpp:817-824
float GetAngle(const float *v1, const float *v2)
{
    const float dot_product = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2];
    const float angle = orc_acosf(dot_product);
    return angle != angle ? 0.0f : angle;  // "the dot product was 1": NaN counts as 0
}

// int(v + 0.5f) of APD.cpp:921-922.  The conversion is undefined in C++ for NaN and |v| >= 2^31; the reference's x86 build
// gets INT_MIN from cvttss2si there, i.e. a pixel outside every image, which is what `false` reports.
bool RoundToPixel(float v, int &pixel)
{
    const float shifted = v + 0.5f;
    if (shifted > -2147483648.0f && shifted < 2147483648.0f) {
        pixel = (int)shifted;
        return true;
    }
    return false;
}

}  // namespace

// Kernels exported for the known-answer tests.
extern "C" float orc_fusion_acos(float x) { return orc_acosf(x); }
extern "C" float orc_fusion_exp(float x) { return orc_expf_c5(x); }

// Same flat arguments as apd_fuse_views with host pointers.  Returns the number of points written, -1 on I/O failure.
extern "C" long long orc_fuse(int num_views, const void *cameras_v, const float *const *images, int image_channels,
                              const float *const *depths, const float *const *normals, const uint8_t *const *weaks,
                              const uint8_t *const *blocks, const int *rows_of, const int *cols_of, const int *pair_offsets,
                              const int *pair_indices, const char *ply_path)
{
    const Camera *cameras = static_cast<const Camera *>(cameras_v);
    std::vector<std::vector<uint8_t>> masks(num_views);  // APD.cpp:881-882: one zeroed mask per view
    for (int i = 0; i < num_views; ++i) {
        masks[i].assign((size_t)rows_of[i] * cols_of[i], 0);
    }
    struct Support {
        int view, index;
    };
    std::vector<uint8_t> cloud;  // 15 bytes per point: x y z float, blue green red uchar (APD.cpp:236-247)
    long long count = 0;
    for (int ref_index = 0; ref_index < num_views; ++ref_index) {  // APD.cpp:893
        const int cols = cols_of[ref_index], rows = rows_of[ref_index];
        const int num_ngb = pair_offsets[ref_index + 1] - pair_offsets[ref_index];
        const int *src_of = pair_indices + pair_offsets[ref_index];
        for (int r = 0; r < rows; ++r) {
            for (int c = 0; c < cols; ++c) {
                const size_t ref_px = (size_t)r * cols + c;
                if (blocks && blocks[ref_index] && blocks[ref_index][ref_px] < 128) {  // :898-900
                    continue;
                }
                if (masks[ref_index][ref_px] == 1) {  // :902-904
                    continue;
                }
                const float ref_depth = depths[ref_index][ref_px];
                if (ref_depth <= 0.0) {  // :906-908
                    continue;
                }
                const float *ref_normal = normals[ref_index] + 3 * ref_px;
                const float3 PointX = Get3DPointonWorld(c, r, ref_depth, cameras[ref_index]);
                int num_consistent = 0;
                float dynamic_consistency = 0.0f;
                std::vector<Support> used_list;
                for (int j = 0; j < num_ngb; ++j) {  // :915-948
                    const int src_index = src_of[j];
                    const int src_cols = cols_of[src_index], src_rows = rows_of[src_index];
                    float2 point;
                    float proj_depth;
                    ProjectCamera(PointX, cameras[src_index], point, proj_depth);
                    int src_r, src_c;
                    if (!RoundToPixel(point.y, src_r) || !RoundToPixel(point.x, src_c)) {
                        continue;
                    }
                    if (!(src_c >= 0 && src_c < src_cols && src_r >= 0 && src_r < src_rows)) {
                        continue;
                    }
                    const size_t src_px = (size_t)src_r * src_cols + src_c;
                    if (masks[src_index][src_px] == 1) {
                        continue;
                    }
                    const float src_depth = depths[src_index][src_px];
                    if (src_depth <= 0.0) {
                        continue;
                    }
                    const float *src_normal = normals[src_index] + 3 * src_px;
                    const float3 tmp_X = Get3DPointonWorld(src_c, src_r, src_depth, cameras[src_index]);
                    float2 tmp_pt;
                    ProjectCamera(tmp_X, cameras[ref_index], tmp_pt, proj_depth);
                    // sqrt(pow(c - tmp_pt.x, 2) + pow(r - tmp_pt.y, 2)): float differences, double pow and sqrt (:931)
                    const double dx = (double)(c - tmp_pt.x), dy = (double)(r - tmp_pt.y);
                    const float reproj_error = (float)std::sqrt(dx * dx + dy * dy);
                    const float relative_depth_diff = std::fabs(proj_depth - ref_depth) / ref_depth;
                    const float angle = GetAngle(ref_normal, src_normal);
                    if (reproj_error < 2.0f && relative_depth_diff < 0.01f && angle < 0.174533f) {  // :935
                        used_list.push_back(Support{src_index, (int)src_px});
                        const float tmp_index = reproj_error + 200 * relative_depth_diff + angle * 10;
                        dynamic_consistency += orc_expf_c5(-tmp_index);
                        num_consistent++;
                    }
                }
                const float factor = (weaks[ref_index][ref_px] == 0 /* WEAK, main.h:70 */ ? 0.45f : 0.3f);  // :949
                if (!(num_consistent >= 1 && (dynamic_consistency > factor * num_consistent))) {
                    continue;
                }
                const int nc = image_channels;
                float consistent_Color[3];
                for (int k = 0; k < 3; ++k) {
                    consistent_Color[k] = images[ref_index][ref_px * nc + (nc == 3 ? k : 0)];
                }
                for (const Support &s : used_list) {  // :954-963: consume the supports, add their colours
                    masks[s.view][s.index] = 1;
                    for (int k = 0; k < 3; ++k) {
                        consistent_Color[k] += images[s.view][(size_t)s.index * nc + (nc == 3 ? k : 0)];
                    }
                }
                uint8_t record[15];
                memcpy(record, &PointX, 12);
                for (int k = 0; k < 3; ++k) {
                    consistent_Color[k] /= (num_consistent + 1);               // :964-966
                    record[12 + k] = static_cast<uint8_t>(consistent_Color[k]);  // (uchar) of the float colour, :243-245
                }
                cloud.insert(cloud.end(), record, record + 15);
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
    const bool ok = cloud.empty() || fwrite(cloud.data(), 1, cloud.size(), f) == cloud.size();
    return (fclose(f) == 0 && ok) ? count : -1;
}
