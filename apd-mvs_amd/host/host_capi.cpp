// host_capi.cpp -- flat C entry points over the host-side file-format helpers, so the Python tests
// can exercise ReadBinMat / WriteBinMat / ReadCamera / image decode / resampling without a GPU.
#include <cstring>

#include "APD.h"
#include "schedule.h"
#include "../csrc/apd_fusion_math.h"

bool DecodeJpegGray(const uint8_t *data, size_t size, std::vector<uint8_t> &gray, int &width, int &height);

extern "C" {

// returns 0 on success; *type gets the OpenCV type code; data copied into `out` (cap bytes)
int apdhost_read_bin_mat(const char *p, int *rows, int *cols, int *type, void *out, size_t cap)
{
    Mat m;
    if (!ReadBinMat(path(p), m)) {
        return -1;
    }
    *rows = m.rows;
    *cols = m.cols;
    *type = m.type;
    const size_t n = m.step() * (size_t)m.rows;
    if (out && n <= cap) {
        memcpy(out, m.data(), n);
    }
    return 0;
}

int apdhost_write_bin_mat(const char *p, int rows, int cols, int type, const void *data)
{
    Mat m(rows, cols, type);
    memcpy(m.data(), data, m.step() * (size_t)rows);
    return WriteBinMat(path(p), m) ? 0 : -1;
}

int apdhost_read_camera(const char *p, apd_camera *cam)
{
    memset(cam, 0, sizeof(*cam));
    return ReadCamera(path(p), *cam) ? 0 : -1;
}

int apdhost_read_gray_image(const char *stem, int *rows, int *cols, float *out, size_t cap_floats)
{
    Mat m;
    if (!ReadGrayImage(path(stem), m)) {
        return -1;
    }
    *rows = m.rows;
    *cols = m.cols;
    if (out && (size_t)m.rows * m.cols <= cap_floats) {
        memcpy(out, m.data(), (size_t)m.rows * m.cols * sizeof(float));
    }
    return 0;
}

// colour read (blue, green, red floats per pixel); `out` holds rows*cols*3 floats
int apdhost_read_color_image(const char *stem, int *rows, int *cols, float *out, size_t cap_floats)
{
    Mat m;
    if (!ReadColorImage(path(stem), m)) {
        return -1;
    }
    *rows = m.rows;
    *cols = m.cols;
    if (out && (size_t)m.rows * m.cols * 3 <= cap_floats) {
        memcpy(out, m.data(), (size_t)m.rows * m.cols * 3 * sizeof(float));
    }
    return 0;
}

int apdhost_resize_linear(const float *src, int rows, int cols, float *dst, int new_rows, int new_cols)
{
    Mat s(rows, cols, MAT_32FC1), d;
    memcpy(s.data(), src, (size_t)rows * cols * sizeof(float));
    ResizeLinear(s, d, new_cols, new_rows);
    memcpy(dst, d.data(), (size_t)new_rows * new_cols * sizeof(float));
    return 0;
}

int apdhost_rescale_nearest_f32(const float *src, int rows, int cols, float *dst, int new_rows, int new_cols)
{
    Mat s(rows, cols, MAT_32FC1), d;
    memcpy(s.data(), src, (size_t)rows * cols * sizeof(float));
    d = s;
    RescaleMatToTargetSize<float>(s, d, new_cols, new_rows);
    memcpy(dst, d.data(), (size_t)new_rows * new_cols * sizeof(float));
    return 0;
}

const char *apdhost_format_index(int index)
{
    static thread_local std::string s;
    s = ToFormatIndex(index);
    return s.c_str();
}

// acos / exp kernels of the fusion arithmetic (csrc/apd_fusion_math.h, contract C9): which = 0 acos_c9, 1 exp_c9
void apdhost_fusion_math(const float *in, int n, int which, float *out)
{
    for (int i = 0; i < n; ++i) {
        out[i] = which == 0 ? apd_fusion::acos_c9(in[i]) : apd_fusion::exp_c9(in[i]);
    }
}

// The pass table of the reference driver (main.cpp:72-88, :168-215) as host/schedule.h builds it for both C++ schedulers: the one
// implementation; apd-mvs_amd/pipeline.py reads it from here.  Row k of `rows` (9 ints) = level, iteration, scale_size, state,
// geom_consistency, use_APD, weak_peak_radius, rotate_time, and ransac_threshold as the bit pattern of the float.
int apdhost_round_num(int width, int height) { return RoundNum(width, height); }

int apdhost_schedule(int round_num, int single_level, int *rows, int cap_rows)
{
    const std::vector<Pass> plan = BuildSchedule(round_num, single_level != 0);
    if (rows) {
        for (size_t k = 0; k < plan.size() && (int)k < cap_rows; ++k) {
            const Pass &p = plan[k];
            int *r = rows + 9 * k;
            r[0] = p.level;
            r[1] = p.iteration;
            r[2] = p.scale_size;
            r[3] = (int)p.state;
            r[4] = p.geom_consistency ? 1 : 0;
            r[5] = p.use_APD ? 1 : 0;
            r[6] = p.weak_peak_radius;
            r[7] = p.rotate_time;
            memcpy(&r[8], &p.ransac_threshold, sizeof(float));
        }
    }
    return (int)plan.size();
}

// HIP device of the fusion started by apdhost_fuse / RunFusion (default 0)
void apdhost_set_fusion_device(int device) { SetFusionDevice(device); }

}  // extern "C"
