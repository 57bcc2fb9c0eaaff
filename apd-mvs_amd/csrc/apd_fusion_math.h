// apd_fusion_math.h -- the per-pixel arithmetic of depth-map fusion (RunFusion, APD.cpp:826-977), written once and
// compiled twice: by g++ into the host fusion (host/fusion.cpp) and by hipcc into the device fusion (apd_fusion.hip).
// Both builds use -ffp-contract=off and no fast-math, every operation below is an IEEE operation in a fixed order, and the
// two libm functions the reference calls (acos, exp; sqrt is correctly rounded everywhere) are fixed polynomial kernels,
// so host and device give the same bits (arithmetic contract C9).
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define APD_HD __host__ __device__ inline
#else
#define APD_HD inline
#endif

namespace apd_fusion {

struct View {
    float K[9], R[9], t[3];
    float centre[3];  // -R^T t in float, as Get3DPointonWorld recomputes it per call (APD.cpp:795-798)
    int rows, cols;
};

APD_HD uint32_t f32_bits(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    return u;
}

// exp for float arguments (the reference's `exp(-tmp_index)`, APD.cpp:922): the polynomial of contract C5
APD_HD float exp_c9(float x)
{
    if (!(x > -87.0f)) {
        return (x != x) ? x : 0.0f;
    }
    if (x > 88.0f) {
        return INFINITY;
    }
    const float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, r * r, r) + 1.0f;
    uint32_t e = (uint32_t)((int)n + 127) << 23;
    float scale;
    memcpy(&scale, &e, 4);
    return p * scale;
}

// acos for float arguments (GetAngle, APD.cpp:817-824): the classic rational approximation on |x| < 0.5 and the
// sqrt((1 -+ x) / 2) reductions elsewhere, evaluated in binary32 with plain multiplies and adds; NaN for |x| > 1
APD_HD float acos_c9(float x)
{
    const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const uint32_t hx = f32_bits(x), ix = hx & 0x7fffffffu;
    if (ix == 0x3f800000u) {
        return (hx >> 31) ? pi + 2.0f * pio2_lo : 0.0f;
    }
    if (ix > 0x3f800000u) {
        return NAN;
    }
    if (ix < 0x3f000000u) {  // |x| < 0.5
        if (ix <= 0x32800000u) {
            return pio2_hi + pio2_lo;
        }
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx >> 31) {  // x < -0.5
        const float z = (1.0f + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (1.0f - x) * 0.5f;  // x > 0.5
    const float s = sqrtf(z);
    uint32_t idf = f32_bits(s) & 0xfffff000u;
    float df;
    memcpy(&df, &idf, 4);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// pixel + depth -> world point (Get3DPointonWorld, APD.cpp:776-803)
APD_HD void lift(const View &v, int x, int y, float depth, float P[3])
{
    const float X = depth * (x - v.K[2]) / v.K[0];
    const float Y = depth * (y - v.K[5]) / v.K[4];
    const float Z = depth;
    const float wx = v.R[0] * X + v.R[3] * Y + v.R[6] * Z;
    const float wy = v.R[1] * X + v.R[4] * Y + v.R[7] * Z;
    const float wz = v.R[2] * X + v.R[5] * Y + v.R[8] * Z;
    P[0] = wx + v.centre[0];
    P[1] = wy + v.centre[1];
    P[2] = wz + v.centre[2];
}

// world point -> pixel coordinates and depth in a view (ProjectCamera, APD.cpp:805-815)
APD_HD void drop(const View &v, const float P[3], float &u, float &w, float &depth)
{
    const float cx = v.R[0] * P[0] + v.R[1] * P[1] + v.R[2] * P[2] + v.t[0];
    const float cy = v.R[3] * P[0] + v.R[4] * P[1] + v.R[5] * P[2] + v.t[1];
    const float cz = v.R[6] * P[0] + v.R[7] * P[1] + v.R[8] * P[2] + v.t[2];
    depth = v.K[6] * cx + v.K[7] * cy + v.K[8] * cz;
    u = (v.K[0] * cx + v.K[1] * cy + v.K[2] * cz) / depth;
    w = (v.K[3] * cx + v.K[4] * cy + v.K[5] * cz) / depth;
}

// int(v + 0.5f) of APD.cpp:897-898 where the conversion is defined; anything else (NaN, |v| >= 2^31) lands outside
// every image on the reference's x86 build (cvttss2si gives INT_MIN) and is reported as such here
APD_HD bool nearest_pixel(float v, int &out)
{
    const float s = v + 0.5f;
    if (!(s > -2147483648.0f && s < 2147483648.0f)) {
        out = -1;
        return false;
    }
    out = (int)s;
    return true;
}

// Forward projection of world point P into view `src`: the source pixel that may vote for it (APD.cpp:896-899).
APD_HD bool vote_target(const View &src, const float P[3], int &sc, int &sr)
{
    float u, w, d;
    drop(src, P, u, w, d);
    if (!nearest_pixel(w, sr) || !nearest_pixel(u, sc)) {
        return false;
    }
    return sc >= 0 && sc < src.cols && sr >= 0 && sr < src.rows;
}

// Backward check of reference pixel (c, r) against source pixel (sc, sr) with depth src_depth and normal src_n:
// thresholds 2 px, 1 % depth, 10 degrees (APD.cpp:905-925).  `weight` is exp(-score), the term added to the consistency.
APD_HD bool vote_check(const View &ref, const View &src, int c, int r, float ref_depth, const float ref_n[3], int sc, int sr,
                       float src_depth, const float src_n[3], float &weight)
{
    float Q[3];
    lift(src, sc, sr, src_depth, Q);
    float bu, bw, back_depth;
    drop(ref, Q, bu, bw, back_depth);
    const double ex = (double)(c - bu), ey = (double)(r - bw);  // float differences, double pow(., 2) and sqrt
    const float reproj_error = (float)sqrt(ex * ex + ey * ey);
    const float relative_depth_diff = fabsf(back_depth - ref_depth) / ref_depth;
    const float dot = ref_n[0] * src_n[0] + ref_n[1] * src_n[1] + ref_n[2] * src_n[2];
    float angle = acos_c9(dot);
    if (angle != angle) {  // acos of a dot product just above 1 is NaN and counts as 0 (APD.cpp:817-824)
        angle = 0.0f;
    }
    if (!(reproj_error < 2.0f && relative_depth_diff < 0.01f && angle < 0.174533f)) {
        return false;
    }
    const float score = reproj_error + 200 * relative_depth_diff + angle * 10;
    weight = exp_c9(-score);
    return true;
}

// WEAK pixels need stronger agreement (APD.cpp:937-938); weak_state 0 == WEAK (main.h:69-73)
APD_HD bool accept_point(int agreeing, float consistency, int weak_state)
{
    const float factor = (weak_state == 0 ? 0.45f : 0.3f);
    return agreeing >= 1 && (consistency > factor * agreeing);
}

}  // namespace apd_fusion
