/*
 * apd_oracle.c -- CPU ORACLE (test infrastructure, PARITY UNPINNED; see apd_oracle.h).
 *
 * Restatement of /root/reference/APD.cu:3-2495 (kernels K1..K15 and the RunPatchMatch
 * schedule) in plain C.  Every function cites the reference lines it follows.
 *
 * ARITHMETIC CONTRACT (identical in the HIP product, which is built with -ffp-contract=off):
 *   C1  All float arithmetic is literal C / IEEE-754 binary32, round-to-nearest, no implicit
 *       contraction, no fast-math.  double appears exactly where the reference promotes
 *       (double literals such as 0.25, 0.8, 0.1, M_PI).
 *   C2  fmaf() is used exactly where written below and nowhere else: homography, point
 *       correspondence, bilinear sampler, NCC moments.  (nvcc's default -fmad contracts the same
 *       multiply-adds in the reference build.)
 *   C3  In the homography and the correspondence the reference's divisions are restated as
 *       multiplications by a correctly rounded reciprocal (nvcc --use_fast_math lowers x/y to
 *       x*rcp(y)); every other division is an IEEE division.
 *   C4  sqrtf is correctly rounded; rsqrtf(x) := 1.0f / sqrtf(x).
 *   C5  sin/cos/exp of the device code are the polynomial kernels orc_sinf/orc_cosf/orc_expf.
 *       Host-side constants of K3 use libm in double.
 *   C6  min/max on floats are fminf/fmaxf (IEEE minNum/maxNum, as CUDA's min/max).
 *   C7  Texture fetch: clamp-to-edge; reference-image fetches hit texel centres (exact texel);
 *       source fetches are bilinear with weights frac(x), frac(y) in float.
 *   C8  RNG: XORWOW, rocRAND 4.2.0 seeding/skip-ahead; uniform = 2^-32 + u*2^-32 in (0,1].
 */
#include "apd_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct orc_state {
    int width, height, num_images;
    orc_params params;
    orc_camera cams[ORC_MAX_IMAGES];
    float *images[ORC_MAX_IMAGES];
    float *depths[ORC_MAX_IMAGES];
    int has_depths;
    float *planes;     /* 4 per pixel */
    float *fit_planes; /* 4 per pixel */
    float *costs;
    uint32_t *rng; /* 6 per pixel: x[0..4], d */
    uint32_t *selected_views;
    uint8_t *view_weight; /* 32 per pixel */
    uint8_t *weak_info;
    uint8_t *weak_reliable;
    int16_t *nearest_strong; /* 2 per pixel */
    int32_t *neighbours_map;
    int16_t *neighbours; /* 2*9 per weak pixel */
    int weak_count;
    /* Region of interest of the kernel loops (orc_set_roi): [rx0, rx1) x [ry0, ry1), the whole image by default.
     * The arrays always have full size; a kernel only visits (and only writes) the pixels of the region.  Every kernel
     * of the path computes a pixel from state the same launch does not write (red/black colouring, APD.cu:1510-1585;
     * per-pixel kernels read their own pixel and read-only maps), so the region's results equal the full run's. */
    int rx0, ry0, rx1, ry1;
};

static int g_threads = 0;
void orc_set_threads(int n) { g_threads = n; }
int orc_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* C5: polynomial kernels                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* sin on |x| <= pi/4 (the path only passes |x| <= 0.01*pi, APD.cu:243-252, 863). */
float orc_sinf(float x)
{
    const float z = x * x;
    float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p * z, x, x);
}

/* cos on |x| <= pi/4. */
float orc_cosf(float x)
{
    const float z = x * x;
    float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}

/* exp for finite x; the path only passes x <= 0 (APD.cu:1225, 1232, 1243). */
float orc_expf(float x)
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
    /* scale by 2^n, n in [-126, 127] here */
    union {
        uint32_t u;
        float f;
    } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return p * s.f;
}

/* ------------------------------------------------------------------------------------------ */
/* C8: XORWOW with (seed, subsequence, offset) initialisation                                  */
/* ------------------------------------------------------------------------------------------ */

/* One step of Marsaglia's xorwow on the 160-bit xorshift part (no Weyl term). */
static void xs_step(uint32_t x[5])
{
    const uint32_t t = x[0] ^ (x[0] >> 2);
    x[0] = x[1];
    x[1] = x[2];
    x[2] = x[3];
    x[3] = x[4];
    x[4] = (x[4] ^ (x[4] << 4)) ^ (t ^ (t << 1));
}

/* 160x160 GF(2) matrix stored as 160 columns of 5 words: column b = image of basis vector b. */
typedef struct {
    uint32_t col[160][5];
} gf2_mat;

static void gf2_apply(const gf2_mat *m, const uint32_t v[5], uint32_t out[5])
{
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; ++w) {
        for (int b = 0; b < 32; ++b) {
            if ((v[w] >> b) & 1u) {
                const uint32_t *c = m->col[w * 32 + b];
                r[0] ^= c[0];
                r[1] ^= c[1];
                r[2] ^= c[2];
                r[3] ^= c[3];
                r[4] ^= c[4];
            }
        }
    }
    memcpy(out, r, sizeof(r));
}

static void gf2_square(const gf2_mat *m, gf2_mat *out)
{
    gf2_mat tmp;
    for (int c = 0; c < 160; ++c) {
        gf2_apply(m, m->col[c], tmp.col[c]);
    }
    *out = tmp;
}

#define ORC_SEQ_POW_BITS 40
static gf2_mat *g_step_pow = NULL; /* A^(2^k), k = 0..63  */
static gf2_mat *g_seq_pow = NULL;  /* A^(2^(67+k)), k = 0..ORC_SEQ_POW_BITS-1 */

static void xorwow_tables(void)
{
    if (g_step_pow) {
        return;
    }
#pragma omp critical(orc_xorwow_tables)
    {
        if (!g_step_pow) {
            gf2_mat *step = (gf2_mat *)malloc(sizeof(gf2_mat) * 64);
            gf2_mat *seq = (gf2_mat *)malloc(sizeof(gf2_mat) * ORC_SEQ_POW_BITS);
            for (int c = 0; c < 160; ++c) {
                uint32_t e[5] = {0, 0, 0, 0, 0};
                e[c / 32] = 1u << (c % 32);
                xs_step(e);
                memcpy(step[0].col[c], e, sizeof(e));
            }
            for (int k = 1; k < 64; ++k) {
                gf2_square(&step[k - 1], &step[k]);
            }
            gf2_mat cur = step[63];
            for (int k = 64; k <= 67; ++k) {
                gf2_square(&cur, &cur);
            }
            seq[0] = cur; /* A^(2^67) */
            for (int k = 1; k < ORC_SEQ_POW_BITS; ++k) {
                gf2_square(&seq[k - 1], &seq[k]);
            }
            g_seq_pow = seq;
            g_step_pow = step;
        }
    }
}

/* rocrand_xorwow.h:96-131 (seeding constants) + published skip-ahead semantics. */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint64_t offset, uint32_t st[6])
{
    xorwow_tables();
    uint32_t x[5] = {123456789u, 362436069u, 521288629u, 88675123u, 5783321u};
    uint32_t d = 6615241u;
    const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fu;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbu;
    const uint32_t t0 = 1228688033u * s0;
    const uint32_t t1 = 2073658381u * s1;
    x[0] += t0;
    x[1] ^= t0;
    x[2] += t1;
    x[3] ^= t1;
    x[4] += t0;
    d += t1 + t0;
    /* skip subsequence * 2^67 steps; the Weyl term is unchanged (2^67 = 0 mod 2^32) */
    for (int k = 0; k < ORC_SEQ_POW_BITS && (subsequence >> k) != 0; ++k) {
        if ((subsequence >> k) & 1u) {
            gf2_apply(&g_seq_pow[k], x, x);
        }
    }
    /* skip offset steps */
    for (int k = 0; k < 64 && (offset >> k) != 0; ++k) {
        if ((offset >> k) & 1u) {
            gf2_apply(&g_step_pow[k], x, x);
        }
    }
    d += (uint32_t)offset * 362437u;
    memcpy(st, x, sizeof(x));
    st[5] = d;
}

uint32_t orc_xorwow_next(uint32_t st[6])
{
    xs_step(st);
    st[5] += 362437u;
    return st[5] + st[4];
}

/* rocrand_uniform.h:65-68 */
float orc_xorwow_uniform(uint32_t st[6])
{
    const float v = (float)orc_xorwow_next(st);
    return 2.3283064e-10f + (v * 2.3283064e-10f);
}

/* ------------------------------------------------------------------------------------------ */
/* small helpers (APD.cu:3-157)                                                                */
/* ------------------------------------------------------------------------------------------ */

static void sort_ascending(float *d, int n) /* APD.cu:3-12 */
{
    for (int i = 1; i < n; ++i) {
        const float v = d[i];
        int j = i;
        while (j >= 1 && v < d[j - 1]) {
            d[j] = d[j - 1];
            --j;
        }
        d[j] = v;
    }
}

static void sort_points_by_weight(int16_t *pts, float *w, int n) /* APD.cu:14-27 */
{
    for (int i = 1; i < n; ++i) {
        const int16_t px = pts[2 * i], py = pts[2 * i + 1];
        const float v = w[i];
        int j = i;
        while (j >= 1 && v < w[j - 1]) {
            pts[2 * j] = pts[2 * (j - 1)];
            pts[2 * j + 1] = pts[2 * (j - 1) + 1];
            w[j] = w[j - 1];
            --j;
        }
        pts[2 * j] = px;
        pts[2 * j + 1] = py;
        w[j] = v;
    }
}

static int last_min_index(const float *c, int n) /* APD.cu:29-40: "<=" -> last minimum wins */
{
    float best = c[0];
    int idx = 0;
    for (int i = 1; i < n; ++i) {
        if (c[i] <= best) {
            best = c[i];
            idx = i;
        }
    }
    return idx;
}

static inline void bit_set(uint32_t *v, unsigned n) { *v |= (uint32_t)(1u << n); } /* :42-45 */
/* APD.cu:47-50: clears bit n AND every lower bit (kept on purpose). */
static inline void bit_unset_quirk(uint32_t *v, unsigned n) { *v &= (uint32_t)(0xFFFFFFFEu << n); }
static inline int bit_test(uint32_t v, unsigned n) { return (int)((v >> n) & 1u); } /* :52-55 */

static inline float rsqrt_c4(float x) { return 1.0f / sqrtf(x); }

static inline void normalize3(float v[3]) /* APD.cu:126-133 */
{
    const float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float inv = rsqrt_c4(n2);
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}

static inline void normalize2(float v[2]) /* APD.cu:135-141 */
{
    const float n2 = v[0] * v[0] + v[1] * v[1];
    const float inv = rsqrt_c4(n2);
    v[0] *= inv;
    v[1] *= inv;
}

static void pdf_to_cdf(float *p, int n) /* APD.cu:143-157 */
{
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        sum += p[i];
    }
    const float inv = 1.0f / sum;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        acc += p[i] * inv;
        p[i] = acc;
    }
}

static int point_in_triangle(const int16_t A[2], const int16_t B[2], const int16_t C[2], int px,
                             int py) /* APD.cu:91-112 */
{
    const float ABx = (float)(B[0] - A[0]), ABy = (float)(B[1] - A[1]);
    const float BCx = (float)(C[0] - B[0]), BCy = (float)(C[1] - B[1]);
    const float CAx = (float)(A[0] - C[0]), CAy = (float)(A[1] - C[1]);
    const float ab = sqrtf(ABx * ABx + ABy * ABy);
    const float bc = sqrtf(BCx * BCx + BCy * BCy);
    const float ca = sqrtf(CAx * CAx + CAy * CAy);
    if (ab <= 2 || bc <= 2 || ca <= 2) {
        return 0;
    }
    if (!(ab + bc > ca && bc + ca > ab && ab + ca > bc)) {
        return 0;
    }
    const float PAx = (float)(A[0] - px), PAy = (float)(A[1] - py);
    const float PBx = (float)(B[0] - px), PBy = (float)(B[1] - py);
    const float PCx = (float)(C[0] - px), PCy = (float)(C[1] - py);
    const float t1 = PAx * PBy - PAy * PBx;
    const float t2 = PBx * PCy - PBy * PCx;
    const float t3 = PCx * PAy - PCy * PAx;
    return t1 * t2 >= 0 && t1 * t3 >= 0;
}

/* ------------------------------------------------------------------------------------------ */
/* plane / depth geometry (APD.cu:159-192, 206-209, 374-392)                                   */
/* ------------------------------------------------------------------------------------------ */

static inline void point3d(const orc_camera *cam, int px, int py, float depth, float X[3]) /* :159-171 */
{
    X[0] = depth * ((float)px - cam->K[2]) / cam->K[0];
    X[1] = depth * ((float)py - cam->K[5]) / cam->K[4];
    X[2] = depth;
}

static inline void view_direction(const orc_camera *cam, int px, int py, float depth, float v[3]) /* :173-185 */
{
    float X[3];
    point3d(cam, px, py, depth, X);
    const float norm = sqrtf(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
    v[0] = X[0] / norm;
    v[1] = X[1] / norm;
    v[2] = X[2] / norm;
}

float orc_distance_to_origin(const orc_camera *cam, int px, int py, float depth, const float n[4]) /* :187-192 */
{
    float X[3];
    point3d(cam, px, py, depth, X);
    return -(n[0] * X[0] + n[1] * X[1] + n[2] * X[2]);
}

float orc_depth_from_plane(const orc_camera *cam, const float pl[4], int px, int py) /* :206-209 */
{
    return -pl[3] * cam->K[0] /
           (((float)px - cam->K[2]) * pl[0] + (cam->K[0] / cam->K[4]) * ((float)py - cam->K[5]) * pl[1] +
            cam->K[0] * pl[2]);
}

static inline void normal_cam_to_world(const orc_camera *cam, const float in[4], float out[4]) /* :374-382 */
{
    const float x = in[0], y = in[1], z = in[2], w = in[3];
    out[0] = cam->R[0] * x + cam->R[3] * y + cam->R[6] * z;
    out[1] = cam->R[1] * x + cam->R[4] * y + cam->R[7] * z;
    out[2] = cam->R[2] * x + cam->R[5] * y + cam->R[8] * z;
    out[3] = w;
}

static inline void normal_world_to_cam(const orc_camera *cam, const float in[4], float out[4]) /* :384-392 */
{
    const float x = in[0], y = in[1], z = in[2], w = in[3];
    out[0] = cam->R[0] * x + cam->R[1] * y + cam->R[2] * z;
    out[1] = cam->R[3] * x + cam->R[4] * y + cam->R[5] * z;
    out[2] = cam->R[6] * x + cam->R[7] * y + cam->R[8] * z;
    out[3] = w;
}

/* ------------------------------------------------------------------------------------------ */
/* hypothesis generation (APD.cu:211-282)                                                      */
/* ------------------------------------------------------------------------------------------ */

static void random_normal(const orc_camera *cam, int px, int py, uint32_t *rng, float depth, float n[4]) /* :211-237 */
{
    float q1 = 1.0f, q2 = 1.0f, s = 2.0f;
    while (s >= 1.0f) {
        q1 = 2.0f * orc_xorwow_uniform(rng) - 1.0f;
        q2 = 2.0f * orc_xorwow_uniform(rng) - 1.0f;
        s = q1 * q1 + q2 * q2;
    }
    const float sq = sqrtf(1.0f - s);
    n[0] = 2.0f * q1 * sq;
    n[1] = 2.0f * q2 * sq;
    n[2] = 1.0f - 2.0f * s;
    n[3] = 0.0f;
    float v[3];
    view_direction(cam, px, py, depth, v);
    const float dot = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
    if (dot > 0.0f) {
        n[0] = -n[0];
        n[1] = -n[1];
        n[2] = -n[2];
    }
    normalize3(n);
}

static void perturbed_normal(const orc_camera *cam, int px, int py, const float normal[4], uint32_t *rng,
                             float perturbation, float out[4]) /* :239-274 */
{
    float v[3];
    view_direction(cam, px, py, 1.0f, v);
    const float a1 = (orc_xorwow_uniform(rng) - 0.5f) * perturbation;
    const float a2 = (orc_xorwow_uniform(rng) - 0.5f) * perturbation;
    const float a3 = (orc_xorwow_uniform(rng) - 0.5f) * perturbation;
    const float s1 = orc_sinf(a1), s2 = orc_sinf(a2), s3 = orc_sinf(a3);
    const float c1 = orc_cosf(a1), c2 = orc_cosf(a2), c3 = orc_cosf(a3);
    float R[9];
    R[0] = c2 * c3;
    R[1] = c3 * s1 * s2 - c1 * s3;
    R[2] = s1 * s3 + c1 * c3 * s2;
    R[3] = c2 * s3;
    R[4] = c1 * c3 + s1 * s2 * s3;
    R[5] = c1 * s2 * s3 - c3 * s1;
    R[6] = -s2;
    R[7] = c2 * s1;
    R[8] = c1 * c2;
    float p[4];
    p[0] = R[0] * normal[0] + R[1] * normal[1] + R[2] * normal[2];
    p[1] = R[3] * normal[0] + R[4] * normal[1] + R[5] * normal[2];
    p[2] = R[6] * normal[0] + R[7] * normal[1] + R[8] * normal[2];
    p[3] = 0.0f; /* the reference leaves .w indeterminate here; every caller overwrites it */
    if (p[0] * v[0] + p[1] * v[1] + p[2] * v[2] >= 0.0f) {
        p[0] = normal[0];
        p[1] = normal[1];
        p[2] = normal[2];
        p[3] = normal[3];
    }
    normalize3(p);
    memcpy(out, p, sizeof(p));
}

static void random_plane(const orc_camera *cam, int px, int py, uint32_t *rng, float dmin, float dmax,
                         float pl[4]) /* :276-282 */
{
    const float depth = orc_xorwow_uniform(rng) * (dmax - dmin) + dmin;
    random_normal(cam, px, py, rng, depth, pl);
    pl[3] = orc_distance_to_origin(cam, px, py, depth, pl);
}

/* ------------------------------------------------------------------------------------------ */
/* homography + correspondence (APD.cu:303-372), contract C2/C3                                */
/* ------------------------------------------------------------------------------------------ */

static inline float dot3_fma(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

void orc_homography(const orc_camera *ref, const orc_camera *src, const float pl[4], float H[9])
{
    /* camera centres, :305-312 */
    float refC[3], srcC[3];
    for (int j = 0; j < 3; ++j) {
        refC[j] = -dot3_fma(ref->R[0 + j], ref->t[0], ref->R[3 + j], ref->t[1], ref->R[6 + j], ref->t[2]);
        srcC[j] = -dot3_fma(src->R[0 + j], src->t[0], src->R[3 + j], src->t[1], src->R[6 + j], src->t[2]);
    }
    /* relative pose, :314-331 */
    float Rr[9], Cr[3], tr[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            Rr[3 * r + c] = dot3_fma(src->R[3 * r + 0], ref->R[3 * c + 0], src->R[3 * r + 1], ref->R[3 * c + 1],
                                     src->R[3 * r + 2], ref->R[3 * c + 2]);
        }
    }
    for (int j = 0; j < 3; ++j) {
        Cr[j] = refC[j] - srcC[j];
    }
    for (int r = 0; r < 3; ++r) {
        tr[r] = dot3_fma(src->R[3 * r + 0], Cr[0], src->R[3 * r + 1], Cr[1], src->R[3 * r + 2], Cr[2]);
    }
    /* R - t n^T / d, :333-341 (C3: n/d once, then fma) */
    const float inv_w = 1.0f / pl[3];
    const float q[3] = {pl[0] * inv_w, pl[1] * inv_w, pl[2] * inv_w};
    float M[9];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            M[3 * r + c] = fmaf(-tr[r], q[c], Rr[3 * r + c]);
        }
    }
    /* times K_ref^-1 (zero skew), :343-352 */
    const float ifx = 1.0f / ref->K[0];
    const float ify = 1.0f / ref->K[4];
    float T[9];
    for (int r = 0; r < 3; ++r) {
        T[3 * r + 0] = M[3 * r + 0] * ifx;
        T[3 * r + 1] = M[3 * r + 1] * ify;
        T[3 * r + 2] = fmaf(-T[3 * r + 1], ref->K[5], fmaf(-T[3 * r + 0], ref->K[2], M[3 * r + 2]));
    }
    /* K_src times, :354-362 */
    for (int c = 0; c < 3; ++c) {
        H[0 + c] = fmaf(src->K[2], T[6 + c], src->K[0] * T[0 + c]);
        H[3 + c] = fmaf(src->K[5], T[6 + c], src->K[4] * T[3 + c]);
        H[6 + c] = src->K[8] * T[6 + c];
    }
}

/* :365-372.  x part first so that loops over y at fixed x can hoist it. */
static inline void correspond(const float H[9], float xf, float yf, float *ox, float *oy)
{
    const float X = fmaf(H[1], yf, fmaf(H[0], xf, H[2]));
    const float Y = fmaf(H[4], yf, fmaf(H[3], xf, H[5]));
    const float Z = fmaf(H[7], yf, fmaf(H[6], xf, H[8]));
    const float inv = 1.0f / Z;
    *ox = X * inv;
    *oy = Y * inv;
}

/* ------------------------------------------------------------------------------------------ */
/* sampler (contract C7)                                                                       */
/* ------------------------------------------------------------------------------------------ */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float fetch_texel(const float *img, int W, int H, int x, int y)
{
    return img[(size_t)clampi(y, 0, H - 1) * W + clampi(x, 0, W - 1)];
}

/* Study knob, NOT part of the contract and off by default (tools/sampler_sensitivity.py): the CUDA texture unit the
 * reference samples with (APD.cpp:598-602, cudaFilterModeLinear) keeps the two interpolation weights in 9-bit fixed point
 * with 8 fractional bits (CUDA C Programming Guide, "Linear Filtering").  Contract C7 uses the float weights frac(x),
 * frac(y); with the knob on they are rounded to the nearest 1/256 first, which is the closest a CPU can get to what a
 * physical run of the reference computes.  Used to measure how far such a run may end from the contract's bits. */
static int g_study_weights_q8 = 0;
void orc_set_study_weights_q8(int on) { g_study_weights_q8 = on; }
int orc_get_study_weights_q8(void) { return g_study_weights_q8; }

float orc_sample_bilinear(const float *img, int W, int H, float sx, float sy)
{
    const float fx = floorf(sx), fy = floorf(sy);
    float a = sx - fx, b = sy - fy;
    if (g_study_weights_q8) {
        a = rintf(a * 256.0f) * (1.0f / 256.0f);
        b = rintf(b * 256.0f) * (1.0f / 256.0f);
    }
    /* clamp in float first so the int conversion is defined for NaN/huge inputs (NaN -> -1) */
    const int x0 = (int)fminf(fmaxf(fx, -1.0f), (float)W);
    const int y0 = (int)fminf(fmaxf(fy, -1.0f), (float)H);
    const int xa = clampi(x0, 0, W - 1), xb = clampi(x0 + 1, 0, W - 1);
    const int ya = clampi(y0, 0, H - 1), yb = clampi(y0 + 1, 0, H - 1);
    const float t00 = img[(size_t)ya * W + xa], t10 = img[(size_t)ya * W + xb];
    const float t01 = img[(size_t)yb * W + xa], t11 = img[(size_t)yb * W + xb];
    const float top = fmaf(a, t10 - t00, t00);
    const float bot = fmaf(a, t11 - t01, t01);
    return fmaf(b, bot - top, top);
}

/* ------------------------------------------------------------------------------------------ */
/* NCC costs (APD.cu:394-614)                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* One patch: centre (cx,cy), offsets -radius..radius step increment in x (outer) and y (inner);
 * reference :461-505 and :561-610.  Returns the clamped 1-NCC cost. */
static float patch_cost(const orc_state *s, const float H[9], int src_idx, int cx, int cy, int radius, int increment)
{
    const int W = s->width, Hh = s->height;
    const float *ref = s->images[0];
    const float *src = s->images[src_idx];
    float sum_r = 0.0f, sum_rr = 0.0f, sum_s = 0.0f, sum_ss = 0.0f, sum_rs = 0.0f, wsum = 0.0f;
    for (int i = -radius; i <= radius; i += increment) {
        float row_r = 0.0f, row_rr = 0.0f, row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f, row_w = 0.0f;
        const float xf = (float)(cx + i);
        const float bx = fmaf(H[0], xf, H[2]);
        const float by = fmaf(H[3], xf, H[5]);
        const float bz = fmaf(H[6], xf, H[8]);
        for (int j = -radius; j <= radius; j += increment) {
            const float r = fetch_texel(ref, W, Hh, cx + i, cy + j);
            const float yf = (float)(cy + j);
            const float inv = 1.0f / fmaf(H[7], yf, bz);
            const float sx = fmaf(H[1], yf, bx) * inv;
            const float sy = fmaf(H[4], yf, by) * inv;
            const float v = orc_sample_bilinear(src, W, Hh, sx, sy);
            row_r += r;
            row_rr = fmaf(r, r, row_rr);
            row_s += v;
            row_ss = fmaf(v, v, row_ss);
            row_rs = fmaf(r, v, row_rs);
            row_w += 1.0f;
        }
        sum_r += row_r;
        sum_rr += row_rr;
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
        wsum += row_w;
    }
    const float inv_w = 1.0f / wsum;
    sum_r *= inv_w;
    sum_rr *= inv_w;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_r = fmaf(-sum_r, sum_r, sum_rr);
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    const float kMinVar = 1e-5f;
    if (var_r < kMinVar || var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-sum_r, sum_s, sum_rs);
    const float denom = sqrtf(var_r * var_s);
    return fmaxf(0.0f, fminf(2.0f, 1.0f - covar / denom));
}

static float ncc_old(const orc_state *s, int px, int py, int src_idx, const float pl[4]) /* :530-614 */
{
    const orc_camera *sc = &s->cams[src_idx];
    float H[9];
    orc_homography(&s->cams[0], sc, pl, H);
    float cx, cy;
    correspond(H, (float)px, (float)py, &cx, &cy);
    if (cx >= (float)sc->width || cx < 0.0f || cy >= (float)sc->height || cy < 0.0f) {
        return 2.0f;
    }
    return patch_cost(s, H, src_idx, px, py, s->params.strong_radius, s->params.strong_increment);
}

static inline const int16_t *neighbour_slot(const orc_state *s, int center, int k) /* :394-398 */
{
    return &s->neighbours[2 * ((size_t)s->neighbours_map[center] * ORC_NEIGHBOUR_NUM + k)];
}

static float ncc_new(const orc_state *s, int px, int py, int src_idx, const float pl[4]) /* :400-528 */
{
    const int W = s->width, Hh = s->height;
    const orc_camera *sc = &s->cams[src_idx];
    const int center = px + py * W;
    float H[9];
    orc_homography(&s->cams[0], sc, pl, H);
    float cx, cy;
    correspond(H, (float)px, (float)py, &cx, &cy);
    if (cx >= (float)sc->width || cx < 0.0f || cy >= (float)sc->height || cy < 0.0f) {
        return 2.0f;
    }
    if (s->weak_info[center] != ORC_WEAK) {
        return 0.0f; /* reference prints "error" and returns 0, :523-527 */
    }
    float center_cost = 0.0f, strong_cost = 0.0f;
    int strong_count = 0;
    for (int k = 0; k < ORC_NEIGHBOUR_NUM; ++k) {
        const int16_t *nb = neighbour_slot(s, center, k);
        if (nb[0] == -1 || nb[1] == -1) {
            continue;
        }
        float nx, ny;
        correspond(H, (float)nb[0], (float)nb[1], &nx, &ny);
        if (nx < 0 || ny < 0 || nx >= (float)W || ny >= (float)Hh) {
            if (k != 0) {
                const uint32_t vi = s->selected_views[nb[0] + nb[1] * W];
                if (bit_test(vi, (unsigned)(src_idx - 1))) {
                    strong_cost += 2.0f;
                    strong_count++;
                }
                continue;
            }
            return 2.0f;
        }
        const int radius = (k == 0) ? s->params.strong_radius : s->params.weak_radius;
        const int increment = (k == 0) ? s->params.strong_increment : s->params.weak_increment;
        const float c = patch_cost(s, H, src_idx, nb[0], nb[1], radius, increment);
        if (k == 0) {
            center_cost = c;
        } else {
            strong_cost += c;
            strong_count++;
        }
    }
    if (strong_count == 0) {
        return center_cost;
    }
    strong_cost /= (float)strong_count;
    strong_cost = (strong_cost > 2.0f) ? 2.0f : strong_cost; /* MIN(strong_cost, cost_max), :519 */
    return (float)(0.25 * (double)center_cost + 0.75 * (double)strong_cost); /* :520, double */
}

/* ------------------------------------------------------------------------------------------ */
/* geometric consistency (APD.cu:718-789)                                                      */
/* ------------------------------------------------------------------------------------------ */

static inline void backproject_world(float x, float y, float depth, const orc_camera *cam, float P[3]) /* :718-738 */
{
    const float X = depth * (x - cam->K[2]) / cam->K[0];
    const float Y = depth * (y - cam->K[5]) / cam->K[4];
    const float Z = depth;
    const float tx = cam->R[0] * X + cam->R[3] * Y + cam->R[6] * Z;
    const float ty = cam->R[1] * X + cam->R[4] * Y + cam->R[7] * Z;
    const float tz = cam->R[2] * X + cam->R[5] * Y + cam->R[8] * Z;
    P[0] = tx + cam->c[0];
    P[1] = ty + cam->c[1];
    P[2] = tz + cam->c[2];
}

static inline void project_camera(const float P[3], const orc_camera *cam, float *u, float *v, float *depth) /* :740-750 */
{
    const float tx = cam->R[0] * P[0] + cam->R[1] * P[1] + cam->R[2] * P[2] + cam->t[0];
    const float ty = cam->R[3] * P[0] + cam->R[4] * P[1] + cam->R[5] * P[2] + cam->t[1];
    const float tz = cam->R[6] * P[0] + cam->R[7] * P[1] + cam->R[8] * P[2] + cam->t[2];
    *depth = cam->K[6] * tx + cam->K[7] * ty + cam->K[8] * tz;
    *u = (cam->K[0] * tx + cam->K[1] * ty + cam->K[2] * tz) / *depth;
    *v = (cam->K[3] * tx + cam->K[4] * ty + cam->K[5] * tz) / *depth;
}

static float geom_cost(const orc_state *s, int px, int py, int src_idx, const float pl[4]) /* :752-789 */
{
    const orc_camera *rc = &s->cams[0];
    const orc_camera *sc = &s->cams[src_idx];
    const float max_cost = 3.0f;
    const float depth = orc_depth_from_plane(rc, pl, px, py);
    float fwd[3];
    backproject_world((float)px, (float)py, depth, rc, fwd);
    float su, sv, sd;
    project_camera(fwd, sc, &su, &sv, &sd);
    /* tex2D(depth, (int)su + 0.5, (int)sv + 0.5): truncation, then clamp (:772) */
    const int ix = (int)fminf(fmaxf(su, -1.0f), (float)s->width);
    const int iy = (int)fminf(fmaxf(sv, -1.0f), (float)s->height);
    const float src_depth = fetch_texel(s->depths[src_idx], s->width, s->height, ix, iy);
    if (src_depth == 0.0f) {
        return max_cost;
    }
    float P[3];
    backproject_world(su, sv, src_depth, sc, P);
    float bu, bv, bd;
    project_camera(P, rc, &bu, &bv, &bd);
    const float dc = (float)px - bu;
    const float dr = (float)py - bv;
    return fminf(max_cost, sqrtf(dc * dc + dr * dr));
}

float orc_ncc_old(orc_state *s, int px, int py, int src_idx, const float pl[4]) { return ncc_old(s, px, py, src_idx, pl); }
float orc_ncc_new(orc_state *s, int px, int py, int src_idx, const float pl[4]) { return ncc_new(s, px, py, src_idx, pl); }
float orc_geom_cost(orc_state *s, int px, int py, int src_idx, const float pl[4]) { return geom_cost(s, px, py, src_idx, pl); }

/* ------------------------------------------------------------------------------------------ */
/* initial costs (APD.cu:616-693)                                                              */
/* ------------------------------------------------------------------------------------------ */

static float initial_cost_and_views(orc_state *s, int px, int py) /* :616-662 */
{
    const int center = px + py * s->width;
    const float *pl = &s->planes[4 * center];
    const int nsrc = s->params.num_images - 1;
    float sorted[32], orig[32];
    int valid = 0;
    for (int i = 0; i < nsrc; ++i) {
        const float c = ncc_old(s, px, py, i + 1, pl);
        sorted[i] = c;
        orig[i] = c;
        if (c < 2.0f) {
            valid++;
        }
    }
    sort_ascending(sorted, nsrc);
    s->selected_views[center] = 0;
    const int top_k = valid < s->params.top_k ? valid : s->params.top_k;
    if (top_k <= 0) {
        return 2.0f;
    }
    float cost = 0.0f;
    for (int i = 0; i < top_k; ++i) {
        cost += sorted[i];
    }
    const float thr = sorted[top_k - 1];
    for (int i = 0; i < nsrc; ++i) {
        if (orig[i] <= thr) {
            bit_set(&s->selected_views[center], (unsigned)i);
        }
    }
    return cost / (float)top_k;
}

static float initial_cost_stored_views(orc_state *s, int px, int py) /* :664-693 */
{
    const int center = px + py * s->width;
    const float *pl = &s->planes[4 * center];
    int count = 0;
    float cost = 0.0f;
    for (int i = 1; i < s->params.num_images; ++i) {
        if (bit_test(s->selected_views[center], (unsigned)(i - 1))) {
            const float c = ncc_old(s, px, py, i, pl);
            if (c < 2.0f) {
                count++;
                cost += c;
            } else {
                bit_unset_quirk(&s->selected_views[center], (unsigned)(i - 1));
            }
        }
    }
    return count == 0 ? 2.0f : cost / (float)count;
}

/* ------------------------------------------------------------------------------------------ */
/* K1, K5 (APD.cu:791-835)                                                                     */
/* ------------------------------------------------------------------------------------------ */

static void k1_init_random_states(orc_state *s) /* :791-804: curand_init(seed, row, col) */
{
    const int W = s->width, H = s->height;
    xorwow_tables();
    (void)H;
#pragma omp parallel for schedule(dynamic, 4) num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        uint32_t st[6];
        orc_xorwow_init(s->params.seed, (uint64_t)y, 0, st);
        for (int x = 0; x < s->rx1; ++x) {
            /* offset x = x steps of the xorshift part; Weyl term advanced alike */
            if (x >= s->rx0) {
                memcpy(&s->rng[6 * ((size_t)y * W + x)], st, sizeof(st));
            }
            xs_step(st);
            st[5] += 362437u;
        }
    }
}

static void k5_random_initialization(orc_state *s) /* :806-835 */
{
    const int W = s->width;
#pragma omp parallel for schedule(dynamic, 4) num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            const int center = y * W + x;
            float *pl = &s->planes[4 * (size_t)center];
            if (s->params.state == ORC_FIRST_INIT) {
                random_plane(&s->cams[0], x, y, &s->rng[6 * (size_t)center], s->params.depth_min, s->params.depth_max, pl);
                s->costs[center] = initial_cost_and_views(s, x, y);
            } else {
                float t[4];
                normal_world_to_cam(&s->cams[0], pl, t);
                const float depth = t[3];
                t[3] = orc_distance_to_origin(&s->cams[0], x, y, depth, t);
                memcpy(pl, t, sizeof(t));
                s->costs[center] = initial_cost_stored_views(s, x, y);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* refinement (APD.cu:837-980)                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* The five hypotheses of :855-867 / :939-951. */
static void make_refinement_set(const orc_state *s, int px, int py, uint32_t *rng, const float plane[4], float depth,
                                float depths[5], float normals[5][4])
{
    const orc_camera *cam = &s->cams[0];
    const float dmin = s->params.depth_min, dmax = s->params.depth_max;
    const float depth_perturbation = 0.02f, normal_perturbation = 0.02f;
    const float depth_rand = orc_xorwow_uniform(rng) * (dmax - dmin) + dmin;
    float n_rand[4];
    random_normal(cam, px, py, rng, depth, n_rand);
    const float lo = (1 - depth_perturbation) * depth;
    const float hi = (1 + depth_perturbation) * depth;
    /* do { } while (d < min && d > max): never loops (:860-862), exactly one draw */
    const float depth_pert = orc_xorwow_uniform(rng) * (hi - lo) + lo;
    float n_pert[4];
    perturbed_normal(cam, px, py, plane, rng, (float)((double)normal_perturbation * M_PI), n_pert);
    depths[0] = depth_rand;
    depths[1] = depth;
    depths[2] = depth_rand;
    depths[3] = depth;
    depths[4] = depth_pert;
    memcpy(normals[0], plane, 4 * sizeof(float));
    memcpy(normals[1], n_rand, 4 * sizeof(float));
    memcpy(normals[2], n_rand, 4 * sizeof(float));
    memcpy(normals[3], n_pert, 4 * sizeof(float));
    memcpy(normals[4], plane, 4 * sizeof(float));
}

static void refine_strong(orc_state *s, int px, int py, float plane[4], float *depth, float *cost, uint32_t *rng,
                          const uint8_t *vw, float weight_norm) /* :837-890 */
{
    const orc_camera *cam = &s->cams[0];
    const int nsrc = s->params.num_images - 1;
    const float dmin = s->params.depth_min, dmax = s->params.depth_max;
    float depths[5], normals[5][4];
    make_refinement_set(s, px, py, rng, plane, *depth, depths, normals);
    for (int i = 0; i < 5; ++i) {
        float t[4];
        memcpy(t, normals[i], sizeof(t));
        t[3] = orc_distance_to_origin(cam, px, py, depths[i], t);
        float tc = 0.0f;
        for (int j = 0; j < nsrc; ++j) {
            const float c = ncc_old(s, px, py, j + 1, t);
            if (vw[j] > 0) {
                tc += (float)vw[j] * c;
            }
        }
        tc /= weight_norm;
        const float d = orc_depth_from_plane(cam, t, px, py);
        if (d >= dmin && d <= dmax && tc < *cost) {
            *depth = d;
            memcpy(plane, t, sizeof(t));
            *cost = tc;
        }
    }
}

/* weighted multi-view cost with NCCNew (+ geometric term), used by :916-928, :957-970 */
static float weighted_cost_new(orc_state *s, int px, int py, const float pl[4], const uint8_t *vw, float weight_norm)
{
    const int nsrc = s->params.num_images - 1;
    float cv[32];
    for (int j = 0; j < nsrc; ++j) {
        cv[j] = ncc_new(s, px, py, j + 1, pl);
    }
    float tc = 0.0f;
    for (int j = 0; j < nsrc; ++j) {
        if (vw[j] > 0) {
            if (s->params.geom_consistency) {
                tc += (float)vw[j] * (cv[j] + s->params.geom_factor * geom_cost(s, px, py, j + 1, pl));
            } else {
                tc += (float)vw[j] * cv[j];
            }
        }
    }
    return tc / weight_norm;
}

static void refine_weak(orc_state *s, int px, int py, float plane[4], float *depth, float *cost, uint32_t *rng,
                        const uint8_t *vw, float weight_norm) /* :892-980 */
{
    const orc_camera *cam = &s->cams[0];
    const float dmin = s->params.depth_min, dmax = s->params.depth_max;
    const int center = px + py * s->width;
    const float *fit = &s->fit_planes[4 * center];
    if (fit[0] == 0 && fit[1] == 0 && fit[2] == 0) {
        return; /* also skips the random refinement, :912-914 */
    }
    {
        float f[4];
        memcpy(f, fit, sizeof(f));
        const float tc = weighted_cost_new(s, px, py, f, vw, weight_norm);
        const float d = orc_depth_from_plane(cam, f, px, py);
        if (d >= dmin && d <= dmax && tc < *cost) {
            *depth = d;
            memcpy(plane, f, sizeof(f));
            *cost = tc;
        }
    }
    float depths[5], normals[5][4];
    make_refinement_set(s, px, py, rng, plane, *depth, depths, normals);
    for (int i = 0; i < 5; ++i) {
        float t[4];
        memcpy(t, normals[i], sizeof(t));
        t[3] = orc_distance_to_origin(cam, px, py, depths[i], t);
        const float tc = weighted_cost_new(s, px, py, t, vw, weight_norm);
        const float d = orc_depth_from_plane(cam, t, px, py);
        if (d >= dmin && d <= dmax && tc < *cost) {
            *depth = d;
            memcpy(plane, t, sizeof(t));
            *cost = tc;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* view selection shared by :1203-1271 and :1365-1434                                          */
/* ------------------------------------------------------------------------------------------ */

static void select_views(orc_state *s, int center, int iter, float cost_array[8][32], const float priors[32],
                         uint8_t *vw, uint32_t *sel_out, float *weight_norm_out)
{
    const int nsrc = s->params.num_images - 1;
    float probs[32];
    memset(probs, 0, sizeof(probs));
    const float thr = (float)(0.8 * (double)orc_expf((float)(iter * iter) / (-90.0f))); /* :1225 */
    for (int i = 0; i < nsrc; ++i) {
        float count = 0;
        int count_false = 0;
        float tmpw = 0;
        for (int j = 0; j < 8; ++j) {
            const float c = cost_array[j][i];
            if (c < thr) {
                tmpw += orc_expf(c * c / (-0.18f));
                count++;
            }
            if (c > 1.2f) {
                count_false++;
            }
        }
        if (count > 2 && count_false < 3) {
            probs[i] = tmpw / count;
        } else if (count_false < 3) {
            probs[i] = orc_expf(thr * thr / (-0.32f));
        }
        probs[i] = probs[i] * priors[i];
    }
    pdf_to_cdf(probs, nsrc);
    uint32_t *rng = &s->rng[6 * (size_t)center];
    for (int sample = 0; sample < 15; ++sample) {
        const float rp = orc_xorwow_uniform(rng) - FLT_EPSILON;
        for (int v = 0; v < nsrc; ++v) {
            if (probs[v] > rp) {
                vw[v] += 1;
                break;
            }
        }
    }
    uint32_t sel = 0;
    float wn = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (vw[i] > 0) {
            bit_set(&sel, (unsigned)i);
            wn += (float)vw[i];
        }
    }
    *sel_out = sel;
    *weight_norm_out = wn;
}

/* ------------------------------------------------------------------------------------------ */
/* strong propagation (APD.cu:982-1321)                                                        */
/* ------------------------------------------------------------------------------------------ */

/* arm order (:1020): 0 up_near 1 up_far 2 down_near 3 down_far 4 left_near 5 left_far 6 right_near 7 right_far */
static const int k_arm_dx[4] = {0, 0, -1, 1};
static const int k_arm_dy[4] = {-1, 1, 0, 0};

static inline int inside(const orc_state *s, int x, int y) { return x >= 0 && y >= 0 && x < s->width && y < s->height; }

/* Returns 1 and the position of the cheapest candidate of the arm, or 0 if the arm is outside. */
static int arm_candidate(const orc_state *s, int px, int py, int arm, int *pos)
{
    const int W = s->width;
    const int d = arm >> 1;
    const int dx = k_arm_dx[d], dy = k_arm_dy[d];
    const float *costs = s->costs;
    if (arm & 1) {
        /* far arm, :1021-1095: +-3, then ten more at stride 2 */
        if (!inside(s, px + 3 * dx, py + 3 * dy)) {
            return 0;
        }
        int best = (px + 3 * dx) + (py + 3 * dy) * W;
        float cmin = costs[best];
        for (int i = 1; i < 11; ++i) {
            const int qx = px + (3 + 2 * i) * dx, qy = py + (3 + 2 * i) * dy;
            if (inside(s, qx, qy)) {
                const int q = qx + qy * W;
                if (costs[q] < cmin) {
                    cmin = costs[q];
                    best = q;
                }
            }
        }
        *pos = best;
        return 1;
    }
    /* near arm, :1097-1199: +-1, then three V-shaped pairs */
    if (!inside(s, px + dx, py + dy)) {
        return 0;
    }
    const int ex = dy != 0 ? 1 : 0, ey = dx != 0 ? 1 : 0; /* perpendicular, negative side first */
    int best = (px + dx) + (py + dy) * W;
    float cmin = costs[best];
    for (int i = 0; i < 3; ++i) {
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const int qx = px + (2 + i) * dx + sgn * (1 + i) * ex;
            const int qy = py + (2 + i) * dy + sgn * (1 + i) * ey;
            if (inside(s, qx, qy)) {
                const int q = qx + qy * W;
                if (costs[q] < cmin) {
                    cmin = costs[q];
                    best = q;
                }
            }
        }
    }
    *pos = best;
    return 1;
}

static void propagate_strong(orc_state *s, int px, int py, int iter)
{
    const int W = s->width;
    const int nsrc = s->params.num_images - 1;
    const orc_camera *cam = &s->cams[0];
    const int center = py * W + px;
    float cost_array[8][32];
    memset(cost_array, 0, sizeof(cost_array));
    cost_array[0][0] = 2.0f; /* "= { 2.0f }" initialises only the first element, :1004 */
    int flag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int positions[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int arm = 0; arm < 8; ++arm) {
        int pos;
        if (arm_candidate(s, px, py, arm, &pos)) {
            flag[arm] = 1;
            positions[arm] = pos;
            for (int v = 0; v < nsrc; ++v) {
                cost_array[arm][v] = ncc_old(s, px, py, v + 1, &s->planes[4 * (size_t)pos]);
            }
        }
    }
    /* view selection, :1203-1271 */
    uint8_t *vw = &s->view_weight[(size_t)center * ORC_MAX_IMAGES];
    memset(vw, 0, ORC_MAX_IMAGES);
    float priors[32];
    memset(priors, 0, sizeof(priors));
    const int nb_pos[4] = {center - W, center + W, center - 1, center + 1};
    for (int i = 0; i < 4; ++i) {
        if (flag[2 * i]) {
            for (int j = 0; j < nsrc; ++j) {
                priors[j] += bit_test(s->selected_views[nb_pos[i]], (unsigned)j) == 1 ? 0.9f : 0.1f;
            }
        }
    }
    uint32_t sel;
    float weight_norm;
    select_views(s, center, iter, cost_array, priors, vw, &sel, &weight_norm);

    float final_costs[8];
    for (int i = 0; i < 8; ++i) {
        float f = 0.0f;
        for (int j = 0; j < nsrc; ++j) {
            if (vw[j] > 0) {
                f += (float)vw[j] * cost_array[i][j];
            }
        }
        final_costs[i] = f / weight_norm;
    }
    const int best = last_min_index(final_costs, 8);

    float *pl_center = &s->planes[4 * (size_t)center];
    float cost_now = 0.0f;
    for (int i = 0; i < nsrc; ++i) {
        cost_now += (float)vw[i] * ncc_old(s, px, py, i + 1, pl_center);
    }
    cost_now /= weight_norm;
    s->costs[center] = cost_now;
    float depth_now = orc_depth_from_plane(cam, pl_center, px, py);
    float plane_now[4];
    memcpy(plane_now, pl_center, sizeof(plane_now));

    if (flag[best]) {
        const float *cand = &s->planes[4 * (size_t)positions[best]];
        const float d = orc_depth_from_plane(cam, cand, px, py);
        if (d >= s->params.depth_min && d <= s->params.depth_max && final_costs[best] < cost_now) {
            depth_now = d;
            memcpy(plane_now, cand, sizeof(plane_now));
            cost_now = final_costs[best];
            s->selected_views[center] = sel;
        }
    }
    refine_strong(s, px, py, plane_now, &depth_now, &cost_now, &s->rng[6 * (size_t)center], vw, weight_norm);

    if (s->params.state == ORC_REFINE_INIT) {
        if ((double)cost_now < (double)s->costs[center] - 0.1) { /* :1312, double */
            s->costs[center] = cost_now;
            memcpy(pl_center, plane_now, sizeof(plane_now));
        }
    } else {
        s->costs[center] = cost_now;
        memcpy(pl_center, plane_now, sizeof(plane_now));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* weak propagation (APD.cu:1323-1508)                                                         */
/* ------------------------------------------------------------------------------------------ */

static void propagate_weak(orc_state *s, int px, int py, int iter)
{
    const int W = s->width;
    const int nsrc = s->params.num_images - 1;
    const orc_camera *cam = &s->cams[0];
    const int center = py * W + px;
    float cost_array[8][32];
    memset(cost_array, 0, sizeof(cost_array));
    cost_array[0][0] = 2.0f; /* :1345 */
    int flag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int positions[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float cand[8][4];
    memset(cand, 0, sizeof(cand));
    for (int i = 0; i < 8; ++i) {
        const int16_t *nb = neighbour_slot(s, center, i + 1);
        if (nb[0] == -1 || nb[1] == -1 || s->weak_info[nb[0] + nb[1] * W] != ORC_STRONG) {
            continue;
        }
        positions[i] = nb[0] + nb[1] * W;
        flag[i] = 1;
        memcpy(cand[i], &s->planes[4 * (size_t)positions[i]], 4 * sizeof(float));
        for (int v = 0; v < nsrc; ++v) {
            cost_array[i][v] = ncc_new(s, px, py, v + 1, cand[i]);
        }
    }
    uint8_t *vw = &s->view_weight[(size_t)center * ORC_MAX_IMAGES];
    memset(vw, 0, ORC_MAX_IMAGES);
    float priors[32];
    memset(priors, 0, sizeof(priors));
    for (int i = 0; i < 8; ++i) {
        const int16_t *nb = neighbour_slot(s, center, i + 1);
        if (nb[0] == -1 || nb[1] == -1) {
            continue;
        }
        for (int j = 0; j < nsrc; ++j) {
            priors[j] += bit_test(s->selected_views[nb[0] + nb[1] * W], (unsigned)j) == 1 ? 0.9f : 0.1f;
        }
    }
    uint32_t sel;
    float weight_norm;
    select_views(s, center, iter, cost_array, priors, vw, &sel, &weight_norm);

    float final_costs[8];
    for (int i = 0; i < 8; ++i) {
        float f = 0.0f;
        for (int j = 0; j < nsrc; ++j) {
            if (vw[j] > 0) {
                if (s->params.geom_consistency) {
                    if (flag[i]) {
                        f += (float)vw[j] * (cost_array[i][j] + s->params.geom_factor * geom_cost(s, px, py, j + 1, cand[i]));
                    } else {
                        f += (float)vw[j] * (cost_array[i][j] + s->params.geom_factor * 3.0f);
                    }
                } else {
                    f += (float)vw[j] * cost_array[i][j];
                }
            }
        }
        final_costs[i] = f / weight_norm;
    }
    const int best = last_min_index(final_costs, 8);

    float *pl_center = &s->planes[4 * (size_t)center];
    float cost_now = 0.0f;
    for (int i = 0; i < nsrc; ++i) {
        const float c = ncc_new(s, px, py, i + 1, pl_center);
        if (s->params.geom_consistency) {
            cost_now += (float)vw[i] * (c + s->params.geom_factor * geom_cost(s, px, py, i + 1, pl_center));
        } else {
            cost_now += (float)vw[i] * c;
        }
    }
    cost_now /= weight_norm;
    s->costs[center] = cost_now;
    float depth_now = orc_depth_from_plane(cam, pl_center, px, py);
    float plane_now[4];
    memcpy(plane_now, pl_center, sizeof(plane_now));

    if (flag[best]) {
        const float d = orc_depth_from_plane(cam, cand[best], px, py);
        if (d >= s->params.depth_min && d <= s->params.depth_max && final_costs[best] < cost_now) {
            depth_now = d;
            memcpy(plane_now, cand[best], sizeof(plane_now));
            cost_now = final_costs[best];
            s->selected_views[center] = sel;
        }
    }
    refine_weak(s, px, py, plane_now, &depth_now, &cost_now, &s->rng[6 * (size_t)center], vw, weight_norm);

    if (s->params.state == ORC_REFINE_INIT) {
        if ((double)cost_now < (double)s->costs[center] - 0.1) {
            s->costs[center] = cost_now;
            memcpy(pl_center, plane_now, sizeof(plane_now));
        }
    } else {
        s->costs[center] = cost_now;
        memcpy(pl_center, plane_now, sizeof(plane_now));
    }
    /* re-score with the fixed patch, :1499-1507 */
    cost_now = 0.0f;
    for (int i = 0; i < nsrc; ++i) {
        cost_now += (float)vw[i] * ncc_old(s, px, py, i + 1, pl_center);
    }
    cost_now /= weight_norm;
    s->costs[center] = cost_now;
}

/* ------------------------------------------------------------------------------------------ */
/* checkerboard launches (APD.cu:1510-1585, 1716-1748, 2400-2407)                              */
/* ------------------------------------------------------------------------------------------ */

/* Rows reachable by the HALF launch: gy < ceil((H/2)/16)*16, y = 2*gy + {0,1}. */
static inline int half_launch_rows(int H) { return ((H / 2 + 15) / 16) * 16; }

/* colour 0 = "black" ((x+y) even), 1 = "red". */
#define ORC_FOR_COLOUR(s, colour, BODY)                                                         \
    {                                                                                           \
        const int W_ = (s)->width, H_ = (s)->height, GY_ = half_launch_rows(H_);                \
        const int G0_ = (s)->ry0 / 2, G1_ = ((s)->ry1 + 1) / 2 < GY_ ? ((s)->ry1 + 1) / 2 : GY_;   \
        _Pragma("omp parallel for schedule(dynamic, 2) num_threads(orc_get_threads())")         \
        for (int gy_ = G0_; gy_ < G1_; ++gy_) {                                                 \
            for (int x = (s)->rx0; x < (s)->rx1; ++x) {                                         \
                const int y = 2 * gy_ + (((x & 1) == 0) ? (colour) : 1 - (colour));             \
                if (y >= H_ || y < (s)->ry0 || y >= (s)->ry1) {                                 \
                    continue;                                                                   \
                }                                                                               \
                BODY                                                                            \
            }                                                                                   \
        }                                                                                       \
    }

static void k67_update_strong(orc_state *s, int colour, int iter) /* :1547-1585 */
{
    ORC_FOR_COLOUR(s, colour, {
        if (s->weak_info[x + y * W_] != ORC_WEAK) {
            propagate_strong(s, x, y, iter);
        }
    })
}

static void k910_update_weak(orc_state *s, int colour, int iter) /* :1510-1545 */
{
    ORC_FOR_COLOUR(s, colour, {
        if (s->weak_info[x + y * W_] == ORC_WEAK) {
            propagate_weak(s, x, y, iter);
        }
    })
}

/* ------------------------------------------------------------------------------------------ */
/* K11..K13 (APD.cu:1587-1748)                                                                 */
/* ------------------------------------------------------------------------------------------ */

static void k11_depth_and_normal(orc_state *s) /* :1587-1602 */
{
#pragma omp parallel for num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            float *pl = &s->planes[4 * ((size_t)y * s->width + x)];
            pl[3] = orc_depth_from_plane(&s->cams[0], pl, x, y);
            float t[4];
            normal_cam_to_world(&s->cams[0], pl, t);
            memcpy(pl, t, sizeof(t));
        }
    }
}

/* the 20-tap stencil of :1642-1704 in reference order: {dx, dy, extra top margin}.  Every
 * reference condition is "tap inside the image", except the two (+-1,-2) taps which also need
 * p.y > 2 (:1691, :1695), i.e. one more row of margin than the tap itself. */
static const int8_t k_filter_taps[20][3] = {
    {0, -1, 0}, {0, -3, 0}, {0, -5, 0},  {0, 1, 0},  {0, 3, 0},   {0, 5, 0},   {-1, 0, 0},
    {-3, 0, 0}, {-5, 0, 0}, {1, 0, 0},   {3, 0, 0},  {5, 0, 0},   {2, -1, 0},  {2, 1, 0},
    {-2, -1, 0}, {-2, 1, 0}, {-1, -2, 1}, {1, -2, 1}, {-1, 2, 0}, {1, 2, 0}};

static void filter_strong(orc_state *s, int px, int py) /* :1604-1714 */
{
    const int W = s->width;
    const int center = py * W + px;
    if (s->costs[center] < 0.001f) {
        return;
    }
    float f[21];
    int n = 0;
    f[n++] = s->planes[4 * (size_t)center + 3];
    for (int t = 0; t < 20; ++t) {
        const int qx = px + k_filter_taps[t][0], qy = py + k_filter_taps[t][1];
        if (inside(s, qx, qy - k_filter_taps[t][2]) && s->weak_info[qx + qy * W] == ORC_STRONG) {
            f[n++] = s->planes[4 * (size_t)(qx + qy * W) + 3];
        }
    }
    sort_ascending(f, n);
    const int m = n / 2;
    s->planes[4 * (size_t)center + 3] = (n % 2 == 0) ? (f[m - 1] + f[m]) / 2 : f[m];
}

static void k1213_filter(orc_state *s, int colour) /* :1716-1748 */
{
    ORC_FOR_COLOUR(s, colour, {
        if (s->weak_info[x + y * W_] != ORC_WEAK) {
            filter_strong(s, x, y);
        }
    })
}

/* ------------------------------------------------------------------------------------------ */
/* K2..K4, K8 : adaptive patch deformation (APD.cu:1750-1987, 2234-2384)                       */
/* ------------------------------------------------------------------------------------------ */

static void k2_find_nearest_strong(orc_state *s) /* :2234-2270 */
{
    const int W = s->width;
#pragma omp parallel for schedule(dynamic, 2) num_threads(orc_get_threads())
    for (int py = s->ry0; py < s->ry1; ++py) {
        for (int px = s->rx0; px < s->rx1; ++px) {
            const int center = px + py * W;
            int16_t *out = &s->nearest_strong[2 * (size_t)center];
            out[0] = -1;
            out[1] = -1;
            if (s->weak_info[center] != ORC_WEAK) {
                continue;
            }
            const int radius = 100;
            float min_dist = 255.0f;
            for (int x = -radius; x <= radius; ++x) {
                for (int y = -radius; y <= radius; ++y) {
                    const int qx = px + x, qy = py + y;
                    if (!inside(s, qx, qy)) {
                        continue;
                    }
                    if (s->weak_info[qx + qy * W] == ORC_STRONG) {
                        const float dist = sqrtf((float)(x * x + y * y));
                        if (dist < min_dist) {
                            min_dist = dist;
                            out[0] = (int16_t)qx;
                            out[1] = (int16_t)qy;
                        }
                    }
                }
            }
        }
    }
}

/* (curand()%2==0 ? 1 : -1) * curand() % range, evaluated in unsigned arithmetic; sign draw
 * first, magnitude second (order fixed by this build; :1813-1814). */
static inline int jitter_shift(uint32_t *rng, int range)
{
    const uint32_t sign = (orc_xorwow_next(rng) % 2u == 0u) ? 1u : 0xFFFFFFFFu;
    const uint32_t mag = orc_xorwow_next(rng);
    return (int)((sign * mag) % (uint32_t)range);
}

static void gen_neighbours_pixel(orc_state *s, int px, int py) /* :1750-1969 */
{
    const int W = s->width, H = s->height;
    const int center = px + py * W;
    const int min_margin = 6;
    const float depth_diff = s->params.depth_max - s->params.depth_min;
    const orc_camera *cam = &s->cams[0];
    uint32_t *rng = &s->rng[6 * (size_t)center];
    int16_t *nb = &s->neighbours[2 * ((size_t)s->neighbours_map[center] * ORC_NEIGHBOUR_NUM)];
    uint8_t *reliable = &s->weak_reliable[center];
    for (int i = 0; i < ORC_NEIGHBOUR_NUM; ++i) {
        nb[2 * i] = -1;
        nb[2 * i + 1] = -1;
    }
    nb[0] = (int16_t)px;
    nb[1] = (int16_t)py;
    int16_t strong_pts[32][2];
    int dir_valid[32];
    for (int i = 0; i < 32; ++i) {
        strong_pts[i][0] = -1;
        strong_pts[i][1] = -1;
        dir_valid[i] = 0;
    }
    int dir_base = -1, found = 0;
    const int rotate_time = s->params.rotate_time;
    const float angle = 45.0f / (float)rotate_time;
    const float cos_a = (float)cos((double)angle * M_PI / (double)180.f);
    const float sin_a = (float)sin((double)angle * M_PI / (double)180.f);
    const float cone = (float)cos((double)(angle / 2.0f) * M_PI / (double)180.0f);
    int shift_range = (int)(tan((double)(angle / 2.0f) * M_PI / (double)180.0f) * 20);
    if (shift_range < 1) {
        shift_range = 1;
    }
    const float ransac_threshold = s->params.ransac_threshold;
    for (int ox = -1; ox <= 1; ++ox) {
        for (int oy = -1; oy <= 1; ++oy) {
            if (ox == 0 && oy == 0) {
                continue;
            }
            float od[2] = {(float)ox, (float)oy};
            normalize2(od);
            dir_base++;
            for (int rot = 0; rot < rotate_time; ++rot) {
                const int slot = dir_base * 4 + rot;
                for (int radius = 2; radius <= ORC_MAX_SEARCH_RADIUS;
                     radius = (radius * 2 < radius + 25) ? radius * 2 : radius + 25) {
                    const float tx = (float)px + od[0] * (float)radius;
                    const float ty = (float)py + od[1] * (float)radius;
                    if (tx < 0 || ty < 0 || tx >= (float)W || ty >= (float)H) {
                        break;
                    }
                    for (int attempt = 0; attempt < 4; ++attempt) {
                        const int sx = jitter_shift(rng, shift_range);
                        const int sy = jitter_shift(rng, shift_range);
                        float dir[2] = {od[0] * 20 + (float)sx, od[1] * 20 + (float)sy};
                        normalize2(dir);
                        int16_t q[2];
                        q[0] = (int16_t)((float)px + dir[0] * (float)radius);
                        q[1] = (int16_t)((float)py + dir[1] * (float)radius);
                        if (q[0] < min_margin || q[1] < min_margin || q[0] >= W - min_margin || q[1] >= H - min_margin) {
                            continue;
                        }
                        int qc = q[0] + q[1] * W;
                        if (s->weak_info[qc] != ORC_STRONG) {
                            q[0] = s->nearest_strong[2 * (size_t)qc];
                            q[1] = s->nearest_strong[2 * (size_t)qc + 1];
                            if (q[0] == -1 || q[1] == -1) {
                                continue;
                            }
                            qc = q[0] + q[1] * W;
                        }
                        float td[2] = {(float)(q[0] - px), (float)(q[1] - py)};
                        normalize2(td);
                        const float ca = td[0] * od[0] + td[1] * od[1];
                        if (ca > cone) {
                            strong_pts[slot][0] = q[0];
                            strong_pts[slot][1] = q[1];
                            dir_valid[slot] = 1;
                            found++;
                            break;
                        }
                    }
                    if (dir_valid[slot]) {
                        break;
                    }
                }
                {
                    float rd[2];
                    rd[0] = od[0] * cos_a - od[1] * sin_a;
                    rd[1] = od[0] * sin_a + od[1] * cos_a;
                    normalize2(rd);
                    od[0] = rd[0];
                    od[1] = rd[1];
                }
            }
        }
    }
    if (found <= 3) {
        *reliable = 0;
        return;
    }
    float best_plane[4] = {0, 0, 0, 0};
    int use_a = -1, use_b = -1, use_c = -1, has_plane = 0;
    int16_t pts[32 * 2];
    float pts3d[32][3];
    int valid = 0;
    float Xc[3];
    point3d(cam, px, py, s->planes[4 * (size_t)center + 3], Xc); /* .w still holds the DEPTH here, :1866 */
    for (int i = 0; i < 32; ++i) {
        pts[2 * i] = -1;
        pts[2 * i + 1] = -1;
        if (dir_valid[i]) {
            const int qc = strong_pts[i][0] + strong_pts[i][1] * W;
            pts[2 * valid] = strong_pts[i][0];
            pts[2 * valid + 1] = strong_pts[i][1];
            point3d(cam, strong_pts[i][0], strong_pts[i][1], s->planes[4 * (size_t)qc + 3], pts3d[valid]);
            valid++;
        }
    }
    {
        int iteration = 50;
        float min_cost = FLT_MAX;
        int max_count = 3;
        while (iteration--) {
            const int a = (int)(orc_xorwow_next(rng) % (uint32_t)valid);
            const int b = (int)(orc_xorwow_next(rng) % (uint32_t)valid);
            const int c = (int)(orc_xorwow_next(rng) % (uint32_t)valid);
            if (a == b || b == c || a == c) {
                continue;
            }
            if (!point_in_triangle(&pts[2 * a], &pts[2 * b], &pts[2 * c], px, py)) {
                continue;
            }
            const float *A = pts3d[a], *B = pts3d[b], *C = pts3d[c];
            const float ACx = A[0] - C[0], ACy = A[1] - C[1], ACz = A[2] - C[2];
            const float BCx = B[0] - C[0], BCy = B[1] - C[1], BCz = B[2] - C[2];
            float n[4];
            n[0] = ACy * BCz - BCy * ACz;
            n[1] = -(ACx * BCz - BCx * ACz);
            n[2] = ACx * BCy - BCx * ACy;
            if ((n[0] == 0 && n[1] == 0 && n[2] == 0) || n[0] != n[0] || n[1] != n[1] || n[2] != n[2]) {
                continue;
            }
            normalize3(n);
            n[3] = -(n[0] * A[0] + n[1] * A[1] + n[2] * A[2]);
            int count = 0;
            for (int k = 0; k < valid; ++k) {
                const float *P = pts3d[k];
                const float dist = fabsf(n[0] * P[0] + n[1] * P[1] + n[2] * P[2] + n[3]);
                if (dist / depth_diff < ransac_threshold) {
                    count++;
                }
            }
            if (count < 6) {
                continue;
            }
            const float cdist = fabsf(n[0] * Xc[0] + n[1] * Xc[1] + n[2] * Xc[2] + n[3]);
            if (count > max_count) {
                max_count = count;
                min_cost = cdist;
                memcpy(best_plane, n, sizeof(n));
                has_plane = 1;
                use_a = a;
                use_b = b;
                use_c = c;
            } else if (count == max_count) {
                if (cdist < min_cost) {
                    min_cost = cdist;
                    memcpy(best_plane, n, sizeof(n));
                    use_a = a;
                    use_b = b;
                    use_c = c;
                }
            }
        }
    }
    if (!has_plane) {
        *reliable = 0;
        return;
    }
    float weight[32];
    for (int i = 0; i < valid; ++i) {
        const float *P = pts3d[i];
        float dist = fabsf(best_plane[0] * P[0] + best_plane[1] * P[1] + best_plane[2] * P[2] + best_plane[3]);
        if (dist / depth_diff >= ransac_threshold) {
            pts[2 * i] = -1;
            pts[2 * i + 1] = -1;
            weight[i] = FLT_MAX;
            continue;
        }
        if (i == use_a || i == use_b || i == use_c) {
            dist -= 1;
        }
        weight[i] = dist;
    }
    sort_points_by_weight(pts, weight, valid);
    for (int i = 1; i < ORC_NEIGHBOUR_NUM; ++i) {
        nb[2 * i] = pts[2 * (i - 1)];
        nb[2 * i + 1] = pts[2 * (i - 1) + 1];
    }
    *reliable = 1;
}

static void k3_gen_neighbours(orc_state *s)
{
    const int W = s->width;
#pragma omp parallel for schedule(dynamic, 2) num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            if (s->weak_info[x + y * W] == ORC_WEAK) {
                gen_neighbours_pixel(s, x, y);
            }
        }
    }
}

static void k4_neighbour_update(orc_state *s) /* :1971-1987 */
{
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            const size_t c = (size_t)y * s->width + x;
            if (s->weak_info[c] == ORC_WEAK && s->weak_reliable[c] != 1) {
                s->weak_info[c] = ORC_UNKNOWN;
            }
        }
    }
}

static void ransac_fit_pixel(orc_state *s, int px, int py) /* :2272-2384 */
{
    const int W = s->width;
    const int center = px + py * W;
    float *fit = &s->fit_planes[4 * (size_t)center];
    const float *pl = &s->planes[4 * (size_t)center];
    if (s->weak_info[center] != ORC_WEAK) {
        memcpy(fit, pl, 4 * sizeof(float));
        return;
    }
    uint32_t *rng = &s->rng[6 * (size_t)center];
    const orc_camera *cam = &s->cams[0];
    int16_t pts[8 * 2];
    float pts3d[8][3];
    int count = 0;
    for (int i = 1; i < ORC_NEIGHBOUR_NUM; ++i) {
        const int16_t *nb = neighbour_slot(s, center, i);
        if (nb[0] == -1 || nb[1] == -1) {
            continue;
        }
        pts[2 * count] = nb[0];
        pts[2 * count + 1] = nb[1];
        const int qc = nb[0] + nb[1] * W;
        const float depth = orc_depth_from_plane(cam, &s->planes[4 * (size_t)qc], nb[0], nb[1]);
        point3d(cam, nb[0], nb[1], depth, pts3d[count]);
        count++;
    }
    if (count < 3) {
        memcpy(fit, pl, 4 * sizeof(float));
        return;
    }
    int iteration = 50;
    float min_cost = FLT_MAX;
    float best[4] = {0, 0, 0, 0};
    int has_best = 0;
    while (iteration--) {
        const int a = (int)(orc_xorwow_next(rng) % (uint32_t)count);
        const int b = (int)(orc_xorwow_next(rng) % (uint32_t)count);
        const int c = (int)(orc_xorwow_next(rng) % (uint32_t)count);
        if (a == b || b == c || a == c) {
            continue;
        }
        if (!point_in_triangle(&pts[2 * a], &pts[2 * b], &pts[2 * c], px, py)) {
            continue;
        }
        const float *A = pts3d[a], *B = pts3d[b], *C = pts3d[c];
        const float ACx = A[0] - C[0], ACy = A[1] - C[1], ACz = A[2] - C[2];
        const float BCx = B[0] - C[0], BCy = B[1] - C[1], BCz = B[2] - C[2];
        float n[4];
        n[0] = ACy * BCz - BCy * ACz;
        n[1] = -(ACx * BCz - BCx * ACz);
        n[2] = ACx * BCy - BCx * ACy;
        if ((n[0] == 0 && n[1] == 0 && n[2] == 0) || n[0] != n[0] || n[1] != n[1] || n[2] != n[2]) {
            continue;
        }
        normalize3(n);
        n[3] = -(n[0] * A[0] + n[1] * A[1] + n[2] * A[2]);
        float tc = 0.0f;
        for (int k = 0; k < count; ++k) {
            if (k == a || k == b || k == c) {
                continue;
            }
            const float *P = pts3d[k];
            tc += fabsf(n[0] * P[0] + n[1] * P[1] + n[2] * P[2] + n[3]);
        }
        if (tc < min_cost) {
            min_cost = tc;
            memcpy(best, n, sizeof(n));
            has_best = 1;
        }
        if (min_cost == 0) {
            break;
        }
    }
    if (has_best) {
        const float depth = orc_depth_from_plane(cam, pl, px, py);
        float v[3];
        view_direction(cam, px, py, depth, v);
        const float dot = best[0] * v[0] + best[1] * v[1] + best[2] * v[2];
        if (dot > 0) {
            best[0] = -best[0];
            best[1] = -best[1];
            best[2] = -best[2];
            best[3] = -best[3];
        }
        memcpy(fit, best, sizeof(best));
    } else {
        fit[0] = fit[1] = fit[2] = fit[3] = 0.0f;
    }
}

static void k8_ransac_fit_plane(orc_state *s)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            ransac_fit_pixel(s, x, y);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* K14, K15 (APD.cu:1990-2232)                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* cost of the current depth and mean baseline over the selected views, :2022-2052 / :2169-2199 */
static int disparity_setup(orc_state *s, int px, int py, const float origin[4], float origin_depth, float *cost_now,
                           float *base_line, float *weight_normal)
{
    const int center = px + py * s->width;
    const orc_camera *cams = s->cams;
    const uint8_t *vw = &s->view_weight[(size_t)center * ORC_MAX_IMAGES];
    float cn = 0.0f, bl = 0, wn = 0.0f;
    int valid = 0;
    for (int src = 1; src < s->params.num_images; ++src) {
        const int v = src - 1;
        if (bit_test(s->selected_views[center], (unsigned)v)) {
            float t[4];
            memcpy(t, origin, sizeof(t));
            t[3] = orc_distance_to_origin(&cams[0], px, py, origin_depth, t);
            float tc = ncc_old(s, px, py, src, t);
            if (s->params.geom_consistency) {
                tc += s->params.geom_factor * geom_cost(s, px, py, src, t);
            }
            cn += tc * (float)vw[v];
            wn += (float)vw[v];
            const float d0 = cams[0].c[0] - cams[src].c[0];
            const float d1 = cams[0].c[1] - cams[src].c[1];
            const float d2 = cams[0].c[2] - cams[src].c[2];
            const double tv = (double)(d0 * d0 + d1 * d1 + d2 * d2);
            bl += sqrtf((float)tv);
            valid++;
        }
    }
    *cost_now = cn;
    *base_line = bl;
    *weight_normal = wn;
    return valid;
}

static void depth_to_weak_pixel(orc_state *s, int px, int py) /* :1990-2144 */
{
    const int W = s->width, H = s->height;
    const int min_margin = 6;
    const int center = px + py * W;
    if (px < min_margin || py < min_margin || px >= W - min_margin || py >= H - min_margin) {
        s->weak_info[center] = ORC_UNKNOWN;
        return;
    }
    const orc_camera *cams = s->cams;
    const uint8_t *vw = &s->view_weight[(size_t)center * ORC_MAX_IMAGES];
    float origin[4];
    normal_world_to_cam(&cams[0], &s->planes[4 * (size_t)center], origin);
    const float origin_depth = origin[3];
    if (origin_depth == 0) {
        s->weak_info[center] = ORC_UNKNOWN;
        return;
    }
    float cost_now, base_line, weight_normal;
    const int valid = disparity_setup(s, px, py, origin, origin_depth, &cost_now, &base_line, &weight_normal);
    if (valid == 0) {
        s->weak_info[center] = ORC_UNKNOWN;
        return;
    }
    cost_now /= weight_normal;
    base_line /= (float)valid;
    const float disp = cams[0].K[0] * base_line / origin_depth;
    enum { RADIUS = 30, NP = 2 * RADIUS + 1 };
    float pc[NP];
    for (int pd = -RADIUS; pd <= RADIUS; ++pd) {
        const float p_depth = cams[0].K[0] * base_line / (disp + (float)pd);
        if (p_depth < s->params.depth_min || p_depth > s->params.depth_max) {
            pc[pd + RADIUS] = 2.0f;
            continue;
        }
        float t[4];
        memcpy(t, origin, sizeof(t));
        t[3] = orc_distance_to_origin(&cams[0], px, py, p_depth, t);
        float p_cost = 0.0f;
        for (int src = 1; src < s->params.num_images; ++src) {
            const int v = src - 1;
            float tc = 0.0f;
            if (bit_test(s->selected_views[center], (unsigned)v)) {
                tc += ncc_old(s, px, py, src, t);
                if (s->params.geom_consistency) {
                    tc += s->params.geom_factor * geom_cost(s, px, py, src, t);
                }
                p_cost += tc * (float)vw[v];
            }
        }
        p_cost /= weight_normal;
        pc[pd + RADIUS] = (2.0f > p_cost) ? p_cost : 2.0f; /* MIN(2.0f, p_cost), :2082: NaN -> 2 */
    }
    int is_peak[NP];
    memset(is_peak, 0, sizeof(is_peak));
    int peak_count = 0, min_peak = 0;
    float min_cost = 2.0f;
    for (int i = 2; i < NP - 2; ++i) {
        if (pc[i - 1] > pc[i] && pc[i + 1] > pc[i]) {
            is_peak[i] = 1;
            peak_count++;
            if (pc[i] < min_cost) {
                min_peak = i;
                min_cost = pc[i];
            }
        }
    }
    if (abs(min_peak - RADIUS) > s->params.weak_peak_radius || pc[min_peak] > 0.5f) {
        s->weak_info[center] = ORC_WEAK;
        return;
    }
    if (peak_count == 1) {
        s->weak_info[center] = (pc[min_peak] <= 0.15f) ? ORC_STRONG : ORC_WEAK;
        return;
    }
    float var = 0.0f;
    for (int i = 2; i < NP - 2; ++i) {
        if (is_peak[i] && i != min_peak) {
            const float dist = pc[i] - min_cost;
            var += dist * dist;
        }
    }
    var = sqrtf(var);
    var /= (float)(peak_count - 1);
    s->weak_info[center] = (var > 0.2f) ? ORC_STRONG : ORC_WEAK;
}

static void local_refine_pixel(orc_state *s, int px, int py) /* :2146-2232 */
{
    const int center = px + py * s->width;
    const orc_camera *cams = s->cams;
    const uint8_t *vw = &s->view_weight[(size_t)center * ORC_MAX_IMAGES];
    float origin[4];
    normal_world_to_cam(&cams[0], &s->planes[4 * (size_t)center], origin);
    const float origin_depth = origin[3];
    if (origin_depth == 0) {
        return;
    }
    float cost_now, base_line, weight_normal;
    const int valid = disparity_setup(s, px, py, origin, origin_depth, &cost_now, &base_line, &weight_normal);
    if (weight_normal == 0 || valid == 0) {
        return;
    }
    cost_now /= weight_normal;
    base_line /= (float)valid;
    const float disp = cams[0].K[0] * base_line / origin_depth;
    const int radius = 5;
    float min_cost = 2.0f;
    float best_depth = origin_depth;
    for (int pd = -radius; pd <= radius; ++pd) {
        const float p_depth = cams[0].K[0] * base_line / (disp + (float)pd);
        if (p_depth < s->params.depth_min || p_depth > s->params.depth_max) {
            continue;
        }
        float t[4];
        memcpy(t, origin, sizeof(t));
        t[3] = orc_distance_to_origin(&cams[0], px, py, p_depth, t);
        float tc = 0.0f;
        for (int src = 1; src < s->params.num_images; ++src) {
            const int v = src - 1;
            if (bit_test(s->selected_views[center], (unsigned)v)) {
                tc += ncc_old(s, px, py, src, t) * (float)vw[v];
                if (s->params.geom_consistency) {
                    tc += s->params.geom_factor * geom_cost(s, px, py, src, t) * (float)vw[v];
                }
            }
        }
        tc /= weight_normal;
        if (tc < min_cost) {
            min_cost = tc;
            best_depth = p_depth;
        }
    }
    if ((double)(cost_now - min_cost) > 0.1) {
        s->planes[4 * (size_t)center + 3] = best_depth;
    }
}

static void k14_depth_to_weak(orc_state *s)
{
#pragma omp parallel for schedule(dynamic, 2) num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            depth_to_weak_pixel(s, x, y);
        }
    }
}

static void k15_local_refine(orc_state *s)
{
#pragma omp parallel for schedule(dynamic, 2) num_threads(orc_get_threads())
    for (int y = s->ry0; y < s->ry1; ++y) {
        for (int x = s->rx0; x < s->rx1; ++x) {
            local_refine_pixel(s, x, y);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* state + schedule                                                                            */
/* ------------------------------------------------------------------------------------------ */

orc_state *orc_create(int width, int height, const orc_params *params, const orc_camera *cameras,
                      const float *const *images, const float *const *depths, const float *prior_planes,
                      const uint32_t *prior_views, const uint8_t *prior_weak)
{
    if (params->num_images < 1 || params->num_images > ORC_MAX_IMAGES) {
        return NULL; /* APD.cpp:428-431 */
    }
    orc_state *s = (orc_state *)calloc(1, sizeof(orc_state));
    const size_t n = (size_t)width * height;
    s->width = width;
    s->height = height;
    s->rx0 = s->ry0 = 0;
    s->rx1 = width;
    s->ry1 = height;
    s->num_images = params->num_images;
    s->params = *params;
    for (int i = 0; i < s->num_images; ++i) {
        s->cams[i] = cameras[i];
        s->images[i] = (float *)malloc(n * sizeof(float));
        memcpy(s->images[i], images[i], n * sizeof(float));
        if (depths) {
            s->depths[i] = (float *)malloc(n * sizeof(float));
            memcpy(s->depths[i], depths[i], n * sizeof(float));
        }
    }
    s->has_depths = depths != NULL;
    s->planes = (float *)calloc(4 * n, sizeof(float));
    s->fit_planes = (float *)calloc(4 * n, sizeof(float)); /* APD.cpp:651 */
    s->costs = (float *)calloc(n, sizeof(float));
    s->rng = (uint32_t *)calloc(6 * n, sizeof(uint32_t));
    s->selected_views = (uint32_t *)calloc(n, sizeof(uint32_t)); /* APD.cpp:551 */
    s->view_weight = (uint8_t *)calloc(32 * n, 1); /* uninitialised in the reference; zero here and in the product */
    s->weak_info = (uint8_t *)malloc(n);
    s->weak_reliable = (uint8_t *)calloc(n, 1);
    s->nearest_strong = (int16_t *)calloc(2 * n, sizeof(int16_t));
    s->neighbours_map = (int32_t *)calloc(n, sizeof(int32_t));
    if (prior_planes) {
        memcpy(s->planes, prior_planes, 4 * n * sizeof(float));
    }
    if (prior_views) {
        memcpy(s->selected_views, prior_views, n * sizeof(uint32_t));
    }
    s->weak_count = 0;
    if (prior_weak) { /* APD.cpp:513-539 */
        memcpy(s->weak_info, prior_weak, n);
        for (size_t c = 0; c < n; ++c) {
            if (s->weak_info[c] == ORC_WEAK) {
                s->neighbours_map[c] = s->weak_count++;
            }
        }
    } else { /* APD.cpp:541-547 */
        memset(s->weak_info, ORC_STRONG, n);
    }
    s->neighbours = (int16_t *)calloc((size_t)(s->weak_count > 0 ? s->weak_count : 1) * ORC_NEIGHBOUR_NUM * 2, sizeof(int16_t));
    return s;
}

void orc_destroy(orc_state *s)
{
    if (!s) {
        return;
    }
    for (int i = 0; i < s->num_images; ++i) {
        free(s->images[i]);
        free(s->depths[i]);
    }
    free(s->planes);
    free(s->fit_planes);
    free(s->costs);
    free(s->rng);
    free(s->selected_views);
    free(s->view_weight);
    free(s->weak_info);
    free(s->weak_reliable);
    free(s->nearest_strong);
    free(s->neighbours_map);
    free(s->neighbours);
    free(s);
}

void orc_set_roi(orc_state *s, int x0, int y0, int x1, int y1)
{
    s->rx0 = x0 < 0 ? 0 : x0;
    s->ry0 = y0 < 0 ? 0 : y0;
    s->rx1 = x1 > s->width ? s->width : x1;
    s->ry1 = y1 > s->height ? s->height : y1;
}

void orc_run_kernel(orc_state *s, int kernel_id, int iter)
{
    switch (kernel_id) {
    case ORC_K1_INIT_RANDOM_STATES: k1_init_random_states(s); break;
    case ORC_K2_FIND_NEAREST_STRONG: k2_find_nearest_strong(s); break;
    case ORC_K3_GEN_NEIGHBOURS: k3_gen_neighbours(s); break;
    case ORC_K4_NEIGHBOUR_UPDATE: k4_neighbour_update(s); break;
    case ORC_K5_RANDOM_INITIALIZATION: k5_random_initialization(s); break;
    case ORC_K6_BLACK_UPDATE_STRONG: k67_update_strong(s, 0, iter); break;
    case ORC_K7_RED_UPDATE_STRONG: k67_update_strong(s, 1, iter); break;
    case ORC_K8_RANSAC_FIT_PLANE: k8_ransac_fit_plane(s); break;
    case ORC_K9_BLACK_UPDATE_WEAK: k910_update_weak(s, 0, iter); break;
    case ORC_K10_RED_UPDATE_WEAK: k910_update_weak(s, 1, iter); break;
    case ORC_K11_GET_DEPTH_NORMAL: k11_depth_and_normal(s); break;
    case ORC_K12_BLACK_FILTER: k1213_filter(s, 0); break;
    case ORC_K13_RED_FILTER: k1213_filter(s, 1); break;
    case ORC_K14_DEPTH_TO_WEAK: k14_depth_to_weak(s); break;
    case ORC_K15_LOCAL_REFINE: k15_local_refine(s); break;
    default: fprintf(stderr, "orc_run_kernel: unknown kernel %d\n", kernel_id); break;
    }
}

void orc_run_sweeps(orc_state *s, int first_iter, int iters) /* APD.cu:2443-2457 */
{
    for (int i = first_iter; i < first_iter + iters; ++i) {
        orc_run_kernel(s, ORC_K6_BLACK_UPDATE_STRONG, i);
        orc_run_kernel(s, ORC_K7_RED_UPDATE_STRONG, i);
        orc_run_kernel(s, ORC_K8_RANSAC_FIT_PLANE, i);
        orc_run_kernel(s, ORC_K9_BLACK_UPDATE_WEAK, i);
        orc_run_kernel(s, ORC_K10_RED_UPDATE_WEAK, i);
    }
}

void orc_run(orc_state *s) /* APD.cu:2386-2495 */
{
    orc_run_kernel(s, ORC_K1_INIT_RANDOM_STATES, 0);
    orc_run_kernel(s, ORC_K2_FIND_NEAREST_STRONG, 0);
    orc_run_kernel(s, ORC_K3_GEN_NEIGHBOURS, 0);
    orc_run_kernel(s, ORC_K4_NEIGHBOUR_UPDATE, 0);
    orc_run_kernel(s, ORC_K5_RANDOM_INITIALIZATION, 0);
    orc_run_sweeps(s, 0, s->params.max_iterations);
    orc_run_kernel(s, ORC_K11_GET_DEPTH_NORMAL, 0);
    orc_run_kernel(s, ORC_K12_BLACK_FILTER, 0);
    orc_run_kernel(s, ORC_K13_RED_FILTER, 0);
    orc_run_kernel(s, ORC_K14_DEPTH_TO_WEAK, 0);
    orc_run_kernel(s, ORC_K15_LOCAL_REFINE, 0);
}

float *orc_planes(orc_state *s) { return s->planes; }
float *orc_fit_planes(orc_state *s) { return s->fit_planes; }
float *orc_costs(orc_state *s) { return s->costs; }
uint32_t *orc_rng(orc_state *s) { return s->rng; }
uint32_t *orc_selected_views(orc_state *s) { return s->selected_views; }
uint8_t *orc_view_weight(orc_state *s) { return s->view_weight; }
uint8_t *orc_weak_info(orc_state *s) { return s->weak_info; }
uint8_t *orc_weak_reliable(orc_state *s) { return s->weak_reliable; }
int16_t *orc_nearest_strong(orc_state *s) { return s->nearest_strong; }
int32_t *orc_neighbours_map(orc_state *s) { return s->neighbours_map; }
int16_t *orc_neighbours(orc_state *s) { return s->neighbours; }
int orc_weak_count(orc_state *s) { return s->weak_count; }
