/*
 * apd_oracle.h -- CPU ORACLE for the APD-MVS PatchMatch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker
 * (or as the timed CPU baseline), never as the thing being measured or shipped.
 *
 * PARITY UNPINNED.  The reference (whoiszzj/APD-MVS) ships no tests, golden vectors or fixtures
 * for this path, and it cannot be built in this image (needs nvcc, OpenCV, Boost: none present,
 * and stand-in headers are not allowed).  This file is therefore a hand restatement of
 * /root/reference/APD.cu:3-2495 whose only anchors are (a) the reference source it cites line
 * by line, (b) known-answer tests of its own building blocks (tests/test_oracle_*.py) and
 * (c) rocRAND 4.2.0's host XORWOW engine for the random stream.
 *
 * Third-party pieces of the reference path that are NOT in /root/reference and how they are
 * restated here (see DESIGN.md "arithmetic contract"):
 *   - cuRAND XORWOW (CUDA toolkit >= 10.2, unpinned, README.md:23)  -> Marsaglia xorwow with
 *     rocRAND 4.2.0's seeding constants and (seed, subsequence, offset) skip-ahead semantics,
 *     implemented from the published algorithm with GF(2) matrix powers; checked against
 *     /opt/rocm/include/rocrand/rocrand_xorwow.h in tests/test_rng.py.
 *   - CUDA texture unit (tex2D, linear filter, clamp)  -> software bilinear sampler with exact
 *     float weights (frac of the coordinate), clamp-to-edge.
 *   - nvcc --use_fast_math / -fmad  -> an explicit arithmetic contract: no implicit contraction,
 *     fmaf() exactly where written, division by reciprocal-multiply in the homography /
 *     correspondence code, correctly rounded sqrt and 1/x, own polynomial sin/cos/exp.
 */
#ifndef APD_ORACLE_H_
#define APD_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_IMAGES 32      /* main.h:37 */
#define ORC_NEIGHBOUR_NUM 9    /* main.h:38 */
#define ORC_MAX_SEARCH_RADIUS 4096 /* main.h:39 */

enum { ORC_FIRST_INIT = 0, ORC_REFINE_INIT = 1, ORC_REFINE_ITER = 2 }; /* main.h:63-67 */
enum { ORC_WEAK = 0, ORC_STRONG = 1, ORC_UNKNOWN = 2 };                /* main.h:69-73 */

/* Same field order and size (112 B) as the reference Camera, main.h:47-56. */
typedef struct {
    float K[9];
    float R[9];
    float t[3];
    float c[3];
    int height;
    int width;
    float depth_min;
    float depth_max;
} orc_camera;

/* Fields of PatchMatchParams (main.h:75-94) that the device path reads, plus the added seed knob. */
typedef struct {
    int max_iterations;
    int num_images;
    int top_k;
    float depth_min;
    float depth_max;
    int geom_consistency;
    int strong_radius;
    int strong_increment;
    int weak_radius;
    int weak_increment;
    int use_APD;
    int weak_peak_radius;
    int rotate_time;
    float ransac_threshold;
    float geom_factor;
    int state;
    uint64_t seed; /* replaces clock64() in APD.cu:803 */
} orc_params;

typedef struct orc_state orc_state;

/* Kernel ids follow SURVEY.md section 2.1 (K1..K15). */
enum {
    ORC_K1_INIT_RANDOM_STATES = 1,
    ORC_K2_FIND_NEAREST_STRONG = 2,
    ORC_K3_GEN_NEIGHBOURS = 3,
    ORC_K4_NEIGHBOUR_UPDATE = 4,
    ORC_K5_RANDOM_INITIALIZATION = 5,
    ORC_K6_BLACK_UPDATE_STRONG = 6,
    ORC_K7_RED_UPDATE_STRONG = 7,
    ORC_K8_RANSAC_FIT_PLANE = 8,
    ORC_K9_BLACK_UPDATE_WEAK = 9,
    ORC_K10_RED_UPDATE_WEAK = 10,
    ORC_K11_GET_DEPTH_NORMAL = 11,
    ORC_K12_BLACK_FILTER = 12,
    ORC_K13_RED_FILTER = 13,
    ORC_K14_DEPTH_TO_WEAK = 14,
    ORC_K15_LOCAL_REFINE = 15
};

/*
 * images: num_images pointers to W*H floats (index 0 = reference view).
 * depths: NULL or num_images pointers to W*H floats (geometric passes).
 * prior_planes (4*W*H, xyz = world normal, w = depth), prior_views (W*H), prior_weak (W*H):
 *   NULL in FIRST_INIT; otherwise the state a previous pass left (APD.cpp:552-581).
 *   prior_weak == NULL means "all STRONG" (APD.cpp:541-547).
 */
orc_state *orc_create(int width, int height, const orc_params *params, const orc_camera *cameras,
                      const float *const *images, const float *const *depths,
                      const float *prior_planes, const uint32_t *prior_views,
                      const uint8_t *prior_weak);
void orc_destroy(orc_state *s);

void orc_set_threads(int n);
/* study knob, off by default and not part of the contract: bilinear weights rounded to 8 fractional bits (the CUDA texture unit) */
void orc_set_study_weights_q8(int on);
int orc_get_study_weights_q8(void);
int orc_get_threads(void);

/* Region of interest: the kernels only visit (and write) pixels of [x0, x1) x [y0, y1); arrays keep their full size.
 * Used by the full-resolution parity tests: the HIP path runs the whole image, the oracle a few windows of it, from the
 * same pre-kernel state.  Default: the whole image. */
void orc_set_roi(orc_state *s, int x0, int y0, int x1, int y1);

void orc_run_kernel(orc_state *s, int kernel_id, int iter);
/* Whole schedule of APD.cu:2386-2495. */
void orc_run(orc_state *s);
/* Only the loop body APD.cu:2443-2457 (K6,K7,K8,K9,K10), `iters` times starting at `first_iter`. */
void orc_run_sweeps(orc_state *s, int first_iter, int iters);

/* Raw views of the state (owned by the oracle). */
float *orc_planes(orc_state *s);          /* 4*W*H */
float *orc_fit_planes(orc_state *s);      /* 4*W*H */
float *orc_costs(orc_state *s);           /* W*H */
uint32_t *orc_rng(orc_state *s);          /* 6*W*H : x0..x4, d */
uint32_t *orc_selected_views(orc_state *s);
uint8_t *orc_view_weight(orc_state *s);   /* 32*W*H */
uint8_t *orc_weak_info(orc_state *s);
uint8_t *orc_weak_reliable(orc_state *s);
int16_t *orc_nearest_strong(orc_state *s); /* 2*W*H */
int32_t *orc_neighbours_map(orc_state *s);
int16_t *orc_neighbours(orc_state *s);     /* 2*9*weak_count */
int orc_weak_count(orc_state *s);

/* Building blocks exported for known-answer tests. */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint64_t offset, uint32_t out_state[6]);
uint32_t orc_xorwow_next(uint32_t state[6]);
float orc_xorwow_uniform(uint32_t state[6]);
float orc_sinf(float x);
float orc_cosf(float x);
float orc_expf(float x);
void orc_homography(const orc_camera *ref, const orc_camera *src, const float plane[4], float H[9]);
float orc_sample_bilinear(const float *img, int width, int height, float sx, float sy);
float orc_ncc_old(orc_state *s, int px, int py, int src_idx, const float plane[4]);
float orc_ncc_new(orc_state *s, int px, int py, int src_idx, const float plane[4]);
float orc_geom_cost(orc_state *s, int px, int py, int src_idx, const float plane[4]);
float orc_depth_from_plane(const orc_camera *cam, const float plane[4], int px, int py);
float orc_distance_to_origin(const orc_camera *cam, int px, int py, float depth, const float n[4]);

#ifdef __cplusplus
}
#endif
#endif
