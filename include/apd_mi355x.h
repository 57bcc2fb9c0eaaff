/*
 * apd_mi355x.h -- C ABI of the MI355X-native PatchMatch path of APD-MVS.
 *
 * The reference (whoiszzj/APD-MVS) has no C ABI / FFI layer: its boundary for this path is the C++
 * class `APD` (APD.h:67-145) driven by `ProcessProblem` (main.cpp:91-138).  This header is the
 * flat `extern "C"` equivalent of that class: one handle == one `APD` object == one
 * (reference view, pass).  Every entry point cites the reference member it replaces.  The C++
 * drop-in class (apd-mvs_amd/host/APD.h) and the Python host mirror (apd-mvs_amd/__init__.py)
 * are thin layers over exactly these symbols.
 *
 * Conventions: all functions return 0 on success and a negative apd_status otherwise (the
 * reference calls exit(); a library must not).  apd_last_error() gives the message of the last
 * failure on the calling thread.  Caller owns every buffer it passes.  A handle is bound to one
 * device and is not thread-safe; distinct handles on distinct devices may run concurrently.
 * Pointers passed to upload/download may be host or device pointers (hipMemcpyDefault).
 */
#ifndef APD_MI355X_H_
#define APD_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APD_MAX_IMAGES 32        /* main.h:37 MAX_IMAGES */
#define APD_NEIGHBOUR_NUM 9      /* main.h:38 NEIGHBOUR_NUM */
#define APD_MAX_SEARCH_RADIUS 4096 /* main.h:39 */

typedef enum { APD_FIRST_INIT = 0, APD_REFINE_INIT = 1, APD_REFINE_ITER = 2 } apd_run_state; /* main.h:63-67 */
typedef enum { APD_WEAK = 0, APD_STRONG = 1, APD_UNKNOWN = 2 } apd_pixel_state;              /* main.h:69-73 */

typedef enum {
    APD_OK = 0,
    APD_ERR_INVALID = -1,   /* bad argument */
    APD_ERR_HIP = -2,       /* HIP runtime error (reference: CudaSafeCall -> exit, APD.cpp:315-323) */
    APD_ERR_TOO_MANY = -3,  /* > APD_MAX_IMAGES images (reference: exit, APD.cpp:428-431) */
    APD_ERR_STATE = -4,     /* call order violated (e.g. run before upload) */
    APD_ERR_UNSUPPORTED = -5,
    APD_ERR_IO = -6         /* a file could not be written */
} apd_status;

/* Byte-compatible with the reference `Camera` (main.h:47-56, 112 bytes). */
typedef struct apd_camera {
    float K[9];
    float R[9];
    float t[3];
    float c[3];
    int height;
    int width;
    float depth_min;
    float depth_max;
} apd_camera;

/* Fields of `PatchMatchParams` (main.h:75-94) in the reference's order, then the additive knobs. */
typedef struct apd_params {
    int max_iterations;     /* 3 */
    int num_images;         /* set by apd_upload_views */
    float sigma_spatial;    /* 5.0  (dead in the reference: weight == 1, APD.cu:473,575) */
    float sigma_color;      /* 3.0  (dead) */
    int top_k;              /* 4 */
    float depth_min;        /* 0.6 * ref camera depth_min (APD.cpp:454) */
    float depth_max;        /* 1.2 * ref camera depth_max (APD.cpp:455) */
    int geom_consistency;   /* bool */
    int strong_radius;      /* 5 */
    int strong_increment;   /* 2 */
    int weak_radius;        /* 5 */
    int weak_increment;     /* 5 */
    int use_APD;            /* bool */
    int weak_peak_radius;   /* 2 */
    int rotate_time;        /* 4 */
    float ransac_threshold; /* 0.005 */
    float geom_factor;      /* 0.2 */
    int state;              /* apd_run_state */
    /* ---- additive knobs (absent in the reference; defaults keep reference behaviour) ---- */
    uint64_t seed;          /* replaces clock64() in curand_init, APD.cu:803 */
} apd_params;

typedef struct apd_context *apd_handle;

/* Kernel ids for apd_run_kernel / apd_profile_get. */
enum {
    APD_K1_INIT_RANDOM_STATES = 1,   /* APD.cu:791  */
    APD_K2_FIND_NEAREST_STRONG = 2,  /* APD.cu:2234 */
    APD_K3_GEN_NEIGHBOURS = 3,       /* APD.cu:1750 */
    APD_K4_NEIGHBOUR_UPDATE = 4,     /* APD.cu:1971 */
    APD_K5_RANDOM_INITIALIZATION = 5,/* APD.cu:806  */
    APD_K6_BLACK_UPDATE_STRONG = 6,  /* APD.cu:1547 */
    APD_K7_RED_UPDATE_STRONG = 7,    /* APD.cu:1567 */
    APD_K8_RANSAC_FIT_PLANE = 8,     /* APD.cu:2272 */
    APD_K9_BLACK_UPDATE_WEAK = 9,    /* APD.cu:1510 */
    APD_K10_RED_UPDATE_WEAK = 10,    /* APD.cu:1529 */
    APD_K11_GET_DEPTH_NORMAL = 11,   /* APD.cu:1587 */
    APD_K12_BLACK_FILTER = 12,       /* APD.cu:1716 */
    APD_K13_RED_FILTER = 13,         /* APD.cu:1733 */
    APD_K14_DEPTH_TO_WEAK = 14,      /* APD.cu:1990 */
    APD_K15_LOCAL_REFINE = 15,       /* APD.cu:2146 */
    APD_KERNEL_COUNT = 16
};

/* State arrays readable with apd_download_state (tests / snapshots). */
enum {
    APD_STATE_PLANES = 0,        /* float4  [H*W]      plane_hypotheses_cuda      */
    APD_STATE_FIT_PLANES = 1,    /* float4  [H*W]      fit_plane_hypotheses_cuda  */
    APD_STATE_COSTS = 2,         /* float   [H*W]      costs_cuda                 */
    APD_STATE_RNG = 3,           /* uint32  [H*W*6]    rand_states_cuda (x0..x4,d)*/
    APD_STATE_SELECTED_VIEWS = 4,/* uint32  [H*W]      selected_views_cuda        */
    APD_STATE_VIEW_WEIGHT = 5,   /* uint8   [H*W*32]   view_weight_cuda           */
    APD_STATE_WEAK_INFO = 6,     /* uint8   [H*W]      weak_info_cuda             */
    APD_STATE_WEAK_RELIABLE = 7, /* uint8   [H*W]      weak_reliable_cuda         */
    APD_STATE_NEAREST_STRONG = 8,/* short2  [H*W]      weak_nearest_strong        */
    APD_STATE_NEIGHBOURS_MAP = 9,/* int32   [H*W]      neighbours_map_cuda        */
    APD_STATE_NEIGHBOURS = 10    /* short2  [weak*9]   neighbours_cuda            */
};

/* Defaults of PatchMatchParams (main.h:75-94), seed = 12345. */
void apd_default_params(apd_params *p);

/* APD::APD(const Problem&) (APD.cpp:356-359) + the allocations of CudaSpaceInitialization
 * (APD.cpp:636-666).  `device` < 0 keeps the current device (reference: cudaSetDevice, main.cpp:153).
 * Limits (APD_ERR_UNSUPPORTED otherwise): width, height <= 16384 (16-bit neighbour coordinates, 24-bit index
 * arithmetic); patch geometry strong 5/2, weak 5/5. */
int apd_create(apd_handle *out, int device, int width, int height, const apd_params *params);

/* ~APD (APD.cpp:361-397). */
int apd_destroy(apd_handle h);

/* Re-arms a handle for another (view, pass) of the same width x height with new parameters: the state arrays return to
 * what apd_create leaves (so a recycled handle gives the same bits as a new one), uploads are forgotten, device buffers
 * are kept.  Saves the ~25 hipMalloc/hipFree pairs per (view, pass) of the construct-run-destroy cycle of
 * ProcessProblem (main.cpp:91-138). */
int apd_reset(apd_handle h, const apd_params *params);

/* Image / depth / camera upload of CudaSpaceInitialization (APD.cpp:588-634).  images[0] is the
 * reference view; `depths` may be NULL unless params.geom_consistency.  All images are W*H floats,
 * row-major, no padding.  Sets params.num_images. */
int apd_upload_views(apd_handle h, int num_images, const apd_camera *cameras, const float *const *images,
                     const float *const *depths);

/* A geometric pass in two halves, for a scheduler that keeps several views in flight: the sources' depth maps are the only
 * input of a (view, pass) that other views of the same pass produce (the reference reads depths.dmb as the files are at that
 * moment, APD.cpp:497-500), and only the weak update (K9/K10), K14 and K15 read them (ComputeGeomConsistencyCost, APD.cu:752).
 *   apd_upload_views_split   = apd_upload_views without depth maps (their buffers are allocated);
 *   apd_run_before_depths    = the longest prefix of the schedule that reads no depth map: K1..K5, iteration 0 of K6..K8 and --
 *                              while no WEAK pixel exists -- the other iterations and K11..K13; without the geometric term the
 *                              whole pass;
 *   apd_upload_depths        = the num_images depth maps (index 0: the view's own), copied on the handle's stream, i.e. after the
 *                              kernels already launched; returns when the copies are done;
 *   apd_run_after_depths     = the rest of the schedule.
 * before + after launch exactly the kernels of apd_run, in its order: same bits.  K9, K10, K14 and K15 return APD_ERR_STATE
 * while the depth maps of a split upload are outstanding. */
int apd_upload_views_split(apd_handle h, int num_images, const apd_camera *cameras, const float *const *images);
int apd_upload_depths(apd_handle h, int num_images, const float *const *depths);
int apd_run_before_depths(apd_handle h);
int apd_run_after_depths(apd_handle h);

/* Shared images.  A pyramid-level image of a view is the reference image of one (view, pass) and a source of ten others, pass after
 * pass; apd_upload_views copies it into the handle, tests it for 8-bit content and packs it again every time.  A scheduler that
 * keeps the level images on the device creates each one ONCE (pixels: width x height floats, host or device; the float plane, the
 * 8-bit test, the packed copy the kernels gather from -- further copies on first demand) and uploads views by reference:
 * apd_upload_views_shared == apd_upload_views_split without the copies (images[0] the reference view; every image of the handle's
 * size and on its device; in a geometric pass the depth maps follow with apd_upload_depths).  An image may serve any number of
 * handles on any threads at once and must outlive the passes that use it.  Same bits as the copying uploads. */
typedef struct apd_image *apd_image_t;
int apd_image_create(apd_image_t *out, int device, int width, int height, const float *pixels);
int apd_image_destroy(apd_image_t image);
const float *apd_image_pixels(apd_image_t image);   /* the float plane on the device (e.g. for a copying apd_upload_views of the same image) */
int apd_upload_views_shared(apd_handle h, int num_images, const apd_camera *cameras, const apd_image_t *images);

/* Prior state of a previous pass (APD.cpp:552-581, 643-661): planes = (world normal xyz, depth w),
 * selected-view bitmasks and weak map.  Any pointer may be NULL: planes/views zero, weak = all STRONG
 * (APD.cpp:541-547).  Builds the weak index map of APD.cpp:526-537. */
int apd_upload_prior(apd_handle h, const float *planes4, const uint32_t *selected_views, const uint8_t *weak_info);

/* APD::RunPatchMatch (APD.cu:2386-2495), without the final device->host copies.  One handle is one (view, pass), like one
 * APD object: K14 rewrites the weak map the WEAK lists, the neighbour table and its index map were sized for, so a second
 * apd_run -- or K3 / K8 / K9 / K10 through apd_run_kernel after K14 or after apd_upload_state(APD_STATE_WEAK_INFO) -- returns
 * APD_ERR_STATE until apd_upload_prior or apd_reset re-arms the handle. */
int apd_run(apd_handle h);

/* One kernel of the schedule (single stepping for snapshot tests). */
int apd_run_kernel(apd_handle h, int kernel_id, int iter);

/* The metric's timed region: loop body APD.cu:2443-2457 (K6,K7,K8,K9,K10) for
 * iter = first_iter .. first_iter+iters-1.  Asynchronous; pair with apd_synchronize. */
int apd_run_sweeps(apd_handle h, int first_iter, int iters);

int apd_synchronize(apd_handle h);

/* The three copies at the end of RunPatchMatch (APD.cu:2490-2492) == GetPlaneHypothesis /
 * GetPixelStates / GetSelectedViews (APD.cpp:701-711).  Any pointer may be NULL. */
int apd_download(apd_handle h, float *planes4, uint8_t *weak_info, uint32_t *selected_views);

/* Raw copy of one state array (see APD_STATE_*).  `bytes` must not exceed the array size. */
int apd_download_state(apd_handle h, int which, void *dst, size_t bytes);
int apd_upload_state(apd_handle h, int which, const void *src, size_t bytes);
size_t apd_state_bytes(apd_handle h, int which);

/* Post-processing of ProcessProblem (main.cpp:105-115) done on the device: depth = plane.w with
 * out-of-range -> 0, normal = plane.xyz.  `depth_dev` (W*H floats) and `normal_dev` (3*W*H floats)
 * are DEVICE pointers (e.g. torch tensors handed to an RCCL all-gather). */
int apd_export_depth_normal_device(apd_handle h, float *depth_dev, float *normal_dev);

/* The same post-processing in the layout apd_upload_prior takes, left on the device (a scheduler that keeps state resident
 * between passes): planes4 = (world normal xyz, depth w) with an out-of-range depth -> 0 and its pixel UNKNOWN in `weak`
 * (main.cpp:109-112), selected views, and the depth map alone (what the geometric term of the next pass reads,
 * APD.cpp:492-509).  DEVICE pointers, W*H elements each; any may be NULL. */
int apd_export_state_device(apd_handle h, float *planes4_dev, uint8_t *weak_dev, uint32_t *views_dev, float *depth_dev);
/* The HIP event (hipEvent_t) the handle records on its stream behind the kernel of its last apd_export_state_device /
 * apd_export_depth_normal_device (NULL before the first export): what a consumer on another stream waits for
 * (apd_exchange_allgather_after).  Owned by the handle, re-recorded by every export. */
int apd_export_event(apd_handle h, void **hip_event);

/* ---- several devices in one process (one host thread + one stream per device) ----------------------------
 * The reference takes one device index (main.cpp:149-153).  A multi-device host shards the reference views over devices
 * and needs, besides the handles above (one per device, each used from its own thread): device memory without a HIP
 * toolchain, the nearest-neighbour resampling of prior state between pyramid levels (RescaleMatToTargetSize,
 * APD.cpp:752-774, swapped factors included) on the device, and the all-gather of per-view maps after every pass -- the
 * exchange the reference does through depths.dmb files (APD.cpp:497-500). */
int apd_device_malloc(int device, size_t bytes, void **out);
int apd_device_free(int device, void *p);
int apd_device_memcpy(int device, void *dst, const void *src, size_t bytes);   /* host or device pointers on either side */
int apd_device_memset(int device, void *dst, int value, size_t bytes);
int apd_device_memory(int device, size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo: does an in-memory run fit? */
int apd_rescale_nearest_device(int device, const void *src, int src_w, int src_h, void *dst, int dst_w, int dst_h, int elem_bytes /* 1, 4, 16 */);
/* The same helpers on a HIP stream (hipStream_t, e.g. apd_get_stream of the handle whose kernels produce or consume the data):
 * asynchronous, ordered with that stream's work and with nothing else -- the device-wide forms above wait for the whole device,
 * which stalls every other view in flight on it.  apd_stream_synchronize waits for that stream alone. */
int apd_device_memcpy_async(int device, void *hip_stream, void *dst, const void *src, size_t bytes);
int apd_rescale_nearest_async(int device, void *hip_stream, const void *src, int src_w, int src_h, void *dst, int dst_w, int dst_h, int elem_bytes);
int apd_stream_synchronize(int device, void *hip_stream);
int apd_stream_create(int device, void **hip_stream);    /* a non-blocking stream of its own (not ordered with the null stream) */
int apd_stream_destroy(int device, void *hip_stream);
/* (float4 plane = world normal xyz + depth w) -> the depth map and the 3-float normal map apd_fuse_views takes; device pointers. */
int apd_split_planes_async(int device, void *hip_stream, const float *planes4, size_t pixels, float *depth, float *normal3);
/* Page-locks / releases a host buffer (hipHostRegister): uploads from it run at the link's rate and asynchronously.
 * apd_host_alloc / apd_host_free: a page-locked buffer of its own (hipHostMalloc) -- a staging buffer that is mapped ONCE: every
 * map / unmap of host pages (hipHostRegister, and the on-the-fly pinning a plain hipMemcpy of pageable memory does) holds up the
 * kernels running on the device for milliseconds. */
int apd_host_register(void *p, size_t bytes);
int apd_host_unregister(void *p);
int apd_host_alloc(size_t bytes, void **out);
int apd_host_free(void *p);

/* All-gather across `num_ranks` ranks of this process, rank r on devices[r]: after apd_exchange_allgather every recv[r]
 * holds send[0] | send[1] | ... (bytes_per_rank each).  prefer_rccl != 0: RCCL (ncclCommInitAll, grouped ncclAllGather,
 * one stream per rank; librccl is opened at run time); direct hipMemcpyPeerAsync copies when librccl is missing, when its
 * initialisation fails or when there is a single rank.  A list that names devices more than once (several scheduler ranks
 * per device, e.g. 0,1,0,1) runs RCCL between one leader rank per device and copies inside the devices; 0,0,0 (one device)
 * uses copies unless RCCL is forced.  Blocking; not thread-safe per exchange object. */
typedef struct apd_exchange *apd_exchange_t;
/* prefer_rccl != 0: RCCL, set up before the call returns.  That takes seconds on a fresh box (5.0 s to dlopen librccl from a cold page
 * cache, 1.0 s warm; ncclCommInitAll 0.65 s for one device: profiles/r05/rccl_init_time.txt); a caller with a single rank should pass 0.
 * (Round 5's asynchronous set-up -- preload thread, communicators initialised behind the first passes -- measured slower and was removed
 * in round 6: profiles/r05/ab_rccl_async_tt24.txt.)  apd_exchange_setup_times: what the dlopen and the initialisation took. */
int apd_exchange_create(apd_exchange_t *out, int num_ranks, const int *devices, int prefer_rccl);
int apd_exchange_setup_times(apd_exchange_t x, double *dlopen_ms, double *init_ms);
int apd_exchange_allgather(apd_exchange_t x, const void *const *send, void *const *recv, size_t bytes_per_rank);
/* ... without the device-wide synchronisation: the exchange's streams wait for the `num_events` HIP events (hipEvent_t; NULL entries are
 * skipped) that mark the send buffers complete -- apd_export_event of every handle that exported a block -- and for nothing else, so
 * kernels queued by other host threads (the next pass's first halves) keep running beside the exchange.  The caller knows the recv
 * buffers to be idle. */
int apd_exchange_allgather_after(apd_exchange_t x, const void *const *send, void *const *recv, size_t bytes_per_rank, int num_events,
                                 void *const *hip_events);
const char *apd_exchange_backend(apd_exchange_t x);   /* "rccl" or "peer-copy" */
int apd_exchange_counts(apd_exchange_t x, int *with_rccl, int *with_copies);   /* exchanges served by either backend so far */
int apd_exchange_destroy(apd_exchange_t x);
const char *apd_exchange_last_error(void);

/* Getters of APD.h:76-81. */
int apd_width(apd_handle h);
int apd_height(apd_handle h);
float apd_depth_min(apd_handle h);
float apd_depth_max(apd_handle h);
int apd_weak_count(apd_handle h);

/* Per-kernel timing with HIP events on the handle's stream. */
int apd_profile_enable(apd_handle h, int on);
int apd_profile_reset(apd_handle h);
int apd_profile_get(apd_handle h, int kernel_id, double *total_ms, int *launches);

/* Options of one handle.  The library reads NOTHING from the process environment: every switch is set here, explicitly,
 * and can be read back.  All options but APD_OPT_FAST_RCP leave every result bit unchanged (they select between
 * implementations that the test-suite compares bit for bit); APD_OPT_FAST_RCP is a tolerance mode and is off by default.
 * Options marked (upload) are latched by the next apd_upload_views. */
enum {
    APD_OPT_FAST_RCP = 0,       /* 0 (default) exact reciprocals: results are the arithmetic contract's bits.  1: the K6/K7 sample
                                 * loops stop at v_rcp_f32 (<= 1 ulp), what the reference's --use_fast_math build does
                                 * (CMakeLists.txt:20); NOT bit-identical, see tests/test_gpu_fast_rcp.py */
    APD_OPT_EARLY_OUT = 1,      /* 1 (default) exact early-outs of the refinement loops, K14 and K15; 0: every NCC the reference evaluates */
    APD_OPT_SOURCE_QUADS = 2,   /* (upload) 1 (default) 8-bit inputs are also kept as packed texel pairs; 0: float texel quads for every input */
    APD_OPT_TILED_COPY = 3,     /* (upload) second, tiled copy of 8-bit sources for random gathers: 0 never, 1 (default) built for FIRST_INIT
                                 * passes and used while the planes are random (first iteration), 2 built always, used by every global gather */
    APD_OPT_K67_WINDOWS = 4,    /* 1 (default) K6/K7 with LDS source windows; 0: every sample from HBM */
    APD_OPT_K1415_WINDOWS = 5,  /* 1 (default) K14/K15 with LDS source windows */
    APD_OPT_COUNT = 6
};
int apd_set_option(apd_handle h, int option, int value);
int apd_get_option(apd_handle h, int option, int *value);

/* Use an existing HIP stream (hipStream_t) instead of the handle's own. */
int apd_set_stream(apd_handle h, void *hip_stream);
/* The stream the handle launches on (hipStream_t), for the stream forms of the device helpers. */
int apd_get_stream(apd_handle h, void **hip_stream);

/* Depth-map fusion on the device: what RunFusion (APD.cpp:826-977, ETH variant) + ExportPointCloud (APD.cpp:214-254)
 * produce, i.e. <dense>/APD/APD.ply, from the final maps of every view (after the all-gather in a multi-GPU run).
 * View i has cameras[i] (K already scaled to the map size, APD.cpp:729-750), image images[i] (floats 0..255,
 * image_channels = 1: grey, 3: blue, green, red interleaved as cv::imread(IMREAD_COLOR) gives them, APD.cpp:859),
 * depths[i] (<= 0: no estimate), normals[i] (3 floats per pixel, world frame), weaks[i] (PixelState), optionally
 * blocks[i] (the `blocks/mask_<id>.jpg` of APD.cpp:849-853: reference pixels below 128 are skipped; `blocks` or any
 * blocks[i] may be NULL), all
 * rows[i] x cols[i]; its sources are pair_indices[pair_offsets[i] .. pair_offsets[i+1]) (indices of views, in
 * pair.txt order).  maps_on_device != 0: the four map arrays hold DEVICE pointers (e.g. the gathered torch tensors).
 * Views are fused in order, and inside a view the raster-order consumption of source pixels (`masks`) is resolved
 * exactly, so the point list is the one the sequential host loop writes.  A view that lists itself as a source is
 * refused (APD_ERR_INVALID). */
int apd_fuse_views(int device, int num_views, const apd_camera *cameras, const float *const *images, int image_channels,
                   const float *const *depths, const float *const *normals, const uint8_t *const *weaks,
                   const uint8_t *const *blocks, const int *rows, const int *cols, const int *pair_offsets, const int *pair_indices, int maps_on_device, const char *ply_path,
                   long long *num_points);
const char *apd_fusion_last_error(void);
/* Where the last apd_fuse_views of this thread spent its time (ms): set-up (allocations, uploads of host maps), the views (kernels,
 * consumption rounds, download of the points), the PLY file (release of the buffers + write). */
int apd_fusion_last_timing(double *setup_ms, double *views_ms, double *file_ms);

/* Host-side constant of K3 (GenNeighbours, APD.cu:1911 / :1946): its inlier test `dist / (depth_max - depth_min) <
 * ransac_threshold` (dist >= 0) is evaluated on the device as `dist < cut`, the same predicate for every binary32 dist because
 * x -> RN(x / d) is monotone.  Returns 1 and the cut, or 0 when the parameters admit none (the kernel then divides).  Needs
 * no device; exported so that the equivalence can be tested on any machine (tests/test_host_constants.py). */
int apd_ransac_distance_cut(float depth_min, float depth_max, float ransac_threshold, float *cut);

const char *apd_last_error(void);
int apd_version(void);
/* Digest (16 hex digits) of the HIP sources, headers and compiler flags the library was built from: equals
 * apd-mvs_amd/build.py:expected_build_id() of the same tree; anything else is a stale binary. */
const char *apd_build_id(void);
int apd_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* APD_MI355X_H_ */
