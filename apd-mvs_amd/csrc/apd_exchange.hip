// apd_exchange.hip -- the pieces a multi-device host needs around the PatchMatch handles (include/apd_mi355x.h, "several
// devices in one process"): device memory for callers built without a HIP toolchain (the C++ drop-in is compiled with g++),
// the post-processed maps of a finished pass left on the device, the reference's nearest-neighbour resampling of prior
// state between pyramid levels on the device, and the all-gather of per-view maps across the devices of one process.
//
// The all-gather is RCCL's (ncclCommInitAll + one grouped ncclAllGather per call, every rank on its own stream), i.e. the
// xGMI path.  librccl is opened at run time (dlopen): the library has no link-time dependency on it, and a missing librccl falls
// back to direct reads (one gather kernel per rank over peer-mapped send buffers; hipMemcpyPeerAsync where a pair of devices has no
// peer access), which give the same bytes.  A device list that repeats itself ("0,1,2,3,0,1,2,3":
// two scheduler ranks per device) runs RCCL between one leader rank per device and copies inside the devices; a list without that
// structure ("0,0,1") uses direct copies.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include <algorithm>
#include <chrono>
#include <mutex>

#include "apd_device.h"

namespace apd {

// RescaleMatToTargetSize (APD.cpp:752-774): dst(r, c) = src((int)(r / scale_x), (int)(c / scale_y)) with
// scale_x = dst_w / (float)src_w and scale_y = dst_h / (float)src_h -- the row is divided by the COLUMN ratio and the column by
// the ROW ratio; pixels whose source index falls outside become 0 (what host/APD.cpp and pipeline.py define for the reference's unwritten pixels).
template <typename T>
__global__ __launch_bounds__(256) void k_rescale_nearest(const T *__restrict__ src, int sw, int sh, T *__restrict__ dst, int dw, int dh)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= dw) {
        return;
    }
    const float scale_x = (float)dw / (float)sw;
    const float scale_y = (float)dh / (float)sh;
    const int o_r = (int)((float)r / scale_x);
    const int o_c = (int)((float)c / scale_y);
    if (o_r < 0 || o_c < 0 || o_r >= sh || o_c >= sw) {
        dst[(size_t)r * dw + c] = T();  // the reference leaves these pixels of a fresh cv::Mat unwritten; the host restatements define them as 0
        return;
    }
    dst[(size_t)r * dw + c] = src[(size_t)o_r * sw + o_c];
}

struct Bytes16 {
    uint32_t v[4];
};

// ProcessProblem's post-processing (main.cpp:105-115) into the layout apd_upload_prior takes: planes4 = (world normal,
// depth), a depth outside [depth_min, depth_max] becomes 0 and its pixel UNKNOWN.
__global__ __launch_bounds__(256) void k_export_state(FrameArgs fa, float4 *planes4, uint8_t *weak, uint32_t *views, float *depth)
{
    const int center = blockIdx.x * 256 + threadIdx.x;
    if (center >= fa.W * fa.H) {
        return;
    }
    float4 pl = fa.planes[center];
    uint8_t w = fa.weak_info[center];
    if (pl.w < fa.depth_min || pl.w > fa.depth_max) {  // false for NaN, as in the reference
        pl.w = 0.0f;
        w = APD_UNKNOWN;
    }
    if (planes4) {
        planes4[center] = pl;
    }
    if (weak) {
        weak[center] = w;
    }
    if (views) {
        views[center] = fa.selected_views[center];
    }
    if (depth) {
        depth[center] = pl.w;
    }
}

hipError_t launch_export_state(const FrameArgs &fa, float4 *planes4, uint8_t *weak, uint32_t *views, float *depth, hipStream_t s)
{
    hipLaunchKernelGGL(k_export_state, dim3((fa.W * fa.H + 255) / 256), dim3(256), 0, s, fa, planes4, weak, views, depth);
    return hipGetLastError();
}

}  // namespace apd

static thread_local std::string g_exchange_error;

static int xfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_exchange_error = buf;
    return code;
}

#define X_TRY(expr)                                                                                                        \
    do {                                                                                                                   \
        hipError_t e_ = (expr);                                                                                            \
        if (e_ != hipSuccess) {                                                                                            \
            return xfail(APD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);           \
        }                                                                                                                  \
    } while (0)

// the six RCCL entry points, resolved with dlsym (signatures of rccl.h 2.2x)
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int datatype, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::mutex m;          // exchanges may be created from several host threads
    double load_ms = 0.0;  // what the first successful load() took (dlopen of a library with code objects for every architecture); written under m
    double loaded_ms()
    {
        std::lock_guard<std::mutex> lock(m);
        return load_ms;
    }
    bool load()
    {
        std::lock_guard<std::mutex> lock(m);
        if (lib) {
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) {
                break;
            }
        }
        if (!lib) {
            return false;
        }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return true;
    }
};
static RcclApi g_rccl;
constexpr int kNcclInt8 = 0;  // ncclDataType_t ncclInt8 / ncclChar: the payload is moved as bytes

// RCCL's set-up takes seconds -- 5.6 s for ONE device on a fresh MI355X box, more than all eight passes of a 12-view 1080p
// reconstruction (4.0 s): 5.0 s of it are the dlopen of librccl.so on a cold page cache (1.0 s warm), ncclCommInitAll is 0.65 s for one
// device (tools/rccl_init_time.hip, profiles/r05/rccl_init_time.txt).  Round 5 tried to run both behind the first passes (a preload
// thread + communicators initialised on a thread, exchanges through copies until RCCL was ready): the dlopen stalls every HIP call of
// the other threads for as long as it runs and the passes beside the initialisation took 9.3 instead of 7.5 s
// (profiles/r05/ab_rccl_async_tt24.txt) -- slower; the code was removed in round 6, the set-up is blocking.  A caller with one rank has
// nothing to exchange between devices and should ask for direct copies (host/multi_device.cpp does).
enum { kRcclOff = 0, kRcclReady = 2, kRcclFailed = 3 };

struct apd_exchange {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    // Several ranks per device ("0,1,2,3,0,1,2,3": the list repeats its first `period` distinct entries): RCCL runs between
    // one leader rank per device -- ranks 0 .. period - 1, one communicator entry each -- once per repetition, and the other ranks of
    // a device copy the gathered blocks from their leader, inside the device.  period == number of ranks: the plain case.
    int period = 0;
    std::vector<void *> comms;  // ncclComm_t per leader rank; used only when rccl_state == kRcclReady
    int rccl_state = kRcclOff;
    std::vector<char> peer_ok;  // [dst * n + src]: rank dst's device can read rank src's memory from a kernel (same device, or peer access enabled)
    std::vector<hipEvent_t> done;   // one per rank: recorded behind a rank's part of an exchange (the other ranks of its device wait for the leader's)
    std::string rccl_error;
    int exchanges_rccl = 0, exchanges_copy = 0;
    std::string backend;        // what apd_exchange_backend last reported
    double init_ms = 0.0;       // dlopen (first exchange of the process) + ncclCommInitAll inside apd_exchange_create
};

static void rccl_initialise(apd_exchange *x)
{
    const auto t0 = std::chrono::steady_clock::now();
    int state = kRcclReady;
    const int n = x->period;  // one communicator entry per distinct device
    if (!g_rccl.load()) {
        x->rccl_error = "librccl not found";
        state = kRcclFailed;
    } else {
        x->comms.assign(n, nullptr);
        const int rc = g_rccl.CommInitAll(x->comms.data(), n, x->devices.data());
        if (rc != 0) {
            x->rccl_error = g_rccl.GetErrorString(rc);
            x->comms.clear();
            state = kRcclFailed;
        }
    }
    x->init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    x->rccl_state = state;
}

extern "C" {

const char *apd_exchange_last_error(void) { return g_exchange_error.c_str(); }

int apd_device_malloc(int device, size_t bytes, void **out)
{
    if (!out) {
        return xfail(APD_ERR_INVALID, "apd_device_malloc: null result pointer");
    }
    *out = nullptr;
    X_TRY(hipSetDevice(device));
    X_TRY(hipMalloc(out, bytes > 0 ? bytes : 1));
    return APD_OK;
}

int apd_device_free(int device, void *p)
{
    if (!p) {
        return APD_OK;
    }
    X_TRY(hipSetDevice(device));
    X_TRY(hipFree(p));
    return APD_OK;
}

int apd_device_memcpy(int device, void *dst, const void *src, size_t bytes)
{
    X_TRY(hipSetDevice(device));
    X_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDefault));  // host / device on either side (unified addressing)
    // a device-to-device hipMemcpy may return before the copy has run; callers hand `dst` to other streams (the exchange's
    // are non-blocking ones, which do not wait for the null stream) and to other threads
    X_TRY(hipStreamSynchronize(nullptr));
    return APD_OK;
}

int apd_device_memset(int device, void *dst, int value, size_t bytes)
{
    X_TRY(hipSetDevice(device));
    X_TRY(hipMemset(dst, value, bytes));
    X_TRY(hipStreamSynchronize(nullptr));  // hipMemset of device memory is asynchronous
    return APD_OK;
}

int apd_device_memory(int device, size_t *free_bytes, size_t *total_bytes)
{
    size_t f = 0, t = 0;
    X_TRY(hipSetDevice(device));
    X_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) {
        *free_bytes = f;
    }
    if (total_bytes) {
        *total_bytes = t;
    }
    return APD_OK;
}

static int rescale_nearest_on(hipStream_t st, const void *src, int src_w, int src_h, void *dst, int dst_w, int dst_h, int elem_bytes, const char *who)
{
    if (!src || !dst || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0) {
        return xfail(APD_ERR_INVALID, "%s: bad argument", who);
    }
    if (src_w == dst_w && src_h == dst_h) {  // the reference returns before touching dst (APD.cpp:754-756); callers want the copy
        X_TRY(hipMemcpyAsync(dst, src, (size_t)src_w * src_h * elem_bytes, hipMemcpyDeviceToDevice, st));
        return APD_OK;
    }
    const dim3 grid((dst_w + 255) / 256, dst_h);
    switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(apd::k_rescale_nearest<uint8_t>, grid, dim3(256), 0, st, (const uint8_t *)src, src_w, src_h, (uint8_t *)dst, dst_w, dst_h); break;
    case 4: hipLaunchKernelGGL(apd::k_rescale_nearest<uint32_t>, grid, dim3(256), 0, st, (const uint32_t *)src, src_w, src_h, (uint32_t *)dst, dst_w, dst_h); break;
    case 16: hipLaunchKernelGGL(apd::k_rescale_nearest<apd::Bytes16>, grid, dim3(256), 0, st, (const apd::Bytes16 *)src, src_w, src_h, (apd::Bytes16 *)dst, dst_w, dst_h); break;
    default: return xfail(APD_ERR_INVALID, "%s: element size %d (1, 4 or 16 bytes)", who, elem_bytes);
    }
    X_TRY(hipGetLastError());
    return APD_OK;
}

int apd_rescale_nearest_device(int device, const void *src, int src_w, int src_h, void *dst, int dst_w, int dst_h, int elem_bytes)
{
    X_TRY(hipSetDevice(device));
    const int rc = rescale_nearest_on(nullptr, src, src_w, src_h, dst, dst_w, dst_h, elem_bytes, "apd_rescale_nearest_device");
    if (rc != APD_OK) {
        return rc;
    }
    X_TRY(hipDeviceSynchronize());
    return APD_OK;
}

int apd_rescale_nearest_async(int device, void *hip_stream, const void *src, int src_w, int src_h, void *dst, int dst_w, int dst_h, int elem_bytes)
{
    X_TRY(hipSetDevice(device));
    return rescale_nearest_on((hipStream_t)hip_stream, src, src_w, src_h, dst, dst_w, dst_h, elem_bytes, "apd_rescale_nearest_async");
}

int apd_device_memcpy_async(int device, void *hip_stream, void *dst, const void *src, size_t bytes)
{
    X_TRY(hipSetDevice(device));
    X_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)hip_stream));
    return APD_OK;
}

int apd_stream_create(int device, void **hip_stream)
{
    if (!hip_stream) {
        return xfail(APD_ERR_INVALID, "apd_stream_create: null result pointer");
    }
    hipStream_t st = nullptr;
    X_TRY(hipSetDevice(device));
    X_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *hip_stream = (void *)st;
    return APD_OK;
}

int apd_stream_destroy(int device, void *hip_stream)
{
    if (!hip_stream) {
        return APD_OK;
    }
    X_TRY(hipSetDevice(device));
    X_TRY(hipStreamDestroy((hipStream_t)hip_stream));
    return APD_OK;
}

int apd_stream_synchronize(int device, void *hip_stream)
{
    X_TRY(hipSetDevice(device));
    X_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    return APD_OK;
}

namespace apd {
__global__ __launch_bounds__(256) void k_split_planes(const float4 *__restrict__ planes, size_t n, float *__restrict__ depth, float *__restrict__ normal)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float4 p = planes[i];
        depth[i] = p.w;
        normal[3 * i] = p.x;
        normal[3 * i + 1] = p.y;
        normal[3 * i + 2] = p.z;
    }
}
}  // namespace apd

int apd_split_planes_async(int device, void *hip_stream, const float *planes4, size_t pixels, float *depth, float *normal3)
{
    if (!planes4 || !depth || !normal3) {
        return xfail(APD_ERR_INVALID, "apd_split_planes_async: null pointer");
    }
    X_TRY(hipSetDevice(device));
    if (pixels > 0) {
        hipLaunchKernelGGL(apd::k_split_planes, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                           reinterpret_cast<const float4 *>(planes4), pixels, depth, normal3);
        X_TRY(hipGetLastError());
    }
    return APD_OK;
}

int apd_host_register(void *p, size_t bytes)
{
    if (!p || bytes == 0) {
        return xfail(APD_ERR_INVALID, "apd_host_register: bad argument");
    }
    X_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return APD_OK;
}

int apd_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) {
        return xfail(APD_ERR_INVALID, "apd_host_alloc: bad argument");
    }
    *out = nullptr;
    X_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return APD_OK;
}

int apd_host_free(void *p)
{
    if (!p) {
        return APD_OK;
    }
    X_TRY(hipHostFree(p));
    return APD_OK;
}

int apd_host_unregister(void *p)
{
    if (!p) {
        return APD_OK;
    }
    X_TRY(hipHostUnregister(p));
    return APD_OK;
}

namespace apd {
// The peer-copy all-gather of one rank as ONE launch: workgroup (x, src) copies 16-byte words of rank src's send buffer (its own
// device's memory, or a peer's mapped over xGMI) into block src of this rank's result.  Replaces num_ranks hipMemcpy(Peer)Async calls
// per rank, which the runtime serves with its blit path one after the other (1.3 s of a 24 x 1080p run, profiles/r05/ab_rccl_async_tt24.txt).
constexpr int kGatherMaxRanks = 64;
typedef unsigned int word4 __attribute__((ext_vector_type(4)));   // a plain 16-byte vector: what the nontemporal builtins take
struct GatherArgs {
    const word4 *src[kGatherMaxRanks];
};

__global__ __launch_bounds__(256) void k_gather_blocks(GatherArgs a, word4 *__restrict__ dst, size_t words_per_rank, size_t tail_bytes)
{
    const int r = blockIdx.y;
    const word4 *__restrict__ src = a.src[r];
    if (!src) {   // this block goes through hipMemcpyPeerAsync (no peer access for the pair)
        return;
    }
    word4 *__restrict__ out = dst + (size_t)r * words_per_rank;   // only reached with tail_bytes == 0 for r > 0 (see the caller)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words_per_rank; i += stride) {
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), out + i);   // streamed once: keep it out of the way of the sweeps' L2 lines
    }
    if (tail_bytes && blockIdx.x == 0 && threadIdx.x < tail_bytes) {   // bytes_per_rank not a multiple of 16: single rank layout only
        reinterpret_cast<uint8_t *>(out + words_per_rank)[threadIdx.x] = reinterpret_cast<const uint8_t *>(src + words_per_rank)[threadIdx.x];
    }
}
}  // namespace apd

int apd_exchange_create(apd_exchange_t *out, int num_ranks, const int *devices, int prefer_rccl)
{
    if (!out || num_ranks < 1 || !devices) {
        return xfail(APD_ERR_INVALID, "apd_exchange_create: bad argument");
    }
    int found = 0;
    X_TRY(hipGetDeviceCount(&found));
    apd_exchange *x = new apd_exchange();
    x->devices.assign(devices, devices + num_ranks);
    bool distinct = true;
    for (int i = 0; i < num_ranks; ++i) {
        if (devices[i] < 0 || devices[i] >= found) {
            delete x;
            return xfail(APD_ERR_INVALID, "apd_exchange_create: device %d requested, %d found", devices[i], found);
        }
        for (int j = 0; j < i; ++j) {
            distinct = distinct && devices[i] != devices[j];
        }
    }
    x->streams.resize(num_ranks, nullptr);
    x->done.resize(num_ranks, nullptr);
    for (int i = 0; i < num_ranks; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&x->streams[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming) != hipSuccess) {
            apd_exchange_destroy(x);
            return xfail(APD_ERR_HIP, "apd_exchange_create: cannot create a stream on device %d", devices[i]);
        }
    }
    // the list's period: its first `period` entries are distinct and the rest repeats them in order
    x->period = num_ranks;
    if (!distinct) {
        int first_repeat = num_ranks;
        for (int i = 1; i < num_ranks && first_repeat == num_ranks; ++i) {
            for (int j = 0; j < i; ++j) {
                if (devices[i] == devices[j]) {
                    first_repeat = i;
                    break;
                }
            }
        }
        bool periodic = num_ranks % first_repeat == 0;
        for (int i = first_repeat; i < num_ranks && periodic; ++i) {
            periodic = devices[i] == devices[i % first_repeat];
        }
        x->period = periodic ? first_repeat : 0;  // 0: no structure RCCL could use, direct copies
    }
    if (prefer_rccl && x->period > 0) {  // one communicator entry per device, all in this process
        rccl_initialise(x);
        if (x->rccl_state == kRcclFailed) {
            fprintf(stderr, "apd_exchange_create: RCCL is not available (%s): using direct copies\n", x->rccl_error.c_str());
        }
    }
    // direct reads between different devices go over xGMI: peer access, where the pair has it (also used by the copies inside a
    // device that follow an RCCL gather between leaders)
    x->peer_ok.assign((size_t)num_ranks * num_ranks, 0);
    for (int i = 0; i < num_ranks; ++i) {
        for (int j = 0; j < num_ranks; ++j) {
            if (devices[i] == devices[j]) {
                x->peer_ok[(size_t)i * num_ranks + j] = 1;
                continue;
            }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                hipSetDevice(devices[i]);
                hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) {
                    x->peer_ok[(size_t)i * num_ranks + j] = 1;
                }
                (void)hipGetLastError();
            }
        }
    }
    *out = x;
    return APD_OK;
}

int apd_exchange_setup_times(apd_exchange_t x, double *dlopen_ms, double *init_ms)
{
    if (!x) {
        return xfail(APD_ERR_INVALID, "apd_exchange_setup_times: null exchange");
    }
    if (dlopen_ms) {
        *dlopen_ms = g_rccl.loaded_ms();
    }
    if (init_ms) {
        *init_ms = x->init_ms;
    }
    return APD_OK;
}

const char *apd_exchange_backend(apd_exchange_t x)
{
    if (!x) {
        return "";
    }
    x->backend = x->rccl_state == kRcclReady ? "rccl" : "peer-copy";
    return x->backend.c_str();
}

int apd_exchange_counts(apd_exchange_t x, int *with_rccl, int *with_copies)
{
    if (!x) {
        return xfail(APD_ERR_INVALID, "apd_exchange_counts: null exchange");
    }
    if (with_rccl) {
        *with_rccl = x->exchanges_rccl;
    }
    if (with_copies) {
        *with_copies = x->exchanges_copy;
    }
    return APD_OK;
}

static int exchange_allgather(apd_exchange_t x, const void *const *send, void *const *recv, size_t bytes_per_rank, bool device_sync, int num_events,
                              void *const *hip_events);

// recv[r] of every rank r ends as send[0] | send[1] | ... | send[num_ranks - 1], `bytes_per_rank` each.  Waits for everything the
// devices were given before the call (the send buffers may have been written on any stream).
int apd_exchange_allgather(apd_exchange_t x, const void *const *send, void *const *recv, size_t bytes_per_rank)
{
    return exchange_allgather(x, send, recv, bytes_per_rank, true, 0, nullptr);
}

// The same without the device-wide synchronisation: the exchange's streams wait for `hip_events` (hipEvent_t, e.g. apd_export_event of
// every handle that wrote a block of a send buffer) and for nothing else, so kernels that other host threads have queued for the next
// pass keep running beside the exchange.  The caller guarantees that nobody reads or writes the recv buffers meanwhile.
int apd_exchange_allgather_after(apd_exchange_t x, const void *const *send, void *const *recv, size_t bytes_per_rank, int num_events,
                                 void *const *hip_events)
{
    if (num_events < 0 || (num_events > 0 && !hip_events)) {
        return xfail(APD_ERR_INVALID, "apd_exchange_allgather_after: bad event list");
    }
    return exchange_allgather(x, send, recv, bytes_per_rank, false, num_events, hip_events);
}

static int exchange_allgather(apd_exchange_t x, const void *const *send, void *const *recv, size_t bytes_per_rank, bool device_sync, int num_events,
                              void *const *hip_events)
{
    if (!x || !send || !recv) {
        return xfail(APD_ERR_INVALID, "apd_exchange_allgather: bad argument");
    }
    const int n = (int)x->devices.size();
    // The send buffers were written on other streams (the handles' own, the null stream of the pack copies); the exchange's
    // streams are non-blocking ones and would not wait for any of them.  Blocking form: everything the devices were
    // given before this call has finished before the first byte moves.  (Found the hard way: at 3100 x 2065 the planes of
    // views 1.. reached the fusion partly or not at all while the 1100 x 64 test passed.)  Event form: every stream of the
    // exchange waits for every writer's event -- an already completed event costs nothing, a pending one is honoured
    // (the contract used to rest on the callers having synchronised their streams).
    for (int r = 0; r < n && device_sync; ++r) {
        X_TRY(hipSetDevice(x->devices[r]));
        X_TRY(hipDeviceSynchronize());
    }
    for (int r = 0; r < n && num_events > 0; ++r) {
        X_TRY(hipSetDevice(x->devices[r]));
        for (int e = 0; e < num_events; ++e) {
            if (hip_events[e]) {
                X_TRY(hipStreamWaitEvent(x->streams[r], (hipEvent_t)hip_events[e], 0));
            }
        }
    }
    if (x->rccl_state == kRcclReady) {
        x->exchanges_rccl++;
        // Repetition l of the device list is ranks l * period .. (l + 1) * period - 1, one per device in communicator order: its
        // all-gather, run by the leaders (a send buffer only has to live on the leader's device), fills blocks l * period ..
        // of every leader's result -- rank order.  One group for all repetitions.
        const int period = x->period, reps = n / period;
        int rc = g_rccl.GroupStart();
        for (int l = 0; l < reps && rc == 0; ++l) {
            for (int d = 0; d < period && rc == 0; ++d) {
                rc = g_rccl.AllGather(send[l * period + d], (char *)recv[d] + (size_t)l * period * bytes_per_rank, bytes_per_rank, kNcclInt8,
                                      x->comms[d], x->streams[d]);
            }
        }
        const int rc_end = g_rccl.GroupEnd();
        rc = rc ? rc : rc_end;
        if (rc != 0) {
            return xfail(APD_ERR_HIP, "ncclAllGather failed: %s", g_rccl.GetErrorString(rc));
        }
        for (int r = period; r < n; ++r) {  // the other ranks of a device: a copy of their leader's result, ordered behind the gather
            X_TRY(hipSetDevice(x->devices[r]));
            X_TRY(hipMemcpyAsync(recv[r], recv[r % period], (size_t)n * bytes_per_rank, hipMemcpyDeviceToDevice, x->streams[r % period]));
        }
    } else {
        x->exchanges_copy++;
        // Every rank pulls every block on its own stream: one gather kernel over the blocks its device can read directly, a peer
        // copy for the others.  Ranks that share a device with an earlier rank copy that rank's finished result instead (one
        // contiguous read of local memory, no second trip over xGMI).
        const size_t words = bytes_per_rank / 16, tail = bytes_per_rank % 16;
        const bool kernel_ok = n <= apd::kGatherMaxRanks && (tail == 0 || n == 1);
        for (int dst = 0; dst < n; ++dst) {
            X_TRY(hipSetDevice(x->devices[dst]));
            int twin = -1;   // an earlier rank on the same device
            for (int q = 0; q < dst && twin < 0; ++q) {
                twin = x->devices[q] == x->devices[dst] ? q : -1;
            }
            apd::GatherArgs ga;
            memset(&ga, 0, sizeof(ga));
            if (twin >= 0) {
                X_TRY(hipStreamWaitEvent(x->streams[dst], x->done[twin], 0));
                const size_t all = (size_t)n * bytes_per_rank;
                if (all % 16 == 0 && all > 0 && ((uintptr_t)recv[dst] % 16) == 0 && ((uintptr_t)recv[twin] % 16) == 0) {
                    ga.src[0] = reinterpret_cast<const apd::word4 *>(recv[twin]);   // the whole result as one block
                    const unsigned gx = (unsigned)std::min<size_t>((all / 16 + 255) / 256, 2048);
                    hipLaunchKernelGGL(apd::k_gather_blocks, dim3(gx, 1), dim3(256), 0, x->streams[dst], ga, reinterpret_cast<apd::word4 *>(recv[dst]), all / 16, (size_t)0);
                    X_TRY(hipGetLastError());
                } else {
                    X_TRY(hipMemcpyAsync(recv[dst], recv[twin], all, hipMemcpyDeviceToDevice, x->streams[dst]));
                }
                X_TRY(hipEventRecord(x->done[dst], x->streams[dst]));
                continue;
            }
            int direct = 0;
            for (int src = 0; src < n; ++src) {
                char *to = (char *)recv[dst] + (size_t)src * bytes_per_rank;
                const bool aligned = ((uintptr_t)send[src] % 16) == 0 && ((uintptr_t)to % 16) == 0;
                if (kernel_ok && aligned && x->peer_ok[(size_t)dst * n + src] && bytes_per_rank > 0) {
                    ga.src[src] = reinterpret_cast<const apd::word4 *>(send[src]);
                    ++direct;
                } else if (x->devices[src] == x->devices[dst]) {
                    X_TRY(hipMemcpyAsync(to, send[src], bytes_per_rank, hipMemcpyDeviceToDevice, x->streams[dst]));
                } else {
                    X_TRY(hipMemcpyPeerAsync(to, x->devices[dst], send[src], x->devices[src], bytes_per_rank, x->streams[dst]));
                }
            }
            if (direct > 0) {
                const unsigned gx = (unsigned)std::min<size_t>((words + 255) / 256 + 1, 2048 / (size_t)std::max(1, std::min(direct, 8)) + 1);
                hipLaunchKernelGGL(apd::k_gather_blocks, dim3(gx, (unsigned)n), dim3(256), 0, x->streams[dst], ga, reinterpret_cast<apd::word4 *>(recv[dst]), words, tail);
                X_TRY(hipGetLastError());
            }
            X_TRY(hipEventRecord(x->done[dst], x->streams[dst]));
        }
    }
    for (int r = 0; r < n; ++r) {
        X_TRY(hipSetDevice(x->devices[r]));
        X_TRY(hipStreamSynchronize(x->streams[r]));
    }
    return APD_OK;
}

int apd_exchange_destroy(apd_exchange_t x)
{
    if (!x) {
        return APD_OK;
    }
    for (void *c : x->comms) {
        if (c) {
            g_rccl.CommDestroy(c);
        }
    }
    for (size_t i = 0; i < x->streams.size(); ++i) {
        hipSetDevice(x->devices[i]);
        if (x->streams[i]) {
            hipStreamDestroy(x->streams[i]);
        }
        if (i < x->done.size() && x->done[i]) {
            hipEventDestroy(x->done[i]);
        }
    }
    delete x;
    return APD_OK;
}

}  // extern "C"
