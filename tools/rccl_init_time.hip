// rccl_init_time.hip -- where the seconds of RCCL's set-up go on this box: dlopen(librccl), ncclCommInitAll (first / second call in the
// process), with a kernel queue kept busy on the device or not.  Run under different environments (RCCL_MSCCL_ENABLE=0, ...):
//   tools/_build/rccl_init_time [devices=1] [busy=0|1]
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void spin(float *p, int n)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) {
        v = v * 1.000001f + 0.5f;
    }
    p[threadIdx.x] = v;
}

int main(int argc, char **argv)
{
    const int ndev = argc > 1 ? atoi(argv[1]) : 1;
    const bool busy = argc > 2 && atoi(argv[2]) != 0;
    int found = 0;
    hipGetDeviceCount(&found);
    float *buf = nullptr;
    hipSetDevice(0);
    hipMalloc(&buf, 4096);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, buf, 10);
    hipStreamSynchronize(st);
    std::atomic<bool> stop{false};
    std::atomic<long> launched{0};
    std::vector<double> gaps;
    std::thread worker;
    if (busy) {   // a host thread that launches short kernels back to back, like a lane of the scheduler: how long do its launches stall?
        worker = std::thread([&]() {
            hipSetDevice(0);
            double last = now_ms(), worst = 0;
            while (!stop.load()) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, buf, 20000);
                hipStreamSynchronize(st);
                const double t = now_ms();
                worst = t - last > worst ? t - last : worst;
                last = t;
                launched++;
            }
            gaps.push_back(worst);
        });
    }
    double t0 = now_ms();
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    const double t_dlopen = now_ms() - t0;
    if (!lib) {
        printf("librccl not found\n");
        return 1;
    }
    auto CommInitAll = (int (*)(void **, int, const int *))dlsym(lib, "ncclCommInitAll");
    auto CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    std::vector<int> devs;
    for (int i = 0; i < ndev; ++i) devs.push_back(i % (found > 0 ? found : 1));
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<void *> comms(ndev, nullptr);
        const long before = launched.load();
        t0 = now_ms();
        const int rc = CommInitAll(comms.data(), ndev, devs.data());
        const double t_init = now_ms() - t0;
        printf("%s ndev %d (found %d) busy %d: dlopen %.0f ms, ncclCommInitAll #%d %.0f ms (rc %d), kernels launched by the other thread meanwhile: %ld\n",
               getenv("TAG") ? getenv("TAG") : "", ndev, found, (int)busy, t_dlopen, rep + 1, t_init, rc, launched.load() - before);
        t0 = now_ms();
        for (void *c : comms) if (c) CommDestroy(c);
        printf("   ncclCommDestroy %.0f ms\n", now_ms() - t0);
    }
    stop.store(true);
    if (busy) {
        worker.join();
        printf("   longest gap between two kernel completions of the busy thread: %.0f ms (%ld kernels)\n", gaps[0], launched.load());
    }
    return 0;
}
