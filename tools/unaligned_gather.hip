// unaligned_gather.hip -- does a dword gather at a 2-byte-aligned address cost more than at a 4-byte-aligned one on gfx950?
// (question behind a 2-byte "column pair" layout of the source images: entry x = {I(x,y), I(x,y+1)}, so the dword at byte
// 2x holds the four taps of a bilinear fetch at half the footprint of the 4-byte quads)
// Every lane gathers `kIter` dwords at pseudo-random positions of a 100 MB buffer; positions are even multiples of 2 bytes
// (aligned run), arbitrary multiples of 2 (half of them misaligned) or, third run, the same random positions but 64 lanes
// within one 2 KB neighbourhood (cache-friendly, to see the L1 path alone).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIter = 256;
template <int MODE>
__global__ __launch_bounds__(256) void gather(const unsigned char *__restrict__ buf, uint32_t entries, uint32_t *out)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t wave_base = ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 7919u * 4096u) % (entries - 4096u);
    uint32_t acc = 0;
#pragma unroll 4
    for (int i = 0; i < kIter; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t e;
        if (MODE == 2 || MODE == 3) {
            e = wave_base + ((s >> 8) & 1023u);  // 64 lanes inside a 2 KB neighbourhood
            if (MODE == 2) e &= ~1u;
        } else {
            e = (s >> 4) % entries;
            if (MODE == 0) e &= ~1u;             // 4-byte aligned
        }
        uint32_t v;
        __builtin_memcpy(&v, buf + 2u * (size_t)e, 4);
        acc += v;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> static int run(const char *name, const unsigned char *buf, uint32_t entries, uint32_t *out)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 64;
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, buf, entries, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, buf, entries, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gathers = (double)blocks * 4 * kIter;
    printf("%-44s %8.3f ms  %7.2f ns per wave-level gather per CU\n", name, ms, ms * 1e6 / (gathers / 256.0));
    return 0;
}
int main()
{
    const uint32_t entries = 50u * 1000u * 1000u;  // 100 MB of 2-byte entries
    unsigned char *buf;
    uint32_t *out;
    CHECK(hipMalloc(&buf, 2 * (size_t)entries + 16));
    CHECK(hipMemset(buf, 1, 2 * (size_t)entries + 16));
    CHECK(hipMalloc(&out, 256 * 64 * 256 * sizeof(uint32_t)));
    int rc = 0;
    rc |= run<0>("random, 4-byte aligned", buf, entries, out);
    rc |= run<1>("random, 2-byte aligned (half misaligned)", buf, entries, out);
    rc |= run<2>("2 KB neighbourhood per wave, 4-byte aligned", buf, entries, out);
    rc |= run<3>("2 KB neighbourhood per wave, 2-byte aligned", buf, entries, out);
    return rc;
}
