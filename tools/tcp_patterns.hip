// tcp_patterns.hip -- what does one wave-level dword gather cost the L1 (TCP) of gfx950, as a function of how the 64 lanes'
// addresses are arranged?  (Question behind every software texture fetch of this repo: K9/K10 and the window-less first
// iteration of K6/K7 are bound by TCP_TOTAL_CACHE_ACCESSES, ~48 and ~41 per gather, and round 3's tap-cooperative gathers --
// nine lanes on the three source rows of one sub-patch instead of 64 lanes on 64 far-apart sub-patches -- did NOT lower the
// count.)  Every pattern keeps the wave inside one 16 KB window that stays L1/L2 resident, so time = tag pipeline, and runs
// under rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum for the access counts.
//
// Usage: tcp_patterns            (prints ns per gather per CU for every pattern)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIter = 512;

__device__ __forceinline__ uint32_t pattern_offset(int p, uint32_t lane)
{
    switch (p) {
    case 0: return lane * 4u;                                   // contiguous dwords: 256 B, 2 lines
    case 1: return lane * 4u + 2u;                              // the same, 2-byte aligned
    case 2: return lane * 8u;                                   // every other dword: 4 lines
    case 3: return lane * 16u;                                  // 8 lines
    case 4: return lane * 64u;                                  // two lanes per line
    case 5: return lane * 128u;                                 // one line per lane
    case 6: return (lane >> 2) * 128u + (lane & 3u) * 4u;       // quads contiguous, one line per quad
    case 7: return (lane >> 2) * 128u + (lane & 3u) * 16u;      // quads inside one line, not contiguous
    case 8: return (lane >> 2) * 128u;                          // quads on ONE dword, one line per quad
    case 9: return (lane >> 4) * 128u + (lane & 15u) * 4u;      // 16 contiguous lanes per line
    case 10: return 0u;                                         // every lane the same dword
    case 11: return (lane >> 1) * 128u + (lane & 1u) * 4u;      // pairs contiguous
    case 12: return ((lane * 37u) & 63u) * 4u;                  // contiguous 256 B, lanes shuffled
    case 13: return (lane / 9u) * 1536u + ((lane % 9u) / 3u) * 512u + ((lane % 9u) % 3u) * 10u;  // tap-cooperative sub-patch: 9 lanes on 3 rows, taps 10 B apart
    case 14: return (lane / 9u) * 1536u + ((lane % 9u) / 3u) * 512u + ((lane % 9u) % 3u) * 4u;   // the same with contiguous taps
    case 15: return (lane & 31u) * 4u + (lane >> 5) * 4096u;    // two half-waves, each 128 contiguous bytes
    default: return (lane * 2654435761u >> 18) & ~3u;           // 64 scattered dwords inside the 16 KB window
    }
}

__global__ __launch_bounds__(256) void gather(const unsigned char *__restrict__ buf, int p, uint32_t windows, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned char *base = buf + (size_t)((wave * 7919u) % windows) * 16384u;
    const uint32_t off = pattern_offset(p, lane);
    uint32_t acc = 0;
#pragma unroll 8
    for (int i = 0; i < kIter; ++i) {
        uint32_t v;
        __builtin_memcpy(&v, base + ((off + (uint32_t)i * 8192u) & 16383u), 4);  // alternate between the two halves of the window
        acc += v;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const uint32_t windows = 2048;  // 32 MB: stays in L2 / Infinity Cache; each wave's 16 KB window stays in its L1
    unsigned char *buf;
    uint32_t *out;
    const int blocks = 256 * 32;
    CHECK(hipMalloc(&buf, (size_t)windows * 16384u + 64));
    CHECK(hipMemset(buf, 1, (size_t)windows * 16384u + 64));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint32_t)));
    static const char *names[17] = {"lane*4 (contiguous)", "lane*4+2 (contiguous, 2-byte aligned)", "lane*8", "lane*16", "lane*64", "lane*128 (one line per lane)",
                                    "quads contiguous, one line per quad", "quads in one line, 16 B apart", "quads on one dword", "16 contiguous lanes per line",
                                    "all lanes one dword", "pairs contiguous, one line per pair", "contiguous 256 B, lanes shuffled",
                                    "coop sub-patch: 9 lanes, 3 rows, taps 10 B apart", "coop sub-patch: 9 lanes, 3 rows, taps contiguous",
                                    "two half-waves, 128 B each", "64 scattered dwords in 16 KB"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int p = 0; p < 17; ++p) {
        hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, p, windows, out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, p, windows, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double gathers = (double)blocks * 4 * kIter;
        printf("pattern %2d  %-52s %8.3f ms  %7.2f ns per wave-level gather per CU\n", p, names[p], ms, ms * 1e6 / (gathers / 256.0));
    }
    return 0;
}
