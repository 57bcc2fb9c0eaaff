// tcp_mix.hip -- what a wave-level gather costs the L1 (TCP) of gfx950 when its lanes go to 64 unrelated places, as a function of
// how many of them miss the L1 and are served by the XCD's L2.  (tcp_patterns.hip answers the question for gathers that hit.)
// K9/K10 and the window-less first iteration of K6/K7 are such gathers: 41 - 54 tag accesses per wave-level load of which a fifth to
// a seventh miss the L1 and 86 - 97 % of those hit the L2; doubling the resident waves does not change K9/K10's time
// (profiles/r03/ab_k910_split.txt), so it is a throughput that bounds them.  This program measures the two rates that
// throughput is made of and whether they add:
//   hit stream   every lane reads a dword of its own 128-byte line inside a region that stays L1 resident
//   miss stream  every lane reads a dword of its own line of a 2 MB per-XCD region, never the same line twice in a row (L1 miss, L2 hit)
//   mixes        h hit gathers per miss gather, and gathers in which only m of the 64 lanes miss
// Run under rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum / TCC_HIT_sum TCC_MISS_sum for the counts.
//
// Usage: tcp_mix      (prints ns per wave-level gather per CU and the implied CU clocks at 2.4 GHz for every case)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIter = 512;
constexpr uint32_t kXcdRegion = 2u << 20;      // bytes per XCD: half of its 4 MB L2
constexpr uint32_t kHotBytes = 1024u;          // per wave: 8 lines, eight lanes per line 16 bytes apart (one tag access per lane, tcp_patterns
                                               // pattern 7); 16 resident waves per CU keep 16 KB of the 32 KB L1

// hits_per_miss: 0 = misses only, -1 = hits only; miss_lanes: lanes of a miss gather that leave the hot region
__global__ __launch_bounds__(256) void mix(const unsigned char *__restrict__ cold, const unsigned char *__restrict__ hot, int hits_per_miss,
                                           int miss_lanes, uint32_t *out)
{
    __shared__ uint32_t pad[10 * 1024];  // 40 KB per workgroup: four workgroups = 16 waves per CU
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7u;
    const unsigned char *hot_base = hot + (size_t)wave * kHotBytes;
    const unsigned char *cold_base = cold + (size_t)xcd * kXcdRegion;
    const uint32_t hot_off = (lane >> 3) * 128u + (lane & 7u) * 16u;
    // a lane-private walk through the XCD's region: 64 lines per step, the steps of different waves interleave
    uint32_t line = (wave * 2654435761u >> 8) + lane * 97u;
    uint32_t acc = 0;
    if (threadIdx.x == 0 && hits_per_miss == 12345) {
        pad[0] = 1;  // keeps the array
    }
    for (int i = 0; i < kIter; ++i) {
        if (hits_per_miss >= 0) {
            line += 64u * 101u;
            const uint32_t off = (line % (kXcdRegion / 128u)) * 128u + ((line >> 3) & 31u) * 4u;
            const unsigned char *addr = lane < (uint32_t)miss_lanes ? cold_base + off : hot_base + hot_off;
            uint32_t v;
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc += v;
        }
        const int hits = hits_per_miss < 0 ? 1 : hits_per_miss;
        for (int h = 0; h < hits; ++h) {
            uint32_t v;
            const unsigned char *addr = hot_base + ((hot_off + (uint32_t)h * 4u) & (kHotBytes - 1u));
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + pad[threadIdx.x & 1023];
}

int main()
{
    const int blocks = 256 * 16;
    unsigned char *cold, *hot;
    uint32_t *out;
    CHECK(hipMalloc(&cold, (size_t)8 * kXcdRegion + 256));
    CHECK(hipMemset(cold, 1, (size_t)8 * kXcdRegion + 256));
    CHECK(hipMalloc(&hot, (size_t)blocks * 4 * kHotBytes + 256));
    CHECK(hipMemset(hot, 1, (size_t)blocks * 4 * kHotBytes + 256));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint32_t)));
    struct Case { int hits_per_miss, miss_lanes; const char *name; };
    const Case cases[] = {
        {-1, 0, "hits only (64 lanes on 8 lines, L1 resident)"},
        {0, 64, "misses only (64 lanes, 64 lines of the XCD's 2 MB region)"},
        {1, 64, "1 hit gather per miss gather"},
        {3, 64, "3 hit gathers per miss gather"},
        {7, 64, "7 hit gathers per miss gather"},
        {0, 32, "one gather: 32 lanes miss, 32 hit"},
        {0, 16, "one gather: 16 lanes miss, 48 hit"},
        {0, 8, "one gather: 8 lanes miss, 56 hit"},
        {0, 4, "one gather: 4 lanes miss, 60 hit"},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Case &c : cases) {
        hipLaunchKernelGGL(mix, dim3(blocks), dim3(256), 0, 0, cold, hot, c.hits_per_miss, c.miss_lanes, out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(mix, dim3(blocks), dim3(256), 0, 0, cold, hot, c.hits_per_miss, c.miss_lanes, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const int per_iter = c.hits_per_miss < 0 ? 1 : 1 + c.hits_per_miss;
        const double gathers = (double)blocks * 4 * kIter * per_iter;
        const double ns = ms * 1e6 / (gathers / 256.0);
        printf("%-62s %8.3f ms  %7.2f ns = %6.1f clocks per wave-level gather per CU\n", c.name, ms, ns, ns * 2.4);
    }
    return 0;
}
