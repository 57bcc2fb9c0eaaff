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
constexpr int kIter = 256;

// quad-level patterns: quad q of the wave sits in its own 128-byte line (q * 256 bytes apart), the four lanes of a quad are
// `stride` bytes apart starting `shift` bytes into the line; order 0 ascending, 1 descending, 2 shuffled (0, 2, 1, 3)
__device__ __forceinline__ uint32_t quad_pattern(uint32_t lane, uint32_t stride, uint32_t shift, int order)
{
    uint32_t k = lane & 3u;
    if (order == 1) {
        k = 3u - k;
    } else if (order == 2) {
        k = (k == 1u) ? 2u : (k == 2u ? 1u : k);
    }
    return (lane >> 2) * 256u + shift + k * stride;
}

__device__ __forceinline__ uint32_t pattern_offset(int p, uint32_t lane)
{
    switch (p) {
    case 0: return lane * 4u;                                   // contiguous dwords: 256 B, 2 lines
    case 1: return lane * 4u + 2u;                              // the same, 2-byte aligned
    case 2: return lane * 8u;                                   // every other dword: 4 lines
    case 3: return lane * 16u;                                  // 8 lines
    case 4: return lane * 64u;                                  // two lanes per line
    case 5: return lane * 128u;                                 // one line per lane
    case 6: return quad_pattern(lane, 4u, 0u, 0);               // quads contiguous, one line per quad
    case 7: return quad_pattern(lane, 16u, 0u, 0);              // quads in one line, 16 B apart
    case 8: return quad_pattern(lane, 0u, 0u, 0);               // quads on ONE dword
    case 9: return (lane >> 4) * 256u + (lane & 15u) * 4u;      // 16 contiguous lanes per line
    case 10: return 0u;                                         // every lane the same dword
    case 11: return (lane >> 1) * 128u + (lane & 1u) * 4u;      // pairs contiguous, one line per pair
    case 12: return ((lane * 37u) & 63u) * 4u;                  // contiguous 256 B, lanes shuffled
    case 13: return (lane / 9u) * 1536u + ((lane % 9u) / 3u) * 512u + ((lane % 9u) % 3u) * 10u;  // tap-cooperative sub-patch: 9 lanes on 3 rows, taps 10 B apart
    case 14: return (lane / 9u) * 1536u + ((lane % 9u) / 3u) * 512u + ((lane % 9u) % 3u) * 4u;   // the same with contiguous taps
    case 15: return (lane & 31u) * 4u + (lane >> 5) * 4096u;    // two half-waves, each 128 contiguous bytes
    case 16: return (lane * 2654435761u >> 18) & ~3u;           // 64 scattered dwords inside the 16 KB window
    case 17: return quad_pattern(lane, 2u, 0u, 0);              // quad lanes 2 B apart (overlapping dwords)
    case 18: return quad_pattern(lane, 6u, 0u, 0);              // 6 B apart
    case 19: return quad_pattern(lane, 8u, 0u, 0);              // 8 B apart: 28 B span
    case 20: return quad_pattern(lane, 10u, 0u, 0);             // 10 B apart (sub-patch taps): 34 B span
    case 21: return quad_pattern(lane, 12u, 0u, 0);
    case 22: return quad_pattern(lane, 4u, 2u, 0);              // contiguous, 2-byte aligned
    case 23: return quad_pattern(lane, 4u, 24u, 0);             // contiguous, straddles a 32-byte boundary
    case 24: return quad_pattern(lane, 4u, 56u, 0);             // contiguous, straddles a 64-byte boundary
    case 25: return quad_pattern(lane, 4u, 120u, 0);            // contiguous, straddles a 128-byte line
    case 26: return quad_pattern(lane, 4u, 0u, 1);              // contiguous, descending
    case 27: return quad_pattern(lane, 4u, 0u, 2);              // contiguous, shuffled
    case 28: return quad_pattern(lane, 8u, 0u, 2);              // 8 B apart, shuffled
    case 29: return (lane >> 3) * 256u + (lane & 7u) * 4u;      // 8 contiguous lanes per line
    case 30: return (lane >> 3) * 256u + (lane & 7u) * 4u + 2u; // the same, 2-byte aligned
    case 31: return (lane >> 3) * 256u + ((lane & 7u) < 6u ? (lane & 7u) : 5u) * 4u + 2u;  // six taps + two repeats of the last (a 6-tap patch row in 8 lanes)
    case 32: return (lane >> 3) * 256u + ((lane & 7u) < 6u ? (lane & 7u) * 4u + ((lane & 7u) >= 3u ? 2u : 0u) : 22u);  // six taps with one 3-px step in the middle
    default: return (lane >> 2) * 256u + ((lane & 3u) < 3u ? (lane & 3u) * 10u : 20u);  // 33: three sub-patch taps 10 B apart + a repeat of the third
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
        const unsigned char *addr = base + off;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr) : "memory");  // opaque: one gather per iteration
        acc += v;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// The same question for wider loads: does a lane that reads an ALIGNED 8- or 16-byte block pay one access or one per dword?
// (If one, a software texture fetch that needs several neighbouring texels of a row can take them with one load.)
// Patterns: 0 every lane its own 128-byte line, 1 quads of lanes on four consecutive blocks, 2 lane * width (contiguous).
template <int DWORDS>
__global__ __launch_bounds__(256) void gather_wide(const unsigned char *__restrict__ buf, int p, uint32_t windows, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned char *base = buf + (size_t)((wave * 7919u) % windows) * 16384u;
    const uint32_t w = 4u * DWORDS;
    // patterns 3..7 (round 4): one line per lane like pattern 0, the block shifted off its natural alignment -- can a software texture
    // fetch take a row segment that starts wherever its first tap lies?  3: +4 B (dword aligned), 4: +2 B, 5: +60 B (dword aligned,
    // crosses a 64-byte boundary), 6: +120 B (crosses the 128-byte line), 7: +6 B
    // patterns 8..11: 64 scattered blocks inside the wave's L1-resident 16 KB window (the regime of K9/K10's sub-patch taps: hits),
    // 16-byte aligned, at +4, +8 and +12 bytes
    static const uint32_t shift[8] = {0u, 0u, 0u, 4u, 2u, 60u, 120u, 6u};
    const uint32_t off = p >= 8 ? (((lane * 2654435761u >> 18) & 0x3FF0u) % 16000u) + 4u * (uint32_t)(p - 8)
                                : (p == 0 || p >= 3 ? lane * 128u + shift[p & 7] : (p == 1 ? (lane >> 2) * 256u + (lane & 3u) * w : lane * w));
    uint32_t acc = 0;
#pragma unroll 8
    for (int i = 0; i < kIter; ++i) {
        const unsigned char *addr = base + off;
        if constexpr (DWORDS == 2) {
            uint64_t v;
            asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc += (uint32_t)v + (uint32_t)(v >> 32);
        } else {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 v;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc += v.x + v.y + v.z + v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

constexpr int kPatterns = 34;
int main()
{
    const uint32_t windows = 2048;  // 32 MB: stays in L2 / Infinity Cache; each wave's 16 KB window stays in its L1
    unsigned char *buf;
    uint32_t *out;
    const int blocks = 256 * 32;
    CHECK(hipMalloc(&buf, (size_t)windows * 16384u + 64));
    CHECK(hipMemset(buf, 1, (size_t)windows * 16384u + 64));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint32_t)));
    static const char *names[kPatterns] = {"lane*4 (contiguous)", "lane*4+2 (contiguous, 2-byte aligned)", "lane*8", "lane*16", "lane*64", "lane*128 (one line per lane)",
                                    "quads contiguous, one line per quad", "quads in one line, 16 B apart", "quads on one dword", "16 contiguous lanes per line",
                                    "all lanes one dword", "pairs contiguous, one line per pair", "contiguous 256 B, lanes shuffled",
                                    "coop sub-patch: 9 lanes, 3 rows, taps 10 B apart", "coop sub-patch: 9 lanes, 3 rows, taps contiguous",
                                    "two half-waves, 128 B each", "64 scattered dwords in 16 KB",
                                    "quad 2 B apart", "quad 6 B apart", "quad 8 B apart", "quad 10 B apart", "quad 12 B apart", "quad contiguous, 2-byte aligned",
                                    "quad contiguous across a 32 B boundary", "quad contiguous across a 64 B boundary", "quad contiguous across a 128 B line",
                                    "quad contiguous, descending", "quad contiguous, shuffled", "quad 8 B apart, shuffled", "8 contiguous lanes per line",
                                    "8 contiguous lanes per line, 2-byte aligned", "6 taps + 2 repeats in 8 lanes, 2-byte aligned", "6 taps with a 3-px step in 8 lanes",
                                    "3 taps 10 B apart + 1 repeat per quad"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int p = 0; p < kPatterns; ++p) {
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
    static const char *wide_names[12] = {"one 128-byte line per lane", "quads of lanes on 4 consecutive blocks, one line pair per quad", "lane * width (contiguous)",
                                        "one line per lane, block at +4 B", "one line per lane, block at +2 B", "one line per lane, block at +60 B (crosses 64 B)",
                                        "one line per lane, block at +120 B (crosses the line)", "one line per lane, block at +6 B",
                                        "64 scattered blocks in 16 KB (L1 hits), 16-byte aligned", "64 scattered blocks in 16 KB, +4 B", "64 scattered blocks in 16 KB, +8 B",
                                        "64 scattered blocks in 16 KB, +12 B"};
    for (int width = 2; width <= 4; width += 2) {
        for (int p = 0; p < 12; ++p) {
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) {
                    CHECK(hipEventRecord(e0));
                }
                if (width == 2) {
                    hipLaunchKernelGGL(gather_wide<2>, dim3(blocks), dim3(256), 0, 0, buf, p, windows, out);
                } else {
                    hipLaunchKernelGGL(gather_wide<4>, dim3(blocks), dim3(256), 0, 0, buf, p, windows, out);
                }
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double gathers = (double)blocks * 4 * kIter;
            printf("dwordx%d pattern %d  %-60s %8.3f ms  %7.2f ns per wave-level gather per CU\n", width, p, wide_names[p], ms, ms * 1e6 / (gathers / 256.0));
        }
    }
    return 0;
}
