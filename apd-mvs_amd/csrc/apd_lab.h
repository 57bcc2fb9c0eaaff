// apd_lab.h -- measurement hooks of the diagnostic build (-DAPD_LAB_WIN_STATS; tools/win_stats.py, tools/weak_stats.py,
// tools/dup_stats.py rebuild the library with it).  In the product build every macro below expands to nothing: no counter,
// no atomic, no exported symbol.  Kept apart from the kernels so that what ships is readable without them.
#pragma once

#ifdef APD_LAB_WIN_STATS

namespace apd {
// [0] NCCs through the window, [1] global fast, [2] global slow, [3] wave-level NCC calls, [4] of those with both window and
// global lanes, [5] windows staged -- one copy per translation unit (no relocatable device code)
static __device__ unsigned long long g_k67w_stats[8];
// K9/K10: [0] lane NCCNew of the propagation phase, [1] wave-level ones, [2] / [3] the same for hypotheses 9..14, [4] lane
// sub-patches, [5] wave sub-patches
static __device__ unsigned long long g_weak_stats[8];
}  // namespace apd

#define APD_WIN_COUNT(i, n) atomicAdd(&g_k67w_stats[i], (unsigned long long)(n))
#define APD_WEAK_COUNT(i, n) atomicAdd(&g_weak_stats[i], (unsigned long long)(n))
#define APD_WEAK_COUNT_WAVE(i)                                                                     \
    do {                                                                                           \
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {       \
            APD_WEAK_COUNT(i, 1);                                                                  \
        }                                                                                          \
    } while (0)
// per wave-level NCC: which 36-sample body its lanes take
#define APD_LAB_NCC_STATS(in_window, fast_recip)                                                                             \
    do {                                                                                                                     \
        const unsigned long long m_all_ = __builtin_amdgcn_ballot_w64(true), m_in_ = __builtin_amdgcn_ballot_w64(in_window); \
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(m_all_)) {                                                            \
            APD_WIN_COUNT(3, 1);                                                                                             \
            APD_WIN_COUNT(4, (m_in_ != 0 && m_in_ != m_all_) ? 1 : 0);                                                       \
        }                                                                                                                    \
        APD_WIN_COUNT((in_window) ? 0 : ((fast_recip) ? 1 : 2), 1);                                                          \
    } while (0)
// sub-patch taps (index i * 3 + j: x offset i, y offset j; quad coordinates after the clamp): could the taps of one y offset share a
// 16-byte row segment of the column-pair image (same row, 0 <= column step <= 6)?  [6] rows of three taps where taps 0 and 1 or
// taps 1 and 2 could, [7] rows looked at
#define APD_LAB_SUBPATCH_ROWS(qx, qy)                                                                      \
    do {                                                                                                   \
        int fits_ = 0;                                                                                     \
        for (int j_ = 0; j_ < 3; ++j_) {                                                                   \
            const bool a_ = qy[j_] == qy[3 + j_] && qx[3 + j_] - qx[j_] >= 0 && qx[3 + j_] - qx[j_] <= 6;  \
            const bool b_ = qy[3 + j_] == qy[6 + j_] && qx[6 + j_] - qx[3 + j_] >= 0 && qx[6 + j_] - qx[3 + j_] <= 6; \
            fits_ += (a_ || b_) ? 1 : 0;                                                                   \
        }                                                                                                  \
        APD_WEAK_COUNT(6, fits_);                                                                          \
        APD_WEAK_COUNT(7, 3);                                                                              \
    } while (0)
#define APD_LAB_STATS_ACCESSOR_(name, symbol)                                                          \
    extern "C" int name(unsigned long long *out, int reset)                                            \
    {                                                                                                  \
        hipDeviceSynchronize();                                                                        \
        hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(symbol), sizeof(symbol));                   \
        if (e == hipSuccess && reset) {                                                                \
            unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                        \
            e = hipMemcpyToSymbol(HIP_SYMBOL(symbol), z, sizeof(z));                                   \
        }                                                                                              \
        return (int)e;                                                                                 \
    }
#define APD_WIN_STATS_ACCESSOR(name) APD_LAB_STATS_ACCESSOR_(name, apd::g_k67w_stats)
#define APD_LAB_WEAK_STATS_ACCESSOR                                    \
    APD_LAB_STATS_ACCESSOR_(apd_debug_weak_stats, apd::g_weak_stats)   \
    APD_WIN_STATS_ACCESSOR(apd_debug_win_stats_weak)

#else

#define APD_WIN_COUNT(i, n) ((void)0)
#define APD_WEAK_COUNT(i, n) ((void)0)
#define APD_WEAK_COUNT_WAVE(i) ((void)0)
#define APD_LAB_NCC_STATS(in_window, fast_recip) ((void)0)
#define APD_LAB_SUBPATCH_ROWS(qx, qy) ((void)0)
#define APD_WIN_STATS_ACCESSOR(name)
#define APD_LAB_WEAK_STATS_ACCESSOR

#endif
