// valu_rates.hip -- gfx950 micro-measurements behind the sample loop of ncc_fixed:
//   (1) issue cost of v_fma_f32 / v_pk_fma_f32 / v_fma_mix_f32 / v_rcp_f32 / IEEE division / rcp+Newton, and the
//       issue rate of one wave against occupancy and instruction-level parallelism
//   (2) exhaustive checks (all 2^32 inputs) of the instruction identities the bit-exact contract uses:
//       rcp + one FMA Newton step == RN(1/z); v_fract_f32; v_cvt_flr_i32_f32
// Built by __graft_entry__.build() into tools/_build/valu_rates (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

constexpr int kChains = 8;
constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float *out, float seed)
{
    float a[kChains];
    v2f p[kChains];
    for (int i = 0; i < kChains; ++i) {
        a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
        p[i] = v2f{a[i], a[i] + 0.5f};
    }
    const float m = 0.999f + seed * 1e-6f, c = 1e-3f;
    const v2f pm = v2f{m, m}, pc = v2f{c, c};
#pragma unroll 1
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (MODE == 0) {
                a[i] = fmaf(a[i], m, c);
            } else if (MODE == 1) {
                p[i] = __builtin_elementwise_fma(p[i], pm, pc);
            } else if (MODE == 2) {
                a[i] = __builtin_amdgcn_rcpf(a[i]);
            } else if (MODE == 3) {
                a[i] = 1.0f / a[i];
            } else if (MODE == 4) {
                const float r = __builtin_amdgcn_rcpf(a[i]);
                const float e = fmaf(-a[i], r, 1.0f);
                a[i] = fmaf(e, r, r);
            } else if (MODE == 5) {
                a[i] = floorf(a[i] * m);
            } else if (MODE == 6) {
                p[i] = p[i] * pm;
            } else if (MODE == 7) {
                p[i] = p[i] + pc;
            } else if (MODE == 8) {
                a[i] = __builtin_amdgcn_fmed3f(a[i], -1.0f, m);
            } else if (MODE == 9) {
                a[i] = (float)(((uint32_t)__float_as_uint(a[i]) >> 8) & 0xFFu) + c;  // cvt_f32_ubyte1 + add
            } else if (MODE == 10) {
                a[i] = sqrtf(a[i]);
            } else if (MODE == 11) {
                a[i] = __builtin_amdgcn_fractf(a[i] * m);
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < kChains; ++i) {
        s += a[i] + p[i].x + p[i].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- exhaustive reciprocal check -----------------------------------------------------------------
__device__ __forceinline__ float recip_newton(float z)
{
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = fmaf(-z, r, 1.0f);
    return fmaf(e, r, r);
}

__global__ __launch_bounds__(256) void recip_check2(uint32_t base, unsigned long long *counts, uint32_t *examples)
{
    const uint32_t bits = base + blockIdx.x * 256u + threadIdx.x;
    const float z = __uint_as_float(bits);
    const float exact = 1.0f / z;
    const float fast = recip_newton(z);
    const float raw = __builtin_amdgcn_rcpf(z);
    const uint32_t ex = (bits >> 23) & 0xFFu;
    const bool same = (__float_as_uint(exact) == __float_as_uint(fast)) || (exact != exact && fast != fast);
    int cat;
    if (ex == 0 || ex == 255) {
        cat = 2;
    } else if (ex >= 27 && ex <= 227) {
        cat = 0;
    } else {
        cat = 1;
    }
    if (!same) {
        const unsigned long long k = atomicAdd(&counts[cat], 1ull);
        if (cat == 0 && k < 16) {
            examples[2 * k] = bits;
            examples[2 * k + 1] = __float_as_uint(fast);
        }
    }
    if (cat == 0 && __float_as_uint(raw) != __float_as_uint(exact)) {
        atomicAdd(&counts[3], 1ull);
    }
}

// ---- the NCC epilogue's square root and division without range scaling (csrc/apd_device.h: sqrt_rn_mid, div_rn_mid) -------------
__device__ __forceinline__ float sqrt_rn_mid(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = fmaf(-below, s, x), r_above = fmaf(-above, s, x);
    float r = (r_below <= 0.0f) ? below : s;
    r = (r_above > 0.0f) ? above : r;
    return r;
}

__device__ __forceinline__ float div_rn_mid(float a, float b)
{
    const float y = recip_newton(b);
    const float q0 = a * y;
    const float q1 = fmaf(fmaf(-b, q0, a), y, q0);
    return fmaf(fmaf(-b, q1, a), y, q1);
}

// every positive binary32 with biased exponent 31..223 ([0] mismatches against sqrtf) and every NaN ([1] results that are not NaN)
__global__ __launch_bounds__(256) void sqrt_check(uint32_t base, unsigned long long *counts, uint32_t *examples)
{
    const uint32_t bits = base + blockIdx.x * 256u + threadIdx.x;
    const float x = __uint_as_float(bits);
    const uint32_t ex = (bits >> 23) & 0xFFu;
    if (x != x) {
        const float r = sqrt_rn_mid(x);
        if (r == r) {
            atomicAdd(&counts[1], 1ull);
        }
        return;
    }
    if ((bits >> 31) != 0u || ex < 31u || ex > 223u) {
        return;
    }
    const float exact = sqrtf(x), mine = sqrt_rn_mid(x);
    if (__float_as_uint(exact) != __float_as_uint(mine)) {
        const unsigned long long k = atomicAdd(&counts[0], 1ull);
        if (k < 16) {
            examples[2 * k] = bits;
            examples[2 * k + 1] = __float_as_uint(mine);
        }
    }
}

// 2^32 pseudo-random pairs over the operand ranges of an NCC epilogue and beyond: b = 2^-20 .. 2^20 (the code has 1e-5 .. 1.7e4),
// a = +-2^-40 .. 2^30 with one pair in 64 a = +-0; [0] results whose bits differ from a / b (an exact zero may differ in sign: the
// caller only forms 1 - q), [2] pairs checked
__global__ __launch_bounds__(256) void div_check(uint32_t base, unsigned long long *counts, uint32_t *examples)
{
    uint32_t h = base + blockIdx.x * 256u + threadIdx.x;
    uint32_t w0 = h * 0x9E3779B1u;
    w0 ^= w0 >> 15;
    w0 *= 0x85EBCA77u;
    w0 ^= w0 >> 13;
    uint32_t w1 = (h ^ 0xA5A5A5A5u) * 0xC2B2AE3Du;
    w1 ^= w1 >> 16;
    w1 *= 0x27D4EB2Fu;
    w1 ^= w1 >> 15;
    const uint32_t eb = 107u + (w0 >> 23) % 41u;                   // biased exponent of b: 2^-20 .. 2^20
    const float b = __uint_as_float((eb << 23) | (w0 & 0x7FFFFFu));
    const uint32_t ea = 87u + (w1 >> 24) % 71u;                    // 2^-40 .. 2^30
    float a = __uint_as_float((w1 & 0x80000000u) | (ea << 23) | (w1 & 0x7FFFFFu));
    if ((h & 63u) == 0u) {
        a = __uint_as_float(w1 & 0x80000000u);
    }
    const float exact = a / b, mine = div_rn_mid(a, b);
    atomicAdd(&counts[2], 1ull);
    const bool same = __float_as_uint(exact) == __float_as_uint(mine) || (exact == 0.0f && mine == 0.0f);
    if (!same) {
        const unsigned long long k = atomicAdd(&counts[0], 1ull);
        if (k < 8) {
            examples[4 * k] = __float_as_uint(a);
            examples[4 * k + 1] = __float_as_uint(b);
            examples[4 * k + 2] = __float_as_uint(mine);
            examples[4 * k + 3] = __float_as_uint(exact);
        }
    }
}

__global__ __launch_bounds__(256) void fract_check(uint32_t base, unsigned long long *counts, uint32_t *examples)
{
    const uint32_t bits = base + blockIdx.x * 256u + threadIdx.x;
    const float x = __uint_as_float(bits);
    const float hw = __builtin_amdgcn_fractf(x);
    const float sub = x - floorf(x);
    const float emu = fminf(sub, 0x1.fffffep-1f);
    const bool finite = ((bits >> 23) & 0xFFu) != 255u;
    const bool same_sub = (__float_as_uint(hw) == __float_as_uint(sub)) || (hw != hw && sub != sub);
    if (finite && __float_as_uint(hw) != __float_as_uint(emu)) {
        const unsigned long long k = atomicAdd(&counts[0], 1ull);
        if (k < 16) {
            examples[2 * k] = bits;
            examples[2 * k + 1] = __float_as_uint(hw);
        }
    }
    if (!same_sub) {
        atomicAdd(&counts[1], 1ull);
    }
    if (!finite && hw == hw) {
        atomicAdd(&counts[2], 1ull);  // Inf / NaN input must give NaN
    }
}


// ---- issue rate of one wave vs occupancy and ILP, and the shader clock ------------------------------
template <int ILP>
__global__ __launch_bounds__(256) void ilp_kernel(float *out, float seed, long long *clocks)
{
    float a[ILP];
    for (int i = 0; i < ILP; ++i) {
        a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    }
    const float m = 0.999f + seed * 1e-6f, c = 1e-3f;
    const long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < kIters * 8 / ILP; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                a[i] = fmaf(a[i], m, c);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < ILP; ++i) {
        s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}

template <int ILP>
static int run_ilp(int waves_per_simd, float *dout, long long *dclk)
{
    const int blocks = 256 * waves_per_simd, threads = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(ilp_kernel<ILP>, dim3(blocks), dim3(threads), 0, 0, dout, 1.5f, dclk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(ilp_kernel<ILP>, dim3(blocks), dim3(threads), 0, 0, dout, 1.5f, dclk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long hclk[2];
    CHECK(hipMemcpy(hclk, dclk, sizeof(hclk), hipMemcpyDeviceToHost));
    const double ops_per_wave = (double)(kIters * 8 / ILP) * 4 * ILP;
    printf("fma ILP=%d waves/SIMD=%d: %7.3f ms  clock64 ticks/op/wave %.2f  wall_clock64 ticks %lld vs clock64 %lld  -> ns per op per SIMD %.3f\n", ILP,
           waves_per_simd, ms, (double)hclk[0] / ops_per_wave, hclk[1], hclk[0], ms * 1e6 / (ops_per_wave * waves_per_simd));
    return 0;
}

// ---- v_cvt_flr_i32_f32 / v_fract_f32 special values, v_fma_mix_f32 rate ---------------------------------
__device__ __forceinline__ int cvt_flr(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__global__ __launch_bounds__(256) void cvt_flr_check(uint32_t base, unsigned long long *counts, uint32_t *examples)
{
    const uint32_t bits = base + blockIdx.x * 256u + threadIdx.x;
    const float x = __uint_as_float(bits);
    const int hw = cvt_flr(x);
    int emu;
    if (x != x) {
        emu = 0;
    } else if (x >= 2147483648.0f) {
        emu = 0x7fffffff;
    } else if (x < -2147483648.0f) {
        emu = (int)0x80000000;
    } else {
        emu = (int)floorf(x);
    }
    if (hw != emu) {
        const unsigned long long k = atomicAdd(&counts[0], 1ull);
        if (k < 16) {
            examples[2 * k] = bits;
            examples[2 * k + 1] = (uint32_t)hw;
        }
        if (x == x) {
            atomicAdd(&counts[1], 1ull);
        }
    }
}

__global__ void special_values(float *out)
{
    const float inf = __builtin_inff();
    out[0] = __builtin_amdgcn_fractf(inf);
    out[1] = __builtin_amdgcn_fractf(-inf);
    out[2] = __builtin_amdgcn_fractf(__builtin_nanf(""));
    out[3] = __builtin_amdgcn_fractf(-1e-10f);
    out[4] = __builtin_amdgcn_fractf(-0.0f);
    out[5] = __builtin_amdgcn_fractf(3.0e38f);
}

__global__ __launch_bounds__(256) void mix_kernel(float *out, float seed, const uint32_t *h)
{
    float a[kChains];
    for (int i = 0; i < kChains; ++i) {
        a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    }
    const uint32_t packed = h[threadIdx.x & 63];  // two halfs
#pragma unroll 1
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            // a = a * half_hi + half_lo
            asm volatile("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(a[i]) : "v"(a[i]), "v"(packed));
        }
    }
    float s = 0.0f;
    for (int i = 0; i < kChains; ++i) {
        s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static int run_rate(const char *name, int ops_per_iter_per_chain, float *dout)
{
    const int blocks = 256 * 8, threads = 256;  // 8 workgroups (32 waves) per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, dout, 1.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, dout, 1.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * threads / 64.0;
    const double wave_insts = waves * kIters * kChains * ops_per_iter_per_chain;
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 256 * 4;
    printf("%-34s %8.3f ms  %6.2f SIMD-cycles per wave-op (at 2.4 GHz)\n", name, ms, simd_cycles / wave_insts);
    return 0;
}

static int run_checks()
{
    unsigned long long *dcounts, hcounts[4];
    uint32_t *dex, hex[32];
    CHECK(hipMalloc(&dcounts, 4 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&dex, 32 * sizeof(uint32_t)));
    // (1) reciprocal
    CHECK(hipMemset(dcounts, 0, 4 * sizeof(unsigned long long)));
    CHECK(hipMemset(dex, 0, 32 * sizeof(uint32_t)));
    for (uint32_t hi = 0; hi < 256; ++hi) {
        hipLaunchKernelGGL(recip_check2, dim3(1u << 16), dim3(256), 0, 0, hi << 24, dcounts, dex);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hcounts, dcounts, sizeof(hcounts), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, dex, sizeof(hex), hipMemcpyDeviceToHost));
    printf("recip rcp+Newton vs IEEE 1/z over all 2^32 inputs: mismatches mid-range(exp 27..227)=%llu, outer normal=%llu, "
           "denormal/zero/inf/nan=%llu; raw v_rcp_f32 != exact in mid-range: %llu\n",
           hcounts[0], hcounts[1], hcounts[2], hcounts[3]);
    printf("CHECK_recip_midrange_mismatches=%llu\n", hcounts[0]);
    for (int i = 0; i < 16 && (unsigned long long)i < hcounts[0]; ++i) {
        float z;
        memcpy(&z, &hex[2 * i], 4);
        printf("   z=%08x (%g) fast=%08x exact=%a\n", hex[2 * i], z, hex[2 * i + 1], 1.0f / z);
    }
    // (2) fract
    CHECK(hipMemset(dcounts, 0, 4 * sizeof(unsigned long long)));
    CHECK(hipMemset(dex, 0, 32 * sizeof(uint32_t)));
    for (uint32_t hi = 0; hi < 256; ++hi) {
        hipLaunchKernelGGL(fract_check, dim3(1u << 16), dim3(256), 0, 0, hi << 24, dcounts, dex);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hcounts, dcounts, sizeof(hcounts), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, dex, sizeof(hex), hipMemcpyDeviceToHost));
    printf("v_fract_f32 vs min(x-floor(x), 0x1.fffffep-1) on finite x: mismatches=%llu ; vs plain x-floor(x) (all x): %llu ; "
           "Inf/NaN inputs not giving NaN: %llu\n", hcounts[0], hcounts[1], hcounts[2]);
    printf("CHECK_fract_finite_mismatches=%llu\nCHECK_fract_nonfinite_not_nan=%llu\n", hcounts[0], hcounts[2]);
    for (int i = 0; i < 16 && (unsigned long long)i < hcounts[0]; ++i) {
        float z;
        memcpy(&z, &hex[2 * i], 4);
        printf("   x=%08x (%g) hw=%08x\n", hex[2 * i], z, hex[2 * i + 1]);
    }
    // (3) cvt_flr
    CHECK(hipMemset(dcounts, 0, 4 * sizeof(unsigned long long)));
    CHECK(hipMemset(dex, 0, 32 * sizeof(uint32_t)));
    for (uint32_t hi = 0; hi < 256; ++hi) {
        hipLaunchKernelGGL(cvt_flr_check, dim3(1u << 16), dim3(256), 0, 0, hi << 24, dcounts, dex);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hcounts, dcounts, sizeof(hcounts), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, dex, sizeof(hex), hipMemcpyDeviceToHost));
    printf("v_cvt_flr_i32_f32 vs saturating (int)floor(x) with NaN -> 0: mismatches=%llu, of which x is not NaN: %llu\n", hcounts[0],
           hcounts[1]);
    printf("CHECK_cvt_flr_non_nan_mismatches=%llu\n", hcounts[1]);
    // (4) square root and division of the NCC epilogue
    CHECK(hipMemset(dcounts, 0, 4 * sizeof(unsigned long long)));
    CHECK(hipMemset(dex, 0, 32 * sizeof(uint32_t)));
    for (uint32_t hi = 0; hi < 256; ++hi) {
        hipLaunchKernelGGL(sqrt_check, dim3(1u << 16), dim3(256), 0, 0, hi << 24, dcounts, dex);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hcounts, dcounts, sizeof(hcounts), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, dex, sizeof(hex), hipMemcpyDeviceToHost));
    printf("sqrt_rn_mid vs sqrtf over every positive binary32 with biased exponent 31..223: mismatches=%llu; NaN inputs not giving NaN: %llu\n", hcounts[0], hcounts[1]);
    printf("CHECK_sqrt_midrange_mismatches=%llu\nCHECK_sqrt_nan_not_nan=%llu\n", hcounts[0], hcounts[1]);
    for (int i = 0; i < 8 && (unsigned long long)i < hcounts[0]; ++i) {
        printf("   x=%08x mine=%08x\n", hex[2 * i], hex[2 * i + 1]);
    }
    CHECK(hipMemset(dcounts, 0, 4 * sizeof(unsigned long long)));
    CHECK(hipMemset(dex, 0, 32 * sizeof(uint32_t)));
    for (uint32_t hi = 0; hi < 256; ++hi) {
        hipLaunchKernelGGL(div_check, dim3(1u << 16), dim3(256), 0, 0, hi << 24, dcounts, dex);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hcounts, dcounts, sizeof(hcounts), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, dex, sizeof(hex), hipMemcpyDeviceToHost));
    printf("div_rn_mid vs a / b over %llu pseudo-random pairs (b = 2^-20..2^20, a = +-2^-40..2^30 and +-0): mismatches=%llu\n", hcounts[2], hcounts[0]);
    printf("CHECK_div_midrange_mismatches=%llu\nCHECK_div_pairs=%llu\n", hcounts[0], hcounts[2]);
    for (int i = 0; i < 8 && (unsigned long long)i < hcounts[0]; ++i) {
        printf("   a=%08x b=%08x mine=%08x exact=%08x\n", hex[4 * i], hex[4 * i + 1], hex[4 * i + 2], hex[4 * i + 3]);
    }
    float *dsv, hsv[6];
    CHECK(hipMalloc(&dsv, sizeof(hsv)));
    hipLaunchKernelGGL(special_values, dim3(1), dim3(1), 0, 0, dsv);
    CHECK(hipMemcpy(hsv, dsv, sizeof(hsv), hipMemcpyDeviceToHost));
    uint32_t u[6];
    memcpy(u, hsv, sizeof(u));
    printf("v_fract_f32: +inf -> %08x, -inf -> %08x, nan -> %08x, -1e-10 -> %08x, -0 -> %08x, 3e38 -> %08x\n", u[0], u[1], u[2], u[3], u[4],
           u[5]);
    return 0;
}

static int run_rates()
{
    float *dout;
    CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
    run_rate<0>("v_fma_f32", 1, dout);
    run_rate<1>("v_pk_fma_f32 (per pk op)", 1, dout);
    run_rate<6>("v_pk_mul_f32 (per pk op)", 1, dout);
    run_rate<7>("v_pk_add_f32 (per pk op)", 1, dout);
    run_rate<2>("v_rcp_f32", 1, dout);
    run_rate<10>("sqrtf (IEEE)", 1, dout);
    run_rate<3>("1.0f/x (IEEE)", 1, dout);
    run_rate<4>("rcp + 2 fma Newton (whole)", 1, dout);
    run_rate<5>("mul + floor (2 ops)", 2, dout);
    run_rate<8>("v_med3_f32", 1, dout);
    run_rate<9>("cvt_f32_ubyte1 + add (2-3 ops)", 2, dout);
    run_rate<11>("mul + v_fract_f32 (2 ops)", 2, dout);
    {
        uint32_t *dh, hh[64];
        for (int i = 0; i < 64; ++i) {
            hh[i] = 0x3c003800u;  // hi = 1.0h, lo = 0.5h
        }
        CHECK(hipMalloc(&dh, sizeof(hh)));
        CHECK(hipMemcpy(dh, hh, sizeof(hh), hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(mix_kernel, dim3(2048), dim3(256), 0, 0, dout, 1.5f, dh);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(mix_kernel, dim3(2048), dim3(256), 0, 0, dout, 1.5f, dh);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.3f ms  %6.2f SIMD-cycles per wave-op (at 2.4 GHz)\n", "v_fma_mix_f32 (f32*f16+f16)", ms,
               ms * 1e-3 * 2.4e9 * 1024 / (2048.0 * 4 * kIters * kChains));
    }
    long long *dclk;
    CHECK(hipMalloc(&dclk, 2 * sizeof(long long)));
    for (int w = 1; w <= 8; w = (w < 4) ? w + 1 : w * 2) {
        run_ilp<1>(w, dout, dclk);
    }
    for (int w = 1; w <= 8; w = (w < 4) ? w + 1 : w * 2) {
        run_ilp<2>(w, dout, dclk);
    }
    for (int w = 1; w <= 8; w = (w < 4) ? w + 1 : w * 2) {
        run_ilp<8>(w, dout, dclk);
    }
    return 0;
}

// usage: valu_rates            rates, then the exhaustive checks
//        valu_rates --check    exhaustive checks only (machine-readable CHECK_* lines; tests/test_gpu_edge_cases.py)
//        valu_rates --rates    rates only
int main(int argc, char **argv)
{
    const bool only_check = argc > 1 && !strcmp(argv[1], "--check");
    const bool only_rates = argc > 1 && !strcmp(argv[1], "--rates");
    if (!only_check && run_rates()) {
        return 1;
    }
    if (!only_rates && run_checks()) {
        return 1;
    }
    return 0;
}
