// valu_issue.hip -- how fast does a gfx950 SIMD issue vector-ALU instructions?
//
// Question behind it: tools/valu_rates.hip printed 4.07 "SIMD-cycles" for v_fma_f32 and 5.29 for
// v_pk_fma_f32 from loop bodies of eight instructions -- the loop's own s_add/s_cmp/s_cbranch was in the number -- while
// its ILP table implied one v_fma_f32 per ~1 ns per SIMD.  Here every body is a straight line of 256 instructions
// (inline asm, so the compiler neither packs nor reorders nor deletes anything), the shader clock is measured inside the
// kernel (s_memtime against s_memrealtime), and each kind runs with 1, 2, 4 and 8 waves per SIMD, on independent
// accumulators (distance 16) and on one dependent chain.
//
// Output per line: kind, waves per SIMD, ns per wave-instruction per SIMD, shader clock, SIMD cycles per wave-instruction.
// The plateau of "cycles per wave-instruction" over the occupancies is the issue cost that a roofline has to use.
//   build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/_build/valu_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Kind {
    FMA, FMA_DEP, PK_FMA, PK_FMA_DEP, MUL, PK_MUL, ADD, PK_ADD, FRACT, CVT_FLR, MED3_I32, MAD_I24, LSHL_ADD, FMA_MIX, RCP, CVT_UBYTE,
    FMA_AND_PK_FMA, LDS_READ2, SUB, MAX_F32, MOV, ADD_U32, AND_B32, CNDMASK, CMP_LT, FLOOR, FMAC, LDS_READ_B32, MUL_LIT, MUL_LO_U32, MUL_U32_U24, MAD_U32_U24, MUL_HI_U32, LSHRREV, LSHLREV, OR_B32, OR3, AND_OR, BFE_U32,
    ADD_LSHL, CVT_U32_F32, CVT_F32_U32, PERM, ADD3, SUB_U32, XAD, KIND_COUNT
};
static const char *kNames[KIND_COUNT] = {
    "v_fma_f32 (16 independent)", "v_fma_f32 (one dependent chain)", "v_pk_fma_f32 (16 independent)", "v_pk_fma_f32 (dependent chain)",
    "v_mul_f32", "v_pk_mul_f32", "v_add_f32", "v_pk_add_f32", "v_fract_f32", "v_cvt_flr_i32_f32", "v_med3_i32", "v_mad_i32_i24",
    "v_lshl_add_u32", "v_fma_mix_f32", "v_rcp_f32", "v_cvt_f32_ubyte1", "v_fma_f32 + v_pk_fma_f32 alternating (per instruction)",
    "ds_read2st64_b32 (8 B per lane, conflict-free)", "v_sub_f32", "v_max_f32", "v_mov_b32", "v_add_u32", "v_and_b32", "v_cndmask_b32",
    "v_cmp_lt_f32 (to vcc)", "v_floor_f32", "v_fmac_f32", "ds_read_b32 (4 B per lane, conflict-free)", "v_mul_f32 by a literal", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32", "v_lshrrev_b32", "v_lshlrev_b32",
    "v_or_b32", "v_or3_b32", "v_and_or_b32", "v_bfe_u32", "v_add_lshl_u32", "v_cvt_u32_f32", "v_cvt_f32_u32", "v_perm_b32", "v_add3_u32", "v_sub_u32", "v_xad_u32"};

constexpr int kBody = 256;   // instructions per loop iteration
constexpr int kLoops = 512;  // iterations

template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(float *out, float seed, long long *clocks)
{
    __shared__ uint32_t lds[64 * 72];
    float a[16];
    v2f p[16];
    int n[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
        p[i] = v2f{a[i], a[i] + 0.5f};
        n[i] = (int)threadIdx.x + i;
    }
    for (int i = threadIdx.x; i < 64 * 72; i += 256) {
        lds[i] = i;
    }
    __syncthreads();
    const float m = 0.999f + seed * 1e-6f, c = 1e-3f;
    const v2f pm = v2f{m, m}, pc = v2f{c, c};
    const int lo = -1, hi = 4000, pitch = 24;
    const uint32_t half2 = 0x3c003800u;
    const uint32_t lds_addr = (uint32_t)(size_t)(&lds[threadIdx.x & 63]) + 0u;
    uint32_t d0 = 0, d1 = 0;
    const long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < kLoops; ++it) {
#pragma unroll
        for (int k = 0; k < kBody; ++k) {
            const int i = k & 15;
            if (KIND == FMA) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (KIND == FMA_DEP) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
            } else if (KIND == PK_FMA) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            } else if (KIND == PK_FMA_DEP) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pm), "v"(pc));
            } else if (KIND == MUL) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (KIND == PK_MUL) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            } else if (KIND == ADD) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (KIND == PK_ADD) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            } else if (KIND == FRACT) {
                asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == CVT_FLR) {
                asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
            } else if (KIND == MED3_I32) {
                asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(lo), "v"(hi));
            } else if (KIND == MAD_I24) {
                asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(pitch), "v"(hi));
            } else if (KIND == LSHL_ADD) {
                asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(n[i]) : "v"(hi));
            } else if (KIND == FMA_MIX) {
                asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(half2));
            } else if (KIND == RCP) {
                asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == CVT_UBYTE) {
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(n[i]));
            } else if (KIND == FMA_AND_PK_FMA) {
                if (k & 1) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                } else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                }
            } else if (KIND == SUB) {
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (KIND == MAX_F32) {
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (KIND == MOV) {
                asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
            } else if (KIND == ADD_U32) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(hi));
            } else if (KIND == AND_B32) {
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(hi));
            } else if (KIND == CNDMASK) {
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(hi) : "vcc");
            } else if (KIND == CMP_LT) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
            } else if (KIND == FLOOR) {
                asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == FMAC) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (KIND == MUL_LIT) {
                asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(a[i]));
            } else if (KIND == MUL_LO_U32) {
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(pitch));
            } else if (KIND == MUL_U32_U24) {
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(pitch));
            } else if (KIND == MAD_U32_U24) {
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(pitch), "v"(hi));
            } else if (KIND == MUL_HI_U32) {
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(n[i]) : "v"(pitch));
            } else if (KIND == LSHRREV) {
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(n[i]) : "v"(n[(i + 1) & 15]));
            } else if (KIND == LSHLREV) {
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(n[i]) : "v"(n[(i + 1) & 15]));
            } else if (KIND == OR_B32) {
                asm volatile("v_or_b32 %0, %0, %1" : "+v"(n[i]) : "v"(hi));
            } else if (KIND == OR3) {
                asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(hi), "v"(lo));
            } else if (KIND == AND_OR) {
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(hi), "v"(lo));
            } else if (KIND == BFE_U32) {
                asm volatile("v_bfe_u32 %0, %1, 3, 8" : "=v"(n[i]) : "v"(n[(i + 1) & 15]));
            } else if (KIND == ADD_LSHL) {
                asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(n[i]) : "v"(hi));
            } else if (KIND == CVT_U32_F32) {
                asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
            } else if (KIND == CVT_F32_U32) {
                asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
            } else if (KIND == PERM) {
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(hi), "v"(lo));
            } else if (KIND == ADD3) {
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(hi), "v"(lo));
            } else if (KIND == SUB_U32) {
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(n[i]) : "v"(hi));
            } else if (KIND == XAD) {
                asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(hi), "v"(lo));
            } else if (KIND == LDS_READ_B32) {
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(8)" : "=v"(n[i]) : "v"(lds_addr) : "memory");
            } else if (KIND == LDS_READ2) {
                asm volatile("ds_read2st64_b32 %0, %2 offset1:1\n\ts_waitcnt lgkmcnt(8)" : "=v"(*(uint64_t *)&p[i]) : "v"(0), "v"(lds_addr) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) {
        s += a[i] + p[i].x + p[i].y + (float)n[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(d0 + d1);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}

template <int KIND>
static int run(float *dout, long long *dclk, FILE *csv)
{
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * w;  // 4 waves per workgroup, 256 CUs -> w waves per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, dout, 1.5f, dclk);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        long long hclk[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, dout, 1.5f, dclk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) {
                best = ms;
                CHECK(hipMemcpy(hclk, dclk, sizeof(hclk), hipMemcpyDeviceToHost));
            }
        }
        const double ops = (double)kBody * kLoops;                 // per wave
        const double ns_per_op_simd = best * 1e6 / (ops * w);      // the SIMD's w waves share its pipe
        const double ghz = hclk[1] > 0 ? (double)hclk[0] / ((double)hclk[1] * 10.0) : 0.0;  // s_memrealtime ticks at 100 MHz
        printf("%-52s waves/SIMD %d  %7.3f ms  %6.3f ns/op/SIMD  clock %.2f GHz  %5.2f cycles/op/SIMD  (first workgroup: %.2f cycles/op/wave)\n",
               kNames[KIND], w, best, ns_per_op_simd, ghz, ns_per_op_simd * ghz, (double)hclk[0] / ops);
        if (csv) {
            fprintf(csv, "\"%s\",%d,%.4f,%.4f,%.3f,%.3f,%.3f\n", kNames[KIND], w, best, ns_per_op_simd, ghz, ns_per_op_simd * ghz, (double)hclk[0] / ops);
        }
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main(int argc, char **argv)
{
    float *dout;
    long long *dclk;
    CHECK(hipMalloc(&dout, 2048 * 256 * sizeof(float)));
    CHECK(hipMalloc(&dclk, 2 * sizeof(long long)));
    FILE *csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) {
        fprintf(csv, "kind,waves_per_simd,launch_ms,ns_per_op_per_simd,clock_ghz,cycles_per_op_per_simd,first_wg_cycles_per_op_per_wave\n");
    }
    int rc = 0;
    rc |= run<FMA>(dout, dclk, csv);
    rc |= run<FMA_DEP>(dout, dclk, csv);
    rc |= run<PK_FMA>(dout, dclk, csv);
    rc |= run<PK_FMA_DEP>(dout, dclk, csv);
    rc |= run<FMA_AND_PK_FMA>(dout, dclk, csv);
    rc |= run<MUL>(dout, dclk, csv);
    rc |= run<PK_MUL>(dout, dclk, csv);
    rc |= run<ADD>(dout, dclk, csv);
    rc |= run<PK_ADD>(dout, dclk, csv);
    rc |= run<FRACT>(dout, dclk, csv);
    rc |= run<CVT_FLR>(dout, dclk, csv);
    rc |= run<MED3_I32>(dout, dclk, csv);
    rc |= run<MAD_I24>(dout, dclk, csv);
    rc |= run<LSHL_ADD>(dout, dclk, csv);
    rc |= run<FMA_MIX>(dout, dclk, csv);
    rc |= run<RCP>(dout, dclk, csv);
    rc |= run<CVT_UBYTE>(dout, dclk, csv);
    rc |= run<LDS_READ2>(dout, dclk, csv);
    rc |= run<LDS_READ_B32>(dout, dclk, csv);
    rc |= run<SUB>(dout, dclk, csv);
    rc |= run<MAX_F32>(dout, dclk, csv);
    rc |= run<MOV>(dout, dclk, csv);
    rc |= run<ADD_U32>(dout, dclk, csv);
    rc |= run<AND_B32>(dout, dclk, csv);
    rc |= run<CNDMASK>(dout, dclk, csv);
    rc |= run<CMP_LT>(dout, dclk, csv);
    rc |= run<FLOOR>(dout, dclk, csv);
    rc |= run<FMAC>(dout, dclk, csv);
    rc |= run<MUL_LIT>(dout, dclk, csv);
    rc |= run<MUL_LO_U32>(dout, dclk, csv);
    rc |= run<MUL_U32_U24>(dout, dclk, csv);
    rc |= run<MAD_U32_U24>(dout, dclk, csv);
    rc |= run<MUL_HI_U32>(dout, dclk, csv);
    rc |= run<LSHRREV>(dout, dclk, csv);
    rc |= run<LSHLREV>(dout, dclk, csv);
    rc |= run<OR_B32>(dout, dclk, csv);
    rc |= run<OR3>(dout, dclk, csv);
    rc |= run<AND_OR>(dout, dclk, csv);
    rc |= run<BFE_U32>(dout, dclk, csv);
    rc |= run<ADD_LSHL>(dout, dclk, csv);
    rc |= run<CVT_U32_F32>(dout, dclk, csv);
    rc |= run<CVT_F32_U32>(dout, dclk, csv);
    rc |= run<PERM>(dout, dclk, csv);
    rc |= run<ADD3>(dout, dclk, csv);
    rc |= run<SUB_U32>(dout, dclk, csv);
    rc |= run<XAD>(dout, dclk, csv);
    if (csv) {
        fclose(csv);
    }
    return rc;
}
