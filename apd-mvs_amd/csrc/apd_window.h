// apd_window.h -- per-wave LDS windows of a source view's texel-quad / texel-pair image and the fixed-patch NCC that reads them
// (used by the K6/K7 and K14/K15 window kernels; see apd_kernels_k67w.hip for the why).
#pragma once

#include "apd_device.h"


namespace apd {

// Window geometry.  One entry per texel (qx, qy): the pair {I(qx,qy), I(qx+1,qy) - I(qx,qy)}.
//   kQuad (8-bit input, staged from the byte quads): two binary16 values in 4 bytes -- integers up to 255 and their
//     differences are exact in binary16 -- lerped with one v_fma_mix_f32 each;
//   otherwise (float grey values, staged from the float texel-quad image): two binary32 values in 8 bytes, one v_fma_f32 each.
// A bilinear fetch at (qx, qy) reads the entries (qx, qy) and (qx, qy + 1) with one two-address LDS read: 4 VALU
// instructions for the whole lerp, same taps, same three fused multiply-adds as the global paths.  A window is as wide
// as the wave, so lane l stages column l of every row.
// The row pitch (in entries) is a template parameter: with pitch 64 the entries of one column share an LDS bank, which
// is free for the 32x4 checkerboard footprint of K6/K7 (a 32-lane group reads two pixel rows of opposite column parity)
// and a four-way conflict for the 8x8 footprint of K14/K15 (a group reads four rows of the same eight columns); those
// kernels use pitch 72, which moves consecutive rows by 8 (binary16 pairs) / 16 (binary32 pairs) banks.
constexpr int kWinW = 64;
// Float images: an entry is the texel alone (4 bytes) and the horizontal difference t(x+1) - t(x) is formed after the read --
// the same rounded subtraction the {texel, difference} pairs of the float texel-quad image store, two more plain FP32
// instructions per sample, half the LDS (K6/K7 47 -> 26 KB, K14/K15 79 -> 41 KB per workgroup: the occupancy of the 8-bit
// kernels).  -DAPD_WIN_F32_PAIRS=1 rebuilds the 8-byte {texel, difference} entries of rounds 1-2.
constexpr bool kWinF32Pairs = APD_WIN_F32_PAIRS != 0;
// LDS dwords of a window with WINH rows of fetch positions (+ the row below the last one)
constexpr int window_dwords(bool quad, int winh, int pitch = kWinW) { return pitch * (winh + 1) * ((quad || !kWinF32Pairs) ? 1 : 2); }
static_assert(kQuadShift == 2, "the window is staged from dword fetches of the quad image");

template <bool kQuad> struct WinEntry;
template <> struct WinEntry<true> {
    typedef uint32_t type;   // {binary16 t, binary16 dx}
    static constexpr int kShift = 2;
};
template <> struct WinEntry<false> {
    typedef pair_t type;     // what a tap read returns: {binary32 t, binary32 dx} (pairs) or {t(x), t(x + 1)} (single texels)
    static constexpr int kShift = kWinF32Pairs ? 3 : 2;
};

typedef __attribute__((address_space(3))) uint32_t *lds_u32_ptr;

// LDS byte address <-> pointer (32-bit in the local address space; the host pass only has to parse this)
__device__ __forceinline__ int lds_address(uint32_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(uintptr_t)(lds_u32_ptr)p;
#else
    return 0;
#endif
}

// entries (qx, qy) and (qx, qy + 1)
template <typename E>
struct WinTaps {
    E top, bot;
};

template <typename E, int kPitch>
__device__ __forceinline__ WinTaps<E> lds_read_pair(int addr)
{
    WinTaps<E> t;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) E *lds_ptr;
    const lds_ptr p = (lds_ptr)(uintptr_t)(uint32_t)addr;
    t.top = p[0];
    t.bot = p[kPitch];
#else
    t.top = t.bot = E();
#endif
    return t;
}

// single-texel float entries: {t(x, y), t(x + 1, y)} and the same of the row below -- two two-address reads
template <int kPitch>
__device__ __forceinline__ WinTaps<pair_t> lds_read_texels(int addr)
{
    WinTaps<pair_t> t;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) float *lds_ptr;
    const lds_ptr p = (lds_ptr)(uintptr_t)(uint32_t)addr;
    t.top.x = p[0];
    t.top.y = p[1];
    t.bot.x = p[kPitch];
    t.bot.y = p[kPitch + 1];
#else
    t.top = t.bot = pair_t();
#endif
    return t;
}


struct SrcWindow {
    int valid;        // wave-uniform: a window is staged
    int wx0, wy0;     // quad coordinates of window entry (0, 0) (wave-uniform)
    float lo_x, hi_x, lo_y, hi_y;  // a patch whose four corner samples lie in [lo, hi) reads the window only
    int addr0;        // LDS byte address of entry (0, 0) minus the byte offset of quad (wx0, wy0): address(qx, qy) =
                      // (qy * pitch + qx) * entry bytes + addr0
};

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v = fminf(v, __shfl_xor(v, m));
    }
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v = fmaxf(v, __shfl_xor(v, m));
    }
    return v;
}

// Every lane of the wave calls this (no divergence): centres a window with WINH rows of fetch positions on the bounding
// box of the points (cx, cy) of the lanes with `ok` and copies it from the quad image.  `win` is this wave's LDS region
// (window_dwords(kQuad, WINH, kPitch) dwords).
template <bool kQuad, int WINH, int kPitch = kWinW>
__device__ __forceinline__ SrcWindow stage_window_around(const FrameArgs &fa, const ViewConst &vc, uint32_t *win, bool ok, float cx, float cy)
{
    constexpr int kWinRows = WINH + 1;
    constexpr int kShift = WinEntry<kQuad>::kShift;
    SrcWindow w;
    const float big = 3.0e38f;
    const float x_lo = wave_min(ok ? cx : big), x_hi = wave_max(ok ? cx : -big);
    const float y_lo = wave_min(ok ? cy : big), y_hi = wave_max(ok ? cy : -big);
    if (!(x_lo <= x_hi)) {  // no live pixel projects into this view
        w.valid = 0;
        w.wx0 = w.wy0 = w.addr0 = 0;
        w.lo_x = w.lo_y = big;
        w.hi_x = w.hi_y = -big;
        return w;
    }
    // centre the window on the bounding box of the projected centres (all values are wave-uniform)
    const int wx0 = __builtin_amdgcn_readfirstlane((int)floorf(0.5f * (x_lo + x_hi)) - kWinW / 2);
    const int wy0 = __builtin_amdgcn_readfirstlane((int)floorf(0.5f * (y_lo + y_hi)) - WINH / 2);
    const int lane = threadIdx.x & 63;
    const int qp = fa.W + 1;
    // entries outside the image replicate the edge entry, exactly like the clamp of the global path
    const int col = med3_i32(wx0 + lane, -1, fa.W - 1) + 1;
    if constexpr (kQuad) {
        const global_quad_ptr srcq = (global_quad_ptr)vc.quad;
        const unsigned rpitch = quad_row_pitch_bytes(fa.W);
        if constexpr (kPair2) {
            // a column-pair dword holds the texels of rows gy and gy + 1 (clamping built in): one gather stages two window rows
            constexpr int kLoads = (kWinRows + 1) / 2;
            uint32_t tmp[kLoads];
#pragma unroll
            for (int k = 0; k < kLoads; ++k) {
                const int gy = min(max(wy0 + 2 * k, -1), fa.H - 1);  // wave-uniform
                tmp[k] = quad_fetch(srcq, (unsigned)(gy + 1) * rpitch + ((unsigned)col << kRowEntryShift));
            }
#pragma unroll
            for (int k = 0; k < kLoads; ++k) {
                float t0, t1, x0, x1;  // bytes {I(x,gy), I(x,gy+1), I(x+1,gy), I(x+1,gy+1)}
                asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(t0) : "v"(tmp[k]));
                asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(t1) : "v"(tmp[k]));
                asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(x0) : "v"(tmp[k]));
                asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(x1) : "v"(tmp[k]));
                win[(2 * k) * kPitch + lane] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(t0, x0 - t0));
                if (2 * k + 1 < kWinRows) {
                    win[(2 * k + 1) * kPitch + lane] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(t1, x1 - t1));
                }
            }
        } else {
            uint32_t tmp[kWinRows];
#pragma unroll
            for (int k = 0; k < kWinRows; ++k) {
                const int gy = min(max(wy0 + k, -1), fa.H - 1);  // wave-uniform
                tmp[k] = quad_fetch(srcq, (unsigned)(gy + 1) * rpitch + ((unsigned)col << kRowEntryShift));
            }
#pragma unroll
            for (int k = 0; k < kWinRows; ++k) {
                const float t0 = (float)(tmp[k] & 0xFFu);
                const float dx = (float)((tmp[k] >> 8) & 0xFFu) - t0;
                win[k * kPitch + lane] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(t0, dx));
            }
        }
    } else if constexpr (!kWinF32Pairs) {
        // texel (clamp(x), clamp(y)) of the plain float image: coalesced 256-byte row reads; the clamp makes the difference
        // formed after the read the one the float texel quads store (0 left of, right of and at the last column of the image)
        typedef const __attribute__((address_space(1))) float *global_float_ptr;
        const global_float_ptr img = (global_float_ptr)vc.img;
        float *winf = reinterpret_cast<float *>(win);
        const int cx = med3_i32(wx0 + lane, 0, fa.W - 1);
        float tmp[kWinRows];
#pragma unroll
        for (int k = 0; k < kWinRows; ++k) {
            const int gy = min(max(wy0 + k, 0), fa.H - 1);  // wave-uniform
            tmp[k] = img[(unsigned)(gy * fa.W + cx)];
        }
#pragma unroll
        for (int k = 0; k < kWinRows; ++k) {
            winf[k * kPitch + lane] = tmp[k];
        }
    } else {
        // the pair of texel row gy is the first half of float quad (., gy); the row below the image (gy == H, a copy
        // of row H - 1 by the clamp) is the second half of quad (., H - 1)
        const global_pair_ptr srcp = (global_pair_ptr)vc.fquad;  // two pairs per quad entry
        pair_t *winp = reinterpret_cast<pair_t *>(win);
        pair_t tmp[kWinRows];
#pragma unroll
        for (int k = 0; k < kWinRows; ++k) {
            const int gy = min(max(wy0 + k, -1), fa.H);  // wave-uniform
            const int qrow = min(gy, fa.H - 1) + 1;
            tmp[k] = srcp[2u * (unsigned)(qrow * qp + col) + (gy == fa.H ? 1u : 0u)];
        }
#pragma unroll
        for (int k = 0; k < kWinRows; ++k) {
            winp[k * kPitch + lane] = tmp[k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) {
        APD_WIN_COUNT(5, 1);
    }
    w.valid = 1;
    w.wx0 = wx0;
    w.wy0 = wy0;
    // one entry of margin on every side absorbs the rounding of the samples between the corners
    // (samples left of / above the image, where both taps are the clamped edge texel, take the global path: the window
    // path computes its LDS addresses in binary32 and relies on X, Y >= 0)
    w.lo_x = (float)max(wx0 + 1, 0);
    // single-texel float entries: a fetch at column c also reads column c + 1
    w.hi_x = (float)(wx0 + kWinW - ((kQuad || kWinF32Pairs) ? 1 : 2));
    w.lo_y = (float)max(wy0 + 1, 0);
    w.hi_y = (float)(wy0 + WINH - 1);
    w.addr0 = __builtin_amdgcn_readfirstlane(lds_address(win) - (wy0 * kPitch + wx0) * (1 << kShift));
    return w;
}

// byte address of window entry (qx, qy): qy * pitch + (qx << shift) + addr0
template <int kShift, int kPitch>
__device__ __forceinline__ int win_byte_address(int qx, int qy, int addr0)
{
    int row, off;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row) : "v"(qy), "v"(kPitch << kShift), "v"(addr0));
    asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(off) : "v"(qx), "v"(row), "n"(kShift));
    return off;
}

// quad_row_issue / fquad_row_issue for samples known to lie inside the window: no clamps, LDS addresses.
// kEnds: the positions of samples 0 and kPatchN - 1 of this row are patch corners the caller has already computed with
// corner_position (the same instructions on the same operands, hence the same bits): (Xa, Ya) and (Xb, Yb).
template <bool kQuad, int kPitch, bool kApprox = false, bool kEnds = false>
__device__ __forceinline__ void win_row_issue(const Homography &H, float bx, float by, float bz, const float (&yf)[kPatchN], int addr0,
                                              float (&a)[kPatchN], float (&b)[kPatchN],
                                              WinTaps<typename WinEntry<kQuad>::type> (&t)[kPatchN], float Xa = 0.0f, float Ya = 0.0f,
                                              float Xb = 0.0f, float Yb = 0.0f)
{
    constexpr int j0 = kEnds ? 1 : 0, j1 = kEnds ? kPatchN - 1 : kPatchN;
    float z[kPatchN], X[kPatchN], Y[kPatchN], r[kPatchN];
#pragma unroll
    for (int j = j0; j < j1; ++j) {
        z[j] = fmaf(H.h[7], yf[j], bz);
        X[j] = fmaf(H.h[1], yf[j], bx);
        Y[j] = fmaf(H.h[4], yf[j], by);
    }
    APD_STAGE();
#pragma unroll
    for (int j = j0; j < j1; ++j) {
        r[j] = __builtin_amdgcn_rcpf(z[j]);
    }
    APD_STAGE();
    if constexpr (!kApprox) {  // Newton step: the correctly rounded reciprocal (tolerance mode APD_OPT_FAST_RCP stops at v_rcp_f32)
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            z[j] = fmaf(-z[j], r[j], 1.0f);
        }
        APD_STAGE();
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            r[j] = fmaf(z[j], r[j], r[j]);
        }
        APD_STAGE();
    }
#pragma unroll
    for (int j = j0; j < j1; ++j) {
        X[j] *= r[j];
        Y[j] *= r[j];
    }
    if constexpr (kEnds) {
        X[0] = Xa;
        Y[0] = Ya;
        X[kPatchN - 1] = Xb;
        Y[kPatchN - 1] = Yb;
    }
    APD_STAGE();
    // LDS address of entry (floor X, floor Y) in binary32.  On gfx950 the plain binary32 multiply / add / FMA issue in 2
    // cycles per wave, everything else (v_fract, conversions, integer multiply-add, shift-add) in 4 (tools/valu_issue.hip),
    // so the two floor conversions + v_mad_i32_i24 + v_lshl_add_u32 (16 cycles) become two subtractions, two FMAs and one
    // conversion (13 cycles).  Exact: samples of the window path have X, Y >= 0 (SrcWindow::lo_x / lo_y), where
    // v_fract_f32(X) == X - floor(X) exactly, so X - fract(X) is floor(X); the address is an integer below 2^24 at every step.
    // The last step is not a conversion either: with 2^23 added to the constant term the sum lies in [2^23, 2^24), where
    // the low 23 bits of the binary32 encoding are the integer itself -- one v_and_b32 (2 cycles) instead of v_cvt_i32_f32
    // (4).  addr0 is above -2^23 - 2^20 for every image apd_create accepts (height <= 16384, window rows of at most 576
    // bytes), so every step stays an integer of magnitude below 2^24.  (ds_read does not ignore the high address bits: tools/lds_addr_bits.hip.)
    constexpr float kEntryBytes = (float)(1 << WinEntry<kQuad>::kShift), kPitchBytes = (float)(kPitch << WinEntry<kQuad>::kShift);
    const float addr0f = (float)addr0 + (APD_WIN_ADDR_MAGIC ? 8388608.0f : 0.0f);
    float fx[kPatchN], fy[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        a[j] = __builtin_amdgcn_fractf(X[j]);
        b[j] = __builtin_amdgcn_fractf(Y[j]);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        fx[j] = X[j] - a[j];
        fy[j] = Y[j] - b[j];
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        fx[j] = fmaf(fx[j], kEntryBytes, addr0f);
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        fx[j] = fmaf(fy[j], kPitchBytes, fx[j]);
    }
    APD_STAGE();
    int addr[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        addr[j] = APD_WIN_ADDR_MAGIC ? (int)(__float_as_uint(fx[j]) & 0x007FFFFFu) : (int)fx[j];
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        if constexpr (kQuad || kWinF32Pairs) {
            t[j] = lds_read_pair<typename WinEntry<kQuad>::type, kPitch>(addr[j]);
        } else {
            t[j] = lds_read_texels<kPitch>(addr[j]);
        }
    }
}

// Pairs + weights of one row -> six bilinear values: fmaf(a, t10 - t00, t00), fmaf(a, t11 - t01, t01), fmaf(b, bot - top, top).
template <bool kQuad>
__device__ __forceinline__ void win_row_lerp(const WinTaps<typename WinEntry<kQuad>::type> (&t)[kPatchN], const float (&a)[kPatchN],
                                             const float (&b)[kPatchN], float (&v)[kPatchN])
{
    float top[kPatchN], bot[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        if constexpr (kQuad) {
            top[j] = lerp_f16_pair(a[j], t[j].top);
            bot[j] = lerp_f16_pair(a[j], t[j].bot);
        } else if constexpr (kWinF32Pairs) {
            top[j] = fmaf(a[j], t[j].top.y, t[j].top.x);
            bot[j] = fmaf(a[j], t[j].bot.y, t[j].bot.x);
        } else {  // {t(x), t(x + 1)}: the difference the pairs store, formed here
            top[j] = fmaf(a[j], t[j].top.y - t[j].top.x, t[j].top.x);
            bot[j] = fmaf(a[j], t[j].bot.y - t[j].bot.x, t[j].bot.x);
        }
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        bot[j] -= top[j];
    }
    APD_STAGE();
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        v[j] = fmaf(b[j], bot[j], top[j]);
    }
}

// ncc_fixed_moments (fast reciprocal) reading the window.
// cX / cY: positions of the four corner samples {(x0,y0), (x0,y1), (x1,y0), (x1,y1)} from the caller's window test; rows 0 and
// kPatchN - 1 take their end samples from there instead of computing them a second time (APD_WIN_CORNER_REUSE=0: recompute).
// kLocal (K14 / K15): the sample coordinates are formed inside the body, by exact binary32 additions to the patch's first
// coordinate taken through an opaque copy -- (float)(px - 5) + 2 i is the integer (float)(px + 2 i - 5) of the other form, bit for
// bit.  Without it the compiler shares these twelve conversions with the global-path body and with the corner test, hoists them
// out of K14's sample / view / chunk loops and, at 128 registers, spills them: the body then opens three rows with a scratch
// reload and `s_waitcnt vmcnt(0)` (a memory round trip per row, profiles/r05/k14_body_spills.txt).
__device__ __forceinline__ float opaque_f32(float v)
{
    asm volatile("" : "+v"(v));
    return v;
}

template <bool kQuad, int kPitch, bool kApprox, bool kLocal = false, typename Ref>
__device__ __forceinline__ void ncc_window_moments(const Ref &rp, const Homography &H, int px, int py, int addr0, float &sum_s,
                                                   float &sum_ss, float &sum_rs, const float (&cX)[4], const float (&cY)[4])
{
    constexpr bool kReuse = APD_WIN_CORNER_REUSE != 0;
    float yf[kPatchN];
    float x_first = 0.0f;
    if constexpr (kLocal) {
        x_first = opaque_f32((float)(px - kPatchRadius));
        const float y_first = opaque_f32((float)(py - kPatchRadius));
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            yf[j] = y_first + (float)(kPatchStep * j);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            yf[j] = (float)(py + kPatchStep * j - kPatchRadius);
        }
    }
    sum_s = 0.0f;
    sum_ss = 0.0f;
    sum_rs = 0.0f;
#if APD_WIN_SETPRIO > 0
    __builtin_amdgcn_s_setprio(APD_WIN_SETPRIO);
#endif
#if APD_WIN_DRAIN_SMEM
    // Scalar loads return out of order: while one is pending (the per-view constants a caller prefetches for its next NCC), the
    // compiler can only wait for "every LDS and scalar access" -- and the first row of the body then ends with `s_waitcnt
    // lgkmcnt(0)` right after the NEXT row's six reads have been issued: a whole LDS round trip per NCC.  Waiting for the scalar
    // loads here, where they have had the whole prologue to arrive, lets every wait of the body be a counted one.
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), vmcnt / expcnt untouched
#endif
    float a[2][kPatchN], b[2][kPatchN];
    WinTaps<typename WinEntry<kQuad>::type> t[2][kPatchN];
    {
        const float xf = kLocal ? x_first : (float)(px - kPatchRadius);
        win_row_issue<kQuad, kPitch, kApprox, kReuse>(H, fmaf(H.h[0], xf, H.h[2]), fmaf(H.h[3], xf, H.h[5]), fmaf(H.h[6], xf, H.h[8]), yf,
                                                      addr0, a[0], b[0], t[0], cX[0], cY[0], cX[1], cY[1]);
    }
#pragma unroll
    for (int i = 0; i < kPatchN; ++i) {
        float v[kPatchN];
        // LDS returns in order: the reference texels of this row are requested before the window entries of the next
        // one, so the reduction below only waits for reads that were issued a whole row of arithmetic ago
        float ref[kPatchN];
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            ref[j] = rp.at(i, j);
        }
        APD_STAGE();
        if (i + 2 < kPatchN) {
            const float xf = kLocal ? x_first + (float)(kPatchStep * (i + 1)) : (float)(px + kPatchStep * (i + 1) - kPatchRadius);
            win_row_issue<kQuad, kPitch, kApprox>(H, fmaf(H.h[0], xf, H.h[2]), fmaf(H.h[3], xf, H.h[5]), fmaf(H.h[6], xf, H.h[8]), yf,
                                                  addr0, a[(i + 1) & 1], b[(i + 1) & 1], t[(i + 1) & 1]);
        } else if (i + 1 < kPatchN) {
            const float xf = kLocal ? x_first + (float)(kPatchStep * (i + 1)) : (float)(px + kPatchStep * (i + 1) - kPatchRadius);
            win_row_issue<kQuad, kPitch, kApprox, kReuse>(H, fmaf(H.h[0], xf, H.h[2]), fmaf(H.h[3], xf, H.h[5]), fmaf(H.h[6], xf, H.h[8]), yf,
                                                          addr0, a[(i + 1) & 1], b[(i + 1) & 1], t[(i + 1) & 1], cX[2], cY[2], cX[3], cY[3]);
        }
        APD_STAGE();
        win_row_lerp<kQuad>(t[i & 1], a[i & 1], b[i & 1], v);
        float row_s = 0.0f, row_ss = 0.0f, row_rs = 0.0f;
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            row_s += v[j];
            row_ss = fmaf(v[j], v[j], row_ss);
            row_rs = fmaf(ref[j], v[j], row_rs);
        }
        sum_s += row_s;
        sum_ss += row_ss;
        sum_rs += row_rs;
    }
#if APD_WIN_SETPRIO > 0
    __builtin_amdgcn_s_setprio(0);
#endif
}

// Sample position of patch corner (xf, yf), computed exactly like the samples themselves (fast reciprocal).
template <bool kApprox = false>
__device__ __forceinline__ void corner_position(const Homography &H, float xf, float yf, float &X, float &Y)
{
    const float z = fmaf(H.h[7], yf, fmaf(H.h[6], xf, H.h[8]));
    const float r = kApprox ? __builtin_amdgcn_rcpf(z) : recip_fast(z);
    X = fmaf(H.h[1], yf, fmaf(H.h[0], xf, H.h[2])) * r;
    Y = fmaf(H.h[4], yf, fmaf(H.h[3], xf, H.h[5])) * r;
}

// ComputeBilateralNCCOld (APD.cu:530-614) for plane q = n/d against source view vc, window first.
// kApprox: tolerance mode (bare v_rcp_f32 everywhere, no IEEE body); see quad_row_issue
template <bool kQuad, int kPitch = kWinW, bool kTiled = false, bool kApprox = false, bool kLocal = false, typename Ref>
__device__ __forceinline__ float ncc_fixed_windowed(const FrameArgs &fa, const ViewConst &vc, const SrcWindow &w, const Ref &rp, int px,
                                                    int py, float qx, float qy, float qz)
{
    const Homography H = make_homography(fa, vc, qx, qy, qz);
    float cx, cy;
    correspond(H, (float)px, (float)py, cx, cy);
    if (cx >= vc.wf || cx < 0.0f || cy >= vc.hf || cy < 0.0f) {
        return 2.0f;
    }
    const float kMinVar = 1e-5f;
    if (rp.var < kMinVar) {
        return 2.0f;
    }
    const float x0 = (float)(px - kPatchRadius), x1 = (float)(px + kPatchRadius);
    const float y0 = (float)(py - kPatchRadius), y1 = (float)(py + kPatchRadius);
    // tolerance mode: every denominator goes through v_rcp_f32; the corner test below still needs one sign
    const bool fast_recip = denominators_fast(H, x0, x1, y0, y1);
    bool in_window = false;
    float cX[4], cY[4];  // only read when in_window, i.e. after the block below has written them
    if (fast_recip && w.valid) {
        // x/z and y/z are monotone along every row and every column of the sample grid while z keeps its sign, so the
        // four corner samples bound all 36
        corner_position<kApprox>(H, x0, y0, cX[0], cY[0]);
        corner_position<kApprox>(H, x0, y1, cX[1], cY[1]);
        corner_position<kApprox>(H, x1, y0, cX[2], cY[2]);
        corner_position<kApprox>(H, x1, y1, cX[3], cY[3]);
        const float xl = fminf(fminf(cX[0], cX[1]), fminf(cX[2], cX[3])), xh = fmaxf(fmaxf(cX[0], cX[1]), fmaxf(cX[2], cX[3]));
        const float yl = fminf(fminf(cY[0], cY[1]), fminf(cY[2], cY[3])), yh = fmaxf(fmaxf(cY[0], cY[1]), fmaxf(cY[2], cY[3]));
        in_window = xl >= w.lo_x && xh < w.hi_x && yl >= w.lo_y && yh < w.hi_y;
    }
    APD_LAB_NCC_STATS(in_window, fast_recip);
    // one path per wave and NCC: lanes inside and outside the window would otherwise run both 36-sample bodies in turn
    in_window = in_window && __builtin_amdgcn_ballot_w64(!in_window) == 0;
    // the same for the two global bodies: if one lane needs the IEEE division, every lane of the wave takes it (same bits)
    const bool fast_body = __builtin_amdgcn_ballot_w64(!fast_recip) == 0;
    float sum_s, sum_ss, sum_rs;
    if (in_window) {
        ncc_window_moments<kQuad, kPitch, kApprox, kLocal>(rp, H, px, py, w.addr0, sum_s, sum_ss, sum_rs, cX, cY);
    } else if constexpr (kApprox) {
        ncc_fixed_moments<kQuad, kRecipApprox, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    } else if (__builtin_expect(fast_body, 1)) {
        ncc_fixed_moments<kQuad, kRecipExact, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    } else {
        ncc_fixed_moments_ieee<kQuad, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    }
    const float inv_w = 1.0f / 36.0f;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    if (var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-rp.mean, sum_s, sum_rs);
    return ncc_cost_from_moments(rp.var, var_s, covar);
}

// The same cost for an already projected centre (the caller has done the bounds test of APD.cu:546): K9/K10's centre patch.
// (A function of its own rather than the tail of ncc_fixed_windowed: routing K6/K7 through it changes that kernel's block layout and
// spills -- 512 -> 592 B of scratch per lane in one arrangement -- and K6/K7 is the headline kernel.)
// kApprox: tolerance mode (bare v_rcp_f32 everywhere, no IEEE body); see quad_row_issue
// kDivergent: lanes inside and lanes outside the window each take their own 36-sample body (a mixed wave runs both in turn).  The
//   default is one body per wave and NCC -- right for the kernels the vector ALU bounds (K6/K7, K14, K15); K9/K10 is bound by the
//   L1's tag accesses, which only the lanes on the global path make.
template <bool kQuad, int kPitch = kWinW, bool kTiled = false, bool kApprox = false, bool kDivergent = false, typename Ref>
__device__ __forceinline__ float ncc_fixed_windowed_from_h(const FrameArgs &fa, const ViewConst &vc, const SrcWindow &w, const Ref &rp,
                                                           const Homography &H, int px, int py)
{
    const float kMinVar = 1e-5f;
    if (rp.var < kMinVar) {
        return 2.0f;
    }
    const float x0 = (float)(px - kPatchRadius), x1 = (float)(px + kPatchRadius);
    const float y0 = (float)(py - kPatchRadius), y1 = (float)(py + kPatchRadius);
    // tolerance mode: every denominator goes through v_rcp_f32; the corner test below still needs one sign
    const bool fast_recip = denominators_fast(H, x0, x1, y0, y1);
    bool in_window = false;
    float cX[4], cY[4];  // only read when in_window, i.e. after the block below has written them
    if (fast_recip && w.valid) {
        // x/z and y/z are monotone along every row and every column of the sample grid while z keeps its sign, so the
        // four corner samples bound all 36
        corner_position<kApprox>(H, x0, y0, cX[0], cY[0]);
        corner_position<kApprox>(H, x0, y1, cX[1], cY[1]);
        corner_position<kApprox>(H, x1, y0, cX[2], cY[2]);
        corner_position<kApprox>(H, x1, y1, cX[3], cY[3]);
        const float xl = fminf(fminf(cX[0], cX[1]), fminf(cX[2], cX[3])), xh = fmaxf(fmaxf(cX[0], cX[1]), fmaxf(cX[2], cX[3]));
        const float yl = fminf(fminf(cY[0], cY[1]), fminf(cY[2], cY[3])), yh = fmaxf(fmaxf(cY[0], cY[1]), fmaxf(cY[2], cY[3]));
        in_window = xl >= w.lo_x && xh < w.hi_x && yl >= w.lo_y && yh < w.hi_y;
    }
    APD_LAB_NCC_STATS(in_window, fast_recip);
    if constexpr (!kDivergent) {
        // one path per wave and NCC: lanes inside and outside the window would otherwise run both 36-sample bodies in turn
        in_window = in_window && __builtin_amdgcn_ballot_w64(!in_window) == 0;
    }
    // the same for the two global bodies: if one lane needs the IEEE division, every lane of the wave takes it (same bits)
    const bool fast_body = __builtin_amdgcn_ballot_w64(!fast_recip) == 0;
    float sum_s, sum_ss, sum_rs;
    if (in_window) {
        ncc_window_moments<kQuad, kPitch, kApprox>(rp, H, px, py, w.addr0, sum_s, sum_ss, sum_rs, cX, cY);
    } else if constexpr (kApprox) {
        ncc_fixed_moments<kQuad, kRecipApprox, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    } else if (__builtin_expect(fast_body, 1)) {
        ncc_fixed_moments<kQuad, kRecipExact, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    } else {
        ncc_fixed_moments_ieee<kQuad, kTiled, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
    }
    const float inv_w = 1.0f / 36.0f;
    sum_s *= inv_w;
    sum_ss *= inv_w;
    sum_rs *= inv_w;
    const float var_s = fmaf(-sum_s, sum_s, sum_ss);
    if (var_s < kMinVar) {
        return 2.0f;
    }
    const float covar = fmaf(-rp.mean, sum_s, sum_rs);
    return ncc_cost_from_moments(rp.var, var_s, covar);
}

}  // namespace apd

