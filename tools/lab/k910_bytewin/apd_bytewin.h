// apd_bytewin.h -- a per-wave LDS window of RAW 8-bit texels of one source view, wide enough for everything an NCCNew of K9/K10
// samples: the pixel's own 6 x 6 patch AND the 3 x 3 sub-patches around its reliable neighbours (APD.cu:400-528).
//
// Why.  K9/K10 was bound by the L1 tag pipeline: 72 of the 108 samples of an NCCNew are sub-patch taps, every one a scattered
// gather that costs one tag access per lane (tools/tcp_patterns.hip).  The {binary16 texel, difference} windows of K6/K7 / K14 are
// 4 bytes per texel and 64 columns wide -- a sub-patch sits where its ANCHOR is, a median of 27 px (p90: 60 px) from the pixel
// (tools/nb_cluster.py, profiles/r06/nb_cluster_*.txt), so those windows only ever held the centre patch.  At ONE byte per texel a
// 128-column window with 96 .. 128 rows costs 12 .. 16 KB and holds 78 .. 85 % of a wave's anchors.
//
// Layout: row-major bytes, kBwCols per row.  A bilinear fetch at (qx, qy) needs texels (qx, qy), (qx + 1, qy), (qx, qy + 1),
// (qx + 1, qy + 1): two dword pairs (ds_read2_b32 at the enclosing dword and the next one, same for the row below), one
// v_alignbyte_b32 each to bring texel qx to byte 0, one v_perm_b32 to interleave the rows -- the result is the SAME dword the global
// column-pair image returns for that fetch ({I(x,y), I(x,y+1), I(x+1,y), I(x+1,y+1)}, apd_device.h: quad_fetch), so everything after
// the fetch (quad_lerp / quad_row_lerp, the moments) is shared with the global path: same taps, same three fused multiply-adds,
// bit-identical.  Which lanes read LDS is decided per lane; a wave with lanes on both sides runs both fetch sequences (a dozen
// instructions each) and one copy of everything else.
#pragma once

#include "apd_window.h"

namespace apd {

constexpr int kBwCols = APD_K910_BW_COLS;   // bytes per window row; a lane stages kBwCols / 64 columns
constexpr int kBwRows = APD_K910_BW_ROWS;   // window rows (even)
static_assert(kBwCols == 128 && (kBwRows % 2) == 0 && kBwRows >= 32, "byte window: 128 columns (two per lane), an even number of rows");
constexpr int bytewin_dwords() { return kBwCols * kBwRows / 4; }

struct ByteWindow {
    int valid;                      // wave-uniform: a window is staged
    float lo_x, hi_x, lo_y, hi_y;   // a sample with lo <= (X, Y) < hi reads the window (all four texels inside, X, Y >= 0)
    float addr0f;                   // 2^23 + LDS byte address of texel (0, 0) of the IMAGE, were the window to reach that far:
                                    // byte address of texel (qx, qy) = qy * kBwCols + qx + (addr0f - 2^23)
};

__device__ __forceinline__ ByteWindow no_byte_window()
{
    ByteWindow w;
    w.valid = 0;
    w.lo_x = w.lo_y = 3.0e38f;
    w.hi_x = w.hi_y = -3.0e38f;
    w.addr0f = 0.0f;
    return w;
}

// Every lane of the wave calls this (no divergence): centres the window on the bounding box of the points (cx, cy) of the lanes with
// `ok` and copies it from the column-pair image.  One pair-image dword = texels (x, y), (x, y + 1), (x + 1, y), (x + 1, y + 1) with
// the clamp-to-edge of the global path built in (entry (t, u) = texel (clamp(t - 1), clamp(u - 1)) and the one below it): lane l
// stages columns 2 l and 2 l + 1, two window rows per load.
__device__ __forceinline__ ByteWindow stage_byte_window(const FrameArgs &fa, const ViewConst &vc, uint32_t *win, bool ok, float cx, float cy)
{
    static_assert(kPair2 && kQuadShift == 2, "the byte window is staged from the 2-byte column pairs");
    ByteWindow w = no_byte_window();
    const float big = 3.0e38f;
    const float x_lo = wave_min(ok ? cx : big), x_hi = wave_max(ok ? cx : -big);
    const float y_lo = wave_min(ok ? cy : big), y_hi = wave_max(ok ? cy : -big);
    if (!(x_lo <= x_hi)) {  // no live pixel projects into this view
        return w;
    }
    const int wx0 = __builtin_amdgcn_readfirstlane((int)floorf(0.5f * (x_lo + x_hi)) - kBwCols / 2);
    const int wy0 = __builtin_amdgcn_readfirstlane((int)floorf(0.5f * (y_lo + y_hi)) - kBwRows / 2);
    const int lane = threadIdx.x & 63;
    const global_quad_ptr srcq = (global_quad_ptr)vc.quad;
    const unsigned rpitch = quad_row_pitch_bytes(fa.W);
    // entry column of texel column x: clamp(x, -1, W - 1) + 1; the dword there also holds entry + 1 = texel clamp(x + 1) (for x < -1 and
    // x >= W - 1 both halves are the same edge texel, which is what the clamp of the global path gives for x and x + 1)
    const int col = med3_i32(wx0 + 2 * lane, -1, fa.W - 1) + 1;
    uint16_t *win16 = reinterpret_cast<uint16_t *>(win);
    constexpr int kLoads = kBwRows / 2;
    constexpr int kBatch = 8;   // loads in flight per lane
#pragma unroll 1
    for (int k0 = 0; k0 < kLoads; k0 += kBatch) {
        uint32_t tmp[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int gy = min(max(wy0 + 2 * (k0 + k), -1), fa.H - 1);  // wave-uniform; rows gy and gy + 1 (clamped like the global path)
            tmp[k] = (k0 + k < kLoads) ? quad_fetch(srcq, (unsigned)(gy + 1) * rpitch + ((unsigned)col << kRowEntryShift)) : 0u;
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (k0 + k < kLoads) {
                // bytes {I(x,y), I(x,y+1), I(x+1,y), I(x+1,y+1)} -> row y: {I(x,y), I(x+1,y)}, row y + 1: {I(x,y+1), I(x+1,y+1)}
                const uint32_t top = __builtin_amdgcn_perm(0u, tmp[k], 0x0c0c0200u);
                const uint32_t bot = __builtin_amdgcn_perm(0u, tmp[k], 0x0c0c0301u);
                win16[(2 * (k0 + k)) * (kBwCols / 2) + lane] = (uint16_t)top;
                win16[(2 * (k0 + k) + 1) * (kBwCols / 2) + lane] = (uint16_t)bot;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    w.valid = 1;
    // texels x and x + 1, rows y and y + 1 must be window entries, and X, Y >= 0 (the address arithmetic below floors by X - fract(X))
    w.lo_x = (float)max(wx0, 0);
    w.hi_x = (float)(wx0 + kBwCols - 1);
    w.lo_y = (float)max(wy0, 0);
    w.hi_y = (float)(wy0 + kBwRows - 1);
    const int addr0 = __builtin_amdgcn_readfirstlane(lds_address(win) - (wy0 * kBwCols + wx0));
    w.addr0f = (float)addr0 + 8388608.0f;   // |addr0| < 2^22 for every image apd_create accepts (height <= 16384): exact
    return w;
}

// The fetch of quad_fetch(srcq, offset of (floor X, floor Y)) from the window, for a sample inside [lo, hi): X, Y >= 0, so
// fx = X - fract(X) and fy = Y - fract(Y) are the floors, exactly; the byte address is an integer below 2^23 and its binary32
// encoding with 2^23 added carries it in the low 23 bits (the trick of apd_window.h: win_row_issue).
__device__ __forceinline__ quad_t bytewin_fetch(float addr0f, float X, float Y, float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float fx = X - a, fy = Y - b;
    const float af = fmaf(fy, (float)kBwCols, fx + addr0f);
    const uint32_t addr = __float_as_uint(af) & 0x007FFFFFu;
    typedef __attribute__((address_space(3))) uint32_t *lds_ptr;
#if APD_K910_BW_U16
    // two 2-byte reads at a byte address that is odd for odd qx: LDS accesses need no alignment on gfx950 (the parity suite would show
    // a wrong byte at once); a quarter of the bank accesses of the dword-pair form below
    typedef __attribute__((address_space(3))) uint16_t __attribute__((aligned(1))) *lds_u16_ptr;
    const lds_u16_ptr q = (lds_u16_ptr)(uintptr_t)addr;
    const uint32_t top = q[0];                 // bytes {I(x,y), I(x+1,y)}
    const uint32_t bot = q[kBwCols / 2];       // bytes {I(x,y+1), I(x+1,y+1)}
#else
    const lds_ptr p = (lds_ptr)(uintptr_t)(addr & ~3u);
    const uint32_t t_lo = p[0], t_hi = p[1], b_lo = p[kBwCols / 4], b_hi = p[kBwCols / 4 + 1];
    const uint32_t sh = addr & 3u;
    const uint32_t top = __builtin_amdgcn_alignbyte(t_hi, t_lo, sh);   // bytes {I(x,y), I(x+1,y), ..}
    const uint32_t bot = __builtin_amdgcn_alignbyte(b_hi, b_lo, sh);   // bytes {I(x,y+1), I(x+1,y+1), ..}
#endif
    return __builtin_amdgcn_perm(bot, top, 0x05010400u);               // {I(x,y), I(x,y+1), I(x+1,y), I(x+1,y+1)}: the column-pair dword
#else
    return 0u;
#endif
}

// global fetch of the same sample (the tail of quad_row_issue / subpatch_issue_quad)
__device__ __forceinline__ quad_t global_tap_fetch(global_quad_ptr srcq, unsigned qpitch, int wm1, int hm1, float X, float Y)
{
    const int qx = med3_i32(cvt_floor_i32(X), -1, wm1);
    const int qy = med3_i32(cvt_floor_i32(Y), -1, hm1);
    return quad_fetch(srcq, quad_byte_offset(qx, qy, (int)qpitch, (int)(qpitch + kRowEntryBytes)));
}

// Sample positions of n points (xf[k], yf[k]) under H with the exact fast reciprocal -- the first stages of quad_row_issue /
// subpatch_issue_quad, instruction for instruction.
template <int N>
__device__ __forceinline__ void warp_positions(const float (&z_in)[N], float (&X)[N], float (&Y)[N])
{
    float z[N], r[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        z[k] = z_in[k];
        r[k] = __builtin_amdgcn_rcpf(z[k]);
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        z[k] = fmaf(-z[k], r[k], 1.0f);
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        r[k] = fmaf(z[k], r[k], r[k]);
    }
    APD_STAGE();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        X[k] *= r[k];
        Y[k] *= r[k];
    }
}

// subpatch_cost_quad<kRecipExact> with the nine fetches from the window for the lanes whose sub-patch lies inside it.  The four
// corner samples of the 3 x 3 grid bound the other five (x / z and y / z are monotone along the rows and columns of the grid while z
// keeps its sign: the caller has checked denominators_fast on the grid's bounding box).
__device__ __forceinline__ float subpatch_cost_bytewin(const Homography &H, const ByteWindow &w, global_quad_ptr srcq, unsigned qpitch, int wm1,
                                                       int hm1, int cx, int cy, const uint32_t (&ref_rows)[kSubN], float mean_r, float var_r)
{
    constexpr int N = kSubN * kSubN;
    float z[N], X[N], Y[N];
#pragma unroll
    for (int i = 0; i < kSubN; ++i) {
        const float xf = (float)(cx + kSubStep * (i - 1));
        const float bx = fmaf(H.h[0], xf, H.h[2]);
        const float by = fmaf(H.h[3], xf, H.h[5]);
        const float bz = fmaf(H.h[6], xf, H.h[8]);
#pragma unroll
        for (int j = 0; j < kSubN; ++j) {
            const float yf = (float)(cy + kSubStep * (j - 1));
            z[i * kSubN + j] = fmaf(H.h[7], yf, bz);
            X[i * kSubN + j] = fmaf(H.h[1], yf, bx);
            Y[i * kSubN + j] = fmaf(H.h[4], yf, by);
        }
    }
    APD_STAGE();
    warp_positions<N>(z, X, Y);
    APD_STAGE();
    float a[N], b[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        a[k] = __builtin_amdgcn_fractf(X[k]);
        b[k] = __builtin_amdgcn_fractf(Y[k]);
    }
    const float xl = fminf(fminf(X[0], X[2]), fminf(X[6], X[8])), xh = fmaxf(fmaxf(X[0], X[2]), fmaxf(X[6], X[8]));
    const float yl = fminf(fminf(Y[0], Y[2]), fminf(Y[6], Y[8])), yh = fmaxf(fmaxf(Y[0], Y[2]), fmaxf(Y[6], Y[8]));
    const bool inside = xl >= w.lo_x && xh < w.hi_x && yl >= w.lo_y && yh < w.hi_y;   // false for NaN and without a window
    APD_LAB_BYTEWIN(inside);
    quad_t t[N];
    if (inside) {
        // three fetches (twelve LDS dwords) in flight at a time: nine would hold 36 registers for the raw dwords alone
#pragma unroll
        for (int i = 0; i < kSubN; ++i) {
#pragma unroll
            for (int j = 0; j < kSubN; ++j) {
                const int k = i * kSubN + j;
                t[k] = bytewin_fetch(w.addr0f, X[k], Y[k], a[k], b[k]);
            }
            APD_STAGE();
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            t[k] = global_tap_fetch(srcq, qpitch, wm1, hm1, X[k], Y[k]);
        }
    }
    APD_STAGE();
    return subpatch_finish_quad(t, a, b, ref_rows, mean_r, var_r);
}

// ncc_fixed_moments<true, kRecipExact> (the 36 samples of the pixel's own patch) with the fetches of the lanes whose patch lies inside
// the window served from it.  cX / cY: the four corner samples from the caller's test.
template <typename Ref>
__device__ __forceinline__ void ncc_bytewin_moments(const Ref &rp, const Homography &H, const ByteWindow &w, bool inside, global_quad_ptr srcq,
                                                    unsigned qpitch, int wm1, int hm1, int px, int py, float &sum_s, float &sum_ss, float &sum_rs)
{
    float yf[kPatchN];
#pragma unroll
    for (int j = 0; j < kPatchN; ++j) {
        yf[j] = (float)(py + kPatchStep * j - kPatchRadius);
    }
    sum_s = 0.0f;
    sum_ss = 0.0f;
    sum_rs = 0.0f;
    float a[2][kPatchN], b[2][kPatchN];
    quad_t t[2][kPatchN];
    auto issue = [&](int row, int buf) {
        const float xf = (float)(px + kPatchStep * row - kPatchRadius);
        const float bx = fmaf(H.h[0], xf, H.h[2]), by = fmaf(H.h[3], xf, H.h[5]), bz = fmaf(H.h[6], xf, H.h[8]);
        float z[kPatchN], X[kPatchN], Y[kPatchN];
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            z[j] = fmaf(H.h[7], yf[j], bz);
            X[j] = fmaf(H.h[1], yf[j], bx);
            Y[j] = fmaf(H.h[4], yf[j], by);
        }
        APD_STAGE();
        warp_positions<kPatchN>(z, X, Y);
        APD_STAGE();
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            a[buf][j] = __builtin_amdgcn_fractf(X[j]);
            b[buf][j] = __builtin_amdgcn_fractf(Y[j]);
        }
        if (inside) {
#pragma unroll
            for (int j = 0; j < kPatchN; ++j) {
                t[buf][j] = bytewin_fetch(w.addr0f, X[j], Y[j], a[buf][j], b[buf][j]);
                if (j == kPatchN / 2 - 1) {
                    APD_STAGE();   // three fetches (twelve LDS dwords) in flight at a time
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kPatchN; ++j) {
                t[buf][j] = global_tap_fetch(srcq, qpitch, wm1, hm1, X[j], Y[j]);
            }
        }
    };
    issue(0, 0);
#pragma unroll
    for (int i = 0; i < kPatchN; ++i) {
        float ref[kPatchN];
#pragma unroll
        for (int j = 0; j < kPatchN; ++j) {
            ref[j] = rp.at(i, j);
        }
        if (i + 1 < kPatchN) {
            issue(i + 1, (i + 1) & 1);
        }
        APD_STAGE();
        float v[kPatchN];
        quad_row_lerp(t[i & 1], a[i & 1], b[i & 1], v);
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
}

// The cost of ncc_fixed_from_h<true> for an already projected centre (K9/K10's k == 0 term and its final re-score).
template <typename Ref>
__device__ __forceinline__ float ncc_fixed_bytewin_from_h(const FrameArgs &fa, const ViewConst &vc, const ByteWindow &w, const Ref &rp,
                                                          const Homography &H, int px, int py)
{
    const float kMinVar = 1e-5f;
    if (rp.var < kMinVar) {
        return 2.0f;
    }
    const float x0 = (float)(px - kPatchRadius), x1 = (float)(px + kPatchRadius);
    const float y0 = (float)(py - kPatchRadius), y1 = (float)(py + kPatchRadius);
    const bool fast_recip = denominators_fast(H, x0, x1, y0, y1);
    bool inside = false;
    if (fast_recip && w.valid) {
        float cX[4], cY[4];
        corner_position(H, x0, y0, cX[0], cY[0]);
        corner_position(H, x0, y1, cX[1], cY[1]);
        corner_position(H, x1, y0, cX[2], cY[2]);
        corner_position(H, x1, y1, cX[3], cY[3]);
        const float xl = fminf(fminf(cX[0], cX[1]), fminf(cX[2], cX[3])), xh = fmaxf(fmaxf(cX[0], cX[1]), fmaxf(cX[2], cX[3]));
        const float yl = fminf(fminf(cY[0], cY[1]), fminf(cY[2], cY[3])), yh = fmaxf(fmaxf(cY[0], cY[1]), fmaxf(cY[2], cY[3]));
        inside = xl >= w.lo_x && xh < w.hi_x && yl >= w.lo_y && yh < w.hi_y;
    }
    APD_LAB_NCC_STATS(inside, fast_recip);
    const global_quad_ptr srcq = (global_quad_ptr)vc.quad;
    const unsigned qpitch = quad_row_pitch_bytes(fa.W);
    // if one lane needs the IEEE division, every lane of the wave takes that body (same bits where both are valid)
    const bool fast_body = __builtin_amdgcn_ballot_w64(!fast_recip) == 0;
    float sum_s, sum_ss, sum_rs;
    if (__builtin_expect(fast_body, 1)) {
        ncc_bytewin_moments(rp, H, w, inside, srcq, qpitch, fa.W - 1, fa.H - 1, px, py, sum_s, sum_ss, sum_rs);
    } else {
        ncc_fixed_moments_ieee<true, false, Ref>(fa, vc, rp, H, px, py, sum_s, sum_ss, sum_rs);
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
