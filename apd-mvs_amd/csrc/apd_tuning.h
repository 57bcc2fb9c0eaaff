// apd_tuning.h -- every build-time parameter of the kernels in one place.
//
// Each value below is the product's setting; the comment says what it is and what it was measured against.  They are
// macros only so that tools/tune.sh can rebuild the library with another value (-DAPD_X=...) for an A/B run on the GPU box --
// nothing in the product sets them, and the library reads nothing from the environment.  Parameters that name a constant of
// a kernel file (kWaveH, kFwLds) are expanded where that file uses them.
#pragma once

// ---- apd_device.h ----
#ifndef APD_ROW_PREFETCH
#define APD_ROW_PREFETCH 1
#endif
#ifndef APD_IEEE_COMPACT
#define APD_IEEE_COMPACT 1
#endif

// ---- apd_sweep.h ----
#ifndef APD_CB_ROWS
#define APD_CB_ROWS 4
#endif
#ifndef APD_REFINE_EARLY_OUT
#define APD_REFINE_EARLY_OUT 1
#endif

// ---- apd_window.h ----
#ifndef APD_WIN_F32_PAIRS
#define APD_WIN_F32_PAIRS 0
#endif
#ifndef APD_WIN_ADDR_MAGIC
#define APD_WIN_ADDR_MAGIC 1
#endif
#ifndef APD_WIN_CORNER_REUSE
#define APD_WIN_CORNER_REUSE 1
#endif
#ifndef APD_WIN_DRAIN_SMEM
#define APD_WIN_DRAIN_SMEM 1  // s_waitcnt lgkmcnt(0) at the head of the window body (see ncc_window_moments)
#endif
#ifndef APD_WIN_SETPRIO
#define APD_WIN_SETPRIO 1  // issue priority for the wave inside its 36-sample burst: +0.7 % on configs[1] (0 = off)
#endif

// ---- apd_kernels.hip ----
#ifndef APD_FF_ROWS
#define APD_FF_ROWS 8
#endif
#ifndef APD_K67_WAVES
#define APD_K67_WAVES 4  // minimum waves per SIMD the register allocator must leave room for (ms per launch at 4096x3072 N=8 -- ref patch in registers: 2: 32.0, 3: 28.3; ref patch in LDS: 3: 28.2, 4: 27.4)
#endif

// ---- apd_kernels_k67w.hip ----
#ifndef APD_WIN_H
#define APD_WIN_H (kWaveH + 16)   // rows of fetch positions: footprint + 2 * (patch radius 5 + 3 texels of slack)
#endif
#ifndef APD_WIN_TRUST
#define APD_WIN_TRUST 0.5f
#endif
#ifndef APD_WIN_FROM_ITER
#define APD_WIN_FROM_ITER 1  // first iteration of a FIRST_INIT pass that stages windows; configs[1] Mpix*iter/s: 0: 207, 1: 216
#endif
#ifndef APD_K67W_WAVES
#define APD_K67W_WAVES 4
#endif
#ifndef APD_K67W_WAVES_F32
#define APD_K67W_WAVES_F32 3  // float windows: three waves per SIMD also with the single-texel entries (4 waves, 128 VGPRs: 32.4 against 29.1 ms
                              // for the first iteration at 2048x1536, 16.0 against 14.6 later)
#endif

// ---- apd_kernels_k1415w.hip ----
#ifndef APD_FW_TILE_PITCH
#define APD_FW_TILE_PITCH (kFwLds + 1)
#endif
#ifndef APD_K1415_WIN_PITCH
#define APD_K1415_WIN_PITCH 72  // entries per window row: a 32-lane group reads four rows of eight columns (apd_window.h)
#endif
#ifndef APD_K14_WIN_H
#define APD_K14_WIN_H 32  // rows of fetch positions: 8 + 2 * (patch radius 5 + 7 texels of slack)
#endif
#ifndef APD_K14_WIN_H_F32
#define APD_K14_WIN_H_F32 32
#endif
#ifndef APD_K14_COMPACT
#define APD_K14_COMPACT 1  // K14 may walk the (sample, lane) pairs of a chunk 64 at a time instead of one sample per wave-level NCC (0: never)
#endif
#ifndef APD_K14_PAIRS_FROM_N
#define APD_K14_PAIRS_FROM_N 10  // ... in launches with at least this many source views (float images)
#endif
#ifndef APD_K14_PAIRS_FROM_N_PHOTO
#define APD_K14_PAIRS_FROM_N_PHOTO 8   // ... 8-bit input, photometric passes
#endif
#ifndef APD_K14_PAIRS_FROM_N_GEOM
#define APD_K14_PAIRS_FROM_N_GEOM 2    // ... 8-bit input, passes with the geometric term
#endif
#ifndef APD_K14_PAIRS_MIN_SAVE
#define APD_K14_PAIRS_MIN_SAVE 2  // photometric passes: a chunk walks its (sample, lane) pairs when that saves at least this many wave-level NCCs (geometric: one)
#endif
#ifndef APD_K14_CHUNK
#define APD_K14_CHUNK 8  // depth samples per staged window = length of the register vector of cost sums: 4, 8 or 16 (K14 ms at 6200x4130, 10 views,
                         // photometric / geometric pass, profiles/r05/tune_k14.txt: 4: 316 / 276, 8: 286 / 275, 16: 377 / 448)
#endif
#ifndef APD_K14W_WAVES
#define APD_K14W_WAVES 4  // ms at 4096x3072, 8 views: 4 waves/SIMD (128 VGPRs, 18 spilled) 140.6, 3 waves 150.1
                          // the (sample, lane)-pair variant, 6200x4130, 10 views: 4 waves (42 spilled) 295.2 / 361.3 ms (photometric / geometric pass), 3 waves (164 VGPRs, none) 306.9 / 388.7 (profiles/r04/ab_k14_pairs_waves.txt)
#endif
#ifndef APD_K15W_WAVES
#define APD_K15W_WAVES 3  // 4 waves/SIMD (100 VGPRs spilled) 30.5, 3 waves 27.1
#endif
#ifndef APD_K1415W_WAVES_F32
#define APD_K1415W_WAVES_F32 3  // float windows (single-texel entries, 9.5 KB per wave like the 8-bit ones); ms at 2048x1536, 8 views, K14 / K15:
                                // 2 waves/SIMD 41.6 / 3.77, 3 waves 34.5 / 3.18, 4 waves 36.5 / 3.87 (8-byte pair entries, 2 waves: 40.4 / 3.80)
#endif
#ifndef APD_K14_CENTRE_FIRST
#define APD_K14_CENTRE_FIRST 1
#endif
#ifndef APD_K15_EARLY_OUT
#define APD_K15_EARLY_OUT 1
#endif

// ---- apd_kernels_weak.hip ----
#ifndef APD_K910_WINDOW
#define APD_K910_WINDOW 1  // 0: no centre-patch window (A/B runs)
#endif
#ifndef APD_K910_WIN_DIVERGENT
#define APD_K910_WIN_DIVERGENT 1  // 0: a wave with lanes outside the window takes the global path whole (A/B runs)
#endif
#ifndef APD_WEAK_SUPER_SHIFT
#define APD_WEAK_SUPER_SHIFT 4
#endif
#ifndef APD_K910_WIN_H
#define APD_K910_WIN_H 28
#endif
#ifndef APD_K910_WAVES
#define APD_K910_WAVES 2
#endif
#ifndef APD_K910_SUBPATCH_TILED
#define APD_K910_SUBPATCH_TILED 0  // 1 (A/B runs, with --opt tiled_copy=2): the 3 x 3 stride-5 sub-patch taps gather from the 7 x 8 pair tiles instead
                                   // of the row-major pairs.  Measured in round 5 (profiles/r05/ab_k910_tiled.txt), see DESIGN.md section 6
#endif
#ifndef APD_K910_REMAP
#define APD_K910_REMAP 1  // propagation phase: the sub-patches of the eight neighbour hypotheses with lane = (pixel, hypothesis) instead of lane = pixel
                          // (apd_kernels_weak.hip: k910_update_weak); 0: rounds 1-5
#endif
#ifndef APD_K910_COMPACT_REFINE
#define APD_K910_COMPACT_REFINE 1
#endif

