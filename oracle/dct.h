/*
 * ORACLE — test infrastructure only.  CPU restatement (plain C) of jxl-oxide's *generic scalar*
 * DCT code.  Nothing under jxl-oxide_amd/ may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it as the checker.
 *
 * Follows, function by function:
 *   dct4 / dct / dct_2d   jxl-render/src/vardct/generic/dct.rs:144-172, 174-293, 5-141
 *   sec_half / scale_f    jxl-render/src/vardct/dct_common.rs:10-70, 77-115
 *
 * Parity pin: the reference's own six 1-D DCT tests (dct.rs:299-435) are reproduced in
 * tests/test_oracle_dct.py against this file.  Everything else in oracle/ is "parity unpinned"
 * (no upstream vectors exist in this container) and is cross-checked against f64
 * analytic formulas instead.
 *
 * Compile with -ffp-contract=off: Rust never contracts a*b+c, and every `mul_add` in the
 * reference is an explicit fmaf() here.
 */
#ifndef JXL_ORACLE_DCT_H_
#define JXL_ORACLE_DCT_H_

#include <stddef.h>

#define ORC_FORWARD 1
#define ORC_INVERSE 0

/* sec_half(n)[k] = 1 / (2 cos((2k+1) pi / 2n)), n = 4..256; tables for n >= 64 are computed in
 * f32 exactly as dct_common.rs:56-66 does (cosf, recip, /2).  `orc_set_sec_half_large` lets a
 * test inject the tables it also hands to the device.                                          */
const float* orc_sec_half(size_t n);
void orc_set_sec_half_large(size_t n, const float* table);
float orc_scale_f(size_t c, size_t logb);

/* dct(): in-place 1-D DCT-II (forward, halving per stage) / DCT-III (inverse, unscaled). */
void orc_dct_1d(float* io, float* scratch, size_t n, int forward);
/* dct_2d(): `io` is a width x height window with row stride `stride`. */
void orc_dct_2d(float* io, size_t stride, size_t width, size_t height, int forward);

#endif
