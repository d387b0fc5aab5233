// xl_inv32_layout.h -- index bookkeeping of the "32 x 4" inverse launch of the polyphase path (xl_inv32.hip:
// xlp_inverse32_kernel), kept apart from the kernel so that it also compiles for the host: tests/c/test_inv32_layout.cpp runs the
// same functions through an emulation of the lanes' data flow (checked against a plain double-precision DFT) and through a model of
// the LDS banks (MI355X_MICROARCH.md, LDS: lane groups and bank widths per instruction) -- without a GPU.
//
// Why a third cut of the same transform (round 5).  As a memory kernel the inverse launch is two patterns: tile loads
// and piece stores.  The LDS transform (xl_polyphase.hip) has the good patterns -- whole 128-byte lines in, 256 consecutive bytes
// of a client row per store instruction out: 43.6 us per block at 4096 clients when it does nothing else (tools/ubench_tile_copy.hip)
// -- and 1.5 x the vector instructions; the 8-lane kernel (xl_inv8.hip) has the lean arithmetic and 64-byte runs both ways (58 us
// for its traffic alone).  This cut has both:
//   m = m1 + 4 m2 (m1 < 4, m2 < 32),  n = t + 32 g (t < 32, g < 4):   w^{m n} = W4^{g m1} * w^{m1 t} * W32^{m2 t},  w = e^{+2 pi j / 128}
//   1. lane (m1, c) of a wave (c < 16: the wave owns 16 client columns = one 128-byte line of every bin row) holds
//      Y[m1 + 4 m2], m2 < 32 -- a load instruction reads four whole lines (bins 4 m2 .. 4 m2 + 3) -- and runs a 32-point inverse
//      transform over m2 in registers: Z_m1[t]          (radix 4, 4, 2 with compile-time twiddles: xl_fft16.h)
//   2. Z'_m1[t] = Z_m1[t] * w^{m1 t} / 128              (the lane's 32 factors from a 1 KB table in LDS, [t][m1]: one address per
//                                                        lane, t in the instruction's offset field; the exact scaling rides along)
//   3. one exchange through the wave's own LDS region, [column][m1][t], in TWO ROUNDS of 8 columns (a region of 8.5 KB: LDS for
//      sixteen waves per CU; the kernel's registers allow twelve): the writer lanes of the round's columns store (t, t + 1) pairs (ds_write_b128), reader lane (cc, t) of pass
//      k < 4 fetches row t of column 8 r + 2 k + cc (4 x ds_read_b64) -- every address = lane base + immediate
//   4. a 4-point inverse transform over m1 in registers: y[t + 32 g], g < 4 -- a store instruction (fixed pass, g) covers 32
//      consecutive outputs of each of two columns
// The phases of a round's epilogue are expanded into the same region afterwards ([column][point], the 8-lane kernel's layout), by
// the lanes in a third role: lane (column, gq) walks the 16 phases behind one table entry.  What a column's stores need (row
// offset, grid shift, output count) sits in a 256-byte table beside it, written by the lanes that computed it for the walk.
// A wave shares nothing with the other waves of its workgroup: no workgroup barrier anywhere.
#ifndef XL_INV32_LAYOUT_H_
#define XL_INV32_LAYOUT_H_
#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define XLI32_FN static __host__ __device__ inline __attribute__((always_inline))
#else
#define XLI32_FN static inline
#endif

#define XLI32_COLS 16u  // client columns per wave: half a tile row

// roles 1, 2, writer side of 3: lane j = (m1, c)
XLI32_FN uint32_t xli32_load_m1(uint32_t j) { return j >> 4; }
XLI32_FN uint32_t xli32_load_c(uint32_t j) { return j & 15u; }
// offset (8-byte units) of role-1 lane j's m2-th value inside the 128 x 32 tile; `half` = which 16 columns
XLI32_FN uint32_t xli32_load(uint32_t half, uint32_t j, uint32_t m2) { return ((j >> 4) + 4u * m2) * 32u + 16u * half + (j & 15u); }
// reader side of 3, role 4, epilogue: lane j = (cc, t), round r, pass k: column 8 r + 2 k + cc
XLI32_FN uint32_t xli32_cc(uint32_t j) { return j >> 5; }
XLI32_FN uint32_t xli32_t(uint32_t j) { return j & 31u; }
// phase expansion: lane j = (column 8 r + (j >> 3), gq = j & 7) in round r
XLI32_FN uint32_t xli32_walk_c8(uint32_t j) { return j >> 3; }
XLI32_FN uint32_t xli32_walk_gq(uint32_t j) { return j & 7u; }

// Exchange region, byte addresses (c8 = column within the round): per column four rows (m1) of 32 values; column pitch 1040 =
// 65 x 16: the eight lanes of a ds_write_b128 group (8 columns, one m1) hit every bank once, the 32 lanes of a ds_read_b64 group
// read 256 contiguous bytes.
#define XLI32_XROW 256u
#define XLI32_XCOL 1040u
XLI32_FN uint32_t xli32_exch(uint32_t c8, uint32_t m1, uint32_t t) { return c8 * XLI32_XCOL + m1 * XLI32_XROW + t * 8u; }
// Phase region (the same memory, afterwards): shared point p = 16 gq + i of column c8 (the 8-lane kernel's pitches: the sixteen
// lanes of a ds_write_b64 group = 2 columns x 8 gq hit every bank once; a reader group of 32 lanes spans two rows 136 bytes apart,
// which meet in two banks: one extra LDS cycle per read)
#define XLI32_PROW 136u
#define XLI32_PCOL 1088u
XLI32_FN uint32_t xli32_phase(uint32_t c8, uint32_t p) { return c8 * XLI32_PCOL + (p >> 4) * XLI32_PROW + (p & 15u) * 8u; }
#define XLI32_REGION (8u * XLI32_PCOL)  // 8704 >= 8 * XLI32_XCOL = 8320
// behind it: the factor table [t][m1] (1 KB) and the columns' store records (16 bytes each)
#define XLI32_TW XLI32_REGION
XLI32_FN uint32_t xli32_tw(uint32_t t, uint32_t m1) { return XLI32_TW + t * 32u + m1 * 8u; }
#define XLI32_META (XLI32_TW + 1024u)
XLI32_FN uint32_t xli32_meta(uint32_t c) { return XLI32_META + c * 16u; }
#define XLI32_WAVE_BYTES (XLI32_META + 16u * XLI32_COLS)  // 9984: sixteen waves' worth per CU

// slot of output t of the in-place 32-point register transform (xl_fft32_inverse, xl_fft16.h): radix 4 (span 8), radix 4 (span 2),
// radix 2
XLI32_FN constexpr int xli32_slot32(int t) { return 8 * (t & 3) + 2 * ((t >> 2) & 3) + (t >> 4); }

#endif  // XL_INV32_LAYOUT_H_
