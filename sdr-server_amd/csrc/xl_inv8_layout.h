// xl_inv8_layout.h -- index bookkeeping of the "8 lanes per column" inverse launch of the polyphase path (xl_inv8.hip:
// xlp_inverse8_kernel), kept apart from the kernel so that it also compiles for the
// host: tests/c/test_inv8_layout.cpp runs the same functions through an emulation of the lanes' data flow (checked against a
// plain double-precision DFT) and through a model of the LDS banks (MI355X_MICROARCH.md, LDS: lane groups and bank widths per
// instruction) -- without a GPU.
//
// The 128-point inverse transform of one (segment, client column) sequence, y[n] = sum_m Y[m] w^{m n}, w = e^{+2 pi j / 128},
// split 16 x 8 so that NO pass of it goes through LDS addresses that have to be computed:
//   m = m1 + 8 m2 (m1 < 8, m2 < 16),  n = 16 g + t (g < 8, t < 16):   w^{m n} = W8^{g m1} * w^{m1 t} * W16^{m2 t}
//   1. lane (column c8, m1) of a wave holds Y[m1 + 8 m2], m2 < 16, and runs a 16-point inverse transform over m2 in registers:
//      Z_m1[t]                                          (two radix-4 stages, compile-time twiddles: xl_fft64.h)
//   2. Z'_m1[t] = Z_m1[t] * w^{m1 t}                    (the lane's 15 twiddles from a 1 KB table in LDS, [t][m1]: one
//                                                        address per lane, t in the instruction's offset field)
//   3. one exchange through LDS, [column][t][m1] with every address = lane base + immediate: writer lane (m1, c8) stores its
//      16 values (ds_write_b64), reader lane (c8, u) fetches the two rows t = u, u + 8 (4 x ds_read_b128 each)
//   4. lane (c8, u) runs two 8-point inverse transforms over m1 in registers: y[16 g + t], g < 8, for t = u and u + 8 --
//      a store instruction (fixed g, t-half) covers 8 consecutive outputs per column
// The phases of the epilogue are expanded into the same LDS region afterwards ([column][g][t], xl_inv8_phase), by the lanes
// in their third role: lane (column, g) walks the 16 phases of the shared points 16 g .. 16 g + 15 (one table entry per lane).
// One column per lane in roles 3 / 4 / epilogue: one XlpCol record per lane instead of five.
//
// The tile is read in the order every inverse launch shares ([bin][32 columns], 8-byte values): role-1 lane (m1, c8) of wave w
// loads element (m1 + 8 m2) * 32 + 8 w + c8, m2 < 16 -- per instruction eight 64-byte runs, the neighbour wave taking the other
// halves of the same 128-byte lines.
#ifndef XL_INV8_LAYOUT_H_
#define XL_INV8_LAYOUT_H_
#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define XLI8_FN static __host__ __device__ inline __attribute__((always_inline))
#else
#define XLI8_FN static inline
#endif

// offset (8-byte units) of role-1 lane j's m2-th value inside the 128 x 32 tile, wave w
XLI8_FN uint32_t xli8_load(uint32_t w, uint32_t j, uint32_t m2) { return ((j >> 3) + 8u * m2) * 32u + 8u * w + (j & 7u); }

// lane roles (j = lane of the wave)
XLI8_FN uint32_t xli8_load_m1(uint32_t j) { return j >> 3; }  // roles 1, 2 and the writer side of 3: (m1, c8)
XLI8_FN uint32_t xli8_load_c8(uint32_t j) { return j & 7u; }
XLI8_FN uint32_t xli8_col(uint32_t j) { return j >> 3; }      // reader side of 3, role 4, phase expansion, epilogue: (c8, u)
XLI8_FN uint32_t xli8_u(uint32_t j) { return j & 7u; }

// Exchange region of a wave, byte addresses: rows of 8 values (64 bytes, dense), 16 rows per column, column pitch 1040 --
// the writers' sixteen lanes of a ds_write_b64 group (2 m1 x 8 columns) and the readers' sixteen of a ds_read_b128 group
// hit every bank once (checked by the host test against the guide's lane groups).
#define XLI8_XROW 64u
#define XLI8_XCOL 1040u
XLI8_FN uint32_t xli8_exch(uint32_t c8, uint32_t t, uint32_t m1) { return c8 * XLI8_XCOL + t * XLI8_XROW + m1 * 8u; }
// Phase region (the same memory, afterwards): shared point p = 16 g + i of column c8
#define XLI8_PROW 136u
#define XLI8_PCOL 1088u
XLI8_FN uint32_t xli8_phase(uint32_t c8, uint32_t p) { return c8 * XLI8_PCOL + (p >> 4) * XLI8_PROW + (p & 15u) * 8u; }
#define XLI8_WAVE_BYTES (8u * XLI8_PCOL)  // 8704 >= 8 * XLI8_XCOL = 8320

// slots of the in-place register transforms (xl_fft64.h): output t of the 16-point one, output g of the 8-point one
XLI8_FN constexpr int xli8_slot16(int t) { return 4 * (t & 3) + (t >> 2); }
XLI8_FN constexpr int xli8_slot8(int g) { return 2 * (g & 3) + (g >> 2); }

#endif  // XL_INV8_LAYOUT_H_
