// xl_fused_layout.h -- index bookkeeping of the FUSED mix + inverse launch of the polyphase path (xl_fused.hip:
// xlp_forward_h_kernel, xlp_tables_h16_kernel, xlp_fused_kernel), kept apart from the kernels so that it also compiles for
// the host: tests/c/test_fused_layout.cpp drives the same functions through an emulation of v_mfma_f32_16x16x32_f16's operand
// and result maps, the 4 x 32 split of the 128-point inverse transform and the exchange between the four waves, and checks
// the result against plain complex arithmetic -- without a GPU.
//
// One matrix instruction: D[16 rows][16 columns] += A[16 rows][32 k] * B[32 k][16 columns].
//   operand registers  lane (kg, i), kg = lane >> 4, i = lane & 15, holds 8 of the 32 k-slots of row i (A) / column i (B): the
//                      slots "8 kg .. 8 kg + 7" -- which k each slot is does not matter to a dot product as long as A and B
//                      agree, and they do: both images are laid out by the functions below
//   result registers   lane (g, c), g = lane >> 4, c = lane & 15, register e = row 4 g + e of column c   (C/D map of the guide)
// The fused launch's use of it, per spectrum bin m (Y[c][s][m] = sum_b X[s][b][m] R[c][b][m], xlating.c:66-69 in the frequency
// domain of the D polyphase branches):
//   k      = (branch b, re / im)      a k-block (one instruction) = 16 branches; branch b -> k-block b >> 4, lane group
//                                     kg = (b >> 2) & 3, dword b & 3 of the lane's 16 bytes (low half-word: first factor)
//   rows   = 8 segments x (re, im)    A row (s, re) = (X.re, X.im), (s, im) = (X.im, -X.re): the image holds the first form only,
//                                     the lanes of the second read the same slot and cross the halves of every dword with one sign
//                                     flipped (one packed half-precision multiply per dword)
//   cols   = 16 client columns        B column = (R.re, -R.im)
//   D[(s, re)][c] = sum_b X.re R.re - X.im R.im      D[(s, im)][c] = sum_b X.im R.re + X.re R.im
// Row r of the instruction: g = r >> 2, e = r & 3 -> segment 4 (e >> 1) + g of the tile's eight, component e & 1 -- so that a
// result lane (g, c) holds in registers (0, 1) the complex value of segment g and in (2, 3) that of segment 4 + g: "half" h of
// the epilogue (registers 2 h, 2 h + 1 of all lanes) covers the four CONSECUTIVE segments 4 h .. 4 h + 3, the lanes 0..31 of a
// wave the first two of them, 32..63 the other two.
#ifndef XL_FUSED_LAYOUT_H_
#define XL_FUSED_LAYOUT_H_
#include <stddef.h>
#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define XLF_FN static __host__ __device__ __forceinline__
#else
#define XLF_FN static inline
#endif

#define XLF_M 128u    // transform length of a fused class
#define XLF_SEGS 8u   // segments per tile (x re / im = the 16 rows of the matrix instruction)
#define XLF_COLS 16u  // client columns per tile (columns of the matrix instruction)
#define XLF_NK_MAX 4u // at most 4 k-blocks of 16 branches (D <= 64)

XLF_FN uint32_t xlf_nk(uint32_t D) { return (D + 15u) >> 4; }
XLF_FN uint32_t xlf_kblock(uint32_t b) { return b >> 4; }
XLF_FN uint32_t xlf_kgroup(uint32_t b) { return (b >> 2) & 3u; }
XLF_FN uint32_t xlf_dword(uint32_t b) { return b & 3u; }
// segment (of the tile's eight) and component that row `row` of the matrix instruction carries
XLF_FN uint32_t xlf_row_seg(uint32_t row) { return 4u * ((row & 3u) >> 1) + (row >> 2); }
XLF_FN uint32_t xlf_row_comp(uint32_t row) { return row & 1u; }

// Operand-form image of the shared spectra ("Xh"), 16-byte slots: (segment group sg = segment / 8, k-block j, lane group kg, bin
// m, term, segment-in-group seg8).  One slot = the 4 branches 16 j + 4 kg .. + 3 of (segment, bin): (re, im) halves of
// X * XLP_H_XSCALE, term 0 = first halves, 1 = second.  An A-operand load of one (j, term) covers, per lane group, the 8 segments'
// slots = one 128-byte line (two lanes per slot); the forward launch writes exactly such lines.
XLF_FN size_t xlf_xh_slot(uint32_t sg, uint32_t nk, uint32_t j, uint32_t kg, uint32_t m, uint32_t term, uint32_t seg8) {
  return (((((size_t)sg * nk + j) * 4u + kg) * XLF_M + m) * 2u + term) * XLF_SEGS + seg8;
}
XLF_FN size_t xlf_xh_slots(uint32_t nsg, uint32_t nk) { return (size_t)nsg * nk * 4u * XLF_M * 2u * XLF_SEGS; }

// Operand-form image of the branch spectra ("Rh16"), 16-byte slots: (16-column group cg16, bin m, k-block j, term, lane): the
// operands of one (cg16, m) -- nk k-blocks x 2 terms x 64 lanes -- are 2 nk consecutive 1 KB runs.  Lane (kg, c) holds the
// branches 16 j + 4 kg .. + 3 of column c: (R.re, -R.im) * column scale, as halves.
XLF_FN size_t xlf_rh_slot(uint32_t cg16, uint32_t nk, uint32_t m, uint32_t j, uint32_t term, uint32_t lane) {
  return ((((size_t)cg16 * XLF_M + m) * nk + j) * 2u + term) * 64u + lane;
}
XLF_FN size_t xlf_rh_bytes_per_cg16(uint32_t nk) { return (size_t)XLF_M * nk * 2u * 64u * 16u; }
XLF_FN uint32_t xlf_lane(uint32_t kg, uint32_t i) { return kg * 16u + i; }

// The 128-point inverse transform of one (segment, client) sequence is split 4 x 32: wave w of the workgroup owns the bins
// m = 4 i + w, i < 32; a lane runs the 32-point inverse transform of its own bins in registers (xl_fft64.h: output k in slot
// xl_fft32_slot(k)), multiplies by e^{+2 pi j w k / 128} and hands Z'_w[k] to the exchange buffer; the consumer forms
//   y[k + 32 q] = sum_w j^{w q} Z'_w[k]                                          (a radix-4 butterfly without twiddles)
XLF_FN uint32_t xlf_bin(uint32_t w, uint32_t i) { return 4u * i + w; }
// Exchange buffer (LDS), 8-byte elements, one SUB-STEP of the epilogue at a time (two segments x 16 columns = the lanes 0..31
// or 32..63 of every wave): producer wave w, pair p = lane & 31 = 16 (segment of the two) + column, point k.  The XOR makes the
// producers' writes (32 lanes, one k, rows of 256 bytes) and the consumers' reads (32 lanes = 32 k of one pair) hit 32 distinct
// 8-byte bank pairs.
XLF_FN uint32_t xlf_exch(uint32_t w, uint32_t p, uint32_t k) { return (w * 32u + p) * 32u + (k ^ p); }
// Phase staging rows (LDS, 8-byte elements), one per client column: 2 segments x V <= 254 outputs + alignment to the table
// stride = at most 18 table entries of 16; row pitch = 1 mod 32 elements, so that the 32 lanes of a store -- 16 columns x 2
// table entries, one phase index -- hit 32 distinct bank pairs.
#define XLF_PH_ENT 18u
#define XLF_PH_ROW 289u

#endif  // XL_FUSED_LAYOUT_H_
