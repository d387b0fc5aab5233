// xl_mixf_layout.h -- index bookkeeping of the mix launch on the matrix cores with FLOAT32 operands (xl_mixf32.hip:
// xlp_mix_f32_kernel, xlp_tables_f_kernel), kept apart from the kernels so that it also compiles for the host:
// tests/c/test_mixf_layout.cpp drives the same functions through an emulation of v_mfma_f32_32x32x2_f32's operand / result maps
// and checks the sums against plain complex arithmetic, without a GPU.
//
// One matrix instruction: D[32 rows][32 columns] += A[32 rows][2 k] * B[2 k][32 columns], float32 in, float32 accumulate -- the
// result is bit for bit a k-ordered chain of fmaf (cdna_hip_programming.md, "FP32-input MFMA"), i.e. the arithmetic of
// xlating.c:66-71's multiply-accumulate in float32, nothing split, nothing scaled.
//   operand registers  ONE float per lane: lane l holds A[row l & 31][k = l >> 5] / B[k = l >> 5][column l & 31]
//   result registers   lane (h, c) register g = row (g & 3) + 8 (g >> 2) + 4 h of column c   (the map every 32x32 form shares)
// The mix's use of it, per spectrum bin m (the sums Y[c][s][m] = sum_b X[s][b][m] R[c][b][m], xl_polyphase.h):
//   one instruction = ONE branch b:  k = 0: the "re" factor, k = 1: the "im" factor
//   rows   = (segment sl of the pass, component): row 2 sl + comp;  A row (sl, re) = (X.re, X.im), (sl, im) = (X.im, -X.re)
//   cols   = client columns, 32 per wave;                           B column      = (R.re, -R.im)
//   D[(sl, re)][c] = sum_b X.re R.re - X.im R.im      D[(sl, im)][c] = sum_b X.im R.re + X.re R.im
// The A operand needs no staging at all: the forward launch's image row X[pass][b][m][0..15] is 32 floats (16 segments x
// (re, im)) = the 32 rows of k = 0 in row order; the k = 1 half is the same 128 bytes with the floats of every pair swapped and
// the second one negated: lane (h, r) reads float r ^ h of the row and flips the sign when h & r & 1.
#ifndef XL_MIXF_LAYOUT_H_
#define XL_MIXF_LAYOUT_H_
#include <stddef.h>
#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define XLMF_FN static __host__ __device__ __forceinline__
#else
#define XLMF_FN static inline
#endif

#define XLMF_NB8_MAX 14u  // k-blocks of 8 branches whose B operands a wave keeps in registers for all its passes (D <= 112); longer
                          // branch lists stream their B operands per pass (xlp_mix_f32_stream_kernel)

// A operand of lane (h = lane >> 5, r = lane & 31): float index inside the 32-float image row, and whether its sign flips
XLMF_FN uint32_t xlmf_a_float(uint32_t lane) { return (lane & 31u) ^ (lane >> 5); }
XLMF_FN uint32_t xlmf_a_negate(uint32_t lane) { return (lane >> 5) & lane & 1u; }

// Operand-form image of the branch spectra: float4 slot of (column group cg, bin m, 32-column quarter w, k-block jb of 8 branches,
// half q of the block, lane); element e of the slot = branch b = 8 jb + 4 q + e; lane (h, c): h = 0: R.re, h = 1: -R.im of column
// 32 w + c.  A wave's operands of one (cg, m) are 2 nb8 consecutive 1 KB runs.
XLMF_FN size_t xlmf_rf_slot(uint32_t cg, uint32_t M, uint32_t m, uint32_t w, uint32_t nb8, uint32_t jb, uint32_t q, uint32_t lane) {
  return (((((size_t)cg * M + m) * 4u + w) * nb8 + jb) * 2u + q) * 64u + lane;
}
XLMF_FN size_t xlmf_rf_bytes_per_group(uint32_t M, uint32_t nb8) { return (size_t)M * 4u * nb8 * 2u * 64u * 16u; }
// where branch b of a column's spectra goes
XLMF_FN uint32_t xlmf_b_block(uint32_t b) { return b >> 3; }
XLMF_FN uint32_t xlmf_b_half(uint32_t b) { return (b >> 2) & 1u; }
XLMF_FN uint32_t xlmf_b_elem(uint32_t b) { return b & 3u; }

// result register g of a lane in half h: the row it holds (even g: the "re" row of segment row / 2, g + 1 its "im" row)
XLMF_FN uint32_t xlmf_result_row(uint32_t g, uint32_t h) { return (g & 3u) + 8u * (g >> 2) + 4u * h; }

#endif  // XL_MIXF_LAYOUT_H_
