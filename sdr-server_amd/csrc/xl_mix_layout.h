// xl_mix_layout.h -- index bookkeeping of the matrix-core mix (xl_polyphase.hip: xlp_mix_mfma_kernel, xlp_tables_h_kernel), kept
// apart from the kernels so that it also compiles for the host: tests/c/test_mix_layout.cpp drives the same functions
// through an emulation of v_mfma_f32_32x32x16_f16's operand / result maps and checks the sums against plain complex
// arithmetic, without a GPU.
//
// One matrix instruction: D[32 rows][32 columns] += A[32 rows][16 k] * B[16 k][32 columns].
//   operand registers  lane (h, i), h = lane >> 5, i = lane & 31, holds 8 of the 16 k-slots of row i (A) / column i (B): the
//                      slots "8 h .. 8 h + 7" -- which k each slot is does not matter to a dot product as long as A and B agree,
//                      and they do: both are laid out by the functions below
//   result registers   lane (h, c) register g = row (g & 3) + 8 (g >> 2) + 4 h of column c   (cdna_hip_programming.md, C/D map)
// The mix's use of it, per spectrum bin m:
//   k      = (branch b, re / im)      a k-block (one instruction) = 8 branches; branch b -> k-block b >> 3, half (b >> 2) & 1,
//                                     dword b & 3 of the lane's 16 bytes (low half-word: the "re" factor, high: the "im" factor)
//   rows   = (segment sl of the pass, component): row 2 sl + comp;  A row (sl, re) = (X.re, X.im), (sl, im) = (X.im, -X.re)
//   cols   = client columns, 32 per wave;                           B column      = (R.re, -R.im)
#ifndef XL_MIX_LAYOUT_H_
#define XL_MIX_LAYOUT_H_
#include <stddef.h>
#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define XLM_FN static __host__ __device__ __forceinline__
#else
#define XLM_FN static inline
#endif

XLM_FN uint32_t xlm_kblock(uint32_t b) { return b >> 3; }
XLM_FN uint32_t xlm_half(uint32_t b) { return (b >> 2) & 1u; }
XLM_FN uint32_t xlm_dword(uint32_t b) { return b & 3u; }
XLM_FN uint32_t xlm_lane(uint32_t h, uint32_t i) { return h * 32u + i; }
XLM_FN uint32_t xlm_row(uint32_t sl, uint32_t comp) { return 2u * sl + comp; }
// result register g of a lane in half h: the row it holds (even g: a "re" row, g + 1 the "im" row of the same segment)
XLM_FN uint32_t xlm_result_row(uint32_t g, uint32_t h) { return (g & 3u) + 8u * (g >> 2) + 4u * h; }

// Operand-form image of the branch spectra: 16-byte slot of (column group cg, bin m, 32-column quarter w, term, k-block kb, lane):
// a wave's operands of one (cg, m) -- 2 terms x nkb k-blocks x 64 lanes -- are 2 nkb consecutive 1 KB runs.
XLM_FN size_t xlm_rh_slot(uint32_t cg, uint32_t M, uint32_t m, uint32_t w, uint32_t term, uint32_t nkb, uint32_t kb, uint32_t lane) {
  return (((((size_t)cg * M + m) * 4u + w) * 2u + term) * nkb + kb) * 64u + lane;
}

// Where lane slot (h, row) of a staged A operand sits in LDS: the four slots of an aligned group of four rows are permuted by
// (h, row bit 4).  The staging writes are 4 bytes per lane -- lane (branch-in-block bb, segment pair sp) writes dword bb & 3 of
// slot (bb >> 2, 4 sp + r): dword address 128 h + 16 sp + 4 r + (bb & 3), i.e. 16 banks for 64 lanes -- and the two bits that (h,
// sp >> 2) would waste go into the slot's low bits instead: 64 lanes, 64 banks.  The 16-byte operand reads do not care (a group
// of four lanes still covers the same 64 bytes).
XLM_FN uint32_t xlm_lds_slot(uint32_t lane_slot) {
#ifdef XLM_NO_LDS_SWIZZLE  // (A/B builds only: tools/experiments/build_variant.sh)
  return lane_slot;
#else
  return lane_slot ^ (((lane_slot >> 5) & 1u) | (((lane_slot >> 4) & 1u) << 1));
#endif
}

// Staging role of a lane of wave w in round q: k-block w + 4 q, branch 8 (w + 4 q) + (lane >> 3), segments 2 (lane & 7) and + 1 of
// the pass (one 16-byte load of the FP32 image row X[pass][branch][m][0..15])
XLM_FN uint32_t xlm_stage_kblock(uint32_t w, uint32_t q) { return w + 4u * q; }
XLM_FN uint32_t xlm_stage_branch_in_block(uint32_t lane) { return lane >> 3; }
XLM_FN uint32_t xlm_stage_segment_pair(uint32_t lane) { return lane & 7u; }

#endif  // XL_MIX_LAYOUT_H_
