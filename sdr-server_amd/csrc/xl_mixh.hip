// xl_mixh.hip -- the two-half matrix-core mix launch of the polyphase overlap-save path for classes of up to 8 k-blocks of 8 branches
// (D <= 64: the server default, D = 42, is 6), and the mix launcher.  See xl_polyphase.h for the algebra, xl_polyphase.hip for the other
// launches of a call, xl_mixh2.hip for 9 .. 14 k-blocks, xl_mixf32.hip for float32 operands.
//
// A FILE OF ITS OWN SINCE ROUND 6, because of how it must be compiled: -fno-slp-vectorize (csrc/Makefile: MIX_FLAGS).  On this chip a
// wave's PACKED FP32 result (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 -- what the SLP vectoriser makes of any two adjacent float
// operations) loses its fourth quarter-wave, lanes 48..63, once in a while when matrix instructions are co-resident on the SIMD: pinned
// for the NCO role's packed recurrence steps in rounds 3-4 (DESIGN_HISTORY.md 3.6: single-lane-width FP32 is immune), met again in
// round 6 in a mix kernel's own epilogue (`y = sum x scale` as v_pk_mul_f32 right behind the products, beside the SIMD's other wave
// still issuing them: 6 of 10 fresh engines wrong in rows 4, 5 x columns 16..31 of a few tiles, 0 of 10 with single v_mul_f32:
// profiles/r06_mix_wide_kmajor_wrong_sums.txt (11)).  The three mix files are therefore compiled without the vectoriser, and
// tests/test_mix_no_packed_fp32.py checks their code objects.  (Writing the arithmetic as inline asm instead hides the consumers of
// the matrix results from the compiler's hazard padding: 16 GPU tests failed by 2-4e-5 on the last pass of small calls.)
#include "xl_polyphase.h"

#include "xl_poly_dev.h"
#include "xl_mix_layout.h"

#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------- mix on the matrix cores
// The same sums as xlp_mix_kernel, Y[c][s][m] = sum_b X[s][b][m] R[c][b][m], as one real matrix product per bin m:
//
//   rows   i = (segment s, component re / im)        A[i][k]   k = 2 b + {0, 1}:   re row: ( X.re, X.im )   im row: ( X.im, -X.re )
//   cols   j = client column c                       B[k][j]                       ( R.re, -R.im )
//   D[(s, re)][c] = sum_b X.re R.re - X.im R.im      D[(s, im)][c] = sum_b X.im R.re + X.re R.im
//
// on v_mfma_f32_32x32x16_f16 (32 rows = the 14 segments of a pass + 2 idle, 32 columns, 16 k = 8 branches per instruction; FP32
// accumulation) with every float32 operand v carried as TWO halves, v * scale = h1 + h2 + O(2^-22 |v|):
//
//   X R ~ (X1 R1) + (X1 R2 + X2 R1)        three matrix instructions per k-block; the dropped X2 R2 is 2^-22 relative.
//
// Measured against the oracle this is as good as the FP32 FMA chain of xlp_mix_kernel (CPU model of the arithmetic:
// 8e-8 of max|y| against 1.9e-7 for the chain: the products are exact in FP32 and the small terms are summed on their own) --
// and costs 12 half-precision MACs per complex MAC on units 16 x faster than the packed FP32 FMAs the other kernel saturates.
// The scales are powers of two: XLP_H_XSCALE for the spectra of the INTEGER input formats (bounded: |X| <= M sqrt 2; a cf32
// stream has no bound, its classes keep xlp_mix_kernel) and per column the one that brings the bound of its branch spectra
// under XLP_H_RMAX (xl_batch.cpp); the sums are multiplied by 1 / (both) before they are stored.  Halves in the subnormal
// range only ever carry 2^-24 of the operand scale.
//
// Workgroup = 4 waves = (bin m, column group of 128 clients, a run of `pp` passes); wave w = the group's columns
// 32 w .. 32 w + 31.  A wave keeps its B operands -- 2 terms x nkb k-blocks x 16 bytes per lane, read ONCE as whole 1 KB runs
// from the operand-form image Rh -- in registers for all its passes.  Per pass the workgroup stages the bin's rows of the
// shared spectra (the FP32 image the forward launch wrote: 128-byte rows X[pass][b][m][0..15]) into LDS in A-operand order,
// scaled and split: wave w converts k-blocks w, w + 4, ..; lane = (branch of the block, pair of segments), one 16-byte load.
// The next pass's rows are requested before this pass's products.  Lane (h, i) of an operand holds k = 8 h .. 8 h + 7 of
// the k-block, A and B alike -- whatever the hardware's assignment of those eight slots to k is, it is the same for both
// operands, which is all a dot product needs.  D: lane (h, c), register g = row (g & 3) + 8 (g >> 2) + 4 h, column c.
// Built for 4 waves per SIMD WITHOUT accumulation registers (124 VGPRs, the products land in VGPRs).  This launch never carries
// the NCO role (a slice of the next call's phase recurrence, xlp_nco_role): the first build of this kernel (128 VGPRs + 32
// AGPRs, 3 waves per SIMD) did something no other kernel of this library has shown -- the phases of lanes 48..63 of a random
// role wave riding in its launch came out wrong from some step on (20 % of 1024 clients hit within 120 one-block calls; the
// role's instructions AND registers identical in the failing and the passing builds; git 7991f21 reproduces it).  The cause
// was never found, so the combination was designed out (round 4): the recurrence rides in the forward and inverse launches or
// runs on the side stream (xl_batch.cpp), xlp_launch_mix refuses a role for this kernel, and
// tests/test_batch_gpu.py::test_matrix_core_mix_role_phases_bit_exact keeps comparing all phases of two engines bit for bit.
// Round 6: (1) up to XLP_NKB_MAX = 14 k-blocks (D <= 112): above XLP_NKB_4W the B operands (8 NKB registers) take the kernel to a
// two-waves-per-SIMD budget -- the launch is bound by its operand and Y streams either way; (2) SEG: cf32 streams, whose spectra have no
// a-priori bound, are scaled per SEGMENT -- rows of the per-bin product are (segment, re / im), so a power-of-two row scale factors out
// of the sums exactly: the forward launch leaves every segment's largest spectrum component in XlpArgs::segmax, the staging multiplies
// the segment's rows by 2^(14 - floor(log2 max)) (every scaled component < 2^15), and the epilogue multiplies the segment's sums by the
// inverse.  The float32 matrix instruction (xl_mixf32.hip) remains for D > 112 and as the exact-float32 option (mix_kernel = 3).
// waves per SIMD: 4 up to 6 k-blocks (122-126 VGPRs; with the segment scales up to 4), 3 up to XLP_NKB_4W = 8 (no spills: at 4 waves
// 7 / 8 k-blocks spilled 10 / 42 registers); wider classes: xlp_mix_mfma_wide_kernel (xl_mixh2.hip), 2 waves
constexpr int xlp_mix_waves(const int nkb, const bool seg) { return nkb > (seg ? 4 : 6) ? 3 : 4; }

template <int NKB, bool SEG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(xlp_mix_waves(NKB, SEG), xlp_mix_waves(NKB, SEG))))
void xlp_mix_mfma_kernel(const XlpArgs a) {
  static_assert(NKB <= (int)XLP_NKB_4W, "wider classes: xl_mixh2.hip");
  // A operands of one pass: [term][k-block][lane][8 halves]; two buffers (one barrier per pass: a buffer is rewritten two
  // barriers after it was read)
  __shared__ uint4 xs[2][2][NKB][64];
  __shared__ float sinv[2][XLP_SEG];  // SEG: what undoes the segments' scales, per buffer
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;
  const uint32_t bid = blockIdx.x;
  const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
  const uint32_t M = a.M;
  // bin, column group and pass run of this workgroup: XCD-aware (xlp_mix_place)
  const uint32_t pp = a.mix_pp, runs = (a.mix_passes + pp - 1u) / pp;
  uint32_t m, cg, run;
  xlp_mix_place(bid, M, runs, m, cg, run);
  const uint32_t p0 = run * pp, p1 = p0 + pp < a.mix_passes ? p0 + pp : a.mix_passes;
  if (p0 >= p1) return;
  const uint32_t h = lane >> 5, c = lane & 31u;
  const float cs_ = a.cscale[cg * XLP_COLS + w * 32u + c];
  // ---- staging role of this lane: branch 8 j + bb of k-block j = w + 4 round, segments 2 sp, 2 sp + 1 of the pass
  constexpr int ROUNDS = (NKB + 3) / 4;
  const uint32_t bb = xlm_stage_branch_in_block(lane), sp = xlm_stage_segment_pair(lane);
  const v4f *__restrict__ Xm = reinterpret_cast<const v4f *>(a.X) + (size_t)m * (XLP_XS / 2u) + sp;
  const size_t xrow = (size_t)M * (XLP_XS / 2u);  // v4f per branch row
  v4f g[ROUNDS];
  uint32_t smx[2] = {0u, 0u};  // SEG: the largest components of this lane's two segments of the requested pass
  const uint32_t *__restrict__ segmax = SEG ? a.segmax + ((size_t)a.seg_par * a.seg_cap + 2u * sp) * XLP_SEGMAX_STRIDE : nullptr;
  auto request = [&](const uint32_t pass) __attribute__((always_inline)) {
    if (SEG) smx[0] = segmax[(size_t)pass * XLP_SEG * XLP_SEGMAX_STRIDE], smx[1] = segmax[((size_t)pass * XLP_SEG + 1u) * XLP_SEGMAX_STRIDE];
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t b = 8u * xlm_stage_kblock(w, (uint32_t)q) + bb;
      // (rows D .. Dpad - 1 of the image are zeros; beyond Dpad there is nothing to read)
      g[q] = (xlm_stage_kblock(w, (uint32_t)q) < (uint32_t)NKB && b < a.D) ? Xm[((size_t)pass * a.Dpad + b) * xrow] : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto stage = [&](const uint32_t buf) __attribute__((always_inline)) {
    const float sx0 = SEG ? xlp_seg_scale(smx[0]) : XLP_H_XSCALE, sx1 = SEG ? xlp_seg_scale(smx[1]) : XLP_H_XSCALE;
    if (SEG && tid < 8u) sinv[buf][2u * sp] = xlp_seg_unscale(smx[0]), sinv[buf][2u * sp + 1u] = xlp_seg_unscale(smx[1]);
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t j = xlm_stage_kblock(w, (uint32_t)q);
      if (j < (uint32_t)NKB) {  // (wave-uniform)
        _Float16 f1[4], f2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xlp_split_h(g[q][e] * (e < 2 ? sx0 : sx1), f1[e], f2[e]);
        // branch bb of the k-block: dword xlm_dword(bb) of the lane slots (half xlm_half(bb), row) of its two segments' rows
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // segment 2 sp + u: (re, im) = f[2 u], f[2 u + 1]
          const uint32_t sre = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(2u * sp + (uint32_t)u, 0u)));
          const uint32_t sim = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(2u * sp + (uint32_t)u, 1u)));
          reinterpret_cast<uint32_t *>(&xs[buf][0][j][sre])[xlm_dword(bb)] = xlp_pack_h(f1[2 * u], f1[2 * u + 1]);
          reinterpret_cast<uint32_t *>(&xs[buf][0][j][sim])[xlm_dword(bb)] = xlp_pack_h(f1[2 * u + 1], -f1[2 * u]);
          reinterpret_cast<uint32_t *>(&xs[buf][1][j][sre])[xlm_dword(bb)] = xlp_pack_h(f2[2 * u], f2[2 * u + 1]);
          reinterpret_cast<uint32_t *>(&xs[buf][1][j][sim])[xlm_dword(bb)] = xlp_pack_h(f2[2 * u + 1], -f2[2 * u]);
        }
      }
    }
  };
  // ---- Y image [cg][segment][sub][bin][CW columns] (the inverse workgroups' tiles): this lane's column of segment s
  const uint32_t CW = xlp_tile_columns(M), NSUB = XLP_COLS / CW;
  const uint32_t col = w * 32u + c;
  v2f *__restrict__ Yc = reinterpret_cast<v2f *>(a.Y) +
                         ((((size_t)cg * a.nseg_cap) * NSUB + col / CW) * M + m) * CW + col % CW;
  const size_t ystride = (size_t)NSUB * M * CW;  // v2f per segment
  // Software pipeline: the rows of pass p + 1 are converted into the other buffer AFTER pass p's products and BEFORE its stores
  // -- the wait for those rows (vmcnt counts loads and stores alike, and the two complete out of order: the only safe wait is
  // "all") then finds nothing younger than the stores of pass p - 1, a whole pass old.  Waiting with pass p's stores just
  // issued made every pass sit out a write latency.
  request(p0);
  // ---- B operands of this wave: 2 NKB runs of 1 KB -- requested BEHIND the first pass's rows, so that staging those rows is not a wait
  // for the operands (loads return in order), and the first pass's products start as the operands arrive (round 6; as xlp_mix_f32_kernel)
  const uint4 *__restrict__ Rp = reinterpret_cast<const uint4 *>(a.Rh);
  v8h r1[NKB], r2[NKB];
#pragma unroll
  for (int j = 0; j < NKB; ++j) {
#ifdef XLP_MIX_EXP_NOOPERANDS
    r1[j] = __builtin_bit_cast(v8h, (uint4){lane, tid, (uint32_t)j, m});
    r2[j] = __builtin_bit_cast(v8h, (uint4){m, lane, tid, (uint32_t)j});
#else
    r1[j] = __builtin_bit_cast(v8h, Rp[xlm_rh_slot(cg, M, m, w, 0u, NKB, (uint32_t)j, lane)]);
    r2[j] = __builtin_bit_cast(v8h, Rp[xlm_rh_slot(cg, M, m, w, 1u, NKB, (uint32_t)j, lane)]);
#endif
  }
  stage(0u);
  if (p0 + 1u < p1) request(p0 + 1u);
  // (LDS hand-offs only: __syncthreads() would also wait for every load and store in flight -- the operands, the next rows, the pass's
  // stores)
  xlp_lds_barrier();
  auto products = [&](const uint32_t pass) __attribute__((always_inline)) {
    const uint32_t buf = (pass - p0) & 1u;
    v16f32 hi, lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) hi[i] = lo[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const v8h a1 = __builtin_bit_cast(v8h, xs[buf][0][j][xlm_lds_slot(lane)]);
      const v8h a2 = __builtin_bit_cast(v8h, xs[buf][1][j][xlm_lds_slot(lane)]);
#ifdef XLP_MIX_EXP_NOMFMA  // (experiments, wrong results: what is the launch's time made of?  profiles/r05_mix_anatomy.txt)
      hi[j] += a1[0] * r1[j][0], lo[j] += a2[1] * r2[j][1];
#else
      lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, r1[j], lo, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r1[j], hi, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r2[j], lo, 0, 0, 0);
#endif
    }
#ifndef XLP_MIX_EXP_NOSTAGE
    if (pass + 1u < p1) stage(buf ^ 1u);
    if (pass + 2u < p1) request(pass + 2u);
#endif
    // this lane's rows: registers g, g + 1 (g even) = (re, im) of the pass's segment xlm_result_row(g, h) / 2 = 2 h + (g >> 1 & 1) +
    // 4 (g >> 2): one 64-bit product per lane (its first segment), then wave-uniform steps; the bounds test is per lane only in the
    // call's last pass
    {
      const uint32_t s0 = pass * XLP_SEG;
      char *__restrict__ const base = reinterpret_cast<char *>(Yc + (size_t)(s0 + 2u * h) * ystride);
      const size_t sb = ystride * sizeof(v2f);
      const bool whole = s0 + XLP_SEG <= a.nseg;  // (wave-uniform)
#pragma unroll
      for (int g2 = 0; g2 < 16; g2 += 2) {
        const uint32_t cs = (uint32_t)(((g2 >> 1) & 1) + 4 * (g2 >> 2));  // (a constant after unrolling)
        v2f y = {(hi[g2] + lo[g2]) * cs_, (hi[g2 + 1] + lo[g2 + 1]) * cs_};
        if (SEG) {
          const float si = sinv[buf][2u * h + cs];
          y.x *= si, y.y *= si;
        }
        v2f *const dst = reinterpret_cast<v2f *>(base + cs * sb);
#ifdef XLP_MIX_EXP_NOSTORE
        if ((whole || s0 + 2u * h + cs < a.nseg) && y.x == 1.2345e-33f) __builtin_nontemporal_store(y, dst);
#elif defined(XLP_Y_TEMPORAL)  // (tools/mall_calibration.sh: the same stores with the default cache policy)
        if (whole || s0 + 2u * h + cs < a.nseg) *dst = y;
#else
        if (whole || s0 + 2u * h + cs < a.nseg) __builtin_nontemporal_store(y, dst);
#endif
      }
    }
    xlp_lds_barrier();  // the other buffer is staged; everybody is done with this one
  };
  // The first pass's products run as the operands arrive (the compiler's waits before product j leave the later operands in flight);
  // for the other passes the operands are waited for HERE, once -- left to itself the compiler puts those waits into the pass loop,
  // where they would also wait for the rows the previous pass has just requested.
  products(p0);
#pragma unroll
  for (int j = 0; j < NKB; ++j) asm volatile("" : "+v"(r1[j]), "+v"(r2[j]));
  for (uint32_t pass = p0 + 1u; pass < p1; ++pass) products(pass);
  xlp_trace_work(a, t_begin);
}

// ------------------------------------------------------------------------------------------- launcher
static bool xlp_valid_m(uint32_t M) { return M == 64u || M == 128u || M == 256u; }

template <int NKB>
static void xlp_launch_mix_mfma_n(const XlpArgs &a, const dim3 grid, hipStream_t s) {
  if (a.segmax != nullptr) hipLaunchKernelGGL((xlp_mix_mfma_kernel<NKB, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((xlp_mix_mfma_kernel<NKB, false>), grid, dim3(256), 0, s, a);
}

hipError_t xlp_launch_mix(const XlpArgs &a0, hipStream_t s) {
  if (!xlp_valid_m(a0.M) || a0.nco_blocks != 0u) return hipErrorInvalidValue;  // (no NCO role next to matrix instructions: see the kernels)
  XlpArgs a = a0;
  a.nco_skip = 0u;
  a.nco_skip_at = 0xFFFFFFFFu;
  a.mix_passes = (a0.nseg + XLP_SEG - 1) / XLP_SEG;
  if (a0.mix_kind == 3u) return xlp_launch_mix_f32(a, s);  // float32 operands (xl_mixf32.hip)
  // (a cf32 stream's spectra are unbounded: only with the per-segment scales)
  if (a0.mix_kind != 1u || a0.nkb == 0u || a0.nkb > XLP_NKB_MAX || a0.D > 8u * a0.nkb || a0.Rh == nullptr || a0.cscale == nullptr ||
      (a0.fmt == XLF_CF32 && a0.segmax == nullptr) || (a0.segmax != nullptr && a0.seg_cap < a.mix_passes * XLP_SEG))
    return hipErrorInvalidValue;
  // (all passes of an 8-block call in one workgroup: the operands are fetched once; A/B at 4096 clients, passes per
  // workgroup 4 / 8 / 16: 42.5 / 38.5 / 34.8 us per block, at 1024 clients 10.3 / 9.3 / 10.0)
  // (64-point classes: twice the segments for the same samples -- 32 passes, so that a call of the same length still is one run)
  if (a.mix_pp == 0u) a.mix_pp = a.M == 64u ? 32u : 16u;
  const uint32_t runs = (a.mix_passes + a.mix_pp - 1u) / a.mix_pp;
  const dim3 grid(a.M * a.ncg * runs);
  switch (a.nkb) {
    case 1: xlp_launch_mix_mfma_n<1>(a, grid, s); break;
    case 2: xlp_launch_mix_mfma_n<2>(a, grid, s); break;
    case 3: xlp_launch_mix_mfma_n<3>(a, grid, s); break;
    case 4: xlp_launch_mix_mfma_n<4>(a, grid, s); break;
    case 5: xlp_launch_mix_mfma_n<5>(a, grid, s); break;
    case 6: xlp_launch_mix_mfma_n<6>(a, grid, s); break;
    case 7: xlp_launch_mix_mfma_n<7>(a, grid, s); break;
    case 8: xlp_launch_mix_mfma_n<8>(a, grid, s); break;
    default: xlp_mix_wide_launch(a, s); break;  // 9 .. 14 k-blocks: xl_mixh2.hip
  }
  return hipGetLastError();
}

