// xl_mixh2.hip -- the two-half matrix-core mix (see xlp_mix_mfma_kernel in xl_polyphase.hip for the algebra, the operand layout and the
// scales) for WIDE classes: 9 .. XLP_NKB_MAX = 14 k-blocks of 8 branches (D = 65 .. 112; BASELINE config 5: D = 100).  Same sums
//
//   Y[c][s][m] = sum_b X[s][b][m] R[c][b][m]        (src/xlating.c:66-71, evaluated per spectrum bin of the D polyphase branches)
//
// same operand images, same results as the narrow kernel -- another schedule, because another resource binds.  A wave's B operands (its
// 32 columns' branch spectra of one bin, two half terms: 8 NKB registers) stay in registers for all passes of the call; at 13 k-blocks
// that is 104 registers, so the kernel lives on a TWO-waves-per-SIMD budget, and with two waves nothing hides a wave's own latencies:
// measured on the first form of this kernel (xlp_mix_mfma_kernel<13>: profiles/r06_mix_halves_cf32.txt) the launch's time was the SUM of
// its phases -- operand arrival 16 us, matrix instructions 16, staging 14, the rest 21 per call.  Hence the points below.  What they do
// NOT change is what bounds the launch (profiles/r06_mix_wide_timeline.txt): every workgroup of a round pulls its 104 KB of operands at
// once -- 53 MB per round at ~11 bytes per cycle and CU, i.e. the 5.3 TB/s the memory system gives --, with idle matrix cores (10 of a workgroup's 25 us), then runs its passes with an idle
// memory system.  A second form that streams the operands under the products of all passes (k-blocks outermost) was built and was no
// faster at 1024 clients (the intake per workgroup is the same; 8 % faster at 2048+): tools/experiments/mix_wide_kmajor/.  Its wrong
// sums in a few workgroups per launch led to the rule this file is compiled under: NO PACKED FP32 beside matrix instructions
// (-fno-slp-vectorize: xl_mixh.hip's header; profiles/r06_mix_wide_kmajor_wrong_sums.txt (11)).
//   * the two LDS buffers of staged A operands are two distinct arrays and the pass loop is unrolled by two, so that the compiler knows
//     that the reads of one pass and the staging writes for the next never alias, and may interleave them;
//   * a pass is ONE basic block -- no per-lane or per-round branch: rows beyond the class's branches are loaded from a clamped address and
//     zeroed with a select, k-blocks beyond NKB are staged into a dump slot, the last pass stages (harmlessly) once more -- whose
//     instruction order is pinned with sched_group_barrier: A-operand LDS reads four k-blocks ahead of their matrix instructions, the
//     next pass's staging (conversions and LDS writes) spread between the matrix instructions;
//   * the first pass's products start as the operands arrive (requested behind the first pass's rows); LDS-only barriers.
#include "xl_polyphase.h"

#include "xl_poly_dev.h"
#include "xl_mix_layout.h"

#ifndef XLMH_PF
#define XLMH_PF 4  // k-blocks whose A operands are read ahead of their matrix instructions
#endif

template <int NKB>
struct XlmhBuf {
  uint4 x[2][NKB + 1][64];  // [term][k-block, + 1 dump slot][lane slot]: the staged A operands of one pass
  float unscale[4][XLP_SEG];  // SEG: what undoes the pass's segment scales (one copy per wave)
};

template <int NKB, bool SEG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void xlp_mix_mfma_wide_kernel(const XlpArgs a) {
  static_assert(NKB > (int)XLP_NKB_4W && NKB <= (int)XLP_NKB_MAX, "the wide classes");
  __shared__ XlmhBuf<NKB> buf0, buf1;
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;
  const uint32_t bid = blockIdx.x;
  const uint32_t tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
  const uint32_t M = a.M;
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
  uint32_t smx[2] = {0u, 0u};
  const uint32_t *__restrict__ segmax = SEG ? a.segmax + ((size_t)a.seg_par * a.seg_cap + 2u * sp) * XLP_SEGMAX_STRIDE : nullptr;
  // rows of pass `pass` (clamped to the run's last pass: the request behind the last pass reads that pass's rows again, for nobody)
  auto request = [&](const uint32_t pass) __attribute__((always_inline)) {
    const uint32_t pc = pass < p1 ? pass : p1 - 1u;
    if (SEG) smx[0] = segmax[(size_t)pc * XLP_SEG * XLP_SEGMAX_STRIDE], smx[1] = segmax[((size_t)pc * XLP_SEG + 1u) * XLP_SEGMAX_STRIDE];
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t b = 8u * xlm_stage_kblock(w, (uint32_t)q) + bb;
      const uint32_t bc = b < a.Dpad ? b : a.Dpad - 1u;
      const v4f v = Xm[((size_t)pc * a.Dpad + bc) * xrow];
      const bool ok = b < a.D;  // (rows D .. Dpad - 1 of the image are zeros; a select, not a branch)
      g[q] = (v4f){ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f};
    }
  };
  auto stage = [&](XlmhBuf<NKB> &dst) __attribute__((always_inline)) {
    const float sx0 = SEG ? xlp_seg_scale(smx[0]) : XLP_H_XSCALE, sx1 = SEG ? xlp_seg_scale(smx[1]) : XLP_H_XSCALE;
    // (the eight lanes of a segment pair write the same two values to the same place)
    if (SEG) *reinterpret_cast<v2f *>(&dst.unscale[w][2u * sp]) = (v2f){xlp_seg_unscale(smx[0]), xlp_seg_unscale(smx[1])};
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t jq = xlm_stage_kblock(w, (uint32_t)q);
      const uint32_t j = jq < (uint32_t)NKB ? jq : (uint32_t)NKB;  // (wave-uniform select: k-blocks beyond the class go to the dump slot)
      _Float16 f1[4], f2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) xlp_split_h(g[q][e] * (e < 2 ? sx0 : sx1), f1[e], f2[e]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // segment 2 sp + u: (re, im) = f[2 u], f[2 u + 1]
        const uint32_t sre = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(2u * sp + (uint32_t)u, 0u)));
        const uint32_t sim = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(2u * sp + (uint32_t)u, 1u)));
        reinterpret_cast<uint32_t *>(&dst.x[0][j][sre])[xlm_dword(bb)] = xlp_pack_h(f1[2 * u], f1[2 * u + 1]);
        reinterpret_cast<uint32_t *>(&dst.x[0][j][sim])[xlm_dword(bb)] = xlp_pack_h(f1[2 * u + 1], -f1[2 * u]);
        reinterpret_cast<uint32_t *>(&dst.x[1][j][sre])[xlm_dword(bb)] = xlp_pack_h(f2[2 * u], f2[2 * u + 1]);
        reinterpret_cast<uint32_t *>(&dst.x[1][j][sim])[xlm_dword(bb)] = xlp_pack_h(f2[2 * u + 1], -f2[2 * u]);
      }
    }
  };
  // ---- Y image [cg][segment][sub][bin][CW columns] (the inverse workgroups' tiles): this lane's column of segment s
  const uint32_t CW = xlp_tile_columns(M), NSUB = XLP_COLS / CW;
  const uint32_t col = w * 32u + c;
  v2f *__restrict__ Yc = reinterpret_cast<v2f *>(a.Y) + ((((size_t)cg * a.nseg_cap) * NSUB + col / CW) * M + m) * CW + col % CW;
  const size_t ystride = (size_t)NSUB * M * CW;  // v2f per segment
  request(p0);
  // ---- B operands of this wave: 2 NKB runs of 1 KB, requested behind the first pass's rows
  const uint4 *__restrict__ Rp = reinterpret_cast<const uint4 *>(a.Rh);
  v8h r1[NKB], r2[NKB];
#pragma unroll
  for (int j = 0; j < NKB; ++j) {
    r1[j] = __builtin_bit_cast(v8h, Rp[xlm_rh_slot(cg, M, m, w, 0u, NKB, (uint32_t)j, lane)]);
    r2[j] = __builtin_bit_cast(v8h, Rp[xlm_rh_slot(cg, M, m, w, 1u, NKB, (uint32_t)j, lane)]);
  }
  stage(buf0);
  request(p0 + 1u);
  xlp_lds_barrier();
  // One pass: products from `cur`, the next pass's operands into `nxt`, the rows of the pass after that requested, the sums stored.
  auto pass_body = [&](const XlmhBuf<NKB> &cur, XlmhBuf<NKB> &nxt, const uint32_t pass) __attribute__((always_inline)) {
    v16f32 hi, lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) hi[i] = lo[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const v8h a1 = __builtin_bit_cast(v8h, cur.x[0][j][xlm_lds_slot(lane)]);
      const v8h a2 = __builtin_bit_cast(v8h, cur.x[1][j][xlm_lds_slot(lane)]);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, r1[j], lo, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r1[j], hi, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r2[j], lo, 0, 0, 0);
    }
    stage(nxt);
    request(pass + 2u);
#ifndef XLMH_EXP_NOSCHED
    // instruction order of the block above: A operands XLMH_PF k-blocks ahead; between the matrix instructions of a k-block and the
    // next one's, a share of the staging (masks: 0x008 matrix, 0x100 LDS read, 0x200 LDS write, 0x002 vector ALU, 0x020 memory read)
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * XLMH_PF, 0);
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      if (j + XLMH_PF < NKB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#ifndef XLMH_EXP_NOINTERLEAVE
      __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
#endif
    }
#endif
    __builtin_amdgcn_sched_barrier(0);  // (the epilogue's LDS reads and stores are not part of the pattern above)
    // this lane's rows: registers g, g + 1 (g even) = (re, im) of the pass's segment 2 h + (g >> 1 & 1) + 4 (g >> 2)
    {
      const uint32_t s0 = pass * XLP_SEG;
      char *__restrict__ const base = reinterpret_cast<char *>(Yc + (size_t)(s0 + 2u * h) * ystride);
      const size_t sb = ystride * sizeof(v2f);
      v2f y[8];
#pragma unroll
      for (int g2 = 0; g2 < 16; g2 += 2) {
        const uint32_t cs = (uint32_t)(((g2 >> 1) & 1) + 4 * (g2 >> 2));  // (a constant after unrolling)
        y[g2 >> 1] = (v2f){(hi[g2] + lo[g2]) * cs_, (hi[g2 + 1] + lo[g2 + 1]) * cs_};
        if (SEG) {
          const float si = cur.unscale[w][2u * h + cs];
          y[g2 >> 1].x *= si, y[g2 >> 1].y *= si;
        }
      }
      if (s0 + XLP_SEG <= a.nseg) {  // (wave-uniform: every pass but the call's last)
#pragma unroll
        for (int g2 = 0; g2 < 16; g2 += 2) {
          const uint32_t cs = (uint32_t)(((g2 >> 1) & 1) + 4 * (g2 >> 2));
          __builtin_nontemporal_store(y[g2 >> 1], reinterpret_cast<v2f *>(base + cs * sb));
        }
      } else {
#pragma unroll
        for (int g2 = 0; g2 < 16; g2 += 2) {
          const uint32_t cs = (uint32_t)(((g2 >> 1) & 1) + 4 * (g2 >> 2));
          if (s0 + 2u * h + cs < a.nseg) __builtin_nontemporal_store(y[g2 >> 1], reinterpret_cast<v2f *>(base + cs * sb));
        }
      }
    }
    xlp_lds_barrier();  // `nxt` is staged; everybody is done with `cur`
  };
  // The first pass's products run as the operands arrive; for the other passes the operands are waited for HERE, once (left to itself
  // the compiler puts those waits into the pass loop, where they would also wait for the rows the previous pass has just requested).
  pass_body(buf0, buf1, p0);
#pragma unroll
  for (int j = 0; j < NKB; ++j) asm volatile("" : "+v"(r1[j]), "+v"(r2[j]));
#ifdef XL_TUNING  // (timeline of the launch: when this wave's first pass -- the one that waits for the operands -- was over)
  if (a.trace && lane == 0u && bid * 4u + w < 6000u) a.trace[4096 + 4 * (size_t)(bid * 4u + w) + 3] = wall_clock64();
#endif
  for (uint32_t pass = p0 + 1u; pass < p1; pass += 2u) {
    pass_body(buf1, buf0, pass);
    if (pass + 1u < p1) pass_body(buf0, buf1, pass + 1u);
  }
  xlp_trace_work(a, t_begin);
}

template <int NKB>
static void xlmh_launch_n(const XlpArgs &a, const dim3 grid, hipStream_t s) {
  if (a.segmax != nullptr) hipLaunchKernelGGL((xlp_mix_mfma_wide_kernel<NKB, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((xlp_mix_mfma_wide_kernel<NKB, false>), grid, dim3(256), 0, s, a);
}

// called by xlp_launch_mix (xl_polyphase.hip) with the checked arguments (mix_pp, mix_passes set)
void xlp_mix_wide_launch(const XlpArgs &a, hipStream_t s) {
  const uint32_t runs = (a.mix_passes + a.mix_pp - 1u) / a.mix_pp;
  const dim3 grid(a.M * a.ncg * runs);
  switch (a.nkb) {
    case 9: xlmh_launch_n<9>(a, grid, s); break;
    case 10: xlmh_launch_n<10>(a, grid, s); break;
    case 11: xlmh_launch_n<11>(a, grid, s); break;
    case 12: xlmh_launch_n<12>(a, grid, s); break;
    case 13: xlmh_launch_n<13>(a, grid, s); break;
    default: xlmh_launch_n<14>(a, grid, s); break;
  }
}
