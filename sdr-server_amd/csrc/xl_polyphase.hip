// xl_polyphase.hip -- polyphase overlap-save evaluation of the frequency-xlating FIR (see xl_polyphase.h for the
// algebra and why it is the same operator as /root/reference/src/xlating.c:52-72).  Hand-written for gfx950.
//
// Three launches per call and class (stream order is the only synchronisation):
//   xlp_forward_kernel  one workgroup per (pass of 16 segments, branch): raw samples -> cf32 (xlating.c:357-378, exact)
//                       -> M-point DFT of the branch per segment -> shared spectra X[pass][b][m][s], stored as whole rows.
//   xlp_mix_mfma_kernel Y[c][s][m] = sum_b X[s][b][m] * R[c][b][m] on the matrix cores: one real matrix product per bin m (rows =
//                       (segment, re / im), columns = clients, k = (branch, re / im)) with every float32 operand carried as two
//                       halves, v_mfma_f32_32x32x16_f16, FP32 accumulation.  Integer input formats, D <= 64 (the default there).
//   xlp_mix_f32_kernel  (xl_mixf32.hip) the same sums with float32 operands on v_mfma_f32_32x32x2_f32 -- the float32 FMA chain itself:
//                       cf32 input, D > 64, and every class on request (option "mix_kernel" = 3).
//   xlp_inverse8_kernel (xl_inv8.hip) / xlp_inverse32_kernel (xl_inv32.hip) (128-point classes: small / big launches) /
//   xlp_inverse_kernel  (256-point classes; 128-point ones on request): per (segment, 32 or 16 columns): Y tile -> M-point inverse DFT
//                       per column -> scale, NCO rotate (xlating.c:70) with the tabulated float32 phase -> out[k], k < K.
// M = 256 or 128 per class (xl_polyphase.h).  When the NCO phases of the next call are not tabulated by the side-stream
// chain kernel (xl_kernels.hip), the forward and the inverse launch each carry a slice of that recurrence ("NCO role").
// Rounds 1-4 also shipped a packed-FMA mix kernel, a fused mix + inverse launch, 48-bit mixed spectra and three more inverse
// kernels: measured, documented (profiles/r03_*, r04_*), and retired in round 5 (tools/experiments/retired/).
#include "xl_polyphase.h"

#include "xl_poly_dev.h"
#include "xl_mix_layout.h"

#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------- forward transforms
// grid = nco_blocks + passes * D transform workgroups + a.roll_blocks history-roll workgroups.  A transform workgroup =
// (pass, branch b): the XLP_SEG = 16 segments of the pass, one transform each on M / 4 lanes (16 * M / 4 threads: 512 or
// 1024).  The spectra go through LDS once more so that the image rows X[pass][b][m][0..15] -- what the mix kernel fetches
// as one 128-byte scalar row -- leave as whole lines: 8 lanes x 16 bytes per row, the workgroup's 16 KB (M = 128) back to
// back.  (One wave per transform storing its 8-byte values 128 bytes apart wrote 40 MB for an 11 MB image and took 25 us
// per call of 8 blocks at 1024 clients.)
template <int M>
__global__ __launch_bounds__(XLP_SEG * M / 4) void xlp_forward_kernel(const XlpArgs a) {
  constexpr uint32_t L = M / 4, NT = XLP_SEG * L;
  __shared__ v2f lds[XLP_SEG][XLP_ROW(M)];
  __shared__ uint32_t tmax[XLP_SEG];  // (cf32 streams on the two-half mix: float bits of the largest |component| of each transform)
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t j = threadIdx.x;
  const uint32_t passes = (a.nseg + XLP_SEG - 1u) / XLP_SEG;
  const uint32_t nwg = passes * a.D;
  if (bid >= nwg) {
    // raw-history roll (as in xl_fir_kernel): hist_out = the last hist_units 2-byte units of [in0 | in1]; nothing in
    // this block's launches reads hist_out
    const uint32_t rb = bid - nwg;
    const uint16_t *__restrict__ h0 = reinterpret_cast<const uint16_t *>(a.in0);
    const uint16_t *__restrict__ h1 = reinterpret_cast<const uint16_t *>(a.in1);
    uint16_t *__restrict__ ho = reinterpret_cast<uint16_t *>(a.hist_out);
    for (uint32_t i = rb * NT + j; i < a.hist_units; i += a.roll_blocks * NT) {
      const uint32_t sidx = a.block_units + i;
      ho[i] = (sidx < a.hist_units) ? h0[sidx] : h1[sidx - a.hist_units];
    }
    return;
  }
  const uint32_t h = j / L, l = j % L;  // transform (= segment of the pass) of this lane, lane within it
  const XlpTw tw = xlp_twiddles<-1, M>(reinterpret_cast<const v2f *>(a.W), l);
  // Which (pass, branch): branch-major over the XCDs.  Workgroup bid runs on XCD bid % 8; the transform workgroups of XCD x are given a
  // CONTIGUOUS range of the branch-major list (branch, pass), so an XCD works on ~D/8 neighbouring branches -- with wide samples (cf32:
  // 16 per 128-byte line, D = 100 branches per period) it then pulls a quarter of the block's lines through its L2 instead of all of
  // them (with (pass, branch) = (bid / D, bid % D) every XCD fetched every line: 67 MB by the counters for an 8.4 MB super-block,
  // profiles/r06_bench_full.json).  Narrow samples (cu8: 64 per line >= D) are unaffected either way.
  uint32_t pass, b;
  {
#ifdef XLP_EXP_FWD_PASS_MAJOR
    pass = bid / a.D, b = bid - pass * a.D;
#else
    const uint32_t x = bid & 7u, kx = bid >> 3;
    uint32_t start = 0u;  // workgroups of the XCDs below x: XCD y holds the bids y, y + 8, .. < nwg
    for (uint32_t y = 0u; y < x; ++y) start += (nwg - y + 7u) >> 3;
    const uint32_t jj = start + kx;
    b = jj / passes, pass = jj - b * passes;
#endif
  }
  const uint32_t s = pass * XLP_SEG + h;
  const bool live = s < a.nseg;  // (the last pass may hold fewer segments: zeros, never read by the mix kernel)
  // branch sample n of segment s = stream sample base + (s V + n) D + b   (base: first tap of shared point 0)
  const uint32_t first = a.base + s * a.V * a.D + b;
  const uint32_t end = a.n0 + a.n1;
  // (two-half mix of a cf32 stream: what the maximum of segment pass * 16 + j stands at, read by thread j < 16 ahead of the transforms
  // -- stale is fine: one entry per 128-byte line (XLP_SEGMAX_STRIDE), and a global atomic only where this workgroup raises what it saw:
  // a few of the D workgroups of a segment, not all -- D x nseg atomics on six cache lines cost the launch 1.6 us per block)
  uint32_t *const smax = (a.segmax != nullptr && j < XLP_SEG && pass * XLP_SEG + j < a.nseg)
                             ? a.segmax + ((size_t)a.seg_par * a.seg_cap + pass * XLP_SEG + j) * XLP_SEGMAX_STRIDE : nullptr;
  const uint32_t seen = smax != nullptr ? __hip_atomic_load(smax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (l == 0u) tmax[h] = 0u;
  v2f u[1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t idx = first + (l + L * r) * a.D;
    const bool ok = live && idx >= a.zero_below && idx < end;  // late joiner: zeros below; past the block: zeros
                                                               // (those outputs lie beyond K and are never stored)
    const bool lo = idx < a.n0;
    const void *src = (lo || !ok) ? a.in0 : a.in1;
    const v2f v = xl_sample(src, (int)a.fmt, ok ? (lo ? idx : idx - a.n0) : 0u);
    u[0][r] = ok ? v : (v2f){0.0f, 0.0f};
  }
  v2f *const bufs[1] = {lds[h]};
  const uint32_t rs0[1] = {0u};
  xlp_dft<-1, 1, M>(u, bufs, tw, l, rs0);
  if (a.segmax != nullptr) {
    // cf32 stream on the two-half mix: the segment's largest spectrum component, over all branches -- this transform's share of it
    // (NaNs drop out of fmaxf: a stream that carries them has no parity to keep), gathered with one LDS atomic per lane (the lanes of a
    // transform sit in one wave, whose LDS operations execute in order: the clear above needs no barrier)
    float mx = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fmaxf(fabsf(u[0][r].x), fabsf(u[0][r].y)));
#ifdef XLP_EXP_FWD_SHUFFLE  // (experiment: the first form -- shuffles, one global atomic per transform)
#pragma unroll
    for (uint32_t o = L / 2u; o > 0u; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, (int)o));
    if (l == 0u && live) atomicMax(a.segmax + ((size_t)a.seg_par * a.seg_cap + s) * XLP_SEGMAX_STRIDE, __float_as_uint(mx));
#else
    atomicMax(&tmax[h], __float_as_uint(mx));
#endif
  }
  // the transform's row, natural order (its own scratch: the LDS operations of a wave execute in order)
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[h][l + L * r] = u[0][r];
  __syncthreads();
#ifndef XLP_EXP_FWD_SHUFFLE
  if (smax != nullptr && tmax[j] > seen) atomicMax(smax, tmax[j]);
#endif
  if (a.segmax != nullptr && bid == 0u)  // the next call's buffer (last read by the previous call's mix launch)
    for (uint32_t i = j; i < a.seg_cap; i += NT) a.segmax[((size_t)(a.seg_par ^ 1u) * a.seg_cap + i) * XLP_SEGMAX_STRIDE] = 0u;
  static_assert(XLP_XS == 16u && XLP_SEG <= XLP_XS, "image rows of 16 complex = 8 x 16 bytes");
  v4f *__restrict__ X = reinterpret_cast<v4f *>(a.X) + ((size_t)pass * a.Dpad + b) * M * (XLP_XS / 2u);
  for (uint32_t i = j; i < (uint32_t)M * (XLP_XS / 2u); i += NT) {
    const uint32_t m = i >> 3, part = i & 7u;  // row m, segments 2 part and 2 part + 1
    const v2f x0 = 2u * part < XLP_SEG ? lds[2u * part][m] : (v2f){0.0f, 0.0f};
    const v2f x1 = 2u * part + 1u < XLP_SEG ? lds[2u * part + 1u][m] : (v2f){0.0f, 0.0f};
    X[i] = (v4f){x0.x, x0.y, x1.x, x1.y};
  }
}

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
        for (int e = 0; e < 4; ++e) xlp_split_h(xl_mul_s(g[q][e], e < 2 ? sx0 : sx1), f1[e], f2[e]);  // (no packed FP32 here: xl_mul_s)
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
  const uint32_t CW = M == 256u ? 16u : 32u, NSUB = XLP_COLS / CW;
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
        // (no packed FP32 beside matrix instructions: xl_poly_dev.h, xl_mul_s)
        v2f y = {xl_mul_s(xl_add_s(hi[g2], lo[g2]), cs_), xl_mul_s(xl_add_s(hi[g2 + 1], lo[g2 + 1]), cs_)};
        if (SEG) {
          const float si = sinv[buf][2u * h + cs];
          y.x = xl_mul_s(y.x, si), y.y = xl_mul_s(y.y, si);
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

// ------------------------------------------------------------------------------------------- inverse + epilogue
// grid = nco_blocks + nseg * ncg * (128 / CW) workgroups of 256 threads; workgroup = (segment, CW columns), CW = 16
// (M = 256: a wave runs its four columns' transforms interleaved) or 32 (M = 128: each half-wave runs four).  The tile
// rows double as the transforms' scratch.
template <int M, class P>
XL_DEV void xlp_inverse_body(const XlpArgs &a) {
  constexpr bool SWZ = !__is_same(P, XlpPosPad);
  static_assert(!SWZ || M == 128, "the swizzled layout is written for rows of 128 elements");
  constexpr uint32_t L = M / 4;            // lanes per transform
  constexpr uint32_t CW = 16u * (256 / M);  // columns per workgroup
  constexpr uint32_t WPC = CW / 4;          // columns per wave
  constexpr uint32_t NSUB = XLP_COLS / CW;  // workgroups per column group
  // [column][padded bin position].  Row length XLP_POS(M - 1) + 1 (319 / 159): 2 banks short of a multiple of 32, so
  // the lanes that fill different rows of one bin hit distinct bank pairs; and 16 x 319 x 8 B = 40832 B lets a CU hold
  // four workgroups (at 41.2 KB it held three: 768 slots for the 832 workgroups of a 1024-client block -> a second round)
  __shared__ v2f tile[CW][SWZ ? M : XLP_ROW(M) - 1];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;  // (as in xlp_mix_kernel; 4-wave workgroups: per CU)
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t sub = bid % NSUB;
  const uint32_t q = bid / NSUB;
  const uint32_t cg = q % a.ncg, s = q / a.ncg;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const uint32_t h = j / L, l = j % L;  // (h = 0 for M = 256)
  const XlpTw tw = xlp_twiddles<+1, M>(reinterpret_cast<const v2f *>(a.W), l);
  {
    // Y tile: M bins x CW columns, one contiguous 32 KB run (the mix kernel lays it out so): thread (mrow, part) takes
    // 16 bytes = 2 columns of bin mrow + MR i -- every load instruction of the workgroup covers 4 KB back to back.
    constexpr uint32_t PARTS = CW / 2, MR = 256 / PARTS;
    const uint32_t part = threadIdx.x % PARTS, mrow = threadIdx.x / PARTS;
    v4f v[8];
    {
      const v4f *__restrict__ src = reinterpret_cast<const v4f *>(
          a.Y + ((((size_t)cg * a.nseg_cap + s) * NSUB + sub) * M) * CW) + part;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(mrow + MR * i) * PARTS];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t m = mrow + MR * i;
      tile[2 * part][P::pos(m, XLP_SWZ_ROW(2u * part))] = (v2f){v[i].x, v[i].y};
      tile[2 * part + 1][P::pos(m, XLP_SWZ_ROW(2u * part + 1u))] = (v2f){v[i].z, v[i].w};
    }
  }
  // the epilogue's operands.  A column's client lies on the class's shared grid with its own offset (xl_grid.h):
  // its output k is the shared point q = k + shift, shift in {0, 1}, and it owns K_c outputs in this call.
  // NCO phases: the table holds every XL_PH_STRIDE-th phase; after the transforms lane (en, gq) = (j / GQ, j % GQ),
  // GQ = M / XL_PH_STRIDE, expands the phases of the shared points gq*XL_PH_STRIDE .. of the segment for the wave's
  // column en into that column's tile row (free by then), and every lane picks the phases of its own points
  // l + L r from there.  The one table entry a lane needs is requested here, before the transforms.
  constexpr uint32_t GQ = M / XL_PH_STRIDE;
  static_assert(WPC * GQ == 64u, "one expansion duty per lane");
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;  // a column with j0 < Nr owns Ka + 1 outputs, else Ka
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  const uint32_t colbase = cg * XLP_COLS + sub * CW + WPC * w;
  uint32_t off[4], ksh[4], kc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const XlpCol c = a.cols[colbase + 4u * h + n];
    const uint32_t j0c = xl_merge_j0(a.j0_ref, c.delta, a.D);
    off[n] = c.out_off;
    ksh[n] = xl_merge_shift(a.j0_ref, c.delta, a.D);
    kc[n] = Ka + (j0c < Nr ? 1u : 0u);
  }
  const uint32_t en = j / GQ, gq = j % GQ;  // expansion duty: column en of the wave, shared points s V + gq*XL_PH_STRIDE ..
  const XlpCol ce = a.cols[colbase + en];
  XlBnd ebnd;
  ebnd.j0 = xl_merge_j0(a.j0_ref, ce.delta, a.D), ebnd.D = a.D, ebnd.S = a.pos.S, ebnd.G = a.pos.G, ebnd.flags = a.pos.pad;
  ebnd.K = Ka + (ebnd.j0 < Nr ? 1u : 0u);
  const uint32_t esh = xl_merge_shift(a.j0_ref, ce.delta, a.D);
  const uint32_t q0 = s * a.V + gq * XL_PH_STRIDE;
  const uint32_t ibeg = q0 < esh ? 1u : 0u;     // (shared point 0 of a column with shift 1 is nobody's output)
  const uint32_t m0 = q0 + ibeg - esh;          // the column's output index of the first phase to expand
  const bool eok = ce.out_off != 0xFFFFFFFFu && gq * XL_PH_STRIDE < a.V && m0 < ebnd.K;
  const v2f pe = ph[eok ? (ce.out_off >> XL_PH_SHIFT) + (m0 >> XL_PH_SHIFT) : 0u];
  __syncthreads();
  const unsigned long long t_loaded = a.trace ? wall_clock64() : 0ull;
  v2f u[4][4];
  v2f *const rows[4] = {tile[WPC * w + 4u * h], tile[WPC * w + 4u * h + 1], tile[WPC * w + 4u * h + 2],
                        tile[WPC * w + 4u * h + 3]};
  const uint32_t rbase = WPC * w + 4u * h;  // (the rows' swizzle constants)
  const uint32_t rs[4] = {XLP_SWZ_ROW(rbase), XLP_SWZ_ROW(rbase + 1u), XLP_SWZ_ROW(rbase + 2u), XLP_SWZ_ROW(rbase + 3u)};
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) u[n][r] = rows[n][P::pos(l + L * r, rs[n])];
  __builtin_amdgcn_wave_barrier();
  xlp_dft<+1, 4, M, P>(u, rows, tw, l, rs);
  __builtin_amdgcn_wave_barrier();
  if (eok) {
    v2f *__restrict__ row = tile[WPC * w + en];
    const uint32_t ers = XLP_SWZ_ROW(WPC * w + en);
    const uint32_t left = ebnd.K - m0, span = XL_PH_STRIDE - ibeg;
    const uint32_t p0 = gq * XL_PH_STRIDE + ibeg;
    xl_phase_walk(pe, m0, left < span ? left : span, (v2f){ce.incr.x, ce.incr.y}, ebnd,
                  [&](uint32_t i, v2f phs) { row[P::pos(p0 + i, ers)] = phs; });
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned long long t_xf = a.trace ? wall_clock64() : 0ull;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t qo = l + L * r, qs = s * a.V + qo;  // shared point of this value
      if (off[n] != 0xFFFFFFFFu && qo < a.V && qs >= ksh[n] && qs - ksh[n] < kc[n]) {
        const v2f y = u[n][r] * (1.0f / (float)M);  // exact scaling by 2^-8 / 2^-7
        out[off[n] + (qs - ksh[n])] = xl_rotate<1>(y, rows[n][P::pos(qo, rs[n])]);
      }
    }
  }
  if (a.trace && threadIdx.x == 0 && bid < 6000u) {  // tuning: start, tile loaded, transforms done, end
    unsigned long long *t = a.trace + 4096 + 4 * (size_t)bid;
    t[0] = t_begin;
    t[1] = wall_clock64();
    t[2] = t_loaded;
    t[3] = t_xf;
  }
}

template <int M, class P = XlpPosPad>
__global__ __launch_bounds__(256) void xlp_inverse_kernel(const XlpArgs a) {
  xlp_inverse_body<M, P>(a);
}
// ------------------------------------------------------------------------------------------- branch spectra
// R[cg][m][b][col] = sum_{a<A} r'_col[D a + b] e^{+2 pi j a m / M}, r' = the column's taps delayed by its grid offset
// (xl_grid.h), in double (xlp_branch_spectrum), for a LIST of columns (all of them when a class is built, the newcomers' when a
// client joins): thread j of block (m, b) handles list entry j -- column colidx[j], taps rt[.][j], delay delta[j].
// In the two-half mix's operand form (xlp_mix_mfma_kernel): (R.re, -R.im) rounded once to float32, times scale[j] -- a power of
// two: another exponent, the same digits -- as two halves each; branch b of column cl = 32 w + c lands in
// dword b & 3 of lane slot (h = (b >> 2) & 1, c) of k-block b >> 3, once per term.  Grid: 8 nkb branches (those >= D: zeros).
__global__ __launch_bounds__(XLP_COLS) void xlp_tables_h_kernel(const float2 *__restrict__ rt,
                                                                const uint32_t *__restrict__ delta,
                                                                const uint32_t *__restrict__ colidx,
                                                                const float *__restrict__ scale, uint32_t nlist, uint32_t T,
                                                                uint32_t D, uint32_t A, uint32_t M, uint32_t nkb,
                                                                uint32_t *__restrict__ Rh) {
  __shared__ double wc[256], ws[256];
  for (uint32_t n = threadIdx.x; n < M; n += blockDim.x) sincospi(2.0 * (double)n / (double)M, &ws[n], &wc[n]);
  __syncthreads();
  const uint32_t m = blockIdx.x % M;
  const uint32_t b = blockIdx.x / M;
  const uint32_t j = blockIdx.y * XLP_COLS + threadIdx.x;
  if (j >= nlist) return;
  const uint32_t col = colidx[j];
  double sr, si;
  xlp_branch_spectrum(rt, nlist, j, delta[j], T, D, A, M, m, b, wc, ws, sr, si);
  const float sc = scale[j];
  _Float16 r1, r2, i1, i2;
  xlp_split_h((float)sr * sc, r1, r2);
  xlp_split_h(-(float)si * sc, i1, i2);
  const uint32_t cg = col / XLP_COLS, cl = col % XLP_COLS;
  const uint32_t w = cl >> 5, ln = xlm_lane(xlm_half(b), cl & 31u);
  Rh[xlm_rh_slot(cg, M, m, w, 0u, nkb, xlm_kblock(b), ln) * 4u + xlm_dword(b)] = xlp_pack_h(r1, i1);
  Rh[xlm_rh_slot(cg, M, m, w, 1u, nkb, xlm_kblock(b), ln) * 4u + xlm_dword(b)] = xlp_pack_h(r2, i2);
}

// ------------------------------------------------------------------------------------------- launchers
static bool xlp_valid_m(uint32_t M) { return M == 128u || M == 256u; }

hipError_t xlp_launch_tables_h(const float2 *rt, const uint32_t *delta, const uint32_t *colidx, const float *scale,
                               uint32_t nlist, uint32_t T, uint32_t D, uint32_t A, uint32_t M, uint32_t nkb, void *Rh,
                               hipStream_t s) {
  if (!xlp_valid_m(M) || nlist == 0u || nkb == 0u || nkb > XLP_NKB_MAX || D > 8u * nkb) return hipErrorInvalidValue;
  hipLaunchKernelGGL(xlp_tables_h_kernel, dim3(M * 8u * nkb, (nlist + XLP_COLS - 1u) / XLP_COLS), dim3(XLP_COLS), 0, s, rt, delta,
                     colidx, scale, nlist, T, D, A, M, nkb, reinterpret_cast<uint32_t *>(Rh));
  return hipGetLastError();
}

hipError_t xlp_launch_forward(const XlpArgs &a, hipStream_t s) {
  if (!xlp_valid_m(a.M)) return hipErrorInvalidValue;
  const uint32_t passes = (a.nseg + XLP_SEG - 1u) / XLP_SEG;
  const dim3 grid(a.nco_blocks + passes * a.D + a.roll_blocks);
  if (a.M == 256u) hipLaunchKernelGGL(xlp_forward_kernel<256>, grid, dim3(XLP_SEG * 64u), 0, s, a);
  else hipLaunchKernelGGL(xlp_forward_kernel<128>, grid, dim3(XLP_SEG * 32u), 0, s, a);
  return hipGetLastError();
}

// The skipped positions must exist and lie behind the NCO-role workgroups, else the launch carries no skip.
static XlpArgs xlp_checked_skip(const XlpArgs &a, uint32_t work_blocks) {
  XlpArgs b = a;
  if (b.nco_skip == 0u || b.nco_skip_at < b.nco_blocks || b.nco_skip_at + b.nco_skip > b.nco_blocks + work_blocks) {
    b.nco_skip = 0u;
    b.nco_skip_at = 0xFFFFFFFFu;
  }
  return b;
}

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
  if (a.mix_pp == 0u) a.mix_pp = 16u;
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

// `done` (optional): recorded with the launch's own completion signal -- one queue packet instead of launch + event record
hipError_t xlp_launch_inverse(const XlpArgs &a0, hipStream_t s, hipEvent_t done) {
  if (!xlp_valid_m(a0.M)) return hipErrorInvalidValue;
  // 128-point classes: eight lanes per column, transforms of 16 and 8 points in registers (xl_inv8.hip), the 32 x 4 cut (xl_inv32.hip),
  // or staged in LDS on dense XOR-swizzled rows -- xlp_inverse_pick() says which; 256-point classes: staged in LDS on padded rows.
  // Workgroup = one tile of 32 (16) columns.
  const uint32_t tiles = a0.nseg * a0.ncg * (a0.M == 256u ? 8u : 4u);
  const uint32_t kind = xlp_inverse_pick(a0.M, a0.inv_reg, tiles);
  const uint32_t work = kind == 6u ? xlp_inverse32_work(tiles) : tiles;
  const XlpArgs a = xlp_checked_skip(a0, work);
  const dim3 grid(a.nco_blocks + a.nco_skip + work);
  if (kind == 6u) {
    xlp_inverse32_launch(a, grid, s, done);
    return hipGetLastError();
  }
  if (kind == 5u) {
    xlp_inverse8_launch(a, grid, s, done);
    return hipGetLastError();
  }
  void (*kern)(const XlpArgs) = a.M == 256u ? xlp_inverse_kernel<256, XlpPosPad> : xlp_inverse_kernel<128, XlpPosSwz>;
  if (done) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}
