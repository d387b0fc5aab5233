// xl_mixf32.hip -- the mix launch of the polyphase path on the matrix cores with FLOAT32 operands (PolyClass::mix_kind 3).
//
// Same sums as xlp_mix_mfma_kernel (xl_polyphase.hip): Y[c][s][m] = sum_b X[s][b][m] R[c][b][m], the per-client multiply-accumulate of
// xlating.c:66-71 evaluated per spectrum bin of the D polyphase branches (xl_polyphase.h) -- here as one real matrix product per bin on
// v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate, bit for bit a chain of fmaf in branch order (the chain rounds 1-4 issued as
// packed FMAs on the vector ALUs: xlp_mix_kernel, retired in round 5).  Nothing is split or scaled, so cf32 streams -- whose spectra
// have no bound -- and any branch count take it, and option "mix_kernel" = 3 gives every class the all-float32 arithmetic.
// rows = (segment of the pass, re / im): the 16 segments of a pass fill the 32 rows; columns = 32 clients; one instruction = one branch
// (k = 0: the "re" factor, k = 1: the "im" factor): xl_mixf_layout.h, checked on the host by tests/c/test_mixf_layout.cpp.
//
// What to expect of it: the float32 matrix instruction has the SAME peak as the packed FP32 FMAs (64 flop per cycle and SIMD).  It
// takes one instruction per 4096 flop where the vector pipe needed 16, so everything else a wave does stops competing with the
// arithmetic -- measured (profiles/r05_mix_f32.txt, 4096 clients, us per block): 57.4 with 16-segment passes against 56.9 for the
// packed-FMA kernel; ~80 % of the matrix pipe's rate on the CUs the launch gets, at a chain-checked clock of 2.3 GHz, and no faster
// with its global loads compiled out: it is bound by the FP32 rate of the chip like its predecessor.  Gauss's three-product form of
// the complex product (3 D / 2 instructions instead of 2 D) was built to parity and measured too: -5 % at D = 42, +20 % at D = 100 --
// its three accumulators cost the occupancy it needs (profiles/r05_mix_f32_gauss_form.txt; tools/experiments/gauss_mix/).
//
// Workgroup = 4 waves = (bin m, column group of 128 clients, a run of passes); wave w = the group's columns 32 w .. 32 w + 31.
//   B operands: the wave's branch spectra of bin m in operand form (Rf: xlmf_rf_slot), ONE float per lane and branch, read once per
//               workgroup as 1 KB runs and kept in registers for all its passes (D <= 112; xlp_mix_f32_stream_kernel beyond).
//   A operands: the forward launch's image row X[pass][b][m][0..15] (16 segments x (re, im) = 128 bytes) IS the k = 0 operand in
//               row order, and the k = 1 operand is the same line with the floats of each pair swapped and the second negated: lane
//               (h, r) takes float r ^ h of the row and flips its sign when h & r & 1.  So the rows are staged into LDS AS THEY ARE
//               (16 bytes per thread, a wave instruction = 8 whole rows, one pass ahead: the software pipeline of
//               xlp_mix_mfma_kernel) and a matrix instruction's operand is one conflict-free ds_read_b32 (64 lanes, 32 distinct
//               floats of one row) -- no conversion, no transposition.  (Round 5's first build loaded the operand straight from
//               the image, one 4-byte load per lane and branch and no LDS: parity-equal, but 42 single-line requests per wave and
//               pass ran 20 % slower -- profiles/r05_mix_f32.txt.)
// This launch never hosts the NCO role (no launch that issues matrix instructions does: DESIGN 3.6).
#include "xl_poly_dev.h"

#include "xl_mixf_layout.h"

#include <hip/hip_ext.h>

typedef float v4f32 __attribute__((ext_vector_type(4)));

// where workgroup `bid` works: bin, column group, passes [p0, p1) (xlp_mix_place: XCD-aware, as in xlp_mix_mfma_kernel)
struct XlmfJob {
  uint32_t m, cg, p0, p1;
};
XL_DEV XlmfJob xlmf_job(const XlpArgs &a, const uint32_t bid) {
  const uint32_t pp = a.mix_pp, runs = (a.mix_passes + pp - 1u) / pp;
  uint32_t run;
  XlmfJob j;
  xlp_mix_place(bid, a.M, runs, j.m, j.cg, run);
  j.p0 = run * pp;
  j.p1 = j.p0 + pp < a.mix_passes ? j.p0 + pp : a.mix_passes;
  return j;
}

// the wave's results of one pass -> Y image [cg][segment][sub][bin][CW columns] (the inverse workgroups' tiles): registers g, g + 1
// (g even) = (re, im) of the pass's segment xlmf_result_row(g, h) / 2 = 2 h + (g >> 1 & 1) + 4 (g >> 2), column c.  One 64-bit
// product per lane (its first segment), then wave-uniform steps; the bounds test is per lane only in the call's last pass.
XL_DEV void xlmf_store_pass(const XlpArgs &a, const v16f32 &acc, v2f *__restrict__ Yc, const size_t ystride, const uint32_t pass,
                            const uint32_t h) {
  static_assert(XLP_SEG == 16u, "the 16 segments of a pass fill the 32 rows of the instruction");
  const uint32_t s0 = pass * XLP_SEG;
  char *__restrict__ const base = reinterpret_cast<char *>(Yc + (size_t)(s0 + 2u * h) * ystride);
  const size_t sb = ystride * sizeof(v2f);  // bytes from a segment's tile to the next one's (wave-uniform)
  const bool whole = s0 + XLP_SEG <= a.nseg;  // (wave-uniform)
#pragma unroll
  for (int g2 = 0; g2 < 16; g2 += 2) {
    const uint32_t cs = (uint32_t)(((g2 >> 1) & 1) + 4 * (g2 >> 2));  // (a constant after unrolling; = xlmf_result_row(g2, 0) / 2)
    // (written once, read once by the next launch: streamed past the L2 lines that hold the operands)
    if (whole || s0 + 2u * h + cs < a.nseg) __builtin_nontemporal_store((v2f){acc[g2], acc[g2 + 1]}, reinterpret_cast<v2f *>(base + cs * sb));
  }
}

// Workgroup barrier for LDS hand-offs only: this wave's LDS writes are done (lgkmcnt), everybody arrives.  __syncthreads() also fences
// global memory, which on this target is `s_waitcnt vmcnt(0)`: a wait for every store of the pass and for the rows requested for the
// pass after next, once per pass.
XL_DEV void xlmf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
XL_DEV float xlmf_flip(const float v, const uint32_t sgn) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) ^ sgn); }

template <int NB8>
__global__ __launch_bounds__(256) void xlp_mix_f32_kernel(const XlpArgs a) {
  constexpr int NJ = 8 * NB8;
  constexpr int ROUNDS = (NJ + 31) / 32;  // staging: 256 threads x 16 bytes = 32 image rows per round
  // the bin's image rows of two passes, as they lie in the image: [branch][16 segments x (re, im)]
  __shared__ __attribute__((aligned(16))) float xs[2][NJ][32];
  const XlmfJob job = xlmf_job(a, blockIdx.x);
  if (job.p0 >= job.p1) return;
  const uint32_t tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
  const uint32_t M = a.M, m = job.m, cg = job.cg;
  float bq[NJ];
  // ---- staging role of this thread: 16 bytes (part) of the image rows jr, jr + 32, .. of the bin: row (pass, branch j, bin m) = 128
  // bytes at X + ((pass Dpad + j) M + m) 128 -- a wave instruction covers 8 whole rows
  const uint32_t D = a.D;
  const uint32_t jr = tid >> 3, part = tid & 7u;
  const char *__restrict__ xb = reinterpret_cast<const char *>(a.X) + (size_t)m * (XLP_XS * sizeof(float2)) + part * 16u;
  const size_t xrow = (size_t)M * (XLP_XS * sizeof(float2));  // bytes from one branch's row of a bin to the next one's
  v4f32 g[ROUNDS];
  auto request = [&](const uint32_t pass) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      // (rows D .. Dpad - 1 of the image are zeros; beyond Dpad there is nothing to read: those lanes re-read the last row and stage()
      // drops it.  Unconditional on purpose: a load under `j < D` comes with a branch and a register copy behind it that waits for
      // EVERYTHING in flight -- the operands included)
      const uint32_t j = jr + 32u * (uint32_t)q, jc = j < a.Dpad ? j : a.Dpad - 1u;
#ifdef XLMF_EXP_NOLOAD
      g[q] = (v4f32){0.25f, -0.5f, 0.125f, 1.0f};
#else
      g[q] = *reinterpret_cast<const v4f32 *>(xb + ((size_t)pass * a.Dpad + jc) * xrow);
#endif
    }
  };
  auto stage = [&](const uint32_t buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t j = jr + 32u * (uint32_t)q;
      if (j < (uint32_t)NJ) *reinterpret_cast<v4f32 *>(&xs[buf][j][4u * part]) = g[q];
    }
  };
  // ---- A operand of lane (h, r): float r ^ h of a row, negated when h & r & 1 (xl_mixf_layout.h)
  const uint32_t h = lane >> 5, c = lane & 31u;
  const uint32_t sgn = xlmf_a_negate(lane) << 31;
  const uint32_t af = xlmf_a_float(lane);
  // ---- Y: this lane's column of segment s
  const uint32_t CW = xlp_tile_columns(M), NSUB = XLP_COLS / CW;
  const uint32_t col = w * 32u + c;
  v2f *__restrict__ Yc = reinterpret_cast<v2f *>(a.Y) + ((((size_t)cg * a.nseg_cap) * NSUB + col / CW) * M + m) * CW + col % CW;
  const size_t ystride = (size_t)NSUB * M * CW;  // v2f per segment
  // Software pipeline (as xlp_mix_mfma_kernel): the rows of pass p + 1 go into the other buffer AFTER pass p's products and BEFORE
  // its stores, so that the wait for them finds no store younger than pass p - 1's.
  request(job.p0);
  // ---- B operands of this wave: 2 NB8 runs of 1 KB -- requested BEHIND the first pass's rows, so that staging those rows is not a wait
  // for the operands (loads return in order)
  {
    const v4f32 *__restrict__ Rp = reinterpret_cast<const v4f32 *>(a.Rh);
#pragma unroll
    for (int jb = 0; jb < NB8; ++jb)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const v4f32 v = Rp[xlmf_rf_slot(cg, M, m, w, (uint32_t)NB8, (uint32_t)jb, (uint32_t)q, lane)];
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[8 * jb + 4 * q + e] = v[e];
      }
  }
  stage(0u);
  if (job.p0 + 1u < job.p1) request(job.p0 + 1u);
  xlmf_lds_barrier();
  auto products = [&](const uint32_t pass, const uint32_t buf) __attribute__((always_inline)) {
    v16f32 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < NJ - 8 || (uint32_t)j < D)  // (wave-uniform; only the last k-block may be short)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xlmf_flip(xs[buf][j][af], sgn), bq[j], acc, 0, 0, 0);
    }
    if (pass + 1u < job.p1) stage(buf ^ 1u);
    if (pass + 2u < job.p1) request(pass + 2u);
    xlmf_store_pass(a, acc, Yc, ystride, pass, h);
    xlmf_lds_barrier();  // the other buffer is staged; everybody is done with this one
  };
#ifndef XLMF_EXP_NO_PEEL
  // The FIRST pass runs its products as the B operands arrive (they were requested first and return in order: the compiler's waits
  // before product j let the later operands -- and the second pass's rows behind them -- stay in flight): a workgroup of a short run
  // (2-3 passes for small classes) no longer sits out the operands' whole latency before its first matrix instruction.
  products(job.p0, 0u);
  if (job.p0 + 1u >= job.p1) return;
  // (for the other passes the operands are waited for HERE, once: left to itself the compiler puts those waits into the pass loop,
  // `vmcnt(5)` ahead of every pass's first products, which also waits for the rows the previous pass has just requested)
#pragma unroll
  for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(bq[j]));
  for (uint32_t pass = job.p0 + 1u; pass < job.p1; ++pass) products(pass, (pass - job.p0) & 1u);
#else
#pragma unroll
  for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(bq[j]));
  for (uint32_t pass = job.p0; pass < job.p1; ++pass) products(pass, (pass - job.p0) & 1u);
#endif
}

// Any branch count (D > 112: huge decimations, few segments per call): the B operands do not fit a wave's registers for all its
// passes, so every pass streams them again, one k-block of 8 branches ahead of the products (they come from L2 after the first
// pass of a workgroup).
__global__ __launch_bounds__(256) void xlp_mix_f32_stream_kernel(const XlpArgs a) {
  const XlmfJob job = xlmf_job(a, blockIdx.x);
  if (job.p0 >= job.p1) return;
  const uint32_t tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
  const uint32_t M = a.M, m = job.m, cg = job.cg, nb8 = a.nkb, D = a.D;
  const v4f32 *__restrict__ Rp = reinterpret_cast<const v4f32 *>(a.Rh) + xlmf_rf_slot(cg, M, m, w, nb8, 0u, 0u, lane);
  const uint32_t h = lane >> 5, c = lane & 31u;
  const uint32_t sgn = xlmf_a_negate(lane) << 31;
  const uint32_t loff = xlmf_a_float(lane) * 4u;
  const char *__restrict__ xb = reinterpret_cast<const char *>(a.X) + (size_t)m * (XLP_XS * sizeof(float2));
  const size_t xrow = (size_t)M * (XLP_XS * sizeof(float2));
  const uint32_t CW = xlp_tile_columns(M), NSUB = XLP_COLS / CW;
  const uint32_t col = w * 32u + c;
  v2f *__restrict__ Yc = reinterpret_cast<v2f *>(a.Y) + ((((size_t)cg * a.nseg_cap) * NSUB + col / CW) * M + m) * CW + col % CW;
  const size_t ystride = (size_t)NSUB * M * CW;
  struct Block {
    v4f32 b0, b1;
    float x[8];
  };
  auto fetch = [&](const uint32_t pass, const uint32_t jb) __attribute__((always_inline)) {
    Block k;
    k.b0 = Rp[(size_t)jb * 128u];  // (xlmf_rf_slot: a k-block = two slots of 64 lanes)
    k.b1 = Rp[(size_t)jb * 128u + 64u];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t j = 8u * jb + (uint32_t)e;
      // (rows D .. Dpad - 1 of the image are zeros, beyond Dpad there is nothing to read)
      k.x[e] = j < D ? *reinterpret_cast<const float *>(xb + ((size_t)pass * a.Dpad + j) * xrow + loff) : 0.0f;
    }
    return k;
  };
  for (uint32_t pass = job.p0; pass < job.p1; ++pass) {
    v16f32 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    Block cur = fetch(pass, 0u);
    for (uint32_t jb = 0; jb < nb8; ++jb) {
      const Block nxt = fetch(pass, jb + 1u < nb8 ? jb + 1u : jb);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float bv = e < 4 ? cur.b0[e & 3] : cur.b1[e & 3];
        if (8u * jb + (uint32_t)e < D) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xlmf_flip(cur.x[e], sgn), bv, acc, 0, 0, 0);
      }
      cur = nxt;
    }
    xlmf_store_pass(a, acc, Yc, ystride, pass, h);
  }
}

// ------------------------------------------------------------------------------------------- branch spectra, operand form
// The values xlp_tables_kernel computes (double arithmetic, rounded once), laid out as this kernel's B operands: (R.re, -R.im) of
// branch b of column cl = 32 w + c -> element b & 3 of the slots of lanes (0, c) and (1, c) of k-block b >> 3, half (b >> 2) & 1.
// Grid: 8 nb8 branches (those >= D: zeros), thread j of block (m, b) handles list entry j.
__global__ __launch_bounds__(XLP_COLS) void xlp_tables_f_kernel(const float2 *__restrict__ rt, const uint32_t *__restrict__ delta,
                                                                const uint32_t *__restrict__ colidx, uint32_t nlist, uint32_t T,
                                                                uint32_t D, uint32_t A, uint32_t M, uint32_t nb8,
                                                                float *__restrict__ Rf) {
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
  const uint32_t cg = col / XLP_COLS, cl = col % XLP_COLS;
  const uint32_t w = cl >> 5, c = cl & 31u;
  Rf[xlmf_rf_slot(cg, M, m, w, nb8, xlmf_b_block(b), xlmf_b_half(b), c) * 4u + xlmf_b_elem(b)] = (float)sr;
  Rf[xlmf_rf_slot(cg, M, m, w, nb8, xlmf_b_block(b), xlmf_b_half(b), 32u + c) * 4u + xlmf_b_elem(b)] = -(float)si;
}

hipError_t xlp_launch_tables_f(const float2 *rt, const uint32_t *delta, const uint32_t *colidx, uint32_t nlist, uint32_t T, uint32_t D,
                               uint32_t A, uint32_t M, uint32_t nb8, void *Rf, hipStream_t s) {
  if ((M != 64u && M != 128u && M != 256u) || nlist == 0u || nb8 == 0u || D > 8u * nb8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(xlp_tables_f_kernel, dim3(M * 8u * nb8, (nlist + XLP_COLS - 1u) / XLP_COLS), dim3(XLP_COLS), 0, s, rt, delta, colidx,
                     nlist, T, D, A, M, nb8, reinterpret_cast<float *>(Rf));
  return hipGetLastError();
}

// Passes per workgroup when the caller names none: the launch is bound by the matrix pipe, so what matters is that every SIMD gets
// the same number of (wave, pass) jobs -- runs of 16 passes (operands fetched once) while that still makes >= 4096 workgroups
// (four rounds of the chip), shorter runs (the operands then come from L2 again) for smaller classes.
// The runs are then made EQUAL: 6 passes cut 4 + 2 leave half the workgroups with twice the work of the others (BASELINE config 5 at
// 2048 clients: 30.8 us per block against 26.0 cut 3 + 3; profiles/r05_mix_f32.txt "balanced runs").
static uint32_t xlmf_default_pp(const XlpArgs &a, uint32_t passes) {
  uint32_t pp = 16u;
  while (pp > 2u && (size_t)a.M * a.ncg * ((passes + pp - 1u) / pp) < 4096u) pp >>= 1;
  const uint32_t runs = (passes + pp - 1u) / pp;
  return runs ? (passes + runs - 1u) / runs : pp;
}

template <int NB8>
static void xlmf_launch_n(const XlpArgs &a, const dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(xlp_mix_f32_kernel<NB8>, grid, dim3(256), 0, s, a);
}

// (called by xlp_launch_mix for mix_kind 3, with a.mix_passes set)
hipError_t xlp_launch_mix_f32(const XlpArgs &a0, hipStream_t s) {
  if (a0.nkb == 0u || a0.D > 8u * a0.nkb || a0.Rh == nullptr || a0.nco_blocks != 0u)  // (no NCO role next to matrix instructions)
    return hipErrorInvalidValue;
  XlpArgs a = a0;
  a.nco_skip = 0u;
  a.nco_skip_at = 0xFFFFFFFFu;
  if (a.mix_pp == 0u) a.mix_pp = xlmf_default_pp(a, a.mix_passes);
  const uint32_t runs = (a.mix_passes + a.mix_pp - 1u) / a.mix_pp;
  const dim3 grid(a.M * a.ncg * runs);
  switch (a.nkb) {
    case 1: xlmf_launch_n<1>(a, grid, s); break;
    case 2: xlmf_launch_n<2>(a, grid, s); break;
    case 3: xlmf_launch_n<3>(a, grid, s); break;
    case 4: xlmf_launch_n<4>(a, grid, s); break;
    case 5: xlmf_launch_n<5>(a, grid, s); break;
    case 6: xlmf_launch_n<6>(a, grid, s); break;
    case 7: xlmf_launch_n<7>(a, grid, s); break;
    case 8: xlmf_launch_n<8>(a, grid, s); break;
    case 9: xlmf_launch_n<9>(a, grid, s); break;
    case 10: xlmf_launch_n<10>(a, grid, s); break;
    case 11: xlmf_launch_n<11>(a, grid, s); break;
    case 12: xlmf_launch_n<12>(a, grid, s); break;
    case 13: xlmf_launch_n<13>(a, grid, s); break;
    case 14: xlmf_launch_n<14>(a, grid, s); break;
    default: hipLaunchKernelGGL(xlp_mix_f32_stream_kernel, grid, dim3(256), 0, s, a); break;
  }
  return hipGetLastError();
}
