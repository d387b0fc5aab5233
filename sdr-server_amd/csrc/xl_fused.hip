// xl_fused.hip -- the polyphase overlap-save path (xl_polyphase.h: same operator as /root/reference/src/xlating.c:52-72) with
// the mixed spectra ON CHIP: two launches per call and class instead of three.
//
//   xlp_forward_h_kernel   raw samples -> cf32 (xlating.c:357-378, exact) -> 128-point DFT of every polyphase branch per
//                          segment -> the shared spectra in the matrix instruction's A-operand form, scaled and split in two
//                          halves ("Xh", xl_fused_layout.h), stored as whole 128-byte lines.
//   xlp_fused_kernel       workgroup = (8 segments x 16 client columns x all 128 bins), 4 waves at 256 registers: TWO workgroups
//                          share a CU.  Per bin the sums over the branches
//                              Y[c][s][m] = sum_b X[s][b][m] R[c][b][m]
//                          are one 16 x 16 x 2D real matrix product on v_mfma_f32_16x16x32_f16 (rows = (segment, re / im):
//                          (X.re, X.im) / (X.im, -X.re); columns = clients: (R.re, -R.im)), every float32 operand carried as two
//                          halves (X R ~ X1 R1 + X1 R2 + X2 R1, FP32 accumulation, small terms first: see xlp_mix_mfma_kernel
//                          in xl_polyphase.hip for the arithmetic and its error model).  The results NEVER leave the chip: wave
//                          w keeps the bins m = 4 i + w of the whole tile in 128 accumulation registers per lane (the register
//                          file is the largest memory of a CU: 512 KB against 160 KB of LDS), runs the 32-point inverse
//                          transforms of its own bins in registers (xl_fft64.h), and the four waves combine their quarters
//                          through LDS (one write + one read of the tile instead of the four of a staged transform) ->
//                          scale -> NCO rotate (xlating.c:70) with the tabulated float32 phases -> out[k].
// What the three-launch path moved through the memory system and this one does not: the mixed spectra Y, written and read back
// once per call (8 M bytes per client and segment each way: 58 % of that path's traffic).  What it costs: the operands of a
// tile (3 KB of spectra + 6 KB of branch spectra per bin) are streamed from L2 by every tile -- no workgroup can hold the
// branch spectra of all bins of its columns -- so the launch is bound by L2 -> CU bandwidth, not by HBM; a tile's operand
// stream (memory-bound) and its epilogue (vector-ALU-bound) overlap with the other workgroup's on the same CU.
// (Round 4's first shape -- 16 x 16 tiles, one wave per SIMD at 512 registers, profiles/r04_fused_designA_timesplit.txt -- had
// two thirds of the operand traffic per client and nothing to overlap its epilogue with: 2.4 x slower than the three launches.)
#include "xl_poly_dev.h"

#include "xl_fused_layout.h"

#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------- forward transforms, operand form
// grid = nco_blocks + ceil(nseg / 8) * ceil(D / 4) transform workgroups + a.roll_blocks history-roll workgroups.
// A transform workgroup = (8 segments, 4 branches): 32 transforms of 128 points on 32 lanes each (1024 threads), the spectra
// staged once in LDS so that thread (bin m, segment) can gather its 4 branches = one 16-byte operand slot per term; the 8
// segments of a bin are 128 contiguous bytes of the image.
#define XLF_FWD_PITCH (XLP_ROW(XLF_M) + 4u)  // row pitch = 4 mod 32 elements: the gather (4 bins x 8 rows per 32 lanes) is conflict-free
__global__ __launch_bounds__(1024) void xlp_forward_h_kernel(const XlpArgs a) {
  constexpr uint32_t M = XLF_M, L = M / 4, NT = 1024;
  __shared__ v2f lds[32][XLF_FWD_PITCH];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t j = threadIdx.x;
  const uint32_t nbq = (a.D + 3u) >> 2, nwg = ((a.nseg + 7u) >> 3) * nbq;
  if (bid >= nwg) {
    // raw-history roll (as in xlp_forward_kernel): hist_out = the last hist_units 2-byte units of [in0 | in1]
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
  const uint32_t h = j / L, l = j % L;   // transform of this lane = (branch bl of the quad, segment sl of the eight)
  const uint32_t bl = h >> 3, sl = h & 7u;
  const uint32_t s8 = bid / nbq, bq = bid - s8 * nbq;
  const uint32_t s = s8 * 8u + sl, b = bq * 4u + bl;
  const XlpTw tw = xlp_twiddles<-1, (int)M>(reinterpret_cast<const v2f *>(a.W), l);
  const bool live = s < a.nseg && b < a.D;  // (dead transforms: zeros -- finite operands whose sums are never stored)
  // branch sample n of segment s = stream sample base + (s V + n) D + b   (base: first tap of shared point 0)
  const uint32_t first = a.base + s * a.V * a.D + b;
  const uint32_t end = a.n0 + a.n1;
  v2f u[1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t idx = first + (l + L * r) * a.D;
    const bool ok = live && idx >= a.zero_below && idx < end;  // late joiner: zeros below; past the call: zeros
    const bool lo = idx < a.n0;
    const void *src = (lo || !ok) ? a.in0 : a.in1;
    const v2f v = xl_sample(src, (int)a.fmt, ok ? (lo ? idx : idx - a.n0) : 0u);
    u[0][r] = ok ? v : (v2f){0.0f, 0.0f};
  }
  v2f *const bufs[1] = {lds[h]};
  const uint32_t rs0[1] = {0u};
  xlp_dft<-1, 1, (int)M>(u, bufs, tw, l, rs0);
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[h][l + L * r] = u[0][r];  // natural order (the row is the transform's own scratch)
  __syncthreads();
  // thread (bin m, segment so): the four branches of the quad -> one operand slot per term
  const uint32_t m = j >> 3, so = j & 7u;
  uint32_t t0[4], t1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v2f x = lds[(uint32_t)q * 8u + so][m];
    _Float16 r1, r2, i1, i2;
    xlp_split_h(x.x * XLP_H_XSCALE, r1, r2);
    xlp_split_h(x.y * XLP_H_XSCALE, i1, i2);
    t0[q] = xlp_pack_h(r1, i1);
    t1[q] = xlp_pack_h(r2, i2);
  }
  uint4 *__restrict__ Xh = reinterpret_cast<uint4 *>(a.X);
  const uint32_t nk = xlf_nk(a.D);
  Xh[xlf_xh_slot(s8, nk, bq >> 2, bq & 3u, m, 0u, so)] = make_uint4(t0[0], t0[1], t0[2], t0[3]);
  Xh[xlf_xh_slot(s8, nk, bq >> 2, bq & 3u, m, 1u, so)] = make_uint4(t1[0], t1[1], t1[2], t1[3]);
}

// ------------------------------------------------------------------------------------------- branch spectra, operand form
// As xlp_tables_h_kernel (xl_polyphase.hip) for the 16 x 16 x 32 instruction's B operand: (R.re, -R.im) * scale[j] as two
// halves; branch b of column col lands in dword b & 3 of lane (kg = (b >> 2) & 3, c = col & 15) of k-block b >> 4, once per
// term.  Grid: 16 nk branches (those >= D: zeros).
__global__ __launch_bounds__(XLP_COLS) void xlp_tables_h16_kernel(const float2 *__restrict__ rt, const uint32_t *__restrict__ delta,
                                                                  const uint32_t *__restrict__ colidx, const float *__restrict__ scale,
                                                                  uint32_t nlist, uint32_t T, uint32_t D, uint32_t A, uint32_t nk,
                                                                  uint32_t *__restrict__ Rh) {
  __shared__ double wc[256], ws[256];
  for (uint32_t n = threadIdx.x; n < XLF_M; n += blockDim.x) sincospi(2.0 * (double)n / (double)XLF_M, &ws[n], &wc[n]);
  __syncthreads();
  const uint32_t m = blockIdx.x % XLF_M;
  const uint32_t b = blockIdx.x / XLF_M;
  const uint32_t j = blockIdx.y * XLP_COLS + threadIdx.x;
  if (j >= nlist) return;
  const uint32_t col = colidx[j];
  double sr, si;
  xlp_branch_spectrum(rt, nlist, j, delta[j], T, D, A, XLF_M, m, b, wc, ws, sr, si);
  const float sc = scale[j];
  _Float16 r1, r2, i1, i2;
  xlp_split_h((float)sr * sc, r1, r2);
  xlp_split_h(-(float)si * sc, i1, i2);
  const uint32_t ln = xlf_lane(xlf_kgroup(b), col & 15u);
  Rh[xlf_rh_slot(col >> 4, nk, m, xlf_kblock(b), 0u, ln) * 4u + xlf_dword(b)] = xlp_pack_h(r1, i1);
  Rh[xlf_rh_slot(col >> 4, nk, m, xlf_kblock(b), 1u, ln) * 4u + xlf_dword(b)] = xlp_pack_h(r2, i2);
}

// ------------------------------------------------------------------------------------------- mix + inverse + epilogue, fused
// A row (s, im) from the slot of row (s, re), per dword (lo, hi) = (X.re, X.im) -> (X.im, -X.re): ONE packed half-precision
// multiply by (+1, -1) with the source halves crossed (op_sel); exact.
XL_DEV v8h xlf_imrow(const v8h v) {
  uint4 x = __builtin_bit_cast(uint4, v);
  const uint32_t pm = 0xBC003C00u;  // (lo, hi) = (+1.0h, -1.0h)
  asm("v_pk_mul_f16 %0, %0, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f16 %1, %1, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f16 %2, %2, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f16 %3, %3, %4 op_sel:[1,0] op_sel_hi:[0,1]"
      : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)
      : "s"(pm));
  return __builtin_bit_cast(v8h, x);
}

// One table entry's worth of NCO phases (xlating.c:70-73) into a staging row: row[i] = phase of output m0 + i, i < count <= 16,
// p = the tabulated phase of output m0 (a multiple of the table stride).  Straight line when no block of the call ends inside
// the run (the common case); else the checked walk of the other consumers (xl_phase_walk), bit-identical to the producer's chain.
XL_DEV void xlf_stage_phases(v2f p, const uint32_t m0, const uint32_t count, const v2f inc, const XlBnd bnd, v2f *__restrict__ row) {
  if (xl_bnd_next(bnd, m0) >= m0 + XL_PH_STRIDE && !(bnd.flags & XL_POS_FMA_STEP)) {
#pragma unroll
    for (uint32_t i = 0; i < XL_PH_STRIDE; ++i) {
      if (i < count) row[i] = p;
      if (i + 1u < XL_PH_STRIDE) p = xl_nco_next(p, inc);
    }
  } else {
    xl_phase_walk(p, m0, count, inc, bnd, [&](uint32_t i, v2f phs) { row[i] = phs; });
  }
}

// Tile order.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each with a private 4 MB L2: XCD x owns the
// 16-column groups cg16 = x mod 8 and walks them against the segment tiles four at a time, so that the tiles in flight on an
// XCD share a few column groups' branch spectra and a few segment groups' spectra -- which they stream bin by bin, roughly in
// step: the L2 needs to hold a window of bins, not the images.
// grid = 8 * (ncg16 / 8) * roundup(nst, 4) workgroups of 256 threads, nst = ceil(nseg / 8) segment tiles.
template <int NK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void xlp_fused_kernel(const XlpArgs a) {
  __shared__ v2f exch[4u * 32u * 32u];        // 32 KB: Z'_w[pair][k] of one sub-step (xlf_exch)
  __shared__ v2f phl[XLF_COLS][XLF_PH_ROW];   // 36 KB: the sub-step's NCO phases per client column
  __shared__ v2f twl[4][32];                  // e^{+2 pi j w k / 128}
  __shared__ uint4 cinfo[XLF_COLS];           // per client column: out row, grid shift, outputs owned, first staged output of the sub-step
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  // ---- which tile
  const uint32_t ncgl = a.ncg;  // 16-column groups per XCD = ncg * 8 / 8
  const uint32_t x = blockIdx.x & 7u, r = blockIdx.x >> 3;
  const uint32_t nst = (a.nseg + XLF_SEGS - 1u) / XLF_SEGS;
  const uint32_t cg16 = ((r >> 2) % ncgl) * 8u + x;
  const uint32_t sg = (r / (4u * ncgl)) * 4u + (r & 3u);
  if (sg >= nst) return;
  // ---- the client column whose phase chains this thread walks (tid & 15: the same in every round), and is anybody here?
  const uint32_t cc = tid & 15u;
  const XlpCol colc = a.cols[cg16 * XLF_COLS + cc];
  if (!__syncthreads_or(colc.out_off != 0xFFFFFFFFu)) return;  // (a group of empty columns)
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;  // a column with j0 < Nr owns Ka + 1 outputs, else Ka
  XlBnd bndc;
  bndc.j0 = xl_merge_j0(a.j0_ref, colc.delta, a.D), bndc.D = a.D, bndc.S = a.pos.S, bndc.G = a.pos.G, bndc.flags = a.pos.pad;
  bndc.K = colc.out_off != 0xFFFFFFFFu ? Ka + (bndc.j0 < Nr ? 1u : 0u) : 0u;
  const uint32_t shiftc = xl_merge_shift(a.j0_ref, colc.delta, a.D);
  if (tid < 128u) {
    const uint32_t e = ((tid >> 5) * (tid & 31u)) & 127u;
    twl[tid >> 5][tid & 31u] = (v2f){xl_w128_cos((int)e), xl_w128_sin((int)e)};
  }
  if (tid < XLF_COLS) cinfo[tid] = make_uint4(colc.out_off, shiftc, bndc.K, 0u);
  // ---- the sums over the branches: wave w, bins m = 4 i + w
  v4f d[32];  // per bin: (re, im) of Y of (client column lane & 15, segment lane >> 4), then of segment 4 + (lane >> 4)
  {
    const uint32_t i16 = lane & 15u, kg = lane >> 4;
    // (uniform base pointer + 32-bit lane offset: the loads address as scalar base + vector offset)
#ifdef XLF_EXP_SAME_OPERANDS  // (tools/experiments: every tile streams the SAME operands -- all L2 hits; WRONG results)
    const uint4 *__restrict__ Xb = reinterpret_cast<const uint4 *>(a.X);
    const uint4 *__restrict__ Rb = reinterpret_cast<const uint4 *>(a.Rh);
#else
    const uint4 *__restrict__ Xb = reinterpret_cast<const uint4 *>(a.X) + xlf_xh_slot(sg, NK, 0u, 0u, 0u, 0u, 0u);
    const uint4 *__restrict__ Rb = reinterpret_cast<const uint4 *>(a.Rh) + xlf_rh_slot(cg16, NK, 0u, 0u, 0u, 0u);
#endif
    const uint32_t xbyte = (uint32_t)xlf_xh_slot(0u, NK, 0u, kg, 0u, 0u, xlf_row_seg(i16)) * 16u, rbyte = lane * 16u;  // (< 2^20)
    const bool imrow = xlf_row_comp(i16) != 0u;
    v8h a1[2][NK], a2[2][NK], b1[2][NK], b2[2][NK];
    auto ld = [](const uint4 *__restrict__ base, const uint32_t byte) __attribute__((always_inline)) {
#ifdef XLF_EXP_NO_LOADS  // (tools/experiments: what the launch costs without its operand stream -- WRONG results)
      const uint32_t v = (uint32_t)(uintptr_t)base + byte;
      return __builtin_bit_cast(v8h, make_uint4(v & 0x03FF03FFu, (v >> 3) & 0x03FF03FFu, (v >> 5) & 0x03FF03FFu, (v >> 7) & 0x03FF03FFu));
#else
      return __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(base) + byte));
#endif
    };
    auto load = [&](const uint32_t m, const int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        // (slot strides: k-block 4 * 128 * 2 * 8, bin 2 * 8, term 8)
        const uint4 *__restrict__ xp = Xb + ((size_t)j * 4u * XLF_M + m) * 16u;
        const uint4 *__restrict__ rp = Rb + ((size_t)m * NK + j) * 128u;
        a1[buf][j] = ld(xp, xbyte);
        a2[buf][j] = ld(xp + 8, xbyte);
        b1[buf][j] = ld(rp, rbyte);
        b2[buf][j] = ld(rp + 64, rbyte);
      }
    };
    load(xlf_bin(w, 0u), 0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int cur = i & 1, nxt = cur ^ 1;
      if (i + 1 < 32) load(xlf_bin(w, (uint32_t)i + 1u), nxt);  // one bin ahead: requested before this bin's products
      __builtin_amdgcn_sched_barrier(0);
      if (imrow) {
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          a1[cur][j] = xlf_imrow(a1[cur][j]);
          a2[cur][j] = xlf_imrow(a2[cur][j]);
        }
      }
      v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
      // small terms first (X2 R1, X1 R2), then X1 R1 on top of them: one accumulator
#pragma unroll
      for (int j = 0; j < NK; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[cur][j], b1[cur][j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NK; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[cur][j], b2[cur][j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NK; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[cur][j], b1[cur][j], acc, 0, 0, 0);
      d[i] = acc;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#ifdef XLF_EXP_NO_EPILOGUE  // (tools/experiments: the operand stream + products alone -- WRONG results)
  {
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    if (sum == 12345.678f) reinterpret_cast<float *>(a.out)[0] = sum;
    return;
  }
#endif
  // ---- epilogue: half h = result registers (2 h, 2 h + 1) = segments 4 h .. 4 h + 3 of the tile; sub-step ab = the lanes
  // 32 ab .. 32 ab + 31 of every wave = segments 4 h + 2 ab, + 1
  const uint32_t c = lane & 15u;  // as a result lane: client column c, segment 4 h + (lane >> 4)
  // what undoes the operand scales (a power of two) and the transform's 1 / M, applied before the transform (linear, exact)
  const float scl = a.cscale[cg16 * XLF_COLS + c] * (1.0f / (float)XLF_M);
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  const v2f incc = {colc.incr.x, colc.incr.y};
  const uint32_t hp = lane >> 5, k = lane & 31u;  // as a consumer lane: point k of the client columns 8 (w & 1) + 2 pp + hp
  __syncthreads();  // (cinfo, twl)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (sg * XLF_SEGS + 4u * (uint32_t)h >= a.nseg) break;  // (workgroup-uniform)
    // -- the lane's sequence of this half -> 32-point inverse transform -> twiddle: z[k] in slot xl_fft32_slot(k)
    v2f u[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) u[i] = (v2f){d[i][2 * h], d[i][2 * h + 1]} * scl;
    XL_FFT_FENCE();
    xl_fft32_inverse<v2f, XlpFftOps>(u);
#pragma unroll
    for (int kk = 1; kk < 32; ++kk) {
      u[xl_fft32_slot(kk)] = xlp_cmul_v(u[xl_fft32_slot(kk)], twl[w][kk]);
      if (kk % 4 == 3) XL_FFT_FENCE();
    }
#pragma unroll 1
    for (uint32_t ab = 0; ab < 2u; ++ab) {
      const uint32_t S0 = sg * XLF_SEGS + 4u * (uint32_t)h + 2u * ab;  // first of the sub-step's two segments
      if (S0 >= a.nseg) break;
      // -- this thread's phase chains: the outputs of column cc inside the sub-step's (live) segments, table entry by entry
      const uint32_t Se = S0 + 2u < a.nseg ? S0 + 2u : a.nseg;
      const uint32_t qlo = S0 * a.V > shiftc ? S0 * a.V : shiftc;
      const uint32_t klo = qlo - shiftc;
      const uint32_t khi = Se * a.V - shiftc < bndc.K ? Se * a.V - shiftc : bndc.K;  // (Se V >= V >= 2 > shiftc)
      const uint32_t base = klo & ~(XL_PH_STRIDE - 1u);
      const uint32_t nent = khi > base ? (khi - base + XL_PH_STRIDE - 1u) >> XL_PH_SHIFT : 0u;
      if (tid < XLF_COLS) cinfo[tid].w = base;
      v2f pe[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {  // (requested before the exchange writes)
        const uint32_t te = (tid >> 4) + 16u * (uint32_t)rr;
        pe[rr] = ph[te < nent ? (colc.out_off >> XL_PH_SHIFT) + (base >> XL_PH_SHIFT) + te : 0u];
      }
      // -- producers: the lanes of this sub-step hand their sequences to the exchange buffer
      if (hp == ab) {
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) exch[xlf_exch(w, k, (uint32_t)kk)] = u[xl_fft32_slot(kk)];
      }
      // -- the phases
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const uint32_t te = (tid >> 4) + 16u * (uint32_t)rr;
        if (te < nent) {
          const uint32_t m0 = base + te * XL_PH_STRIDE;
          const uint32_t left = khi - m0;
          xlf_stage_phases(pe[rr], m0, left < XL_PH_STRIDE ? left : XL_PH_STRIDE, incc, bndc, &phl[cc][te * XL_PH_STRIDE]);
        }
      }
      __syncthreads();
      // -- consumers: waves 0, 1 = segment S0, waves 2, 3 = S0 + 1; wave parity = which 8 of the 16 columns; lane = point k
      // of two columns at a time
      const uint32_t seg = S0 + (w >> 1);
      if (seg < a.nseg) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const uint32_t cl = 8u * (w & 1u) + 2u * (uint32_t)pp + hp, p = 16u * (w >> 1) + cl;
          const v2f z0 = exch[xlf_exch(0u, p, k)], z1 = exch[xlf_exch(1u, p, k)];
          const v2f z2 = exch[xlf_exch(2u, p, k)], z3 = exch[xlf_exch(3u, p, k)];
          const uint4 ci = cinfo[cl];
          // y[k + 32 q] = sum_w j^{w q} z_w
          const v2f t0 = z0 + z2, t1 = z0 - z2, t2 = z1 + z3, t3 = z1 - z3;
          v2f y[4];
          y[0] = t0 + t2;
          y[1] = XlpFftOps::add_j(t1, t3);
          y[2] = t0 - t2;
          y[3] = XlpFftOps::sub_j(t1, t3);
          // shared point of y[q]: seg V + k + 32 q; the column's output index is that - shift, staged phase index that - base.
          // (32-bit index arithmetic on purpose: where a shared point lies below the column's first output the differences wrap,
          // and wrap back for the points that are outputs -- as offsets of a 64-bit pointer they would not)
          const uint32_t q0 = seg * a.V + k;
          const bool inner = q0 >= ci.y + k && q0 - k + a.V <= ci.y + ci.z;  // every point of the segment is an output of the column
          const uint32_t o0 = ci.x + (q0 - ci.y), p0 = cl * XLF_PH_ROW + (q0 - ci.y - ci.w);
          const v2f *__restrict__ phf = &phl[0][0];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t n = k + 32u * (uint32_t)q, qs = q0 + 32u * (uint32_t)q;
            if (n < a.V && (inner || (qs >= ci.y && qs - ci.y < ci.z))) out[o0 + 32u * (uint32_t)q] = xl_rotate<1>(y[q], phf[p0 + 32u * (uint32_t)q]);
          }
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------- launchers
hipError_t xlp_launch_tables_h16(const float2 *rt, const uint32_t *delta, const uint32_t *colidx, const float *scale,
                                 uint32_t nlist, uint32_t T, uint32_t D, uint32_t A, uint32_t nk, void *Rh, hipStream_t s) {
  if (nlist == 0u || nk == 0u || nk > XLF_NK_MAX || D > 16u * nk) return hipErrorInvalidValue;
  hipLaunchKernelGGL(xlp_tables_h16_kernel, dim3(XLF_M * 16u * nk, (nlist + XLP_COLS - 1u) / XLP_COLS), dim3(XLP_COLS), 0, s, rt,
                     delta, colidx, scale, nlist, T, D, A, nk, reinterpret_cast<uint32_t *>(Rh));
  return hipGetLastError();
}

hipError_t xlp_launch_forward_h(const XlpArgs &a, hipStream_t s) {
  if (a.M != XLF_M || a.D > 16u * XLF_NK_MAX) return hipErrorInvalidValue;
  const dim3 grid(a.nco_blocks + ((a.nseg + 7u) >> 3) * ((a.D + 3u) >> 2) + a.roll_blocks);
  hipLaunchKernelGGL(xlp_forward_h_kernel, grid, dim3(1024), 0, s, a);
  return hipGetLastError();
}

// `done` (optional): recorded with the launch's own completion signal
hipError_t xlp_launch_fused(const XlpArgs &a, hipStream_t s, hipEvent_t done) {
  const uint32_t nk = xlf_nk(a.D);
  if (a.M != XLF_M || nk == 0u || nk > XLF_NK_MAX || a.Rh == nullptr || a.cscale == nullptr || a.fmt == XLF_CF32 ||
      a.A < 2u || a.V + a.A != XLF_M + 1u || a.ncg == 0u || a.nco_blocks != 0u)
    return hipErrorInvalidValue;
  const uint32_t nst = (a.nseg + XLF_SEGS - 1u) / XLF_SEGS;
  const dim3 grid(8u * a.ncg * ((nst + 3u) & ~3u));
  void (*kern)(const XlpArgs) = nk == 1u ? xlp_fused_kernel<1> : nk == 2u ? xlp_fused_kernel<2> : nk == 3u ? xlp_fused_kernel<3> : xlp_fused_kernel<4>;
  if (done) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}
