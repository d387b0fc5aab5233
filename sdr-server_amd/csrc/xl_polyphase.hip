// xl_polyphase.hip -- polyphase overlap-save evaluation of the frequency-xlating FIR (see xl_polyphase.h for the
// algebra and why it is the same operator as /root/reference/src/xlating.c:52-72).  Hand-written for gfx950.
//
// Three launches per call and class (stream order is the only synchronisation):
//   xlp_forward_kernel  one workgroup per (pass of 14 segments, branch): raw samples -> cf32 (xlating.c:357-378, exact)
//                       -> M-point DFT of the branch per segment -> shared spectra X[pass][b][m][s], stored as whole rows.
//   xlp_mix_mfma_kernel Y[c][s][m] = sum_b X[s][b][m] * R[c][b][m] on the matrix cores: one real matrix product per bin m (rows =
//                       (segment, re / im), columns = clients, k = (branch, re / im)) with every float32 operand carried as two
//                       halves, v_mfma_f32_32x32x16_f16, FP32 accumulation.  Integer input formats, D <= 64 (the default there).
//   xlp_mix_kernel      the same sums as packed FP32 FMAs (cf32 input, D > 64, option): lane = two client columns, the bin m is
//                       workgroup-uniform: its rows of X arrive through the scalar cache as SGPR operands of the FMAs,
//                       R is streamed coalesced (16 bytes per lane and branch): 8 * D * M bytes per client and call.
//   xlp_inverse_kernel  per (segment, 16 or 32 columns): Y tile -> LDS (transposed) -> M-point inverse DFT per column
//                       -> scale, NCO rotate (xlating.c:70) with the tabulated float32 phase -> out[k], k < K.
// M = 256 or 128 per class (xl_polyphase.h).  When the NCO phases of the next call are not tabulated by the side-stream
// chain kernel (xl_kernels.hip), each launch also carries a slice of that recurrence ("NCO role").
#include "xl_polyphase.h"

#include "xl_poly_dev.h"
#include "xl_mix_layout.h"
#include "xl_y6.h"

#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------- forward transforms
// grid = nco_blocks + passes * D transform workgroups + a.roll_blocks history-roll workgroups.  A transform workgroup =
// (pass, branch b): the XLP_SEG = 14 segments of the pass, one transform each on M / 4 lanes (14 * M / 4 threads: 448 or
// 896).  The spectra go through LDS once more so that the image rows X[pass][b][m][0..15] -- what the mix kernel fetches
// as one 128-byte scalar row -- leave as whole lines: 8 lanes x 16 bytes per row, the workgroup's 16 KB (M = 128) back to
// back.  (One wave per transform storing its 8-byte values 128 bytes apart wrote 40 MB for an 11 MB image and took 25 us
// per call of 8 blocks at 1024 clients.)
template <int M>
__global__ __launch_bounds__(XLP_SEG * M / 4) void xlp_forward_kernel(const XlpArgs a) {
  constexpr uint32_t L = M / 4, NT = XLP_SEG * L;
  __shared__ v2f lds[XLP_SEG][XLP_ROW(M)];
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
  const uint32_t pass = bid / a.D, b = bid - pass * a.D;
  const uint32_t s = pass * XLP_SEG + h;
  const bool live = s < a.nseg;  // (the last pass may hold fewer segments: zeros, never read by the mix kernel)
  // branch sample n of segment s = stream sample base + (s V + n) D + b   (base: first tap of shared point 0)
  const uint32_t first = a.base + s * a.V * a.D + b;
  const uint32_t end = a.n0 + a.n1;
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
  // the transform's row, natural order (its own scratch: the LDS operations of a wave execute in order)
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[h][l + L * r] = u[0][r];
  __syncthreads();
  static_assert(XLP_XS == 16u && XLP_SEG <= XLP_XS, "image rows of 16 complex = 8 x 16 bytes");
  v4f *__restrict__ X = reinterpret_cast<v4f *>(a.X) + ((size_t)pass * a.Dpad + b) * M * (XLP_XS / 2u);
  for (uint32_t i = j; i < (uint32_t)M * (XLP_XS / 2u); i += NT) {
    const uint32_t m = i >> 3, part = i & 7u;  // row m, segments 2 part and 2 part + 1
    const v2f x0 = 2u * part < XLP_SEG ? lds[2u * part][m] : (v2f){0.0f, 0.0f};
    const v2f x1 = 2u * part + 1u < XLP_SEG ? lds[2u * part + 1u][m] : (v2f){0.0f, 0.0f};
    X[i] = (v4f){x0.x, x0.y, x1.x, x1.y};
  }
}

// ------------------------------------------------------------------------------------------- mix (the hot kernel)
// acc += r * x with ONE accumulator pair: two v_pk_fma_f32, the second negates x.im in its low half (neg_lo) --
//   acc.re += r.re*x.re;  acc.im += r.re*x.im;      acc.re += r.im*(-x.im);  acc.im += r.im*x.re
// (running the first halves of four products before their second halves, to space the dependent pairs, changed nothing
// at 1024 clients and cost 5 % at 4096: a lone wave issues a packed FMA every ~6.7 cycles whatever the spacing)
XL_DEV void xlp_cmac(v2f &acc, const v2f r, const v2f x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+v"(acc)
      : "v"(r), "v"(x));
}

// the same product with the X operand in SGPRs (wave-uniform)
XL_DEV void xlp_cmac_s(v2f &acc, const v2f r, const v2f x) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+v"(acc)
      : "v"(r), "s"(x));
}
#if XLP_SEG == 14u
// one row of the shared-spectrum image: XLP_SEG = 14 complex values = 28 dwords, fetched by scalar loads
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
struct XlpXRow {
  v16f a;  // segments 0..7
  v8f b;   // segments 8..11
  v4f c;   // segments 12, 13
};
static_assert(XLP_SEG == 14u && XLP_XS == 16u, "XlpXRow is written for rows of 14 segments in 128 bytes");
// Scalar loads return out of order and the compiler sinks them to their first use (one exposed latency per row), so the
// request and the wait are placed by hand: xlp_xrow_request() only ISSUES the three loads -- the row is not valid until
// xlp_xrow_wait(), which every later use depends on through its tied operands.  Between the two the row must not be
// touched (the compiler has no reason to: nothing else reads it).
// `pin` (a VGPR value the following multiplies read / the preceding ones wrote) keeps the compiler from moving them across.
XL_DEV void xlp_xrow_request(XlpXRow &x, const uint64_t row, v4f &pin) {
  asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x40\n\ts_load_dwordx4 %2, %4, 0x60"
               : "=&s"(x.a), "=&s"(x.b), "=&s"(x.c), "+v"(pin)
               : "s"(row));
}
XL_DEV void xlp_xrow_wait(XlpXRow &x, v2f &pin0, v2f &pin1) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x.a), "+s"(x.b), "+s"(x.c), "+v"(pin0), "+v"(pin1));
}
#define XLP_X2(ROW, I)                                                   \
  ((I) < 8    ? (v2f){(ROW).a[2 * ((I) & 7)], (ROW).a[2 * ((I) & 7) + 1]} \
   : (I) < 12 ? (v2f){(ROW).b[2 * ((I) & 3)], (ROW).b[2 * ((I) & 3) + 1]} \
              : (v2f){(ROW).c[2 * ((I) & 1)], (ROW).c[2 * ((I) & 1) + 1]})

// grid = nco_blocks + M * ncg * passes workgroups of ONE wave = (bin m, column group, pass = 14 segments); lane l =
// client columns cg*128 + 2l, 2l+1; the spectrum bin m is workgroup-uniform.  The bin's column of the shared spectra
// (Dpad rows of 14 segments, 128 bytes each) is wave-uniform: it is fetched by scalar loads, one row ahead, and multiplies
// as an SGPR operand (staged in LDS and read back as broadcast ds_read_b128 -- round 1 -- every FMA read three 64-bit VGPR
// operands: 135 instead of 119 us per call).  R is streamed through a ring of register slots, 16 bytes per lane and
// branch, ring - 1 rows ahead of the multiply (a stage-wise double buffer ran one short stage ahead and every stage
// waited out a memory latency; a ring of 14 changed nothing, a ring of 2-3 for more waves per SIMD was slower).  A block with more than 14 segments (M = 128 at the
// server default: 27) takes several passes over the same R rows: the passes of one (m, cg) sit 8 positions apart in
// the grid -- same XCD (workgroups are dealt to the XCDs round-robin), dispatched together -- so that R comes from HBM
// once and the other passes hit that XCD's L2.  (The passes as waves of one workgroup gave the same traffic but an
// uneven deal: two-wave workgroups left SIMDs with 1 to 3 waves, and the launch ends with the fullest.)
// TRIPS > 0: the branch count is TRIPS * XLP_BSTEP and the row loop is unrolled completely; 0: any count, a loop.  (A row
// waiting in SGPRs across the loop's back edge is something the compiler will not do: it parks the row in VGPRs, and the
// first row of every trip multiplies from there -- 28 more VGPRs, 14 copies per trip.  The server default, D = 42, gets
// the straight-line variant.)
template <int TRIPS>
__global__ __launch_bounds__(64) void xlp_mix_kernel(const XlpArgs a) {
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  // Workgroups are dealt to the SIMDs round-robin in blockIdx order (measured: the work waves that shared a SIMD with
  // chain wave i were blocks i + 1024 and i + 2048), and a chain wave next to TWO memory-bound work waves oversubscribes
  // the SIMD's VALU issue (both slow down, the launch ends 6-10 us late).  a.nco_skip workgroups right after position
  // a.nco_skip_at exit at once, so that the chain SIMDs' second slot stays empty and they host one work wave only.
  // (emptying the third slot as well gained nothing: the launch then ends with the SIMDs that got a third work wave)
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t lane = threadIdx.x;
  const uint32_t M = a.M;  // (a power of two >= 128: M * ncg is a multiple of 8)
  const uint32_t grp = bid / (8u * a.mix_passes), rr = bid - grp * 8u * a.mix_passes;
  const uint32_t pass = rr >> 3, pair = grp * 8u + (rr & 7u);
  const uint32_t m = pair & (M - 1u), cg = pair / M;
  // R image [cg][m][Dpad][128 columns]: the Dpad rows of a workgroup are one contiguous run (42 KB at D = 42)
  const v4f *__restrict__ Rp =
      reinterpret_cast<const v4f *>(a.R) + ((size_t)cg * M + m) * a.Dpad * (XLP_COLS / 2) + lane;
  const size_t rstride = XLP_COLS / 2;
  // (ring slots: the straight-line variant knows every row's slot at compile time and gets by with 4 -- 8 VGPRs fewer,
  // which is what takes it from 4 to 5 waves per SIMD; the loop needs a slot count that divides its trip length)
  constexpr int RING = TRIPS > 0 ? 4 : (int)XLP_BSTEP;
  v4f r[RING];
#pragma unroll
  for (int u = 0; u < RING - 1; ++u) r[u] = Rp[(size_t)u * rstride];
  // X rows of this (pass, bin): 128 bytes each (14 segments + pad), M * 128 bytes apart -- wave-uniform, so they travel
  // through the scalar cache into SGPRs (one row = s_load_dwordx16 + x8 + x4) and enter the packed FMAs as the scalar
  // operand: no LDS traffic, and an FMA reads two 64-bit VGPR operands instead of three.
  // Y image [cg][segment][sub][bin][CW columns]: the tile one inverse workgroup reads -- (segment, CW columns), all bins --
  // is one contiguous 32 KB run; this wave's 128 columns of one bin land as 128 / CW pieces of CW * 8 bytes.  (Address and
  // bounds are worked out here, before the row loop: what it keeps alive across it is then two VGPRs and two SGPRs --
  // the loop itself needs all the SGPRs it can get.)
  const uint32_t s0 = pass * XLP_SEG;
  const uint32_t CW = M == 256u ? 16u : 32u, NSUB = XLP_COLS / CW;
  const uint32_t sub = (2u * lane) / CW, cw = (2u * lane) % CW;
  v4f *__restrict__ Yp = reinterpret_cast<v4f *>(a.Y) +
                         ((((size_t)cg * a.nseg_cap + s0) * NSUB + sub) * M + m) * (CW / 2) + cw / 2;
  const uint32_t ystride = NSUB * M * (CW / 2);  // v4f per segment
  const uint32_t nvalid = a.nseg > s0 ? a.nseg - s0 : 0u;
  // (wave-uniform by construction; the readfirstlanes make it so for the register allocator as well -- a uniform value
  // it chose to compute on the VALU would otherwise reach the s_loads in VGPRs)
  const uint64_t xbase_v = (uint64_t)(uintptr_t)(a.X + ((size_t)pass * a.Dpad * M + m) * XLP_XS);
  const uint64_t xbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(xbase_v >> 32)) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)xbase_v);  // (the builtin returns int)
  const uint32_t xstride = M * XLP_XS * (uint32_t)sizeof(float2);
  XlpXRow xa, xb;
  xlp_xrow_request(xa, xbase, r[0]);  // (u below: the row's position in the ring's cycle)
  v2f acc0[XLP_SEG], acc1[XLP_SEG];
#pragma unroll
  for (int i = 0; i < (int)XLP_SEG; ++i) acc0[i] = acc1[i] = (v2f){0.0f, 0.0f};
  xlp_xrow_wait(xa, acc0[0], acc1[0]);
  // one row: request the next (clamped to the last: loaded, never used), multiply this one, wait for the next
  auto row = [&](const XlpXRow &cur, XlpXRow &nxt, const uint32_t b, const int u) __attribute__((always_inline)) {
    // (the last trips request R rows past this workgroup's: the next one's, or the tail padding the engine allocates)
    constexpr int PF = RING - 1;
    r[(u + PF) % RING] = Rp[(size_t)(b + PF) * rstride];
    const uint32_t nrows = TRIPS > 0 ? (uint32_t)TRIPS * XLP_BSTEP : a.Dpad;
    const uint32_t bn = b + 1u < nrows ? b + 1u : b;
    xlp_xrow_request(nxt, xbase + (uint32_t)__builtin_amdgcn_readfirstlane(bn * xstride), r[u % RING]);
    const v2f ra = {r[u % RING].x, r[u % RING].y}, rb = {r[u % RING].z, r[u % RING].w};
#pragma unroll
    for (int i = 0; i < (int)XLP_SEG; ++i) {
      xlp_cmac_s(acc0[i], ra, XLP_X2(cur, i));
      xlp_cmac_s(acc1[i], rb, XLP_X2(cur, i));
    }
    xlp_xrow_wait(nxt, acc0[XLP_SEG - 1], acc1[XLP_SEG - 1]);
  };
  static_assert(XLP_BSTEP % 2u == 0u, "the two row buffers alternate: an even number of rows per trip");
  if (TRIPS > 0) {
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
      for (int u = 0; u < (int)XLP_BSTEP; u += 2) {
        row(xa, xb, (uint32_t)(t * (int)XLP_BSTEP + u), t * (int)XLP_BSTEP + u);
        row(xb, xa, (uint32_t)(t * (int)XLP_BSTEP + u + 1), t * (int)XLP_BSTEP + u + 1);
      }
    }
  } else {
    for (uint32_t b0 = 0; b0 < a.Dpad; b0 += XLP_BSTEP) {
#pragma unroll
      for (int u = 0; u < (int)XLP_BSTEP; u += 2) {
        row(xa, xb, b0 + (uint32_t)u, u);
        row(xb, xa, b0 + (uint32_t)u + 1u, u + 1);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < (int)XLP_SEG; ++i)
    if ((uint32_t)i < nvalid)  // (written once, read once by the next launch: streamed past the L2 lines that hold R and X;
                               // A/B on one box: 34.2 -> 33.3 us per block at 1024 clients, 122.3 -> 120.6 at 4096)
      __builtin_nontemporal_store((v4f){acc0[i].x, acc0[i].y, acc1[i].x, acc1[i].y}, &Yp[(size_t)i * ystride]);
  xlp_trace_work(a, t_begin);
}

#endif  // XLP_SEG == 14

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
// lane ^ 1's value (DPP quad_perm [1,0,3,2]; inline assembly: see xlp_dpp_pair below for why, and for the s_nop)
XL_DEV uint32_t xlp_dpp_pair_u32(const uint32_t v) {
  uint32_t r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  return r;
}

template <int NKB, bool Y6>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void xlp_mix_mfma_kernel(const XlpArgs a) {
  // A operands of one pass: [term][k-block][lane][8 halves]; two buffers (one barrier per pass: a buffer is rewritten two
  // barriers after it was read)
  __shared__ uint4 xs[2][2][NKB][64];
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;
  const uint32_t bid = blockIdx.x;
  const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
  const uint32_t M = a.M;
  // the pass runs of one (bin, column group) sit 8 positions apart in the grid: same XCD, dispatched together -- the
  // group's operands come from HBM once (as in xlp_mix_kernel)
  const uint32_t pp = a.mix_pp, runs = (a.mix_passes + pp - 1u) / pp;
  const uint32_t grp = bid / (8u * runs), rr = bid - grp * 8u * runs;
  const uint32_t run = rr >> 3, pair = grp * 8u + (rr & 7u);
  const uint32_t m = pair & (M - 1u), cg = pair / M;
  const uint32_t p0 = run * pp, p1 = p0 + pp < a.mix_passes ? p0 + pp : a.mix_passes;
  if (p0 >= p1) return;
  // ---- B operands of this wave: 2 NKB runs of 1 KB
  const uint4 *__restrict__ Rp = reinterpret_cast<const uint4 *>(a.Rh);
  v8h r1[NKB], r2[NKB];
#pragma unroll
  for (int j = 0; j < NKB; ++j) {
    r1[j] = __builtin_bit_cast(v8h, Rp[xlm_rh_slot(cg, M, m, w, 0u, NKB, (uint32_t)j, lane)]);
    r2[j] = __builtin_bit_cast(v8h, Rp[xlm_rh_slot(cg, M, m, w, 1u, NKB, (uint32_t)j, lane)]);
  }
  const uint32_t h = lane >> 5, c = lane & 31u;
  const float cs = a.cscale[cg * XLP_COLS + w * 32u + c];
  // ---- staging role of this lane: branch 8 j + bb of k-block j = w + 4 round, segments 2 sp, 2 sp + 1 of the pass
  constexpr int ROUNDS = (NKB + 3) / 4;
  const uint32_t bb = xlm_stage_branch_in_block(lane), sp = xlm_stage_segment_pair(lane);
  const v4f *__restrict__ Xm = reinterpret_cast<const v4f *>(a.X) + (size_t)m * (XLP_XS / 2u) + sp;
  const size_t xrow = (size_t)M * (XLP_XS / 2u);  // v4f per branch row
  v4f g[ROUNDS];
  auto request = [&](const uint32_t pass) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t b = 8u * xlm_stage_kblock(w, (uint32_t)q) + bb;
      // (rows D .. Dpad - 1 of the image are zeros; beyond Dpad there is nothing to read)
      g[q] = (xlm_stage_kblock(w, (uint32_t)q) < (uint32_t)NKB && b < a.D) ? Xm[((size_t)pass * a.Dpad + b) * xrow] : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto stage = [&](const uint32_t buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      const uint32_t j = xlm_stage_kblock(w, (uint32_t)q);
      if (j < (uint32_t)NKB) {  // (wave-uniform)
        _Float16 f1[4], f2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xlp_split_h(g[q][e] * XLP_H_XSCALE, f1[e], f2[e]);
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
  stage(0u);
  if (p0 + 1u < p1) request(p0 + 1u);
  __syncthreads();
  for (uint32_t pass = p0; pass < p1; ++pass) {
    const uint32_t buf = (pass - p0) & 1u;
    v16f32 hi, lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) hi[i] = lo[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const v8h a1 = __builtin_bit_cast(v8h, xs[buf][0][j][xlm_lds_slot(lane)]);
      const v8h a2 = __builtin_bit_cast(v8h, xs[buf][1][j][xlm_lds_slot(lane)]);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, r1[j], lo, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r1[j], hi, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r2[j], lo, 0, 0, 0);
    }
    if (pass + 1u < p1) stage(buf ^ 1u);
    if (pass + 2u < p1) request(pass + 2u);
    // this lane's rows: registers g, g + 1 (g even) = (re, im) of the pass's segment xlm_result_row(g, h) / 2
    const uint32_t s0 = pass * XLP_SEG;
    if constexpr (Y6) {
      // 48 bits per value (xl_y6.h): the UNSCALED sums with a shared exponent; the reader applies the column's factor.  An even
      // lane fetches its odd neighbour's value (DPP) and stores the pair: 12 bytes, 16 lanes x 12 = 192 contiguous bytes per
      // (segment, bin).
      char *__restrict__ Yb = reinterpret_cast<char *>(a.Y);
#pragma unroll
      for (int g2 = 0; g2 < 16; g2 += 2) {
        const uint32_t sl = xlm_result_row((uint32_t)g2, h) >> 1;
        uint32_t w0, w1;
        xly6_encode(hi[g2] + lo[g2], hi[g2 + 1] + lo[g2 + 1], &w0, &w1);
        const uint32_t n0 = xlp_dpp_pair_u32(w0), n1 = xlp_dpp_pair_u32(w1);  // (lane ^ 1's)
        if (sl < XLP_SEG && s0 + sl < a.nseg && (c & 1u) == 0u) {
          char *__restrict__ t = Yb + xly6_tile(cg, a.nseg_cap, s0 + sl, NSUB, col / CW, M, CW) + xly6_pair(m, CW, (col % CW) >> 1);
          typedef uint32_t v3u __attribute__((ext_vector_type(3)));
          __builtin_nontemporal_store((v3u){w0, n0, w1 | (n1 << 16)}, reinterpret_cast<v3u *>(t));
        }
      }
    } else {
#pragma unroll
      for (int g2 = 0; g2 < 16; g2 += 2) {
        const uint32_t sl = xlm_result_row((uint32_t)g2, h) >> 1;
        const v2f y = {(hi[g2] + lo[g2]) * cs, (hi[g2 + 1] + lo[g2 + 1]) * cs};
#ifdef XLP_Y_TEMPORAL  // (tools/mall_calibration.sh: the same stores with the default cache policy)
        if (sl < XLP_SEG && s0 + sl < a.nseg) Yc[(size_t)(s0 + sl) * ystride] = y;
#else
        if (sl < XLP_SEG && s0 + sl < a.nseg) __builtin_nontemporal_store(y, &Yc[(size_t)(s0 + sl) * ystride]);
#endif
      }
    }
    __syncthreads();  // the other buffer is staged; everybody is done with this one
  }
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
    if (a.y6) {
      // 48-bit values (xl_y6.h): per bin the 12 bytes of this thread's column pair; the columns' power-of-two factors (cscale: what
      // undoes the mix's operand scales) go into the decoding's exponent
      const char *__restrict__ t = reinterpret_cast<const char *>(a.Y) + xly6_tile(cg, a.nseg_cap, s, NSUB, sub, M, CW);
      const uint32_t cbase = cg * XLP_COLS + sub * CW + 2u * part;
      const int k0 = (int)((xly6_bits(a.cscale[cbase]) >> 23) & 0xFFu) - 127, k1 = (int)((xly6_bits(a.cscale[cbase + 1u]) >> 23) & 0xFFu) - 127;
      typedef uint32_t v3u __attribute__((ext_vector_type(3)));
      v3u wp[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wp[i] = *reinterpret_cast<const v3u *>(t + xly6_pair(mrow + MR * i, CW, part));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float r0, i0, r1, i1;
        xly6_decode(wp[i].x, wp[i].z & 0xFFFFu, k0, &r0, &i0);
        xly6_decode(wp[i].y, wp[i].z >> 16, k1, &r1, &i1);
        v[i] = (v4f){r0, i0, r1, i1};
      }
    } else {
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
// the swizzled layout at five workgroups per CU: 32 KB of LDS each fit, 96 registers make the waves fit (4 spilled dwords)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void xlp_inverse_swz5_kernel(const XlpArgs a) {
  xlp_inverse_body<128, XlpPosSwz>(a);
}

// ------------------------------------------------------------------------------------------- inverse, register transform
// The M = 128 inverse launch with the transform in REGISTERS (xl_fft64.h): a wave = one Y tile = (segment, 32 client
// columns, all 128 bins); a PAIR of lanes owns a column, 64 bins each.  What the LDS version above pays for -- the tile
// transposed into LDS, three exchange passes per transform (44 % of its LDS cycles were bank conflicts), twiddles fetched
// from a table, phases expanded through LDS -- is gone:
//   * the tile is loaded straight into registers: 64 x 8-byte loads per lane, all in flight at once (lane (c, hf) reads
//     bin 2 i + hf of column c: every load instruction covers two adjacent 256-byte rows of the tile, 512 bytes back to back);
//   * the same in-place 64-point transform in both lanes (compile-time twiddles as scalar operands), then ONE exchange
//     with the partner lane (DPP quad_perm [1,0,3,2], no LDS): lane hf ends up with the shared points n = 64 hf + k of
//     its column;
//   * the NCO phases are walked in the lane itself, one step per output (the producer's own three IEEE operations per
//     step, renormalised at block ends);
//   * LDS is used once: the rotated outputs are transposed 2 x 32 shared points at a time through a wave-private
//     [32 columns][2 halves][33] buffer (odd row stride: the sixteen rows of a write group hit 16 distinct bank pairs), so
//     that the stores leave as 256-byte runs of one client's row.
// grid = nco_blocks + nseg * ncg workgroups of 256 threads; workgroup = (segment, column group), wave = sub-tile.
// The value of lane ^ 1 / lane ^ 2 of the own quad: DPP moves, written as inline assembly.  (Through the compiler's
// __builtin_amdgcn_update_dpp, two moves of the two halves of a float2 written next to each other came out as ONE move
// whose result was used for both halves -- seen twice with this toolchain, in two different spellings; the assembly leaves
// nothing to merge.  The s_nop covers the wait states a DPP read needs after a VALU write of its source, which the
// compiler only inserts for instructions it selected itself.)
XL_DEV float xlp_dpp_pair(const float f) {  // quad_perm [1,0,3,2]
  float r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(f));
  return r;
}
XL_DEV float xlp_dpp_cross(const float f) {  // quad_perm [2,3,0,1]
  float r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(f));
  return r;
}
struct XlpPairExchange {
  bool odd;  // hf == 1
  XL_MEM v2f select(const v2f z, const v2f own) const { return (v2f){odd ? z.x : own.x, odd ? z.y : own.y}; }
  template <int K>
  XL_MEM v2f partner(const v2f v) const {
    return (v2f){xlp_dpp_pair(v.x), xlp_dpp_pair(v.y)};
  }
};

// u[xl_fft64_slot(t)] (shared point n = 64 hf + t of the lane's column) <- u * 2^-7 * phase, t < 64; p = the phase of
// the lane's first output, advancing one output per t.  valid0 false: the lane's first point is nobody's output (shared
// point 0 of a column whose grid lies one step behind): p already belongs to the second one and stays at t = 0.
// Plain walk: no block of the call ends inside the lane's range (the common case; the caller has checked).
template <int T = 0>
XL_DEV void xlp_rotate_plain(v2f (&u)[64], v2f &p, const v2f inc, const bool valid0) {
  constexpr int slot = xl_fft64_slot(T);
  u[slot] = xlp_cmul_v(u[slot] * (1.0f / 128.0f), p);  // exact scaling by 2^-7, then xlating.c:70 `out = temp * phase`
  XL_FFT_PIN(u[slot]);  // (before the chain moves on: the recurrence steps are volatile asm, this product is not, and 64
                        // phases waiting for their products are 128 registers)
  const v2f q = xl_nco_next(p, inc);
  if (T == 0) p = (v2f){valid0 ? q.x : p.x, valid0 ? q.y : p.y};
  else p = q;
  if constexpr (T % 4 == 3) XL_FFT_FENCE();
  if constexpr (T + 1 < 64) xlp_rotate_plain<T + 1>(u, p, inc, valid0);
}

// Checked walk (a block of the call ends inside the range: the phase is renormalised there, xlating.c:73): the phases
// of 16 points at a time go through the wave's staging buffer (`pl`: 17 slots per lane) from a compact run-time loop
// that compares every step with the next block start; m = output index of p.
template <int CH = 0>
XL_DEV void xlp_rotate_checked(v2f (&u)[64], v2f &p, uint32_t &m, uint32_t &nb, const v2f inc, const XlBnd bnd, const bool valid0,
                               v2f *__restrict__ pl) {
#pragma unroll 1
  for (uint32_t tt = 0; tt < 16u; ++tt) {
    pl[tt] = p;
    if (CH == 0 && tt == 0u && !valid0) continue;
    p = xl_nco_next_any(p, inc, bnd.flags);
    if (++m == nb) {
      p = xl_nco_renorm(p);
      nb = xl_bnd_next(bnd, m);
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) {
    const int slot = xl_fft64_slot(CH * 16 + tt);  // (a constant after unrolling)
    u[slot] = xlp_cmul_v(u[slot] * (1.0f / 128.0f), pl[tt]);
    XL_FFT_PIN(u[slot]);
    if (tt % 4 == 3) XL_FFT_FENCE();
  }
  __builtin_amdgcn_wave_barrier();
  if constexpr (CH + 1 < 4) xlp_rotate_checked<CH + 1>(u, p, m, nb, inc, bnd, valid0, pl);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void xlp_inverse_reg_kernel(const XlpArgs a) {
  constexpr uint32_t M = 128u, CW = 32u, NSUB = XLP_COLS / CW, PR = 33u;
  __shared__ v2f stage[NSUB][CW][2][PR];   // 67584 bytes: two workgroups per CU
  __shared__ uint32_t cinfo[NSUB][CW][4];  // per column: out row, k of shared point 0 (may be -1), outputs owned, pad
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t cg = bid % a.ncg, s = bid / a.ncg;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const uint32_t c = j >> 1, hf = j & 1u;
  // ---- the tile, straight into registers: u[i] = Y[bin 2 i + hf][column c]
  v2f u[64];
  {
    // (wave-uniform tile pointer + one 32-bit lane offset: the loads address as scalar base + vector offset; 64 separate
    // 64-bit lane addresses would cost 128 registers before the first value arrives)
    const v2f *__restrict__ tile = reinterpret_cast<const v2f *>(a.Y) + (((size_t)cg * a.nseg_cap + s) * NSUB + w) * M * CW;
    const uint32_t lane_off = hf * CW + c;
#pragma unroll
    for (int i = 0; i < 64; ++i) u[i] = __builtin_nontemporal_load(tile + (size_t)i * 2u * CW + lane_off);
  }
  // ---- the column of this lane pair on the class's shared grid (xl_grid.h)
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;
  const XlpCol col = a.cols[cg * XLP_COLS + w * CW + c];
  XlBnd bnd;
  bnd.j0 = xl_merge_j0(a.j0_ref, col.delta, a.D), bnd.D = a.D, bnd.S = a.pos.S, bnd.G = a.pos.G, bnd.flags = a.pos.pad;
  bnd.K = Ka + (bnd.j0 < Nr ? 1u : 0u);
  const uint32_t shift = xl_merge_shift(a.j0_ref, col.delta, a.D);
  const int32_t k0 = (int32_t)(s * a.V) - (int32_t)shift;  // output index of shared point n = 0 of this segment
  const bool live = col.out_off != 0xFFFFFFFFu;
  if (hf == 0u) {
    cinfo[w][c][0] = col.out_off;
    cinfo[w][c][1] = (uint32_t)k0;
    cinfo[w][c][2] = live ? bnd.K : 0u;
  }
  // the lane's first output: k0 + 64 hf, or -- when that is -1 -- the next one
  const int32_t kf = k0 + (int32_t)(64u * hf);
  const bool valid0 = kf >= 0;
  const uint32_t mb = valid0 ? (uint32_t)kf : 0u;
  const bool walk = live && mb < bnd.K;
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  v2f p = ph[walk ? (col.out_off >> XL_PH_SHIFT) + (mb >> XL_PH_SHIFT) : 0u];  // (requested before the transform)
  // ---- transform
  XL_FFT_FENCE();
  xl_fft64_inverse<v2f, XlpFftOps>(u);
  {
    const XlpPairExchange ex{hf != 0u};
    xl_fft128_combine<v2f, XlpFftOps>(u, hf ? -1.0f : 1.0f, ex);
  }
  // ---- phases: from the table entry at mb rounded down to the stride up to mb, then one step per output
  const v2f inc = {col.incr.x, col.incr.y};
  uint32_t m = mb & ~(XL_PH_STRIDE - 1u);
  uint32_t nb = xl_bnd_next(bnd, m);
  if (walk) {
    for (; m < mb; ++m) {
      p = xl_nco_next_any(p, inc, bnd.flags);
      if (m + 1u == nb) {
        p = xl_nco_renorm(p);
        nb = xl_bnd_next(bnd, m + 1u);
      }
    }
  }
  // (a block of the call ends inside this lane's 64 outputs: rare -- 8 of 216 segments of the bench call -- and wave
  // uniform for all practical purposes; the checked walk does the per-step comparison the plain one leaves out)
  const bool crosses = walk && nb <= mb + 64u;
  if (__builtin_amdgcn_ballot_w64(crosses) != 0ull || (a.pos.pad & XL_POS_FMA_STEP))  // (the plain walk is the plain step)
    xlp_rotate_checked(u, p, m, nb, inc, bnd, valid0, &stage[w][0][0][0] + 17u * j);
  else
    xlp_rotate_plain(u, p, inc, valid0);
  // ---- transpose through LDS, 2 x 32 shared points at a time, and store 256-byte runs of the clients' rows
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  const uint32_t rh = j >> 5, rn = j & 31u;  // read-back duty: half rh, point rn of the chunk, one column per trip
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
    for (int tt = 0; tt < 32; ++tt) stage[w][c][hf][tt] = u[xl_fft64_slot(ch * 32 + tt)];
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = 64u * rh + (uint32_t)(ch * 32) + rn;
#pragma unroll
    for (int cc = 0; cc < (int)CW; ++cc) {
      const v2f v = stage[w][cc][rh][rn];
      const uint32_t off = cinfo[w][cc][0];
      const int32_t kk = (int32_t)cinfo[w][cc][1] + (int32_t)n;
      if (n < a.V && kk >= 0 && (uint32_t)kk < cinfo[w][cc][2]) out[(size_t)off + (uint32_t)kk] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------- inverse, register transform, quad
// The same idea with FOUR lanes per client column (32 bins each; xl_fft64.h, quad variant): 64 data registers per lane
// instead of 128, so four waves fit a SIMD (the lane-pair kernel above fits two and is latency-bound for it).
// A wave = 16 columns x 128 bins = half a Y tile; lane (c, q) reads bins 4 i + q of column c (every load instruction covers
// four 128-byte row halves), runs the 32-point transform in place, multiplies by its twiddles e^{+2 pi j q k / 128} (a
// 4 x 32 table in LDS) and exchanges twice inside its quad (DPP) -- lane q ends up with the shared points
// n = XL_QUAD_NOFF(q) + k, k < 32, of its column: a run of 32 outputs, walked with one phase step each, transposed
// 16 at a time through a wave-private [64 rows][17] LDS buffer and stored as 128-byte runs.
// grid = nco_blocks + nseg * ncg * 2 workgroups of 256 threads; workgroup = (segment, column group, half): 64 columns.
struct XlpQuadExchange {
  bool lane3;
  const v2f *tw;  // LDS: the lane's 32 twiddles
  template <int K>
  XL_MEM v2f lane_twiddle(const v2f v) const {
    return xlp_cmul_v(v, tw[K]);
  }
  XL_MEM v2f partner2(const v2f v) const { return (v2f){xlp_dpp_cross(v.x), xlp_dpp_cross(v.y)}; }  // lane ^ 2
  XL_MEM v2f partner1(const v2f v) const { return (v2f){xlp_dpp_pair(v.x), xlp_dpp_pair(v.y)}; }
  XL_MEM v2f rot_lane3(const v2f t) const { return (v2f){lane3 ? -t.y : t.x, lane3 ? t.x : t.y}; }
};

template <int T = 0>
XL_DEV void xlp_rotate_plain32(v2f (&u)[32], v2f &p, const v2f inc, const bool valid0) {
  constexpr int slot = xl_fft32_slot(T);
  u[slot] = xlp_cmul_v(u[slot] * (1.0f / 128.0f), p);  // exact scaling by 2^-7, then xlating.c:70 `out = temp * phase`
  XL_FFT_PIN(u[slot]);
  const v2f q = xl_nco_next(p, inc);
  if (T == 0) p = (v2f){valid0 ? q.x : p.x, valid0 ? q.y : p.y};
  else p = q;
  if constexpr (T % 4 == 3) XL_FFT_FENCE();
  if constexpr (T + 1 < 32) xlp_rotate_plain32<T + 1>(u, p, inc, valid0);
}

template <int CH = 0>
XL_DEV void xlp_rotate_checked32(v2f (&u)[32], v2f &p, uint32_t &m, uint32_t &nb, const v2f inc, const XlBnd bnd, const bool valid0,
                                 v2f *__restrict__ pl) {
#pragma unroll 1
  for (uint32_t tt = 0; tt < 16u; ++tt) {
    pl[tt] = p;
    if (CH == 0 && tt == 0u && !valid0) continue;
    p = xl_nco_next_any(p, inc, bnd.flags);
    if (++m == nb) {
      p = xl_nco_renorm(p);
      nb = xl_bnd_next(bnd, m);
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) {
    const int slot = xl_fft32_slot(CH * 16 + tt);  // (a constant after unrolling)
    u[slot] = xlp_cmul_v(u[slot] * (1.0f / 128.0f), pl[tt]);
    XL_FFT_PIN(u[slot]);
    if (tt % 4 == 3) XL_FFT_FENCE();
  }
  __builtin_amdgcn_wave_barrier();
  if constexpr (CH + 1 < 2) xlp_rotate_checked32<CH + 1>(u, p, m, nb, inc, bnd, valid0, pl);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void xlp_inverse_quad_kernel(const XlpArgs a) {
  constexpr uint32_t M = 128u, CW = 32u, NSUB = XLP_COLS / CW, PR = 17u, WC = 16u;  // WC: columns per wave
  __shared__ v2f stage[4][64][PR];        // 34816 bytes: four workgroups per CU
  __shared__ v2f tw[4][32];               // e^{+2 pi j q k / 128}
  __shared__ uint32_t cinfo[4][WC][4];    // per column: out row, k of shared point 0 (may be -1), outputs owned, pad
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t hb = bid & 1u, cg = (bid >> 1) % a.ncg, s = (bid >> 1) / a.ncg;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const uint32_t sub = 2u * hb + (w >> 1), hw = w & 1u;  // sub-tile of 32 columns, and which 16 of them
  const uint32_t c = j >> 2, q = j & 3u;
  // ---- the wave's half tile, straight into registers: u[i] = Y[bin 4 i + q][column 16 hw + c]
  v2f u[32];
  {
    const v2f *__restrict__ tile = reinterpret_cast<const v2f *>(a.Y) + (((size_t)cg * a.nseg_cap + s) * NSUB + sub) * M * CW;
    const uint32_t lane_off = q * CW + WC * hw + c;
#pragma unroll
    for (int i = 0; i < 32; ++i) u[i] = __builtin_nontemporal_load(tile + (size_t)i * 4u * CW + lane_off);
  }
  if (threadIdx.x < 128u) {
    const uint32_t e = ((threadIdx.x >> 5) * (threadIdx.x & 31u)) & 127u;
    tw[threadIdx.x >> 5][threadIdx.x & 31u] = (v2f){xl_w128_cos((int)e), xl_w128_sin((int)e)};
  }
  // ---- the column of this lane quad on the class's shared grid (xl_grid.h)
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;
  const XlpCol col = a.cols[cg * XLP_COLS + sub * CW + WC * hw + c];
  XlBnd bnd;
  bnd.j0 = xl_merge_j0(a.j0_ref, col.delta, a.D), bnd.D = a.D, bnd.S = a.pos.S, bnd.G = a.pos.G, bnd.flags = a.pos.pad;
  bnd.K = Ka + (bnd.j0 < Nr ? 1u : 0u);
  const uint32_t shift = xl_merge_shift(a.j0_ref, col.delta, a.D);
  const int32_t k0 = (int32_t)(s * a.V) - (int32_t)shift;  // output index of shared point n = 0 of this segment
  const bool live = col.out_off != 0xFFFFFFFFu;
  if (q == 0u) {
    cinfo[w][c][0] = col.out_off;
    cinfo[w][c][1] = (uint32_t)k0;
    cinfo[w][c][2] = live ? bnd.K : 0u;
  }
  const uint32_t noff = XL_QUAD_NOFF(q);
  const int32_t kf = k0 + (int32_t)noff;  // the lane's first output, or -- when that is -1 -- the next one
  const bool valid0 = kf >= 0;
  const uint32_t mb = valid0 ? (uint32_t)kf : 0u;
  const bool walk = live && mb < bnd.K;
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  v2f p = ph[walk ? (col.out_off >> XL_PH_SHIFT) + (mb >> XL_PH_SHIFT) : 0u];  // (requested before the transform)
  __syncthreads();  // (the twiddle table)
  // ---- transform
  XL_FFT_FENCE();
  xl_fft32_inverse<v2f, XlpFftOps>(u);
  {
    const XlpQuadExchange ex{q == 3u, &tw[q][0]};
    xl_fft128_combine_quad(u, (q & 2u) ? -1.0f : 1.0f, (q & 1u) ? -1.0f : 1.0f, ex);
  }
  // ---- phases: from the table entry at mb rounded down to the stride up to mb, then one step per output
  const v2f inc = {col.incr.x, col.incr.y};
  uint32_t m = mb & ~(XL_PH_STRIDE - 1u);
  uint32_t nb = xl_bnd_next(bnd, m);
  if (walk) {
    for (; m < mb; ++m) {
      p = xl_nco_next_any(p, inc, bnd.flags);
      if (m + 1u == nb) {
        p = xl_nco_renorm(p);
        nb = xl_bnd_next(bnd, m + 1u);
      }
    }
  }
  const bool crosses = walk && nb <= mb + 32u;
  if (__builtin_amdgcn_ballot_w64(crosses) != 0ull || (a.pos.pad & XL_POS_FMA_STEP))  // (the plain walk is the plain step)
    xlp_rotate_checked32(u, p, m, nb, inc, bnd, valid0, &stage[w][0][0] + PR * j);
  else
    xlp_rotate_plain32(u, p, inc, valid0);
  // ---- transpose through LDS, 16 points per lane at a time, and store 128-byte runs of the clients' rows
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  const uint32_t rr = j >> 4, rk = j & 15u;  // read-back duty: row 4 it + rr (= column it, quad lane rr), point rk
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) stage[w][j][tt] = u[xl_fft32_slot(ch * 16 + tt)];
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = XL_QUAD_NOFF(rr) + (uint32_t)(ch * 16) + rk;
#pragma unroll
    for (int cc = 0; cc < (int)WC; ++cc) {
      const v2f v = stage[w][4 * cc + rr][rk];
      const uint32_t off = cinfo[w][cc][0];
      const int32_t kk = (int32_t)cinfo[w][cc][1] + (int32_t)n;
      if (n < a.V && kk >= 0 && (uint32_t)kk < cinfo[w][cc][2]) out[(size_t)off + (uint32_t)kk] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------- branch spectra
// R[cg][m][b][col] = sum_{a<A} r'_col[D a + b] e^{+2 pi j a m / M}, r' = the column's taps delayed by its grid offset
// (xl_grid.h), in double, rounded once.  For a LIST of columns (all of them when a class is built, the newcomers' when a
// client joins): thread j of block (m, b) handles list entry j -- column colidx[j], taps rt[.][j], delay delta[j].
__global__ __launch_bounds__(XLP_COLS) void xlp_tables_kernel(const float2 *__restrict__ rt,
                                                              const uint32_t *__restrict__ delta,
                                                              const uint32_t *__restrict__ colidx, uint32_t nlist,
                                                              uint32_t T, uint32_t D, uint32_t Dpad, uint32_t A,
                                                              uint32_t M, float2 *__restrict__ R) {
  __shared__ double wc[256], ws[256];  // e^{+2 pi j n / M} in double
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
  R[(((size_t)cg * M + m) * Dpad + b) * XLP_COLS + cl] = make_float2((float)sr, (float)si);
}

// The same values in the matrix-core mix's operand form (xlp_mix_mfma_kernel): (R.re, -R.im) * scale[j] -- a power of two, so the
// float32 value is the one above with another exponent -- as two halves each; branch b of column cl = 32 w + c lands in
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

hipError_t xlp_launch_tables(const float2 *rt, const uint32_t *delta, const uint32_t *colidx, uint32_t nlist, uint32_t T,
                             uint32_t D, uint32_t Dpad, uint32_t A, uint32_t M, float2 *R, hipStream_t s) {
  if (!xlp_valid_m(M) || nlist == 0u) return hipErrorInvalidValue;
  hipLaunchKernelGGL(xlp_tables_kernel, dim3(M * Dpad, (nlist + XLP_COLS - 1u) / XLP_COLS), dim3(XLP_COLS), 0, s, rt, delta, colidx,
                     nlist, T, D, Dpad, A, M, R);
  return hipGetLastError();
}

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
  if (a.y6) hipLaunchKernelGGL((xlp_mix_mfma_kernel<NKB, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((xlp_mix_mfma_kernel<NKB, false>), grid, dim3(256), 0, s, a);
}

hipError_t xlp_launch_mix(const XlpArgs &a0, hipStream_t s) {
  if (!xlp_valid_m(a0.M)) return hipErrorInvalidValue;
  const uint32_t passes = (a0.nseg + XLP_SEG - 1) / XLP_SEG;
  if (a0.mix_kind == 1u) {
    if (a0.nkb == 0u || a0.nkb > XLP_NKB_MAX || a0.D > 8u * a0.nkb || a0.Rh == nullptr || a0.cscale == nullptr ||
        a0.fmt == XLF_CF32 || a0.nco_blocks != 0u)  // (no NCO role next to matrix instructions: see the kernel)
      return hipErrorInvalidValue;
    XlpArgs a = a0;
    a.nco_skip = 0u;  // (the skipped positions are the one-wave kernel's device)
    a.nco_skip_at = 0xFFFFFFFFu;
    a.mix_passes = passes;
    // (all passes of an 8-block call in one workgroup: the operands are fetched once; A/B at 4096 clients, passes per
    // workgroup 4 / 8 / 16: 42.5 / 38.5 / 34.8 us per block, at 1024 clients 10.3 / 9.3 / 10.0)
    if (a.mix_pp == 0u) a.mix_pp = 16u;
    const uint32_t runs = (passes + a.mix_pp - 1u) / a.mix_pp;
    const dim3 grid(a.M * a.ncg * runs);
    switch (a.nkb) {
      case 1: xlp_launch_mix_mfma_n<1>(a, grid, s); break;
      case 2: xlp_launch_mix_mfma_n<2>(a, grid, s); break;
      case 3: xlp_launch_mix_mfma_n<3>(a, grid, s); break;
      case 4: xlp_launch_mix_mfma_n<4>(a, grid, s); break;
      case 5: xlp_launch_mix_mfma_n<5>(a, grid, s); break;
      case 6: xlp_launch_mix_mfma_n<6>(a, grid, s); break;
      case 7: xlp_launch_mix_mfma_n<7>(a, grid, s); break;
      default: xlp_launch_mix_mfma_n<8>(a, grid, s); break;
    }
    return hipGetLastError();
  }
  if (a0.y6) return hipErrorInvalidValue;  // (the 48-bit form of Y is the matrix-core mix's: its sums are bounded by construction)
  if (a0.mix_kind == 3u) {  // matrix cores, float32 operands (xl_mixf32.hip)
    XlpArgs a = a0;
    a.mix_passes = passes;
    return xlp_launch_mix_f32(a, s);
  }
  const uint32_t work = a0.M * a0.ncg * passes;
  XlpArgs a = xlp_checked_skip(a0, work);
  a.mix_passes = passes;
  const dim3 grid(a.nco_blocks + a.nco_skip + work);
#if XLP_SEG == 14u
  if (a.Dpad == 7u * XLP_BSTEP) hipLaunchKernelGGL(xlp_mix_kernel<7>, grid, dim3(64), 0, s, a);
  else hipLaunchKernelGGL(xlp_mix_kernel<0>, grid, dim3(64), 0, s, a);
  return hipGetLastError();
#else
  (void)grid;
  return hipErrorInvalidValue;
#endif
}

// `done` (optional): recorded with the launch's own completion signal -- one queue packet instead of launch + event record
hipError_t xlp_launch_inverse(const XlpArgs &a0, hipStream_t s, hipEvent_t done) {
  if (!xlp_valid_m(a0.M)) return hipErrorInvalidValue;
  // M = 128: 0 = transform staged in LDS (workgroup = one 32-column tile), 1 = registers, lane pair per column (workgroup
  // = (segment, column group): four tiles), 2 = registers, lane quad per column (workgroup = two tiles)
  // 3 = staged in LDS like 0, dense rows with an XOR swizzle instead of the pad; 4 = the same built for five workgroups per CU
  // 5 = eight lanes per column, transforms of 16 and 8 points in registers (xl_inv8.hip)
  const uint32_t kind = a0.M == 128u ? a0.inv_reg : 0u;
  if (kind == 5u && a0.y6) return hipErrorInvalidValue;
  if (a0.y6 && (kind == 1u || kind == 2u || a0.cscale == nullptr)) return hipErrorInvalidValue;  // (the register-transform kernels read float32 pairs)
  uint32_t work = a0.nseg * a0.ncg * (kind == 1u ? 1u : (kind == 2u ? 2u : (a0.M == 256u ? 8u : 4u)));
  XlpArgs a1 = a0;
  if (kind != 5u || a1.inv_wgs >= work) a1.inv_wgs = 0u;  // (persistent form: the 8-lane kernel only, and only with more tiles than workgroups)
  if (a1.inv_wgs) work = a1.inv_wgs;
  const XlpArgs a = xlp_checked_skip(a1, work);
  const dim3 grid(a.nco_blocks + a.nco_skip + work);
  if (kind == 5u) {
    xlp_inverse8_launch(a, grid, s, done);
    return hipGetLastError();
  }
  void (*kern)(const XlpArgs) = kind == 1u   ? xlp_inverse_reg_kernel
                                : kind == 2u ? xlp_inverse_quad_kernel
                                : kind == 3u ? xlp_inverse_kernel<128, XlpPosSwz>
                                : kind == 4u ? xlp_inverse_swz5_kernel
                                             : (a.M == 256u ? xlp_inverse_kernel<256> : xlp_inverse_kernel<128>);
  if (done) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}
