// xl_polyphase.hip -- polyphase overlap-save evaluation of the frequency-xlating FIR (see xl_polyphase.h for the
// algebra and why it is the same operator as /root/reference/src/xlating.c:52-72).  Hand-written for gfx950.
//
// Three launches per call and class (stream order is the only synchronisation):
//   xlp_forward_kernel  one workgroup per (pass of 16 segments, branch): raw samples -> cf32 (xlating.c:357-378, exact)
//                       -> M-point DFT of the branch per segment -> shared spectra X[pass][b][m][s], stored as whole rows.
//   xlp_mix_mfma_kernel (xl_mixh.hip; 9 .. 14 k-blocks: xl_mixh2.hip) Y[c][s][m] = sum_b X[s][b][m] * R[c][b][m] on the matrix cores: one real matrix product per bin m (rows =
//                       (segment, re / im), columns = clients, k = (branch, re / im)) with every float32 operand carried as two
//                       halves, v_mfma_f32_32x32x16_f16, FP32 accumulation.  Integer input formats, D <= 64 (the default there).
//   xlp_mix_f32_kernel  (xl_mixf32.hip) the same sums with float32 operands on v_mfma_f32_32x32x2_f32 -- the float32 FMA chain itself:
//                       cf32 input, D > 64, and every class on request (option "mix_kernel" = 3).
//   xlp_inverse8_kernel (xl_inv8.hip) / xlp_inverse32_kernel (xl_inv32.hip) (128-point classes: small / big launches) /
//   xlp_inverse_kernel  (256-point classes; 128-point ones on request): per (segment, 32 or 16 columns): Y tile -> M-point inverse DFT
//                       per column -> scale, NCO rotate (xlating.c:70) with the tabulated float32 phase -> out[k], k < K.
// M = 256, 128 or 64 per class (xl_polyphase.h).  When the NCO phases of the next call are not tabulated by the side-stream
// chain kernel (xl_kernels.hip), the forward and the inverse launch each carry a slice of that recurrence ("NCO role").
// Rounds 1-4 also shipped a packed-FMA mix kernel, a fused mix + inverse launch, 48-bit mixed spectra and three more inverse
// kernels: measured, documented (profiles/r03_*, r04_*), and retired in round 5 (tools/experiments/retired/).
#include "xl_polyphase.h"

#include "xl_poly_dev.h"
#include "xl_mix_layout.h"

#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------- forward transforms
// grid = nco_blocks + passes * ceil(D / NB) transform workgroups + a.roll_blocks history-roll workgroups.  A transform workgroup =
// (pass, group of NB ADJACENT branches): the XLP_SEG = 16 segments of the pass on M / 4 lanes each (16 * M / 4 threads: 512 or 1024),
// every lane running the NB branches' transforms of its segment side by side.  NB = 1 for the integer input formats; cf32 streams of
// at least XLP_FWD_GROUP_MIN_WGS grouped workgroups: NB = 4 (M = 128) or 2 (M = 256), 80 KB of LDS.
//
// Why groups of branches for cf32 (round 6, second session): branch b of a segment is every D-th sample, so a lane's four points lie D
// samples apart -- one cache line per lane and point --, but the NB branches' samples of one point are NB adjacent samples: ONE load of
// NB x 8 bytes.  With one branch per workgroup a call of 8 cf32 blocks asked the L2s for 1.2 M lines of which it used 8 bytes each
// (157 MB of L2 -> CU traffic for an 8.4 MB super-block), and the load phase was 5 us median / 9 us p90 of a 13.7 us launch by the
// kernel's own clock stamps (profiles/r06_forward_anatomy.txt); four branches per load: a quarter of the requests, a quarter of the
// workgroups, 13.7 -> 10.7 us (profiles/r06_forward_groups.txt).  The 2-byte formats gain nothing (a line holds 64 samples >= D: the
// single-branch form already hits L1 / L2 with most requests; their load phase is 3.5-4 us either way) and lose the parallelism that
// short calls need (one block per call: 84 workgroups -> 22, 8.7 -> 13.5 us): they keep NB = 1.
// The spectra go through LDS once more so that the image rows X[pass][b][m][0..15] -- what the mix kernel fetches as one 128-byte
// row -- leave as whole lines: 8 lanes x 16 bytes per row, a branch's 16 KB (M = 128) back to back.
//
// NB adjacent samples i .. i + NB - 1 of a raw buffer -> cf32 (xl_sample's maps, xlating.c:357-378: exact), one load (the address is
// aligned to the SAMPLE, not to the load: global loads take any alignment on this target)
typedef uint32_t xlp_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t xlp_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t xlp_u1_a2 __attribute__((aligned(2)));
typedef xlp_u2 xlp_u2_a2 __attribute__((aligned(2)));
typedef xlp_u2 xlp_u2_a4 __attribute__((aligned(4)));
typedef xlp_u4 xlp_u4_a4 __attribute__((aligned(4)));
typedef xlp_u4 xlp_u4_a8 __attribute__((aligned(8)));
template <int NB>
XL_DEV void xlp_samples(const void *__restrict__ p, const int fmt, const uint32_t i, v2f (&out)[NB]) {
  static_assert(NB == 1 || NB == 2 || NB == 4, "one, two or four adjacent samples");
  if (NB == 1) {
    out[0] = xl_sample(p, fmt, i);
    return;
  }
  uint32_t w[2 * NB];  // the raw bytes, as much as the format needs
  if (fmt == XLF_CU8 || fmt == XLF_CS8) {
    const char *q = reinterpret_cast<const char *>(p) + 2u * (size_t)i;
    if (NB == 4) {
      const xlp_u2 v = *reinterpret_cast<const xlp_u2_a2 *>(q);
      w[0] = v.x, w[1] = v.y;
    } else {
      w[0] = *reinterpret_cast<const xlp_u1_a2 *>(q);
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const uint32_t v = (w[n >> 1] >> (16 * (n & 1))) & 0xFFFFu;
      if (fmt == XLF_CU8) {
        out[n].x = ((float)(v & 0xFFu) - 127.5f) / 128.0f;
        out[n].y = ((float)(v >> 8) - 127.5f) / 128.0f;
      } else {
        out[n].x = (float)((int32_t)(int8_t)(v & 0xFFu)) / 128.0f;
        out[n].y = (float)((int32_t)(int8_t)(v >> 8)) / 128.0f;
      }
    }
  } else if (fmt == XLF_CS16) {
    const char *q = reinterpret_cast<const char *>(p) + 4u * (size_t)i;
    if (NB == 4) {
      const xlp_u4 v = *reinterpret_cast<const xlp_u4_a4 *>(q);
      w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
      const xlp_u2 v = *reinterpret_cast<const xlp_u2_a4 *>(q);
      w[0] = v.x, w[1] = v.y;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int32_t v = (int32_t)w[n];
      out[n].x = (float)((int32_t)(int16_t)(v & 0xFFFF)) / 32768.0f;
      out[n].y = (float)(v >> 16) / 32768.0f;
    }
  } else {
    const char *q = reinterpret_cast<const char *>(p) + 8u * (size_t)i;
#pragma unroll
    for (int k = 0; k < NB / 2; ++k) {
      const xlp_u4 v = reinterpret_cast<const xlp_u4_a8 *>(q)[k];
      w[4 * k] = v.x, w[4 * k + 1] = v.y, w[4 * k + 2] = v.z, w[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) out[n] = (v2f){__uint_as_float(w[2 * n]), __uint_as_float(w[2 * n + 1])};
  }
}

template <int M, int NB>
__global__ __launch_bounds__(XLP_SEG * M / 4) void xlp_forward_kernel(const XlpArgs a) {
  constexpr uint32_t L = M / 4, NT = XLP_SEG * L;
  static_assert(XLP_SEG * NB * XLP_ROW(M) * 8 <= 80 * 1024, "two workgroups' rows in a CU's LDS");
  __shared__ v2f lds[XLP_SEG][NB][XLP_ROW(M)];
  // (cf32 streams on the two-half mix: float bits of the largest |component| of each segment's transforms -- in the one element of a
  // segment's first row that XLP_POS never addresses, so that two workgroups' 80 KB fit a CU's LDS exactly)
  static_assert(XLP_POS(M - 1) < XLP_ROW(M) - 1, "the last element of a padded row is free");
  auto tmax = [&](const uint32_t seg) __attribute__((always_inline)) -> uint32_t & {
    return reinterpret_cast<uint32_t *>(&lds[seg][0][XLP_ROW(M) - 1])[0];
  };
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  const unsigned long long t_begin = a.trace ? wall_clock64() : 0ull;  // (tuning builds: XL_EXP_POLY_TRACE_FWD)
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t j = threadIdx.x;
  const uint32_t passes = (a.nseg + XLP_SEG - 1u) / XLP_SEG;
  const uint32_t ngrp = (a.D + (uint32_t)NB - 1u) / (uint32_t)NB;
  const uint32_t nwg = passes * ngrp;
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
  const uint32_t h = j / L, l = j % L;  // segment of the pass this lane works on, lane within its transforms
  const XlpTw tw = xlp_twiddles<-1, M>(reinterpret_cast<const v2f *>(a.W), l);
  // Which (pass, group): group-major over the XCDs.  Workgroup bid runs on XCD bid % 8; the transform workgroups of XCD x are given a
  // CONTIGUOUS range of the group-major list (group, pass), so an XCD works on neighbouring branches -- with wide samples (cf32: 16 per
  // 128-byte line, D = 100 branches per period) it then pulls a fraction of the block's lines through its L2 instead of all of them.
  uint32_t pass, grp;
  {
#ifdef XLP_EXP_FWD_PASS_MAJOR
    pass = bid / ngrp, grp = bid - pass * ngrp;
#else
    const uint32_t x = bid & 7u, kx = bid >> 3;
    uint32_t start = 0u;  // workgroups of the XCDs below x: XCD y holds the bids y, y + 8, .. < nwg
    for (uint32_t y = 0u; y < x; ++y) start += (nwg - y + 7u) >> 3;
    const uint32_t jj = start + kx;
    grp = jj / passes, pass = jj - grp * passes;
#endif
  }
  const uint32_t b0 = grp * (uint32_t)NB;  // branches b0 .. b0 + NB - 1 (those >= D: computed from real samples, never stored)
  const uint32_t s = pass * XLP_SEG + h;
  const bool live = s < a.nseg;  // (the last pass may hold fewer segments: zeros, never read by the mix kernel)
  // branch sample n of segment s = stream sample base + (s V + n) D + b   (base: first tap of shared point 0)
  const uint32_t first = a.base + s * a.V * a.D + b0;
  const uint32_t end = a.n0 + a.n1;
  // (two-half mix of a cf32 stream: what the maximum of segment pass * 16 + j stands at, read by thread j < 16 ahead of the transforms
  // -- stale is fine: one entry per 128-byte line (XLP_SEGMAX_STRIDE), and a global atomic only where this workgroup raises what it saw)
  uint32_t *const smax = (a.segmax != nullptr && j < XLP_SEG && pass * XLP_SEG + j < a.nseg)
                             ? a.segmax + ((size_t)a.seg_par * a.seg_cap + pass * XLP_SEG + j) * XLP_SEGMAX_STRIDE : nullptr;
  const uint32_t seen = smax != nullptr ? __hip_atomic_load(smax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if (l == 0u) tmax(h) = 0u;
  v2f u[NB][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t idx = first + (l + L * r) * a.D;  // sample of branch b0; branch b0 + n: idx + n
    v2f v[NB];
    // all NB samples real and in ONE of the two buffers: one load.  Else (the window's edges: below the late joiners' zero line, across
    // history | block, past the block's end -- those outputs lie beyond K and are never stored) sample by sample.
#ifdef XLP_EXP_FWD_NOLOAD  // (anatomy: wrong results)
#pragma unroll
    for (int n = 0; n < NB; ++n) v[n] = (v2f){(float)(idx + n), 1.0f};
#else
    const bool whole = NB > 1 && live && idx >= a.zero_below && idx + (uint32_t)NB <= end && (idx + (uint32_t)NB <= a.n0 || idx >= a.n0);
    if (whole) {
      const bool lo = idx < a.n0;
      xlp_samples<NB>(lo ? a.in0 : a.in1, (int)a.fmt, lo ? idx : idx - a.n0, v);
    } else {
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const uint32_t in = idx + (uint32_t)n;
        const bool ok = live && in >= a.zero_below && in < end;
        const bool lo = in < a.n0;
        const v2f t = xl_sample((lo || !ok) ? a.in0 : a.in1, (int)a.fmt, ok ? (lo ? in : in - a.n0) : 0u);
        v[n] = ok ? t : (v2f){0.0f, 0.0f};
      }
    }
#endif
#pragma unroll
    for (int n = 0; n < NB; ++n) u[n][r] = v[n];
  }
  unsigned long long t_loaded = 0ull;
  if (a.trace) {
    __builtin_amdgcn_s_waitcnt(0);  // (tuning only: samples and twiddles have arrived)
    t_loaded = wall_clock64();
  }
  {
    v2f *bufs[NB];
    uint32_t rs0[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bufs[n] = lds[h][n], rs0[n] = 0u;
    v2f *const(&cb)[NB] = bufs;
    const uint32_t(&crs)[NB] = rs0;
#ifndef XLP_EXP_FWD_NODFT  // (anatomy: wrong results)
    xlp_dft<-1, NB, M>(u, cb, tw, l, crs);
#endif
  }
  if (a.segmax != nullptr) {
    // cf32 stream on the two-half mix: the segment's largest spectrum component, over all branches -- these transforms' share of it
    // (NaNs drop out of fmaxf: a stream that carries them has no parity to keep), gathered with one LDS atomic per lane (the lanes of a
    // transform sit in one wave, whose LDS operations execute in order: the clear above needs no barrier)
    float mx = 0.0f;
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (b0 + (uint32_t)n < a.D) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fmaxf(fabsf(u[n][r].x), fabsf(u[n][r].y)));
      }
    atomicMax(&tmax(h), __float_as_uint(mx));
  }
  // the transforms' rows, natural order (their own scratch: the LDS operations of a wave execute in order)
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[h][n][l + L * r] = u[n][r];
  __syncthreads();
  if (smax != nullptr && tmax(j) > seen) atomicMax(smax, tmax(j));
  if (a.segmax != nullptr && bid == 0u)  // the next call's buffer (last read by the previous call's mix launch)
    for (uint32_t i = j; i < a.seg_cap; i += NT) a.segmax[((size_t)(a.seg_par ^ 1u) * a.seg_cap + i) * XLP_SEGMAX_STRIDE] = 0u;
  static_assert(XLP_XS == 16u && XLP_SEG <= XLP_XS, "image rows of 16 complex = 8 x 16 bytes");
  unsigned long long t_xf = 0ull;
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    if (b0 + (uint32_t)n >= a.D) break;  // (workgroup-uniform; rows D .. Dpad - 1 of the image stay zero)
    v4f *__restrict__ X = reinterpret_cast<v4f *>(a.X) + ((size_t)pass * a.Dpad + b0 + (uint32_t)n) * M * (XLP_XS / 2u);
    for (uint32_t i = j; i < (uint32_t)M * (XLP_XS / 2u); i += NT) {
      const uint32_t m = i >> 3, part = i & 7u;  // row m, segments 2 part and 2 part + 1
      const v2f x0 = 2u * part < XLP_SEG ? lds[2u * part][n][m] : (v2f){0.0f, 0.0f};
      const v2f x1 = 2u * part + 1u < XLP_SEG ? lds[2u * part + 1u][n][m] : (v2f){0.0f, 0.0f};
#ifdef XLP_EXP_FWD_NOSTORE  // (anatomy: wrong results)
      if (x0.x == 1.2345e-30f)
#endif
      X[i] = (v4f){x0.x, x0.y, x1.x, x1.y};
    }
  }
  if (a.trace && bid < 6000u) {  // tuning: start, end (stores issued and acknowledged), samples loaded, transforms in LDS
    t_xf = wall_clock64();
    __builtin_amdgcn_s_waitcnt(0);
    if (j == 0u) {
      unsigned long long *t = a.trace + 4096 + 4 * (size_t)bid;
      t[0] = t_begin;
      t[1] = wall_clock64();
      t[2] = t_loaded;
      t[3] = t_xf;
    }
  }
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
static bool xlp_valid_m(uint32_t M) { return M == 64u || M == 128u || M == 256u; }

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
  const uint32_t nb = a.M == 256u ? 2u : 4u;  // branches per workgroup of the grouped form (two: level, profiles/r06_forward_groups.txt)
  const uint32_t ngrp = (a.D + nb - 1u) / nb;
  const bool grouped = a.fmt == XLF_CF32 && passes * ngrp >= XLP_FWD_GROUP_MIN_WGS;
  const dim3 grid(a.nco_blocks + passes * (grouped ? ngrp : a.D) + a.roll_blocks);
  if (a.M == 256u) {
    if (grouped) hipLaunchKernelGGL((xlp_forward_kernel<256, 2>), grid, dim3(XLP_SEG * 64u), 0, s, a);
    else hipLaunchKernelGGL((xlp_forward_kernel<256, 1>), grid, dim3(XLP_SEG * 64u), 0, s, a);
  } else if (a.M == 64u) {
    if (grouped) hipLaunchKernelGGL((xlp_forward_kernel<64, 4>), grid, dim3(XLP_SEG * 16u), 0, s, a);
    else hipLaunchKernelGGL((xlp_forward_kernel<64, 1>), grid, dim3(XLP_SEG * 16u), 0, s, a);
  } else {
    if (grouped) hipLaunchKernelGGL((xlp_forward_kernel<128, 4>), grid, dim3(XLP_SEG * 32u), 0, s, a);
    else hipLaunchKernelGGL((xlp_forward_kernel<128, 1>), grid, dim3(XLP_SEG * 32u), 0, s, a);
  }
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

// `done` (optional): recorded with the launch's own completion signal -- one queue packet instead of launch + event record
hipError_t xlp_launch_inverse(const XlpArgs &a0, hipStream_t s, hipEvent_t done) {
  if (!xlp_valid_m(a0.M)) return hipErrorInvalidValue;
  // 128-point classes: eight lanes per column, transforms of 16 and 8 points in registers (xl_inv8.hip), the 32 x 4 cut (xl_inv32.hip),
  // or staged in LDS on dense XOR-swizzled rows -- xlp_inverse_pick() says which; 256-point classes: staged in LDS on padded rows.
  // Workgroup = one tile of 32 (16) columns.
  const uint32_t tiles = a0.nseg * a0.ncg * (XLP_COLS / xlp_tile_columns(a0.M));
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
  void (*kern)(const XlpArgs) = a.M == 256u ? xlp_inverse_kernel<256, XlpPosPad>
                                : (a.M == 64u ? xlp_inverse_kernel<64, XlpPosPad> : xlp_inverse_kernel<128, XlpPosSwz>);
  if (done) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}
