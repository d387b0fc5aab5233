// xl_polyphase.hip -- polyphase overlap-save evaluation of the frequency-xlating FIR (see xl_polyphase.h for the
// algebra and why it is the same operator as /root/reference/src/xlating.c:52-72).  Hand-written for gfx950.
//
// Three launches per block and class (stream order is the only synchronisation):
//   xlp_forward_kernel  one wave per (segment, branch): raw samples -> cf32 (xlating.c:357-378, exact) -> 256-point
//                       DFT of the branch -> shared spectra X[pass][b][m][s].  D * nseg small transforms: ~1 MB.
//   xlp_mix_kernel      Y[c][s][m] = sum_b X[s][b][m] * R[c][b][m].  lane = two client columns, the bin m is
//                       workgroup-uniform: its column of X is staged in LDS once and broadcast-read row by row, R is
//                       streamed exactly once, coalesced (16 bytes per lane and branch).  HBM-bound on R:
//                       8 * D * M bytes per client and block.
//   xlp_inverse_kernel  per (segment, 16 columns): Y tile -> LDS (transposed) -> 256-point inverse DFT per column ->
//                       scale, NCO rotate (xlating.c:70) with the tabulated float32 phase -> out[k], k < K.
// Each launch also carries a slice of the NEXT block's NCO phase recurrence (a ~57 us dependent chain per block
// that would otherwise serialise with these short kernels).
#include "xl_polyphase.h"

#include "xl_dev_inline.h"

XL_DEV v2f xlp_cmul(const v2f a, const v2f b) {
  return (v2f){__builtin_fmaf(-a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y)};
}

// 256-point DFT by one wave: radix-4 Stockham autosort, passes p = 1, 4, 16, 64; lane j holds points j + 64 r.
// In: u[r] = x[j + 64 r].  Out: u[r] = X[j + 64 r] (natural order).  SIGN -1 forward, +1 inverse (unnormalised).
// The twiddles of a lane depend only on (pass, r, j): xlp_twiddles() fetches the nine of them once (one exposed
// global latency instead of three), e^{-2 pi j n / 256} from the table W, conjugated for the inverse.
// `lds` = XLP_ROW complex owned by this wave, addressed through XLP_POS (may be the input row itself); LDS operations of one wave execute in
// order, so no barrier is needed between a pass's scatter and the next gather.
struct XlpTw {
  v2f w[3][3];  // [pass - 1][r - 1]
};
// LDS position of transform element i: one pad element per four.  The scatter of pass p writes elements
// jo + r p with jo = 4 (j - j % p) + j % p -- strides of 4, 16, 64 elements of 8 bytes across lanes, a 4- to 16-way
// bank conflict on a dense row (measured: the inverse kernel spent ~20 of its 28 us there); with the pad the 16
// lanes of a quarter-wave hit 16 distinct bank pairs in passes 0 and 1 and at most 2-way conflicts elsewhere.
#define XLP_POS(i) ((i) + ((i) >> 2))
#define XLP_ROW (XLP_M + XLP_M / 4)  // padded row length in elements

template <int SIGN>
XL_DEV XlpTw xlp_twiddles(const v2f *__restrict__ W, const uint32_t j) {
  XlpTw t;
#pragma unroll
  for (int pass = 1; pass < 4; ++pass) {
    const uint32_t p = 1u << (2 * pass);
    const uint32_t k = j & (p - 1u);
    const uint32_t step = 64u >> (2 * pass);  // 256 / (4 p)
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      v2f w = W[(r * k * step) & 255u];
      if (SIGN > 0) w.y = -w.y;
      t.w[pass - 1][r - 1] = w;
    }
  }
  return t;
}

// one pass (butterflies) on registers
template <int SIGN>
XL_DEV void xlp_dft256_butterfly(v2f (&u)[4], const XlpTw &tw, const int pass) {
  if (pass > 0) {
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = xlp_cmul(u[r], tw.w[pass - 1][r - 1]);
  }
  const v2f v0 = u[0] + u[2], v1 = u[0] - u[2], v2 = u[1] + u[3], t = u[1] - u[3];
  const v2f v3 = SIGN > 0 ? (v2f){-t.y, t.x} : (v2f){t.y, -t.x};  // * (SIGN * j)
  u[0] = v0 + v2;
  u[1] = v1 + v3;
  u[2] = v0 - v2;
  u[3] = v1 - v3;
}

// NI independent transforms interleaved (instruction-level parallelism for a wave that runs almost alone)
template <int SIGN, int NI>
XL_DEV void xlp_dft256(v2f (&u)[NI][4], v2f *const (&lds)[NI], const XlpTw &tw, const uint32_t j) {
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t p = 1u << (2 * pass);
    const uint32_t k = j & (p - 1u);
#pragma unroll
    for (int n = 0; n < NI; ++n) xlp_dft256_butterfly<SIGN>(u[n], tw, pass);
    if (pass < 3) {
      const uint32_t jo = ((j - k) << 2) + k;
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[n][XLP_POS(jo + r * p)] = u[n][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) u[n][r] = lds[n][XLP_POS(j + 64u * r)];
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// NCO role of a launch: the first a.nco_blocks workgroups carry XL_NCO_LANES clients each (first wave only).
XL_DEV void xlp_nco_role(const XlpArgs &a, const XlDynArgs &dyn_next) {
  if (a.nco_prio == 3u) __builtin_amdgcn_s_setprio(3);
  else if (a.nco_prio == 2u) __builtin_amdgcn_s_setprio(2);
  else if (a.nco_prio == 1u) __builtin_amdgcn_s_setprio(1);
  if (threadIdx.x >= XL_NCO_LANES) return;
  const unsigned long long t0 = a.trace ? wall_clock64() : 0ull;
  const uint32_t c = blockIdx.x * XL_NCO_LANES + threadIdx.x;
  if (c >= a.nco_nclients) return;
  const XlNcoClient k = a.nco_clients[c];
  const uint32_t K = dyn_next.d[k.cls].K;
  const uint32_t kb = a.nco_k0 == 0u ? 0u : (uint32_t)(((uint64_t)K * a.nco_k0) >> 16) & ~(2u * XL_PH_STRIDE - 1u);
  const bool final = a.nco_k1 >= 65536u;
  const uint32_t ke = final ? K : (uint32_t)(((uint64_t)K * a.nco_k1) >> 16) & ~(2u * XL_PH_STRIDE - 1u);
  unsigned long long st[2] = {0ull, 0ull};
  xl_nco_client_slice(k, K, kb, ke, final, a.nco_state_src, a.nco_state_dst, a.nco_tab, a.trace ? st : nullptr);
  if (a.trace && threadIdx.x == 0) {
    unsigned long long *t = a.trace + 8 + 8 * blockIdx.x;
    t[0] = t0;
    t[1] = st[0];
    t[2] = st[1];
    t[3] = wall_clock64();
    t[4] = ke - kb;
    t[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
           __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  }
}

// tuning: time span of the work (non-NCO) waves of a launch
XL_DEV void xlp_trace_work(const XlpArgs &a, const unsigned long long t0) {
  if (a.trace && (threadIdx.x & 63u) == 0u) {  // per work workgroup: start, end, placement (own slot: no atomics)
    const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
    if (bid < 6000u) {
      unsigned long long *t = a.trace + 4096 + 4 * (size_t)bid;
      t[0] = t0;
      t[1] = wall_clock64();
      t[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
             __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
  }
}

// ------------------------------------------------------------------------------------------- forward transforms
// grid = nco_blocks + nseg * D transform workgroups (one wave each) + a.roll_blocks history-roll workgroups.
__global__ __launch_bounds__(64) void xlp_forward_kernel(const XlpArgs a, const XlDynArgs dyn,
                                                         const XlDynArgs dyn_next) {
  __shared__ v2f lds[XLP_ROW];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a, dyn_next);
    return;
  }
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t j = threadIdx.x;
  if (bid >= a.nseg * a.D) {
    // raw-history roll (as in xl_fir_kernel): hist_out = the last hist_units 2-byte units of [in0 | in1]; nothing in
    // this block's launches reads hist_out
    const uint32_t rb = bid - a.nseg * a.D;
    const uint16_t *__restrict__ h0 = reinterpret_cast<const uint16_t *>(a.in0);
    const uint16_t *__restrict__ h1 = reinterpret_cast<const uint16_t *>(a.in1);
    uint16_t *__restrict__ ho = reinterpret_cast<uint16_t *>(a.hist_out);
    for (uint32_t i = rb * 64u + j; i < a.hist_units; i += a.roll_blocks * 64u) {
      const uint32_t sidx = a.block_units + i;
      ho[i] = (sidx < a.hist_units) ? h0[sidx] : h1[sidx - a.hist_units];
    }
    return;
  }
  const XlpTw tw = xlp_twiddles<-1>(reinterpret_cast<const v2f *>(a.W), j);
  const uint32_t s = bid / a.D, b = bid - s * a.D;
  const XlDyn d = dyn.d[a.cls];
  // branch sample n of segment s = stream sample base + (s V + n) D + b   (base: first tap of output 0)
  const uint32_t first = d.base + s * a.V * a.D + b;
  const uint32_t end = a.n0 + a.n1;
  v2f u[1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t idx = first + (j + 64u * r) * a.D;
    const bool ok = idx >= d.zero_below && idx < end;  // late joiner: zeros below; past the block: zeros (those
                                                       // outputs lie beyond K and are never stored)
    const bool lo = idx < a.n0;
    const void *src = (lo || !ok) ? a.in0 : a.in1;
    const v2f v = xl_sample(src, (int)a.fmt, ok ? (lo ? idx : idx - a.n0) : 0u);
    u[0][r] = ok ? v : (v2f){0.0f, 0.0f};
  }
  v2f *const bufs[1] = {lds};
  xlp_dft256<-1, 1>(u, bufs, tw, j);
  const uint32_t pass = s / XLP_SEG, si = s - pass * XLP_SEG;
  v2f *__restrict__ X = reinterpret_cast<v2f *>(a.X);
#pragma unroll
  for (int r = 0; r < 4; ++r) X[(((size_t)pass * a.Dpad + b) * XLP_M + (j + 64u * r)) * XLP_XS + si] = u[0][r];
}

// ------------------------------------------------------------------------------------------- mix (the hot kernel)
// acc += r * x with ONE accumulator pair: two v_pk_fma_f32, the second negates x.im in its low half (neg_lo) --
//   acc.re += r.re*x.re;  acc.im += r.re*x.im;      acc.re += r.im*(-x.im);  acc.im += r.im*x.re
XL_DEV void xlp_cmac(v2f &acc, const v2f r, const v2f x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+v"(acc)
      : "v"(r), "v"(x));
}

// grid = nco_blocks + M * ncg * passes workgroups of ONE wave; lane l = client columns cg*128 + 2l, 2l+1; the spectrum
// bin m is workgroup-uniform.  The bin's column of the shared spectra (Dpad rows of 14 segments, 128 bytes each) is
// staged in LDS once and read back row by row as broadcast reads (every lane the same address): a uniform operand with
// short, in-order latency (scalar loads of the rows were latency-bound).  R is streamed from HBM exactly once, by
// exactly one wave: 16 bytes per lane and branch through a ring of XLP_BSTEP register slots, XLP_BSTEP - 1 rows ahead
// of the multiply (a stage-wise double buffer ran one short stage ahead and every stage waited out a memory latency;
// a two-wave workgroup sharing the R rows through L1 was 15 % slower at 4096 clients).
__global__ __launch_bounds__(64) void xlp_mix_kernel(const XlpArgs a, const XlDynArgs dyn_next) {
  extern __shared__ __attribute__((aligned(16))) v4f xlp_xcol[];  // [Dpad][8]
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a, dyn_next);
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
  const uint32_t m = bid % XLP_M;
  const uint32_t q = bid / XLP_M;
  const uint32_t cg = q % a.ncg, pass = q / a.ncg;
  // R image [cg][m][Dpad][128 columns]: the Dpad rows of a workgroup are one contiguous run (42 KB at D = 42)
  const v4f *__restrict__ Rp =
      reinterpret_cast<const v4f *>(a.R) + ((size_t)cg * XLP_M + m) * a.Dpad * (XLP_COLS / 2) + lane;
  const size_t rstride = XLP_COLS / 2;
  v4f r[XLP_BSTEP];
#pragma unroll
  for (int u = 0; u < (int)XLP_BSTEP - 1; ++u) r[u] = Rp[(size_t)u * rstride];
  {
    const v4f *__restrict__ Xc =
        reinterpret_cast<const v4f *>(a.X + ((size_t)pass * a.Dpad * XLP_M + m) * XLP_XS);  // row stride M * 8 v4f
    const uint32_t n8 = a.Dpad * 8u;
    for (uint32_t base = 0; base < n8; base += 512u) {  // one trip for D <= 64; all loads of a trip in flight together
      v4f t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t i = base + lane + 64u * u;
        const uint32_t ic = i < n8 ? i : 0u;
        t[u] = Xc[(size_t)(ic >> 3) * (XLP_M * 8u) + (ic & 7u)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t i = base + lane + 64u * u;
        if (i < n8) xlp_xcol[i] = t[u];
      }
    }
  }
  __syncthreads();
  v2f acc0[XLP_SEG], acc1[XLP_SEG];
#pragma unroll
  for (int i = 0; i < (int)XLP_SEG; ++i) acc0[i] = acc1[i] = (v2f){0.0f, 0.0f};
  for (uint32_t b0 = 0; b0 < a.Dpad; b0 += XLP_BSTEP) {
#pragma unroll
    for (int u = 0; u < (int)XLP_BSTEP; ++u) {
      // (the last trips request rows past this workgroup's: the next one's, or the tail padding the engine
      // allocates -- loaded, never used)
      constexpr int PF = (int)XLP_BSTEP - 1;
      r[(u + PF) % (int)XLP_BSTEP] = Rp[(size_t)(b0 + u + PF) * rstride];
      const v4f *__restrict__ xr = xlp_xcol + (b0 + u) * 8u;
      const v2f ra = {r[u].x, r[u].y}, rb = {r[u].z, r[u].w};
#pragma unroll
      for (int i = 0; i < (int)XLP_SEG / 2; ++i) {
        const v4f x2 = xr[i];
        const v2f xa = {x2.x, x2.y}, xb = {x2.z, x2.w};
        xlp_cmac(acc0[2 * i], ra, xa);
        xlp_cmac(acc1[2 * i], rb, xa);
        xlp_cmac(acc0[2 * i + 1], ra, xb);
        xlp_cmac(acc1[2 * i + 1], rb, xb);
      }
    }
  }
  const uint32_t s0 = pass * XLP_SEG;
  v4f *__restrict__ Yp =
      reinterpret_cast<v4f *>(a.Y) + (((size_t)cg * a.nseg_cap + s0) * XLP_M + m) * (XLP_COLS / 2) + lane;
#pragma unroll
  for (int i = 0; i < (int)XLP_SEG; ++i)
    if (s0 + i < a.nseg) Yp[(size_t)i * XLP_M * (XLP_COLS / 2)] = (v4f){acc0[i].x, acc0[i].y, acc1[i].x, acc1[i].y};
  xlp_trace_work(a, t_begin);
}

// ------------------------------------------------------------------------------------------- inverse + epilogue
// grid = nco_blocks + nseg * ncg * 8 workgroups of 256 threads; workgroup = (segment, 16 columns).  The tile rows
// double as the transforms' scratch; each wave runs its four columns' transforms interleaved.
__global__ __launch_bounds__(256) void xlp_inverse_kernel(const XlpArgs a, const XlDynArgs dyn,
                                                          const XlDynArgs dyn_next) {
  // [column][padded bin position].  Row length 319 (= XLP_POS(255) + 1): 638 dwords = -2 banks per row, so the 8 lanes
  // that fill 8 different rows of one bin hit distinct bank pairs; and 16 x 319 x 8 B = 40832 B lets a CU hold four
  // workgroups (at 41.2 KB it held three: 768 slots for the 832 workgroups of a 1024-client block -> a second round)
  __shared__ v2f tile[16][XLP_ROW - 1];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a, dyn_next);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;  // (as in xlp_mix_kernel; 4-wave workgroups: per CU)
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t sub = bid & 7u;
  const uint32_t q = bid >> 3;
  const uint32_t cg = q % a.ncg, s = q / a.ncg;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const XlpTw tw = xlp_twiddles<+1>(reinterpret_cast<const v2f *>(a.W), j);
  {
    // Y tile: 256 bins x 16 columns = 256 chunks of 128 contiguous bytes, 2 KB apart.  Eight lanes share a chunk
    // (16 bytes = 2 columns each), so a load instruction touches 8 full lines instead of 64 partial ones.
    const uint32_t part = threadIdx.x & 7u, mrow = threadIdx.x >> 3;  // mrow 0..31
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(
        a.Y + (((size_t)cg * a.nseg_cap + s) * XLP_M) * XLP_COLS + sub * 16u) + part;
    v4f v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(mrow + 32u * i) * (XLP_COLS / 2)];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t m = mrow + 32u * i;
      tile[2 * part][XLP_POS(m)] = (v2f){v[i].x, v[i].y};
      tile[2 * part + 1][XLP_POS(m)] = (v2f){v[i].z, v[i].w};
    }
  }
  // the epilogue's operands.  NCO phases: the table holds every XL_PH_STRIDE-th phase; after the transforms lane
  // (n, gq) = (j / GQ, j % GQ), GQ = 256 / XL_PH_STRIDE, expands the phases of outputs gq*XL_PH_STRIDE .. of the wave's
  // column n into that column's tile row (free by then), and every lane picks the phases of its own outputs j + 64 r
  // from there.  The one table entry a lane needs is requested here, before the transforms.
  constexpr uint32_t GQ = XLP_M / XL_PH_STRIDE;
  static_assert(4u * GQ <= 64u, "one expansion duty per lane");
  const uint32_t K = dyn.d[a.cls].K;
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  const uint32_t colbase = cg * XLP_COLS + sub * 16u + 4u * w;
  uint32_t off[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) off[n] = a.col_out[colbase + n];
  const uint32_t en = (j / GQ) & 3u, gq = j % GQ;  // expansion duty: column en, outputs gq*XL_PH_STRIDE ..
  const uint32_t eoff = a.col_out[colbase + en];
  const float2 eci = a.col_incr[colbase + en];
  const uint32_t k0 = s * a.V + gq * XL_PH_STRIDE;
  const bool eok = j < 4u * GQ && eoff != 0xFFFFFFFFu && gq * XL_PH_STRIDE < a.V && k0 < K;
  v2f pe = ph[eok ? (eoff >> XL_PH_SHIFT) + (k0 >> XL_PH_SHIFT) : 0u];
  __syncthreads();
  v2f u[4][4];
  v2f *const rows[4] = {tile[4 * w], tile[4 * w + 1], tile[4 * w + 2], tile[4 * w + 3]};
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) u[n][r] = rows[n][XLP_POS(j + 64u * r)];
  __builtin_amdgcn_wave_barrier();
  xlp_dft256<+1, 4>(u, rows, tw, j);
  __builtin_amdgcn_wave_barrier();
  if (eok) {
    const v2f einc = {eci.x, eci.y};
    for (uint32_t i = k0 & (XL_PH_STRIDE - 1u); i > 0u; --i) pe = xl_nco_next(pe, einc);  // (segments start anywhere)
    v2f *__restrict__ row = tile[4 * w + en];
    const uint32_t count = K - k0 < XL_PH_STRIDE ? K - k0 : XL_PH_STRIDE;
    for (uint32_t i = 0; i < count; ++i) {
      row[XLP_POS(gq * XL_PH_STRIDE + i)] = pe;
      pe = xl_nco_next(pe, einc);
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int n = 0; n < 4; ++n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t qo = j + 64u * r, k = s * a.V + qo;
      if (off[n] != 0xFFFFFFFFu && qo < a.V && k < K) {
        const v2f y = u[n][r] * (1.0f / (float)XLP_M);  // exact scaling by 2^-8
        out[off[n] + k] = xl_rotate<1>(y, rows[n][XLP_POS(qo)]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- branch spectra
// R[cg][m][b][col] = sum_{a<A} r_col[D a + b] e^{+2 pi j a m / M}, in double, rounded once.  One-time per plan.
__global__ __launch_bounds__(XLP_COLS) void xlp_tables_kernel(const float2 *__restrict__ rt, uint32_t ncols, uint32_t T,
                                                              uint32_t D, uint32_t Dpad, uint32_t A, float2 *__restrict__ R) {
  __shared__ double wc[XLP_M], ws[XLP_M];  // e^{+2 pi j n / M} in double
  for (uint32_t n = threadIdx.x; n < XLP_M; n += blockDim.x) sincospi(2.0 * (double)n / (double)XLP_M, &ws[n], &wc[n]);
  __syncthreads();
  const uint32_t m = blockIdx.x % XLP_M;
  const uint32_t q = blockIdx.x / XLP_M;
  const uint32_t b = q % Dpad, cg = q / Dpad;
  const uint32_t col = cg * XLP_COLS + threadIdx.x;
  double sr = 0.0, si = 0.0;
  if (col < ncols && b < D) {
    for (uint32_t aa = 0; aa < A; ++aa) {
      const uint32_t i = D * aa + b;
      if (i >= T) break;
      const uint32_t n = (aa * m) & (XLP_M - 1u);
      const double cs = wc[n], sn = ws[n];
      const float2 tv = rt[(size_t)i * ncols + col];  // [tap][column]: coalesced across the columns of the block
      const double tr = tv.x, ti = tv.y;
      sr += tr * cs - ti * sn;
      si += tr * sn + ti * cs;
    }
  }
  R[(((size_t)cg * XLP_M + m) * Dpad + b) * XLP_COLS + threadIdx.x] = make_float2((float)sr, (float)si);
}

// ------------------------------------------------------------------------------------------- launchers
hipError_t xlp_launch_tables(const float2 *rt, uint32_t ncols, uint32_t T, uint32_t D, uint32_t Dpad, uint32_t A,
                             uint32_t ncg, float2 *R, hipStream_t s) {
  hipLaunchKernelGGL(xlp_tables_kernel, dim3(XLP_M * Dpad * ncg), dim3(XLP_COLS), 0, s, rt, ncols, T, D, Dpad, A, R);
  return hipGetLastError();
}

hipError_t xlp_launch_forward(const XlpArgs &a, const XlDynArgs &dyn, const XlDynArgs &dyn_next, hipStream_t s) {
  hipLaunchKernelGGL(xlp_forward_kernel, dim3(a.nco_blocks + a.nseg * a.D + a.roll_blocks), dim3(64), 0, s, a, dyn,
                     dyn_next);
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

hipError_t xlp_launch_mix(const XlpArgs &a0, const XlDynArgs &dyn_next, hipStream_t s) {
  const uint32_t passes = (a0.nseg + XLP_SEG - 1) / XLP_SEG;
  const size_t lds = (size_t)a0.Dpad * 8u * sizeof(v4f);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const uint32_t work = XLP_M * a0.ncg * passes;
  const XlpArgs a = xlp_checked_skip(a0, work);
  hipLaunchKernelGGL(xlp_mix_kernel, dim3(a.nco_blocks + a.nco_skip + work), dim3(64), lds, s, a, dyn_next);
  return hipGetLastError();
}

hipError_t xlp_launch_inverse(const XlpArgs &a0, const XlDynArgs &dyn, const XlDynArgs &dyn_next, hipStream_t s) {
  const uint32_t work = a0.nseg * a0.ncg * 8u;
  const XlpArgs a = xlp_checked_skip(a0, work);
  hipLaunchKernelGGL(xlp_inverse_kernel, dim3(a.nco_blocks + a.nco_skip + work), dim3(256), 0, s, a, dyn, dyn_next);
  return hipGetLastError();
}
