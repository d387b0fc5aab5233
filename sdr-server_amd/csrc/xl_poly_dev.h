// xl_poly_dev.h -- device helpers shared by the polyphase kernels (xl_polyphase.hip: forward / mix / inverse launches;
// xl_fused.hip: the fused mix + inverse launch): complex products, the M-point Stockham transform staged in LDS, the
// packed-instruction arithmetic policy of the register transforms (xl_fft16.h), the two-half split of float32 values for
// the matrix cores, and the branch spectra of a client column.  Include only from .hip files compiled -ffp-contract=off.
#ifndef XL_POLY_DEV_H_
#define XL_POLY_DEV_H_
#include "xl_polyphase.h"

#include "xl_dev_inline.h"
#include "xl_fft16.h"

XL_DEV v2f xlp_cmul(const v2f a, const v2f b) {
  return (v2f){__builtin_fmaf(-a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y)};
}

// M-point DFT, M = 256, 128 or 64, by M/4 lanes (a wave, a half-wave, a quarter-wave): radix-4
// Stockham autosort, passes p = 1, 4, 16 (and 64 for M = 256); lane l < M/4 holds points l + (M/4) r.  M = 128 ends with
// a radix-2 pass that needs no exchange: after the third scatter a lane's four points are the operands of its two
// radix-2 butterflies, (l, l + 64) and (l + 32, l + 96).
// In: u[r] = x[l + (M/4) r].  Out: u[r] = X[l + (M/4) r] (natural order).  SIGN -1 forward, +1 inverse (unnormalised).
// The twiddles of a lane depend only on (pass, r, l): xlp_twiddles() fetches them once (one exposed global latency
// instead of three), e^{-2 pi j n / 256} from the table W, conjugated for the inverse.
// `lds` = XLP_ROW(M) complex owned by this transform, addressed through XLP_POS (may be the input row itself); LDS
// operations of one wave execute in order, so no barrier is needed between a pass's scatter and the next gather.
struct XlpTw {
  v2f w[3][3];  // [pass - 1][r - 1]; M = 128: w[2][0], w[2][1] are the radix-2 twiddles
};
// LDS position of transform element i: one pad element per four.  The scatter of pass p writes elements
// jo + r p with jo = 4 (l - l % p) + l % p -- strides of 4, 16, 64 elements of 8 bytes across lanes, a 4- to 16-way
// bank conflict on a dense row (measured: the inverse kernel spent ~20 of its 28 us there); with the pad the 16
// lanes of a quarter-wave hit 16 distinct bank pairs in passes 0 and 1 and at most 2-way conflicts elsewhere.
#define XLP_POS(i) ((i) + ((i) >> 2))
#define XLP_ROW(M) ((M) + (M) / 4)  // padded row length in elements
struct XlpPosPad {  // the padded layout above; `rs` (per-row constant) unused
  static XL_MEM uint32_t pos(const uint32_t i, const uint32_t) { return XLP_POS(i); }
};
// Dense rows of 128 elements with an XOR swizzle instead of the pad (M = 128 inverse kernel, option "inverse_kernel" = 3):
//   pos(i) = i ^ g(a) ^ rs,   a = (i >> 4) & 7,   g(a) = 5 a mod 16 = {0, 5, 10, 15, 4, 9, 14, 3},   rs = a per-row constant < 16
// * gathers (32 lanes read elements l + 32 r of one row; 64 banks of 4 bytes = 32 elements): a depends on bit 4 of l and on r
//   only, and g < 16 leaves bit 4 alone -> a bijection of the 32 elements: conflict-free;
// * scatters (16-lane groups; 32 banks = 16 elements): pass 1 writes 4 l + r, a = l >> 2: within one a the XOR permutes the
//   four values 4 (l & 3) + r, and two a never meet because g(a) ^ g(a') is never a multiple of 4 inside a group of four
//   (differences 5, 10, 15 / 13, 10, 7); pass 4 writes 16 a + k + 4 r: the same argument with differences never in
//   {1, 2, 3}; pass 16 writes l + 16 r: one constant XOR per group;
// * the phase expansion (a 16-lane group = 8 lanes x 2 rows writing element 16 gq + c): the eight g(gq) are distinct, and
//   the two rows' constants differ by 8, which is no difference of two g values;
// * the tile fill (16 lanes write the same bin of rows 2 part, then of rows 2 part + 1): the sixteen row constants of
//   either set are distinct.    rs(row) = ((row >> 1) & 15) ^ ((row & 1) << 3)   (XLP_SWZ_ROW)
// A workgroup's tile is then exactly 32 KB.
#define XLP_SWZ_ROW(row) ((((row) >> 1) & 15u) ^ (((row) & 1u) << 3))
struct XlpPosSwz {
  static XL_MEM uint32_t pos(const uint32_t i, const uint32_t rs) { return i ^ ((5u * ((i >> 4) & 7u)) & 15u) ^ rs; }
};

template <int SIGN, int M>
XL_DEV XlpTw xlp_twiddles(const v2f *__restrict__ W, const uint32_t l) {
  XlpTw t;
  constexpr int NP4 = M == 256 ? 4 : 3;  // radix-4 passes
#pragma unroll
  for (int pass = 1; pass < NP4; ++pass) {
    const uint32_t p = 1u << (2 * pass);
    const uint32_t k = l & (p - 1u);
    const uint32_t step = 64u >> (2 * pass);  // W_{4p}^{r k} = W_256^{r k 64 / p}, whatever M
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      v2f w = W[(r * k * step) & 255u];
      if (SIGN > 0) w.y = -w.y;
      t.w[pass - 1][r - 1] = w;
    }
  }
  if (M == 128) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v2f w = W[(2u * l + 64u * i) & 255u];  // W_128^{l + 32 i}
      if (SIGN > 0) w.y = -w.y;
      t.w[2][i] = w;
    }
  }
  return t;
}

// one radix-4 pass (butterflies) on registers
template <int SIGN>
XL_DEV void xlp_dft_butterfly(v2f (&u)[4], const XlpTw &tw, const int pass) {
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

// NI independent transforms per lane interleaved (instruction-level parallelism for a wave that runs almost alone)
template <int SIGN, int NI, int M, class P = XlpPosPad>
XL_DEV void xlp_dft(v2f (&u)[NI][4], v2f *const (&lds)[NI], const XlpTw &tw, const uint32_t l, const uint32_t (&rs)[NI]) {
  constexpr uint32_t L = M / 4;
  constexpr int NP4 = M == 256 ? 4 : 3;
#pragma unroll
  for (int pass = 0; pass < NP4; ++pass) {
    const uint32_t p = 1u << (2 * pass);
    const uint32_t k = l & (p - 1u);
#pragma unroll
    for (int n = 0; n < NI; ++n) xlp_dft_butterfly<SIGN>(u[n], tw, pass);
    if (pass < NP4 - 1 || M == 128) {  // (no exchange behind the last radix-4 pass, unless a radix-2 pass follows: M = 128)
      const uint32_t jo = ((l - k) << 2) + k;
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[n][P::pos(jo + r * p, rs[n])] = u[n][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) u[n][r] = lds[n][P::pos(l + L * r, rs[n])];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (M == 128) {
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const v2f a0 = u[n][0], a1 = xlp_cmul(u[n][2], tw.w[2][0]);
      const v2f b0 = u[n][1], b1 = xlp_cmul(u[n][3], tw.w[2][1]);
      u[n][0] = a0 + a1;
      u[n][1] = b0 + b1;
      u[n][2] = a0 - a1;
      u[n][3] = b0 - b1;
    }
  }
}

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v16f32 __attribute__((ext_vector_type(16)));

// v * scale as two halves; (lo, hi) of the returned pairs: first terms, second terms
// Workgroup barrier for LDS hand-offs only: this wave's LDS operations are done (lgkmcnt), everybody arrives.  __syncthreads() also
// fences global memory, which on this target is `s_waitcnt vmcnt(0)`: a wait for every load and store the wave has in flight.
#ifdef XLP_EXP_SYNCTHREADS  // (A/B builds only, tools/experiments/build_variant.sh: the barrier of rounds 3-5)
XL_DEV void xlp_lds_barrier() { __syncthreads(); }
#else
XL_DEV void xlp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// Two-half mix of a cf32 stream: the power-of-two scale of a segment's rows from the float bits of its largest spectrum component
// (XlpArgs::segmax): 2^(14 - e), e = floor(log2 max) -- every scaled component < 2^15 --, and what undoes it.
XL_DEV uint32_t xlp_seg_exp(const uint32_t maxbits) {  // biased exponent of the segment's largest component, kept where both powers are normal
  const uint32_t ex = maxbits >> 23;
  return ex < 27u ? 27u : (ex > 254u ? 254u : ex);
}
XL_DEV float xlp_seg_scale(const uint32_t maxbits) { return __uint_as_float((268u - xlp_seg_exp(maxbits)) << 23); }  // 2^(14 - e)
XL_DEV float xlp_seg_unscale(const uint32_t maxbits) { return __uint_as_float((xlp_seg_exp(maxbits) - 14u) << 23); }  // 2^(e - 14)


XL_DEV void xlp_split_h(const float v, _Float16 &h1, _Float16 &h2) {
  h1 = (_Float16)v;
  h2 = (_Float16)(v - (float)h1);
}
XL_DEV uint32_t xlp_pack_h(const _Float16 lo, const _Float16 hi) {
  const v2h p = {lo, hi};
  return __builtin_bit_cast(uint32_t, p);
}

// v * (cs.x + j cs.y): two packed instructions, the swap and the sign in the operand modifiers
//   t = (v.x, v.y) * (c, c);   r = (-v.y, v.x) * (s, s) + t
XL_DEV v2f xlp_cmul_s(const v2f v, const v2f cs) {  // twiddle in a scalar register pair
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : "=&v"(r)
      : "v"(v), "s"(cs));
  return r;
}
XL_DEV v2f xlp_cmul_v(const v2f v, const v2f cs) {  // factor in a vector register pair (the NCO phase)
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : "=&v"(r)
      : "v"(v), "v"(cs));
  return r;
}
struct XlpFftOps {
  template <int N>
  static XL_MEM v2f twiddle(const v2f v) {
    return xlp_cmul_s(v, (v2f){xl_w128_cos(N), xl_w128_sin(N)});
  }
  static XL_MEM v2f add_j(const v2f a, const v2f d) {  // (a.x - d.y, a.y + d.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
  }
  static XL_MEM v2f sub_j(const v2f a, const v2f d) {  // (a.x + d.y, a.y - d.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
  }
};

// (sr, si) = R_b[m] of list entry j; `wc`, `ws`: e^{+2 pi j n / M} in double
XL_DEV void xlp_branch_spectrum(const float2 *__restrict__ rt, const uint32_t nlist, const uint32_t j, const uint32_t dl,
                                const uint32_t T, const uint32_t D, const uint32_t A, const uint32_t M, const uint32_t m,
                                const uint32_t b, const double *wc, const double *ws, double &sr, double &si) {
  sr = 0.0, si = 0.0;
  if (b >= D) return;
  for (uint32_t aa = 0; aa < A; ++aa) {  // the column's taps are delayed by dl samples: r'[i] = r[i - dl]
    if (D * aa + b < dl) continue;
    const uint32_t i = D * aa + b - dl;
    if (i >= T) break;
    const uint32_t n = (aa * m) & (M - 1u);
    const double cs = wc[n], sn = ws[n];
    const float2 tv = rt[(size_t)i * nlist + j];  // [tap][list entry]: coalesced across the entries of the block
    const double tr = tv.x, ti = tv.y;
    sr += tr * cs - ti * sn;
    si += tr * sn + ti * cs;
  }
}

// Where workgroup `bid` of a mix launch works: bin m, column group cg, pass run `run` (of `runs`).  Workgroups go to the 8 XCDs round
// robin (bid % 8), so
//   * the pass runs of one (bin, column group) sit 8 positions apart in the grid: same XCD, dispatched together -- the group's operands
//     come from HBM once;
//   * XCD x takes the bins x M/8 .. (x + 1) M/8 - 1 of every column group, in order (round 5): the 16 (32) workgroups an XCD runs side
//     by side for a column group write ADJACENT 256-byte runs of the Y tiles -- 4 KB contiguous from one L2 at about the same time
//     instead of runs 2 KB apart from eight --, and an XCD's L2 holds an eighth of the shared spectra X instead of all of them.  The
//     launch's store pattern alone (tools/ubench_tile_copy.hip, 4096 clients): 148 against 169 us per launch; the launches:
//     profiles/r05_mix_xcd_bins.txt.
XL_DEV void xlp_mix_place(const uint32_t bid, const uint32_t M, const uint32_t runs, uint32_t &m, uint32_t &cg, uint32_t &run) {
  const uint32_t grp = bid / (8u * runs), rr = bid - grp * 8u * runs;
  run = rr >> 3;
#ifdef XLP_MIX_BINS_STRIDED  // (experiment: round 3's placement -- XCD x takes the bins = x mod 8)
  const uint32_t pair = grp * 8u + (rr & 7u);
  m = pair & (M - 1u), cg = pair / M;
#else
  const uint32_t per = M >> 3;
  m = (rr & 7u) * per + grp % per, cg = grp / per;
#endif
}

// NCO role of a launch: the first a.nco_blocks workgroups carry XL_NCO_LANES clients each (first wave only) through
// this launch's slice of the NEXT call's phase recurrence.
XL_DEV void xlp_nco_role(const XlpArgs &a) {
  if (a.nco_prio == 3u) __builtin_amdgcn_s_setprio(3);
  else if (a.nco_prio == 2u) __builtin_amdgcn_s_setprio(2);
  else if (a.nco_prio == 1u) __builtin_amdgcn_s_setprio(1);
  if (threadIdx.x >= XL_NCO_LANES) return;
  const unsigned long long t0 = a.trace ? wall_clock64() : 0ull;
  const uint32_t c = blockIdx.x * XL_NCO_LANES + threadIdx.x;
  if (c >= a.nco_nclients) return;
  const XlNcoClient k = a.nco_clients[c];
  const XlBnd bnd = xl_nco_bnd(k, xl_grid_next(a.pos), 0xFFFFFFFFu);
  const uint32_t K = bnd.K;
  const uint32_t kb = a.nco_k0 == 0u ? 0u : (uint32_t)(((uint64_t)K * a.nco_k0) >> 16) & ~(2u * XL_PH_STRIDE - 1u);
  const uint32_t ke = a.nco_k1 >= 65536u ? K : (uint32_t)(((uint64_t)K * a.nco_k1) >> 16) & ~(2u * XL_PH_STRIDE - 1u);
  unsigned long long st[2] = {0ull, 0ull};
  xl_nco_client_chain<false>(k, bnd, kb, ke, a.nco_state_src, a.nco_state_dst, a.nco_tab, a.trace ? st : nullptr);
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
#ifdef XL_TUNING  // (the engine only ever sets a.trace in a tuning build; outside one the bookkeeping is not carried along)
  if (a.trace && (threadIdx.x & 63u) == 0u) {  // per work wave: start, end, placement (own slot: no atomics)
    const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
    const uint32_t slot = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot < 6000u) {
      unsigned long long *t = a.trace + 4096 + 4 * (size_t)slot;
      t[0] = t0;
      t[1] = wall_clock64();
      t[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
             __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
  }
#else
  (void)a;
  (void)t0;
#endif
}

#endif  // XL_POLY_DEV_H_
