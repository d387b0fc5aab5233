// xl_dev_inline.h -- device-side helpers shared by the kernel translation units (xl_kernels.hip, xl_polyphase.hip):
// vector typedefs, the exact sample converters, the complex accumulators and the NCO rotate.
// Reference semantics: /root/reference/src/xlating.c (cited per helper).  Include only from .hip files compiled with
// -ffp-contract=off (nothing is fused unless written as a fused operation).
#ifndef XL_DEV_INLINE_H_
#define XL_DEV_INLINE_H_
#include "xl_device.h"

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// constant address space: uniform loads through these pointers are selected as scalar (SMEM) loads
typedef const float __attribute__((address_space(4))) *cfloat_p;
typedef const uint32_t __attribute__((address_space(4))) *cu32_p;

#define XL_DEV static __device__ __forceinline__
#define XL_MEM __device__ __forceinline__

// ------------------------------------------------------------------------------------------- sample converters
// xlating.c:357-358 / 367-368 / 377-378: all three maps are exact in float32.
XL_DEV v2f xl_sample(const void *__restrict__ p, int fmt, uint32_t i) {
  v2f r;
  if (fmt == XLF_CU8) {
    const uint32_t v = reinterpret_cast<const uint16_t *>(p)[i];
    r.x = ((float)(v & 0xFFu) - 127.5f) / 128.0f;
    r.y = ((float)(v >> 8) - 127.5f) / 128.0f;
  } else if (fmt == XLF_CS8) {
    const int32_t v = reinterpret_cast<const int16_t *>(p)[i];
    r.x = (float)((int32_t)(int8_t)(v & 0xFF)) / 128.0f;
    r.y = (float)(v >> 8) / 128.0f;
  } else if (fmt == XLF_CS16) {
    const int32_t v = reinterpret_cast<const int32_t *>(p)[i];
    r.x = (float)((int32_t)(int16_t)(v & 0xFFFF)) / 32768.0f;
    r.y = (float)(v >> 16) / 32768.0f;
  } else {
    r = reinterpret_cast<const v2f *>(p)[i];
  }
  return r;
}

// ------------------------------------------------------------------------------------------- complex arithmetic
// One complex accumulator per (output, client).
// MODE 0 (native): the reference's scalar expression tree, xlating.c:68 `temp += x * h` in C99 complex float:
//   p = (xr*hr - xi*hi) + j(xr*hi + xi*hr);  acc += p        -- 4 mul, 2 add/sub, 2 add, each rounded.
// MODE 1 (optimized): the sum is kept as two packed halves that need no negation inside the loop,
//   a += xr * (hr, hi)      b += xi * (hi, hr)      acc = (a.x - b.x, a.y + b.y)
//   = exactly two v_pk_fma_f32 per complex MAC (op_sel broadcasts xr / xi and swaps the tap), same tap order.
template <int MODE>
struct XlAcc;

template <>
struct XlAcc<0> {
  v2f s;
  XL_MEM void clear() { s = (v2f){0.0f, 0.0f}; }
  // Same roundings as the scalar tree  pr = xr*hr - xi*hi;  pi = xr*hi + xi*hr;  s += (pr, pi)  (every product and
  // sum rounded once, nothing fused: this TU is compiled -ffp-contract=off), arranged so that each step is one packed
  // instruction without register shuffles:  p1 = xr*(hr,hi)   p2 = xi*(hi,hr)   p = (p1.x - p2.x, p1.y + p2.y)   s += p
  XL_MEM void mac(const v2f x, const float hr, const float hi) {
    // hand-placed: the compiler builds the (-p2.x, p2.y) operand with an extra v_pk_add and a v_mov (6 VALU per
    // MAC); op_sel / neg_lo do it for free (4 VALU per MAC).  IEEE mul/add, one rounding each, nothing fused.
    const v2f h = {hr, hi};
    v2f p1, p2;
    asm("v_pk_mul_f32 %0, %3, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %3, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %2, %2, %0"
        : "=&v"(p1), "=&v"(p2), "+v"(s)
        : "v"(x), "s"(h));
  }
  XL_MEM v2f value() const { return s; }
};

template <>
struct XlAcc<1> {
  v2f a, b;
  XL_MEM void clear() {
    a = (v2f){0.0f, 0.0f};
    b = (v2f){0.0f, 0.0f};
  }
  XL_MEM void mac(const v2f x, const float hr, const float hi) {
    a = __builtin_elementwise_fma((v2f){x.x, x.x}, (v2f){hr, hi}, a);
    b = __builtin_elementwise_fma((v2f){x.y, x.y}, (v2f){hi, hr}, b);
  }
  XL_MEM v2f value() const { return (v2f){a.x - b.x, a.y + b.y}; }
};

// xlating.c:70 `out = temp * phase`
template <int MODE>
XL_DEV v2f xl_rotate(const v2f a, const v2f p) {
  v2f r;
  if (MODE == 0) {
    r.x = a.x * p.x - a.y * p.y;
    r.y = a.x * p.y + a.y * p.x;
  } else {
    r.x = __builtin_fmaf(-a.y, p.y, a.x * p.x);
    r.y = __builtin_fmaf(a.y, p.x, a.x * p.y);
  }
  return r;
}

// ------------------------------------------------------------------------------------------- NCO recurrence
// xlating.c:70-73: the phasor is a float32 RECURRENCE p <- p * incr (never re-seeded), renormalised once per call
// that could produce output; hypotf: glibc evaluates sqrt(x*x + y*y) in double and narrows (restated with IEEE double ops).
// One recurrence step p <- p * incr as the reference's C99 complex float product (xlating.c:71):
//   re = pr*ir - pi*ii, im = pr*ii + pi*ir, every operation rounded once (IEEE mul / add, nothing fused).
// A lone wave issues one VALU instruction every ~5-8 cycles whatever its width, so the step is written as THREE
// packed instructions (left to the compiler it became 6-8 with register shuffles, ~50 cycles per step):
//   t1 = (pr, pi) * (ir, ir)        t2 = (pr, pi) * (ii, ii)        p = (t1.x - t2.y, t1.y + t2.x)
XL_DEV void xl_nco_step(v2f &p, const v2f inc) {
  v2f t1, t2;
  asm volatile(
      "v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]"
      : "=&v"(t1), "=&v"(t2)
      : "v"(p), "v"(inc));
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(p) : "v"(t1), "v"(t2));
}

// the same step into a fresh register pair (xl_nco_client_slice keeps 16 phases live for the table stores)
XL_DEV v2f xl_nco_next(const v2f p, const v2f inc) {
  v2f t1, t2, r;
  asm volatile(
      "v_pk_mul_f32 %0, %3, %4 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %1, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_add_f32 %2, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
      : "=&v"(t1), "=&v"(t2), "=&v"(r)
      : "v"(p), "v"(inc));
  return r;
}

// The same step as a reference build with FMA contraction computes it (gcc -O3 -ffast-math -mfma turns xlating.c:338
// `phase * phase_incr` into  re = fma(pr, ir, -(pi * ii)),  im = fma(pr, ii, pi * ir): found by matching the unmodified
// reference's phase sequence bit for bit, oracle/_ref/libref_fast.so; tests/test_oracle.py): one packed multiply, one
// packed FMA.
XL_DEV v2f xl_nco_next_fma(const v2f p, const v2f inc) {
  v2f t, r;
  asm volatile(
      "v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
      : "=&v"(t), "=&v"(r)
      : "v"(p), "v"(inc));
  return r;
}
// the step a call's flags ask for (xl_grid.h: XL_POS_FMA_STEP); wave-uniform choice
XL_DEV v2f xl_nco_next_any(const v2f p, const v2f inc, const uint32_t flags) {
  return (flags & XL_POS_FMA_STEP) ? xl_nco_next_fma(p, inc) : xl_nco_next(p, inc);
}

// The step for the NCO ROLE (xl_nco_client_chain: a wave that runs the recurrence for thousands of steps INSIDE a launch, next to
// whatever else is resident on its SIMD): the same IEEE operations as SIX SCALAR instructions.  Round 4 pinned the corruption
// round 3 could not explain: a role wave stepping with the packed instructions above got the phases of its lanes
// 48..63 wrong -- about once per 10^7 steps -- whenever waves issuing MATRIX instructions were resident on the chip at the same
// time, its own launch's or ANOTHER engine's (two engines in one process: each alone bit-exact over 1500 calls, together the
// packed role of one corrupt within 100; tools/experiments/dbg_soak.py).  Work waves next to matrix-core launches compute the same
// bits (dbg_soak2.py), the short phase walks of the consumers too, the side-stream chain kernel owns its SIMDs outright; with
// the scalar step the role is immune (1500 calls x 4096 clients beside a matrix-core engine: bit-exact) at ~1.4 x the cycles per
// step -- which only launches long enough to hide it carry anyway.
XL_DEV v2f xl_nco_role_step(const v2f p, const v2f inc, const uint32_t flags) {
  // (inline assembly on purpose: written in C the vectoriser turns the four products and two sums back into v_pk_mul_f32 /
  // v_pk_add_f32; plain VALU dependencies are interlocked by the hardware, no wait states are needed in here)
  float a, b, c, d;
  if (flags & XL_POS_FMA_STEP) {  // re = fma(pr, ir, -(pi ii)), im = fma(pr, ii, pi ir)
    asm volatile(
        "v_mul_f32 %0, %3, %5\n\t"
        "v_mul_f32 %1, %3, %4\n\t"
        "v_fma_f32 %0, %2, %4, -%0\n\t"
        "v_fma_f32 %1, %2, %5, %1"
        : "=&v"(a), "=&v"(b)
        : "v"(p.x), "v"(p.y), "v"(inc.x), "v"(inc.y));
    return (v2f){a, b};
  }
  asm volatile(
      "v_mul_f32 %0, %4, %6\n\t"
      "v_mul_f32 %1, %5, %7\n\t"
      "v_mul_f32 %2, %4, %7\n\t"
      "v_mul_f32 %3, %5, %6\n\t"
      "v_sub_f32 %0, %0, %1\n\t"
      "v_add_f32 %1, %3, %2"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p.x), "v"(p.y), "v"(inc.x), "v"(inc.y));
  return (v2f){a, b};
}

// xlating.c:73 `phase /= hypotf(re, im)`: glibc's hypotf evaluates sqrt(x*x + y*y) in double and narrows; restated with
// IEEE double operations (equal to libm on 2e8 inputs, tests/test_oracle.py) and correctly rounded float divisions.
XL_DEV v2f xl_nco_renorm(const v2f p) {
  const double mag2 = (double)p.x * (double)p.x + (double)p.y * (double)p.y;
  const float mag = (float)__dsqrt_rn(mag2);
  return (v2f){p.x / mag, p.y / mag};
}

// PACKED: the three packed instructions per step -- ONLY for a wave that owns its SIMD (xl_nco_table_kernel claims all 512 registers,
// like the side-stream chain kernel); false: the six scalar instructions of xl_nco_role_step (every role inside a launch).
// The work of one lane = one client: advance the recurrence over the outputs [kb, ke) of a call of bnd.K outputs in
// bnd.G blocks, tabulating every XL_PH_STRIDE-th phase (entry (out_off + m) / XL_PH_STRIDE = phase of output m, m on
// the call's output index) and renormalising at every block end inside the range (xlating.c:73; the table entry of a
// block's first output is the renormalised phase).  The running phase comes from state_src[slot] and goes to
// state_dst[slot]: a slice that ends the call (ke == bnd.K) leaves the committed post-call phase there.
template <bool PACKED>
XL_DEV void xl_nco_client_chain(const XlNcoClient k, const XlBnd bnd, const uint32_t kb, const uint32_t ke,
                                const float2 *state_src, float2 *state_dst, float2 *__restrict__ tab,
                                unsigned long long *stamp = nullptr) {
  v2f p = {state_src[k.slot].x, state_src[k.slot].y};
  if (stamp) {  // tuning: when did the running phase arrive, when did the recurrence end
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[0] = wall_clock64();
  }
  if (bnd.K == 0u) {  // no output possible in this call: the reference leaves the phase untouched (xlating.c:58)
    state_dst[k.slot] = make_float2(p.x, p.y);
    return;
  }
  const v2f inc = {k.incr.x, k.incr.y};
  v2f *__restrict__ o = reinterpret_cast<v2f *>(tab) + (k.out_off >> XL_PH_SHIFT);  // out_off = 0 mod 2 * XL_PH_STRIDE: 16-byte pairs
  v4f *__restrict__ o4 = reinterpret_cast<v4f *>(o);
  uint32_t m = kb;
  while (m < ke) {
    const uint32_t nb = xl_bnd_next(bnd, m);  // end of the block output m lies in
    const uint32_t me = nb < ke ? nb : ke;
    for (; m < me && (m & (2u * XL_PH_STRIDE - 1u)) != 0u; ++m) {  // head: up to the next pair boundary
      if (tab != nullptr && (m & (XL_PH_STRIDE - 1u)) == 0u) o[m >> XL_PH_SHIFT] = p;
      p = PACKED ? xl_nco_next_any(p, inc, bnd.flags) : xl_nco_role_step(p, inc, bnd.flags);
    }
    // 2 * XL_PH_STRIDE steps and ONE store (two entries) per trip (plain step; FMA-step calls take the loops around it)
    for (; !(bnd.flags & XL_POS_FMA_STEP) && m + 2u * XL_PH_STRIDE <= me; m += 2u * XL_PH_STRIDE) {
      const v2f q0 = p;
      for (uint32_t i = 0; i < XL_PH_STRIDE; i += 16u) {
#pragma unroll
        for (int j = 0; j < 16; ++j) p = PACKED ? xl_nco_next(p, inc) : xl_nco_role_step(p, inc, 0u);
      }
      const v2f q1 = p;
      for (uint32_t i = 0; i < XL_PH_STRIDE; i += 16u) {
#pragma unroll
        for (int j = 0; j < 16; ++j) p = PACKED ? xl_nco_next(p, inc) : xl_nco_role_step(p, inc, 0u);
      }
      if (tab != nullptr) o4[m >> (XL_PH_SHIFT + 1u)] = (v4f){q0.x, q0.y, q1.x, q1.y};
    }
    for (; m < me; ++m) {
      if (tab != nullptr && (m & (XL_PH_STRIDE - 1u)) == 0u) o[m >> XL_PH_SHIFT] = p;
      p = PACKED ? xl_nco_next_any(p, inc, bnd.flags) : xl_nco_role_step(p, inc, bnd.flags);
    }
    if (me == nb) p = xl_nco_renorm(p);  // a block of the call ends here
  }
  if (stamp) stamp[1] = wall_clock64();
  state_dst[k.slot] = make_float2(p.x, p.y);
}

// Block boundaries of client k in a call at stream position pos (xl_grid.h); explicit_K != 0xFFFFFFFF: the
// single-filter path's one-block call of explicit_K outputs.
XL_DEV XlBnd xl_nco_bnd(const XlNcoClient k, const XlPos pos, const uint32_t explicit_K) {
  XlBnd b;
  if (explicit_K != 0xFFFFFFFFu) {
    b.j0 = 0u, b.D = k.D, b.S = 0xFFFFFFFFu, b.G = 1u, b.K = explicit_K;
  } else {
    const XlDyn d = xl_grid_dyn(k.D, 1u, k.rem0, 0u, pos);
    b.j0 = d.j0, b.D = k.D, b.S = pos.S, b.G = pos.G, b.K = d.K;
  }
  b.flags = pos.pad;
  return b;
}

// Consumer side, one lane: the phases of outputs m0 .. m0 + count - 1 of a client, handed to store(i, phase), i < count.
// `p` is the tabulated phase of output mt = m0 rounded down to the table stride (the caller loaded it, early); the
// phases in between follow with the producer's own three IEEE operations, renormalised where a block of the call
// ends (bit-identical to the producer's chain).
template <class Store>
XL_DEV void xl_phase_walk(v2f p, const uint32_t m0, const uint32_t count, const v2f inc, const XlBnd bnd, Store store) {
  uint32_t m = m0 & ~(XL_PH_STRIDE - 1u);
  uint32_t nb = xl_bnd_next(bnd, m);
  const uint32_t end = m0 + count;
  for (; m < end; ++m) {
    if (m >= m0) store(m - m0, p);
    p = xl_nco_next_any(p, inc, bnd.flags);
    if (m + 1u == nb) {
      p = xl_nco_renorm(p);
      nb = xl_bnd_next(bnd, m + 1u);
    }
  }
}

#endif  // XL_DEV_INLINE_H_
