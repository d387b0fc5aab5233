/* xl_grid.h -- the integer bookkeeping of the streaming rule, shared by host code, device code and the CPU tests
 * (tests/test_grid.py compiles this header with gcc and checks it against brute force).
 *
 * Reference: /root/reference/src/xlating.c:52-83.  A filter created at stream position `join` produces output k from
 * the T samples that END at sample k*D of ITS stream (the reference's working buffer starts with T-1 zeros,
 * xlating.c:552-559), in the process_* call during which that newest sample arrives, and renormalises its NCO phase
 * once at the end of every call that produced output (xlating.c:73).
 *
 * One engine call covers G consecutive blocks of S samples each (G = 1: the reference's call; G > 1: a "group", the
 * results of G successive calls from one set of launches).  With
 *     consumed = samples the client has seen before the call        j0 = (-consumed) mod D
 * output m of the call (m = 0 .. K-1, K = ceil((G*S - j0) / D)) has its newest sample at call-local position
 * j0 + m*D, belongs to block (j0 + m*D) / S, and its window starts at XL_HCAP - (T-1) + j0 + m*D in
 * [history | blocks] coordinates.  Plans are resident: a class stores (rem0, hv0) = (consumed mod D,
 * min(consumed, XL_HCAP)) at plan time and every launch gets the stream position since then (XlPos) as a kernel
 * argument -- nothing per class travels per call, whatever the number of classes.
 */
#ifndef XL_GRID_H_
#define XL_GRID_H_
#include <stdint.h>

#define XL_HCAP 16384u /* raw history kept on the device, in samples; T - 1 + D <= XL_HCAP */

#if defined(__HIPCC__)
#define XL_HD __host__ __device__ static inline
#else
#define XL_HD static inline
#endif

typedef struct XlPos {
  uint32_t trel; /* samples the engine consumed between the plan and the start of this call (< 2^31; the engine re-plans before it overflows) */
  uint32_t S;    /* samples per block of this call */
  uint32_t G;    /* blocks in this call (>= 1) */
  uint32_t pad;  /* flags: XL_POS_NORENORM = the call never renormalises the NCO phase (the reference's x86 AVX build, xlating.c:338-339) */
} XlPos;
#define XL_POS_NORENORM 1u
#define XL_POS_FMA_STEP 2u /* the phase recurrence step as gcc -ffast-math -mfma contracts `phase * phase_incr`: re = fma(pr, ir, -(pi*ii)),
                              im = fma(pr, ii, pi*ir) -- reference builds with FMA enabled (e.g. -march=native on an AVX2 host) */

typedef struct XlDyn {
  uint32_t base;       /* sample index in [history | blocks] coordinates of the first tap of output 0 */
  uint32_t K;          /* outputs of the call */
  uint32_t zero_below; /* samples with index < zero_below read as 0 (the client joined mid-stream) */
  uint32_t j0;         /* call-local position of output 0's newest sample */
} XlDyn;

/* per-call numbers of a class from its plan-time record */
XL_HD XlDyn xl_grid_dyn(uint32_t D, uint32_t T, uint32_t rem0, uint32_t hv0, XlPos p) {
  const uint32_t rem = (rem0 % D + p.trel % D) % D;
  const uint32_t j0 = (D - rem) % D;
  const uint32_t N = p.S * p.G;
  uint32_t hv = hv0 + (p.trel < XL_HCAP ? p.trel : XL_HCAP);
  XlDyn d;
  if (hv > XL_HCAP) hv = XL_HCAP;
  d.j0 = j0;
  d.K = N > j0 ? (N - j0 + D - 1u) / D : 0u;
  d.base = XL_HCAP - (T - 1u) + j0;
  d.zero_below = XL_HCAP - hv;
  return d;
}

/* stream position of the NEXT call if it has the same shape (the guess the NCO look-ahead makes) */
XL_HD XlPos xl_grid_next(XlPos p) {
  XlPos n = p;
  n.trel = p.trel + p.S * p.G;
  return n;
}

/* index of the first output of block g (g = 0 .. G): the outputs whose newest sample lies before g*S */
XL_HD uint32_t xl_grid_mstart(uint32_t j0, uint32_t D, uint32_t S, uint32_t g) {
  const uint32_t n = g * S;
  return n > j0 ? (n - j0 + D - 1u) / D : 0u;
}

/* Block boundaries of a call on one client's output index (NCO renormalisation points, xlating.c:73). */
typedef struct XlBnd {
  uint32_t j0, D, S, G, K;
  uint32_t flags; /* XL_POS_NORENORM: no renormalisation point exists */
} XlBnd;

/* smallest block-start index > m, or K when m lies in the last block: the phase is renormalised between
 * outputs xl_bnd_next(m) - 1 and xl_bnd_next(m).  Needs every block of a multi-block call to hold an output (S >= D). */
XL_HD uint32_t xl_bnd_next(const XlBnd b, uint32_t m) {
  uint32_t g, nb;
  if (b.flags & XL_POS_NORENORM) return 0xFFFFFFFFu; /* never reached: nobody renormalises */
  if (b.G <= 1u) return b.K;
  g = (b.j0 + m * b.D) / b.S;
  if (g + 1u >= b.G) return b.K;
  nb = xl_grid_mstart(b.j0, b.D, b.S, g + 1u);
  return nb < b.K ? nb : b.K;
}

/* ---- merged polyphase classes (xl_polyphase.hip): clients of one (D, T) whose output grids are offset against each
 * other share ONE grid of D-spaced window starts, the "shared grid", whose point q = 0 lies D samples before the
 * window of output 0 of a virtual reference client with j0 = j0_ref.  A member with
 *     delta = (j0_c - j0_ref) mod D        (constant over the calls: both move by -G*S mod D per call)
 * is evaluated with its taps delayed by delta samples (delta leading zeros baked into its branch spectra), and its
 * output k is the shared point q = k + 1 - wrap, wrap = (j0_ref + delta >= D). */
XL_HD uint32_t xl_merge_j0(uint32_t j0_ref, uint32_t delta, uint32_t D) {
  const uint32_t j = j0_ref + delta;
  return j >= D ? j - D : j;
}
/* q - k of a member */
XL_HD uint32_t xl_merge_shift(uint32_t j0_ref, uint32_t delta, uint32_t D) { return j0_ref + delta >= D ? 0u : 1u; }
/* shared points a call must evaluate so that every member gets all its outputs (q < Kq) */
XL_HD uint32_t xl_merge_points(uint32_t D, XlPos p) { return (p.S * p.G + D - 1u) / D + 1u; }

#endif /* XL_GRID_H_ */
