// xl_fft64.h -- a 128-point inverse DFT held in the registers of a PAIR of lanes (64 points each), for the inverse
// launch of the polyphase path (xl_polyphase.hip: xlp_inverse_reg_kernel).  Written so that the same text compiles for
// the device (V = float ext_vector_type(2), partner values through DPP) and for the host (tests/c/test_fft64.cpp: a
// two-"lane" emulation checked against a double-precision DFT), because the index bookkeeping of a register FFT is the
// kind of thing that is either exactly right or silently wrong.
//
//   x[n] = sum_{m<128} Y[m] e^{+2 pi j m n / 128}          (unnormalised; the caller scales by 1/128)
//
// Split (the radix-2 stage ACROSS the lane pair comes LAST; lane half hf = 0 / 1):
//   lane hf holds   u[i] = Y[2 i + hf], i < 64                     (even bins in lane 0, odd bins in lane 1)
//   E / O = the 64-point inverse transform of the lane's own values  -> the same instruction stream in both lanes
//   x[k] = E[k] + W^k O[k],   x[64 + k] = E[k] - W^k O[k],   W = e^{+2 pi j / 128}
//   -> both lanes form z = (own result) * W^k with the twiddle as a scalar operand; lane 1 needs it, lane 0 ignores it:
//      w = hf ? z : own,  r = partner's w,  result = r + sgn w  (sgn = +1 / -1): lane hf ends up with x[64 hf + k]
// The 64-point transform runs in place as three radix-4 DIF stages with compile-time twiddles (every index below is a
// constant after unrolling: the "array" is 64 named registers); output k ends up in slot xl_fft64_slot(k) (base-4 digit
// reversal).
#ifndef XL_FFT64_H_
#define XL_FFT64_H_

#if defined(__HIPCC__) || defined(__HIP__)
#define XL_FFT_FN static __device__ __forceinline__
// The instruction scheduler may not move anything across this point.  Everything here is straight-line code over 128
// live registers; left alone the scheduler hoists dozens of independent butterflies / twiddles for latency hiding, runs
// out of registers and spills to scratch.  A fence every few independent operations keeps a window of instruction-level
// parallelism and bounds the live ranges.
#define XL_FFT_FENCE() __builtin_amdgcn_sched_barrier(0)
// "This value exists HERE": an empty volatile asm the value passes through.  Without it the optimiser SINKS a result whose
// only users sit in two later branches (the plain / checked phase walks) into both of them and keeps its operands alive
// instead -- two register pairs per point where one would do, 128 registers over the combine stage.
#define XL_FFT_PIN(v) asm volatile("" : "+v"(v))
#else
#define XL_FFT_FN static inline __attribute__((always_inline))
#define XL_FFT_FENCE() ((void)0)
#define XL_FFT_PIN(v) ((void)0)
#endif

// slot of output k of the in-place 64-point DIF transform: base-4 digits reversed
constexpr int xl_fft64_slot(int k) { return ((k & 3) << 4) | (k & 12) | ((k >> 4) & 3); }

// cos / sin of 2 pi n / 128, n < 128, as float literals (rounded once from double; exact at the multiples of 32)
#define XL_W128_COS                                                                                                     \
  {1.0f, 0.99879545f, 0.9951847f, 0.9891765f, 0.98078525f, 0.97003126f, 0.95694035f, 0.94154406f, 0.9238795f,           \
   0.9039893f, 0.8819213f, 0.8577286f, 0.8314696f, 0.8032075f, 0.77301043f, 0.7409511f, 0.70710677f, 0.671559f,         \
   0.6343933f, 0.5956993f, 0.55557024f, 0.51410276f, 0.47139674f, 0.42755508f, 0.38268343f, 0.33688986f, 0.29028466f,   \
   0.24298018f, 0.19509032f, 0.14673047f, 0.09801714f, 0.049067676f, 0.0f, -0.049067676f, -0.09801714f, -0.14673047f,   \
   -0.19509032f, -0.24298018f, -0.29028466f, -0.33688986f, -0.38268343f, -0.42755508f, -0.47139674f, -0.51410276f,      \
   -0.55557024f, -0.5956993f, -0.6343933f, -0.671559f, -0.70710677f, -0.7409511f, -0.77301043f, -0.8032075f,            \
   -0.8314696f, -0.8577286f, -0.8819213f, -0.9039893f, -0.9238795f, -0.94154406f, -0.95694035f, -0.97003126f,           \
   -0.98078525f, -0.9891765f, -0.9951847f, -0.99879545f, -1.0f, -0.99879545f, -0.9951847f, -0.9891765f, -0.98078525f,   \
   -0.97003126f, -0.95694035f, -0.94154406f, -0.9238795f, -0.9039893f, -0.8819213f, -0.8577286f, -0.8314696f,           \
   -0.8032075f, -0.77301043f, -0.7409511f, -0.70710677f, -0.671559f, -0.6343933f, -0.5956993f, -0.55557024f,            \
   -0.51410276f, -0.47139674f, -0.42755508f, -0.38268343f, -0.33688986f, -0.29028466f, -0.24298018f, -0.19509032f,      \
   -0.14673047f, -0.09801714f, -0.049067676f, 0.0f, 0.049067676f, 0.09801714f, 0.14673047f, 0.19509032f, 0.24298018f,   \
   0.29028466f, 0.33688986f, 0.38268343f, 0.42755508f, 0.47139674f, 0.51410276f, 0.55557024f, 0.5956993f, 0.6343933f,   \
   0.671559f, 0.70710677f, 0.7409511f, 0.77301043f, 0.8032075f, 0.8314696f, 0.8577286f, 0.8819213f, 0.9039893f,         \
   0.9238795f, 0.94154406f, 0.95694035f, 0.97003126f, 0.98078525f, 0.9891765f, 0.9951847f, 0.99879545f}

// e^{+2 pi j n / 128} = (cos, sin); sin(2 pi n / 128) = cos(2 pi (n - 32) / 128)
XL_FFT_FN constexpr float xl_w128_cos(int n) {
  constexpr float t[128] = XL_W128_COS;
  return t[n & 127];
}
XL_FFT_FN constexpr float xl_w128_sin(int n) { return xl_w128_cos(n - 32 + 128); }

// ---- arithmetic policy.  The transform is written against three operations so that the device can place the packed
// instructions by hand (op_sel / neg modifiers instead of swaps and sign flips, twiddles as scalar-register operands)
// while the host test runs the same index bookkeeping in plain C++:
//   Ops::template twiddle<N>(v)   v * e^{+2 pi j N / 128}, N a compile-time constant
//   Ops::add_j(a, d)              a + j d
//   Ops::sub_j(a, d)              a - j d
template <class V>
struct XlFftPlainOps {
  template <int N>
  XL_FFT_FN V twiddle(const V v) {
    const float c = xl_w128_cos(N), s = xl_w128_sin(N);
    return (V){v.x * c - v.y * s, v.y * c + v.x * s};
  }
  XL_FFT_FN V add_j(const V a, const V d) { return (V){a.x - d.y, a.y + d.x}; }
  XL_FFT_FN V sub_j(const V a, const V d) { return (V){a.x + d.y, a.y - d.x}; }
};

// one radix-4 butterfly of an INVERSE transform on slots i0 .. i3, followed by the twiddles e^{+2 pi j tw q / 128}, q = 1..3
template <class V, class Ops, int I0, int I1, int I2, int I3, int TW, int NN>
XL_FFT_FN void xl_fft_bfly4(V (&u)[NN]) {
  const V a0 = u[I0], a1 = u[I1], a2 = u[I2], a3 = u[I3];
  const V t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
  const V b0 = t0 + t2, b1 = Ops::add_j(t1, d), b2 = t0 - t2, b3 = Ops::sub_j(t1, d);
  u[I0] = b0;
  if constexpr (TW == 0) {
    u[I1] = b1;
    u[I2] = b2;
    u[I3] = b3;
  } else {
    u[I1] = Ops::template twiddle<TW>(b1);
    u[I2] = Ops::template twiddle<2 * TW>(b2);
    u[I3] = Ops::template twiddle<3 * TW>(b3);
  }
}

// stage with span L (16, 4, 1): groups of 4 L slots; twiddle of element i of a group: W_{4L}^{i q} = W_128^{i q 32 / L}
template <class V, class Ops, int L, int G, int I>
struct XlFftStage {
  XL_FFT_FN void run(V (&u)[64]) {
    constexpr int base = G * 4 * L + I;
    xl_fft_bfly4<V, Ops, base, base + L, base + 2 * L, base + 3 * L, (L > 1 ? I * (32 / L) : 0), 64>(u);
    if constexpr ((G * L + I) % 2 == 1) XL_FFT_FENCE();  // two butterflies (8 slots, 14 temporaries) per scheduling window
    if constexpr (I + 1 < L) XlFftStage<V, Ops, L, G, I + 1>::run(u);
    else if constexpr ((G + 1) * 4 * L < 64) XlFftStage<V, Ops, L, G + 1, 0>::run(u);
  }
};

// in place: u[i] = v[i] in, x-contribution of output k in u[xl_fft64_slot(k)] out
template <class V, class Ops>
XL_FFT_FN void xl_fft64_inverse(V (&u)[64]) {
  XlFftStage<V, Ops, 16, 0, 0>::run(u);
  XlFftStage<V, Ops, 4, 0, 0>::run(u);
  XlFftStage<V, Ops, 1, 0, 0>::run(u);
}

// The cross-lane last stage.  In: u[xl_fft64_slot(k)] = E[k] (lane 0) / O[k] (lane 1).  Out: u[xl_fft64_slot(k)] =
// x[64 hf + k].  ex.select(z, own) = hf ? z : own; ex.template partner<K>(w) = the pair lane's w (device: one DPP move
// per component); sgn = +1 (hf = 0) / -1 (hf = 1).
template <class V, class Ops, class Ex, int K = 0>
XL_FFT_FN void xl_fft128_combine(V (&u)[64], const float sgn, const Ex &ex) {
  constexpr int slot = xl_fft64_slot(K);
  V w = u[slot];
  if constexpr (K != 0) w = ex.select(Ops::template twiddle<K>(u[slot]), u[slot]);
  const V r = ex.template partner<K>(w);
  u[slot] = r + w * (V){sgn, sgn};
  XL_FFT_PIN(u[slot]);
  if constexpr (K % 4 == 3) XL_FFT_FENCE();
  if constexpr (K + 1 < 64) xl_fft128_combine<V, Ops, Ex, K + 1>(u, sgn, ex);
}

// =====================================================================================================================
// The same transform over a QUAD of lanes (32 points each): half the registers per lane, twice the waves per SIMD.
//   lane q (0..3) holds   u[i] = Y[4 i + q], i < 32                (bins = q mod 4)
//   F_q = the 32-point inverse transform of the lane's own values: radix-4 (span 8), radix-4 (span 2), radix-2, in place;
//         output k ends up in slot xl_fft32_slot(k) = 8 (k & 3) + 2 ((k >> 2) & 3) + (k >> 4)
//   x[k + 32 r] = sum_q j^{q r} W^{q k} F_q[k],  W = e^{+2 pi j / 128}  -> z = F_q[k] * W^{q k} (the lane's twiddle, handed in by
//         the caller: device = a 4 x 32 table in LDS), then a radix-4 butterfly across the quad as two exchange stages:
//           A (partner q ^ 2):  t = partner + sA * z                   sA = -1 in lanes 2, 3
//           lane 3:             t = j t
//           B (partner q ^ 1):  r = partner + sB * t                   sB = -1 in lanes 1, 3
//         lane q ends up with x[k + XL_QUAD_NOFF(q)]: 0, 64, 32, 96 for q = 0, 1, 2, 3.
constexpr int xl_fft32_slot(int k) { return 8 * (k & 3) + 2 * ((k >> 2) & 3) + (k >> 4); }
#define XL_QUAD_NOFF(q) (32 * (2 * ((q) & 1) + ((q) >> 1)))

template <class V, class Ops, int L, int G, int I>
struct XlFft32Stage {  // radix-4 stage with span L (8 or 2) of a 32-point transform: twiddle W_{4L}^{i q} = W_128^{i q 32 / L}
  XL_FFT_FN void run(V (&u)[32]) {
    constexpr int base = G * 4 * L + I;
    xl_fft_bfly4<V, Ops, base, base + L, base + 2 * L, base + 3 * L, I * (32 / L), 32>(u);
    if constexpr ((G * L + I) % 2 == 1) XL_FFT_FENCE();
    if constexpr (I + 1 < L) XlFft32Stage<V, Ops, L, G, I + 1>::run(u);
    else if constexpr ((G + 1) * 4 * L < 32) XlFft32Stage<V, Ops, L, G + 1, 0>::run(u);
  }
};

template <class V, int J = 0>
XL_FFT_FN void xl_fft32_radix2(V (&u)[32]) {
  const V a = u[2 * J], b = u[2 * J + 1];
  u[2 * J] = a + b;
  u[2 * J + 1] = a - b;
  if constexpr (J + 1 < 16) xl_fft32_radix2<V, J + 1>(u);
}

template <class V, class Ops>
XL_FFT_FN void xl_fft32_inverse(V (&u)[32]) {
  XlFft32Stage<V, Ops, 8, 0, 0>::run(u);
  XlFft32Stage<V, Ops, 2, 0, 0>::run(u);
  xl_fft32_radix2(u);
}

// ex.template lane_twiddle<K>(v)  v * W^{q K} for THIS lane's q      ex.partner2(v) / ex.partner1(v)  the value of lane q ^ 2 / q ^ 1
// ex.rot_lane3(t)  j t in lane 3, t elsewhere                      sA, sB: the lane's signs
template <class V, class Ex, int K = 0>
XL_FFT_FN void xl_fft128_combine_quad(V (&u)[32], const float sA, const float sB, const Ex &ex) {
  constexpr int slot = xl_fft32_slot(K);
  V z = u[slot];
  if constexpr (K != 0) z = ex.template lane_twiddle<K>(u[slot]);
  const V t = ex.rot_lane3(ex.partner2(z) + z * (V){sA, sA});
  u[slot] = ex.partner1(t) + t * (V){sB, sB};
  XL_FFT_PIN(u[slot]);
  if constexpr (K % 4 == 3) XL_FFT_FENCE();
  if constexpr (K + 1 < 32) xl_fft128_combine_quad<V, Ex, K + 1>(u, sA, sB, ex);
}

// =====================================================================================================================
// Small in-register inverse transforms for the 16 x 8 split of xl_inv8_layout.h (one lane, all indices compile-time).
//   xl_fft16_inverse: x[t] = sum_{k<16} v[k] e^{+2 pi j k t / 16}: radix-4 (span 4, twiddles W_16^{i q} = W_128^{8 i q}),
//                     radix-4 (span 1); output t in slot 4 (t & 3) + (t >> 2)
//   xl_fft8_inverse:  x[g] = sum_{k<8} v[k] e^{+2 pi j k g / 8}:   radix-4 (span 2, twiddles W_8^{i q} = W_128^{16 i q}),
//                     radix-2 on the slot pairs; output g = q + 4 r in slot 2 q + r
template <class V, class Ops, int NN, int L, int G, int I>
struct XlFftNStage {
  XL_FFT_FN void run(V (&u)[NN]) {
    constexpr int base = G * 4 * L + I;
    xl_fft_bfly4<V, Ops, base, base + L, base + 2 * L, base + 3 * L, (L > 1 ? I * (32 / L) : 0), NN>(u);
    if constexpr (I + 1 < L) XlFftNStage<V, Ops, NN, L, G, I + 1>::run(u);
    else if constexpr ((G + 1) * 4 * L < NN) XlFftNStage<V, Ops, NN, L, G + 1, 0>::run(u);
  }
};
template <class V, class Ops>
XL_FFT_FN void xl_fft16_inverse(V (&u)[16]) {
  XlFftNStage<V, Ops, 16, 4, 0, 0>::run(u);
  XlFftNStage<V, Ops, 16, 1, 0, 0>::run(u);
}
template <class V, class Ops>
XL_FFT_FN void xl_fft8_inverse(V (&u)[8]) {
  XlFftNStage<V, Ops, 8, 2, 0, 0>::run(u);
#if defined(__clang__)
#pragma unroll
#endif
  for (int q = 0; q < 4; ++q) {
    const V a = u[2 * q], b = u[2 * q + 1];
    u[2 * q] = a + b;
    u[2 * q + 1] = a - b;
  }
}

#endif  // XL_FFT64_H_

