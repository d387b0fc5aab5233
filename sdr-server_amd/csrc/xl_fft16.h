// xl_fft16.h -- the 32-, 16-, 8- and 4-point inverse DFTs one lane runs in its registers in the inverse launches of the polyphase
// path (xl_inv8.hip: the 128-point transform cut 16 x 8, xl_inv8_layout.h; xl_inv32.hip: cut 32 x 4, xl_inv32_layout.h).  Written so
// that the same text compiles for the device (V = float ext_vector_type(2), packed instructions placed by hand: XlpFftOps in
// xl_poly_dev.h) and for the host (tests/c/test_inv8_layout.cpp, test_inv32_layout.cpp: an emulation of the lanes checked against a
// double-precision DFT), because the index bookkeeping
// of a register FFT is the kind of thing that is either exactly right or silently wrong.  All indices are compile-time constants:
// the "arrays" are named registers.  (Rounds 3-4 also held 64- and 32-point transforms here for inverse kernels that kept a whole
// column in one lane pair / quad; they measured slower and live in tools/experiments/retired/xl_fft64.h.txt.)
#ifndef XL_FFT16_H_
#define XL_FFT16_H_

#if defined(__HIPCC__) || defined(__HIP__)
#define XL_FFT_FN static __device__ __forceinline__
#else
#define XL_FFT_FN static inline __attribute__((always_inline))
#endif

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

// 32-point inverse transform for the 32 x 4 split of xl_inv32_layout.h: x[t] = sum_{k<32} v[k] e^{+2 pi j k t / 32}: radix-4 (span 8,
// twiddles W_32^{i q} = W_128^{4 i q}), radix-4 (span 2, W_8^{i q}), radix-2 on the slot pairs; output t = q + 4 q' + 16 r in slot
// 8 q + 2 q' + r
template <class V, class Ops>
XL_FFT_FN void xl_fft32_inverse(V (&u)[32]) {
  XlFftNStage<V, Ops, 32, 8, 0, 0>::run(u);
  XlFftNStage<V, Ops, 32, 2, 0, 0>::run(u);
#if defined(__clang__)
#pragma unroll
#endif
  for (int q = 0; q < 16; ++q) {
    const V a = u[2 * q], b = u[2 * q + 1];
    u[2 * q] = a + b;
    u[2 * q + 1] = a - b;
  }
}
// 4-point inverse transform, natural order in and out: x[g] = sum_{k<4} v[k] (+j)^{k g}
template <class V, class Ops>
XL_FFT_FN void xl_fft4_inverse(V (&u)[4]) {
  xl_fft_bfly4<V, Ops, 0, 1, 2, 3, 0, 4>(u);
}

#endif  // XL_FFT16_H_
