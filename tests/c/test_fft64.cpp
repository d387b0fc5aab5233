// tests/c/test_fft64.cpp -- host check of sdr-server_amd/csrc/xl_fft64.h (the register FFT of xlp_inverse_reg_kernel):
// a two-"lane" emulation of the in-place 64-point transform + the pair exchange + the output slot map, against a
// double-precision DFT.  Built with the ROCm clang (ext_vector_type), run by tests/test_fft64.py.  Prints the largest
// error relative to the largest output; exit code 1 above 2e-6.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../sdr-server_amd/csrc/xl_fft64.h"

typedef float V __attribute__((ext_vector_type(2)));
typedef XlFftPlainOps<V> Ops;

struct Exchange {
  const V *other;  // the partner lane's transform results (before the combine stage)
  int hf;
  V select(const V z, const V own) const { return hf ? z : own; }
  template <int K>
  V partner(const V) const {  // what the partner lane computes as ITS w for the same k
    const V o = other[xl_fft64_slot(K)];
    if (K == 0) return o;
    return hf ? o : Ops::template twiddle<K>(o);  // partner half = 1 - hf
  }
};

int main() {
  double worst = 0.0;
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&]() {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    return (float)((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0);
  };
  for (int trial = 0; trial < 50; ++trial) {
    V Y[128];
    for (int m = 0; m < 128; ++m) Y[m] = (V){rnd(), rnd()};
    if (trial == 0)
      for (int m = 0; m < 128; ++m) Y[m] = (V){m == 5 ? 1.0f : 0.0f, 0.0f};  // one bin: a pure complex exponential
    V u[2][64], keep[2][64];
    for (int hf = 0; hf < 2; ++hf) {
      for (int i = 0; i < 64; ++i) u[hf][i] = Y[2 * i + hf];
      xl_fft64_inverse<V, Ops>(u[hf]);
      for (int i = 0; i < 64; ++i) keep[hf][i] = u[hf][i];
    }
    for (int hf = 0; hf < 2; ++hf) {
      Exchange ex{keep[1 - hf], hf};
      xl_fft128_combine<V, Ops>(u[hf], hf ? -1.0f : 1.0f, ex);
    }
    double big = 0.0, err = 0.0;
    for (int n = 0; n < 128; ++n) {
      double re = 0.0, im = 0.0;
      for (int m = 0; m < 128; ++m) {
        const double a = 2.0 * M_PI * (double)((m * n) & 127) / 128.0;
        re += (double)Y[m].x * cos(a) - (double)Y[m].y * sin(a);
        im += (double)Y[m].x * sin(a) + (double)Y[m].y * cos(a);
      }
      const V got = u[n >> 6][xl_fft64_slot(n & 63)];
      big = fmax(big, hypot(re, im));
      err = fmax(err, hypot(re - got.x, im - got.y));
    }
    worst = fmax(worst, err / big);
  }
  printf("max relative error %.3g\n", worst);
  // ---- the quad variant: four "lanes" of 32 points
  double worst4 = 0.0;
  for (int trial = 0; trial < 50; ++trial) {
    V Y[128];
    for (int m = 0; m < 128; ++m) Y[m] = (V){rnd(), rnd()};
    if (trial == 0)
      for (int m = 0; m < 128; ++m) Y[m] = (V){m == 77 ? 1.0f : 0.0f, 0.0f};
    V u[4][32], z[4][32], t[4][32];
    for (int q = 0; q < 4; ++q) {
      for (int i = 0; i < 32; ++i) u[q][i] = Y[4 * i + q];
      xl_fft32_inverse<V, Ops>(u[q]);
    }
    // the exchange stages, lane by lane in lockstep (what the DPP moves do on the device)
    for (int k = 0; k < 32; ++k) {
      const int slot = xl_fft32_slot(k);
      for (int q = 0; q < 4; ++q) {
        const int e = (q * k) & 127;
        const float c = xl_w128_cos(e), sn = xl_w128_sin(e);
        const V v = u[q][slot];
        z[q][slot] = (V){v.x * c - v.y * sn, v.y * c + v.x * sn};
      }
      for (int q = 0; q < 4; ++q) {
        const float sA = (q & 2) ? -1.0f : 1.0f;
        V tt = z[q ^ 2][slot] + z[q][slot] * (V){sA, sA};
        if (q == 3) tt = (V){-tt.y, tt.x};
        t[q][slot] = tt;
      }
      for (int q = 0; q < 4; ++q) {
        const float sB = (q & 1) ? -1.0f : 1.0f;
        u[q][slot] = t[q ^ 1][slot] + t[q][slot] * (V){sB, sB};
      }
    }
    double big = 0.0, err = 0.0;
    for (int n = 0; n < 128; ++n) {
      double re = 0.0, im = 0.0;
      for (int m = 0; m < 128; ++m) {
        const double a = 2.0 * M_PI * (double)((m * n) & 127) / 128.0;
        re += (double)Y[m].x * cos(a) - (double)Y[m].y * sin(a);
        im += (double)Y[m].x * sin(a) + (double)Y[m].y * cos(a);
      }
      int q = -1;
      for (int qq = 0; qq < 4; ++qq)
        if (n >= XL_QUAD_NOFF(qq) && n < XL_QUAD_NOFF(qq) + 32) q = qq;
      const V got = u[q][xl_fft32_slot(n - XL_QUAD_NOFF(q))];
      big = fmax(big, hypot(re, im));
      err = fmax(err, hypot(re - got.x, im - got.y));
    }
    worst4 = fmax(worst4, err / big);
  }
  printf("quad variant: max relative error %.3g\n", worst4);
  return worst <= 2e-6 && worst4 <= 2e-6 ? 0 : 1;
}
