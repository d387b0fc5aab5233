// xl_y6.h -- the 48-bit form of a complex float32 value in the mixed-spectra image Y of the polyphase path (xl_polyphase.hip:
// written by xlp_mix_mfma_kernel, read by xlp_inverse_kernel): ONE 6-bit exponent shared by the two components and two signed
// 21-bit mantissas, instead of two float32 -- Y is written once and read once per call and was 58 % of the path's HBM traffic.
//
//   value = mantissa * 2^(XLY6_SH - off)     off = clamp(XLY6_TOP - biased exponent of max(|re|, |im|), 0, 63)
//
// so the larger component keeps 20 significant bits (its mantissa lies in [2^19, 2^20): rounding error <= 2^-20 of it, 2^-21 at
// the top of a binade), the smaller one the same absolute step.  What is encoded are the matrix-core mix's UNSCALED sums: both operand families are scaled into fixed ranges
// (|X * 128| < 2^15.6, |R * column scale| < 2^13: xl_polyphase.h), so |sum| < 84 * 2^28.6 < 2^35 whatever the taps' gain, and the
// column's power-of-two factor that undoes the scales (cscale) is applied by the reader as an exponent offset -- exactly.
// Layout of a (segment, 32- or 16-column) tile: [bin][column PAIR][12 bytes] = the two columns' low words (mantissa of re, low
// 11 bits of the mantissa of im) and one word with their two high half-words (high 10 bits of the mantissa of im, off): 6 CW
// bytes per bin in a pitch of 8 CW.  A pair is what one thread of the inverse launch's tile fill handles -- one 12-byte load where the float32 form
// took one of 16 -- and what an even lane of the mix launch stores after fetching its odd neighbour's value through DPP: one
// 12-byte store per two values, 192 contiguous bytes per (segment, bin).  (A first layout with a plane of low words and a plane of
// half-words moved the same bytes with twice the store instructions, half of them 2-byte stores: 16 % fewer bytes, 10 % MORE
// time -- profiles/r04_y48.txt.)  Compiles for the host too: tests/test_y6_model.py checks the arithmetic against numpy.
#ifndef XL_Y6_H_
#define XL_Y6_H_
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define XLY_FN static __host__ __device__ __forceinline__
#else
#define XLY_FN static inline
#endif

#define XLY6_TOP 163  // biased float32 exponent of 2^36: above every sum the mix can form
#define XLY6_SH (XLY6_TOP - 127 - 19)  // the larger component's mantissa lands in [2^19, 2^20)

XLY_FN uint32_t xly6_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, sizeof(u));
  return u;
}

XLY_FN void xly6_encode(float re, float im, uint32_t *lo, uint32_t *hi16) {
  const float m = fmaxf(fabsf(re), fabsf(im));
  int off = XLY6_TOP - (int)((xly6_bits(m) >> 23) & 0xFFu);
  off = off < 0 ? 0 : (off > 63 ? 63 : off);
  // (rintf: round to nearest even, the conversion is then exact; the clamp only ever acts on a mantissa that rounds up to 2^20)
  int ma = (int)rintf(ldexpf(re, off - XLY6_SH)), mb = (int)rintf(ldexpf(im, off - XLY6_SH));
  ma = ma > 1048575 ? 1048575 : (ma < -1048575 ? -1048575 : ma);
  mb = mb > 1048575 ? 1048575 : (mb < -1048575 ? -1048575 : mb);
  *lo = ((uint32_t)ma & 0x1FFFFFu) | ((uint32_t)mb << 21);
  *hi16 = (((uint32_t)mb >> 11) & 0x3FFu) | ((uint32_t)off << 10);
}

// kexp: exponent of the column's power-of-two factor (cscale = 2^kexp), folded into the scaling
XLY_FN void xly6_decode(uint32_t lo, uint32_t hi16, int kexp, float *re, float *im) {
  const int ma = (int)(lo << 11) >> 11;
  const int mb = (int)(((lo >> 21) | ((hi16 & 0x3FFu) << 11)) << 11) >> 11;
  const int sh = XLY6_SH - (int)((hi16 >> 10) & 63u) + kexp;
  *re = ldexpf((float)ma, sh);
  *im = ldexpf((float)mb, sh);
}

// byte offsets inside the Y image: tile (cg, segment, sub) and, inside it, the 12 bytes of (bin m, column pair cw / 2)
// (a bin's row of CW / 2 pairs = 6 CW bytes sits in a pitch of 8 CW bytes -- 192 of 256, 96 of 128 -- so that no 128-byte line is
// shared by two bins: their rows are written by different workgroups of the mix launch, on different XCDs)
XLY_FN size_t xly6_tile(uint32_t cg, uint32_t nseg_cap, uint32_t seg, uint32_t nsub, uint32_t sub, uint32_t M, uint32_t CW) {
  return ((((size_t)cg * nseg_cap + seg) * nsub + sub) * M) * CW * 8u;
}
XLY_FN uint32_t xly6_pair(uint32_t m, uint32_t CW, uint32_t pair) { return m * CW * 8u + pair * 12u; }

#endif  // XL_Y6_H_
