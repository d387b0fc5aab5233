// tests/c/test_fused_layout.cpp -- TEST INFRASTRUCTURE (CPU): the index bookkeeping of the fused mix + inverse launch
// (sdr-server_amd/csrc/xl_fused_layout.h + xl_fft64.h, shared with xl_fused.hip) driven through an emulation of
//   * xlp_forward_h_kernel's and xlp_tables_h16_kernel's image writes,
//   * v_mfma_f32_16x16x32_f16's operand and result maps (lane (kg, i): 8 k-slots of row / column i; result lane (g, c),
//     register e = row 4 g + e of column c), 
//   * the rows' (segment, re / im) map with the second form made from the first in registers, the 4 x 32 split of the
//     128-point inverse transform over the four waves, the exchange buffer sub-step by sub-step and the consumer's radix-4 combine,
// against plain complex arithmetic: Y = sum_b X R per bin, y = IDFT_128(Y).  One term (float32 values stand in for the two-half
// operands: tests/test_mix_split_model.py covers that arithmetic): this checks WHERE every value goes.
// Built with the ROCm clang (ext_vector_type), run by tests/test_fused_layout.py.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sdr-server_amd/csrc/xl_fft64.h"
#include "../../sdr-server_amd/csrc/xl_fused_layout.h"

typedef float V __attribute__((ext_vector_type(2)));
typedef XlFftPlainOps<V> Ops;
typedef std::complex<double> cd;
static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

// nseg segments, ncg16 groups of 16 columns
static int run(uint32_t D, uint32_t nseg, uint32_t ncg16) {
  const uint32_t M = XLF_M, nk = xlf_nk(D), nsg = (nseg + 7) / 8, ncols = ncg16 * 16;
  std::vector<cd> X((size_t)nseg * D * M), R((size_t)ncols * D * M);
  for (auto &v : X) v = cd(rnd(), rnd());
  for (auto &v : R) v = cd(rnd(), rnd());
  auto Xat = [&](uint32_t s, uint32_t b, uint32_t m) -> cd & { return X[((size_t)s * D + b) * M + m]; };
  auto Rat = [&](uint32_t c, uint32_t b, uint32_t m) -> cd & { return R[((size_t)c * D + b) * M + m]; };
  // ---- the operand-form images (8 values per 16-byte slot: dword q = (low, high) half-words)
  std::vector<double> Xh(xlf_xh_slots(nsg, nk) * 8, 0.0), Rh((size_t)ncg16 * xlf_rh_bytes_per_cg16(nk) / 16 * 8, 1e30);
  // xlp_forward_h_kernel: workgroup (8 segments s8, branch quad bq), thread (bin m, segment so) writes one slot per term; dead
  // transforms (segment >= nseg, branch >= D) are zeros.  Quads beyond ceil(D / 4) are never written: cleared at allocation
  // (the engine's memset: the 0.0 above) -- but every slot of a live (segment group, branch quad) must be written exactly once.
  std::vector<char> xw(xlf_xh_slots(nsg, nk), 0);
  for (uint32_t s8 = 0; s8 < nsg; ++s8)
    for (uint32_t bq = 0; bq < (D + 3) / 4; ++bq)
      for (uint32_t m = 0; m < M; ++m)
        for (uint32_t so = 0; so < 8; ++so) {
          const uint32_t seg = s8 * 8 + so;
          for (uint32_t term = 0; term < 2; ++term) {
            const size_t slot = xlf_xh_slot(s8, nk, bq >> 2, bq & 3u, m, term, so);
            if (slot >= xw.size()) return printf("FAIL: Xh slot out of range\n"), 1;
            if (term == 0 && xw[slot]++) return printf("FAIL: Xh slot written twice\n"), 1;
            for (uint32_t q = 0; q < 4; ++q) {
              const uint32_t b = bq * 4 + q;
              const cd x = (seg < nseg && b < D) ? Xat(seg, b, m) : cd(0, 0);
              if (xlf_kblock(b) != (bq >> 2) || xlf_kgroup(b) != (bq & 3u) || xlf_dword(b) != q) return printf("FAIL: branch -> slot maps disagree\n"), 1;
              Xh[slot * 8 + 2 * q] = term == 0 ? x.real() : 0.0;
              Xh[slot * 8 + 2 * q + 1] = term == 0 ? x.imag() : 0.0;
            }
          }
        }
  // xlp_tables_h16_kernel: one thread per (m, b < 16 nk, column)
  for (uint32_t col = 0; col < ncols; ++col)
    for (uint32_t m = 0; m < M; ++m)
      for (uint32_t b = 0; b < 16 * nk; ++b) {
        const cd r = b < D ? Rat(col, b, m) : cd(0, 0);
        const uint32_t ln = xlf_lane(xlf_kgroup(b), col & 15u);
        for (uint32_t term = 0; term < 2; ++term) {
          const size_t dw = xlf_rh_slot(col >> 4, nk, m, xlf_kblock(b), term, ln) * 4u + xlf_dword(b);
          Rh[2 * dw] = term == 0 ? r.real() : 0.0;
          Rh[2 * dw + 1] = term == 0 ? -r.imag() : 0.0;
        }
      }
  for (double v : Rh)
    if (v == 1e30) return printf("FAIL: a branch-spectrum operand slot was never written (D %u)\n", D), 1;
  // ---- the fused launch, tile by tile
  double worst = 0.0, big = 0.0;
  std::vector<char> done((size_t)nseg * ncols * M, 0);
  for (uint32_t sg = 0; sg < nsg; ++sg)
    for (uint32_t cg16 = 0; cg16 < ncg16; ++cg16) {
      // d [wave][bin i][lane][e]
      std::vector<double> d((size_t)4 * 32 * 64 * 4);
      for (uint32_t w = 0; w < 4; ++w)
        for (uint32_t i = 0; i < 32; ++i) {
          const uint32_t m = xlf_bin(w, i);
          // operand registers of the 64 lanes, as the kernel's `load` addresses them
          std::vector<double> A((size_t)nk * 64 * 8), B(A.size());
          for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t i16 = lane & 15u, kg = lane >> 4;
            const size_t xb = xlf_xh_slot(sg, nk, 0, 0, 0, 0, 0), xlane = xlf_xh_slot(0, nk, 0, kg, 0, 0, xlf_row_seg(i16));
            const size_t rb = xlf_rh_slot(cg16, nk, 0, 0, 0, 0);
            for (uint32_t j = 0; j < nk; ++j) {
              for (uint32_t e8 = 0; e8 < 8; ++e8) {
                A[((size_t)j * 64 + lane) * 8 + e8] = Xh[(xb + ((size_t)j * 4 * M + m) * 16 + xlane) * 8 + e8];
                B[((size_t)j * 64 + lane) * 8 + e8] = Rh[(rb + ((size_t)m * nk + j) * 128 + lane) * 8 + e8];
              }
              if (xlf_row_comp(i16))  // xlf_imrow: per dword (lo, hi) -> (hi, -lo)
                for (uint32_t q = 0; q < 4; ++q) {
                  double &lo = A[((size_t)j * 64 + lane) * 8 + 2 * q], &hi = A[((size_t)j * 64 + lane) * 8 + 2 * q + 1];
                  const double l0 = lo, h0 = hi;
                  lo = h0, hi = -l0;
                }
            }
          }
          for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t g = lane >> 4, c = lane & 15u;
            for (uint32_t e = 0; e < 4; ++e) {
              const uint32_t row = 4 * g + e;
              double acc = 0.0;
              for (uint32_t j = 0; j < nk; ++j)
                for (uint32_t kg = 0; kg < 4; ++kg)
                  for (uint32_t e8 = 0; e8 < 8; ++e8)
                    acc += A[((size_t)j * 64 + xlf_lane(kg, row)) * 8 + e8] * B[((size_t)j * 64 + xlf_lane(kg, c)) * 8 + e8];
              d[(((size_t)w * 32 + i) * 64 + lane) * 4 + e] = acc;
            }
          }
        }
      // ---- epilogue: half h (registers 2 h, 2 h + 1), sub-step ab (lanes 32 ab ..)
      for (uint32_t h = 0; h < 2; ++h) {
        if (sg * 8 + 4 * h >= nseg) break;
        std::vector<V> z((size_t)4 * 64 * 32);  // [wave][lane][k]: the twiddled transform results, kept in registers
        for (uint32_t w = 0; w < 4; ++w)
          for (uint32_t lane = 0; lane < 64; ++lane) {
            V u[32];
            for (int i = 0; i < 32; ++i)
              u[i] = (V){(float)d[(((size_t)w * 32 + i) * 64 + lane) * 4 + 2 * h], (float)d[(((size_t)w * 32 + i) * 64 + lane) * 4 + 2 * h + 1]};
            xl_fft32_inverse<V, Ops>(u);
            for (uint32_t k = 0; k < 32; ++k) {
              const V t = u[xl_fft32_slot((int)k)];
              const double a = 2.0 * M_PI * (double)((w * k) & 127u) / 128.0;
              z[((size_t)w * 64 + lane) * 32 + k] = (V){(float)(t.x * cos(a) - t.y * sin(a)), (float)(t.x * sin(a) + t.y * cos(a))};
            }
          }
        for (uint32_t ab = 0; ab < 2; ++ab) {
          const uint32_t S0 = sg * 8 + 4 * h + 2 * ab;
          if (S0 >= nseg) break;
          std::vector<V> exch(4 * 32 * 32, (V){1e30f, 1e30f});
          for (uint32_t w = 0; w < 4; ++w)
            for (uint32_t lane = 0; lane < 64; ++lane) {
              if ((lane >> 5) != ab) continue;
              for (uint32_t k = 0; k < 32; ++k) {
                V &dst = exch[xlf_exch(w, lane & 31u, k)];
                if (dst.x != 1e30f) return printf("FAIL: exchange slot written twice\n"), 1;
                dst = z[((size_t)w * 64 + lane) * 32 + k];
              }
            }
          for (uint32_t w = 0; w < 4; ++w) {  // consumer wave w: segment S0 + (w >> 1), columns 8 (w & 1) ..
            const uint32_t seg = S0 + (w >> 1);
            if (seg >= nseg) continue;
            for (uint32_t lane = 0; lane < 64; ++lane) {
              const uint32_t hp = lane >> 5, k = lane & 31u;
              for (uint32_t pp = 0; pp < 4; ++pp) {
                const uint32_t cl = 8 * (w & 1u) + 2 * pp + hp, p = 16 * (w >> 1) + cl;
                const V z0 = exch[xlf_exch(0, p, k)], z1 = exch[xlf_exch(1, p, k)], z2 = exch[xlf_exch(2, p, k)], z3 = exch[xlf_exch(3, p, k)];
                const V t0 = z0 + z2, t1 = z0 - z2, t2 = z1 + z3, t3 = z1 - z3;
                const V y[4] = {t0 + t2, Ops::add_j(t1, t3), t0 - t2, Ops::sub_j(t1, t3)};
                const uint32_t col = cg16 * 16 + cl;
                for (uint32_t q = 0; q < 4; ++q) {
                  const uint32_t n = k + 32 * q;
                  cd want(0, 0);
                  for (uint32_t m = 0; m < M; ++m) {
                    cd Y(0, 0);
                    for (uint32_t b = 0; b < D; ++b) Y += Xat(seg, b, m) * Rat(col, b, m);
                    want += Y * std::polar(1.0, 2.0 * M_PI * (double)((m * n) & 127u) / 128.0);
                  }
                  worst = fmax(worst, std::abs(want - cd(y[q].x, y[q].y)));
                  big = fmax(big, std::abs(want));
                  if (done[((size_t)seg * ncols + col) * M + n]++) return printf("FAIL: an output came out twice\n"), 1;
                }
              }
            }
          }
        }
      }
    }
  for (char v : done)
    if (v != 1) return printf("FAIL: an output never came out (D %u, nseg %u)\n", D, nseg), 1;
  printf("D %2u (k-blocks %u) segments %2u: max |difference| / max |y| %.3e\n", D, nk, nseg, worst / big);
  return worst / big < 3e-6 ? 0 : (printf("FAIL\n"), 1);
}

int main() {
  srand(11);
  int bad = 0;
  // rows -> (segment, component): every (segment, component) once; result registers (2 h, 2 h + 1) of lane group g = (re, im) of
  // segment 4 h + g
  {
    bool seen[16] = {};
    for (uint32_t r = 0; r < 16; ++r) {
      const uint32_t id = 2 * xlf_row_seg(r) + xlf_row_comp(r);
      if (xlf_row_seg(r) >= 8 || seen[id]) bad |= printf("FAIL: row map is not a bijection\n");
      seen[id] = true;
    }
    for (uint32_t g = 0; g < 4; ++g)
      for (uint32_t h = 0; h < 2; ++h)
        for (uint32_t cmp = 0; cmp < 2; ++cmp)
          if (xlf_row_seg(4 * g + 2 * h + cmp) != 4 * h + g || xlf_row_comp(4 * g + 2 * h + cmp) != cmp) bad |= printf("FAIL: result registers (2 h, 2 h + 1) are not segment 4 h + g\n");
  }
  // exchange buffer: 32 producer lanes of one instruction (fixed k) / 32 consumer lanes of one pair -> 32 distinct 8-byte slots mod 32
  for (uint32_t k = 0; k < 32; ++k) {
    bool bank[32] = {};
    for (uint32_t l = 0; l < 32; ++l) {
      const uint32_t bk = xlf_exch(1, l, k) & 31u;
      if (bank[bk]) bad |= printf("FAIL: exchange write bank conflict\n");
      bank[bk] = true;
    }
  }
  for (uint32_t p = 0; p < 32; ++p) {
    bool bank[32] = {};
    for (uint32_t k = 0; k < 32; ++k) {
      const uint32_t bk = xlf_exch(2, p, k) & 31u;
      if (bank[bk]) bad |= printf("FAIL: exchange read bank conflict\n");
      bank[bk] = true;
    }
  }
  // phase staging: the 32 lanes of a store = 16 columns x 2 consecutive table entries, one phase index
  for (uint32_t te0 = 0; te0 < XLF_PH_ENT - 1; te0 += 2)
    for (uint32_t i = 0; i < 16; ++i) {
      bool bank[32] = {};
      for (uint32_t l = 0; l < 32; ++l) {
        const uint32_t cc = l & 15u, te = te0 + (l >> 4), bk = (cc * XLF_PH_ROW + te * 16 + i) & 31u;
        if (bank[bk]) bad |= printf("FAIL: phase staging bank conflict\n");
        bank[bk] = true;
      }
    }
  if (XLF_PH_ENT * 16 > XLF_PH_ROW) bad |= printf("FAIL: phase staging row too short\n");
  // the forward launch's gather: 32 lanes = 4 bins x 8 rows, row pitch = 4 mod 32 elements -> 32 distinct 8-byte bank pairs
  {
    const uint32_t pitch = (128 + 128 / 4) + 4;
    bool bank[32] = {};
    for (uint32_t l = 0; l < 32; ++l) {
      const uint32_t m = l >> 3, so = l & 7u, bk = (so * pitch + m) & 31u;
      if (bank[bk]) bad |= printf("FAIL: forward gather bank conflict\n");
      bank[bk] = true;
    }
  }
  const struct { uint32_t D, nseg, ncg16; } shapes[] = {{42, 19, 2}, {42, 8, 1}, {5, 16, 1}, {21, 27, 1}, {64, 9, 1}, {16, 3, 1}, {1, 5, 1}, {50, 13, 1}};
  for (auto &sh : shapes) bad |= run(sh.D, sh.nseg, sh.ncg16);
  if (!bad) printf("fused layout: ok\n");
  return bad ? 1 : 0;
}
