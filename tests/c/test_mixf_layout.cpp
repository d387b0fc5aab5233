// tests/c/test_mixf_layout.cpp -- TEST INFRASTRUCTURE (CPU): the index bookkeeping of the mix launch on the matrix cores with float32
// operands (sdr-server_amd/csrc/xl_mixf_layout.h, shared with xlp_mix_f32_kernel / xlp_tables_f_kernel) driven through an emulation
// of v_mfma_f32_32x32x2_f32's operand and result maps (lane l: A[row l & 31][k l >> 5], B[k l >> 5][column l & 31]; result register g
// of lane (h, c) = row (g & 3) + 8 (g >> 2) + 4 h of column c): operand-form image of the branch spectra -> a wave's B registers, the
// forward launch's image rows -> A registers by the "float r ^ h, sign h & r & 1" rule, the products, the result registers ->
// Y[segment][column], against plain complex sums.  Double arithmetic: this checks WHERE every value goes.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sdr-server_amd/csrc/xl_mixf_layout.h"

typedef std::complex<double> cd;
static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

static int run(uint32_t D, uint32_t M, uint32_t ncg, uint32_t nseg_pass) {
  const uint32_t nb8 = (D + 7) / 8, ncols = ncg * 128;
  std::vector<cd> R((size_t)ncols * M * D), X((size_t)D * M * 16);
  for (auto &v : R) v = cd(rnd(), rnd());
  for (auto &v : X) v = cd(rnd(), rnd());
  auto Rat = [&](uint32_t col, uint32_t m, uint32_t b) -> cd & { return R[((size_t)col * M + m) * D + b]; };
  // the forward launch's image of one pass: row (b, m) = 16 segments x (re, im) = 32 floats; slots >= nseg_pass hold zeros
  std::vector<double> Ximg((size_t)D * M * 32, 0.0);
  for (uint32_t b = 0; b < D; ++b)
    for (uint32_t m = 0; m < M; ++m)
      for (uint32_t s = 0; s < nseg_pass; ++s) {
        Ximg[((size_t)b * M + m) * 32 + 2 * s] = X[((size_t)b * M + m) * 16 + s].real();
        Ximg[((size_t)b * M + m) * 32 + 2 * s + 1] = X[((size_t)b * M + m) * 16 + s].imag();
      }
  // ---- operand-form image, as xlp_tables_f_kernel fills it (one thread per (m, b, column); b < 8 nb8, zeros beyond D)
  std::vector<double> Rf(xlmf_rf_bytes_per_group(M, nb8) / 4 * ncg, 1e30);  // (poisoned: every element must be written)
  for (uint32_t col = 0; col < ncols; ++col)
    for (uint32_t m = 0; m < M; ++m)
      for (uint32_t b = 0; b < 8 * nb8; ++b) {
        const cd r = b < D ? Rat(col, m, b) : cd(0, 0);
        const uint32_t cg = col / 128, cl = col % 128, w = cl >> 5, c = cl & 31u;
        Rf[xlmf_rf_slot(cg, M, m, w, nb8, xlmf_b_block(b), xlmf_b_half(b), c) * 4u + xlmf_b_elem(b)] = r.real();
        Rf[xlmf_rf_slot(cg, M, m, w, nb8, xlmf_b_block(b), xlmf_b_half(b), 32u + c) * 4u + xlmf_b_elem(b)] = -r.imag();
      }
  for (double v : Rf)
    if (v == 1e30) return printf("FAIL: an operand element was never written (D %u)\n", D), 1;
  double worst = 0.0;
  for (uint32_t cg = 0; cg < ncg; ++cg)
    for (uint32_t m = 0; m < M; m += (M > 4 ? 37 : 1))  // (a few bins: the bin only enters through the slot functions)
      for (uint32_t w = 0; w < 4; ++w) {
        // ---- the wave's registers: B as the kernel loads it (float4 per (k-block, half)), A per branch by the r ^ h rule
        std::vector<double> A((size_t)64 * D), B((size_t)64 * D);
        for (uint32_t lane = 0; lane < 64; ++lane)
          for (uint32_t j = 0; j < D; ++j) {
            B[(size_t)lane * D + j] = Rf[xlmf_rf_slot(cg, M, m, w, nb8, j >> 3, (j >> 2) & 1u, lane) * 4u + (j & 3u)];
            const double x = Ximg[((size_t)j * M + m) * 32 + xlmf_a_float(lane)];
            A[(size_t)lane * D + j] = xlmf_a_negate(lane) ? -x : x;
          }
        // ---- one matrix instruction per branch: D[row][col] += sum_k A[row][k] B[k][col]
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t h = lane >> 5, c = lane & 31u;
          double acc[16];
          for (uint32_t g = 0; g < 16; ++g) {
            acc[g] = 0.0;
            const uint32_t row = xlmf_result_row(g, h);
            for (uint32_t j = 0; j < D; ++j)
              for (uint32_t k = 0; k < 2; ++k) acc[g] += A[(size_t)(k * 32 + row) * D + j] * B[(size_t)(k * 32 + c) * D + j];
          }
          // ---- the kernel's store loop
          for (uint32_t g2 = 0; g2 < 16; g2 += 2) {
            const uint32_t row = xlmf_result_row(g2, h), sl = row >> 1;
            if ((row & 1u) != 0u || xlmf_result_row(g2 + 1, h) != row + 1u)
              return printf("FAIL: result registers %u, %u are not the (re, im) rows of one segment\n", g2, g2 + 1), 1;
            if (sl >= nseg_pass) continue;
            cd want(0, 0);
            const uint32_t col = cg * 128 + w * 32 + c;
            for (uint32_t b = 0; b < D; ++b) want += X[((size_t)b * M + m) * 16 + sl] * Rat(col, m, b);
            const double err = std::abs(cd(acc[g2], acc[g2 + 1]) - want);
            worst = err > worst ? err : worst;
          }
        }
      }
  printf("D %3u (k-blocks %2u) M %3u, %2u segments per pass: max |difference| %.3e\n", D, nb8, M, nseg_pass, worst);
  return worst < 1e-12 ? 0 : (printf("FAIL\n"), 1);
}

int main() {
  srand(11);
  int bad = 0;
  // every segment slot of a pass comes out exactly once per (lane half, register pair)
  bool seen[16] = {};
  for (uint32_t h = 0; h < 2; ++h)
    for (uint32_t g2 = 0; g2 < 16; g2 += 2) {
      const uint32_t sl = xlmf_result_row(g2, h) >> 1;
      if (sl >= 16 || seen[sl]) bad |= printf("FAIL: segment %u twice or out of range\n", sl);
      seen[sl] = true;
    }
  // the A rule reads every float of the 128-byte row exactly once per k
  for (uint32_t h = 0; h < 2; ++h) {
    bool got[32] = {};
    for (uint32_t r = 0; r < 32; ++r) {
      const uint32_t f = xlmf_a_float(h * 32 + r);
      if (f >= 32 || got[f]) bad |= printf("FAIL: float %u of the row read twice (k %u)\n", f, h);
      got[f] = true;
    }
  }
  const uint32_t shapes[][3] = {{42, 128, 16}, {42, 128, 9}, {5, 128, 16}, {21, 256, 16}, {8, 128, 16}, {100, 128, 16}, {113, 128, 16}, {1, 128, 3}};
  for (auto &sh : shapes) bad |= run(sh[0], sh[1], 2, sh[2]);
  if (!bad) printf("float32 matrix-core mix layout: ok\n");
  return bad ? 1 : 0;
}
