// tests/c/test_mix_layout.cpp -- TEST INFRASTRUCTURE (CPU): the index bookkeeping of the matrix-core mix
// (sdr-server_amd/csrc/xl_mix_layout.h, shared with xlp_mix_mfma_kernel / xlp_tables_h_kernel) driven through an emulation of
// v_mfma_f32_32x32x16_f16's operand and result maps: operand-form image of the branch spectra -> a wave's B registers,
// the staged A operands of a pass, the products, the result registers -> Y[segment][column], against plain complex sums.
// One term, double arithmetic: this checks WHERE every value goes, not the two-half arithmetic (tests/test_mix_split_model.py).
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sdr-server_amd/csrc/xl_mix_layout.h"

typedef std::complex<double> cd;
static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

static int run(uint32_t D, uint32_t M, uint32_t ncg, uint32_t nseg_pass) {
  const uint32_t nkb = (D + 7) / 8, ncols = ncg * 128;
  std::vector<cd> R((size_t)ncols * M * D), X((size_t)D * M * 16);
  for (auto &v : R) v = cd(rnd(), rnd());
  for (auto &v : X) v = cd(rnd(), rnd());
  auto Rat = [&](uint32_t col, uint32_t m, uint32_t b) -> cd & { return R[((size_t)col * M + m) * D + b]; };
  auto Xat = [&](uint32_t b, uint32_t m, uint32_t s) -> cd & { return X[((size_t)b * M + m) * 16 + s]; };
  // ---- operand-form image, as xlp_tables_h_kernel fills it (one thread per (m, b, column); b < 8 nkb, zeros beyond D)
  std::vector<double> Rh((size_t)ncg * M * 4 * 2 * nkb * 64 * 8, 1e30);  // (poisoned: every slot must be written)
  for (uint32_t col = 0; col < ncols; ++col)
    for (uint32_t m = 0; m < M; ++m)
      for (uint32_t b = 0; b < 8 * nkb; ++b) {
        const cd r = b < D ? Rat(col, m, b) : cd(0, 0);
        const uint32_t cg = col / 128, cl = col % 128, w = cl >> 5, ln = xlm_lane(xlm_half(b), cl & 31u);
        for (uint32_t term = 0; term < 2; ++term) {
          const size_t dw = xlm_rh_slot(cg, M, m, w, term, nkb, xlm_kblock(b), ln) * 4u + xlm_dword(b);
          Rh[2 * dw] = term == 0 ? r.real() : 0.0;  // (low half-word, high half-word of the dword; term 1: the second halves -- zero here)
          Rh[2 * dw + 1] = term == 0 ? -r.imag() : 0.0;
        }
      }
  for (double v : Rh)
    if (v == 1e30) return printf("FAIL: an operand slot was never written (D %u)\n", D), 1;
  double worst = 0.0;
  for (uint32_t cg = 0; cg < ncg; ++cg)
    for (uint32_t m = 0; m < M; ++m) {
      // ---- the workgroup (m, cg): staging of one pass, as the kernel's `stage` does it
      std::vector<double> xs((size_t)nkb * 64 * 8, 1e30);
      for (uint32_t w = 0; w < 4; ++w)
        for (uint32_t q = 0; q < (nkb + 3) / 4; ++q) {
          const uint32_t j = xlm_stage_kblock(w, q);
          if (j >= nkb) continue;
          for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t bb = xlm_stage_branch_in_block(lane), sp = xlm_stage_segment_pair(lane), b = 8 * j + bb;
            for (uint32_t u = 0; u < 2; ++u) {
              const uint32_t sl = 2 * sp + u;
              const cd x = (b < D && sl < nseg_pass) ? Xat(b, m, sl) : cd(0, 0);
              const uint32_t sre = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(sl, 0))), sim = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(sl, 1)));
              double *pre = &xs[((size_t)j * 64 + sre) * 8 + 2 * xlm_dword(bb)], *pim = &xs[((size_t)j * 64 + sim) * 8 + 2 * xlm_dword(bb)];
              pre[0] = x.real(), pre[1] = x.imag();
              pim[0] = x.imag(), pim[1] = -x.real();
            }
          }
        }
      for (double v : xs)
        if (v == 1e30) return printf("FAIL: an A-operand slot was never staged (D %u)\n", D), 1;
      for (uint32_t w = 0; w < 4; ++w)
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t h = lane >> 5, c = lane & 31u;
          // ---- the matrix instruction's result registers of this lane: D[row][c] = sum over k-blocks, halves, slots
          double acc[16];
          for (uint32_t g = 0; g < 16; ++g) {
            acc[g] = 0.0;
            const uint32_t row = xlm_result_row(g, h);
            for (uint32_t j = 0; j < nkb; ++j)
              for (uint32_t hh = 0; hh < 2; ++hh)
                for (uint32_t e = 0; e < 8; ++e)
                  acc[g] += xs[((size_t)j * 64 + xlm_lds_slot(xlm_lane(hh, row))) * 8 + e] *
                            Rh[(xlm_rh_slot(cg, M, m, w, 0u, nkb, j, xlm_lane(hh, c))) * 8 + e];
          }
          // ---- the kernel's store loop
          for (uint32_t g2 = 0; g2 < 16; g2 += 2) {
            const uint32_t sl = xlm_result_row(g2, h) >> 1;
            if ((xlm_result_row(g2, h) & 1u) != 0u || xlm_result_row(g2 + 1, h) != xlm_result_row(g2, h) + 1u)
              return printf("FAIL: result registers %u, %u are not the (re, im) rows of one segment\n", g2, g2 + 1), 1;
            if (sl >= nseg_pass) continue;
            cd want(0, 0);
            const uint32_t col = cg * 128 + w * 32 + c;
            for (uint32_t b = 0; b < D; ++b) want += Xat(b, m, sl) * Rat(col, m, b);
            const double err = std::abs(cd(acc[g2], acc[g2 + 1]) - want);
            worst = err > worst ? err : worst;
          }
        }
    }
  printf("D %2u (k-blocks %u) M %u: max |difference| %.3e\n", D, nkb, M, worst);
  return worst < 1e-12 ? 0 : (printf("FAIL\n"), 1);
}

int main() {
  srand(7);
  int bad = 0;
  // every segment of a pass must come out exactly once per (lane half, register pair)
  bool seen[16] = {};
  for (uint32_t h = 0; h < 2; ++h)
    for (uint32_t g2 = 0; g2 < 16; g2 += 2) {
      const uint32_t sl = xlm_result_row(g2, h) >> 1;
      if (sl >= 16 || seen[sl]) bad |= printf("FAIL: segment %u twice or out of range\n", sl);
      seen[sl] = true;
    }
  // the staging writes of one instruction (fixed segment-of-the-pair u and component): 64 lanes, 64 distinct banks of 4 bytes
  for (uint32_t r2 = 0; r2 < 4; ++r2) {
    bool bank[64] = {};
    for (uint32_t lane = 0; lane < 64; ++lane) {
      const uint32_t bb = xlm_stage_branch_in_block(lane), sp = xlm_stage_segment_pair(lane);
      const uint32_t slot = xlm_lds_slot(xlm_lane(xlm_half(bb), xlm_row(2 * sp + (r2 >> 1), r2 & 1u)));
      const uint32_t bk = (slot * 4 + xlm_dword(bb)) & 63u;
      if (bank[bk]) bad |= printf("FAIL: staging write bank %u hit twice (rows +%u)\n", bk, r2);
      bank[bk] = true;
    }
  }
  const uint32_t shapes[][2] = {{42, 3}, {5, 2}, {21, 2}, {8, 1}, {50, 2}, {64, 2}, {1, 1}};
  for (auto &sh : shapes) bad |= run(sh[0], sh[1], 2, 16);
  if (!bad) printf("matrix-core mix layout: ok\n");
  return bad ? 1 : 0;
}
