// tests/c/test_inv32_layout.cpp -- host check of sdr-server_amd/csrc/xl_inv32_layout.h + the 32- / 4-point register transforms of
// xl_fft16.h (the index bookkeeping of xlp_inverse32_kernel, xl_inv32.hip):
//   1. a wave's data flow -- tile [bin][column] -> role-1 lanes (32-point transforms) -> factors -> exchange through a byte-addressed
//      LDS image, in two rounds of 8 columns -> role-4 lanes (4-point transforms) -> output t + 32 g of column 8 r + 2 k + cc in lane
//      (cc, t), pass k, slot g -- against
//      a double-precision DFT of every column (scaled by 1 / 128);
//   2. the LDS accesses of the exchange, of the factor table, of the phase staging and of the store records through a model of the
//      banks (MI355X_MICROARCH.md, LDS: ds_write_b128 = contiguous 8-lane groups on 32 banks, ds_write_b64 = contiguous 16-lane
//      groups on 32 banks, ds_read_b64 = 32-lane halves on 64 banks, ds_read_b128 = four listed 16-lane groups on 64 banks; identical
//      addresses broadcast): conflict-free except the phase reads (two rows per lane group meet in two banks: at most 2-way);
//   3. the role-1 loads of the two waves of a tile cover it exactly once, four whole 128-byte lines per instruction;
//   4. the regions do not overlap.
// Built with the ROCm clang (ext_vector_type), run by tests/test_inv32_layout.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "../../sdr-server_amd/csrc/xl_fft16.h"
#include "../../sdr-server_amd/csrc/xl_inv32_layout.h"

typedef float V __attribute__((ext_vector_type(2)));
typedef XlFftPlainOps<V> Ops;

static int fails = 0;
#define CHECK(cond, ...)                                                    \
  do {                                                                      \
    if (!(cond)) {                                                          \
      if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } \
      ++fails;                                                              \
    }                                                                       \
  } while (0)

static int worst_conflict(const std::vector<uint32_t> &addr, const std::vector<std::vector<int>> &groups, int width, int banks) {
  int worst = 0;
  for (const auto &g : groups) {
    std::map<int, std::set<uint32_t>> use;  // bank -> distinct addresses (identical addresses broadcast)
    for (int l : g)
      for (int k = 0; k < width / 4; ++k) use[(int)((addr[l] / 4 + k) % banks)].insert(addr[l]);
    for (auto &kv : use) worst = std::max(worst, (int)kv.second.size());
  }
  return worst;
}
static std::vector<std::vector<int>> contiguous_groups(int n) {
  std::vector<std::vector<int>> g;
  for (int b = 0; b < 64; b += n) {
    g.emplace_back();
    for (int l = b; l < b + n; ++l) g.back().push_back(l);
  }
  return g;
}
static std::vector<std::vector<int>> b128_groups() {
  std::vector<std::vector<int>> g(4);
  auto add = [&](int gi, int a, int b) { for (int l = a; l <= b; ++l) g[gi].push_back(l); };
  add(0, 0, 3), add(0, 12, 15), add(0, 20, 27);
  add(1, 4, 11), add(1, 16, 19), add(1, 28, 31);
  add(2, 32, 35), add(2, 44, 47), add(2, 52, 59);
  add(3, 36, 43), add(3, 48, 51), add(3, 60, 63);
  return g;
}

int main() {
  // ---- 3. the loads
  {
    std::vector<int> seen(128 * 32, 0);
    for (uint32_t h = 0; h < 2; ++h)
      for (uint32_t m2 = 0; m2 < 32; ++m2) {
        std::set<uint32_t> lines;
        for (uint32_t j = 0; j < 64; ++j) {
          const uint32_t o = xli32_load(h, j, m2);
          CHECK(o == (xli32_load_m1(j) + 4 * m2) * 32 + 16 * h + xli32_load_c(j), "load offset");
          CHECK(o - xli32_load(h, j, 0) == m2 * 128, "load stride");
          if (o < 128 * 32) ++seen[o];
          lines.insert(o / 16);
        }
        CHECK(lines.size() == 4, "a load instruction touches %zu lines", lines.size());
      }
    for (int o = 0; o < 128 * 32; ++o) CHECK(seen[o] == 1, "tile element %d loaded %d times", o, seen[o]);
  }
  // ---- 1. data flow of the two waves of a tile
  double worst = 0.0;
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&]() {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    return (float)((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0);
  };
  for (int trial = 0; trial < 12; ++trial) {
    static V tile[128 * 32];  // [bin][column]
    static V Y[32][128];
    for (int c = 0; c < 32; ++c)
      for (int m = 0; m < 128; ++m) {
        Y[c][m] = (V){rnd(), rnd()};
        if (trial == 0) Y[c][m] = (V){m == (c * 5 + 3) % 128 ? 1.0f : 0.0f, 0.0f};
        tile[m * 32 + c] = Y[c][m];
      }
    for (uint32_t h = 0; h < 2; ++h) {
      static unsigned char lds[XLI32_WAVE_BYTES];
      memset(lds, 0xFF, sizeof lds);
      for (uint32_t t = 0; t < 32; ++t)  // the factor table, scaled
        for (uint32_t m1 = 0; m1 < 4; ++m1) {
          const double a = 2.0 * M_PI * (double)((m1 * t) & 127) / 128.0;
          const V tw = {(float)(cos(a) / 128.0), (float)(sin(a) / 128.0)};
          memcpy(lds + xli32_tw(t, m1), &tw, 8);
        }
      static V zs[64][32];  // the lanes' factor-multiplied transforms, kept in registers across the rounds
      for (uint32_t j = 0; j < 64; ++j) {  // roles 1, 2
        const uint32_t m1 = xli32_load_m1(j);
        V z[32];
        for (int m2 = 0; m2 < 32; ++m2) z[m2] = tile[xli32_load(h, j, m2)];
        xl_fft32_inverse<V, Ops>(z);
        for (int t = 0; t < 32; ++t) {
          V tw;
          memcpy(&tw, lds + xli32_tw(t, m1), 8);
          const V v = z[xli32_slot32(t)];
          zs[j][t] = (V){v.x * tw.x - v.y * tw.y, v.y * tw.x + v.x * tw.y};
        }
      }
      for (uint32_t r = 0; r < 2; ++r) {
        memset(lds, 0xFF, XLI32_REGION);
        for (uint32_t j = 0; j < 64; ++j) {  // role 3 (writer): the lanes of the round's 8 columns
          const uint32_t m1 = xli32_load_m1(j), c = xli32_load_c(j);
          if ((c >> 3) != r) continue;
          for (int i = 0; i < 16; ++i) {  // one ds_write_b128: (t, t + 1) = (2 i, 2 i + 1)
            CHECK(xli32_exch(c & 7, m1, 2 * i) % 16 == 0 && xli32_exch(c & 7, m1, 2 * i + 1) == xli32_exch(c & 7, m1, 2 * i) + 8, "pair store");
            memcpy(lds + xli32_exch(c & 7, m1, 2 * i), &zs[j][2 * i], 16);
          }
        }
        for (uint32_t k = 0; k < 4; ++k)  // roles 3 (reader) - 4
          for (uint32_t j = 0; j < 64; ++j) {
            const uint32_t c8 = 2 * k + xli32_cc(j), t = xli32_t(j);
            V v[4];
            for (int m1 = 0; m1 < 4; ++m1) memcpy(&v[m1], lds + xli32_exch(c8, m1, t), 8);
            xl_fft4_inverse<V, Ops>(v);
            for (int g = 0; g < 4; ++g) {
              const int n = (int)t + 32 * g;
              const int c = 16 * h + 8 * r + c8;
              double re = 0.0, im = 0.0, big = 0.0;
              for (int m = 0; m < 128; ++m) {
                const double a = 2.0 * M_PI * (double)((m * n) & 127) / 128.0;
                re += (double)Y[c][m].x * cos(a) - (double)Y[c][m].y * sin(a);
                im += (double)Y[c][m].x * sin(a) + (double)Y[c][m].y * cos(a);
                big += hypot((double)Y[c][m].x, (double)Y[c][m].y);
              }
              const V got = v[g];
              const double err = hypot(re / 128.0 - got.x, im / 128.0 - got.y) / (big / 128.0);
              worst = fmax(worst, err);
              CHECK(err < 2e-6, "column %d output %d: got (%g, %g), want (%g, %g)", c, n, got.x, got.y, re / 128.0, im / 128.0);
            }
          }
      }
    }
  }
  printf("largest error / sum |Y|: %.3g\n", worst);
  // ---- 2. banks
  std::vector<uint32_t> a(64);
  for (uint32_t i = 0; i < 16; ++i) {  // exchange, writers: ds_write_b128, fixed pair
    for (uint32_t j = 0; j < 64; ++j) a[j] = xli32_exch(xli32_load_c(j) & 7, xli32_load_m1(j), 2 * i);  // (a group of 8 lanes = one round's columns)
    CHECK(worst_conflict(a, contiguous_groups(8), 16, 32) == 1, "exchange write pair %u conflicts", i);
  }
  for (uint32_t k = 0; k < 4; ++k)
    for (uint32_t m1 = 0; m1 < 4; ++m1) {  // exchange, readers: ds_read_b64
      for (uint32_t j = 0; j < 64; ++j) a[j] = xli32_exch(2 * k + xli32_cc(j), m1, xli32_t(j));
      CHECK(worst_conflict(a, contiguous_groups(32), 8, 64) == 1, "exchange read k=%u m1=%u conflicts", k, m1);
      for (uint32_t j = 0; j < 64; ++j) CHECK(a[j] - a[j & 32] == (j & 31) * 8, "exchange read not contiguous");
    }
  for (uint32_t i = 0; i < 16; ++i) {  // phases, writers: lane (column, gq) stores point 16 gq + i
    for (uint32_t j = 0; j < 64; ++j) a[j] = xli32_phase(xli32_walk_c8(j), 16 * xli32_walk_gq(j) + i);
    CHECK(worst_conflict(a, contiguous_groups(16), 8, 32) == 1, "phase write i=%u conflicts", i);
  }
  int phase_read_worst = 0;
  for (uint32_t k = 0; k < 4; ++k)
    for (uint32_t g = 0; g < 4; ++g) {  // phases, readers: ds_read_b64, lane (cc, t) reads point t + 32 g of column 2 k + cc
      for (uint32_t j = 0; j < 64; ++j) a[j] = xli32_phase(2 * k + xli32_cc(j), xli32_t(j) + 32 * g);
      phase_read_worst = std::max(phase_read_worst, worst_conflict(a, contiguous_groups(32), 8, 64));
    }
  CHECK(phase_read_worst <= 2, "phase reads: %d-way", phase_read_worst);
  printf("phase reads: at most %d-way\n", phase_read_worst);
  for (uint32_t t = 0; t < 32; ++t) {  // factor table [t][m1]: ds_read_b64, 4 distinct addresses (broadcast across the columns)
    for (uint32_t j = 0; j < 64; ++j) a[j] = xli32_tw(t, xli32_load_m1(j));
    CHECK(worst_conflict(a, contiguous_groups(32), 8, 64) == 1, "factor read t=%u conflicts", t);
  }
  for (uint32_t k = 0; k < 8; ++k) {  // store records: ds_read_b128, two distinct addresses
    for (uint32_t j = 0; j < 64; ++j) a[j] = xli32_meta(2 * k + xli32_cc(j));
    CHECK(worst_conflict(a, b128_groups(), 16, 64) == 1, "record read k=%u conflicts", k);
    for (uint32_t j = 0; j < 64; ++j) CHECK(a[j] % 16 == 0, "record not 16-byte aligned");
  }
  // ---- 4. regions
  CHECK(xli32_exch(7, 3, 31) + 8 <= XLI32_REGION && xli32_phase(7, 127) + 8 <= XLI32_REGION, "exchange / phase region");
  CHECK(xli32_tw(0, 0) >= XLI32_REGION && xli32_tw(31, 3) + 8 <= XLI32_META && xli32_meta(15) + 16 <= XLI32_WAVE_BYTES, "tables");
  CHECK(XLI32_WAVE_BYTES * 16u <= 160u * 1024u, "sixteen waves per CU");
  if (fails) {
    printf("inv32 layout: %d FAILED\n", fails);
    return 1;
  }
  printf("inv32 layout: ok\n");
  return 0;
}
