// tests/c/test_inv8_layout.cpp -- host check of sdr-server_amd/csrc/xl_inv8_layout.h + the 16- / 8-point register transforms of
// xl_fft16.h (the index bookkeeping of xlp_inverse8_kernel, xl_inv8.hip):
//   1. a wave's data flow -- tile [bin][column] -> role-1 lanes (16-point transforms) -> twiddles -> exchange through a
//      byte-addressed LDS image -> role-4 lanes (8-point transforms) -> output n of column c in lane (c, n & 7) -- against a
//      double-precision DFT of every column;
//   2. the LDS accesses of the exchange, of the twiddle table and of the phase staging through a model of the banks
//      (MI355X_MICROARCH.md, LDS: ds_write_b64 = contiguous 16-lane groups on 32 banks, ds_read_b64 = 32-lane halves on 64 banks,
//      ds_read_b128 = four listed 16-lane groups on 64 banks; identical addresses broadcast): no group touches a bank twice;
//   3. the role-1 loads cover the tile exactly once.
// Built with the ROCm clang (ext_vector_type), run by tests/test_inv8_layout.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "../../sdr-server_amd/csrc/xl_fft16.h"
#include "../../sdr-server_amd/csrc/xl_inv8_layout.h"

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

// ---- LDS bank model.  width = bytes per lane; groups = the lane sets served in one cycle; banks = 32 or 64 dword banks
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
  // ---- 3. the loads: every (bin, column) of the tile is loaded by exactly one (wave, lane, m2), as bin m1 + 8 m2 of column 8 w + c8
  {
    std::vector<int> seen(128 * 32, 0);
    for (uint32_t w = 0; w < 4; ++w)
      for (uint32_t j = 0; j < 64; ++j)
        for (uint32_t m2 = 0; m2 < 16; ++m2) {
          const uint32_t o = xli8_load(w, j, m2);
          CHECK(o == (xli8_load_m1(j) + 8 * m2) * 32 + 8 * w + xli8_load_c8(j), "load offset");
          CHECK(xli8_load(w, j, m2) - xli8_load(w, j, 0) == m2 * 256, "load stride");
          if (o < 128 * 32) ++seen[o];
        }
    for (int o = 0; o < 128 * 32; ++o) CHECK(seen[o] == 1, "tile element %d loaded %d times", o, seen[o]);
  }
  // ---- 1. data flow of one wave (8 columns)
  double worst = 0.0;
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&]() {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    return (float)((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0);
  };
  for (int trial = 0; trial < 20; ++trial) {
    static V tile[128 * 32];  // [bin][column]
    static V Y[32][128];
    for (int c = 0; c < 32; ++c)
      for (int m = 0; m < 128; ++m) {
        Y[c][m] = (V){rnd(), rnd()};
        if (trial == 0) Y[c][m] = (V){m == (c + 3) % 128 ? 1.0f : 0.0f, 0.0f};
        tile[m * 32 + c] = Y[c][m];
      }
    for (uint32_t w = 0; w < 4; ++w) {
      static unsigned char lds[XLI8_WAVE_BYTES];
      memset(lds, 0xFF, sizeof lds);
      // roles 1 - 3 (writer): lane j = (m1, c8)
      for (uint32_t j = 0; j < 64; ++j) {
        const uint32_t m1 = xli8_load_m1(j), c8 = xli8_load_c8(j);
        V z[16];
        for (int m2 = 0; m2 < 16; ++m2) z[m2] = tile[xli8_load(w, j, m2)];
        xl_fft16_inverse<V, Ops>(z);
        for (int t = 0; t < 16; ++t) {
          const double a = 2.0 * M_PI * (double)((m1 * t) & 127) / 128.0;
          const V tw = {(float)cos(a), (float)sin(a)};
          const V v = z[xli8_slot16(t)];
          const V r = {v.x * tw.x - v.y * tw.y, v.y * tw.x + v.x * tw.y};
          memcpy(lds + xli8_exch(c8, t, m1), &r, 8);
        }
      }
      // roles 3 (reader) - 4: lane j = (c8, u)
      for (uint32_t j = 0; j < 64; ++j) {
        const uint32_t c8 = xli8_col(j), u = xli8_u(j);
        for (int e = 0; e < 2; ++e) {
          V v[8];
          for (int i = 0; i < 4; ++i) memcpy(&v[2 * i], lds + xli8_exch(c8, u + 8 * e, 2 * i), 16);  // one ds_read_b128
          xl_fft8_inverse<V, Ops>(v);
          for (int g = 0; g < 8; ++g) {
            const int n = 16 * g + 8 * e + (int)u;
            const int c = 8 * w + c8;
            double re = 0.0, im = 0.0, big = 0.0;
            for (int m = 0; m < 128; ++m) {
              const double a = 2.0 * M_PI * (double)((m * n) & 127) / 128.0;
              re += (double)Y[c][m].x * cos(a) - (double)Y[c][m].y * sin(a);
              im += (double)Y[c][m].x * sin(a) + (double)Y[c][m].y * cos(a);
              big += hypot((double)Y[c][m].x, (double)Y[c][m].y);
            }
            const V got = v[xli8_slot8(g)];
            const double err = hypot(re - got.x, im - got.y) / big;
            worst = fmax(worst, err);
            CHECK(err < 2e-6, "column %d output %d: got (%g, %g), want (%g, %g)", c, n, got.x, got.y, re, im);
          }
        }
      }
    }
  }
  printf("largest error / sum |Y|: %.3g\n", worst);
  // ---- 2. banks
  std::vector<uint32_t> a(64);
  for (uint32_t t = 0; t < 16; ++t) {  // exchange, writers: ds_write_b64, fixed t
    for (uint32_t j = 0; j < 64; ++j) a[j] = xli8_exch(xli8_load_c8(j), t, xli8_load_m1(j));
    CHECK(worst_conflict(a, contiguous_groups(16), 8, 32) == 1, "exchange write t=%u conflicts", t);
  }
  for (uint32_t e = 0; e < 2; ++e)
    for (uint32_t i = 0; i < 4; ++i) {  // exchange, readers: ds_read_b128, fixed (row half, m1 pair)
      for (uint32_t j = 0; j < 64; ++j) a[j] = xli8_exch(xli8_col(j), xli8_u(j) + 8 * e, 2 * i);
      for (uint32_t j = 0; j < 64; ++j) CHECK(a[j] % 16 == 0, "b128 read not 16-byte aligned");
      CHECK(worst_conflict(a, b128_groups(), 16, 64) == 1, "exchange read e=%u i=%u conflicts", e, i);
    }
  for (uint32_t i = 0; i < 16; ++i) {  // phases, writers: lane (column, g) stores point 16 g + i
    for (uint32_t j = 0; j < 64; ++j) a[j] = xli8_phase(xli8_col(j), 16 * xli8_u(j) + i);
    CHECK(worst_conflict(a, contiguous_groups(16), 8, 32) == 1, "phase write i=%u conflicts", i);
  }
  for (uint32_t g = 0; g < 8; ++g)
    for (uint32_t e = 0; e < 2; ++e) {  // phases, readers: ds_read_b64, lane (c8, u) reads point 16 g + 8 e + u
      for (uint32_t j = 0; j < 64; ++j) a[j] = xli8_phase(xli8_col(j), 16 * g + 8 * e + xli8_u(j));
      CHECK(worst_conflict(a, contiguous_groups(32), 8, 64) == 1, "phase read g=%u e=%u conflicts", g, e);
    }
  for (uint32_t t = 1; t < 16; ++t) {  // twiddle table [t][m1]: ds_read_b64, 8 distinct addresses (broadcast across the columns)
    for (uint32_t j = 0; j < 64; ++j) a[j] = (t * 8 + xli8_load_m1(j)) * 8;
    CHECK(worst_conflict(a, contiguous_groups(32), 8, 64) == 1, "twiddle read t=%u conflicts", t);
  }
  CHECK(8u * XLI8_XCOL <= XLI8_WAVE_BYTES && xli8_phase(7, 127) + 8 <= XLI8_WAVE_BYTES && xli8_exch(7, 15, 7) + 8 <= 8u * XLI8_XCOL, "region sizes");
  if (fails) {
    printf("inv8 layout: %d FAILED\n", fails);
    return 1;
  }
  printf("inv8 layout: ok\n");
  return 0;
}
