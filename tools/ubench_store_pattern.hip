// tools/ubench_store_pattern.hip -- what does the inverse launch's OUTPUT STORE PATTERN cost on an MI355X?  (round 4, session v)
// The polyphase path's inverse launch writes, per (segment, client column), V = 116 consecutive outputs of 8 bytes into the client's
// row [client][time]: pieces of 928 bytes, 8-byte aligned, one per (segment, column), 32 columns per workgroup.  With transform
// and phases compiled out the launch is no faster (profiles/r04_inverse_traffic.txt): its traffic alone sets the time, and the
// write half runs at 3.0 TB/s.  This program times store-only kernels over the same footprint (4096 rows x 8 blocks) in
// variations of that pattern, to see which property costs what:
//   linear        every workgroup writes 32 KB contiguous (16 bytes per lane): the device's plain fill rate
//   pieces<...>   workgroup = (segment, 32 columns) like the launch; per column a piece of PIECE bytes at row + seg * PIECE + shift;
//                 LPC lanes x BPL bytes per store instruction and column (8 x 8 = the 8-lane kernel, 32 x 8 = the LDS kernel, 8 x 16,
//                 16 x 16); SEGS consecutive segments per workgroup; non-temporal or plain
// argv: [rows = 4096] [dynamic LDS bytes per workgroup = 0] [row pitch in bytes = 200704]: 35840 gives the pieces kernels the real launch's 4 workgroups per CU
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_store_pattern.hip -o sdr-server_amd/build/ubench_store_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill_linear(v4f *out, const size_t n16) {
  const size_t base = (size_t)blockIdx.x * 2048u;  // 32 KB per workgroup
  for (int k = 0; k < 8; ++k) {
    const size_t i = base + (size_t)k * 256u + threadIdx.x;
    if (i < n16) out[i] = (v4f){1.0f, 2.0f, 3.0f, (float)k};
  }
}

// workgroup = (segment group, 32 columns); wave = 32 / (4 waves) = 8 columns at LPC lanes per column -> 64 / LPC columns per instruction
template <int LPC, int BPL, int SEGS, bool NT>
__global__ __launch_bounds__(256) void fill_pieces(char *out, const uint32_t ncols, const uint32_t nseg, const size_t row_bytes,
                                                   const uint32_t piece, const uint32_t shift8) {
  constexpr int CPI = 64 / LPC;        // columns per wave instruction
  constexpr int RUN = LPC * BPL;       // contiguous bytes per column and instruction
  const uint32_t sub = blockIdx.x % (ncols / 32u), sg = blockIdx.x / (ncols / 32u);
  const uint32_t w = threadIdx.x >> 6, j = threadIdx.x & 63u;
  const uint32_t lc = j / LPC, ll = j % LPC;
  for (int ss = 0; ss < SEGS; ++ss) {
    const uint32_t s = sg * SEGS + ss;
    if (s >= nseg) return;
    for (int cb = 0; cb < 8 / CPI; ++cb) {  // the wave's 8 columns, CPI at a time
      const uint32_t col = sub * 32u + w * 8u + cb * CPI + lc;
      char *p = out + (size_t)col * row_bytes + (size_t)s * piece + ((col & 1u) ? shift8 : 0u);
      for (uint32_t o = ll * BPL; o + BPL <= piece; o += RUN) {
        if (BPL == 8) {
          const v2f v = {1.0f, (float)o};
          if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2f *>(p + o));
          else *reinterpret_cast<v2f *>(p + o) = v;
        } else {
          typedef float v4u __attribute__((ext_vector_type(4), aligned(8)));
          const v4u v = {1.0f, 2.0f, 3.0f, (float)o};
          if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(p + o));
          else *reinterpret_cast<v4u *>(p + o) = v;
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  const uint32_t ncols = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096u;
  const size_t lds = argc > 2 ? (size_t)atoi(argv[2]) : 0u;  // dynamic LDS per workgroup: caps the workgroups per CU like the real launch's 35 840 bytes do
  const uint32_t nseg = 216u;  // 8 blocks of 27 segments
  const size_t row_bytes = argc > 3 ? (size_t)atol(argv[3]) : 200704;  // >= 216 * 928 + 8, a multiple of 256 (the engine's rows at 8 blocks per call: 199936)
  if (row_bytes < (size_t)nseg * 928 + 8 || ncols % 32u != 0u) {
    fprintf(stderr, "row pitch must hold %u pieces of 928 bytes + 8 (>= %zu), rows a multiple of 32\n", nseg, (size_t)nseg * 928 + 8);
    return 2;
  }
  const size_t total = (size_t)ncols * row_bytes;
  char *out;
  CK(hipMalloc(&out, total + 4096));
  CK(hipMemset(out, 0, total + 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto time = [&](const char *name, size_t bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-78s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, (double)bytes * reps / (ms * 1e-3) / 1e12);
  };
  const size_t n16 = (size_t)ncols * nseg * 928 / 16;
  printf("rows %u, %u segments per row (8 blocks), footprint %.0f MB, %zu bytes of LDS per workgroup (%s)\n", ncols, nseg, (double)ncols * nseg * 928 / 1e6, lds,
         lds ? "workgroups per CU capped by it" : "occupancy not capped");
  time("linear: 32 KB contiguous per workgroup, 16 B per lane", n16 * 16, [&] { hipLaunchKernelGGL(fill_linear, dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, st, reinterpret_cast<v4f *>(out), n16); });
  const dim3 g1(ncols / 32 * nseg), g4(ncols / 32 * ((nseg + 3) / 4));
  const size_t pb = (size_t)ncols * nseg * 928, pa = (size_t)ncols * nseg * 896;
#define RUNP(desc, LPC, BPL, SEGS, NT, grid, piece, shift, bytes) \
  time(desc, bytes, [&] { hipLaunchKernelGGL((fill_pieces<LPC, BPL, SEGS, NT>), grid, dim3(256), lds, st, out, ncols, nseg, row_bytes, piece, shift); })
  RUNP("pieces 928 B (+8 B on odd columns),  8 lanes x  8 B (the 8-lane kernel)", 8, 8, 1, false, g1, 928u, 8u, pb);
  RUNP("pieces 928 B (+8 B on odd columns), 32 lanes x  8 B (the LDS-transform kernel)", 32, 8, 1, false, g1, 928u, 8u, pb);
  RUNP("pieces 928 B (+8 B on odd columns),  8 lanes x 16 B", 8, 16, 1, false, g1, 928u, 8u, pb);
  RUNP("pieces 928 B (+8 B on odd columns), 16 lanes x 16 B", 16, 16, 1, false, g1, 928u, 8u, pb);
  RUNP("pieces 928 B (+8 B on odd columns),  8 lanes x  8 B, non-temporal", 8, 8, 1, true, g1, 928u, 8u, pb);
  RUNP("pieces 928 B, no shift,              8 lanes x  8 B", 8, 8, 1, false, g1, 928u, 0u, pb);
  RUNP("pieces 896 B = 7 lines, no shift,    8 lanes x  8 B (pieces on line boundaries)", 8, 8, 1, false, g1, 896u, 0u, pa);
  RUNP("pieces 896 B = 7 lines, no shift,    8 lanes x 16 B", 8, 16, 1, false, g1, 896u, 0u, pa);
  RUNP("pieces 928 B (+8 B), 4 consecutive segments per workgroup, 8 lanes x 8 B", 8, 8, 4, false, g4, 928u, 8u, pb);
  RUNP("pieces 928 B (+8 B), 4 consecutive segments per workgroup, 16 lanes x 16 B", 16, 16, 4, false, g4, 928u, 8u, pb);
  return 0;
}
