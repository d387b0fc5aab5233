// tools/ubench_tile_copy.hip -- the polyphase path's inverse launch AS A MEMORY KERNEL: what does its traffic alone cost on an
// MI355X, and which property of it costs what?  (round 5; VERDICT r4 item 3: "tune the inverse launch as the memory kernel it is")
// The launch reads, per workgroup, one 32 KB tile of mixed spectra (128 bins x 32 client columns x 8 bytes, contiguous) and writes,
// per column, V = 116 consecutive outputs of 8 bytes into the client's row [client][time] (pieces of 928 bytes, 8-byte aligned,
// odd columns 8 bytes later).  This program times kernels that do ONLY that, over the launch's footprint at 4096 clients x 8 blocks
// (216 segments: 906 MB in, 821 MB out), in variations:
//   copy_linear      read 32 KB contiguous, write 29 KB contiguous per workgroup: the device's plain copy rate at this read : write mix
//   tile_pieces<W>   the launch's pattern: 32 lanes x 8 bytes per column and store instruction (the LDS-transform kernel's epilogue);
//                    W = 0: lane l writes piece element l + 32 r (runs start wherever the piece starts);
//                    W = 1: lane l writes the row element whose index is l mod 32 -- every store instruction covers two whole,
//                    aligned 128-byte lines per column (5 instead of 4 instructions per column, the ragged ends masked)
//   tile_pieces_pipe the same, persistent: a workgroup walks tiles with the next tile's loads in flight while it stores this one's
// argv: [rows = 4096] [dynamic LDS bytes per workgroup = 0: caps the workgroups per CU like the real launch's 32 KB + registers (4 per CU)]
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_tile_copy.hip -o sdr-server_amd/build/ubench_tile_copy
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr uint32_t V = 116u, NSEG = 216u, TILE16 = 2048u;  // a tile = 2048 x 16 bytes

__global__ __launch_bounds__(256) void copy_linear(const v4f *__restrict__ in, v4f *__restrict__ out, const uint32_t out16) {
  const size_t ib = (size_t)blockIdx.x * TILE16, ob = (size_t)blockIdx.x * out16;
  v4f z[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = in[ib + k * 256u + threadIdx.x];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k * 256u + threadIdx.x < out16) out[ob + k * 256u + threadIdx.x] = z[k];
}

__device__ __forceinline__ void load_tile(const v4f *__restrict__ in, const size_t tile, v4f (&z)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = in[tile * TILE16 + k * 256u + threadIdx.x];
}

// the wave's 8 columns, two per store instruction (lane = 32 * (column & 1) + l)
template <int WINDOWS>
__device__ __forceinline__ void store_pieces(char *__restrict__ out, const uint32_t sub, const uint32_t s, const size_t row_bytes,
                                             const v4f (&z)[8]) {
  const uint32_t w = threadIdx.x >> 6, j = threadIdx.x & 63u, l = j & 31u;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const uint32_t col = sub * 32u + w * 8u + cb * 2u + (j >> 5);
    v2f *__restrict__ row = reinterpret_cast<v2f *>(out + (size_t)col * row_bytes);
    const uint32_t first = s * V + (col & 1u);  // the piece: row elements first .. first + V - 1
    if (WINDOWS == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t qo = l + 32u * r;
        const v4f t = z[(cb * 4 + r) & 7];
        if (qo < V) row[first + qo] = (v2f){t.x, t.y + (float)cb};
      }
    } else {
      const uint32_t base = first & ~31u;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const uint32_t i = base + 32u * r + l;
        const v4f t = z[(cb * 4 + r) & 7];
        if (i >= first && i < first + V) row[i] = (v2f){t.z, t.w + (float)cb};
      }
    }
  }
}

template <int WINDOWS>
__global__ __launch_bounds__(256) void tile_pieces(const v4f *__restrict__ in, char *__restrict__ out, const uint32_t nsub,
                                                   const size_t row_bytes) {
  const uint32_t sub = blockIdx.x % nsub, s = blockIdx.x / nsub;
  v4f z[8];
  load_tile(in, blockIdx.x, z);
  store_pieces<WINDOWS>(out, sub, s, row_bytes, z);
}

template <int WINDOWS>
__global__ __launch_bounds__(256) void tile_pieces_pipe(const v4f *__restrict__ in, char *__restrict__ out, const uint32_t nsub,
                                                        const size_t row_bytes, const uint32_t ntiles) {
  v4f za[8], zb[8];
  uint32_t t = blockIdx.x;
  if (t >= ntiles) return;
  load_tile(in, t, za);
  for (;;) {
    const uint32_t t1 = t + gridDim.x;
    if (t1 < ntiles) load_tile(in, t1, zb);
    store_pieces<WINDOWS>(out, t % nsub, t / nsub, row_bytes, za);
    if (t1 >= ntiles) return;
    const uint32_t t2 = t1 + gridDim.x;
    if (t2 < ntiles) load_tile(in, t2, za);
    store_pieces<WINDOWS>(out, t1 % nsub, t1 / nsub, row_bytes, zb);
    if (t2 >= ntiles) return;
    t = t2;
  }
}

// loads only / stores only, the launch's patterns
__global__ __launch_bounds__(256) void tile_loads(const v4f *__restrict__ in, char *__restrict__ out) {
  v4f z[8];
  load_tile(in, blockIdx.x, z);
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) a += z[k].x + z[k].y + z[k].z + z[k].w;
  if (a == 1.2345e-33f) out[threadIdx.x] = 1;
}
template <int WINDOWS>
__global__ __launch_bounds__(256) void tile_stores(char *__restrict__ out, const uint32_t nsub, const size_t row_bytes) {
  v4f z[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) z[k] = (v4f){(float)k, 1.f, (float)threadIdx.x, 2.f};
  store_pieces<WINDOWS>(out, blockIdx.x % nsub, blockIdx.x / nsub, row_bytes, z);
}

// The MIX launch's store pattern: workgroup = (bin m, column group of 128 columns), wave = 32 columns (one tile column block), all
// passes of 16 segments: per pass and wave 8 store instructions of two 256-byte runs (segments 2 h + cs, lanes = 32 columns x 8 bytes)
// into Y[cg][segment][sub][bin][32]: runs of 256 bytes, 32 KB apart.  MAP 0: workgroup index = bin fastest (adjacent bins on
// different XCDs, as the launch has it); MAP 1: XCD x (= index % 8) takes bins 16 x .. 16 x + 15 of a column group: adjacent runs
// come from one XCD's CUs at about the same time.  NT: non-temporal stores (the launch's) or plain.
template <int MAP, bool NT>
__global__ __launch_bounds__(256) void mix_stores(v2f *__restrict__ Y, const uint32_t ncg, const uint32_t nseg) {
  const uint32_t bid = blockIdx.x;
  uint32_t m, cg;
  if (MAP == 0) m = bid & 127u, cg = bid >> 7;
  else {
    const uint32_t x = bid & 7u, idx = bid >> 3;
    m = 16u * x + (idx & 15u), cg = idx >> 4;
  }
  if (cg >= ncg) return;
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, h = lane >> 5, c = lane & 31u;
  v2f *__restrict__ Yc = Y + ((((size_t)cg * nseg) * 4u + w) * 128u + m) * 32u + c;
  const size_t ystride = 4u * 128u * 32u;  // v2f per segment
  for (uint32_t s0 = 0; s0 < nseg; s0 += 16u) {
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      const uint32_t cs = (uint32_t)((g2 & 1) + 4 * (g2 >> 1));
      const uint32_t sg = s0 + 2u * h + cs;
      const v2f y = {(float)sg, (float)m};
      if (sg < nseg) {
        if (NT) __builtin_nontemporal_store(y, Yc + (size_t)sg * ystride);
        else Yc[(size_t)sg * ystride] = y;
      }
    }
  }
}

int main(int argc, char **argv) {
  const uint32_t ncols = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096u;
  const size_t lds = argc > 2 ? (size_t)atoi(argv[2]) : 0u;
  const size_t row_bytes = 200704;  // >= 216 * 928 + 8, a multiple of 256
  const uint32_t nsub = ncols / 32u, ntiles = nsub * NSEG;
  const size_t in_bytes = (size_t)ntiles * 32768u, out_bytes = (size_t)ncols * row_bytes;
  char *in, *out;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, out_bytes + 4096));
  CK(hipMemset(in, 0, in_bytes));
  CK(hipMemset(out, 0, out_bytes + 4096));
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
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-92s %8.1f us  %6.2f TB/s  (%5.1f us per block)\n", name, ms * 1e3 / reps, (double)bytes * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps / 8.0);
  };
  const size_t rd = in_bytes, wr = (size_t)ncols * NSEG * V * 8u;
  printf("rows %u, %u segments (8 blocks): %.0f MB read, %.0f MB written per launch; %zu bytes of dynamic LDS per workgroup\n", ncols, NSEG, rd / 1e6, wr / 1e6, lds);
  const v4f *in4 = reinterpret_cast<const v4f *>(in);
  const uint32_t out16 = 32u * V * 8u / 16u;  // 1856 x 16 bytes written per workgroup
  time("copy_linear: 32 KB in, 29 KB out per workgroup, both contiguous", rd + wr, [&] { hipLaunchKernelGGL(copy_linear, dim3(ntiles), dim3(256), lds, st, in4, reinterpret_cast<v4f *>(out), out16); });
  time("tile loads only", rd, [&] { hipLaunchKernelGGL(tile_loads, dim3(ntiles), dim3(256), lds, st, in4, out); });
  time("piece stores only, runs from the piece start (the launch's pattern)", wr, [&] { hipLaunchKernelGGL(tile_stores<0>, dim3(ntiles), dim3(256), lds, st, out, nsub, row_bytes); });
  time("piece stores only, aligned 256-byte windows", wr, [&] { hipLaunchKernelGGL(tile_stores<1>, dim3(ntiles), dim3(256), lds, st, out, nsub, row_bytes); });
  time("tile -> pieces, runs from the piece start (the launch's pattern)", rd + wr, [&] { hipLaunchKernelGGL(tile_pieces<0>, dim3(ntiles), dim3(256), lds, st, in4, out, nsub, row_bytes); });
  time("tile -> pieces, aligned 256-byte windows", rd + wr, [&] { hipLaunchKernelGGL(tile_pieces<1>, dim3(ntiles), dim3(256), lds, st, in4, out, nsub, row_bytes); });
  for (uint32_t per_cu : {2u, 4u, 8u}) {
    char name[160];
    snprintf(name, sizeof name, "tile -> pieces, persistent (%u workgroups per CU), next tile's loads in flight, piece start", per_cu);
    time(name, rd + wr, [&] { hipLaunchKernelGGL(tile_pieces_pipe<0>, dim3(256u * per_cu), dim3(256), 0, st, in4, out, nsub, row_bytes, ntiles); });
    snprintf(name, sizeof name, "tile -> pieces, persistent (%u workgroups per CU), next tile's loads in flight, aligned windows", per_cu);
    time(name, rd + wr, [&] { hipLaunchKernelGGL(tile_pieces_pipe<1>, dim3(256u * per_cu), dim3(256), 0, st, in4, out, nsub, row_bytes, ntiles); });
  }
  {
    const uint32_t ncg = ncols / 128u;
    v2f *Y = reinterpret_cast<v2f *>(in);  // (the tile image: 906 MB at 4096 rows)
    printf("the mix launch's store pattern alone (%u workgroups, each all %u segments of one bin and 128 columns):\n", 128u * ncg, NSEG);
    time("mix stores, bin fastest across the grid (adjacent bins on different XCDs), non-temporal", rd, [&] { hipLaunchKernelGGL((mix_stores<0, true>), dim3(128u * ncg), dim3(256), lds, st, Y, ncg, NSEG); });
    time("mix stores, bin fastest across the grid, plain stores", rd, [&] { hipLaunchKernelGGL((mix_stores<0, false>), dim3(128u * ncg), dim3(256), lds, st, Y, ncg, NSEG); });
    time("mix stores, 16 adjacent bins per XCD, non-temporal", rd, [&] { hipLaunchKernelGGL((mix_stores<1, true>), dim3(128u * ncg), dim3(256), lds, st, Y, ncg, NSEG); });
    time("mix stores, 16 adjacent bins per XCD, plain stores", rd, [&] { hipLaunchKernelGGL((mix_stores<1, false>), dim3(128u * ncg), dim3(256), lds, st, Y, ncg, NSEG); });
  }
  return 0;
}
