// tools/ubench_chain5.hip -- is the NCO chain equally fast on all XCDs?  Blocks 0..63 run the store-free chain (one
// wave each; block b lands on XCD b % 8), optionally next to HBM-streaming workgroups; prints ns/step per XCD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain5.hip -o sdr-server_amd/build/ubench_chain5
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f nxt(v2f p, v2f q) {
  v2f t1, t2, r;
  asm volatile("v_pk_mul_f32 %0, %3, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
               "v_pk_add_f32 %2, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(t1), "=&v"(t2), "=&v"(r) : "v"(p), "v"(q));
  return r;
}

__global__ __launch_bounds__(256) void k(float *tab, const v4f *big, size_t nbig, float *sink, int steps, unsigned long long *res, volatile int *done) {
  const int l = threadIdx.x & 63;
  if (blockIdx.x < 64) {
    if (threadIdx.x >= 64) return;
    __builtin_amdgcn_s_setprio(3);
    v2f p = {1.0f, 1e-3f * l}, q = {0.9999f, 0.01f};
    const long long t0 = wall_clock64();
    for (int m = 0; m < steps; m += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) p = nxt(p, q);
    }
    const long long t1 = wall_clock64();
    if (l == 0) {
      res[2 * blockIdx.x] = t1 - t0;
      res[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 7u;
      atomicAdd((int *)done, 1);
    }
    tab[blockIdx.x * 64 + l] = p.x;
    return;
  }
  v4f acc = {0, 0, 0, 0};
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  long long n = 0;
  while (*done < 64 && n < 200000) {
    v4f v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = big[i % nbig]; i += stride; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    ++n;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  float *tab, *sink; v4f *big; unsigned long long *res, h[128]; int *flag;
  const size_t nbig = (size_t)1 << 27;
  (void)hipMalloc(&tab, 64 * 64 * 4); (void)hipMalloc(&big, nbig * 16); (void)hipMalloc(&res, sizeof(h)); (void)hipMalloc(&flag, 4);
  (void)hipMalloc(&sink, 4096 * 256 * 4);
  (void)hipMemset(big, 0, nbig * 16);
  const int steps = 3120;
  for (int streamers : {0, 2048}) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemset(flag, 0, 4);
      hipLaunchKernelGGL(k, dim3(64 + streamers), dim3(256), 0, 0, tab, big, nbig, sink, steps, res, flag);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost);
    double sum[8] = {0}, mx[8] = {0}; int cnt[8] = {0};
    for (int b = 0; b < 64; ++b) { const int x = (int)h[2 * b + 1]; const double ns = (double)h[2 * b] * 10.0 / steps; sum[x] += ns; cnt[x]++; if (ns > mx[x]) mx[x] = ns; }
    printf("%4d streaming workgroups: ns/step per XCD (mean, max over %d waves):", streamers, cnt[0]);
    for (int x = 0; x < 8; ++x) printf("  [%d] %.2f %.2f", x, cnt[x] ? sum[x] / cnt[x] : 0.0, mx[x]);
    printf("\n");
  }
  return 0;
}
