// tools/ubench_chain4.hip -- does a memory-saturated chip slow the NCO chain through its TABLE STORES?  Block 0 runs the
// chain (64 lanes, one table row per lane) storing a dwordx4 every `every` steps (0 = never); all other workgroups stream
// a 2 GB buffer from HBM (dwordx4 loads, 8 in flight per lane) until the chain is done.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain4.hip -o sdr-server_amd/build/ubench_chain4
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

template <int EVERY>  // store one dwordx4 (2 table entries) every EVERY steps; 0 = no stores
__global__ __launch_bounds__(256) void k(float *tab, const v4f *big, size_t nbig, float *sink, int steps, long long *cyc, volatile int *stopflag,
                                         int write_too) {
  const int l = threadIdx.x & 63;
  if (blockIdx.x == 0) {
    if (threadIdx.x >= 64) return;
    __builtin_amdgcn_s_setprio(3);
    v2f p = {1.0f, 1e-3f * l}, q = {0.9999f, 0.01f};
    v4f *o4 = (v4f *)(tab + (size_t)l * 6400);
    const long long t0 = wall_clock64();
    for (int m = 0; m + 32 <= steps; m += 32) {
      v2f s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] = p;
#pragma unroll
        for (int i = 0; i < 8; ++i) p = nxt(p, q);
        if (EVERY == 8) o4[(m >> 3) + j] = (v4f){s[j].x, s[j].y, p.x, p.y};
        if (EVERY == 16 && (j & 1)) o4[(m >> 4) + (j >> 1)] = (v4f){s[j - 1].x, s[j - 1].y, s[j].x, s[j].y};
      }
      if (EVERY == 32) o4[m >> 5] = (v4f){s[0].x, s[0].y, s[2].x, s[2].y};
    }
    const long long t1 = wall_clock64();
    if (l == 0) { cyc[0] = t1 - t0; *stopflag = 1; }
    tab[l] += p.x;
    return;
  }
  // streamers
  v4f acc = {0, 0, 0, 0};
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  long long n = 0;
  while (*stopflag == 0 && n < 200000) {
    v4f v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = big[i % nbig]; i += stride; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    if (write_too) ((v4f *)big)[(i + 77) % nbig] = acc;
    ++n;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  float *tab, *sink; v4f *big; long long *cyc, h; int *flag;
  const size_t nbig = (size_t)1 << 27;  // 2 GB of v4f
  (void)hipMalloc(&tab, 64 * 6400 * 4 * 4 + 8192); (void)hipMalloc(&big, nbig * 16); (void)hipMalloc(&cyc, 16); (void)hipMalloc(&flag, 4);
  (void)hipMalloc(&sink, 4096 * 256 * 4);
  (void)hipMemset(big, 0, nbig * 16);
  const int steps = 3104;
  for (int streamers : {0, 2048})
    for (int wr : {0, 1}) {
      if (streamers == 0 && wr) continue;
      for (int every : {0, 8, 16, 32}) {
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipMemset(flag, 0, 4);
          switch (every) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(1 + streamers), dim3(256), 0, 0, tab, big, nbig, sink, steps, cyc, flag, wr); break;
            case 8: hipLaunchKernelGGL(k<8>, dim3(1 + streamers), dim3(256), 0, 0, tab, big, nbig, sink, steps, cyc, flag, wr); break;
            case 16: hipLaunchKernelGGL(k<16>, dim3(1 + streamers), dim3(256), 0, 0, tab, big, nbig, sink, steps, cyc, flag, wr); break;
            case 32: hipLaunchKernelGGL(k<32>, dim3(1 + streamers), dim3(256), 0, 0, tab, big, nbig, sink, steps, cyc, flag, wr); break;
          }
          (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%4d streaming workgroups (%s), table store every %2d steps: %6.2f ns / step  (%5.1f us per block)\n", streamers,
               wr ? "read+write" : "read only ", every, (double)h * 10.0 / steps, (double)h * 10.0 / 1000.0);
      }
    }
  return 0;
}
