// tools/ubench_chain3.hip -- how much does a dependent-chain wave (the NCO recurrence, priority 3) slow down when other
// waves share its SIMD?  One workgroup of 64*NW threads: wave 0 runs the chain, waves 4, 8, ... (same SIMD as wave 0 when
// waves are dealt round-robin over the 4 SIMDs) run a "hog" at priority 0: packed-FMA stream, LDS broadcast reads + FMAs,
// or global loads + FMAs.  Prints ns per chain step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain3.hip -o sdr-server_amd/build/ubench_chain3
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

// HOG: 0 none, 1 pk_fma stream (16 independent accumulators), 2 LDS broadcast b128 + 8 FMAs, 3 global load + 8 FMAs
template <int HOG>
__global__ __launch_bounds__(1024) void k(float *tab, const v4f *gsrc, int steps, int prio_chain, long long *cyc, volatile int *stopflag, int idle_chain) {
  __shared__ v4f lds[512];
  __shared__ unsigned simd_of_chain, nhogs;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = (v4f){1e-3f * i, 1.f, 0.5f, 0.25f};
  const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
  const unsigned simd = (hwid >> 4) & 3u;                                               // SIMD_ID bits 5:4
  if (threadIdx.x == 0) { simd_of_chain = simd; nhogs = 0; }
  __syncthreads();
  if (w != 0 && l == 0 && simd == simd_of_chain) atomicAdd(&nhogs, 1u);
  __syncthreads();
  if (w == 0 && l == 0) cyc[1] = nhogs;
  if (w == 0) {
    if (prio_chain == 3) __builtin_amdgcn_s_setprio(3);
    v2f p = {1.0f, 1e-3f * l}, q = {0.9999f, 0.01f};
    v4f *o4 = (v4f *)(tab + (size_t)l * 6400);
    const long long t0 = wall_clock64();
    if (idle_chain) {  // no VALU work: just let the same wall time pass
      while (wall_clock64() - t0 < (long long)idle_chain) __builtin_amdgcn_s_sleep(8);
    } else
    for (int m = 0; m + 16 <= steps; m += 16) {
      v2f s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] = p; p = nxt(nxt(nxt(nxt(p, q), q), q), q); }
      o4[m >> 3] = (v4f){s[0].x, s[0].y, s[1].x, s[1].y};
      o4[(m >> 3) + 1] = (v4f){s[2].x, s[2].y, s[3].x, s[3].y};
    }
    const long long t1 = wall_clock64();
    if (l == 0) { cyc[0] = t1 - t0; *stopflag = 1; }
    tab[l] += p.x;
    return;
  }
  if (simd != simd_of_chain || HOG == 0) return;  // only waves on wave 0's SIMD hog
  v2f acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v2f){(float)l, 1.f};
  v2f x = {1.0001f, 0.5f};
  long long n = 0;
  while (*stopflag == 0 && n < 4000000) {
    if (HOG == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(x));
    } else if (HOG == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const v4f v = lds[(n * 8 + r) & 511];
        const v2f a = {v.x, v.y}, b = {v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * i]) : "v"(x), "v"(a));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * i + 1]) : "v"(x), "v"(b));
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const v4f v = gsrc[((n * 8 + r) * 64 + l) & 0xFFFFF];
        const v2f a = {v.x, v.y}, b = {v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * i]) : "v"(x), "v"(a));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[2 * i + 1]) : "v"(x), "v"(b));
        }
      }
    }
    ++n;
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y;
  tab[64 * 6400 + threadIdx.x] = r;
  if (l == 0) atomicAdd((unsigned long long *)&cyc[2], (unsigned long long)n);
}

int main() {
  float *tab; v4f *g; long long *cyc, h[3]; int *flag;
  (void)hipMalloc(&tab, 64 * 6400 * 4 * 4 + 8192); (void)hipMalloc(&g, (1 << 20) * 16); (void)hipMalloc(&cyc, 32); (void)hipMalloc(&flag, 4);
  (void)hipMemset(g, 0, (1 << 20) * 16);
  const int steps = 3120;
  const char *names[] = {"alone", "pk_fma stream", "LDS broadcast + FMA", "global load + FMA"};
  for (int prio : {3, 0})
    for (int nw : {8, 12, 16})
      for (int hog = 1; hog < 4; ++hog) {
        long long chain_ticks = 0, hog_with = 0, hog_alone = 0, nh = 0;
        for (int idle = 0; idle < 2; ++idle) {
          for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(flag, 0, 4);
            (void)hipMemset(cyc, 0, 32);
            const int idle_ticks = idle ? (int)chain_ticks : 0;
            switch (hog) {
              case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * nw), 0, 0, tab, g, steps, prio, cyc, flag, idle_ticks); break;
              case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * nw), 0, 0, tab, g, steps, prio, cyc, flag, idle_ticks); break;
              case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * nw), 0, 0, tab, g, steps, prio, cyc, flag, idle_ticks); break;
            }
            (void)hipDeviceSynchronize();
          }
          (void)hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
          if (!idle) { chain_ticks = h[0]; hog_with = h[2]; nh = h[1]; } else hog_alone = h[2];
        }
        printf("chain prio %d, %lld hog wave(s) on its SIMD, hog = %-22s chain %6.2f ns/step; hog iterations next to the chain / next to an idle wave: %lld / %lld = %.2f\n",
               prio, nh, names[hog], (double)chain_ticks * 10.0 / steps, hog_with, hog_alone, hog_alone ? (double)hog_with / (double)hog_alone : 0.0);
      }
  return 0;
}
