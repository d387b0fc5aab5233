// tools/ubench_chain2.hip -- what do the NCO table stores cost the recurrence chain?  One wave alone, L active lanes,
// each lane writes its own row (row stride 25 KB), chain = the 3-op NCO step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain2.hip -o sdr-server_amd/build/ubench_chain2
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

// MODE 0: no stores; 1: dwordx4 per 2 steps (row per lane); 2: dwordx2 per step; 3: dwordx4 per 2 steps but all lanes to ONE row region
// (coalesced, wrong layout); 4: via LDS transpose, 64-lane coalesced stores of 16 rows x 64 B
template <int MODE>
__global__ __launch_bounds__(64) void k(float *tab, int steps, int lanes, long long *cyc) {
  __shared__ v2f lds[16 * 17];
  const int l = threadIdx.x;
  v2f p = {1.0f, 1e-3f * l}, q = {0.9999f, 0.01f};
  const bool act = l < lanes;
  const long long t0 = wall_clock64();
  if (MODE == 4) {
    v4f *o = (v4f *)tab;
    for (int m = 0; m + 16 <= steps; m += 16) {
      if (act) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { lds[l * 17 + j] = p; p = nxt(p, q); }
      }
      __builtin_amdgcn_wave_barrier();
      // lane l: client l / 4, phases 4 (l % 4) .. +3
      const int c = l >> 2, f = (l & 3) * 4;
      if (c < lanes) {
        v2f a0 = lds[c * 17 + f], a1 = lds[c * 17 + f + 1], a2 = lds[c * 17 + f + 2], a3 = lds[c * 17 + f + 3];
        v4f *row = (v4f *)(tab + (size_t)c * 6400) + ((m + f) >> 1);
        row[0] = (v4f){a0.x, a0.y, a1.x, a1.y};
        row[1] = (v4f){a2.x, a2.y, a3.x, a3.y};
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else if (act) {
    v4f *o4 = (v4f *)(tab + (size_t)(MODE == 3 ? 0 : l) * 6400) + (MODE == 3 ? l : 0);
    v2f *o2 = (v2f *)(tab + (size_t)l * 6400);
    for (int m = 0; m + 16 <= steps; m += 16) {
      v2f s[16];
      s[0] = p;
#pragma unroll
      for (int j = 1; j < 16; ++j) s[j] = nxt(s[j - 1], q);
      p = nxt(s[15], q);
      if (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o4[(MODE == 3 ? 64 : 1) * ((m >> 1) + j)] = (v4f){s[2 * j].x, s[2 * j].y, s[2 * j + 1].x, s[2 * j + 1].y};
      }
      if (MODE == 5) {
        o4[(m >> 3)] = (v4f){s[0].x, s[0].y, s[4].x, s[4].y};
        o4[(m >> 3) + 1] = (v4f){s[8].x, s[8].y, s[12].x, s[12].y};
      }
      if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o2[m + j] = s[j];
      }
    }
  }
  const long long t1 = wall_clock64();
  if (l == 0) cyc[0] = t1 - t0;
  if (MODE == 0 && act) tab[l] = p.x + p.y;
}

int main() {
  float *tab; long long *cyc, h;
  (void)hipMalloc(&tab, 64 * 6400 * 4 * 64); (void)hipMalloc(&cyc, 16);
  const int steps = 3120;
  const char *names[] = {"no stores", "dwordx4 / 2 steps, row per lane", "dwordx2 / step, row per lane", "dwordx4 / 2 steps, coalesced (wrong layout)", "LDS transpose, 64-lane stores", "every 4th phase: dwordx4 / 8 steps, row per lane"};
  for (int lanes : {16, 64})
    for (int mode = 0; mode < 6; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, tab, steps, lanes, cyc); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, tab, steps, lanes, cyc); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, tab, steps, lanes, cyc); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, tab, steps, lanes, cyc); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, tab, steps, lanes, cyc); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, tab, steps, lanes, cyc); break;
        }
        (void)hipDeviceSynchronize();
      }
      (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("lanes %2d  %-46s %6.2f ns / step   (%5.1f us per 3120-step block)\n", lanes, names[mode], (double)h * 10.0 / steps, (double)h * 10.0 / 1000.0);
    }
  return 0;
}
