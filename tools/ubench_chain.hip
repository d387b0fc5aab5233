// tools/ubench_chain.hip -- latency of DEPENDENT VALU chains on gfx950, one wave alone (design input for the NCO
// phase recurrence, which is one dependent chain of 3120 steps per client and block).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o sdr-server_amd/build/ubench_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc) {
  v2f p = {1.0f + threadIdx.x * 1e-7f, 0.5f};
  v2f q = {0.999f, 0.01f};
  v2f t1 = {0, 0}, t2 = {0, 0};
  float a = 1.0f + threadIdx.x * 1e-7f, b = 0.9999f, c = 1e-8f;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
    if (KIND == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
    if (KIND == 2) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 3) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
    if (KIND == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 5) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));) }
    if (KIND == 6) {  // the NCO step as written in xl_dev_inline.h (3 packed ops)
      REP16(asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
                         : "+v"(p), "+v"(t1), "+v"(t2) : "v"(q));)
    }
    if (KIND == 7) {  // the NCO step with scalar ops: 4 mul + sub + add
      float pr = p.x, pi = p.y, m0, m1, m2, m3;
      REP16(asm volatile("v_mul_f32 %2, %0, %6\n\tv_mul_f32 %3, %1, %7\n\tv_mul_f32 %4, %0, %7\n\tv_mul_f32 %5, %1, %6\n\t"
                         "v_sub_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5"
                         : "+v"(pr), "+v"(pi), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3) : "v"(q.x), "v"(q.y));)
      p.x = pr; p.y = pi;
    }
    if (KIND == 8) {  // two independent NCO chains interleaved (ILP)
      REP16(asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
                         : "+v"(p), "+v"(t1), "+v"(t2) : "v"(q));
            asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
                         : "+v"(*(v2f *)&a), "+v"(t1), "+v"(t2) : "v"(q));)
    }
    if (KIND == 9) {  // NCO step, mul pair written as mul + fma-free: pk_mul, then pk_fma?  (NOT bit-exact; latency probe)
      REP16(asm volatile("v_pk_mul_f32 %1, %0, %2 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_fma_f32 %0, %0, %2, %1 op_sel_hi:[1,0,1]"
                         : "+v"(p), "+v"(t1) : "v"(q));)
    }
  }
  const long long c1 = clock64();
  const long long t1w = wall_clock64();
  out[threadIdx.x] = p.x + p.y + a + t1.x + t2.x;
  if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1w - t0; }
}

int main() {
  float *out; long long *cyc, h[2];
  (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 16);
  const char *names[] = {"v_pk_mul_f32 dep", "v_pk_add_f32 dep", "v_mul_f32 dep", "v_add_f32 dep", "v_fma_f32 dep", "v_pk_fma_f32 dep",
                         "NCO step (3 pk ops)", "NCO step (6 scalar ops)", "2 NCO chains interleaved (per pair)", "mul+fma step (2 pk ops)"};
  const int iters = 20000;
  for (int kind = 0; kind < 10; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 9: hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
      }
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-38s %7.2f shader cycles / unit   %7.2f ns / unit (100 MHz wall clock)\n", names[kind], (double)h[0] / (iters * 16.0),
           (double)h[1] * 10.0 / (iters * 16.0));
  }
  return 0;
}
