// tools/ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (design input for xl_fir_kernel):
// how fast do v_fma_f32 / v_pk_fma_f32 issue with an SGPR operand, at 1..8 waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o sdr-server_amd/build/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ __launch_bounds__(256) void k(const float *__restrict__ coef, float *out, int iters) {
  typedef const float __attribute__((address_space(4))) *cp;
  cp c = (cp)(uintptr_t)coef;
  v2f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v2f){(float)threadIdx.x, 1.0f};
  v2f x = {1.0f + threadIdx.x * 1e-9f, 0.5f};
  for (int it = 0; it < iters; ++it) {
    const float s0 = c[(it & 7) * 2], s1 = c[(it & 7) * 2 + 1];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PK) {
          v2f h = {s0, s1};
          acc[i] = __builtin_elementwise_fma(x, h, acc[i]);
        } else {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "s"(s0));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(x.y), "s"(s1));
        }
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float *coef, *out;
  (void)hipMalloc(&coef, 64);
  (void)hipMalloc(&out, 4 << 20);
  float h[16];
  for (int i = 0; i < 16; ++i) h[i] = 1e-7f * i;
  (void)hipMemcpy(coef, h, 64, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk)
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
      const int blocks = 256 * bpc;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, coef, out, iters);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, coef, out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        const double flops = (double)blocks * 256 * iters * 8 * 8 * 2 * 2;
        if (rep) printf("pk=%d blocks/CU=%d (waves/SIMD=%d): %.3f ms  %.1f TFLOP/s\n", pk, bpc, bpc, ms, flops / ms / 1e9);
      }
    }
  return 0;
}
