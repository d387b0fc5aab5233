// tools/ubench_smem.hip -- scalar (SMEM) load throughput on gfx950: how many bytes per clock per CU can
// s_load_dwordx16 deliver when hitting the scalar cache?  (design input: taps travel through this path)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NLOADS>
__global__ __launch_bounds__(256) void k(const float *p, float *out, int iters, int spread) {
  const float *q = p + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % spread) * 1024;
  const float *q0 = (const float *)__builtin_amdgcn_readfirstlane((unsigned long long)q & 0xffffffffu) ; (void)q0;
  unsigned long long qa = (unsigned long long)q;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qa), hi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
  unsigned long long base = ((unsigned long long)hi << 32) | lo;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    f16v a, b, c, d;
    unsigned long long addr = base + (unsigned long long)((it & 15) * 256);
    if (NLOADS == 4) {
      asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\t"
                   "s_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(addr));
      acc += a[0] + b[1] + c[2] + d[3];
    } else {
      asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(addr));
      acc += a[0];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  float *p, *out;
  (void)hipMalloc(&p, 64 << 20);
  (void)hipMemset(p, 0, 64 << 20);
  (void)hipMalloc(&out, 16 << 20);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int iters = 20000;
  for (int nl = 4; nl >= 1; nl -= 3)
    for (int spread : {1, 4096})
      for (int bpc = 1; bpc <= 8; bpc *= 2) {
        const int blocks = 256 * bpc;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(a);
          if (nl == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, p, out, iters, spread);
          else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, p, out, iters, spread);
          (void)hipEventRecord(b);
          (void)hipEventSynchronize(b);
          (void)hipEventElapsedTime(&ms, a, b);
        }
        const double bytes = (double)blocks * 4 * iters * nl * 64;
        printf("loads/iter=%d spread=%4d waves/SIMD=%d: %.3f ms  %.2f TB/s  = %.2f B/clk/CU @2.4GHz, %.0f ns/iter/wave\n", nl, spread,
               bpc, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e6 / iters);
      }
  return 0;
}
