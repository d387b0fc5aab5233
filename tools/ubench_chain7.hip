// tools/ubench_chain7.hip -- what slows xl_nco_chain_kernel (side-stream NCO chain, one wave per SIMD, LDS ring)?
// Runs the product kernel for a call of 8 server-default blocks (24966 steps per client)
//   (a) alone on an idle chip,  (b) next to a VALU-heavy kernel,  (c) next to a memory-streaming kernel,
// for 64 / 128 / 1024 clients.  Reports us per launch and ns per step.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I sdr-server_amd/csrc tools/ubench_chain7.hip -o sdr-server_amd/build/ubench_chain7
#include "../sdr-server_amd/csrc/xl_kernels.hip"

#include <stdio.h>
#include <vector>

__global__ void hog_valu(float *out, int iters) {
  v2f a = {1.0f + threadIdx.x * 1e-6f, 0.5f}, b = {0.999f, 0.001f}, c0 = {0, 0}, c1 = {0, 0}, c2 = {0, 0}, c3 = {0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_elementwise_fma(a, b, c0);
      c1 = __builtin_elementwise_fma(a, b, c1);
      c2 = __builtin_elementwise_fma(a, b, c2);
      c3 = __builtin_elementwise_fma(a, b, c3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x + c1.y + c2.x + c3.y;
}

__global__ void hog_mem(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const uint32_t S = 131072, G = 8, D = 42;
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  float *hout;
  (void)hipMalloc(&hout, 4u << 20);
  const size_t nvec = (size_t)(512u << 20) / 16;
  float4 *msrc, *mdst;
  (void)hipMalloc(&msrc, nvec * 16);
  (void)hipMalloc(&mdst, nvec * 16);
  (void)hipMemset(msrc, 0, nvec * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (uint32_t n : {64u, 128u, 1024u}) {
    std::vector<XlNcoClient> cl(n);
    const uint32_t rows = (G * (S / D + 1) + 31) / 32 * 32;
    for (uint32_t i = 0; i < n; ++i) {
      cl[i].incr = make_float2(0.99995f, 0.01f);
      cl[i].out_off = i * rows;
      cl[i].slot = i;
      cl[i].D = D;
      cl[i].rem0 = 0;
    }
    XlNcoClient *dcl;
    float2 *st0, *st1, *tab;
    (void)hipMalloc(&dcl, n * sizeof(XlNcoClient));
    (void)hipMemcpy(dcl, cl.data(), n * sizeof(XlNcoClient), hipMemcpyHostToDevice);
    (void)hipMalloc(&st0, n * 8);
    (void)hipMalloc(&st1, n * 8);
    (void)hipMalloc(&tab, ((size_t)n * rows / 16 + 64) * 8);
    std::vector<float2> one(n, make_float2(1.0f, 0.0f));
    (void)hipMemcpy(st0, one.data(), n * 8, hipMemcpyHostToDevice);
    XlPos pos = {0, S, G, 0};
    const double steps = (double)((S * G + D - 1) / D);
    for (int mode = 0; mode < 3; ++mode) {
      double best = 1e30;
      for (int rep = 0; rep < 4; ++rep) {
        // the chain first (it needs whole CUs: behind a launch that fills the chip it would wait for that launch to end)
        (void)hipEventRecord(e0, s1);
        XlChainCalls cc = {{tab}, {st1}, 1u};
        (void)xl_launch_nco_chain(dcl, n, st0, cc, pos, nullptr, s1, nullptr);
        (void)hipEventRecord(e1, s1);
        if (mode == 1) hipLaunchKernelGGL(hog_valu, dim3(256 * 16), dim3(256), 0, s2, hout, 3000);
        if (mode == 2) hipLaunchKernelGGL(hog_mem, dim3(256 * 8), dim3(256), 0, s2, msrc, mdst, nvec, 2);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("clients %4u  %-28s %8.1f us per call  %6.2f ns per step\n", n,
             mode == 0 ? "alone" : (mode == 1 ? "next to a VALU-heavy kernel" : "next to a memory stream"), best * 1e3, best * 1e6 / steps);
    }
    (void)hipFree(dcl);
    (void)hipFree(st0);
    (void)hipFree(st1);
    (void)hipFree(tab);
  }
  return 0;
}
