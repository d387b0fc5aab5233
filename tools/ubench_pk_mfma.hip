// tools/ubench_pk_mfma.hip -- a stand-alone reproducer for DESIGN_HISTORY.md 3.6: does a wave that steps a complex float32 recurrence
// with PACKED instructions (v_pk_mul_f32 x2, v_pk_add_f32: the NCO role's step, xl_dev_inline.h) compute different bits when
// waves issuing MATRIX instructions are resident on the chip at the same time?
//   chain kernel    256 workgroups of one wave; lane l of workgroup g rotates p <- p * inc_(g,l) for `steps` steps and stores p.
//                   Variants: packed step (inline asm, as shipped until round 4), scalar step (six instructions), and the packed
//                   step with a 16-byte store every 32 steps (the role's table stores).
//   neighbours      none | a kernel spinning on v_mfma_f32_32x32x16_f16 on another stream | one spinning on v_pk_fma_f32 | one shaped
//                   like xlp_mix_mfma_kernel (LDS staging behind a barrier, three matrix instructions, non-temporal stores).
// Every (variant, neighbour) pair is run `reps` times and compared bit for bit with the variant's result WITHOUT a neighbour;
// mismatching lanes are histogrammed by lane index.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_pk_mfma.hip -o sdr-server_amd/build/ubench_pk_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v2f step_packed(const v2f p, const v2f inc) {
  v2f t1, t2, r;
  asm volatile(
      "v_pk_mul_f32 %0, %3, %4 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %1, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_add_f32 %2, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
      : "=&v"(t1), "=&v"(t2), "=&v"(r)
      : "v"(p), "v"(inc));
  return r;
}
__device__ __forceinline__ v2f step_scalar(const v2f p, const v2f inc) {
  float a, b, c, d;
  asm volatile(
      "v_mul_f32 %0, %4, %6\n\tv_mul_f32 %1, %5, %7\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %5, %6\n\t"
      "v_sub_f32 %0, %0, %1\n\tv_add_f32 %1, %3, %2"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p.x), "v"(p.y), "v"(inc.x), "v"(inc.y));
  return (v2f){a, b};
}

template <int KIND>  // 0 packed, 1 scalar, 2 packed + a 16-byte store every 32 steps
__global__ __launch_bounds__(64) void chain(v2f *__restrict__ out, v4f *__restrict__ tab, const int steps) {
  const unsigned id = blockIdx.x * 64u + threadIdx.x;
  const float ang = 1e-3f + 1e-6f * (float)id;
  const v2f inc = {__cosf(ang), __sinf(ang)};
  v2f p = {1.0f, 0.0f};
  for (int s = 0; s < steps; s += 32) {
    const v2f q0 = p;
#pragma unroll
    for (int j = 0; j < 32; ++j) p = KIND == 1 ? step_scalar(p, inc) : step_packed(p, inc);
    if (KIND == 2) tab[(size_t)id * 64u + ((unsigned)(s >> 5) & 63u)] = (v4f){q0.x, q0.y, p.x, p.y};
    if (((s >> 5) & 1023) == 1023) {  // renormalise now and then so that the values stay O(1) (exact ops: same bits every run)
      const float m = __fsqrt_rn(p.x * p.x + p.y * p.y);
      p = (v2f){p.x / m, p.y / m};
    }
  }
  out[id] = p;
}

__global__ __launch_bounds__(256) void spin_mfma(float *sink, const int iters) {
  v8h a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (float)(threadIdx.x + i)), b[i] = (_Float16)(0.002f * (float)(i + 1));
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 1.2345f) sink[0] = s;
}
// closer to xlp_mix_mfma_kernel: operands staged through LDS behind a barrier, three matrix instructions, a non-temporal store
__global__ __launch_bounds__(256) void spin_mfma_lds(float *sink, float *dump, const int iters) {
  __shared__ uint4 xs[2][256];
  v8h b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.002f * (float)(i + 1));
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const unsigned t = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    xs[it & 1][t] = make_uint4(0x3C003C00u + (unsigned)it, 0x3C003800u, 0x38003C00u, 0x3C003C00u ^ t);
    __syncthreads();
    const v8h a = __builtin_bit_cast(v8h, xs[it & 1][(t + 64u) & 255u]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    if ((it & 7) == 7) {
      __builtin_nontemporal_store(acc[it & 15], &dump[((size_t)blockIdx.x * 256u + t) + (size_t)(it & 1023) * 524288u]);
      for (int i = 0; i < 16; ++i) acc[i] *= 1e-3f;
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(256) void spin_pkfma(float *sink, const int iters) {
  v2f acc[8], x = {1.0001f, 0.9999f}, y = {1e-7f * (float)threadIdx.x, 1e-7f};
  for (int i = 0; i < 8; ++i) acc[i] = (v2f){(float)i, 0.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(x), "v"(y));
  }
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (s == 1.2345f) sink[0] = s;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 1 << 20, reps = argc > 2 ? atoi(argv[2]) : 12, NWG = 256, N = NWG * 64;
  v2f *out; v4f *tab; float *sink;
  CK(hipMalloc(&out, N * sizeof(v2f))); CK(hipMalloc(&tab, (size_t)N * 64 * sizeof(v4f))); CK(hipMalloc(&sink, 64));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const char *kinds[3] = {"packed step (v_pk_mul x2, v_pk_add)", "scalar step (v_mul x4, v_sub, v_add)", "packed step + table stores"};
  const char *nbs[4] = {"alone", "beside v_mfma_f32_32x32x16_f16 waves", "beside v_pk_fma_f32 waves", "beside mfma + LDS + barrier + nt-store waves"};
  float *dump; CK(hipMalloc(&dump, (size_t)1024 * 524288 * sizeof(float)));
  std::vector<v2f> ref(N), got(N);
  for (int kind = 0; kind < 3; ++kind) {
    for (int nb = 0; nb < 4; ++nb) {
      long bad_runs = 0, bad_lanes = 0; int hist[4] = {0, 0, 0, 0};
      for (int r = 0; r < (nb == 0 ? 3 : reps); ++r) {
        CK(hipMemsetAsync(out, 0, N * sizeof(v2f), s1)); CK(hipStreamSynchronize(s1));
        // the neighbour first (long enough to cover the chain), then the chain on the other stream
        if (nb == 1) hipLaunchKernelGGL(spin_mfma, dim3(2048), dim3(256), 0, s2, sink, steps / 16);
        if (nb == 2) hipLaunchKernelGGL(spin_pkfma, dim3(2048), dim3(256), 0, s2, sink, steps / 2);
        if (nb == 3) hipLaunchKernelGGL(spin_mfma_lds, dim3(2048), dim3(256), 0, s2, sink, dump, steps / 8);
        if (kind == 0) hipLaunchKernelGGL(chain<0>, dim3(NWG), dim3(64), 0, s1, out, tab, steps);
        if (kind == 1) hipLaunchKernelGGL(chain<1>, dim3(NWG), dim3(64), 0, s1, out, tab, steps);
        if (kind == 2) hipLaunchKernelGGL(chain<2>, dim3(NWG), dim3(64), 0, s1, out, tab, steps);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(got.data(), out, N * sizeof(v2f), hipMemcpyDeviceToHost));
        if (nb == 0 && r == 0) { ref = got; continue; }
        long bad = 0;
        for (int i = 0; i < N; ++i)
          if (memcmp(&got[i], &ref[i], sizeof(v2f)) != 0) { ++bad; ++hist[(i & 63) >> 4]; }
        bad_lanes += bad; bad_runs += bad != 0;
      }
      printf("%-40s %-40s: %ld of %d runs differ, %ld lanes in all; by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %d / %d / %d / %d\n",
             kinds[kind], nbs[nb], bad_runs, nb == 0 ? 2 : reps, bad_lanes, hist[0], hist[1], hist[2], hist[3]);
      fflush(stdout);
    }
  }
  return 0;
}
