// tools/ubench_chain8.hip -- the NCO recurrence step (xlating.c:71) with FOUR lanes per client: each lane of a quad owns one of
// the four products, the partner values arrive through DPP inside the instruction itself, so a step is TWO dependent VALU
// instructions (v_mul_f32_dpp -> v_subrev_f32_dpp) instead of three.  Same IEEE operations, same roundings: the quad holds
// [re, -re, im, -im]; products [re*c, im*d, re*d, (-im)*c] = [ac, bd, ad, -bc]; own - neighbour = [re', -re', im', -im'].
// Variants probe how many wait states the DPP read of a just-written VGPR needs (results are checked against the packed step).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain8.hip -o sdr-server_amd/build/ubench_chain8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X

#define QSTEP(NOPA, NOPB)                                                                      \
  "v_mul_f32_dpp %1, %0, %2 quad_perm:[0,2,0,3] row_mask:0xf bank_mask:0xf\n\t" NOPA            \
  "v_subrev_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" NOPB

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc) {
  const uint32_t l4 = threadIdx.x & 3u, client = threadIdx.x >> 2;
  const float ang = 0.001f + 0.0137f * (float)client;
  const v2f q = {cosf(ang), sinf(ang)};
  v2f p = {1.0f, 0.0f};
  v2f t1 = {0, 0}, t2 = {0, 0};
  float v = l4 == 0 ? 1.0f : l4 == 1 ? -1.0f : 0.0f;  // [re, -re, im, -im]
  const float kk = (l4 == 0 || l4 == 3) ? q.x : q.y;  // [c, d, d, c]
  float a;
  const long long c0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      REP16(asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
                         : "+v"(p), "+v"(t1), "+v"(t2) : "v"(q));)
    }
    if (KIND == 1) { REP16(asm volatile(QSTEP("", "") : "+v"(v), "=&v"(a) : "v"(kk));) }
    if (KIND == 2) { REP16(asm volatile(QSTEP("s_nop 0\n\t", "s_nop 0\n\t") : "+v"(v), "=&v"(a) : "v"(kk));) }
    if (KIND == 3) { REP16(asm volatile(QSTEP("s_nop 1\n\t", "s_nop 1\n\t") : "+v"(v), "=&v"(a) : "v"(kk));) }
    if (KIND == 4) { REP16(asm volatile(QSTEP("s_nop 3\n\t", "s_nop 3\n\t") : "+v"(v), "=&v"(a) : "v"(kk));) }
  }
  const long long c1c = clock64();
  const long long w1 = wall_clock64();
  if (KIND == 0) { out[2 * threadIdx.x] = p.x; out[2 * threadIdx.x + 1] = p.y; }
  else out[threadIdx.x] = v;
  if (threadIdx.x == 0) { cyc[0] = c1c - c0; cyc[1] = w1 - w0; }
}

int main() {
  float *out; long long *cyc, h[2];
  (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 16);
  const char *names[] = {"packed step, 3 ops (as shipped), 64 clients / wave", "quad step, 2 DPP ops, no nops", "quad step, s_nop 0 after each",
                         "quad step, s_nop 1 after each", "quad step, s_nop 3 after each"};
  const int iters = 2000;
  float ref[128], got[64];
  for (int kind = 0; kind < 5; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, iters, cyc); break;
      }
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    if (kind == 0) (void)hipMemcpy(ref, out, 512, hipMemcpyDeviceToHost);
    else {
      (void)hipMemcpy(got, out, 256, hipMemcpyDeviceToHost);
      // packed lane c*4 (client index = lane >> 2 in both kernels -> compare with packed lanes 4c..4c+3, all the same client)
      for (int c = 0; c < 16; ++c) {
        const float re = ref[2 * (4 * c)], im = ref[2 * (4 * c) + 1];
        const float want[4] = {re, -re, im, -im};
        for (int j = 0; j < 4; ++j) bad += memcmp(&want[j], &got[4 * c + j], 4) != 0;
      }
    }
    printf("%-52s %7.2f shader cycles / step   %7.2f ns / step   %s\n", names[kind], (double)h[0] / (iters * 16.0),
           (double)h[1] * 10.0 / (iters * 16.0), kind == 0 ? "(reference)" : bad ? "MISMATCH vs packed" : "bit-identical to packed");
  }
  return 0;
}
