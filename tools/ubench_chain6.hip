// tools/ubench_chain6.hip -- can the NCO recurrence step (xlating.c:71; 3 packed ops, ~17 cycles) be made shorter
// with two lanes per client (one holds re, one im; the partner's value arrives through DPP inside the multiply)?
// Per step: v_mul_f32 own*c1, v_mul_f32_dpp partner*c2, v_add_f32 -- the same IEEE operations, same roundings.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain6.hip -o sdr-server_amd/build/ubench_chain6
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc) {
  const bool odd = threadIdx.x & 1;
  v2f p = {1.0f, 0.0f};
  const v2f q = {0.99995f, 0.01f};
  v2f t1 = {0, 0}, t2 = {0, 0};
  // lane pair: even lane holds re, odd lane holds im.  new_re = re*ir + im*(-ii);  new_im = im*ir + re*ii
  float v = odd ? 0.0f : 1.0f;
  const float c1 = q.x, c2 = odd ? q.y : -q.y;
  float a, b;
  const long long c0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {  // reference: the packed step
      REP16(asm volatile("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
                         : "+v"(p), "+v"(t1), "+v"(t2) : "v"(q));)
    }
    if (KIND == 1) {  // lane pair with DPP quad_perm [1,0,3,2] on the second multiply
      REP16(asm volatile("v_mul_f32 %1, %0, %3\n\t"
                         "v_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_add_f32 %0, %1, %2"
                         : "+v"(v), "=&v"(a), "=&v"(b) : "v"(c1), "v"(c2));)
    }
    if (KIND == 2) {  // same, add first operand order swapped and no nop (assembler inserts required waits?)
      REP16(asm volatile("v_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mul_f32 %1, %0, %3\n\t"
                         "v_add_f32 %0, %1, %2"
                         : "+v"(v), "=&v"(a), "=&v"(b) : "v"(c1), "v"(c2));)
    }
    if (KIND == 3) {  // plain dependent v_mul_f32 -> v_add_f32 pair (latency probe)
      REP16(asm volatile("v_mul_f32 %1, %0, %2\n\tv_add_f32 %0, %1, %3" : "+v"(v), "=&v"(a) : "v"(c1), "v"(c2));)
    }
    if (KIND == 4) {  // packed dependent pk_mul -> pk_add pair (latency probe)
      REP16(asm volatile("v_pk_mul_f32 %1, %0, %2\n\tv_pk_add_f32 %0, %1, %2" : "+v"(p), "=&v"(t1) : "v"(q));)
    }
  }
  const long long c1c = clock64();
  const long long w1 = wall_clock64();
  out[threadIdx.x] = p.x + p.y + v + t1.x + t2.x;
  if (threadIdx.x == 0) { cyc[0] = c1c - c0; cyc[1] = w1 - w0; }
}

int main() {
  float *out; long long *cyc, h[2];
  (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 16);
  const char *names[] = {"NCO step, 3 packed ops (as shipped)", "NCO step, lane pair + DPP (mul, mul_dpp, nop, add)",
                         "NCO step, lane pair + DPP (mul_dpp, mul, add)", "v_mul_f32 -> v_add_f32 dependent pair", "v_pk_mul_f32 -> v_pk_add_f32 dependent pair"};
  const int iters = 20000;
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
    printf("%-58s %7.2f shader cycles / step   %7.2f ns / step\n", names[kind], (double)h[0] / (iters * 16.0), (double)h[1] * 10.0 / (iters * 16.0));
  }
  return 0;
}
