// tools/ubench_cumask.hip -- how hipExtStreamCreateWithCUMask maps mask bits to (XCC, SE, CU) on MI355X, and whether a
// kernel on a masked stream stays off the CUs of a complementary mask (CU reservation for the NCO chain kernel).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_cumask.hip -o sdr-server_amd/build/ubench_cumask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <set>
#include <vector>

__global__ void where(unsigned *out, int spin) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

static void run(const char *name, hipStream_t s, unsigned *d, int nblk) {
  std::vector<unsigned> h(2 * nblk);
  hipLaunchKernelGGL(where, dim3(nblk), dim3(64), 0, s, d, 200);
  (void)hipStreamSynchronize(s);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;  // xcc -> set of (se, cu)
  for (int i = 0; i < nblk; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  size_t tot = 0;
  printf("%-26s", name);
  for (auto &kv : per) {
    printf(" xcc%u:%zu", kv.first, kv.second.size());
    tot += kv.second.size();
  }
  printf("  = %zu distinct CUs\n", tot);
  if (tot <= 40) {
    for (auto &kv : per) {
      printf("    xcc%u:", kv.first);
      for (unsigned v : kv.second) printf(" se%u.sh%u.cu%u", v >> 8, (v >> 4) & 1, v & 0xF);
      printf("\n");
    }
  }
}

int main() {
  unsigned *d;
  const int nblk = 16384;
  (void)hipMalloc(&d, 2 * nblk * 4);
  hipStream_t s0;
  (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  run("no mask", s0, d, nblk);
  for (int test = 0; test < 5; ++test) {
    unsigned mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = 0;
    const char *name = "";
    if (test == 0) { name = "bits 0..15"; mask[0] = 0xFFFF; }
    if (test == 1) { name = "bits 0..31"; mask[0] = 0xFFFFFFFF; }
    if (test == 2) { name = "bits 32..47"; mask[1] = 0xFFFF; }
    if (test == 3) { name = "all but bits 0..15"; for (int i = 0; i < 8; ++i) mask[i] = 0xFFFFFFFF; mask[0] = 0xFFFF0000; }
    if (test == 4) { name = "every 16th bit"; for (int i = 0; i < 8; ++i) mask[i] = 0x00010001; }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) {
      printf("%-26s hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e));
      continue;
    }
    run(name, s, d, nblk);
    (void)hipStreamDestroy(s);
  }
  return 0;
}
