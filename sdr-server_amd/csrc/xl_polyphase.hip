// xl_polyphase.hip -- polyphase overlap-save evaluation of the frequency-xlating FIR (see xl_polyphase.h for the
// algebra and why it is the same operator as /root/reference/src/xlating.c:52-72).  Hand-written for gfx950.
//
// Three launches per block and class (stream order is the only synchronisation):
//   xlp_forward_kernel  one wave per (segment, branch): raw samples -> cf32 (xlating.c:357-378, exact) -> 256-point
//                       DFT of the branch -> shared spectra X[pass][b][m][s].  D * nseg small transforms: ~1 MB.
//   xlp_mix_kernel      Y[c][s][m] = sum_b X[s][b][m] * R[c][b][m].  lane = client column (256 per workgroup), m is
//                       workgroup-uniform: X comes through SCALAR loads as SGPR operands of v_pk_fma_f32 (13 segments
//                       per row = two s_load_dwordx16), R is streamed exactly once, coalesced (8 bytes per lane, 2 KB
//                       per workgroup and branch).  HBM-bound on R: 8 * D * M bytes per client and block.
//   xlp_inverse_kernel  per (segment, 16 columns): Y tile -> LDS (transposed) -> 256-point inverse DFT per column ->
//                       scale, NCO rotate (xlating.c:70) with the tabulated float32 phase -> out[k], k < K.
// Each launch also carries a slice of the NEXT block's NCO phase recurrence (a ~57 us dependent chain per block
// that would otherwise serialise with these short kernels).
#include "xl_polyphase.h"

#include "xl_dev_inline.h"

XL_DEV v2f xlp_cmul(const v2f a, const v2f b) {
  return (v2f){__builtin_fmaf(-a.y, b.y, a.x * b.x), __builtin_fmaf(a.y, b.x, a.x * b.y)};
}

// 256-point DFT by one wave: radix-4 Stockham autosort, passes p = 1, 4, 16, 64; lane j holds points j + 64 r.
// In: u[r] = x[j + 64 r].  Out: u[r] = X[j + 64 r] (natural order).  SIGN -1 forward, +1 inverse (unnormalised).
// W[n] = e^{-2 pi j n / 256}.  `lds` = 256 complex of scratch owned by this wave; LDS operations of one wave execute
// in order, so no barrier is needed between a pass's scatter and the next gather.
template <int SIGN>
XL_DEV void xlp_dft256(v2f (&u)[4], v2f *__restrict__ lds, const v2f *__restrict__ W, const uint32_t j) {
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t p = 1u << (2 * pass);
    const uint32_t k = j & (p - 1u);
    if (pass > 0) {
      const uint32_t step = 64u >> (2 * pass);  // 256 / (4 p)
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        v2f w = W[(r * k * step) & 255u];
        if (SIGN > 0) w.y = -w.y;
        u[r] = xlp_cmul(u[r], w);
      }
    }
    const v2f v0 = u[0] + u[2], v1 = u[0] - u[2], v2 = u[1] + u[3], t = u[1] - u[3];
    const v2f v3 = SIGN > 0 ? (v2f){-t.y, t.x} : (v2f){t.y, -t.x};  // * (SIGN * j)
    v2f y[4] = {v0 + v2, v1 + v3, v0 - v2, v1 - v3};
    if (pass < 3) {
      const uint32_t jo = ((j - k) << 2) + k;
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[jo + r * p] = y[r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = lds[j + 64u * r];
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = y[r];
    }
  }
}

// NCO role of a launch: the first a.nco_blocks workgroups carry XL_NCO_LANES clients each (first wave only).
XL_DEV void xlp_nco_role(const XlpArgs &a, const XlDynArgs &dyn_next) {
  __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x >= XL_NCO_LANES) return;
  const uint32_t c = blockIdx.x * XL_NCO_LANES + threadIdx.x;
  if (c >= a.nco_nclients) return;
  const XlNcoClient k = a.nco_clients[c];
  const uint32_t K = dyn_next.d[k.cls].K;
  const uint32_t kb = a.nco_k0 == 0u ? 0u : (uint32_t)(((uint64_t)K * a.nco_k0) >> 16) & ~1u;
  const bool final = a.nco_k1 >= 65536u;
  const uint32_t ke = final ? K : (uint32_t)(((uint64_t)K * a.nco_k1) >> 16) & ~1u;
  xl_nco_client_slice(k, K, kb, ke, final, a.nco_state_src, a.nco_state_dst, a.nco_tab);
}

// ------------------------------------------------------------------------------------------- forward transforms
__global__ __launch_bounds__(64) void xlp_forward_kernel(const XlpArgs a, const XlDynArgs dyn,
                                                         const XlDynArgs dyn_next) {
  __shared__ v2f lds[XLP_M];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a, dyn_next);
    return;
  }
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t j = threadIdx.x;
  const uint32_t s = bid / a.D, b = bid - s * a.D;
  const XlDyn d = dyn.d[a.cls];
  // branch sample n of segment s = stream sample base + (s V + n) D + b   (base: first tap of output 0)
  const uint32_t first = d.base + s * a.V * a.D + b;
  const uint32_t end = a.n0 + a.n1;
  v2f u[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t idx = first + (j + 64u * r) * a.D;
    const bool ok = idx >= d.zero_below && idx < end;  // late joiner: zeros below; past the block: zeros (those
                                                       // outputs lie beyond K and are never stored)
    const bool lo = idx < a.n0;
    const void *src = (lo || !ok) ? a.in0 : a.in1;
    const v2f v = xl_sample(src, (int)a.fmt, ok ? (lo ? idx : idx - a.n0) : 0u);
    u[r] = ok ? v : (v2f){0.0f, 0.0f};
  }
  xlp_dft256<-1>(u, lds, reinterpret_cast<const v2f *>(a.W), j);
  const uint32_t pass = s / XLP_SEG, si = s - pass * XLP_SEG;
  v2f *__restrict__ X = reinterpret_cast<v2f *>(a.X);
#pragma unroll
  for (int r = 0; r < 4; ++r) X[(((size_t)pass * a.Dpad + b) * XLP_M + (j + 64u * r)) * XLP_XS + si] = u[r];
}

// ------------------------------------------------------------------------------------------- mix (the hot kernel)
// grid = nco_blocks + M * nsg * passes workgroups of 256 threads; thread t = client column sg * 256 + t.
__global__ __launch_bounds__(256) void xlp_mix_kernel(const XlpArgs a, const XlDynArgs dyn_next) {
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a, dyn_next);
    return;
  }
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t m = bid % XLP_M;
  const uint32_t q = bid / XLP_M;
  const uint32_t sg = q % a.nsg, pass = q / a.nsg;
  const v2f *__restrict__ Rp =
      reinterpret_cast<const v2f *>(a.R) + ((size_t)sg * a.Dpad * XLP_M + m) * XLP_COLS + threadIdx.x;
  const size_t rstride = (size_t)XLP_M * XLP_COLS;
  const cfloat_p Xp = (cfloat_p)(uintptr_t)(a.X + ((size_t)pass * a.Dpad * XLP_M + m) * XLP_XS);
  const size_t xstride = (size_t)XLP_M * XLP_XS * 2;  // floats per branch
  XlAcc<1> acc[XLP_SEG];
#pragma unroll
  for (int i = 0; i < (int)XLP_SEG; ++i) acc[i].clear();
  // branches in stages of XLP_BSTEP: the R rows of the next stage are in flight while this one is multiplied
  v2f r[XLP_BSTEP], rn[XLP_BSTEP];
#pragma unroll
  for (int u = 0; u < (int)XLP_BSTEP; ++u) r[u] = Rp[(size_t)u * rstride];
  for (uint32_t b0 = 0; b0 < a.Dpad; b0 += XLP_BSTEP) {
    // (the last stage prefetches rows past this supergroup's image: the next supergroup's, or the XLP_BSTEP rows of
    // tail padding the engine allocates -- loaded, never used)
#pragma unroll
    for (int u = 0; u < (int)XLP_BSTEP; ++u) rn[u] = Rp[(size_t)(b0 + XLP_BSTEP + u) * rstride];
#pragma unroll
    for (int u = 0; u < (int)XLP_BSTEP; ++u) {
      const cfloat_p x = Xp + (size_t)(b0 + u) * xstride;
#pragma unroll
      for (int i = 0; i < (int)XLP_SEG; ++i) acc[i].mac(r[u], x[2 * i], x[2 * i + 1]);
    }
#pragma unroll
    for (int u = 0; u < (int)XLP_BSTEP; ++u) r[u] = rn[u];
  }
  const uint32_t s0 = pass * XLP_SEG;
  v2f *__restrict__ Yp =
      reinterpret_cast<v2f *>(a.Y) + (((size_t)sg * a.nseg_cap + s0) * XLP_M + m) * XLP_COLS + threadIdx.x;
#pragma unroll
  for (int i = 0; i < (int)XLP_SEG; ++i)
    if (s0 + i < a.nseg) Yp[(size_t)i * XLP_M * XLP_COLS] = acc[i].value();
}

// ------------------------------------------------------------------------------------------- inverse + epilogue
// grid = nco_blocks + nseg * nsg * 16 workgroups of 256 threads; workgroup = (segment, 16 columns).
__global__ __launch_bounds__(256) void xlp_inverse_kernel(const XlpArgs a, const XlDynArgs dyn,
                                                          const XlDynArgs dyn_next) {
  __shared__ v2f tile[16][XLP_M];     // [column][bin]
  __shared__ v2f scratch[4][XLP_M];   // per wave
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a, dyn_next);
    return;
  }
  const uint32_t bid = blockIdx.x - a.nco_blocks;
  const uint32_t sub = bid & 15u;
  const uint32_t q = bid >> 4;
  const uint32_t sg = q % a.nsg, s = q / a.nsg;
  {
    const uint32_t m = threadIdx.x;
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(
        a.Y + (((size_t)sg * a.nseg_cap + s) * XLP_M + m) * XLP_COLS + sub * 16u);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const v4f v = src[i];
      tile[2 * i][m] = (v2f){v.x, v.y};
      tile[2 * i + 1][m] = (v2f){v.z, v.w};
    }
  }
  __syncthreads();
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const uint32_t K = dyn.d[a.cls].K;
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  for (uint32_t ci = 0; ci < 4u; ++ci) {
    const uint32_t i = 4u * w + ci;
    const uint32_t off = a.col_out[sg * XLP_COLS + sub * 16u + i];
    if (off == 0xFFFFFFFFu) continue;
    v2f u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = tile[i][j + 64u * r];
    xlp_dft256<+1>(u, scratch[w], reinterpret_cast<const v2f *>(a.W), j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t qo = j + 64u * r;
      const uint32_t k = s * a.V + qo;
      if (qo < a.V && k < K) {
        const v2f y = u[r] * (1.0f / (float)XLP_M);  // exact scaling by 2^-8
        out[off + k] = xl_rotate<1>(y, ph[off + k]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- branch spectra
// R[sg][b][m][col] = sum_{a<A} r_col[D a + b] e^{+2 pi j a m / M}, in double, rounded once.  One-time per plan.
__global__ __launch_bounds__(256) void xlp_tables_kernel(const float2 *__restrict__ rt, uint32_t ncols, uint32_t T,
                                                         uint32_t D, uint32_t Dpad, uint32_t A, float2 *__restrict__ R) {
  const uint32_t m = blockIdx.x % XLP_M;
  const uint32_t q = blockIdx.x / XLP_M;
  const uint32_t b = q % Dpad, sg = q / Dpad;
  const uint32_t col = sg * XLP_COLS + threadIdx.x;
  double sr = 0.0, si = 0.0;
  if (col < ncols && b < D) {
    const float2 *__restrict__ t = rt + (size_t)col * T;
    for (uint32_t aa = 0; aa < A; ++aa) {
      const uint32_t i = D * aa + b;
      if (i >= T) break;
      double sn, cs;
      sincospi(2.0 * (double)((aa * m) & (XLP_M - 1u)) / (double)XLP_M, &sn, &cs);
      const double tr = t[i].x, ti = t[i].y;
      sr += tr * cs - ti * sn;
      si += tr * sn + ti * cs;
    }
  }
  R[(((size_t)sg * Dpad + b) * XLP_M + m) * XLP_COLS + threadIdx.x] = make_float2((float)sr, (float)si);
}

// ------------------------------------------------------------------------------------------- launchers
hipError_t xlp_launch_tables(const float2 *rt, uint32_t ncols, uint32_t T, uint32_t D, uint32_t Dpad, uint32_t A,
                             uint32_t nsg, float2 *R, hipStream_t s) {
  hipLaunchKernelGGL(xlp_tables_kernel, dim3(XLP_M * Dpad * nsg), dim3(256), 0, s, rt, ncols, T, D, Dpad, A, R);
  return hipGetLastError();
}

hipError_t xlp_launch_forward(const XlpArgs &a, const XlDynArgs &dyn, const XlDynArgs &dyn_next, hipStream_t s) {
  hipLaunchKernelGGL(xlp_forward_kernel, dim3(a.nco_blocks + a.nseg * a.D), dim3(64), 0, s, a, dyn, dyn_next);
  return hipGetLastError();
}

hipError_t xlp_launch_mix(const XlpArgs &a, const XlDynArgs &dyn_next, hipStream_t s) {
  const uint32_t passes = (a.nseg + XLP_SEG - 1) / XLP_SEG;
  hipLaunchKernelGGL(xlp_mix_kernel, dim3(a.nco_blocks + XLP_M * a.nsg * passes), dim3(256), 0, s, a, dyn_next);
  return hipGetLastError();
}

hipError_t xlp_launch_inverse(const XlpArgs &a, const XlDynArgs &dyn, const XlDynArgs &dyn_next, hipStream_t s) {
  hipLaunchKernelGGL(xlp_inverse_kernel, dim3(a.nco_blocks + a.nseg * a.nsg * 16u), dim3(256), 0, s, a, dyn, dyn_next);
  return hipGetLastError();
}
