// xl_inv8.hip -- the inverse launch of the polyphase path with EIGHT LANES PER CLIENT COLUMN (the default for 128-point
// classes; option "inverse_kernel" = 3 selects the LDS transform of xl_polyphase.hip instead).
//
// Same job as xlp_inverse_kernel (xl_polyphase.hip): per (segment, client column) the 128-point inverse transform of the mixed
// spectra, the valid outputs scaled, rotated by the client's NCO phases and stored (xlating.c:70 `out = temp * phase`).  What is
// different is how the transform is cut (xl_inv8_layout.h): 16 x 8 instead of 4 x 4 x 4 x 2, so that
//   * the tile goes from HBM straight into the registers of the lanes that transform it (16 loads of 8 bytes per lane) -- no
//     transposing fill pass through LDS, no workgroup barrier behind it;
//   * two of the three exchanges through LDS are gone (a lane's 16-point and 8-point transforms run in registers with
//     compile-time twiddles), and the one that is left, like the phase staging and the twiddle table, uses addresses of the form
//     "one register per lane + an immediate": the 4 x 4 x 4 x 2 kernel spends 40 % of its vector instructions on LDS addresses
//     (XOR swizzles per access), this one a handful (counters per 1024-client call: 21.1 M vector / 1.53 M LDS instructions against
//     31.1 M / 4.42 M, no bank-conflict cycles);
//   * a lane serves ONE client column in the epilogue (one XlpCol record, one set of output bounds) instead of four.
// A wave owns 8 columns of the tile and a private 8.5 KB LDS region; the four waves of a workgroup meet once, behind the
// twiddle table's fill (while their tile loads are in flight).
// grid = nco_blocks + nseg * ncg * 4 workgroups of 256 threads, as xlp_inverse_kernel<128>.
#include "xl_poly_dev.h"

#include "xl_inv8_layout.h"

#include <hip/hip_ext.h>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void xlp_inverse8_kernel(const XlpArgs a) {
  constexpr uint32_t M = 128u, CW = 32u, NSUB = XLP_COLS / CW;
  __shared__ __attribute__((aligned(16))) unsigned char region[4][XLI8_WAVE_BYTES];
  __shared__ v2f twl[16][8];  // e^{+2 pi j m1 t / 128}, [t][m1]
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;  // (as in xlp_mix_kernel)
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
#ifdef XLI8_EXP_SFAST  // experiment: consecutive workgroups = consecutive segments of the same 32 columns
  const uint32_t s = bid % a.nseg;
  const uint32_t q = bid / a.nseg;
  const uint32_t sub = q % NSUB, cg = q / NSUB;
#else
  const uint32_t sub = bid % NSUB;
  const uint32_t q = bid / NSUB;
  const uint32_t cg = q % a.ncg, s = q / a.ncg;
#endif
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  // ---- role 1: lane (m1, c8) takes bins m1 + 8 m2, m2 < 16, of column 8 w + c8: element (w * 16 + m2) * 64 + j of the tile
  v2f z[16];
  {
    // the tile in the common [bin][32 columns] order: a wave instruction reads eight 64-byte runs (bins m1 + 8 m2 of the wave's 8
    // columns), the other halves of the same lines being the neighbour wave's.  (A tile order of its own -- [wave][m2][m1][c8],
    // 512 contiguous bytes per instruction -- was built and measured: the inverse launch gains 2 %, the mix launch, whose stores
    // then go out in 64-byte runs, loses 21 %: profiles/r04_inverse8.txt)
    const v2f *__restrict__ tile = reinterpret_cast<const v2f *>(a.Y) + ((((size_t)cg * a.nseg_cap + s) * NSUB + sub) * M) * CW;
    const uint32_t o0 = xli8_load(w, j, 0u);
    constexpr uint32_t os = 8u * CW;  // = xli8_load(w, j, 1) - xli8_load(w, j, 0)
#pragma unroll
#ifdef XLI8_EXP_NOLOAD
    for (int m2 = 0; m2 < 16; ++m2) z[m2] = (v2f){(float)(o0 + m2 * os), 1.0f};
#else
    for (int m2 = 0; m2 < 16; ++m2) z[m2] = tile[o0 + m2 * os];
#endif
  }
  if (threadIdx.x < 128u) {  // the twiddle table: a.W = e^{-2 pi j n / 256}
    v2f tv = reinterpret_cast<const v2f *>(a.W)[(2u * (threadIdx.x & 7u) * (threadIdx.x >> 3)) & 255u];
    tv.y = -tv.y;
    twl[threadIdx.x >> 3][threadIdx.x & 7u] = tv;
  }
  // ---- the lane's client column in roles 3 (reader), 4, phase expansion and epilogue: column 8 w + (j >> 3).  It lies on the
  // class's shared grid with its own offset (xl_grid.h): its output k is the shared point q = k + shift, shift in {0, 1}, and
  // it owns K outputs in this call.  Phase expansion duty of lane (column, g): the phases of the shared points 16 g .. 16 g + 15
  // of the segment, from the one table entry requested here (the table holds every XL_PH_STRIDE-th phase).
  static_assert(XL_PH_STRIDE == 16u, "one table entry per lane: 8 entries per column and segment");
  const uint32_t c8 = xli8_col(j), u = xli8_u(j);
#ifdef XLI8_EXP_NOMETA  // experiment: no column record, no phase-table entry (synthetic rows of 199936 bytes)
  XlpCol ce;
  ce.out_off = (cg * XLP_COLS + sub * CW + 8u * w + c8) * 24992u, ce.delta = 0u, ce.incr = make_float2(1.0f, 0.0f);
#else
  const XlpCol ce = a.cols[cg * XLP_COLS + sub * CW + 8u * w + c8];
#endif
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;  // a column with j0 < Nr owns Ka + 1 outputs, else Ka
  XlBnd ebnd;
  ebnd.j0 = xl_merge_j0(a.j0_ref, ce.delta, a.D), ebnd.D = a.D, ebnd.S = a.pos.S, ebnd.G = a.pos.G, ebnd.flags = a.pos.pad;
  ebnd.K = Ka + (ebnd.j0 < Nr ? 1u : 0u);
  const uint32_t esh = xl_merge_shift(a.j0_ref, ce.delta, a.D);
  const uint32_t q0 = s * a.V + u * XL_PH_STRIDE;
  const uint32_t ibeg = q0 < esh ? 1u : 0u;  // (shared point 0 of a column with shift 1 is nobody's output)
  const uint32_t m0 = q0 + ibeg - esh;       // the column's output index of the first phase to expand
  const bool eok = ce.out_off != 0xFFFFFFFFu && u * XL_PH_STRIDE < a.V && m0 < ebnd.K;
#ifdef XLI8_EXP_NOMETA
  const v2f pe = {1.0f, eok ? 0.0f : 1.0f};
#else
  const v2f pe = reinterpret_cast<const v2f *>(a.phtab)[eok ? (ce.out_off >> XL_PH_SHIFT) + (m0 >> XL_PH_SHIFT) : 0u];
#endif
  __syncthreads();  // (the table; the tile loads are still travelling)
  unsigned char *const reg = region[w];
#ifdef XLI8_EXP_COPY  // experiment (wrong results): the launch's memory traffic alone -- tile loads, output stores in the same pattern, no transform, no phases
  if (ce.out_off != 0xFFFFFFFFu) {
    v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out) + ce.out_off;
    const uint32_t qs0 = s * a.V + u;
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t qo = 16u * g + 8u * e + u, qs = qs0 + 16u * g + 8u * e;
#ifdef XLI8_EXP_NOSTORE
        if (qo < a.V && qs >= esh && qs - esh < ebnd.K && z[8 * e + g].x == 1.2345e-33f) out[qs - esh] = z[8 * e + g] * pe.x;
#else
        if (qo < a.V && qs >= esh && qs - esh < ebnd.K) out[qs - esh] = z[8 * e + g] * pe.x;
#endif
      }
  }
  (void)reg;
  return;
#endif
  // ---- roles 1, 2: Z_m1[t] = 16-point inverse transform over m2, times w^{m1 t}; role 3: into the exchange rows
  xl_fft16_inverse<v2f, XlpFftOps>(z);
  {
    const uint32_t m1 = xli8_load_m1(j);
    const v2f *__restrict__ twp = &twl[0][m1];
    unsigned char *const wr = reg + xli8_exch(xli8_load_c8(j), 0u, m1);
    *reinterpret_cast<v2f *>(wr) = z[xli8_slot16(0)];
#pragma unroll
    for (int t = 1; t < 16; ++t) *reinterpret_cast<v2f *>(wr + t * XLI8_XROW) = xlp_cmul_v(z[xli8_slot16(t)], twp[t * 8]);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- role 3 (reader), role 4: rows t = u and u + 8 of the own column, 8-point inverse transforms over m1
  v2f y[2][8];
  {
    const unsigned char *const rd = reg + xli8_exch(c8, u, 0u);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v4f pr = *reinterpret_cast<const v4f *>(rd + e * 8 * XLI8_XROW + i * 16);
        y[e][2 * i] = (v2f){pr.x, pr.y};
        y[e][2 * i + 1] = (v2f){pr.z, pr.w};
      }
  }
  __builtin_amdgcn_wave_barrier();  // (everybody has read: the region is free for the phases)
  xl_fft8_inverse<v2f, XlpFftOps>(y[0]);
  xl_fft8_inverse<v2f, XlpFftOps>(y[1]);
  // ---- the phases of the wave's 8 x 128 shared points
  if (eok) {
    const uint32_t left = ebnd.K - m0, span = XL_PH_STRIDE - ibeg;
    unsigned char *const pw = reg + xli8_phase(c8, u * XL_PH_STRIDE + ibeg);
    xl_phase_walk(pe, m0, left < span ? left : span, (v2f){ce.incr.x, ce.incr.y}, ebnd,
                  [&](uint32_t i, v2f phs) { *reinterpret_cast<v2f *>(pw + i * 8u) = phs; });
  }
  __builtin_amdgcn_wave_barrier();
  // ---- epilogue: lane (c8, u) holds the shared points 16 g + 8 e + u of its column: a store instruction (fixed g, e) covers 8
  // consecutive outputs per column
  if (ce.out_off != 0xFFFFFFFFu) {
    v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out) + ce.out_off;
    const unsigned char *const pr = reg + xli8_phase(c8, u);
    const uint32_t qs0 = s * a.V + u;
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t qo = 16u * g + 8u * e + u, qs = qs0 + 16u * g + 8u * e;  // shared point of this value: in the segment, in the call
        if (qo < a.V && qs >= esh && qs - esh < ebnd.K) {
          const v2f val = y[e][xli8_slot8(g)] * (1.0f / (float)M);  // exact scaling by 2^-7
          out[qs - esh] = xl_rotate<1>(val, *reinterpret_cast<const v2f *>(pr + g * XLI8_PROW + e * 64));
        }
      }
  }
}

void xlp_inverse8_launch(const XlpArgs &a, const dim3 grid, hipStream_t s, hipEvent_t done) {
  if (done) hipExtLaunchKernelGGL(xlp_inverse8_kernel, grid, dim3(256), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(xlp_inverse8_kernel, grid, dim3(256), 0, s, a);
}
