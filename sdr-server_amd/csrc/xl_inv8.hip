// xl_inv8.hip -- the inverse launch of the polyphase path with EIGHT LANES PER CLIENT COLUMN (option "inverse_kernel" 5, the
// default for 128-point classes).
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

// ------------------------------------------------------------------------------------------- the same, PERSISTENT (option)
// a.inv_wgs work workgroups walk the tiles (bid, bid + inv_wgs, ..); every wave runs on its own after the start (the four waves of a
// workgroup share nothing but the twiddle table).  The next tile -- its column record, its 16 values per lane, then its phase-table
// entry -- is requested while the current tile's exchange, second transforms, phases and stores run, and the stores of a tile are
// never waited for: the epilogue is branch-free (a point that is nobody's output is stored to a dump address instead -- the wave's
// own, already consumed element of the tile in Y), so that the compiler's wait counts stay exact, and one `s_waitcnt vmcnt(16)`
// behind the sixteen stores says "the next tile has arrived, the stores may still fly" (the compiler would add what it needs on
// top: the explicit wait only keeps it from falling back to vmcnt(0) at the loop's back edge).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void xlp_inverse8p_kernel(const XlpArgs a) {
  constexpr uint32_t M = 128u, CW = 32u, NSUB = XLP_COLS / CW;
  __shared__ __attribute__((aligned(16))) unsigned char region[4][XLI8_WAVE_BYTES];
  __shared__ v2f twl[16][8];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t ntiles = a.nseg * a.ncg * NSUB, stride = a.inv_wgs;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const uint32_t c8 = xli8_col(j), u = xli8_u(j);
  if (threadIdx.x < 128u) {
    v2f tv = reinterpret_cast<const v2f *>(a.W)[(2u * (threadIdx.x & 7u) * (threadIdx.x >> 3)) & 255u];
    tv.y = -tv.y;
    twl[threadIdx.x >> 3][threadIdx.x & 7u] = tv;
  }
  __syncthreads();
  unsigned char *const reg = region[w];
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;
  const v2f *__restrict__ ph = reinterpret_cast<const v2f *>(a.phtab);
  const uint32_t o0 = xli8_load(w, j, 0u);
  constexpr uint32_t os = 8u * CW;
  // tile t -> its first element in Y, its first column, its segment
  auto tile_of = [&](const uint32_t t, uint32_t &col0, uint32_t &seg) __attribute__((always_inline)) {
    const uint32_t sub = t % NSUB, q = t / NSUB;
    const uint32_t cg = q % a.ncg;
    seg = q / a.ncg;
    col0 = cg * XLP_COLS + sub * CW + 8u * w;
    return reinterpret_cast<v2f *>(a.Y) + ((((size_t)cg * a.nseg_cap + seg) * NSUB + sub) * M) * CW;
  };
  // what the lane needs of its column in a tile: bounds, shift, the output index of the first phase it expands
  struct Duty {
    XlBnd bnd;
    uint32_t esh, ibeg, m0;
    bool eok;
  };
  auto duty_of = [&](const XlpCol &c, const uint32_t seg) __attribute__((always_inline)) {
    Duty d;
    d.bnd.j0 = xl_merge_j0(a.j0_ref, c.delta, a.D), d.bnd.D = a.D, d.bnd.S = a.pos.S, d.bnd.G = a.pos.G, d.bnd.flags = a.pos.pad;
    d.bnd.K = Ka + (d.bnd.j0 < Nr ? 1u : 0u);
    d.esh = xl_merge_shift(a.j0_ref, c.delta, a.D);
    const uint32_t q0 = seg * a.V + u * XL_PH_STRIDE;
    d.ibeg = q0 < d.esh ? 1u : 0u;
    d.m0 = q0 + d.ibeg - d.esh;
    d.eok = c.out_off != 0xFFFFFFFFu && u * XL_PH_STRIDE < a.V && d.m0 < d.bnd.K;
    return d;
  };
  uint32_t t = bid, col0, seg;
  v2f *tile = tile_of(t, col0, seg);
  XlpCol ce = a.cols[col0 + c8];
  v2f z[16];
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) z[m2] = tile[o0 + m2 * os];
  v2f pe;
  {
    const Duty d = duty_of(ce, seg);
    pe = ph[d.eok ? (ce.out_off >> XL_PH_SHIFT) + (d.m0 >> XL_PH_SHIFT) : 0u];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered with nothing in flight
#pragma unroll 1
  for (;;) {
    const bool last = t + stride >= ntiles;
    const uint32_t tn = last ? t : t + stride;  // (the last round requests its own tile again: no branch around the loads)
    const Duty d = duty_of(ce, seg);
    v2f *const dump = tile + o0;  // this lane's first element of the tile: consumed, nobody else's
    // ---- roles 1 - 3 (writer)
    xl_fft16_inverse<v2f, XlpFftOps>(z);
    {
      const uint32_t m1 = xli8_load_m1(j);
      const v2f *__restrict__ twp = &twl[0][m1];
      unsigned char *const wr = reg + xli8_exch(xli8_load_c8(j), 0u, m1);
      *reinterpret_cast<v2f *>(wr) = z[xli8_slot16(0)];
#pragma unroll
      for (int tt = 1; tt < 16; ++tt) *reinterpret_cast<v2f *>(wr + tt * XLI8_XROW) = xlp_cmul_v(z[xli8_slot16(tt)], twp[tt * 8]);
    }
    // ---- the next tile: column record, then the 16 values per lane (into the registers just freed)
    uint32_t col0n, segn;
    v2f *const tilen = tile_of(tn, col0n, segn);
    const XlpCol cen = a.cols[col0n + c8];
#pragma unroll
    for (int m2 = 0; m2 < 16; ++m2) z[m2] = tilen[o0 + m2 * os];
    __builtin_amdgcn_wave_barrier();
    // ---- role 3 (reader), role 4
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
    __builtin_amdgcn_wave_barrier();
    xl_fft8_inverse<v2f, XlpFftOps>(y[0]);
    xl_fft8_inverse<v2f, XlpFftOps>(y[1]);
    // ---- phases of this tile
    if (d.eok) {
      const uint32_t left = d.bnd.K - d.m0, span = XL_PH_STRIDE - d.ibeg;
      unsigned char *const pw = reg + xli8_phase(c8, u * XL_PH_STRIDE + d.ibeg);
      xl_phase_walk(pe, d.m0, left < span ? left : span, (v2f){ce.incr.x, ce.incr.y}, d.bnd,
                    [&](uint32_t i, v2f phs) { *reinterpret_cast<v2f *>(pw + i * 8u) = phs; });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the next tile's phase-table entry (its column record has been back for a while: it was requested ahead of the tile)
    const Duty dn = duty_of(cen, segn);
    const v2f pen = ph[dn.eok ? (cen.out_off >> XL_PH_SHIFT) + (dn.m0 >> XL_PH_SHIFT) : 0u];
    // ---- epilogue, branch-free
    {
      v2f *const out = reinterpret_cast<v2f *>(a.out) + (ce.out_off != 0xFFFFFFFFu ? ce.out_off : 0u);
      const bool colok = ce.out_off != 0xFFFFFFFFu;
      const unsigned char *const pr = reg + xli8_phase(c8, u);
      const uint32_t qs0 = seg * a.V + u;
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t qo = 16u * g + 8u * e + u, qs = qs0 + 16u * g + 8u * e;
          const bool ok = colok && qo < a.V && qs >= d.esh && qs - d.esh < d.bnd.K;
          const v2f val = y[e][xli8_slot8(g)] * (1.0f / (float)M);
          // (a point without a phase in the region reads whatever is there: its product goes to the dump)
          const v2f r = xl_rotate<1>(val, *reinterpret_cast<const v2f *>(pr + g * XLI8_PROW + e * 64));
          v2f *const dst = ok ? out + (qs - d.esh) : dump;
          *dst = r;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16): the next tile and its table entry are here, the sixteen stores may still fly
    if (last) break;
    t = tn, tile = tilen, seg = segn, ce = cen, pe = pen;
  }
}

void xlp_inverse8_launch(const XlpArgs &a, const dim3 grid, hipStream_t s, hipEvent_t done) {
  void (*kern)(const XlpArgs) = a.inv_wgs ? xlp_inverse8p_kernel : xlp_inverse8_kernel;
  if (done) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
}
