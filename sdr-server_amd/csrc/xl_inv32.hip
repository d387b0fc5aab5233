// xl_inv32.hip -- the inverse launch of the polyphase path with the 128-point transform cut 32 x 4 (option "inverse_kernel" = 6; what
// the default, 0, takes for launches of more than 8192 tiles: xlp_inverse_pick).
//
// Same job as xlp_inverse_kernel (xl_polyphase.hip) and xlp_inverse8_kernel (xl_inv8.hip): per (segment, client column) the 128-point
// inverse transform of the mixed spectra, the valid outputs scaled, rotated by the client's NCO phases and stored (xlating.c:70
// `out = temp * phase`).  Built for what bounds this launch -- its memory traffic (tools/ubench_tile_copy.hip: with whole-line loads
// and 256-byte store runs the traffic alone takes 43.6 us per block at 4096 clients; in the 8-lane kernel's 64-byte runs, 58):
//   * a wave owns 16 client columns = one 128-byte line of every bin row of the tile: a load instruction reads four whole lines, and
//     the values land in the registers of the lane that transforms them (32-point transform over m2 in registers: no fill pass);
//   * one exchange through the wave's private LDS region -- in two rounds of 8 columns, so that a wave needs 10 KB of LDS, not 18.7: the
//     twelve waves per CU its 154 registers allow, not eight (-9 % launch time) -- then 4-point transforms in registers; lane (cc, t) ends up with outputs t + 32 g of column 2 k + cc: a store
//     instruction covers 256 consecutive bytes of each of two client rows;
//   * every LDS address is "one register per lane + an immediate" (xl_inv32_layout.h; tests/c/test_inv32_layout.cpp runs the same
//     index functions through an emulation of the lanes and a model of the banks);
//   * a wave shares nothing with its neighbours: no workgroup barrier, the unit of scheduling is half a tile.
// grid = nco_blocks + nseg * ncg * 4 workgroups of 128 threads (two waves = one tile).
#include "xl_poly_dev.h"

#include "xl_inv32_layout.h"

#include <hip/hip_ext.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// Registers: 154, three waves per SIMD.  (Held to 128 for four the kernel spills 24 registers and runs 35 % slower; a form that takes
// its two rounds over the rows instead of the columns fits 120 -- and loses more, 77 against 49 us per block at 4096 clients: its store
// instructions write 128-byte runs whose lines the other round completes microseconds later.  profiles/r05_inverse_cut32.txt)
#define XLI32_WPE 3
#define XLI32_WAVES 2  // waves per workgroup (1 / 2 / 4 measured: 48.5 / 47.5 / 58 us per block at 4096 clients)
__global__ __launch_bounds__(64 * XLI32_WAVES) __attribute__((amdgpu_waves_per_eu(XLI32_WPE, XLI32_WPE))) void xlp_inverse32_kernel(const XlpArgs a) {
  constexpr uint32_t WAVES = XLI32_WAVES, M = 128u, CW = 32u, NSUB = XLP_COLS / CW;
  __shared__ __attribute__((aligned(16))) unsigned char region[WAVES][XLI32_WAVE_BYTES];
  if (blockIdx.x < a.nco_blocks) {
    xlp_nco_role(a);
    return;
  }
  if (blockIdx.x >= a.nco_skip_at && blockIdx.x < a.nco_skip_at + a.nco_skip) return;  // (as in xlp_inverse_kernel)
  const uint32_t bid = blockIdx.x - a.nco_blocks - (blockIdx.x >= a.nco_skip_at ? a.nco_skip : 0u);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = threadIdx.x & 63u;
  const uint32_t unit = bid * WAVES + w;  // half tiles, in the tiles' order (sub fastest, then column group, then segment)
  const uint32_t half = unit & 1u, sub = (unit >> 1) % NSUB;
  const uint32_t qq = (unit >> 1) / NSUB;
  const uint32_t cg = qq % a.ncg, s = qq / a.ncg;
  if (s >= a.nseg) return;
  unsigned char *const reg = region[w];
  // ---- the factor table of role 2, [t][m1]: e^{+2 pi j m1 t / 128} / 128 (a.W = e^{-2 pi j n / 256}); requested first, so that its
  // wait is not a wait for the tile
  v2f tw0, tw1;
  {
    const v2f *__restrict__ W = reinterpret_cast<const v2f *>(a.W);
    tw0 = W[(2u * (j & 3u) * (j >> 2)) & 255u];
    tw1 = W[(2u * (j & 3u) * ((j >> 2) + 16u)) & 255u];
  }
  // ---- the lane's two client columns of the third role (phase expansion): column 8 r + (j >> 3) of the wave's 16, one per round.  A
  // column lies on the class's shared grid with its own offset (xl_grid.h): its output k is the shared point k + shift, shift in
  // {0, 1}, and it owns K outputs in this call.
  static_assert(XL_PH_STRIDE == 16u, "one table entry per lane and round: 8 entries per column and segment");
  const uint32_t col0 = cg * XLP_COLS + sub * CW + 16u * half;
  XlpCol ce[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) ce[r] = a.cols[col0 + 8u * r + xli32_walk_c8(j)];
  // ---- role 1: lane (m1, c) takes bins m1 + 4 m2, m2 < 32, of column c: four whole lines per instruction
  v2f z[32];
  {
    const v2f *__restrict__ tile = reinterpret_cast<const v2f *>(a.Y) + ((((size_t)cg * a.nseg_cap + s) * NSUB + sub) * M) * CW;
    const uint32_t o0 = xli32_load(half, j, 0u);
#pragma unroll
    for (int m2 = 0; m2 < 32; ++m2) z[m2] = tile[o0 + m2 * 4u * CW];
  }
  {
    tw0.y = -tw0.y, tw1.y = -tw1.y;
    *reinterpret_cast<v2f *>(reg + xli32_tw(j >> 2, j & 3u)) = tw0 * (1.0f / (float)M);  // exact scaling by 2^-7
    *reinterpret_cast<v2f *>(reg + xli32_tw((j >> 2) + 16u, j & 3u)) = tw1 * (1.0f / (float)M);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- roles 1, 2: Z_m1[t] = 32-point inverse transform over m2, times w^{m1 t} / 128 (in place)
  xl_fft32_inverse<v2f, XlpFftOps>(z);
#pragma unroll
  for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(z[i]));  // (all 32 outputs now, not the last stage's operands: fewer registers)
  // ---- what the walk and the stores need of the two columns, and the walk's table entries (the table holds every 16th phase)
  const uint32_t N = a.pos.S * a.pos.G;
  const uint32_t Ka = N / a.D, Nr = N - Ka * a.D;  // a column with j0 < Nr owns Ka + 1 outputs, else Ka
  const uint32_t gq = xli32_walk_gq(j);
  XlBnd ebnd[2];
  uint32_t m0[2], cnt[2];  // the column's output index of the first phase to expand, phases to expand (0: none)
  v2f pe[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    ebnd[r].j0 = xl_merge_j0(a.j0_ref, ce[r].delta, a.D), ebnd[r].D = a.D, ebnd[r].S = a.pos.S, ebnd[r].G = a.pos.G, ebnd[r].flags = a.pos.pad;
    ebnd[r].K = ce[r].out_off != 0xFFFFFFFFu ? Ka + (ebnd[r].j0 < Nr ? 1u : 0u) : 0u;
    const uint32_t esh = xl_merge_shift(a.j0_ref, ce[r].delta, a.D);
    const uint32_t q0 = s * a.V + gq * XL_PH_STRIDE;
    const uint32_t ibeg = q0 < esh ? 1u : 0u;  // (shared point 0 of a column with shift 1 is nobody's output)
    m0[r] = q0 + ibeg - esh;
    const bool eok = gq * XL_PH_STRIDE < a.V && m0[r] < ebnd[r].K;
    const uint32_t left = ebnd[r].K - m0[r], span = XL_PH_STRIDE - ibeg;
    cnt[r] = eok ? (left < span ? left : span) : 0u;
    pe[r] = reinterpret_cast<const v2f *>(a.phtab)[eok ? (ce[r].out_off >> XL_PH_SHIFT) + (m0[r] >> XL_PH_SHIFT) : 0u];
    if (gq == 0u) *reinterpret_cast<v4u *>(reg + xli32_meta(8u * r + xli32_walk_c8(j))) = (v4u){ce[r].out_off, esh, ebnd[r].K, ibeg};
    m0[r] |= ibeg << 31;  // (kept for the walk's LDS address)
  }
  {
    const unsigned char *const twp = reg + xli32_tw(0u, xli32_load_m1(j));
#pragma unroll
    for (int t = 0; t < 32; ++t) z[xli32_slot32(t)] = xlp_cmul_v(z[xli32_slot32(t)], *reinterpret_cast<const v2f *>(twp + t * 32));
  }
  const uint32_t cc = xli32_cc(j), t = xli32_t(j);
  const uint32_t qs0 = s * a.V + t;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    // ---- role 3: the lanes of the round's 8 columns put their rows into the exchange region; reader lane (cc, t) of pass k takes row
    // t of column 2 k + cc; role 4: 4-point inverse transforms over m1
    if ((xli32_load_c(j) >> 3) == (uint32_t)r) {
      unsigned char *const wr = reg + xli32_exch(xli32_load_c(j) & 7u, xli32_load_m1(j), 0u);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const v2f p0 = z[xli32_slot32(2 * i)], p1 = z[xli32_slot32(2 * i + 1)];
        *reinterpret_cast<v4f *>(wr + i * 16) = (v4f){p0.x, p0.y, p1.x, p1.y};
      }
    }
    __builtin_amdgcn_wave_barrier();
    v2f y[4][4];
    {
      const unsigned char *const rd = reg + xli32_exch(cc, 0u, t);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m1 = 0; m1 < 4; ++m1) y[k][m1] = *reinterpret_cast<const v2f *>(rd + k * 2 * XLI32_XCOL + m1 * XLI32_XROW);
    }
    __builtin_amdgcn_wave_barrier();  // (everybody has read: the region is free for the phases)
#pragma unroll
    for (int k = 0; k < 4; ++k) xl_fft4_inverse<v2f, XlpFftOps>(y[k]);
    // ---- the phases of the round's 8 x 128 shared points
    if (cnt[r] != 0u) {
      const uint32_t ibeg = m0[r] >> 31;
      unsigned char *const pw = reg + xli32_phase(xli32_walk_c8(j), gq * XL_PH_STRIDE + ibeg);
      xl_phase_walk(pe[r], m0[r] & 0x7FFFFFFFu, cnt[r], (v2f){ce[r].incr.x, ce[r].incr.y}, ebnd[r],
                    [&](uint32_t i, v2f phs) { *reinterpret_cast<v2f *>(pw + i * 8u) = phs; });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- epilogue: lane (cc, t) holds the shared points t + 32 g of columns 2 k + cc: a store instruction (fixed k, g) covers 32
    // consecutive outputs of each of two columns
    {
      const unsigned char *const pr = reg + xli32_phase(cc, t);  // point t + 32 g: + g * 2 rows; column 2 k + cc: + k * 2 columns
      const unsigned char *const mr = reg + xli32_meta(8u * r + cc);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const v4u mt = *reinterpret_cast<const v4u *>(mr + k * 32);  // row offset, shift, outputs owned (0: vacant column)
        v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out) + mt.x;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t qo = t + 32u * g, idx = qs0 + 32u * g - mt.y;  // (a point ahead of the column's first output wraps past K)
          if (qo < a.V && idx < mt.z)
            out[idx] = xl_rotate<1>(y[k][g], *reinterpret_cast<const v2f *>(pr + k * 2 * XLI32_PCOL + g * 2 * XLI32_PROW));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the phases have been read: the region is free for the next round's rows)
  }
}

// `tiles` = nseg * ncg * 4; the launch's work = 2 * tiles half tiles, XLI32_WAVES per workgroup
uint32_t xlp_inverse32_work(uint32_t tiles) { return (2u * tiles + XLI32_WAVES - 1u) / XLI32_WAVES; }

void xlp_inverse32_launch(const XlpArgs &a, const dim3 grid, hipStream_t s, hipEvent_t done) {
  if (done) hipExtLaunchKernelGGL(xlp_inverse32_kernel, grid, dim3(64u * XLI32_WAVES), 0, s, nullptr, done, 0, a);
  else hipLaunchKernelGGL(xlp_inverse32_kernel, grid, dim3(64u * XLI32_WAVES), 0, s, a);
}
