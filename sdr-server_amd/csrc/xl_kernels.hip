// xl_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for sdr-server's frequency-xlating FIR path.
//
// Reference semantics: /root/reference/src/xlating.c (cited per kernel).  Compiled with -ffp-contract=off:
// nothing is fused unless written as __builtin_fmaf, so the "native" kernels evaluate exactly the reference's
// scalar float32 expression tree (one rounding per multiply/add) and the "optimized" kernels fuse by choice.
//
// Kernel map
//   xl_fir_kernel<CT,MODE>   fused  sample convert -> LDS window image -> T-tap complex FIR at stride D ->
//                            NCO rotate -> store, for a batch of clients.           (xlating.c:52-72, 352-414)
//   xl_nco_table_kernel      float32 phase recurrence + per-call hypotf renormalisation (xlating.c:71,73)
//   xl_convert_*             raw -> cf32 / Q15 sample images of the single-filter path  (xlating.c:356-433)
//   xl_move_down_kernel      history memmove                                            (xlating.c:76-79)
//   xl_update_history_kernel raw history roll of the batch engine
//   xl_fir_q15_kernel / xl_nco_table_q15_kernel   the Q15 family                        (xlating.c:92-140)
//
// Work decomposition of xl_fir_kernel (the hot kernel):
//   lane      = one output sample m (64 consecutive outputs per wave)           -> no cross-lane reduction
//   registers = CT complex accumulators, one per client of the wave's tile      -> each LDS sample read feeds
//                                                                                  CT complex MACs (4*CT FMAs)
//   taps      = wave-uniform: fetched with SCALAR loads (s_load_dwordx16 = tap i of 8 clients) from a
//               [tap][client] interleaved image and used as SGPR operands of v_fma_f32 -> zero VALU/LDS cost
//   window    = converted once per workgroup into LDS as cf32; the NW (4..8) waves of a workgroup (NW tiles = up
//               to 8*NW clients) share it.  Lane m reads sample (m*D + i): for odd D ds_read_b64 is bank-conflict
//               free; for even D two samples are read per ds_read_b128 (conflict-free when D = 2 mod 4).
//   grid      = (groups) x (output tiles), group-major, cut into 8 equal contiguous chunks, one per XCD: the
//               output tiles of a group re-read the same taps, which then stay in that XCD's L2.
#include "xl_dev_inline.h"

#include <hip/hip_ext.h>

#include <stdlib.h>

// ------------------------------------------------------------------------------------------- NCO phase table
// xlating.c:70-73: the phasor is a float32 RECURRENCE p <- p * incr (never re-seeded), renormalised once per
// call that could produce output.  It is data independent, so one lane per client tabulates the K phases of
// the block ahead of the FIR kernel; the recurrence itself must stay sequential to be bit-exact.
// hypotf: glibc evaluates sqrt(x*x + y*y) in double and narrows; restated with IEEE double ops.
// Stand-alone launch (single-filter path; first block / wrong length guess of the batch engine).  `lanes`
// (XL_NCO_LANES) lanes of a wave carry a client each; every 4th phase is stored, two entries (16 bytes) per store, to
// the client's own table row.  The kernel is a pure dependent chain: ~9 ns per step (two dependent packed
// operations) plus the store issue time, which is why only every 4th phase is tabulated (tools/ubench_chain2.hip).
__global__ __launch_bounds__(64) void xl_nco_table_kernel(const XlNcoClient *__restrict__ cl, uint32_t n,
                                                          const float2 *state_in, float2 *state_out,
                                                          float2 *__restrict__ tab, const XlPos pos,
                                                          const uint32_t explicit_K, const uint32_t prio,
                                                          const uint32_t lanes) {
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (threadIdx.x >= lanes) return;
  const uint32_t c = blockIdx.x * lanes + threadIdx.x;
  if (c >= n) return;
  const XlNcoClient k = cl[c];
  const XlBnd bnd = xl_nco_bnd(k, pos, explicit_K);
  // This wave steps with the PACKED instructions (the single filter's look-ahead table is on the drop-in's critical path: 44 -> 56 us
  // per block with the scalar step), so it must own its SIMD: packed FP32 chains next to another launch's matrix instructions lose
  // lanes 48..63.  v255 / a255 clobbered = the descriptor asks for all 512 registers: one wave per SIMD, like the
  // side-stream chain kernel below.
  asm volatile("" ::: "v255", "a255");
  xl_nco_client_chain<true>(k, bnd, 0u, bnd.K, state_in, state_out, tab);
}

// The same tabulation for a call of many blocks, run on the engine's side stream WHILE the previous call's launches run
// on the main stream (xl_batch.cpp).  The recurrence is a dependent chain of ~16.5 cycles per step and it is what bounds
// a call once the filtering itself takes less than ~22 us per block, so everything that could slow the chain wave is
// kept away from it:
//  * neighbours on its SIMD cost it 35-60 % (their packed FMAs occupy the VALU 4 cycles at a time: 7.3 ns per step
//    alone, 9.8-13.5 ns next to the mix kernel's waves) -> the kernel claims every VGPR of its SIMDs (v255 / a255 are
//    declared clobbered: the kernel descriptor then asks for 512 registers per lane and the hardware places ONE wave per
//    SIMD); a workgroup of four waves is alone on its CU;
//  * its own table stores cost it most: on a chip whose memory system is saturated by the mix / inverse launches every
//    global store blocked the wave's instruction issue ~265 ns (measured: 15.6 ns per step with one 16-byte store per 32
//    steps, whatever the layout) -> the chain wave (wave 0) only writes the phases into an LDS ring; the other three
//    waves of the workgroup drain the ring into the table, and it is they who wait for the memory system.
// One workgroup = 64 clients: wave 0 lane l carries client blockIdx * 64 + l through all its steps in lockstep with
// the other lanes; drainer j (waves 1-3) stores the entry pairs q = j, j + 3, ... (16 bytes per client).
#define XLC_RING 64u  // ring entries (x 64 lanes x 8 bytes = 32 KB)
// LDS mailbox operations of the chain kernel, hand-placed: `volatile` accesses would make the compiler wait for ALL
// outstanding memory operations (vmcnt(0) + lgkmcnt(0)) around each of them -- an LDS round trip per table entry on the
// chain wave, a completed global store per poll on the drainers.  LDS operations of one wave execute in order.
XL_DEV uint32_t xl_lds_off(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
XL_DEV void xl_lds_post(const uint32_t addr, const uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
XL_DEV void xl_lds_post64(const uint32_t addr, const v2f v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
XL_DEV uint32_t xl_lds_poll(const uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// One table entry of the chain wave = 16 recurrence steps, hand-scheduled.  A lone wave issues in order, and each step
// is mul, mul, (wait ~6 cycles), add, (wait ~6 cycles): an independent instruction placed in a wait costs nothing,
// anywhere else it costs its 4-5 issue cycles (the compiler's version of the entry bookkeeping took the step from 17.5
// to 25.7 cycles).  So the bookkeeping rides in the shadow of the first steps' multiplies:
//   step 0: the entry (phase of output 16 e, still in p) into the ring    step 1-2: entry count + 1, posted
//   step 3-5: ring address of the next entry ((offset + 512) mod 32 KB + base)
// Every block below starts on a 64-byte boundary: a lone wave pays for an 8-byte instruction that straddles a fetch
// line (~6 cycles) -- the same blocks ran at 16.5 or at 18.8 cycles per step depending on where the compiler put them.
#define XLC_ALIGN ".p2align 6\n\t"
#define XLC_MUL "v_pk_mul_f32 %[t1], %[p], %[inc] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %[t2], %[p], %[inc] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
#define XLC_ADD "v_pk_add_f32 %[p], %[t1], %[t2] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
#define XLC_STEP XLC_MUL XLC_ADD
#define XLC_ENTRY                                                       \
  XLC_MUL "ds_write_b64 %[addr], %[p]\n\t" XLC_ADD                      \
  XLC_MUL "v_add_u32 %[cnt], 1, %[cnt]\n\t" XLC_ADD                     \
  XLC_MUL "ds_write_b32 %[paddr], %[cnt]\n\t" XLC_ADD                   \
  XLC_MUL "v_add_u32 %[off], 0x200, %[off]\n\t" XLC_ADD                 \
  XLC_MUL "v_and_b32 %[off], 0x7fff, %[off]\n\t" XLC_ADD                \
  XLC_MUL "v_add_u32 %[addr], %[off], %[base]\n\t" XLC_ADD              \
  XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP
// Four consecutive entries whose ring slots do not wrap (entry index = 0 mod 4): the slots are addressed with immediate
// offsets, the ring address is advanced once per four entries, and the loop around it closes once per 64 steps.
#define XLC_STEPS10 XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP
#define XLC_QENTRY(OFFS) XLC_MUL "ds_write_b64 %[addr], %[p] offset:" OFFS "\n\t" XLC_ADD XLC_STEP XLC_STEP
// (the entry count is posted once per block of entries: the drainers take the entries in pairs and are a few entries
// behind anyway; fewer instructions in the chain wave is what counts -- each one outside a wait costs ~3.6 cycles)
#define XLC_BLOCK_END(N, ADV)                                           \
  XLC_MUL "v_add_u32 %[cnt], " N ", %[cnt]\n\t" XLC_ADD                 \
  XLC_MUL "ds_write_b32 %[paddr], %[cnt]\n\t" XLC_ADD                   \
  XLC_MUL "v_add_u32 %[off], " ADV ", %[off]\n\t" XLC_ADD               \
  XLC_MUL "v_and_b32 %[off], 0x7fff, %[off]\n\t" XLC_ADD                \
  XLC_MUL "v_add_u32 %[addr], %[off], %[base]\n\t" XLC_ADD XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP XLC_STEP
#define XLC_QUAD                                                        \
  XLC_QENTRY("0") XLC_STEP XLC_STEP XLC_STEP XLC_STEPS10                \
  XLC_QENTRY("512") XLC_STEP XLC_STEP XLC_STEP XLC_STEPS10              \
  XLC_QENTRY("1024") XLC_STEP XLC_STEP XLC_STEP XLC_STEPS10             \
  XLC_MUL "ds_write_b64 %[addr], %[p] offset:1536\n\t" XLC_ADD XLC_BLOCK_END("4", "0x800")
// a whole entry without bookkeeping: 16 steps, the ring slot at an immediate offset
#define XLC_E(OFFS) XLC_QENTRY(OFFS) XLC_STEP XLC_STEP XLC_STEP XLC_STEPS10
// eight / sixteen / thirty-two entries (entry index = 0 mod 8 / 16 / 32): straight-line code, 13 instructions of
// bookkeeping per block.  Longer blocks are faster -- measured cycles per step: single entries 19.7, blocks of four 19.2,
// of eight 16.7: fewer extra instructions and fewer taken branches (a lone wave pays each refetch in full).
#define XLC_E7(B0, B1, B2, B3, B4, B5, B6) XLC_E(B0) XLC_E(B1) XLC_E(B2) XLC_E(B3) XLC_E(B4) XLC_E(B5) XLC_E(B6)
#define XLC_E8(B0, B1, B2, B3, B4, B5, B6, B7) XLC_E7(B0, B1, B2, B3, B4, B5, B6) XLC_E(B7)
#define XLC_OCT                                                         \
  XLC_E7("0", "512", "1024", "1536", "2048", "2560", "3072")           \
  XLC_MUL "ds_write_b64 %[addr], %[p] offset:3584\n\t" XLC_ADD XLC_BLOCK_END("8", "0x1000")
#define XLC_HEX                                                         \
  XLC_E8("0", "512", "1024", "1536", "2048", "2560", "3072", "3584")   \
  XLC_E7("4096", "4608", "5120", "5632", "6144", "6656", "7168")       \
  XLC_MUL "ds_write_b64 %[addr], %[p] offset:7680\n\t" XLC_ADD XLC_BLOCK_END("16", "0x2000")
#define XLC_B32                                                         \
  XLC_E8("0", "512", "1024", "1536", "2048", "2560", "3072", "3584")   \
  XLC_E8("4096", "4608", "5120", "5632", "6144", "6656", "7168", "7680") \
  XLC_E8("8192", "8704", "9216", "9728", "10240", "10752", "11264", "11776") \
  XLC_E7("12288", "12800", "13312", "13824", "14336", "14848", "15360") \
  XLC_MUL "ds_write_b64 %[addr], %[p] offset:15872\n\t" XLC_ADD XLC_BLOCK_END("32", "0x4000")
static_assert(XL_PH_STRIDE == 16u && XLC_RING * 64u * 8u == 0x8000u, "XLC_ENTRY is written for 16 steps per entry and a 32 KB ring");

// One launch tabulates `calls.n` consecutive calls of the same shape (pos, then xl_grid_next of it, ...): table and final
// phases per call -- the launch's fixed costs (~11 us of prologue / epilogue, ~6 us between two dependent launches) are paid
// once per calls.n calls.  The phases stay in the chain wave's registers from one call to the next.
__global__ __launch_bounds__(256) void xl_nco_chain_kernel(const XlNcoClient *__restrict__ cl, uint32_t n,
                                                           const float2 *state_in, const XlChainCalls calls,
                                                           XlPos pos, unsigned long long *stats) {
  asm volatile("" ::: "v255", "a255");
  const unsigned long long t_entry = stats ? wall_clock64() : 0ull;  // (tuning: 100 MHz ticks)
  __shared__ v2f ring[XLC_RING][64];
  __shared__ uint32_t s_emax;
  __shared__ uint32_t s_prod;     // entries written by the chain wave
  __shared__ uint32_t s_next[3];  // per drainer: the next entry pair it will store
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t c = blockIdx.x * 64u + lane;
  const bool have = c < n;
  XlNcoClient k;
  k.incr = make_float2(1.0f, 0.0f);
  k.out_off = 0u, k.slot = 0u, k.D = 1u, k.rem0 = 0u;
  if (have) k = cl[c];
  v2f p = {1.0f, 0.0f};  // (the chain wave's running phases)
  if (w == 0 && have) p = (v2f){state_in[k.slot].x, state_in[k.slot].y};
  unsigned long long cyc = 0ull, ticks = 0ull, wfirst = 0ull;
  uint32_t entries = 0u;
  for (uint32_t call = 0; call < calls.n; ++call, pos = xl_grid_next(pos)) {
    float2 *const tab = calls.tab[call];
    float2 *const state_out = calls.state_out[call];
    if (call) __syncthreads();  // (the ring and its counters start over)
    XlBnd bnd = xl_nco_bnd(k, pos, 0xFFFFFFFFu);
    if (!have) bnd.K = 0u;
    const uint32_t K = bnd.K, E = (K + XL_PH_STRIDE - 1u) >> XL_PH_SHIFT;  // this client's table entries
    if (threadIdx.x == 0) {
      s_emax = 0u;
      s_prod = 0u;
      s_next[0] = 0u, s_next[1] = 1u, s_next[2] = 2u;
    }
    __syncthreads();
    if (w == 0 && E > 0u) atomicMax(&s_emax, E);
    __syncthreads();
    const uint32_t Emax = s_emax;
    const uint32_t a_prod = xl_lds_off(&s_prod), a_ring0 = xl_lds_off(&ring[0][0]), a_ring = xl_lds_off(&ring[0][lane]);
    v2f *__restrict__ o = reinterpret_cast<v2f *>(tab) + (k.out_off >> XL_PH_SHIFT);  // out_off = 0 mod 2 * XL_PH_STRIDE: 16-byte pairs
    if (w == 0) {
      __builtin_amdgcn_s_setprio(3);
      const v2f inc = {k.incr.x, k.incr.y};
      uint32_t nb = xl_bnd_next(bnd, 0u);  // the phase is renormalised after output nb - 1 (xlating.c:73)
      const uint32_t a_n0 = xl_lds_off(&s_next[0]), a_n1 = xl_lds_off(&s_next[1]), a_n2 = xl_lds_off(&s_next[2]);
      const unsigned long long c0 = stats ? clock64() : 0ull, w0 = stats ? wall_clock64() : 0ull;  // (tuning: shader cycles / 100 MHz ticks)
      uint32_t e = 0;  // (wave-uniform: the lanes step in lockstep)
      while (e < Emax) {
        // the next output index at which ANY lane has something other than a plain step to do: its block ends (renormalise)
        // or its call ends.  Up to there the loop below is branch-free per lane: an entry into the ring, 16 steps.
        uint32_t ev = (e << XL_PH_SHIFT) < K ? (nb < K ? nb : K) : 0xFFFFFFFFu;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)ev, sh);
          ev = other < ev ? other : ev;
        }
        const uint32_t evs = __builtin_amdgcn_readfirstlane(ev);
        // entries e .. e_stop - 1: their steps hold no block end for anybody ((e + 1) * 16 < evs)
        uint32_t e_stop = evs == 0u ? 0u : (evs - 1u) >> XL_PH_SHIFT;
        e_stop = e_stop < Emax ? e_stop : Emax;
        if (pos.pad & XL_POS_FMA_STEP) e_stop = e;  // (the hand-scheduled blocks are the plain step: a call with the FMA-contracted
                                                    // step of an -mfma reference build takes the per-step path throughout)
        if ((e << XL_PH_SHIFT) < K) {  // (a lane whose call has ended sits the region out; the others' mask is constant in it)
          uint32_t ee = e;
          while (ee < e_stop) {
            if ((ee & (XLC_RING / 2u - 1u)) == 0u && ee >= XLC_RING) {
              // the next XLC_RING / 2 entries go to the slots of entries e - RING .. e - RING / 2 - 1: all pairs below
              // (e - RING / 2) / 2 must have left the ring (checked once per half ring: an LDS round trip is ~50 ns).
              // Bounded: a drainer that never shows up must not hang the device (cannot happen while the four waves of
              // the workgroup are resident, which a launch guarantees) -- ~0.1 s, then the table is wrong, the launch ends.
              const uint32_t q = (ee - XLC_RING / 2u) >> 1;
              for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
                if (xl_lds_poll(a_n0) >= q && xl_lds_poll(a_n1) >= q && xl_lds_poll(a_n2) >= q) break;
                __builtin_amdgcn_s_sleep(1);
              }
            }
            // entries ee .. chunk_end - 1 (up to the next drain check), two per trip
            const uint32_t next_check = (ee | (XLC_RING / 2u - 1u)) + 1u;
            const uint32_t chunk_end = e_stop < next_check ? e_stop : next_check;
            uint32_t off = ((ee & (XLC_RING - 1u)) << 9) + lane * (uint32_t)sizeof(v2f);  // ring offset of entry ee, this lane
            uint32_t addr = a_ring0 + off, cnt = ee;
            v2f t1, t2;
#define XLC_RUN(BLOCK)                                                                                            \
  asm volatile(XLC_ALIGN BLOCK                                                                                   \
               : [p] "+v"(p), [off] "+v"(off), [addr] "+v"(addr), [cnt] "+v"(cnt), [t1] "=&v"(t1), [t2] "=&v"(t2) \
               : [inc] "v"(inc), [base] "v"(a_ring0), [paddr] "v"(a_prod)                                        \
               : "memory")
            for (; ee < chunk_end && (ee & 3u) != 0u; ++ee) XLC_RUN(XLC_ENTRY);  // up to a multiple of four
            if ((ee & 7u) == 4u && ee + 4u <= chunk_end) {                       // up to a multiple of eight
              XLC_RUN(XLC_QUAD);
              ee += 4u;
            }
            if ((ee & 15u) == 8u && ee + 8u <= chunk_end) {  // up to a multiple of sixteen
              XLC_RUN(XLC_OCT);
              ee += 8u;
            }
            if ((ee & 31u) == 16u && ee + 16u <= chunk_end) {
              XLC_RUN(XLC_HEX);
              ee += 16u;
            }
            for (; ee + 32u <= chunk_end; ee += 32u) XLC_RUN(XLC_B32);
            for (; ee + 16u <= chunk_end; ee += 16u) XLC_RUN(XLC_HEX);
            for (; ee + 8u <= chunk_end; ee += 8u) XLC_RUN(XLC_OCT);
            for (; ee + 4u <= chunk_end; ee += 4u) XLC_RUN(XLC_QUAD);
            for (; ee < chunk_end; ++ee) XLC_RUN(XLC_ENTRY);
#undef XLC_RUN
          }
        }
        if (e_stop > e) e = e_stop;
        // ---- the entry that holds the event (or the tail of the call): per-step checks, every lane for itself
        if (e < Emax) {
          if ((e & (XLC_RING / 2u - 1u)) == 0u && e >= XLC_RING) {
            const uint32_t q = (e - XLC_RING / 2u) >> 1;
            for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
              if (xl_lds_poll(a_n0) >= q && xl_lds_poll(a_n1) >= q && xl_lds_poll(a_n2) >= q) break;
              __builtin_amdgcn_s_sleep(1);
            }
          }
          const uint32_t m0 = e << XL_PH_SHIFT;
          if (m0 < K) xl_lds_post64(a_ring + (e & (XLC_RING - 1u)) * 64u * (uint32_t)sizeof(v2f), p);
          xl_lds_post(a_prod, e + 1u);
          for (uint32_t m = m0; m < m0 + XL_PH_STRIDE && m < K; ++m) {
            p = xl_nco_next_any(p, inc, bnd.flags);
            if (m + 1u == nb) {
              p = xl_nco_renorm(p);
              nb = xl_bnd_next(bnd, m + 1u);
            }
          }
          ++e;
        }
      }
      if (have) state_out[k.slot] = make_float2(p.x, p.y);  // (K == 0: untouched, xlating.c:58)
      if (stats) {
        cyc += clock64() - c0;
        const unsigned long long w1 = wall_clock64();
        ticks += w1 - w0;
        entries += Emax;
        if (call == 0u) wfirst = w0;
        if (lane == 0u && blockIdx.x < 1024u) {  // timeline of the launch: entry, per call start / end of the stepping, exit
          stats[8192u + 8u * blockIdx.x + 1u + 2u * call] = w0 - t_entry;
          stats[8192u + 8u * blockIdx.x + 2u + 2u * call] = w1 - t_entry;
        }
      }
      continue;
    }
    // ---- drainers: ring -> table, two entries (16 bytes) per client and store
    const uint32_t j = w - 1u;
    const uint32_t a_next = xl_lds_off(&s_next[j]);
    v4f *__restrict__ o4 = reinterpret_cast<v4f *>(o);
    for (uint32_t q = j; 2u * q < Emax; q += 3u) {
      const uint32_t need = 2u * q + 2u < Emax ? 2u * q + 2u : Emax;
      for (uint32_t spin = 0; spin < (1u << 24); ++spin) {
        if (xl_lds_poll(a_prod) >= need) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (2u * q < E) {
        v2f a, b2 = {0.0f, 0.0f};
        const uint32_t ra = a_ring + ((2u * q) & (XLC_RING - 1u)) * 64u * (uint32_t)sizeof(v2f);
        const uint32_t rb = a_ring + ((2u * q + 1u) & (XLC_RING - 1u)) * 64u * (uint32_t)sizeof(v2f);
        const bool two = 2u * q + 1u < E;
        asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(ra) : "memory");
        if (two) asm volatile("ds_read_b64 %0, %1" : "=v"(b2) : "v"(rb) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring slots have been read: they may be overwritten
        if (two) o4[q] = (v4f){a.x, a.y, b2.x, b2.y};
        else o[2u * q] = a;
      }
      xl_lds_post(a_next, q + 3u);
    }
    xl_lds_post(a_next, 0xFFFFFFFFu);
  }  // calls
  if (stats) __syncthreads();  // (tuning: the exit stamp is taken when every wave is through)
  if (stats && threadIdx.x == 0u) {
    if (blockIdx.x < 1024u) {
      stats[8192u + 8u * blockIdx.x] = t_entry;
      stats[8192u + 8u * blockIdx.x + 7u] = wall_clock64() - t_entry;
    }
    stats[4u * blockIdx.x] = cyc;
    stats[4u * blockIdx.x + 1u] = ticks;
    stats[4u * blockIdx.x + 2u] = entries;
    stats[4u * blockIdx.x + 3u] = wfirst;
  }
}

// Window staging: raw samples -> cf32 image in LDS.  Four independent loads per thread are issued before any is
// consumed (every workgroup of a launch stages at the same time and all its waves wait at the barrier, so this
// phase is pure latency: 12 dependent load->convert->write rounds measured 6 us of a 130 us launch).
template <int FMT>
XL_DEV void xl_stage_window(const XlFirArgs &a, const uint32_t zero_below, const uint32_t win0, const uint32_t wlen,
                            v2f *__restrict__ win) {
  const uint32_t bd = blockDim.x;
  for (uint32_t j0 = threadIdx.x; j0 < wlen; j0 += 4u * bd) {
    v2f v[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u * bd;
      const uint32_t s = win0 + j;
      const bool first = s < a.n0;
      ok[u] = j < wlen && s >= zero_below && (first || s - a.n0 < a.n1);
      const void *src = (first || !ok[u]) ? a.in0 : a.in1;  // in0 always holds >= 1 sample: safe dummy address
      const uint32_t idx = ok[u] ? (first ? s : s - a.n0) : 0u;
      v[u] = xl_sample(src, FMT, idx);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u * bd;
      if (j < wlen) win[j] = ok[u] ? v[u] : (v2f){0.0f, 0.0f};
    }
  }
}

// ------------------------------------------------------------------------------------------- the FIR kernel
// (A two-outputs-per-lane variant -- half the scalar tap traffic per FMA, which at ~3.7 B/clk/CU of missing s_load
// traffic sits right behind the VALU as the second ceiling -- was measured slower: it halves the resident waves.)
// SGPR cap 96: 64 hold one batch of taps; at <= 96 the CU admits 7 waves per SIMD instead of 6 (the allocation
// granule is 16 and 800 SGPRs serve a SIMD), which lets a 5-wave-per-workgroup launch keep 25 waves per CU.
template <int CT, int MODE, bool WIDE>
__global__ __launch_bounds__(64 * XL_NW_MAX) __attribute__((amdgpu_num_sgpr(96))) void xl_fir_kernel(const XlFirArgs a) {
  extern __shared__ __attribute__((aligned(16))) v2f xl_win[];
  // outputs per tile: 64 (one per lane), or fewer active lanes (a.ota = 32/16/8) for huge decimations
  const uint32_t OT = a.ota;

  // ---- NCO role: the first nco_blocks workgroups tabulate the NEXT block's phases (data independent float32
  // recurrence, xlating.c:70-73) while the rest of this launch filters the current block.  One launch per block
  // then does everything -- no side stream, no events between consecutive launches (they cost a 24 us gap).
  if (blockIdx.x < a.nco_blocks) {
    // a pure dependent chain that must not starve behind the FIR waves; flags bits 4-5: its priority (tuning)
    if (((a.flags >> 4) & 3u) == 3u) __builtin_amdgcn_s_setprio(3);
    else if (((a.flags >> 4) & 3u) == 2u) __builtin_amdgcn_s_setprio(2);
    else if (((a.flags >> 4) & 3u) == 1u) __builtin_amdgcn_s_setprio(1);
    const uint32_t nwave = threadIdx.x >> 6, nlane = threadIdx.x & 63u;
    if (nwave < a.nco_wpw && nlane < XL_NCO_LANES) {
      const unsigned long long t0 = a.trace ? wall_clock64() : 0ull;
      const uint32_t c = (blockIdx.x * a.nco_wpw + nwave) * XL_NCO_LANES + nlane;
      if (c < a.nco_nclients) {
        const XlNcoClient k = a.nco_clients[c];
        const XlBnd bnd = xl_nco_bnd(k, xl_grid_next(a.pos), 0xFFFFFFFFu);
        xl_nco_client_chain<false>(k, bnd, 0u, bnd.K, a.nco_state_in, a.nco_state_out, a.nco_tab);
      }
      if (a.trace && nlane == 0) {  // tuning: stamp the NCO-role wave (stamps 1, 2 stay 0 = "NCO role")
        unsigned long long *tn = a.trace + ((size_t)blockIdx.x * XL_NW_MAX + nwave) * 6;
        tn[0] = t0;
        tn[3] = wall_clock64();
        tn[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        tn[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
      }
    }
    return;
  }

  // block -> (group y, output tile x).  The work list is group-major (w = y * xtiles + x, so that all tiles of a
  // group -- which stream the same taps -- are neighbours) and is cut into 8 equal contiguous chunks, one per XCD
  // (block b runs on XCD b % 8 on this part): each XCD's L2 then holds the taps of ~1/8 of the groups, and every
  // XCD gets the same number of workgroups (whole groups per XCD left one XCD with 196 workgroups on 192 slots).
  // Enter at top priority: a workgroup still staging its window at the default priority 0 starves against
  // neighbours already in the tap loop at priority 3 (measured: staging up to 68 us instead of 4 us; those
  // workgroups then finish last and set the launch time).
  if (!(a.flags & 2u)) __builtin_amdgcn_s_setprio(3);
  const uint32_t b = blockIdx.x - a.nco_blocks;
  const unsigned long long t_entry = a.trace ? wall_clock64() : 0ull;

  // ---- raw-history roll, folded into this launch: hist_out = the last hist_units 2-byte units of [in0 | in1].
  // No workgroup of this launch reads hist_out (they read in0/in1), the next block's launch follows in stream order.
  const uint32_t roll_blocks = gridDim.x - a.nco_blocks < XL_ROLL_BLOCKS ? gridDim.x - a.nco_blocks : XL_ROLL_BLOCKS;  // small launches have fewer workgroups than XL_ROLL_BLOCKS
  if (a.hist_out != nullptr && b < roll_blocks) {
    const uint16_t *__restrict__ h0 = reinterpret_cast<const uint16_t *>(a.in0);
    const uint16_t *__restrict__ h1 = reinterpret_cast<const uint16_t *>(a.in1);
    uint16_t *__restrict__ ho = reinterpret_cast<uint16_t *>(a.hist_out);
    const uint32_t hu = a.hist_units, nu = a.block_units;
    for (uint32_t j = b * blockDim.x + threadIdx.x; j < hu; j += roll_blocks * blockDim.x) {
      const uint32_t sidx = nu + j;
      ho[j] = (sidx < hu) ? h0[sidx] : h1[sidx - hu];
    }
  }
  const uint32_t xcd = b & 7u, idx = b >> 3;
  const uint32_t total = a.ngroups * a.xtiles;
  const uint32_t per = total >> 3, extra = total & 7u;
  const uint32_t len = per + (xcd < extra ? 1u : 0u);
  if (idx >= len) return;
  const uint32_t wi = xcd * per + (xcd < extra ? xcd : extra) + idx;
  const uint32_t y = wi / a.xtiles;
  const uint32_t x = wi - y * a.xtiles;
  if (y >= a.ngroups) return;

  const cu32_p g = (cu32_p)(uintptr_t)(a.groups + y);  // XlGroup as dwords, scalar-loaded
  const uint32_t D = g[0], Tpad = g[2], ntiles = g[4];
  // per-call numbers of the group's class from its plan-time record and the stream position (xl_grid.h)
  const XlDyn d = a.explicit_dyn ? a.dyn1 : xl_grid_dyn(D, g[1], g[3], g[7], a.pos);
  const uint32_t K = d.K;
  const bool live = x * OT < K;
  if (!live && a.nco_lanes == 0u) return;

  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- NCO phases of this wave's outputs, before the window image takes the LDS.  The table holds every
  // XL_PH_STRIDE-th phase; lane (c, e) = (lane / G, lane % G), G = 64 / XL_PH_STRIDE, expands the phases of outputs
  // x*OT + e*XL_PH_STRIDE .. of client c of the tile into this wave's [client][64] slice of the LDS, then every lane
  // picks its own output's phase of each client into registers.  One dependent table load per lane at the START of
  // the kernel instead of CT of them in the epilogue.
  v2f phs[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) phs[c] = (v2f){1.0f, 0.0f};
  if (live) {
    if (w < ntiles) {
      constexpr uint32_t G = 64u / XL_PH_STRIDE;
      const uint32_t lane = threadIdx.x & 63u;
      v2f *__restrict__ pl = xl_win + w * (CT * 64u);
      const uint32_t *__restrict__ tg = reinterpret_cast<const uint32_t *>(a.groups + y) + 8u + w * XL_TILE_DWORDS;
      const uint32_t c = lane / G, e = lane % G;
      const uint32_t m0 = x * OT + e * XL_PH_STRIDE;
      if (c < tg[1] && e * XL_PH_STRIDE < OT && m0 < K) {
        const uint32_t off = tg[2 + c];
        const float2 ci = reinterpret_cast<const float2 *>(tg + 2 + XL_CT_MAX)[c];
        const uint32_t left = K - m0, span = OT < XL_PH_STRIDE ? OT : XL_PH_STRIDE;
        XlBnd bnd;
        bnd.j0 = d.j0, bnd.D = D, bnd.S = a.pos.S, bnd.G = a.explicit_dyn ? 1u : a.pos.G, bnd.K = K, bnd.flags = a.pos.pad;
        v2f *__restrict__ dst = pl + c * 64u + e * XL_PH_STRIDE;
        xl_phase_walk((reinterpret_cast<const v2f *>(a.phtab) + (off >> XL_PH_SHIFT))[m0 >> XL_PH_SHIFT], m0,
                      left < span ? left : span, (v2f){ci.x, ci.y}, bnd, [&](uint32_t i, v2f ph) { dst[i] = ph; });
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) phs[cc] = pl[cc * 64u + lane];
    }
    __syncthreads();  // the phases are in registers: the LDS is free for the window image

    // ---- stage the window image: samples [win0, win0 + (OT-1)*D + Tpad) of the stream [in0 | in1], as cf32
    const uint32_t win0 = d.base + x * OT * D;
    const uint32_t wlen = (OT - 1u) * D + Tpad;
    if (a.fmt == XLF_CU8) xl_stage_window<XLF_CU8>(a, d.zero_below, win0, wlen, xl_win);
    else if (a.fmt == XLF_CS8) xl_stage_window<XLF_CS8>(a, d.zero_below, win0, wlen, xl_win);
    else if (a.fmt == XLF_CS16) xl_stage_window<XLF_CS16>(a, d.zero_below, win0, wlen, xl_win);
    else xl_stage_window<XLF_CF32>(a, d.zero_below, win0, wlen, xl_win);
  }
  __syncthreads();

  if (w >= ntiles) {
    // ---- NCO rider: a wave this group has no tile for tabulates the NEXT block's phases of nco_lanes clients.
    // It shares its SIMD with one FIR wave fewer than the others do, which is about what the chain costs the SIMD
    // (3 packed ops per step at top priority ~ a third of the issue slots for ~57 us ~ one FIR wave): workgroups
    // of their own for the role put it ON TOP of a full set of FIR waves and those finished ~18 us late.
    const uint32_t nlane = threadIdx.x & 63u;
    const uint32_t slot = (g[6] + (w - ntiles)) * a.xtiles + x;
    if (a.nco_lanes != 0u && slot < a.nco_slots && nlane < a.nco_lanes) {
      __builtin_amdgcn_s_setprio(3);  // a pure dependent chain: must not starve behind the FIR waves
      const unsigned long long t0 = a.trace ? wall_clock64() : 0ull;
      const uint32_t c = slot * a.nco_lanes + nlane;
      if (c < a.nco_nclients) {
        const XlNcoClient k = a.nco_clients[c];
        const XlBnd bnd = xl_nco_bnd(k, xl_grid_next(a.pos), 0xFFFFFFFFu);
        xl_nco_client_chain<false>(k, bnd, 0u, bnd.K, a.nco_state_in, a.nco_state_out, a.nco_tab);
      }
      if (a.trace && nlane == 0) {
        unsigned long long *tn = a.trace + ((size_t)blockIdx.x * XL_NW_MAX + w) * 6;
        tn[0] = t0;
        tn[3] = wall_clock64();
        tn[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        tn[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
      }
    }
    return;
  }
  if (!live) return;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long *tr = a.trace ? a.trace + ((size_t)blockIdx.x * XL_NW_MAX + w) * 6 : nullptr;
  if (tr && lane == 0) {
    tr[0] = t_entry;
    tr[1] = wall_clock64();
    tr[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    tr[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  }
  const cu32_p t = g + 8 + w * XL_TILE_DWORDS;  // XlTile of this wave
  const uint32_t ncl = t[1];
  const cfloat_p tp = (cfloat_p)(uintptr_t)(a.taps + t[0]);

  XlAcc<MODE> acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c].clear();

  constexpr int STEP = (CT == 9 || CT == 10) ? 6 : 4;  // taps per iteration (xl_tap_step): ~60 tap SGPRs in flight
  // idle lanes (lane >= OT when ota < 64) alias lane 0's window so that their reads stay inside the image
  const v2f *lp = xl_win + (lane >= OT ? 0u : lane) * D;
  // The tap loop runs in segments of falling wave priority.  The SIMD arbiter otherwise favours the oldest wave,
  // so co-resident waves -- which all have the same work -- finish one after another and the last ones run alone,
  // latency-bound (measured: identical waves ending between 57 and 142 us of a 144 us launch).  With priority =
  // remaining work a wave that gets ahead yields to those behind, all waves of a SIMD finish together and the
  // VALU stays fed to the end.
  // one quarter of the tap loop; WIDE (even D: lane*D and 64*D even) reads 16-byte aligned pairs of samples
#define XL_TAP_LOOP(I0, I1)                                                                       \
  for (uint32_t i = (I0); i < (I1); i += STEP) {                                                  \
    v2f xs[STEP];                                                                                 \
    if (WIDE) {                                                                                   \
      _Pragma("unroll") for (int u = 0; u < STEP; u += 2) {                                       \
        const v4f q = *reinterpret_cast<const v4f *>(lp + i + u);                                 \
        xs[u] = (v2f){q.x, q.y};                                                                  \
        xs[u + 1] = (v2f){q.z, q.w};                                                              \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int u = 0; u < STEP; ++u) xs[u] = lp[i + u];                         \
    }                                                                                             \
    const cfloat_p tq = tp + (size_t)i * (2 * CT);                                                \
    _Pragma("unroll") for (int u = 0; u < STEP; ++u) {                                            \
      _Pragma("unroll") for (int c = 0; c < CT; ++c)                                              \
        acc[c].mac(xs[u], tq[(u * CT + c) * 2], tq[(u * CT + c) * 2 + 1]);                        \
    }                                                                                             \
  }
  if (a.flags & 2u) {  // flat priority (multi-round launches, tuning)
    XL_TAP_LOOP(0u, Tpad)
  } else {
    // priority 3 is reserved for the short latency-bound phases (staging above, epilogue below, NCO role): a
    // workgroup that is still staging must outrank its neighbours' tap loops or it starves (measured: the last
    // dispatched workgroups of every XCD staged for 56 us instead of 4 us and then set the launch time).
    // The tap loop itself runs at 2, 1, 0 over [0, 1/2), [1/2, 7/8), [7/8, 1] of the taps: remaining-work priority,
    // short last segment = tight finish (equal thirds and a four-level variant measured slower).
    const uint32_t steps = Tpad / STEP;
    const uint32_t b1 = steps / 2, b2 = (7 * steps) / 8;
    const uint32_t e1 = b1 * STEP < Tpad ? b1 * STEP : Tpad;
    const uint32_t e2 = b2 * STEP < Tpad ? b2 * STEP : Tpad;
    __builtin_amdgcn_s_setprio(2);
    XL_TAP_LOOP(0u, e1)
    __builtin_amdgcn_s_setprio(1);
    XL_TAP_LOOP(e1, e2)
    __builtin_amdgcn_s_setprio(0);
    XL_TAP_LOOP(e2, Tpad)
    __builtin_amdgcn_s_setprio(3);
  }
#undef XL_TAP_LOOP

  if (tr && lane == 0) tr[2] = wall_clock64();
  // ---- epilogue: derotate with the NCO phase (expanded in the prologue) and store (coalesced: lanes = consecutive outputs)
  v2f *__restrict__ out = reinterpret_cast<v2f *>(a.out);
  {
    const uint32_t m = x * OT + lane;
    if (m < K && lane < OT) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if ((uint32_t)c < ncl) out[t[2 + c] + m] = xl_rotate<MODE>(acc[c].value(), phs[c]);
      }
    }
  }
  if (tr && lane == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    tr[3] = wall_clock64();
  }
}

size_t xl_fir_lds_bytes_ota(uint32_t D, uint32_t Tpad, uint32_t ota) {
  return ((size_t)(ota - 1u) * D + Tpad) * sizeof(v2f);
}

uint32_t xl_fir_pick_ota(uint32_t D, uint32_t Tpad, size_t budget) {
  for (uint32_t ota = 64; ota >= 8; ota >>= 1)
    if (xl_fir_lds_bytes_ota(D, Tpad, ota) <= budget) return ota;
  return 0;
}

template <int CT, int MODE, bool WIDE>
static hipError_t xl_fir_go2(int nw, const XlFirArgs &a, size_t lds, hipStream_t s) {
  static bool attr_done = false;  // per instantiation; benign race (idempotent)
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&xl_fir_kernel<CT, MODE, WIDE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const uint32_t nblocks = a.nco_blocks + 8u * ((a.ngroups * a.xtiles + 7u) / 8u);
  if (nblocks == 0) return hipSuccess;
  // the prologue expands the NCO phases through the LDS: [wave][client][64] before the window image is staged
  const size_t need = (size_t)nw * CT * 64u * sizeof(float2);
  if (lds < need) lds = need;
  hipLaunchKernelGGL((xl_fir_kernel<CT, MODE, WIDE>), dim3(nblocks), dim3(64 * nw), lds, s, a);
  return hipGetLastError();
}

// a.flags bit 0 set by the caller = every group of the launch has an even decimation -> 16-byte LDS reads
template <int CT, int MODE>
static hipError_t xl_fir_go(int nw, const XlFirArgs &a, size_t lds, hipStream_t s) {
  return (a.flags & 1u) ? xl_fir_go2<CT, MODE, true>(nw, a, lds, s) : xl_fir_go2<CT, MODE, false>(nw, a, lds, s);
}

// a.xtiles must be ceil(max K / a.ota)
hipError_t xl_launch_fir(int ct, int mode, int nw, const XlFirArgs &a, size_t lds, hipStream_t s) {
  if (lds > 160 * 1024 || nw < 1 || nw > XL_NW_MAX) return hipErrorInvalidValue;
  if (a.ota != 64 && a.ota != 32 && a.ota != 16 && a.ota != 8) return hipErrorInvalidValue;
  switch (ct * 2 + (mode ? 1 : 0)) {
    case 2: return xl_fir_go<1, 0>(nw, a, lds, s);
    case 3: return xl_fir_go<1, 1>(nw, a, lds, s);
    case 4: return xl_fir_go<2, 0>(nw, a, lds, s);
    case 5: return xl_fir_go<2, 1>(nw, a, lds, s);
    case 8: return xl_fir_go<4, 0>(nw, a, lds, s);
    case 9: return xl_fir_go<4, 1>(nw, a, lds, s);
    case 16: return xl_fir_go<8, 0>(nw, a, lds, s);
    case 17: return xl_fir_go<8, 1>(nw, a, lds, s);
    case 18: return xl_fir_go<9, 0>(nw, a, lds, s);
    case 19: return xl_fir_go<9, 1>(nw, a, lds, s);
    case 20: return xl_fir_go<10, 0>(nw, a, lds, s);
    case 21: return xl_fir_go<10, 1>(nw, a, lds, s);
    case 24: return xl_fir_go<12, 0>(nw, a, lds, s);
    case 25: return xl_fir_go<12, 1>(nw, a, lds, s);
    default: return hipErrorInvalidValue;
  }
}

// (NCO phase-table code: see above the FIR kernel)
hipError_t xl_launch_nco_table(const XlNcoClient *clients, uint32_t nclients, const float2 *state_in,
                               float2 *state_out, float2 *phtab, XlPos pos, uint32_t explicit_K, uint32_t prio,
                               hipStream_t s) {
  if (nclients == 0) return hipSuccess;
  const uint32_t lanes = XL_NCO_LANES;
  hipLaunchKernelGGL(xl_nco_table_kernel, dim3((nclients + lanes - 1) / lanes), dim3(64), 0, s, clients, nclients,
                     state_in, state_out, phtab, pos, explicit_K, prio, lanes);
  return hipGetLastError();
}

// `done` (optional): recorded with the kernel's own completion signal (hipExtLaunchKernelGGL's stop event) -- one packet
// on the queue instead of the kernel plus a separate event record.
hipError_t xl_launch_nco_chain(const XlNcoClient *clients, uint32_t nclients, const float2 *state_in,
                               const XlChainCalls &calls, XlPos pos, unsigned long long *stats, hipStream_t s,
                               hipEvent_t done) {
  if (calls.n < 1u || calls.n > XL_CHAIN_MAXCALLS) return hipErrorInvalidValue;
  if (nclients == 0) return done ? hipEventRecord(done, s) : hipSuccess;
  if (done)
    hipExtLaunchKernelGGL(xl_nco_chain_kernel, dim3((nclients + 63u) / 64u), dim3(256), 0, s, nullptr, done, 0, clients,
                          nclients, state_in, calls, pos, stats);
  else
    hipLaunchKernelGGL(xl_nco_chain_kernel, dim3((nclients + 63u) / 64u), dim3(256), 0, s, clients, nclients, state_in,
                       calls, pos, stats);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------- single-filter helpers
__global__ void xl_convert_cf32_kernel(const void *__restrict__ raw, int fmt, uint32_t n, v2f *__restrict__ dst) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    dst[i] = xl_sample(raw, fmt, i);
}

hipError_t xl_launch_convert_cf32(const void *raw, int fmt, uint32_t nsamples, float2 *dst, hipStream_t s) {
  if (nsamples == 0) return hipSuccess;
  const uint32_t blocks = (nsamples + 255) / 256 < 2048 ? (nsamples + 255) / 256 : 2048;
  hipLaunchKernelGGL(xl_convert_cf32_kernel, dim3(blocks), dim3(256), 0, s, raw, fmt, nsamples,
                     reinterpret_cast<v2f *>(dst));
  return hipGetLastError();
}

// xlating.c:418 ((u8 - 128) << 8), :425 (s8 << 8), :432 (copy); per scalar element
__global__ void xl_convert_q15_kernel(const void *__restrict__ raw, int fmt, uint32_t n, int16_t *__restrict__ dst) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int32_t v;
    if (fmt == XLF_CU8) {
      v = ((int32_t) reinterpret_cast<const uint8_t *>(raw)[i] - 128) * 256;
    } else if (fmt == XLF_CS8) {
      v = (int32_t) reinterpret_cast<const int8_t *>(raw)[i] * 256;
    } else {
      v = reinterpret_cast<const int16_t *>(raw)[i];
    }
    dst[i] = (int16_t)v;
  }
}

hipError_t xl_launch_convert_q15(const void *raw, int fmt, uint32_t nelems, int16_t *dst, hipStream_t s) {
  if (nelems == 0) return hipSuccess;
  const uint32_t blocks = (nelems + 255) / 256 < 2048 ? (nelems + 255) / 256 : 2048;
  hipLaunchKernelGGL(xl_convert_q15_kernel, dim3(blocks), dim3(256), 0, s, raw, fmt, nelems, dst);
  return hipGetLastError();
}

// memmove towards lower addresses (xlating.c:76-79), dword granularity, ONE workgroup: chunk k is read
// completely (barrier) before it is written, and chunks go in increasing order, so overlap is safe.
__global__ __launch_bounds__(1024) void xl_move_down_kernel(uint32_t *buf, uint32_t from_dw, uint32_t count_dw) {
  for (uint32_t base = 0; base < count_dw; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    uint32_t v = 0;
    if (i < count_dw) v = buf[from_dw + i];
    __syncthreads();
    if (i < count_dw) buf[i] = v;
    __syncthreads();
  }
}

hipError_t xl_launch_move_down(void *buf, uint32_t from, uint32_t count, uint32_t elem_bytes, hipStream_t s) {
  if (count == 0 || from == 0) return hipSuccess;
  const uint32_t k = elem_bytes / 4;
  hipLaunchKernelGGL(xl_move_down_kernel, dim3(1), dim3(1024), 0, s, reinterpret_cast<uint32_t *>(buf), from * k,
                     count * k);
  return hipGetLastError();
}

// batch engine: roll the raw history (the last h samples of [hist | block]) into the other history buffer
__global__ void xl_update_history_kernel(const uint16_t *__restrict__ hist, const uint16_t *__restrict__ block,
                                         uint32_t h_u, uint32_t n_u, uint16_t *__restrict__ out) {
  // units of 2 bytes; out[j] = concat(hist, block)[n_u + j], j < h_u
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < h_u; j += gridDim.x * blockDim.x) {
    const uint32_t s = n_u + j;
    out[j] = (s < h_u) ? hist[s] : block[s - h_u];
  }
}

hipError_t xl_launch_update_history(const void *hist, const void *block, uint32_t h, uint32_t n, uint32_t bps,
                                    void *new_hist, hipStream_t s) {
  if (h == 0) return hipSuccess;
  const uint32_t u = bps / 2;
  const uint32_t hu = h * u, nu = n * u;
  const uint32_t blocks = (hu + 255) / 256 < 256 ? (hu + 255) / 256 : 256;
  hipLaunchKernelGGL(xl_update_history_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(hist),
                     reinterpret_cast<const uint16_t *>(block), hu, nu, reinterpret_cast<uint16_t *>(new_hist));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------- Q15 family
XL_DEV int32_t xl_sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }  // xlating.c:85-90

// xlating.c:126-129: truncating Q15 phase recurrence, never renormalised.  One thread (one filter).
// One step of the Q15 phase recurrence (xlating.c:126-129): (pr + j pi) * (ir + j ii) >> 15, truncating, saturated --
// two packed dot products (v_dot2_i32_i16: p . (ir, -ii) and p . (ii, ir); |sum| <= 2 * 32767^2 < 2^31) instead of four
// multiplies and two adds.  p, a, b: int16 pairs in one register (low = real part).
XL_DEV uint32_t xl_q15_step(const uint32_t p, const uint32_t a, const uint32_t b) {
  typedef short s2 __attribute__((ext_vector_type(2)));
  const int32_t tr = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, p), __builtin_bit_cast(s2, a), 0, false);
  const int32_t ti = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, p), __builtin_bit_cast(s2, b), 0, false);
  // >> 15, then v_cvt_pk_i16_i32: both halves saturated to int16 and packed by ONE instruction (the step is a dependent
  // chain on a lone lane: every instruction costs its ~6 issue cycles -- five instead of eight)
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(tr >> 15, ti >> 15));
}

// Single filter: every XL_PH_STRIDE-th phase of a call's K outputs into tab, the phase after the call into state_out
// (state_in / state_out may alias).  A dependent chain on one lane (~20 cycles per step); a store per step -- the first
// version -- blocked the lane ~60 ns each and made this kernel 190 of the Q15 call's 275 us.
__global__ void xl_nco_table_q15_kernel(int32_t ir, int32_t ii, const short2 *state_in, short2 *state_out,
                                        uint32_t *__restrict__ tab, uint32_t K) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t a = ((uint32_t)ir & 0xFFFFu) | ((uint32_t)(-ii) << 16), b = ((uint32_t)ii & 0xFFFFu) | ((uint32_t)ir << 16);
  uint32_t p = ((uint32_t)(uint16_t)state_in->x) | ((uint32_t)(uint16_t)state_in->y << 16);
  uint32_t m = 0;
  for (; m + XL_PH_STRIDE <= K; m += XL_PH_STRIDE) {
    tab[m >> XL_PH_SHIFT] = p;
#pragma unroll
    for (uint32_t i = 0; i < XL_PH_STRIDE; ++i) p = xl_q15_step(p, a, b);
  }
  if (m < K) tab[m >> XL_PH_SHIFT] = p;
  for (; m < K; ++m) p = xl_q15_step(p, a, b);
  *state_out = make_short2((short)(p & 0xFFFFu), (short)(p >> 16));
}

hipError_t xl_launch_nco_table_q15(int16_t incr_re, int16_t incr_im, const short2 *state_in, short2 *state_out, short2 *phtab,
                                   uint32_t K, hipStream_t s) {
  if (K == 0) return hipSuccess;
  hipLaunchKernelGGL(xl_nco_table_q15_kernel, dim3(1), dim3(64), 0, s, (int32_t)incr_re, (int32_t)incr_im, state_in, state_out,
                     reinterpret_cast<uint32_t *>(phtab), K);
  return hipGetLastError();
}

// xlating.c:100-124: int16 x int16 products accumulated in int64, >>15, saturate, rotate by the Q15 phase.
// Lane = output; taps wave-uniform (scalar loads of packed int16 pairs); window read straight from L2, two samples per
// load.  Per sample two packed dot products ((xr, xi) . (hr, -hi) and (xr, xi) . (hi, hr): each < 2^31) added into
// 64-bit sums.  The lane's phase: the tabulated one at m rounded down to the stride, stepped m mod 16 times.
__global__ __launch_bounds__(256) void xl_fir_q15_kernel(const uint32_t *__restrict__ work, const uint32_t *taps,
                                                         uint32_t T, uint32_t D, uint32_t K, int32_t ir, int32_t ii,
                                                         const uint32_t *__restrict__ phtab, short2 *__restrict__ out) {
  typedef short s2 __attribute__((ext_vector_type(2)));
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= K) return;
  const uint32_t *__restrict__ w = work + (size_t)m * D;
  const cu32_p tq = (cu32_p)(uintptr_t)taps;
  long long sr = 0, si = 0;
  auto mac = [&](const uint32_t xv, const uint32_t hv) {
    // (hr, -hi): -hi of hi = -32768 would not fit an int16 -- the taps are (int16)(x * 32768) of |x| < 1 values scaled by a
    // window, the reference's designer never produces -32768; guarded all the same by doing that product in 32 bits)
    const int32_t hr = (int16_t)(hv & 0xFFFFu), hi = (int32_t)hv >> 16;
    const int32_t xr = (int16_t)(xv & 0xFFFFu), xi = (int32_t)xv >> 16;
    const uint32_t swp = (hv >> 16) | (hv << 16);  // (hi, hr)
    si += (long long)__builtin_amdgcn_sdot2(__builtin_bit_cast(s2, xv), __builtin_bit_cast(s2, swp), 0, false);
    sr += (long long)(xr * hr - xi * hi);
  };
  uint32_t i = 0;
  if ((((uintptr_t)w) & 7u) == 0u) {
    for (; i + 2u <= T; i += 2u) {
      const uint2 xv = *reinterpret_cast<const uint2 *>(w + i);
      mac(xv.x, tq[i]);
      mac(xv.y, tq[i + 1u]);
    }
  }
  for (; i < T; ++i) mac(w[i], tq[i]);
  const int32_t ar = xl_sat16((int32_t)(sr >> 15));
  const int32_t ai = xl_sat16((int32_t)(si >> 15));
  const uint32_t a = ((uint32_t)ir & 0xFFFFu) | ((uint32_t)(-ii) << 16), b = ((uint32_t)ii & 0xFFFFu) | ((uint32_t)ir << 16);
  uint32_t p = phtab[m >> XL_PH_SHIFT];
  for (uint32_t k = m & (XL_PH_STRIDE - 1u); k > 0u; --k) p = xl_q15_step(p, a, b);
  const int32_t pr = (int16_t)(p & 0xFFFFu), pi = (int32_t)p >> 16;
  const int32_t tr = ar * pr - ai * pi;
  const int32_t ti = ar * pi + ai * pr;
  out[m] = make_short2((short)xl_sat16(tr >> 15), (short)xl_sat16(ti >> 15));
}

hipError_t xl_launch_fir_q15(const short2 *work, const short2 *taps, uint32_t T, uint32_t D, uint32_t K, int16_t incr_re,
                             int16_t incr_im, const short2 *phtab, short2 *out, hipStream_t s) {
  if (K == 0) return hipSuccess;
  hipLaunchKernelGGL(xl_fir_q15_kernel, dim3((K + 63) / 64), dim3(64), 0, s, reinterpret_cast<const uint32_t *>(work),
                     reinterpret_cast<const uint32_t *>(taps), T, D, K, (int32_t)incr_re, (int32_t)incr_im,
                     reinterpret_cast<const uint32_t *>(phtab), out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------- Q15 family, batched
// xlating.c:92-140 for every client of the engine.  The reference accumulates int16 x int16 products in int64; here
// the same integers are carried in float64 FMAs (a product is < 2^30, a sum of 2 T of them < 2^53: every operation is
// exact, so the result IS the integer sum) -- v_fma_f64 runs at half the FP32 rate, an int64 multiply-add at a quarter
// or less.  Window image: the block's samples as Q15 integers held in float32 (exact), shared by the 4 tiles of a group;
// taps: wave-uniform doubles through scalar loads, SGPR operands of the FMAs, like the float kernel.
XL_DEV v2f xl_sample_q15(const void *__restrict__ p, int fmt, uint32_t i) {
  v2f r;
  if (fmt == XLF_CU8) {  // xlating.c:418: ((int16_t) u8 - 128) << 8
    const uint32_t v = reinterpret_cast<const uint16_t *>(p)[i];
    r.x = (float)(((int32_t)(v & 0xFFu) - 128) * 256);
    r.y = (float)(((int32_t)(v >> 8) - 128) * 256);
  } else if (fmt == XLF_CS8) {  // :425: s8 << 8
    const int32_t v = reinterpret_cast<const int16_t *>(p)[i];
    r.x = (float)((int32_t)(int8_t)(v & 0xFF) * 256);
    r.y = (float)((v >> 8) * 256);
  } else {  // :432: the int16 samples as they are
    const int32_t v = reinterpret_cast<const int32_t *>(p)[i];
    r.x = (float)(int32_t)(int16_t)(v & 0xFFFF);
    r.y = (float)(v >> 16);
  }
  return r;
}

// one lane per client: the truncating Q15 phase recurrence (xlating.c:126-129), every XL_PH_STRIDE-th phase tabulated
__global__ __launch_bounds__(64) void xl_nco_q15_batch_kernel(const XlNcoClient *__restrict__ cl, const uint32_t *__restrict__ qinc,
                                                              uint32_t n, short2 *qstate, short2 *__restrict__ tab, const XlPos pos) {
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  if (c >= n) return;
  const XlNcoClient k = cl[c];
  const uint32_t K = xl_nco_bnd(k, pos, 0xFFFFFFFFu).K;
  const int32_t ir = (int16_t)(qinc[c] & 0xFFFFu), ii = (int16_t)(qinc[c] >> 16);
  int32_t pr = qstate[k.slot].x, pi = qstate[k.slot].y;
  short2 *__restrict__ o = tab + (k.out_off >> XL_PH_SHIFT);
  for (uint32_t m = 0; m < K; ++m) {
    if ((m & (XL_PH_STRIDE - 1u)) == 0u) o[m >> XL_PH_SHIFT] = make_short2((short)pr, (short)pi);
    const int32_t tr = pr * ir - pi * ii, ti = pr * ii + pi * ir;
    pr = xl_sat16(tr >> 15);
    pi = xl_sat16(ti >> 15);
  }
  qstate[k.slot] = make_short2((short)pr, (short)pi);
}

hipError_t xl_launch_nco_q15_batch(const XlNcoClient *clients, const uint32_t *qinc, uint32_t nclients, short2 *qstate,
                                   short2 *qphtab, XlPos pos, hipStream_t s) {
  if (nclients == 0) return hipSuccess;
  hipLaunchKernelGGL(xl_nco_q15_batch_kernel, dim3((nclients + 63u) / 64u), dim3(64), 0, s, clients, qinc, nclients, qstate,
                     qphtab, pos);
  return hipGetLastError();
}

typedef const double __attribute__((address_space(4))) *cdouble_p;

template <int CT>
__global__ __launch_bounds__(64 * XL_NW_MAX) void xl_fir_q15_batch_kernel(const XlFirArgs a, const double *__restrict__ qtaps,
                                                                          const short2 *__restrict__ qphtab) {
  extern __shared__ __attribute__((aligned(16))) v2f xl_qwin[];
  const uint32_t OT = a.ota;
  const uint32_t b = blockIdx.x;
  // raw-history roll, as in xl_fir_kernel
  const uint32_t roll_blocks = gridDim.x < XL_ROLL_BLOCKS ? gridDim.x : XL_ROLL_BLOCKS;
  if (a.hist_out != nullptr && b < roll_blocks) {
    const uint16_t *__restrict__ h0 = reinterpret_cast<const uint16_t *>(a.in0);
    const uint16_t *__restrict__ h1 = reinterpret_cast<const uint16_t *>(a.in1);
    uint16_t *__restrict__ ho = reinterpret_cast<uint16_t *>(a.hist_out);
    for (uint32_t j = b * blockDim.x + threadIdx.x; j < a.hist_units; j += roll_blocks * blockDim.x) {
      const uint32_t sidx = a.block_units + j;
      ho[j] = (sidx < a.hist_units) ? h0[sidx] : h1[sidx - a.hist_units];
    }
  }
  const uint32_t y = b / a.xtiles, x = b - y * a.xtiles;
  if (y >= a.ngroups) return;
  const cu32_p g = (cu32_p)(uintptr_t)(a.groups + y);
  const uint32_t D = g[0], T = g[1], Tpad = g[2], ntiles = g[4];
  const XlDyn d = xl_grid_dyn(D, T, g[3], g[7], a.pos);
  const uint32_t K = d.K;
  if (x * OT >= K) return;
  {
    const uint32_t win0 = d.base + x * OT * D, wlen = (OT - 1u) * D + Tpad;
    for (uint32_t j = threadIdx.x; j < wlen; j += blockDim.x) {
      const uint32_t sidx = win0 + j;
      const bool first = sidx < a.n0;
      const bool ok = sidx >= d.zero_below && (first || sidx - a.n0 < a.n1);
      const v2f v = xl_sample_q15((first || !ok) ? a.in0 : a.in1, a.fmt, ok ? (first ? sidx : sidx - a.n0) : 0u);
      xl_qwin[j] = ok ? v : (v2f){0.0f, 0.0f};
    }
  }
  __syncthreads();
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (w >= ntiles) return;
  const cu32_p t = g + 8 + w * XL_TILE_DWORDS;
  const uint32_t ncl = t[1];
  const cdouble_p tq = (cdouble_p)(uintptr_t)(qtaps + (size_t)t[0] * 2u);
  double accr[CT], acci[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) accr[c] = acci[c] = 0.0;
  const v2f *lp = xl_qwin + (lane >= OT ? 0u : lane) * D;
  for (uint32_t i = 0; i < T; ++i) {
    const v2f xs = lp[i];
    const double xr = (double)xs.x, xi = (double)xs.y;
    const cdouble_p h = tq + (size_t)i * (2 * CT);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const double hr = h[2 * c], hi = h[2 * c + 1];
      accr[c] = __builtin_fma(xr, hr, accr[c]);  // temp_real += ar * br - ai * bi   (xlating.c:114)
      accr[c] = __builtin_fma(-xi, hi, accr[c]);
      acci[c] = __builtin_fma(xr, hi, acci[c]);  // temp_imag += ar * bi + ai * br   (:115)
      acci[c] = __builtin_fma(xi, hr, acci[c]);
    }
  }
  const uint32_t m = x * OT + lane;
  if (m >= K || lane >= OT) return;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if ((uint32_t)c >= ncl) continue;
    const uint32_t off = t[2 + c], qi = t[2 + 3 * XL_CT_MAX + c];
    // >> 15 of the int64 sums (arithmetic shift = floor), saturated (xlating.c:118-119)
    const int32_t ar = xl_sat16((int32_t)__builtin_floor(accr[c] * (1.0 / 32768.0)));
    const int32_t ai = xl_sat16((int32_t)__builtin_floor(acci[c] * (1.0 / 32768.0)));
    // this output's phase: the tabulated one below it, stepped m % XL_PH_STRIDE times (:126-129)
    const short2 p0 = qphtab[(off >> XL_PH_SHIFT) + (m >> XL_PH_SHIFT)];
    const int32_t ir = (int16_t)(qi & 0xFFFFu), ii = (int16_t)(qi >> 16);
    int32_t pr = p0.x, pi = p0.y;
    for (uint32_t j = m & (XL_PH_STRIDE - 1u); j > 0u; --j) {
      const int32_t tr = pr * ir - pi * ii, ti = pr * ii + pi * ir;
      pr = xl_sat16(tr >> 15);
      pi = xl_sat16(ti >> 15);
    }
    const int32_t orr = ar * pr - ai * pi, oi = ar * pi + ai * pr;  // xlating.c:121-124
    reinterpret_cast<short2 *>(a.out + off)[m] = make_short2((short)xl_sat16(orr >> 15), (short)xl_sat16(oi >> 15));
  }
}

template <int CT>
static hipError_t xl_fir_q15_go(int nw, const XlFirArgs &a, const double *qtaps, const short2 *qphtab, size_t lds, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&xl_fir_q15_batch_kernel<CT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const uint32_t nblocks = a.ngroups * a.xtiles;
  if (nblocks == 0) return hipSuccess;
  hipLaunchKernelGGL((xl_fir_q15_batch_kernel<CT>), dim3(nblocks), dim3(64 * nw), lds, s, a, qtaps, qphtab);
  return hipGetLastError();
}

hipError_t xl_launch_fir_q15_batch(int ct, int nw, const XlFirArgs &a, const double *qtaps, const short2 *qphtab,
                                   size_t lds, hipStream_t s) {
  if (lds > 160 * 1024 || nw < 1 || nw > XL_NW_MAX) return hipErrorInvalidValue;
  switch (ct) {
    case 1: return xl_fir_q15_go<1>(nw, a, qtaps, qphtab, lds, s);
    case 2: return xl_fir_q15_go<2>(nw, a, qtaps, qphtab, lds, s);
    case 4: return xl_fir_q15_go<4>(nw, a, qtaps, qphtab, lds, s);
    case 8: return xl_fir_q15_go<8>(nw, a, qtaps, qphtab, lds, s);
    case 9: return xl_fir_q15_go<9>(nw, a, qtaps, qphtab, lds, s);
    case 10: return xl_fir_q15_go<10>(nw, a, qtaps, qphtab, lds, s);
    case 12: return xl_fir_q15_go<12>(nw, a, qtaps, qphtab, lds, s);
    default: return hipErrorInvalidValue;
  }
}
