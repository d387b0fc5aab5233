// xl_device.h -- device-side data layout shared by the kernels (xl_kernels.hip) and the host engines
// (xl_filter.cpp: the xlating.h drop-in; xl_batch.cpp: the batched fan-out engine).
//
// Vocabulary (follows the reference, src/xlating.c):
//   sample   one complex IQ sample of the input stream (cu8/cs8: 2 B, cs16: 4 B, cf32: 8 B)
//   window   the T consecutive samples one output is computed from (xlating.c:61-69)
//   client   one xlating filter = (decimation D, T reversed band-pass taps, NCO phase + increment)
//   tile     up to CT clients that share (D, T, window grid) -> one wavefront's work, taps interleaved
//            [tap i][client c] so that one scalar load fetches tap i of every client of the tile
//   group    up to NW (<= XL_NW_MAX) tiles of the same class -> one workgroup; its waves share one LDS window image
//   class    all groups sharing (D, T, stream offset mod D, valid-history length).  A class's per-call numbers
//            (window origin, output count) follow from its plan-time record and the stream position of the call
//            (XlPos, 16 bytes of kernel arguments): any number of classes, nothing uploaded per call (xl_grid.h).
//   call     G >= 1 consecutive blocks of S samples handled by one set of launches ("group" of blocks); the NCO
//            phase is renormalised at every block end exactly as G successive reference calls would.
#ifndef XL_DEVICE_H_
#define XL_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xl_grid.h"

// The kernels are hand-written for CDNA3/4 wave64 targets: inline `s_waitcnt lgkmcnt(0); s_barrier` LDS-only barriers (gfx9 wait-counter
// encodings, in-order LDS), v_pk_*_f32 with op_sel, v_mfma_f32_32x32x16_f16 / 32x32x2_f32 operand maps, CU-masked streams of 8 XCDs.
// Nothing here was written for, or tested on, another target: refuse it instead of mis-compiling quietly.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "sdr-server_amd kernels target gfx950 (MI355X; gfx942 builds for comparison only): build with --offload-arch=gfx950"
#endif

#define XL_NW_MAX 4      // waves (tiles) per workgroup: one per SIMD (5..7 measured slower: unbalanced SIMDs; 8 ties)
#define XL_NW_DEFAULT 4
#define XL_CT_MAX 12     // most clients per tile (register-tile height: 1, 2, 4, 8, 9, 10 or 12)
#define XL_TAP_UNROLL 4  // taps per inner-loop step for heights <= 8 and 12; heights 9, 10 step by 6 (xl_tap_step)
#define XL_ROLL_BLOCKS 16  // workgroups of a FIR launch that also roll the raw history

enum { XLF_CU8 = 0, XLF_CS8 = 1, XLF_CS16 = 2, XLF_CF32 = 3 };

#define XL_PH_STRIDE 16u  // the NCO phase table holds every 16th phase: entry (out_off + m) / 16 = phase of output m,
#define XL_PH_SHIFT 4u    // m = 0 mod 16; the consumers expand the phases in between themselves with the same three IEEE
                          // operations (bit-identical), cooperatively through LDS.  Why: the recurrence is a dependent
                          // chain that caps the block rate, and on a memory-saturated chip every table store blocks
                          // the chain wave ~350 ns, while the bare chain is indifferent to VALU/LDS work sharing its
                          // SIMD (tools/ubench_chain2/3/4.hip: 7.3 ns per step without stores, 54 ns with a store per
                          // 8 steps next to HBM-bound waves).  Measured strides: 4 -> 16: 1024-client polyphase block
                          // 59.5 -> 53.5 us, 4096 clients 189 -> 162 us; 64 is slower again (the consumers' expansion is
                          // a dependent chain too: 6 us more in the inverse kernel, 5 us in the direct kernel).
struct XlTile {
  uint32_t tap_off;              // float2 index into the tap image; layout [Tpad][ct]
  uint32_t nclients;             // 1..ct real clients (the rest of the tile has zero taps)
  uint32_t out_off[XL_CT_MAX];   // per client: float2 index into the output image (a multiple of 2 * XL_PH_STRIDE); / XL_PH_STRIDE into the phase table
  float2 incr[XL_CT_MAX];        // per client: NCO phase increment (xlating.c:544)
  uint32_t qincr[XL_CT_MAX];     // per client: Q15 phase increment, (uint16) re | (uint16) im << 16 (xlating.c:548-549)
};
#define XL_TILE_DWORDS (2u + 4u * XL_CT_MAX)

struct XlGroup {
  uint32_t D, T, Tpad;
  uint32_t rem0;                 // class record at plan time: consumed mod D (xl_grid_dyn)
  uint32_t ntiles;               // 1..XL_NW_MAX (<= waves of the launch)
  uint32_t wide;                 // 1: D even (informational; the launch-level flag selects the 16-byte read kernel)
  uint32_t idle_before;          // spare waves (launch waves - ntiles) of the groups before this one: rider slot base
  uint32_t hv0;                  // class record at plan time: min(consumed, XL_HCAP); XL_HCAP once every window lies inside the client's own stream
  XlTile tiles[XL_NW_MAX];
};


struct XlNcoClient {
  float2 incr;          // phase increment cexpf(-j*w0*D) (xlating.c:544)
  uint32_t out_off;     // float2 index of the client's row in the phase-table image
  uint32_t slot;        // index of the client's running phase in the phase-state array
  uint32_t D;           // decimation
  uint32_t rem0;        // plan-time record: consumed mod D (its K and block boundaries follow from XlPos, xl_grid.h)
};

struct XlFirArgs {
  const void *in0;      // first part of the sample stream seen by this launch (history), n0 samples
  const void *in1;      // second part (the new block), n1 samples; may be null when n1 == 0
  uint32_t n0, n1;
  int fmt;              // XLF_*
  XlPos pos;            // stream position of this call (the groups' per-call numbers follow from it: xl_grid_dyn)
  uint32_t explicit_dyn;  // 1: single-filter path -- every group uses dyn1 as given (a one-block call, G = 1)
  XlDyn dyn1;
  const XlGroup *groups;
  uint32_t ngroups;
  uint32_t xtiles;      // ceil(max K / outputs per tile)
  uint32_t ota;         // outputs per wave: 64, or 32/16/8 (upper lanes idle) when a 64-output window image would
                        // not fit the LDS (very large decimations)
  uint32_t flags;       // bit 0: every group of the launch has even D (16-byte LDS reads); bit 1: flat wave priority; bit 2: (set; historic) priority segments end at 1/2 and 7/8
  const float2 *taps;   // tap image
  const float2 *phtab;  // NCO phase table: every XL_PH_STRIDE-th phase, entry (out index) / XL_PH_STRIDE
  float2 *out;
  void *hist_out;       // batch engine: where to write the rolled raw history (null: no roll)
  uint32_t hist_units;  // history length in 2-byte units (= n0 * bytes-per-sample / 2)
  uint32_t block_units; // block length in 2-byte units (= n1 * bytes-per-sample / 2)
  // "NCO role": the first nco_blocks workgroups of the launch tabulate the NEXT call's phase table (stream position
  // xl_grid_next(pos): same shape assumed) instead of filtering (xl_batch.cpp).  All null/0 when the launch carries
  // no NCO role.
  const XlNcoClient *nco_clients;
  uint32_t nco_nclients;
  uint32_t nco_blocks;   // = ceil(nco_nclients / (XL_NCO_LANES * nco_wpw))
  uint32_t nco_wpw;      // waves of an NCO-role workgroup that carry clients (1..waves per workgroup)
  // "riders": instead of workgroups of its own the NCO role rides in the spare waves of groups with fewer tiles than
  // the launch has waves (slot = (idle_before + wave - ntiles) * xtiles + x); slots < nco_slots carry nco_lanes clients.
  uint32_t nco_lanes;    // 0: no riders (nco_blocks workgroups do the role, if any)
  uint32_t nco_slots;
  const float2 *nco_state_in;
  float2 *nco_state_out;
  float2 *nco_tab;
  unsigned long long *trace;  // tuning only: per wave 6 words (wall_clock64 at entry, staged, filtered, stored; HW_ID; XCC_ID)
};


// ---- launchers (xl_kernels.hip).  All return hipError_t of the launch. -------------------------------------
// mode: 0 native (bit-exact scalar order, unfused), 1 optimized (fma).  ct: 1, 2, 4, 8, 9, 10 or 12 clients per tile.
// a.xtiles = ceil(max K / a.ota); lds_bytes = xl_fir_lds_bytes_ota(D, Tpad, a.ota) maximised over the groups.
// nw: waves per workgroup (>= the largest ntiles of the groups).
hipError_t xl_launch_fir(int ct, int mode, int nw, const XlFirArgs &a, size_t lds_bytes, hipStream_t s);
#define XL_NCO_LANES 64u  // clients per wave in the NCO table kernel / NCO role (with every 4th phase stored the
                          // stores are rare enough that a full wave costs the chain nothing: 8.8 vs 10.6 ns per step)
// reads the running phases from state_in[slot], writes the post-block phases to state_out[slot] (may alias).
// Every client's out_off must be a multiple of 2 * XL_PH_STRIDE (16-byte stores of table entry pairs).  prio: wave priority 0..3 of the kernel.
// pos: stream position of the call to tabulate; explicit_K != 0xFFFFFFFF: single-filter path, one block of explicit_K outputs.
hipError_t xl_launch_nco_table(const XlNcoClient *clients, uint32_t nclients, const float2 *state_in,
                               float2 *state_out, float2 *phtab, XlPos pos, uint32_t explicit_K, uint32_t prio,
                               hipStream_t s);
// the same for a whole call on a side stream: one wave per SIMD (all of its VGPRs), 64 clients per workgroup (one
// chain wave writing into an LDS ring + three waves draining it into the table)
// stats (tuning, may be null): per workgroup {shader cycles, 100 MHz ticks, entries, start tick} of the chain wave
#define XL_CHAIN_MAXCALLS 4u
struct XlChainCalls {  // what one chain launch produces: per call the phase table and the phases after the call
  float2 *tab[XL_CHAIN_MAXCALLS];
  float2 *state_out[XL_CHAIN_MAXCALLS];
  uint32_t n;
};
hipError_t xl_launch_nco_chain(const XlNcoClient *clients, uint32_t nclients, const float2 *state_in,
                               const XlChainCalls &calls, XlPos pos, unsigned long long *stats, hipStream_t s,
                               hipEvent_t done);
// raw -> converted sample images of the single-filter path (xlating.c:352-433)
hipError_t xl_launch_convert_cf32(const void *raw, int fmt, uint32_t nsamples, float2 *dst, hipStream_t s);
hipError_t xl_launch_convert_q15(const void *raw, int fmt, uint32_t nelems, int16_t *dst, hipStream_t s);
// memmove(buf, buf + from, count) in elements of `elem_bytes` (4 or 8) with overlap-safe forward order
hipError_t xl_launch_move_down(void *buf, uint32_t from, uint32_t count, uint32_t elem_bytes, hipStream_t s);
// new_hist[j] = concat(hist[0..h), block[0..n))[n + j], j < h   (raw samples, `bps` bytes each)
hipError_t xl_launch_update_history(const void *hist, const void *block, uint32_t h, uint32_t n, uint32_t bps,
                                    void *new_hist, hipStream_t s);
// Q15 family (xlating.c:92-140): one filter.  The phase table holds every XL_PH_STRIDE-th phase (the FIR kernel steps
// the rest); state_in -> state_out: the running phase before / after the call (may alias)
hipError_t xl_launch_nco_table_q15(int16_t incr_re, int16_t incr_im, const short2 *state_in, short2 *state_out, short2 *phtab,
                                   uint32_t K, hipStream_t s);
hipError_t xl_launch_fir_q15(const short2 *work, const short2 *taps, uint32_t T, uint32_t D, uint32_t K, int16_t incr_re,
                             int16_t incr_im, const short2 *phtab, short2 *out, hipStream_t s);

// Q15 family on the batched boundary (xlating.c:92-140 per client): phase table (every XL_PH_STRIDE-th phase of the
// truncating int16 recurrence, never renormalised) and the FIR launch -- exact integer arithmetic carried in float64 FMAs.
//   qinc[i]: increment of clients[i] packed like XlTile::qincr; qstate[slot]: running Q15 phase (updated in place)
hipError_t xl_launch_nco_q15_batch(const XlNcoClient *clients, const uint32_t *qinc, uint32_t nclients, short2 *qstate,
                                   short2 *qphtab, XlPos pos, hipStream_t s);
//   a: as for xl_launch_fir (no NCO role); qtaps: [tile][Tpad][ct] (re, im) doubles, same indexing as a.taps;
//   outputs: short2 rows at the addresses of the float2 rows (a.out + out_off)
hipError_t xl_launch_fir_q15_batch(int ct, int nw, const XlFirArgs &a, const double *qtaps, const short2 *qphtab,
                                   size_t lds_bytes, hipStream_t s);

// window image bytes for `ota` outputs per tile
size_t xl_fir_lds_bytes_ota(uint32_t D, uint32_t Tpad, uint32_t ota);
// largest outputs-per-wave in {64, 32, 16, 8} whose window image fits `budget` bytes of LDS; 0 if none
uint32_t xl_fir_pick_ota(uint32_t D, uint32_t Tpad, size_t budget);
// taps consumed per inner-loop iteration by the kernel of tile height ct; Tpad must be a multiple of it
static inline uint32_t xl_tap_step(int ct) { return (ct == 9 || ct == 10) ? 6u : 4u; }

#endif
