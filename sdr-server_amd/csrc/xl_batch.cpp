// xl_batch.cpp -- batched fan-out engine behind include/xlating_batch.h.
//
// What the reference does per IQ block with N clients (src/tcp_server.c:257-271 -> src/dsp_worker.c:202-204 ->
// src/queue.c:87-119 -> dsp_worker.c:57-65): N memcpy's of the block and N process_* calls on N threads.
// Here: the block lives ONCE in HBM, one to three launches serve every client of this GPU, outputs stay in HBM until
// fetched.  A CALL covers G >= 1 consecutive blocks ("group"): the results are those of G successive reference calls
// (the NCO phase is renormalised at every block end, xlating.c:73) from one set of launches -- the branch spectra of
// the polyphase path are streamed once per call instead of once per block, the launches are G times bigger, and the
// NCO recurrence (a dependent chain of ~23 us per block whatever the client count) has G times more work to hide in.
//
// HBM layout (all resident across calls; a call's only per-call data is XlPos, 16 bytes of kernel arguments):
//   hist[2]      raw history (ping-pong): the last XL_HCAP samples of the stream in the INPUT format
//   block        engine-owned copy of the current blocks (host path) -- or the caller's device buffer, in place
//   taps         [tile][Tpad][ct] float2, tap i of the ct clients of a tile contiguous (one s_load_dwordx16)
//   groups[ct]   XlGroup descriptors (<= 4 tiles of one class each) incl. the class's plan-time stream record
//   nco          XlNcoClient per client; phase[2][slot] running NCO phases (committed / next)
//   phtab[2]     [client][K_cap / 16] float2 phase tables (ping-pong), out[2] outputs (ping-pong), same indexing
//
// Streaming rule (xl_grid.h): a client's outputs lie on the global grid n = k*D of ITS stream; output
// k's newest sample is stream sample k*D.  Samples older than the client (it joined mid-stream) must read as zero.
// Classes: clients sharing (D, T, stream offset mod D, valid history) share tiles of the direct kernel; their number is
// unlimited (the per-call numbers of a class are computed on the device from its record and XlPos).  In optimized mode
// all "mature" clients (every window inside their own stream) of one (D, T) form ONE polyphase class whatever their
// grid offsets (xl_polyphase.h).
//
// Streams.  A call's outputs depend on (raw history, blocks, phase table) only.  The phase table is data independent
// (float32 recurrence p <- p * incr, xlating.c:70-73), so call c+1's table is tabulated inside call c's launches (the
// "NCO role"), guessing that c+1 has the same shape; the running phases are double-buffered (committed / next) so that
// a wrong guess is simply redone by a small launch of its own.  Everything runs on the caller's stream in stream order.
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/xlating_batch.h"
#include "xl_common.h"
#include "xl_device.h"
#include "xl_mixf_layout.h"
#include "xl_polyphase.h"
#include "xl_taps.h"

#define XL_NLAUNCH 7
#define XL_GROUP_MAX 64u  // most blocks per call

namespace {

// Phase tables / phase buffers in the engine's rings.  A chain launch made at call c for calls c+1 .. c+n overwrites the
// tables of calls c+n-XL_NTAB+1 ..; with XL_NTAB = 2 n the latest of them belongs to the launch before the previous one, whose
// readers ran while the previous launch was stepping -- the new launch never has to wait for them on a chain-bound engine.
#define XL_NTAB (2 * XL_CHAIN_MAXCALLS)
static inline int xl_nx(int i) { return (i + 1) % XL_NTAB; }

struct Client {
  bool alive = false;
  uint32_t D = 0, T = 0, Tpad = 0;
  std::vector<float> rt;  // [Tpad] interleaved re,im, zero padded
  std::vector<int16_t> rtq;  // [T] the same taps in Q15 (xlating.c:486-487), interleaved re,im
  float incr[2] = {1.0f, 0.0f};
  int16_t qincr[2] = {0, 0};  // Q15 phase increment (xlating.c:548-549)
  uint64_t consumed = 0;
  uint32_t out_off = 0, out_cap = 0;  // the client's row in the output / phase-table images: assigned when it joins, kept for its lifetime
  uint32_t row_len = 0;               // elements reserved for the row (out_cap rounded up to the table-entry pair)
  uint64_t uid = 0;                   // unique over the engine's life (client ids are recycled, their taps are not)
  uint32_t last_K = 0;
  std::vector<uint32_t> last_Kg;  // outputs per block of the latest call
  bool planned_mature = false;
};

// every window of the client's next outputs lies inside its own stream (no zeros below its join point are needed)
static inline bool xl_mature(const Client &c) { return c.consumed >= (uint64_t)c.T - 1u; }

struct DirectClass {
  uint32_t D, T, rem0, hv0;
  std::vector<int> members;
};

struct Launch {
  int ct = 0;
  int nw = XL_NW_DEFAULT;  // waves (tiles) per workgroup
  uint32_t ota = 64;       // outputs per wave (smaller only when a 64-output window image exceeds the LDS)
  std::vector<XlGroup> groups;
  XlGroup *d_groups = nullptr;
  size_t lds = 0;  // window image bytes of the launch (max over its groups)
  uint32_t idle_waves = 0;  // spare waves over all groups (NCO rider slots per output tile)
  bool all_wide = true;  // every group has an even decimation
  uint32_t maxD = 1, minD = 0xFFFFFFFFu;
};

// A class of clients evaluated by the polyphase overlap-save path (xl_polyphase.hip) in optimized mode: all mature
// clients of one (D, T), whatever their grid offsets -- or clients of one (D, T) that joined together and are still
// inside their zero history (one grid, one zero_below).
#define XL_SIDE_ONE_BLOCK_MAX 2048u  // one-block polyphase calls: side-stream chain kernel up to this many clients

struct PolyClass {
  uint32_t D = 0, Dpad = 0, T = 0, A = 0, V = 0;
  uint32_t M = 256;          // transform length (128 or 256), V = M - A + 1
  uint32_t ncols = 0, ncg = 0, nseg_cap = 0;
  uint32_t rem_ref0 = 0;     // plan-time record of the shared grid's reference client (xl_grid.h)
  uint32_t hv0 = XL_HCAP;    // plan-time valid history of the members (XL_HCAP: mature)
  uint32_t dmax = 0;         // largest grid offset of a member
  std::vector<int> members;
  // The class outlives re-plans (incremental planning): a member keeps its column for its lifetime, a column that a leaving
  // member frees is handed to the next joiner, and the branch spectra are computed for NEW columns only -- a join costs
  // one column of R (8 D M bytes), not the class's whole image.
  std::vector<int> col_client;        // column -> client id, -1 = free
  std::vector<uint32_t> col_delta;    // delay the column's spectra were built with
  std::vector<uint64_t> col_uid;      // ... and for whom (Client::uid)
  std::map<int, uint32_t> col_of;     // client id -> column
  uint32_t ncg_cap = 0;               // column groups the R / Y / cols buffers hold
  bool keep = false;                  // (planning scratch: the class was taken over by the new plan)
  // The branch spectra live in the mix launch's B-operand order (d_Rh).  mix_kind 1 (xlp_mix_mfma_kernel): scaled per column by a power
  // of two and split in two halves; per column the scale (host) and what undoes it (device).  mix_kind 3 (xlp_mix_f32_kernel): float32
  uint32_t mix_kind = 1, nkb = 0;
  void *d_Rh = nullptr;
  std::vector<float> col_scale;
  float *d_cscale = nullptr;
  float2 *d_X = nullptr;     // shared spectra [passes][Dpad][M][16]
  // two-half mix of a cf32 stream: per segment the largest component of its shared spectra, found by the forward launch (XlpArgs::segmax):
  // two buffers of seg_cap entries, a call uses buffer seg_par; lives and dies with d_X
  uint32_t *d_segmax = nullptr;
  uint32_t seg_cap = 0, seg_par = 0;
  float2 *d_Y = nullptr;     // mixed spectra  [ncg][nseg_cap][M][128]
  XlpCol *d_cols = nullptr;  // per column: output row, grid offset, NCO increment
  int last_inv = -1;         // which inverse kernel the class's latest launch took (describe; xlp_inverse_pick): 3 / 5 / 6, -1 = none yet
};

}  // namespace

struct xlating_batch_t {
  uint32_t fs = 0;
  int fmt = 0;
  uint32_t bps = 2;
  uint32_t max_samples = 0;  // per block
  uint32_t gcap = 1;         // most blocks per call
  int device = -1;
  hipStream_t own_stream = nullptr;   // used when the caller passes no stream / host path
  hipStream_t last_stream = nullptr;  // caller stream of the latest call
  hipEvent_t dep_ev = nullptr;        // orders a call on a new stream behind the previous call's stream
  // Side-stream NCO chain (calls of several blocks): the next call's phase table is tabulated by xl_nco_chain_kernel on
  // nco_stream while this call's launches run on the caller's stream.  ev_chain[t]: the tabulation of table t is complete (the
  // caller's stream waits for it before the first launch that reads the table); ev_done[t]: the launches that read
  // table t have been passed by the caller's stream (nco_stream waits for it before overwriting that table).
  hipStream_t nco_stream = nullptr;
  // CU reservation: the chain kernel needs whole CUs (one wave per SIMD) and finds them only if nothing else is resident
  // there -- behind a launch that fills the chip its workgroups waited for the next kernel boundary (measured: chain
  // kernel 215 us alone, 262-293 us launched next to the forward / mix launches).  So nco_stream is created with a CU
  // mask of `reserve_r` CUs per XCD (mask bit b = XCD b % 8, CU b / 8 of it; tools/ubench_cumask.hip) and the engine's
  // own compute stream for side-stream calls, cs_masked, with the complement.  Callers that pass XL_STREAM_ENGINE get it.
  hipStream_t cs_masked = nullptr;
  hipStream_t last_nco = nullptr;    // the side stream of the latest chain launch
  unsigned long long *d_chain_stats = nullptr;  // tuning (XL_EXP_CHAIN_STATS): per chain workgroup cycles / ticks of the latest launch
  hipStream_t nco_masked = nullptr;  // the side stream that goes with cs_masked; nco_stream (unmasked) serves callers' own streams
  uint32_t reserve_r = 0;
  int reserve_band = -1;  // band of the reservation rule the latest plan was in (xl_chain_band; -1: none yet): hysteresis at the band edges
  uint32_t expected_clients = 0;  // option "expected_clients": the CUs are reserved for this many clients from the first plan on
  hipEvent_t ev_chain[XL_NTAB] = {}, ev_done[XL_NTAB] = {};  // per phase table
  bool ev_done_valid[XL_NTAB] = {};
  hipStream_t ev_done_stream[XL_NTAB] = {};  // where ev_done[t] was recorded
  // Round 4: a launch that carries a completion event keeps the queue ~8 us from starting the next launch (a fifth of a one-block
  // call), and in the steady side-stream pattern only ONE call in `chain_calls` needs its event: a chain launch made at call k
  // overwrites the tables last read by calls k-7 .. k-4, and call k-4 is the previous chain-launching call.  So side-stream calls
  // record ev_done only when they launch a chain; tab_call[] / done_call let a chain launch check that the latest recorded event
  // post-dates every reader of its tables (calls run in order: an event behind call j covers all calls <= j), and fall back to an
  // event recorded on the spot when it does not (irregular patterns only).
  uint64_t tab_call[XL_NTAB] = {};  // ncalls + 1 of the latest call that read table t (0: never)
  uint64_t done_call = 0;           // ncalls + 1 of the latest call that recorded an ev_done (0: none)
  int done_tab = 0;                 // ... and which one
  bool spec_on_side = false;  // the look-ahead table was produced on nco_stream (ev_chain must be waited for)
  double macs_all = 0.0, macs_rest = 0.0;  // complex MACs per sample of a block: all clients' direct launches / those outside `poly`
  int nco_side = -1;          // option "nco_side_stream": 1 always, 0 never (NCO role inside the launches), -1: calls of >= 2 blocks
  bool poisoned = false;              // a launch failed mid-call: device state is undefined, every later call fails

  std::vector<Client> clients;
  int nalive = 0;
  bool dirty = true;
  // Plan buffers are recycled across re-plans (a client joining or leaving rebuilds the plan; hipMalloc / hipFree of a few
  // hundred MB took 30-90 ms per re-plan at 1024-4096 clients): live allocations with their capacities, and the ones the
  // previous plan released, waiting to be taken again.
  std::map<void *, size_t> plan_caps;
  std::vector<void *> plan_spare;
  bool want_q15 = false;  // the Q15 tap image is built from the first XL_MODE_Q15 call on
  int planned_immature = 0;  // clients that were not mature when the plan was built (they merge once they are)
  uint32_t trel = 0;         // samples consumed since the plan was built (XlPos::trel)
  uint64_t calls_since_plan = 0;  // calls run with the current plan (their outputs live in its rows)
  uint32_t plan_maxD = 1;
  std::vector<DirectClass> classes;       // direct classes over ALL clients (native mode)
  std::vector<DirectClass> classes_rest;  // direct classes over the clients outside `poly` (optimized mode)
  Launch launches[XL_NLAUNCH];       // one per register-tile height 12, 10, 9, 8, 4, 2, 1: ALL clients (native mode).  Built on
                                     // demand (xl_batch_build_all_set): an engine that only ever runs optimized calls on the polyphase
                                     // path never pays for the all-clients tap image (4 MB and 1.4 ms per re-plan at 1024 clients)
  bool all_built = false;            // launches[] / d_taps / d_qtaps match the current plan
  int big_h = 8;                     // register-tile height of the large direct classes (chosen per plan)
  uint64_t next_uid = 1;
  std::map<uint32_t, uint32_t> free_rows;  // output rows: free extents (offset -> length) below rows_end
  uint32_t rows_end = 0;
  Launch launches_rest[XL_NLAUNCH];  // same, over classes_rest (optimized mode)
  std::vector<PolyClass> poly;       // classes on the polyphase overlap-save path in optimized mode
  float2 *d_W = nullptr;             // e^{-2 pi j n/256}
  float2 *d_phase_run = nullptr;     // running phases between the NCO slices of a call
  size_t phase_run_cap = 0;
  int poly_mode = -1;        // option "polyphase": 0 never, 1 whenever the shape allows, -1 (default) by the size rule
  bool poly_min_set = false;        // "polyphase_min_clients" was given: it holds for every class (else 32 where the mix runs on the matrix cores)
  uint32_t poly_min_clients = 32;   // XL_EXP_POLY_MIN (tuning): smallest class that takes the polyphase path under the size rule
  uint32_t poly_m = 0;        // option "polyphase_m": force the transform length (64 / 128 / 256); 0 = by the size rule
  int num_cus = 256;
  uint32_t inv_reg = 0;       // option "inverse_kernel", M = 128 classes: 0 (default) = by the launch's size (xlp_inverse_pick: the 8-lane kernel
                              // for launches of up to 2048 tiles, the LDS transform up to 8192, the 32 x 4 cut beyond), 5 = always eight lanes per column, 16- and
                              // 8-point transforms in registers (xl_inv8.hip), 6 = always the 32 x 4 cut (xl_inv32.hip: 32-point transforms
                              // in registers, whole-line loads, 256-byte store runs), 3 = always staged in LDS on dense rows with an XOR swizzle
  uint32_t mix_kernel = 1;    // option "mix_kernel": 1 (default) = two-half float16 operands on the matrix cores where the class allows them
                              // (integer input format, D <= 64), float32 operands on the matrix cores everywhere else (cf32 input,
                              // D > 64); 3 = float32 operands for every class (the all-float32 arithmetic of the path)
  uint32_t mix_pp = 0;        // XL_EXP_MIX_PP (tuning): passes per workgroup of the mix launch; 0 = the launcher's default
  uint32_t inv_skip_at = 256;  // inverse launch (4-wave workgroups, dealt per CU): one workgroup slot kept empty on the chain CUs
  uint32_t poly_exp = 0;     // XL_TUNING builds: tuning switches of the mix kernel
  uint32_t poly_slice_fi = 20000;  // NCO role inside the launches: TWO slices, forward | inverse, boundary in 1/65536 of the call (no launch
                                   // that issues matrix instructions hosts the role; the inverse launch is the longer one: 31 % | 69 %
                                   // keeps both slices inside their launches at 4096 clients with the scalar role step)
  std::vector<XlNcoClient> nco;
  size_t out_total = 0;
  uint64_t ncalls = 0;  // calls processed

  void *d_hist[2] = {nullptr, nullptr};
  int hcur = 0;  // d_hist[hcur] = history in front of the next call
  void *d_block = nullptr;
  void *h_block = nullptr;  // pinned staging
  float2 *d_taps = nullptr;       // tap image of the all-clients launch set
  float2 *d_taps_rest = nullptr;  // tap image of the optimized-mode launch set (the clients outside the polyphase classes)
  double *d_qtaps = nullptr;   // Q15 taps as doubles, same indexing as d_taps (XL_MODE_Q15)
  uint32_t *d_qinc = nullptr;  // per nco entry: packed Q15 phase increment
  short2 *d_qphase = nullptr;  // per slot: running Q15 phase (xlating.c:546-547: starts at 32767 + 0j)
  short2 *d_qphtab = nullptr;  // Q15 phase table (every XL_PH_STRIDE-th phase)
  bool last_q15 = false;       // the latest call produced cs16 outputs
  XlNcoClient *d_nco = nullptr;
  // Rings of XL_NTAB phase buffers and phase tables: [pcur] = committed running phases, [pcur + 1] = after the next call,
  // [pcur + 2] = after the one behind it (a chain launch may tabulate two calls ahead); table [tab] = the latest call's.
  float2 *d_phase[XL_NTAB] = {};
  int pcur = 0;
  size_t phase_cap = 0;
  float2 *d_phtab[XL_NTAB] = {};
  float2 *d_out[2] = {nullptr, nullptr};
  int ocur = 0;  // d_out[ocur] holds the latest call's outputs
  size_t out_alloc = 0;
  float2 *h_out = nullptr;
  size_t h_out_alloc = 0;
  bool fetched = false;

  int tab = 0;              // table used by the latest call
  // Look-ahead: tables [tab + 1] .. [tab + spec_n] hold the phases of the next spec_n calls assuming spec_S samples x
  // spec_G blocks each (the phases after them: d_phase[pcur + 1 ..]); produced by the latest call's launches (spec_n = 1)
  // or by one chain launch on the side stream (spec_n <= 2), whose completion is ev_chain[spec_ev].
  int spec_n = 0;
  uint32_t spec_S = 0, spec_G = 0, spec_flags = 0;
  int spec_ev = 0;
  bool waited_valid = false;  // stream waited_stream has waited for ev_chain[waited_ev] since that event was last recorded
  int waited_ev = 0;
  hipStream_t waited_stream = nullptr;
  int chain_calls = 4;  // XL_EXP_CHAIN_CALLS (tuning): most calls one side-stream chain launch tabulates ahead (1 .. XL_CHAIN_MAXCALLS)
  int chain_ahead = 4;  // ... and how many the NEXT launch does: 1 after a wrong shape guess (the look-ahead it drops is then one call of
                        // chain work the call has to wait out, not four), doubled by every launch whose calls were all consumed.
                        // A stream of irregular block lengths, one block per call, 1024 clients: 134 us per call with four calls
                        // ahead every time (profiles/r05_ragged_blocks.txt), against 39.5 us for constant lengths
  bool exp_nofuse = false;  // XL_TUNING: keep the NCO tabulation a launch of its own

  uint32_t exp_flags = 0;  // tuning knobs
  // Wave priority of the NCO role / NCO launch (3: a pure dependent chain must not queue behind the FIR waves;
  // 1..3 measured equal for 505 taps, 3 best for short filters).
  uint32_t nco_prio = 3;
  uint32_t nco_wpw = 1;   // waves per NCO-role workgroup that carry clients
  bool riders = true;     // NCO role rides in spare waves of the FIR workgroups when the plan has some
  int riders_min_wgs = 512;
  int exp_h = 0;   // forces the tile height of the large classes (8 | 9 | 10 | 12)
#ifdef XL_TUNING
  const char *poly_trace = nullptr;  // XL_EXP_POLY_TRACE=<file>: timeline of the latest mix launch
  unsigned long long *d_ptrace = nullptr;
  const char *exp_trace = nullptr;  // XL_EXP_TRACE=<file>: dump per-wave timestamps of the latest FIR launch
  unsigned long long *d_trace = nullptr;
  size_t trace_cap = 0;
#endif
  uint32_t timing_every = 1;  // xlating_batch_timing_stride: bracket only every n-th call (an event pair costs a few us of stream time)
  int timing = 0;  // 1: bracket every call's launches; 2: also time the three polyphase launches separately
  std::vector<hipEvent_t> ev;       // pairs: start, stop of a call's launches (on the launch stream)
  std::vector<hipEvent_t> ev_ncot;  // pairs: start, stop of stand-alone NCO launches (rare)
  std::vector<hipEvent_t> ev_poly;  // quadruples: before forward, after forward, after mix, after inverse (timing == 2)
  double poly_ms[3] = {0.0, 0.0, 0.0};
  int timed_poly = 0;
  std::vector<hipEvent_t> ev_pool;  // recycled timing events (hipEventCreate per call would bound the host)
  double fir_ms = 0.0, nco_ms = 0.0;
  int timed_launches = 0, timed_nco = 0;
};

static hipError_t xl_batch_timing_event(xlating_batch *b, hipEvent_t *out) {
  if (!b->ev_pool.empty()) {
    *out = b->ev_pool.back();
    b->ev_pool.pop_back();
    return hipSuccess;
  }
  return hipEventCreate(out);
}

static void xl_batch_sync_all(xlating_batch *b) {
  (void)hipStreamSynchronize(b->last_stream);  // may be the NULL (legacy default) stream: still a real stream
  if (b->own_stream) (void)hipStreamSynchronize(b->own_stream);
  if (b->nco_stream) (void)hipStreamSynchronize(b->nco_stream);
  if (b->cs_masked) (void)hipStreamSynchronize(b->cs_masked);
  if (b->nco_masked) (void)hipStreamSynchronize(b->nco_masked);
}

// A plan buffer of at least `bytes`: the smallest spare one that fits (and is not more than twice too big), else a fresh
// allocation with 1/8 of headroom, so that the next few joins find room in place.
static hipError_t xl_plan_alloc(xlating_batch *b, void **out, size_t bytes) {
  int best = -1;
  for (size_t i = 0; i < b->plan_spare.size(); ++i) {
    const size_t cap = b->plan_caps[b->plan_spare[i]];
    if (cap >= bytes && cap <= 2 * bytes + 4096 && (best < 0 || cap < b->plan_caps[b->plan_spare[best]])) best = (int)i;
  }
  if (best >= 0) {
    *out = b->plan_spare[best];
    b->plan_spare.erase(b->plan_spare.begin() + best);
    return hipSuccess;
  }
  const size_t cap = bytes + bytes / 8 + 256;
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess) return e;
  b->plan_caps[p] = cap;
  *out = p;
  return hipSuccess;
}

static void xl_plan_release(xlating_batch *b, void *p) {
  if (p) b->plan_spare.push_back(p);
}

// what the finished plan did not take again goes back to the device
static void xl_plan_trim(xlating_batch *b) {
  for (void *p : b->plan_spare) {
    (void)hipFree(p);
    b->plan_caps.erase(p);
  }
  b->plan_spare.clear();
}

static void xl_poly_release(xlating_batch *b, PolyClass &pc) {
  void *dev[] = {pc.d_X, pc.d_Y, pc.d_cols, pc.d_Rh, pc.d_cscale, pc.d_segmax};
  for (void *q : dev) xl_plan_release(b, q);
  pc.d_X = pc.d_Y = nullptr;
  pc.d_segmax = nullptr;
  pc.d_cols = nullptr;
  pc.d_Rh = nullptr;
  pc.d_cscale = nullptr;
}

static void xl_release_launch_set(xlating_batch *b, Launch *set) {
  for (int i = 0; i < XL_NLAUNCH; ++i) {
    Launch &l = set[i];
    xl_plan_release(b, l.d_groups);
    l.d_groups = nullptr;
    l.groups.clear();
  }
}

// Drops what a plan builds from scratch every time (launch sets, tap images, NCO records).  The polyphase classes
// survive re-plans (xl_batch_plan takes over the ones that still fit); `all` releases them too.
static void xl_batch_free_plan(xlating_batch *b, bool all) {
  xl_release_launch_set(b, b->launches);
  xl_release_launch_set(b, b->launches_rest);
  b->all_built = false;
  if (all) {
    for (PolyClass &pc : b->poly) xl_poly_release(b, pc);
    b->poly.clear();
  }
  xl_plan_release(b, b->d_taps);
  xl_plan_release(b, b->d_taps_rest);
  xl_plan_release(b, b->d_qtaps);
  xl_plan_release(b, b->d_qinc);
  xl_plan_release(b, b->d_nco);
  b->d_taps = b->d_taps_rest = nullptr;
  b->d_qtaps = nullptr;
  b->d_qinc = nullptr;
  b->d_nco = nullptr;
}

extern "C" void xlating_batch_destroy(xlating_batch *b) {
  if (b == nullptr) return;
  if (b->device >= 0) (void)hipSetDevice(b->device);
  xl_batch_sync_all(b);
  xl_batch_free_plan(b, true);
  xl_plan_trim(b);
  void *dev[] = {b->d_hist[0], b->d_hist[1], b->d_block, b->d_out[0], b->d_out[1], b->d_W, b->d_phase_run, b->d_qphase, b->d_qphtab};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  for (int i = 0; i < XL_NTAB; ++i) {
    if (b->d_phase[i]) (void)hipFree(b->d_phase[i]);
    if (b->d_phtab[i]) (void)hipFree(b->d_phtab[i]);
  }
  if (b->d_chain_stats) (void)hipFree(b->d_chain_stats);
#ifdef XL_TUNING
  if (b->d_trace) (void)hipFree(b->d_trace);
  if (b->d_ptrace) (void)hipFree(b->d_ptrace);
#endif
  if (b->h_block) (void)hipHostFree(b->h_block);
  if (b->h_out) (void)hipHostFree(b->h_out);
  for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->ev_ncot) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->ev_poly) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->ev_pool) (void)hipEventDestroy(e);
  if (b->dep_ev) (void)hipEventDestroy(b->dep_ev);
  for (hipEvent_t e : b->ev_chain)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->ev_done)
    if (e) (void)hipEventDestroy(e);
  if (b->nco_stream) (void)hipStreamDestroy(b->nco_stream);
  if (b->cs_masked) (void)hipStreamDestroy(b->cs_masked);
  if (b->nco_masked) (void)hipStreamDestroy(b->nco_masked);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
}

// Tuning knobs that are not part of the documented option set: reachable through XL_EXP_* environment variables read at create
// (tests and tools force a path this way; all of them are result-neutral within the parity bars)
static int xl_batch_set_tuning(xlating_batch *b, const std::string &n, long value) {
  if (n == "polyphase_min_clients") {
    if (value < 1) return -EINVAL;
    b->poly_min_clients = (uint32_t)value;
    b->poly_min_set = true;
  } else if (n == "mix_passes_per_workgroup") {
    if (value < 0 || value > 64) return -EINVAL;
    b->mix_pp = (uint32_t)value;
  } else if (n == "riders") {
    b->riders = value != 0;
  } else if (n == "riders_min_workgroups") {
    b->riders_min_wgs = (int)value;
  } else if (n == "tile_height") {
    if (value != 0 && value != 8 && value != 9 && value != 10 && value != 12) return -EINVAL;
    b->exp_h = (int)value;
  } else if (n == "nco_calls_per_launch") {
    if (value < 1 || value > (long)XL_CHAIN_MAXCALLS) return -EINVAL;
    b->chain_calls = (int)value;
  } else if (n == "nco_slice") {  // forward | inverse boundary of the in-launch NCO role, in 1/65536 of a call
    if (value < 0 || value > 65536) return -EINVAL;
    b->poly_slice_fi = (uint32_t)value;
  } else {
    return -ENOENT;
  }
  b->dirty = true;
  return 0;
}

extern "C" int xlating_batch_set_option(xlating_batch *b, const char *name, long value) {
  if (b == nullptr || name == nullptr) return -EINVAL;
  const std::string n(name);
  if (n == "polyphase") {
    if (value < -1 || value > 1) return -EINVAL;
    b->poly_mode = (int)value;
  } else if (n == "polyphase_m") {
    if (value != 0 && value != 64 && value != 128 && value != 256) return -EINVAL;
    b->poly_m = (uint32_t)value;
  } else if (n == "inverse_kernel") {
    if (value != 0 && value != 3 && value != 5 && value != 6) return -EINVAL;
    b->inv_reg = (uint32_t)value;
  } else if (n == "mix_kernel") {
    if (value != 1 && value != 3) return -EINVAL;
    b->mix_kernel = (uint32_t)value;
  } else if (n == "expected_clients") {
    if (value < 0 || value > 8192) return -EINVAL;
    b->expected_clients = (uint32_t)value;
  } else if (n == "nco_side_stream") {
    if (value < -1 || value > 1) return -EINVAL;
    b->nco_side = (int)value;
  } else {
    return -ENOENT;
  }
  b->dirty = true;
  return 0;
}

extern "C" int xlating_batch_create_grouped(uint32_t sampling_freq, int input_format, uint32_t max_input_buffer_length,
                                            unsigned max_group_blocks, int device, xlating_batch **batch) {
  if (batch == nullptr || input_format < XL_FMT_CU8 || input_format > XL_FMT_CF32 || sampling_freq == 0 ||
      max_input_buffer_length < 2 || max_group_blocks < 1 || max_group_blocks > XL_GROUP_MAX ||
      (uint64_t)(max_input_buffer_length / 2) * max_group_blocks > (1u << 28))
    return -EINVAL;
  const int dev = xl_hip_select_device(device);
  if (dev < 0) {
    XL_LOG_ERR("no usable HIP device (%s); this build has no CPU arithmetic path", xlating_hip_device_info());
    return -ENODEV;
  }
  xlating_batch *b = new (std::nothrow) xlating_batch_t();
  if (b == nullptr) return -ENOMEM;
  b->fs = sampling_freq;
  b->fmt = input_format;
  b->bps = xl_bytes_per_sample(input_format);
  b->max_samples = max_input_buffer_length / 2;
  b->gcap = max_group_blocks;
  b->device = dev;
  {
    const size_t hbytes = (size_t)XL_HCAP * b->bps;
    const size_t bbytes = (size_t)b->max_samples * b->gcap * b->bps + 16;
    XL_TRY(hipSetDevice(dev));
    XL_TRY(hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking));
    if (hipDeviceGetAttribute(&b->num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || b->num_cus <= 0) b->num_cus = 256;
    XL_TRY(hipEventCreateWithFlags(&b->dep_ev, hipEventDisableTiming));
    XL_TRY(hipStreamCreateWithFlags(&b->nco_stream, hipStreamNonBlocking));
    for (int i = 0; i < XL_NTAB; ++i) {
      XL_TRY(hipEventCreateWithFlags(&b->ev_chain[i], hipEventDisableTiming));
      XL_TRY(hipEventCreateWithFlags(&b->ev_done[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 2; ++i) {
      XL_TRY(hipMalloc(&b->d_hist[i], hbytes));
      XL_TRY(hipMemsetAsync(b->d_hist[i], 0, hbytes, b->own_stream));
    }
    XL_TRY(hipMalloc(&b->d_block, bbytes));
    XL_TRY(hipHostMalloc(&b->h_block, bbytes, hipHostMallocDefault));
    XL_TRY(hipStreamSynchronize(b->own_stream));
  }
  // Result-neutral plan options may also come from the environment (tests and tools force a path this way); the
  // switches that trace launches or break results exist in -DXL_TUNING builds only (tools/experiments/).
  {
    static const struct {
      const char *env, *opt;
      bool documented;
    } knobs[] = {{"XL_EXP_POLY", "polyphase", true},          {"XL_EXP_POLY_M", "polyphase_m", true},
                 {"XL_EXP_INV", "inverse_kernel", true},      {"XL_EXP_MIX", "mix_kernel", true},
                 {"XL_EXP_NCO_SIDE", "nco_side_stream", true}, {"XL_EXP_EXPECTED", "expected_clients", true},
                 {"XL_EXP_POLY_MIN", "polyphase_min_clients", false}, {"XL_EXP_MIX_PP", "mix_passes_per_workgroup", false},
                 {"XL_EXP_H", "tile_height", false},          {"XL_EXP_RIDERS", "riders", false},
                 {"XL_EXP_RIDERS_MIN", "riders_min_workgroups", false}, {"XL_EXP_CHAIN_CALLS", "nco_calls_per_launch", false},
                 {"XL_EXP_NCO_SLICE", "nco_slice", false}};
    for (const auto &k : knobs)
      if (const char *v = xl_exp_getenv(k.env)) {
        const int rc = k.documented ? xlating_batch_set_option(b, k.opt, atol(v)) : xl_batch_set_tuning(b, k.opt, atol(v));
        if (rc != 0) XL_LOG_ERR("%s=%s is not a value of \"%s\" (ignored)", k.env, v, k.opt);
      }
  }
  if (xl_exp_getenv("XL_EXP_CHAIN_STATS")) (void)hipMalloc((void **)&b->d_chain_stats, 4096 * 4 * sizeof(unsigned long long));
  if (xl_exp_getenv("XL_EXP_INVSKIP")) b->inv_skip_at = (uint32_t)atoi(xl_exp_getenv("XL_EXP_INVSKIP"));
  if (xl_exp_getenv("XL_EXP_NCOPRIO")) b->nco_prio = (uint32_t)atoi(xl_exp_getenv("XL_EXP_NCOPRIO")) & 3u;
  if (xl_exp_getenv("XL_EXP_FLATPRIO")) b->exp_flags |= 2u;
#ifdef XL_TUNING
  b->exp_trace = xl_exp_getenv("XL_EXP_TRACE");
  b->exp_nofuse = xl_exp_getenv("XL_EXP_NOFUSE") != nullptr;
  b->poly_trace = xl_exp_getenv("XL_EXP_POLY_TRACE");
  if (xl_exp_getenv("XL_EXP_POLY_EXP")) b->poly_exp = (uint32_t)atoi(xl_exp_getenv("XL_EXP_POLY_EXP"));
#endif
  b->last_stream = b->own_stream;
  b->dirty = true;
  *batch = b;
  return 0;
fail:
  xlating_batch_destroy(b);
  return xl_errno_of_last_hip_error();
}

extern "C" int xlating_batch_create(uint32_t sampling_freq, int input_format, uint32_t max_input_buffer_length,
                                    int device, xlating_batch **batch) {
  return xlating_batch_create_grouped(sampling_freq, input_format, max_input_buffer_length, 1, device, batch);
}

extern "C" int xlating_batch_num_clients(const xlating_batch *b) { return b ? b->nalive : 0; }

// Output rows.  A client's row (its outputs of a call, and 1/16 of that in the phase tables) is reserved when it joins and
// stays where it is until it leaves: re-plans never move anybody's outputs.  First fit over the free extents, else the end.
static uint32_t xl_row_alloc(xlating_batch *b, uint32_t len) {
  for (auto it = b->free_rows.begin(); it != b->free_rows.end(); ++it) {
    if (it->second < len) continue;
    const uint32_t off = it->first, rest = it->second - len;
    b->free_rows.erase(it);
    if (rest) b->free_rows[off + len] = rest;
    return off;
  }
  const uint32_t off = b->rows_end;
  b->rows_end += len;
  return off;
}

static void xl_row_free(xlating_batch *b, uint32_t off, uint32_t len) {
  if (len == 0) return;
  auto it = b->free_rows.emplace(off, len).first;
  auto nx = std::next(it);
  if (nx != b->free_rows.end() && it->first + it->second == nx->first) {
    it->second += nx->second;
    b->free_rows.erase(nx);
  }
  if (it != b->free_rows.begin()) {
    auto pv = std::prev(it);
    if (pv->first + pv->second == it->first) {
      pv->second += it->second;
      b->free_rows.erase(it);
      it = pv;
    }
  }
  if (it->first + it->second == b->rows_end) {  // the last extent gives the space back
    b->rows_end = it->first;
    b->free_rows.erase(it);
  }
}

extern "C" int xlating_batch_add_client(xlating_batch *b, uint32_t decimation, const float *taps, size_t taps_len,
                                        int32_t center_freq) {
  if (taps_len == 0) return -1;  // like create_frequency_xlating_filter (xlating.c:496-498)
  if (b == nullptr || taps == nullptr || decimation == 0) return -EINVAL;
  if (taps_len - 1 + decimation > XL_HCAP) {
    XL_LOG_ERR("%zu taps at decimation %u exceed the engine's history capacity (%u samples)", taps_len, decimation, XL_HCAP);
    return -EINVAL;
  }
  const uint32_t Tpad = xl_roundup((uint32_t)taps_len, XL_TAP_UNROLL);
  if (xl_fir_pick_ota(decimation, xl_roundup((uint32_t)taps_len, 12), 160 * 1024) == 0) {
    XL_LOG_ERR("decimation %u with %zu taps needs a %zu-byte window image even for 8 outputs per wave (> 160 KiB LDS)",
               decimation, taps_len, xl_fir_lds_bytes_ota(decimation, Tpad, 8));
    return -EINVAL;
  }
  int id = -1;
  for (size_t i = 0; i < b->clients.size(); ++i)
    if (!b->clients[i].alive) {
      id = (int)i;
      break;
    }
  if (id < 0) {
    b->clients.emplace_back();
    id = (int)b->clients.size() - 1;
  }
  Client &c = b->clients[id];
  c = Client();
  c.alive = true;
  c.uid = b->next_uid++;
  c.D = decimation;
  c.T = (uint32_t)taps_len;
  c.Tpad = Tpad;
  c.rt.assign(2 * (size_t)Tpad, 0.0f);
  c.rtq.assign(2 * taps_len, 0);
  xl_prepare_taps(taps, taps_len, center_freq, b->fs, decimation, c.rt.data(), c.rtq.data(), c.incr, c.qincr);
  c.out_cap = b->gcap * (b->max_samples / decimation + 1);  // xlating.c:568 per block
  c.row_len = xl_roundup(c.out_cap, 2 * XL_PH_STRIDE);  // rows start at multiples of 2 strides: the NCO role stores pairs of
                                                        // table entries as 16 bytes
  c.out_off = xl_row_alloc(b, c.row_len);
  b->nalive++;
  b->dirty = true;
  // The running phase of a new client starts at 1 + 0j (xlating.c:543); slot = client id.  Any phase table
  // tabulated ahead is for the old client set: drop it (the committed phases are untouched by it).
  (void)hipSetDevice(b->device);
  xl_batch_sync_all(b);
  b->spec_n = 0;
  // Every failure from here on rolls the client back (a phantom client with an undefined phase would otherwise be
  // planned and filtered on every later call, with no id in the caller's hands to remove it).
  auto rollback = [&](int code) {
    c.alive = false;
    c.rt.clear();
    xl_row_free(b, c.out_off, c.row_len);
    c.row_len = 0;
    b->nalive--;
    return code;
  };
  if ((size_t)id >= b->phase_cap) {
    // grow all running-phase buffers or none: allocate the whole new set first, swap it in only when complete
    const size_t ncap = std::max<size_t>(1024, 2 * b->clients.size());
    float2 *np[XL_NTAB] = {};
    short2 *nq = nullptr;
    bool ok = hipMalloc((void **)&nq, ncap * sizeof(short2)) == hipSuccess;
    for (int i = 0; ok && i < XL_NTAB; ++i) ok = hipMalloc((void **)&np[i], ncap * sizeof(float2)) == hipSuccess;
    for (int i = 0; ok && i < XL_NTAB; ++i)
      if (b->d_phase[i]) ok = hipMemcpy(np[i], b->d_phase[i], b->phase_cap * sizeof(float2), hipMemcpyDeviceToDevice) == hipSuccess;
    if (ok && b->d_qphase) ok = hipMemcpy(nq, b->d_qphase, b->phase_cap * sizeof(short2), hipMemcpyDeviceToDevice) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      for (int i = 0; i < XL_NTAB; ++i)
        if (np[i]) (void)hipFree(np[i]);
      if (nq) (void)hipFree(nq);
      return rollback(-ENOMEM);
    }
    for (int i = 0; i < XL_NTAB; ++i) {
      if (b->d_phase[i]) (void)hipFree(b->d_phase[i]);
      b->d_phase[i] = np[i];
    }
    if (b->d_qphase) (void)hipFree(b->d_qphase);
    b->d_qphase = nq;
    b->phase_cap = ncap;
  }
  {
    const float2 one = make_float2(1.0f, 0.0f);
    const short2 qone = make_short2(INT16_MAX, 0);  // xlating.c:546-547
    if (hipMemcpy(b->d_phase[b->pcur] + id, &one, sizeof(one), hipMemcpyHostToDevice) != hipSuccess) return rollback(-EIO);
    if (hipMemcpy(b->d_qphase + id, &qone, sizeof(qone), hipMemcpyHostToDevice) != hipSuccess) return rollback(-EIO);
  }
  return id;
}

extern "C" int xlating_batch_remove_client(xlating_batch *b, int id) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive) return -EINVAL;
  b->clients[id].alive = false;
  b->clients[id].rt.clear();
  xl_row_free(b, b->clients[id].out_off, b->clients[id].row_len);
  b->clients[id].row_len = 0;
  b->nalive--;
  b->dirty = true;
  return 0;
}

// (Re)build the resident plan: classes, tiles (register-tile heights 8/4/2/1), groups, tap image, NCO table.
// NCO riders (xl_kernels.hip) pay in a window: the launch is one round of workgroups (all resident at once -- in a
// multi-round launch the dispatcher evens things out by itself and riders, which sit in one XCD's share of the work
// list, measured 5-12 % slower), and the FIR work of a SIMD clearly outlasts the chain (~18.5 ns per output; when the
// chain is the critical path -- short filters, few clients -- it is quicker alone in workgroups of its own: no
// staging first, 16 lanes).  Measured at 505 taps, D = 42: 384..1024 clients 3-8 % faster with riders; 101 taps
// 25 % slower.
static bool xl_riders_window(size_t wgs, int nw, uint32_t Tpad, int ct, uint32_t K, size_t lds, int min_wgs) {
  const size_t cap = 256 * std::max<size_t>(1, std::min<size_t>((160 * 1024) / std::max<size_t>(lds, 1), 7));
  const double fir_us = (double)wgs * nw / 1024.0 * (double)Tpad * ct * 8.0 / 2000.0;  // 4-cycle packed FMAs at ~2 GHz
  const double chain_us = 0.0185 * (double)K;
  return min_wgs <= 1 || (wgs >= (size_t)min_wgs && wgs <= cap && fir_us >= 1.3 * chain_us);
}

static const int kHeights[XL_NLAUNCH] = {12, 10, 9, 8, 4, 2, 1};

// Direct classes over the clients selected by `use`: key (D, T, consumed mod D, valid history).  A mature client's
// windows never reach below its join point, so all mature clients of one grid share a class whatever their age.
static void xl_direct_classes(const xlating_batch *b, const std::vector<bool> &use, std::vector<DirectClass> *out) {
  out->clear();
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, size_t> cls_of;
  for (size_t i = 0; i < b->clients.size(); ++i) {
    const Client &c = b->clients[i];
    if (!c.alive || !use[i]) continue;
    const uint32_t rem = (uint32_t)(c.consumed % c.D);
    const uint32_t hv = xl_mature(c) ? XL_HCAP : (uint32_t)c.consumed;
    auto key = std::make_tuple(c.D, c.T, rem, hv);
    auto it = cls_of.find(key);
    if (it == cls_of.end()) {
      it = cls_of.emplace(key, out->size()).first;
      out->push_back(DirectClass{c.D, c.T, rem, hv, {}});
    }
    (*out)[it->second].members.push_back((int)i);
  }
}

// Builds one set of direct-FIR launches over `classes`: tiles, groups, tap image rows (appended to `image`).
static int xl_build_launches(xlating_batch *b, Launch *Ls, const std::vector<DirectClass> &classes, int big_h,
                             std::vector<float> *image, std::vector<double> *imageq) {
  struct TileDesc {
    size_t cls;
    std::vector<int> ids;
  };
  std::vector<TileDesc> tiles_of[XL_NLAUNCH];
  const uint32_t cap_samples = b->max_samples * b->gcap;
  for (int li = 0; li < XL_NLAUNCH; ++li) {
    Ls[li].ct = kHeights[li];
    Ls[li].lds = 0;
    Ls[li].nw = XL_NW_DEFAULT;
    Ls[li].all_wide = true;
    Ls[li].maxD = 1;
    Ls[li].minD = 0xFFFFFFFFu;
  }
  for (size_t k = 0; k < classes.size(); ++k) {
    const std::vector<int> &m = classes[k].members;
    int h = big_h;
    if (m.size() < 8) h = m.size() > 4 ? 8 : (m.size() > 2 ? 4 : (m.size() > 1 ? 2 : 1));
    int li = 0;
    while (kHeights[li] != h) ++li;
    for (size_t next = 0; next < m.size(); next += (size_t)h) {
      const size_t cnt = std::min<size_t>((size_t)h, m.size() - next);
      tiles_of[li].push_back(TileDesc{k, std::vector<int>(m.begin() + next, m.begin() + next + cnt)});
    }
  }

  for (int li = 0; li < XL_NLAUNCH; ++li) {
    Launch &L = Ls[li];
    if (tiles_of[li].empty()) continue;
    const int ct = L.ct;
    int gi = -1;
    size_t gcls = 0;
    for (const TileDesc &td : tiles_of[li]) {
      const DirectClass &cs = classes[td.cls];
      const uint32_t Tpad = xl_roundup(cs.T, xl_tap_step(ct));
      if (gi < 0 || gcls != td.cls || L.groups[gi].ntiles == (uint32_t)L.nw) {
        L.groups.emplace_back();
        gi = (int)L.groups.size() - 1;
        gcls = td.cls;
        XlGroup *g = &L.groups[gi];
        memset(g, 0, sizeof(*g));
        g->D = cs.D;
        g->T = cs.T;
        g->Tpad = Tpad;
        g->rem0 = cs.rem0;
        g->hv0 = cs.hv0;
        g->wide = (cs.D % 2 == 0) ? 1u : 0u;
        if (!g->wide) L.all_wide = false;
        L.lds = std::max(L.lds, xl_fir_lds_bytes_ota(cs.D, Tpad, 64));
        L.maxD = std::max(L.maxD, cs.D);
        L.minD = std::min(L.minD, cs.D);
      }
      XlGroup *g = &L.groups[gi];
      XlTile &t = g->tiles[g->ntiles++];
      const uint32_t real_off = (uint32_t)(image->size() / 2);
      t.tap_off = real_off;
      t.nclients = (uint32_t)td.ids.size();  // a partial last tile keeps zero taps for the missing clients
      image->resize(image->size() + (size_t)2 * Tpad * ct, 0.0f);
      if (imageq) imageq->resize(image->size(), 0.0);
      float *dst = image->data() + (size_t)2 * real_off;
      for (size_t j = 0; j < td.ids.size(); ++j) {
        const Client &c = b->clients[td.ids[j]];
        t.out_off[j] = c.out_off;
        t.incr[j] = make_float2(c.incr[0], c.incr[1]);
        t.qincr[j] = (uint32_t)(uint16_t)c.qincr[0] | ((uint32_t)(uint16_t)c.qincr[1] << 16);
        for (uint32_t i = 0; i < cs.T; ++i) {
          dst[((size_t)i * ct + j) * 2] = c.rt[2 * i];
          dst[((size_t)i * ct + j) * 2 + 1] = c.rt[2 * i + 1];
          if (imageq) {
            (*imageq)[(size_t)2 * real_off + ((size_t)i * ct + j) * 2] = (double)c.rtq[2 * i];
            (*imageq)[(size_t)2 * real_off + ((size_t)i * ct + j) * 2 + 1] = (double)c.rtq[2 * i + 1];
          }
        }
      }
    }
  }
  // ---- spare waves for the NCO riders (xl_kernels.hip): groups with fewer tiles than the launch has waves.  The
  // launch that carries the role (the first one with groups) gets a spare wave by splitting its last full group
  // into 3 + 1 tiles when it has none and the engine is big enough for the balance to matter.
  {
    bool first = true;
    for (int lq = 0; lq < XL_NLAUNCH; ++lq) {
      Launch &L = Ls[lq];
      if (L.groups.empty()) continue;
      uint32_t idle = 0;
      for (const XlGroup &g : L.groups) idle += (uint32_t)L.nw - g.ntiles;
      const uint32_t kest = cap_samples / L.groups[0].D + 1;
      if (first && idle == 0 && b->riders && L.nw == XL_NW_MAX &&
          xl_riders_window((L.groups.size() + 1) * ((kest + 63) / 64), L.nw, L.groups[0].Tpad, L.ct, kest, L.lds,
                           b->riders_min_wgs)) {
        XlGroup &last = L.groups.back();
        XlGroup extra = last;
        extra.ntiles = 1;
        extra.tiles[0] = last.tiles[XL_NW_MAX - 1];
        last.ntiles = XL_NW_MAX - 1;
        L.groups.push_back(extra);
      }
      // (riders are only used in one-round launches -- xl_riders_window -- where every workgroup is dispatched within
      // ~10 us of the start, so the groups with spare waves can stay where they are: last, which suits the tail)
      idle = 0;
      for (XlGroup &g : L.groups) {
        g.idle_before = idle;
        idle += (uint32_t)L.nw - g.ntiles;
      }
      L.idle_waves = idle;
      first = false;
    }
  }
  for (int lq = 0; lq < XL_NLAUNCH; ++lq) {
    Launch &L = Ls[lq];
    L.ota = 64;
    if (L.lds > 160 * 1024) {  // huge decimation: fewer active lanes per wave so that the window image fits
      for (L.ota = 32; L.ota >= 8; L.ota >>= 1) {
        size_t need = 0;
        for (const XlGroup &g : L.groups) need = std::max(need, xl_fir_lds_bytes_ota(g.D, g.Tpad, L.ota));
        if (need <= 160 * 1024) {
          L.lds = need;
          break;
        }
      }
      if (L.ota < 8) return -EINVAL;  // (add_client already refused such a shape)
    }
  }
  return 0;
}

// Direct FIR launches whose own work is short against the NCO chain (~25-32 us per block) gain from the side-stream chain
// kernel on reserved CUs like the polyphase launches do; heavier ones hide the chain in their spare waves for free and
// would only lose the reserved CUs.  Measured, 8 blocks per call, us per block fused -> side: 128 clients x 101 taps
// (40 M complex MACs per block) 29.9 -> 27.1; 128 x 505 native (202 M) 40.5 -> 35.5; 1024 x 101 (323 M) 37.9 -> 40.4;
// 1024 x 505 native (1615 M) 203 -> 227.
static bool xl_direct_is_light(double macs_per_block) { return macs_per_block < 250e6; }

// Uploads one launch set: its tap image (+ the Q15 image) and its group descriptors.
static int xl_upload_launch_set(xlating_batch *b, Launch *set, const std::vector<float> &image, float2 **d_image,
                                const std::vector<double> *imageq) {
  if (!image.empty()) {
    XL_TRY(xl_plan_alloc(b, (void **)d_image, image.size() * sizeof(float) + 256));
    XL_TRY(hipMemcpy(*d_image, image.data(), image.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (imageq && !imageq->empty()) {
    XL_TRY(xl_plan_alloc(b, (void **)&b->d_qtaps, imageq->size() * sizeof(double) + 256));
    XL_TRY(hipMemcpy(b->d_qtaps, imageq->data(), imageq->size() * sizeof(double), hipMemcpyHostToDevice));
  }
  for (int i = 0; i < XL_NLAUNCH; ++i) {
    Launch &L = set[i];
    if (L.groups.empty()) continue;
    XL_TRY(xl_plan_alloc(b, (void **)&L.d_groups, L.groups.size() * sizeof(XlGroup)));
    XL_TRY(hipMemcpy(L.d_groups, L.groups.data(), L.groups.size() * sizeof(XlGroup), hipMemcpyHostToDevice));
  }
  return 0;
fail:
  return xl_errno_of_last_hip_error();
}

// The all-clients launch set (native and Q15 calls, optimized calls too small for the polyphase path): built when a call
// first needs it after the client set changed.  Its tap image is the one thing of a plan whose cost grows with every
// client (T x 8 bytes each, gathered tap-major on the host and uploaded).
static int xl_batch_build_all_set(xlating_batch *b) {
  if (b->all_built) return 0;
  xl_batch_sync_all(b);  // (launches in flight may still read the previous image)
  xl_release_launch_set(b, b->launches);
  xl_plan_release(b, b->d_taps);
  xl_plan_release(b, b->d_qtaps);
  xl_plan_release(b, b->d_qinc);
  b->d_taps = nullptr;
  b->d_qtaps = nullptr;
  b->d_qinc = nullptr;
  std::vector<float> image;
  std::vector<double> imageq;
  const bool has_q15 = b->fmt != XL_FMT_CF32 && b->want_q15;
  int rc = xl_build_launches(b, b->launches, b->classes, b->big_h, &image, has_q15 ? &imageq : nullptr);
  if (rc != 0) return rc;
  rc = xl_upload_launch_set(b, b->launches, image, &b->d_taps, has_q15 ? &imageq : nullptr);
  if (rc != 0) return rc;
  if (has_q15) {
    std::vector<uint32_t> qinc;
    for (const XlNcoClient &nc : b->nco) {
      const Client &c = b->clients[nc.slot];
      qinc.push_back((uint32_t)(uint16_t)c.qincr[0] | ((uint32_t)(uint16_t)c.qincr[1] << 16));
    }
    XL_TRY(xl_plan_alloc(b, (void **)&b->d_qinc, qinc.size() * sizeof(uint32_t)));
    XL_TRY(hipMemcpy(b->d_qinc, qinc.data(), qinc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  xl_plan_trim(b);
  b->all_built = true;
  return 0;
fail:
  return xl_errno_of_last_hip_error();
}

// Transform length of a polyphase class: the mix launch streams D x M branch-spectrum values per client and call from HBM,
// which is what bounds it with many clients and one block per call; M = 128 halves that for ~5-10 % more arithmetic
// (valid outputs per segment M - A + 1) while the filter is short against the segment.  Measured at D = 42, 505 taps, one
// block per call: x1.17 at 4096 clients, x1.08 at 2048, x1.015 at 1024, x0.99 at 512 and below.
// M = 64 (round 6): classes of more than 64 branches (9+ k-blocks of 8: the wide two-half mix, D = 65 .. 112; the float32 mixes above
// that or on request) with up to 8 taps per branch (57+ of 64 outputs per segment valid).  A wide workgroup holds 104 KB of operands for ONE bin of 128 columns, and half the bins is half the workgroups
// and half the operand stream (config 5 at 1024 clients: 512 workgroups = ONE round instead of two): config 5 (cf32, D = 100, 3 taps
// per branch) 8 blocks per call at 1024 / 2048 / 4096 clients 16.9 / 29.1 / 51.0 -> 16.0 / 26.2 / 47.4 us per block, ONE block per call
// 48.4 -> 40.8 (4096 clients: 138 -> 94); D = 72 / 100 off cu8 streams with 3 / 5 / 8 taps per branch: ahead or level at 1024 and 4096
// clients; at 128-768 clients level with 128 and ahead of 256 (which the rule above picked there: 12.0 / 39.4 against 11.9 / 29.1 us
// per block at 256 clients x 8 / 1 blocks per call).  Narrow classes (D <= 64) LOSE with 64 points (D = 64, 4096 clients: 59.4 -> 65.6):
// their workgroups hold less and the inverse launch, whose tiles stay 32 KB, gains nothing.  The float32 mixes gain too (config 5 with
// mix_kernel = 3: -4 %; D = 128 / 200 on the streamed kernel: -9 / -17 % at 1024 clients, -6 / -11 % at 256); 10-13 taps per branch: level
// (not taken).  profiles/r06_transform_length_64.txt
static uint32_t xl_poly_pick_m(const xlating_batch *b, uint32_t A, size_t members, uint32_t D) {
  if (A > 64) return 256u;
  if (b->poly_m) return A > 32 && b->poly_m == 64u ? 128u : b->poly_m;  // (forced; a class needs A <= M / 2)
  const uint32_t nkb = (D + 7u) / 8u;
  if (nkb > XLP_NKB_4W && A <= 8) return 64u;
  return A <= 32 && members >= 768 ? 128u : 256u;
}

// Which mix launch a class of D branches takes (PolyClass::mix_kind): the two-half kernel (1) carries the spectra as pairs of halves
// -- bounded by the input format, or (cf32 streams) scaled per segment by what the forward launch found (PolyClass::d_segmax) -- and
// holds at most XLP_NKB_MAX k-blocks of 8 branches (D <= 112); the float32 matrix instruction (3) has no such condition: it is what
// D > 112 takes, and every class on request (option "mix_kernel" = 3: all-float32 products).
static uint32_t xl_poly_mix_kind(const xlating_batch *b, uint32_t D) {
  const bool halves_ok = D <= 8u * XLP_NKB_MAX;
  return (b->mix_kernel == 3u || !halves_ok) ? 3u : 1u;
}

// Power-of-two scale of a column's branch spectra for the matrix-core mix: every component of R_b[m] = sum_a r_b[a] e^{..} is at
// most L = max_b sum_a |r_b[a]| (the same for the delayed taps: a delay permutes the branches); scale = 2^floor(log2(RMAX / L)).
static float xl_poly_col_scale(const Client &c, uint32_t D, uint32_t T) {
  std::vector<double> l1(D, 0.0);
  for (uint32_t i = 0; i < T; ++i) l1[i % D] += hypot((double)c.rt[2 * i], (double)c.rt[2 * i + 1]);
  double L = 0.0;
  for (double v : l1) L = std::max(L, v);
  if (!(L > 0.0) || !std::isfinite(L)) return 1.0f;
  int e = (int)floor(log2((double)XLP_H_RMAX / L));
  e = std::max(-100, std::min(100, e));
  return (float)ldexp(1.0, e);
}

// Brings the device images of a polyphase class in line with its member list: columns, branch spectra of the NEW columns.
static int xl_poly_sync_device(xlating_batch *b, PolyClass &pc, const std::vector<uint32_t> &new_cols, bool fresh, uint32_t cap_samples) {
  // the images a growing class moves into: owned by nobody until they are handed to `pc` below -- a failure in between gives
  // them back (fail:)
  float2 *nY = nullptr;
  void *nRh = nullptr;
  float *ncs = nullptr;
  XlpCol *ncols = nullptr;
  // ---- capacity: column groups (Rh, Y, cols) and segments (Y, X)
  const uint32_t need_cg = ((uint32_t)pc.col_client.size() + XLP_COLS - 1) / XLP_COLS;
  const uint32_t nseg_cap = (cap_samples / pc.D + 2 + pc.V - 1) / pc.V + 1;
  const uint32_t passes = (nseg_cap + XLP_SEG - 1) / XLP_SEG;
  if (fresh || need_cg > pc.ncg_cap || nseg_cap != pc.nseg_cap) {
    // grow by an eighth (at least one group) so that the next joins find room; the old spectra move over on the device
    const uint32_t cap = fresh ? need_cg : std::max(need_cg, pc.ncg_cap + std::max(1u, pc.ncg_cap / 8u));
    // operand-form image, a group's part contiguous: the old groups are one copy, the new ones start as zeros (empty columns are
    // multiplied into sums that are never stored, but must be finite)
    const size_t per_cg = pc.mix_kind == 3u ? xlmf_rf_bytes_per_group(pc.M, pc.nkb) : xlp_rh_bytes_per_group(pc.M, pc.nkb);
    XL_TRY(xl_plan_alloc(b, &nRh, (size_t)cap * per_cg));
    const size_t old_bytes = fresh ? 0 : (size_t)pc.ncg_cap * per_cg;
    if (old_bytes) XL_TRY(hipMemcpyAsync(nRh, pc.d_Rh, old_bytes, hipMemcpyDeviceToDevice, b->own_stream));
    XL_TRY(hipMemsetAsync((char *)nRh + old_bytes, 0, (size_t)cap * per_cg - old_bytes, b->own_stream));
    XL_TRY(xl_plan_alloc(b, (void **)&ncs, (size_t)cap * XLP_COLS * sizeof(float)));
    XL_TRY(xl_plan_alloc(b, (void **)&nY, (size_t)cap * nseg_cap * pc.M * XLP_COLS * sizeof(float2)));
    XL_TRY(xl_plan_alloc(b, (void **)&ncols, (size_t)cap * XLP_COLS * sizeof(XlpCol)));
    XL_TRY(hipStreamSynchronize(b->own_stream));
    xl_plan_release(b, pc.d_Rh);
    xl_plan_release(b, pc.d_cscale);
    xl_plan_release(b, pc.d_Y);
    xl_plan_release(b, pc.d_cols);
    pc.d_Rh = nRh, pc.d_cscale = ncs, pc.d_Y = nY, pc.d_cols = ncols;
    nY = nullptr, nRh = nullptr, ncs = nullptr, ncols = nullptr;
    pc.ncg_cap = cap;
    if (fresh || nseg_cap != pc.nseg_cap || pc.d_X == nullptr) {
      xl_plan_release(b, pc.d_X);
      pc.d_X = nullptr;
      // (X: the padding branches and the unused segment slots of the last pass must be finite: cleared once)
      const size_t xbytes = (size_t)passes * pc.Dpad * pc.M * XLP_XS * sizeof(float2);
      XL_TRY(xl_plan_alloc(b, (void **)&pc.d_X, xbytes));
      XL_TRY(hipMemsetAsync(pc.d_X, 0, xbytes, b->own_stream));
      xl_plan_release(b, pc.d_segmax);
      pc.d_segmax = nullptr;
      pc.seg_cap = passes * XLP_SEG, pc.seg_par = 0;
      if (pc.mix_kind == 1u && b->fmt == XL_FMT_CF32) {
        const size_t sbytes = 2u * (size_t)pc.seg_cap * XLP_SEGMAX_STRIDE * sizeof(uint32_t);
        XL_TRY(xl_plan_alloc(b, (void **)&pc.d_segmax, sbytes));
        XL_TRY(hipMemsetAsync(pc.d_segmax, 0, sbytes, b->own_stream));
      }
    }
    pc.nseg_cap = nseg_cap;
  }
  pc.ncg = need_cg;
  // ---- per-column records (16 bytes each: all of them, every plan)
  {
    std::vector<XlpCol> cols((size_t)pc.ncg_cap * XLP_COLS);
    for (XlpCol &cc : cols) {
      cc.out_off = 0xFFFFFFFFu;
      cc.delta = 0;
      cc.incr = make_float2(0.0f, 0.0f);
    }
    for (size_t j = 0; j < pc.col_client.size(); ++j) {
      if (pc.col_client[j] < 0) continue;
      const Client &c = b->clients[pc.col_client[j]];
      cols[j].out_off = c.out_off;
      cols[j].delta = pc.col_delta[j];
      cols[j].incr = make_float2(c.incr[0], c.incr[1]);
    }
    XL_TRY(hipMemcpy(pc.d_cols, cols.data(), cols.size() * sizeof(XlpCol), hipMemcpyHostToDevice));
    // what the two-half mix multiplies a column's sums by: 1 / (its scale * the spectra's).  (float32 operands are neither scaled nor
    // split: the table stays at 1 and nobody reads it)
    std::vector<float> cs((size_t)pc.ncg_cap * XLP_COLS, 1.0f);
    pc.col_scale.resize(pc.col_client.size(), 1.0f);
    if (pc.mix_kind == 1u) {
      for (uint32_t j : new_cols) pc.col_scale[j] = xl_poly_col_scale(b->clients[pc.col_client[j]], pc.D, pc.T);
      for (size_t j = 0; j < pc.col_client.size(); ++j) {
        if (pc.col_client[j] < 0) continue;
        cs[j] = b->fmt == XL_FMT_CF32 ? 1.0f / pc.col_scale[j] : 1.0f / (pc.col_scale[j] * XLP_H_XSCALE);  // (cf32: the segments' own scales, in the kernel)
      }
    }
    XL_TRY(hipMemcpy(pc.d_cscale, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // ---- branch spectra of the new columns (device kernel, double arithmetic)
  if (!new_cols.empty()) {
    const size_t nn = new_cols.size();
    std::vector<float> rt(nn * pc.T * 2);  // [tap][new column]
    std::vector<uint32_t> meta(3 * nn);    // [new column]: delay, then column index, then (two-half mix) the column scale's bits
    for (size_t j = 0; j < nn; ++j) {
      const Client &c = b->clients[pc.col_client[new_cols[j]]];
      meta[j] = pc.col_delta[new_cols[j]];
      meta[nn + j] = new_cols[j];
      const float sc = pc.col_scale[new_cols[j]];
      memcpy(&meta[2 * nn + j], &sc, sizeof(float));
      for (uint32_t i = 0; i < pc.T; ++i) {
        rt[((size_t)i * nn + j) * 2] = c.rt[2 * i];
        rt[((size_t)i * nn + j) * 2 + 1] = c.rt[2 * i + 1];
      }
    }
    float2 *d_rt = nullptr;
    uint32_t *d_meta = nullptr;
    XL_TRY(xl_plan_alloc(b, (void **)&d_rt, rt.size() * sizeof(float)));
    hipError_t e = xl_plan_alloc(b, (void **)&d_meta, meta.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_rt, rt.data(), rt.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_meta, meta.data(), meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = pc.mix_kind == 3u
              ? xlp_launch_tables_f(d_rt, d_meta, d_meta + nn, (uint32_t)nn, pc.T, pc.D, pc.A, pc.M, pc.nkb, pc.d_Rh, b->own_stream)
              : xlp_launch_tables_h(d_rt, d_meta, d_meta + nn, reinterpret_cast<const float *>(d_meta + 2 * nn), (uint32_t)nn, pc.T,
                                    pc.D, pc.A, pc.M, pc.nkb, pc.d_Rh, b->own_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->own_stream);
    xl_plan_release(b, d_rt);  // (scratch: spare again at once)
    xl_plan_release(b, d_meta);
    if (e != hipSuccess) {
      xl_last_hip_error = e;
      goto fail;
    }
  }
  return 0;
fail : {
  void *unowned[] = {nRh, ncs, nY, ncols};
  for (void *q : unowned) xl_plan_release(b, q);
}
  return xl_errno_of_last_hip_error();
}

// (Re)build the resident plan.  INCREMENTAL where it matters: clients keep their output rows, polyphase classes keep their
// columns and branch spectra (a join computes one column), the all-clients launch set is built only when a call needs it;
// rebuilt every time: the class lists, the optimized-mode direct set (the few clients outside the polyphase classes) and the
// small per-client records.
static int xl_batch_plan(xlating_batch *b) {
  // tuning: XL_EXP_PLAN_TIMING=1 prints where a re-plan spends its time
  static const bool plan_timing = xl_exp_getenv("XL_EXP_PLAN_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!plan_timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "plan: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  xl_batch_sync_all(b);
  lap("sync");
  b->spec_n = 0;
  const uint32_t advanced = b->trel;  // samples since the previous plan: every plan-time stream record moves by this much
  xl_batch_free_plan(b, false);
  b->classes.clear();
  b->classes_rest.clear();
  b->nco.clear();
  b->trel = 0;
  b->calls_since_plan = 0;
  b->planned_immature = 0;
  b->plan_maxD = 1;
  const uint32_t cap_samples = b->max_samples * b->gcap;
  for (size_t i = 0; i < b->clients.size(); ++i) {
    Client &c = b->clients[i];
    if (!c.alive) continue;
    c.planned_mature = xl_mature(c);
    if (!c.planned_mature) b->planned_immature++;
    b->plan_maxD = std::max(b->plan_maxD, c.D);
    XlNcoClient nc;
    memset(&nc, 0, sizeof(nc));
    nc.incr = make_float2(c.incr[0], c.incr[1]);
    nc.out_off = c.out_off;
    nc.slot = (uint32_t)i;
    nc.D = c.D;
    nc.rem0 = (uint32_t)(c.consumed % c.D);
    b->nco.push_back(nc);
  }
  b->out_total = b->rows_end;

  // ---- polyphase classes (optimized mode): all mature clients of one (D, T) -- many clients (its lanes are client
  // columns and its cost per client does not depend on the tap count) with a filter long enough to be worth it
  std::vector<bool> all_use(b->clients.size(), true), rest_use(b->clients.size(), true);
  std::vector<PolyClass> next_poly;
  struct Pending {
    size_t idx;
    std::vector<uint32_t> new_cols;
    bool fresh;
  };
  std::vector<Pending> pending;
  for (PolyClass &pc : b->poly) pc.keep = false;
  {
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::vector<int>> by_shape;
    for (size_t i = 0; i < b->clients.size(); ++i) {
      const Client &c = b->clients[i];
      if (c.alive) by_shape[std::make_tuple(c.D, c.T, c.planned_mature ? XL_HCAP : (uint32_t)c.consumed)].push_back((int)i);
    }
    for (auto &kv : by_shape) {
      const uint32_t D = std::get<0>(kv.first), T = std::get<1>(kv.first), hv0 = std::get<2>(kv.first);
      const std::vector<int> &m = kv.second;
      std::vector<uint32_t> rems;
      for (int id : m) rems.push_back((uint32_t)(b->clients[id].consumed % D));
      std::vector<uint32_t> distinct(rems);
      std::sort(distinct.begin(), distinct.end());
      distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
      // An existing class of this shape whose members are (mostly) still here: same (D, T), same kind (mature), or the
      // immature class these very clients formed when they joined together.  Its shared grid moved with the stream.
      PolyClass *old = nullptr;
      for (PolyClass &oc : b->poly) {
        if (oc.keep || oc.D != D || oc.T != T) continue;
        const bool same_kind = oc.hv0 == hv0 || (hv0 == XL_HCAP && oc.hv0 != XL_HCAP && !m.empty() && oc.col_of.count(m[0]));
        if (same_kind) {
          old = &oc;
          break;
        }
      }
      uint32_t ref = 0, dmax = 0;
      bool reuse = false;
      if (old != nullptr) {
        ref = (old->rem_ref0 + advanced % D) % D;
        for (uint32_t r : distinct) dmax = std::max(dmax, (ref + D - r) % D);
        const uint32_t A = (T + dmax + D - 1) / D;
        reuse = A == old->A && xl_poly_pick_m(b, A, m.size(), D) == old->M && xl_poly_mix_kind(b, D) == old->mix_kind;
      }
      if (!reuse) {
        // the shared grid's reference: the member offset that keeps the largest delay of a member smallest
        uint32_t best_ref = distinct[0], best_dmax = 0xFFFFFFFFu;
        for (uint32_t cand : distinct) {
          uint32_t dm = 0;
          for (uint32_t r : distinct) dm = std::max(dm, (cand + D - r) % D);  // delta = (j0_c - j0_ref) mod D = (rem_ref - rem_c) mod D
          if (dm < best_dmax) best_dmax = dm, best_ref = cand;
        }
        ref = best_ref, dmax = best_dmax;
      }
      const uint32_t A = (T + dmax + D - 1) / D;
      const uint32_t M = xl_poly_pick_m(b, A, m.size(), D);
      const bool fits = A >= 2 && A <= M / 2 && D <= 504;
      // crossover: with the mix on the matrix cores the path costs the same whatever the filter length and little beside the
      // recurrence in small classes (A/B at 8 blocks per call, direct kernel -> polyphase, us per block: 101 taps 37.3 -> 28.2 at
      // 1024 clients, 124.8 -> 88.7 at 4096, 23.3 -> 22.9 at 128; 505 taps 24.8 -> 22.9 at 96 clients, 23.1 -> 22.7 at 32; cf32 10
      // Msps, D = 100, 257 taps: 38.4 -> 23.4 at 1024 clients, 12.8 -> 11.8 at 256, 11.4 -> 11.3 at 64): 2 taps per branch, 32 clients
      // Classes of more than XLMF_NB8_MAX k-blocks (D > 112: float32 operands re-streamed every pass, xlp_mix_f32_stream_kernel): the
      // same 2 taps per branch from 128 clients on -- measured in round 6 at D = 128 / 200 / 400, 1.2 / 2.4 / 4.8 / 12 taps per branch,
      // 32 .. 1024 clients (profiles/r06_plan_rules_other_shapes.txt): the path costs 10.6-11.5 us per block up to 128 clients whatever
      // the filter, the direct kernel 11.4-11.9 at 128 clients x 2.4 taps per branch (1.04-1.07 x) and 47-62 at 1024 (1.6-2.1 x; rounds
      // 4-5 sent those to the direct kernel: 4.5 taps per branch was the only crossover a measurement of that kernel stood behind);
      // at 64 clients the direct kernel is still ahead up to 4.8 taps per branch.
      const bool streamed = (D + 7u) / 8u > XLMF_NB8_MAX;
      const size_t min_clients = b->poly_min_set ? b->poly_min_clients : (streamed ? 128u : 32u);
      const bool pays = m.size() >= min_clients && T >= 2 * D;
      if (b->poly_mode == 0 || !fits || (b->poly_mode < 0 && !pays)) continue;
      PolyClass pc;
      Pending pd;
      pd.fresh = !reuse;
      if (reuse) {
        pc = std::move(*old);
        old->keep = true;
        old->d_X = old->d_Y = nullptr;
        old->d_segmax = nullptr;
        old->d_cols = nullptr;
        old->d_Rh = nullptr;
        old->d_cscale = nullptr;
        // members that left give their columns back
        std::vector<bool> here(b->clients.size(), false);
        for (int id : m) here[id] = true;
        for (auto it = pc.col_of.begin(); it != pc.col_of.end();) {
          if (!here[it->first]) {
            pc.col_client[it->second] = -1;
            it = pc.col_of.erase(it);
          } else {
            ++it;
          }
        }
        while (!pc.col_client.empty() && pc.col_client.back() < 0) {  // (trailing free columns shrink the class)
          pc.col_client.pop_back();
          pc.col_delta.pop_back();
          pc.col_uid.pop_back();
        }
        pc.col_scale.resize(pc.col_client.size(), 1.0f);
      } else {
        pc.D = D;
        pc.Dpad = xl_roundup(D, XLP_BSTEP);
        pc.T = T;
        pc.A = A;
        pc.M = M;
        pc.V = M - A + 1;
        pc.mix_kind = xl_poly_mix_kind(b, D);
        pc.nkb = (D + 7u) / 8u;
      }
      pc.keep = false;
      pc.rem_ref0 = ref;
      pc.hv0 = hv0;
      pc.dmax = dmax;
      pc.members = m;
      // newcomers (and, for a recycled client id, a changed delay) take the free columns first, then new ones
      size_t next_free = 0;
      for (int id : m) {
        const uint32_t delta = (ref + D - (uint32_t)(b->clients[id].consumed % D)) % D;
        auto it = pc.col_of.find(id);
        if (it != pc.col_of.end() && pc.col_delta[it->second] == delta && pc.col_uid[it->second] == b->clients[id].uid) continue;
        uint32_t col;
        if (it != pc.col_of.end()) {
          col = it->second;
        } else {
          while (next_free < pc.col_client.size() && pc.col_client[next_free] >= 0) ++next_free;
          if (next_free == pc.col_client.size()) {
            pc.col_client.push_back(-1);
            pc.col_delta.push_back(0);
            pc.col_uid.push_back(0);
          }
          col = (uint32_t)next_free;
          pc.col_client[col] = id;
          pc.col_of[id] = col;
        }
        pc.col_delta[col] = delta;
        pc.col_uid[col] = b->clients[id].uid;
        pd.new_cols.push_back(col);
      }
      pc.ncols = (uint32_t)pc.col_client.size();
      pd.idx = next_poly.size();
      next_poly.push_back(std::move(pc));
      pending.push_back(std::move(pd));
      for (int id : m) rest_use[id] = false;
    }
  }
  for (PolyClass &oc : b->poly)
    if (!oc.keep) xl_poly_release(b, oc);
  b->poly = std::move(next_poly);
  xl_direct_classes(b, all_use, &b->classes);
  if (!b->poly.empty()) xl_direct_classes(b, rest_use, &b->classes_rest);
  lap("classes");

  // ---- register-tile height.  Every wave does the same work (64 outputs x H clients x T taps) and a CU holds
  // floor(160 KiB / window image) workgroups of 4 waves (6 at the server-default shape).  A launch whose workgroups
  // do not all fit runs the surplus in a second round almost alone -- latency-bound, about one lone-workgroup
  // duration (measured 48 us of a 161 us launch at 1024 clients with H = 8: 32 x 49 = 1568 workgroups on 1536
  // slots).  Taller tiles trade a little per-wave time for fewer workgroups: pick the height whose launch costs
  // least in (waves on the busiest SIMD) x (work per wave).  Classes with fewer than 8 clients use one small
  // tile.
  int big_h = 8;
  {
    size_t lds1 = 0;
    for (const DirectClass &cs : b->classes)
      if (cs.members.size() >= 8) lds1 = std::max(lds1, xl_fir_lds_bytes_ota(cs.D, xl_roundup(cs.T, 12), 64));
    if (lds1 > 0) {
      const long slots = std::max<long>(1, std::min<long>((long)(160 * 1024 / lds1), 7));
      const long cap = slots * 256;
      long best = -1;
      static const int cand[4] = {8, 9, 10, 12};
      for (int h : cand) {
        long wgs = 0;
        for (const DirectClass &cs : b->classes) {
          const long n = (long)cs.members.size();
          if (n < 8) continue;
          const long kest = b->max_samples / cs.D + 1;
          wgs += (((n + h - 1) / h + XL_NW_MAX - 1) / XL_NW_MAX) * ((kest + 63) / 64);
        }
        const long full = wgs / cap, rem = wgs % cap;
        long cost = full * slots * h;
        if (rem) cost += std::max<long>((rem + 255) / 256, 3) * h;
        if (best < 0 || cost < best) {
          best = cost;
          big_h = h;
        }
      }
    }
    if (b->exp_h == 8 || b->exp_h == 9 || b->exp_h == 10 || b->exp_h == 12) big_h = b->exp_h;
  }
  b->big_h = big_h;
  std::vector<float> image_rest;  // tap image of the optimized-mode launch set
  if (!b->poly.empty()) {
    int rc = xl_build_launches(b, b->launches_rest, b->classes_rest, big_h, &image_rest, nullptr);
    if (rc != 0) {
      // (b->poly already holds the updated classes whose new columns have no branch spectra yet: a plan that stops here must not
      // be reused -- drop it like every other failure does)
      xl_batch_free_plan(b, true);
      b->dirty = true;
      return rc;
    }
  }
  lap("tiles + tap images (host)");

  // ---- CU reservation for the side-stream chain kernel (64 clients per workgroup = per CU, dealt round-robin to the
  // 8 XCDs): recreate the two masked streams when the number of reserved CUs changes
  {
    const uint32_t nwg = ((uint32_t)b->nco.size() + 63u) / 64u;
    b->macs_all = b->macs_rest = 0.0;
    for (const DirectClass &cs : b->classes) b->macs_all += (double)cs.members.size() * cs.T / cs.D;
    for (const DirectClass &cs : b->classes_rest) b->macs_rest += (double)cs.members.size() * cs.T / cs.D;
    const bool light = xl_direct_is_light(b->macs_all * b->max_samples) || (!b->poly.empty() && xl_direct_is_light(b->macs_rest * b->max_samples));
    // (one-block calls of a polyphase plan take the side stream too, up to XL_SIDE_ONE_BLOCK_MAX clients: see side_call)
    const bool one_block_side = !b->poly.empty() && b->nco.size() <= XL_SIDE_ONE_BLOCK_MAX;
    // (a server that knows how many clients it admits says so -- option "expected_clients" --, and the reservation is made for
    // that many at once: the 25 ms of a stream re-creation then never fall on a call between two joins)
    const uint32_t nwg_res = std::max(nwg, (b->expected_clients + 63u) / 64u);
    // (one CU per chain workgroup while the recurrence bounds the call, none in a band above that, beyond it the chain launch runs in
    // rounds on fewer CUs -- by the plan's load: launch time per unit of chain time, in clients of the measured shape: xl_plan_rules.h)
    uint32_t load_wgs = nwg_res;
    if (!b->poly.empty() && b->gcap >= 2) {  // (engines of one-block calls: the bands as measured by client count -- their calls are short, and nothing else was measured)
      double ps = b->macs_rest * (double)b->max_samples * 72.0;  // (direct-kernel clients of an optimized call: ~0.072 ns per complex MAC)
      uint32_t kmax = 1u;
      for (const PolyClass &pc : b->poly) {
        const uint32_t K = (b->max_samples + pc.D - 1u) / pc.D;
        ps += (double)pc.members.size() * xl_client_launch_ps(pc.M, K, pc.V, 8u * pc.nkb, pc.mix_kind, b->gcap);
        kmax = std::max(kmax, K);
      }
      for (const DirectClass &cs : b->classes_rest) kmax = std::max(kmax, (b->max_samples + cs.D - 1u) / cs.D);
      // (scaled to the population the CUs are reserved for: option "expected_clients")
      load_wgs = xl_plan_load_wgs(ps * (double)nwg_res / (double)std::max(nwg, 1u), kmax);
    }
    // (the bands have edges where the reservation jumps: stay in the band of the previous plan until the load is two workgroups past one)
    load_wgs = xl_chain_load_with_hysteresis(load_wgs, b->reserve_band);
    b->reserve_band = xl_chain_band(load_wgs);
    const bool side_plan = (b->gcap >= 2 || one_block_side) && (!b->poly.empty() || light || b->nco_side > 0) && b->nco_side != 0;
    // (the no-reservation band and the rounds were measured on polyphase plans only; a direct-only plan -- whose side-stream chain needs
    // the masked pair, see side_call -- keeps one CU per chain workgroup, up to half the chip)
    uint32_t want = !side_plan ? 0u : (b->poly.empty() ? ((nwg_res + 7u) / 8u <= 16u ? (nwg_res + 7u) / 8u : 0u) : xl_chain_reserve_per_xcd(nwg_res, load_wgs));
    // A wide two-half class (9 .. 14 k-blocks: xl_mixh2.hip) runs two workgroups per CU and launches M x column groups of them -- a
    // multiple of 512 at every 512 clients: on the 240 CUs a reservation of 16 leaves, BASELINE config 5 at 1024 clients took a third,
    // quarter-full round of workgroups (56 us per mix launch; the launch's own timeline: profiles/r06_mix_wide_timeline.txt) -- and its
    // forward and inverse launches lose a sixteenth of the chip as well.  No reservation for such plans: the chain workgroups take CUs as
    // the launches' tails free them (as in the 33..47 band of the rule).
    for (const PolyClass &pc : b->poly)
      if (pc.mix_kind == 1u && pc.nkb > XLP_NKB_4W) want = 0u;
    if (xl_exp_getenv("XL_EXP_NOMASK")) want = 0u;
    if (want > 0u && xl_exp_getenv("XL_EXP_ROUNDS1")) want = std::min(16u, (nwg_res + 7u) / 8u);  // (tuning: round 3's rule, one CU per chain workgroup)
    if (want > 0u && xl_exp_getenv("XL_EXP_RESERVE")) want = std::min(want, (uint32_t)atoi(xl_exp_getenv("XL_EXP_RESERVE")));  // (tuning: fewer CUs, more rounds)
    // (creating a masked stream pair takes ~25 ms: grow at once, shrink only when two CUs per XCD too many are held, so that
    // a client count hovering around a multiple of 512 does not recreate the streams at every join and leave)
    if (want > b->reserve_r || want + 2u <= b->reserve_r || (want == 0u && b->reserve_r != 0u)) {
      if (b->cs_masked) (void)hipStreamDestroy(b->cs_masked);
      if (b->nco_masked) (void)hipStreamDestroy(b->nco_masked);
      b->cs_masked = b->nco_masked = nullptr;
      b->last_nco = nullptr;
      b->reserve_r = 0;
      b->last_stream = b->own_stream;  // (everything was synchronised at the top of the plan)
      for (int i = 0; i < XL_NTAB; ++i) b->ev_done_valid[i] = false, b->tab_call[i] = 0;
      b->done_call = 0;
      if (want > 0u) {
        uint32_t chain_mask[8], main_mask[8];
        for (uint32_t wd = 0; wd < 8; ++wd) {
          chain_mask[wd] = 0u;
          for (uint32_t bit = 0; bit < 32; ++bit)
            if (wd * 32u + bit < 8u * want) chain_mask[wd] |= 1u << bit;
          main_mask[wd] = ~chain_mask[wd];
        }
        if (hipExtStreamCreateWithCUMask(&b->nco_masked, 8, chain_mask) == hipSuccess &&
            hipExtStreamCreateWithCUMask(&b->cs_masked, 8, main_mask) == hipSuccess) {
          b->reserve_r = want;
        } else {
          XL_LOG_ERR("CU-masked streams are not available (%s): the NCO chain kernel shares the chip", hipGetErrorString(hipGetLastError()));
          if (b->cs_masked) (void)hipStreamDestroy(b->cs_masked);
          if (b->nco_masked) (void)hipStreamDestroy(b->nco_masked);
          b->cs_masked = b->nco_masked = nullptr;
        }
      }
    }
  }

  lap("masked streams");
  // upload
  if (b->nco.empty()) {
    xl_plan_trim(b);
    b->dirty = false;
    return 0;
  }
  XL_TRY(xl_plan_alloc(b, (void **)&b->d_nco, b->nco.size() * sizeof(XlNcoClient)));
  XL_TRY(hipMemcpy(b->d_nco, b->nco.data(), b->nco.size() * sizeof(XlNcoClient), hipMemcpyHostToDevice));
  if (!b->poly.empty()) {
    if (xl_upload_launch_set(b, b->launches_rest, image_rest, &b->d_taps_rest, nullptr) != 0) goto fail;
  }
  lap("uploads (taps, groups, nco)");
  // ---- polyphase classes: images and the branch spectra of the new columns
  if (!b->poly.empty()) {
    if (b->d_W == nullptr) {
      std::vector<float> w(2 * 256);
      for (uint32_t n = 0; n < 256; ++n) {
        const double ang = -2.0 * M_PI * (double)n / 256.0;
        w[2 * n] = (float)cos(ang);
        w[2 * n + 1] = (float)sin(ang);
      }
      w[0] = 1.0f, w[1] = 0.0f;
      w[2 * 64] = 0.0f, w[2 * 64 + 1] = -1.0f;
      w[2 * 128] = -1.0f, w[2 * 128 + 1] = 0.0f;
      w[2 * 192] = 0.0f, w[2 * 192 + 1] = 1.0f;
      XL_TRY(hipMalloc((void **)&b->d_W, 256 * sizeof(float2)));
      XL_TRY(hipMemcpy(b->d_W, w.data(), 256 * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (b->phase_run_cap < b->phase_cap) {
      if (b->d_phase_run) (void)hipFree(b->d_phase_run);
      b->d_phase_run = nullptr;
      b->phase_run_cap = 0;
      XL_TRY(hipMalloc((void **)&b->d_phase_run, b->phase_cap * sizeof(float2)));
      b->phase_run_cap = b->phase_cap;
    }
    for (const Pending &pd : pending)
      if (xl_poly_sync_device(b, b->poly[pd.idx], pd.new_cols, pd.fresh, cap_samples) != 0) goto fail;
  }
  lap("polyphase images + R kernel");
  if (b->out_total > b->out_alloc) {
    const size_t want = b->out_total + b->out_total / 8 + 64;  // (headroom: the next joins do not reallocate the outputs)
    for (int i = 0; i < 2; ++i) {
      if (b->d_out[i]) (void)hipFree(b->d_out[i]);
      b->d_out[i] = nullptr;
    }
    for (int i = 0; i < XL_NTAB; ++i) {
      if (b->d_phtab[i]) (void)hipFree(b->d_phtab[i]);
      b->d_phtab[i] = nullptr;
    }
    b->out_alloc = 0;
    if (b->d_qphtab) (void)hipFree(b->d_qphtab);
    b->d_qphtab = nullptr;
    XL_TRY(hipMalloc((void **)&b->d_qphtab, (want / XL_PH_STRIDE + 8) * sizeof(short2)));
    for (int i = 0; i < 2; ++i) XL_TRY(hipMalloc((void **)&b->d_out[i], want * sizeof(float2)));
    for (int i = 0; i < XL_NTAB; ++i)
      XL_TRY(hipMalloc((void **)&b->d_phtab[i], (want / XL_PH_STRIDE + 8) * sizeof(float2)));  // every XL_PH_STRIDE-th phase
    b->out_alloc = want;
  }
  lap("output buffers");
  xl_plan_trim(b);
  lap("trim");
  b->dirty = false;
  return 0;
fail:
  // a half-built plan must not be used: drop it and stay dirty, the next call plans again (or fails again)
  xl_batch_free_plan(b, true);
  b->dirty = true;
  return xl_errno_of_last_hip_error();
}

// Tabulate the phases of a call as a launch of its own on stream `st`: committed phases -> table[tab] + next
// phases.  Only needed when no table was tabulated ahead (first call, changed call shape, changed client set).
static hipError_t xl_batch_nco(xlating_batch *b, const XlPos pos, int tab, hipStream_t st) {
  hipError_t e;
  hipEvent_t n0 = nullptr, n1 = nullptr;
  if (b->timing) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t ev;
      e = xl_batch_timing_event(b, &ev);
      if (e != hipSuccess) return e;
      b->ev_ncot.push_back(ev);
    }
    n0 = b->ev_ncot[b->ev_ncot.size() - 2];
    n1 = b->ev_ncot[b->ev_ncot.size() - 1];
    e = hipEventRecord(n0, st);
    if (e != hipSuccess) return e;
  }
  e = xl_launch_nco_table(b->d_nco, (uint32_t)b->nco.size(), b->d_phase[b->pcur], b->d_phase[xl_nx(b->pcur)],
                          b->d_phtab[tab], pos, 0xFFFFFFFFu, b->nco_prio, st);
  if (e != hipSuccess) return e;
  return n1 ? hipEventRecord(n1, st) : hipSuccess;
}

#ifdef XL_TUNING
static hipError_t xl_dump_trace(const char *path, const unsigned long long *d, size_t n, hipStream_t s) {
  std::vector<unsigned long long> h(n);
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  e = hipMemcpy(h.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  if (FILE *f = fopen(path, "wb")) {
    fwrite(h.data(), sizeof(unsigned long long), n, f);
    fclose(f);
  }
  return hipSuccess;
}
#endif

// One call: G blocks of S samples each, contiguous at d_blocks.
#define XL_STREAM_ENGINE_P (reinterpret_cast<hipStream_t>((intptr_t)-1))

// One call: G blocks of S samples each, contiguous at d_blocks.  s_in: the caller's stream, or XL_STREAM_ENGINE_P = the
// engine's own compute stream (the CU-masked one for calls whose NCO chain runs on the side stream).  wait_ev / record_ev:
// optional events of the caller, waited for before / recorded after the call's work on that stream.
static int xl_batch_run(xlating_batch *b, const void *d_blocks, size_t input_len, unsigned G, int mode, hipStream_t s_in,
                        hipEvent_t wait_ev = nullptr, hipEvent_t record_ev = nullptr) {
  const size_t S = input_len / 2;
  hipStream_t s = s_in;
  if (S > b->max_samples || G < 1 || G > b->gcap ||
      (mode != XL_MODE_NATIVE && mode != XL_MODE_OPTIMIZED && mode != XL_MODE_Q15 && mode != XL_MODE_OPTIMIZED_X86 &&
       mode != XL_MODE_OPTIMIZED_X86_FMA) ||
      (mode == XL_MODE_Q15 && b->fmt == XL_FMT_CF32))
    return -EINVAL;
  // the x86 AVX build's optimized variant = the optimized arithmetic with the phase never renormalised (xlating.c:338-339):
  // the stream position carries the flag to every kernel that walks phases (xl_grid.h: XL_POS_NORENORM)
  const uint32_t pos_flags = mode == XL_MODE_OPTIMIZED_X86 ? XL_POS_NORENORM
                             : (mode == XL_MODE_OPTIMIZED_X86_FMA ? (XL_POS_NORENORM | XL_POS_FMA_STEP) : 0u);
  if (mode == XL_MODE_OPTIMIZED_X86 || mode == XL_MODE_OPTIMIZED_X86_FMA) mode = XL_MODE_OPTIMIZED;
  if (mode == XL_MODE_Q15 && !b->want_q15) {  // (the Q15 tap image is built from the first Q15 call on)
    b->want_q15 = true;
    b->all_built = false;
  }
  if (b->poisoned) return -EIO;
  // a client that was still inside its zero-history when the plan was built may be mature by now: it then joins
  // the class of its grid (direct kernel) / its (D, T) class (polyphase) -- re-plan
  if (!b->dirty && b->planned_immature > 0)
    for (const Client &c : b->clients)
      if (c.alive && !c.planned_mature && xl_mature(c)) {
        b->dirty = true;
        break;
      }
  if (!b->dirty && (uint64_t)b->trel + (uint64_t)S * G >= (1ull << 31)) b->dirty = true;  // re-base the classes' stream records
  if (b->dirty) {
    int rc = xl_batch_plan(b);
    if (rc != 0) return rc;
  }
  if (G > 1 && S < b->plan_maxD && !b->nco.empty()) {
    XL_LOG_ERR("a call of %u blocks needs blocks of at least the largest decimation (%u samples); got %zu", G, b->plan_maxD, S);
    return -EINVAL;
  }
  XlPos pos;
  pos.trel = b->trel;
  pos.S = (uint32_t)S;
  pos.G = G;
  pos.pad = pos_flags;
  // optimized mode: the polyphase classes leave the direct launches (tiny calls stay direct: a segment is 128 or 256
  // branch samples whatever the call holds)
  uint32_t maxK = 0;  // the most outputs any client produces in this call
  for (const DirectClass &cs : b->classes) maxK = std::max(maxK, xl_grid_dyn(cs.D, cs.T, cs.rem0, cs.hv0, pos).K);
  const bool use_poly = mode == XL_MODE_OPTIMIZED && !b->poly.empty() && maxK >= 2 * XLP_M_MAX;
  if (!use_poly && !b->nco.empty()) {  // this call runs the all-clients launch set: build it if the plan has not yet
    int rc = xl_batch_build_all_set(b);
    if (rc != 0) return rc;
  }
  const bool light = xl_direct_is_light((use_poly ? b->macs_rest : b->macs_all) * (double)S);
  // One-block calls (the reference's call granularity) on the polyphase path: the three launches are short since the mix runs on
  // the matrix cores, and a slice of the recurrence inside each made every one of them last as long as its slice (1024 clients:
  // 50.2 us per block; the chain kernel beside them, four calls per launch: 47.7; 128 clients: 40.4 -> 29.2; 4096: 129 -> 134,
  // hence the limit)
  const bool one_block_side = G == 1 && use_poly && b->nco.size() <= XL_SIDE_ONE_BLOCK_MAX;
  const bool side_call = mode != XL_MODE_Q15 && (b->nco_side > 0 || (b->nco_side < 0 && (G >= 2 || one_block_side) && (use_poly || (light && b->cs_masked && s == XL_STREAM_ENGINE_P))));  // (a caller's own, unmasked stream would keep filling the chain's CUs)
  if (s == XL_STREAM_ENGINE_P) s = (side_call && b->cs_masked) ? b->cs_masked : b->own_stream;
  // Calls depend on each other through the engine's device state (history, phases, tables): a call on another stream than the
  // previous one is ordered behind it.  (Overlapping consecutive one-block calls through two compute streams -- round 4's
  // "pipeline_calls" -- cost this runtime more in cross-stream events than it gained: profiles/r04_one_block_pipelining.txt.)
  if (s != b->last_stream) {
    XL_TRY(hipEventRecord(b->dep_ev, b->last_stream));
    XL_TRY(hipStreamWaitEvent(s, b->dep_ev, 0));
  }
  b->last_stream = s;
  b->fetched = false;
  if (wait_ev) XL_TRY(hipStreamWaitEvent(s, wait_ev, 0));
  if (b->nco.empty()) {
    if (record_ev) XL_TRY(hipEventRecord(record_ev, s));
    return 0;
  }

  if (mode == XL_MODE_Q15) {
    // ---- the Q15 family (xlating.c:92-140, 416-447): its own phase (never renormalised), the same raw history.  The
    // float phases stay where they are; a float phase table tabulated ahead was for a call at this stream position and
    // is dropped.
    const int p = (int)(b->ncalls & 1);
    const int hb = b->hcur, hn = b->hcur ^ 1;
    const uint32_t N = (uint32_t)(S * G);
    if (b->spec_n > 0 && b->spec_on_side) XL_TRY(hipStreamWaitEvent(s, b->ev_chain[b->spec_ev], 0));
    b->spec_n = 0;
    b->poisoned = true;
    XL_TRY(xl_launch_nco_q15_batch(b->d_nco, b->d_qinc, (uint32_t)b->nco.size(), b->d_qphase, b->d_qphtab, pos, s));
    bool rolled = false;
    for (int lq = 0; lq < XL_NLAUNCH && maxK > 0; ++lq) {
      Launch &L = b->launches[lq];
      if (L.groups.empty()) continue;
      XlFirArgs a;
      memset(&a, 0, sizeof(a));
      a.in0 = b->d_hist[hb];
      a.n0 = XL_HCAP;
      a.in1 = d_blocks;
      a.n1 = N;
      a.fmt = b->fmt;
      a.pos = pos;
      a.groups = L.d_groups;
      a.ngroups = (uint32_t)L.groups.size();
      a.ota = L.ota;
      a.xtiles = (std::min((N + L.minD - 1) / L.minD, maxK) + L.ota - 1) / L.ota;
      a.out = b->d_out[p];
      if (!rolled) {
        a.hist_out = b->d_hist[hn];
        a.hist_units = XL_HCAP * (b->bps / 2);
        a.block_units = N * (b->bps / 2);
        rolled = true;
      }
      XL_TRY(xl_launch_fir_q15_batch(L.ct, L.nw, a, b->d_qtaps, b->d_qphtab, L.lds, s));
    }
    if (!rolled) XL_TRY(xl_launch_update_history(b->d_hist[hb], d_blocks, XL_HCAP, N, b->bps, b->d_hist[hn], s));
    if (record_ev) XL_TRY(hipEventRecord(record_ev, s));
    b->poisoned = false;
    for (Client &c : b->clients) {
      if (!c.alive) continue;
      const uint32_t j0 = (uint32_t)((c.D - c.consumed % c.D) % c.D);
      c.last_Kg.resize(G);
      uint32_t prev = 0;
      for (uint32_t g = 1; g <= G; ++g) {
        const uint32_t ms = xl_grid_mstart(j0, c.D, (uint32_t)S, g);
        c.last_Kg[g - 1] = ms - prev;
        prev = ms;
      }
      c.last_K = prev;
      c.consumed += N;
    }
    b->trel += N;
    b->ocur = p;
    b->hcur = hn;
    b->ncalls++;
    b->calls_since_plan++;
    b->last_q15 = true;
    return 0;
  }
  b->last_q15 = false;

  {
    const int p = (int)(b->ncalls & 1);  // parity of this call: output buffer
    const int hb = b->hcur, hn = b->hcur ^ 1;
    const uint32_t N = (uint32_t)(S * G);
    // Who tabulates the NEXT call's phases: the NCO role inside this call's launches (fuse), or xl_nco_chain_kernel on
    // the side stream, concurrently with them (side).  The side stream pays for calls of several blocks on the
    // polyphase path, whose three launches are short against the chain (measured, 8 blocks per call: 42.7 -> 36.8 us per
    // block at 1024 clients, 30.7 -> 28.8 at 128); a direct FIR launch hides the chain in its spare waves for free and
    // would only lose the chain kernel's CUs (1024 clients, native: 203 -> 212 us per block), and with one block per
    // call the cross-stream events cost more than the overlap gains (51.3 -> 58.6).
    const bool side = side_call;
    bool nco_fused = false;
    bool chain_wait = false;  // this call's table comes from the side stream: wait for it before the first reader

    // ---- this call's phase table: tabulated ahead by the previous call's launches if the shape guess was right
    const int tab = xl_nx(b->tab);
    int pcur = b->pcur;
    int spec_left = 0;  // look-ahead calls that stay valid behind this one (a chain launch covers up to two)
    const int chain_ev = b->spec_ev;
    bool tab_from_s = true;
    if (b->spec_n > 0 && b->spec_S == S && b->spec_G == G && b->spec_flags == pos_flags) {
      tab_from_s = !b->spec_on_side;
      // (the second call of a chain launch's pair: this stream has already waited for that launch)
      chain_wait = b->spec_on_side && !(b->waited_valid && b->waited_ev == chain_ev && b->waited_stream == s);
      spec_left = b->spec_n - 1;
    } else {
      // (a look-ahead of the wrong shape may still be running on the side stream, on these very buffers)
      if (b->spec_n > 0 && b->spec_on_side) XL_TRY(hipStreamWaitEvent(s, b->ev_chain[chain_ev], 0));
      if (b->spec_n > 0) b->chain_ahead = 1;  // (a wrong guess: look less far ahead until the guesses hold again)
      if (b->ev_done_valid[tab]) XL_TRY(hipStreamWaitEvent(s, b->ev_done[tab], 0));  // (same stream normally: a no-op)
      XL_TRY(xl_batch_nco(b, pos, tab, s));
    }
    pcur = xl_nx(pcur);  // the phases written by that tabulation are now the committed ones
    b->spec_n = 0;       // (from here on a failure poisons the engine)
    b->poisoned = true;
    // (with a look-ahead table still in hand the launches carry no NCO role: the call after this one has its table)
#ifdef XL_TUNING
    const bool fuse = !side && spec_left == 0 && !b->exp_nofuse;  // tuning: tabulate by a launch of its own before every call
#else
    const bool fuse = !side && spec_left == 0;
#endif
    int launched_n = 0;

    // ---- side stream: the NEXT call's table (same shape assumed) into table[tab ^ 1], concurrently with the launches
    // below.  That table was last read by the previous call's launches (ev_done), the committed phases d_phase[pcur]
    // were written by the tabulation of THIS call's table (earlier on the same side stream, or on `s`: ordered below).
    if (side && spec_left == 0) {
      // (the CU-masked pair of streams goes together: a chain kernel confined to CUs that the caller's own, unmasked
      // stream keeps filling would wait for kernel boundaries)
      hipStream_t ns = (s == b->cs_masked && b->nco_masked) ? b->nco_masked : b->nco_stream;
      if (ns != b->last_nco) {  // consecutive chains depend on each other through the phase buffers
        if (b->last_nco) {
          XL_TRY(hipEventRecord(b->dep_ev, b->last_nco));
          XL_TRY(hipStreamWaitEvent(ns, b->dep_ev, 0));
        }
        b->last_nco = ns;
      }
      if (tab_from_s) {  // this call's table was tabulated on `s` (just now, or by the previous call's launches): order the side stream behind it
        XL_TRY(hipEventRecord(b->dep_ev, s));
        XL_TRY(hipStreamWaitEvent(ns, b->dep_ev, 0));
      }
      XlChainCalls cc;
      memset(&cc, 0, sizeof(cc));
      cc.n = (uint32_t)std::min<int>(std::max(std::min(b->chain_calls, b->chain_ahead), 1), (int)XL_CHAIN_MAXCALLS);
      b->chain_ahead = std::min(2 * b->chain_ahead, (int)XL_CHAIN_MAXCALLS);  // (this launch is made because the previous one's calls were all used)
      int tt[XL_CHAIN_MAXCALLS];
      for (int i = 0, t = tab, pp = pcur; i < (int)cc.n; ++i) {
        t = xl_nx(t), pp = xl_nx(pp);
        cc.tab[i] = b->d_phtab[t];
        cc.state_out[i] = b->d_phase[pp];
        tt[i] = t;
      }
      // the tables' last readers (XL_NTAB - 1 - i calls back): one wait for the latest of them covers the earlier ones that
      // were recorded on the same stream (every wait is a queue packet of its own, ~4 us in front of the chain launch)
      {
        uint64_t need = 0;  // the latest call that read one of these tables
        for (int i = 0; i < (int)cc.n; ++i) need = std::max(need, b->tab_call[tt[i]]);
        if (need != 0) {
          if (b->done_call >= need) {
            XL_TRY(hipStreamWaitEvent(ns, b->ev_done[b->done_tab], 0));
          } else if (!tab_from_s) {  // (tab_from_s: the side stream was just ordered behind everything on `s`)
            XL_TRY(hipEventRecord(b->dep_ev, s));  // all earlier calls' launches precede this point of `s`
            XL_TRY(hipStreamWaitEvent(ns, b->dep_ev, 0));
          }
        }
      }
      XL_TRY(xl_launch_nco_chain(b->d_nco, (uint32_t)b->nco.size(), b->d_phase[pcur], cc, xl_grid_next(pos),
                                 b->d_chain_stats, ns, b->ev_chain[xl_nx(tab)]));
      launched_n = (int)cc.n;
      b->waited_valid = false;  // (ev_chain[xl_nx(tab)] now stands for this launch)
    }

    // ---- the launches on the caller's stream: window images from [d_hist[hb] | blocks], phases from table[tab] ->
    // d_out[p]; the first launch also rolls the raw history into d_hist[hn], and the launches tabulate table[tab ^ 1]
    // for the next call
    hipEvent_t f0 = nullptr, f1 = nullptr;
    if (b->timing && maxK > 0 && b->ncalls % b->timing_every == 0) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t ev;
        XL_TRY(xl_batch_timing_event(b, &ev));
        b->ev.push_back(ev);
      }
      f0 = b->ev[b->ev.size() - 2];
      f1 = b->ev[b->ev.size() - 1];
    }
    bool rolled = false;
    bool done_attached = false;  // ev_done[tab] rides on the call's last launch
    // does this call record ev_done[tab]?  Side-stream calls: only the ones that launch a chain (see tab_call); non-side calls
    // after a side stream was used: always
    const bool want_done = (side || b->last_nco != nullptr) && (!side || launched_n > 0);
    bool record_attached = false;  // the caller's record_ev rides on the last launch instead (no ev_done wanted there)
    Launch *const Ls = use_poly ? b->launches_rest : b->launches;
    if (maxK > 0) {
      if (f0) XL_TRY(hipEventRecord(f0, s));
      for (int lq = 0; lq < XL_NLAUNCH; ++lq) {
        Launch &L = Ls[lq];
        if (L.groups.empty()) continue;
        XlFirArgs a;
        memset(&a, 0, sizeof(a));
        a.in0 = b->d_hist[hb];
        a.n0 = XL_HCAP;
        a.in1 = d_blocks;
        a.n1 = N;
        a.fmt = b->fmt;
        a.pos = pos;
        a.groups = L.d_groups;
        a.ngroups = (uint32_t)L.groups.size();
        a.ota = L.ota;
        const uint32_t Kl = (N + L.minD - 1) / L.minD;  // (an upper bound of the launch's largest output count)
        a.xtiles = (std::min(Kl, maxK) + L.ota - 1) / L.ota;
        // wave-priority segments pay when the launch is about one round of workgroups, and cost when new
        // workgroups keep arriving (they would outrank nearly finished ones): enable up to two rounds
        const size_t wgs = (size_t)a.ngroups * a.xtiles;
        const size_t cap = 256 * std::max<size_t>(1, std::min<size_t>((160 * 1024) / std::max<size_t>(L.lds, 1), 7));
        const bool flat = (b->exp_flags & 2u) || wgs > 2 * cap;
        a.flags = (L.all_wide ? 1u : 0u) | (flat ? 2u : 0u) | 4u | ((b->nco_prio & 3u) << 4);
        a.taps = use_poly ? b->d_taps_rest : b->d_taps;
        a.phtab = b->d_phtab[tab];
        a.out = b->d_out[p];
        if (!rolled) {
          a.hist_out = b->d_hist[hn];
          a.hist_units = XL_HCAP * (b->bps / 2);
          a.block_units = N * (b->bps / 2);
          rolled = true;
          if (fuse) {
            a.nco_clients = b->d_nco;
            a.nco_nclients = (uint32_t)b->nco.size();
            const uint32_t slots = L.idle_waves * a.xtiles;
            const uint32_t want = (a.nco_nclients + XL_NCO_LANES - 1) / XL_NCO_LANES;
            if (b->riders && slots > 0 && (uint64_t)slots * 64u >= a.nco_nclients &&
                xl_riders_window(wgs, L.nw, L.groups[0].Tpad, L.ct, maxK, L.lds, b->riders_min_wgs)) {
              a.nco_slots = std::min(slots, want);
              a.nco_lanes = (a.nco_nclients + a.nco_slots - 1) / a.nco_slots;
            } else {
              a.nco_wpw = std::max<uint32_t>(1, std::min<uint32_t>(b->nco_wpw, (uint32_t)L.nw));
              a.nco_blocks = (a.nco_nclients + XL_NCO_LANES * a.nco_wpw - 1) / (XL_NCO_LANES * a.nco_wpw);
            }
            a.nco_state_in = b->d_phase[pcur];
            a.nco_state_out = b->d_phase[xl_nx(pcur)];
            a.nco_tab = b->d_phtab[xl_nx(tab)];
            nco_fused = true;
          }
        }
#ifdef XL_TUNING
        size_t trace_n = 0;
        if (b->exp_trace) {
          trace_n = ((size_t)a.nco_blocks + (size_t)8 * ((a.ngroups * a.xtiles + 7) / 8)) * XL_NW_MAX * 6;
          if (trace_n > b->trace_cap) {
            if (b->d_trace) (void)hipFree(b->d_trace);
            b->d_trace = nullptr;
            XL_TRY(hipMalloc((void **)&b->d_trace, trace_n * sizeof(unsigned long long)));
            b->trace_cap = trace_n;
          }
          XL_TRY(hipMemsetAsync(b->d_trace, 0, trace_n * sizeof(unsigned long long), s));
          a.trace = b->d_trace;
        }
#endif
        if (chain_wait) {
          XL_TRY(hipStreamWaitEvent(s, b->ev_chain[chain_ev], 0));
          chain_wait = false;
          b->waited_valid = true, b->waited_ev = chain_ev, b->waited_stream = s;
        }
        XL_TRY(xl_launch_fir(L.ct, mode, L.nw, a, L.lds, s));
#ifdef XL_TUNING
        if (b->exp_trace) XL_TRY(xl_dump_trace(b->exp_trace, b->d_trace, trace_n, s));
#endif
      }
      if (use_poly) {
        for (PolyClass &pc : b->poly) {
          XlpArgs pa;
          memset(&pa, 0, sizeof(pa));
          if (!rolled) {  // the forward launch also rolls the raw history
            pa.hist_out = b->d_hist[hn];
            pa.hist_units = XL_HCAP * (b->bps / 2);
            pa.block_units = N * (b->bps / 2);
            pa.roll_blocks = 32;
            rolled = true;
          }
          pa.in0 = b->d_hist[hb];
          pa.n0 = XL_HCAP;
          pa.in1 = d_blocks;
          pa.n1 = N;
          pa.fmt = (uint32_t)b->fmt;
          pa.pos = pos;
          // the class's shared grid in this call (xl_grid.h): one D-step ahead of the reference client's output 0
          const XlDyn dref = xl_grid_dyn(pc.D, pc.T, pc.rem_ref0, pc.hv0, pos);
          pa.j0_ref = dref.j0;
          pa.base = dref.base - pc.D;
          pa.Kq = xl_merge_points(pc.D, pos);
          pa.zero_below = dref.zero_below;  // (0 for mature members: nothing below their join points is weighted)
          pa.D = pc.D;
          pa.Dpad = pc.Dpad;
          pa.T = pc.T;
          pa.A = pc.A;
          pa.V = pc.V;
          pa.M = pc.M;
          pa.nseg = (pa.Kq + pc.V - 1) / pc.V;
          pa.nseg_cap = pc.nseg_cap;
          if (pa.nseg > pa.nseg_cap) {
            XL_LOG_ERR("internal: %u segments exceed the plan's capacity %u", pa.nseg, pa.nseg_cap);
            goto fail;
          }
          pa.ncg = pc.ncg;
          pa.exp = b->poly_exp;
          pa.inv_reg = b->inv_reg;
          pa.mix_kind = pc.mix_kind;
          pa.nkb = pc.nkb;
          pa.mix_pp = b->mix_pp;
          pa.Rh = pc.d_Rh;
          pa.cscale = pc.d_cscale;
          pa.segmax = pc.d_segmax;
          pa.seg_par = pc.seg_par;
          pa.seg_cap = pc.seg_cap;
          pc.seg_par ^= 1u;  // (a failed call leaves stale maxima behind at worst: a smaller scale than necessary, never a wrong one)
          pa.W = b->d_W;
          pa.X = pc.d_X;
          pa.Y = pc.d_Y;
          pa.cols = pc.d_cols;
          pa.phtab = b->d_phtab[tab];
          pa.out = b->d_out[p];
          // the NEXT call's phase recurrence rides in these launches (a direct launch above carries all of it if there is
          // one): TWO slices, forward | inverse -- never the mix launch: no launch that issues matrix instructions hosts the role
          // 
          const bool carry = fuse && !nco_fused;
          const uint32_t sl1 = std::min(b->poly_slice_fi, 60000u);
          if (carry) {
            pa.nco_clients = b->d_nco;
            pa.nco_nclients = (uint32_t)b->nco.size();
            pa.nco_blocks = (pa.nco_nclients + XL_NCO_LANES - 1) / XL_NCO_LANES;
            pa.nco_tab = b->d_phtab[xl_nx(tab)];
            pa.nco_prio = b->nco_prio;
            pa.nco_k0 = 0;
            pa.nco_k1 = sl1;
            pa.nco_state_src = b->d_phase[pcur];
            pa.nco_state_dst = b->d_phase_run;
          }
          hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
          if (b->timing == 2) {
            for (int i = 0; i < 4; ++i) {
              XL_TRY(xl_batch_timing_event(b, &pe[i]));
              b->ev_poly.push_back(pe[i]);
            }
            XL_TRY(hipEventRecord(pe[0], s));
          }
          // the call's last launch carries the "table has been read" event the side stream waits for
          const bool last_launch = side && rolled && &pc == &b->poly.back();
#ifdef XL_TUNING
          const bool trace_fwd = b->poly_trace && xl_exp_getenv("XL_EXP_POLY_TRACE_FWD");  // (the forward launch instead)
          if (trace_fwd) {
            if (!b->d_ptrace) XL_TRY(hipMalloc((void **)&b->d_ptrace, 32768 * sizeof(unsigned long long)));
            XL_TRY(hipMemsetAsync(b->d_ptrace, 0, 32768 * sizeof(unsigned long long), s));
            pa.trace = b->d_ptrace;
          }
#endif
          XL_TRY(xlp_launch_forward(pa, s));
#ifdef XL_TUNING
          if (trace_fwd) {
            pa.trace = nullptr;
            XL_TRY(xl_dump_trace(b->poly_trace, b->d_ptrace, 32768, s));
          }
#endif
          if (pe[1]) XL_TRY(hipEventRecord(pe[1], s));
          pa.roll_blocks = 0;
          pa.nco_blocks = 0;  // (no role in the mix launch)
#ifdef XL_TUNING
          const bool trace_inv = b->poly_trace && !trace_fwd && xl_exp_getenv("XL_EXP_POLY_TRACE_INV");  // (the inverse launch instead)
          if (b->poly_trace && !trace_inv && !trace_fwd) {  // timeline of the mix launch (work waves' span + each NCO wave)
            if (!b->d_ptrace) XL_TRY(hipMalloc((void **)&b->d_ptrace, 32768 * sizeof(unsigned long long)));
            XL_TRY(hipMemsetAsync(b->d_ptrace, 0, 32768 * sizeof(unsigned long long), s));
            pa.trace = b->d_ptrace;
          }
#endif
          XL_TRY(xlp_launch_mix(pa, s));
          pa.nco_skip = 0;
#ifdef XL_TUNING
          if (b->poly_trace && !trace_inv && !trace_fwd) {
            pa.trace = nullptr;
            XL_TRY(xl_dump_trace(b->poly_trace, b->d_ptrace, 32768, s));
          }
#endif
          if (pe[2]) XL_TRY(hipEventRecord(pe[2], s));
          if (carry) {
            pa.nco_tab = b->d_phtab[xl_nx(tab)];
            pa.nco_blocks = (pa.nco_nclients + XL_NCO_LANES - 1) / XL_NCO_LANES;
            pa.nco_k0 = sl1;
            pa.nco_k1 = 65536;
            pa.nco_state_src = b->d_phase_run;
            if (b->inv_skip_at > 0) {
              pa.nco_skip_at = b->inv_skip_at;
              pa.nco_skip = pa.nco_blocks;
            }
            pa.nco_state_dst = b->d_phase[xl_nx(pcur)];
            nco_fused = true;
          }
#ifdef XL_TUNING
          if (trace_inv) {
            if (!b->d_ptrace) XL_TRY(hipMalloc((void **)&b->d_ptrace, 32768 * sizeof(unsigned long long)));
            XL_TRY(hipMemsetAsync(b->d_ptrace, 0, 32768 * sizeof(unsigned long long), s));
            pa.trace = b->d_ptrace;
          }
#endif
          if (chain_wait) {  // (the forward and mix launches do not read the table)
            XL_TRY(hipStreamWaitEvent(s, b->ev_chain[chain_ev], 0));
            chain_wait = false;
            b->waited_valid = true, b->waited_ev = chain_ev, b->waited_stream = s;
          }
          {
            const bool attach = last_launch
#ifdef XL_TUNING
                                && !trace_inv
#endif
                ;
            XL_TRY(xlp_launch_inverse(pa, s, attach ? (want_done ? b->ev_done[tab] : record_ev) : nullptr));
            pc.last_inv = (int)xlp_inverse_pick(pa.M, pa.inv_reg, pa.nseg * pa.ncg * 4u);
            done_attached = attach && want_done;
            record_attached = attach && !want_done && record_ev != nullptr;
          }
#ifdef XL_TUNING
          if (trace_inv) {
            pa.trace = nullptr;
            XL_TRY(xl_dump_trace(b->poly_trace, b->d_ptrace, 32768, s));
          }
#endif
          if (pe[3]) XL_TRY(hipEventRecord(pe[3], s));
        }
      }
      if (f1) XL_TRY(hipEventRecord(f1, s));
    }
    if (!rolled)  // no client produced output in this call (tiny block): roll the history on its own
      XL_TRY(xl_launch_update_history(b->d_hist[hb], d_blocks, XL_HCAP, N, b->bps, b->d_hist[hn], s));
    // table[tab] has been read by everything enqueued so far (only the side stream ever waits for this)
    // -- and a LATER side-stream chain launch may find this slot in its ring even when this call ran no side stream
    // (alternating modes, alternating caller streams, nco_calls_per_launch < 4): once a side stream has been used the
    // event is recorded for every call.  Engines that never use the side stream never pay for it.
    if (side || b->last_nco != nullptr) b->tab_call[tab] = b->ncalls + 1;
    if (want_done) {
      if (!done_attached) XL_TRY(hipEventRecord(b->ev_done[tab], s));
      b->ev_done_valid[tab] = true;
      b->ev_done_stream[tab] = s;
      b->done_call = b->ncalls + 1;
      b->done_tab = tab;
    } else {
      b->ev_done_valid[tab] = false;
    }

    if (record_ev && !record_attached) XL_TRY(hipEventRecord(record_ev, s));
    // ---- everything is enqueued: commit the host-side state of the call
    b->poisoned = false;
    for (Client &c : b->clients) {
      if (!c.alive) continue;
      const uint32_t j0 = (uint32_t)((c.D - c.consumed % c.D) % c.D);
      c.last_Kg.resize(G);
      uint32_t prev = 0;
      for (uint32_t g = 1; g <= G; ++g) {
        const uint32_t ms = xl_grid_mstart(j0, c.D, (uint32_t)S, g);
        c.last_Kg[g - 1] = ms - prev;
        prev = ms;
      }
      c.last_K = prev;
      c.consumed += N;
    }
    b->trel += N;
    b->pcur = pcur;
    b->tab = tab;
    b->ocur = p;
    b->hcur = hn;
    b->ncalls++;
    b->calls_since_plan++;
    // ---- the NEXT call's phases, guessing it has the same shape, were tabulated inside the launches above;
    // without them (tiny call, or the tuning switch) the next call tabulates for itself
    if (launched_n > 0) {
      b->spec_n = launched_n;
      b->spec_on_side = true;
      b->spec_ev = xl_nx(tab);
    } else if (spec_left > 0) {
      b->spec_n = spec_left;  // (same launch, same event)
    } else if (nco_fused) {
      b->spec_n = 1;
      b->spec_on_side = false;
    }
    b->spec_S = (uint32_t)S;
    b->spec_G = G;
    b->spec_flags = pos_flags;
  }
  return 0;
fail:
  return b->poisoned ? -EIO : xl_errno_of_last_hip_error();
}

extern "C" int xlating_batch_describe(xlating_batch *b, char *buf, size_t n) {
  if (b == nullptr || buf == nullptr || n == 0) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  // A re-plan reassigns every client's output row and may reallocate the output buffers, while the latest call's outputs
  // stay valid "until the next process call" (xlating_batch.h): with calls behind it a dirty engine reports the plan
  // those calls ran with and says that a new one is pending; before the first call after a plan nothing can be lost.
  const bool pending = b->dirty && b->calls_since_plan > 0;
  if (b->dirty && !pending) {
    int rc = xl_batch_plan(b);
    if (rc != 0) return rc;
  }
  std::string d = "clients " + std::to_string(b->nalive) + " classes " + std::to_string(b->classes.size()) + " | direct:";
  if (!b->all_built && !b->nco.empty()) d += " (built by the first call that runs all clients on the direct kernel)";
  for (const Launch &L : b->launches)
    if (!L.groups.empty()) d += " h" + std::to_string(L.ct) + " x " + std::to_string(L.groups.size()) + " groups";
  d += " | polyphase:";
  if (b->poly.empty()) d += " none";
  for (size_t k = 0; k < b->poly.size(); ++k) {
    const PolyClass &pc = b->poly[k];
    d += " cls" + std::to_string(k) + " D" + std::to_string(pc.D) + " T" + std::to_string(pc.T) + " cols" +
         std::to_string(pc.members.size()) + " V" + std::to_string(pc.V) + " M" + std::to_string(pc.M);
    if (pc.dmax) d += " offsets<=" + std::to_string(pc.dmax);
    d += pc.mix_kind == 3u ? " mix=mf32" : " mix=mfma";
    if (pc.M == 128u) {
      // (before the class's first launch: what the size rule picks for a call of the plan's full size -- gcap blocks of max_samples)
      const uint32_t kq_full = (uint32_t)(((uint64_t)b->max_samples * b->gcap + pc.D - 1u) / pc.D) + 1u;
      const int k = b->inv_reg ? (int)b->inv_reg
                               : (pc.last_inv >= 0 ? pc.last_inv : (int)xlp_inverse_pick(pc.M, 0u, ((kq_full + pc.V - 1u) / pc.V) * pc.ncg * 4u));
      d += k == 6 ? " inv=cut32" : (k == 5 ? " inv=lanes8" : (k == 3 ? " inv=lds" : " inv=auto"));
    }
  }
  if (!b->poly.empty()) {
    d += " | optimized-mode direct:";
    bool any = false;
    for (const Launch &L : b->launches_rest)
      if (!L.groups.empty()) {
        d += " h" + std::to_string(L.ct) + " x " + std::to_string(L.groups.size()) + " groups";
        any = true;
      }
    if (!any) d += " none";
  }
  if (b->reserve_r) d += " | side kernel: " + std::to_string(8u * b->reserve_r) + " CUs reserved";
  if (pending) d += " | re-plan pending (client set or options changed: the next process call plans again)";
  const size_t len = std::min(d.size(), n - 1);
  memcpy(buf, d.data(), len);
  buf[len] = 0;
  return (int)len;
}

extern "C" int xlating_batch_process_device_group_ev(xlating_batch *b, const void *d_input, size_t input_len,
                                                     unsigned nblocks, int mode, void *hip_stream, void *wait_event,
                                                     void *record_event) {
  if (b == nullptr || (d_input == nullptr && input_len > 0)) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  // NULL is HIP's legacy default stream (what torch.cuda.current_stream().cuda_stream is by default): pass it through
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  return xl_batch_run(b, d_input, input_len, nblocks, mode, s, reinterpret_cast<hipEvent_t>(wait_event),
                      reinterpret_cast<hipEvent_t>(record_event));
}

extern "C" int xlating_batch_process_device_group(xlating_batch *b, const void *d_input, size_t input_len,
                                                  unsigned nblocks, int mode, void *hip_stream) {
  return xlating_batch_process_device_group_ev(b, d_input, input_len, nblocks, mode, hip_stream, nullptr, nullptr);
}

extern "C" int xlating_batch_process_device(xlating_batch *b, const void *d_input, size_t input_len, int mode,
                                            void *hip_stream) {
  return xlating_batch_process_device_group(b, d_input, input_len, 1, mode, hip_stream);
}

extern "C" int xlating_batch_process_host_group(xlating_batch *b, const void *input, size_t input_len, unsigned nblocks,
                                                int mode) {
  if (b == nullptr || (input == nullptr && input_len > 0)) return -EINVAL;
  const size_t S = input_len / 2;
  if (S > b->max_samples || nblocks < 1 || nblocks > b->gcap) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  hipStream_t s = b->own_stream;
  // the pinned staging buffer and the device block buffer are reused: wait until the previous call is consumed
  xl_batch_sync_all(b);
  const size_t bytes = S * nblocks * b->bps;
  if (bytes) {
    memcpy(b->h_block, input, bytes);
    if (hipMemcpyAsync(b->d_block, b->h_block, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return -EIO;
  }
  return xl_batch_run(b, b->d_block, input_len, nblocks, mode, s);
}

extern "C" int xlating_batch_process_host(xlating_batch *b, const void *input, size_t input_len, int mode) {
  return xlating_batch_process_host_group(b, input, input_len, 1, mode);
}

extern "C" size_t xlating_batch_output_len(const xlating_batch *b, int id) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive) return 0;
  return b->clients[id].last_K;
}

extern "C" size_t xlating_batch_output_len_block(const xlating_batch *b, int id, unsigned block) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive ||
      block >= b->clients[id].last_Kg.size())
    return 0;
  return b->clients[id].last_Kg[block];
}

extern "C" int xlating_batch_sync(xlating_batch *b) {
  if (b == nullptr) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  return hipStreamSynchronize(b->last_stream) == hipSuccess ? 0 : -EIO;
}

extern "C" int xlating_batch_query(xlating_batch *b) {
  if (b == nullptr) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  const hipError_t e = hipStreamQuery(b->last_stream);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();  // (not an error: clear the sticky code)
    return 0;
  }
  return e == hipSuccess ? 1 : -EIO;
}

extern "C" int xlating_batch_record_event(xlating_batch *b, void *hip_event) {
  if (b == nullptr || hip_event == nullptr) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  hipEvent_t ev = reinterpret_cast<hipEvent_t>(hip_event);
  return hipEventRecord(ev, b->last_stream) == hipSuccess ? 0 : -EIO;
}

extern "C" int xlating_batch_fetch(xlating_batch *b) {
  if (b == nullptr) return -EINVAL;
  int rc = xlating_batch_sync(b);
  if (rc != 0) return rc;
  if (b->out_total == 0 || b->d_out[b->ocur] == nullptr) {
    b->fetched = true;
    return 0;
  }
  // (clients that joined since the latest call have rows -- assigned at add_client -- that may lie beyond what the device images hold:
  // those grow with the next plan.  Only what exists is copied; such a client has no outputs yet.)
  const size_t n = std::min(b->out_total, b->out_alloc);
  if (n > b->h_out_alloc) {
    if (b->h_out) (void)hipHostFree(b->h_out);
    b->h_out = nullptr;
    b->h_out_alloc = 0;
    if (hipHostMalloc((void **)&b->h_out, n * sizeof(float2), hipHostMallocDefault) != hipSuccess)
      return -ENOMEM;
    b->h_out_alloc = n;
  }
  if (hipMemcpyAsync(b->h_out, b->d_out[b->ocur], n * sizeof(float2), hipMemcpyDeviceToHost, b->own_stream) != hipSuccess)
    return -EIO;
  if (hipStreamSynchronize(b->own_stream) != hipSuccess) return -EIO;
  b->fetched = true;
  return 0;
}

extern "C" int xlating_batch_output_host_cs16(xlating_batch *b, int id, const int16_t **output, size_t *output_len) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive || !b->fetched ||
      output == nullptr || output_len == nullptr || !b->last_q15)
    return -EINVAL;
  const Client &c = b->clients[id];
  if (c.last_K == 0 || (size_t)c.out_off + c.last_K > b->h_out_alloc) {  // (joined since the latest call: a row, but nothing in it yet)
    *output = nullptr;
    *output_len = 0;
    return c.last_K == 0 ? 0 : -EINVAL;
  }
  *output = b->h_out ? reinterpret_cast<const int16_t *>(b->h_out + c.out_off) : nullptr;
  *output_len = c.last_K;
  return 0;
}

extern "C" int xlating_batch_output_host(xlating_batch *b, int id, const float **output, size_t *output_len) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive || !b->fetched ||
      output == nullptr || output_len == nullptr || b->last_q15)
    return -EINVAL;
  const Client &c = b->clients[id];
  // (a client that joined since the latest call has a row -- assigned at add_client -- but no outputs, and its row may lie
  // beyond what the fetched image holds: the images grow with the next plan)
  if (c.last_K == 0 || (size_t)c.out_off + c.last_K > b->h_out_alloc) {
    *output = nullptr;
    *output_len = 0;
    return c.last_K == 0 ? 0 : -EINVAL;
  }
  *output = b->h_out ? reinterpret_cast<const float *>(b->h_out + c.out_off) : nullptr;
  *output_len = c.last_K;
  return 0;
}

extern "C" int xlating_batch_output_device(xlating_batch *b, int id, const void **d_output, size_t *output_len) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive || d_output == nullptr ||
      output_len == nullptr || b->d_out[b->ocur] == nullptr)  // (a dirty engine still holds the latest call's rows: the plan
                                                               // is only rebuilt by the next process call)
    return -EINVAL;
  const Client &c = b->clients[id];
  if (c.last_K == 0 || (size_t)c.out_off + c.last_K > b->out_alloc) {  // (joined since the latest call: a row, but nothing in it yet)
    *d_output = nullptr;
    *output_len = 0;
    return c.last_K == 0 ? 0 : -EINVAL;
  }
  *d_output = b->d_out[b->ocur] + c.out_off;
  *output_len = c.last_K;
  return 0;
}

extern "C" int xlating_batch_client_phase(xlating_batch *b, int id, float *re, float *im) {
  if (b == nullptr || id < 0 || (size_t)id >= b->clients.size() || !b->clients[id].alive) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  xl_batch_sync_all(b);
  // the look-ahead NCO role has already produced the phases AFTER the next call into d_phase[pcur^1];
  // the committed ones (after the latest processed call) are d_phase[pcur]
  float2 ph;
  if (hipMemcpy(&ph, b->d_phase[b->pcur] + id, sizeof(ph), hipMemcpyDeviceToHost) != hipSuccess) return -EIO;
  *re = ph.x;
  *im = ph.y;
  return 0;
}

static int xl_batch_drain_events(xlating_batch *b) {
  xl_batch_sync_all(b);
  for (size_t i = 0; i + 2 <= b->ev.size(); i += 2) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]) != hipSuccess) return -EIO;
    b->fir_ms += ms;
    b->timed_launches++;
  }
  for (size_t i = 0; i + 2 <= b->ev_ncot.size(); i += 2) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, b->ev_ncot[i], b->ev_ncot[i + 1]) != hipSuccess) return -EIO;
    b->nco_ms += ms;
    b->timed_nco++;
  }
  for (size_t i = 0; i + 4 <= b->ev_poly.size(); i += 4) {
    for (int k = 0; k < 3; ++k) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, b->ev_poly[i + k], b->ev_poly[i + k + 1]) != hipSuccess) return -EIO;
      b->poly_ms[k] += ms;
    }
    b->timed_poly++;
  }
  b->ev_pool.insert(b->ev_pool.end(), b->ev_poly.begin(), b->ev_poly.end());
  b->ev_poly.clear();
  b->ev_pool.insert(b->ev_pool.end(), b->ev.begin(), b->ev.end());
  b->ev_pool.insert(b->ev_pool.end(), b->ev_ncot.begin(), b->ev_ncot.end());
  b->ev.clear();
  b->ev_ncot.clear();
  return 0;
}

// tuning: cycles / wall ticks of the chain wave of the first `n` chain workgroups of the latest side-stream launch
extern "C" int xlating_batch_debug_chain_stats(xlating_batch *b, unsigned long long *out, int n) {
  if (b == nullptr || out == nullptr || b->d_chain_stats == nullptr || n < 1 || n > 4096) return -EINVAL;
  xl_batch_sync_all(b);
  return hipMemcpy(out, b->d_chain_stats, (size_t)n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -EIO;
}

extern "C" int xlating_batch_timing(xlating_batch *b, int enable) {
  if (b == nullptr) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  if (b->timing && !enable) (void)xl_batch_drain_events(b);
  if (enable > 0 && b->ev_pool.size() < 2048) {  // pre-create so that the timed region itself creates none
    for (int i = 0; i < 2048; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) break;
      b->ev_pool.push_back(e);
    }
  }
  b->timing = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return 0;
}

extern "C" int xlating_batch_timing_stride(xlating_batch *b, unsigned every_n) {
  if (b == nullptr) return -EINVAL;
  b->timing_every = every_n ? every_n : 1;
  return 0;
}

extern "C" int xlating_batch_timing_polyphase(xlating_batch *b, double ms_total[3], int reset) {
  if (b == nullptr || ms_total == nullptr) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  int rc = xl_batch_drain_events(b);
  if (rc != 0) return rc;
  for (int k = 0; k < 3; ++k) ms_total[k] = b->poly_ms[k];
  const int n = b->timed_poly;
  if (reset) {
    b->poly_ms[0] = b->poly_ms[1] = b->poly_ms[2] = 0.0;
    b->timed_poly = 0;
  }
  return n;
}

extern "C" int xlating_batch_timing_read(xlating_batch *b, double *fir_ms_total, double *nco_ms_total, int reset) {
  if (b == nullptr) return -EINVAL;
  if (hipSetDevice(b->device) != hipSuccess) return -EIO;
  int rc = xl_batch_drain_events(b);
  if (rc != 0) return rc;
  if (fir_ms_total) *fir_ms_total = b->fir_ms;
  // the NCO launches are not one-to-one with calls (one extra look-ahead): report the per-call equivalent
  if (nco_ms_total) *nco_ms_total = b->timed_nco ? b->nco_ms / b->timed_nco * b->timed_launches : 0.0;
  const int n = b->timed_launches;
  if (reset) {
    b->fir_ms = b->nco_ms = 0.0;
    b->timed_launches = b->timed_nco = 0;
  }
  return n;
}
