// xl_polyphase.h -- launchers of the polyphase overlap-save path (xl_polyphase.hip): the batch engine's OPTIMIZED
// arithmetic for big classes of clients with long filters.  Same linear operator as xlating.c:52-72, evaluated in
// the frequency domain of the D polyphase branches:
//
//   y[k] = sum_{i<T} r[i] x[kD + i]                      (r = reversed band-pass taps, xlating.c:525-535)
//        = sum_{b<D} sum_{a<A} r_b[a] x_b[k + a]         r_b[a] = r[D a + b], x_b[n] = x[D n + b], A = ceil(T/D)
//
// i.e. D short correlations at the OUTPUT rate.  Over a segment of M branch samples (256; 128 for big classes of
// short filters: half the R stream; 64 for classes of many branches with very short branch filters) each correlation is a circular one, exact for the first V = M - A + 1 outputs:
//
//   y_seg = IDFT_M( sum_b DFT_M(x_b) * R_b ),            R_b[m] = sum_a r_b[a] e^{+2 pi j a m / M}
//
// Clients of one (D, T) whose output grids are offset against each other (they joined at different stream positions:
// dsp_worker.c:98-104, xlating.c:552) still form ONE class: a client whose grid lies delta samples behind the class's
// shared grid is evaluated with its taps delayed by delta samples -- delta leading zeros, at most one more tap per
// branch -- which is the same sum (xl_grid.h, XlpCol::delta).
//
// DFT_M(x_b) is shared by ALL clients of the class (D transforms per segment, whatever the client count) and the
// per-client work is D complex MACs per spectrum bin plus one inverse transform per segment: ~45 complex MACs per
// output instead of T (= 505 at the server default) -- and those MACs are a matrix product per bin (clients x branches x
// segments), which runs on the matrix cores with two-half operands where the input format bounds the spectra.  The NCO derotation (exact float32 recurrence table) and the
// streaming rules (global output grid, history, late joiners) are those of the direct kernels.
// Measured against the oracle: max|d|/max|y| <= 1e-6, which is the reference's own float32 summation noise (an
// all-double evaluation of the same operator differs from the reference by the same amount).
#ifndef XL_POLYPHASE_H_
#define XL_POLYPHASE_H_
#include "xl_device.h"
#include "xl_plan_rules.h"

#define XLP_M_MAX 256u  // transform length M (branch samples per segment): 256, 128 or 64, chosen per class (xl_batch.cpp: xl_poly_pick_m)
#define XLP_SEG 16u    // segments per pass of the mix launches: with (re, im) the 32 rows of a matrix instruction (round 5; 14 before: the
                       // packed-FMA mix kernel's register budget -- 16 took 7 % off the two-half mix launch, profiles/r05_mix_f32.txt)
#define XLP_XS 16u     // row stride (complex) of the shared-spectrum image: the XLP_SEG segments of a pass = 128 bytes
#define XLP_COLS 128u  // client columns per column group (= one mix workgroup: a wave with two columns per lane)
#define XLP_NKB_MAX 14u // matrix-core mix: at most 14 k-blocks of 8 branches (D <= 112): a wave keeps its B operands in registers
#define XLP_NKB_4W 8u   // ... up to 8 k-blocks (D <= 64) at four waves per SIMD, above that at two (xlp_mix_mfma_kernel)
#define XLP_H_XSCALE 128.0f  // matrix-core mix: the shared spectra are multiplied by this before the split in halves: integer input
                             // formats give |X| <= M sqrt(2) <= 363, so the first half stays below 46 400 < 65 504.
                             // A cf32 stream has no such bound: its spectra are scaled per SEGMENT by the power of two that brings the
                             // segment's largest component (XlpArgs::segmax, found by the forward launch) under 2^15 -- a row scale of
                             // the per-bin matrix product, undone exactly in the mix launch's epilogue (xlp_seg_scale)
#define XLP_H_RMAX 8192.0f   // ... and a column's spectra by the power of two that brings their bound max_b sum_a |r_b[a]| under this
#define XLP_SEGMAX_STRIDE 32u  // XlpArgs::segmax: one entry per 128-byte line (the forward launch's atomics spread over the L2 channels)
#define XLP_FWD_GROUP_MIN_WGS 96u  // forward launch, cf32 streams: groups of adjacent branches per workgroup from this many workgroups on
#define XLP_BSTEP 2u   // the branch count is padded to a multiple of this in the shared-spectrum image (rows D .. Dpad - 1: zeros)

// Columns of one tile of the mixed spectra Y = [cg][segment][sub][bin M][CW columns] (one inverse workgroup's tile, M x CW x 8 bytes =
// 32 KB contiguous): 16 (M = 256), 32 (M = 128), 64 (M = 64)
static inline __host__ __device__ uint32_t xlp_tile_columns(uint32_t M) { return M == 256u ? 16u : (M == 64u ? 64u : 32u); }

// One client column of a class: 16 bytes, one load.
struct XlpCol {
  uint32_t out_off;  // float2 index of the client's row in out (/ XL_PH_STRIDE in the phase table); 0xFFFFFFFF = empty column
  uint32_t delta;    // offset of the client's output grid against the class's shared grid, 0 .. D-1 (xl_grid.h); its
                     // branch spectra are those of its taps delayed by delta samples
  float2 incr;       // NCO phase increment
};

struct XlpArgs {
  // input stream [in0 | in1] as in XlFirArgs (in1 = the G blocks of the call, contiguous)
  const void *in0;
  const void *in1;
  uint32_t n0, n1;
  uint32_t fmt;
  XlPos pos;           // stream position of the call (per-column output counts, block boundaries, the NCO role)
  uint32_t j0_ref;     // shared grid of the class in this call: j0 of its virtual reference client (xl_grid.h)
  uint32_t base;       // sample index in [in0 | in1] coordinates of the first tap of shared point q = 0
  uint32_t Kq;         // shared points evaluated, q < Kq
  uint32_t zero_below; // samples below read as 0
  uint32_t D, Dpad;    // decimation = number of branches; padded to a multiple of XLP_BSTEP in the images
  uint32_t T, A, V;    // taps, taps per branch of the delayed filters, valid outputs per segment = M - A + 1
  uint32_t M;          // transform length: 256, 128 or 64
  uint32_t mix_passes; // (set by xlp_launch_mix) passes of XLP_SEG segments = ceil(nseg / XLP_SEG)
  uint32_t nseg;       // segments of this call = ceil(Kq / V)
  uint32_t nseg_cap;   // segment capacity of the Y image
  uint32_t ncg;        // column groups of XLP_COLS client columns
  uint32_t exp;        // tuning switches (XL_TUNING builds only; 0 otherwise)
  uint32_t inv_reg;    // M = 128: the inverse launch's transform: 5 = registers of eight lanes per column (xl_inv8.hip), 6 = cut 32 x 4
                       // (xl_inv32.hip), 3 = staged in LDS on swizzled rows (xlp_inverse_kernel<128>), 0 = by the launch's size
                       // (xlp_inverse_pick)
  uint32_t mix_kind;   // the mix launch: 1 = matrix cores on two-term half splits (xlp_mix_mfma_kernel), 3 = matrix cores with float32
                       // operands (xl_mixf32.hip: xlp_mix_f32_kernel; any input format, any branch count)
  uint32_t nkb;        // k-blocks of 8 branches = ceil(D / 8) (mix_kind 1: <= XLP_NKB_MAX)
  uint32_t mix_pp;     // passes per workgroup of the mix launch (0 = the launcher's default)
  // two-half mix of a cf32 stream: per segment the largest |component| of its shared spectra (float bits; all branches, all bins),
  // gathered by the forward launch with one atomicMax per (segment, branch).  Two buffers of seg_cap entries: a call uses buffer
  // seg_par, and its forward launch clears the other one for the next call; entry i at word i * XLP_SEGMAX_STRIDE.  nullptr: integer formats (constant scale XLP_H_XSCALE)
  uint32_t *segmax;
  uint32_t seg_par, seg_cap;
  unsigned long long *trace;  // tuning only: [0..2] min start / max end of the work waves, [8 + 4 i ..] per NCO wave: start, loaded, end
  const float2 *W;     // e^{-2 pi j n / 256}, n < 256
  float2 *X;           // shared spectra   [pass][Dpad][M][XLP_XS]
  const void *Rh;      // branch spectra in the mix launch's B-operand order.  mix_kind 1: scaled per column and split in two halves,
                       //   [cg][M][32-column quarter][term 2][k-block nkb][lane 64][8 halves] (see xlp_mix_mfma_kernel)
                       // mix_kind 3: the same values as float32 (R.re, -R.im) in v_mfma_f32_32x32x2_f32's B-operand order
                       //   [cg][M][32-column quarter][k-block nkb][half 2][lane 64][4 branches] (xl_mixf_layout.h)
  const float *cscale; // mix_kind 1: per column, what the sums are multiplied by = 1 / (column scale * XLP_H_XSCALE) (segmax: 1 / column scale)
  float2 *Y;           // mixed spectra    [cg][nseg_cap][sub][M][CW], CW = 32 (M = 128) / 16 (M = 256) columns: one inverse tile contiguous
  const XlpCol *cols;  // per column
  const float2 *phtab;
  float2 *out;
  // raw-history roll, carried by the forward launch (as XlFirArgs): null / 0 = none
  void *hist_out;
  uint32_t hist_units, block_units, roll_blocks;
  // NCO role pieces (see XlFirArgs): the forward and the inverse launch of a call each carry a slice [nco_k0, nco_k1) of the
  // NEXT call's phase recurrence (stream position xl_grid_next(pos)); block ends inside a slice renormalise
  // (xlating.c:73).
  const XlNcoClient *nco_clients;
  uint32_t nco_nclients;
  uint32_t nco_blocks;
  uint32_t nco_prio;        // wave priority of the role (0..3)
  uint32_t nco_skip_at, nco_skip;  // inverse launch: workgroups [nco_skip_at, nco_skip_at + nco_skip) exit at once (a slot kept empty on the role's CUs)
  uint32_t nco_k0, nco_k1;  // in 1/65536 of the call's outputs: slice = [K*k0 >> 16, K*k1 >> 16) rounded down to pairs of table entries
  const float2 *nco_state_src;  // phases at the start of the slice (committed phases for the first slice)
  float2 *nco_state_dst;        // phases after the slice (committed post-call phases for the last slice)
  float2 *nco_tab;
};

// reversed band-pass taps of a LIST of columns -> their branch spectra (double arithmetic, rounded once to float) in the two-half
// mix's operand form (XlpArgs::Rh)
//   rt: [T][nlist] float2 (tap-major); delta[j]: delay of entry j's taps in samples (xl_grid.h: merged classes);
//   colidx[j]: the column the entry goes to; scale[j]: the entry's power-of-two column scale; A = ceil((T + max delta) / D)
hipError_t xlp_launch_tables_h(const float2 *rt, const uint32_t *delta, const uint32_t *colidx, const float *scale,
                               uint32_t nlist, uint32_t T, uint32_t D, uint32_t A, uint32_t M, uint32_t nkb, void *Rh,
                               hipStream_t s);
// bytes of that image per column group
static inline size_t xlp_rh_bytes_per_group(uint32_t M, uint32_t nkb) { return (size_t)M * 4u * 2u * nkb * 64u * 16u; }
// mix_kind 3 (xl_mixf32.hip): the branch spectra as float32 B operands, and the mix launch itself (called by xlp_launch_mix)
hipError_t xlp_launch_tables_f(const float2 *rt, const uint32_t *delta, const uint32_t *colidx, uint32_t nlist, uint32_t T, uint32_t D,
                               uint32_t A, uint32_t M, uint32_t nb8, void *Rf, hipStream_t s);
hipError_t xlp_launch_mix_f32(const XlpArgs &a, hipStream_t s);
// (xl_mixh2.hip: the two-half mix of 9 .. XLP_NKB_MAX k-blocks; called by xlp_launch_mix with the checked arguments)
void xlp_mix_wide_launch(const XlpArgs &a, hipStream_t s);
hipError_t xlp_launch_forward(const XlpArgs &a, hipStream_t s);
hipError_t xlp_launch_mix(const XlpArgs &a, hipStream_t s);
hipError_t xlp_launch_inverse(const XlpArgs &a, hipStream_t s, hipEvent_t done);
// Which inverse kernel a launch takes (option "inverse_kernel" = 0): xlp_inverse_pick, xl_plan_rules.h.
// (xl_inv8.hip: the 8-lane kernel; called by xlp_launch_inverse with the checked arguments and the launch's grid)
void xlp_inverse8_launch(const XlpArgs &a, const dim3 grid, hipStream_t s, hipEvent_t done);
// (xl_inv32.hip: the 32 x 4 cut; its workgroups take half tiles)
uint32_t xlp_inverse32_work(uint32_t tiles);
void xlp_inverse32_launch(const XlpArgs &a, const dim3 grid, hipStream_t s, hipEvent_t done);

#endif  // XL_POLYPHASE_H_
