/* xl_plan_rules.h -- the two SIZE RULES of the engine's plan that were set from measurements of round 5, as plain C so that the
 * CPU suite can pin them (tests/test_plan_rules.py): which inverse kernel a polyphase launch takes, and how many CUs the side-stream
 * recurrence kernel gets.  No HIP types. */
#ifndef XL_PLAN_RULES_H_
#define XL_PLAN_RULES_H_
#include <stdint.h>

/* Which inverse kernel a launch of `tiles` tiles (of 32 columns x one segment) takes: 3 = the transform staged in LDS
 * (xlp_inverse_kernel: the only one for 256-point classes), 5 = eight lanes per column (xl_inv8.hip), 6 = the 32 x 4 cut
 * (xl_inv32.hip).  Option "inverse_kernel": 0 = this rule, 3 / 5 / 6 = that kernel for every 128-point launch.  Measured alternating
 * in one process on every box of round 5 (profiles/r05_inverse_ab_same_box.txt, r05_inverse_cut32.txt, bench.py's "inverse launch
 * A/B"):
 *   up to ~2000 tiles (one block per call at up to 2048 clients): the 8-lane kernel, 5-9 % ahead -- no table fill, the least work per
 *     tile: it is the launch's latency that counts there;
 *   from ~13 000 tiles (8 blocks per call at >= 2048 clients): the 32 x 4 cut, 6-10 % ahead of the LDS transform, which beat the
 *     8-lane kernel by 3-5 % there -- whole-line loads, 256-byte store runs and a third fewer instructions;
 *   in between: the LDS transform -- 10 % ahead of both others at 2688 tiles (BASELINE config 5 at 1024 clients: 45.4 against
 *     49.8 / 50.2 us per launch), 3 % at 3456 (4096 clients, one block), level with the 32 x 4 cut at 6912 (1024 clients x 8 blocks,
 *     where the NCO recurrence bounds the call anyway). */
static inline uint32_t xlp_inverse_pick(uint32_t M, uint32_t inv_reg, uint32_t tiles) {
  if (M != 128u) return 3u;
  if (inv_reg == 3u || inv_reg == 5u || inv_reg == 6u) return inv_reg;
  return tiles <= 2048u ? 5u : (tiles <= 8192u ? 3u : 6u);
}

/* CUs per XCD reserved for the side-stream recurrence (chain) kernel, whose workgroups -- one per 64 clients -- each own a CU
 * (xl_kernels.hip: one chain wave per SIMD by register exhaustion).  `chain_wgs` = ceil(clients / 64); `load_wgs` = the same count
 * scaled by how much launch time a client of this plan brings per unit of chain time, relative to the shape the bands were measured
 * on (xl_plan_load_wgs below; = chain_wgs for the server default: 2.016 Msps cu8, 48 kHz clients, 505 taps).
 *   load up to 32 workgroups (2048 default-shape clients): one CU per workgroup (round 3's rule).  The recurrence bounds or nearly
 *     bounds the call there (1024 clients: 194 of 198 us; 1536: 212 of 280), and without CUs of its own it runs 25-35 % slower.
 *   33 .. 47 (2049 .. 3008 clients): NONE -- the chain workgroups take whole CUs as the launches' tails free them.  One CU each would
 *     be 40-48 CUs (a sixth of the chip) for a kernel that is busy half the call, and two rounds on half as many do not fit the
 *     calls yet: measured (profiles/r05_chain_reservation.txt (3)) 2304 / 2560 / 2816 clients 49.1-49.9 / 53.9-54.7 / 58.2-58.6 us
 *     per block without against 53.7-54.2 / 57.2-58.1 / 62.3-63.2 with the reservation (2048: level).
 *   from 48 (3009+ clients): the chain launch runs in ROUNDS on fewer CUs (its workgroups queue on the stream's CU mask), as many as
 *     fit the calls it looks ahead of -- the chain's time per client does not grow with the client count, the launches' does: 2
 *     rounds from 3072 clients (4096: 82.3 against 87.5 us per block with one CU per workgroup; 3 rounds there make the chain the
 *     bound again: 85.1), 3 from 5120, 4 from 7168 (5120 / 6144 / 8192 clients: 100-102 / 118-121 / 155-160 against 115-119 /
 *     138-142 / 217-226; no reservation: 103 / 119-123 / 160-163).
 * More than 16 CUs per XCD (half the chip) is never reserved: 0 then. */
static inline uint32_t xl_chain_rounds(uint32_t load_wgs) {
  const uint32_t r = (64u * load_wgs + 1024u) / 2048u;
  return r < 1u ? 1u : r;
}
static inline uint32_t xl_chain_reserve_per_xcd(uint32_t chain_wgs, uint32_t load_wgs) {
  const uint32_t rounds = xl_chain_rounds(load_wgs);
  if (rounds == 1u && load_wgs > 32u) return 0u;
  /* (rounds only for plans whose clients weigh about what the measured shape's do: where a client brings 1.5 x the launch time per
   * unit of chain time or more, the chain kernel is a small part of the call, even a few reserved CUs idle most of it, and no
   * reservation measures better -- BASELINE config 5 at 2048 / 4096 clients: 45.1 / 72.3 us per block without against 47.4-48.0 /
   * 75.6-75.9 in rounds and 48.1-49.0 / 82.8-83.5 with one CU per chain workgroup) */
  if (rounds > 1u && 2u * load_wgs > 3u * chain_wgs) return 0u;
  const uint32_t per_round = 8u * rounds;
  const uint32_t want = (chain_wgs + per_round - 1u) / per_round;
  return want > 16u ? 0u : want;
}

/* Which band of the rule above a load falls in: 0 = one CU per chain workgroup, 1 = none, 2 = rounds.  The rule is NOT monotonic in the
 * client count (CUs up to 32 workgroups, none for 33..47, rounds from 48) and re-creating the CU-masked stream pair costs ~25 ms plus a
 * full synchronisation, so the plan keeps the band it is in until the load has moved XL_BAND_HYST workgroups past an edge: a population
 * hovering at 2048 / 2049 or 3008 / 3009 clients then stays where it is instead of destroying and re-creating the streams at every
 * crossing.  `last_band` < 0: no history.  Returns the load to evaluate the rule at (the load itself, or the nearest load of the band
 * the plan stays in). */
#define XL_BAND_HYST 2u
static inline int xl_chain_band(uint32_t load_wgs) { return xl_chain_rounds(load_wgs) > 1u ? 2 : (load_wgs > 32u ? 1 : 0); }
static inline uint32_t xl_chain_load_with_hysteresis(uint32_t load_wgs, int last_band) {
  if (last_band < 0 || xl_chain_band(load_wgs) == last_band) return load_wgs;
  for (uint32_t d = 1u; d <= XL_BAND_HYST; ++d) {
    if (load_wgs >= d && xl_chain_band(load_wgs - d) == last_band) return load_wgs - d;
    if (xl_chain_band(load_wgs + d) == last_band) return load_wgs + d;
  }
  return load_wgs;
}

/* The load of a plan, for the rule above.  What decides between the bands is the ratio of the launches' time to the chain kernel's,
 * and a client of another shape brings another ratio: BASELINE config 5 (cf32 10 Msps, D = 100: 1311 recurrence steps per block
 * instead of 3121, a float32 mix of 100 branches) has 2.2 x the default's, and indeed does better WITHOUT reserved CUs from 1024
 * clients on (22.6-22.8 against 23.4-23.9 us per block; 2048: 45.2 against 48.2-49.0; 512: reserved 15.3-15.4 against 15.6-15.9:
 * profiles/r05_chain_reservation.txt (6)).  Both times are modelled per (client, block) from the plan's numbers, with the rates the
 * launches and the chain kernel were measured at on this chip:
 *   inverse launch   (nseg M 8 + K 8) bytes at 4.5 TB/s (= bytes per ps)   (reads the mixed spectra, writes the outputs)
 *   mix, two-half    (nseg M 8 + operands / blocks per call) bytes at 4.8 TB/s, operands = 8 Dpad M bytes
 *   mix, float32     8 nseg M Dpad flop at 79 TFLOP/s (= flop per ps)      (the float32 matrix pipe at half its peak)
 *   chain kernel     7.9 ns per output (16.5 cycles at the clock a loaded chip leaves it) + 1 us
 * with K = outputs per block, nseg = K / V segments.  xl_client_load() = launch ns / chain ns of one client (x 1000). */
static inline uint32_t xl_client_launch_ps(uint32_t M, uint32_t K, uint32_t V, uint32_t Dpad, uint32_t mix_kind, uint32_t blocks_per_call) {
  const double nseg = (double)K / (double)(V ? V : 1u), g = blocks_per_call ? (double)blocks_per_call : 1.0;
  const double inv = (nseg * M * 8.0 + K * 8.0) / 4.5;                                                                  /* ps: bytes / (4.5 bytes per ps x 1000) ... */
  const double mix = mix_kind == 3u ? 8.0 * nseg * M * Dpad / 79.0 : (nseg * M * 8.0 + 8.0 * Dpad * M / g) / 4.8;  /* ... = bytes / 4.5 with the rate in TB/s */
  return (uint32_t)(inv + mix + 0.5);
}
static inline uint32_t xl_chain_block_ns(uint32_t K) { return (uint32_t)(7.9 * K + 1000.0); }
/* launch time of all clients / chain time, in units of the measured shape's per-client ratio x 64: the client count of that shape
 * with the same ratio, in chain workgroups.  `launch_ps_sum` = sum of xl_client_launch_ps over the clients (+ what direct-kernel
 * clients of the call add), `kmax` = most outputs per block of any client */
static inline uint32_t xl_plan_load_wgs(double launch_ps_sum, uint32_t kmax) {
  const double ref = (double)xl_client_launch_ps(128u, 3121u, 116u, 48u, 1u, 8u) / (double)xl_chain_block_ns(3121u);  /* the server default */
  const double eq_clients = launch_ps_sum / (double)xl_chain_block_ns(kmax) / ref;
  return (uint32_t)((eq_clients + 63.0) / 64.0);
}

#endif /* XL_PLAN_RULES_H_ */
