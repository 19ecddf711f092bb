/*
 * ntc_oracle.h — CPU restatement of ntCard's ntHash -> sample -> count hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / reported CPU baseline.  The product path (ntcard_amd/csrc) never links it.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Parity status: PINNED — checked against the reference's own known-answer vector
 * (vendor/ntHash/unittest/UnitTests.cpp:39,45-48) and, in the build container, against the real
 * reference compiled into oracle/_ref (see oracle/Makefile, tests/test_oracle_vs_ref.py) and the
 * committed fixtures under tests/golden/ that were produced by that reference build.
 */
#ifndef NTC_ORACLE_H
#define NTC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1/a2/a3: seeds and the split rotate ------------------------------------------------- */
uint64_t orc_seed(uint8_t c);            /* nthash.hpp:25-64 seedTab[c]                         */
uint64_t orc_seed_comp(uint8_t c);       /* nthash.hpp:16,232 seedTab[c & cpOff]                */
uint64_t orc_srol(uint64_t x, unsigned n);   /* nthash.hpp:66-183,186-217: 33/31-bit split rotate */
uint64_t orc_sror1(uint64_t x);          /* nthash.hpp:191-193,214-217                          */

/* ---- a7: closed-form hash of one clean k-window (nthash.hpp:220-239,467-492) --------------- */
/* returns 0 and sets *loc_bad to the index of the LAST non-ACGTU byte when the window is dirty */
int orc_window_hash(const char *w, unsigned k, uint64_t *fh, uint64_t *rh, unsigned *loc_bad);

/* ---- a8: ntHashIterator semantics (ntHashIterator.hpp:59-86) -------------------------------- */
/* emits, in order, the canonical hash (and start position) of every window of k clean bases.   */
/* out_hash / out_pos may be NULL (count only).  Returns the number of windows (== F1 share).   */
size_t orc_hash_read(const char *seq, size_t len, unsigned k,
                     uint64_t *out_hash, uint32_t *out_pos, size_t cap);

/* ---- a9: stHashIterator / NTMSM64 with one seed, m=m2=1 (stHashIterator.hpp:53-87,
 *          nthash.hpp:620-678).  gap_pos = don't-care positions (parseSeed, :23-33)           */
size_t orc_gap_positions(unsigned k, unsigned gap, uint32_t *out_pos /* cap >= gap */);  /* ntcard.cpp:407-413 */
size_t orc_sthash_read(const char *seq, size_t len, unsigned k,
                       const uint32_t *gap_pos, size_t n_gap,
                       uint64_t *out_hash, uint32_t *out_pos, size_t cap);

/* multi-hash extension h_i = f(h_0, i, k) (nthash.hpp:381-390), used only for the h=3 KAT     */
uint64_t orc_multihash(uint64_t h0, unsigned i, unsigned k);

/* ---- a10: ntComp — which sample (0/1) accepts h, or 2 for none (ntcard.cpp:132-145) -------- */
unsigned orc_sample_of(uint64_t h, unsigned s_bits);

/* ---- a11-a13: ntRead/stRead over a batch of reads into t_Counter (ntcard.cpp:147-171,437-466)
 * counters: uint16_t [n_k][2][1<<r_bits], updated with wrapping 16-bit atomic increments.
 * bases/offsets: concatenated raw read bytes, read i = bases[offsets[i] .. offsets[i+1]).
 * f1[n_k] is ADDED to.  gap != 0 requires n_k == 1.  n_threads <= 0 -> omp default.           */
void orc_sketch_update(uint16_t *counters, const char *bases, const uint64_t *offsets,
                       uint64_t n_reads, const uint32_t *klist, uint32_t n_k, uint32_t gap,
                       uint32_t r_bits, uint32_t s_bits, uint64_t *f1, int n_threads);

/* ---- C6: compEst (ntcard.cpp:237-275) -------------------------------------------------------- */
/* p-histogram of counter values: p[2][65536] (ntcard.cpp:240-247) */
void orc_value_hist(const uint16_t *counters_k, uint32_t r_bits, uint32_t *p /* [2][65536] */);
/* estimator proper from p (ntcard.cpp:249-274).  f_mean has 65536 doubles.  `limit` (1..65535)
 * stops the f_i recurrence early: f_i for i <= limit only depends on indices <= i, so the
 * entries that are computed are identical to the reference's; entries above limit are 0.        */
void orc_comp_est_p(const uint32_t *p, uint32_t r_bits, uint32_t s_bits, uint32_t limit,
                    double *F0, double *f_mean);
void orc_comp_est(const uint16_t *counters_k, uint32_t r_bits, uint32_t s_bits, uint32_t limit,
                  double *F0, double *f_mean);

/* ---- C7: outDefault body for one k (ntcard.cpp:291-294): returns bytes written (excl. NUL)   */
size_t orc_format_hist(uint64_t f1, double F0, const double *f_mean, uint32_t cov_max,
                       char *buf, size_t cap);

/* ---- a15: nthll (nthll.cpp:92-105 ntComp/ntRead, :238-243 max merge, :247-254 estimate) ------------ */
/* regs: uint8_t [1<<n_bits], updated with max; n_threads <= 0 -> omp default                      */
void orc_hll_update(uint8_t *regs, uint32_t n_bits, const char *bases, const uint64_t *offsets,
                    uint64_t n_reads, uint32_t k, int n_threads);
double orc_hll_estimate(const uint8_t *regs, uint32_t n_bits);

/* ---- synthetic read generator (spec: DESIGN.md "Synthetic workloads"; no reference analogue) */
/* dist 0 = uniform i.i.d. ACGT; dist 1 = reads from an implicit random genome, 1 % subs, 0.05 % N */
/* ntRead for plain k-mers with per-k precomputed seed tables, one thread per shard of the batch (the reference's
 * threading shape, ntcard.cpp:445-446): the timed cpu_baseline form; same results as orc_sketch_update(gap = 0). */
void orc_sketch_update_sharded(uint16_t *counters, const char *bases, const uint64_t *offsets,
                               uint64_t n_reads, const uint32_t *klist, uint32_t n_k,
                               uint32_t r_bits, uint32_t s_bits, uint64_t *f1, int n_threads);

void orc_gen_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                   uint32_t stride, uint32_t dist, uint64_t genome_len, uint8_t *out);

/* FNV-1a 64 digest helper for large buffers */
uint64_t orc_fnv1a64(const void *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
