/*
 * ntc_oracle.c — CPU restatement of ntCard's ntHash -> sample -> count hot path (plain C, OpenMP).
 *
 * TEST INFRASTRUCTURE ONLY (see ntc_oracle.h).  Written from the behavioural spec in SURVEY.md
 * §8 / App. B; every function cites the reference lines (relative to /root/reference) it restates.
 * No reference source text is reproduced: the only reference *data* here are the four 64-bit base
 * seeds, the complement rule, the two sampling bit patterns and the estimator formula.
 */
#include "ntc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------------
 * a1: byte -> seed.  nthash.hpp:25-28 (the four constants), :31-64 (which bytes map to which).
 * A/a, C/c, G/g, T/t and U/u carry a seed; every other byte is "N" (seed 0).
 * ------------------------------------------------------------------------------------------- */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL

uint64_t orc_seed(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return SEED_A;
    case 'C': case 'c': return SEED_C;
    case 'G': case 'g': return SEED_G;
    case 'T': case 't': case 'U': case 'u': return SEED_T;
    /* nthash.hpp:32 — table slots 0..7 double as the complement lookup (index = byte & 7) */
    case 1: return SEED_T;
    case 3: return SEED_G;
    case 4: case 5: return SEED_A;
    case 7: return SEED_C;
    default: return 0;
    }
}

/* nthash.hpp:16 (cpOff = 7) and :232,:252-253 — complement seed = table[byte & 7] */
uint64_t orc_seed_comp(uint8_t c) { return orc_seed((uint8_t)(c & 7u)); }

/* ---------------------------------------------------------------------------------------------
 * a2/a3: split rotate.  Bits [0,33) rotate left as a 33-bit word, bits [33,64) as a 31-bit word.
 * nthash.hpp:186-188 + :208-211 (rol1 then swap bits 0/33) is srol by 1; the msTab33r/msTab31l
 * tables (:66-183) hold srol^i(seed) for i < 33 / 31.
 * ------------------------------------------------------------------------------------------- */
uint64_t orc_srol(uint64_t x, unsigned n)
{
    const uint64_t M33 = (1ULL << 33) - 1, M31 = (1ULL << 31) - 1;
    uint64_t lo = x & M33, hi = x >> 33;
    unsigned a = n % 33u, b = n % 31u;
    if (a) lo = ((lo << a) | (lo >> (33u - a))) & M33;
    if (b) hi = ((hi << b) | (hi >> (31u - b))) & M31;
    return lo | (hi << 33);
}

/* nthash.hpp:191-193 + :214-217 (ror1 then swap bits 32/63) is srol by -1 */
uint64_t orc_sror1(uint64_t x) { return orc_srol(x, 33u * 31u - 1u); }

/* ---------------------------------------------------------------------------------------------
 * a7: base hash of one window, N-aware.  nthash.hpp:467-492 scans i = k-1 .. 0, bails out at the
 * first dirty byte it meets (so loc_bad is the LAST dirty index), and accumulates
 *   fh = XOR_i srol^(k-1-i)(seed(c_i)),  rh = XOR_i srol^i(comp(c_i))      (:220-239)
 * ------------------------------------------------------------------------------------------- */
int orc_window_hash(const char *w, unsigned k, uint64_t *fh, uint64_t *rh, unsigned *loc_bad)
{
    uint64_t f = 0, r = 0;
    for (int i = (int)k - 1; i >= 0; --i) {
        if (orc_seed((uint8_t)w[i]) == 0) {
            if (loc_bad) *loc_bad = (unsigned)i;
            return 0;
        }
    }
    for (unsigned i = 0; i < k; ++i) {
        f = orc_srol(f, 1) ^ orc_seed((uint8_t)w[i]);
        r = orc_srol(r, 1) ^ orc_seed_comp((uint8_t)w[k - 1 - i]);
    }
    *fh = f;
    *rh = r;
    return 1;
}

/* a4/a5: one rolling step.  nthash.hpp:242-248 (forward), :251-257 (reverse). */
static inline void roll(uint64_t *fh, uint64_t *rh, unsigned k, uint8_t out, uint8_t in)
{
    *fh = orc_srol(*fh, 1) ^ orc_seed(in) ^ orc_srol(orc_seed(out), k);
    *rh = orc_sror1(*rh ^ orc_srol(orc_seed_comp(in), k) ^ orc_seed_comp(out));
}

/* a6: canonical value.  nthash.hpp:275-279 — (rh < fh) ? rh : fh, unsigned */
static inline uint64_t canon(uint64_t fh, uint64_t rh) { return rh < fh ? rh : fh; }

/* ---------------------------------------------------------------------------------------------
 * a8: ntHashIterator.  ntHashIterator.hpp:59-70 (init: skip forward past the last dirty byte of
 * the window until a clean window or the end), :73-86 (next: roll, or if the incoming byte is
 * dirty jump k ahead and re-init).  k > len yields nothing (:61-64).
 * ------------------------------------------------------------------------------------------- */
size_t orc_hash_read(const char *seq, size_t len, unsigned k,
                     uint64_t *out_hash, uint32_t *out_pos, size_t cap)
{
    size_t n = 0;
    if (k == 0 || (size_t)k > len) return 0;
    const size_t last = len - k; /* last valid window start */
    size_t pos = 0;
    uint64_t fh = 0, rh = 0;
    int have = 0;
    while (pos <= last) {
        if (!have) {
            unsigned bad = 0;
            if (!orc_window_hash(seq + pos, k, &fh, &rh, &bad)) {
                pos += (size_t)bad + 1;
                continue;
            }
            have = 1;
        }
        if (n < cap) {
            if (out_hash) out_hash[n] = canon(fh, rh);
            if (out_pos) out_pos[n] = (uint32_t)pos;
        }
        ++n;
        /* advance */
        ++pos;
        if (pos > last) break;
        uint8_t in = (uint8_t)seq[pos + k - 1];
        if (orc_seed(in) == 0) {
            pos += k;
            have = 0;
        } else {
            roll(&fh, &rh, k, (uint8_t)seq[pos - 1], in);
        }
    }
    return n;
}

/* ntcard.cpp:407-413 + stHashIterator.hpp:23-33: seed "1"x(k-g)/2 "0"xg "1"x(k-g)/2; the
 * don't-care positions are the indices of the non-'1' characters.                              */
size_t orc_gap_positions(unsigned k, unsigned gap, uint32_t *out_pos)
{
    unsigned ones = (k - gap) / 2;
    for (unsigned i = 0; i < gap; ++i) out_pos[i] = ones + i;
    return gap;
}

/* a9: spaced-seed value from the full-window fh/rh.  nthash.hpp:641-646 / :665-670 */
static inline uint64_t gapped(const char *w, unsigned k, uint64_t fh, uint64_t rh,
                              const uint32_t *gp, size_t ng)
{
    uint64_t fs = fh, rs = rh;
    for (size_t t = 0; t < ng; ++t) {
        unsigned i = gp[t];
        fs ^= orc_srol(orc_seed((uint8_t)w[i]), k - 1 - i);
        rs ^= orc_srol(orc_seed_comp((uint8_t)w[i]), i);
    }
    return rs < fs ? rs : fs;
}

/* stHashIterator.hpp:60-87 — same window walk as a8; a dirty byte anywhere in the k window
 * (gap included) kills the window (nthash.hpp:625-629).                                          */
size_t orc_sthash_read(const char *seq, size_t len, unsigned k,
                       const uint32_t *gap_pos, size_t n_gap,
                       uint64_t *out_hash, uint32_t *out_pos, size_t cap)
{
    size_t n = 0;
    if (k == 0 || (size_t)k > len) return 0;
    const size_t last = len - k;
    size_t pos = 0;
    uint64_t fh = 0, rh = 0;
    int have = 0;
    while (pos <= last) {
        if (!have) {
            unsigned bad = 0;
            if (!orc_window_hash(seq + pos, k, &fh, &rh, &bad)) {
                pos += (size_t)bad + 1;
                continue;
            }
            have = 1;
        }
        if (n < cap) {
            if (out_hash) out_hash[n] = gapped(seq + pos, k, fh, rh, gap_pos, n_gap);
            if (out_pos) out_pos[n] = (uint32_t)pos;
        }
        ++n;
        ++pos;
        if (pos > last) break;
        uint8_t in = (uint8_t)seq[pos + k - 1];
        if (orc_seed(in) == 0) {
            pos += k;
            have = 0;
        } else {
            roll(&fh, &rh, k, (uint8_t)seq[pos - 1], in);
        }
    }
    return n;
}

/* nthash.hpp:381-390: h_i = t ^ (t >> 27) with t = h_0 * (i ^ k * multiSeed)                      */
uint64_t orc_multihash(uint64_t h0, unsigned i, unsigned k)
{
    uint64_t t = h0 * ((uint64_t)i ^ ((uint64_t)k * 0x90b45d39fb6da1faULL));
    return t ^ (t >> 27);
}

/* ---------------------------------------------------------------------------------------------
 * a10: ntComp.  ntcard.cpp:132-145.  sMask = 2^(sBits-1) - 1 (ntcard.cpp:438).
 * ------------------------------------------------------------------------------------------- */
unsigned orc_sample_of(uint64_t h, unsigned s_bits)
{
    unsigned which = 2;
    uint64_t s_mask = (1ULL << (s_bits - 1)) - 1;
    if ((h >> (63 - s_bits)) == 1) which = 0;
    if ((h >> (64 - s_bits)) == s_mask) which = 1;
    return which;
}

/* ---------------------------------------------------------------------------------------------
 * a11-a13: ntRead / stRead over many reads.  ntcard.cpp:147-171 (per-k loop, F1 count),
 * :142-143 (wrapping atomic ++ on uint16_t), :437-439 (layout [k][sample][bucket]),
 * :464-466 (F1 merge).  The reference fans out one thread per FILE (:445); here the unit is a
 * block of reads, which is equivalent because every update is a commutative atomic.
 * ------------------------------------------------------------------------------------------- */
static void one_read(uint16_t *counters, const char *seq, size_t len, const uint32_t *klist,
                     uint32_t n_k, const uint32_t *gp, size_t ng, uint32_t r_bits, uint32_t s_bits,
                     uint64_t *f1_local)
{
    const uint64_t r_buck = 1ULL << r_bits;
    for (uint32_t ki = 0; ki < n_k; ++ki) {
        const unsigned k = klist[ki];
        uint16_t *plane = counters + (size_t)ki * 2 * r_buck;
        if (k == 0 || (size_t)k > len) continue;
        const size_t last = len - k;
        size_t pos = 0;
        uint64_t fh = 0, rh = 0;
        int have = 0;
        while (pos <= last) {
            if (!have) {
                unsigned bad = 0;
                if (!orc_window_hash(seq + pos, k, &fh, &rh, &bad)) {
                    pos += (size_t)bad + 1;
                    continue;
                }
                have = 1;
            }
            uint64_t h = ng ? gapped(seq + pos, k, fh, rh, gp, ng) : canon(fh, rh);
            unsigned which = orc_sample_of(h, s_bits);
            if (which < 2)
                __atomic_fetch_add(&plane[which * r_buck + (h & (r_buck - 1))], (uint16_t)1,
                                   __ATOMIC_RELAXED);
            ++f1_local[ki];
            ++pos;
            if (pos > last) break;
            uint8_t in = (uint8_t)seq[pos + k - 1];
            if (orc_seed(in) == 0) {
                pos += k;
                have = 0;
            } else {
                roll(&fh, &rh, k, (uint8_t)seq[pos - 1], in);
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Timed form of ntRead for plain k-mers (the cpu_baseline leg of bench.py): the same walk as one_read with
 * what the reference precomputes precomputed here as well — per byte value the seed, the complement seed
 * and their srol^k images for this k (nthash.hpp:66-183 are such tables; ours are derived with orc_srol,
 * not copied) — and the +-1 split rotates written out as a plain 64-bit rotate plus the swap of the two
 * bits that crossed the 33/31 boundary (nthash.hpp:186-217).  Checked against one_read by the tests.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t s[256], sk[256], c[256], ck[256]; /* seed, srol^k(seed), complement seed, srol^k(complement seed) */
} orc_ktab;

static void build_ktab(orc_ktab *t, unsigned k)
{
    for (unsigned b = 0; b < 256; ++b) {
        t->s[b] = orc_seed((uint8_t)b);
        t->c[b] = t->s[b] ? orc_seed_comp((uint8_t)b) : 0;
        t->sk[b] = orc_srol(t->s[b], k);
        t->ck[b] = orc_srol(t->c[b], k);
    }
}

static inline uint64_t srol1_fast(uint64_t x)
{
    uint64_t y = (x << 1) | (x >> 63);          /* bit 63 -> 0 and bit 32 -> 33: both on the wrong side */
    uint64_t d = ((y >> 33) ^ y) & 1u;
    return y ^ (d | (d << 33));
}

static inline uint64_t sror1_fast(uint64_t x)
{
    uint64_t y = (x >> 1) | (x << 63);          /* bit 0 -> 63 and bit 33 -> 32 */
    uint64_t d = ((y >> 32) ^ (y >> 63)) & 1u;
    return y ^ ((d << 32) | (d << 63));
}

static void one_read_fast(uint16_t *plane, const orc_ktab *t, const char *seq, size_t len, unsigned k,
                          uint32_t r_bits, uint32_t s_bits, uint64_t *f1_local)
{
    if (k == 0 || (size_t)k > len) return;
    const uint64_t r_buck = 1ULL << r_bits, r_mask = r_buck - 1, s_mask = (1ULL << (s_bits - 1)) - 1;
    const size_t last = len - k;
    const uint8_t *u = (const uint8_t *)seq;
    size_t pos = 0;
    uint64_t fh = 0, rh = 0, n = 0;
    int have = 0;
    while (pos <= last) {
        if (!have) { /* NTMC64 base, nthash.hpp:467-492 */
            int bad = -1;
            for (int i = (int)k - 1; i >= 0; --i)
                if (t->s[u[pos + i]] == 0) {
                    bad = i;
                    break;
                }
            if (bad >= 0) {
                pos += (size_t)bad + 1;
                continue;
            }
            fh = rh = 0;
            for (unsigned i = 0; i < k; ++i) {
                fh = srol1_fast(fh) ^ t->s[u[pos + i]];
                rh = srol1_fast(rh) ^ t->c[u[pos + k - 1 - i]];
            }
            have = 1;
        }
        const uint64_t h = rh < fh ? rh : fh;
        if ((h >> (63 - s_bits)) == 1) { /* ntComp, ntcard.cpp:132-145; the two patterns exclude each other for sBits >= 2 */
            __atomic_fetch_add(&plane[h & r_mask], (uint16_t)1, __ATOMIC_RELAXED);
        } else if ((h >> (64 - s_bits)) == s_mask) {
            __atomic_fetch_add(&plane[r_buck + (h & r_mask)], (uint16_t)1, __ATOMIC_RELAXED);
        }
        ++n;
        ++pos;
        if (pos > last) break;
        const uint8_t in = u[pos + k - 1], out = u[pos - 1];
        if (t->s[in] == 0) {
            pos += k;
            have = 0;
        } else { /* NTF64 / NTR64, nthash.hpp:242-257 */
            fh = srol1_fast(fh) ^ t->s[in] ^ t->sk[out];
            rh = sror1_fast(rh ^ t->ck[in] ^ t->c[out]);
        }
    }
    *f1_local += n;
}

/* one thread per shard of the batch (the reference: one thread per input file, ntcard.cpp:445-446) */
void orc_sketch_update_sharded(uint16_t *counters, const char *bases, const uint64_t *offsets,
                               uint64_t n_reads, const uint32_t *klist, uint32_t n_k,
                               uint32_t r_bits, uint32_t s_bits, uint64_t *f1, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    orc_ktab *tabs = (orc_ktab *)malloc(sizeof(orc_ktab) * n_k);
    for (uint32_t ki = 0; ki < n_k; ++ki) build_ktab(&tabs[ki], klist[ki]);
    const uint64_t r_buck = 1ULL << r_bits;
#pragma omp parallel for schedule(static, 1) num_threads(n_threads)
    for (int sh = 0; sh < n_threads; ++sh) {
        uint64_t local[64];
        memset(local, 0, sizeof local);
        const uint64_t lo = n_reads * (uint64_t)sh / (uint64_t)n_threads, hi = n_reads * (uint64_t)(sh + 1) / (uint64_t)n_threads;
        for (uint64_t i = lo; i < hi; ++i)
            for (uint32_t ki = 0; ki < n_k; ++ki) /* ntRead's loop over kList, ntcard.cpp:147-158 */
                one_read_fast(counters + (size_t)ki * 2 * r_buck, &tabs[ki], bases + offsets[i], (size_t)(offsets[i + 1] - offsets[i]),
                              klist[ki], r_bits, s_bits, &local[ki]);
        for (uint32_t ki = 0; ki < n_k; ++ki)
            __atomic_fetch_add(&f1[ki], local[ki], __ATOMIC_RELAXED);
    }
    free(tabs);
}

void orc_sketch_update(uint16_t *counters, const char *bases, const uint64_t *offsets,
                       uint64_t n_reads, const uint32_t *klist, uint32_t n_k, uint32_t gap,
                       uint32_t r_bits, uint32_t s_bits, uint64_t *f1, int n_threads)
{
    uint32_t gp[512];
    size_t ng = 0;
    if (gap) ng = orc_gap_positions(klist[0], gap, gp);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        uint64_t local[64];
        memset(local, 0, sizeof local);
#pragma omp for schedule(dynamic, 4096)
        for (uint64_t i = 0; i < n_reads; ++i) {
            one_read(counters, bases + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), klist,
                     n_k, gp, ng, r_bits, s_bits, local);
        }
        for (uint32_t ki = 0; ki < n_k; ++ki)
            __atomic_fetch_add(&f1[ki], local[ki], __ATOMIC_RELAXED);
    }
}

/* ---------------------------------------------------------------------------------------------
 * C6: compEst.  ntcard.cpp:240-247 (value histogram), :249-256 (mean of the two samples),
 * :258-259 (F0), :260-273 (f_i recurrence), :274 (|(ssize_t)(f_i * F0)|).
 * Built with -ffp-contract=off; operation order mirrors the formula in SURVEY.md App. B.
 * ------------------------------------------------------------------------------------------- */
void orc_value_hist(const uint16_t *counters_k, uint32_t r_bits, uint32_t *p)
{
    const size_t r_buck = (size_t)1 << r_bits;
    memset(p, 0, sizeof(uint32_t) * 2 * 65536);
    for (unsigned s = 0; s < 2; ++s)
        for (size_t j = 0; j < r_buck; ++j) ++p[s * 65536 + counters_k[s * r_buck + j]];
}

void orc_comp_est_p(const uint32_t *p, uint32_t r_bits, uint32_t s_bits, uint32_t limit,
                    double *F0, double *f_mean)
{
    double *pm = (double *)malloc(sizeof(double) * 65536);
    if (limit > 65535) limit = 65535;
    for (size_t i = 0; i < 65536; ++i) {
        double m = 0.0;
        for (size_t j = 0; j < 2; ++j) m += p[j * 65536 + i];
        m /= 1.0 * 2;
        pm[i] = m;
    }
    double f0 = (ssize_t)((r_bits * log(2) - log(pm[0])) * 1.0 * ((uint64_t)1 << (s_bits + r_bits)));
    *F0 = f0;
    for (size_t i = 0; i < 65536; ++i) f_mean[i] = 0;
    if (pm[0] * (log(pm[0]) - r_bits * log(2)) == 0) {
        free(pm);
        return;
    }
    f_mean[1] = -1.0 * pm[1] / (pm[0] * (log(pm[0]) - r_bits * log(2)));
    for (size_t i = 2; i <= limit; ++i) {
        double sum = 0.0;
        for (size_t j = 1; j < i; ++j) sum += j * pm[i - j] * f_mean[j];
        f_mean[i] = -1.0 * pm[i] / (pm[0] * (log(pm[0]) - r_bits * log(2))) - sum / (i * pm[0]);
    }
    for (size_t i = 1; i <= limit; ++i) f_mean[i] = (double)labs((long)(ssize_t)(f_mean[i] * f0));
    free(pm);
}

void orc_comp_est(const uint16_t *counters_k, uint32_t r_bits, uint32_t s_bits, uint32_t limit,
                  double *F0, double *f_mean)
{
    uint32_t *p = (uint32_t *)malloc(sizeof(uint32_t) * 2 * 65536);
    orc_value_hist(counters_k, r_bits, p);
    orc_comp_est_p(p, r_bits, s_bits, limit, F0, f_mean);
    free(p);
}

/* C7: ntcard.cpp:291-294 — "F1\t<n>\nF0\t<n>\n" then "<i>\t<n>\n" for i = 1..covMax */
size_t orc_format_hist(uint64_t f1, double F0, const double *f_mean, uint32_t cov_max,
                       char *buf, size_t cap)
{
    size_t n = 0;
    n += (size_t)snprintf(buf + n, n < cap ? cap - n : 0, "F1\t%llu\n", (unsigned long long)f1);
    n += (size_t)snprintf(buf + n, n < cap ? cap - n : 0, "F0\t%llu\n",
                          (unsigned long long)(uint64_t)F0);
    for (uint32_t i = 1; i <= cov_max; ++i)
        n += (size_t)snprintf(buf + n, n < cap ? cap - n : 0, "%u\t%llu\n", i,
                              (unsigned long long)(uint64_t)f_mean[i]);
    return n;
}

/* ---------------------------------------------------------------------------------------------
 * a15: nthll.  nthll.cpp:92-97 (bucket = low n_bits, value = clz of the remaining bits, keep max),
 * :99-105 (every clean window of the read), :221-243 (thread-private registers merged by max),
 * :247-254 (alpha * m^2 / sum 2^-M, alpha halved for canonical hashes; the reference always is).
 * ------------------------------------------------------------------------------------------- */
void orc_hll_update(uint8_t *regs, uint32_t n_bits, const char *bases, const uint64_t *offsets,
                    uint64_t n_reads, uint32_t k, int n_threads)
{
    const uint64_t n_buck = 1ULL << n_bits;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        uint8_t *mine = (uint8_t *)calloc(n_buck, 1);
        uint64_t *h = NULL;
        size_t cap = 0;
#pragma omp for schedule(dynamic, 1024)
        for (uint64_t i = 0; i < n_reads; ++i) {
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
            if (len + 1 > cap) {
                cap = 2 * (len + 1);
                h = (uint64_t *)realloc(h, cap * sizeof *h);
            }
            const size_t n = orc_hash_read(bases + offsets[i], len, k, h, NULL, cap);
            for (size_t j = 0; j < n; ++j) {
                const uint64_t rest = h[j] & ~(n_buck - 1);
                if (rest) {
                    const uint8_t run0 = (uint8_t)__builtin_clzll(rest);
                    uint8_t *slot = &mine[h[j] & (n_buck - 1)];
                    if (run0 > *slot) *slot = run0;
                }
            }
        }
#pragma omp critical(orc_hll_merge)
        for (uint64_t j = 0; j < n_buck; ++j)
            if (regs[j] < mine[j]) regs[j] = mine[j];
        free(mine);
        free(h);
    }
}

double orc_hll_estimate(const uint8_t *regs, uint32_t n_bits)
{
    const unsigned n_buck = 1u << n_bits;
    double alpha = 1.4426 / (1 + 1.079 / n_buck);
    alpha /= 2;
    double p_est = 0.0;
    for (unsigned j = 0; j < n_buck; ++j) p_est += 1.0 / ((uint64_t)1 << regs[j]);
    const double z_est = 1.0 / p_est;
    return alpha * n_buck * n_buck * z_est;
}

/* ---------------------------------------------------------------------------------------------
 * Synthetic workload generator (own spec, DESIGN.md "Synthetic workloads"): counter-based, so a
 * read depends only on (seed, read index) and the CPU and the GPU generator (K0) agree bit for bit.
 * ------------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline unsigned genome_code(uint64_t gseed, uint64_t g)
{
    uint64_t h = mix64(gseed + (g >> 5));
    return (unsigned)(h >> (2 * (g & 31))) & 3u;
}

void orc_gen_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                   uint32_t stride, uint32_t dist, uint64_t genome_len, uint8_t *out)
{
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    const uint64_t rseed = mix64(seed);
    const uint64_t gseed = mix64(seed ^ 0x47454E4F4D45ULL);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n_reads; ++i) {
        const uint64_t r = first_read + i;
        uint8_t *dst = out + i * (uint64_t)stride;
        const uint64_t hr = mix64(rseed + r);
        if (dist == 0) {
            uint64_t h = 0;
            for (uint32_t j = 0; j < read_len; ++j) {
                if ((j & 31) == 0) h = mix64(hr + (j >> 5));
                dst[j] = (uint8_t)ACGT[(h >> (2 * (j & 31))) & 3u];
            }
        } else {
            const uint64_t span = genome_len - read_len + 1;
            const uint64_t pos = (uint64_t)(((__uint128_t)hr * span) >> 64);
            const uint64_t hs = mix64(hr ^ 0xA5A5A5A5A5A5A5A5ULL);
            const unsigned rev = (unsigned)(hs & 1u);
            uint64_t hm = 0;
            for (uint32_t j = 0; j < read_len; ++j) {
                unsigned code = rev ? 3u - genome_code(gseed, pos + read_len - 1 - j)
                                    : genome_code(gseed, pos + j);
                if ((j & 3) == 0) hm = mix64(hs + 1 + (j >> 2));
                unsigned u = (unsigned)(hm >> (16 * (j & 3))) & 0xFFFFu;
                uint8_t c;
                if (u < 655u)
                    c = (uint8_t)ACGT[(code + 1u + (u % 3u)) & 3u];
                else if (u < 688u)
                    c = 'N';
                else
                    c = (uint8_t)ACGT[code];
                dst[j] = c;
            }
        }
        for (uint32_t j = read_len; j < stride; ++j) dst[j] = 'A'; /* padding: never hashed */
    }
}

uint64_t orc_fnv1a64(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}
