/*
 * ntcard_hip.h — C ABI of the MI355X (gfx950) ntHash -> sample -> count engine.
 *
 * Drop-in boundary (seam B2).  The reference has no FFI; the seam this library
 * replaces is the in-process call the record parsers make for every sequence,
 *     ntRead / stRead(const std::string& seq, const std::vector<unsigned>& kList,
 *                     uint16_t* t_Counter, size_t totKmer[])      ntcard.cpp:147-171
 * (call sites ntcard.cpp:182,185,203,205,230,232), together with the state that call mutates,
 *     uint16_t t_Counter[nK][nSamp=2][1<<rBits]                  ntcard.cpp:437-439
 *     size_t   totalKmers[nK]  (F1)                              ntcard.cpp:433-435,464-466
 * its configuration globals opt::{rBits,sBits,sMask,rBuck,nSamp,gap,seedSet} (ntcard.cpp:52-67),
 * and the consumer of that state, compEst (ntcard.cpp:237-275) + outDefault/outCompact
 * (ntcard.cpp:277-315).  INTEGRATION.md shows the patch a reference maintainer would apply.
 *
 * Conventions: plain C, plain pointers and sizes, no torch/HIP types in signatures (a HIP stream is
 * passed as void*).  Every function returns 0 on success and a negative ntc_status on failure;
 * ntc_last_error() returns a thread-local message.  There is NO CPU fallback: if no gfx950 device
 * or kernel image is available, ntc_create fails with NTC_ERR_DEVICE.
 */
#ifndef NTCARD_HIP_H
#define NTCARD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTC_ABI_VERSION 6
#define NTC_MAX_K_LIST 32

typedef enum {
    NTC_OK = 0,
    NTC_ERR_ARG = -1,      /* bad argument / unsupported configuration              */
    NTC_ERR_DEVICE = -2,   /* no usable HIP device, or a HIP runtime call failed    */
    NTC_ERR_MEMORY = -3,   /* host or device allocation failed                      */
    NTC_ERR_STATE = -4     /* call not valid in the engine's current state          */
} ntc_status;

typedef struct ntc_engine ntc_engine; /* opaque: owns the device sketch, F1 and staging buffers */

/* Replaces opt::{kList(-k), gap(-g), rBits(-r), sBits(-s)} (ntcard.cpp:52-67,325-363).  The CALLER
 * applies the "total input < 50 GB => sBits = 7" rule (ntcard.cpp:427-431) before ntc_create.   */
typedef struct {
    uint32_t n_k;              /* number of k values (order defines sketch plane order)          */
    const uint32_t *k;         /* k list; every k >= 1 and <= NTC_MAX_K (see ntc_max_k())        */
    uint32_t gap;              /* 0, or g: seed = 1^((k-g)/2) 0^g 1^((k-g)/2); needs n_k == 1    */
    uint32_t r_bits;           /* log2 buckets per sample (reference default 27); 8..30          */
    uint32_t s_bits;           /* sampling bits (reference: 7 or 11); 2..24                      */
    int32_t device;            /* HIP device ordinal                                             */
    void *stream;              /* hipStream_t to run on, or NULL for the device's null stream    */
    void *ext_sketch;          /* optional caller-owned DEVICE memory for the sketch:
                                  uint32_t [n_k][2][1<<r_bits]; NULL -> engine allocates.
                                  (lets a host framework own/merge the buffer, e.g. RCCL reduce)
                                  Its CONTENTS are the engine's from ntc_create / ntc_reset (which zero
                                  it) on: the caller may read it behind ntc_flush / ntc_finish, and may
                                  WRITE to it (an in-place reduce, a merge) only after a call of
                                  ntc_device_state — which is what tells the engine that the counters
                                  can change behind its back (the first sketch update after a reset
                                  otherwise writes its counts into the zeroed buffer without reading) */
    void *ext_f1;              /* optional caller-owned DEVICE uint64_t [n_k]; NULL -> engine    */
    uint32_t flags;            /* NTC_FLAG_*                                                      */
    uint64_t log_entries;      /* capacity of the hit log in 4-byte entries, 0 = default (four per
                                  counter, at most 2^30: 4 GiB at rBits = 27 and one k).  ntComp's `++t_Counter[...]` (ntcard.cpp:142-143) is
                                  deferred: the kernels log the counter index of every sampled k-mer
                                  and the log is applied to the sketch when it fills up and whenever
                                  the counters are needed (ntc_finish, ntc_flush, ...)              */
} ntc_config;

#define NTC_FLAG_NONE 0u
#define NTC_FLAG_SIMPLE_KERNEL 1u  /* run the simple validation kernel instead of the production ones */
#define NTC_FLAG_LANE_KERNEL 32u    /* the lane-per-read kernel K1 takes every batch, tiled ones too (re-laid out as row slots): cross-check, A/B runs.
                                      (Flag value 4, round 2-4's NTC_FLAG_BITSLICE_KERNEL, is unused since ABI 5: the bit-sliced row-slot kernel K1b is gone.) */
#define NTC_FLAG_ALWAYS_LOG 8u      /* keep logging whatever the data looks like (by default the engine switches to direct
                                      atomics when an apply finds few distinct counters per increment: repeats)       */
#define NTC_FLAG_PARTITION_ALWAYS 16u /* validation: apply even a small hit log through the partition + histogram passes
                                         (by default fewer than 4 M pending entries are applied with plain atomics)   */
#define NTC_FLAG_DEFER_REDO 128u    /* device-resident batches (ntc_submit_device, ntc_submit_tiled_device): the caller promises to keep every
                                      submitted buffer valid AND UNCHANGED until ntc_sync / ntc_finish returns.  The second passes over a batch's
                                      reads with non-ACGTU bytes are then shared by several batches instead of run per batch: the fix-up
                                      kernels K1f behind up to 8 launches of the one-wave-per-tile kernel K1h.  Since ABI 6 the HASH launches wait as
                                      well: up to 8 tiled batches (ntc_submit_tiled_device / _ragged_device) are hashed by one launch per k, started
                                      when the eighth arrives or when ntc_flush / ntc_sync / ntc_finish / ntc_reset / a merge / ntc_device_state needs
                                      the counters — work submitted under this flag is only guaranteed to have been ENQUEUED after one of those calls
                                      (a small batch then costs its share of a large launch instead of a launch of its own).  Without the flag a
                                      buffer may be reused as soon as the stream has passed the submit call. */
#define NTC_FLAG_REQUIRE_TILED 64u  /* validation: a tiled batch must be served by the tiled kernels for EVERY k of the list, else ntc_submit_tiled_device fails */
#define NTC_FLAG_DIRECT_ATOMICS 2u /* no hit log: every sampled k-mer is one device atomic on the sketch
                                      (the literal form of ntcard.cpp:142-143; cross-check and A/B runs) */

uint32_t ntc_abi_version(void);
uint32_t ntc_max_k(void);
const char *ntc_last_error(void);

/* ntcard.cpp:437-439 (allocate + zero t_Counter), :433-435 (zero F1) */
int ntc_create(const ntc_config *cfg, ntc_engine **out);
void ntc_destroy(ntc_engine *e);                 /* ntcard.cpp:474 */
int ntc_reset(ntc_engine *e);                    /* re-zero sketch and F1 */

/* ntRead/stRead for a batch of sequences (ntcard.cpp:147-171).  HOST buffers:
 * bases = concatenated raw sequence bytes exactly as the parsers produced them (any case, any
 * IUPAC/N byte), offsets[n_reads+1] delimits read i = bases[offsets[i], offsets[i+1]).
 * The engine copies what it needs before returning (caller keeps ownership): reads are packed
 * into one of a few pinned staging buffers, then copy + kernels are queued on the engine's
 * stream; device-side errors surface at ntc_sync/ntc_finish.  Thread-safe: may be called
 * concurrently from several parser threads like the reference's seam (packing runs in parallel,
 * only the enqueue is serialised).                                                             */
int ntc_submit(ntc_engine *e, const char *bases, const uint64_t *offsets, uint64_t n_reads);

/* The same for reads that are SPANS of one host buffer: read i = buf[starts[i], starts[i] + lens[i]).  This is what a
 * block-based record splitter produces (the sequence lines of a FASTQ block, field 10 of SAM lines): the bytes are
 * copied exactly once, from the file block into the pinned staging buffer.  Same threading and ownership rules.  */
int ntc_submit_spans(ntc_engine *e, const char *buf, const uint64_t *starts, const uint32_t *lens, uint64_t n_reads);

/* Same for a batch that is already DEVICE-resident in the engine's slot layout: read i occupies
 * d_slots[i*stride, i*stride+read_len), stride % 4 == 0, d_slots 16-byte aligned.  Padding bytes
 * (read_len..stride) are never hashed; filling them with a base letter ('A') keeps the kernel on
 * its fast path (a non-ACGTU byte anywhere in a wave's 64 slots selects the dirty-window path).
 * Asynchronous on the engine's stream: the buffer may be reused as soon as the stream has passed the call (stream-ordered,
 * like a kernel launch) — unless the engine was created with NTC_FLAG_DEFER_REDO, see there.
 * A wave parks its 64 slots in LDS, so stride is limited to about 2.4 KB (64 * stride + tables <= 160 KiB);
 * longer sequences go through ntc_submit, which splits them into overlapping chunks.            */
int ntc_submit_device(ntc_engine *e, const void *d_slots, uint64_t n_reads, uint32_t read_len,
                      uint32_t stride);

/* The same for a DEVICE-resident batch in the engine's TILED slot layout — the layout the hot kernel pair (K1h + K1f) streams:
 *     tile t   = reads [2048 t, 2048 t + 2048) of the batch,       n_chunks = ceil(read_len / 16)
 *     piece    = the 16 raw bytes of bases [16 c, 16 c + 16) of read i, at byte offset
 *                ((i / 2048 * n_chunks + c) * 2048 + i % 2048) * 16          (ntc_tiled_bytes() bytes in all)
 * i.e. the pieces of one base range of a tile's 2048 reads are contiguous (32 KiB): one coalesced load hands every lane
 * the next 16 bases of "its" read, in the position-major order the bit-sliced hash walk consumes.  Bytes are raw sequence
 * bytes as the parsers produce them (any case, N / IUPAC); bytes behind a read's end (inside its last piece) must hold a base
 * letter ('A'); the slots behind the batch's last read (the rest of the last tile) are ignored whatever they hold, so any
 * prefix of a tiled buffer is a valid batch.  All reads of a batch have the same length.  Asynchronous on the engine's
 * stream; the buffer may be reused as soon as the stream has passed the call — unless the engine was created with
 * NTC_FLAG_DEFER_REDO, see there.  A list of k is served by one launch per k.  The tiled kernels are built for k = 12 .. 32
 * (spaced seeds: ntcard's -g seed at k = 12 / gap 2 and k = 32 / gap 8) and sBits >= 7; a list of which only a part lies in 12 .. 32
 * is served by both kernels from the same tiles (the general kernel stages the tiles for its k: no re-layout); a configuration in
 * which NO k is theirs (every k outside 12 .. 32, other spaced seeds, nthll, sBits < 7) is re-laid out on the device and takes the
 * general kernel: same results, not the fast path.  Host batches reach the same kernels: ntc_submit / ntc_submit_spans pack them
 * into tiles in pinned staging (reads of one length as one batch, mixed lengths as the length bins of ntc_submit_tiled_bins_device). */
int ntc_submit_tiled_device(ntc_engine *e, const void *d_tiles, uint64_t n_reads, uint32_t read_len);
uint64_t ntc_tiled_bytes(uint64_t n_reads, uint32_t read_len);

/* The same for reads of UNEQUAL length (ntRead takes any sequence, ntcard.cpp:173-189; adapter-trimmed FASTQ) — ABI 5.  A ragged tiled batch holds
 * reads of 16 * n_chunks - 15 .. 16 * n_chunks bases (one bin of ceil(len / 16)) in the tiled layout of ntc_submit_tiled_device with
 * read_len = 16 * n_chunks; bytes behind a read's end hold a base letter ('A').  EVERY TILE IS SORTED LONGEST READ FIRST (counting is
 * order-independent, so a producer may reorder), and d_tails[tile * 16 + d], d = 0 .. 15, is the number of reads of that tile with more than d
 * bases in their last 16-base piece (d_tails[tile * 16] = the reads of the tile; non-increasing in d).  The kernel masks every window that ends
 * behind a read's end with the prefix of the tile that is long enough: no per-read length array, 64 bytes per tile.  d_tails follows the rules of
 * d_tiles (device memory, valid until the stream has passed the call — until ntc_sync with NTC_FLAG_DEFER_REDO).  Fails with NTC_ERR_ARG on an
 * engine in whose configuration NO k is the tiled kernels' (ABI 6: a list of which only a part is theirs is fine — the general kernel stages the same
 * tiles for its k and takes every read's length from a table the engine derives from d_tails).  ntc_submit / ntc_submit_spans build such batches themselves: a host batch of
 * mixed lengths is binned by ceil(len / 16), bins of >= 1024 reads go this way, the rest takes row slots.                                      */
int ntc_submit_tiled_ragged_device(ntc_engine *e, const void *d_tiles, uint64_t n_reads, uint32_t n_chunks, const uint32_t *d_tails);

/* Several tiled batches of different geometry in ONE call — the length bins of a ragged read set (ABI 5).  Bin i is d_tiles[i] with n_reads[i] reads of
 * read_len[i] bases; d_tails[i] == NULL (or d_tails == NULL: all of them): an equal-length batch as for ntc_submit_tiled_device, otherwise a ragged one as for
 * ntc_submit_tiled_ragged_device (read_len[i] = 16 * n_chunks).  Counts exactly what n_bins separate calls count; the difference is speed: the
 * hash kernel takes up to 8 bins per launch and shares its workgroups among them in proportion to their work, where separate calls are separate,
 * small launches (four 2.5 M-read bins: 0.55 ms one by one, 0.39 ms together).  Empty bins are skipped.  NTC_ERR_ARG as for the single calls.    */
int ntc_submit_tiled_bins_device(ntc_engine *e, uint32_t n_bins, const void *const *d_tiles, const uint64_t *n_reads, const uint32_t *read_len,
                                 const uint32_t *const *d_tails);

int ntc_sync(ntc_engine *e); /* wait for all submitted work */

/* Apply the pending hit log to the device sketch (asynchronous on the engine's stream).  After it the
 * device counters are the reference's t_Counter state for everything submitted so far (ntcard.cpp:142-143). */
int ntc_flush(ntc_engine *e);

/* End of stream: the state compEst/outDefault consume.
 * t_counter_out: HOST uint16_t [n_k][2][1<<r_bits] (== the reference's t_Counter) or NULL
 * p_hist_out:    HOST uint32_t [n_k][2][65536], p[s][v] = #buckets of sample s whose counter == v
 *                (ntcard.cpp:240-247) or NULL
 * f1_out:        HOST uint64_t [n_k] (totalKmers, ntcard.cpp:464-466) or NULL
 * May be called repeatedly; does not clear the sketch.                                          */
int ntc_finish(ntc_engine *e, uint16_t *t_counter_out, uint32_t *p_hist_out, uint64_t *f1_out);

/* compEst's first loop (ntcard.cpp:240-247) over an arbitrary run of DEVICE counters: for each of the n uint32
 * counters, ++d_hist_u32[counter & 0xffff] (the histogram is accumulated, not zeroed).  Asynchronous on `stream`.
 * Used by the multi-GPU merge: after a reduce-scatter every rank histograms its own slice of the summed sketch and
 * only the 256 KiB histograms travel to rank 0 (ntcard_amd/parallel.py).                                      */
int ntc_value_hist_device(int32_t device, void *stream, const void *d_counters_u32, uint64_t n, void *d_hist_u32);

/* The device steps of the multi-GPU merge for a caller that moves the counter slices itself — one process per GPU, the slices travel in one RCCL
 * all-to-all (ntcard_amd/parallel.py) — the same kernels ntc_merge_devices runs between its peer copies.  t_Counter wraps at 16 bits
 * (ntcard.cpp:142-143,439), so only the low halves travel:
 *   ntc_narrow_u16_device      d_out_u16[i] = d_counters_u32[i] mod 2^16, i < n
 *   ntc_sum_slices_u16_device  slice 0 += slices 1 .. n_slices - 1 (wrapping 16-bit adds); slice r = d_slices_u16[r * stride, r * stride + len)
 *   ntc_value_hist_u16_device  ++d_hist_u32[d_counters_u16[i]], i < n (compEst's first loop, ntcard.cpp:240-247, over a summed slice)
 * All asynchronous on `stream`; buffers 16-byte aligned, stride a multiple of 8 elements.                                              */
int ntc_narrow_u16_device(int32_t device, void *stream, const void *d_counters_u32, uint64_t n, void *d_out_u16);
int ntc_sum_slices_u16_device(int32_t device, void *stream, void *d_slices_u16, uint64_t stride, uint32_t n_slices, uint64_t len);
int ntc_value_hist_u16_device(int32_t device, void *stream, const void *d_counters_u16, uint64_t n, void *d_hist_u32);

/* The multi-GPU merge that ships HITS instead of counters (ABI 6) — for runs whose sampled k-mers are fewer than their counters' bytes, i.e. the reference's
 * sBits = 11 branch (inputs >= 50 GB, ntcard.cpp:427-431; BASELINE config 3: 125 M reads per GPU log 14.5 M keys = 58 MB where the 16-bit slice exchange
 * moves 448 MiB per rank).  Counter range p of n_parts ("owner" p) = the counters [p * C / n_parts, (p + 1) * C / n_parts), C = n_k * 2 * 2^r_bits.
 *   ntc_log_export_device   the PENDING hit log (everything counted since the last sketch update: the operands of ntComp's `++t_Counter[...]`,
 *                           ntcard.cpp:142-143) split by owner.  counts_out[p] (host) = the keys of owner p; with d_keys_u32 != NULL they are written, 32-bit
 *                           counter indices in arbitrary order, to d_keys_u32[part_offset[p] .. part_offset[p] + counts_out[p]) (part_offset: host, n_parts
 *                           entries; call once with d_keys_u32 == NULL to learn the counts).  Synchronises the engine's stream.  The log stays pending.
 *                           NTC_ERR_STATE when the sketch already holds counts (an update ran since the last reset, or a kernel incremented it directly) or
 *                           the engine has no hit log: the caller then merges counters (ntc_narrow_u16_device ...), which is always possible.
 *   ntc_log_replace_device  drops the pending log and makes the n_keys counter indices at d_keys_u32 (device) the pending log instead: the next
 *                           ntc_flush / ntc_finish counts exactly them.  An owner calls it with the keys it received for its range (its own included):
 *                           its sketch then holds the SUMMED counters of its range (and zeros elsewhere), to be histogrammed with ntc_value_hist_device.
 * ntcard_amd/parallel.py (merge_owner) runs the exchange between the two calls with one RCCL all-to-all.                                              */
int ntc_log_export_device(ntc_engine *e, uint32_t n_parts, void *d_keys_u32, const uint64_t *part_offset, uint64_t *counts_out);
int ntc_log_replace_device(ntc_engine *e, const void *d_keys_u32, uint64_t n_keys);

/* Sketch load / merge: adds a t_Counter image dumped by ntc_finish (same k list, r_bits;
 * uint16 [n_k][2][1<<r_bits]) and its F1 values (may be NULL) into this engine.  Counting is a commutative sum
 * mod 2^16 (ntcard.cpp:142-143), so runs split across processes, nodes or days merge exactly.               */
int ntc_merge_counters(ntc_engine *e, const uint16_t *t_counter, const uint64_t *f1);

/* Multi-GPU merge inside ONE host process: engines[0..n) are engines with the same configuration,
 * normally one per GPU of the node, each fed with its own share of the reads.  After the call engine 0 holds the
 * element-wise SUM of all sketches and F1 values (MAX of the registers for nthll engines) — the state the
 * reference's threads build in their one shared t_Counter / totalKmers (ntcard.cpp:142-143,464-466; nthll.cpp:
 * 238-243) — and the other engines are reset to zero.  t_Counter wraps at 16 bits (ntcard.cpp:439), so the counters travel
 * as their low halves: every engine narrows its sketch to uint16, slice j of every engine is copied to engine j's device (all
 * N x (N-1) peer copies in flight together, one per point-to-point xGMI link; hipMemcpyPeerAsync, staged by the runtime when two
 * devices have no peer access), engine j adds its N slices with wrapping 16-bit adds, the summed slices are gathered on engine
 * 0's device and widened into its sketch — the same exchange bench.py runs between processes with RCCL's all-to-all.  Engine 0's
 * counters afterwards hold their value mod 2^16 (all t_Counter ever held) and it may keep counting; F1 and nthll registers are
 * merged at full width.  No communicator, no library beyond HIP.                                                        */
int ntc_merge_devices(ntc_engine *const *engines, int32_t n_engines);

/* Device pointers of the live sketch / F1 (for a host framework's collective); flushes the hit log first */
int ntc_device_state(ntc_engine *e, void **d_sketch_u32, uint64_t *n_counters, void **d_f1_u64);

/* Validation kernel (K1d): canonical hash of every window of ONE k for a device-resident slot
 * batch.  d_hash_out: DEVICE uint64_t [n_reads][max_win], d_count_out: DEVICE uint32_t [n_reads].
 * The hashes of read i are written compacted, in window order, and d_count_out[i] says how many
 * there are (== that read's F1 share; may exceed max_win, then only max_win are stored), mirroring ntHashIterator
 * (ntHashIterator.hpp:59-86) / stHashIterator when gap != 0.                                     */
int ntc_hash_dump_device(int32_t device, void *stream, const void *d_slots, uint64_t n_reads,
                         uint32_t read_len, uint32_t stride, uint32_t k, uint32_t gap,
                         uint32_t max_win, void *d_hash_out, void *d_count_out);
/* The same dump produced by a validation build of the PRODUCTION kernel K1 (its filter lets every window through, so
 * every 64-bit value comes out of K1's closed-form resolve stage; spaced seeds included: ntcard.cpp:160-171,
 * stHashIterator.hpp:60-87, nthash.hpp:641-646).  ntc_hash_dump_device forwards here when gap != 0.  Synchronous. */
int ntc_hash_dump_k1_device(int32_t device, void *stream, const void *d_slots, uint64_t n_reads,
                            uint32_t read_len, uint32_t stride, uint32_t k, uint32_t gap,
                            uint32_t max_win, void *d_hash_out, void *d_count_out);

/* Synthetic workload generator (K0), bit-identical to oracle/orc_gen_reads; DESIGN.md
 * "Synthetic workloads".  Fills d_slots[n_reads*stride].                                         */
int ntc_gen_reads_device(int32_t device, void *stream, void *d_slots, uint64_t seed,
                         uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint32_t stride,
                         uint32_t dist, uint64_t genome_len);

/* the same reads (bit-identical bases) in the tiled layout of ntc_submit_tiled_device; fills ntc_tiled_bytes() bytes */
int ntc_gen_reads_tiled_device(int32_t device, void *stream, void *d_tiles, uint64_t seed, uint64_t first_read,
                               uint64_t n_reads, uint32_t read_len, uint32_t dist, uint64_t genome_len);

/* compEst (ntcard.cpp:249-274) from the value histogram of ONE k.  f_out has cov_max+1 doubles
 * (f_out[0] unused); only i <= cov_max is evaluated (identical values).          */
int ntc_estimate(const uint32_t *p_hist /* [2][65536] */, uint32_t r_bits, uint32_t s_bits,
                 uint32_t cov_max, double *F0_out, double *f_out);

/* outDefault body for one k (ntcard.cpp:283,291-294): writes "<prefix>_k<k>.hist" when path is
 * given verbatim.  Returns 0 or NTC_ERR_ARG if the file cannot be written.                      */
int ntc_write_hist(const char *path, uint64_t f1, double F0, const double *f, uint32_t cov_max);

/* ---- nthll: HyperLogLog-style F0 of the same canonical ntHash stream --------------
 * Replaces nthll.cpp's per-thread `uint8_t mVec[nBuck]`, its ntRead/ntComp (nthll.cpp:92-105: bucket = low
 * n_bits of the hash, value = leading zeros of the remaining bits, keep the max), the max-merge under
 * `omp critical` (nthll.cpp:238-243) and the estimate (nthll.cpp:247-254).  Reads are fed with
 * ntc_submit / ntc_submit_device exactly as for an ntcard engine.                                          */
int ntc_hll_create(uint32_t k, uint32_t n_bits /* nthll -b, default 16 */, int32_t device, void *stream,
                   ntc_engine **out);
/* regs_out: HOST uint8_t [1<<n_bits] (== tVec of nthll.cpp:212-243); f1_out: number of k-mers hashed, or NULL */
int ntc_hll_finish(ntc_engine *e, uint8_t *regs_out, uint64_t *f1_out);
int ntc_hll_estimate(const uint8_t *regs, uint32_t n_bits, double *est_out);

/* Timing of the hot kernel as measured with HIP events on the engine's stream (for bench.py's
 * roofline leg): accumulated milliseconds and launch count since create/reset.                  */
int ntc_kernel_time(ntc_engine *e, double *ms_total, uint64_t *launches);
/* same for the deferred sketch update (partition + count passes): milliseconds and number of applies */
int ntc_apply_time(ntc_engine *e, double *ms_total, uint64_t *applies);
/* same for the fix-up kernels K1f of the one-wave-per-tile kernel when the engine defers them (NTC_FLAG_DEFER_REDO: one K1f launch over up to 8
 * batches, outside the hash kernels' events, so this time is NOT part of ntc_kernel_time; without the flag K1f follows every K1h launch and is) */
int ntc_fixup_time(ntc_engine *e, double *ms_total);
/* device buffers, copy streams and events ntc_merge_devices has created for this engine so far: they are kept between merges, so the count
 * stops growing after the first merge of a given group of engines (diagnostic) */
int ntc_merge_allocations(ntc_engine *e, uint64_t *n);
int ntc_set_profiling(ntc_engine *e, int enable);
/* how ntComp's increment is currently carried out on the device: 0 = hit log + partitioned apply, 1 = direct atomics
 * (NTC_FLAG_DIRECT_ATOMICS, or chosen by the engine after an apply found mostly repeated counters); waits for the stream */
int ntc_update_mode(ntc_engine *e, uint32_t *mode_out);

#ifdef __cplusplus
}
#endif
#endif
