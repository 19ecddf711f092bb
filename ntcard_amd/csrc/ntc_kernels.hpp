// ntc_kernels.hpp — argument blocks and launch helpers shared by ntc_kernels.hip / ntc_engine.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nthash_tables.hpp"

namespace ntc {

constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / 64;
constexpr int kTableBytes = kSlots * 32; // A (20 x 16 B) + B (20 x 16 B)

struct HashArgs {
	const unsigned char* slots; // [n_slots][stride] raw read bytes, slot i at i*stride
	const uint32_t* meta;       // optional [n_slots]: len | (window-start limit << 16); NULL: uniform
	uint64_t n_slots;
	uint32_t stride;            // bytes per slot, multiple of 4
	uint32_t read_len;          // uniform read length when meta == NULL
	uint32_t k;
	uint32_t r_bits, s_bits;
	uint32_t max_win;           // dump mode: capacity per read
	uint32_t* sketch;           // MODE 0: uint32 [2][1<<r_bits] plane pair of this k
	unsigned long long* f1;     // MODE 0: F1 of this k
	uint64_t* dump;             // MODE 1: [n_slots][max_win]
	uint32_t* dump_count;       // MODE 1: [n_slots]
	const void* t1;             // H-filter kernel: [ceil(k/2)][16] x {fwd.lo, fwd.hi, rev.lo, rev.hi} pre-rotated seed pairs (device)
	const void* gapt;           // H-filter kernel, spaced seed: [ceil(gap/2)][16] x {f.Hd, r.Hd, 0, 0} terms to XOR out
	uint32_t gap, gap_first;    // number of don't-care positions and index of the first one (ntcard.cpp:407-413)
	uint32_t hll_bits;          // != 0: nthll mode — `sketch` is uint32 M[1<<hll_bits] (max leading-zero runs, nthll.cpp:92-97)
	const uint32_t* hll_thr;    // nthll mode: device word, only hashes whose top 32 bits are < *hll_thr can raise a register
	uint32_t init[6];           // fast kernel: strand registers of the k x 'A' window {flo,fB,fHd,rlo,rB,rHd}
	HashTables tab;
};

// K1 (sketch_hf_kernel): one launch hashes a resident batch for up to kMaxFusedK values of k (the reference's
// ntRead loops over its k list per read, ntcard.cpp:147-158): the batch is staged and decoded once.
constexpr int kMaxFusedK = 4;
struct HfK {
	uint32_t k;
	uint32_t init_f, init_r;    // H halves ((H << 1) | H[30]) of the hash of k x 'A', forward / reverse
	uint32_t pad_;
	uint32_t* sketch;           // uint32 [2][1<<r_bits] plane pair of this k (nthll: uint32 M[1<<hll_bits])
	unsigned long long* f1;     // F1 of this k
	const void* t1;             // [ceil(k/2)][16] x {fwd.lo, fwd.hi, rev.lo, rev.hi} pre-rotated seed pairs (device)
	uint32_t key_base;          // index of this k's first counter in the engine's sketch array (hit-log keys are global indices)
	uint32_t pad2_;
	uint32_t tabh[kMainSlots][2]; // per (in,out) base pair: {Tf.Hd, Tr.Hd} step terms of the H halves
};
struct HfArgs {
	const unsigned char* slots; // as HashArgs
	const uint32_t* meta;
	uint64_t n_slots;
	uint32_t stride, read_len;
	uint32_t r_bits, s_bits;
	uint32_t n_k;               // 1..kMaxFusedK
	uint32_t gap, gap_first;    // spaced seed: single k only
	uint32_t hll_bits;          // nthll mode: single k only
	uint32_t log_regions, log_region_cap; // hit log geometry (0 regions: ntComp's increment is a direct device atomic)
	uint32_t* log;              // [log_regions][log_region_cap] counter indices of sampled k-mers, relative to sketch0
	uint32_t* log_fill;         // [log_regions] entries used per region (persists across launches until the log is applied)
	uint32_t* sketch0;          // the engine's whole sketch array (overflow fallback of the log)
	const uint32_t* log_mode;   // device word: 0 = append to the log, 1 = direct atomics (set by the apply pass, see ntc_apply.hip)
	uint64_t* dump;             // validation build of K1 (kDump): [n_slots][dump_win] canonical hash of the window starting at each position
	uint32_t* dump_valid;       //   [n_slots][ceil(dump_win / 32)] bit set where a hash was written (window without a non-ACGTU byte)
	uint32_t dump_win, pad3_;
	const uint64_t* gather;     // != NULL: the batch is the listed slots, gather[i] = ADDRESS of slot i's bytes (they may lie in different buffers), i < *gather_count
	const uint32_t* gather_count; // (reads the bit-sliced kernel K1b handed back: a non-ACGTU byte somewhere in the read)
	const void* gapt;
	const uint32_t* hll_thr;
	uint32_t tabg[kMainSlots][2]; // spaced seed, rolling form: per (leaving, entering) base pair of the don't-care block
	HfK ks[kMaxFusedK];
};

// K1b (sketch_bs_kernel, ntc_sketch_bs.hip): bit-sliced filter walk over whole tiles of 2048 equal-length slots
struct BsArgs {
	const unsigned char* slots; // tile t = slots [2048 t, 2048 (t + 1))
	uint64_t n_tiles;
	uint32_t stride, read_len;
	uint32_t k, r_bits, s_bits;
	uint32_t nq;                // 16-window blocks per wave segment: 4 waves x 16 nq windows >= read_len - k + 1
	uint32_t key_base;
	uint32_t log_regions, log_region_cap;
	uint32_t* log;
	uint32_t* log_fill;
	uint32_t* sketch0;
	const uint32_t* log_mode;
	unsigned long long* f1;
	const void* t4;             // [k/4][256] x {fwd.lo, fwd.hi, rev.lo, rev.hi}: closed form, 4 bases per entry (code2 order)
	uint64_t* redo_list;        // addresses of the slots left to the lane-per-read kernel (appended: the list outlives the launch)
	uint32_t* redo_count;
	uint64_t* dbg;              // instrumentation builds only (NTC_BS_TIMERS)
};
hipError_t launch_sketch_bs(const BsArgs& a, unsigned grid, hipStream_t st);
// append the addresses of slots [first, first + n) of a batch to a redo list (the tail behind K1b's whole tiles)
hipError_t launch_append_slots(uint64_t* list, uint32_t* count, const unsigned char* slots, uint32_t stride, uint64_t first, uint32_t n, hipStream_t st);
hipError_t set_sketch_bs_smem_limit(size_t smem);
size_t sketch_bs_smem(uint32_t k, uint32_t stride);
bool sketch_bs_supports(uint32_t k, uint32_t s_bits);

// K1c (sketch_ts_kernel, ntc_sketch_ts.hip): tiled streaming kernel.  TILED slot layout: tile t = reads [2048 t, 2048 t + 2048);
// the 16 raw bytes of bases [16 c, 16 c + 16) of read r of tile t sit at ((t * n_chunks + c) * 2048 + r) * 16.
constexpr uint32_t kTileReads = 2048;
struct TsArgs {
	const unsigned char* tiles;
	uint64_t n_reads;           // reads of the batch; slots behind the last one (in the last tile) hold 'A's and are ignored
	uint32_t n_tiles, n_chunks; // ceil(n_reads / 2048), ceil(read_len / 16)
	uint32_t read_len;
	uint32_t k, r_bits, s_bits;
	uint32_t key_base;
	uint32_t log_regions, log_region_cap;
	uint32_t* log;
	uint32_t* log_fill;
	uint32_t* sketch0;
	const uint32_t* log_mode;
	unsigned long long* f1;
	const void* t4;             // [k/4][256] x {fwd.lo, fwd.hi, rev.lo, rev.hi}: closed form, 4 bases per entry (code2 order)
	uint32_t* dbg;              // debugging builds only (tools/dbg)
};
hipError_t launch_sketch_ts(const TsArgs& a, unsigned grid, hipStream_t st);
hipError_t set_sketch_ts_smem_limit(size_t smem);
size_t sketch_ts_smem(uint32_t k);
bool sketch_ts_supports(uint32_t k, uint32_t s_bits);
// K1h + K1f (ntc_sketch_k1h.hip; kernel body generated by gen_k1h.py): one wave per tile.  The generated code reads the first 88 bytes
// with scalar loads at the offsets of gen_k1h.KARG — keep the two in step (static_asserts in ntc_sketch_k1h.hip).
struct K1hArgs {
	const unsigned char* tiles;   // tiled slots (as TsArgs)
	uint32_t* log;                // hit log, [log_regions][log_region_cap] (log_regions == 0: direct atomics on sketch0)
	uint32_t* log_fill;
	uint32_t* sketch0;
	unsigned long long* f1;
	uint32_t* dirty;              // [n_tiles][n_chunks][64]: bit m of word (t, c, lane) = the 16-byte piece of read 64 m + lane holds a non-ACGTU byte
	uint32_t* tie;                // [n_tiles][blocks][64]: bit m = some window of that block of that read has both strands flagged
	uint32_t n_tiles, n_chunks;
	uint32_t read_len, nv_last;   // nv_last: reads in the last tile (1 .. 2048)
	uint32_t key_base, rmask2;    // rmask2 = (2 << r_bits) - 1: the counter index and the sample bit of a table word
	uint32_t log_regions, log_region_cap;
	const uint32_t* table;        // build_k1h_table
	uint32_t s_bits, r_bits;
	uint32_t blocks_per_wave;     // ceil(n_tiles * blocks / waves of the launch): wave w walks that many consecutive blocks from w * blocks_per_wave
	uint32_t nb_magic;            // floor(2^32 / blocks)
	uint4* sus;                   // suspect list: [waves][sus_cap] x {counter index, tile, read | window << 11, 0}: candidates K1h resolved but whose
	                              // window lies near a dirty piece — K1f counts them iff their bytes are all bases
	uint32_t* sus_count;          // [waves]: entries written (0xffffffff: the wave ran out of room — K1f then walks every dirty-affected block itself)
	uint32_t sus_cap, launch_id;
	uint32_t* fix_state;          // K1f scratch: [0] = launch_id of the last launch that must take the slow path, [2..3] = F1 correction (uint64)
};
bool sketch_k1h_supports(uint32_t k, uint32_t gap, uint32_t s_bits, uint32_t r_bits);
uint32_t sketch_k1h_blocks(uint32_t k, uint32_t read_len);
uint32_t sketch_k1h_waves();
uint32_t sketch_k1h_min_blocks(); // blocks per wave below which launch_sketch_k1h uses fewer workgroups
void build_k1h_table(uint32_t k, uint32_t gap, uint32_t r_bits, uint32_t s_bits, uint32_t* out /* 2 * ceil(k / 3) * 64 dwords */);
hipError_t set_sketch_k1h_smem_limit();
hipError_t launch_sketch_k1h(const K1hArgs& a, uint32_t k, uint32_t gap, unsigned cus, hipStream_t st, K1hArgs* args_out, uint32_t* n_waves);
// K1f takes up to kK1fBatch K1h launches at a time (blockIdx.y = the launch): its kernels wait on memory, not on issue slots, so the launches of
// several batches cost little more than those of one (an engine whose caller keeps the batches unchanged until ntc_sync defers them)
constexpr uint32_t kK1fBatch = 8;
struct K1fItem {
	K1hArgs a;                    // as launched (launch_sketch_k1h's args_out)
	const void* t4;               // build_t4 of this k, with the engine's gap
	uint32_t k, n_waves;          // n_waves: K1h waves of the launch = suspect regions
};
struct K1fBatch {
	K1fItem item[kK1fBatch];
};
hipError_t launch_k1h_fixup(const K1fBatch& b, uint32_t n_items, unsigned cus, hipStream_t st);
// tiled layout -> row-major slots: device-side re-layout for the configurations the tiled kernels are not built for
hipError_t launch_gen_tiled(unsigned char* out, uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint32_t dist, uint64_t glen, hipStream_t st);
hipError_t launch_untile(const unsigned char* tiles, unsigned char* slots, uint64_t n_reads, uint32_t read_len, uint32_t stride, hipStream_t st);

// ---- deferred sketch update (ntc_apply.hip) ----
// A1/A2: radix partition of key runs.  Input run `seg` = in[seg * in_cap, +min(in_cnt[seg], in_cap)).
//   mode 0: workgroup w reads runs w, w + grid, ...                      (first pass over the raw log regions)
//   mode 1: workgroup w = b * parts + p reads runs (p + t * parts) * nb_in + b, t = 0, 1, ...   (bucket b of pass 1)
// Output run (w, digit) = out[(w * 2^bits + digit) * out_cap, +out_cnt[w * 2^bits + digit]).
struct SplitArgs {
	const uint32_t* in;
	const uint32_t* in_cnt;
	uint32_t in_cap, n_in;
	uint32_t mode, parts, nb_in;
	uint32_t shift, bits;
	uint32_t out_cap;
	uint32_t* out;
	uint32_t* out_cnt;
	uint32_t* sketch; // uint32 [2][1 << r_bits] of this k: overflow fallback
};
// A3: slice s = keys [s << slice_bits, (s + 1) << slice_bits).  mode 0: raw regions (one slice), 1: after one
// split pass (runs (w1, s), w1 < nwg1), 2: after two (runs ((b, p), d2), s = b * nb2 + d2, p < parts).
struct CountArgs {
	const uint32_t* in;
	const uint32_t* in_cnt;
	uint32_t in_cap, n_in;
	uint32_t mode, nb1, nwg1, parts, nb2;
	uint32_t slice_bits, n_slices;
	uint32_t* sketch;
};
hipError_t launch_log_atomics(const uint32_t* log, uint32_t* fill, uint32_t region_cap, uint32_t n_regions, uint32_t* total16, uint32_t* sketch, hipStream_t st);
// sample of the first batch's log -> *mode (0 keep logging, 1 direct atomics: the sampled keys repeat); ntc_apply.hip
hipError_t launch_log_probe(const uint32_t* log, const uint32_t* fill, uint32_t region_cap, uint32_t n_regions, uint32_t per_region, uint32_t* table,
                            uint32_t table_slots, unsigned long long* stats, uint32_t* mode, hipStream_t st);
hipError_t launch_split(const SplitArgs& a, unsigned grid, hipStream_t st);
hipError_t launch_count(const CountArgs& a, unsigned grid, hipStream_t st);
hipError_t set_apply_smem_limit();

hipError_t launch_hash(int mode, const HashArgs& a, unsigned grid, size_t smem, hipStream_t st);
hipError_t set_hash_smem_limit(size_t smem);
hipError_t launch_sketch_hf(const HfArgs& a, unsigned grid, unsigned waves_per_block, size_t smem, hipStream_t st);
hipError_t launch_compact_dump(const uint64_t* full, const uint32_t* valid, uint64_t n_reads, uint32_t n_win, uint32_t max_win, uint64_t* out, uint32_t* count, hipStream_t st);
hipError_t set_sketch_hf_smem_limit(size_t smem);
bool sketch_hf_deep_prefetch(uint32_t stride);
hipError_t launch_hll_threshold(const uint32_t* regs, uint32_t n_regs, uint32_t* thr, hipStream_t st);
hipError_t launch_finalize(const uint32_t* sketch, uint64_t n_per_sample, uint32_t* p_hist,
                           uint16_t* out16, hipStream_t st);
hipError_t launch_value_hist(const uint32_t* counters, uint64_t n, uint32_t* p_hist, hipStream_t st);
// dst[i] = dst[i] + src[i] (or max for nthll registers): the full-width part of ntc_merge_devices (F1, nthll register files)
hipError_t launch_fold_u32(uint32_t* dst, const uint32_t* src, uint64_t n, bool take_max, hipStream_t st);
hipError_t launch_fold_u64(unsigned long long* dst, const unsigned long long* src, uint64_t n, hipStream_t st);
hipError_t launch_add_counters(uint32_t* sketch, const uint16_t* add16, uint64_t n, hipStream_t st);
hipError_t launch_narrow_u16(const uint32_t* src, uint16_t* dst, uint64_t n, hipStream_t st);
hipError_t launch_sum_slices_u16(uint16_t* slices, uint64_t stride, uint32_t n_slices, uint64_t len, hipStream_t st);
hipError_t launch_widen_u16(const uint16_t* src, uint32_t* dst, uint64_t n, hipStream_t st);
hipError_t launch_gen(unsigned char* out, uint64_t seed, uint64_t first, uint64_t n, uint32_t len,
                      uint32_t stride, uint32_t dist, uint64_t glen, hipStream_t st);

} // namespace ntc
