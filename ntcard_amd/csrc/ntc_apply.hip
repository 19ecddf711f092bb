// ntc_apply.hip — deferred sketch update: hit log -> radix partition -> single-writer LDS histogram -> t_Counter.
//
// ntComp increments one uint16 counter per sampled k-mer (ntcard.cpp:132-145: `++t_Counter[indBit*rBuck + (h & mask)]`
// under `omp atomic`).  One device atomic per sampled k-mer caps any kernel at the chip's memory-side atomic rate
// (27 G/s measured, profiles/r01_ubench_atomic_flavours.txt), so the hash kernels do not touch the sketch at all:
// they append the counter index (`key` = indBit << rBits | h & mask) of every sampled k-mer to a hit log with
// coalesced stores.  Counting is a commutative sum (the reference's threads increment in arbitrary order), so the
// increments can be applied later and in any order:
//
//   A1/A2  split_kernel   radix partition of the log by the top bits of the key (one or two passes, <= 256 ways
//                         each; per-workgroup private output runs, so no global cursor atomics).  Per round a
//                         workgroup sorts 8192 keys by digit in LDS: ONE returning ds_add per key yields the digit
//                         count and the key's rank, the next round's keys are already in flight, and the sorted
//                         tile leaves as ~256-byte segments.  One 1024-thread workgroup per CU: the number of open
//                         runs (workgroups x digits x one 128-byte line) has to stay inside L2 for the run tails to
//                         be written once, so parallelism comes from the workgroup size, not from their number.
//                         Round 2 history per 183 M keys: two LDS atomics per key, 256 threads x 1024 workgroups
//                         0.85 + 0.65 ms; one returning atomic + prefetch, 256 x 512: 0.51 + 0.36 ms; this form ~0.6 ms.
//                         The last pass writes its runs as uint16 (round 5): the run says which slice, A3 needs the low
//                         slice_bits <= 15 bits only — 2 B per key less written and 2 B less read, half the scratch of that pass.
//   A3     count_kernel   one workgroup per slice of 2^15 counters: histogram of the slice's keys in LDS
//                         (ds_add on 16-bit fields, single writer per slice), then one coalesced `sketch[i] += n` sweep.
//
// Every run has a fixed capacity; a key that does not fit falls back to `atomicAdd(sketch + key, 1)`, which is
// exact too (just slower), so skewed inputs (one k-mer repeated millions of times) stay correct.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ntc_kernels.hpp"

namespace ntc {

namespace {

constexpr uint32_t kSplitThreads = 1024;
// keys per thread and round (template parameter of split_kernel): 8 -> 8192 keys per workgroup round, segments of ~64 keys (256 B) per digit at 128
// ways; 4 -> 4096 keys for passes of <= 64 ways — the same 256-byte segments, and a pass whose input runs are short (the second pass of an apply
// that comes long before the log is full: bench.py's 20 steps leave runs of ~11 K keys) wastes less of its last round per run (round 5: 1.36 rounds'
// worth of keys took 2 rounds of 8192, now 2.7 take 3 of 4096)
constexpr uint32_t kSplitKeysMax = 8;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	const int lane = threadIdx.x & 63;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(v, o);
		if (lane >= o) v += t;
	}
	return v;
}

} // namespace

// Packed runs between two partition passes (round 6): what the first pass leaves of a key — key_bits - b1 <= 21 bits — is written three to a 64-bit
// word (bits 0 .. 20, 21 .. 41, 42 .. 62; bit 63: fewer than three keys, then bit 42 says "one" instead of "two"), 2.7 B per key written and read
// instead of 4.  Every round pads the keys of a digit to whole words, so a word never mixes rounds and no run position is read back.
constexpr uint32_t kPackBits = 21;
constexpr uint32_t kPackMask = (1u << kPackBits) - 1u;
__device__ __forceinline__ uint32_t pack_count(unsigned long long w) { return (w >> 63) ? ((w >> 42) & 1ull ? 1u : 2u) : 3u; }

// A1/A2: partition the keys of this workgroup's input runs by digit = (key >> shift) & (2^bits - 1).  kPackOut: the runs are written packed (the first of two
// passes); kPackIn: the input runs are packed (the second).
//
// Round 6, "flat rounds": the lengths of ALL of a workgroup's input runs are fetched at once into LDS (up to 256 at a time), and the round that is being sorted
// always has the NEXT round's keys in flight — the next keys of the same run or the first ones of the next run that holds any.  Before, every run cost two
// exposed round trips to memory, its length and then its first keys, one after the other: 68 regions per workgroup in the first pass (a third of them hold
// keys), 64 runs in the second, ~2 us each.
#ifdef NTC_SPLIT_CLOCKS // timing experiment (tools/ab_build.sh <name> -DNTC_SPLIT_CLOCKS): first / last clock (100 MHz) and hardware id of every workgroup of the LAST second-pass launch
__device__ unsigned long long g_split_clocks[3 * 1024];
} // namespace ntc
extern "C" int ntc_dbg_split_clocks(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ntc::g_split_clocks), sizeof(unsigned long long) * 3 * 1024); }
namespace ntc {
#endif

template <uint32_t kSplitKeys, bool kPackIn, bool kPackOut>
__device__ __forceinline__ void split_body(const SplitArgs& a)
{
#ifdef NTC_SPLIT_CLOCKS
	const unsigned long long sc_t0 = __builtin_amdgcn_s_memrealtime();
#endif
	// keys of one round: kSplitKeys per thread, or (packed input) kSplitKeys / 2 words of up to three (A/B: -DNTC_AB_SPLIT_WORDS_DIV=4 — one word per thread in
	// the second pass — 546 against 476 us, profiles/r06_apply_packed_runs.txt)
#ifndef NTC_AB_SPLIT_WORDS_DIV
#define NTC_AB_SPLIT_WORDS_DIV 2
#endif
	constexpr uint32_t kWords = kSplitKeys / NTC_AB_SPLIT_WORDS_DIV;
	constexpr uint32_t kPerThread = kPackIn ? 3u * kWords : kSplitKeys;
	constexpr uint32_t kSplitRound = kSplitThreads * kPerThread;
	constexpr uint32_t kLoads = kPackIn ? kWords : kSplitKeys;
	constexpr uint32_t kLoadRound = kSplitThreads * kLoads;
#ifndef NTC_AB_RUN_CHUNK
#define NTC_AB_RUN_CHUNK 256
#endif
	constexpr uint32_t kRunChunk = NTC_AB_RUN_CHUNK; // run lengths held in LDS at a time (tests of the chunk loop: tools/ab_build.sh <name> -DNTC_AB_RUN_CHUNK=16 + NTCARD_LIB)
	// hist: digit counts of the round; excl: their exclusive scan (packed output: of the counts rounded up to whole words); rel: gcur - excl (run offset
	// of sorted position 0 of a digit; packed output: in words); gcur: keys (words) this workgroup has written per digit so far; cntd: the round's counts
	__shared__ uint32_t hist[256], excl[256], rel[256], gcur[256], cntd[kPackOut ? 256 : 1], run_n[kRunChunk], tot[1];
	__shared__ uint32_t sorted[kSplitRound + (kPackOut ? 2 * 256 : 0)]; // (+ the padding of a packed output's digits to whole words)
	const uint32_t tid = threadIdx.x, w = blockIdx.x;
	const uint32_t nb = 1u << a.bits, dmask = nb - 1u;
	constexpr bool pack_out = kPackOut; // (SplitArgs::pack_out; a template parameter: the plain form must not pay scalar registers for it, see split_packed_kernel)
	if (tid < 256) {
		gcur[tid] = 0;
		hist[tid] = 0;
	}
	// The t-th input run of this workgroup.  mode 0 (log regions): region (w + t) mod G of the t-th group of G = gridDim.x regions, not always the w-th — K1h's
	// wave g owns the regions g, g + W, ... and the first four waves of its workgroups log 18 % more than the last four (they walk more blocks:
	// sketch_k1h_kernel), so with G a multiple of 8 the w-th region of every group comes from the same kind of wave.  mode 1: the runs (w1, b), w1 = p, p + parts, ...
	uint32_t seg0, step, n_t, hi = 0;
	if (a.mode == 0) {
		seg0 = 0;
		step = gridDim.x;
		n_t = (a.n_in + step - 1u) / step;
	} else {
		const uint32_t b = w / a.parts, p = w % a.parts;
		seg0 = p * a.nb_in + b;
		step = a.parts * a.nb_in;
		n_t = seg0 < a.n_in ? (a.n_in - seg0 + step - 1u) / step : 0u;
		hi = b << a.hi_shift; // (packed input: the bits the first pass took, for the overflow fall-back's counter index)
	}
	auto seg_of = [&](uint32_t t) -> uint32_t {
#ifdef NTC_AB_OLD_ORDER
		return a.mode == 0 ? t * step + w : seg0 + t * step;
#else
		return a.mode == 0 ? t * step + (w + t) % step : seg0 + t * step;
#endif
	};
	// this workgroup's runs: uint32 keys, uint16 keys (narrow) or 64-bit words of three (pack_out: out_cap counts words) — ONE base pointer (scalar registers are
	// what decides whether two of these workgroups fit a CU, see split_packed_kernel)
	const bool narrow = a.narrow != 0;
	const uint32_t esz_log = pack_out ? 3u : (narrow ? 1u : 2u);
	unsigned char* const outb = reinterpret_cast<unsigned char*>(a.out) + (((uint64_t)w * nb * a.out_cap) << esz_log);
	using load_t = typename std::conditional<kPackIn, unsigned long long, uint32_t>::type;
	for (uint32_t t0 = 0; t0 < n_t; t0 += kRunChunk) {
		const uint32_t t_end = n_t - t0 < kRunChunk ? n_t : t0 + kRunChunk;
		__syncthreads(); // (run_n of the chunk before is spent; first chunk: gcur / hist are set)
		if (tid < kRunChunk && t0 + tid < t_end) {
			const uint32_t sg = seg_of(t0 + tid);
			const uint32_t n = sg < a.n_in ? a.in_cnt[sg] : 0u; // keys, or words of a packed run
			run_n[tid] = n < a.in_cap ? n : a.in_cap;
		}
		__syncthreads();
		auto next_run = [&](uint32_t t) -> uint32_t { // the first run from t on that holds anything (t_end: none)
			while (t < t_end && run_n[t - t0] == 0u)
				++t;
			return t;
		};
		load_t nxt[kLoads];
		auto fetch = [&](uint32_t t, uint32_t base) { // addresses clamped instead of predicated loads: branch-free, coalesced
			const uint32_t n = run_n[t - t0];
			const load_t* src = reinterpret_cast<const load_t*>(a.in) + (uint64_t)seg_of(t) * a.in_cap;
#pragma unroll
			for (int j = 0; j < (int)kLoads; ++j) {
				const uint32_t i = base + (uint32_t)j * kSplitThreads + tid;
				nxt[j] = src[i < n ? i : n - 1u];
			}
		};
		uint32_t t = next_run(t0), base = 0;
		if (t < t_end) fetch(t, 0);
		while (t < t_end) {
			const uint32_t n = run_n[t - t0];
			const uint32_t m_in = n - base < kLoadRound ? n - base : kLoadRound;
			// the round after this one: the same run's next keys, or the next run's first
			uint32_t tn = t, basen = base + kLoadRound;
			if (basen >= n) {
				tn = next_run(t + 1u);
				basen = 0;
			}
			uint32_t key[kPerThread], rank[kPerThread];
			bool have[kPerThread];
#pragma unroll
			for (int j = 0; j < (int)kLoads; ++j) {
				const uint32_t i = (uint32_t)j * kSplitThreads + tid;
				if constexpr (kPackIn) {
					const unsigned long long wd = nxt[j];
					const uint32_t nv = i < m_in ? pack_count(wd) : 0u;
#pragma unroll
					for (int q = 0; q < 3; ++q) {
						key[3 * j + q] = ((uint32_t)(wd >> (kPackBits * q)) & kPackMask) | hi;
						have[3 * j + q] = (uint32_t)q < nv;
					}
				} else {
					key[j] = (uint32_t)nxt[j];
					have[j] = i < m_in;
				}
			}
#pragma unroll
			for (int j = 0; j < (int)kPerThread; ++j) // ONE returning LDS atomic per key gives both the digit count and the key's rank inside its digit
				rank[j] = have[j] ? atomicAdd(&hist[(key[j] >> a.shift) & dmask], 1u) : 0u;
			if (tn < t_end) fetch(tn, basen); // the next round's keys are in flight during the sort
			__syncthreads();
			if (tid < 64) { // exclusive scan of the 256 digit counts (packed output: rounded up to whole words of three)
				uint32_t v[4], q[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					v[u] = hist[4 * tid + u];
					q[u] = pack_out ? (v[u] + 2u) / 3u * 3u : v[u];
				}
				const uint32_t s4 = q[0] + q[1] + q[2] + q[3];
				const uint32_t incl = wave_incl_scan(s4);
				uint32_t b0 = incl - s4;
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					excl[4 * tid + u] = b0;
					rel[4 * tid + u] = gcur[4 * tid + u] - (pack_out ? b0 / 3u : b0);
					if constexpr (kPackOut) cntd[4 * tid + u] = v[u];
					b0 += q[u];
				}
				if (tid == 63) tot[0] = incl;
			}
			__syncthreads();
#pragma unroll
			for (int j = 0; j < (int)kPerThread; ++j)
				if (have[j]) sorted[excl[(key[j] >> a.shift) & dmask] + rank[j]] = key[j];
			if (tid < 256) { // every thread owns its digit's slots: book the round, clear the counts for the next one
				gcur[tid] += pack_out ? (hist[tid] + 2u) / 3u : hist[tid];
				hist[tid] = 0;
			}
			__syncthreads();
			const uint32_t m = tot[0];
			if constexpr (kPackOut) {
				for (uint32_t j = tid; j < m / 3u; j += kSplitThreads) {
					const uint32_t k0 = sorted[3u * j], k1 = sorted[3u * j + 1u], k2 = sorted[3u * j + 2u];
					const uint32_t d = (k0 >> a.shift) & dmask;
					const uint32_t left = cntd[d] - (3u * j - excl[d]); // keys of the digit from this word on
					const uint32_t nv = left < 3u ? left : 3u;
					const uint32_t off = rel[d] + j;
					if (off < a.out_cap) {
						unsigned long long wd = (unsigned long long)(k0 & kPackMask);
						if (nv >= 2u) wd |= (unsigned long long)(k1 & kPackMask) << kPackBits;
						if (nv == 3u) wd |= (unsigned long long)(k2 & kPackMask) << (2u * kPackBits);
						else wd |= (1ull << 63) | (nv == 1u ? 1ull << 42 : 0ull);
						reinterpret_cast<unsigned long long*>(outb)[(uint64_t)d * a.out_cap + off] = wd;
					} else { // run is full: apply directly (exact, slower)
						atomicAdd(a.sketch + k0, 1u);
						if (nv >= 2u) atomicAdd(a.sketch + k1, 1u);
						if (nv == 3u) atomicAdd(a.sketch + k2, 1u);
						if (a.sk_dirty) *a.sk_dirty = 1u;
					}
				}
			} else {
				for (uint32_t i = tid; i < m; i += kSplitThreads) {
					const uint32_t kk = sorted[i];
					const uint32_t off = rel[(kk >> a.shift) & dmask] + i;
					if (off < a.out_cap) {
						const uint64_t at = (uint64_t)((kk >> a.shift) & dmask) * a.out_cap + off;
						if (narrow) reinterpret_cast<uint16_t*>(outb)[at] = (uint16_t)kk;
						else reinterpret_cast<uint32_t*>(outb)[at] = kk;
					} else {
						atomicAdd(a.sketch + kk, 1u); // run is full: apply directly (exact, slower)
						if (a.sk_dirty) *a.sk_dirty = 1u;
					}
				}
			}
			__syncthreads(); // sorted / rel are rewritten by the next round
			t = tn;
			base = basen;
		}
	}
	__syncthreads();
	if (tid < nb) a.out_cnt[(uint64_t)w * nb + tid] = gcur[tid] < a.out_cap ? gcur[tid] : a.out_cap;
#ifdef NTC_SPLIT_CLOCKS
	if (threadIdx.x == 0 && a.mode == 1 && blockIdx.x < 1024u) {
		g_split_clocks[3 * w] = sc_t0;
		g_split_clocks[3 * w + 1] = __builtin_amdgcn_s_memrealtime();
		g_split_clocks[3 * w + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); // HW_ID, XCC_ID
	}
#endif
}

template <uint32_t kSplitKeys, bool kPackOut>
__global__ __launch_bounds__(kSplitThreads) void split_kernel(const SplitArgs a)
{
	split_body<kSplitKeys, false, kPackOut>(a);
}
// (the packed-input form needed 81 + 6 SGPRs as the compiler allocated it: 96 per wave — and then only ONE workgroup of 1024 threads ran per CU instead of two,
// measured with per-workgroup clocks: 512 workgroups in two rounds of 256, 0.86 ms for the pass instead of 0.5; the plain form at 78 runs two)
template <uint32_t kSplitKeys>
__global__ __launch_bounds__(kSplitThreads) __attribute__((amdgpu_num_sgpr(80))) void split_packed_kernel(const SplitArgs a)
{
	split_body<kSplitKeys, true, false>(a);
}

// A3: one workgroup per slice of 2^slice_bits counters (<= 2^15): LDS histogram of the slice's keys, then
// sketch[slice] += histogram.  The slice has exactly one writer, so the sweep needs no atomics.  The LDS counts are
// 16 bits wide, two per dword (64 KiB per slice: two workgroups per CU overlap their phases); a pass takes at most
// 65535 keys, so no count can carry into its neighbour, and a slice with more keys is done in several passes.
// (round 6: the key width is a template parameter and the sweep has two straight-line forms.  As one body with `a.in16 ? ... : ...` per load and the sketch
// loads behind `!fresh && ...` the compiler put a branch around every load and s_waitcnt vmcnt(0) in front of every key's LDS atomic and in front of EVERY
// store of the sweep — eight stores per thread and slice, each waiting for the one before it to be acknowledged: the writes of a 64 KiB slice took as long as
// its key phase.)
#ifndef NTC_AB_COUNT_LOADS
#define NTC_AB_COUNT_LOADS 8
#endif
constexpr uint32_t kCountLoads = NTC_AB_COUNT_LOADS;
template <bool kIn16>
__global__ __launch_bounds__(1024) void count_kernel(const CountArgs a)
{
	extern __shared__ __align__(16) uint32_t cnt[]; // [(1 << slice_bits) / 2]
	using key_t = typename std::conditional<kIn16, uint16_t, uint32_t>::type;
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	// the run lengths through the scalar cache (they are uniform, and written by the kernel before this one): a vector load's s_waitcnt vmcnt(0) would also wait
	// for every store of the previous slice's sweep — the counter is in order — four dependent round trips per slice
	typedef const __attribute__((address_space(4))) uint32_t* scalar_ptr;
	const scalar_ptr in_cnt = (scalar_ptr)(uintptr_t)a.in_cnt;
	const uint32_t n_cnt = 1u << a.slice_bits, cmask = n_cnt - 1u, n_words = n_cnt >> 1;
	// the first apply behind a reset, and nothing has incremented the sketch directly: its counters are zero, so a slice's first pass WRITES its counts
	// (no read: half the sweep's traffic) and leaves the groups it has no key for alone
	const bool clean = a.first != 0u && (a.sk_dirty == nullptr || __builtin_amdgcn_readfirstlane((int)*a.sk_dirty) == 0);
	// the counts are zero between passes: zeroed once here, and the sweep clears every word it reads (round 6: no zeroing loop and one barrier less per pass:
	// 0.57 -> 0.50 ms per apply of 366 M keys)
	for (uint32_t i = tid; i < n_words / 4; i += nt)
		reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0, 0, 0, 0);
	if (a.clear_fill != nullptr) // (the log is spent: its regions were the first partition pass's input)
		for (uint32_t i = blockIdx.x * nt + tid; i < a.n_clear; i += gridDim.x * nt)
			a.clear_fill[i] = 0u;
	__syncthreads();
	for (uint32_t slice = blockIdx.x; slice < a.n_slices; slice += gridDim.x) {
		bool fresh = clean; // this slice has not been written yet
		uint32_t seg_add, seg_mul, seg_cnt;
		if (a.mode == 0) { // raw log regions, single slice
			seg_add = 0;
			seg_mul = 1;
			seg_cnt = a.n_in;
		} else if (a.mode == 1) { // runs of one split pass: (w1, b = slice)
			seg_add = slice;
			seg_mul = a.nb1;
			seg_cnt = a.nwg1;
		} else { // runs of two split passes: ((b, p), d2), slice = b * nb2 + d2
			const uint32_t b = slice / a.nb2, d2 = slice % a.nb2;
			seg_add = b * a.parts * a.nb2 + d2;
			seg_mul = a.nb2;
			seg_cnt = a.parts;
		}
		uint32_t* dst = a.sketch + ((uint64_t)slice << a.slice_bits);
		uint32_t t = 0, off = 0; // next run, and how much of it earlier passes took (a run longer than one pass is taken in pieces)
		while (t < seg_cnt && in_cnt[t * seg_mul + seg_add] == 0u) // leading empty runs (an empty log costs no LDS traffic at all)
			++t;
		while (t < seg_cnt) {
			uint32_t taken = 0; // keys of this pass (the same for every thread): at most 65535, so that no 16-bit count wraps into its neighbour
			while (t < seg_cnt && taken < 65535u) {
				const uint32_t seg = t * seg_mul + seg_add;
				uint32_t n = in_cnt[seg];
				n = n < a.in_cap ? n : a.in_cap;
				const uint32_t take = n - off < 65535u - taken ? n - off : 65535u - taken;
				const key_t* src = reinterpret_cast<const key_t*>(a.in) + (uint64_t)seg * a.in_cap + off;
				// kCountLoads loads in flight per thread before the first LDS atomic
				for (uint32_t base = 0; base < take; base += nt * kCountLoads) {
					uint32_t kq[kCountLoads];
#pragma unroll
					for (uint32_t j = 0; j < kCountLoads; ++j) {
						const uint32_t i = base + j * nt + tid;
						const uint32_t ic = i < take ? i : take - 1u; // (clamped address, not a predicated load)
						kq[j] = (uint32_t)src[ic];
					}
#pragma unroll
					for (uint32_t j = 0; j < kCountLoads; ++j) {
						const bool have = base + j * nt + tid < take;
						const uint32_t kk = kq[j] & cmask;
						// hot counters (a few thousand distinct k-mers sampled at huge coverage: every key of a run is the same) would put all 64
						// lanes on one LDS word, 64 serialised atomics per instruction: a wave whose keys are all equal adds their number once
						const uint64_t act = __ballot(have);
						if (act == 0) continue; // (wave-uniform)
						const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)kk, __builtin_ctzll(act));
						if (__ballot(have && kk == first) == act) {
							if ((tid & 63u) == (uint32_t)__builtin_ctzll(act)) atomicAdd(&cnt[kk >> 1], (uint32_t)__popcll(act) << ((kk & 1u) * 16u));
						} else if (have) {
							atomicAdd(&cnt[kk >> 1], 1u << ((kk & 1u) * 16u));
						}
					}
				}
				taken += take;
				off += take;
				if (off == n) {
					++t;
					off = 0;
				}
			}
			__syncthreads();
			if (taken != 0) {
				// 2 dwords of LDS = 4 counters = one uint4 of the sketch; four groups per thread and turn
				if (fresh) { // write-only: no load anywhere, the stores need not wait for each other
					for (uint32_t i0 = tid; i0 < n_words / 2; i0 += nt * 4u) {
						uint2 c[4];
#pragma unroll
						for (uint32_t j = 0; j < 4u; ++j) {
							const uint32_t i = i0 + j * nt;
							c[j] = i < n_words / 2 ? reinterpret_cast<const uint2*>(cnt)[i] : make_uint2(0, 0);
						}
#pragma unroll
						for (uint32_t j = 0; j < 4u; ++j) {
							const uint32_t i = i0 + j * nt;
							if ((c[j].x | c[j].y) != 0u) {
								reinterpret_cast<uint2*>(cnt)[i] = make_uint2(0, 0);
								reinterpret_cast<uint4*>(dst)[i] = make_uint4(c[j].x & 0xffffu, c[j].x >> 16, c[j].y & 0xffffu, c[j].y >> 16);
							}
						}
					}
				} else { // read-modify-write: the four groups' sketch words loaded together; a group without a key loads the slice's first word instead of its own (one
				         // cached line for all of them, no branch): a slice with a handful of keys — the other planes of a multi-k sketch — must not read 128 KiB
					for (uint32_t i0 = tid; i0 < n_words / 2; i0 += nt * 4u) {
						uint2 c[4];
						uint4 s4[4];
#pragma unroll
						for (uint32_t j = 0; j < 4u; ++j) {
							const uint32_t i = i0 + j * nt;
							c[j] = i < n_words / 2 ? reinterpret_cast<const uint2*>(cnt)[i] : make_uint2(0, 0);
							s4[j] = reinterpret_cast<const uint4*>(dst)[(c[j].x | c[j].y) != 0u ? i : 0u];
						}
#pragma unroll
						for (uint32_t j = 0; j < 4u; ++j) {
							const uint32_t i = i0 + j * nt;
							if ((c[j].x | c[j].y) != 0u) {
								reinterpret_cast<uint2*>(cnt)[i] = make_uint2(0, 0);
								uint4 s = s4[j];
								s.x += c[j].x & 0xffffu;
								s.y += c[j].x >> 16;
								s.z += c[j].y & 0xffffu;
								s.w += c[j].y >> 16;
								reinterpret_cast<uint4*>(dst)[i] = s;
							}
						}
					}
				}
				fresh = false;
			}
			__syncthreads();
		}
	}
}

// Log or direct atomics?  A direct atomic per sampled k-mer is cheap when the counters it hits stay in the Infinity
// Cache, i.e. when the k-mers of a batch repeat moderately (coverage of a mid-sized genome); for mostly distinct k-mers it
// costs a 128-byte HBM round trip each, for a handful of hot counters the atomics queue up on the same addresses: in both
// of those cases the log + partition wins.  The probe looks at a sample of what the
// first batch logged: keys go into a small open hash table, a key that finds itself there is a repeat.
__global__ __launch_bounds__(256) void log_probe_kernel(const uint32_t* __restrict__ log, const uint32_t* __restrict__ fill, uint32_t region_cap,
                                                        uint32_t n_regions, uint32_t per_region, uint32_t* __restrict__ table, uint32_t table_mask,
                                                        unsigned long long* __restrict__ stats)
{
	uint32_t seen = 0, rep = 0;
	for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
		uint32_t n = fill[r];
		n = n < region_cap ? n : region_cap;
		n = n < per_region ? n : per_region;
		for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
			const uint32_t key = log[(uint64_t)r * region_cap + i];
			const uint32_t slot = (key * 0x9e3779b1u) >> 8 & table_mask;
			const uint32_t old = atomicCAS(&table[slot], 0u, key + 1u);
			++seen;
			rep += old == key + 1u;
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		seen += __shfl_xor(seen, o);
		rep += __shfl_xor(rep, o);
	}
	if ((threadIdx.x & 63u) == 0u && seen) {
		atomicAdd(stats, (unsigned long long)seen);
		atomicAdd(stats + 1, (unsigned long long)rep);
	}
}

__global__ void log_decide_kernel(unsigned long long* stats, uint32_t* mode, unsigned long long min_keys)
{
	// uniform k-mers: ~0.05 % of a 256 K sample repeat; 15 x coverage of a 100 Mbp genome per batch: ~5 %.  When nearly every
	// sampled key repeats (a few thousand hot counters: a small genome at huge coverage, or short / spaced k-mers that saturate
	// their 4^k space) the atomics serialise on the same addresses and the log wins again (tools/mode_sweep.py: k = 12 with gap 2
	// 1.20 vs 1.47 ms per 10 M reads; k = 20 on a 100 kbp genome 1.19 vs 1.65 ms)
	if (stats[0] >= min_keys && *mode == 0u) *mode = (stats[1] * 100ull > stats[0] && stats[1] * 100ull < 90ull * stats[0]) ? 1u : 0u;
	stats[0] = 0;
	stats[1] = 0;
}

hipError_t launch_log_probe(const uint32_t* log, const uint32_t* fill, uint32_t region_cap, uint32_t n_regions, uint32_t per_region, uint32_t* table,
                            uint32_t table_slots, unsigned long long* stats, uint32_t* mode, hipStream_t st)
{
	hipError_t rc = hipMemsetAsync(table, 0, (size_t)table_slots * 4, st);
	if (rc != hipSuccess) return rc;
	hipLaunchKernelGGL(log_probe_kernel, dim3(n_regions < 1024u ? n_regions : 1024u), dim3(256), 0, st, log, fill, region_cap, n_regions, per_region, table,
	                   table_slots - 1u, stats);
	hipLaunchKernelGGL(log_decide_kernel, dim3(1), dim3(1), 0, st, stats, mode, (unsigned long long)(1u << 16));
	return hipGetLastError();
}


// A log that holds little (the head batch the mode was decided on, a tail at finish, everything after the switch to
// direct atomics) is applied with plain atomics: the partition passes and the sweep over every slice have a fixed cost of
// ~0.4 ms at rBits = 27.  The host does not know how much was logged (the mode lives on the device), so the first
// kernel adds up the region fills and the second acts on the sum; the regions it applies are marked empty.
__global__ __launch_bounds__(1024) void log_total_kernel(const uint32_t* __restrict__ fill, uint32_t n_regions, uint32_t* __restrict__ total)
{
	__shared__ uint32_t part[16];
	uint32_t s = 0;
	for (uint32_t r = threadIdx.x; r < n_regions; r += blockDim.x)
		s += fill[r] >> 4; // in units of 16 entries: cannot overflow
	for (int o = 32; o > 0; o >>= 1)
		s += __shfl_xor(s, o);
	if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (uint32_t w = 0; w < blockDim.x / 64; ++w)
			t += part[w];
		*total = t;
	}
}
__global__ __launch_bounds__(256) void log_atomics_kernel(const uint32_t* __restrict__ log, uint32_t* __restrict__ fill, uint32_t region_cap,
                                                          uint32_t n_regions, const uint32_t* __restrict__ total16, uint32_t max16, uint32_t* __restrict__ sketch,
                                                          uint32_t* __restrict__ sk_dirty)
{
	if (*total16 > max16) return; // plenty: the partition passes take it
	if (sk_dirty && blockIdx.x == 0 && threadIdx.x == 0) *sk_dirty = 1u;
	for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
		uint32_t n = fill[r];
		n = n < region_cap ? n : region_cap;
		for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
			atomicAdd(sketch + log[(uint64_t)r * region_cap + i], 1u);
		__syncthreads();
		if (threadIdx.x == 0) fill[r] = 0;
	}
}

hipError_t launch_log_atomics(const uint32_t* log, uint32_t* fill, uint32_t region_cap, uint32_t n_regions, uint32_t* total16, uint32_t* sketch, uint32_t* sk_dirty,
                              hipStream_t st)
{
	hipLaunchKernelGGL(log_total_kernel, dim3(1), dim3(1024), 0, st, fill, n_regions, total16);
	hipLaunchKernelGGL(log_atomics_kernel, dim3(n_regions < 4096u ? n_regions : 4096u), dim3(256), 0, st, log, fill, region_cap, n_regions, total16,
	                   (4u << 20) >> 4, sketch, sk_dirty);
	return hipGetLastError();
}

// ntc_log_export_device: the keys of a region counted per owner in LDS, ONE cursor atomic per owner and region, then placed by a second LDS count
// (the region is read twice; the second time from L2)
__global__ __launch_bounds__(256) void log_export_kernel(const uint32_t* __restrict__ log, const uint32_t* __restrict__ fill, uint32_t region_cap, uint32_t n_regions,
                                                         uint32_t n_parts, uint32_t keys_per_part, uint32_t* __restrict__ out,
                                                         const unsigned long long* __restrict__ part_off, unsigned long long* __restrict__ cursor)
{
	__shared__ uint32_t cnt[64];
	__shared__ unsigned long long base[64];
	const uint32_t tid = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
		uint32_t n = fill[r];
		n = n < region_cap ? n : region_cap;
		if (n == 0) continue; // (the same for every thread)
		const uint32_t* src = log + (uint64_t)r * region_cap;
		if (tid < 64) cnt[tid] = 0;
		__syncthreads();
		for (uint32_t i = tid; i < n; i += 256u) {
			const uint32_t o = src[i] / keys_per_part;
			atomicAdd(&cnt[o < n_parts ? o : n_parts - 1u], 1u);
		}
		__syncthreads();
		if (tid < n_parts) {
			base[tid] = cnt[tid] ? atomicAdd(&cursor[tid], (unsigned long long)cnt[tid]) : 0ull;
			cnt[tid] = 0;
		}
		__syncthreads();
		if (out != nullptr)
			for (uint32_t i = tid; i < n; i += 256u) {
				const uint32_t key = src[i];
				uint32_t o = key / keys_per_part;
				o = o < n_parts ? o : n_parts - 1u;
				out[part_off[o] + base[o] + atomicAdd(&cnt[o], 1u)] = key;
			}
		__syncthreads();
	}
}

hipError_t launch_log_export(const uint32_t* log, const uint32_t* fill, uint32_t region_cap, uint32_t n_regions, uint32_t n_parts, uint32_t keys_per_part,
                             uint32_t* out, const unsigned long long* part_off, unsigned long long* cursor, hipStream_t st)
{
	hipLaunchKernelGGL(log_export_kernel, dim3(n_regions < 2048u ? n_regions : 2048u), dim3(256), 0, st, log, fill, region_cap, n_regions, n_parts, keys_per_part, out,
	                   part_off, cursor);
	return hipGetLastError();
}

__global__ void log_set_fill_kernel(uint32_t* __restrict__ fill, uint32_t n_regions, uint32_t region_cap, unsigned long long n_keys)
{
	for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_regions; r += gridDim.x * blockDim.x) {
		const unsigned long long lo = (unsigned long long)r * region_cap;
		fill[r] = n_keys <= lo ? 0u : (n_keys - lo < region_cap ? (uint32_t)(n_keys - lo) : region_cap);
	}
}

hipError_t launch_log_set_fill(uint32_t* fill, uint32_t n_regions, uint32_t region_cap, unsigned long long n_keys, hipStream_t st)
{
	hipLaunchKernelGGL(log_set_fill_kernel, dim3((n_regions + 255u) / 256u), dim3(256), 0, st, fill, n_regions, region_cap, n_keys);
	return hipGetLastError();
}

hipError_t launch_split(const SplitArgs& a, unsigned grid, hipStream_t st)
{
	static_assert(kSplitKeysMax == 8, "two sizes of a round");
	if (a.pack_in) {
		if (a.bits >= 7) hipLaunchKernelGGL(split_packed_kernel<8>, dim3(grid), dim3(kSplitThreads), 0, st, a);
		else hipLaunchKernelGGL(split_packed_kernel<4>, dim3(grid), dim3(kSplitThreads), 0, st, a);
	} else {
		if (a.pack_out) {
			if (a.bits >= 7) hipLaunchKernelGGL((split_kernel<8, true>), dim3(grid), dim3(kSplitThreads), 0, st, a);
			else hipLaunchKernelGGL((split_kernel<4, true>), dim3(grid), dim3(kSplitThreads), 0, st, a);
		} else {
			if (a.bits >= 7) hipLaunchKernelGGL((split_kernel<8, false>), dim3(grid), dim3(kSplitThreads), 0, st, a);
			else hipLaunchKernelGGL((split_kernel<4, false>), dim3(grid), dim3(kSplitThreads), 0, st, a);
		}
	}
	return hipGetLastError();
}

hipError_t launch_count(const CountArgs& a, unsigned grid, hipStream_t st)
{
	const size_t smem = (sizeof(uint32_t) << a.slice_bits) / 2;
	// (two workgroups of 1024 threads or four of 512 per CU)
	if (a.in16) hipLaunchKernelGGL(count_kernel<true>, dim3(grid), dim3(a.slice_bits >= 15 ? 1024 : 512), smem, st, a);
	else hipLaunchKernelGGL(count_kernel<false>, dim3(grid), dim3(a.slice_bits >= 15 ? 1024 : 512), smem, st, a);
	return hipGetLastError();
}

hipError_t set_apply_smem_limit()
{
	hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&count_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
	if (rc == hipSuccess) rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&count_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
	return rc;
}

} // namespace ntc
