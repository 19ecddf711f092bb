// ntc_sketch_hf.hip — K1 "H-filter": the production ntHash -> sample -> count kernel for gfx950.
//
// Profiling of the straightforward lane-per-read kernel (ntc_sketch_fast.hip, profiles/r01_*) showed
// it to be bound by the number of wave instructions issued, not by HBM, LDS or latency.  This
// variant therefore does the minimum per k-mer and moves everything that is only needed for the
// ~2^(1-sBits) sampled k-mers out of the per-base loop:
//
//   * the sampling decision of ntComp (ntcard.cpp:135-138) only looks at the top sBits+1 bits of
//     min(fh, rh); those bits live in the 31-bit rotating half H of the hash (nthash.hpp:186-217), and
//     H rolls independently of the 33-bit half.  The per-base loop therefore rolls ONLY the two H
//     halves (3 VALU ops each, one ds_read_b64 of seed terms) and tests min(fHd, rHd);
//   * a lane that sees a sampled window just shifts a 1 into a mask register (one v_addc): no branch, no
//     memory traffic;
//   * every 32 steps the wave queues the sampled (lane, step) pairs (one register per lane, filled through
//     ds_permute) and, 64 pairs at a time, recomputes the full 64-bit forward and reverse hashes of those windows
//     from the closed form
//     fh = XOR_i srol^(k-1-i)(seed(c_i)), rh = XOR_i srol^i(comp(c_i))      (nthash.hpp:220-239)
//     with a ceil(k/2) x 16 table of pre-rotated seed PAIRS in LDS (two bases per lookup), takes the canonical
//     min (nthash.hpp:275-279) and updates the sketch.  Full 64-bit compare: exact, no tie special case;
//   * equal-length waves start from the same closed form instead of rolling the k-1 window-filling steps; up to
//     four values of k share one launch (the batch is staged and decoded once); spaced seeds roll the spaced
//     value itself; nthll swaps the sample test for a register threshold.  LDS holds nothing but the decoded
//     slots and the tables, which is what lets 16 waves of 150 bp reads share a CU.
//
// Semantics reproduced: ntRead (ntcard.cpp:147-158), ntHashIterator (ntHashIterator.hpp:59-86),
// NTMC64 (nthash.hpp:381-390,467-492), ntComp (ntcard.cpp:132-145).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ntc_kernels.hpp"

namespace ntc {

namespace {

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
	return __builtin_amdgcn_alignbit(hi, lo, sh);
}
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
	return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
	return __builtin_amdgcn_perm(s0, s1, sel);
}

// LDS by integer offset: lets a table address be formed with one v_and_or_b32 / SDWA v_or (index | aligned base)
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4u32* lds_v4_ptr;
typedef __attribute__((address_space(3))) unsigned char* lds_byte_ptr;
__device__ __forceinline__ uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)(lds_byte_ptr)p; }
__device__ __forceinline__ v4u32 lds_read4(uint32_t off) { return *(lds_v4_ptr)(uintptr_t)off; }
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// Closed-form table addresses of 4 window bases w = [c0 c1 c2 c3] (code<<6 per byte): pair (c0,c1) -> a0, (c2,c3) -> a1.
// Entry offset of a pair is c_even<<6 | c_odd<<4; `tp` is the 256-byte aligned LDS offset of the table of this pair
// position (the next position's table follows 256 B later).  5 VALU instead of the 9 hipcc derives from the C form.
__device__ __forceinline__ void pair_addresses(uint32_t w, uint32_t tp, uint32_t& a0, uint32_t& a1)
{
	uint32_t m_even = 0x00c000c0u, m_odd = 0x00300030u, tp1 = tp + 256u;
	uint32_t hi, y;
	asm("v_lshrrev_b32_e32 %0, 10, %1" : "=v"(hi) : "v"(w));
	asm("v_and_b32_e32 %0, %1, %2" : "=v"(y) : "s"(m_even), "v"(w));
	asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(y) : "v"(hi), "s"(m_odd), "v"(y));
	asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(a0) : "v"(y), "s"(tp));
	asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a1) : "v"(y), "s"(tp1));
}

// v_perm tables indexed by (byte & 7): 1:A 3:C 7:G 4:T 5:U, 0/2/6: not a base (nthash.hpp:16,32 trick)
constexpr uint32_t kExpS0 = 0x47ff5554u; // 'G', ff, 'U', 'T'
constexpr uint32_t kExpS1 = 0x43ff41ffu; // 'C', ff, 'A', ff
constexpr uint32_t kIn6S0 = 0x8000c0c0u; // code<<6 : G=2, -, U=3, T=3
constexpr uint32_t kIn6S1 = 0x40000000u; //           C=1, -, A=0, -

// x << 1 as a full-rate v_add_u32 (hipcc canonicalises x + x back into the half-rate v_lshlrev_b32)
__device__ __forceinline__ uint32_t dbl(uint32_t x)
{
	uint32_t r;
	asm("v_add_u32_e32 %0, %1, %1" : "=v"(r) : "v"(x));
	return r;
}

// decode one dword of raw bytes -> code<<6 per byte, bit 0 set on bytes that are not ACGTU/acgtu
__device__ __forceinline__ uint32_t decode4(uint32_t w, uint32_t& badacc)
{
	const uint32_t sel = w & 0x07070707u;
	const uint32_t bad = (perm(kExpS0, kExpS1, sel) ^ w) & 0xdfdfdfdfu;
	uint32_t code = perm(kIn6S0, kIn6S1, sel);
	if (bad != 0u) {
		badacc = 1u; // noted on the rare path only
		const uint32_t nz = (((bad & 0x7f7f7f7fu) + 0x7f7f7f7fu) | bad) & 0x80808080u;
		code = (code & ~(nz >> 1) & ~nz) | (nz >> 7); // dirty byte: code 0, mark bit 0
	}
	return code;
}

// wave ballot of a bool without the int round trip hipcc's ballot() goes through (saves 2 VALU ops per use)
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }


} // namespace

// kMulti = false: exactly one k (the loop over the k list folds away and every per-k value is a launch constant)
// kMode: 0 plain k-mers, 1 spaced seed (-g), 2 nthll — separate instantiations keep each one's register budget
// free of the other modes' state (the spaced-seed walks cost the plain kernel 5 spilled VGPRs otherwise)
// kPref: 1 KiB chunks of the NEXT batch a wave keeps in flight in registers (10 covers slots of up to 160 B at 16 waves
// per CU; 16 covers 256 B slots, whose LDS footprint allows 12 waves at most, hence the smaller launch bound)
// kDump: validation build (ntc_hash_dump_k1_device): the filter lets EVERY window through, so the resolve stage
// re-derives the full canonical hash of every window with the production code path, and writes it out instead of
// sampling it (what ntHashIterator / stHashIterator enumerate, ntHashIterator.hpp:59-86, stHashIterator.hpp:60-87)
#ifdef NTC_HF_CLOCKS // timing experiment (tools/ab_build.sh <name> -DNTC_HF_CLOCKS): first / last clock (100 MHz) of every wave of the LAST launch
__device__ unsigned long long g_hf_clocks[2 * 8192];
} // namespace ntc
extern "C" int ntc_dbg_hf_clocks(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ntc::g_hf_clocks), sizeof(unsigned long long) * 2 * 8192); }
namespace ntc {
#endif

template <bool kMulti, int kMode, int kPref, bool kDump = false, bool kTiled = false>
__global__ __launch_bounds__(kPref <= 10 ? 1024 : 768) void sketch_hf_kernel(const HfArgs a)
{
#ifdef NTC_HF_CLOCKS
	const unsigned long long hc_t0 = __builtin_amdgcn_s_memrealtime();
#endif
	const uint32_t n_k = kMulti ? a.n_k : 1u;
	// Nothing but the tables and the decoded slots lives in LDS: hit masks and the compaction queue stay in registers, so that a CU's 160 KiB
	// hold 16 waves of 150 bp reads (4 per SIMD) instead of 12.
	extern __shared__ __align__(16) unsigned char smem[];
	// static LDS: per-(in,out) seed terms of the H halves {Tf.Hd, Tr.Hd}, 16-byte stride (offset = idx byte)
	__shared__ __align__(16) uint32_t tabH[kMaxFusedK][kMainSlots * 4];
	__shared__ __align__(16) uint32_t tabG[kMainSlots * 4]; // spaced seed, rolling form (same addressing as tabH)
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	// waves per block: chosen by the host so that one CU holds as many waves as its 160 KiB of LDS allow
	// (every wave parks its 64 slots in LDS; the closed-form table is shared by the block)
	const uint32_t wpb = blockDim.x >> 6;
	const uint32_t stride = a.stride;
	// per-k state of the fused multi-k loop (the lambdas below see the current values)
	uint32_t k = a.ks[0].k;
	uint32_t t1_off[kMaxFusedK + 1]; // LDS offsets of the closed-form tables, [n_k] = total
	t1_off[0] = 0;
	for (uint32_t j = 0; j < (uint32_t)kMaxFusedK; ++j)
		t1_off[j + 1] = t1_off[j] + (j < n_k ? ((a.ks[j].k + 1u) >> 1) * 256u : 0u);
	// dynamic LDS: [closed-form tables of every fused k][gap table][16 B pad][waves x 64 slots].  The tables come first:
	// with 1280 B of static LDS in front they start 256-byte aligned, so `index | base` addresses them.
	const uint32_t tables_bytes = t1_off[kMaxFusedK] + ((a.gap + 1u) >> 1) * 256u;
	unsigned char* const wdata = smem + tables_bytes + 16 + (size_t)wave * 64u * stride;
	const unsigned char* const mine = wdata + (size_t)lane * stride;
	unsigned char* const t1_base = smem;
	if ((lds_off(smem) & 255u) != 0u) __builtin_trap(); // the static tables are sized to keep this aligned
	unsigned char* t1 = t1_base;
	// spaced seed (stRead, ntcard.cpp:160-171): per pair of don't-care positions, the H halves of the terms to XOR out
	const uint32_t ngp = (a.gap + 1u) >> 1;
	unsigned char* const gapT = t1_base + t1_off[kMaxFusedK];
	// Sample 0 of ntComp wants the top sBits+1 bits of min(fh,rh) to be 0..01.  Both strands are carried with
	// that one bit flipped (folded into the step table: x' = x ^ c rolls with the term t ^ c ^ rotl(c)), so the
	// test becomes min(f',r') < c: a superset (extra: one strand 0..01 while the other is 0..00, p = 2^-2(sBits+1)),
	// made exact by the resolve stage, which re-derives everything from the bases.  Sample 1 looks at the bits
	// above c only and is unaffected.  nthll compares against a moving threshold and runs unflipped.
	const uint32_t flipc = kMode == 2 ? 0u : 1u << (31 - a.s_bits);
	const uint32_t flipx = flipc ^ (flipc << 1);
	{
		for (uint32_t j = 0; j < n_k; ++j) {
			for (int i = tid; i < kMainSlots * 4; i += (int)blockDim.x) {
				const int slot = i >> 2, w = i & 3;
				tabH[j][i] = w < 2 ? (a.ks[j].tabh[slot][w] ^ flipx) : 0u;
			}
			const uint4* src = reinterpret_cast<const uint4*>(a.ks[j].t1);
			uint4* dst = reinterpret_cast<uint4*>(t1_base + t1_off[j]);
			for (uint32_t i = tid; i < (t1_off[j + 1] - t1_off[j]) / 16u; i += blockDim.x)
				dst[i] = src[i];
		}
		if (a.gap != 0)
			for (int i = tid; i < kMainSlots * 4; i += (int)blockDim.x)
				tabG[i] = (i & 3) < 2 ? a.tabg[i >> 2][i & 3] : 0u;
		const uint4* gsrc = reinterpret_cast<const uint4*>(a.gapt);
		for (uint32_t i = tid; i < ngp * 16u; i += blockDim.x)
			reinterpret_cast<uint4*>(gapT)[i] = gsrc[i];
	}
	__syncthreads();
	const unsigned char* tabHb = reinterpret_cast<const unsigned char*>(tabH[0]);
	uint32_t* sketch_k = a.ks[0].sketch;
	uint32_t key_base = a.ks[0].key_base;

	// sample windows on the top bits (ntcard.cpp:135-138); VGPR-resident on purpose (SGPR sources halve the VALU rate)
	uint32_t lo0 = 1u << (31 - a.s_bits);
	int32_t lo1 = (int32_t)(((1u << (a.s_bits - 1)) - 1u) << (32 - a.s_bits));
	asm volatile("" : "+v"(lo0), "+v"(lo1));
	const uint32_t s_bits = a.s_bits;
	uint32_t hll_thr = a.hll_bits ? *a.hll_thr : 0u;
	asm volatile("" : "+v"(hll_thr));
	const uint32_t rmask = (1u << a.r_bits) - 1u;
	const uint32_t rbuck = 1u << a.r_bits;

	const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * wpb + wave);
	const uint64_t n_slots = a.n_slots;
	const uint64_t n_wb = (n_slots + 63) / 64;
	// Hit log: ntComp's `++t_Counter[...]` (ntcard.cpp:142-143) is not executed here.  The wave appends the counter
	// index of every sampled k-mer to its private log regions gwave, gwave + W, ... (W = waves of this launch) with one
	// coalesced store per resolve round; ntc_apply.hip adds them to the sketch later (counting commutes).  A wave
	// that runs out of regions falls back to the direct atomic, which is exact as well.
	const bool use_log = kMode != 2 && a.log_regions != 0 && (a.log_mode == nullptr || __builtin_amdgcn_readfirstlane(*a.log_mode) == 0u);
	const uint32_t log_w = gridDim.x * wpb;
	uint32_t lreg = gwave, lfill = 0;
	if (use_log && lreg < a.log_regions) lfill = __builtin_amdgcn_readfirstlane(a.log_fill[lreg]);
	auto log_emit = [&](bool hit, uint32_t key) {
		const uint64_t m = ballot(hit);
		if (m == 0) return;
		const uint32_t c = (uint32_t)__popcll(m);
		while (lreg < a.log_regions && c > a.log_region_cap - lfill) { // region full: book its fill, take this wave's next one
			if (lane == 0) a.log_fill[lreg] = lfill;
			lreg += log_w;
			lfill = lreg < a.log_regions ? __builtin_amdgcn_readfirstlane(a.log_fill[lreg]) : 0u;
		}
		if (lreg < a.log_regions) {
			const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			if (hit) a.log[(uint64_t)lreg * a.log_region_cap + lfill + pos] = key;
			lfill += c;
		} else if (hit) {
			atomicAdd(a.sketch0 + key, 1u);
			if (a.sk_dirty) *a.sk_dirty = 1u; // (out of log regions: the sketch is no longer what the last reset / apply left, ntc_apply.hip count_kernel)
		}
	};
	if (kMode != 2 && !use_log && a.sk_dirty != nullptr && lane == 0) *a.sk_dirty = 1u; // direct atomics from here on (the device mode word, or no regions)
	uint64_t f1_acc[kMaxFusedK] = {0, 0, 0, 0};
	uint32_t shb = (0u - k) & 3u; // byte phase of the outgoing-base stream

	// ---- global -> LDS staging with register prefetch of the next batch (see ntc_sketch_fast.hip) ----
	const uint32_t full_bytes = 64u * stride;
	const uint32_t nchunk = (full_bytes + 1023u) >> 10;
	const bool can_prefetch = nchunk <= (uint32_t)kPref;
	uint4 pref[kPref];
	// A TILED batch (a.tiled, round 5: the k of a list that K1h is not built for are hashed here from the same tiles, without a re-layout pass): the wave's
	// 64 reads are 64 consecutive 16-byte pieces of every chunk row of their tile, so round c of the staging is ONE coalesced 1 KiB load — piece c of read
	// `lane` — parked at lane * stride + 16 c: the same LDS picture as a row-slot batch of stride 16 n_chunks.  The slots behind a batch's last read exist
	// (a tile buffer always holds whole tiles), so a partial last wave loads like a full one and masks its lanes afterwards.
	constexpr bool tiled = kTiled; // (an instantiation of its own: as a run-time flag the two addressing forms cost every variant 150 - 330 B of scratch per lane)
	const uint32_t n_pieces = stride >> 4; // (tiled: chunks per read)
	auto load_round = [&](uint64_t wb_, uint32_t c0) {
		if (tiled) {
			const uint64_t r0 = wb_ * 64u;
			const unsigned char* src = a.slots + ((r0 / kTileReads) * n_pieces * kTileReads + (r0 % kTileReads) + (uint32_t)lane) * 16u;
#pragma unroll
			for (int c = 0; c < kPref; ++c)
				if (c0 + c < n_pieces) pref[c] = *reinterpret_cast<const uint4*>(src + (size_t)(c0 + c) * kTileReads * 16u);
			return;
		}
		const unsigned char* src = a.slots + wb_ * full_bytes;
#pragma unroll
		for (int c = 0; c < kPref; ++c) {
			const uint32_t off = lane * 16u + (c0 + c) * 1024u;
			if (off + 16u <= full_bytes) pref[c] = *reinterpret_cast<const uint4*>(src + off);
		}
	};
	auto store_round = [&](uint32_t c0, uint32_t& badacc) {
#pragma unroll
		for (int c = 0; c < kPref; ++c) {
			const uint32_t off = tiled ? (uint32_t)lane * stride + (c0 + c) * 16u : lane * 16u + (c0 + c) * 1024u;
			if (tiled ? c0 + c < n_pieces : off + 16u <= full_bytes) {
				uint4 v = pref[c];
				v.x = decode4(v.x, badacc);
				v.y = decode4(v.y, badacc);
				v.z = decode4(v.z, badacc);
				v.w = decode4(v.w, badacc);
				*reinterpret_cast<uint4*>(wdata + off) = v;
			}
		}
	};
	auto is_full = [&](uint64_t wb_) { return tiled || wb_ * 64 + 64 <= n_slots; };
	// The wave's batches (round 6): a contiguous share of its workgroup's contiguous share of the launch's wave-batches, weighed by the wave's AGE on its SIMD.
	// Before, wave g took the batches g, g + W, g + 2 W, ... — the same number for every wave — and the SIMDs, which issue their older waves first, finished the
	// waves of a workgroup in the order of their age: 2090 / 2161 / 2255 us into a 2.33 ms launch with three waves per SIMD (config 4), 806 / 820 / 865 / 877 into
	// 0.92 ms with four (per-wave clocks of a -DNTC_HF_CLOCKS build, profiles/r06_k1_wave_clocks.txt); the youngest ran on alone at the end.  Waves 4 r .. 4 r + 3 of
	// a workgroup are the r-th on their SIMDs; the weights are the speeds those clocks showed, in 1 / 1024 of a SIMD's batches.
	uint64_t wb_begin, wb_end;
	{
		static constexpr uint16_t kAgeShare[5][4] = { { 0, 0, 0, 0 }, { 1024, 0, 0, 0 }, { 530, 494, 0, 0 }, { 354, 342, 328, 0 }, { 267, 263, 249, 245 } };
		const uint32_t ranks = (wpb + 3u) / 4u; // (1 .. 4)
		auto cum = [&](uint32_t w) -> uint32_t { // the shares of the waves in front of wave w, in 1 / 1024 of a SIMD's batches
			uint32_t c = 0;
			for (uint32_t r = 0; r < ranks; ++r) {
				const uint32_t in_rank = w > 4u * r ? (w - 4u * r < 4u ? w - 4u * r : 4u) : 0u; // waves of rank r in front of w
				c += in_rank * (wpb % 4u == 0u ? kAgeShare[ranks][r] : 1024u / ranks);
			}
			return c;
		};
		const uint64_t per_wg = (n_wb + gridDim.x - 1) / gridDim.x;
		const uint64_t g0 = (uint64_t)blockIdx.x * per_wg < n_wb ? (uint64_t)blockIdx.x * per_wg : n_wb;
		const uint64_t g1 = g0 + per_wg < n_wb ? g0 + per_wg : n_wb;
		const uint32_t total = cum(wpb);
		wb_begin = g0 + (g1 - g0) * cum((uint32_t)wave) / total;
		wb_end = (uint32_t)wave + 1u == wpb ? g1 : g0 + (g1 - g0) * cum((uint32_t)wave + 1u) / total;
		wb_begin = __builtin_amdgcn_readfirstlane((uint32_t)wb_begin) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wb_begin >> 32)) << 32);
		wb_end = __builtin_amdgcn_readfirstlane((uint32_t)wb_end) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wb_end >> 32)) << 32);
	}
	if (can_prefetch && wb_begin < wb_end && is_full(wb_begin)) load_round(wb_begin, 0);
	for (uint64_t wb = wb_begin; wb < wb_end; ++wb) {
		const uint64_t wb_next = wb + 1 < wb_end ? wb + 1 : n_wb; // (n_wb: none)
		const uint64_t slot0 = wb * 64;
		const uint32_t nvalid = (uint32_t)((n_slots - slot0) < 64 ? (n_slots - slot0) : 64);
		uint32_t badacc = 0;
		// Wave priorities follow the age of the work: staging (waits on memory anyway) 0, the walk 1, the walk after
		// its first compaction 2, compaction + resolve 3.  A batch that has been started gets finished ahead of
		// younger ones on the same SIMD instead of all four waves trading issue slots evenly: +13-15 % on genome-like
		// data (1.03 -> 0.90 ms per launch), -5 % on uniform data, which runs at the atomic-rate floor either way.
		// Slots too long for the register prefetch keep the staging at the walk's level (it would starve otherwise).
		if (can_prefetch) __builtin_amdgcn_s_setprio(0); // without the register prefetch the staging waits for its own loads: keep its rank
		__builtin_amdgcn_wave_barrier();
		if ((nvalid == 64 || tiled) && can_prefetch) {
			store_round(0, badacc);
			if (wb_next < n_wb && is_full(wb_next)) load_round(wb_next, 0);
		} else if (nvalid == 64 || tiled) {
			for (uint32_t c0 = 0; c0 < nchunk; c0 += kPref) {
				load_round(wb, c0);
				store_round(c0, badacc);
			}
		} else {
			const unsigned char* src = a.slots + slot0 * stride;
			const uint32_t bytes = nvalid * stride;
			for (uint32_t off = lane * 16u; off < bytes; off += 1024u)
				for (uint32_t o = off; o < bytes && o < off + 16u; o += 4)
					*reinterpret_cast<uint32_t*>(wdata + o) = decode4(*reinterpret_cast<const uint32_t*>(src + o), badacc);
		}
		__builtin_amdgcn_wave_barrier();
		if (tiled && (uint32_t)lane >= nvalid) badacc = 0; // (a lane stages its own read there: what lies behind the batch's last read is nobody's)
		const bool wave_dirty = __builtin_amdgcn_readfirstlane(ballot(badacc != 0u) != 0 ? 1 : 0) != 0;
		if (can_prefetch) __builtin_amdgcn_s_setprio(1);

		// ---- per-lane read geometry ----
		uint32_t len = a.read_len, wlim = a.read_len;
		const bool in_batch = (uint32_t)lane < nvalid;
		if (a.meta != nullptr && in_batch) {
			const uint32_t m = a.meta[slot0 + lane];
			len = m & 0xffffu;
			wlim = m >> 16;
		}
		// ---- every k of the list over the staged batch (ntRead's loop over kList, ntcard.cpp:147-158) ----
		for (uint32_t ki = 0; ki < n_k; ++ki) {
		k = a.ks[ki].k;
		t1 = t1_base + t1_off[ki];
		tabHb = reinterpret_cast<const unsigned char*>(tabH[ki]);
		sketch_k = a.ks[ki].sketch;
		key_base = a.ks[ki].key_base;
		shb = (0u - k) & 3u;
		uint64_t f1_wave = 0;
		int32_t endq = (int32_t)(len < wlim + k - 1 ? len : wlim + k - 1); // steps q in [0,endq)
		if (!in_batch || len < k) endq = 0;
		int32_t maxq = endq, minq = endq;
		for (int o = 32; o > 0; o >>= 1) {
			const int32_t omax = __shfl_xor(maxq, o), omin = __shfl_xor(minq, o);
			maxq = omax > maxq ? omax : maxq;
			minq = omin < minq ? omin : minq;
		}
		maxq = __builtin_amdgcn_readfirstlane(maxq);
		minq = __builtin_amdgcn_readfirstlane(minq);
		// wave classes: CLEAN (equal lengths, no dirty byte), DIRTY (equal lengths, some non-ACGTU byte),
		// RAGGED (lanes end at different steps, e.g. the last partial batch or a ragged host batch)
		constexpr int CLEAN = 0, DIRTY = 1, RAGGED = 2;
		const int wclass = minq != maxq ? RAGGED : (wave_dirty ? DIRTY : CLEAN);

		// The walk starts from the H halves of the hash of k virtual 'A's and feeds 'A' as the outgoing
		// base of the first k steps, so ONE step body serves window filling and steady state.
		uint32_t fHd = a.ks[ki].init_f ^ flipc, rHd = a.ks[ki].init_r ^ flipc;
		int32_t nextok = endq > 0 ? (int32_t)k - 1 : 0x7fffffff; // emission allowed from this step on
		uint32_t hmask = 0;                                      // sampled steps of the current 32-step block

		const unsigned char* const tabGb = reinterpret_cast<const unsigned char*>(tabG);
		auto roll2 = [&](const uint2 t, const uint2 g) { // spaced seed, rolling form: two more terms per strand
			fHd = alignbit(fHd, dbl(fHd), 31) ^ t.x ^ g.x;
			const uint32_t xh = rHd ^ t.y ^ g.y;
			rHd = alignbit(xh >> 1, xh, 1);
		};
		auto roll = [&](const uint2 t) {
			fHd = alignbit(fHd, dbl(fHd), 31) ^ t.x;          // rotl31 in the (H<<1)|H[30] layout, then ^ Tf
			const uint32_t xh = rHd ^ t.y;                    // reverse strand: ^ Tr, then rotr31
			rHd = alignbit(xh >> 1, xh, 1);
		};
		// ---- resolve: 64 (lane, step) pairs at a time, recompute the full hashes from the bases ----
		auto resolve_round = [&](uint32_t e, uint32_t count) {
			// entry -> window start in LDS (any lane's slot), then the closed form over k bases
			const bool act = (uint32_t)lane < count;
			const uint32_t src_lane = e >> 16, q = e & 0xffffu;
			const uint32_t base = act ? src_lane * stride + q + 1u - k : 0u; // byte offset of the window in wdata
			const uint32_t sh = base & 3u;
			const uint32_t* dp = reinterpret_cast<const uint32_t*>(wdata + (base & ~3u));
			uint32_t flo = 0, fhi = 0, rlo = 0, rhi = 0, dirty = 0;
			uint32_t cur = dp[0];
			uint32_t tp = __builtin_amdgcn_readfirstlane(lds_off(t1)); // 256-byte aligned: a pair offset (a<<6 | b<<4) is OR-ed in
			const uint32_t kfull = k & ~3u;
			uint32_t i = 0;
			for (; i < kfull; i += 4) { // 4 window bases = 2 pair lookups per iteration, no branch inside
				const uint32_t nxt = dp[(i >> 2) + 1];
				const uint32_t w = alignbyte(nxt, cur, sh); // 4 code bytes (code<<6) of window positions i..i+3
				cur = nxt;
				dirty |= w; // bit 0 of a byte: not ACGTU
				uint32_t a0, a1;
				pair_addresses(w, tp, a0, a1);
				const v4u32 t0 = lds_read4(a0), t1v = lds_read4(a1);
				flo = xor3(flo, t0.x, t1v.x);
				fhi = xor3(fhi, t0.y, t1v.y);
				rlo = xor3(rlo, t0.z, t1v.z);
				rhi = xor3(rhi, t0.w, t1v.w);
				tp += 512u;
			}
			if (k & 3u) { // 1..3 bases left: a base beyond k contributes nothing because the odd-k table drops the b term
				const uint32_t w = alignbyte(dp[(i >> 2) + 1], cur, sh);
				dirty |= w & (0xffffffffu >> (8 * (4 - (k & 3u))));
				uint32_t a0, a1;
				pair_addresses(w, tp, a0, a1);
				const v4u32 t0 = lds_read4(a0);
				flo ^= t0.x;
				fhi ^= t0.y;
				rlo ^= t0.z;
				rhi ^= t0.w;
				if ((k & 3u) == 3u) {
					const v4u32 t1v = lds_read4(a1);
					flo ^= t1v.x;
					fhi ^= t1v.y;
					rlo ^= t1v.z;
					rhi ^= t1v.w;
				}
			}
			bool hit = false;
			uint32_t key = 0, rel = 0; // rel: counter index inside this k's plane pair; key: index in the engine's whole sketch (hit-log keys)
			if (act && (dirty & 0x01010101u) == 0u) { // a window with a non-ACGTU byte yields no k-mer (ntHashIterator.hpp:59-86)
				const bool rev = (rhi < fhi) | ((rhi == fhi) & (rlo < flo)); // nthash.hpp:275-279
				const uint32_t hi = rev ? rhi : fhi;
				const uint32_t lo = rev ? rlo : flo;
				if constexpr (kDump) {
					const uint32_t win = q + 1u - k; // window start
					const uint64_t row = slot0 + src_lane;
					if (win < a.dump_win) {
						a.dump[row * a.dump_win + win] = ((uint64_t)hi << 32) | lo;
						atomicOr(a.dump_valid + row * ((a.dump_win + 31u) >> 5) + (win >> 5), 1u << (win & 31u));
					}
				} else if constexpr (kMode == 2) {
					// nthll's ntComp (nthll.cpp:92-97): bucket = low bits, value = leading zeros of the rest
					const uint32_t bmask = (1u << a.hll_bits) - 1u;
					const uint32_t lo_rest = lo & ~bmask;
					if ((hi | lo_rest) != 0u) {
						const uint32_t run0 = hi ? (uint32_t)__builtin_clz(hi) : 32u + (uint32_t)__builtin_clz(lo_rest);
						atomicMax(sketch_k + (lo & bmask), run0);
					}
				} else {
					// ntComp (ntcard.cpp:132-145) on the canonical value; sample 1 wins when both match
					const bool c1 = (hi >> (32 - s_bits)) == ((1u << (s_bits - 1)) - 1u);
					const bool c0 = (hi >> (31 - s_bits)) == 1u;
					hit = c0 | c1;
					rel = (lo & rmask) + (c1 ? rbuck : 0u);
					key = key_base + rel;
				}
			}
			if constexpr (kMode != 2 && !kDump) {
				if (use_log)
					log_emit(hit, key); // the increment itself happens later (ntc_apply.hip)
				else if (hit)
					atomicAdd(sketch_k + rel, 1u); // not sketch0 + key: without a log the sketch may hold more than 2^32 counters (32-bit keys would alias)
			}
		};
		// End of a 32-step block: queue the block's sampled steps as (lane, step) pairs.  The queue is ONE register
		// per lane (entry i of the queue lives in lane i); each pass moves one pair per lane with a forward
		// permute through the LDS crossbar (no LDS memory): a lane with a hit sends it to slot npend + rank, the
		// others send a dummy to the remaining slots so that the destinations form a permutation.  When 64 are
		// queued they are resolved; pairs beyond 64 are already sitting in the low lanes of the permute result.
		uint32_t pend = 0, npend = 0;
		auto compact = [&](int32_t last) { // `last` = the step recorded at bit 0 of hmask
			if (can_prefetch) __builtin_amdgcn_s_setprio(3);
			uint32_t cur = hmask;
			hmask = 0;
			uint32_t np = __builtin_amdgcn_readfirstlane(npend); // wave-uniform: the queue fill lives on the scalar unit
			for (;;) {
				const uint64_t m = ballot(cur != 0u);
				if (m == 0) break;
				const uint32_t c = (uint32_t)__popcll(m);
				const bool hit = cur != 0u;
				uint32_t bit;
				asm("v_ffbl_b32 %0, %1" : "=v"(bit) : "v"(cur)); // lowest set bit (all ones for 0: the pair is a dummy then)
				cur &= cur - 1u;
				const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				const uint32_t entry = ((uint32_t)lane << 16) | (((uint32_t)last - bit) & 0xffffu);
				const uint32_t dest = (hit ? pos : c + (uint32_t)lane - pos) + np; // a permutation of 0..63 (mod 64)
				const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)entry);
				pend = (uint32_t)lane >= np ? recv : pend;
				np += c;
				if (np >= 64u) {
					resolve_round(pend, 64u);
					pend = recv;
					np -= 64u;
				}
			}
			npend = np;
			if (can_prefetch) __builtin_amdgcn_s_setprio(2);
		};
		// Table offsets (one byte per base: in<<6 | out<<4) of the 4 steps of group q0.
		//   FILL : every step has q < k     -> outgoing base is the virtual 'A' (code 0)
		//   MIXED: the group straddles k    -> mask the steps with q < k
		//   MAIN : every step has q >= k
		constexpr int FILL = 0, MIXED = 1, MAIN = 2;
		auto group_idx = [&](auto kind, int32_t q0, uint32_t& ain) -> uint32_t {
			ain = *reinterpret_cast<const uint32_t*>(mine + q0);
			if (kind.value == FILL) return ain & 0xc0c0c0c0u;
			uint32_t aout = 0;
			if (kind.value == MAIN || q0 + 3 >= (int32_t)k) {
				const uint32_t* p = reinterpret_cast<const uint32_t*>(mine + ((q0 - (int32_t)k) & ~3));
				// one funnel shift aligns the outgoing bytes to the group AND moves their codes from bits 7:6 to 5:4
				// (what spills over from the neighbouring bytes lands in bits 7:6 / 3:0..: 7:6 are replaced below, the
				// low nibble of a code byte holds nothing but its mark in bit 0, which is shifted out of the byte)
				aout = alignbit(p[1], p[0], 8u * shb + 2u);
				if (kind.value == MIXED && q0 < (int32_t)k) aout &= 0xffffffffu << (8 * ((int32_t)k - q0));
			}
			// bits 7:6 of every byte from ain (incoming code), bits 5:4 from the shifted outgoing code: one v_bfi_b32
			constexpr uint32_t M = 0xc0c0c0c0u;
			uint32_t idx4 = (ain & M) | (aout & ~M);
			asm volatile("" : "+v"(idx4)); // keep it ONE v_bfi_b32: stops hipcc from re-deriving byte 0 with three more ops
			return idx4;
		};
		struct Tab4 {
			uint2 t[4];
		};
		auto issue = [&](uint32_t idx4, Tab4& T) {
			T.t[0] = *reinterpret_cast<const uint2*>(tabHb + (idx4 & 0xffu));
			T.t[1] = *reinterpret_cast<const uint2*>(tabHb + ((idx4 >> 8) & 0xffu));
			T.t[2] = *reinterpret_cast<const uint2*>(tabHb + ((idx4 >> 16) & 0xffu));
			T.t[3] = *reinterpret_cast<const uint2*>(tabHb + (idx4 >> 24));
		};

		// group ranges: [0,e0) never emit and never see a real outgoing base; [e0,e1) are the (at most
		// two) groups around steps k-1 and k; [e1, full_groups) are steady state
		const int32_t full_groups = maxq >> 2;
		const int32_t gk = ((int32_t)k - 1) >> 2; // group of step k-1
		const int32_t gm = ((int32_t)k + 3) >> 2; // first group with q0 >= k
		const int32_t e0 = gk < full_groups ? gk : full_groups;
		const int32_t e1 = gm < full_groups ? gm : full_groups;

		// Sampled steps are recorded without a branch: every step from the first non-FILL group on shifts
		// the lane's mask left and shifts the wave's "sampled" condition in as carry (one v_addc_co_u32);
		// after 32 steps the mask is compacted into the resolve queue.  The block's last step sits at bit 0.
		auto push = [&](uint64_t m) { asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(hmask), "+s"(m)); };
		const int32_t qs = e0 << 2; // first step that is recorded
		uint32_t f1_lane = 0;       // DIRTY / RAGGED: clean windows of this lane (accumulated on the rare path)

		auto walk = [&](auto wc, auto gapped, auto hll) {
			// a dirty byte at step q closes the current run of clean windows [nextok, q) and reopens at q+k
			auto on_mark = [&](int32_t q) {
				if (nextok != 0x7fffffff) {
					if (q > nextok) f1_lane += (uint32_t)(q - nextok);
					nextok = q + (int32_t)k;
				}
			};
			auto on_end = [&](int32_t q) { // RAGGED: the lane's last step was q-1
				if (q >= endq && nextok != 0x7fffffff) {
					if (endq > nextok) f1_lane += (uint32_t)(endq - nextok);
					nextok = 0x7fffffff;
				}
			};
			auto record = [&](int32_t q, bool emitting) {
				uint64_t m = 0;
				if (emitting) {
					uint32_t fs = fHd, rs = rHd;
					if (gapped.value && wc.value == RAGGED) {
						// NTMSM64 (nthash.hpp:641-646,665-670): XOR the don't-care bases' rotated seeds back out
						const unsigned char* gp = mine + (q - (int32_t)k + 1 + (int32_t)a.gap_first);
						for (uint32_t p = 0; p < ngp; ++p) {
							const uint32_t off = (gp[2 * p] & 0xc0u) | ((gp[2 * p + 1] >> 2) & 0x30u);
							const uint2 g = *reinterpret_cast<const uint2*>(gapT + p * 256u + off);
							fs ^= g.x;
							rs ^= g.y;
						}
					}
					const uint32_t mn = fs < rs ? fs : rs; // top bits of min(fh,rh) (of the spaced-seed values when gapped)
					if (kDump)
						m = ballot(true);
					else if (hll.value)
						m = ballot(mn < hll_thr); // nthll: only a hash with enough leading zeros can raise a register
					else
						m = ballot(mn < lo0) | ballot((int32_t)mn >= lo1); // two v_cmp + s_or_b64 (mn carries the flipped bit)
					if (wc.value == RAGGED) m &= ballot(nextok <= q); // DIRTY: windows over a dirty byte are dropped by the resolve stage
				}
				push(m);
			};
			auto run = [&](auto kind, int32_t g0, int32_t g1) {
				for (int32_t g = g0; g < g1; ++g) {
					const int32_t q0 = g << 2;
					uint32_t ain;
					Tab4 T, TG;
					issue(group_idx(kind, q0, ain), T);
					constexpr bool kGapRoll = gapped.value && wc.value != RAGGED; // equal lengths: the spaced value itself rolls
					if (kGapRoll) {
						// the bases leaving / entering the don't-care block during the 4 steps of this group
						const int32_t o1 = q0 - (int32_t)k + (int32_t)a.gap_first, o2 = o1 + (int32_t)a.gap;
						const uint32_t* p1 = reinterpret_cast<const uint32_t*>(mine + (o1 & ~3));
						const uint32_t* p2 = reinterpret_cast<const uint32_t*>(mine + (o2 & ~3));
						const uint32_t w1 = alignbyte(p1[1], p1[0], (uint32_t)o1 & 3u);
						const uint32_t w2 = alignbit(p2[1], p2[0], 8u * ((uint32_t)o2 & 3u) + 2u); // aligned and >> 2 in one funnel shift
						uint32_t ig = (w1 & 0xc0c0c0c0u) | (w2 & 0x3f3f3f3fu);
						asm volatile("" : "+v"(ig));
						TG.t[0] = *reinterpret_cast<const uint2*>(tabGb + (ig & 0xffu));
						TG.t[1] = *reinterpret_cast<const uint2*>(tabGb + ((ig >> 8) & 0xffu));
						TG.t[2] = *reinterpret_cast<const uint2*>(tabGb + ((ig >> 16) & 0xffu));
						TG.t[3] = *reinterpret_cast<const uint2*>(tabGb + (ig >> 24));
					}
					auto step = [&](int b) {
						if (kGapRoll)
							roll2(T.t[b], TG.t[b]);
						else
							roll(T.t[b]);
					};
					uint64_t fixm = 0;
					if (wc.value == DIRTY) fixm = ballot((ain & 0x01010101u) != 0u);
					if (wc.value == RAGGED) // lanes that are shut off never trigger the extra work
						fixm = ballot((((ain & 0x01010101u) != 0u) | (endq < q0 + 4)) & (nextok != 0x7fffffff));
					uint32_t fix = (uint32_t)(fixm | (fixm >> 32));
					asm volatile("" : "+s"(fix)); // opaque SGPR: the test below must stay a scalar branch
					if (wc.value == DIRTY) {
						// the marks only feed F1 here (dirty windows are dropped by the resolve stage), so they are
						// booked in one rare divergent region ahead of the group and the steps stay on the fast path
						if (fix) {
							uint32_t mk = ain & 0x01010101u;
							while (mk != 0u) {
								on_mark(q0 + (int32_t)((uint32_t)__builtin_ctz(mk) >> 3));
								mk &= mk - 1u;
							}
						}
#pragma unroll
						for (int b = 0; b < 4; ++b) {
							step(b);
							if (kind.value != FILL) record(q0 + b, kind.value == MAIN || q0 + b >= (int32_t)k - 1);
						}
					} else if (wc.value == RAGGED && fix) {
						// two copies of the group body behind ONE scalar branch: the common (no dirty byte, no read
						// end in this group) copy carries no per-lane bookkeeping at all
#pragma unroll
						for (int b = 0; b < 4; ++b) {
							on_end(q0 + b);
							if ((ain >> (8 * b)) & 1u) on_mark(q0 + b);
							step(b);
							if (kind.value != FILL) record(q0 + b, kind.value == MAIN || q0 + b >= (int32_t)k - 1);
						}
					} else {
#pragma unroll
						for (int b = 0; b < 4; ++b) {
							step(b);
							if (kind.value != FILL) record(q0 + b, kind.value == MAIN || q0 + b >= (int32_t)k - 1);
						}
					}
				}
			};
			auto single = [&](int32_t q) { // one step outside the 4-step groups (q >= k - 1 only on the closed-form path)
				const uint32_t ain = mine[q];
				if (wc.value != CLEAN) {
					if (wc.value == RAGGED) on_end(q);
					if (ain & 1u) on_mark(q);
				}
				const uint32_t off = (ain & 0xc0u) | (q >= (int32_t)k ? ((mine[q - (int32_t)k] >> 2) & 0x30u) : 0u);
				if (gapped.value && wc.value != RAGGED) { // only reached with q >= k (closed-form start)
					const int32_t o1 = q - (int32_t)k + (int32_t)a.gap_first;
					const uint32_t og = (mine[o1] & 0xc0u) | ((mine[o1 + (int32_t)a.gap] >> 2) & 0x30u);
					roll2(*reinterpret_cast<const uint2*>(tabHb + off), *reinterpret_cast<const uint2*>(tabGb + og));
				} else {
					roll(*reinterpret_cast<const uint2*>(tabHb + off));
				}
				record(q, q >= (int32_t)k - 1);
			};
			if (wc.value != RAGGED) {
				// Equal-length waves skip the k-1 window-filling steps: the H halves of window 0 come from the closed
				// form over the first k bases (the resolve stage's pair table, 2 bases per lookup), which costs about
				// half of rolling them in and far less for large k.  Steps k .. A-1 (A = k rounded up to a group
				// boundary) run singly, the rest in 4-step groups; every step from k-1 on is recorded.
				if (maxq >= (int32_t)k) {
					uint32_t fhi = 0, rhi = 0;
					const uint32_t* dp = reinterpret_cast<const uint32_t*>(mine);
					uint32_t tp = __builtin_amdgcn_readfirstlane(lds_off(t1));
					const uint32_t kfull = k & ~3u;
					auto book_marks = [&](uint32_t i, uint32_t w) { // marks among the first k bases only feed F1 (rare divergent region)
						uint32_t mk = w & 0x01010101u;
						if (ballot(mk != 0u) != 0)
							while (mk != 0u) {
								on_mark((int32_t)i + (int32_t)((uint32_t)__builtin_ctz(mk) >> 3));
								mk &= mk - 1u;
							}
					};
					uint32_t i = 0;
					for (; i < kfull; i += 4) {
						const uint32_t w = dp[i >> 2];
						if (wc.value == DIRTY) book_marks(i, w);
						uint32_t a0, a1;
						pair_addresses(w, tp, a0, a1);
						const v4u32 t0 = lds_read4(a0), t1v = lds_read4(a1);
						fhi = xor3(fhi, t0.y, t1v.y);
						rhi = xor3(rhi, t0.w, t1v.w);
						tp += 512u;
					}
					if (k & 3u) {
						const uint32_t w = dp[i >> 2];
						if (wc.value == DIRTY) book_marks(i, w & (0xffffffffu >> (8 * (4 - (k & 3u)))));
						uint32_t a0, a1;
						pair_addresses(w, tp, a0, a1);
						const v4u32 t0 = lds_read4(a0);
						fhi ^= t0.y;
						rhi ^= t0.w;
						if ((k & 3u) == 3u) {
							const v4u32 t1v = lds_read4(a1);
							fhi ^= t1v.y;
							rhi ^= t1v.w;
						}
					}
					// high word of the 64-bit hash = (H << 1) | L[32]  ->  walk layout (H << 1) | H[30], plus the sample-bit flip
					fHd = ((fhi & ~1u) | (fhi >> 31)) ^ flipc;
					rHd = ((rhi & ~1u) | (rhi >> 31)) ^ flipc;
					record((int32_t)k - 1, true);
					const int32_t A = ((int32_t)k + 3) & ~3;
					for (int32_t q = (int32_t)k; q < (A < maxq ? A : maxq); ++q)
						single(q);
					if (A >= maxq) {
						compact(maxq - 1);
					} else {
						int32_t g = A >> 2, room = 7; // the first block also holds the (at most 4) steps recorded above
						for (;;) {
							const int32_t ge = g + room < full_groups ? g + room : full_groups;
							run(std::integral_constant<int, MAIN>{}, g, ge);
							if (ge == full_groups) {
								if (ge - g == room && (maxq & 3) != 0) compact((ge << 2) - 1); // block is full: the tail gets its own
								for (int32_t q = full_groups << 2; q < maxq; ++q) // partial last group
									single(q);
								compact(maxq - 1);
								break;
							}
							compact((ge << 2) - 1);
							g = ge;
							room = 8;
						}
					}
				}
			} else {
				run(std::integral_constant<int, FILL>{}, 0, e0);
				// recorded steps start at qs = 4*e0 and are handled in blocks of 32 (8 groups): walk, then compact
				const int32_t nblk = (maxq - qs + 31) >> 5;
				for (int32_t blk = 0; blk < nblk; ++blk) {
					const int32_t gb = e0 + (blk << 3);
					const int32_t ge = gb + 8 < full_groups ? gb + 8 : full_groups;
					run(std::integral_constant<int, MIXED>{}, gb, ge < e1 ? ge : e1); // only the first block has MIXED groups
					run(std::integral_constant<int, MAIN>{}, gb > e1 ? gb : e1, ge);
					int32_t last = (ge << 2) - 1;
					if (blk == nblk - 1) {
						for (int32_t q = full_groups << 2; q < maxq; ++q) // partial last group
							single(q);
						last = maxq - 1;
					}
					compact(last);
				}
			}
			if (wc.value == CLEAN) {
				if (maxq >= (int32_t)k) f1_wave += (uint64_t)__popcll(ballot(true)) * (uint32_t)(maxq - (int32_t)k + 1);
			} else {
				if (nextok != 0x7fffffff && endq > nextok) f1_lane += (uint32_t)(endq - nextok);
				uint32_t v = f1_lane;
				for (int o = 32; o > 0; o >>= 1)
					v += __shfl_xor(v, o);
				f1_wave += __builtin_amdgcn_readfirstlane(v);
			}
		};
		if constexpr (kMode == 2) {
			if (wclass == CLEAN)
				walk(std::integral_constant<int, CLEAN>{}, std::false_type{}, std::true_type{});
			else if (wclass == DIRTY)
				walk(std::integral_constant<int, DIRTY>{}, std::false_type{}, std::true_type{});
			else
				walk(std::integral_constant<int, RAGGED>{}, std::false_type{}, std::true_type{});
		} else if constexpr (kMode == 1) {
			if (wclass == CLEAN)
				walk(std::integral_constant<int, CLEAN>{}, std::true_type{}, std::false_type{});
			else if (wclass == DIRTY)
				walk(std::integral_constant<int, DIRTY>{}, std::true_type{}, std::false_type{});
			else
				walk(std::integral_constant<int, RAGGED>{}, std::true_type{}, std::false_type{});
		} else {
			if (wclass == CLEAN)
				walk(std::integral_constant<int, CLEAN>{}, std::false_type{}, std::false_type{});
			else if (wclass == DIRTY)
				walk(std::integral_constant<int, DIRTY>{}, std::false_type{}, std::false_type{});
			else
				walk(std::integral_constant<int, RAGGED>{}, std::false_type{}, std::false_type{});
		}

		// ---- leftovers of the compaction queue: one last, partially filled resolve round ----
		if (npend != 0) resolve_round(pend, npend);
		f1_acc[ki] += f1_wave;
		} // k list
	}
	for (uint32_t j = 0; j < n_k; ++j)
		if (lane == 0 && f1_acc[j]) atomicAdd(a.ks[j].f1, (unsigned long long)f1_acc[j]);
	if (use_log && lane == 0 && lreg < a.log_regions) a.log_fill[lreg] = lfill;
#ifdef NTC_HF_CLOCKS
	if (lane == 0 && gwave < 8192u) {
		g_hf_clocks[2 * gwave] = hc_t0;
		g_hf_clocks[2 * gwave + 1] = __builtin_amdgcn_s_memrealtime();
	}
#endif
}

hipError_t launch_sketch_hf(const HfArgs& a, unsigned grid, unsigned waves_per_block, size_t smem, hipStream_t st)
{
	const dim3 g(grid), b(64u * waves_per_block);
	const bool deep = sketch_hf_deep_prefetch(a.stride) && waves_per_block <= 12; // plain k-mer mode only
	if (a.tiled != 0u && (a.dump != nullptr || a.gap != 0 || a.hll_bits != 0)) return hipErrorInvalidValue; // (tiled staging: plain k-mer mode only)
	if (a.dump != nullptr && a.gap != 0)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 1, 10, true>), g, b, smem, st, a);
	else if (a.dump != nullptr)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 0, 10, true>), g, b, smem, st, a);
	else if (a.hll_bits != 0)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 2, 10>), g, b, smem, st, a);
	else if (a.gap != 0)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 1, 10>), g, b, smem, st, a);
	else if (a.tiled != 0u && a.n_k > 1 && deep)
		hipLaunchKernelGGL((sketch_hf_kernel<true, 0, 16, false, true>), g, b, smem, st, a);
	else if (a.tiled != 0u && a.n_k > 1)
		hipLaunchKernelGGL((sketch_hf_kernel<true, 0, 10, false, true>), g, b, smem, st, a);
	else if (a.tiled != 0u && deep)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 0, 16, false, true>), g, b, smem, st, a);
	else if (a.tiled != 0u)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 0, 10, false, true>), g, b, smem, st, a);
	else if (a.n_k > 1 && deep)
		hipLaunchKernelGGL((sketch_hf_kernel<true, 0, 16>), g, b, smem, st, a);
	else if (a.n_k > 1)
		hipLaunchKernelGGL((sketch_hf_kernel<true, 0, 10>), g, b, smem, st, a);
	else if (deep)
		hipLaunchKernelGGL((sketch_hf_kernel<false, 0, 16>), g, b, smem, st, a);
	else
		hipLaunchKernelGGL((sketch_hf_kernel<false, 0, 10>), g, b, smem, st, a);
	return hipGetLastError();
}

// validation: per read, the hashes of the valid windows in window order (the layout ntc_hash_dump_device documents)
__global__ __launch_bounds__(256) void compact_dump_kernel(const uint64_t* __restrict__ full, const uint32_t* __restrict__ valid, uint64_t n_reads,
                                                           uint32_t n_win, uint32_t max_win, uint64_t* __restrict__ out, uint32_t* __restrict__ count)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const uint32_t vw = (n_win + 31u) >> 5;
	uint32_t n = 0;
	for (uint32_t w = 0; w < n_win; ++w)
		if ((valid[r * vw + (w >> 5)] >> (w & 31u)) & 1u) {
			if (n < max_win) out[r * max_win + n] = full[r * n_win + w];
			++n;
		}
	count[r] = n;
}
hipError_t launch_compact_dump(const uint64_t* full, const uint32_t* valid, uint64_t n_reads, uint32_t n_win, uint32_t max_win, uint64_t* out,
                               uint32_t* count, hipStream_t st)
{
	hipLaunchKernelGGL(compact_dump_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, full, valid, n_reads, n_win, max_win, out, count);
	return hipGetLastError();
}

// slots of 164..256 B: a 16-chunk register prefetch covers them (the 10-chunk one would fall back to blocking loads)
bool sketch_hf_deep_prefetch(uint32_t stride) { return 64u * stride > 10u * 1024u && 64u * stride <= 16u * 1024u; }

hipError_t set_sketch_hf_smem_limit(size_t smem)
{
	const void* fns[] = { reinterpret_cast<const void*>(&sketch_hf_kernel<false, 0, 10>), reinterpret_cast<const void*>(&sketch_hf_kernel<true, 0, 10>),
		              reinterpret_cast<const void*>(&sketch_hf_kernel<false, 0, 16>), reinterpret_cast<const void*>(&sketch_hf_kernel<true, 0, 16>),
		              reinterpret_cast<const void*>(&sketch_hf_kernel<false, 1, 10>), reinterpret_cast<const void*>(&sketch_hf_kernel<false, 2, 10>),
		              reinterpret_cast<const void*>(&sketch_hf_kernel<false, 0, 10, true>), reinterpret_cast<const void*>(&sketch_hf_kernel<false, 1, 10, true>),
		              reinterpret_cast<const void*>(&sketch_hf_kernel<false, 0, 10, false, true>), reinterpret_cast<const void*>(&sketch_hf_kernel<true, 0, 10, false, true>),
		              reinterpret_cast<const void*>(&sketch_hf_kernel<false, 0, 16, false, true>), reinterpret_cast<const void*>(&sketch_hf_kernel<true, 0, 16, false, true>) };
	for (const void* f : fns) {
		const hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (rc != hipSuccess) return rc;
	}
	return hipSuccess;
}

} // namespace ntc
