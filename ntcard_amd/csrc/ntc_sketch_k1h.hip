// ntc_sketch_k1h.hip — K1h + K1f: the round-4 kernel pair for equal-length batches in the TILED slot layout on gfx950.
//
// What the pair computes is ntRead + ntComp (ntcard.cpp:132-158) for one k of 12 .. 32: for every window of k consecutive
// ACGTU bases the canonical ntHash (nthash.hpp:242-257,275-279), ntComp's two sampling patterns on its top bits, one increment
// of t_Counter[sample][hash & (rBuck - 1)] per sampled window (as a hit-log entry, ntc_apply.hip), and F1.
//
// K1h (sketch_k1h_kernel): ONE wave per 2048-read tile, eight waves per CU (two on every SIMD), no wave ever waits for another.  Its body is a
// generated assembly string (gen_k1h.py: explicit physical registers — 62 bit-sliced state planes, 96 base planes, 32 registers
// of loads in flight, exactly 255 VGPRs; the same instruction list runs on the CPU in tests/test_k1h_emulator.py).  C++ here only
// stages the closed-form table in LDS and hands the kernel-argument pointer to the asm statement.  K1h resolves every sampled
// window it is SURE about; it drops every candidate of a (read, block) whose three chunks hold a 16-byte piece with a
// non-ACGTU byte, and every window whose two strands tie on the top 8 bits, and leaves one bit per such (read, chunk) /
// (read, block) in two arrays at fixed positions.
//
// K1f (k1h_fixup_kernel): ntHashIterator's semantics (ntHashIterator.hpp:59-86: a window with a non-ACGTU byte yields nothing)
// from the raw bytes for exactly those (read, block) pairs: validity of every window that ends in the block, F1 taken back for
// the invalid ones, canonical 64-bit hash -> ntComp -> increment for the valid ones (dirty blocks) or for the windows both
// strands flag (tie blocks).  A lane walks one dirty piece's blocks with the plain rolling recurrence (nthash.hpp:242-257).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "ntc_kernels.hpp"
#include "ntc_tile_bits.hpp"

namespace ntc {

namespace {

#include "ntc_k1h_gen_defs.inc"

constexpr uint32_t kK1hWaves = K1H_GEN_WAVES;
constexpr uint32_t kK1hWArea = K1H_GEN_WAREA;
constexpr uint32_t kK1hTableOff = kK1hWaves * kK1hWArea;
constexpr uint32_t kK1hMinBlocks = 2;
constexpr uint32_t k1h_table_bytes(uint32_t k) { return 2u * ((k + 2u) / 3u) * 256u; }
constexpr uint32_t k1h_lds_bytes(uint32_t k) { return kK1hTableOff + k1h_table_bytes(k); } // the wave areas and the table (ntc_sketch_k1h_body.hip)

} // namespace

// the kernels live in ntc_sketch_k1h_body.hip, one object per part
#define K1H_DECL_PART(p)                                                                                                                      \
	hipError_t k1h_launch_part##p(uint32_t k, uint32_t gap, bool sb7, unsigned grid, uint32_t lds, hipStream_t st, const K1hMulti& b, bool* found); \
	hipError_t k1h_set_smem_part##p();
K1H_DECL_PART(0) K1H_DECL_PART(1) K1H_DECL_PART(2) K1H_DECL_PART(3)
static_assert(K1H_GEN_PARTS == 4, "one declaration / call per part");

// ---- K1f ------------------------------------------------------------------------------------------------------------------------
// What K1h leaves behind: dirty[tile][chunk][lane] (bit m: the 16-byte piece of read 64 m + lane holds a byte that is no ACGTU letter),
// tie[tile][block][lane] (bit m: some window of that block of that read has both strands flagged), and per wave a list of SUSPECTS —
// candidates it resolved (counter index known) whose block touches a dirty piece of the read.  A block is "dirty-affected" for a read
// when one of its chunks b - 2 .. b is dirty; block b = the window ends [16 b - 16 + phi, 16 b + phi), phi = (k - 1) mod 16.
//   k1h_f1_kernel     one thread per dirty word: the exact byte masks of the piece and (only where the dirty bits say so) of the two
//                     behind it -> the windows whose RIGHTMOST non-base byte lies in the piece leave F1 (ntHashIterator.hpp:59-86);
//                     a byte of the reference table's slots 1, 3, 4, 5, 7 (nthash.hpp:32: bases to the reference, not letters to K1h)
//                     sends the whole launch down the slow path
//   k1h_fixup_kernel  fast path: a suspect counts iff its window holds no non-base byte (<= 3 pieces, read only where dirty);
//                     slow path (a suspect region overflowed, or a table-slot byte turned up): every window of every dirty-affected
//                     block is re-derived from the bytes with the rolling recurrence (nthash.hpp:242-257), suspects are ignored;
//                     always: the windows of tie blocks that both strands flag, by the same walk.
// The slow path's walk: a wave compacts items into an LDS queue and takes 64 at a time, one per lane; the lane's <= 6 raw pieces are staged
// in LDS, then one step per position: the byte's 2-bit code shifts into a 64-bit register (the last 32 bases), a counter tells how many
// bases in a row were letters of the reference's table, and where a window ends its hash is the closed form over four bases per look-up
// (nthash.hpp:220-239; K1c's table, built with the engine's spaced seed if it has one).  Bytes 1, 3, 4, 5, 7 are bases here, with the codes
// and the complement rule of the reference's table (nthash.hpp:16,32).
constexpr uint32_t kFixQCap = 128;
constexpr uint32_t kFixStage = 6 * 16; // bytes per lane: up to 6 pieces (k - 1 + 48 positions from any offset in the first one)

__device__ __forceinline__ uint32_t ballot_rank(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// length of read r of tile t: read_len for a batch of equal-length reads; in a ragged batch (round 5: K1hArgs.tails) the reads are
// 16 (C - 1) + 1 .. 16 C bases long, every tile is sorted longest first, and tails[t][d] = its reads with more than d bases in their last piece
__device__ __forceinline__ uint32_t read_length(const K1hArgs& a, uint32_t t, uint32_t r)
{
	if (a.tails == nullptr) return a.read_len;
	uint32_t tail = 0;
#pragma unroll
	for (uint32_t d = 0; d < 16u; ++d)
		tail += a.tails[(size_t)t * 16u + d] > r;
	return 16u * (a.n_chunks - 1u) + tail;
}

__device__ __forceinline__ tilebits::v4u32 raw_piece(const K1hArgs& a, uint32_t t, uint32_t c, uint32_t r)
{
	return *reinterpret_cast<const tilebits::v4u32*>(a.tiles + (((size_t)t * a.n_chunks + c) * kTileReads + r) * 16u);
}
// some byte of the piece is one of the reference table's slots 1, 3, 4, 5, 7 (nthash.hpp:32: bases to the reference, no letters to K1h)
__device__ __forceinline__ bool has_slot_byte(const tilebits::v4u32 v)
{
	bool any = false;
	const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
	for (int d = 0; d < 4; ++d) {
		const uint32_t z = w[d] & 0xf8f8f8f8u; // a zero byte here = a byte below 8
		if (((z - 0x01010101u) & ~z & 0x80808080u) != 0u) // (rare)
			for (int j = 0; j < 4; ++j) {
				const uint32_t c = (w[d] >> (8 * j)) & 0xffu;
				any |= c == 1u || c == 3u || c == 4u || c == 5u || c == 7u;
			}
	}
	return any;
}

// F1 correction: a wave compacts the dirty pieces of its rows into an LDS queue and takes 64 at a time, one per lane (a scattered 16-byte
// load each, all in flight together)
__device__ __forceinline__ void k1f_f1_role(const K1fItem& item, const uint32_t bx, const uint32_t nbx)
{
	const K1hArgs& a = item.a;
	const uint32_t k = item.k, n_sus_waves = item.n_waves;
	__shared__ uint2 s_q[4][128];
	__shared__ uint32_t s_sum[4], s_slow[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t C = a.n_chunks;
	const uint64_t n_rows = (uint64_t)a.n_tiles * C;
	uint32_t f1_sub = 0;
	bool slow = false;
	// (the suspect regions of THIS batch: those of the workgroups that walked it — a launch over several batches, K1hMulti, leaves the other waves' regions
	// of this batch's list untouched, with whatever an earlier launch left in their counts)
	const uint32_t sus_w0 = a.first_wg * kK1hWaves;
	for (uint32_t i = bx * 256u + tid; i < n_sus_waves; i += nbx * 256u)
		slow |= a.sus_count[sus_w0 + i] == 0xffffffffu; // a K1h wave ran out of room for its suspects
	uint2* const queue = s_q[wv];
	uint32_t qhead = 0, qtail = 0;
	auto take = [&](uint32_t n_items) {
		if (lane < n_items) {
			const uint2 it = queue[(qhead + lane) & 127u];
			const uint32_t t = it.x, r = it.y & 2047u, c = (it.y >> 11) & 0xfffu;
			const tilebits::v4u32 pa = raw_piece(a, t, c, r);
			const uint32_t A = tilebits::inv16(pa);
			slow |= has_slot_byte(pa);
			uint32_t B = 0;
			if ((it.y >> 24) & 1u) B |= tilebits::inv16(raw_piece(a, t, c + 1u, r));
			if ((it.y >> 25) & 1u) B |= tilebits::inv16(raw_piece(a, t, c + 2u, r)) << 16;
			// window w = 16 c - (k - 1) + j ends at base 16 c + j: it holds a byte of A iff some bit of A lies in [j - k + 1, j] (A smeared over
			// k positions), and no later non-letter byte iff j <= 15 + ctz(B) (B covers the 32 bases behind the piece); 0 <= w < W
			uint64_t E = A;
			uint32_t cov = 1;
			while (cov * 2u <= k) {
				E |= E << cov;
				cov *= 2u;
			}
			if (cov < k) E |= E << (k - cov);
			const int tzb = B ? __builtin_ctz(B) : 32;
			const int jmin = max(0, (int)k - 1 - 16 * (int)c);
			const int jmax = min(min(46, 15 + tzb), (int)read_length(a, t, r) - 1 - 16 * (int)c); // (the last window ends at the read's last base)
			if (A != 0u && jmax >= jmin) f1_sub += (uint32_t)__popcll((E >> jmin) & ((2ull << (jmax - jmin)) - 1ull));
		}
		qhead += n_items;
	};
	const uint64_t wave_g = (uint64_t)bx * 4u + wv, n_wv = (uint64_t)nbx * 4u;
	for (uint64_t row4 = wave_g * 4u; row4 < n_rows; row4 += n_wv * 4u) {
		uint32_t w[6];
#pragma unroll
		for (int q = 0; q < 6; ++q)
			w[q] = row4 + (uint64_t)q < n_rows ? a.dirty[(row4 + (uint64_t)q) * 64u + lane] : 0u;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint64_t row = row4 + (uint64_t)q;
			if (row >= n_rows) break;
			uint32_t word = w[q];
			if (__builtin_amdgcn_ballot_w64(word != 0u) == 0) continue;
			const uint32_t t = (uint32_t)(row / C), c = (uint32_t)(row % C);
			const uint32_t d1 = c + 1u < C ? w[q + 1] : 0u, d2 = c + 2u < C ? w[q + 2] : 0u;
			const uint32_t nv = t + 1u == a.n_tiles ? a.nv_last : kTileReads; // slots behind the batch's last read
			const uint32_t groups = nv > lane ? (nv - lane + 63u) >> 6 : 0u;
			word &= groups >= 32u ? 0xffffffffu : (1u << groups) - 1u;
			while (__builtin_amdgcn_ballot_w64(word != 0u) != 0) {
				const bool have = word != 0u;
				const uint32_t m = have ? (uint32_t)__builtin_ctz(word) : 0u;
				word &= word - 1u;
				const uint64_t bm = __builtin_amdgcn_ballot_w64(have);
				if (have) queue[(qtail + tilebits::mbcnt(bm)) & 127u] = make_uint2(t, (64u * m + lane) | (c << 11) | (((d1 >> m) & 1u) << 24) | (((d2 >> m) & 1u) << 25));
				qtail += (uint32_t)__popcll(bm);
				if (qtail - qhead >= 64u) take(64u);
			}
		}
	}
	if (qtail != qhead) take(qtail - qhead);
	for (int o = 32; o > 0; o >>= 1)
		f1_sub += (uint32_t)__shfl_xor((int)f1_sub, o);
	const bool any_slow = __builtin_amdgcn_ballot_w64(slow) != 0;
	if (lane == 0u) {
		s_sum[wv] = f1_sub;
		s_slow[wv] = any_slow ? 1u : 0u;
	}
	__syncthreads();
	if (tid == 0) { // one atomic per block: thousands of them on one address would take longer than the kernel
		const uint32_t sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
		if (sum != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(a.fix_state + 2), (unsigned long long)sum);
		if (s_slow[0] | s_slow[1] | s_slow[2] | s_slow[3]) a.fix_state[0] = a.launch_id; // (every writer writes the same value)
	}
}

// the suspects (near a dirty piece, or with tied strands): the window's hash from its bytes — packed to 2 bits per base, four bases per look-up
// in K1c's closed-form table (nthash.hpp:220-239) — counted iff every byte is a letter.  It runs BESIDE the F1 role (both wait on scattered
// loads; side by side they take little longer than one of them), so it cannot know yet whether the launch must take the slow path (a table-slot
// byte in some dirty piece): it counts, and leaves the counter it incremented in the entry (x = counter index, w = 1) — the slow path takes
// those increments back before it re-derives everything from the bytes.  No LDS, few registers.
// Round 6: "counts" = appends the counter index to the hit log (ntc_apply.hip applies it with everything else: counting commutes, ntcard.cpp:142-143) — the
// 0.3 M scattered device atomics per 10 M reads of dist g were a fifth of K1f's scattered accesses, and with them gone the sketch stays untouched
// between a reset and the first apply.  A wave takes a place for its hits in one of K1f's own log regions with ONE cursor atomic; a full region falls
// back to the atomic (exact) and says so in the engine's sk_dirty word.
__device__ __forceinline__ void k1f_suspect_role(const K1fItem& item, const uint32_t bx, const uint32_t nbx)
{
	const K1hArgs& a = item.a;
	const uint32_t k = item.k, n_sus_waves = item.n_waves;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	uint32_t kreg = 0;
	auto count_hit = [&](bool hit, uint32_t idx) {
		const uint64_t bm = __builtin_amdgcn_ballot_w64(hit);
		if (bm == 0) return;
		if (item.klog_n == 0u) {
			if (hit) {
				atomicAdd(a.sketch0 + idx, 1u);
				if (a.sk_dirty) *a.sk_dirty = 1u;
			}
			return;
		}
		const uint32_t leader = (uint32_t)__builtin_ctzll(bm);
		uint32_t base = 0;
		if (lane == leader) base = atomicAdd(item.klog_fill + kreg, (uint32_t)__popcll(bm));
		base = (uint32_t)__shfl((int)base, (int)leader);
		if (hit) {
			const uint32_t pos = base + ballot_rank(bm);
			if (pos < item.klog_cap) {
				item.klog[(size_t)kreg * item.klog_cap + pos] = idx;
			} else {
				atomicAdd(a.sketch0 + idx, 1u);
				if (a.sk_dirty) *a.sk_dirty = 1u;
			}
		}
	};
	const uint32_t C = a.n_chunks, s_bits = a.s_bits, r_bits = a.r_bits;
	const uint32_t rmask = (1u << r_bits) - 1u;
	const uint4* const t4v = reinterpret_cast<const uint4*>(item.t4);
	const uint32_t sus_w0 = a.first_wg * kK1hWaves; // (the regions of the workgroups that walked this batch)
	for (uint32_t reg = sus_w0 + bx; reg < sus_w0 + n_sus_waves; reg += nbx) { // a K1h wave's region per block (eight waves per CU: ~180 suspects per region and 10 M reads of dist g)
		uint32_t n = a.sus_count[reg];
		if (n == 0xffffffffu) n = 0; // the region overflowed: the launch takes the slow path, which ignores the suspects
		if (item.klog_n) kreg = (reg * 4u + (tid >> 6) + blockIdx.y * 61u) % item.klog_n; // (the K1f waves of a launch spread over K1f's log regions)
		for (uint32_t i0 = 0; i0 < n; i0 += 256u) { // (n is the block's: every wave takes every turn)
		  const uint32_t i = i0 + tid;
		  bool hit = false;
		  uint32_t hit_idx = 0;
		  if (i < n) do {
			uint4* const ep = a.sus + (size_t)reg * a.sus_cap + i;
			const uint4 e = *ep;
			const uint32_t t = e.y, r = e.z & 2047u, w = e.z >> 11;
			const uint32_t c0 = w >> 4, off = w & 15u;
			const uint64_t win = k >= 64u ? ~0ull : (1ull << k) - 1ull;
			// Fast path: the suspect is not a tie (mark 4), so K1h's verdict on the candidate stands (the strand it resolved IS the canonical one, its
			// counter index and pattern test are right) provided the window holds no non-base byte — and for that only the DIRTY pieces of the
			// window need to be looked at; the entry says which (marks 16, 32, 64: the read's dirty bits in the chunks c0, c0 + 1, c0 + 2 — round 5;
			// round 4 fetched the tie word and three dirty words per suspect, four scattered 4-byte reads, to learn the same).  A tie (about one
			// suspect in a hundred) takes the full path below: both strands' hashes from the bytes.
			if ((e.w & 4u) == 0u) {
				if (e.w & 2u) continue; // (s_bits >= 8: the pattern fails below the walk's 8-bit prefix)
				uint64_t inv = 0;
#pragma unroll
				for (uint32_t j = 0; j < 3; ++j)
					if (c0 + j < C && 16u * j < off + k && ((e.w >> (4u + j)) & 1u))
						inv |= (uint64_t)tilebits::inv16(raw_piece(a, t, c0 + j, r)) << (16u * j);
				if (((inv >> off) & win) != 0ull) continue; // a non-letter byte inside the window: nothing (ntHashIterator.hpp:59-86)
				hit = true;
				hit_idx = e.x;
				ep->w = 1u; // (x already is the counter index)
				continue;
			}
			uint32_t pk[3], bad[3];
			uint64_t inv = 0;
#pragma unroll
			for (uint32_t j = 0; j < 3; ++j) { // the window's <= 3 pieces
				pk[j] = bad[j] = 0;
				if (c0 + j < C && 16u * j < off + k) {
					const tilebits::v4u32 v = raw_piece(a, t, c0 + j, r);
					pk[j] = tilebits::pack16(v, bad[j]);
					if (bad[j]) inv |= (uint64_t)tilebits::inv16(v) << (16u * j); // (rare-ish: the exact positions)
				}
			}
			if (((inv >> off) & win) != 0ull) continue; // a non-letter byte inside the window: nothing (ntHashIterator.hpp:59-86)
			const uint32_t lo = tilebits::alignbit(pk[1], pk[0], 2u * off), hi = tilebits::alignbit(pk[2], pk[1], 2u * off);
			uint32_t f0 = 0, f1 = 0, r0 = 0, r1 = 0;
			for (uint32_t g = 0; g < (k + 3u) / 4u; ++g) {
				const uint32_t b = ((g < 4u ? lo : hi) >> (8u * (g & 3u))) & 0xffu;
				const uint4 x = t4v[g * 256u + b];
				f0 ^= x.x;
				f1 ^= x.y;
				r0 ^= x.z;
				r1 ^= x.w;
			}
			const uint64_t fh = ((uint64_t)f1 << 32) | f0, rh = ((uint64_t)r1 << 32) | r0;
			const uint64_t h = rh < fh ? rh : fh;                                                          // nthash.hpp:275-279
			uint32_t smp = 2;                                                                             // ntcard.cpp:132-145
			if ((h >> (63u - s_bits)) == 1ull) smp = 0;
			if ((h >> (64u - s_bits)) == (1ull << (s_bits - 1u)) - 1ull) smp = 1;
			if (smp < 2u) {
				const uint32_t idx = a.key_base + (smp << r_bits) + (uint32_t)(h & (uint64_t)rmask); // (engines with a hit log have < 2^32 counters; K1h itself keys them with 32 bits)
				hit = true;
				hit_idx = idx;
				ep->x = idx;
				ep->w = 1u;
			}
		  } while (false);
		  count_hit(hit, hit_idx);
		}
	}
}

// K1f, first launch: the F1 role and the suspect role side by side (blocks [0, n_f1) and the rest), blockIdx.y = the K1h launch of the batch
__global__ __launch_bounds__(256) void k1h_fix_kernel(const K1fBatch batch, const uint32_t n_f1_blocks)
{
	const K1fItem& item = batch.item[blockIdx.y];
	// (Blocks are handed out in index order, so the F1 role's blocks fill the device first and the suspect role follows as they retire.  Alternating
	// the roles by block parity — both resident from the start — was measured SLOWER, 0.054 against 0.049 ms per 10 M reads of dist g: both roles are
	// bound by the rate of scattered 16-byte fetches, F1 alone 0.029 ms, suspects alone 0.033 ms, and side by side they only disturb each other's
	// locality: profiles/r05_k1f_roles.txt.)
	if (blockIdx.x < n_f1_blocks) k1f_f1_role(item, blockIdx.x, n_f1_blocks);
	else k1f_suspect_role(item, blockIdx.x - n_f1_blocks, gridDim.x - n_f1_blocks);
}

// byte -> bits 0..2: code2 of the base (A 0, C 1, T/U 2, G 3; 4: no base), bits 4..5: the code whose letter-complement is what the
// reference complements this byte to, table[byte & 7] (nthash.hpp:16,32).  For a letter that is its own code; for the table's slots
// 1, 3, 4, 5, 7 — bases to the reference — it is not: their "complement" is the slot's own base.
__device__ __forceinline__ uint32_t base_class(uint32_t c)
{
	auto code = [](uint32_t b) -> uint32_t { // 4: no base
		switch (b) {
		case 'A': case 'a': case 4: case 5: return 0u;
		case 'C': case 'c': case 7: return 1u;
		case 'T': case 't': case 'U': case 'u': case 1: return 2u;
		case 'G': case 'g': case 3: return 3u;
		default: return 4u;
		}
	};
	const uint32_t f = code(c);
	return f | (((code(c & 7u) ^ 2u) & 3u) << 4);
}

__global__ __launch_bounds__(256) void k1h_slow_kernel(const K1fBatch batch)
{
	const K1fItem& item = batch.item[blockIdx.y];
	const K1hArgs& a = item.a;
	const void* const t4 = item.t4;
	const uint32_t k = item.k;
	__shared__ uint2 s_queue[4][kFixQCap];
	__shared__ __align__(16) unsigned char s_stage[4][64 * kFixStage];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
	const uint32_t phi = (k - 1u) & 15u;
	const uint32_t nb = ((a.read_len - 1u + 16u - phi) >> 4) + 1u, C = a.n_chunks;
	const uint64_t n_dirty = (uint64_t)a.n_tiles * C * 64u, n_tie = (uint64_t)a.n_tiles * nb * 64u;
	const uint32_t s_bits = a.s_bits, r_bits = a.r_bits, L = a.read_len;
	const uint32_t rmask = (1u << r_bits) - 1u;
	const uint4* const t4v = reinterpret_cast<const uint4*>(t4);
	uint2* const queue = s_queue[wv];
	unsigned char* const stage = s_stage[wv] + lane * kFixStage;
	uint32_t qhead = 0, qtail = 0; // wave-uniform
	uint32_t f1_sub = 0;
	// The launch takes the slow path only when a suspect region overflowed or a table-slot byte turned up.  Otherwise k1h_fix_kernel has done
	// everything but one subtraction: the F1 correction its F1 role summed up.
	const bool flagged = a.fix_state[0] == a.launch_id;
	if (blockIdx.x == 0 && tid == 0) {
		unsigned long long* fs = reinterpret_cast<unsigned long long*>(a.fix_state + 2);
		const unsigned long long sub = *fs;
		if (!flagged && sub) atomicAdd(a.f1, (unsigned long long)0 - sub);
		*fs = 0ull; // (ready for the next launch; the slow path counts F1 itself)
	}
	if (!flagged) return;
	const bool slow = true;
	if (a.sk_dirty && tid == 0) *a.sk_dirty = 1u; // (the slow path takes increments back and counts with device atomics: the sketch is no longer what the last apply left)
	// what the suspect role counted goes back first: every window of a dirty-affected block is re-derived below, tie windows included
	for (uint32_t reg = a.first_wg * kK1hWaves + blockIdx.x; reg < a.first_wg * kK1hWaves + item.n_waves; reg += gridDim.x) {
		uint32_t n = a.sus_count[reg];
		if (n == 0xffffffffu) n = 0;
		for (uint32_t i = tid; i < n; i += 256u) {
			const uint4 e = a.sus[(size_t)reg * a.sus_cap + i];
			if (e.w == 1u) atomicSub(a.sketch0 + e.x, 1u);
		}
	}

	// ---- walk the 64 (or fewer) items at the head of the queue ----
	auto walk = [&](uint32_t n_items) {
		const bool act = lane < n_items;
		const uint2 it = queue[(qhead + (act ? lane : 0u)) % kFixQCap];
		const uint32_t t = it.x, r = it.y & 2047u, c = (it.y >> 11) & 0x1fffu, nblk = (it.y >> 24) & 3u;
		const bool ties = (it.y >> 26) & 1u;
		int e_lo = 16 * (int)c - 16 + (int)phi, e_hi = 16 * (int)(c + nblk - 1u) + (int)phi - 1;
		e_lo = max(e_lo, (int)k - 1);
		e_hi = min(e_hi, (int)(act ? read_length(a, t, r) : L) - 1);
		const int p0 = e_lo - (int)k + 1;
		const int len = act && e_hi >= e_lo ? e_hi - p0 + 1 : 0; // positions to walk
		// stage the pieces [p0 >> 4, e_hi >> 4] of the read
		const int c0 = p0 >> 4, np = len > 0 ? (e_hi >> 4) - c0 + 1 : 0;
#pragma unroll
		for (int j = 0; j < 6; ++j)
			if (j < np) *reinterpret_cast<tilebits::v4u32*>(stage + 16 * j) = raw_piece(a, t, (uint32_t)(c0 + j), r);
		// (a lane reads back only what it wrote itself: no barrier)
		const uint32_t off0 = (uint32_t)(p0 & 15);
		const int first_end = e_lo - p0; // step index of the first window end
		int max_len = len;
		for (int o = 32; o > 0; o >>= 1)
			max_len = max(max_len, __shfl_xor(max_len, o));
		uint64_t fw = 0, rc = 0; // 2-bit codes of the last 32 bases, oldest lowest: the bases themselves / their complements
		uint32_t good = 0;
		for (int i = 0; i < max_len; ++i) {
			if (i >= len) continue;
			const uint32_t cls = base_class(stage[off0 + (uint32_t)i]);
			if ((cls & 4u) != 0u) { // not a base: the window restarts behind it
				good = 0;
			} else {
				++good;
				fw = (fw >> 2) | ((uint64_t)(cls & 3u) << 62);
				rc = (rc >> 2) | ((uint64_t)((cls >> 4) & 3u) << 62);
			}
			if (i < first_end) continue;
			if (good < k) {
				if (!ties) ++f1_sub;
				continue;
			}
			// the window's k bases, base i at bits 2i+1:2i -> four bases per look-up: fh = XOR srol^(k-1-i) seed(c_i), rh = XOR srol^i comp(c_i)
			const uint64_t wf = fw >> (64u - 2u * k), wr = rc >> (64u - 2u * k);
			uint32_t f0 = 0, f1 = 0, r0 = 0, r1 = 0;
			for (uint32_t g = 0; g < (k + 3u) / 4u; ++g) {
				const uint4 xf = t4v[g * 256u + (uint32_t)((wf >> (8u * g)) & 0xffu)];
				f0 ^= xf.x;
				f1 ^= xf.y;
				const uint4 xr = t4v[g * 256u + (uint32_t)((wr >> (8u * g)) & 0xffu)];
				r0 ^= xr.z;
				r1 ^= xr.w;
			}
			const uint64_t fh = ((uint64_t)f1 << 32) | f0, rh = ((uint64_t)r1 << 32) | r0;
			if (ties) {
				const uint32_t f8 = (uint32_t)(fh >> 56), r8 = (uint32_t)(rh >> 56);
				bool cf, cr;
				if (s_bits == 7u) {
					cf = ((f8 >> 1) == 0x3fu && (r8 >> 1) >= 0x3fu) || (f8 == 1u && r8 >= 1u);
					cr = ((r8 >> 1) == 0x3fu && (f8 >> 1) >= 0x3fu) || (r8 == 1u && f8 >= 1u);
				} else {
					cf = (f8 == 0x7fu && r8 >= 0x7fu) || f8 == 0u;
					cr = (r8 == 0x7fu && f8 >= 0x7fu) || r8 == 0u;
				}
				if (!(cf && cr)) continue;
			}
			const uint64_t h = rh < fh ? rh : fh;                                                          // nthash.hpp:275-279
			uint32_t smp = 2;                                                                             // ntcard.cpp:132-145
			if ((h >> (63u - s_bits)) == 1ull) smp = 0;
			if ((h >> (64u - s_bits)) == (1ull << (s_bits - 1u)) - 1ull) smp = 1;
			if (smp < 2u) atomicAdd(a.sketch0 + (size_t)a.key_base + ((size_t)smp << r_bits) + (size_t)(h & (uint64_t)rmask), 1u);
		}
		qhead += n_items;
	};
	auto push = [&](bool have, uint32_t x, uint32_t y) {
		const uint64_t m = __builtin_amdgcn_ballot_w64(have);
		if (have) queue[(qtail + ballot_rank(m)) % kFixQCap] = make_uint2(x, y);
		qtail += (uint32_t)__popcll(m);
		if (qtail - qhead >= 64u) walk(64u);
	};

	// rows of 64 words: (tile, chunk) of the dirty array, then (tile, block) of the tie array.  A wave takes four consecutive rows
	// per iteration (their loads are in flight together; the two dirty rows behind them say whether a piece is the last dirty one)
	const uint64_t n_drows = n_dirty / 64u, n_rows = (n_dirty + n_tie) / 64u;
	const uint64_t wave_g = (uint64_t)blockIdx.x * 4u + wv, n_wv = (uint64_t)gridDim.x * 4u;
	const uint64_t row_hi = n_rows;
	const uint64_t row_first = slow ? 0u : n_rows; // (fast path: the suspects above were everything)
	for (uint64_t row4 = row_first + wave_g * 4u; row4 < n_rows; row4 += n_wv * 4u) { // (strided: the heavy dirty rows and the light tie rows spread over all waves)
		uint32_t w[6];
#pragma unroll
		for (int q = 0; q < 6; ++q) {
			const uint64_t row = row4 + (uint64_t)q;
			w[q] = 0;
			if (row < n_rows && (q < 4 || row < n_drows)) w[q] = row < n_drows ? a.dirty[row * 64u + lane] : a.tie[(row - n_drows) * 64u + lane];
		}
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint64_t row = row4 + (uint64_t)q;
			if (row >= row_hi) break;
			uint32_t word = w[q];
			if (__builtin_amdgcn_ballot_w64(word != 0u) == 0) continue;
			const bool is_d = row < n_drows;
			uint32_t t, c, d1 = 0, d2 = 0;
			if (is_d) {
				t = (uint32_t)(row / C);
				c = (uint32_t)(row % C);
				if (c + 1u < C) d1 = w[q + 1];
				if (c + 2u < C) d2 = w[q + 2];
				const uint32_t nv = t + 1u == a.n_tiles ? a.nv_last : kTileReads; // slots behind the batch's last read
				const uint32_t groups = nv > lane ? (nv - lane + 63u) >> 6 : 0u;
				word &= groups >= 32u ? 0xffffffffu : (1u << groups) - 1u;
			} else {
				const uint64_t j = row - n_drows;
				t = (uint32_t)(j / nb);
				c = (uint32_t)(j % nb);
				if (slow) // the dirty items of this launch cover every window of the dirty-affected blocks, both-flag windows included
					for (int cc = (int)c - 2; cc <= (int)c; ++cc)
						if (cc >= 0 && (uint32_t)cc < C) word &= ~a.dirty[((size_t)t * C + (uint32_t)cc) * 64u + lane];
			}
			while (__builtin_amdgcn_ballot_w64(word != 0u) != 0) {
				const bool have = word != 0u;
				const uint32_t m = have ? (uint32_t)__builtin_ctz(word) : 0u;
				word &= word - 1u;
				uint32_t nblk = 1;
				if (is_d) { // blocks c .. c + 2 this piece is the last dirty chunk of
					if (!((d1 >> m) & 1u)) nblk = ((d2 >> m) & 1u) ? 2u : 3u;
					if (c + nblk > nb) nblk = nb - c;
				}
				push(have, t, (64u * m + lane) | (c << 11) | (nblk << 24) | (is_d ? 0u : 1u << 26));
			}
		}
	}
	if (qtail != qhead) walk(qtail - qhead);
	for (int o = 32; o > 0; o >>= 1)
		f1_sub += (uint32_t)__shfl_xor((int)f1_sub, o);
	if (lane == 0u && f1_sub != 0u) atomicAdd(a.f1, (unsigned long long)0 - (unsigned long long)f1_sub);
}


// the resolve pass works on ONE 32-bit word per candidate: the low r_bits bits of the hash, the bit that tells the samples apart, and
// (s_bits >= 8) the s_bits - 7 bits between the 8-bit prefix the walk tests and the end of ntComp's patterns
bool sketch_k1h_supports(uint32_t k, uint32_t gap, uint32_t s_bits, uint32_t r_bits)
{
	bool built = false;
#define K1H_IS(kk, gg) built |= k == kk && gap == gg;
	K1H_VARIANTS_ALL(K1H_IS)
#undef K1H_IS
	return built && s_bits >= 7 && s_bits <= 30 && r_bits + 1 + (s_bits - 7) <= 32;
}

uint32_t sketch_k1h_blocks(uint32_t k, uint32_t read_len) { return ((read_len - 1u + 16u - ((k - 1u) & 15u)) >> 4) + 1u; }

// closed-form table of the resolve passes: [strand][group g][64 values] dwords, 3 bases per entry (code2 order: A=0 C=1 T/U=2 G=3, base t
// of the group in bits 2t+1:2t).  Word = low r_bits bits of the strand's term (nthash.hpp:220-239: fh = XOR srol^(k-1-i) seed(c_i),
// rh = XOR srol^i comp(c_i)) | its bit 62 << r_bits: ntComp's patterns differ in that bit, and the counter index is
// key_base + (sample << r_bits) + (hash & (rBuck - 1)) (ntcard.cpp:132-145); s_bits >= 8: + hash bits 55 .. 63 - s_bits above that
void build_k1h_table(uint32_t k, uint32_t gap, uint32_t r_bits, uint32_t s_bits, uint32_t* out)
{
	static const unsigned code_of_code2[4] = { 0, 1, 3, 2 };
	const uint32_t ng = (k + 2) / 3;
	for (uint32_t st = 0; st < 2; ++st)
		for (uint32_t g = 0; g < ng; ++g)
			for (uint32_t v = 0; v < 64; ++v) {
				uint64_t x = 0;
				for (uint32_t t = 0; t < 3; ++t) {
					const uint32_t i = 3 * g + t;
					if (i >= k) break;
					if (gap && i >= (k - gap) / 2 && i < (k - gap) / 2 + gap) continue; // a don't-care position of the spaced seed (ntcard.cpp:407-413)
					const unsigned c = code_of_code2[(v >> (2 * t)) & 3u];
					x ^= st == 0 ? srol(seed_of(c), k - 1 - i) : srol(comp_of(c), i);
				}
				const uint32_t ext = (uint32_t)((x >> (63 - s_bits)) & ((1ull << (s_bits - 7)) - 1ull)); // hash bits 55 .. 63 - s_bits
				out[(st * ng + g) * 64 + v] = (uint32_t)(x & ((1ull << r_bits) - 1ull)) | ((uint32_t)((x >> 62) & 1u) << r_bits) | (s_bits > 7 ? ext << (r_bits + 1) : 0u);
			}
}

hipError_t set_sketch_k1h_smem_limit()
{
	hipError_t rc = k1h_set_smem_part0();
	if (rc == hipSuccess) rc = k1h_set_smem_part1();
	if (rc == hipSuccess) rc = k1h_set_smem_part2();
	if (rc == hipSuccess) rc = k1h_set_smem_part3();
	return rc;
}

// Blocks a wave should have at least: a wave that starts inside a tile walks two masked blocks first (~1.2 blocks' worth of instructions), so a
// small batch is better off with fewer, longer shares — down to this many (measured, ms per step at 0.5 / 1 / 2 M reads of 150 bp: at least
// 4 blocks 0.126 / 0.141 / 0.199, at least 2 blocks 0.100 / 0.142 / 0.196: profiles/r04_batch_size_sweep.txt)
uint32_t sketch_k1h_min_blocks() { return kK1hMinBlocks; }
uint32_t sketch_k1h_waves() { return kK1hWaves; } // waves per workgroup (the engine sizes the suspect regions by it)

// Workgroups and block shares of a launch over n batches: the grid is what ONE batch with all the blocks would get (a wave needs ~4 blocks to be worth its
// start-up), every batch a share of it in proportion to its blocks, at least one workgroup
unsigned plan_sketch_k1h(const K1hArgs* a, uint32_t n, uint32_t k, unsigned cus, K1hArgs* out)
{
	uint64_t total[kK1hSegs], all = 0;
	for (uint32_t i = 0; i < n; ++i) {
		total[i] = (uint64_t)a[i].n_tiles * sketch_k1h_blocks(k, a[i].read_len);
		all += total[i];
	}
	unsigned grid = (unsigned)std::min<uint64_t>((all + kK1hMinBlocks * kK1hWaves - 1) / (kK1hMinBlocks * kK1hWaves), cus);
	grid = std::max<unsigned>(grid, n);
	// largest-remainder apportionment of the workgroups
	unsigned wg[kK1hSegs], given = 0;
	double frac[kK1hSegs];
	for (uint32_t i = 0; i < n; ++i) {
		const double share = all ? (double)grid * (double)total[i] / (double)all : 1.0;
		wg[i] = std::max<unsigned>(1u, (unsigned)share);
		frac[i] = share - (double)wg[i];
		given += wg[i];
	}
	while (given < grid) { // hand the rest to the largest remainders
		uint32_t best = 0;
		for (uint32_t i = 1; i < n; ++i)
			if (frac[i] > frac[best]) best = i;
		++wg[best];
		frac[best] -= 1.0;
		++given;
	}
	while (given > grid) { // (the minimum of one pushed the sum over: take from the batch with the fewest blocks per workgroup)
		uint32_t best = n;
		for (uint32_t i = 0; i < n; ++i)
			if (wg[i] > 1u && (best == n || (double)total[i] / wg[i] < (double)total[best] / wg[best])) best = i;
		if (best == n) break;
		--wg[best];
		--given;
	}
	unsigned first = 0;
	for (uint32_t i = 0; i < n; ++i) {
		out[i] = a[i];
		out[i].first_wg = first;
		out[i].n_wg = wg[i];
		out[i].blocks_per_wave = (uint32_t)((total[i] + (uint64_t)wg[i] * kK1hWaves - 1) / ((uint64_t)wg[i] * kK1hWaves));
		out[i].nb_magic = (uint32_t)((1ull << 32) / sketch_k1h_blocks(k, a[i].read_len));
		first += wg[i];
	}
	return first;
}

hipError_t launch_sketch_k1h_multi(const K1hArgs* a, uint32_t n, uint32_t k, uint32_t gap, unsigned cus, hipStream_t st, K1hArgs* args_out, uint32_t* n_waves)
{
	if (n == 0 || n > kK1hSegs) return hipErrorInvalidValue;
	for (uint32_t i = 0; i < n; ++i)
		if (!sketch_k1h_supports(k, gap, a[i].s_bits, a[i].r_bits) || a[i].s_bits != a[0].s_bits || a[i].table != a[0].table) return hipErrorInvalidValue;
	K1hMulti m;
	std::memset(&m, 0, sizeof m);
	m.n_segs = n;
	const unsigned grid = plan_sketch_k1h(a, n, k, cus, m.seg);
	const uint32_t lds = k1h_lds_bytes(k);
	const bool sb7 = a[0].s_bits == 7;
	bool found = false;
	hipError_t rc = k1h_launch_part0(k, gap, sb7, grid, lds, st, m, &found);
	if (!found) rc = k1h_launch_part1(k, gap, sb7, grid, lds, st, m, &found);
	if (!found) rc = k1h_launch_part2(k, gap, sb7, grid, lds, st, m, &found);
	if (!found) rc = k1h_launch_part3(k, gap, sb7, grid, lds, st, m, &found);
	if (!found) return hipErrorInvalidValue;
	if (rc != hipSuccess) return rc;
	for (uint32_t i = 0; i < n; ++i)
		args_out[i] = m.seg[i];
	*n_waves = grid * kK1hWaves;
	return hipGetLastError();
}

// K1h over one batch on stream st; *args_out = the arguments as launched (block shares filled in), *n_waves = its waves (suspect regions)
hipError_t launch_sketch_k1h(const K1hArgs& a, uint32_t k, uint32_t gap, unsigned cus, hipStream_t st, K1hArgs* args_out, uint32_t* n_waves)
{
	return launch_sketch_k1h_multi(&a, 1, k, gap, cus, st, args_out, n_waves);
}

// K1f for the batches K1h has been launched over (arguments as launched), on any stream ordered behind those launches
hipError_t launch_k1h_fixup(const K1fBatch& b, uint32_t n_items, unsigned cus, hipStream_t st)
{
	if (n_items == 0 || n_items > kK1fBatch) return hipErrorInvalidValue;
	size_t rows = 0;
	uint32_t waves = 0;
	for (uint32_t i = 0; i < n_items; ++i) { // (a block of a launch with fewer rows / regions finds its loops empty)
		rows = std::max(rows, (size_t)b.item[i].a.n_tiles * b.item[i].a.n_chunks);
		waves = std::max(waves, b.item[i].n_waves);
	}
	// F1-role blocks per CU.  Round 6: TWO per CU for a launch over one batch, FOUR for one over several (round 5: eight = every resident block of the chip): with
	// fewer of them the suspect role's blocks are resident beside the F1 role's from the start and the two streams of scattered fetches overlap instead of
	// following each other with a ramp and a tail each — fix-up 0.068 -> 0.050 ms per step behind every launch, 0.049 -> 0.045 deferred; one per CU is too few
	// (0.065).  profiles/r06_k1f_f1_blocks.txt.  (A/B builds override: tools/ab_build.sh <name> -DNTC_AB_F1_SINGLE=n -DNTC_AB_F1_MULTI=n.)
#ifndef NTC_AB_F1_SINGLE
#define NTC_AB_F1_SINGLE 2
#endif
#ifndef NTC_AB_F1_MULTI
#define NTC_AB_F1_MULTI 4
#endif
	const unsigned n_f1 = (unsigned)std::min<size_t>((rows + 15) / 16, (size_t)cus * (n_items == 1 ? NTC_AB_F1_SINGLE : NTC_AB_F1_MULTI));
	unsigned n_sus = waves;
#ifdef NTC_K1F_TIME_ROLE_BUILD // timing builds only (tools/k1h_variant.sh, EXTRA=-DNTC_K1F_TIME_ROLE_BUILD): never in the product library — the results are WRONG
	if (const char* ev = std::getenv("NTC_K1F_TIME_ROLE")) { // 1 = the F1 role alone, 2 = the suspect role alone
		if (ev[0] == '1') n_sus = 0;
		if (ev[0] == '2') {
			hipLaunchKernelGGL(k1h_fix_kernel, dim3(waves, n_items), dim3(256), 0, st, b, 0u);
			return hipGetLastError();
		}
	}
#endif
	hipLaunchKernelGGL(k1h_fix_kernel, dim3(n_f1 + n_sus, n_items), dim3(256), 0, st, b, n_f1);
	// the F1 correction of the fast path; the slow path (LDS, a CU's worth of blocks) only for a flagged launch
	hipLaunchKernelGGL(k1h_slow_kernel, dim3(cus, n_items), dim3(256), 0, st, b);
	return hipGetLastError();
}

} // namespace ntc
