// ntc_sketch_k1h.hip — K1h + K1f: the round-4 kernel pair for equal-length batches in the TILED slot layout on gfx950.
//
// What the pair computes is ntRead + ntComp (ntcard.cpp:132-158) for one k of 12 .. 32: for every window of k consecutive
// ACGTU bases the canonical ntHash (nthash.hpp:242-257,275-279), ntComp's two sampling patterns on its top bits, one increment
// of t_Counter[sample][hash & (rBuck - 1)] per sampled window (as a hit-log entry, ntc_apply.hip), and F1.
//
// K1h (sketch_k1h_kernel): ONE wave per 2048-read tile, six waves per CU, no wave ever waits for another.  Its body is a
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

#include "ntc_k1h_gen.inc"

static_assert(offsetof(K1hArgs, tiles) == 0 && offsetof(K1hArgs, log) == 8 && offsetof(K1hArgs, log_fill) == 16 && offsetof(K1hArgs, sketch0) == 24 &&
                  offsetof(K1hArgs, f1) == 32 && offsetof(K1hArgs, dirty) == 40 && offsetof(K1hArgs, tie) == 48 && offsetof(K1hArgs, n_tiles) == 56 &&
                  offsetof(K1hArgs, n_chunks) == 60 && offsetof(K1hArgs, read_len) == 64 && offsetof(K1hArgs, nv_last) == 68 && offsetof(K1hArgs, key_base) == 72 &&
                  offsetof(K1hArgs, rmask2) == 76 && offsetof(K1hArgs, log_regions) == 80 && offsetof(K1hArgs, log_region_cap) == 84 && offsetof(K1hArgs, table) == 88 &&
                  offsetof(K1hArgs, blocks_per_wave) == 104 && offsetof(K1hArgs, nb_magic) == 108 && offsetof(K1hArgs, sus) == 112 &&
                  offsetof(K1hArgs, sus_count) == 120 && offsetof(K1hArgs, sus_cap) == 128,
              "gen_k1h.KARG");
constexpr uint32_t kK1hWaves = K1H_GEN_WAVES;
constexpr uint32_t kK1hWArea = K1H_GEN_WAREA;
static_assert(kK1hWaves == 6, "the launch (384 threads) and the share-out by SIMD assume six waves per workgroup");
constexpr uint32_t kK1hTableOff = kK1hWaves * kK1hWArea;
constexpr uint32_t k1h_table_bytes(uint32_t k) { return 2u * ((k + 2u) / 3u) * 256u; }
constexpr uint32_t k1h_lds_bytes(uint32_t k) { return kK1hTableOff + k1h_table_bytes(k) + 32u; } // the wave areas, the table, the waves' SIMD numbers

#define K1H_CLOBBERS_V                                                                                                                 \
	"v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20",   \
	    "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",   \
	    "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",   \
	    "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",   \
	    "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",   \
	    "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",  \
	    "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",       \
	    "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145",       \
	    "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161",       \
	    "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177",       \
	    "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193",       \
	    "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",       \
	    "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225",       \
	    "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241",       \
	    "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254"
#define K1H_CLOBBERS_S "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "memory"

} // namespace

template <int K, int SB, int GAP>
__global__ __launch_bounds__(384) void sketch_k1h_kernel(const K1hArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	{ // the closed-form table: [2 strands][ceil(k / 3)][64] dwords behind the six wave areas
		constexpr uint32_t n = 2u * ((K + 2) / 3) * 64u;
		uint32_t* dst = reinterpret_cast<uint32_t*>(smem + kK1hTableOff);
		for (uint32_t i = threadIdx.x; i < n; i += 384u)
			dst[i] = a.table[i];
	}
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// Six waves on four SIMDs: two SIMDs hold a pair, two a single wave.  A lone wave issues an instruction every ~4.4 clocks whatever it is; the
	// waves of a pair take turns for everything that is not a plain bit operation (v_perm, shifts-and-or, compares, multiplies occupy the SIMD for 4 clocks:
	// profiles/r04_ubench_issue.txt), so a pair's waves are slower by a quarter.  Every wave therefore tells the others which SIMD it runs on
	// (HW_ID bits 5:4) and the workgroup's blocks are shared out by weight: lone_weight sixteenths to a lone wave for 16 to one of a pair.
	volatile uint32_t* const simd_of = reinterpret_cast<volatile uint32_t*>(smem + kK1hTableOff + k1h_table_bytes(K));
	if ((threadIdx.x & 63u) == 0u) simd_of[wave] = (uint32_t)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // hwreg(HW_REG_HW_ID, 4, 2)
	__syncthreads();
	uint32_t first_block, end_block;
	{
		uint32_t wsum = 0, wbefore = 0, wmine = 0;
		for (uint32_t i = 0; i < kK1hWaves; ++i) {
			uint32_t same = 0;
			for (uint32_t j = 0; j < kK1hWaves; ++j)
				same += simd_of[j] == simd_of[i];
			const uint32_t wt = same >= 2u ? 16u : a.lone_weight;
			if (i < wave) wbefore += wt;
			if (i == wave) wmine = wt;
			wsum += wt;
		}
		const uint32_t quota = a.blocks_per_wave * kK1hWaves, wg0 = blockIdx.x * quota; // (n_tiles * blocks < 2^32 / 64: the products below fit 64 bits easily)
		first_block = wg0 + (uint32_t)((uint64_t)quota * wbefore / wsum);
		end_block = wg0 + (uint32_t)((uint64_t)quota * (wbefore + wmine) / wsum);
		first_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)first_block);
		end_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)end_block);
	}
	const uint32_t wave_gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kK1hWaves + wave));
	const uint32_t n_waves = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gridDim.x * kK1hWaves));
	const uint32_t lds_wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave * kK1hWArea));
	const uint64_t karg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
	const uint32_t karg_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)karg);
	const uint32_t karg_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(karg >> 32));
	// gen_k1h.py (VARIANTS) emits one body per (k, gap) and s_bits class (7, >= 8)
#define K1H_BODY(text)                                                                                                                       \
	asm volatile(text ::"s"(karg_lo), "s"(karg_hi), "s"(wave_gid), "s"(n_waves), "s"(lds_wbase), "s"(first_block), "s"(end_block) : K1H_CLOBBERS_V, K1H_CLOBBERS_S)
	if constexpr (K == 32 && GAP == 0 && SB == 7) K1H_BODY(K1H_ASM_K32_G0_S7);
	else if constexpr (K == 32 && GAP == 0 && SB == 8) K1H_BODY(K1H_ASM_K32_G0_S8);
	else if constexpr (K == 12 && GAP == 2 && SB == 7) K1H_BODY(K1H_ASM_K12_G2_S7);
	else if constexpr (K == 12 && GAP == 2 && SB == 8) K1H_BODY(K1H_ASM_K12_G2_S8);
	else static_assert(K < 0, "gen_k1h.py emits no body for this (k, gap)");
#undef K1H_BODY
}

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
	const uint32_t C = a.n_chunks, L = a.read_len, W = L - k + 1u;
	const uint64_t n_rows = (uint64_t)a.n_tiles * C;
	uint32_t f1_sub = 0;
	bool slow = false;
	for (uint32_t i = bx * 256u + tid; i < n_sus_waves; i += nbx * 256u)
		slow |= a.sus_count[i] == 0xffffffffu; // a K1h wave ran out of room for its suspects
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
			const int jmax = min(min(46, 15 + tzb), (int)W + (int)k - 2 - 16 * (int)c);
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
__device__ __forceinline__ void k1f_suspect_role(const K1fItem& item, const uint32_t bx, const uint32_t nbx)
{
	const K1hArgs& a = item.a;
	const uint32_t k = item.k, n_sus_waves = item.n_waves;
	const uint32_t tid = threadIdx.x;
	const uint32_t C = a.n_chunks, s_bits = a.s_bits, r_bits = a.r_bits;
	const uint32_t rmask = (1u << r_bits) - 1u;
	const uint4* const t4v = reinterpret_cast<const uint4*>(item.t4);
	for (uint32_t hb = bx; hb < 2u * n_sus_waves; hb += nbx) { // half a K1h wave's region per block
		const uint32_t reg = hb >> 1;
		uint32_t n = a.sus_count[reg];
		if (n == 0xffffffffu) n = 0; // the region overflowed: the launch takes the slow path, which ignores the suspects
		for (uint32_t i = (hb & 1u) * 256u + tid; i < n; i += 512u) {
			uint4* const ep = a.sus + (size_t)reg * a.sus_cap + i;
			const uint4 e = *ep;
			const uint32_t t = e.y, r = e.z & 2047u, w = e.z >> 11;
			const uint32_t c0 = w >> 4, off = w & 15u;
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
			const uint64_t win = k >= 64u ? ~0ull : (1ull << k) - 1ull;
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
				atomicAdd(a.sketch0 + idx, 1u);
				ep->x = idx;
				ep->w = 1u;
			}
		}
	}
}

// K1f, first launch: the F1 role and the suspect role side by side (blocks [0, n_f1) and the rest), blockIdx.y = the K1h launch of the batch
__global__ __launch_bounds__(256) void k1h_fix_kernel(const K1fBatch batch, const uint32_t n_f1_blocks)
{
	const K1fItem& item = batch.item[blockIdx.y];
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
	// what the suspect role counted goes back first: every window of a dirty-affected block is re-derived below, tie windows included
	for (uint32_t reg = blockIdx.x; reg < item.n_waves; reg += gridDim.x) {
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
		e_hi = min(e_hi, (int)L - 1);
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
	return ((k == 32 && gap == 0) || (k == 12 && gap == 2)) && s_bits >= 7 && s_bits <= 30 && r_bits + 1 + (s_bits - 7) <= 32;
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

namespace {
template <typename F>
hipError_t for_each_k1h_kernel(F f) // every instantiation, with its k
{
	hipError_t rc = f(reinterpret_cast<const void*>(&sketch_k1h_kernel<32, 7, 0>), 32u);
	if (rc == hipSuccess) rc = f(reinterpret_cast<const void*>(&sketch_k1h_kernel<32, 8, 0>), 32u);
	if (rc == hipSuccess) rc = f(reinterpret_cast<const void*>(&sketch_k1h_kernel<12, 7, 2>), 12u);
	if (rc == hipSuccess) rc = f(reinterpret_cast<const void*>(&sketch_k1h_kernel<12, 8, 2>), 12u);
	return rc;
}
} // namespace
hipError_t set_sketch_k1h_smem_limit()
{
	return for_each_k1h_kernel([](const void* fn, uint32_t k) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k1h_lds_bytes(k)); });
}

// sixteenths of a paired wave's share that a wave alone on its SIMD takes (NTC_K1H_LONE_WEIGHT: tuning runs)
static uint32_t k1h_lone_weight()
{
	static const uint32_t w = [] {
		const char* s = std::getenv("NTC_K1H_LONE_WEIGHT");
		const long v = s ? std::strtol(s, nullptr, 10) : 0;
		return (uint32_t)(v >= 8 && v <= 64 ? v : 20);
	}();
	return w;
}

// K1h over one batch on stream st; *args_out = the arguments as launched (block shares filled in), *n_waves = its waves (suspect regions)
hipError_t launch_sketch_k1h(const K1hArgs& a, uint32_t k, uint32_t gap, unsigned cus, hipStream_t st, K1hArgs* args_out, uint32_t* n_waves)
{
	if (!sketch_k1h_supports(k, gap, a.s_bits, a.r_bits)) return hipErrorInvalidValue;
	const uint32_t nb = sketch_k1h_blocks(k, a.read_len);
	const uint64_t total = (uint64_t)a.n_tiles * nb; // blocks of the batch, shared out evenly: a wave needs at least ~4 blocks to be worth its start-up
	const unsigned grid = (unsigned)std::min<uint64_t>((total + 4 * kK1hWaves - 1) / (4 * kK1hWaves), cus);
	K1hArgs b = a;
	b.blocks_per_wave = (uint32_t)((total + (uint64_t)grid * kK1hWaves - 1) / ((uint64_t)grid * kK1hWaves));
	b.nb_magic = (uint32_t)((1ull << 32) / nb);
	if (b.lone_weight == 0) b.lone_weight = k1h_lone_weight();
	const uint32_t lds = k1h_lds_bytes(k);
	if (k == 32 && a.s_bits == 7) hipLaunchKernelGGL((sketch_k1h_kernel<32, 7, 0>), dim3(grid), dim3(384), lds, st, b);
	else if (k == 32) hipLaunchKernelGGL((sketch_k1h_kernel<32, 8, 0>), dim3(grid), dim3(384), lds, st, b);
	else if (a.s_bits == 7) hipLaunchKernelGGL((sketch_k1h_kernel<12, 7, 2>), dim3(grid), dim3(384), lds, st, b);
	else hipLaunchKernelGGL((sketch_k1h_kernel<12, 8, 2>), dim3(grid), dim3(384), lds, st, b);
	*args_out = b;
	*n_waves = grid * kK1hWaves;
	return hipGetLastError();
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
	const unsigned n_f1 = (unsigned)std::min<size_t>((rows + 15) / 16, (size_t)cus * 8);
	hipLaunchKernelGGL(k1h_fix_kernel, dim3(n_f1 + 2u * waves, n_items), dim3(256), 0, st, b, n_f1);
	// the F1 correction of the fast path; the slow path (LDS, a CU's worth of blocks) only for a flagged launch
	hipLaunchKernelGGL(k1h_slow_kernel, dim3(cus, n_items), dim3(256), 0, st, b);
	return hipGetLastError();
}

} // namespace ntc
