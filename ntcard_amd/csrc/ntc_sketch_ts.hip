// ntc_sketch_ts.hip — K1c "tiled streaming" kernel: ntHash -> sample -> count for equal-length batches in the TILED
// slot layout (include/ntcard_hip.h, ntc_submit_tiled_device) on gfx950.
//
// What it computes is ntRead + ntComp (ntcard.cpp:132-158) for one k of 12 .. 32: for every window of k consecutive
// ACGTU bases the canonical ntHash (nthash.hpp:242-257,275-279), the two sampling patterns on its top bits, and one
// increment of t_Counter[sample][hash & (rBuck - 1)] per sampled window (as a hit-log entry, ntc_apply.hip), plus F1 =
// the number of such windows.  ntHashIterator's N semantics (ntHashIterator.hpp:59-86: a window that contains a
// non-ACGTU byte yields nothing) are handled here, exactly, without a second kernel.
//
// Formulation (see gen_ts.py): the sampling decision only needs the top sBits + 1 bits of min(fh, rh), and those
// live in the 31-bit rotating half of the hash (nthash.hpp:186-217).  That half is walked BIT-SLICED — one VGPR
// holds one bit of it for 32 reads, a wave carries a tile of 2048 reads, a rotate is a renaming of registers —
// and the ~2^(1-sBits) candidate windows are re-derived exactly (full 64-bit fh and rh, canonical min, ntComp's
// patterns, counter index) from the packed bases by a resolve stage.  The first k / 16 blocks of a read run a filling body
// (nothing goes out), one XOR with a generated constant then puts the state on the track of a walk that started from the hash
// of k 'A's, and every later step is the one main body per (k, strand).
//
// Data layout: tile t = reads [2048 t, 2048 t + 2048); chunk c = bases [16 c, 16 c + 16) of every read of the tile;
// the 16 raw bytes of (t, c, read r) sit at ((t C + c) 2048 + r) 16.  A chunk of a tile is 32 KiB of contiguous HBM
// and one coalesced 16-byte load per lane hands lane l the bases of read 64 m + l — already in the position-major
// order the bit-sliced walk consumes, so a tile STREAMS through the CU chunk by chunk (any read length, no 78 KB
// tile image, no window re-filling).
//
// Work decomposition: one 512-thread workgroup per CU = two TEAMS of four waves, each team streaming its own tiles:
//   A1  loads a chunk (32 x 1 KiB, buffer loads one chunk ahead), packs it to 2 bits per base + a dirty flag per 16
//       bytes into a ring of 5 x 8 KiB packed chunks in LDS, and between chunks resolves the FORWARD strand's
//       candidates;
//   F,R walk one strand each: read the chunk's packed words, transpose the 32 x 32 bit matrix into bit planes
//       (registers), then per base step one generated body (31 state registers, function planes of the incoming and
//       outgoing base) and a candidate test that uses the other walker's ">= pattern" plane, exchanged through LDS
//       every half block; a step's candidate plane goes to the strand's LDS queue as one ballot-compacted 8-byte
//       item per lane;
//   A2  resolves the REVERSE strand's candidates and does the bookkeeping of non-ACGTU bytes (F1 corrections, dirty
//       pieces handed over by A1).
//   A resolver takes candidates out of its queue into per-lane slots, recomputes fh / rh from the packed ring with a
//   4-bases-per-lookup closed-form table, keeps the candidate of the canonical strand only, applies ntComp and logs
//   the counter index; windows that touch a dirty 16-byte piece are settled exactly from the raw bytes.
// The waves of a team only meet through monotonic LDS counters (no workgroup barrier after start-up).  On every SIMD
// a walker of one team shares the issue slots with an assistant of the other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ntc_kernels.hpp"
#include "ntc_tile_bits.hpp"

namespace ntc {

namespace {

#include "ntc_ts_gen.inc"

using namespace tilebits;

template <int J>
__device__ __forceinline__ void transpose_stage(uint32_t (&A)[32])
{
	constexpr uint32_t m = J == 4 ? 0x0f0f0f0fu : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
	for (int k = 0; k < 32; ++k) {
		if ((k & J) == 0) {
			const uint32_t x = A[k], y = A[k + J];
			if constexpr (J == 16) {
				A[k] = perm(y, x, 0x05040100u);
				A[k + J] = perm(y, x, 0x07060302u);
			} else if constexpr (J == 8) {
				A[k] = perm(y, x, 0x06020400u);
				A[k + J] = perm(y, x, 0x07030501u);
			} else {
				A[k] = bfi(m, x, y << J);
				A[k + J] = bfi(m, x >> J, y);
			}
		}
	}
}
// in-place transpose of a 32 x 32 bit matrix held in 32 registers (row i = A[i], column c = bit c)
__device__ __forceinline__ void transpose32(uint32_t (&A)[32])
{
	transpose_stage<16>(A);
	transpose_stage<8>(A);
	transpose_stage<4>(A);
	transpose_stage<2>(A);
	transpose_stage<1>(A);
}

// materialise the strand registers here: left alone, the compiler defers every state update it does not need for the next
// candidate test and keeps 16 steps' worth of function planes alive instead (256 VGPRs and spills)
__device__ __forceinline__ void pin31(uint32_t (&X)[31])
{
	asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]), "+v"(X[8]), "+v"(X[9]), "+v"(X[10]));
	asm volatile("" : "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]), "+v"(X[16]), "+v"(X[17]), "+v"(X[18]), "+v"(X[19]), "+v"(X[20]));
	asm volatile("" : "+v"(X[21]), "+v"(X[22]), "+v"(X[23]), "+v"(X[24]), "+v"(X[25]), "+v"(X[26]), "+v"(X[27]), "+v"(X[28]), "+v"(X[29]), "+v"(X[30]));
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (no LDS round trip, no scalar dependency)
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
	uint32_t s = v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);              // row_shr:2
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);              // row_shr:3
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, false);              // row_shr:4, banks 1-3
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, false);              // row_shr:8, banks 2-3
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false);              // row_bcast:15 -> rows 1, 3
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);              // row_bcast:31 -> rows 2, 3
	return s;
}

template <int SB>
__device__ __forceinline__ uint32_t ts_cand(const uint32_t (&S)[31])
{
	if constexpr (SB == 2) return ts_cand_s2(S);
	else if constexpr (SB == 3) return ts_cand_s3(S);
	else if constexpr (SB == 4) return ts_cand_s4(S);
	else if constexpr (SB == 5) return ts_cand_s5(S);
	else if constexpr (SB == 6) return ts_cand_s6(S);
	else if constexpr (SB == 7) return ts_cand_s7(S);
	else return ts_cand_s8(S);
}
template <int SB>
__device__ __forceinline__ void ts_xplanes(const uint32_t (&S)[31], uint32_t& eqA, uint32_t& geA, uint32_t& eqB)
{
	if constexpr (SB == 2) ts_xplanes_s2(S, eqA, geA, eqB);
	else if constexpr (SB == 3) ts_xplanes_s3(S, eqA, geA, eqB);
	else if constexpr (SB == 4) ts_xplanes_s4(S, eqA, geA, eqB);
	else if constexpr (SB == 5) ts_xplanes_s5(S, eqA, geA, eqB);
	else if constexpr (SB == 6) ts_xplanes_s6(S, eqA, geA, eqB);
	else if constexpr (SB == 7) ts_xplanes_s7(S, eqA, geA, eqB);
	else ts_xplanes_s8(S, eqA, geA, eqB);
}
// ---- LDS plan ------------------------------------------------------------------------------------------------
constexpr uint32_t kTile = kTileReads;
constexpr uint32_t kRing = 5;            // packed chunks kept: a block's candidates need chunks n-2 .. n, the walkers are one ahead, and a fifth
                                         // slot lets A1 park a chunk before the block two behind is resolved (no copy held in registers)
constexpr uint32_t kQCap = 512;          // candidate items per strand queue (power of two)
constexpr uint32_t kSCap = 320;          // suspects (candidates next to a dirty piece) waiting for their raw bytes: up to 256 join per pass
constexpr uint32_t kDCap = 128;          // dirty pieces (A1 -> A2)
constexpr uint32_t kOffPR = 0;                               // packed ring  [kRing][32][64] dwords
constexpr uint32_t kOffDB = kOffPR + kRing * 8192u;          // dirty words  [kRing][64] dwords
constexpr uint32_t kOffQF = kOffDB + kRing * 256u;           // F queue      [kQCap + 64] x 8 B
constexpr uint32_t kOffQR = kOffQF + (kQCap + 64u) * 8u;     // R queue
constexpr uint32_t kOffSQ = kOffQR + (kQCap + 64u) * 8u;     // suspects     [kSCap] x 8 B
constexpr uint32_t kOffS2 = kOffSQ + kSCap * 8u;             // suspects of the second resolver
constexpr uint32_t kOffDQ = kOffS2 + kSCap * 8u;             // dirty queue  [kDCap] x 8 B
constexpr uint32_t kOffXF = kOffDQ + kDCap * 8u;             // F's "top bits >= 01..1" planes of half a block [8][64] dwords
constexpr uint32_t kOffXR = kOffXF + 2048u;                  // R's
constexpr uint32_t kOffCT = kOffXR + 2048u;                  // control words
constexpr uint32_t kTeamBytes = kOffCT + 256u;
// control words (all monotonic).  Words that one wave reads together sit together: {BLK_F, QF_TAIL, BLK_R, QR_TAIL} is one
// ds_read_b128 for A2, {PL_TAKEN_F, PL_TAKEN_R} one ds_read_b64 for A1, the block-end tails are {F, R} pairs.
enum { C_BLK_F = 0, C_QF_TAIL = 1, C_BLK_R = 2, C_QR_TAIL = 3, C_RES_R = 4, C_PR_READY = 7, C_RES_F = 8,
       C_QF_HEAD = 9, C_QR_HEAD = 10, C_DQ_TAIL = 11, C_DQ_HEAD = 12,
       C_XPUB_F = 13, C_XPUB_R = 14 /* half blocks whose planes are in the exchange area */, C_XCONS_F = 15 /* half blocks of R's planes F has read */,
       C_XCONS_R = 32, C_BT = 16 /* 8 pairs: queue tails {F, R} at the end of block b & 7 */ };

// Control words are read and written with explicit DS instructions: a `volatile` access through a generic pointer is
// compiled to FLAT, and a FLAT store to LDS is not ordered against the DS writes before it (nor a FLAT load against the DS
// reads behind it) — and it waits on vmcnt, i.e. on every global load the wave has in flight.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; } // low half of the aperture address = LDS offset
__device__ __forceinline__ void lds_publish(uint32_t* p, uint32_t v)
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p)
{
	uint32_t v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
	return rfl(v);
}
__device__ __forceinline__ uint2 lds_peek2(const uint32_t* p) // 8-byte aligned pair
{
	uint2 v;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
	return make_uint2(rfl(v.x), rfl(v.y));
}
__device__ __forceinline__ uint4 lds_peek4(const uint32_t* p) // 16-byte aligned quad
{
	v4u32 v;
	asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
	return make_uint4(rfl(v.x), rfl(v.y), rfl(v.z), rfl(v.w));
}
// spin until *p >= v (wrapping compare)
__device__ __forceinline__ void lds_wait_ge(const uint32_t* p, uint32_t v)
{
	while ((int32_t)(lds_peek(p) - v) < 0)
		__builtin_amdgcn_s_sleep(1);
}
#include "ntc_instr.hpp" // TS_T / TS_ACC / TS_WAIT / TS_FLUSH: no-ops unless built with -DTS_TIMERS (tools/dbg)


// ---- the resolve stage of ONE strand's candidates (shared by the two assistant waves of a team) -----------------------------
// Candidate words come out of the strand's LDS queue.  Every lane owns NI item SLOTS (registers): a slot holds one candidate
// word of some walker lane and step and gives up its lowest set bit (one read) per pass; an empty slot takes the next word of
// the queue.  One pass = up to 64 NI candidates: packed bases from the ring -> closed form, 4 bases per look-up -> canonical
// hash (nthash.hpp:275-279) -> ntComp's patterns (ntcard.cpp:132-145) -> hit log.  The three LDS round trips of a pass (words, packed
// bases, table entries) are each issued for all NI slots at once, and slot / log positions come from one prefix sum per pass:
// a lone wave pays ~20 clk for every vector compare whose mask a scalar instruction then reads.
// Blocks (= chunks) are TAKEN in order (cur: every word of the blocks before it has left the queue) and RESOLVED in order (res:
// no slot holds a word of a block before it; published: the packer may reuse the ring slots those blocks read).  A word with
// several bits stays in its slot over several passes, also across block boundaries; whenever the wave would otherwise wait it
// runs a pass for the slots alone, so `res` never trails for want of new candidates.
template <int K, int NI>
struct TsResolver {
	static constexpr int NG = (K + 3) / 4; // table groups: 4 bases per look-up (the last one is partial when k % 4 != 0: its table has no terms beyond base k - 1)
	const TsArgs* a;
	unsigned char* tb;
	uint32_t* ctl;
	const unsigned char* t4;
	const uint2* queue; // the strand's queue
	uint2* sq;          // suspects: candidates whose window touches a 16-byte piece with a non-ACGTU byte somewhere
	uint32_t c_pair, c_bt, c_head, c_res; // control words: {blocks walked, queue tail}, tails at block ends (stride 2), queue head, blocks resolved
	uint32_t lane, C, n_teams, team_g, n_valid_last, rmask, rbuck, s_bits, log_stride;
	bool has_partial, use_log;
	// state
	uint32_t sh[NI], sm[NI], sb[NI]; // slot: candidate word, its meta word, the block it was taken in
	uint32_t head, sq_fill, lreg, lfill;
	uint32_t cur, res, t, seq, c, lim;
	bool all_taken, ring_ok, done, tile_event;

	__device__ __forceinline__ void init(const TsArgs* a_, unsigned char* tb_, uint32_t* ctl_, const unsigned char* t4_, bool fwd, uint32_t lane_, uint32_t C_, uint32_t n_teams_,
	                                     uint32_t team_g_, uint32_t log_first, uint32_t log_stride_)
	{
		a = a_;
		tb = tb_;
		ctl = ctl_;
		t4 = t4_;
		queue = reinterpret_cast<const uint2*>(tb + (fwd ? kOffQF : kOffQR));
		sq = reinterpret_cast<uint2*>(tb + (fwd ? kOffS2 : kOffSQ));
		c_pair = fwd ? C_BLK_F : C_BLK_R;
		c_bt = C_BT + (fwd ? 0u : 1u);
		c_head = fwd ? C_QF_HEAD : C_QR_HEAD;
		c_res = fwd ? C_RES_F : C_RES_R;
		lane = lane_;
		C = C_;
		n_teams = n_teams_;
		team_g = team_g_;
		has_partial = (a->n_reads & (kTile - 1u)) != 0u;
		n_valid_last = has_partial ? (uint32_t)(a->n_reads & (kTile - 1u)) : kTile;
		rmask = (1u << a->r_bits) - 1u;
		rbuck = 1u << a->r_bits;
		s_bits = a->s_bits;
		log_stride = log_stride_;
		use_log = a->log_regions != 0 && (a->log_mode == nullptr || rfl(*a->log_mode) == 0u);
		lreg = log_first;
		lfill = 0;
		if (use_log && lreg < a->log_regions) lfill = rfl(a->log_fill[lreg]);
#pragma unroll
		for (int j = 0; j < NI; ++j)
			sh[j] = sm[j] = sb[j] = 0;
		head = sq_fill = 0;
		cur = res = seq = c = lim = 0;
		t = team_g;
		all_taken = team_g >= a->n_tiles;
		ring_ok = done = tile_event = false;
	}
	__device__ __forceinline__ v4u32 raw_piece(uint32_t tt, uint32_t cc, uint32_t r) const
	{
		return *reinterpret_cast<const v4u32*>(a->tiles + (((size_t)tt * C + cc) * kTile + r) * 16u);
	}
	// append the wave's hits to its hit-log region (ntComp's increment, deferred: ntc_apply.hip); cnt = hits of this lane (0 / 1 per key)
	template <int N>
	__device__ __forceinline__ void log_append(const uint32_t (&hit)[N], const uint32_t (&key)[N], uint32_t nhit)
	{
		uint32_t hincl, total;
		if constexpr (N == 1) {
			const uint64_t hmk = ballot(nhit != 0u);
			hincl = mbcnt(hmk) + nhit;
			total = (uint32_t)__popcll(hmk);
		} else {
			hincl = wave_scan(nhit);
			total = (uint32_t)__builtin_amdgcn_readlane((int)hincl, 63);
		}
		if (total == 0u) return;
		if (use_log) {
			while (lreg < a->log_regions && total > a->log_region_cap - lfill) {
				if (lane == 0) a->log_fill[lreg] = lfill;
				lreg += log_stride;
				lfill = lreg < a->log_regions ? rfl(a->log_fill[lreg]) : 0u;
			}
		}
		if (use_log && lreg < a->log_regions) {
			uint32_t* dst = a->log + (uint64_t)lreg * a->log_region_cap + lfill + (hincl - nhit); // a lane's hits go behind those of the lanes below it
#pragma unroll
			for (int j = 0; j < N; ++j) {
				if (hit[j]) *dst = key[j];
				dst += hit[j];
			}
			lfill += total;
		} else { // no log, or this wave's regions are full: the literal form, one device atomic per sampled k-mer (ntcard.cpp:142-143)
#pragma unroll
			for (int j = 0; j < N; ++j)
				if (hit[j]) atomicAdd(a->sketch0 + key[j], 1u);
		}
	}
	// suspects: exact validity from the raw bytes.  entry: {counter index, read | window << 11 | (tile sequence number & 15) << 27}
	__device__ __forceinline__ void flush_suspects()
	{
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		for (uint32_t base = 0; base < sq_fill; base += 64u) {
			const bool act = base + lane < sq_fill;
			const uint2 s = sq[act ? base + lane : 0u];
			const uint32_t r = s.y & 2047u, w = (s.y >> 11) & 0xffffu, c0 = w >> 4;
			const uint32_t tt = team_g + (seq - ((seq - (s.y >> 27)) & 15u)) * n_teams; // taken at most a tile ago
			uint32_t m0 = 0, m1 = 0, m2 = 0;
			if (act) {
				m0 = inv16(raw_piece(tt, c0, r));
				if (c0 + 1u < C) m1 = inv16(raw_piece(tt, c0 + 1u, r)); // (k <= 16 windows may end in their first piece)
				if (c0 + 2u < C) m2 = inv16(raw_piece(tt, c0 + 2u, r));
			}
			const uint64_t all = (uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32);
			constexpr uint32_t kWin = K == 32 ? 0xffffffffu : (1u << (K & 31)) - 1u;
			const uint32_t ok[1] = {act && ((uint32_t)(all >> (w & 15u)) & kWin) == 0u ? 1u : 0u}; // no non-ACGTU byte among the window's k bases
			const uint32_t ky[1] = {s.x};
			log_append<1>(ok, ky, ok[0]);
		}
		sq_fill = 0;
	}
	__device__ __forceinline__ bool slots_busy() const
	{
		bool any = false;
#pragma unroll
		for (int j = 0; j < NI; ++j)
			any |= sh[j] != 0u;
		return ballot(any) != 0;
	}
	__device__ __forceinline__ bool slots_from(uint32_t blk) const // some slot still holds a word taken in block `blk`
	{
		bool any = false;
#pragma unroll
		for (int j = 0; j < NI; ++j)
			any |= sh[j] != 0u && sb[j] == blk;
		return ballot(any) != 0;
	}
	__device__ __forceinline__ void retire() // blocks before `cur` are resolved once no slot holds a word of theirs
	{
		while (res < cur && !slots_from(res)) {
			++res;
			lds_publish(ctl + c_res, res);
		}
	}
	__device__ __forceinline__ void pass(uint32_t avail) // avail: words the queue may hand out now
	{
		const uint32_t* const pr = reinterpret_cast<const uint32_t*>(tb + kOffPR);
		const uint32_t* const db = reinterpret_cast<const uint32_t*>(tb + kOffDB);
		// 1. refill the empty slots from the queue: free slot number x of the wave takes word x.  Straight-line: a slot that takes
		// nothing reads the queue's spare area
		uint32_t nfree = 0;
#pragma unroll
		for (int j = 0; j < NI; ++j)
			nfree += sh[j] == 0u ? 1u : 0u;
		uint32_t fincl, n_free;
		if constexpr (NI == 1) {
			const uint64_t fm = ballot(nfree != 0u);
			fincl = mbcnt(fm) + nfree;
			n_free = (uint32_t)__popcll(fm);
		} else {
			fincl = wave_scan(nfree);
			n_free = (uint32_t)__builtin_amdgcn_readlane((int)fincl, 63);
		}
		const uint32_t n_new = n_free < avail ? n_free : avail;
		uint2 nw[NI];
		bool take[NI];
		uint32_t ix = fincl - nfree;
#pragma unroll
		for (int j = 0; j < NI; ++j) {
			const bool fr = sh[j] == 0u;
			take[j] = fr && ix < n_new;
			nw[j] = queue[take[j] ? (head + ix) & (kQCap - 1u) : kQCap + lane];
			ix += fr ? 1u : 0u;
		}
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int j = 0; j < NI; ++j) {
			sh[j] = take[j] ? nw[j].x : sh[j];
			sm[j] = take[j] ? nw[j].y : sm[j];
			sb[j] = take[j] ? cur : sb[j];
		}
		head += n_new;
		if (n_new) lds_publish(ctl + c_head, head); // (behind the item reads above)
		// 2. one candidate per slot: its packed words + the dirty words of the pieces its window touches
		uint32_t d0[NI], d1[NI], d2[NI], b0[NI], b1[NI], b2[NI], rr[NI], ww[NI], mm[NI], yy[NI];
#pragma unroll
		for (int j = 0; j < NI; ++j) {
			const uint32_t h = sh[j], y = sm[j];
			const uint32_t m = (uint32_t)__builtin_ctz(h | 0x80000000u);
			sh[j] = h & (h - 1u);
			const uint32_t l0 = y & 63u, s0 = (y >> 8) & 7u;
			const uint32_t col = m * 64u + l0;
			const uint32_t s1 = s0 + 1u == kRing ? 0u : s0 + 1u, s2 = s1 + 1u == kRing ? 0u : s1 + 1u;
			d0[j] = pr[s0 * 2048u + col];
			d1[j] = pr[s1 * 2048u + col];
			d2[j] = pr[s2 * 2048u + col];
			b0[j] = db[s0 * 64u + l0];
			b1[j] = db[s1 * 64u + l0];
			b2[j] = db[s2 * 64u + l0];
			rr[j] = col;
			ww[j] = (y >> 11) & 0xffffu;
			mm[j] = m;
			yy[j] = h != 0u ? y | 0x80000000u : 0u; // bit 31: the slot holds a candidate (the walkers leave it clear)
		}
		__builtin_amdgcn_sched_barrier(0);
		// 3. closed form, 4 bases per look-up: all table addresses, then all look-ups in flight together, then the XORs
		uint32_t toff[NI][NG];
#pragma unroll
		for (int j = 0; j < NI; ++j) {
			const uint32_t shf = (ww[j] & 15u) * 2u;
#pragma unroll
			for (int i = 0; i < (NG + 3) / 4; ++i) {
				const uint32_t x = i == 0 ? alignbit(d1[j], d0[j], shf) : alignbit(d2[j], d1[j], shf); // 16 bases of the window
#pragma unroll
				for (int g = 0; g < 4; ++g)
					if (i * 4 + g < NG) toff[j][i * 4 + g] = ((x >> (8 * g)) & 0xffu) * 16u;
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		v4u32 tv[NI][NG];
#pragma unroll
		for (int j = 0; j < NI; ++j)
#pragma unroll
			for (int i = 0; i < NG; ++i)
				tv[j][i] = *reinterpret_cast<const v4u32*>(t4 + (uint32_t)i * 4096u + toff[j][i]); // the group's 4 KiB rides in the offset field
		__builtin_amdgcn_sched_barrier(0);
		uint32_t nhit = 0, anysus = 0;
		uint32_t hit[NI], key[NI], sus[NI]; // 0 / 1
#pragma unroll
		for (int j = 0; j < NI; ++j) {
			// XOR of the NG table entries, three inputs per instruction
			v4u32 acc = tv[j][0];
#pragma unroll
			for (int i = 1; i + 1 < NG; i += 2) {
				acc.x = (uint32_t)__builtin_amdgcn_bitop3_b32(acc.x, tv[j][i].x, tv[j][i + 1].x, 0x96);
				acc.y = (uint32_t)__builtin_amdgcn_bitop3_b32(acc.y, tv[j][i].y, tv[j][i + 1].y, 0x96);
				acc.z = (uint32_t)__builtin_amdgcn_bitop3_b32(acc.z, tv[j][i].z, tv[j][i + 1].z, 0x96);
				acc.w = (uint32_t)__builtin_amdgcn_bitop3_b32(acc.w, tv[j][i].w, tv[j][i + 1].w, 0x96);
			}
			if constexpr (NG % 2 == 0) {
				acc.x ^= tv[j][NG - 1].x;
				acc.y ^= tv[j][NG - 1].y;
				acc.z ^= tv[j][NG - 1].z;
				acc.w ^= tv[j][NG - 1].w;
			}
			const uint32_t flo = acc.x, fhi = acc.y, rlo = acc.z, rhi = acc.w;
			const uint64_t fh = ((uint64_t)fhi << 32) | flo, rh = ((uint64_t)rhi << 32) | rlo;
			const bool rev = rh < fh; // nthash.hpp:275-279
			const uint32_t hi = rev ? rhi : fhi, lo = rev ? rlo : flo;
			// ntComp (ntcard.cpp:132-145) on the canonical value; sample 1 wins when both match
			const bool c1 = (hi >> (32 - s_bits)) == ((1u << (s_bits - 1)) - 1u);
			const bool c0m = (hi >> (31 - s_bits)) == 1u;
			const uint32_t y = yy[j];
			// the candidate of the canonical strand only (both strands may have flagged the window)
			bool ht = ((y >> 31) != 0u) & (rev == ((y & 64u) != 0u)) & (c0m | c1);
			if (has_partial) ht &= (y & 128u) == 0u || rr[j] < n_valid_last; // slots behind the last read of the batch
			key[j] = a->key_base + (lo & rmask) + (c1 ? rbuck : 0u);
			// a window that touches a 16-byte piece with a non-ACGTU byte somewhere is settled from the raw bytes (the third piece
			// only counts when the window is not chunk-aligned)
			const uint32_t dd = (b0[j] | ((ww[j] & 15u) + (uint32_t)K > 16u ? b1[j] : 0u) | ((ww[j] & 15u) + (uint32_t)K > 32u ? b2[j] : 0u)) >> mm[j];
			sus[j] = ht ? dd & 1u : 0u;
			hit[j] = (ht ? 1u : 0u) & ~sus[j];
			nhit += hit[j];
			anysus |= sus[j];
#ifdef TS_DEBUG
			if ((y >> 31) && a->dbg) {
				const uint32_t ixd = atomicAdd(a->dbg, 1u);
				uint32_t* o = a->dbg + 16 + 12 * (size_t)ixd;
				o[0] = rr[j]; o[1] = ww[j]; o[2] = 0; o[3] = y; o[4] = d0[j]; o[5] = d1[j]; o[6] = d2[j]; o[7] = flo; o[8] = fhi; o[9] = rlo; o[10] = rhi;
				o[11] = (rev ? 1u : 0u) | ((y & 64u) ? 2u : 0u) | (ht ? 4u : 0u);
			}
#endif
		}
		log_append<NI>(hit, key, nhit);
		if (ballot(anysus != 0u) != 0) { // rare
#pragma unroll
			for (int j = 0; j < NI; ++j) {
				const uint64_t smk = ballot(sus[j] != 0u);
				if (sus[j]) sq[sq_fill + mbcnt(smk)] = make_uint2(key[j], rr[j] | (ww[j] << 11) | (yy[j] & 0x78000000u)); // + the walker's tile tag (bits 27..30)
				sq_fill += (uint32_t)__popcll(smk);
			}
			if (sq_fill > kSCap - 64u * NI) flush_suspects();
		}
	}
	// one turn of the resolver: 0 nothing to do right now, 1 worked, 2 every block of every tile is resolved
	__device__ __forceinline__ int step()
	{
		bool run = false;
		uint32_t avail = 0;
		int rc = 0;
		if (all_taken) {
			if (!slots_busy()) return 2;
			run = true;
		} else if (!ring_ok) {
			ring_ok = (int32_t)(lds_peek(ctl + C_PR_READY) - (cur + 1u)) >= 0; // the packed words of this block's chunk
			if (!ring_ok) run = slots_busy();
		}
		if (ring_ok) {
			if (!done && lim - head < 64u * NI) {
				// blocks walked and queue tail in one read; a tail read together with "block walked" may already hold words of later
				// blocks, whose chunks are not in the ring yet: then the tail noted at the end of the block counts
				const uint2 q = lds_peek2(ctl + c_pair);
				done = (int32_t)(q.x - (cur + 1u)) >= 0;
				lim = done ? lds_peek(ctl + c_bt + 2u * (cur & 7u)) : q.y;
			}
			avail = lim - head;
			if (avail >= 48u * NI || (done && avail != 0u)) { // a pass costs the same whether its slots are full or not: wait for 3/4 of them
				run = true;
			} else if (done) { // every word of block cur has left the queue (some may still sit in slots): next block
				++cur;
				ring_ok = done = false;
				rc = 1;
				if (++c == C) {
					c = 0;
					tile_event = true; // (the caller books the tile: F1, dirty pieces)
					if (sq_fill != 0u) flush_suspects(); // a suspect never waits longer than a tile: its tile tag stays unambiguous however many tiles a team walks
					++seq;
					t += n_teams;
					all_taken = t >= a->n_tiles;
				}
			} else {
				run = slots_busy();
			}
		}
		if (run) {
			pass(avail);
			rc = 1;
		}
		retire();
		return rc;
	}
	__device__ __forceinline__ void finish()
	{
		flush_suspects();
		if (use_log && lane == 0 && lreg < a->log_regions) a->log_fill[lreg] = lfill;
	}
};

} // namespace

template <int K, int SB>
__global__ __launch_bounds__(512, 2) void sketch_ts_kernel(const TsArgs a)
{
	static_assert(K >= 12 && K <= 32, "K1c: a window spans at most 3 chunks and leaves at most 2 chunks behind the one being walked (gen_ts.py emits k = 12 .. 32)");
	extern __shared__ __align__(16) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t wave = rfl((uint32_t)tid >> 6);
	// wave w and w + 4 share a SIMD: a walker of one team sits next to an assistant of the other
	const uint32_t team = wave < 4u ? wave >> 1 : ((wave >> 1) & 1u) ^ 1u;
	const uint32_t role = wave < 4u ? (wave & 1u) : 2u + (wave & 1u); // 0 F, 1 R, 2 A1, 3 A2
	const uint32_t t4_bytes = (uint32_t)((K + 3) / 4) * 4096u;
	unsigned char* const t4 = smem;
	unsigned char* const tb = smem + t4_bytes + team * kTeamBytes;
	uint32_t* const ctl = reinterpret_cast<uint32_t*>(tb + kOffCT);
	{
		const uint4* src = reinterpret_cast<const uint4*>(a.t4);
		for (uint32_t i = tid; i < t4_bytes / 16u; i += 512u)
			reinterpret_cast<uint4*>(t4)[i] = src[i];
		if (tid < 128) reinterpret_cast<uint32_t*>(smem + t4_bytes + (uint32_t)(tid >> 6) * kTeamBytes + kOffCT)[tid & 63] = 0;
	}
	__syncthreads(); // the only workgroup barrier: table and zeroed control words are in place
	const uint32_t C = a.n_chunks;
	const uint32_t W = a.read_len - (uint32_t)K + 1u; // windows per read
	const uint32_t n_teams = gridDim.x * 2u;
	const uint32_t team_g = blockIdx.x * 2u + team;
	const bool has_partial = (a.n_reads & (kTile - 1u)) != 0u;
#ifdef TS_TIMERS
	uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const uint64_t t_start = __builtin_readcyclecounter();
#endif

	if (role == 2u) {
		// =============================== A1: load, pack, publish; resolve the forward strand ===============================
		__builtin_amdgcn_s_setprio(3); // the walkers run ahead of this wave anyway: when both want the SIMD, it goes first (measured: 0.87 -> 0.75 ms per 10 M reads)
		uint32_t n = 0;
		v4u32 raw[32];
		const uint32_t voff = (uint32_t)lane * 16u;
		// The chunk's 32 loads are BUFFER loads: resource descriptor of the chunk (scalar registers) + one 32-bit lane offset +
		// scalar / immediate offsets (as flat global loads the in-loop copies get a 64-bit vector address each: 64 more registers).
		// Software pipeline with a distance of one chunk, item by item: as soon as the 16 bytes of read group m are packed, the
		// same registers take the load of group m of the NEXT chunk, so every load has a whole chunk period to land (issuing
		// the next chunk's loads only after the whole pack exposed the full HBM latency once per chunk: 8.7 k clk per chunk
		// instead of ~3 k).
		auto chunk_rsrc = [&](uint32_t t, uint32_t c) {
			const unsigned char* p = a.tiles + ((size_t)rfl(t) * C + rfl(c)) * (size_t)(kTile * 16u);
			return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), 0, (int)(kTile * 16u), 0x00020000);
		};
		// This wave also resolves the FORWARD strand's candidates (the other assistant takes the reverse strand's): whenever the ring
		// has no room for the next chunk it runs resolver turns, and one turn per chunk otherwise.  NI = 2 slots per lane: the chunk
		// in flight keeps 128 registers.
		TsResolver<K, 2> rs;
		rs.init(&a, tb, ctl, t4, /*fwd=*/true, (uint32_t)lane, C, n_teams, team_g, team_g * 2u + 1u, n_teams * 2u);
		uint32_t dq_tail = 0;
		const uint32_t my_tiles = team_g < a.n_tiles ? (a.n_tiles - team_g + n_teams - 1u) / n_teams : 0u;
		const uint32_t total = my_tiles * C; // this team's chunks as one flat sequence
		if (total != 0u) {
			const __amdgpu_buffer_rsrc_t r0 = chunk_rsrc(team_g, 0u);
#pragma unroll
			for (int m = 0; m < 32; ++m)
				raw[m] = __builtin_amdgcn_raw_buffer_load_b128(r0, voff + (uint32_t)(m & 3) * 1024u, (m >> 2) * 4096, 2 /* slc: read once */);
		}
		{
			uint32_t t = team_g, seq = 0, c = 0;
			for (; n < total; ++n) {
				uint32_t dirtyword = 0;
				const bool last_c = c + 1u == C;
				const uint32_t tn = last_c ? t + n_teams : t, cn = last_c ? 0u : c + 1u;
				const bool more = n + 1u < total;
				const __amdgpu_buffer_rsrc_t rn = chunk_rsrc(more ? tn : t, more ? cn : c); // (the last chunk re-reads itself: one unconditional load site)
				// One resolver turn per chunk keeps the forward queue moving while the walkers have chunks ahead of them; more of them
				// while the ring has no room for this chunk: its packed words overwrite chunk n - 5, which blocks <= n - 3 read (both
				// strands' resolvers).  (ONE call site inside the chunk loop: the chunk in flight keeps 128 registers.)
				{
					TS_T(tg0);
					while (true) {
						TS_T(ts0);
						const int rc = rs.step();
						TS_T(ts1);
#ifdef TS_TIMERS
						if (rc != 0) {
							tacc[5] += ts1 - ts0;
							tacc[6] += 1;
						}
#endif
						if (n < 3u || ((int32_t)(rs.res - (n - 2u)) >= 0 && (int32_t)(lds_peek(ctl + C_RES_R) - (n - 2u)) >= 0)) break;
						if (rc == 0) __builtin_amdgcn_s_sleep(1);
					}
					TS_T(tg1);
					TS_ACC(1, tg0, tg1);
				}
				const uint32_t slot = n % kRing;
				uint32_t* pr = reinterpret_cast<uint32_t*>(tb + kOffPR) + slot * 2048u + lane;
#ifdef TS_TIMERS
				uint64_t tpk0, tpk1;
				__builtin_amdgcn_sched_barrier(0);
				asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tpk0)::"memory");
				__builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
				for (int m = 0; m < 32; ++m) {
					uint32_t bad;
					pr[m * 64] = pack16(raw[m], bad);
					dirtyword |= (bad != 0u ? 1u : 0u) << m;
					raw[m] = __builtin_amdgcn_raw_buffer_load_b128(rn, voff + (uint32_t)(m & 3) * 1024u, (m >> 2) * 4096, 2);
					if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0); // keep pack and reload of a group together
				}
				reinterpret_cast<uint32_t*>(tb + kOffDB)[slot * 64u + lane] = dirtyword;
#ifdef TS_TIMERS
				__builtin_amdgcn_sched_barrier(0);
				asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tpk1) : "v"(dirtyword) : "memory");
				__builtin_amdgcn_sched_barrier(0);
				tacc[3] += tpk1 - tpk0;
#endif
				if (has_partial && t == a.n_tiles - 1u) { // slots behind the batch's last read: whatever they hold, it is not a read
					const uint32_t nv = rs.n_valid_last, groups = nv > (uint32_t)lane ? (nv - (uint32_t)lane + 63u) >> 6 : 0u; // read groups m with 64 m + lane < nv
					dirtyword &= groups >= 32u ? 0xffffffffu : (1u << groups) - 1u;
				}
				const uint64_t dm = ballot(dirtyword != 0u);
				if (dm != 0) { // rare: hand the dirty pieces to A2 (F1 corrections)
					const uint32_t cnt = (uint32_t)__popcll(dm);
					while ((int32_t)(dq_tail + cnt - lds_peek(ctl + C_DQ_HEAD) - kDCap) > 0)
						__builtin_amdgcn_s_sleep(1);
					if (dirtyword != 0u)
						reinterpret_cast<uint2*>(tb + kOffDQ)[(dq_tail + mbcnt(dm)) & (kDCap - 1u)] = make_uint2(dirtyword, (uint32_t)lane | ((seq & 15u) << 6) | (c << 10));
					dq_tail += cnt;
					lds_publish(ctl + C_DQ_TAIL, dq_tail);
				}
				lds_publish(ctl + C_PR_READY, n + 1u);
				seq += last_c ? 1u : 0u;
				t = tn;
				c = cn;
			}
		}
		while (true) { // every chunk is packed: the rest of the forward strand's candidates
			const int rc = rs.step();
			if (rc == 2) break;
			if (rc == 0) __builtin_amdgcn_s_sleep(1);
		}
		rs.finish();
		TS_FLUSH(2);
		return;
	}

	if (role <= 1u) {
		// =============================== F / R: walk one strand of the 31-bit half ===============================
		auto walk = [&](auto fwd_c) {
			constexpr bool FWD = decltype(fwd_c)::value;
			uint2* const queue = reinterpret_cast<uint2*>(tb + (FWD ? kOffQF : kOffQR));
			uint32_t* const c_tail = ctl + (FWD ? C_QF_TAIL : C_QR_TAIL);
			const uint32_t* const c_head = ctl + (FWD ? C_QF_HEAD : C_QR_HEAD);
			uint32_t qtail = 0, qhead_c = 0; // wave-uniform
			uint32_t n = 0;
			uint32_t xhalf = 0; // half blocks of planes exchanged so far
			uint32_t* const x_mine = reinterpret_cast<uint32_t*>(tb + (FWD ? kOffXF : kOffXR)) + lane;
			const uint32_t* const x_other = reinterpret_cast<const uint32_t*>(tb + (FWD ? kOffXR : kOffXF)) + lane;
			uint32_t* const x_pub_mine = ctl + (FWD ? C_XPUB_F : C_XPUB_R);
			const uint32_t* const x_pub_other = ctl + (FWD ? C_XPUB_R : C_XPUB_F);
			uint32_t* const x_cons_mine = ctl + (FWD ? C_XCONS_F : C_XCONS_R);
			const uint32_t* const x_cons_other = ctl + (FWD ? C_XCONS_R : C_XCONS_F);
			uint32_t meta_tile = 0, nb0 = 0; // nb0: ring slot of chunk 0 of the current tile
			auto push = [&](uint32_t h, uint32_t w) { // h: bit m set <=> read 64 m + lane is a candidate at window w
				// straight-line: lanes without a candidate store to their spare slot behind the queue
				h = w < W ? h : 0u;
				const uint64_t m = ballot(h != 0u);
				const uint32_t slot = (qtail + mbcnt(m)) & (kQCap - 1u);
				queue[h != 0u ? slot : kQCap + (uint32_t)lane] = make_uint2(h, meta_tile | (((nb0 + (w >> 4)) % kRing) << 8) | (w << 11)); // wave-uniform arithmetic
				qtail = rfl(qtail + (uint32_t)__popcll(m));
				if (__builtin_expect(qtail - qhead_c > kQCap - 64u, 0)) { // the next step may not fit: let A2 catch up
					lds_publish(c_tail, qtail);
					TS_T(tg0);
					do {
						qhead_c = lds_peek(c_head);
						if (qtail - qhead_c <= kQCap - 64u) break;
						__builtin_amdgcn_s_sleep(1);
					} while (true);
					TS_T(tg1);
					TS_ACC(2, tg0, tg1);
				}
			};
			for (uint32_t t = team_g, seq = 0; t < a.n_tiles; t += n_teams, ++seq) {
				meta_tile = (uint32_t)lane | (FWD ? 0u : 64u) | ((has_partial && t == a.n_tiles - 1u) ? 128u : 0u) | ((seq & 15u) << 27); // bits 27..30: which tile (suspects)
				nb0 = (seq * C) % kRing;
				// The first k / 16 blocks of a tile only fill the window: state 0, the filling body (nothing goes out).  From then on every
				// step is the main body, which for the rest of the first k steps sees 'A' (code 0: the zeroed history planes) go out — it
				// runs on the track of a walk started from the hash of k 'A's (gen_ts.py, poly_a_state), and one XOR with a constant
				// (ts_fix) moves the filling state onto that track.  After step k - 1 the state is the hash of the first k real bases.
				uint32_t S[31];
#pragma unroll
				for (int j = 0; j < 31; ++j)
					S[j] = 0;
				if constexpr (K / 16 == 0) ts_fix<FWD, K>(S); // k < 16: no filling block, the main body runs from the first step
				uint32_t H[2][32]; // the planes of the two chunks before the one being walked: the outgoing base is k <= 32 bases back
#pragma unroll
				for (int b = 0; b < 2; ++b)
#pragma unroll
					for (int i = 0; i < 32; ++i)
						H[b][i] = 0;
#pragma unroll 1
				for (uint32_t c = 0; c < C; ++c, ++n) {
					uint32_t I[32];
					TS_WAIT(1, ctl + C_PR_READY, n + 1u);
					{ // the chunk's packed words of this lane's 32 reads -> 32 bit planes (both walkers do this: it is cheaper than a
					  // third party publishing planes through one more LDS slot and one more hand-shake)
						const uint32_t* pw = reinterpret_cast<const uint32_t*>(tb + kOffPR) + (n % kRing) * 2048u + lane;
#pragma unroll
						for (int m = 0; m < 32; ++m)
							I[m] = pw[m * 64];
						transpose32(I); // I[2 q + b] = bit b of the code of base 16 c + q, one bit per read
					}
					// step q of this block takes base 16 c + q in and base 16 c + q - k out (planes 2 o, 2 o + 1 of the history, o = q + 32 - k)
					// and completes window 16 c + q - (k - 1) (none while that is negative: the unsigned value fails push()'s w < W)
					const uint32_t w0 = 16u * c - (uint32_t)K + 1u;
					if ((int32_t)(16u * c) + 15 <= K - 1) {
						// a block of nothing but window filling (the first k / 16 blocks of a tile): no outgoing base, no exchange, no queue;
						// when its last step completes window 0 (k = 16, 32) that window's candidates come from this strand's own patterns —
						// a superset the resolver narrows to the canonical strand like every other candidate
#pragma unroll
						for (int q = 0; q < 16; ++q) {
							ts_warm<FWD, K>(S, I[2 * q], I[2 * q + 1]);
							pin31(S);
						}
						if (c + 1u == (uint32_t)(K / 16)) ts_fix<FWD, K>(S); // the last filling block: onto the main body's track
						if ((int32_t)(16u * c) + 15 == K - 1) push(ts_cand<SB>(S), 0u);
					} else
#pragma unroll
					for (int hb = 0; hb < 2; ++hb) {
						// Candidates with the other strand's help: a window is sampled through THIS strand iff its top bits carry a pattern and
						// the other strand's are not smaller (gen_ts.py, emit_xplanes).  Each walker posts one plane per step (its top bits
						// >= 01..1) and, every half block, combines its own equality planes with the other walker's: a third fewer
						// candidates for the resolve stage than either strand's pattern alone.
						uint32_t eqA[8], eqB[8];
						if (xhalf != 0u) lds_wait_ge(x_cons_other, xhalf); // the other walker has read my planes of the previous half block
#pragma unroll
						for (int q8 = 0; q8 < 8; ++q8) {
							const int q = hb * 8 + q8, o = q + 32 - K;
							ts_main<FWD, K>(S, I[2 * q], I[2 * q + 1], o < 16 ? H[0][2 * (o & 15)] : o < 32 ? H[1][2 * (o & 15)] : I[2 * (o & 15)], o < 16 ? H[0][2 * (o & 15) + 1] : o < 32 ? H[1][2 * (o & 15) + 1] : I[2 * (o & 15) + 1]); // (k < 16: the outgoing base may lie in this chunk)
							pin31(S);
							uint32_t ge;
							ts_xplanes<SB>(S, eqA[q8], ge, eqB[q8]);
							x_mine[q8 * 64] = ge;
						}
						++xhalf;
						lds_publish(x_pub_mine, xhalf);
						TS_WAIT(3, x_pub_other, xhalf);
						uint32_t og[8];
#pragma unroll
						for (int q8 = 0; q8 < 8; ++q8)
							og[q8] = x_other[q8 * 64];
						lds_publish(x_cons_mine, xhalf); // (behind the reads above)
						if ((int32_t)(16u * c) + 8 * hb + 7 >= K - 1) { // (no window ends in the first k - 1 steps of a tile)
#pragma unroll
							for (int q8 = 0; q8 < 8; ++q8)
								push((eqA[q8] & og[q8]) | eqB[q8], w0 + (uint32_t)(hb * 8 + q8));
						}
					}
#pragma unroll
					for (int i = 0; i < 32; ++i) {
						H[0][i] = H[1][i];
						H[1][i] = I[i];
					}
					// block n is complete: its items end at qtail
					lds_publish(ctl + C_BT + 2u * (n & 7u) + (FWD ? 0u : 1u), qtail);
					lds_publish(c_tail, qtail);
					lds_publish(ctl + (FWD ? C_BLK_F : C_BLK_R), n + 1u);
				}
			}
		};
		if (role == 0u) walk(std::true_type{});
		else walk(std::false_type{});
		TS_FLUSH(role);
		return;
	}

	// =============================== A2: resolve the reverse strand's candidates, settle dirty pieces, F1 ===============================
	{
		__builtin_amdgcn_s_setprio(3);
		TsResolver<K, 2> rs;
		rs.init(&a, tb, ctl, t4, /*fwd=*/false, (uint32_t)lane, C, n_teams, team_g, team_g * 2u, n_teams * 2u);
		const uint2* const dq = reinterpret_cast<const uint2*>(tb + kOffDQ);
		uint32_t dq_head = 0;
		uint64_t f1_add = 0;  // wave-uniform: reads x windows
		uint32_t f1_sub = 0;  // per lane: windows lost to non-ACGTU bytes
		const uint32_t n_valid_last = rs.n_valid_last;
		// ---- dirty pieces: F1 loses the windows whose RIGHTMOST non-ACGTU byte lies in the piece.  The packer is gated on BLOCKS (it may be
		// three blocks ahead of the resolvers), and with reads of one chunk a block is a tile: items carry four bits of the tile sequence
		// number (round 3 carried one: F1 came out too high for reads of <= 16 bp on teams that walk three tiles or more) ----
		auto drain_dirty = [&](uint32_t t_cur, uint32_t seq_cur) {
			const uint32_t tail = lds_peek(ctl + C_DQ_TAIL);
			while (dq_head != tail) {
				const uint32_t cnt = tail - dq_head < 64u ? tail - dq_head : 64u;
				uint2 it = make_uint2(0u, 0u);
				if ((uint32_t)lane < cnt) it = dq[(dq_head + (uint32_t)lane) & (kDCap - 1u)];
				uint32_t word = it.x;
				const uint32_t l0 = it.y & 63u, c = it.y >> 10;
				const uint32_t t = t_cur + ((((it.y >> 6) & 15u) - seq_cur) & 15u) * n_teams; // the item's tile: this one or one of the next few
				while (word != 0u) {
					const uint32_t r = (uint32_t)__builtin_ctz(word) * 64u + l0;
					word &= word - 1u;
					const uint32_t A = inv16(rs.raw_piece(t, c, r));
					uint32_t B = 0;
					if (c + 1u < C) B = inv16(rs.raw_piece(t, c + 1u, r));
					if (c + 2u < C) B |= inv16(rs.raw_piece(t, c + 2u, r)) << 16;
					if (A != 0u) {
						// window w = 16 c - (k - 1) + j ends at base 16 c + j: it holds a byte of A iff some bit of A lies in [j - k + 1, j] (bit j
						// of A smeared over k positions: with k < 16 a window fits BETWEEN two non-ACGTU bytes of one piece, so lo .. hi + k - 1
						// is not the answer), and no later non-ACGTU byte iff j <= 15 + ctz(B) (k <= 32 + 1: B covers the 32 bases behind the
						// piece); 0 <= w < W
						uint64_t E = A;
						int cov = 1;
#pragma unroll
						for (int i = 0; i < 5; ++i)
							if (cov * 2 <= K) {
								E |= E << cov;
								cov *= 2;
							}
						if (cov < K) E |= E << (K - cov);
						const int tzb = B ? __builtin_ctz(B) : 32;
						const int jmin = max(0, (int)K - 1 - 16 * (int)c);
						const int jmax = min(min(46, 15 + tzb), (int)W + (int)K - 2 - 16 * (int)c);
						if (jmax >= jmin) f1_sub += (uint32_t)__popcll((E >> jmin) & ((2ull << (jmax - jmin)) - 1ull));
					}
				}
				dq_head += cnt;
				lds_publish(ctl + C_DQ_HEAD, dq_head);
			}
		};
		while (true) {
			const uint32_t t_now = rs.t, seq_now = rs.seq;
			TS_T(tr0);
			const int rc = rs.step();
			TS_T(tr1);
			if (rs.tile_event) { // tile t_now is walked and taken: every dirty piece of it is in the queue by now (and perhaps the first of the next tile)
				rs.tile_event = false;
				drain_dirty(t_now, seq_now);
				f1_add += (uint64_t)((has_partial && t_now == a.n_tiles - 1u) ? n_valid_last : kTile) * W;
			}
			if (rc == 2) break;
			if (rc == 1 && lds_peek(ctl + C_DQ_TAIL) - dq_head >= kDCap / 2u) drain_dirty(rs.t, rs.seq); // busy or not: the packer must not wait for room in the dirty queue
			if (rc == 0) {
				drain_dirty(rs.t, rs.seq);
				__builtin_amdgcn_s_sleep(1);
				TS_T(ti1);
				TS_ACC(2, tr0, ti1);
			} else {
				TS_ACC(3, tr0, tr1);
#ifdef TS_TIMERS
				tacc[4] += 1;
#endif
			}
		}
		rs.finish();
		// F1 (ntcard.cpp:154): one per window without a non-ACGTU byte
		uint32_t sub = f1_sub;
		for (int o = 32; o > 0; o >>= 1)
			sub += (uint32_t)__shfl_xor((int)sub, o);
		if (lane == 0 && f1_add != 0) atomicAdd(a.f1, (unsigned long long)(f1_add - sub));
		TS_FLUSH(3);
	}
}

// ---- instantiations -------------------------------------------------------------------------------------------
// One kernel per (k, sBits class); the object files ntc_sketch_ts_p{0..3}.o each carry the k with k % 4 == part
// (Makefile: -DTS_PART=n, built in parallel); -DTS_ONLY_K=k (tools/dbg) builds a single k.
#ifndef TS_PART
#define TS_PART 0
#define TS_ALL_PARTS 1
#endif
#ifdef TS_ONLY_K
#define TS_MINE(k) ((k) == TS_ONLY_K)
#elif defined(TS_ALL_PARTS)
#define TS_MINE(k) true
#else
#define TS_MINE(k) (((k) & 3) == TS_PART)
#endif
#define TS_FOR_K(X) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

namespace {
template <int K, int SB>
hipError_t launch_one(const TsArgs& a, unsigned grid, size_t smem, hipStream_t st)
{
	if constexpr (TS_MINE(K)) {
		hipLaunchKernelGGL((sketch_ts_kernel<K, SB>), dim3(grid), dim3(512), smem, st, a);
		return hipGetLastError();
	} else {
		return hipErrorInvalidValue;
	}
}
template <int K>
hipError_t set_limit_one(size_t smem)
{
	if constexpr (TS_MINE(K)) {
		hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_ts_kernel<K, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (rc == hipSuccess) rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_ts_kernel<K, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		return rc;
	} else {
		return hipSuccess;
	}
}
} // namespace

#define TS_CAT2(a, b) a##b
#define TS_CAT(a, b) TS_CAT2(a, b)
// this object's share: launch (false: k belongs to another part) and the LDS attribute of its kernels
bool TS_CAT(sketch_ts_launch_part, TS_PART)(const TsArgs& a, unsigned grid, size_t smem, hipStream_t st, hipError_t* rc)
{
	switch (a.k) {
#define X(k)                                                                                                     \
	case k:                                                                                                      \
		if (!TS_MINE(k)) return false;                                                                           \
		*rc = a.s_bits == 7 ? launch_one<k, 7>(a, grid, smem, st) : launch_one<k, 8>(a, grid, smem, st);         \
		return true;
		TS_FOR_K(X)
#undef X
	default:
		return false;
	}
}
hipError_t TS_CAT(sketch_ts_limit_part, TS_PART)(size_t smem)
{
	hipError_t rc = hipSuccess;
#define X(k) \
	if (rc == hipSuccess) rc = set_limit_one<k>(smem);
	TS_FOR_K(X)
#undef X
	return rc;
}

#if TS_PART == 0
#ifndef TS_ALL_PARTS
bool sketch_ts_launch_part1(const TsArgs&, unsigned, size_t, hipStream_t, hipError_t*);
bool sketch_ts_launch_part2(const TsArgs&, unsigned, size_t, hipStream_t, hipError_t*);
bool sketch_ts_launch_part3(const TsArgs&, unsigned, size_t, hipStream_t, hipError_t*);
hipError_t sketch_ts_limit_part1(size_t);
hipError_t sketch_ts_limit_part2(size_t);
hipError_t sketch_ts_limit_part3(size_t);
#endif

bool sketch_ts_supports(uint32_t k, uint32_t s_bits) { return k >= 12 && k <= 32 && s_bits >= 7; }

size_t sketch_ts_smem(uint32_t k) { return (size_t)((k + 3) / 4) * 4096 + 2 * (size_t)kTeamBytes; }

hipError_t launch_sketch_ts(const TsArgs& a, unsigned grid, hipStream_t st)
{
	if (!sketch_ts_supports(a.k, a.s_bits)) return hipErrorInvalidValue;
	const size_t smem = sketch_ts_smem(a.k);
	hipError_t rc = hipErrorInvalidValue;
	if (sketch_ts_launch_part0(a, grid, smem, st, &rc)) return rc;
#ifndef TS_ALL_PARTS
	if (sketch_ts_launch_part1(a, grid, smem, st, &rc)) return rc;
	if (sketch_ts_launch_part2(a, grid, smem, st, &rc)) return rc;
	if (sketch_ts_launch_part3(a, grid, smem, st, &rc)) return rc;
#endif
	return hipErrorInvalidValue;
}

hipError_t set_sketch_ts_smem_limit(size_t smem)
{
	hipError_t rc = sketch_ts_limit_part0(smem);
#ifndef TS_ALL_PARTS
	if (rc == hipSuccess) rc = sketch_ts_limit_part1(smem);
	if (rc == hipSuccess) rc = sketch_ts_limit_part2(smem);
	if (rc == hipSuccess) rc = sketch_ts_limit_part3(smem);
#endif
	return rc;
}
#endif

} // namespace ntc
