// ntc_sketch_ts.hip — K1c "tiled streaming" kernel: ntHash -> sample -> count for equal-length batches in the TILED
// slot layout (include/ntcard_hip.h, ntc_submit_tiled_device) on gfx950.
//
// What it computes is ntRead + ntComp (ntcard.cpp:132-158) for one k: for every window of k consecutive ACGTU bases
// the canonical ntHash (nthash.hpp:242-257,275-279), the two sampling patterns on its top bits, and one increment
// of t_Counter[sample][hash & (rBuck - 1)] per sampled window (as a hit-log entry, ntc_apply.hip), plus F1 = the
// number of such windows.  ntHashIterator's N semantics (ntHashIterator.hpp:59-86: a window that contains a
// non-ACGTU byte yields nothing) are handled here, exactly, without a second kernel.
//
// Formulation (see gen_ts.py): the sampling decision only needs the top sBits + 1 bits of min(fh, rh), and those
// live in the 31-bit rotating half of the hash (nthash.hpp:186-217).  That half is walked BIT-SLICED — one VGPR
// holds one bit of it for 32 reads, a wave carries a tile of 2048 reads, a rotate is a renaming of registers —
// and the ~2^(1-sBits) candidate windows are re-derived exactly (full 64-bit fh and rh, canonical min, ntComp's
// patterns, counter index) from the packed bases by a resolve stage.
//
// Data layout: tile t = reads [2048 t, 2048 t + 2048); chunk c = bases [16 c, 16 c + 16) of every read of the tile;
// the 16 raw bytes of (t, c, read r) sit at ((t C + c) 2048 + r) 16.  A chunk of a tile is 32 KiB of contiguous HBM
// and one coalesced 16-byte load per lane hands lane l the bases of read 64 m + l — already in the position-major
// order the bit-sliced walk consumes, so a tile STREAMS through the CU chunk by chunk (any read length, no 78 KB
// tile image, no window re-filling).
//
// Work decomposition: one 512-thread workgroup per CU = two TEAMS of four waves, each team streaming its own tile:
//   A1  loads a chunk (32 x 1 KiB), packs it to 2 bits per base + a dirty flag per 16 bytes, transposes the
//       32 x 32 bit matrix into 32 bit planes (registers only) and publishes the planes (8 KiB slot) and the packed
//       words (ring of 4 x 8 KiB) in LDS;
//   F,R walk one strand each: 31 state registers, one three-input v_bitop3_b32 per hash bit and base step + the
//       function planes of the incoming / outgoing base + a candidate test on the strand's own top bits; a step's
//       candidate plane goes to the strand's LDS queue as one ballot-compacted 8-byte item per lane;
//   A2  drains both queues 64 candidates at a time: closed-form fh / rh from the packed ring with a
//       4-bases-per-lookup table, keeps the candidate of the canonical strand only, applies ntComp, logs the counter
//       index; windows that touch a dirty 16-byte piece and F1's corrections are settled exactly from the raw bytes.
// The waves of a team only meet through monotonic LDS counters (no workgroup barrier after start-up).  On every SIMD
// a walker of one team shares the issue slots with an assistant of the other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ntc_kernels.hpp"

namespace ntc {

namespace {

#include "ntc_ts_gen.inc"

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t x, uint32_t y) { return (x & m) | (y & ~m); } // v_bfi_b32
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

// v_perm table indexed by (byte & 7): the only letter each index may stand for (nthash.hpp:16,32 trick), 0xff: none
constexpr uint32_t kExpS0 = 0x47ff5554u; // idx 7:'G' 6:- 5:'U' 4:'T'
constexpr uint32_t kExpS1 = 0x43ff41ffu; // idx 3:'C' 2:- 1:'A' 0:-

// 16 raw bytes -> 32 bits (2 per base, code2 = (ascii >> 1) & 3: A=0 C=1 T/U=2 G=3, base q in bits 2q+1:2q);
// dirty = 1 if some byte is not ACGTU/acgtu, else 0
__device__ __forceinline__ uint32_t pack16(const v4u32 v, uint32_t& dirty)
{
	const uint32_t p0 = (v.x & 0x06060606u) * 0x00820820u, p1 = (v.y & 0x06060606u) * 0x00820820u;
	const uint32_t p2 = (v.z & 0x06060606u) * 0x00820820u, p3 = (v.w & 0x06060606u) * 0x00820820u;
	const uint32_t lo = perm(p1, p0, 0x0c0c0703u);
	const uint32_t hi = perm(p3, p2, 0x07030c0cu);
	uint32_t x = perm(kExpS0, kExpS1, v.x & 0x07070707u) ^ v.x;
	x |= perm(kExpS0, kExpS1, v.y & 0x07070707u) ^ v.y;
	x |= perm(kExpS0, kExpS1, v.z & 0x07070707u) ^ v.z;
	x |= perm(kExpS0, kExpS1, v.w & 0x07070707u) ^ v.w;
	x &= 0xdfdfdfdfu;
	dirty = (x | (0u - x)) >> 31;
	return lo | hi;
}

// exact 16-bit mask of the non-ACGTU bytes of a 16-byte piece (bit q = byte q); rare paths only
__device__ __forceinline__ uint32_t inv4(uint32_t v)
{
	uint32_t x = (perm(kExpS0, kExpS1, v & 0x07070707u) ^ v) & 0xdfdfdfdfu;
	x |= x >> 4;
	x |= x >> 2;
	x |= x >> 1;
	x &= 0x01010101u;
	return ((x * 0x01020408u) >> 24) & 0xfu;
}
__device__ __forceinline__ uint32_t inv16(const v4u32 v) { return inv4(v.x) | (inv4(v.y) << 4) | (inv4(v.z) << 8) | (inv4(v.w) << 12); }

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
template <bool FWD>
__device__ __forceinline__ void ts_warm(uint32_t (&S)[31], uint32_t i0, uint32_t i1)
{
	if constexpr (FWD) ts_warm_F_k32(S, i0, i1);
	else ts_warm_R_k32(S, i0, i1);
}
template <bool FWD>
__device__ __forceinline__ void ts_main(uint32_t (&S)[31], uint32_t i0, uint32_t i1, uint32_t o0, uint32_t o1)
{
	if constexpr (FWD) ts_main_F_k32(S, i0, i1, o0, o1);
	else ts_main_R_k32(S, i0, i1, o0, o1);
}

// ---- LDS plan ------------------------------------------------------------------------------------------------
constexpr uint32_t kTile = kTileReads;
constexpr uint32_t kRing = 5;            // packed chunks kept: a block's candidates need chunks n-2 .. n, the walkers are one ahead, and a fifth
                                         // slot lets A1 park a chunk before the block two behind is resolved (no copy held in registers)
constexpr uint32_t kQCap = 512;          // candidate items per strand queue (power of two)
constexpr uint32_t kLCap = 128;          // A2's private ring of words with bits left
constexpr uint32_t kSCap = 128;          // suspects (candidates next to a dirty piece) waiting for their raw bytes
constexpr uint32_t kDCap = 128;          // dirty pieces (A1 -> A2)
constexpr uint32_t kOffPR = 0;                               // packed ring  [kRing][32][64] dwords
constexpr uint32_t kOffPL = kOffPR + kRing * 8192u;          // plane slot   [8][64] x 16 B
constexpr uint32_t kOffDB = kOffPL + 8192u;                  // dirty words  [kRing][64] dwords
constexpr uint32_t kOffQF = kOffDB + kRing * 256u;           // F queue      [kQCap + 64] x 8 B
constexpr uint32_t kOffQR = kOffQF + (kQCap + 64u) * 8u;     // R queue
constexpr uint32_t kOffLQ = kOffQR + (kQCap + 64u) * 8u;     // leftover ring [kLCap] x 8 B
constexpr uint32_t kOffSQ = kOffLQ + kLCap * 8u;             // suspects     [kSCap] x 16 B
constexpr uint32_t kOffDQ = kOffSQ + kSCap * 16u;            // dirty queue  [kDCap] x 8 B
constexpr uint32_t kOffCT = kOffDQ + kDCap * 8u;             // control words
constexpr uint32_t kTeamBytes = kOffCT + 128u;
// control words (all monotonic)
enum { C_PL_READY = 0, C_PL_TAKEN_F, C_PL_TAKEN_R, C_PR_READY, C_RESOLVED, C_QF_TAIL, C_QF_HEAD, C_QR_TAIL, C_QR_HEAD, C_BLK_F, C_BLK_R,
       C_DQ_TAIL, C_DQ_HEAD, C_BT_F = 16 /* 8 words: F's queue tail at the end of block b & 7 */, C_BT_R = 24 };

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
// spin until *p >= v (wrapping compare)
__device__ __forceinline__ void lds_wait_ge(const uint32_t* p, uint32_t v)
{
	while ((int32_t)(lds_peek(p) - v) < 0)
		__builtin_amdgcn_s_sleep(1);
}

} // namespace

template <int K, int SB>
__global__ __launch_bounds__(512, 2) void sketch_ts_kernel(const TsArgs a)
{
	static_assert(K == 32, "K1c step bodies are generated for k = 32 (gen_ts.py)");
	constexpr int KB = K / 16; // chunks between a base entering and leaving the window
	extern __shared__ __align__(16) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t wave = rfl((uint32_t)tid >> 6);
	// wave w and w + 4 share a SIMD: a walker of one team sits next to an assistant of the other
	const uint32_t team = wave < 4u ? wave >> 1 : ((wave >> 1) & 1u) ^ 1u;
	const uint32_t role = wave < 4u ? (wave & 1u) : 2u + (wave & 1u); // 0 F, 1 R, 2 A1, 3 A2
	const uint32_t t4_bytes = (uint32_t)(K / 4) * 4096u;
	unsigned char* const t4 = smem;
	unsigned char* const tb = smem + t4_bytes + team * kTeamBytes;
	uint32_t* const ctl = reinterpret_cast<uint32_t*>(tb + kOffCT);
	{
		const uint4* src = reinterpret_cast<const uint4*>(a.t4);
		for (uint32_t i = tid; i < t4_bytes / 16u; i += 512u)
			reinterpret_cast<uint4*>(t4)[i] = src[i];
		if (tid < 64) {
			reinterpret_cast<uint32_t*>(smem + t4_bytes + 0u * kTeamBytes + kOffCT)[tid & 31] = 0;
			reinterpret_cast<uint32_t*>(smem + t4_bytes + 1u * kTeamBytes + kOffCT)[tid & 31] = 0;
		}
	}
	__syncthreads(); // the only workgroup barrier: table and zeroed control words are in place
	const uint32_t C = a.n_chunks;
	const uint32_t W = a.read_len - (uint32_t)K + 1u; // windows per read
	const uint32_t n_teams = gridDim.x * 2u;
	const uint32_t team_g = blockIdx.x * 2u + team;
	const bool has_partial = (a.n_reads & (kTile - 1u)) != 0u;

#ifndef TS_NO_A1
	if (role == 2u) {
		// =============================== A1: load, pack, transpose, publish ===============================
		uint32_t n = 0;
		v4u32 raw[32];
		const uint32_t voff = (uint32_t)lane * 16u;
		auto issue = [&](uint32_t t, uint32_t c) {
			// wave-uniform chunk base (scalar registers) + one 32-bit lane offset: global_load ... saddr, no 64-bit address per load
			const unsigned char* p = a.tiles + ((size_t)t * C + c) * (size_t)(kTile * 16u);
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const unsigned char* pj = p + (size_t)j * 4096u; // scalar add: keeps the immediate offsets within 12 bits
#pragma unroll
				for (int i = 0; i < 4; ++i)
					raw[4 * j + i] = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(pj + (voff + (uint32_t)i * 1024u)));
			}
		};
		uint32_t dq_tail = 0;
		// this team's chunks as one flat sequence (one load site: two of them cost the register allocator its plan)
		const uint32_t my_tiles = team_g < a.n_tiles ? (a.n_tiles - team_g + n_teams - 1u) / n_teams : 0u;
		const uint32_t total = my_tiles * C;
		if (total != 0u) issue(team_g, 0);
		{
			uint32_t t = team_g, seq = 0, c = 0;
			for (; n < total; ++n) {
				uint32_t P[32];
				uint32_t dirtyword = 0;
#pragma unroll
				for (int m = 0; m < 32; ++m) {
					uint32_t d;
					P[m] = pack16(raw[m], d);
					dirtyword |= d << m;
				}
				__builtin_amdgcn_sched_barrier(0);
				// the next chunk's loads fly while this one is transposed and published
				const bool last_c = c + 1u == C;
				const uint32_t tn = last_c ? t + n_teams : t, cn = last_c ? 0u : c + 1u;
				const bool more = n + 1u < total;
				issue(more ? tn : t, more ? cn : c); // unconditional (a conditional load site makes all 128 registers a phi): the last one re-reads its chunk
				__builtin_amdgcn_sched_barrier(0);
				// the packed words overwrite chunk n - 5, which blocks <= n - 3 read: this wave is never that far ahead of A2 in practice
				if (n >= 3u) lds_wait_ge(ctl + C_RESOLVED, n - 2u);
				const uint32_t slot = n % kRing;
				{
					uint32_t* pr = reinterpret_cast<uint32_t*>(tb + kOffPR) + slot * 2048u + lane;
#pragma unroll
					for (int m = 0; m < 32; ++m)
						pr[m * 64] = P[m];
					reinterpret_cast<uint32_t*>(tb + kOffDB)[slot * 64u + lane] = dirtyword;
				}
				transpose32(P); // P[2 q + b] = bit b of the code of base 16 c + q, one bit per read
				lds_wait_ge(ctl + C_PL_TAKEN_F, n);
				lds_wait_ge(ctl + C_PL_TAKEN_R, n);
				{
					v4u32* pl = reinterpret_cast<v4u32*>(tb + kOffPL) + lane;
#pragma unroll
					for (int g = 0; g < 8; ++g) {
						v4u32 v;
						v.x = P[4 * g];
						v.y = P[4 * g + 1];
						v.z = P[4 * g + 2];
						v.w = P[4 * g + 3];
						pl[g * 64] = v;
					}
				}
				lds_publish(ctl + C_PL_READY, n + 1u);
				const uint64_t dm = ballot(dirtyword != 0u);
				if (dm != 0) { // rare: hand the dirty pieces to A2 (F1 corrections)
					const uint32_t cnt = (uint32_t)__popcll(dm);
					while ((int32_t)(dq_tail + cnt - lds_peek(ctl + C_DQ_HEAD) - kDCap) > 0)
						__builtin_amdgcn_s_sleep(1);
					if (dirtyword != 0u)
						reinterpret_cast<uint2*>(tb + kOffDQ)[(dq_tail + mbcnt(dm)) & (kDCap - 1u)] = make_uint2(dirtyword, (uint32_t)lane | ((seq & 1u) << 6) | (c << 7));
					dq_tail += cnt;
					lds_publish(ctl + C_DQ_TAIL, dq_tail);
				}
				lds_publish(ctl + C_PR_READY, n + 1u);
				seq += last_c ? 1u : 0u;
				t = tn;
				c = cn;
			}
		}
		return;
	}

#endif
#ifndef TS_NO_W
	if (role <= 1u) {
		// =============================== F / R: walk one strand of the 31-bit half ===============================
		auto walk = [&](auto fwd_c) {
			constexpr bool FWD = decltype(fwd_c)::value;
			uint2* const queue = reinterpret_cast<uint2*>(tb + (FWD ? kOffQF : kOffQR));
			uint32_t* const c_tail = ctl + (FWD ? C_QF_TAIL : C_QR_TAIL);
			const uint32_t* const c_head = ctl + (FWD ? C_QF_HEAD : C_QR_HEAD);
			uint32_t qtail = 0, qhead_c = 0; // wave-uniform
			uint32_t n = 0;
			uint32_t meta_tile = 0;
			auto push = [&](uint32_t h, uint32_t w) { // h: bit m set <=> read 64 m + lane is a candidate at window w
				// straight-line: lanes without a candidate store to their spare slot behind the queue
				h = w < W ? h : 0u;
				const uint64_t m = ballot(h != 0u);
				const uint32_t slot = (qtail + mbcnt(m)) & (kQCap - 1u);
				queue[h != 0u ? slot : kQCap + (uint32_t)lane] = make_uint2(h, meta_tile | (w << 11));
				qtail = rfl(qtail + (uint32_t)__popcll(m));
				if (__builtin_expect(qtail - qhead_c > kQCap - 64u, 0)) { // the next step may not fit: let A2 catch up
					lds_publish(c_tail, qtail);
					do {
						qhead_c = lds_peek(c_head);
						if (qtail - qhead_c <= kQCap - 64u) break;
						__builtin_amdgcn_s_sleep(1);
					} while (true);
				}
			};
			for (uint32_t t = team_g, seq = 0; t < a.n_tiles; t += n_teams, ++seq) {
				meta_tile = (uint32_t)lane | (FWD ? 0u : 64u) | ((has_partial && t == a.n_tiles - 1u) ? 128u : 0u) | (((seq * C) % kRing) << 8);
				uint32_t S[31];
#pragma unroll
				for (int j = 0; j < 31; ++j)
					S[j] = 0;
				uint32_t H[KB][32];
#pragma unroll
				for (int b = 0; b < KB; ++b)
#pragma unroll
					for (int i = 0; i < 32; ++i)
						H[b][i] = 0;
#pragma unroll 1
				for (uint32_t c = 0; c < C; ++c, ++n) {
					uint32_t I[32];
					lds_wait_ge(ctl + C_PL_READY, n + 1u);
					{
						const v4u32* pl = reinterpret_cast<const v4u32*>(tb + kOffPL) + lane;
						v4u32 v[8];
#pragma unroll
						for (int g = 0; g < 8; ++g)
							v[g] = pl[g * 64];
#pragma unroll
						for (int g = 0; g < 8; ++g) {
							I[4 * g] = v[g].x;
							I[4 * g + 1] = v[g].y;
							I[4 * g + 2] = v[g].z;
							I[4 * g + 3] = v[g].w;
						}
					}
					lds_publish(ctl + (FWD ? C_PL_TAKEN_F : C_PL_TAKEN_R), n + 1u);
					if (c < (uint32_t)KB) { // window filling: no outgoing base; the last step completes window 0
#pragma unroll
						for (int q = 0; q < 16; ++q) {
							ts_warm<FWD>(S, I[2 * q], I[2 * q + 1]);
							pin31(S);
						}
						if (c == (uint32_t)KB - 1u) push(ts_cand<SB>(S), 0u);
					} else {
						const uint32_t w0 = 16u * c - (uint32_t)K + 1u;
#pragma unroll
						for (int q = 0; q < 16; ++q) {
							ts_main<FWD>(S, I[2 * q], I[2 * q + 1], H[0][2 * q], H[0][2 * q + 1]);
							pin31(S);
							push(ts_cand<SB>(S), w0 + (uint32_t)q);
						}
					}
#pragma unroll
					for (int hh = 0; hh + 1 < KB; ++hh)
#pragma unroll
						for (int i = 0; i < 32; ++i)
							H[hh][i] = H[hh + 1][i];
#pragma unroll
					for (int i = 0; i < 32; ++i)
						H[KB - 1][i] = I[i];
					// block n is complete: its items end at qtail
					lds_publish(ctl + (FWD ? C_BT_F : C_BT_R) + (n & 7u), qtail);
					lds_publish(c_tail, qtail);
					lds_publish(ctl + (FWD ? C_BLK_F : C_BLK_R), n + 1u);
				}
			}
		};
		if (role == 0u) walk(std::true_type{});
		else walk(std::false_type{});
		return;
	}

#endif
#ifndef TS_NO_A2
	// =============================== A2: resolve candidates, log hits, settle dirty pieces, F1 ===============================
	{
		const uint32_t rmask = (1u << a.r_bits) - 1u, rbuck = 1u << a.r_bits, s_bits = a.s_bits;
		// ---- hit log (see ntc_sketch_hf.hip): this wave's regions are team_g, team_g + n_teams, ... ----
		const bool use_log = a.log_regions != 0 && (a.log_mode == nullptr || rfl(*a.log_mode) == 0u);
		uint32_t lreg = team_g, lfill = 0;
		if (use_log && lreg < a.log_regions) lfill = rfl(a.log_fill[lreg]);
		auto log_emit = [&](bool hit, uint32_t key) {
			const uint64_t m = ballot(hit);
			if (m == 0) return;
			if (!use_log) {
				if (hit) atomicAdd(a.sketch0 + key, 1u);
				return;
			}
			const uint32_t c = (uint32_t)__popcll(m);
			while (lreg < a.log_regions && c > a.log_region_cap - lfill) {
				if (lane == 0) a.log_fill[lreg] = lfill;
				lreg += n_teams;
				lfill = lreg < a.log_regions ? rfl(a.log_fill[lreg]) : 0u;
			}
			if (lreg < a.log_regions) {
				if (hit) a.log[(uint64_t)lreg * a.log_region_cap + lfill + mbcnt(m)] = key;
				lfill += c;
			} else if (hit) {
				atomicAdd(a.sketch0 + key, 1u);
			}
		};
		const uint32_t* const pr = reinterpret_cast<const uint32_t*>(tb + kOffPR);
		const uint32_t* const db = reinterpret_cast<const uint32_t*>(tb + kOffDB);
		uint2* const qF = reinterpret_cast<uint2*>(tb + kOffQF);
		uint2* const qR = reinterpret_cast<uint2*>(tb + kOffQR);
		uint2* const lq = reinterpret_cast<uint2*>(tb + kOffLQ);
		uint4* const sq = reinterpret_cast<uint4*>(tb + kOffSQ);
		const uint2* const dq = reinterpret_cast<const uint2*>(tb + kOffDQ);
		uint32_t hF = 0, hR = 0, lq_head = 0, lq_fill = 0, sq_fill = 0, dq_head = 0; // wave-uniform
		uint64_t f1_add = 0;  // wave-uniform: reads x windows
		uint32_t f1_sub = 0;  // per lane: windows lost to non-ACGTU bytes
		const uint32_t n_valid_last = has_partial ? (uint32_t)(a.n_reads & (kTile - 1u)) : kTile;
		auto raw_piece = [&](uint32_t t, uint32_t c, uint32_t r) {
			return *reinterpret_cast<const v4u32*>(a.tiles + (((size_t)t * C + c) * kTile + r) * 16u);
		};
		// ---- suspects: candidates whose window touches a dirty piece; exact validity from the raw bytes ----
		auto flush_suspects = [&]() {
			for (uint32_t base = 0; base < sq_fill; base += 64u) {
				const bool act = base + (uint32_t)lane < sq_fill;
				const uint4 s = sq[act ? base + (uint32_t)lane : 0u];
				const uint32_t r = s.y, w = s.z, t = s.w, c0 = w >> 4;
				uint32_t m0 = 0, m1 = 0, m2 = 0;
				if (act) {
					m0 = inv16(raw_piece(t, c0, r));
					m1 = inv16(raw_piece(t, c0 + 1u, r));
					if (c0 + 2u < C) m2 = inv16(raw_piece(t, c0 + 2u, r));
				}
				const uint64_t all = (uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32);
				const bool ok = act && (uint32_t)(all >> (w & 15u)) == 0u; // no non-ACGTU byte among the window's 32 bases (k = 32)
				log_emit(ok, s.x);
			}
			sq_fill = 0;
		};
		// ---- dirty pieces: F1 loses the windows whose RIGHTMOST non-ACGTU byte lies in the piece.  A1 is at most one tile
		// ahead of this wave, so one bit of the tile sequence number identifies an item's tile ----
		auto drain_dirty = [&](uint32_t t_cur, uint32_t seq_cur) {
			const uint32_t tail = lds_peek(ctl + C_DQ_TAIL);
			while (dq_head != tail) {
				const uint32_t cnt = tail - dq_head < 64u ? tail - dq_head : 64u;
				uint2 it = make_uint2(0u, 0u);
				if ((uint32_t)lane < cnt) it = dq[(dq_head + (uint32_t)lane) & (kDCap - 1u)];
				uint32_t word = it.x;
				const uint32_t l0 = it.y & 63u, c = it.y >> 7;
				const uint32_t t = ((it.y >> 6) & 1u) == (seq_cur & 1u) ? t_cur : t_cur + n_teams;
				while (word != 0u) {
					const uint32_t r = (uint32_t)__builtin_ctz(word) * 64u + l0;
					word &= word - 1u;
					const uint32_t A = inv16(raw_piece(t, c, r));
					uint32_t B = 0;
					if (c + 1u < C) B = inv16(raw_piece(t, c + 1u, r));
					if (c + 2u < C) B |= inv16(raw_piece(t, c + 2u, r)) << 16;
					if (A != 0u) {
						// window w = 16 c - (k - 1) + j ends at base 16 c + j: it holds a byte of A iff lo <= j <= hi + k - 1, and no later
						// non-ACGTU byte iff j <= 15 + ctz(B) (k <= 32 + 1: B covers the 32 bases behind the piece); 0 <= w < W
						const int lo = __builtin_ctz(A), hi = 31 - __builtin_clz(A);
						const int tzb = B ? __builtin_ctz(B) : 32;
						const int jmin = max(lo, (int)K - 1 - 16 * (int)c);
						const int jmax = min(min(hi + (int)K - 1, 15 + tzb), (int)W + (int)K - 2 - 16 * (int)c);
						if (jmax >= jmin) f1_sub += (uint32_t)(jmax - jmin + 1);
					}
				}
				dq_head += cnt;
				lds_publish(ctl + C_DQ_HEAD, dq_head);
			}
		};
		// ---- one round: up to 64 items -> full canonical hash -> ntComp -> log ----
		auto round = [&](uint32_t nL, uint32_t nF, uint32_t nR, uint32_t t) {
			const uint32_t cnt = nL + nF + nR;
			const bool act = (uint32_t)lane < cnt;
			uint2 it;
			if ((uint32_t)lane < nL) it = lq[(lq_head + (uint32_t)lane) & (kLCap - 1u)];
			else if ((uint32_t)lane < nL + nF) it = qF[(hF + (uint32_t)lane - nL) & (kQCap - 1u)];
			else it = qR[(hR + (uint32_t)lane - nL - nF) & (kQCap - 1u)];
			if (!act) it = make_uint2(0u, 0u);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the items are in registers: their queue slots may be reused
			lq_head += nL;
			lq_fill -= nL;
			hF += nF;
			hR += nR;
			if (nF) lds_publish(ctl + C_QF_HEAD, hF);
			if (nR) lds_publish(ctl + C_QR_HEAD, hR);
			// words with more than one bit go back in line (private ring: at most 64 stay after a round that took them first)
			const uint32_t rest = it.x & (it.x - 1u);
			const uint64_t rm = ballot(rest != 0u);
			if (rm != 0) {
				if (rest != 0u) lq[(lq_head + lq_fill + mbcnt(rm)) & (kLCap - 1u)] = make_uint2(rest, it.y);
				lq_fill += (uint32_t)__popcll(rm);
			}
			const uint32_t m = (uint32_t)__builtin_ctz(it.x | 0x80000000u);
			const uint32_t l0 = it.y & 63u, w = it.y >> 11, nb = (it.y >> 8) & 7u;
			const bool from_r = (it.y & 64u) != 0u;
			const uint32_t r = m * 64u + l0;
			const uint32_t c0 = w >> 4, sh = (w & 15u) * 2u;
			const uint32_t col = m * 64u + l0;
			const uint32_t s0 = (nb + c0) % kRing, s1 = s0 + 1u == kRing ? 0u : s0 + 1u, s2 = s1 + 1u == kRing ? 0u : s1 + 1u;
			const uint32_t d0 = pr[s0 * 2048u + col], d1 = pr[s1 * 2048u + col], d2 = pr[s2 * 2048u + col];
			// dirty words of the pieces the window touches (the third only when the window is not chunk-aligned)
			const uint32_t dd = (db[s0 * 64u + l0] | db[s1 * 64u + l0] | (sh != 0u ? db[s2 * 64u + l0] : 0u)) >> m;
			uint32_t toff[K / 4];
#pragma unroll
			for (int i = 0; i < KB; ++i) {
				const uint32_t x = i == 0 ? alignbit(d1, d0, sh) : alignbit(d2, d1, sh); // 16 bases of the window
#pragma unroll
				for (int g = 0; g < 4; ++g)
					toff[i * 4 + g] = (uint32_t)(i * 4 + g) * 4096u + ((x >> (8 * g)) & 0xffu) * 16u;
			}
			__builtin_amdgcn_sched_barrier(0);
			v4u32 tv[K / 4];
#pragma unroll
			for (int j = 0; j < K / 4; ++j)
				tv[j] = *reinterpret_cast<const v4u32*>(t4 + toff[j]);
			__builtin_amdgcn_sched_barrier(0);
			uint32_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
#pragma unroll
			for (int j = 0; j < K / 4; ++j) {
				flo ^= tv[j].x;
				fhi ^= tv[j].y;
				rlo ^= tv[j].z;
				rhi ^= tv[j].w;
			}
			const bool rev = (rhi < fhi) | ((rhi == fhi) & (rlo < flo)); // nthash.hpp:275-279
			const uint32_t hi = rev ? rhi : fhi, lo = rev ? rlo : flo;
			// ntComp (ntcard.cpp:132-145) on the canonical value; sample 1 wins when both match
			const bool c1 = (hi >> (32 - s_bits)) == ((1u << (s_bits - 1)) - 1u);
			const bool c0m = (hi >> (31 - s_bits)) == 1u;
			// the candidate of the canonical strand only (both strands may have flagged the window)
			bool hit = act & (rev == from_r) & (c0m | c1);
			if ((it.y & 128u) != 0u) hit &= r < n_valid_last; // slots behind the last read of the batch
			const uint32_t key = a.key_base + (lo & rmask) + (c1 ? rbuck : 0u);
			const bool suspect = hit & ((dd & 1u) != 0u);
			const uint64_t sm = ballot(suspect);
			if (sm != 0) { // rare: the window touches a 16-byte piece with a non-ACGTU byte somewhere
				if (suspect) sq[sq_fill + mbcnt(sm)] = make_uint4(key, r, w, t);
				sq_fill += (uint32_t)__popcll(sm);
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				if (sq_fill > kSCap - 64u) flush_suspects();
			}
#ifdef TS_DEBUG
			if (act && a.dbg) {
				const uint32_t idx = atomicAdd(a.dbg, 1u);
				uint32_t* o = a.dbg + 16 + 12 * (size_t)idx;
				o[0] = r; o[1] = w; o[2] = it.x; o[3] = it.y; o[4] = d0; o[5] = d1; o[6] = d2; o[7] = flo; o[8] = fhi; o[9] = rlo; o[10] = rhi;
				o[11] = (rev ? 1u : 0u) | (from_r ? 2u : 0u) | (hit ? 4u : 0u);
			}
#endif
			log_emit(hit & !suspect, key);
		};
		uint32_t n_res = 0;
		for (uint32_t t = team_g, seq = 0; t < a.n_tiles; t += n_teams, ++seq) {
			for (uint32_t c = 0; c < C; ++c, ++n_res) {
				while ((int32_t)(lds_peek(ctl + C_PR_READY) - (n_res + 1u)) < 0) { // A1 may be waiting for room in the dirty queue
					drain_dirty(t, seq);
					__builtin_amdgcn_s_sleep(1);
				}
				while (true) {
					// block counters first, then tails: a tail read after "block complete" covers the whole block
					const uint32_t bF = lds_peek(ctl + C_BLK_F), bR = lds_peek(ctl + C_BLK_R);
					const bool doneF = (int32_t)(bF - (n_res + 1u)) >= 0, doneR = (int32_t)(bR - (n_res + 1u)) >= 0;
					const uint32_t tF = doneF ? lds_peek(ctl + C_BT_F + (n_res & 7u)) : lds_peek(ctl + C_QF_TAIL);
					const uint32_t tR = doneR ? lds_peek(ctl + C_BT_R + (n_res & 7u)) : lds_peek(ctl + C_QR_TAIL);
					const uint32_t aF = tF - hF, aR = tR - hR;
					const uint32_t avail = lq_fill + aF + aR;
					const bool complete = doneF & doneR;
					if (avail >= 64u || (complete && avail != 0u)) {
						const uint32_t nL = lq_fill < 64u ? lq_fill : 64u;
						const uint32_t nF = aF < 64u - nL ? aF : 64u - nL;
						const uint32_t nR = aR < 64u - nL - nF ? aR : 64u - nL - nF;
						round(nL, nF, nR, t);
					} else if (complete) {
						break;
					} else {
						drain_dirty(t, seq);
						__builtin_amdgcn_s_sleep(1);
					}
				}
				lds_publish(ctl + C_RESOLVED, n_res + 1u);
			}
			drain_dirty(t, seq); // every piece of tile t is in the queue by now (and perhaps the first of the next tile)
			f1_add += (uint64_t)((has_partial && t == a.n_tiles - 1u) ? n_valid_last : kTile) * W;
		}
		flush_suspects();
		if (use_log && lane == 0 && lreg < a.log_regions) a.log_fill[lreg] = lfill;
		// F1 (ntcard.cpp:154): one per window without a non-ACGTU byte
		uint32_t sub = f1_sub;
		for (int o = 32; o > 0; o >>= 1)
			sub += (uint32_t)__shfl_xor((int)sub, o);
		if (lane == 0 && f1_add != 0) atomicAdd(a.f1, (unsigned long long)(f1_add - sub));
	}
#endif
}

namespace {
template <int K, int SB>
hipError_t launch_one(const TsArgs& a, unsigned grid, size_t smem, hipStream_t st)
{
	hipLaunchKernelGGL((sketch_ts_kernel<K, SB>), dim3(grid), dim3(512), smem, st, a);
	return hipGetLastError();
}
} // namespace

bool sketch_ts_supports(uint32_t k, uint32_t s_bits) { return k == 32 && s_bits >= 7; }

size_t sketch_ts_smem(uint32_t k) { return (size_t)(k / 4) * 4096 + 2 * (size_t)kTeamBytes; }

hipError_t launch_sketch_ts(const TsArgs& a, unsigned grid, hipStream_t st)
{
	const size_t smem = sketch_ts_smem(a.k);
	if (a.k != 32) return hipErrorInvalidValue;
	if (a.s_bits < 7) return hipErrorInvalidValue;
	return a.s_bits == 7 ? launch_one<32, 7>(a, grid, smem, st) : launch_one<32, 8>(a, grid, smem, st);
}

hipError_t set_sketch_ts_smem_limit(size_t smem)
{
	const void* fns[] = { reinterpret_cast<const void*>(&sketch_ts_kernel<32, 7>), reinterpret_cast<const void*>(&sketch_ts_kernel<32, 8>) };
	for (const void* f : fns) {
		const hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (rc != hipSuccess) return rc;
	}
	return hipSuccess;
}

} // namespace ntc
