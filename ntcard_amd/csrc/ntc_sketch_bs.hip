// ntc_sketch_bs.hip — K1b "bit-sliced filter": ntHash -> sample -> count for equal-length batches on gfx950.
//
// The lane-per-read kernel (ntc_sketch_hf.hip) is bound by VALU issue: ~12 wave instructions per base step for 64
// reads.  Here the sampling decision of ntComp (ntcard.cpp:135-138: the top sBits+1 bits of min(fh, rh)) is
// evaluated BIT-SLICED: one VGPR holds one bit of the 31-bit rotating half of the hash (nthash.hpp:186-217) for 32
// reads, a wave carries 2048 reads, a rotate is a renaming of registers, and one base step of NTF64 + NTR64
// (nthash.hpp:242-257) costs 62 three-input XORs + 10 function planes for 2048 reads instead of ~10 instructions
// for 64 (gen_bs.py).  Everything that is only needed for the ~2^(1-sBits) sampled windows — the full 64-bit
// canonical hash (nthash.hpp:220-239,275-279), ntComp's two patterns, the counter index — is re-derived exactly
// from the bases by the resolve stage, as in K1.
//
// Work decomposition (one 512-thread workgroup per CU: on every SIMD one WALKER wave and one HELPER wave, 256 VGPRs each;
// a single wave cannot fill a SIMD's VALU issue slots, two with complementary work nearly do):
//   * a TILE is 2048 consecutive slots.  The helpers load its raw bytes once with coalesced 16-byte loads, pack them
//     to 2 bits per base (code2 = (ascii >> 1) & 3: A=0 C=1 T/U=2 G=3) and park the result in their registers
//     while the previous tile is being walked; at the tile switch the packed image (one dword per 16 bases, same
//     geometry as the slots) is dumped to LDS (77.8 KB for 150 bp reads).  The register file of the helper waves is
//     the double buffer the LDS has no room for;
//   * the four walkers walk the SAME tile, each one quarter of the window positions (segment = 16*nq windows,
//     preceded by the k-1 window-filling steps); a lane's 32 reads are 64*i + lane, i = 0..31; per 16-base chunk
//     the lane fetches its 32 packed words from LDS and transposes the 32x32 bit matrix into 32 bit planes;
//   * after every step the walker stores its hit plane (one bit per read) as ONE ballot-compacted 8-byte item
//     (plane word, lane, window) per lane with a hit in its LDS queue; every 16 steps the queue is drained 64 items at a
//     time: a resolve round takes the lowest set bit of every item, recomputes the full canonical hash with a
//     4-bases-per-lookup closed-form table and sends the counter index to the hit log (ntc_apply.hip);
//   * a read with any non-ACGTU byte is NOT handled here: it is left out of F1 and of the sketch and its slot
//     index is appended to a device list that the lane-per-read kernel processes right after (gather mode), which
//     keeps ntHashIterator's N semantics (ntHashIterator.hpp:59-86) in one place.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ntc_kernels.hpp"

namespace ntc {

namespace {

#include "ntc_bs_gen.inc"

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t x, uint32_t y) { return (x & m) | (y & ~m); } // v_bfi_b32

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

// v_perm table indexed by (byte & 7): the only letter each index may stand for (nthash.hpp:16,32 trick), 0xff: none
constexpr uint32_t kExpS0 = 0x47ff5554u; // idx 7:'G' 6:- 5:'U' 4:'T'
constexpr uint32_t kExpS1 = 0x43ff41ffu; // idx 3:'C' 2:- 1:'A' 0:-

// 16 raw bytes -> 32 bits (2 per base); dirty = 1 if some byte is not ACGTU/acgtu, else 0
__device__ __forceinline__ uint32_t pack16(const uint4 v, uint32_t& dirty)
{
	// code2 of 4 bytes lands in the top byte of (w & 0x06060606) * 0x00820820 (fields 2 bits wide, no carries)
	const uint32_t p0 = (v.x & 0x06060606u) * 0x00820820u, p1 = (v.y & 0x06060606u) * 0x00820820u;
	const uint32_t p2 = (v.z & 0x06060606u) * 0x00820820u, p3 = (v.w & 0x06060606u) * 0x00820820u;
	const uint32_t lo = perm(p1, p0, 0x0c0c0703u); // byte0 = p0.byte3, byte1 = p1.byte3
	const uint32_t hi = perm(p3, p2, 0x07030c0cu); // byte2 = p2.byte3, byte3 = p3.byte3
	uint32_t x = perm(kExpS0, kExpS1, v.x & 0x07070707u) ^ v.x;
	x |= perm(kExpS0, kExpS1, v.y & 0x07070707u) ^ v.y;
	x |= perm(kExpS0, kExpS1, v.z & 0x07070707u) ^ v.z;
	x |= perm(kExpS0, kExpS1, v.w & 0x07070707u) ^ v.w;
	x &= 0xdfdfdfdfu;
	dirty = (x | (0u - x)) >> 31; // branch-free "x != 0" (a v_cmp + v_cndmask pair issues far slower)
	return lo | hi;
}

// in-place transpose of a 32 x 32 bit matrix held in 32 registers (row i = A[i], column c = bit c): five butterfly
// stages, each swapping the off-diagonal J x J blocks of every 2J x 2J block (two v_bfi_b32 + two shifts per pair)
template <int J>
__device__ __forceinline__ void transpose_stage(uint32_t (&A)[32])
{
	constexpr uint32_t m = J == 16 ? 0x0000ffffu : J == 8 ? 0x00ff00ffu : J == 4 ? 0x0f0f0f0fu : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
	for (int k = 0; k < 32; ++k) {
		if ((k & J) == 0) {
			const uint32_t x = A[k], y = A[k + J];
			if constexpr (J == 16) { // whole bytes move: one v_perm_b32 per output instead of a shift + v_bfi_b32
				A[k] = perm(y, x, 0x05040100u);     // x.b0 x.b1 y.b0 y.b1
				A[k + J] = perm(y, x, 0x07060302u); // x.b2 x.b3 y.b2 y.b3
			} else if constexpr (J == 8) {
				A[k] = perm(y, x, 0x06020400u);     // x.b0 y.b0 x.b2 y.b2
				A[k + J] = perm(y, x, 0x07030501u); // x.b1 y.b1 x.b3 y.b3
			} else {
				A[k] = bfi(m, x, y << J);
				A[k + J] = bfi(m, x >> J, y);
			}
		}
	}
}
__device__ __forceinline__ void transpose32(uint32_t (&A)[32])
{
	transpose_stage<16>(A);
	transpose_stage<8>(A);
	transpose_stage<4>(A);
	transpose_stage<2>(A);
	transpose_stage<1>(A);
}

// materialise the strand registers here (stops the compiler from sinking the tail of a step past the drain loop,
// which keeps that step's inputs alive across it and spills)
__device__ __forceinline__ void pin31(uint32_t (&X)[31])
{
	asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]), "+v"(X[8]), "+v"(X[9]), "+v"(X[10]));
	asm volatile("" : "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]), "+v"(X[16]), "+v"(X[17]), "+v"(X[18]), "+v"(X[19]), "+v"(X[20]));
	asm volatile("" : "+v"(X[21]), "+v"(X[22]), "+v"(X[23]), "+v"(X[24]), "+v"(X[25]), "+v"(X[26]), "+v"(X[27]), "+v"(X[28]), "+v"(X[29]), "+v"(X[30]));
}

#ifdef NTC_BS_TIMERS // instrumentation build (tools/ab_build.sh): per-phase cycle counts, summed over waves
#define BS_T(var) const uint64_t var = __builtin_readcyclecounter()
#define BS_ACC(slot, t0, t1) tacc[slot] += (t1) - (t0)
#else
#define BS_T(var)
#define BS_ACC(slot, t0, t1)
#endif

// kTileReads (2048) comes from ntc_kernels.hpp
constexpr int kGroupLoads = 10;  // loads per staging group
constexpr int kGroups = 8;       // 80 loads of 1 KiB per helper and tile: strides up to 160 B
constexpr uint32_t kQueueCap = 1088; // queue items per walker: 63 left over from the previous block + 16 steps x 64 lanes

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (no LDS round trip); used for the redo list only
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

} // namespace

template <int K, int SB>
__global__ __launch_bounds__(512, 2) void sketch_bs_kernel(const BsArgs a)
{
	static_assert(K == 32, "K1b step bodies are generated for k = 32 only (gen_bs.py): any other K would walk nothing");
	constexpr int KB = K / 16; // window-filling blocks; block b consumes bases [16 b, 16 b + 16) of the segment
	extern __shared__ __align__(16) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const bool walker = wave < 4;
	const uint32_t part = (uint32_t)wave & 3u; // walker: segment of the window positions; helper: quarter of the tile
	const uint32_t stride = a.stride, s4 = stride >> 2;
	const uint32_t tile_dw = 128u * stride; // packed dwords per tile (one per 16 raw bytes)
	const uint32_t qd = 32u * stride;       // chunks (= packed dwords) per helper quarter
	const uint32_t cbm_words = tile_dw >> 5;
	// LDS: [packed tile + 64 dwords][closed-form table][2 chunk-dirty bitmaps][read-dirty bitmap][4 item queues]
	uint32_t* const tile = reinterpret_cast<uint32_t*>(smem);
	unsigned char* const t4 = smem + (size_t)(tile_dw + 64u) * 4u;
	const uint32_t t4_bytes = (uint32_t)(K / 4) * 4096u;
	uint32_t* const cbm0 = reinterpret_cast<uint32_t*>(t4 + t4_bytes);
	uint32_t* const cbm1 = cbm0 + cbm_words;
	uint32_t* const rdirty = cbm1 + cbm_words; // 64 words
	uint2* const queue = reinterpret_cast<uint2*>(rdirty + 64) + part * (kQueueCap + 64u); // 8-byte aligned: every size before it is a multiple of 8; + 64 spare slots
	{
		const uint4* src = reinterpret_cast<const uint4*>(a.t4);
		for (uint32_t i = tid; i < t4_bytes / 16u; i += 512u)
			reinterpret_cast<uint4*>(t4)[i] = src[i];
		for (uint32_t i = tid; i < 2u * cbm_words + 64u; i += 512u)
			cbm0[i] = 0;
		if (tid < 64) tile[tile_dw + tid] = 0;
	}
	__syncthreads(); // tables and zeroed bitmaps are in place
	const uint32_t W = a.read_len - (uint32_t)K + 1u; // windows per read
#ifdef NTC_BS_TIMERS
	uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

	if (!walker) {
		// ======================= HELPER: stages quarter `part` of the next tile while the current one is walked =======================
		const uint32_t magic = 0xffffffffu / stride + 1u; // x / stride = umulhi(x, magic) for x < 2^19
		uint32_t park[kGroups * kGroupLoads];
		const uint32_t nl = __builtin_amdgcn_readfirstlane(stride >> 1); // 1 KiB loads per quarter (qd / 64)
		uint64_t f1_acc = 0;
		uint32_t par = 1; // the tile staged during an iteration lands in bitmap par ^ 1
		// iteration -1 has no current tile: it only stages the workgroup's first one
		for (int64_t ts = (int64_t)blockIdx.x - (int64_t)gridDim.x; ts < (int64_t)a.n_tiles; ts += gridDim.x, par ^= 1u) {
			const bool has_cur = ts >= 0;
			const uint64_t t = (uint64_t)(has_cur ? ts : 0);
			const uint64_t tn = (uint64_t)(ts + (int64_t)gridDim.x);
			const bool has_next = tn < a.n_tiles;
			uint32_t* const cbm_cur = par ? cbm1 : cbm0;
			uint32_t* const cbm_nxt = par ? cbm0 : cbm1;
			if (has_cur) {
				BS_T(ts0);
				__syncthreads(); // B1: the previous tile is finished (walked and resolved)
				BS_T(ts1);
				BS_ACC(0, ts0, ts1);
#pragma unroll
				for (int c = 0; c < kGroups * kGroupLoads; ++c) {
					const uint32_t i = (uint32_t)c * 64u + (uint32_t)lane;
					tile[(uint32_t)c < nl ? part * qd + i : tile_dw + (uint32_t)lane] = park[c]; // loads past the quarter land in the slack behind the image
				}
				if (part == 0) rdirty[lane] = 0;
				__syncthreads(); // B2
				// chunk-dirty -> read-dirty: a dirty 16-byte chunk taints the read(s) it overlaps (pad bytes included: a
				// read tainted needlessly is merely processed by the other kernel)
				for (uint32_t wi = (uint32_t)lane; wi < (qd >> 5); wi += 64u) {
					const uint32_t widx = part * (qd >> 5) + wi;
					uint32_t word = cbm_cur[widx];
					if (word != 0u) {
						cbm_cur[widx] = 0;
						while (word != 0u) {
							const uint32_t g = widx * 32u + (uint32_t)__builtin_ctz(word);
							word &= word - 1u;
							const uint32_t r0 = __umulhi(g * 16u, magic), r1 = __umulhi(g * 16u + 15u, magic);
							atomicOr(&rdirty[r0 >> 5], 1u << (r0 & 31u));
							if (r1 < (uint32_t)kTileReads) atomicOr(&rdirty[r1 >> 5], 1u << (r1 & 31u));
						}
					}
				}
				__syncthreads(); // B3: the tile image and its read-dirty bits are final; the walkers start
				if (part == 0) { // F1 of the clean reads (ntcard.cpp:154: one per window); the dirty ones go to the redo list
					const uint32_t word = rdirty[lane];
					const uint32_t cnt = (uint32_t)__popc(word);
					const uint32_t incl = wave_scan(cnt);
					const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
					f1_acc += (uint64_t)W * (uint32_t)(kTileReads - total);
					if (total != 0u) {
						uint32_t base = 0;
						if (lane == 0) base = atomicAdd(a.redo_count, total);
						base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + incl - cnt;
						uint32_t w2 = word;
						while (w2 != 0u) {
							a.redo_list[base++] = (uint64_t)(uintptr_t)a.slots + ((uint64_t)(t * kTileReads) + (uint32_t)lane * 32u + (uint32_t)__builtin_ctz(w2)) * stride;
							w2 &= w2 - 1u;
						}
					}
				}
			}
			BS_T(ts2);
			if (has_next) {
				// staging of the next tile: 8 groups of 10 loads, group g + 1 in flight while group g is packed
				const unsigned char* src = a.slots + tn * ((uint64_t)kTileReads * stride) + (uint64_t)part * 16u * qd;
				uint4 raw[2][kGroupLoads];
				uint32_t vlane = (uint32_t)lane;
				auto issue = [&](auto gc) {
					constexpr int g = decltype(gc)::value;
#pragma unroll
					for (int c = 0; c < kGroupLoads; ++c) {
						const uint32_t i = (uint32_t)(g * kGroupLoads + c) * 64u + vlane;
						// strides >= 128 B: the first 64 loads always exist (one address register serves them all); a later
						// load past the quarter re-reads its last chunk (no branch)
						raw[g & 1][c] = *reinterpret_cast<const uint4*>(src + 16u * (g * kGroupLoads + c < 64 || i < qd ? i : qd - 1u));
					}
				};
				auto pack = [&](auto gc) {
					constexpr int g = decltype(gc)::value;
					uint32_t dm = 0; // bit (kGroupLoads - 1 - c): chunk c of this group holds a non-ACGTU byte
#pragma unroll
					for (int c = 0; c < kGroupLoads; ++c) {
						uint32_t dirty;
						park[g * kGroupLoads + c] = pack16(raw[g & 1][c], dirty);
						dm = (dm << 1) | dirty;
					}
					while (dm != 0u) { // rare: note the chunk; the read(s) it belongs to are sorted out at the tile switch
						const uint32_t c = (uint32_t)(kGroupLoads - 1) - (uint32_t)__builtin_ctz(dm);
						dm &= dm - 1u;
						const uint32_t i = ((uint32_t)(g * kGroupLoads) + c) * 64u + (uint32_t)lane;
						const uint32_t gi = part * qd + i;
						if (i < qd) atomicOr(&cbm_nxt[gi >> 5], 1u << (gi & 31u));
					}
					// keep the pipeline two groups deep, not deeper (every group in flight costs 40 VGPRs): the addresses of
					// group g + 2 "depend" on this group's packed words
					uint32_t* q = park + g * kGroupLoads;
					uint32_t vl = vlane;
					asm volatile("" : "+v"(vl) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]), "v"(q[8]), "v"(q[9]));
					vlane = vl;
				};
				using std::integral_constant;
				issue(integral_constant<int, 0>{});
				issue(integral_constant<int, 1>{}); pack(integral_constant<int, 0>{});
				issue(integral_constant<int, 2>{}); pack(integral_constant<int, 1>{});
				issue(integral_constant<int, 3>{}); pack(integral_constant<int, 2>{});
				issue(integral_constant<int, 4>{}); pack(integral_constant<int, 3>{});
				issue(integral_constant<int, 5>{}); pack(integral_constant<int, 4>{});
				issue(integral_constant<int, 6>{}); pack(integral_constant<int, 5>{});
				issue(integral_constant<int, 7>{}); pack(integral_constant<int, 6>{});
				pack(integral_constant<int, 7>{});
			}
			BS_T(ts3);
			BS_ACC(2, ts2, ts3);
		}
		if (part == 0 && lane == 0 && f1_acc) atomicAdd(a.f1, (unsigned long long)f1_acc);
#ifdef NTC_BS_TIMERS
		if (lane == 0 && a.dbg)
			for (int i = 0; i < 4; ++i)
				atomicAdd((unsigned long long*)a.dbg + 8 + i, (unsigned long long)tacc[i]);
#endif
		return;
	}

	// =========================== WALKER: windows [sb, sb + Q) of every read of every tile ===========================
	const uint32_t Q = 16u * a.nq; // windows per walker segment
	const uint32_t sb = part * Q;  // first window (= first base) of segment `part`
	const uint32_t rmask = (1u << a.r_bits) - 1u, rbuck = 1u << a.r_bits, s_bits = a.s_bits;
	// ---- hit log (see ntc_sketch_hf.hip): this wave's regions are gwave, gwave + log_w, ... ----
	const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + part);
	const uint32_t log_w = gridDim.x * 4u;
	const bool use_log = a.log_regions != 0 && (a.log_mode == nullptr || __builtin_amdgcn_readfirstlane(*a.log_mode) == 0u);
	uint32_t lreg = gwave, lfill = 0;
	if (use_log && lreg < a.log_regions) lfill = __builtin_amdgcn_readfirstlane(a.log_fill[lreg]);
	auto log_emit = [&](bool hit, uint32_t key) {
		const uint64_t m = ballot(hit);
		if (m == 0) return;
		const uint32_t c = (uint32_t)__popcll(m);
		while (lreg < a.log_regions && c > a.log_region_cap - lfill) {
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
		}
	};
	// ---- resolve: 64 (read, window) pairs -> full canonical hash from the packed bases -> ntComp -> log ----
	auto resolve_round = [&](uint32_t mask, uint32_t lw, uint32_t count) {
		// item = (hit plane word of one lane and step, that lane << 8 | window): this round takes the lowest set bit (read
		// 64 bit + lane); the queue loop puts what is left of the word back in line
		const bool act = (uint32_t)lane < count;
		const uint32_t r = act ? (uint32_t)__builtin_ctz(mask | 0x80000000u) * 64u + (lw >> 8) : 0u, win = act ? lw & 0xffu : 0u;
		const uint32_t B = r * stride + win;          // tile byte offset of the window's first base
		const uint32_t sh = (B & 15u) * 2u;           // bit offset inside the packed dword
		const uint32_t* dp = tile + (B >> 4);
		const uint32_t rd = rdirty[r >> 5];
		uint32_t d[KB + 1]; // KB + 1 aligned dwords cover the window's 2 K bits at any shift
#pragma unroll
		for (int i = 0; i < KB + 1; ++i)
			d[i] = dp[i];
		// every table address first, then all K / 4 lookups in flight together, then the XORs (see fetch_planes)
		uint32_t toff[K / 4];
#pragma unroll
		for (int i = 0; i < KB; ++i) {
			const uint32_t w = alignbit(d[i + 1], d[i], sh); // 16 bases of the window
#pragma unroll
			for (int g = 0; g < 4; ++g)
				toff[i * 4 + g] = (uint32_t)(i * 4 + g) * 4096u + ((w >> (8 * g)) & 0xffu) * 16u;
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
		const bool c0 = (hi >> (31 - s_bits)) == 1u;
		const bool clean = ((rd >> (r & 31u)) & 1u) == 0u; // dirty reads are handed to the lane-per-read kernel as a whole
		const bool hit = act & clean & (c0 | c1);
		const uint32_t key = a.key_base + (lo & rmask) + (c1 ? rbuck : 0u);
		if (use_log)
			log_emit(hit, key);
		else if (hit)
			atomicAdd(a.sketch0 + key, 1u);
	};
	// ---- LDS queue of hit-plane items: [qhead, qhead + qfill) mod kQueueCap.  A step hands its hit plane over with ONE
	// ballot-compacted 8-byte store per lane that saw a hit (no per-bit loop, no prefix sum over counts); the bits of a
	// word are taken one per resolve round ----
	uint32_t qhead = 0, qfill = 0; // wave-uniform
	const uint32_t lane8 = (uint32_t)lane << 8;
	auto qslot = [&](uint32_t x) { return x >= kQueueCap ? x - kQueueCap : x; }; // x < 2 kQueueCap
	auto drain_queue = [&](uint32_t keep) { // resolve until at most `keep` items are left (keep < 64: the last round is partial)
		while (qfill > keep) {
			const uint32_t n = qfill < 64u ? qfill : 64u;
			const uint2 it = queue[qslot(qhead + (uint32_t)lane)];
			const bool act = (uint32_t)lane < n;
			const uint32_t rest = act ? it.x & (it.x - 1u) : 0u;
			const uint64_t m = ballot(rest != 0u);
			// words with more bits go to the back of the line first (the slots this round frees are not reused before it ends:
			// the tail never catches up with the head because every round frees n and re-queues at most n)
			if (m != 0) {
				const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				if (rest != 0u) queue[qslot(qslot(qhead + qfill) + pos)] = make_uint2(rest, it.y);
			}
			resolve_round(it.x, it.y, n);
			qhead = (uint32_t)__builtin_amdgcn_readfirstlane((int)qslot(qhead + n));
			qfill = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qfill - n + (uint32_t)__popcll(m)));
		}
	};
	auto push = [&](uint32_t h, uint32_t win, bool valid) { // h: bit i set <=> read 64 i + lane sampled at window `win`; valid: wave-uniform
		// straight-line on purpose (a branch inside the 16-step block costs the register allocator its plan, measured 148
		// spilled VGPRs): lanes without a hit store to their spare slot behind the queue; room is guaranteed by the caller
		h = valid ? h : 0u;
		const uint64_t m = ballot(h != 0u);
		const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
		const uint32_t slot = qslot(qslot(qhead + qfill) + pos);
		queue[h != 0u ? slot : kQueueCap + (uint32_t)lane] = make_uint2(h, lane8 | win); // a spare slot per lane: one shared slot is a 40-way bank conflict
		qfill = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qfill + (uint32_t)__popcll(m)));
	};
	const uint32_t abase = (uint32_t)lane * s4; // packed BYTE offset of read `lane`; reads 64 i + lane follow every 16 * s4 dwords
	auto fetch_planes = [&](uint32_t (&P)[32], uint32_t pos) { // planes of bases [pos, pos + 16) of the lane's 32 reads
		const uint32_t byte0 = abase + (pos >> 2);
		const uint32_t sh = (byte0 & 3u) * 8u + (pos & 3u) * 2u;
		const uint32_t* p = tile + (byte0 >> 2);
		// all loads of a half first, then their uses: left alone, the scheduler (minimising live ranges) waits for every
		// single LDS load, and with one wave per SIMD nothing hides those round trips
#pragma unroll
		for (int hf = 0; hf < 2; ++hf) {
			uint32_t d0[16], d1[16];
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				d0[i] = p[(uint32_t)(hf * 16 + i) * 16u * s4];
				d1[i] = p[(uint32_t)(hf * 16 + i) * 16u * s4 + 1u];
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < 16; ++i)
				P[hf * 16 + i] = alignbit(d1[i], d0[i], sh);
			__builtin_amdgcn_sched_barrier(0);
		}
		transpose32(P); // P[2 q + b] = bit b of the code of base pos + q, one bit per read
	};
	for (uint64_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
		BS_T(tw0);
		__syncthreads(); // B1
		__syncthreads(); // B2
		__syncthreads(); // B3: the tile image and its read-dirty bits are final
		BS_T(tw1);
		BS_ACC(0, tw0, tw1);
		uint32_t F[31], R[31];
#pragma unroll
		for (int j = 0; j < 31; ++j)
			F[j] = R[j] = 0;
		uint32_t H[KB][32]; // bit planes of the last KB chunks (H[0]: the bases that leave the window in the next block)
		auto test = [&]() -> uint32_t {
			if constexpr (SB == 7) return bs_test_s7(F, R);
			else return bs_test_s8(F, R);
		};
#pragma unroll
		for (int b = 0; b < KB; ++b) { // window filling: no outgoing base; the last step completes window sb
			BS_T(tf0);
			fetch_planes(H[b], sb + 16u * (uint32_t)b);
			BS_T(tf1);
			BS_ACC(4, tf0, tf1);
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				if constexpr (K == 32) bs_step_warm_k32(F, R, H[b][2 * q], H[b][2 * q + 1]);
			}
#ifdef NTC_BS_TIMERS
			pin31(F);
			pin31(R);
#endif
			BS_T(tf2);
			BS_ACC(5, tf1, tf2);
		}
		{ // window sb itself
			const uint32_t h0 = test();
			push(h0, sb, Q != 0u && sb < W);
		}
		BS_T(tw2);
		BS_ACC(1, tw1, tw2);
#pragma unroll 1
		for (uint32_t bq = 0; bq < a.nq; ++bq) { // steady state: 16 windows per block
			BS_T(tw3);
			uint32_t I[32];
			fetch_planes(I, sb + 16u * ((uint32_t)KB + bq));
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				if constexpr (K == 32) bs_step_main_k32(F, R, I[2 * q], I[2 * q + 1], H[0][2 * q], H[0][2 * q + 1]);
				{ // local window 16 bq + q + 1 (window Q belongs to the next walker); the condition is wave-uniform
					const uint32_t hq = test(), lw = 16u * bq + (uint32_t)q + 1u;
					push(hq, sb + lw, lw < Q && sb + lw < W);
				}
			}
			// the chunk that just came in leaves the window KB blocks from now
#pragma unroll
			for (int hh = 0; hh + 1 < KB; ++hh)
#pragma unroll
				for (int i = 0; i < 32; ++i)
					H[hh][i] = H[hh + 1][i];
#pragma unroll
			for (int i = 0; i < 32; ++i)
				H[KB - 1][i] = I[i];
			pin31(F);
			pin31(R);
			BS_T(tw4);
			BS_ACC(2, tw3, tw4);
			// at most 63 items stay queued: the next block adds at most 16 x 64
			drain_queue(63u);
			BS_T(tw5);
			BS_ACC(3, tw4, tw5);
		}
		drain_queue(0u); // the image changes at the tile switch: nothing of this tile may stay queued
	}
	if (use_log && lane == 0 && lreg < a.log_regions) a.log_fill[lreg] = lfill;
#ifdef NTC_BS_TIMERS
	if (lane == 0 && a.dbg)
		for (int i = 0; i < 6; ++i)
			atomicAdd((unsigned long long*)a.dbg + i, (unsigned long long)tacc[i]);
#endif
}

namespace {
template <int K, int SB>
hipError_t launch_one(const BsArgs& a, unsigned grid, size_t smem, hipStream_t st)
{
	hipLaunchKernelGGL((sketch_bs_kernel<K, SB>), dim3(grid), dim3(512), smem, st, a);
	return hipGetLastError();
}
} // namespace

bool sketch_bs_supports(uint32_t k, uint32_t s_bits) { return k == 32 && s_bits >= 7; }

size_t sketch_bs_smem(uint32_t k, uint32_t stride)
{
	const size_t tile_dw = 128u * (size_t)stride;
	return (tile_dw + 64) * 4 + (size_t)(k / 4) * 4096 + 2 * (tile_dw / 32) * 4 + 64 * 4 + 4 * (size_t)(kQueueCap + 64) * 8;
}

hipError_t launch_sketch_bs(const BsArgs& a, unsigned grid, hipStream_t st)
{
	const size_t smem = sketch_bs_smem(a.k, a.stride);
	if (a.k == 32 && a.s_bits == 7) return launch_one<32, 7>(a, grid, smem, st);
	if (a.k == 32 && a.s_bits >= 8) return launch_one<32, 8>(a, grid, smem, st);
	return hipErrorInvalidValue;
}

__global__ __launch_bounds__(256) void append_slots_kernel(uint64_t* list, uint32_t* count, const unsigned char* slots, uint32_t stride, uint64_t first, uint32_t n)
{
	const uint32_t base = *count; // one workgroup, stream-ordered behind the kernel that filled the list so far
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
		list[base + i] = (uint64_t)(uintptr_t)slots + (first + i) * stride;
	if (threadIdx.x == 0) *count = base + n;
}

hipError_t launch_append_slots(uint64_t* list, uint32_t* count, const unsigned char* slots, uint32_t stride, uint64_t first, uint32_t n, hipStream_t st)
{
	hipLaunchKernelGGL(append_slots_kernel, dim3(1), dim3(256), 0, st, list, count, slots, stride, first, n);
	return hipGetLastError();
}

hipError_t set_sketch_bs_smem_limit(size_t smem)
{
	const void* fns[] = { reinterpret_cast<const void*>(&sketch_bs_kernel<32, 7>), reinterpret_cast<const void*>(&sketch_bs_kernel<32, 8>) };
	for (const void* f : fns) {
		const hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (rc != hipSuccess) return rc;
	}
	return hipSuccess;
}

} // namespace ntc
