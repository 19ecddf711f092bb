// ntc_sketch_fast.hip — K1, the production ntHash -> sample -> count kernel for gfx950 (CDNA4).
//
// Same contract as nthash_kernel<0> in ntc_kernels.hip (which stays as the simple, independently
// written variant used for validation), restructured around what the MI355X micro-benchmarks say
// (tools/ubench*.hip): only xor/and/or/add/sub/lshr/mov issue at the full
// VALU rate, everything else (alignbit, perm, bfe, min, cmp, 3-operand ops, SGPR operands) at half
// rate, and device-scope atomics top out near 20 G/s.  Hence:
//   * bytes are decoded ONCE, while the wave stages its slots into LDS (ASCII -> code<<6 per byte,
//     bit 0 marks a byte that is not ACGTU);
//   * the steady-state step is table lookup (ds_read_b128 + ds_read_b32) + 12 VALU ops of rolling
//     (ntc::roll) + min + 2 compares; dirty-window / read-end handling lives in a separate code
//     path that a wave only enters when one of its lanes needs it;
//   * a sampled hash is NOT resolved inside the hot loop: the lane drops its raw strand registers
//     into a per-lane queue (one 16-byte buffer store), and resolves canonical strand, sample plane
//     and bucket index for all of its hits after the read, where lanes are dense.
//
// Reference semantics reproduced: ntRead (ntcard.cpp:147-158), ntHashIterator (ntHashIterator.hpp:
// 59-86), NTMC64/NTF64/NTR64 (nthash.hpp:242-257,381-390,467-492), ntComp (ntcard.cpp:132-145).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ntc_kernels.hpp"

#ifndef NTC_PIPE
#define NTC_PIPE 0 // 1: table entries of group g+1 are fetched while group g is hashed
#endif
#ifndef NTC_PREF
#define NTC_PREF 1 // 1: global loads of batch i+1 stay in flight (registers) while batch i is hashed
#endif

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

// v_perm tables indexed by (byte & 7): 1:A 3:C 7:G 4:T 5:U, 0/2/6: not a base (nthash.hpp:16,32 trick)
constexpr uint32_t kExpS0 = 0x47ff5554u; // 'G', ff, 'U', 'T'
constexpr uint32_t kExpS1 = 0x43ff41ffu; // 'C', ff, 'A', ff
constexpr uint32_t kIn6S0 = 0x8000c0c0u; // code<<6 : G=2, -, U=3, T=3
constexpr uint32_t kIn6S1 = 0x40000000u; //           C=1, -, A=0, -

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// x << 1 as a full-rate v_add_u32 (hipcc canonicalises x + x back into the half-rate v_lshlrev_b32)
__device__ __forceinline__ uint32_t dbl(uint32_t x)
{
	uint32_t r;
	asm("v_add_u32_e32 %0, %1, %1" : "=v"(r) : "v"(x));
	return r;
}

struct Strands {
	uint32_t flo, fB, fHd; // forward:  L[0..31], L[32] in bit 31, (H<<1)|H[30]
	uint32_t rlo, rB, rHd; // reverse:  L[0..31], L[32] in bit 0,  (H<<1)|H[30]
};

// One rolling step (NTF64 + NTR64, nthash.hpp:242-257); see nthash_tables.hpp for the layout.
// x + x instead of x << 1: v_add_u32 issues at the full VALU rate, v_lshlrev_b32 at half.
__device__ __forceinline__ void roll(Strands& s, const uint4 t, const uint32_t tbb)
{
	const uint32_t nflo = alignbit(s.flo, s.fB, 31) ^ t.x;
	s.fB = s.flo ^ tbb;
	s.flo = nflo;
	s.fHd = alignbit(s.fHd, dbl(s.fHd), 31) ^ t.y;
	const uint32_t xlo = s.rlo ^ t.z;
	const uint32_t xb = s.rB ^ tbb;
	s.rlo = alignbit(xb, xlo, 1);
	s.rB = xlo;
	const uint32_t xh = s.rHd ^ t.w;
	s.rHd = alignbit(xh >> 1, xh, 1);
}

__device__ __forceinline__ bool rev_smaller(const Strands& s)
{
	if (s.rHd != s.fHd) return s.rHd < s.fHd;
	const uint32_t fb = s.fB >> 31, rb = s.rB & 1u;
	if (fb != rb) return rb < fb;
	return s.rlo < s.flo;
}

// decode one dword of raw bytes -> code<<6 per byte, bit 0 set on bytes that are not ACGTU/acgtu
__device__ __forceinline__ uint32_t decode4(uint32_t w, uint32_t& badacc)
{
	const uint32_t sel = w & 0x07070707u;
	const uint32_t bad = (perm(kExpS0, kExpS1, sel) ^ w) & 0xdfdfdfdfu;
	uint32_t code = perm(kIn6S0, kIn6S1, sel);
	badacc |= bad;
	if (bad != 0u) {
		const uint32_t nz = (((bad & 0x7f7f7f7fu) + 0x7f7f7f7fu) | bad) & 0x80808080u;
		code = (code & ~(nz >> 1) & ~(nz >> 0)) | (nz >> 7); // dirty byte: code 0, mark bit 0
	}
	return code;
}

} // namespace

__global__ __launch_bounds__(kBlockThreads, 4) void sketch_fast_kernel(const HashArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[]; // per-wave slot data
	__shared__ __align__(16) uint32_t tabw[kSlots * 8];     // static: offsets fold into ds_read immediates
	const unsigned char* const tabA = reinterpret_cast<const unsigned char*>(tabw);
	const unsigned char* const tabB = tabA + kSlots * 16;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	{
		const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.tab);
		for (int i = tid; i < kSlots * 8; i += kBlockThreads)
			tabw[i] = src[i];
	}
	__syncthreads();

	const uint32_t stride = a.stride;
	const uint32_t k = a.k;
	unsigned char* const wdata = smem + 16 + (size_t)wave * 64u * stride; // 16 B pad: out-base loads reach 4 B below a slot
	const unsigned char* const mine = wdata + (size_t)lane * stride;

	// sample windows on the top bits (ntcard.cpp:135-138).  Kept in VGPRs on purpose: a VALU op with an
	// SGPR source issues at half rate on gfx950 (tools/ubench2).
	uint32_t lo0 = 1u << (31 - a.s_bits);
	int32_t lo1 = (int32_t)(((1u << (a.s_bits - 1)) - 1u) << (32 - a.s_bits));
	asm volatile("" : "+v"(lo0), "+v"(lo1));
	const uint32_t rmask = (1u << a.r_bits) - 1u;
	const uint32_t rbuck = 1u << a.r_bits;

	// per-wave hit queue: row j holds the j-th hit of every lane (64 x 16 B), rows are 1 KiB apart
	// readfirstlane: the buffer descriptor must be provably wave-uniform or hipcc wraps every
	// buffer op in a waterfall loop (cdna_hip_programming.md T20)
	const uint32_t gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wave);
	const uint64_t qbytes = (uint64_t)a.queue_rows * 1024u;
	__amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(
	    reinterpret_cast<unsigned char*>(a.queue) + (uint64_t)gwave * qbytes, 0, (int)qbytes, 0x00020000);

	const uint64_t n_wb = (a.n_slots + 63) / 64;
	uint64_t f1_wave = 0;

	const uint32_t shb = (0u - k) & 3u; // byte phase of the outgoing-base stream

	// ---- global -> LDS staging.  A full wave batch (64 slots) is 64*stride contiguous bytes =
	// `nchunk` coalesced 16-byte loads per lane.  Up to kPref of them are kept in flight in
	// registers: the loads of batch i+1 are issued before batch i is hashed, so HBM latency is
	// paid under the hash loop instead of in front of it.
	constexpr int kPref = NTC_PREF ? 10 : 4; // 10 KiB per wave: slots of up to 160 bytes
	const uint32_t full_bytes = 64u * stride;
	const uint32_t nchunk = (full_bytes + 1023u) >> 10;
	const bool can_prefetch = NTC_PREF && nchunk <= (uint32_t)kPref;
	const uint64_t wb_step = (uint64_t)gridDim.x * kWavesPerBlock;
	uint4 pref[kPref];
	auto load_round = [&](uint64_t wb_, uint32_t c0) {
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
			const uint32_t off = lane * 16u + (c0 + c) * 1024u;
			if (off + 16u <= full_bytes) {
				uint4 v = pref[c];
				v.x = decode4(v.x, badacc);
				v.y = decode4(v.y, badacc);
				v.z = decode4(v.z, badacc);
				v.w = decode4(v.w, badacc);
				*reinterpret_cast<uint4*>(wdata + off) = v;
			}
		}
	};
	auto is_full = [&](uint64_t wb_) { return wb_ * 64 + 64 <= a.n_slots; };
	if (can_prefetch && gwave < n_wb && is_full(gwave)) load_round(gwave, 0);

	for (uint64_t wb = gwave; wb < n_wb; wb += wb_step) {
		const uint64_t slot0 = wb * 64;
		const uint32_t nvalid = (uint32_t)((a.n_slots - slot0) < 64 ? (a.n_slots - slot0) : 64);
		// ---- stage + decode: coalesced 16 B global loads -> code bytes in LDS ----
		uint32_t badacc = 0;
		__builtin_amdgcn_wave_barrier();
		if (nvalid == 64 && can_prefetch) {
			store_round(0, badacc);
			if (wb + wb_step < n_wb && is_full(wb + wb_step)) load_round(wb + wb_step, 0);
		} else if (nvalid == 64) {
			for (uint32_t c0 = 0; c0 < nchunk; c0 += kPref) {
				load_round(wb, c0);
				store_round(c0, badacc);
			}
		} else {
			// last, partial batch of the launch
			const unsigned char* src = a.slots + slot0 * stride;
			const uint32_t bytes = nvalid * stride; // multiple of 4
			for (uint32_t off = lane * 16u; off < bytes; off += 1024u) {
				for (uint32_t o = off; o < bytes && o < off + 16u; o += 4)
					*reinterpret_cast<uint32_t*>(wdata + o) =
					    decode4(*reinterpret_cast<const uint32_t*>(src + o), badacc);
			}
		}
		__builtin_amdgcn_wave_barrier();
		const bool wave_dirty = __any(badacc != 0u);

		// ---- per-lane read geometry ----
		uint32_t len = a.read_len, wlim = a.read_len;
		const bool in_batch = (uint32_t)lane < nvalid;
		if (a.meta != nullptr && in_batch) {
			const uint32_t m = a.meta[slot0 + lane];
			len = m & 0xffffu;
			wlim = m >> 16;
		}
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
		// a "uniform" wave: every lane walks the same number of steps and no lane saw a dirty byte
		const bool uniform = (minq == maxq) && !wave_dirty;

		// The walk starts from the hash of a virtual window of k 'A's and feeds 'A' (code 0) as the
		// outgoing base of the first k steps: the rolling identity then removes the virtual bases
		// again, so ONE step body serves window filling and steady state (no special table slots).
		Strands s = { a.init[0], a.init[1], a.init[2], a.init[3], a.init[4], a.init[5] };
		int32_t nextok = endq > 0 ? (int32_t)k - 1 : 0x7fffffff; // emission allowed from this step on
		uint32_t qoff = lane * 16u;                              // byte offset of my next queue record

		auto lookup = [&](uint32_t off, uint4& t, uint32_t& tbb) {
			t = *reinterpret_cast<const uint4*>(tabA + off);
			tbb = *reinterpret_cast<const uint32_t*>(tabB + off);
		};
		auto sampled = [&]() -> bool {
			const uint32_t m = s.fHd < s.rHd ? s.fHd : s.rHd; // top bits of min(fh,rh)
			return ((m ^ lo0) < lo0) | ((int32_t)m >= lo1);
		};
		auto enqueue = [&]() {
			const v4u rec = { s.flo, s.rlo, s.fHd, s.rHd };
			__builtin_amdgcn_raw_buffer_store_b128(rec, qrsrc, qoff, 0, 0);
			qoff += 1024u;
		};
		// table offsets (one byte per base) of the 4 steps of group q0
		auto group_idx = [&](int32_t q0, uint32_t& ain) -> uint32_t {
			ain = *reinterpret_cast<const uint32_t*>(mine + q0);
			uint32_t aout = 0;
			if (q0 + 3 >= (int32_t)k) { // at least one step of the group has a real outgoing base
				const uint32_t* p = reinterpret_cast<const uint32_t*>(mine + ((q0 - (int32_t)k) & ~3));
				aout = shb ? alignbyte(p[1], p[0], shb) : p[0];
				if (q0 < (int32_t)k) aout &= 0xffffffffu << (8 * ((int32_t)k - q0)); // steps q < k: virtual 'A'
			}
			return (ain & 0xc0c0c0c0u) | ((aout >> 2) & 0x30303030u);
		};
		struct Tab4 {
			uint4 t[4];
			uint32_t tb[4];
		};
		auto issue = [&](uint32_t idx4, Tab4& T) {
#pragma unroll
			for (int b = 0; b < 4; ++b)
				lookup((idx4 >> (8 * b)) & 0xffu, T.t[b], T.tb[b]);
		};

		// group kinds by position: FILL (all 4 steps before k-1: no window yet), MIXED (the group that
		// contains step k-1, or a partial last group), MAIN (every step emits)
		constexpr int FILL = 0, MIXED = 1, MAIN = 2;
		const int32_t full_groups = maxq >> 2;                       // groups with 4 steps for the longest lane
		const int32_t first_main = ((int32_t)k - 1 + 3) >> 2;        // first group with q0 >= k-1
		const int32_t fill_end = ((int32_t)k - 1) >> 2;              // groups [0, fill_end) are pure FILL

		if (uniform) {
			// ---- clean wave: every lane walks the same steps, no per-lane bookkeeping at all ----
			const uint32_t nact = __popcll(__ballot(true));
			auto compute = [&](auto kind, int32_t q0, const Tab4& T) {
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					roll(s, T.t[b], T.tb[b]);
					if (kind.value == MAIN || (kind.value == MIXED && q0 + b >= (int32_t)k - 1)) {
						if (sampled()) enqueue();
					}
				}
			};
			auto run = [&](auto kind, int32_t g0, int32_t g1) {
				uint32_t ain;
#if NTC_PIPE
				Tab4 A, B;
				if (g0 < g1) issue(group_idx(g0 << 2, ain), A);
				for (int32_t g = g0; g < g1; g += 2) {
					if (g + 1 < g1) issue(group_idx((g + 1) << 2, ain), B);
					compute(kind, g << 2, A);
					if (g + 1 < g1) {
						if (g + 2 < g1) issue(group_idx((g + 2) << 2, ain), A);
						compute(kind, (g + 1) << 2, B);
					}
				}
#else
				for (int32_t g = g0; g < g1; ++g) {
					Tab4 A;
					issue(group_idx(g << 2, ain), A);
					compute(kind, g << 2, A);
				}
#endif
			};
			const int32_t e0 = fill_end < full_groups ? fill_end : full_groups;
			const int32_t e1 = first_main < full_groups ? first_main : full_groups;
			run(std::integral_constant<int, FILL>{}, 0, e0);
			run(std::integral_constant<int, MIXED>{}, e0, e1 > e0 ? e1 : e0);
			run(std::integral_constant<int, MAIN>{}, e1 > e0 ? e1 : e0, full_groups);
			// partial last group: one step at a time
			for (int32_t q = full_groups << 2; q < maxq; ++q) {
				const uint32_t ain = mine[q];
				const uint32_t off = (ain & 0xc0u) | (q >= (int32_t)k ? ((mine[q - (int32_t)k] >> 2) & 0x30u) : 0u);
				uint4 t;
				uint32_t tbb;
				lookup(off, t, tbb);
				roll(s, t, tbb);
				if (q >= (int32_t)k - 1 && sampled()) enqueue();
			}
			if (maxq >= (int32_t)k) f1_wave += (uint64_t)nact * (uint32_t)(maxq - (int32_t)k + 1);
		} else {
			// ---- dirty / ragged wave: per-lane `nextok` (first step whose window is clean again) ----
			auto step_tail = [&](int32_t q, uint32_t mark) {
				if (mark && nextok != 0x7fffffff) nextok = q + (int32_t)k;
				if (q >= endq) nextok = 0x7fffffff;
			};
			auto emit = [&](int32_t q) {
				const bool live = nextok <= q;
				f1_wave += __popcll(__ballot(live));
				if (live && sampled()) enqueue();
			};
			auto compute = [&](auto kind, int32_t q0, uint32_t ain, const Tab4& T) {
				// lanes that are shut off (outside the batch / finished) never trigger the extra work
				const bool alive = nextok != 0x7fffffff;
				const bool fix = __any((((ain & 0x01010101u) != 0u) | (endq < q0 + 4)) & alive);
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const int32_t q = q0 + b;
					if (fix) step_tail(q, (ain >> (8 * b)) & 1u);
					roll(s, T.t[b], T.tb[b]);
					if (kind.value == MAIN || (kind.value == MIXED && q >= (int32_t)k - 1)) emit(q);
				}
			};
			auto run = [&](auto kind, int32_t g0, int32_t g1) {
#if NTC_PIPE
				Tab4 A, B;
				uint32_t ainA = 0, ainB = 0;
				if (g0 < g1) issue(group_idx(g0 << 2, ainA), A);
				for (int32_t g = g0; g < g1; g += 2) {
					if (g + 1 < g1) issue(group_idx((g + 1) << 2, ainB), B);
					compute(kind, g << 2, ainA, A);
					if (g + 1 < g1) {
						if (g + 2 < g1) issue(group_idx((g + 2) << 2, ainA), A);
						compute(kind, (g + 1) << 2, ainB, B);
					}
				}
#else
				for (int32_t g = g0; g < g1; ++g) {
					Tab4 A;
					uint32_t ainA;
					issue(group_idx(g << 2, ainA), A);
					compute(kind, g << 2, ainA, A);
				}
#endif
			};
			const int32_t e0 = fill_end < full_groups ? fill_end : full_groups;
			const int32_t e1 = first_main < full_groups ? first_main : full_groups;
			run(std::integral_constant<int, FILL>{}, 0, e0);
			run(std::integral_constant<int, MIXED>{}, e0, e1 > e0 ? e1 : e0);
			run(std::integral_constant<int, MAIN>{}, e1 > e0 ? e1 : e0, full_groups);
			for (int32_t q = full_groups << 2; q < maxq; ++q) {
				const uint32_t ain = mine[q];
				step_tail(q, ain & 1u);
				const uint32_t off = (ain & 0xc0u) | (q >= (int32_t)k ? ((mine[q - (int32_t)k] >> 2) & 0x30u) : 0u);
				uint4 t;
				uint32_t tbb;
				lookup(off, t, tbb);
				roll(s, t, tbb);
				if (q >= (int32_t)k - 1) emit(q);
			}
		}

		// ---- drain: resolve my queued hits (canonical strand, sample plane, bucket) ----
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const uint32_t nrec = qoff >> 10;
		auto resolve = [&](const v4u rec) {
			{
				const uint32_t flo = rec.x, rlo = rec.y, fHd = rec.z, rHd = rec.w;
				bool rev = rHd < fHd;
				if (rHd == fHd && rlo != flo) {
					// top 31 bits tie (p = 2^-31): bit 32 decides first; re-walk the read to recover it
					Strands w = { a.init[0], a.init[1], a.init[2], a.init[3], a.init[4], a.init[5] };
					for (int32_t q = 0; q < endq; ++q) {
						const uint32_t ain = mine[q];
						const uint32_t off = (ain & 0xc0u) | (q >= (int32_t)k ? ((mine[q - (int32_t)k] >> 2) & 0x30u) : 0u);
						uint4 t;
						uint32_t tbb;
						lookup(off, t, tbb);
						roll(w, t, tbb);
						if (w.flo == flo && w.rlo == rlo && w.fHd == fHd && w.rHd == rHd) {
							rev = rev_smaller(w);
							break;
						}
					}
				}
				const uint32_t m = rev ? rHd : fHd;
				const uint32_t lo = rev ? rlo : flo;
				const uint32_t idx = (lo & rmask) + ((int32_t)m >= lo1 ? rbuck : 0u);
				atomicAdd(a.sketch + idx, 1u);
			}
		};
		// four queue rows per round: the loads go out together, so their L2 latency is paid once
		for (uint32_t j0 = 0; __any(j0 < nrec); j0 += 4) {
			v4u rec[4];
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u)
				if (j0 + u < nrec)
					rec[u] = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane * 16u + (j0 + u) * 1024u, 0, 1 /*glc: bypass L1*/);
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u)
				if (j0 + u < nrec) resolve(rec[u]);
		}
	}
	if (lane == 0 && f1_wave) atomicAdd(a.f1, (unsigned long long)f1_wave);
}

hipError_t launch_sketch_fast(const HashArgs& a, unsigned grid, size_t smem, hipStream_t st)
{
	hipLaunchKernelGGL(sketch_fast_kernel, dim3(grid), dim3(kBlockThreads), smem, st, a);
	return hipGetLastError();
}

hipError_t set_sketch_fast_smem_limit(size_t smem)
{
	return hipFuncSetAttribute(reinterpret_cast<const void*>(&sketch_fast_kernel),
	                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

} // namespace ntc
