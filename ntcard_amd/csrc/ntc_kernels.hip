// ntc_kernels.hip — gfx950 (CDNA4) kernels of the ntHash -> sample -> count hot path.
//
// Replaces, on the device, the reference's per-sequence loop
//   ntRead / stRead  (ntcard.cpp:147-171)  ->  ntHashIterator (ntHashIterator.hpp:59-86)
//   -> NTMC64 base/roll (nthash.hpp:467-492,381-390; NTF64 :242-248, NTR64 :251-257, min :275-279)
//   -> ntComp (ntcard.cpp:132-145)
// Work decomposition: one LANE walks one read slot with the rolling
// recurrence; a wave stages its 64 slots (one contiguous region) through LDS with coalesced
// 16-byte loads; the per-(in,out)-base seed terms come from a 20-entry table in LDS
// (bank-conflict-free ds_read_b128 + ds_read_b32, see nthash_tables.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntc_kernels.hpp"

namespace ntc {

// ---- small device helpers ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
	return __builtin_amdgcn_alignbit(hi, lo, sh); // ({hi,lo} >> sh)[31:0]
}
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
	return __builtin_amdgcn_alignbyte(hi, lo, sh); // ({hi,lo} >> 8*sh)[31:0]
}
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
	return __builtin_amdgcn_perm(s0, s1, sel); // byte i = {s0,s1}.byte[sel.byte[i]], 0-3 -> s1
}

// v_perm tables indexed by (byte & 7): 1:A 3:C 7:G 4:T 5:U, 0/2/6: not a base.
// (the same low-3-bit trick the reference uses for complements, nthash.hpp:16,32)
//                          idx:   7     6     5     4            3     2     1     0
constexpr uint32_t kExpS0 = 0x47ff5554u; // 'G', ff, 'U', 'T'
constexpr uint32_t kExpS1 = 0x43ff41ffu; // 'C', ff, 'A', ff
constexpr uint32_t kIn6S0 = 0x8000c0c0u; // code<<6 : G=2, -, U=3, T=3
constexpr uint32_t kIn6S1 = 0x40000000u; //           C=1, -, A=0, -
constexpr uint32_t kIn4S0 = 0x20003030u; // code<<4
constexpr uint32_t kIn4S1 = 0x10000000u;

struct Strands {
	uint32_t flo, fB, fHd; // forward:  L[0..31], L[32] in bit 31, (H<<1)|H[30]
	uint32_t rlo, rB, rHd; // reverse:  L[0..31], L[32] in bit 0,  (H<<1)|H[30]
};

// One rolling step (NTF64 + NTR64, nthash.hpp:242-257) with the seed terms of this (in,out) pair.
__device__ __forceinline__ void roll(Strands& s, const uint4 t, const uint32_t tbb)
{
	// forward: fh' = srol1(fh) ^ Tf
	const uint32_t nflo = alignbit(s.flo, s.fB, 31) ^ t.x;
	s.fB = s.flo ^ tbb;
	s.flo = nflo;
	s.fHd = alignbit(s.fHd, s.fHd << 1, 31) ^ t.y;
	// reverse: rh' = srol^-1(rh ^ Tr)
	const uint32_t xlo = s.rlo ^ t.z;
	const uint32_t xb = s.rB ^ tbb;
	s.rlo = alignbit(xb, xlo, 1);
	s.rB = xlo;
	const uint32_t xh = s.rHd ^ t.w;
	s.rHd = alignbit(xh >> 1, xh, 1);
}

// canonical choice (nthash.hpp:275-279): true when the reverse strand is strictly smaller
__device__ __forceinline__ bool rev_smaller(const Strands& s)
{
	if (s.rHd != s.fHd) return s.rHd < s.fHd;
	const uint32_t fb = s.fB >> 31, rb = s.rB & 1u;
	if (fb != rb) return rb < fb;
	return s.rlo < s.flo;
}

__device__ __forceinline__ uint64_t assemble(uint32_t lo, uint32_t b32, uint32_t hd)
{
	return (uint64_t)lo | ((uint64_t)b32 << 32) | ((uint64_t)(hd >> 1) << 33);
}

template <int MODE>
struct Emit;

// ------------------------------------------------------------------------------------------------
// K1 / K1d: MODE 0 = sketch update, MODE 1 = hash dump (validation)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlockThreads) void nthash_kernel(const HashArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	uint32_t* const tabw = reinterpret_cast<uint32_t*>(smem);
	const unsigned char* const tabA = smem;                    // kSlots x 16 B
	const unsigned char* const tabB = smem + kSlots * 16;      // kSlots x 16 B (dword 0 used)
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
	unsigned char* const wdata = smem + kTableBytes + (size_t)wave * 64u * stride;
	const unsigned char* const mine = wdata + (size_t)lane * stride;

	const uint32_t lo0 = 1u << (31 - a.s_bits);                          // sample 0: m in [lo0, 2*lo0)
	const int32_t lo1 = (int32_t)(((1u << (a.s_bits - 1)) - 1u) << (32 - a.s_bits)); // sample 1: m in [lo1, 2^31)
	const uint32_t rmask = (1u << a.r_bits) - 1u;
	const uint32_t rbuck = 1u << a.r_bits;

	const uint64_t n_wb = (a.n_slots + 63) / 64;
	uint64_t f1_wave = 0;

	for (uint64_t wb = (uint64_t)blockIdx.x * kWavesPerBlock + wave; wb < n_wb;
	     wb += (uint64_t)gridDim.x * kWavesPerBlock) {
		const uint64_t slot0 = wb * 64;
		const uint32_t nvalid = (uint32_t)((a.n_slots - slot0) < 64 ? (a.n_slots - slot0) : 64);
		// ---- stage the wave's contiguous slot region into LDS (coalesced 16 B per lane) ----
		{
			const unsigned char* src = a.slots + slot0 * stride;
			const uint32_t bytes = nvalid * stride; // multiple of 4
			__builtin_amdgcn_wave_barrier();
			for (uint32_t off = lane * 16u; off < bytes; off += 1024u) {
				if (off + 16u <= bytes) {
					const uint4 v = *reinterpret_cast<const uint4*>(src + off);
					*reinterpret_cast<uint4*>(wdata + off) = v;
				} else {
					for (uint32_t o = off; o < bytes; o += 4)
						*reinterpret_cast<uint32_t*>(wdata + o) =
						    *reinterpret_cast<const uint32_t*>(src + o);
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
		// ---- per-lane read geometry ----
		uint32_t len = a.read_len, wlim = a.read_len;
		if (a.meta != nullptr && (uint32_t)lane < nvalid) {
			const uint32_t m = a.meta[slot0 + lane];
			len = m & 0xffffu;
			wlim = m >> 16;
		}
		int32_t endq = (int32_t)(len < wlim + k - 1 ? len : wlim + k - 1); // steps q in [0,endq)
		if ((uint32_t)lane >= nvalid || len < k) endq = 0;
		int32_t maxq = endq;
		for (int o = 32; o > 0; o >>= 1) {
			const int32_t other = __shfl_xor(maxq, o);
			maxq = other > maxq ? other : maxq;
		}
		maxq = __builtin_amdgcn_readfirstlane(maxq);

		Strands s = { 0, 0, 0, 0, 0, 0 };
		// emission allowed from step `nextok` on: k-1 initially, (bad position)+k after a bad byte
		int32_t nextok = endq > 0 ? (int32_t)k - 1 : 0x7fffffff;
		uint32_t nemit = 0; // dump mode: windows written so far by this lane
		const uint64_t myslot = slot0 + lane;

		auto do_emit = [&](int32_t q, bool live) {
			// sample test on the canonical hash's top bits: top bits of min(fh,rh) == min of top bits
			const uint32_t m = s.fHd < s.rHd ? s.fHd : s.rHd;
			if (MODE == 0) {
				const bool c0 = (m ^ lo0) < lo0;
				const bool c1 = (int32_t)m >= lo1;
				if (live && (c0 | c1)) {
					const uint32_t lo = rev_smaller(s) ? s.rlo : s.flo;
					const uint32_t idx = (lo & rmask) + (c1 ? rbuck : 0u);
					atomicAdd(a.sketch + idx, 1u);
				}
			} else {
				if (live) {
					const bool rv = rev_smaller(s);
					const uint64_t h = rv ? assemble(s.rlo, s.rB & 1u, s.rHd)
					                      : assemble(s.flo, s.fB >> 31, s.fHd);
					if (nemit < a.max_win) a.dump[myslot * a.max_win + nemit] = h;
					++nemit;
				}
			}
			(void)q;
		};

		const uint32_t n_groups = (uint32_t)(maxq + 3) >> 2;
		const uint32_t gA = (k - 1) >> 2; // group that contains step k-1
		const uint32_t shb = (0u - k) & 3u; // byte phase of the out stream
		for (uint32_t g = 0; g < n_groups; ++g) {
			const int32_t q0 = (int32_t)(g << 2);
			const uint32_t w_in = *reinterpret_cast<const uint32_t*>(mine + q0);
			const uint32_t sel = w_in & 0x07070707u;
			const uint32_t bad = (perm(kExpS0, kExpS1, sel) ^ w_in) & 0xdfdfdfdfu; // byte != 0: not ACGTU
			// lanes that are already shut off (inactive / finished) never force the generic path
			const bool slow = __any(((bad != 0u) | (endq < q0 + 4)) & (nextok != 0x7fffffff));
			if (g > gA && !slow) {
				// ---- steady state: 4 bases, all with an outgoing base, no dirty byte, no read end ----
				uint32_t w_out;
				{
					const int32_t qo = q0 - (int32_t)k; // >= 0 here
					const uint32_t* p = reinterpret_cast<const uint32_t*>(mine + (qo & ~3));
					w_out = shb ? alignbyte(p[1], p[0], shb) : p[0];
				}
				const uint32_t idx4 = perm(kIn6S0, kIn6S1, sel) |
				                      perm(kIn4S0, kIn4S1, w_out & 0x07070707u);
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const uint32_t off = (idx4 >> (8 * b)) & 0xffu;
					const uint4 t = *reinterpret_cast<const uint4*>(tabA + off);
					const uint32_t tbb = *reinterpret_cast<const uint32_t*>(tabB + off);
					roll(s, t, tbb);
					const int32_t q = q0 + b;
					const bool live = nextok <= q;
					f1_wave += __popcll(__ballot(live));
					do_emit(q, live);
				}
			} else {
				// ---- generic path: window filling, dirty bytes, read ends (per base, any mix) ----
				const uint32_t in4 = perm(kIn4S0, kIn4S1, sel);
#pragma unroll 1
				for (int b = 0; b < 4; ++b) {
					const int32_t q = q0 + b;
					if (q >= maxq) break;
					const uint32_t cin = (in4 >> (8 * b)) & 0xffu; // code<<4
					if (((bad >> (8 * b)) & 0xffu) != 0u) nextok = nextok == 0x7fffffff ? nextok : q + (int32_t)k;
					if (q >= endq) nextok = 0x7fffffff;
					uint32_t off;
					if (q >= (int32_t)k) {
						const uint32_t co = mine[q - (int32_t)k] & 7u;
						const uint32_t cout = (perm(kIn4S0, kIn4S1, co)) & 0xffu; // code<<4
						off = (cin << 2) | cout;
					} else {
						off = kMainSlots * 16 + cin;
					}
					const uint4 t = *reinterpret_cast<const uint4*>(tabA + off);
					const uint32_t tbb = *reinterpret_cast<const uint32_t*>(tabB + off);
					roll(s, t, tbb);
					const bool live = nextok <= q;
					f1_wave += __popcll(__ballot(live));
					do_emit(q, live);
				}
			}
		}
		if (MODE == 1 && (uint32_t)lane < nvalid) a.dump_count[myslot] = nemit;
	}
	if (MODE == 0 && lane == 0 && f1_wave) atomicAdd(a.f1, (unsigned long long)f1_wave);
}

template __global__ void nthash_kernel<0>(const HashArgs);
template __global__ void nthash_kernel<1>(const HashArgs);

// ------------------------------------------------------------------------------------------------
// K2: finalize — value histogram p[2][65536] of one k plane pair + optional uint16 truncation.
// (compEst's first loop, ntcard.cpp:240-247; the uint16 wrap of ntcard.cpp:142-143 is the & 0xffff)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(const uint32_t* __restrict__ sketch,
                                                       uint64_t n_per_sample, uint32_t* __restrict__ p_hist,
                                                       uint16_t* __restrict__ out16)
{
	// grid.y = sample.  Counter values are tiny for almost every bucket, so each block histograms the
	// values below kLocal in LDS and only the rare large ones go to global atomics directly.
	constexpr uint32_t kLocal = 2048;
	__shared__ uint32_t lh[kLocal];
	for (uint32_t i = threadIdx.x; i < kLocal; i += blockDim.x)
		lh[i] = 0;
	__syncthreads();
	const unsigned s = blockIdx.y;
	const uint32_t* src = sketch + (uint64_t)s * n_per_sample;
	uint16_t* dst = out16 ? out16 + (uint64_t)s * n_per_sample : nullptr;
	uint32_t* p = p_hist + s * 65536u;
	uint32_t zeros = 0;
	const uint64_t n4 = n_per_sample / 4;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 v = reinterpret_cast<const uint4*>(src)[i];
		const uint32_t c[4] = { v.x & 0xffffu, v.y & 0xffffu, v.z & 0xffffu, v.w & 0xffffu };
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (c[j] == 0)
				++zeros;
			else if (c[j] < kLocal)
				atomicAdd(&lh[c[j]], 1u);
			else
				atomicAdd(p + c[j], 1u);
		}
		if (dst) {
			uint2 o;
			o.x = c[0] | (c[1] << 16);
			o.y = c[2] | (c[3] << 16);
			reinterpret_cast<uint2*>(dst)[i] = o;
		}
	}
	for (int o = 32; o > 0; o >>= 1)
		zeros += __shfl_xor(zeros, o);
	if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(p, zeros);
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < kLocal; i += blockDim.x) {
		const uint32_t n = lh[i];
		if (n) atomicAdd(p + i, n);
	}
}

// ------------------------------------------------------------------------------------------------
// K0: synthetic read generator (spec in DESIGN.md, mirrored by oracle/orc_gen_reads)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned genome_code(uint64_t gseed, uint64_t g)
{
	const uint64_t h = mix64(gseed + (g >> 5));
	return (unsigned)(h >> (2 * (g & 31))) & 3u;
}

__global__ __launch_bounds__(256) void gen_reads_kernel(unsigned char* __restrict__ out, uint64_t seed,
                                                        uint64_t first_read, uint64_t n_reads,
                                                        uint32_t read_len, uint32_t stride, uint32_t dist,
                                                        uint64_t genome_len)
{
	const uint64_t rseed = mix64(seed);
	const uint64_t gseed = mix64(seed ^ 0x47454E4F4D45ULL);
	const uint32_t acgt = 0x54474341u; // 'A','C','G','T' little-endian
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t r = first_read + i;
		unsigned char* dst = out + i * (uint64_t)stride;
		const uint64_t hr = mix64(rseed + r);
		uint64_t pos = 0, hs = 0, hm = 0, h = 0;
		unsigned rev = 0;
		if (dist != 0) {
			const uint64_t span = genome_len - read_len + 1;
			pos = __umul64hi(hr, span);
			hs = mix64(hr ^ 0xA5A5A5A5A5A5A5A5ULL);
			rev = (unsigned)(hs & 1u);
		}
		for (uint32_t j0 = 0; j0 < stride; j0 += 4) {
			uint32_t word = 0;
			for (uint32_t t = 0; t < 4; ++t) {
				const uint32_t j = j0 + t;
				uint32_t c = 'A'; // padding: never hashed, keeps waves on the clean path
				if (j < read_len) {
					if (dist == 0) {
						if ((j & 31) == 0) h = mix64(hr + (j >> 5));
						c = (acgt >> (8 * ((h >> (2 * (j & 31))) & 3u))) & 0xffu;
					} else {
						unsigned code = rev ? 3u - genome_code(gseed, pos + read_len - 1 - j)
						                    : genome_code(gseed, pos + j);
						if ((j & 3) == 0) hm = mix64(hs + 1 + (j >> 2));
						const unsigned u = (unsigned)(hm >> (16 * (j & 3))) & 0xFFFFu;
						if (u < 655u)
							c = (acgt >> (8 * ((code + 1u + (u % 3u)) & 3u))) & 0xffu;
						else if (u < 688u)
							c = 'N';
						else
							c = (acgt >> (8 * code)) & 0xffu;
					}
				}
				word |= c << (8 * t);
			}
			*reinterpret_cast<uint32_t*>(dst + j0) = word;
		}
	}
}

// K0 for the TILED slot layout (ntc_kernels.hpp, TsArgs): the same reads as gen_reads_kernel (bit-identical bases), the 16-byte
// piece c of read i at ((i / 2048 * n_chunks + c) * 2048 + i % 2048) * 16; bytes behind a read's end and the slots behind the
// last read of the last tile are 'A'
__global__ __launch_bounds__(256) void gen_reads_tiled_kernel(unsigned char* __restrict__ out, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                                              uint32_t read_len, uint32_t dist, uint64_t genome_len)
{
	const uint64_t rseed = mix64(seed);
	const uint64_t gseed = mix64(seed ^ 0x47454E4F4D45ULL);
	const uint32_t acgt = 0x54474341u; // 'A','C','G','T' little-endian
	const uint32_t n_chunks = (read_len + 15u) / 16u;
	const uint64_t n_slots = (n_reads + kTileReads - 1) / kTileReads * kTileReads;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t r = first_read + i;
		unsigned char* dst = out + ((i / kTileReads) * n_chunks * kTileReads + (i % kTileReads)) * 16u;
		const bool real = i < n_reads;
		const uint64_t hr = mix64(rseed + r);
		uint64_t pos = 0, hs = 0, hm = 0, h = 0;
		unsigned rev = 0;
		if (dist != 0) {
			const uint64_t span = genome_len - read_len + 1;
			pos = __umul64hi(hr, span);
			hs = mix64(hr ^ 0xA5A5A5A5A5A5A5A5ULL);
			rev = (unsigned)(hs & 1u);
		}
		for (uint32_t c = 0; c < n_chunks; ++c) {
			uint32_t w4[4];
			for (uint32_t q = 0; q < 4; ++q) {
				uint32_t word = 0;
				for (uint32_t t = 0; t < 4; ++t) {
					const uint32_t j = 16u * c + 4u * q + t;
					uint32_t ch = 'A';
					if (real && j < read_len) {
						if (dist == 0) {
							if ((j & 31) == 0) h = mix64(hr + (j >> 5));
							ch = (acgt >> (8 * ((h >> (2 * (j & 31))) & 3u))) & 0xffu;
						} else {
							unsigned code = rev ? 3u - genome_code(gseed, pos + read_len - 1 - j) : genome_code(gseed, pos + j);
							if ((j & 3) == 0) hm = mix64(hs + 1 + (j >> 2));
							const unsigned u = (unsigned)(hm >> (16 * (j & 3))) & 0xFFFFu;
							if (u < 655u)
								ch = (acgt >> (8 * ((code + 1u + (u % 3u)) & 3u))) & 0xffu;
							else if (u < 688u)
								ch = 'N';
							else
								ch = (acgt >> (8 * code)) & 0xffu;
						}
					}
					word |= ch << (8 * t);
				}
				w4[q] = word;
			}
			*reinterpret_cast<uint4*>(dst + (uint64_t)c * kTileReads * 16u) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
		}
	}
}

// tiled layout -> row-major slots (fallback for configurations K1c is not instantiated for; validation)
__global__ __launch_bounds__(256) void untile_kernel(const unsigned char* __restrict__ tiles, unsigned char* __restrict__ slots, uint64_t n_reads,
                                                     uint32_t read_len, uint32_t stride)
{
	const uint32_t n_chunks = (read_len + 15u) / 16u;
	const uint32_t dw = stride / 4u; // dwords per slot
	for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_reads * dw; idx += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t i = idx / dw;
		const uint32_t d = (uint32_t)(idx % dw), c = d / 4u;
		uint32_t v = 0x41414141u;
		if (c < n_chunks)
			v = *reinterpret_cast<const uint32_t*>(tiles + ((i / kTileReads) * n_chunks * kTileReads + (c * (uint64_t)kTileReads) + (i % kTileReads)) * 16u + (d & 3u) * 4u);
		reinterpret_cast<uint32_t*>(slots + i * stride)[d] = v;
	}
}

// nthll: the next batch only needs to look at hashes that can still raise SOME register: run0 > min_j M[j].
// thr = 2^(32 - (min+1)) on the top 32 bits (hash < thr<<32  <=>  at least min+1 leading zeros).
__global__ __launch_bounds__(1024) void hll_threshold_kernel(const uint32_t* __restrict__ regs, uint32_t n_regs, uint32_t* thr)
{
	__shared__ uint32_t smin[16];
	uint32_t m = 0xffffffffu;
	for (uint32_t i = threadIdx.x; i < n_regs; i += blockDim.x)
		m = regs[i] < m ? regs[i] : m;
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t other = __shfl_xor(m, o);
		m = other < m ? other : m;
	}
	if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (unsigned w = 1; w < blockDim.x / 64; ++w)
			m = smin[w] < m ? smin[w] : m;
		const uint32_t need = m + 1u; // leading zeros a hash must have to matter
		*thr = need >= 31u ? 2u : (1u << (32u - need)); // clamp: always safe to look at MORE hashes
	}
}

hipError_t launch_hll_threshold(const uint32_t* regs, uint32_t n_regs, uint32_t* thr, hipStream_t st)
{
	hipLaunchKernelGGL(hll_threshold_kernel, dim3(1), dim3(1024), 0, st, regs, n_regs, thr);
	return hipGetLastError();
}

// ---- host-side launch helpers (called from ntc_engine.hip) -----------------------------------------
hipError_t launch_hash(int mode, const HashArgs& a, unsigned grid, size_t smem, hipStream_t st)
{
	if (mode == 0)
		hipLaunchKernelGGL(nthash_kernel<0>, dim3(grid), dim3(kBlockThreads), smem, st, a);
	else
		hipLaunchKernelGGL(nthash_kernel<1>, dim3(grid), dim3(kBlockThreads), smem, st, a);
	return hipGetLastError();
}

hipError_t set_hash_smem_limit(size_t smem)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nthash_kernel<0>),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
	if (e != hipSuccess) return e;
	return hipFuncSetAttribute(reinterpret_cast<const void*>(&nthash_kernel<1>),
	                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

// merge of a dumped sketch: sketch[i] += add16[i]; the uint16 wrap happens at finalize
__global__ __launch_bounds__(256) void add_counters_kernel(uint32_t* __restrict__ sketch, const uint16_t* __restrict__ add16, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t v = reinterpret_cast<const uint32_t*>(add16)[i];
		uint2 s = reinterpret_cast<uint2*>(sketch)[i];
		s.x += v & 0xffffu;
		s.y += v >> 16;
		reinterpret_cast<uint2*>(sketch)[i] = s;
	}
}

__global__ __launch_bounds__(256) void fold_u32_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n, int take_max)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t a = dst[i], b = src[i];
		dst[i] = take_max ? (a > b ? a : b) : a + b;
	}
}
__global__ void fold_u64_kernel(unsigned long long* dst, const unsigned long long* src, uint64_t n)
{
	for (uint64_t i = threadIdx.x; i < n; i += blockDim.x)
		dst[i] += src[i];
}
hipError_t launch_fold_u32(uint32_t* dst, const uint32_t* src, uint64_t n, bool take_max, hipStream_t st)
{
	hipLaunchKernelGGL(fold_u32_kernel, dim3(4096), dim3(256), 0, st, dst, src, n, take_max ? 1 : 0);
	return hipGetLastError();
}
hipError_t launch_fold_u64(unsigned long long* dst, const unsigned long long* src, uint64_t n, hipStream_t st)
{
	hipLaunchKernelGGL(fold_u64_kernel, dim3(1), dim3(64), 0, st, dst, src, n);
	return hipGetLastError();
}

// ---- multi-device merge (ntc_merge_devices): counters travel as their low 16 bits, t_Counter wraps there (ntcard.cpp:142-143,439) ----
// dst16[i] = src32[i] mod 2^16
__global__ __launch_bounds__(256) void narrow_u16_kernel(const uint32_t* __restrict__ src, uint16_t* __restrict__ dst, uint64_t n)
{
	const uint64_t n8 = n / 8, step = (uint64_t)gridDim.x * blockDim.x, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (uint64_t i = tid; i < n8; i += step) {
		const uint4 a = reinterpret_cast<const uint4*>(src)[2 * i], b = reinterpret_cast<const uint4*>(src)[2 * i + 1];
		reinterpret_cast<uint4*>(dst)[i] = make_uint4((a.x & 0xffffu) | (a.y << 16), (a.z & 0xffffu) | (a.w << 16), (b.x & 0xffffu) | (b.y << 16), (b.z & 0xffffu) | (b.w << 16));
	}
	for (uint64_t i = n8 * 8 + tid; i < n; i += step)
		dst[i] = (uint16_t)src[i];
}
// slice 0 += slices 1 .. n_slices-1 (wrapping 16-bit adds, two counters per dword lane-wise); slices lie `stride` elements apart
__global__ __launch_bounds__(256) void sum_slices_u16_kernel(uint16_t* __restrict__ slices, uint64_t stride, uint32_t n_slices, uint64_t len)
{
	const uint64_t n8 = len / 8, step = (uint64_t)gridDim.x * blockDim.x, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	auto add2 = [](uint32_t x, uint32_t y) { return ((x + y) & 0xffffu) | ((x & 0xffff0000u) + (y & 0xffff0000u)); };
	for (uint64_t i = tid; i < n8; i += step) {
		uint4 acc = reinterpret_cast<const uint4*>(slices)[i];
		for (uint32_t r = 1; r < n_slices; ++r) {
			const uint4 v = reinterpret_cast<const uint4*>(slices + r * stride)[i];
			acc = make_uint4(add2(acc.x, v.x), add2(acc.y, v.y), add2(acc.z, v.z), add2(acc.w, v.w));
		}
		reinterpret_cast<uint4*>(slices)[i] = acc;
	}
	for (uint64_t i = n8 * 8 + tid; i < len; i += step) {
		uint16_t acc = slices[i];
		for (uint32_t r = 1; r < n_slices; ++r)
			acc = (uint16_t)(acc + slices[r * stride + i]);
		slices[i] = acc;
	}
}
// dst32[i] = src16[i]
__global__ __launch_bounds__(256) void widen_u16_kernel(const uint16_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n)
{
	const uint64_t n8 = n / 8, step = (uint64_t)gridDim.x * blockDim.x, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (uint64_t i = tid; i < n8; i += step) {
		const uint4 v = reinterpret_cast<const uint4*>(src)[i];
		reinterpret_cast<uint4*>(dst)[2 * i] = make_uint4(v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16);
		reinterpret_cast<uint4*>(dst)[2 * i + 1] = make_uint4(v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16);
	}
	for (uint64_t i = n8 * 8 + tid; i < n; i += step)
		dst[i] = src[i];
}
hipError_t launch_narrow_u16(const uint32_t* src, uint16_t* dst, uint64_t n, hipStream_t st)
{
	hipLaunchKernelGGL(narrow_u16_kernel, dim3(4096), dim3(256), 0, st, src, dst, n);
	return hipGetLastError();
}
hipError_t launch_sum_slices_u16(uint16_t* slices, uint64_t stride, uint32_t n_slices, uint64_t len, hipStream_t st)
{
	hipLaunchKernelGGL(sum_slices_u16_kernel, dim3(4096), dim3(256), 0, st, slices, stride, n_slices, len);
	return hipGetLastError();
}
hipError_t launch_widen_u16(const uint16_t* src, uint32_t* dst, uint64_t n, hipStream_t st)
{
	hipLaunchKernelGGL(widen_u16_kernel, dim3(4096), dim3(256), 0, st, src, dst, n);
	return hipGetLastError();
}

hipError_t launch_add_counters(uint32_t* sketch, const uint16_t* add16, uint64_t n, hipStream_t st)
{
	hipLaunchKernelGGL(add_counters_kernel, dim3(4096), dim3(256), 0, st, sketch, add16, n);
	return hipGetLastError();
}

hipError_t launch_finalize(const uint32_t* sketch, uint64_t n_per_sample, uint32_t* p_hist,
                           uint16_t* out16, hipStream_t st)
{
	hipLaunchKernelGGL(finalize_kernel, dim3(2048, 2), dim3(256), 0, st, sketch, n_per_sample, p_hist, out16);
	return hipGetLastError();
}

// the same for counters that are already 16 bits wide (a summed slice of the multi-GPU merge): ADDED to p_hist[65536]
__global__ __launch_bounds__(256) void value_hist_u16_kernel(const uint16_t* __restrict__ src, uint64_t n, uint32_t* __restrict__ p)
{
	constexpr uint32_t kLocal = 2048;
	__shared__ uint32_t lh[kLocal];
	for (uint32_t i = threadIdx.x; i < kLocal; i += blockDim.x)
		lh[i] = 0;
	__syncthreads();
	uint32_t zeros = 0;
	const uint64_t n8 = n / 8;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 v = reinterpret_cast<const uint4*>(src)[i];
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint32_t c = (w[j >> 1] >> (16 * (j & 1))) & 0xffffu;
			if (c == 0)
				++zeros;
			else if (c < kLocal)
				atomicAdd(&lh[c], 1u);
			else
				atomicAdd(p + c, 1u);
		}
	}
	if (blockIdx.x == 0)
		for (uint64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) {
			const uint32_t c = src[i];
			if (c == 0) ++zeros;
			else atomicAdd(p + c, 1u);
		}
	for (int o = 32; o > 0; o >>= 1)
		zeros += __shfl_xor(zeros, o);
	if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(p, zeros);
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < kLocal; i += blockDim.x)
		if (lh[i]) atomicAdd(p + i, lh[i]);
}
hipError_t launch_value_hist_u16(const uint16_t* counters, uint64_t n, uint32_t* p_hist, hipStream_t st)
{
	hipLaunchKernelGGL(value_hist_u16_kernel, dim3(2048), dim3(256), 0, st, counters, n, p_hist);
	return hipGetLastError();
}

// value histogram of an arbitrary run of counters (one "sample" of n counters), ADDED to p_hist[65536]
hipError_t launch_value_hist(const uint32_t* counters, uint64_t n, uint32_t* p_hist, hipStream_t st)
{
	hipLaunchKernelGGL(finalize_kernel, dim3(2048, 1), dim3(256), 0, st, counters, n, p_hist, (uint16_t*)nullptr);
	return hipGetLastError();
}

hipError_t launch_gen_tiled(unsigned char* out, uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint32_t dist, uint64_t glen, hipStream_t st)
{
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL(gen_reads_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, seed, first, n, len, dist, glen);
	return hipGetLastError();
}

// K1's slot table for a RAGGED tiled batch: a read's length from the tile's prefix table (ntc_submit_tiled_ragged_device)
__global__ __launch_bounds__(256) void tails_to_meta_kernel(const uint32_t* __restrict__ tails, uint64_t n_reads, uint32_t n_chunks, uint32_t* __restrict__ meta)
{
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * 256ull) {
		const uint64_t t = i / kTileReads;
		const uint32_t r = (uint32_t)(i % kTileReads);
		uint32_t tail = 0;
#pragma unroll
		for (uint32_t d = 0; d < 16u; ++d)
			tail += tails[t * 16u + d] > r;
		const uint32_t len = 16u * (n_chunks - 1u) + tail;
		meta[i] = len | (len << 16);
	}
}

hipError_t launch_tails_to_meta(const uint32_t* tails, uint64_t n_reads, uint32_t n_chunks, uint32_t* meta, hipStream_t st)
{
	const uint64_t want = (n_reads + 255) / 256;
	const unsigned grid = (unsigned)(want < 4096 ? (want ? want : 1) : 4096);
	hipLaunchKernelGGL(tails_to_meta_kernel, dim3(grid), dim3(256), 0, st, tails, n_reads, n_chunks, meta);
	return hipGetLastError();
}

hipError_t launch_untile(const unsigned char* tiles, unsigned char* slots, uint64_t n_reads, uint32_t read_len, uint32_t stride, hipStream_t st)
{
	uint64_t blocks = (n_reads * (stride / 4u) + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL(untile_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tiles, slots, n_reads, read_len, stride);
	return hipGetLastError();
}

hipError_t launch_gen(unsigned char* out, uint64_t seed, uint64_t first, uint64_t n, uint32_t len,
                      uint32_t stride, uint32_t dist, uint64_t glen, hipStream_t st)
{
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL(gen_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, seed, first, n, len,
	                   stride, dist, glen);
	return hipGetLastError();
}

} // namespace ntc
