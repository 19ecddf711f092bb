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
//                         each; per-workgroup private output runs, so no global cursor atomics)
//   A3     count_kernel   one workgroup per slice of 2^15 counters: histogram of the slice's keys in LDS
//                         (ds_add, single writer per slice), then one coalesced `sketch[i] += n` sweep.
//
// Every run has a fixed capacity; a key that does not fit falls back to `atomicAdd(sketch + key, 1)`, which is
// exact too (just slower), so skewed inputs (one k-mer repeated millions of times) stay correct.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntc_kernels.hpp"

namespace ntc {

namespace {

constexpr uint32_t kSplitRound = 2048; // keys per workgroup round (8 per thread)

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

// A1/A2: partition the keys of this workgroup's input runs by digit = (key >> shift) & (2^bits - 1).
__global__ __launch_bounds__(256) void split_kernel(const SplitArgs a)
{
	__shared__ uint32_t hist[256], excl[256], gcur[256], rankc[256];
	__shared__ uint32_t sorted[kSplitRound];
	const uint32_t tid = threadIdx.x, w = blockIdx.x;
	const uint32_t nb = 1u << a.bits, dmask = nb - 1u;
	gcur[tid] = 0;
	uint32_t seg, step;
	if (a.mode == 0) {
		seg = w;
		step = gridDim.x;
	} else {
		const uint32_t b = w / a.parts, p = w % a.parts;
		seg = p * a.nb_in + b;
		step = a.parts * a.nb_in;
	}
	uint32_t* const outw = a.out + (uint64_t)w * nb * a.out_cap;
	for (; seg < a.n_in; seg += step) {
		uint32_t n = a.in_cnt[seg];
		n = n < a.in_cap ? n : a.in_cap;
		const uint32_t* src = a.in + (uint64_t)seg * a.in_cap;
		for (uint32_t base = 0; base < n; base += kSplitRound) {
			const uint32_t m = n - base < kSplitRound ? n - base : kSplitRound;
			hist[tid] = 0;
			rankc[tid] = 0;
			__syncthreads();
			uint32_t key[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t i = (uint32_t)j * 256u + tid;
				key[j] = 0;
				if (i < m) {
					key[j] = src[base + i];
					atomicAdd(&hist[(key[j] >> a.shift) & dmask], 1u);
				}
			}
			__syncthreads();
			if (tid < 64) { // exclusive scan of the 256 digit counts
				const uint32_t v0 = hist[4 * tid], v1 = hist[4 * tid + 1], v2 = hist[4 * tid + 2], v3 = hist[4 * tid + 3];
				const uint32_t s = v0 + v1 + v2 + v3;
				const uint32_t b0 = wave_incl_scan(s) - s;
				excl[4 * tid] = b0;
				excl[4 * tid + 1] = b0 + v0;
				excl[4 * tid + 2] = b0 + v0 + v1;
				excl[4 * tid + 3] = b0 + v0 + v1 + v2;
			}
			__syncthreads();
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t i = (uint32_t)j * 256u + tid;
				if (i < m) {
					const uint32_t d = (key[j] >> a.shift) & dmask;
					sorted[excl[d] + atomicAdd(&rankc[d], 1u)] = key[j];
				}
			}
			__syncthreads();
			for (uint32_t i = tid; i < m; i += 256u) {
				const uint32_t kk = sorted[i];
				const uint32_t d = (kk >> a.shift) & dmask;
				const uint32_t off = gcur[d] + (i - excl[d]);
				if (off < a.out_cap)
					outw[(uint64_t)d * a.out_cap + off] = kk;
				else
					atomicAdd(a.sketch + kk, 1u); // run is full: apply directly (exact, slower)
			}
			__syncthreads();
			gcur[tid] += hist[tid]; // the same thread zeroes hist[tid] at the top of the next round
		}
	}
	__syncthreads();
	if (tid < nb) a.out_cnt[(uint64_t)w * nb + tid] = gcur[tid] < a.out_cap ? gcur[tid] : a.out_cap;
}

// A3: one workgroup per slice of 2^slice_bits counters (<= 2^15): LDS histogram of the slice's keys, then
// sketch[slice] += histogram.  The slice has exactly one writer, so the sweep needs no atomics.
__global__ __launch_bounds__(1024) void count_kernel(const CountArgs a)
{
	extern __shared__ __align__(16) uint32_t cnt[]; // [1 << slice_bits]
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	const uint32_t n_cnt = 1u << a.slice_bits, cmask = n_cnt - 1u;
	for (uint32_t slice = blockIdx.x; slice < a.n_slices; slice += gridDim.x) {
		for (uint32_t i = tid; i < n_cnt / 4; i += nt)
			reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0, 0, 0, 0);
		__syncthreads();
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
		uint32_t any = 0;
		for (uint32_t t = 0; t < seg_cnt; ++t) {
			const uint32_t seg = t * seg_mul + seg_add;
			uint32_t n = a.in_cnt[seg];
			n = n < a.in_cap ? n : a.in_cap;
			any |= n;
			const uint32_t* src = a.in + (uint64_t)seg * a.in_cap;
			for (uint32_t i = tid; i < n; i += nt)
				atomicAdd(&cnt[src[i] & cmask], 1u);
		}
		__syncthreads();
		if (any != 0) { // wave-uniform (every thread saw the same counts)
			uint32_t* dst = a.sketch + ((uint64_t)slice << a.slice_bits);
			for (uint32_t i = tid; i < n_cnt / 4; i += nt) {
				const uint4 c = reinterpret_cast<const uint4*>(cnt)[i];
				if ((c.x | c.y | c.z | c.w) != 0u) {
					uint4 s = reinterpret_cast<uint4*>(dst)[i];
					s.x += c.x;
					s.y += c.y;
					s.z += c.z;
					s.w += c.w;
					reinterpret_cast<uint4*>(dst)[i] = s;
				}
			}
		}
		__syncthreads();
	}
}

hipError_t launch_split(const SplitArgs& a, unsigned grid, hipStream_t st)
{
	hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_count(const CountArgs& a, unsigned grid, hipStream_t st)
{
	const size_t smem = sizeof(uint32_t) << a.slice_bits;
	hipLaunchKernelGGL(count_kernel, dim3(grid), dim3(1024), smem, st, a);
	return hipGetLastError();
}

hipError_t set_apply_smem_limit()
{
	return hipFuncSetAttribute(reinterpret_cast<const void*>(&count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

} // namespace ntc
