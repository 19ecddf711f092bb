// per-wave cost of pack16 variants (K1c's A1 role): 1 or 2 waves per SIMD, data from registers (no memory)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
constexpr uint32_t kExpS0 = 0x47ff5554u, kExpS1 = 0x43ff41ffu;
template <int MODE>
__device__ __forceinline__ uint32_t pack16(const v4u32 v, uint32_t& bad)
{
	uint32_t lo, hi;
	if (MODE == 2) { // shift/or gather instead of the multiply
		auto g = [](uint32_t w) { uint32_t t = w & 0x06060606u; t |= t << 6; t |= t << 12; return (t >> 19) & 0xffu; };
		lo = g(v.x) | (g(v.y) << 8);
		hi = (g(v.z) << 16) | (g(v.w) << 24);
	} else {
		const uint32_t p0 = (v.x & 0x06060606u) * 0x00820820u, p1 = (v.y & 0x06060606u) * 0x00820820u;
		const uint32_t p2 = (v.z & 0x06060606u) * 0x00820820u, p3 = (v.w & 0x06060606u) * 0x00820820u;
		lo = perm(p1, p0, 0x0c0c0703u);
		hi = perm(p3, p2, 0x07030c0cu);
	}
	if (MODE == 1) { bad = 0; return lo | hi; } // no validity
	uint32_t x = perm(kExpS0, kExpS1, v.x & 0x07070707u) ^ v.x;
	x = (uint32_t)__builtin_amdgcn_bitop3_b32(perm(kExpS0, kExpS1, v.y & 0x07070707u), v.y, x, 0xbe);
	x = (uint32_t)__builtin_amdgcn_bitop3_b32(perm(kExpS0, kExpS1, v.z & 0x07070707u), v.z, x, 0xbe);
	x = (uint32_t)__builtin_amdgcn_bitop3_b32(perm(kExpS0, kExpS1, v.w & 0x07070707u), v.w, x, 0xbe);
	bad = x & 0xdfdfdfdfu;
	return lo | hi;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, uint32_t seed)
{
	v4u32 raw[8];
	for (int i = 0; i < 8; ++i) raw[i] = v4u32{seed * (i + 1) + threadIdx.x, seed ^ (i * 77u), seed + i, threadIdx.x * 3u + i};
	uint32_t acc = 0, dw = 0;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			uint32_t bad;
			acc ^= pack16<MODE>(raw[m], bad);
			dw |= (bad != 0u ? 1u : 0u) << m;
			raw[m].x += acc; // keep the inputs changing
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ dw;
}
template <int MODE>
void run(const char* name, int threads)
{
	uint32_t* d;
	hipMalloc(&d, 256 * 1024 * 4);
	const int iters = 4000;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	k<MODE><<<256, threads>>>(d, 10, 1);
	hipDeviceSynchronize();
	hipEventRecord(a);
	k<MODE><<<256, threads>>>(d, iters, 12345);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double clk = ms * 1e-3 * 2.1e9;
	printf("%-34s %d waves/SIMD: %.1f clk per pack16 per wave, %.1f per SIMD (@2.1GHz)\n", name, threads / 256, clk / (iters * 8.0), clk / (iters * 8.0) / (threads / 256));
	hipFree(d);
}
int main()
{
	for (int t : {256, 512}) {
		run<0>("pack16 (mul gather + validity)", t);
		run<1>("pack16 (mul gather, no validity)", t);
		run<2>("pack16 (shift gather + validity)", t);
	}
	return 0;
}
