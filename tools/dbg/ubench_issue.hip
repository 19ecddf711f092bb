// Single-wave issue rate of the VALU forms the bit-sliced walk is made of (gfx950): is a v_bitop3 slow because of its three
// VGPR operands or because of its 8-byte encoding?   hipcc --offload-arch=gfx950 -O3 ubench_issue.hip -o ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* clk, int iters)
{
	uint32_t r[32];
#pragma unroll
	for (int i = 0; i < 32; ++i)
		r[i] = threadIdx.x * 2654435761u + i;
	uint32_t a = threadIdx.x, b = threadIdx.x * 7u;
	const uint64_t t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 32; ++i) {
			if constexpr (KIND == 0) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(a), "v"(b));
			if constexpr (KIND == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x3c" : "+v"(r[i]) : "v"(a));
			if constexpr (KIND == 2) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
			if constexpr (KIND == 3) asm volatile("v_xor_b32_e64 %0, %1, %0" : "+v"(r[i]) : "v"(a));
			if constexpr (KIND == 4) asm volatile("v_not_b32_e32 %0, %0" : "+v"(r[i]));
			if constexpr (KIND == 5) asm volatile("v_xor_b32_e32 %0, %1, %0\n\tv_xor_b32_e32 %0, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
			if constexpr (KIND == 6) asm volatile("v_xnor_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
			if constexpr (KIND == 7) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(a), "s"(it));
		}
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	uint32_t x = 0;
#pragma unroll
	for (int i = 0; i < 32; ++i)
		x ^= r[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = x;
	if ((threadIdx.x & 63) == 0) atomicAdd(clk, (unsigned long long)(t1 - t0));
}
template <int KIND>
void run(const char* name, int waves_per_simd, int ops_per_item)
{
	uint32_t* out;
	unsigned long long* clk;
	hipMalloc(&out, 256 * 1024 * 4);
	hipMalloc(&clk, 8);
	const int iters = 2000, threads = 256 * waves_per_simd;
	for (int rep = 0; rep < 2; ++rep) {
		hipMemset(clk, 0, 8);
		hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, clk, iters);
		hipDeviceSynchronize();
	}
	unsigned long long h;
	hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
	const double waves = 256.0 * threads / 64;
	const double per_wave = (double)h / waves / ((double)iters * 32 * ops_per_item);
	printf("%-44s %d wave(s)/SIMD: %.2f clk per instruction per wave, %.2f per SIMD\n", name, waves_per_simd, per_wave, per_wave / waves_per_simd);
	hipFree(out);
	hipFree(clk);
}
int main()
{
	for (int w = 1; w <= 4; ++w) {
		run<0>("v_bitop3 (3 distinct VGPRs)", w, 1);
		run<1>("v_bitop3 (2 distinct VGPRs)", w, 1);
		run<7>("v_bitop3 (2 VGPRs + SGPR)", w, 1);
		run<2>("v_xor_b32_e32 (VOP2, 4 bytes)", w, 1);
		run<3>("v_xor_b32_e64 (VOP3 encoding, 8 bytes)", w, 1);
		run<6>("v_xnor_b32_e32", w, 1);
		run<4>("v_not_b32_e32 (VOP1)", w, 1);
		run<5>("2 x v_xor_b32_e32 (dependent pair)", w, 2);
	}
	return 0;
}
