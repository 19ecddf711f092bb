// tools/ubench3.hip — issue rate of v_bitop3_b32 / v_xor_b32 with 62 rotating state registers (the shape of K1b's
// step bodies) as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, uint32_t seed)
{
	uint32_t F[31], R[31];
	for (int j = 0; j < 31; ++j) {
		F[j] = seed * (j + 1) + threadIdx.x;
		R[j] = seed * (j + 77) ^ threadIdx.x;
	}
	uint32_t i0 = seed ^ threadIdx.x, i1 = seed * 3 + threadIdx.x;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			uint32_t nF[31], nR[31];
#pragma unroll
			for (int j = 0; j < 31; ++j) {
				if (MODE == 0) {
					nF[j] = __builtin_amdgcn_bitop3_b32(F[(j + 30) % 31], i0, i1, 0x96);
					nR[j] = __builtin_amdgcn_bitop3_b32(R[(j + 1) % 31], i1, i0, 0x69);
				} else if (MODE == 1) {
					nF[j] = F[(j + 30) % 31] ^ i0;
					nR[j] = R[(j + 1) % 31] ^ i1;
				} else {
					nF[j] = __builtin_amdgcn_bitop3_b32(F[(j + 30) % 31], F[(j + 7) % 31], R[(j + 3) % 31], 0x96);
					nR[j] = __builtin_amdgcn_bitop3_b32(R[(j + 1) % 31], R[(j + 9) % 31], F[(j + 5) % 31], 0x69);
				}
			}
#pragma unroll
			for (int j = 0; j < 31; ++j) {
				F[j] = nF[j];
				R[j] = nR[j];
			}
			i0 += 0x9e3779b9u;
			i1 ^= i0;
		}
	}
	uint32_t acc = 0;
	for (int j = 0; j < 31; ++j)
		acc ^= F[j] ^ R[j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, int threads)
{
	uint32_t* d;
	hipMalloc(&d, 256 * 1024 * 4);
	const int iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	k<MODE><<<256, threads>>>(d, 10, 1);
	hipDeviceSynchronize();
	hipEventRecord(a);
	k<MODE><<<256, threads>>>(d, iters, 12345);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double insts_per_wave = (double)iters * 16 * 62;
	const int waves_per_simd = threads / 256;
	const double clk = ms * 1e-3 * 2.4e9;
	printf("%-28s %4d thr/blk (%d waves/SIMD): %.3f ms  %.2f clk per instr per wave, %.2f clk per instr per SIMD (@2.4GHz)\n", name, threads,
	       waves_per_simd ? waves_per_simd : 1, ms, clk / insts_per_wave, clk / (insts_per_wave * (waves_per_simd ? waves_per_simd : 1)));
	hipFree(d);
}

int main()
{
	for (int t : {256, 512, 1024}) {
		run<0>("bitop3 (1 state + 2 shared)", t);
		run<1>("v_xor (1 state + 1 shared)", t);
		run<2>("bitop3 (3 state regs)", t);
	}
	return 0;
}
