// ubench.hip — MI355X micro-benchmarks that size the hash kernel's ceilings:
//   1. issue rate of the integer VALU ops the rolling step is made of (per CU per clock)
//   2. LDS table-lookup rate (ds_read_b128 + ds_read_b32 from a 16-slot table)
//   3. random global atomic-add rate into a 1 GiB / 2 MiB uint32 region (sketch update pattern)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
	do {                                                                                           \
		hipError_t e = (x);                                                                        \
		if (e != hipSuccess) {                                                                     \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                 \
			exit(1);                                                                               \
		}                                                                                          \
	} while (0)

constexpr int ITERS = 4096;

template <int WHICH>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t* out, uint32_t seed)
{
	uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13,
	         a7 = a0 * 17;
	uint32_t b = a0 ^ 0x9e3779b9u;
	uint32_t c = seed | 1u;
	for (int i = 0; i < ITERS; ++i) {
		// 8 independent chains, 1 op each per line; 8 lines = 64 ops per iteration
#define LINE(OPS) asm volatile(OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(c) : "vcc");
#define REP8(T)                                                                                    \
	LINE(T) LINE(T) LINE(T) LINE(T) LINE(T) LINE(T) LINE(T) LINE(T)
		if (WHICH == 0) {
			REP8("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\n"
			     "v_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8")
		} else if (WHICH == 1) {
			REP8("v_alignbit_b32 %0, %0, %8, 31\nv_alignbit_b32 %1, %1, %8, 31\nv_alignbit_b32 %2, %2, %8, 31\n"
			     "v_alignbit_b32 %3, %3, %8, 31\nv_alignbit_b32 %4, %4, %8, 31\nv_alignbit_b32 %5, %5, %8, 31\n"
			     "v_alignbit_b32 %6, %6, %8, 31\nv_alignbit_b32 %7, %7, %8, 31")
		} else if (WHICH == 2) {
			REP8("v_perm_b32 %0, %0, %8, %8\nv_perm_b32 %1, %1, %8, %8\nv_perm_b32 %2, %2, %8, %8\n"
			     "v_perm_b32 %3, %3, %8, %8\nv_perm_b32 %4, %4, %8, %8\nv_perm_b32 %5, %5, %8, %8\n"
			     "v_perm_b32 %6, %6, %8, %8\nv_perm_b32 %7, %7, %8, %8")
		} else if (WHICH == 3) {
			REP8("v_bfe_u32 %0, %0, 1, 31\nv_bfe_u32 %1, %1, 1, 31\nv_bfe_u32 %2, %2, 1, 31\nv_bfe_u32 %3, %3, 1, 31\n"
			     "v_bfe_u32 %4, %4, 1, 31\nv_bfe_u32 %5, %5, 1, 31\nv_bfe_u32 %6, %6, 1, 31\nv_bfe_u32 %7, %7, 1, 31")
		} else if (WHICH == 4) {
			REP8("v_min_u32 %0, %0, %8\nv_min_u32 %1, %1, %8\nv_min_u32 %2, %2, %8\nv_min_u32 %3, %3, %8\n"
			     "v_min_u32 %4, %4, %8\nv_min_u32 %5, %5, %8\nv_min_u32 %6, %6, %8\nv_min_u32 %7, %7, %8")
		} else if (WHICH == 5) {
			REP8("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\n"
			     "v_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7")
		} else if (WHICH == 6) {
			REP8("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n"
			     "v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n"
			     "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n"
			     "v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x96")
		} else if (WHICH == 7) {
			REP8("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_u32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %8, vcc\n"
			     "v_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_u32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %8, vcc")
		} else if (WHICH == 8) {
			REP8("v_add_lshl_u32 %0, %0, %8, 1\nv_add_lshl_u32 %1, %1, %8, 1\nv_add_lshl_u32 %2, %2, %8, 1\n"
			     "v_add_lshl_u32 %3, %3, %8, 1\nv_add_lshl_u32 %4, %4, %8, 1\nv_add_lshl_u32 %5, %5, %8, 1\n"
			     "v_add_lshl_u32 %6, %6, %8, 1\nv_add_lshl_u32 %7, %7, %8, 1")
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// LDS table lookups like the hash kernel: ds_read_b128 + ds_read_b32 at a random 16-slot offset
__global__ __launch_bounds__(256) void lds_kernel(uint32_t* out, uint32_t seed, int with_b32)
{
	__shared__ __align__(16) uint32_t tab[16 * 4 * 2];
	for (int i = threadIdx.x; i < 128; i += 256)
		tab[i] = i * 2654435761u + seed;
	__syncthreads();
	uint32_t x = threadIdx.x * 2654435761u + seed, acc = 0;
	const unsigned char* base = reinterpret_cast<const unsigned char*>(tab);
	for (int i = 0; i < ITERS; ++i) {
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint32_t off = (x >> (4 * j)) & 0xf0u;
			const uint4 t = *reinterpret_cast<const uint4*>(base + off);
			acc ^= t.x ^ t.y ^ t.z ^ t.w;
			if (with_b32) acc ^= *reinterpret_cast<const uint32_t*>(base + 256 + off);
		}
		x = x * 1664525u + 1013904223u + acc;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// random atomics: each thread fires `per_thread` adds at pseudo-random words of a region
__global__ __launch_bounds__(256) void atomic_kernel(uint32_t* region, uint64_t mask, int per_thread, uint32_t seed)
{
	uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + seed;
	for (int i = 0; i < per_thread; ++i) {
		x ^= x >> 29;
		x *= 0xBF58476D1CE4E5B9ULL;
		x ^= x >> 32;
		atomicAdd(region + (x & mask), 1u);
	}
}

template <typename F>
float time_ms(F f, int reps = 3)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	f();
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int r = 0; r < reps; ++r) {
		CHECK(hipEventRecord(a));
		f();
		CHECK(hipEventRecord(b));
		CHECK(hipEventSynchronize(b));
		float ms;
		CHECK(hipEventElapsedTime(&ms, a, b));
		if (ms < best) best = ms;
	}
	return best;
}

int main()
{
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	printf("device: %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
	uint32_t* out;
	const int blocks = cus * 8;
	CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
	const char* names[] = { "v_xor_b32", "v_alignbit_b32", "v_perm_b32", "v_bfe_u32", "v_min_u32", "v_lshlrev_b32",
		                    "v_bitop3_b32", "v_cmp+v_cndmask(pair)", "v_add_lshl_u32" };
	for (int w = 0; w < 9; ++w) {
		float ms = 0;
		auto run = [&]() {
			switch (w) {
			case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 3: hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 4: hipLaunchKernelGGL(valu_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 5: hipLaunchKernelGGL(valu_kernel<5>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 6: hipLaunchKernelGGL(valu_kernel<6>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 7: hipLaunchKernelGGL(valu_kernel<7>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			case 8: hipLaunchKernelGGL(valu_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, 1u); break;
			}
		};
		ms = time_ms(run);
		const double ops_per_thread = (double)ITERS * 64.0;
		const double lane_ops = ops_per_thread * blocks * 256.0;
		printf("%-24s %8.3f ms  %7.2f T lane-ops/s  = %6.1f lane-ops/clk/CU @2.4GHz\n", names[w], ms,
		       lane_ops / ms / 1e9, lane_ops / (ms * 1e-3) / cus / 2.4e9);
	}
	for (int wb = 0; wb < 2; ++wb) {
		float ms = time_ms([&]() { hipLaunchKernelGGL(lds_kernel, dim3(blocks), dim3(256), 0, 0, out, 7u, wb); });
		const double lookups = (double)ITERS * 8 * blocks * 256.0;
		printf("lds lookup b128%s      %8.3f ms  %7.2f T lookups/s = %6.2f lookups/clk/CU @2.4GHz\n", wb ? "+b32" : "     ",
		       ms, lookups / ms / 1e9, lookups / (ms * 1e-3) / cus / 2.4e9);
	}
	for (uint64_t words : { (1ull << 28), (1ull << 19), (1ull << 14) }) {
		uint32_t* region;
		CHECK(hipMalloc(&region, words * 4));
		CHECK(hipMemset(region, 0, words * 4));
		const int per_thread = 64;
		const int ab = cus * 16;
		float ms = time_ms([&]() { hipLaunchKernelGGL(atomic_kernel, dim3(ab), dim3(256), 0, 0, region, words - 1, per_thread, 3u); });
		const double n = (double)ab * 256 * per_thread;
		printf("random atomicAdd over %8.1f MiB: %8.3f ms  %7.2f G atomics/s\n", words * 4.0 / (1 << 20), ms, n / ms / 1e6);
		CHECK(hipFree(region));
	}
	return 0;
}
