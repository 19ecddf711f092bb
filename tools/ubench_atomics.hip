// tools/ubench_atomics.hip — rate of random increments into a counter array, by flavour:
//   agent-scope atomicAdd (what K1 uses), workgroup-scope atomic (executes in the issuing XCD's L2: only valid for
//   XCD-private data), 64-bit agent-scope, and plain non-atomic read-modify-write (upper bound of the load/store path).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_atomics.hip -o tools/ubench_atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>

template <int kFlavour>
__global__ __launch_bounds__(256) void k(uint32_t* region, uint64_t mask, int per_thread, uint32_t seed)
{
	uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b97f4a7c15ull + seed;
	for (int i = 0; i < per_thread; ++i) {
		x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
		uint32_t* p = region + (x & mask);
		if (kFlavour == 0) atomicAdd(p, 1u);
		if (kFlavour == 1) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (kFlavour == 2) atomicAdd(reinterpret_cast<unsigned long long*>(region + ((x & mask) & ~1ull)), 1ull);
		if (kFlavour == 3) *p = *p + 1u;
		if (kFlavour == 4) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

static float time_ms(const std::function<void()>& f)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	f(); hipDeviceSynchronize();
	hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
	const char* names[] = { "agent-scope atomicAdd u32", "workgroup-scope atomic u32", "agent-scope atomicAdd u64", "plain load+store (racy)", "system-scope atomic u32" };
	for (uint64_t words : { (1ull << 28), (1ull << 26), (1ull << 19) }) {
		uint32_t* region = nullptr;
		hipMalloc(&region, words * 4);
		hipMemset(region, 0, words * 4);
		const int blocks = 256 * 16, per_thread = 64;
		const double n = (double)blocks * 256 * per_thread;
		for (int f = 0; f < 5; ++f) {
			float ms = 0;
			if (f == 0) ms = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, region, words - 1, per_thread, 3u); });
			if (f == 1) ms = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, region, words - 1, per_thread, 3u); });
			if (f == 2) ms = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, region, words - 1, per_thread, 3u); });
			if (f == 3) ms = time_ms([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, region, words - 1, per_thread, 3u); });
			if (f == 4) ms = time_ms([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, region, words - 1, per_thread, 3u); });
			printf("%8.1f MiB  %-28s %8.3f ms  %7.2f G/s\n", words * 4.0 / (1 << 20), names[f], ms, n / ms / 1e6);
		}
		hipFree(region);
	}
	// is the ceiling per CU or chip-wide?  same total work from fewer / more resident blocks (2 MiB region, agent scope)
	{
		uint32_t* region = nullptr;
		hipMalloc(&region, 1 << 21);
		hipMemset(region, 0, 1 << 21);
		for (int blocks : { 32, 64, 128, 256, 512, 2048 }) {
			const int per_thread = 64 * 4096 / blocks;
			const double n = (double)blocks * 256 * per_thread;
			float ms = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, region, (1ull << 19) - 1, per_thread, 3u); });
			printf("2 MiB region, %4d blocks x 256 threads: %8.3f ms  %7.2f G/s\n", blocks, ms, n / ms / 1e6);
		}
		// one wave per block, 1 / 16 / 64 active lanes per atomic instruction
		hipFree(region);
	}
	return 0;
}
