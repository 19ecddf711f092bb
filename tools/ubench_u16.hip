// tools/ubench_u16.hip — A/B of the two t_Counter layouts under the direct-atomic update.
//   A: one uint32 per counter, atomicAdd without return (what the engine does; wraps to uint16 on read-back)
//   B: two uint16 counters per uint32 word; the low half's wrap 0xffff -> 0 carries into the high half, so the add must
//      RETURN the old word and the one thread that sees the wrap takes the carry back out (exact, order-independent)
// Key streams: u = uniform over 2^28 counters, g = 3/4 of the keys drawn from a hot set of 2^16 (repeat-rich genome).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/ubench_u16.hip -o /tmp/ubench_u16 && /tmp/ubench_u16
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
	return (uint32_t)x;
}
__global__ void gen(uint32_t* keys, uint64_t n, int hot)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t a = mix(i * 2 + 1), b = mix(i * 2 + 2);
		uint32_t key = a & ((1u << 28) - 1u);
		if (hot && (b & 3u)) key = mix(b >> 2 & 0xffffu) & ((1u << 28) - 1u);
		keys[i] = key;
	}
}
__global__ void add32(const uint32_t* keys, uint64_t n, uint32_t* s)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		atomicAdd(s + keys[i], 1u);
}
__global__ void add16(const uint32_t* keys, uint64_t n, uint32_t* s)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t key = keys[i];
		if (key & 1u) {
			atomicAdd(s + (key >> 1), 65536u); // the high half wraps by itself
		} else {
			const uint32_t old = atomicAdd(s + (key >> 1), 1u);
			if ((old & 0xffffu) == 0xffffu) atomicSub(s + (key >> 1), 65536u);
		}
	}
}
__global__ void check(const uint32_t* s32, const uint32_t* s16, uint64_t words, unsigned long long* bad)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t want = (s32[2 * i] & 0xffffu) | (s32[2 * i + 1] << 16);
		if (want != s16[i]) atomicAdd(bad, 1ull);
	}
}
int main()
{
	const uint64_t n = 64ull << 20, counters = 1ull << 28;
	uint32_t *keys, *s32, *s16;
	unsigned long long* bad;
	hipMalloc(&keys, n * 4); hipMalloc(&s32, counters * 4); hipMalloc(&s16, counters * 2); hipMalloc(&bad, 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int hot = 0; hot < 2; ++hot) {
		gen<<<4096, 256>>>(keys, n, hot);
		hipMemset(s32, 0, counters * 4); hipMemset(s16, 0, counters * 2); hipMemset(bad, 0, 8);
		float t32 = 0, t16 = 0;
		const int reps = hot ? 24 : 3; // the hot set has to wrap (> 65535 hits per counter) for the carry path to run
		for (int r = 0; r < reps; ++r) {
			float t;
			hipEventRecord(e0); add32<<<8192, 256>>>(keys, n, s32); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1); t32 += t;
			hipEventRecord(e0); add16<<<8192, 256>>>(keys, n, s16); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1); t16 += t;
		}
		check<<<4096, 256>>>(s32, s16, counters / 2, bad);
		unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
		printf("%s: u32 no-return %.3f ms / %llu M keys (%.1f G/s) | packed u16 return+carry fix %.3f ms (%.1f G/s) | mismatching words %llu\n",
		       hot ? "g (hot set)" : "u (uniform)", t32 / reps, (unsigned long long)(n >> 20), n / (t32 / reps) / 1e6, t16 / reps, n / (t16 / reps) / 1e6, hb);
	}
	return 0;
}
