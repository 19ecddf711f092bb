// how fast can 2 waves per CU stream tiled chunks with a rolling window of 32 x 1 KiB buffer loads each (K1c's A1 pattern)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
template <int AUX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const unsigned char* tiles, uint32_t* out, uint32_t n_chunks_total)
{
	const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t me = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
	const uint32_t voff = lane * 16u;
	v4u32 raw[32];
	v4u32 acc = {0, 0, 0, 0};
	auto rsrc = [&](uint32_t ch) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(tiles + (size_t)ch * 32768u), 0, 32768, 0x00020000); };
	if (me < n_chunks_total) {
		auto r0 = rsrc(me);
#pragma unroll
		for (int m = 0; m < 32; ++m) raw[m] = __builtin_amdgcn_raw_buffer_load_b128(r0, voff + (m & 3) * 1024u, (m >> 2) * 4096, AUX);
	}
	for (uint32_t ch = me; ch < n_chunks_total; ch += nw) {
		const uint32_t nx = ch + nw < n_chunks_total ? ch + nw : ch;
		auto rn = rsrc(nx);
#pragma unroll
		for (int m = 0; m < 32; ++m) {
			acc ^= raw[m];
			raw[m] = __builtin_amdgcn_raw_buffer_load_b128(rn, voff + (m & 3) * 1024u, (m >> 2) * 4096, AUX);
			if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
		}
	}
	out[blockIdx.x * 64 * WAVES + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
template <int AUX, int WAVES>
void run(const unsigned char* d, uint32_t* o, uint32_t nch)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	k<AUX, WAVES><<<256, 64 * WAVES>>>(d, o, nch);
	hipDeviceSynchronize();
	hipEventRecord(a);
	k<AUX, WAVES><<<256, 64 * WAVES>>>(d, o, nch);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	printf("aux %d, %d waves/CU: %.3f ms for %.2f GB = %.2f TB/s\n", AUX, WAVES, ms, nch * 32768.0 / 1e9, nch * 32768.0 / ms / 1e9);
}
int main()
{
	const uint32_t nch = 48830; // 4883 tiles x 10 chunks
	unsigned char* d; uint32_t* o;
	hipMalloc(&d, (size_t)nch * 32768); hipMemset(d, 65, (size_t)nch * 32768);
	hipMalloc(&o, 256 * 1024 * 4);
	run<0, 2>(d, o, nch); run<2, 2>(d, o, nch); run<1, 2>(d, o, nch); run<3, 2>(d, o, nch);
	run<0, 4>(d, o, nch); run<2, 4>(d, o, nch); run<0, 8>(d, o, nch); run<2, 8>(d, o, nch);
	return 0;
}
