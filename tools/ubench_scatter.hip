// tools/ubench_scatter.hip — what does HBM make of the WRITE PATTERN of a radix partition with many ways?  (round 6, DESIGN 5 "the sketch update")
//
// The apply's partition passes (ntc_apply.hip, split_kernel) sort R keys per round in LDS and append ~R / D keys to each of D private runs.  With D = 128 the
// segments are ~256 bytes; a single pass over 2^12 .. 2^13 ways (which would remove a whole pass: 6 of the 16 bytes the apply moves per key) leaves
// segments of 8 .. 64 bytes.  This bench issues exactly that pattern — G workgroups x D runs, per round a coalesced READ of 4 R bytes and `seg` elements
// of 2 or 4 bytes appended to every run — without any sorting, and reports the bytes moved per second, next to a plain copy.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter.hip -o tools/ubench_scatter && tools/ubench_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                                     \
	do {                                                                                             \
		hipError_t e_ = (x);                                                                         \
		if (e_ != hipSuccess) {                                                                      \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                  \
			exit(1);                                                                                 \
		}                                                                                            \
	} while (0)

template <typename T>
__global__ __launch_bounds__(1024) void part_pattern(const uint32_t* __restrict__ in, T* __restrict__ out, uint32_t D, uint32_t seg, uint32_t rounds,
                                                     uint32_t run_cap, uint32_t misalign)
{
	const uint32_t tid = threadIdx.x, w = blockIdx.x;
	const uint32_t R = D * seg; // keys per round
	const uint32_t* src = in + (uint64_t)w * rounds * R;
	T* dst = out + (uint64_t)w * D * run_cap;
	uint32_t acc = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		for (uint32_t e = tid; e < R; e += 1024)
			acc ^= src[(uint64_t)r * R + e];
		for (uint32_t e = tid; e < R; e += 1024) {
			const uint32_t d = e / seg, pos = e - d * seg;
			dst[(uint64_t)d * run_cap + (misalign ? (d * 5u) % seg : 0u) + r * seg + pos] = (T)(acc + e);
		}
	}
}

// the same bytes, perfectly coalesced: read 4 B, write sizeof(T) per element
template <typename T> __global__ __launch_bounds__(1024) void copy_pattern(const uint32_t* __restrict__ in, T* __restrict__ out, uint64_t n)
{
	for (uint64_t i = blockIdx.x * 1024ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024ull)
		out[i] = (T)in[i];
}

// read side of the count pass: slice d gathers run (w, d) of every workgroup w — G short runs of `len` elements
template <typename T> __global__ __launch_bounds__(1024) void gather_pattern(const T* __restrict__ in, uint32_t G, uint32_t D, uint32_t run_cap, uint32_t len, uint32_t* sink)
{
	uint32_t acc = 0;
	for (uint32_t d = blockIdx.x; d < D; d += gridDim.x)
		for (uint32_t w = 0; w < G; ++w) {
			const T* src = in + ((uint64_t)w * D + d) * run_cap;
			for (uint32_t i = threadIdx.x; i < len; i += 1024)
				acc += src[i];
		}
	if (acc == 0x12345678u) *sink = acc;
}

int main()
{
	const uint64_t N = 256ull << 20; // keys
	uint32_t *in, *sink;
	void* out;
	CHECK(hipMalloc(&in, N * 4));
	CHECK(hipMalloc(&out, N * 4 * 2 + (64u << 20)));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(in, 1, N * 4));
	CHECK(hipMemset(out, 0, N * 4 * 2));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	auto timeit = [&](auto&& launch) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			CHECK(hipEventRecord(e0));
			launch();
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (rep && ms < best) best = ms;
		}
		CHECK(hipGetLastError());
		return best;
	};
	{
		const float a = timeit([&] { copy_pattern<uint32_t><<<1024, 1024>>>(in, (uint32_t*)out, N); });
		const float b = timeit([&] { copy_pattern<uint16_t><<<1024, 1024>>>(in, (uint16_t*)out, N); });
		printf("copy r4 w4: %.3f ms  %.2f TB/s     copy r4 w2: %.3f ms  %.2f TB/s   (%.0f M keys)\n", a, N * 8 / a * 1e-9, b, N * 6 / b * 1e-9, N * 1e-6);
	}
	printf("%5s %5s %4s %3s %5s | %8s %9s %10s | %s\n", "G", "D", "seg", "EB", "align", "ms", "TB/s", "Gkeys/s", "gather ms (TB/s)");
	const uint32_t Gs[] = {128, 256, 512};
	const uint32_t Ds[] = {128, 1024, 4096, 8192};
	const uint32_t segs[] = {4, 8, 16, 32, 64, 128};
	for (uint32_t eb : {2u, 4u})
		for (uint32_t G : Gs)
			for (uint32_t D : Ds)
				for (uint32_t seg : segs)
					for (uint32_t mis : {0u, 1u}) {
						const uint64_t R = (uint64_t)D * seg;
						if (R > 65536 || R < 4096) continue; // what a workgroup can stage in LDS / at least 4 keys per thread
						if (mis && (seg > 32 || G != 256)) continue;
						const uint32_t rounds = (uint32_t)(N / (G * R));
						if (rounds < 2) continue;
						const uint32_t run_cap = rounds * seg + seg;
						if ((uint64_t)G * D * run_cap * eb > N * 8) continue;
						float ms;
						if (eb == 2) ms = timeit([&] { part_pattern<uint16_t><<<G, 1024>>>(in, (uint16_t*)out, D, seg, rounds, run_cap, mis); });
						else ms = timeit([&] { part_pattern<uint32_t><<<G, 1024>>>(in, (uint32_t*)out, D, seg, rounds, run_cap, mis); });
						const double keys = (double)G * rounds * R;
						float gms = 0;
						if (D >= 1024 && !mis) {
							if (eb == 2) gms = timeit([&] { gather_pattern<uint16_t><<<512, 1024>>>((const uint16_t*)out, G, D, run_cap, rounds * seg, sink); });
							else gms = timeit([&] { gather_pattern<uint32_t><<<512, 1024>>>((const uint32_t*)out, G, D, run_cap, rounds * seg, sink); });
						}
						printf("%5u %5u %4u %3u %5s | %8.3f %9.2f %10.1f | %.3f (%.2f)\n", G, D, seg, eb, mis ? "off" : "on", ms, keys * (4 + eb) / ms * 1e-9, keys / ms * 1e-6, gms,
						       gms > 0 ? keys * eb / gms * 1e-9 : 0.0);
						fflush(stdout);
					}
	return 0;
}
