// timing harness for K1c: generates tiles on the device, runs the kernel a few times, prints per-role cycle counters (TS_TIMERS)
#define TS_TIMERS 1
#define TS_ONLY_K 32
#include "../../ntcard_amd/csrc/ntc_sketch_ts.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv)
{
	const uint64_t n_reads = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
	const uint32_t L = argc > 2 ? atoi(argv[2]) : 150;
	const uint32_t dist = argc > 3 ? atoi(argv[3]) : 1;
	const uint32_t r_bits = 27, s_bits = 7;
	const uint32_t C = (L + 15) / 16;
	const uint64_t n_tiles = (n_reads + 2047) / 2048;
	unsigned char* d_tiles;
	hipMalloc(&d_tiles, n_tiles * C * 2048 * 16);
	ntc::launch_gen_tiled(d_tiles, 1, 0, n_reads, L, dist, 100000000ull, 0);
	std::vector<uint32_t> t4(8 * 256 * 4);
	ntc::build_t4(32, t4.data());
	void* d_t4;
	hipMalloc(&d_t4, t4.size() * 4);
	hipMemcpy(d_t4, t4.data(), t4.size() * 4, hipMemcpyHostToDevice);
	uint32_t* d_sk;
	hipMalloc(&d_sk, (2ull << r_bits) * 4);
	hipMemset(d_sk, 0, (2ull << r_bits) * 4);
	unsigned long long* d_f1;
	hipMalloc(&d_f1, 8);
	hipMemset(d_f1, 0, 8);
	uint32_t *d_dbg, *d_log, *d_fill;
	hipMalloc(&d_dbg, 4096);
	const uint32_t regions = 8192, cap = 32768;
	hipMalloc(&d_log, (size_t)regions * cap * 4);
	hipMalloc(&d_fill, regions * 4);
	ntc::TsArgs a;
	memset(&a, 0, sizeof a);
	a.tiles = d_tiles;
	a.n_reads = n_reads;
	a.n_tiles = (uint32_t)n_tiles;
	a.n_chunks = C;
	a.read_len = L;
	a.k = 32;
	a.r_bits = r_bits;
	a.s_bits = s_bits;
	a.sketch0 = d_sk;
	a.f1 = d_f1;
	a.t4 = d_t4;
	a.dbg = d_dbg;
	a.log = d_log;
	a.log_fill = d_fill;
	a.log_regions = regions;
	a.log_region_cap = cap;
	ntc::set_sketch_ts_smem_limit(160 * 1024 - 2048);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const unsigned grid = (unsigned)std::min<uint64_t>((n_tiles + 1) / 2, prop.multiProcessorCount);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int it = 0; it < 3; ++it) {
		hipMemset(d_fill, 0, regions * 4);
		hipMemset(d_dbg, 0, 4096);
		hipEventRecord(e0);
		hipError_t rc = ntc::launch_sketch_ts(a, grid, 0);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		unsigned long long h[32];
		hipMemcpy(h, d_dbg, sizeof h, hipMemcpyDeviceToHost);
		const double waves = grid * 2.0;
		printf("run %d rc %d: %.3f ms, grid %u, tiles %llu (%.2f per team)\n", it, (int)rc, ms, grid, (unsigned long long)n_tiles, n_tiles / waves);
		const char* names[4] = {"F ", "R ", "A1", "A2"};
		for (int r = 0; r < 4; ++r) {
			printf("  %s per wave (kclk): total %.0f", names[r], h[r * 8] / waves / 1e3);
			if (r < 2) printf("  wait planes %.0f  wait queue %.0f", h[r * 8 + 1] / waves / 1e3, h[r * 8 + 2] / waves / 1e3);
			if (r == 2) printf("  gate loop %.0f (of which resolver turns %.0f kclk in %.0f turns)  outside the gate loop %.0f (pack loop %.0f)", h[r * 8 + 1] / waves / 1e3, h[r * 8 + 5] / waves / 1e3, h[r * 8 + 6] / waves, (h[r * 8] - (double)h[r * 8 + 1]) / waves / 1e3, h[r * 8 + 3] / waves / 1e3);
			if (r == 3) printf("  wait ring %.0f  idle %.0f  rounds %.0f kclk, %.0f passes, %.0f clk/pass", h[r * 8 + 1] / waves / 1e3, h[r * 8 + 2] / waves / 1e3, h[r * 8 + 3] / waves / 1e3,
				                  h[r * 8 + 4] / waves, (double)h[r * 8 + 3] / h[r * 8 + 4]);
			if (r == 3) printf("\n     pass phases per pass (clk): refill %.0f  ring+tables issue %.0f  xor/test %.0f  log+rest %.0f", (double)h[r * 8 + 5] / h[r * 8 + 4], (double)h[r * 8 + 6] / h[r * 8 + 4],
				                  (double)h[r * 8 + 7] / h[r * 8 + 4], ((double)h[r * 8 + 3] - h[r * 8 + 5] - h[r * 8 + 6] - h[r * 8 + 7]) / h[r * 8 + 4]);
			if (r < 2) printf("  wait exchange %.0f", h[r * 8 + 3] / waves / 1e3);
			printf("\n");
		}
	}
	return 0;
}
