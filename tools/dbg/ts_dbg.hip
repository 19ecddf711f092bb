// debugging harness for K1c: runs the kernel on tiles.bin and prints what the resolve stage saw per candidate
#define TS_DEBUG 1
#define TS_ONLY_K 32
#include "../../ntcard_amd/csrc/ntc_sketch_ts.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv)
{
	const char* path = argv[1];
	const uint64_t n_reads = strtoull(argv[2], 0, 10);
	const uint32_t L = atoi(argv[3]);
	const uint32_t r_bits = 18, s_bits = argc > 4 ? atoi(argv[4]) : 7;
	FILE* f = fopen(path, "rb");
	const uint32_t C = (L + 15) / 16;
	const uint64_t n_tiles = (n_reads + 2047) / 2048;
	std::vector<unsigned char> h(n_tiles * C * 2048 * 16);
	if (fread(h.data(), 1, h.size(), f) != h.size()) return 1;
	unsigned char* d_tiles;
	hipMalloc(&d_tiles, h.size());
	hipMemcpy(d_tiles, h.data(), h.size(), hipMemcpyHostToDevice);
	std::vector<uint32_t> t4(8 * 256 * 4);
	ntc::build_t4(32, t4.data());
	void* d_t4;
	hipMalloc(&d_t4, t4.size() * 4);
	hipMemcpy(d_t4, t4.data(), t4.size() * 4, hipMemcpyHostToDevice);
	uint32_t* d_sk;
	hipMalloc(&d_sk, (2u << r_bits) * 4);
	hipMemset(d_sk, 0, (2u << r_bits) * 4);
	unsigned long long* d_f1;
	hipMalloc(&d_f1, 8);
	hipMemset(d_f1, 0, 8);
	uint32_t* d_dbg;
	const size_t dbg_words = 1 << 22;
	hipMalloc(&d_dbg, dbg_words * 4);
	hipMemset(d_dbg, 0, dbg_words * 4);
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
	ntc::set_sketch_ts_smem_limit(160 * 1024 - 2048);
	hipError_t rc = ntc::launch_sketch_ts(a, (unsigned)((n_tiles + 1) / 2), 0);
	printf("launch %d sync %d\n", (int)rc, (int)hipDeviceSynchronize());
	std::vector<uint32_t> dbg(dbg_words);
	hipMemcpy(dbg.data(), d_dbg, dbg_words * 4, hipMemcpyDeviceToHost);
	unsigned long long f1;
	hipMemcpy(&f1, d_f1, 8, hipMemcpyDeviceToHost);
	printf("F1 %llu records %u\n", f1, dbg[0]);
	for (uint32_t i = 0; i < dbg[0] && i < 200000; ++i) {
		const uint32_t* r = &dbg[16 + 12 * i];
		printf("r %u w %u h %08x meta %08x d %08x %08x %08x f %08x%08x r %08x%08x rev %u fromr %u hit %u\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[8], r[7], r[10], r[9], r[11] & 1, (r[11] >> 1) & 1, (r[11] >> 2) & 1);
	}
	return 0;
}
