/*
 * ref_hll_tool.cpp — whitebox driver around the UNMODIFIED reference nthll translation unit.
 * TEST INFRASTRUCTURE ONLY (build container only, output in oracle/_ref/): includes
 * /root/reference/nthll.cpp where it lies (main renamed), feeds every line of a text file to its
 * ntRead (nthll.cpp:99-105), writes the uint8 registers and prints the estimate line exactly as
 * its main does (nthll.cpp:247-258).
 *   ref_hll_tool <k> <nbits> <seqs.txt> <regs.bin>
 *   ref_hll_tool fullsize <seed> <n_reads> <len> <dist> <k> <nbits> <threads> <regs.bin>
 *          full-size golden maker (tools/make_fullsize_digests.py, case "hll"): the reference's ntRead over the repo's synthetic read stream
 *          (orc_gen_reads, bit-identical to the device generator K0) in chunks of 2 M reads, one private register file per thread merged by
 *          maximum exactly as the reference's main merges its per-thread mVec (nthll.cpp:236-244)
 */
#include <stdint.h>
extern "C" void orc_gen_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint32_t stride,
                              uint32_t dist, uint64_t genome_len, uint8_t* out);
#define main nthll_reference_main
#include "nthll.cpp"
#undef main

static int print_estimate(const uint8_t* tVec)
{
	double pEst = 0.0, zEst = 0.0, eEst = 0.0, alpha = 0.0;
	alpha = 1.4426 / (1 + 1.079 / opt::nBuck);
	if (opt::canon) alpha /= 2;
	for (unsigned j = 0; j < opt::nBuck; j++)
		pEst += 1.0 / ((uint64_t)1 << tVec[j]);
	zEst = 1.0 / pEst;
	eEst = alpha * opt::nBuck * opt::nBuck * zEst;
	std::cout << "F0, Exp# of distnt kmers(k=" << opt::kmLen << "): " << (unsigned long long)eEst << "\n";
	return 0;
}

int main(int argc, char** argv)
{
	if (argc == 10 && std::string(argv[1]) == "fullsize") {
		const uint64_t seed = strtoull(argv[2], 0, 10), n_total = strtoull(argv[3], 0, 10);
		const unsigned len = atoi(argv[4]), dist = atoi(argv[5]);
		opt::kmLen = atoi(argv[6]);
		opt::nBits = atoi(argv[7]);
		opt::nBuck = ((unsigned)1) << opt::nBits;
		const int threads = atoi(argv[8]);
		std::vector<uint8_t> tVec(opt::nBuck, 0);
		std::vector<std::vector<uint8_t> > mVec(threads, std::vector<uint8_t>(opt::nBuck, 0));
		const unsigned stride = (len + 3) & ~3u;
		const uint64_t chunk = 2000000;
		std::vector<uint8_t> slots((size_t)chunk * stride);
		omp_set_num_threads(threads);
		for (uint64_t first = 0; first < n_total; first += chunk) {
			const uint64_t n = std::min<uint64_t>(chunk, n_total - first);
			orc_gen_reads(seed, first, n, len, stride, dist, 100000000ull, slots.data());
#pragma omp parallel for schedule(static, 1)
			for (int sh = 0; sh < threads; ++sh) {
				const uint64_t lo = n * (uint64_t)sh / threads, hi = n * (uint64_t)(sh + 1) / threads;
				std::string seq;
				for (uint64_t i = lo; i < hi; ++i) {
					seq.assign((const char*)&slots[i * stride], len);
					ntRead(seq, mVec[sh].data());
				}
			}
		}
		for (int sh = 0; sh < threads; ++sh)
			for (unsigned j = 0; j < opt::nBuck; j++)
				if (tVec[j] < mVec[sh][j]) tVec[j] = mVec[sh][j];
		FILE* out = fopen(argv[9], "wb");
		fwrite(tVec.data(), 1, opt::nBuck, out);
		fclose(out);
		return print_estimate(tVec.data());
	}
	if (argc != 5) {
		fprintf(stderr, "usage: ref_hll_tool <k> <nbits> <seqs.txt> <regs.bin>\n");
		return 2;
	}
	opt::kmLen = atoi(argv[1]);
	opt::nBits = atoi(argv[2]);
	opt::nBuck = ((unsigned)1) << opt::nBits;
	uint8_t* tVec = new uint8_t[opt::nBuck];
	for (unsigned j = 0; j < opt::nBuck; j++)
		tVec[j] = 0;
	std::ifstream in(argv[3], std::ios::binary);
	std::string s;
	while (std::getline(in, s))
		if (s.length() >= opt::kmLen) ntRead(s, tVec);
	FILE* out = fopen(argv[4], "wb");
	fwrite(tVec, 1, opt::nBuck, out);
	fclose(out);
	double pEst = 0.0, zEst = 0.0, eEst = 0.0, alpha = 0.0;
	alpha = 1.4426 / (1 + 1.079 / opt::nBuck);
	if (opt::canon) alpha /= 2;
	for (unsigned j = 0; j < opt::nBuck; j++)
		pEst += 1.0 / ((uint64_t)1 << tVec[j]);
	zEst = 1.0 / pEst;
	eEst = alpha * opt::nBuck * opt::nBuck * zEst;
	std::cout << "F0, Exp# of distnt kmers(k=" << opt::kmLen << "): " << (unsigned long long)eEst << "\n";
	return 0;
}
