/*
 * ref_hll_tool.cpp — whitebox driver around the UNMODIFIED reference nthll translation unit.
 * TEST INFRASTRUCTURE ONLY (build container only, output in oracle/_ref/): includes
 * /root/reference/nthll.cpp where it lies (main renamed), feeds every line of a text file to its
 * ntRead (nthll.cpp:99-105), writes the uint8 registers and prints the estimate line exactly as
 * its main does (nthll.cpp:247-258).
 *   ref_hll_tool <k> <nbits> <seqs.txt> <regs.bin>
 */
#define main nthll_reference_main
#include "nthll.cpp"
#undef main

int main(int argc, char** argv)
{
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
