/*
 * ref_tool.cpp — whitebox driver around the UNMODIFIED reference translation unit.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it textually includes
 * /root/reference/ntcard.cpp where it lies (main renamed) and calls its functions, so the oracle
 * restatement and the golden fixtures can be checked against the real thing.  It is built only in
 * the build container (oracle/Makefile, target `ref`), into oracle/_ref/ (git-ignored); it does
 * not exist on the GPU box and nothing at run time there depends on it.
 *
 * Commands (all I/O through files so Python can drive it with subprocess):
 *   hash   <k> <h> <seqs.txt> <out.txt>        ntHashIterator over each line   (ntHashIterator.hpp)
 *   sthash <k> <gap> <seqs.txt> <out.txt>      stHashIterator, ntcard's seed   (ntcard.cpp:407-413)
 *   sketch <rbits> <sbits> <gap> <k,k,..> <seqs.txt> <out.bin>   ntRead/stRead (ntcard.cpp:147-171)
 *          out.bin = u64 F1[nk] then raw uint16 t_Counter[nk][2][1<<rbits]
 *   est    <rbits> <sbits> <counters.bin(one k)> <out.bin>       compEst       (ntcard.cpp:237-275)
 *          out.bin = double F0 then double fMean[65536]
 *   seeds  <out.txt>                            seedTab[0..255] and srol tables probe
 */
#define main ntcard_reference_main
#include "ntcard.cpp"
#undef main

#include <cstdio>
#include <cstring>

static std::vector<std::string> read_lines(const char* path)
{
	std::vector<std::string> v;
	std::ifstream in(path, std::ios::binary);
	std::string s;
	while (std::getline(in, s))
		v.push_back(s);
	return v;
}

static std::vector<unsigned> parse_klist(const char* s)
{
	std::vector<unsigned> ks;
	std::stringstream ss(s);
	std::string tok;
	while (std::getline(ss, tok, ','))
		ks.push_back((unsigned)atoi(tok.c_str()));
	return ks;
}

static void set_gap_seed(unsigned k, unsigned gap)
{
	opt::gap = gap;
	opt::seedSet.clear();
	if (gap != 0) {
		std::string g(gap, '0');
		std::string ng((k - gap) / 2, '1');
		std::vector<std::string> seedString;
		seedString.push_back(ng + g + ng);
		opt::seedSet = stHashIterator::parseSeed(seedString);
	}
}

int main(int argc, char** argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: ref_tool <cmd> ...\n");
		return 2;
	}
	std::string cmd = argv[1];
	if (cmd == "hash" && argc == 6) {
		unsigned k = atoi(argv[2]), h = atoi(argv[3]);
		std::vector<std::string> seqs = read_lines(argv[4]);
		FILE* out = fopen(argv[5], "w");
		for (size_t i = 0; i < seqs.size(); ++i) {
			ntHashIterator itr(seqs[i], h, k);
			std::vector<std::string> rows;
			while (itr != itr.end()) {
				char buf[256];
				int n = snprintf(buf, sizeof buf, "%zu", itr.pos());
				for (unsigned j = 0; j < h; ++j)
					n += snprintf(buf + n, sizeof buf - n, " %016llx", (unsigned long long)(*itr)[j]);
				rows.push_back(buf);
				++itr;
			}
			fprintf(out, "R %zu\n", rows.size());
			for (size_t r = 0; r < rows.size(); ++r)
				fprintf(out, "%s\n", rows[r].c_str());
		}
		fclose(out);
		return 0;
	}
	if (cmd == "sthash" && argc == 6) {
		unsigned k = atoi(argv[2]), gap = atoi(argv[3]);
		set_gap_seed(k, gap);
		std::vector<std::string> seqs = read_lines(argv[4]);
		FILE* out = fopen(argv[5], "w");
		for (size_t i = 0; i < seqs.size(); ++i) {
			stHashIterator itr(seqs[i], opt::seedSet, 1, 1, k);
			std::vector<std::string> rows;
			while (itr != itr.end()) {
				char buf[128];
				snprintf(buf, sizeof buf, "%zu %016llx", itr.pos(), (unsigned long long)(*itr)[0]);
				rows.push_back(buf);
				++itr;
			}
			fprintf(out, "R %zu\n", rows.size());
			for (size_t r = 0; r < rows.size(); ++r)
				fprintf(out, "%s\n", rows[r].c_str());
		}
		fclose(out);
		return 0;
	}
	if (cmd == "sketch" && argc == 8) {
		opt::rBits = atoi(argv[2]);
		opt::sBits = atoi(argv[3]);
		unsigned gap = atoi(argv[4]);
		std::vector<unsigned> kList = parse_klist(argv[5]);
		opt::nK = kList.size();
		set_gap_seed(kList[0], gap);
		opt::rBuck = ((size_t)1) << opt::rBits;
		opt::sMask = (((size_t)1) << (opt::sBits - 1)) - 1;
		std::vector<std::string> seqs = read_lines(argv[6]);
		size_t nCnt = opt::nK * opt::nSamp * opt::rBuck;
		uint16_t* t_Counter = new uint16_t[nCnt]();
		std::vector<size_t> tot(kList.size(), 0);
		for (size_t i = 0; i < seqs.size(); ++i) {
			if (gap == 0)
				ntRead(seqs[i], kList, t_Counter, &tot[0]);
			else
				stRead(seqs[i], kList, t_Counter, &tot[0]);
		}
		FILE* out = fopen(argv[7], "wb");
		for (size_t i = 0; i < tot.size(); ++i) {
			uint64_t v = tot[i];
			fwrite(&v, 8, 1, out);
		}
		fwrite(t_Counter, 2, nCnt, out);
		fclose(out);
		delete[] t_Counter;
		return 0;
	}
	if (cmd == "est" && argc == 6) {
		opt::rBits = atoi(argv[2]);
		opt::sBits = atoi(argv[3]);
		opt::rBuck = ((size_t)1) << opt::rBits;
		opt::sMask = (((size_t)1) << (opt::sBits - 1)) - 1;
		size_t nCnt = opt::nSamp * opt::rBuck;
		uint16_t* t_Counter = new uint16_t[nCnt]();
		FILE* in = fopen(argv[4], "rb");
		if (!in || fread(t_Counter, 2, nCnt, in) != nCnt) {
			fprintf(stderr, "ref_tool est: short read\n");
			return 1;
		}
		fclose(in);
		double F0 = 0.0;
		double* fMean = new double[65536];
		compEst(t_Counter, F0, fMean);
		FILE* out = fopen(argv[5], "wb");
		fwrite(&F0, 8, 1, out);
		fwrite(fMean, 8, 65536, out);
		fclose(out);
		return 0;
	}
	if (cmd == "seeds" && argc == 3) {
		FILE* out = fopen(argv[2], "w");
		for (unsigned c = 0; c < 256; ++c)
			fprintf(out, "S %u %016llx\n", c, (unsigned long long)seedTab[c]);
		/* rolling-table probe: srol^k(seed) as the reference composes it (nthash.hpp:246) */
		const char bases[4] = { 'A', 'C', 'G', 'T' };
		for (unsigned b = 0; b < 4; ++b)
			for (unsigned k = 0; k < 300; ++k)
				fprintf(
				    out,
				    "T %c %u %016llx\n",
				    bases[b],
				    k,
				    (unsigned long long)(msTab31l[(unsigned char)bases[b]][k % 31] | msTab33r[(unsigned char)bases[b]][k % 33]));
		fclose(out);
		return 0;
	}
	fprintf(stderr, "ref_tool: bad command line\n");
	return 2;
}
