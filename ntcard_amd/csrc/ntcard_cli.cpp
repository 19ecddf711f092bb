// ntcard_cli.cpp — drop-in `ntcard` command line front end over the MI355X engine (C ABI).
//
// Mirrors the reference's process boundary B1: options and their validation
// (ntcard.cpp:69-87,325-405), `@list` expansion (:415-425), the input-size rule (:427-431), one worker
// per input FILE (:445-446), record splitting for FASTQ / FASTA / SAM incl. its quirks (:105-130,
// 173-235), decompression by file extension through the same external tools the reference pipes
// through (Common/Uncompress.cpp:32-53), and the two output formats (:277-315) plus the
// "Runtime(sec)" line (:476).  Everything between "a sequence was parsed" and "the counters are
// final" runs on the GPU through include/ntcard_hip.h; there is no CPU hashing path in this binary.
#include <getopt.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "cli_common.hpp"

const char* const cli::kProgram = "ntCard";

namespace {

using namespace cli;
const char* const PROGRAM = cli::kProgram;

void usage(std::ostream& os)
{
	os << "Usage: " << PROGRAM << " [OPTION]... FILE(S)...\n"
	   << "Estimates the k-mer coverage histogram of FILE(S) on an AMD MI355X.\n\n"
	   << "Input: fastq, fasta, sam (bam through samtools), plain or .gz/.bz2/.xz/.zip compressed.\n"
	   << "A file of file names (one per line) can be given with an @ prefix.\n\n"
	   << " Options:\n\n"
	   << "  -t, --threads=N\tparser threads, one per input file at a time [1]\n"
	   << "  -k, --kmer=N\tk-mer length, or a comma separated list of lengths\n"
	   << "  -g, --gap=N\tlength of the gap in a gapped seed [0]; g mod 2 must equal k mod 2; single k only\n"
	   << "  -c, --cov=N\tlargest coverage reported [1000]\n"
	   << "  -p, --pref=STRING\tprefix of the per-k output files <STRING>_k<k>.hist\n"
	   << "  -o, --output=STRING\tsingle tab separated output file (k, f, n)\n"
	   << "      --help\tdisplay this help and exit\n"
	   << "      --version\toutput version information and exit\n";
}

struct Options {
	unsigned threads = 1;
	unsigned gap = 0;
	unsigned r_bits = 27; // ntcard.cpp:58
	unsigned s_bits = 11; // ntcard.cpp:59
	unsigned cov_max = 1000;
	std::string prefix, output;
	std::vector<unsigned> klist;
};

void process_file(const std::string& path, ntc_engine* eng)
{
	LineReader in(path);
	std::string first;
	in.getline(first);
	bool sam_has_header = true;
	const unsigned type = sniff(first, sam_has_header);
	Batcher batch(eng);
	if (type == 0)
		parse_fastq_blocks(in, eng);
	else if (type == 1)
		parse_fasta_blocks(in, eng, batch);
	else if (type == 2)
		parse_sam_blocks(in, eng, batch, first, sam_has_header);
	else {
		std::cerr << "Error in reading file: " << path << std::endl; // ntcard.cpp:459-462
		std::exit(EXIT_FAILURE);
	}
	batch.flush();
}

} // namespace

int main(int argc, char** argv)
{
	const auto t_start = std::chrono::steady_clock::now();
	static const char shortopts[] = "t:s:r:k:c:l:p:f:o:g:";
	enum { OPT_HELP = 1, OPT_VERSION };
	static const struct option longopts[] = { { "threads", required_argument, nullptr, 't' },
		                                      { "kmer", required_argument, nullptr, 'k' },
		                                      { "gap", required_argument, nullptr, 'g' },
		                                      { "cov", required_argument, nullptr, 'c' },
		                                      { "rbit", required_argument, nullptr, 'r' },
		                                      { "sbit", required_argument, nullptr, 's' },
		                                      { "output", required_argument, nullptr, 'o' },
		                                      { "pref", required_argument, nullptr, 'p' },
		                                      { "help", no_argument, nullptr, OPT_HELP },
		                                      { "version", no_argument, nullptr, OPT_VERSION },
		                                      { nullptr, 0, nullptr, 0 } };
	Options opt;
	bool die = false;
	for (int c; (c = getopt_long(argc, argv, shortopts, longopts, nullptr)) != -1;) {
		bool clean = true;
		switch (c) {
		case '?': die = true; break;
		case 't': clean = parse_value(optarg, opt.threads); break;
		case 's': clean = parse_value(optarg, opt.s_bits); break;
		case 'r': clean = parse_value(optarg, opt.r_bits); break;
		case 'c':
			clean = parse_value(optarg, opt.cov_max);
			if (opt.cov_max > 65535) opt.cov_max = 65535;
			break;
		case 'p': clean = parse_value(optarg, opt.prefix); break;
		case 'o': clean = parse_value(optarg, opt.output); break;
		case 'g': clean = parse_value(optarg, opt.gap); break;
		case 'k': {
			std::istringstream arg(optarg ? optarg : "");
			std::string token;
			while (std::getline(arg, token, ',')) {
				unsigned k = 0;
				std::stringstream ss(token);
				ss >> k;
				opt.klist.push_back(k);
			}
			break;
		}
		case OPT_HELP: usage(std::cerr); return EXIT_SUCCESS;
		case OPT_VERSION:
			std::cerr << PROGRAM << " 1.2.2 \nMI355X (gfx950) engine, C ABI version " << ntc_abi_version()
			          << "; command line compatible with bcgsc/ntCard 1.2.2\n";
			return EXIT_SUCCESS;
		default: clean = false; break; // -l / -f: accepted by getopt, never handled (ntcard.cpp:69,372-375)
		}
		if (optarg != nullptr && !clean) {
			std::cerr << PROGRAM << ": invalid option: `-" << (char)c << optarg << "'\n";
			return EXIT_FAILURE;
		}
	}
	if (argc - optind < 1) {
		std::cerr << PROGRAM << ": missing arguments\n";
		die = true;
	}
	if (opt.gap != 0 && !opt.klist.empty() && opt.gap % 2 != opt.klist[0] % 2) {
		std::cerr << PROGRAM << "Gap size and kmer must have the same modulus\n";
		die = true;
	}
	if (opt.klist.empty()) {
		std::cerr << PROGRAM << ": missing argument -k ... \n";
		die = true;
	}
	if (opt.prefix.empty() && opt.output.empty()) {
		std::cerr << PROGRAM << ": missing argument -p/-o ... \n";
		die = true;
	}
	if (opt.gap != 0 && opt.klist.size() != 1) {
		std::cerr << PROGRAM << ": -g does not support multiple k currently.\n";
		die = true;
	}
	// engine limits the reference does not have (README "Limits"): k <= ntc_max_k() (the closed-form tables of the
	// resolve stage live in LDS), at most NTC_MAX_K_LIST values of k
	for (unsigned k : opt.klist)
		if (k < 1 || k > ntc_max_k()) {
			std::cerr << PROGRAM << ": k=" << k << " is outside the range 1.." << ntc_max_k() << " this GPU engine supports\n";
			die = true;
		}
	if (opt.klist.size() > NTC_MAX_K_LIST) {
		std::cerr << PROGRAM << ": at most " << NTC_MAX_K_LIST << " values of k per run\n";
		die = true;
	}
	if (die) {
		std::cerr << "Try `" << PROGRAM << " --help' for more information.\n";
		return EXIT_FAILURE;
	}

	std::vector<std::string> files;
	for (int i = optind; i < argc; ++i) {
		std::string f(argv[i]);
		if (!f.empty() && f[0] == '@') {
			LineReader list(f.substr(1));
			std::string name;
			while (list.getline(name))
				files.push_back(name);
		} else {
			files.push_back(f);
		}
	}

	// ntcard.cpp:427-431: on-disk size of all inputs decides the sampling rate
	uint64_t total = 0;
	for (const auto& f : files) {
		struct stat st;
		total += stat(f.c_str(), &st) == 0 ? (uint64_t)st.st_size : (uint64_t)-1; // tellg() == -1 on a missing file
	}
	if (total < 50000000000ULL) opt.s_bits = 7;

	ntc_config cfg;
	std::memset(&cfg, 0, sizeof cfg);
	cfg.n_k = (uint32_t)opt.klist.size();
	cfg.k = opt.klist.data();
	cfg.gap = opt.gap;
	cfg.r_bits = opt.r_bits;
	cfg.s_bits = opt.s_bits;
	// Devices: NTCARD_DEVICES="0,1,2,..." spreads the input files over several GPUs (one private sketch each, merged
	// at the end: counting is a commutative sum); NTCARD_DEVICE=<n> or nothing selects a single one.
	std::vector<int> devices;
	if (const char* devs = std::getenv("NTCARD_DEVICES")) {
		std::istringstream ds(devs);
		std::string tok;
		while (std::getline(ds, tok, ','))
			if (!tok.empty()) devices.push_back(std::atoi(tok.c_str()));
	}
	if (devices.empty()) devices.push_back(std::getenv("NTCARD_DEVICE") ? std::atoi(std::getenv("NTCARD_DEVICE")) : 0);
	if (devices.size() > files.size()) devices.resize(std::max<size_t>(1, files.size()));
	std::vector<ntc_engine*> engines(devices.size(), nullptr);
	for (size_t d = 0; d < devices.size(); ++d) {
		cfg.device = devices[d];
		if (ntc_create(&cfg, &engines[d]) != 0) die_engine();
	}
	ntc_engine* eng = engines[0];

	// ntcard.cpp:445-446: `#pragma omp parallel for schedule(dynamic)` over the files
	std::atomic<size_t> next(0);
	auto worker = [&]() {
		for (size_t i; (i = next.fetch_add(1)) < files.size();)
			process_file(files[i], engines[i % engines.size()]);
	};
	std::vector<std::thread> pool;
	const unsigned n_threads = opt.threads == 0 ? 1 : opt.threads;
	// the threads that have no file of their own help the others read and split their blocks (round 6; the reference gives a file one thread, ntcard.cpp:445-446)
	cli::g_file_helpers = std::max<unsigned>(1, n_threads / (unsigned)std::max<size_t>(1, std::min<size_t>(n_threads, files.size())));
	for (unsigned t = 1; t < n_threads && t < files.size(); ++t)
		pool.emplace_back(worker);
	worker();
	for (auto& th : pool)
		th.join();

	const size_t nk = opt.klist.size();
	std::vector<uint32_t> p(nk * 2 * 65536);
	std::vector<uint64_t> f1(nk);
	if (engines.size() > 1) { // one sketch per GPU -> their sum on the first one (16-bit slice exchange over peer copies inside the library)
		if (ntc_merge_devices(engines.data(), (int32_t)engines.size()) != 0) die_engine();
		for (size_t d = 1; d < engines.size(); ++d)
			ntc_destroy(engines[d]);
	}
	if (ntc_finish(eng, nullptr, p.data(), f1.data()) != 0) die_engine();
	std::vector<double> f(opt.cov_max + 1);
	if (opt.output.empty()) { // outDefault, ntcard.cpp:277-298
		for (size_t ki = 0; ki < nk; ++ki) {
			double F0 = 0;
			if (ntc_estimate(&p[ki * 2 * 65536], opt.r_bits, opt.s_bits, opt.cov_max, &F0, f.data()) != 0) die_engine();
			std::ostringstream name;
			name << opt.prefix << "_k" << opt.klist[ki] << ".hist";
			if (ntc_write_hist(name.str().c_str(), f1[ki], F0, f.data(), opt.cov_max) != 0) {
				std::cerr << PROGRAM << ": cannot write " << name.str() << "\n";
				return EXIT_FAILURE;
			}
		}
	} else { // outCompact, ntcard.cpp:300-315
		FILE* out = std::fopen(opt.output.c_str(), "w");
		if (!out) {
			std::cerr << PROGRAM << ": cannot write " << opt.output << "\n";
			return EXIT_FAILURE;
		}
		std::fprintf(out, "k\tf\tn\n");
		for (size_t ki = 0; ki < nk; ++ki) {
			double F0 = 0;
			if (ntc_estimate(&p[ki * 2 * 65536], opt.r_bits, opt.s_bits, opt.cov_max, &F0, f.data()) != 0) die_engine();
			std::cerr << "k=" << opt.klist[ki] << "\tF1\t" << f1[ki] << "\n";
			std::cerr << "k=" << opt.klist[ki] << "\tF0\t" << (uint64_t)F0 << "\n";
			for (unsigned i = 1; i <= opt.cov_max; ++i)
				std::fprintf(out, "%u\t%u\t%llu\n", opt.klist[ki], i, (unsigned long long)(uint64_t)f[i]);
		}
		std::fclose(out);
	}
	ntc_destroy(eng);
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	std::cerr << "Runtime(sec): " << std::setprecision(4) << std::fixed << secs << "\n";
	return 0;
}
