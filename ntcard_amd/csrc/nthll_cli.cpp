// nthll_cli.cpp — drop-in `nthll` front end: HyperLogLog-style estimate of the
// number of distinct canonical k-mers.  Mirrors nthll.cpp: options (:55-68,150-186), `@list`
// (:187-198), one worker per input file (:218-237), the same record splitters as ntcard except that
// any first line that is neither '>' nor '@' is taken as header-less SAM (:71-90), and the single
// result line on stdout (:258).  Hashing, register update and max-merge run on the GPU through
// ntc_hll_create / ntc_submit / ntc_hll_finish; the estimate is ntc_hll_estimate (nthll.cpp:247-254).
#include <getopt.h>

#include <atomic>
#include <thread>

#include "cli_common.hpp"

const char* const cli::kProgram = "nthll";

namespace {

using namespace cli;

void process_file(const std::string& path, ntc_engine* eng)
{
	LineReader in(path);
	std::string first;
	in.getline(first);
	const char c0 = first.empty() ? '\0' : first[0];
	bool sam_has_header = true;
	unsigned type = sniff(first, sam_has_header);
	if (c0 != '>' && c0 != '@') { // nthll.cpp:87-89: no field check, always header-less SAM
		type = 2;
		sam_has_header = false;
	}
	Batcher batch(eng);
	if (type == 0)
		parse_fastq_blocks(in, eng);
	else if (type == 1)
		parse_fasta_blocks(in, eng, batch);
	else
		parse_sam_blocks(in, eng, batch, first, sam_has_header);
	batch.flush();
}

} // namespace

int main(int argc, char** argv)
{
	static const char shortopts[] = "t:k:b:s:hc";
	enum { OPT_HELP = 1, OPT_VERSION };
	static const struct option longopts[] = { { "threads", required_argument, nullptr, 't' },
		                                      { "kmer", required_argument, nullptr, 'k' },
		                                      { "bit", required_argument, nullptr, 'b' },
		                                      { "sit", required_argument, nullptr, 's' },
		                                      { "hash", required_argument, nullptr, 'h' },
		                                      { "help", no_argument, nullptr, OPT_HELP },
		                                      { "version", no_argument, nullptr, OPT_VERSION },
		                                      { nullptr, 0, nullptr, 0 } };
	unsigned threads = 1, k = 64, n_bits = 16, s_unused = 22;
	bool die = false;
	for (int c; (c = getopt_long(argc, argv, shortopts, longopts, nullptr)) != -1;) {
		bool clean = true;
		switch (c) {
		case '?': die = true; break;
		case 't': clean = parse_value(optarg, threads); break;
		case 'b': clean = parse_value(optarg, n_bits); break;
		case 's': clean = parse_value(optarg, s_unused); break;
		case 'k': clean = parse_value(optarg, k); break;
		case 'c': break; // canonical hashing is always on (nthll.cpp:51,167-169)
		case OPT_HELP:
			std::cerr << "Usage: nthll [OPTION]... FILE(S)...\n"
			          << "Estimates the number of distinct k-mers (F0) in FILE(S) on an AMD MI355X.\n\n"
			          << "  -t, --threads=N\tparser threads [1]\n  -k, --kmer=N\tk-mer length [64]\n"
			          << "  -b, --bit=N\tlog2 of the number of registers [16]\n"
			          << "      --help\tdisplay this help and exit\n      --version\toutput version information and exit\n";
			return EXIT_SUCCESS;
		case OPT_VERSION:
			std::cerr << "nthll 1.0.0 \nMI355X (gfx950) engine, C ABI version " << ntc_abi_version() << "\n";
			return EXIT_SUCCESS;
		default: break;
		}
		if (optarg != nullptr && !clean) {
			std::cerr << kProgram << ": invalid option: `-" << (char)c << optarg << "'\n";
			return EXIT_FAILURE;
		}
	}
	if (argc - optind < 1) {
		std::cerr << kProgram << ": missing arguments\n";
		die = true;
	}
	if (die) {
		std::cerr << "Try `" << kProgram << " --help' for more information.\n";
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
	int device = 0;
	if (const char* dev = std::getenv("NTCARD_DEVICE")) device = std::atoi(dev);
	if (k < 1 || k > ntc_max_k()) { // engine limit the reference does not have (README "Limits")
		std::cerr << kProgram << ": k=" << k << " is outside the range 1.." << ntc_max_k() << " this GPU engine supports\n";
		std::exit(EXIT_FAILURE);
	}
	ntc_engine* eng = nullptr;
	if (ntc_hll_create(k, n_bits, device, nullptr, &eng) != 0) die_engine();
	std::atomic<size_t> next(0);
	auto worker = [&]() {
		for (size_t i; (i = next.fetch_add(1)) < files.size();)
			process_file(files[files.size() - i - 1], eng); // nthll.cpp:225-226 walks the list backwards
	};
	std::vector<std::thread> pool;
	cli::g_file_helpers = std::max<unsigned>(1, (threads ? threads : 1) / (unsigned)std::max<size_t>(1, std::min<size_t>(threads ? threads : 1, files.size())));
	for (unsigned t = 1; t < (threads ? threads : 1) && t < files.size(); ++t)
		pool.emplace_back(worker);
	worker();
	for (auto& th : pool)
		th.join();
	std::vector<uint8_t> regs((size_t)1 << n_bits);
	if (ntc_hll_finish(eng, regs.data(), nullptr) != 0) die_engine();
	double est = 0;
	if (ntc_hll_estimate(regs.data(), n_bits, &est) != 0) die_engine();
	ntc_destroy(eng);
	std::cout << "F0, Exp# of distnt kmers(k=" << k << "): " << (unsigned long long)est << "\n";
	return 0;
}
