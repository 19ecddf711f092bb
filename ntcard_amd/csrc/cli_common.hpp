// cli_common.hpp — input side shared by the `ntcard` and `nthll` front ends: plain files or decompressor
// pipes chosen by extension (Common/Uncompress.cpp:32-53), std::getline-compatible line reading, the
// reference's record splitters with their quirks (ntcard.cpp:105-130,173-235 / nthll.cpp:71-148) and
// the batcher that replaces the per-sequence ntRead call with batched ntc_submit calls.
#pragma once
#include <sys/stat.h>

#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <functional>
#include <future>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ntcard_hip.h"

namespace cli {

extern const char* const kProgram;

[[noreturn]] inline void die_engine(const std::string& msg)
{
	std::cerr << kProgram << ": " << msg << "\n";
	std::exit(EXIT_FAILURE);
}
// ntc_last_error() is per thread: call this on the thread whose engine call failed
[[noreturn]] inline void die_engine() { die_engine(ntc_last_error()); }

// ---- input: plain file or a pipe from the decompressor the reference would have used ----------
inline bool ends_with(const std::string& s, const char* suf)
{
	const size_t n = std::strlen(suf);
	return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

inline const char* unpack_command(const std::string& path) // Common/Uncompress.cpp:32-53 (same tools, same order)
{
	if (ends_with(path, ".ar")) return "ar -p";
	if (ends_with(path, ".tar")) return "tar -xOf";
	if (ends_with(path, ".tar.Z") || ends_with(path, ".tar.gz")) return "tar -zxOf";
	if (ends_with(path, ".tar.bz2")) return "tar -jxOf";
	if (ends_with(path, ".tar.xz")) return "tar --use-compress-program=xzdec -xOf";
	if (ends_with(path, ".Z") || ends_with(path, ".gz")) return "gunzip -c";
	if (ends_with(path, ".bz2")) return "bunzip2 -c";
	if (ends_with(path, ".xz")) return "xzdec -c";
	if (ends_with(path, ".zip")) return "unzip -p";
	if (ends_with(path, ".bam")) return "samtools view -h";
	return nullptr;
}

inline std::string shell_quote(const std::string& s)
{
	std::string q = "'";
	for (char c : s) {
		if (c == '\'')
			q += "'\\''";
		else
			q += c;
	}
	return q + "'";
}

class LineReader { // std::getline semantics on a FILE*: false only when nothing could be extracted
public:
	explicit LineReader(const std::string& path)
	{
		struct stat st;
		if (stat(path.c_str(), &st) != 0) return;
		if (const char* cmd = unpack_command(path)) {
			fp_ = popen((std::string(cmd) + " " + shell_quote(path)).c_str(), "r");
			piped_ = true;
		} else {
			fp_ = std::fopen(path.c_str(), "rb");
		}
		if (fp_) buf_.resize(1 << 20);
	}
	~LineReader()
	{
		if (!fp_) return;
		if (piped_) {
			if (pclose(fp_) != 0) { // Common/SignalHandler.cpp:32-52: a failed decompressor is fatal
				std::cerr << kProgram << ": the decompressor of an input file failed\n";
				std::exit(EXIT_FAILURE);
			}
		} else {
			std::fclose(fp_);
		}
	}
	bool ok() const { return fp_ != nullptr; }
	bool getline(std::string& out)
	{
		out.clear();
		if (!fp_) return false;
		bool any = false;
		for (;;) {
			if (pos_ == len_) {
				len_ = std::fread(&buf_[0], 1, buf_.size(), fp_);
				pos_ = 0;
				if (len_ == 0) return any;
			}
			const char* b = &buf_[pos_];
			const char* nl = static_cast<const char*>(std::memchr(b, '\n', len_ - pos_));
			any = true;
			if (nl) {
				out.append(b, nl - b);
				pos_ += (nl - b) + 1;
				return true;
			}
			out.append(b, len_ - pos_);
			pos_ = len_;
		}
	}

	// raw bytes for the block-based splitter: what is left of the line buffer first, then straight from the file
	size_t read_raw(char* dst, size_t cap)
	{
		if (!fp_ || cap == 0) return 0;
		if (pos_ < len_) {
			const size_t n = std::min(cap, len_ - pos_);
			std::memcpy(dst, &buf_[pos_], n);
			pos_ += n;
			return n;
		}
		return std::fread(dst, 1, cap, fp_);
	}

	// the same, filled by several threads when the input is a plain file: what is left of the line buffer first; from then on the file is read with pread at
	// offsets this object keeps itself (run(n, fn) runs fn(0 .. n - 1) on the caller's helper threads).  A pipe (decompressor) is read by the caller alone.
	template <class Pool> size_t read_parallel(char* dst, size_t cap, Pool& pool)
	{
		if (!fp_ || cap == 0) return 0;
		if (pos_ < len_ || piped_) return read_raw(dst, cap);
		if (fd_off_ < 0) { // first call behind the line buffer: the stdio position is where the buffer ended
			fd_off_ = (long long)ftello(fp_);
			struct stat st;
			fd_size_ = fstat(fileno(fp_), &st) == 0 ? (long long)st.st_size : -1;
			if (fd_off_ < 0 || fd_size_ < 0) {
				piped_ = true; // (not seekable after all: plain sequential reads)
				return read_raw(dst, cap);
			}
		}
		const long long left = fd_size_ - fd_off_;
		if (left <= 0) { // the file may have grown since fstat: one more look
			const ssize_t n = pread(fileno(fp_), dst, cap, (off_t)fd_off_);
			if (n > 0) fd_off_ += n;
			return n > 0 ? (size_t)n : 0;
		}
		const size_t want = (size_t)std::min<long long>((long long)cap, left);
		const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(pool.size(), want / (1u << 20) + 1));
		std::vector<size_t> got(parts, 0);
		const int fd = fileno(fp_);
		const long long off0 = fd_off_;
		pool.run(parts, [&](unsigned i) {
			const size_t lo = want * i / parts, hi = want * (i + 1) / parts;
			size_t done = 0;
			while (lo + done < hi) {
				const ssize_t n = pread(fd, dst + lo + done, hi - lo - done, (off_t)(off0 + (long long)(lo + done)));
				if (n <= 0) break;
				done += (size_t)n;
			}
			got[i] = done;
		});
		size_t total = 0;
		for (unsigned i = 0; i < parts; ++i) { // (a short part — the file shrank — ends the block there)
			total += got[i];
			if (got[i] != want * (i + 1) / parts - want * i / parts) break;
		}
		fd_off_ += (long long)total;
		return total;
	}

private:
	FILE* fp_ = nullptr;
	bool piped_ = false;
	std::string buf_;
	size_t pos_ = 0, len_ = 0;
	long long fd_off_ = -1, fd_size_ = -1;
};

// ---- the seam: what replaces ntRead / stRead (ntcard.cpp:147-171) ---------------------------------
class Batcher {
public:
	explicit Batcher(ntc_engine* e) : eng_(e) { offsets_.push_back(0); }
	void add(const std::string& seq)
	{
		bases_ += seq;
		offsets_.push_back(bases_.size());
		if (bases_.size() >= (48u << 20)) flush();
	}
	void flush()
	{
		if (offsets_.size() > 1 && ntc_submit(eng_, bases_.data(), offsets_.data(), offsets_.size() - 1) != 0) die_engine();
		bases_.clear();
		offsets_.assign(1, 0);
	}

private:
	ntc_engine* eng_;
	std::string bases_;
	std::vector<uint64_t> offsets_;
};

inline bool is_number(const std::string& s) // ntcard.cpp:96-103
{
	if (s.empty()) return false;
	for (char c : s)
		if (c < '0' || c > '9') return false;
	return true;
}

// ntcard.cpp:105-130: classify by the first line; 0 fastq, 1 fasta, 2 sam, 3 unknown
inline unsigned sniff(const std::string& first, bool& sam_has_header)
{
	const char c0 = first.empty() ? '\0' : first[0];
	const char c1 = first.size() > 1 ? first[1] : '\0', c2 = first.size() > 2 ? first[2] : '\0';
	if (c0 == '>') return 1;
	if (c0 == '@') {
		const bool tag = (c1 == 'H' && c2 == 'D') || (c1 == 'S' && c2 == 'Q') || (c1 == 'R' && c2 == 'G') ||
		                 (c1 == 'P' && c2 == 'G') || (c1 == 'C' && c2 == 'O');
		return tag ? 2 : 0;
	}
	std::istringstream fields(first);
	std::string f[11];
	for (auto& x : f)
		fields >> x;
	if (is_number(f[1]) && is_number(f[4])) {
		sam_has_header = false;
		return 2;
	}
	return 3;
}

// Block-based record splitters (FASTQ since round 2, FASTA and SAM since round 5): the file is read in 32 MiB blocks, the lines are
// found with memchr, and the sequences are handed to the engine as SPANS of the block (ntc_submit_spans), which copies them once, straight into its
// pinned staging buffer.  A second thread packs block i while this one reads and splits block i + 1.  What a format does with a line is its Handler:
//   line(b, pos, end)  the line [pos, end) of block b: push spans (b.starts / b.lens) or, for a sequence that is not one contiguous run of bytes
//                      (a FASTA record over several lines), copy it to the Batcher
//   keep(pos)          first byte of the block that is still needed when the block is exhausted at `pos` (an unfinished record)
//   rebase(b, keep)    the bytes from `keep` on move to the front of the next block: positions held by the handler shift by -keep
//   finish(b, nl)      end of file; nl: its last line ended in a newline
// Record semantics are the reference's (ntcard.cpp:173-235), quirks included — see each handler.
struct SpanBlock {
	std::vector<char> buf;
	std::vector<uint64_t> starts;
	std::vector<uint32_t> lens;
	std::future<std::string> pending; // error text of the packing task (ntc_last_error is thread-local: the worker has to fetch it), empty = ok
	void add(size_t start, size_t len)
	{
		starts.push_back(start);
		lens.push_back((uint32_t)len);
	}
};

// Round 6: ONE file no longer means one core.  With helper threads (g_file_helpers: the -t threads that have no file of their own — `-t 8` on one file gives it
// eight) a block is READ in parallel (pread of sub-ranges; plain files only) and its newlines are FOUND in parallel (memchr over sub-ranges); the record state
// machine then runs over the list of line ends, which is cheap, while the helpers already fetch the next block, and up to three blocks are being packed
// (ntc_submit_spans) at the same time.  Blocks keep a headroom in front of their data: the unfinished record of block i is copied there when block i has been
// parsed, so fetching block i + 1 does not have to wait for it.
inline unsigned g_file_helpers = 1;

class HelperPool { // parallel_for over a few indices on persistent threads (the caller takes part)
public:
	explicit HelperPool(unsigned n) : n_(n > 1 ? n : 1)
	{
		for (unsigned t = 1; t < n_; ++t)
			th_.emplace_back([this] { loop(); });
	}
	~HelperPool()
	{
		{
			std::lock_guard<std::mutex> lk(mu_);
			quit_ = true;
		}
		cv_.notify_all();
		for (auto& t : th_)
			t.join();
	}
	unsigned size() const { return n_; }
	void run(unsigned n_tasks, const std::function<void(unsigned)>& fn)
	{
		if (n_ == 1 || n_tasks <= 1) {
			for (unsigned i = 0; i < n_tasks; ++i)
				fn(i);
			return;
		}
		{
			std::lock_guard<std::mutex> lk(mu_);
			fn_ = &fn;
			total_ = n_tasks;
			next_ = 0;
			left_ = n_tasks;
			++gen_;
		}
		cv_.notify_all();
		work();
		std::unique_lock<std::mutex> lk(mu_);
		done_.wait(lk, [this] { return left_ == 0; });
		fn_ = nullptr;
	}

private:
	void work()
	{
		for (;;) {
			unsigned i;
			const std::function<void(unsigned)>* f;
			{
				std::lock_guard<std::mutex> lk(mu_);
				if (!fn_ || next_ >= total_) return;
				i = next_++;
				f = fn_;
			}
			(*f)(i);
			std::lock_guard<std::mutex> lk(mu_);
			if (--left_ == 0) done_.notify_all();
		}
	}
	void loop()
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
				if (quit_) return;
				seen = gen_;
			}
			work();
		}
	}
	unsigned n_;
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::condition_variable cv_, done_;
	const std::function<void(unsigned)>* fn_ = nullptr;
	unsigned total_ = 0, next_ = 0, left_ = 0;
	uint64_t gen_ = 0;
	bool quit_ = false;
};

template <class Handler> inline void parse_blocks(LineReader& in, ntc_engine* eng, Handler& h)
{
	size_t kBlock = 32u << 20;
	if (const char* t = std::getenv("NTC_CLI_BLOCK_BYTES")) // test knob: tiny blocks put every record across a block boundary
		kBlock = std::max<size_t>(64, std::strtoull(t, nullptr, 10));
	const size_t kHead = std::min<size_t>(1u << 20, kBlock); // headroom in front of a block's data for the previous block's unfinished record (grown on demand)
	HelperPool pool(g_file_helpers);
	constexpr int kRing = 4; // blocks: one being parsed, one being fetched, up to two more being packed (the engine has four staging pairs)
	struct Block : SpanBlock {
		size_t head = 0;             // buf[head ..) is where a fetch puts its bytes
		size_t begin = 0, end = 0;   // the block's bytes: buf[begin .. end) (begin <= head: the carried-over tail sits in [begin, head))
		std::vector<size_t> nl;      // positions of the newlines in [head, end), ascending
		bool eof = false;            // the fetch hit the end of the file
	};
	Block blk[kRing];
	auto wait = [&](SpanBlock& b) {
		if (!b.pending.valid()) return;
		const std::string err = b.pending.get();
		if (!err.empty()) die_engine(err);
	};
	// newlines of buf[from, to) appended to b.nl (parallel over sub-ranges)
	auto scan = [&](Block& b, size_t from, size_t to) {
		const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(pool.size(), (to - from) / (256u << 10) + 1));
		std::vector<std::vector<size_t>> found(parts);
		pool.run(parts, [&](unsigned i) {
			const size_t lo = from + (to - from) * i / parts, hi = from + (to - from) * (i + 1) / parts;
			const char* base = b.buf.data();
			auto& v = found[i];
			v.reserve((hi - lo) / 64 + 16);
			for (const char* p = base + lo; p < base + hi;) {
				const char* q = static_cast<const char*>(std::memchr(p, '\n', (size_t)(base + hi - p)));
				if (!q) break;
				v.push_back((size_t)(q - base));
				p = q + 1;
			}
		});
		for (auto& v : found)
			b.nl.insert(b.nl.end(), v.begin(), v.end());
	};
	// up to `want` more bytes behind b.end (parallel pread for a plain file once the line buffer is drained), then their newlines
	auto fetch_more = [&](Block& b, size_t want) {
		if (b.buf.size() < b.end + want) b.buf.resize(b.end + want);
		const size_t from = b.end;
		size_t got = 0;
		while (got < want && !b.eof) {
			const size_t n = in.read_parallel(b.buf.data() + from + got, want - got, pool);
			if (n == 0) b.eof = true;
			got += n;
		}
		b.end = from + got;
		scan(b, from, b.end);
	};
	auto fetch = [&](Block& b) { // a fresh block
		wait(b); // its previous contents have been packed
		b.starts.clear();
		b.lens.clear();
		b.nl.clear();
		b.eof = false;
		if (b.head < kHead) b.head = kHead;
		if (b.buf.size() < b.head + kBlock) b.buf.resize(b.head + kBlock);
		b.begin = b.end = b.head;
		fetch_more(b, kBlock);
	};
	int cur = 0;
	fetch(blk[0]);
	size_t pos = blk[0].begin; // next unparsed byte of the current block
	bool last_nl = true;
	for (;;) {
		Block& b = blk[cur];
		Block& nx = blk[(cur + 1) % kRing];
		std::future<void> pre;
		if (!b.eof) pre = std::async(std::launch::async, [&] { fetch(nx); }); // (runs the pool: the coordinator does not use it meanwhile)
		// a line longer than everything fetched so far (a chromosome on one FASTA line): read on into THIS block, doubling
		if (b.nl.empty() && !b.eof) {
			if (pre.valid()) pre.get(); // (the next block has consumed file bytes that belong to this line: fold them in)
			for (;;) {
				Block& n2 = blk[(cur + 1) % kRing];
				const size_t add = n2.end - n2.head;
				if (b.buf.size() < b.end + add) b.buf.resize(std::max(b.buf.size() * 2, b.end + add));
				std::memcpy(b.buf.data() + b.end, n2.buf.data() + n2.head, add);
				for (size_t x : n2.nl)
					b.nl.push_back(x - n2.head + b.end);
				b.end += add;
				b.eof = n2.eof;
				if (!b.nl.empty() || b.eof) break;
				fetch(n2);
			}
			if (!b.eof) pre = std::async(std::launch::async, [&] { fetch(nx); });
		}
		for (size_t x : b.nl) { // the record state machine over the line ends
			h.line(b, pos, x);
			pos = x + 1;
		}
		if (b.eof) {
			if (pos < b.end) { // last line of the file, no newline
				h.line(b, pos, b.end);
				pos = b.end;
				last_nl = false;
			}
			h.finish(b, last_nl);
			if (!b.starts.empty() && ntc_submit_spans(eng, b.buf.data(), b.starts.data(), b.lens.data(), b.starts.size()) != 0) die_engine();
			break;
		}
		pre.get();
		// the block is exhausted: everything from the first byte still needed moves into the headroom of the next block
		const size_t keep = h.keep(pos);
		const size_t tail = b.end - keep;
		if (tail > nx.head) { // (rare: an unfinished record longer than the headroom — make room in front of the fetched bytes)
			const size_t grow = ((tail - nx.head) + (1u << 20)) & ~(size_t)0xfffff;
			std::vector<char> nb(nx.buf.size() + grow);
			std::memcpy(nb.data() + nx.head + grow, nx.buf.data() + nx.head, nx.end - nx.head);
			nx.buf.swap(nb);
			for (auto& x : nx.nl)
				x += grow;
			nx.head += grow;
			nx.end += grow;
		}
		h.rebase(b, keep - (nx.head - tail)); // positions held by the handler: byte `keep` of this block becomes byte nx.head - tail of the next
		std::memcpy(nx.buf.data() + nx.head - tail, b.buf.data() + keep, tail);
		nx.begin = nx.head - tail;
		if (!b.starts.empty()) {
			SpanBlock* pb = &b;
			b.pending = std::async(std::launch::async, [eng, pb]() -> std::string {
				if (ntc_submit_spans(eng, pb->buf.data(), pb->starts.data(), pb->lens.data(), pb->starts.size()) == 0) return std::string();
				const std::string msg = ntc_last_error();
				return msg.empty() ? std::string("ntc_submit_spans failed") : msg;
			});
		}
		pos = nx.begin + (pos - keep);
		cur = (cur + 1) % kRing;
	}
	for (auto& b : blk)
		wait(b);
}

// FASTQ, ntcard.cpp:173-189 (four-line records, the first header already consumed by the sniffer): a record counts once its quality line could be
// read (with or without a final newline); CR bytes and lower case stay in the sequence
struct FastqLines {
	int phase = 0; // line expected next: 0 sequence, 1 '+', 2 quality, 3 header of the next record
	size_t s_start = 0, s_len = 0;
	void line(SpanBlock& b, size_t pos, size_t end)
	{
		if (phase == 0) {
			s_start = pos;
			s_len = end - pos;
		} else if (phase == 2) { // the quality line could be read: the record counts
			b.add(s_start, s_len);
		}
		phase = (phase + 1) & 3;
	}
	size_t keep(size_t pos) const { return (phase == 1 || phase == 2) ? s_start : pos; }
	void rebase(SpanBlock&, size_t keep) { s_start -= keep; } // (only meaningful while a sequence line is pending)
	void finish(SpanBlock&, bool) {}
};
inline void parse_fastq_blocks(LineReader& in, ntc_engine* eng)
{
	FastqLines h;
	parse_blocks(in, eng, h);
}

// FASTA, ntcard.cpp:191-208 (the first header already consumed): a sequence is a maximal run of lines that do not start with '>', concatenated
// (k-mers span the line breaks, CR bytes stay); a record of ONE line — reads — is a span of the block, a wrapped record is copied to the Batcher
struct FastaLines {
	Batcher& out;
	bool one = false, many = false; // the current record has one line (a span so far) / several (in `multi`)
	size_t p_start = 0, p_len = 0;
	std::string multi;
	explicit FastaLines(Batcher& o) : out(o) {}
	void flush(SpanBlock& b)
	{
		if (many) out.add(multi);
		else if (one) b.add(p_start, p_len);
		one = many = false;
	}
	void line(SpanBlock& b, size_t pos, size_t end)
	{
		const char* base = b.buf.data();
		if (end > pos && base[pos] == '>') {
			flush(b);
			return;
		}
		if (many) {
			multi.append(base + pos, end - pos);
		} else if (one) {
			multi.assign(base + p_start, p_len);
			multi.append(base + pos, end - pos);
			one = false;
			many = true;
		} else {
			one = true;
			p_start = pos;
			p_len = end - pos;
		}
	}
	size_t keep(size_t pos) const { return one ? p_start : pos; }
	void rebase(SpanBlock&, size_t keep) { p_start -= keep; }
	void finish(SpanBlock& b, bool) { flush(b); }
};
inline void parse_fasta_blocks(LineReader& in, ntc_engine* eng, Batcher& out)
{
	FastaLines h(out);
	parse_blocks(in, eng, h);
}

// SAM, ntcard.cpp:210-235: the header lines ('@...') are skipped up to the first line that is empty or does not start with '@'; from then on EVERY
// line is a record whose sequence is its 10th whitespace-separated field — and a line with fewer fields (an empty one included) leaves the previous
// sequence in place, which is then counted AGAIN, exactly like the reference's `fields >> seq`.  (Should the file end inside the header without a final
// newline, the reference treats the last header line as a record: so does this.)
struct SamLines {
	Batcher& out;
	bool in_header;
	bool prev_in_block = false;
	size_t pv_start = 0, pv_len = 0;
	std::string prev_copy, last_header;
	SamLines(Batcher& o, bool has_header) : out(o), in_header(has_header) {}
	static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }
	// the 10th field of [p, e): -> (start, len) or false
	static bool field10(const char* p, const char* e, const char*& fs, size_t& fl)
	{
		for (int f = 0; f < 10; ++f) {
			while (p < e && is_space(*p)) ++p;
			if (p == e) return false;
			const char* t = p;
			while (p < e && !is_space(*p)) ++p;
			if (f == 9) {
				fs = t;
				fl = (size_t)(p - t);
				return true;
			}
		}
		return false;
	}
	void record(SpanBlock& b, size_t pos, size_t end)
	{
		const char* base = b.buf.data();
		const char* fs;
		size_t fl;
		if (field10(base + pos, base + end, fs, fl)) {
			pv_start = (size_t)(fs - base);
			pv_len = fl;
			prev_in_block = true;
			b.add(pv_start, pv_len);
		} else if (prev_in_block) {
			b.add(pv_start, pv_len); // the previous sequence, once more
		} else if (!prev_copy.empty()) {
			out.add(prev_copy);
		}
	}
	void first_line(const std::string& first) // a file without header: the sniffer has consumed its first record
	{
		const char* fs;
		size_t fl;
		if (field10(first.data(), first.data() + first.size(), fs, fl)) {
			prev_copy.assign(fs, fl);
			out.add(prev_copy);
		}
	}
	void line(SpanBlock& b, size_t pos, size_t end)
	{
		if (in_header) {
			if (end > pos && b.buf[pos] == '@') {
				last_header.assign(b.buf.data() + pos, end - pos);
				return;
			}
			in_header = false;
		}
		record(b, pos, end);
	}
	size_t keep(size_t pos) const { return pos; }
	void rebase(SpanBlock& b, size_t)
	{
		if (prev_in_block) prev_copy.assign(b.buf.data() + pv_start, pv_len); // the block goes away: the previous sequence is kept as a copy
		prev_in_block = false;
	}
	void finish(SpanBlock&, bool last_nl)
	{
		// the file ended inside the header: the reference's do-loop (ntcard.cpp:225-233) runs once on whatever the failed getline left in samLine —
		// nothing after a final newline, the last header line if the file ends without one
		if (in_header && !last_nl) first_line(last_header);
		in_header = false;
	}
};
inline void parse_sam_blocks(LineReader& in, ntc_engine* eng, Batcher& out, const std::string& first, bool has_header)
{
	SamLines h(out, has_header);
	if (!has_header) h.first_line(first);
	parse_blocks(in, eng, h);
}

template <typename T>
bool parse_value(const char* text, T& out) // `arg >> value` followed by the reference's `!arg.eof()` check
{
	std::istringstream arg(text ? text : "");
	arg >> out;
	return arg.eof();
}

} // namespace cli
