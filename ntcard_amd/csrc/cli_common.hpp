// cli_common.hpp — input side shared by the `ntcard` and `nthll` front ends: plain files or decompressor
// pipes chosen by extension (Common/Uncompress.cpp:32-53), std::getline-compatible line reading, the
// reference's record splitters with their quirks (ntcard.cpp:105-130,173-235 / nthll.cpp:71-148) and
// the batcher that replaces the per-sequence ntRead call with batched ntc_submit calls.
#pragma once
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <future>
#include <sstream>
#include <string>
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

private:
	FILE* fp_ = nullptr;
	bool piped_ = false;
	std::string buf_;
	size_t pos_ = 0, len_ = 0;
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

template <class Handler> inline void parse_blocks(LineReader& in, ntc_engine* eng, Handler& h)
{
	size_t kBlock = 32u << 20;
	if (const char* t = std::getenv("NTC_CLI_BLOCK_BYTES")) // test knob: tiny blocks put every record across a block boundary
		kBlock = std::max<size_t>(64, std::strtoull(t, nullptr, 10));
	SpanBlock blk[2];
	int cur = 0;
	blk[0].buf.resize(kBlock);
	blk[1].buf.resize(kBlock);
	size_t have = 0, pos = 0;  // bytes in the current block, next unparsed byte
	bool eof = false, last_nl = true;
	auto wait = [&](SpanBlock& b) {
		if (!b.pending.valid()) return;
		const std::string err = b.pending.get();
		if (!err.empty()) die_engine(err);
	};
	for (;;) {
		SpanBlock& b = blk[cur];
		if (!eof) {
			const size_t n = in.read_raw(b.buf.data() + have, b.buf.size() - have);
			if (n == 0) eof = true;
			have += n;
		}
		for (;;) { // split the lines of [pos, have)
			const char* base = b.buf.data();
			const char* nl = static_cast<const char*>(std::memchr(base + pos, '\n', have - pos));
			size_t line_end, next;
			if (nl) {
				line_end = (size_t)(nl - base);
				next = line_end + 1;
			} else if (eof && pos < have) { // last line of the file, no newline
				line_end = next = have;
				last_nl = false;
			} else {
				break;
			}
			h.line(b, pos, line_end);
			pos = next;
		}
		if (eof && pos >= have) {
			h.finish(b, last_nl);
			if (!b.starts.empty() && ntc_submit_spans(eng, b.buf.data(), b.starts.data(), b.lens.data(), b.starts.size()) != 0) die_engine();
			break;
		}
		// the block is exhausted: everything from the first byte still needed moves to the front of the other block
		const size_t keep = h.keep(pos);
		SpanBlock& o = blk[cur ^ 1];
		wait(o); // its previous contents have been packed
		if (keep == 0 && have == b.buf.size()) { // one record longer than the block: grow and read on
			b.buf.resize(b.buf.size() * 2);
			continue;
		}
		const size_t tail = have - keep;
		if (o.buf.size() < b.buf.size()) o.buf.resize(b.buf.size());
		h.rebase(b, keep);
		std::memcpy(o.buf.data(), b.buf.data() + keep, tail);
		o.starts.clear();
		o.lens.clear();
		if (!b.starts.empty()) {
			SpanBlock* pb = &b;
			b.pending = std::async(std::launch::async, [eng, pb]() -> std::string {
				if (ntc_submit_spans(eng, pb->buf.data(), pb->starts.data(), pb->lens.data(), pb->starts.size()) == 0) return std::string();
				const std::string msg = ntc_last_error();
				return msg.empty() ? std::string("ntc_submit_spans failed") : msg;
			});
		}
		pos -= keep;
		have = tail;
		cur ^= 1;
	}
	wait(blk[0]);
	wait(blk[1]);
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
