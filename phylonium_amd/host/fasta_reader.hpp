// fasta_reader.hpp — FASTA files → genomes as the path wants them: nucleotides
// filtered to ACGT (upper-cased), contigs joined by '!'.
//
// Mirrors /root/reference/src/io.cxx:36-59 (genome names) and
// src/sequence.cxx:109-199 (filter_nucl, join) over libs/pfasta.c's record
// reader.  Shared by the host driver (phylonium_amd_cli.cpp) and the library's
// host helper phylo_host_read_fasta (phylo_abi.hip).
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

namespace phyfasta {

struct Genome {
	std::string name; // file name without directory and .fa/.fas/.fasta (io.cxx:36-59)
	std::string nucl; // contigs joined by '!' (sequence.cxx:171-199)
};

inline std::string genome_name(const std::string &path)
{
	size_t left = path.rfind('/');
	left = (left == std::string::npos) ? 0 : left + 1;
	size_t right = path.rfind('.');
	if (right != std::string::npos) {
		std::string ext = path.substr(right);
		if (!(ext == ".fa" || ext == ".fas" || ext == ".fasta")) right = path.size();
	} else {
		right = path.size();
	}
	return path.substr(left, right - left);
}

// FASTA records of one file → nucleotides filtered to ACGT (upper-cased,
// sequence.cxx:109-146), contigs joined by '!'.  The file is read in one piece
// and filtered through a 256-entry table; errors are returned, not raised, so
// that files can be read by several threads and the first bad one *in command
// line order* is still the one reported (the reference reads them in order).
struct ReadResult {
	Genome g;
	std::string error;
};

inline ReadResult read_genome(const std::string &path)
{
	ReadResult R;
	R.g.name = genome_name(path);
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) {
		R.error = path + ": " + strerror(errno);
		return R;
	}
	std::string buf;
	struct stat st;
	if (fstat(fileno(f), &st) == 0 && st.st_size > 0) buf.reserve((size_t)st.st_size);
	char chunk[1 << 16];
	size_t got;
	while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, got);
	fclose(f);

	static const auto lut = [] {
		std::array<char, 256> t{};
		t[(unsigned char)'A'] = t[(unsigned char)'a'] = 'A';
		t[(unsigned char)'C'] = t[(unsigned char)'c'] = 'C';
		t[(unsigned char)'G'] = t[(unsigned char)'g'] = 'G';
		t[(unsigned char)'T'] = t[(unsigned char)'t'] = 'T';
		return t;
	}();
	std::string &out = R.g.nucl;
	out.resize(buf.size() + 1);
	size_t w = 0;
	bool in_record = false;
	size_t records = 0;
	const char *p = buf.data(), *end = p + buf.size();
	while (p < end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end;
		if (le > p && *p == '>') {
			// a record starts; every record after the first is preceded by the separator
			if (records++) out[w++] = '!';
			in_record = true;
		} else if (!in_record) {
			for (const char *c = p; c < le; c++)
				if (!isspace((unsigned char)*c)) {
					R.error = path + ": File is not in FASTA format.";
					return R;
				}
		} else {
			for (const char *c = p; c < le; c++) {
				const char v = lut[(unsigned char)*c];
				out[w] = v;
				w += v != 0;
			}
		}
		p = nl ? nl + 1 : end;
	}
	out.resize(w);
	if (!records) R.error = path + ": Empty file.";
	return R;
}

// All files, in parallel over up to `threads` host threads.  *error receives the message of
// the first file (in command line order) that could not be read.
inline std::vector<Genome> read_genomes(const std::vector<std::string> &files, size_t threads, std::string *error)
{
	std::vector<ReadResult> res(files.size());
	std::atomic<size_t> next{0};
	auto work = [&] {
		for (;;) {
			size_t i = next.fetch_add(1);
			if (i >= files.size()) break;
			res[i] = read_genome(files[i]);
		}
	};
	threads = std::max<size_t>(1, std::min(threads, files.size()));
	std::vector<std::thread> pool;
	for (size_t t = 1; t < threads; t++) pool.emplace_back(work);
	work();
	for (auto &t : pool) t.join();
	std::vector<Genome> q(files.size());
	for (size_t i = 0; i < files.size(); i++) {
		if (!res[i].error.empty() && error->empty()) *error = res[i].error;
		q[i] = std::move(res[i].g);
	}
	return q;
}

} // namespace phyfasta
