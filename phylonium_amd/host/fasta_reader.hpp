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
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

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

// A file's bytes: mapped when it can be (no copy through a read buffer), read otherwise (pipes)
struct FileBytes {
	const char *data = nullptr;
	size_t size = 0;
	std::string error; // strerror text when the file could not be opened
	FileBytes(const std::string &path)
	{
		const int fd = open(path.c_str(), O_RDONLY);
		if (fd < 0) {
			error = strerror(errno);
			return;
		}
		struct stat st;
		if (fstat(fd, &st) != 0 || S_ISDIR(st.st_mode)) {
			error = strerror(S_ISDIR(st.st_mode) ? EISDIR : errno);
			close(fd);
			return;
		}
		if (st.st_size > 0) {
			map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
			if (map != MAP_FAILED) {
				madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
				data = (const char *)map;
				size = map_size = (size_t)st.st_size;
			}
		}
		if (!data) {
			char chunk[1 << 16];
			ssize_t got;
			while ((got = read(fd, chunk, sizeof chunk)) > 0) slurp.append(chunk, (size_t)got);
			data = slurp.data();
			size = slurp.size();
		}
		close(fd);
	}
	~FileBytes()
	{
		if (map != MAP_FAILED) munmap(map, map_size);
	}
	FileBytes(const FileBytes &) = delete;
	FileBytes &operator=(const FileBytes &) = delete;

  private:
	void *map = MAP_FAILED;
	size_t map_size = 0;
	std::string slurp;
};

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
	FileBytes buf(path);
	if (!buf.error.empty()) {
		R.error = path + ": " + buf.error;
		return R;
	}

	static const auto lut = [] {
		std::array<char, 256> t{};
		t[(unsigned char)'A'] = t[(unsigned char)'a'] = 'A';
		t[(unsigned char)'C'] = t[(unsigned char)'c'] = 'C';
		t[(unsigned char)'G'] = t[(unsigned char)'g'] = 'G';
		t[(unsigned char)'T'] = t[(unsigned char)'t'] = 'T';
		return t;
	}();
	std::string &out = R.g.nucl;
	out.resize(buf.size + 1);
	size_t w = 0;
	bool in_record = false;
	size_t records = 0;
	const char *p = buf.data, *end = p + buf.size;
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

// ───────────── the same genome as 2-bit codes, straight from the mapped file ─────────────
//
// What the device wants (lean_core.h: Q2 / QBAD): 16 bases per 32-bit word, first base in bits
// 31..30, A0 C1 G2 T3; the only other byte a filtered genome can hold is the '!' between two
// records, kept as code 0 plus its position in a sorted list.  The file is mapped, not copied;
// a line of nucleotides goes through 32 bytes at a time (AVX2 + BMI2 where the CPU has them:
// upper-case, test for ACGT, two movemasks and two bit deposits per 32 bases) and anything else
// — lower-case runs are fine, IUPAC codes, digits, '\r' — byte by byte through the same table
// as read_genome.  A quarter of the bytes leave for the device (PCIe) and none are written
// twice on the host.
struct PackedGenome {
	std::string name;
	uint32_t *q2 = nullptr; // (len + 15) / 16 words, codes past `len` are 0; the caller's buffer, or malloc when `own`
	bool own = false;
	uint64_t len = 0; // nucleotides + separators
	std::vector<uint32_t> bad; // positions of the '!' separators, ascending
};
struct PackedResult {
	PackedGenome g;
	std::string error;
};

struct Packer {
	uint32_t *out;
	size_t w = 0; // words written
	uint64_t acc = 0; // pending bases (fewer than 16) in the top bits
	unsigned nb = 0;
	// r <= 32 bases, first base in bits 63..62 of v, bits behind the r-th base zero
	inline void append(uint64_t v, unsigned r)
	{
		uint64_t hi = acc | (nb ? v >> (2 * nb) : v);
		const uint64_t lo = nb ? v << (64 - 2 * nb) : 0;
		unsigned tot = nb + r;
		if (tot >= 32) {
			out[w++] = (uint32_t)(hi >> 32);
			out[w++] = (uint32_t)hi;
			hi = lo;
			tot -= 32;
		}
		if (tot >= 16) {
			out[w++] = (uint32_t)(hi >> 32);
			hi <<= 32;
			tot -= 16;
		}
		acc = hi;
		nb = tot;
	}
	inline void push(uint32_t code) { append((uint64_t)code << 62, 1); }
	inline uint64_t bases() const { return 16 * (uint64_t)w + nb; }
	inline void finish()
	{
		if (nb) out[w++] = (uint32_t)(acc >> 32);
		acc = 0;
		nb = 0;
	}
};

// code + 1 of the nucleotides (either case), 0 for every other byte
inline const std::array<uint8_t, 256> &code_lut()
{
	static const auto lut = [] {
		std::array<uint8_t, 256> t{};
		t[(unsigned char)'A'] = t[(unsigned char)'a'] = 1;
		t[(unsigned char)'C'] = t[(unsigned char)'c'] = 2;
		t[(unsigned char)'G'] = t[(unsigned char)'g'] = 3;
		t[(unsigned char)'T'] = t[(unsigned char)'t'] = 4;
		return t;
	}();
	return lut;
}

inline void pack_bytes_scalar(Packer &P, const char *p, const char *e)
{
	const auto &lut = code_lut();
	for (; p < e; p++) {
		const uint8_t v = lut[(unsigned char)*p];
		if (v) P.push(v - 1u);
	}
}

#if defined(__x86_64__)
// [p, e) with at least 32 readable bytes behind every position below e - (e - p) % 32 … the caller
// passes `safe_end` = last address up to which a 32-byte load may start
__attribute__((target("avx2,bmi2"))) inline void pack_bytes_avx2(Packer &P, const char *p, const char *e, const char *safe_end)
{
	const __m256i up = _mm256_set1_epi8((char)0xdf);
	const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
	// byte order reversed, so that movemask puts the first base into the top bit
	const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4,
										 3, 2, 1, 0);
	while (p < e) {
		const size_t left = (size_t)(e - p);
		const unsigned r = left >= 32 ? 32u : (unsigned)left;
		if (p > safe_end) break;
		__m256i x = _mm256_loadu_si256((const __m256i *)p);
		x = _mm256_and_si256(x, up);
		const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, cA), _mm256_cmpeq_epi8(x, cC)),
										   _mm256_or_si256(_mm256_cmpeq_epi8(x, cG), _mm256_cmpeq_epi8(x, cT)));
		const uint32_t okm = (uint32_t)_mm256_movemask_epi8(ok);
		const uint32_t want = r == 32 ? 0xffffffffu : ((1u << r) - 1u);
		if ((okm & want) != want) {
			// something else in here: this piece byte by byte
			pack_bytes_scalar(P, p, p + r);
			p += r;
			continue;
		}
		// code = ((b >> 1) & 3) ^ ((b >> 2) & 1): A0 C1 G2 T3
		const __m256i b1 = _mm256_srli_epi16(x, 1), b2 = _mm256_srli_epi16(x, 2);
		const __m256i code = _mm256_xor_si256(_mm256_and_si256(b1, _mm256_set1_epi8(3)), _mm256_and_si256(b2, _mm256_set1_epi8(1)));
		__m256i c = _mm256_shuffle_epi8(code, rev);
		c = _mm256_permute2x128_si256(c, c, 1);
		const uint32_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 7));
		const uint32_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 6));
		uint64_t v = _pdep_u64(m0, 0x5555555555555555ull) | _pdep_u64(m1, 0xaaaaaaaaaaaaaaaaull);
		if (r < 32) v &= ~0ull << (64 - 2 * r);
		P.append(v, r);
		p += r;
	}
	if (p < e) pack_bytes_scalar(P, p, e);
}
#endif

inline bool have_avx2_bmi2()
{
#if defined(__x86_64__)
	static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")
#ifdef PHY_DEV_HOOKS
							&& !getenv("PHYLONIUM_AMD_NO_SIMD") // the byte-wise loop only (tests compare the two)
#endif
		;
	return yes;
#else
	return false;
#endif
}

// Words of output a file of `size` bytes can need (one base per byte at most, and the tail word)
inline size_t packed_words_bound(size_t size) { return size / 16 + 4; }

// simd: -1 = what the CPU has, 0 = the byte-wise loop only (tests compare the two).  `dst` (room for
// packed_words_bound(file size) words), when given, receives the codes; a file that turns out larger than
// `dst_words` allows (it grew, or is not a regular file) gets a buffer of its own (g.own).
inline PackedResult read_genome_packed(const std::string &path, int simd = -1, uint32_t *dst = nullptr, size_t dst_words = 0)
{
	PackedResult R;
	R.g.name = genome_name(path);
	FileBytes file(path);
	if (!file.error.empty()) {
		R.error = path + ": " + file.error;
		return R;
	}
	const char *base = file.data;
	const size_t n = file.size;

	uint32_t *out = dst;
	const bool own = !dst || packed_words_bound(n) > dst_words;
	if (own) out = (uint32_t *)malloc(packed_words_bound(n) * sizeof(uint32_t));
	if (!out) {
		R.error = path + ": out of memory";
		return R;
	}
	Packer P{out};
	const bool vec = simd != 0 && have_avx2_bmi2();
	bool in_record = false;
	size_t records = 0;
	const char *p = base, *end = base + n;
	const char *safe_end = n >= 32 ? end - 32 : base - 1;
	while (p < end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end;
		if (le > p && *p == '>') {
			if (records++) {
				R.g.bad.push_back((uint32_t)P.bases());
				P.push(0);
			}
			in_record = true;
		} else if (!in_record) {
			for (const char *c = p; c < le; c++)
				if (!isspace((unsigned char)*c)) {
					R.error = path + ": File is not in FASTA format.";
					if (own) free(out);
					return R;
				}
		} else {
#if defined(__x86_64__)
			if (vec) pack_bytes_avx2(P, p, le, safe_end);
			else
#endif
				pack_bytes_scalar(P, p, le);
		}
		p = nl ? nl + 1 : end;
	}
	(void)vec;
	R.g.len = P.bases();
	P.finish();
	if (!records) {
		R.error = path + ": Empty file.";
		if (own) free(out);
		return R;
	}
	R.g.q2 = out;
	R.g.own = own;
	return R;
}

// the bytes back (reference genome for the suffix array, -p): "ACGT"[code], '!' at the listed positions
inline std::string unpack_genome(const PackedGenome &g)
{
	static const auto quad = [] { // a byte of four codes → its four letters, first base in the top bits
		std::array<std::array<char, 4>, 256> t{};
		for (unsigned b = 0; b < 256; b++)
			for (unsigned i = 0; i < 4; i++) t[b][i] = "ACGT"[(b >> (6 - 2 * i)) & 3u];
		return t;
	}();
	const size_t words = (size_t)((g.len + 15) / 16);
	std::string s(words * 16, 'A');
	for (size_t w = 0; w < words; w++) {
		const uint32_t c = g.q2[w];
		for (unsigned k = 0; k < 4; k++) memcpy(&s[16 * w + 4 * k], quad[(c >> (24 - 8 * k)) & 0xffu].data(), 4);
	}
	s.resize((size_t)g.len);
	for (uint32_t b : g.bad) s[b] = '!';
	return s;
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

// All files as packed genomes inside ONE allocation (*arena, released with free() once the genomes are on the
// device): a thousand separate buffers cost a thousand munmaps, each of which the GPU driver's MMU notifier
// sees after the buffer was pinned for a copy — 0.3 s at 1024 genomes.
inline std::vector<PackedGenome> read_genomes_packed(const std::vector<std::string> &files, size_t threads, std::string *error,
													 uint32_t **arena, size_t *error_index = nullptr)
{
	std::vector<size_t> woff(files.size() + 1, 0);
	for (size_t i = 0; i < files.size(); i++) {
		struct stat st;
		const size_t size = (stat(files[i].c_str(), &st) == 0 && S_ISREG(st.st_mode)) ? (size_t)st.st_size : 0;
		woff[i + 1] = woff[i] + (packed_words_bound(size) + 15) / 16 * 16; // 64-byte aligned starts
	}
	*arena = (uint32_t *)malloc(std::max<size_t>(woff.back(), 16) * sizeof(uint32_t));
	std::vector<PackedResult> res(files.size());
	std::atomic<size_t> next{0};
	auto work = [&] {
		for (;;) {
			size_t i = next.fetch_add(1);
			if (i >= files.size()) break;
			res[i] = read_genome_packed(files[i], -1, *arena ? *arena + woff[i] : nullptr, *arena ? woff[i + 1] - woff[i] : 0);
		}
	};
	threads = std::max<size_t>(1, std::min(threads, files.size()));
	std::vector<std::thread> pool;
	for (size_t t = 1; t < threads; t++) pool.emplace_back(work);
	work();
	for (auto &t : pool) t.join();
	std::vector<PackedGenome> q(files.size());
	for (size_t i = 0; i < files.size(); i++) {
		if (!res[i].error.empty() && error->empty()) {
			*error = res[i].error;
			if (error_index) *error_index = i;
		}
		q[i] = std::move(res[i].g);
	}
	return q;
}

} // namespace phyfasta
