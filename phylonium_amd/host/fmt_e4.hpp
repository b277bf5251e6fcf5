// fmt_e4.hpp — "%.4e" of a double without printf: what just_print of /root/reference/src/io.cxx:141-163 writes for
// every cell of the matrix (std::scientific, precision 4), a million times at N = 1024.
//
// printf rounds the exact binary value to five significant decimal digits.  Here the value is scaled by an exact
// power of ten in 80-bit arithmetic (10^k is exact up to k = 27; the product is off by at most 2^-64 of itself), cut
// to an integer, and rounded on the fraction — unless the fraction lies within 10^-6 of one half, the only place
// where that error (or a tie, which printf rounds to even) could matter: then, as for anything that is not a positive
// finite number in the range the table covers, snprintf answers.  Byte-identical to "%.4e" by construction;
// tests/test_abi_cpu.py compares the two on a few million values.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace phyfmt {

// writes "%.4e" of v (no terminator needed by the caller: the length is returned; buf holds 32 bytes)
inline size_t e4(char *buf, double v)
{
	static const long double P10[28] = {1e0L,  1e1L,  1e2L,  1e3L,  1e4L,  1e5L,  1e6L,  1e7L,  1e8L,  1e9L,  1e10L, 1e11L, 1e12L, 1e13L,
										1e14L, 1e15L, 1e16L, 1e17L, 1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
	if (v == 0.0 && !std::signbit(v)) {
		memcpy(buf, "0.0000e+00", 10);
		return 10;
	}
	if (!(v > 0.0) || !std::isfinite(v) || v < 1e-22 || v >= 1e5) return (size_t)snprintf(buf, 32, "%.4e", v);
	int e2;
	(void)std::frexp(v, &e2);
	int e10 = (int)std::floor((e2 - 1) * 0.30102999566398120); // floor(log10(v)) or one below it
	for (int attempt = 0; attempt < 3; attempt++) {
		const int k = 4 - e10; // v * 10^k in [10^4, 10^5) when e10 is right
		if (k < 0 || k > 27) break;
		const long double scaled = (long double)v * P10[k];
		if (scaled >= 100000.0L) {
			e10++;
			continue;
		}
		if (scaled < 10000.0L) {
			e10--;
			continue;
		}
		uint32_t r = (uint32_t)scaled;
		const long double frac = scaled - (long double)r;
		if (frac > 0.499999L && frac < 0.500001L) break; // too close to call: printf's exact arithmetic decides
		if (frac > 0.5L) r++;
		if (r == 100000u) {
			r = 10000u;
			e10++;
		}
		buf[0] = (char)('0' + r / 10000u);
		buf[1] = '.';
		buf[2] = (char)('0' + r / 1000u % 10u);
		buf[3] = (char)('0' + r / 100u % 10u);
		buf[4] = (char)('0' + r / 10u % 10u);
		buf[5] = (char)('0' + r % 10u);
		buf[6] = 'e';
		const int ae = e10 < 0 ? -e10 : e10;
		buf[7] = e10 < 0 ? '-' : '+';
		buf[8] = (char)('0' + ae / 10);
		buf[9] = (char)('0' + ae % 10);
		return 10; // (|e10| <= 22 here: two digits, as printf prints them)
	}
	return (size_t)snprintf(buf, 32, "%.4e", v);
}

} // namespace phyfmt
