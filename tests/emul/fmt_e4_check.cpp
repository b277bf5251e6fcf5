// fmt_e4_check — phyfmt::e4 (phylonium_amd/host/fmt_e4.hpp) against snprintf("%.4e") on values drawn over the range the
// matrix holds and beyond, on every representable neighbour of decimal ties and near-ties, and on the specials.
// Prints the number of values checked; exits 1 at the first difference.  Built and run by tests/test_abi_cpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../phylonium_amd/host/fmt_e4.hpp"

static unsigned long long checked = 0;
static void check(double v)
{
	char a[64], b[64];
	const size_t n = phyfmt::e4(a, v);
	a[n] = 0;
	snprintf(b, sizeof b, "%.4e", v);
	checked++;
	if (strcmp(a, b)) {
		printf("differs for %.17g (%a): '%s' vs '%s'\n", v, v, a, b);
		exit(1);
	}
}

int main(int argc, char **argv)
{
	const unsigned long long n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3000000ull;
	std::mt19937_64 rng(12345);
	std::uniform_real_distribution<double> u(0.0, 1.0);
	for (unsigned long long i = 0; i < n; i++) {
		const double x = u(rng);
		check(x);                                         // raw distances
		check(-0.75 * std::log(1.0 - 4.0 / 3.0 * x * 0.7)); // Jukes-Cantor over the usable range
		check(std::ldexp(x + 0.5, (int)(rng() % 90) - 75));  // 2^-75 .. 2^15
	}
	// decimal ties and their neighbours: d.dddd5 x 10^e is where rounding decides
	for (int e = -12; e <= 4; e++)
		for (unsigned m = 10000; m < 100000; m += 7) {
			const double t = ((double)m + 0.5) * std::pow(10.0, e - 4);
			double lo = t, hi = t;
			for (int k = 0; k < 3; k++) {
				check(lo);
				check(hi);
				lo = std::nextafter(lo, 0.0);
				hi = std::nextafter(hi, 1e300);
			}
		}
	for (double v : {0.0, -0.0, 1.0, 9.99995, 9.999949999, 0.099999499999, 0.5, 0.125, 12345.5, 12346.5, 99999.5, 1e5, 1e-22, 9.9e-23, 1e-300, 5e-324,
					 1e22, (double)INFINITY, -(double)INFINITY, (double)NAN, -1.5, 2.4833e-02})
		check(v);
	printf("%llu\n", checked);
	return 0;
}
