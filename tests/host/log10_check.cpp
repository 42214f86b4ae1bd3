// Host check of pyc2ray_amd/csrc/asora_log10.hpp against 80-bit log10l.
// prints: max error in ulp (of the result), max absolute error, fraction not correctly rounded,
// fraction differing from glibc log10, results more than 1.5 ulp away from glibc's, max distance from glibc's in ulp
// (the last three where |log10 x| > 1e-2).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../pyc2ray_amd/csrc/asora_log10.hpp"
int main() {
    double T[asora::LOG10_TABLE_DOUBLES];
    asora::build_log10_table(T);
    std::mt19937_64 g(42);
    std::uniform_real_distribution<double> u(-20.0, 5.0), v(-1e-3, 1e-3);
    double max_ulp = 0, max_abs = 0, max_gl = 0;
    long n = 4000000, ncr = 0, ndiff = 0, ndiff2 = 0;
    for (long i = 0; i < n; i++) {
        double x = (i % 4 == 3) ? 1.0 + v(g) : std::pow(10.0, u(g));
        double y = asora::log10_tab(x, (const double *)T);
        long double t = log10l((long double)x);
        double cr = (double)t;
        double gl = std::log10(x);
        long double err = fabsl((long double)y - t);
        double ulp = std::nextafter(std::fabs(cr), INFINITY) - std::fabs(cr);
        if (std::fabs(cr) > 1e-2 && (double)(err / ulp) > max_ulp) max_ulp = (double)(err / ulp);
        if ((double)err > max_abs) max_abs = (double)err;
        if (y != cr) ncr++;
        if (y != gl) ndiff++;
        if (std::fabs(y - gl) > 1.5 * ulp && std::fabs(cr) > 1e-2) ndiff2++;
        if (std::fabs(cr) > 1e-2 && std::fabs(y - gl) / ulp > max_gl) max_gl = std::fabs(y - gl) / ulp;
    }
    printf("%.4f %.3e %.5f %.5f %ld %.3f\n", max_ulp, max_abs, (double)ncr / n, (double)ndiff / n, ndiff2, max_gl);
    return 0;
}
