// asora_log10.hpp -- table-driven double-precision log10 for the table lookups of the sweep
// (photorates.f90:139: logtau = log10(max(1e-20, tau))).
//
// Why not the device library's log10: it costs ~105 VALU instructions (two per rate-producing
// cell, a third of the kernel); this one costs ~30 and is at least as accurate.
//
// Method (our own construction; the table idea is the classic one used by correctly-rounded
// and near-correctly-rounded libm logs): x = 2^k * z with z in [0.6875, 1.375); the top 7
// mantissa bits of z select a bin i with a tabulated double invc_i ~ 1/centre_i, so that
//     r = fma(z, invc_i, -1),  |r| <= 2^-8
// and with c_i := 1/invc_i EXACTLY,   log10 x = k*log10(2) + log10(c_i) + log10(1 + r).
//   * k*L2HI + CHI_i is exact in double (both are multiples of 2^-42, the sum is < 2^10);
//   * r*C1 (C1 = 1/ln 10) is formed as an exact two-term product (mul + fma residual);
//   * the leading sum is a Fast2Sum; every small term (k*L2LO, CLO_i, the residuals and the
//     r^2.. r^7 series) is accumulated in `lo` and added once.
// The only unrounded quantity is the final hi + lo, so the result is within ~0.51 ulp of the
// true value wherever |log10 x| >~ 2^-6 and within ~2^-58 absolute everywhere (the lookups add
// 20 to it, so absolute accuracy near x = 1 is what matters).  Domain: finite normal x > 0
// (the path clamps tau to >= 1e-20).  Checked against 80-bit log10l on the host
// (tests/host/log10_check.cpp, compiled and run by tests/test_boundary.py::test_log10_table_against_long_double);
// the device runs the same source (the rates of every -m gpu parity test go through it).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ASORA_HD __host__ __device__ __forceinline__
#else
#define ASORA_HD inline
#endif

namespace asora {

constexpr int LOG10_BINS = 128;
constexpr int LOG10_TABLE_DOUBLES = 3 * LOG10_BINS + 8;  // bins {invc, chi, clo} + constants

// layout of the constant tail
enum { L10_L2HI = 3 * LOG10_BINS, L10_L2LO, L10_C1HI, L10_C1LO };

constexpr uint32_t LOG10_OFF_HI = 0x3fe60000u;  // high word of 0.6875

// Host-side table construction in 80-bit long double (64-bit significand).
}  // namespace asora
#include <cmath>
#include <cstring>
namespace asora {
inline void build_log10_table(double *T) {
    const long double LD_L2 = 0.30102999566398119521373889472449302676818988146210854131L;
    const long double LD_C1 = 0.43429448190325182765112891891660508229439700580366656611L;
    const long double two42 = 4398046511104.0L;
    for (int i = 0; i < LOG10_BINS; i++) {
        // bin i: high word in [OFF + (i << 13), OFF + ((i+1) << 13))
        uint64_t lo_bits = ((uint64_t)(LOG10_OFF_HI + ((uint32_t)i << 13))) << 32;
        uint64_t hi_bits = ((uint64_t)(LOG10_OFF_HI + ((uint32_t)(i + 1) << 13))) << 32;
        double zlo, zhi;
        std::memcpy(&zlo, &lo_bits, 8);
        std::memcpy(&zhi, &hi_bits, 8);
        const double centre = 0.5 * (zlo + zhi);
        const double invc = (double)(1.0L / (long double)centre);
        const long double l10c = -log10l((long double)invc);  // log10 of c := 1/invc
        const double chi = (double)(roundl(l10c * two42) / two42);
        const double clo = (double)(l10c - (long double)chi);
        T[3 * i] = invc;
        T[3 * i + 1] = chi;
        T[3 * i + 2] = clo;
    }
    const double l2hi = (double)(floorl(LD_L2 * two42) / two42);
    T[L10_L2HI] = l2hi;
    T[L10_L2LO] = (double)(LD_L2 - (long double)l2hi);
    const double c1hi = (double)LD_C1;
    T[L10_C1HI] = c1hi;
    T[L10_C1LO] = (double)(LD_C1 - (long double)c1hi);
    for (int i = L10_C1LO + 1; i < LOG10_TABLE_DOUBLES; i++) T[i] = 0.0;
}

// -C1/2, C1/3, -C1/4, C1/5, -C1/6, C1/7 (C1 = 1/ln 10), rounded to double
constexpr double L10_P2 = -0.21714724095162591382556445945830254;
constexpr double L10_P3 = 0.14476482730108394255037630630553503;
constexpr double L10_P4 = -0.10857362047581295691278222972915127;
constexpr double L10_P5 = 0.086858896380650365530225783783321016;
constexpr double L10_P6 = -0.072382413650541971275188153152767514;
constexpr double L10_P7 = 0.062042068843321689664447417630943583;

template <typename TablePtr>
ASORA_HD double log10_tab(double x, TablePtr T) {
    uint64_t ix;
    __builtin_memcpy(&ix, &x, 8);
    const uint32_t hi32 = (uint32_t)(ix >> 32);
    const uint32_t tmp = hi32 - LOG10_OFF_HI;
    const int k = (int32_t)tmp >> 20;
    const int i = (int)((tmp >> 13) & (LOG10_BINS - 1));
    const uint64_t iz = ((uint64_t)(hi32 - (tmp & 0xfff00000u)) << 32) | (ix & 0xffffffffu);
    double z;
    __builtin_memcpy(&z, &iz, 8);
    const double invc = T[3 * i], chi = T[3 * i + 1], clo = T[3 * i + 2];
    const double L2HI = T[L10_L2HI], L2LO = T[L10_L2LO], C1HI = T[L10_C1HI], C1LO = T[L10_C1LO];
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, L2HI, chi);  // exact
    const double th = r * C1HI;
    const double tl = __builtin_fma(r, C1HI, -th);  // exact residual
    const double hi = w + th;
    const double e = (w - hi) + th;                 // Fast2Sum (exponent(w) >= exponent(th) or w == 0)
    const double r2 = r * r;
    double p = __builtin_fma(r, L10_P7, L10_P6);
    p = __builtin_fma(r, p, L10_P5);
    p = __builtin_fma(r, p, L10_P4);
    p = __builtin_fma(r, p, L10_P3);
    p = __builtin_fma(r, p, L10_P2);
    double lo = __builtin_fma(kd, L2LO, clo);
    lo = lo + e;
    lo = lo + tl;
    lo = __builtin_fma(r, C1LO, lo);
    lo = __builtin_fma(r2, p, lo);
    return hi + lo;
}

}  // namespace asora
