// asora_chem.hpp -- the chemistry pass that follows every raytrace in the evolve loop, on
// device-resident grids.
//
// WHAT is computed is the reference's chemistry.global_pass (src/c2ray/chemistry.f90:13-49):
// per cell evolve0D_global (:54-115) -> do_chemistry (:123-209, the <=400-iteration fixed point on
// the time-averaged ionised fraction) -> doric (:226-317, the analytic solution of the rate
// equation over dt), plus the count of cells whose time average moved by more than the tolerance
// (conv_flag) -- called on the host at pyc2ray/evolve.py:326-340.  The operations, their order and
// the default-real (float32-rounded) literals 1e4, 1e-3, 1e-8 follow the Fortran; the library is
// built with -ffp-contract=off, `/` and sqrt are IEEE-exact, exp and pow are the device library's
// (<= 1-2 ulp from glibc's, the only source of last-digit differences).
//
// HOW: cells are independent, so one lane per cell, grid-stride, seven coalesced 8-B streams
// (5 reads, 2 read-modify-writes = 72 B/cell).  The reference then forms sum(xh_intermed) and
// sum(1 - xh_intermed) with two more passes over the grid on the host (evolve.py:350-351); here the
// three reductions (conv_flag and the two sums) ride along: DPP-free wave shuffles, one LDS stage,
// three atomics per workgroup.  The temperature-only factors (recombination and collisional
// coefficients, one pow + one exp + one sqrt) are hoisted out of the fixed-point loop, and kept from
// cell to cell while the temperature does not change (the reference's drivers pass temp0 * ones):
// 256^3, one inner iteration per cell: 0.233 ms = 5.2 TB/s of the 72 B/cell (the device's copy rate);
// with a different temperature in every cell 0.38 ms (470 VALU instructions per cell, VALU 67 % busy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asora {

struct ChemStats {
    unsigned long long conv_flag;  // cells not converged (chemistry.f90:105-109)
    double sum_xh1;                // sum of xh_intermed           (evolve.py:350)
    double sum_xh0;                // sum of (1 - xh_intermed)     (evolve.py:351)
};

struct ChemArgs {
    const double *ndens, *temp, *xh, *phi_ion, *clump;
    double *xh_av, *xh_intermed;
    ChemStats *stats;
    long long ncell;
    double dt, bh00, albpow, colh0, temph0, abu_c;
};

constexpr int CHEM_THREADS = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ __launch_bounds__(CHEM_THREADS) void chemistry_kernel(ChemArgs A) {
    const double epsilon = 1e-14;                               // chemistry.f90:8
    const double minimum_fractional_change = (double)1.0e-3f;   // chemistry.f90:9 (default-real literal)
    const double minimum_fraction_of_atoms = (double)1.0e-8f;   // chemistry.f90:10
    const double small_deltht = (double)1.0e-8f;                // chemistry.f90:300
    const double t4 = (double)1e4f;                             // chemistry.f90:261
    unsigned long long nconv = 0;
    double s1 = 0.0, s0 = 0.0;
    const long long stride = (long long)gridDim.x * CHEM_THREADS;
    // The input streams of the NEXT cell of this lane are requested before the current cell's arithmetic runs
    // (xh_intermed is write-only: the fixed point always assigns it before it is read).
    struct CellIn { double ndens, temp, xh, phi, clump, xh_av; };
    auto fetch = [&](long long i) {
        CellIn c;
        c.ndens = __builtin_nontemporal_load(A.ndens + i); c.temp = __builtin_nontemporal_load(A.temp + i);
        c.xh = __builtin_nontemporal_load(A.xh + i); c.phi = __builtin_nontemporal_load(A.phi_ion + i);
        c.clump = __builtin_nontemporal_load(A.clump + i); c.xh_av = A.xh_av[i];
        return c;
    };
    double temp_key = __builtin_nan(""), pow_t4 = 0.0, acolh0 = 0.0;  // (NaN never compares equal: first cell computes)
    long long id = (long long)blockIdx.x * CHEM_THREADS + threadIdx.x;
    CellIn nxt = fetch(id < A.ncell ? id : 0);
    for (; id < A.ncell; id += stride) {
        const CellIn cur = nxt;
        nxt = fetch(id + stride < A.ncell ? id + stride : id);
        const double ndens_p = cur.ndens, temp_p = cur.temp, xh_p = cur.xh;
        const double phi_p = cur.phi, clump_p = cur.clump;
        const double xh_av_in = cur.xh_av;
        double xh_av_p = xh_av_in, xh_intermed_p;
        const double yh_av_p = 1.0 - xh_av_p;                   // chemistry.f90:98
        // doric, temperature-only part (chemistry.f90:261-266); isothermal => invariant of the fixed-point loop
        // Every driver of the reference fills the temperature grid with ONE value (temp0 * ones, c2ray_test.py:175,
        // c2ray_cubep3m.py:239): the two temperature-only factors -- 300 of the 470 instructions of a cell -- are
        // kept while no lane of the wavefront sees its temperature change (same function of the same argument:
        // bit-identical).
        if (__builtin_amdgcn_ballot_w64(temp_p != temp_key) != 0) {
            temp_key = temp_p;
            pow_t4 = pow(temp_p / t4, A.albpow);
            const double sqrtt0 = sqrt(temp_p);
            acolh0 = A.colh0 * sqrtt0 * exp(-A.temph0 / temp_p);
        }
        const double brech0 = clump_p * A.bh00 * pow_t4;
        int nit = 0;
        for (;;) {                                              // do_chemistry, chemistry.f90:151-208
            nit = nit + 1;
            const double xh_av_p_old = xh_av_p;
            const double de = ndens_p * (xh_av_p + A.abu_c);    // chemistry.f90:167
            // doric, chemistry.f90:279-312
            const double aih0 = phi_p + de * acolh0;
            const double delth = aih0 + de * brech0;
            const double eqxh = aih0 / delth;
            const double deltht = delth * A.dt;
            const double ee = exp(-deltht);
            double xh_new = (xh_p - eqxh) * ee + eqxh;
            if (xh_new < epsilon) xh_new = epsilon;
            const double avg_factor = (deltht < small_deltht) ? 1.0 : (1.0 - ee) / deltht;
            double xh_av_new = eqxh + (xh_p - eqxh) * avg_factor;
            if (xh_av_new < epsilon) xh_av_new = epsilon;
            xh_intermed_p = xh_new;
            xh_av_p = xh_av_new;
            // chemistry.f90:187-194 (the temperature term is identically 0 < tolerance)
            if (fabs((xh_av_p - xh_av_p_old) / (1.0 - xh_av_p)) < minimum_fractional_change ||
                (1.0 - xh_av_p < minimum_fraction_of_atoms))
                break;
            if (nit > 400) break;                               // chemistry.f90:197-207
        }
        // chemistry.f90:104-109
        if (fabs(xh_av_p - xh_av_in) > minimum_fractional_change &&
            fabs((xh_av_p - xh_av_in) / yh_av_p) > minimum_fractional_change &&
            yh_av_p > minimum_fraction_of_atoms)
            nconv++;
        A.xh_intermed[id] = xh_intermed_p;
        A.xh_av[id] = xh_av_p;
        s1 += xh_intermed_p;
        s0 += 1.0 - xh_intermed_p;
    }
    // workgroup reduction -> three atomics
    __shared__ double red[3][CHEM_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double c = wave_sum((double)nconv);  // exact: counts < 2^53
    s1 = wave_sum(s1);
    s0 = wave_sum(s0);
    if (lane == 0) { red[0][wave] = c; red[1][wave] = s1; red[2][wave] = s0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, d = 0.0;
        for (int w = 0; w < CHEM_THREADS / 64; w++) { a += red[0][w]; b += red[1][w]; d += red[2][w]; }
        if (a != 0.0) atomicAdd(&A.stats->conv_flag, (unsigned long long)a);
        atomicAdd(&A.stats->sum_xh1, b);
        atomicAdd(&A.stats->sum_xh0, d);
    }
}

}  // namespace asora
