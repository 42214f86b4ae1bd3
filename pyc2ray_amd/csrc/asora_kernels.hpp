// asora_kernels.hpp -- device code of the MI355X (gfx950) ASORA raytracer.
//
// WHAT IS COMPUTED is fixed by the reference: for every source, every cell of
// octahedron(q_max) ∩ periodic-box limits (src/asora/raytracing.cu:110,140-141,214-255) gets
//   cdh_in  = short-characteristics interpolation of the outgoing column density of its four
//             upwind neighbours (src/c2ray/raytracing.f90:576-815 == src/asora/raytracing.cu:354-439)
//   cdh_out = cdh_in + nHI * path                                  (raytracing.f90:488)
//   Gamma   = table rates(cdh_in, cdh_out) / nHI, added to phi_ion  (raytracing.f90:495-537,
//             src/c2ray/photorates.f90:62-149), unless cdh_in > 2e30f or dist^2/dr^2 > R^2.
// The ARITHMETIC follows the Fortran oracle operation for operation (float32-rounded sqrt2/sqrt3,
// 1e-7f threshold, tau_in for the thin lookup, no FMA contraction -- SURVEY.md 3.4).
//
// HOW it is computed is NOT the reference's design (one block per source, Manhattan shells,
// an N^3 column-density grid per in-flight source in global memory, a barrier per shell):
//
//   * A cell reads only cells of the plane one step closer to the source along its DOMINANT axis.
//     With the reference's tie rule (z before y before x) the upwind stencil of a closed pyramid
//     {|d_p| >= |d_u|, |d_p| >= |d_v|} never leaves that pyramid with non-zero weight, and inside a
//     pyramid the four sign quadrants (su, sv) are closed too.  So one source splits into
//     6 pyramids x 4 quadrants = 24 independent SECTORS.  A sector is a march over planes
//     n = 1..nmax; plane n holds cells (a, b), 0 <= a, b <= n, and cell (a, b) reads
//     (a-1,b-1), (a,b-1), (a-1,b), (a,b) of plane n-1.
//   * One 256-thread workgroup owns one (source, sector) task.  Two planes of column densities
//     live in LDS (double buffer, one barrier per plane); nothing but the final rates ever goes
//     to HBM.  Memory per task is O(n^2), not O(N^3), so the number of sources in flight is set
//     by occupancy, not by a batch-size parameter.
//   * Cells on sector boundaries (a = 0, b = 0, a = n, b = n) are computed by 2-3 sectors (they
//     are needed as upwind values) but OWNED -- i.e. produce rates / cdh output -- by exactly one,
//     chosen with the reference's own rules (tie order z,y,x; sign(0) = +1).
//   * In-plane coordinate b runs along the memory-contiguous axis: x- and y-sectors read the
//     neutral density and add rates in the [i][j][k] layout, z-sectors in a transposed [i][k][j]
//     copy.  A wavefront therefore touches whole rows (coalesced loads and coalesced f64 atomics).
//     The two rate grids are merged by finalize_kernel.
//   * nHI = ndens * (1 - xh) is formed once per call (prep_nhi_kernel), so a visit loads 8 B.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "asora_log10.hpp"

namespace asora {

#ifndef ASORA_SWEEP_THREADS
#define ASORA_SWEEP_THREADS 256
#endif
constexpr int SWEEP_THREADS = ASORA_SWEEP_THREADS;
constexpr int SECTORS = 24;
constexpr double FOURPI = 4.0 * 3.14159265358979323846264338;  // raytracing.f90:47,457
constexpr double SQRT3_F32 = 1.7320507764816284;                // (double)sqrtf(3.0f), raytracing.f90:608
constexpr double SQRT2_F32 = 1.4142135381698608;                // (double)sqrtf(2.0f), raytracing.f90:609
constexpr double SQRT3_CUDA = 1.73205080757;                    // src/asora/raytracing.cu:20-21
constexpr double SQRT2_CUDA = 1.41421356237;

// One plane of one geometry variant in the cell tables: cells [off, off+cntA) produce rates
// (dist2/dr2 <= R2 with the oracle's exact arithmetic), [off+cntA, off+cntT) only propagate cdh.
// Entry 0 of a variant (the source plane) holds off = staging position of cell (0, 0).
// (alam = (n - 1/2) / n, raytracing.f90:683, the same for every cell of plane n, is formed on the device: cinterp_alam.)
// 12 bytes per plane: the table is copied into every workgroup's LDS, where a few hundred bytes decide how many
// workgroups a CU holds (R = 30: 41 640 B with 24-byte entries, three workgroups per CU; 40 748 B, four).
struct PlaneInfo {
    uint32_t off, cntA, cntT;
};
static_assert(sizeof(PlaneInfo) == 12, "plane table entries are packed");
// doubles the plane table of n planes occupies in LDS
__host__ __device__ constexpr int plane_table_doubles(int n) { return (n * (int)sizeof(PlaneInfo) + 7) / 8; }

struct SweepArgs {
    const double *nhiA;  // nHI, [i][j][k]
    const double *nhiB;  // nHI, [i][k][j]
    double *phiA, *phiB;    // rate accumulators in the two layouts
    double *heatA, *heatB;  // heating accumulators (HEAT only)
    double *cdhA, *cdhB;    // column density of source `cdh_src` in the two layouts (may be null)
    const int *src_pos;     // 0-based, interleaved, as given by the caller (may lie outside [0, N): periodic)
    const double *src_flux;
    const double *thick, *thin, *hthick, *hthin;  // padded tables
    const double *log10_table;                    // LOG10_TABLE_DOUBLES doubles (asora_log10.hpp)
    double *gscratch;       // global staging (when LDS is too small), 2*plane_elems per block
    long long ntasks;       // 24 * number of sources of this launch (fewer when a workgroup takes several sectors / sources)
    int nsrc;               // sources of this launch (the packed kernels' last group may be partial)
    int src_begin;          // first source of this launch
    int cdh_src;            // source whose column density is exported (-1: none)
    int N, q_max, last_l, last_r;  // last_l > 0: lowest offset is -last_l
    int S;                  // row stride of a staged plane (max width + 1 guard)
    int plane_elems;        // rows * S
    int num_tau;
    double R, sig, dr, minlogtau, dlogtau;
    double rdlogtau;        // correctly rounded 1/dlogtau (host)
    // table-driven fast path (sweep_tab_kernel)
    const PlaneInfo *planes;   // [variant][nplanes]
    const uint2 *cell_meta;    // .x = a | b << 10 | flags << 20, .y = staging position of (a-1,b-1) | of (a,b-1) << 16;
                               // list A (rate-producing) then list B per plane
    const double2 *cell_geo;   // {path * dr, vol_ph = dist2 * (path*dr) * 4pi}
    int nplanes;
    int cw_entries;            // staged (column density, weight) pairs per task (max over variants)
    int wtab;                  // rows/columns per plane (max) = entries of the per-row / per-column LDS tables
    int vmap[8];               // ((u_is_neg*2 + v_is_neg)*2 + (axis != 0)) -> variant
    int rate_only;             // option "skip_unobserved": sources other than cdh_src sweep list A only
    int nmax_rate;             // last plane with a cell inside the R-sphere
    double tau_sat;            // optical depth from which both table look-ups of a cell clamp to the last entry
    int single;                // ONE grid layout (short calls): z-dominant sectors address the [i][j][k] grids too (nhiA / phiA /
                               // heatA / cdhA, in-plane coordinate b strided by N); no transposed copies, no merge pass
};

// ------------------------------------------------------------------------------------------
// IEEE-exact f64 division without the range-handling instructions.  hipcc expands a/b into
// v_div_scale x2, v_rcp, two Newton steps on the reciprocal, q = a*r, one residual correction
// (v_div_fmas) and v_div_fixup.  The scale/fixup instructions only matter when an operand or the
// quotient is subnormal / near overflow; every divisor on this path is a normal number far from
// both ends (0.6 <= max(0.6, tau), sums of positive weights, cell volumes, densities, table
// steps), so the same Newton + correction sequence without them gives the identical, correctly
// rounded quotient in 8 instead of 12 instructions (checked bit-for-bit on the GPU against `/`
// by asora_test_division, tests/test_gpu_parity.py::test_exact_division_helpers).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_refined(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double div_exact(double n, double d) {
    const double r = rcp_refined(d);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
__device__ __forceinline__ double rcp_exact(double d) {  // 1.0 / d
    const double r = rcp_refined(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
// x / d with rd = the correctly rounded 1/d computed once (Markstein's correction step)
__device__ __forceinline__ double div_const(double x, double d, double rd) {
    const double q = x * rd;
    return __builtin_fma(__builtin_fma(-d, q, x), rd, q);
}

struct Lookup {
    int i0;
    double residual;
};

// log10 through the LDS-resident table of asora_log10.hpp (ASORA_OCML_LOG10: device library's log10)
extern __shared__ __attribute__((aligned(16))) double asora_smem[];
struct Log10LDS {
    int off;  // position of the log10 table in dynamic LDS (doubles)
    __device__ __forceinline__ double operator()(double x) const {
#ifdef ASORA_OCML_LOG10
        return log10(x);
#else
        return log10_tab(x, asora_smem + off);
#endif
    }
};
__device__ __forceinline__ Log10LDS load_log10_table(const double *__restrict__ src, int off, int tid, int nthreads) {
    for (int i = tid; i < LOG10_TABLE_DOUBLES; i += nthreads) asora_smem[off + i] = src[i];
    __syncthreads();
    Log10LDS l;
    l.off = off;
    return l;
}

// photorates.f90:139-143 (index part of photo_lookuptable)
__device__ __forceinline__ Lookup table_index(double tau, double minlogtau, double dlogtau, double rdlogtau,
                                              int num_tau, Log10LDS lg) {
    double logtau = lg(fmax(1.0e-20, tau));
    double real_i = fmin((double)num_tau, fmax(0.0, 1.0 + div_const(logtau - minlogtau, dlogtau, rdlogtau)));
    Lookup L;
    L.i0 = (int)real_i;
    L.residual = real_i - (double)L.i0;
    return L;
}

// photorates.f90:146; the table is padded so that i0 + 1 is always readable.
__device__ __forceinline__ double table_eval(const double *__restrict__ t, Lookup L) {
    double t0 = t[L.i0], t1 = t[L.i0 + 1];
    return t0 + L.residual * (t1 - t0);
}

// photorates.f90:62-125 followed by the division by nHI of raytracing.f90:531-532
template <bool HEAT>
__device__ __forceinline__ void cell_rates(const SweepArgs &A, Log10LDS lg, double flux, double cdh_in,
                                           double cdh_out, double vol_ph, double nHI, double &phi, double &heat) {
    const double tau_photo_limit = (double)1.0e-7f;
    double tau_in = cdh_in * A.sig;
    double tau_out = cdh_out * A.sig;
    double prefact = div_exact(flux, vol_ph);
    Lookup Lin = table_index(tau_in, A.minlogtau, A.dlogtau, A.rdlogtau, A.num_tau, lg);
    double phi_in = prefact * table_eval(A.thick, Lin);
    if (fabs(tau_out - tau_in) > tau_photo_limit) {
        Lookup Lout = table_index(tau_out, A.minlogtau, A.dlogtau, A.rdlogtau, A.num_tau, lg);
        double phi_out = prefact * table_eval(A.thick, Lout);
        phi = phi_in - phi_out;
        if (HEAT) heat = prefact * (table_eval(A.hthick, Lin) - table_eval(A.hthick, Lout));
    } else {
        phi = prefact * (tau_out - tau_in) * table_eval(A.thin, Lin);
        if (HEAT) heat = prefact * (tau_out - tau_in) * table_eval(A.hthin, Lin);
    }
    phi = div_exact(phi, nHI);
    if (HEAT) heat = div_exact(heat, nHI);
}

// The same rates in three steps, so that a wave can have the table look-ups of several cells in
// flight at once: (1) optical depths and table positions, (2) the loads, (3) the arithmetic.  Both
// branches of photorates.f90:113-125 are evaluated and the result is selected, which is
// bit-identical to taking the branch.
struct RatePrep {
    Lookup Lin, Lout;
    double prefact, dtau;
};
struct RateLoads {
    double2 thick_in, thick_out, hthick_in, hthick_out;
};
__device__ __forceinline__ RatePrep rates_prepare(const SweepArgs &A, Log10LDS lg, double flux, double cdh_in,
                                                  double cdh_out, double vol_ph) {
    RatePrep r;
    const double tau_in = cdh_in * A.sig;
    const double tau_out = cdh_out * A.sig;
    r.prefact = div_exact(flux, vol_ph);
    r.dtau = tau_out - tau_in;
    r.Lin = table_index(tau_in, A.minlogtau, A.dlogtau, A.rdlogtau, A.num_tau, lg);
    r.Lout = table_index(tau_out, A.minlogtau, A.dlogtau, A.rdlogtau, A.num_tau, lg);
    return r;
}
__device__ __forceinline__ double2 table_pair(const double *__restrict__ t, int i0) {
    const char *p = reinterpret_cast<const char *>(t) + (uint32_t)i0 * 8u;  // SGPR base + 32-bit offset
    double2 v;
    v.x = *reinterpret_cast<const double *>(p);
    v.y = *reinterpret_cast<const double *>(p + 8);
    return v;
}
template <bool HEAT>
__device__ __forceinline__ RateLoads rates_load(const SweepArgs &A, const RatePrep &r) {
    RateLoads l;
    l.thick_in = table_pair(A.thick, r.Lin.i0);
    l.thick_out = table_pair(A.thick, r.Lout.i0);
    if (HEAT) {
        l.hthick_in = table_pair(A.hthick, r.Lin.i0);
        l.hthick_out = table_pair(A.hthick, r.Lout.i0);
    }
    return l;
}
__device__ __forceinline__ double pair_eval(double2 t, double residual) { return t.x + residual * (t.y - t.x); }
template <bool HEAT>
__device__ __forceinline__ void rates_finish(const SweepArgs &A, const RatePrep &r, const RateLoads &l, double nHI,
                                             double &phi, double &heat) {
    const double tau_photo_limit = (double)1.0e-7f;
    const bool thick = fabs(r.dtau) > tau_photo_limit;
    const double phi_in = r.prefact * pair_eval(l.thick_in, r.Lin.residual);
    const double phi_out = r.prefact * pair_eval(l.thick_out, r.Lout.residual);
    double p = phi_in - phi_out;
    double h = 0.0;
    if (HEAT) h = r.prefact * (pair_eval(l.hthick_in, r.Lin.residual) - pair_eval(l.hthick_out, r.Lout.residual));
    // optically thin cells are rare: their table is only touched by waves that hold one
    if (__builtin_amdgcn_ballot_w64(!thick) != 0) {
        const double p_thin = r.prefact * r.dtau * pair_eval(table_pair(A.thin, r.Lin.i0), r.Lin.residual);
        p = thick ? p : p_thin;
        if (HEAT) {
            const double h_thin = r.prefact * r.dtau * pair_eval(table_pair(A.hthin, r.Lin.i0), r.Lin.residual);
            h = thick ? h : h_thin;
        }
    }
    phi = div_exact(p, nHI);
    if (HEAT) heat = div_exact(h, nHI);
}

// raytracing.f90:807-813
__device__ __forceinline__ double weightf(double cd, double sig) { return rcp_exact(fmax(0.6, cd * sig)); }

// The interpolation of raytracing.f90:662-710 / 715-758 / 764-800 in sector coordinates.
//   n >= 1 : distance along the dominant axis;  a, b >= 0 : in-plane distances (u, v axes)
//   su, sv : signs of the in-plane offsets;  u1, v1 : 1-BASED source coordinates on those axes
//            (the oracle's crossing point xc = alam*di + real(i0) carries the absolute position)
//   alam   : ((n-1) + 0.5)/n, the oracle's (real(km-k0)+sgnk*0.5)/dk  (sign-free)
//   c1..c4 : outgoing column densities of (a-1,b-1), (a,b-1), (a-1,b), (a,b) in plane n-1
__device__ __forceinline__ double cinterp_cd(int n, int a, int b, int su, int sv, int u1, int v1, double alam,
                                             double c1, double c2, double c3, double c4, double sig,
                                             double sqrt3, double sqrt2) {
    double dU = (double)(su * a), dV = (double)(sv * b);
    double uc = alam * dU + (double)u1;
    double vc = alam * dV + (double)v1;
    double uh = (double)(u1 + su * (a - 1)) + 0.5 * (double)su;
    double vh = (double)(v1 + sv * (b - 1)) + 0.5 * (double)sv;
    double du = 2.0 * fabs(uc - uh);
    double dv = 2.0 * fabs(vc - vh);
    double s1 = (1. - du) * (1. - dv);
    double s2 = (1. - dv) * du;
    double s3 = (1. - du) * dv;
    double s4 = du * dv;
    double w1 = s1 * weightf(c1, sig);
    double w2 = s2 * weightf(c2, sig);
    double w3 = s3 * weightf(c3, sig);
    double w4 = s4 * weightf(c4, sig);
    double cdensi = div_exact(c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4, w1 + w2 + w3 + w4);
    if (n == 1 && (a == 1 || b == 1)) {
        if (a == 1 && b == 1) cdensi = sqrt3 * cdensi;
        else cdensi = sqrt2 * cdensi;
    }
    return cdensi;
}
__device__ __forceinline__ double cinterp_alam(int n) {
    const double dn = (double)n;
    return div_exact(dn - 0.5, dn);
}
// path length through the cell in cell units, raytracing.f90:710 / 758 / 800
__device__ __forceinline__ double cinterp_path(int n, int a, int b) {
    const double dn = (double)n;
    return sqrt(div_exact((double)(a * a + b * b), dn * dn) + 1.0);
}
__device__ __forceinline__ void cinterp_sector(int n, int a, int b, int su, int sv, int u1, int v1, double c1,
                                               double c2, double c3, double c4, double sig, double sqrt3,
                                               double sqrt2, double &cdensi, double &path) {
    cdensi = cinterp_cd(n, a, b, su, sv, u1, v1, cinterp_alam(n), c1, c2, c3, c4, sig, sqrt3, sqrt2);
    path = cinterp_path(n, a, b);
}

// Fortran-type modulo (src/asora/raytracing.cu:32): source positions may lie outside [0, N) -- the reference wraps
// them on every grid access (mem_offst_gpu) while its interpolation carries the coordinate AS GIVEN
// (raytracing.f90:683 xc = alam*di + real(i0)), so both forms are kept: wrapped for addressing, raw for arithmetic.
__device__ __forceinline__ int fmodulo(int a, int N) {
    const int r = a % N;
    return r < 0 ? r + N : r;
}
__device__ __forceinline__ int wrap(int x, int N) {
    x += (x < 0) ? N : 0;
    x -= (x >= N) ? N : 0;
    return x;
}

template <bool LDS_STAGE>
struct Stage {
    double *g;
    __device__ __forceinline__ double ld(int off) const;
    __device__ __forceinline__ void st(int off, double v) const;
};
template <>
__device__ __forceinline__ double Stage<true>::ld(int off) const { return asora_smem[off]; }
template <>
__device__ __forceinline__ void Stage<true>::st(int off, double v) const { asora_smem[off] = v; }
template <>
__device__ __forceinline__ double Stage<false>::ld(int off) const { return g[off]; }
template <>
__device__ __forceinline__ void Stage<false>::st(int off, double v) const { g[off] = v; }

// ------------------------------------------------------------------------------------------
// The sweep: one workgroup per (source, sector) task (grid-stride over tasks).
// ------------------------------------------------------------------------------------------
template <bool LDS_STAGE, bool HEAT>
__global__ __launch_bounds__(SWEEP_THREADS) void sweep_kernel(SweepArgs A) {
    Stage<LDS_STAGE> stg;
    stg.g = LDS_STAGE ? nullptr : A.gscratch + (size_t)blockIdx.x * 2 * (size_t)A.plane_elems;
    const int tid = threadIdx.x;
    const int S = A.S;
    const int plane = A.plane_elems;
    const int N = A.N;
    const size_t N2 = (size_t)N * N;
    const double max_coldensh = (double)2e30f;  // raytracing.f90:368
    const double R2 = A.R * A.R;
    const double dr2 = A.dr * A.dr;
    const Log10LDS lg = load_log10_table(A.log10_table, LDS_STAGE ? 2 * plane : 0, tid, SWEEP_THREADS);

    for (long long task = blockIdx.x; task < A.ntasks; task += gridDim.x) {
        const int s = (int)(task / SECTORS);
        const int t = (int)(task - (long long)s * SECTORS);
        const int src = A.src_begin + s;
        const int axis = t >> 3;  // 0: z-dominant, 1: y-dominant, 2: x-dominant
        const int sp = (t & 4) ? -1 : 1;
        const int su = (t & 2) ? -1 : 1;
        const int sv = (t & 1) ? -1 : 1;
        const int i0r = A.src_pos[3 * src + 0], j0r = A.src_pos[3 * src + 1], k0r = A.src_pos[3 * src + 2];
        const int i0 = fmodulo(i0r, A.N), j0 = fmodulo(j0r, A.N), k0 = fmodulo(k0r, A.N);
        const double flux = A.src_flux[src];
        int p0, u0, v0, ur1, vr1;  // wrapped position (addressing); 1-based in-plane coordinates as given (arithmetic)
        if (axis == 0) { p0 = k0; u0 = i0; v0 = j0; ur1 = i0r + 1; vr1 = j0r + 1; }
        else if (axis == 1) { p0 = j0; u0 = i0; v0 = k0; ur1 = i0r + 1; vr1 = k0r + 1; }
        else { p0 = i0; u0 = j0; v0 = k0; ur1 = j0r + 1; vr1 = k0r + 1; }
        const double *__restrict__ nhi = (axis == 0) ? A.nhiB : A.nhiA;
        double *__restrict__ phi = (axis == 0) ? A.phiB : A.phiA;
        double *__restrict__ heat = (axis == 0) ? A.heatB : A.heatA;
        double *__restrict__ cdh = (axis == 0) ? A.cdhB : A.cdhA;
        const bool write_cdh = (src == A.cdh_src) && (cdh != nullptr);
        const int limp = sp > 0 ? A.last_r : A.last_l;
        const int limu = su > 0 ? A.last_r : A.last_l;
        const int limv = sv > 0 ? A.last_r : A.last_l;
        const int nmax = min(limp, A.q_max);
        // index = pw*strideP + uw*strideU + vw
        const size_t strideP = (axis == 2) ? N2 : (size_t)N;
        const size_t strideU = (axis == 2) ? (size_t)N : N2;

        // guard cells and never-written cells must read as 0 (their weights are 0 or O(eps))
        for (int i = tid; i < 2 * plane; i += SWEEP_THREADS) stg.st(i, 0.0);
        __syncthreads();

        if (tid == 0) {
            // source cell, raytracing.f90:430-439
            size_t g0 = (size_t)p0 * strideP + (size_t)u0 * strideU + (size_t)v0;
            double nHI = nhi[g0];
            double path = 0.5 * A.dr;
            double cdo = 0.0 + nHI * path;
            stg.st(S + 1, cdo);
            if (t == 0) {
                double vol_ph = A.dr * A.dr * A.dr;
                double ph, ht = 0.0;
                cell_rates<HEAT>(A, lg, flux, 0.0, cdo, vol_ph, nHI, ph, ht);
                unsafeAtomicAdd(phi + g0, ph);
                if (HEAT) unsafeAtomicAdd(heat + g0, ht);
                if (write_cdh) cdh[g0] = cdo;
            }
        }
        __syncthreads();

        for (int n = 1; n <= nmax; n++) {
            const int cur = (n & 1) * plane, prev = ((n - 1) & 1) * plane;
            const int cut = A.q_max - n;  // a + b <= cut
            const int Hn = min(min(n, limu), cut) + 1;
            const int Wn = min(min(n, limv), cut) + 1;
            const int ncell = Hn * Wn;
            const float invW = 1.0f / (float)Wn;
            const int pw = wrap(p0 + sp * n, N);
            const size_t baseP = (size_t)pw * strideP;
            const double P2 = (A.dr * (double)n) * (A.dr * (double)n);
            for (int idx = tid; idx < ncell; idx += SWEEP_THREADS) {
                const int a = (int)(((float)idx + 0.5f) * invW);
                const int b = idx - a * Wn;
                if (a + b > cut) continue;
                const int uw = wrap(u0 + su * a, N), vw = wrap(v0 + sv * b, N);
                const size_t g = baseP + (size_t)uw * strideU + (size_t)vw;
                const double nHI = nhi[g];
                const double c1 = stg.ld(prev + a * S + b);
                const double c2 = stg.ld(prev + (a + 1) * S + b);
                const double c3 = stg.ld(prev + a * S + b + 1);
                const double c4 = stg.ld(prev + (a + 1) * S + b + 1);
                double cdh_in, path;
                cinterp_sector(n, a, b, su, sv, ur1, vr1, c1, c2, c3, c4, A.sig, SQRT3_F32, SQRT2_F32,
                               cdh_in, path);
                path = path * A.dr;
                // raytracing.f90:448-451: dist2 = xs*xs + ys*ys + zs*zs in x, y, z order
                const double U2 = (A.dr * (double)a) * (A.dr * (double)a);
                const double V2 = (A.dr * (double)b) * (A.dr * (double)b);
                const double dist2 = (axis == 0) ? (U2 + V2) + P2 : (U2 + P2) + V2;
                const double vol_ph = dist2 * path * FOURPI;
                const bool stop = (div_exact(dist2, dr2) > R2) || (cdh_in > max_coldensh);
                const double cdo = cdh_in + nHI * path;
                stg.st(cur + (a + 1) * S + b + 1, cdo);
                // ownership: reference tie order (z, y, x) and sign(0) = +1
                const bool ghost = (su < 0 && a == 0) || (sv < 0 && b == 0) || (axis >= 1 && b == n) ||
                                   (axis == 2 && a == n);
                if (!ghost) {
                    if (write_cdh) cdh[g] = cdo;
                    if (!stop) {
                        double ph, ht = 0.0;
                        cell_rates<HEAT>(A, lg, flux, cdh_in, cdo, vol_ph, nHI, ph, ht);
                        unsafeAtomicAdd(phi + g, ph);
                        if (HEAT) unsafeAtomicAdd(heat + g, ht);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Table-driven sweep (the fast path).  Same task decomposition and arithmetic as sweep_kernel;
// the kernel is bound by VALU issue (rocprofv3: 62 % VALU busy, 244 instructions per wave-visit in
// the first version), so everything that is not the oracle's f64 arithmetic on cell DATA is
// moved out of the per-cell path:
//
//   * per (N, R, dr), on the host (build_geometry in asora_capi.hip), with the same IEEE operations
//     as the oracle: membership in octahedron ∩ box, the exact R-sphere predicate, path*dr, vol_ph,
//     the ownership flags and the staging positions of every cell -> 24 B per cell, streamed from
//     L2 / Infinity Cache.  Per plane the cells form two dense lists: A (inside the R sphere:
//     column density + rates) and B (outside: column density only), cut into 64-cell chunks, so
//     wavefronts are fully populated and never diverge on the sphere or the octahedron cut.
//   * per task, by the workgroup, into LDS: the periodic wrap of every row and column
//     (rowU[a], colV[b]: a cell's grid offset is base + rowU[a] + colV[b]).
//   * per plane, by the workgroup, into LDS: the interpolation offsets, which depend only on the row
//     (du[a], 1 - du[a]) or only on the column (dv[b], 1 - dv[b]) -- raytracing.f90:688-700.
//   * per PRODUCED cell instead of per consumed cell: the weight 1/max(0.6, tau) (raytracing.f90:
//     807-813) is stored next to the column density; each staged value is read by four cells.
//
// Staging.  ONE array of (column density, weight) pairs, in LDS, updated IN PLACE, laid out as the
// packed union of all planes' domains (rows of decreasing length, one leading guard per row and
// a guard row): 16 B x ~2000 entries at R = 60.  Occupancy on this kernel is worth a lot (measured:
// 3 -> 2 workgroups per CU costs 29 %), hence the packing and the single array.  In-place is safe
// because of the processing order.  A plane is worked off in PHASES of up to NW*P chunks; within a
// phase every wave first computes its cells from the old values (results stay in registers),
// then -- after a barrier -- all waves write.  Across phases chunks are taken in DECREASING order:
// list B from its last chunk down, then list A from its last chunk down.  Cell (a, b) reads the old
// values at (a-1,b-1), (a,b-1), (a-1,b), (a,b): all at row-major positions <= its own, and for an
// A cell all inside list A (the sphere's row extent shrinks with a), so nothing it reads has been
// overwritten by an earlier phase.  Positions a plane does not write (beyond the octahedron cut)
// are never read later with non-zero weight; never-written positions and guards hold (0, 1/0.6).
// ------------------------------------------------------------------------------------------
// chunks per wave per phase.  Measured on the 256^3 / R = 60 sweep (4 waves per SIMD, ms per 1e4 sources):
// 1: 135, 2: 121 (no spills), 3: 122, 4: 292 (spills); with the branch-free fast path: 2: 100, 3: 103
// Workgroups of at least this many threads run their two halves half a phase apart (STAG in sweep_tab_kernel); 1 << 30: never
#ifndef ASORA_STAGGER_MIN_THREADS
#define ASORA_STAGGER_MIN_THREADS 1024
#endif

#ifndef ASORA_PHASE_CHUNKS
#define ASORA_PHASE_CHUNKS 2
#endif
#ifndef ASORA_WAVES_PER_EU
#define ASORA_WAVES_PER_EU 4
#endif
// ... of the variants with the heating rate as second output (their rate batch holds twice the table pairs)
#ifndef ASORA_WAVES_PER_EU_HEAT
#define ASORA_WAVES_PER_EU_HEAT ASORA_WAVES_PER_EU
#endif
#ifndef ASORA_HEAT_BATCH
#define ASORA_HEAT_BATCH 0
#endif

// Element access at a 32-bit BYTE offset from a wave-uniform base: lets the compiler use the
// "SGPR base + 32-bit VGPR offset" addressing mode instead of forming a 64-bit address per lane.
// (the table-driven kernel only runs when every grid fits 2^31 bytes: host check `cells < 2^28`)
template <typename T>
__device__ __forceinline__ const T &at_byte(const T *base, uint32_t byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ T *ptr_byte(T *base, uint32_t byte_off) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}

// Rate accumulation through BUFFER atomics: a lane that has nothing to add gets an offset beyond the
// descriptor's range and the hardware drops it -- no branch around the instruction.  That matters because the
// memory counter retires in issue order and the compiler must assume the SHORTEST path when it counts how
// many younger operations may stay in flight: with a conditional atomic every wait for a load issued
// before the atomics (the next phase's geometry / neutral density) also waited for the atomics themselves
// (memory-side, ~2 us).  Issued unconditionally they are counted exactly and complete behind the next
// phase's interpolation.  (The intrinsic is the one the backend selects buffer_atomic_add_f64 from; clang has
// no builtin for the f64 form.)
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__device__ double llvm_amdgcn_raw_buffer_atomic_fadd_f64(double vdata, rsrc_t rsrc, int voffset, int soffset,
                                                         int aux) __asm("llvm.amdgcn.raw.buffer.atomic.fadd.f64");
constexpr uint32_t BUFFER_OOB = 0x80000000u;  // >= num_records of every grid the table-driven kernel accepts
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));  // stride 0: raw buffer
    r.z = (int)bytes;
    r.w = 0x00020000;  // gfx9 raw-buffer word 3 (32-bit data format), no swizzle
    return r;
}

// cell_meta.x
constexpr uint32_t META_COORD_BITS = 10, META_COORD_MASK = (1u << META_COORD_BITS) - 1;
constexpr uint32_t META_FLAG_SHIFT = 2 * META_COORD_BITS;
enum : uint32_t { FLAG_A0 = 1, FLAG_B0 = 2, FLAG_AN = 4, FLAG_BN = 8 };  // a == 0, b == 0, a == n, b == n

// Staged column densities of one workgroup.  WSTAGED: the weight 1/max(0.6, tau) is staged next to the
// column density (16 B per entry, computed once per produced cell); otherwise only the column density is
// staged (8 B per entry: twice the radius range fits in LDS) and the four weights are recomputed per cell.
template <bool LDS_STAGE, bool WSTAGED>
struct StageCW {
    double2 *g;  // global staging (when LDS is too small)
    double sig;
    __device__ __forceinline__ double2 ld(uint32_t p) const {
        if (!WSTAGED) {
            double2 v;
            v.x = asora_smem[p];
            v.y = weightf(v.x, sig);
            return v;
        }
        if (LDS_STAGE) return reinterpret_cast<const double2 *>(asora_smem)[p];
        return g[p];
    }
    __device__ __forceinline__ void st(uint32_t p, double c, double w) const {
        if (!WSTAGED) {
            asora_smem[p] = c;
            return;
        }
        double2 v;
        v.x = c;
        v.y = w;
        if (LDS_STAGE) reinterpret_cast<double2 *>(asora_smem)[p] = v;
        else g[p] = v;
    }
};

// Staging array + log10 table of one workgroup.
template <bool LDS_STAGE, bool WSTAGED>
struct PlaneCtx {
    StageCW<LDS_STAGE, WSTAGED> stg;
    Log10LDS lg;
};

// raytracing.f90:683-700 for one row (or column): x1 = 1-based source coordinate, s = sign
__device__ __forceinline__ double2 interp_offsets(int i, int s, int x1, double alam) {
    const double dX = (double)(s * i);
    const double xc = alam * dX + (double)x1;
    const double xh = (double)(x1 + s * (i - 1)) + 0.5 * (double)s;
    double2 r;
    r.x = 2.0 * fabs(xc - xh);
    r.y = 1. - r.x;
    return r;
}

// One PHASE of one plane: the chunks [jbase, jbase + NW*P) of the plane's decreasing chunk sequence.
// Everything here is wave-uniform (SGPRs).
struct PhaseDesc {
    int n, jbase;
    uint32_t off;
    int cntA, cntT, nchunkA, nchunk;  // nchunk chunks in all; the first nchunkA of them hold rate-producing cells
    bool valid;
};
// QS: log2 of the sectors a wavefront works on side by side (0, or 2 in the four-quadrant kernel): a chunk of
// 64 lanes then holds 64 >> QS cells, each for 1 << QS quadrants.
template <int QS>
__device__ __forceinline__ void phase_load_plane(PhaseDesc &d, const PlaneInfo *planes, int nmax, bool rate_only) {
    d.valid = d.n <= nmax;
    if (d.valid) {
        const PlaneInfo pi = planes[d.n];
        d.off = pi.off;
        d.cntA = (int)pi.cntA;
        d.cntT = rate_only ? (int)pi.cntA : (int)pi.cntT;
        d.nchunkA = (d.cntA + (64 >> QS) - 1) >> (6 - QS);
        d.nchunk = (d.cntT + (64 >> QS) - 1) >> (6 - QS);
    }
}
template <int STEP, int QS>
__device__ __forceinline__ PhaseDesc phase_next(PhaseDesc d, const PlaneInfo *planes, int nmax, bool rate_only) {
    if (!d.valid) return d;
    d.jbase += STEP;
    if (d.jbase >= d.nchunk) {
        d.n++;
        d.jbase = 0;
        phase_load_plane<QS>(d, planes, nmax, rate_only);
    }
    return d;
}
// slot q of this wave in phase d: table index of the lane's cell (0 when the lane has none).  A plane's cells --
// list A (rate-producing) followed by list B -- are ONE sequence cut into 64-cell chunks, taken from the last
// chunk down; the chunk that straddles the two lists counts as rate-producing (its B lanes never fire).
// rates: wave-uniform, the chunk holds cells of list A; isA: this lane's cell is one of them.
// SP != 0 (staggered halves, see sweep_tab_kernel): the first NW/2 waves take the first NW/2 * SP chunks of the phase,
// the other half the rest (SP = chunks per wave per phase).
template <int NW, int SP>
__device__ __forceinline__ int slot_chunk(const PhaseDesc &d, int wave, int q) {  // wave-uniform
    if (SP) return d.jbase + (wave >= NW / 2 ? (NW / 2) * SP + (wave - NW / 2) : wave) + q * (NW / 2);
    return d.jbase + wave + q * NW;
}
template <int NW, int QS, int SP = 0>
__device__ __forceinline__ uint32_t slot_index(const PhaseDesc &d, int wave, int q, int lane, bool &valid, bool &isA,
                                               bool &rates) {
    const int jj = slot_chunk<NW, SP>(d, wave, q);
    const int c = d.nchunk - 1 - jj;
    const int i = (c * 64 + lane) >> QS;  // (the low QS bits of the lane number select the quadrant)
    rates = c < d.nchunkA;
    isA = i < d.cntA;
    valid = d.valid && (jj < d.nchunk) && (i < d.cntT);
    return (uint32_t)(valid ? i : 0);
}
template <int NW, int QS, int SP = 0>
__device__ __forceinline__ uint32_t slot_index(const PhaseDesc &d, int wave, int q, int lane, bool &valid, bool &isA) {
    bool rates;
    return slot_index<NW, QS, SP>(d, wave, q, lane, valid, isA, rates);
}

// The sweep is latency-bound (rocprofv3: waves wait 73 % of their cycles; an ablation without the
// f64 atomics runs 30 % faster, without barriers only 5 %), so the memory operations of a phase are
// issued one to two phases ahead, in plain C++ (the compiler keeps the s_waitcnt bookkeeping):
//     top of phase p : cell coordinates of phase p+1
//     phase p        : staging reads and interpolation of all slots; then geometry + neutral density of
//                      phase p+1 (their addresses need its cell coordinates, which have landed behind
//                      the interpolation); then the rates, whose table look-ups are the only point where
//                      the wave drains its memory counter
//     end of phase p : all f64 atomics of the phase in one burst (they complete behind the
//                      barriers and the next phase's interpolation), barrier, deferred staging
//                      writes, barrier.  The barriers wait for LDS traffic only.
template <bool LDS_STAGE>
__device__ __forceinline__ void phase_barrier() {
#if defined(ASORA_ABL_NOBARRIER)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ablation build: WRONG results
#else
    if (LDS_STAGE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();  // staging in global memory: stores must have completed
#endif
}

// THREADS: 256 (one wave per SIMD and workgroup, 4 workgroups per CU) or 1024 (ONE workgroup per CU that may use the
// whole LDS: same 16 waves per CU, staging arrays up to ~150 KB)
// QUAD: one workgroup per (source, pyramid) instead of per (source, sector): the four sign quadrants of a pyramid
// share the plane structure (same geometry variant -- checked on the host -- same cell lists), so lane l works on
// quadrant l & 3 of cell (chunk * 64 + l) >> 2.  Planes hold four times the cells, which is what small radii
// lack: their planes fill a fraction of a wavefront.  Per-quadrant: staging array, sign of the in-plane
// directions (two sets of row / column tables), ownership mask.  Same arithmetic per cell: bit-identical.
// QSHIFT = 3: the same with all eight sign octants of an axis (both pyramids) per workgroup, for the smallest radii;
// requires that the box limits bind on neither side (host check), so that both pyramids have the same planes.
// SSHIFT > 0 (with QUAD / OCT): the workgroup takes 1 << SSHIFT consecutive SOURCES of the (curve-sorted) list side
// by side as well -- lane l works on quadrant l & (NQ - 1) of source (l >> QSHIFT) & (NSRC - 1) of its group and on
// cell (chunk * 64 + l) >> (QSHIFT + SSHIFT).  All sources share the geometry tables (same N, R; box limits not
// binding), so the cell lists, chunks and phases are those of one source with NSRC times the lanes per cell: at
// R = 5 a plane of one source fills half a wavefront, eight sources fill it, and the per-task set-up (tables, plane
// list, log10 table, first loads) is paid once per group.  Per (source, quadrant): staging array; per source: row /
// column / interpolation tables (the interpolation carries the source's own coordinate), flux, position.
// SATSKIP: option "skip_saturated" compiled in (single-sector kernels only; a run-time test costs the default path 2 %).
template <bool LDS_STAGE, bool HEAT, int THREADS, bool WSTAGED, int QSHIFT = 0, int SSHIFT = 0, bool SATSKIP = false>
__global__ __launch_bounds__(THREADS)
__attribute__((amdgpu_waves_per_eu(HEAT ? ASORA_WAVES_PER_EU_HEAT : ASORA_WAVES_PER_EU, HEAT ? ASORA_WAVES_PER_EU_HEAT : ASORA_WAVES_PER_EU)))
void sweep_tab_kernel(SweepArgs A) {
    static_assert(!SATSKIP || QSHIFT == 0, "skip_saturated rides on the single-sector kernels");
    constexpr int P = ASORA_PHASE_CHUNKS, NW = THREADS / 64;
    constexpr bool QUAD = QSHIFT != 0, OCT = QSHIFT == 3, PACK = SSHIFT != 0;
    static_assert(QSHIFT == 0 || QSHIFT == 2 || QSHIFT == 3, "1, 4 or 8 sectors per workgroup");
    static_assert(SSHIFT == 0 || (QUAD && QSHIFT + SSHIFT <= 6), "source packing rides on the multi-sector kernels");
    constexpr int QS = QSHIFT + SSHIFT;  // log2 of the lanes that share a cell
    // STAG (large workgroups, one or two per CU): the two halves of the workgroup run half a phase apart -- while one
    // half interpolates (a latency-bound dependent chain per cell) the other computes its rates (throughput-bound):
    //     first half :  I(k)  | W R(k) | I(k+1)  | W R(k+1) | ...          I: staging reads + interpolation
    //     second half:        | I(k)   | W R(k)  | I(k+1)   | ...          W: staging writes, R: rates + atomics
    // ("|" = workgroup barrier; the second half passes one barrier before it starts, the first half one after it ends).
    // The first half's chunks of a phase precede the second half's in the plane's sequence, so the in-place order holds:
    // what a half reads while the other writes lies below the cells being written (see the barriers below).  At a
    // plane change each half passes one more barrier before its first interpolation -- the last writes of the plane
    // before are the other half's -- and the first half rebuilds the interpolation tables in that slot.
    constexpr bool STAG = LDS_STAGE && !QUAD && THREADS >= ASORA_STAGGER_MIN_THREADS;
    constexpr int SP = STAG ? P : 0;
    constexpr int NQ = 1 << QSHIFT, NSRC = 1 << SSHIFT, NS = QUAD ? 2 : 1;  // sectors, sources per wave; sign sets per table
    static_assert(!QUAD || LDS_STAGE, "the four-quadrant kernel stages in LDS");
    const double max_coldensh = (double)2e30f;  // raytracing.f90:368
    static_assert(LDS_STAGE || WSTAGED, "column-density-only staging is an LDS layout");
    PlaneCtx<LDS_STAGE, WSTAGED> C;
    C.stg.sig = A.sig;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int N = A.N;
    const uint32_t N2 = (uint32_t)N * (uint32_t)N;
    // LDS (in doubles): [cw pairs: 2*E, or column densities: E rounded up to even (LDS staging only)] [du: 2W] [dv: 2W] [rowU, colV: W ints each] [log10 table]
    const int E = A.cw_entries, W = A.wtab;
    const int SW1 = NS * W;     // entries per row / column table of ONE source (QUAD: the +1 direction set, then the -1 set)
    const int SW = NSRC * SW1;  // ... of the workgroup's sources
    const int off_du = LDS_STAGE ? NSRC * NQ * (WSTAGED ? 2 * E : (E + 1) & ~1) : 0;
    double2 *const du_tab = reinterpret_cast<double2 *>(asora_smem + off_du);
    double2 *const dv_tab = du_tab + SW;
    int *const rowU = reinterpret_cast<int *>(asora_smem + off_du + 4 * SW);
    int *const colV = rowU + SW;
    // the variant's plane table: read through LDS so that stepping to the next plane never touches vmcnt
    PlaneInfo *const planes = reinterpret_cast<PlaneInfo *>(asora_smem + off_du + 5 * SW + (SW & 1));
    // PACK: the 1-based in-plane coordinates (u, v) of the group's sources, for the per-plane interpolation tables
    int *const src_uv = reinterpret_cast<int *>(asora_smem + off_du + 5 * SW + (SW & 1) + plane_table_doubles(A.nplanes));
    const int off_log = off_du + 5 * SW + (SW & 1) + plane_table_doubles(A.nplanes) + (PACK ? NSRC : 0);
    C.stg.g = LDS_STAGE ? nullptr : reinterpret_cast<double2 *>(A.gscratch) + (size_t)blockIdx.x * (size_t)E;
    C.lg = load_log10_table(A.log10_table, off_log, tid, THREADS);
    // this lane's quadrant and source of the group: staging array, sign sets
    const int quad = QUAD ? (lane & (NQ - 1)) : 0;
    const int sgrp = PACK ? ((lane >> QSHIFT) & (NSRC - 1)) : 0;
    const uint32_t stg_q = (uint32_t)((sgrp * NQ + quad) * E);
    const uint32_t set_u = QUAD ? (uint32_t)(sgrp * SW1 + ((quad >> 1) & 1) * W) : 0u;
    const uint32_t set_v = QUAD ? (uint32_t)(sgrp * SW1 + (quad & 1) * W) : 0u;
    const double w_empty = weightf(0.0, A.sig);  // weight of a guard / never-written entry

    for (long long task = blockIdx.x; task < A.ntasks; task += gridDim.x) {
        constexpr int TASKS_PER_SOURCE = SECTORS / NQ;
        const int s = (int)(task / TASKS_PER_SOURCE);  // source, or (PACK) group of NSRC consecutive sources
        const int t = (int)(task - (long long)s * TASKS_PER_SOURCE) * NQ;  // first sector of this task
        const int src0 = A.src_begin + s * NSRC;       // first source of the group (wave-uniform)
        // this lane's source; lanes of a partial last group re-read the last source and never produce output
        const bool src_ok = !PACK || s * NSRC + sgrp < A.nsrc;
        const int src = PACK ? min(src0 + sgrp, A.src_begin + A.nsrc - 1) : src0;
        const int axis = t >> 3;  // 0: z-dominant, 1: y-dominant, 2: x-dominant
        const int sp = OCT ? ((quad & 4) ? -1 : 1) : ((t & 4) ? -1 : 1);  // OCT: per lane
        const int su = QUAD ? ((quad & 2) ? -1 : 1) : ((t & 2) ? -1 : 1);  // QUAD: per lane
        const int sv = QUAD ? ((quad & 1) ? -1 : 1) : ((t & 1) ? -1 : 1);
        const int i0r = A.src_pos[3 * src + 0], j0r = A.src_pos[3 * src + 1], k0r = A.src_pos[3 * src + 2];
        const int i0 = fmodulo(i0r, N), j0 = fmodulo(j0r, N), k0 = fmodulo(k0r, N);
        int p0, u0, v0, ur1, vr1;  // wrapped position (addressing); 1-based in-plane coordinates as given (arithmetic)
        if (axis == 0) { p0 = k0; u0 = i0; v0 = j0; ur1 = i0r + 1; vr1 = j0r + 1; }
        else if (axis == 1) { p0 = j0; u0 = i0; v0 = k0; ur1 = i0r + 1; vr1 = k0r + 1; }
        else { p0 = i0; u0 = j0; v0 = k0; ur1 = j0r + 1; vr1 = k0r + 1; }
        const double flux = A.src_flux[src];
        // z-dominant sectors work on the transposed copies [i][k][j], where their in-plane coordinate b (= j) is the
        // contiguous one -- unless the call runs on one layout (A.single: short calls, where the transposed copies and
        // their merge pass would cost more than the strided accesses of these sectors)
        const bool layoutB = axis == 0 && !A.single;
        const double *__restrict__ nhi = layoutB ? A.nhiB : A.nhiA;
        double *__restrict__ phi = layoutB ? A.phiB : A.phiA;
        double *__restrict__ heat = layoutB ? A.heatB : A.heatA;
        // column-density export: wave-uniform pointer when the group holds the exported source, per-lane flag
        const bool cdh_group = PACK ? (A.cdh_src >= src0 && A.cdh_src < src0 + NSRC) : (src == A.cdh_src);
        double *__restrict__ cdh = cdh_group ? (layoutB ? A.cdhB : A.cdhA) : nullptr;
        const bool cdh_lane = !PACK || src == A.cdh_src;
        const rsrc_t phi_rs = make_rsrc(phi, N2 * (uint32_t)N * 8u);
        const rsrc_t heat_rs = make_rsrc(HEAT ? heat : phi, N2 * (uint32_t)N * 8u);
        // ownership: reference tie order (z, y, x) and sign(0) = +1
        const uint32_t ghostmask = (su < 0 ? FLAG_A0 : 0u) | (sv < 0 ? FLAG_B0 : 0u) | (axis >= 1 ? FLAG_BN : 0u) |
                                   (axis == 2 ? FLAG_AN : 0u);
        // Cells outside the R-sphere (list B) are never upwind of a cell inside it: they only feed the exported
        // column density.  With "skip_unobserved" every other source sweeps list A only.
        const bool rate_only = A.rate_only && !cdh_group;
        // (OCT: the host has checked q_max <= min(last_l, last_r): both pyramids march the same planes)
        const int nmax = min(min(OCT ? min(A.last_r, A.last_l) : (sp > 0 ? A.last_r : A.last_l), A.q_max),
                             rate_only ? A.nmax_rate : 0x7fffffff);
        // cell = p * strideP + u * strideU + v * strideV: x-sectors (p, u, v) = (i, j, k), y-sectors (j, i, k), z-sectors
        // (k, i, j) -- in [i][k][j] unless A.single
        const uint32_t strideP = (axis == 2) ? N2 : (axis == 0 && A.single) ? 1u : (uint32_t)N;
        const uint32_t strideU = (axis == 2) ? (uint32_t)N : N2;
        const uint32_t strideV = (axis == 0 && A.single) ? (uint32_t)N : 1u;
        // (QUAD: the host has checked that the four quadrants share their variant)
        const int variant = QUAD ? A.vmap[axis != 0 ? 1 : 0]
                                 : A.vmap[(((su < 0) ? 2 : 0) + ((sv < 0) ? 1 : 0)) * 2 + (axis != 0 ? 1 : 0)];
        // (the previous task's last barrier guarantees nobody still reads the plane table)
        for (int i = tid; i < A.nplanes; i += THREADS) planes[i] = A.planes[(size_t)variant * A.nplanes + i];
        __syncthreads();
        const uint32_t pos00 = planes[0].off;  // staging position of cell (0, 0) in this variant's layout

        // phase descriptors of p and p+1, cell coordinates of p
        PhaseDesc d0, d1;
        d0.n = 1; d0.jbase = 0; d0.off = 0; d0.cntA = d0.cntT = d0.nchunkA = d0.nchunk = 0;
        phase_load_plane<QS>(d0, planes, nmax, rate_only);
        d1 = phase_next<NW * P, QS>(d0, planes, nmax, rate_only);
        uint2 meta_a[P], meta_b[P];  // (the two register sets of the phase pipeline, see the loop)
#pragma unroll
        for (int q = 0; q < P; q++) {
            bool v, ia;
            meta_a[q] = at_byte(A.cell_meta, (d0.off + slot_index<NW, QS, SP>(d0, wave, q, lane, v, ia)) * 8u);
        }

        // (the previous task's last barrier guarantees nobody still reads the arrays below)
        // entry i of a row / column table: offset i % W in direction +1 (first set) or, QUAD only, -1 (second set);
        // without QUAD the single set has the task's direction
        // (entry i: source i / SW1 of the group, then as above within that source's SW1 entries)
        auto table_dir = [&](int i, int s_task) { return QUAD ? ((PACK ? i % SW1 : i) >= W ? -1 : 1) : s_task; };
        auto table_off = [&](int i) { const int j = PACK ? i % SW1 : i; return j >= W ? j - W : j; };
        // in-plane source coordinates of table entry i's source: (wrapped u, wrapped v, 1-based u, 1-based v as given)
        auto table_src = [&](int i, int &tu0, int &tv0, int &tur1, int &tvr1) {
            if (!PACK) { tu0 = u0; tv0 = v0; tur1 = ur1; tvr1 = vr1; return; }
            const int sj = min(src0 + i / SW1, A.src_begin + A.nsrc - 1);
            const int ir = A.src_pos[3 * sj + 0], jr = A.src_pos[3 * sj + 1], kr = A.src_pos[3 * sj + 2];
            const int ur = axis == 2 ? jr : ir, vr = axis == 0 ? jr : kr;
            tu0 = fmodulo(ur, N); tv0 = fmodulo(vr, N); tur1 = ur + 1; tvr1 = vr + 1;
        };
        for (int i = tid; i < NSRC * NQ * E; i += THREADS) C.stg.st((uint32_t)i, 0.0, w_empty);
        for (int i = tid; i < SW; i += THREADS) {
            int tu0, tv0, tur1, tvr1;
            table_src(i, tu0, tv0, tur1, tvr1);
            const int o = table_off(i);
            rowU[i] = (int)((uint32_t)wrap(tu0 + table_dir(i, su) * o, N) * strideU);
            colV[i] = (int)((uint32_t)wrap(tv0 + table_dir(i, sv) * o, N) * strideV);
            if (PACK && i % SW1 == 0) { src_uv[2 * (i / SW1)] = tur1; src_uv[2 * (i / SW1) + 1] = tvr1; }
        }
        if (PACK) __syncthreads();  // src_uv
        // interpolation offsets of plane n for every row / column of every source of the group
        // ... of rows / columns 0 .. nlim only (plane n has none beyond min(n, W - 1)): entry i -> (table, sign set,
        // source, offset) with the offsets padded to a power of two, so that small planes cost one short pass
        auto fill_offsets = [&](double alam, int nlim, int nthreads = THREADS) {  // (threads 0 .. nthreads-1 call it)
            const int nl = min(nlim, W - 1);
            const int sh = 32 - __clz(nl);  // 1 << sh > nl
            constexpr int NSETS = NS * NSRC;
            for (int i = tid; i < (2 * NSETS) << sh; i += nthreads) {
                const int o = i & ((1 << sh) - 1), st = i >> sh;
                if (o > nl) continue;
                const bool is_u = st < NSETS;
                const int s2 = is_u ? st : st - NSETS;        // (source, sign set) of this table
                const int j = (s2 / NS) * SW1 + (s2 % NS) * W + o;
                const int c1 = PACK ? src_uv[2 * (s2 / NS) + (is_u ? 0 : 1)] : (is_u ? ur1 : vr1);
                if (is_u) du_tab[j] = interp_offsets(o, table_dir(j, su), c1, alam);
                else dv_tab[j] = interp_offsets(o, table_dir(j, sv), c1, alam);
            }
        };
        fill_offsets(cinterp_alam(1), 1);
        __syncthreads();
        if (tid < NQ * NSRC) {
            // source cell, raytracing.f90:430-439 (QUAD: one copy per quadrant's staging array)
            const uint32_t g0 = (uint32_t)p0 * strideP + (uint32_t)u0 * strideU + (uint32_t)v0 * strideV;
            double nHI = nhi[g0];
            double path = 0.5 * A.dr;
            double cdo = 0.0 + nHI * path;
            C.stg.st(pos00 + stg_q, cdo, weightf(cdo, A.sig));
            if (t == 0 && quad == 0 && src_ok) {  // (tid < NQ * NSRC <= 64: tid is the lane number)
                double vol_ph = A.dr * A.dr * A.dr;
                double ph, ht = 0.0;
                cell_rates<HEAT>(A, C.lg, flux, 0.0, cdo, vol_ph, nHI, ph, ht);
                unsafeAtomicAdd(phi + g0, ph);
                if (HEAT) unsafeAtomicAdd(heat + g0, ht);
                if (cdh && cdh_lane) cdh[g0] = cdo;
            }
        }
        __syncthreads();

        // geometry + neutral density of the first phase (rowU / colV are ready now)
        double2 geo_a[P], geo_b[P];
        double nhi_a[P], nhi_b[P];
        uint32_t g_a[P], g_b[P];
        {
            const uint32_t baseP = (uint32_t)wrap(p0 + sp * d0.n, N) * strideP;
#pragma unroll
            for (int q = 0; q < P; q++) {
                bool v, ia;
                const uint32_t idx = slot_index<NW, QS, SP>(d0, wave, q, lane, v, ia);
                const uint32_t a = meta_a[q].x & META_COORD_MASK, b = (meta_a[q].x >> META_COORD_BITS) & META_COORD_MASK;
                g_a[q] = baseP + (uint32_t)rowU[set_u + a] + (uint32_t)colV[set_v + b];
                geo_a[q] = at_byte(A.cell_geo, (d0.off + idx) * 16u);
                nhi_a[q] = at_byte(nhi, g_a[q] * 8u);
            }
        }

        const bool second_half = STAG && wave >= NW / 2;
        int plane_prev = d0.n;
        if (STAG && second_half) phase_barrier<LDS_STAGE>();  // half a phase behind
        // One phase: works the cells of phase d0 from the register set (meta_c, geo_c, nhi_c, g_c), loads the set of
        // phase d1 into (meta_n, geo_n, nhi_n, g_n), and steps d0 / d1.  The loop below calls it with the two sets in
        // alternating roles, so no register moves from phase to phase (18 per wave and phase otherwise).
        auto one_phase = [&](uint2 (&meta_c)[P], double2 (&geo_c)[P], double (&nhi_c)[P], uint32_t (&g_c)[P],
                             uint2 (&meta_n)[P], double2 (&geo_n)[P], double (&nhi_n)[P],
                             uint32_t (&g_n)[P]) __attribute__((always_inline)) {
            if (STAG && d0.n != plane_prev) {
                // first phase of a plane: the last writes of the plane before are the other half's and land in this slot;
                // the first half rebuilds the interpolation tables meanwhile (nobody interpolates the old plane any more)
                if (!second_half) fill_offsets(cinterp_alam(d0.n), d0.n, THREADS / 2);
                phase_barrier<LDS_STAGE>();
                plane_prev = d0.n;
            }
            // ---- cell coordinates of the next phase: in flight during this phase's interpolation ------
            // (unconditional -- past the last phase slot_index gives cell 0 of the last plane -- so that the
            // compiler counts the operations in flight exactly, see make_rsrc)
            uint32_t idx_n[P];
#pragma unroll
            for (int q = 0; q < P; q++) {
                bool v, ia;
                idx_n[q] = d1.off + slot_index<NW, QS, SP>(d1, wave, q, lane, v, ia);
                meta_n[q] = at_byte(A.cell_meta, idx_n[q] * 8u);
            }

            // ---- phase p: interpolation and rates from the old staged values ---------------------
            const int n = d0.n;
            double cdo[P], wo[P], ph[P], ht[HEAT ? P : 1], cin[P];
            int woff[P];
            bool fire[P], rate_slot[P];
            // one slot: 4-point interpolation, new column density and its weight (raytracing.f90:662-813)
            auto interp = [&](const int q, const bool plane1, const bool with_cdh) __attribute__((always_inline)) {
                bool valid, isA, rates;
                (void)slot_index<NW, QS, SP>(d0, wave, q, lane, valid, isA, rates);
                const uint2 m = meta_c[q];
                const uint32_t a = m.x & META_COORD_MASK, b = (m.x >> META_COORD_BITS) & META_COORD_MASK;
                const double2 du = du_tab[set_u + a], dv = dv_tab[set_v + b];
                const uint32_t o_up = (m.y & 0xffffu) + stg_q, o_cur = (m.y >> 16) + stg_q;
                const double2 e1 = C.stg.ld(o_up);       // (a-1, b-1)
                const double2 e3 = C.stg.ld(o_up + 1);   // (a-1, b)
                const double2 e2 = C.stg.ld(o_cur);      // (a,   b-1)
                const double2 e4 = C.stg.ld(o_cur + 1);  // (a,   b)
                // raytracing.f90:701-709 (weights read instead of recomputed)
                const double s1 = du.y * dv.y;
                const double s2 = dv.y * du.x;
                const double s3 = du.y * dv.x;
                const double s4 = du.x * dv.x;
                const double w1 = s1 * e1.y;
                const double w2 = s2 * e2.y;
                const double w3 = s3 * e3.y;
                const double w4 = s4 * e4.y;
                double cdh_in = div_exact(e1.x * w1 + e2.x * w2 + e3.x * w3 + e4.x * w4, w1 + w2 + w3 + w4);
                if (plane1) {  // the 20 diagonal nearest neighbours, raytracing.f90:703-708
                    if (a == 1 && b == 1) cdh_in = SQRT3_F32 * cdh_in;
                    else if (a == 1 || b == 1) cdh_in = SQRT2_F32 * cdh_in;
                }
                const double c_out = cdh_in + nhi_c[q] * geo_c[q].x;
                cin[q] = cdh_in;
                cdo[q] = c_out;
                wo[q] = WSTAGED ? weightf(c_out, A.sig) : 0.0;
                woff[q] = valid ? (int)o_cur + 1 : -1;
                const bool ghost = ((m.x >> META_FLAG_SHIFT) & ghostmask) != 0;
                const bool own = valid && !ghost && src_ok;
                if (with_cdh) {
                    if (cdh && own && cdh_lane) *ptr_byte(cdh, g_c[q] * 8u) = c_out;
                }
#ifdef ASORA_ABL_NORATES
                fire[q] = own && isA && cdh_in == 1.2345e300;  // ablation build: column densities only
                rate_slot[q] = false;
#else
                fire[q] = own && isA && !(cdh_in > max_coldensh);
                rate_slot[q] = rates;
#endif
            };
            // wave priority: low while interpolating, high from the rates to the end of the phase -- a wave close to
            // the barrier its workgroup waits at goes first (0.6 %)
            __builtin_amdgcn_s_setprio(0);
            const int nchunks = d0.nchunk;
#ifdef ASORA_ABL_GENERIC_PHASE
            const bool all_on = false;  // ablation build: every phase through the per-slot path
#else
            const bool all_on = slot_chunk<NW, SP>(d0, wave, P - 1) < nchunks && n != 1 && !cdh;
#endif
#if defined(ASORA_ABL_NORATES) || defined(ASORA_NO_RATE_BATCH)
            const bool all_rates = false;
#else
            // (slot 0 holds the wave's highest chunk; with the heating tables the batch would spill)
            // (... and so it does -- 20 to 37 registers -- in the packed kernels, whose lanes carry their own source)
            // ... and in the single-wave multi-sector kernels (R <~ 12), where batching costs 2 - 8 %)
            const bool all_rates = (!HEAT || ASORA_HEAT_BATCH) && !PACK && !(QUAD && THREADS == 64) && all_on &&
                                   d0.nchunk - 1 - slot_chunk<NW, SP>(d0, wave, 0) < d0.nchunkA;
#endif
            if (all_on) {
                // the usual phase: every slot of this wave is occupied -> ONE basic block, so that the scheduler
                // interleaves the slots' dependent chains (the column-density part is latency-bound)
#pragma unroll
                for (int q = 0; q < P; q++) {
                    ph[q] = 0.0;
                    if (HEAT) ht[q] = 0.0;
                    interp(q, false, false);
                }
            } else {
#pragma unroll
                for (int q = 0; q < P; q++) {
                    fire[q] = false;
                    rate_slot[q] = false;
                    ph[q] = 0.0;
                    if (HEAT) ht[q] = 0.0;
                    woff[q] = -1;
                    cdo[q] = 0.0;
                    wo[q] = 0.0;
                    cin[q] = 0.0;
                    if (slot_chunk<NW, SP>(d0, wave, q) < nchunks) interp(q, n == 1, true);  // wave-uniform
                }
            }
            // ---- geometry + neutral density of the next phase: in flight during this phase's rates ------
            {
                const uint32_t baseP = (uint32_t)wrap(p0 + sp * min(d1.n, nmax), N) * strideP;
#pragma unroll
                for (int q = 0; q < P; q++) {
                    const uint32_t a = meta_n[q].x & META_COORD_MASK, b = (meta_n[q].x >> META_COORD_BITS) & META_COORD_MASK;
                    g_n[q] = baseP + (uint32_t)rowU[set_u + a] + (uint32_t)colV[set_v + b];
                    geo_n[q] = at_byte(A.cell_geo, idx_n[q] * 16u);
                    nhi_n[q] = at_byte(nhi, g_n[q] * 8u);
                }
            }
            // Option "skip_saturated".  Behind an optical depth beyond the table's last entry both look-ups of a thick cell
            // clamp to that entry with residual 0, so phi_in == phi_out bit for bit and the cell's rate (and heating
            // rate) is exactly +0: adding it changes nothing.  Such lanes stop firing, and a slot none of whose lanes
            // fires skips its rates altogether (photorates.f90:113-125, 139-146: same values, no work).
            bool slot_live[P];
            // (Not in the multi-sector kernels of the small radii: their registers are spoken for, and a source's
            // light rarely saturates within ~45 cells.)
            if (SATSKIP) {
                const double tau_photo_limit = (double)1.0e-7f;
#pragma unroll
                for (int q = 0; q < P; q++) {
                    const double tau_in = cin[q] * A.sig, tau_out = cdo[q] * A.sig;
                    const bool sat = tau_in >= A.tau_sat && fabs(tau_out - tau_in) > tau_photo_limit;
                    fire[q] = fire[q] && !sat;
                    slot_live[q] = __builtin_amdgcn_ballot_w64(fire[q]) != 0;
                }
            } else {
#pragma unroll
                for (int q = 0; q < P; q++) slot_live[q] = true;
            }
            bool all_live = true;
#pragma unroll
            for (int q = 0; q < P; q++) all_live = all_live && slot_live[q];
            const bool batch = all_rates && all_live;
            if (STAG) {
                // Both halves have read what this half is about to overwrite: the other half's current interpolation
                // is of LATER chunks (second half) or started behind this half's previous barrier (first half) --
                // and what it reads meanwhile lies below this half's cells.
                phase_barrier<LDS_STAGE>();
#pragma unroll
                for (int q = 0; q < P; q++)
                    if (woff[q] >= 0) C.stg.st((uint32_t)woff[q], cdo[q], wo[q]);
            }
            __builtin_amdgcn_s_setprio(2);
            // rates: when every slot of the wave holds rate-producing cells, all of them together (their log10 /
            // look-up chains interleave); otherwise one slot at a time
            if (batch) {
                RatePrep rp[P];
                RateLoads rl[P];
#pragma unroll
                for (int q = 0; q < P; q++) rp[q] = rates_prepare(A, C.lg, flux, cin[q], cdo[q], geo_c[q].y);
#pragma unroll
                for (int q = 0; q < P; q++) rl[q] = rates_load<HEAT>(A, rp[q]);
#pragma unroll
                for (int q = 0; q < P; q++) {
                    double p_, h_ = 0.0;
                    rates_finish<HEAT>(A, rp[q], rl[q], nhi_c[q], p_, h_);
                    ph[q] = p_;
                    if (HEAT) ht[q] = h_;
                }
            } else {
#pragma unroll
                for (int q = 0; q < P; q++)
                    if (rate_slot[q] && slot_live[q]) {
                        const RatePrep rp = rates_prepare(A, C.lg, flux, cin[q], cdo[q], geo_c[q].y);
                        const RateLoads rl = rates_load<HEAT>(A, rp);
                        double p_, h_ = 0.0;
                        rates_finish<HEAT>(A, rp, rl, nhi_c[q], p_, h_);
                        ph[q] = p_;
                        if (HEAT) ht[q] = h_;
                    }
            }
            // ---- all atomics of the phase, after its last table look-up --------------------------
#pragma unroll
            for (int q = 0; q < P; q++) {
#ifdef ASORA_ABL_NOATOMIC
                const bool add = fire[q] && ph[q] == 1.2345e300;  // ablation build: rates computed, (almost) never added
#else
                const bool add = fire[q];
#endif
                // A rate of exactly zero -- behind an optical depth beyond the table's last entry both look-ups clamp to
                // that entry and phi_in - phi_out == +0 (photorates.f90:140-146) -- changes nothing when added: such lanes
                // are sent out of range too.  The f64 atomics execute at the memory side (~1.3 TB/s of added bytes
                // chip-wide, MI355X_MICROARCH.md): with the heating rate as second output they are what bounds the sweep.
#ifdef ASORA_ADD_ZEROS
                const int off = (int)(add ? g_c[q] * 8u : BUFFER_OOB);
                const int off_h = off;
#else
                const int off = (int)(add && ph[q] != 0.0 ? g_c[q] * 8u : BUFFER_OOB);
                const int off_h = (int)(add && (HEAT ? ht[q] : 0.0) != 0.0 ? g_c[q] * 8u : BUFFER_OOB);
#endif
                (void)llvm_amdgcn_raw_buffer_atomic_fadd_f64(ph[q], phi_rs, off, 0, 0);
                if (HEAT) (void)llvm_amdgcn_raw_buffer_atomic_fadd_f64(ht[q], heat_rs, off_h, 0, 0);
            }
            if (!STAG) {  // (STAG: written behind the barrier above; the tables follow at the top of the next plane)
                phase_barrier<LDS_STAGE>();  // every wave has read the old values it needs
#pragma unroll
                for (int q = 0; q < P; q++)
                    if (woff[q] >= 0) C.stg.st((uint32_t)woff[q], cdo[q], wo[q]);
                if (d1.n != n)  // last phase of this plane: interpolation offsets of the next one
                    fill_offsets(cinterp_alam(n + 1), n + 1);
            }
            // New values visible -- to the NEXT PLANE.  Within a plane the next phase takes lower chunks of the sequence:
            // it reads old values at positions below everything this phase writes (A cells read A cells only; inside a
            // list positions fall with the sequence) and writes only behind its own barrier, so with the staging in LDS
            // (one program order per wave) no second barrier is needed between the phases of a plane.
#ifndef ASORA_TWO_BARRIERS
            if (STAG || !LDS_STAGE || d1.n != n)
#endif
            phase_barrier<LDS_STAGE>();
            // ---- step the pipeline ------------------------------------------------------------------
            d0 = d1;
            d1 = phase_next<NW * P, QS>(d1, planes, nmax, rate_only);
        };
        // (Not for the unpacked eight-octant kernel: its registers are spoken for, the second copy of the phase spills
        // 30 - 40 of them -- R = 10 ... 12: 11 - 13 % slower; everywhere else +0.5 ... 2.7 %.)
#ifdef ASORA_NO_PHASE_UNROLL
        constexpr bool TWO_SETS = false;
#else
        constexpr bool TWO_SETS = !(OCT && !PACK);
#endif
        if (TWO_SETS) {
            for (;;) {
                if (!d0.valid) break;
                one_phase(meta_a, geo_a, nhi_a, g_a, meta_b, geo_b, nhi_b, g_b);
                if (!d0.valid) break;
                one_phase(meta_b, geo_b, nhi_b, g_b, meta_a, geo_a, nhi_a, g_a);
            }
        } else {
            while (d0.valid) {
                one_phase(meta_a, geo_a, nhi_a, g_a, meta_b, geo_b, nhi_b, g_b);
#pragma unroll
                for (int q = 0; q < P; q++) {
                    meta_a[q] = meta_b[q];
                    geo_a[q] = geo_b[q];
                    nhi_a[q] = nhi_b[q];
                    g_a[q] = g_b[q];
                }
            }
        }
        if (STAG && !second_half) phase_barrier<LDS_STAGE>();  // (the second half's last slot)
    }
}

// ------------------------------------------------------------------------------------------
// nhiA[i][j][k] = ndens*(1-xh) and its [i][k][j] transpose (raytracing.f90:421-422 arithmetic).
// grid (N/32, N/32, N), block (32, 8).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_nhi_kernel(const double *__restrict__ ndens,
                                                       const double *__restrict__ xh, double *__restrict__ nhiA,
                                                       double *__restrict__ nhiB, double *__restrict__ zeroA,
                                                       double *__restrict__ zeroB, int N) {
    // (zeroA / zeroB: the two rate accumulators, cleared in the same pass -- every cell of either layout is written
    // exactly once here.  Inputs are read once and the cleared grids are next touched by memory-side atomics:
    // non-temporal; the neutral density is what the sweep reads next: default policy.)
    __shared__ double tile[32][33];
    const size_t slab = (size_t)blockIdx.z * N * N;
    const int k = blockIdx.x * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j = blockIdx.y * 32 + r;
        if (j < N && k < N) {
            const size_t id = slab + (size_t)j * N + k;
            const double v = __builtin_nontemporal_load(ndens + id) * (1.0 - __builtin_nontemporal_load(xh + id));
            nhiA[id] = v;
            __builtin_nontemporal_store(0.0, zeroA + id);
            tile[r][threadIdx.x] = v;
        }
    }
    __syncthreads();
    const int j2 = blockIdx.y * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k2 = blockIdx.x * 32 + r;
        if (j2 < N && k2 < N) {
            const size_t id = slab + (size_t)k2 * N + j2;
            nhiB[id] = tile[threadIdx.x][r];
            __builtin_nontemporal_store(0.0, zeroB + id);
        }
    }
}

// One-layout calls: nhi = ndens*(1-xh) (raytracing.f90:421-422 arithmetic) and the zeroing of up to three output grids
// (rates, heating rates, column density: the sweep adds straight into them) in ONE streaming pass.  VEC: all pointers
// 16-byte aligned and the cell count even.
template <bool VEC>
__global__ __launch_bounds__(256) void prep_single_kernel(const double *__restrict__ ndens, const double *__restrict__ xh,
                                                          double *__restrict__ nhi, double *__restrict__ z0,
                                                          double *__restrict__ z1, double *__restrict__ z2, size_t cells) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VEC) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d *n2 = reinterpret_cast<const v2d *>(ndens), *x2 = reinterpret_cast<const v2d *>(xh);
        v2d *o2 = reinterpret_cast<v2d *>(nhi);
        const v2d zero = {0.0, 0.0};
        // (inputs are read once and the zeroed grids are next touched by memory-side atomics: non-temporal; the neutral
        // density is what the sweep reads next: default policy)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells / 2; i += stride) {
            const v2d n = __builtin_nontemporal_load(n2 + i), x = __builtin_nontemporal_load(x2 + i);
            o2[i] = n * (1.0 - x);
            __builtin_nontemporal_store(zero, reinterpret_cast<v2d *>(z0) + i);
            if (z1) __builtin_nontemporal_store(zero, reinterpret_cast<v2d *>(z1) + i);
            if (z2) __builtin_nontemporal_store(zero, reinterpret_cast<v2d *>(z2) + i);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
            nhi[i] = ndens[i] * (1.0 - xh[i]);
            z0[i] = 0.0;
            if (z1) z1[i] = 0.0;
            if (z2) z2[i] = 0.0;
        }
    }
}

// out[i][j][k] = A[i][j][k] + B[i][k][j]  (merge of the two accumulation layouts)
__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ A, const double *__restrict__ B,
                                                       double *__restrict__ out, int N) {
    __shared__ double tile[32][33];
    const size_t slab = (size_t)blockIdx.z * N * N;
    // read B[i][k][j] with j contiguous
    const int j = blockIdx.y * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = blockIdx.x * 32 + r;
        if (j < N && k < N) tile[r][threadIdx.x] = __builtin_nontemporal_load(B + slab + (size_t)k * N + j);
    }
    __syncthreads();
    const int k2 = blockIdx.x * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j2 = blockIdx.y * 32 + r;
        if (j2 < N && k2 < N) {
            const size_t id = slab + (size_t)j2 * N + k2;
            __builtin_nontemporal_store(__builtin_nontemporal_load(A + id) + tile[threadIdx.x][r], out + id);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Known-answer harness (replaces cinterp_gpu_kernel, src/asora/tests.cu:12-25): one thread per
// offset of a 4x4x4 block, gathers from a global grid and calls the SAME cinterp_sector.
// ------------------------------------------------------------------------------------------
__global__ void cinterp_test_kernel(double *out, const double *dens, int i0, int j0, int k0, int N,
                                    double sigma, int lo, int fortran_constants) {
    const int di = lo + (int)threadIdx.x, dj = lo + (int)threadIdx.y, dk = lo + (int)threadIdx.z;
    const int ai = abs(di), aj = abs(dj), ak = abs(dk);
    const int si = di >= 0 ? 1 : -1, sj = dj >= 0 ? 1 : -1, sk = dk >= 0 ? 1 : -1;
    const size_t N2 = (size_t)N * N;
    auto at = [&](int i, int j, int k) {
        return dens[(size_t)wrap(i % N, N) * N2 + (size_t)wrap(j % N, N) * N + wrap(k % N, N)];
    };
    const int i = i0 + di, j = j0 + dj, k = k0 + dk;
    double c1, c2, c3, c4, cd, path;
    int n, a, b, su, sv, u1, v1;
    if (ak >= aj && ak >= ai) {
        n = ak; a = ai; b = aj; su = si; sv = sj; u1 = i0 + 1; v1 = j0 + 1;
        c1 = at(i - si, j - sj, k - sk); c2 = at(i, j - sj, k - sk); c3 = at(i - si, j, k - sk); c4 = at(i, j, k - sk);
    } else if (aj >= ai && aj >= ak) {
        n = aj; a = ai; b = ak; su = si; sv = sk; u1 = i0 + 1; v1 = k0 + 1;
        c1 = at(i - si, j - sj, k - sk); c2 = at(i, j - sj, k - sk); c3 = at(i - si, j - sj, k); c4 = at(i, j - sj, k);
    } else {
        n = ai; a = aj; b = ak; su = sj; sv = sk; u1 = j0 + 1; v1 = k0 + 1;
        c1 = at(i - si, j - sj, k - sk); c2 = at(i - si, j, k - sk); c3 = at(i - si, j - sj, k); c4 = at(i - si, j, k);
    }
    if (n == 0) {
        cd = __builtin_nan("");
        path = __builtin_nan("");
    } else {
        cinterp_sector(n, a, b, su, sv, u1, v1, c1, c2, c3, c4, sigma, fortran_constants ? SQRT3_F32 : SQRT3_CUDA,
                       fortran_constants ? SQRT2_F32 : SQRT2_CUDA, cd, path);
    }
    const int idx = threadIdx.z + blockDim.z * (threadIdx.y + blockDim.y * threadIdx.x);
    out[2 * idx] = cd;
    out[2 * idx + 1] = path;
}

// Bit-for-bit check of div_exact / rcp_exact / div_const against the compiler's IEEE division.
// den[i] doubles as the constant divisor of div_const (its reciprocal is formed with `/`).
__global__ void division_test_kernel(const double *num, const double *den, long long n, unsigned long long *cnt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = num[i], d = den[i];
    const double q = a / d, r = 1.0 / d;
    if (__double_as_longlong(div_exact(a, d)) != __double_as_longlong(q)) atomicAdd(cnt + 0, 1ULL);
    if (__double_as_longlong(rcp_exact(d)) != __double_as_longlong(r)) atomicAdd(cnt + 1, 1ULL);
    if (__double_as_longlong(div_const(a, d, r)) != __double_as_longlong(q)) atomicAdd(cnt + 2, 1ULL);
}

}  // namespace asora
