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
struct PlaneInfo {
    uint32_t off, cntA, cntT, pad;
};

struct SweepArgs {
    const double *nhiA;  // nHI, [i][j][k]
    const double *nhiB;  // nHI, [i][k][j]
    double *phiA, *phiB;    // rate accumulators in the two layouts
    double *heatA, *heatB;  // heating accumulators (HEAT only)
    double *cdhA, *cdhB;    // column density of source `cdh_src` in the two layouts (may be null)
    const int *src_pos;     // 0-based, interleaved
    const double *src_flux;
    const double *thick, *thin, *hthick, *hthin;  // padded tables
    const double *log10_table;                    // LOG10_TABLE_DOUBLES doubles (asora_log10.hpp)
    double *gscratch;       // global staging (when LDS is too small), 2*plane_elems per block
    long long ntasks;       // 24 * number of sources of this launch
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
    const uint32_t *cell_ab;   // (a << 16) | b, list A (rate-producing) then list B per plane
    const double *cell_pathdr; // path * dr
    const double *cell_vol;    // vol_ph = dist2 * (path*dr) * 4pi
    int nplanes;
    int vmap[8];               // ((u_is_neg*2 + v_is_neg)*2 + (axis != 0)) -> variant
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
extern __shared__ double asora_smem[];
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
        const int i0 = A.src_pos[3 * src + 0], j0 = A.src_pos[3 * src + 1], k0 = A.src_pos[3 * src + 2];
        const double flux = A.src_flux[src];
        int p0, u0, v0;
        if (axis == 0) { p0 = k0; u0 = i0; v0 = j0; }
        else if (axis == 1) { p0 = j0; u0 = i0; v0 = k0; }
        else { p0 = i0; u0 = j0; v0 = k0; }
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
                cinterp_sector(n, a, b, su, sv, u0 + 1, v0 + 1, c1, c2, c3, c4, A.sig, SQRT3_F32, SQRT2_F32,
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
// everything about a cell that depends only on its offset (n, a, b) from the source -- membership
// in the octahedron/box, the oracle's exact R-sphere predicate, path*dr and vol_ph -- is computed
// once per call on the host with the same IEEE operations (build_geometry in asora_capi.hip) and
// streamed from L2.  Per plane the cells come as two dense lists: A (inside the R sphere: column
// density + rates) and B (outside: column density only), cut into 64-cell chunks, so wavefronts
// are fully populated and never diverge on the sphere or the octahedron cut.
//
// ONE plane of column densities is staged (in LDS), updated IN PLACE: occupancy on this kernel is
// worth more than anything else (measured: 3 -> 2 workgroups per CU costs 29 %), and a single
// plane halves the LDS footprint.  In-place is safe because of the processing order.  A plane is
// worked off in PHASES of up to NW*P chunks; within a phase every wave first computes its cells
// from the old values (results stay in registers), then -- after a barrier -- all waves write.
// Across phases chunks are taken in DECREASING order: list B from its last chunk down, then list A
// from its last chunk down.  Cell (a, b) reads the old values at (a-1,b-1), (a,b-1), (a-1,b), (a,b):
// all at row-major positions <= its own, and for an A cell all inside list A (the sphere's row
// extent shrinks with a), so nothing it reads has been overwritten by an earlier phase.
// Positions a plane does not write (beyond the octahedron cut) are never read later with
// non-zero weight; never-written positions and the guard row/column read as 0.
// ------------------------------------------------------------------------------------------
#ifndef ASORA_GROUP
#define ASORA_GROUP 1
#endif
#ifndef ASORA_PHASE_CHUNKS
#define ASORA_PHASE_CHUNKS 4
#endif

// Everything the cells of one plane of one task share.
template <bool LDS_STAGE>
struct PlaneCtx {
    Stage<LDS_STAGE> stg;
    Log10LDS lg;
    const double *__restrict__ nhi;
    double *__restrict__ phi;
    double *__restrict__ heat;
    double *__restrict__ cdh;  // null unless this task exports its column density
    const uint32_t *__restrict__ cab;  // tables of the current plane
    const double *__restrict__ cpd;
    const double *__restrict__ cvol;
    uint32_t baseP, strideU;
    double alam, flux;
    int n, S, N, u0, v0, su, sv, axis;
    int cntA, cntT, nchunkA, nchunkB;
};

// G chunks of one wave (sequence positions j, j + NW, ...): all table loads, then all nHI loads and
// LDS reads, then the arithmetic.  Three dependent memory round trips per GROUP instead of per
// cell.  Results (new column density + LDS position) are returned for the deferred write.
template <int G, bool HEAT, bool LDS_STAGE>
__device__ __forceinline__ void process_group(const SweepArgs &A, const PlaneCtx<LDS_STAGE> &C, int j, int lane,
                                              double *cdo_out, int *woff_out) {
    constexpr int NW = SWEEP_THREADS / 64;
    const double max_coldensh = (double)2e30f;  // raytracing.f90:368
    const int nchunk = C.nchunkA + C.nchunkB;
    bool valid[G], isA[G];
    uint32_t idx[G], ab[G], g[G];
    double pathdr[G], vol[G], nHI[G], c1[G], c2[G], c3[G], c4[G];
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int jj = j + k * NW;  // wave-uniform
        // decreasing order: B chunks from the last down, then A chunks from the last down
        isA[k] = jj >= C.nchunkB;
        const int c = isA[k] ? C.nchunkA - 1 - (jj - C.nchunkB) : C.nchunkB - 1 - jj;
        const int i = isA[k] ? c * 64 + lane : C.cntA + c * 64 + lane;
        const int lim = isA[k] ? C.cntA : C.cntT;
        valid[k] = (jj < nchunk) && (i < lim);
        idx[k] = (uint32_t)(valid[k] ? i : 0);
        ab[k] = C.cab[idx[k]];
        pathdr[k] = C.cpd[idx[k]];
        vol[k] = C.cvol[idx[k]];
    }
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int a = (int)(ab[k] >> 16), b = (int)(ab[k] & 0xffffu);
        const int uw = wrap(C.u0 + C.su * a, C.N), vw = wrap(C.v0 + C.sv * b, C.N);
        g[k] = C.baseP + (uint32_t)uw * C.strideU + (uint32_t)vw;
        nHI[k] = C.nhi[g[k]];
        const int o = a * C.S + b;
        c1[k] = C.stg.ld(o);
        c3[k] = C.stg.ld(o + 1);
        c2[k] = C.stg.ld(o + C.S);
        c4[k] = C.stg.ld(o + C.S + 1);
    }
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int a = (int)(ab[k] >> 16), b = (int)(ab[k] & 0xffffu);
        const double cdh_in = cinterp_cd(C.n, a, b, C.su, C.sv, C.u0 + 1, C.v0 + 1, C.alam, c1[k], c2[k], c3[k], c4[k],
                                         A.sig, SQRT3_F32, SQRT2_F32);
        const double cdo = cdh_in + nHI[k] * pathdr[k];
        // ownership: reference tie order (z, y, x) and sign(0) = +1
        const bool ghost = (C.su < 0 && a == 0) || (C.sv < 0 && b == 0) || (C.axis >= 1 && b == C.n) ||
                           (C.axis == 2 && a == C.n);
        cdo_out[k] = cdo;
        woff_out[k] = valid[k] ? (a + 1) * C.S + b + 1 : -1;
        if (valid[k] && !ghost) {
            if (C.cdh) C.cdh[g[k]] = cdo;
            if (isA[k]) {
#ifdef ASORA_ABL_NORATES
                if (cdh_in == 1.2345e300) {  // ablation build: column densities only
#else
                if (!(cdh_in > max_coldensh)) {
#endif
                    double ph, ht = 0.0;
                    cell_rates<HEAT>(A, C.lg, C.flux, cdh_in, cdo, vol[k], nHI[k], ph, ht);
#ifdef ASORA_ABL_NOATOMIC
                    if (ph == 1.2345e300)  // ablation build: rates computed, (almost) never added
#endif
                    unsafeAtomicAdd(C.phi + g[k], ph);
                    if (HEAT) unsafeAtomicAdd(C.heat + g[k], ht);
                }
            }
        }
    }
}

// register budget: 5 waves per SIMD (<= 96 VGPRs; measured faster than 4 or 6 on the 256^3 / R = 60 sweep)
#ifndef ASORA_WAVES_PER_EU
#define ASORA_WAVES_PER_EU 5
#endif
template <bool LDS_STAGE, bool HEAT>
__global__ __launch_bounds__(SWEEP_THREADS) __attribute__((amdgpu_waves_per_eu(ASORA_WAVES_PER_EU, ASORA_WAVES_PER_EU)))
void sweep_tab_kernel(SweepArgs A) {
    constexpr int G = ASORA_GROUP, P = ASORA_PHASE_CHUNKS, NW = SWEEP_THREADS / 64;
    static_assert(P % G == 0, "a phase is a whole number of groups");
    PlaneCtx<LDS_STAGE> C;
    C.stg.g = LDS_STAGE ? nullptr : A.gscratch + (size_t)blockIdx.x * (size_t)A.plane_elems;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int S = A.S;
    const int plane = A.plane_elems;
    const int N = A.N;
    const uint32_t N2 = (uint32_t)N * (uint32_t)N;
    C.lg = load_log10_table(A.log10_table, LDS_STAGE ? plane : 0, tid, SWEEP_THREADS);
    C.S = S;
    C.N = N;

    for (long long task = blockIdx.x; task < A.ntasks; task += gridDim.x) {
        const int s = (int)(task / SECTORS);
        const int t = (int)(task - (long long)s * SECTORS);
        const int src = A.src_begin + s;
        const int axis = t >> 3;  // 0: z-dominant, 1: y-dominant, 2: x-dominant
        const int sp = (t & 4) ? -1 : 1;
        const int su = (t & 2) ? -1 : 1;
        const int sv = (t & 1) ? -1 : 1;
        const int i0 = A.src_pos[3 * src + 0], j0 = A.src_pos[3 * src + 1], k0 = A.src_pos[3 * src + 2];
        int p0, u0, v0;
        if (axis == 0) { p0 = k0; u0 = i0; v0 = j0; }
        else if (axis == 1) { p0 = j0; u0 = i0; v0 = k0; }
        else { p0 = i0; u0 = j0; v0 = k0; }
        C.flux = A.src_flux[src];
        C.nhi = (axis == 0) ? A.nhiB : A.nhiA;
        C.phi = (axis == 0) ? A.phiB : A.phiA;
        C.heat = (axis == 0) ? A.heatB : A.heatA;
        C.cdh = (src == A.cdh_src) ? ((axis == 0) ? A.cdhB : A.cdhA) : nullptr;
        C.u0 = u0; C.v0 = v0; C.su = su; C.sv = sv; C.axis = axis;
        const int nmax = min(sp > 0 ? A.last_r : A.last_l, A.q_max);
        const uint32_t strideP = (axis == 2) ? N2 : (uint32_t)N;
        C.strideU = (axis == 2) ? (uint32_t)N : N2;
        const int variant = A.vmap[(((su < 0) ? 2 : 0) + ((sv < 0) ? 1 : 0)) * 2 + (axis != 0 ? 1 : 0)];
        const PlaneInfo *__restrict__ planes = A.planes + (size_t)variant * A.nplanes;

        // guard cells and never-written cells must read as 0 (their weights are 0 or O(eps))
        for (int i = tid; i < plane; i += SWEEP_THREADS) C.stg.st(i, 0.0);
        __syncthreads();
        if (tid == 0) {
            // source cell, raytracing.f90:430-439
            const uint32_t g0 = (uint32_t)p0 * strideP + (uint32_t)u0 * C.strideU + (uint32_t)v0;
            double nHI = C.nhi[g0];
            double path = 0.5 * A.dr;
            double cdo = 0.0 + nHI * path;
            C.stg.st(S + 1, cdo);
            if (t == 0) {
                double vol_ph = A.dr * A.dr * A.dr;
                double ph, ht = 0.0;
                cell_rates<HEAT>(A, C.lg, C.flux, 0.0, cdo, vol_ph, nHI, ph, ht);
                unsafeAtomicAdd(C.phi + g0, ph);
                if (HEAT) unsafeAtomicAdd(C.heat + g0, ht);
                if (C.cdh) C.cdh[g0] = cdo;
            }
        }
        __syncthreads();

        for (int n = 1; n <= nmax; n++) {
            const PlaneInfo pi = planes[n];
            C.n = n;
            C.alam = cinterp_alam(n);
            C.baseP = (uint32_t)wrap(p0 + sp * n, N) * strideP;
            C.cab = A.cell_ab + pi.off;
            C.cpd = A.cell_pathdr + pi.off;
            C.cvol = A.cell_vol + pi.off;
            C.cntA = (int)pi.cntA;
            C.cntT = (int)pi.cntT;
            C.nchunkA = (C.cntA + 63) >> 6;
            C.nchunkB = (C.cntT - C.cntA + 63) >> 6;
            const int nchunk = C.nchunkA + C.nchunkB;
            for (int jbase = 0; jbase < nchunk; jbase += NW * P) {
                double cdo[P];
                int woff[P];
#pragma unroll
                for (int q = 0; q < P; q += G) {
                    const int j = jbase + wave + q * NW;
                    if (j < nchunk) {
                        process_group<G, HEAT, LDS_STAGE>(A, C, j, lane, cdo + q, woff + q);
                    } else {
#pragma unroll
                        for (int k = 0; k < G; k++) { cdo[q + k] = 0.0; woff[q + k] = -1; }
                    }
                }
                __syncthreads();  // every wave has read the old values it needs
#pragma unroll
                for (int q = 0; q < P; q++)
                    if (woff[q] >= 0) C.stg.st(woff[q], cdo[q]);
                __syncthreads();  // new values visible
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// nhiA[i][j][k] = ndens*(1-xh) and its [i][k][j] transpose (raytracing.f90:421-422 arithmetic).
// grid (N/32, N/32, N), block (32, 8).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_nhi_kernel(const double *__restrict__ ndens,
                                                       const double *__restrict__ xh, double *__restrict__ nhiA,
                                                       double *__restrict__ nhiB, int N) {
    __shared__ double tile[32][33];
    const size_t slab = (size_t)blockIdx.z * N * N;
    const int k = blockIdx.x * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j = blockIdx.y * 32 + r;
        if (j < N && k < N) {
            const size_t id = slab + (size_t)j * N + k;
            const double v = ndens[id] * (1.0 - xh[id]);
            nhiA[id] = v;
            tile[r][threadIdx.x] = v;
        }
    }
    __syncthreads();
    const int j2 = blockIdx.y * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k2 = blockIdx.x * 32 + r;
        if (j2 < N && k2 < N) nhiB[slab + (size_t)k2 * N + j2] = tile[threadIdx.x][r];
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
        if (j < N && k < N) tile[r][threadIdx.x] = B[slab + (size_t)k * N + j];
    }
    __syncthreads();
    const int k2 = blockIdx.x * 32 + threadIdx.x;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j2 = blockIdx.y * 32 + r;
        if (j2 < N && k2 < N) {
            const size_t id = slab + (size_t)j2 * N + k2;
            out[id] = A[id] + tile[threadIdx.x][r];
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
