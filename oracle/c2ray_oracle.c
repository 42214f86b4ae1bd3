/*
 * c2ray_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded restatement of the reference's Fortran CPU path
 *   raytracing.do_all_sources -> do_source -> evolve2D -> evolve0D -> cinterp
 *   photorates.photoion_rates / photo_lookuptable
 *   chemistry.global_pass -> evolve0D_global -> do_chemistry -> doric
 * written from scratch for this repository.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker / reported
 * CPU baseline -- never as a fallback of the HIP path.
 *
 * PINNING: this restatement is pinned bit-for-bit (see tests/test_oracle.py::test_port_vs_compiled_reference*
 * and tests/golden/make_golden.py) against the reference Fortran itself, compiled in the
 * build container by oracle/Makefile (`make ref`) from the sources where they lie under
 * /root/reference, and against the reference's own fixtures tests/data/cinterp_output_*.npy
 * and photo_ionization_rate.npy (tests/test_reference_fixtures.py, on the committed copies
 * tests/golden/ref_*.npy; at the level SURVEY.md section 3.4 explains).
 *
 * Every function cites the reference file:line it follows.  Arithmetic notes that matter
 * for bit-parity (all verified against the flang build):
 *   - Fortran default-real (float32) literals/conversions are kept as float:
 *     sqrt(3.0), sqrt(2.0), 2e30, 1.0e-7, 1.0e-3, 1.0e-8, real(int).
 *   - no FMA contraction (compile with -ffp-contract=off), left-to-right evaluation.
 *   - libm log10/exp/pow/sqrt of the host (same glibc the flang build calls).
 *
 * Array layout: all grids are C-ordered [i][j][k] (k fastest), 0-based in memory, i.e. the
 * layout libasora uses (src/asora/raytracing.cu:38-40).  The Fortran reference indexes the
 * same logical (i,j,k) in column-major order; traversal order and arithmetic are
 * layout-independent.  Source positions are 1-BASED like the Fortran (raytracing.f90:415-417).
 *
 * Documented deviations from the reference (places where the reference reads undefined data):
 *   D1 photo_lookuptable can index one element past the table (photorates.f90:140-146,
 *      SURVEY 3.4 #9); here that element is taken equal to the last one.
 *   D2 phi_ion_out / phi_heat_p are uninitialised in the reference when the radiative
 *      transfer is stopped (raytracing.f90:495-532); here they are 0.
 *   D3 flux_mode: 0 reproduces the reference's use of normflux(NumSrc) for every source
 *      (raytracing.f90:500,503); 1 uses normflux(ns) (what ASORA does,
 *      src/asora/raytracing.cu:206).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Test-only variant (oracle/Makefile `port_devlog10`, -DORACLE_LOG10_DEVICE): the lookup's log10 is the host build
 * of the HIP kernels' table-driven log10 (oracle/log10_device.cpp) instead of glibc's.  Nothing else changes.  It
 * isolates libm-to-libm variation of log10 as the cause of the cells tests/parity.py's conditioning term admits. */
#ifdef ORACLE_LOG10_DEVICE
double c2ray_log10_device(double x);
#define LOOKUP_LOG10(x) c2ray_log10_device(x)
int c2ray_oracle_log10_kind(void) { return 1; }
#else
#define LOOKUP_LOG10(x) log10(x)
int c2ray_oracle_log10_kind(void) { return 0; }
#endif

#define IDX(i, j, k, m2, m3) (((size_t)(i) * (size_t)(m2) + (size_t)(j)) * (size_t)(m3) + (size_t)(k))

static const double PI_D = 3.14159265358979323846264338; /* raytracing.f90:47 */
static const double S_STAR = 1.0e48;                     /* photorates.f90:7 */

/* Fortran modulo(a, b) for b > 0 (result in [0, b)). */
static inline int f_modulo(int a, int b) {
    int r = a % b;
    return r < 0 ? r + b : r;
}
static inline int f_sign1(int x) { return x >= 0 ? 1 : -1; } /* sign(1, x) */
static inline int i_abs(int x) { return x < 0 ? -x : x; }
static inline int i_min(int a, int b) { return a < b ? a : b; }
static inline int i_max(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * photorates.f90:130-147  photo_lookuptable (internal function of photoion_rates)
 * ---------------------------------------------------------------------------------------- */
static double photo_lookuptable(double tau, const double *table, double minlogtau, double dlogtau,
                                int NumTau) {
    double logtau = LOOKUP_LOG10(fmax(1.0e-20, tau));
    /* real(NumTau) is default real */
    double real_i = fmin((double)(float)NumTau, fmax(0.0, 1.0 + (logtau - minlogtau) / dlogtau));
    int i0 = (int)real_i;
    int i1 = i_min(NumTau, i0 + 1);
    double residual = real_i - (double)(float)i0;
    /* table(i0+1) with 1-based Fortran indexing == table[i0]; D1: clamp the overrun */
    int a = i_min(i0, NumTau - 1), b = i_min(i1, NumTau - 1);
    return table[a] + residual * (table[b] - table[a]);
}

/* Conditioning probe (NOT part of the reference): the same lookup with the libm log10 result moved by
 * `shift` ulps.  Used to measure how much a rate depends on the last bit of log10, i.e. how far two
 * correct implementations with different libm's may legitimately differ in that cell. */
static double photo_lookuptable_shift(double tau, const double *table, double minlogtau, double dlogtau,
                                      int NumTau, int shift) {
    double logtau = LOOKUP_LOG10(fmax(1.0e-20, tau));
    for (int s = 0; s < (shift < 0 ? -shift : shift); s++) logtau = nextafter(logtau, shift < 0 ? -INFINITY : INFINITY);
    double real_i = fmin((double)(float)NumTau, fmax(0.0, 1.0 + (logtau - minlogtau) / dlogtau));
    int i0 = (int)real_i;
    int i1 = i_min(NumTau, i0 + 1);
    double residual = real_i - (double)(float)i0;
    int a = i_min(i0, NumTau - 1), b = i_min(i1, NumTau - 1);
    return table[a] + residual * (table[b] - table[a]);
}

static double lookup_spread(double tau, const double *table, double minlogtau, double dlogtau, int NumTau) {
    double c = photo_lookuptable_shift(tau, table, minlogtau, dlogtau, NumTau, 0);
    double p = photo_lookuptable_shift(tau, table, minlogtau, dlogtau, NumTau, 1);
    double m = photo_lookuptable_shift(tau, table, minlogtau, dlogtau, NumTau, -1);
    return fmax(fabs(p - c), fabs(m - c));
}

/* |d rate| for a +-1 ulp change of the log10 calls inside photoion_rates (before the division by nHI) */
void c2ray_photoion_rates_sens(double normflux, double coldens_in, double coldens_out, double Vfact, double sig,
                               double *sens_photo, double *sens_heat, const double *photo_thin_table,
                               const double *photo_thick_table, const double *heat_thin_table,
                               const double *heat_thick_table, double minlogtau, double dlogtau, int NumTau) {
    const double tau_photo_limit = (double)1.0e-7f;
    double tau_in = coldens_in * sig, tau_out = coldens_out * sig, prefact = normflux / Vfact;
    if (fabs(tau_out - tau_in) > tau_photo_limit) {
        *sens_photo = fabs(prefact) * (lookup_spread(tau_in, photo_thick_table, minlogtau, dlogtau, NumTau) +
                                       lookup_spread(tau_out, photo_thick_table, minlogtau, dlogtau, NumTau));
        *sens_heat = fabs(prefact) * (lookup_spread(tau_in, heat_thick_table, minlogtau, dlogtau, NumTau) +
                                      lookup_spread(tau_out, heat_thick_table, minlogtau, dlogtau, NumTau));
    } else {
        *sens_photo = fabs(prefact * (tau_out - tau_in)) * lookup_spread(tau_in, photo_thin_table, minlogtau, dlogtau, NumTau);
        *sens_heat = fabs(prefact * (tau_out - tau_in)) * lookup_spread(tau_in, heat_thin_table, minlogtau, dlogtau, NumTau);
    }
}

/* ------------------------------------------------------------------------------------------
 * photorates.f90:62-149  photoion_rates
 * ---------------------------------------------------------------------------------------- */
void c2ray_photoion_rates(double normflux, double coldens_in, double coldens_out, double Vfact,
                          double sig, double *phi_photo_cell, double *phi_photo_out,
                          double *phi_heat_cell, const double *photo_thin_table,
                          const double *photo_thick_table, const double *heat_thin_table,
                          const double *heat_thick_table, double minlogtau, double dlogtau,
                          int NumTau) {
    const double tau_photo_limit = (double)1.0e-7f; /* photorates.f90:69, default-real literal */
    double tau_in = coldens_in * sig;
    double tau_out = coldens_out * sig;
    double prefact = normflux / Vfact;
    double phi_photo_in = prefact * photo_lookuptable(tau_in, photo_thick_table, minlogtau, dlogtau, NumTau);
    if (fabs(tau_out - tau_in) > tau_photo_limit) {
        *phi_photo_out = prefact * photo_lookuptable(tau_out, photo_thick_table, minlogtau, dlogtau, NumTau);
        *phi_photo_cell = phi_photo_in - *phi_photo_out;
        *phi_heat_cell = prefact * (photo_lookuptable(tau_in, heat_thick_table, minlogtau, dlogtau, NumTau) -
                                    photo_lookuptable(tau_out, heat_thick_table, minlogtau, dlogtau, NumTau));
    } else {
        *phi_photo_cell = prefact * (tau_out - tau_in) *
                          photo_lookuptable(tau_in, photo_thin_table, minlogtau, dlogtau, NumTau);
        *phi_photo_out = phi_photo_in - *phi_photo_cell;
        *phi_heat_cell = prefact * (tau_out - tau_in) *
                         photo_lookuptable(tau_in, heat_thin_table, minlogtau, dlogtau, NumTau);
    }
}

/* ------------------------------------------------------------------------------------------
 * photorates.f90:13-57  photoion_rates_test (grey analytic rates, -DGREY_NOTABLES only)
 * ---------------------------------------------------------------------------------------- */
void c2ray_photoion_rates_test(double normflux, double coldens_in, double coldens_out, double Vfact,
                               double nHI, double sig, double *phi_photo_cell, double *phi_photo_out) {
    (void)nHI;
    const double tau_photo_limit = (double)1.0e-7f;
    double tau_in = coldens_in * sig;
    double tau_out = coldens_out * sig;
    double prefact = normflux * S_STAR / Vfact;
    double phi_photo_in = prefact * exp(-tau_in);
    if (fabs(tau_out - tau_in) > tau_photo_limit) {
        *phi_photo_out = prefact * exp(-tau_out);
        *phi_photo_cell = phi_photo_in - *phi_photo_out;
    } else {
        *phi_photo_cell = prefact * (tau_out - tau_in) * exp(-tau_in);
        *phi_photo_out = phi_photo_in - *phi_photo_cell;
    }
}

/* raytracing.f90:807-813 */
static inline double weightf(double cd, double sigma) { return 1.0 / fmax(0.6, cd * sigma); }

/* ------------------------------------------------------------------------------------------
 * raytracing.f90:576-815  cinterp.  pos / srcpos are 1-based, pos is NOT wrapped.
 * ---------------------------------------------------------------------------------------- */
void c2ray_cinterp(const int *pos, const int *srcpos, double *cdensi, double *path,
                   const double *cdh, double sigma, int m1, int m2, int m3) {
    const double sqrt3 = (double)sqrtf(3.0f); /* raytracing.f90:608-609: default-real sqrt */
    const double sqrt2 = (double)sqrtf(2.0f);
    int i = pos[0], j = pos[1], k = pos[2];
    int i0 = srcpos[0], j0 = srcpos[1], k0 = srcpos[2];
    int idel = i - i0, jdel = j - j0, kdel = k - k0;
    int idela = i_abs(idel), jdela = i_abs(jdel), kdela = i_abs(kdel);
    int sgni = f_sign1(idel), sgnj = f_sign1(jdel), sgnk = f_sign1(kdel);
    int im = i - sgni, jm = j - sgnj, km = k - sgnk;
    double di = (double)(float)idel, dj = (double)(float)jdel, dk = (double)(float)kdel;
    double alam, xc, yc, zc, dx, dy, dz, s1, s2, s3, s4, c1, c2, c3, c4, w1, w2, w3, w4;

#define CD(ii, jj, kk) cdh[IDX((ii)-1, (jj)-1, (kk)-1, m2, m3)]
    if (kdela >= jdela && kdela >= idela) {
        /* z-plane crossing, raytracing.f90:662-710 */
        alam = (double)((float)(km - k0) + (float)sgnk * 0.5f) / dk;
        xc = alam * di + (double)(float)i0;
        yc = alam * dj + (double)(float)j0;
        dx = 2.0 * fabs(xc - (double)((float)im + 0.5f * (float)sgni));
        dy = 2.0 * fabs(yc - (double)((float)jm + 0.5f * (float)sgnj));
        s1 = (1. - dx) * (1. - dy);
        s2 = (1. - dy) * dx;
        s3 = (1. - dx) * dy;
        s4 = dx * dy;
        int ip = f_modulo(i - 1, m1) + 1, imp = f_modulo(im - 1, m1) + 1;
        int jp = f_modulo(j - 1, m2) + 1, jmp = f_modulo(jm - 1, m2) + 1;
        int kmp = f_modulo(km - 1, m3) + 1;
        c1 = CD(imp, jmp, kmp);
        c2 = CD(ip, jmp, kmp);
        c3 = CD(imp, jp, kmp);
        c4 = CD(ip, jp, kmp);
        w1 = s1 * weightf(c1, sigma);
        w2 = s2 * weightf(c2, sigma);
        w3 = s3 * weightf(c3, sigma);
        w4 = s4 * weightf(c4, sigma);
        *cdensi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
        if (kdela == 1 && (idela == 1 || jdela == 1)) {
            if (idela == 1 && jdela == 1) *cdensi = sqrt3 * *cdensi;
            else *cdensi = sqrt2 * *cdensi;
        }
        *path = sqrt((di * di + dj * dj) / (dk * dk) + 1.0);
    } else if (jdela >= idela && jdela >= kdela) {
        /* y-plane crossing, raytracing.f90:715-758 */
        alam = (double)((float)(jm - j0) + (float)sgnj * 0.5f) / dj;
        zc = alam * dk + (double)(float)k0;
        xc = alam * di + (double)(float)i0;
        dz = 2.0 * fabs(zc - (double)((float)km + 0.5f * (float)sgnk));
        dx = 2.0 * fabs(xc - (double)((float)im + 0.5f * (float)sgni));
        s1 = (1. - dx) * (1. - dz);
        s2 = (1. - dz) * dx;
        s3 = (1. - dx) * dz;
        s4 = dx * dz;
        int ip = f_modulo(i - 1, m1) + 1, imp = f_modulo(im - 1, m1) + 1;
        int jmp = f_modulo(jm - 1, m2) + 1;
        int kp = f_modulo(k - 1, m3) + 1, kmp = f_modulo(km - 1, m3) + 1;
        c1 = CD(imp, jmp, kmp);
        c2 = CD(ip, jmp, kmp);
        c3 = CD(imp, jmp, kp);
        c4 = CD(ip, jmp, kp);
        w1 = s1 * weightf(c1, sigma);
        w2 = s2 * weightf(c2, sigma);
        w3 = s3 * weightf(c3, sigma);
        w4 = s4 * weightf(c4, sigma);
        *cdensi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
        if (jdela == 1 && (idela == 1 || kdela == 1)) {
            if (idela == 1 && kdela == 1) *cdensi = sqrt3 * *cdensi;
            else *cdensi = sqrt2 * *cdensi;
        }
        *path = sqrt((di * di + dk * dk) / (dj * dj) + 1.0);
    } else {
        /* x-plane crossing, raytracing.f90:764-800 (idela >= jdela && idela >= kdela) */
        alam = (double)((float)(im - i0) + (float)sgni * 0.5f) / di;
        zc = alam * dk + (double)(float)k0;
        yc = alam * dj + (double)(float)j0;
        dz = 2.0 * fabs(zc - (double)((float)km + 0.5f * (float)sgnk));
        dy = 2.0 * fabs(yc - (double)((float)jm + 0.5f * (float)sgnj));
        s1 = (1. - dz) * (1. - dy);
        s2 = (1. - dz) * dy;
        s3 = (1. - dy) * dz;
        s4 = dy * dz;
        int imp = f_modulo(im - 1, m1) + 1;
        int jp = f_modulo(j - 1, m2) + 1, jmp = f_modulo(jm - 1, m2) + 1;
        int kp = f_modulo(k - 1, m3) + 1, kmp = f_modulo(km - 1, m3) + 1;
        c1 = CD(imp, jmp, kmp);
        c2 = CD(imp, jp, kmp);
        c3 = CD(imp, jmp, kp);
        c4 = CD(imp, jp, kp);
        w1 = s1 * weightf(c1, sigma);
        w2 = s2 * weightf(c2, sigma);
        w3 = s3 * weightf(c3, sigma);
        w4 = s4 * weightf(c4, sigma);
        *cdensi = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
        if (idela == 1 && (jdela == 1 || kdela == 1)) {
            if (jdela == 1 && kdela == 1) *cdensi = sqrt3 * *cdensi;
            else *cdensi = sqrt2 * *cdensi;
        }
        *path = sqrt(1.0 + (dj * dj + dk * dk) / (di * di));
    }
#undef CD
}

/* Everything evolve0D needs that does not change between cells. */
typedef struct {
    const double *normflux;
    const int *srcpos; /* (3,NumSrc) Fortran layout: srcpos[3*ns + d], 1-based values */
    int ns;            /* 0-based source number */
    int NumSrc;
    double *cdh;
    double sig, dr;
    const double *ndens;
    const double *xh_av;
    double *phi_ion, *phi_heat;
    const double *thin, *thick, *hthin, *hthick;
    double minlogtau, dlogtau, R_max_LLS;
    int NumTau, m1, m2, m3;
    int flux_mode;
    long long visits; /* statistics: number of cells actually evaluated */
    double *phi_sens, *heat_sens; /* optional conditioning grids (see c2ray_set_sens_outputs) */
} rt_ctx;

static double *g_phi_sens = 0, *g_heat_sens = 0;
/* Register (or clear, with NULLs) N^3 grids that the next c2ray_do_all_sources calls fill with the
 * summed per-source sensitivity of phi_ion / phi_heat to a +-1 ulp change of log10.  Zeroed here. */
void c2ray_set_sens_outputs(double *phi_sens, double *heat_sens, long long ncell) {
    g_phi_sens = phi_sens;
    g_heat_sens = heat_sens;
    if (phi_sens) memset(phi_sens, 0, (size_t)ncell * sizeof(double));
    if (heat_sens) memset(heat_sens, 0, (size_t)ncell * sizeof(double));
}

/* ------------------------------------------------------------------------------------------
 * raytracing.f90:347-567  evolve0D
 * ---------------------------------------------------------------------------------------- */
static void evolve0D(rt_ctx *c, const int *rtpos, const int *last_l, const int *last_r,
                     double *photon_loss_src) {
    const double max_coldensh = (double)2e30f; /* raytracing.f90:368, default-real literal */
    const int *sp = c->srcpos + 3 * c->ns;
    int stop_rad_transfer = 0;
    int p0 = f_modulo(rtpos[0] - 1, c->m1), p1 = f_modulo(rtpos[1] - 1, c->m2),
        p2 = f_modulo(rtpos[2] - 1, c->m3); /* 0-based wrapped */
    size_t id = IDX(p0, p1, p2, c->m2, c->m3);
    double xh_av_p = c->xh_av[id];
    double nHI_p = c->ndens[id] * (1.0 - xh_av_p);
    if (c->cdh[id] == 0.0) {
        double coldensh_in, path, vol_ph;
        double phi_ion_p = 0.0, phi_ion_out = 0.0, phi_heat_p = 0.0; /* D2 */
        c->visits++;
        if (rtpos[0] == sp[0] && rtpos[1] == sp[1] && rtpos[2] == sp[2]) {
            coldensh_in = 0.0;
            path = 0.5 * c->dr;
            vol_ph = c->dr * c->dr * c->dr;
        } else {
            c2ray_cinterp(rtpos, sp, &coldensh_in, &path, c->cdh, c->sig, c->m1, c->m2, c->m3);
            path = path * c->dr;
            double xs = c->dr * (double)(float)(rtpos[0] - sp[0]);
            double ys = c->dr * (double)(float)(rtpos[1] - sp[1]);
            double zs = c->dr * (double)(float)(rtpos[2] - sp[2]);
            double dist2 = xs * xs + ys * ys + zs * zs;
            vol_ph = dist2 * path * (4.0 * PI_D);
            if (dist2 / (c->dr * c->dr) > c->R_max_LLS * c->R_max_LLS) stop_rad_transfer = 1;
            if (coldensh_in > max_coldensh) stop_rad_transfer = 1;
        }
        double cdo = coldensh_in + nHI_p * path;
        c->cdh[id] = cdo;
        if (!stop_rad_transfer) {
            double flux = c->flux_mode ? c->normflux[c->ns] : c->normflux[c->NumSrc - 1]; /* D3 */
            c2ray_photoion_rates(flux, coldensh_in, cdo, vol_ph, c->sig, &phi_ion_p, &phi_ion_out,
                                 &phi_heat_p, c->thin, c->thick, c->hthin, c->hthick, c->minlogtau,
                                 c->dlogtau, c->NumTau);
            if (c->phi_sens) {
                double sp_, sh_;
                c2ray_photoion_rates_sens(flux, coldensh_in, cdo, vol_ph, c->sig, &sp_, &sh_, c->thin, c->thick,
                                          c->hthin, c->hthick, c->minlogtau, c->dlogtau, c->NumTau);
                c->phi_sens[id] += fabs(sp_ / nHI_p);
                if (c->heat_sens) c->heat_sens[id] += fabs(sh_ / nHI_p);
            }
        } else {
            phi_ion_p = 0.0;
        }
        phi_ion_p = phi_ion_p / nHI_p;
        phi_heat_p = phi_heat_p / nHI_p;
        c->phi_ion[id] = c->phi_ion[id] + phi_ion_p;
        c->phi_heat[id] = c->phi_heat[id] + phi_heat_p;
        /* USE_SUBBOX photon-loss statistic, raytracing.f90:540-552 */
        if (rtpos[0] == last_l[0] || rtpos[1] == last_l[1] || rtpos[2] == last_l[2] ||
            rtpos[0] == last_r[0] || rtpos[1] == last_r[1] || rtpos[2] == last_r[2]) {
            *photon_loss_src = *photon_loss_src + phi_ion_out * (c->dr * c->dr * c->dr);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * raytracing.f90:258-340  evolve2D: one z-plane, four quadrant sweeps outward from the source
 * ---------------------------------------------------------------------------------------- */
static void evolve2D(rt_ctx *c, int k, const int *last_l, const int *last_r, double *photon_loss_src) {
    const int *sp = c->srcpos + 3 * c->ns;
    int rtpos[3];
    rtpos[2] = k;
    for (int j = sp[1]; j <= last_r[1]; j++) {
        rtpos[1] = j;
        for (int i = sp[0]; i <= last_r[0]; i++) { rtpos[0] = i; evolve0D(c, rtpos, last_l, last_r, photon_loss_src); }
        for (int i = sp[0] - 1; i >= last_l[0]; i--) { rtpos[0] = i; evolve0D(c, rtpos, last_l, last_r, photon_loss_src); }
    }
    for (int j = sp[1] - 1; j >= last_l[1]; j--) {
        rtpos[1] = j;
        for (int i = sp[0]; i <= last_r[0]; i++) { rtpos[0] = i; evolve0D(c, rtpos, last_l, last_r, photon_loss_src); }
        for (int i = sp[0] - 1; i >= last_l[0]; i--) { rtpos[0] = i; evolve0D(c, rtpos, last_l, last_r, photon_loss_src); }
    }
}

/* ------------------------------------------------------------------------------------------
 * raytracing.f90:127-249  do_source (USE_SUBBOX branch, as the reference is built:
 * src/c2ray/meson.build:53)
 * ---------------------------------------------------------------------------------------- */
static void do_source(rt_ctx *c, int max_subbox, int subboxsize, float loss_fraction, int *sum_nbox,
                      double *photon_loss) {
    const int *sp = c->srcpos + 3 * c->ns;
    int lastpos_l[3], lastpos_r[3], last_l[3], last_r[3];
    size_t ncell = (size_t)c->m1 * c->m2 * c->m3;
    for (int d = 0; d < 3; d++) {
        lastpos_r[d] = sp[d] + i_min(max_subbox, c->m1 / 2 - 1 + c->m1 % 2);
        lastpos_l[d] = sp[d] - i_min(max_subbox, c->m1 / 2);
    }
    memset(c->cdh, 0, ncell * sizeof(double)); /* raytracing.f90:181 */
    int nbox = 0;
    double photon_loss_src = c->normflux[c->ns] * S_STAR;
    for (int d = 0; d < 3; d++) last_r[d] = last_l[d] = sp[d];
    /* loss_fraction is default real: loss_fraction*normflux(ns)*S_star is evaluated left to
     * right in double after promotion (raytracing.f90:193) */
    while (photon_loss_src > (double)loss_fraction * c->normflux[c->ns] * S_STAR &&
           last_r[2] < lastpos_r[2] && last_l[2] > lastpos_l[2]) {
        photon_loss_src = 0.0;
        nbox = nbox + 1;
        for (int d = 0; d < 3; d++) {
            last_r[d] = i_min(sp[d] + subboxsize * nbox, lastpos_r[d]);
            last_l[d] = i_max(sp[d] - subboxsize * nbox, lastpos_l[d]);
        }
        for (int k = sp[2]; k <= last_r[2]; k++) evolve2D(c, k, last_l, last_r, &photon_loss_src);
        for (int k = sp[2] - 1; k >= last_l[2]; k--) evolve2D(c, k, last_l, last_r, &photon_loss_src);
    }
    *sum_nbox = *sum_nbox + nbox;
    *photon_loss = *photon_loss + photon_loss_src;
}

/* ------------------------------------------------------------------------------------------
 * raytracing.f90:52-119  do_all_sources.  Returns the number of cells evaluated (statistic the
 * reference does not have; used to report visited-cells/s for the CPU baseline).
 * phi_heat is NOT zeroed, exactly like the reference (only phi_ion is, raytracing.f90:95).
 * ---------------------------------------------------------------------------------------- */
long long c2ray_do_all_sources(const double *normflux, const int *srcpos, int max_subbox, int subboxsize,
                               double *coldensh_out, double sig, double dr, const double *ndens,
                               const double *xh_av, double *phi_ion, double *phi_heat, int *sum_nbox,
                               double *photon_loss, float loss_fraction, const double *photo_thin_table,
                               const double *photo_thick_table, const double *heat_thin_table,
                               const double *heat_thick_table, double minlogtau, double dlogtau,
                               double R_max_LLS, int NumTau, int NumSrc, int m1, int m2, int m3,
                               int flux_mode) {
    rt_ctx c;
    size_t ncell = (size_t)m1 * m2 * m3;
    memset(phi_ion, 0, ncell * sizeof(double));
    c.normflux = normflux; c.srcpos = srcpos; c.NumSrc = NumSrc; c.cdh = coldensh_out;
    c.sig = sig; c.dr = dr; c.ndens = ndens; c.xh_av = xh_av; c.phi_ion = phi_ion; c.phi_heat = phi_heat;
    c.thin = photo_thin_table; c.thick = photo_thick_table; c.hthin = heat_thin_table; c.hthick = heat_thick_table;
    c.minlogtau = minlogtau; c.dlogtau = dlogtau; c.R_max_LLS = R_max_LLS; c.NumTau = NumTau;
    c.m1 = m1; c.m2 = m2; c.m3 = m3; c.flux_mode = flux_mode; c.visits = 0;
    c.phi_sens = g_phi_sens; c.heat_sens = g_heat_sens;
    *sum_nbox = 0;
    *photon_loss = 0.0;
    for (int ns = 0; ns < NumSrc; ns++) {
        c.ns = ns;
        do_source(&c, max_subbox, subboxsize, loss_fraction, sum_nbox, photon_loss);
    }
    return c.visits;
}

/* ------------------------------------------------------------------------------------------
 * chemistry.f90:226-317  doric
 * ---------------------------------------------------------------------------------------- */
void c2ray_doric(double xh_old, double dt, double temp_p, double rhe, double phi_p, double bh00,
                 double albpow, double colh0, double temph0, double clumping, double *xh, double *xh_av) {
    const double epsilon = 1e-14; /* chemistry.f90:8 */
    double brech0 = clumping * bh00 * pow(temp_p / (double)1e4f, albpow);
    double sqrtt0 = sqrt(temp_p);
    double acolh0 = colh0 * sqrtt0 * exp(-temph0 / temp_p);
    double aphoth0 = phi_p;
    double aih0 = aphoth0 + rhe * acolh0;
    double delth = aih0 + rhe * brech0;
    double eqxh = aih0 / delth;
    double deltht = delth * dt;
    double ee = exp(-deltht);
    double avg_factor;
    *xh = (xh_old - eqxh) * ee + eqxh;
    if (*xh < epsilon) *xh = epsilon;
    if (deltht < (double)1.0e-8f) avg_factor = 1.0;
    else avg_factor = (1.0 - ee) / deltht;
    *xh_av = eqxh + (xh_old - eqxh) * avg_factor;
    if (*xh_av < epsilon) *xh_av = epsilon;
}

/* ------------------------------------------------------------------------------------------
 * chemistry.f90:123-209  do_chemistry (isothermal: the temperature test is always true)
 * ---------------------------------------------------------------------------------------- */
static void do_chemistry(double dt, double ndens_p, double temperature_start, double xh_p, double *xh_av_p,
                         double *xh_intermed_p, double phi_ion_p, double clump_p, double bh00,
                         double albpow, double colh0, double temph0, double abu_c) {
    const double minimum_fractional_change = (double)1.0e-3f; /* chemistry.f90:9-10: default-real literals */
    const double minimum_fraction_of_atoms = (double)1.0e-8f;
    double temperature_end = temperature_start, temperature_previous_iteration;
    int nit = 0;
    for (;;) {
        nit = nit + 1;
        temperature_previous_iteration = temperature_end;
        double xh_av_p_old = *xh_av_p;
        double de = ndens_p * (*xh_av_p + abu_c);
        c2ray_doric(xh_p, dt, temperature_end, de, phi_ion_p, bh00, albpow, colh0, temph0, clump_p,
                    xh_intermed_p, xh_av_p);
        if ((fabs((*xh_av_p - xh_av_p_old) / (1.0 - *xh_av_p)) < minimum_fractional_change ||
             (1.0 - *xh_av_p < minimum_fraction_of_atoms)) &&
            (fabs((temperature_end - temperature_previous_iteration) / temperature_end) < minimum_fractional_change))
            break;
        if (nit > 400) break;
    }
}

/* ------------------------------------------------------------------------------------------
 * chemistry.f90:13-49 global_pass + :54-115 evolve0D_global.  Returns conv_flag.
 * The loop order (k outer, i inner in Fortran memory order) has no effect on the results:
 * cells are independent; conv_flag is a count.
 * ---------------------------------------------------------------------------------------- */
int c2ray_global_pass(double dt, const double *ndens, const double *temp, const double *xh, double *xh_av,
                      double *xh_intermed, const double *phi_ion, const double *clump, double bh00,
                      double albpow, double colh0, double temph0, double abu_c, int m1, int m2, int m3) {
    const double minimum_fractional_change = (double)1.0e-3f;
    const double minimum_fraction_of_atoms = (double)1.0e-8f;
    int conv_flag = 0;
    size_t ncell = (size_t)m1 * m2 * m3;
    for (size_t id = 0; id < ncell; id++) {
        double xh_av_p = xh_av[id];
        double xh_intermed_p = xh_intermed[id];
        double yh_av_p = 1.0 - xh_av_p;
        do_chemistry(dt, ndens[id], temp[id], xh[id], &xh_av_p, &xh_intermed_p, phi_ion[id], clump[id], bh00,
                     albpow, colh0, temph0, abu_c);
        double xh_av_p_old = xh_av[id];
        if (fabs(xh_av_p - xh_av_p_old) > minimum_fractional_change &&
            fabs((xh_av_p - xh_av_p_old) / yh_av_p) > minimum_fractional_change &&
            yh_av_p > minimum_fraction_of_atoms)
            conv_flag = conv_flag + 1;
        xh_intermed[id] = xh_intermed_p;
        xh_av[id] = xh_av_p;
    }
    return conv_flag;
}

/* ------------------------------------------------------------------------------------------
 * Unit of work of the benchmark metric: number of cells ASORA
 * visits per source = |octahedron(q_max) ∩ box(last_l..last_r)|, and how many of them lie
 * inside the R sphere (src/asora/raytracing.cu:110,140-141,214-255,320).  Pure integer count,
 * used by tests to pin the closed-form counter of the product's host code.
 * ---------------------------------------------------------------------------------------- */
void c2ray_asora_visit_count(double R, int m1, long long *visits, long long *in_sphere) {
    const double SQRT3 = 1.73205080757; /* the constant ASORA uses for q_max */
    double rr = R < SQRT3 * m1 / 2.0 ? R : SQRT3 * m1 / 2.0;
    int q_max = (int)ceil(SQRT3 * rr);
    int last_r = m1 / 2 - 1 + m1 % 2, last_l = -(m1 / 2);
    long long v = 0, s = 0;
    for (int i = last_l; i <= last_r; i++)
        for (int j = last_l; j <= last_r; j++)
            for (int k = last_l; k <= last_r; k++) {
                if (i_abs(i) + i_abs(j) + i_abs(k) > q_max) continue;
                v++;
                double d2 = (double)i * i + (double)j * j + (double)k * k;
                if (!(d2 > R * R)) s++;
            }
    *visits = v;
    *in_sphere = s;
}
