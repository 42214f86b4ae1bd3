/*
 * asora_hip.h -- C-ABI of libasora_hip.so, the MI355X (gfx950) implementation of pyC2Ray's
 * ASORA raytracing path.  Plain pointers and sizes only; no Python, torch or HIP types.
 *
 * Every entry point returns 0 on success and a non-zero code on failure;
 * asora_last_error() then returns a human-readable message (the reference prints HIP/CUDA
 * errors to stderr and swallows them, src/asora/utils.cu:9-19 + memory.cu:55-57; callers of
 * this library must check the return code -- the Python shim raises RuntimeError).
 *
 * One device per process, process-global state, not re-entrant -- like the reference
 * (src/asora/memory.cu:28-37).
 *
 * Grids are C-ordered double[N*N*N], flat index N*N*i + N*j + k (src/asora/raytracing.cu:38-40).
 * Source positions are 0-based int32, interleaved x0,y0,z0,x1,... (pyc2ray/utils/sourceutils.py:30).
 */
#ifndef ASORA_HIP_H
#define ASORA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the six functions of the reference's `libasora` module --------------------------------- */

/* replaces asora::device_init, src/asora/memory.cu:42-86 (Python: python_module.cu:64-74 "iiii").
 * Selects device mpi_rank % num_gpus and allocates the grid buffers for mesh size N.
 * num_src_par (the reference's source batch size) is accepted for API compatibility; the number
 * of sources in flight is decided by the hardware scheduler here (no per-source N^3 buffer). */
int asora_device_init(int N, int num_src_par, int mpi_rank, int num_gpus);

/* replaces asora::device_close, src/asora/memory.cu:138-150 */
int asora_device_close(void);

/* The HIP device asora_device_init selected (mpi_rank % num_gpus, src/asora/memory.cu:46).  Every entry point makes
 * that device current for the duration of the call and restores the caller's, so a caller that switched devices in
 * between stays correct; device pointers handed to the *_dev entry points must live on this device. */
int asora_device_index(int *device);

/* replaces asora::density_to_device, src/asora/memory.cu:91-98 (host pointer, N*N*N doubles) */
int asora_density_to_device(const double *ndens, int N);

/* replaces asora::photo_table_to_device, src/asora/memory.cu:100-115.  Copies num_tau doubles of
 * each table (as the reference does) and pads the device copy by two replicated entries so the
 * reference's one-past-the-end read (src/asora/rates.cu:96-100) is defined. */
int asora_photo_table_to_device(const double *thin_table, const double *thick_table, int num_tau);

/* replaces asora::source_data_to_device, src/asora/memory.cu:117-133.  Positions are PERIODIC like in the
 * reference (built with -DPERIODIC, src/asora/meson.build:33; every grid access wraps through modulo(i, N),
 * src/asora/raytracing.cu:32-40): a coordinate outside [0, N) -- e.g. N in a 1-based list handed over unshifted --
 * addresses cell (coordinate mod N), while the interpolation carries the coordinate as given (raytracing.f90:683),
 * exactly as both reference codes do.  Only |coordinate| > 2^24 is refused. */
int asora_source_data_to_device(const int32_t *pos, const double *flux, int num_src);

/* replaces asora::do_all_sources_gpu, src/asora/raytracing.cu:98-169 (Python:
 * python_module.cu:21-59 "dOddOOOiiddi").  All pointers are HOST pointers.  `ndens` is ignored
 * like in the reference (the copy made by asora_density_to_device is used).  phi_ion and
 * coldensh_out (N^3 doubles each) are overwritten.  coldensh_out receives the outgoing column
 * density of the LAST source of the list (the Fortran semantics, raytracing.f90:181; the reference
 * GPU code returns an unspecified source's grid for num_src > 1), zero outside
 * the region that source visits. */
int asora_do_all_sources(double R, double *coldensh_out, double sig, double dr, const double *ndens,
                         const double *xh_av, double *phi_ion, int num_src, int m1, double minlogtau,
                         double dlogtau, int num_tau);

/* ---- extensions (no counterpart in the H-only reference module) ------------------------------- */

/* Heating tables + heating-rate output; arithmetic of src/c2ray/photorates.f90:118,124
 * (the multi-species reference variant exposes the same idea: src/asora_he/python_module_He.cu:22-102). */
int asora_heat_table_to_device(const double *heat_thin_table, const double *heat_thick_table, int num_tau);
int asora_do_all_sources_heat(double R, double *coldensh_out, double sig, double dr, const double *ndens,
                              const double *xh_av, double *phi_ion, double *phi_heat, int num_src, int m1,
                              double minlogtau, double dlogtau, int num_tau);

/* Device-resident variant: xh_av_dev, phi_ion_dev, phi_heat_dev (nullable), coldensh_out_dev
 * (nullable) are DEVICE pointers to N^3 doubles; nothing crosses PCIe.  `stream` is a hipStream_t
 * (NULL = default stream); the call is asynchronous with respect to the host.  This is what
 * bench.py times and what the multi-GPU driver all-reduces (replaces the host Reduce+Bcast of
 * pyc2ray/evolve.py:309-317). */
int asora_do_all_sources_dev(double R, double *coldensh_out_dev, double sig, double dr,
                             const double *xh_av_dev, double *phi_ion_dev, double *phi_heat_dev,
                             int num_src, int m1, double minlogtau, double dlogtau, int num_tau,
                             void *stream);

/* Device-pointer upload of the density (device-to-device copy on `stream`). */
int asora_density_from_device(const double *ndens_dev, int N, void *stream);

/* ---- chemistry pass of the evolve loop ---------------------------------- */

/* replaces chemistry.global_pass, src/c2ray/chemistry.f90:13-49 (f2py call at pyc2ray/evolve.py:326-340:
 * conv_flag = global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, clump, bh00, albpow, colh0,
 * temph0, abu_c)).  HOST pointers to m1^3 doubles in the same flat C order as the raytracing grids
 * (cells are independent, so the layout only has to be the same for all seven); xh_av and xh_intermed
 * are updated in place, *conv_flag receives the number of non-converged cells. */
int asora_global_pass(double dt, const double *ndens, const double *temp, const double *xh, double *xh_av,
                      double *xh_intermed, const double *phi_ion, const double *clump, double bh00, double albpow,
                      double colh0, double temph0, double abu_c, int m1, int *conv_flag);

/* Same on DEVICE pointers, asynchronous on `stream` (hipStream_t).  Also forms the two sums the
 * reference's convergence test takes on the host afterwards (sum(xh_intermed), sum(1 - xh_intermed),
 * pyc2ray/evolve.py:350-351).  If any of conv_flag / sum_xh1 / sum_xh0 is non-NULL the call waits for
 * the stream and stores the results there; with all three NULL it returns right after the launch. */
int asora_global_pass_dev(double dt, const double *ndens_dev, const double *temp_dev, const double *xh_dev,
                          double *xh_av_dev, double *xh_intermed_dev, const double *phi_ion_dev,
                          const double *clump_dev, double bh00, double albpow, double colh0, double temph0,
                          double abu_c, int m1, long long *conv_flag, double *sum_xh1, double *sum_xh0,
                          void *stream);

/* The one entry point that ALLOCATES NOTHING.  No counterpart in the reference.  It is for callers that cannot
 * call hipMalloc while they run: code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation.  Same arguments, results and kernel as asora_global_pass_dev, but the kernel's reduction record
 * (convergence counter + the two sums) lives in a file-scope device array of ASORA_NOALLOC_SCRATCH_BYTES = 256
 * bytes that each GPU holds from library load to process end (its per-device address is taken once, in
 * asora_device_init), and the results land in a page-locked buffer made at init: the call itself performs no
 * hipMalloc / hipFree / hipHostMalloc and no allocating library routine.  Mesh sizes above ASORA_NOALLOC_MAX_M1
 * are refused with an error.  All calls on one GPU share that GPU's scratch: it holds a ring of 8 reduction records,
 * taken in turn and cleared at the start of the call, so up to 8 calls may be in flight on one GPU at a time (other
 * streams, calls that do not ask for results); a ninth overlapping call races with the first.  The landing buffer of
 * the results is one per device: calls that ASK for results must not run concurrently from several host threads (the
 * library's state is process-global and not thread-safe, like the reference's). */
#define ASORA_NOALLOC_MAX_M1 1024
int asora_global_pass_dev_noalloc(double dt, const double *ndens_dev, const double *temp_dev, const double *xh_dev,
                                  double *xh_av_dev, double *xh_intermed_dev, const double *phi_ion_dev,
                                  const double *clump_dev, double bh00, double albpow, double colh0, double temph0,
                                  double abu_c, int m1, long long *conv_flag, double *sum_xh1, double *sum_xh0,
                                  void *stream);

/* replaces asoratest::cinterp_gpu, src/asora/tests.cu:30-59 (Python module libasoratest.cinterp,
 * python_test_module.cu): for the 4x4x4 block of offsets (i,j,k) in [0,4)^3 from pos0, column density
 * and path from the short-characteristics interpolation on `dens` (N^3 doubles, used as the
 * outgoing-column-density grid), sigma = 1.  out = double[4*4*4*2].  `fortran_constants` != 0
 * uses the float32-rounded sqrt(2), sqrt(3) of the Fortran oracle (what the product path uses),
 * 0 the truncated decimal constants of src/asora/raytracing.cu:20-21. */
int asora_test_cinterp(const int32_t pos0[3], const double *dens, int N, double *out, int fortran_constants);

/* Same harness for a cnt^3 block of offsets lo .. lo+cnt-1 per axis (cnt <= 8, negative offsets
 * allowed, periodic wrap) and any sigma; out = double[cnt*cnt*cnt*2]. */
int asora_test_cinterp_ex(const int32_t pos0[3], const double *dens, int N, double sigma, int lo, int cnt,
                          double *out, int fortran_constants);

/* Unit of work of the benchmark metric: cells visited per source (octahedron(q_max) ∩ periodic
 * box limits) and how many of those lie inside the R sphere; follows
 * src/asora/raytracing.cu:110,140-141,214-255,320.  Pure host arithmetic. */
int asora_visit_count(double R, int m1, long long *visits, long long *in_sphere);

/* Host-side self-check of the sweep's cell tables (no device needed): builds the geometry tables of
 * (m1, R, dr) exactly as a do_all_sources call would and counts, over the 24 sectors of one source, the
 * cells that are OWNED (produce output) -- must equal asora_visit_count's `visits` --, the owned cells
 * inside the R sphere by the oracle's floating-point predicate, and all cells computed (owned + the
 * sector-boundary cells recomputed by neighbouring sectors).  Also returns the staged pairs per task and
 * the number of distinct geometry variants. */
int asora_geometry_census(double R, double dr, int m1, long long *owned, long long *owned_rate, long long *computed,
                          int *staging_entries, int *variants);

/* Order in which the library sweeps a source list (pure host function, no device needed): perm[i] = index in
 * `pos` (x0,y0,z0,x1,... 0-based, as for asora_source_data_to_device) of the i-th source swept.  order 0: as
 * given, 1: Morton (Z) curve, 2: Hilbert curve (the default, option "source_order").  The rate grids are sums
 * over sources, so the order only decides which sources are in flight together; neighbours share their
 * neutral-density reads and meet in cache with their rate atomics.  coldensh_out is that of the caller's last
 * source whatever the order.  SURVEY.md 8(f)4 (source-list preprocessing); no counterpart in the reference,
 * which sweeps in list order (src/asora/raytracing.cu:143-158). */
int asora_source_order(const int32_t *pos, int num_src, int m1, int order, int32_t *perm);

/* Duration of the raytracing sweep kernel of the last do_all_sources* call, from HIP events
 * recorded on the launch stream (synchronises on the stop event). */
int asora_last_sweep_ms(float *ms);

/* Launch geometry of the last sweep (for profile matching): workgroups, threads per workgroup, dynamic LDS
 * bytes per workgroup, and where the staged column densities lived: 0 global scratch, 1 LDS as (column
 * density, weight) pairs, 2 LDS as column densities only; +4 * log2(sectors per workgroup) when the four-quadrant /
 * eight-octant kernel ran; +16 * log2(sources per workgroup) when it took several sources side by side; +64 when the call
 * ran on one grid layout (option "single_layout"). */
int asora_last_launch_info(int *grid, int *block, int *lds_bytes, int *staging_is_lds);

/* Tuning / test switches: "force_generic" (1: always run the table-free sweep kernel),
 * "max_table_cells" (geometry-table budget in cells; larger problems use the table-free kernel),
 * "lds_stage_limit" (bytes of LDS per workgroup above which the staged column densities live in global
 * memory; 0 = automatic: half a CU's LDS), "wg_threads" (64/128/192/256/512/1024: workgroup size of the
 * table-driven kernel; 0 = automatic: the smallest size that keeps >= 14 waves on a CU given the LDS one
 * workgroup needs),
 * "compact_staging" (1: stage column densities only and recompute the weights, 8 instead of 16 B
 * of LDS per entry; 0 = automatic: when the pairs would allow at most one workgroup per CU; -1: never),
 * "quad_kernel" (four sign quadrants of a pyramid per workgroup, for small radii; 0 = automatic: up to
 * "quad_max_entries" staged entries per quadrant, default 1210, i.e. R <~ 46 cells; 1: whenever the quadrants share
 * their geometry; 2: eight octants of an axis per workgroup, automatic up to "oct_max_entries", default 120, R <~ 12;
 * -1: never),
 * "src_pack" (the multi-sector kernels take 1 << src_pack consecutive sources of the sweep order per workgroup, side by
 * side in the lanes of a wavefront: 0 = automatic, while four 256-thread workgroups of that kind fit on a CU --
 * "pack_max_lds" bytes each, default 40 KiB, R <~ 12 cells; 2 / 3: four / eight sources; -1: never),
 * "stride_residue" (multi-sector kernels: -1 = pad the per-sector staging arrays so that the lanes of a cell hit different
 * LDS banks, while the workgroups per CU stay the same; -2: the geometry's size; >= 0: that size modulo 32),
 * "source_order" (see asora_source_order),
 * "skip_unobserved" (1: cells outside the R-sphere -- never upwind of a cell inside it -- are swept only for
 * the source whose column density is exported; phi and coldensh_out unchanged, ~27 % less time at 256^3 / R = 60;
 * off by default so that the default path visits exactly the cells the reference visits),
 * "skip_saturated" (1: cells behind an optical depth beyond the photo table's last entry -- where both look-ups of
 * photo_lookuptable clamp to that entry and the rate is exactly +0, photorates.f90:140-146 -- evaluate no rates;
 * identical results; LDS-staged single-sector kernels of >= 256 threads only; off by default),
 * "single_layout" (0 = automatic: calls of up to "single_max_visits" (source, cell) visits (default 0: 1.2 per cell of the mesh) at R <~ 46 and all calls with q_max <= 10 run on ONE grid
 * layout -- the z-dominant sectors address the [i][j][k] grids with a stride, the sweep adds straight into the caller's
 * grids, and the per-call passes shrink from 9 to 4 N^3 doubles of traffic; 1: whenever the table-driven kernel runs;
 * -1: never.  Same arithmetic per cell: bit-identical for a single source),
 * "copy_threads" (host threads staging the grids of the host-pointer entry points through page-locked memory,
 * default 4, 0: plain hipMemcpy), "lds_pad" (extra LDS per workgroup, occupancy experiments). */
int asora_set_option(const char *name, long long value);

/* Current value of one of the switches above (a caller that changes one for the length of a call puts it back
 * afterwards: pyc2ray_amd/evolve.py).  No counterpart in the reference. */
int asora_get_option(const char *name, long long *value);

/* 1 if the last sweep ran the table-driven kernel, 0 for the table-free one. */
int asora_last_sweep_is_table_driven(void);

/* One line saying why the last sweep ran the (about 3x slower) table-free kernel, "" when it ran the table-driven
 * one.  The table-driven kernel addresses the grids with 32-bit byte offsets, 10-bit in-plane coordinates and
 * 16-bit staging positions: meshes of 2^28 cells or more (N >= 646), plane extents >= 1024 cells, more than 65535
 * staged entries per sector or geometry tables beyond "max_table_cells" take the fallback.  No counterpart in the
 * reference (diagnostics of this library).  The pointer stays valid until the next sweep. */
const char *asora_last_launch_notice(void);

/* Device self-test: counts the inputs for which the kernels' division helpers (div_exact,
 * rcp_exact, div_const) differ in any bit from the compiler's IEEE-754 division num[i]/den[i]. */
int asora_test_division(const double *num, const double *den, long long n, long long *mismatch_div,
                        long long *mismatch_rcp, long long *mismatch_const);

const char *asora_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ASORA_HIP_H */
