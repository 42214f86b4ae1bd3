// asora_capi.hip -- host side of libasora_hip.so: process-global device state and the
// extern "C" entry points declared in include/asora_hip.h.
//
// State mirrors what the reference keeps in src/asora/memory.cu:28-37 (one device per process),
// minus the per-source N^3 column-density buffer (cdh_dev[M*N^3], memory.cu:75), which the sector
// decomposition in asora_kernels.hpp does not need.
#include "asora_kernels.hpp"
#include "asora_chem.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/asora_hip.h"

// Scratch of asora_global_pass_dev_noalloc (the entry point that allocates nothing): the chemistry kernel's
// reduction record (asora::ChemStats: convergence counter + two sums, 24 bytes) lives in this file-scope device
// array instead of a hipMalloc'ed one.  Each GPU the library runs on holds its own copy until the process ends.
// In host code the NAME below is only a stand-in for the symbol: the memory is reached through the per-device
// pointer that hipGetSymbolAddress returns (noalloc_scratch_ptr, set in asora_device_init), never through the name.
#define ASORA_NOALLOC_SCRATCH_BYTES 256
static __device__ __align__(256) unsigned char asora_noalloc_scratch[ASORA_NOALLOC_SCRATCH_BYTES];

namespace {

std::string g_err;

constexpr int MAX_DEVICES = 64;
void *noalloc_scratch_ptr[MAX_DEVICES] = {nullptr};  // device address of asora_noalloc_scratch on each device
// The chemistry kernel's reduction record is taken from a ring of CHEM_RECORDS slots, one per call in turn, so that up
// to CHEM_RECORDS calls in flight on one GPU (other streams, asynchronous calls) do not share a record.
constexpr int CHEM_RECORDS = 8, CHEM_RECORD_STRIDE = 32;
static_assert(sizeof(asora::ChemStats) <= CHEM_RECORD_STRIDE && CHEM_RECORDS * CHEM_RECORD_STRIDE <= ASORA_NOALLOC_SCRATCH_BYTES,
              "scratch too small");
std::atomic<unsigned> g_chem_turn{0};
asora::ChemStats *chem_record(void *ring) {
    return reinterpret_cast<asora::ChemStats *>(static_cast<char *>(ring) + (g_chem_turn.fetch_add(1) % CHEM_RECORDS) * CHEM_RECORD_STRIDE);
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(100 + (int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                        __LINE__);                                                                        \
    } while (0)

// Tuning options (asora_set_option) and the record of the last launch (asora_last_launch_info): both survive a
// close / re-init of the device.
struct Options {
    int source_order = 2;         // option "source_order": 0 as given, 1 Morton, 2 Hilbert curve
    int lds_stage_limit = 0;      // option "lds_stage_limit": larger staging arrays go to global memory (0: automatic)
    int lds_pad = 0;              // option "lds_pad": extra dynamic LDS per workgroup (occupancy experiments)
    int wg_threads = 0;           // option "wg_threads": workgroup size of the table-driven kernel, 256/512/1024 (0: automatic)
    int quad_kernel = 0;          // option "quad_kernel": four quadrants per workgroup (0: automatic, 1: whenever possible, -1: never)
    int stride_residue = -1;      // option "stride_residue": entries between the sectors' staging arrays (-1: padded for the LDS banks, -2: as built, >= 0: modulo 32)
    int oct_max_entries = 120;    // option "oct_max_entries": automatic choice of eight octants per workgroup up to this many entries (R <~ 12)
    int quad_max_entries = 1210;  // option "quad_max_entries": automatic choice up to this many staged entries per quadrant (R <~ 46)
    int src_pack = 0;             // option "src_pack": sources per workgroup of the multi-sector kernels as a shift (0: automatic, -1: one, 2: four, 3: eight)
    int pack_max_lds = 40 * 1024; // option "pack_max_lds": automatic packing only while a workgroup's LDS stays below this (four 256-thread workgroups per CU)
    int pack_max_entries = 64;    // option "pack_max_entries": ... and up to this many staged entries per sector (R <~ 8)
    int compact_staging = 0;      // option "compact_staging": stage column densities only (0: automatic, 1: always, -1: never)
    int skip_unobserved = 0;      // option "skip_unobserved": cells outside the R-sphere only for the exported column density
    int skip_saturated = 0;       // option "skip_saturated": no rate evaluation behind optical depths beyond the table's end
    int single_layout = 0;        // option "single_layout": z-dominant sectors on the [i][j][k] grids too, no transposed copies / merge pass (0: automatic, short calls; 1: whenever the table-driven kernel runs; -1: never)
    long long single_max_visits = 0;  // option "single_max_visits": automatic choice up to this many (source, cell) visits per call (0: 1.2 x the cells of the mesh)
    int force_generic = 0;        // option "force_generic": always use the table-free kernel
    long long max_table_cells = 48LL << 20;  // option "max_table_cells": beyond this the generic kernel runs
};
struct LaunchRecord {
    std::string notice;  // one line when the last sweep could not use the table-driven kernel (asora_last_launch_notice)
    int last_tab = 0;
    int last_grid = 0, last_block = 0, last_lds = 0, last_is_lds = 1;
};

struct State {
    Options opt;
    LaunchRecord rec;
    bool init = false;
    int device = 0;
    int N = 0;
    int num_src_par = 0;
    size_t cells = 0;
    // grids (N^3 doubles each)
    double *n_dev = nullptr, *x_dev = nullptr;        // density, ionised fraction (host-API staging)
    double *nhiA = nullptr, *nhiB = nullptr;          // neutral density, two layouts
    double *phiA = nullptr, *phiB = nullptr;          // rate accumulators, two layouts
    double *heatA = nullptr, *heatB = nullptr;        // lazily allocated
    double *cdhA = nullptr, *cdhB = nullptr;          // lazily allocated
    double *out0 = nullptr, *out1 = nullptr, *out2 = nullptr;  // merged outputs for the host API (lazy)
    // tables (padded by 2)
    double *thin = nullptr, *thick = nullptr, *hthin = nullptr, *hthick = nullptr;
    int table_len = 0, heat_table_len = 0;
    // sources
    int *src_pos = nullptr;       // device copies, in SWEEP order (see stage_sources)
    double *src_flux = nullptr;
    int num_src = 0;
    std::vector<int32_t> host_pos;   // the caller's list, in the caller's order
    std::vector<double> host_flux;
    int staged_n = -1;            // the device copies hold the first staged_n sources of the caller's list ...
    int staged_order = -1;        // ... sorted with this "source_order"
    int staged_last = -1;         // position of the caller's source staged_n - 1 in the sweep order
    // geometry tables of the table-driven sweep (cached per (N, R, dr))
    asora::PlaneInfo *planes_dev = nullptr;
    uint2 *cell_meta_dev = nullptr;
    double2 *cell_geo_dev = nullptr;
    int geo_N = 0, geo_nplanes = 0, geo_entries = 0, geo_vmap[8] = {0}, geo_nmax_rate = 0;
    long long geo_visits = -1, geo_in_sphere = -1;  // asora_visit_count of the cached geometry (-1: not counted yet)
    double geo_R = -1.0, geo_dr = -1.0;
    bool geo_valid = false;
    double *log10_table = nullptr;  // asora_log10.hpp table
    // staging scratch for planes that do not fit in LDS
    double *gscratch = nullptr;
    size_t gscratch_bytes = 0;
    // chemistry pass: reduction results, and grid staging for the host-pointer entry point (lazy)
    void *chem_stats = nullptr;                     // ring of CHEM_RECORDS reduction records (allocating entry point, lazy)
    asora::ChemStats *chem_stats_pinned = nullptr;  // page-locked landing buffer of the no-alloc entry point (made at init)
    double *chem_buf[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // timing / launch info
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int max_lds = 65536;
    int num_cu = 256;
};

State S;

// The grids, tables and sources live on S.device, and every launch / copy must be issued with that device
// current.  The caller's thread may have switched devices since asora_device_init (torch.cuda.set_device, another
// thread): each entry point that touches the device makes S.device current and restores the caller's on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    DeviceGuard() {
        if (!S.init) return;
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != S.device) {
            err = hipSetDevice(S.device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define ASORA_ON_DEVICE(fn)                                                                                    \
    DeviceGuard guard_;                                                                                        \
    if (guard_.err != hipSuccess)                                                                              \
        return fail(100 + (int)guard_.err, fn ": cannot make device %d current: %s", S.device, hipGetErrorString(guard_.err))

// source coordinates are periodic (the reference wraps them on every access, src/asora/raytracing.cu:32-40)
inline uint32_t wrapped(int32_t v, int N) {
    const int r = v % N;
    return (uint32_t)(r < 0 ? r + N : r);
}
constexpr int32_t MAX_SOURCE_COORD = 1 << 24;  // |coordinate| bound: keeps every int / double conversion on the path exact

// Set-up of the entry point that allocates nothing, once per device, with that device current (called from
// asora_device_init): the device address of THIS device's copy of the scratch array -- each GPU holds its own copy
// at its own address -- and a page-locked landing buffer for the three results.  (A plain host function on
// purpose: the reference to the array must be seen as a host-side use of the symbol.)
int noalloc_setup(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return fail(1, "asora_device_init: device index %d beyond %d", dev, MAX_DEVICES);
    void *p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(asora_noalloc_scratch)));
    if (!p) return fail(1, "asora_device_init: no device address for the scratch array");
    noalloc_scratch_ptr[dev] = p;
    HIP_TRY(hipHostMalloc((void **)&S.chem_stats_pinned, sizeof(asora::ChemStats), hipHostMallocDefault));
    return 0;
}

void free_ptr(double *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

int alloc_grid(double *&p) {
    if (p) return 0;
    HIP_TRY(hipMalloc(&p, S.cells * sizeof(double)));
    return 0;
}

int upload_table(double *&dev, const double *host, int n) {
    // n entries from the host + 2 replicated entries
    std::vector<double> tmp((size_t)n + 2);
    std::memcpy(tmp.data(), host, (size_t)n * sizeof(double));
    tmp[n] = tmp[n + 1] = host[n - 1];
    if (dev) (void)hipFree(dev);
    dev = nullptr;
    HIP_TRY(hipMalloc(&dev, tmp.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(dev, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

// q_max exactly as the reference computes it (src/asora/raytracing.cu:110, with its SQRT3 macro)
int compute_q_max(double R, int m1) {
    const double SQRT3 = 1.73205080757;
    return (int)std::ceil(SQRT3 * std::min(R, SQRT3 * m1 / 2.0));
}

// ------------------------------------------------------------------------------------------
// Geometry tables for sweep_tab_kernel.  For each distinct (limit_u, limit_v, summation order)
// variant and each plane n: list A = cells with dist2/dr2 <= R2 evaluated EXACTLY as the oracle does
// (raytracing.f90:448-451,474: xs = dr*real(di), dist2 = xs*xs + ys*ys + zs*zs in x,y,z order,
// no contraction -- this file is compiled with -ffp-contract=off for the host too), list B = the
// other cells of octahedron ∩ box, both in row-major (a, b) order.  Per cell: path = sqrt((a^2+b^2)/n^2 + 1)
// (raytracing.f90:710) stored as path*dr, vol_ph = dist2*path*dr*4pi (raytracing.f90:457), the
// ownership flags, and its positions in the variant's staging layout: the packed union of all
// planes' domains, row a holding one leading guard (b = -1) and b = 0 .. rowlen[a]-1, preceded by
// a guard row (a = -1).  Every position a cell reads lies inside that union (smaller coordinates
// of a member are members), so no trailing guards are needed.
// ------------------------------------------------------------------------------------------
void free_geometry() {
    if (S.planes_dev) (void)hipFree(S.planes_dev);
    if (S.cell_meta_dev) (void)hipFree(S.cell_meta_dev);
    if (S.cell_geo_dev) (void)hipFree(S.cell_geo_dev);
    S.planes_dev = nullptr; S.cell_meta_dev = nullptr; S.cell_geo_dev = nullptr;
    S.geo_valid = false;
}

long long geometry_cells_per_variant(int q_max, int lim, int nmaxmax) {
    long long c = 0;
    for (int n = 1; n <= nmaxmax; n++) {
        const int cut = q_max - n, h = std::min(std::min(n, lim), cut) + 1;
        c += (long long)h * h;
    }
    return c;
}

// widest in-plane extent the octahedron and the box allow
int plane_extent(int q_max, int last_l, int last_r) {
    const int lim = std::max(last_l, last_r), nmaxmax = std::min(lim, q_max);
    int cap = 0;
    for (int n = 0; n <= nmaxmax; n++) cap = std::max(cap, std::min(std::min(n, lim), q_max - n));
    return cap;
}

// staged entries of the packed layout for in-plane limits (LU, LV); fills rowoff (a + 1 -> first position)
int packed_layout(int q_max, int nmaxmax, int LU, int LV, std::vector<int> &rowoff) {
    std::vector<int> rowlen(LU + 1, 0);
    for (int n = 0; n <= nmaxmax; n++) {
        const int cut = q_max - n;
        const int H = std::min(std::min(n, LU), cut) + 1, W = std::min(std::min(n, LV), cut) + 1;
        for (int a = 0; a < H; a++) rowlen[a] = std::max(rowlen[a], std::min(W - 1, cut - a) + 1);
    }
    rowoff.assign(LU + 3, 0);
    rowoff[0] = 0;                 // guard row a = -1: as long as row 0
    rowoff[1] = rowlen[0] + 1;
    for (int a = 0; a <= LU; a++) rowoff[a + 2] = rowoff[a + 1] + rowlen[a] + 1;
    return rowoff[LU + 2];
}

struct HostGeometry {
    std::vector<asora::PlaneInfo> planes;  // [variant][nplanes]
    std::vector<uint2> meta;
    std::vector<double2> geo;
    int vmap[8] = {0};
    int nplanes = 0, nvariants = 0, entries_max = 0;
};

// Pure host arithmetic (no device needed): also what asora_geometry_census counts.
void make_geometry(int N, double R, double dr, int q_max, int last_l, int last_r, HostGeometry &G) {
    (void)N;
    const int nmaxmax = std::min(std::max(last_l, last_r), q_max);
    const int nplanes = nmaxmax + 1;
    const int cap = plane_extent(q_max, last_l, last_r);
    const double R2 = R * R, dr2 = dr * dr;
    std::vector<asora::PlaneInfo> &planes = G.planes;
    std::vector<uint2> &meta = G.meta;
    std::vector<double2> &geo = G.geo;
    std::vector<std::array<int, 3>> keys;  // (LU, LV, order)
    std::vector<int> rowoff;
    int entries_max = 0;
    for (int code = 0; code < 8; code++) {
        const int LU = std::min((code & 4) ? last_l : last_r, cap), LV = std::min((code & 2) ? last_l : last_r, cap);
        const int order = code & 1;  // 0: z-sector (U2+V2)+P2, 1: x/y-sector (U2+P2)+V2
        int found = -1;
        for (size_t k = 0; k < keys.size(); k++)
            if (keys[k][0] == LU && keys[k][1] == LV && keys[k][2] == order) found = (int)k;
        if (found >= 0) { G.vmap[code] = found; continue; }
        G.vmap[code] = (int)keys.size();
        keys.push_back({LU, LV, order});
        const int entries = packed_layout(q_max, nmaxmax, LU, LV, rowoff);
        entries_max = std::max(entries_max, entries);
        auto pos = [&](int a, int b) { return (uint32_t)(rowoff[a + 1] + b + 1); };
        planes.resize(planes.size() + nplanes);
        asora::PlaneInfo *pl = planes.data() + planes.size() - nplanes;
        pl[0] = {pos(0, 0), 0, (uint32_t)entries};
        for (int n = 1; n <= nmaxmax; n++) {
            const int cut = q_max - n;
            const int H = std::min(std::min(n, LU), cut) + 1, W = std::min(std::min(n, LV), cut) + 1;
            const double dn = (double)n;
            const double P2 = (dr * dn) * (dr * dn);
            pl[n].off = (uint32_t)meta.size();  // (alam = (n - 1/2) / n, raytracing.f90:683, is formed on the device)
            for (int pass = 0; pass < 2; pass++) {
                for (int a = 0; a < H; a++) {
                    const int bend = std::min(W - 1, cut - a);
                    const double U2 = (dr * (double)a) * (dr * (double)a);
                    for (int b = 0; b <= bend; b++) {
                        const double V2 = (dr * (double)b) * (dr * (double)b);
                        const double dist2 = order == 0 ? (U2 + V2) + P2 : (U2 + P2) + V2;
                        const bool stop = dist2 / dr2 > R2;
                        if (stop != (pass == 1)) continue;
                        const double path = std::sqrt((double)(a * a + b * b) / (dn * dn) + 1.0);
                        const double pd = path * dr;
                        const uint32_t flags = (a == 0 ? asora::FLAG_A0 : 0u) | (b == 0 ? asora::FLAG_B0 : 0u) |
                                               (a == n ? asora::FLAG_AN : 0u) | (b == n ? asora::FLAG_BN : 0u);
                        uint2 m;
                        m.x = (uint32_t)a | ((uint32_t)b << asora::META_COORD_BITS) | (flags << asora::META_FLAG_SHIFT);
                        m.y = pos(a - 1, b - 1) | (pos(a, b - 1) << 16);
                        double2 gq;
                        gq.x = pd;
                        gq.y = dist2 * pd * asora::FOURPI;
                        meta.push_back(m);
                        geo.push_back(gq);
                    }
                }
                if (pass == 0) pl[n].cntA = (uint32_t)meta.size() - pl[n].off;
            }
            pl[n].cntT = (uint32_t)meta.size() - pl[n].off;
        }
    }
    if (meta.empty()) { meta.push_back(uint2{0, 0}); geo.push_back(double2{0, 0}); }
    G.nplanes = nplanes;
    G.nvariants = (int)keys.size();
    G.entries_max = entries_max;
}

int build_geometry(int N, double R, double dr, int q_max, int last_l, int last_r, hipStream_t st) {
    if (S.geo_valid && S.geo_N == N && S.geo_R == R && S.geo_dr == dr) return 0;
    if (S.geo_valid) HIP_TRY(hipStreamSynchronize(st));  // tables may still be in use by a previous launch
    free_geometry();
    HostGeometry G;
    make_geometry(N, R, dr, q_max, last_l, last_r, G);
    HIP_TRY(hipMalloc(&S.planes_dev, G.planes.size() * sizeof(asora::PlaneInfo)));
    HIP_TRY(hipMalloc(&S.cell_meta_dev, G.meta.size() * sizeof(uint2)));
    HIP_TRY(hipMalloc(&S.cell_geo_dev, G.geo.size() * sizeof(double2)));
    // synchronous copies: the host vectors die at the end of this function
    HIP_TRY(hipMemcpy(S.planes_dev, G.planes.data(), G.planes.size() * sizeof(asora::PlaneInfo), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(S.cell_meta_dev, G.meta.data(), G.meta.size() * sizeof(uint2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(S.cell_geo_dev, G.geo.data(), G.geo.size() * sizeof(double2), hipMemcpyHostToDevice));
    for (int i = 0; i < 8; i++) S.geo_vmap[i] = G.vmap[i];
    S.geo_nmax_rate = 0;
    for (size_t i = 0; i < G.planes.size(); i++)
        if (G.planes[i].cntA > 0) S.geo_nmax_rate = std::max(S.geo_nmax_rate, (int)(i % (size_t)G.nplanes));
    S.geo_visits = S.geo_in_sphere = -1;
    S.geo_N = N; S.geo_R = R; S.geo_dr = dr; S.geo_nplanes = G.nplanes; S.geo_entries = G.entries_max; S.geo_valid = true;
    return 0;
}

template <bool L, bool H, int T, bool WS = true, int QD = 0, int SS = 0, bool SAT = false>
int launch_sweep_tab(const asora::SweepArgs &A, int grid, size_t lds, hipStream_t st) {
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)asora::sweep_tab_kernel<L, H, T, WS, QD, SS, SAT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((asora::sweep_tab_kernel<L, H, T, WS, QD, SS, SAT>), dim3(grid), dim3(T), lds, st, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool L, bool H>
int launch_sweep(const asora::SweepArgs &A, int grid, size_t lds, hipStream_t st) {
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)asora::sweep_kernel<L, H>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((asora::sweep_kernel<L, H>), dim3(grid), dim3(asora::SWEEP_THREADS), lds, st, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Key of a cell on the 3-D Hilbert curve of 2^bits cells per axis (Skilling's transpose form, AIP Conf. Proc.
// 707, 381 (2004)): consecutive keys are face neighbours.
uint64_t hilbert_key(uint32_t x, uint32_t y, uint32_t z, int bits) {
    uint32_t X[3] = {x, y, z};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; i++) X[i] ^= t;
    uint64_t key = 0;
    for (int b = bits - 1; b >= 0; b--)
        for (int i = 0; i < 3; i++) key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
}
uint64_t morton_key(uint32_t x, uint32_t y, uint32_t z, int bits) {
    uint64_t key = 0;
    for (int b = bits - 1; b >= 0; b--) key = (key << 3) | (((x >> b) & 1u) << 2) | (((y >> b) & 1u) << 1) | ((z >> b) & 1u);
    return key;
}

// The sweep works the source list off in order, ~40 sources in flight at a time on the whole chip, and the rate
// grid is a SUM over sources: the order is free.  Sorting the list along a space-filling curve makes the
// sources in flight neighbours, so their neutral-density reads and rate atomics meet in L2 / Infinity Cache
// (256^3, 1e4 random sources, R = 60: 119.5 -> 105.5 ms).  Done once per uploaded list (and per num_src, since
// do_all_sources may use a prefix of the list); coldensh_out stays that of the caller's LAST source.
void sweep_order(const int32_t *pos, int num_src, int N, int order, int *perm) {
    for (int i = 0; i < num_src; i++) perm[i] = i;
    if (order == 0 || num_src < 2) return;
    int bits = 1;
    while ((1 << bits) < N) bits++;
    std::vector<uint64_t> key((size_t)num_src);
    for (int i = 0; i < num_src; i++) {
        const uint32_t x = wrapped(pos[3 * (size_t)i], N), y = wrapped(pos[3 * (size_t)i + 1], N), z = wrapped(pos[3 * (size_t)i + 2], N);
        key[i] = order == 1 ? morton_key(x, y, z, bits) : hilbert_key(x, y, z, bits);
    }
    std::stable_sort(perm, perm + num_src, [&](int a, int b) { return key[a] < key[b]; });
}

int stage_sources(int num_src) {
    if (S.staged_n == num_src && S.staged_order == S.opt.source_order) return 0;
    std::vector<int> perm((size_t)std::max(num_src, 1));
    sweep_order(S.host_pos.data(), num_src, S.N, S.opt.source_order, perm.data());
    std::vector<int32_t> pos(3 * (size_t)num_src);
    std::vector<double> flux((size_t)num_src);
    S.staged_last = -1;
    for (int i = 0; i < num_src; i++) {
        const size_t j = (size_t)perm[i];
        pos[3 * (size_t)i] = S.host_pos[3 * j];
        pos[3 * (size_t)i + 1] = S.host_pos[3 * j + 1];
        pos[3 * (size_t)i + 2] = S.host_pos[3 * j + 2];
        flux[i] = S.host_flux[j];
        if (perm[i] == num_src - 1) S.staged_last = i;
    }
    if (num_src > 0) {  // synchronous copies: nothing may still read the previous staging, nothing outlives the vectors
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(S.src_pos, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(S.src_flux, flux.data(), flux.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    S.staged_n = num_src;
    S.staged_order = S.opt.source_order;
    return 0;
}

// Common implementation on device pointers.
int do_all_sources_impl(double R, double *cdh_out_dev, double sig, double dr, const double *xh_dev,
                        double *phi_out_dev, double *heat_out_dev, int num_src, int m1, double minlogtau,
                        double dlogtau, int num_tau, hipStream_t st) {
    if (!S.init) return fail(2, "asora_do_all_sources: device not initialised (call asora_device_init)");
    if (m1 != S.N) return fail(3, "asora_do_all_sources: m1=%d differs from device_init N=%d", m1, S.N);
    if (!S.thin || !S.thick) return fail(4, "asora_do_all_sources: photo tables not uploaded");
    if (num_tau < 1 || num_tau > S.table_len)
        return fail(5, "asora_do_all_sources: num_tau=%d exceeds the uploaded table length %d", num_tau, S.table_len);
    if (num_src < 0 || num_src > S.num_src)
        return fail(6, "asora_do_all_sources: num_src=%d but only %d sources were uploaded", num_src, S.num_src);
    if (heat_out_dev && (!S.hthin || !S.hthick || num_tau > S.heat_table_len))
        return fail(7, "asora_do_all_sources_heat: heating tables not uploaded (or shorter than num_tau)");
    if (!(dr > 0.0) || !(R >= 0.0)) return fail(8, "asora_do_all_sources: need dr > 0 and R >= 0");
    if (!xh_dev || !phi_out_dev) return fail(9, "asora_do_all_sources: null xh_av / phi_ion pointer");
    if (int rc = stage_sources(num_src)) return rc;

    const int N = S.N;
    const size_t bytes = S.cells * sizeof(double);
    const bool want_heat = heat_out_dev != nullptr;
    const bool want_cdh = cdh_out_dev != nullptr;
    const dim3 tb(32, 8), tg((N + 31) / 32, (N + 31) / 32, N);

    asora::SweepArgs A;
    std::memset(&A, 0, sizeof A);
    A.src_pos = S.src_pos; A.src_flux = S.src_flux;
    A.thick = S.thick; A.thin = S.thin; A.hthick = S.hthick; A.hthin = S.hthin;
    A.log10_table = S.log10_table;
    A.ntasks = (long long)asora::SECTORS * num_src;  // (a quarter of that for the four-quadrant kernel, set below)
    A.nsrc = num_src;
    A.src_begin = 0;
    A.cdh_src = want_cdh ? S.staged_last : -1;
    A.N = N;
    A.q_max = compute_q_max(R, N);
    A.last_r = N / 2 - 1 + N % 2;  // src/asora/raytracing.cu:140-141
    A.last_l = N / 2;
    A.num_tau = num_tau;
    A.R = R; A.sig = sig; A.dr = dr; A.minlogtau = minlogtau; A.dlogtau = dlogtau;
    A.rdlogtau = 1.0 / dlogtau;
    // "skip_saturated": from tau_sat on, 1 + (log10(tau) - minlogtau) / dlogtau >= num_tau, i.e. photo_lookuptable clamps to
    // the last entry (photorates.f90:140-146).  The margin of 1e-6 table steps is ~1e9 times the error of the device's
    // log10 and division, so a lane that passes the test clamps for certain.
    A.tau_sat = std::pow(10.0, minlogtau + ((double)num_tau - 1.0 + 1.0e-6) * dlogtau) * (1.0 + 1.0e-12);
    // widest plane any sector stages: max_n min(n, lim, q_max - n) + 1, lim = max(last_l, last_r)
    const int wmax = plane_extent(A.q_max, A.last_l, A.last_r);
    A.S = wmax + 2;  // b = -1 guard .. wmax
    A.plane_elems = (wmax + 2) * A.S;
    A.wtab = wmax + 1;
    const size_t lds_log = asora::LOG10_TABLE_DOUBLES * sizeof(double);  // log10 table, always in LDS
    const int nmaxmax = std::min(std::max(A.last_l, A.last_r), A.q_max);
    // the table-driven kernel needs 32-bit cell offsets, 10-bit in-plane coordinates and 16-bit staging positions
    bool use_tab = !S.opt.force_generic && S.cells < (1ULL << 28) && wmax < 1024 &&
                   8 * geometry_cells_per_variant(A.q_max, std::max(A.last_l, A.last_r), nmaxmax) <= S.opt.max_table_cells;
    S.rec.notice.clear();
    if (!use_tab) {
        char buf[256];
        if (S.opt.force_generic) std::snprintf(buf, sizeof buf, "table-free kernel: option force_generic");
        else if (S.cells >= (1ULL << 28))
            std::snprintf(buf, sizeof buf, "table-free fallback kernel (about 3x slower): %d^3 cells >= 2^28, the table-driven "
                          "kernel addresses the grids with 32-bit byte offsets (N <= 645)", N);
        else if (wmax >= 1024) std::snprintf(buf, sizeof buf, "table-free fallback kernel: plane extent %d >= 1024", wmax);
        else std::snprintf(buf, sizeof buf, "table-free fallback kernel: geometry tables beyond option max_table_cells = %lld",
                           S.opt.max_table_cells);
        S.rec.notice = buf;
    }
    if (use_tab && num_src > 0) {
        if (int rc2 = build_geometry(N, R, dr, A.q_max, A.last_l, A.last_r, st)) return rc2;
        if (S.geo_entries > 65535) {
            use_tab = false;
            S.rec.notice = "table-free fallback kernel: more than 65535 staged entries per sector (16-bit staging positions)";
        }
    }
    // Grid layouts.  Usually two: the neutral density and the rate accumulators exist as [i][j][k] and as the transposed
    // [i][k][j], in which the z-dominant sectors' in-plane coordinate is contiguous, and a merge pass adds the two rate
    // grids into the caller's -- 9 N^3 doubles of streaming traffic per call whatever the list holds (256^3: 0.23 ms).
    // A short call (few sources or a small radius: option "single_max_visits") runs on ONE layout instead: the
    // z-dominant sectors address the [i][j][k] grids with a stride (their rows are short at such radii anyway), the sweep
    // adds straight into the caller's grids, and what is left per call is one pass that forms the neutral density and
    // zeroes the outputs -- 4 N^3 doubles.  Measured at 256^3 (profiles/small_r_probe.py, two layouts -> one): the fixed
    // passes 0.23 -> 0.11 ms per call; the sweep itself R = 3 0.133 -> 0.116 ms and R = 5 0.268 -> 0.251 ms per 1e4 sources
    // (faster: no second grid to miss in), R = 10 +15 %, R = 15 +30 %, R = 20 +46 %, R = 30 +65 % (64 lanes of a z-sector
    // in 64 cache lines) -- about 4e-3 ns per visit against 5 N^3 doubles of traffic saved, so the automatic choice
    // takes one layout up to 1.2 visits per cell of the mesh (256^3: 2e7 visits per call) at the radii of the
    // multi-sector kernels (R <~ 46; the long rows of larger radii lose more: 128^3 full box, one source, 0.23 -> 0.28 ms),
    // and whenever q_max <= 10 (R <= 5.7 cells).
    if (use_tab && num_src > 0 && S.geo_visits < 0) (void)asora_visit_count(R, N, &S.geo_visits, &S.geo_in_sphere);
    const bool single = use_tab && num_src > 0 &&
                        (S.opt.single_layout > 0 ||
                         (S.opt.single_layout == 0 && S.geo_entries <= S.opt.quad_max_entries &&
                          (S.geo_visits * (long long)num_src <=
                               (S.opt.single_max_visits > 0 ? S.opt.single_max_visits : (long long)(1.2 * (double)S.cells)) ||
                           A.q_max <= 10)));
    if (single) {
        const bool vec = ((reinterpret_cast<uintptr_t>(S.n_dev) | reinterpret_cast<uintptr_t>(xh_dev) |
                           reinterpret_cast<uintptr_t>(phi_out_dev) | reinterpret_cast<uintptr_t>(heat_out_dev) |
                           reinterpret_cast<uintptr_t>(cdh_out_dev)) & 15) == 0 && S.cells % 2 == 0;
        const int pgrid = (int)std::min<size_t>((S.cells / (vec ? 2 : 1) + 255) / 256, (size_t)S.num_cu * 64);
        if (vec) hipLaunchKernelGGL(asora::prep_single_kernel<true>, dim3(pgrid), dim3(256), 0, st, S.n_dev, xh_dev, S.nhiA,
                                    phi_out_dev, heat_out_dev, cdh_out_dev, S.cells);
        else hipLaunchKernelGGL(asora::prep_single_kernel<false>, dim3(pgrid), dim3(256), 0, st, S.n_dev, xh_dev, S.nhiA,
                                phi_out_dev, heat_out_dev, cdh_out_dev, S.cells);
        HIP_TRY(hipGetLastError());
        A.single = 1;
        A.nhiA = S.nhiA; A.phiA = phi_out_dev; A.heatA = heat_out_dev; A.cdhA = cdh_out_dev;
    } else {
        if (want_heat) {
            if (int rc = alloc_grid(S.heatA)) return rc;
            if (int rc = alloc_grid(S.heatB)) return rc;
        }
        if (want_cdh) {
            if (int rc = alloc_grid(S.cdhA)) return rc;
            if (int rc = alloc_grid(S.cdhB)) return rc;
        }
        // (the two rate accumulators are cleared by the same pass)
        hipLaunchKernelGGL(asora::prep_nhi_kernel, tg, tb, 0, st, S.n_dev, xh_dev, S.nhiA, S.nhiB, S.phiA, S.phiB, N);
        HIP_TRY(hipGetLastError());
        if (want_heat) {
            HIP_TRY(hipMemsetAsync(S.heatA, 0, bytes, st));
            HIP_TRY(hipMemsetAsync(S.heatB, 0, bytes, st));
        }
        if (want_cdh) {
            HIP_TRY(hipMemsetAsync(S.cdhA, 0, bytes, st));
            HIP_TRY(hipMemsetAsync(S.cdhB, 0, bytes, st));
        }
        A.nhiA = S.nhiA; A.nhiB = S.nhiB; A.phiA = S.phiA; A.phiB = S.phiB;
        A.heatA = want_heat ? S.heatA : nullptr; A.heatB = want_heat ? S.heatB : nullptr;
        A.cdhA = want_cdh ? S.cdhA : nullptr; A.cdhB = want_cdh ? S.cdhB : nullptr;
    }
    size_t lds_stage, lds_stage_c = 0, lds_fixed;
    bool quad = false;
    int qshift = 0, sshift = 0;
    if (use_tab) {
        // ONE packed array of (column density, weight) pairs, updated in place, + per-row/column tables
        A.planes = S.planes_dev; A.cell_meta = S.cell_meta_dev; A.cell_geo = S.cell_geo_dev;
        A.nplanes = S.geo_nplanes; A.cw_entries = S.geo_entries;
        A.rate_only = S.opt.skip_unobserved ? 1 : 0; A.nmax_rate = S.geo_nmax_rate;
        for (int i = 0; i < 8; i++) A.vmap[i] = S.geo_vmap[i];
        // four-quadrant / eight-octant kernel (small radii): the sectors a workgroup takes side by side must share
        // their geometry variant (and, for the octants, their planes), and their staging arrays must fit in LDS
        quad = S.opt.quad_kernel >= 0 && num_src > 0 && S.opt.lds_stage_limit <= 0;
        for (int i = 2; i < 8; i++) quad = quad && S.geo_vmap[i] == S.geo_vmap[i & 1];
        if (quad && S.opt.quad_kernel == 0)  // automatic: small radii, and no workgroup size forced that the variant lacks
            quad = A.cw_entries <= S.opt.quad_max_entries &&
                   (S.opt.wg_threads == 0 || S.opt.wg_threads == 64 || S.opt.wg_threads == 128 || S.opt.wg_threads == 256);
        qshift = quad ? 2 : 0;
        if (quad && A.q_max <= std::min(A.last_l, A.last_r) &&
            (S.opt.quad_kernel == 2 || (S.opt.quad_kernel == 0 && A.cw_entries <= S.opt.oct_max_entries)))
            qshift = 3;
        auto lds_need = [&](int qs, int ss, bool q) {
            const size_t nq = (size_t)1 << qs, nsrc = (size_t)1 << ss, sw = (size_t)(q ? 2 : 1) * A.wtab * nsrc;
            lds_stage = nsrc * nq * (size_t)A.cw_entries * 2 * sizeof(double);
            lds_stage_c = nsrc * nq * (size_t)((A.cw_entries + 1) & ~1) * sizeof(double);  // column densities only
            lds_fixed = (size_t)(5 * sw + (sw & 1) + asora::plane_table_doubles(A.nplanes) + (ss ? nsrc : 0)) * sizeof(double) + lds_log;
        };
        for (;;) {
            lds_need(qshift, 0, quad);
            if (!quad || std::min(lds_stage, S.opt.compact_staging >= 0 ? lds_stage_c : lds_stage) + lds_fixed +
                                 (size_t)S.opt.lds_pad <= (size_t)S.max_lds)
                break;
            if (S.opt.quad_kernel > 0)
                return fail(13, "asora_do_all_sources: the multi-sector kernel does not fit in LDS at this radius");
            if (qshift == 3) qshift = 2;
            else { quad = false; qshift = 0; }
        }
        // Several consecutive sources of the (curve-sorted) list per workgroup: the smallest radii, whose planes leave
        // most lanes of a wavefront idle and whose tasks are dominated by their set-up.  256 threads, (column density,
        // weight) pairs; automatic while four such workgroups fit on a CU (option "pack_max_lds").
        if (quad && S.opt.src_pack >= 0 && S.opt.compact_staging <= 0 && (S.opt.wg_threads == 0 || S.opt.wg_threads == 256)) {
            // (measured, 1e4 sources at 256^3, unpacked -> four sources per workgroup: R = 3 0.21 -> 0.13 ms, R = 5 0.30 -> 0.27,
            // R = 8 0.76 -> 0.66; R = 10 1.08 -> 1.17: from there on a phase of one source keeps the wavefronts busy and
            // the packed lanes' rate atomics meet in the same cells)
            // eight sources per workgroup pay up to ~32 staged entries (R <= 5: 1e5 sources 2.12 -> 1.87 ms, R = 3 0.175 -> 0.133)
            for (int ss = (S.opt.src_pack > 0 ? S.opt.src_pack : (A.cw_entries <= 32 ? 3 : 2)); ss >= 2; ss--) {
                if (qshift + ss > 6) continue;
                lds_need(qshift, ss, true);
                const size_t full = lds_stage + lds_fixed + (size_t)S.opt.lds_pad;
                // (a forced packing that does not fit in LDS degrades to fewer sources per workgroup)
                if (S.opt.src_pack > 0 ? full <= (size_t)S.max_lds
                                   : (full <= (size_t)S.opt.pack_max_lds && A.cw_entries <= S.opt.pack_max_entries &&
                                      num_src >= (8 << ss))) { sshift = ss; break; }
            }
            lds_need(qshift, sshift, true);
        }
        if (quad) A.ntasks = (long long)(asora::SECTORS >> qshift) * ((num_src + (1 << sshift) - 1) >> sshift);
    } else {
        lds_stage = 2 * (size_t)A.plane_elems * sizeof(double);  // table-free kernel: two square planes
        lds_fixed = lds_log;
    }
    // Where the staged entries live and how large a workgroup is (table-driven kernel): what counts is ~16
    // waves per CU, in the SMALLEST workgroups that reach it (small planes leave the waves of a large
    // workgroup idle, and its barriers span more waves) -- 64 threads at R <~ 25 cells, 128 up to ~45, 192/256
    // up to ~70, 512 x 2 up to 80 KB (R <~ 92 with pairs, <~ 130 with column densities only), ONE 1024-thread
    // workgroup with the whole LDS up to R ~ 180; beyond that (or above the "lds_stage_limit" option) the entries
    // live in a global scratch, 256 threads.
    // Measured at 256^3, 1000 sources, updates/s: R = 20: 64 thr 5.4e10, 256 thr 3.2e10; R = 40: 128 thr 9.4e10,
    // 256 thr 8.7e10; R = 60: 256 thr 1.21e11, 512 thr 1.0e11; R = 80: 256 thr 9.4e10, 512 thr 1.26e11;
    // R = 100: 256 thr 5.9e10, global 7.3e10, 1024 thr 1.12e11; R = 120: global 7.2e10, 1024 thr 1.15e11.
    const int lds_limit = S.opt.lds_stage_limit > 0 ? std::min(S.max_lds, S.opt.lds_stage_limit)
                                                : std::max(S.max_lds / 2, std::min(S.max_lds, 65536));
    // Column-density-only staging (8 B per entry, the four weights recomputed per cell: +13 % instructions) once
    // the (column density, weight) pairs would leave room for ONE workgroup per CU or none: two 512-thread
    // workgroups beat one of 1024 threads (256^3, 1000 sources: R = 100 1.18e11 -> 1.31e11, R = 128 1.19e11 ->
    // 1.25e11 updates/s; equal at R = 80), and the LDS range doubles (R = 140: global 7.2e10 -> 1.04e11,
    // R = 160: 7.0e10 -> 9.9e10; 256^3 full box and 512^3 at R = 150 fit).
    bool compact = false;
    if (use_tab && S.opt.lds_stage_limit <= 0 && S.opt.compact_staging >= 0 && !sshift &&  // (the packed kernels stage pairs)
        lds_stage_c + lds_fixed + (size_t)S.opt.lds_pad <= (size_t)S.max_lds &&
        (S.opt.compact_staging > 0 || lds_stage + lds_fixed + (size_t)S.opt.lds_pad > (size_t)S.max_lds / (quad ? 3 : 2) ||
         // two workgroups per CU with pairs, four of 256 threads with column densities only: the latter wins while
         // it stays below ~33 KB (R = 73: 79.5 -> 76.1 ms per 4000 sources; equal at R = 76; pairs win from R = 80)
         (!quad && S.opt.wg_threads == 0 && lds_stage + lds_fixed + (size_t)S.opt.lds_pad > (size_t)S.max_lds / 3 &&
          lds_stage_c + lds_fixed + (size_t)S.opt.lds_pad <= 33 * 1024))) {
        compact = true;
        lds_stage = lds_stage_c;
    }
    bool use_lds = lds_stage + lds_fixed + (size_t)S.opt.lds_pad <= (size_t)lds_limit;
    int threads = asora::SWEEP_THREADS;
    if (use_tab) {
        const size_t full = lds_stage + lds_fixed + (size_t)S.opt.lds_pad;
        if (sshift) {  // packed sources: 256 threads, pairs in LDS (checked above)
            threads = 256;
            use_lds = true;
        } else if (S.opt.wg_threads == 64 || S.opt.wg_threads == 128 || S.opt.wg_threads == 192 || S.opt.wg_threads == 256 ||
            S.opt.wg_threads == 512 || S.opt.wg_threads == 1024) {
            threads = S.opt.wg_threads;
            if (compact && threads < 256 && !quad) return fail(13, "asora_do_all_sources: compact staging needs wg_threads >= 256");
            if (threads != 256 || compact || quad) {
                if (full > (size_t)S.max_lds) return fail(13, "asora_do_all_sources: wg_threads=%d needs %zu B of LDS", threads, full);
                use_lds = true;
            }
        } else if (S.opt.lds_stage_limit <= 0 && full <= (size_t)S.max_lds) {
            // smallest workgroup that still puts >= 14 of the 16 possible waves on a CU, else the one with the most waves
            static const int sizes[] = {64, 128, 192, 256, 512, 1024};  // (320: measured 30 % slower than 256 / 512)
            const int fit = (int)((size_t)S.max_lds / ((full + 1023) & ~(size_t)1023));
            int best_waves = 0;
            for (int t : sizes) {
                if (compact && t < 256 && !quad) continue;
                if (quad && (t == 192 || t > 256)) continue;
                const int waves = std::min(fit, 16 / (t / 64)) * (t / 64);
                if (waves > best_waves) { best_waves = waves; threads = t; }
                // (256 threads x 3 workgroups beat 512 x 2: R = 64 51.3 vs 57.1 ms, R = 70 65.2 vs 71.8 per 4000 sources)
                if (waves >= (t >= 256 ? 12 : 14)) break;
            }
            // few sources (single-source tests: 24 tasks): the chip is not full, so larger workgroups put
            // more waves on each sector (128^3 full box, 1 source: 0.40 ms at 256 threads, 0.25 ms at 1024)
            for (int t : sizes)
                if (t > threads && !(quad && (t == 192 || t > 256)) && A.ntasks * (t / 64) <= (long long)S.num_cu * 16) threads = t;
            use_lds = true;
        }
    }
    size_t lds = (use_lds ? lds_stage : 0) + lds_fixed + (size_t)S.opt.lds_pad;
    // Multi-sector kernels: the lanes of a cell read the same position of 4 / 8 staging arrays, E entries apart.  With
    // E a multiple of 8 (16) the four (eight) reads of a cell fall into the same LDS banks; E = 4 mod 8 (2 mod 4) lets
    // the 16 lanes of one LDS pass tile its banks.  Measured at 1e4 sources: R = 25 (E = 400) 9.12 -> 8.63 ms, R = 30
    // 14.66 -> 14.39 ms; a forced multiple of 32 costs 2 - 6 % at R = 28 ... 45.  The array is padded only while the
    // workgroups per CU stay the same (option "stride_residue": -1 this rule, -2 as the geometry gives it, >= 0: E modulo 32).
    // (Packed sources: 32 - 64 arrays, one per lane of a cell; only a multiple of 16 hurts -- R = 5: 0.273 -> 0.326 ms.)
    if (use_tab && use_lds && quad && S.opt.stride_residue != -2) {
        const int mod = S.opt.stride_residue >= 0 ? 32 : (qshift == 3 ? 4 : 8);
        const int want = S.opt.stride_residue >= 0 ? S.opt.stride_residue % 32 : (qshift == 3 ? 2 : 4);
        int delta = ((want - A.cw_entries) % mod + mod) % mod;
        if (sshift && S.opt.stride_residue < 0) delta = A.cw_entries % 16 == 0 ? 1 : 0;
        // (the size is formed from the padded entry count exactly as the kernel lays it out: column-density-only
        // staging rounds the array of each sector up to an even number of doubles)
        auto staged_bytes = [&](int entries) {
            const size_t arrays = (size_t)1 << (qshift + sshift);
            return arrays * (compact ? (size_t)((entries + 1) & ~1) * sizeof(double) : (size_t)entries * 2 * sizeof(double));
        };
        const size_t lds2 = lds - staged_bytes(A.cw_entries) + staged_bytes(A.cw_entries + delta);
        auto per_cu = [&](size_t b) { return (size_t)S.max_lds / ((b + 1023) & ~(size_t)1023); };
        if (delta && lds2 <= (size_t)S.max_lds && per_cu(lds2) == per_cu(lds)) {
            A.cw_entries += delta;
            lds = lds2;
        }
    }
    if (lds > (size_t)S.max_lds) return fail(13, "asora_do_all_sources: per-row tables (%zu B) exceed LDS", lds);

    S.timed = false;
    int grid, rc = 0;
    if (num_src > 0) {
        if (use_lds) {
            grid = (int)std::min<long long>(A.ntasks, 0x7fffffffLL);
        } else {
            grid = (int)std::min<long long>(A.ntasks, (long long)S.num_cu * 4);
            const size_t need = (size_t)grid * lds_stage;
            if (need > S.gscratch_bytes) {
                if (S.gscratch) (void)hipFree(S.gscratch);
                S.gscratch = nullptr;
                S.gscratch_bytes = 0;
                HIP_TRY(hipMalloc(&S.gscratch, need));
                S.gscratch_bytes = need;
            }
            A.gscratch = S.gscratch;
        }
        S.rec.last_tab = use_tab ? 1 : 0;
        HIP_TRY(hipEventRecord(S.ev0, st));
        if (use_tab) {
            constexpr int T = asora::SWEEP_THREADS;
            // ("skip_saturated": the LDS-staged single-sector kernels of 256 / 512 / 1024 threads have a variant with the test
            // compiled in -- the radii at which a source's light can saturate the table; the others ignore the option)
            const bool sat = S.opt.skip_saturated && !quad && use_lds && threads >= 256;
#define ASORA_LAUNCH_TAB(L, TH, WS) \
    (sat && L && TH >= 256 ? ASORA_LAUNCH_SAT(TH, WS) : ASORA_LAUNCH_PLAIN(L, TH, WS))
#define ASORA_LAUNCH_PLAIN(L, TH, WS) \
    (want_heat ? launch_sweep_tab<L, true, TH, WS>(A, grid, lds, st) : launch_sweep_tab<L, false, TH, WS>(A, grid, lds, st))
#define ASORA_LAUNCH_SAT(TH, WS) \
    (want_heat ? launch_sweep_tab<true, true, (TH >= 256 ? TH : 256), WS, 0, 0, true>(A, grid, lds, st) \
               : launch_sweep_tab<true, false, (TH >= 256 ? TH : 256), WS, 0, 0, true>(A, grid, lds, st))
#define ASORA_LAUNCH_QUAD(TH, WS, QD) \
    (want_heat ? launch_sweep_tab<true, true, TH, WS, QD>(A, grid, lds, st) \
               : launch_sweep_tab<true, false, TH, WS, QD>(A, grid, lds, st))
#define ASORA_LAUNCH_QUAD_T(TH) \
    (qshift == 3 ? (compact ? ASORA_LAUNCH_QUAD(TH, false, 3) : ASORA_LAUNCH_QUAD(TH, true, 3)) \
                 : (compact ? ASORA_LAUNCH_QUAD(TH, false, 2) : ASORA_LAUNCH_QUAD(TH, true, 2)))
#define ASORA_LAUNCH_PACK(QD, SS) \
    (want_heat ? launch_sweep_tab<true, true, 256, true, QD, SS>(A, grid, lds, st) \
               : launch_sweep_tab<true, false, 256, true, QD, SS>(A, grid, lds, st))
            if (quad && sshift) {
                if (qshift == 3 && sshift == 3) rc = ASORA_LAUNCH_PACK(3, 3);
                else if (qshift == 3 && sshift == 2) rc = ASORA_LAUNCH_PACK(3, 2);
                else if (qshift == 2 && sshift == 3) rc = ASORA_LAUNCH_PACK(2, 3);
                else if (qshift == 2 && sshift == 2) rc = ASORA_LAUNCH_PACK(2, 2);
                else return fail(13, "asora_do_all_sources: unsupported packing %d/%d", qshift, sshift);
            } else if (quad) {
                if (threads == 64) rc = ASORA_LAUNCH_QUAD_T(64);
                else if (threads == 128) rc = ASORA_LAUNCH_QUAD_T(128);
                else if (threads == 256) rc = ASORA_LAUNCH_QUAD_T(256);
                else return fail(13, "asora_do_all_sources: the multi-sector kernel runs 64, 128 or 256 threads");
            } else if (compact) {
                if (threads == 1024) rc = ASORA_LAUNCH_TAB(true, 1024, false);
                else if (threads == 512) rc = ASORA_LAUNCH_TAB(true, 512, false);
                else rc = ASORA_LAUNCH_TAB(true, T, false);
            } else if (threads == 1024) rc = ASORA_LAUNCH_TAB(true, 1024, true);
            else if (threads == 512) rc = ASORA_LAUNCH_TAB(true, 512, true);
            else if (threads == 192) rc = ASORA_LAUNCH_TAB(true, 192, true);
            else if (threads == 128) rc = ASORA_LAUNCH_TAB(true, 128, true);
            else if (threads == 64) rc = ASORA_LAUNCH_TAB(true, 64, true);
            else if (use_lds) rc = ASORA_LAUNCH_TAB(true, T, true);
            else rc = ASORA_LAUNCH_TAB(false, T, true);
#undef ASORA_LAUNCH_TAB
#undef ASORA_LAUNCH_PLAIN
#undef ASORA_LAUNCH_SAT
#undef ASORA_LAUNCH_QUAD
#undef ASORA_LAUNCH_QUAD_T
#undef ASORA_LAUNCH_PACK
        } else {
            if (use_lds) rc = want_heat ? launch_sweep<true, true>(A, grid, lds, st) : launch_sweep<true, false>(A, grid, lds, st);
            else rc = want_heat ? launch_sweep<false, true>(A, grid, lds, st) : launch_sweep<false, false>(A, grid, lds, st);
        }
        if (rc) return rc;
        HIP_TRY(hipEventRecord(S.ev1, st));
        S.timed = true;
        S.rec.last_grid = grid; S.rec.last_block = use_tab ? threads : asora::SWEEP_THREADS; S.rec.last_lds = (int)lds;
        S.rec.last_is_lds = (use_lds ? (compact ? 2 : 1) : 0) | (qshift << 2) | (sshift << 4) | (single ? 64 : 0);
    }
    if (single) return 0;  // the sweep has added straight into the caller's grids

    hipLaunchKernelGGL(asora::finalize_kernel, tg, tb, 0, st, S.phiA, S.phiB, phi_out_dev, N);
    HIP_TRY(hipGetLastError());
    if (want_heat) {
        hipLaunchKernelGGL(asora::finalize_kernel, tg, tb, 0, st, S.heatA, S.heatB, heat_out_dev, N);
        HIP_TRY(hipGetLastError());
    }
    if (want_cdh) {
        hipLaunchKernelGGL(asora::finalize_kernel, tg, tb, 0, st, S.cdhA, S.cdhB, cdh_out_dev, N);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Host <-> device copies of the host-pointer entry points (the reference's callers hand over pageable numpy buffers:
// 1 x N^3 up, 1 - 2 x N^3 down per call, src/asora/raytracing.cu:130,163-166).  A pageable hipMemcpy is staged through
// page-locked memory by ONE host thread (~16 GB/s measured: 25 ms for the three 134 MB grids of a 256^3 call, a quarter
// of the sweep itself).  Here `copy_threads` host threads (option, default 4; 0: plain hipMemcpy) each stage their
// share of 8 MiB chunks through two page-locked slots of their own and drive the DMA engine on a stream of their
// own, so the staging memcpys run in parallel and overlap the transfers.  The caller's buffers are never registered
// or kept: nothing outlives the call.
// ------------------------------------------------------------------------------------------
struct Stager {
    static constexpr size_t CHUNK = 8u << 20;
    static constexpr int MAXT = 8, SLOTS = 2;
    int nthreads = 0;
    char *pinned = nullptr;  // nthreads * SLOTS * CHUNK bytes
    hipStream_t stream[MAXT] = {};
    hipEvent_t event[MAXT][SLOTS] = {};
    char *slot(int t, int k) const { return pinned + ((size_t)t * SLOTS + k) * CHUNK; }
};
Stager G_stager;
int g_copy_threads = 4;  // option "copy_threads"

void stager_release() {
    Stager &G = G_stager;
    for (int t = 0; t < G.nthreads; t++) {
        if (G.stream[t]) (void)hipStreamDestroy(G.stream[t]);
        for (int k = 0; k < Stager::SLOTS; k++)
            if (G.event[t][k]) (void)hipEventDestroy(G.event[t][k]);
    }
    if (G.pinned) (void)hipHostFree(G.pinned);
    G = Stager();
}

int stager_setup(int want) {
    Stager &G = G_stager;
    want = std::min(want, (int)Stager::MAXT);
    if (G.nthreads == want) return 0;
    stager_release();
    // nthreads counts what stager_release has to visit; a failure half-way gives everything back, so the next call
    // starts from scratch instead of finding nthreads == want with null streams / events
    const int rc = [&]() -> int {
        HIP_TRY(hipHostMalloc((void **)&G.pinned, (size_t)want * Stager::SLOTS * Stager::CHUNK, hipHostMallocDefault));
        G.nthreads = want;
        for (int t = 0; t < want; t++) {
            HIP_TRY(hipStreamCreateWithFlags(&G.stream[t], hipStreamNonBlocking));
            for (int k = 0; k < Stager::SLOTS; k++) HIP_TRY(hipEventCreateWithFlags(&G.event[t][k], hipEventDisableTiming));
        }
        return 0;
    }();
    if (rc) {
        const std::string msg = g_err;
        stager_release();
        g_err = msg;
    }
    return rc;
}

// dir: true = host -> device.  All device work issued before the call must have completed (the worker streams do
// not synchronise with other streams); on return the copy is complete.
int staged_copy(void *dev, void *host, size_t bytes, bool to_device) {
    const int T = g_copy_threads;
    if (T <= 0 || bytes < 4 * Stager::CHUNK) {
        HIP_TRY(to_device ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    if (int rc = stager_setup(T)) return rc;
    const Stager &G = G_stager;
    const size_t nchunks = (bytes + Stager::CHUNK - 1) / Stager::CHUNK;
    const int device = S.device;
    std::atomic<int> err{0};
    auto worker = [&](int t) {
        hipError_t e = hipSetDevice(device);
        auto chunk_len = [&](size_t c) { return std::min(Stager::CHUNK, bytes - c * Stager::CHUNK); };
        size_t k = 0;  // this thread's chunk counter: slot = k % SLOTS
        if (to_device) {
            for (size_t c = (size_t)t; c < nchunks && e == hipSuccess; c += (size_t)G.nthreads, k++) {
                const int sl = (int)(k % Stager::SLOTS);
                if (k >= (size_t)Stager::SLOTS) e = hipEventSynchronize(G.event[t][sl]);  // the slot's previous transfer
                if (e != hipSuccess) break;
                std::memcpy(G.slot(t, sl), (char *)host + c * Stager::CHUNK, chunk_len(c));
                e = hipMemcpyAsync((char *)dev + c * Stager::CHUNK, G.slot(t, sl), chunk_len(c), hipMemcpyHostToDevice, G.stream[t]);
                if (e == hipSuccess) e = hipEventRecord(G.event[t][sl], G.stream[t]);
            }
        } else {
            auto issue = [&](size_t c, int sl) {
                hipError_t x = hipMemcpyAsync(G.slot(t, sl), (char *)dev + c * Stager::CHUNK, chunk_len(c), hipMemcpyDeviceToHost, G.stream[t]);
                return x == hipSuccess ? hipEventRecord(G.event[t][sl], G.stream[t]) : x;
            };
            if ((size_t)t < nchunks) e = issue((size_t)t, 0);
            for (size_t c = (size_t)t; c < nchunks && e == hipSuccess; c += (size_t)G.nthreads, k++) {
                const int sl = (int)(k % Stager::SLOTS);
                const size_t next = c + (size_t)G.nthreads;
                if (next < nchunks) e = issue(next, (sl + 1) % Stager::SLOTS);  // (that slot's memcpy finished an iteration ago)
                if (e == hipSuccess) e = hipEventSynchronize(G.event[t][sl]);
                if (e != hipSuccess) break;
                std::memcpy((char *)host + c * Stager::CHUNK, G.slot(t, sl), chunk_len(c));
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(G.stream[t]);
        if (e != hipSuccess) err.store((int)e);
    };
    // (thread creation can throw std::system_error; nothing may cross the extern "C" boundary: the chunks of a thread
    // that did not start are copied by this one)
    std::vector<std::thread> pool;
    std::vector<int> not_started;
    for (int t = 1; t < G.nthreads; t++) {
        try {
            pool.emplace_back(worker, t);
        } catch (...) {
            not_started.push_back(t);
        }
    }
    worker(0);
    for (int t : not_started) worker(t);
    for (auto &th : pool) th.join();
    if (err.load()) return fail(100 + err.load(), "asora_do_all_sources: staged %s copy failed: %s", to_device ? "upload" : "download",
                                hipGetErrorString((hipError_t)err.load()));
    return 0;
}

int do_all_sources_host(double R, double *coldensh_out, double sig, double dr, const double *xh_av,
                        double *phi_ion, double *phi_heat, int num_src, int m1, double minlogtau,
                        double dlogtau, int num_tau) {
    if (!S.init) return fail(2, "asora_do_all_sources: device not initialised (call asora_device_init)");
    if (m1 != S.N) return fail(3, "asora_do_all_sources: m1=%d differs from device_init N=%d", m1, S.N);
    if (!xh_av || !phi_ion) return fail(9, "asora_do_all_sources: null xh_av / phi_ion pointer");
    ASORA_ON_DEVICE("asora_do_all_sources");
    const size_t bytes = S.cells * sizeof(double);
    if (int rc = alloc_grid(S.out0)) return rc;
    if (coldensh_out) if (int rc = alloc_grid(S.out1)) return rc;
    if (phi_heat) if (int rc = alloc_grid(S.out2)) return rc;
    HIP_TRY(hipDeviceSynchronize());  // (nothing may still read x_dev; the staged copy runs on streams of its own)
    if (int rc = staged_copy(S.x_dev, const_cast<double *>(xh_av), bytes, true)) return rc;  // src/asora/raytracing.cu:130
    int rc = do_all_sources_impl(R, coldensh_out ? S.out1 : nullptr, sig, dr, S.x_dev, S.out0,
                                 phi_heat ? S.out2 : nullptr, num_src, m1, minlogtau, dlogtau, num_tau, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (int rc2 = staged_copy(S.out0, phi_ion, bytes, false)) return rc2;  // raytracing.cu:163-166
    if (coldensh_out) if (int rc2 = staged_copy(S.out1, coldensh_out, bytes, false)) return rc2;
    if (phi_heat) if (int rc2 = staged_copy(S.out2, phi_heat, bytes, false)) return rc2;
    return 0;
}

}  // namespace

extern "C" {

const char *asora_last_error(void) { return g_err.c_str(); }

int asora_device_init(int N, int num_src_par, int mpi_rank, int num_gpus) {
    if (N < 2) return fail(1, "asora_device_init: mesh size N=%d must be >= 2", N);
    if (num_gpus < 1) return fail(1, "asora_device_init: num_gpus=%d must be >= 1", num_gpus);
    if (S.init) asora_device_close();
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(1, "asora_device_init: no HIP device visible");
    const int dev = ((mpi_rank % num_gpus) + num_gpus) % num_gpus;  // src/asora/memory.cu:46
    if (dev >= ndev) return fail(1, "asora_device_init: device %d requested but only %d visible", dev, ndev);
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    S.device = dev;
    S.N = N;
    S.num_src_par = num_src_par;
    S.cells = (size_t)N * N * N;
    S.num_cu = prop.multiProcessorCount;
    // leave a little LDS head-room; a workgroup may use up to the whole 160 KiB of a CU
    S.max_lds = (int)std::min<size_t>(prop.sharedMemPerBlock > 0 ? prop.sharedMemPerBlock : 65536, 160 * 1024);
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
            S.max_lds = std::max(S.max_lds, std::min(v, 160 * 1024));
    }
    S.init = true;  // from here on asora_device_close releases whatever exists
    const int rc = [&]() -> int {
        double **grids[] = {&S.n_dev, &S.x_dev, &S.nhiA, &S.nhiB, &S.phiA, &S.phiB};
        for (double **g : grids) {
            hipError_t e = hipMalloc(g, S.cells * sizeof(double));
            if (e != hipSuccess)
                return fail(100 + (int)e, "asora_device_init: hipMalloc of %zu bytes failed: %s", S.cells * sizeof(double),
                            hipGetErrorString(e));
        }
        double tab[asora::LOG10_TABLE_DOUBLES];
        asora::build_log10_table(tab);
        HIP_TRY(hipMalloc(&S.log10_table, sizeof tab));
        HIP_TRY(hipMemcpy(S.log10_table, tab, sizeof tab, hipMemcpyHostToDevice));
        HIP_TRY(hipEventCreate(&S.ev0));
        HIP_TRY(hipEventCreate(&S.ev1));
        // set-up of the entry point that allocates nothing: this device's copy of the scratch array (each GPU holds
        // its own, at its own address) and a page-locked landing buffer for its three results
        if (int rc2 = noalloc_setup(dev)) return rc2;
        return 0;
    }();
    if (rc) {  // any failure after the first allocation: give everything back, keep the message
        const std::string msg = g_err;
        (void)asora_device_close();
        g_err = msg;
    }
    return rc;
}

int asora_device_index(int *device) {
    if (!device) return fail(9, "asora_device_index: null pointer");
    if (!S.init) return fail(2, "asora_device_index: device not initialised");
    *device = S.device;
    return 0;
}

int asora_device_close(void) {
    if (!S.init) return fail(2, "asora_device_close: device not initialised");
    ASORA_ON_DEVICE("asora_device_close");
    (void)hipDeviceSynchronize();
    double **grids[] = {&S.n_dev, &S.x_dev, &S.nhiA, &S.nhiB, &S.phiA, &S.phiB, &S.heatA, &S.heatB, &S.cdhA,
                        &S.cdhB, &S.out0, &S.out1, &S.out2, &S.thin, &S.thick, &S.hthin, &S.hthick,
                        &S.src_flux, &S.gscratch, &S.log10_table};
    for (double **g : grids) free_ptr(*g);
    if (S.src_pos) (void)hipFree(S.src_pos);
    S.src_pos = nullptr;
    if (S.chem_stats) (void)hipFree(S.chem_stats);
    S.chem_stats = nullptr;
    if (S.chem_stats_pinned) (void)hipHostFree(S.chem_stats_pinned);
    S.chem_stats_pinned = nullptr;
    stager_release();
    for (double *&b : S.chem_buf) free_ptr(b);
    free_geometry();
    if (S.ev0) (void)hipEventDestroy(S.ev0);
    if (S.ev1) (void)hipEventDestroy(S.ev1);
    State fresh;  // options and the last launch record survive a close
    fresh.opt = S.opt;
    fresh.rec = S.rec;
    const int dev = S.device;
    S = fresh;
    S.device = dev;  // (the guard above still restores the caller's device)
    return 0;
}

int asora_density_to_device(const double *ndens, int N) {
    if (!S.init) return fail(2, "asora_density_to_device: device not initialised");
    if (N != S.N) return fail(3, "asora_density_to_device: N=%d differs from device_init N=%d", N, S.N);
    if (!ndens) return fail(9, "asora_density_to_device: null pointer");
    ASORA_ON_DEVICE("asora_density_to_device");
    HIP_TRY(hipMemcpy(S.n_dev, ndens, S.cells * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int asora_density_from_device(const double *ndens_dev, int N, void *stream) {
    if (!S.init) return fail(2, "asora_density_from_device: device not initialised");
    if (N != S.N) return fail(3, "asora_density_from_device: N=%d differs from device_init N=%d", N, S.N);
    if (!ndens_dev) return fail(9, "asora_density_from_device: null pointer");
    ASORA_ON_DEVICE("asora_density_from_device");
    HIP_TRY(hipMemcpyAsync(S.n_dev, ndens_dev, S.cells * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return 0;
}

int asora_photo_table_to_device(const double *thin_table, const double *thick_table, int num_tau) {
    if (!S.init) return fail(2, "asora_photo_table_to_device: device not initialised");
    if (num_tau < 1 || !thin_table || !thick_table) return fail(9, "asora_photo_table_to_device: bad arguments");
    ASORA_ON_DEVICE("asora_photo_table_to_device");
    if (int rc = upload_table(S.thin, thin_table, num_tau)) return rc;
    if (int rc = upload_table(S.thick, thick_table, num_tau)) return rc;
    S.table_len = num_tau;
    return 0;
}

int asora_heat_table_to_device(const double *heat_thin_table, const double *heat_thick_table, int num_tau) {
    if (!S.init) return fail(2, "asora_heat_table_to_device: device not initialised");
    if (num_tau < 1 || !heat_thin_table || !heat_thick_table) return fail(9, "asora_heat_table_to_device: bad arguments");
    ASORA_ON_DEVICE("asora_heat_table_to_device");
    if (int rc = upload_table(S.hthin, heat_thin_table, num_tau)) return rc;
    if (int rc = upload_table(S.hthick, heat_thick_table, num_tau)) return rc;
    S.heat_table_len = num_tau;
    return 0;
}

int asora_source_order(const int32_t *pos, int num_src, int m1, int order, int32_t *perm) {
    if (num_src < 0 || m1 < 1 || order < 0 || order > 2 || (num_src > 0 && (!pos || !perm)))
        return fail(9, "asora_source_order: bad arguments");
    for (int i = 0; i < 3 * num_src; i++)
        if (pos[i] < -MAX_SOURCE_COORD || pos[i] > MAX_SOURCE_COORD)
            return fail(10, "asora_source_order: source %d coordinate %d = %d is not a grid position", i / 3, i % 3, pos[i]);
    sweep_order(pos, num_src, m1, order, perm);
    return 0;
}

int asora_source_data_to_device(const int32_t *pos, const double *flux, int num_src) {
    if (!S.init) return fail(2, "asora_source_data_to_device: device not initialised");
    if (num_src < 0 || (num_src > 0 && (!pos || !flux))) return fail(9, "asora_source_data_to_device: bad arguments");
    // Positions are periodic like in the reference (-DPERIODIC, src/asora/meson.build:33: every access goes through
    // modulo(i, N), raytracing.cu:32-40), e.g. the 1-based lists of pyc2ray's read_sources / generate_test_sourcefile
    // (1 .. N) run as they are.  Only values that cannot be grid positions at all are refused.
    for (int i = 0; i < 3 * num_src; i++)
        if (pos[i] < -MAX_SOURCE_COORD || pos[i] > MAX_SOURCE_COORD)
            return fail(10, "asora_source_data_to_device: source %d coordinate %d = %d is not a grid position", i / 3,
                        i % 3, pos[i]);
    ASORA_ON_DEVICE("asora_source_data_to_device");
    // the reference re-allocates on every call and never frees (memory.cu:121,127); we do free
    if (S.src_pos) (void)hipFree(S.src_pos);
    if (S.src_flux) (void)hipFree(S.src_flux);
    S.src_pos = nullptr;
    S.src_flux = nullptr;
    S.num_src = 0;
    const size_t n = (size_t)std::max(num_src, 1);
    HIP_TRY(hipMalloc(&S.src_pos, 3 * n * sizeof(int)));
    HIP_TRY(hipMalloc(&S.src_flux, n * sizeof(double)));
    S.host_pos.assign(pos, pos + 3 * (size_t)num_src);
    S.host_flux.assign(flux, flux + (size_t)num_src);
    S.staged_n = -1;
    S.num_src = num_src;
    return stage_sources(num_src);  // the usual case: the whole list is raytraced
}

int asora_do_all_sources(double R, double *coldensh_out, double sig, double dr, const double *ndens,
                         const double *xh_av, double *phi_ion, int num_src, int m1, double minlogtau,
                         double dlogtau, int num_tau) {
    (void)ndens;  // ignored, like the reference (src/asora/raytracing.cu:127-129)
    return do_all_sources_host(R, coldensh_out, sig, dr, xh_av, phi_ion, nullptr, num_src, m1, minlogtau, dlogtau,
                               num_tau);
}

int asora_do_all_sources_heat(double R, double *coldensh_out, double sig, double dr, const double *ndens,
                              const double *xh_av, double *phi_ion, double *phi_heat, int num_src, int m1,
                              double minlogtau, double dlogtau, int num_tau) {
    (void)ndens;
    if (!phi_heat) return fail(9, "asora_do_all_sources_heat: null phi_heat pointer");
    return do_all_sources_host(R, coldensh_out, sig, dr, xh_av, phi_ion, phi_heat, num_src, m1, minlogtau, dlogtau,
                               num_tau);
}

int asora_do_all_sources_dev(double R, double *coldensh_out_dev, double sig, double dr, const double *xh_av_dev,
                             double *phi_ion_dev, double *phi_heat_dev, int num_src, int m1, double minlogtau,
                             double dlogtau, int num_tau, void *stream) {
    ASORA_ON_DEVICE("asora_do_all_sources_dev");
    return do_all_sources_impl(R, coldensh_out_dev, sig, dr, xh_av_dev, phi_ion_dev, phi_heat_dev, num_src, m1,
                               minlogtau, dlogtau, num_tau, (hipStream_t)stream);
}

// ---- chemistry pass (asora_chem.hpp) ---------------------------------------------------------
// One chemistry pass; `stats` is the device record the kernel reduces into (zeroed here), `landing` a host buffer
// the three results are copied to when the caller wants them (the call then waits for the stream).
static int global_pass_launch(double dt, const double *ndens_dev, const double *temp_dev, const double *xh_dev,
                              double *xh_av_dev, double *xh_intermed_dev, const double *phi_ion_dev,
                              const double *clump_dev, double bh00, double albpow, double colh0, double temph0,
                              double abu_c, int m1, long long *conv_flag, double *sum_xh1, double *sum_xh0,
                              hipStream_t st, asora::ChemStats *stats, asora::ChemStats *landing) {
    HIP_TRY(hipMemsetAsync(stats, 0, sizeof(asora::ChemStats), st));
    asora::ChemArgs A;
    A.ndens = ndens_dev; A.temp = temp_dev; A.xh = xh_dev; A.phi_ion = phi_ion_dev; A.clump = clump_dev;
    A.xh_av = xh_av_dev; A.xh_intermed = xh_intermed_dev; A.stats = stats;
    A.ncell = (long long)m1 * m1 * m1;
    A.dt = dt; A.bh00 = bh00; A.albpow = albpow; A.colh0 = colh0; A.temph0 = temph0; A.abu_c = abu_c;
    const long long want = (A.ncell + asora::CHEM_THREADS - 1) / asora::CHEM_THREADS;
    const int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)S.num_cu * 16));
    hipLaunchKernelGGL(asora::chemistry_kernel, dim3(grid), dim3(asora::CHEM_THREADS), 0, st, A);
    HIP_TRY(hipGetLastError());
    if (conv_flag || sum_xh1 || sum_xh0) {
        HIP_TRY(hipMemcpyAsync(landing, stats, sizeof(asora::ChemStats), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (conv_flag) *conv_flag = (long long)landing->conv_flag;
        if (sum_xh1) *sum_xh1 = landing->sum_xh1;
        if (sum_xh0) *sum_xh0 = landing->sum_xh0;
    }
    return 0;
}

int asora_global_pass_dev(double dt, const double *ndens_dev, const double *temp_dev, const double *xh_dev,
                          double *xh_av_dev, double *xh_intermed_dev, const double *phi_ion_dev,
                          const double *clump_dev, double bh00, double albpow, double colh0, double temph0,
                          double abu_c, int m1, long long *conv_flag, double *sum_xh1, double *sum_xh0,
                          void *stream) {
    if (!S.init) return fail(2, "asora_global_pass_dev: device not initialised (call asora_device_init)");
    if (m1 < 1) return fail(9, "asora_global_pass_dev: bad mesh size %d", m1);
    if (!ndens_dev || !temp_dev || !xh_dev || !xh_av_dev || !xh_intermed_dev || !phi_ion_dev || !clump_dev)
        return fail(9, "asora_global_pass_dev: null grid pointer");
    ASORA_ON_DEVICE("asora_global_pass_dev");
    if (!S.chem_stats) HIP_TRY(hipMalloc(&S.chem_stats, CHEM_RECORDS * CHEM_RECORD_STRIDE));
    asora::ChemStats h;
    return global_pass_launch(dt, ndens_dev, temp_dev, xh_dev, xh_av_dev, xh_intermed_dev, phi_ion_dev, clump_dev, bh00,
                              albpow, colh0, temph0, abu_c, m1, conv_flag, sum_xh1, sum_xh0, (hipStream_t)stream,
                              chem_record(S.chem_stats), &h);
}

int asora_global_pass_dev_noalloc(double dt, const double *ndens_dev, const double *temp_dev, const double *xh_dev,
                                  double *xh_av_dev, double *xh_intermed_dev, const double *phi_ion_dev,
                                  const double *clump_dev, double bh00, double albpow, double colh0, double temph0,
                                  double abu_c, int m1, long long *conv_flag, double *sum_xh1, double *sum_xh0,
                                  void *stream) {
    if (!S.init) return fail(2, "asora_global_pass_dev_noalloc: device not initialised (call asora_device_init)");
    if (m1 < 1) return fail(9, "asora_global_pass_dev_noalloc: bad mesh size %d", m1);
    if (m1 > ASORA_NOALLOC_MAX_M1)
        return fail(14, "asora_global_pass_dev_noalloc: mesh size %d exceeds the maximum of %d this entry point accepts",
                    m1, ASORA_NOALLOC_MAX_M1);
    if (!ndens_dev || !temp_dev || !xh_dev || !xh_av_dev || !xh_intermed_dev || !phi_ion_dev || !clump_dev)
        return fail(9, "asora_global_pass_dev_noalloc: null grid pointer");
    ASORA_ON_DEVICE("asora_global_pass_dev_noalloc");
    // this device's copy of the scratch array, by the address taken at init -- never by the array's name
    void *ring = noalloc_scratch_ptr[S.device];
    if (!ring || !S.chem_stats_pinned) return fail(2, "asora_global_pass_dev_noalloc: scratch not set up on device %d", S.device);
    return global_pass_launch(dt, ndens_dev, temp_dev, xh_dev, xh_av_dev, xh_intermed_dev, phi_ion_dev, clump_dev, bh00,
                              albpow, colh0, temph0, abu_c, m1, conv_flag, sum_xh1, sum_xh0, (hipStream_t)stream,
                              chem_record(ring), S.chem_stats_pinned);
}

int asora_global_pass(double dt, const double *ndens, const double *temp, const double *xh, double *xh_av,
                      double *xh_intermed, const double *phi_ion, const double *clump, double bh00, double albpow,
                      double colh0, double temph0, double abu_c, int m1, int *conv_flag) {
    if (!S.init) return fail(2, "asora_global_pass: device not initialised (call asora_device_init)");
    if (m1 != S.N) return fail(3, "asora_global_pass: m1=%d differs from device_init N=%d", m1, S.N);
    if (!ndens || !temp || !xh || !xh_av || !xh_intermed || !phi_ion || !clump || !conv_flag)
        return fail(9, "asora_global_pass: null pointer");
    ASORA_ON_DEVICE("asora_global_pass");
    const size_t bytes = S.cells * sizeof(double);
    for (double *&b : S.chem_buf)
        if (int rc = alloc_grid(b)) return rc;
    const double *src[7] = {ndens, temp, xh, xh_av, xh_intermed, phi_ion, clump};
    HIP_TRY(hipDeviceSynchronize());  // (the staged copies run on streams of their own)
    for (int i = 0; i < 7; i++)
        if (int rc = staged_copy(S.chem_buf[i], const_cast<double *>(src[i]), bytes, true)) return rc;
    long long cf = 0;
    if (int rc = asora_global_pass_dev(dt, S.chem_buf[0], S.chem_buf[1], S.chem_buf[2], S.chem_buf[3], S.chem_buf[4],
                                       S.chem_buf[5], S.chem_buf[6], bh00, albpow, colh0, temph0, abu_c, m1, &cf,
                                       nullptr, nullptr, nullptr))
        return rc;
    if (int rc = staged_copy(S.chem_buf[3], xh_av, bytes, false)) return rc;  // (asora_global_pass_dev has waited for the stream)
    if (int rc = staged_copy(S.chem_buf[4], xh_intermed, bytes, false)) return rc;
    *conv_flag = (int)cf;
    return 0;
}

int asora_test_cinterp_ex(const int32_t pos0[3], const double *dens, int N, double sigma, int lo, int cnt,
                          double *out, int fortran_constants);

int asora_test_cinterp(const int32_t pos0[3], const double *dens, int N, double *out, int fortran_constants) {
    return asora_test_cinterp_ex(pos0, dens, N, 1.0, 0, 4, out, fortran_constants);
}

int asora_test_cinterp_ex(const int32_t pos0[3], const double *dens, int N, double sigma, int lo, int cnt,
                          double *out, int fortran_constants) {
    if (!pos0 || !dens || !out || N < 2 || cnt < 1 || cnt > 8) return fail(9, "asora_test_cinterp: bad arguments");
    const size_t nb = (size_t)N * N * N * sizeof(double), ob = (size_t)cnt * cnt * cnt * 2 * sizeof(double);
    double *d_dens = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_dens, nb));
    hipError_t e = hipMalloc(&d_out, ob);
    if (e != hipSuccess) {
        (void)hipFree(d_dens);
        return fail(100 + (int)e, "asora_test_cinterp: hipMalloc failed");  // reference: MemoryError
    }
    int rc = 0;
    do {
        if ((e = hipMemcpy(d_dens, dens, nb, hipMemcpyHostToDevice)) != hipSuccess) break;
        hipLaunchKernelGGL(asora::cinterp_test_kernel, dim3(1), dim3(cnt, cnt, cnt), 0, nullptr, d_out, d_dens,
                           pos0[0], pos0[1], pos0[2], N, sigma, lo, fortran_constants);
        if ((e = hipGetLastError()) != hipSuccess) break;
        if ((e = hipMemcpy(out, d_out, ob, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    if (e != hipSuccess) rc = fail(100 + (int)e, "asora_test_cinterp: %s", hipGetErrorString(e));
    (void)hipFree(d_dens);
    (void)hipFree(d_out);
    return rc;
}

int asora_visit_count(double R, int m1, long long *visits, long long *in_sphere) {
    if (m1 < 2 || !visits || !in_sphere) return fail(9, "asora_visit_count: bad arguments");
    const int q_max = compute_q_max(R, m1);
    const int last_r = m1 / 2 - 1 + m1 % 2, last_l = -(m1 / 2);
    long long v = 0, s = 0;
    // rows of constant (i, j): k ranges over |k| <= q_max - |i| - |j| clipped to the box
    for (int i = last_l; i <= last_r; i++)
        for (int j = last_l; j <= last_r; j++) {
            const int rem = q_max - std::abs(i) - std::abs(j);
            if (rem < 0) continue;
            const int klo = std::max(last_l, -rem), khi = std::min(last_r, rem);
            v += khi - klo + 1;
            const double r2 = R * R - (double)i * i - (double)j * j;
            if (r2 < 0) continue;
            const int kr = (int)std::floor(std::sqrt(r2));
            int a = std::max(klo, -kr), b = std::min(khi, kr);
            // fix rounding of sqrt on the boundary
            while (a - 1 >= klo && (double)(a - 1) * (a - 1) <= r2) a--;
            while (b + 1 <= khi && (double)(b + 1) * (b + 1) <= r2) b++;
            while (a <= b && (double)a * a > r2) a++;
            while (b >= a && (double)b * b > r2) b--;
            if (b >= a) s += b - a + 1;
        }
    *visits = v;
    *in_sphere = s;
    return 0;
}

int asora_geometry_census(double R, double dr, int m1, long long *owned, long long *owned_rate, long long *computed,
                          int *staging_entries, int *variants) {
    if (m1 < 2 || !(dr > 0.0) || !(R >= 0.0) || !owned || !owned_rate || !computed || !staging_entries || !variants)
        return fail(9, "asora_geometry_census: bad arguments");
    const int q_max = compute_q_max(R, m1);
    const int last_r = m1 / 2 - 1 + m1 % 2, last_l = m1 / 2;
    HostGeometry G;
    make_geometry(m1, R, dr, q_max, last_l, last_r, G);
    long long own = 1, rate = 1, comp = 24;  // the source cell: owned (and rated) once, staged by every sector
    for (int t = 0; t < asora::SECTORS; t++) {
        const int axis = t >> 3, sp = (t & 4) ? -1 : 1, su = (t & 2) ? -1 : 1, sv = (t & 1) ? -1 : 1;
        const uint32_t ghostmask = (su < 0 ? asora::FLAG_A0 : 0u) | (sv < 0 ? asora::FLAG_B0 : 0u) |
                                   (axis >= 1 ? asora::FLAG_BN : 0u) | (axis == 2 ? asora::FLAG_AN : 0u);
        const int nmax = std::min(sp > 0 ? last_r : last_l, q_max);
        const int variant = G.vmap[(((su < 0) ? 2 : 0) + ((sv < 0) ? 1 : 0)) * 2 + (axis != 0 ? 1 : 0)];
        const asora::PlaneInfo *pl = G.planes.data() + (size_t)variant * G.nplanes;
        for (int n = 1; n <= nmax; n++)
            for (uint32_t i = 0; i < pl[n].cntT; i++) {
                const uint32_t fl = G.meta[pl[n].off + i].x >> asora::META_FLAG_SHIFT;
                comp++;
                if (fl & ghostmask) continue;
                own++;
                if (i < pl[n].cntA) rate++;
            }
    }
    *owned = own; *owned_rate = rate; *computed = comp;
    *staging_entries = G.entries_max; *variants = G.nvariants;
    return 0;
}

int asora_last_sweep_ms(float *ms) {
    if (!ms) return fail(9, "asora_last_sweep_ms: null pointer");
    if (!S.init || !S.timed) return fail(11, "asora_last_sweep_ms: no sweep has been timed");
    ASORA_ON_DEVICE("asora_last_sweep_ms");
    HIP_TRY(hipEventSynchronize(S.ev1));
    HIP_TRY(hipEventElapsedTime(ms, S.ev0, S.ev1));
    return 0;
}

int asora_set_option(const char *name, long long value) {
    if (!name) return fail(9, "asora_set_option: null name");
    if (!std::strcmp(name, "force_generic")) S.opt.force_generic = value != 0;
    else if (!std::strcmp(name, "max_table_cells")) S.opt.max_table_cells = value;
    else if (!std::strcmp(name, "lds_pad")) S.opt.lds_pad = (int)value;
    else if (!std::strcmp(name, "lds_stage_limit")) S.opt.lds_stage_limit = (int)value;
    else if (!std::strcmp(name, "wg_threads")) S.opt.wg_threads = (int)value;
    else if (!std::strcmp(name, "source_order")) S.opt.source_order = (int)value;
    else if (!std::strcmp(name, "quad_kernel")) S.opt.quad_kernel = (int)value;
    else if (!std::strcmp(name, "quad_max_entries")) S.opt.quad_max_entries = (int)value;
    else if (!std::strcmp(name, "oct_max_entries")) S.opt.oct_max_entries = (int)value;
    else if (!std::strcmp(name, "stride_residue")) S.opt.stride_residue = (int)value;
    else if (!std::strcmp(name, "compact_staging")) S.opt.compact_staging = (int)value;
    else if (!std::strcmp(name, "src_pack")) S.opt.src_pack = (int)value;
    else if (!std::strcmp(name, "pack_max_lds")) S.opt.pack_max_lds = (int)value;
    else if (!std::strcmp(name, "pack_max_entries")) S.opt.pack_max_entries = (int)value;
    else if (!std::strcmp(name, "skip_unobserved")) S.opt.skip_unobserved = value != 0;
    else if (!std::strcmp(name, "skip_saturated")) S.opt.skip_saturated = value != 0;
    else if (!std::strcmp(name, "single_layout")) S.opt.single_layout = (int)value;
    else if (!std::strcmp(name, "single_max_visits")) S.opt.single_max_visits = value;
    else if (!std::strcmp(name, "copy_threads")) g_copy_threads = (int)std::max<long long>(0, std::min<long long>(value, Stager::MAXT));
    else return fail(12, "asora_set_option: unknown option '%s'", name);
    return 0;
}

int asora_get_option(const char *name, long long *value) {
    if (!name || !value) return fail(9, "asora_get_option: null pointer");
    if (!std::strcmp(name, "force_generic")) *value = S.opt.force_generic;
    else if (!std::strcmp(name, "max_table_cells")) *value = S.opt.max_table_cells;
    else if (!std::strcmp(name, "lds_pad")) *value = S.opt.lds_pad;
    else if (!std::strcmp(name, "lds_stage_limit")) *value = S.opt.lds_stage_limit;
    else if (!std::strcmp(name, "wg_threads")) *value = S.opt.wg_threads;
    else if (!std::strcmp(name, "source_order")) *value = S.opt.source_order;
    else if (!std::strcmp(name, "quad_kernel")) *value = S.opt.quad_kernel;
    else if (!std::strcmp(name, "quad_max_entries")) *value = S.opt.quad_max_entries;
    else if (!std::strcmp(name, "oct_max_entries")) *value = S.opt.oct_max_entries;
    else if (!std::strcmp(name, "stride_residue")) *value = S.opt.stride_residue;
    else if (!std::strcmp(name, "compact_staging")) *value = S.opt.compact_staging;
    else if (!std::strcmp(name, "src_pack")) *value = S.opt.src_pack;
    else if (!std::strcmp(name, "pack_max_lds")) *value = S.opt.pack_max_lds;
    else if (!std::strcmp(name, "pack_max_entries")) *value = S.opt.pack_max_entries;
    else if (!std::strcmp(name, "skip_unobserved")) *value = S.opt.skip_unobserved;
    else if (!std::strcmp(name, "skip_saturated")) *value = S.opt.skip_saturated;
    else if (!std::strcmp(name, "single_layout")) *value = S.opt.single_layout;
    else if (!std::strcmp(name, "single_max_visits")) *value = S.opt.single_max_visits;
    else if (!std::strcmp(name, "copy_threads")) *value = g_copy_threads;
    else return fail(12, "asora_get_option: unknown option '%s'", name);
    return 0;
}

int asora_last_sweep_is_table_driven(void) { return S.rec.last_tab; }

const char *asora_last_launch_notice(void) { return S.rec.notice.c_str(); }

int asora_test_division(const double *num, const double *den, long long n, long long *mismatch_div,
                        long long *mismatch_rcp, long long *mismatch_const) {
    if (!num || !den || n < 1 || !mismatch_div || !mismatch_rcp || !mismatch_const)
        return fail(9, "asora_test_division: bad arguments");
    struct Scoped {  // device buffers of this call: freed on every return path
        void *p = nullptr;
        ~Scoped() { if (p) (void)hipFree(p); }
    } b_n, b_d, b_c;
    HIP_TRY(hipMalloc(&b_n.p, n * sizeof(double)));
    HIP_TRY(hipMalloc(&b_d.p, n * sizeof(double)));
    HIP_TRY(hipMalloc(&b_c.p, 3 * sizeof(unsigned long long)));
    double *d_n = static_cast<double *>(b_n.p), *d_d = static_cast<double *>(b_d.p);
    unsigned long long *d_c = static_cast<unsigned long long *>(b_c.p);
    HIP_TRY(hipMemcpy(d_n, num, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, den, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_c, 0, 3 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(asora::division_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_n, d_d,
                       n, d_c);
    HIP_TRY(hipGetLastError());
    unsigned long long c[3];
    HIP_TRY(hipMemcpy(c, d_c, sizeof c, hipMemcpyDeviceToHost));
    *mismatch_div = (long long)c[0]; *mismatch_rcp = (long long)c[1]; *mismatch_const = (long long)c[2];
    return 0;
}

int asora_last_launch_info(int *grid, int *block, int *lds_bytes, int *staging_is_lds) {
    if (grid) *grid = S.rec.last_grid;
    if (block) *block = S.rec.last_block;
    if (lds_bytes) *lds_bytes = S.rec.last_lds;
    if (staging_is_lds) *staging_is_lds = S.rec.last_is_lds;
    return 0;
}

}  // extern "C"
