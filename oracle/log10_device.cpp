// log10_device.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Host build of the HIP kernels' table-driven log10 (pyc2ray_amd/csrc/asora_log10.hpp: the same source the device
// compiles, the same table builder the library runs on the host before uploading the table), exported for the
// oracle variant `libc2ray_oracle_devlog10.so` (oracle/Makefile `port_devlog10`).  That variant is c2ray_oracle.c
// with ONE change: photo_lookuptable's `log10` (photorates.f90:139) is this function instead of glibc's.  It exists
// to demonstrate what tests/parity.py's conditioning term stands for: with the oracle's log10 made equal to the
// device's, the GPU rates must meet the plain 1e-10 bar with no cell admitted through that term
// (tests/test_gpu_parity.py::test_hip_vs_oracle_with_device_log10).
// __builtin_fma without -mfma is glibc's correctly rounded fma(): the same result as the device's v_fma_f64.
#include "../pyc2ray_amd/csrc/asora_log10.hpp"

static double g_table[asora::LOG10_TABLE_DOUBLES];
static bool g_built = false;

extern "C" double c2ray_log10_device(double x) {
    if (!g_built) {
        asora::build_log10_table(g_table);
        g_built = true;
    }
    return asora::log10_tab(x, (const double *)g_table);
}
