// Host side of the long-horizon condensing + QP kernels (mpc_wide.h): when they apply, and the launches
// for nx = 2 (nx = 3, 4: mpc_wide34.hip).
#include <cstdlib>

#include "mpc_wide_fast.h"

namespace nmpc {

// Output feedback with the slack variable (the reference's long-horizon configuration), horizons from
// NMPC_MPC_WIDE_MIN (default 24) up: below that a controller fits one wavefront's registers / a few KB of
// LDS and sixteen of them share a CU.  Horizons whose workgroup would not fit the CU's LDS stay on the
// one-wavefront kernels.
bool mpc_wide_applies(int Nc, int nx, int mode, int slack) {
    static std::atomic<int> cache{-1};
    int lo = cache.load(std::memory_order_relaxed);
    if (lo < 0) {
        const char* e = getenv("NMPC_MPC_WIDE_MIN");
        lo = e ? atoi(e) : 24;
        if (lo < 2) lo = 2;
        cache.store(lo, std::memory_order_relaxed);
    }
    if (!(mode == 0 && slack == 1 && nx >= 2 && nx <= 4 && Nc >= lo && Nc <= 63)) return false;
    return sizeof(double) * (size_t)WideLds::doubles(Nc, nx, wide_kh(Nc, nx)) <= 160 * 1024;
}

int launch_mpc_qp_wide(const MpcQpArgs& a, hipStream_t s) {
    if (a.nx != 2) return launch_mpc_qp_wide_nx34(a, s);
    const int kh = wide_kh(a.c.Nc, 2);
    if (kh == 16) return launch_qp_wide_t<2, 16>(a, s);
    if (kh == 25) return launch_qp_wide_t<2, 25>(a, s);
    return launch_qp_wide_t<2, 32>(a, s);
}

// Two stages when the carried working sets are tried first (NMPC_QP_WARM = 2, the default): the small-LDS kernel that
// decides the common case, then the full kernel over what it left (mpc_wide_fast.h).
int launch_mpc_solve_wide(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s) {
    if (d.nx != 2) return launch_mpc_solve_wide_nx34(d, rp, it, k, s);
    const int kh = wide_kh(d.Nc, 2);
    const bool two = d.warm >= 2 && d.slow;
    if (kh == 16) return (two && launch_solve_fast_t<2, 16>(d, rp, it, k, s)) ? 1 : launch_solve_wide_t<2, 16>(d, rp, it, k, s);
    if (kh == 25) return (two && launch_solve_fast_t<2, 25>(d, rp, it, k, s)) ? 1 : launch_solve_wide_t<2, 25>(d, rp, it, k, s);
    return (two && launch_solve_fast_t<2, 32>(d, rp, it, k, s)) ? 1 : launch_solve_wide_t<2, 32>(d, rp, it, k, s);
}

}  // namespace nmpc
