// Host side of the long-horizon condensing + QP kernels (mpc_wide.h): when they apply, and their launches.
#include <cstdlib>

#include "mpc_wide.h"

namespace nmpc {

// Output feedback with the slack variable (the reference's long-horizon configuration), horizons from
// NMPC_MPC_WIDE_MIN (default 24) up: below that a controller fits one wavefront's registers / a few KB of
// LDS and sixteen of them share a CU.
bool mpc_wide_applies(int Nc, int nx, int mode, int slack) {
    static std::atomic<int> cache{-1};
    int lo = cache.load(std::memory_order_relaxed);
    if (lo < 0) {
        const char* e = getenv("NMPC_MPC_WIDE_MIN");
        lo = e ? atoi(e) : 24;
        if (lo < 2) lo = 2;
        cache.store(lo, std::memory_order_relaxed);
    }
    return mode == 0 && slack == 1 && nx >= 2 && nx <= 4 && Nc >= lo && Nc <= 63 &&
           sizeof(double) * (size_t)WideLds::doubles(Nc, nx) <= 160 * 1024;
}

template <int NX>
static int launch_qp_t(const MpcQpArgs& a, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_qp_wide_kernel<NX>)) return 1;
        mark_used_on_device(once);
    }
    const size_t lds = sizeof(double) * (size_t)WideLds::doubles(a.c.Nc, NX);
    hipLaunchKernelGGL((mpc_qp_wide_kernel<NX>), dim3(a.nb), dim3(WIDE_THREADS), lds, s, a);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int NX>
static int launch_solve_t(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_solve_wide_kernel<NX>)) return 1;
        mark_used_on_device(once);
    }
    const size_t lds = sizeof(double) * (size_t)WideLds::doubles(d.Nc, NX);
    hipLaunchKernelGGL((mpc_solve_wide_kernel<NX>), dim3(d.B), dim3(WIDE_THREADS), lds, s, d, rp, it, k);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_mpc_qp_wide(const MpcQpArgs& a, hipStream_t s) {
    if (a.nx == 2) return launch_qp_t<2>(a, s);
    if (a.nx == 3) return launch_qp_t<3>(a, s);
    if (a.nx == 4) return launch_qp_t<4>(a, s);
    set_error("nmpc_mpc_qp: no long-horizon kernel for nx=%d", a.nx);
    return 1;
}

int launch_mpc_solve_wide(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s) {
    if (d.nx == 2) return launch_solve_t<2>(d, rp, it, k, s);
    if (d.nx == 3) return launch_solve_t<3>(d, rp, it, k, s);
    if (d.nx == 4) return launch_solve_t<4>(d, rp, it, k, s);
    set_error("nmpc_ctrl_run: no long-horizon kernel for nx=%d", d.nx);
    return 1;
}

}  // namespace nmpc
