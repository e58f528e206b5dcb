// Long-horizon condensing + QP kernels (mpc_wide.h) for nx = 3 and 4.
#include "mpc_wide_fast.h"

namespace nmpc {

int launch_mpc_qp_wide_nx34(const MpcQpArgs& a, hipStream_t s) {
    if (a.nx == 3) return launch_qp_wide_t<3, 32>(a, s);
    if (a.nx == 4) return launch_qp_wide_t<4, 32>(a, s);
    set_error("nmpc_mpc_qp: no long-horizon kernel for nx=%d", a.nx);
    return 1;
}

int launch_mpc_solve_wide_nx34(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s) {
    const bool two = d.warm >= 2 && d.slow;
    if (d.nx == 3) return (two && launch_solve_fast_t<3, 32>(d, rp, it, k, s)) ? 1 : launch_solve_wide_t<3, 32>(d, rp, it, k, s);
    if (d.nx == 4) return (two && launch_solve_fast_t<4, 32>(d, rp, it, k, s)) ? 1 : launch_solve_wide_t<4, 32>(d, rp, it, k, s);
    set_error("nmpc_ctrl_run: no long-horizon kernel for nx=%d", d.nx);
    return 1;
}

}  // namespace nmpc
