// Condensing + QP kernels for nx = 2, including the instantiation with the horizon of the reference's
// canonical configuration (Nc = 10: NMPC_test.py:326-344) as a compile-time constant.
#include "mpc_kernels.h"

namespace nmpc {

int dispatch_mpc_qp_nx2(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s) {
    if (a.c.Nc == 10 && a.c.mode == 0 && a.c.slack && g.GS == 64 && g.mpl <= 1) return launch_mpc_qp<64, 1, 2, 10>(a, g, s);
    NMPC_GEOM_SWITCH(launch_mpc_qp, 2, g, a, g, s)
    set_error("nmpc_mpc_qp: no kernel for nx=%d GS=%d rows/lane=%d", a.nx, g.GS, g.mpl);
    return 1;
}

int dispatch_mpc_solve_nx2(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s) {
    if (d.Nc == 10 && d.c.mode == 0 && d.c.slack && g.GS == 64 && g.mpl <= 1) return launch_mpc_solve<64, 1, 2, 10>(d, rp, it, k, g, s);
    NMPC_GEOM_SWITCH(launch_mpc_solve, 2, g, d, rp, it, k, g, s)
    set_error("nmpc_ctrl_run: no kernel for nx=%d GS=%d rows/lane=%d", d.nx, g.GS, g.mpl);
    return 1;
}

}  // namespace nmpc
