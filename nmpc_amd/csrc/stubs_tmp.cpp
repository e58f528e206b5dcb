#include "../../include/nmpc_amd.h"
#include "common.h"
extern "C" int nmpc_lpv_rk4(int, int, int, const double*, const double*, const double*, const double*, double, const double*, double*, void*) { nmpc::set_error("todo"); return 1; }
extern "C" int nmpc_condense(int, int, int, int, const double*, const double*, const double*, double*, double*, double*, void*) { nmpc::set_error("todo"); return 1; }
extern "C" int nmpc_qp_solve(int, int, int, int, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*, int*, int*, void*) { nmpc::set_error("todo"); return 1; }
