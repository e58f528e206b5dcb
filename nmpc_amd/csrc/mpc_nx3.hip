// Condensing + QP kernels for nx = 3.
#include "mpc_kernels.h"

namespace nmpc {

NMPC_DEFINE_MPC_DISPATCH(3)

}  // namespace nmpc
