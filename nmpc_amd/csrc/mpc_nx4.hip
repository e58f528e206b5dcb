// Condensing + QP kernels for nx = 4.
#include "mpc_kernels.h"

namespace nmpc {

NMPC_DEFINE_MPC_DISPATCH(4)

}  // namespace nmpc
