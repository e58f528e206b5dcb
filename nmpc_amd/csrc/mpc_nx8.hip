// Condensing + QP kernels for nx = 8 (deepSI encoders with more states than the disc model: LTI.ipynb cell 5,
// sinCos.ipynb cell 6; the synthetic nx = 8 configuration).
#include "mpc_kernels.h"

namespace nmpc {

NMPC_DEFINE_MPC_DISPATCH64(8)

}  // namespace nmpc
