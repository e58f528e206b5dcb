"""nmpc_amd: MI355X-native LPV-NMPC inner loop behind the reference's Python call signatures."""
from ._lib import NmpcError, LIB_PATH  # noqa: F401
from .model import Encoder, Norm, load_system, DISC_WEIGHTS  # noqa: F401
from .lpv import (Jacobian, jacobians, quad_nodes, embed, lpv_int, lpv_int_C, lambda_simpson,  # noqa: F401
                  lambda_trap, par_lpv_int, MappedGet, mapped_get, getABlist, getABClist, getClist)
from .mpcutil import getPhi, getGamma, getF0, getZ, getPsi, getOmega, getDEMc, getDEMc_out  # noqa: F401
from .qp import solve_qp  # noqa: F401
from .nmpc import (OutputUnbalancedDisc, NoisyUnbalancedDisc, ModelPlant, Controller, mpc_qp, output_NMPC_linear,  # noqa: F401
                   par_NMPC_linear, NMPC_linear)
from .target import GetABC, getXsUs_Cs, getXsUs  # noqa: F401
from .lpvrk4 import lpv_rk4  # noqa: F401
from .reference import stackReferences, randomLevelReference, random_level_batch, lpv_histogram, comp_time_stats  # noqa: F401
