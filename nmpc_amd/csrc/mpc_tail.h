// Per-controller tail of the closed-loop step: the LPV passes from `it0` on in one launch (mpc_tail.hip).
#pragma once
#include "embed.h"
#include "mpc_kernels.h"

namespace nmpc {

// true when the tail kernel is instantiated for this controller set-up (the reference's output-feedback loop,
// NMPC_test.py:326-344, with two or three states, nu = ny = 1, horizons up to 15, slack on every row; the canonical
// Nc = 10 at nx = 2 with the horizon as a compile-time constant) and the model's split-fp16 packs are in use
bool mpc_tail_applies(const CtrlDev& d, const ModelDev& m, const MpcGeom& g);
int launch_mpc_tail(const CtrlDev& d, const RunPtrs& rp, const ModelDev& m, const NodeTable& nt, int it0, int k, hipStream_t s);

}  // namespace nmpc
