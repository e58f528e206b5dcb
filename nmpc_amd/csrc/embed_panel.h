// Split-fp16 embedding kernel for wide two-hidden-layer nets (hidden width 128 or 256): embed_panel.hip.
#pragma once
#include "embed.h"

namespace nmpc {

struct PanelArgs {
    PanelNet net;
    const float* pa;   // [rows, na]
    const float* pb;   // [rows, nb] (may be null when nb = 0)
    float* oa;         // [rows, nout, na]
    float* ob;         // [rows, nout, nb]
    int na, nb, nout;
    int npts;
    const int* count_dev;
    const int* act;
    int group;
};

int panel_launch_net(const PanelNet& net, PanelArgs a, const NodeTable& nt, hipStream_t s, bool loop);

}  // namespace nmpc
