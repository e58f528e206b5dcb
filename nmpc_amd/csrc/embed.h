// Internal interface of the LPV embedding kernel (shared with the controller step).
#pragma once
#include "../../include/nmpc_amd.h"
#include "common.h"

namespace nmpc {

constexpr int EMBED_THREADS = 256;
constexpr int EMBED_TILE = 16;      // scheduling points per workgroup

// Quadrature nodes.  Callers fill n, lam, w; embed_launch moves nodes at lambda = 0 (where the
// Jacobian is a constant of the net) out of the table into w_zero and sets w_sum.
struct NodeTable {
    int n;
    float w_zero;   // total weight of the nodes at lambda = 0
    float w_sum;    // total weight of all nodes (multiplies the linear branch)
    float lam[NMPC_MAX_NODES];
    float w[NMPC_MAX_NODES];
    // set by embed_launch: recur = 1 if the nodes of every wave's quarter [n*w/4, n*(w+1)/4) are equally
    // spaced, dlam[w] = that spacing (the split-fp16 kernel then steps 2^-a from node to node by a
    // multiplication instead of evaluating it)
    int recur;
    float dlam[4];
    int prepared;   // set by embed_prepare_nodes
};

struct EmbedArgs {
    const float* fpack;
    const float* hpack;
    const float* x;    // [rows, nx]   scheduling states for A, B
    const float* u;    // [rows, nu]
    const float* xc;   // [rows, nx]   scheduling states for C
    float* A;          // [rows, nx, nx]
    float* B;          // [rows, nx, nu]
    float* C;          // [rows, ny, nx]
    int npts;              // number of points (upper bound when count_dev is set)
    const int* count_dev;  // optional device word: number of active groups
    const int* act;        // optional active-group list: point p -> row act[p/group]*group + p%group
    int group;
    int n_ftiles;
    // optional diagnostics (split-fp16 kernel): workgroup b stores the shader-clock cycles and the 100 MHz
    // real-time ticks it lived for to stamps[2 (b % stamp_slots)], [.. + 1]; null in every production launch
    unsigned long long* stamps;
    int stamp_slots;
};

int embed_launch(const ModelDev& m, EmbedArgs a, const NodeTable& nt, hipStream_t stream, bool loop = false);
// The node table a launch uses (lambda = 0 nodes folded, equal weights adjacent, stepping data) depends only
// on the model and the caller's nodes: callers that launch repeatedly prepare it once.
int embed_prepare_nodes(const ModelDev& m, const NodeTable& nodes, NodeTable& prepared);
int embed_launch_prepared(const ModelDev& m, EmbedArgs a, const NodeTable& prepared, hipStream_t stream, bool loop = false);
// true unless NMPC_EMBED selects another kernel than the split-fp16 one for the 64-wide nets
bool embed_split16_selected();
// split-fp16 kernels of the nets with five to eight states (embed_nx5.hip ..): ny = 1 or 2, mode 1 plain, 2 with the
// node recurrence; loop launches exist for ny = 1
int embed_dispatch_nx5(int ny, const EmbedArgs& a, const NodeTable& nt, int grid, size_t lds_bytes, int mode, bool loop, hipStream_t s);
int embed_dispatch_nx6(int ny, const EmbedArgs& a, const NodeTable& nt, int grid, size_t lds_bytes, int mode, bool loop, hipStream_t s);
int embed_dispatch_nx7(int ny, const EmbedArgs& a, const NodeTable& nt, int grid, size_t lds_bytes, int mode, bool loop, hipStream_t s);
int embed_dispatch_nx8(int ny, const EmbedArgs& a, const NodeTable& nt, int grid, size_t lds_bytes, int mode, bool loop, hipStream_t s);
// general-width kernel (embed_wide.hip); nt must already have its lambda = 0 nodes folded into w_zero
int embed_wide_launch(const ModelDev& m, const EmbedArgs& a, const NodeTable& nt, hipStream_t stream, bool loop);

}  // namespace nmpc
