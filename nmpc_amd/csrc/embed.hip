// LPV embedding kernel: path-integrated network Jacobians
//     A[p] = sum_j w_j dfdx(lam_j z_p),  B[p] = sum_j w_j dfdu(lam_j z_p),  C[p] = sum_j w_j dhdx(lam_j x_p)
// (reference: lpv_int.py:4-100, par_lpv_int.py:4-35, NMPC_test.py:129-201) for nets
//     net(z) = W_lin z + b_lin + W4 tanh(W2 tanh(W0 z + b0) + b2) + b4,   hidden width 64,
// whose Jacobian is  W_lin + W4 diag(d2) W2 diag(d1) W0,  d = 1 - tanh^2.
//
// Mapping to CDNA4.  One workgroup (4 waves) owns a tile of 16 scheduling points of one net; each
// wave owns a quarter of the quadrature nodes.  Lane l serves point (l & 15) and the quarter
// (l >> 4) of the hidden units (16 units per lane keeps the per-lane state near 130 VGPRs).  Per node:
//   1. layer 1 on the VALU: a1 = lam*(W0 z) + b0 is ONE fma per hidden unit because W0 z does not
//      depend on lam; tanh through v_exp_f32 / v_rcp_f32 (weights pre-scaled by 2 log2 e);
//   2. forward product  W2 t1  as 64 v_mfma_f32_16x16x4_f32 with the A fragments of W2 read from
//      LDS and t1 produced directly in B-fragment layout;
//   3. d2 from the accumulator in place - the accumulator's row layout is, by construction of
//      kappa(), the B-fragment layout of the next product, so nothing moves between lanes;
//   4. reverse products  (W4[c] .* W2)^T d2  (64 MFMAs per output row c);
//   5. q[c][k] += w * d1[k] * r1[c][k]  - the contraction with W0 is linear and is hoisted out of
//      the node loop, done once per tile.
// The four waves' partial sums are combined in LDS in a fixed order (bitwise reproducible).
#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"
#include "embed_tile.h"
#include "embed_kernel.h"

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <string>

namespace nmpc {


// NMPC_EMBED=f32 selects the fp32-MFMA kernel, NMPC_EMBED=wide the general-width kernel, anything else
// the split-fp16 kernel (when the model's weights suit it: see pack_net_h2)
static int embed_mode() {
    static std::atomic<int> cache{-1};
    int v = cache.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("NMPC_EMBED");
        v = (e && std::string(e) == "f32") ? 0 : ((e && std::string(e) == "wide") ? 2 : 1);
        cache.store(v, std::memory_order_relaxed);
    }
    return v;
}
static bool embed_use_h2() { return embed_mode() == 1; }
bool embed_split16_selected() { return embed_use_h2(); }

// equally spaced nodes within every wave's quarter: 2^-a is stepped by multiplication (NMPC_EMBED_RECUR=0 disables)
static bool recur_enabled() {
    static std::atomic<int> v{-1};
    int r = v.load(std::memory_order_relaxed);
    if (r < 0) { const char* e = getenv("NMPC_EMBED_RECUR"); r = (e && atoi(e) == 0) ? 0 : 1; v.store(r, std::memory_order_relaxed); }
    return r == 1;
}

int embed_prepare_nodes(const ModelDev& m, const NodeTable& nodes, NodeTable& nt) {
    NMPC_REQUIRE(nodes.n >= 1 && nodes.n <= NMPC_MAX_NODES, "embed: node count %d out of range", nodes.n);
    nt = NodeTable{};
    for (int i = 0; i < nodes.n; ++i) {
        nt.w_sum += nodes.w[i];
        if (nodes.lam[i] == 0.0f) { nt.w_zero += nodes.w[i]; continue; }
        if (nodes.w[i] == 0.0f) continue;          // contributes nothing
        nt.lam[nt.n] = nodes.lam[i]; nt.w[nt.n] = nodes.w[i]; ++nt.n;
    }
    {   // equal weights next to each other (the sum over nodes does not depend on their order): the
        // split-fp16 kernel applies a weight once per run of equal weights
        int idx[NMPC_MAX_NODES];
        for (int i = 0; i < nt.n; ++i) idx[i] = i;
        std::stable_sort(idx, idx + nt.n, [&](int a, int b) { return nt.w[a] > nt.w[b]; });
        NodeTable sorted = nt;
        for (int i = 0; i < nt.n; ++i) { sorted.lam[i] = nt.lam[idx[i]]; sorted.w[i] = nt.w[idx[i]]; }
        nt = sorted;
    }
    const bool h2 = m.narrow_ok && embed_use_h2() && m.fpack_h2 && m.hpack_h2;
    nt.recur = (h2 && recur_enabled() && m.step_ok) ? 1 : 0;
    for (int wv = 0; wv < 4 && nt.recur; ++wv) {
        const int n0 = (nt.n * wv) >> 2, n1 = (nt.n * (wv + 1)) >> 2;
        nt.dlam[wv] = n1 - n0 >= 2 ? (float)(((double)nt.lam[n1 - 1] - (double)nt.lam[n0]) / (n1 - n0 - 1)) : 0.0f;
        for (int i = n0 + 1; i < n1; ++i)      // lam[i] reproduced by lam[n0] + (i - n0) dlam to a few ulp
            if (std::fabs((double)nt.lam[n0] + (double)(i - n0) * (double)nt.dlam[wv] - (double)nt.lam[i]) > 5e-7) nt.recur = 0;
    }
    nt.prepared = 1;
    return 0;
}

int embed_launch(const ModelDev& m, EmbedArgs a, const NodeTable& nodes, hipStream_t stream, bool loop) {
    NodeTable nt;
    if (embed_prepare_nodes(m, nodes, nt)) return 1;
    return embed_launch_prepared(m, a, nt, stream, loop);
}

int embed_launch_prepared(const ModelDev& m, EmbedArgs a, const NodeTable& nt, hipStream_t stream, bool loop) {
    NMPC_REQUIRE(nt.prepared == 1, "embed: node table not prepared");
    NMPC_REQUIRE(a.npts >= 0, "embed: negative point count");
    if (a.npts == 0) return 0;
    const bool do_f = a.x && a.u && a.A && a.B, do_h = a.xc && a.C;
    NMPC_REQUIRE(do_f || do_h, "embed: nothing to compute (need x,u,A,B and/or xc,C)");
    if (a.group < 1) a.group = 1;
    const bool h2 = embed_use_h2() && m.fpack_h2 && m.hpack_h2;
    if (!m.narrow_ok || embed_mode() == 2 || (m.nx >= 4 && !h2)) return embed_wide_launch(m, a, nt, stream, loop);
    const int tiles = (a.npts + EMBED_TILE - 1) / EMBED_TILE;
    a.fpack = h2 ? m.fpack_h2 : m.fpack; a.hpack = h2 ? m.hpack_h2 : m.hpack;
    // nx = 4: two tiles of two output rows per 16 points; nx = 5..8: ceil(nx / 3) tiles of up to three rows
    a.n_ftiles = do_f ? (m.nx >= 5 ? (m.nx + 2) / 3 * tiles : (m.nx == 4 ? 2 * tiles : tiles)) : 0;
    const int grid = a.n_ftiles + (do_h ? tiles : 0);
    const int nout_tile = m.nx >= 5 ? 3 : (m.nx == 4 ? 2 : (m.nx > m.ny ? m.nx : m.ny));
    const int nin = m.nx + m.nu;
    const size_t lds_bytes = sizeof(float) * (4096 + 4096 * nout_tile + (nin > 8 ? PK_TAIL_EXT2 : (nin > 4 ? PK_TAIL_EXT : PK_TAIL)));
    const int mode = !h2 ? 0 : (nt.recur ? 2 : 1);
#define NMPC_EMBED_CASE(NX_, NU_, NY_)                                                                   \
    if (m.nx == NX_ && m.nu == NU_ && m.ny == NY_) {                                                     \
        if (mode == 2) return loop ? embed_launch_t<NX_, NU_, NY_, true, 2>(a, nt, grid, lds_bytes, stream)  \
                                   : embed_launch_t<NX_, NU_, NY_, false, 2>(a, nt, grid, lds_bytes, stream); \
        if (mode == 1) return loop ? embed_launch_t<NX_, NU_, NY_, true, 1>(a, nt, grid, lds_bytes, stream)  \
                                   : embed_launch_t<NX_, NU_, NY_, false, 1>(a, nt, grid, lds_bytes, stream); \
        return loop ? embed_launch_t<NX_, NU_, NY_, true, 0>(a, nt, grid, lds_bytes, stream)             \
                    : embed_launch_t<NX_, NU_, NY_, false, 0>(a, nt, grid, lds_bytes, stream);           \
    }
    NMPC_EMBED_CASE(2, 1, 1)
    NMPC_EMBED_CASE(3, 1, 1)
    NMPC_EMBED_CASE(4, 1, 1)
#undef NMPC_EMBED_CASE
    if (m.nu == 1 && m.ny >= 1 && m.ny <= 2 && mode >= 1) {
        if (m.nx == 5) return embed_dispatch_nx5(m.ny, a, nt, grid, lds_bytes, mode, loop, stream);
        if (m.nx == 6) return embed_dispatch_nx6(m.ny, a, nt, grid, lds_bytes, mode, loop, stream);
        if (m.nx == 7) return embed_dispatch_nx7(m.ny, a, nt, grid, lds_bytes, mode, loop, stream);
        if (m.nx == 8) return embed_dispatch_nx8(m.ny, a, nt, grid, lds_bytes, mode, loop, stream);
    }
    set_error("embed: no kernel instantiated for nx=%d nu=%d ny=%d", m.nx, m.nu, m.ny);
    return 1;
}

}  // namespace nmpc

using namespace nmpc;

extern "C" int nmpc_lpv_embed(const nmpc_model* m, int npts, const float* x, const float* u, const float* xc,
                              const float* lam, const float* w, int n_nodes, float* A, float* B, float* C,
                              void* stream) {
    NMPC_REQUIRE(m && lam && w, "nmpc_lpv_embed: null argument");
    NMPC_REQUIRE(on_device(m->device), "nmpc_lpv_embed: the model lives on device %d, which is not the current device", m->device);
    NMPC_REQUIRE(n_nodes >= 1 && n_nodes <= NMPC_MAX_NODES, "nmpc_lpv_embed: n_nodes=%d out of [1,%d]", n_nodes,
                 NMPC_MAX_NODES);
    NMPC_REQUIRE(npts >= 0, "nmpc_lpv_embed: negative point count");
    if (npts == 0) return 0;                     // an empty batch may come with null pointers
    NodeTable nt;
    nt.n = n_nodes;
    for (int i = 0; i < NMPC_MAX_NODES; ++i) { nt.lam[i] = i < n_nodes ? lam[i] : 0.f; nt.w[i] = i < n_nodes ? w[i] : 0.f; }
    EmbedArgs a{};
    a.x = x; a.u = u; a.xc = xc; a.A = A; a.B = B; a.C = C; a.npts = npts; a.group = 1;
    return embed_launch(m->d, a, nt, (hipStream_t)stream);
}
