// LPV embedding for nets the 64-wide kernel (embed.hip) does not cover: hidden width up to 256,
// up to 16 inputs and 8 outputs (reference: mpcUtil.py:6-31 CasADiExp handles any deepSI
// simple_res_net; LTI.ipynb cell 5 nx = 4, sinCos.ipynb cell 6 nx = 8).  Same quantities as embed.hip:
//     J = sum_nodes w * W4 diag(1 - t2^2) W2 diag(1 - t1^2) W0   (+ linear branch, + nodes at lambda = 0)
//
// Mapping.  One workgroup (4 waves) owns 16 scheduling points, each wave a quarter of the nodes, lane l
// serves point (l & 15) and the hidden units {16 T + 4 (l >> 4) + r}.  All products are exact-fp32
// v_mfma_f32_16x16x4_f32:
//   forward   a2 = (c2 W2) t1 + c2 b2         A-fragments streamed from L2 (the matrix does not fit LDS),
//   reverse   v_c = W2^T (d2 .* 4 W4[c])      the output row c is folded into the B operand, so all rows
//                                            share one set of A-fragments,
//   contract  J_c += W0^T (w d1 .* v_c)       a 16 x H product whose accumulator tile holds J_c[input][point],
// and the accumulator layout of one product is the B layout of the next (kappa() in model.cpp), so the
// activations never leave the lane.  Per-lane arrays are indexed with compile-time constants only
// (the tile loops are unrolled, the output-row loop is not; its accumulators live in LDS).
#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"
#include "embed_wide.h"
#include "embed_panel.h"

#include <cstdlib>
#include <cstring>

namespace nmpc {

// LDS (floats): G[4 HT][64] layer-1 slopes | CB0[H] | CB2[H] | W4S[NOUT][H] | ACC[4][NOUT][64][4]
template <int HT, int NOUT>
__host__ __device__ constexpr int wide_lds_floats() { return 4 * HT * 64 + 2 * 16 * HT + NOUT * 16 * HT + 4 * NOUT * 256; }

template <int HT, int NOUT, bool LOOP>
__global__ __launch_bounds__(256, 1) void embed_wide_kernel(WideArgs a, NodeTable nt) {
    extern __shared__ __align__(16) float wlds[];
    constexpr int H = 16 * HT, KS = 4 * HT;
    float* G = wlds;
    float* CB0 = G + KS * 64;
    float* CB2 = CB0 + H;
    float* W4S = CB2 + H;
    f32x4* ACC = reinterpret_cast<f32x4*>(W4S + NOUT * H);

    int npts = a.npts;
    if (a.count_dev) npts = min(npts, *a.count_dev * a.group);
    const int tile = blockIdx.x;
    if (tile * 16 >= npts) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, hq = lane >> 4;
    const WideNet& N = a.net;

    const int p = tile * 16 + n;
    const bool valid = p < npts;
    const int pc = valid ? p : npts - 1;
    const int row = a.act ? a.act[pc / a.group] * a.group + pc % a.group : pc;
    float z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v = 0.0f;
        if (j < a.na) v = a.pa[(size_t)row * a.na + j];
        else if (j < a.na + a.nb) v = a.pb[(size_t)row * a.nb + (j - a.na)];
        z[j] = v;
    }
    // tables, and the layer-1 slopes c2 (W0 z): wave w fills k-steps [w HT, (w+1) HT)
    for (int i = tid; i < H; i += 256) { CB0[i] = N.cb0[i]; CB2[i] = N.cb2[i]; }
    for (int i = tid; i < NOUT * H; i += 256) W4S[i] = N.w4s[i];
    for (int i = tid; i < 4 * NOUT * 64; i += 256) ACC[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr float C2 = 2.8853900817779268f;
    for (int s = wave * HT; s < (wave + 1) * HT; ++s) {
        const int k = 16 * (s >> 2) + 4 * hq + (s & 3);
        const f32x4* w0 = reinterpret_cast<const f32x4*>(N.w0 + (size_t)k * 16);
        float g = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = w0[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) g = fmaf(w[j], z[4 * q + j], g);
        }
        G[s * 64 + lane] = C2 * g;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t FWD = frag_rsrc(N.fwd, HT * HT * 256);
    const __amdgpu_buffer_rsrc_t REV = frag_rsrc(N.rev, HT * HT * 256);
    const __amdgpu_buffer_rsrc_t W0T = frag_rsrc(N.w0t, HT * 256);
    const int nn = nt.n;
    const int n0 = (nn * wave) >> 2, n1 = (nn * (wave + 1)) >> 2;
    for (int node = n0; node < n1; ++node) {
        const float lam = nt.lam[node];
        const float w4 = 4.0f * nt.w[node];
        int lo = lane, ho = 4 * hq;
        asm volatile("" : "+v"(lo), "+v"(ho));      // keeps loop-invariant table reads out of registers
        float bv[KS], wd1[KS], d2[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 16 * (s >> 2) + (s & 3);
            const float r = sigm2(fmaf(lam, G[s * 64 + lo], CB0[k + ho]));
            bv[s] = fmaf(2.0f, r, -1.0f);            // t1
            wd1[s] = w4 * fmaf(-r, r, r);            // w (1 - t1^2)
            if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bounds the table reads in flight
        }
#pragma unroll
        for (int T = 0; T < HT; ++T) {
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = CB2[16 * T + r + ho];
#pragma unroll
            for (int S = 0; S < HT; ++S) {
                const f32x4 af = frag_load(FWD, 16 * lo, T * HT + S);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bv[4 * S + j], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rr = sigm2(acc[r]);
                d2[4 * T + r] = fmaf(-rr, rr, rr);   // (1 - t2^2) / 4
            }
            __builtin_amdgcn_sched_barrier(0);       // keep each tile's fragment loads inside its own region
        }
        for (int c = 0; c < NOUT; ++c) {             // not unrolled: code size
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bv[s] = d2[s] * W4S[c * H + 16 * (s >> 2) + (s & 3) + ho];
                if ((s & 15) == 15) __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 jc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < HT; ++T) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int S = 0; S < HT; ++S) {
                    const f32x4 af = frag_load(REV, 16 * lo, T * HT + S);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bv[4 * S + j], acc, 0, 0, 0);
                }
                const f32x4 cf = frag_load(W0T, 16 * lo, T);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    jc = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[j], acc[j] * wd1[4 * T + j], jc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4* slot = ACC + (wave * NOUT + c) * 64 + lo;
            *slot = *slot + jc;
        }
    }
    __syncthreads();
    // lane (n, hq) of wave 0 holds J_c[input 4 hq + r] of point n
    if (wave == 0 && valid) {
        const float sumw = nt.w_sum, wz = nt.w_zero;
        for (int c = 0; c < a.nout; ++c) {
            f32x4 v = ACC[c * 64 + lane];
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) v = v + ACC[(wv * NOUT + c) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * hq + r;
                const float o = fmaf(sumw, N.lin[c * 16 + j], fmaf(wz, N.j0[c * 16 + j], v[r]));
                if (j < a.na) a.oa[((size_t)row * a.nout + c) * a.na + j] = o;
                else if (j < a.na + a.nb) a.ob[((size_t)row * a.nout + c) * a.nb + (j - a.na)] = o;
            }
        }
    }
}

template <int HT, int NOUT>
static int wide_launch_t(const WideArgs& a, const NodeTable& nt, int grid, hipStream_t s, bool loop) {
    constexpr size_t lds = sizeof(float) * wide_lds_floats<HT, NOUT>();
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)embed_wide_kernel<HT, NOUT, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)embed_wide_kernel<HT, NOUT, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mark_used_on_device(once);
    }
    if (loop) hipLaunchKernelGGL((embed_wide_kernel<HT, NOUT, true>), dim3(grid), dim3(256), lds, s, a, nt);
    else hipLaunchKernelGGL((embed_wide_kernel<HT, NOUT, false>), dim3(grid), dim3(256), lds, s, a, nt);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

static int wide_launch_net(const WideNet& net, WideArgs a, const NodeTable& nt, hipStream_t s, bool loop) {
    if (net.layers != 2) return deep_launch_net(net, a, nt, s, loop);
    a.net = net;
    const int grid = (a.npts + 15) / 16;
#define NMPC_WIDE_CASE(HT_, NO_) if (net.ht == HT_ && net.nout_pad == NO_) return wide_launch_t<HT_, NO_>(a, nt, grid, s, loop);
    NMPC_WIDE_CASE(4, 1) NMPC_WIDE_CASE(4, 4) NMPC_WIDE_CASE(4, 8)
    NMPC_WIDE_CASE(8, 1) NMPC_WIDE_CASE(8, 4) NMPC_WIDE_CASE(8, 8)
    NMPC_WIDE_CASE(16, 1) NMPC_WIDE_CASE(16, 4) NMPC_WIDE_CASE(16, 8)
#undef NMPC_WIDE_CASE
    set_error("embed: no wide kernel for hidden tiles %d, outputs %d", net.ht, net.nout_pad);
    return 1;
}

// NMPC_EMBED_WIDE=f32 keeps nets of width 128 / 256 on the exact-fp32 kernel of this file (diagnostic);
// otherwise they run on the split-fp16 kernel of embed_panel.hip when their weights suit it
static bool panel_enabled() {
    static std::atomic<int> v{-1};
    int r = v.load(std::memory_order_relaxed);
    if (r < 0) { const char* e = getenv("NMPC_EMBED_WIDE"); r = (e && !strcmp(e, "f32")) ? 0 : 1; v.store(r, std::memory_order_relaxed); }
    return r == 1;
}

static int wide_or_panel(const WideNet& wn, const PanelNet& pn, const WideArgs& a, const NodeTable& nt, hipStream_t s, bool loop) {
    if (pn.ok && panel_enabled()) {
        PanelArgs p{};
        p.pa = a.pa; p.pb = a.pb; p.oa = a.oa; p.ob = a.ob; p.na = a.na; p.nb = a.nb; p.nout = a.nout;
        p.npts = a.npts; p.count_dev = a.count_dev; p.act = a.act; p.group = a.group;
        return panel_launch_net(pn, p, nt, s, loop);
    }
    return wide_launch_net(wn, a, nt, s, loop);
}

// nt: nodes with lambda = 0 already moved into w_zero (embed_launch)
int embed_wide_launch(const ModelDev& m, const EmbedArgs& e, const NodeTable& nt, hipStream_t s, bool loop) {
    const bool do_f = e.x && e.u && e.A && e.B, do_h = e.xc && e.C;
    WideArgs a{};
    a.npts = e.npts; a.count_dev = e.count_dev; a.act = e.act; a.group = e.group < 1 ? 1 : e.group;
    if (do_f) {
        a.pa = e.x; a.pb = e.u; a.oa = e.A; a.ob = e.B; a.na = m.nx; a.nb = m.nu; a.nout = m.nx;
        if (wide_or_panel(m.wide_f, m.panel_f, a, nt, s, loop)) return 1;
    }
    if (do_h) {
        a.pa = e.xc; a.pb = nullptr; a.oa = e.C; a.ob = nullptr; a.na = m.nx; a.nb = 0; a.nout = m.ny;
        if (wide_or_panel(m.wide_h, m.panel_h, a, nt, s, loop)) return 1;
    }
    return 0;
}

}  // namespace nmpc
