// LPV embedding for nets whose number of hidden layers is not two (reference: mpcUtil.py:6-31 CasADiExp loops
// over f_n_hidden_layers; it is the re-expression par_NMPC_linear uses, NMPC.py:517):
//     net(z) = W_lin z + b_lin + W_out t_L + b_out,   t_l = tanh(W_l t_{l-1} + b_l),  t_0 = z,
//     J = sum_nodes w * W_out D_L W_L D_{L-1} ... W_2 D_1 W_0   (+ linear branch, + nodes at lambda = 0),
// D_l = diag(1 - t_l^2) at the node's point lambda z.
//
// Same mapping as the two-layer kernel (embed_wide.hip): one workgroup of four waves owns 16 scheduling points,
// each wave a quarter of the nodes, lane l serves point (l & 15) and hidden units {16 T + 4 (l >> 4) + r};
// exact-fp32 v_mfma_f32_16x16x4_f32 with the A-fragments of W_l / W_l^T streamed from L2, accumulator layout =
// B layout of the next product.  The layer loops are run-time loops over one unrolled product body each
// (forward, reverse between hidden layers, reverse into the W_0 contraction).  The forward sweep leaves
// (1 - t_l^2)/4 of the layers l >= 2 in LDS (one float per lane and unit: conflict-free), the reverse sweep
// reads them back; layer 1's stays in registers with the node weight folded in.  Powers of four: the output
// matrix is stored times 4 (for the last layer's quarter), W_l^T times 4 for l > 2 (the layer below), and the
// node weight times 4 (layer 1); with one hidden layer the output matrix is stored as it is.
#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"
#include "embed_wide.h"

namespace nmpc {

// LDS (floats): G[4 HT][64] layer-1 slopes | CB[L][H] | W4S[NOUT][H] | ACC[4][NOUT][64][4] | DL[L-1][4 HT][256]
static size_t deep_lds_floats(int HT, int NOUT, int L) {
    return (size_t)4 * HT * 64 + (size_t)L * 16 * HT + (size_t)NOUT * 16 * HT + (size_t)4 * NOUT * 256 + (size_t)(L - 1) * 4 * HT * 256;
}

template <int HT, int NOUT, bool LOOP>
__global__ __launch_bounds__(256, 1) void embed_deep_kernel(WideArgs a, NodeTable nt) {
    extern __shared__ __align__(16) float dlds[];
    constexpr int H = 16 * HT, KS = 4 * HT;
    const WideNet& N = a.net;
    const int L = N.layers;
    float* G = dlds;
    float* CB = G + KS * 64;
    float* W4S = CB + L * H;
    f32x4* ACC = reinterpret_cast<f32x4*>(W4S + NOUT * H);
    float* DL = reinterpret_cast<float*>(ACC + 4 * NOUT * 64);

    int npts = a.npts;
    if (a.count_dev) npts = min(npts, *a.count_dev * a.group);
    const int tile = blockIdx.x;
    if (tile * 16 >= npts) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, hq = lane >> 4;

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
    for (int i = tid; i < H; i += 256) CB[i] = N.cb0[i];
    for (int i = tid; i < (L - 1) * H; i += 256) CB[H + i] = N.cb2[i];
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

    const int nm = L - 1;
    const __amdgpu_buffer_rsrc_t FWD = frag_rsrc(N.fwd, nm * HT * HT * 256);
    const __amdgpu_buffer_rsrc_t REV = frag_rsrc(N.rev, nm * HT * HT * 256);
    const __amdgpu_buffer_rsrc_t W0T = frag_rsrc(N.w0t, HT * 256);
    const int nn = nt.n;
    const int n0 = (nn * wave) >> 2, n1 = (nn * (wave + 1)) >> 2;
    for (int node = n0; node < n1; ++node) {
        const float lam = nt.lam[node];
        const float w4 = 4.0f * nt.w[node];
        int lo = lane, ho = 4 * hq, to = tid;
        asm volatile("" : "+v"(lo), "+v"(ho), "+v"(to));      // keeps loop-invariant table reads out of registers
        float bv[KS], wd1[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 16 * (s >> 2) + (s & 3);
            const float r = sigm2(fmaf(lam, G[s * 64 + lo], CB[k + ho]));
            bv[s] = fmaf(2.0f, r, -1.0f);            // t1
            wd1[s] = w4 * fmaf(-r, r, r);            // w (1 - t1^2)
            if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        // forward sweep over the layers l = 2..L: t_l into bv, (1 - t_l^2)/4 into DL[l-2]
        for (int li = 0; li < nm; ++li) {
            float tn[KS];
            const float* cb = CB + (li + 1) * H;
            float* dl = DL + (size_t)li * KS * 256;
#pragma unroll
            for (int T = 0; T < HT; ++T) {
                f32x4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = cb[16 * T + r + ho];
#pragma unroll
                for (int S = 0; S < HT; ++S) {
                    const f32x4 af = frag_load(FWD, 16 * lo, li * HT * HT + T * HT + S);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bv[4 * S + j], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rr = sigm2(acc[r]);
                    tn[4 * T + r] = fmaf(2.0f, rr, -1.0f);
                    dl[(4 * T + r) * 256 + to] = fmaf(-rr, rr, rr);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) bv[s] = tn[s];
        }
        for (int c = 0; c < NOUT; ++c) {             // not unrolled: code size
            f32x4 jc = {0.f, 0.f, 0.f, 0.f};
            if (nm == 0) {
                // one hidden layer: J_c += W0^T (w d1 .* W_out[c])
#pragma unroll
                for (int T = 0; T < HT; ++T) {
                    const f32x4 cf = frag_load(W0T, 16 * lo, T);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = wd1[4 * T + j] * W4S[c * H + 16 * T + j + ho];
                        jc = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[j], v, jc, 0, 0, 0);
                    }
                }
            } else {
                {
                    const float* dl = DL + (size_t)(nm - 1) * KS * 256;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        bv[s] = dl[s * 256 + to] * W4S[c * H + 16 * (s >> 2) + (s & 3) + ho];
                        if ((s & 15) == 15) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // reverse between hidden layers: v_{l-1} = (1 - t_{l-1}^2)/4 .* (4 W_l^T v_l),  l = L..3
                for (int li = nm - 1; li >= 1; --li) {
                    float tn[KS];
                    const float* dl = DL + (size_t)(li - 1) * KS * 256;
#pragma unroll
                    for (int T = 0; T < HT; ++T) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int S = 0; S < HT; ++S) {
                            const f32x4 af = frag_load(REV, 16 * lo, li * HT * HT + T * HT + S);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bv[4 * S + j], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) tn[4 * T + j] = acc[j] * dl[(4 * T + j) * 256 + to];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < KS; ++s) bv[s] = tn[s];
                }
                // layer 2 into the contraction with W_0
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
static int deep_launch_t(const WideArgs& a, const NodeTable& nt, int grid, hipStream_t s, bool loop) {
    const size_t lds = sizeof(float) * deep_lds_floats(HT, NOUT, a.net.layers);
    NMPC_REQUIRE(lds <= 160 * 1024, "embed: %d hidden layers of %d units (outputs padded to %d) need %zu KB of LDS, more than the 160 KB of a CU",
                 a.net.layers, 16 * HT, NOUT, lds / 1024);
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)embed_deep_kernel<HT, NOUT, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)embed_deep_kernel<HT, NOUT, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mark_used_on_device(once);
    }
    if (loop) hipLaunchKernelGGL((embed_deep_kernel<HT, NOUT, true>), dim3(grid), dim3(256), lds, s, a, nt);
    else hipLaunchKernelGGL((embed_deep_kernel<HT, NOUT, false>), dim3(grid), dim3(256), lds, s, a, nt);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

int deep_launch_net(const WideNet& net, WideArgs a, const NodeTable& nt, hipStream_t s, bool loop) {
    NMPC_REQUIRE(net.layers >= 1 && net.layers <= 4, "embed: %d hidden layers unsupported", net.layers);
    a.net = net;
    const int grid = (a.npts + 15) / 16;
#define NMPC_DEEP_CASE(HT_, NO_) if (net.ht == HT_ && net.nout_pad == NO_) return deep_launch_t<HT_, NO_>(a, nt, grid, s, loop);
    NMPC_DEEP_CASE(4, 1) NMPC_DEEP_CASE(4, 4) NMPC_DEEP_CASE(4, 8)
    NMPC_DEEP_CASE(8, 1) NMPC_DEEP_CASE(8, 4) NMPC_DEEP_CASE(8, 8)
    NMPC_DEEP_CASE(16, 1) NMPC_DEEP_CASE(16, 4) NMPC_DEEP_CASE(16, 8)
#undef NMPC_DEEP_CASE
    set_error("embed: no kernel for hidden tiles %d, outputs %d, %d hidden layers", net.ht, net.nout_pad, net.layers);
    return 1;
}

}  // namespace nmpc
