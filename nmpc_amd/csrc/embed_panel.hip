// LPV embedding for wide two-hidden-layer nets (hidden width 128 or 256, up to 16 inputs and 8 outputs) on the
// fp16 matrix pipe (reference: mpcUtil.py:6-31 CasADiExp for any simple_res_net; lpv_int.py:4-100):
//     J = sum_nodes w W4 D2 W2 D1 W0,   D = diag(1 - tanh^2) = 4 diag(r - r^2),  r = 1 / (1 + 2^-(c2 a))
// Every fp32 product is formed from three v_mfma_f32_16x16x32_f16 products of fp16 planes, W x = Wh (xh + xl) +
// Wl xh (embed.hip explains the split and its accuracy), which runs at 16x the rate of the fp32-input MFMA the
// general-width kernel (embed_wide.hip) uses.
//
// The 256 x 256 weight planes do not fit LDS, so the blocking is that of a GEMM  C[units x columns] = A B  whose
// A (weights) streams from L2 and whose B (activations) is shared through LDS:
//   * a workgroup (4 waves, one per SIMD) owns CT = PT * NOUTP column tiles of 16: PT point tiles times the
//     NOUTP output rows of the reverse sweep, and walks all quadrature nodes;
//   * wave w owns the hidden units of k-steps [w KW, (w+1) KW) = output tiles [2 w KW, 2 (w+1) KW): it evaluates the
//     activations of those units (VALU), writes their fp16 planes into the LDS panel as ready-made B fragments,
//     and multiplies its own tile rows against the whole panel, so that each weight fragment it loads (2 KB,
//     buffer_load from L2, prefetched through a register ring) feeds 3 CT MFMAs per tile;
//   * forward (W2 r1), reverse (W2^T (d2 o W4[c])) and the contraction with W0 chain through registers: an
//     accumulator pair of tiles (2 ks, 2 ks + 1) is by construction the B operand of k-step ks (PanelNet).
//   * the forward product is taken for a group of nodes at once (FCT column tiles = nodes x point tiles share each
//     fragment of W2), the reverse sweep then runs node by node from the (1 - t2^2) / 4 kept in registers.
// Two barriers per node and three per group order the uses of the panel.  Scales (powers of two chosen by the host
// so that every operand sits towards the top of the fp16 range, where the fp16 remainder is a normal number) are
// undone in the last step.
#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"
#include "embed_panel.h"

namespace nmpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sigm2p(float a) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-a)); }

// 8 fp32 values (one k-step's elements of a lane) -> the B fragment in two planes: H = pairs rounded to fp16,
// L = the fp16-rounded remainders.  One asm block with the wait states an MFMA needs behind it (see split16 in
// embed.hip: the hazard recognizer does not look into inline asm).
__device__ __forceinline__ void split8(const float* p, f16x8& H, f16x8& L) {
    u32x4 h, l;
#pragma unroll
    for (int m = 0; m < 4; ++m)       // v_cvt_pk_f16_f32: round to nearest, so the remainder is at most half an fp16 ulp
        h[m] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{p[2 * m], p[2 * m + 1]}, f16x2));
    unsigned l0, l1, l2, l3;
    asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]),
          "v"(p[6]), "v"(p[7]));
    l[0] = l0; l[1] = l1; l[2] = l2; l[3] = l3;
    H = __builtin_bit_cast(f16x8, h);
    L = __builtin_bit_cast(f16x8, l);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t half_rsrc(const _Float16* p, int nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p), 0, nbytes, 0x00020000);
}
__device__ __forceinline__ f16x8 half_load(__amdgpu_buffer_rsrc_t r, int lane_bytes, int off_bytes) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, off_bytes, 0));
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// layer-1 slopes of the own units in LDS: only where the register budget needs it and the panel leaves room
template <int HT, int PT>
__host__ __device__ constexpr bool GLDS() { return HT == 16 && PT == 1; }
template <int HT, int NOUTP, int PT>
__host__ __device__ constexpr int panel_lds_floats() {
    return (HT / 2) * 8 * 2 * 64 * 4 + NOUTP * 16 * HT + (GLDS<HT, PT>() ? (HT / 8) * PT * 8 * 256 + 16 * HT : 0);
}

template <int HT, int NOUTP, int PT, int RING, bool LOOP>
__global__ __launch_bounds__(256, 1) void embed_panel_kernel(PanelArgs a, NodeTable nt) {
    constexpr int H = 16 * HT, KST = HT / 2, KW = KST / 4, CT = PT * NOUTP;
    constexpr int NF = KW * KST;            // weight stages of one matrix: KW own tile pairs of KST k-steps
    constexpr int FCT = HT >= 16 ? 4 : 8;   // column tiles of the forward product (register budget: (1 - t2^2) / 4 of a group stays in registers)
    constexpr int NGRP = FCT / PT;          // = nodes per group x point tiles
    constexpr int PF = RING - 1;
    constexpr int BR = 4;                   // panel-fragment slots: reads run BR - 1 column tiles ahead
    static_assert(KW >= 1 && NF % RING == 0 && CT <= 8 && FCT <= 8 && NGRP >= 1, "ring position must repeat from node to node");
    extern __shared__ __align__(16) float plds[];
    f16x8* PANEL = reinterpret_cast<f16x8*>(plds);                  // [k-step][column tile][plane][lane]
    float* W4S = plds + KST * 8 * 2 * 64 * 4;                        // [NOUTP][H]
    f32x4* RED = reinterpret_cast<f32x4*>(plds);                     // [wave][column tile][lane], after the last node

    int npts = a.npts;
    if (a.count_dev) npts = min(npts, *a.count_dev * a.group);
    const int tile0 = blockIdx.x * PT;
    if (tile0 * 16 >= npts) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const PanelNet& N = a.net;

    bool valid[PT];
    int row[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (tile0 + pt) * 16 + n;
        valid[pt] = p < npts;
        const int pc = valid[pt] ? p : npts - 1;
        row[pt] = a.act ? a.act[pc / a.group] * a.group + pc % a.group : pc;
    }
    for (int i = tid; i < NOUTP * H; i += 256) W4S[i] = N.w4s[i];
    if constexpr (GLDS<HT, PT>()) for (int i = tid; i < H; i += 256) (W4S + NOUTP * H + (HT / 8) * PT * 8 * 256)[i] = N.cb0[i];

    // per-lane tables of the wave's own units: element j of own k-step i is unit 16 (2 ks + (j >> 2)) + 4 g + (j & 3)
    constexpr bool GL = GLDS<HT, PT>();     // layer-1 slopes in LDS instead of registers (frees 8 KW PT registers)
    float* GLp = W4S + NOUTP * H;           // [KW][PT][8][256] floats when GL
    float* CB0S = GLp + KW * PT * 8 * 256;  // [H] c2 b0 when GL
    float G[GL ? 1 : KW][GL ? 1 : PT][8], CB0[GL ? 1 : KW][8];
    {
        float z[PT][16];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = 0.0f;
                if (j < a.na) v = a.pa[(size_t)row[pt] * a.na + j];
                else if (j < a.na + a.nb) v = a.pb[(size_t)row[pt] * a.nb + (j - a.na)];
                z[pt][j] = v;
            }
#pragma unroll
        for (int i = 0; i < KW; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = 16 * (2 * (wave * KW + i) + (j >> 2)) + 4 * g + (j & 3);
                if constexpr (!GL) CB0[i][j] = N.cb0[unit];
                const f32x4* w0 = reinterpret_cast<const f32x4*>(N.w0g + (size_t)unit * 16);
                const f32x4 wa = w0[0], wb = w0[1], wc = w0[2], wd = w0[3];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    float s = 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s = fmaf(wa[q], z[pt][q], s); s = fmaf(wb[q], z[pt][4 + q], s);
                        s = fmaf(wc[q], z[pt][8 + q], s); s = fmaf(wd[q], z[pt][12 + q], s);
                    }
                    if constexpr (GL) GLp[((i * PT + pt) * 8 + j) * 256 + tid] = s; else G[i][pt][j] = s;
                }
            }
    }
    const __amdgpu_buffer_rsrc_t FWD = half_rsrc(N.fwd, HT * KST * 2048);
    const __amdgpu_buffer_rsrc_t REV = half_rsrc(N.rev, HT * KST * 2048);
    f16x8 W0h[KW], W0l[KW];
    {
        const __amdgpu_buffer_rsrc_t W0T = half_rsrc(N.w0t, KST * 2048);
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            W0h[i] = half_load(W0T, 16 * lane, ((wave * KW + i) * 2 + 0) * 1024);
            W0l[i] = half_load(W0T, 16 * lane, ((wave * KW + i) * 2 + 1) * 1024);
        }
    }
    const float inv_sf = N.inv_sf, s_v = N.s_v;

    // Weight stages: a stage = (own tile pair i, k-step kk) of the forward or the reverse matrix; its four
    // fragments (two tiles, two planes) sit in ring slot s % RING and are requested PF stages ahead.
    const int wbase = wave * KW * 2 * KST * 2048;
    f16x8 Rh[RING][2], Rl[RING][2];
    auto stage_load = [&](bool rev, int s, int lo_) {
        const int i = s / KST, kk = s % KST;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int off = wbase + (((2 * i + t) * KST + kk) * 2) * 1024;
            Rh[s % RING][t] = half_load(rev ? REV : FWD, lo_, off);
            Rl[s % RING][t] = half_load(rev ? REV : FWD, lo_, off + 1024);
        }
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 jacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) jacc[ct] = zero4;
    const int nn = nt.n;
    float wcur = nn > 0 ? nt.w[0] : 0.0f;
    __syncthreads();                                   // W4S in place

    // Nodes go in groups of NGRP: the forward product of a whole group is one pass over W2 (its FCT = NGRP PT column
    // tiles share every weight fragment; alone, a node's forward product would fetch 2 KB per three MFMAs and sit on
    // the L2 port), then the reverse sweep runs node by node on the (1 - t2^2) / 4 kept in registers.
#ifdef NMPC_PANEL_STAMPS
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tq, tq0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#define STAMP0() tq = __builtin_amdgcn_s_memtime()
#define STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[k] += t_ - tq; tq = t_; }
#else
#define STAMP0()
#define STAMP(k)
#endif
    for (int n0 = 0; n0 < nn; n0 += NGRP) {
        STAMP0();
        int lo = 16 * lane, ln = lane;
        asm volatile("" : "+v"(lo), "+v"(ln));          // keeps the loop-invariant loads inside the loop
#pragma unroll
        for (int s = 0; s < PF; ++s) stage_load(false, s, lo);
        __syncthreads();                               // the previous node's reverse panel has been read by every wave
        // layer 1 of the own units at every node of the group; 256 r1 = 128 (1 + t1) is the forward operand (the
        // factor keeps the fp16 remainder of small r1 normal)
        float wd1[KW][PT][8];                          // s_v (1 - t1^2) / 4 of the own units at the node being swept
#pragma unroll
        for (int ng = 0; ng < NGRP; ++ng) {
            const float lam = nt.lam[min(n0 + ng, nn - 1)];
#pragma unroll
            for (int i = 0; i < KW; ++i)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    float r[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float gv, cb;
                        if constexpr (GL) { gv = GLp[((i * PT + pt) * 8 + j) * 256 + tid]; cb = CB0S[16 * (2 * (wave * KW + i) + (j >> 2)) + 4 * g + (j & 3)]; }
                        else { gv = G[i][pt][j]; cb = CB0[i][j]; }
                        const float e = __builtin_amdgcn_exp2f(-fmaf(lam, gv, cb));
                        r[j] = __builtin_amdgcn_rcpf(fmaf(e, 0.00390625f, 0.00390625f));
                        if constexpr (NGRP == 1) {     // a group of one node: its layer-1 slopes are these
                            const float rr = 0.00390625f * r[j], q = s_v * rr;
                            wd1[i][pt][j] = fmaf(-rr, q, q);
                        }
                    }
                    f16x8 bh, bl;
                    split8(r, bh, bl);
                    PANEL[(((wave * KW + i) * FCT + ng * PT + pt) * 2 + 0) * 64 + ln] = bh;
                    PANEL[(((wave * KW + i) * FCT + ng * PT + pt) * 2 + 1) * 64 + ln] = bl;
                }
        }
        __syncthreads();
        STAMP(0);

        // forward: own tile rows of (s_f 2 c2 W2) (256 r1) + 256 s_f c2 (b2 - W2 1), then (1 - t2^2) / 4 of the own units
        float d2q[KW][FCT][8];
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            f32x4 acc[2][FCT];
#pragma unroll
            for (int t = 0; t < 2; ++t) {              // rows 4 g + r of tile 2 ks + t
                const f32x4 cb = *reinterpret_cast<const f32x4*>(N.cb2s + 16 * (2 * (wave * KW + i) + t) + 4 * g);
#pragma unroll
                for (int fc = 0; fc < FCT; ++fc) acc[t][fc] = cb;
            }
            __builtin_amdgcn_sched_barrier(0);
            // panel fragments are read BR - 1 column tiles ahead of their products (one wave per SIMD: nothing else
            // would cover the LDS latency)
            f16x8 pbh[BR], pbl[BR];
#pragma unroll
            for (int q = 0; q < BR - 1; ++q) {
                pbh[q % BR] = PANEL[(q * 2 + 0) * 64 + ln];
                pbl[q % BR] = PANEL[(q * 2 + 1) * 64 + ln];
            }
#pragma unroll
            for (int q = 0; q < KST * FCT; ++q) {
                const int kk = q / FCT, fc = q % FCT, s = i * KST + kk;
                if (q + BR - 1 < KST * FCT) {
                    pbh[(q + BR - 1) % BR] = PANEL[((q + BR - 1) * 2 + 0) * 64 + ln];
                    pbl[(q + BR - 1) % BR] = PANEL[((q + BR - 1) * 2 + 1) * 64 + ln];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t][fc] = MFMA16(Rl[s % RING][t], pbh[q % BR], acc[t][fc]);
                    acc[t][fc] = MFMA16(Rh[s % RING][t], pbl[q % BR], acc[t][fc]);
                    acc[t][fc] = MFMA16(Rh[s % RING][t], pbh[q % BR], acc[t][fc]);
                }
                if (fc == FCT - 1) {                   // the last PF requests are the first reverse stages
                    if (s + PF < NF) stage_load(false, s + PF, lo);
                    else stage_load(true, s + PF - NF, lo);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int fc = 0; fc < FCT; ++fc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float rr = sigm2p(acc[t][fc][r] * inv_sf);
                        d2q[i][fc][4 * t + r] = fmaf(-rr, rr, rr);
                    }
        }

        STAMP(1);
        for (int ng = 0; ng < NGRP; ++ng) {            // not unrolled; d2q[.][0..PT) is the current node's, rotated below
            const int node = n0 + ng;
            if (node >= nn) break;
            const float lam = nt.lam[node];
            const float wn = nt.w[node];
            if (wn != wcur) {                          // uniform, rare: the sum is kept in units of the current weight
                const float sc = wcur / wn;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) jacc[ct] = jacc[ct] * sc;
                wcur = wn;
            }
            if constexpr (NGRP > 1)
#pragma unroll
            for (int i = 0; i < KW; ++i)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float gv, cb;
                        if constexpr (GL) { gv = GLp[((i * PT + pt) * 8 + j) * 256 + tid]; cb = CB0S[16 * (2 * (wave * KW + i) + (j >> 2)) + 4 * g + (j & 3)]; }
                        else { gv = G[i][pt][j]; cb = CB0[i][j]; }
                        const float rr = sigm2p(fmaf(lam, gv, cb));
                        const float q = s_v * rr;
                        wd1[i][pt][j] = fmaf(-rr, q, q);
                    }
            __syncthreads();                           // the panel (forward, or the previous node's reverse) has been read by every wave
            STAMP(2);

            // reverse operands of the own units, one column tile per (point tile, output row); the rows of W4 are
            // read one step ahead (LDS reads and writes share one in-order counter)
            {
                f32x4 w4a[2], w4b[2];
                w4a[0] = *reinterpret_cast<const f32x4*>(W4S + 16 * (2 * (wave * KW)) + 4 * g);
                w4b[0] = *reinterpret_cast<const f32x4*>(W4S + 16 * (2 * (wave * KW) + 1) + 4 * g);
#pragma unroll
                for (int ic = 0; ic < KW * NOUTP; ++ic) {
                    const int i = ic / NOUTP, c = ic % NOUTP, ks = wave * KW + i;
                    if (ic + 1 < KW * NOUTP) {
                        const int i1 = (ic + 1) / NOUTP, c1 = (ic + 1) % NOUTP, ks1 = wave * KW + i1;
                        w4a[(ic + 1) & 1] = *reinterpret_cast<const f32x4*>(W4S + c1 * H + 16 * (2 * ks1) + 4 * g);
                        w4b[(ic + 1) & 1] = *reinterpret_cast<const f32x4*>(W4S + c1 * H + 16 * (2 * ks1 + 1) + 4 * g);
                    }
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        float b[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { b[r] = d2q[i][pt][r] * w4a[ic & 1][r]; b[4 + r] = d2q[i][pt][4 + r] * w4b[ic & 1][r]; }
                        f16x8 ph, pl;
                        split8(b, ph, pl);
                        PANEL[((ks * CT + pt * NOUTP + c) * 2 + 0) * 64 + ln] = ph;
                        PANEL[((ks * CT + pt * NOUTP + c) * 2 + 1) * 64 + ln] = pl;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            STAMP(3);

            // reverse: own tile rows of (s_r W2^T) (d2 o s_c W4[c]) for every column tile, then the contraction of
            // s_v d1 o (.) with s_0 W0^T over the own k-step
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                f32x4 racc[2][CT];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) racc[t][ct] = zero4;
                f16x8 pbh[BR], pbl[BR];
#pragma unroll
                for (int q = 0; q < BR - 1; ++q) {
                    pbh[q % BR] = PANEL[(q * 2 + 0) * 64 + ln];
                    pbl[q % BR] = PANEL[(q * 2 + 1) * 64 + ln];
                }
#pragma unroll
                for (int q = 0; q < KST * CT; ++q) {
                    const int kk = q / CT, ct = q % CT, s = i * KST + kk;
                    if (q + BR - 1 < KST * CT) {
                        pbh[(q + BR - 1) % BR] = PANEL[((q + BR - 1) * 2 + 0) * 64 + ln];
                        pbl[(q + BR - 1) % BR] = PANEL[((q + BR - 1) * 2 + 1) * 64 + ln];
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        racc[t][ct] = MFMA16(Rl[s % RING][t], pbh[q % BR], racc[t][ct]);
                        racc[t][ct] = MFMA16(Rh[s % RING][t], pbl[q % BR], racc[t][ct]);
                        racc[t][ct] = MFMA16(Rh[s % RING][t], pbh[q % BR], racc[t][ct]);
                    }
                    if (ct == CT - 1) stage_load(true, (s + PF) % NF, lo);    // wraps to the next node's first stages (same matrix)
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int pt = ct / NOUTP;
                    float u[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] = racc[j >> 2][ct][j & 3] * wd1[i][pt][j];
                    f16x8 uh, ul;
                    split8(u, uh, ul);
                    jacc[ct] = MFMA16(W0l[i], uh, jacc[ct]);
                    jacc[ct] = MFMA16(W0h[i], ul, jacc[ct]);
                    jacc[ct] = MFMA16(W0h[i], uh, jacc[ct]);
                }
            }
            STAMP(4);
#pragma unroll
            for (int i = 0; i < KW; ++i)               // the next node of the group moves to the front
#pragma unroll
                for (int fc = 0; fc + PT < FCT; ++fc)
#pragma unroll
                    for (int j = 0; j < 8; ++j) d2q[i][fc][j] = d2q[i][fc + PT][j];
        }
    }

#ifdef NMPC_PANEL_STAMPS
    const unsigned long long tq1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
#endif
    // the four waves' partial sums (each over its own hidden units) -> outputs; lane (n, g) holds inputs 4 g + r of point n
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) RED[(wave * CT + ct) * 64 + lane] = jacc[ct] * wcur;
    __syncthreads();
    const float sumw = nt.w_sum, wz = nt.w_zero;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        if ((ct & 3) != wave) continue;
        const int c = ct % NOUTP, pt = ct / NOUTP;
        if (c >= a.nout || !valid[pt]) continue;
        f32x4 v = RED[ct * 64 + lane];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) v = v + RED[(wv * CT + ct) * 64 + lane];
        const float os = N.oscale[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * g + r;
            const float o = fmaf(sumw, N.lin[c * 16 + j], fmaf(wz, N.j0[c * 16 + j], v[r] * os));
            if (j < a.na) a.oa[((size_t)row[pt] * a.nout + c) * a.na + j] = o;
            else if (j < a.na + a.nb) a.ob[((size_t)row[pt] * a.nout + c) * a.nb + (j - a.na)] = o;
        }
    }
#ifdef NMPC_PANEL_STAMPS
    __syncthreads();
    if (tid == 0 && a.nout * a.na >= 8) {
        float* o = a.oa + (size_t)row[0] * a.nout * a.na;
        for (int k = 0; k < 5; ++k) o[k] = (float)tph[k] / (float)nn;
        o[5] = (float)(tq1 - tq0) / (float)nn;
        o[6] = (float)(tq1 - tq0) / (float)(rt1 - rt0) * 0.1f;
    }
#endif
}

template <int HT, int NOUTP, int PT, int RING>
static int panel_launch_t(const PanelArgs& a, const NodeTable& nt, hipStream_t s, bool loop) {
    constexpr size_t lds = sizeof(float) * panel_lds_floats<HT, NOUTP, PT>();
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)embed_panel_kernel<HT, NOUTP, PT, RING, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)embed_panel_kernel<HT, NOUTP, PT, RING, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mark_used_on_device(once);
    }
    const int grid = (a.npts + 16 * PT - 1) / (16 * PT);
    if (loop) hipLaunchKernelGGL((embed_panel_kernel<HT, NOUTP, PT, RING, true>), dim3(grid), dim3(256), lds, s, a, nt);
    else hipLaunchKernelGGL((embed_panel_kernel<HT, NOUTP, PT, RING, false>), dim3(grid), dim3(256), lds, s, a, nt);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

// nt: nodes with lambda = 0 already moved into w_zero, equal weights adjacent (embed_prepare_nodes)
int panel_launch_net(const PanelNet& net, PanelArgs a, const NodeTable& nt, hipStream_t s, bool loop) {
    NMPC_REQUIRE(net.ok, "embed: this net has no split-fp16 panel layout");
    a.net = net;
#define NMPC_PANEL_CASE(HT_, NO_, PT_, RING_) \
    if (net.ht == HT_ && net.nout_pad == NO_) return panel_launch_t<HT_, NO_, PT_, RING_>(a, nt, s, loop);
    NMPC_PANEL_CASE(8, 1, 4, 4) NMPC_PANEL_CASE(8, 4, 2, 4) NMPC_PANEL_CASE(8, 8, 1, 4)
    NMPC_PANEL_CASE(16, 1, 4, 4) NMPC_PANEL_CASE(16, 4, 2, 2) NMPC_PANEL_CASE(16, 8, 1, 4)
#undef NMPC_PANEL_CASE
    set_error("embed: no panel kernel for hidden tiles %d, outputs %d", net.ht, net.nout_pad);
    return 1;
}

}  // namespace nmpc
