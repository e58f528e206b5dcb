// The split-fp16 embedding tile (one workgroup of four waves, 16 scheduling points of one net): shared by the
// embedding kernel (embed.hip) and the per-controller tail of the closed-loop step (mpc_tail.hip).
#pragma once
#include "common.h"
#include "embed.h"

namespace nmpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// first hidden-unit index of k-step s (lane-quarter hq adds 4*hq): see kappa() in model.cpp
__device__ __forceinline__ constexpr int kbase(int s) { return 16 * (s >> 2) + (s & 3); }

// ------------------------------------------------------------------------------------------------
// Split-fp16 variant.  On gfx950 the fp32-input MFMA issues at the fp32 VECTOR rate and does not
// overlap with VALU work (PMC: matrix-busy + VALU-busy cycles = kernel cycles, zero co-execution),
// so the fp32 kernel above runs at (MFMA + VALU) time.  Here every fp32 operand x is written as
//     x = xh + xl,   xh = x with its significand cut to 11 bits (exact in fp16),
//                    xl = fp16(x - xh)   (unscaled, mostly subnormal: gfx950 MFMAs honour subnormals)
// and a product W x is evaluated as  Wh xh + Wh xl + Wl xh  with three v_mfma_f32_16x16x32_f16 into
// ONE fp32 accumulator (fp16 products are exact in fp32; the dropped Wl xl term is 2^-22 relative).
// These run on the matrix pipe at 16x the fp32 rate, which leaves the kernel bound by instruction
// issue: the activation math (v_exp/v_rcp), the splits and the MFMA issue slots themselves.
// Unit numbering, accumulator->operand reuse and the hoisted W0 contraction are as in embed_tile:
// element j of k-step ks of a B fragment is unit s = 8 ks + j = 4 T + r of the accumulator tiles.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 fp32 values (unit order) -> B fragments of the two k-steps, high and low planes.
// H = pairs truncated to fp16 (v_cvt_pkrtz); L = the fp16-rounded remainders a - float(H.lo), b - float(H.hi),
// each formed by one mixed-precision fma (fp16 * -1.0 + fp32, exact before the final rounding).
// The eight v_fma_mix of a k-step are ONE asm block: the compiler's hazard recognizer does not see
// instructions inside inline asm, so the distances they need are built in - the write of a register's low
// half and the read-modify-write of its high half are four instructions apart, and the block ends in the two
// wait states an MFMA needs after a VALU write of one of its operands (one instantiation read a stale half of
// a B operand when these were separate statements with an MFMA one instruction behind the last of them).
__device__ __forceinline__ void split16(const float* v, f16x8* BH, f16x8* BL) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const float* p = v + 8 * ks;
        u32x4 h, l;
#pragma unroll
        for (int m = 0; m < 4; ++m) h[m] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[2 * m], p[2 * m + 1]));
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
        BH[ks] = __builtin_bit_cast(f16x8, h);
        BL[ks] = __builtin_bit_cast(f16x8, l);
    }
}

// RECUR: the nodes of a wave are equally spaced (nt.dlam): e = 2^-a of layer 1 is stepped from node to
// node by one multiplication with 2^(-dlam g) instead of being evaluated (a = lam g + b0 is linear in lam).
// C0, NOUT_ALL: the tile computes output rows C0 .. C0 + NOUT - 1 of a net with NOUT_ALL rows (a net with four
// outputs runs as two tiles of two rows each, so that three workgroups still share a CU; the forward product is
// then evaluated twice; nets with five to eight outputs likewise as three or four tiles, rows 4..7 from the second
// block of reverse fragments).  Nets with more than four inputs read the tables of inputs 4..7 behind the others,
// those with more than eight a third table; the linear branch and the origin Jacobian of nets with more than four
// outputs or eight inputs come from the pack in global memory (sixteen lanes, once per tile).
template <int NA, int NB, int NOUT, bool RECUR = false, int C0 = 0, int NOUT_ALL = NOUT>
__device__ __forceinline__ void embed_tile_h2(const float* __restrict__ pack, float* __restrict__ lds,
                                              const float* __restrict__ pa, const float* __restrict__ pb,
                                              float* __restrict__ oa, float* __restrict__ ob, int tile, int npts,
                                              const int* __restrict__ act, int group, const NodeTable& nt) {
    constexpr int NIN = NA + NB;
    constexpr int TAIL = NIN > 8 ? PK_TAIL_EXT2 : (NIN > 4 ? PK_TAIL_EXT : PK_TAIL);
    constexpr int L_WR = 4096, L_REC1 = 4096 + 4096 * NOUT, L_REC2 = L_REC1 + 256, L_LIN = L_REC2 + 256,
                  L_REC3 = L_REC1 + (PK_REC3 - PK_REC1), L_LIN2 = L_REC1 + (PK_LIN2 - PK_REC1),
                  L_REC4 = L_REC1 + (PK_REC4 - PK_REC1), L_RED = 0;
    constexpr bool XTAB = NIN > 8 || NOUT_ALL > 4;      // linear branch / origin Jacobian from the [8][12] tables
    static_assert(4 * EMBED_TILE * NOUT * NIN <= 4096, "reduction buffer must fit in the WF region");
    static_assert(NIN <= 12 && C0 + NOUT <= NOUT_ALL && NOUT_ALL <= 8, "table sizes");
    constexpr bool STRADDLE = C0 < 4 && C0 + NOUT > 4;       // output rows on both sides of the two reverse blocks of the pack
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, hq = lane >> 4;
    {   // stage the packed net: all loads of a lane in flight before the first LDS write (a load-then-store loop
        // would pay the L2 latency once per 16 bytes)
        constexpr int NVF = 4096 / 4 / EMBED_THREADS, NVR = 4096 * NOUT / 4 / EMBED_THREADS, NV = NVF + NVR;
        static_assert(4096 / 4 % EMBED_THREADS == 0 && TAIL / 4 <= 2 * EMBED_THREADS, "fill plan");
        const f32x4* srcf = reinterpret_cast<const f32x4*>(pack);
        const f32x4* srcr = reinterpret_cast<const f32x4*>(pack + (C0 < 4 ? PK_WR + 4096 * C0 : PK_WR2 + 4096 * (C0 - 4))) - NVF * EMBED_THREADS;
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        const f32x4* src2 = reinterpret_cast<const f32x4*>(pack + PK_REC1);
        f32x4* dst2 = reinterpret_cast<f32x4*>(lds + L_REC1);
        f32x4 stage[NV], stage2 = {0.f, 0.f, 0.f, 0.f}, stage3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if constexpr (STRADDLE) {              // row by row: four vectors per thread and row
                const int c = C0 + (k - NVF) / (NVR / NOUT);
                const f32x4* srow = reinterpret_cast<const f32x4*>(pack + (c < 4 ? PK_WR + 4096 * c : PK_WR2 + 4096 * (c - 4))) - (NVF + (c - C0) * (NVR / NOUT)) * EMBED_THREADS;
                stage[k] = (k < NVF ? srcf : srow)[tid + k * EMBED_THREADS];
            } else {
                stage[k] = (k < NVF ? srcf : srcr)[tid + k * EMBED_THREADS];
            }
        }
        if (tid < TAIL / 4) stage2 = src2[tid];
        if constexpr (TAIL / 4 > EMBED_THREADS) { if (tid + EMBED_THREADS < TAIL / 4) stage3 = src2[tid + EMBED_THREADS]; }
#pragma unroll
        for (int k = 0; k < NV; ++k) dst[tid + k * EMBED_THREADS] = stage[k];
        if (tid < TAIL / 4) dst2[tid] = stage2;
        if constexpr (TAIL / 4 > EMBED_THREADS) { if (tid + EMBED_THREADS < TAIL / 4) dst2[tid + EMBED_THREADS] = stage3; }
    }
    __syncthreads();
    // fragment planes: [plane][T][ks][lane] of 8 halves (16 B)
    const f16x8* WF = reinterpret_cast<const f16x8*>(lds);
    const f16x8* WR = reinterpret_cast<const f16x8*>(lds + L_WR);
    const f32x4* REC1 = reinterpret_cast<const f32x4*>(lds + L_REC1);
    const f32x4* REC2 = reinterpret_cast<const f32x4*>(lds + L_REC2);
    const f32x4* REC3 = reinterpret_cast<const f32x4*>(lds + L_REC3);      // inputs 4..7 (NIN > 4 only)
    const f32x4* REC4 = reinterpret_cast<const f32x4*>(lds + L_REC4);      // inputs 8..11 (NIN > 8 only)

    const int p = tile * EMBED_TILE + n;
    const bool valid = p < npts;
    const int pc = valid ? p : npts - 1;
    const int row = act ? act[pc / group] * group + pc % group : pc;
    float z[NIN];
#pragma unroll
    for (int j = 0; j < NA; ++j) z[j] = pa[(size_t)row * NA + j];
#pragma unroll
    for (int j = 0; j < NB; ++j) z[NA + j] = pb[(size_t)row * NB + j];

    constexpr float C2 = 2.8853900817779268f;
    float g[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 w0 = REC2[kbase(s) + 4 * hq];
        float a = w0[0] * z[0];
#pragma unroll
        for (int j = 1; j < (NIN < 4 ? NIN : 4); ++j) a = fmaf(w0[j], z[j], a);
        if constexpr (NIN > 4) {
            const f32x4 w1 = REC3[kbase(s) + 4 * hq];
#pragma unroll
            for (int j = 4; j < (NIN < 8 ? NIN : 8); ++j) a = fmaf(w1[j - 4], z[j], a);
        }
        if constexpr (NIN > 8) {
            const f32x4 w2 = REC4[kbase(s) + 4 * hq];
#pragma unroll
            for (int j = 8; j < NIN; ++j) a = fmaf(w2[j - 8], z[j], a);
        }
        g[s] = C2 * a;
    }
    float q[NOUT][16];
#pragma unroll
    for (int c = 0; c < NOUT; ++c)
#pragma unroll
        for (int s = 0; s < 16; ++s) q[c][s] = 0.0f;

    const int nn = nt.n;
    const int n0 = (nn * wave) >> 2, n1 = (nn * (wave + 1)) >> 2;
    // Activation math with e = 2^-a (a = c2 * preactivation), r = 1/(1 + e) in [0, 1]:
    //   tanh = 2 r - 1,  (1 - tanh^2) / 4 = r - r^2.  e = inf gives r = 0 and e = 0 gives r = 1, both exact
    //   limits; errors are absolute (~1 ulp of r).
    // Products: W x = Wh (xh + xl) + Wl xh, three fp16 MFMAs into one fp32 accumulator (Wl, xl: the fp16
    // remainders, mostly subnormal, absolute resolution 2^-24).
    float b0[16];
    f32x4 b2[4];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 rec = REC1[kbase(s) + 4 * hq];
        b0[s] = rec[0];
        b2[s >> 2][s & 3] = rec[1];
    }
    // The node weight is applied once per run of equal weights (the host orders the nodes by weight):
    // q holds the sum in units of the current weight and is rescaled when the weight changes.
    float wcur = n0 < n1 ? nt.w[n0] : 0.0f;
    float e1[16], rho[16];
    if constexpr (RECUR) {
        const float lam0 = n0 < n1 ? nt.lam[n0] : 0.0f, dl = nt.dlam[wave];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            e1[s] = __builtin_amdgcn_exp2f(-fmaf(lam0, g[s], b0[s]));
            // the step factor stays finite and non-zero, so inf / 0 (saturated units) propagate as they are
            rho[s] = __builtin_amdgcn_exp2f(fminf(fmaxf(-dl * g[s], -64.0f), 64.0f));
        }
    }
    // The matrix products of a node are NG = 8 (1 + NOUT) groups of three MFMAs on one pair of weight
    // fragments (high and low plane): forward groups in the order (T, ks), reverse groups in the order
    // (c, T, ks).  The fragments are loop invariant and sit at fixed LDS addresses, so the pair of group
    // i + PF is read while group i multiplies (a ring of RING register slots, carried over the node
    // boundary and over the activation phases); sched_barrier fences keep the groups in this order.
    constexpr int NG = 8 * (1 + NOUT), RING = NOUT <= 2 ? 4 : 2, PF = RING - 1;
    static_assert(NG % RING == 0, "ring position must repeat from node to node");
    f16x8 fh[RING], fl[RING];
    auto frag_read = [&](int i, int lo_) {           // i: group index within a node (compile-time after unrolling)
        if (i < 8) {
            const int T = i >> 1, ks = i & 1;
            fh[i % RING] = WF[((0 * 4 + T) * 2 + ks) * 64 + lo_];
            fl[i % RING] = WF[((1 * 4 + T) * 2 + ks) * 64 + lo_];
        } else {
            const int j = i - 8, c = j >> 3, T = (j >> 1) & 3, ks = j & 1;
            fh[i % RING] = WR[c * 1024 + ((0 * 4 + T) * 2 + ks) * 64 + lo_];
            fl[i % RING] = WR[c * 1024 + ((1 * 4 + T) * 2 + ks) * 64 + lo_];
        }
    };
    {
        int lo = lane;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int i = 0; i < PF; ++i) frag_read(i, lo);
    }
    for (int node = n0; node < n1; ++node) {
        const float lam = nt.lam[node];
        const float wn = nt.w[node];
        if (wn != wcur) {                          // wave-uniform, rare
            const float sc = wcur / wn;
#pragma unroll
            for (int c = 0; c < NOUT; ++c)
#pragma unroll
                for (int s = 0; s < 16; ++s) q[c][s] *= sc;
            wcur = wn;
        }
        int lo = lane;
        asm volatile("" : "+v"(lo));               // keeps the fragment reads inside the loop
        float r1[16], wd1[16];                     // r1 = (1 + t1) / 2 is the operand of the forward product
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float e;
            if constexpr (RECUR) e = e1[s];        // stepped to the next node inside the reverse groups below
            else e = __builtin_amdgcn_exp2f(-fmaf(lam, g[s], b0[s]));
            const float r = __builtin_amdgcn_rcpf(1.0f + e);
            r1[s] = r;
            wd1[s] = fmaf(-r, r, r);               // (1 - t1^2) / 4
        }
        f16x8 BH[2], BL[2];
        split16(r1, BH, BL);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc[4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {              // forward: W2 r1
            const int T = i >> 1, ks = i & 1;
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i % RING], BH[ks], ks == 0 ? b2[T] : acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i % RING], BL[ks], acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i % RING], BH[ks], acc[T], 0, 0, 0);
            frag_read(i + PF, lo);
            __builtin_amdgcn_sched_barrier(0);
        }
        float d2q[16];                             // (1 - t2^2) / 4
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-acc[s >> 2][s & 3]));
            d2q[s] = fmaf(-r, r, r);
        }
        split16(d2q, BH, BL);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8 * NOUT; ++j) {       // reverse: ((4 W4[c]) o W2)^T d2, then q[c] += d1 o (.)
            const int i = 8 + j, T = (j >> 1) & 3, ks = j & 1;
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i % RING], BH[ks], ks == 0 ? zero4 : acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i % RING], BL[ks], acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i % RING], BH[ks], acc[T], 0, 0, 0);
            frag_read((i + PF) % NG, lo);          // the last PF reads are the next node's first forward groups
            if constexpr (RECUR)                   // in place, long after 1 + e was formed: no register copies
                if (j < 8) { e1[2 * j] *= rho[2 * j]; e1[2 * j + 1] *= rho[2 * j + 1]; }
            if (ks == 1 && j >= 3) {               // the tile that finished two groups ago
                const int jp = j - 2, cp = jp >> 3, Tp = (jp >> 1) & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) q[cp][4 * Tp + r] = fmaf(acc[Tp][r], wd1[4 * Tp + r], q[cp][4 * Tp + r]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            constexpr int jp = 8 * NOUT - 1, cp = jp >> 3, Tp = (jp >> 1) & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) q[cp][4 * Tp + r] = fmaf(acc[Tp][r], wd1[4 * Tp + r], q[cp][4 * Tp + r]);
        }
    }

    float pj[NOUT][NIN];
#pragma unroll
    for (int c = 0; c < NOUT; ++c)
#pragma unroll
        for (int j = 0; j < NIN; ++j) pj[c][j] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 w0 = REC2[kbase(s) + 4 * hq];
#pragma unroll
        for (int c = 0; c < NOUT; ++c)
#pragma unroll
            for (int j = 0; j < (NIN < 4 ? NIN : 4); ++j) pj[c][j] = fmaf(q[c][s], w0[j], pj[c][j]);
        if constexpr (NIN > 4) {
            const f32x4 w1 = REC3[kbase(s) + 4 * hq];
#pragma unroll
            for (int c = 0; c < NOUT; ++c)
#pragma unroll
                for (int j = 4; j < (NIN < 8 ? NIN : 8); ++j) pj[c][j] = fmaf(q[c][s], w1[j - 4], pj[c][j]);
        }
        if constexpr (NIN > 8) {
            const f32x4 w2 = REC4[kbase(s) + 4 * hq];
#pragma unroll
            for (int c = 0; c < NOUT; ++c)
#pragma unroll
                for (int j = 8; j < NIN; ++j) pj[c][j] = fmaf(q[c][s], w2[j - 8], pj[c][j]);
        }
    }
    const float wlast = 4.0f * wcur;               // the weight of the last run and the 4 of (1 - t1^2) / 4
    float* red = lds + L_RED;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NOUT; ++c)
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            float v = (wlast * lds[L_REC1 + 4 * (C0 + c) + 2]) * pj[c][j];     // weight of the last run, inverse pack scale of the row
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (hq == 0) red[(wave * EMBED_TILE + n) * (NOUT * NIN) + c * NIN + j] = v;
        }
    __syncthreads();
    if (tid < EMBED_TILE && valid) {
        const float sumw = nt.w_sum, wz = nt.w_zero;
        const float* j0 = lds + L_LIN + 16;
        const float* lin = lds + L_LIN;
        const float* j02 = lds + L_LIN2 + 16;
        const float* lin2 = lds + L_LIN2;
#pragma unroll
        for (int c = 0; c < NOUT; ++c)
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                float v = red[(0 * EMBED_TILE + n) * (NOUT * NIN) + c * NIN + j];
#pragma unroll
                for (int wv = 1; wv < 4; ++wv) v += red[(wv * EMBED_TILE + n) * (NOUT * NIN) + c * NIN + j];
                float lv, jv;
                if constexpr (XTAB) { lv = pack[PK_LINX + 12 * (C0 + c) + j]; jv = pack[PK_J0X + 12 * (C0 + c) + j]; }
                else {
                    lv = j < 4 ? lin[4 * (C0 + c) + j] : lin2[4 * (C0 + c) + (j - 4)];
                    jv = j < 4 ? j0[4 * (C0 + c) + j] : j02[4 * (C0 + c) + (j - 4)];
                }
                v = fmaf(sumw, lv, fmaf(wz, jv, v));
                if (j < NA) oa[((size_t)row * NOUT_ALL + C0 + c) * NA + j] = v;
                else ob[((size_t)row * NOUT_ALL + C0 + c) * NB + (j - NA)] = v;
            }
    }
}

}  // namespace nmpc
