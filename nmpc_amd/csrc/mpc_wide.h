// Condensing + QP for LONG horizons (reference: NMPC_test.py:206-238 with Nc of 16..63; the reference's
// long-horizon runs use Nc = 50): one WORKGROUP of four wavefronts per controller instead of one wavefront.
//
// A QP with n = Nc + 1 variables and 4 Nc + 2 rows keeps a single wavefront busy for about a million cycles
// (every phase is bound by that one wave's instruction issue and LDS latency), so here the row-parallel
// phases run on all 256 lanes and the storage is cut to what the structure of the problem needs:
//   * the rows come in pairs  -a'u - eps <= h-,  +a'u - eps <= h+  (input bounds: a = e_i; output bounds:
//     a = (C_i Gamma_{i,:})'), and the Hessian is blkdiag(G, rho), so in the solver's coordinates
//     zt = L'z both rows of a pair have the normals [+-alpha; 1/sqrt(rho)] with ONE vector
//     alpha = L_G^-1 a: the pair table Al[k][pair] holds alpha, half the rows of the dense formulation,
//     and the slack variable never enters a product;
//   * lane 2 p + h owns row h of pair p and keeps the entries k = 2 m + h of alpha_p in REGISTERS (they do not
//     change during the iteration): every Nc-long dot product of a pair is split between the pair's two
//     lanes, reads only its broadcast operand from LDS with all loads in flight at once, and is combined
//     with one DPP swap;
//   * the working set's factor is kept as its INVERSE M = Lw^-1 (appending a row costs the vector r that the
//     step computes anyway), so the two triangular solves of an iteration are two mat-vec products
//     whose lanes do not wait for each other;
//   * the small sequential parts of the dual active-set iteration (these products, step lengths, the
//     decision which row enters or leaves) are computed by every wave redundantly from the same LDS data,
//     bit for bit the same, which leaves two workgroup barriers per iteration (candidate search, Gram
//     column of the candidate);
//   * the Cholesky factorisation of G and the final back substitution run in one wave with one matrix row /
//     column per lane in registers (pivots through v_readlane); the forward substitution of all pair
//     columns is a fully unrolled right-looking sweep over the registers.
// KH = rows of the pair table per lane (horizons up to 2 KH): the unrolled loops run to 2 KH without
// tests, the LDS arrays are padded with zero rows up to there.
// The algorithm is the Goldfarb-Idnani iteration of qp_gi.h (same pivoting rule, tolerances and status
// codes); only the data layout and the work distribution differ.
#pragma once
#include <type_traits>

#include "mpc_kernels.h"

namespace nmpc {

constexpr int WIDE_THREADS = 256;
constexpr int WIDE_SLOW_GRID = 256;   // workgroups that walk the slow list (one per CU is resident)

__host__ __device__ inline int wide_npairs(int Nc) { return 2 * Nc + 1; }
// Columns of the pair table: the pairs, then the linear term; padded to 16 mod 32 so that the two lanes of a
// pair, which read rows k and k+1 of the table in the same instruction, fall into different halves of the banks.
__host__ __device__ inline int wide_cols(int Nc) { return ((wide_npairs(Nc) + 1 + 15) / 32) * 32 + 16; }
// instantiated: nx = 2 with KH 16 / 25 / 32 (horizons to 32 / 50 / 63), nx = 3, 4 with KH 32
__host__ __device__ inline int wide_kh(int Nc, int nx) { return nx != 2 ? 32 : (Nc <= 32 ? 16 : (Nc <= 50 ? 25 : 32)); }

// doubles of scratch wide_hessian needs
__host__ __device__ inline int wide_hessian_scratch(int Nc, int nx) { return 4 * Nc * nx * nx; }

struct WideLds {
    double *Lm, *Lw, *Mi, *Al, *Pm, *sseg, *Ak, *Bk, *Ck, *qd, *xi, *x0, *Xs, *Us, *uold, *hv, *invdL, *invdW, *Dp, *usol, *wbuf, *red;
    int* Wtab;
    // The factor of the Hessian (transform, recovery of z) and the working-set Gram matrix and inverse factor
    // (active-set iteration only) are never live together: they share one region, and the former sits in registers
    // while the iteration runs (wide_solve).  Pm: the prefix products P_m = A_m .. A_1 (P_0 = I), from which every
    // block of Gamma follows as Gamma_{i,j} = P_{i-j} B_j (mpcUtil.py:93-104) - Gamma itself is never formed.
    __host__ __device__ static int shared_doubles(int Nc, int nx, int KH) {
        const int n = Nc + 1, a = 2 * KH * Nc, b = 2 * n * n;
        return a > b ? a : b;
    }
    __host__ __device__ static int doubles(int Nc, int nx, int KH) {
        const int n = Nc + 1, NPc = wide_cols(Nc), R = 2 * KH;
        return shared_doubles(Nc, nx, KH) + R * NPc + 2 * Nc * nx * nx + wide_hessian_scratch(Nc, nx) + 5 * Nc * nx + 4 + 2 * Nc + 2 * wide_npairs(Nc) + 2 +
               R + n + NPc + n + 3 * 4 * 64 + 64 + 2 * n + 2;
    }
    __device__ void carve(double* p, int Nc, int nx, int KH) {
        const int n = Nc + 1, NPc = wide_cols(Nc), R = 2 * KH;
        Lm = p; Lw = p; Mi = p + n * n; p += shared_doubles(Nc, nx, KH); Al = p; p += R * NPc;
        Pm = p; p += Nc * nx * nx; sseg = p; p += 4 * Nc * nx * nx;
        Ak = p; p += Nc * nx * nx; Bk = p; p += Nc * nx; Ck = p; p += Nc * nx; qd = p; p += Nc * nx; xi = p; p += Nc * nx;
        Xs = p; p += Nc * nx; x0 = p; p += 4; Us = p; p += Nc; uold = p; p += Nc; hv = p; p += 2 * wide_npairs(Nc) + 2;
        invdL = p; p += R; invdW = p; p += n; Dp = p; p += NPc; usol = p; p += n; wbuf = p; p += 3 * 4 * 64; red = p; p += 64;
        Wtab = reinterpret_cast<int*>(p);     // n pair ids, then n signs as doubles behind them
    }
};

// red: [0..3] score, [4..7] slack of the candidate, [8..11] row (as double), [12] sp after a drop, [13] flag;
// [16..47] the carried working set's row ids as bytes, one copy of 64 per wave
__device__ __forceinline__ double dpp_swap1(double v) { return dpp_d<DPP_XOR1>(v); }

// Cholesky of the leading n x n block of Lm (row-major, ld n, lower triangle valid; rows n .. NMAX-1 exist and
// are zero), one wave, row i in the registers of lane i.  Rows take the update over their whole tail: entries
// right of the diagonal are never read.  invd[j] = 1 / L[j][j].
// The pivot column's entries are fetched eight at a time and only then used: one wave alone on its SIMD pays the
// full latency of every v_readlane -> v_fma pair that is issued back to back.  (Template recursion instead of
// loops: every register index is a compile-time constant by construction.)
template <int J, int C0, int NMAX>
__device__ __forceinline__ void wide_chol_tail(double (&row)[NMAX]) {
    if constexpr (C0 < NMAX) {
        constexpr int N = NMAX - C0 < 8 ? NMAX - C0 : 8;
        double pv[N];
#pragma unroll
        for (int u = 0; u < N; ++u) pv[u] = readlane_d(row[J], C0 + u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < N; ++u) row[C0 + u] = fma(-row[J], pv[u], row[C0 + u]);
        __builtin_amdgcn_sched_barrier(0);
        wide_chol_tail<J, C0 + 8, NMAX>(row);
    }
}
template <int J, int NMAX>
__device__ __forceinline__ void wide_chol_steps(double (&row)[NMAX], double* invd, int n, bool& ok) {
    if constexpr (J < NMAX) {
        if (J < n) {                                  // uniform
            const int lane = threadIdx.x & 63;
            const double d = readlane_d(row[J], J);
            if (!(d > 0.0)) ok = false;
            const double inv = rsqrt_d(d > 0.0 ? d : 1.0);
            row[J] *= inv;
            if (lane == 0) invd[J] = inv;
            wide_chol_tail<J, J + 1, NMAX>(row);
        }
        wide_chol_steps<J + 1, NMAX>(row, invd, n, ok);
    }
}
template <int NMAX>
__device__ __forceinline__ bool wide_chol_regs(double* Lm, double* invd, int n) {
    const int lane = threadIdx.x & 63;
    double row[NMAX];
    const int lr = lane < n ? lane : 0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
        const double v = Lm[lr * n + (c < n ? c : 0)];
        row[c] = (lane < n && c <= lane) ? v : 0.0;
    }
    bool ok = true;
    wide_chol_steps<0, NMAX>(row, invd, n, ok);
    if (lane < n) {
#pragma unroll
        for (int c = 0; c < NMAX; ++c)
            if (c <= lane && c < n) Lm[lane * n + c] = row[c];
    }
    return ok;
}


// ---------------------------------------------------------------------------------------------------------------
// Condensing from the prefix products P_m = A_m .. A_1 (P_0 = I).  The reference's matrices are
//   Gamma_{i,j} = P_{i-j} B_j,  Phi_i = P_i A_0,  F0_i = (sum_{j<=i} P_j) f0      (mpcUtil.py:84-104, 322-331)
// so every quantity the QP needs is a sum over LAGS of nx x nx blocks, O(Nc^2) of them instead of the O(Nc^3)
// block products of forming Gamma' Omega Gamma from Gamma; Gamma is never stored.
// LT: any struct with the members Pm, Ak, Bk, Ck, qd, xi, x0, Xs, Us, Lm (WideLds, FastLds).

// wave 0 only: Pm and the free response xi_i = P_i A_0 x0 + (sum_{j<=i} P_j) f0 (lane i).  Both are prefix
// operations over the horizon - an ordered product of nx x nx matrices and a sum of nx-vectors - and run as
// inclusive scans over the lanes (six doubling rounds) instead of Nc dependent steps.
template <int NX, class LT>
__device__ __forceinline__ void wide_prefix(const MpcConst& c, LT& L, int Nc) {
    constexpr int NN = NX * NX;
    const int lane = threadIdx.x & 63;
    double M[NN];                                      // lane m: A_m (m >= 1), the identity on lane 0 and beyond the horizon
#pragma unroll
    for (int e = 0; e < NN; ++e) {
        const double a = L.Ak[(lane >= 1 && lane < Nc ? lane : 0) * NN + e];
        M[e] = (lane >= 1 && lane < Nc) ? a : ((e / NX == e % NX) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {           // lane m: (A_m .. A_{m-off+1}) (A_{m-off} .. ) - later factors on the left
        double Y[NN], Pn[NN];
#pragma unroll
        for (int e = 0; e < NN; ++e) Y[e] = __shfl_up(M[e], off, 64);
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a = fma(M[r * NX + l], Y[l * NX + cc], a);
                Pn[r * NX + cc] = a;
            }
        if (lane >= off) {
#pragma unroll
            for (int e = 0; e < NN; ++e) M[e] = Pn[e];
        }
    }
    if (lane < Nc) {
#pragma unroll
        for (int e = 0; e < NN; ++e) L.Pm[lane * NN + e] = M[e];
    }
    double w0[NX], sf[NX], v[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) a = fma(L.Ak[r * NX + l], L.x0[l], a);
        w0[r] = a;
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        double a = 0.0, b2 = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) { a = fma(M[r * NX + l], w0[l], a); b2 = fma(M[r * NX + l], c.f0[l], b2); }
        v[r] = a; sf[r] = b2;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int r = 0; r < NX; ++r) { const double y = __shfl_up(sf[r], off, 64); sf[r] += lane >= off ? y : 0.0; }
    }
    if (lane < Nc) {
#pragma unroll
        for (int r = 0; r < NX; ++r) L.xi[lane * NX + r] = v[r] + sf[r];
    }
}

// Whole workgroup (contains barriers).  Hessian, lower triangle into Lm (row-major, ld Nc):
//   G_ab = 2 (sum_{i>=a} Gamma_{i,a}' Q Gamma_{i,b} + R [a = b]) = 2 (B_a' S(a-b, Nc-1-a) B_b + R [a = b]),
//   S(d, m) = sum_{m'<=m} P_m'' Q P_{m'+d}.
// Thread (lag d = lane, quarter = wave) walks a quarter of the lag's range m = 0 .. Nc-1-d, accumulates the running
// sum from the quarter's start and emits the entry (a, b) = (Nc-1-m, Nc-1-m-d) at every step; the quarters' totals
// meet in `sseg` and each thread then adds what the earlier quarters contribute to its entries.
// Linear term F_a = 2 (B_a' sum_m P_m' qd_{a+m} - R Us_a), the same way, handed to put_f(a, F_a).
template <int NX, class LT, class PutF>
__device__ __forceinline__ void wide_hessian(const MpcConst& c, LT& L, int Nc, double* sseg, PutF put_f) {
    constexpr int NN = NX * NX;
    const int tid = threadIdx.x, d = tid & 63, seg = tid >> 6;
    const bool lag = d < Nc;
    const int len = lag ? Nc - d : 0;
    const int m0 = seg * len / 4, m1 = (seg + 1) * len / 4;
    const int steps = (Nc + 3) / 4 + 1;                     // uniform bound of m1 - m0
    double S[NN], f[NX];
#pragma unroll
    for (int e = 0; e < NN; ++e) S[e] = 0.0;
#pragma unroll
    for (int r = 0; r < NX; ++r) f[r] = 0.0;
    auto quad = [&](const double* Sx, int a_, int b_) {       // B_a' Sx B_b
        double g = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double sb = 0.0;
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) sb = fma(Sx[r * NX + cc], L.Bk[b_ * NX + cc], sb);
            g = fma(L.Bk[a_ * NX + r], sb, g);
        }
        return g;
    };
#pragma unroll 2
    for (int st = 0; st < steps; ++st) {
        const int m = m0 + st;
        const bool on = m < m1;
        const int mm = on ? m : 0, md = on ? m + d : 0;
        double Pq[NN], Pd[NN], QP[NN], qv[NX];
#pragma unroll
        for (int e = 0; e < NN; ++e) { Pq[e] = L.Pm[mm * NN + e]; Pd[e] = L.Pm[md * NN + e]; }
#pragma unroll
        for (int k = 0; k < NX; ++k) qv[k] = L.qd[md * NX + k];          // (lane a = d of the linear term: i = a + m)
        if (on) {
#pragma unroll
            for (int r = 0; r < NX; ++r)
#pragma unroll
                for (int cc = 0; cc < NX; ++cc) {
                    double a = 0.0;
#pragma unroll
                    for (int l = 0; l < NX; ++l) a = fma(c.Q[r * NX + l], Pd[l * NX + cc], a);
                    QP[r * NX + cc] = a;
                }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
#pragma unroll
                for (int cc = 0; cc < NX; ++cc)
#pragma unroll
                    for (int k = 0; k < NX; ++k) S[r * NX + cc] = fma(Pq[k * NX + r], QP[k * NX + cc], S[r * NX + cc]);
#pragma unroll
                for (int k = 0; k < NX; ++k) f[r] = fma(Pq[k * NX + r], qv[k], f[r]);
            }
            const int a_ = Nc - 1 - m, b_ = a_ - d;
            L.Lm[a_ * Nc + b_] = 2.0 * (quad(S, a_, b_) + (d == 0 ? c.R : 0.0));
        }
    }
    if (lag) {
#pragma unroll
        for (int e = 0; e < NN; ++e) sseg[(seg * Nc + d) * NN + e] = S[e];
    }
    __syncthreads();
    if (lag && seg > 0) {
        double So[NN];
#pragma unroll
        for (int e = 0; e < NN; ++e) So[e] = 0.0;
        for (int s2 = 0; s2 < seg; ++s2)
#pragma unroll
            for (int e = 0; e < NN; ++e) So[e] += sseg[(s2 * Nc + d) * NN + e];
        for (int m = m0; m < m1; ++m) {
            const int a_ = Nc - 1 - m, b_ = a_ - d;
            L.Lm[a_ * Nc + b_] += 2.0 * quad(So, a_, b_);
        }
    }
    __syncthreads();                                               // sseg is free again: the linear term's partial sums
    if (lag) {
#pragma unroll
        for (int r = 0; r < NX; ++r) sseg[(seg * Nc + d) * NX + r] = f[r];
    }
    __syncthreads();
    if (tid < Nc) {
        double fa = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const double fr = (sseg[(0 * Nc + tid) * NX + r] + sseg[(1 * Nc + tid) * NX + r]) +
                              (sseg[(2 * Nc + tid) * NX + r] + sseg[(3 * Nc + tid) * NX + r]);
            fa = fma(L.Bk[tid * NX + r], fr, fa);
        }
        put_f(tid, 2.0 * (fa - c.R * L.Us[tid]));
    }
}

// entry k of the untransformed normal of pair p: input pairs (p even, p < 2 Nc) are unit vectors, output pair of
// y_{a+1} is (C_a Gamma_{a,:})' = (C_a P_{a-k} B_k)_k for k <= a; pair 1 (the measured y_0) has no dependence on u
template <int NX, class LT>
__device__ __forceinline__ double wide_col_entry(const LT& L, int Nc, int p, int k) {
    if (!(p & 1) && p < 2 * Nc) return (p >> 1) == k ? 1.0 : 0.0;
    const int a_ = p == 2 * Nc ? Nc - 1 : (p - 3) / 2;
    if (p < 3 || k > a_ || k >= Nc) return 0.0;
    const double* P = L.Pm + (a_ - k) * NX * NX;
    double cg = 0.0;
#pragma unroll
    for (int cc = 0; cc < NX; ++cc) {
        double vc = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) vc = fma(L.Ck[a_ * NX + r], P[r * NX + cc], vc);
        cg = fma(vc, L.Bk[k * NX + cc], cg);
    }
    return cg;
}

// whole workgroup (contains barriers): x_pred = Phi x0 + Gamma u + F0 into `out` (may be xi itself),
// x_i = xi_i + sum_{j<=i} P_{i-j} (B_j u_j)
template <int NX, class LT>
__device__ __forceinline__ void wide_predict_into(LT& L, int Nc, const double* usol, double* out) {
    const int tid = threadIdx.x;
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) L.qd[e] = L.Bk[e] * usol[e / NX];      // qd is free by now
    __syncthreads();
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) {
        const int i = e / NX, r = e % NX;
        double a = L.xi[e];
#pragma unroll 4
        for (int j = 0; j <= i; ++j) {
            const double* P = L.Pm + (i - j) * NX * NX + r * NX;
#pragma unroll
            for (int l = 0; l < NX; ++l) a = fma(P[l], L.qd[j * NX + l], a);
        }
        out[e] = a;
    }
    __syncthreads();
}
template <int NX, class LT>
__device__ __forceinline__ void wide_predict(LT& L, int Nc, const double* usol) { wide_predict_into<NX>(L, Nc, usol, L.xi); }

struct WideDump { double *Ge, *Fe, *Le, *hv; };

// Working set carried from one solve of a controller to its next (row ids 2 pair + sign bit, at most 64):
// `in` / n_in the rows to start from (n_in = 0: cold start), `out` / n_out receive the final working set.
struct WideWarm { const unsigned char* in; int n_in; unsigned char* out; int* n_out; int fast; };

// On entry Ak, Bk, Ck, x0, Xs, Us are loaded (and a barrier passed).  On exit usol[0..Nc) = u, usol[Nc] = eps,
// xi = predicted states; returns the solver status (uniform), *iters_out the active-set iterations.
// st (diagnostics): clock stamps [0] entry, after [1] recurrences [2] qd [3] Hessian / rows [4] Cholesky
// [5] transform [6] slack init [7] active-set loop [8] recovery of z [9] prediction; [10] iterations;
// [11] the whole solve in ticks of the constant 100 MHz clock; the closed-loop kernel adds [12] kernel entry,
// [13] kernel exit, [14] the LPV pass; [15] end of the warm start (between [6] and [7]).
template <int NX, int KH>
__device__ __forceinline__ int wide_solve(const MpcConst& c, WideLds& L, double ymeas, int* iters_out, const WideDump* dump,
                                          unsigned long long* st = nullptr, const WideWarm* warm = nullptr) {
    constexpr int R = 2 * KH;
    const int tid = threadIdx.x, lane = tid & 63;
#define WIDE_STAMP(i) do { if (st && tid == 0) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
    WIDE_STAMP(0);
    const unsigned long long rt0 = st ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nc = c.Nc, n = Nc + 1, NP = wide_npairs(Nc), NPc = wide_cols(Nc);
    double* Al = L.Al;
    // ---- P1: wave 0 forms the prefix products and the free response (wide_prefix); the other waves clear the pair
    //          table, the inverse factor and the padding rows of Lm
    if (wave == 0) {
        wide_prefix<NX>(c, L, Nc);
    } else {
        const int t3 = tid - 64, n3 = WIDE_THREADS - 64;
        for (int e = t3; e < R * NPc; e += n3) Al[e] = 0.0;
        for (int e = Nc * Nc + t3; e < R * Nc; e += n3) L.Lm[e] = 0.0;
        for (int e = Nc + t3; e < R; e += n3) L.invdL[e] = 0.0;
        for (int e = t3; e < 3 * 4 * 64; e += n3) L.wbuf[e] = 0.0;
    }
    __syncthreads();
    WIDE_STAMP(1);
    // ---- P2: qd_i = Q (xi_i - Xs_i)
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) {
        const int i = e / NX, r = e % NX;
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) a = fma(c.Q[r * NX + l], L.xi[i * NX + l] - L.Xs[i * NX + l], a);
        L.qd[e] = a;
    }
    if (tid < Nc) Al[tid * NPc + 2 * tid] = 1.0;             // pair 2 i: the bounds on u_i, a = e_i
    __syncthreads();
    WIDE_STAMP(2);
    // ---- P3: Hessian (lower triangle) and linear term from the prefix products (wide_hessian), then the
    //          output-row coefficients C_a Gamma_{a,b} = C_a P_{a-b} B_b, one item per pair (a, b <= a)
    wide_hessian<NX>(c, L, Nc, L.sseg, [&](int a_, double fv) { Al[a_ * NPc + NP] = fv; });
    {
        const int items = Nc * (Nc + 1) / 2;
        for (int t = tid; t < items; t += WIDE_THREADS) {
            int a_ = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((a_ + 1) * (a_ + 2) / 2 <= t) ++a_;
            while (a_ * (a_ + 1) / 2 > t) --a_;
            const int b_ = t - a_ * (a_ + 1) / 2;
            const int pair = a_ + 1 < Nc ? 2 * (a_ + 1) + 1 : 2 * Nc;      // y_{a+1}: rows -/+ of block a+1; y_Nc last
            Al[b_ * NPc + pair] = wide_col_entry<NX>(L, Nc, pair, b_);
        }
    }
    __syncthreads();
    WIDE_STAMP(3);
    // ---- P4: right-hand sides; P5: Cholesky of G in wave 0 beside it (after the optional dumps)
    auto fill_hv = [&](int first, int stride) {
        for (int row = first; row < 2 * NP; row += stride) {
            const int p = row >> 1, hh = row & 1;
            double hval;
            if (!(p & 1) && p < 2 * Nc) {
                hval = hh ? c.umax : -c.umin;
            } else {
                const int blk = p == 2 * Nc ? Nc : (p - 1) >> 1;
                double yv = ymeas;
                if (blk > 0) {
                    yv = c.h0;
#pragma unroll
                    for (int r = 0; r < NX; ++r) yv = fma(L.Ck[(blk - 1) * NX + r], L.xi[(blk - 1) * NX + r], yv);
                }
                hval = hh ? c.ymax - yv : yv - c.ymin;
            }
            L.hv[row] = hval;
        }
    };
    if (dump) {
        fill_hv(tid, WIDE_THREADS);
        __syncthreads();
        const int mt = 2 * NP;
        if (dump->Ge) for (int e = tid; e < n * n; e += WIDE_THREADS) {
            const int i = e / n, j = e % n;
            dump->Ge[e] = (i < Nc && j < Nc) ? (j <= i ? L.Lm[i * Nc + j] : 0.0) : (i == Nc && j == Nc ? c.rho : 0.0);
        }
        if (dump->Fe) for (int e = tid; e < n; e += WIDE_THREADS) dump->Fe[e] = e < Nc ? Al[e * NPc + NP] : 0.0;
        if (dump->Le) for (int e = tid; e < mt * n; e += WIDE_THREADS) {
            const int row = e / n, k = e % n;
            dump->Le[e] = k == Nc ? -1.0 : ((row & 1) ? 1.0 : -1.0) * Al[k * NPc + (row >> 1)];
        }
        if (dump->hv) for (int e = tid; e < mt; e += WIDE_THREADS) dump->hv[e] = L.hv[e];
        __syncthreads();
    }
    if (wave == 0) {
        const bool chol_ok = wide_chol_regs<R>(L.Lm, L.invdL, Nc);
        if (tid == 0) L.red[13] = chol_ok ? 1.0 : 0.0;
    } else if (!dump) {
        fill_hv(tid - 64, WIDE_THREADS - 64);
    }
    __syncthreads();
    if (L.red[13] == 0.0) {
        if (iters_out) *iters_out = 0;
        return 3;
    }
    WIDE_STAMP(4);
    // ---- rows of the (untransformed) columns in registers: lane 2 i + h holds rows k = 2 m + h of column i in
    //      areg[m] (pairs i < NP, then the linear term, which enters as right-hand side -q)
    const int ci = tid >> 1, h = tid & 1;
    const bool has_col = ci <= NP;
    const int cc = has_col ? ci : NP;
    double areg[KH];
    {
        const double sgn = ci == NP ? -1.0 : 1.0;
#pragma unroll
        for (int m = 0; m < KH; ++m) areg[m] = sgn * Al[(2 * m + h) * NPc + cc];
    }
    const int row = tid, pi = tid >> 1;
    const bool has_row = pi < NP;
    const double sigma = 1.0 / sqrt(c.rho), sig2 = 1.0 / c.rho;
    double s = 0.0, iscale = 1.0;
    auto pair_dot = [&](const double* op, int stride) {      // sum_k alpha_own[k] op[k stride], both lanes of the pair
        double p0 = 0.0, p1 = 0.0;
        const double* o = op + h * stride;
#pragma unroll
        for (int m = 0; m + 1 < KH; m += 2) {
            p0 = fma(areg[m], o[2 * m * stride], p0);
            p1 = fma(areg[m + 1], o[(2 * m + 2) * stride], p1);
        }
        if (KH & 1) p0 = fma(areg[KH - 1], o[2 * (KH - 1) * stride], p0);
        double part = p0 + p1;
        part += dpp_swap1(part);
        return part;
    };
    int q = 0, iters = 0, status = 0;
    double lamj = 0.0, wsgj = 1.0;      // lane j < q of EVERY wave: multiplier, pair and sign of working-set member j
    int wpj = 0;
    bool inW = false;
    const int max_iter = 10 * (n + 2 * NP) + 20;
    double* avec = L.wbuf + wave * 64;            // this wave's copies: direction, kp, l
    double* kpb = L.wbuf + (4 + wave) * 64;
    double* lb = L.wbuf + (8 + wave) * 64;
    double* Mi = L.Mi;
    // z from the multipliers: L' z = zt_unc + Nt_W lam; wave 0 solves with column k of L in lane k's registers,
    // the slack variable is separate (needs the factor of G in LDS and the transformed columns of the working set
    // and of the linear term in the pair table)
    auto recover_z = [&]() {
        if (wave == 0) {
            const int kr = lane < Nc ? lane : 0;
            double creg[R];
#pragma unroll
            for (int i = 0; i < R; ++i) creg[i] = L.Lm[i * Nc + kr];          // L[i][k]; rows >= Nc are zero
            const double* ak = Al + kr * NPc;
            double v = ak[NP];
            const double ls = lamj * wsgj;
#pragma unroll 4
            for (int j = 0; j < q; ++j) v = fma(ak[__builtin_amdgcn_readlane(wpj, j)], readlane_d(ls, j), v);
            if (lane >= Nc) v = 0.0;
            const double lsum = wave_sum(lane < q ? lamj : 0.0);
            const double invl = L.invdL[lane < R ? lane : 0];
            double xres = 0.0;
#pragma unroll
            for (int i = R - 1; i >= 0; --i) {
                const double xv = readlane_d(v, i) * readlane_d(invl, i);     // zero for the padding rows
                if (lane == i) xres = xv;
                v = lane < i ? fma(-creg[i], xv, v) : v;
            }
            if (lane < Nc) L.usol[lane] = xres;
            if (lane == 0) L.usol[Nc] = sig2 * lsum;
        }
    };
    // ---- Fast path.  The previous working set W of this controller is usually the optimal one: transform only its
    //      columns and the linear term, solve K lambda = -s_W for the multipliers (K = Gram matrix of W), recover z
    //      and CHECK it - every multiplier positive, every row satisfied to the tolerance of the iteration below.
    //      Then z is the optimum of the strictly convex QP and the transform of the other ~100 columns, their
    //      slacks and the iteration are not needed; otherwise the columns are put back and the full algorithm runs.
    //      Members whose multiplier comes out non-positive are dropped and the (small) system solved again, up to
    //      three times: a constraint that has just become inactive does not cost the full algorithm.
    bool fast_done = false;
    int fast_why = 3;                    // diagnostics: 0 fast path taken, 1 a multiplier not positive, 2 a row violated, 3 not tried
    auto fast_try = [&](auto fqc) -> bool {
        constexpr int FQ = decltype(fqc)::value, CW = (FQ + 1 + 3) / 4;
        const int q0 = warm->n_in, nfc = q0 + 1;
        const int kr = lane < Nc ? lane : 0;
        unsigned char* fl = reinterpret_cast<unsigned char*>(L.red + 48);   // the rows in play (one list for all waves)
        double* kb = L.wbuf;                                // K, then its factor: packed lower triangle
        double* rhs = L.wbuf + FQ * (FQ + 1) / 2;           // s_W in, multipliers out
        double ycol[CW], orig[CW];
        int colid[CW];
        if (wave == 0 && lane < q0) fl[lane] = warm->in[lane];
        __syncthreads();                                     // every areg load is done before a column changes
#pragma unroll
        for (int cw = 0; cw < CW; ++cw) {                    // wave w: members w, w + 4, ..; the linear term comes last
            const int j = wave + 4 * cw;
            const bool on = j < nfc;
            const int col = (on && j < q0) ? ((int)fl[j] >> 1) : NP;
            colid[cw] = on ? (col < NP ? col : NP) : -1;
            const double v = Al[kr * NPc + (colid[cw] >= 0 ? colid[cw] : NP)];
            orig[cw] = v;
            ycol[cw] = (on && lane < Nc) ? ((j < q0) ? v : -v) : 0.0;
        }
        __syncthreads();                                     // a column shared by two members is read before either writes
        {   // forward substitution, lane = row: w_k -= (L[k][l] / L[l][l]) w_l for k > l, y = w / diag at the end.
            // The lane's row of L is fetched up front so that no step waits for LDS.
            const double dreg = L.invdL[lane < R ? lane : 0];
            double lrow[R];
#pragma unroll
            for (int l = 0; l < R; ++l) lrow[l] = L.Lm[kr * Nc + (l < kr ? l : kr)];
#pragma unroll
            for (int l = 0; l < R; ++l) {
                if (l >= Nc) continue;                      // uniform
                const double coef = lane > l ? lrow[l] * readlane_d(dreg, l) : 0.0;
                double pv[CW];
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) pv[cw] = readlane_d(ycol[cw], l);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) ycol[cw] = fma(-coef, pv[cw], ycol[cw]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) ycol[cw] *= dreg;
        }
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
            if (lane < Nc && colid[cw] >= 0) Al[lane * NPc + colid[cw]] = ycol[cw];
        int qf = q0;
        bool ok = false;
        for (int round = 0; round < 3; ++round) {
            ++iters;
            __syncthreads();
            // Gram matrix and right-hand side s_W = h_W + sg (alpha_W . zt_unc) by the whole workgroup
            const int nk = qf * (qf + 1) / 2;
            for (int e = tid; e < nk + qf; e += WIDE_THREADS) {
                int a_ = 0, b_ = 0;
                if (e < nk) {
                    a_ = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                    while ((a_ + 1) * (a_ + 2) / 2 <= e) ++a_;
                    while (a_ * (a_ + 1) / 2 > e) --a_;
                    b_ = e - a_ * (a_ + 1) / 2;
                } else { a_ = e - nk; }
                const int ra = (int)fl[a_], rb = (int)fl[b_];
                const int wa = ra >> 1, wb = e < nk ? (rb >> 1) : NP;
                const double sga = (ra & 1) ? -1.0 : 1.0, sgb = (rb & 1) ? -1.0 : 1.0;
                double acc = 0.0;
#pragma unroll 4
                for (int k = 0; k < Nc; ++k) acc = fma(Al[k * NPc + wa], Al[k * NPc + wb], acc);
                if (e < nk) kb[e] = fma(sga * sgb, acc, sig2);
                else rhs[a_] = L.hv[ra] + sga * acc;
            }
            __syncthreads();
            if (wave == 0) {       // Cholesky of K with row i in lane i's registers, the two triangular solves
                const int lr = lane < qf ? lane : 0;
                double krow[FQ];
#pragma unroll
                for (int cx = 0; cx < FQ; ++cx) {
                    const double v = kb[lr * (lr + 1) / 2 + (cx <= lr ? cx : lr)];
                    krow[cx] = (lane < qf && cx <= lane) ? v : (cx == lane ? 1.0 : 0.0);
                }
                bool okc = true;
                wide_chol_steps<0, FQ>(krow, L.invdW, qf, okc);
                g_sync();
#pragma unroll
                for (int cx = 0; cx < FQ; ++cx)
                    if (lane < qf && cx <= lane) kb[lane * (lane + 1) / 2 + cx] = krow[cx];
                g_sync();
                const double invd = L.invdW[lr];
                double v = lane < qf ? -rhs[lr] : 0.0;       // forward: Lk y = -s_W
#pragma unroll
                for (int j = 0; j < FQ; ++j) {
                    if (j < qf) {
                        const double yj = readlane_d(v, j) * readlane_d(invd, j);
                        v = lane == j ? yj : (lane > j ? fma(-krow[j], yj, v) : v);
                    }
                }
                double ktr[FQ];                              // column `lane` of the factor: L[j][lane], j >= lane
#pragma unroll
                for (int j = 0; j < FQ; ++j) ktr[j] = (j < qf && lane <= j) ? kb[j * (j + 1) / 2 + lane] : 0.0;
#pragma unroll
                for (int j = FQ - 1; j >= 0; --j) {          // backward: Lk' lambda = y
                    if (j < qf) {
                        const double xj = readlane_d(v, j) * readlane_d(invd, j);
                        v = lane == j ? xj : (lane < j ? fma(-ktr[j], xj, v) : v);
                    }
                }
                const bool keep_me = lane < qf && v > 0.0;
                const unsigned long long keep = __ballot(keep_me);
                const int qn = __popcll(keep);
                int state = 0;                               // 0 solved, 1 members dropped: solve again, 2 give up
                if (!okc) state = 2;
                else if (qn < qf) {
                    state = 1;
                    const int rid = (int)fl[lr];
                    const int pos = __popcll(keep & ((1ull << lane) - 1ull));
                    g_sync();
                    if (keep_me) fl[pos] = (unsigned char)rid;
                } else if (lane < qf) rhs[lane] = v;
                if (lane == 0) { L.red[5] = (double)state; L.red[6] = (double)qn; }
            }
            __syncthreads();
            const int state = (int)L.red[5];
            qf = (int)L.red[6];
            if (state != 1) { ok = state == 0; break; }
        }
        fast_why = ok ? 0 : 1;
        WIDE_STAMP(5);
        if (ok) {                                            // uniform over the workgroup
            q = qf;
            const int rid = lane < q ? (int)fl[lane] : 0;
            wpj = rid >> 1; wsgj = (rid & 1) ? -1.0 : 1.0; lamj = lane < q ? rhs[lane] : 0.0;
            if (lane >= q) { wpj = 0; wsgj = 1.0; }
            recover_z();
            __syncthreads();
            // rows in original coordinates: slack = h + (-+) a'u + eps
            g_sync();
            avec[lane] = lane < Nc ? L.usol[lane] : 0.0;
            g_sync();
            const double part = pair_dot(avec, 1);
            bool viol = false;
            if (has_row) {
                const double hval = L.hv[row];
                const double sv = hval + (h ? -part : part) + L.usol[Nc];
                viol = sv * (1.0 / (1.0 + fabs(hval))) < -1e-10;
            }
            const bool any = __ballot(viol) != 0ull;
            if (lane == 0) L.red[wave] = any ? 1.0 : 0.0;
            __syncthreads();
            ok = (L.red[0] + L.red[1]) + (L.red[2] + L.red[3]) == 0.0;
            if (!ok) fast_why = 2;
            __syncthreads();                                 // red is reused below
        }
        if (!ok) {                                           // put the columns back; areg still holds the untransformed rows
#pragma unroll
            for (int cw = 0; cw < CW; ++cw)
                if (lane < Nc && colid[cw] >= 0) Al[lane * NPc + colid[cw]] = orig[cw];
            q = 0; lamj = 0.0; wpj = 0; wsgj = 1.0;
            __syncthreads();
        }
        return ok;
    };
    if (warm && warm->fast && warm->n_in < n && !dump) {
        if (warm->n_in <= 16) fast_done = fast_try(std::integral_constant<int, 16>{});
        else if (warm->n_in <= 32) fast_done = fast_try(std::integral_constant<int, 32>{});
    }
    if (!fast_done) {
        // ---- P6: alpha = L_G^-1 a for every pair and zt = -L_G^-1 F (column NP): right-looking forward
        //          substitution over the registers; row l's solution is formed by the lane that owns it, passed to
        //          its partner, and both update their later rows.
        {
            const double dreg = L.invdL[lane < R ? lane : 0];    // lane l: 1 / L[l][l]
            const double* Lh = L.Lm + h * Nc;                    // row 2 m + h of L starts at Lh + 2 m Nc
#pragma unroll
            for (int l = 0; l < R; ++l) {
                if (l >= Nc) continue;                          // uniform; padding rows have nothing to solve
                const int ml = l >> 1, hl = l & 1;
                const double ym = areg[ml] * readlane_d(dreg, l);
                const double yo = dpp_swap1(ym);
                const bool mine = h == hl;
                const double y = mine ? ym : yo;
                areg[ml] = mine ? y : areg[ml];
                if (mine && has_col && l < Nc) Al[l * NPc + cc] = y;
#pragma unroll
                for (int m = (l + 1) >> 1; m < KH; ++m) {
                    double coef = Lh[2 * m * Nc + l];
                    if (2 * m == l) coef = h ? coef : 0.0;       // row 2 m + 0 = l is the pivot row itself
                    areg[m] = fma(-coef, y, areg[m]);
                }
            }
        }
        __syncthreads();
        WIDE_STAMP(5);
        // ---- P7: slacks s = h + nt . zt_unc (the slack component of zt_unc is zero).  Lane 2 p + h owns row h of pair p.
        {
            const double part = pair_dot(Al + NP, NPc);
            if (has_row) {
                const double hval = L.hv[row];
                iscale = 1.0 / (1.0 + fabs(hval));
                s = hval + (h ? -part : part);
            }
        }
        WIDE_STAMP(6);
        // ---- P8: dual active-set iteration
        // The factor of G moves to registers for the duration of the iteration; its LDS becomes the working set's
        // Gram matrix and inverse factor.
        constexpr int SAVE_N = (R * R + WIDE_THREADS - 1) / WIDE_THREADS;
        const int save_len = R * Nc;
        double saved[SAVE_N];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SAVE_N; ++i) {
            const int e = tid + WIDE_THREADS * i;
            saved[i] = L.Lm[e < save_len ? e : 0];
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += WIDE_THREADS) Mi[e] = 0.0;
        // Gram matrix of the working set (whole workgroup), its Cholesky factor and M = Lw^-1 (wave 0); false if the
        // rows are numerically dependent.  Every wave holds the same q, wpj, wsgj.
        auto rebuild = [&]() -> bool {
            double* Wsg = reinterpret_cast<double*>(L.Wtab + 2 * ((n + 1) / 2));
            __syncthreads();               // every wave is done with the old inverse factor
            if (wave == 0 && lane < q) { L.Wtab[lane] = wpj; Wsg[lane] = wsgj; }
            __syncthreads();
            for (int e = tid; e < q * (q + 1) / 2; e += WIDE_THREADS) {
                int a_ = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while ((a_ + 1) * (a_ + 2) / 2 <= e) ++a_;
                while (a_ * (a_ + 1) / 2 > e) --a_;
                const int b_ = e - a_ * (a_ + 1) / 2;
                const int wa = L.Wtab[a_], wb = L.Wtab[b_];
                double acc = 0.0;
#pragma unroll 4
                for (int k = 0; k < Nc; ++k) acc = fma(Al[k * NPc + wa], Al[k * NPc + wb], acc);
                L.Lw[a_ * n + b_] = fma(Wsg[a_] * Wsg[b_], acc, sig2);
            }
            __syncthreads();
            if (wave == 0) {
                const bool ok = chol_rows_lds(L.Lw, q, n, L.invdW);
                g_sync();
                if (ok && lane < q) {       // column `lane` of M = Lw^-1 by forward substitution
                    for (int i = 0; i < q; ++i) {
                        double x = 0.0;
                        if (i == lane) x = L.invdW[i];
                        else if (i > lane) {
                            double acc = 0.0;
                            for (int k = lane; k < i; ++k) acc = fma(L.Lw[i * n + k], Mi[k * n + lane], acc);
                            x = -acc * L.invdW[i];
                        }
                        Mi[i * n + lane] = x;
                    }
                }
                if (lane == 0) L.red[13] = ok ? 1.0 : 0.0;
            }
            __syncthreads();
            return L.red[13] != 0.0;
        };
        // Warm start: the rows of the controller's previous working set enter together.  With K their Gram matrix,
        // the minimiser on their intersection has multipliers lambda = -K^-1 s_W (s = slacks at the unconstrained
        // minimiser); members whose multiplier is not positive are dropped and the rest solved again, until all
        // are positive - minimiser and set are then a valid state of the dual iteration, which goes on from
        // there.  A numerically dependent set falls back to the cold start.
        if (warm && warm->n_in > 0) {
            q = warm->n_in < n ? warm->n_in : n;
            {
                const int rid = lane < q ? (int)warm->in[lane] : 0;
                wpj = rid >> 1; wsgj = (rid & 1) ? -1.0 : 1.0;
                if (lane >= q || wpj >= NP) { wpj = 0; wsgj = 1.0; }
            }
            if (has_row) L.hv[row] = s;                          // right-hand sides are spent: hv now holds s
            while (q > 0) {
                ++iters;
                if (!rebuild()) { q = 0; break; }
                const int iq = lane < q ? lane : 0;
                g_sync();
                kpb[lane] = lane < q ? L.hv[2 * wpj + (wsgj < 0.0 ? 1 : 0)] : 0.0;
                g_sync();
                double lj = 0.0;
                {
                    const double* mr = Mi + iq * n;
#pragma unroll 4
                    for (int j = 0; j < q; ++j) lj = fma(mr[j], kpb[j], lj);
                }
                if (lane >= q) lj = 0.0;
                lb[lane] = lj;
                g_sync();
                double rj = 0.0;
                {
                    const double* mc = Mi + iq;
#pragma unroll 4
                    for (int i = 0; i < q; ++i) rj = fma(mc[i * n], lb[i], rj);
                }
                lamj = lane < q ? -rj : 0.0;
                const bool keep_me = lane < q && lamj > 0.0;
                const unsigned long long keep = __ballot(keep_me);
                const int qn = __popcll(keep);
                if (qn == q) break;
                const int pos = __popcll(keep & ((1ull << lane) - 1ull));
                g_sync();
                if (keep_me) kpb[pos] = (double)(2 * wpj + (wsgj < 0.0 ? 1 : 0));
                g_sync();
                q = qn;
                const int rid = lane < q ? (int)kpb[lane] : 0;
                wpj = rid >> 1; wsgj = (rid & 1) ? -1.0 : 1.0; lamj = 0.0;
            }
            if (q > 0) {
                const int kr = lane < Nc ? lane : 0;
                const double* ak = Al + kr * NPc;
                double a = 0.0;
                const double ls = lamj * wsgj;
#pragma unroll 4
                for (int j = 0; j < q; ++j) a = fma(ak[__builtin_amdgcn_readlane(wpj, j)], readlane_d(ls, j), a);
                const double lsum = wave_sum(lane < q ? lamj : 0.0);
                g_sync();
                avec[lane] = lane < Nc ? a : 0.0;
                g_sync();
                const double part = pair_dot(avec, 1);
                if (has_row) s += (h ? -part : part) + sig2 * lsum;
                for (int j = 0; j < q; ++j) {
                    const int rid = 2 * __builtin_amdgcn_readlane(wpj, j) + (readlane_d(wsgj, j) < 0.0 ? 1 : 0);
                    if (has_row && row == rid) { s = 0.0; inW = true; }
                }
            } else {
                lamj = 0.0; wpj = 0; wsgj = 1.0;
            }
        }
        WIDE_STAMP(15);
        for (;;) {
            double best = QP_INF; int bi = 0x7fffffff;
            if (has_row && !inW) { best = s * iscale; bi = row; }
            g_argmin<64>(best, bi);
            if (has_row && row == bi) L.red[4 + wave] = s;
            if (lane == 0) { L.red[wave] = best; L.red[8 + wave] = (double)bi; }
            __syncthreads();
            double gb = L.red[0], sp = L.red[4]; int p = (int)L.red[8];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const double v = L.red[w]; const int vi = (int)L.red[8 + w];
                if (v < gb || (v == gb && vi < p)) { gb = v; p = vi; sp = L.red[4 + w]; }
            }
            if (!(gb < -1e-10)) break;
            const int pp = p >> 1;
            const double sgp = (p & 1) ? -1.0 : 1.0;
            {   // Gram column of the candidate: Dp[i] = alpha_i . alpha_p
                const double part = pair_dot(Al + pp, NPc);
                if (has_row && h == 0) L.Dp[pi] = part;
            }
            __syncthreads();
            const double kpp = L.Dp[pp] + sig2;
            double lam_p = 0.0;
            for (;;) {
                if (++iters > max_iter) { status = 1; break; }
                const int iq = lane < q ? lane : 0;
                // l = Lw^-1 kp = M kp,  r = Lw^-T l = M' l   (M lower triangular, zero above the diagonal)
                g_sync();
                kpb[lane] = lane < q ? fma(wsgj * sgp, L.Dp[wpj], sig2) : 0.0;
                g_sync();
                double lj = 0.0;
                {
                    const double* mr = Mi + iq * n;
#pragma unroll 4
                    for (int j = 0; j < q; ++j) lj = fma(mr[j], kpb[j], lj);
                }
                if (lane >= q) lj = 0.0;
                const double dd = kpp - wave_sum(lj * lj);
                lb[lane] = lj;
                g_sync();
                double rj = 0.0;
                {
                    const double* mc = Mi + iq;
#pragma unroll 4
                    for (int i = 0; i < q; ++i) rj = fma(mc[i * n], lb[i], rj);
                }
                if (lane >= q) rj = 0.0;
                double t1 = QP_INF; int kd = 0x7fffffff;
                if (lane < q && rj > 1e-13) { t1 = lamj / rj; kd = lane; }
                g_argmin<64>(t1, kd);
                const bool primal = dd > 1e-12 * kpp && kpp > 0.0;
                const double t2 = primal ? -sp / dd : QP_INF;
                const double t = t1 < t2 ? t1 : t2;
                if (!(t < QP_INF)) { status = 2; break; }
                if (primal) {
                    // direction in zt coordinates, components k < Nc on lane k (every wave its own copy); the slack
                    // component is sigma (1 - sum r)
                    const int kr = lane < Nc ? lane : 0;
                    const double* ak = Al + kr * NPc;
                    double a = sgp * ak[pp];
                    const double rs = rj * wsgj;
#pragma unroll 4
                    for (int j = 0; j < q; ++j) a = fma(-ak[__builtin_amdgcn_readlane(wpj, j)], readlane_d(rs, j), a);
                    const double vn = sigma * (1.0 - wave_sum(rj));
                    avec[lane] = lane < Nc ? a : 0.0;
                    g_sync();
                    const double part = pair_dot(avec, 1);
                    if (has_row) s = fma(t, (h ? -part : part) + sigma * vn, s);
                }
                if (lane < q) lamj -= t * rj;
                lam_p += t;
                if (primal && t2 <= t1) {          // full step: p joins the working set at position q
                    const double dl = sqrt(dd), idl = 1.0 / dl;
                    if (lane < n) Mi[q * n + lane] = lane < q ? -rj * idl : (lane == q ? idl : 0.0);
                    if (lane == q) { lamj = lam_p; wpj = pp; wsgj = sgp; }
                    if (has_row && row == p) { s = 0.0; inW = true; }
                    ++q;
                    break;
                }
                // partial (or dual-only) step: the member at position kd leaves the working set
                {
                    const int dropped = 2 * __builtin_amdgcn_readlane(wpj, kd) + (readlane_d(wsgj, kd) < 0.0 ? 1 : 0);
                    if (has_row && row == dropped) inW = false;
                    const double lnext = g_next<64>(lamj), snext = g_next<64>(wsgj);
                    const int wnext = g_next_i<64>(wpj);
                    if (lane >= kd && lane < q - 1) { lamj = lnext; wpj = wnext; wsgj = snext; }
                    --q;
                    if (lane >= q) { lamj = 0.0; wpj = 0; wsgj = 1.0; }
                    // rebuild: K_ab = sg_a sg_b alpha_a . alpha_b + 1/rho by the whole workgroup, Lw = chol(K) and
                    // M = Lw^-1 by wave 0
                    if (has_row && row == p) L.red[12] = s;
                    if (!rebuild()) { status = 2; break; }
                    sp = L.red[12];
                }
            }
            if (status) break;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SAVE_N; ++i) {
            const int e = tid + WIDE_THREADS * i;
            if (e < save_len) L.Lm[e] = saved[i];
        }
        __syncthreads();
        WIDE_STAMP(7);
        recover_z();
    } else {
        WIDE_STAMP(6); WIDE_STAMP(15); WIDE_STAMP(7);
    }
    if (warm && warm->out && wave == 0) {
        const int qo = status == 0 ? q : 0;
        if (lane < qo) warm->out[lane] = (unsigned char)(2 * wpj + (wsgj < 0.0 ? 1 : 0));
        if (lane == 0) *warm->n_out = qo;
    }
    __syncthreads();
    WIDE_STAMP(8);
    // ---- P10: x_pred = Phi x0 + Gamma u + F0 (into xi)
    wide_predict<NX>(L, Nc, L.usol);
    WIDE_STAMP(9);
    if (st && tid == 0) { st[10] = (unsigned long long)iters | ((unsigned long long)fast_why << 32); st[11] = __builtin_amdgcn_s_memrealtime() - rt0; }
#undef WIDE_STAMP
    if (iters_out) *iters_out = iters;
    return status;
}

template <int NX, int KH>
__global__ __launch_bounds__(WIDE_THREADS) void mpc_qp_wide_kernel(MpcQpArgs a) {
    extern __shared__ __align__(16) double mpc_lds[];
    const int prob = blockIdx.x, tid = threadIdx.x;
    if (prob >= a.nb) return;
    const int Nc = a.c.Nc, n = Nc + 1, mt = 4 * Nc + 2;
    WideLds L;
    L.carve(mpc_lds, Nc, NX, KH);
    for (int e = tid; e < Nc * NX * NX; e += WIDE_THREADS) L.Ak[e] = a.lA[(size_t)prob * Nc * NX * NX + e];
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) {
        L.Bk[e] = a.lB[(size_t)prob * Nc * NX + e];
        L.Ck[e] = a.lC[(size_t)prob * Nc * NX + e];
        L.Xs[e] = a.Xs[(size_t)prob * Nc * NX + e];
    }
    for (int e = tid; e < Nc; e += WIDE_THREADS) L.Us[e] = a.Us[(size_t)prob * Nc + e];
    for (int e = tid; e < NX; e += WIDE_THREADS) L.x0[e] = a.x0[(size_t)prob * NX + e];
    __syncthreads();
    WideDump dump{a.Ge ? a.Ge + (size_t)prob * n * n : nullptr, a.Fe ? a.Fe + (size_t)prob * n : nullptr,
                  a.Le ? a.Le + (size_t)prob * mt * n : nullptr, a.hv ? a.hv + (size_t)prob * mt : nullptr};
    const bool want_dump = a.Ge || a.Fe || a.Le || a.hv;
    const int st = wide_solve<NX, KH>(a.c, L, a.ymeas ? a.ymeas[prob] : 0.0, nullptr, want_dump ? &dump : nullptr,
                                      a.stamps ? a.stamps + (size_t)prob * 16 : nullptr);
    if (a.u) for (int e = tid; e < Nc; e += WIDE_THREADS) a.u[(size_t)prob * Nc + e] = L.usol[e];
    if (a.xpred) for (int e = tid; e < Nc * NX; e += WIDE_THREADS) a.xpred[(size_t)prob * Nc * NX + e] = L.xi[e];
    if (tid == 0) {
        if (a.eps) a.eps[prob] = L.usol[Nc];
        if (a.status) a.status[prob] = st;
    }
}

// One LPV pass of controller b by the whole workgroup.  try_fast: first try the controller's carried working set as
// a whole (off when the first-stage kernel of mpc_wide_fast.h has already done so).
template <int NX, int KH>
__device__ __forceinline__ void wide_ctrl_pass(const CtrlDev& d, const RunPtrs& rp, int it, int k, int b, double* mpc_lds, bool try_fast) {
    const int tid = threadIdx.x;
    const int Nc = d.Nc, ntot = d.n_sim + Nc;
    unsigned long long* stp = d.stamps ? d.stamps + (size_t)b * 16 : nullptr;
    if (stp && tid == 0) stp[12] = __builtin_amdgcn_s_memtime();
    WideLds L;
    L.carve(mpc_lds, Nc, NX, KH);
    for (int e = tid; e < Nc * NX * NX; e += WIDE_THREADS) L.Ak[e] = (double)d.A32[(size_t)b * Nc * NX * NX + e];
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) {
        L.Bk[e] = (double)d.B32[(size_t)b * Nc * NX + e];
        L.Ck[e] = (double)d.C32[(size_t)b * Nc * NX + e];
        const int i = e / NX, r = e % NX;
        L.Xs[e] = rp.Xs[((size_t)b * ntot + (k + 1 + i)) * NX + r];
    }
    for (int e = tid; e < Nc; e += WIDE_THREADS) {
        L.Us[e] = rp.Us[(size_t)b * ntot + k + e];
        L.uold[e] = d.uplan[(size_t)b * Nc + e];
    }
    for (int e = tid; e < NX; e += WIDE_THREADS) L.x0[e] = d.x0n[(size_t)b * NX + e];
    // The working set of this controller's previous solve is the starting guess.  Later passes of a step solve
    // a QP with the same rows; the first pass of a step inherits the last step's set moved one stage forward:
    // input rows u_i -> u_{i-1}, output rows y_j -> y_{j-1} (the new y_0 is a measurement: not carried), and
    // rows of the last stage stay on the last stage as well (the plan's tail is repeated by the shift).
    WideWarm warm{nullptr, 0, nullptr, nullptr, 0};
    if (d.wset) {
        unsigned char* ws = d.wset + (size_t)b * 64;
        unsigned char* wl = reinterpret_cast<unsigned char*>(L.red + 16) + (tid >> 6) * 64;   // this wave's copy
        int cnt = d.wcnt[b];
        cnt = cnt < 0 ? 0 : (cnt > Nc + 1 ? Nc + 1 : cnt);
        const int lane = tid & 63;
        int r0 = lane < cnt ? (int)ws[lane] : -1, r1 = -1;
        if (it == 0 && r0 >= 0) {
            const int p = r0 >> 1, hbit = r0 & 1;
            int p0 = -1, p1 = -1;
            if (p == 2 * Nc) { p0 = 2 * Nc - 1; p1 = 2 * Nc; }                   // y_Nc -> y_{Nc-1}, y_Nc
            else if (p & 1) { p0 = p >= 5 ? p - 2 : -1; }                          // y_j -> y_{j-1}, j >= 2
            else { p0 = p >= 2 ? p - 2 : -1; if (p == 2 * Nc - 2) p1 = p; }      // u_i -> u_{i-1}; u_{Nc-1} twice
            if (Nc < 2) { p0 = -1; p1 = -1; }
            r0 = p0 >= 0 ? 2 * p0 + hbit : -1;
            r1 = p1 >= 0 ? 2 * p1 + hbit : -1;
        }
        const unsigned long long m0 = __ballot(r0 >= 0), m1 = __ballot(r1 >= 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int c0 = __popcll(m0), pos0 = __popcll(m0 & below), pos1 = c0 + __popcll(m1 & below);
        if (r0 >= 0 && pos0 < 64) wl[pos0] = (unsigned char)r0;
        if (r1 >= 0 && pos1 < 64) wl[pos1] = (unsigned char)r1;
        int tot = c0 + __popcll(m1);
        tot = tot > Nc + 1 ? Nc + 1 : tot;
        warm.in = wl; warm.n_in = d.warm ? tot : 0; warm.out = ws; warm.n_out = d.wcnt + b; warm.fast = d.warm >= 2 && try_fast;
    }
    __syncthreads();
    int qp_it = 0;
    const int st = wide_solve<NX, KH>(d.c, L, d.ymeas[b], &qp_it, nullptr, stp, &warm);
    if (st != 0) {          // workgroup-uniform: keep the plan the controller came with, flag it (see mpc_solve_body)
        if (tid == 0) {
            d.cnt[b] = it + 1;
            atomicAdd(d.qp_fail, 1ull);
            d.qp_stat[2 * b] += 1u;
            d.qp_stat[2 * b + 1] += (unsigned)qp_it;
            if (st > d.qp_worst[b]) d.qp_worst[b] = st;
        }
        return;
    }
    double du = 0.0;
    for (int e = tid; e < Nc; e += WIDE_THREADS) {
        const double un = L.usol[e], dd = un - L.uold[e];
        du += dd * dd;
        d.uplan[(size_t)b * Nc + e] = un;
        d.us32[(size_t)b * Nc + e] = (float)un;
    }
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) {
        const double xv = L.xi[e];
        d.xpred[(size_t)b * Nc * NX + e] = xv;
        d.xc32[(size_t)b * Nc * NX + e] = (float)xv;
        d.xs32[(size_t)b * Nc * NX + e] = e < NX ? (float)L.x0[e] : (float)L.xi[e - NX];
    }
    if (tid < 64) {                                 // Nc <= 63: the plan's entries all sit in wave 0
        du = wave_sum(du);
        if (tid == 0) {
            d.eps[b] = L.usol[Nc];
            d.cnt[b] = it + 1;
            d.qp_stat[2 * b] += 1u;
            d.qp_stat[2 * b + 1] += (unsigned)qp_it;
            if (it + 1 < d.maxit && !(sqrt(du) < d.tol)) {
                const int pos = atomicAdd(&d.nact[it + 1], 1);
                d.act[(size_t)(it + 1) * d.B + pos] = b;
            }
        }
    }
    if (stp && tid == 0) { stp[13] = __builtin_amdgcn_s_memtime(); stp[14] = (unsigned long long)it; }
}

// slow_mode 0: the controllers of pass `it` (all of them in pass 0, the work list afterwards), one per workgroup.
// slow_mode 1: the controllers the first-stage kernel left undecided (slow[it][..], nslow[it]); a small grid
// walks the list, because a workgroup of this kernel needs most of a CU's LDS even if it only comes to exit.
template <int NX, int KH>
__global__ __launch_bounds__(WIDE_THREADS) void mpc_solve_wide_kernel(CtrlDev d, RunPtrs rp, int it, int k, int slow_mode) {
    extern __shared__ __align__(16) double mpc_lds[];
    const int count = slow_mode ? min(d.nslow[it], d.B) : (it == 0 ? d.B : min(d.nact[it], d.B));
    for (int gid = blockIdx.x; gid < count; gid += gridDim.x) {
        const int b = slow_mode ? d.slow[(size_t)it * d.B + gid] : (it == 0 ? gid : d.act[(size_t)it * d.B + gid]);
        wide_ctrl_pass<NX, KH>(d, rp, it, k, b, mpc_lds, slow_mode == 0);
        __syncthreads();                             // the LDS is reused by the next controller
    }
}

// host side (mpc_wide.hip, mpc_wide34.hip)
bool mpc_wide_applies(int Nc, int nx, int mode, int slack);
int launch_mpc_qp_wide(const MpcQpArgs& a, hipStream_t s);
int launch_mpc_solve_wide(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s);
int launch_mpc_qp_wide_nx34(const MpcQpArgs& a, hipStream_t s);
int launch_mpc_solve_wide_nx34(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s);

template <int NX, int KH>
static int launch_qp_wide_t(const MpcQpArgs& a, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_qp_wide_kernel<NX, KH>)) return 1;
        mark_used_on_device(once);
    }
    const size_t lds = sizeof(double) * (size_t)WideLds::doubles(a.c.Nc, NX, KH);
    hipLaunchKernelGGL((mpc_qp_wide_kernel<NX, KH>), dim3(a.nb), dim3(WIDE_THREADS), lds, s, a);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int NX, int KH>
static int launch_solve_wide_t(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_solve_wide_kernel<NX, KH>)) return 1;
        mark_used_on_device(once);
    }
    const size_t lds = sizeof(double) * (size_t)WideLds::doubles(d.Nc, NX, KH);
    const int slow_mode = (d.warm >= 2 && d.slow) ? 1 : 0;       // the first-stage kernel has run: only its leftovers
    const int grid = slow_mode ? (d.B < WIDE_SLOW_GRID ? d.B : WIDE_SLOW_GRID) : d.B;
    hipLaunchKernelGGL((mpc_solve_wide_kernel<NX, KH>), dim3(grid), dim3(WIDE_THREADS), lds, s, d, rp, it, k, slow_mode);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nmpc
