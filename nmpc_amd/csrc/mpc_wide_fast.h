// Long-horizon condensing + QP, first stage (reference: NMPC_test.py:206-238 with Nc of 24..63).
//
// In the closed loop nine solves out of ten end on the working set the controller's previous solve ended on (or on
// none at all).  Those need no active-set iteration: condense, factor the Hessian, transform the normals of the
// carried working set W and the linear term, solve the small system K lambda = -s_W, recover z and CHECK it -
// every multiplier positive, every row satisfied to the iteration's tolerance.  The point then is the optimum of
// the strictly convex QP.  This kernel does exactly that and nothing else, which lets it live in a third of a CU's
// LDS (Lm, the prefix products and a table of at most FAST_Q + 1 transformed columns; the untransformed normals
// are recomputed from the prefix products where they are needed), so three controllers - or one and two embedding
// workgroups - share a CU.  A controller whose check fails, or whose carried set is larger than FAST_Q, is put on
// the step's slow list; `mpc_solve_wide_kernel` (mpc_wide.h: full pair table, dual active-set iteration, 100 KB of
// LDS) then solves those from scratch.  The arithmetic of a solve that succeeds here is that of the fast path the
// full kernel used to try first.
#pragma once
#include "mpc_wide.h"

namespace nmpc {

constexpr int FAST_Q = 32;          // largest carried working set taken here
constexpr int FAST_ADDS = 6;        // violated rows the first-stage kernel adds to the carried set before it gives up

struct FastLds {
    double *Lm, *invdL, *Pm, *Ak, *Bk, *Ck, *qd, *xi, *Xs, *x0, *Us, *uold, *hv, *Fv, *usol, *red, *invdW, *Wc, *kb, *avec;
    // Wc: transformed columns, member slot j at Wc[j Nc ..], the linear term in slot FAST_Q; before the transform the
    // region is wide_hessian's scratch
    __host__ __device__ static int wc_doubles(int Nc, int nx) {
        const int a = (FAST_Q + 1) * Nc, b = wide_hessian_scratch(Nc, nx);
        return a > b ? a : b;
    }
    __host__ __device__ static int doubles(int Nc, int nx, int KH) {
        const int R = 2 * KH, n = Nc + 1;
        return R * Nc + R + 2 * Nc * nx * nx + 5 * Nc * nx + 4 + 2 * Nc + 2 * wide_npairs(Nc) + 2 + Nc + (n + 1) + 64 + FAST_Q +
               wc_doubles(Nc, nx) + FAST_Q * (FAST_Q + 1) / 2 + FAST_Q + 4 * 64;
    }
    __device__ void carve(double* p, int Nc, int nx, int KH) {
        const int R = 2 * KH, n = Nc + 1;
        Lm = p; p += R * Nc; invdL = p; p += R; Pm = p; p += Nc * nx * nx; Ak = p; p += Nc * nx * nx;
        Bk = p; p += Nc * nx; Ck = p; p += Nc * nx; qd = p; p += Nc * nx; xi = p; p += Nc * nx; Xs = p; p += Nc * nx;
        x0 = p; p += 4; Us = p; p += Nc; uold = p; p += Nc; hv = p; p += 2 * wide_npairs(Nc) + 2; Fv = p; p += Nc;
        usol = p; p += n + 1; red = p; p += 64; invdW = p; p += FAST_Q; Wc = p; p += wc_doubles(Nc, nx);
        kb = p; p += FAST_Q * (FAST_Q + 1) / 2 + FAST_Q; avec = p;
    }
};

// Returns 0: solved (usol[0..Nc) = u, usol[Nc] = eps, the final working set in `out_rows` / *out_n), -1: not decided
// here (the caller hands the controller to the full kernel), 3: Hessian not positive definite.
// in_rows / n_in: the carried working set (row ids 2 pair + sign bit, n_in <= FQ), in LDS.
// red: [5] state, [6] count, [13] Cholesky flag, [0..3] per-wave violation flags; bytes of red[48..55] the rows in play,
// of red[56..63] their column slots.
template <int NX, int KH, int FQ>
__device__ __forceinline__ int wide_fast_solve(const MpcConst& c, FastLds& L, double ymeas, const unsigned char* in_rows, int n_in,
                                               unsigned char* out_rows, int* out_n, int* iters_out,
                                               unsigned long long* st = nullptr) {
    constexpr int R = 2 * KH, CW = (FQ + 1 + 3) / 4;
#define FAST_STAMP(i) do { if (st && threadIdx.x == 0) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
    FAST_STAMP(0);
    static_assert(FQ <= FAST_Q, "column table too small");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nc = c.Nc, NP = wide_npairs(Nc);
    // ---- prefix products and free response (wave 0); padding rows of Lm and of its diagonal (other waves)
    if (wave == 0) {
        wide_prefix<NX>(c, L, Nc);
    } else {
        const int t3 = tid - 64, n3 = WIDE_THREADS - 64;
        for (int e = Nc * Nc + t3; e < R * Nc; e += n3) L.Lm[e] = 0.0;
        for (int e = Nc + t3; e < R; e += n3) L.invdL[e] = 0.0;
    }
    __syncthreads();
    FAST_STAMP(1);
    for (int e = tid; e < Nc * NX; e += WIDE_THREADS) {          // qd_i = Q (xi_i - Xs_i)
        const int i = e / NX, r = e % NX;
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) a = fma(c.Q[r * NX + l], L.xi[i * NX + l] - L.Xs[i * NX + l], a);
        L.qd[e] = a;
    }
    __syncthreads();
    FAST_STAMP(2);
    wide_hessian<NX>(c, L, Nc, L.Wc, [&](int a_, double fv) { L.Fv[a_] = fv; });
    __syncthreads();
    FAST_STAMP(3);
    // ---- right-hand sides (waves 1..3) beside the Cholesky factorisation of G (wave 0)
    if (wave == 0) {
        const bool chol_ok = wide_chol_regs<R>(L.Lm, L.invdL, Nc);
        if (tid == 0) L.red[13] = chol_ok ? 1.0 : 0.0;
    } else {
        for (int row = tid - 64; row < 2 * NP; row += WIDE_THREADS - 64) {
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
    }
    __syncthreads();
    if (L.red[13] == 0.0) { *iters_out = 0; return 3; }
    FAST_STAMP(4);
    const int pi = tid >> 1, h = tid & 1;
    const bool has_row = pi < NP;
    const double sig2 = 1.0 / c.rho;
    const int kr = lane < Nc ? lane : 0;
    unsigned char* fl = reinterpret_cast<unsigned char*>(L.red + 48);
    unsigned char* sl = reinterpret_cast<unsigned char*>(L.red + 56);
    double* kb = L.kb;
    double* rhs = L.kb + FAST_Q * (FAST_Q + 1) / 2;
    const int q0 = n_in;
    if (wave == 0 && lane < q0) { fl[lane] = in_rows[lane]; sl[lane] = (unsigned char)lane; }
    __syncthreads();
    // Forward substitution y = L_G^-1 w, lane = row, for NC columns held one per register: w_k -= (L[k][l] / L[l][l]) w_l
    // for k > l, y = w / diag at the end.  The lane's row of L comes from LDS eight entries at a time, one chunk ahead
    // of its use.
    const double dreg = L.invdL[lane < R ? lane : 0];
    auto forward = [&](auto ncc, double* ycol) {
        constexpr int NC = decltype(ncc)::value;
        const double* lr = L.Lm + kr * Nc;
        double nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nxt[u] = lr[u < kr ? u : kr];
#pragma unroll 1
        for (int l0 = 0; l0 < R; l0 += 8) {
            double cur[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
            if (l0 + 8 < R) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int l = l0 + 8 + u; nxt[u] = lr[l < kr ? l : kr]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int l = l0 + u;
                if (l >= R || l >= Nc) continue;            // uniform
                const double coef = lane > l ? cur[u] * readlane_d(dreg, l) : 0.0;
                double pv[NC];
#pragma unroll
                for (int cw = 0; cw < NC; ++cw) pv[cw] = readlane_d(ycol[cw], l);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cw = 0; cw < NC; ++cw) ycol[cw] = fma(-coef, pv[cw], ycol[cw]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int cw = 0; cw < NC; ++cw) ycol[cw] *= dreg;
    };
    // ---- transform: y = L_G^-1 a for the members of W (wave w: members w, w + 4, ..) and zt_unc = -L_G^-1 F
    {
        double ycol[CW];
        int slot[CW];
#pragma unroll
        for (int cw = 0; cw < CW; ++cw) {
            const int j = wave + 4 * cw;
            const bool on = j <= q0, lin = j == q0;
            slot[cw] = on ? (lin ? FAST_Q : j) : -1;
            double v = 0.0;
            if (on && !lin) v = wide_col_entry<NX>(L, Nc, (int)fl[j] >> 1, kr);
            if (lin) v = -L.Fv[kr];
            ycol[cw] = (on && lane < Nc) ? v : 0.0;
        }
        const int ncol = (q0 + 1 + 3) / 4;                       // column slots in use per wave (uniform)
        if (ncol <= 1) forward(std::integral_constant<int, 1>{}, ycol);
        else if (ncol <= 2) forward(std::integral_constant<int, 2>{}, ycol);
        else if (ncol <= 3 || CW <= 3) forward(std::integral_constant<int, CW < 3 ? CW : 3>{}, ycol);
        else if (ncol <= 5 || CW <= 5) forward(std::integral_constant<int, CW < 5 ? CW : 5>{}, ycol);
        else forward(std::integral_constant<int, CW>{}, ycol);
#pragma unroll
        for (int cw = 0; cw < CW; ++cw)
            if (lane < Nc && slot[cw] >= 0) L.Wc[slot[cw] * Nc + lane] = ycol[cw];
    }
    FAST_STAMP(5);
    int qf = q0, iters = 0, next_slot = q0;
    // Attempts: solve on W; members with non-positive multipliers leave (up to three rounds); then z is recovered and
    // every row checked.  A violated row joins W - the most violated one, as in the dual iteration - and W is solved
    // again, up to FAST_ADDS times.  Whatever route W took, the final check is the certificate: multipliers positive,
    // rows satisfied, so z is the optimum; without it the controller goes to the full kernel.
    for (int attempt = 0;; ++attempt) {
        bool ok = false;
        for (int round = 0; round < 3; ++round) {
            ++iters;
            __syncthreads();
            // Gram matrix of W and right-hand side s_W = h_W + sg (alpha_W . zt_unc) by the whole workgroup
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
                const double* wa = L.Wc + (int)sl[a_] * Nc;
                const double* wb = L.Wc + (e < nk ? (int)sl[b_] : FAST_Q) * Nc;
                const double sga = (ra & 1) ? -1.0 : 1.0, sgb = (rb & 1) ? -1.0 : 1.0;
                double acc = 0.0;
#pragma unroll 4
                for (int k2 = 0; k2 < Nc; ++k2) acc = fma(wa[k2], wb[k2], acc);
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
                    const int rid = (int)fl[lr], sid = (int)sl[lr];
                    const int pos = __popcll(keep & ((1ull << lane) - 1ull));
                    g_sync();
                    if (keep_me) { fl[pos] = (unsigned char)rid; sl[pos] = (unsigned char)sid; }
                } else if (lane < qf) rhs[lane] = v;
                if (lane == 0) { L.red[5] = (double)state; L.red[6] = (double)qn; }
            }
            __syncthreads();
            const int state = (int)L.red[5];
            qf = (int)L.red[6];
            if (state != 1) { ok = state == 0; break; }
        }
        *iters_out = iters;
        if (attempt == 0) FAST_STAMP(6);
        if (!ok) return -1;                                      // uniform over the workgroup
        // ---- z from the multipliers: L' z = zt_unc + Nt_W lam (wave 0: lane k solves for z_k, column k of L from LDS
        //      eight entries at a time, one chunk ahead)
        const int q = qf;
        const int rid = lane < q ? (int)fl[lane] : 0;
        if (wave == 0) {
            const int wsl = lane < q ? (int)sl[lane] : 0;
            const double wsgj = (lane < q && (rid & 1)) ? -1.0 : 1.0;
            const double lamj = lane < q ? rhs[lane] : 0.0;
            double v = L.Wc[FAST_Q * Nc + kr];
            const double ls = lamj * wsgj;
#pragma unroll 4
            for (int j = 0; j < q; ++j) v = fma(L.Wc[__builtin_amdgcn_readlane(wsl, j) * Nc + kr], readlane_d(ls, j), v);
            if (lane >= Nc) v = 0.0;
            const double lsum = wave_sum(lane < q ? lamj : 0.0);
            double xres = 0.0;
            constexpr int RC = (R + 7) / 8 * 8;
            const double* lc = L.Lm + kr;                     // L[i][k] at lc[i Nc]; rows >= Nc are zero
            double nxt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = RC - 1 - u; nxt[u] = i < R ? lc[i * Nc] : 0.0; }
#pragma unroll 1
            for (int i0 = RC - 1; i0 >= 0; i0 -= 8) {
                double cur[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
                if (i0 - 8 >= 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) nxt[u] = lc[(i0 - 8 - u) * Nc];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 - u;
                    if (i >= R) continue;
                    const double xv = readlane_d(v, i) * readlane_d(dreg, i);     // zero for the padding rows
                    if (lane == i) xres = xv;
                    v = lane < i ? fma(-cur[u], xv, v) : v;
                }
            }
            if (lane < Nc) L.usol[lane] = xres;
            if (lane == 0) L.usol[Nc] = sig2 * lsum;
        }
        __syncthreads();
        if (attempt == 0) FAST_STAMP(15);
        // ---- the check, in original coordinates.  Input rows: -u_i - eps <= -umin, u_i - eps <= umax.  Output rows:
        //      the predicted outputs y_{i+1} = C_i x_{i+1} + h0 against ymin - eps, ymax + eps (x = Phi x0 + Gamma u + F0,
        //      formed in `xp`; the measured y_0 does not depend on u).  Row 2 p + hh of pair p has the slack
        //      s = h + (-+) a'u + eps, scaled by 1 / (1 + |h|) as in the iteration.
        wide_predict_into<NX>(L, Nc, L.usol, L.Xs);              // Xs has served its purpose (qd): the prediction's buffer
        if (attempt == 0) FAST_STAMP(7);
        double score = QP_INF; int bi = 0x7fffffff;
        if (has_row) {
            const int hh = h;
            double lhs;                                          // a'u of this row's pair
            if (!(pi & 1) && pi < 2 * Nc) lhs = L.usol[pi >> 1];
            else if (pi == 1) lhs = 0.0;
            else {
                const int a_ = pi == 2 * Nc ? Nc - 1 : (pi - 3) / 2;
                double yv = 0.0, y0v = 0.0;                      // C_a (x_{a+1} - xi_a) = C_a Gamma_{a,:} u
#pragma unroll
                for (int r = 0; r < NX; ++r) { yv = fma(L.Ck[a_ * NX + r], L.Xs[a_ * NX + r], yv); y0v = fma(L.Ck[a_ * NX + r], L.xi[a_ * NX + r], y0v); }
                lhs = yv - y0v;
            }
            const double hval = L.hv[tid];
            const double sv = hval + (hh ? -lhs : lhs) + L.usol[Nc];
            const double sc = sv * (1.0 / (1.0 + fabs(hval)));
            if (sc < -1e-10) { score = sc; bi = tid; }
        }
        g_argmin<64>(score, bi);
        if (lane == 0) { L.red[wave] = score; L.red[8 + wave] = (double)bi; }
        __syncthreads();
        double gb = L.red[0]; int p = (int)L.red[8];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const double v = L.red[w]; const int vi = (int)L.red[8 + w];
            if (v < gb || (v == gb && vi < p)) { gb = v; p = vi; }
        }
        if (!(gb < -1e-10)) {                                    // no row violated: z is the optimum
            for (int e = tid; e < Nc * NX; e += WIDE_THREADS) L.xi[e] = L.Xs[e];
            __syncthreads();
            FAST_STAMP(8);
            if (st && threadIdx.x == 0) st[10] = (unsigned long long)iters;
            if (out_rows && wave == 0) {
                if (lane < q) out_rows[lane] = (unsigned char)rid;
                if (lane == 0) *out_n = q;
            }
            return 0;
        }
        if (attempt >= FAST_ADDS || qf >= FQ || next_slot >= FQ) return -1;
        // ---- row p joins W: its transformed column into the next free slot (wave 0; the others wait at the barrier
        //      that opens the next attempt)
        if (wave == 0) {
            double y1[1];
            y1[0] = lane < Nc ? wide_col_entry<NX>(L, Nc, p >> 1, kr) : 0.0;
            forward(std::integral_constant<int, 1>{}, y1);
            if (lane < Nc) L.Wc[next_slot * Nc + lane] = y1[0];
            if (lane == 0) { fl[qf] = (unsigned char)p; sl[qf] = (unsigned char)next_slot; }
        }
        ++qf; ++next_slot;
    }
#undef FAST_STAMP
}

// Work lists of the long-horizon solve: the controllers of pass `it` come from `act` (all of them in pass 0), those
// this kernel does not decide go to slow[it][..] / nslow[it].
template <int NX, int KH>
__global__ __launch_bounds__(WIDE_THREADS, 2) void mpc_solve_fast_kernel(CtrlDev d, RunPtrs rp, int it, int k) {
    extern __shared__ __align__(16) double mpc_lds[];
    const int gid = blockIdx.x, tid = threadIdx.x;
    const int count = it == 0 ? d.B : min(d.nact[it], d.B);
    if (gid >= count) return;
    const int b = it == 0 ? gid : d.act[(size_t)it * d.B + gid];
    const int Nc = d.Nc, ntot = d.n_sim + Nc;
    unsigned long long* stp = d.stamps ? d.stamps + (size_t)b * 16 : nullptr;
    if (stp && tid == 0) { stp[12] = __builtin_amdgcn_s_memtime(); stp[11] = __builtin_amdgcn_s_memrealtime(); }
    FastLds L;
    L.carve(mpc_lds, Nc, NX, KH);
    auto to_slow = [&]() {
        if (tid == 0) {
            const int pos = atomicAdd(&d.nslow[it], 1);
            d.slow[(size_t)it * d.B + pos] = b;
            atomicAdd(d.qp_fail + 1, 1ull);              // tally: solves handed to the full kernel
        }
    };
    // the carried working set, moved one stage forward at the first pass of a step (as in mpc_solve_wide_kernel)
    unsigned char* wl = reinterpret_cast<unsigned char*>(L.avec);            // 64 bytes; avec is not in use yet
    int tot = 0;
    {
        unsigned char* ws = d.wset + (size_t)b * 64;
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
        if (tid < 64) {
            if (r0 >= 0 && pos0 < 64) wl[pos0] = (unsigned char)r0;
            if (r1 >= 0 && pos1 < 64) wl[pos1] = (unsigned char)r1;
        }
        tot = c0 + __popcll(m1);
        tot = tot > Nc + 1 ? Nc + 1 : tot;
    }
    if (tot > FAST_Q) { to_slow(); return; }                       // uniform (every wave computes the same count)
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
    __syncthreads();
    // the rows move out of avec (which the solve uses) into the byte list the solve reads them from
    unsigned char* inr = reinterpret_cast<unsigned char*>(L.red + 40);        // red[40..47]: 64 bytes
    if (tid < 64 && tid < tot) inr[tid] = wl[tid];
    __syncthreads();
    int qp_it = 0;
    int st;
    if (tot <= 16 - FAST_ADDS) st = wide_fast_solve<NX, KH, 16>(d.c, L, d.ymeas[b], inr, tot, d.wset + (size_t)b * 64, d.wcnt + b, &qp_it, stp);
    else st = wide_fast_solve<NX, KH, FAST_Q>(d.c, L, d.ymeas[b], inr, tot, d.wset + (size_t)b * 64, d.wcnt + b, &qp_it, stp);
    if (st == -1) { to_slow(); return; }
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
    if (stp && tid == 0) stp[9] = __builtin_amdgcn_s_memtime();     // (the prediction is part of the solve's check)
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
    if (stp && tid == 0) { stp[13] = __builtin_amdgcn_s_memtime(); stp[14] = (unsigned long long)it; stp[11] = __builtin_amdgcn_s_memrealtime() - stp[11]; }
}

template <int NX, int KH>
static int launch_solve_fast_t(const CtrlDev& d, const RunPtrs& rp, int it, int k, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_solve_fast_kernel<NX, KH>)) return 1;
        mark_used_on_device(once);
    }
    const size_t lds = sizeof(double) * (size_t)FastLds::doubles(d.Nc, NX, KH);
    hipLaunchKernelGGL((mpc_solve_fast_kernel<NX, KH>), dim3(d.B), dim3(WIDE_THREADS), lds, s, d, rp, it, k);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nmpc
