// Condensing + QP kernels of the LPV-NMPC step (reference: NMPC_test.py:149-260, mpcUtil.py:84-134) as
// templates, shared by the translation units that instantiate them per state dimension
// (mpc_nx2.hip, mpc_nx3.hip, mpc_nx4.hip) and by controller.hip (host side, plant kernels).
//
// `mpc_solve_kernel`: one lane group per controller
//   - runs the condensing recurrences Gamma_{j+d,j} = A_d Gamma_{j+d-1,j} and the free response
//     Phi x0 + F0 per lane in registers,
//   - assembles G, F, the constraint rows [M Z Gamma + E, -1] and h directly in the solver's LDS
//     layout (Phi, Gamma, L never touch HBM),
//   - solves the QP (qp_gi.h), updates the predicted trajectory and tests ||u - u_old|| < tol,
//   - appends itself to the next pass's work list if it has not converged.
// `mpc_qp_kernel`: the same assembly + solve for the stand-alone entry point (NMPC_test.py:206-238).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"
#include "qp_gi.h"

namespace nmpc {

// mode 0: output feedback (rows -/+u_i, -/+y_i through C_k: getDEMc_out, mpcUtil.py:340-362);
// mode 1: state feedback (rows -/+u_i, -/+x_i on the measured and predicted states: getDEMc,
// mpcUtil.py:112-134).  slack = 1: one extra variable with weight rho and coefficient -1 on every row.
struct MpcConst {
    int Nc, mode, slack;
    int as_written;                     // mode 1 only: the linear term adds Psi Us (NMPC.py:591) instead of subtracting it
    double Q[64];                       // nx * nx, row-major (nx <= 8)
    double R, rho, umin, umax, ymin, ymax, h0;
    double f0[8];
    double xmin[8], xmax[8];
};
__host__ __device__ inline int mpc_nvar(int Nc, int slack) { return Nc + (slack ? 1 : 0); }
__host__ __device__ inline int mpc_nrows(int Nc, int nx, int mode) { return mode ? 2 * (nx + 1) * Nc + 2 * nx : 4 * Nc + 2; }

// LDS carve for one controller's condensing + QP
struct MpcLds {
    QPWork qp;
    double *Ak, *Bk, *Ck, *qd, *xi, *Gam, *x0, *Xs, *Us, *uold;
    // Gam holds the block lower triangle of Gamma, packed: block (i, j <= i) at ((i (i+1))/2 + j) * nx
    __device__ static int tri(int i, int j) { return i * (i + 1) / 2 + j; }
    __device__ void carve(double* base, int Nc, int nx, int n, int mt) {
        qp.carve(base, n, mt);
        double* p = base + qp_lds_doubles(n, mt);
        Ak = p; p += Nc * nx * nx; Bk = p; p += Nc * nx; Ck = p; p += Nc * nx;
        qd = p; p += Nc * nx; xi = p; p += Nc * nx;
        Gam = p; p += Nc * (Nc + 1) / 2 * nx; x0 = p; p += nx; Xs = p; p += Nc * nx; Us = p; p += Nc; uold = p;
    }
};
__host__ __device__ inline int mpc_lds_doubles(int Nc, int nx, int n, int mt) {
    return qp_lds_doubles(n, mt) + Nc * nx * nx + 5 * Nc * nx + Nc * (Nc + 1) / 2 * nx + nx + 2 * Nc + 2;
}

// Fills qp.Lm (Hessian, lower triangle), qp.xt (linear term), qp.Nt / qp.hv (constraint rows) from
// Ak, Bk, Ck, x0, Xs, Us.  Leaves xi and Gam for mpc_predict.
//
// Condensing runs as independent recurrences, one per lane and entirely in registers (the only
// LDS traffic on the critical path is the broadcast read of A_d):
//   lane j < Nc : column j of Gamma,  Gamma_{j,j} = B_j,  Gamma_{j+d,j} = A_d Gamma_{j+d-1,j}
//                 (P_d B_j with the prefix products P_d = A_d P_{d-1}, P_0 = I of mpcUtil.py:98-117)
//   lane Nc     : free response xi_i = P_i A_0 x0 + (sum_{j<=i} P_j) f0   (mpcUtil.py:84-96,119-134)
// The Hessian and the output-constraint rows are then one work item per lower-triangular pair
// (a, b): G_ab = 2 sum_i Gamma_{i,a}' Q Gamma_{i,b} (+ 2R on the diagonal), the diagonal item also
// carries F_a, and the same item writes the two y-rows' coefficient C_a Gamma_{a,b}.
template <int GS, int NX, int NC = 0>
__device__ __forceinline__ void mpc_assemble(const MpcConst& c, MpcLds& L, double ymeas, int gl, unsigned long long* st = nullptr) {
    // NC > 0: horizon known at compile time (output feedback with slack only)
    const int Nc = NC ? NC : c.Nc, n = NC ? NC + 1 : mpc_nvar(Nc, c.slack), mt = NC ? 4 * NC + 2 : mpc_nrows(Nc, NX, c.mode);
    const bool state_rows = NC ? false : c.mode != 0;
    double* Lm = L.qp.Lm; double* Nt = L.qp.Nt; double* qd = L.qd;      // Q (xi_i - Xs_i)
    if (st && gl == 0) st[8] = __builtin_amdgcn_s_memtime();
    for (int e = gl; e < n * n; e += GS) Lm[e] = (n > Nc && e == n * n - 1) ? c.rho : 0.0;
    for (int e = gl; e < n * mt; e += GS) Nt[e] = e >= Nc * mt ? -1.0 : 0.0;      // slack column: -1 on every row
    if (gl == 0 && n > Nc) L.qp.xt[Nc] = 0.0;
    g_sync();
    if (st && gl == 0) st[9] = __builtin_amdgcn_s_memtime();
    // A_d is the same for every lane: with one controller per wave it is kept one entry per lane in
    // registers and broadcast through v_readlane, otherwise it is read from LDS.
    const bool a_in_regs = GS == 64 && Nc * NX * NX <= 128;
    double areg[2] = {0.0, 0.0};
    if (a_in_regs) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (gl + 64 * h < Nc * NX * NX) areg[h] = L.Ak[gl + 64 * h];
    }
    for (int j0 = 0; j0 < n; j0 += GS) {           // one trip when Nc + 1 <= GS
        const int j = j0 + gl;
        const bool col = j < Nc, fr = j == Nc;
        double v[NX], pf[NX], sf[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            v[r] = col ? L.Bk[j * NX + r] : 0.0;
            pf[r] = c.f0[r]; sf[r] = c.f0[r];
        }
        if (fr) {
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a = fma(L.Ak[r * NX + l], L.x0[l], a);
                v[r] = a;
            }
        }
        auto emit = [&](int d) {
            if (col && j + d < Nc) {
#pragma unroll
                for (int r = 0; r < NX; ++r) L.Gam[MpcLds::tri(j + d, j) * NX + r] = v[r];
            }
            if (fr) {
#pragma unroll
                for (int r = 0; r < NX; ++r) L.xi[d * NX + r] = v[r] + sf[r];
            }
        };
        emit(0);
        auto step = [&](const double* A, int d) {
            double nv[NX], np[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                double a = 0.0, b2 = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) { a = fma(A[r * NX + l], v[l], a); b2 = fma(A[r * NX + l], pf[l], b2); }
                nv[r] = a; np[r] = b2;
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) { v[r] = nv[r]; pf[r] = np[r]; sf[r] += np[r]; }
            emit(d);
        };
        if (a_in_regs) {
#pragma unroll
            for (int d = 1; d < (NC ? NC : 1); ++d) {          // fully unrolled when the horizon is a constant
                double A[NX * NX];
#pragma unroll
                for (int e = 0; e < NX * NX; ++e) {
                    const int idx = d * NX * NX + e;
                    A[e] = readlane_d(areg[idx >> 6], idx & 63);
                }
                step(A, d);
            }
            for (int d = NC ? NC : 1; d < Nc; ++d) {
                double A[NX * NX];
#pragma unroll
                for (int e = 0; e < NX * NX; ++e) {
                    const int idx = d * NX * NX + e;
                    const double lo_ = readlane_d(areg[0], idx & 63), hi_ = readlane_d(areg[1], idx & 63);
                    A[e] = idx < 64 ? lo_ : hi_;
                }
                step(A, d);
            }
        } else {
            for (int d = 1; d < Nc; ++d) {
                double A[NX * NX];
#pragma unroll
                for (int e = 0; e < NX * NX; ++e) A[e] = L.Ak[d * NX * NX + e];
                step(A, d);
            }
        }
    }
    g_sync();
    for (int e = gl; e < Nc * NX; e += GS) {       // qd_i = Q (xi_i - Xs_i)
        const int i = e / NX, r = e % NX;
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) a = fma(c.Q[r * NX + l], L.xi[i * NX + l] - L.Xs[i * NX + l], a);
        qd[e] = a;
    }
    g_sync();
    if (st && gl == 0) st[10] = __builtin_amdgcn_s_memtime();
    constexpr int UNR_I = NC ? NC : 5;
    const int items = Nc * (Nc + 1) / 2;
    for (int t0 = 0; t0 < items; t0 += GS) {
        const int t = t0 + gl;
        const bool on = t < items;
        int a_ = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((a_ + 1) * (a_ + 2) / 2 <= t) ++a_;
        while (a_ * (a_ + 1) / 2 > t) --a_;
        if (!on) a_ = 0;
        const int b_ = on ? t - a_ * (a_ + 1) / 2 : 0;
        double g = 0.0, f = 0.0;
#pragma unroll UNR_I
        for (int i = 0; i < Nc; ++i) {             // Gamma_{i,a} = 0 for i < a: uniform trip count, clamped reads
            double ga[NX], gb[NX];
            const int ic = i > a_ ? i : a_;          // b <= a <= ic: both blocks exist
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                ga[r] = L.Gam[MpcLds::tri(ic, a_) * NX + r]; gb[r] = L.Gam[MpcLds::tri(ic, b_) * NX + r];
                ga[r] = i >= a_ ? ga[r] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                double qg = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) qg = fma(c.Q[r * NX + l], gb[l], qg);
                g = fma(ga[r], qg, g);
                f = fma(ga[r], qd[i * NX + r], f);
            }
        }
        if (on) {
            Lm[a_ * n + b_] = 2.0 * (g + (a_ == b_ ? c.R : 0.0));
            if (a_ == b_) L.qp.xt[a_] = 2.0 * (f + ((NC == 0 && c.as_written) ? c.R : -c.R) * L.Us[a_]);
            if (!state_rows) {
                // y_{a+1} = C_a x_{a+1} + h0:  rows -/+ of block a+1 (the last block has its own two rows)
                double cg = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) cg = fma(L.Ck[a_ * NX + r], L.Gam[MpcLds::tri(a_, b_) * NX + r], cg);
                const int rowm = a_ + 1 < Nc ? 4 * (a_ + 1) + 2 : 4 * Nc;
                Nt[b_ * mt + rowm] = -cg;
                Nt[b_ * mt + rowm + 1] = cg;
            } else {
                // x_{a+1}: rows -x (NX of them) then +x of block a+1; terminal rows after the last block
                const int rowm = a_ + 1 < Nc ? 2 * (NX + 1) * (a_ + 1) + 2 : 2 * (NX + 1) * Nc;
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    const double gv = L.Gam[MpcLds::tri(a_, b_) * NX + r];
                    Nt[b_ * mt + rowm + r] = -gv;
                    Nt[b_ * mt + rowm + NX + r] = gv;
                }
            }
        }
    }
    if (st && gl == 0) st[11] = __builtin_amdgcn_s_memtime();
    // rows 4i+{0,1}: -/+ u_i;  4i+{2,3}: -/+ y_i (y_0 measured);  last two: -/+ y_Nc
    for (int row = gl; row < mt; row += GS) {
        if (state_rows) {
            const int R6 = 2 * (NX + 1);
            const int blk = row < R6 * Nc ? row / R6 : Nc;
            const int kk = row < R6 * Nc ? row % R6 : 2 + (row - R6 * Nc);
            double hval;
            if (kk < 2) {
                Nt[blk * mt + row] = kk ? 1.0 : -1.0;
                hval = kk ? c.umax : -c.umin;
            } else {
                const int r = (kk - 2) % NX;
                const bool up = kk - 2 >= NX;
                const double xv = blk == 0 ? L.x0[r] : L.xi[(blk - 1) * NX + r];
                hval = up ? c.xmax[r] - xv : xv - c.xmin[r];
            }
            L.qp.hv[row] = hval;
            continue;
        }
        const int blk = row < 4 * Nc ? row / 4 : Nc;
        const int kind = row < 4 * Nc ? row % 4 : 2 + (row - 4 * Nc);
        const double sg = (kind & 1) ? 1.0 : -1.0;
        double hval;
        if (kind < 2) {
            Nt[blk * mt + row] = sg;
            hval = kind == 0 ? -c.umin : c.umax;
        } else {
            double yv = ymeas;
            if (blk > 0) {
                yv = c.h0;
#pragma unroll
                for (int r = 0; r < NX; ++r) yv = fma(L.Ck[(blk - 1) * NX + r], L.xi[(blk - 1) * NX + r], yv);
            }
            hval = (kind == 2 ? -c.ymin : c.ymax) - sg * yv;
        }
        L.qp.hv[row] = hval;
    }
    g_sync();
}

// x_pred = Phi x0 + Gamma u + F0 with u = qp.xt[0..Nc); writes into xi (in place)
template <int GS, int NX, int NC = 0>
__device__ __forceinline__ void mpc_predict(const MpcConst& c, MpcLds& L, int gl) {
    const int Nc = NC ? NC : c.Nc;
    g_sync();
    for (int e = gl; e < Nc * NX; e += GS) {
        const int i = e / NX, r = e % NX;
        double a = L.xi[e];
        for (int j = 0; j <= i; ++j) a += L.Gam[MpcLds::tri(i, j) * NX + r] * L.qp.xt[j];
        L.xi[e] = a;
    }
    g_sync();
}

// ---------------------------------------------------------------- stand-alone MPC QP ----------
struct MpcQpArgs {
    int nb, nx;
    const double *lA, *lB, *lC, *x0, *ymeas, *Xs, *Us;
    double *Ge, *Fe, *Le, *hv, *u, *eps, *xpred;
    int* status;
    unsigned long long* stamps;   // optional diagnostics: 16 clock values per problem
    MpcConst c;
};

// (workgroups have at most four waves; saying so lets the nx > 4 instantiations keep their nx x nx blocks in registers)
template <int GS, int MPL, int NX, int NC = 0>
__global__ __launch_bounds__(NX > 4 ? 256 : 1024) void mpc_qp_kernel(MpcQpArgs a) {
    extern __shared__ __align__(16) double mpc_lds[];
    const int gpb = blockDim.x / GS, g = threadIdx.x / GS, gl = threadIdx.x % GS;
    const int prob = blockIdx.x * gpb + g;
    if (prob >= a.nb) return;
    const int Nc = NC ? NC : a.c.Nc, n = NC ? NC + 1 : mpc_nvar(Nc, a.c.slack), mt = NC ? 4 * NC + 2 : mpc_nrows(Nc, NX, a.c.mode);
    MpcLds L;
    L.carve(mpc_lds + (size_t)g * mpc_lds_doubles(Nc, NX, n, mt), Nc, NX, n, mt);
    unsigned long long* st = a.stamps ? a.stamps + (size_t)prob * 16 : nullptr;
    if (st && gl == 0) st[0] = __builtin_amdgcn_s_memtime();
    for (int e = gl; e < Nc * NX * NX; e += GS) L.Ak[e] = a.lA[(size_t)prob * Nc * NX * NX + e];
    for (int e = gl; e < Nc * NX; e += GS) {
        L.Bk[e] = a.lB[(size_t)prob * Nc * NX + e];
        L.Ck[e] = a.lC ? a.lC[(size_t)prob * Nc * NX + e] : 0.0;
        L.Xs[e] = a.Xs[(size_t)prob * Nc * NX + e];
    }
    for (int e = gl; e < Nc; e += GS) L.Us[e] = a.Us[(size_t)prob * Nc + e];
    for (int e = gl; e < NX; e += GS) L.x0[e] = a.x0[(size_t)prob * NX + e];
    g_sync();
    mpc_assemble<GS, NX, NC>(a.c, L, a.ymeas ? a.ymeas[prob] : 0.0, gl, st);
    if (st && gl == 0) st[1] = __builtin_amdgcn_s_memtime();
    if (a.Ge) for (int e = gl; e < n * n; e += GS) a.Ge[(size_t)prob * n * n + e] = L.qp.Lm[e];
    if (a.Fe) for (int e = gl; e < n; e += GS) a.Fe[(size_t)prob * n + e] = L.qp.xt[e];
    if (a.Le) for (int e = gl; e < mt * n; e += GS) a.Le[(size_t)prob * mt * n + e] = L.qp.Nt[(e % n) * mt + e / n];
    if (a.hv) for (int e = gl; e < mt; e += GS) a.hv[(size_t)prob * mt + e] = L.qp.hv[e];
    if (!a.u && !a.xpred && !a.eps) return;
    int stt;
    if constexpr (GS == 64 && NC > 0 && NC + 1 <= 16) stt = gi_solve<GS, MPL, NC + 1>(n, mt, 0, L.qp, nullptr, nullptr, st ? st + 2 : nullptr);
    else stt = (GS == 64 && n <= 16) ? gi_solve<GS, MPL, (GS == 64 ? 16 : 0)>(n, mt, 0, L.qp, nullptr, nullptr, st ? st + 2 : nullptr)
                                     : gi_solve<GS, MPL>(n, mt, 0, L.qp, nullptr, nullptr, st ? st + 2 : nullptr);
    const double epsv = n > Nc ? L.qp.xt[Nc] : 0.0;
    mpc_predict<GS, NX, NC>(a.c, L, gl);
    if (st && gl == 0) st[7] = __builtin_amdgcn_s_memtime();
    if (a.u) for (int e = gl; e < Nc; e += GS) a.u[(size_t)prob * Nc + e] = L.qp.xt[e];
    if (a.xpred) for (int e = gl; e < Nc * NX; e += GS) a.xpred[(size_t)prob * Nc * NX + e] = L.xi[e];
    if (gl == 0) {
        if (a.eps) a.eps[prob] = epsv;
        if (a.status) a.status[prob] = stt;
    }
}

// ---------------------------------------------------------------- closed loop ------------------
struct CtrlDev {
    int B, Nc, nx, maxit, n_sim, na, nyh;   // nyh = nb + 1 output-history entries
    double tol;
    MpcConst c;
    double u0, ustd, y0, ystd, dt;
    double x0v[8], xstdv[8];   // state normalisation of the state-feedback mode (norm.y0, norm.ystd vectors)
    double plant[6];
    // state (fp64)
    double *x0n, *xpred, *uplan, *ymeas, *xp, *uhist, *yhist, *eps;
    int *cnt, *act, *nact;
    // long-horizon solve in two stages: controllers the first-stage kernel leaves undecided, per pass
    // (slow[it * B + ..], nslow[it]); null when that kernel is not used
    int *slow, *nslow;
    unsigned long long* qp_fail;   // failed solves
    unsigned* qp_stat;             // [B][2]: solves, active-set iterations per controller
    int* qp_worst;                 // [B]: worst QP status among the passes of the current step (cleared by the plant kernels)
    int us_fixed;                  // state feedback as written: the input targets of every step are Us[0 .. Nc) (NMPC.py:580)
    // fp32 scheduling points and embedding outputs
    float *xs32, *us32, *xc32, *A32, *B32, *C32;
    const float *enc, *hplain, *fplain;
    int n_enc_in, hidden;
    int plant_mode;   // 0 disc ODE + observer, 1 disc ODE with measured state, 2 the model as its own plant
    // long-horizon kernels: final working set of each controller's last QP solve (row ids, 64 per controller) and
    // its size; null when the working set is not carried over
    unsigned char* wset;
    int* wcnt;
    int warm;   // NMPC_QP_WARM: 0 the carried set is recorded but every solve starts cold, 1 warm-started iteration,
                // 2 (default) the verified fast path first
    unsigned long long* stamps;   // optional diagnostics (long-horizon kernels): 16 clock values per controller
};

struct RunPtrs {
    const double *Xs, *Us, *noise;
    double *x_log, *u_log, *e_log, *y_log, *y_est_log, *plan_log;
    int* lpv_counter;
    signed char* qp_status_log;   // [batch, n_sim]: worst QP status of the step's passes (0 = every solve succeeded)
    int noise_ld;   // controllers per row of `noise` (the whole batch, also for a slice view)
};

// One LPV pass of controller b by one group of GS lanes (lane gl of the group): condensing, QP assembly and solve,
// prediction, stop test; writes the plan, the prediction and the next pass's scheduling points.  Returns (to every
// lane of the group) whether the controller needs another pass.
template <int GS, int MPL, int NX, int NC = 0>
__device__ __forceinline__ bool mpc_solve_body(const CtrlDev& d, const RunPtrs& rp, int it, int k, int b, double* lds, int gl) {
    const int Nc = NC ? NC : d.Nc, n = NC ? NC + 1 : mpc_nvar(Nc, d.c.slack), mt = NC ? 4 * NC + 2 : mpc_nrows(Nc, NX, d.c.mode);
    const int ntot = d.n_sim + Nc;
    MpcLds L;
    L.carve(lds, Nc, NX, n, mt);
    for (int e = gl; e < Nc * NX * NX; e += GS) L.Ak[e] = (double)d.A32[(size_t)b * Nc * NX * NX + e];
    for (int e = gl; e < Nc * NX; e += GS) {
        L.Bk[e] = (double)d.B32[(size_t)b * Nc * NX + e];
        L.Ck[e] = d.c.mode ? 0.0 : (double)d.C32[(size_t)b * Nc * NX + e];
        const int i = e / NX, r = e % NX;
        L.Xs[e] = rp.Xs[((size_t)b * ntot + (k + 1 + i)) * NX + r];
    }
    for (int e = gl; e < Nc; e += GS) {
        L.Us[e] = rp.Us[(size_t)b * ntot + ((NC == 0 && d.us_fixed) ? 0 : k) + e];
        L.uold[e] = d.uplan[(size_t)b * Nc + e];
    }
    for (int e = gl; e < NX; e += GS) L.x0[e] = d.x0n[(size_t)b * NX + e];
    g_sync();
    mpc_assemble<GS, NX, NC>(d.c, L, d.ymeas[b], gl);
    int qp_it = 0;
    int st;
    if constexpr (GS == 64 && NC > 0 && NC + 1 <= 16) st = gi_solve<GS, MPL, NC + 1>(n, mt, 0, L.qp, nullptr, &qp_it);
    else st = (GS == 64 && n <= 16) ? gi_solve<GS, MPL, (GS == 64 ? 16 : 0)>(n, mt, 0, L.qp, nullptr, &qp_it)
                                    : gi_solve<GS, MPL>(n, mt, 0, L.qp, nullptr, &qp_it);
    if (st != 0) {
        // A failed solve (infeasible hard constraints, iteration limit, indefinite Hessian) leaves no plan: the
        // controller keeps the plan, prediction and scheduling points it came with (the previous step's, shifted),
        // takes no further pass in this step and is flagged in the status log; the rest of the batch is unaffected.
        if (gl == 0) {
            d.cnt[b] = it + 1;
            atomicAdd(d.qp_fail, 1ull);
            d.qp_stat[2 * b] += 1u;
            d.qp_stat[2 * b + 1] += (unsigned)qp_it;
            if (st > d.qp_worst[b]) d.qp_worst[b] = st;
        }
        return false;
    }
    const double epsv = n > Nc ? L.qp.xt[Nc] : 0.0;
    mpc_predict<GS, NX, NC>(d.c, L, gl);
    double du = 0.0;
    for (int e = gl; e < Nc; e += GS) {
        const double un = L.qp.xt[e], dd = un - L.uold[e];
        du += dd * dd;
        d.uplan[(size_t)b * Nc + e] = un;
        d.us32[(size_t)b * Nc + e] = (float)un;
    }
    du = g_sum<GS>(du);
    for (int e = gl; e < Nc * NX; e += GS) {
        const double xv = L.xi[e];
        d.xpred[(size_t)b * Nc * NX + e] = xv;
        d.xc32[(size_t)b * Nc * NX + e] = (float)xv;
        // A_k, B_k of the next pass are scheduled at x_0 = x0n, x_k = x_pred[k-1]
        d.xs32[(size_t)b * Nc * NX + e] = e < NX ? (float)L.x0[e] : (float)L.xi[e - NX];
    }
    if (gl == 0) {
        d.eps[b] = epsv;
        d.cnt[b] = it + 1;
        d.qp_stat[2 * b] += 1u;                 // per-controller tallies: no shared-address atomics
        d.qp_stat[2 * b + 1] += (unsigned)qp_it;
    }
    return it + 1 < d.maxit && !(sqrt(du) < d.tol);
}

template <int GS, int MPL, int NX, int NC = 0>
__global__ __launch_bounds__(NX > 4 ? 256 : 1024) void mpc_solve_kernel(CtrlDev d, RunPtrs rp, int it, int k) {
    extern __shared__ __align__(16) double mpc_lds[];
    const int gpb = blockDim.x / GS, g = threadIdx.x / GS, gl = threadIdx.x % GS;
    const int gid = blockIdx.x * gpb + g;
    const int count = it == 0 ? d.B : min(d.nact[it], d.B);
    if (gid >= count) return;
    const int b = it == 0 ? gid : d.act[(size_t)it * d.B + gid];
    const int Nc = NC ? NC : d.Nc, n = NC ? NC + 1 : mpc_nvar(Nc, d.c.slack), mt = NC ? 4 * NC + 2 : mpc_nrows(Nc, NX, d.c.mode);
    const bool more = mpc_solve_body<GS, MPL, NX, NC>(d, rp, it, k, b, mpc_lds + (size_t)g * mpc_lds_doubles(Nc, NX, n, mt), gl);
    if (gl == 0 && more) {
        const int pos = atomicAdd(&d.nact[it + 1], 1);
        d.act[(size_t)(it + 1) * d.B + pos] = b;
    }
}

// launch geometry shared by the stand-alone MPC QP and the closed-loop solve kernel
struct MpcGeom { int GS, gpb, mpl; size_t lds; };

// per-state-dimension launchers (mpc_nx<N>.hip): pick the instantiation for the geometry and launch
int dispatch_mpc_qp_nx2(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_qp_nx3(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_qp_nx4(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_qp_nx5(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_qp_nx6(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_qp_nx7(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_qp_nx8(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx2(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx3(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx4(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx5(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx6(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx7(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);
int dispatch_mpc_solve_nx8(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s);

template <typename K>
static int set_lds_attr(K kernel) {
    NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

// NC: horizon as a compile-time constant (0 = read it from the arguments)
template <int GS, int MPL, int NX, int NC = 0>
static int launch_mpc_qp(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_qp_kernel<GS, MPL, NX, NC>)) return 1;
        mark_used_on_device(once);
    }
    hipLaunchKernelGGL((mpc_qp_kernel<GS, MPL, NX, NC>), dim3((a.nb + g.gpb - 1) / g.gpb), dim3(g.gpb * GS), g.lds, s, a);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int GS, int MPL, int NX, int NC = 0>
static int launch_mpc_solve(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        if (set_lds_attr(mpc_solve_kernel<GS, MPL, NX, NC>)) return 1;
        mark_used_on_device(once);
    }
    hipLaunchKernelGGL((mpc_solve_kernel<GS, MPL, NX, NC>), dim3((d.B + g.gpb - 1) / g.gpb), dim3(g.gpb * GS), g.lds, s, d, rp, it, k);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

// (GS, MPL) pairs instantiated: GS=16 rows/lane 1..4 (Nc <= 15), GS=64 rows/lane 1, 2, 4 (Nc <= 63)
#define NMPC_GEOM_SWITCH(FN, NX_, g, ...)                                         \
    if (g.GS == 16 && g.mpl <= 1) return FN<16, 1, NX_>(__VA_ARGS__);             \
    if (g.GS == 16 && g.mpl <= 2) return FN<16, 2, NX_>(__VA_ARGS__);             \
    if (g.GS == 16 && g.mpl <= 3) return FN<16, 3, NX_>(__VA_ARGS__);             \
    if (g.GS == 16 && g.mpl <= 4) return FN<16, 4, NX_>(__VA_ARGS__);             \
    if (g.GS == 64 && g.mpl <= 1) return FN<64, 1, NX_>(__VA_ARGS__);             \
    if (g.GS == 64 && g.mpl <= 2) return FN<64, 2, NX_>(__VA_ARGS__);             \
    if (g.GS == 64 && g.mpl <= 4) return FN<64, 4, NX_>(__VA_ARGS__);

// body of one mpc_nx<N>.hip
#define NMPC_DEFINE_MPC_DISPATCH(NX_)                                                                              \
    int dispatch_mpc_qp_nx##NX_(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s) {                             \
        NMPC_GEOM_SWITCH(launch_mpc_qp, NX_, g, a, g, s)                                                           \
        set_error("nmpc_mpc_qp: no kernel for nx=%d GS=%d rows/lane=%d", a.nx, g.GS, g.mpl);                       \
        return 1;                                                                                                  \
    }                                                                                                              \
    int dispatch_mpc_solve_nx##NX_(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s) { \
        NMPC_GEOM_SWITCH(launch_mpc_solve, NX_, g, d, rp, it, k, g, s)                                             \
        set_error("nmpc_ctrl_run: no kernel for nx=%d GS=%d rows/lane=%d", d.nx, g.GS, g.mpl);                     \
        return 1;                                                                                                  \
    }

// state dimensions 5..8: whole-wave groups only (a 16-lane group has too few registers for the nx x nx blocks)
#define NMPC_GEOM_SWITCH64(FN, NX_, g, ...)                                       \
    if (g.GS == 64 && g.mpl <= 1) return FN<64, 1, NX_>(__VA_ARGS__);             \
    if (g.GS == 64 && g.mpl <= 2) return FN<64, 2, NX_>(__VA_ARGS__);             \
    if (g.GS == 64 && g.mpl <= 4) return FN<64, 4, NX_>(__VA_ARGS__);
#define NMPC_DEFINE_MPC_DISPATCH64(NX_)                                                                            \
    int dispatch_mpc_qp_nx##NX_(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s) {                             \
        NMPC_GEOM_SWITCH64(launch_mpc_qp, NX_, g, a, g, s)                                                         \
        set_error("nmpc_mpc_qp: no kernel for nx=%d GS=%d rows/lane=%d", a.nx, g.GS, g.mpl);                       \
        return 1;                                                                                                  \
    }                                                                                                              \
    int dispatch_mpc_solve_nx##NX_(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s) { \
        NMPC_GEOM_SWITCH64(launch_mpc_solve, NX_, g, d, rp, it, k, g, s)                                           \
        set_error("nmpc_ctrl_run: no kernel for nx=%d GS=%d rows/lane=%d", d.nx, g.GS, g.mpl);                     \
        return 1;                                                                                                  \
    }

}  // namespace nmpc
