// Dual active-set (Goldfarb-Idnani) solver for small strictly convex QPs, one problem per lane
// GROUP (GS = 16, 32 or 64 lanes of one wavefront), all data in LDS, fp64.
//
//     min 1/2 z'Pz + q'z   s.t.  a_i'z = b_i (i < meq),   g_i'z <= h_i
//
// (reference: the qpsolvers.solve_qp call sites NMPC_test.py:233, NMPC.py:458,602,767,
// mpcUtil.py:224,302; "quadprog", NMPC.py:458, is this algorithm.)
//
// Formulation.  With P = L L', constraint normals n_i = -g_i and the change of variables
// zt = L'z the Hessian becomes the identity, constraint i reads  s_i = h_i + nt_i . zt >= 0  with
// nt_i = L^-1 n_i, and the whole iteration runs on the n x mt matrix Nt = [nt_i]:
//   - the working-set Gram matrix K_WW = Nt_W' Nt_W is kept as its Cholesky factor Lw; adding
//     constraint p appends the row [l, delta] with l = Lw^-1 Nt_W' nt_p, delta^2 = |nt_p|^2 - |l|^2
//     (delta^2 is also the curvature of the primal step), dropping one rebuilds Lw;
//   - dual direction r = Lw^-T l, primal direction v = nt_p - Nt_W r (in zt coordinates);
//   - every lane owns constraints i = gl, gl+GS, ... and updates their slacks s_i += t nt_i . v,
//     so the "most violated constraint" search is a group reduction over registers.
// The primal point is recovered once at the end from the multipliers:  L' z = zt_unc + Nt_W lam.
#pragma once
#include <hip/hip_runtime.h>

namespace nmpc {

constexpr double QP_INF = 1e300;

__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Whole-wave (GS = 64) reductions stay off the LDS crossbar: butterflies inside each row of 16
// lanes on the DPP network (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), the four row
// results through v_readlane.  Every lane ends with the same value; the order of operations is fixed.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d<DPP_XOR1>(v); v += dpp_d<DPP_XOR2>(v); v += dpp_d<DPP_HALF_MIRROR>(v); v += dpp_d<DPP_MIRROR>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double wave_min(double v) {
    v = fmin(v, dpp_d<DPP_XOR1>(v)); v = fmin(v, dpp_d<DPP_XOR2>(v));
    v = fmin(v, dpp_d<DPP_HALF_MIRROR>(v)); v = fmin(v, dpp_d<DPP_MIRROR>(v));
    return fmin(fmin(readlane_d(v, 0), readlane_d(v, 16)), fmin(readlane_d(v, 32), readlane_d(v, 48)));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, __builtin_amdgcn_mov_dpp(v, DPP_XOR1, 0xf, 0xf, true)); v = min(v, __builtin_amdgcn_mov_dpp(v, DPP_XOR2, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_mov_dpp(v, DPP_HALF_MIRROR, 0xf, 0xf, true)); v = min(v, __builtin_amdgcn_mov_dpp(v, DPP_MIRROR, 0xf, 0xf, true));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// value of group lane src_gl, which must be the same in every lane of the group
template <int GS>
__device__ __forceinline__ double g_bcast(double v, int src_gl) {
    if constexpr (GS == 64) return readlane_d(v, src_gl);
    const int lane = threadIdx.x & 63;
    return __shfl(v, (lane & ~(GS - 1)) + src_gl, 64);
}
template <int GS>
__device__ __forceinline__ int g_bcast_i(int v, int src_gl) {
    if constexpr (GS == 64) return __builtin_amdgcn_readlane(v, src_gl);
    const int lane = threadIdx.x & 63;
    return __shfl(v, (lane & ~(GS - 1)) + src_gl, 64);
}
// value of the next lane of the group (cyclic)
template <int GS>
__device__ __forceinline__ double g_next(double v) {
    const int lane = threadIdx.x & 63;
    return __shfl(v, (lane & ~(GS - 1)) + ((lane + 1) & (GS - 1)), 64);
}
template <int GS>
__device__ __forceinline__ int g_next_i(int v) {
    const int lane = threadIdx.x & 63;
    return __shfl(v, (lane & ~(GS - 1)) + ((lane + 1) & (GS - 1)), 64);
}
template <int GS>
__device__ __forceinline__ double g_sum(double v) {
    if constexpr (GS == 64) return wave_sum(v);
#pragma unroll
    for (int m = GS / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// minimum and its index (ties -> smaller index), result identical in every lane of the group
template <int GS>
__device__ __forceinline__ void g_argmin(double& v, int& idx) {
    if constexpr (GS == 64) {
        const double m = wave_min(v);
        idx = wave_min_i(v == m ? idx : 0x7fffffff);
        v = m;
        return;
    }
#pragma unroll
    for (int m = GS / 2; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(v, m, 64);
        const int oi = __shfl_xor(idx, m, 64);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}
// orders this wave's LDS traffic for the compiler; the hardware executes one wave's DS
// instructions in order
__device__ __forceinline__ void g_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In-place Cholesky of the leading nn x nn block of M (row-major, leading dimension ld), lower
// factor; invd[j] = 1 / L[j][j].  Returns false if a pivot is not positive.
template <int GS>
__device__ __forceinline__ bool g_chol(double* M, int nn, int ld, double* invd, int gl) {
    for (int j = 0; j < nn; ++j) {
        const double d = M[j * ld + j];
        if (!(d > 0.0)) return false;
        const double inv = 1.0 / sqrt(d);
        g_sync();
        for (int i = gl; i < nn; i += GS)
            if (i >= j) M[i * ld + j] *= inv;
        if (gl == 0) invd[j] = inv;
        g_sync();
        for (int i = gl; i < nn; i += GS)
            if (i > j) {
                const double lij = M[i * ld + j];
                for (int c = j + 1; c <= i; ++c) M[i * ld + c] -= lij * M[c * ld + j];
            }
        g_sync();
    }
    return true;
}

// 1/sqrt(d) to full fp64 accuracy: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_d(double d) {
    double r = __builtin_amdgcn_rsq(d);
    r = r * fma(-0.5 * d * r, r, 1.5);
    r = r * fma(-0.5 * d * r, r, 1.5);
    return r;
}

// Whole-wave (GS = 64) Cholesky for any nn <= 64 with lane i owning row i of M in LDS: the pivot
// column travels through v_readlane and every lane touches only its own row, so the factorisation
// needs no barrier.  Rows below the pivot take the symmetric update over their whole tail (the part
// right of the diagonal is never read again), which keeps the chunked inner loop free of per-element
// predicates.
__device__ __forceinline__ bool chol_rows_lds(double* M, int nn, int ld, double* invd) {
    constexpr int UNR = 8;
    const int lane = threadIdx.x & 63;
    if (nn <= 0) return true;
    const bool own = lane < nn;
    double* row = M + (own ? lane : nn - 1) * ld;   // idle lanes read row nn-1 and never store
    for (int j = 0; j < nn; ++j) {
        double lij = row[j];
        const double d = readlane_d(lij, j);
        if (!(d > 0.0)) return false;               // wave-uniform
        const double inv = rsqrt_d(d);
        lij *= inv;
        if (own && lane >= j) row[j] = lij;
        if (lane == 0) invd[j] = inv;
        int c0 = j + 1;
        for (; c0 + UNR <= nn; c0 += UNR) {
            double m[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) m[u] = row[c0 + u];
#pragma unroll
            for (int u = 0; u < UNR; ++u) m[u] = fma(-lij, readlane_d(lij, c0 + u), m[u]);
            if (own && lane > j) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) row[c0 + u] = m[u];
            }
        }
        for (; c0 < nn; ++c0) {
            const double m = fma(-lij, readlane_d(lij, c0), row[c0]);
            if (own && lane > j) row[c0] = m;
        }
    }
    return true;
}

// Whole-wave forward substitution of every column of Nt and of xt by the factor in Lm (any n <= 64):
// column <- L^-1 (-column).  Constraint columns: one per lane and slot, all slots advancing together
// so that one broadcast read of L[k][l] serves them all (lanes past the last column read in-bounds
// neighbours and store nothing).  xt: lane k owns component k and the solved entries travel through
// v_readlane.  Needs 64 * MPL >= mt; reads up to 64 * MPL - mt doubles past the end of Nt, which the
// work area carved behind it (mt + 8 n doubles, n > 16 on this path) covers for MPL in {1, 2, 4}.
template <int MPL>
__device__ __forceinline__ void transform_cols_lds(const double* Lm, double* Nt, double* xt, const double* invd, int n, int mt) {
    constexpr int UNR = 4;
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < n; ++k) {
        double v[MPL];
#pragma unroll
        for (int sl = 0; sl < MPL; ++sl) v[sl] = -Nt[k * mt + lane + 64 * sl];
        int l0 = 0;
        for (; l0 + UNR <= k; l0 += UNR) {
            double lv[UNR], y[UNR][MPL];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                lv[u] = Lm[k * n + l0 + u];
#pragma unroll
                for (int sl = 0; sl < MPL; ++sl) y[u][sl] = Nt[(l0 + u) * mt + lane + 64 * sl];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int sl = 0; sl < MPL; ++sl) v[sl] = fma(-lv[u], y[u][sl], v[sl]);
        }
        for (; l0 < k; ++l0) {
            const double lv = Lm[k * n + l0];
#pragma unroll
            for (int sl = 0; sl < MPL; ++sl) v[sl] = fma(-lv, Nt[l0 * mt + lane + 64 * sl], v[sl]);
        }
        const double ik = invd[k];
#pragma unroll
        for (int sl = 0; sl < MPL; ++sl)
            if (lane + 64 * sl < mt) Nt[k * mt + lane + 64 * sl] = v[sl] * ik;
    }
    const int kr = lane < n ? lane : n - 1;
    double v = -xt[kr], res = 0.0;
    const double invl = invd[kr];
    double lnext = Lm[kr * n];
    for (int l = 0; l < n; ++l) {
        const double lcur = lnext;
        if (l + 1 < n) lnext = Lm[kr * n + l + 1];
        const double yl = readlane_d(v, l) * readlane_d(invl, l);
        if (lane == l) res = yl;
        if (lane > l) v = fma(-lcur, yl, v);
    }
    if (lane < n) xt[lane] = res;
}

// Whole-wave (GS = 64) Cholesky and transform for n <= NMAX with lane i holding row i of the
// matrix in registers: the pivot column travels through v_readlane (an SGPR broadcast), so no LDS
// round trip or barrier sits on the critical path.  On exit Lm holds L (row-major), invd the
// inverse diagonal, and every column of Nt as well as xt has been replaced by L^-1 (-column).
template <int NMAX, int MPL>
__device__ __forceinline__ bool chol_transform_rows(double* Lm, double* Nt, double* xt, double* invd, int n, int mt) {
    const int lane = threadIdx.x & 63;
    double row[NMAX], inv[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) row[c] = (lane < n && c <= lane && c < n) ? Lm[lane * n + c] : 0.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (j < n) {
            const double d = readlane_d(row[j], j);
            if (!(d > 0.0)) ok = false;
            inv[j] = rsqrt_d(d > 0.0 ? d : 1.0);
            row[j] *= inv[j];                      // lane j: sqrt(d); lanes i > j: L[i][j]
#pragma unroll
            for (int c = j + 1; c < NMAX; ++c)
                if (c < n) {
                    const double lcj = readlane_d(row[j], c);
                    if (lane >= c) row[c] = fma(-row[j], lcj, row[c]);
                }
        } else {
            inv[j] = 0.0;
        }
    }
    if (!ok) return false;
    if (lane < n) {
#pragma unroll
        for (int c = 0; c < NMAX; ++c)
            if (c <= lane && c < n) Lm[lane * n + c] = row[c];
    }
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
        if (j < n && lane == 0) invd[j] = inv[j];
    // forward substitution of every column (constraint normals and q), one column per lane and slot
#pragma unroll
    for (int sl = 0; sl <= MPL; ++sl) {
        const int cid = lane + 64 * sl;
        if (64 * sl <= mt) {                       // wave-uniform
            const bool act = cid <= mt;
            double* col = cid < mt ? Nt + cid : xt;
            const int st = cid < mt ? mt : 1;
            double y[NMAX];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) {
                if (k < n) {
                    double v = act ? -col[k * st] : 0.0;
#pragma unroll
                    for (int l = 0; l < k; ++l) v = fma(-readlane_d(row[l], k), y[l], v);
                    y[k] = v * inv[k];
                } else {
                    y[k] = 0.0;
                }
            }
            if (act) {
#pragma unroll
                for (int k = 0; k < NMAX; ++k)
                    if (k < n) col[k * st] = y[k];
            }
        }
    }
    return true;
}

// LDS footprint in doubles of one problem (ints of the working set included, rounded up)
__host__ __device__ inline int qp_lds_doubles(int n, int mt) { return 2 * n * n + n * mt + mt + 8 * n; }

struct QPWork {
    double *Lm, *Nt, *hv, *xt, *Lw, *vv, *rW, *lamW, *invdL, *invdW, *lam_out;
    int* Wl;
    __device__ void carve(double* base, int n, int mt) {
        Lm = base; Lw = Lm + n * n; Nt = Lw + n * n; hv = Nt + n * mt; xt = hv + mt; vv = xt + n; rW = vv + n;
        lamW = rW + n; invdL = lamW + n; invdW = invdL + n; lam_out = nullptr;
        Wl = reinterpret_cast<int*>(invdW + n);   // n ints in the 2n-double tail
    }
};

// On entry: Lm = P (full symmetric n x n), Nt[k*mt+i] = coefficient k of constraint row i
// (equalities first), hv = right-hand sides, xt = q.  On exit: xt = optimum z; if lam != null,
// lam[i] = multiplier of row i.  Requires n <= GS and mt <= GS*MPL.
// Returns 0 ok, 1 iteration limit, 2 infeasible, 3 P not positive definite.
// `stamp` (optional, diagnostics): 5 clock values [after chol, after transform, after slack init,
// after the active-set loop, at exit] written by the group's first lane.
// NMAX > 0 (only with GS = 64): n <= NMAX, Cholesky and transform run in registers.
template <int GS, int MPL, int NMAX = 0>
__device__ __forceinline__ int gi_solve(int n, int mt, int meq, QPWork& w, double* lam, int* iters_out,
                        unsigned long long* stamp = nullptr) {
    const int gl = (threadIdx.x & 63) & (GS - 1);
    double *Lm = w.Lm, *Nt = w.Nt, *hv = w.hv, *xt = w.xt, *Lw = w.Lw;
    int status = 0, iters = 0;
    g_sync();
    if constexpr (GS == 64 && NMAX > 0) {
        const bool ok = chol_transform_rows<NMAX, MPL>(Lm, Nt, xt, w.invdL, n, mt);
        if (stamp && gl == 0) stamp[0] = __builtin_amdgcn_s_memtime();
        if (!ok) {
            if (iters_out) *iters_out = 0;
            return 3;
        }
    } else if constexpr (GS == 64) {
        if (!chol_rows_lds(Lm, n, n, w.invdL)) {
            if (iters_out) *iters_out = 0;
            return 3;
        }
        if (stamp && gl == 0) stamp[0] = __builtin_amdgcn_s_memtime();
        g_sync();
        transform_cols_lds<MPL>(Lm, Nt, xt, w.invdL, n, mt);
    } else {
        if (!g_chol<GS>(Lm, n, n, w.invdL, gl)) {
            if (iters_out) *iters_out = 0;
            return 3;
        }
        if (stamp && gl == 0) stamp[0] = __builtin_amdgcn_s_memtime();
        // nt_i = L^-1 (-g_i) for every constraint, zt_unc = -L^-1 q  (one column per lane)
        for (int i = gl; i <= mt; i += GS) {
            double* col = i < mt ? Nt + i : xt;
            const int st = i < mt ? mt : 1;
            for (int k = 0; k < n; ++k) {
                double v = -col[k * st];
                for (int l = 0; l < k; ++l) v -= Lm[k * n + l] * col[l * st];
                col[k * st] = v * w.invdL[k];
            }
        }
    }
    g_sync();
    if (stamp && gl == 0) stamp[1] = __builtin_amdgcn_s_memtime();
    // FAST: one problem per wave (n <= 64).  Loops over the n variables run in unrolled chunks with
    // clamped addresses (a chunk's loads are in flight together), group broadcasts are v_readlane, and the
    // primal step keeps its direction in registers instead of LDS.
    constexpr bool FAST = GS == 64;
    constexpr int UNR = 4;   // loads in flight per chunk of the n-loops
    double s[MPL], sgn[MPL], iscale[MPL];
    unsigned amask = 0;
#pragma unroll
    for (int sl = 0; sl < MPL; ++sl) {
        const int i = gl + sl * GS;
        double v = 0.0;
        iscale[sl] = 1.0;
        if (i < mt) {
            v = hv[i];
            iscale[sl] = 1.0 / (1.0 + fabs(v));
            if constexpr (FAST) {
                for (int k0 = 0; k0 < n; k0 += UNR) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int k = k0 + u, kk = k < n ? k : n - 1;
                        const double a = Nt[kk * mt + i], x = xt[kk];
                        v = fma(k < n ? a : 0.0, x, v);
                    }
                }
            } else {
                for (int k = 0; k < n; ++k) v += Nt[k * mt + i] * xt[k];
            }
        }
        s[sl] = v; sgn[sl] = 1.0;
    }
    if (stamp && gl == 0) stamp[2] = __builtin_amdgcn_s_memtime();
    int q = 0, eq_next = 0;
    double lamj = 0.0;        // multiplier of working-set position gl
    int wj = -1;              // constraint index at working-set position gl
    const int max_iter = 10 * (n + mt) + 20;
    for (;;) {
        int p; double sp;
        if (eq_next < meq) {
            p = eq_next++;
            double sv = 0.0;
#pragma unroll
            for (int sl = 0; sl < MPL; ++sl) if (sl == p / GS) sv = s[sl];
            sp = g_bcast<GS>(sv, p % GS);
            if (fabs(sp) <= 1e-12 * (1.0 + fabs(hv[p]))) continue;
            if (sp > 0.0) {   // flip the row so that it reads as violated
                g_sync();
                for (int k = gl; k < n; k += GS) Nt[k * mt + p] = -Nt[k * mt + p];
                g_sync();
#pragma unroll
                for (int sl = 0; sl < MPL; ++sl) if (gl + sl * GS == p) { s[sl] = -s[sl]; sgn[sl] = -1.0; }
                sp = -sp;
            }
        } else {
            double best = QP_INF; int bi = 0x7fffffff;
#pragma unroll
            for (int sl = 0; sl < MPL; ++sl) {
                const int i = gl + sl * GS;
                if (i < mt && i >= meq && !(amask >> sl & 1u)) {
                    const double sc = s[sl] * iscale[sl];
                    if (sc < best) { best = sc; bi = i; }
                }
            }
            g_argmin<GS>(best, bi);
            if (!(best < -1e-10)) break;
            p = bi;
            double sv = 0.0;
#pragma unroll
            for (int sl = 0; sl < MPL; ++sl) if (sl == p / GS) sv = s[sl];
            sp = g_bcast<GS>(sv, p % GS);
        }
        double lam_p = 0.0;
        for (;;) {
            if (++iters > max_iter) { status = 1; break; }
            double kp = 0.0, kpp = 0.0;
            if constexpr (FAST) {
                const int wc = gl < q ? wj : p;
                for (int k0 = 0; k0 < n; k0 += UNR) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int k = k0 + u, kk = k < n ? k : n - 1;
                        double np_k = Nt[kk * mt + p];
                        const double nw = Nt[kk * mt + wc];
                        np_k = k < n ? np_k : 0.0;
                        kpp = fma(np_k, np_k, kpp);
                        kp = fma(nw, np_k, kp);
                    }
                }
                kp = gl < q ? kp : 0.0;
            } else {
                for (int k = 0; k < n; ++k) {
                    const double np_k = Nt[k * mt + p];
                    kpp += np_k * np_k;
                    if (gl < q) kp += Nt[k * mt + wj] * np_k;
                }
            }
            // l = Lw^-1 kp
            double v = kp, lj = 0.0;
            const int glq = gl < q ? gl : 0;                     // in-range row / column of Lw for idle lanes
            const double invw = FAST ? w.invdW[glq] : 0.0;       // lane i: 1 / Lw[i][i]
            if constexpr (FAST) {
                double lnext = Lw[glq * n];
                for (int i = 0; i < q; ++i) {
                    const double lcur = lnext;
                    if (i + 1 < q) lnext = Lw[glq * n + i + 1];
                    const double li = readlane_d(v, i) * readlane_d(invw, i);
                    if (gl == i) lj = li;
                    if (gl > i && gl < q) v = fma(-lcur, li, v);
                }
            } else {
                for (int i = 0; i < q; ++i) {
                    const double li = g_bcast<GS>(v, i) * w.invdW[i];
                    if (gl == i) lj = li;
                    if (gl > i && gl < q) v -= Lw[gl * n + i] * li;
                }
            }
            const double dd = kpp - g_sum<GS>(gl < q ? lj * lj : 0.0);
            // r = Lw^-T l
            v = lj;
            double rj = 0.0;
            if constexpr (FAST) {
                double lnext = q > 0 ? Lw[(q - 1) * n + glq] : 0.0;
                for (int i = q - 1; i >= 0; --i) {
                    const double lcur = lnext;
                    if (i > 0) lnext = Lw[(i - 1) * n + glq];
                    const double ri = readlane_d(v, i) * readlane_d(invw, i);
                    if (gl == i) rj = ri;
                    if (gl < i) v = fma(-lcur, ri, v);
                }
            } else {
                for (int i = q - 1; i >= 0; --i) {
                    const double ri = g_bcast<GS>(v, i) * w.invdW[i];
                    if (gl == i) rj = ri;
                    if (gl < i) v -= Lw[i * n + gl] * ri;
                }
            }
            double t1 = QP_INF; int kd = 0x7fffffff;
            if (gl < q && wj >= meq && rj > 1e-13) { t1 = lamj / rj; kd = gl; }
            g_argmin<GS>(t1, kd);
            const bool primal = dd > 1e-12 * kpp && kpp > 0.0;
            const double t2 = primal ? -sp / dd : QP_INF;
            const double t = t1 < t2 ? t1 : t2;
            if (!(t < QP_INF)) { status = 2; break; }
            if (primal) {
                if constexpr (FAST) {
                    // lane k < n: component k of the primal direction v = nt_p - Nt_W r
                    const int kr = gl < n ? gl : n - 1;
                    double a = Nt[kr * mt + p];
                    for (int j = 0; j < q; ++j)
                        a = fma(-Nt[kr * mt + __builtin_amdgcn_readlane(wj, j)], readlane_d(rj, j), a);
                    // all row slots advance together: a chunk keeps UNR * MPL loads in flight
                    double dot[MPL];
                    int ic[MPL];
#pragma unroll
                    for (int sl = 0; sl < MPL; ++sl) {
                        dot[sl] = 0.0;
                        ic[sl] = gl + sl * GS < mt ? gl + sl * GS : mt - 1;
                    }
                    for (int k0 = 0; k0 < n; k0 += UNR) {
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            const int k = k0 + u, kk = k < n ? k : n - 1;
                            const double ak = k < n ? readlane_d(a, kk) : 0.0;
#pragma unroll
                            for (int sl = 0; sl < MPL; ++sl) dot[sl] = fma(Nt[kk * mt + ic[sl]], ak, dot[sl]);
                        }
                    }
#pragma unroll
                    for (int sl = 0; sl < MPL; ++sl)
                        if (gl + sl * GS < mt) s[sl] = fma(t, dot[sl], s[sl]);
                } else {
                    g_sync();
                    if (gl < q) { w.rW[gl] = rj; w.Wl[gl] = wj; }
                    g_sync();
                    for (int k = gl; k < n; k += GS) {
                        double a = Nt[k * mt + p];
                        for (int j = 0; j < q; ++j) a -= Nt[k * mt + w.Wl[j]] * w.rW[j];
                        w.vv[k] = a;
                    }
                    g_sync();
#pragma unroll
                    for (int sl = 0; sl < MPL; ++sl) {
                        const int i = gl + sl * GS;
                        if (i < mt) {
                            double dot = 0.0;
                            for (int k = 0; k < n; ++k) dot += Nt[k * mt + i] * w.vv[k];
                            s[sl] += t * dot;
                        }
                    }
                }
            }
            if (gl < q) lamj -= t * rj;
            lam_p += t;
            if (primal && t2 <= t1) {          // full step: p joins the working set at position q
                g_sync();
                if (gl < q) Lw[q * n + gl] = lj;
                if (gl == q) {
                    const double dl = sqrt(dd);
                    Lw[q * n + q] = dl; w.invdW[q] = 1.0 / dl;
                    lamj = lam_p; wj = p;
                }
                g_sync();
#pragma unroll
                for (int sl = 0; sl < MPL; ++sl) if (gl + sl * GS == p) { s[sl] = 0.0; amask |= 1u << sl; }
                ++q;
                break;
            }
            // partial (or dual-only) step: constraint at position kd leaves the working set
            {
                const int dropped = g_bcast_i<GS>(wj, kd);
#pragma unroll
                for (int sl = 0; sl < MPL; ++sl) if (gl + sl * GS == dropped) amask &= ~(1u << sl);
                const double lnext = g_next<GS>(lamj);
                const int wnext = g_next_i<GS>(wj);
                if (gl >= kd && gl < q - 1) { lamj = lnext; wj = wnext; }
                --q;
                if (gl >= q) { lamj = 0.0; wj = -1; }
                // rebuild Lw = chol(Nt_W' Nt_W)
                g_sync();
                if (gl < q) w.Wl[gl] = wj;
                g_sync();
                if constexpr (FAST) {
                    if (gl < q)
                        for (int b = 0; b <= gl; ++b) {
                            const int wb = w.Wl[b];
                            double a = 0.0;
                            for (int k0 = 0; k0 < n; k0 += UNR) {
#pragma unroll
                                for (int u = 0; u < UNR; ++u) {
                                    const int k = k0 + u, kk = k < n ? k : n - 1;
                                    const double x = Nt[kk * mt + wj], y = Nt[kk * mt + wb];
                                    a = fma(k < n ? x : 0.0, y, a);
                                }
                            }
                            Lw[gl * n + b] = a;
                        }
                    g_sync();
                    if (!chol_rows_lds(Lw, q, n, w.invdW)) { status = 2; break; }
                    g_sync();
                } else {
                    if (gl < q)
                        for (int b = 0; b <= gl; ++b) {
                            const int wb = w.Wl[b];
                            double a = 0.0;
                            for (int k = 0; k < n; ++k) a += Nt[k * mt + wj] * Nt[k * mt + wb];
                            Lw[gl * n + b] = a;
                        }
                    g_sync();
                    if (!g_chol<GS>(Lw, q, n, w.invdW, gl)) { status = 2; break; }
                }
                double sv = 0.0;
#pragma unroll
                for (int sl = 0; sl < MPL; ++sl) if (sl == p / GS) sv = s[sl];
                sp = g_bcast<GS>(sv, p % GS);
            }
        }
        if (status) break;
    }
    if (stamp && gl == 0) stamp[3] = __builtin_amdgcn_s_memtime();
    // z:  L' z = zt_unc + Nt_W lam
    g_sync();
    if constexpr (FAST) {
        const int kr = gl < n ? gl : n - 1;
        double v = xt[kr];
        for (int j = 0; j < q; ++j) v = fma(Nt[kr * mt + __builtin_amdgcn_readlane(wj, j)], readlane_d(lamj, j), v);
        const double invl = w.invdL[kr];
        double xres = 0.0, lnext = Lm[(n - 1) * n + kr];
        for (int i = n - 1; i >= 0; --i) {
            const double lcur = lnext;
            if (i > 0) lnext = Lm[(i - 1) * n + kr];
            const double xi = readlane_d(v, i) * readlane_d(invl, i);
            if (gl == i) xres = xi;
            if (gl < i) v = fma(-lcur, xi, v);
        }
        g_sync();
        if (gl < n) xt[gl] = xres;
    } else {
        if (gl < q) { w.lamW[gl] = lamj; w.Wl[gl] = wj; }
        g_sync();
        for (int k = gl; k < n; k += GS) {
            double a = xt[k];
            for (int j = 0; j < q; ++j) a += Nt[k * mt + w.Wl[j]] * w.lamW[j];
            w.vv[k] = a;
        }
        g_sync();
        double v = gl < n ? w.vv[gl] : 0.0, xres = 0.0;
        for (int i = n - 1; i >= 0; --i) {
            const double xi = g_bcast<GS>(v, i) * w.invdL[i];
            if (gl == i) xres = xi;
            if (gl < i) v -= Lm[i * n + gl] * xi;
        }
        g_sync();
        if (gl < n) xt[gl] = xres;
    }
    if (lam) {   // a flipped equality row carries its sign with the lane that owns the row
        g_sync();
        for (int i = gl; i < mt; i += GS) lam[i] = 0.0;
        g_sync();
        if (gl < q) lam[wj] = lamj;
        g_sync();
#pragma unroll
        for (int sl = 0; sl < MPL; ++sl) {
            const int i = gl + sl * GS;
            if (i < mt) lam[i] *= sgn[sl];
        }
    }
    g_sync();
    if (stamp && gl == 0) stamp[4] = __builtin_amdgcn_s_memtime();
    if (iters_out) *iters_out = iters;
    return status;
}

}  // namespace nmpc
