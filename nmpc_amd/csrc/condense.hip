// Stand-alone batched versions of the reference's horizon matrices (mpcUtil.py:84-104, 322-331)
// and of lpv_rk4 (lpv_rk4.py:1-11), fp64.  The closed-loop step never materialises Phi/Gamma in
// HBM (controller.hip keeps them in LDS); these entry points serve callers that want the
// matrices themselves.
//
// All three matrices come from one prefix-product scan  P_0 = I, P_m = A_m P_{m-1}:
//   Phi_i = P_i A_0,   Gamma_{i,j} = P_{i-j} B_j (j <= i),   F0_i = (sum_{j<=i} P_j) f0.
// Thread (b, i) walks m = 0..i once, emitting Gamma_{i,i-m} as soon as P_m is known.
#include "../../include/nmpc_amd.h"
#include "common.h"

namespace nmpc {

template <int NX>
__global__ void condense_kernel(int nb, int Nc, int nu, const double* __restrict__ lA, const double* __restrict__ lB,
                                const double* __restrict__ f0, double* __restrict__ Phi, double* __restrict__ Gamma,
                                double* __restrict__ F0) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * Nc) return;
    const int b = t / Nc, i = t % Nc;
    const double* A = lA + (size_t)b * Nc * NX * NX;
    const double* Bm = lB ? lB + (size_t)b * Nc * NX * nu : nullptr;
    double P[NX * NX], S[NX * NX], T[NX * NX];
#pragma unroll
    for (int e = 0; e < NX * NX; ++e) { P[e] = (e / NX == e % NX) ? 1.0 : 0.0; S[e] = 0.0; }
    double* Gi = Gamma ? Gamma + ((size_t)b * Nc + i) * NX * Nc * nu : nullptr;
    for (int m = 0; m <= i; ++m) {
        if (m > 0) {
            const double* Am = A + (size_t)m * NX * NX;
#pragma unroll
            for (int r = 0; r < NX; ++r)
#pragma unroll
                for (int c = 0; c < NX; ++c) {
                    double a = 0.0;
#pragma unroll
                    for (int l = 0; l < NX; ++l) a += Am[r * NX + l] * P[l * NX + c];
                    T[r * NX + c] = a;
                }
#pragma unroll
            for (int e = 0; e < NX * NX; ++e) P[e] = T[e];
        }
#pragma unroll
        for (int e = 0; e < NX * NX; ++e) S[e] += P[e];
        if (Gi) {
            const int j = i - m;
            const double* Bj = Bm + (size_t)j * NX * nu;
            for (int cu = 0; cu < nu; ++cu)
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    double a = 0.0;
#pragma unroll
                    for (int l = 0; l < NX; ++l) a += P[r * NX + l] * Bj[l * nu + cu];
                    Gi[(size_t)r * Nc * nu + j * nu + cu] = a;
                }
        }
    }
    if (Gi)
        for (int j = i + 1; j < Nc; ++j)
            for (int cu = 0; cu < nu; ++cu)
#pragma unroll
                for (int r = 0; r < NX; ++r) Gi[(size_t)r * Nc * nu + j * nu + cu] = 0.0;
    if (Phi) {
        double* Ph = Phi + ((size_t)b * Nc + i) * NX * NX;
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                double a = 0.0;
#pragma unroll
                for (int l = 0; l < NX; ++l) a += P[r * NX + l] * A[l * NX + c];
                Ph[r * NX + c] = a;
            }
    }
    if (F0) {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < NX; ++l) a += S[r * NX + l] * f0[l];
            F0[((size_t)b * Nc + i) * NX + r] = a;
        }
    }
}

template <int NX>
__global__ void lpv_rk4_kernel(int n, int nu, const double* __restrict__ x, const double* __restrict__ u,
                               const double* __restrict__ A, const double* __restrict__ B, double dt,
                               const double* __restrict__ c, double* __restrict__ xn) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double Am[NX * NX], bu[NX], x0[NX], k[NX], acc[NX], xs[NX];
#pragma unroll
    for (int e = 0; e < NX * NX; ++e) Am[e] = A[(size_t)t * NX * NX + e];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        double a = c[r];
        for (int cu = 0; cu < nu; ++cu) a += B[((size_t)t * NX + r) * nu + cu] * u[(size_t)t * nu + cu];
        bu[r] = a; x0[r] = x[(size_t)t * NX + r]; xs[r] = x0[r]; acc[r] = 0.0;
    }
    const double wgt[4] = {1.0, 2.0, 2.0, 1.0}, adv[3] = {0.5, 0.5, 1.0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double a = bu[r];
#pragma unroll
            for (int l = 0; l < NX; ++l) a += Am[r * NX + l] * xs[l];
            k[r] = dt * a;
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            acc[r] += wgt[st] * k[r];
            if (st < 3) xs[r] = x0[r] + adv[st] * k[r];
        }
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) xn[(size_t)t * NX + r] = x0[r] + acc[r] / 6.0;
}

}  // namespace nmpc

using namespace nmpc;

#define NMPC_NX_SWITCH(nx, CALL)                                                        \
    switch (nx) {                                                                       \
        case 1: CALL(1) break; case 2: CALL(2) break; case 3: CALL(3) break;           \
        case 4: CALL(4) break; case 5: CALL(5) break; case 6: CALL(6) break;           \
        case 7: CALL(7) break; case 8: CALL(8) break;                                   \
        default: set_error("nx=%d unsupported (1..8)", nx); return 1;                   \
    }

extern "C" int nmpc_condense(int nb, int Nc, int nx, int nu, const double* list_A, const double* list_B,
                             const double* f0, double* Phi, double* Gamma, double* F0, void* stream) {
    NMPC_REQUIRE(nb >= 0 && Nc >= 1 && nu >= 1, "nmpc_condense: bad sizes");
    if (nb == 0) return 0;                       // an empty batch may come with null pointers
    NMPC_REQUIRE(list_A, "nmpc_condense: list_A is null");
    NMPC_REQUIRE(!Gamma || list_B, "nmpc_condense: Gamma needs list_B");
    NMPC_REQUIRE(!F0 || f0, "nmpc_condense: F0 needs f0");
    const int threads = 128, grid = (nb * Nc + threads - 1) / threads;
#define CALL(NX_) hipLaunchKernelGGL((condense_kernel<NX_>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, nb, Nc, nu, list_A, list_B, f0, Phi, Gamma, F0);
    NMPC_NX_SWITCH(nx, CALL)
#undef CALL
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int nmpc_lpv_rk4(int n, int nx, int nu, const double* x, const double* u, const double* A,
                            const double* B, double dt, const double* c, double* xnext, void* stream) {
    NMPC_REQUIRE(n >= 0 && nu >= 1, "nmpc_lpv_rk4: bad sizes");
    if (n == 0) return 0;
    NMPC_REQUIRE(x && u && A && B && c && xnext, "nmpc_lpv_rk4: null argument");
    const int threads = 128, grid = (n + threads - 1) / threads;
#define CALL(NX_) hipLaunchKernelGGL((lpv_rk4_kernel<NX_>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, n, nu, x, u, A, B, dt, c, xnext);
    NMPC_NX_SWITCH(nx, CALL)
#undef CALL
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}
