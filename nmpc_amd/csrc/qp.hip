// Batched dense QP entry point: one problem per lane group, data staged global -> LDS, solved by
// the dual active-set iteration of qp_gi.h.
#include <cstdlib>

#include "../../include/nmpc_amd.h"
#include "common.h"
#include "qp_gi.h"

namespace nmpc {

struct QPArgs {
    int nb, n, m, meq;
    const double *P, *q, *G, *h, *Aeq, *beq;
    double *z, *lambda;
    int *status, *iters;
};

template <int GS, int MPL>
__global__ void qp_kernel(QPArgs a) {
    extern __shared__ __align__(16) double qp_lds[];
    const int gpb = blockDim.x / GS;
    const int g = threadIdx.x / GS, gl = threadIdx.x % GS;
    const int prob = blockIdx.x * gpb + g;
    if (prob >= a.nb) return;
    const int n = a.n, mt = a.m + a.meq, meq = a.meq;
    const int per = qp_lds_doubles(n, mt) + mt;
    QPWork w;
    w.carve(qp_lds + (size_t)g * per, n, mt);
    double* lam = qp_lds + (size_t)g * per + qp_lds_doubles(n, mt);
    const double* P = a.P + (size_t)prob * n * n;
    for (int e = gl; e < n * n; e += GS) w.Lm[e] = P[e];
    for (int e = gl; e < n; e += GS) w.xt[e] = a.q[(size_t)prob * n + e];
    for (int e = gl; e < meq * n; e += GS) {
        const int i = e / n, k = e % n;
        w.Nt[k * mt + i] = a.Aeq[(size_t)prob * meq * n + e];
    }
    for (int e = gl; e < a.m * n; e += GS) {
        const int i = e / n, k = e % n;
        w.Nt[k * mt + meq + i] = a.G[(size_t)prob * a.m * n + e];
    }
    for (int e = gl; e < meq; e += GS) w.hv[e] = a.beq[(size_t)prob * meq + e];
    for (int e = gl; e < a.m; e += GS) w.hv[meq + e] = a.h[(size_t)prob * a.m + e];
    int it = 0;
    const int st = (GS == 64 && n <= 16) ? gi_solve<GS, MPL, (GS == 64 ? 16 : 0)>(n, mt, meq, w, lam, &it)
                                         : gi_solve<GS, MPL>(n, mt, meq, w, lam, &it);
    for (int e = gl; e < n; e += GS) a.z[(size_t)prob * n + e] = w.xt[e];
    if (a.lambda)
        for (int e = gl; e < mt; e += GS) a.lambda[(size_t)prob * mt + e] = lam[e];
    if (gl == 0) {
        a.status[prob] = st;
        if (a.iters) a.iters[prob] = it;
    }
}

template <int GS, int MPL>
static int qp_launch_t(const QPArgs& a, int gpb, size_t lds_bytes, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_set{0};
    if (first_use_on_device(attr_set)) {
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)qp_kernel<GS, MPL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mark_used_on_device(attr_set);
    }
    const int grid = (a.nb + gpb - 1) / gpb;
    hipLaunchKernelGGL((qp_kernel<GS, MPL>), dim3(grid), dim3(gpb * GS), lds_bytes, stream, a);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

int qp_launch(const QPArgs& a, hipStream_t stream) {
    const int n = a.n, mt = a.m + a.meq;
    int GS = n <= 16 ? 16 : 64;
    if (const char* e = getenv("NMPC_QP_GS")) {
        const int v = atoi(e);
        if ((v == 16 || v == 64) && n <= v) GS = v;
    }
    const size_t per_bytes = sizeof(double) * (size_t)(qp_lds_doubles(n, mt) + mt);
    if (GS == 16 && (mt > 64 || per_bytes * 4 > 160 * 1024)) GS = 64;
    NMPC_REQUIRE(n <= 64, "qp: n=%d exceeds 64 variables", n);
    NMPC_REQUIRE(mt <= 256, "qp: %d constraint rows exceed 256", mt);
    NMPC_REQUIRE(per_bytes <= 160 * 1024, "qp: problem of %zu bytes does not fit in LDS", per_bytes);
    const int per_wave = 64 / GS;
    int waves = (int)((64 * 1024) / (per_bytes * per_wave));   // aim at <= 64 KB per workgroup
    waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
    const int gpb = waves * per_wave;
    const size_t lds_bytes = per_bytes * gpb;
    const int mpl = (mt + GS - 1) / GS;
#define NMPC_QP_CASE(GS_, MPL_) \
    if (GS == GS_ && mpl <= MPL_) return qp_launch_t<GS_, MPL_>(a, gpb, lds_bytes, stream);
    NMPC_QP_CASE(16, 1) NMPC_QP_CASE(16, 2) NMPC_QP_CASE(16, 3) NMPC_QP_CASE(16, 4)
    NMPC_QP_CASE(64, 1) NMPC_QP_CASE(64, 2) NMPC_QP_CASE(64, 4)
#undef NMPC_QP_CASE
    set_error("qp: no kernel for n=%d rows=%d", n, mt);
    return 1;
}

}  // namespace nmpc

using namespace nmpc;

extern "C" int nmpc_qp_solve(int nb, int n, int m, int meq, const double* P, const double* q, const double* G,
                             const double* h, const double* Aeq, const double* beq, double* z, double* lambda,
                             int* status, int* iters, void* stream) {
    NMPC_REQUIRE(nb >= 0 && n >= 1 && m >= 0 && meq >= 0, "nmpc_qp_solve: bad sizes");
    if (nb == 0) return 0;                       // an empty batch may come with null pointers
    NMPC_REQUIRE(P && q && z && status, "nmpc_qp_solve: null argument");
    NMPC_REQUIRE(m == 0 || (G && h), "nmpc_qp_solve: G/h missing");
    NMPC_REQUIRE(meq == 0 || (Aeq && beq), "nmpc_qp_solve: Aeq/beq missing");
    NMPC_REQUIRE(meq <= n, "nmpc_qp_solve: more equalities than variables");
    QPArgs a{nb, n, m, meq, P, q, G, h, Aeq, beq, z, lambda, status, iters};
    return qp_launch(a, (hipStream_t)stream);
}
