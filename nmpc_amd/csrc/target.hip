// Steady-state target selector (reference: mpcUtil.py:241-320 getXsUs_Cs), batched over
// controllers, sequential over reference samples (each sample is warm-started from the previous
// one, mpcUtil.py:270-271,306-307).
//
// The reference solves, per LPV pass, a QP in (xs, us, e) with the nx+ny equalities
//     (I - A) xs - B us = f0,   C xs = r - h0
// and four soft box rows  -us - e <= -umin, us - e <= umax, -C xs - e <= -ymin, C xs - e <= ymax
// (the right-hand side is used WITHOUT the h0 shift: the (4,1)-(4,) broadcast at mpcUtil.py:254
// followed by h[:,0] at :302 drops it).  With nu = ny = 1 the equalities are nx+1 equations in the
// nx+1 unknowns (xs, us), so they fix (xs, us) and the QP only chooses the slack:
// e = max(0, largest row violation).  That closed form is what the kernel evaluates (Gaussian
// elimination with partial pivoting, fp64); a singular system is flagged in `status`.
#include <atomic>
#include <vector>

#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"

namespace nmpc {

struct TargetArgs {
    int nb, nsamp, max_lpv;
    double tol, umin, umax, ymin, ymax, h0;
    double f0[8];
    // state-constrained variant (mpcUtil.py:167-239 getXsUs): fixed output row Cfix, box on xs itself
    int state;
    double Cfix[8], xmin[8], xmax[8];
    const double* yref;
    double *Xs, *Us, *Es;
    int *iters, *status;
    // workspace
    double *xs, *us;
    float *xs32, *us32, *A32, *B32, *C32;
    int *act, *counts;
};

template <int NX>
__global__ void target_kernel(TargetArgs a, int j, int i) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = i == 0 ? a.nb : min(a.counts[i], a.nb);
    if (gid >= count) return;
    const int b = i == 0 ? gid : a.act[(size_t)(i & 1) * a.nb + gid];
    constexpr int N = NX + 1;
    double M[N][N + 1];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
#pragma unroll
        for (int c = 0; c < NX; ++c) M[r][c] = (r == c ? 1.0 : 0.0) - (double)a.A32[((size_t)b * NX + r) * NX + c];
        M[r][NX] = -(double)a.B32[(size_t)b * NX + r];
        M[r][N] = a.f0[r];
    }
    double Cs[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) { Cs[c] = a.state ? a.Cfix[c] : (double)a.C32[(size_t)b * NX + c]; M[NX][c] = Cs[c]; }
    M[NX][NX] = 0.0;
    M[NX][N] = a.yref[(size_t)b * a.nsamp + j] - a.h0;
    bool singular = false;
#pragma unroll
    for (int col = 0; col < N; ++col) {
        int piv = col; double best = fabs(M[col][col]);
#pragma unroll
        for (int r = col + 1; r < N; ++r) if (fabs(M[r][col]) > best) { best = fabs(M[r][col]); piv = r; }
        if (!(best > 1e-13)) { singular = true; best = 1.0; }
#pragma unroll
        for (int r = col + 1; r < N; ++r)
            if (r == piv) {
#pragma unroll
                for (int c = 0; c <= N; ++c) { const double t = M[col][c]; M[col][c] = M[r][c]; M[r][c] = t; }
            }
        const double inv = 1.0 / (singular && fabs(M[col][col]) <= 1e-13 ? 1.0 : M[col][col]);
#pragma unroll
        for (int r = col + 1; r < N; ++r) {
            const double f = M[r][col] * inv;
#pragma unroll
            for (int c = col; c <= N; ++c) M[r][c] -= f * M[col][c];
        }
    }
    double z[N];
#pragma unroll
    for (int r = N - 1; r >= 0; --r) {
        double v = M[r][N];
#pragma unroll
        for (int c = r + 1; c < N; ++c) v -= M[r][c] * z[c];
        z[r] = v / (fabs(M[r][r]) > 1e-13 ? M[r][r] : 1.0);
    }
    const double usn = z[NX];
    double cy = 0.0, dx = 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        cy += Cs[c] * z[c];
        const double d = z[c] - a.xs[(size_t)b * NX + c];
        dx += d * d;
    }
    const double du = fabs(usn - a.us[b]);
    double v = fmax(-usn + a.umin, usn - a.umax);
    if (a.state) {
#pragma unroll
        for (int c = 0; c < NX; ++c) v = fmax(v, fmax(-z[c] + a.xmin[c], z[c] - a.xmax[c]));
    } else {
        v = fmax(v, fmax(-cy + a.ymin, cy - a.ymax));
    }
    const double e = v > 0.0 ? v : 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        a.xs[(size_t)b * NX + c] = z[c];
        a.xs32[(size_t)b * NX + c] = (float)z[c];
        a.Xs[((size_t)b * a.nsamp + j) * NX + c] = z[c];
    }
    a.us[b] = usn; a.us32[b] = (float)usn;
    a.Us[(size_t)b * a.nsamp + j] = usn;
    a.Es[(size_t)b * a.nsamp + j] = e;
    if (a.iters) a.iters[(size_t)b * a.nsamp + j] = i + 1;
    if (singular) a.status[b] = 1;
    const bool conv = sqrt(dx) <= a.tol && du <= a.tol;          // mpcUtil.py:310
    if (!conv && i + 1 < a.max_lpv) {
        const int pos = atomicAdd(&a.counts[i + 1], 1);
        a.act[(size_t)((i + 1) & 1) * a.nb + pos] = b;
    }
}

}  // namespace nmpc

using namespace nmpc;

// workspace of one call: xs, us (fp64) | xs32 us32 A32 B32 C32 (fp32) | act[2][nb], counts[max passes + 2] (int)
static size_t target_ws_bytes(int nx, int nb, int max_lpv) {
    const size_t nB = (size_t)nb;
    const int passes = max_lpv > 0 ? max_lpv : 2;
    size_t bytes = nB * (nx + 1) * 8 + nB * (size_t)(nx + 1 + nx * nx + nx + nx) * 4 + (2 * nB + passes + 2) * 4;
    return (bytes + 255) & ~(size_t)255;
}

// Issues the whole selection on `stream` into the caller's workspace `ws` (target_ws_bytes, 16-byte aligned);
// allocates nothing and does not synchronise.  max_lpv = 0 selects the pass schedule the reference's code
// executes (see nmpc_target_select in the header): at most two passes for sample 0, one for every later sample.
static int target_issue(const nmpc_model* m, TargetArgs a, const float* lam, const float* w, int n_nodes, char* ws,
                        void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int nx = m->d.nx, nb = a.nb, nsamp = a.nsamp;
    const bool as_executed = a.max_lpv == 0;
    const int max_passes = as_executed ? 2 : a.max_lpv;
    NodeTable raw, nt;
    raw.n = n_nodes;
    for (int i = 0; i < NMPC_MAX_NODES; ++i) { raw.lam[i] = i < n_nodes ? lam[i] : 0.f; raw.w[i] = i < n_nodes ? w[i] : 0.f; }
    if (embed_prepare_nodes(m->d, raw, nt)) return 1;
    const size_t nB = (size_t)nb;
    NMPC_HIP_CHECK(hipMemsetAsync(ws, 0, target_ws_bytes(nx, nb, a.max_lpv), s));
    char* p = ws;
    a.xs = (double*)p; p += nB * nx * 8; a.us = (double*)p; p += nB * 8;
    a.xs32 = (float*)p; p += nB * nx * 4; a.us32 = (float*)p; p += nB * 4;
    a.A32 = (float*)p; p += nB * nx * nx * 4; a.B32 = (float*)p; p += nB * nx * 4; a.C32 = (float*)p; p += nB * nx * 4;
    a.act = (int*)p; p += 2 * nB * 4; a.counts = (int*)p;
    NMPC_HIP_CHECK(hipMemsetAsync(a.status, 0, nB * 4, s));
    const int threads = 64, grid = (nb + threads - 1) / threads;
    for (int j = 0; j < nsamp; ++j) {
        const int lim = as_executed ? (j == 0 ? 2 : 1) : max_passes;
        a.max_lpv = lim;
        if (lim > 1) NMPC_HIP_CHECK(hipMemsetAsync(a.counts, 0, (lim + 1) * 4, s));
        for (int i = 0; i < lim; ++i) {
            EmbedArgs ea{};
            ea.x = a.xs32; ea.u = a.us32; ea.A = a.A32; ea.B = a.B32;
            if (!a.state) { ea.xc = a.xs32; ea.C = a.C32; }
            ea.npts = nb; ea.group = 1;
            if (i > 0) { ea.count_dev = a.counts + i; ea.act = a.act + (size_t)(i & 1) * nb; }
            if (embed_launch_prepared(m->d, ea, nt, s, false)) return 1;
            switch (nx) {
#define NMPC_TARGET_CASE(NX_) case NX_: hipLaunchKernelGGL((target_kernel<NX_>), dim3(grid), dim3(threads), 0, s, a, j, i); break;
                NMPC_TARGET_CASE(1) NMPC_TARGET_CASE(2) NMPC_TARGET_CASE(3) NMPC_TARGET_CASE(4)
                NMPC_TARGET_CASE(5) NMPC_TARGET_CASE(6) NMPC_TARGET_CASE(7) NMPC_TARGET_CASE(8)
#undef NMPC_TARGET_CASE
                default: set_error("nmpc_target_select: nx=%d unsupported (1..8)", nx); return 1;
            }
            NMPC_HIP_CHECK(hipGetLastError());
        }
    }
    return 0;
}

// allocating form: own workspace, synchronises before freeing it
static int target_run(const nmpc_model* m, const TargetArgs& a, const float* lam, const float* w, int n_nodes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    char* slab = nullptr;
    NMPC_HIP_CHECK(hipMalloc((void**)&slab, target_ws_bytes(m->d.nx, a.nb, a.max_lpv)));
    const int rc = target_issue(m, a, lam, w, n_nodes, slab, stream);
    hipError_t e1 = hipStreamSynchronize(s);
    hipError_t e2 = hipGetLastError();
    (void)hipFree(slab);
    if (rc) return 1;
    NMPC_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "nmpc_target_select: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return 0;
}

static int target_fill_output(const nmpc_model* m, TargetArgs& a, int nb, int nsamp, const double* y_ref_n, double u_min,
                              double u_max, double y_min, double y_max, const float* lam, const float* w, int n_nodes,
                              int max_lpv, double tol, double* Xs, double* Us, double* Es, int* iters, int* status) {
    NMPC_REQUIRE(m && y_ref_n && lam && w && Xs && Us && Es && status, "nmpc_target_select: null argument");
    NMPC_REQUIRE(on_device(m->device), "nmpc_target_select: the model lives on device %d, which is not the current device", m->device);
    NMPC_REQUIRE(m->d.nu == 1 && m->d.ny == 1, "nmpc_target_select: needs nu = ny = 1 (as the reference)");
    NMPC_REQUIRE(nb >= 1 && nsamp >= 1 && max_lpv >= 0 && max_lpv <= 1024, "nmpc_target_select: bad sizes");
    NMPC_REQUIRE(n_nodes >= 1 && n_nodes <= NMPC_MAX_NODES, "nmpc_target_select: bad node count");
    a = TargetArgs{};
    a.nb = nb; a.nsamp = nsamp; a.max_lpv = max_lpv; a.tol = tol;
    a.umin = u_min; a.umax = u_max; a.ymin = y_min; a.ymax = y_max; a.h0 = m->d.h0[0];
    for (int r = 0; r < m->d.nx; ++r) a.f0[r] = m->d.f0[r];
    a.yref = y_ref_n; a.Xs = Xs; a.Us = Us; a.Es = Es; a.iters = iters; a.status = status;
    return 0;
}

static int target_fill_state(const nmpc_model* m, TargetArgs& a, int nb, int nsamp, const double* y_ref_n, double u_min,
                             double u_max, const double* x_min, const double* x_max, const double* C, const double* f0,
                             double h0, const float* lam, const float* w, int n_nodes, int max_lpv, double tol, double* Xs,
                             double* Us, double* Es, int* iters, int* status) {
    NMPC_REQUIRE(m && y_ref_n && x_min && x_max && C && lam && w && Xs && Us && Es && status,
                 "nmpc_target_select_state: null argument");
    NMPC_REQUIRE(on_device(m->device), "nmpc_target_select_state: the model lives on device %d, which is not the current device", m->device);
    NMPC_REQUIRE(m->d.nu == 1, "nmpc_target_select_state: needs nu = 1 and one output row (as the reference)");
    NMPC_REQUIRE(nb >= 1 && nsamp >= 1 && max_lpv >= 0 && max_lpv <= 1024, "nmpc_target_select_state: bad sizes");
    NMPC_REQUIRE(n_nodes >= 1 && n_nodes <= NMPC_MAX_NODES, "nmpc_target_select_state: bad node count");
    a = TargetArgs{};
    a.nb = nb; a.nsamp = nsamp; a.max_lpv = max_lpv; a.tol = tol;
    a.umin = u_min; a.umax = u_max; a.h0 = h0; a.state = 1;
    for (int r = 0; r < m->d.nx; ++r) {
        a.f0[r] = f0 ? f0[r] : m->d.f0[r];
        a.Cfix[r] = C[r]; a.xmin[r] = x_min[r]; a.xmax[r] = x_max[r];
    }
    a.yref = y_ref_n; a.Xs = Xs; a.Us = Us; a.Es = Es; a.iters = iters; a.status = status;
    return 0;
}

static int target_check_ws(const nmpc_model* m, const TargetArgs& a, const void* ws, unsigned long long ws_bytes) {
    NMPC_REQUIRE(ws && ((size_t)ws & 15) == 0, "nmpc_target_select: workspace must be a 16-byte aligned device pointer");
    const size_t need = target_ws_bytes(m->d.nx, a.nb, a.max_lpv);
    NMPC_REQUIRE(ws_bytes >= need, "nmpc_target_select: workspace of %llu bytes is too small (%zu needed)", ws_bytes, need);
    return 0;
}

// ---- allocation-free form -----------------------------------------------------------------------
// Scratch of nmpc_target_select_static: part of the code object, so every GPU the library runs on holds
// its own copy at its own address until the process ends.  In host code the array's NAME is only a
// stand-in for the symbol; the device address is looked up once per device and kept in scratch_ptr.
constexpr size_t TARGET_STATIC_BYTES = (size_t)NMPC_TARGET_STATIC_BYTES;
static __device__ __align__(256) unsigned char scratch[TARGET_STATIC_BYTES];
constexpr int TARGET_MAX_DEVICES = 64;
static std::atomic<void*> scratch_ptr[TARGET_MAX_DEVICES];

// device address of `scratch` on the current device (looked up on the first call on that device)
static int target_static_scratch(char** out) {
    int dev = -1;
    NMPC_HIP_CHECK(hipGetDevice(&dev));
    NMPC_REQUIRE(dev >= 0 && dev < TARGET_MAX_DEVICES, "nmpc_target_select_static: device %d out of range", dev);
    void* p = scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        NMPC_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        NMPC_REQUIRE(p != nullptr, "nmpc_target_select_static: no device address for the scratch array");
        scratch_ptr[dev].store(p, std::memory_order_release);     // racing threads store the same address
    }
    *out = (char*)p;
    return 0;
}

extern "C" int nmpc_target_select_static(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                                         double u_max, double y_min, double y_max, const float* lam, const float* w,
                                         int n_nodes, int max_lpv, double tol, double* Xs, double* Us, double* Es,
                                         int* iters, int* status, void* stream) {
    TargetArgs a;
    if (target_fill_output(m, a, nb, nsamp, y_ref_n, u_min, u_max, y_min, y_max, lam, w, n_nodes, max_lpv, tol, Xs, Us, Es,
                           iters, status)) return 1;
    const size_t need = target_ws_bytes(m->d.nx, nb, max_lpv);
    NMPC_REQUIRE(need <= TARGET_STATIC_BYTES,
                 "nmpc_target_select_static: %d controllers with nx=%d need %zu bytes of scratch, the static buffer holds %zu "
                 "(use nmpc_target_select_ws with a caller workspace)", nb, m->d.nx, need, TARGET_STATIC_BYTES);
    char* ws = nullptr;
    if (target_static_scratch(&ws)) return 1;
    // target_issue clears the part of the scratch it uses (it still holds the previous call's contents)
    return target_issue(m, a, lam, w, n_nodes, ws, stream);
}

extern "C" unsigned long long nmpc_target_workspace_bytes(int nx, int nb, int max_lpv) {
    if (nx < 1 || nb < 1 || max_lpv < 0) return 0;
    return (unsigned long long)target_ws_bytes(nx, nb, max_lpv);
}

extern "C" int nmpc_target_select(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                                  double u_max, double y_min, double y_max, const float* lam, const float* w,
                                  int n_nodes, int max_lpv, double tol, double* Xs, double* Us, double* Es,
                                  int* iters, int* status, void* stream) {
    TargetArgs a;
    if (target_fill_output(m, a, nb, nsamp, y_ref_n, u_min, u_max, y_min, y_max, lam, w, n_nodes, max_lpv, tol, Xs, Us, Es,
                           iters, status)) return 1;
    return target_run(m, a, lam, w, n_nodes, stream);
}

extern "C" int nmpc_target_select_ws(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                                     double u_max, double y_min, double y_max, const float* lam, const float* w,
                                     int n_nodes, int max_lpv, double tol, double* Xs, double* Us, double* Es,
                                     int* iters, int* status, void* workspace, unsigned long long workspace_bytes,
                                     void* stream) {
    TargetArgs a;
    if (target_fill_output(m, a, nb, nsamp, y_ref_n, u_min, u_max, y_min, y_max, lam, w, n_nodes, max_lpv, tol, Xs, Us, Es,
                           iters, status)) return 1;
    if (target_check_ws(m, a, workspace, workspace_bytes)) return 1;
    return target_issue(m, a, lam, w, n_nodes, (char*)workspace, stream);
}

extern "C" int nmpc_target_select_state(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                                        double u_max, const double* x_min, const double* x_max, const double* C,
                                        const double* f0, double h0, const float* lam, const float* w, int n_nodes,
                                        int max_lpv, double tol, double* Xs, double* Us, double* Es, int* iters,
                                        int* status, void* stream) {
    TargetArgs a;
    if (target_fill_state(m, a, nb, nsamp, y_ref_n, u_min, u_max, x_min, x_max, C, f0, h0, lam, w, n_nodes, max_lpv, tol, Xs,
                          Us, Es, iters, status)) return 1;
    return target_run(m, a, lam, w, n_nodes, stream);
}

extern "C" int nmpc_target_select_state_ws(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                                           double u_max, const double* x_min, const double* x_max, const double* C,
                                           const double* f0, double h0, const float* lam, const float* w, int n_nodes,
                                           int max_lpv, double tol, double* Xs, double* Us, double* Es, int* iters,
                                           int* status, void* workspace, unsigned long long workspace_bytes,
                                           void* stream) {
    TargetArgs a;
    if (target_fill_state(m, a, nb, nsamp, y_ref_n, u_min, u_max, x_min, x_max, C, f0, h0, lam, w, n_nodes, max_lpv, tol, Xs,
                          Us, Es, iters, status)) return 1;
    if (target_check_ws(m, a, workspace, workspace_bytes)) return 1;
    return target_issue(m, a, lam, w, n_nodes, (char*)workspace, stream);
}
