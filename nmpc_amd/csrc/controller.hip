// Closed-loop LPV-NMPC step (reference: NMPC_test.py:149-315) for a batch of independent
// controllers, and the stand-alone condensed MPC QP (NMPC_test.py:206-238): host side and the
// plant / observer kernels.  Per LPV pass two kernels run: the MFMA embedding (embed.hip,
// embed_wide.hip) over the scheduling points of the still-active controllers, then
// `mpc_solve_kernel` (mpc_kernels.h); `plant_kernel` then advances plant, measurement, observer,
// logs and the warm-start shift.
#include <cstring>
#include <vector>

#include "mpc_kernels.h"
#include "mpc_wide.h"
#include "mpc_tail.h"

namespace nmpc {

// plain-layout net  W_lin z + b_lin + W4 tanh(W2 tanh(W0 z + b0) + b2) + b4  evaluated by one
// wavefront in fp64 (fp32 weights): lane k owns hidden unit k.  `in` (nin values) and the scratch
// `t1s` (64 doubles) live in LDS; the result is returned in every lane.
template <int NOUT>
__device__ void wave_mlp(const float* __restrict__ p, int nin, const double* in, double* t1s, double* out) {
    const int lane = threadIdx.x & 63;
    const float* lin_w = p; const float* lin_b = lin_w + NOUT * nin;
    const float* w0 = lin_b + NOUT; const float* b0 = w0 + HID * nin;
    const float* w2 = b0 + HID; const float* b2 = w2 + HID * HID;
    const float* w4 = b2 + HID; const float* b4 = w4 + NOUT * HID;
    const float* w2t = b4 + NOUT;
    double a = b0[lane];
    for (int j = 0; j < nin; ++j) a += (double)w0[lane * nin + j] * in[j];
    g_sync();
    t1s[lane] = tanh(a);
    g_sync();
    a = b2[lane];
#pragma unroll 8
    for (int k = 0; k < HID; ++k) a += (double)w2t[k * HID + lane] * t1s[k];
    const double t2 = tanh(a);
#pragma unroll
    for (int c = 0; c < NOUT; ++c) {
        double v = g_sum<64>((double)w4[c * HID + lane] * t2);
        v += (double)lin_b[c] + (double)b4[c];
        for (int j = 0; j < nin; ++j) v += (double)lin_w[c * nin + j] * in[j];
        out[c] = v;
    }
}

__device__ __forceinline__ void disc_deriv(const double* p, double w, double th, double u, double& dw, double& dth) {
    // Systems.py:90-94: g, J, Km, I, M, tau = p[0..5]
    dw = -p[4] * p[0] * p[3] / p[1] * sin(th) - 1.0 / p[5] * w + p[2] / p[5] * u;
    dth = w;
}

constexpr int MAX_HIST = 32;
constexpr int MAX_SLICES = 8;     // independent sub-batches of one controller batch, each on its own stream
constexpr int PLANT_WAVES = 4;

// one wavefront per controller
template <int NX>
__global__ __launch_bounds__(64 * PLANT_WAVES) void plant_kernel(CtrlDev d, RunPtrs rp, int k) {
    __shared__ double sm[PLANT_WAVES][MAX_HIST + HID];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * PLANT_WAVES + wv;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int it = 0; it <= d.maxit; ++it) { if (it) d.nact[it] = 0; if (d.nslow) d.nslow[it] = 0; }
    if (b >= d.B) return;
    double* in = sm[wv]; double* t1s = sm[wv] + MAX_HIST;
    const int Nc = d.Nc;
    double* up = d.uplan + (size_t)b * Nc;
    double* xpd = d.xpred + (size_t)b * Nc * NX;
    const double u_n = up[0];
    const double u_den = d.ustd * u_n + d.u0;                       // NMPC_test.py:271
    // plant: one RK4 step of the disc ODE (Systems.py:90-97); every lane computes the same values
    double w = d.xp[2 * b], th = d.xp[2 * b + 1];
    double k1w, k1t, k2w, k2t, k3w, k3t, k4w, k4t;
    disc_deriv(d.plant, w, th, u_den, k1w, k1t); k1w *= d.dt; k1t *= d.dt;
    disc_deriv(d.plant, w + k1w / 2, th + k1t / 2, u_den, k2w, k2t); k2w *= d.dt; k2t *= d.dt;
    disc_deriv(d.plant, w + k2w / 2, th + k2t / 2, u_den, k3w, k3t); k3w *= d.dt; k3t *= d.dt;
    disc_deriv(d.plant, w + k3w, th + k3t, u_den, k4w, k4t); k4w *= d.dt; k4t *= d.dt;
    w += (k1w + 2 * k2w + 2 * k3w + k4w) / 6;
    th += (k1t + 2 * k2t + 2 * k3t + k4t) / 6;
    const double y = th + (rp.noise ? rp.noise[(size_t)k * rp.noise_ld + b] : 0.0);
    const double yn = (y - d.y0) / d.ystd;                          // NMPC_test.py:280
    // observer histories, oldest first (NMPC_test.py:283-289); net input = [u_hist, y_hist]
    double* uh = d.uhist + (size_t)b * d.na;
    double* yh = d.yhist + (size_t)b * d.nyh;
    double hv = 0.0;
    if (lane < d.na) hv = lane + 1 < d.na ? uh[lane + 1] : u_n;
    else if (lane < d.na + d.nyh) { const int j = lane - d.na; hv = j + 1 < d.nyh ? yh[j + 1] : yn; }
    g_sync();
    if (lane < d.na) uh[lane] = hv;
    else if (lane < d.na + d.nyh) yh[lane - d.na] = hv;
    if (lane < d.na + d.nyh) in[lane] = hv;
    g_sync();
    double zest[NX], yest[1];
    wave_mlp<NX>(d.enc, d.n_enc_in, in, t1s, zest);
    g_sync();
    if (lane < NX) in[lane] = zest[lane];
    g_sync();
    wave_mlp<1>(d.hplain, NX, in, t1s, yest);
    const int ns = d.n_sim;
    // warm-start shift (NMPC_test.py:307-310) and scheduling points of the next step's first pass
    double upv = 0.0;
    if (lane < Nc) upv = up[lane];
    const double upn = __shfl(upv, lane + 1 < Nc ? lane + 1 : Nc - 1, 64);
    double xv[2] = {0.0, 0.0};                                       // up to 128 predicted-state entries
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = lane + 64 * t;
        if (e < Nc * NX) xv[t] = xpd[e + NX < Nc * NX ? e + NX : e];
    }
    g_sync();
    if (lane == 0) {
        d.xp[2 * b] = w; d.xp[2 * b + 1] = th;
        d.ymeas[b] = yn;
        if (rp.y_log) rp.y_log[(size_t)b * (ns + 1) + k + 1] = y;
        if (rp.y_est_log) rp.y_est_log[(size_t)b * (ns + 1) + k + 1] = d.ystd * yest[0] + d.y0;
        if (rp.u_log) rp.u_log[(size_t)b * ns + k] = u_den;
        if (rp.e_log) rp.e_log[(size_t)b * ns + k] = d.eps[b];
        if (rp.lpv_counter) rp.lpv_counter[(size_t)b * ns + k] = d.cnt[b];
        if (rp.qp_status_log) rp.qp_status_log[(size_t)b * ns + k] = (signed char)d.qp_worst[b];
        d.qp_worst[b] = 0;
    }
    if (lane < NX) {
        if (rp.x_log) rp.x_log[((size_t)b * (ns + 1) + k + 1) * NX + lane] = zest[lane];
        d.x0n[(size_t)b * NX + lane] = zest[lane];
    }
    if (lane < Nc) {
        if (rp.plan_log) rp.plan_log[((size_t)b * ns + k) * Nc + lane] = upv;
        up[lane] = upn;
        d.us32[(size_t)b * Nc + lane] = (float)upn;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = lane + 64 * t;
        if (e < Nc * NX) {
            xpd[e] = xv[t];
            d.xc32[(size_t)b * Nc * NX + e] = (float)xv[t];
        }
    }
    // x_k of the next first pass: x_0 = zest, x_k = shifted prediction k-1 (taken from registers)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = lane + 64 * t;
        const int ep = e >= NX ? e - NX : 0;
        const double s0 = __shfl(xv[0], ep & 63, 64), s1 = __shfl(xv[1], ep & 63, 64);
        double sv = (ep >> 6) ? s1 : s0;
#pragma unroll
        for (int r = 0; r < NX; ++r) if (r == e) sv = zest[r];
        if (e < Nc * NX) d.xs32[(size_t)b * Nc * NX + e] = (float)sv;
    }
}

// plain-layout net evaluated by one workgroup of 256 threads in fp64 (fp32 weights), hidden width H <= 256: thread k
// owns hidden unit k; `in` (nin values), t1s, t2s (H doubles each) and out (nout <= 8) live in LDS.
__device__ void block_mlp(const float* __restrict__ p, int nin, int nout, int H, const double* in, double* t1s, double* t2s,
                          double* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lin_w = p; const float* lin_b = lin_w + nout * nin;
    const float* w0 = lin_b + nout; const float* b0 = w0 + H * nin;
    const float* w2 = b0 + H; const float* b2 = w2 + H * H;
    const float* w4 = b2 + H; const float* b4 = w4 + nout * H;
    const float* w2t = b4 + nout;
    if (tid < H) {
        double a = b0[tid];
        for (int j = 0; j < nin; ++j) a += (double)w0[tid * nin + j] * in[j];
        t1s[tid] = tanh(a);
    }
    __syncthreads();
    if (tid < H) {
        double a = b2[tid];
#pragma unroll 8
        for (int k = 0; k < H; ++k) a += (double)w2t[k * H + tid] * t1s[k];
        t2s[tid] = tanh(a);
    }
    __syncthreads();
    for (int c = wave; c < nout; c += 4) {
        double part = 0.0;
        for (int k = lane; k < H; k += 64) part += (double)w4[c * H + k] * t2s[k];
        double v = wave_sum(part);
        v += (double)lin_b[c] + (double)b4[c];
        for (int j = 0; j < nin; ++j) v += (double)lin_w[c * nin + j] * in[j];
        if (lane == 0) out[c] = v;
    }
    __syncthreads();
}

// The model as its own plant (no counterpart among the reference's plants, which are ODEs of the disc: Systems.py;
// used for models of other systems, whose plants are not part of the hot path): x+ = f(x, u) and y = h(x+) evaluated
// with the model's nets in fp64, measurement noise added to y, the state handed to the controller as it is.  One
// workgroup per controller; logs, warm-start shift and next scheduling points as in plant_kernel.
template <int NX>
__global__ __launch_bounds__(256) void plant_model_kernel(CtrlDev d, RunPtrs rp, int k) {
    __shared__ double in[16], outv[8], t1s[256], t2s[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    if (b == 0 && tid == 0)
        for (int it = 0; it <= d.maxit; ++it) { if (it) d.nact[it] = 0; if (d.nslow) d.nslow[it] = 0; }
    const int Nc = d.Nc, H = d.hidden, ns = d.n_sim;
    double* up = d.uplan + (size_t)b * Nc;
    double* xpd = d.xpred + (size_t)b * Nc * NX;
    const double u_n = up[0];
    if (tid < NX) in[tid] = d.x0n[(size_t)b * NX + tid];
    if (tid == NX) in[NX] = u_n;
    __syncthreads();
    block_mlp(d.fplain, NX + 1, NX, H, in, t1s, t2s, outv);
    double xn[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) xn[r] = outv[r];
    __syncthreads();
    if (tid < NX) in[tid] = outv[tid];
    __syncthreads();
    block_mlp(d.hplain, NX, 1, H, in, t1s, t2s, outv);
    const double yhat = outv[0];
    const double y = d.ystd * yhat + d.y0 + (rp.noise ? rp.noise[(size_t)k * rp.noise_ld + b] : 0.0);
    const double yn = (y - d.y0) / d.ystd;
    // everything that is shifted is read before anything is written
    const int e = tid;
    const bool one = e < Nc * NX;
    const double xold = one ? xpd[e] : 0.0;
    const double xsh = one ? xpd[e + NX < Nc * NX ? e + NX : e] : 0.0;
    const double upv = e < Nc ? up[e] : 0.0;
    const double upn = e < Nc ? up[e + 1 < Nc ? e + 1 : Nc - 1] : 0.0;
    const double epsv = d.eps[b];
    const int cntv = d.cnt[b];
    __syncthreads();
    if (tid == 0) {
        d.ymeas[b] = yn;
        if (rp.y_log) rp.y_log[(size_t)b * (ns + 1) + k + 1] = y;
        if (rp.y_est_log) rp.y_est_log[(size_t)b * (ns + 1) + k + 1] = d.ystd * yhat + d.y0;
        if (rp.u_log) rp.u_log[(size_t)b * ns + k] = d.ustd * u_n + d.u0;
        if (rp.e_log) rp.e_log[(size_t)b * ns + k] = epsv;
        if (rp.lpv_counter) rp.lpv_counter[(size_t)b * ns + k] = cntv;
        if (rp.qp_status_log) rp.qp_status_log[(size_t)b * ns + k] = (signed char)d.qp_worst[b];
        d.qp_worst[b] = 0;
    }
    if (tid < NX) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) if (r == tid) v = xn[r];
        if (rp.x_log) rp.x_log[((size_t)b * (ns + 1) + k + 1) * NX + tid] = v;
        d.x0n[(size_t)b * NX + tid] = v;
    }
    if (e < Nc) {
        if (rp.plan_log) rp.plan_log[((size_t)b * ns + k) * Nc + e] = upv;
        up[e] = upn;
        d.us32[(size_t)b * Nc + e] = (float)upn;
    }
    if (one) {
        xpd[e] = xsh;
        d.xc32[(size_t)b * Nc * NX + e] = (float)xsh;
        double sv = xold;                 // x_k of the next first pass: x_0 = new state, x_k = shifted prediction k-1
#pragma unroll
        for (int r = 0; r < NX; ++r) if (r == e) sv = xn[r];
        d.xs32[(size_t)b * Nc * NX + e] = (float)sv;
    }
}

// Output-feedback step for the models plant_kernel does not cover (observer / output nets of any hidden width up to
// 256, state dimensions up to 8; reference: NMPC_test.py:262-315, Systems.py:79-97): the same step - disc ODE, noisy
// measurement, observer net on the shifted histories, output estimate, logs, warm-start shift - with one workgroup
// per controller and the nets evaluated block-wide in fp64.
template <int NX>
__global__ __launch_bounds__(256) void plant_obs_block_kernel(CtrlDev d, RunPtrs rp, int k) {
    __shared__ double in[MAX_HIST], outv[8], t1s[256], t2s[256];
    const int tid = threadIdx.x, b = blockIdx.x;
    if (b == 0 && tid == 0)
        for (int it = 0; it <= d.maxit; ++it) { if (it) d.nact[it] = 0; if (d.nslow) d.nslow[it] = 0; }
    const int Nc = d.Nc, H = d.hidden, ns = d.n_sim;
    double* up = d.uplan + (size_t)b * Nc;
    double* xpd = d.xpred + (size_t)b * Nc * NX;
    const double u_n = up[0];
    const double u_den = d.ustd * u_n + d.u0;                       // NMPC_test.py:271
    double w = d.xp[2 * b], th = d.xp[2 * b + 1];                   // every thread integrates the same two states
    double k1w, k1t, k2w, k2t, k3w, k3t, k4w, k4t;
    disc_deriv(d.plant, w, th, u_den, k1w, k1t); k1w *= d.dt; k1t *= d.dt;
    disc_deriv(d.plant, w + k1w / 2, th + k1t / 2, u_den, k2w, k2t); k2w *= d.dt; k2t *= d.dt;
    disc_deriv(d.plant, w + k2w / 2, th + k2t / 2, u_den, k3w, k3t); k3w *= d.dt; k3t *= d.dt;
    disc_deriv(d.plant, w + k3w, th + k3t, u_den, k4w, k4t); k4w *= d.dt; k4t *= d.dt;
    w += (k1w + 2 * k2w + 2 * k3w + k4w) / 6;
    th += (k1t + 2 * k2t + 2 * k3t + k4t) / 6;
    const double y = th + (rp.noise ? rp.noise[(size_t)k * rp.noise_ld + b] : 0.0);
    const double yn = (y - d.y0) / d.ystd;                          // NMPC_test.py:280
    double* uh = d.uhist + (size_t)b * d.na;
    double* yh = d.yhist + (size_t)b * d.nyh;
    double hv = 0.0;                                                // histories, oldest first (NMPC_test.py:283-289)
    if (tid < d.na) hv = tid + 1 < d.na ? uh[tid + 1] : u_n;
    else if (tid < d.na + d.nyh) { const int j = tid - d.na; hv = j + 1 < d.nyh ? yh[j + 1] : yn; }
    // everything that is shifted is read before anything is written
    const int e = tid;
    const bool one = e < Nc * NX;
    const double xold = one ? xpd[e] : 0.0;
    const double xsh = one ? xpd[e + NX < Nc * NX ? e + NX : e] : 0.0;
    const double upv = e < Nc ? up[e] : 0.0;
    const double upn = e < Nc ? up[e + 1 < Nc ? e + 1 : Nc - 1] : 0.0;
    const double epsv = d.eps[b];
    const int cntv = d.cnt[b];
    __syncthreads();
    if (tid < d.na) uh[tid] = hv;
    else if (tid < d.na + d.nyh) yh[tid - d.na] = hv;
    if (tid < d.na + d.nyh) in[tid] = hv;
    __syncthreads();
    block_mlp(d.enc, d.n_enc_in, NX, H, in, t1s, t2s, outv);
    double zest[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) zest[r] = outv[r];
    __syncthreads();
    if (tid < NX) in[tid] = outv[tid];
    __syncthreads();
    block_mlp(d.hplain, NX, 1, H, in, t1s, t2s, outv);
    const double yest = outv[0];
    if (tid == 0) {
        d.xp[2 * b] = w; d.xp[2 * b + 1] = th;
        d.ymeas[b] = yn;
        if (rp.y_log) rp.y_log[(size_t)b * (ns + 1) + k + 1] = y;
        if (rp.y_est_log) rp.y_est_log[(size_t)b * (ns + 1) + k + 1] = d.ystd * yest + d.y0;
        if (rp.u_log) rp.u_log[(size_t)b * ns + k] = u_den;
        if (rp.e_log) rp.e_log[(size_t)b * ns + k] = epsv;
        if (rp.lpv_counter) rp.lpv_counter[(size_t)b * ns + k] = cntv;
        if (rp.qp_status_log) rp.qp_status_log[(size_t)b * ns + k] = (signed char)d.qp_worst[b];
        d.qp_worst[b] = 0;
    }
    if (tid < NX) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) if (r == tid) v = zest[r];
        if (rp.x_log) rp.x_log[((size_t)b * (ns + 1) + k + 1) * NX + tid] = v;
        d.x0n[(size_t)b * NX + tid] = v;
    }
    if (e < Nc) {
        if (rp.plan_log) rp.plan_log[((size_t)b * ns + k) * Nc + e] = upv;
        up[e] = upn;
        d.us32[(size_t)b * Nc + e] = (float)upn;
    }
    if (one) {
        xpd[e] = xsh;
        d.xc32[(size_t)b * Nc * NX + e] = (float)xsh;
        double sv = xold;                 // x_k of the next first pass: x_0 = new estimate, x_k = shifted prediction k-1
#pragma unroll
        for (int r = 0; r < NX; ++r) if (r == e) sv = zest[r];
        d.xs32[(size_t)b * Nc * NX + e] = (float)sv;
    }
}

// State-feedback step (NMPC.py:484-495, 628-640): plant RK4, the measured state (plus noise
// [n_sim, batch, NX]) becomes x0, logs, warm-start shift.  One thread per controller; the disc
// plant has two states, so NX = 2.
template <int NX>
__global__ void plant_state_kernel(CtrlDev d, RunPtrs rp, int k) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0)
        for (int it = 0; it <= d.maxit; ++it) { if (it) d.nact[it] = 0; if (d.nslow) d.nslow[it] = 0; }
    if (b >= d.B) return;
    const int Nc = d.Nc, ns = d.n_sim;
    double* up = d.uplan + (size_t)b * Nc;
    double* xpd = d.xpred + (size_t)b * Nc * NX;
    const double u_den = d.ustd * up[0] + d.u0;
    double w = d.xp[2 * b], th = d.xp[2 * b + 1];
    double k1w, k1t, k2w, k2t, k3w, k3t, k4w, k4t;
    disc_deriv(d.plant, w, th, u_den, k1w, k1t); k1w *= d.dt; k1t *= d.dt;
    disc_deriv(d.plant, w + k1w / 2, th + k1t / 2, u_den, k2w, k2t); k2w *= d.dt; k2t *= d.dt;
    disc_deriv(d.plant, w + k2w / 2, th + k2t / 2, u_den, k3w, k3t); k3w *= d.dt; k3t *= d.dt;
    disc_deriv(d.plant, w + k3w, th + k3t, u_den, k4w, k4t); k4w *= d.dt; k4t *= d.dt;
    w += (k1w + 2 * k2w + 2 * k3w + k4w) / 6;
    th += (k1t + 2 * k2t + 2 * k3t + k4t) / 6;
    d.xp[2 * b] = w; d.xp[2 * b + 1] = th;
    const double xm[2] = {w + (rp.noise ? rp.noise[((size_t)k * rp.noise_ld + b) * NX] : 0.0),
                          th + (rp.noise ? rp.noise[((size_t)k * rp.noise_ld + b) * NX + 1] : 0.0)};
    double x0n[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        x0n[r] = (xm[r] - d.x0v[r]) / d.xstdv[r];
        d.x0n[(size_t)b * NX + r] = x0n[r];
        if (rp.x_log) rp.x_log[((size_t)b * (ns + 1) + k + 1) * NX + r] = xm[r];
    }
    if (rp.u_log) rp.u_log[(size_t)b * ns + k] = u_den;
    if (rp.e_log) rp.e_log[(size_t)b * ns + k] = d.eps[b];
    if (rp.lpv_counter) rp.lpv_counter[(size_t)b * ns + k] = d.cnt[b];
    if (rp.qp_status_log) rp.qp_status_log[(size_t)b * ns + k] = (signed char)d.qp_worst[b];
    d.qp_worst[b] = 0;
    for (int i = 0; i < Nc; ++i) {
        const double uv = up[i];
        if (rp.plan_log) rp.plan_log[((size_t)b * ns + k) * Nc + i] = uv;
        const double un = up[i + 1 < Nc ? i + 1 : Nc - 1];
        up[i] = un;                                   // in order: entry i+1 is still the old value
        d.us32[(size_t)b * Nc + i] = (float)un;
    }
    // predictions x_1..x_Nc shift by one state; scheduling states of the next pass: x0, x_2, .., x_Nc
    for (int e = 0; e < Nc * NX; ++e) {
        const double xv = xpd[e + NX < Nc * NX ? e + NX : e];
        xpd[e] = xv;
    }
    for (int e = 0; e < Nc * NX; ++e) d.xs32[(size_t)b * Nc * NX + e] = e < NX ? (float)x0n[e] : (float)xpd[e - NX];
}

struct State8 { double v[8]; };

template <int NX>
__global__ void reset_kernel(CtrlDev d, double u0n, State8 x0s, double pw, double pth) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {
        for (int i = 0; i < (d.maxit + 2) * MAX_SLICES; ++i) { d.nact[i] = 0; if (d.nslow) d.nslow[i] = 0; }
        d.qp_fail[0] = 0ull; d.qp_fail[1] = 0ull;
    }
    if (b >= d.B) return;
    const double* x0 = x0s.v;
    const int Nc = d.Nc;
    for (int r = 0; r < NX; ++r) d.x0n[(size_t)b * NX + r] = x0[r];
    for (int e = 0; e < Nc * NX; ++e) {
        d.xpred[(size_t)b * Nc * NX + e] = x0[e % NX];
        d.xc32[(size_t)b * Nc * NX + e] = (float)x0[e % NX];
        d.xs32[(size_t)b * Nc * NX + e] = (float)x0[e % NX];
    }
    for (int i = 0; i < Nc; ++i) { d.uplan[(size_t)b * Nc + i] = u0n; d.us32[(size_t)b * Nc + i] = (float)u0n; }
    d.ymeas[b] = 0.0; d.eps[b] = 0.0; d.cnt[b] = 0;
    d.qp_stat[2 * b] = 0u; d.qp_stat[2 * b + 1] = 0u; d.qp_worst[b] = 0;
    if (d.wcnt) d.wcnt[b] = 0;
    d.xp[2 * b] = pw; d.xp[2 * b + 1] = pth;
    for (int j = 0; j < d.na; ++j) d.uhist[(size_t)b * d.na + j] = 0.0;
    for (int j = 0; j < d.nyh; ++j) d.yhist[(size_t)b * d.nyh + j] = 0.0;
}

static int pick_gs(int Nc) {
    int GS = 64;
    if (const char* e = getenv("NMPC_MPC_GS")) {
        const int v = atoi(e);
        if ((v == 16 || v == 64) && Nc + 1 <= v) GS = v;
    }
    return GS;
}

// launch geometry shared by the stand-alone MPC QP and the closed-loop solve kernel
static int mpc_geom(int Nc, int nx, int mode, int slack, MpcGeom& g) {
    NMPC_REQUIRE(Nc >= 1 && Nc + 1 <= 64, "horizon Nc=%d unsupported (1..63)", Nc);
    NMPC_REQUIRE(nx >= 2 && nx <= 8, "nx=%d unsupported by the fused MPC kernels (2..8)", nx);
    g.GS = nx > 4 ? 64 : pick_gs(Nc);
    const int n = mpc_nvar(Nc, slack), mt = mpc_nrows(Nc, nx, mode);
    NMPC_REQUIRE(mt <= 4 * 64, "horizon Nc=%d gives %d constraint rows (at most 256)", Nc, mt);
    const size_t per = sizeof(double) * (size_t)mpc_lds_doubles(Nc, nx, n, mt);
    // (horizons the workgroup-per-controller kernels take never run on the one-wavefront kernel this sizes)
    NMPC_REQUIRE(mpc_wide_applies(Nc, nx, mode, slack) || per * (64 / g.GS) <= 160 * 1024,
                 "horizon Nc=%d needs %zu bytes of LDS per controller: too large", Nc, per);
    const int per_wave = 64 / g.GS;
    int waves = (int)((48 * 1024) / (per * per_wave));
    waves = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
    g.gpb = waves * per_wave;
    g.lds = per * g.gpb;
    g.mpl = (mt + g.GS - 1) / g.GS;
    return 0;
}

}  // namespace nmpc

using namespace nmpc;

static int dispatch_mpc_qp(const MpcQpArgs& a, const MpcGeom& g, hipStream_t s) {
    if (mpc_wide_applies(a.c.Nc, a.nx, a.c.mode, a.c.slack)) return launch_mpc_qp_wide(a, s);
    if (a.nx == 2) return dispatch_mpc_qp_nx2(a, g, s);
    if (a.nx == 3) return dispatch_mpc_qp_nx3(a, g, s);
    if (a.nx == 4) return dispatch_mpc_qp_nx4(a, g, s);
    if (a.nx == 5) return dispatch_mpc_qp_nx5(a, g, s);
    if (a.nx == 6) return dispatch_mpc_qp_nx6(a, g, s);
    if (a.nx == 7) return dispatch_mpc_qp_nx7(a, g, s);
    if (a.nx == 8) return dispatch_mpc_qp_nx8(a, g, s);
    set_error("nmpc_mpc_qp: no kernel for nx=%d", a.nx);
    return 1;
}

static int dispatch_mpc_solve(const CtrlDev& d, const RunPtrs& rp, int it, int k, const MpcGeom& g, hipStream_t s) {
    if (mpc_wide_applies(d.Nc, d.nx, d.c.mode, d.c.slack)) return launch_mpc_solve_wide(d, rp, it, k, s);
    if (d.nx == 2) return dispatch_mpc_solve_nx2(d, rp, it, k, g, s);
    if (d.nx == 3) return dispatch_mpc_solve_nx3(d, rp, it, k, g, s);
    if (d.nx == 4) return dispatch_mpc_solve_nx4(d, rp, it, k, g, s);
    if (d.nx == 5) return dispatch_mpc_solve_nx5(d, rp, it, k, g, s);
    if (d.nx == 6) return dispatch_mpc_solve_nx6(d, rp, it, k, g, s);
    if (d.nx == 7) return dispatch_mpc_solve_nx7(d, rp, it, k, g, s);
    if (d.nx == 8) return dispatch_mpc_solve_nx8(d, rp, it, k, g, s);
    set_error("nmpc_ctrl_run: no kernel for nx=%d", d.nx);
    return 1;
}

static void fill_const(MpcConst& c, int Nc, int nx, const double* Q, double R, double rho, double umin, double umax,
                       double ymin, double ymax, const double* f0, double h0) {
    std::memset(&c, 0, sizeof(c));
    c.Nc = Nc; c.mode = 0; c.slack = 1;
    for (int e = 0; e < nx * nx; ++e) c.Q[e] = Q[e];
    for (int r = 0; r < nx; ++r) c.f0[r] = f0[r];
    c.R = R; c.rho = rho; c.umin = umin; c.umax = umax; c.ymin = ymin; c.ymax = ymax; c.h0 = h0;
}

extern "C" int nmpc_mpc_qp(int nb, int Nc, int nx, const double* list_A, const double* list_B, const double* list_C,
                           const double* x0, const double* y_meas, const double* Xs, const double* Us,
                           const double* f0, double h0, const double* Q, double R, double rho, double u_min,
                           double u_max, double y_min, double y_max, double* Ge, double* Fe, double* Le, double* hv,
                           double* u, double* eps, double* xpred, int* status, void* stream) {
    NMPC_REQUIRE(nb >= 0, "nmpc_mpc_qp: negative batch");
    if (nb == 0) return 0;
    NMPC_REQUIRE(list_A && list_B && list_C && x0 && y_meas && Xs && Us && f0 && Q, "nmpc_mpc_qp: null input");
    MpcGeom g;
    if (mpc_geom(Nc, nx, 0, 1, g)) return 1;
    MpcQpArgs a{};
    a.nb = nb; a.nx = nx; a.lA = list_A; a.lB = list_B; a.lC = list_C; a.x0 = x0; a.ymeas = y_meas; a.Xs = Xs; a.Us = Us;
    a.Ge = Ge; a.Fe = Fe; a.Le = Le; a.hv = hv; a.u = u; a.eps = eps; a.xpred = xpred; a.status = status;
    fill_const(a.c, Nc, nx, Q, R, rho, u_min, u_max, y_min, y_max, f0, h0);
    // diagnostics: NMPC_MPC_STAMPS=<device pointer, hex> receives 16 clock stamps per problem
    if (const char* e = getenv("NMPC_MPC_STAMPS")) {
        void* p = (void*)strtoull(e, nullptr, 16);
        hipPointerAttribute_t at;
        NMPC_REQUIRE(p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice,
                     "nmpc_mpc_qp: NMPC_MPC_STAMPS does not name device memory");
        a.stamps = (unsigned long long*)p;       // the caller provides 16 * nb words
    }
    return dispatch_mpc_qp(a, g, (hipStream_t)stream);
}

struct ProfRec { hipEvent_t a, b; int kind, step; };

// A slice is a contiguous range of the batch with its own stream, work lists and counters: the
// controllers are independent, so slices advance through the step sequence concurrently and the
// latency-bound QP tail of one slice overlaps the issue-bound embedding pass of another.
struct CtrlSlice {
    CtrlDev d;          // view: batch-major pointers advanced to the slice's first controller, B = slice size
    int b0;
    hipStream_t stream; // null when the caller's stream is used (single slice)
    hipEvent_t done;
};

struct nmpc_ctrl {
    CtrlDev d;
    std::vector<CtrlSlice> slices;
    hipEvent_t start = nullptr;
    ModelDev m;
    NodeTable nt;
    MpcGeom g;
    unsigned long long* embed_stamps = nullptr;   // diagnostics (nmpc_ctrl_set_embed_stamps)
    int embed_stamp_slots = 0;
    signed char* qp_status_log = nullptr;   // caller's per-step status log (nmpc_ctrl_set_status_log), may be null
    int tail_from = 0;                // LPV passes from this one on run in the per-controller tail kernel (0: none do)
    std::vector<void*> allocs;
    int device;
    int prof_every = 0;               // 0: off; n: event pairs around the launches of steps k % n == 0
    bool prof_on = false;             // set per step by nmpc_ctrl_run
    std::vector<ProfRec> prof;        // recorded pairs
    std::vector<ProfRec> prof_pool;   // idle pairs
};

static bool prof_begin(nmpc_ctrl* c, int kind, int step, hipStream_t s, ProfRec& r) {
    if (!c->prof_on) return false;
    if (!c->prof_pool.empty()) { r = c->prof_pool.back(); c->prof_pool.pop_back(); }
    else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
    r.kind = kind; r.step = step;
    return hipEventRecord(r.a, s) == hipSuccess;
}
static void prof_end(nmpc_ctrl* c, bool on, hipStream_t s, ProfRec& r) {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    c->prof.push_back(r);
}

extern "C" int nmpc_ctrl_create(nmpc_ctrl** out, const nmpc_model* m, const nmpc_ctrl_config* cfg) {
    NMPC_REQUIRE(out && m && cfg, "nmpc_ctrl_create: null argument");
    NMPC_REQUIRE(cfg->struct_size == (int)sizeof(nmpc_ctrl_config),
                 "nmpc_ctrl_create: cfg->struct_size is %d, this library's nmpc_ctrl_config has %d bytes (caller built against "
                 "another version of nmpc_amd.h?)", cfg->struct_size, (int)sizeof(nmpc_ctrl_config));
    NMPC_REQUIRE(on_device(m->device), "nmpc_ctrl_create: the model lives on device %d, which is not the current device", m->device);
    NMPC_REQUIRE(m->d.nu == 1 && m->d.ny == 1, "nmpc_ctrl_create: closed loop needs nu = ny = 1 (as the reference)");
    const int mode = cfg->mode;
    NMPC_REQUIRE(mode >= 0 && mode <= 2, "nmpc_ctrl_create: mode %d unknown (0 output feedback, 1 state feedback, 2 model as plant)", mode);
    NMPC_REQUIRE(mode != 0 || (m->d.enc && m->d.hplain && m->d.n_enc_in <= MAX_HIST), "nmpc_ctrl_create: model has no observer net");
    NMPC_REQUIRE(mode != 0 || (m->d.nx >= 2 && m->d.nx <= 8), "nmpc_ctrl_create: the observer loop supports nx = 2..8 (model has nx = %d)", m->d.nx);
    NMPC_REQUIRE(mode != 1 || m->d.nx == 2, "nmpc_ctrl_create: state feedback drives the two-state disc plant (nx = 2)");
    NMPC_REQUIRE(mode != 2 || (m->d.fplain && m->d.hplain), "nmpc_ctrl_create: the model-as-plant loop needs nets with two hidden layers, nu = ny = 1");
    {   // the one-wavefront plant step (hidden 64, nx <= 4) shifts 128 predicted-state entries, the block-wide ones 256
        const int lim = (mode == 0 && m->d.hidden == HID && m->d.nx <= 4) ? 128 : 256;
        NMPC_REQUIRE(cfg->Nc * m->d.nx <= lim, "nmpc_ctrl_create: Nc*nx=%d exceeds %d", cfg->Nc * m->d.nx, lim);
    }
    NMPC_REQUIRE(cfg->batch >= 1 && cfg->max_iterations >= 1 && cfg->max_iterations <= 64 && cfg->n_sim >= 1,
                 "nmpc_ctrl_create: bad batch/max_iterations/n_sim");
    NMPC_REQUIRE(cfg->n_nodes >= 1 && cfg->n_nodes <= NMPC_MAX_NODES, "nmpc_ctrl_create: bad node count");
    nmpc_ctrl* c = new nmpc_ctrl();
    const int slack = (mode == 1 && cfg->hard) ? 0 : 1;
    if (mpc_geom(cfg->Nc, m->d.nx, mode == 1 ? 1 : 0, slack, c->g)) { delete c; return 1; }
    c->m = m->d;
    NMPC_HIP_CHECK(hipGetDevice(&c->device));
    CtrlDev& d = c->d;
    std::memset(&d, 0, sizeof(d));
    const int B = cfg->batch, Nc = cfg->Nc, nx = m->d.nx;
    d.B = B; d.Nc = Nc; d.nx = nx; d.maxit = cfg->max_iterations; d.n_sim = cfg->n_sim;
    d.n_enc_in = mode ? 1 : m->d.n_enc_in;
    d.plant_mode = mode; d.hidden = m->d.hidden;
    // observer histories (NMPC_test.py:115-118): n_uhist past inputs then n_yhist past outputs feed the encoder
    // net; 0/0 in the config means the symmetric split nb = na, outputs = na + 1
    d.na = cfg->n_uhist; d.nyh = cfg->n_yhist;
    if (mode != 0) { d.na = 0; d.nyh = 1; }
    else if (d.na == 0 && d.nyh == 0) { d.na = (d.n_enc_in - 1) / 2; d.nyh = d.n_enc_in - d.na; }
    if (mode == 0 && (d.na < 0 || d.nyh < 1 || d.na + d.nyh != d.n_enc_in)) {
        set_error("nmpc_ctrl_create: n_uhist=%d + n_yhist=%d does not match the observer net's %d inputs", d.na, d.nyh, d.n_enc_in);
        delete c;
        return 1;
    }
    d.tol = cfg->tol;
    fill_const(d.c, Nc, nx, cfg->Q, cfg->R, cfg->rho, cfg->u_min, cfg->u_max, cfg->y_min, cfg->y_max, m->d.f0, m->d.h0[0]);
    if (mode == 1) {   // the state-feedback controllers drop the model offset (rhs_c, NMPC.py:387,520)
        d.c.mode = 1; d.c.slack = slack;
        d.c.as_written = cfg->as_written ? 1 : 0; d.us_fixed = d.c.as_written;
        static_assert(sizeof(cfg->x_min) == sizeof(d.c.xmin) && sizeof(cfg->x_norm0) == sizeof(d.x0v), "state bounds / normalisation: config and device constants differ in size");
        for (int r = 0; r < 8; ++r) { d.c.f0[r] = 0.0; d.c.xmin[r] = cfg->x_min[r]; d.c.xmax[r] = cfg->x_max[r]; d.x0v[r] = cfg->x_norm0[r]; d.xstdv[r] = r < nx ? cfg->x_normstd[r] : 1.0; }
    }
    d.u0 = cfg->u0; d.ustd = cfg->ustd; d.y0 = cfg->y0; d.ystd = cfg->ystd; d.dt = cfg->dt;
    for (int i = 0; i < 6; ++i) d.plant[i] = cfg->plant[i];
    d.enc = m->d.enc; d.hplain = m->d.hplain; d.fplain = m->d.fplain;
    {   // the node table every embedding launch of this controller uses, prepared once
        NodeTable raw;
        raw.n = cfg->n_nodes;
        for (int i = 0; i < NMPC_MAX_NODES; ++i) { raw.lam[i] = i < cfg->n_nodes ? cfg->lam[i] : 0.f; raw.w[i] = i < cfg->n_nodes ? cfg->w[i] : 0.f; }
        if (embed_prepare_nodes(c->m, raw, c->nt)) { delete c; return 1; }
    }
    auto alloc = [&](void** p, size_t bytes) -> int {
        NMPC_HIP_CHECK(hipMalloc(p, bytes));
        NMPC_HIP_CHECK(hipMemset(*p, 0, bytes));
        c->allocs.push_back(*p);
        return 0;
    };
    const size_t sb = (size_t)B;
    int rc = 0;
    rc |= alloc((void**)&d.x0n, sb * nx * 8); rc |= alloc((void**)&d.xpred, sb * Nc * nx * 8);
    rc |= alloc((void**)&d.uplan, sb * Nc * 8); rc |= alloc((void**)&d.ymeas, sb * 8);
    rc |= alloc((void**)&d.xp, sb * 2 * 8); rc |= alloc((void**)&d.uhist, sb * (d.na > 0 ? d.na : 1) * 8);
    rc |= alloc((void**)&d.yhist, sb * (d.nyh > 0 ? d.nyh : 1) * 8); rc |= alloc((void**)&d.eps, sb * 8);
    rc |= alloc((void**)&d.cnt, sb * 4); rc |= alloc((void**)&d.act, sb * (d.maxit + 1) * 4);
    rc |= alloc((void**)&d.nact, (size_t)(d.maxit + 2) * MAX_SLICES * 4); rc |= alloc((void**)&d.qp_fail, 16); rc |= alloc((void**)&d.qp_stat, sb * 8);
    rc |= alloc((void**)&d.qp_worst, sb * 4);
    rc |= alloc((void**)&d.xs32, sb * Nc * nx * 4); rc |= alloc((void**)&d.us32, sb * Nc * 4);
    rc |= alloc((void**)&d.xc32, sb * Nc * nx * 4); rc |= alloc((void**)&d.A32, sb * Nc * nx * nx * 4);
    rc |= alloc((void**)&d.B32, sb * Nc * nx * 4); rc |= alloc((void**)&d.C32, sb * Nc * nx * 4);
    if (mpc_wide_applies(Nc, nx, d.c.mode, d.c.slack)) {
        // long-horizon kernels carry each controller's working set from one QP solve to the next
        rc |= alloc((void**)&d.wset, sb * 64); rc |= alloc((void**)&d.wcnt, sb * 4);
        rc |= alloc((void**)&d.slow, sb * (d.maxit + 1) * 4); rc |= alloc((void**)&d.nslow, (size_t)(d.maxit + 2) * MAX_SLICES * 4);
        const char* e = getenv("NMPC_QP_WARM");
        d.warm = e ? atoi(e) : 2;          // 0 cold starts, 1 warm-started iteration, 2 verified fast path first
        if (d.warm < 0 || d.warm > 2) d.warm = 2;
    }
    if (const char* e = getenv("NMPC_MPC_STAMPS")) {      // diagnostics: 16 clock stamps per controller (device memory)
        void* p = (void*)strtoull(e, nullptr, 16);
        hipPointerAttribute_t at;
        if (p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice) d.stamps = (unsigned long long*)p;
    }
    if (rc) { nmpc_ctrl_destroy(c); return 1; }
    // Passes from tail_from on go to one per-controller launch (mpc_tail.hip) where the work list is expected to be
    // almost empty: after two batch-wide passes when at most four passes are allowed (the reference's setting, with
    // its stop tolerance of 0.1); runs that iterate to convergence (more passes allowed) keep batch-wide passes.
    // NMPC_CTRL_TAIL=p overrides (0: never).
    c->tail_from = (d.maxit >= 3 && d.maxit <= 4) ? 2 : 0;
    if (const char* e = getenv("NMPC_CTRL_TAIL")) c->tail_from = atoi(e);
    if (c->tail_from < 1 || c->tail_from >= d.maxit || !mpc_tail_applies(d, c->m, c->g) ||
        mpc_wide_applies(d.Nc, d.nx, d.c.mode, d.c.slack)) c->tail_from = 0;
    // slices: NMPC_CTRL_SLICES (1..8) overrides the default (3 from 1536 controllers, 2 from 1024, else 1;
    // three slice streams and the caller's stream fill the four hardware queues of a process)
    int ns = B >= 1536 ? 3 : (B >= 1024 ? 2 : 1);
    if (const char* e = getenv("NMPC_CTRL_SLICES")) ns = atoi(e);
    ns = ns < 1 ? 1 : (ns > MAX_SLICES ? MAX_SLICES : ns);
    if (ns > B) ns = B;
    const int ntot = d.n_sim + Nc;
    (void)ntot;
    for (int i = 0; i < ns; ++i) {
        CtrlSlice sl;
        sl.d = d; sl.stream = nullptr; sl.done = nullptr;
        const int b0 = (int)((long long)B * i / ns), b1 = (int)((long long)B * (i + 1) / ns);
        const size_t o = (size_t)b0;
        sl.b0 = b0; sl.d.B = b1 - b0;
        sl.d.x0n += o * nx; sl.d.xpred += o * Nc * nx; sl.d.uplan += o * Nc; sl.d.ymeas += o; sl.d.xp += o * 2;
        sl.d.uhist += o * d.na; sl.d.yhist += o * d.nyh; sl.d.eps += o; sl.d.cnt += o;
        sl.d.act += o * (d.maxit + 1); sl.d.nact += (size_t)(d.maxit + 2) * i; sl.d.qp_stat += o * 2; sl.d.qp_worst += o;
        sl.d.xs32 += o * Nc * nx; sl.d.us32 += o * Nc; sl.d.xc32 += o * Nc * nx;
        sl.d.A32 += o * Nc * nx * nx; sl.d.B32 += o * Nc * nx; sl.d.C32 += o * Nc * nx;
        if (d.wset) { sl.d.wset += o * 64; sl.d.wcnt += o; }
        if (d.slow) { sl.d.slow += o * (d.maxit + 1); sl.d.nslow += (size_t)(d.maxit + 2) * i; }
        if (d.stamps) sl.d.stamps += o * 16;
        if (ns > 1) {
            if (hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
                set_error("nmpc_ctrl_create: cannot create slice stream");
                c->slices.push_back(sl);
                nmpc_ctrl_destroy(c);
                return 1;
            }
        }
        c->slices.push_back(sl);
    }
    if (ns > 1 && hipEventCreateWithFlags(&c->start, hipEventDisableTiming) != hipSuccess) {
        set_error("nmpc_ctrl_create: cannot create event");
        nmpc_ctrl_destroy(c);
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" void nmpc_ctrl_destroy(nmpc_ctrl* c) {
    if (!c) return;
    for (auto* v : {&c->prof, &c->prof_pool})
        for (ProfRec& r : *v) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (CtrlSlice& sl : c->slices) {
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (c->start) (void)hipEventDestroy(c->start);
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

extern "C" int nmpc_ctrl_reset(nmpc_ctrl* c, const double* x0n, double u0n, const double* plant_x0, void* stream) {
    NMPC_REQUIRE(c && x0n, "nmpc_ctrl_reset: null argument");
    NMPC_REQUIRE(on_device(c->device), "nmpc_ctrl_reset: the controllers live on device %d, which is not the current device", c->device);
    State8 x0{};
    for (int r = 0; r < c->d.nx; ++r) x0.v[r] = x0n[r];
    const double pw = plant_x0 ? plant_x0[0] : 0.0, pth = plant_x0 ? plant_x0[1] : 0.0;
    const int threads = 128, grid = (c->d.B + threads - 1) / threads;
    hipStream_t st = (hipStream_t)stream;
    switch (c->d.nx) {
#define NMPC_RESET_CASE(NX_) case NX_: hipLaunchKernelGGL((reset_kernel<NX_>), dim3(grid), dim3(threads), 0, st, c->d, u0n, x0, pw, pth); break;
        NMPC_RESET_CASE(2) NMPC_RESET_CASE(3) NMPC_RESET_CASE(4) NMPC_RESET_CASE(5) NMPC_RESET_CASE(6) NMPC_RESET_CASE(7) NMPC_RESET_CASE(8)
#undef NMPC_RESET_CASE
        default: set_error("nmpc_ctrl_reset: nx=%d unsupported", c->d.nx); return 1;
    }
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int nmpc_ctrl_run(nmpc_ctrl* c, int k0, int nsteps, const double* Xs, const double* Us, const double* noise,
                             double* x_log, double* u_log, double* e_log, double* y_log, double* y_est_log,
                             int* lpv_counter, double* plan_log, void* stream) {
    NMPC_REQUIRE(c && Xs && Us, "nmpc_ctrl_run: null argument");
    NMPC_REQUIRE(on_device(c->device), "nmpc_ctrl_run: the controllers live on device %d, which is not the current device", c->device);
    NMPC_REQUIRE(k0 >= 0 && nsteps >= 0 && k0 + nsteps <= c->d.n_sim, "nmpc_ctrl_run: steps [%d,%d) outside n_sim=%d", k0,
                 k0 + nsteps, c->d.n_sim);
    std::vector<RunPtrs> sl_rp;
    hipStream_t s0 = (hipStream_t)stream;
    const int ns = c->d.n_sim, Nc = c->d.Nc, nx = c->d.nx, ntot = ns + Nc;
    const bool multi = c->slices.size() > 1;
    if (multi) NMPC_HIP_CHECK(hipEventRecord(c->start, s0));
    for (CtrlSlice& sl : c->slices) {
        hipStream_t s = multi ? sl.stream : s0;
        if (multi) NMPC_HIP_CHECK(hipStreamWaitEvent(s, c->start, 0));
        const CtrlDev& d = sl.d;
        const size_t o = (size_t)sl.b0;
        RunPtrs rp{};
        rp.Xs = Xs + o * ntot * nx; rp.Us = Us + o * ntot; rp.noise = noise ? noise + o * (c->d.plant_mode == 1 ? nx : 1) : nullptr; rp.noise_ld = c->d.B;
        rp.x_log = x_log ? x_log + o * (ns + 1) * nx : nullptr; rp.u_log = u_log ? u_log + o * ns : nullptr;
        rp.e_log = e_log ? e_log + o * ns : nullptr; rp.y_log = y_log ? y_log + o * (ns + 1) : nullptr;
        rp.y_est_log = y_est_log ? y_est_log + o * (ns + 1) : nullptr;
        rp.plan_log = plan_log ? plan_log + o * ns * Nc : nullptr;
        rp.lpv_counter = lpv_counter ? lpv_counter + o * ns : nullptr;
        rp.qp_status_log = c->qp_status_log ? c->qp_status_log + o * ns : nullptr;
        sl_rp.push_back(rp);
        (void)d; (void)s;
    }
    // steps are issued slice by slice within a step so that the streams stay evenly fed; a failed launch
    // stops the issue, and the slice streams are joined to the caller's stream in either case
    auto issue = [&]() -> int {
    for (int k = k0; k < k0 + nsteps; ++k) {
        c->prof_on = c->prof_every > 0 && k % c->prof_every == 0;
        for (size_t i = 0; i < c->slices.size(); ++i) {
            CtrlSlice& sl = c->slices[i];
            hipStream_t s = multi ? sl.stream : s0;
            const CtrlDev& d = sl.d;
            const RunPtrs& rp = sl_rp[i];
            const int wide_passes = c->tail_from > 0 ? c->tail_from : d.maxit;
            for (int it = 0; it < wide_passes; ++it) {
                EmbedArgs a{};
                a.x = d.xs32; a.u = d.us32; a.A = d.A32; a.B = d.B32;
                if (d.c.mode == 0) { a.xc = d.xc32; a.C = d.C32; }
                a.npts = d.B * d.Nc; a.group = d.Nc;
                if (it > 0) { a.count_dev = d.nact + it; a.act = d.act + (size_t)it * d.B; }
                else if (c->embed_stamps) { a.stamps = c->embed_stamps; a.stamp_slots = c->embed_stamp_slots; }   // first passes only
                ProfRec pr;
                bool on = prof_begin(c, 0, k, s, pr);
                if (embed_launch_prepared(c->m, a, c->nt, s, true)) return 1;
                prof_end(c, on, s, pr);
                on = prof_begin(c, 1, k, s, pr);
                if (dispatch_mpc_solve(d, rp, it, k, c->g, s)) return 1;
                prof_end(c, on, s, pr);
            }
            if (c->tail_from > 0) {
                ProfRec pr;
                const bool on = prof_begin(c, 1, k, s, pr);          // counted with the solves (mostly an empty launch)
                if (launch_mpc_tail(d, rp, c->m, c->nt, c->tail_from, k, s)) return 1;
                prof_end(c, on, s, pr);
            }
            ProfRec pr;
            const bool on = prof_begin(c, 2, k, s, pr);
            const int threads = 64 * PLANT_WAVES, grid = (d.B + PLANT_WAVES - 1) / PLANT_WAVES;
            if (d.plant_mode == 2) {
                switch (d.nx) {
#define NMPC_PM_CASE(NX_) case NX_: hipLaunchKernelGGL((plant_model_kernel<NX_>), dim3(d.B), dim3(256), 0, s, d, rp, k); break;
                    NMPC_PM_CASE(2) NMPC_PM_CASE(3) NMPC_PM_CASE(4) NMPC_PM_CASE(5) NMPC_PM_CASE(6) NMPC_PM_CASE(7) NMPC_PM_CASE(8)
#undef NMPC_PM_CASE
                    default: set_error("nmpc_ctrl_run: nx=%d unsupported", d.nx); return 1;
                }
            }
            else if (d.c.mode == 1) hipLaunchKernelGGL((plant_state_kernel<2>), dim3((d.B + 127) / 128), dim3(128), 0, s, d, rp, k);
            else if (d.hidden != HID || d.nx > 4) {
                switch (d.nx) {
#define NMPC_PO_CASE(NX_) case NX_: hipLaunchKernelGGL((plant_obs_block_kernel<NX_>), dim3(d.B), dim3(256), 0, s, d, rp, k); break;
                    NMPC_PO_CASE(2) NMPC_PO_CASE(3) NMPC_PO_CASE(4) NMPC_PO_CASE(5) NMPC_PO_CASE(6) NMPC_PO_CASE(7) NMPC_PO_CASE(8)
#undef NMPC_PO_CASE
                    default: set_error("nmpc_ctrl_run: nx=%d unsupported", d.nx); return 1;
                }
            }
            else if (d.nx == 2) hipLaunchKernelGGL((plant_kernel<2>), dim3(grid), dim3(threads), 0, s, d, rp, k);
            else if (d.nx == 3) hipLaunchKernelGGL((plant_kernel<3>), dim3(grid), dim3(threads), 0, s, d, rp, k);
            else if (d.nx == 4) hipLaunchKernelGGL((plant_kernel<4>), dim3(grid), dim3(threads), 0, s, d, rp, k);
            else { set_error("nmpc_ctrl_run: nx=%d unsupported", d.nx); return 1; }
            NMPC_HIP_CHECK(hipGetLastError());
            prof_end(c, on, s, pr);
        }
    }
    return 0;
    };
    const int rc = issue();
    if (multi)
        for (CtrlSlice& sl : c->slices) {
            NMPC_HIP_CHECK(hipEventRecord(sl.done, sl.stream));
            NMPC_HIP_CHECK(hipStreamWaitEvent(s0, sl.done, 0));
        }
    return rc;
}

extern "C" int nmpc_ctrl_set_status_log(nmpc_ctrl* c, signed char* qp_status_log) {
    NMPC_REQUIRE(c, "nmpc_ctrl_set_status_log: null argument");
    c->qp_status_log = qp_status_log;
    return 0;
}

extern "C" int nmpc_ctrl_set_embed_stamps(nmpc_ctrl* c, unsigned long long* stamps, int slots) {
    NMPC_REQUIRE(c && (stamps == nullptr || slots >= 1), "nmpc_ctrl_set_embed_stamps: bad argument");
    c->embed_stamps = stamps; c->embed_stamp_slots = stamps ? slots : 0;
    return 0;
}

extern "C" int nmpc_ctrl_profile(nmpc_ctrl* c, int enable) {
    NMPC_REQUIRE(c, "nmpc_ctrl_profile: null argument");
    c->prof_every = enable > 0 ? enable : 0;
    return 0;
}

extern "C" int nmpc_ctrl_tail_from(const nmpc_ctrl* c) { return c ? c->tail_from : 0; }

extern "C" int nmpc_ctrl_profile_read(nmpc_ctrl* c, int* launches, double* total_ms, void* stream) {
    NMPC_REQUIRE(c && launches && total_ms, "nmpc_ctrl_profile_read: null argument");
    NMPC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < 3; ++i) { launches[i] = 0; total_ms[i] = 0.0; }
    for (ProfRec& r : c->prof) {
        float ms = 0.f;
        NMPC_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        launches[r.kind] += 1; total_ms[r.kind] += ms;
        c->prof_pool.push_back(r);
    }
    c->prof.clear();
    return 0;
}

extern "C" int nmpc_ctrl_profile_read_steps(nmpc_ctrl* c, int k0, int nsteps, double* ms, void* stream) {
    NMPC_REQUIRE(c && ms && nsteps >= 0, "nmpc_ctrl_profile_read_steps: bad argument");
    NMPC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < 3 * nsteps; ++i) ms[i] = 0.0;
    for (ProfRec& r : c->prof) {
        float t = 0.f;
        NMPC_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        if (r.step >= k0 && r.step < k0 + nsteps) ms[3 * (r.step - k0) + r.kind] += t;
        c->prof_pool.push_back(r);
    }
    c->prof.clear();
    return 0;
}

extern "C" int nmpc_ctrl_stats(nmpc_ctrl* c, long long* qp_fail, void* stream) {
    NMPC_REQUIRE(c && qp_fail, "nmpc_ctrl_stats: null argument");
    unsigned long long v[2] = {0, 0};
    std::vector<unsigned> st(2 * (size_t)c->d.B);
    NMPC_HIP_CHECK(hipMemcpyAsync(v, c->d.qp_fail, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    NMPC_HIP_CHECK(hipMemcpyAsync(st.data(), c->d.qp_stat, st.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    NMPC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    qp_fail[0] = (long long)v[0]; qp_fail[1] = 0; qp_fail[2] = 0; qp_fail[3] = (long long)v[1];
    for (int b = 0; b < c->d.B; ++b) { qp_fail[1] += st[2 * b]; qp_fail[2] += st[2 * b + 1]; }
    return 0;
}
