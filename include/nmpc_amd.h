/*
 * nmpc_amd - C ABI of the MI355X (gfx950) implementation of the LPV-NMPC inner loop.
 *
 * The reference (Mixxxxx358/NMPC) has no FFI: its hot path is a set of plain Python functions
 * star-imported into the controller scripts (NMPC.py:10-14).  Each entry point below names the
 * reference function(s) whose work it performs; the Python package `nmpc_amd` re-exports the
 * reference's function names on top of these through ctypes.
 *
 * Conventions
 *   - every array argument is CALLER-OWNED DEVICE memory, contiguous, row-major, batch-major,
 *     unless its comment says "host";
 *   - fp32 for the network/embedding arrays, fp64 for condensing / QP / controller state;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     asynchronous on it; no entry point synchronises the device unless stated;
 *   - return value 0 = ok, non-zero = error, message via nmpc_last_error() (thread-local);
 *   - nothing here allocates device memory per call except the *_create functions and the allocating
 *     forms of the pre-loop target selectors (which say so; their _ws forms take a caller workspace);
 *   - a model or controller handle is used with the device that was current at its creation
 *     current again (one process per GPU needs no care; other callers get an error otherwise).
 */
#ifndef NMPC_AMD_H
#define NMPC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmpc_model nmpc_model;
typedef struct nmpc_ctrl nmpc_ctrl;

#define NMPC_MAX_NODES 256

/* quadrature rules of the reference's lpv_int.py / par_lpv_int.py */
enum { NMPC_RULE_SIMPSON = 0, NMPC_RULE_TRAP = 1 };

/* Version of this header; nmpc_version() returns the library's.  Bumped whenever a struct layout or a signature
 * changes (110: nmpc_ctrl_config gained struct_size / as_written and eight-entry state bounds, status log added). */
#define NMPC_AMD_VERSION 110

const char* nmpc_last_error(void);
int nmpc_version(void);

/* ---- model: the learned state-space nets (reference: mpcUtil.py:6-82 CasADiExp/Fn/Hn) -------
 * Each net is  W_lin z + b_lin + W4 tanh(W2 tanh(W0 z + b0) + b2) + b4  with `hidden` units.
 * params = 8 HOST pointers in the reference's parameter order (mpcUtil.py:18-29):
 *   [0] W_lin[out,in] [1] b_lin[out] [2] W0[hidden,in] [3] b0 [4] W2[hidden,hidden] [5] b2
 *   [6] W4[out,hidden] [7] b4.
 * fn: in = nx+nu, out = nx.  hn: in = nx, out = ny.  enc (observer, may be NULL): in = n_enc_in,
 * out = nx.  The weights are repacked into device memory owned by the model. */
int nmpc_model_create(nmpc_model** out, int nx, int nu, int ny, int hidden, int n_enc_in,
                      const float* const* fn_params, const float* const* hn_params,
                      const float* const* enc_params);
/* Nets with another number of hidden layers (reference: mpcUtil.py:6-31 CasADiExp loops over
 * f_n_hidden_layers; called from NMPC.py:517):  W_lin z + b_lin + W_out t_L + b_out,  t_l = tanh(W_l t_{l-1} + b_l),
 * t_0 = z.  params = 4 + 2 L HOST pointers in CasADiExp's order: [0] W_lin [1] b_lin, then for hidden layer
 * i = 0..L-1 [2+2i] W[hidden, in or hidden] [3+2i] b[hidden], then [2+2L] W_out[out,hidden] [3+2L] b_out.
 * 1..4 hidden layers per net (more than two: hidden <= 128); the observer net, when given, has two. */
int nmpc_model_create_layers(nmpc_model** out, int nx, int nu, int ny, int hidden, int f_hidden_layers,
                             int h_hidden_layers, int n_enc_in, const float* const* fn_params,
                             const float* const* hn_params, const float* const* enc_params);
void nmpc_model_destroy(nmpc_model* m);

/* ---- LPV embedding (reference: lpv_int.py:4-100 lpv_int / lpv_int_C / lambda_simpson /
 * lambda_trap, par_lpv_int.py:4-35, mpcUtil.py:136-165 getABlist/getABClist/getClist and the
 * functorch variant NMPC_test.py:129-201).
 *   A[p] = sum_j w_j dfdx(lam_j x[p], lam_j u[p]),  B[p] likewise with dfdu,
 *   C[p] = sum_j w_j dhdx(lam_j xc[p]).
 * x[npts,nx], u[npts,nu], xc[npts,nx] (device fp32).  A[npts,nx,nx], B[npts,nx,nu], C[npts,ny,nx].
 * lam/w: HOST arrays of n_nodes <= NMPC_MAX_NODES quadrature nodes and weights.
 * x,u,A,B may all be NULL (C only); xc,C may both be NULL (A,B only). */
int nmpc_lpv_embed(const nmpc_model* m, int npts, const float* x, const float* u, const float* xc,
                   const float* lam, const float* w, int n_nodes, float* A, float* B, float* C,
                   void* stream);

/* Node table of a rule exactly as the reference enumerates it, bit-identical nodes merged
 * (lam accumulates by repeated addition of dlam, lpv_int.py:30).  HOST outputs, capacity
 * NMPC_MAX_NODES each; returns the node count or -1.  dlam <= 0 means 1/stages. */
int nmpc_quad_nodes(int rule, int stages, double dlam, float* lam, float* w);

/* ---- LPV RK4 predictor (reference: lpv_rk4.py:1-11).  x[n,nx] u[n,nu] A[n,nx,nx] B[n,nx,nu]
 * c[nx] (device, fp64), xnext[n,nx]. */
int nmpc_lpv_rk4(int n, int nx, int nu, const double* x, const double* u, const double* A,
                 const double* B, double dt, const double* c, double* xnext, void* stream);

/* ---- condensed prediction matrices (reference: mpcUtil.py:84-104 getPhi/getGamma,
 * :322-331 getF0).  list_A[nb,Nc*nx,nx], list_B[nb,Nc*nx,nu], f0[nx] (device fp64).
 * Phi[nb,Nc*nx,nx], Gamma[nb,Nc*nx,Nc*nu], F0[nb,Nc*nx]; any output may be NULL. */
int nmpc_condense(int nb, int Nc, int nx, int nu, const double* list_A, const double* list_B,
                  const double* f0, double* Phi, double* Gamma, double* F0, void* stream);

/* ---- batched dense QP (reference: qpsolvers.solve_qp call sites NMPC_test.py:233,
 * NMPC.py:458,602,767, mpcUtil.py:224,302):  min 1/2 z'Pz + q'z  s.t. G z <= h, A z = b.
 * P[nb,n,n] (positive definite), q[nb,n], G[nb,m,n], h[nb,m], Aeq[nb,meq,n], beq[nb,meq]
 * (device fp64; G/h and Aeq/beq may be NULL with m/meq = 0).  z[nb,n]; lambda[nb,meq+m]
 * (multipliers, equalities first; may be NULL); status[nb] (0 ok, 1 iteration limit,
 * 2 infeasible, 3 P not positive definite); iters[nb] (may be NULL). */
int nmpc_qp_solve(int nb, int n, int m, int meq, const double* P, const double* q, const double* G,
                  const double* h, const double* Aeq, const double* beq, double* z, double* lambda,
                  int* status, int* iters, void* stream);

/* ---- one condensed MPC QP of the output-constrained controller (reference: NMPC_test.py:206-238):
 * F0/Phi/Gamma/Z, G = 2(Psi + Gamma' Omega Gamma), F = 2(Gamma' Omega (Phi x0 + F0 - Xs) - Psi Us),
 * L = M Z Gamma + E with one slack column of -1, h = c - D y - M Z (Phi x0 + F0) - M H0
 * (getDEMc_out rows, mpcUtil.py:340-362), solve, x = Phi x0 + Gamma u + F0.   nu = ny = 1.
 * Inputs (device fp64): list_A[nb,Nc*nx,nx] list_B[nb,Nc*nx,1] list_C[nb,Nc,nx] x0[nb,nx] y_meas[nb]
 * Xs[nb,Nc*nx] Us[nb,Nc]; host: f0[nx] h0 Q[nx*nx] R rho and NORMALISED bounds.
 * Outputs (device fp64, each may be NULL): Ge[nb,n,n] (lower triangle valid) Fe[nb,n] Le[nb,m,n]
 * hv[nb,m] with n = Nc+1, m = 4Nc+2;  u[nb,Nc] eps[nb] xpred[nb,Nc*nx] status[nb] (int).
 * Diagnostics: with the environment variable NMPC_MPC_STAMPS set to the address (hex) of a device buffer of
 * 16 * nb 64-bit words (16 * batch for nmpc_ctrl_create, which reads it too), the kernels store 16 shader-clock
 * stamps per problem there; the buffer must stay valid while the handle / call is in use. */
int nmpc_mpc_qp(int nb, int Nc, int nx, const double* list_A, const double* list_B, const double* list_C,
                const double* x0, const double* y_meas, const double* Xs, const double* Us, const double* f0,
                double h0, const double* Q, double R, double rho, double u_min, double u_max, double y_min,
                double y_max, double* Ge, double* Fe, double* Le, double* hv, double* u, double* eps,
                double* xpred, int* status, void* stream);

/* ---- closed loop (reference: NMPC_test.py:22-320 output_NMPC_linear; plant Systems.py:79-97) -----
 * Horizons: Nc <= 15 one wavefront per controller with everything in registers, 16..23 one wavefront with the
 * tables in LDS, from 24 on one workgroup per controller; the longest horizon whose tables fit a CU's LDS is 60 at
 * nx = 2, 55 at nx = 3, 52 at nx = 4, 44..39 at nx = 5..8 (output feedback), 42 with state feedback; longer ones are
 * refused by nmpc_ctrl_create / nmpc_mpc_qp.
 * `batch` independent controllers of the same model advance together; per step and controller:
 * up to max_iterations passes of {LPV embedding at the predicted trajectory, condensing, QP,
 * prediction update} until ||u - u_old||_2 < tol, then plant RK4 step, measurement, observer,
 * logging and warm-start shift.  Controllers that have converged drop out of the later passes. */
typedef struct nmpc_ctrl_config {
    int struct_size;                     /* sizeof(nmpc_ctrl_config) of the caller's header; checked by nmpc_ctrl_create */
    int batch, Nc, max_iterations, n_sim;
    double tol;                          /* NMPC_test.py:257 */
    double rho;                          /* slack weight, NMPC_test.py:225 */
    double Q[64];                        /* nx*nx row-major, nx <= 8 */
    double R;
    double u_min, u_max, y_min, y_max;   /* NORMALISED bounds, NMPC_test.py:70-73 */
    double u0, ustd, y0, ystd;           /* encoder.norm */
    double dt;
    double plant[6];                     /* unbalanced disc g, J, Km, I, M, tau (Systems.py:82-87) */
    int n_nodes;
    float lam[NMPC_MAX_NODES], w[NMPC_MAX_NODES];
    /* mode 0: output feedback with observer (output_NMPC_linear, NMPC_test.py:22-320).
     * mode 1: state feedback (par_NMPC_linear / NMPC_linear, NMPC.py:374-650): the measured plant state
     * is x0, box constraints x_min..x_max (normalised) act on measured and predicted states through
     * getDEMc (mpcUtil.py:112-134), y_min/y_max are unused, the model offset f(0,0) is dropped;
     * hard = 1 removes the slack variable (NMPC_linear's quadprog QP).  x_norm0 / x_normstd: the state
     * normalisation vectors (norm.y0, norm.ystd of a full-state model).
     * mode 2: the model is its own plant - x+ = f(x, u), y = h(x) + noise, the state is known to the controller
     * (no observer); otherwise the QP of mode 0.  For models that have no physical plant here (nx up to 8, hidden
     * width up to 256: the reference's larger encoders, LTI.ipynb cell 5, sinCos.ipynb cell 6, and the synthetic wide
     * configuration); x_log holds the model state, y_log / y_est_log the noisy / noise-free output. */
    int mode, hard;
    double x_min[8], x_max[8];
    double x_norm0[8], x_normstd[8];
    /* mode 1 only.  0 (default): input targets follow the step, Us = us[k .. k+Nc), and the linear cost term is
     * 2 (Gamma' Omega (Phi x0 - Xs) - Psi Us), what output_NMPC_linear does for the same step (NMPC_test.py:152,213).
     * 1: par_NMPC_linear as written - Us = us[0 .. Nc) at every step (NMPC.py:580) and + Psi' Us (NMPC.py:591). */
    int as_written;
    /* observer histories (NMPC_test.py:115-118: uhist has encoder.na entries, yhist encoder.nb + 1; the encoder
     * net reads [uhist, yhist]).  Their sum must equal the model's n_enc_in; 0, 0 = (n_enc_in - 1) / 2 inputs
     * and the rest outputs. */
    int n_uhist, n_yhist;
} nmpc_ctrl_config;

int nmpc_ctrl_create(nmpc_ctrl** out, const nmpc_model* m, const nmpc_ctrl_config* cfg);
void nmpc_ctrl_destroy(nmpc_ctrl* c);
/* Initial condition of every controller (NMPC_test.py:100-124): x0n[nx] (host) normalised initial
 * state, u0n normalised initial input, plant_x0[2] (host, may be NULL = 0). */
int nmpc_ctrl_reset(nmpc_ctrl* c, const double* x0n, double u0n, const double* plant_x0, void* stream);
/* Steps k0 .. k0+nsteps-1.  Xs[batch,n_sim+Nc,nx], Us[batch,n_sim+Nc] steady-state targets;
 * noise[n_sim,batch] measurement noise (mode 1: [n_sim,batch,nx]; may be NULL).  Logs (device fp64 / int32, may be NULL):
 * x_log[batch,n_sim+1,nx] u_log[batch,n_sim] e_log[batch,n_sim] y_log[batch,n_sim+1]
 * y_est_log[batch,n_sim+1] lpv_counter[batch,n_sim] plan_log[batch,n_sim,Nc]; entries k+1 (x, y, y_est)
 * and k (others) are written for every step k run. */
int nmpc_ctrl_run(nmpc_ctrl* c, int k0, int nsteps, const double* Xs, const double* Us, const double* noise,
                  double* x_log, double* u_log, double* e_log, double* y_log, double* y_est_log,
                  int* lpv_counter, double* plan_log, void* stream);
/* Per-controller QP status (reference: qp.solve_qp returning None would raise at NMPC_test.py:233-235 and end the
 * run).  Here a controller whose QP solve fails (status as nmpc_qp_solve: 1 iteration limit, 2 infeasible, 3 Hessian
 * not positive definite) keeps the plan it came with - the previous step's plan, shifted - for that step, takes no
 * further LPV pass in it, and the worst status of the step's solves is written to qp_status_log[batch, n_sim]
 * (device, int8; 0 = all solves of the step succeeded) by later nmpc_ctrl_run calls; the other controllers of the
 * batch are not affected.  NULL (the default) switches the log off. */
int nmpc_ctrl_set_status_log(nmpc_ctrl* c, signed char* qp_status_log);
/* stats[4] (host): [0] (controller, pass) QP solves since reset that did not return status 0,
 * [1] QP solves, [2] active-set iterations summed over them, [3] (long horizons) solves that the first-stage
 * kernel handed to the full active-set kernel.  Synchronises the stream. */
int nmpc_ctrl_stats(nmpc_ctrl* c, long long* stats, void* stream);

/* Optional per-kernel timing of a controller's launches with HIP events recorded on the stream the
 * kernels are launched on.  nmpc_ctrl_profile(c, n) with n >= 1 makes later nmpc_ctrl_run calls
 * bracket every launch of the steps k with k % n == 0 with an event pair (n = 1: every step; an
 * event pair costs a few microseconds of stream time, so a sparse sample leaves the run's
 * throughput untouched), n = 0 stops it; nmpc_ctrl_profile_read synchronises the stream, returns per kind
 * (0 embedding, 1 condense+QP solve, 2 plant/observer) the number of launches and their summed
 * duration in ms (arrays of 3), and clears the record. */
int nmpc_ctrl_profile(nmpc_ctrl* c, int enable);
int nmpc_ctrl_profile_read(nmpc_ctrl* c, int* launches, double* total_ms, void* stream);
/* Same record, per step: ms[nsteps][3] = summed launch durations of steps k0 .. k0+nsteps-1 per kind
 * (the getAB / solve / sim buckets of the reference's components_time, NMPC_test.py:152-300). */
int nmpc_ctrl_profile_read_steps(nmpc_ctrl* c, int k0, int nsteps, double* ms, void* stream);
/* Diagnostics: the clock the chip holds under the embedding kernel.  With `stamps` set (device memory, 2 * slots
 * 64-bit words), every workgroup b of the first-pass embedding launches of later nmpc_ctrl_run calls stores the
 * shader-clock cycles and the 100 MHz real-time ticks it lived for to stamps[2 (b % slots)], [.. + 1] (64-wide
 * split-fp16 kernel only); cycles / ticks * 0.1 is the clock in GHz.  NULL (the default) switches it off; no stamp
 * instruction executes then. */
int nmpc_ctrl_set_embed_stamps(nmpc_ctrl* c, unsigned long long* stamps, int slots);
/* The LPV re-scheduling loop of a step (NMPC_test.py:154-260) runs its first passes as batch-wide launches and,
 * for the canonical shapes, the remaining ones in one per-controller launch: returns the first pass (0-based) that
 * runs there, 0 if none does.  Launches of that kernel are timed with kind 1. */
int nmpc_ctrl_tail_from(const nmpc_ctrl* c);

/* ---- steady-state target selector (reference: mpcUtil.py:241-320 getXsUs_Cs, called pre-loop at
 * NMPC_test.py:79-80).  For every reference sample j, warm-started from sample j-1, iterate LPV passes:
 * embed A,B,C at (xs,us); solve the reference's 4-variable QP
 *   (I-A) xs - B us = f0,  C xs = r_j - h0,  soft input/output box with slack e (weight lam_e).
 * nu = ny = 1: the nx+1 equalities determine (xs,us) and the QP reduces to e = max(0, worst box violation),
 * which is what is computed.
 * Pass schedule.  max_lpv >= 1: up to max_lpv passes per sample, stop when ||xs-xs_old|| <= tol and
 * ||us-us_old|| <= tol (the loop as mpcUtil.py:288-313 intends it).  max_lpv = 0: the schedule that code
 * EXECUTES - xs/us are views of the solution buffer (mpcUtil.py:306-307) which the next solve overwrites
 * (:302) before xold = np.copy(xs) is taken (:304), so from the second solve on the stop test compares the
 * solution with itself: sample 0 takes at most two passes, every later sample exactly one, embedded at the
 * previous sample's target (the same holds for getXsUs, mpcUtil.py:224-233).
 * y_ref_n[nb,nsamp] normalised reference (device fp64).  Outputs (device): Xs[nb,nsamp,nx] Us[nb,nsamp]
 * Es[nb,nsamp] (fp64), iters[nb,nsamp] (int32, may be NULL), status[nb] (int32: 0 ok, 1 singular equality
 * system met).
 * nmpc_target_select allocates and frees its own workspace and synchronises the stream.  nmpc_target_select_ws
 * does neither: the caller passes `workspace` (device memory, 16-byte aligned, at least
 * nmpc_target_workspace_bytes(nx, nb, max_lpv) bytes; contents need not be preserved between calls) and the
 * whole selection is issued asynchronously on `stream`. */
unsigned long long nmpc_target_workspace_bytes(int nx, int nb, int max_lpv);
int nmpc_target_select(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                       double u_max, double y_min, double y_max, const float* lam, const float* w,
                       int n_nodes, int max_lpv, double tol, double* Xs, double* Us, double* Es,
                       int* iters, int* status, void* stream);
int nmpc_target_select_ws(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                          double u_max, double y_min, double y_max, const float* lam, const float* w,
                          int n_nodes, int max_lpv, double tol, double* Xs, double* Us, double* Es,
                          int* iters, int* status, void* workspace, unsigned long long workspace_bytes,
                          void* stream);
/* No counterpart in the reference: nmpc_target_select for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same arguments and results as
 * nmpc_target_select; the workspace is a static device array of NMPC_TARGET_STATIC_BYTES inside the library
 * (one copy per GPU, held until the process ends), so the call allocates nothing, frees nothing and does not
 * synchronise: everything is issued asynchronously on `stream`.  Sizes whose workspace
 * (nmpc_target_workspace_bytes) exceeds NMPC_TARGET_STATIC_BYTES are rejected with an error - about 55 000
 * controllers at nx = 2, 9 600 at nx = 8.  All calls on one GPU share the array: calls that overlap in time on
 * the same GPU (different streams or threads) race; serialise them or use nmpc_target_select_ws. */
#define NMPC_TARGET_STATIC_BYTES (4u << 20)
int nmpc_target_select_static(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                              double u_max, double y_min, double y_max, const float* lam, const float* w,
                              int n_nodes, int max_lpv, double tol, double* Xs, double* Us, double* Es,
                              int* iters, int* status, void* stream);
/* State-constrained selector (reference: mpcUtil.py:167-239 getXsUs, called at NMPC.py:541-542): the
 * same iteration with A, B only, a fixed output row C[nx] (host), equality offsets f0[nx] (host; NULL =
 * the model's f(0,0)) and h0, and the soft box -u, u, -xs, xs (x_min[nx], x_max[nx], host, normalised).
 * _ws: caller's workspace, asynchronous, as above. */
int nmpc_target_select_state(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                             double u_max, const double* x_min, const double* x_max, const double* C,
                             const double* f0, double h0, const float* lam, const float* w, int n_nodes,
                             int max_lpv, double tol, double* Xs, double* Us, double* Es, int* iters,
                             int* status, void* stream);
int nmpc_target_select_state_ws(const nmpc_model* m, int nb, int nsamp, const double* y_ref_n, double u_min,
                                double u_max, const double* x_min, const double* x_max, const double* C,
                                const double* f0, double h0, const float* lam, const float* w, int n_nodes,
                                int max_lpv, double tol, double* Xs, double* Us, double* Es, int* iters,
                                int* status, void* workspace, unsigned long long workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
