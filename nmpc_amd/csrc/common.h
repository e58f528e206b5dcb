// Shared host/device helpers of the nmpc_amd HIP library.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdarg>
#include <cstdio>
#include <string>

namespace nmpc {

void set_error(const char* fmt, ...);

// hipFuncSetAttribute acts on the current device only: `mask` (one per kernel) remembers the devices
// on which the attribute has been set.  first_use_on_device is true until mark_used_on_device has been
// called for the current device - callers set the attribute, then mark; two threads racing here both set
// the (idempotent) attribute, and neither launches before its own call has returned.
inline bool first_use_on_device(const std::atomic<unsigned long long>& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (mask.load(std::memory_order_acquire) & (1ull << (dev & 63))) == 0;
}
inline void mark_used_on_device(std::atomic<unsigned long long>& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
}

// a handle's device memory belongs to the device that was current when it was created
inline bool on_device(int device) {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess && dev == device;
}

#define NMPC_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            nmpc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                            __LINE__);                                                        \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define NMPC_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            nmpc::set_error(__VA_ARGS__);        \
            return 1;                            \
        }                                        \
    } while (0)

constexpr int HID = 64;             // hidden width handled by the MFMA embedding kernel
// Packed-net layout (float offsets), see pack_net in model.cpp.
constexpr int PK_WF = 0;          // [4][4][64][4]    forward A-fragments of c2*W2
constexpr int PK_WR = 4096;       // [4][4][4][64][4] reverse A-fragments of (4 W4[c]) .* W2^T, c = 0..3
constexpr int PK_REC1 = 20480;    // [64][4] {c2*b0[k], c2*b2[k], split-fp16 pack: inverse scale of reverse matrix k (k < 8), 0}
constexpr int PK_REC2 = 20736;    // [64][4] W0[k][0..3] (zero padded)
constexpr int PK_LIN = 20992;     // [4][4]  W_lin[c][j] (zero padded)
constexpr int PK_J0 = 21008;      // [4][4]  Jacobian of the nonlinear branch at the origin, W4 D2 W2 D1 W0 (fp64-evaluated)
constexpr int PK_REC3 = 21024;    // [64][4] W0[k][4..7] (zero padded): nets with more than four inputs
constexpr int PK_LIN2 = 21280;    // [4][4]  W_lin[c][4..7]
constexpr int PK_J02 = 21296;     // [4][4]  origin Jacobian of the nonlinear branch, inputs 4..7
constexpr int PK_REC4 = 21312;    // [64][4] W0[k][8..11] (zero padded): nets with more than eight inputs
constexpr int PK_LINX = 21568;    // [8][12] W_lin[c][j]: nets with more than four outputs or eight inputs read these ...
constexpr int PK_J0X = 21664;     // [8][12] ... and this origin Jacobian (from global memory, once per tile)
constexpr int PK_WR2 = 21760;     // [4][4][4][64][4] reverse A-fragments of output rows c = 4..7 (split-fp16 pack only)
constexpr int PACK_FLOATS = PK_WR2 + 4 * 4096;
constexpr int PK_TAIL = PK_REC3 - PK_REC1;       // REC1 + REC2 + LIN + J0, contiguous: what nets with <= 4 inputs stage
constexpr int PK_TAIL_EXT = PK_REC4 - PK_REC1;   // the same with the tables of inputs 4..7 behind them
constexpr int PK_TAIL_EXT2 = PK_LINX - PK_REC1;  // and with W0's columns 8..11 behind those

// One net in the layout of the general-width embedding kernel (embed_wide.hip); hidden width padded
// to 16*ht (ht = 4, 8, 16), outputs padded to nout_pad (1, 4, 8), inputs to 16; all fp32, device.
struct WideNet {
    const float* fwd;   // [layers-1][ht][ht][64][4]  A-fragments of c2*W_l, l = 2..layers:  W[16T + (l&15)][16S + 4(l>>4) + j]
    const float* rev;   // [layers-1][ht][ht][64][4]  A-fragments of W_l^T (times 4 for l > 2)
    const float* w0t;   // [ht][64][4]      A-fragments of W0^T:   W0[16T + 4(l>>4) + j][l&15]
    const float* w0;    // [H][16]          W0 rows
    const float* cb0;   // [H]  c2*b0
    const float* cb2;   // [layers-1][H]  c2*b_l of hidden layers l = 2..layers
    const float* w4s;   // [nout_pad][H]    output matrix, times 4 when layers >= 2
    const float* lin;   // [nout_pad][16]   W_lin
    const float* j0;    // [nout_pad][16]   nonlinear-branch Jacobian at the origin (fp64-evaluated)
    int ht, nout_pad;
    int layers;         // hidden layers (mpcUtil.py:7,18-25: CasADiExp loops over f_n_hidden_layers); 2 = embed_wide.hip, else embed_deep.hip
};

// One net in the layout of embed_panel.hip.  Units are numbered by MFMA position: element j (0..7) that lane
// group g = lane >> 4 supplies to k-step ks (32 units) is unit 16 (2 ks + (j >> 2)) + 4 g + (j & 3), which is
// also row 4 g + (j & 3) of the 16-row output tile 2 ks + (j >> 2): an accumulator pair is the next B operand.
struct PanelNet {
    const _Float16* fwd;   // [HT][KST][2 planes][64][8]  s_f 2 c2 W2   rows = units of tile T, K = units of k-step
    const _Float16* rev;   // [HT][KST][2][64][8]          s_r W2^T
    const _Float16* w0t;   // [KST][2][64][8]              s_0 W0^T      rows = inputs (16, zero padded)
    const float* w0g;      // [H][16]   c2 W0 rows (slopes of the layer-1 exponents)
    const float* cb0;      // [H]       c2 b0
    const float* cb2s;     // [H]       256 s_f c2 (b2 - W2 1)   (the forward operand is 256 r1)
    const float* w4s;      // [nout_pad][H]  s_c W4[c]
    const float* oscale;   // [nout_pad]     16 / (s_0 s_v s_r s_c)
    const float* lin;      // [nout_pad][16] W_lin
    const float* j0;       // [nout_pad][16] nonlinear-branch Jacobian at the origin
    float inv_sf, s_v;     // 1 / (256 s_f); scale of the contraction operand
    int ht, nout_pad;
    int ok;                // 0: this net does not suit the format (the fp32 kernel serves it)
};

// Device-resident packed networks.
struct ModelDev {
    int nx, nu, ny, hidden, n_enc_in;
    int f_layers, h_layers;   // hidden layers of the f and h nets
    float* fpack;   // f-net, fp32 MFMA layout
    float* hpack;   // h-net, fp32 MFMA layout
    float* fpack_h2;   // f-net, split-fp16 MFMA layout (same offsets, fragment regions hold halves)
    float* hpack_h2;   // h-net, split-fp16 MFMA layout
    WideNet wide_f, wide_h;   // both nets for the general-width kernel
    float* wide_slab;         // their storage
    PanelNet panel_f, panel_h;   // split-fp16 layout of the wide kernel (two hidden layers, width 65..256); .ok = 0 if absent
    float* panel_slab;
    int narrow_ok;            // the 64-wide kernels cover this model's sizes
    int step_ok;              // layer-1 biases small enough for node-to-node stepping of 2^-a (embed.hip)
    float* enc;     // observer net, plain row-major params concatenated (may be null):
                    // lin_w[out,in] lin_b w0[H,in] b0 w2[H,H] b2 w4[out,H] b4 w2^T[H,H]
    float* hplain;  // h-net in the same plain layout (for the output estimate)
    float* fplain;  // f-net in the same plain layout (the model as its own plant)
    double f0[16];  // f(0,0)  ("correction",   NMPC_test.py:38)
    double h0[16];  // h(0)    ("correction_h", NMPC_test.py:41)
};

}  // namespace nmpc

struct nmpc_model {
    nmpc::ModelDev d;
    int device;
};
