// Model handle: repacks the reference-ordered network parameters into the layouts the kernels
// read, and evaluates the constant offsets f(0,0), h(0) (NMPC_test.py:38-43) in fp64.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/nmpc_amd.h"
#include "common.h"

namespace nmpc {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// Hidden-unit index supplied by lane-quarter hq at MFMA k-step s (v_mfma_f32_16x16x4_f32).  It
// equals the row that register (s & 3) of output tile (s >> 2) of a 16x16 f32 accumulator holds in
// lane-quarter hq, so an accumulator can be fed straight back as the B operand of the next product.
static inline int kappa(int s, int hq) { return 16 * (s >> 2) + 4 * hq + (s & 3); }

static void pack_net(const float* const* p, int n_in, int n_out, std::vector<float>& out) {
    const float *lin_w = p[0], *w0 = p[2], *b0 = p[3], *w2 = p[4], *b2 = p[5], *w4 = p[6];
    const double c2 = 2.0 * 1.4426950408889634074;  // tanh(a) = 1 - 2/(2^(c2 a) + 1)
    out.assign(PACK_FLOATS, 0.0f);
    for (int T = 0; T < 4; ++T)
        for (int s = 0; s < 16; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int hq = lane >> 4, r = lane & 15, k = kappa(s, hq);
                const size_t o = ((size_t)(T * 4 + (s >> 2)) * 64 + lane) * 4 + (s & 3);
                out[PK_WF + o] = (float)(c2 * (double)w2[(16 * T + r) * HID + k]);
                // reverse product r1[kk] = sum_i W2[i][kk] * (4 W4[c][i]) * (d2[i]/4); reduction index i = k
                for (int c = 0; c < n_out && c < 4; ++c)       // (the fp32-MFMA kernel serves nets with up to three outputs)
                    out[PK_WR + 4096 * c + o] = (float)((double)w2[k * HID + 16 * T + r] * 4.0 * (double)w4[c * HID + k]);
            }
    for (int k = 0; k < HID; ++k) {
        out[PK_REC1 + 4 * k + 0] = (float)(c2 * (double)b0[k]);
        out[PK_REC1 + 4 * k + 1] = (float)(c2 * (double)b2[k]);
        for (int j = 0; j < n_in; ++j) out[(j < 4 ? PK_REC2 : (j < 8 ? PK_REC3 : PK_REC4)) + 4 * k + (j & 3)] = w0[k * n_in + j];
    }
    for (int c = 0; c < n_out; ++c)
        for (int j = 0; j < n_in; ++j) {
            if (c < 4 && j < 8) out[(j < 4 ? PK_LIN : PK_LIN2) + 4 * c + (j & 3)] = lin_w[c * n_in + j];
            out[PK_LINX + 12 * c + j] = lin_w[c * n_in + j];
        }
    // nonlinear-branch Jacobian at the origin: W4 diag(1 - t2^2) W2 diag(1 - t1^2) W0
    std::vector<double> t1(HID), d2(HID), v(HID);
    for (int k = 0; k < HID; ++k) t1[k] = std::tanh((double)b0[k]);
    for (int i = 0; i < HID; ++i) {
        double a = b2[i];
        for (int k = 0; k < HID; ++k) a += (double)w2[i * HID + k] * t1[k];
        const double t = std::tanh(a);
        d2[i] = 1.0 - t * t;
    }
    for (int c = 0; c < n_out; ++c) {
        for (int k = 0; k < HID; ++k) {
            double a = 0.0;
            for (int i = 0; i < HID; ++i) a += (double)w4[c * HID + i] * d2[i] * (double)w2[i * HID + k];
            v[k] = a * (1.0 - t1[k] * t1[k]);
        }
        for (int j = 0; j < n_in; ++j) {
            double a = 0.0;
            for (int k = 0; k < HID; ++k) a += v[k] * (double)w0[k * n_in + j];
            if (c < 4 && j < 8) out[(j < 4 ? PK_J0 : PK_J02) + 4 * c + (j & 3)] = (float)a;
            out[PK_J0X + 12 * c + j] = (float)a;
        }
    }
}

// Split-fp16 packing: every fragment element W is stored as two halves, Wh = fp16(W) and
// Wl = fp16(W - Wh) (unscaled; subnormal for |W| < 1/8, absolute resolution 2^-24).  Returns false
// when the fragments do not suit that format: an entry beyond the fp16 range, or a forward matrix whose
// largest entry is below 2^-6 (the absolute resolution would cost relative accuracy).  Fragment of tile T, k-step ks: lane l (row 16T + (l&15), quarter
// hq = l>>4) holds 8 halves j = 0..7 for the hidden unit 16 (2 ks + (j>>2)) + 4 hq + (j&3), i.e.
// unit s = 8 ks + j in the kernel's numbering.  Region layout: [plane][T][ks][lane][8 halves].
static bool pack_net_h2(const float* const* p, int n_in, int n_out, std::vector<float>& out) {
    std::vector<float> f32;
    pack_net(p, n_in, n_out, f32);                 // small tables (REC1, REC2, LIN) are shared
    out = f32;
    const float *w2 = p[4], *w4 = p[6];
    const double c2 = 2.0 * 1.4426950408889634074;
    _Float16* base = reinterpret_cast<_Float16*>(out.data());
    bool ok = true;
    double wmax[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // forward matrix, reverse matrices c = 0..7
    auto rev_region = [](int c) { return (size_t)(c < 4 ? PK_WR + 4096 * c : PK_WR2 + 4096 * (c - 4)); };
    auto put = [&](size_t region_floats, int T, int ks, int lane, int j, double w) {
        if (!(std::fabs(w) < 6.0e4)) ok = false;
        double& mx = wmax[region_floats == (size_t)PK_WF ? 0
                          : (region_floats >= (size_t)PK_WR2 ? 5 + (region_floats - PK_WR2) / 4096 : 1 + (region_floats - PK_WR) / 4096)];
        mx = std::fabs(w) > mx ? std::fabs(w) : mx;
        const _Float16 h = (_Float16)(float)w;
        const _Float16 l = (_Float16)(float)(w - (double)(float)h);   // unscaled remainder (subnormals kept)
        _Float16* reg = base + 2 * region_floats;
        reg[((((size_t)0 * 4 + T) * 2 + ks) * 64 + lane) * 8 + j] = h;
        reg[((((size_t)1 * 4 + T) * 2 + ks) * 64 + lane) * 8 + j] = l;
    };
    // Each reverse matrix is stored times a power of two that brings its largest entry into [0.5, 1), so
    // that the unscaled fp16 remainders resolve it to 2^-24 of that entry whatever the size of W4's
    // row; the kernel multiplies the accumulated row by the inverse (REC1[4 c + 2]) once at the end.
    double rscale[8] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    for (int c = 0; c < n_out; ++c) {
        double mx = 0.0;
        for (int k = 0; k < HID; ++k)
            for (int i = 0; i < HID; ++i) mx = std::fmax(mx, std::fabs((double)w2[k * HID + i] * 4.0 * (double)w4[c * HID + k]));
        if (mx > 0.0 && std::isfinite(mx)) {
            int e = 0;
            std::frexp(mx, &e);                       // mx = f 2^e, f in [0.5, 1)
            rscale[c] = std::ldexp(1.0, -e);
        }
        out[PK_REC1 + 4 * c + 2] = (float)(1.0 / rscale[c]);
    }
    for (int T = 0; T < 4; ++T)
        for (int ks = 0; ks < 2; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int hq = lane >> 4, r = lane & 15;
                    const int k = 16 * (2 * ks + (j >> 2)) + 4 * hq + (j & 3);
                    put(PK_WF, T, ks, lane, j, 2.0 * c2 * (double)w2[(16 * T + r) * HID + k]);
                    for (int c = 0; c < n_out; ++c)
                        put(rev_region(c), T, ks, lane, j,
                            rscale[c] * (double)w2[k * HID + 16 * T + r] * 4.0 * (double)w4[c * HID + k]);
                }
    // The forward product takes r = (1 + t1) / 2 = 1 / (1 + 2^-a) as its operand instead of t1 = tanh:
    // W2 t1 + b2 = (2 W2) r + (b2 - W2 1); the fragments above carry the 2, the bias the constant.
    const float* b2 = p[5];
    for (int k = 0; k < HID; ++k) {
        double a = b2[k];
        for (int i = 0; i < HID; ++i) a -= (double)w2[k * HID + i];
        out[PK_REC1 + 4 * k + 1] = (float)(c2 * a);
    }
    if (wmax[0] < 1.0 / 64.0) ok = false;          // forward matrix (not rescaled: its product feeds exp2 directly)
    for (int c = 0; c < n_out; ++c)
        if (!(wmax[1 + c] > 0.0)) ok = false;      // an all-zero output row: leave it to the fp32 kernel
    return ok;
}

// Layout of the general-width kernels (embed_wide.hip, embed_deep.hip): see WideNet in common.h.  Appends the
// net's arrays to `slab` (floats) and records the offsets in `off` (fwd, rev, w0t, w0, cb0, cb2, w4s,
// lin, j0).  hidden is padded with zero units to 16*ht, outputs with zero rows to nout_pad.
// p = 4 + 2 L arrays in the reference's order (mpcUtil.py:18-29): linear skip, L hidden layers, output layer.
static void pack_wide(const float* const* p, int L, int n_in, int n_out, int hidden, int ht, int nout_pad,
                      std::vector<float>& slab, size_t off[9]) {
    const float *lin_w = p[0], *w0 = p[2], *b0 = p[3], *wout = p[2 + 2 * L];
    const double c2 = 2.0 * 1.4426950408889634074;
    const int H = 16 * ht, nm = L - 1;                 // nm square layers l = 2..L
    auto grab = [&](size_t nfl) { const size_t o = slab.size(); slab.resize(o + ((nfl + 3) & ~(size_t)3), 0.0f); return o; };
    off[0] = grab((size_t)nm * ht * ht * 256); off[1] = grab((size_t)nm * ht * ht * 256); off[2] = grab((size_t)ht * 256);
    off[3] = grab((size_t)H * 16); off[4] = grab(H); off[5] = grab((size_t)(nm > 0 ? nm : 1) * H); off[6] = grab((size_t)nout_pad * H);
    off[7] = grab((size_t)nout_pad * 16); off[8] = grab((size_t)nout_pad * 16);
    for (int li = 0; li < nm; ++li) {
        const float* w = p[4 + 2 * li];
        const float* b = p[5 + 2 * li];
        auto W = [&](int i, int k) { return (i < hidden && k < hidden) ? (double)w[(size_t)i * hidden + k] : 0.0; };
        // the reverse sweep multiplies the product of layer l > 2 by (1 - t^2)/4 of the layer below
        const double rs = li >= 1 ? 4.0 : 1.0;
        for (int T = 0; T < ht; ++T)
            for (int S = 0; S < ht; ++S)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const size_t o = ((((size_t)li * ht + T) * ht + S) * 64 + l) * 4 + j;
                        const int r = 16 * T + (l & 15), c = 16 * S + 4 * (l >> 4) + j;
                        slab[off[0] + o] = (float)(c2 * W(r, c));
                        slab[off[1] + o] = (float)(rs * W(c, r));            // W^T[r][c]
                    }
        for (int k = 0; k < hidden; ++k) slab[off[5] + (size_t)li * H + k] = (float)(c2 * (double)b[k]);
    }
    for (int T = 0; T < ht; ++T)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * T + 4 * (l >> 4) + j, jn = l & 15;
                slab[off[2] + ((size_t)T * 64 + l) * 4 + j] = (k < hidden && jn < n_in) ? w0[(size_t)k * n_in + jn] : 0.0f;
            }
    const float osc = L >= 2 ? 4.0f : 1.0f;
    for (int k = 0; k < hidden; ++k) {
        for (int j = 0; j < n_in; ++j) slab[off[3] + (size_t)k * 16 + j] = w0[(size_t)k * n_in + j];
        slab[off[4] + k] = (float)(c2 * (double)b0[k]);
        for (int c = 0; c < n_out; ++c) slab[off[6] + (size_t)c * H + k] = osc * wout[(size_t)c * hidden + k];
    }
    for (int c = 0; c < n_out; ++c)
        for (int j = 0; j < n_in; ++j) slab[off[7] + (size_t)c * 16 + j] = lin_w[(size_t)c * n_in + j];
    // nonlinear-branch Jacobian at the origin (fp64): W_out D_L W_L ... D_1 W_0
    std::vector<double> t(hidden), tn(hidden), M((size_t)hidden * n_in), Mn((size_t)hidden * n_in);
    for (int k = 0; k < hidden; ++k) {
        t[k] = std::tanh((double)b0[k]);
        for (int j = 0; j < n_in; ++j) M[(size_t)k * n_in + j] = (1.0 - t[k] * t[k]) * (double)w0[(size_t)k * n_in + j];
    }
    for (int li = 0; li < nm; ++li) {
        const float* w = p[4 + 2 * li];
        const float* b = p[5 + 2 * li];
        for (int i = 0; i < hidden; ++i) {
            double a = b[i];
            for (int k = 0; k < hidden; ++k) a += (double)w[(size_t)i * hidden + k] * t[k];
            tn[i] = std::tanh(a);
            for (int j = 0; j < n_in; ++j) {
                double m = 0.0;
                for (int k = 0; k < hidden; ++k) m += (double)w[(size_t)i * hidden + k] * M[(size_t)k * n_in + j];
                Mn[(size_t)i * n_in + j] = (1.0 - tn[i] * tn[i]) * m;
            }
        }
        t = tn; M = Mn;
    }
    for (int c = 0; c < n_out; ++c)
        for (int j = 0; j < n_in; ++j) {
            double a = 0.0;
            for (int k = 0; k < hidden; ++k) a += (double)wout[(size_t)c * hidden + k] * M[(size_t)k * n_in + j];
            slab[off[8] + (size_t)c * 16 + j] = (float)a;
        }
}

// Layout of the split-fp16 wide kernel (embed_panel.hip; PanelNet in embed_panel.h).  Appends to `slab` (floats;
// fragment regions hold two halves per float) and records offsets in `off` (fwd, rev, w0t, w0g, cb0, cb2s, w4s,
// oscale).  Returns false when the net does not suit the format (non-finite or all-zero matrices).
static bool pack_panel(const float* const* p, int n_in, int n_out, int hidden, int ht, int nout_pad,
                       std::vector<float>& slab, size_t off[8], float* inv_sf, float* s_v_out) {
    const float *w0 = p[2], *b0 = p[3], *w2 = p[4], *b2 = p[5], *w4 = p[6];
    const double c2 = 2.0 * 1.4426950408889634074;
    const int H = 16 * ht, kst = ht / 2;
    auto grab = [&](size_t nfl) { const size_t o = slab.size(); slab.resize(o + ((nfl + 3) & ~(size_t)3), 0.0f); return o; };
    off[0] = grab((size_t)ht * kst * 512); off[1] = grab((size_t)ht * kst * 512); off[2] = grab((size_t)kst * 512);
    off[3] = grab((size_t)H * 16); off[4] = grab(H); off[5] = grab(H); off[6] = grab((size_t)nout_pad * H); off[7] = grab(nout_pad);
    auto W2 = [&](int i, int k) { return (i < hidden && k < hidden) ? (double)w2[(size_t)i * hidden + k] : 0.0; };
    auto W0 = [&](int k, int j) { return (k < hidden && j < n_in) ? (double)w0[(size_t)k * n_in + j] : 0.0; };
    auto W4 = [&](int c, int k) { return (c < n_out && k < hidden) ? (double)w4[(size_t)c * hidden + k] : 0.0; };
    // power of two that brings mx into [lo, 2 lo)
    auto pow2_to = [](double mx, double lo) { int e = 0; std::frexp(mx / lo, &e); return std::ldexp(1.0, 1 - e); };
    double m2 = 0.0, m0 = 0.0;
    bool ok = true;
    for (int i = 0; i < hidden; ++i)
        for (int k = 0; k < hidden; ++k) { const double v = std::fabs(W2(i, k)); if (!std::isfinite(v)) ok = false; m2 = std::fmax(m2, v); }
    for (int k = 0; k < hidden; ++k)
        for (int j = 0; j < n_in; ++j) { const double v = std::fabs(W0(k, j)); if (!std::isfinite(v)) ok = false; m0 = std::fmax(m0, v); }
    for (int k = 0; k < hidden; ++k) if (!std::isfinite((double)b0[k]) || !std::isfinite((double)b2[k])) ok = false;
    if (!ok || !(m2 > 0.0) || !(m0 > 0.0)) return false;
    const double s_f = pow2_to(2.0 * c2 * m2, 512.0), s_r = pow2_to(m2, 512.0), s_0 = pow2_to(m0, 512.0);
    std::vector<double> s_c(nout_pad, 1.0);
    for (int c = 0; c < n_out; ++c) {
        double mx = 0.0;
        for (int k = 0; k < hidden; ++k) { const double v = std::fabs(W4(c, k)); if (!std::isfinite(v)) return false; mx = std::fmax(mx, v); }
        if (mx > 0.0) s_c[c] = pow2_to(mx, 128.0);
    }
    // |s_r W2^T (d2 o s_c W4[c])| d1 <= bound with d1, d2 <= 1/4: s_v brings the bound to 2^10 at most
    double bound = 0.0;
    for (int c = 0; c < n_out; ++c)
        for (int k = 0; k < hidden; ++k) {
            double a = 0.0;
            for (int j = 0; j < hidden; ++j) a += std::fabs(W2(j, k)) * 0.25 * s_c[c] * std::fabs(W4(c, j));
            bound = std::fmax(bound, 0.25 * s_r * a);
        }
    double s_v = 1.0;
    if (bound > 0.0) { int e = 0; std::frexp(bound, &e); s_v = std::ldexp(1.0, 10 - e); }
    const double s_b = 256.0;                          // the kernel forms 256 r1 as the forward operand
    *inv_sf = (float)(1.0 / (s_f * s_b));
    *s_v_out = (float)s_v;
    _Float16* fwd = reinterpret_cast<_Float16*>(slab.data() + off[0]);
    _Float16* rev = reinterpret_cast<_Float16*>(slab.data() + off[1]);
    _Float16* w0t = reinterpret_cast<_Float16*>(slab.data() + off[2]);
    auto put = [&](_Float16* reg, size_t frag, int lane, int j, double w) {
        if (!(std::fabs(w) < 6.0e4)) ok = false;
        const _Float16 h = (_Float16)(float)w;
        const _Float16 l = (_Float16)(float)(w - (double)(float)h);
        reg[((frag * 2 + 0) * 64 + lane) * 8 + j] = h;
        reg[((frag * 2 + 1) * 64 + lane) * 8 + j] = l;
    };
    for (int ks = 0; ks < kst; ++ks)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int g = lane >> 4, r = lane & 15;
                const int k = 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3);       // unit of K element j
                for (int T = 0; T < ht; ++T) {
                    put(fwd, (size_t)T * kst + ks, lane, j, s_f * 2.0 * c2 * W2(16 * T + r, k));
                    put(rev, (size_t)T * kst + ks, lane, j, s_r * W2(k, 16 * T + r));
                }
                put(w0t, ks, lane, j, s_0 * W0(k, r));
            }
    for (int k = 0; k < hidden; ++k) {
        for (int j = 0; j < n_in; ++j) slab[off[3] + (size_t)k * 16 + j] = (float)(c2 * W0(k, j));
        slab[off[4] + k] = (float)(c2 * (double)b0[k]);
        double a = b2[k];
        for (int i = 0; i < hidden; ++i) a -= W2(k, i);
        slab[off[5] + k] = (float)(s_f * s_b * c2 * a);
        for (int c = 0; c < n_out; ++c) slab[off[6] + (size_t)c * H + k] = (float)(s_c[c] * W4(c, k));
    }
    for (int c = 0; c < nout_pad; ++c) slab[off[7] + c] = (float)(16.0 / (s_0 * s_v * s_r * s_c[c]));
    return ok;
}

static void eval_net_at_zero(const float* const* p, int L, int n_out, int hidden, double* y) {
    std::vector<double> t(hidden), tn(hidden);
    for (int k = 0; k < hidden; ++k) t[k] = std::tanh((double)p[3][k]);
    for (int li = 0; li + 1 < L; ++li) {
        for (int i = 0; i < hidden; ++i) {
            double a = p[5 + 2 * li][i];
            for (int k = 0; k < hidden; ++k) a += (double)p[4 + 2 * li][(size_t)i * hidden + k] * t[k];
            tn[i] = std::tanh(a);
        }
        t = tn;
    }
    for (int c = 0; c < n_out; ++c) {
        double a = (double)p[1][c] + (double)p[3 + 2 * L][c];
        for (int k = 0; k < hidden; ++k) a += (double)p[2 + 2 * L][(size_t)c * hidden + k] * t[k];
        y[c] = a;
    }
}

}  // namespace nmpc

using namespace nmpc;

extern "C" const char* nmpc_last_error(void) { return g_err.c_str(); }
extern "C" int nmpc_version(void) { return NMPC_AMD_VERSION; }

// Packs and uploads the nets into `m` (zero-initialised); on failure the caller destroys `m`.
static int model_fill(nmpc_model* m, int nx, int nu, int ny, int hidden, int Lf, int Lh, int n_enc_in,
                      const float* const* fn, const float* const* hn, const float* const* enc) {
    // Two-hidden-layer nets narrower than 64 units (the reference trains 32-wide encoders too: UnbalancedDisc.ipynb,
    // n_nodes_per_layer = 32; CasADiExp takes any width, mpcUtil.py:6-31) are padded to 64 with units whose weights
    // and biases are zero: tanh(0) = 0 feeds nothing forward and the zero rows / columns of W2 and W4 pass nothing
    // back, so values and Jacobians are those of the net as given, and the 64-wide kernels and the observer loop
    // serve it.
    if (hidden < HID && Lf == 2 && Lh == 2) {
        std::vector<std::vector<float>> store;
        store.reserve(24);
        auto pad = [&](const float* const* p, int n_in, int n_out, const float* q[8]) {
            auto keep = [&](std::vector<float> v) { store.push_back(std::move(v)); return store.back().data(); };
            std::vector<float> w0((size_t)HID * n_in, 0.f), b0(HID, 0.f), w2((size_t)HID * HID, 0.f), b2(HID, 0.f), w4((size_t)n_out * HID, 0.f);
            for (int k = 0; k < hidden; ++k) {
                for (int j = 0; j < n_in; ++j) w0[(size_t)k * n_in + j] = p[2][(size_t)k * n_in + j];
                b0[k] = p[3][k]; b2[k] = p[5][k];
                for (int i = 0; i < hidden; ++i) w2[(size_t)k * HID + i] = p[4][(size_t)k * hidden + i];
                for (int c = 0; c < n_out; ++c) w4[(size_t)c * HID + k] = p[6][(size_t)c * hidden + k];
            }
            q[0] = p[0]; q[1] = p[1]; q[2] = keep(std::move(w0)); q[3] = keep(std::move(b0)); q[4] = keep(std::move(w2));
            q[5] = keep(std::move(b2)); q[6] = keep(std::move(w4)); q[7] = p[7];
        };
        const float *fq[8], *hq[8], *eq[8];
        pad(fn, nx + nu, nx, fq); pad(hn, nx, ny, hq);
        if (enc) pad(enc, n_enc_in, nx, eq);
        return model_fill(m, nx, nu, ny, HID, Lf, Lh, n_enc_in, fq, hq, enc ? eq : nullptr);
    }
    // the 64-wide kernels (embed.hip) are instantiated for these sizes; everything else runs on the
    // general-width kernel (embed_wide.hip)
    // (nx = 4: split-fp16 kernel only, each f tile as two tiles of two output rows; nets that do not suit the
    // split-fp16 format stay on the general-width kernel)
    // (nx = 5..8 with one or two outputs, the shape of the reference's sin/cos encoder, sinCos.ipynb cell 6: split-fp16
    // kernel only, f tiles of two output rows each)
    const bool narrow = hidden == HID && nu == 1 && Lf == 2 && Lh == 2 &&
                        ((ny == 1 && nx >= 2 && nx <= 8) || (ny == 2 && nx >= 5 && nx <= 8));
    m->d.f_layers = Lf; m->d.h_layers = Lh;
    m->d.nx = nx; m->d.nu = nu; m->d.ny = ny; m->d.hidden = hidden; m->d.n_enc_in = enc ? n_enc_in : 0;
    NMPC_HIP_CHECK(hipGetDevice(&m->device));
    m->d.narrow_ok = narrow ? 1 : 0;
    {   // a = lam g + b0 keeps the sign of g once |lam g| exceeds |b0|, so a unit that has left the range of
        // fp32's exp2 can only come back if its bias is itself of that size: stepping is enabled for
        // |c2 b0| <= 32 (tanh arguments up to 11; larger biases pin a unit at +-1 for all sensible inputs)
        const double c2 = 2.0 * 1.4426950408889634074;
        double bmax = 0.0;
        for (int k = 0; k < hidden; ++k) bmax = std::fmax(bmax, std::fmax(std::fabs(c2 * fn[3][k]), std::fabs(c2 * hn[3][k])));
        m->d.step_ok = bmax <= 32.0 ? 1 : 0;
    }
    {
        const int ht = hidden <= 64 ? 4 : (hidden <= 128 ? 8 : 16);
        auto padout = [](int n) { return n <= 1 ? 1 : (n <= 4 ? 4 : 8); };
        std::vector<float> slab;
        size_t of[9], oh[9];
        pack_wide(fn, Lf, nx + nu, nx, hidden, ht, padout(nx), slab, of);
        pack_wide(hn, Lh, nx, ny, hidden, ht, padout(ny), slab, oh);
        NMPC_HIP_CHECK(hipMalloc(&m->d.wide_slab, slab.size() * sizeof(float)));
        NMPC_HIP_CHECK(hipMemcpy(m->d.wide_slab, slab.data(), slab.size() * sizeof(float), hipMemcpyHostToDevice));
        auto bind = [&](WideNet& w, const size_t* o, int nout_pad, int layers) {
            const float* b = m->d.wide_slab;
            w.fwd = b + o[0]; w.rev = b + o[1]; w.w0t = b + o[2]; w.w0 = b + o[3]; w.cb0 = b + o[4]; w.cb2 = b + o[5];
            w.w4s = b + o[6]; w.lin = b + o[7]; w.j0 = b + o[8]; w.ht = ht; w.nout_pad = nout_pad; w.layers = layers;
        };
        bind(m->d.wide_f, of, padout(nx), Lf); bind(m->d.wide_h, oh, padout(ny), Lh);
        // split-fp16 layout of the wide kernel for two-hidden-layer nets of width 65..256 (embed_panel.hip)
        if (ht >= 8 && Lf == 2 && Lh == 2) {
            std::vector<float> ps;
            size_t pf[8], ph[8];
            float isf_f = 1.f, sv_f = 1.f, isf_h = 1.f, sv_h = 1.f;
            const bool okf = pack_panel(fn, nx + nu, nx, hidden, ht, padout(nx), ps, pf, &isf_f, &sv_f);
            const bool okh = pack_panel(hn, nx, ny, hidden, ht, padout(ny), ps, ph, &isf_h, &sv_h);
            NMPC_HIP_CHECK(hipMalloc(&m->d.panel_slab, ps.size() * sizeof(float)));
            NMPC_HIP_CHECK(hipMemcpy(m->d.panel_slab, ps.data(), ps.size() * sizeof(float), hipMemcpyHostToDevice));
            auto pbind = [&](PanelNet& w, const WideNet& wn, const size_t* o, int nout_pad, float isf, float sv, bool ok) {
                const float* b = m->d.panel_slab;
                w.fwd = reinterpret_cast<const _Float16*>(b + o[0]); w.rev = reinterpret_cast<const _Float16*>(b + o[1]);
                w.w0t = reinterpret_cast<const _Float16*>(b + o[2]); w.w0g = b + o[3]; w.cb0 = b + o[4]; w.cb2s = b + o[5];
                w.w4s = b + o[6]; w.oscale = b + o[7]; w.lin = wn.lin; w.j0 = wn.j0; w.inv_sf = isf; w.s_v = sv;
                w.ht = ht; w.nout_pad = nout_pad; w.ok = ok ? 1 : 0;
            };
            pbind(m->d.panel_f, m->d.wide_f, pf, padout(nx), isf_f, sv_f, okf);
            pbind(m->d.panel_h, m->d.wide_h, ph, padout(ny), isf_h, sv_h, okh);
        }
    }
    eval_net_at_zero(fn, Lf, nx, hidden, m->d.f0);
    eval_net_at_zero(hn, Lh, ny, hidden, m->d.h0);
    // plain-layout nets for the closed loop's observer / output step (one hidden unit per lane or per thread)
    const bool loop_ok = hidden == HID && nu == 1 && ny == 1 && nx >= 2 && nx <= 4 && Lf == 2 && Lh == 2;
    auto upload_plain = [&](const float* const* p, int n_in, int n_out, float** dst) -> int {
        const int sz[8] = {n_out * n_in, n_out, hidden * n_in, hidden, hidden * hidden, hidden, n_out * hidden, n_out};
        std::vector<float> e;
        for (int i = 0; i < 8; ++i) e.insert(e.end(), p[i], p[i] + sz[i]);
        for (int k = 0; k < hidden; ++k)            // W2 transposed, appended for lane-coalesced reads
            for (int i = 0; i < hidden; ++i) e.push_back(p[4][i * hidden + k]);
        NMPC_HIP_CHECK(hipMalloc(dst, e.size() * sizeof(float)));
        NMPC_HIP_CHECK(hipMemcpy(*dst, e.data(), e.size() * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    };
    (void)loop_ok;
    NMPC_REQUIRE(!enc || (Lf == 2 && Lh == 2 && nu == 1 && ny == 1 && nx >= 2),
                 "nmpc_model_create: an observer net needs f and h nets with two hidden layers, nu = ny = 1, nx = 2..8");
    if (Lf == 2 && Lh == 2 && nu == 1 && ny == 1) {      // closed loops: output estimate; the model as its own plant
        if (upload_plain(hn, nx, ny, &m->d.hplain)) return 1;
        if (upload_plain(fn, nx + nu, nx, &m->d.fplain)) return 1;
    }
    if (enc && upload_plain(enc, n_enc_in, nx, &m->d.enc)) return 1;
    if (!narrow) {           // no 64-wide packs: the general-width kernel serves the embedding
        return 0;
    }
    std::vector<float> buf;
    pack_net(fn, nx + nu, nx, buf);
    NMPC_HIP_CHECK(hipMalloc(&m->d.fpack, PACK_FLOATS * sizeof(float)));
    NMPC_HIP_CHECK(hipMemcpy(m->d.fpack, buf.data(), PACK_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    pack_net(hn, nx, ny, buf);
    NMPC_HIP_CHECK(hipMalloc(&m->d.hpack, PACK_FLOATS * sizeof(float)));
    NMPC_HIP_CHECK(hipMemcpy(m->d.hpack, buf.data(), PACK_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    // nets whose weights do not suit the split-fp16 fragments run on the fp32-MFMA kernel (both null)
    std::vector<float> buf2;
    if (pack_net_h2(fn, nx + nu, nx, buf) && pack_net_h2(hn, nx, ny, buf2)) {
        NMPC_HIP_CHECK(hipMalloc(&m->d.fpack_h2, PACK_FLOATS * sizeof(float)));
        NMPC_HIP_CHECK(hipMemcpy(m->d.fpack_h2, buf.data(), PACK_FLOATS * sizeof(float), hipMemcpyHostToDevice));
        NMPC_HIP_CHECK(hipMalloc(&m->d.hpack_h2, PACK_FLOATS * sizeof(float)));
        NMPC_HIP_CHECK(hipMemcpy(m->d.hpack_h2, buf2.data(), PACK_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int nmpc_model_create_layers(nmpc_model** out, int nx, int nu, int ny, int hidden, int f_hidden_layers,
                                        int h_hidden_layers, int n_enc_in, const float* const* fn,
                                        const float* const* hn, const float* const* enc) {
    NMPC_REQUIRE(out && fn && hn, "nmpc_model_create: null argument");
    NMPC_REQUIRE(hidden >= 1 && hidden <= 256, "nmpc_model_create: hidden width %d unsupported (1..256)", hidden);
    NMPC_REQUIRE(f_hidden_layers >= 1 && f_hidden_layers <= 4 && h_hidden_layers >= 1 && h_hidden_layers <= 4,
                 "nmpc_model_create: hidden layer counts %d / %d unsupported (1..4)", f_hidden_layers, h_hidden_layers);
    NMPC_REQUIRE(hidden <= 128 || (f_hidden_layers <= 2 && h_hidden_layers <= 2),
                 "nmpc_model_create: more than two hidden layers need hidden <= 128 (got %d)", hidden);
    NMPC_REQUIRE(nx >= 1 && nu >= 1 && ny >= 1 && nx <= 8 && nx + nu <= 16 && ny <= 8,
                 "nmpc_model_create: dims nx=%d nu=%d ny=%d unsupported (need nx<=8, nx+nu<=16, ny<=8)", nx, nu, ny);
    NMPC_REQUIRE(!enc || (n_enc_in >= 1 && n_enc_in <= 32), "nmpc_model_create: bad encoder input size %d",
                 n_enc_in);
    nmpc_model* m = new nmpc_model();
    std::memset(m, 0, sizeof(*m));
    if (model_fill(m, nx, nu, ny, hidden, f_hidden_layers, h_hidden_layers, n_enc_in, fn, hn, enc)) {
        nmpc_model_destroy(m);          // frees whatever was uploaded before the failure
        return 1;
    }
    *out = m;
    return 0;
}

extern "C" int nmpc_model_create(nmpc_model** out, int nx, int nu, int ny, int hidden, int n_enc_in,
                                 const float* const* fn, const float* const* hn,
                                 const float* const* enc) {
    return nmpc_model_create_layers(out, nx, nu, ny, hidden, 2, 2, n_enc_in, fn, hn, enc);
}

extern "C" void nmpc_model_destroy(nmpc_model* m) {
    if (!m) return;
    if (m->d.fpack) (void)hipFree(m->d.fpack);
    if (m->d.hpack) (void)hipFree(m->d.hpack);
    if (m->d.fpack_h2) (void)hipFree(m->d.fpack_h2);
    if (m->d.hpack_h2) (void)hipFree(m->d.hpack_h2);
    if (m->d.enc) (void)hipFree(m->d.enc);
    if (m->d.hplain) (void)hipFree(m->d.hplain);
    if (m->d.fplain) (void)hipFree(m->d.fplain);
    if (m->d.wide_slab) (void)hipFree(m->d.wide_slab);
    if (m->d.panel_slab) (void)hipFree(m->d.panel_slab);
    delete m;
}

extern "C" int nmpc_quad_nodes(int rule, int stages, double dlam, float* lam_out, float* w_out) {
    if (stages < 1 || !lam_out || !w_out) { set_error("nmpc_quad_nodes: bad argument"); return -1; }
    if (dlam <= 0) dlam = 1.0 / stages;
    std::vector<double> nodes, weights;
    auto add = [&](double l, double w) {
        for (size_t i = 0; i < nodes.size(); ++i)
            if (nodes[i] == l) { weights[i] += w; return; }
        nodes.push_back(l); weights.push_back(w);
    };
    double lam = 0;
    for (int i = 0; i < stages; ++i) {
        if (rule == NMPC_RULE_SIMPSON) {
            const double w = dlam * 1 / 6;
            add(lam, w); add(lam + dlam / 2, 4 * w); add(lam + dlam, w);
        } else if (rule == NMPC_RULE_TRAP) {
            const double w = dlam * 1 / 2;
            add(lam, w); add(lam + dlam, w);
        } else { set_error("nmpc_quad_nodes: unknown rule %d", rule); return -1; }
        lam = lam + dlam;
    }
    if (nodes.size() > NMPC_MAX_NODES) { set_error("nmpc_quad_nodes: %zu nodes exceed %d", nodes.size(), NMPC_MAX_NODES); return -1; }
    for (size_t i = 0; i < nodes.size(); ++i) { lam_out[i] = (float)nodes[i]; w_out[i] = (float)weights[i]; }
    return (int)nodes.size();
}
