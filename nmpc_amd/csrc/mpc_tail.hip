// Tail of the closed-loop step (reference: the LPV re-scheduling loop of NMPC_test.py:154-260).  The first passes of a
// step run as batch-wide launches {embedding, solve} over the work list of controllers that have not converged; after
// two passes that list is almost always empty (the reference's recorded run: 1 pass 313 steps, 2 passes 135, 3 passes
// 2 of 450), but only the device knows.  Instead of two launches per remaining pass, ONE launch handles all of them:
// workgroup w takes controller act[it0][w] and loops {embed its Nc scheduling points with the tile of embed_tile.h
// (f net, then h net, in the workgroup's LDS), solve on wave 0 with the body of mpc_solve_kernel} until the stop
// test passes or max_iterations is reached.  The arithmetic is that of the batch-wide kernels, point for point and
// controller for controller, so the logs do not depend on where a pass ran.
#include "../../include/nmpc_amd.h"
#include "common.h"
#include "embed.h"
#include "embed_tile.h"
#include "mpc_kernels.h"
#include "mpc_tail.h"

namespace nmpc {

constexpr int TAIL_GRID = 64;     // workgroups of a tail launch (each walks the work list with this stride)

struct TailArgs {
    const float* fpack;
    const float* hpack;
    int it0;
};

template <int NX, int NU, int NY, int MPL, int NC, bool RECUR>
__global__ __launch_bounds__(EMBED_THREADS, 1) void mpc_tail_kernel(CtrlDev d, RunPtrs rp, TailArgs t, NodeTable nt, int k) {
    extern __shared__ __align__(16) float tail_lds[];
    __shared__ int more_s;
    const int count = min(d.nact[t.it0], d.B);
    const int Nc = NC ? NC : d.Nc;                    // NC = 0: horizon known at run time (as the batch-wide solve kernel of that shape)
    const int tiles = (Nc + EMBED_TILE - 1) / EMBED_TILE;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // a small grid walks the work list: the launch is empty most of the time, and a workgroup of this kernel needs a
    // whole CU's registers, so every workgroup that is only dispatched to exit would still wait for one
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
    const int* slot = d.act + (size_t)t.it0 * d.B + w;
    const int b = *slot;
    for (int it = t.it0; it < d.maxit; ++it) {
        // the scheduling points of this pass were written by the solve of the previous one (another kernel for
        // it0, wave 0 of this workgroup afterwards): complete the stores, drop this CU's cached lines
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll 1
        for (int tile = 0; tile < tiles; ++tile) {
            embed_tile_h2<NX, NU, NX, RECUR>(t.fpack, tail_lds, d.xs32, d.us32, d.A32, d.B32, tile, Nc, slot, Nc, nt);
            __syncthreads();
        }
#pragma unroll 1
        for (int tile = 0; tile < tiles; ++tile) {
            embed_tile_h2<NX, 0, NY, RECUR>(t.hpack, tail_lds, d.xc32, nullptr, d.C32, nullptr, tile, Nc, slot, Nc, nt);
            __syncthreads();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (wave == 0) {
            const bool more = mpc_solve_body<64, MPL, NX, NC>(d, rp, it, k, b, reinterpret_cast<double*>(tail_lds), lane);
            if (lane == 0) more_s = more ? 1 : 0;
        }
        __syncthreads();
        if (!more_s) break;
    }
    __syncthreads();                                  // more_s and the LDS are reused by the next controller
    }
}

static size_t tail_lds_bytes(const CtrlDev& d, const ModelDev& m) {
    const int nout_max = m.nx > m.ny ? m.nx : m.ny;
    const size_t e = sizeof(float) * (4096 + 4096 * nout_max + PK_TAIL);
    const int n = mpc_nvar(d.Nc, d.c.slack), mt = mpc_nrows(d.Nc, d.nx, d.c.mode);
    const size_t q = sizeof(double) * (size_t)mpc_lds_doubles(d.Nc, d.nx, n, mt);
    return e > q ? e : q;
}

// Output feedback with the slack variable, nu = ny = 1, two or three states (whole-net tiles), horizons of the
// one-wavefront register solve with one constraint row per lane (Nc <= 15).
bool mpc_tail_applies(const CtrlDev& d, const ModelDev& m, const MpcGeom& g) {
    return m.narrow_ok && m.fpack_h2 && m.hpack_h2 && embed_split16_selected() && (d.nx == 2 || d.nx == 3) && m.nu == 1 && m.ny == 1 &&
           d.Nc <= 15 && d.c.mode == 0 && d.c.slack && g.GS == 64 && g.mpl <= 1;
}

template <int NX, int NC, bool RECUR>
static int tail_launch_t(const CtrlDev& d, const RunPtrs& rp, const TailArgs& t, const NodeTable& nt, int k, size_t lds,
                         hipStream_t s) {
    static std::atomic<unsigned long long> once{0};
    if (first_use_on_device(once)) {
        NMPC_HIP_CHECK(hipFuncSetAttribute((const void*)mpc_tail_kernel<NX, 1, 1, 1, NC, RECUR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));   // beside 16 B static
        mark_used_on_device(once);
    }
    const int grid = d.B < TAIL_GRID ? d.B : TAIL_GRID;
    hipLaunchKernelGGL((mpc_tail_kernel<NX, 1, 1, 1, NC, RECUR>), dim3(grid), dim3(EMBED_THREADS), lds, s, d, rp, t, nt, k);
    NMPC_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_mpc_tail(const CtrlDev& d, const RunPtrs& rp, const ModelDev& m, const NodeTable& nt, int it0, int k, hipStream_t s) {
    NMPC_REQUIRE(nt.prepared == 1 && it0 >= 1 && it0 < d.maxit, "tail launch: bad arguments");
    TailArgs t{m.fpack_h2, m.hpack_h2, it0};
    const size_t lds = tail_lds_bytes(d, m);
    // the solve body is the instantiation the batch-wide kernel of the same shape uses (mpc_nx2.hip, mpc_nx3.hip), so the
    // logs do not depend on where a pass ran
    if (d.nx == 2 && d.Nc == 10) return nt.recur ? tail_launch_t<2, 10, true>(d, rp, t, nt, k, lds, s) : tail_launch_t<2, 10, false>(d, rp, t, nt, k, lds, s);
    if (d.nx == 2) return nt.recur ? tail_launch_t<2, 0, true>(d, rp, t, nt, k, lds, s) : tail_launch_t<2, 0, false>(d, rp, t, nt, k, lds, s);
    return nt.recur ? tail_launch_t<3, 0, true>(d, rp, t, nt, k, lds, s) : tail_launch_t<3, 0, false>(d, rp, t, nt, k, lds, s);
}

}  // namespace nmpc
