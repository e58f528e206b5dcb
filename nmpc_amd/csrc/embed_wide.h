// Shared pieces of the general-width embedding kernels (embed_wide.hip: two hidden layers; embed_deep.hip: any
// other number of hidden layers).
#pragma once
#include "embed.h"

namespace nmpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WideArgs {
    WideNet net;
    const float* pa;   // [rows, na]
    const float* pb;   // [rows, nb] (may be null when nb = 0)
    float* oa;         // [rows, nout, na]
    float* ob;         // [rows, nout, nb]
    int na, nb, nout;
    int npts;
    const int* count_dev;
    const int* act;
    int group;
};

// A-fragment loads go through buffer instructions: one SGPR descriptor per matrix, the lane's byte
// offset in one VGPR and the fragment's (compile-time) offset in the scalar operand.  Flat global loads
// would need a 64-bit VGPR address per fragment - 256 of them per matrix at hidden width 256.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const float* p, int nfloats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, nfloats * 4, 0x00020000);
}
__device__ __forceinline__ f32x4 frag_load(__amdgpu_buffer_rsrc_t r, int lane_bytes, int frag) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, frag * 1024, 0));
}

__device__ __forceinline__ float sigm2(float a) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-a)); }

// embed_deep.hip: nets with 1, 3 or 4 hidden layers
int deep_launch_net(const WideNet& net, WideArgs a, const NodeTable& nt, hipStream_t s, bool loop);

}  // namespace nmpc
