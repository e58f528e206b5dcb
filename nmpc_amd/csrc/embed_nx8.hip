// Split-fp16 embedding kernels for nets with 8 states (hidden width up to 64, one input, one or two outputs).
#include "embed_kernel.h"

namespace nmpc {

int embed_dispatch_nx8(int ny, const EmbedArgs& a, const NodeTable& nt, int grid, size_t lds_bytes, int mode, bool loop, hipStream_t s) {
    if (ny == 1) {
        if (mode == 2) return loop ? embed_launch_t<8, 1, 1, true, 2>(a, nt, grid, lds_bytes, s) : embed_launch_t<8, 1, 1, false, 2>(a, nt, grid, lds_bytes, s);
        return loop ? embed_launch_t<8, 1, 1, true, 1>(a, nt, grid, lds_bytes, s) : embed_launch_t<8, 1, 1, false, 1>(a, nt, grid, lds_bytes, s);
    }
    if (mode == 2) return embed_launch_t<8, 1, 2, false, 2>(a, nt, grid, lds_bytes, s);
    return embed_launch_t<8, 1, 2, false, 1>(a, nt, grid, lds_bytes, s);
}

}  // namespace nmpc
