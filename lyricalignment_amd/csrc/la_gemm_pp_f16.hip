// la_gemm_pp_f16.hip -- the 256 x 256 GEMM kernel (la_gemm_pp_kernel.h) instantiated for IEEE half operands.
#ifdef LA_TILE_STAMPS
#undef LA_TILE_STAMPS      // (the tile-timeline diagnostic build stamps the bfloat16 kernels only)
#endif
#include "la_gemm_pp_kernel.h"

namespace la {
namespace gemm {

int launch_pp_f16(GemmParams p, int batch, bool out_f32, hipStream_t stream) {
    return out_f32 ? launch_pp<true, la::f16_t>(p, batch, stream) : launch_pp<false, la::f16_t>(p, batch, stream);
}

int launch_split_f16(GemmParams p, int batch, hipStream_t stream) { return launch_split<la::f16_t>(p, batch, stream); }

// f16x2 products (la_gemm_f16x2): segmented K over the (hi, lo) planes, f32 out, row / column scale epilogue (LNM 6)
int launch_x2_f16(GemmParams p, int batch, hipStream_t stream) { return launch_pp_loop<true, true, la::f16_t, 6>(p, batch, stream); }

}  // namespace gemm
}  // namespace la
