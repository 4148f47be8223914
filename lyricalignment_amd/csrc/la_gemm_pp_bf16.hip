// la_gemm_pp_bf16.hip -- the 256 x 256 GEMM kernel (la_gemm_pp_kernel.h) instantiated for bfloat16 operands.
#include "la_gemm_pp_kernel.h"

namespace la {
namespace gemm {

int launch_pp_bf16(GemmParams p, int batch, bool out_f32, hipStream_t stream) {
    return out_f32 ? launch_pp<true, bf16_t>(p, batch, stream) : launch_pp<false, bf16_t>(p, batch, stream);
}

int launch_split_bf16(GemmParams p, int batch, hipStream_t stream) { return launch_split<bf16_t>(p, batch, stream); }

}  // namespace gemm
}  // namespace la

#ifdef LA_TILE_STAMPS
extern "C" int la_debug_set_tile_stamps(void *buf) {
    const int rc = la::gemm::set_tile_stamps_here(buf);
#ifdef LA_EXPERIMENTS
    if (rc == LA_OK) return la::gemm::lab_set_tile_stamps(buf);
#endif
    return rc;
}
#endif
