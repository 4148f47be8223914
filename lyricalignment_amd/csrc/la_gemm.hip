// la_gemm.hip -- C = epi(A W^T): every nn.Linear / nn.Conv1d-as-GEMM of the encoder and
// the GRU input projections (whisper.model.AudioEncoder; module/align_model.py:23-33).
// Main loop: la_gemm_core.h.  bf16 operands use v_mfma_f32_16x16x32_bf16, f32 operands
// v_mfma_f32_16x16x4_f32 (exact fmaf chains: the parity mode).
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "la_gemm_core.h"
#include "la_gemm_pp.h"

using la::bf16_t;
using namespace la::gemm;

namespace {

struct GemmParams {
    int M, N, K;
    const void *A;
    int64_t lda, strideA;
    const void *W;
    int64_t ldw, strideW;
    void *C;
    int64_t ldc, strideC;
    const float *bias;
    int64_t strideBias;
    const float *residual;
    int64_t ldr, strideR;
    int epilogue;
    int tiles_m, tiles_n, group;
    int mblock = 0;                    // ping-pong kernel: row tiles per M block of the tile order (tile_coord_mb; 0 = column groups over all of M)
    // LayerNorm folded into the GEMMs around it (ping-pong kernel only; la_gemm_fused_ln):
    void *C2 = nullptr;                // producer: second, 16-bit copy of the f32 result rows (the next GEMM's raw A operand)
    int64_t ldc2 = 0, strideC2 = 0;
    const float *ln_stats = nullptr;   // consumer: per-row (mean, rstd) of the raw A rows, [M][2]
    const float *ln_csum = nullptr;    // consumer: c[n] = sum_k W'[n][k] of the gamma-folded weights, [N]
    int K_tail = 0;                    // split-K: K of the LAST batch slot when the chunks are uneven (0 = p.K)
    float *ln_part = nullptr;          // producer (optional): per-row partial statistics of the 16-bit copy, [N/64][M][2] =
                                       // (mean, sum of squared deviations) of each 64-column segment (la_ln_stats_finalize)
};

template <typename T, bool OUT_F32, typename CF, bool TA = false, bool TW = false>
__global__ __launch_bounds__(CF::THREADS, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int m0 = tc.tm * CF::TM, n0 = tc.tn * BN;
    const int z = blockIdx.y;
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)z * p.strideA;
    const T *W = reinterpret_cast<const T *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    f32x4 acc[4][4];
    mainloop<T, CF, TA, TW>(A, p.lda, p.M, W, p.ldw, p.N, (p.K_tail && z == (int)gridDim.y - 1) ? p.K_tail : p.K, m0, n0, lds, acc);

    typedef typename std::conditional<OUT_F32, float, T>::type TC;
    TC *C = reinterpret_cast<TC *>(p.C) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool do_mish = p.epilogue & LA_EPI_MISH;
    const bool do_res = (p.epilogue & LA_EPI_RESIDUAL) && R;

    // bias + activation in registers (lane: 4 consecutive columns n of row m); the activation switches are
    // wave-uniform and hoisted so each variant is a straight-line pass over the 64 accumulators
    if (has_bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + q * 4;
            float b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = bias[min(n + j, p.N - 1)];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mi][ni][j] += b4[j];
        }
    }
    if (do_gelu) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (!OUT_F32 && sizeof(T) == 2) {      // result is rounded to 16 bits: the same form as the 256x256 kernel's
                    if (p.epilogue & 4096) {                     // (LA_GELU_PK=1: the erfc form on the packed pipe)
                        const la::f32x2 lo = la::gelu_pk(la::f32x2{acc[mi][ni][0], acc[mi][ni][1]});
                        const la::f32x2 hi = la::gelu_pk(la::f32x2{acc[mi][ni][2], acc[mi][ni][3]});
                        acc[mi][ni] = f32x4{lo.x, lo.y, hi.x, hi.y};
                    } else {
                        if (p.epilogue & 8192) {                 // developer A/B (LA_GELU_PK=2): one value at a time
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[mi][ni][j] = la::gelu_sig(acc[mi][ni][j]);
                        } else {
                            const la::f32x2 lo = la::gelu_sig2(la::f32x2{acc[mi][ni][0], acc[mi][ni][1]});
                            const la::f32x2 hi = la::gelu_sig2(la::f32x2{acc[mi][ni][2], acc[mi][ni][3]});
                            acc[mi][ni] = f32x4{lo.x, lo.y, hi.x, hi.y};
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mi][ni][j] = la::gelu_erf(acc[mi][ni][j]);
                }
            }
    } else if (do_mish) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mi][ni][j] = la::mish(acc[mi][ni][j]);
    }

    // Store path.  Each wave transposes its 64x64 tile through LDS (two passes of 32 rows, f32, row pitch 272 B:
    // conflict-free b128 writes and reads) so that 16 lanes cover 64 CONSECUTIVE columns of one row: full 128-B
    // (bf16) / 256-B (f32) line segments per row instead of 32-B pieces scattered over 16 rows (measured: the
    // scattered form ran the C write at 1.4 TB/s and cost 30-45 % of the kernel).
    constexpr int PITCH = 272;
    __syncthreads();  // every wave is done reading the operand stages
    unsigned char *reg = lds + wave * (32 * PITCH);
    const int wrow0 = m0 + wm * 64, wcol0 = n0 + wn * 64;
    const bool fast_c = ((p.ldc * (int64_t)sizeof(TC)) % 16 == 0) && ((uintptr_t)C % 16 == 0);
    const bool fast_r = do_res && (p.ldr % 4 == 0) && ((uintptr_t)R % 16 == 0);
    const bool probe_nostore = p.epilogue & 256;  // developer probe: main loop without the C stores
    // interior wave tiles: straight-line form (see the ping-pong kernel's epilogue)
    if (wrow0 + 64 <= p.M && wcol0 + 64 <= p.N && fast_c && (!do_res || fast_r) && !probe_nostore) {
        TC *cw = C + (int64_t)wrow0 * p.ldc + wcol0 + r * 4;
        const float *rw = do_res ? R + (int64_t)wrow0 * p.ldr + wcol0 + r * 4 : nullptr;
        auto fast = [&](auto resc) {
            constexpr bool RES = decltype(resc)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 t[8];
                if constexpr (RES) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) t[it] = *reinterpret_cast<const float4 *>(rw + (int64_t)(h * 32 + it * 4 + q) * p.ldr);
                }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rl = it * 4 + q;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16);
                    if constexpr (RES) { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                    TC *c = cw + (int64_t)(h * 32 + rl) * p.ldc;
                    if constexpr (sizeof(TC) == 4) {
                        *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        *reinterpret_cast<ushort4 *>(c) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        };
        if (do_res) fast(std::true_type{}); else fast(std::false_type{});
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + q;                       // row inside this 32-row pass
            const int m = wrow0 + h * 32 + rl;
            const int n = wcol0 + r * 4;                     // 4 consecutive columns
            f32x4 v = *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16);
            if (m >= p.M || n >= p.N) continue;
            if (probe_nostore && v[0] != 12345.678f) continue;
            const int nv = min(4, p.N - n);
            if (do_res) {
                const float *rr = R + (int64_t)m * p.ldr + n;
                if (fast_r && nv == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(rr);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else {
                    for (int j = 0; j < nv; ++j) v[j] += rr[j];
                }
            }
            TC *c = C + (int64_t)m * p.ldc + n;
            if (fast_c && nv == 4) {
                if constexpr (sizeof(TC) == 4) {
                    *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *reinterpret_cast<ushort4 *>(c) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                }
            } else {
                for (int j = 0; j < nv; ++j) la::Elem<TC>::store(c + j, v[j]);
            }
        }
    }
}

// 256x256 ping-pong kernel (bf16 operands only): same epilogue contract as gemm_kernel.
// sum over the 16 lanes of a DPP row (every lane of the row gets it): quad_perm [1,0,3,2], [2,3,0,1], row_ror 4, row_ror 8
__device__ __forceinline__ float row16_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
    return x;
}
__device__ __forceinline__ float half_bits_to_f32(unsigned short b, bf16_t) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ float half_bits_to_f32(unsigned short b, la::f16_t) { return (float)__builtin_bit_cast(_Float16, b); }

// Partial LayerNorm statistics of one 64-column row segment held by the 16 lanes of a DPP row (4 columns each), taken from
// the ROUNDED values the next GEMM will read: exact two-pass inside the segment (mean, then squared deviations).
template <typename T16>
__device__ __forceinline__ float2 segment_stats(const ushort4 pk) {
    const float e0 = half_bits_to_f32(pk.x, T16{}), e1 = half_bits_to_f32(pk.y, T16{}), e2 = half_bits_to_f32(pk.z, T16{}),
                e3 = half_bits_to_f32(pk.w, T16{});
    const float mean = row16_sum((e0 + e1) + (e2 + e3)) * (1.0f / 64.0f);
    const float d0 = e0 - mean, d1 = e1 - mean, d2 = e2 - mean, d3 = e3 - mean;
    return make_float2(mean, row16_sum(fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3)));
}

// LNM: 0 = plain; 1 = producer of the LayerNorm fold (second, 16-bit copy of the f32 rows); 2 = consumer (LayerNorm epilogue).
// Separate instantiations: one body with run-time switches for all three spilled 40-48 VGPRs in every mode.
// DUO selects the main loop: true = the hand-placed flat stream (mainloop_duo_asm), false = the quadrant ping-pong (mainloop_pp,
// for K that is not a multiple of 128 or below 256; LA_PP_DBG=99 forces it: the bit-identical A/B partner).
// Epilogue of one wave's 128x64 output tile (rows wrow0.., columns wcol0..) held in the MFMA accumulator layout acc[mi][ni][j]
// = C[wrow0 + 16 mi + r][wcol0 + 16 ni + 4 q + j]: LayerNorm fold / bias, GELU, f32 residual, stores (wide, through the wave's
// own 32 x EPI_PITCH bytes of LDS at `reg`), the 16-bit copy and the per-segment row statistics.  bias_l / csum_l: the bias and
// the LN column sum of column wcol0 + lane, requested by the caller BEFORE its main loop.  Shared by the 8-wave ping-pong
// kernel (one call per wave) and the one-wave-per-SIMD kernel (two calls per wave, one per 64-column half).
constexpr int EPI_PITCH = 272;
// One row-major quad (4 consecutive columns of one row) through the epilogue arithmetic, in the order every GEMM kernel of this
// file uses: LayerNorm fold (rstd (acc - mean c) + b' as two FMAs) or bias, activation.  b4 / cs4: bias and LN column sums of
// the lane's four columns; st = (mean, rstd) of the row.
template <bool OUT_F32, int LNM>
__device__ __forceinline__ void epi_quad(f32x4 &v, const float (&b4)[4], const float (&cs4)[4], float2 st, bool has_bias, bool do_gelu, int epilogue) {
    if constexpr (LNM == 2) {
        // A held the RAW rows x (16-bit copy of the residual stream) and W the gamma-folded weights W' = gamma o W:
        // LN(x) W^T + b = rstd (x W'^T - mean c) + b'
        const float rs = st.y, bm = -st.x * st.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], rs, fmaf(bm, cs4[j], b4[j]));
    } else if (has_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b4[j];
    }
    if (do_gelu) {
        if constexpr (!OUT_F32) {                        // result is rounded to 16 bits: the 11-slot sigmoid form
            if (epilogue & 4096) {                       // developer A/B (LA_GELU_PK=1): the erfc form on the packed pipe
                const la::f32x2 lo = la::gelu_pk(la::f32x2{v[0], v[1]}), hi = la::gelu_pk(la::f32x2{v[2], v[3]});
                v = f32x4{lo.x, lo.y, hi.x, hi.y};
            } else if (epilogue & 8192) {                // developer A/B (LA_GELU_PK=2): one value at a time
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = la::gelu_sig(v[j]);
            } else {
                const la::f32x2 lo = la::gelu_sig2(la::f32x2{v[0], v[1]}), hi = la::gelu_sig2(la::f32x2{v[2], v[3]});
                v = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = la::gelu_erf(v[j]);
        }
    }
}

// The wave's 128x64 tile leaves the accumulator layout FIRST: per pass of 32 rows the 8 accumulator tuples go through the wave's
// LDS region and come back row-major (lane (r, q) = columns 4r..4r+3 of row 4 it + q), and the whole epilogue arithmetic runs
// there, where a lane's four columns are fixed for the tile (bias / column sums: 8 registers, fetched from the one-column-per-
// lane values the caller requested before its main loop) and the pass's row operands (residual rows, LayerNorm statistics) are
// requested together before the staging.  Round 2 applied the LayerNorm fold and the GELU in the accumulator layout, before the
// staging: 16 row statistics + 32 broadcast column operands live beside the 128 accumulators -- the LayerNorm-consumer
// instantiations (QKV, MLP-up) sat at 256 VGPRs with 107-127 spilled registers and 112 B of scratch per lane.
// LNM = 4: as 2, with the row statistics taken by the main loop itself (mainloop_duo_asm STAT_WC) and left in LDS: stats_tab[row
// of the tile] = (mean, rstd), tile_m0 = the tile's first row.
// STG: how an interior wave tile is staged through LDS on its way to the row-major layout.  0 = passes of 32 rows, row pitch 272 B
// (8.5 KiB per wave at `reg`); 1 = passes of 16 rows in 4 KiB per wave, 256-byte rows with the 16-byte chunks XOR-swizzled by the row
// (chunk c of row r at c ^ r: conflict-free for the transposing b128 writes and the row-major b128 reads) -- the persistent kernel's
// form: 8 x 4 KiB = 32 KiB beside the four ring slots, so the NEXT tile's first stages can land in the ring under this epilogue.
// Edge wave tiles always take the element-wise path below with the 32-row staging.
template <bool OUT_F32, typename T16, int LNM, int STG = 0>
__device__ __forceinline__ void wave_epilogue(const GemmParams &p, int z, f32x4 (&acc)[8][4], int wrow0, int wcol0, bool has_bias,
                                              float bias_l, float csum_l, unsigned char *reg, const float2 *stats_tab = nullptr,
                                              int tile_m0 = 0, unsigned char *reg_edge = nullptr, float2 sr0 = float2{0.f, 1.f},
                                              float2 sr1 = float2{0.f, 1.f}, int lane_in = -1) {
    // LNM = 5 (persistent kernel: no LDS left for a statistics table): the statistics of the wave's 128 rows sit in the wave's own
    // registers -- sr0 = (mean, rstd) of row `lane`, sr1 of row 64 + lane -- and reach the lane that needs them by ds_bpermute.
    constexpr bool LNC = LNM == 2 || LNM == 4 || LNM == 5;   // LayerNorm consumer
    constexpr int NPASS = STG ? 8 : 4, NIT = STG ? 4 : 8, RP = STG ? 16 : 32;    // passes per wave tile, row quads and rows per pass
    // (lane_in: the persistent kernel hands in a lane id it has made opaque per tile, so that hipcc does not hoist this function's
    //  lane arithmetic out of the tile loop and keep it alive across the main loop, where every register is spoken for)
    const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
    const int r = lane & 15, q = lane >> 4;
    typedef typename std::conditional<OUT_F32, float, T16>::type TC;
    TC *C = reinterpret_cast<TC *>(p.C) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool do_res = (p.epilogue & LA_EPI_RESIDUAL) && R;
    const int epi = p.epilogue;
    // the lane's four columns 4r .. 4r+3 of the wave's 64: from the lanes that hold them (bias_l / csum_l = column `lane`)
    float b4[4], cs4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b4[j] = __int_as_float(__builtin_amdgcn_ds_bpermute((r * 4 + j) * 4, __float_as_int(bias_l)));
        cs4[j] = LNC ? __int_as_float(__builtin_amdgcn_ds_bpermute((r * 4 + j) * 4, __float_as_int(csum_l))) : 0.f;
    }
    const float2 *stats = reinterpret_cast<const float2 *>(p.ln_stats);      // LNM == 2: [M] rows in memory
    constexpr int PITCH = EPI_PITCH;
    const bool fast_c = ((p.ldc * (int64_t)sizeof(TC)) % 16 == 0) && ((uintptr_t)C % 16 == 0);
    const bool fast_r = do_res && (p.ldr % 4 == 0) && ((uintptr_t)R % 16 == 0);
    // Interior wave tiles (all but the last row / column of tiles): straight-line code, no bounds or alignment branches, so
    // the residual loads of a pass's rows are in flight together (a generic loop waits out one HBM round trip per row:
    // 32 dependent round trips per wave and tile, most of the 54 us the residual GEMMs once lost to their epilogue).
    if (wrow0 + 128 <= p.M && wcol0 + 64 <= p.N && fast_c && (!do_res || fast_r)) {
        TC *cw = C + (int64_t)wrow0 * p.ldc + wcol0 + r * 4;
        const float *rw = do_res ? R + (int64_t)wrow0 * p.ldr + wcol0 + r * 4 : nullptr;
        // producer: the 16-bit copy has the row pitch and batch stride of C (checked on the host), so one element offset serves both
        T16 *c2w = nullptr;
        float2 *part = nullptr;            // this wave's 128 rows of segment (wcol0 / 64): [N/64][M] (mean, M2) pairs
        if constexpr (LNM == 1 && OUT_F32) {
            c2w = reinterpret_cast<T16 *>(p.C2) + (int64_t)z * p.strideC + (int64_t)wrow0 * p.ldc + wcol0 + r * 4;
            if (p.ln_part) part = reinterpret_cast<float2 *>(p.ln_part) + (int64_t)(wcol0 >> 6) * p.M + wrow0;
        }
        auto fast = [&](auto resc) {
            constexpr bool RES = decltype(resc)::value;
#pragma unroll
            for (int h = 0; h < NPASS; ++h) {
                float4 t[NIT];
                float2 st[NIT];
                if constexpr (RES) {
                    if (epi & (1 << 16)) {                 // developer probe (LA_EPI_PROBE & 1): no residual loads
#pragma unroll
                        for (int it = 0; it < NIT; ++it) t[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) t[it] = *reinterpret_cast<const float4 *>(rw + (int64_t)(h * RP + it * 4 + q) * p.ldr);
                    }
                }
                if constexpr (LNM == 2) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) st[it] = stats[wrow0 + h * RP + it * 4 + q];
                }
                if constexpr (LNM == 4) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) st[it] = stats_tab[wrow0 - tile_m0 + h * RP + it * 4 + q];
                }
                if constexpr (LNM == 5) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int src = (((h * RP + it * 4) & 63) + q) * 4;
                        const float2 sr = h * RP >= 64 ? sr1 : sr0;
                        st[it] = make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sr.x))),
                                             __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sr.y))));
                    }
                }
                if constexpr (STG == 0) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *reinterpret_cast<f32x4 *>(reg + r * 256 + (((ni * 4 + q) ^ r) << 4)) = acc[h][ni];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int rl = it * 4 + q;
                    f32x4 v = STG == 0 ? *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16)
                                       : *reinterpret_cast<const f32x4 *>(reg + rl * 256 + ((r ^ rl) << 4));
                    epi_quad<OUT_F32, LNC ? 2 : LNM>(v, b4, cs4, LNC ? st[it] : make_float2(0.f, 0.f), has_bias, do_gelu, epi);
                    if constexpr (RES) { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                    const int64_t off = (int64_t)(h * RP + rl) * p.ldc;
                    TC *c = cw + off;
                    if constexpr (sizeof(TC) == 4) {
                        if (!(epi & (2 << 16)) || v[0] == 12345.678f)      // developer probe (LA_EPI_PROBE & 2): no f32 store
                            *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                        if constexpr (LNM == 1) {
                            const ushort4 pk = la::Pack4<T16>::run(v[0], v[1], v[2], v[3]);
                            if (!(epi & (4 << 16)) || v[1] == 12345.678f)  // developer probe (LA_EPI_PROBE & 4): no 16-bit copy
                                *reinterpret_cast<ushort4 *>(c2w + off) = pk;
                            if (part) {
                                const float2 sg = segment_stats<T16>(pk);
                                if (r == 0) part[h * RP + rl] = sg;
                            }
                        }
                    } else {
                        *reinterpret_cast<ushort4 *>(c) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        };
        if (do_res) fast(std::true_type{}); else fast(std::false_type{});
        return;
    }
    T16 *C2 = nullptr;
    if constexpr (LNM == 1 && OUT_F32) C2 = reinterpret_cast<T16 *>(p.C2) + (int64_t)z * p.strideC;
    if (reg_edge) reg = reg_edge;                            // (STG 1 callers: the 32-row staging lives elsewhere)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + q;
            const int m = wrow0 + h * 32 + rl;
            const int n = wcol0 + r * 4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16);
            float2 st = make_float2(0.f, 0.f);
            if constexpr (LNM == 2) st = stats[min(m, p.M - 1)];
            if constexpr (LNM == 4) st = stats_tab[wrow0 - tile_m0 + h * 32 + rl];
            if constexpr (LNM == 5) {
                const int src = (((h * 32 + it * 4) & 63) + q) * 4;
                const float2 sr = h * 32 >= 64 ? sr1 : sr0;
                st = make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sr.x))),
                                 __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sr.y))));
            }
            epi_quad<OUT_F32, LNC ? 2 : LNM>(v, b4, cs4, st, has_bias, do_gelu, epi);
            if (m >= p.M || n >= p.N) continue;
            const int nv = min(4, p.N - n);
            if (do_res) {
                const float *rr = R + (int64_t)m * p.ldr + n;
                if (fast_r && nv == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(rr);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else {
                    for (int j = 0; j < nv; ++j) v[j] += rr[j];
                }
            }
            TC *c = C + (int64_t)m * p.ldc + n;
            if (fast_c && nv == 4) {
                if constexpr (sizeof(TC) == 4) {
                    *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *reinterpret_cast<ushort4 *>(c) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                }
            } else {
                for (int j = 0; j < nv; ++j) la::Elem<TC>::store(c + j, v[j]);
            }
            if constexpr (LNM == 1) {
                for (int j = 0; j < nv; ++j) la::Elem<T16>::store(C2 + (int64_t)m * p.ldc + n + j, v[j]);
                if (p.ln_part) {                             // N % 64 == 0 on this path (host check): the 16 lanes of the row are all here
                    const float2 sg = segment_stats<T16>(la::Pack4<T16>::run(v[0], v[1], v[2], v[3]));
                    if (r == 0) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)(wcol0 >> 6) * p.M + m] = sg;
                }
            }
        }
    }
}

// Epilogue of the producers of the SPLIT residual stream (LNM = 3; la_gemm_split, la_common.h SplitRes): the wave's 128x64 tile
//   x = epi(acc) (+ an f32 residual: the stem's positional embedding | + the stream's own rows (hi, lo), updated in place)
// leaves as hi = x rounded to T16 (p.C2: the next GEMM's raw A operand) and lo = one byte per element (p.C) -- 3 + 3 bytes per
// element through HBM instead of the 4 + 4 + 2 of wave_epilogue's f32 stream with a 16-bit copy.  Same staging (through the
// wave's LDS region, row-major quads), same order of operations on the f32 values as wave_epilogue<true, T16, 1>.
constexpr int LA_EPI_SPLIT_INPLACE = 1 << 20;    // internal: the residual is the split stream itself
constexpr int LA_EPI_Q4_PRIO = 1 << 23;          // internal (LA_GEMM_Q4_PRIO=1): gemm_q4_kernel's prologue / epilogue at wave priority 3
constexpr int LA_EPI_SPLIT_PASS32 = 1 << 21;     // internal (LA_EPI_SPLIT_PASS=32): the 32-row passes without the one-pass-ahead requests
template <typename T16, int STG = 0>
__device__ __forceinline__ void wave_epilogue_split(const GemmParams &p, int z, f32x4 (&acc)[8][4], int wrow0, int wcol0, bool has_bias,
                                                    float bias_l, unsigned char *reg, unsigned char *reg_edge = nullptr, int lane_in = -1) {
    constexpr int NPASS = STG ? 8 : 4, NIT = STG ? 4 : 8, RP = STG ? 16 : 32;    // as wave_epilogue
    const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63);
    const int r = lane & 15, q = lane >> 4;
    unsigned char *LO = reinterpret_cast<unsigned char *>(p.C) + (int64_t)z * p.strideC;
    T16 *HI = reinterpret_cast<T16 *>(p.C2) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool res_f32 = (p.epilogue & LA_EPI_RESIDUAL) && R;
    const bool res_split = p.epilogue & LA_EPI_SPLIT_INPLACE;
    const int epi = p.epilogue;
    float b4[4];
    const float cs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = __int_as_float(__builtin_amdgcn_ds_bpermute((r * 4 + j) * 4, __float_as_int(bias_l)));
    constexpr int PITCH = EPI_PITCH;
    const bool fast_c = (p.ldc % 4 == 0) && ((uintptr_t)LO % 4 == 0) && ((uintptr_t)HI % 8 == 0) && (p.strideC % 4 == 0);
    const bool fast_r = res_f32 && (p.ldr % 4 == 0) && ((uintptr_t)R % 16 == 0);
    if (wrow0 + 128 <= p.M && wcol0 + 64 <= p.N && fast_c && (!res_f32 || fast_r)) {
        const int64_t base = (int64_t)wrow0 * p.ldc + wcol0 + r * 4;
        unsigned char *low = LO + base;
        T16 *hiw = HI + base;
        const float *rw = res_f32 ? R + (int64_t)wrow0 * p.ldr + wcol0 + r * 4 : nullptr;
        float2 *part = p.ln_part ? reinterpret_cast<float2 *>(p.ln_part) + (int64_t)(wcol0 >> 6) * p.M + wrow0 : nullptr;
        auto fast = [&](auto rkc) {
            constexpr int RK = decltype(rkc)::value;             // 0: no residual, 1: f32 rows, 2: the split stream in place
            // The residual rows of a pass are requested ONE PASS AHEAD in the 16-row forms (two register sets of 12 = the 24 registers
            // the 32-row form holds at once): only the first pass waits out a memory round trip, the others find their rows there.
            constexpr int NSET = STG ? 2 : 1;
            float4 tt[NSET][NIT];
            ushort4 tth[NSET][NIT];
            unsigned ttl[NSET][NIT];
            auto request = [&](int h, auto setc) __attribute__((always_inline)) {
                constexpr int S = decltype(setc)::value;
                if constexpr (RK == 1) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) tt[S][it] = *reinterpret_cast<const float4 *>(rw + (int64_t)(h * RP + it * 4 + q) * p.ldr);
                }
                if constexpr (RK == 2) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int64_t off = (int64_t)(h * RP + it * 4 + q) * p.ldc;
                        tth[S][it] = *reinterpret_cast<const ushort4 *>(hiw + off);
                        ttl[S][it] = *reinterpret_cast<const unsigned *>(low + off);
                    }
                }
            };
            if constexpr (STG != 0) request(0, std::integral_constant<int, 0>{});
            la::gemm::static_for<0, NPASS>([&](auto hc) __attribute__((always_inline)) {
                constexpr int h = decltype(hc)::value;
                constexpr int CS = STG ? (h & 1) : 0;
                if constexpr (STG == 0) request(h, std::integral_constant<int, 0>{});
                else if constexpr (h + 1 < NPASS) request(h + 1, std::integral_constant<int, (h + 1) & 1>{});
                float4 (&t)[NIT] = tt[CS];
                ushort4 (&th)[NIT] = tth[CS];
                unsigned (&tl)[NIT] = ttl[CS];
                if constexpr (STG == 0) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *reinterpret_cast<f32x4 *>(reg + r * 256 + (((ni * 4 + q) ^ r) << 4)) = acc[h][ni];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int rl = it * 4 + q;
                    f32x4 v = STG == 0 ? *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16)
                                       : *reinterpret_cast<const f32x4 *>(reg + rl * 256 + ((r ^ rl) << 4));
                    epi_quad<true, 0>(v, b4, cs4, make_float2(0.f, 0.f), has_bias, do_gelu, epi);
                    if constexpr (RK == 1) { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                    if constexpr (RK == 2) {
                        const unsigned w = tl[it];
                        v[0] += la::split_decode<T16>(th[it].x, (float)(w & 0xffu));
                        v[1] += la::split_decode<T16>(th[it].y, (float)((w >> 8) & 0xffu));
                        v[2] += la::split_decode<T16>(th[it].z, (float)((w >> 16) & 0xffu));
                        v[3] += la::split_decode<T16>(th[it].w, (float)(w >> 24));
                    }
                    float q0, q1, q2, q3;
                    ushort4 pk;
                    pk.x = la::split_encode<T16>(v[0], q0); pk.y = la::split_encode<T16>(v[1], q1);
                    pk.z = la::split_encode<T16>(v[2], q2); pk.w = la::split_encode<T16>(v[3], q3);
                    const int64_t off = (int64_t)(h * RP + rl) * p.ldc;
                    *reinterpret_cast<ushort4 *>(hiw + off) = pk;
                    *reinterpret_cast<unsigned *>(low + off) = la::pack_u8x4(q0, q1, q2, q3);
                    if (part) {
                        const float2 sg = segment_stats<T16>(pk);
                        if (r == 0) part[h * RP + rl] = sg;
                    }
                }
            });
        };
        if (res_split) fast(std::integral_constant<int, 2>{});
        else if (res_f32) fast(std::integral_constant<int, 1>{});
        else fast(std::integral_constant<int, 0>{});
        return;
    }
    // edge wave tiles (the last row of tiles of M = 48000 = 187.5 x 256, any unaligned call): element by element
    if (reg_edge) reg = reg_edge;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + q;
            const int m = wrow0 + h * 32 + rl;
            const int n = wcol0 + r * 4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16);
            epi_quad<true, 0>(v, b4, cs4, make_float2(0.f, 0.f), has_bias, do_gelu, epi);
            if (m >= p.M || n >= p.N) continue;
            const int nv = min(4, p.N - n);
            const int64_t off = (int64_t)m * p.ldc + n;
            for (int j = 0; j < nv; ++j) {
                float x = v[j];
                if (res_f32) x += R[(int64_t)m * p.ldr + n + j];
                if (res_split) x += la::split_decode<T16>(reinterpret_cast<const unsigned short *>(HI)[off + j], (float)LO[off + j]);
                float qf;
                reinterpret_cast<unsigned short *>(HI)[off + j] = la::split_encode<T16>(x, qf);
                LO[off + j] = (unsigned char)(la::pack_u8x4(qf, 0.f, 0.f, 0.f) & 0xffu);
                v[j] = x;
            }
            if (p.ln_part) {                                     // N % 64 == 0 (host check): the 16 lanes of the row are all here
                float qd;
                ushort4 pk;
                pk.x = la::split_encode<T16>(v[0], qd); pk.y = la::split_encode<T16>(v[1], qd);
                pk.z = la::split_encode<T16>(v[2], qd); pk.w = la::split_encode<T16>(v[3], qd);
                const float2 sg = segment_stats<T16>(pk);
                if (r == 0) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)(wcol0 >> 6) * p.M + m] = sg;
            }
        }
    }
}

#ifdef LA_TILE_STAMPS
// Diagnostic build only (tools/tile_timeline.py): per workgroup (wall clock at entry, after the prologue, after the main loop, at
// the end; HW_ID) into a buffer that nothing else reads -- where a tile's lifetime goes and how long a CU waits for its next one.
__device__ unsigned long long *g_tile_stamps = nullptr;
extern "C" int la_debug_set_tile_stamps(void *buf) {
    unsigned long long *b = static_cast<unsigned long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), &b, sizeof(b)) == hipSuccess ? LA_OK : LA_EHIP;
}
#endif

template <bool OUT_F32, bool DUO, typename T16, int LNM = 0>
__global__ __launch_bounds__(PP::THREADS, 2) void gemm_pp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_t1 = stamp_t0;
#endif
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord_mb(tile, p.tiles_m, p.tiles_n, p.group, p.mblock);
    const int m0 = tc.tm * PP::TM, n0 = tc.tn * PP::TN;
    const int z = blockIdx.y;
    const T16 *A = reinterpret_cast<const T16 *>(p.A) + (int64_t)z * p.strideA;
    const T16 *W = reinterpret_cast<const T16 *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    // Per-column epilogue operands are requested BEFORE the main loop, one column per lane (this wave's 64 columns), and
    // handed to the row-major epilogue layout with ds_bpermute afterwards: all eight waves reach the epilogue together, so a load
    // issued there is a fully exposed L2 round trip per tile.
    const int ncol = min(n0 + wc * 64 + lane, p.N - 1);
    const float bias_l = has_bias ? bias[ncol] : 0.f;
    float csum_l = 0.f;
    if constexpr (LNM == 2 || LNM == 4) csum_l = p.ln_csum[ncol];
    // LNM == 2: the tile's 256 row statistics are requested here too (32 rows per wave) and handed to the epilogue through LDS after
    // the main loop: loaded inside the epilogue they were one exposed L2 round trip per pass of 32 rows, four per tile.
    float2 st_pre = make_float2(0.f, 1.f);
    if constexpr (LNM == 2) {
        if (lane < 32) st_pre = reinterpret_cast<const float2 *>(p.ln_stats)[min(m0 + wr * 128 + wc * 32 + lane, p.M - 1)];
    }

    f32x4 acc[8][4];
#ifdef LA_TILE_STAMPS
#define LA_STAMP_ARG , stamp_t1
#else
#define LA_STAMP_ARG
#endif
    float sacc[4] = {0.f, 0.f, 0.f, 0.f};
    float2 *stats_tab = reinterpret_cast<float2 *>(lds + 120 * 1024);        // LNM == 4: inside ring slot 3, clear of the epilogue staging
    if constexpr (LNM == 4) {
        static_assert(DUO || LNM != 4, "the main loop takes the row statistics only in its hand-placed form");
        const int wc_u = __builtin_amdgcn_readfirstlane(wc);
        switch (wc_u) {                                                       // (the fragment registers are named at compile time)
            case 0: mainloop_duo_asm<T16, 0>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
            case 1: mainloop_duo_asm<T16, 1>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
            case 2: mainloop_duo_asm<T16, 2>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
            default: mainloop_duo_asm<T16, 3>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
        }
        // sum over the four lanes that hold one row's four k chunks, then (mean, rstd) of rows wr * 128 + (2 wc + i) * 16 + r
        const float inv_k = 1.0f / (float)p.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s1 = sacc[2 * i], s2 = sacc[2 * i + 1];
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float mean = s1 * inv_k;
            const float var = fmaxf(fmaf(-mean, mean, s2 * inv_k), 0.f);
            if (lane < 16) stats_tab[wr * 128 + (2 * wc + i) * 16 + lane] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
        }
    } else if constexpr (DUO) {
        mainloop_duo_asm<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG);
    } else {
        mainloop_pp<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc);
    }
    if constexpr (LNM == 2) {
        if (lane < 32) stats_tab[wr * 128 + wc * 32 + lane] = st_pre;       // (both main loops end behind a barrier: slot 3 is free)
    }
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t2 = __builtin_amdgcn_s_memrealtime();
#endif

    __syncthreads();
    if constexpr (LNM == 3) {
        // 16-row passes with the residual rows requested one pass ahead (default), or the 32-row passes (LA_EPI_SPLIT_PASS=32: A/B)
        if (p.epilogue & LA_EPI_SPLIT_PASS32) wave_epilogue_split<T16, 0>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, lds + wave * (32 * EPI_PITCH));
        else wave_epilogue_split<T16, 1>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, lds + wave * (32 * EPI_PITCH), lds + wave * (32 * EPI_PITCH));
    } else wave_epilogue<OUT_F32, T16, LNM == 2 ? 4 : LNM>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, csum_l, lds + wave * (32 * EPI_PITCH), stats_tab, m0);
#ifdef LA_TILE_STAMPS
    if (threadIdx.x == 0 && g_tile_stamps) {
        unsigned long long *o = g_tile_stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = stamp_t0; o[1] = stamp_t1; o[2] = stamp_t2; o[3] = __builtin_amdgcn_s_memrealtime();
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        o[4] = hw_id; o[5] = xcc_id;
        o[6] = tile;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Four-wave workgroups, two resident per CU (mainloop_q4_asm, la_gemm_pp.h; LA_GEMM_Q4=1): 256 x 128 tiles, the 8-wave kernel's wave
// tiles and epilogues.  One workgroup's prologue / epilogue / dispatch gap runs under the other's main loop.
template <bool OUT_F32, typename T16, int LNM = 0, int REM = 8, bool WIDE = false>
__global__ __launch_bounds__(Q4::THREADS, 2) void gemm_q4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_t1 = stamp_t0;
#endif
    // LA_GEMM_Q4_PRIO=1: prologue and epilogue at raised wave priority -- they are short, latency-bound phases that share the SIMDs
    // with the OTHER workgroup's main loop; at equal priority the timeline shows them stretched 2-4 x (tools/tile_timeline.py).
    const bool prio = p.epilogue & LA_EPI_Q4_PRIO;
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord_mb(tile, p.tiles_m, p.tiles_n, p.group, p.mblock);
    constexpr int TM = WIDE ? Q4::TN : Q4::TM, TN = WIDE ? Q4::TM : Q4::TN;      // 256 x 128, or 128 x 256 (WIDE)
    const int m0 = tc.tm * TM, n0 = tc.tn * TN;
    const int z = blockIdx.y;
    const T16 *A = reinterpret_cast<const T16 *>(p.A) + (int64_t)z * p.strideA;
    const T16 *W = reinterpret_cast<const T16 *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = WIDE ? 0 : wave >> 1, wc = WIDE ? wave : wave & 1;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    const int ncol = min(n0 + wc * 64 + lane, p.N - 1);          // per-column operands before the main loop (as gemm_pp_kernel)
    const float bias_l = has_bias ? bias[ncol] : 0.f;
    float csum_l = 0.f;
    if constexpr (LNM == 2) csum_l = p.ln_csum[ncol];
    float2 st_pre = make_float2(0.f, 1.f);
    constexpr int SROWS = TM / 4;                               // rows of the tile whose statistics this wave fetches
    if constexpr (LNM == 2) {
        if (lane < SROWS) st_pre = reinterpret_cast<const float2 *>(p.ln_stats)[min(m0 + wave * SROWS + lane, p.M - 1)];
    }

    f32x4 acc[8][4];
    mainloop_q4_asm<T16, REM, WIDE>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG);
    if (prio) __builtin_amdgcn_s_setprio(3);
    float2 *stats_tab = reinterpret_cast<float2 *>(lds + Q4::STATS);         // behind the ring: clear of the epilogue staging
    if constexpr (LNM == 2) {
        if (lane < SROWS) stats_tab[wave * SROWS + lane] = st_pre;
    }
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
    if constexpr (LNM == 3) {
        wave_epilogue_split<T16, 1>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, lds + wave * (32 * EPI_PITCH), lds + wave * (32 * EPI_PITCH));
    } else wave_epilogue<OUT_F32, T16, LNM == 2 ? 4 : LNM>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, csum_l, lds + wave * (32 * EPI_PITCH), stats_tab, m0);
#ifdef LA_TILE_STAMPS
    if (threadIdx.x == 0 && g_tile_stamps) {
        unsigned long long *o = g_tile_stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = stamp_t0; o[1] = stamp_t1; o[2] = stamp_t2; o[3] = __builtin_amdgcn_s_memrealtime();
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        o[4] = hw_id; o[5] = xcc_id;
        o[6] = tile;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the 256x256 kernel (round 4; hand-placed main loop, batch 1, 16-bit operands).
// tools/tile_timeline.py: 26-43 % of a K = 1024 tile's time on its CU is not the main loop -- 2.4-3.0 us of prologue (waiting for its
// first two stages), the epilogue, and 1.1-4.0 us until the next workgroup enters (a wave retires only when its stores are
// acknowledged).  Here a workgroup stays: it DRAWS tiles from a ticket counter (one per XCD, so an XCD keeps walking its own
// contiguous run of tile ids -- the L2 sharing of the hardware-dispatched form -- and steals from the next XCD's run when its own
// is empty), and before it starts a tile's epilogue it has already issued the NEXT tile's stages 0 .. 3 into the ring: they land
// under the epilogue, the epilogue's stores drain under the next main loop (duo_run PREFETCHED: its first waits are counted
// so that no store is forced), and nothing waits for a dispatch.  Dynamic tickets keep what the dispatcher gives for free: a
// workgroup that starts late (CUs held by the head stream's resident recurrence) simply draws fewer tiles.
//   * Ticket: wave 0 issues a returning atomic at the TOP of a tile's main loop and reads it at the end (in flight ~1 us, the loop
//     takes 25-100).  The result lands asynchronously, so it is parked in a register hipcc does not manage across that span: the
//     physical v255, named in both asm statements and as their clobber (the kernel's other values sit in v0 .. ~v240; the build
//     checks the assembly: v255 may not be written between the two).  An AGPR would be the natural home, but any AGPR use makes
//     hipcc split the 256-register budget 128 / 128 and spill ~550 registers.  Broadcast through one LDS word, two barriers.
//   * LDS: all 160 KiB -- ring slots [0, 128 K) and 8 x 4 KiB of epilogue staging behind them (wave_epilogue STG = 1); the LayerNorm
//     statistics travel in registers (LNM = 5).  A partial (edge) tile takes the element-wise epilogue with its 32-row staging at the
//     ring's front, so nothing is prefetched before it.
//   * Exit: the last workgroup to leave (a second counter) zeroes the tickets for the next launch on the stream.
struct PersistTickets { unsigned head[8]; unsigned done; unsigned pad[7]; };

template <bool OUT_F32, typename T16, int LNM>
__global__ __launch_bounds__(PP::THREADS, 2) void gemm_pp_persist_kernel(GemmParams p, PersistTickets *tk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int EM = LNM == 2 ? 5 : LNM;                   // epilogue form: LayerNorm statistics from the wave's registers
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(la::lds_addr_u32(lds));
    const int nt = p.tiles_m * p.tiles_n;
    const int q8 = nt >> 3, r8 = nt & 7;
    unsigned xcc_raw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_raw));
    int cur_x = (int)(xcc_raw & 7u);                         // the XCD whose run of tile ids this workgroup currently draws from
    int tried = 0;                                           // runs found empty so far
    int *tslot = reinterpret_cast<int *>(lds);               // ticket broadcast word (ring slot 0; free whenever it is used)
    const T16 *A = reinterpret_cast<const T16 *>(p.A);
    const T16 *W = reinterpret_cast<const T16 *>(p.W);
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && p.bias;

    // ticket k of run x -> tile id, or -1 and move on to the next run (wave 0 only; all values wave-uniform)
    auto resolve = [&](unsigned k) -> int {
        while (true) {
            const int len = q8 + (cur_x < r8 ? 1 : 0);
            if ((int)k < len) return (cur_x < r8 ? cur_x * (q8 + 1) : r8 * (q8 + 1) + (cur_x - r8) * q8) + (int)k;
            if (++tried >= 8) return -1;
            cur_x = (cur_x + 1) & 7;
            unsigned kk = 0;
            if ((tid & 63) == 0) kk = __hip_atomic_fetch_add(&tk->head[cur_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            k = __builtin_amdgcn_readfirstlane(kk);
        }
    };
    auto broadcast = [&](int v) -> int {                     // wave 0's value to every wave (the ring is free at both call sites)
        if (tid == 0) *tslot = v;
        __syncthreads();
        const int r = __builtin_amdgcn_readfirstlane(*tslot);
        __syncthreads();
        return r;
    };

    int t = 0;
    if (wave == 0) {
        unsigned kk = 0;
        if ((tid & 63) == 0) kk = __hip_atomic_fetch_add(&tk->head[cur_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = resolve(__builtin_amdgcn_readfirstlane(kk));
    }
    t = broadcast(t);

    la::gemm::DuoCtx c;
    float bias_l = 0.f, csum_l = 0.f;
    float2 sr0 = make_float2(0.f, 1.f), sr1 = make_float2(0.f, 1.f);
    int m0 = 0, n0 = 0;
    // everything a tile needs before its main loop: coordinates, DMA addresses, the stages 0 .. 3, the epilogue's per-column /
    // per-row operands (requested here, used after the loop)
    auto open_tile = [&](int tile, la::gemm::DuoCtx &cc, int &mm0, int &nn0, float &b_l, float &cs_l, float2 &s0, float2 &s1) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));                       // (keeps hipcc from hoisting a tile's lane arithmetic across the other tile's loop)
        const TileCoord tc = tile_coord_mb(tile, p.tiles_m, p.tiles_n, p.group, p.mblock);
        mm0 = tc.tm * PP::TM; nn0 = tc.tn * PP::TN;
        la::gemm::duo_setup<T16>(cc, A, p.lda, p.M, W, p.ldw, p.N, mm0, nn0, lds0, wave, lane);
        la::gemm::duo_issue_prologue(cc);
        const int ncol = min(nn0 + wc * 64 + lane, p.N - 1);
        b_l = has_bias ? p.bias[ncol] : 0.f;
        if constexpr (LNM == 2) {
            cs_l = p.ln_csum[ncol];
            const float2 *st = reinterpret_cast<const float2 *>(p.ln_stats);
            s0 = st[min(mm0 + wr * 128 + lane, p.M - 1)];
            s1 = st[min(mm0 + wr * 128 + 64 + lane, p.M - 1)];
        }
    };
    if (t >= 0) open_tile(t, c, m0, n0, bias_l, csum_l, sr0, sr1);
    bool prefetched = false;
    for (int guard = 0; t >= 0 && guard <= nt; ++guard) {
        if (wave == 0 && tried < 8) {                        // the next ticket: in flight during the main loop, parked in v255
            unsigned one = 1;
            unsigned *hp = &tk->head[cur_x];
            if ((tid & 63) == 0)
                asm volatile("global_atomic_add v255, %0, %1, off sc0" ::"v"(hp), "v"(one) : "memory", "v255");
        }
        f32x4 acc[8][4];
#ifdef LA_TILE_STAMPS
        const unsigned long long ps0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long ps1 = ps0;
        if (prefetched) la::gemm::duo_run<T16, -1, true>(c, p.K, acc, ps1);
        else la::gemm::duo_run<T16, -1, false>(c, p.K, acc, ps1);
        const unsigned long long ps2 = __builtin_amdgcn_s_memrealtime();
#else
        if (prefetched) la::gemm::duo_run<T16, -1, true>(c, p.K, acc);
        else la::gemm::duo_run<T16, -1, false>(c, p.K, acc);
#endif
        int tn = -1;
        if (wave == 0 && tried < 8) {
            unsigned kk = 0;
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v255" : "=v"(kk)::"memory", "v255");
            tn = resolve(__builtin_amdgcn_readfirstlane(kk));
        }
        tn = broadcast(tn);
        const bool interior = m0 + PP::TM <= p.M && n0 + PP::TN <= p.N;
        const bool chain = tn >= 0 && interior;
        la::gemm::DuoCtx cn;
        float bias_n = 0.f, csum_n = 0.f;
        float2 sn0 = make_float2(0.f, 1.f), sn1 = make_float2(0.f, 1.f);
        int mn0 = 0, nn0 = 0;
        if (chain) open_tile(tn, cn, mn0, nn0, bias_n, csum_n, sn0, sn1);      // its stages land under this tile's epilogue
        unsigned char *stg = lds + 4 * 32768 + wave * 4096, *stg_edge = lds + wave * (32 * EPI_PITCH);
#ifdef LA_TILE_STAMPS
        const unsigned long long ps3 = __builtin_amdgcn_s_memrealtime();
#endif
        int lane_e = tid & 63;
        asm volatile("" : "+v"(lane_e));
        if constexpr (LNM == 3) wave_epilogue_split<T16, 1>(p, 0, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, stg, stg_edge, lane_e);
        else wave_epilogue<OUT_F32, T16, EM, 1>(p, 0, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, csum_l, stg, nullptr, m0, stg_edge, sr0, sr1, lane_e);
#ifdef LA_TILE_STAMPS
        if (tid == 0 && g_tile_stamps) {       // per tile: top of the iteration, first fragments in, main loop done, epilogue begins / ends
            unsigned long long *o = g_tile_stamps + (size_t)t * 8;
            unsigned hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            o[0] = ps0; o[1] = ps1; o[2] = ps2; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = hw_id; o[5] = xcc_id; o[6] = ps3; o[7] = prefetched ? 1 : 0;
        }
#endif
        if (tn < 0) break;
        if (chain) {
            c = cn; m0 = mn0; n0 = nn0; bias_l = bias_n; csum_l = csum_n; sr0 = sn0; sr1 = sn1;
        } else {
            __syncthreads();                                 // an edge tile staged at the ring's front: everyone is done with it
            open_tile(tn, c, m0, n0, bias_l, csum_l, sr0, sr1);
        }
        prefetched = chain;
        t = tn;
    }
    // the last workgroup out re-arms the counters for the next launch on this stream
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&tk->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) __hip_atomic_store(&tk->head[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tk->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Epilogue of the one-wave-per-SIMD kernel: one wave's 128x128 tile, accumulators in the AGPRs.  Per pass of 32 rows the wave
// writes its accumulator tuples STRAIGHT from the AGPRs into its own 32 x 528 bytes of LDS (ds_write_b128 takes AGPR data: no
// v_accvgpr_read, no VGPR copy of the tile) and reads them back row-major -- lane (lane >> 5, lane & 31) = (row parity, four
// consecutive columns), two whole 512-byte rows per instruction -- where the LayerNorm fold / bias, GELU, the f32 residual,
// the stores, the 16-bit copy and the per-segment row statistics are applied in the same order and with the same operations
// as wave_epilogue (bit-identical results).  The lane's four columns are fixed, so its bias / column-sum values (b4, cs4) are
// loaded once by the caller before the main loop; the pass's residual rows and LayerNorm row statistics are requested before
// the staging, 16 rows in flight per lane.
constexpr int MONO_PITCH = 528;
template <int OFF> __device__ __forceinline__ void ds_write128_agpr(unsigned addr, const f32x4 &a) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "a"(a), "n"(OFF) : "memory");
}
template <bool OUT_F32, typename T16, int LNM>
__device__ __forceinline__ void mono_epilogue(const GemmParams &p, int z, f32x4 (&acc)[8][8], int wrow0, int wcol0, bool has_bias,
                                              const float (&b4)[4], const float (&cs4)[4], unsigned char *reg) {
    constexpr int PITCH = MONO_PITCH;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4, c = lane & 31, rp = lane >> 5;
    typedef typename std::conditional<OUT_F32, float, T16>::type TC;
    TC *C = reinterpret_cast<TC *>(p.C) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool do_res = (p.epilogue & LA_EPI_RESIDUAL) && R;
    const bool fast_c = ((p.ldc * (int64_t)sizeof(TC)) % 16 == 0) && ((uintptr_t)C % 16 == 0);
    const bool fast_r = do_res && (p.ldr % 4 == 0) && ((uintptr_t)R % 16 == 0);
    const bool full = wrow0 + 128 <= p.M && wcol0 + 128 <= p.N && fast_c && (!do_res || fast_r);     // wave-uniform
    const int n = wcol0 + c * 4;
    T16 *C2 = nullptr;
    if constexpr (LNM == 1 && OUT_F32) C2 = reinterpret_cast<T16 *>(p.C2) + (int64_t)z * p.strideC;
    const unsigned wr_addr = la::lds_addr_u32(reg) + (unsigned)(r * PITCH + q * 16);
    const unsigned char *rd = reg + rp * PITCH + c * 16;
    // the 16 accumulator tuples of pass h -> LDS (the AGPR names are compile-time: one arm per pass)
    auto stage = [&](auto hc) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        la::gemm::static_for<0, 16>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, mm = i >> 3, ni = i & 7;
            ds_write128_agpr<mm * 16 * PITCH + ni * 64>(wr_addr, acc[2 * h + mm][ni]);
        });
    };
    // one pass of 32 rows; the row code exists once per (FULL, RES) -- the passes are a run-time loop around it, so the whole
    // epilogue stays a few thousand instructions (fully unrolled it was 45 k: every wave then streams its code from L2)
    auto pass = [&](int h, auto fullc, auto resc) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(fullc)::value, RES = decltype(resc)::value;
        const int row0 = wrow0 + h * 32;
        float4 t[16];
        float2 st[16];
        // (row addresses by pointer increments: a 64-bit multiply per row costs more vector cycles than the row's arithmetic)
        if constexpr (FULL && RES) {
            const float *rr = R + (int64_t)(row0 + rp) * p.ldr + n;
            const int64_t rstep = 2 * p.ldr;
#pragma unroll
            for (int it = 0; it < 16; ++it) { t[it] = *reinterpret_cast<const float4 *>(rr); rr += rstep; }
        }
        const int64_t cstep = 2 * p.ldc;
        int64_t off = (int64_t)(row0 + rp) * p.ldc + n - cstep;
        if constexpr (LNM == 2) {
#pragma unroll
            for (int it = 0; it < 16; ++it) st[it] = reinterpret_cast<const float2 *>(p.ln_stats)[min(row0 + it * 2 + rp, p.M - 1)];
        }
        switch (h) {
            case 0: stage(std::integral_constant<int, 0>{}); break;
            case 1: stage(std::integral_constant<int, 1>{}); break;
            case 2: stage(std::integral_constant<int, 2>{}); break;
            default: stage(std::integral_constant<int, 3>{}); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(rd + it * 2 * PITCH);
            const int m = row0 + it * 2 + rp;
            off += cstep;
            if constexpr (LNM == 2) {
                const float rs = st[it].y, bm = -st[it].x * st[it].y;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], rs, fmaf(bm, cs4[j], b4[j]));
            } else if (has_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += b4[j];
            }
            if (do_gelu) {
                if constexpr (!OUT_F32) {
                    const la::f32x2 lo = la::gelu_sig2(la::f32x2{v[0], v[1]}), hi = la::gelu_sig2(la::f32x2{v[2], v[3]});
                    v = f32x4{lo.x, lo.y, hi.x, hi.y};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = la::gelu_erf(v[j]);
                }
            }
            if constexpr (FULL) {
                if constexpr (RES) { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                if ((p.epilogue & 2048) && v[0] != 12345.678f) continue;      // developer probe: the epilogue without its stores
                if constexpr (sizeof(TC) == 4) {
                    *reinterpret_cast<float4 *>(C + off) = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (LNM == 1) {
                        const ushort4 pk = la::Pack4<T16>::run(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<ushort4 *>(C2 + off) = pk;
                        if (p.ln_part) {
                            const float2 sg = segment_stats<T16>(pk);
                            if (r == 0) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)((wcol0 >> 6) + (c >> 4)) * p.M + m] = sg;
                        }
                    }
                } else {
                    *reinterpret_cast<ushort4 *>(C + off) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                }
            } else {
                // ragged tile / unaligned operands (wave_epilogue's generic path, same order of operations)
                const bool in = m < p.M && n < p.N;
                const int nv = in ? min(4, p.N - n) : 0;
                if (do_res && in) {
                    const float *rr = R + (int64_t)m * p.ldr + n;
                    if (fast_r && nv == 4) {
                        const float4 tt = *reinterpret_cast<const float4 *>(rr);
                        v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w;
                    } else {
                        for (int j = 0; j < nv; ++j) v[j] += rr[j];
                    }
                }
                if (in) {
                    TC *cc = C + off;
                    if (fast_c && nv == 4) {
                        if constexpr (sizeof(TC) == 4) *reinterpret_cast<float4 *>(cc) = make_float4(v[0], v[1], v[2], v[3]);
                        else *reinterpret_cast<ushort4 *>(cc) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                    } else {
                        for (int j = 0; j < nv; ++j) la::Elem<TC>::store(cc + j, v[j]);
                    }
                }
                if constexpr (LNM == 1) {
                    if (in) {
                        for (int j = 0; j < nv; ++j) la::Elem<T16>::store(C2 + off + j, v[j]);
                    }
                    if (p.ln_part) {       // N % 64 == 0 on this path (host check): a segment's 16 lanes are in or out together
                        const float2 sg = segment_stats<T16>(la::Pack4<T16>::run(v[0], v[1], v[2], v[3]));
                        if (r == 0 && in) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)((wcol0 >> 6) + (c >> 4)) * p.M + m] = sg;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // one pass's loads and rows in flight at a time (register budget)
    };
    typedef std::true_type TT;
    typedef std::false_type FF;
    if (full) {
        if (do_res) {
#pragma nounroll
            for (int h = 0; h < 4; ++h) pass(h, TT{}, TT{});
        } else {
#pragma nounroll
            for (int h = 0; h < 4; ++h) pass(h, TT{}, FF{});
        }
    } else {
#pragma nounroll
        for (int h = 0; h < 4; ++h) pass(h, FF{}, FF{});
    }
}

// The one-wave-per-SIMD kernel (LA_PP_DBG=73): 256x256 tile, 4 waves x 128x128 wave tiles, hand-placed main loop
// (mainloop_mono_asm), accumulators in the AGPRs from the first MFMA to the epilogue's ds_write.
template <bool OUT_F32, typename T16, int LNM = 0>
__global__ __launch_bounds__(MONO::THREADS, 1) void gemm_mono_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int m0 = tc.tm * 256, n0 = tc.tn * 256;
    const int z = blockIdx.y;
    const T16 *A = reinterpret_cast<const T16 *>(p.A) + (int64_t)z * p.strideA;
    const T16 *W = reinterpret_cast<const T16 *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    // per-column epilogue operands of this lane's four columns, requested before the main loop
    float b4[4], cs4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ncol = min(n0 + wc * 128 + (lane & 31) * 4 + j, p.N - 1);
        b4[j] = has_bias ? bias[ncol] : 0.f;
        cs4[j] = 0.f;
        if constexpr (LNM == 2) cs4[j] = p.ln_csum[ncol];
    }
    f32x4 acc[8][8];
    mainloop_mono_asm<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc);
    if (p.epilogue & 256) return;            // developer probe (KB_NOSTORE): the main loop (volatile asm: not removable) alone
    mono_epilogue<OUT_F32, T16, LNM>(p, z, acc, m0 + wr * 128, n0 + wc * 128, has_bias, b4, cs4, lds + wave * (32 * MONO_PITCH));
}

template <bool OUT_F32, typename T16, int LNM = 0>
int launch_mono(GemmParams p, int batch, hipStream_t stream) {
    auto kern = gemm_mono_kernel<OUT_F32, T16, LNM>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MONO::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, 256);
    p.tiles_n = la::cdiv(p.N, 256);
    p.group = getenv("LA_GEMM_GROUP") ? std::min(p.group, p.tiles_n) : std::min(p.tiles_n, std::max(4, p.group / 2));
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K * batch);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(MONO::THREADS), MONO::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// the one-wave-per-SIMD kernel by epilogue mode (the same three the ping-pong kernel has)
template <bool OUT_F32, typename T16>
int launch_mono_modes(GemmParams p, int batch, hipStream_t stream) {
    if (p.ln_stats) return launch_mono<OUT_F32, T16, 2>(p, batch, stream);
    if constexpr (OUT_F32) {
        if (p.C2) return launch_mono<OUT_F32, T16, 1>(p, batch, stream);
    }
    return launch_mono<OUT_F32, T16, 0>(p, batch, stream);
}

// LA_GEMM_Q4=1 (read per launch): the four-wave, two-workgroups-per-CU form for the shapes the hand-placed loop takes (bf16).
template <bool OUT_F32, typename T16, int LNM, bool WIDE>
int launch_q4(GemmParams p, int batch, hipStream_t stream) {
    auto kern = gemm_q4_kernel<OUT_F32, T16, LNM, 8, WIDE>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q4::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, WIDE ? Q4::TN : Q4::TM);
    p.tiles_n = la::cdiv(p.N, WIDE ? Q4::TM : Q4::TN);
    // column groups of the same WIDTH and row blocks of the same HEIGHT as the 8-wave kernel's (which counts 256 x 256 tiles)
    const int gw = WIDE ? 1 : 2, mh = WIDE ? 2 : 1;
    p.group = getenv("LA_GEMM_GROUP") ? std::min(gw * p.group, p.tiles_n) : std::min(p.tiles_n, gw * std::max(4, p.group / 2));
    p.mblock = p.tiles_n > p.group ? 32 * mh : 0;
    if (const char *g = getenv("LA_GEMM_MBLOCK")) p.mblock = atoi(g);
    if (const char *g = getenv("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    if (const char *g = getenv("LA_GEMM_Q4_PRIO")) { if (atoi(g) == 1) p.epilogue |= LA_EPI_Q4_PRIO; }
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K * batch);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(Q4::THREADS), Q4::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

template <bool OUT_F32, bool DUO, typename T16, int LNM = 0>
int launch_pp_loop(GemmParams p, int batch, hipStream_t stream) {
    if constexpr (DUO && std::is_same<T16, bf16_t>::value && ((LNM == 3 && OUT_F32) || ((LNM == 0 || LNM == 2) && !OUT_F32))) {
        const char *q4 = getenv("LA_GEMM_Q4");
        if (q4 && Q4::rem_of(p.K / 32) == 8) {                     // (K = 256, 1024, 4096, ..); 1 = 256 x 128 tiles, 2 = 128 x 256
            if (atoi(q4) == 1) return launch_q4<OUT_F32, T16, LNM, false>(p, batch, stream);
            if (atoi(q4) == 2) return launch_q4<OUT_F32, T16, LNM, true>(p, batch, stream);
        }
    }
    auto kern = gemm_pp_kernel<OUT_F32, DUO, T16, LNM>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, PP::TM);
    p.tiles_n = la::cdiv(p.N, PP::TN);
    // column tiles that walk the M dimension together (their W panels share the XCD's L2 with the streaming A panel).
    // At least 4: with fewer, the K=4096 GEMM (N = 4 tiles) re-reads its 2 MB-per-row-block A panel once per column tile
    // (in-pipeline sweep: 1 -> 46.1 ms/step, 4 -> 45.7, 8 -> 45.9, 16 -> 46.4).
    p.group = getenv("LA_GEMM_GROUP") ? std::min(p.group, p.tiles_n) : std::min(p.tiles_n, std::max(4, p.group / 2));
    // More than one column group: the groups follow each other inside blocks of 32 row tiles (tile_coord_mb), so a block's A panels
    // (16 MB at K = 1024) are re-read per group out of the Infinity Cache instead of once per sweep over all of M.  Alone on the chip
    // with cold A the MLP-up shape runs 438 -> 352 us (tools/kbench.py order); inside the pipeline, where A was just written, 0.6 %
    // of the step (profiles/r4_ab_mblock.txt).
    p.mblock = p.tiles_n > p.group ? 32 : 0;
    if (const char *g = getenv("LA_GEMM_MBLOCK")) p.mblock = atoi(g);                  // developer sweep (read per launch)
    if (const char *g = getenv("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    if (const char *g = getenv("LA_EPI_PROBE")) p.epilogue |= (atoi(g) & 7) << 16;     // developer probes of the epilogue's memory legs
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K * batch);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(PP::THREADS), PP::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// Ticket words of the persistent kernel: one block per (device, stream), zeroed when created (on that stream) and left zeroed by
// every launch's last workgroup.  Launches on one stream are ordered, so a block is never shared by two running kernels.
PersistTickets *persist_tickets(hipStream_t stream) {
    struct Slot { int dev; hipStream_t stream; PersistTickets *ptr; };
    static std::mutex mu;
    static std::vector<Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (const Slot &sl : slots)
        if (sl.dev == dev && sl.stream == stream) return sl.ptr;
    PersistTickets *ptr = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&ptr), sizeof(PersistTickets)) != hipSuccess) return nullptr;
    if (hipMemsetAsync(ptr, 0, sizeof(PersistTickets), stream) != hipSuccess) { (void)hipFree(ptr); return nullptr; }
    slots.push_back(Slot{dev, stream, ptr});
    return ptr;
}

int device_cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    dev &= 63;
    if (n[dev] == 0) {
        int v = 0;
        n[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n[dev];
}

// The persistent form takes a launch when its tiles outnumber the CUs (there is a next tile to prefetch), the hand-placed main
// loop fits, and every interior wave tile will take the wide epilogue path (row pitches / pointers aligned: checked HERE, because
// the persistent kernel prefetches into the LDS the element-wise path would stage through).  LA_GEMM_PERSIST=0 (read per launch)
// keeps one workgroup per tile: the A/B partner.
template <bool OUT_F32, typename T16, int LNM>
bool persist_eligible(const GemmParams &p, int batch, bool duo) {
    // MEASURED SLOWER than one workgroup per tile (profiles/r4_kbench_persistent_ab.txt, r4_persistent_kernel_timeline.txt; DESIGN.md
    // "GEMM, round 4"): opt-in with LA_GEMM_PERSIST=1 (read per launch), bf16 only.
    const char *e = getenv("LA_GEMM_PERSIST");
    if (!(e && e[0] == '1')) return false;
    if (!std::is_same<T16, bf16_t>::value) return false;
    if (!duo || batch != 1) return false;
    const int nt = la::cdiv(p.M, PP::TM) * la::cdiv(p.N, PP::TN);
    if (nt <= device_cu_count()) return false;
    if (p.N % PP::TN != 0) return false;                                   // (column edges would need the element-wise path too)
    if constexpr (LNM == 3) {
        if (!(p.ldc % 4 == 0 && (uintptr_t)p.C % 4 == 0 && (uintptr_t)p.C2 % 8 == 0)) return false;
        if ((p.epilogue & LA_EPI_RESIDUAL) && p.residual && !(p.ldr % 4 == 0 && (uintptr_t)p.residual % 16 == 0)) return false;
    } else {
        const int64_t es = OUT_F32 ? 4 : 2;
        if (!((p.ldc * es) % 16 == 0 && (uintptr_t)p.C % 16 == 0)) return false;
        if ((p.epilogue & LA_EPI_RESIDUAL) && p.residual && !(p.ldr % 4 == 0 && (uintptr_t)p.residual % 16 == 0)) return false;
    }
    return true;
}

template <bool OUT_F32, typename T16, int LNM>
int launch_pp_persist(GemmParams p, hipStream_t stream) {
    auto kern = gemm_pp_persist_kernel<OUT_F32, T16, LNM>;
    constexpr int LDS_BYTES = 160 * 1024;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_once.mark();
    }
    PersistTickets *tk = persist_tickets(stream);
    if (!tk) { la::set_error("gemm: ticket block allocation failed"); return LA_EHIP; }
    p.tiles_m = la::cdiv(p.M, PP::TM);
    p.tiles_n = la::cdiv(p.N, PP::TN);
    p.group = getenv("LA_GEMM_GROUP") ? std::min(p.group, p.tiles_n) : std::min(p.tiles_n, std::max(4, p.group / 2));
    p.mblock = p.tiles_n > p.group ? 32 : 0;
    if (const char *g = getenv("LA_GEMM_MBLOCK")) p.mblock = atoi(g);
    if (const char *g = getenv("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    const int grid = std::min(p.tiles_m * p.tiles_n, device_cu_count());
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PP::THREADS), LDS_BYTES, stream, p, tk);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// Main loop: the hand-placed flat stream (mainloop_duo_asm) where its k-step structure fits -- K a multiple of 128 (four k-steps
// of 32 per ring turn), at least 256 -- else the quadrant ping-pong; LA_PP_DBG=99 forces the ping-pong, 73 selects the
// one-wave-per-SIMD kernel (read per launch: tools/kbench.py flips it between rounds of one process).  Same tile, same
// accumulation order, same epilogue arithmetic: all three give identical bits.
template <bool OUT_F32, typename T16>
int launch_pp(GemmParams p, int batch, hipStream_t stream) {
    const char *dbg_env = getenv("LA_PP_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    const bool fits = p.K % 128 == 0 && p.K >= 256;
    if constexpr (std::is_same<T16, bf16_t>::value) {       // the one-wave-per-SIMD experiment exists for bf16
        if (dbg == 73 && fits) return launch_mono_modes<OUT_F32, T16>(p, batch, stream);
    }
    const bool duo = fits && dbg != 99;
    if (p.ln_csum && !p.ln_stats) {                   // LayerNorm consumer whose main loop takes the row statistics itself (opt-in, bf16)
        if constexpr (std::is_same<T16, bf16_t>::value) {
            if (!duo) {
                la::set_error("gemm_fused_ln: in-loop row statistics need the hand-placed main loop (K %% 128 == 0, K >= 256; K = %d)", p.K);
                return LA_EUNSUPPORTED;
            }
            return launch_pp_loop<OUT_F32, true, T16, 4>(p, batch, stream);
        } else {
            la::set_error("gemm_fused_ln: in-loop row statistics are built for bfloat16 (measured slower than la_row_stats16: an A/B switch)");
            return LA_EUNSUPPORTED;
        }
    }
    if (p.ln_stats) {
        if constexpr (std::is_same<T16, bf16_t>::value && !OUT_F32) {
            if (persist_eligible<OUT_F32, T16, 2>(p, batch, duo)) return launch_pp_persist<OUT_F32, T16, 2>(p, stream);
        }
        return duo ? launch_pp_loop<OUT_F32, true, T16, 2>(p, batch, stream) : launch_pp_loop<OUT_F32, false, T16, 2>(p, batch, stream);
    }
    if constexpr (OUT_F32) {
        if (p.C2) return duo ? launch_pp_loop<OUT_F32, true, T16, 1>(p, batch, stream) : launch_pp_loop<OUT_F32, false, T16, 1>(p, batch, stream);
    }
    if constexpr (std::is_same<T16, bf16_t>::value && !OUT_F32) {
        if (persist_eligible<OUT_F32, T16, 0>(p, batch, duo)) return launch_pp_persist<OUT_F32, T16, 0>(p, stream);
    }
    return duo ? launch_pp_loop<OUT_F32, true, T16>(p, batch, stream) : launch_pp_loop<OUT_F32, false, T16>(p, batch, stream);
}

template <typename T, bool OUT_F32, typename CF, bool TA = false, bool TW = false>
int launch(GemmParams p, int batch, hipStream_t stream, const char *family) {
    auto kern = gemm_kernel<T, OUT_F32, CF, TA, TW>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, CF::TM);
    if (const char *g = getenv("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    la::TimerScope ts(family, stream, 2.0 * p.M * p.N * (p.K_tail ? (double)((batch - 1) * (int64_t)p.K + p.K_tail) : (double)p.K * batch));
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(CF::THREADS), CF::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// Measured on the config-2 shapes (tools/kbench.py, round 1): the 128x128 / 2-workgroups-per-CU configuration beats the
// 256x128 / 3-stage / 1-workgroup-per-CU one by 5-15 % on every shape, so it is the default; LA_GEMM_TILE=256 selects
// the big tile for A/B runs.
bool use_big_tile(int M, int N, int batch) {
    static const char *force = getenv("LA_GEMM_TILE");
    return force && atoi(force) == 256 && (int64_t)la::cdiv(M, 256) * la::cdiv(N, BN) * batch >= 512;
}

}  // namespace

// ldw: row pitch of W in elements (0 = dense [N][K])
// Split-K tail: C[m][n] = epilogue(sum_s P[s][m][n]) in a fixed order (deterministic), same epilogue order as the GEMM kernels
// (bias, activation, residual).
__global__ void splitk_reduce_kernel(const float *P, int S, int M, int N, float *C, int64_t ldc, const float *bias,
                                     const float *residual, int64_t ldr, int epilogue) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += P[(int64_t)s * M * N + i];
    if ((epilogue & LA_EPI_BIAS) && bias) v += bias[n];
    if (epilogue & LA_EPI_GELU) v = la::gelu_erf(v);
    else if (epilogue & LA_EPI_MISH) v = la::mish(v);
    if ((epilogue & LA_EPI_RESIDUAL) && residual) v += residual[(int64_t)m * ldr + n];
    C[(int64_t)m * ldc + n] = v;
}

// float32 kernel by operand layout: NT (both K-contiguous), TT (both [K][rows]: weight gradients), NT/TW (input gradients)
static int launch_f32(const GemmParams &p, int batch, bool tA, bool tW, hipStream_t stream) {
    typedef Cfg<2, 2> Small;
    if (tA && tW) return launch<float, true, Small, true, true>(p, batch, stream, "gemm_f32");
    if (tW) return launch<float, true, Small, false, true>(p, batch, stream, "gemm_f32");
    if (tA) return launch<float, true, Small, true, false>(p, batch, stream, "gemm_f32");
    return launch<float, true, Small>(p, batch, stream, "gemm_f32");
}

struct LnFuse {
    void *C2; int64_t ldc2, strideC2;
    const float *stats, *csum;
    float *part;
};

// Row statistics from the producers' per-segment partials (Chan et al. combination of equal-sized groups):
// mean = avg(mean_k), M2 = sum(M2_k) + 64 sum((mean_k - mean)^2), rstd = 1 / sqrt(M2 / (64 slots) + eps).
__global__ void ln_stats_finalize_kernel(const float2 *part, int slots, int M, float eps, float2 *stats) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    float mean = 0.f;
    for (int k = 0; k < slots; ++k) mean += part[(int64_t)k * M + row].x;
    mean /= (float)slots;
    float m2 = 0.f;
    for (int k = 0; k < slots; ++k) {
        const float2 pk = part[(int64_t)k * M + row];
        const float d = pk.x - mean;
        m2 += pk.y + 64.0f * d * d;
    }
    stats[row] = make_float2(mean, 1.0f / sqrtf(m2 / (64.0f * (float)slots) + eps));
}

static int gemm_run_ldw(int dtype, int M, int N, int K, int batch, const void *A, int64_t lda, int64_t strideA, const void *W,
                        int64_t ldw_arg, int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias, int64_t strideBias,
                 const float *residual, int64_t ldr, int64_t strideR, int epilogue, hipStream_t stream, const LnFuse *ln = nullptr) {
    if (M == 0 || N == 0 || batch == 0) return LA_OK;
    LA_CHECK_ARG(A && W && C, "gemm: null pointer");
    LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad sizes");
    LA_CHECK_ARG(dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16, "gemm: bad dtype");
    const int ke = dtype == LA_F32 ? 32 : 64;
    const int es = dtype == LA_F32 ? 4 : 2;
    const bool tA = epilogue & LA_GEMM_TRANS_A, tW = epilogue & LA_GEMM_TRANS_W;
    epilogue &= ~(LA_GEMM_TRANS_A | LA_GEMM_TRANS_W);
    if (tA || tW) {
        LA_CHECK_ARG(dtype == LA_F32, "gemm: transposed operands are float32 only");
        LA_CHECK_ARG((!tA || (M % 4 == 0 && lda >= M)) && (!tW || (N % 4 == 0 && (ldw_arg == 0 || ldw_arg >= N))),
                     "gemm: a transposed operand [K][rows] needs rows %% 4 == 0 and a pitch >= rows");
        LA_CHECK_ARG(!ln, "gemm: no LayerNorm fold with transposed operands");
    }
    LA_CHECK_ARG(K % ke == 0 || ((tA || tW) && (tA || lda >= la::round_up(K, ke)) && (tW || (ldw_arg >= la::round_up(K, ke)))),
                 "gemm: K=%d must be a multiple of %d (or: a transposed operand, and K-contiguous rows pitched to the rounded-up K)", K, ke);
    LA_CHECK_ARG((lda * es) % 16 == 0 && (strideA * es) % 16 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0),
                 "gemm: A/W rows must be 16-byte aligned");
    LA_CHECK_ARG(!(epilogue & LA_EPI_RESIDUAL) || residual, "gemm: residual epilogue without pointer");
    LA_CHECK_ARG(!(epilogue & LA_EPI_BIAS) || bias, "gemm: bias epilogue without pointer");
    LA_CHECK_ARG((strideW * es) % 16 == 0 && (ldw_arg * es) % 16 == 0, "gemm: W batch stride / row pitch must be 16-byte aligned");
    GemmParams p{M, N, K, A, lda, strideA, W, ldw_arg > 0 ? ldw_arg : (int64_t)(tW ? N : K), strideW, C, ldc, strideC, bias, strideBias, residual, ldr, strideR, epilogue,
                 0, la::cdiv(N, BN), pick_group(K, es, la::cdiv(N, BN))};
    const bool out_f32 = epilogue & LA_EPI_OUT_F32;
    typedef Cfg<2, 2> Small;
    typedef Cfg<4, 3> Big;
    if (dtype == LA_BF16 || dtype == LA_F16) {
        // 256x256 ping-pong kernel once it can fill the chip (>= 192 tiles); LA_GEMM_TILE=512 forces it, 128/256 forbid it
        // (a 256-column tile on N <= 128 -- the gathered-label logits, N = Lmax + 1 -- would compute mostly padding)
        static const char *force = getenv("LA_GEMM_TILE");
        const int forced = force ? atoi(force) : 0;
        const bool pp = forced == 512 || (forced == 0 && N > 128 && (int64_t)la::cdiv(M, 256) * la::cdiv(N, 256) * batch >= 192);
        const bool half = dtype == LA_F16;
        if (ln) {
            if (!pp || (epilogue & LA_EPI_MISH)) {
                la::set_error("gemm_fused_ln: shape M=%d N=%d batch=%d does not run on the 256x256 kernel the fusion is built into", M, N, batch);
                return LA_EUNSUPPORTED;
            }
            LA_CHECK_ARG(!ln->C2 || (out_f32 && ln->ldc2 == ldc && ln->strideC2 == strideC && ldc % 4 == 0 && (uintptr_t)ln->C2 % 8 == 0),
                         "gemm_fused_ln: the 16-bit copy accompanies an f32 result and shares its row pitch / batch stride");
            LA_CHECK_ARG(!(ln->C2 && ln->stats), "gemm_fused_ln: a GEMM is the producer or the consumer of a folded LayerNorm, not both");
            LA_CHECK_ARG(!ln->stats || ln->csum, "gemm_fused_ln: row statistics without the column sums of the folded weights");
            LA_CHECK_ARG(!ln->csum || batch == 1, "gemm_fused_ln: the LayerNorm epilogue takes batch 1");
            LA_CHECK_ARG(!(ln->C2 && ln->csum), "gemm_fused_ln: a GEMM is the producer or the consumer of a folded LayerNorm, not both");
            LA_CHECK_ARG(!ln->part || (ln->C2 && N % 64 == 0 && batch == 1 && (uintptr_t)ln->part % 8 == 0),
                         "gemm_fused_ln: partial statistics go with the 16-bit copy, N % 64 == 0, batch 1");
            p.C2 = ln->C2; p.ldc2 = ln->ldc2; p.strideC2 = ln->strideC2; p.ln_stats = ln->stats; p.ln_csum = ln->csum; p.ln_part = ln->part;
        }
        if (pp && !(epilogue & LA_EPI_MISH)) {
            if (half) return out_f32 ? launch_pp<true, la::f16_t>(p, batch, stream) : launch_pp<false, la::f16_t>(p, batch, stream);
            return out_f32 ? launch_pp<true, bf16_t>(p, batch, stream) : launch_pp<false, bf16_t>(p, batch, stream);
        }
        if (!half && use_big_tile(M, N, batch))
            return out_f32 ? launch<bf16_t, true, Big>(p, batch, stream, "gemm_bf16") : launch<bf16_t, false, Big>(p, batch, stream, "gemm_bf16");
        if (half) return out_f32 ? launch<la::f16_t, true, Small>(p, batch, stream, "gemm_bf16") : launch<la::f16_t, false, Small>(p, batch, stream, "gemm_bf16");
        return out_f32 ? launch<bf16_t, true, Small>(p, batch, stream, "gemm_bf16") : launch<bf16_t, false, Small>(p, batch, stream, "gemm_bf16");
    }
    // float32 (training / parity mode) with few tiles and a long K -- the text decoder's 80-row GEMMs are 8 tiles on 256 CUs,
    // each a serial K = 1024..4096 walk: K is cut into S equal chunks that run as S batch slots of the same kernel into a
    // partial-sum buffer, and a second kernel adds them in a fixed order and applies the epilogue.
    if (batch == 1) {
        const int tiles = la::cdiv(M, Small::TM) * la::cdiv(N, BN);
        // S: chunks of whole 32-element k-steps, the last one may be shorter.  Few tiles -> fill the chip (<= 256 workgroups);
        // a partial last round of tiles (192 or 384 tiles on 256 CUs) -> S = 4 / 2 makes the rounds finer (768 quarter / half
        // tiles = 3 full rounds) when K is long enough to pay for the partial-sum pass.
        const int ksteps = la::cdiv(K, 32);
        // (two 128x128 workgroups fit a CU, so 512 slots would fill it twice over: LA_GEMM_SPLIT_SLOTS=512 measured no
        //  difference on the fused fine-tune step, 868 vs 868 ms; one workgroup per CU stays the limit)
        static const int kSlots = getenv("LA_GEMM_SPLIT_SLOTS") ? atoi(getenv("LA_GEMM_SPLIT_SLOTS")) : 256;
        int S = 1;
        while (S < 16 && tiles * S * 2 <= kSlots && ksteps / (S * 2) >= 4) S *= 2;
        if (S == 1 && K >= 1024 && (int64_t)M * N <= ((int64_t)4 << 20)) {
            if (tiles * 4 % 256 == 0 && tiles % 256 != 0 && tiles < 256) S = 4;
            else if (tiles * 2 % 256 == 0 && tiles % 256 != 0 && tiles < 512) S = 2;
        }
        static const bool no_split = getenv("LA_GEMM_NO_SPLITK") != nullptr;
        if (S > 1 && !no_split && !(epilogue & 256)) {
            float *part = static_cast<float *>(la::stream_scratch(stream, la::SCRATCH_SPLITK, (size_t)S * M * N * sizeof(float)));
            if (!part) { la::set_error("gemm: split-K scratch allocation failed"); return LA_EHIP; }
            const int Kc = la::cdiv(ksteps, S) * 32;                   // S - 1 chunks of Kc, the last one takes the rest
            if (K - (S - 1) * Kc <= 0) S = la::cdiv(K, Kc);
            GemmParams ps{M, N, Kc, A, lda, tA ? (int64_t)Kc * lda : (int64_t)Kc, W, p.ldw, tW ? (int64_t)Kc * p.ldw : (int64_t)Kc, part, (int64_t)N, (int64_t)M * N, nullptr, 0, nullptr, 0, 0,
                          LA_EPI_OUT_F32, 0, la::cdiv(N, BN), pick_group(Kc, es, la::cdiv(N, BN))};
            ps.K_tail = K - (S - 1) * Kc;
            const int rc = launch_f32(ps, S, tA, tW, stream);
            if (rc == LA_OK) {
                const int64_t total = (int64_t)M * N;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)la::cdiv(total, (int64_t)256)), dim3(256), 0, stream, part, S, M, N,
                                   reinterpret_cast<float *>(C), ldc, bias, residual, ldr, epilogue);
            }
            if (rc != LA_OK) return rc;
            LA_LAUNCH_CHECK();
            return LA_OK;
        }
    }
    return launch_f32(p, batch, tA, tW, stream);
}

// LayerNorm folded into the GEMMs on either side of it (encoder blocks, 16-bit modes, shapes that run on the 256x256 kernel):
//   producer  (C2 != null): besides the f32 result rows (residual stream) also stores them rounded to `dtype` into C2;
//   consumer  (ln_stats != null): A = those raw rows, W = gamma-folded weights, ln_stats [M][2] = (mean, rstd) per row,
//             ln_csum [N] = row sums of W; bias must already hold b + W beta.  LA_EUNSUPPORTED for other shapes / f32.
extern "C" int la_gemm_fused_ln(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                                int64_t strideA, const void *W, void *C, int64_t ldc, int64_t strideC, const float *bias,
                                const float *residual, int64_t ldr, int64_t strideR, int32_t epilogue, void *C2, int64_t ldc2,
                                int64_t strideC2, const float *ln_stats, const float *ln_csum, float *ln_part, void *stream_) {
    if (dtype != LA_BF16 && dtype != LA_F16) {
        la::set_error("gemm_fused_ln: 16-bit compute dtypes only");
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(C2 || ln_csum, "gemm_fused_ln: neither a second output nor the consumer's column sums given");
    const LnFuse ln{C2, ldc2, strideC2, ln_stats, ln_csum, ln_part};
    return gemm_run_ldw(dtype, M, N, K, batch, A, lda, strideA, W, 0, 0, C, ldc, strideC, bias, 0, residual, ldr, strideR, epilogue,
                        (hipStream_t)stream_, &ln);
}

// The residual GEMMs of the 16-bit encoder on the SPLIT stream (SplitRes in la_common.h): (hi, lo) <- epi(A W^T) + residual, where
// the residual is an f32 array (the stem: conv2 + positional embedding writes the stream) or the stream itself (in place:
// x += out-proj / x += mlp).  hi [M][ld] `dtype` is at the same time the raw A operand of the next LayerNorm-folded GEMM.
extern "C" int la_gemm_split(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda, int64_t strideA,
                             const void *W, void *hi, void *lo, int64_t ld, int64_t stride, const float *bias, const float *residual,
                             int64_t ldr, int64_t strideR, int32_t epilogue, float *ln_part, void *stream_) {
    if (M == 0 || N == 0 || batch == 0) return LA_OK;
    if (dtype != LA_BF16 && dtype != LA_F16) {
        la::set_error("gemm_split: 16-bit compute dtypes only");
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(A && W && hi && lo, "gemm_split: null pointer");
    LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_split: bad sizes");
    if (N <= 128 || (int64_t)la::cdiv(M, 256) * la::cdiv(N, 256) * batch < 192 || K % 64 != 0) {
        la::set_error("gemm_split: shape M=%d N=%d K=%d batch=%d does not run on the 256x256 kernel the split epilogue is built into", M, N, K, batch);
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG((epilogue & ~(LA_EPI_BIAS | LA_EPI_GELU | LA_EPI_RESIDUAL)) == 0, "gemm_split: epilogue takes BIAS, GELU, RESIDUAL only");
    LA_CHECK_ARG((lda * 2) % 16 == 0 && (strideA * 2) % 16 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0),
                 "gemm_split: A/W rows must be 16-byte aligned");
    LA_CHECK_ARG(!(epilogue & LA_EPI_BIAS) || bias, "gemm_split: bias epilogue without pointer");
    LA_CHECK_ARG(ld >= N && ((uintptr_t)hi % 2 == 0), "gemm_split: row pitch below N");
    LA_CHECK_ARG(!ln_part || (N % 64 == 0 && batch == 1 && (uintptr_t)ln_part % 8 == 0), "gemm_split: partial statistics need N %% 64 == 0, batch 1");
    int epi = epilogue & (LA_EPI_BIAS | LA_EPI_GELU);
    if (epilogue & LA_EPI_RESIDUAL) epi |= residual ? LA_EPI_RESIDUAL : LA_EPI_SPLIT_INPLACE;
    if (const char *e = getenv("LA_EPI_SPLIT_PASS")) { if (atoi(e) == 32) epi |= LA_EPI_SPLIT_PASS32; }
    GemmParams p{M, N, K, A, lda, strideA, W, (int64_t)K, 0, lo, ld, stride, bias, 0, residual, ldr, strideR, epi,
                 0, la::cdiv(N, BN), pick_group(K, 2, la::cdiv(N, BN))};
    p.C2 = hi; p.ldc2 = ld; p.strideC2 = stride; p.ln_part = ln_part;
    hipStream_t stream = (hipStream_t)stream_;
    const char *dbg_env = getenv("LA_PP_DBG");
    const bool duo = K % 128 == 0 && K >= 256 && !(dbg_env && atoi(dbg_env) == 99);
    if (dtype == LA_F16) {
        return duo ? launch_pp_loop<true, true, la::f16_t, 3>(p, batch, stream) : launch_pp_loop<true, false, la::f16_t, 3>(p, batch, stream);
    }
    if (persist_eligible<true, bf16_t, 3>(p, batch, duo)) return launch_pp_persist<true, bf16_t, 3>(p, stream);
    return duo ? launch_pp_loop<true, true, bf16_t, 3>(p, batch, stream) : launch_pp_loop<true, false, bf16_t, 3>(p, batch, stream);
}

extern "C" int la_ln_stats_finalize(const float *part, int32_t slots, int32_t M, float eps, float *stats, void *stream_) {
    if (M == 0) return LA_OK;
    LA_CHECK_ARG(part && stats && slots > 0 && M > 0, "ln_stats_finalize: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    la::TimerScope ts("layernorm", stream);
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3(la::cdiv(M, 256)), dim3(256), 0, stream, reinterpret_cast<const float2 *>(part), slots, M,
                       eps, reinterpret_cast<float2 *>(stats));
    LA_LAUNCH_CHECK();
    return LA_OK;
}

int la::gemm_run(int dtype, int M, int N, int K, int batch, const void *A, int64_t lda, int64_t strideA, const void *W,
                 int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias, int64_t strideBias,
                 const float *residual, int64_t ldr, int64_t strideR, int epilogue, hipStream_t stream) {
    return gemm_run_ldw(dtype, M, N, K, batch, A, lda, strideA, W, 0, strideW, C, ldc, strideC, bias, strideBias, residual, ldr,
                        strideR, epilogue, stream);
}

extern "C" int la_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                       int64_t strideA, const void *W, void *C, int64_t ldc, int64_t strideC, const float *bias,
                       const float *residual, int64_t ldr, int64_t strideR, int32_t epilogue, void *stream_) {
    return la::gemm_run(dtype, M, N, K, batch, A, lda, strideA, W, 0, C, ldc, strideC, bias, 0, residual, ldr, strideR,
                        epilogue, (hipStream_t)stream_);
}

// General face used by the backward pass: W rows with their own pitch (ldw >= K) and per-batch strides for A, W, C, bias.
extern "C" int la_gemm_ex(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                          int64_t strideA, const void *W, int64_t ldw, int64_t strideW, void *C, int64_t ldc, int64_t strideC,
                          const float *bias, int32_t epilogue, void *stream_) {
    LA_CHECK_ARG(ldw >= ((epilogue & LA_GEMM_TRANS_W) ? N : K), "gemm_ex: ldw smaller than a row of W");
    return gemm_run_ldw(dtype, M, N, K, batch, A, lda, strideA, W, ldw, strideW, C, ldc, strideC, bias, 0, nullptr, 0, 0, epilogue,
                        (hipStream_t)stream_);
}
