// la_gemm_epilogue.h -- epilogues of the 256 x 256 GEMM kernel (la_gemm_pp_kernel.h): one wave's 128 x 64 output tile from the MFMA
// accumulator layout through the wave's LDS region to row-major quads -- LayerNorm fold / bias, GELU, residual (f32 rows or the
// split stream), wide stores, the 16-bit copy, per-segment row statistics.
#pragma once
#include <type_traits>

#include "la_gemm_params.h"
#include "la_gemm_pp.h"

namespace la {
namespace gemm {

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
    } else if constexpr (LNM == 6) {
        // f16x2 products: the operands were scaled per row by powers of two before their split; st.y = the A row's inverse scale,
        // cs4 = the W rows' (exact multiplications), then the bias: one rounding, as in the float32 kernel
        const float rs = st.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j] * rs, cs4[j], b4[j]);
    } else if (has_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b4[j];
    }
    if (do_gelu) {
        if constexpr (!OUT_F32) {                        // result is rounded to 16 bits: the 11-slot sigmoid form
            if (epilogue & 4096) {                       // developer A/B (LA_GELU_PK=1): the erfc form on the packed pipe
                const la::f32x2 lo = la::gelu_pk(la::f32x2{v[0], v[1]}), hi = la::gelu_pk(la::f32x2{v[2], v[3]});
                v = f32x4{lo.x, lo.y, hi.x, hi.y};
            } else if (LA_DEV_BIT(epilogue, 8192)) {     // experiment build (LA_GELU_PK=2): one value at a time
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
    constexpr bool LNC = LNM == 2 || LNM == 4 || LNM == 5 || LNM == 6;   // LayerNorm consumer (6: the row / column scale form of the f16x2 products)
    constexpr int EQ = LNM == 6 ? 6 : (LNC ? 2 : LNM);                    // epi_quad's arithmetic
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
                    if (LA_DEV_BIT(epi, 1 << 16)) {        // experiment build (LA_EPI_PROBE & 1): no residual loads
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
                if constexpr (LNM == 4 || LNM == 6) {
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
                    epi_quad<OUT_F32, EQ>(v, b4, cs4, LNC ? st[it] : make_float2(0.f, 0.f), has_bias, do_gelu, epi);
                    if constexpr (RES) {
                        if (LNM == 6 && (epi & LA_EPI_RES_GELU_GRAD)) {     // f16x2 products only: dX of the MLP's second Linear times gelu'(u)
                            v[0] *= la::gelu_erf_grad(t[it].x); v[1] *= la::gelu_erf_grad(t[it].y);
                            v[2] *= la::gelu_erf_grad(t[it].z); v[3] *= la::gelu_erf_grad(t[it].w);
                        } else { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                    }
                    const int64_t off = (int64_t)(h * RP + rl) * p.ldc;
                    TC *c = cw + off;
                    if constexpr (sizeof(TC) == 4) {
                        if (!LA_DEV_BIT(epi, 2 << 16) || v[0] == 12345.678f)      // (experiment build, LA_EPI_PROBE & 2: no f32 store)
                            *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                        if constexpr (LNM == 1) {
                            const ushort4 pk = la::Pack4<T16>::run(v[0], v[1], v[2], v[3]);
                            if (!LA_DEV_BIT(epi, 4 << 16) || v[1] == 12345.678f)  // (experiment build, LA_EPI_PROBE & 4: no 16-bit copy)
                                *reinterpret_cast<ushort4 *>(c2w + off) = pk;
                            if (part) {
                                const float2 sg = segment_stats<T16>(pk);
                                if (r == 0) part[h * RP + rl] = sg;
                            }
                        }
                    } else {
                        if (!LA_DEV_BIT(epi, 2 << 16) || v[0] == 12345.678f)      // (experiment build, LA_EPI_PROBE & 2: no 16-bit store)
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
            if constexpr (LNM == 4 || LNM == 6) st = stats_tab[wrow0 - tile_m0 + h * 32 + rl];
            if constexpr (LNM == 5) {
                const int src = (((h * 32 + it * 4) & 63) + q) * 4;
                const float2 sr = h * 32 >= 64 ? sr1 : sr0;
                st = make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sr.x))),
                                 __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sr.y))));
            }
            epi_quad<OUT_F32, EQ>(v, b4, cs4, st, has_bias, do_gelu, epi);
            if (m >= p.M || n >= p.N) continue;
            const int nv = min(4, p.N - n);
            if (do_res) {
                const float *rr = R + (int64_t)m * p.ldr + n;
                if (LNM == 6 && (epi & LA_EPI_RES_GELU_GRAD)) {
                    for (int j = 0; j < nv; ++j) v[j] *= la::gelu_erf_grad(rr[j]);
                } else if (fast_r && nv == 4) {
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

}  // namespace gemm
}  // namespace la
