// attn_mfma_shape.hip -- what would the MFMA SHAPE alone be worth to the attention kernel (developer tool, DESIGN.md "Attention")?
// One iteration = the instruction mix of one 64-key tile of one wave with 32 queries (head_dim 64), operands in registers:
//   shape 32: 16 x v_mfma_f32_32x32x16_bf16  (8 for S^T = K Q^T, 8 for O^T += V^T P^T)      -- what la_attention.hip issues
//   shape 16: 32 x v_mfma_f32_16x16x32_bf16  (the same flops)
// each with the tile's vector work (32 v_exp_f32 + 32 v_add_f32 + 16 v_cvt_pk_bf16_f32 on distinct registers) in the two
// arrangements hipcc produces (blocks) and a hand interleave, at 4 waves per SIMD (the kernel's occupancy), random operand bits.
// Prints ns per tile per SIMD share, the in-kernel shader clock (s_memtime ticks per 100 MHz s_memrealtime tick), and the
// MFMA-only / vector-only times.  No re-layout of P, no cross-lane row sums: an UPPER bound for the 16x16x32 form.
//   hipcc --offload-arch=gfx950 -O3 tools/attn_mfma_shape.hip -o /tmp/attn_mfma_shape && /tmp/attn_mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma32(f32x16 &acc, const s16x8 &a, const s16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16(f32x4 &acc, const s16x8 &a, const s16x8 &b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void vexp(float &d, float a) { asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a)); }
__device__ __forceinline__ void vadd(float &d, float a, float b) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }
__device__ __forceinline__ void vcvt(float &d, float a, float b) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }

// SHAPE 32 | 16; WHAT 0 both (blocks), 1 MFMAs only, 2 vector only, 3 both (interleaved: one MFMA group, then its share of the vector work)
template <int SHAPE, int WHAT>
__global__ __launch_bounds__(256) void tile_kernel(int iters, float *sink, long long *clk) {
    // operand fragments: 4 "K / V" and 4 "Q / P" register sets with lane-dependent pseudo-random bf16 bits
    s16x8 a[4], b[4];
    unsigned st = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            st = st * 1664525u + 1013904223u; a[i][j] = (short)(0x3c00 + ((st >> 9) & 0x3ff) - ((st >> 3) & 0x8000));
            st = st * 1664525u + 1013904223u; b[i][j] = (short)(0x3c00 + ((st >> 9) & 0x3ff) - ((st >> 3) & 0x8000));
        }
    f32x16 acc32[4];
    f32x4 acc16[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc16[i][j] = 0.f;
    float s[32], p[32], l[4] = {0.f, 0.f, 0.f, 0.f}, pk[16];
    for (int i = 0; i < 32; ++i) { s[i] = -0.01f * (float)((threadIdx.x * 7 + i * 13) & 63); p[i] = 0.f; }
    for (int i = 0; i < 16; ++i) pk[i] = 0.f;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHAT == 0 || WHAT == 1) {
            if constexpr (SHAPE == 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) mfma32(acc32[i & 3], a[i & 3], b[(i >> 2) & 3]);
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) mfma16(acc16[i & 15], a[i & 3], b[(i >> 2) & 3]);
            }
        }
        if constexpr (WHAT == 0 || WHAT == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) vexp(p[i], s[i]);
#pragma unroll
            for (int i = 0; i < 32; ++i) vadd(l[i & 3], l[i & 3], p[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) vcvt(pk[i], p[2 * i], p[2 * i + 1]);
        }
        if constexpr (WHAT == 3) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if constexpr (SHAPE == 32) mfma32(acc32[g & 3], a[g & 3], b[(g >> 2) & 3]);
                else { mfma16(acc16[(2 * g) & 15], a[g & 3], b[(g >> 2) & 3]); mfma16(acc16[(2 * g + 1) & 15], a[(g + 1) & 3], b[(g >> 2) & 3]); }
                vexp(p[2 * g], s[2 * g]); vexp(p[2 * g + 1], s[2 * g + 1]);
                vadd(l[g & 3], l[g & 3], p[2 * g]); vadd(l[(g + 2) & 3], l[(g + 2) & 3], p[2 * g + 1]);
                vcvt(pk[g], p[2 * g], p[2 * g + 1]);
            }
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    float r = l[0] + l[1] + l[2] + l[3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc32[i][j];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) r += acc16[i][j];
    for (int i = 0; i < 16; ++i) r += pk[i];
    if (r == 12345.678f) sink[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

template <int SHAPE, int WHAT>
void run(const char *name, float *sink, long long *clk, int cus) {
    const int iters = 4000, w = 4;
    double best = 1e30, ghz = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((tile_kernel<SHAPE, WHAT>), dim3(cus * w), dim3(256), 0, 0, iters, sink, clk);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((tile_kernel<SHAPE, WHAT>), dim3(cus * w), dim3(256), 0, 0, iters, sink, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long h[2];
        CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
        const double ns = ms * 1e6 / iters / w;
        if (ns < best) { best = ns; ghz = (double)h[0] / ((double)h[1] * 10.0); }
    }
    printf("  %-44s %7.1f ns per tile per SIMD share   (counter ratio %.2f GHz)\n", name, best, ghz);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    float *sink; long long *clk;
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&clk, 64));
    const int cus = prop.multiProcessorCount;
    printf("one 64-key attention tile of a 32-query wave (head_dim 64), operands in registers, 4 waves per SIMD, %d CUs\n", cus);
    for (int rd = 0; rd < 2; ++rd) {
        run<32, 1>("32x32x16: 16 MFMAs only", sink, clk, cus);
        run<16, 1>("16x16x32: 32 MFMAs only", sink, clk, cus);
        run<32, 2>("vector work only (32 exp, 32 add, 16 cvt_pk)", sink, clk, cus);
        run<32, 0>("32x32x16 + vector work, blocks", sink, clk, cus);
        run<16, 0>("16x16x32 + vector work, blocks", sink, clk, cus);
        run<32, 3>("32x32x16 + vector work, interleaved", sink, clk, cus);
        run<16, 3>("16x16x32 + vector work, interleaved", sink, clk, cus);
    }
    return 0;
}
