// valu_mfma_overlap.hip -- how do the matrix pipe and the vector pipe of one SIMD share time (developer tool, DESIGN.md "Attention")?
// One iteration = the instruction mix of one attention key tile of one wave: 16 x v_mfma_f32_32x32x16_bf16 and NV vector
// instructions (v_exp_f32 or v_fma_f32), in four arrangements, at 1 / 2 / 4 waves per SIMD:
//   mfma      : the 16 MFMAs alone (two independent accumulator chains)
//   valu      : the vector instructions alone
//   blocks    : 16 MFMAs, then the vector block (what hipcc emits for the flash loop)
//   interleave: one MFMA, then NV/16 vector instructions, 16 times (independent registers)
// Prints cycles per iteration per SIMD (wall clock x the measured shader clock), so "max" vs "sum" behaviour is visible.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_overlap.hip -o /tmp/valu_mfma_overlap && /tmp/valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma(f32x16 &acc, const s16x8 &a, const s16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int OP> __device__ __forceinline__ void vop(float &x) {
    if constexpr (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    else if constexpr (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == 3) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(x));
    else if constexpr (OP == 4) asm volatile("v_log_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x));
    else asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP> __device__ __forceinline__ void vop2(f32x2 &x) {
    if constexpr (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(x));
    else if constexpr (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(x));
    else asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(x));
}

// instruction issue rates alone: NV independent registers, one instruction each per iteration
template <int OP, bool PK, int NV>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float *sink) {
    float e[NV];
    f32x2 e2[NV];
    for (int i = 0; i < NV; ++i) { e[i] = 1.0f + 0.001f * (float)(threadIdx.x + i); e2[i] = f32x2{e[i], e[i]}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (PK) vop2<OP>(e2[i]); else vop<OP>(e[i]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NV; ++i) s += e[i] + e2[i].x + e2[i].y;
    if (s == 12345.678f) sink[0] = s;
}

template <int OP, bool PK>
void rate(const char *name, float *sink, int cus) {
    const int iters = 4000;
    constexpr int NV = 32;
    printf("  %-18s", name);
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL((rate_kernel<OP, PK, NV>), dim3(cus * w), dim3(256), 0, 0, iters, sink);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<OP, PK, NV>), dim3(cus * w), dim3(256), 0, 0, iters, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %6.2f ns", ms * 1e6 / ((double)iters * NV * w));
    }
    printf("\n");
    fflush(stdout);
}

// ARR 0 mfma, 1 valu, 2 blocks, 3 interleave; OP 0 exp, 1 fma; NV vector instructions per iteration (multiple of 16)
template <int ARR, int OP, int NV>
__global__ __launch_bounds__(256) void mix_kernel(int iters, float *sink, long long *clk) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x); b[i] = (short)(0x3c00 + i); }
    float e[NV];
    for (int i = 0; i < NV; ++i) e[i] = -0.001f * (float)(threadIdx.x + i);
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (ARR == 0 || ARR == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { mfma(acc0, a, b); mfma(acc1, a, b); }
        }
        if constexpr (ARR == 1 || ARR == 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) vop<OP>(e[i]);
        }
        if constexpr (ARR == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                mfma((i & 1) ? acc1 : acc0, a, b);
#pragma unroll
                for (int j = 0; j < NV / 16; ++j) vop<OP>(e[i * (NV / 16) + j]);
            }
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < NV; ++i) s += e[i];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

static double g_ghz = 0;

template <int ARR, int OP, int NV>
double run(int waves_per_simd, float *sink, long long *clk, int cus) {
    const int iters = 2000;
    const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD; waves_per_simd workgroups per CU
    hipLaunchKernelGGL((mix_kernel<ARR, OP, NV>), dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<ARR, OP, NV>), dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[2];
    CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    // the kernel's wall time (events) in shader cycles -- clock rate from the kernel's own two counters (s_memtime ticks per
    // 100 MHz s_memrealtime tick) -- per iteration, divided by the waves sharing a SIMD
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    g_ghz = ghz;
    return ms * 1e6 * ghz / iters / waves_per_simd;
}

template <int OP, int NV>
void table(const char *name, float *sink, long long *clk, int cus) {
    printf("%s, %d vector instructions + 16 MFMA 32x32x16 per iteration: shader cycles per iteration per SIMD-share\n", name, NV);
    printf("  waves/SIMD      mfma      valu    blocks  interleave\n");
    for (int w : {1, 2, 4}) {
        const double m = run<0, OP, NV>(w, sink, clk, cus), v = run<1, OP, NV>(w, sink, clk, cus);
        const double bl = run<2, OP, NV>(w, sink, clk, cus), il = run<3, OP, NV>(w, sink, clk, cus);
        printf("  %10d %9.0f %9.0f %9.0f %11.0f   (counter ratio %.2f GHz)\n", w, m, v, bl, il, g_ghz);
    }
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    float *sink; long long *clk;
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&clk, 64));
    printf("issue rate alone, ns per wave instruction per SIMD at 1 / 2 / 4 waves per SIMD (x shader GHz = cycles)\n");
    rate<1, false>("v_fma_f32", sink, prop.multiProcessorCount);
    rate<0, true>("v_pk_fma_f32", sink, prop.multiProcessorCount);
    rate<1, true>("v_pk_mul_f32", sink, prop.multiProcessorCount);
    rate<2, true>("v_pk_add_f32", sink, prop.multiProcessorCount);
    rate<3, false>("v_med3_f32", sink, prop.multiProcessorCount);
    rate<6, false>("v_max3_f32", sink, prop.multiProcessorCount);
    rate<5, false>("v_cvt_pk_bf16_f32", sink, prop.multiProcessorCount);
    rate<0, false>("v_exp_f32", sink, prop.multiProcessorCount);
    rate<2, false>("v_rcp_f32", sink, prop.multiProcessorCount);
    rate<4, false>("v_log_f32", sink, prop.multiProcessorCount);
    table<0, 32>("v_exp_f32", sink, clk, prop.multiProcessorCount);
    table<1, 32>("v_fma_f32", sink, clk, prop.multiProcessorCount);
    table<1, 96>("v_fma_f32", sink, clk, prop.multiProcessorCount);
    return 0;
}
