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

// Realistic operand mix: does the overlap survive vector instructions with three DISTINCT source registers (register-file read
// ports), and does it matter whether the MFMA accumulators live in VGPRs or AGPRs?  16 MFMAs then 48 v_pk_fma_f32 d, a, b, c (+ 16
// v_exp_f32) on distinct registers, "blocks" arrangement, ACC = 0: accumulators "+v", 1: "+a".
__device__ __forceinline__ void mfma_a(f32x16 &acc, const s16x8 &a, const s16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int ACC, int WHAT, int NSRC = 3>     // WHAT 0 both, 1 mfma only, 2 valu only; NSRC = VGPR sources of the v_pk_fma_f32 (others inline constants)
__global__ __launch_bounds__(256) void real_kernel(int iters, float *sink) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x); b[i] = (short)(0x3c00 + i); }
    f32x2 x[16], y[16], z[16];
    float e[16];
    for (int i = 0; i < 16; ++i) { x[i] = f32x2{0.001f * i, 0.002f * i}; y[i] = f32x2{1.0f, 0.999f}; z[i] = f32x2{0.1f * threadIdx.x, 0.2f}; e[i] = -0.01f * i; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHAT != 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (ACC) { mfma_a(acc0, a, b); mfma_a(acc1, a, b); } else { mfma(acc0, a, b); mfma(acc1, a, b); }
            }
        }
        if constexpr (WHAT != 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (NSRC == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(y[(i + r) & 15]), "v"(z[(i + 2 * r + 1) & 15]), "v"(x[(i + 5) & 15]));
                    else if constexpr (NSRC == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, 1.0" : "=v"(x[i]) : "v"(y[(i + r) & 15]), "v"(x[(i + 5) & 15]));
                    else asm volatile("v_pk_fma_f32 %0, %1, 0.5, 1.0" : "=v"(x[i]) : "v"(x[(i + 5) & 15]));
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i]) : "v"(e[(i + 3) & 15]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + x[i].x + x[i].y + e[i];
    if (s == 12345.678f) sink[0] = s;
}
// Per instruction type: 16 MFMA 32x32x16 (VGPR accumulators) then 64 instructions of ONE type on distinct registers; how much of
// the shorter block hides under the other (4 waves per SIMD)?
template <int OP> __device__ __forceinline__ void op3(float &d, float a, float b, float c, f32x2 &d2, f32x2 a2, f32x2 b2, f32x2 c2) {
    if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else if constexpr (OP == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 3) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else if constexpr (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 5) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a));
    else if constexpr (OP == 6) asm volatile("v_rcp_f32 %0, %1" : "=v"(d) : "v"(a));
    else if constexpr (OP == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(a2), "v"(b2), "v"(c2));
    else if constexpr (OP == 8) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d2) : "v"(a2), "v"(b2));
    else if constexpr (OP == 9) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d2) : "v"(a2), "v"(b2));
    else if constexpr (OP == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(a));
    else if constexpr (OP == 11) asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (OP == 12) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
}
template <int OP, int WHAT>
__global__ __launch_bounds__(256) void type_kernel(int iters, float *sink) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x); b[i] = (short)(0x3c00 + i); }
    float x[16], y[16];
    f32x2 x2[16], y2[16];
    for (int i = 0; i < 16; ++i) { x[i] = 0.5f + 0.001f * i; y[i] = 1.0f - 0.002f * i; x2[i] = f32x2{x[i], y[i]}; y2[i] = f32x2{y[i], x[i]}; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHAT != 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { mfma(acc0, a, b); mfma(acc1, a, b); }
        }
        if constexpr (WHAT != 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) op3<OP>(x[i], y[(i + r) & 15], x[(i + 5) & 15], y[(i + 9 + r) & 15], x2[i], y2[(i + r) & 15], x2[(i + 5) & 15], y2[(i + 9 + r) & 15]);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + x[i] + x2[i].x + x2[i].y;
    if (s == 12345.678f) sink[0] = s;
}
template <int OP, int WHAT>
double run_type(float *sink, int cus) {
    const int iters = 2000, w = 4;
    hipLaunchKernelGGL((type_kernel<OP, WHAT>), dim3(cus * w), dim3(256), 0, 0, iters, sink);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((type_kernel<OP, WHAT>), dim3(cus * w), dim3(256), 0, 0, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6 / iters / w;
}
template <int OP> void type_row(const char *name, float *sink, int cus) {
    const double m = run_type<OP, 1>(sink, cus), v = run_type<OP, 2>(sink, cus), bth = run_type<OP, 0>(sink, cus);
    printf("  %-20s valu alone %5.0f ns (%.2f ns per instruction)   with the 16 MFMAs (%3.0f ns alone) %5.0f ns   hidden %3.0f %% of the shorter block\n",
           name, v, v / 64.0, m, bth, 100.0 * (m + v - bth) / (m < v ? m : v));
    fflush(stdout);
}

template <int ACC, int WHAT, int NSRC = 3>
double run_real(int w, float *sink, int cus) {
    const int iters = 2000;
    hipLaunchKernelGGL((real_kernel<ACC, WHAT, NSRC>), dim3(cus * w), dim3(256), 0, 0, iters, sink);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((real_kernel<ACC, WHAT, NSRC>), dim3(cus * w), dim3(256), 0, 0, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6 / iters / w;       // ns per iteration per SIMD-share
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
    printf("16 MFMA 32x32x16 + 48 v_pk_fma_f32 (three distinct sources) + 16 v_exp_f32 per iteration, blocks: ns per iteration per SIMD-share\n");
    printf("  waves/SIMD   mfma(VGPR acc)  mfma(AGPR acc)   valu    both(VGPR acc)  both(AGPR acc)\n");
    for (int w : {1, 2, 4}) {
        const int cus = prop.multiProcessorCount;
        printf("  %10d %12.0f %15.0f %9.0f %14.0f %15.0f\n", w, run_real<0, 1>(w, sink, cus), run_real<1, 1>(w, sink, cus), run_real<0, 2>(w, sink, cus),
               run_real<0, 0>(w, sink, cus), run_real<1, 0>(w, sink, cus));
        fflush(stdout);
    }
    printf("the same with 2 / 1 VGPR sources per v_pk_fma_f32 (the rest inline constants), 4 waves per SIMD: valu alone, both (VGPR acc)\n");
    {
        const int cus = prop.multiProcessorCount;
        printf("  3 sources: %6.0f %6.0f   2 sources: %6.0f %6.0f   1 source: %6.0f %6.0f   (mfma alone %.0f)\n", run_real<0, 2, 3>(4, sink, cus), run_real<0, 0, 3>(4, sink, cus),
               run_real<0, 2, 2>(4, sink, cus), run_real<0, 0, 2>(4, sink, cus), run_real<0, 2, 1>(4, sink, cus), run_real<0, 0, 1>(4, sink, cus), run_real<0, 1, 3>(4, sink, cus));
        fflush(stdout);
    }
    printf("which vector instructions run beside the matrix pipe (64 of one type on distinct registers after 16 MFMAs, 4 waves per SIMD)\n");
    {
        const int cus = prop.multiProcessorCount;
        type_row<0>("v_fma_f32", sink, cus); type_row<1>("v_add_f32", sink, cus); type_row<2>("v_mul_f32", sink, cus); type_row<11>("v_max_f32", sink, cus);
        type_row<3>("v_max3_f32", sink, cus); type_row<4>("v_cvt_pk_bf16_f32", sink, cus); type_row<5>("v_exp_f32", sink, cus); type_row<6>("v_rcp_f32", sink, cus);
        type_row<7>("v_pk_fma_f32", sink, cus); type_row<8>("v_pk_add_f32", sink, cus); type_row<9>("v_pk_mul_f32", sink, cus);
        type_row<10>("v_mov_b32", sink, cus); type_row<12>("v_add_u32", sink, cus); type_row<13>("v_perm_b32", sink, cus);
    }
    table<0, 32>("v_exp_f32", sink, clk, prop.multiProcessorCount);
    table<1, 32>("v_fma_f32", sink, clk, prop.multiProcessorCount);
    table<1, 96>("v_fma_f32", sink, clk, prop.multiProcessorCount);
    return 0;
}
