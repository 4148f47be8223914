// bdirect_ceiling.hip -- round-3 feed experiment for the 256x256 GEMM main loop (developer tool, DESIGN.md "GEMM, round 3").
//   hipcc --offload-arch=gfx950 -O3 tools/bdirect_ceiling.hip -o /tmp/bdc && /tmp/bdc
// Round 2 found the shipped loop LDS-bound: per k-step (K = 32) a CU's LDS serves 96 KiB of fragment reads and 32 KiB of LDS-DMA
// writes = its whole bandwidth at MFMA rate.  This tool runs two hand-placed (asm volatile) instruction streams with the GEMM's
// geometry -- 8 waves = 2 (M) x 4 (N), 128x64 wave tiles, 16x16x32 bf16 MFMAs, one barrier per k-step -- on random data with
// L2-resident panels, and prints what each sustains:
//   duo     : the shipped feed: both operands staged by LDS-DMA (4 pieces per wave and k-step), 8 A + 4 W ds_read_b128 per k-step
//   bdirect : the W operand never touches the LDS: every wave loads its own 4 W fragments per k-step straight from L2 into
//             registers (global_load_dwordx4, ring of 3 register sets, prefetch distance 2 k-steps); only A goes through the
//             LDS (2 DMA pieces + 8 reads per wave and k-step): 64 + 16 KiB of LDS traffic per k-step instead of 96 + 32,
//             48 KiB through the texture path instead of 32 (the two waves that share a W column block load the same lines).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Stamp { unsigned long long cyc, rt; };

__device__ __forceinline__ void glds16_so(unsigned voff, const void *sbase, unsigned m0_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(m0_dst) : "memory");
}
__device__ __forceinline__ void gload16(u32x4 &dst, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void mfma16(f32x4 &acc, const u32x4 &b, const u32x4 &a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(a));
}
template <int OFF> __device__ __forceinline__ void dsread(u32x4 &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int swz2(int row) { return (-(row >> 2)) & 3; }

template <bool BDIRECT>
__global__ __launch_bounds__(512, 2) void feed_kernel(const uint4 *seed, float *sink, Stamp *stamps, int iters, const unsigned char *panel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int STAGE = BDIRECT ? 16384 : 32768, OPS = 16384, SB = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * STAGE / 16; i += 512) reinterpret_cast<uint4 *>(lds)[i] = seed[(blockIdx.x * 131 + i) & 4095];
    __syncthreads();
    const int wr = wave >> 2, wc = wave & 3, r = lane & 15, q = lane >> 4;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)lds);
    const unsigned char *pa = panel + (size_t)(blockIdx.x % 64) * 256 * 2048, *pw = panel + (size_t)(64 + (blockIdx.x / 8) % 4) * 256 * 2048;
    // LDS-DMA pieces: 16 rows x 64 B each.  duo: waves 0..3 stage the W image, 4..7 the A image, 4 pieces each per k-step;
    // bdirect: all 8 waves stage the A image, 2 pieces each
    constexpr int NP = BDIRECT ? 2 : 4;
    const bool is_w = !BDIRECT && wave < 4;
    const int pw_i = BDIRECT ? wave : (wave & 3);
    unsigned voff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int rt = (NP * pw_i + i) * 16 + (lane >> 2);
        voff[i] = (unsigned)(rt * 2048 + (((lane & 3) ^ swz2(rt)) << 4));
    }
    const unsigned char *src0 = is_w ? pw : pa;
    const unsigned piece0 = lds0 + (is_w ? OPS : 0) + (NP * pw_i) * 1024;
    const unsigned fa_lo = lds0 + (unsigned)((wr * 128 + r) * SB + ((q ^ swz2(r)) << 4)), fa_hi = fa_lo + 2 * STAGE;
    const unsigned fw_lo = lds0 + OPS + (unsigned)((wc * 64 + r) * SB + ((q ^ swz2(r)) << 4)), fw_hi = fw_lo + 2 * STAGE;
    // bdirect: this lane's 16 bytes of W row (wc*64 + ni*16 + r) at k-slot q; + ni * 16 rows, + 64 B per k-step
    unsigned wv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wv[ni] = (unsigned)((wc * 64 + ni * 16 + r) * 2048 + q * 16);

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    u32x4 fa[2][8], fw[3][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dsread<0>(fa[0][i], fa_lo + i * 1024); fa[1][i] = fa[0][i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { dsread<0>(fw[0][i], fa_lo + i * 1024 + 512); fw[1][i] = fw[0][i]; fw[2][i] = fw[0][i]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    // one k-step: CUR = A fragment set in use, WC = W fragment set in use (bdirect: ring of 3), SLOT = ring slot of this stage
    auto kstep = [&](int it, auto curc, auto wcc, auto slotc) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value, WCUR = decltype(wcc)::value, SLOT = decltype(slotc)::value;
        constexpr int SN = (SLOT + 1) & 3;
        constexpr int OFFN = (SN & 1) * STAGE;
        const unsigned fan = SN >= 2 ? fa_hi : fa_lo, fwn = SN >= 2 ? fw_hi : fw_lo;
        const unsigned char *src = src0 + ((it * 6 + SLOT) & 31) * SB;
        const unsigned char *wsrc = pw + ((it * 6 + SLOT + 2) & 31) * SB;
        static_for<0, 32>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, mi = j >> 2, ni = j & 3;
            if constexpr (BDIRECT) mfma16(acc[mi][ni], fw[WCUR][ni], fa[CUR][mi]);
            else mfma16(acc[mi][ni], fw[CUR][ni], fa[CUR][mi]);
            if constexpr (BDIRECT) {
                // next k-step's 8 A fragments at j = 1, 3, .., 15; W fragments of k-step s + 2 (ring set (WCUR + 2) % 3) at
                // j = 2, 6, 10, 14; the 2 A pieces of stage s + 4 at j = 18, 22
                if constexpr ((j & 1) == 1 && j / 2 < 8) dsread<OFFN + (j / 2) * 1024>(fa[CUR ^ 1][j / 2], fan);
                if constexpr ((j & 3) == 2 && j < 16) gload16(fw[(WCUR + 2) % 3][j >> 2], wv[j >> 2], wsrc);
                if constexpr (j == 18 || j == 22) glds16_so(voff[(j - 18) >> 2], src, piece0 + SLOT * STAGE + ((j - 18) >> 2) * 1024);
                // W of k-step s + 1 (issued one k-step ago) and A stage s + 1 (four k-steps ago) must have landed: younger than
                // them are DMA(s+3) x 2, W(s+2) x 4, DMA(s+4) x 2
                if constexpr (j == 28) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else {
                if constexpr ((j & 1) == 1 && j / 2 < 12) {
                    constexpr int i = j / 2;
                    if constexpr (i < 4) dsread<OFFN + i * 1024>(fw[CUR ^ 1][i], fwn);
                    else dsread<OFFN + (i - 4) * 1024>(fa[CUR ^ 1][i - 4], fan);
                }
                if constexpr ((j & 7) == 2) glds16_so(voff[j >> 3], src, piece0 + SLOT * STAGE + (j >> 3) * 1024);
                if constexpr (j == 28) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            }
            if constexpr (j == 29) asm volatile("s_barrier" ::: "memory");
        });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    for (int it = 0; it < iters; ++it) {          // 12 k-steps per iteration (lcm of the A double buffer, the W ring of 3, the 4 slots)
        kstep(it, I0{}, I0{}, I0{}); kstep(it, I1{}, I1{}, I1{}); kstep(it, I0{}, I2{}, I2{}); kstep(it, I1{}, I0{}, I3{});
        kstep(it, I0{}, I1{}, I0{}); kstep(it, I1{}, I2{}, I1{}); kstep(it, I0{}, I0{}, I2{}); kstep(it, I1{}, I1{}, I3{});
        kstep(it, I0{}, I2{}, I0{}); kstep(it, I1{}, I0{}, I1{}); kstep(it, I0{}, I1{}, I2{}); kstep(it, I1{}, I2{}, I3{});
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float acc_out = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc_out += acc[m][n][0] + acc[m][n][3];
    if (lane == 0 && wave == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
    if (acc_out == 12345.678f) sink[blockIdx.x] = acc_out;
}

template <bool BDIRECT>
void run(const char *name, const uint4 *seed, float *sink, Stamp *stamps, int blocks, const unsigned char *panel) {
    const int lds_bytes = 4 * (BDIRECT ? 16384 : 32768);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(feed_kernel<BDIRECT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const int iters = 700;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((feed_kernel<BDIRECT>), dim3(blocks), dim3(512), lds_bytes, 0, seed, sink, stamps, iters, panel);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int rr = 0; rr < 7; ++rr) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((feed_kernel<BDIRECT>), dim3(blocks), dim3(512), lds_bytes, 0, seed, sink, stamps, iters, panel);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<Stamp> h(blocks);
    CHECK(hipMemcpy(h.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto &s : h) { clk.push_back((double)s.cyc / (double)s.rt * 100.0); cyc.push_back((double)s.cyc); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double ksteps = (double)iters * 12;
    const double flops = (double)blocks * 8 * ksteps * (128.0 * 64 * 32 * 2);
    printf("%-8s %8.1f TFLOP/s (median of 7; min-time %8.1f)   cycles per k-step %.0f (1024 = MFMA-paced, two waves per SIMD)   in-kernel clock %.0f MHz\n",
           name, flops / (ms[3] * 1e-3) / 1e12, flops / (ms[0] * 1e-3) / 1e12, cyc[blocks / 2] / ksteps, clk[blocks / 2]);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    std::vector<unsigned> host(4096 * 4);
    unsigned s = 12345u;
    for (auto &v : host) {
        s = s * 1664525u + 1013904223u; unsigned lo = s >> 16;
        s = s * 1664525u + 1013904223u; unsigned hi = s >> 16;
        auto fix = [](unsigned h) { return (h & 0x807Fu) | ((124u + (h >> 7) % 4u) << 7); };
        v = fix(lo & 0xFFFF) | (fix(hi & 0xFFFF) << 16);
    }
    uint4 *seed; float *sink; Stamp *stamps;
    CHECK(hipMalloc(&seed, host.size() * 4)); CHECK(hipMalloc(&sink, blocks * 4)); CHECK(hipMalloc(&stamps, blocks * sizeof(Stamp)));
    CHECK(hipMemcpy(seed, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    unsigned char *panel;
    CHECK(hipMalloc(&panel, (size_t)68 * 256 * 2048));
    CHECK(hipMemset(panel, 0x3c, (size_t)68 * 256 * 2048));
    printf("device: %s, %d CUs\n", prop.gcnArchName, blocks);
    for (int rep = 0; rep < 2; ++rep) {
        run<false>("duo", seed, sink, stamps, blocks, panel);
        run<true>("bdirect", seed, sink, stamps, blocks, panel);
    }
    return 0;
}
