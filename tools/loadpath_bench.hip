// loadpath_bench.hip -- how fast can ONE CU pull L2-resident operand tiles, by path (developer tool, DESIGN.md "GEMM" cites it):
//   dma      : global_load_lds_dwordx4, pieces of 8 rows x 128 B (the GEMM's staging form), 8 waves per CU
//   dma16x64 : global_load_lds_dwordx4, pieces of 16 rows x 64 B (the k2 stage form)
//   vgpr_row : global_load_dwordx4 to registers, 1 KiB contiguous per wave instruction
//   vgpr_frag: global_load_dwordx4 to registers, fragment shaped (16 rows x 64 B per wave instruction)
//   mix      : half of the bytes by dma, half by vgpr_frag, issued alternately
// Every workgroup walks the same panels a GEMM workgroup would: a 256-row A panel of its own (row pitch 2 KiB) and a 256-row
// W panel shared by the workgroups of its XCD, K-tile after K-tile, over and over (L2-resident after the first touch).
//   hipcc --offload-arch=gfx950 -O3 tools/loadpath_bench.hip -o /tmp/loadpath_bench && /tmp/loadpath_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16_so(unsigned voff, const void *sbase, unsigned m0_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(m0_dst) : "memory");
}
__device__ __forceinline__ uint4 gload16(unsigned voff, const void *sbase) {
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}

// MODE 0 dma 8x128, 1 dma 16x64, 2 vgpr_row, 3 vgpr_frag, 4 mix (dma 8x128 for A, vgpr_frag for W)
template <int MODE, int NT = 512>
__global__ __launch_bounds__(NT) void pull_kernel(const unsigned char *A, const unsigned char *W, int K_bytes, int iters, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;      // (16-wave run: two waves share a destination; nobody reads it)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)lds);
    const unsigned char *a0 = A + (size_t)(blockIdx.x % 64) * 256 * K_bytes;     // 64 different A panels
    const unsigned char *w0 = W + (size_t)((blockIdx.x / 8) % 4) * 256 * K_bytes; // 4 W panels
    const int nkt = K_bytes / 128;
    unsigned acc = 0;
    // per K-tile (128 B of K per row, 256 rows per operand): 32 KiB per operand = 32 wave-instructions of 1 KiB; 4 + 4 per wave
    unsigned voff8[4], voff16[4], vfrag[4];
    for (int i = 0; i < 4; ++i) {
        voff8[i] = (unsigned)(((4 * wave + i) * 8 + (lane >> 3)) * K_bytes + ((lane & 7) << 4));                    // 8 rows x 128 B
        voff16[i] = (unsigned)((((2 * wave + (i >> 1)) * 16 + (lane >> 2)) * K_bytes) + ((lane & 3) << 4) + (i & 1) * 64);   // 16 rows x 64 B, two k halves
        vfrag[i] = voff16[i];
    }
    uint4 va[8], vb[8];
    for (int i = 0; i < 8; ++i) va[i] = vb[i] = uint4{0, 0, 0, 0};
    auto step = [&](int kt, uint4 (&v)[8], uint4 (&prev)[8]) {     // issue K-tile kt, then wait for / consume the previous one
        const unsigned char *sa = a0 + kt * 128, *sw = w0 + kt * 128;
        const unsigned dst = lds0 + (kt & 1) * 65536 + wave * 4096;
        if constexpr (MODE == 0) {
            for (int i = 0; i < 4; ++i) { glds16_so(voff8[i], sa, dst + i * 1024); glds16_so(voff8[i], sw, dst + 32768 + i * 1024); }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if constexpr (MODE == 1) {
            for (int i = 0; i < 4; ++i) { glds16_so(voff16[i], sa, dst + i * 1024); glds16_so(voff16[i], sw, dst + 32768 + i * 1024); }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if constexpr (MODE == 2 || MODE == 3) {
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = gload16(MODE == 2 ? voff8[i] : vfrag[i], sa);
                v[2 * i + 1] = gload16(MODE == 2 ? voff8[i] : vfrag[i], sw);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(prev[i].x), "+v"(prev[i].w)); acc ^= prev[i].x ^ prev[i].w; }
        } else {
            for (int i = 0; i < 4; ++i) { glds16_so(voff8[i], sa, dst + i * 1024); v[i] = gload16(vfrag[i], sw); }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(prev[i].x), "+v"(prev[i].w)); acc ^= prev[i].x ^ prev[i].w; }
        }
    };
    for (int it = 0; it < iters; ++it)
        for (int kt = 0; kt < nkt; kt += 2) { step(kt, va, vb); step(kt + 1, vb, va); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc ^= reinterpret_cast<unsigned *>(lds)[tid];
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int MODE, int NT = 512>
void run(const char *name, const unsigned char *A, const unsigned char *W, unsigned *sink, int blocks) {
    const int K_bytes = 2048, iters = 40;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pull_kernel<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((pull_kernel<MODE, NT>), dim3(blocks), dim3(NT), 131072, 0, A, W, K_bytes, iters, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((pull_kernel<MODE, NT>), dim3(blocks), dim3(NT), 131072, 0, A, W, K_bytes, iters, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes_per_cu = (double)iters * (K_bytes / 128) * 65536.0 * NT / 512.0;     // every wave moves 8 KiB per K-tile step
    printf("%-10s %7.1f GB/s per CU   %6.2f TB/s chip   (%.3f ms per launch; one 64 KiB K-tile per %.2f us)\n", name,
           bytes_per_cu / (ms[3] * 1e-3) / 1e9, bytes_per_cu * blocks / (ms[3] * 1e-3) / 1e12, ms[3], ms[3] * 1e3 / (iters * (K_bytes / 128)));
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    unsigned char *A, *W; unsigned *sink;
    const size_t a_bytes = (size_t)64 * 256 * 2048, w_bytes = (size_t)4 * 256 * 2048;
    CHECK(hipMalloc(&A, a_bytes)); CHECK(hipMalloc(&W, w_bytes)); CHECK(hipMalloc(&sink, blocks * 4));
    CHECK(hipMemset(A, 1, a_bytes)); CHECK(hipMemset(W, 2, w_bytes));
    printf("%d CUs, 512-thread workgroup per CU, 64 KiB per K-tile (A 32 KiB of the workgroup's own panel + W 32 KiB of a shared panel)\n", blocks);
    run<0>("dma8x128", A, W, sink, blocks);
    run<1>("dma16x64", A, W, sink, blocks);
    run<2>("vgpr_row", A, W, sink, blocks);
    run<3>("vgpr_frag", A, W, sink, blocks);
    run<4>("mix", A, W, sink, blocks);
    printf("waves per CU (same 8 pieces per wave and step; is the rate per wave or per CU?)\n");
    run<0, 256>("dma 4 waves", A, W, sink, blocks);
    run<0, 512>("dma 8 waves", A, W, sink, blocks);
    run<0, 1024>("dma 16 waves", A, W, sink, blocks);
    run<2, 256>("vgpr 4 waves", A, W, sink, blocks);
    run<2, 1024>("vgpr 16 wav", A, W, sink, blocks);
    return 0;
}
