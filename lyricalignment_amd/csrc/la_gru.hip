// la_gru.hip -- persistent bidirectional GRU recurrence (nn.GRU, gate order r,z,n;
// module/align_model.py:23-28,36).  The input projections W_ih x + b_ih of both directions are one
// big GEMM (la_gemm) done beforehand; this kernel runs the T strictly sequential steps
//     r = sigmoid(gi_r + W_hr h + b_hr)      z = sigmoid(gi_z + W_hz h + b_hz)
//     n = tanh(gi_n + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
// for both directions at once.
//
// Decomposition (DESIGN.md "GRU"): W_hh of one direction (3H x H) does not fit one CU, so the H
// hidden units are split over H/(16*NW) workgroups per direction; each of a workgroup's NW waves
// owns 16 hidden units and keeps the 48 matching rows of W_hh RESIDENT for the whole sequence
// (bf16: in VGPRs as MFMA B-fragments; f32: in LDS), so a step costs only the h exchange.  Per step
// a wave computes [batch x H] * [H x 48] with MFMA 16x16 tiles (batch on the rows), applies the
// gates lane-locally (the r/z/n accumulators share one layout), writes its h slice into `out`
// (which IS the exchange buffer: the next layer / the FC read it anyway) and signals a per-step
// arrival counter.  Hand-off (cdna guide, Guideline 16), default = the write-through form: every
// handed-off byte is stored sc1 -> every wave vmcnt(0) -> barrier -> lane 0 relaxed agent-scope
// counter add; consumers poll the counter with sc1 loads, pass a barrier, then read with sc1 loads
// only.  LA_GRU_FENCE=1 selects the release / acquire form instead (plain stores -> vmcnt(0) ->
// barrier -> lane 0 release fence -> counter add; one lane acquires, barrier, plain loads), 1.7x
// slower per step.  Every wait is bounded.
// Clips are independent, so batches larger than 16 become extra workgroup groups (grid.z).
#include <algorithm>
#include <type_traits>

#include "la_common.h"

using la::bf16_t;

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int GROUP = 16;      // clips per workgroup group = one 16-row MFMA batch tile (MT = 1).  32-clip groups (MT = 2: two batch
                               // tiles per workgroup, kept as a template parameter) halve the workgroups but every step then does
                               // twice the MFMAs, gates and exchange bytes on its critical path: 4.40 vs 3.62 us per step at 32
                               // clips.  More clips = more groups running side by side on other CUs, not longer steps.

struct GruParams {
    const float *gi;     // [B][T][2][3H]
    const void *w_hh;    // [2][3H][H]
    const float *b_hh;   // [2][3H]
    void *out;           // [B][T][2H]
    void *out_mish;      // optional
    int B, T, H;
    unsigned *counters;  // [groups][2][T] arrival counters, zeroed per call
    int *abort_flag;     // workspace word, zeroed per call
    int *timeout_flag;   // caller's (optional)
    int nsplit;
    float *gates;        // optional [B][T][2][4H] f32: r, z, n, (W_hn h + b_hn) of every step, for the backward sweep
    unsigned long long poll_budget = 300000000ull;   // bound of one inter-workgroup wait in 100 MHz ticks (option gru_timeout_us; default 3 s)
    int fault_step = 0;  // test hook (option gru_fault_step, one launch): workgroup (0, 0, 0) leaves at this step without publishing, as if it had never become resident
};

// Gate non-linearities on the hardware exponential / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each).  The libm forms
// (expf + IEEE divide, tanhf, log1pf) cost ~200 instructions per (clip, unit) pair; 8 pairs per lane and step made the
// gate arithmetic -- not the hand-off -- a third of every step (14.9 -> 9.5 ms per layer at B=32, T=1500).  Absolute errors
// stay at the 1e-7 level (sigmoid 9e-8, tanh 2e-7, Mish 1e-7): the parity tests are unchanged.
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_fast(float x) {      // 1 - 2 / (e^{2x} + 1): exact limits at +-inf
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)), 1.0f);
}
__device__ __forceinline__ float mish_fast(float x) {      // x tanh(log(1 + e^x)) = x n / (n + 2), n = e^x (e^x + 2); torch's softplus threshold 20
    const float w = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 20.0f));
    const float n = w * (w + 2.0f);
    return x > 20.0f ? x : x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}

// h rounded to the 16-bit storage type from its FLOAT32 value (as the reference's half-precision nn.GRU hands h on): the empty asm keeps
// hipcc from fusing the gate's last fma with the conversion (v_fma_mixlo_f16 rounds the unrounded fma result once -- a different
// half in ~1 % of the cases, and only in some of the kernels that share this code: the hand-off forms then disagreed in float16)
template <typename T>
__device__ __forceinline__ unsigned short round16(float v) {
    asm volatile("" : "+v"(v));
    T h;
    la::Elem<T>::store(&h, v);
    return __builtin_bit_cast(unsigned short, h);
}

// bounded wait for `*ctr >= target`; returns false on timeout / abort
__device__ __forceinline__ bool wait_counter(unsigned *ctr, unsigned target, int *abort_flag, unsigned long long budget = 300000000ull) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {  // 3 s by default (option gru_timeout_us)
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

template <typename T> struct GruTraits;
template <> struct GruTraits<bf16_t> {
    static constexpr int NW = 4;        // waves per workgroup
    static constexpr int KSTEP = 32;    // k elements per 16-byte-per-lane fragment step
    static constexpr bool W_IN_REGS = true;
};
template <> struct GruTraits<la::f16_t> {
    static constexpr int NW = 4;
    static constexpr int KSTEP = 32;
    static constexpr bool W_IN_REGS = true;
};
template <> struct GruTraits<float> {
    static constexpr int NW = 2;
    static constexpr int KSTEP = 16;
    static constexpr bool W_IN_REGS = false;
};

__device__ __forceinline__ void mma_step(const uint4 &a, const uint4 &w, f32x4 &acc, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, w), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const uint4 &a, const uint4 &w, f32x4 &acc, la::f16_t) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, w), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const uint4 &a, const uint4 &w, f32x4 &acc, float) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(w.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(w.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(w.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(w.w), acc, 0, 0, 0);
}

// MAXKS: compile-time bound on H / KSTEP (register-resident W for bf16: H <= 512)
// WT: write-through hand-off (cdna guide G16, valid-forms row 1): h is stored with sc1 (write-through) 4-byte stores,
//     every storing wave drains vmcnt, one lane signals with a relaxed agent-scope add; consumers poll that counter with
//     an sc1 load, pass a workgroup barrier and read h with sc1 16-byte buffer loads only -> no release / acquire fence
//     (each costs ~1.7 us per step here).  WT = true is the default; LA_GRU_FENCE=1 selects the fence form (WT = false).
// NW: waves per workgroup.  16-bit modes: 8 where hidden is a multiple of 128 (two waves per SIMD at <= 256 registers each: a
//     wave's W_hh slice is 12 * MAXKS registers, 144 at hidden <= 384) -- one (direction, 16-clip group) is then hidden / 128
//     workgroups = CUs instead of hidden / 64: the resident recurrence takes 24 instead of 48 CUs from the encoder's GEMMs at 64
//     clips, and a step's hand-off has 3 instead of 6 participants.  (768-thread workgroups -- 16 CUs -- do not fit: three waves
//     per SIMD leave 168 registers per wave, 144 of which the W slice takes; acc + gates + prefetched inputs need ~50 more.)
template <typename T, int MAXKS, bool WT, int MT, int NW = GruTraits<T>::NW>
__global__ __launch_bounds__(NW * 64, 1) void gru_kernel(GruParams p) {
    typedef GruTraits<T> TR;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int slice = blockIdx.x, dir = blockIdx.y, group = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int H = p.H, T_ = p.T;
    const int nks = H / TR::KSTEP;
    const int j0 = (slice * NW + wave) * 16;       // first hidden unit of this wave
    const int jcol = j0 + r16;                      // this lane's hidden unit (C layout: col = lane & 15)
    const int b0 = group * GROUP;
    const int nb = min(GROUP, p.B - b0);            // clips in this group
    const T *Wd = reinterpret_cast<const T *>(p.w_hh) + (int64_t)dir * 3 * H * H;
    const float *bh = p.b_hh + dir * 3 * H;
    T *out = reinterpret_cast<T *>(p.out);
    T *outm = reinterpret_cast<T *>(p.out_mish);
    const int64_t out_bs = (int64_t)T_ * 2 * H, out_ts = 2 * H;
    const int64_t gi_bs = (int64_t)T_ * 6 * H, gi_ts = 6 * H;
    unsigned *ctr = p.counters + ((int64_t)group * 2 + dir) * T_;
    // buffer descriptor over `out` for the sc1 (L1-bypassing) accesses of the write-through hand-off
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.out, 0, (int)((int64_t)p.B * out_bs * (int64_t)sizeof(T)), 0x00020000);

    // ---- resident W_hh slice: B-operand fragments, lane (n = r16, q): W[g*H + j0 + n][KSTEP*ks + (16/sizeof T)*q ..] ----
    uint4 wreg[TR::W_IN_REGS ? 3 : 1][TR::W_IN_REGS ? MAXKS : 1];
    const int row_bytes = H * (int)sizeof(T);
    if constexpr (TR::W_IN_REGS) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks)
                if (ks < nks)
                    wreg[g][ks] = *reinterpret_cast<const uint4 *>(
                        reinterpret_cast<const unsigned char *>(Wd + (int64_t)(g * H + jcol) * H) + ks * 64 + q * 16);
    } else {
        // LDS image [wave][gate][16 rows][row_bytes], 16-B slot s of row n stored at s ^ n (low 4 bits)
        const int slots = row_bytes / 16;
        for (int idx = tid; idx < NW * 3 * 16 * slots; idx += NW * 64) {
            const int s = idx % slots, n = (idx / slots) % 16, g = (idx / (slots * 16)) % 3, w = idx / (slots * 48);
            const uint4 v = *reinterpret_cast<const uint4 *>(
                reinterpret_cast<const unsigned char *>(Wd + (int64_t)(g * H + (slice * NW + w) * 16 + n) * H) + s * 16);
            *reinterpret_cast<uint4 *>(lds + 16 + ((int64_t)((w * 3 + g) * 16 + n)) * row_bytes + ((s ^ n) << 4)) = v;
        }
        __syncthreads();
    }
    int *ok_s = reinterpret_cast<int *>(lds);  // first 16 B of the dynamic region (no static LDS: keeps it 16-B aligned)
    const unsigned char *wl = lds + 16 + (int64_t)(wave * 3) * 16 * row_bytes;

    const float bhr = bh[jcol], bhz = bh[H + jcol], bhn = bh[2 * H + jcol];
    float hprev[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) hprev[mt][i] = 0.f;

    // clip rows of this lane: A-fragment row (b0 + mt*16 + r16), C rows (b0 + mt*16 + 4q + i)
    int arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = b0 + min(mt * 16 + r16, nb - 1);

    float gin[MT][4][3];  // prefetched input projections of the current step
    auto load_gi = [&](int t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bl = min(mt * 16 + 4 * q + i, nb - 1);
                const float *g = p.gi + (int64_t)(b0 + bl) * gi_bs + (int64_t)t * gi_ts + dir * 3 * H + jcol;
                gin[mt][i][0] = g[0];
                gin[mt][i][1] = g[H];
                gin[mt][i][2] = g[2 * H];
            }
    };
    load_gi(dir == 0 ? 0 : T_ - 1);

    bool alive = true;
    for (int step = 0; step < T_; ++step) {
        if (p.fault_step > 0 && step == p.fault_step && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) return;   // test hook: see the parameter
        const int t = dir == 0 ? step : T_ - 1 - step;
        const int tprev = dir == 0 ? t - 1 : t + 1;
        f32x4 acc[3][MT];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            // ---- wait until every slice of this (direction, group) has published h_{t-1} ----
            if (tid == 0) {
                const bool ok = wait_counter(ctr + (step - 1), (unsigned)p.nsplit, p.abort_flag, p.poll_budget);
                if (!WT) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *ok_s = ok ? 1 : 0;
            }
            __syncthreads();
            alive = *ok_s != 0;
            if (!alive) break;
            // ---- gh = h_{t-1} W_hh^T for this wave's 48 gate columns ----
            if constexpr (TR::W_IN_REGS) {
                // Every wave needs ALL of h_{t-1} (16*MT clips x H): fetched once per workgroup into LDS, 16 B per thread and
                // request, then read as MFMA A fragments from there.  (Each wave loading its own copy made the loads 4x
                // redundant: 96 KB per CU and step from beyond this XCD's L2 -- measured 3.7 us of a 6.4 us step at 32 clips.)
                // Row pitch H*2 + 16 B: the 16 rows of an m-tile start 16 B apart modulo 256 B -> conflict-free b128 reads.
                const int chunks = row_bytes >> 4, pitch = row_bytes + 16;
                unsigned char *hl = lds + 16;
                // all of this thread's requests go out before the first one is waited for (a rolled load -> LDS-store loop
                // serialised 3 / 6 memory round trips: that, not the distance to the other XCDs, was the 2.7-3.7 us fetch)
                constexpr int NREQ = (16 * MT * MAXKS * 4 + NW * 64 - 1) / (NW * 64);
                uint4 hv[NREQ];
#pragma unroll
                for (int k = 0; k < NREQ; ++k) {
                    const int idx = tid + k * NW * 64;
                    if (idx < 16 * MT * chunks) {
                        const int row = idx / chunks, c = idx - row * chunks;
                        const int64_t eoff = (int64_t)(b0 + min(row, nb - 1)) * out_bs + (int64_t)tprev * out_ts + dir * H;
                        if constexpr (WT) {
                            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                            const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(out_rsrc, (int)(eoff * (int64_t)sizeof(T)) + c * 16, 0, 16 /* sc1 */);
                            hv[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                        } else {
                            hv[k] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(out + eoff) + c * 16);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NREQ; ++k) {
                    const int idx = tid + k * NW * 64;
                    if (idx < 16 * MT * chunks) {
                        const int row = idx / chunks, c = idx - row * chunks;
                        *reinterpret_cast<uint4 *>(hl + row * pitch + c * 16) = hv[k];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks) {
                    if (ks < nks) {
                        uint4 a[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            a[mt] = *reinterpret_cast<const uint4 *>(hl + (mt * 16 + r16) * pitch + ks * 64 + q * 16);
#pragma unroll
                        for (int g = 0; g < 3; ++g)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) mma_step(a[mt], wreg[g][ks], acc[g][mt], T{});
                    }
                }
            } else if (MT == 1 && nb <= 4) {
                // <= 4 clips (the fine-tune batch): a 16x16x4 f32 MFMA spends its 32 cycles on 16 batch rows of which <= 4 exist --
                // 288 of them per wave and step, 4.4 us.  Here each lane multiplies its 4-k slice of the 48 weight rows with the
                // clips' h directly (v_fma_f32, 4 x fewer issue cycles at 2 clips), and the four k-quarters (q) are added across lanes.
                auto small = [&](auto nbc) {
                    constexpr int NBV = decltype(nbc)::value;
                    float part[3][NBV];
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int b = 0; b < NBV; ++b) part[g][b] = 0.f;
#pragma unroll
                    for (int kb = 0; kb < MAXKS; kb += 12) {
                        uint4 hb[12][NBV];
#pragma unroll
                        for (int u = 0; u < 12; ++u)
#pragma unroll
                            for (int b = 0; b < NBV; ++b)
                                if (kb + u < nks) {
                                    const int64_t eoff = (int64_t)(b0 + b) * out_bs + (int64_t)tprev * out_ts + dir * H;
                                    if constexpr (WT) {
                                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                            out_rsrc, (int)(eoff * (int64_t)sizeof(T)) + (kb + u) * 64 + q * 16, 0, 16 /* sc1 */);
                                        hb[u][b] = make_uint4(v[0], v[1], v[2], v[3]);
                                    } else {
                                        hb[u][b] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(out + eoff) + (kb + u) * 64 + q * 16);
                                    }
                                }
#pragma unroll
                        for (int u = 0; u < 12; ++u)
                            if (kb + u < nks) {
                                const int ks = kb + u;
#pragma unroll
                                for (int g = 0; g < 3; ++g) {
                                    const uint4 w = *reinterpret_cast<const uint4 *>(wl + (int64_t)(g * 16 + r16) * row_bytes + (((ks * 4 + q) ^ r16) << 4));
#pragma unroll
                                    for (int b = 0; b < NBV; ++b) {
                                        float a = part[g][b];
                                        a = fmaf(__uint_as_float(w.x), __uint_as_float(hb[u][b].x), a);
                                        a = fmaf(__uint_as_float(w.y), __uint_as_float(hb[u][b].y), a);
                                        a = fmaf(__uint_as_float(w.z), __uint_as_float(hb[u][b].z), a);
                                        a = fmaf(__uint_as_float(w.w), __uint_as_float(hb[u][b].w), a);
                                        part[g][b] = a;
                                    }
                                }
                            }
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int b = 0; b < NBV; ++b) {
                            float v = part[g][b];
                            v += __shfl_xor(v, 16);
                            v += __shfl_xor(v, 32);
                            acc[g][0][b] = v;               // clip b = C row b of the q = 0 lanes (rows >= nb are never stored)
                        }
                };
                if (nb == 1) small(std::integral_constant<int, 1>{});
                else if (nb == 2) small(std::integral_constant<int, 2>{});
                else if (nb == 3) small(std::integral_constant<int, 3>{});
                else small(std::integral_constant<int, 4>{});
            } else {
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks) {
                    if (ks < nks) {
                        uint4 a[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const int64_t eoff = (int64_t)arow[mt] * out_bs + (int64_t)tprev * out_ts + dir * H;
                            if constexpr (WT) {
                                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                    out_rsrc, (int)(eoff * (int64_t)sizeof(T)) + ks * 64 + q * 16, 0, 16 /* sc1 */);
                                a[mt] = make_uint4(v[0], v[1], v[2], v[3]);
                            } else {
                                a[mt] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(out + eoff) + ks * 64 + q * 16);
                            }
                        }
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const uint4 w = *reinterpret_cast<const uint4 *>(wl + (int64_t)(g * 16 + r16) * row_bytes + (((ks * 4 + q) ^ r16) << 4));
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) mma_step(a[mt], w, acc[g][mt], T{});
                        }
                    }
                }
            }
        }
        // ---- gates (lane-local: acc[g][mt][i] <-> clip b0 + mt*16 + 4q + i, hidden unit jcol) ----
        float hnew[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = sigm(gin[mt][i][0] + (acc[0][mt][i] + bhr));
                const float z = sigm(gin[mt][i][1] + (acc[1][mt][i] + bhz));
                const float hn = acc[2][mt][i] + bhn;
                const float n = tanh_fast(gin[mt][i][2] + r * hn);
                hnew[mt][i] = (1.0f - z) * n + z * hprev[mt][i];
                hprev[mt][i] = hnew[mt][i];
                if (p.gates) {
                    const int bl = mt * 16 + 4 * q + i;
                    if (bl < nb) {
                        float *gp = p.gates + (((int64_t)(b0 + bl) * T_ + t) * 2 + dir) * 4 * H + jcol;
                        gp[0] = r; gp[H] = z; gp[2 * H] = n; gp[3 * H] = hn;
                    }
                }
            }
        // ---- publish h_t into out (and Mish(h_t) into out_mish) ----
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bl = mt * 16 + 4 * q + i;
                const int64_t o = (int64_t)(b0 + (bl < nb ? bl : 0)) * out_bs + (int64_t)t * out_ts + dir * H + jcol;
                if constexpr (WT) {
                    if constexpr (sizeof(T) == 2) {
                        // pair (jcol, jcol+1) -> one 4-byte write-through store by the even lane (neighbour = lane + 1)
                        const unsigned mine = round16<T>(hnew[mt][i]);
                        const unsigned nb_bits = (unsigned)__shfl_down((int)mine, 1);
                        if (bl < nb && (r16 & 1) == 0)
                            __builtin_amdgcn_raw_buffer_store_b32(mine | (nb_bits << 16), out_rsrc, (int)(o * 2), 0, 16 /* sc1 */);
                    } else {
                        if (bl < nb)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hnew[mt][i]), out_rsrc, (int)(o * 4), 0, 16 /* sc1 */);
                    }
                } else if (bl < nb) {
                    if constexpr (sizeof(T) == 2) *reinterpret_cast<unsigned short *>(out + o) = round16<T>(hnew[mt][i]);
                    else la::Elem<T>::store(out + o, hnew[mt][i]);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (!WT) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(ctr + step, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (step + 1 < T_) load_gi(dir == 0 ? t + 1 : t - 1);  // independent of h: in flight during the next wait
        // Mish(h_t) for the FC is nobody's input inside the recurrence: computed and stored AFTER the hand-off signal,
        // under the other slices' step (its libm-grade softplus / tanh took ~0.4 us of every step's critical path before)
        if (outm) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bl = mt * 16 + 4 * q + i;
                    if (bl < nb) {
                        const int64_t o = (int64_t)(b0 + bl) * out_bs + (int64_t)t * out_ts + dir * H + jcol;
                        la::Elem<T>::store(outm + o, mish_fast(hnew[mt][i]));
                    }
                }
        }
    }
    if (!alive && tid == 0 && p.timeout_flag) *p.timeout_flag = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Granule hand-off form of the 16-bit recurrence (8-wave workgroups, MT = 1): the data IS the flag.
// The counter form above pays four dependent memory round trips per step -- producers' write-through stores are drained
// (vmcnt(0)), a barrier, one lane's counter add; the consumer's poll sees the add, a barrier, THEN the h loads go out.  Here
// every pair of hidden units leaves as ONE naturally aligned 8-byte sc1 store {h[c][2j], h[c][2j+1] | tag = step + 1}
// (cdna guide Guideline 16, recipe R2: a granule written by one store needs no ordering with anything else) into a ring
// of two step slots per (group, direction); consumers poll the granules themselves with 16-byte sc1 loads (two granules
// each) and keep what has the step's tag: no drain, no counter, no atomics -- one barrier per step (LDS double-buffered),
// the stores fire and are forgotten.  Slot reuse is safe with two slots: a workgroup can write step t + 2 (the slot of step t)
// only after it has consumed every other workgroup's step t + 1, which those produce only after they have read step t.
// `out` (the layer's result for the next GEMM / the FC) is written with plain stores off the critical path.
// Every wait is bounded (abort flag + 3 s).
struct GruGranuleParams {
    const float *gi;
    const void *w_hh;
    const float *b_hh;
    void *out, *out_mish;
    int B, T, H;
    unsigned long long *xch;   // [groups][2 dirs][2 slots][16 clips][H / 2] granules, zeroed per call
    int *abort_flag, *timeout_flag;
    int nsplit;
    int poll_delay;            // 64-clock sleeps between a step's publish and its first poll (option gru_poll_delay)
    unsigned long long poll_budget = 300000000ull;
    int fault_step = 0;
};

#ifndef LA_GRU_PROBE
#define LA_GRU_PROBE 0      // experiment build: timing-only knock-outs (1 = no gi loads after step 0, 2 = no out / out_mish stores)
#endif
template <typename T, int MAXKS>
__global__ __launch_bounds__(512, 1) void gru_granule_kernel(GruGranuleParams p) {
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int slice = blockIdx.x, dir = blockIdx.y, group = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int H = p.H, T_ = p.T;
    const int nks = H / 32;
    const int j0 = (slice * NW + wave) * 16;
    const int jcol = j0 + r16;
    const int b0 = group * GROUP;
    const int nb = min(GROUP, p.B - b0);
    const T *Wd = reinterpret_cast<const T *>(p.w_hh) + (int64_t)dir * 3 * H * H;
    const float *bh = p.b_hh + dir * 3 * H;
    T *out = reinterpret_cast<T *>(p.out);
    T *outm = reinterpret_cast<T *>(p.out_mish);
    const int64_t out_bs = (int64_t)T_ * 2 * H, out_ts = 2 * H;
    const int64_t gi_bs = (int64_t)T_ * 6 * H, gi_ts = 6 * H;
    const int gpr = H / 2;                                     // granules per clip row
    const int slot_bytes = 16 * gpr * 8;
    unsigned char *xbase = reinterpret_cast<unsigned char *>(p.xch) + ((int64_t)group * 2 + dir) * 2 * slot_bytes;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xbase, 0, 2 * slot_bytes, 0x00020000);

    uint4 wreg[3][MAXKS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < nks)
                wreg[g][ks] = *reinterpret_cast<const uint4 *>(
                    reinterpret_cast<const unsigned char *>(Wd + (int64_t)(g * H + jcol) * H) + ks * 64 + q * 16);
    int *ok_s = reinterpret_cast<int *>(lds);
    if (tid == 0) *ok_s = 1;
    const int row_bytes = H * 2, pitch = row_bytes + 16;
    unsigned char *hl0 = lds + 16, *hl1 = hl0 + 16 * pitch;

    const float bhr = bh[jcol], bhz = bh[H + jcol], bhn = bh[2 * H + jcol];
    float hprev[4] = {0.f, 0.f, 0.f, 0.f};
    float gin[4][3];  // prefetched input projections of the coming step
    auto load_gi = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bl = min(4 * q + i, nb - 1);
            const float *g = p.gi + (int64_t)(b0 + bl) * gi_bs + (int64_t)t * gi_ts + dir * 3 * H + jcol;
            gin[i][0] = g[0]; gin[i][1] = g[H]; gin[i][2] = g[2 * H];
        }
    };
    // h of frame t (this lane's four clips) to `out` / `out_mish`: plain stores, nobody inside this launch reads them
    auto store_out = [&](int t, const float (&h)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mine = round16<T>(h[i]);
            const unsigned nbr = (unsigned)__shfl_down((int)mine, 1);
            const int bl = 4 * q + i;
            const int64_t o = (int64_t)(b0 + (bl < nb ? bl : 0)) * out_bs + (int64_t)t * out_ts + dir * H + jcol;
            if (bl < nb && (r16 & 1) == 0) *reinterpret_cast<unsigned *>(out + o) = mine | (nbr << 16);
            if (outm && bl < nb) la::Elem<T>::store(outm + o, mish_fast(h[i]));
        }
    };
    load_gi(dir == 0 ? 0 : T_ - 1);
    // this thread's chunks of a slot: 16 bytes = two granules = units 4c' .. 4c'+3 of one clip; chunk id = tid + k * 512
    constexpr int NCH = (16 * MAXKS * 16 + 511) / 512;          // chunks per thread: 16 clips x (H / 4) chunks, H <= 32 MAXKS
    const int cpr = H / 4;                                      // chunks per clip row
    const int nchunks = 16 * cpr;
    __syncthreads();

    bool alive = true;
    for (int step = 0; step < T_; ++step) {
        if (p.fault_step > 0 && step == p.fault_step && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) return;   // test hook: see the parameter
        const int t = dir == 0 ? step : T_ - 1 - step;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char *hl = (step & 1) ? hl1 : hl0;
        if (step > 0) {
            // ---- h_{t-1} of all 16 clips x H units: poll the previous step's slot until every granule carries its tag ----
            const unsigned want = (unsigned)step;                       // tag of step - 1 is (step - 1) + 1
            const int sbase = ((step - 1) & 1) * slot_bytes;
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (tid + k * 512 < nchunks) pending |= 1u << k;
            // The other workgroups published at about the time this one did, and their stores take most of a memory round trip to become
            // visible: a poll issued at once comes back stale and the retry costs a second full round trip.  A short sleep first lets
            // the first poll find the data (tools/kbench.py gru_delay).
            for (int d = 0; d < p.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            while (pending) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                u32x4 v[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if (pending & (1u << k)) v[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, sbase + (tid + k * 512) * 16, 0, 16 /* sc1 */);
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if ((pending & (1u << k)) && v[k][1] == want && v[k][3] == want) {
                        const int idx = tid + k * 512;
                        const int row = idx / cpr, c = idx - row * cpr;
                        *reinterpret_cast<uint2 *>(hl + row * pitch + c * 8) = make_uint2(v[k][0], v[k][2]);
                        pending &= ~(1u << k);
                    }
                if (pending && (++spins & 63u) == 0) {
                    if (__hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { *ok_s = 0; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > p.poll_budget) {
                        __hip_atomic_store(p.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *ok_s = 0;
                        break;
                    }
                }
            }
            __syncthreads();
            alive = *ok_s != 0;
            if (!alive) break;
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks) {
                if (ks < nks) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(hl + r16 * pitch + ks * 64 + q * 16);
#pragma unroll
                    for (int g = 0; g < 3; ++g) mma_step(a, wreg[g][ks], acc[g], T{});
                }
            }
        }
        float hnew[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = sigm(gin[i][0] + (acc[0][i] + bhr));
            const float z = sigm(gin[i][1] + (acc[1][i] + bhz));
            const float hn = acc[2][i] + bhn;
            const float n = tanh_fast(gin[i][2] + r * hn);
            hnew[i] = (1.0f - z) * n + z * hprev[i];
            hprev[i] = hnew[i];
        }
        // ---- publish: one 8-byte granule per (clip, unit pair), written by the even lane of the pair; clips beyond nb publish too
        //      (their rows are copies of clip nb - 1's inputs: consumers read all 16 rows of the slot) ----
        unsigned pair[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mine = round16<T>(hnew[i]);
            const unsigned nbr = (unsigned)__shfl_down((int)mine, 1);
            pair[i] = mine | (nbr << 16);
        }
        if ((r16 & 1) == 0) {
            const int wbase = (step & 1) * slot_bytes;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{pair[i], (unsigned)(step + 1)}, xr, wbase + ((4 * q + i) * gpr + (jcol >> 1)) * 8, 0, 16 /* sc1 */);
            }
        }
#if !(LA_GRU_PROBE & 1)
        if (step + 1 < T_) load_gi(dir == 0 ? t + 1 : t - 1);      // independent of h: in flight during the next poll
#endif
#if !(LA_GRU_PROBE & 2)
        store_out(t, hnew);                                         // nobody's input inside the recurrence: after the hand-off stores
#endif
    }
    if (!alive && tid == 0 && p.timeout_flag) *p.timeout_flag = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Training forward (float32 in / out, gates stored) with the recurrent product on the f16 pipe at float32 accuracy ("f16x2",
// la_f32x2.hip): W_hh rows scaled by powers of two and split into two IEEE-half fragment sets resident in registers (hi + lo =
// 22 bits), h (|h| <= 1, fixed scale 2^14) split the same way by its producer, and per k-step the three products
// h_lo W_hi + h_hi W_lo + h_hi W_hi accumulated in float32 by v_mfma_f32_16x16x32_f16 -- 108 MFMAs of 16 cycles per step and wave
// instead of the 288 float32 MFMAs of 32 cycles of gru_kernel<float> (3.8 us of its 11.4 us step at 16 clips), with that
// kernel's per-wave operand loads replaced by the granule hand-off of gru_granule_kernel: h travels as 8-byte
// {h_hi | h_lo, step tag} granules, polled by the consumers, staged once per workgroup into LDS.
// 4-wave workgroups (one wave per SIMD: 512 registers; the two W fragment sets are 288), 64 hidden units each.
struct GruTrainX2Params {
    const float *gi, *w_hh, *b_hh;
    float *out, *gates;        // gates NULL: inference (float32 la_gru_layer on the f16x2 products), nothing but the layer output is stored
    float *out_mish;           // optional: Mish(out), the head's last layer (module/align_model.py:36-37)
    int B, T, H;
    unsigned long long *xch;   // [groups][2 dirs][2 slots][16 clips][H] granules, zeroed per call
    int *abort_flag, *timeout_flag;
    unsigned long long poll_budget = 300000000ull;
    int fault_step = 0;
};

__device__ __forceinline__ unsigned x2_pack_hi_lo(float x) {          // x (already scaled) -> f16 hi | f16 lo << 16
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

template <int MAXKS>
__global__ __launch_bounds__(256, 1) void gru_train_x2_kernel(GruTrainX2Params p) {
    constexpr int NW = 4;
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int slice = blockIdx.x, dir = blockIdx.y, group = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int H = p.H, T_ = p.T;
    const int nks = H / 32;
    const int jcol = (slice * NW + wave) * 16 + r16;
    const int b0 = group * GROUP;
    const int nb = min(GROUP, p.B - b0);
    const float *Wd = p.w_hh + (int64_t)dir * 3 * H * H;
    const float *bh = p.b_hh + dir * 3 * H;
    const int64_t out_bs = (int64_t)T_ * 2 * H, out_ts = 2 * H;
    const int64_t gi_bs = (int64_t)T_ * 6 * H, gi_ts = 6 * H;
    const int slot_bytes = 16 * H * 8;
    unsigned char *xbase = reinterpret_cast<unsigned char *>(p.xch) + ((int64_t)group * 2 + dir) * 2 * slot_bytes;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xbase, 0, 2 * slot_bytes, 0x00020000);

    // ---- resident W_hh slice as two half-precision fragment sets; row (gate g, unit jcol) scaled to [2^13, 2^14) ----
    uint4 whi[3][MAXKS], wlo[3][MAXKS];
    float wscale[3];                                            // 2^-14 (h's scale) x the row's inverse scale
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float *wr = Wd + (int64_t)(g * H + jcol) * H + q * 8;
        float mx = 0.f;
        for (int ks = 0; ks < nks; ++ks) {
            const float4 a = *reinterpret_cast<const float4 *>(wr + ks * 32), b = *reinterpret_cast<const float4 *>(wr + ks * 32 + 4);
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                 fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sc = 1.f, inv = 1.f;
        if (mx > 0.f && mx < 3.0e38f) {
            int e;
            (void)frexpf(mx, &e);
            int sh = 14 - e;
            sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
            sc = ldexpf(1.f, sh);
            inv = ldexpf(1.f, -sh);
        }
        wscale[g] = inv * 6.103515625e-05f;                     // x 2^-14
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < nks) {
                const float4 a = *reinterpret_cast<const float4 *>(wr + ks * 32), b = *reinterpret_cast<const float4 *>(wr + ks * 32 + 4);
                const unsigned p0 = x2_pack_hi_lo(a.x * sc), p1 = x2_pack_hi_lo(a.y * sc), p2 = x2_pack_hi_lo(a.z * sc), p3 = x2_pack_hi_lo(a.w * sc);
                const unsigned p4 = x2_pack_hi_lo(b.x * sc), p5 = x2_pack_hi_lo(b.y * sc), p6 = x2_pack_hi_lo(b.z * sc), p7 = x2_pack_hi_lo(b.w * sc);
                whi[g][ks] = make_uint4((p0 & 0xffffu) | (p1 << 16), (p2 & 0xffffu) | (p3 << 16), (p4 & 0xffffu) | (p5 << 16), (p6 & 0xffffu) | (p7 << 16));
                wlo[g][ks] = make_uint4((p0 >> 16) | (p1 & 0xffff0000u), (p2 >> 16) | (p3 & 0xffff0000u), (p4 >> 16) | (p5 & 0xffff0000u),
                                        (p6 >> 16) | (p7 & 0xffff0000u));
            }
    }
    int *ok_s = reinterpret_cast<int *>(lds);
    if (tid == 0) *ok_s = 1;
    const int row_bytes = H * 2, pitch = row_bytes + 16;
    // LDS: two step stages x (hi plane, lo plane) of [16 clips][H halves]
    auto stage = [&](int st, int plane) { return lds + 16 + (size_t)((st * 2 + plane) * 16) * pitch; };

    const float bhr = bh[jcol], bhz = bh[H + jcol], bhn = bh[2 * H + jcol];
    float hprev[4] = {0.f, 0.f, 0.f, 0.f};
    float gin[4][3];
    auto load_gi = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bl = min(4 * q + i, nb - 1);
            const float *g = p.gi + (int64_t)(b0 + bl) * gi_bs + (int64_t)t * gi_ts + dir * 3 * H + jcol;
            gin[i][0] = g[0]; gin[i][1] = g[H]; gin[i][2] = g[2 * H];
        }
    };
    load_gi(dir == 0 ? 0 : T_ - 1);
    constexpr int NCH = (16 * MAXKS * 32 / 2 + 255) / 256;      // 16-byte chunks (two granules = two units of one clip) per thread
    const int cpr = H / 2;
    const int nchunks = 16 * cpr;
    __syncthreads();

    bool alive = true;
    for (int step = 0; step < T_; ++step) {
        if (p.fault_step > 0 && step == p.fault_step && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) return;   // test hook: see the parameter
        const int t = dir == 0 ? step : T_ - 1 - step;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            unsigned char *shi = stage(step & 1, 0), *slo = stage(step & 1, 1);
            const unsigned want = (unsigned)step;
            const int sbase = ((step - 1) & 1) * slot_bytes;
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (tid + k * 256 < nchunks) pending |= 1u << k;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            while (pending) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                u32x4 v[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if (pending & (1u << k)) v[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, sbase + (tid + k * 256) * 16, 0, 16 /* sc1 */);
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if ((pending & (1u << k)) && v[k][1] == want && v[k][3] == want) {
                        const int idx = tid + k * 256;
                        const int row = idx / cpr, c = idx - row * cpr;
                        *reinterpret_cast<unsigned *>(shi + row * pitch + c * 4) = (v[k][0] & 0xffffu) | (v[k][2] << 16);
                        *reinterpret_cast<unsigned *>(slo + row * pitch + c * 4) = (v[k][0] >> 16) | (v[k][2] & 0xffff0000u);
                        pending &= ~(1u << k);
                    }
                if (pending && (++spins & 63u) == 0) {
                    if (__hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { *ok_s = 0; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > p.poll_budget) {
                        __hip_atomic_store(p.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *ok_s = 0;
                        break;
                    }
                }
            }
            __syncthreads();
            alive = *ok_s != 0;
            if (!alive) break;
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks) {
                if (ks < nks) {
                    const uint4 ah = *reinterpret_cast<const uint4 *>(shi + r16 * pitch + ks * 64 + q * 16);
                    const uint4 al = *reinterpret_cast<const uint4 *>(slo + r16 * pitch + ks * 64 + q * 16);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {           // small products first
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, whi[g][ks]), acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wlo[g][ks]), acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, whi[g][ks]), acc[g], 0, 0, 0);
                    }
                }
            }
        }
        float hnew[4], gr[4], gz[4], gn[4], ghn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = sigm(gin[i][0] + (acc[0][i] * wscale[0] + bhr));
            const float z = sigm(gin[i][1] + (acc[1][i] * wscale[1] + bhz));
            const float hn = acc[2][i] * wscale[2] + bhn;
            const float n = tanh_fast(gin[i][2] + r * hn);
            hnew[i] = (1.0f - z) * n + z * hprev[i];
            hprev[i] = hnew[i];
            gr[i] = r; gz[i] = z; gn[i] = n; ghn[i] = hn;
        }
        {   // publish: one granule per (clip, unit): {f16 hi | f16 lo of h x 2^14, step + 1}
            const int wbase = (step & 1) * slot_bytes;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{x2_pack_hi_lo(hnew[i] * 16384.0f), (unsigned)(step + 1)}, xr,
                                                      wbase + ((4 * q + i) * H + jcol) * 8, 0, 16 /* sc1 */);
            }
        }
        if (step + 1 < T_) load_gi(dir == 0 ? t + 1 : t - 1);
        // layer output and the saved gates of the backward sweep: plain stores, nobody inside this launch reads them
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bl = 4 * q + i;
            if (bl < nb) {
                const int64_t o = (int64_t)(b0 + bl) * out_bs + (int64_t)t * out_ts + dir * H + jcol;
                p.out[o] = hnew[i];
                if (p.out_mish) p.out_mish[o] = mish_fast(hnew[i]);
                if (p.gates) {
                    float *gp = p.gates + (((int64_t)(b0 + bl) * T_ + t) * 2 + dir) * 4 * H + jcol;
                    gp[0] = gr[i]; gp[H] = gz[i]; gp[2 * H] = gn[i]; gp[3 * H] = ghn[i];
                }
            }
        }
    }
    if (!alive && tid == 0 && p.timeout_flag) *p.timeout_flag = 1;
}

}  // namespace

static int gru_groups(int batch) { return la::cdiv(batch, GROUP); }
static size_t gru_ctr_bytes(int batch, int frames) { return (size_t)la::round_up(16 + (int64_t)gru_groups(batch) * 2 * frames * 4, 256); }
// granule exchange ring behind the counters: [groups][2][2 slots][16][granules per clip row] x 8 bytes
// (sized for the largest user: the backward sweep's reduce-scatter blocks, (hidden / 64)^2 pairs of 16 x 64 granules per slot; the float32
//  training forward uses 16 x hidden granules per slot, the 16-bit inference form half of that)
static size_t gru_xch_fwd_bytes(int batch, int hidden) { return (size_t)gru_groups(batch) * 2 * 2 * 16 * hidden * 8; }
static size_t gru_xch_bwd_bytes(int batch, int hidden) { return (size_t)gru_groups(batch) * 2 * 2 * (hidden / 64) * (hidden / 64) * 16 * 64 * 8; }
static size_t gru_xch_bytes(int batch, int hidden) { return std::max(gru_xch_fwd_bytes(batch, hidden), hidden % 64 == 0 ? gru_xch_bwd_bytes(batch, hidden) : (size_t)0); }

extern "C" int la_gru_workspace_bytes(int32_t batch, int32_t frames, int32_t hidden, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && frames > 0 && hidden > 0, "gru_workspace_bytes: bad arguments");
    // [16 B header: abort flag] + counters [groups][2][frames] u32, padded to 256 B + the granule exchange ring
    *bytes = gru_ctr_bytes(batch, frames) + gru_xch_bytes(batch, hidden);
    return LA_OK;
}

// gates != nullptr: the training forward (float32) also stores r, z, n and W_hn h + b_hn of every step
static int gru_forward(int32_t dtype, const float *gi, const void *w_hh, const float *b_hh, void *out, void *out_mish,
                       int32_t batch, int32_t frames, int32_t hidden, void *workspace, size_t workspace_bytes,
                       int32_t *timeout_flag, float *gates, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || frames == 0) return LA_OK;
    LA_CHECK_ARG(gi && w_hh && b_hh && out && workspace, "gru_layer: null pointer");
    LA_CHECK_ARG(dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16, "gru_layer: bad dtype");
    LA_CHECK_ARG(batch > 0 && frames > 0 && hidden > 0, "gru_layer: bad sizes");
    LA_CHECK_ARG((uintptr_t)w_hh % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)workspace % 16 == 0, "gru_layer: alignment");
    size_t need = 0;
    la_gru_workspace_bytes(batch, frames, hidden, &need);
    LA_CHECK_ARG(workspace_bytes >= need, "gru_layer: workspace too small (%zu < %zu)", workspace_bytes, need);
    const int groups = gru_groups(batch);
    const unsigned long long budget = la::opts().gru_timeout_us > 0 ? (unsigned long long)la::opts().gru_timeout_us * 100ull : 300000000ull;
    const int fault = la::opts().gru_fault_step;              // one-shot test hook: the launch that reads it clears it
    if (fault) la::opts().gru_fault_step = 0;
    if (hidden % 64 != 0 || (dtype != LA_F32 && hidden > 512) || (dtype == LA_F32 && hidden > 384)) {
        la::set_error("gru_layer: hidden=%d unsupported (multiple of 64; bf16 <= 512, f32 <= 384)", hidden);
        return LA_EUNSUPPORTED;
    }
    // 16-bit modes: 8-wave workgroups (128 hidden units each) where hidden allows; LA_GRU_NW=4 keeps the 4-wave form (A/B)
    const bool force_nw4 = la::opts().gru_nw == 4;
    const bool wide = dtype != LA_F32 && hidden % 128 == 0 && hidden <= 384 && !force_nw4;   // (a 192-register W slice spills at 256)
    const int nw = dtype == LA_F32 ? 2 : (wide ? 8 : 4);
    const int nsplit = hidden / (16 * nw);
    if (nsplit * 2 * groups > 224) {
        la::set_error("gru_layer: %d co-resident workgroups needed (batch too large for one launch; split the batch)", nsplit * 2 * groups);
        return LA_EUNSUPPORTED;
    }
    // float32 training forward (gates stored): recurrent product as f16x2 with the granule hand-off (gru_train_x2_kernel); option
    // gru_handoff = 1 keeps gru_kernel<float> (float32 MFMA, counter form): the A/B partner
    // (inference, gates == NULL: the same kernel without the gate stores where option x2_inference is on and more than 4 clips share the launch --
    //  la_align_head_forward's float32 route; a handful of clips keeps gru_kernel<float>, whose v_fma path is faster there)
    if (dtype == LA_F32 && (gates || (la::opts().x2_inference && batch > 4)) && hidden % 64 == 0 && hidden <= 384 && la::opts().gru_handoff == 0 &&
        la::opts().gru_fence == 0 && (hidden / 64) * 2 * groups <= 224) {
        unsigned char *wsb = reinterpret_cast<unsigned char *>(workspace);
        const size_t ctrb = gru_ctr_bytes(batch, frames);
        LA_HIP(hipMemsetAsync(wsb, 0, 16, stream));
        LA_HIP(hipMemsetAsync(wsb + ctrb, 0, gru_xch_fwd_bytes(batch, hidden), stream));
        GruTrainX2Params tp{gi, reinterpret_cast<const float *>(w_hh), b_hh, reinterpret_cast<float *>(out), gates, reinterpret_cast<float *>(out_mish), batch, frames, hidden,
                            reinterpret_cast<unsigned long long *>(wsb + ctrb), reinterpret_cast<int *>(workspace), timeout_flag, budget, fault};
        const size_t lds_t = 16 + (size_t)2 * 2 * 16 * (hidden * 2 + 16);
        la::TimerScope ts("gru_f32", stream);
        hipLaunchKernelGGL((gru_train_x2_kernel<12>), dim3(hidden / 64, 2, groups), dim3(256), lds_t, stream, tp);
        LA_LAUNCH_CHECK();
        return LA_OK;
    }
    // 16-bit recurrence in 8-wave workgroups: the granule hand-off (option gru_handoff = 0, default) or the counter form (1)
    // (measured, tools/kbench.py gru: 3.28 against 3.38 us per step at 32 - 64 clips, level at 16, 2.86 against 2.64 at one clip -- the counter
    //  form keeps the single 16-clip group; gru_handoff = 2 forces granules for every batch)
    const bool granules = wide && !gates && la::opts().gru_fence == 0 && hidden % 128 == 0 &&
                          (la::opts().gru_handoff == 2 || (la::opts().gru_handoff == 0 && batch > 16));
    if (granules) {
        LA_CHECK_ARG(frames < 0x7fffffff, "gru_layer: too many frames");
        unsigned char *wsb = reinterpret_cast<unsigned char *>(workspace);
        const size_t ctrb = gru_ctr_bytes(batch, frames);
        LA_HIP(hipMemsetAsync(wsb, 0, 16, stream));
        LA_HIP(hipMemsetAsync(wsb + ctrb, 0, gru_xch_fwd_bytes(batch, hidden), stream));
        GruGranuleParams gp{gi, w_hh, b_hh, out, out_mish, batch, frames, hidden, reinterpret_cast<unsigned long long *>(wsb + ctrb),
                            reinterpret_cast<int *>(workspace), timeout_flag, nsplit, la::opts().gru_poll_delay, budget, fault};
        const size_t lds_g = 16 + (size_t)2 * 16 * (hidden * 2 + 16);          // flag + two h stages
        la::TimerScope ts("gru_bf16", stream);
        if (dtype == LA_F16) hipLaunchKernelGGL((gru_granule_kernel<la::f16_t, 12>), dim3(nsplit, 2, groups), dim3(512), lds_g, stream, gp);
        else hipLaunchKernelGGL((gru_granule_kernel<bf16_t, 12>), dim3(nsplit, 2, groups), dim3(512), lds_g, stream, gp);
        LA_LAUNCH_CHECK();
        return LA_OK;
    }
    LA_HIP(hipMemsetAsync(workspace, 0, gru_ctr_bytes(batch, frames), stream));
    GruParams p{gi, w_hh, b_hh, out, out_mish, batch, frames, hidden,
                reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(workspace) + 16),
                reinterpret_cast<int *>(workspace), timeout_flag, nsplit, gates, budget, fault};
    const dim3 grid(nsplit, 2, groups);
    // Hand-off forms (tools/kbench.py gru, 32 clips, T=1500, H=384; tools/handoff_bench.hip for the bare protocol costs):
    // write-through (sc1 stores, drained; relaxed counter; sc1 loads) 5.4 ms per layer, release / acquire fences 9.0 ms.
    // The write-through form is the default; LA_GRU_FENCE=1 selects the fence form.
    const bool use_fence = la::opts().gru_fence != 0;
    LA_CHECK_ARG((int64_t)batch * frames * 2 * hidden * (dtype == LA_F32 ? 4 : 2) < (int64_t)2147483647, "gru_layer: out buffer exceeds the 2 GiB buffer-descriptor range");
    if (dtype == LA_BF16 || dtype == LA_F16) {
        la::TimerScope ts("gru_bf16", stream);
        const int mt = (batch <= 16 || GROUP == 16) ? 1 : 2;
        const size_t lds_b = 16 + (size_t)16 * mt * (hidden * 2 + 16);          // flag + the staged h rows of the batch tiles
        // MAXKS = the W slice's compile-time k-steps: 12 with 8 waves (hidden <= 384, 144 registers), 16 with 4 (hidden <= 512, 192)
#define LA_GRU_LAUNCH16_(T_, KS_, NW_)                                                                                     \
    do {                                                                                                                   \
        if (mt == 1) {                                                                                                     \
            if (use_fence) hipLaunchKernelGGL((gru_kernel<T_, KS_, false, 1, NW_>), grid, dim3(NW_ * 64), lds_b, stream, p); \
            else hipLaunchKernelGGL((gru_kernel<T_, KS_, true, 1, NW_>), grid, dim3(NW_ * 64), lds_b, stream, p);          \
        } else {                                                                                                           \
            if (use_fence) hipLaunchKernelGGL((gru_kernel<T_, KS_, false, 2, NW_>), grid, dim3(NW_ * 64), lds_b, stream, p); \
            else hipLaunchKernelGGL((gru_kernel<T_, KS_, true, 2, NW_>), grid, dim3(NW_ * 64), lds_b, stream, p);          \
        }                                                                                                                  \
    } while (0)
#define LA_GRU_LAUNCH16(T_)                                                                                                \
    do {                                                                                                                   \
        if (wide) LA_GRU_LAUNCH16_(T_, 12, 8);                                                                             \
        else LA_GRU_LAUNCH16_(T_, 16, 4);                                                                                  \
    } while (0)
        if (dtype == LA_F16) LA_GRU_LAUNCH16(la::f16_t); else LA_GRU_LAUNCH16(bf16_t);
#undef LA_GRU_LAUNCH16
#undef LA_GRU_LAUNCH16_
    } else {
        const size_t lds_bytes = 16 + (size_t)2 * 3 * 16 * hidden * 4;
        static la::DeviceOnce attr_once;
        if (attr_once.pending()) {
            const int max_lds = 16 + 2 * 3 * 16 * 384 * 4;
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_kernel<float, 24, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_kernel<float, 24, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_kernel<float, 24, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_kernel<float, 24, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            attr_once.mark();
        }
        la::TimerScope ts("gru_f32", stream);
        if (batch <= 16 || GROUP == 16) {
            if (use_fence) hipLaunchKernelGGL((gru_kernel<float, 24, false, 1>), grid, dim3(128), lds_bytes, stream, p);
            else hipLaunchKernelGGL((gru_kernel<float, 24, true, 1>), grid, dim3(128), lds_bytes, stream, p);
        } else {
            if (use_fence) hipLaunchKernelGGL((gru_kernel<float, 24, false, 2>), grid, dim3(128), lds_bytes, stream, p);
            else hipLaunchKernelGGL((gru_kernel<float, 24, true, 2>), grid, dim3(128), lds_bytes, stream, p);
        }
    }
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_gru_layer(int32_t dtype, const float *gi, const void *w_hh, const float *b_hh, void *out, void *out_mish,
                            int32_t batch, int32_t frames, int32_t hidden, void *workspace, size_t workspace_bytes,
                            int32_t *timeout_flag, void *stream) {
    return gru_forward(dtype, gi, w_hh, b_hh, out, out_mish, batch, frames, hidden, workspace, workspace_bytes, timeout_flag, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// training face (fine-tune row, float32 like the reference): forward that also stores the gates, and the backward sweep
// ---------------------------------------------------------------------------------------------------------------
extern "C" int la_gru_layer_train_fwd(const float *gi, const float *w_hh, const float *b_hh, float *out, float *gates,
                                      int32_t batch, int32_t frames, int32_t hidden, void *workspace, size_t workspace_bytes,
                                      int32_t *timeout_flag, void *stream) {
    LA_CHECK_ARG(gates, "gru_layer_train_fwd: gates buffer missing");
    return gru_forward(LA_F32, gi, w_hh, b_hh, out, nullptr, batch, frames, hidden, workspace, workspace_bytes, timeout_flag, gates, stream);
}

namespace {

struct GruBwdParams {
    const float *gates;  // [B][T][2][4H]
    const float *out;    // [B][T][2H]  h_t of the forward pass
    const float *dout;   // [B][T][2H]  gradient w.r.t. the layer output
    const float *w_hh;   // [2][3H][H]
    float *dgi;          // [B][T][2][3H]  gradient w.r.t. the input projections (W_ih x + b_ih)
    float *dgh;          // [B][T][2][3H]  gradient w.r.t. the recurrent projections (W_hh h + b_hh); also the exchange buffer
    int B, T, H;
    unsigned *counters;
    int *abort_flag;
    int *timeout_flag;
    int nsplit;
    unsigned long long poll_budget = 300000000ull;
};

// Backward recurrence dh_{t-1} = dh_t * z_t + dgh_t W_hh: the contraction runs over all 3H gate units, so (as in the
// forward kernel) the H hidden units are split over workgroups, each wave keeps its 16 columns of W_hh (as W_hh^T rows,
// [16][3H] f32 = 72 KiB) resident in LDS, and dgh of the previous step is exchanged through HBM with the same
// write-through (sc1) hand-off.  2 waves per workgroup, H/32 workgroups per direction.
template <int MT>
__global__ __launch_bounds__(128, 1) void gru_bwd_kernel(GruBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NW = 2;
    const int slice = blockIdx.x, dir = blockIdx.y, group = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int H = p.H, T_ = p.T, K3 = 3 * p.H;
    const int nks = K3 / 16;
    const int kcol = (slice * NW + wave) * 16 + r16;   // this lane's hidden unit
    const int b0 = group * GROUP;
    const int nb = min(GROUP, p.B - b0);
    const float *Wd = p.w_hh + (int64_t)dir * K3 * H;
    const int row_bytes = K3 * 4;
    const int slots = row_bytes / 16;
    // W_hh^T image: [wave][16 columns][3H], 16-B slot s of column n stored at s ^ n (low 4 bits)
    for (int idx = tid; idx < NW * 16 * slots; idx += NW * 64) {
        const int s4 = idx % slots, n = (idx / slots) % 16, w = idx / (slots * 16);
        const int k = (slice * NW + w) * 16 + n;
        float4 v;
        v.x = Wd[(int64_t)(s4 * 4 + 0) * H + k]; v.y = Wd[(int64_t)(s4 * 4 + 1) * H + k];
        v.z = Wd[(int64_t)(s4 * 4 + 2) * H + k]; v.w = Wd[(int64_t)(s4 * 4 + 3) * H + k];
        *reinterpret_cast<float4 *>(lds + 16 + ((int64_t)(w * 16 + n)) * row_bytes + ((s4 ^ n) << 4)) = v;
    }
    __syncthreads();
    int *ok_s = reinterpret_cast<int *>(lds);
    const unsigned char *wl = lds + 16 + (int64_t)(wave * 16) * row_bytes;
    unsigned *ctr = p.counters + ((int64_t)group * 2 + dir) * T_;
    const int64_t g4 = 4 * (int64_t)H, g3 = 3 * (int64_t)H;
    // write-through hand-off of dgh (as in the forward kernel): sc1 stores, drained, relaxed counter, sc1 loads -- no fences
    const __amdgpu_buffer_rsrc_t dgh_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.dgh, 0, (int)((int64_t)p.B * T_ * 2 * g3 * 4), 0x00020000);

    float carry[MT][4];   // dh_t * z_t of the step processed before (the direct path of the recurrence)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) carry[mt][i] = 0.f;
    int arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = b0 + min(mt * 16 + r16, nb - 1);

    bool alive = true;
    for (int step = 0; step < T_; ++step) {
        const int t = dir == 0 ? T_ - 1 - step : step;          // reverse of the forward scan order
        const int tnext = dir == 0 ? t + 1 : t - 1;              // processed in the previous iteration
        const int tprev = dir == 0 ? t - 1 : t + 1;              // the forward pass's previous step (h_{t-1})
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // this step's saved gates, h_{t-1} and upstream gradient do not depend on the hand-off: requested before the wait
        float g_r[MT][4], g_z[MT][4], g_n[MT][4], g_hn[MT][4], g_hp[MT][4], g_do[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + min(mt * 16 + 4 * q + i, nb - 1);
                const float *gp = p.gates + (((int64_t)b * T_ + t) * 2 + dir) * g4 + kcol;
                g_r[mt][i] = gp[0]; g_z[mt][i] = gp[H]; g_n[mt][i] = gp[2 * H]; g_hn[mt][i] = gp[3 * H];
                g_hp[mt][i] = (tprev >= 0 && tprev < T_) ? p.out[((int64_t)b * T_ + tprev) * 2 * H + dir * H + kcol] : 0.f;
                g_do[mt][i] = p.dout[((int64_t)b * T_ + t) * 2 * H + dir * H + kcol];
            }
        if (step > 0) {
            if (tid == 0) {
                const bool ok = wait_counter(ctr + (step - 1), (unsigned)p.nsplit, p.abort_flag, p.poll_budget);
                *ok_s = ok ? 1 : 0;
            }
            __syncthreads();
            alive = *ok_s != 0;
            if (!alive) break;
            // dgh of the step before, 12 k-steps (12 x MT fragments) requested at a time: with one fragment per iteration the
            // loop was 3H/16 = 72 dependent L2 round trips per step (the whole 22.7 us of it).  3H/16 = 12 (H/64).
            int arow_off[MT];            // byte offsets into dgh
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) arow_off[mt] = (int)(((((int64_t)arow[mt] * T_ + tnext) * 2 + dir) * g3) * 4) + q * 16;
            if (MT == 1 && nb <= 4) {
                // <= 4 clips: v_fma on the lane's 4-k slices instead of 16-row f32 MFMAs (see the forward kernel)
                auto small = [&](auto nbc) {
                    constexpr int NBV = decltype(nbc)::value;
                    float part[NBV];
#pragma unroll
                    for (int b = 0; b < NBV; ++b) part[b] = 0.f;
                    int boff[NBV];
#pragma unroll
                    for (int b = 0; b < NBV; ++b) boff[b] = (int)(((((int64_t)(b0 + b) * T_ + tnext) * 2 + dir) * g3) * 4) + q * 16;
                    for (int kb = 0; kb < nks; kb += 12) {
                        uint4 a[12][NBV];
#pragma unroll
                        for (int u = 0; u < 12; ++u)
#pragma unroll
                            for (int b = 0; b < NBV; ++b) {
                                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                                const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(dgh_rsrc, boff[b] + (kb + u) * 64, 0, 16 /* sc1 */);
                                a[u][b] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                            }
#pragma unroll
                        for (int u = 0; u < 12; ++u) {
                            const int ks = kb + u;
                            const uint4 w = *reinterpret_cast<const uint4 *>(wl + (int64_t)r16 * row_bytes + (((ks * 4 + q) ^ r16) << 4));
#pragma unroll
                            for (int b = 0; b < NBV; ++b) {
                                float s = part[b];
                                s = fmaf(__uint_as_float(w.x), __uint_as_float(a[u][b].x), s);
                                s = fmaf(__uint_as_float(w.y), __uint_as_float(a[u][b].y), s);
                                s = fmaf(__uint_as_float(w.z), __uint_as_float(a[u][b].z), s);
                                s = fmaf(__uint_as_float(w.w), __uint_as_float(a[u][b].w), s);
                                part[b] = s;
                            }
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NBV; ++b) {
                        float v = part[b];
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
                        acc[0][b] = v;
                    }
                };
                if (nb == 1) small(std::integral_constant<int, 1>{});
                else if (nb == 2) small(std::integral_constant<int, 2>{});
                else if (nb == 3) small(std::integral_constant<int, 3>{});
                else small(std::integral_constant<int, 4>{});
            } else
            for (int kb = 0; kb < nks; kb += 12) {
                uint4 a[12][MT];
#pragma unroll
                for (int u = 0; u < 12; ++u)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                        const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(dgh_rsrc, arow_off[mt] + (kb + u) * 64, 0, 16 /* sc1 */);
                        a[u][mt] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                    }
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int ks = kb + u;
                    const uint4 w = *reinterpret_cast<const uint4 *>(wl + (int64_t)r16 * row_bytes + (((ks * 4 + q) ^ r16) << 4));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma_step(a[u][mt], w, acc[mt], 0.0f);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bl = mt * 16 + 4 * q + i;
                const int b = b0 + min(bl, nb - 1);
                const float r = g_r[mt][i], z = g_z[mt][i], n = g_n[mt][i], hn = g_hn[mt][i], hp = g_hp[mt][i];
                const float dh = g_do[mt][i] + acc[mt][i] + carry[mt][i];
                const float dn_pre = dh * (1.0f - z) * (1.0f - n * n);
                const float dz_pre = dh * (hp - n) * z * (1.0f - z);
                const float dr_pre = dn_pre * hn * r * (1.0f - r);
                carry[mt][i] = dh * z;
                if (bl < nb) {
                    float *o1 = p.dgi + (((int64_t)b * T_ + t) * 2 + dir) * g3 + kcol;
                    const int o2 = (int)(((((int64_t)b * T_ + t) * 2 + dir) * g3 + kcol) * 4);
                    o1[0] = dr_pre; o1[H] = dz_pre; o1[2 * H] = dn_pre;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dr_pre), dgh_rsrc, o2, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz_pre), dgh_rsrc, o2 + H * 4, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dn_pre * r), dgh_rsrc, o2 + 2 * H * 4, 0, 16);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr + step, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!alive && tid == 0 && p.timeout_flag) *p.timeout_flag = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward sweep with the recurrent product on the f16 pipe at float32 accuracy and a REDUCE-SCATTER hand-off.
// gru_bwd_kernel above is an all-gather: every workgroup fetches the whole dgh of the step before (16 clips x 3H floats: 74 KB
// at H = 384, per wave) and multiplies it with its columns of W_hh on the float32 matrix pipe (288 MFMAs of 32 cycles): 15.5 us
// per step at 16 clips.  Here a workgroup owns 64 hidden units on BOTH sides of the product
//     dh[b][k] = sum_j dgh[b][j] W_hh[j][k],   j over the 3H gate columns:
// it multiplies ITS OWN 192 columns of dgh (its units' dr, dz, dn r: produced in its own registers, never exchanged) with the
// matching 192 rows of W_hh for ALL H output units k -- [16 x 192] x [192 x H], K = 192 -- and hands each partial sum to the
// workgroup that owns unit k: 16 x 64 floats per pair of workgroups and step (40 KB received at H = 384 instead of 147 KB
// fetched), as 8-byte {float, step tag} granules polled by the receiver (Guideline 16, R2: no counter, no drain).  The owner adds
// the partials in workgroup order (deterministic).  The product runs as f16x2 (la_f32x2.hip): the W rows of a workgroup, scaled per
// output column by powers of two, sit in registers as two half-precision fragment sets; the workgroup's dgh block is scaled per
// clip (its largest magnitude over the 192 local columns: known locally) and split into two half planes in LDS; three MFMA
// 16x16x32 f16 products per k-step and 16-column tile, float32 accumulate: 108 MFMAs of 16 cycles per wave and step.
struct GruBwdX2Params {
    const float *gates, *out, *dout, *w_hh;
    float *dgi, *dgh;
    int B, T, H;
    unsigned long long *xch;   // [groups][2 dirs][2 slots][NS dest][NS src][16 clips][64 units] granules, zeroed per call
    int *abort_flag, *timeout_flag;
    unsigned long long poll_budget = 300000000ull;
};

template <int NS>                                               // NS = H / 64: workgroups per (group, direction) = column tiles per wave
__global__ __launch_bounds__(256, 1) void gru_bwd_x2_kernel(GruBwdX2Params p) {
    constexpr int NW = 4, MAXT = NS;                            // 4 waves x 16 units
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int s_wg = blockIdx.x, dir = blockIdx.y, group = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int H = p.H, T_ = p.T;
    const int ul = wave * 16 + r16;                             // this lane's unit inside the workgroup's 64
    const int kcol = s_wg * 64 + ul;                            // ... and in the layer
    const int b0 = group * GROUP;
    const int nb = min(GROUP, p.B - b0);
    const float *Wd = p.w_hh + (int64_t)dir * 3 * H * H;
    const int64_t g4 = 4 * (int64_t)H, g3 = 3 * (int64_t)H;
    const int pair_bytes = 16 * 64 * 8;                         // one (dest, src) block of a slot
    const int slot_bytes = NS * NS * pair_bytes;
    unsigned char *xbase = reinterpret_cast<unsigned char *>(p.xch) + ((int64_t)group * 2 + dir) * 2 * slot_bytes;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xbase, 0, 2 * slot_bytes, 0x00020000);

    // LDS: [16 B flag][cmax: 4 waves x 16 clips f32][A_hi, A_lo: 16 clips x (192 halves + 8 pad)][part: NS x 16 clips x 64 units f32]
    int *ok_s = reinterpret_cast<int *>(lds);
    float *cmax = reinterpret_cast<float *>(lds + 16);
    constexpr int APITCH = 192 * 2 + 16;
    unsigned char *a_hi = lds + 16 + 4 * 16 * 4, *a_lo = a_hi + 16 * APITCH;
    float *part = reinterpret_cast<float *>(a_lo + 16 * APITCH);
    if (tid == 0) *ok_s = 1;

    // ---- resident rows of W_hh: local reduction index j = g * 64 + u  <->  row g * H + 64 s_wg + u; B fragments of tile tau hold
    //      column k = (wave * NS + tau) * 16 + r16, eight consecutive j per lane; per-column power-of-two scale ----
    uint4 whi[MAXT][6], wlo[MAXT][6];
    float wscale[MAXT];
#pragma unroll
    for (int tau = 0; tau < MAXT; ++tau) {
        wscale[tau] = 1.f;
        if (tau < NS) {
            const int k = (wave * NS + tau) * 16 + r16;
            auto wj = [&](int j) { return Wd[(int64_t)((j >> 6) * H + s_wg * 64 + (j & 63)) * H + k]; };
            float mx = 0.f;
            for (int ks = 0; ks < 6; ++ks)
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(wj(ks * 32 + q * 8 + e)));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sc = 1.f, inv = 1.f;
            if (mx > 0.f && mx < 3.0e38f) {
                int e2;
                (void)frexpf(mx, &e2);
                int sh = 14 - e2;
                sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
                sc = ldexpf(1.f, sh);
                inv = ldexpf(1.f, -sh);
            }
            wscale[tau] = inv;
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = x2_pack_hi_lo(wj(ks * 32 + q * 8 + e) * sc);
                whi[tau][ks] = make_uint4((pk[0] & 0xffffu) | (pk[1] << 16), (pk[2] & 0xffffu) | (pk[3] << 16), (pk[4] & 0xffffu) | (pk[5] << 16),
                                          (pk[6] & 0xffffu) | (pk[7] << 16));
                wlo[tau][ks] = make_uint4((pk[0] >> 16) | (pk[1] & 0xffff0000u), (pk[2] >> 16) | (pk[3] & 0xffff0000u),
                                          (pk[4] >> 16) | (pk[5] & 0xffff0000u), (pk[6] >> 16) | (pk[7] & 0xffff0000u));
            }
        }
    }
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    const int nrecv = (NS - 1) * 512;                           // 16-byte chunks (two granules) to receive per step
    __syncthreads();

    bool alive = true;
    for (int step = 0; step < T_; ++step) {
        const int t = dir == 0 ? T_ - 1 - step : step;
        const int tprev = dir == 0 ? t - 1 : t + 1;
        float g_r[4], g_z[4], g_n[4], g_hn[4], g_hp[4], g_do[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + min(4 * q + i, nb - 1);
            const float *gp = p.gates + (((int64_t)b * T_ + t) * 2 + dir) * g4 + kcol;
            g_r[i] = gp[0]; g_z[i] = gp[H]; g_n[i] = gp[2 * H]; g_hn[i] = gp[3 * H];
            g_hp[i] = (tprev >= 0 && tprev < T_) ? p.out[((int64_t)b * T_ + tprev) * 2 * H + dir * H + kcol] : 0.f;
            g_do[i] = p.dout[((int64_t)b * T_ + t) * 2 * H + dir * H + kcol];
        }
        float rec[4] = {0.f, 0.f, 0.f, 0.f};                     // sum_j dgh[step - 1][clip][j] W[j][kcol]
        if (step > 0) {
            // ---- the other workgroups' partial sums for this workgroup's 64 units: poll until every granule carries the step's tag ----
            const unsigned want = (unsigned)step;
            const int rbase = ((step - 1) & 1) * slot_bytes + s_wg * NS * pair_bytes;        // dest = this workgroup
            constexpr int NCH = NS > 1 ? 2 * (NS - 1) : 1;
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (tid + k * 256 < nrecv) pending |= 1u << k;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            while (pending) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                u32x4 v[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if (pending & (1u << k)) {
                        const int c = tid + k * 256;
                        int src = c >> 9;
                        src += src >= s_wg ? 1 : 0;              // (own block skipped)
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, rbase + src * pair_bytes + (c & 511) * 16, 0, 16 /* sc1 */);
                    }
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if ((pending & (1u << k)) && v[k][1] == want && v[k][3] == want) {
                        const int c = tid + k * 256;
                        int src = c >> 9;
                        src += src >= s_wg ? 1 : 0;
                        *reinterpret_cast<float2 *>(part + src * 1024 + (c & 511) * 2) = make_float2(__uint_as_float(v[k][0]), __uint_as_float(v[k][2]));
                        pending &= ~(1u << k);
                    }
                if (pending && (++spins & 63u) == 0) {
                    if (__hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { *ok_s = 0; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > p.poll_budget) {
                        __hip_atomic_store(p.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *ok_s = 0;
                        break;
                    }
                }
            }
            __syncthreads();
            alive = *ok_s != 0;
            if (!alive) break;
            for (int src = 0; src < NS; ++src)                   // fixed order: deterministic sums
#pragma unroll
                for (int i = 0; i < 4; ++i) rec[i] += part[src * 1024 + (4 * q + i) * 64 + ul];
        }
        // ---- gate gradients of this step (lane-local) ----
        float x[4][3];
        float cm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bl = 4 * q + i;
            const int b = b0 + min(bl, nb - 1);
            const float r = g_r[i], z = g_z[i], n = g_n[i], hn = g_hn[i], hp = g_hp[i];
            const float dh = g_do[i] + rec[i] + carry[i];
            const float dn_pre = dh * (1.0f - z) * (1.0f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.0f - z);
            const float dr_pre = dn_pre * hn * r * (1.0f - r);
            carry[i] = dh * z;
            x[i][0] = dr_pre; x[i][1] = dz_pre; x[i][2] = dn_pre * r;
            cm[i] = fmaxf(fmaxf(fabsf(x[i][0]), fabsf(x[i][1])), fabsf(x[i][2]));
            if (bl < nb) {
                const int64_t o = (((int64_t)b * T_ + t) * 2 + dir) * g3 + kcol;
                p.dgi[o] = dr_pre; p.dgi[o + H] = dz_pre; p.dgi[o + 2 * H] = dn_pre;
                p.dgh[o] = x[i][0]; p.dgh[o + H] = x[i][1]; p.dgh[o + 2 * H] = x[i][2];
            }
        }
        if (step + 1 == T_) break;                               // nobody consumes the last step's products
        // ---- per-clip scale of the workgroup's 16 x 192 block: max over the 16 units of the wave (DPP row), then over the 4 waves ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float m = cm[i];
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
            if (r16 == 0) cmax[wave * 16 + 4 * q + i] = m;
        }
        __syncthreads();
        float inv_c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i;
            const float m = fmaxf(fmaxf(cmax[c], cmax[16 + c]), fmaxf(cmax[32 + c], cmax[48 + c]));
            float sc = 1.f, inv = 1.f;
            if (m > 0.f && m < 3.0e38f) {
                int e2;
                (void)frexpf(m, &e2);
                int sh = 14 - e2;
                sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
                sc = ldexpf(1.f, sh);
                inv = ldexpf(1.f, -sh);
            }
            inv_c[i] = inv;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const unsigned pk = x2_pack_hi_lo(x[i][g] * sc);
                *reinterpret_cast<unsigned short *>(a_hi + c * APITCH + (g * 64 + ul) * 2) = (unsigned short)(pk & 0xffffu);
                *reinterpret_cast<unsigned short *>(a_lo + c * APITCH + (g * 64 + ul) * 2) = (unsigned short)(pk >> 16);
            }
        }
        __syncthreads();
        // ---- partial dh for all H units from the local 192 columns: three f16 products per k-step and tile ----
        f32x4 acc[MAXT];
#pragma unroll
        for (int tau = 0; tau < MAXT; ++tau) acc[tau] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const uint4 ah = *reinterpret_cast<const uint4 *>(a_hi + r16 * APITCH + ks * 64 + q * 16);
            const uint4 al = *reinterpret_cast<const uint4 *>(a_lo + r16 * APITCH + ks * 64 + q * 16);
#pragma unroll
            for (int tau = 0; tau < MAXT; ++tau)
                if (tau < NS) {
                    acc[tau] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, whi[tau][ks]), acc[tau], 0, 0, 0);
                    acc[tau] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wlo[tau][ks]), acc[tau], 0, 0, 0);
                    acc[tau] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, whi[tau][ks]), acc[tau], 0, 0, 0);
                }
        }
        // ---- hand every partial to the owner of its unit: own units through LDS, the others as {float, tag} granules ----
        const int wbase = (step & 1) * slot_bytes;
#pragma unroll
        for (int tau = 0; tau < MAXT; ++tau)
            if (tau < NS) {
                const int k0 = (wave * NS + tau) * 16;
                const int dest = k0 >> 6, u = (k0 & 63) + r16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[tau][i] * (inv_c[i] * wscale[tau]);
                    if (dest == s_wg) {
                        part[s_wg * 1024 + (4 * q + i) * 64 + u] = v;
                    } else {
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v), (unsigned)(step + 1)}, xr,
                                                              wbase + (dest * NS + s_wg) * pair_bytes + ((4 * q + i) * 64 + u) * 8, 0, 16 /* sc1 */);
                    }
                }
            }
    }
    if (!alive && tid == 0 && p.timeout_flag) *p.timeout_flag = 1;
}

}  // namespace

extern "C" int la_gru_layer_bwd(const float *gates, const float *out, const float *dout, const float *w_hh, float *dgi,
                                float *dgh, int32_t batch, int32_t frames, int32_t hidden, void *workspace,
                                size_t workspace_bytes, int32_t *timeout_flag, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || frames == 0) return LA_OK;
    LA_CHECK_ARG(gates && out && dout && w_hh && dgi && dgh && workspace, "gru_layer_bwd: null pointer");
    LA_CHECK_ARG(batch > 0 && frames > 0 && hidden > 0 && hidden % 64 == 0 && hidden <= 384, "gru_layer_bwd: hidden must be a multiple of 64, <= 384");
    size_t need = 0;
    la_gru_workspace_bytes(batch, frames, hidden, &need);
    LA_CHECK_ARG(workspace_bytes >= need && (uintptr_t)workspace % 16 == 0, "gru_layer_bwd: workspace too small");
    const int groups = gru_groups(batch);
    // default: the f16x2 product with the reduce-scatter granule hand-off (gru_bwd_x2_kernel); option gru_handoff = 1 keeps the
    // float32-MFMA all-gather kernel below (the A/B partner).  <= 4 clips stay on it too: its v_fma path beats a 16-row MFMA tile there.
    if (la::opts().gru_handoff == 0 && batch > 4 && (hidden / 64) * 2 * groups <= 224) {
        unsigned char *wsb = reinterpret_cast<unsigned char *>(workspace);
        const size_t ctrb = gru_ctr_bytes(batch, frames);
        LA_HIP(hipMemsetAsync(wsb, 0, 16, stream));
        LA_HIP(hipMemsetAsync(wsb + ctrb, 0, gru_xch_bwd_bytes(batch, hidden), stream));
        GruBwdX2Params xp{gates, out, dout, w_hh, dgi, dgh, batch, frames, hidden, reinterpret_cast<unsigned long long *>(wsb + ctrb),
                          reinterpret_cast<int *>(workspace), timeout_flag,
                          la::opts().gru_timeout_us > 0 ? (unsigned long long)la::opts().gru_timeout_us * 100ull : 300000000ull};
        const size_t lds_x = 16 + 4 * 16 * 4 + (size_t)2 * 16 * (192 * 2 + 16) + (size_t)(hidden / 64) * 1024 * 4;
        la::TimerScope ts("gru_bwd_f32", stream);
        switch (hidden / 64) {
            case 1: hipLaunchKernelGGL(gru_bwd_x2_kernel<1>, dim3(1, 2, groups), dim3(256), lds_x, stream, xp); break;
            case 2: hipLaunchKernelGGL(gru_bwd_x2_kernel<2>, dim3(2, 2, groups), dim3(256), lds_x, stream, xp); break;
            case 3: hipLaunchKernelGGL(gru_bwd_x2_kernel<3>, dim3(3, 2, groups), dim3(256), lds_x, stream, xp); break;
            case 4: hipLaunchKernelGGL(gru_bwd_x2_kernel<4>, dim3(4, 2, groups), dim3(256), lds_x, stream, xp); break;
            case 5: hipLaunchKernelGGL(gru_bwd_x2_kernel<5>, dim3(5, 2, groups), dim3(256), lds_x, stream, xp); break;
            default: hipLaunchKernelGGL(gru_bwd_x2_kernel<6>, dim3(6, 2, groups), dim3(256), lds_x, stream, xp); break;
        }
        LA_LAUNCH_CHECK();
        return LA_OK;
    }
    const int nsplit = hidden / 32;
    LA_CHECK_ARG(nsplit * 2 * groups <= 224, "gru_layer_bwd: batch too large for one co-resident launch");
    LA_CHECK_ARG((int64_t)batch * frames * 2 * 3 * hidden * 4 < (int64_t)2147483647, "gru_layer_bwd: dgh exceeds the 2 GiB buffer-descriptor range");
    LA_HIP(hipMemsetAsync(workspace, 0, need, stream));
    GruBwdParams p{gates, out, dout, w_hh, dgi, dgh, batch, frames, hidden,
                   reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(workspace) + 16),
                   reinterpret_cast<int *>(workspace), timeout_flag, nsplit,
                   la::opts().gru_timeout_us > 0 ? (unsigned long long)la::opts().gru_timeout_us * 100ull : 300000000ull};
    const size_t lds_bytes = 16 + (size_t)2 * 16 * 3 * hidden * 4;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16 + 2 * 16 * 3 * 384 * 4));
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gru_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16 + 2 * 16 * 3 * 384 * 4));
        attr_once.mark();
    }
    la::TimerScope ts("gru_bwd_f32", stream);
    if (batch <= 16 || GROUP == 16) hipLaunchKernelGGL(gru_bwd_kernel<1>, dim3(nsplit, 2, groups), dim3(128), lds_bytes, stream, p);
    else hipLaunchKernelGGL(gru_bwd_kernel<2>, dim3(nsplit, 2, groups), dim3(128), lds_bytes, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
