// la_common.h -- shared host/device helpers for liblyricalign_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lyricalign.h"

namespace la {

// ---- thread-local error text (la_last_error) -------------------------------
char *err_buf();
void set_error(const char *fmt, ...);

// library-owned scratch for one (device, stream, purpose); nullptr on allocation failure (la_runtime.cpp)
enum { SCRATCH_SPLITK = 0, SCRATCH_COLSUM = 1, SCRATCH_X2 = 2 };
void *stream_scratch(hipStream_t stream, int purpose, size_t bytes);

#define LA_CHECK_ARG(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            la::set_error(__VA_ARGS__);        \
            return LA_EINVAL;                  \
        }                                      \
    } while (0)

#define LA_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            la::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return LA_EHIP;                                                              \
        }                                                                                \
    } while (0)

#define LA_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            la::set_error("%s:%d launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return LA_EHIP;                                                             \
        }                                                                               \
    } while (0)

// ---- library options (include/lyricalign.h la_set_option): resolved once per process from the environment, read by the launch paths
struct Options {
    int gemm_tile = 0, gemm_loop = 0, gemm_splitk = 1, attn_nw = 0, gru_nw = 0, gru_fence = 0, gru_handoff = 0, gru_poll_delay = 0, viterbi_dpp = 1,
        head_clip_cap = 0, ln_fusion = 1, resid_split = 1, x2_inference = 1, gru_timeout_us = 0, gru_fault_step = 0;
};
Options &opts();
// Developer switches of the experiment build (-DLA_EXPERIMENTS, tools/build_variant.sh): re-read on every launch there, so that
// tools/kbench.py can flip them between rounds of one process.  The shipped library has none: the call folds to nullptr.
#ifdef LA_EXPERIMENTS
inline const char *dev_env(const char *name) { return getenv(name); }
#define LA_DEV_BIT(word, bits) ((word) & (bits))      // timing probes / A/B forms selected by internal epilogue bits
#else
inline const char *dev_env(const char *) { return nullptr; }
#define LA_DEV_BIT(word, bits) false
#endif

// ---- a function attribute (dynamic-LDS ceiling) set once per DEVICE, not once per process ----
// hipFuncSetAttribute applies to the current device's code object: a process driving two devices must set it on both.
// Bit d of `done` = set on device d; concurrent first callers both set it (idempotent).
struct DeviceOnce {
    unsigned long long done[4] = {0, 0, 0, 0};
    static int current() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        return d & 255;
    }
    bool pending() const {
        const int d = current();
        return !((__atomic_load_n(&done[d >> 6], __ATOMIC_ACQUIRE) >> (d & 63)) & 1ull);
    }
    void mark() {
        const int d = current();
        __atomic_fetch_or(&done[d >> 6], 1ull << (d & 63), __ATOMIC_RELEASE);
    }
};

// ---- optional per-kernel-family event timer (bench.py roofline leg) ---------
struct TimerScope {
    bool active;
    hipStream_t stream;
    TimerScope(const char *family, hipStream_t s, double work = 0.0);   // work: flops (or bytes) of the launch, summed for bracketed launches
    ~TimerScope();
};

// generic GEMM launcher with per-batch W / bias strides (la_gemm.hip); la_gemm is its strideW = 0 face
int gemm_run(int dtype, int M, int N, int K, int batch, const void *A, int64_t lda, int64_t strideA, const void *W,
             int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias, int64_t strideBias,
             const float *residual, int64_t ldr, int64_t strideR, int epilogue, hipStream_t stream);

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 on device --------------------------------------------------
typedef unsigned short bf16_t;  // raw bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// IEEE half: a distinct fundamental type, so the 16-bit kernels specialise on it beside bf16_t
typedef _Float16 f16_t;

// four f32 -> four 16-bit (or f32) storage elements, packed for one 8-byte (16-byte) store
template <typename T> struct Pack4;
template <> struct Pack4<bf16_t> {
    typedef ushort4 type;
    __device__ static __forceinline__ ushort4 run(float a, float b, float c, float d) {
        ushort4 r; r.x = f32_to_bf16(a); r.y = f32_to_bf16(b); r.z = f32_to_bf16(c); r.w = f32_to_bf16(d); return r;
    }
};
template <> struct Pack4<_Float16> {
    typedef ushort4 type;
    __device__ static __forceinline__ ushort4 run(float a, float b, float c, float d) {
        ushort4 r;
        r.x = __builtin_bit_cast(unsigned short, (_Float16)a); r.y = __builtin_bit_cast(unsigned short, (_Float16)b);
        r.z = __builtin_bit_cast(unsigned short, (_Float16)c); r.w = __builtin_bit_cast(unsigned short, (_Float16)d);
        return r;
    }
};

// ---- the "split" residual stream of the 16-bit modes -------------------------------------------------------------------
// The encoder's residual stream x is f32 in the reference (whisper/model.py ResidualAttentionBlock: x = x + attn(..); x = x + mlp(..)).
// In the 16-bit modes it lives as TWO arrays of the same [M][d] shape:
//   hi = x rounded to the operand type (bf16 / f16) -- at the same time the RAW A operand of the next LayerNorm-folded GEMM,
//        which LDS-DMA can only read as stored bytes;
//   lo = one byte per element: the remainder x - hi in units of ulp(hi) / 254, offset by 128 (|x - hi| <= ulp / 2 -> 1 .. 255,
//        symmetric, 128 = no remainder: an x that is exactly representable -- zero included -- comes back exactly).
// x is reproduced to 8 bits below the operand type's last place (bf16: 16 significant bits, f16: 19), far inside the rounding the
// 16-bit GEMM operands add every layer, and a read-modify-write of the stream moves 3 + 3 bytes per element instead of the
// 4 + 4 + 2 of an f32 stream with a 16-bit copy beside it: the residual GEMMs' epilogue is bound by exactly those bytes
// (tools/kbench.py epi).  frexp / ldexp keep the scaling exact for every exponent; hi = 0 decodes to exactly 0 + (lo - 128) 2^-SH.
template <typename T16> struct SplitRes;
template <> struct SplitRes<bf16_t> {
    static constexpr int SH = 16;                                    // unit = 2^(E - 7 - 8), frexp exponent e = E + 1
    static constexpr int EMIN = -200;                                // (bf16 subnormals start at 2^-126: never a residual value)
    __device__ static __forceinline__ float hi_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
    __device__ static __forceinline__ unsigned short round16(float x) { return f32_to_bf16(x); }
};
template <> struct SplitRes<_Float16> {
    static constexpr int SH = 19;                                    // unit = 2^(E - 10 - 8)
    static constexpr int EMIN = -13;                                 // f16 subnormals (|hi| < 2^-14) all have the ulp of E = -14
    __device__ static __forceinline__ float hi_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
    __device__ static __forceinline__ unsigned short round16(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
};
// x from (hi bits, lo byte as a float 0 .. 255): hi + (q - 128) (128 / 127) 2^(e - SH), e = frexp exponent of hi
constexpr float kSplitStep = 128.0f / 127.0f;
template <typename T16> __device__ __forceinline__ float split_decode(unsigned short hb, float q) {
    const float hf = SplitRes<T16>::hi_f32(hb);
    const int e = SplitRes<T16>::EMIN > -100 ? max(__builtin_amdgcn_frexp_expf(hf), SplitRes<T16>::EMIN) : __builtin_amdgcn_frexp_expf(hf);
    return hf + __builtin_amdgcn_ldexpf(fmaf(q, kSplitStep, -128.0f * kSplitStep), e - SplitRes<T16>::SH);
}
// x -> hi bits; q = the lo byte before rounding (1 .. 255 for |x - hi| <= ulp / 2; v_cvt_pk_u8_f32 rounds to nearest and saturates)
template <typename T16> __device__ __forceinline__ unsigned short split_encode(float x, float &q) {
    const unsigned short hb = SplitRes<T16>::round16(x);
    const float hf = SplitRes<T16>::hi_f32(hb);
    const int e = SplitRes<T16>::EMIN > -100 ? max(__builtin_amdgcn_frexp_expf(hf), SplitRes<T16>::EMIN) : __builtin_amdgcn_frexp_expf(hf);
    q = fmaf(__builtin_amdgcn_ldexpf(x - hf, SplitRes<T16>::SH - e), 127.0f / 128.0f, 128.0f);
    return hb;
}
__device__ __forceinline__ unsigned pack_u8x4(float a, float b, float c, float d) {
    unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
    w = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, w);
    return __builtin_amdgcn_cvt_pk_u8_f32(d, 3, w);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kDtype = LA_F32;
    __device__ static __forceinline__ float load(const float *p) { return *p; }
    __device__ static __forceinline__ void store(float *p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kDtype = LA_BF16;
    __device__ static __forceinline__ float load(const bf16_t *p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};
template <> struct Elem<_Float16> {
    static constexpr int kDtype = LA_F16;
    __device__ static __forceinline__ float load(const _Float16 *p) { return (float)*p; }
    __device__ static __forceinline__ void store(_Float16 *p, float v) { *p = (_Float16)v; }
};

// erfc(|z|) = t exp(-z^2 + P(t)), t = 1/(1 + z/2): Chebyshev fit with fractional error < 1.2e-7 everywhere
// (Numerical Recipes erfcc).  Branch-free, ~18 VALU ops; libm's erff costs ~100 with its range branches executed
// by every lane, which made the exact-GELU epilogue of the MLP-up GEMM as expensive as its main loop.
__device__ __forceinline__ float erf_fast(float x) {
    const float z = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.0f));
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f);
    p = fmaf(p, t, 1.48851587f);
    p = fmaf(p, t, -1.13520398f);
    p = fmaf(p, t, 0.27886807f);
    p = fmaf(p, t, -0.18628806f);
    p = fmaf(p, t, 0.09678418f);
    p = fmaf(p, t, 0.37409196f);
    p = fmaf(p, t, 1.00002368f);
    p = fmaf(p, t, -1.26551223f);
    const float erfc_z = t * __expf(fmaf(-z, z, p));
    const float e = 1.0f - erfc_z;
    return x < 0.f ? -e : e;
}

// exact (erf) GELU of whisper's nn.GELU(); abs error of the erf fit <= 1.2e-7
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// d/dx of the erf GELU: Phi(x) + x phi(x)  (la_gelu_bwd_f32, and the LA_EPI_RES_GELU_GRAD epilogue of la_gemm_f16x2)
__device__ __forceinline__ float gelu_erf_grad(float v) {
    const float cdf = 0.5f * (1.0f + erf_fast(v * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    return cdf + v * pdf;
}

// GELU for results that are rounded to bf16 anyway, two values per lane on the packed-FP32 pipe (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: 2 lanes per issue).  Same erfc form, degree-5 fit (fractional error of erfc <= 2.8e-5
// everywhere, i.e. 1/140 of a bf16 ulp; max abs error of gelu 1.3e-6), log2(e) folded into the coefficients so the
// exponential is a bare v_exp_f32:   gelu(x) = 0.5 (x + |x| (1 - t 2^(q(t) - 0.5 log2e x^2))),  t = 1 / (1 + |x| / (2 sqrt 2)).
// 15.5 issue slots per value against 32 for gelu_erf: the MLP-up epilogue runs at 1 workgroup per CU with nothing to
// hide it behind (DESIGN.md "GEMM findings"), so its VALU count is wall time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 d = __builtin_elementwise_fma(ax, f32x2{0.35355339059f, 0.35355339059f}, f32x2{1.f, 1.f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 p = f32x2{3.483918905e-01f, 3.483918905e-01f};
    p = __builtin_elementwise_fma(p, t, f32x2{-1.077869773e+00f, -1.077869773e+00f});
    p = __builtin_elementwise_fma(p, t, f32x2{7.467902899e-01f, 7.467902899e-01f});
    p = __builtin_elementwise_fma(p, t, f32x2{3.336617649e-01f, 3.336617649e-01f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.477083683e+00f, 1.477083683e+00f});
    p = __builtin_elementwise_fma(p, t, f32x2{-1.828017831e+00f, -1.828017831e+00f});
    const f32x2 y2 = __builtin_elementwise_fma(x * f32x2{-0.72134752044f, -0.72134752044f}, x, p);
    const f32x2 e = {__builtin_amdgcn_exp2f(y2.x), __builtin_amdgcn_exp2f(y2.y)};
    const f32x2 u = __builtin_elementwise_fma(-ax, t * e, ax);          // |x| (1 - erfc(|x| / sqrt 2))
    return (x + u) * f32x2{0.5f, 0.5f};
}

// GELU as x * sigmoid(z(x)), z an odd quintic in x clamped to |x| <= 8: one multiply-chain, v_exp_f32, v_rcp_f32 -- 11 issue
// slots per value against 15.5 for gelu_pk (the MLP-up epilogue is VALU-bound at one workgroup per CU).  Minimax fit of
// x sigmoid(x (a + b x^2 + c x^4)) to x Phi(x) on [-9, 9] (tools fit in DESIGN.md): max ABSOLUTE error 2.5e-5 (at x = -0.56);
// for results rounded to bf16 / f16 only.  The clamp keeps z monotone where the quintic's negative x^5 term would turn it
// (|x| > 11) and saturates the sigmoid to exactly 1 / ~1e-12 beyond |x| = 8, where Phi is 1 - 6e-16 / 6e-16.
// log2(e) is folded into the coefficients (and the sign, so that the exponential is exp2(-z log2 e)).
__device__ __forceinline__ float gelu_sig(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    const float s = xc * xc;
    float p = fmaf(1.0148166172e-03f, s, -1.0677913190e-01f);     // -log2e * (c, b)
    p = fmaf(p, s, -2.3011175945e+00f);                            // -log2e * a
    const float e = __builtin_amdgcn_exp2f(p * xc);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// gelu_sig on two values: the same operations in the same order (bit-identical results), with the squares, the two polynomial
// steps, p * x, 1 + e and the final product as packed f32 instructions -- 2 clamps + 5 packed + 4 transcendental issue slots
// per PAIR instead of 14 + 4.
__device__ __forceinline__ f32x2 gelu_sig2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -8.0f, 8.0f), __builtin_amdgcn_fmed3f(x.y, -8.0f, 8.0f)};
    const f32x2 s = xc * xc;
    f32x2 p = __builtin_elementwise_fma(f32x2{1.0148166172e-03f, 1.0148166172e-03f}, s, f32x2{-1.0677913190e-01f, -1.0677913190e-01f});
    p = __builtin_elementwise_fma(p, s, f32x2{-2.3011175945e+00f, -2.3011175945e+00f});
    const f32x2 z = p * xc;
    const f32x2 e = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
    const f32x2 d = e + f32x2{1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return x * r;
}

// nn.Mish: x * tanh(softplus(x)); softplus with torch's threshold 20
__device__ __forceinline__ float mish(float x) {
    float sp = x > 20.0f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

// global -> LDS DMA, 16 B per lane (global_load_lds_dwordx4): lane i's 16 bytes land at lds_base + 16*i, lds_base wave-uniform.
// Issued from INLINE ASM on purpose: hipcc models an LDS-DMA issued through the builtin as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the next ds_read of the loop (seen in the ISA of every kernel here), which drains the
// prefetch it was issued for.  From asm the DMA is invisible to that bookkeeping; completion is OUR job: a counted
// s_waitcnt vmcnt(N) by the issuing wave, then a barrier, then the ds_read (cdna guide 5.7).  M0 (the LDS base) is
// saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void *gsrc, const void *lds_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// Same DMA with the address split as the hardware wants it in a hot loop: a wave-uniform 64-bit base in SGPRs plus a
// loop-invariant per-lane 32-bit byte offset in one VGPR, and the LDS destination as a ready M0 value.  Two SALU + the
// DMA per issue (the generic form above spends ~25 instructions on 64-bit per-lane address arithmetic).  M0 is not
// restored: nothing else in these kernels reads it (LDS-DMA is its only user on gfx950 besides GWS / sendmsg).
__device__ __forceinline__ void glds16_so(unsigned voff, const void *sbase, unsigned m0_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(m0_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr_u32(const void *p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace la
