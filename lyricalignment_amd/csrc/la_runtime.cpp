// la_runtime.cpp -- library-level entry points: version, error text, arch check, kernel timer.
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <vector>

#include "la_common.h"

namespace la {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
}

// ---- per-stream scratch -------------------------------------------------------
// Small internal scratch (split-K partial sums, column-sum partials): one buffer per (device, stream, purpose), grown on demand
// and kept.  Uses on one stream are ordered, so a buffer is never shared by kernels that can overlap.  (hipMallocAsync /
// hipFreeAsync per call cost ~50 us of host time each on this stack: more than the kernels they served.)
void *stream_scratch(hipStream_t stream, int purpose, size_t bytes) {
    struct Slot { int dev; hipStream_t stream; int purpose; void *ptr; size_t bytes; };
    static std::mutex mu;
    static std::vector<Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (Slot &sl : slots)
        if (sl.dev == dev && sl.stream == stream && sl.purpose == purpose) {
            if (sl.bytes >= bytes) return sl.ptr;
            if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;      // earlier users of the small buffer
            (void)hipFree(sl.ptr);
            sl.ptr = nullptr; sl.bytes = 0;
            if (hipMalloc(&sl.ptr, bytes) != hipSuccess) return nullptr;
            sl.bytes = bytes;
            return sl.ptr;
        }
    Slot sl{dev, stream, purpose, nullptr, bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes};
    if (hipMalloc(&sl.ptr, sl.bytes) != hipSuccess) return nullptr;
    slots.push_back(sl);
    return sl.ptr;
}

// ---- options --------------------------------------------------------------------
namespace {
struct OptionDesc { const char *name, *env; int Options::*field; bool env_inverts; };
// env_inverts: the variable's presence (LA_GEMM_NO_SPLITK, LA_VITERBI_NO_DPP) switches the option OFF
const OptionDesc kOptions[] = {
    {"gemm_tile", "LA_GEMM_TILE", &Options::gemm_tile, false},       {"gemm_loop", "LA_PP_DBG", &Options::gemm_loop, false},
    {"gemm_splitk", "LA_GEMM_NO_SPLITK", &Options::gemm_splitk, true}, {"attn_nw", "LA_ATTN_NW", &Options::attn_nw, false},
    {"gru_nw", "LA_GRU_NW", &Options::gru_nw, false},                 {"gru_fence", "LA_GRU_FENCE", &Options::gru_fence, false},
    {"gru_handoff", "LA_GRU_HANDOFF", &Options::gru_handoff, false}, {"gru_poll_delay", "LA_GRU_POLL_DELAY", &Options::gru_poll_delay, false},
    {"viterbi_dpp", "LA_VITERBI_NO_DPP", &Options::viterbi_dpp, true}, {"head_clip_cap", "LA_HEAD_CLIP_CAP", &Options::head_clip_cap, false},
    {"ln_fusion", "LA_LN_FUSION", &Options::ln_fusion, false},       {"resid_split", "LA_RESID_SPLIT", &Options::resid_split, false},
    {"x2_inference", "LA_X2_INFERENCE", &Options::x2_inference, false},
    {"gru_timeout_us", "LA_GRU_TIMEOUT_US", &Options::gru_timeout_us, false}, {"gru_fault_step", "LA_GRU_FAULT_STEP", &Options::gru_fault_step, false},
};
}  // namespace

Options &opts() {
    static Options o = [] {
        Options v;
        for (const OptionDesc &d : kOptions) {
            const char *e = getenv(d.env);
            if (!e) continue;
            v.*(d.field) = d.env_inverts ? 0 : atoi(e);
        }
        return v;
    }();
    return o;
}

// ---- kernel-family timer ------------------------------------------------------
namespace {
struct TimerState {
    std::mutex mu;
    bool enabled = false;
    std::string family;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;  // recorded
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;   // reusable
    // Sampling: a hipEventRecord is a barrier packet of its own -- two per launch cost the stream ~6.6 us of idle time each between
    // back-to-back kernels (rocprofv3 trace, round 3: 0.66 ms of a 44 ms step with every GEMM launch bracketed).  period = n brackets
    // every n-th launch of the family; the work (flops) of the bracketed launches is summed beside their time.
    int period = 1;
    int64_t seen = 0;          // launches of the family since reset
    double work = 0.0;         // summed work of the bracketed launches
};
TimerState &ts() {
    static TimerState s;
    return s;
}
}  // namespace

TimerScope::TimerScope(const char *family, hipStream_t s, double work) : active(false), stream(s) {
    TimerState &t = ts();
    if (!t.enabled) return;
    std::lock_guard<std::mutex> lk(t.mu);
    if (!t.enabled || t.family != family) return;
    if (t.seen++ % t.period != 0) return;
    std::pair<hipEvent_t, hipEvent_t> pr;
    if (!t.pool.empty()) {
        pr = t.pool.back();
        t.pool.pop_back();
    } else {
        if (hipEventCreate(&pr.first) != hipSuccess) return;
        if (hipEventCreate(&pr.second) != hipSuccess) { (void)hipEventDestroy(pr.first); return; }
    }
    (void)hipEventRecord(pr.first, stream);
    t.pairs.push_back(pr);
    t.work += work;
    active = true;
}

TimerScope::~TimerScope() {
    if (!active) return;
    TimerState &t = ts();
    std::lock_guard<std::mutex> lk(t.mu);
    if (!t.pairs.empty()) (void)hipEventRecord(t.pairs.back().second, stream);
}

}  // namespace la

extern "C" int la_version(void) { return 2; }

extern "C" const char *la_last_error(void) { return la::err_buf(); }

extern "C" int la_device_arch_ok(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int la_set_option(const char *name, int64_t value) {
    LA_CHECK_ARG(name, "set_option: null name");
    for (const la::OptionDesc &d : la::kOptions)
        if (strcmp(d.name, name) == 0) {
            la::opts().*(d.field) = (int)value;
            return LA_OK;
        }
    la::set_error("set_option: unknown option '%s'", name);
    return LA_EINVAL;
}

extern "C" int la_get_option(const char *name, int64_t *value) {
    LA_CHECK_ARG(name && value, "get_option: null pointer");
    for (const la::OptionDesc &d : la::kOptions)
        if (strcmp(d.name, name) == 0) {
            *value = la::opts().*(d.field);
            return LA_OK;
        }
    la::set_error("get_option: unknown option '%s'", name);
    return LA_EINVAL;
}

extern "C" int la_has_experiments(void) {
#ifdef LA_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int la_timer_enable(const char *name) {
    if (!name) return LA_EINVAL;
    la::TimerState &t = la::ts();
    std::lock_guard<std::mutex> lk(t.mu);
    t.family = name;
    t.enabled = true;
    return LA_OK;
}

extern "C" int la_timer_disable(void) {
    la::TimerState &t = la::ts();
    std::lock_guard<std::mutex> lk(t.mu);
    t.enabled = false;
    return LA_OK;
}

extern "C" int la_timer_reset(void) {
    la::TimerState &t = la::ts();
    std::lock_guard<std::mutex> lk(t.mu);
    for (auto &p : t.pairs) t.pool.push_back(p);
    t.pairs.clear();
    t.seen = 0;
    t.work = 0.0;
    return LA_OK;
}

extern "C" int la_timer_sample(int32_t period) {
    if (period < 1) return LA_EINVAL;
    la::TimerState &t = la::ts();
    std::lock_guard<std::mutex> lk(t.mu);
    t.period = period;
    return LA_OK;
}

extern "C" int la_timer_read_work(double *total_ms, int64_t *timed_launches, double *timed_work, int64_t *all_launches) {
    int64_t n = 0;
    const int rc = la_timer_read(total_ms, &n);
    if (rc != LA_OK) return rc;
    la::TimerState &t = la::ts();
    std::lock_guard<std::mutex> lk(t.mu);
    if (timed_launches) *timed_launches = n;
    if (timed_work) *timed_work = t.work;
    if (all_launches) *all_launches = t.seen;
    return LA_OK;
}

extern "C" int la_timer_read(double *total_ms, int64_t *launches) {
    if (!total_ms || !launches) return LA_EINVAL;
    la::TimerState &t = la::ts();
    std::lock_guard<std::mutex> lk(t.mu);
    double sum = 0.0;
    for (auto &p : t.pairs) {
        hipError_t e = hipEventSynchronize(p.second);
        if (e != hipSuccess) { la::set_error("timer: %s", hipGetErrorString(e)); return LA_EHIP; }
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, p.first, p.second);
        if (e != hipSuccess) { la::set_error("timer: %s", hipGetErrorString(e)); return LA_EHIP; }
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)t.pairs.size();
    return LA_OK;
}
