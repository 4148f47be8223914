// Inter-workgroup hand-off latency on gfx950 (one-way, producer store -> consumer sees it), the quantity that bounds one step of
// the persistent GRU recurrence (csrc/la_gru.hip).  Two workgroups bounce a sequence number; variants differ in the cache
// policy of the store / the polling load and in whether the two workgroups sit on the same XCD (read from HW_REG_XCC_ID).
//   hipcc --offload-arch=gfx950 -O3 -o handoff_bench tools/handoff_bench.hip && ./handoff_bench
// Also: the same bounce with a PAYLOAD (bytes written by 256 threads, flag after vmcnt(0)+barrier, consumer reads it all).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
enum Mode { SC1 = 0, SC0 = 1, SC0SC1 = 2, PLAIN_FENCE = 3, ATOMIC_AGENT = 4, NT_SC1 = 5 };

template <int MODE> __device__ __forceinline__ void st(unsigned *p, unsigned v) {
    if (MODE == SC1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == SC0) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (MODE == SC0SC1) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == NT_SC1) asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (MODE == ATOMIC_AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else { *(volatile unsigned *)p = v; }
}
template <int MODE> __device__ __forceinline__ unsigned ld(unsigned *p) {
    unsigned v;
    if (MODE == SC1) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == SC0) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == SC0SC1) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == NT_SC1) asm volatile("global_load_dword %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == ATOMIC_AGENT) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v = *(volatile unsigned *)p;
    return v;
}

// table[0..nblk): xcc id per block; table[64]: arrival count; table[65]: partner block id; slots: two flags 4 KB apart
template <int MODE>
__global__ void bounce(unsigned *table, unsigned *slots, int iters, int want_same, unsigned long long *ticks) {
    const int b = blockIdx.x, nblk = gridDim.x;
    __shared__ int partner_s;
    if (threadIdx.x == 0) {
        __hip_atomic_store(table + b, xcc_id() + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(table + 64, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spin = 0;
        while (__hip_atomic_load(table + 64, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk && ++spin < (1L << 26)) {}
        const unsigned x0 = __hip_atomic_load(table + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int partner = -1;
        for (int i = 1; i < nblk; ++i) {
            const unsigned xi = __hip_atomic_load(table + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((xi == x0) == (want_same != 0)) { partner = i; break; }
        }
        partner_s = partner;
        if (b == 0) table[65] = (unsigned)partner;
    }
    __syncthreads();
    const int partner = partner_s;
    if (partner < 0 || (b != 0 && b != partner)) return;
    if (threadIdx.x != 0) return;
    unsigned *mine = slots + (b == 0 ? 0 : 1024), *theirs = slots + (b == 0 ? 1024 : 0);
    const long limit = 1L << 22;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; ++i) {
        if (b == 0) {
            st<MODE>(mine, (unsigned)i);
            long spin = 0;
            while (ld<MODE>(theirs) != (unsigned)i && ++spin < limit) {}
            if (spin >= limit) break;
        } else {
            long spin = 0;
            while (ld<MODE>(theirs) != (unsigned)i && ++spin < limit) {}
            if (spin >= limit) break;
            st<MODE>(mine, (unsigned)i);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (b == 0) ticks[0] = t1 - t0;
}

// Payload bounce: TEAM workgroups of 256 threads; each step every member writes `bytes_per_wg` (16 B per thread and store,
// sc1 or plain+fence), publishes (vmcnt(0), barrier, [release], counter add), waits for all members' counters, [acquire],
// reads ALL members' payload (TEAM * bytes_per_wg) and folds it into the next payload.  = the GRU's step without its math.
//   PROTO 0: agent-scope release / acquire fences + plain accesses (the GRU's default)
//   PROTO 1: sc1 stores / sc1 loads, relaxed counter (write-through form)
//   PROTO 2: same-XCD form: plain stores, sc0 loads, counter atomics at workgroup.. no: agent-scope relaxed atomics (executed at L2)
template <int PROTO>
__global__ void __launch_bounds__(256) team_step(unsigned *table, uint4 *payload, unsigned *ctr, int steps, int bytes_per_wg, int team,
                                                 int want_same, unsigned long long *ticks) {
    // membership: the first `team` blocks whose XCC id equals (want_same) / blocks with pairwise different ids (!want_same)
    const int b = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
    __shared__ int rank_s;
    if (tid == 0) {
        __hip_atomic_store(table + b, xcc_id() + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(table + 64, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spin = 0;
        while (__hip_atomic_load(table + 64, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk && ++spin < (1L << 26)) {}
        int rank = -1, taken = 0;
        unsigned used = 0;
        const unsigned x0 = __hip_atomic_load(table + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < nblk && taken < team; ++i) {
            const unsigned xi = __hip_atomic_load(table + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool take;
            if (want_same) take = xi == x0;
            else { take = !(used & (1u << xi)); }
            if (take) { used |= 1u << xi; if (i == b) rank = taken; ++taken; }
        }
        if (taken < team) rank = -1;
        rank_s = rank;
        if (b == 0) table[65] = (unsigned)taken;
    }
    __syncthreads();
    const int rank = rank_s;
    if (rank < 0) return;
    const int n16 = bytes_per_wg / 16;               // 16-B units per member
    uint4 acc = make_uint4(tid, rank, 0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ int ok_s;
    unsigned bad = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 *dst = payload + (size_t)(s & 1) * team * n16 + (size_t)rank * n16;
        acc.w = (unsigned)s + 1u;                    // tag: which step wrote this chunk (checked by every reader)
        if (PROTO == 4) {
            // data polling: no drain, no barrier, no counter -- sc1 stores, readers poll the chunks' tags
            for (int i = tid; i < n16; i += 256) {
                const u32x4 a4 = {acc.x, acc.y, acc.z, acc.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(a4) : "memory");
            }
            const uint4 *src = payload + (size_t)(s & 1) * team * n16;
            const int mine_n = (team * n16 + 255 - tid) / 256;         // chunks of this thread (<= 8)
            u32x4 v[8];
            unsigned pending = (1u << mine_n) - 1u;
            long spin = 0;
            while (pending && ++spin < (1L << 20)) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (pending & (1u << k)) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(src + tid + k * 256) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((pending & (1u << k)) && v[k][3] == (unsigned)s + 1u) pending &= ~(1u << k);
            }
            if (pending) { bad |= 0x80000000u; break; }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < mine_n) { acc.x += v[k][0]; acc.y ^= v[k][1]; acc.z += v[k][2]; }
            __syncthreads();                                           // (the GRU would stage into LDS here)
            continue;
        }
        for (int i = tid; i < n16; i += 256) {
            if (PROTO == 1 || PROTO == 3) {
                const u32x4 a4 = {acc.x, acc.y, acc.z, acc.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(a4) : "memory");
            } else dst[i] = acc;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (PROTO == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(ctr + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spin = 0;
            while (__hip_atomic_load(ctr + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)team && ++spin < (1L << 22)) {}
            if (PROTO == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            ok_s = spin < (1L << 22);
        }
        __syncthreads();
        if (!ok_s) break;
        const uint4 *src = payload + (size_t)(s & 1) * team * n16;
        if (PROTO == 3) {
            u32x4 v[8];
            const int mine_n = (team * n16 + 255 - tid) / 256;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < mine_n) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(src + tid + k * 256) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < mine_n) { acc.x += v[k][0]; acc.y ^= v[k][1]; acc.z += v[k][2]; bad += v[k][3] != (unsigned)s + 1u; }
            continue;
        }
        for (int i = tid; i < team * n16; i += 256) {
            uint4 v;
            u32x4 v4;
            if (PROTO == 1) { asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v4) : "v"(src + i) : "memory"); v = make_uint4(v4[0], v4[1], v4[2], v4[3]); }
            else if (PROTO == 2) { asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v4) : "v"(src + i) : "memory"); v = make_uint4(v4[0], v4[1], v4[2], v4[3]); }
            else v = src[i];
            acc.x += v.x; acc.y ^= v.y; acc.z += v.z; bad += v.w != (unsigned)s + 1u;
        }
        // NOTE: double-buffered payload: step s+1 writes the other half, and nobody can be two steps ahead (counter of s+1
        // needs everyone's arrival), so a half is never overwritten while still being read.
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (rank == 0 && tid == 0) { ticks[0] = t1 - t0; ticks[1] = acc.x + acc.y + acc.z + acc.w; }
    if (bad) atomicAdd((unsigned *)(ticks + 2), bad & 0x7fffffffu ? 1u : 0x10000u);
}

int main(int argc, char **argv) {
    const int iters = 20000, nblk = 32;
    unsigned *table, *slots, *ctr;
    unsigned long long *ticks;
    uint4 *payload;
    CK(hipMalloc(&table, 4096)); CK(hipMalloc(&slots, 16384)); CK(hipMalloc(&ticks, 64));
    CK(hipMalloc(&payload, 1 << 22)); CK(hipMalloc(&ctr, 4 * 65536));
    const char *names[] = {"sc1 store / sc1 load", "sc0 store / sc0 load", "sc0 sc1 / sc0 sc1", "volatile plain", "agent-scope relaxed atomics", "sc1 nt"};
    for (int same = 0; same <= 1; ++same)
        for (int mode = 0; mode < 6; ++mode) {
            if (!same && (mode == SC0 || mode == PLAIN_FENCE)) continue;     // not coherent across XCDs: would spin to the limit
            CK(hipMemset(table, 0, 4096)); CK(hipMemset(slots, 0, 16384)); CK(hipMemset(ticks, 0, 64));
            switch (mode) {
            case 0: bounce<0><<<nblk, 64>>>(table, slots, iters, same, ticks); break;
            case 1: bounce<1><<<nblk, 64>>>(table, slots, iters, same, ticks); break;
            case 2: bounce<2><<<nblk, 64>>>(table, slots, iters, same, ticks); break;
            case 3: bounce<3><<<nblk, 64>>>(table, slots, iters, same, ticks); break;
            case 4: bounce<4><<<nblk, 64>>>(table, slots, iters, same, ticks); break;
            case 5: bounce<5><<<nblk, 64>>>(table, slots, iters, same, ticks); break;
            }
            CK(hipDeviceSynchronize());
            unsigned long long t; unsigned tb[66], sl[2048];
            CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(tb, table, sizeof(tb), hipMemcpyDeviceToHost));
            CK(hipMemcpy(sl, slots, sizeof(sl), hipMemcpyDeviceToHost));
            printf("bounce %-30s %s XCD (partner block %d, xcc %u/%u): done %u/%d, one-way %.0f ns\n", names[mode], same ? "same" : "diff",
                   (int)tb[65], tb[0] - 1, (int)tb[65] >= 0 ? tb[tb[65]] - 1 : 99, sl[0], iters, t * 10.0 / (2.0 * (sl[0] ? sl[0] : 1)));
        }
    const int steps = 5000;
    const int sizes[] = {256, 2048, 4096};   // 6 x 4096 B = 1536 chunks = 6 per thread
    for (int same = 0; same <= 1; ++same)
        for (int team : {2, 6})
            for (int bytes : sizes)
                for (int proto = 0; proto < 5; ++proto) {
                    if (proto == 2 && !same) continue;
                    CK(hipMemset(table, 0, 4096)); CK(hipMemset(ctr, 0, 4 * 65536)); CK(hipMemset(ticks, 0, 64));
                    const int grid = same ? 64 : 32;
                    if (proto == 0) team_step<0><<<grid, 256>>>(table, payload, ctr, steps, bytes, team, same, ticks);
                    if (proto == 1) team_step<1><<<grid, 256>>>(table, payload, ctr, steps, bytes, team, same, ticks);
                    if (proto == 2) team_step<2><<<grid, 256>>>(table, payload, ctr, steps, bytes, team, same, ticks);
                    if (proto == 3) team_step<3><<<grid, 256>>>(table, payload, ctr, steps, bytes, team, same, ticks);
                    if (proto == 4) team_step<4><<<grid, 256>>>(table, payload, ctr, steps, bytes, team, same, ticks);
                    CK(hipDeviceSynchronize());
                    unsigned long long t[3]; unsigned tb[66];
                    CK(hipMemcpy(t, ticks, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(tb, table, sizeof(tb), hipMemcpyDeviceToHost));
                    printf("team %d x %4d B  %s XCD  proto %d (%s): members %u, %.2f us / step, stale/timeout threads %llu\n", team, bytes, same ? "same" : "diff", proto,
                           proto == 0 ? "release/acquire fences" : proto == 1 ? "sc1 stores+loads, serial" : proto == 2 ? "plain stores, sc0 loads"
                           : proto == 3 ? "sc1, loads in flight together" : "sc1 data polling, no counter", tb[65],
                           t[0] * 0.01 / steps, t[2]);
                }
    return 0;
}
