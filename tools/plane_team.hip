// Tuning aid (Next 1a): can the y -> z -> y passes of ONE x-plane of a 512^3 half spectrum run out of an
// XCD's 4 MiB L2 instead of the Infinity Cache?
//
// A plane is 512 k_y rows x 256 packed k_z columns x 8 B = 1 MiB (row pitch 2176 B as in the product).  The product
// runs   y-inverse (16 tiles of 512 rows x 128 B per plane)  ->  z (c2r, exp, r2c: 512 rows of 2 KiB; writes the real
// plane, 1 MiB)  ->  y-forward   as three launches per batch of 64-128 planes; the planes wait in the Infinity Cache
// between the launches (tools/mall_copy.hip: 6.3-6.5 TB/s).  Here: resident "teams" of 16 workgroups x 1024 threads,
// all on one XCD (formed at run time from HW_REG_XCC_ID, never assumed), take one plane at a time from a queue and
// step it through the three phases, handing it over through team barriers:
//     every wave: s_waitcnt vmcnt(0) (its stores are in the XCD's L2)  ->  workgroup barrier  ->  one lane: atomic add
//     on the team's counter + sc1-load poll (bounded, error flag)  ->  [agent acquire = L1 invalidate]  ->  barrier.
// The phases move the product's bytes with the product's access patterns (tile columns / whole rows, 8 B per lane)
// and add 1 to every value, so that a stale read shows up in the check (expected: +3 everywhere, real plane = +2).
// Compared with the same phases as three launches per 128-plane batch.
//   hipcc --offload-arch=gfx950 -O3 tools/plane_team.hip -o /tmp/plane_team && /tmp/plane_team
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define RANGE 0xFFFFFFF0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)RANGE, 0x00020000);
}
template <int AUX> __device__ __forceinline__ float2 ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX));
}
template <int AUX> __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)soff, AUX);
}

constexpr int N = 512, NZP = 272, TZ = 16, TEAM = 16, NT = 1024;
constexpr long long PITCH = NZP * 8, HPLANE = (long long)(N + 1) * PITCH, RPLANE = (long long)N * N * 4;
constexpr int SC1 = 16, NTP = 2;

// control block (32-bit words; every hot word on a 128-byte line of its own)
enum { W_TICKET = 0 /* [8] x 32 */, W_ARRIVED = 256, W_NEXT = 288, W_ERROR = 320, W_EXITED = 352, W_TEAMS = 512 /* team: 64 words */ };
enum { T_COUNT = 0, T_WORD = 32 };

struct Args {
    char* half; char* real; unsigned* ctl;
    int nplanes; int flags; int delay;
};
enum { F_SC1_LOADS = 1, F_AGENT_ATOMICS = 2, F_NO_ACQUIRE = 4, F_NO_REAL = 8, F_NO_BARRIER = 16 };

__device__ __forceinline__ void spin_delay(int d) { for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(8); }

// the y-pass tile: rows t + 64 e of tile column m
template <int LAUX> __device__ __forceinline__ void phase_y(char* plane, int m, int delay) {
    const int c = threadIdx.x % TZ, t = threadIdx.x / TZ;
    const int t0 = __builtin_amdgcn_readfirstlane(t);
    const __amdgpu_buffer_rsrc_t r = rsrc(plane + (long long)m * TZ * 8 + (long long)t0 * PITCH);
    const unsigned voff = (unsigned)((t - t0) * PITCH + c * 8);
    float2 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ld<LAUX>(r, voff, (unsigned)(e * 64 * PITCH));
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e].x += 1.f; v[e].y += 1.f; }
    spin_delay(delay);
#pragma unroll
    for (int e = 0; e < 8; ++e) st<0>(r, voff, (unsigned)(e * 64 * PITCH), v[e]);
}
// the z pass: rows 32 m .. 32 m + 31, 32 lanes per row, 8 points per lane; writes the real plane (streaming) and the row back
template <int LAUX> __device__ __forceinline__ void phase_z(char* plane, char* rplane, int m, int delay, bool real_out) {
    const int l = threadIdx.x / 32, t = threadIdx.x % 32;
    const int row = 32 * m + l;
    const int row0 = __builtin_amdgcn_readfirstlane(row);
    const __amdgpu_buffer_rsrc_t r = rsrc(plane + (long long)row0 * PITCH);
    const unsigned voff = (unsigned)((row - row0) * PITCH + t * 8);
    float2 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ld<LAUX>(r, voff, (unsigned)(e * 256));
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e].x += 1.f; v[e].y += 1.f; }
    spin_delay(delay);
    if (real_out) {
        const __amdgpu_buffer_rsrc_t rr = rsrc(rplane + (long long)row0 * (N * 4));
        const unsigned ro = (unsigned)((row - row0) * (N * 4) + t * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) st<NTP>(rr, ro, (unsigned)(e * 256), v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) st<0>(r, voff, (unsigned)(e * 256), v[e]);
}

__device__ __forceinline__ unsigned poll_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_word(unsigned* p, unsigned v, bool agent) {
    return agent ? __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                 : __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// exact read: an atomic read-modify-write is performed where the counter's adds are (memory side for agent scope, the XCD's L2 for
// workgroup scope) -- used where a decision is taken on a NEGATIVE observation (belt and braces: no stale sc1 poll was ever observed)
__device__ __forceinline__ unsigned rmw_read(unsigned* p, bool agent) { return add_word(p, 0u, agent); }
constexpr int SPIN_LIMIT = 1 << 21;        // x ~0.2 us: a wait that long is a failure, never an open spin

// returns false on time-out (error flag set; the caller leaves)
__device__ __forceinline__ bool team_barrier(unsigned* team, unsigned target, unsigned* ctl, int flags, unsigned* sh, unsigned want_seq) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int it = 0;
        if (!(flags & F_NO_BARRIER)) {
            add_word(team + T_COUNT, 1u, flags & F_AGENT_ATOMICS);
            while (it < SPIN_LIMIT) {      // (a stale poll can only read low: a positive answer is final; every 8th poll is exact)
                const unsigned v = (it & 7) == 7 ? rmw_read(team + T_COUNT, flags & F_AGENT_ATOMICS) : poll_load(team + T_COUNT);
                if (v >= target) break;
                __builtin_amdgcn_s_sleep(2); ++it;
            }
        }
        unsigned ok = it < SPIN_LIMIT;
        if (!ok) {
            __hip_atomic_store(ctl + W_ERROR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* d = ctl + 7168 + 8 * (__hip_atomic_fetch_add(ctl + 7000, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 63u);
            d[0] = (unsigned)(team - ctl - W_TEAMS) / 64; d[1] = target; d[2] = poll_load(team + T_COUNT); d[3] = sh[3]; d[4] = poll_load(team + T_WORD);
            d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u; d[6] = blockIdx.x;
        }
        sh[0] = ok;
        if (want_seq) {      // the team's next plane (published by the leader before it arrived; polled only by the no-barrier variant)
            unsigned w = poll_load(team + T_WORD + (want_seq & 1u));
            while ((w >> 16) != want_seq && it < SPIN_LIMIT) { __builtin_amdgcn_s_sleep(2); ++it; w = poll_load(team + T_WORD + (want_seq & 1u)); }
            sh[1] = w;
        }
        if (!(flags & (F_NO_ACQUIRE | F_SC1_LOADS))) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    return sh[0] != 0;
}

template <int LAUX>
__global__ __launch_bounds__(NT, 8) void k_plane_team(Args a) {
    extern __shared__ char smem[];
    unsigned* sh = reinterpret_cast<unsigned*>(smem);          // [0] ok, [1] word, [2] team, [3] member
    unsigned* ctl = a.ctl;
    const bool agent = a.flags & F_AGENT_ATOMICS;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
        const unsigned ticket = __hip_atomic_fetch_add(ctl + W_TICKET + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the ticket must be counted before the arrival is: a workgroup that sees "everyone has arrived" decides from the
        // ticket counters whether its team is complete (two independent atomics may be performed in either order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctl + W_ARRIVED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned tl = ticket / TEAM, full = (tl + 1) * TEAM;
        // wait until the team is complete, or until every workgroup of the grid has a ticket (then it never will be)
        int it = 0; bool ok = false;
        while (it < SPIN_LIMIT) {
            if (poll_load(ctl + W_TICKET + xcc * 32) >= full) { ok = true; break; }
            if (poll_load(ctl + W_ARRIVED) >= gridDim.x) { ok = rmw_read(ctl + W_TICKET + xcc * 32, true) >= full; break; }
            __builtin_amdgcn_s_sleep(4); ++it;
        }
        if (it >= SPIN_LIMIT) __hip_atomic_store(ctl + W_ERROR, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[0] = ok && tl < 8; sh[2] = xcc * 8 + (tl & 7); sh[3] = ticket % TEAM;
    }
    __syncthreads();
    const bool in_team = sh[0] != 0;
    const int team_id = (int)sh[2], m = (int)sh[3];
    unsigned* team = ctl + W_TEAMS + team_id * 64;
    __syncthreads();
    if (in_team) {
        // first plane: the leader draws it and publishes (sequence number in the high half).  Two alternating slots: the word of
        // plane k + 1 is published while slow members may still be waiting for the word of plane k (the first version had ONE slot
        // and overwrote word 1 with word 2 at once: members dispatched late never saw word 1 and their team hung at its first barrier)
        if (threadIdx.x == 0) {
            if (m == 0) {
                const unsigned p = __hip_atomic_fetch_add(ctl + W_NEXT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(team + T_WORD + 1, (1u << 16) | (p < 0xffffu ? p : 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int it = 0; unsigned w = 0;
            while (it < SPIN_LIMIT) { w = poll_load(team + T_WORD + 1); if ((w >> 16) == 1u) break; __builtin_amdgcn_s_sleep(2); ++it; }
            if (it >= SPIN_LIMIT) __hip_atomic_store(ctl + W_ERROR, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[0] = it < SPIN_LIMIT; sh[1] = w;
        }
        __syncthreads();
        bool ok = sh[0] != 0;
        unsigned cur = sh[1] & 0xffffu, seq = 1, bar = 0;
        __syncthreads();
        while (ok && (int)cur < a.nplanes) {
            char* plane = a.half + (long long)cur * HPLANE;
            char* rplane = a.real + (long long)cur * RPLANE;
            if (threadIdx.x == 0 && m == 0) {      // the leader draws the team's next plane before it arrives at the first barrier
                const unsigned p = __hip_atomic_fetch_add(ctl + W_NEXT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(team + T_WORD + ((seq + 1) & 1u), ((seq + 1) << 16) | (p < 0xffffu ? p : 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            phase_y<0>(plane, m, a.delay);             // (first touch of the plane by this launch: plain loads)
            ok = team_barrier(team, ++bar * TEAM, ctl, a.flags, sh, (seq + 1) & 0xffffu);
            const unsigned w = sh[1];
            if (!ok) break;
            phase_z<LAUX>(plane, rplane, m, a.delay, !(a.flags & F_NO_REAL));
            ok = team_barrier(team, ++bar * TEAM, ctl, a.flags, sh, 0);
            if (!ok) break;
            phase_y<LAUX>(plane, m, a.delay);
            if ((w >> 16) != ((seq + 1) & 0xffffu)) {     // cannot happen: the word was published before the leader's arrival
                if (threadIdx.x == 0) __hip_atomic_store(ctl + W_ERROR, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            cur = w & 0xffffu; ++seq;
            __syncthreads();       // sh[] is rewritten by the next barrier
        }
    }
    // the last workgroup out clears the control block for the next launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned e = __hip_atomic_fetch_add(ctl + W_EXITED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e == gridDim.x - 1) {
            unsigned keep = poll_load(ctl + W_ERROR);
            for (int i = 0; i < 8; ++i) ctl[6144 + i] = poll_load(ctl + W_TICKET + i * 32);      // statistics of the last launch
            for (int i = 0; i < W_TEAMS + 64 * 64; ++i) __hip_atomic_store(ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl[6144 + 8] = keep;
        }
    }
}

// the same phases as separate launches (one workgroup per tile / per 32 rows)
__global__ __launch_bounds__(NT, 8) void k_y(char* half, int delay) {
    extern __shared__ char smem[];
    phase_y<0>(half + (long long)(blockIdx.x / TEAM) * HPLANE, blockIdx.x % TEAM, delay);
}
__global__ __launch_bounds__(NT, 8) void k_z(char* half, char* real, int delay, int real_out) {
    extern __shared__ char smem[];
    phase_z<0>(half + (long long)(blockIdx.x / TEAM) * HPLANE, real + (long long)(blockIdx.x / TEAM) * RPLANE, blockIdx.x % TEAM, delay, real_out);
}
__global__ void k_fill(float* p, long long n, float v) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 3; } } while (0)

int main(int argc, char** argv) {
    const int P = 512, LDS = 78 * 1024, reps = 6;
    char *half, *real; unsigned* ctl;
    const long long hbytes = HPLANE * P, rbytes = RPLANE * P;
    CK(hipMalloc(&half, hbytes + 4096)); CK(hipMalloc(&real, rbytes)); CK(hipMalloc(&ctl, 8192 * 4));
    CK(hipMemset(ctl, 0, 8192 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_plane_team<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_plane_team<SC1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_y), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_z), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> hh(HPLANE / 4 * 4), hr((size_t)N * N);
    auto fill = [&]() { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (float*)half, hbytes / 4, 1.0f); };
    // check planes 0, 255, 511: every valid column == expect_h, the real plane == expect_r
    auto check = [&](float expect_h, float expect_r, bool real_out) -> long long {
        long long bad = 0;
        for (int pl : {0, 255, 511}) {
            hipMemcpy(hh.data(), half + (long long)pl * HPLANE, HPLANE, hipMemcpyDeviceToHost);
            for (int r = 0; r < N; ++r) for (int c = 0; c < 512; ++c) bad += hh[(size_t)r * (PITCH / 4) + c] != expect_h;
            if (real_out) {
                hipMemcpy(hr.data(), real + (long long)pl * RPLANE, RPLANE, hipMemcpyDeviceToHost);
                for (size_t i = 0; i < hr.size(); ++i) bad += hr[i] != expect_r;
            }
        }
        return bad;
    };
    printf("# tools/plane_team.hip: y -> z -> y of a 512^3 half spectrum (512 planes of 1 MiB + 1 MiB real plane each)\n");
    for (int delay : {0, 6}) {
        // reference: three launches per batch of B planes
        for (int B : {64, 128, 512}) {
            float tot = 0;
            for (int r = 0; r < reps + 1; ++r) {
                fill(); hipMemsetAsync(real, 0, rbytes, 0);
                hipEventRecord(e0);
                for (int p0 = 0; p0 < P; p0 += B) {
                    hipLaunchKernelGGL(k_y, dim3(B * TEAM), dim3(NT), LDS, 0, half + p0 * HPLANE, delay);
                    hipLaunchKernelGGL(k_z, dim3(B * TEAM), dim3(NT), LDS, 0, half + p0 * HPLANE, real + p0 * RPLANE, delay, 1);
                    hipLaunchKernelGGL(k_y, dim3(B * TEAM), dim3(NT), LDS, 0, half + p0 * HPLANE, delay);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (r) tot += ms;
            }
            const long long bad = check(4.f, 3.f, true);
            printf("delay %d  launches, batches of %3d planes: %7.1f us per box   (%5.0f GB/s on 7 half sweeps)  bad %lld\n", delay, B,
                   tot / reps * 1e3, 7.0 * 512 * 512 * 256 * 8 / (tot / reps) / 1e6, bad);
        }
        struct V { const char* name; int flags; };
        const V vs[] = {{"L2 atomics + acquire   ", 0}, {"L2 atomics + sc1 loads ", F_SC1_LOADS}, {"agent atomics + acquire", F_AGENT_ATOMICS},
                        {"agent atomics + sc1 lds", F_AGENT_ATOMICS | F_SC1_LOADS}, {"L2 atomics, NO acquire (unsafe: how often stale?)", F_NO_ACQUIRE},
                        {"L2 atomics + acquire, no real plane", F_NO_REAL},
                        {"NO team barrier at all (wrong results: the transport alone)", F_NO_BARRIER | F_NO_ACQUIRE}};
        for (const V& v : vs) for (int G : {256, 384, 512}) {
            Args a{half, real, ctl, P, v.flags, delay};
            float tot = 0; unsigned stat[16] = {0};
            for (int r = 0; r < reps + 1; ++r) {
                fill(); hipMemsetAsync(real, 0, rbytes, 0);
                hipEventRecord(e0);
                if (v.flags & F_SC1_LOADS) hipLaunchKernelGGL(k_plane_team<SC1>, dim3(G), dim3(NT), LDS, 0, a);
                else hipLaunchKernelGGL(k_plane_team<0>, dim3(G), dim3(NT), LDS, 0, a);
                hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 4; }
                float ms; hipEventElapsedTime(&ms, e0, e1); if (r) tot += ms;
            }
            hipMemcpy(stat, ctl + 6144, sizeof(stat), hipMemcpyDeviceToHost);
            const bool ro = !(v.flags & F_NO_REAL);
            const long long bad = check(4.f, 3.f, ro);
            printf("delay %d  teams, grid %3d, %s: %7.1f us per box  bad %lld  error %u  workgroups per XCD %u %u %u %u %u %u %u %u\n", delay, G, v.name,
                   tot / reps * 1e3, bad, stat[8], stat[0], stat[1], stat[2], stat[3], stat[4], stat[5], stat[6], stat[7]);
            unsigned dbg[8 * 64 + 200]; hipMemcpy(dbg, ctl + 7000, sizeof(dbg), hipMemcpyDeviceToHost);
            if (dbg[0]) {
                printf("  time-outs recorded %u (last launch with an error: its flag is kept)\n", dbg[0]);
                for (unsigned q = 0; q < dbg[0] && q < 24; ++q) { const unsigned* d = dbg + 168 + 8 * q;
                    printf("    team %u target %u count %u member %u word %x xcc %u block %u\n", d[0], d[1], d[2], d[3], d[4], d[5], d[6]); }
                hipMemset(ctl + 7000, 0, 4 * (168 + 512));
            }
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
