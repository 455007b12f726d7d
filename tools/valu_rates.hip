// Tuning aid (GPU box): issue rates of the vector instructions the noise generator is made of, and the
// rate of whole counter-based generators (Threefry4x32-R, Philox4x32-R), on every CU at 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 512, CH = 8;

#define OPKERNEL(NAME, ASMSTR)                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                   \
        uint32_t x[CH];                                                                           \
        _Pragma("unroll") for (int i = 0; i < CH; ++i) x[i] = seed + threadIdx.x * 7u + i;       \
        uint32_t c = seed | 1u;                                                                   \
        for (int it = 0; it < ITER; ++it) {                                                       \
            _Pragma("unroll") for (int i = 0; i < CH; ++i) asm volatile(ASMSTR : "+v"(x[i]) : "v"(c)); \
        }                                                                                         \
        uint32_t s = 0;                                                                           \
        _Pragma("unroll") for (int i = 0; i < CH; ++i) s ^= x[i];                                \
        if (s == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = s;                            \
    }

OPKERNEL(k_add, "v_add_u32 %0, %0, %1")
OPKERNEL(k_xor, "v_xor_b32 %0, %0, %1")
OPKERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %0, 7")
OPKERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1")
OPKERNEL(k_xad, "v_xad_u32 %0, %0, %1, %1")
OPKERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
OPKERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
OPKERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
OPKERNEL(k_fma, "v_fma_f32 %0, %0, %1, %1")
OPKERNEL(k_log, "v_log_f32 %0, %0")
OPKERNEL(k_sin, "v_sin_f32 %0, %0")
OPKERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
OPKERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")

__global__ __launch_bounds__(256) void k_mad64(uint32_t* out, uint32_t seed) {
    unsigned long long x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = seed + threadIdx.x * 7u + i;
    uint32_t c = seed | 1u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            uint32_t lo = (uint32_t)x[i];
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x[i]) : "v"(lo), "v"(c) : "vcc");
        }
    }
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s ^= x[i];
    if (s == 0x12345678ull) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pkfma(uint32_t* out, uint32_t seed) {
    f2 x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = f2{(float)(seed + i), (float)threadIdx.x};
    f2 c = {1.0001f, 0.9999f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i].x + x[i].y;
    if (s == 0.12345f) out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s;
}

// ---- whole generators: B calls per thread per iteration, 4 x 32 bits each -------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

template <int ROUNDS, int B>
__device__ __forceinline__ void threefry(const uint32_t (&ctr)[B][4], const uint32_t (&key)[4], uint32_t (&X)[B][4]) {
    const uint32_t ks[5] = {key[0], key[1], key[2], key[3], 0x1BD11BDAu ^ key[0] ^ key[1] ^ key[2] ^ key[3]};
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) X[b][i] = ctr[b][i] + ks[i];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if ((r & 1) == 0) {
                X[b][0] += X[b][1]; X[b][1] = rotl32(X[b][1], R[r & 7][0]) ^ X[b][0];
                X[b][2] += X[b][3]; X[b][3] = rotl32(X[b][3], R[r & 7][1]) ^ X[b][2];
            } else {
                X[b][0] += X[b][3]; X[b][3] = rotl32(X[b][3], R[r & 7][0]) ^ X[b][0];
                X[b][2] += X[b][1]; X[b][1] = rotl32(X[b][1], R[r & 7][1]) ^ X[b][2];
            }
            if ((r & 3) == 3) {
                const int s = (r + 1) >> 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) X[b][i] += ks[(s + i) % 5];
                X[b][3] += (uint32_t)s;
            }
        }
    }
}

template <int ROUNDS, int B>
__device__ __forceinline__ void philox(const uint32_t (&ctr)[B][4], const uint32_t (&key)[2], uint32_t (&X)[B][4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) X[b][i] = ctr[b][i];
    uint32_t k0 = key[0], k1 = key[1];          // wave-uniform: scalar registers
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const unsigned long long p0 = (unsigned long long)M0 * X[b][0];
            const unsigned long long p1 = (unsigned long long)M1 * X[b][2];
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ X[b][1] ^ k0;
            const uint32_t y2 = (uint32_t)(p0 >> 32) ^ X[b][3] ^ k1;
            X[b][0] = y0; X[b][1] = (uint32_t)p1; X[b][2] = y2; X[b][3] = (uint32_t)p0;
        }
        k0 += W0; k1 += W1;
    }
}

template <int KIND, int ROUNDS>
__global__ __launch_bounds__(256) void k_gen(uint32_t* out, uint32_t seed, int iters) {
    constexpr int B = 4;
    uint32_t acc = 0;
    const uint32_t key4[4] = {seed, seed ^ 0x55u, 3u, 0u};
    const uint32_t key2[2] = {seed, seed ^ 0x55u};
    for (int it = 0; it < iters; ++it) {
        uint32_t ctr[B][4], X[B][4];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            ctr[b][0] = (blockIdx.x * 256 + threadIdx.x) * 4 + b; ctr[b][1] = it; ctr[b][2] = 0; ctr[b][3] = 0;
        }
        if constexpr (KIND == 0) threefry<ROUNDS, B>(ctr, key4, X);
        else philox<ROUNDS, B>(ctr, key2, X);
#pragma unroll
        for (int b = 0; b < B; ++b) acc ^= X[b][0] ^ X[b][1] ^ X[b][2] ^ X[b][3];
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class F>
static float time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    uint32_t* out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(uint32_t)));
    const int grid = 256 * 8;           // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const double inst_per_simd = (double)ITER * CH * 8;      // wave-instructions one SIMD issues
#define RUN(NAME) do { float ms = time_ms([&] { hipLaunchKernelGGL(NAME, dim3(grid), dim3(256), 0, 0, out, 12345u); }); \
        printf("%-12s %8.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz\n", #NAME, ms, ms * 1e-3 * 2.4e9 / inst_per_simd); } while (0)
    RUN(k_mullo); RUN(k_mulhi); RUN(k_mul24); RUN(k_mad64); RUN(k_log); RUN(k_sin); RUN(k_sqrt); RUN(k_cvt);
    const int iters = 64;
    const double calls = (double)grid * 256 * 4 * iters;
#define GEN(KIND, R, LABEL) do { float ms = time_ms([&] { hipLaunchKernelGGL((k_gen<KIND, R>), dim3(grid), dim3(256), 0, 0, out, 12345u, iters); }); \
        printf("%-16s %8.3f ms  %7.2f G calls/s (4x32 bits each); 33.7 M calls (one 512^3 box) = %6.1f us\n", LABEL, ms, calls / ms * 1e-6, 33.7e6 / (calls / ms * 1e3) * 1e6); } while (0)
    GEN(0, 20, "threefry4x32-20"); GEN(0, 12, "threefry4x32-12"); GEN(1, 10, "philox4x32-10"); GEN(1, 7, "philox4x32-7");
    return 0;
}
