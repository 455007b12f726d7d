// Tuning aid (GPU box): how many scalar-ALU instructions does a CU issue per cycle, and does scalar work compete with
// vector work?  A wave runs ITER x 16 independent s_add_u32 (or v_add_u32, or both interleaved); every CU gets W waves.
//   hipcc -O3 --offload-arch=gfx950 tools/salu_rate.hip -o /tmp/salu_rate && /tmp/salu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048;

#define S16 "s_add_u32 s44, s44, s60\n s_add_u32 s45, s45, s60\n s_add_u32 s46, s46, s60\n s_add_u32 s47, s47, s60\n" \
            "s_add_u32 s48, s48, s60\n s_add_u32 s49, s49, s60\n s_add_u32 s50, s50, s60\n s_add_u32 s51, s51, s60\n" \
            "s_add_u32 s52, s52, s60\n s_add_u32 s53, s53, s60\n s_add_u32 s54, s54, s60\n s_add_u32 s55, s55, s60\n" \
            "s_add_u32 s56, s56, s60\n s_add_u32 s57, s57, s60\n s_add_u32 s58, s58, s60\n s_add_u32 s59, s59, s60\n"
#define V16 "v_add_u32 v20, v20, v36\n v_add_u32 v21, v21, v36\n v_add_u32 v22, v22, v36\n v_add_u32 v23, v23, v36\n" \
            "v_add_u32 v24, v24, v36\n v_add_u32 v25, v25, v36\n v_add_u32 v26, v26, v36\n v_add_u32 v27, v27, v36\n" \
            "v_add_u32 v28, v28, v36\n v_add_u32 v29, v29, v36\n v_add_u32 v30, v30, v36\n v_add_u32 v31, v31, v36\n" \
            "v_add_u32 v32, v32, v36\n v_add_u32 v33, v33, v36\n v_add_u32 v34, v34, v36\n v_add_u32 v35, v35, v36\n"
#define SV16 "s_add_u32 s44, s44, s60\n v_add_u32 v20, v20, v36\n s_add_u32 s45, s45, s60\n v_add_u32 v21, v21, v36\n" \
             "s_add_u32 s46, s46, s60\n v_add_u32 v22, v22, v36\n s_add_u32 s47, s47, s60\n v_add_u32 v23, v23, v36\n" \
             "s_add_u32 s48, s48, s60\n v_add_u32 v24, v24, v36\n s_add_u32 s49, s49, s60\n v_add_u32 v25, v25, v36\n" \
             "s_add_u32 s50, s50, s60\n v_add_u32 v26, v26, v36\n s_add_u32 s51, s51, s60\n v_add_u32 v27, v27, v36\n"
#define CLOB "s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60", \
             "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","scc"

#define KERNEL(NAME, BODY)                                                               \
    __global__ __launch_bounds__(1024) void NAME(uint32_t* out) {                        \
        for (int it = 0; it < ITER; ++it) asm volatile(BODY ::: CLOB);                   \
        if (out && threadIdx.x == 12345) out[0] = 1;                                     \
    }
KERNEL(k_s, S16)
KERNEL(k_v, V16)
KERNEL(k_sv, SV16)

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d CUs, clock %d kHz\n", cus, pr.clockRate);
    for (int waves : {4, 8, 16, 32}) {           // waves per CU: blocks of 64*waves' threads... one block per CU
        for (int which = 0; which < 3; ++which) {
            const int threads = waves >= 16 ? 1024 : waves * 64, blocks = cus * (waves >= 16 ? waves / 16 : 1);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(k_s, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)nullptr);
                if (which == 1) hipLaunchKernelGGL(k_v, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)nullptr);
                if (which == 2) hipLaunchKernelGGL(k_sv, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)nullptr);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double instr_per_cu = (double)waves * ITER * 16;       // per kind (sv: 8 of each per 16)
            const double cyc = best * 1e-3 * 2.4e9;                       // at the nominal 2.4 GHz
            printf("%2d waves/CU  %-22s %8.3f ms  -> %.2f cycles (2.4 GHz) per wave-instruction per CU\n", waves,
                   which == 0 ? "16 s_add" : which == 1 ? "16 v_add" : "8 s_add + 8 v_add", best,
                   cyc / (which == 2 ? instr_per_cu : instr_per_cu));
        }
    }
    return 0;
}
