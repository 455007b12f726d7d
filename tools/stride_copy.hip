// Tuning aid: how fast can MI355X stream a 512^3 half spectrum when every workgroup moves a
// tile of ROWS rows x SEG bytes whose rows are `stride` bytes apart (the access pattern of a
// strided FFT pass), as a function of the contiguous segment length SEG?
//   hipcc --offload-arch=gfx950 -O3 tools/stride_copy.hip -o /tmp/stride_copy && /tmp/stride_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VEC>   // bytes per lane per access: 8 or 16
__global__ __launch_bounds__(1024) void k_tile_copy(const char* __restrict__ in, char* __restrict__ out, long long stride,
                                                     long long outer_stride, int seg, int rows, int ntx, int ntiles, int mode) {
    const int lanes_per_row = seg / VEC;
    const int r0 = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
    const int rstep = blockDim.x / lanes_per_row;
    if (r0 >= rstep) return;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = (long long)(tile / ntx) * outer_stride + (long long)(tile % ntx) * seg + c * VEC;
        for (int r = r0; r < rows; r += rstep * 8) {
            if (VEC == 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (mode != 2 && r + u * rstep < rows) ? *(const double*)(in + base + (long long)(r + u * rstep) * stride) : 1.0;
                if (mode == 1) { double acc = 0; for (int u = 0; u < 8; ++u) acc += v[u]; if (acc == 1.2345) out[0] = 1; }
                else {
#pragma unroll
                for (int u = 0; u < 8; ++u) if (r + u * rstep < rows) *(double*)(out + base + (long long)(r + u * rstep) * stride) = v[u];
                }
            } else {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (r + u * rstep < rows) ? *(const double2*)(in + base + (long long)(r + u * rstep) * stride) : double2{0, 0};
#pragma unroll
                for (int u = 0; u < 8; ++u) if (r + u * rstep < rows) *(double2*)(out + base + (long long)(r + u * rstep) * stride) = v[u];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    const long long step = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
        float4 a = in[i], b = in[i + step], c = in[i + 2 * step], d = in[i + 3 * step];
        out[i] = a; out[i + step] = b; out[i + 2 * step] = c; out[i + 3 * step] = d;
    }
    for (; i < n; i += step) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read4(const float4* __restrict__ in, float* __restrict__ out, long long n) {
    const long long step = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) { float4 a = in[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write4(float4* __restrict__ out, long long n) {
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) out[i] = float4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    const int N = 512, NZP = 272;
    const long long bytes = (long long)N * N * NZP * 8;
    char *a, *b;
    hipMalloc(&a, bytes + (8 << 20)); hipMalloc(&b, bytes + (8 << 20));   // slack: the last tile of a row may overrun by < 1 row
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rowbytes = NZP * 8;                  // 2176
    struct Cfg { const char* name; long long stride, outer; int nouter; } cfgs[4] = {
        {"y-like (stride 2176 B)", rowbytes, (long long)N * rowbytes, N},
        {"x-like (stride 1.1 MB)", (long long)N * rowbytes, rowbytes, N},
        {"x-like (plane + 1 row)", (long long)(N + 1) * rowbytes, rowbytes, N},
        {"x-like (plane + 256 B)", (long long)N * rowbytes + 256, rowbytes, N}};
    for (auto& cf : cfgs)
      for (int mode : {0, 1, 2})
        for (int seg : {128, 256})
            for (int vec : {8}) {
                if (seg / vec > 1024 || seg % vec) continue;
                const int ntx = (rowbytes + seg - 1) / seg;     // last tile overruns into padding: fine for timing
                const int ntiles = ntx * cf.nouter;
                const double moved = 2.0 * (double)ntx * seg * N * cf.nouter;
                for (int blocks : {256, 512}) {
                    float best = 1e9f;
                    for (int rep = 0; rep < 5; ++rep) {
                        hipEventRecord(e0);
                        if (vec == 8) hipLaunchKernelGGL(k_tile_copy<8>, dim3(blocks), dim3(1024), 0, 0, a, b, cf.stride, cf.outer, seg, N, ntx, ntiles, mode);
                        else hipLaunchKernelGGL(k_tile_copy<16>, dim3(blocks), dim3(1024), 0, 0, a, b, cf.stride, cf.outer, seg, N, ntx, ntiles, mode);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        if (ms < best) best = ms;
                    }
                    printf("%s %s seg %4d B  %2d B/lane  %4d blocks: %7.1f us  %6.0f GB/s\n", cf.name, mode == 0 ? "copy " : (mode == 1 ? "read " : "write"), seg, vec, blocks,
                           best * 1e3, (mode == 0 ? moved : moved / 2) / best / 1e6);
                }
            }
    // reference: plain contiguous float4 streams over the same buffers
    const long long n4 = bytes / 16;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        float bc = 1e9f, br = 1e9f, bw = 1e9f, ms;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < bc) bc = ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float*)b, n4);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < br) br = ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_write4, dim3(blocks), dim3(256), 0, 0, (float4*)b, n4);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < bw) bw = ms;
        }
        printf("contiguous float4, %5d blocks: copy %6.1f us %5.0f GB/s | read %6.1f us %5.0f GB/s | write %6.1f us %5.0f GB/s\n",
               blocks, bc * 1e3, 2.0 * bytes / bc / 1e6, br * 1e3, bytes / br / 1e6, bw * 1e3, bytes / bw / 1e6);
    }
    return 0;
}
