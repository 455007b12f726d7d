// Tuning aid: does a strided-pass tile copy run faster when its working set stays in the 256 MiB
// Infinity Cache?  The y-pass tile pattern of tools/stride_copy.hip (512 rows x 128 B per workgroup,
// rows 2176 B apart), in place, over the first P x-planes of a 512^3 half spectrum, launched back to
// back so that from the second launch on the planes are as resident as they can be.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_copy.hip -o /tmp/mall_copy && /tmp/mall_copy
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void k_tile_copy(const char* in, char* out, long long stride, long long outer_stride,
                                                     int seg, int rows, int ntx, int ntiles, int mode) {
    const int lanes_per_row = seg / 8;
    const int r0 = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
    const int rstep = blockDim.x / lanes_per_row;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = (long long)(tile / ntx) * outer_stride + (long long)(tile % ntx) * seg + c * 8;
        for (int r = r0; r < rows; r += rstep * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (mode != 2 && r + u * rstep < rows) ? *(const double*)(in + base + (long long)(r + u * rstep) * stride) : 1.0;
            if (mode == 1) { double acc = 0; for (int u = 0; u < 8; ++u) acc += v[u]; if (acc == 1.2345) out[0] = 1; }
            else {
#pragma unroll
                for (int u = 0; u < 8; ++u) if (r + u * rstep < rows) *(double*)(out + base + (long long)(r + u * rstep) * stride) = v[u];
            }
        }
    }
}

int main() {
    const int N = 512, NZP = 272, seg = 128;
    const long long rowbytes = NZP * 8, plane = (long long)N * rowbytes, bytes = plane * N;
    char *a, *b;
    if (hipMalloc(&a, bytes + (8 << 20)) != hipSuccess || hipMalloc(&b, bytes + (8 << 20)) != hipSuccess) return 1;
    hipMemset(a, 1, bytes + (8 << 20)); hipMemset(b, 0, bytes + (8 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntx = (int)((rowbytes + seg - 1) / seg);
    const int reps = 8;
    for (int mode : {0, 1, 2})
      for (int inplace : {1, 0}) {
        if (mode != 0 && !inplace) continue;
        for (int P : {8, 16, 32, 64, 96, 128, 192, 256, 384, 512}) {
            const int ntiles = ntx * P;                       // highest byte touched < P * plane + one row: inside the slack
            const double moved = (mode == 0 ? 2.0 : 1.0) * (double)ntx * seg * N * P;
            for (int w = 0; w < 2; ++w)
                hipLaunchKernelGGL(k_tile_copy, dim3(512), dim3(1024), 0, 0, a, inplace ? a : b, rowbytes, plane, seg, N, ntx, ntiles, mode);
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(k_tile_copy, dim3(512), dim3(1024), 0, 0, a, inplace ? a : b, rowbytes, plane, seg, N, ntx, ntiles, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("%s %s P %3d planes (%6.1f MB footprint): %7.1f us/launch  %6.0f GB/s  -> %6.1f us per 512 planes\n",
                   mode == 0 ? "copy " : (mode == 1 ? "read " : "write"), inplace ? "in place " : "a -> b   ", P,
                   (inplace ? 1 : 2) * P * plane / 1e6, ms * 1e3, moved / ms / 1e6, ms * 1e3 * 512 / P);
        }
      }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
