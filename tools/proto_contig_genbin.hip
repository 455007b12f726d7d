// Prototype for the round-2 layout question (DESIGN.md "what 1000/s would take"): how fast are the fused generator
// and the fused binning if they live in the CONTIGUOUS-axis FFT kernel (x as the Hermitian half axis: lines of N
// complex along z, 64 threads per line, 4 lines per 256-thread workgroup, ~20 KB of LDS => 8 workgroups per CU)
// instead of the strided-tile kernel (two 1024-thread workgroups per CU)?  Timing only: same arithmetic per mode
// (Threefry4x32-20, Box-Muller, amplitude from a [|m_x|][|m_y|][k_z] table, 512-point transform; |X|^2 binned with
// wave-uniform two-bin splits), synthetic tables, no parity.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastbox_amd/csrc -I include tools/proto_contig_genbin.hip -o /tmp/proto && /tmp/proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "fb_fft.h"
#include "fb_rng.h"
using namespace fb;

constexpr int N = 512, E = 8, TPL = N / E, LPW = 4, NXH = N / 2 + 1, NZP = 272, M = N / 2 + 1;

__device__ __forceinline__ float wsum(float v) {
#define DPP(ctrl, rm, bm) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rm, bm, false))
    DPP(0x111, 0xf, 0xf); DPP(0x112, 0xf, 0xf); DPP(0x114, 0xf, 0xe); DPP(0x118, 0xf, 0xc); DPP(0x142, 0xa, 0xf); DPP(0x143, 0xc, 0xf);
#undef DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int MODE>   // 0: plain c2c (read + write), 1: generator (write only), 2: binning (read only)
__global__ __launch_bounds__(LPW * TPL) void k_proto(const cx<float>* __restrict__ in, cx<float>* __restrict__ out,
                                                     const cx<float>* __restrict__ tw, const float* __restrict__ sym,
                                                     const int* __restrict__ thr, int nbins, double* __restrict__ partial,
                                                     RngKey key, long long nlines) {
    constexpr int LP = LineLayout<float>::padded(N);
    __shared__ cx<float> lines[LPW * LP];
    __shared__ cx<float> twl[N];
    __shared__ int lthr[64];
    __shared__ double acc[LPW][128];
    const int tid = threadIdx.x, t = tid % TPL, l = tid / TPL;
    for (int i = tid; i < N; i += LPW * TPL) twl[i] = tw[i];
    if (MODE == 2) { for (int i = tid; i < nbins; i += LPW * TPL) lthr[i] = thr[i]; for (int i = tid; i < LPW * 128; i += LPW * TPL) (&acc[0][0])[i] = 0.0; }
    const long long line = (long long)blockIdx.x * LPW + l;
    const int kx = (int)(line / N), ky = (int)(line % N);              // kx in [0, N/2], ky in [0, N)
    const int my = ky < N / 2 ? ky : ky - N, amy = my < 0 ? -my : my;
    LineLayout<float> lay{lines + l * LP};
    cx<float> v[E];
    if (MODE == 1) {
        uint32_t ctr[E / 2][4], rnd[E / 2][4];
#pragma unroll
        for (int j = 0; j < E / 2; ++j) {
            const unsigned long long idx = ((unsigned long long)line * (N / 2)) + t + j * TPL;
            ctr[j][0] = (uint32_t)idx; ctr[j][1] = (uint32_t)(idx >> 32); ctr[j][2] = 0u; ctr[j][3] = 0u;
        }
        threefry4x32_20_batch<E / 2>(ctr, key.k, rnd);
        const float* row = sym + ((long long)kx * M + amy) * NZP;
#pragma unroll
        for (int j = 0; j < E / 2; ++j) {
            const int kz0 = t + j * TPL, kz1 = kz0 + N / 2;             // second half mirrors: |m_z| = N - kz1
            float a0, a1, b0, b1;
            box_muller(rnd[j][0], rnd[j][1], a0, a1);
            box_muller(rnd[j][2], rnd[j][3], b0, b1);
            const float A0 = row[kz0], A1 = row[N - kz1];
            v[j] = cx<float>{A0 * a0, A0 * a1};
            v[j + E / 2] = cx<float>{A1 * b0, A1 * b1};
        }
    } else {
        const cx<float>* src = in + line * N;
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = src[t + e * TPL];
    }
    __syncthreads();
    if (MODE == 2) fft_stages<float, N, E, -1, 1, 1>(v, t, twl, lay);
    else           fft_stages<float, N, E, +1, 1, 1>(v, t, twl, lay);
    if (MODE != 2) {
        cx<float>* dst = out + line * N;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[t + e * TPL] = v[e];
    } else {
        // one wave = one line.  The bins the line can touch follow from its end points (n2row .. n2row + (N/2)^2);
        // a mode's bin = blo + number of those thresholds it has reached; one pair of DPP sums per touched bin.
        const int n2row = kx * kx + my * my;
        const float w = (kx == 0 || kx == N / 2) ? 1.f : 2.f;
        int blo = 0;
        while (blo < nbins && lthr[blo] <= n2row) ++blo;
        int bhi = blo;
        while (bhi < nbins && lthr[bhi] <= n2row + (N / 2) * (N / 2)) ++bhi;
        const int span = bhi - blo;                                    // <= 8 assumed by this prototype
        int tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tv[u] = (u < span) ? lthr[blo + u] : 0x7fffffff;
        float p[E]; int rel[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int kz = t + e * TPL, mz = kz < N / 2 ? kz : kz - N;
            const int n2 = n2row + mz * mz;
            p[e] = w * (v[e].x * v[e].x + v[e].y * v[e].y);
            int r = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) r += (n2 >= tv[u]) ? 1 : 0;
            rel[e] = r;
        }
        for (int u = 0; u <= span; ++u) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) { const bool in = rel[e] == u; s1 += in ? p[e] : 0.f; s2 += in ? p[e] * p[e] : 0.f; }
            s1 = wsum(s1); s2 = wsum(s2);
            if (t == 0 && blo + u < nbins) { acc[l][2 * (blo + u)] += s1; acc[l][2 * (blo + u) + 1] += s2; }
        }
        __syncthreads();
        if (tid < 2 * nbins) partial[(size_t)tid * gridDim.x + blockIdx.x] = acc[0][tid] + acc[1][tid] + acc[2][tid] + acc[3][tid];
    }
}

// mode 3: the geometry of the real z pass -- 256-point lines (2 KB), 32 threads per line, 8 lines per workgroup, row
// pitch 272 complex, one spare row after every 512 lines -- as a plain c2c (read + write), to see what the line
// length alone costs
template <int DUMMY>
__global__ __launch_bounds__(256) void k_proto_short(const cx<float>* __restrict__ in, cx<float>* __restrict__ out,
                                                     const cx<float>* __restrict__ tw) {
    constexpr int NF = 256, TPLs = NF / 8, LPWs = 8, LP = LineLayout<float>::padded(NF);
    __shared__ cx<float> lines[LPWs * LP];
    __shared__ cx<float> twl[NF];
    const int tid = threadIdx.x, t = tid % TPLs, l = tid / TPLs;
    for (int i = tid; i < NF; i += 256) twl[i] = tw[2 * i];
    const long long line = (long long)blockIdx.x * LPWs + l;
    const long long row = line + line / 512;
    LineLayout<float> lay{lines + l * LP};
    cx<float> v[8];
    const cx<float>* src = in + row * 272;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[t + e * TPLs];
    __syncthreads();
    fft_stages<float, NF, 8, +1, 1, 1>(v, t, twl, lay);
    cx<float>* dst = out + row * 272;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[t + e * TPLs] = v[e];
}

int main() {
    const long long nlines = (long long)NXH * N;
    cx<float>*a, *b, *tw; float* sym; int* thr; double* partial;
    // sized for the larger of the two geometries: 257 x 512 lines of 512, or 512 x 513 rows of pitch 272 (mode 3)
    const size_t bytes = (size_t)512 * 513 * 272 * 8 > (size_t)nlines * N * 8 ? (size_t)512 * 513 * 272 * 8 : (size_t)nlines * N * 8;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&tw, N * 8);
    hipMalloc(&sym, (size_t)M * M * NZP * 4); hipMalloc(&thr, 64 * 4); hipMalloc(&partial, (size_t)64 * (nlines / LPW) * 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    std::vector<cx<float>> htw(N);
    for (int i = 0; i < N; ++i) htw[i] = cx<float>{(float)cos(2 * M_PI * i / N), (float)-sin(2 * M_PI * i / N)};
    hipMemcpy(tw, htw.data(), N * 8, hipMemcpyHostToDevice);
    std::vector<float> hs((size_t)M * M * NZP, 1.f);
    hipMemcpy(sym, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> ht(64, 0x7fffffff);
    const int nb = 20; double r = pow(3.0 * 256 * 256, 1.0 / (nb - 1));
    for (int i = 0; i < nb; ++i) ht[i] = (int)pow(r, i);
    hipMemcpy(thr, ht.data(), 64 * 4, hipMemcpyHostToDevice);
    RngKey key{{1, 2, 3, 4}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)(nlines / LPW);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k_proto<0>, dim3(grid), dim3(LPW * TPL), 0, 0, a, b, tw, sym, thr, nb, partial, key, nlines);
                if (mode == 1) hipLaunchKernelGGL(k_proto<1>, dim3(grid), dim3(LPW * TPL), 0, 0, a, b, tw, sym, thr, nb, partial, key, nlines);
                if (mode == 2) hipLaunchKernelGGL(k_proto<2>, dim3(grid), dim3(LPW * TPL), 0, 0, b, a, tw, sym, thr, nb, partial, key, nlines);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms / 10 < best ? ms / 10 : best;
        }
        const double gb = nlines * N * 8.0 / 1e9;
        printf("%s: %.1f us  (%.2f GB %s)\n", mode == 0 ? "contiguous c2c, read+write " : (mode == 1 ? "contiguous generator (write)" : "contiguous binning (read)   "),
               best * 1e3, mode == 0 ? 2 * gb : gb, mode == 0 ? "moved" : (mode == 1 ? "written" : "read"));
    }
    {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it)
                hipLaunchKernelGGL(k_proto_short<0>, dim3(512 * 512 / 8), dim3(256), 0, 0, a, b, tw);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms / 10 < best ? ms / 10 : best;
        }
        printf("z-pass geometry (256-pt lines, pitch 272), plain c2c read+write: %.1f us  (%.2f GB moved)\n", best * 1e3,
               2 * 512.0 * 512 * 256 * 8 / 1e9);
    }
    hipError_t err = hipGetLastError();
    printf("status: %s\n", hipGetErrorString(err));
    return err != hipSuccess;
}
