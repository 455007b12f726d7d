// 3-D FFT passes built from fb_fft.h.  Two kernels:
//   k_fft_strided : c2c along an axis whose lines are strided in memory (x, y).
//                   One workgroup owns a tile of TZ adjacent k_z columns x the
//                   whole line; global traffic is 128-byte row segments.
//   k_fft_contig  : along the contiguous z axis; c2c, r2c (forward) and c2r
//                   (inverse) through the packed half-length complex transform.
#pragma once
#include "fb_fft.h"

namespace fb {

// columns per tile of the strided pass: one 128-byte row segment, shrunk so the
// tile stays within 64 KiB of LDS (two workgroups per CU).
template <typename T> constexpr int tile_cols(int n) {
    return fb_max(2, fb_min(128 / (2 * (int)sizeof(T)), 65536 / (n * 2 * (int)sizeof(T))));
}

template <typename T> struct StridedArgs {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;         // W_N^j
    long long stride;        // elements between consecutive points of a line
    long long outer_stride;  // elements between tiles along blockIdx.y
    int ncols;               // valid contiguous columns
    T scale;
};

template <typename T, int N>
__global__ __launch_bounds__(tile_cols<T>(N) * (N / elems_per_thread(N)))
void k_fft_strided(StridedArgs<T> a, int sign) {
    constexpr int E = elems_per_thread(N);
    constexpr int TPL = N / E;
    constexpr int TZ = tile_cols<T>(N);
    constexpr int NT = TZ * TPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* tile = reinterpret_cast<cx<T>*>(smem);
    cx<T>* twl = tile + N * TZ;

    const int tid = threadIdx.x;
    const int c = tid % TZ;
    const int t = tid / TZ;
    for (int i = tid; i < N; i += NT) twl[i] = a.tw[i];

    const int col = blockIdx.x * TZ + c;
    const bool valid = col < a.ncols;
    const long long base = (long long)blockIdx.y * a.outer_stride + col;

    cx<T> v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (valid) v[e] = a.in[base + (long long)(t + e * TPL) * a.stride];
        else v[e] = cx<T>{0, 0};
    }
    __syncthreads();
    TileLayout<T, TZ> lay{tile, c};
    if (sign < 0) fft_stages<T, N, E, -1, 1, 1>(v, t, twl, lay);
    else          fft_stages<T, N, E, +1, 1, 1>(v, t, twl, lay);
    if (valid) {
#pragma unroll
        for (int e = 0; e < E; ++e)
            a.out[base + (long long)(t + e * TPL) * a.stride] = cscale(v[e], a.scale);
    }
}

enum { ZMODE_C2C = 0, ZMODE_R2C = 1, ZMODE_C2R = 2 };

template <typename T> struct ContigArgs {
    const void* in;
    void* out;
    const cx<T>* tw;        // W_M^j with M = n (c2c) or 2n (r2c / c2r)
    long long in_pitch;     // elements of the input type between lines
    long long out_pitch;    // elements of the output type between lines
    long long nlines;
    T scale;
    int pre_exp;            // r2c: transform exp(x) instead of x (log-normal fusion)
};

template <int NF> constexpr int contig_lines() {   // lines per workgroup
    return fb_max(1, 256 / (NF / elems_per_thread(NF)));
}

// NF = complex transform length (N for c2c, N/2 for r2c/c2r)
template <typename T, int NF, int MODE>
__global__ __launch_bounds__(contig_lines<NF>() * (NF / elems_per_thread(NF)))
void k_fft_contig(ContigArgs<T> a, int sign_c2c) {
    constexpr int E = elems_per_thread(NF);
    constexpr int TPL = NF / E;
    constexpr int LPW = contig_lines<NF>();
    constexpr int NT = LPW * TPL;
    constexpr int TWS = (MODE == ZMODE_C2C) ? 1 : 2;
    constexpr int M = NF * TWS;
    constexpr int LP = LineLayout<T>::padded(NF);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* lines = reinterpret_cast<cx<T>*>(smem);
    cx<T>* twl = lines + LPW * LP;

    const int tid = threadIdx.x;
    const int t = tid % TPL;
    const int l = tid / TPL;
    for (int i = tid; i < M; i += NT) twl[i] = a.tw[i];
    const long long line = (long long)blockIdx.x * LPW + l;
    const bool valid = line < a.nlines;
    LineLayout<T> lay{lines + l * LP};

    cx<T> v[E];
    if constexpr (MODE == ZMODE_C2R) {
        // Z[k] = (X[k] + conj X[n-k]) + i e^{+2 pi i k/N} (X[k] - conj X[n-k]); the
        // imaginary parts of X[0], X[n] are dropped (Hermitian projection).
        const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in) + line * a.in_pitch;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = t + e * TPL;
            cx<T> xk{0, 0}, xn{0, 0};
            if (valid) { xk = in[k]; xn = in[NF - k]; }
            if (k == 0) { xk.y = 0; xn.y = 0; }
            cx<T> s = xk + cconj(xn), d = xk - cconj(xn);
            cx<T> w = cconj(twl[k]);
            cx<T> wd = cmul(w, d);
            v[e] = cx<T>{s.x - wd.y, s.y + wd.x};
        }
        fft_stages<T, NF, E, +1, TWS, 1>(v, t, twl, lay);
        if (valid) {
            cx<T>* out = reinterpret_cast<cx<T>*>(reinterpret_cast<T*>(a.out) + line * a.out_pitch);
#pragma unroll
            for (int e = 0; e < E; ++e) out[t + e * TPL] = cscale(v[e], a.scale);
        }
    } else if constexpr (MODE == ZMODE_R2C) {
        const cx<T>* in = reinterpret_cast<const cx<T>*>(reinterpret_cast<const T*>(a.in) + line * a.in_pitch);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (valid) {
                v[e] = in[t + e * TPL];
                if (a.pre_exp) { v[e].x = exp(v[e].x); v[e].y = exp(v[e].y); }
            } else v[e] = cx<T>{0, 0};
        }
        __syncthreads();
        fft_stages<T, NF, E, -1, TWS, 1>(v, t, twl, lay);
        // untangle: X[k] = (Z[k] + conj Z[n-k])/2 - (i/2) W_N^k (Z[k] - conj Z[n-k])
#pragma unroll
        for (int e = 0; e < E; ++e) lay.at(t + e * TPL) = v[e];
        __syncthreads();
        if (valid) {
            cx<T>* out = reinterpret_cast<cx<T>*>(a.out) + line * a.out_pitch;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int k = t + e * TPL;
                cx<T> zk = v[e];
                cx<T> zn = cconj(lay.at((NF - k) & (NF - 1)));
                cx<T> s = zk + zn, d = zk - zn;
                cx<T> wd = cmul(twl[k], d);
                out[k] = cx<T>{(T)0.5 * (s.x + wd.y) * a.scale, (T)0.5 * (s.y - wd.x) * a.scale};
                if (k == 0) out[NF] = cx<T>{(zk.x - zk.y) * a.scale, (T)0};
            }
        }
    } else {
        const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in) + line * a.in_pitch;
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = valid ? in[t + e * TPL] : cx<T>{0, 0};
        __syncthreads();
        if (sign_c2c < 0) fft_stages<T, NF, E, -1, TWS, 1>(v, t, twl, lay);
        else              fft_stages<T, NF, E, +1, TWS, 1>(v, t, twl, lay);
        if (valid) {
            cx<T>* out = reinterpret_cast<cx<T>*>(a.out) + line * a.out_pitch;
#pragma unroll
            for (int e = 0; e < E; ++e) out[t + e * TPL] = cscale(v[e], a.scale);
        }
    }
}

}  // namespace fb
