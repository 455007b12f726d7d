// 3-D FFT passes built from fb_fft.h.  Two kernels:
//   k_fft_strided : c2c along an axis whose lines are strided in memory (x, y).
//                   One workgroup owns a tile of TZ adjacent k_z columns x the
//                   whole line; global traffic is 128-byte row segments.
//                   MODE GEN  fuses the Gaussian-field generator (box.py:161-176) into
//                             the loads of the first inverse pass (nothing is read);
//                   MODE BIN  fuses |delta_k|^2 shell binning (box.py:741-764) into the
//                             stores of the last forward pass (nothing need be written).
//   k_fft_contig  : along the contiguous z axis; c2c, r2c (forward, optional exp() on
//                   load for the log-normal transform) and c2r (inverse) through the
//                   packed half-length complex transform.
#pragma once
#include "fb_fft.h"
#include "fb_field_kernels.h"

namespace fb {

// columns per tile of the strided pass: one 128-byte row segment, shrunk so the
// tile stays within 64 KiB of LDS (two workgroups per CU).
template <typename T> constexpr int tile_cols(int n) {
    // ... and widened for tiny grids so that a workgroup is at least one full wave
    return fb_max(fb_max(2, fb_min(128 / (2 * (int)sizeof(T)), 65536 / (n * 2 * (int)sizeof(T)))),
                  64 / (n / elems_per_thread(n)));
}

enum { SMODE_PLAIN = 0, SMODE_GEN = 1, SMODE_BIN = 2 };

template <typename T> struct StridedArgs {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;         // W_N^j
    long long stride;        // elements between consecutive points of a line
    long long outer_stride;  // elements between tiles along blockIdx.y
    int ncols;               // valid contiguous columns
    T scale;
};

// operands of the fused modes (x pass of a half spectrum: line index = k_x,
// blockIdx.y = k_y, column = k_z)
template <typename T> struct StridedOp {
    KGeom g;
    AmpSrc<T> amp;       // GEN
    RngKey key;          // GEN
    const int* thr;      // BIN: shell thresholds (cubic boxes only)
    const double* bins;  // BIN: edges, for the shells listed in amb[]
    double* partial;     // BIN: [gridDim.y * gridDim.x][2 * nbins]
    int nbins, namb, store;
    int amb[8];
};

__device__ __forceinline__ int shell_bin(const int* lthr, int nbins, int n2) {
    int lo = 0, hi = nbins;                       // number of thr[] <= n2
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (lthr[mid] <= n2) lo = mid + 1; else hi = mid; }
    return lo;
}

// every lane of the wave calls this; lanes with `have` add (s1, s2) to bin b of the wave's row
__device__ __forceinline__ void wave_flush(int b, double s1, double s2, bool have, double* row) {
    unsigned long long rem = __ballot(have);
    const int lane = threadIdx.x & 63;
    while (rem) {
        const int lead = __ffsll((long long)rem) - 1;
        const int bl = __shfl(b, lead, 64);
        const bool mine = have && b == bl;
        const double r1 = wave_sum(mine ? s1 : 0.0), r2 = wave_sum(mine ? s2 : 0.0);
        if (lane == 0) { row[2 * bl] += r1; row[2 * bl + 1] += r2; }
        rem &= ~__ballot(mine);
    }
}

template <typename T, int N, int MODE>
__global__ __launch_bounds__(tile_cols<T>(N) * (N / elems_per_thread(N)))
void k_fft_strided(StridedArgs<T> a, int sign, StridedOp<T> op) {
    constexpr int E = elems_per_thread(N);
    constexpr int TPL = N / E;
    constexpr int TZ = tile_cols<T>(N);
    constexpr int NT = TZ * TPL;
    constexpr int NW = (NT + 63) / 64;
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
    if constexpr (MODE == SMODE_GEN) {
        // generator mode k_x = t + j TPL (< N/2) also serves k_x + N/2 (fb_rng.h)
        const int ky = blockIdx.y;
        const int my = mode_of(ky, N);
        const int c2 = my * my + col * col;            // col = k_z <= N/2 is its own mode number
        const T pf = plane_factor<T>(col, N);
#pragma unroll
        for (int j = 0; j < E / 2; ++j) {
            const int kx = t + j * TPL;
            if (valid) {
                const unsigned long long idx = ((unsigned long long)kx * N + ky) * op.g.NZV + col;
                T a0, a1, b0, b1;
                mode_noise_pair<T>(idx, 0u, op.key, a0, a1, b0, b1);
                T A0, A1;
                if (op.amp.shell) {
                    const int mh = kx - (N >> 1);
                    A0 = op.amp.shell[kx * kx + c2] * pf;
                    A1 = op.amp.shell[mh * mh + c2] * pf;
                } else {
                    A0 = op.amp.dense[((long long)kx * N + ky) * op.g.NZP + col] * pf;
                    A1 = op.amp.dense[((long long)(kx + (N >> 1)) * N + ky) * op.g.NZP + col] * pf;
                }
                v[j] = cx<T>{A0 * a0, A0 * a1};
                v[j + E / 2] = cx<T>{A1 * b0, A1 * b1};
            } else {
                v[j] = cx<T>{0, 0};
                v[j + E / 2] = cx<T>{0, 0};
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (valid) v[e] = a.in[base + (long long)(t + e * TPL) * a.stride];
            else v[e] = cx<T>{0, 0};
        }
    }
    __syncthreads();
    TileLayout<T, TZ> lay{tile, c};
    if constexpr (MODE == SMODE_GEN) fft_stages<T, N, E, +1, 1, 1>(v, t, twl, lay);
    else if constexpr (MODE == SMODE_BIN) fft_stages<T, N, E, -1, 1, 1>(v, t, twl, lay);
    else {
        if (sign < 0) fft_stages<T, N, E, -1, 1, 1>(v, t, twl, lay);
        else          fft_stages<T, N, E, +1, 1, 1>(v, t, twl, lay);
    }
    if (MODE != SMODE_BIN || op.store) {
        if (valid) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                a.out[base + (long long)(t + e * TPL) * a.stride] = cscale(v[e], a.scale);
        }
    }
    if constexpr (MODE == SMODE_BIN) {
        // tile[] is free: every wave passed the barrier that ended the last exchange
        const int nb = op.nbins;
        double* acc = reinterpret_cast<double*>(smem);             // [NW][2 nb]
        int* lthr = reinterpret_cast<int*>(acc + (size_t)NW * 2 * nb);
        __syncthreads();
        for (int i = tid; i < NW * 2 * nb; i += NT) acc[i] = 0.0;
        for (int i = tid; i < nb; i += NT) lthr[i] = op.thr[i];
        __syncthreads();
        double* row = acc + (size_t)(tid >> 6) * 2 * nb;
        const int ky = blockIdx.y;
        const int my = mode_of(ky, N);
        const int c2 = my * my + col * col;
        const double w = (col == 0 || col == (N >> 1)) ? 1.0 : 2.0;
        // |m_x| is smallest for e = 0 or E-1 and largest for e = E/2-1 or E/2
        const int mlo = t < TPL - t ? t : TPL - t;
        const int mhi_a = t + (E / 2 - 1) * TPL, mhi_b = N - (t + (E / 2) * TPL);
        const int mhi = mhi_a > mhi_b ? mhi_a : mhi_b;
        const int n2lo = mlo * mlo + c2, n2hi = mhi * mhi + c2;
        const int blo = shell_bin(lthr, nb, n2lo), bhi = shell_bin(lthr, nb, n2hi);
        bool hit = false;
        for (int q = 0; q < op.namb; ++q) hit |= (op.amb[q] >= n2lo && op.amb[q] <= n2hi);
        if (__all(!valid || (blo == bhi && !hit))) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double p = (double)(v[e].x * v[e].x + v[e].y * v[e].y);
                s1 += p; s2 += p * p;
            }
            wave_flush(blo, w * s1, w * s2, valid && blo < nb, row);
        } else {
            int cur = -1;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int e = (q & 1) ? E - 1 - (q >> 1) : (q >> 1);      // ascending |m_x|
                const int kx = t + e * TPL;
                const int mx = mode_of(kx, N);
                const int n2 = mx * mx + c2;
                int b = shell_bin(lthr, nb, n2);
                for (int z = 0; z < op.namb; ++z)
                    if (op.amb[z] == n2) b = bin_exact(op.bins, nb, kmag_exact(op.g, kx, ky, col));
                if (!valid) b = cur;
                if (__any(b != cur)) {
                    wave_flush(cur, w * s1, w * s2, valid && cur >= 0 && cur < nb, row);
                    s1 = 0.0; s2 = 0.0; cur = b;
                }
                const double p = (double)(v[e].x * v[e].x + v[e].y * v[e].y);
                s1 += p; s2 += p * p;
            }
            wave_flush(cur, w * s1, w * s2, valid && cur >= 0 && cur < nb, row);
        }
        __syncthreads();
        double* dst = op.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * nb;
        for (int i = tid; i < 2 * nb; i += NT) {
            double s = 0.0;
            for (int wv = 0; wv < NW; ++wv) s += acc[(size_t)wv * 2 * nb + i];
            dst[i] = s;
        }
    }
}

// out[q] = sum_r partial[r][q], fixed order: one workgroup per column
static __global__ __launch_bounds__(256) void k_sum_columns(const double* __restrict__ partial, long long nrows,
                                                             int nvals, double* __restrict__ out) {
    __shared__ double sh[256];
    const int q = blockIdx.x;
    double s = 0.0;
    for (long long r = threadIdx.x; r < nrows; r += 256) s += partial[r * nvals + q];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = sh[0];
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
    double* exp_partial;    // r2c + pre_exp: [gridDim.x] block sums of exp(x)
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
        double esum = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (valid) {
                v[e] = in[t + e * TPL];
                if (a.pre_exp) { v[e].x = exp(v[e].x); v[e].y = exp(v[e].y); esum += (double)v[e].x + (double)v[e].y; }
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
        if (a.pre_exp) {                       // wave-uniform
            __syncthreads();                   // lines[] no longer needed
            double* red = reinterpret_cast<double*>(smem);
            const double ws = wave_sum(esum);
            if ((tid & 63) == 0) red[tid >> 6] = ws;
            __syncthreads();
            if (tid == 0) {
                double s = 0.0;
                for (int q = 0; q < (NT + 63) / 64; ++q) s += red[q];
                a.exp_partial[blockIdx.x] = s;
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
