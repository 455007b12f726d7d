// FFT passes for grid sizes that are NOT powers of two (round 4): any even N = 2^a 3^b 5^c in 16 .. 1024.
//
// The reference takes any nsamp (numpy.fft: fastbox/box.py:25-26, 187, 193); its own tests, examples and every caller use powers
// of two, and those run through the tuned kernels of fb_fft_kernels.h.  This file is the plain path for the other sizes: ONE
// kernel, Stockham auto-sort with radices 4, 2, 3, 5 chosen at run time, a line's points ping-ponging between two LDS buffers,
// one workgroup per tile of TZ lines -- lines along a strided axis (TZ adjacent columns) or along the contiguous axis (TZ
// consecutive rows), with the real <-> half-complex packing of the z axis (N reals = N/2 complex points + untangle, as
// c2r_line / r2c_line) done in the same kernel.  Nothing is fused into it: the Python layer runs the generator, the filters, the
// log-normal transform and the binning of such a box as separate kernels (they never assumed a power of two).
#pragma once
#include "fb_fft.h"

namespace fb {

struct RadixPlan { int n; int r[16]; };

// radices of an N-point transform: 4s, a 2, 3s, 5s; false if N has another prime factor
inline bool factor_smooth(int N, RadixPlan& rp) {
    rp.n = 0;
    while (N % 4 == 0) { rp.r[rp.n++] = 4; N /= 4; }
    while (N % 2 == 0) { rp.r[rp.n++] = 2; N /= 2; }
    while (N % 3 == 0) { rp.r[rp.n++] = 3; N /= 3; }
    while (N % 5 == 0) { rp.r[rp.n++] = 5; N /= 5; }
    return N == 1 && rp.n <= 16;
}

template <int SIGN, typename T> __device__ __forceinline__ void dft3(cx<T>* u) {
    // w = exp(SIGN 2 pi i / 3) = -1/2 + SIGN i sqrt(3)/2
    const T s = (T)0.86602540378443864676;
    const cx<T> t = u[1] + u[2], d = u[1] - u[2];
    const cx<T> m{u[0].x - (T)0.5 * t.x, u[0].y - (T)0.5 * t.y};
    const cx<T> r = mul_si<SIGN>(cscale(d, s));                  // SIGN i s (u1 - u2)
    u[0] = u[0] + t; u[1] = m + r; u[2] = m - r;
}
template <int SIGN, typename T> __device__ __forceinline__ void dft5(cx<T>* u) {
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;      // cos(2 pi / 5), cos(4 pi / 5)
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;       // sin(2 pi / 5), sin(4 pi / 5)
    const cx<T> a1 = u[1] + u[4], b1 = u[1] - u[4], a2 = u[2] + u[3], b2 = u[2] - u[3];
    const cx<T> m1{u[0].x + c1 * a1.x + c2 * a2.x, u[0].y + c1 * a1.y + c2 * a2.y};
    const cx<T> m2{u[0].x + c2 * a1.x + c1 * a2.x, u[0].y + c2 * a1.y + c1 * a2.y};
    const cx<T> n1 = mul_si<SIGN>(cx<T>{s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y});
    const cx<T> n2 = mul_si<SIGN>(cx<T>{s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y});
    u[0] = u[0] + a1 + a2; u[1] = m1 + n1; u[4] = m1 - n1; u[2] = m2 + n2; u[3] = m2 - n2;
}
// one radix-R butterfly of the Stockham stage with P = product of the radices done: inputs A[(i + q nb)], outputs B[(i - k) R + k + q P]
template <int R, int SIGN, typename T>
__device__ __forceinline__ void generic_butterfly(const cx<T>* A, cx<T>* B, int i, int nb, int P, int TZ, int c, const cx<T>* tw,
                                                  int twstep /* N / (P R) x table stride */) {
    cx<T> u[R];
    const int k = i % P;
#pragma unroll
    for (int q = 0; q < R; ++q) u[q] = A[(i + q * nb) * TZ + c];
    if (P > 1) {
#pragma unroll
        for (int q = 1; q < R; ++q) {
            cx<T> w = tw[q * k * twstep];
            if (SIGN > 0) w.y = -w.y;
            u[q] = cmul(u[q], w);
        }
    }
    if constexpr (R == 2) dft2<SIGN>(u[0], u[1]);
    else if constexpr (R == 3) dft3<SIGN>(u);
    else if constexpr (R == 4) { cx<T> a = u[0], b = u[1], cc = u[2], d = u[3];
        const cx<T> t0 = a + cc, t1 = a - cc, t2 = b + d, t3 = mul_si<SIGN>(b - d);
        u[0] = t0 + t2; u[1] = t1 + t3; u[2] = t0 - t2; u[3] = t1 - t3; }
    else dft5<SIGN>(u);
    const int j = (i - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) B[(j + q * P) * TZ + c] = u[q];
}

enum { GMODE_C2C = 0, GMODE_R2C = 1, GMODE_C2R = 2 };
template <typename T> struct GenericArgs {
    const void* in; void* out;
    const cx<T>* tw;            // W_N^j, j < N (the plan's table)
    int N;                      // points of the plan's lines (table length)
    int n;                      // complex points of THIS transform: N (c2c), N / 2 (r2c / c2r)
    long long pstride_in, pstride_out;     // elements between consecutive points of a line (in units of the side's element type)
    long long cstride_in, cstride_out;     // elements between the lines of a tile
    long long ostride_in, ostride_out;     // elements between tiles along the outer index
    int nlines;                 // lines per outer index (tiles of TZ lines; the last may be short)
    int skip_in, skip_out;      // contiguous passes over a half spectrum: one spare row after every `skip` lines (0: none)
    int sign;
    T scale;
    RadixPlan rp;
};

// grid: (tiles per outer index, outer indices); block: 256 threads; LDS: 2 n TZ cx<T>
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_fft_generic(GenericArgs<T> a, int TZ) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n, tid = threadIdx.x, NT = blockDim.x;
    cx<T>* A = reinterpret_cast<cx<T>*>(smem);
    cx<T>* B = A + (size_t)n * TZ;
    const int line0 = blockIdx.x * TZ;
    const int nl = (a.nlines - line0 < TZ) ? a.nlines - line0 : TZ;
    const long long obase_in = (long long)blockIdx.y * a.ostride_in, obase_out = (long long)blockIdx.y * a.ostride_out;
    auto line_in = [&](int c) -> long long { const long long l = line0 + c; return obase_in + (l + (a.skip_in ? l / a.skip_in : 0)) * a.cstride_in; };
    auto line_out = [&](int c) -> long long { const long long l = line0 + c; return obase_out + (l + (a.skip_out ? l / a.skip_out : 0)) * a.cstride_out; };
    // contiguous lines: consecutive threads take consecutive points; strided lines: consecutive threads take the tile's columns
    const bool pos_fast = a.pstride_in == 1;
    auto split = [&](int idx, int count, int& pos, int& c) {
        if (pos_fast) { pos = idx % count; c = idx / count; } else { c = idx % TZ; pos = idx / TZ; }
    };
    // ---- load (A[pos * TZ + c])
    if constexpr (MODE == GMODE_R2C) {
        const T* src = reinterpret_cast<const T*>(a.in);
        for (int idx = tid; idx < n * TZ; idx += NT) {
            int pos, c; split(idx, n, pos, c);
            cx<T> v{0, 0};
            if (c < nl) { const T* p = src + line_in(c) + 2LL * pos * a.pstride_in; v = cx<T>{p[0], p[a.pstride_in]}; }
            A[pos * TZ + c] = v;
        }
    } else if constexpr (MODE == GMODE_C2R) {
        // Z[k] = (X[k] + conj X[n-k]) + i e^{+2 pi i k / N} (X[k] - conj X[n-k]); Im X[0], Im X[n] dropped (Hermitian projection)
        const cx<T>* src = reinterpret_cast<const cx<T>*>(a.in);
        for (int idx = tid; idx < n * TZ; idx += NT) {
            int k, c; split(idx, n, k, c);
            cx<T> z{0, 0};
            if (c < nl) {
                const long long b = line_in(c);
                cx<T> xk = src[b + (long long)k * a.pstride_in], xn = src[b + (long long)(n - k) * a.pstride_in];
                if (k == 0) { xk.y = 0; xn.y = 0; }
                const cx<T> s = xk + cconj(xn), d = xk - cconj(xn);
                const cx<T> w = cconj(a.tw[k]);                      // e^{+2 pi i k / N}
                const cx<T> wd = cmul(w, d);
                z = cx<T>{s.x - wd.y, s.y + wd.x};
            }
            A[k * TZ + c] = z;
        }
    } else {
        const cx<T>* src = reinterpret_cast<const cx<T>*>(a.in);
        for (int idx = tid; idx < n * TZ; idx += NT) {
            int pos, c; split(idx, n, pos, c);
            A[pos * TZ + c] = c < nl ? src[line_in(c) + (long long)pos * a.pstride_in] : cx<T>{0, 0};
        }
    }
    __syncthreads();
    // ---- stages
    const int twmul_ = a.N / n;                 // table stride: W_n^j = W_N^{j N / n}
    int P = 1;
    for (int st = 0; st < a.rp.n; ++st) {
        const int R = a.rp.r[st], nb = n / R, twstep = (n / (P * R)) * twmul_;
        for (int idx = tid; idx < nb * TZ; idx += NT) {
            const int c = idx % TZ, i = idx / TZ;
            if (a.sign < 0) {
                if (R == 4) generic_butterfly<4, -1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
                else if (R == 2) generic_butterfly<2, -1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
                else if (R == 3) generic_butterfly<3, -1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
                else generic_butterfly<5, -1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
            } else {
                if (R == 4) generic_butterfly<4, +1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
                else if (R == 2) generic_butterfly<2, +1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
                else if (R == 3) generic_butterfly<3, +1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
                else generic_butterfly<5, +1>(A, B, i, nb, P, TZ, c, a.tw, twstep);
            }
        }
        __syncthreads();
        cx<T>* t = A; A = B; B = t;
        P *= R;
    }
    // ---- store (from A)
    if constexpr (MODE == GMODE_C2R) {
        T* dst = reinterpret_cast<T*>(a.out);
        for (int idx = tid; idx < n * TZ; idx += NT) {
            int pos, c; split(idx, n, pos, c);
            if (c < nl) { T* p = dst + line_out(c) + 2LL * pos * a.pstride_out; const cx<T> v = A[pos * TZ + c]; p[0] = v.x * a.scale; p[a.pstride_out] = v.y * a.scale; }
        }
    } else if constexpr (MODE == GMODE_R2C) {
        // X[k] = (Z[k] + conj Z[n-k]) / 2 - (i / 2) W_N^k (Z[k] - conj Z[n-k]), k = 0 .. n (Z[n] = Z[0])
        cx<T>* dst = reinterpret_cast<cx<T>*>(a.out);
        for (int idx = tid; idx < (n + 1) * TZ; idx += NT) {
            int k, c;
            if (pos_fast) { k = idx % (n + 1); c = idx / (n + 1); } else { c = idx % TZ; k = idx / TZ; }
            if (c < nl) {
                const cx<T> zk = A[(k % n) * TZ + c], zn = cconj(A[((n - k) % n) * TZ + c]);
                const cx<T> s = zk + zn, d = zk - zn;
                cx<T> w = k < n ? a.tw[k] : cx<T>{(T)-1, (T)0};         // W_N^{N/2} = -1
                const cx<T> wd = cmul(w, d);
                cx<T> res{(T)0.5 * (s.x + wd.y) * a.scale, (T)0.5 * (s.y - wd.x) * a.scale};
                if (k == 0 || k == n) res.y = 0;
                dst[line_out(c) + (long long)k * a.pstride_out] = res;
            }
        }
    } else {
        cx<T>* dst = reinterpret_cast<cx<T>*>(a.out);
        for (int idx = tid; idx < n * TZ; idx += NT) {
            int pos, c; split(idx, n, pos, c);
            if (c < nl) dst[line_out(c) + (long long)pos * a.pstride_out] = cscale(A[pos * TZ + c], a.scale);
        }
    }
}

}  // namespace fb
