// Element-wise k-space / real-space kernels of the density-field hot path.
// Every kernel cites the reference expression (fastbox/box.py) it replaces.
//
// Layouts (z fastest, C order like the reference's numpy arrays):
//   real  : T      [N][N][N]
//   half  : cx<T>  [N][NR][NZP]  k_z = 0..N/2 stored (NZV = N/2+1), row pitch NZP, NR = N+1 rows
//                                per x-plane of which N are used
//   full  : cx<T>  [N][N][N]
#pragma once
#include "fb_fft.h"
#include "fb_rng.h"

namespace fb {

#define FB_TWO_PI 6.283185307179586

__device__ __forceinline__ int mode_of(int i, int N) { return i < (N >> 1) ? i : i - N; }   // box.py:119
__device__ __forceinline__ int mirror_of(int i, int N) { return i == 0 ? 0 : N - i; }

// Geometry tables computed on the host with the reference's own numpy
// expressions, so that |k| is bit-identical to box.py:125-127.
struct KGeom {
    const double* axis2;   // [3][N]  (m/L_a)**2                 box.py:125-127
    const double* ksc;     // [3][N]  m * (2 pi / L_a)            box.py:254-256
    const double* kpar;    // [N]     2 pi m / Lz                 box.py:375
    int N, NZV, NZP;
    int NR;                // rows per x-plane of a half spectrum as stored: N + 1 (the spare row
                           // keeps the x stride off a multiple of 64 KiB, which would put every row of
                           // a tile on the same HBM channel: tools/stride_copy.hip, 265 -> 222 us)
};

__device__ __forceinline__ void fb_sincospi(float x, float* s, float* c) { sincospif(x, s, c); }
__device__ __forceinline__ void fb_sincospi(double x, double* s, double* c) { sincospi(x, s, c); }
__device__ __forceinline__ void fb_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void fb_sincos(double x, double* s, double* c) { sincos(x, s, c); }

__device__ __forceinline__ double kmag_exact(const KGeom& g, int i, int j, int l) {
#pragma clang fp contract(off)
    double s = (g.axis2[i] + g.axis2[g.N + j]) + g.axis2[2 * g.N + l];
    return FB_TWO_PI * sqrt(s);
}
__device__ __forceinline__ double kperp_exact(const KGeom& g, int i, int j) {   // box.py:374
#pragma clang fp contract(off)
    double s = g.axis2[i] + g.axis2[g.N + j];
    return FB_TWO_PI * sqrt(s);
}
__device__ __forceinline__ int shell_of(int i, int j, int l, int N) {
    int a = mode_of(i, N), b = mode_of(j, N), c = mode_of(l, N);
    return a * a + b * b + c * c;
}

template <typename T> __device__ __forceinline__ T nan_to_num(T v) {   // np.nan_to_num defaults
    if (v != v) return (T)0;
    if (sizeof(T) == 4) { if (v > (T)3.4028234663852886e38) return (T)3.4028234663852886e38;
                          if (v < (T)-3.4028234663852886e38) return (T)-3.4028234663852886e38; }
    else { if (v > (T)1.7976931348623157e308) return (T)1.7976931348623157e308;
           if (v < (T)-1.7976931348623157e308) return (T)-1.7976931348623157e308; }
    return v;
}

// ---- sqrt(P(k) boxfactor) source ------------------------------------------------------
template <typename T> struct AmpSrc {
    const T* shell;   // [n^2] for cubic boxes, or
    const T* dense;   // [N][N][NZP] evaluated per stored mode on the host
    const T* sym;     // [N/2+1][N/2+1][NZP] = shell[a^2 + b^2 + c^2]: the shell table spread over
                      // (|m_x|, |m_y|, k_z) so that the fused generator reads it in k_z-contiguous runs
};
// A 64-lane gather from the n^2-indexed shell table touches ~64 cache lines (neighbouring k_z are
// 2 k_z + 1 entries apart) and costs the generator pass more than its random numbers; the spread
// table is read like the data itself, 16 consecutive k_z per row.  Built once per amplitude table.
template <typename T>
__global__ void k_build_amp_sym(const T* __restrict__ shell, T* __restrict__ sym, int N, int NZP) {
    const int M = (N >> 1) + 1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y, a = blockIdx.z;
    if (c >= NZP) return;
    sym[((long long)a * M + b) * NZP + c] = c < M ? shell[a * a + b * b + c * c] : (T)0;
}
template <typename T>
__device__ __forceinline__ T amp_at(const AmpSrc<T>& a, const KGeom& g, int i, int j, int l) {
    if (a.shell) return a.shell[shell_of(i, j, l, g.N)];
    if (a.sym) {
        const int mi = mode_of(i, g.N), mj = mode_of(j, g.N), ml = mode_of(l, g.N);
        const int M = (g.N >> 1) + 1;
        return a.sym[((long long)(mi < 0 ? -mi : mi) * M + (mj < 0 ? -mj : mj)) * g.NZP + (ml < 0 ? -ml : ml)];
    }
    return a.dense[((long long)i * g.NR + j) * g.NZP + l];
}

// ---- Gaussian field, parity mode ---------------------------------------------------------
// box.py:174-187: X = (re + i im) sqrt(pk); delta_x = Re ifftn(X).  Re ifftn(X) is
// ifftn of the Hermitian part (X(k) + conj X(-k))/2, which is what is stored here.
template <typename T>
__global__ void k_colour_noise(const T* __restrict__ re, const T* __restrict__ im, cx<T>* __restrict__ out,
                               AmpSrc<T> amp, KGeom g) {
    const int N = g.N;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = blockIdx.z;
    if (l >= g.NZV) return;
    const long long a = ((long long)i * N + j) * N + l;
    const long long b = ((long long)mirror_of(i, N) * N + mirror_of(j, N)) * N + mirror_of(l, N);
    const T A = (T)0.5 * amp_at(amp, g, i, j, l);
    out[((long long)i * g.NR + j) * g.NZP + l] = cx<T>{A * (re[a] + re[b]), A * (im[a] - im[b])};
}

// ---- Gaussian field, throughput mode (counter-based RNG, fb_rng.h) ------------------------------
// Statistically identical to box.py:174-187 followed by the Hermitian projection that Re ifftn() applies:
// a stored mode gets A z with E |z|^2 = 1, and the self-mirrored planes k_z = 0, N/2 are drawn Hermitian
// (fb_rng.h mode_noise), so the half spectrum IS the transform of a real field.
template <typename T>
__global__ void k_colour_device(cx<T>* __restrict__ out, AmpSrc<T> amp, KGeom g, RngKey key) {
    const int N = g.N;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = blockIdx.z;
    if (l >= g.NZV) return;
    T re, im;
    mode_noise<T>(i, j, l, N, g.NZV, key, amp_at(amp, g, i, j, l), re, im);
    out[((long long)i * g.NR + j) * g.NZP + l] = cx<T>{re, im};
}

// ---- shell binning of |delta_k|^2 --------------------------------------------------------------
// box.py:741-764.  idx = np.digitize(k, bins) = number of edges <= k.  For cubic boxes the
// bin is a step function of the integer shell n^2, given as thresholds thr[b] = first
// n^2 whose |k| reaches edge b; the (rare) shells whose |k| is within rounding of an
// edge are listed in amb[] and decided with the exact fp64 expression.
struct BinGeom {
    const double* bins;   // [nbins] edges, ascending
    const int* thr;       // [nbins] or null (non-cubic: always exact)
    int nbins;
    int namb;
    int amb[8];
};
#define FB_BIN_WAVES 4

__device__ __forceinline__ int bin_exact(const double* bins, int nbins, double k) {
    int b = 0;
    for (int q = 0; q < nbins; ++q) b += (bins[q] <= k) ? 1 : 0;
    return b;
}
__device__ __forceinline__ int bin_of_mode(const BinGeom& bg, const KGeom& g, const double* lbins,
                                           const int* lthr, int i, int j, int l) {
    if (bg.thr) {
        const int n2 = shell_of(i, j, l, g.N);
        bool amb = false;
        for (int q = 0; q < bg.namb; ++q) amb |= (bg.amb[q] == n2);
        if (!amb) {
            int lo = 0, hi = bg.nbins;          // count of thr[] <= n2 (thr ascending)
            while (lo < hi) { int mid = (lo + hi) >> 1; if (lthr[mid] <= n2) lo = mid + 1; else hi = mid; }
            return lo;
        }
    }
    return bin_exact(lbins, bg.nbins, kmag_exact(g, i, j, l));
}

// out[q] = sum over workgroups of partial[q][workgroup]; fixed order (thread-strided chains, LDS tree)
static __global__ __launch_bounds__(256) void k_bin_finish(const double* __restrict__ partial, int nblocks, int nvals,
                                                           double* __restrict__ out) {
    __shared__ double sh[256];
    const int q = blockIdx.x;
    const double* src = partial + (size_t)q * nblocks;
    double c[4] = {0, 0, 0, 0};
    int r = threadIdx.x;
    for (; r + 3 * 256 < nblocks; r += 4 * 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] += src[r + u * 256];
    }
    for (int u = 0; r < nblocks; r += 256, ++u) c[u & 3] += src[r];
    sh[threadIdx.x] = (c[0] + c[1]) + (c[2] + c[3]);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = sh[0];
}

// number of full-grid modes per bin (data independent; done once per bin set)
static __global__ void k_bin_count(unsigned long long* __restrict__ counts, KGeom g, BinGeom bg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lbins = reinterpret_cast<double*>(smem);
    unsigned long long* lc = reinterpret_cast<unsigned long long*>(lbins + bg.nbins);
    int* lthr = reinterpret_cast<int*>(lc + bg.nbins);
    const int tid = threadIdx.x;
    for (int q = tid; q < bg.nbins; q += blockDim.x) { lbins[q] = bg.bins[q]; lc[q] = 0; lthr[q] = bg.thr ? bg.thr[q] : 0; }
    __syncthreads();
    const long long nrows = (long long)g.N * g.N;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int i = (int)(row / g.N), j = (int)(row % g.N);
        for (int l = tid; l < g.NZV; l += blockDim.x) {
            const int b = bin_of_mode(bg, g, lbins, lthr, i, j, l);
            if (b < bg.nbins) atomicAdd(&lc[b], (l == 0 || l == (g.N >> 1)) ? 1ull : 2ull);
        }
    }
    __syncthreads();
    for (int q = tid; q < bg.nbins; q += blockDim.x) if (lc[q]) atomicAdd(&counts[q], lc[q]);
}

// ---- reductions ---------------------------------------------------------------------------------
// Sum over the 64 lanes, returned in every lane.  float: six DPP adds (row_shr 1, 2, 4, 8, then
// row_bcast 15 and 31 carry the row totals forward; lanes a step does not reach add the `old`
// operand 0), total in lane 63, one v_readlane -- no LDS crossbar round trips.
__device__ __forceinline__ float wave_sum(float v) {
#define FB_DPP_ADD(ctrl, rmask, bmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, bmask, false))
    FB_DPP_ADD(0x111, 0xf, 0xf);     // row_shr:1
    FB_DPP_ADD(0x112, 0xf, 0xf);     // row_shr:2
    FB_DPP_ADD(0x114, 0xf, 0xe);     // row_shr:4, lanes 4..15 of each row
    FB_DPP_ADD(0x118, 0xf, 0xc);     // row_shr:8, lanes 8..15 of each row
    FB_DPP_ADD(0x142, 0xa, 0xf);     // row_bcast:15 into rows 1 and 3
    FB_DPP_ADD(0x143, 0xc, 0xf);     // row_bcast:31 into rows 2 and 3
#undef FB_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// min / max of an int over the wave (same DPP ladder as wave_sum; `old` = the identity element)
template <bool MAX>
__device__ __forceinline__ int wave_minmax(int v) {
    constexpr int ident = MAX ? (int)0x80000000 : 0x7fffffff;
#define FB_DPP_MM(ctrl, rmask, bmask) do { \
        const int o = __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, bmask, false); \
        v = MAX ? (o > v ? o : v) : (o < v ? o : v); } while (0)
    FB_DPP_MM(0x111, 0xf, 0xf);
    FB_DPP_MM(0x112, 0xf, 0xf);
    FB_DPP_MM(0x114, 0xf, 0xe);
    FB_DPP_MM(0x118, 0xf, 0xc);
    FB_DPP_MM(0x142, 0xa, 0xf);
    FB_DPP_MM(0x143, 0xc, 0xf);
#undef FB_DPP_MM
    return __builtin_amdgcn_readlane(v, 63);
}

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

// the same with the wave's sums formed in single precision (six DPP adds instead of twelve cross-lane moves)
__device__ __forceinline__ void wave_flush(int b, float s1, float s2, bool have, double* row) {
    unsigned long long rem = __ballot(have);
    const int lane = threadIdx.x & 63;
    while (rem) {
        const int lead = __ffsll((long long)rem) - 1;
        const int bl = __shfl(b, lead, 64);
        const bool mine = have && b == bl;
        const float r1 = wave_sum(mine ? s1 : 0.f), r2 = wave_sum(mine ? s2 : 0.f);
        if (lane == 0) { row[2 * bl] += (double)r1; row[2 * bl + 1] += (double)r2; }
        rem &= ~__ballot(mine);
    }
}

// block partial sums of f(x); OP 0: x, 1: x^2, 2: exp(x - shift) (also written to out), 3: block maxima of x instead of sums
template <typename T, int OP>
__global__ __launch_bounds__(256) void k_reduce_real(const T* __restrict__ in, T* __restrict__ out, long long n,
                                                      double* __restrict__ partial, double shift = 0) {
    __shared__ double ws[4];
    double s = OP == 3 ? -1.7976931348623157e308 : 0.0;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
        T x = in[q];
        if (OP == 1) x = x * x;
        // (difference and exponential in double: rounded to single precision, x - shift ~ -100 would carry 4e-6 of
        // absolute error into the exponent -- 40 times the 1e-7 that exp(x) of the stored value has; streaming kernel)
        if (OP == 2) { x = (T)exp((double)x - shift); out[q] = x; }
        if (OP == 3) s = (double)x > s ? (double)x : s;        // (a NaN never wins: the maximum of the finite values)
        else s += (double)x;
    }
    if (OP == 3) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(s, o, 64); s = t > s ? t : s; }
    } else s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (OP == 3) { const double a = ws[0] > ws[1] ? ws[0] : ws[1], b = ws[2] > ws[3] ? ws[2] : ws[3]; partial[blockIdx.x] = a > b ? a : b; }
        else partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    }
}
// sum over the FULL grid of |delta_k|^2 from a half spectrum (box.py:946)
template <typename T>
__global__ __launch_bounds__(256) void k_sumsq_half(const cx<T>* __restrict__ half, KGeom g, double* __restrict__ partial) {
    __shared__ double ws[4];
    double s = 0.0;
    const long long nrows = (long long)g.N * g.N;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x)
        for (int l = threadIdx.x; l < g.NZV; l += blockDim.x) {
            const cx<T> d = half[((row / g.N) * g.NR + row % g.N) * g.NZP + l];
            const double p = (double)d.x * d.x + (double)d.y * d.y;
            s += (l == 0 || l == (g.N >> 1)) ? p : 2.0 * p;
        }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
// y = a x + b   (log-normal finish: exp(d)/mean - 1, box.py:458-459)
template <typename T>
__global__ void k_affine(T* __restrict__ x, long long n, T a, T b) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x)
        x[q] = x[q] * a + b;
}

// out = a x + b y + c over a real cube (y may be null): the caller-side arithmetic of the end-to-end flow
// (examples/example_endtoend.py:47, :75, :86: T_b (1 + delta), cube + foregrounds, += noise) kept on the device
template <typename T>
__global__ void k_axpby(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, long long n, T a, T b, T c) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x)
        out[q] = y ? x[q] * a + y[q] * b + c : x[q] * a + c;
}
// out = x * y (element-wise)
template <typename T>
__global__ void k_mul_real(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, long long n) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x)
        out[q] = x[q] * y[q];
}

// ---- k-space multipliers ---------------------------------------------------------------------------
// box.py:374-379 (apply_transfer_fn) and :651-653 (smooth_field).
enum { FILT_TABLE = 0, FILT_BEAM_HIGHPASS = 1, FILT_WEDGE = 2, FILT_TOPHAT = 3 };
struct FilterSpec {
    int kind;
    double p[4];
    const void* table;   // FILT_TABLE: real multiplier in the layout of the field
};
// BEAM_HIGHPASS: (1 - exp(-0.5 (|kpar|/p0)^p2)) [p0 > 0]  *  exp(-0.5 (kperp/p1)^2) [p1 > 0]
// WEDGE       : 0 where |kpar| < p0 * kperp + p1, else 1
// TOPHAT      : 3 (sin x - x cos x)/x^3, x = |k| p0   (NaN at k=0 -> 0 after nan_to_num)
// TOPHAT = false: a caller that never sees that kind (the fused filtering pass of the strided FFT kernel, whose
// registers the sine / cosine evaluation would not fit beside the line) compiles without it.
template <typename T, bool TOPHAT = true>
__device__ __forceinline__ T filter_value(const FilterSpec& f, const KGeom& g, int i, int j, int l, long long idx,
                                          double kperp_row) {
    if (f.kind == FILT_TABLE) return reinterpret_cast<const T*>(f.table)[idx];
    if (TOPHAT && f.kind == FILT_TOPHAT) {
        const T x = (T)(kmag_exact(g, i, j, l) * f.p[0]);
        T s, c;
        fb_sincos(x, &s, &c);
        return ((T)3 / (x * x * x)) * (s - x * c);
    }
    if (f.kind == FILT_WEDGE) {
        // a mask: decided in double on the tables numpy's own expression would see (k_par = 2 pi m / L_z, k_perp as
        // box.py:374 forms it), product and sum rounded separately as numpy does -- modes exactly on the wedge's
        // edge (|m_z| = slope * m_perp happens on integer grids) fall on the same side as in the reference
#pragma clang fp contract(off)
        const double edge = f.p[0] * kperp_row + f.p[1];
        return (fabs(g.kpar[l]) < edge) ? (T)0 : (T)1;
    }
    const T kperp = (T)kperp_row;
    const T kpar = (T)g.kpar[l];
    T v = (T)1;
    if (f.p[0] > 0) {
        const T r = fabs(kpar) / (T)f.p[0];
        const T rp = (f.p[2] == 2.0) ? r * r : pow(r, (T)f.p[2]);
        v *= (T)1 - exp((T)-0.5 * rp);
    }
    if (f.p[1] > 0) { const T r = kperp / (T)f.p[1]; v *= exp((T)-0.5 * r * r); }
    return v;
}
// The k-space pointwise kernels below run one (k_x, k_y) row per group of 64 lanes (blockDim =
// (64, FB_ROW_GROUPS)); a lane walks k_z = lane, lane + 64, ...  Row constants (k_perp, the
// k_x^2 + k_y^2 part of k^2) are computed once per lane instead of once per mode.
#define FB_ROW_GROUPS 4
// out = nan_to_num(in * T(k)); pitch = NZP (half) or N (full); nz = stored k_z count
template <typename T>
__global__ __launch_bounds__(64 * FB_ROW_GROUPS)
void k_apply_filter(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, FilterSpec f, KGeom g, int pitch, int nz) {
    const long long row = (long long)blockIdx.x * FB_ROW_GROUPS + threadIdx.y;
    if (row >= (long long)g.N * g.N) return;
    const int i = (int)(row / g.N), j = (int)(row % g.N);
    const long long base = ((long long)i * (nz == g.NZV ? g.NR : g.N) + j) * pitch;
    const double kperp = kperp_exact(g, i, j);
    for (int l = threadIdx.x; l < nz; l += 64) {
        const T m = filter_value<T>(f, g, i, j, l, base + l, kperp);
        const cx<T> d = in[base + l];
        out[base + l] = cx<T>{nan_to_num(d.x * m), nan_to_num(d.y * m)};
    }
}

// ---- stand-alone shell binning of a stored spectrum (box.py:741-764) ----------------------------------
// partial[2*nbins][block] = (sum w p, sum w p^2), p = |delta_k|^2 (times T(k)^2 when a filter is given:
// the power spectrum of apply_transfer_fn's result without storing the filtered spectrum), w = multiplicity
// of the stored mode in the full grid (2 for 0 < k_z < N/2 of a half spectrum).
// One (k_x, k_y) row per wave, 64 consecutive k_z per step.  Cubic boxes: the n^2 range of the step is
// wave-uniform, so the step lies in one bin or straddles one edge almost always -- each lane splits at that
// edge, four DPP wave sums, one lane adds into the wave's fp64 row.  Otherwise (several edges, a shell
// within rounding of an edge, non-cubic box): bin per lane, one reduction per distinct bin (wave_flush).
// Deterministic: fixed row -> wave assignment, fixed-order reductions.
#define FB_BIN_GROUP 5          // elements a lane holds per step: a half-spectrum row of N <= 512 is one step
#define FB_BIN_MAXSPAN 8        // bins one row may touch on the fast path
template <typename T>
__global__ __launch_bounds__(64 * FB_BIN_WAVES)
void k_bin_rows(const cx<T>* __restrict__ spec, double* __restrict__ partial, KGeom g, BinGeom bg, FilterSpec f,
                int full_layout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lbins = reinterpret_cast<double*>(smem);
    double* acc = lbins + bg.nbins;                                   // [waves][2 nbins]
    int* lthr = reinterpret_cast<int*>(acc + FB_BIN_WAVES * bg.nbins * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = bg.nbins, N = g.N;
    for (int q = tid; q < nb; q += blockDim.x) { lbins[q] = bg.bins[q]; lthr[q] = bg.thr ? bg.thr[q] : 0; }
    for (int q = tid; q < FB_BIN_WAVES * nb * 2; q += blockDim.x) acc[q] = 0.0;
    __syncthreads();
    double* row_acc = acc + (size_t)wave * nb * 2;
    const int nz = full_layout ? N : g.NZV, pitch = full_layout ? N : g.NZP;
    const long long nrows = (long long)N * N;
    for (long long row = (long long)blockIdx.x * FB_BIN_WAVES + wave; row < nrows;
         row += (long long)gridDim.x * FB_BIN_WAVES) {
        const int i = (int)(row / N), j = (int)(row % N);
        const long long base = (full_layout ? row : (long long)i * g.NR + j) * pitch;
        const int mx = mode_of(i, N), my = mode_of(j, N);
        const int n2row = mx * mx + my * my;
        const double kperp = f.kind >= 0 ? kperp_exact(g, i, j) : 0.0;
        // The row's n^2 runs from n2row (k_z = 0) to n2row + (N/2)^2, whichever the layout: the bins it can touch
        // are blo..bhi, and a mode's bin is blo + the number of thresholds thr[blo..bhi-1] it has reached.
        int blo = 0, bhi = 0, span;
        int tv[FB_BIN_MAXSPAN];
        double ev[FB_BIN_MAXSPAN];
        if (bg.thr) {
            blo = shell_bin(lthr, nb, n2row);
            bhi = shell_bin(lthr, nb, n2row + (N >> 1) * (N >> 1));
            span = bhi - blo;                                          // thresholds inside the row's range
#pragma unroll
            for (int u = 0; u < FB_BIN_MAXSPAN; ++u) tv[u] = (u < span && blo + u < nb) ? lthr[blo + u] : 0x7fffffff;
        } else {
            // any box shape: the same with the bin edges themselves and the exact fp64 |k| of every mode
            // (|k| grows with |m_z| along the row, so its end points bound the bins)
            blo = bin_exact(lbins, nb, kmag_exact(g, i, j, 0));
            bhi = bin_exact(lbins, nb, kmag_exact(g, i, j, N >> 1));
            span = bhi - blo;
#pragma unroll
            for (int u = 0; u < FB_BIN_MAXSPAN; ++u) ev[u] = (u < span && blo + u < nb) ? lbins[blo + u] : 1.0e300;
        }
        for (int g0 = 0; g0 < nz; g0 += 64 * FB_BIN_GROUP) {
            // a lane takes FB_BIN_GROUP CONSECUTIVE k_z (its modes then differ little in |k|); all loads first
            cx<T> dv[FB_BIN_GROUP];
#pragma unroll
            for (int c = 0; c < FB_BIN_GROUP; ++c) {
                const int l = g0 + lane * FB_BIN_GROUP + c;
                dv[c] = l < nz ? spec[base + l] : cx<T>{0, 0};
            }
            double wp[FB_BIN_GROUP], wp2[FB_BIN_GROUP];   // fp64 sums: a bin of equal modes must come out with zero spread
            int rel[FB_BIN_GROUP];
            bool special = span > FB_BIN_MAXSPAN;                      // too many bins / no thresholds: per-mode path
#pragma unroll
            for (int c = 0; c < FB_BIN_GROUP; ++c) {
                const int l = g0 + lane * FB_BIN_GROUP + c;
                const bool ok = l < nz;
                cx<T> d = dv[c];
                if (ok && f.kind >= 0) {
                    const T m = filter_value<T>(f, g, i, j, l, base + l, kperp);
                    d = cx<T>{nan_to_num(d.x * m), nan_to_num(d.y * m)};
                }
                const double p = (double)(d.x * d.x + d.y * d.y);
                const double w = !ok ? 0.0 : ((full_layout || l == 0 || l == (N >> 1)) ? 1.0 : 2.0);
                wp[c] = w * p; wp2[c] = wp[c] * p;
                const int ml = mode_of(l, N);
                const int n2 = n2row + ml * ml;
                int r = 0;
                if (bg.thr) {
#pragma unroll
                    for (int u = 0; u < FB_BIN_MAXSPAN; ++u) r += (n2 >= tv[u]) ? 1 : 0;
                    for (int z = 0; z < bg.namb; ++z) special |= ok && (bg.amb[z] == n2);   // needs the exact |k|
                } else {
                    const double k = kmag_exact(g, i, j, ok ? l : 0);
#pragma unroll
                    for (int u = 0; u < FB_BIN_MAXSPAN; ++u) r += (k >= ev[u]) ? 1 : 0;
                }
                rel[c] = r;
            }
            if (!__any(special)) {
                for (int u = 0; u <= span; ++u) {                       // wave-uniform trip count
                    if (blo + u >= nb) break;
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                    for (int c = 0; c < FB_BIN_GROUP; ++c) { const bool in = rel[c] == u; s1 += in ? wp[c] : 0.0; s2 += in ? wp2[c] : 0.0; }
                    s1 = wave_sum(s1); s2 = wave_sum(s2);
                    if (lane == 0) { row_acc[2 * (blo + u)] += s1; row_acc[2 * (blo + u) + 1] += s2; }
                }
            } else {
#pragma unroll
                for (int c = 0; c < FB_BIN_GROUP; ++c) {
                    const int l = g0 + lane * FB_BIN_GROUP + c;
                    const bool ok = l < nz;
                    const int b = ok ? bin_of_mode(bg, g, lbins, lthr, i, j, l) : nb;
                    wave_flush(b, wp[c], wp2[c], ok && b < nb, row_acc);
                }
            }
        }
    }
    __syncthreads();
    for (int q = tid; q < 2 * nb; q += blockDim.x) {
        double s = 0.0;
        for (int w = 0; w < FB_BIN_WAVES; ++w) s += acc[(size_t)w * nb * 2 + q];
        partial[(size_t)q * gridDim.x + blockIdx.x] = s;          // [value][workgroup]: see k_bin_finish
    }
}

// ---- velocity / potential ------------------------------------------------------------------------------
// box.py:251-284: A_c = i delta_k k_c / k^2, NaN -> 0, plane m_c = -N/2 zeroed, times fac.
// fp64 plans follow the reference's operation order (k = 2 pi sqrt(.), k^2 = k k, two divisions);
// fp32 plans form k_c fac / k^2 once per mode with a single-precision division (3e-7 relative).
// s2 = (m_x/L_x)^2 + (m_y/L_y)^2 + (m_z/L_z)^2 summed in kmag_exact's order; ic = index along comp.
template <typename T>
__device__ __forceinline__ cx<T> velocity_of(const KGeom& g, int comp, double fac, int ic, double s2, cx<T> d) {
#pragma clang fp contract(off)
    cx<T> v{0, 0};
    if (s2 > 0.0 && ic != (g.N >> 1)) {
        const double kc = g.ksc[comp * g.N + ic];
        if constexpr (sizeof(T) == 4) {
            const float m = (float)(kc * fac) / (float)((FB_TWO_PI * FB_TWO_PI) * s2);
            v.x = -d.y * m;
            v.y = d.x * m;
        } else {
            const double k = FB_TWO_PI * sqrt(s2);
            const double k2 = k * k;
            // (i d) * kc / k2  with i d = (-d.y, d.x)
            v.x = (T)(((double)(-d.y) * kc) / k2 * fac);
            v.y = (T)(((double)d.x * kc) / k2 * fac);
        }
    }
    return v;
}
template <typename T>
__global__ __launch_bounds__(64 * FB_ROW_GROUPS)
void k_velocity(const cx<T>* __restrict__ dk, cx<T>* __restrict__ out, KGeom g, int comp, double fac, int pitch, int nz) {
    const long long row = (long long)blockIdx.x * FB_ROW_GROUPS + threadIdx.y;
    if (row >= (long long)g.N * g.N) return;
    const int i = (int)(row / g.N), j = (int)(row % g.N);
    const long long base = ((long long)i * (nz == g.NZV ? g.NR : g.N) + j) * pitch;
    const double rowsum = g.axis2[i] + g.axis2[g.N + j];
    for (int l = threadIdx.x; l < nz; l += 64) {
#pragma clang fp contract(off)
        const int ic = comp == 0 ? i : (comp == 1 ? j : l);
        const double s2 = rowsum + g.axis2[2 * g.N + l];
        const cx<T> v = velocity_of<T>(g, comp, fac, ic, s2, dk[base + l]);
        out[base + l] = v;
    }
}
// box.py:347-348
template <typename T>
__global__ __launch_bounds__(64 * FB_ROW_GROUPS)
void k_potential(const cx<T>* __restrict__ dk, cx<T>* __restrict__ out, KGeom g, int pitch, int nz) {
    const long long row = (long long)blockIdx.x * FB_ROW_GROUPS + threadIdx.y;
    if (row >= (long long)g.N * g.N) return;
    const int i = (int)(row / g.N), j = (int)(row % g.N);
    const long long base = ((long long)i * (nz == g.NZV ? g.NR : g.N) + j) * pitch;
    const double rowsum = g.axis2[i] + g.axis2[g.N + j];
    for (int l = threadIdx.x; l < nz; l += 64) {
#pragma clang fp contract(off)
        const cx<T> d = dk[base + l];
        cx<T> v{0, 0};
        if (i | j | l) {
            if constexpr (sizeof(T) == 4) {
                const float r = 1.0f / (float)((FB_TWO_PI * FB_TWO_PI) * (rowsum + g.axis2[2 * g.N + l]));
                v.x = d.x * r; v.y = d.y * r;
            } else {
                const double k = FB_TWO_PI * sqrt(rowsum + g.axis2[2 * g.N + l]);
                const double k2 = k * k;
                v.x = (T)((double)d.x / k2); v.y = (T)((double)d.y / k2);
            }
        }
        out[base + l] = v;
    }
}

// ---- layout helpers ----------------------------------------------------------------------------------------
// full[k] = half[k] (k_z <= N/2) or conj(half[-k]) -- Hermitian extension for host views of delta_k
template <typename T>
__global__ void k_expand_half(const cx<T>* __restrict__ half, cx<T>* __restrict__ full, KGeom g) {
    const int N = g.N;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = blockIdx.z;
    if (l >= N) return;
    cx<T> v;
    if (l <= (N >> 1)) v = half[((long long)i * g.NR + j) * g.NZP + l];
    else v = cconj(half[((long long)mirror_of(i, N) * g.NR + mirror_of(j, N)) * g.NZP + (N - l)]);
    full[((long long)i * N + j) * N + l] = v;
}
template <typename T>
__global__ void k_crop_full(const cx<T>* __restrict__ full, cx<T>* __restrict__ half, KGeom g) {
    const int N = g.N;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = blockIdx.z;
    if (l >= g.NZV) return;
    half[((long long)i * g.NR + j) * g.NZP + l] = full[((long long)i * N + j) * N + l];
}

// ---- slab-decomposed FFT: x-slab <-> all-to-all buffer ----------------------------------------
// pack  : buf[s][xl][kyl][pitch] = half_local[xl][s*nyl + kyl][pitch]   (before the exchange)
// unpack: half_local[xl][q*nyl + kyl][pitch] = buf[q][xl][kyl][pitch]   (after the exchange)
// One 16-byte vector per lane along the row; rows = nparts*nyl = N.
template <int PACK>
__global__ void k_slab_permute(const uint4* __restrict__ in, uint4* __restrict__ out, int nxl, int nyl, int nparts,
                               int row_vec, int plane_rows) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= row_vec) return;
    const int ky = blockIdx.y, xl = blockIdx.z;
    const int sidx = ky / nyl, kyl = ky % nyl;
    const long long slab = ((long long)xl * plane_rows + ky) * row_vec + v;
    const long long buf = (((long long)sidx * nxl + xl) * nyl + kyl) * row_vec + v;
    if (PACK) out[buf] = in[slab]; else out[slab] = in[buf];
}

// ---- redshift-space remap ------------------------------------------------------------------------------------
// box.py:412-437, one workgroup per line of sight: s = z - (v + sigma n)/H, periodic wrap,
// sort (s, delta) in LDS (bitonic), then for every grid point the scipy griddata/np.interp
// rule: left bracket by bisection, slope*(x - x_lo) + y_lo, exact hit -> y, outside -> fill.
// method 2 ('cubic', round 4): what griddata(method='cubic') is in one dimension -- interp1d(kind='cubic') =
// make_interp_spline(k=3): THE not-a-knot cubic spline through the sorted (s, delta) pairs, the fill value outside
// [min s, max s].  Second derivatives M_i from the tridiagonal system  h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} =
// 6 (D_i - D_{i-1})  (h_i = s_{i+1} - s_i, D_i = (delta_{i+1} - delta_i) / h_i) with the end conditions "third derivative
// continuous at s_1 and s_{n-2}" eliminated into its first and last row, solved by one thread (Thomas: 2 n dependent steps in
// LDS; the spline is unique, so this is scipy's banded solve up to rounding); every thread then evaluates its grid points.
// Equal abscissae make scipy raise; here they give non-finite values on that line.
template <typename T>
__global__ void k_rsd(const T* __restrict__ delta, const T* __restrict__ vz, const T* __restrict__ noise,
                      T* __restrict__ out, const double* __restrict__ zgrid, int N, double Hz, double sigma_nl,
                      RngKey rkey, int nearest, int N2) {
    // N2: N rounded up to a power of two -- the length of the sorting network (grids that are not powers of two: the
    // extra keys are +infinity and stay behind the N real ones)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* key = reinterpret_cast<double*>(smem);   // [N2]
    double* val = key + N2;                           // [N2]
    [[maybe_unused]] double* M2 = val + N2;           // [N]  cubic: second derivatives (the Thomas right-hand sides on the way)
    [[maybe_unused]] double* cp = M2 + N2;            // [N]  cubic: modified upper diagonal
    const long long los = blockIdx.x;
    const T* d = delta + los * N;
    const T* v = vz + los * N;
    const double zmin = zgrid[0], zmax = zgrid[N - 1];
    const double len = zmax - zmin;
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
#pragma clang fp contract(off)
        double vel = (double)v[m];
        if (sigma_nl > 0.0) {
            double n;
            if (noise) n = (double)noise[los * N + m];
            else n = (double)los_noise_at<T>((unsigned long long)los * N + m, rkey);
            vel = vel + sigma_nl * n;
        }
        double s = zgrid[m] - vel / Hz;
        double r = fmod(s - zmin, len);               // numpy % : result takes the divisor's sign
        if (r != 0.0) { if (r < 0.0) r += len; } else r = 0.0;
        key[m] = r + zmin;
        val[m] = (double)d[m];
    }
    for (int m = N + threadIdx.x; m < N2; m += blockDim.x) { key[m] = __builtin_huge_val(); val[m] = 0.0; }
    __syncthreads();
    for (int k = 2; k <= N2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            // one compare-exchange per thread and step: pair q -> (m, m | j) with bit j of m clear
            for (int q = threadIdx.x; q < (N2 >> 1); q += blockDim.x) {
                const int m = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const int p = m | j;
                const bool up = (m & k) == 0;
                const double a = key[m], b = key[p];
                if ((a > b) == up) { key[m] = b; key[p] = a; const double t = val[m]; val[m] = val[p]; val[p] = t; }
            }
            __syncthreads();
        }
    const double fill = 0.5 * ((double)d[0] + (double)d[N - 1]);
    if (nearest == 2) {
        if (threadIdx.x == 0) {
#pragma clang fp contract(off)
            const int n = N;
            auto h = [&](int i) { return key[i + 1] - key[i]; };
            auto D = [&](int i) { return (val[i + 1] - val[i]) / (key[i + 1] - key[i]); };
            // row 1 with M_0 = (1 + h0/h1) M_1 - (h0/h1) M_2 substituted
            {
                const double h0 = h(0), h1 = h(1);
                const double b = 2.0 * (h0 + h1) + h0 + h0 * h0 / h1, c = h1 - h0 * h0 / h1;
                cp[1] = c / b;
                M2[1] = 6.0 * (D(1) - D(0)) / b;
            }
            for (int i = 2; i <= n - 2; ++i) {
                const double a = h(i - 1), hi = h(i);
                double b = 2.0 * (a + hi), c = hi, aa = a;
                if (i == n - 2) {          // last row with M_{n-1} = (1 + h_{n-2}/h_{n-3}) M_{n-2} - (h_{n-2}/h_{n-3}) M_{n-3} substituted
                    aa = a - hi * hi / a;
                    b = 2.0 * (a + hi) + hi + hi * hi / a;
                    c = 0.0;
                }
                const double den = b - aa * cp[i - 1];
                cp[i] = c / den;
                M2[i] = (6.0 * (D(i) - D(i - 1)) - aa * M2[i - 1]) / den;
            }
            for (int i = n - 3; i >= 1; --i) M2[i] = M2[i] - cp[i] * M2[i + 1];
            {
                const double r0 = h(0) / h(1), rn = h(n - 2) / h(n - 3);
                M2[0] = (1.0 + r0) * M2[1] - r0 * M2[2];
                M2[n - 1] = (1.0 + rn) * M2[n - 2] - rn * M2[n - 3];
            }
        }
        __syncthreads();
        for (int m = threadIdx.x; m < N; m += blockDim.x) {
#pragma clang fp contract(off)
            const double x = zgrid[m];
            double y;
            if (x < key[0] || x > key[N - 1]) y = fill;
            else {
                int lo = 0, hi = N;                        // count of key[] <= x
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= x) lo = mid + 1; else hi = mid; }
                int jx = lo - 1;
                if (jx > N - 2) jx = N - 2;                // x == key[N - 1]: the last interval
                const double hh = key[jx + 1] - key[jx], A = key[jx + 1] - x, B = x - key[jx];
                y = M2[jx] * (A * A * A) / (6.0 * hh) + M2[jx + 1] * (B * B * B) / (6.0 * hh)
                  + (val[jx] / hh - M2[jx] * hh / 6.0) * A + (val[jx + 1] / hh - M2[jx + 1] * hh / 6.0) * B;
            }
            out[los * N + m] = (T)y;
        }
        return;
    }
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
#pragma clang fp contract(off)
        const double x = zgrid[m];
        double y;
        if (nearest) {
            // griddata(method='nearest') in 1-D = interp1d(kind='nearest', fill_value='extrapolate'): the key whose
            // neighbourhood, bounded by the midpoints k_a/2 + k_b/2, holds x; a midpoint itself goes to the lower key
            int lo = 0, hi = N;                        // count of key[] <= x
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= x) lo = mid + 1; else hi = mid; }
            if (lo == 0) y = val[0];
            else if (lo == N) y = val[N - 1];
            else y = (key[lo - 1] * 0.5 + key[lo] * 0.5 < x) ? val[lo] : val[lo - 1];
        } else if (x < key[0] || x > key[N - 1]) y = fill;
        else {
            int lo = 0, hi = N;                        // count of key[] <= x
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= x) lo = mid + 1; else hi = mid; }
            const int jx = lo - 1;
            if (jx == N - 1 || key[jx] == x) y = val[jx];
            else {
                const double slope = (val[jx + 1] - val[jx]) / (key[jx + 1] - key[jx]);
                y = slope * (x - key[jx]) + val[jx];
                if (y != y) {
                    y = slope * (x - key[jx + 1]) + val[jx + 1];
                    if (y != y && val[jx] == val[jx + 1]) y = val[jx];
                }
            }
        }
        out[los * N + m] = (T)y;
    }
}

// ---- redshift-space remap without a sort (N >= 64) -----------------------------------------
// np.interp on the sorted (s, delta) pairs only ever needs, for grid point z_c, the largest
// s <= z_c and the smallest s > z_c.  With cell(s) = c for z_c <= s < z_{c+1}: the predecessor of
// z_c is the maximum of the nearest non-empty cell below c, the successor the minimum of the
// nearest non-empty cell at or above c, and an exact hit is min(cell c) == z_c.  So: per-cell
// min / max by 64-bit LDS atomics on an order-preserving encoding of s, the values attached in a
// second sweep, nearest non-empty cells by a wave scan.  One wavefront per line of sight, lane l
// owns cells l*E .. l*E+E-1.  O(N) per line instead of the O(N log^2 N) sorting network of k_rsd.
__device__ __forceinline__ unsigned long long order_bits(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double order_value(unsigned long long u) {
    return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}
// fmod(a, b) for b > 0: the correctly rounded quotient truncates to the true integer quotient or
// one past it, a - q b is then exactly representable (fma), one conditional add repairs the latter.
__device__ __forceinline__ double fmod_pos(double a, double b) {
    const double q = trunc(a / b);
    if (!(fabs(q) < 4.5e15)) return fmod(a, b);
    double r = fma(-q, b, a);
    if (a >= 0.0 ? r < 0.0 : r > 0.0) r += (a >= 0.0 ? b : -b);
    return r;
}

#ifndef FB_RSD_WAVES_SMALL
#define FB_RSD_WAVES_SMALL 4      // lines of sight (waves) per workgroup up to 8 cells per lane (tuning: 8 with FB_RSD_OCC 6)
#endif
#ifndef FB_RSD_OCC
#define FB_RSD_OCC 6              // waves per SIMD the single-precision kernel is compiled for, up to 8 cells per lane (32-bit keys: 80 registers)
#endif
constexpr int rsd_waves(int E) { return E <= 8 ? FB_RSD_WAVES_SMALL : 4; }

// A line of sight is one wave's: its key / value arrays are touched by no other wave, and a wave's LDS instructions
// (atomics included) execute in issue order, so between the phases the compiler only has to be kept from reordering
// -- no workgroup barrier, and the block's four lines of sight do not run in lockstep.
__device__ __forceinline__ void rsd_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Key of a sample inside its cell: 64-bit order-preserving position bits (double-precision plans), or -- single
// precision -- 31 bits of the in-cell fraction + 1 (0 and ~0 stay free as the "empty cell" marks).
template <typename T> struct rsd_key { typedef unsigned long long type; };
template <> struct rsd_key<float> { typedef unsigned int type; };

// The remap of one line of sight by one wave, single-precision plans: the line in CELL units.  A sample's position is
// s = fract((z_m - zmin - v / H) / len) (N - 1) in [0, N - 1): its cell is floor(s) (the grid is uniform) and its place
// inside the cell a 31-bit fraction -- 4.7e-10 of a cell, three orders below what a single-precision velocity carries
// (1e-7 of |v| / H, i.e. of a few cells).  Keys are then 32 bits: half the LDS bytes of the 64-bit position keys, the
// 32-bit LDS atomics, 16 registers less (a sixth wave per SIMD).  A bracket's position is (cell + fraction): the
// differences the interpolation needs are formed exactly in fp64 from the integer cell distance and the two fractions
// (close keys cancel), the quotient in fp32.  Branch-free as the fp64 form below (see there for the phases).
template <int E, bool ZG_GLOBAL>
__device__ __forceinline__ void rsd_remap_line32(const float (&vin)[E], const float (&val)[E], const float* __restrict__ noise,
                                                 const long long los, const double fill, const double* zg, unsigned int* kex,
                                                 float* vex, const double zmin, const double len, const double Hz,
                                                 const double sigma_nl, const RngKey rkey, const int nearest, const int lane,
                                                 float (&y_out)[E]) {
    constexpr int N = E * 64;
    typedef unsigned int u32;
    auto sw = [](int c) { return (c % E) * 64 + c / E; };
    auto zown = [&](int e) -> double {               // this lane's e-th grid point, measured from zmin and moved up by len
        if constexpr (ZG_GLOBAL) return (zg[lane * E + e] - zmin) + len;
        else return zg[lane + 64 * e];
    };
    const double inv_Hz = 1.0 / Hz, inv_len = 1.0 / len;
    const double cells = (double)(N - 1);
    const double s_max = __longlong_as_double(__double_as_longlong(cells) - 1);     // largest double < N - 1
    u32 kb[E];
    int cell[E];
    float nz[E];
    if (sigma_nl > 0.0) {
        const unsigned long long idx0 = (unsigned long long)los * N + lane * E;
        if (noise) {
#pragma unroll
            for (int e = 0; e < E; ++e) nz[e] = noise[idx0 + e];
        } else if constexpr (E % 4 == 0) {
#pragma unroll
            for (int q = 0; q < E / 4; ++q)
                stream_normals4<float>((idx0 >> 2) + q, 1u, rkey, nz[4 * q], nz[4 * q + 1], nz[4 * q + 2], nz[4 * q + 3]);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) nz[e] = los_noise_at<float>(idx0 + e, rkey);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma clang fp contract(off)
        double vel = (double)vin[e];
        // (asm: keeps the wave-uniform case a scalar branch -- if-converted, its fp64 instructions would run for every cell)
        if (sigma_nl > 0.0) { asm volatile(""); vel = vel + sigma_nl * (double)nz[e]; }
        // reciprocals replace the two fp64 divisions (displacement, wrap); the wrap is a fract()
        const double a = fma(-vel, inv_Hz, zown(e) - len);                        // z_m - zmin - vel / H
        double sp = __builtin_amdgcn_fract(a * inv_len) * cells;                  // in [0, N - 1]
        sp = sp < s_max ? sp : s_max;
        const int c = (int)sp;                                                     // 0 .. N - 2
        kb[e] = (u32)((sp - (double)c) * 2147483648.0) + 1u;                       // 1 .. 2^31
        cell[e] = sw(c);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) atomicMax(&kex[cell[e]], kb[e]);
    rsd_wave_sync();
    // (the key that won a cell attaches its value: an unconditional store, the losers' to a spare slot behind the array)
#pragma unroll
    for (int e = 0; e < E; ++e) vex[kex[cell[e]] == kb[e] ? cell[e] : N] = val[e];
    rsd_wave_sync();
    unsigned occ = 0;                                 // which of this lane's cells are occupied
    int last = -1, first = N;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool o = kex[lane + 64 * e] != 0u;
        occ |= o ? (1u << e) : 0u;
        first = (o && first == N) ? lane * E + e : first;
        last = o ? lane * E + e : last;
    }
    int below = last, above = first;                  // inclusive scans over lanes: max from the left, min from the right
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int b = __shfl_up(below, sft), a = __shfl_down(above, sft);
        if (lane >= sft) below = b > below ? b : below;
        if (lane + sft < 64) above = a < above ? a : above;
    }
    int run_below = __shfl_up(below, 1);              // exclusive: lanes to the left only
    if (lane == 0) run_below = -1;
    int nxt_above = __shfl_down(above, 1);
    if (lane == 63) nxt_above = N;
    int ab[E];                                        // nearest occupied cell at or above each of this lane's cells (N: none)
    {
        int a = nxt_above;
#pragma unroll
        for (int e = E - 1; e >= 0; --e) { a = ((occ >> e) & 1u) ? lane * E + e : a; ab[e] = a; }
    }
    // lower bracket of every cell of this lane: (largest key, its value) of the nearest occupied cell below
    u32 pk[E];
    float pv[E];
    {
        int rb = run_below;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int q = sw(rb < 0 ? 0 : rb);
            pk[e] = kex[q];
            pv[e] = vex[q];
            rb = ((occ >> e) & 1u) ? lane * E + e : rb;
        }
    }
    rsd_wave_sync();
    // second use of the arrays: per-cell minima
#pragma unroll
    for (int e = 0; e < E; ++e) kex[lane + 64 * e] = ~0u;
    rsd_wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) atomicMin(&kex[cell[e]], kb[e]);
    rsd_wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) vex[kex[cell[e]] == kb[e] ? cell[e] : N] = val[e];
    rsd_wave_sync();
    int rb = run_below;                               // cell of the lower bracket, as in the loop that fetched it
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma clang fp contract(off)
        const int c = lane + 64 * e, own = lane * E + e;       // LDS index and number of this cell; its grid point sits at `own`
        const bool filled = (occ >> e) & 1u;
        // every LDS read unconditional (asm: the value counts as used, so the compiler cannot sink the read behind a per-lane
        // branch -- exec-mask bookkeeping runs on the CU's one scalar unit), bracket indices clamped into range, the cases
        // selected at the end
        const int cn = ab[e] < N ? ab[e] : N - 1;
        const int ra = sw(cn);
        u32 cmn = kex[c], kna = kex[ra];
        float vn = vex[ra], vc = vex[c];
        asm volatile("" : "+v"(cmn), "+v"(kna), "+v"(vn), "+v"(vc));
        const bool exact = filled & (cmn == 1u);                // a sample exactly on the grid point
        const bool nofill = (rb >= 0) & (ab[e] < N);
        const float vj = pv[e];
        const double fj = (double)(pk[e] - 1u) * 4.656612873077392578125e-10;     // 2^-31
        const double fn = (double)(kna - 1u) * 4.656612873077392578125e-10;
        const double xj = (double)(own - rb) - fj;              // grid point - lower bracket, in cells (> 0)
        const double nj = (double)(cn - rb) + (fn - fj);        // upper bracket - lower bracket
        const float w = (float)xj * __builtin_amdgcn_rcpf((float)nj);
        float yi = vj + (vn - vj) * w;
        yi = (yi != yi && vj == vn) ? vj : yi;
        float yl = nofill ? yi : (float)fill;
        if (nearest) {                                // (wave-uniform) see k_rsd: the nearer bracket, the lower one on a midpoint
            asm volatile("");                         // (a scalar branch, not if-converted fp64 instructions per cell)
            const double mid = fj * 0.5 + ((double)(cn - rb) + fn) * 0.5;          // midpoint of the brackets, from the lower one's cell
            yl = (rb < 0) ? vn : ((ab[e] >= N) ? vj : ((mid < (double)(own - rb)) ? vn : vj));
        }
        y_out[e] = exact ? vc : yl;
        rb = filled ? own : rb;
    }
}

// The remap of ONE line of sight by one wave.  On entry: vin[] / val[] = this lane's E consecutive velocities / densities
// (cells lane E .. lane E + E - 1), zg[] = the grid in the kernel's frame and LDS order (see k_rsd_cells), kex[] zeroed,
// all of it published to the wave; fill = (delta[0] + delta[N-1]) / 2 (np.interp outside the samples, box.py:433).
// On exit y_out[] = the lane's E cells of the redshift-space line; kex / vex are free again.  (Shared by k_rsd_cells and
// by the fused z pass k_rsd_turn of fb_fft_kernels.h, which must give bit-identical lines.)
template <typename T, int E>
__device__ __forceinline__ void rsd_remap_line64(const T (&vin)[E], const T (&val)[E], const T* __restrict__ noise, const long long los,
                                               const double fill, const double* zg, unsigned long long* kex, T* vex,
                                               const double zmin, const double len, const double Hz, const double sigma_nl,
                                               const RngKey rkey, const int nearest, const int lane, T (&y_out)[E]) {
    constexpr bool ZG_GLOBAL = false;
    constexpr int N = E * 64;
    typedef unsigned long long u64;
    constexpr bool SHIFTED = sizeof(T) == 4;
    auto sw = [](int c) { return (c % E) * 64 + c / E; };
    static_assert(!ZG_GLOBAL || SHIFTED, "the grid is read from the plan's table by the single-precision path only");
    // position of this lane's e-th cell in the kernel's frame: from the LDS copy, or (ZG_GLOBAL: zg = the plan's table in
    // global memory, 8 N bytes that stay in the vector cache) formed as the LDS copy is
    auto zown = [&](int e) -> double {
        if constexpr (ZG_GLOBAL) return (zg[lane * E + e] - zmin) + len;
        else return zg[lane + 64 * e];
    };
    const double inv_dz = (double)(N - 1) / len;
    [[maybe_unused]] const double inv_Hz = 1.0 / Hz, inv_len = 1.0 / len;
    [[maybe_unused]] const double len_below = __longlong_as_double(__double_as_longlong(len) - 1);   // largest double < len
    auto enc = [](double x) -> u64 { if constexpr (SHIFTED) return (u64)__double_as_longlong(x); else return order_bits(x); };
    auto dec = [](u64 b) -> double { if constexpr (SHIFTED) return __longlong_as_double((long long)b); else return order_value(b); };
    u64 kb[E];
    int cell[E], cfin[E];
    int unsettled = 0;
    T nz[E];
    if (sigma_nl > 0.0) {
        const unsigned long long idx0 = (unsigned long long)los * N + lane * E;
        if (noise) {
#pragma unroll
            for (int e = 0; e < E; ++e) nz[e] = noise[idx0 + e];
        } else if constexpr (E % 4 == 0) {
#pragma unroll
            for (int q = 0; q < E / 4; ++q)
                stream_normals4<T>((idx0 >> 2) + q, 1u, rkey, nz[4 * q], nz[4 * q + 1], nz[4 * q + 2], nz[4 * q + 3]);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) nz[e] = los_noise_at<T>(idx0 + e, rkey);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma clang fp contract(off)
        const int m = lane * E + e;
        double vel = (double)vin[e];
        // (asm: keeps the wave-uniform case a scalar branch -- if-converted, its three fp64 instructions would run for
        // every cell of every call)
        if (sigma_nl > 0.0) { asm volatile(""); vel = vel + sigma_nl * (double)nz[e]; }
        if constexpr (SHIFTED) {
            // the inputs carry 1e-7 relative error: reciprocals replace the two fp64 divisions (displacement, wrap), the
            // wrap is a fract(), and the cell follows from the position itself -- a key within rounding (1e-13 cells) of a
            // grid point may sit in the neighbouring cell, which moves an interpolation weight by that much
            const double a = fma(-vel, inv_Hz, zown(e) - len);                   // z_m - zmin - vel / H
            double r = __builtin_amdgcn_fract(a * inv_len) * len;                 // in [0, len]
            r = r < len_below ? r : len_below;
            int c = (int)(r * inv_dz);
            cfin[e] = c < 0 ? 0 : (c > N - 1 ? N - 1 : c);
            kb[e] = enc(r + len);
            continue;
        }
        double r;
        {
            const double s = zg[sw(m)] - vel / Hz;
            r = fmod_pos(s - zmin, len);              // numpy % : result takes the divisor's sign
            if (r != 0.0) { if (r < 0.0) r += len; } else r = 0.0;
        }
        const double key = r + zmin;
        int c = (int)((key - zmin) * inv_dz);
        c = c < 0 ? 0 : (c > N - 1 ? N - 1 : c);
        // settle on z_c <= key < z_{c+1}: the estimate is off by at most one cell, so one step down or one step up --
        // with unconditional LDS reads and selects, no branches: a CU issues ONE scalar instruction per cycle for all
        // its waves (tools/salu_rate.hip), and the exec-mask bookkeeping of per-lane branches here (round 2: ~175
        // scalar instructions per cell) was what bounded this kernel, not its arithmetic.  Whether that one step was
        // enough is checked with a third read and, for the whole wave at once, after the loop (never observed).
        // (asm: the value is "used" here, so the read stays unconditional -- the compiler would otherwise sink it
        // behind the comparison that needs it and branch around it)
        double zc = zg[sw(c)];
        asm volatile("" : "+v"(zc));
        const int down = (int)(key < zc);                 // (c = 0: z_0 = zmin <= key, never)
        c -= down;
        double zn = zg[sw(c < N - 1 ? c + 1 : N - 1)];
        asm volatile("" : "+v"(zn));
        const int up = (down ^ 1) & (int)(c < N - 1) & (int)(key >= zn);
        c += up;
        double zx = zg[sw(down ? c : (c < N - 1 ? c + 1 : N - 1))];
        asm volatile("" : "+v"(zx));
        unsettled |= (down & (int)(key < zx)) | (up & (int)(c < N - 1) & (int)(key >= zx));
        cfin[e] = c;
        kb[e] = enc(key);
    }
    if (__builtin_expect(__any(unsettled), 0)) {       // wave-uniform: walk to the cell that holds the key
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double key = dec(kb[e]);
            int c = cfin[e];
            for (int it = 0; it < N && c > 0 && key < zg[sw(c)]; ++it) --c;
            for (int it = 0; it < N && c < N - 1 && key >= zg[sw(c + 1)]; ++it) ++c;
            cfin[e] = c;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        cell[e] = sw(cfin[e]);                          // from here on the cell's LDS index
        atomicMax(&kex[cell[e]], kb[e]);
    }
    rsd_wave_sync();
    // (the key that won a cell attaches its value: an unconditional store, the losers' to a spare slot behind the
    // wave's array -- a per-lane branch would cost exec-mask bookkeeping on the scalar unit)
#pragma unroll
    for (int e = 0; e < E; ++e) vex[kex[cell[e]] == kb[e] ? cell[e] : N] = val[e];
    rsd_wave_sync();
    // nearest non-empty cell strictly below / at-or-above each of this lane's cells
    // (which of this lane's cells are occupied: one bit each -- the maxima themselves are only needed as "non-empty", and
    // eight 64-bit registers less are what lets a sixth wave per SIMD in)
    unsigned occ = 0;
    int last = -1, first = N;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool o = kex[lane + 64 * e] != 0ull;
        occ |= o ? (1u << e) : 0u;
        first = (o && first == N) ? lane * E + e : first;
        last = o ? lane * E + e : last;
    }
    int below = last, above = first;                  // inclusive scans over lanes: max from the left, min from the right
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int b = __shfl_up(below, sft), a = __shfl_down(above, sft);
        if (lane >= sft) below = b > below ? b : below;
        if (lane + sft < 64) above = a < above ? a : above;
    }
    int run_below = __shfl_up(below, 1);              // exclusive: lanes to the left only
    if (lane == 0) run_below = -1;
    int nxt_above = __shfl_down(above, 1);
    if (lane == 63) nxt_above = N;
    int ab[E];
    {
        int a = nxt_above;
#pragma unroll
        for (int e = E - 1; e >= 0; --e) { a = ((occ >> e) & 1u) ? lane * E + e : a; ab[e] = a; }
    }
    // lower bracket of every cell of this lane: (largest key, its value) of the nearest occupied cell below
    u64 pk[E];
    T pv[E];
    unsigned has_below = 0;                           // bit e: there is an occupied cell below this lane's e-th cell
    {
        int rb = run_below;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            has_below |= rb >= 0 ? (1u << e) : 0u;
            const int q = sw(rb < 0 ? 0 : rb);
            pk[e] = kex[q];
            pv[e] = vex[q];
            rb = ((occ >> e) & 1u) ? lane * E + e : rb;
        }
    }
    rsd_wave_sync();
    // second use of the arrays: per-cell minima
#pragma unroll
    for (int e = 0; e < E; ++e) kex[lane + 64 * e] = ~0ull;
    rsd_wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) atomicMin(&kex[cell[e]], kb[e]);
    rsd_wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) vex[kex[cell[e]] == kb[e] ? cell[e] : N] = val[e];
    rsd_wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma clang fp contract(off)
        const int c = lane + 64 * e;                   // LDS index of cell lane E + e
        const double x = zown(e);
        const bool filled = (occ >> e) & 1u;
        const int run_below = ((has_below >> e) & 1u) ? 0 : -1;         // (only its sign is used)
        const u64 cmn_e = kex[c];
        double y = 0.0;
        if constexpr (sizeof(T) == 4) {
            // branch-free and in single precision but for the two key differences: bracket indices clamped into range,
            // every LDS read unconditional (asm: the value counts as used, so the compiler cannot sink the read behind a
            // per-lane branch -- exec-mask bookkeeping runs on the CU's one scalar unit, which is what bounded this
            // kernel), the cases selected at the end
            const bool exact = filled & (dec(cmn_e) == x);
            const bool nofill = (run_below >= 0) & (ab[e] < N);
            const int ra = sw(ab[e] < N ? ab[e] : N - 1);
            u64 kna = kex[ra];
            float vn = vex[ra], vc = vex[c];
            asm volatile("" : "+v"(kna), "+v"(vn), "+v"(vc));
            const double kj = dec(pk[e]), kn = dec(kna);
            const float vj = pv[e];
            // differences in fp64 (close keys cancel), the quotient in fp32
            const float w = (float)(x - kj) * __builtin_amdgcn_rcpf((float)(kn - kj));
            float yi = vj + (vn - vj) * w;
            yi = (yi != yi && vj == vn) ? vj : yi;
            float yl = nofill ? yi : (float)fill;
            if (nearest) {                            // (wave-uniform) see k_rsd: the nearer bracket, the lower one on a midpoint
                asm volatile("");                     // (a scalar branch, not four if-converted fp64 instructions per cell)
                yl = (run_below < 0) ? vn : ((ab[e] >= N) ? vj : ((kj * 0.5 + kn * 0.5 < x) ? vn : vj));
            }
            y_out[e] = exact ? vc : yl;
            continue;
        } else {
            if (filled && order_value(cmn_e) == x) y = (double)vex[c];
            else if (nearest) {
                const int qa = sw(ab[e] < N ? ab[e] : N - 1);
                const double kj = order_value(pk[e]), kn = order_value(kex[qa]);
                y = (run_below < 0) ? (double)vex[qa] : ((ab[e] >= N) ? (double)pv[e] : ((kj * 0.5 + kn * 0.5 < x) ? (double)vex[qa] : (double)pv[e]));
            } else if (run_below < 0 || ab[e] >= N) y = fill;
            else {
                const double kj = order_value(pk[e]), vj = (double)pv[e];
                const double kn = order_value(kex[sw(ab[e])]), vn = (double)vex[sw(ab[e])];
                const double slope = (vn - vj) / (kn - kj);
                y = slope * (x - kj) + vj;
                if (y != y) {
                    y = slope * (x - kn) + vn;
                    if (y != y && vj == vn) y = vj;
                }
            }
        }
        y_out[e] = (T)y;
    }
}

template <typename T, int E, bool ZG_GLOBAL = false>
__device__ __forceinline__ void rsd_remap_line(const T (&vin)[E], const T (&val)[E], const T* __restrict__ noise, const long long los,
                                               const double fill, const double* zg, typename rsd_key<T>::type* kex, T* vex,
                                               const double zmin, const double len, const double Hz, const double sigma_nl,
                                               const RngKey rkey, const int nearest, const int lane, T (&y_out)[E]) {
    if constexpr (sizeof(T) == 4) {
        rsd_remap_line32<E, ZG_GLOBAL>(vin, val, noise, los, fill, zg, kex, vex, zmin, len, Hz, sigma_nl, rkey, nearest, lane, y_out);
    } else {
        rsd_remap_line64<T, E>(vin, val, noise, los, fill, zg, kex, vex, zmin, len, Hz, sigma_nl, rkey, nearest, lane, y_out);
    }
}
template <typename T, int E>
__global__ __launch_bounds__(64 * rsd_waves(E), (sizeof(T) == 4 && E <= 8) ? FB_RSD_OCC : 1) void k_rsd_cells(
        const T* __restrict__ delta, const T* __restrict__ vz, const T* __restrict__ noise, T* __restrict__ out,
        const double* __restrict__ zgrid, double Hz, double sigma_nl, RngKey rkey, int nearest) {
    constexpr int N = E * 64;
    constexpr int FB_RSD_WAVES = rsd_waves(E);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    typedef typename rsd_key<T>::type key_t;
    double* zg = reinterpret_cast<double*>(smem);                                   // [N]       shared by the block
    // one key array and one value array per wave, used twice: first for the per-cell maxima (from which each
    // lane takes the brackets below its cells into registers), then again for the minima.  Half the LDS of
    // keeping both, and LDS is what limits the number of resident waves here.
    key_t* kex = reinterpret_cast<key_t*>(smem + sizeof(double) * N) + w * N;       // [N] per wave
    T* vex = reinterpret_cast<T*>(smem + sizeof(double) * N + sizeof(key_t) * N * FB_RSD_WAVES) + w * (N + 16);   // [N] per wave + a spare slot
    const long long los = (long long)blockIdx.x * FB_RSD_WAVES + w;                 // N*N % WAVES == 0
    const T* d = delta + los * N;
    const T* v = vz + los * N;
    // LDS layout: cell c lives at index sw(c) = (c % E) 64 + c / E.  Lane l owns the E consecutive cells l E .. l E + E-1
    // (one 16/32-byte global load each way), and nearly every LDS access of the kernel is "my e-th cell" or a cell a
    // few places from it: stored in cell order those are 8 E bytes apart from lane to lane -- an E-way bank conflict on
    // every one of the ~110 LDS instructions per line; transposed, lane l's e-th cell is word l of row e.
    auto sw = [](int c) { return (c % E) * 64 + c / E; };
    // Single-precision plans work in a shifted frame: positions are measured from zmin and moved up by the length of the
    // line, p = (z - zmin) + len in [len, 2 len).  Positive doubles order like their bit patterns, so the order-
    // preserving encoding of the keys is the bit pattern itself (0 and ~0 stay free as the "empty cell" marks), and
    // the wrap is a fract().  Differences of positions -- all the interpolation needs -- do not see the shift.
    constexpr bool SHIFTED = sizeof(T) == 4;
    const double zmin = zgrid[0], zmax = zgrid[N - 1];
    const double len = zmax - zmin;
    for (int m = threadIdx.x; m < N; m += 64 * FB_RSD_WAVES) zg[sw(m)] = SHIFTED ? (zgrid[m] - zmin) + len : zgrid[m];
#pragma unroll
    for (int e = 0; e < E; ++e) kex[lane + 64 * e] = 0;
    __syncthreads();
    T val[E];
    // this lane's E consecutive velocities and densities in 16-byte loads (a lane's cells are contiguous)
    T vin[E];
    if constexpr (E * sizeof(T) % 16 == 0) {
        typedef T vec16 __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int PER = 16 / sizeof(T);
#pragma unroll
        for (int q = 0; q < E / PER; ++q) {
            const vec16 a = reinterpret_cast<const vec16*>(v + lane * E)[q], b = reinterpret_cast<const vec16*>(d + lane * E)[q];
#pragma unroll
            for (int u = 0; u < PER; ++u) { vin[q * PER + u] = a[u]; val[q * PER + u] = b[u]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { vin[e] = v[lane * E + e]; val[e] = d[lane * E + e]; }
    }
    const double fill = 0.5 * ((double)d[0] + (double)d[N - 1]);
    T y_out[E];
    rsd_remap_line<T, E>(vin, val, noise, los, fill, zg, kex, vex, zmin, len, Hz, sigma_nl, rkey, nearest, lane, y_out);
#pragma unroll
    for (int e = 0; e < E; ++e) out[los * N + lane * E + e] = y_out[e];
}

// ---- the steps after the density-field path: foreground maps / cube, radiometer noise ------
// 2-D maps are T[N][N] over (x, y); cubes T[N][N][N] with the frequency axis last (fastest), as box.py's fields.
// RNG streams of the counter generator: 1 line-of-sight noise, 2 foreground map, 3 spectral index, 4 noise cube.

// foregrounds.py:99-103: fg_k = (re + i im) sqrt(C_ell); amp2d holds sqrt(C_ell) with the zero mode already 0
template <typename T>
__global__ void k_sky_colour_map(const T* __restrict__ amp2d, const T* __restrict__ re, const T* __restrict__ im,
                                 cx<T>* __restrict__ out, int n2, RngKey key) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    T a, b;
    if (re) { a = re[q]; b = im[q]; }
    else {
        T g0, g1, g2, g3;
        stream_normals4<T>((unsigned long long)(q >> 1), 2u, key, g0, g1, g2, g3);
        a = (q & 1) ? g2 : g0; b = (q & 1) ? g3 : g1;
    }
    const T A = amp2d[q];
    out[q] = cx<T>{A * a, A * b};
}
// foregrounds.py:107: fg_x = ifftn(fg_k).real + monopole
template <typename T>
__global__ void k_sky_real_plus(const cx<T>* __restrict__ in, T* __restrict__ out, int n2, T add) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n2) out[q] = in[q].x + add;
}
// foregrounds.py:137-138: alpha = normal(mean, std)
template <typename T>
__global__ void k_sky_normal_map(const T* __restrict__ unit, T* __restrict__ out, int n2, double mean, double std, RngKey key) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n2) return;
    T n;
    if (unit) n = unit[q];
    else n = stream_noise_at<T>((unsigned long long)q, 3u, key);
    out[q] = (T)(mean + std * (double)n);
}
// scipy.ndimage.gaussian_filter(mode='wrap') along one axis: out[i] = sum_j w[j] in[(i + j - r) mod N]
// (axis 0: stride N between neighbours, axis 1: stride 1); fp64 accumulation, weights as scipy forms them
template <typename T>
__global__ void k_sky_gauss_axis(const T* __restrict__ in, T* __restrict__ out, const double* __restrict__ w,
                                 int radius, int N, int axis) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N * N) return;
    const int x = q / N, y = q % N;
    const int i = axis == 0 ? x : y;
    double acc = 0.0;
    for (int j = 0; j <= 2 * radius; ++j) {
        int k = (i + j - radius) % N;
        if (k < 0) k += N;
        acc += w[j] * (double)(axis == 0 ? in[(long long)k * N + y] : in[(long long)x * N + k]);
    }
    out[q] = (T)acc;
}
// foregrounds.py:165-175: cube = amps[x,y] (freqs[z]/freq_ref)^alpha[x,y]; lr[z] = log2(freqs[z]/freq_ref)
template <typename T>
__global__ __launch_bounds__(256) void k_sky_fg_cube(const T* __restrict__ amps, const T* __restrict__ alpha,
                                                     double alpha_scalar, const double* __restrict__ ratio,
                                                     T* __restrict__ out, int N) {
    const long long row = blockIdx.x;                                   // (x, y)
    const T a = amps[row];
    const double al = alpha ? (double)alpha[row] : alpha_scalar;
    for (int z = threadIdx.x; z < N; z += blockDim.x) {
        T v;
        if constexpr (sizeof(T) == 4) v = a * __builtin_amdgcn_exp2f((float)al * (float)ratio[N + z]);   // 2^(alpha lr)
        else v = (T)((double)a * pow(ratio[z], al));
        out[row * N + z] = v;
    }
}
// noise.py:72-74: unit normals scaled per frequency channel; device RNG: 4 consecutive channels per call
template <typename T>
__global__ __launch_bounds__(256) void k_sky_noise_cube(const T* __restrict__ unit, const double* __restrict__ sigma,
                                                        T* __restrict__ out, long long n4, int N, RngKey key) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // group of 4 consecutive voxels
    if (q >= n4) return;
    T g[4];
    if (unit) { for (int u = 0; u < 4; ++u) g[u] = unit[4 * q + u]; }
    else stream_normals4<T>((unsigned long long)q, 4u, key, g[0], g[1], g[2], g[3]);
    const int z0 = (int)((4 * q) % N);
#pragma unroll
    for (int u = 0; u < 4; ++u) out[4 * q + u] = (T)((double)g[u] * sigma[z0 + u]);
}

// real cube -> complex cube (imaginary part 0); complex cube *= mask2d[k_x][k_y] (filters.py:79-89)
template <typename T>
__global__ void k_real_to_complex(const T* __restrict__ in, cx<T>* __restrict__ out, long long n) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x)
        out[q] = cx<T>{in[q], (T)0};
}
template <typename T>
__global__ __launch_bounds__(256) void k_mask_xy(cx<T>* __restrict__ cube, const T* __restrict__ mask2d, int N) {
    const long long row = blockIdx.x;                     // (k_x, k_y)
    const T m = mask2d[row];
    if (m == (T)1) return;                                // block-uniform: untouched rows cost nothing
    for (int z = threadIdx.x; z < N; z += blockDim.x) {
        cx<T> v = cube[row * N + z];
        cube[row * N + z] = cx<T>{v.x * m, v.y * m};
    }
}

// ---- beam convolution (fastbox/beams.py:63-137): per-channel 2-D convolution of a field with a beam cube, through an
// M x M transverse transform of both (M = 2n zero-padded for scipy.signal.fftconvolve's linear convolution, M = n for
// convolve2d's boundary='wrap').  One (x, y) row of the M x M grid per workgroup, the n channels contiguous. ---------
template <typename T>
__global__ __launch_bounds__(256) void k_beam_embed(const T* __restrict__ in, cx<T>* __restrict__ out, int n, int M) {
    const long long row = blockIdx.x;
    const int x = (int)(row / M), y = (int)(row % M);
    const bool inside = x < n && y < n;                   // block-uniform
    const T* src = in + ((long long)(inside ? x : 0) * n + (inside ? y : 0)) * n;
    for (int z = threadIdx.x; z < n; z += blockDim.x) out[row * n + z] = cx<T>{inside ? src[z] : (T)0, (T)0};
}
template <typename T>
__global__ __launch_bounds__(256) void k_beam_multiply(cx<T>* __restrict__ a, const cx<T>* __restrict__ b, long long count) {
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += step) {
        const cx<T> u = a[i], v = b[i];
        a[i] = cx<T>{u.x * v.x - u.y * v.y, u.x * v.y + u.y * v.x};
    }
}
// mode='same': out[i, j, z] = conv[(i + off) mod M, (j + off) mod M, z] / sum_xy beam[:, :, z]; the sum is the beam's
// own (k_x, k_y) = (0, 0) coefficient, beam_k[z]
template <typename T>
__global__ __launch_bounds__(256) void k_beam_crop(const cx<T>* __restrict__ conv, const cx<T>* __restrict__ beam_k,
                                                   T* __restrict__ out, int n, int M, int off) {
    const long long row = blockIdx.x;                     // (i, j) of the n x n output
    const int i = (int)(row / n), j = (int)(row % n);
    const long long src = ((long long)((i + off) % M) * M + (j + off) % M) * n;
    for (int z = threadIdx.x; z < n; z += blockDim.x) out[row * n + z] = conv[src + z].x / beam_k[z].x;
}

// ---- PCA foreground cleaning (fastbox/filters.py:93-183): channel means, frequency-frequency covariance,
// projection onto the leading modes.  The cube is T[pixel = (x, y)][channel] with the channel contiguous. ----------

// partial[block][channel] = sum over the block's pixels (fp64); k_bin_finish-style fixed-order finish below
template <typename T>
__global__ __launch_bounds__(256) void k_channel_sums(const T* __restrict__ cube, double* __restrict__ partial,
                                                      long long npix, int N) {
    const long long per = (npix + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        long long p = p0;
        for (; p + 3 < p1; p += 4) {
            a0 += (double)cube[p * N + c]; a1 += (double)cube[(p + 1) * N + c];
            a2 += (double)cube[(p + 2) * N + c]; a3 += (double)cube[(p + 3) * N + c];
        }
        for (; p < p1; ++p) a0 += (double)cube[p * N + c];
        partial[(size_t)blockIdx.x * N + c] = (a0 + a1) + (a2 + a3);
    }
}
// blockDim = (64 channels, 4 groups of workgroup partials); fixed order: group sums, then the four groups
static __global__ void k_channel_means(const double* __restrict__ partial, int nblocks, int N, double inv_npix,
                                       double* __restrict__ mean) {
    __shared__ double sh[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x, grp = threadIdx.y;
    double s = 0.0;
    if (c < N) {
        const int per = (nblocks + 3) / 4, b1 = (grp + 1) * per < nblocks ? (grp + 1) * per : nblocks;
        double s0 = 0.0, s1 = 0.0;
        int b = grp * per;
        for (; b + 1 < b1; b += 2) { s0 += partial[(size_t)b * N + c]; s1 += partial[(size_t)(b + 1) * N + c]; }
        if (b < b1) s0 += partial[(size_t)b * N + c];
        s = s0 + s1;
    }
    sh[grp][threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && c < N) mean[c] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x])) * inv_npix;
}

// Covariance partials on the fp64 matrix cores.  cov[a][b] = sum_p x[p][a] x[p][b], x = cube - mean: a rank-4
// update per v_mfma_f64_16x16x4_f64 (A: 16 channels x 4 pixels, lane l holds x[p0 + (l >> 4)][a0 + (l & 15)]; B the
// same for the b block; D: col = l & 15, row = (l >> 4) + 4 reg).  One wave owns a (16 MA)^2 block of the matrix
// (MA^2 accumulators of 8 VGPRs) for one slice of the pixels; operands come straight from global memory (each wave's
// loads are 64-byte runs along the channel axis, shared through L1/L2 by the waves of the same pixel slice).  Only
// blocks with bi <= bj are launched; partial[slice][a][b] are summed in slice order afterwards (deterministic).
typedef double fb_d4 __attribute__((ext_vector_type(4)));
// EDGE: N is not a multiple of the block (grids that are not powers of two): channels past N read channel N - 1, count as 0
// and are not stored.
template <typename T, int MA, bool EDGE = false>
__global__ __launch_bounds__(64) void k_channel_cov(const T* __restrict__ cube, const double* __restrict__ mean,
                                                    double* __restrict__ partial, long long npix, int N, int nbs) {
    // blockIdx.x -> (bi <= bj) pair, blockIdx.y -> pixel slice
    int bi = 0, rem = blockIdx.x;
    while (rem >= nbs - bi) { rem -= nbs - bi; ++bi; }
    const int bj = bi + rem;
    const int lane = threadIdx.x, lc = lane & 15, lk = lane >> 4;
    const int a0 = bi * 16 * MA, b0 = bj * 16 * MA;
    const long long per = ((npix + gridDim.y - 1) / gridDim.y + 3) & ~3LL;
    const long long p0 = (long long)blockIdx.y * per, p1 = p0 + per < npix ? p0 + per : npix;
    double ma[MA], mb[MA];
    int ca[MA], cb[MA];                    // channel of this lane in block row i / block column i
    bool oka[MA], okb[MA];
#pragma unroll
    for (int i = 0; i < MA; ++i) {
        ca[i] = a0 + 16 * i + lc; cb[i] = b0 + 16 * i + lc;
        oka[i] = !EDGE || ca[i] < N; okb[i] = !EDGE || cb[i] < N;
        if (EDGE) { ca[i] = ca[i] < N ? ca[i] : N - 1; cb[i] = cb[i] < N ? cb[i] : N - 1; }
        ma[i] = mean[ca[i]]; mb[i] = mean[cb[i]];
    }
    fb_d4 acc[MA][MA];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < MA; ++j) acc[i][j] = fb_d4{0.0, 0.0, 0.0, 0.0};
    // Software pipeline: the raw operands of the NEXT group of KU rank-4 updates are fetched before the matrix
    // instructions of the current group are issued, so that memory latency hides behind 16 MA^2/4 ... KU of them.
    constexpr int KU = 4;
    T fa[2][KU][MA], fb_[2][KU][MA];
    auto fetch = [&](int buf, long long p) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const long long row = p + 4 * u + lk;
            const T* src = cube + (row < p1 ? row : p0) * N;       // past the slice: any valid row, masked below
#pragma unroll
            for (int i = 0; i < MA; ++i) { fa[buf][u][i] = src[ca[i]]; fb_[buf][u][i] = src[cb[i]]; }
        }
    };
    auto update = [&](int buf, long long p) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const bool ok = p + 4 * u + lk < p1;
            double av[MA], bv[MA];
#pragma unroll
            for (int i = 0; i < MA; ++i) {
                av[i] = (ok && oka[i]) ? (double)fa[buf][u][i] - ma[i] : 0.0;
                bv[i] = (ok && okb[i]) ? (double)fb_[buf][u][i] - mb[i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < MA; ++i)
#pragma unroll
                for (int j = 0; j < MA; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    fetch(0, p0);
    for (long long p = p0; p < p1; p += 8 * KU) {
        fetch(1, p + 4 * KU);
        update(0, p);
        if (p + 4 * KU < p1) {
            fetch(0, p + 8 * KU);
            update(1, p + 4 * KU);
        }
    }
    double* dst = partial + (size_t)blockIdx.y * N * N;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < MA; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!EDGE || (a0 + 16 * i + lk + 4 * r < N && b0 + 16 * j + lc < N))
                    dst[(size_t)(a0 + 16 * i + lk + 4 * r) * N + b0 + 16 * j + lc] = acc[i][j][r];
}
// cov[a][b] = cov[b][a] = sum over slices / (npix - 1) for a's block <= b's block (np.cov's divisor)
static __global__ void k_cov_finish(const double* __restrict__ partial, int nslices, int N, int bs, double inv,
                                    double* __restrict__ cov) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (b >= N || a / bs > b / bs) return;
    double s = 0.0;
    for (int q = 0; q < nslices; ++q) s += partial[((size_t)q * N + a) * N + b];
    s *= inv;
    cov[(size_t)a * N + b] = s;
    if (a / bs != b / bs) cov[(size_t)b * N + a] = s;
}

// cleaned[p][c] = x[p][c] - sum_m U[c][m] amp_m(p), amp_m(p) = sum_c U[c][m] x[p][c], x = cube - mean
// (filters.py:170-176).  One wave per pixel; U = [N][nm] fp64 (L1/L2 resident); optional amps_out[nm][npix].
template <typename T, int CPL>          // CPL = channels per lane = max(1, N / 64)
__global__ __launch_bounds__(256) void k_pca_clean(const T* __restrict__ cube, const double* __restrict__ mean,
                                                   const double* __restrict__ U, int nm, T* __restrict__ out,
                                                   double* __restrict__ amps_out, long long npix, int N) {
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= npix) return;
    double x[CPL], rec[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane + 64 * q;
        x[q] = c < N ? (double)cube[p * N + c] - mean[c] : 0.0;
        rec[q] = 0.0;
    }
    for (int m = 0; m < nm; ++m) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < CPL; ++q) { const int c = lane + 64 * q; if (c < N) a += U[(size_t)c * nm + m] * x[q]; }
        a = wave_sum(a);
        if (amps_out && lane == 0) amps_out[(size_t)m * npix + p] = a;
#pragma unroll
        for (int q = 0; q < CPL; ++q) { const int c = lane + 64 * q; if (c < N) rec[q] += U[(size_t)c * nm + m] * a; }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) { const int c = lane + 64 * q; if (c < N) out[p * N + c] = (T)(x[q] - rec[q]); }
}

}  // namespace fb
