// Stockham auto-sort FFT building blocks for gfx950 (wave64, LDS exchange).
//
// Replaces numpy.fft.ifftn / fftn on the reference hot path
// (fastbox/box.py:187, 193, 246, 337, 380, 654, 736).
//
// Model: a line of N points is owned by TPL = N/E threads; thread t keeps the
// E points  x[t + e*TPL]  (e = 0..E-1) in registers.  A stage of radix R with
// p = product of the radices already done performs, per thread, E/R
// butterflies i = t + m*TPL:
//      u[q]  = x[i + q*N/R] * w_{pR}^{q (i mod p)}        (registers)
//      u     = DFT_R(u)
//      y[(i - i mod p) R + (i mod p) + q p] = u[q]        (LDS scatter)
// and every thread then re-reads y[t + e*TPL].  The register<->position map
// is therefore identical before the first and after the last stage: global
// loads/stores of a pass never need a transposition of their own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fb {

template <typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };

template <typename T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T> __device__ __forceinline__ cx<T> cconj(cx<T> a) { return {a.x, -a.y}; }
template <typename T> __device__ __forceinline__ cx<T> cscale(cx<T> a, T s) { return {a.x * s, a.y * s}; }
// a * (SIGN * i)
template <int SIGN, typename T> __device__ __forceinline__ cx<T> mul_si(cx<T> a) {
    if constexpr (SIGN > 0) return {-a.y, a.x}; else return {a.y, -a.x};
}

// ---- packed fp32 complex primitives -------------------------------------------------------
// A complex float is a VGPR pair, and the VOP3P packed ops take per-half source selectors
// (op_sel / op_sel_hi) and negations (neg_lo / neg_hi), so multiplying by +-i, complex products and
// the w8 rotations need no register shuffles.  hipcc does not find these forms by itself (the plain
// C++ below compiles to ~95 v_mov_b32 per 512-point line), hence the explicit instructions.
typedef float fb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fb_f2 to_f2(cx<float> a) { return __builtin_bit_cast(fb_f2, a); }
__device__ __forceinline__ cx<float> to_cx(fb_f2 a) { return __builtin_bit_cast(cx<float>, a); }
// a + SIGN i b
template <int SIGN> __device__ __forceinline__ cx<float> pk_add_i(cx<float> a, cx<float> b) {
    fb_f2 r;
    if constexpr (SIGN > 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(to_f2(a)), "v"(to_f2(b)));
    else                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(to_f2(a)), "v"(to_f2(b)));
    return to_cx(r);
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ cx<float> pk_add_conj(cx<float> a, cx<float> b) {
    fb_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(to_f2(a)), "v"(to_f2(b)));
    return to_cx(r);
}
__device__ __forceinline__ cx<float> pk_sub_conj(cx<float> a, cx<float> b) {
    fb_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(to_f2(a)), "v"(to_f2(b)));
    return to_cx(r);
}
// a * w (SIGN < 0) or a * conj(w) (SIGN > 0): tables hold forward twiddles
template <int SIGN> __device__ __forceinline__ cx<float> pk_cmul(cx<float> a, cx<float> w) {
    fb_f2 r;
    if constexpr (SIGN < 0) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(to_f2(a)), "v"(to_f2(w)));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(to_f2(a)), "v"(to_f2(w)));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(to_f2(a)), "v"(to_f2(w)));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(to_f2(a)), "v"(to_f2(w)));
    }
    return to_cx(r);
}
// a * exp(SIGN i pi/4) and a * exp(SIGN 3 i pi/4)
template <int SIGN> __device__ __forceinline__ cx<float> pk_rot8(cx<float> a) {
    const fb_f2 c = {0.70710678118654752440f, 0.70710678118654752440f};
    fb_f2 t, r;   // SIGN<0: (x + y, y - x); SIGN>0: (x - y, y + x)
    if constexpr (SIGN < 0) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(to_f2(a)));
    else                    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(to_f2(a)));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(c));
    return to_cx(r);
}
template <int SIGN> __device__ __forceinline__ cx<float> pk_rot83(cx<float> a) {
    const fb_f2 c = {0.70710678118654752440f, 0.70710678118654752440f};
    fb_f2 t, r;   // SIGN<0: (-x + y, -y - x); SIGN>0: (-x - y, -y + x)
    if constexpr (SIGN < 0) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(t) : "v"(to_f2(a)));
    else                    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(t) : "v"(to_f2(a)));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(c));
    return to_cx(r);
}

// ---- small DFTs, natural-order output, w = exp(SIGN 2 pi i / R) ------------
template <int SIGN, typename T> __device__ __forceinline__ void dft2(cx<T>& a, cx<T>& b) {
    cx<T> s = a + b; b = a - b; a = s;
}
template <int SIGN, typename T>
__device__ __forceinline__ void dft4(cx<T>& u0, cx<T>& u1, cx<T>& u2, cx<T>& u3) {
    cx<T> t0 = u0 + u2, t1 = u0 - u2, t2 = u1 + u3, t3 = mul_si<SIGN>(u1 - u3);
    u0 = t0 + t2; u1 = t1 + t3; u2 = t0 - t2; u3 = t1 - t3;
}
template <int SIGN, typename T> __device__ __forceinline__ void dft8(cx<T>* u) {
    const T c = (T)0.70710678118654752440;
    dft4<SIGN>(u[0], u[2], u[4], u[6]);      // even -> u0,u2,u4,u6 hold a0..a3
    dft4<SIGN>(u[1], u[3], u[5], u[7]);      // odd  -> u1,u3,u5,u7 hold b0..b3
    cx<T> b0 = u[1];
    cx<T> b1 = {c * (u[3].x - SIGN * u[3].y), c * (SIGN * u[3].x + u[3].y)};
    cx<T> b2 = mul_si<SIGN>(u[5]);
    cx<T> b3 = {c * (-u[7].x - SIGN * u[7].y), c * (SIGN * u[7].x - u[7].y)};
    cx<T> a0 = u[0], a1 = u[2], a2 = u[4], a3 = u[6];
    u[0] = a0 + b0; u[4] = a0 - b0;
    u[1] = a1 + b1; u[5] = a1 - b1;
    u[2] = a2 + b2; u[6] = a2 - b2;
    u[3] = a3 + b3; u[7] = a3 - b3;
}
// float: the same butterflies on the packed primitives (28 instructions for the radix-8 one)
template <int SIGN>
__device__ __forceinline__ void dft4(cx<float>& u0, cx<float>& u1, cx<float>& u2, cx<float>& u3) {
    const cx<float> t0 = u0 + u2, t1 = u0 - u2, t2 = u1 + u3, d = u1 - u3;
    u0 = t0 + t2; u2 = t0 - t2;
    u1 = pk_add_i<SIGN>(t1, d); u3 = pk_add_i<-SIGN>(t1, d);
}
template <int SIGN> __device__ __forceinline__ void dft8(cx<float>* u) {
    dft4<SIGN>(u[0], u[2], u[4], u[6]);
    dft4<SIGN>(u[1], u[3], u[5], u[7]);
    const cx<float> a0 = u[0], a1 = u[2], a2 = u[4], a3 = u[6];
    const cx<float> b0 = u[1], b1 = pk_rot8<SIGN>(u[3]), b2 = u[5], b3 = pk_rot83<SIGN>(u[7]);
    u[0] = a0 + b0; u[4] = a0 - b0;
    u[1] = a1 + b1; u[5] = a1 - b1;
    u[2] = pk_add_i<SIGN>(a2, b2); u[6] = pk_add_i<-SIGN>(a2, b2);
    u[3] = a3 + b3; u[7] = a3 - b3;
}
// u * W (forward table entry w; the inverse transform uses its conjugate)
template <int SIGN, typename T> __device__ __forceinline__ cx<T> twmul(cx<T> u, cx<T> w) {
    if constexpr (SIGN > 0) w.y = -w.y;
    return cmul(u, w);
}
template <int SIGN> __device__ __forceinline__ cx<float> twmul(cx<float> u, cx<float> w) { return pk_cmul<SIGN>(u, w); }

template <int R, int SIGN, typename T> __device__ __forceinline__ void dft(cx<T>* u) {
    static_assert(R == 2 || R == 4 || R == 8, "radix");
    if constexpr (R == 2) dft2<SIGN>(u[0], u[1]);
    else if constexpr (R == 4) dft4<SIGN>(u[0], u[1], u[2], u[3]);
    else dft8<SIGN>(u);
}

// ---- buffer (SRSRC) access: wave-uniform 64-bit base in scalar registers + one 32-bit
// per-lane byte offset.  Offsets at or beyond FB_BUF_RANGE are dropped by the hardware range
// check (loads return 0, stores are ignored), which doubles as the column-validity predicate.
#define FB_BUF_RANGE 0xFFFFFFF0u
#define FB_BUF_OOB 0xFFFFFFF8u
typedef unsigned int fb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int fb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)FB_BUF_RANGE, 0x00020000);
}
// soff: wave-uniform byte offset (a scalar register of the instruction; not part of the range check)
template <int AUX = 0> __device__ __forceinline__ cx<float> buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const cx<float>*) {
    return __builtin_bit_cast(cx<float>, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX));
}
template <int AUX = 0> __device__ __forceinline__ cx<double> buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const cx<double>*) {
    return __builtin_bit_cast(cx<double>, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
// AUX: cache policy bits of the instruction (0 default, 2 = nt: streaming data that nothing re-reads soon)
template <int AUX = 0> __device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, cx<float> v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fb_u32x2, v), r, (int)voff, (int)soff, AUX);
}
template <int AUX = 0> __device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, cx<double> v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fb_u32x4, v), r, (int)voff, (int)soff, AUX);
}

constexpr int fb_min(int a, int b) { return a < b ? a : b; }
constexpr int fb_max(int a, int b) { return a > b ? a : b; }

// elements per thread for an n-point line
// (16 from length 1024 in single precision: the 64 threads of a line are then one wavefront and the z pass of a 2048^3 box
// exchanges without workgroup barriers -- 85.3 -> 82.3 ms per step; fp64 keeps 8: the registers)
#ifndef FB_CONTIG_E16_FROM
#define FB_CONTIG_E16_FROM 1024
#endif
template <typename T> constexpr int elems_per_thread(int n) { return (sizeof(T) == 4 && n >= (FB_CONTIG_E16_FROM)) ? 16 : fb_min(8, n); }

// ---- LDS exchange layouts --------------------------------------------------
// Strided-axis passes: tile[pos][col], col fastest.  A 16-lane (fp32) or
// 8-lane (fp64) group always touches one contiguous 128-byte row, so both the
// scattered writes and the linear reads are bank-conflict free.
template <typename T, int TZ> struct TileLayout {
    static constexpr bool split = false;
    cx<T>* base; int col;
    __device__ __forceinline__ cx<T>& at(int pos) const { return base[pos * TZ + col]; }
};
// The same tile at half the LDS: real parts and imaginary parts go through it one after the other (a thread keeps the
// half that is not on its way in registers), so that a CU holds twice as many tiles -- or workgroups half the size.
// SW (16 four-byte columns: a row is 64 bytes, HALF of the 32 banks a 32-lane group of ds_read/write_b32 spans): rows whose
// positions differ by a multiple of 8 -- the first stage's scattered writes -- would all land in the same half; exchanging the two
// halves of every row pair by bit 3 of the position makes a lane group's two rows hit different halves in every stage.
template <typename T, int TZ, bool SW = false> struct SplitTileLayout {
    static constexpr bool split = true;
    T* base; int col;
    __device__ __forceinline__ T& at(int pos) const {
        if constexpr (SW) return base[(pos * TZ + col) ^ (((pos >> 3) & 1) * TZ)];
        else return base[pos * TZ + col];
    }
};
// Contiguous-axis passes: one line per thread group, one pad element every 8
// so that the radix-8 first-stage scatter (lane stride 8 elements) spreads
// over all banks.
template <typename T> struct LineLayout {
    static constexpr bool split = false;
    cx<T>* base;
    static __host__ __device__ constexpr int padded(int n) { return n + (n >> 3) + 1; }
    __device__ __forceinline__ cx<T>& at(int pos) const { return base[pos + (pos >> 3)]; }
};

// ---- the stages --------------------------------------------------------------
// tw: LDS table of forward twiddles W_M^j = exp(-2 pi i j / M), M = N * TWS.
// WAVE: the N/E threads of a line sit in one wavefront (contiguous-axis passes up to 512 complex points), so the
// exchange needs no workgroup barrier -- a wave's LDS instructions execute in issue order; only the compiler has to
// be kept from moving the reads above the writes.
template <bool WAVE> __device__ __forceinline__ void exchange_sync() {
    if constexpr (WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}
// hook(s): called after the s-th exchange (s = 0, 1, ...), i.e. between two stages -- a place for independent work of the
// caller (the resident generator pass trickles the previous tile's stores there)
struct NoStageHook { template <int S> __device__ __forceinline__ void at() const {} };
template <class F, int BASE> struct StageHook {        // calls f(integral_constant<int, BASE + s>) after the s-th exchange
    const F& f;
    template <int S> __device__ __forceinline__ void at() const { f(std::integral_constant<int, BASE + S>{}); }
};
template <typename T, int N, int E, int SIGN, int TWS, int P, class Layout, bool WAVE = false, class Hook = NoStageHook, int STAGE = 0>
__device__ __forceinline__ void fft_stages(cx<T> (&v)[E], const int t, const cx<T>* __restrict__ tw,
                                           const Layout& lds, const Hook& hook = Hook()) {
    constexpr int REM = N / P;
    constexpr int R = REM >= 8 ? 8 : REM;
    constexpr int NB = E / R;
    constexpr int TPL = N / E;
    static_assert(E % R == 0, "radix must divide elements per thread");
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        cx<T> u[R];
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = v[m + q * NB];
        if constexpr (P > 1) {
            const int k = (t + m * TPL) & (P - 1);
#pragma unroll
            for (int q = 1; q < R; ++q) {
                u[q] = twmul<SIGN>(u[q], tw[q * k * (N / (P * R)) * TWS]);
            }
        }
        dft<R, SIGN>(u);
#pragma unroll
        for (int q = 0; q < R; ++q) v[m + q * NB] = u[q];
    }
    if constexpr (P * R < N) {
        if constexpr (Layout::split) {
            // half-size tile: the real parts make the round trip, then the imaginary parts (which wait in v[].y, still in
            // the old distribution, while v[].x already holds the new one)
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                const int i = t + m * TPL;
                const int k = i & (P - 1);
                const int j = (i - k) * R + k;
#pragma unroll
                for (int q = 0; q < R; ++q) lds.at(j + q * P) = v[m + q * NB].x;
            }
            exchange_sync<WAVE>();
#pragma unroll
            for (int e = 0; e < E; ++e) v[e].x = lds.at(t + e * TPL);
            exchange_sync<WAVE>();
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                const int i = t + m * TPL;
                const int k = i & (P - 1);
                const int j = (i - k) * R + k;
#pragma unroll
                for (int q = 0; q < R; ++q) lds.at(j + q * P) = v[m + q * NB].y;
            }
            exchange_sync<WAVE>();
#pragma unroll
            for (int e = 0; e < E; ++e) v[e].y = lds.at(t + e * TPL);
            exchange_sync<WAVE>();
        } else {
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int i = t + m * TPL;
            const int k = i & (P - 1);
            const int j = (i - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; ++q) lds.at(j + q * P) = v[m + q * NB];
        }
        exchange_sync<WAVE>();
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = lds.at(t + e * TPL);
        exchange_sync<WAVE>();
        }
        hook.template at<STAGE>();
        fft_stages<T, N, E, SIGN, TWS, P * R, Layout, WAVE, Hook, STAGE + 1>(v, t, tw, lds, hook);
    }
}

}  // namespace fb
