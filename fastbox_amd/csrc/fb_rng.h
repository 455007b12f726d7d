// Counter-based device RNG for the throughput mode of realise_density (box.py:174-176
// draws from numpy's sequential legacy stream, which cannot be generated in parallel).
//
// Philox4x32-10 (Salmon et al. 2011, Random123 constants; known-answer vectors are checked in
// tests/test_rng.py against the host model fastbox_amd/rng.py).  On gfx950 the 32 x 32 -> 64 bit
// multiply-add v_mad_u64_u32 issues at (nearly) the rate of a 32-bit add (tools/valu_rates.hip,
// profiles/r02_valu_rates.txt: 5.4 against 4.8 cycles per wave-instruction), so a Philox round is
// 2 multiplies + 4 xors per 128 bits with the round keys in scalar registers: one 512^3 box's
// 33.7 M calls take 49 us of the chip's vector time against 122 us for Threefry4x32-20, the
// generator of round 1 (71 us with 12 rounds).
//
// Call:  o = philox4x32_10(ctr = (idx_lo, idx_hi | stream << 24, real_lo, real_hi), key = (seed_lo, seed_hi))
//        u = (word + 0.5) 2^-32;  (g0, g1) = sqrt(-2 ln u_a) (cos 2 pi u_b, sin 2 pi u_b)
//
// Noise of the stored mode (ix, iy, iz), iz <= N/2, of a half spectrum (stream 0):
//   g = ix mod N/2, h = ix >= N/2;  idx = (g N + iy) (N/2+1) + iz;  (a, b) = h ? (o2, o3) : (o0, o1)
//   0 < iz < N/2 :  z = (g0 + i g1) / sqrt 2
//   iz = 0, N/2  :  the plane is its own mirror image, delta(-k) = conj delta(k), and is drawn Hermitian:
//                   the mode with  iy in (0, N/2),  or  iy in {0, N/2} and ix in (0, N/2),  is drawn as above;
//                   its mirror image ((N - ix) mod N, (N - iy) mod N) is the complex conjugate of that draw;
//                   the four self-mirrored modes (ix, iy in {0, N/2}) are real:  z = g0.
// (Round 1 drew the planes like every other mode, with twice the variance, and left the projection on the
// Hermitian part to the c2r pass, which drops Im of k_z = 0, N/2 -- the same distribution.  Drawn Hermitian,
// the two planes of a half spectrum can share one complex plane, see `packed` in fb_fft_kernels.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef FB_PHILOX_ROUNDS
#define FB_PHILOX_ROUNDS 10     // Random123 default (7 is the paper's Crush-resistant minimum)
#endif

namespace fb {

struct RngKey { uint32_t k[4]; };     // (seed_lo, seed_hi, realisation_lo, realisation_hi)

// B independent blocks, round-major, so that the B dependency chains interleave in the instruction stream.
// k0, k1 are wave-uniform: the round keys live in scalar registers.
template <int B>
__device__ __forceinline__ void philox4x32_batch(uint32_t (&X)[B][4], uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < FB_PHILOX_ROUNDS; ++r) {
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
// counter of call idx of a stream
__device__ __forceinline__ void philox_counter(unsigned long long idx, uint32_t stream, const RngKey& key, uint32_t (&c)[4]) {
    c[0] = (uint32_t)idx; c[1] = (uint32_t)(idx >> 32) | (stream << 24); c[2] = key.k[2]; c[3] = key.k[3];
}
__device__ __forceinline__ void philox4x32(unsigned long long idx, uint32_t stream, const RngKey& key, uint32_t (&o)[4]) {
    uint32_t X[1][4];
    philox_counter(idx, stream, key, X[0]);
    philox4x32_batch<1>(X, key.k[0], key.k[1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = X[0][i];
}

// Box-Muller.  float: hardware log2 / sqrt / sin / cos (v_sin_f32 takes revolutions).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& g0, float& g1) {
    const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
#ifdef FB_EXPERIMENT_NOBM     // knock-out build (tools/knockout.sh): no transcendentals
    g0 = u1; g1 = u2;
#else
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln2 log2(u1)
    g0 = r * __builtin_amdgcn_cosf(u2);
    g1 = r * __builtin_amdgcn_sinf(u2);
#endif
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, double& g0, double& g1) {
    const double u1 = ((double)a + 0.5) * 2.3283064365386963e-10;
    const double u2 = ((double)b + 0.5) * 2.3283064365386963e-10;
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    g0 = r * c; g1 = r * s;
}

// the four normals of call idx of a stream
template <typename T>
__device__ __forceinline__ void stream_normals4(unsigned long long idx, uint32_t stream, const RngKey& key,
                                                T& a0, T& a1, T& b0, T& b1) {
    uint32_t o[4];
    philox4x32(idx, stream, key, o);
    box_muller(o[0], o[1], a0, a1);
    box_muller(o[2], o[3], b0, b1);
}

// Element idx of a stream of single normals takes output idx & 3 of call idx >> 2 (four consecutive
// elements share one call): small-scale velocities of the redshift-space remap (stream 1), spectral-index map (3).
template <typename T>
__device__ __forceinline__ T stream_noise_at(unsigned long long idx, uint32_t stream, const RngKey& key) {
    T g0, g1, g2, g3;
    stream_normals4<T>(idx >> 2, stream, key, g0, g1, g2, g3);
    const int r = (int)(idx & 3ull);
    return r == 0 ? g0 : (r == 1 ? g1 : (r == 2 ? g2 : g3));
}
template <typename T>
__device__ __forceinline__ T los_noise_at(unsigned long long idx, const RngKey& key) {
    return stream_noise_at<T>(idx, 1u, key);
}

// ---- half-spectrum noise (stream 0), see the header comment -------------------------------------
// Where the draw of stored mode (ix, iy, iz) comes from: the mode itself or, on the planes iz = 0, N/2,
// possibly its mirror image (then `conj`).  `real_only`: one of the four self-mirrored modes of a plane.
struct ModeDraw { int ix, iy; bool conj, real_only; };
__device__ __forceinline__ ModeDraw mode_draw(int ix, int iy, int iz, int N) {
    const int H = N >> 1;
    ModeDraw d{ix, iy, false, false};
    if (iz == 0 || iz == H) {
        const bool ys = (iy == 0 || iy == H), xs = (ix == 0 || ix == H);
        d.real_only = ys && xs;
        d.conj = ys ? (ix > H) : (iy > H);
        if (d.conj) { d.ix = ix ? N - ix : 0; d.iy = iy ? N - iy : 0; }      // (N - i) mod N, any even N
    }
    return d;
}
// unit-variance complex noise of one stored mode, times `s` (the caller's amplitude): s z, with
// E |z|^2 = 1 for every mode (z = (g0 + i g1)/sqrt 2, or g0 alone where the mode is real)
template <typename T>
__device__ __forceinline__ void mode_noise(int ix, int iy, int iz, int N, int NZV, const RngKey& key, T s, T& re, T& im) {
    const ModeDraw d = mode_draw(ix, iy, iz, N);
    const int H = N >> 1;
    const int g = d.ix >= H ? d.ix - H : d.ix;          // d.ix mod N/2 (d.ix < N)
    const unsigned long long idx = ((unsigned long long)g * N + d.iy) * NZV + iz;
    T a0, a1, b0, b1;
    stream_normals4<T>(idx, 0u, key, a0, a1, b0, b1);
    const bool hi = d.ix >= H;
    const T x = hi ? b0 : a0, y = hi ? b1 : a1;
    if (d.real_only) { re = s * x; im = (T)0; }
    else { const T q = s * (T)0.70710678118654752440; re = q * x; im = d.conj ? -(q * y) : q * y; }
}

}  // namespace fb
