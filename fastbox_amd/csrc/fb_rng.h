// Counter-based device RNG for the throughput mode of realise_density (box.py:174-176
// draws from numpy's sequential legacy stream, which cannot be generated in parallel).
//
// Threefry4x32-20 (Salmon et al. 2011, Random123 constants; known-answer vectors are
// checked in tests/test_rng.py against the host model fastbox_amd/rng.py).  Threefry
// rather than Philox: it needs only 32-bit add/rotate/xor, all full rate on CDNA4,
// while Philox's 32x32 multiplies are quarter rate.
//
// Noise of the stored mode (ix, iy, iz), iz <= N/2:
//   g = ix mod N/2, h = ix >= N/2
//   o = threefry(ctr = (idx_lo, idx_hi, stream, 0), key = (seed_lo, seed_hi, real_lo, real_hi)),
//       idx = (g N + iy) (N/2+1) + iz
//   (a, b) = h ? (o2, o3) : (o0, o1);  u = (word + 0.5) 2^-32
//   (g0, g1) = sqrt(-2 ln u_a) (cos 2 pi u_b, sin 2 pi u_b)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef FB_THREEFRY_ROUNDS
#define FB_THREEFRY_ROUNDS 20     // Random123 default; 12 is the paper's Crush-resistant minimum
#endif

namespace fb {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

__device__ __forceinline__ void threefry4x32_20(const uint32_t (&ctr)[4], const uint32_t (&key)[4], uint32_t (&X)[4]) {
    const uint32_t ks[5] = {key[0], key[1], key[2], key[3], 0x1BD11BDAu ^ key[0] ^ key[1] ^ key[2] ^ key[3]};
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] = ctr[i] + ks[i];
#pragma unroll
    for (int r = 0; r < FB_THREEFRY_ROUNDS; ++r) {
        if ((r & 1) == 0) {
            X[0] += X[1]; X[1] = rotl32(X[1], R[r & 7][0]) ^ X[0];
            X[2] += X[3]; X[3] = rotl32(X[3], R[r & 7][1]) ^ X[2];
        } else {
            X[0] += X[3]; X[3] = rotl32(X[3], R[r & 7][0]) ^ X[0];
            X[2] += X[1]; X[1] = rotl32(X[1], R[r & 7][1]) ^ X[2];
        }
        if ((r & 3) == 3) {
            const int s = (r + 1) >> 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) X[i] += ks[(s + i) % 5];
            X[3] += (uint32_t)s;
        }
    }
}

// B independent blocks, round-major, so that the B dependency chains interleave in the
// instruction stream (each round is a 3-instruction serial chain per half block)
template <int B>
__device__ __forceinline__ void threefry4x32_20_batch(const uint32_t (&ctr)[B][4], const uint32_t (&key)[4],
                                                      uint32_t (&X)[B][4]) {
    const uint32_t ks[5] = {key[0], key[1], key[2], key[3], 0x1BD11BDAu ^ key[0] ^ key[1] ^ key[2] ^ key[3]};
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) X[b][i] = ctr[b][i] + ks[i];
#pragma unroll
    for (int r = 0; r < FB_THREEFRY_ROUNDS; ++r) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if ((r & 1) == 0) {
                X[b][0] += X[b][1]; X[b][1] = rotl32(X[b][1], R[r & 7][0]) ^ X[b][0];
                X[b][2] += X[b][3]; X[b][3] = rotl32(X[b][3], R[r & 7][1]) ^ X[b][2];
            } else {
                X[b][0] += X[b][3]; X[b][3] = rotl32(X[b][3], R[r & 7][0]) ^ X[b][0];
                X[b][2] += X[b][1]; X[b][1] = rotl32(X[b][1], R[r & 7][1]) ^ X[b][2];
            }
            if ((r & 3) == 3) {
                const int s = (r + 1) >> 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) X[b][i] += ks[(s + i) % 5];
                X[b][3] += (uint32_t)s;
            }
        }
    }
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

struct RngKey { uint32_t k[4]; };

// the two complex normals of generator mode `idx`: (z0 for ix < N/2, z1 for ix >= N/2)
template <typename T>
__device__ __forceinline__ void mode_noise_pair(unsigned long long idx, uint32_t stream, const RngKey& key,
                                                T& a0, T& a1, T& b0, T& b1) {
    const uint32_t ctr[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u};
    uint32_t o[4];
    threefry4x32_20(ctr, key.k, o);
    box_muller(o[0], o[1], a0, a1);
    box_muller(o[2], o[3], b0, b1);
}

// Small-scale velocity noise of the redshift-space remap (stream 1): element idx of the (N,N,N)
// grid takes output idx & 3 of call idx >> 2, so four consecutive line-of-sight cells share one call.
template <typename T>
__device__ __forceinline__ T stream_noise_at(unsigned long long idx, uint32_t stream, const RngKey& key) {
    T g0, g1, g2, g3;
    mode_noise_pair<T>(idx >> 2, stream, key, g0, g1, g2, g3);
    const int r = (int)(idx & 3ull);
    return r == 0 ? g0 : (r == 1 ? g1 : (r == 2 ? g2 : g3));
}
template <typename T>
__device__ __forceinline__ T los_noise_at(unsigned long long idx, const RngKey& key) {
    return stream_noise_at<T>(idx, 1u, key);
}

}  // namespace fb
