// Unit check of the packed-fp32 complex primitives (VOP3P op_sel / neg modifiers) used by fb_fft.h.
//   hipcc --offload-arch=gfx950 -O3 tools/pkmath_test.hip -o /tmp/pkmath && /tmp/pkmath
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../fastbox_amd/csrc/fb_fft.h"
using namespace fb;

__global__ void k(const cx<float>* in, cx<float>* out) {
    const int i = threadIdx.x;
    cx<float> a = in[2 * i], b = in[2 * i + 1];
    out[8 * i + 0] = pk_add_i<+1>(a, b);
    out[8 * i + 1] = pk_add_i<-1>(a, b);
    out[8 * i + 2] = pk_cmul<-1>(a, b);
    out[8 * i + 3] = pk_cmul<+1>(a, b);
    out[8 * i + 4] = pk_rot8<-1>(a);
    out[8 * i + 5] = pk_rot8<+1>(a);
    out[8 * i + 6] = pk_rot83<-1>(a);
    out[8 * i + 7] = pk_rot83<+1>(a);
}
int main() {
    const int n = 64;
    cx<float> h[2 * n], o[8 * n], *d, *e;
    for (int i = 0; i < 2 * n; ++i) h[i] = {(float)sin(i * 1.3 + 0.2), (float)cos(i * 0.7 - 0.4)};
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    double worst = 0; const double c = 0.70710678118654752440;
    for (int i = 0; i < n; ++i) {
        double ax = h[2 * i].x, ay = h[2 * i].y, bx = h[2 * i + 1].x, by = h[2 * i + 1].y;
        double want[8][2] = {{ax - by, ay + bx}, {ax + by, ay - bx},                                  // a + i b, a - i b
                             {ax * bx - ay * by, ax * by + ay * bx}, {ax * bx + ay * by, -ax * by + ay * bx},  // a b, a conj(b)
                             {c * (ax + ay), c * (-ax + ay)}, {c * (ax - ay), c * (ax + ay)},          // a w8 (fwd: e^{-i pi/4}), inverse
                             {c * (-ax + ay), c * (-ax - ay)}, {c * (-ax - ay), c * (ax - ay)}};       // a w8^3
        for (int q = 0; q < 8; ++q) {
            worst = fmax(worst, fabs(o[8 * i + q].x - want[q][0]));
            worst = fmax(worst, fabs(o[8 * i + q].y - want[q][1]));
        }
    }
    printf("max abs error %.3g  %s\n", worst, worst < 1e-6 ? "OK" : "FAIL");
    return worst < 1e-6 ? 0 : 1;
}
