// double instantiation of the FFT passes
#define FB_REAL double
#define FB_SUFFIX f64
#define FB_DEFINE_SLAB_PERMUTE 1
#include "fb_fft_launch.inc"
