// double instantiation of the FFT passes
#define FB_REAL double
#define FB_SUFFIX f64
#include "fb_fft_launch.inc"
