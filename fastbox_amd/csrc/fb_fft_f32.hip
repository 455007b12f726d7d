// float instantiation of the FFT passes
#define FB_REAL float
#define FB_SUFFIX f32
#include "fb_fft_launch.inc"
