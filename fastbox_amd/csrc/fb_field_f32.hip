// float instantiation of the element-wise kernels
#define FB_REAL float
#define FB_SUFFIX f32
#include "fb_field_launch.inc"
