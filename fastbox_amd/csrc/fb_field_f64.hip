// double instantiation of the element-wise kernels (+ the precision-independent bin count)
#define FB_REAL double
#define FB_SUFFIX f64
#define FB_DEFINE_COMMON 1
#include "fb_field_launch.inc"
