// wavelet-likelihood chain kernels, 1 wave(s) per chain (k_wavelet_impl.h): polynomials of 2..4
// coefficients; design matrices and the general degree are in k_wavelet_w1b.hip
#define MC3_WW 1
#define MC3_WL_TAG w1
#define MC3_MT_SPLIT
#define MC3_WL_TAG2 w1b
#include "k_wavelet_impl.h"
