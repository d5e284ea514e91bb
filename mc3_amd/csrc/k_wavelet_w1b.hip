// wavelet-likelihood chain kernels, 1 wave(s) per chain (k_wavelet_impl.h): design matrices and
// polynomials of the general degree (the sister unit of k_wavelet_w1.hip)
#define MC3_WW 1
#define MC3_WL_TAG w1b
#define MC3_MT_SPLIT
#define MC3_MT_SECOND
#include "k_wavelet_impl.h"
