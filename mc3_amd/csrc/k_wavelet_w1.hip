// wavelet-likelihood chain kernels, 1 wave(s) per chain (k_wavelet_impl.h)
#define MC3_WW 1
#define MC3_WL_TAG w1
#include "k_wavelet_impl.h"
