// wavelet-likelihood chain kernels, 8 wave(s) per chain (k_wavelet_impl.h)
#define MC3_WW 8
#define MC3_WL_TAG w8
#include "k_wavelet_impl.h"
