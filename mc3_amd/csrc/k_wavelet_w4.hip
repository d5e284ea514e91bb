// wavelet-likelihood chain kernels, 4 wave(s) per chain (k_wavelet_impl.h)
#define MC3_WW 4
#define MC3_WL_TAG w4
#include "k_wavelet_impl.h"
