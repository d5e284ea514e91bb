// wavelet-likelihood chain kernels for more than 2^14 points (k_wavelet_impl.h)
#define MC3_WL_LARGE 1
#define MC3_WL_TAG large
#include "k_wavelet_impl.h"
