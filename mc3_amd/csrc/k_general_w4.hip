// Gaussian-likelihood chain kernels, 4 wave(s) per chain (k_general_impl.h)
#define MC3_WPC 4
#include "k_general_impl.h"
