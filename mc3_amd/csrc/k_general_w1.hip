// Gaussian-likelihood chain kernels, 1 wave(s) per chain (k_general_impl.h)
#define MC3_WPC 1
#include "k_general_impl.h"
