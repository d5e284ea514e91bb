// Gaussian-likelihood chain kernels, 16 wave(s) per chain (k_general_impl.h)
#define MC3_WPC 16
#include "k_general_impl.h"
