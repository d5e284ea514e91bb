// philox.h -- counter-based random numbers for the chain update.
//
// The reference draws from NumPy's global MT19937 stream inside each worker
// process (mc3/chain.py:180-259).  A stream cannot be shared by thousands of
// lock-stepped walkers, so every draw here is a pure function
//     Philox4x32-10(key = seed; counter = (iteration, chain, slot))
// held entirely in registers.  The counter/key layout equals rocRAND's
// rocrand_state_philox4x32_10 after rocrand_init(seed, subsequence =
// chain | slot << 32, offset = 4*iteration), so the stream is the rocRAND
// Philox stream; the rounds are written out here so that the same few lines
// can be checked against the Random123 known-answer vectors and against the
// CPU checker's independent implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc3hip {

struct Draw2 {
    uint64_t q0, q1;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                              uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t lo0 = 0xD2511F53u * c0;
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0);
        const uint32_t lo1 = 0xCD9E8D57u * c2;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2);
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// two 64-bit draws for (iteration, chain, slot)
__device__ __forceinline__ Draw2 draw2(uint64_t seed, uint64_t iter, uint32_t chain,
                                       uint32_t slot)
{
    uint32_t w[4];
    philox4x32_10((uint32_t)iter, (uint32_t)(iter >> 32), chain, slot, (uint32_t)seed,
                  (uint32_t)(seed >> 32), w);
    Draw2 d;
    d.q0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    d.q1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    return d;
}

// uniform in [0,1) with 53 random bits (np.random.uniform, chain.py:201,259)
__device__ __forceinline__ double u01(uint64_t q) { return (double)(q >> 11) * 0x1.0p-53; }
// uniform in (0,1] (argument of the Box-Muller logarithm)
__device__ __forceinline__ double u01_open(uint64_t q)
{
    return (double)((q >> 11) + 1) * 0x1.0p-53;
}
// integer in [0,n) (np.random.randint, chain.py:197-203,223-227): high half of the
// 128-bit product, bias < n/2^64
__device__ __forceinline__ uint64_t bounded(uint64_t q, uint64_t n) { return __umul64hi(q, n); }
// same from a 32-bit word (chain indices, n <= 2^20: bias < n/2^32)
__device__ __forceinline__ uint32_t bounded32(uint32_t w, uint32_t n) { return __umulhi(w, n); }

}  // namespace mc3hip
