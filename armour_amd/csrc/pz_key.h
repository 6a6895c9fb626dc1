// The monomial key of the device-side polynomial-zonotope arithmetic.
//
// The reference packs a monomial's degrees LSB first into a u64 -- k (n x 2 bit), qde, qdae, qddae (n x 1 bit each), cosqe, sinqe (n x 2 bit
// each): 9 n bits, so 7 factors fill it (RT/PZsparse.h:23-40) -- and multiplies monomials by plain integer addition of keys
// (RT/PZsparse.cu:938-940).  The shipped library is that: pzkey_t = uint64_t, ARMOUR_MAX_FACTORS = 7.  A build with -DARMOUR_KEY128
// (make k128: libarmour_hip_k128.so, a second ABI -- include/armour_types.h) carries the same packing in 128 bits and holds 8 factors, the
// "8-DOF" arm of BASELINE configs[4] that the reference's own key cannot represent.  Everything that touches a key goes through this type:
// comparisons, sums and shifts are the integer operators either way; what differs is the lane-to-lane broadcast and the LDS / arena bytes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#if defined(ARMOUR_KEY128)
typedef unsigned __int128 pzkey_t;
#else
typedef uint64_t pzkey_t;
#endif

#define PZKEY_MAX (~(pzkey_t)0)
__host__ __device__ inline pzkey_t pzkey_bit(int s) { return (pzkey_t)1 << s; }

// lane l's key, in every lane (l wave-uniform)
__device__ inline pzkey_t pzkey_readlane(pzkey_t v, int l) {
    const uint64_t lo = (uint64_t)v;
    pzkey_t r = (pzkey_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(lo >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)lo, l));
#if defined(ARMOUR_KEY128)
    const uint64_t hi = (uint64_t)(v >> 64);
    r |= (pzkey_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(hi >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)hi, l)) << 64;
#endif
    return r;
}
