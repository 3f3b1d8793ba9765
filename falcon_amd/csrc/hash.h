// MurmurHash3_x86_32 of one 4-byte little-endian key (Appleby's public-domain algorithm);
// the feature hash of reference README.md:124-131.  Exact integer arithmetic, so the host
// table (fal_hash_lookup) and the in-kernel hash agree bit for bit.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace fal {
__host__ __device__ static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ static inline uint32_t murmur3_32(uint32_t key, uint32_t seed) {
    uint32_t k = key * 0xcc9e2d51u;
    k = rotl32(k, 15);
    k *= 0x1b873593u;
    uint32_t h = seed ^ k;
    h = rotl32(h, 13);
    h = h * 5u + 0xe6546b64u;
    h ^= 4u;  // key length in bytes
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
}  // namespace fal
