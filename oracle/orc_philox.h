/* TEST INFRASTRUCTURE (oracle) -- not part of the product.
 * Philox4x32-10 counter-based RNG (Salmon et al., SC'11), the build's RNG spec
 * (include/lsim.h "lsim_rng_tag").  The reference uses torch's global generator
 * (SURVEY.md 8a quirk 12); parity fixtures inject these uniforms into the reference. */
#ifndef ORC_PHILOX_H
#define ORC_PHILOX_H
#include <stdint.h>

static inline void orc_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* uniform in [0,1): draw `idx` of site `tag` for (env, step) */
static inline float orc_u01(uint32_t seed, uint32_t rank, uint32_t env, uint32_t step, uint32_t tag, uint32_t idx) {
    uint32_t c[4] = {env, step, tag, idx >> 2};
    orc_philox4x32_10(c, seed, rank);
    return (float)(c[idx & 3u] >> 8) * (1.0f / 16777216.0f);
}
#endif
