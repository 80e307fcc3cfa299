// rt_seed_schedule.h — the seed schedule: which random number every call site gets.
//
// The reference draws everything from rand 0.7.3 `thread_rng()` (raytracer/src/types.rs:48-55,
// 106-108), which is OS-seeded and cannot be reproduced.  North-star parity is defined for
// "same scene and seed schedule", so this header IS that schedule: one PCG32 stream per
// pixel sample, consumed in the reference's call order (camera.rs:94 time, then per bounce
// constant_medium.rs:78, metal.rs:26 / dielectric.rs:45 / isotropic.rs:32, pdf.rs:73,
// hittable_list.rs:74, rectangles.rs:71-73 / types.rs:154-155 / types.rs:142-143).
// The HIP kernels and the CPU oracle both include it; it holds integer arithmetic and
// single IEEE f32 operations only, so both sides get the same bits.
#ifndef RT_SEED_SCHEDULE_H
#define RT_SEED_SCHEDULE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RTS_HD __host__ __device__
#else
#define RTS_HD
#endif

namespace rts {

// 32-bit finaliser (Murmur3 fmix32).
RTS_HD inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
RTS_HD inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = mix32(a ^ 0x9e3779b9u);
    h = mix32(h ^ (b + 0x7f4a7c15u));
    h = mix32(h ^ (c + 0x165667b1u));
    return h;
}

// PCG32 (O'Neill, XSH-RR 64/32).
struct Rng {
    uint64_t state;
    uint64_t inc;
};

RTS_HD inline uint32_t next_u32(Rng &r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ull + r.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

RTS_HD inline void seed_stream(Rng &r, uint64_t init_state, uint64_t stream) {
    r.state = 0u;
    r.inc = (stream << 1u) | 1u;
    (void)next_u32(r);
    r.state += init_state;
    (void)next_u32(r);
}

// Stream of one pixel sample: key = (seed, pixel index y*W+x, absolute sample index).
RTS_HD inline void path_rng(Rng &r, uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t a = hash3(seed, pixel, sample);
    uint32_t b = hash3(seed ^ 0xa511e9b3u, sample, pixel);
    seed_stream(r, ((uint64_t)a << 32) | b, ((uint64_t)pixel << 24) ^ sample);
}

// Host-side streams (scene generation, Perlin tables, sampler tables): key = (seed, purpose).
RTS_HD inline void host_rng(Rng &r, uint32_t seed, uint32_t purpose) {
    seed_stream(r, ((uint64_t)hash3(seed, purpose, 0x5eedu) << 32) | hash3(purpose, seed, 1u), purpose);
}

// types.rs:48-50 random_real(): uniform in [0,1).  24 random bits (the reference casts an
// f64 sample to f32 and can return exactly 1.0 with p ~ 2^-25; this schedule cannot).
RTS_HD inline float random_real(Rng &r) {
    return (float)(next_u32(r) >> 8) * (1.0f / 16777216.0f);
}

// types.rs:52-55 random_real_range(min,max): rand's UniformFloat = u * (max - min) + min.
RTS_HD inline float random_real_range(Rng &r, float lo, float hi) {
    float u = random_real(r);
    float scale = hi - lo;
    float v = u * scale;
    return v + lo;
}

// types.rs:106-108 random_int(min,max), inclusive bounds.
RTS_HD inline int32_t random_int(Rng &r, int32_t lo, int32_t hi) {
    uint32_t range = (uint32_t)(hi - lo) + 1u;
    uint32_t x = next_u32(r);
    return lo + (int32_t)(((uint64_t)x * (uint64_t)range) >> 32);
}

// Sampler position (sampling.rs:137-163).  The reference keeps one `count` per worker; a
// pixel consumes 2*spp slots (square then disk per sample, camera.rs:196 and :88), so every
// pixel starts at count % spp == 0 and re-draws `jump` at slots 0 and spp.  Here the state
// is a function of (pixel, slot): slot = 2*sample (+1 for the disk lookup).
RTS_HD inline uint32_t sampler_index(uint32_t seed, uint32_t pixel, uint32_t slot, uint32_t sets,
                                     uint32_t spp, const uint32_t *shuffled_indices) {
    // slot < 2*spp, so the quotient is 0 or 1; the set is picked by a multiply-high (no integer
    // division on the device: `random_int(0, MAX-1) % sets` of sampling.rs:139 is not reproducible anyway)
    uint32_t epoch = slot >= spp ? 1u : 0u;
    uint32_t rem = slot - (epoch ? spp : 0u);
    uint32_t set = (uint32_t)(((uint64_t)hash3(seed ^ 0x51ed270bu, pixel, epoch) * (uint64_t)sets) >> 32);
    uint32_t jump = set * spp;
    return jump + shuffled_indices[jump + rem];
}

}  // namespace rts
#endif
