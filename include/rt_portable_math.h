// rt_portable_math.h — f32 elementary functions built only from + - * / sqrt and bit moves.
//
// The reference calls libm through f32::{sin,cos,acos,atan2,ln} (checker_texture.rs:38-40,
// noise_texture.rs:30, objects/sphere.rs:34-35, types.rs:146-148, constant_medium.rs:78).
// glibc's and ROCm OCML's versions differ in the last ulp, which makes paths diverge.
// These versions give the SAME bits on the host and on gfx950 when compiled without
// FMA contraction (every operation below is a single IEEE-754 f32 operation), so a CPU
// render and a GPU render of one seed can be compared bit for bit.  Polynomials are the
// classic Cephes single-precision ones (Moshier, netlib cephes/single); accuracy is
// checked against glibc in tests/test_portable_math.py (<= 4 ulp on the ranges used).
#ifndef RT_PORTABLE_MATH_H
#define RT_PORTABLE_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace pm {

RT_HD inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RT_HD inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define PM_PI    3.14159265358979323846f
#define PM_PIO2  1.57079632679489661923f
#define PM_PIO4  0.78539816339744830962f
#define PM_FOPI  1.27323954473516f          /* 4/pi */
#define PM_DP1   0.78515625f
#define PM_DP2   2.4187564849853515625e-4f
#define PM_DP3   3.77489497744594108e-8f

RT_HD inline float sin_poly(float x, float z) {
    return (((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x) + x;
}
RT_HD inline float cos_poly(float z) {
    float y = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

// sin and cos together (Rust's sin_cos(), sampling.rs:100, transform.rs:67).
RT_HD inline void sincosf_(float xx, float *s, float *c) {
    if (!(xx == xx) || !(fabsf(xx) < 1.0e8f)) {   // NaN / inf / beyond the reduction's range
        float r = (xx == xx && fabsf(xx) < INFINITY) ? 0.0f : NAN;
        *s = r; *c = (r == r) ? 1.0f : r;
        return;
    }
    float x = fabsf(xx);
    uint32_t j = (uint32_t)(PM_FOPI * x);
    float y = (float)j;
    if (j & 1u) { j += 1u; y = y + 1.0f; }
    j &= 7u;
    float ssign = (xx < 0.0f) ? -1.0f : 1.0f;
    float csign = 1.0f;
    if (j > 3u) { j -= 4u; ssign = -ssign; csign = -csign; }
    if (j > 1u) csign = -csign;
    x = ((x - y * PM_DP1) - y * PM_DP2) - y * PM_DP3;
    float z = x * x;
    float ps = sin_poly(x, z), pc = cos_poly(z);
    bool swap = (j == 1u || j == 2u);
    *s = ssign * (swap ? pc : ps);
    *c = csign * (swap ? ps : pc);
}
RT_HD inline float sinf_(float x) { float s, c; sincosf_(x, &s, &c); return s; }
RT_HD inline float cosf_(float x) { float s, c; sincosf_(x, &s, &c); return c; }

RT_HD inline float asinf_(float xx) {
    float a = fabsf(xx);
    if (!(a <= 1.0f)) return NAN;
    float x, z;
    bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); x = sqrtf(z); }
    else     { x = a; z = x * x; }
    z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z
         + 1.6666752422E-1f) * z * x + x;
    if (big) { z = z + z; z = PM_PIO2 - z; }
    return (xx < 0.0f) ? -z : z;
}

RT_HD inline float acosf_(float x) {
    if (!(fabsf(x) <= 1.0f)) return NAN;       // f32::acos: NaN outside [-1,1]
    if (x > 0.5f)  return 2.0f * asinf_(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return PM_PI - 2.0f * asinf_(sqrtf(0.5f * (1.0f + x)));
    return PM_PIO2 - asinf_(x);
}

RT_HD inline float atanf_(float xx) {
    float x = fabsf(xx), y;
    if (x > 2.414213562373095f)       { y = PM_PIO2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PM_PIO4; x = (x - 1.0f) / (x + 1.0f); }
    else                              { y = 0.0f; }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x);
    return (xx < 0.0f) ? -y : y;
}

RT_HD inline float atan2f_(float y, float x) {
    if (!(x == x) || !(y == y)) return NAN;
    bool yneg = (f2u(y) >> 31) != 0, xneg = (f2u(x) >> 31) != 0;
    if (x == 0.0f) {
        if (y == 0.0f) return xneg ? (yneg ? -PM_PI : PM_PI) : y;
        return yneg ? -PM_PIO2 : PM_PIO2;
    }
    if (y == 0.0f) return xneg ? (yneg ? -PM_PI : PM_PI) : y;
    float z = atanf_(y / x);
    if (x < 0.0f) z = yneg ? (z - PM_PI) : (z + PM_PI);
    return z;
}

// Natural log; ln(0) = -inf as f32::ln (constant_medium.rs:78 relies on it).
RT_HD inline float logf_(float x) {
    if (!(x == x) || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (!(x < INFINITY)) return x;
    int e = 0;
    uint32_t b = f2u(x);
    if ((b >> 23) == 0u) { x = x * 33554432.0f; b = f2u(x); e = -25; }   // subnormal
    e += (int)(b >> 23) - 126;
    x = u2f((b & 0x007fffffu) | 0x3f000000u);                              // mantissa in [0.5,1)
    if (x < 0.707106781186547524f) { e -= 1; x = x + x - 1.0f; }
    else                           { x = x - 1.0f; }
    float z = x * x;
    float y = ((((((((7.0376836292E-2f * x - 1.1514610310E-1f) * x + 1.1676998740E-1f) * x
                  - 1.2420140846E-1f) * x + 1.4249322787E-1f) * x - 1.6668057665E-1f) * x
               + 2.0000714765E-1f) * x - 2.4999993993E-1f) * x + 3.3333331174E-1f) * x * z;
    float fe = (float)e;
    y = y + (-2.12194440e-4f * fe);
    y = y + (-0.5f * z);
    z = x + y;
    z = z + 0.693359375f * fe;
    return z;
}

// Rust `as i32` on f32: saturating, NaN -> 0 (image_texture.rs:66-67, perlin.rs:89-91).
RT_HD inline int32_t f2i_sat(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
}

// f32::is_normal (main.rs:1369): not zero, subnormal, infinite or NaN.
RT_HD inline bool is_normal_(float f) {
    uint32_t ex = (f2u(f) >> 23) & 0xffu;
    return ex != 0u && ex != 0xffu;
}

}  // namespace pm
#endif
