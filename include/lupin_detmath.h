/*
 * lupin_detmath.h -- the transcendental functions of the path, DEFINED as exact sequences of
 * IEEE-754 double operations rounded once to f32.
 *
 * Why this exists: the reference's WGSL calls sin/cos/atan/atan2/acos/exp/log/pow
 * (pathtracer.wgsl:1450,1519,1619-1628,1908-1916,2094,2220-2221,2583,2601-2603,2732), whose
 * results are whatever the Vulkan driver's lowering gives (unpinned, tolerance of several ulp
 * and an absolute 2^-11 for sin/cos).  A path tracer branches on these values
 * (`rnl < fresnel`, Russian roulette), so two implementations only follow the same paths on the
 * same RNG stream if their transcendentals agree bit for bit.  These functions use nothing but
 * +,-,*,/ ,sqrt, rint, conversions and EXPLICIT fused multiply-adds (LPM_FMA = fma(), one rounding, in
 * the polynomial steps) in double precision -- all correctly rounded on x86-64 and on gfx950 -- and no
 * implicit contraction, so host and device produce identical bits.  Accuracy: |error| < 1e-11 relative
 * before the final rounding, i.e. correctly rounded f32 except for astronomically rare ties -- far
 * inside the precision WGSL guarantees.  (Round 1 evaluated the polynomials as separate multiplies and
 * adds and carried them to 1e-15: on the GPU these double-precision steps were 15 % of the shading
 * kernel's vector instructions, at half rate; fused steps and series cut to what an f32 result can
 * show make them about 2.5 times cheaper.)
 *
 * Both the CPU oracle (oracle/) and the HIP kernels include this header; it is part of the
 * arithmetic specification, not of either implementation.  tests/test_detmath.py pins it
 * against numpy's float64 libm.
 *
 * Compile with -ffp-contract=off (hipcc defaults to fast contraction!); the CPU side wants -mfma so that
 * fma() is one instruction (without it the C library's correctly rounded fma() gives the same bits, slowly).
 */
#ifndef LUPIN_DETMATH_H
#define LUPIN_DETMATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define LPM_FN __host__ __device__ static inline
#else
#define LPM_FN static inline
#endif

#define LPM_FMA(a, b, c) __builtin_fma((a), (b), (c))

LPM_FN uint64_t lpm_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
LPM_FN double   lpm_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
LPM_FN double   lpm_nan(void) { return lpm_u2d(0x7FF8000000000000ull); }
LPM_FN double   lpm_inf(void) { return lpm_u2d(0x7FF0000000000000ull); }
/* 2^e for e in [-1022, 1023] */
LPM_FN double   lpm_pow2i(int e) { return lpm_u2d((uint64_t)(e + 1023) << 52); }

#define LPM_PI      3.14159265358979323846
#define LPM_PIO2    1.57079632679489661923
#define LPM_PIO4    0.78539816339744830962

/* sin and cos of x (double in, double out, |err| < 1e-13 for |x| < 1e5) */
LPM_FN void lpm_sincos_d(double x, double *s_out, double *c_out)
{
    double ax = fabs(x);
    if (!(ax < 1.0e15)) { *s_out = lpm_nan(); *c_out = lpm_nan(); return; }

    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11;  /* pi/2 - PIO2_HI */
    double kd = rint(x * TWO_OVER_PI);
    double r = LPM_FMA(-kd, PIO2_LO, LPM_FMA(-kd, PIO2_HI, x));
    long long k = (long long)kd;
    int q = (int)(k & 3);

    double r2 = r * r;
    /* |r| <= pi/4.  sin r = r + r^3 * S(r^2), Taylor through r^13 (next term < 2e-14 relative) */
    double S = 1.6059043836821614599e-10;              /*  1/13! */
    S = LPM_FMA(S, r2, -2.5052108385441718775e-08);    /* -1/11! */
    S = LPM_FMA(S, r2, 2.7557319223985890653e-06);     /*  1/9!  */
    S = LPM_FMA(S, r2, -1.9841269841269841270e-04);    /* -1/7!  */
    S = LPM_FMA(S, r2, 8.3333333333333333333e-03);     /*  1/5!  */
    S = LPM_FMA(S, r2, -1.6666666666666666667e-01);    /* -1/3!  */
    double sr = LPM_FMA(r * r2, S, r);
    /* cos r = 1 - r^2/2 + r^4 * C(r^2), Taylor through r^14 (next term < 1e-15) */
    double C = -1.1470745597729724714e-11;             /* -1/14! */
    C = LPM_FMA(C, r2, 2.0876756987868098979e-09);     /*  1/12! */
    C = LPM_FMA(C, r2, -2.7557319223985890653e-07);    /* -1/10! */
    C = LPM_FMA(C, r2, 2.4801587301587301587e-05);     /*  1/8!  */
    C = LPM_FMA(C, r2, -1.3888888888888888889e-03);    /* -1/6!  */
    C = LPM_FMA(C, r2, 4.1666666666666666667e-02);     /*  1/4!  */
    double cr = LPM_FMA(r2 * r2, C, LPM_FMA(-0.5, r2, 1.0));

    double s, c;
    if (q == 0)      { s = sr;  c = cr;  }
    else if (q == 1) { s = cr;  c = -sr; }
    else if (q == 2) { s = -sr; c = -cr; }
    else             { s = -cr; c = sr;  }
    *s_out = s; *c_out = c;
}

/* atan(x), |err| < 1e-12 */
LPM_FN double lpm_atan_d(double x)
{
    const double T3P8 = 2.41421356237309504880;  /* tan(3pi/8) */
    const double TP8  = 0.41421356237309504880;  /* tan(pi/8)  */
    double t = fabs(x);
    double base, r;
    if (t > T3P8)     { base = LPM_PIO2; r = -1.0 / t; }
    else if (t > TP8) { base = LPM_PIO4; r = (t - 1.0) / (t + 1.0); }
    else              { base = 0.0;      r = t; }
    double z = r * r;
    /* |r| <= tan(pi/8): sum_{n=0}^{13} (-1)^n z^n / (2n+1), Horner (next term < 4e-13 relative) */
    double p = -1.0 / 27.0;
    p = LPM_FMA(p, z, 1.0 / 25.0);
    p = LPM_FMA(p, z, -1.0 / 23.0);
    p = LPM_FMA(p, z, 1.0 / 21.0);
    p = LPM_FMA(p, z, -1.0 / 19.0);
    p = LPM_FMA(p, z, 1.0 / 17.0);
    p = LPM_FMA(p, z, -1.0 / 15.0);
    p = LPM_FMA(p, z, 1.0 / 13.0);
    p = LPM_FMA(p, z, -1.0 / 11.0);
    p = LPM_FMA(p, z, 1.0 / 9.0);
    p = LPM_FMA(p, z, -1.0 / 7.0);
    p = LPM_FMA(p, z, 1.0 / 5.0);
    p = LPM_FMA(p, z, -1.0 / 3.0);
    p = LPM_FMA(p, z, 1.0);
    double a = LPM_FMA(r, p, base);
    return (x < 0.0) ? -a : a;
}

LPM_FN double lpm_atan2_d(double y, double x)
{
    if (x != x || y != y) return lpm_nan();
    if (x > 0.0) return lpm_atan_d(y / x);
    if (x < 0.0) {
        double a = lpm_atan_d(y / x);
        return (y >= 0.0) ? a + LPM_PI : a - LPM_PI;
    }
    if (y > 0.0) return LPM_PIO2;
    if (y < 0.0) return -LPM_PIO2;
    return 0.0;
}

LPM_FN double lpm_acos_d(double x)
{
    if (!(fabs(x) <= 1.0)) return lpm_nan();
    return lpm_atan2_d(sqrt((1.0 - x) * (1.0 + x)), x);
}

/* exp(x), relative error < 1e-12 */
LPM_FN double lpm_exp_d(double x)
{
    if (x != x) return x;
    if (x > 709.0) return lpm_inf();
    if (x < -745.0) return 0.0;
    const double LOG2E  = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kd = rint(x * LOG2E);
    double r = LPM_FMA(-kd, LN2_LO, LPM_FMA(-kd, LN2_HI, x));
    int k = (int)kd;
    /* |r| <= ln2 / 2: Taylor through r^10 (next term < 3e-13) */
    double p = 1.0 / 3628800.0;             /* 1/10! */
    p = LPM_FMA(p, r, 1.0 / 362880.0);
    p = LPM_FMA(p, r, 1.0 / 40320.0);
    p = LPM_FMA(p, r, 1.0 / 5040.0);
    p = LPM_FMA(p, r, 1.0 / 720.0);
    p = LPM_FMA(p, r, 1.0 / 120.0);
    p = LPM_FMA(p, r, 1.0 / 24.0);
    p = LPM_FMA(p, r, 1.0 / 6.0);
    p = LPM_FMA(p, r, 0.5);
    p = LPM_FMA(p, r, 1.0);
    p = LPM_FMA(p, r, 1.0);
    int k1 = k / 2;
    int k2 = k - k1;
    return (p * lpm_pow2i(k1)) * lpm_pow2i(k2);
}

/* natural log, |err| < 1e-14 relative */
LPM_FN double lpm_log_d(double x)
{
    if (x != x) return x;
    if (x < 0.0) return lpm_nan();
    if (x == 0.0) return -lpm_inf();
    uint64_t u = lpm_d2u(x);
    if (u == 0x7FF0000000000000ull) return x;
    int e = 0;
    if (((u >> 52) & 0x7FF) == 0) {   /* subnormal double: rescale */
        x = x * 18014398509481984.0;  /* 2^54 */
        u = lpm_d2u(x);
        e = -54;
    }
    e += (int)((u >> 52) & 0x7FF) - 1023;
    double m = lpm_u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);  /* [1,2) */
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    /* |s| <= 0.1716: 2 s sum_{n=0}^{8} z^n / (2n+1) (next term < 2e-16 relative) */
    double p = 1.0 / 17.0;
    p = LPM_FMA(p, z, 1.0 / 15.0);
    p = LPM_FMA(p, z, 1.0 / 13.0);
    p = LPM_FMA(p, z, 1.0 / 11.0);
    p = LPM_FMA(p, z, 1.0 / 9.0);
    p = LPM_FMA(p, z, 1.0 / 7.0);
    p = LPM_FMA(p, z, 1.0 / 5.0);
    p = LPM_FMA(p, z, 1.0 / 3.0);
    p = LPM_FMA(p, z, 1.0);
    double logm = 2.0 * s * p;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double ed = (double)e;
    return LPM_FMA(ed, LN2_HI, LPM_FMA(ed, LN2_LO, logm));
}

/* ---- f32 entry points: one rounding at the end ---- */

LPM_FN float lpm_sinf(float x) { double s, c; lpm_sincos_d((double)x, &s, &c); return (float)s; }
LPM_FN float lpm_cosf(float x) { double s, c; lpm_sincos_d((double)x, &s, &c); return (float)c; }
LPM_FN void  lpm_sincosf(float x, float *s_out, float *c_out)
{
    double s, c; lpm_sincos_d((double)x, &s, &c); *s_out = (float)s; *c_out = (float)c;
}
LPM_FN float lpm_atanf(float x) { return (float)lpm_atan_d((double)x); }
LPM_FN float lpm_atan2f(float y, float x) { return (float)lpm_atan2_d((double)y, (double)x); }
LPM_FN float lpm_acosf(float x) { return (float)lpm_acos_d((double)x); }
LPM_FN float lpm_expf(float x) { return (float)lpm_exp_d((double)x); }
LPM_FN float lpm_logf(float x) { return (float)lpm_log_d((double)x); }
/* WGSL pow(x, y) = exp2(y * log2(x)): NaN for x < 0 */
LPM_FN float lpm_powf(float x, float y)
{
    return (float)lpm_exp_d((double)y * lpm_log_d((double)x));
}

/* normalize(v) (WGSL leaves its precision to the implementation, "inherited from v / length(v)"): defined here, for the oracle
 * and the kernels alike, as  v * (1 / sqrt(v.v))  -- the squared length summed left to right, one correctly rounded square
 * root, ONE correctly rounded division, three multiplications.  Rounds 1 and 2 divided each component by the length: three
 * correctly rounded f32 divisions cost ~30 vector instructions on gfx950, a quarter of the shading kernel's arithmetic with
 * the square roots; the reciprocal form drops two of them.  Zero-length input gives NaN / inf components
 * either way. */
LPM_FN void lpm_normalize3f(float x, float y, float z, float *ox, float *oy, float *oz)
{
    const float l = sqrtf(x * x + y * y + z * z);
    const float inv = 1.0f / l;
    *ox = x * inv; *oy = y * inv; *oz = z * inv;
}

#endif /* LUPIN_DETMATH_H */
