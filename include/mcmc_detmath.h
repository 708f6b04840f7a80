/*
 * mcmc_detmath.h -- deterministic double-precision cos(), exp() and log().
 *
 * Why this exists: the reference calls libm's std::cos (covid.h:619, the behaviour
 * proposal), std::exp (mcmc.h:776, the acceptance test; test.h:93-96) and std::log
 * (test.h:98).  glibc's and the GPU's
 * (ocml) implementations differ in the last bit for some arguments, which would make
 * the chain state on the GPU drift from a CPU run by an ulp and defeat bit-exact
 * comparison.  Both the HIP kernels and the CPU oracle therefore use the two functions
 * below: a Cody-Waite reduction plus minimax polynomials in the style of the classic
 * Sun fdlibm algorithms (restated from the published method; error < 1 ulp, checked
 * against libm in tests/test_detmath.py).  Every operation is a single IEEE-754
 * double operation in a fixed order, so the result is identical on any conforming
 * target as long as the translation unit is compiled with floating-point contraction
 * OFF (-ffp-contract=off): no FMA may be formed here.
 *
 * Usable from C99, C++ and HIP device code.
 */
#ifndef MCMC_DETMATH_H
#define MCMC_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MCMC_HD __host__ __device__ static inline
#else
#define MCMC_HD static inline
#endif

MCMC_HD double mcmc_bits2d(uint64_t u) {
    union { uint64_t u; double d; } x; x.u = u; return x.d;
}
MCMC_HD uint64_t mcmc_d2bits(double d) {
    union { uint64_t u; double d; } x; x.d = d; return x.u;
}

/* round-to-nearest-even of a double with |x| < 2^51 without calling libm */
MCMC_HD double mcmc_rint_small(double x) {
    const double big = 6755399441055744.0; /* 1.5 * 2^52 */
    double t = x + big;
    return t - big;
}

/* cos on [-pi/4, pi/4] of x + y (y = tail of the reduced argument) */
MCMC_HD double mcmc_kcos(double x, double y) {
    const double C1 =  4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 =  2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 =  2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* sin on [-pi/4, pi/4] of x + y */
MCMC_HD double mcmc_ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 =  8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 =  2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 =  1.58969099521155010221e-10;
    double z = x * x;
    double w = z * z;
    double r = S2 + z * (S3 + z * S4) + (z * w) * (S5 + z * S6);
    double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/*
 * Deterministic cos(x).  Accurate (< 1 ulp) for |x| up to ~1e5, which covers the only
 * call site (omega*i <= 2*3.1415*5*nB/T < 32, covid.h:604,619).  Larger arguments still
 * give a deterministic result but lose accuracy (three-term Cody-Waite only).
 * NaN/Inf give NaN.
 */
MCMC_HD double mcmc_cos(double x) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double p1  = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double p2  = 6.07710050630396597660e-11;  /* next 33 bits */
    const double p2t = 2.02226624879595063154e-21;  /* pi/2 - p1 - p2 */
    /* Straight-line on purpose: on the GPU 64 chains evaluate this in lockstep with arguments in
     * every quadrant, so each data-dependent branch is executed by the whole wave anyway, plus the
     * exec-mask bookkeeping around it.  Both kernels are evaluated and the quadrant selects.
     * (|x| <= pi/4 needs no case of its own: there fn = 0, the reduction returns (ax, 0) exactly
     * and quadrant 0 evaluates kcos(ax, 0) = kcos(x, 0).) */
    const double ax = x < 0 ? -x : x;
    const double fn = mcmc_rint_small(ax * invpio2);
    /* two-stage Cody-Waite reduction, always both stages: fixed operation count.
     * pi/2 = p1 + p2 + p2t to ~118 bits; fn*p1 and fn*p2 are exact for fn < 2^20. */
    const double t = ax - fn * p1;
    double w = fn * p2;
    const double r = t - w;
    w = fn * p2t - ((t - r) - w);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const int64_t n = (ax < 1.0e300) ? (int64_t)fn : 0;
    const double c = mcmc_kcos(y0, y1), s = mcmc_ksin(y0, y1);
    double v = (n & 1) ? s : c;                      /* quadrants 0..3: c, -s, -c, s */
    v = ((n + 1) & 2) ? -v : v;
    return (ax < 1.0e300) ? v : x - x;               /* inf or nan -> nan (huge finite: garbage-in, 0) */
}

/*
 * Deterministic exp(x).  Overflow -> +inf, underflow -> 0 (through subnormals),
 * NaN -> NaN.
 */
MCMC_HD double mcmc_exp(double x) {
    const double o_thr =  7.09782712893383973096e+02;
    const double u_thr = -7.45133219101941108420e+02;
    const double ln2hi =  6.93147180369123816490e-01;
    const double ln2lo =  1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 =  1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 =  6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 =  4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > o_thr) return mcmc_bits2d(0x7ff0000000000000ULL);
    if (x < u_thr) return 0.0;
    double ax = x < 0 ? -x : x;
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (ax > 0.34657359027997264) {            /* |x| > 0.5 ln2 */
        double kf = mcmc_rint_small(x * invln2);
        k = (int)kf;
        hi = x - kf * ln2hi;
        lo = kf * ln2lo;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {   /* |x| < 2^-28 */
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y;
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    /* scale by 2^k in two exact steps so subnormal results round once */
    if (k >= -1021) {
        if (k == 1024) return (y * 2.0) * mcmc_bits2d(0x7fe0000000000000ULL);
        return y * mcmc_bits2d((uint64_t)(0x3ff + k) << 52);
    }
    return (y * mcmc_bits2d((uint64_t)(0x3ff + (k + 1000)) << 52)) *
           mcmc_bits2d((uint64_t)(0x3ff - 1000) << 52);
}

/*
 * Deterministic log(x) (FourGaussians::eval, test.h:98).  log(0) = -inf, log(<0) = NaN,
 * subnormals handled.  Argument reduction x = 2^k (1+f), log(1+f) by the classic minimax
 * polynomial in s = f/(2+f).
 */
MCMC_HD double mcmc_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits = mcmc_d2bits(x);
    int32_t hx = (int32_t)(bits >> 32);
    uint32_t lx = (uint32_t)bits;
    int32_t k = 0;
    if (hx < 0x00100000) {                              /* x < 2^-1022 */
        if (((hx & 0x7fffffff) | lx) == 0) return -mcmc_bits2d(0x7ff0000000000000ULL);
        if (hx < 0) return (x - x) / (x - x);           /* NaN */
        k -= 54;
        x *= two54;
        bits = mcmc_d2bits(x);
        hx = (int32_t)(bits >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    bits = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (bits & 0xffffffffULL);
    x = mcmc_bits2d(bits);                              /* normalise x or x/2 */
    k += (i >> 20);
    const double f = x - 1.0;
    const double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {                  /* |f| < 2^-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    i = hx - 0x6147a;
    const double w = z * z;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

#endif /* MCMC_DETMATH_H */
