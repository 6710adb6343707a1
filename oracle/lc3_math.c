/* ORACLE (test infrastructure, not product code).  See lc3_math.h for provenance.
 *
 * Restatement of the msun-derived single-precision routines that the Rust
 * `libm` crate (pulled in by num-traits, reference Cargo.toml:17) executes at
 * run time.  All arithmetic is plain IEEE-754 binary32 / binary64, evaluated as
 * written (compile with -ffp-contract=off).
 */
#include "lc3_math.h"
#include <math.h>
#include <string.h>

#define F(u) lc3m_from_bits(u)
#define B(f) lc3m_to_bits(f)

/* ------------------------------------------------------------------ powf (e_powf.c) */
float lc3m_powf(float x, float y) {
    static const float bp[2] = {1.0f, 1.5f};
    const float dp_h[2] = {0.0f, F(0x3f15c000u)}; /* 5.84960938e-01 */
    const float dp_l[2] = {0.0f, F(0x35d1cfdcu)}; /* 1.56322085e-06 */
    const float two24 = 16777216.0f;
    const float huge = 1.0e30f, tiny = 1.0e-30f;
    const float L1 = F(0x3f19999au), L2 = F(0x3edb6db7u), L3 = F(0x3eaaaaabu);
    const float L4 = F(0x3e8ba305u), L5 = F(0x3e6c3255u), L6 = F(0x3e53f142u);
    const float P1 = F(0x3e2aaaabu), P2 = F(0xbb360b61u), P3 = F(0x388ab355u);
    const float P4 = F(0xb5ddea0eu), P5 = F(0x3331bb4cu);
    const float lg2 = F(0x3f317218u), lg2_h = F(0x3f317200u), lg2_l = F(0x35bfbe8cu);
    const float ovt = 4.2995665694e-08f;
    const float cp = F(0x3f76384fu), cp_h = F(0x3f764000u), cp_l = F(0xb8f623c6u);
    const float ivln2 = F(0x3fb8aa3bu), ivln2_h = F(0x3fb8aa00u), ivln2_l = F(0x36eca570u);

    float z, ax, z_h, z_l, p_h, p_l;
    float y1, t1, t2, r, s, sn, t, u, v, w;
    int32_t i, j, k, yisint, n;
    int32_t hx, hy, ix, iy, is;

    hx = (int32_t)B(x);
    hy = (int32_t)B(y);
    ix = hx & 0x7fffffff;
    iy = hy & 0x7fffffff;

    if (iy == 0) return 1.0f;
    if (hx == 0x3f800000) return 1.0f;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;

    yisint = 0;
    if (hx < 0) {
        if (iy >= 0x4b800000) yisint = 2;
        else if (iy >= 0x3f800000) {
            k = (iy >> 23) - 0x7f;
            j = iy >> (23 - k);
            if ((j << (23 - k)) == iy) yisint = 2 - (j & 1);
        }
    }

    if (iy == 0x7f800000) {
        if (ix == 0x3f800000) return 1.0f;
        else if (ix > 0x3f800000) return hy >= 0 ? y : 0.0f;
        else return hy >= 0 ? 0.0f : -y;
    }
    if (iy == 0x3f800000) return hy >= 0 ? x : 1.0f / x;
    if (hy == 0x40000000) return x * x;
    if (hy == 0x3f000000) {
        if (hx >= 0) return sqrtf(x);
    }

    ax = fabsf(x);
    if (ix == 0x7f800000 || ix == 0 || ix == 0x3f800000) {
        z = ax;
        if (hy < 0) z = 1.0f / z;
        if (hx < 0) {
            if (((ix - 0x3f800000) | yisint) == 0) z = (z - z) / (z - z);
            else if (yisint == 1) z = -z;
        }
        return z;
    }

    sn = 1.0f;
    if (hx < 0) {
        if (yisint == 0) return (x - x) / (x - x);
        if (yisint == 1) sn = -1.0f;
    }

    if (iy > 0x4d000000) {
        if (ix < 0x3f7ffff8) return hy < 0 ? sn * huge * huge : sn * tiny * tiny;
        if (ix > 0x3f800007) return hy > 0 ? sn * huge * huge : sn * tiny * tiny;
        t = ax - 1.0f;
        w = (t * t) * (0.5f - t * (0.333333333333f - t * 0.25f));
        u = ivln2_h * t;
        v = t * ivln2_l - w * ivln2;
        t1 = u + v;
        is = (int32_t)B(t1);
        t1 = F((uint32_t)is & 0xfffff000u);
        t2 = v - (t1 - u);
    } else {
        float s2, s_h, s_l, t_h, t_l;
        n = 0;
        if (ix < 0x00800000) {
            ax *= two24;
            n -= 24;
            ix = (int32_t)B(ax);
        }
        n += ((ix) >> 23) - 0x7f;
        j = ix & 0x007fffff;
        ix = j | 0x3f800000;
        if (j <= 0x1cc471) k = 0;
        else if (j < 0x5db3d7) k = 1;
        else {
            k = 0;
            n += 1;
            ix -= 0x00800000;
        }
        ax = F((uint32_t)ix);

        u = ax - bp[k];
        v = 1.0f / (ax + bp[k]);
        s = u * v;
        s_h = s;
        is = (int32_t)B(s_h);
        s_h = F((uint32_t)is & 0xfffff000u);
        is = (int32_t)((((uint32_t)ix >> 1) & 0xfffff000u) | 0x20000000u);
        t_h = F((uint32_t)is + 0x00400000u + ((uint32_t)k << 21));
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        s2 = s * s;
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + s);
        s2 = s_h * s_h;
        t_h = 3.0f + s2 + r;
        is = (int32_t)B(t_h);
        t_h = F((uint32_t)is & 0xfffff000u);
        t_l = r - ((t_h - 3.0f) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * s;
        p_h = u + v;
        is = (int32_t)B(p_h);
        p_h = F((uint32_t)is & 0xfffff000u);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h;
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        t = (float)n;
        t1 = (((z_h + z_l) + dp_h[k]) + t);
        is = (int32_t)B(t1);
        t1 = F((uint32_t)is & 0xfffff000u);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }

    is = (int32_t)B(y);
    y1 = F((uint32_t)is & 0xfffff000u);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    j = (int32_t)B(z);
    if (j > 0x43000000) return sn * huge * huge;
    else if (j == 0x43000000) {
        if (p_l + ovt > z - p_h) return sn * huge * huge;
    } else if ((j & 0x7fffffff) > 0x43160000) return sn * tiny * tiny;
    else if ((uint32_t)j == 0xc3160000u) {
        if (p_l <= z - p_h) return sn * tiny * tiny;
    }

    i = j & 0x7fffffff;
    k = (i >> 23) - 0x7f;
    n = 0;
    if (i > 0x3f000000) {
        n = j + (0x00800000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 23) - 0x7f;
        t = F((uint32_t)n & ~(0x007fffffu >> k));
        n = ((n & 0x007fffff) | 0x00800000) >> (23 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = p_l + p_h;
    is = (int32_t)B(t);
    t = F((uint32_t)is & 0xffff8000u);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / (t1 - 2.0f) - (w + z * w);
    z = 1.0f - (r - z);
    j = (int32_t)B(z);
    j += (int32_t)((uint32_t)n << 23);
    if ((j >> 23) <= 0) z = scalbnf(z, n);
    else z = F((uint32_t)j);
    return sn * z;
}

/* ------------------------------------------------------------------ log2f / log10f (e_log2f.c, e_log10f.c) */
static const uint32_t LG1 = 0x3f2aaaaau; /* 0xaaaaaa.0p-24 */
static const uint32_t LG2 = 0x3eccce13u; /* 0xccce13.0p-25 */
static const uint32_t LG3 = 0x3e91e9eeu; /* 0x91e9ee.0p-25 */
static const uint32_t LG4 = 0x3e789e26u; /* 0xf89e26.0p-26 */

float lc3m_log2f(float x) {
    const float ivln2hi = F(0x3fb8b000u); /*  1.4428710938e+00 */
    const float ivln2lo = F(0xb9389ad4u); /* -1.7605285393e-04 */
    const float Lg1 = F(LG1), Lg2 = F(LG2), Lg3 = F(LG3), Lg4 = F(LG4);
    float hfsq, f, s, z, R, w, t1, t2, hi, lo;
    uint32_t ix = B(x);
    int k = 0;

    if (ix < 0x00800000u || (ix >> 31)) {
        if ((ix << 1) == 0) return -1.0f / (x * x);
        if (ix >> 31) return (x - x) / 0.0f;
        k -= 25;
        x *= 33554432.0f; /* 0x1p25f */
        ix = B(x);
    } else if (ix >= 0x7f800000u) {
        return x;
    } else if (ix == 0x3f800000u) {
        return 0.0f;
    }

    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = F(ix);

    f = x - 1.0f;
    s = f / (2.0f + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * Lg4);
    t2 = z * (Lg1 + w * Lg3);
    R = t2 + t1;
    hfsq = 0.5f * f * f;

    hi = f - hfsq;
    hi = F(B(hi) & 0xfffff000u);
    lo = (f - hi) - hfsq + s * (hfsq + R);
    return (lo + hi) * ivln2lo + lo * ivln2hi + hi * ivln2hi + (float)k;
}

float lc3m_log10f(float x) {
    const float ivln10hi = F(0x3ede6000u);  /*  4.3432617188e-01 */
    const float ivln10lo = F(0xb804ead9u);  /* -3.1689971365e-05 */
    const float log10_2hi = F(0x3e9a2080u); /*  3.0102920532e-01 */
    const float log10_2lo = F(0x355427dbu); /*  7.9034151668e-07 */
    const float Lg1 = F(LG1), Lg2 = F(LG2), Lg3 = F(LG3), Lg4 = F(LG4);
    float hfsq, f, s, z, R, w, t1, t2, dk, hi, lo;
    uint32_t ix = B(x);
    int k = 0;

    if (ix < 0x00800000u || (ix >> 31)) {
        if ((ix << 1) == 0) return -1.0f / (x * x);
        if (ix >> 31) return (x - x) / 0.0f;
        k -= 25;
        x *= 33554432.0f;
        ix = B(x);
    } else if (ix >= 0x7f800000u) {
        return x;
    } else if (ix == 0x3f800000u) {
        return 0.0f;
    }

    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = F(ix);

    f = x - 1.0f;
    s = f / (2.0f + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * Lg4);
    t2 = z * (Lg1 + w * Lg3);
    R = t2 + t1;
    hfsq = 0.5f * f * f;

    hi = f - hfsq;
    hi = F(B(hi) & 0xfffff000u);
    lo = f - hi - hfsq + s * (hfsq + R);
    dk = (float)k;
    return dk * log10_2lo + (lo + hi) * ivln10lo + lo * ivln10hi + hi * ivln10hi + dk * log10_2hi;
}

/* ------------------------------------------------------------------ exp2f (s_exp2f.c, TBLSIZE = 16) */
float lc3m_exp2f(float x) {
    /* exp2ft[i] = 2^((i-8)/16), correctly rounded doubles */
    static const uint64_t exp2ft_bits[16] = {
        0x3fe6a09e667f3bcdull, 0x3fe7a11473eb0187ull, 0x3fe8ace5422aa0dbull, 0x3fe9c49182a3f090ull,
        0x3feae89f995ad3adull, 0x3fec199bdd85529cull, 0x3fed5818dcfba487ull, 0x3feea4afa2a490daull,
        0x3ff0000000000000ull, 0x3ff0b5586cf9890full, 0x3ff172b83c7d517bull, 0x3ff2387a6e756238ull,
        0x3ff306fe0a31b715ull, 0x3ff3dea64c123422ull, 0x3ff4bfdad5362a27ull, 0x3ff5ab07dd485429ull,
    };
    const float redux = F(0x4b400000u) / 16.0f; /* 0x1.8p23f / TBLSIZE */
    const float P1 = F(0x3f317218u), P2 = F(0x3e75fdf0u), P3 = F(0x3d6359a4u), P4 = F(0x3c1d964eu);
    double t, r, z, tbl, scale;
    uint32_t ui = B(x);
    uint32_t ix = ui & 0x7fffffffu, i0, k;
    uint64_t uk;
    float uf;

    if (ix > 0x42fc0000u) { /* |x| > 126 */
        if (ix > 0x7f800000u) return x;
        if (ui >= 0x43000000u && ui < 0x80000000u) { /* x >= 128 */
            x *= F(0x7f000000u); /* 0x1p127f */
            return x;
        }
        if (ui >= 0x80000000u) { /* x < -126 */
            if (ui >= 0xc3160000u) return 0.0f; /* x <= -150 */
        }
    } else if (ix <= 0x33000000u) { /* |x| <= 0x1p-25 */
        return 1.0f + x;
    }

    uf = x + redux;
    i0 = B(uf);
    i0 += 16 / 2;
    k = i0 / 16;
    uk = (uint64_t)(0x3ffu + k) << 52;
    i0 &= 16 - 1;
    uf -= redux;
    z = (double)(x - uf);
    memcpy(&tbl, &exp2ft_bits[i0], 8);
    r = tbl;
    t = r * z;
    r = r + t * ((double)P1 + z * (double)P2) + t * (z * z) * ((double)P3 + z * (double)P4);
    memcpy(&scale, &uk, 8);
    return (float)(r * scale);
}

/* ------------------------------------------------------------------ asinf (e_asinf.c) */
static float asinf_R(float z) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f;
    const float qS1 = -7.0662963390e-01f;
    float p, q;
    p = z * (pS0 + z * (pS1 + z * pS2));
    q = 1.0f + z * qS1;
    return p / q;
}

float lc3m_asinf(float x) {
    const double pio2 = 1.570796326794896558e+00;
    double s;
    float z;
    uint32_t hx = B(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3f800000u) {
        if (ix == 0x3f800000u) return (float)((double)x * pio2 + 7.5231638452626401e-37 /* 0x1p-120f */);
        return 0.0f / (x - x);
    }
    if (ix < 0x3f000000u) {
        if (ix < 0x39800000u && ix >= 0x00800000u) return x;
        return x + x * asinf_R(x * x);
    }
    z = (1.0f - fabsf(x)) * 0.5f;
    s = sqrt((double)z);
    x = (float)(pio2 - 2.0 * (s + s * (double)asinf_R(z)));
    if (hx >> 31) return -x;
    return x;
}

/* ------------------------------------------------------------------ sinf (s_sinf.c + k_sinf.c + k_cosf.c) */
static float k_sindf(double x) {
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59;
    const double S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    double r, s, w, z;
    z = x * x;
    w = z * z;
    r = S3 + z * S4;
    s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}
static float k_cosdf(double x) {
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57;
    const double C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    double r, w, z;
    z = x * x;
    w = z * z;
    r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}

float lc3m_sinf(float x) {
    const double s1pio2 = 1.5707963267948966, s2pio2 = 3.1415926535897931;
    uint32_t ix = B(x);
    int sign = (int)(ix >> 31);
    ix &= 0x7fffffffu;
    if (ix <= 0x3f490fdau) { /* |x| ~<= pi/4 */
        if (ix < 0x39800000u) return x;
        return k_sindf((double)x);
    }
    if (ix <= 0x407b53d1u) { /* |x| ~<= 5*pi/4 */
        if (ix <= 0x4016cbe3u) { /* |x| ~<= 3pi/4 */
            if (sign) return -k_cosdf((double)x + s1pio2);
            else return k_cosdf((double)x - s1pio2);
        }
        return k_sindf(sign ? -((double)x + s2pio2) : -((double)x - s2pio2));
    }
    /* the codec only evaluates sin on |x| <= 8*pi/17; larger arguments are out of contract here */
    return (float)sin((double)x);
}

/* ------------------------------------------------------------------ fast_math::exp2_raw (fast-math 0.1.1) */
float lc3m_exp2_raw(float x) {
    const float A = 8388608.0f; /* 2^23 */
    const float E = 1.1920929e-7f;
    const float C0 = (0.3371894346f * E) * E;
    const float C1 = 0.657636276f * E;
    const float C2 = 1.00172476f;
    float a = A * x;
    int32_t mul = lc3m_f32_to_i32(a);
    int32_t fl = (int32_t)((uint32_t)mul & 0xff800000u);
    float frac = (float)(int32_t)((uint32_t)mul - (uint32_t)fl);
    float approx = (C0 * frac + C1) * frac + C2;
    return F(B(approx) + (uint32_t)fl);
}

float lc3m_powi(float base, int exp) {
    /* num_traits: negative exponent -> recip() first, then pow(base, exp as usize) */
    unsigned e;
    float acc;
    if (exp < 0) {
        base = 1.0f / base;
        e = (unsigned)(-(long)exp);
    } else {
        e = (unsigned)exp;
    }
    if (e == 0) return 1.0f;
    while ((e & 1u) == 0) {
        base = base * base;
        e >>= 1;
    }
    if (e == 1) return base;
    acc = base;
    while (e > 1) {
        e >>= 1;
        base = base * base;
        if (e & 1u) acc = acc * base;
    }
    return acc;
}

/* ------------------------------------------------------------------ Rust `as` casts */
int32_t lc3m_f32_to_i32(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
int16_t lc3m_f32_to_i16(float x) {
    if (x != x) return 0;
    if (x >= 32767.0f) return 32767;
    if (x <= -32768.0f) return -32768;
    return (int16_t)x;
}
int8_t lc3m_f32_to_i8(float x) {
    if (x != x) return 0;
    if (x >= 127.0f) return 127;
    if (x <= -128.0f) return -128;
    return (int8_t)x;
}
uint16_t lc3m_f32_to_u16(float x) {
    if (x != x) return 0;
    if (x >= 65535.0f) return 65535;
    if (x <= 0.0f) return 0;
    return (uint16_t)x;
}
uint64_t lc3m_f64_to_usize(double x) {
    if (x != x) return 0;
    if (x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}
