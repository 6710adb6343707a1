/* ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See lc3_oracle.h.
 * Restates reference src/common/{config,kissfft,dct_iv}.rs. */
#include "lc3_oracle.h"
#include <math.h>
#include <string.h>

/* common/config.rs:42-100 */
int lc3o_config_new(lc3o_config *c, int fs_hz, int frame_us) {
    c->spec_flags = 0;
    static const int fs_tab[6] = {8000, 16000, 24000, 32000, 44100, 48000};
    static const int ind_tab[6] = {0, 1, 2, 3, 4, 4};
    static const int nf75[6] = {60, 120, 180, 240, 360, 360};
    static const int nf10[6] = {80, 160, 240, 320, 480, 480};
    int i, k = -1;
    for (i = 0; i < 6; i++)
        if (fs_tab[i] == fs_hz) k = i;
    if (k < 0) return -1;
    if (frame_us != 7500 && frame_us != 10000) return -1;
    c->fs = fs_hz;
    c->fs_ind = ind_tab[k];
    c->n_ms_10 = frame_us == 10000;
    if (!c->n_ms_10) {
        c->nf = nf75[k];
        c->ne = c->nf == 360 ? 300 : c->nf;
        c->nb = fs_hz == 8000 ? 60 : 64;
        c->z = 7 * c->nf / 30;
    } else {
        c->nf = nf10[k];
        c->ne = c->nf == 480 ? 400 : c->nf;
        c->nb = 64;
        c->z = 3 * c->nf / 8;
    }
    return 0;
}

/* common/complex.rs:16-24 */
static inline lc3o_cpx cmul(lc3o_cpx a, lc3o_cpx b) {
    lc3o_cpx r;
    r.r = a.r * b.r - a.i * b.i;
    r.i = a.r * b.i + a.i * b.r;
    return r;
}
static inline lc3o_cpx cadd(lc3o_cpx a, lc3o_cpx b) { lc3o_cpx r = {a.r + b.r, a.i + b.i}; return r; }
static inline lc3o_cpx csub(lc3o_cpx a, lc3o_cpx b) { lc3o_cpx r = {a.r - b.r, a.i - b.i}; return r; }

/* common/kissfft.rs:47-76 */
static void kf_factor(int n, int *factors) {
    int p = 4, i = 0;
    float floor_sqrt = floorf(sqrtf((float)n));
    for (;;) {
        while ((n % p) != 0) {
            if (p == 4) p = 2;
            else if (p == 2) p = 3;
            else p += 2;
            if ((float)p > floor_sqrt) p = n;
        }
        n /= p;
        factors[i++] = p;
        factors[i++] = n;
        if (n <= 1) break;
    }
}

/* common/kissfft.rs:17-41 : twiddles in f64, cast to f32 */
void lc3o_fft_init(lc3o_fft *f, int nfft) {
    int i;
    const double PI = 3.14159265358979323846264338327950288;
    memset(f, 0, sizeof(*f));
    f->nfft = nfft;
    for (i = 0; i < nfft; i++) {
        double phase = -2.0 * PI * (double)i / (double)nfft;
        f->tw[i].r = (float)cos(phase);
        f->tw[i].i = (float)sin(phase);
    }
    kf_factor(nfft, f->factors);
}

/* kissfft.rs:133-141 */
static void kf_bfly2(const lc3o_fft *f, lc3o_cpx *fout, int fstride, int m) {
    int i;
    for (i = 0; i < m; i++) {
        lc3o_cpx t = cmul(fout[m + i], f->tw[i * fstride]);
        fout[m + i] = csub(fout[i], t);
        fout[i] = cadd(fout[i], t);
    }
}

/* kissfft.rs:143-175 (forward transform only: inverse == false) */
static void kf_bfly4(const lc3o_fft *f, lc3o_cpx *fout, int fstride, int m) {
    int m2 = 2 * m, m3 = 3 * m, i, tw1 = 0, tw2 = 0, tw3 = 0;
    for (i = 0; i < m; i++) {
        lc3o_cpx s0 = cmul(fout[i + m], f->tw[tw1]);
        lc3o_cpx s1 = cmul(fout[i + m2], f->tw[tw2]);
        lc3o_cpx s2 = cmul(fout[i + m3], f->tw[tw3]);
        lc3o_cpx s5 = csub(fout[i], s1);
        lc3o_cpx s3, s4;
        fout[i] = cadd(fout[i], s1);
        s3 = cadd(s0, s2);
        s4 = csub(s0, s2);
        fout[i + m2] = csub(fout[i], s3);
        fout[i] = cadd(fout[i], s3);
        tw1 += fstride;
        tw2 += fstride * 2;
        tw3 += fstride * 3;
        fout[i + m].r = s5.r + s4.i;
        fout[i + m].i = s5.i - s4.r;
        fout[i + m3].r = s5.r - s4.i;
        fout[i + m3].i = s5.i + s4.r;
    }
}

/* kissfft.rs:177-205 */
static void kf_bfly3(const lc3o_fft *f, lc3o_cpx *fout, int fstride, int m) {
    int m2 = 2 * m, i, tw1 = 0, tw2 = 0;
    lc3o_cpx epi3 = f->tw[fstride * m];
    for (i = 0; i < m; i++) {
        lc3o_cpx s1 = cmul(fout[i + m], f->tw[tw1]);
        lc3o_cpx s2 = cmul(fout[i + m2], f->tw[tw2]);
        lc3o_cpx s3 = cadd(s1, s2);
        lc3o_cpx s0 = csub(s1, s2);
        lc3o_cpx fouti, foutm;
        tw1 += fstride;
        tw2 += fstride * 2;
        fouti = fout[i];
        fout[i + m].r = fouti.r - (s3.r * 0.5f);
        fout[i + m].i = fouti.i - (s3.i * 0.5f);
        s0.r *= epi3.i;
        s0.i *= epi3.i;
        fout[i] = cadd(fout[i], s3);
        foutm = fout[i + m];
        fout[i + m2].r = foutm.r + s0.i;
        fout[i + m2].i = foutm.i - s0.r;
        fout[i + m].r = foutm.r - s0.i;
        fout[i + m].i = foutm.i + s0.r;
    }
}

/* kissfft.rs:207-256 */
static void kf_bfly5(const lc3o_fft *f, lc3o_cpx *fout, int fstride, int m) {
    lc3o_cpx ya = f->tw[fstride * m], yb = f->tw[fstride * 2 * m];
    int m1 = m, m2 = 2 * m, m3 = 3 * m, m4 = 4 * m, i;
    for (i = 0; i < m; i++) {
        lc3o_cpx s0 = fout[i];
        lc3o_cpx s1 = cmul(fout[i + m1], f->tw[i * fstride]);
        lc3o_cpx s2 = cmul(fout[i + m2], f->tw[i * 2 * fstride]);
        lc3o_cpx s3 = cmul(fout[i + m3], f->tw[i * 3 * fstride]);
        lc3o_cpx s4 = cmul(fout[i + m4], f->tw[i * 4 * fstride]);
        lc3o_cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
        lc3o_cpx s5, s6, s11, s12;
        fout[i].r += s7.r + s8.r;
        fout[i].i += s7.i + s8.i;
        s5.r = s0.r + (s7.r * ya.r) + (s8.r * yb.r);
        s5.i = s0.i + (s7.i * ya.r) + (s8.i * yb.r);
        s6.r = (s10.i * ya.i) + (s9.i * yb.i);
        s6.i = -(s10.r * ya.i) - (s9.r * yb.i);
        fout[i + m1] = csub(s5, s6);
        fout[i + m4] = cadd(s5, s6);
        s11.r = s0.r + (s7.r * yb.r) + (s8.r * ya.r);
        s11.i = s0.i + (s7.i * yb.r) + (s8.i * ya.r);
        s12.r = -(s10.i * yb.i) + (s9.i * ya.i);
        s12.i = (s10.r * yb.i) - (s9.r * ya.i);
        fout[i + m2] = cadd(s11, s12);
        fout[i + m3] = csub(s11, s12);
    }
}

/* kissfft.rs:86-131 (in_stride == 1) */
static void kf_work(const lc3o_fft *f, lc3o_cpx *fout, const lc3o_cpx *fin, int fstride, int factor_idx,
                    int fin_idx, int fout_idx) {
    int p = f->factors[factor_idx], m = f->factors[factor_idx + 1];
    int begin = fout_idx, end = fout_idx + p * m;
    factor_idx += 2;
    if (m == 1) {
        int k, j = fin_idx;
        for (k = fout_idx; k < end; k++, j += fstride) fout[k] = fin[j];
    } else {
        do {
            kf_work(f, fout, fin, fstride * p, factor_idx, fin_idx, fout_idx);
            fin_idx += fstride;
            fout_idx += m;
        } while (fout_idx != end);
    }
    switch (p) {
    case 2: kf_bfly2(f, fout + begin, fstride, m); break;
    case 3: kf_bfly3(f, fout + begin, fstride, m); break;
    case 4: kf_bfly4(f, fout + begin, fstride, m); break;
    case 5: kf_bfly5(f, fout + begin, fstride, m); break;
    default: break; /* generic radix never reached for LC3 sizes */
    }
}

void lc3o_fft_run(const lc3o_fft *f, const lc3o_cpx *fin, lc3o_cpx *fout) { kf_work(f, fout, fin, 1, 0, 0, 0); }

/* common/dct_iv.rs:22-47 */
void lc3o_dct4_init(lc3o_dct4 *d, int nf) {
    int i, count = nf / 2;
    const double PI = 3.14159265358979323846264338327950288;
    d->nf = nf;
    lc3o_fft_init(&d->fft, count);
    for (i = 0; i < count; i++) {
        double temp = -PI * (double)(8 * i + 1) / (8.0 * ((double)count) * 2.0);
        d->tw[i].r = (float)cos(temp);
        d->tw[i].i = (float)sin(temp);
    }
}

/* common/dct_iv.rs:49-67 */
void lc3o_dct4_run(lc3o_dct4 *d, float *buf) {
    int n, nf = d->nf, count = nf / 2;
    for (n = 0; n < count; n++) {
        lc3o_cpx c = {buf[2 * n], buf[nf - 2 * n - 1]};
        d->in[n] = cmul(d->tw[n], c);
    }
    lc3o_fft_run(&d->fft, d->in, d->out);
    for (n = 0; n < count; n++) {
        lc3o_cpx c = cmul(d->tw[n], d->out[n]);
        buf[2 * n] = c.r * 2.0f;
        buf[nf - 2 * n - 1] = -c.i * 2.0f;
    }
}
