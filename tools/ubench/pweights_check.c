// Host check of the partial-mode weight function of figbird_amd/csrc/fig_engine_partial.h (fig_pweights: t = ln p, w = 10^t
// from the ROUNDED t, as Figbird.cpp:3169-3179 takes pow(10, log(p))): the same operation sequence in plain C (frexp for
// v_frexp_mant/exp, 1/b for v_rcp_f64 -- both are refined by the same Newton step) against glibc's log and pow.
//   gcc -O2 -ffp-contract=off -o /tmp/pweights_check tools/ubench/pweights_check.c -lm && /tmp/pweights_check 20000000
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void pweights(double x, double *t_out, double *w_out) {
    int e; double m = frexp(x, &e);                       // m in [0.5, 1)
    const int lowhalf = m < 0.70710678118654752440;
    m = lowhalf ? m + m : m; e = lowhalf ? e - 1 : e;
    const double a = m - 1.0, b = m + 1.0;
    double r = 1.0 / b;
    { const double e0 = fma(-b, r, 1.0); r = fma(r, e0, r); }
    const double s = a * r, z = s * s;
    double q = 2.0 / 23.0;
    q = fma(q, z, 2.0 / 21.0); q = fma(q, z, 2.0 / 19.0); q = fma(q, z, 2.0 / 17.0); q = fma(q, z, 2.0 / 15.0); q = fma(q, z, 2.0 / 13.0);
    q = fma(q, z, 2.0 / 11.0); q = fma(q, z, 2.0 / 9.0); q = fma(q, z, 2.0 / 7.0); q = fma(q, z, 2.0 / 5.0); q = fma(q, z, 2.0 / 3.0);
    const double s_lo = fma(-s, b, a) * r;
    const double t3 = s * z * q;
    const double lm_hi = 2.0 * s, lm_lo = fma(2.0, s_lo, t3);
    const double LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    const double ed = (double)e;
    const double r_hi = ed * LN2_HI;
    double sum = r_hi + lm_hi, err = (r_hi - sum) + lm_hi;
    if (e == 0) { sum = lm_hi; err = 0.0; }
    double t = sum + (err + fma(ed, LN2_LO, lm_lo));
    if (x == 0.0) t = -INFINITY;
    // 10^t
    const double C_HI = 0x1.26bb1bbb55516p+1, C_LO = -0x1.f48ad494ea3e9p-53, L2E = 0x1.71547652b82fep+0;
    const double u_hi = t * C_HI, u_lo = fma(t, C_HI, -u_hi) + t * C_LO;
    const double kd = rint(u_hi * L2E);
    const double rh = fma(-kd, LN2_HI, u_hi), rl = fma(-kd, LN2_LO, u_lo), rr = rh + rl;
    double g = 1.0 / 6227020800.0;
    g = fma(g, rr, 1.0 / 479001600.0); g = fma(g, rr, 1.0 / 39916800.0); g = fma(g, rr, 1.0 / 3628800.0); g = fma(g, rr, 1.0 / 362880.0);
    g = fma(g, rr, 1.0 / 40320.0); g = fma(g, rr, 1.0 / 5040.0); g = fma(g, rr, 1.0 / 720.0); g = fma(g, rr, 1.0 / 120.0);
    g = fma(g, rr, 1.0 / 24.0); g = fma(g, rr, 1.0 / 6.0); g = fma(g, rr, 0.5);
    const double tt = rr * rr * g;
    const double ss = 1.0 + rh, ee = (1.0 - ss) + rh;
    double w = ldexp(ss + ((ee + rl) + tt), (int)kd);
    if (!(t >= -330.0)) w = 0.0;
    *t_out = t; *w_out = w;
}

static int64_t ulps(double a, double b) { int64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return x > y ? x - y : y - x; }

int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 10000000;
    uint64_t st = 0x9E3779B97F4A7C15ull;
    long same_t = 0, same_w = 0, same_w_given_t = 0, nw_t = 0; int64_t max_t = 0, max_w = 0;
    for (long i = 0; i < n; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        // products of <= 101 probabilities: exponent uniform in [2^-1070, 1], random mantissa; every 16th near 1
        double m = 1.0 + (double)(st >> 12) / 4503599627370496.0;
        int e = (i & 15) == 0 ? -(int)((st >> 3) % 4) : -(int)((st >> 3) % 1070);
        double x = ldexp(m / 2, e + 1);
        if (x > 1.0) x = 1.0;
        double t, w; pweights(x, &t, &w);
        const double tr = log(x), wr = pow(10.0, tr);
        int64_t ut = ulps(t, tr), uw = ulps(w, wr);
        if (ut == 0) { same_t++; nw_t++; if (uw == 0) same_w_given_t++; }
        if (uw == 0) same_w++;
        if (ut > max_t) max_t = ut;
        if (ut == 0 && wr >= 2.3e-308 && uw > max_w) max_w = uw;
    }
    printf("n = %ld: ln p equal to glibc on %.4f %% (max %lld ulp); 10^t equal on %.4f %% overall, %.4f %% where ln p agreed (max %lld ulp among normal results with the same ln p)\n",
           n, 100.0 * same_t / n, (long long)max_t, 100.0 * same_w / n, 100.0 * same_w_given_t / (nw_t ? nw_t : 1), (long long)max_w);
    double t, w; pweights(0.0, &t, &w); printf("p = 0: t = %g w = %g;  ", t, w); pweights(1.0, &t, &w); printf("p = 1: t = %g w = %g\n", t, w);
    return 0;
}
