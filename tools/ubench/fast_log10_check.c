#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
// candidate device log10: atanh series, no tables, ~35 FP64 operations
static double fast_log10(double x) {
    if (x == 0.0) return -INFINITY;
    int e; double m = frexp(x, &e);           // m in [0.5, 1)   (v_frexp_mant_f64 / v_frexp_exp_i32_f64 handle denormals)
    if (m < 0.70710678118654752440) { m = m + m; e -= 1; }
    const double a = m - 1.0, b = m + 1.0;
    float rf = 1.0f / (float)b;                      // seed like v_rcp (here only ~24 bits) + one Newton step
    double r = (double)rf; { const double e0 = fma(-b, r, 1.0); r = fma(r, e0, r); }
    const double s = a * r;
    const double z = s * s;
    // log(m) = 2 s (1 + z/3 + z^2/5 + ... ), |s| <= 0.1716, z <= 0.02944
    double q = 2.0 / 23.0;
    q = fma(q, z, 2.0 / 21.0); q = fma(q, z, 2.0 / 19.0); q = fma(q, z, 2.0 / 17.0); q = fma(q, z, 2.0 / 15.0); q = fma(q, z, 2.0 / 13.0);
    q = fma(q, z, 2.0 / 11.0); q = fma(q, z, 2.0 / 9.0); q = fma(q, z, 2.0 / 7.0); q = fma(q, z, 2.0 / 5.0); q = fma(q, z, 2.0 / 3.0);
    // lm = 2 s + s z q, with the rounding error of s = a / b recovered: s_lo = (a - s b) / b
    const double sb_err = fma(-s, b, a);       // a - s*b exactly (b's own rounding error ignored: m + 1 is exact for m >= 1... see test)
    const double s_lo = sb_err * r;
    const double t = s * z * q;
    const double lm_hi = 2.0 * s;
    const double lm_lo = fma(2.0, s_lo, t);
    // log10(x) = e log10(2) + (lm_hi + lm_lo) log10(e)
    const double L2hi = 0x1.34413509f78p-2, L2lo = 0x1.fef311f12b358p-46;      // log10(2) split: hi has 41 significant bits
    const double IE_hi = 0x1.bcb7b1526e50ep-2, IE_lo = 0x1.95355baaafad3p-57;    // log10(e)
    const double ed = (double)e;
    const double p_hi = lm_hi * IE_hi;
    const double p_lo = fma(lm_hi, IE_hi, -p_hi) + fma(lm_hi, IE_lo, lm_lo * IE_hi);
    const double r_hi = ed * L2hi;             // exact
    double sum = r_hi + p_hi;
    double err = (r_hi - sum) + p_hi;          // Fast2Sum (|r_hi| >= |p_hi| or r_hi == 0)
    if (e == 0) { sum = p_hi; err = 0; }
    return sum + (err + fma(ed, L2lo, p_lo));
}
static double ulp_of(double v) { int e; frexp(v, &e); return ldexp(1.0, e - 53); }
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 20000000;
    uint64_t st = 88172645463325252ull;
    double maxu = 0, maxu_g = 0; long n1 = 0, ng1 = 0, ndiff_g = 0; double worst = 0;
    for (long i = 0; i < n; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        // x in (0, 1]: exponent uniform over the whole range incl. denormals, mantissa random
        int ex = (int)((st >> 52) % 1075);       // 0..1074
        uint64_t man = st & 0xfffffffffffffull;
        double x = ldexp(1.0 + (double)man / 4503599627370496.0, -ex - 1);
        if (i % 97 == 0) x = 1.0 - ldexp((double)(man >> 20), -53 - (int)(st % 20));   // near 1
        if (x <= 0 || x > 1) continue;
        long double ref = log10l((long double)x);
        double f = fast_log10(x), g = log10(x);
        double u = ulp_of((double)ref);
        if (ref == 0) continue;
        double ef = fabs((double)((long double)f - ref)) / u, eg = fabs((double)((long double)g - ref)) / u;
        if (ef > maxu) { maxu = ef; worst = x; }
        if (eg > maxu_g) maxu_g = eg;
        if (ef > 1.0) n1++;
        if (eg > 1.0) ng1++;
        if (f != g) ndiff_g++;
    }
    printf("n=%ld  fast: max err %.3f ulp (x=%a), >1ulp: %ld   glibc: max %.3f ulp, >1ulp %ld   fast != glibc: %ld (%.2f %%)\n", n, maxu, worst, n1, maxu_g, ng1, ndiff_g, 100.0 * ndiff_g / n);
    printf("log10(0)=%g log10(1)=%g log10(denorm_min)=%.17g vs %.17g\n", fast_log10(0.0), fast_log10(1.0), fast_log10(4.9406564584124654e-324), log10(4.9406564584124654e-324));
    return 0;
}
