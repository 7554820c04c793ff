// Host check of the exp inside fig_weights_n (fig_engine_shared.h): the same operation sequence in C (fma = the device's
// v_fma_f64, rint = v_rndne_f64, ldexp = v_ldexp_f64), compared with expl() and with glibc's exp() over the argument range of
// the weights, x = 0.5 log10(p) in [-162, 0].   gcc -O2 -ffp-contract=off -o fast_exp_check fast_exp_check.c -lm
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static double fast_exp(double x) {
    const double L2E = 0x1.71547652b82fep+0, LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    const double kd = rint(x * L2E);
    const double r_hi = fma(-kd, LN2_HI, x), r_lo = -kd * LN2_LO;     // r_hi is exact (kd * LN2_HI has <= 44 bits)
    const double r = r_hi + r_lo;
    double q = 1.0 / 6227020800.0;                                    // 1/13!
    q = fma(q, r, 1.0 / 479001600.0); q = fma(q, r, 1.0 / 39916800.0); q = fma(q, r, 1.0 / 3628800.0); q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0); q = fma(q, r, 1.0 / 5040.0); q = fma(q, r, 1.0 / 720.0); q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0); q = fma(q, r, 1.0 / 6.0); q = fma(q, r, 0.5);
    const double t = r * r * q;                                       // exp(r) = 1 + r_hi + r_lo + t: 1 + r_hi as an exact sum s + e
    const double s = 1.0 + r_hi, e = (1.0 - s) + r_hi;
    const double res = ldexp(s + ((e + r_lo) + t), (int)kd);
    return x < -745.0 ? 0.0 : res;
}
static double ulp_of(double v) { int e; frexp(v, &e); return ldexp(1.0, e - 53); }
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 20000000;
    uint64_t st = 88172645463325252ull;
    double maxu = 0, maxg = 0; long ndiff = 0, n1 = 0;
    for (long i = 0; i < n; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double x = -162.0 * (double)(st >> 11) / 9007199254740992.0;
        if (i % 50 == 0) x = -ldexp((double)(st >> 12), -52 - (int)(st % 40));       // small |x|
        long double ref = expl((long double)x);
        double f = fast_exp(x), g = exp(x), u = ulp_of((double)ref);
        double ef = fabs((double)((long double)f - ref)) / u, eg = fabs((double)((long double)g - ref)) / u;
        if (ef > maxu) maxu = ef;
        if (eg > maxg) maxg = eg;
        if (ef > 1.0) n1++;
        if (f != g) ndiff++;
    }
    printf("n=%ld  fast: max %.3f ulp, %ld over 1 ulp;  glibc exp: max %.3f ulp;  fast != glibc on %.3f %%\n", n, maxu, n1, maxg, 100.0 * ndiff / n);
    printf("exact cases: exp(0)=%g exp(-inf)=%g exp(-800)=%g\n", fast_exp(0.0), fast_exp(-INFINITY), fast_exp(-800.0));
    return 0;
}
