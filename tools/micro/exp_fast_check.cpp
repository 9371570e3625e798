// Accuracy of exp_fast (remixt_amd/csrc/rmx_device.h) against long-double expl over 2e7 arguments: g++ -O2 -ffp-contract=off exp_fast_check.cpp && ./a.out
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
static inline double exp_fast(double x) {
    const double k = std::rint(x * 1.4426950408889634074);
    double r = std::fma(-k, 6.93147180369123816490e-01, x);
    r = std::fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;        // 1/13!
    p = std::fma(p, r, 2.08767569878681e-09);  // 1/12!
    p = std::fma(p, r, 2.505210838544172e-08); // 1/11!
    p = std::fma(p, r, 2.755731922398589e-07); // 1/10!
    p = std::fma(p, r, 2.7557319223985893e-06);// 1/9!
    p = std::fma(p, r, 2.48015873015873e-05);  // 1/8!
    p = std::fma(p, r, 1.984126984126984e-04); // 1/7!
    p = std::fma(p, r, 1.388888888888889e-03); // 1/6!
    p = std::fma(p, r, 8.333333333333333e-03); // 1/5!
    p = std::fma(p, r, 4.1666666666666664e-02);// 1/4!
    p = std::fma(p, r, 1.6666666666666666e-01);// 1/3!
    p = std::fma(p, r, 0.5);
    p = std::fma(p, r, 1.0);
    p = std::fma(p, r, 1.0);
    double v = std::ldexp(p, (int)k);
    return x < -746. ? 0. : v;
}
static double ulp_err(double a, long double ref) {
    if (ref == 0) return a == 0 ? 0 : 1e9;
    int e; std::frexp((double)ref, &e);
    long double ulp = std::ldexp(1.0L, e - 53);
    return (double)(fabsl((long double)a - ref) / ulp);
}
int main() {
    std::mt19937_64 g(1);
    double worst = 0, worstx = 0; double worst_lib = 0;
    for (int i = 0; i < 20000000; i++) {
        double x;
        int m = i % 4;
        if (m == 0) x = -std::uniform_real_distribution<double>(0, 1)(g);
        else if (m == 1) x = -std::uniform_real_distribution<double>(0, 40)(g);
        else if (m == 2) x = -std::uniform_real_distribution<double>(0, 700)(g);
        else x = std::uniform_real_distribution<double>(-1e-3, 1e-3)(g);
        long double ref = expl((long double)x);
        double e = ulp_err(exp_fast(x), ref);
        if (e > worst) { worst = e; worstx = x; }
        double e2 = ulp_err(std::exp(x), ref);
        if (e2 > worst_lib) worst_lib = e2;
    }
    printf("exp_fast worst %.3f ulp at x=%.17g; libm worst %.3f ulp\n", worst, worstx, worst_lib);
    printf("edge: %g %g %g %g %g\n", exp_fast(-745.2), exp_fast(-800.), exp_fast(-INFINITY), exp_fast(0.), exp_fast(NAN));
    printf("denormal region: x=-740 fast %.17g lib %.17g\n", exp_fast(-740.), std::exp(-740.));
    return 0;
}
