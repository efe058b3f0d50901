#include "tables.hpp"

#include <cmath>
#include <mutex>

namespace hpsdf {
namespace {

// Minimal double-double arithmetic (error-free transformations); only used to
// round Gauss-Legendre nodes and weights correctly to double.
struct DD {
    double hi = 0.0, lo = 0.0;
    DD() = default;
    DD(double h) : hi(h), lo(0.0) {}
    DD(double h, double l) : hi(h), lo(l) {}
};
inline DD fastTwoSum(double a, double b) {
    const double s = a + b;
    return DD(s, b - (s - a));
}
inline DD twoSum(double a, double b) {
    const double s = a + b;
    const double v = s - a;
    return DD(s, (a - (s - v)) + (b - v));
}
inline DD twoProd(double a, double b) {
    const double p = a * b;
    return DD(p, std::fma(a, b, -p));
}
inline DD operator+(const DD& a, const DD& b) {
    DD s = twoSum(a.hi, b.hi);
    const DD t = twoSum(a.lo, b.lo);
    s.lo += t.hi;
    s = fastTwoSum(s.hi, s.lo);
    s.lo += t.lo;
    return fastTwoSum(s.hi, s.lo);
}
inline DD operator-(const DD& a) { return DD(-a.hi, -a.lo); }
inline DD operator-(const DD& a, const DD& b) { return a + (-b); }
inline DD operator*(const DD& a, const DD& b) {
    DD p = twoProd(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return fastTwoSum(p.hi, p.lo);
}
inline DD operator/(const DD& a, const DD& b) {
    const double q1 = a.hi / b.hi;
    DD r = a - b * DD(q1);
    const double q2 = r.hi / b.hi;
    r = r - b * DD(q2);
    const double q3 = r.hi / b.hi;
    return fastTwoSum(q1, q2) + DD(q3);
}

// P_n and P_n' at x
void legendre(int n, const DD& x, DD& pn, DD& dpn) {
    DD prev(1.0), cur = x;
    for (int k = 2; k <= n; ++k) {
        const DD next = (DD(2.0 * k - 1.0) * x * cur - DD(k - 1.0) * prev) / DD(double(k));
        prev = cur;
        cur = next;
    }
    pn = cur;
    dpn = DD(double(n)) * (x * cur - prev) / (x * x - DD(1.0));
}

// The n-point rule in the storage order of Include/HP/Legendre.h: the zero
// node first when n is odd, then (-x, +x) pairs by ascending |x|; the pairs of
// n = 6 and n = 9 are stored out of order there and the same permutation is
// applied here (n = 9 is the rule of every degree-2 fit).
void glRule(int n, double* x, double* w) {
    const int pairs = n / 2, odd = n & 1;
    double px[32], pw[32];
    if (odd) {
        DD p, dp;
        legendre(n, DD(0.0), p, dp);
        x[0] = 0.0;
        w[0] = (DD(2.0) / (dp * dp)).hi;
    }
    for (int k = 0; k < pairs; ++k) {
        DD r(std::cos(M_PI * ((pairs - k) - 0.25) / (n + 0.5)));
        DD p, dp;
        for (int it = 0; it < 8; ++it) {
            legendre(n, r, p, dp);
            r = r - p / dp;
        }
        legendre(n, r, p, dp);
        px[k] = r.hi;
        pw[k] = (DD(2.0) / ((DD(1.0) - r * r) * dp * dp)).hi;
    }
    static const int order6[3] = {1, 0, 2};
    static const int order9[4] = {2, 3, 0, 1};
    for (int k = 0; k < pairs; ++k) {
        const int s = n == 6 ? order6[k] : n == 9 ? order9[k] : k;
        x[odd + 2 * k] = -px[s];
        x[odd + 2 * k + 1] = px[s];
        w[odd + 2 * k] = w[odd + 2 * k + 1] = pw[s];
    }
}

// Include/HP/Utility.h:25-35: 100 Newton steps starting at x
double newtonSqrt(double x) {
    double g = x;
    for (int i = 0; i < 100; ++i) g = 0.5 * (g + x / g);
    return g;
}

Tables* build() {
    Tables* t = new Tables();
    for (int n = 1; n <= 64; ++n) glRule(n, t->roots + glOffset(n), t->weights + glOffset(n));
    for (uint64_t i = 0; i < 4 * kMaxDegree + 2; ++i) t->sumToN[i] = i * (i + 1) / 2;
    for (int i = 0; i <= kMaxDegree; ++i) {
        double pw = 1.0;  // 2^j by repeated multiplication (Utility.h:14-24)
        for (int j = 0; j <= kMaxDepth; ++j) {
            t->normalisedLengths[i][j] = newtonSqrt((2.0 * i + 1.0) * pw);
            pw = 2.0 * pw;
        }
    }
    {
        const double sixth = 1.0 / 6.0;  // Utility.h:91-96, evaluated in f64 then truncated
        for (uint64_t i = 0; i <= kMaxDegree; ++i) t->coeffCount[i] = (uint64_t)(sixth * (i + 1) * (i + 2) * (i + 3));
    }
    t->recurrence[0][0] = t->recurrence[0][1] = 0.0;
    for (int i = 1; i <= kMaxDegree; ++i) {
        t->recurrence[i][0] = (2.0 * i - 1.0) / i;
        t->recurrence[i][1] = (i - 1.0) / i;
    }
    int row = 0;  // Utility.h:133-160: total degree, then first and second index ascending
    for (int p = 0; p <= kMaxDegree; ++p)
        for (int a = 0; a <= p; ++a)
            for (int b = 0; a + b <= p; ++b) {
                t->basisIndex[row][0] = a;
                t->basisIndex[row][1] = b;
                t->basisIndex[row][2] = p - a - b;
                ++row;
            }
    return t;
}

}  // namespace

const Tables& tables() {
    static std::once_flag once;
    static Tables* inst = nullptr;
    std::call_once(once, [] { inst = build(); });
    return *inst;
}

}  // namespace hpsdf
