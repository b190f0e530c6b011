// counted.h -- a double that counts its arithmetic.  tests/flop_count.py compiles oracle/redmax_tensorfree.c (the scalar CPU twin
// of the kernels' algorithm) as C++ with `double` replaced by this type, to obtain the flops the ALGORITHM needs per residual
// evaluation, per Hessian and per solve ("useful" work: no idle lanes, no replicated pivot columns, no masked MFMA tiles).
// Counting rule of SURVEY.md 8(d): 1 per add / subtract / multiply (an FMA is 2); divisions, square roots and sin / cos are
// counted separately.  Negation, comparisons, fabs and copies are free.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

extern long long fc_add, fc_mul, fc_div, fc_trans;

struct cd {
    double v;
    cd() {}
    cd(double x) : v(x) {}
    cd(int x) : v(x) {}
    cd(long x) : v((double)x) {}
    cd(unsigned long x) : v((double)x) {}
    cd& operator+=(const cd& o) { ++fc_add; v += o.v; return *this; }
    cd& operator-=(const cd& o) { ++fc_add; v -= o.v; return *this; }
    cd& operator*=(const cd& o) { ++fc_mul; v *= o.v; return *this; }
    cd& operator/=(const cd& o) { ++fc_div; v /= o.v; return *this; }
    cd operator-() const { return cd(-v); }
    explicit operator bool() const { return v != 0.0; }
};
inline cd operator+(const cd& a, const cd& b) { ++fc_add; return cd(a.v + b.v); }
inline cd operator-(const cd& a, const cd& b) { ++fc_add; return cd(a.v - b.v); }
inline cd operator*(const cd& a, const cd& b) { ++fc_mul; return cd(a.v * b.v); }
inline cd operator/(const cd& a, const cd& b) { ++fc_div; return cd(a.v / b.v); }
inline bool operator<(const cd& a, const cd& b) { return a.v < b.v; }
inline bool operator>(const cd& a, const cd& b) { return a.v > b.v; }
inline bool operator<=(const cd& a, const cd& b) { return a.v <= b.v; }
inline bool operator>=(const cd& a, const cd& b) { return a.v >= b.v; }
inline bool operator==(const cd& a, const cd& b) { return a.v == b.v; }
inline bool operator!=(const cd& a, const cd& b) { return a.v != b.v; }
inline cd fabs(const cd& a) { return cd(::fabs(a.v)); }
inline cd sqrt(const cd& a) { ++fc_trans; return cd(::sqrt(a.v)); }
inline cd sin(const cd& a) { ++fc_trans; return cd(::sin(a.v)); }
inline cd cos(const cd& a) { ++fc_trans; return cd(::cos(a.v)); }
