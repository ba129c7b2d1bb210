// dual.hpp -- forward-mode differentiation scalars for the continuous-dynamics kernels.
//
// Dual<T> = v + d eps with eps^2 = 0: running a kernel body on Dual<double> instead of double yields, next to every value, its
// exact derivative along one input direction (no step size, no truncation: every elementary operation is differentiated by its
// rule, as the reference's hand-written tables do term by term).  Dual<Dual<double>> carries two directions and their mixed second
// derivative (the d.d part).  The second derivatives of the continuous dynamics (reference calc_dynamics_deriv2,
// system.c:1301-2029) are the first-derivative kernel (MODE_DYN_DERIV1) run on Dual<double> once per input variable; the third-
// and fourth-order Lagrangian derivatives (System_L_dqdqdq ... L_ddqddqdqdq, system.c:204-622) the Lagrangian kernel on
// Dual<double> / Dual<Dual<double>>.  Comparisons (pivot choice, spline piece, tolerance) look at the value part only.
//
// The namespace is separate from tg:: on purpose: inside tg:: an unqualified fma / sqrt / fabs then still finds the global
// double versions by ordinary lookup and these overloads by argument-dependent lookup.
#pragma once
#include <cmath>
#include <type_traits>

#if defined(__HIPCC__)
#define TGD_HD __host__ __device__ __forceinline__
#else
#define TGD_HD inline
#endif

namespace tgdual {

template <class T> struct Dual {
    T v, d;
    TGD_HD Dual() {}
    TGD_HD Dual(double a) : v(a), d(0.0) {}
    TGD_HD Dual(const T &v_, const T &d_) : v(v_), d(d_) {}
};

template <class X> struct is_dual : std::false_type {};
template <class T> struct is_dual<Dual<T>> : std::true_type {};

// the plain number underneath (comparisons, table look-ups) and the highest-order coefficient (the derivative asked for)
TGD_HD double primal(double x) { return x; }
template <class T> TGD_HD double primal(const Dual<T> &x) { return primal(x.v); }
TGD_HD double top(double x) { return x; }
template <class T> TGD_HD double top(const Dual<T> &x) { return top(x.d); }

// value + eps_inner [s1] + eps_outer [s2]
template <class R> struct Seed;
template <> struct Seed<double> { TGD_HD static double make(double v, bool, bool) { return v; } };
template <> struct Seed<Dual<double>> { TGD_HD static Dual<double> make(double v, bool s1, bool) { return Dual<double>(v, s1 ? 1.0 : 0.0); } };
template <> struct Seed<Dual<Dual<double>>> {
    TGD_HD static Dual<Dual<double>> make(double v, bool s1, bool s2) {
        return Dual<Dual<double>>(Dual<double>(v, s1 ? 1.0 : 0.0), Dual<double>(s2 ? 1.0 : 0.0, 0.0));
    }
};

template <class T> TGD_HD Dual<T> operator-(const Dual<T> &a) { return Dual<T>(-a.v, -a.d); }
template <class T> TGD_HD Dual<T> operator+(const Dual<T> &a) { return a; }
template <class T> TGD_HD Dual<T> operator+(const Dual<T> &a, const Dual<T> &b) { return Dual<T>(a.v + b.v, a.d + b.d); }
template <class T> TGD_HD Dual<T> operator-(const Dual<T> &a, const Dual<T> &b) { return Dual<T>(a.v - b.v, a.d - b.d); }
template <class T> TGD_HD Dual<T> operator*(const Dual<T> &a, const Dual<T> &b) { return Dual<T>(a.v * b.v, a.v * b.d + a.d * b.v); }
template <class T> TGD_HD Dual<T> operator/(const Dual<T> &a, const Dual<T> &b) {
    const T q = a.v / b.v;
    return Dual<T>(q, (a.d - q * b.d) / b.v);
}
template <class T> TGD_HD Dual<T> operator+(const Dual<T> &a, double b) { return Dual<T>(a.v + b, a.d); }
template <class T> TGD_HD Dual<T> operator+(double a, const Dual<T> &b) { return Dual<T>(a + b.v, b.d); }
template <class T> TGD_HD Dual<T> operator-(const Dual<T> &a, double b) { return Dual<T>(a.v - b, a.d); }
template <class T> TGD_HD Dual<T> operator-(double a, const Dual<T> &b) { return Dual<T>(a - b.v, -b.d); }
template <class T> TGD_HD Dual<T> operator*(const Dual<T> &a, double b) { return Dual<T>(a.v * b, a.d * b); }
template <class T> TGD_HD Dual<T> operator*(double a, const Dual<T> &b) { return Dual<T>(a * b.v, a * b.d); }
template <class T> TGD_HD Dual<T> operator/(const Dual<T> &a, double b) { return Dual<T>(a.v / b, a.d / b); }
template <class T> TGD_HD Dual<T> operator/(double a, const Dual<T> &b) { return Dual<T>(a) / b; }
template <class T, class B> TGD_HD Dual<T> &operator+=(Dual<T> &a, const B &b) { a = a + b; return a; }
template <class T, class B> TGD_HD Dual<T> &operator-=(Dual<T> &a, const B &b) { a = a - b; return a; }
template <class T, class B> TGD_HD Dual<T> &operator*=(Dual<T> &a, const B &b) { a = a * b; return a; }
template <class T, class B> TGD_HD Dual<T> &operator/=(Dual<T> &a, const B &b) { a = a / b; return a; }

#define TGD_COMPARE(op) \
    template <class T> TGD_HD bool operator op(const Dual<T> &a, const Dual<T> &b) { return primal(a) op primal(b); } \
    template <class T> TGD_HD bool operator op(const Dual<T> &a, double b) { return primal(a) op b; } \
    template <class T> TGD_HD bool operator op(double a, const Dual<T> &b) { return a op primal(b); }
TGD_COMPARE(<) TGD_COMPARE(>) TGD_COMPARE(<=) TGD_COMPARE(>=) TGD_COMPARE(==) TGD_COMPARE(!=)
#undef TGD_COMPARE

TGD_HD void sincos_of(double x, double *s, double *c) { *s = ::sin(x); *c = ::cos(x); }
template <class T> TGD_HD void sincos_of(const Dual<T> &x, Dual<T> *s, Dual<T> *c) {
    T sv, cv;
    sincos_of(x.v, &sv, &cv);
    *s = Dual<T>(sv, cv * x.d);
    *c = Dual<T>(cv, -(sv * x.d));
}
template <class T> TGD_HD Dual<T> sqrt(const Dual<T> &x) {
    using ::sqrt;
    const T r = sqrt(x.v);
    return Dual<T>(r, x.d / (2.0 * r));
}
template <class T> TGD_HD Dual<T> fabs(const Dual<T> &x) { return primal(x) < 0.0 ? -x : x; }
template <class A, class B, class C, class = typename std::enable_if<is_dual<A>::value || is_dual<B>::value || is_dual<C>::value>::type>
TGD_HD auto fma(const A &a, const B &b, const C &c) -> decltype(a * b + c) { return a * b + c; }

}  // namespace tgdual
