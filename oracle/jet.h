// TEST INFRASTRUCTURE — CPU oracle for the mv-lm-icp hot path. NOT part of the shipped product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Forward-mode dual numbers ("Jets"), restating the published behaviour of ceres::Jet<double,N>
// (Ceres Solver < 2.2, include/ceres/jet.h — third-party dependency of the reference, NOT in
// /root/reference; the reference relies on it through ceres::AutoDiffCostFunction at
// /root/reference/include/icp-ceres.h:62,110,155,200,249,291).  A Jet carries a value `a` and N
// partial derivatives `v`; every arithmetic operator applies the chain rule exactly as Ceres does,
// so evaluating one of the restated cost functors on Jets yields the same Jacobian Ceres' autodiff
// would hand to its trust-region solver.
#pragma once
#include <cmath>

namespace orc {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  explicit Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  // ceres/jet.h: h = f/g ; h.v = (f.v - h.a * g.v) / g.a
  Jet<N> h; const double ginv = 1.0 / g.a; h.a = f.a * ginv;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - h.a * g.v[i]) * ginv;
  return h;
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) {
  Jet<N> h; h.a = s / g.a; const double m = -s / (g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = g.v[i] * m;
  return h;
}
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, const Jet<N>& g) { f = f * g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, double s) { f = f * s; return f; }

template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double s) { return f.a < s; }
template <int N> inline bool operator>(const Jet<N>& f, double s) { return f.a > s; }
template <int N> inline bool operator>=(const Jet<N>& f, double s) { return f.a >= s; }
template <int N> inline bool operator<=(const Jet<N>& f, double s) { return f.a <= s; }

template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
  Jet<N> h; h.a = std::sqrt(f.a); const double k = 1.0 / (2.0 * h.a);
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * k;
  return h;
}
template <int N> inline Jet<N> sin(const Jet<N>& f) {
  Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i];
  return h;
}
template <int N> inline Jet<N> cos(const Jet<N>& f) {
  Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i];
  return h;
}
template <int N> inline Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  // d atan2(g,f) = (f dg - g df) / (f^2 + g^2)
  Jet<N> h; h.a = std::atan2(g.a, f.a); const double k = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = (f.a * g.v[i] - g.a * f.v[i]) * k;
  return h;
}

inline double sqrt(double x) { return std::sqrt(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Jet<N>& x) { return x.a; }

}  // namespace orc
