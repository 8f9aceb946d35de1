// dist_fam.h -- per-family arithmetic of the element-wise sample-site kernels (log density and
// its partial derivatives), shared by dist.hip and multisite.hip.
#pragma once
#include "common.h"

namespace pa {

template <typename T> struct Consts;
template <> struct Consts<float> {
  static constexpr float half_log_2pi = 0.91893853320467274178f;
  static constexpr float log_2_over_pi = -0.45158270528945486473f;  // log(2) - log(pi)
  static constexpr float log2 = 0.69314718055994530942f;
};
template <> struct Consts<double> {
  static constexpr double half_log_2pi = 0.91893853320467274178;
  static constexpr double log_2_over_pi = -0.45158270528945486473;
  static constexpr double log2 = 0.69314718055994530942;
};

template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log(double x) { return log(x); }
template <typename T> __device__ __forceinline__ T t_log1p(T x);
template <> __device__ __forceinline__ float t_log1p(float x) { return log1pf(x); }
template <> __device__ __forceinline__ double t_log1p(double x) { return log1p(x); }
template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }
template <typename T> __device__ __forceinline__ T t_inf();
template <> __device__ __forceinline__ float t_inf() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double t_inf() { return __builtin_huge_val(); }

// ---- per-family arithmetic ------------------------------------------------------------------
// lp(v,a,b) and grad(v,a,b,&dv,&da,&db) = partial derivatives of lp.
template <int DIST, typename T> struct Fam;

template <typename T> struct Fam<PA_DIST_NORMAL, T> {  // a=loc b=scale; torch normal.py:88-103
  static __device__ __forceinline__ T lp(T v, T a, T b) {
    T d = v - a;
    return -(d * d) / (T(2) * b * b) - t_log(b) - Consts<T>::half_log_2pi;
  }
  static __device__ __forceinline__ void grad(T v, T a, T b, T& dv, T& da, T& db) {
    T d = v - a, iv = T(1) / (b * b);
    da = d * iv;
    dv = -da;
    db = d * d * iv / b - T(1) / b;
  }
};
template <typename T> struct Fam<PA_DIST_BERNOULLI_LOGITS, T> {  // a=logits; bernoulli.py:121-125
  static __device__ __forceinline__ T lp(T v, T a, T) {
    // -BCEWithLogits(a, v) = v*a - softplus(a), softplus(a) = max(a,0) + log1p(exp(-|a|))
    return v * a - ((a > T(0) ? a : T(0)) + t_log1p(t_exp(-t_abs(a))));
  }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    T e = t_exp(-t_abs(a));
    T sig = a >= T(0) ? T(1) / (T(1) + e) : e / (T(1) + e);
    da = v - sig;
    dv = a;
    db = T(0);
  }
};
template <typename T> struct Fam<PA_DIST_HALF_CAUCHY, T> {  // a=scale; half_cauchy.py:74-83
  static __device__ __forceinline__ T lp(T v, T a, T) {
    T q = v / a;
    T r = Consts<T>::log_2_over_pi - t_log(a) - t_log1p(q * q);
    return v >= T(0) ? r : -t_inf<T>();
  }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    T den = a * a + v * v;
    dv = -T(2) * v / den;
    da = (v * v - a * a) / (a * den);
    db = T(0);
  }
};
template <typename T> struct Fam<PA_DIST_LOG_NORMAL, T> {  // Normal(a,b) pushed through exp
  static __device__ __forceinline__ T lp(T v, T a, T b) {
    T lv = t_log(v);
    return Fam<PA_DIST_NORMAL, T>::lp(lv, a, b) - lv;
  }
  static __device__ __forceinline__ void grad(T v, T a, T b, T& dv, T& da, T& db) {
    T lv = t_log(v), dn;
    Fam<PA_DIST_NORMAL, T>::grad(lv, a, b, dn, da, db);
    dv = (dn - T(1)) / v;
  }
};
template <typename T> struct Fam<PA_DIST_EXPONENTIAL, T> {  // a=rate
  static __device__ __forceinline__ T lp(T v, T a, T) { return t_log(a) - a * v; }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    dv = -a;
    da = T(1) / a - v;
    db = T(0);
  }
};
template <typename T> struct Fam<PA_DIST_HALF_NORMAL, T> {  // a=scale; half_normal.py
  static __device__ __forceinline__ T lp(T v, T a, T) {
    T r = Fam<PA_DIST_NORMAL, T>::lp(v, T(0), a) + Consts<T>::log2;
    return v >= T(0) ? r : -t_inf<T>();
  }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    T d0;
    Fam<PA_DIST_NORMAL, T>::grad(v, T(0), a, dv, d0, da);
    db = T(0);
  }
};

template <int DIST> struct NParams { static constexpr int n = 1; };
template <> struct NParams<PA_DIST_NORMAL> { static constexpr int n = 2; };
template <> struct NParams<PA_DIST_LOG_NORMAL> { static constexpr int n = 2; };


#define PA_DISPATCH_DIST(DIST_ID, T, CALL)                                              \
  switch (DIST_ID) {                                                                    \
    case PA_DIST_NORMAL: { constexpr int D_ = PA_DIST_NORMAL; CALL; } break;              \
    case PA_DIST_BERNOULLI_LOGITS: { constexpr int D_ = PA_DIST_BERNOULLI_LOGITS; CALL; } break; \
    case PA_DIST_HALF_CAUCHY: { constexpr int D_ = PA_DIST_HALF_CAUCHY; CALL; } break;    \
    case PA_DIST_LOG_NORMAL: { constexpr int D_ = PA_DIST_LOG_NORMAL; CALL; } break;      \
    case PA_DIST_EXPONENTIAL: { constexpr int D_ = PA_DIST_EXPONENTIAL; CALL; } break;    \
    case PA_DIST_HALF_NORMAL: { constexpr int D_ = PA_DIST_HALF_NORMAL; CALL; } break;    \
    default: return fail(PA_ERR_UNSUPPORTED, "distribution id %d not implemented", DIST_ID); \
  }


}  // namespace pa
