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

// log and reciprocal of a positive, normal-range parameter (a scale).  f32: v_log_f32 / v_rcp_f32
// (1 ulp) instead of the library log and the IEEE division sequence -- a large Normal site is
// VALU-bound with those (2.9 TB/s on [64, 1e6]), HBM-bound without.  f64: library / true division.
template <typename T> __device__ __forceinline__ T pos_log(T x) { return t_log(x); }
template <> __device__ __forceinline__ float pos_log(float x) {
  return 0.69314718055994530942f * __builtin_amdgcn_logf(x);
}
template <typename T> __device__ __forceinline__ T pos_rcp(T x) { return T(1) / x; }
template <> __device__ __forceinline__ float pos_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <typename T> struct Fam<PA_DIST_NORMAL, T> {  // a=loc b=scale; torch normal.py:88-103
  static __device__ __forceinline__ T lp(T v, T a, T b) {
    const T d = (v - a) * pos_rcp(b);
    return T(-0.5) * d * d - pos_log(b) - Consts<T>::half_log_2pi;
  }
  static __device__ __forceinline__ void grad(T v, T a, T b, T& dv, T& da, T& db) {
    const T ib = pos_rcp(b), d = (v - a) * ib;
    da = d * ib;
    dv = -da;
    db = (d * d - T(1)) * ib;
  }
};
// exp(-|a|) and log1p of it.  f32: the hardware exp2 / log2 (v_exp_f32, v_log_f32) -- e in (0, 1]
// so 1 + e is exact to 6e-8 ABSOLUTE and log2(1 + e) * ln 2 carries that absolute error into a term
// of size O(1): the same arithmetic as the fused GLM kernels (glm.hip), 3x fewer instructions than
// the library expf / log1pf on a site of 6.4e7 elements (which is VALU-bound, not HBM-bound, with
// the library calls).  f64 keeps the library functions.
template <typename T> __device__ __forceinline__ T exp_neg_abs(T a) { return t_exp(-t_abs(a)); }
template <> __device__ __forceinline__ float exp_neg_abs(float a) {
  return __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(a));
}
template <typename T> __device__ __forceinline__ T log1p_unit(T e) { return t_log1p(e); }
template <> __device__ __forceinline__ float log1p_unit(float e) {
  return 0.69314718055994530942f * __builtin_amdgcn_logf(1.0f + e);
}

template <typename T> struct Fam<PA_DIST_BERNOULLI_LOGITS, T> {  // a=logits; bernoulli.py:121-125
  static __device__ __forceinline__ T lp(T v, T a, T) {
    // -BCEWithLogits(a, v) = v*a - softplus(a), softplus(a) = max(a,0) + log1p(exp(-|a|))
    return v * a - ((a > T(0) ? a : T(0)) + log1p_unit(exp_neg_abs(a)));
  }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    T e = exp_neg_abs(a);
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

// ---- families with Gamma-function normalisers ---------------------------------------------------
template <typename T> __device__ __forceinline__ T t_lgamma(T x);
template <> __device__ __forceinline__ float t_lgamma(float x) { return lgammaf(x); }
// f64: an out-of-line call -- the library lgamma inlined into the one-workgroup site kernels
// (1024 threads, 128 VGPRs) spills every family's path, not just the Gamma-function ones
template <> inline __device__ __noinline__ double t_lgamma(double x) { return lgamma(x); }
// x * log(y) with the convention 0 * log(0) = 0 (torch.xlogy)
template <typename T> __device__ __forceinline__ T t_xlogy(T x, T y) {
  return x == T(0) ? T(0) : x * t_log(y);
}
// psi(x), x > 0: upward recurrence psi(x) = psi(x+1) - 1/x into the asymptotic range, then
//   psi(x) ~ ln x - 1/(2x) - sum_k B_2k / (2k x^2k)
// (the series is cut where its first dropped term is below the type's rounding: x >= 6 with four
// terms for f32 (1e-8), x >= 10 with seven terms for f64 (1e-15)).
template <typename T> __device__ __forceinline__ T t_digamma(T x) {
  constexpr bool f32 = sizeof(T) == 4;
  const T lim = f32 ? T(6) : T(10);
  T r = T(0);
  // the arguments here are concentrations / counts + 1: positive.  Anything else (garbage in a row
  // that a mask removes afterwards: -inf, -1e30 sentinels) must not reach the recurrence -- x + 1
  // == x below -2^24 (f32) and the loop would never end
  if (!(x > T(0))) return (T)__builtin_nan("");
  while (x < lim) {
    r -= T(1) / x;
    x += T(1);
  }
  const T ix = T(1) / x, f = ix * ix;
  T t;
  if constexpr (f32) {
    t = f * (T(-1.0 / 12) + f * (T(1.0 / 120) + f * (T(-1.0 / 252) + f * T(1.0 / 240))));
  } else {
    t = f * (T(-1.0 / 12) + f * (T(1.0 / 120) + f * (T(-1.0 / 252) + f * (T(1.0 / 240) +
        f * (T(-1.0 / 132) + f * (T(691.0 / 32760) + f * T(-1.0 / 12)))))));
  }
  return r + t_log(x) - T(0.5) * ix + t;
}

template <typename T> struct Fam<PA_DIST_GAMMA, T> {  // a=concentration b=rate; torch gamma.py
  static __device__ __forceinline__ T lp(T v, T a, T b) {
    return t_xlogy(a, b) + t_xlogy(a - T(1), v) - b * v - t_lgamma(a);
  }
  static __device__ __forceinline__ void grad(T v, T a, T b, T& dv, T& da, T& db) {
    dv = (a - T(1)) / v - b;
    da = t_log(b) + t_log(v) - t_digamma(a);
    db = a / b - v;
  }
};
template <typename T> struct Fam<PA_DIST_BETA, T> {  // a=concentration1 b=concentration0
  // torch beta.py scores Dirichlet([a, b]) at [v, 1 - v] (dirichlet.py log_prob)
  static __device__ __forceinline__ T lp(T v, T a, T b) {
    return t_xlogy(a - T(1), v) + t_xlogy(b - T(1), T(1) - v) + t_lgamma(a + b) - t_lgamma(a) -
           t_lgamma(b);
  }
  static __device__ __forceinline__ void grad(T v, T a, T b, T& dv, T& da, T& db) {
    const T u = T(1) - v, pab = t_digamma(a + b);
    dv = (a - T(1)) / v - (b - T(1)) / u;
    da = t_log(v) + pab - t_digamma(a);
    db = t_log(u) + pab - t_digamma(b);
  }
};
template <typename T> struct Fam<PA_DIST_POISSON, T> {  // a=rate; torch poisson.py
  static __device__ __forceinline__ T lp(T v, T a, T) {
    return t_xlogy(v, a) - a - t_lgamma(v + T(1));
  }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    dv = t_log(a) - t_digamma(v + T(1));
    da = v / a - T(1);
    db = T(0);
  }
};
template <typename T> struct Fam<PA_DIST_BINOMIAL_LOGITS, T> {  // a=logits b=total_count
  // k*a - n*softplus(a) + log C(n, k)   (pyro/distributions/torch.py:83-101, tol = 0)
  static __device__ __forceinline__ T lp(T v, T a, T n) {
    const T sp = (a > T(0) ? a : T(0)) + t_log1p(t_exp(-t_abs(a)));
    return v * a - n * sp + t_lgamma(n + T(1)) - t_lgamma(v + T(1)) - t_lgamma(n - v + T(1));
  }
  static __device__ __forceinline__ void grad(T v, T a, T n, T& dv, T& da, T& db) {
    const T e = t_exp(-t_abs(a));
    const T sig = a >= T(0) ? T(1) / (T(1) + e) : e / (T(1) + e);
    const T sp = (a > T(0) ? a : T(0)) + t_log1p(e);
    const T pnk = t_digamma(n - v + T(1));
    da = v - n * sig;
    dv = a - t_digamma(v + T(1)) + pnk;
    db = t_digamma(n + T(1)) - pnk - sp;
  }
};

// -KL(Normal(lq, sq) || Normal(lp, sp)) = KL_NORMAL_LOC(lq; lp, sp) + KL_NORMAL_SCALE(sq; sp)
// (torch kl.py _kl_normal_normal: 0.5 * (sq^2/sp^2 + (lq - lp)^2/sp^2 - 1 - log(sq^2/sp^2)))
template <typename T> struct Fam<PA_DIST_KL_NORMAL_LOC, T> {   // v=lq a=lp b=sp
  static __device__ __forceinline__ T lp(T v, T a, T b) {
    const T d = (v - a) * pos_rcp(b);
    return T(-0.5) * d * d - pos_log(b);
  }
  static __device__ __forceinline__ void grad(T v, T a, T b, T& dv, T& da, T& db) {
    Fam<PA_DIST_NORMAL, T>::grad(v, a, b, dv, da, db);
  }
};
template <typename T> struct Fam<PA_DIST_KL_NORMAL_SCALE, T> {   // v=sq a=sp
  static __device__ __forceinline__ T lp(T v, T a, T) {
    const T q = v * pos_rcp(a);
    return pos_log(v) + T(0.5) - T(0.5) * q * q;
  }
  static __device__ __forceinline__ void grad(T v, T a, T, T& dv, T& da, T& db) {
    const T ia = pos_rcp(a), q = v * ia;
    dv = pos_rcp(v) - q * ia;
    da = q * q * ia;
    db = T(0);
  }
};

template <int DIST> struct NParams { static constexpr int n = 1; };
template <> struct NParams<PA_DIST_NORMAL> { static constexpr int n = 2; };
template <> struct NParams<PA_DIST_LOG_NORMAL> { static constexpr int n = 2; };
template <> struct NParams<PA_DIST_GAMMA> { static constexpr int n = 2; };
template <> struct NParams<PA_DIST_BETA> { static constexpr int n = 2; };
template <> struct NParams<PA_DIST_BINOMIAL_LOGITS> { static constexpr int n = 2; };
template <> struct NParams<PA_DIST_KL_NORMAL_LOC> { static constexpr int n = 2; };
// host-side twin of NParams<>
__host__ __device__ static inline int dist_nparams(int dist) {
  return (dist == PA_DIST_NORMAL || dist == PA_DIST_LOG_NORMAL || dist == PA_DIST_GAMMA ||
          dist == PA_DIST_BETA || dist == PA_DIST_BINOMIAL_LOGITS ||
          dist == PA_DIST_KL_NORMAL_LOC) ? 2 : 1;
}


#define PA_DISPATCH_DIST(DIST_ID, T, CALL)                                              \
  switch (DIST_ID) {                                                                    \
    case PA_DIST_NORMAL: { constexpr int D_ = PA_DIST_NORMAL; CALL; } break;              \
    case PA_DIST_BERNOULLI_LOGITS: { constexpr int D_ = PA_DIST_BERNOULLI_LOGITS; CALL; } break; \
    case PA_DIST_HALF_CAUCHY: { constexpr int D_ = PA_DIST_HALF_CAUCHY; CALL; } break;    \
    case PA_DIST_LOG_NORMAL: { constexpr int D_ = PA_DIST_LOG_NORMAL; CALL; } break;      \
    case PA_DIST_EXPONENTIAL: { constexpr int D_ = PA_DIST_EXPONENTIAL; CALL; } break;    \
    case PA_DIST_HALF_NORMAL: { constexpr int D_ = PA_DIST_HALF_NORMAL; CALL; } break;    \
    case PA_DIST_GAMMA: { constexpr int D_ = PA_DIST_GAMMA; CALL; } break;                \
    case PA_DIST_BETA: { constexpr int D_ = PA_DIST_BETA; CALL; } break;                  \
    case PA_DIST_POISSON: { constexpr int D_ = PA_DIST_POISSON; CALL; } break;            \
    case PA_DIST_BINOMIAL_LOGITS: { constexpr int D_ = PA_DIST_BINOMIAL_LOGITS; CALL; } break; \
    case PA_DIST_KL_NORMAL_LOC: { constexpr int D_ = PA_DIST_KL_NORMAL_LOC; CALL; } break; \
    case PA_DIST_KL_NORMAL_SCALE: { constexpr int D_ = PA_DIST_KL_NORMAL_SCALE; CALL; } break; \
    default: return fail(PA_ERR_UNSUPPORTED, "distribution id %d not implemented", DIST_ID); \
  }


}  // namespace pa
