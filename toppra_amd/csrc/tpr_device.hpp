// tpr_device.hpp -- device-side building blocks shared by the TOPP-RA kernels (gfx950).
//
// Everything here is fp64 and must be compiled with -ffp-contract=off: the reference is an
// x86-64 build without FMA and the parity target is bit-exactness, so every multiply and add
// is rounded separately and in the reference's order.  fp64 '/' and sqrt() lower to the
// correctly rounded v_div_scale/v_div_fmas/v_div_fixup and v_sqrt+refinement sequences.
//
// Reference map (hungpham2511/toppra v0.6.2):
//   path_eval            toppra/interpolator.py:419-430  (scipy PPoly of cspl.derivative())
//   velocity_xbound      toppra/_CythonUtils.pyx:16-59, constraint/linear_joint_velocity.py:43-53
//   lp1d / lp2d          toppra/solverwrapper/cy_seidel_solverwrapper.pyx:93-144 / :149-390
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpr {

constexpr double kTiny = 1e-10;        // cy_seidel_solverwrapper.pyx:17
constexpr double kSmall = 1e-8;        // :18
constexpr double kVarMin = -1.0e8;     // :22
constexpr double kVarMax = 1.0e8;      // :23
constexpr double kInf1d = 1.0e10;      // :27
constexpr double kPyTiny = 1e-8;       // toppra/constants.py:16
constexpr double kPySmall = 1e-5;      // toppra/constants.py:17
constexpr int kMaxTries = 10;          // toppra/constants.py:24
constexpr double kFeasMaxX = 10000.0;  // toppra/constants.py:42 (CVXPY_MAXX)
constexpr float kJvelMaxSd = 1e8f;     // toppra/_CythonUtils.pyx:14 (a C float there)

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// x = sd^2 of a boundary path velocity.  The reference squares with Python's `**` (reachability_algorithm.py:226,
// :262-266), i.e. libm pow() for Python floats -- NOT correctly rounded: it differs from sd * sd by one ulp for
// about 0.08 % of the values -- and numpy's square (sd * sd) for numpy scalars and arrays.  A caller that has
// squared the velocities itself, in the reference's own expression, says so with TPR_BOUNDARY_SQUARED (the drop-in
// class does); otherwise the device squares them (exact for arrays, which is what the batch entries take).
__device__ __forceinline__ double boundary_x(int flags, double v) { return (flags & TPR_BOUNDARY_SQUARED) ? v : v * v; }

// One trajectory's read-only inputs.
struct Traj {
    const double *coef;    // [4][nseg][d]
    const double *breaks;  // [nseg+1]
    const double *grid;    // [N+1]
    const double *vlim;    // [d][2]
    const double *alim;    // [d][2]
    int d, nseg, N;
    bool has_vel, has_acc, interp;
    __device__ int nC() const { return 2 + (has_acc ? (interp ? 4 * d : 2 * d) : 0); }
};

// Spline segment holding s: breaks[j] <= s < breaks[j+1], clamped to the first/last segment
// (scipy find_interval with extrapolation).
__device__ __forceinline__ int find_segment(const double *breaks, int nseg, double s) {
    int j = 0;
    for (int m = 1; m < nseg; ++m)
        if (s >= breaks[m]) j = m;
    return j;
}

// 1 / d to a few ulp (hardware estimate + one Newton step; 4 instructions where the IEEE sequence takes 11).  Only for
// values that feed searches and margin tests -- a vertex to test residuals at, a candidate bound -- never an output.
__device__ __forceinline__ double rcp_approx(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

// q'(s), q''(s) of one dof from its four cubic coefficients (highest power first) at local
// parameter t.  Power-basis accumulation in scipy's order, not Horner.
__device__ __forceinline__ void cubic_d1_d2(double c0, double c1, double c2, double t, double &q1,
                                            double &q2) {
    const double d0 = c0 * 3.0, d1 = c1 * 2.0, d2 = c2;
    const double e0 = d0 * 2.0, e1 = d1;
    q1 = (d2 + d1 * t) + d0 * (t * t);
    q2 = e1 + e0 * t;
}

// The same from the derivative's coefficients d0 = 3 c0, d1 = 2 c1, d2 = c2 (segment constants: a kernel that keeps
// a segment's coefficients in registers scales them once per segment instead of once per gridpoint).
__device__ __forceinline__ void cubic_d1_d2_scaled(double d0, double d1, double d2, double t, double &q1, double &q2) {
    const double e0 = d0 * 2.0, e1 = d1;
    q1 = (d2 + d1 * t) + d0 * (t * t);
    q2 = e1 + e0 * t;
}

__device__ inline void path_eval(const Traj &T, double s, double *q1, double *q2) {
    const int j = find_segment(T.breaks, T.nseg, s);
    const double t = s - T.breaks[j];
    const double *c0 = T.coef + (size_t)(0 * T.nseg + j) * T.d;
    const double *c1 = T.coef + (size_t)(1 * T.nseg + j) * T.d;
    const double *c2 = T.coef + (size_t)(2 * T.nseg + j) * T.d;
    for (int k = 0; k < T.d; ++k) cubic_d1_d2(c0[k], c1[k], c2[k], t, q1[k], q2[k]);
}

// Velocity limit -> bounds on x = sd^2 at one gridpoint.  sdmin/sdmax are C floats in the
// reference: every min/max result is rounded to fp32, the upper bound is squared in fp32
// (powf), the lower bound in fp64.
__device__ inline void velocity_xbound(const Traj &T, const double *q1, double &xlo, double &xhi) {
    float sdmin = -kJvelMaxSd, sdmax = kJvelMaxSd;
    for (int k = 0; k < T.d; ++k) {
        const double q = q1[k];
        if (q > 0) {
            const double r1 = T.vlim[2 * k + 1] / q, r0 = T.vlim[2 * k] / q;
            sdmax = (float)(r1 <= (double)sdmax ? r1 : (double)sdmax);
            sdmin = (float)(r0 >= (double)sdmin ? r0 : (double)sdmin);
        } else if (q < 0) {
            const double r0 = T.vlim[2 * k] / q, r1 = T.vlim[2 * k + 1] / q;
            sdmax = (float)(r0 <= (double)sdmax ? r0 : (double)sdmax);
            sdmin = (float)(r1 >= (double)sdmin ? r1 : (double)sdmin);
        }
    }
    const float up = sdmax * sdmax;
    const double lo = (double)sdmin >= 0.0 ? (double)sdmin : 0.0;
    xlo = lo * lo;
    xhi = (double)up;
}

struct Lp1dOut {
    bool ok;
    double x;
    int active;
};

// max v0*x  s.t.  a_i x + b_i <= 0, low <= x <= high.  Rows with |a_i| <= 1e-10 are ignored,
// strict comparisons so the first index wins ties.
template <class RowA, class RowB>
__device__ inline Lp1dOut lp1d(double v0, int nrows, RowA a, RowB b, double low, double high) {
    double cur_min = low, cur_max = high;
    int amin = -1, amax = -2;
    for (int i = 0; i < nrows; ++i) {
        const double ai = a(i);
        if (ai > kTiny) {
            const double x = -b(i) / ai;
            if (x < cur_max) { cur_max = x; amax = i; }
        } else if (ai < -kTiny) {
            const double x = -b(i) / ai;
            if (x > cur_min) { cur_min = x; amin = i; }
        }
    }
    Lp1dOut o;
    o.ok = !(cur_min > cur_max);
    const bool pick_min = fabs(v0) < kTiny || v0 < 0;
    o.x = pick_min ? cur_min : cur_max;
    o.active = pick_min ? amin : amax;
    return o;
}

struct Lp2dOut {
    bool ok;
    double u, x;
    int ac0, ac1;
    bool hint_full = false;  // family 3: the vertex looked optimal but could not be certified -- skip the walk, iterate
#ifdef TPR_DEBUG_PREDICT
    int why = 0;  // debug builds: why a certificate failed (tpr_cert.hip.inc)
#endif
};

// Incremental (Seidel) 2-variable LP in the reference's deterministic order.
//   max v0*u + v1*x  s.t.  a_i u + b_i x + c_i <= 0,  low <= (u,x) <= high
// wac0/wac1: warm-start rows (previous active set); order[] is nrows bytes of scratch.
__device__ inline Lp2dOut lp2d(double v0, double v1, int nrows, const double *a, const double *b,
                               const double *c, double low0, double high0, double low1, double high1,
                               int wac0, int wac1, unsigned char *order) {
    Lp2dOut o;
    o.ok = false;
    o.u = o.x = qnan();
    o.ac0 = o.ac1 = 0;
    if (low0 > high0 || low1 > high1) return o;
    double cu, cx;
    if (v0 > kTiny) { cu = high0; o.ac0 = -2; } else { cu = low0; o.ac0 = -1; }
    if (v1 > kTiny) { cx = high1; o.ac1 = -4; } else { cx = low1; o.ac1 = -3; }

    if (wac0 >= 0 && wac0 < nrows && wac1 >= 0 && wac1 < nrows && wac0 != wac1) {
        order[0] = (unsigned char)wac1;
        order[1] = (unsigned char)wac0;
        int w = 2;
        for (int i = 0; i < nrows; ++i)
            if (i != wac0 && i != wac1) order[w++] = (unsigned char)i;
    } else {
        for (int i = 0; i < nrows; ++i) order[i] = (unsigned char)i;
    }

    for (int k = 0; k < nrows; ++k) {
        const int i = order[k];
        const double ai = a[i], bi = b[i], ci = c[i];
        if (ai * cu + bi * cx + ci < kTiny) continue;
        o.ac0 = i;
        const double den = ai * ai + bi * bi;
        if (den == 0.0) return o;  // the reference raises ZeroDivisionError here
        const double zp0 = (-ai * ci) / den;
        const double zp1 = (-bi * ci) / den;
        const double dt0 = -bi, dt1 = ai;
        const double v1d = dt0 * v0 + dt1 * v1;
        double cur_min = -kInf1d, cur_max = kInf1d;
        int amin = -1, amax = -2;
        for (int j = 0; j < k + 4; ++j) {
            double aj, bj, cj;
            if (j < k) { const int r = order[j]; aj = a[r]; bj = b[r]; cj = c[r]; }
            else if (j == k) { aj = -1; bj = 0; cj = low0; }
            else if (j == k + 1) { aj = 1; bj = 0; cj = -high0; }
            else if (j == k + 2) { aj = 0; bj = -1; cj = low1; }
            else { aj = 0; bj = 1; cj = -high1; }
            const double denom = dt0 * aj + dt1 * bj;
            const double num = cj + zp1 * bj + zp0 * aj;
            if (denom > kTiny) {
                const double t = -num / denom;
                if (t < cur_max) { cur_max = t; amax = j; }
            } else if (denom < -kTiny) {
                const double t = -num / denom;
                if (t > cur_min) { cur_min = t; amin = j; }
            } else if (num > kSmall) {
                return o;  // parallel and infeasible
            }
        }
        if (cur_min > cur_max) return o;
        const bool pick_min = fabs(v1d) < kTiny || v1d < 0;
        const double t = pick_min ? cur_min : cur_max;
        const int act = pick_min ? amin : amax;
        cu = zp0 + t * dt0;
        cx = zp1 + t * dt1;
        // active index bookkeeping; the +-1e10 line bounds (act < 0) are "infeasible" in the
        // reference (int vs unsigned comparison at :366-383)
        if (act < 0) return o;
        if (act < k) o.ac1 = order[act];
        else o.ac1 = -1 - (act - k);
    }
    o.ok = true;
    o.u = cu;
    o.x = cx;
    return o;
}

}  // namespace tpr
