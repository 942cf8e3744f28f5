// device_math.hpp -- small device helpers shared by the assembly kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/mrhyde_amd.h"
#include "device_types.hpp"

namespace mha {

// Inverse and determinant of the cell Jacobian (CellTools::setJacobianInv/Det,
// reference: src/interfaces/discretizationInterface.cpp:923-929).
template <int DIM>
__device__ __forceinline__ void invert(const double *J, double *Ji, double &det);

template <>
__device__ __forceinline__ void invert<2>(const double *J, double *Ji, double &det) {
  det = J[0] * J[3] - J[1] * J[2];
  const double r = 1.0 / det;
  Ji[0] = J[3] * r; Ji[1] = -J[1] * r; Ji[2] = -J[2] * r; Ji[3] = J[0] * r;
}

template <>
__device__ __forceinline__ void invert<3>(const double *J, double *Ji, double &det) {
  const double c0 = J[4] * J[8] - J[5] * J[7], c1 = J[5] * J[6] - J[3] * J[8], c2 = J[3] * J[7] - J[4] * J[6];
  det = J[0] * c0 + J[1] * c1 + J[2] * c2;
  const double r = 1.0 / det;
  Ji[0] = c0 * r; Ji[1] = (J[2] * J[7] - J[1] * J[8]) * r; Ji[2] = (J[1] * J[5] - J[2] * J[4]) * r;
  Ji[3] = c1 * r; Ji[4] = (J[0] * J[8] - J[2] * J[6]) * r; Ji[5] = (J[2] * J[3] - J[0] * J[5]) * r;
  Ji[6] = c2 * r; Ji[7] = (J[1] * J[6] - J[0] * J[7]) * r; Ji[8] = (J[0] * J[4] - J[1] * J[3]) * r;
}

// sin(x) for moderate |x|: Cody-Waite reduction by pi/2 and the fdlibm kernel polynomials (< 1 ulp);
// falls back to the library routine for huge arguments.  About a quarter of the instructions of the
// full-range sin(), which matters because the source term is evaluated at every integration point.
// sin_reduced: the reduction + polynomials alone, valid for |x| < 1e5 -- for kernels whose launcher has checked the bound
// (no branch, no call: the full-range routine inlined at every integration point made those kernels several times larger).
__device__ __forceinline__ double sin_reduced(double x);
__device__ __forceinline__ double sin_moderate(double x) {
  if (!(fabs(x) < 1.0e5)) return sin(x);
  return sin_reduced(x);
}
__device__ __forceinline__ double sin_reduced(double x) {
  const double kd = rint(x * 6.36619772367581382433e-01);
  const int k = (int)kd;
  double r = fma(-kd, 1.57079632673412561417e+00, x);
  r = fma(-kd, 6.07710050650619224932e-11, r);
  r = fma(-kd, 2.02226624879595063154e-21, r);
  const double z = r * r;
  const double sp = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                             2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(r * z, sp, r);
  const double cp = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                             -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double cr = fma(z * z, cp, fma(-0.5, z, 1.0));
  const double v = (k & 1) ? cr : sr;
  return (k & 2) ? -v : v;
}

// MHA_FUNC_EXPRESSION: postfix program over x y z t nx ny nz h pi (expression.hpp); normals / h are 0 when the caller
// has none.  The reference evaluates one kernel per node of the expression DAG (functionManager.cpp:543-860).
// Inlined: as a call it made every deck-string kernel save its live registers around each evaluation (1.0-1.3 KB of
// scratch per lane, 200+ spilled registers in the row-owner and engine kernels; inlined 0.15-0.55 KB).
template <int DIM>
__device__ __forceinline__ double eval_expression(const FuncDesc &f, const double *x, const double *nrm, double h) {
  double st[kExprStack];
  int sp = 0;
  for (const int32_t *pc = f.code;; ++pc) {
    const int op = *pc;
    if (op == EXPR_END) break;
    if (op == EXPR_CONST) { st[sp++] = f.consts[*++pc]; continue; }
    if (op <= EXPR_PI) {
      double v = 0.0;
      switch (op) {
        case EXPR_X: v = x[0]; break;
        case EXPR_Y: v = x[1]; break;
        case EXPR_Z: v = DIM > 2 ? x[DIM - 1] : 0.0; break;
        case EXPR_T: v = f.t; break;
        case EXPR_NX: v = nrm ? nrm[0] : 0.0; break;
        case EXPR_NY: v = nrm ? nrm[1] : 0.0; break;
        case EXPR_NZ: v = (nrm && DIM > 2) ? nrm[DIM - 1] : 0.0; break;
        case EXPR_H: v = h; break;
        default: v = 3.141592653589793238; break;
      }
      st[sp++] = v;
      continue;
    }
    if (op <= EXPR_GE) {
      const double b = st[--sp], a = st[sp - 1];
      double v;
      switch (op) {
        case EXPR_ADD: v = a + b; break;
        case EXPR_SUB: v = a - b; break;
        case EXPR_MUL: v = a * b; break;
        case EXPR_DIV: v = a / b; break;
        case EXPR_POW: v = pow(a, b); break;
        case EXPR_LT: v = a < b ? 1.0 : 0.0; break;
        case EXPR_GT: v = a > b ? 1.0 : 0.0; break;
        case EXPR_LE: v = a <= b ? 1.0 : 0.0; break;
        default: v = a >= b ? 1.0 : 0.0; break;
      }
      st[sp - 1] = v;
      continue;
    }
    const double a = st[sp - 1];
    double v;
    switch (op) {
      case EXPR_NEG: v = -a; break;
      case EXPR_SIN: v = sin(a); break;
      case EXPR_COS: v = cos(a); break;
      case EXPR_TAN: v = tan(a); break;
      case EXPR_EXP: v = exp(a); break;
      case EXPR_LOG: v = log(a); break;
      case EXPR_ABS: v = fabs(a); break;
      case EXPR_SQRT: v = sqrt(a); break;
      case EXPR_SINH: v = sinh(a); break;
      default: v = cosh(a); break;
    }
    st[sp - 1] = v;
  }
  return st[0];
}

}  // namespace mha
#include "dual.hpp"
namespace mha {

// The same program on Dual numbers, with the solution fields of the point as operands (EXPR_FIELD: U[slot],
// EXPR_FIELD_T: Ud[slot]): value and ONE directional derivative -- the chain rule through a coefficient that depends on
// the solution, which Sacado gives the reference for free (FunctionManager<AD>::evaluate, functionManager.cpp:543-860).
// Comparisons and abs differentiate as the reference's AD does (piecewise).
template <int DIM>
__device__ __forceinline__ Dual eval_expression_dual(const FuncDesc &f, const double *x, const double *nrm, double h,
                                                     const Dual *U, const Dual *Ud) {
  Dual st[kExprStack];
  int sp = 0;
  for (const int32_t *pc = f.code;; ++pc) {
    const int op = *pc;
    if (op == EXPR_END) break;
    if (op == EXPR_CONST) { st[sp++] = mk(f.consts[*++pc]); continue; }
    if (op == EXPR_FIELD) { st[sp++] = U[*++pc]; continue; }
    if (op == EXPR_FIELD_T) { st[sp++] = Ud[*++pc]; continue; }
    if (op <= EXPR_PI) {
      double v = 0.0;
      switch (op) {
        case EXPR_X: v = x[0]; break;
        case EXPR_Y: v = x[1]; break;
        case EXPR_Z: v = DIM > 2 ? x[DIM - 1] : 0.0; break;
        case EXPR_T: v = f.t; break;
        case EXPR_NX: v = nrm ? nrm[0] : 0.0; break;
        case EXPR_NY: v = nrm ? nrm[1] : 0.0; break;
        case EXPR_NZ: v = (nrm && DIM > 2) ? nrm[DIM - 1] : 0.0; break;
        case EXPR_H: v = h; break;
        default: v = 3.141592653589793238; break;
      }
      st[sp++] = mk(v);
      continue;
    }
    if (op <= EXPR_GE) {
      const Dual b = st[--sp], a = st[sp - 1];
      Dual v;
      switch (op) {
        case EXPR_ADD: v = a + b; break;
        case EXPR_SUB: v = a - b; break;
        case EXPR_MUL: v = a * b; break;
        case EXPR_DIV: v = a / b; break;
        case EXPR_POW: {
          // a^b: d = a^b (b' log a + b a'/a); a constant exponent (the usual case) needs no logarithm
          const double p = pow(a.v, b.v);
          double d = b.v * pow(a.v, b.v - 1.0) * a.d;
          if (b.d != 0.0) d += p * log(a.v) * b.d;
          v = mk(p, d);
          break;
        }
        case EXPR_LT: v = mk(a.v < b.v ? 1.0 : 0.0); break;
        case EXPR_GT: v = mk(a.v > b.v ? 1.0 : 0.0); break;
        case EXPR_LE: v = mk(a.v <= b.v ? 1.0 : 0.0); break;
        default: v = mk(a.v >= b.v ? 1.0 : 0.0); break;
      }
      st[sp - 1] = v;
      continue;
    }
    const Dual a = st[sp - 1];
    Dual v;
    switch (op) {
      case EXPR_NEG: v = -a; break;
      case EXPR_SIN: v = mk(sin(a.v), cos(a.v) * a.d); break;
      case EXPR_COS: v = mk(cos(a.v), -sin(a.v) * a.d); break;
      case EXPR_TAN: { const double t = tan(a.v); v = mk(t, (1.0 + t * t) * a.d); break; }
      case EXPR_EXP: { const double e = exp(a.v); v = mk(e, e * a.d); break; }
      case EXPR_LOG: v = mk(log(a.v), a.d / a.v); break;
      case EXPR_ABS: v = a.v < 0.0 ? -a : a; break;
      case EXPR_SQRT: v = dsqrt(a); break;
      case EXPR_SINH: v = mk(sinh(a.v), cosh(a.v) * a.d); break;
      default: v = mk(cosh(a.v), sinh(a.v) * a.d); break;
    }
    st[sp - 1] = v;
  }
  return st[0];
}

// Value of a named function at integration point (e,q) with physical coordinates x (nrm: unit normal on sides).
// EXPR: whether MHA_FUNC_EXPRESSION can occur.  The interpreter is inlined (a private 96-byte stack per evaluation); a
// kernel that merely contains it pays its register budget and scratch (the affine element kernel went from 8 to 2 waves
// per SIMD), so kernels are instantiated both ways and the launcher picks by has_expression().
template <int DIM, bool EXPR = false, bool SMALL_ARGS = false>
__device__ __forceinline__ double eval_func(const FuncDesc &f, int e, int q, int nq, const double *x,
                                            const double *nrm = nullptr, double h = 0.0) {
  if (f.kind == MHA_FUNC_CONSTANT) return f.amp;
  if (f.kind == MHA_FUNC_IP_ARRAY) return f.ip[(size_t)e * nq + q];
  if constexpr (EXPR) {
    if (f.kind == MHA_FUNC_EXPRESSION) return eval_expression<DIM>(f, x, nrm, h);
  }
  double s = f.amp;
#pragma unroll
  for (int d = 0; d < DIM; ++d) s *= SMALL_ARGS ? sin_reduced(f.freq[d] * x[d]) : sin_moderate(f.freq[d] * x[d]);
  return s;
}

inline bool has_expression(const FuncDesc &f) { return f.kind == MHA_FUNC_EXPRESSION; }
inline bool uses_fields(const FuncDesc &f) { return f.kind == MHA_FUNC_EXPRESSION && f.uses_fields != 0; }
inline bool has_expression(const ThermalDev &ph) {
  return has_expression(ph.source) || has_expression(ph.diff) || has_expression(ph.cp) || has_expression(ph.rho);
}
inline bool uses_fields(const PhysParamsDev &pp) {
  for (const FuncDesc &f : pp.f)
    if (uses_fields(f)) return true;
  return false;
}
inline bool has_expression(const PhysParamsDev &pp) {
  for (const FuncDesc &f : pp.f)
    if (has_expression(f)) return true;
  return false;
}

// reference-space gradient and value of sum_j c[j] N_j at integration point q (tensor basis)
template <int DIM, int P, int NQ1>
__device__ __forceinline__ void eval_ref(const double *c, const double *phi, const double *dphi, int q,
                                         double *grad, double &val) {
  constexpr int M = P + 1;
  const int q0 = q % NQ1, q1 = (q / NQ1) % NQ1, q2 = q / (NQ1 * NQ1);
  if constexpr (DIM == 2) {
    double g0 = 0, g1 = 0, v = 0;
#pragma unroll
    for (int b1 = 0; b1 < M; ++b1) {
      double s = 0, sd = 0;
#pragma unroll
      for (int a = 0; a < M; ++a) {
        const double u = c[b1 * M + a];
        s += u * phi[a * NQ1 + q0];
        sd += u * dphi[a * NQ1 + q0];
      }
      g0 += sd * phi[b1 * NQ1 + q1];
      g1 += s * dphi[b1 * NQ1 + q1];
      v += s * phi[b1 * NQ1 + q1];
    }
    grad[0] = g0; grad[1] = g1; val = v;
    (void)q2;
  } else {
    double g0 = 0, g1 = 0, g2 = 0, v = 0;
#pragma unroll
    for (int c2 = 0; c2 < M; ++c2) {
      double t = 0, tx = 0, ty = 0;
#pragma unroll
      for (int b1 = 0; b1 < M; ++b1) {
        double s = 0, sd = 0;
#pragma unroll
        for (int a = 0; a < M; ++a) {
          const double u = c[(c2 * M + b1) * M + a];
          s += u * phi[a * NQ1 + q0];
          sd += u * dphi[a * NQ1 + q0];
        }
        t += s * phi[b1 * NQ1 + q1];
        tx += sd * phi[b1 * NQ1 + q1];
        ty += s * dphi[b1 * NQ1 + q1];
      }
      g0 += tx * phi[c2 * NQ1 + q2];
      g1 += ty * phi[c2 * NQ1 + q2];
      g2 += t * dphi[c2 * NQ1 + q2];
      v += t * phi[c2 * NQ1 + q2];
    }
    grad[0] = g0; grad[1] = g1; grad[DIM - 1] = g2; val = v;
  }
}

// Position of column `col` in CRS row [lo,hi) (ascending colind), or -1.
// Plays the role of the column search inside KokkosSparse sumIntoValues
// (reference call site: src/managers/assemblyManager.cpp:4138).
__device__ __forceinline__ int find_col(const int32_t *colind, int lo, int hi, int col) {
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);  // lo + hi overflows int once nnz > 2^30
    const int c = colind[mid];
    if (c == col) return mid;
    if (c < col) lo = mid + 1; else hi = mid;
  }
  return -1;
}

}  // namespace mha
