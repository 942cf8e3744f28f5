// physics_points.hpp -- the physics modules as POINT functions.
//
// A module's volumeResidual in the reference is a loop nest `res(e, off(dof)) += sum_k F_k(e,pt) * T_k(e,dof,pt)`
// where T_k runs over the test function's value / gradient components / divergence and F_k are expressions in the
// solution fields at the point.  Here each module states only F(U, Udot, x): the engine (point_engine.hip) supplies the
// fields, differentiates F with Dual numbers direction by direction and contracts with the basis tables.
// Slot numbering: variables in the module's myvars order; HGRAD -> [value, d/dx, d/dy(, d/dz)], HVOL -> [value],
// HDIV -> [v_x, v_y(, v_z), div].  The integration weight is applied by the engine.
#pragma once
#include "device_math.hpp"
#include "dual.hpp"

namespace mha {

template <int DIM>
struct PointArgs {
  const Dual *U;    // fields at the point, slot order
  const Dual *Ud;   // time derivatives (value-like slots only)
  const double *x;  // physical coordinates
  double h, dt;     // element size (sum of wts)^(1/dim), time step
  int transient;
  int e, q, nq;
  const PhysParamsDev *pp;
};

// a function of the module at the point as a Dual: field-dependent deck strings carry the derivative with respect to the
// direction the point's fields are seeded with, everything else is a constant of that direction
template <int DIM>
__device__ __forceinline__ Dual func_dual(const FuncDesc &f, const PointArgs<DIM> &a) {
  if (f.kind == MHA_FUNC_EXPRESSION) return eval_expression_dual<DIM>(f, a.x, nullptr, a.h, a.U, a.Ud);
  return mk(eval_func<DIM, false>(f, a.e, a.q, a.nq, a.x));
}

// thermal (reference: src/physics/thermal.cpp:71-165); functions {source, diffusion, specific heat, density}
// EXPR: 0 constants / closed forms / arrays, 1 deck strings in the coordinates, 2 deck strings that read the solution
// fields ("1+e*e": a nonlinear diffusion; the reference's FunctionManager<AD> differentiates them with Sacado)
template <int DIM, int EXPR>
__device__ __forceinline__ void thermal_point(const PointArgs<DIM> &a, Dual *F) {
  const PhysParamsDev &pp = *a.pp;
  if constexpr (EXPR == 2) {
    const Dual f = func_dual<DIM>(pp.f[0], a), kap = func_dual<DIM>(pp.f[1], a);
    const Dual cp = func_dual<DIM>(pp.f[2], a), rho = func_dual<DIM>(pp.f[3], a);
    F[0] = a.Ud[0] * (rho * cp) - f;
#pragma unroll
    for (int d = 0; d < DIM; ++d) F[1 + d] = a.U[1 + d] * kap;
    return;
  } else {
  constexpr bool EX = EXPR != 0;
  const double f = eval_func<DIM, EX>(pp.f[0], a.e, a.q, a.nq, a.x), kap = eval_func<DIM, EX>(pp.f[1], a.e, a.q, a.nq, a.x);
  const double cp = eval_func<DIM, EX>(pp.f[2], a.e, a.q, a.nq, a.x), rho = eval_func<DIM, EX>(pp.f[3], a.e, a.q, a.nq, a.x);
  F[0] = a.Ud[0] * (rho * cp) - f;
#pragma unroll
  for (int d = 0; d < DIM; ++d) F[1 + d] = a.U[1 + d] * kap;
  // have_advection: (b . grad e) against the value of the test function (thermal.cpp:150-160).  Not in the deck-string
  // instantiation: with the interpreter inlined the kernel is at the scratch it may use (the host refuses that mix)
  if constexpr (!EX)
  if (pp.p[0] != 0.0) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) F[0] = F[0] + a.U[1 + d] * eval_func<DIM, false>(pp.f[4 + d], a.e, a.q, a.nq, a.x);  // (no deck strings: host checks)
  }
  }
}

// porousMixed (reference: src/physics/porousMixed.cpp:158-338); myvars {p (HVOL), u (HDIV)};
// functions {source, Kinv_xx, Kinv_yy, Kinv_zz, total_mobility}
template <int DIM, bool EXPR>
__device__ __forceinline__ void porous_point(const PointArgs<DIM> &a, Dual *F) {
  const PhysParamsDev &pp = *a.pp;
  const double src = eval_func<DIM, EXPR>(pp.f[0], a.e, a.q, a.nq, a.x), mob = eval_func<DIM, EXPR>(pp.f[4], a.e, a.q, a.nq, a.x);
  const Dual p = a.U[0], divu = a.U[1 + DIM];
  F[0] = mk(src) - divu;  // (source - div u, q)
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const double Kinv = eval_func<DIM, EXPR>(pp.f[1 + d], a.e, a.q, a.nq, a.x);
    F[1 + d] = a.U[1 + d] * Kinv / mob;  // ((mobility K)^-1 u, v)
  }
  F[1 + DIM] = -p;  // -(p, div v)
}

// navierstokes (reference: src/physics/navierstokes.cpp:82-849, computeTau :1054-1079); myvars {ux, pr, uy[, uz]};
// functions {source ux, source pr, source uy, source uz, density, viscosity}; p = {useSUPG, usePSPG, fix_uz_offsets}
template <int DIM, bool EXPR>
__device__ __forceinline__ void navierstokes_point(const PointArgs<DIM> &a, Dual *F) {
  constexpr int S = 1 + DIM;                 // slots per HGRAD variable
  constexpr int vnum[3] = {0, 2, 3}, prnum = 1;
  const PhysParamsDev &pp = *a.pp;
  const bool useSUPG = pp.p[0] != 0.0, usePSPG = pp.p[1] != 0.0, fix_uz = pp.p[2] != 0.0;
  const double dens = eval_func<DIM, EXPR>(pp.f[4], a.e, a.q, a.nq, a.x), visc = eval_func<DIM, EXPR>(pp.f[5], a.e, a.q, a.nq, a.x);
  const double src[3] = {eval_func<DIM, EXPR>(pp.f[0], a.e, a.q, a.nq, a.x), eval_func<DIM, EXPR>(pp.f[2], a.e, a.q, a.nq, a.x),
                         DIM == 3 ? eval_func<DIM, EXPR>(pp.f[3], a.e, a.q, a.nq, a.x) : 0.0};
  Dual vel[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) vel[d] = a.U[vnum[d] * S];
  const Dual pr = a.U[prnum * S];
  Dual tau = mk(0.0);
  if (useSUPG || usePSPG) {
    const double C1 = 4.0, C2 = 2.0, C3 = a.transient ? 2.0 : 0.0;
    Dual nvel = mk(0.0);
#pragma unroll
    for (int d = 0; d < DIM; ++d) nvel += vel[d] * vel[d];
    if (nvel.v > 1e-12) nvel = dsqrt(nvel);
    const Dual t2 = nvel * (C2 / a.h);
    const double c1 = C1 * visc / a.h / a.h, c3 = C3 / a.dt;
    tau = 1.0 / dsqrt(t2 * t2 + (c1 * c1 + c3 * c3));
  }
  Dual stab[DIM];
  Dual divu = mk(0.0);
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    const int b = vnum[i] * S;
    Dual conv = mk(0.0);
#pragma unroll
    for (int d = 0; d < DIM; ++d) conv += vel[d] * a.U[b + 1 + d];
    const Dual acc = a.Ud[b] + conv;
    F[b] = (acc - src[i]) * dens;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      F[b + 1 + d] = a.U[b + 1 + d] * visc;
      if (d == i) F[b + 1 + d] -= pr;
    }
    divu += a.U[b + 1 + i];
    if (useSUPG || usePSPG) stab[i] = acc * dens + a.U[prnum * S + 1 + i] - dens * src[i];
    if (useSUPG) {
      const Dual ts = tau * stab[i];
#pragma unroll
      for (int d = 0; d < DIM; ++d) F[b + 1 + d] += ts * vel[d];
    }
  }
  F[prnum * S] = divu;
#pragma unroll
  for (int d = 0; d < DIM; ++d) F[prnum * S + 1 + d] = usePSPG ? stab[d] * tau / dens : mk(0.0);
  if (DIM == 3 && !fix_uz) {
    // the reference scatters the uz momentum block through uy's offsets (navierstokes.cpp:688): uy's rows receive both
    // blocks (same HGRAD basis), uz's rows stay empty
#pragma unroll
    for (int s = 0; s < S; ++s) {
      F[vnum[1] * S + s] += F[vnum[DIM - 1] * S + s];
      F[vnum[DIM - 1] * S + s] = mk(0.0);
    }
  }
}

// shallowwaterHybridized (reference: src/physics/shallowwaterHybridized.cpp:113-184 with computeFluxVector(false)
// :409-480); myvars {H, Hux, Huy} (2-D); functions {source H, source Hux, source Huy}; p = {g}
template <int DIM, bool EXPR>
__device__ __forceinline__ void swhdg_point(const PointArgs<DIM> &a, Dual *F) {
  static_assert(DIM >= 2, "");
  constexpr int S = 1 + DIM;
  const PhysParamsDev &pp = *a.pp;
  const double g = pp.p[0];
  const Dual H = a.U[0], Hux = a.U[S], Huy = a.U[2 * S];
  const Dual hh = H * H * (0.5 * g);
  Dual Fl[3][2];
  Fl[0][0] = Hux; Fl[0][1] = Huy;
  Fl[1][0] = Hux * Hux / H + hh; Fl[1][1] = Hux * Huy / H;
  Fl[2][0] = Hux * Huy / H; Fl[2][1] = Huy * Huy / H + hh;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    F[i * S] = a.Ud[i * S] - eval_func<DIM, EXPR>(pp.f[i], a.e, a.q, a.nq, a.x);  // (v, dS/dt) - (v, source)
    F[i * S + 1] = -Fl[i][0];                                               // -(dv/dx, F_x)
    F[i * S + 2] = -Fl[i][1];                                               // -(dv/dy, F_y)
  }
}

}  // namespace mha
