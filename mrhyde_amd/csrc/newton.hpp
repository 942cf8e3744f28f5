// newton.hpp -- the nonlinear-solve protocol around the assembler, on the host side of the C ABI.
//
// reference: SolverManager::nonlinearSolver (src/managers/solverManager.cpp:1465-1709).  Per iteration:
//   zero the residual; AUTOTUNE (:1538-1552): first assembleRes (residual only, the ScalarT path); Export(ADD) of the
//   residual (:1556); |res|_inf (:1571); iteration 0 fixes resnorm_first, later resnorm_scaled = resnorm / resnorm_first
//   (:1575-1581); backtracking when allowed and the scaled norm grew beyond 1.1: step halved, sol -= alpha du, no solve
//   (:1591-1616); relative / absolute tolerance tests end the loop (:1617-1632); only if a solve is needed:
//   assembleJacRes (AD) (:1640), matrix export, the caller's linear solve, import, sol += alpha du with alpha = 1
//   (:1652-1672); NLiter++ and the iteration limit (:1681-1685).
// The linear solver, the Export / Import between ranks and the norm's all-reduce are the caller's (SURVEY.md section 2:
// out of scope): the driver is a state machine the caller steps through with plain calls, no callbacks --
//   residual()  -> [caller: Export(ADD) of res]  -> norm()  [-> caller: all-reduce max]  -> decide(resnorm) -> action
//   action SOLVE:       jacobian() -> [caller: Export(ADD) of J, solve J du = res, Import du] -> update(du)
//   action BACKTRACKED: the solution has been moved back half a step; next iteration
//   action DONE:        converged, or the iteration limit
// step() runs residual + norm + decide (+ jacobian) for a single-rank caller.
#pragma once
#include <cstdint>

#include "assembly_manager.hpp"

namespace mha {

enum NewtonAction { NEWTON_SOLVE = 1, NEWTON_BACKTRACKED = 2, NEWTON_DONE = 3 };

struct NewtonSettings {
  int max_iter = 10;            // "max nonlinear iters" (solverManager.cpp:74)
  double nl_tol = 1e-6;         // "nonlinear TOL", relative (:75)
  double nl_abs_tol = 1e-6;     // "absolute nonlinear TOL" (:76)
  int use_relative = 1, use_absolute = 0;  // (:78-79)
  int allow_backtracking = 0;   // (:80)
  int autotune = 1;             // residual-only pass first; 0: assembleJacRes at once (adjoint / multiscale runs, :1540-1543)
};

class NewtonDriver {
 public:
  NewtonDriver(AssemblyManager &mgr, const NewtonSettings &s);
  void reset();
  // -res.val() of the current state into res (zeroed first); boundary groups included
  void residual(const double *u, const double *u_prev, const double *u_stage, double *res);
  double norm(const double *res);  // |res|_inf of this rank's rows (synchronises the stream: the decision is the host's)
  int decide(double resnorm, double *u);
  // Jacobian + residual at the current state (both zeroed first), fixed rows get their unit diagonal (dofConstraints)
  void jacobian(const double *u, const double *u_prev, const double *u_stage, double *res, double *crs_vals);
  void update(double *u, const double *du);  // sol += alpha du (alpha = 1), du kept for backtracking; closes the iteration
  int step(double *u, const double *u_prev, const double *u_stage, double *res, double *crs_vals);
  int iteration() const { return iter_; }
  double resnorm() const { return resnorm_; }
  double resnormScaled() const { return scaled_; }
  double resnormFirst() const { return first_; }
  double alpha() const { return alpha_; }
  int status() const { return status_; }  // 0 running / converged, 1 the iteration limit was reached without convergence

 private:
  AssemblyManager &mgr_;
  NewtonSettings s_;
  int iter_ = 0, status_ = 0;
  double resnorm_ = 0.0, scaled_ = 1.0, first_ = 0.0, alpha_ = 1.0;
  DeviceBuffer<double> du_;                 // the last update (backtracking)
  DeviceBuffer<unsigned long long> nbits_;  // norm scratch
  bool have_du_ = false;
};

}  // namespace mha
