// newton.cpp -- see newton.hpp.
#include "newton.hpp"

#include <cmath>
#include <cstring>

#include "kernels/launch.hpp"

namespace mha {

NewtonDriver::NewtonDriver(AssemblyManager &mgr, const NewtonSettings &s) : mgr_(mgr), s_(s) {
  MHA_REQUIRE(s.max_iter >= 1 && s.nl_tol >= 0.0 && s.nl_abs_tol >= 0.0, MHA_ERR_INVALID, "bad nonlinear solver settings");
  nbits_.resize(1);
}

void NewtonDriver::reset() {
  iter_ = 0;
  status_ = 0;
  resnorm_ = 0.0;
  scaled_ = 1.0;
  first_ = 0.0;
  alpha_ = 1.0;
  have_du_ = false;
}

void NewtonDriver::residual(const double *u, const double *u_prev, const double *u_stage, double *res) {
  MHA_REQUIRE(u && res, MHA_ERR_INVALID, "null argument");
  MHA_HIP(hipMemsetAsync(res, 0, sizeof(double) * static_cast<size_t>(mgr_.numRows()), mgr_.stream()));
  mgr_.assembleJacRes(0, MHA_PATH_AUTO, u, u_prev, u_stage, res, nullptr);  // assembleRes: the residual-only path
  if (mgr_.numBoundaryGroups() > 0) mgr_.assembleBoundary(0, u, u_prev, u_stage, res, nullptr);
}

double NewtonDriver::norm(const double *res) {
  launch_norm_inf(mgr_.numRows(), res, nbits_.data(), mgr_.stream());
  unsigned long long bits = 0;
  MHA_HIP(hipMemcpyAsync(&bits, nbits_.data(), sizeof bits, hipMemcpyDeviceToHost, mgr_.stream()));
  MHA_HIP(hipStreamSynchronize(mgr_.stream()));
  double v;
  std::memcpy(&v, &bits, sizeof v);
  return v;
}

int NewtonDriver::decide(double resnorm, double *u) {
  MHA_REQUIRE(u != nullptr, MHA_ERR_INVALID, "null argument");
  resnorm_ = resnorm;
  if (iter_ == 0) {
    first_ = resnorm;
    scaled_ = 1.0;
  } else {
    scaled_ = resnorm / first_;
  }
  bool solve = true, proceed = true;
  int action = NEWTON_SOLVE;
  if (s_.allow_backtracking && scaled_ > 1.1 && have_du_) {
    solve = false;
    alpha_ *= 0.5;
    launch_axpy(mgr_.numRows(), -alpha_, du_.data(), u, mgr_.stream());  // sol -= alpha du (solverManager.cpp:1606-1611)
    action = NEWTON_BACKTRACKED;
  } else {
    if (s_.use_relative) {
      if (scaled_ < s_.nl_tol || resnorm < 1.0e-100) { solve = false; proceed = false; }
    } else if (s_.use_absolute && resnorm < s_.nl_abs_tol) {
      solve = false;
      proceed = false;
    }
  }
  if (!(resnorm == resnorm)) {  // a NaN residual: nothing to solve for
    status_ = 1;
    return NEWTON_DONE;
  }
  if (!solve) {
    ++iter_;  // the loop's NLiter++ (update() does it on the solve branch)
    if (!proceed) return NEWTON_DONE;
    if (iter_ >= s_.max_iter) { status_ = 1; return NEWTON_DONE; }
    return action;
  }
  return NEWTON_SOLVE;
}

void NewtonDriver::jacobian(const double *u, const double *u_prev, const double *u_stage, double *res, double *crs_vals) {
  MHA_REQUIRE(u && res && crs_vals, MHA_ERR_INVALID, "null argument");
  MHA_HIP(hipMemsetAsync(res, 0, sizeof(double) * static_cast<size_t>(mgr_.numRows()), mgr_.stream()));
  // (the volume assembly overwrites the CRS values: the caller's setAllToScalar(0) is fused, solverManager.cpp:1528-1533)
  mgr_.assembleJacRes(MHA_ASSEMBLE_JACOBIAN | MHA_ASSEMBLE_OVERWRITE, MHA_PATH_AUTO, u, u_prev, u_stage, res, crs_vals);
  if (mgr_.numBoundaryGroups() > 0) mgr_.assembleBoundary(MHA_ASSEMBLE_JACOBIAN, u, u_prev, u_stage, res, crs_vals);
  mgr_.applyDbcDiag(crs_vals);  // dofConstraints (assemblyManager.cpp:1166-1179)
}

void NewtonDriver::update(double *u, const double *du) {
  MHA_REQUIRE(u && du, MHA_ERR_INVALID, "null argument");
  const size_t n = static_cast<size_t>(mgr_.numRows());
  if (du_.size() != n) du_.resize(n);
  MHA_HIP(hipMemcpyAsync(du_.data(), du, sizeof(double) * n, hipMemcpyDeviceToDevice, mgr_.stream()));
  have_du_ = true;
  alpha_ = 1.0;
  launch_axpy(mgr_.numRows(), alpha_, du, u, mgr_.stream());
  ++iter_;
  if (iter_ >= s_.max_iter) status_ = 1;  // (the caller's next step() reports DONE)
}

int NewtonDriver::step(double *u, const double *u_prev, const double *u_stage, double *res, double *crs_vals) {
  if (iter_ >= s_.max_iter) return NEWTON_DONE;
  if (s_.autotune) {
    residual(u, u_prev, u_stage, res);
  } else {
    jacobian(u, u_prev, u_stage, res, crs_vals);
  }
  const int action = decide(norm(res), u);
  if (action == NEWTON_SOLVE && s_.autotune) jacobian(u, u_prev, u_stage, res, crs_vals);
  return action;
}

}  // namespace mha
