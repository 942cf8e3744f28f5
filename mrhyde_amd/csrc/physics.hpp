// physics.hpp -- physics-module operator API of the MI355X path.
//
// Mirrors template<class EvalT> class PhysicsBase (reference: src/physics/physicsBase.hpp:30-205):
// modules are created by name, get a FunctionManager-like table of named coefficient functions
// through defineFunctions(), are pointed at a Workset with setWorkset(), and expose
// volumeResidual() / boundaryResidual() / faceResidual() / computeFlux() with no arguments and
// no return value -- all I/O goes through the workset.  There is one instance per block instead of
// one per EvalT: the derivative array (EvalT = SFad<W>) is produced inside the kernel.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "workset.hpp"

namespace mha {

// The slice of FunctionManager<EvalT> the functors use: evaluate(name,"ip")
// (reference: src/managers/functionManager.cpp:543-760).  A function is a constant, per-ip data,
// or a member of a small closed family evaluated at the physical integration points.
class FunctionManager {
 public:
  void addFunction(const std::string &name, const FuncDesc &f) { funcs_[name] = f; programs_.erase(name); texts_.erase(name); dirty_ = true; }
  // a deck string: compiled to a postfix program kept on the device for the life of the function (expression.hpp).
  // Compilation is deferred to the first evaluate(): a string may name solution fields of the block (setFieldSlots)
  // and functions of the deck that are defined later, as FunctionManager::decomposeFunctions allows
  // (functionManager.cpp:95-540).
  void addExpression(const std::string &name, const std::string &text);
  // names of Workset::getSolutionField -> slot of the point engine's field array; `_t` names -> time-derivative slots
  void setFieldSlots(const std::map<std::string, int> &fields, const std::map<std::string, int> &fields_t) {
    field_slot_ = fields;
    field_t_slot_ = fields_t;
    dirty_ = true;
  }
  bool has(const std::string &name) const { return funcs_.count(name) != 0 || texts_.count(name) != 0; }
  void setTime(double t) { time_ = t; }
  FuncDesc evaluate(const std::string &name) const {
    if (dirty_) compileAll();
    auto it = funcs_.find(name);
    // reference: TEUCHOS_TEST_FOR_EXCEPTION "function manager could not evaluate" (functionManager.cpp:573)
    MHA_REQUIRE(it != funcs_.end(), MHA_ERR_INVALID, "function manager could not evaluate: " << name);
    FuncDesc f = it->second;
    f.t = time_;
    return f;
  }

 private:
  struct Program { DeviceBuffer<int32_t> code; DeviceBuffer<double> consts; };
  void compileAll() const;
  void compileOne(const std::string &name, std::vector<std::string> &open, std::vector<int32_t> &code, std::vector<double> &consts,
                  bool &fields) const;
  mutable std::map<std::string, FuncDesc> funcs_;
  mutable std::map<std::string, std::shared_ptr<Program>> programs_;
  std::map<std::string, std::string> texts_;
  std::map<std::string, int> field_slot_, field_t_slot_;
  mutable bool dirty_ = false;
  double time_ = 0.0;
};

class PhysicsBase {
 public:
  virtual ~PhysicsBase() = default;
  virtual void defineFunctions(FunctionManager &fm) { functionManager = &fm; }
  virtual void volumeResidual() { notImplemented("volumeResidual"); }
  virtual void boundaryResidual() { notImplemented("boundaryResidual"); }
  virtual void faceResidual() { notImplemented("faceResidual"); }
  virtual void computeFlux() { notImplemented("computeFlux"); }
  virtual void setWorkset(Workset *w) { wkset = w; }
  // true when the current settings need terms only the point-engine form of the module has
  virtual bool pointEngineOnly() const { return false; }
  // scalar settings a module reads from its parameter list in the reference constructor
  virtual void setParameter(const std::string &name, double) {
    throw Error(MHA_ERR_INVALID, "physics module '" + label + "' has no parameter '" + name + "'");
  }

  std::string label;
  Workset *wkset = nullptr;
  FunctionManager *functionManager = nullptr;
  std::vector<std::string> myvars, mybasistypes;

 protected:
  // the reference prints a message for un-overridden hooks (physicsBase.cpp); we raise
  void notImplemented(const char *what) const {
    throw Error(MHA_ERR_INVALID, "physics module '" + label + "' does not implement " + what);
  }
};

// thermal: rho*cp*de/dt - div(kappa grad e) = source
// (reference: src/physics/thermal.hpp, src/physics/thermal.cpp:24-165)
class thermal : public PhysicsBase {
 public:
  thermal();
  void defineFunctions(FunctionManager &fm) override;
  void volumeResidual() override;
  void boundaryResidual() override;
  void computeFlux() override;
  void setParameter(const std::string &name, double value) override;
  ThermalDev device_params() const;
  double formparam = 1.0;  // settings "form_param" (reference: thermal.cpp:35)
  bool have_advection = false;  // settings "include advection" (reference: thermal.cpp:39): adds (b . grad e, v)
  // the advection term and coefficients that depend on the solution ("1+e*e") exist in the point-engine form only
  bool pointEngineOnly() const override;
};

// porousMixed: mixed Darcy, (K mobility)^-1 u + grad p = 0, div u = source
// (reference: src/physics/porousMixed.hpp, src/physics/porousMixed.cpp:24-432); myvars {p (HVOL), u (HDIV)}
class porousMixed : public PhysicsBase {
 public:
  porousMixed();
  void defineFunctions(FunctionManager &fm) override;
  void volumeResidual() override;
  void boundaryResidual() override;
  void computeFlux() override;
};

// navierstokes: incompressible Navier-Stokes with optional SUPG / PSPG
// (reference: src/physics/navierstokes.hpp, src/physics/navierstokes.cpp:20-849, 1054-1079); myvars {ux, pr, uy[, uz]}
class navierstokes : public PhysicsBase {
 public:
  navierstokes();
  void defineFunctions(FunctionManager &fm) override;
  void volumeResidual() override;
  void computeFlux() override;
  void setParameter(const std::string &name, double value) override;
  bool useSUPG = false, usePSPG = false;  // navierstokes.cpp:45-46
  bool fix_uz_offsets = false;            // false reproduces navierstokes.cpp:688
};

// shallowwaterHybridized: HDG shallow water, interior unknowns H, Hux, Huy; the trace unknowns are the module's "aux"
// variables (reference: src/physics/shallowwaterHybridized.hpp, .cpp:24-844).  volumeResidual runs on the point engine;
// the side terms (computeFlux, boundaryResidual's fluxes) are the stateless batch entry point mha_swhdg_side_terms:
// the workset-level aux-variable plumbing of the subgrid solver that feeds them is not built.
class shallowwaterHybridized : public PhysicsBase {
 public:
  shallowwaterHybridized();
  void defineFunctions(FunctionManager &fm) override;
  void volumeResidual() override;
  void boundaryResidual() override;
  bool roestab = true;    // settings "Roe-like stabilization" / "max EV stabilization" (shallowwaterHybridized.cpp:60-66)
  void setParameter(const std::string &name, double value) override;
  double gravity = 9.81;  // settings "g" (shallowwaterHybridized.cpp:72)
};

// PhysicsImporter::import equivalent (reference: src/physics/physicsImporter.cpp:48-204)
std::unique_ptr<PhysicsBase> import_physics(int physics_id);

}  // namespace mha
