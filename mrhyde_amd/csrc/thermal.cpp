#include "physics.hpp"

namespace mha {

thermal::thermal() {
  label = "thermal";
  myvars = {"e"};            // reference: thermal.cpp:31
  mybasistypes = {"HGRAD"};  // reference: thermal.cpp:32
}

// Defaults of thermal::defineFunctions (reference: src/physics/thermal.cpp:52-63):
// source 0, diffusion 1, specific heat 1, density 1.
void thermal::defineFunctions(FunctionManager &fm) {
  functionManager = &fm;
  auto constant = [](double v) { FuncDesc f; f.kind = MHA_FUNC_CONSTANT; f.amp = v; return f; };
  if (!fm.has("thermal source")) fm.addFunction("thermal source", constant(0.0));
  if (!fm.has("thermal diffusion")) fm.addFunction("thermal diffusion", constant(1.0));
  if (!fm.has("specific heat")) fm.addFunction("specific heat", constant(1.0));
  if (!fm.has("density")) fm.addFunction("density", constant(1.0));
}

void thermal::setParameter(const std::string &name, double value) {
  if (name == "form_param") formparam = value;  // reference: thermal.cpp:35
  else PhysicsBase::setParameter(name, value);
}

ThermalDev thermal::device_params() const {
  ThermalDev p;
  p.source = functionManager->evaluate("thermal source");
  p.diff = functionManager->evaluate("thermal diffusion");
  p.cp = functionManager->evaluate("specific heat");
  p.rho = functionManager->evaluate("density");
  p.time = wkset->time_dev;
  return p;
}

// reference: thermal<EvalT>::volumeResidual (src/physics/thermal.cpp:71-165).  Launches the
// per-element kernel over the workset's element range; results go to wkset->res.
void thermal::volumeResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "thermal::volumeResidual called without a workset");
  Workset &w = *wkset;
  BlockDev b = w.dev;
  b.e_begin = w.first_elem;
  b.e_count = w.numElem;
  if (w.use_general)
    launch_thermal_general(w.dimension, w.order, w.nq1, b, device_params(), w.tables, w.elem_slot, w.elem_slot_bytes,
                           w.res, w.stream);
  else
    launch_thermal_element(w.dimension, w.order, w.nq1, b, device_params(), w.res, w.stream);
}

// reference: thermal<EvalT>::boundaryResidual (src/physics/thermal.cpp:172-281).  The boundary-condition type of
// "e" on the current side selects the branch; the data comes from the function "Neumann e <side>" or
// "Dirichlet e <side>" evaluated at the side integration points (thermal.cpp:217, 238).
void thermal::boundaryResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "thermal::boundaryResidual called without a workset");
  Workset &w = *wkset;
  BoundaryDev bd = w.bnd;
  bd.bc_type = w.current_bc;
  bd.form_param = formparam;
  if (w.current_bc == MHA_BC_NEUMANN) {
    bd.data = functionManager->evaluate("Neumann e " + w.sidename);
  } else if (w.current_bc == MHA_BC_WEAK_DIRICHLET) {
    bd.data = functionManager->evaluate("Dirichlet e " + w.sidename);
    bd.diff = functionManager->evaluate("thermal diffusion");
    MHA_REQUIRE(bd.diff.kind != MHA_FUNC_IP_ARRAY, MHA_ERR_INVALID,
                "weak Dirichlet needs 'thermal diffusion' at the side points: give it as a constant or closed form");
  } else {
    return;  // strong Dirichlet / none: no boundary term (thermal.cpp:188-216 falls through)
  }
  launch_thermal_boundary(w.dev, w.side_tables, bd, w.time_dev, w.res, w.stream);
}

std::unique_ptr<PhysicsBase> import_physics(int physics_id) {
  if (physics_id == MHA_PHYSICS_THERMAL) return std::unique_ptr<PhysicsBase>(new thermal());
  throw Error(MHA_ERR_INVALID, "unknown physics module id " + std::to_string(physics_id));
}

}  // namespace mha
