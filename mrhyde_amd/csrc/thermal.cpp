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

std::unique_ptr<PhysicsBase> import_physics(int physics_id) {
  if (physics_id == MHA_PHYSICS_THERMAL) return std::unique_ptr<PhysicsBase>(new thermal());
  throw Error(MHA_ERR_INVALID, "unknown physics module id " + std::to_string(physics_id));
}

}  // namespace mha
