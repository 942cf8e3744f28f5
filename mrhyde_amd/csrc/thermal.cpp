#include <algorithm>

#include "physics.hpp"

#include "expression.hpp"

#include <cstdlib>

namespace mha {

void FunctionManager::addExpression(const std::string &name, const std::string &text) {
  // syntax errors surface here; identifiers the string may name later (fields, other functions) are accepted for now
  std::vector<int32_t> code;
  std::vector<double> consts;
  compile_expression(text, code, consts, [](const std::string &, std::vector<int32_t> &c, std::vector<double> &) {
    c.push_back(EXPR_PI);  // placeholder operand
    return true;
  });
  texts_[name] = text;
  funcs_.erase(name);
  programs_.erase(name);
  dirty_ = true;
}

// one named expression -> program; `open` = the chain of functions being resolved (cycle detection)
void FunctionManager::compileOne(const std::string &name, std::vector<std::string> &open, std::vector<int32_t> &code,
                                 std::vector<double> &consts, bool &fields) const {
  MHA_REQUIRE(std::find(open.begin(), open.end(), name) == open.end(), MHA_ERR_INVALID,
              "function '" << name << "' refers to itself (through " << open.size() << " other functions)");
  open.push_back(name);
  ExprResolver resolver = [&](const std::string &id, std::vector<int32_t> &c, std::vector<double> &k) {
    auto fs = field_slot_.find(id);
    if (fs != field_slot_.end()) { c = {EXPR_FIELD, fs->second, EXPR_END}; return true; }
    auto ft = field_t_slot_.find(id);
    if (ft != field_t_slot_.end()) { c = {EXPR_FIELD_T, ft->second, EXPR_END}; return true; }
    auto tx = texts_.find(id);
    if (tx != texts_.end()) {  // another deck string: inlined
      bool f2 = false;
      compileOne(id, open, c, k, f2);
      fields = fields || f2;
      return true;
    }
    auto fn = funcs_.find(id);
    if (fn != funcs_.end() && fn->second.kind == MHA_FUNC_CONSTANT) {  // a constant function: its value
      c = {EXPR_CONST, 0, EXPR_END};
      k = {fn->second.amp};
      return true;
    }
    return false;
  };
  bool f1 = false;
  compile_expression(texts_.at(name), code, consts, resolver, &f1);
  fields = fields || f1;
  open.pop_back();
}

void FunctionManager::compileAll() const {
  for (const auto &kv : texts_) {
    std::vector<int32_t> code;
    std::vector<double> consts;
    std::vector<std::string> open;
    bool fields = false;
    compileOne(kv.first, open, code, consts, fields);
    auto prog = std::make_shared<Program>();
    prog->code.upload(code);
    if (consts.empty()) consts.push_back(0.0);
    prog->consts.upload(consts);
    FuncDesc f;
    f.kind = MHA_FUNC_EXPRESSION;
    f.code = prog->code.data();
    f.consts = prog->consts.data();
    f.uses_fields = fields ? 1 : 0;
    funcs_[kv.first] = f;
    programs_[kv.first] = prog;
  }
  dirty_ = false;
}

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
  for (const char *k : {"bx", "by", "bz"})  // "advection x|y|z", default 0 (thermal.cpp:59-61)
    if (!fm.has(k)) fm.addFunction(k, constant(0.0));
}

void thermal::setParameter(const std::string &name, double value) {
  if (name == "form_param") formparam = value;  // reference: thermal.cpp:35
  else if (name == "include advection") have_advection = value != 0.0;  // reference: thermal.cpp:39
  else PhysicsBase::setParameter(name, value);
}

bool thermal::pointEngineOnly() const {
  if (have_advection) return true;
  if (!functionManager) return false;
  for (const char *k : {"thermal source", "thermal diffusion", "specific heat", "density"})
    if (functionManager->has(k) && functionManager->evaluate(k).uses_fields) return true;
  return false;
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
  if (w.use_point_engine) {
    PhysParamsDev pp;
    pp.physics = MHA_PHYSICS_THERMAL;
    pp.f[0] = functionManager->evaluate("thermal source");
    pp.f[1] = functionManager->evaluate("thermal diffusion");
    pp.f[2] = functionManager->evaluate("specific heat");
    pp.f[3] = functionManager->evaluate("density");
    // (b . grad e, v) with b = ("bx","by","bz") at the points (thermal.cpp:150-160)
    pp.p[0] = have_advection ? 1.0 : 0.0;
    if (have_advection) {
      const char *bn[3] = {"bx", "by", "bz"};
      for (int d = 0; d < w.dimension; ++d) {
        pp.f[4 + d] = functionManager->evaluate(bn[d]);
      }
      // the deck-string instantiation of the engine is at the scratch it may use (common.hpp: require_modest_scratch)
      // and does not carry the advection term
      for (int k = 0; k < 7; ++k)
        MHA_REQUIRE(pp.f[k].kind != MHA_FUNC_EXPRESSION, MHA_ERR_INVALID,
                    "thermal with 'include advection': give the functions as constants, closed forms or per-point arrays "
                    "(deck strings are evaluated by the caller into an ip array)");
    }
    launch_point_engine(b, w.layout, pp, w.time_dev, w.res, w.elem_slot, w.elem_slot_bytes, w.stream);
  } else if (have_advection) {
    throw Error(MHA_ERR_INVALID, "thermal with 'include advection' runs on the point engine (MHA_PATH_AUTO / _ROW_GATHER / _POINT_ENGINE)");
  } else if (w.use_general)
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
  } else if (w.current_bc == MHA_BC_WEAK_DIRICHLET || w.current_bc == MHA_BC_INTERFACE) {
    // "interface": the same terms with the trace value "aux e" at the side points in place of the Dirichlet data
    // (thermal.cpp:227-243)
    bd.data = functionManager->evaluate((w.current_bc == MHA_BC_INTERFACE ? "aux e " : "Dirichlet e ") + w.sidename);
    bd.diff = functionManager->evaluate("thermal diffusion");
    MHA_REQUIRE(bd.diff.kind != MHA_FUNC_IP_ARRAY, MHA_ERR_INVALID,
                "weak Dirichlet needs 'thermal diffusion' at the side points: give it as a constant or closed form");
  } else {
    return;  // strong Dirichlet / none: no boundary term (thermal.cpp:188-216 falls through)
  }
  launch_thermal_boundary(w.dev, w.side_tables, bd, w.time_dev, w.res, w.stream);
}

// reference: thermal<EvalT>::computeFlux (src/physics/thermal.cpp:288-347): wkset->flux(elem, e, pt) on the current
// boundary group = (10 / h) kappa (lambda - T) + kappa grad T . n with lambda = "aux e <side>" at the side points
void thermal::computeFlux() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "thermal::computeFlux called without a workset");
  Workset &w = *wkset;
  BoundaryDev bd = w.bnd;
  MHA_REQUIRE(bd.flux != nullptr, MHA_ERR_INVALID, "computeFlux: no flux array on the workset");
  bd.bc_type = MHA_BC_INTERFACE;
  bd.form_param = formparam;
  bd.data = functionManager->evaluate("aux e " + w.sidename);
  bd.diff = functionManager->evaluate("thermal diffusion");
  MHA_REQUIRE(bd.diff.kind != MHA_FUNC_IP_ARRAY, MHA_ERR_INVALID,
              "computeFlux needs 'thermal diffusion' at the side points: give it as a constant or closed form");
  launch_thermal_boundary(w.dev, w.side_tables, bd, w.time_dev, ElemOut(), w.stream);
}

// ---- porousMixed -------------------------------------------------------------------------------------------------
porousMixed::porousMixed() {
  label = "porousMixed";
  myvars = {"p", "u"};               // reference: porousMixed.cpp:33-43
  mybasistypes = {"HVOL", "HDIV"};
}

// reference: porousMixed::defineFunctions (porousMixed.cpp:134-152): source 0, Kinv_* 1, total_mobility 1
void porousMixed::defineFunctions(FunctionManager &fm) {
  functionManager = &fm;
  auto constant = [](double v) { FuncDesc f; f.kind = MHA_FUNC_CONSTANT; f.amp = v; return f; };
  if (!fm.has("source")) fm.addFunction("source", constant(0.0));
  for (const char *k : {"Kinv_xx", "Kinv_yy", "Kinv_zz", "total_mobility"})
    if (!fm.has(k)) fm.addFunction(k, constant(1.0));
}

// reference: porousMixed::volumeResidual (porousMixed.cpp:158-338) as the point function porous_point
void porousMixed::volumeResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "porousMixed::volumeResidual called without a workset");
  Workset &w = *wkset;
  BlockDev b = w.dev;
  b.e_begin = w.first_elem;
  b.e_count = w.numElem;
  PhysParamsDev pp;
  pp.physics = MHA_PHYSICS_POROUS_MIXED;
  const char *names[5] = {"source", "Kinv_xx", "Kinv_yy", "Kinv_zz", "total_mobility"};
  for (int k = 0; k < 5; ++k) pp.f[k] = functionManager->evaluate(names[k]);
  // dense element arrays (row-gather path, mha_compute_local_jacres): the thread-per-element kernel; global outputs
  // (atomic scatter): the point engine.  MHA_POROUS_KERNEL=engine forces the engine.
  static const bool force_engine = [] { const char *m = std::getenv("MHA_POROUS_KERNEL"); return m && m[0] == 'e'; }();
  if (!force_engine && w.res.res == nullptr && w.res.crs_vals == nullptr)
    launch_porous_element(b, w.layout, pp, w.time_dev, w.res, w.stream);
  else
    launch_point_engine(b, w.layout, pp, w.time_dev, w.res, w.elem_slot, w.elem_slot_bytes, w.stream);
}

// reference: porousMixed::boundaryResidual (porousMixed.cpp:345-432): bcs(pnum, side) == "Dirichlet" adds
// "Dirichlet p <side>" * wts * (v . n) to the u rows
void porousMixed::boundaryResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "porousMixed::boundaryResidual called without a workset");
  Workset &w = *wkset;
  if (w.current_bc != MHA_BC_WEAK_DIRICHLET) return;  // other types contribute nothing (porousMixed.cpp:400, 420)
  BoundaryDev bd = w.bnd;
  bd.bc_type = w.current_bc;
  bd.data = functionManager->evaluate("Dirichlet p " + w.sidename);
  launch_porous_boundary(w.dev, w.side_tables, bd, w.layout, w.res, w.stream);
}

// ---- navierstokes ------------------------------------------------------------------------------------------------
navierstokes::navierstokes() {
  label = "navierstokes";
  myvars = {"ux", "pr", "uy", "uz"};  // reference: navierstokes.cpp:27-34 (uz only in 3-D)
  mybasistypes = {"HGRAD", "HGRAD", "HGRAD", "HGRAD"};
}

// reference: navierstokes::defineFunctions (navierstokes.cpp:62-76): sources 0, density 1, viscosity 1
void navierstokes::defineFunctions(FunctionManager &fm) {
  functionManager = &fm;
  auto constant = [](double v) { FuncDesc f; f.kind = MHA_FUNC_CONSTANT; f.amp = v; return f; };
  for (const char *k : {"source ux", "source pr", "source uy", "source uz"})
    if (!fm.has(k)) fm.addFunction(k, constant(0.0));
  for (const char *k : {"density", "viscosity"})
    if (!fm.has(k)) fm.addFunction(k, constant(1.0));
}

void navierstokes::setParameter(const std::string &name, double value) {
  if (name == "useSUPG") useSUPG = value != 0.0;
  else if (name == "usePSPG") usePSPG = value != 0.0;
  else if (name == "fix_uz_offsets") fix_uz_offsets = value != 0.0;
  else PhysicsBase::setParameter(name, value);
}

// reference: porousMixed<EvalT>::computeFlux (src/physics/porousMixed.cpp:440-500): flux(elem, auxp, pt) = u . n
void porousMixed::computeFlux() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "porousMixed::computeFlux called without a workset");
  Workset &w = *wkset;
  MHA_REQUIRE(w.bnd.flux != nullptr, MHA_ERR_INVALID, "computeFlux: no flux array on the workset");
  launch_porous_flux(w.dev, w.side_tables, w.bnd, w.layout, w.time_dev, w.stream);
}

// reference: navierstokes<EvalT>::computeFlux (src/physics/navierstokes.cpp:1016-1018) is empty: the flux view keeps the
// zeros the workset reset left in it
void navierstokes::computeFlux() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "navierstokes::computeFlux called without a workset");
  Workset &w = *wkset;
  MHA_REQUIRE(w.bnd.flux != nullptr, MHA_ERR_INVALID, "computeFlux: no flux array on the workset");
  const size_t npt = static_cast<size_t>(w.bnd.num) * w.side_tables.nqs;
  MHA_HIP(hipMemsetAsync(w.bnd.flux, 0, sizeof(double) * npt, w.stream));
  if (w.bnd.dflux_du) MHA_HIP(hipMemsetAsync(w.bnd.dflux_du, 0, sizeof(double) * npt * w.dev.n, w.stream));
  if (w.bnd.dflux_daux) MHA_HIP(hipMemsetAsync(w.bnd.dflux_daux, 0, sizeof(double) * npt, w.stream));
}

// reference: navierstokes::volumeResidual (navierstokes.cpp:82-849) as the point function navierstokes_point
void navierstokes::volumeResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "navierstokes::volumeResidual called without a workset");
  Workset &w = *wkset;
  BlockDev b = w.dev;
  b.e_begin = w.first_elem;
  b.e_count = w.numElem;
  PhysParamsDev pp;
  pp.physics = MHA_PHYSICS_NAVIERSTOKES;
  const char *names[6] = {"source ux", "source pr", "source uy", "source uz", "density", "viscosity"};
  for (int k = 0; k < 6; ++k) pp.f[k] = functionManager->evaluate(names[k]);
  pp.p[0] = useSUPG ? 1.0 : 0.0;
  pp.p[1] = usePSPG ? 1.0 : 0.0;
  pp.p[2] = fix_uz_offsets ? 1.0 : 0.0;
  launch_point_engine(b, w.layout, pp, w.time_dev, w.res, w.elem_slot, w.elem_slot_bytes, w.stream);
}

// ---- shallowwaterHybridized --------------------------------------------------------------------------------------
shallowwaterHybridized::shallowwaterHybridized() {
  label = "shallowwaterHybridized";
  myvars = {"H", "Hux", "Huy"};                // reference: shallowwaterHybridized.cpp:41-46
  mybasistypes = {"HGRAD", "HGRAD", "HGRAD"};
}

// reference: shallowwaterHybridized::defineFunctions (:78-108): sources 0
void shallowwaterHybridized::defineFunctions(FunctionManager &fm) {
  functionManager = &fm;
  auto constant = [](double v) { FuncDesc f; f.kind = MHA_FUNC_CONSTANT; f.amp = v; return f; };
  for (const char *k : {"source H", "source Hux", "source Huy"})
    if (!fm.has(k)) fm.addFunction(k, constant(0.0));
}

void shallowwaterHybridized::setParameter(const std::string &name, double value) {
  if (name == "g") gravity = value;
  else if (name == "Roe-like stabilization") roestab = value != 0.0;
  else if (name == "max EV stabilization") roestab = value == 0.0;
  else PhysicsBase::setParameter(name, value);
}

// reference: shallowwaterHybridized::volumeResidual (:113-184) as the point function swhdg_point
void shallowwaterHybridized::volumeResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "shallowwaterHybridized::volumeResidual called without a workset");
  Workset &w = *wkset;
  BlockDev b = w.dev;
  b.e_begin = w.first_elem;
  b.e_count = w.numElem;
  PhysParamsDev pp;
  pp.physics = MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED;
  const char *names[3] = {"source H", "source Hux", "source Huy"};
  for (int k = 0; k < 3; ++k) pp.f[k] = functionManager->evaluate(names[k]);
  pp.p[0] = gravity;
  launch_point_engine(b, w.layout, pp, w.time_dev, w.res, w.elem_slot, w.elem_slot_bytes, w.stream);
}

// reference: shallowwaterHybridized::boundaryResidual (:190-263) on the current boundary group; the trace ("aux") state
// and the far-field state are functions evaluated at the side points
void shallowwaterHybridized::boundaryResidual() {
  MHA_REQUIRE(wkset != nullptr, MHA_ERR_STATE, "shallowwaterHybridized::boundaryResidual called without a workset");
  Workset &w = *wkset;
  if (w.current_bc < MHA_BC_SWH_INTERFACE || w.current_bc > MHA_BC_SWH_SLIP) return;
  SwhBoundaryDev sw;
  sw.side_type = w.current_bc - MHA_BC_SWH_INTERFACE;
  sw.roe = roestab ? 1 : 0;
  sw.g = gravity;
  const char *vars[3] = {"H", "Hux", "Huy"};
  for (int i = 0; i < 3; ++i) {
    sw.aux[i] = functionManager->evaluate(std::string("aux ") + vars[i] + " " + w.sidename);
    if (w.current_bc == MHA_BC_SWH_FARFIELD)
      sw.farfield[i] = functionManager->evaluate(std::string("Far-field ") + vars[i] + " " + w.sidename);
    else
      sw.farfield[i] = sw.aux[i];
  }
  launch_swhdg_boundary(w.dev, w.side_tables, w.bnd, sw, w.time_dev, w.res, w.stream);
}

std::unique_ptr<PhysicsBase> import_physics(int physics_id) {
  if (physics_id == MHA_PHYSICS_THERMAL) return std::unique_ptr<PhysicsBase>(new thermal());
  if (physics_id == MHA_PHYSICS_POROUS_MIXED) return std::unique_ptr<PhysicsBase>(new porousMixed());
  if (physics_id == MHA_PHYSICS_NAVIERSTOKES) return std::unique_ptr<PhysicsBase>(new navierstokes());
  if (physics_id == MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED) return std::unique_ptr<PhysicsBase>(new shallowwaterHybridized());
  throw Error(MHA_ERR_INVALID, "unknown physics module id " + std::to_string(physics_id));
}

}  // namespace mha
