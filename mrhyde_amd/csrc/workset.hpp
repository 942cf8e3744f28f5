// workset.hpp -- host mirror of the reference's Workset<EvalT> for the MI355X path.
//
// In the reference a Workset (src/tools/workset.hpp:22-584) is the per-block scratch object the
// physics functors read (basis, basis_grad, wts, offsets, solution fields) and write (res).  Here
// it describes a contiguous element range of the block plus where the functor's result goes:
// the fused kernels evaluate gather / seeding / basis / fields on chip, so `res` is not a
// (numElem, maxRes) SFad array but a target descriptor (dense local_J/local_res, or the global
// residual + CRS values).  The classic views remain available through update_views() for code
// written against the reference API.
#pragma once
#include <map>
#include <string>

#include "common.hpp"
#include "kernels/device_types.hpp"
#include "kernels/launch.hpp"

namespace mha {

// Device array + extents, LayoutRight (the reference's View_Sc2/View_Sc4 on AssemblyDevice).
struct View {
  void *ptr = nullptr;
  int rank = 0;
  int64_t extent[4] = {0, 0, 0, 0};
  bool is_int = false;
};

class Workset {
 public:
  // --- what the reference exposes as public members (workset.hpp:360-560) ---
  int block = 0;
  int dimension = 0;
  int numElem = 0;       // elements in the current range
  int numip = 0;
  int numVars = 0;
  int maxElem = 0;       // "workset size"
  double time = 0.0;
  double deltat = 1.0;
  bool isTransient = false;
  int current_stage = 0;

  // --- element range + block data the kernels consume ---
  int first_elem = 0;
  BlockDev dev;          // device pointers of the whole block
  TimeDev time_dev;      // seeding coefficients (Workset::computeSolnTransientSeeded)
  ElemOut res;           // where volumeResidual()'s result is accumulated
  // general-element kernel (kernels/thermal_general.hip): 1-D tables and the element-major slot map;
  // elem_slot == nullptr selects the baseline one-wave-per-element kernel (kernels/thermal_element.hip)
  AffineDev tables;
  const void *elem_slot = nullptr;
  int elem_slot_bytes = 1;
  bool use_general = false;
  // multi-variable point engine (kernels/point_engine.hip): variable/slot layout, orientation signs
  VarLayoutDev layout;
  bool use_point_engine = false;
  bool single_hgrad = true;  // basis / basis_grad views exist for single-variable HGRAD blocks only
  hipStream_t stream = nullptr;
  int order = 0, nq1 = 0;

  // --- boundary state (reference: wkset->sidename / currentside / var_bcs, set by updateWorksetBoundary,
  // assemblyManager.cpp:5646-5710) ---
  std::string sidename;     // side-set name of the boundary group being processed
  int current_bc = 0;       // MHA_BC_* of variable "e" on that side
  BoundaryDev bnd;          // entries of the group (data/diff/form_param are filled by the physics module)
  SideTablesDev side_tables;

  // reference: Workset::setTime / setDeltat / setStage (workset.hpp:127-137)
  void setTime(double t) { time = t; }
  void setDeltat(double dt) { deltat = dt; }
  void setStage(int s) { current_stage = s; }

  // Evaluate basis / basis_grad / wts / x,y,z for the current range into owned buffers
  // (Group::computeBasis + updateWorkset aliasing, reference: src/tools/group.cpp:134-243,
  // src/managers/assemblyManager.cpp:6512-6596).
  void update_views();
  // reference: getBasis/getBasisGrad/getWeights/getScalarField/getOffsets (workset.hpp:193-316)
  View get(const std::string &name) const;

 private:
  DeviceBuffer<double> basis_, basis_grad_, wts_, xyz_[3];
  int views_first_ = -1, views_num_ = 0;
};

inline void Workset::update_views() {
  const size_t ne = static_cast<size_t>(numElem), n = dev.n, nq = dev.nq, d = dev.dim;
  const size_t cap = static_cast<size_t>(maxElem > numElem ? maxElem : numElem);
  if (single_hgrad && basis_.size() < cap * n * nq) {
    basis_.resize(cap * n * nq);
    basis_grad_.resize(cap * n * nq * d);
  }
  if (wts_.size() < cap * nq) {
    wts_.resize(cap * nq);
    for (size_t k = 0; k < d; ++k) xyz_[k].resize(cap * nq);
  }
  WorksetViewsDev v;
  v.basis = single_hgrad ? basis_.data() : nullptr;
  v.basis_grad = single_hgrad ? basis_grad_.data() : nullptr;
  v.wts = wts_.data();
  for (size_t k = 0; k < d; ++k) v.xyz[k] = xyz_[k].data();
  launch_workset_views(dev, first_elem, static_cast<int>(ne), v, stream);
  views_first_ = first_elem;
  views_num_ = numElem;
}

inline View Workset::get(const std::string &name) const {
  View v;
  const int64_t ne = views_num_, n = dev.n, nq = dev.nq;
  auto need_views = [&]() {
    MHA_REQUIRE(views_first_ >= 0, MHA_ERR_STATE, "workset views requested before mha_workset_update");
  };
  if ((name == "basis" || name == "basis_grad") && !single_hgrad) {
    throw Error(MHA_ERR_UNKNOWN_FIELD, "'" + name + "' views are available for single-variable HGRAD blocks");
  } else if (name == "basis") {
    need_views();
    v.ptr = basis_.data(); v.rank = 4; v.extent[0] = ne; v.extent[1] = n; v.extent[2] = nq; v.extent[3] = 1;
  } else if (name == "basis_grad") {
    need_views();
    v.ptr = basis_grad_.data(); v.rank = 4; v.extent[0] = ne; v.extent[1] = n; v.extent[2] = nq; v.extent[3] = dev.dim;
  } else if (name == "wts") {
    need_views();
    v.ptr = wts_.data(); v.rank = 2; v.extent[0] = ne; v.extent[1] = nq;
  } else if (name == "x" || name == "y" || name == "z") {
    need_views();
    const int k = name[0] - 'x';
    MHA_REQUIRE(k < dev.dim, MHA_ERR_UNKNOWN_FIELD, "scalar field '" << name << "' does not exist in " << dev.dim << "-D");
    v.ptr = xyz_[k].data(); v.rank = 2; v.extent[0] = ne; v.extent[1] = nq;
  } else if (name == "LIDs") {
    need_views();
    v.ptr = const_cast<int32_t *>(dev.lids + static_cast<size_t>(views_first_) * n);
    v.rank = 2; v.extent[0] = ne; v.extent[1] = n; v.is_int = true;
  } else if (name == "offsets") {
    // one flat list for the block: the dofs of variable v occupy positions varptr[v] .. varptr[v + 1] of it (the
    // reference's offsets(var, dof) ragged array, workset.hpp:61, concatenated)
    v.ptr = const_cast<int32_t *>(dev.offsets); v.rank = 2; v.extent[0] = 1; v.extent[1] = n; v.is_int = true;
  } else {
    // the reference prints "Error: could not find ..." and continues (workset.cpp:1576-1577)
    throw Error(MHA_ERR_UNKNOWN_FIELD, "unknown workset view '" + name + "'");
  }
  return v;
}

}  // namespace mha
