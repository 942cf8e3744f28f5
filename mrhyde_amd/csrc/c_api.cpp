// c_api.cpp -- extern "C" boundary (include/mrhyde_amd.h) over the C++ host layer.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>

#include "assembly_manager.hpp"
#include "expression.hpp"
#include "mesh.hpp"
#include "row_blocks.hpp"
#include "block_pattern.hpp"
#include "newton.hpp"
#include "test_hooks.h"
#include "scatter_plan.hpp"
#include "export_plan.hpp"

struct mha_row_partition {
  mha::RowBlocks rb;
};

struct mha_export_plan {
  std::unique_ptr<mha::ExportPlan> plan;
};
struct mha_comm {
  std::unique_ptr<mha::Comm> comm;
};

struct mha_context {
  mha::AssemblyManager mgr;
  explicit mha_context(const mha_block_desc &d) : mgr(d) {}
};

namespace {
thread_local std::string g_last_error;
// device the caller had current when the entry point was called, once mgr() has switched to the context's (-1: not switched)
thread_local int g_caller_device = -1;

struct RestoreCallerDevice {
  RestoreCallerDevice() { g_caller_device = -1; }
  ~RestoreCallerDevice() {
    if (g_caller_device >= 0) (void)hipSetDevice(g_caller_device);
    g_caller_device = -1;
  }
};

template <class F>
int guarded(F &&f) {
  RestoreCallerDevice restore;
  try {
    f();
    g_last_error.clear();
    return MHA_OK;
  } catch (const mha::Error &e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return MHA_ERR_INVALID;
  } catch (...) {
    g_last_error = "unknown error";
    return MHA_ERR_INVALID;
  }
}

// The context's manager, with the context's device made current (uploads, allocations, launches, stream and event
// creation all go to the device the context was created on); guarded() restores the caller's device.
mha::AssemblyManager &mgr(mha_context *ctx) {
  MHA_REQUIRE(ctx != nullptr, MHA_ERR_INVALID, "null context");
  int cur = -1;
  MHA_HIP(hipGetDevice(&cur));
  if (cur != ctx->mgr.device()) {
    if (g_caller_device < 0) g_caller_device = cur;
    MHA_HIP(hipSetDevice(ctx->mgr.device()));
  }
  return ctx->mgr;
}
}  // namespace

extern "C" {

const char *mha_last_error(void) { return g_last_error.c_str(); }
const char *mha_version(void) { return "mrhyde_amd 0.1 (gfx950)"; }

int mha_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mha_block_create(const mha_block_desc *desc, mha_context **out) {
  return guarded([&] {
    MHA_REQUIRE(desc && out, MHA_ERR_INVALID, "null argument");
    *out = nullptr;
    *out = new mha_context(*desc);
  });
}

void mha_block_destroy(mha_context *ctx) {
  if (!ctx) return;
  (void)guarded([&] {
    mha::DeviceGuard guard(ctx->mgr.device());
    delete ctx;
  });
}

int mha_set_stream(mha_context *ctx, void *hip_stream) {
  return guarded([&] { mgr(ctx).setStream(static_cast<hipStream_t>(hip_stream)); });
}

int mha_set_mesh(mha_context *ctx, int num_elems, const double *nodes, const int32_t *lids, const int32_t *offsets,
                 int num_rows, const uint8_t *is_fixed) {
  return guarded([&] { mgr(ctx).setMesh(num_elems, nodes, lids, offsets, num_rows, is_fixed); });
}

int mha_set_orientation(mha_context *ctx, const int8_t *signs_host) {
  return guarded([&] { mgr(ctx).setOrientation(signs_host); });
}

int mha_set_graph(mha_context *ctx, const int32_t *rowptr, const int32_t *colind) {
  return guarded([&] { mgr(ctx).setGraph(rowptr, colind); });
}

int mha_get_graph_sizes(mha_context *ctx, int *num_rows, int64_t *nnz) {
  return guarded([&] {
    MHA_REQUIRE(num_rows && nnz, MHA_ERR_INVALID, "null argument");
    *num_rows = mgr(ctx).numRows();
    *nnz = mgr(ctx).info("nnz");
  });
}

int mha_get_graph(mha_context *ctx, int32_t *rowptr, int32_t *colind) {
  return guarded([&] {
    auto &m = mgr(ctx);
    MHA_REQUIRE(!m.rowptr().empty(), MHA_ERR_STATE, "no CRS graph: call mha_set_graph first");
    MHA_REQUIRE(rowptr && colind, MHA_ERR_INVALID, "null argument");
    std::memcpy(rowptr, m.rowptr().data(), m.rowptr().size() * sizeof(int32_t));
    std::memcpy(colind, m.colind().data(), m.colind().size() * sizeof(int32_t));
  });
}

int mha_physics_select(mha_context *ctx, int physics_id) {
  return guarded([&] { mgr(ctx).selectPhysics(physics_id); });
}

int mha_set_function(mha_context *ctx, const char *name, int kind, double amp, const double *freq3,
                     const double *ip_array_dev) {
  return guarded([&] {
    MHA_REQUIRE(name != nullptr, MHA_ERR_INVALID, "null function name");
    mgr(ctx).setFunction(name, kind, amp, freq3, ip_array_dev);
  });
}

int mha_set_function_expression(mha_context *ctx, const char *name, const char *expression) {
  return guarded([&] {
    MHA_REQUIRE(name && expression, MHA_ERR_INVALID, "null argument");
    mgr(ctx).setFunctionExpression(name, expression);
  });
}

int mha_check_expression(const char *expression) {
  return guarded([&] {
    MHA_REQUIRE(expression, MHA_ERR_INVALID, "null argument");
    std::vector<int32_t> code;
    std::vector<double> consts;
    mha::compile_expression(expression, code, consts);
  });
}

int mha_set_time(mha_context *ctx, double time) {
  return guarded([&] { mgr(ctx).setTime(time); });
}

int mha_set_time_integration(mha_context *ctx, int transient, int num_steps, int num_stages, int stage,
                             double deltat, const double *A, const double *b, const double *bdf) {
  return guarded([&] { mgr(ctx).setTimeIntegration(transient, num_steps, num_stages, stage, deltat, A, b, bdf); });
}

int mha_assemble_jacres(mha_context *ctx, int flags, int path, const double *u, const double *u_prev,
                        const double *u_stage, double *res, double *crs_vals) {
  return guarded([&] { mgr(ctx).assembleJacRes(flags, path, u, u_prev, u_stage, res, crs_vals); });
}

int mha_compute_local_jacres(mha_context *ctx, int compute_jacobian, const double *u, const double *u_prev,
                             const double *u_stage, double *local_J, double *local_res) {
  return guarded([&] { mgr(ctx).computeLocalJacRes(compute_jacobian, u, u_prev, u_stage, local_J, local_res); });
}

int mha_get_mass(mha_context *ctx, const double *masswts_host, double *local_mass_dev) {
  return guarded([&] { mgr(ctx).getMass(masswts_host, local_mass_dev); });
}

int mha_scatter_local(mha_context *ctx, const double *local_J, const double *local_res, double *res,
                      double *crs_vals) {
  return guarded([&] { mgr(ctx).scatterLocal(local_J, local_res, res, crs_vals); });
}

int mha_apply_dbc_diag(mha_context *ctx, double *crs_vals) {
  return guarded([&] { mgr(ctx).applyDbcDiag(crs_vals); });
}

int mha_gather(mha_context *ctx, const double *vec, double *elem_vals) {
  return guarded([&] { mgr(ctx).gather(vec, elem_vals); });
}

int mha_num_worksets(mha_context *ctx) { return ctx ? ctx->mgr.numWorksets() : 0; }

int mha_workset_update(mha_context *ctx, int index) {
  return guarded([&] { mgr(ctx).worksetUpdate(index); });
}

int mha_workset_view(mha_context *ctx, const char *name, void **dev_ptr, int64_t extents[4], int *rank) {
  return guarded([&] {
    MHA_REQUIRE(name && dev_ptr && extents && rank, MHA_ERR_INVALID, "null argument");
    const mha::View v = mgr(ctx).worksetView(name);
    *dev_ptr = v.ptr;
    *rank = v.rank;
    for (int k = 0; k < 4; ++k) extents[k] = v.extent[k];
  });
}

int mha_add_boundary_group(mha_context *ctx, const char *sidename, int bc_type, int num_sides,
                           const int32_t *elem_ids_host, const int32_t *local_side_ids_host, int *group_id) {
  return guarded([&] {
    MHA_REQUIRE(sidename && group_id, MHA_ERR_INVALID, "null argument");
    *group_id = mgr(ctx).addBoundaryGroup(sidename, bc_type, num_sides, elem_ids_host, local_side_ids_host);
  });
}

int mha_add_flux_group(mha_context *ctx, const char *sidename, const char *varname, int num_sides,
                       const int32_t *elem_ids_host, const int32_t *local_side_ids_host, int *group_id) {
  return guarded([&] {
    MHA_REQUIRE(sidename && varname && group_id, MHA_ERR_INVALID, "null argument");
    *group_id = mgr(ctx).addFluxGroup(sidename, varname, num_sides, elem_ids_host, local_side_ids_host);
  });
}

int mha_add_dirichlet_group(mha_context *ctx, const char *sidename, const char *varname, int num_sides,
                            const int32_t *elem_ids_host, const int32_t *local_side_ids_host, int *group_id) {
  return guarded([&] {
    MHA_REQUIRE(sidename && varname && group_id, MHA_ERR_INVALID, "null argument");
    *group_id = mgr(ctx).addDirichletGroup(sidename, varname, num_sides, elem_ids_host, local_side_ids_host);
  });
}

int mha_set_initial(mha_context *ctx, int lump_mass, double *rhs, double *mass_vals) {
  return guarded([&] { mgr(ctx).setInitial(lump_mass, rhs, mass_vals); });
}

int mha_set_initial_nodal(mha_context *ctx, double *initial) {
  return guarded([&] { mgr(ctx).setInitialNodal(initial); });
}

int mha_set_dirichlet(mha_context *ctx, int lump_mass, double *rhs, double *mass_vals) {
  return guarded([&] { mgr(ctx).setDirichlet(lump_mass, rhs, mass_vals); });
}

int mha_workset_compute_solution(mha_context *ctx, const double *u, const double *u_prev, const double *u_stage) {
  return guarded([&] { mgr(ctx).worksetComputeSolution(u, u_prev, u_stage); });
}

int mha_workset_compute_residual(mha_context *ctx, int compute_jacobian, const double *u, const double *u_prev,
                                 const double *u_stage) {
  return guarded([&] { mgr(ctx).worksetComputeResidual(compute_jacobian, u, u_prev, u_stage); });
}

int mha_clear_boundary_groups(mha_context *ctx) {
  return guarded([&] { mgr(ctx).clearBoundaryGroups(); });
}

int mha_num_boundary_groups(mha_context *ctx) { return ctx ? ctx->mgr.numBoundaryGroups() : 0; }

int mha_assemble_boundary(mha_context *ctx, int flags, const double *u, const double *u_prev, const double *u_stage,
                          double *res, double *crs_vals) {
  return guarded([&] { mgr(ctx).assembleBoundary(flags, u, u_prev, u_stage, res, crs_vals); });
}

struct mha_newton {
  std::unique_ptr<mha::NewtonDriver> drv;
  mha_context *ctx = nullptr;
};

int mha_newton_create(mha_context *ctx, int max_iter, double nl_tol, double nl_abs_tol, int use_relative, int use_absolute,
                      int allow_backtracking, int autotune, mha_newton **out) {
  return guarded([&] {
    MHA_REQUIRE(out != nullptr, MHA_ERR_INVALID, "null argument");
    *out = nullptr;
    mha::NewtonSettings s;
    s.max_iter = max_iter;
    s.nl_tol = nl_tol;
    s.nl_abs_tol = nl_abs_tol;
    s.use_relative = use_relative;
    s.use_absolute = use_absolute;
    s.allow_backtracking = allow_backtracking;
    s.autotune = autotune;
    auto p = std::make_unique<mha_newton>();
    p->drv = std::make_unique<mha::NewtonDriver>(mgr(ctx), s);
    p->ctx = ctx;
    *out = p.release();
  });
}
void mha_newton_destroy(mha_newton *nw) { delete nw; }
int mha_newton_reset(mha_newton *nw) {
  return guarded([&] { MHA_REQUIRE(nw, MHA_ERR_INVALID, "null driver"); nw->drv->reset(); });
}
int mha_newton_residual(mha_newton *nw, const double *u, const double *u_prev, const double *u_stage, double *res) {
  return guarded([&] { MHA_REQUIRE(nw, MHA_ERR_INVALID, "null driver"); mgr(nw->ctx); nw->drv->residual(u, u_prev, u_stage, res); });
}
int mha_newton_norm(mha_newton *nw, const double *res, double *resnorm) {
  return guarded([&] { MHA_REQUIRE(nw && resnorm, MHA_ERR_INVALID, "null argument"); mgr(nw->ctx); *resnorm = nw->drv->norm(res); });
}
int mha_newton_decide(mha_newton *nw, double resnorm, double *u, int *action) {
  return guarded([&] { MHA_REQUIRE(nw && action, MHA_ERR_INVALID, "null argument"); mgr(nw->ctx); *action = nw->drv->decide(resnorm, u); });
}
int mha_newton_jacobian(mha_newton *nw, const double *u, const double *u_prev, const double *u_stage, double *res,
                        double *crs_vals) {
  return guarded([&] { MHA_REQUIRE(nw, MHA_ERR_INVALID, "null driver"); mgr(nw->ctx); nw->drv->jacobian(u, u_prev, u_stage, res, crs_vals); });
}
int mha_newton_update(mha_newton *nw, double *u, const double *du) {
  return guarded([&] { MHA_REQUIRE(nw, MHA_ERR_INVALID, "null driver"); mgr(nw->ctx); nw->drv->update(u, du); });
}
int mha_newton_step(mha_newton *nw, double *u, const double *u_prev, const double *u_stage, double *res, double *crs_vals,
                    int *action) {
  return guarded([&] {
    MHA_REQUIRE(nw && action, MHA_ERR_INVALID, "null argument");
    mgr(nw->ctx);
    *action = nw->drv->step(u, u_prev, u_stage, res, crs_vals);
  });
}
int mha_newton_state(const mha_newton *nw, int *iteration, double *resnorm, double *resnorm_scaled, double *resnorm_first,
                     double *alpha, int *status) {
  return guarded([&] {
    MHA_REQUIRE(nw, MHA_ERR_INVALID, "null driver");
    if (iteration) *iteration = nw->drv->iteration();
    if (resnorm) *resnorm = nw->drv->resnorm();
    if (resnorm_scaled) *resnorm_scaled = nw->drv->resnormScaled();
    if (resnorm_first) *resnorm_first = nw->drv->resnormFirst();
    if (alpha) *alpha = nw->drv->alpha();
    if (status) *status = nw->drv->status();
  });
}
int mha_dirichlet_lift(mha_context *ctx, double *u, const double *fixed_soln, double scalar_value) {
  return guarded([&] { mgr(ctx).dirichletLift(u, fixed_soln, scalar_value); });
}

int mha_compute_flux(mha_context *ctx, int group_id, const double *u, const double *u_prev, const double *u_stage,
                     double *flux, double *dflux_du, double *dflux_daux) {
  return guarded([&] { mgr(ctx).computeFlux(group_id, u, u_prev, u_stage, flux, dflux_du, dflux_daux); });
}

int mha_boundary_update(mha_context *ctx, int group_id) {
  return guarded([&] { mgr(ctx).boundaryUpdate(group_id); });
}

int mha_boundary_view(mha_context *ctx, int group_id, const char *name, void **dev_ptr, int64_t extents[4], int *rank) {
  return guarded([&] {
    MHA_REQUIRE(name && dev_ptr && extents && rank, MHA_ERR_INVALID, "null argument");
    const mha::View v = mgr(ctx).boundaryView(group_id, name);
    *dev_ptr = v.ptr;
    *rank = v.rank;
    for (int k = 0; k < 4; ++k) extents[k] = v.extent[k];
  });
}

int mha_set_physics_parameter(mha_context *ctx, const char *name, double value) {
  return guarded([&] {
    MHA_REQUIRE(name, MHA_ERR_INVALID, "null argument");
    mgr(ctx).setPhysicsParameter(name, value);
  });
}

int mha_swhdg_side_terms(int side_type, int roe_stabilization, double g, int64_t npts, const double *S, const double *Shat,
                         const double *normals, const double *Sinf, double *fluxvec, double *term, double *iflux,
                         double *d_iflux_dS, double *d_iflux_dShat, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(side_type == MHA_SWH_INTERFACE || side_type == MHA_SWH_FARFIELD || side_type == MHA_SWH_SLIP,
                MHA_ERR_INVALID, "unknown side type " << side_type);
    MHA_REQUIRE(npts >= 0 && (npts == 0 || (S && Shat && normals)), MHA_ERR_INVALID, "null state / normal arrays");
    MHA_REQUIRE(side_type != MHA_SWH_FARFIELD || Sinf, MHA_ERR_INVALID, "Far-field sides need the far-field state");
    MHA_REQUIRE(g > 0.0, MHA_ERR_INVALID, "g must be positive");
    mha::SwhSideArgs a;
    a.npts = npts; a.side_type = side_type; a.roe = roe_stabilization ? 1 : 0; a.g = g;
    a.S = S; a.Shat = Shat; a.normals = normals; a.Sinf = Sinf;
    a.fluxvec = fluxvec; a.term = term; a.iflux = iflux; a.d_dS = d_iflux_dS; a.d_dShat = d_iflux_dShat;
    mha::launch_swhdg_side(a, static_cast<hipStream_t>(hip_stream));
  });
}

int mha_swhdg_element_blocks(mha_context *ctx, const double *u, const double *u_prev, const double *u_stage,
                             const double *lambda, const uint8_t *side_types, const double *farfield_host, double *res,
                             double *blocks) {
  return guarded([&] { mgr(ctx).swhdgElementBlocks(u, u_prev, u_stage, lambda, side_types, farfield_host, res, blocks); });
}

int mha_batched_condense(int n_int, int n_trace, int64_t num_elems, const double *blocks, const double *res, double *schur,
                         double *gvec, double *du, int *num_singular_host, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(num_elems >= 0 && (num_elems == 0 || (blocks && res)), MHA_ERR_INVALID, "null element blocks");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    mha::DeviceBuffer<int> flag(1);
    MHA_HIP(hipMemsetAsync(flag.data(), 0, sizeof(int), st));
    mha::launch_condense(n_int, n_trace, num_elems, blocks, res, schur, gvec, du, flag.data(), st);
    int h = 0;
    MHA_HIP(hipMemcpyAsync(&h, flag.data(), sizeof(int), hipMemcpyDeviceToHost, st));
    MHA_HIP(hipStreamSynchronize(st));
    if (num_singular_host) *num_singular_host = h;
    MHA_REQUIRE(h == 0 || num_singular_host, MHA_ERR_INVALID, h << " element(s) with a singular interior block");
  });
}

int mha_swhdg_subgrid_workspace_bytes(mha_context *ctx, int64_t *bytes) {
  return guarded([&] {
    MHA_REQUIRE(bytes, MHA_ERR_INVALID, "null argument");
    *bytes = static_cast<int64_t>(mgr(ctx).subgridWorkspaceBytes());
  });
}

int mha_swhdg_subgrid_solve(mha_context *ctx, double *u, const double *u_prev, const double *u_stage, const double *lambda,
                            const uint8_t *side_types, const double *farfield_host, int max_iter, double tol, void *workspace,
                            int64_t workspace_bytes, double *schur, double *gvec, int32_t *iters, double *resnorm_scaled,
                            int32_t *num_singular) {
  return guarded([&] {
    mgr(ctx).subgridSolve(u, u_prev, u_stage, lambda, side_types, farfield_host, max_iter, tol, workspace,
                          static_cast<size_t>(std::max<int64_t>(workspace_bytes, 0)), schur, gvec, iters, resnorm_scaled,
                          num_singular);
  });
}

int mha_swhdg_condensed_element(mha_context *ctx, const double *u, const double *u_prev, const double *u_stage,
                                const double *lambda, const uint8_t *side_types, const double *farfield_host, double *schur,
                                double *gvec, double *du, int32_t *num_singular) {
  return guarded([&] {
    MHA_REQUIRE(schur || gvec || du, MHA_ERR_INVALID, "no output requested");
    mha::SwhFusedOut o;
    o.schur = schur;
    o.gvec = gvec;
    o.du = du;
    o.singular = num_singular;
    mgr(ctx).swhdgCondensedElement(u, u_prev, u_stage, lambda, side_types, farfield_host, o);
  });
}

int mha_swhdg_eigendecomp(double g, int64_t npts, const double *Shat, const double *normals, double *L, double *lam,
                          double *R, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(npts >= 0 && (npts == 0 || (Shat && normals && L && lam && R)), MHA_ERR_INVALID, "null argument");
    MHA_REQUIRE(g > 0.0, MHA_ERR_INVALID, "g must be positive");
    mha::SwhSideArgs a;
    a.npts = npts; a.g = g; a.S = Shat; a.Shat = Shat; a.normals = normals;
    a.L = L; a.lam = lam; a.R = R;
    mha::launch_swhdg_side(a, static_cast<hipStream_t>(hip_stream));
  });
}

int mha_mesh_sizes(int dim, int order, const int *ncell, int *nverts, int *nelem, int64_t *ndof) {
  return guarded([&] {
    MHA_REQUIRE(ncell && nverts && nelem && ndof, MHA_ERR_INVALID, "null argument");
    mha::mesh_sizes(dim, order, ncell, nverts, nelem, ndof);
  });
}

int mha_mesh_structured(int dim, int order, const int *ncell, const double *lo, const double *hi, double *verts,
                        int32_t *cell2vert, int32_t *lids, int32_t *offsets, uint8_t *boundary_dof) {
  return guarded([&] {
    MHA_REQUIRE(ncell && lo && hi && verts && cell2vert && lids && offsets, MHA_ERR_INVALID, "null argument");
    MHA_REQUIRE(order >= 1 && order <= 8, MHA_ERR_INVALID, "HGRAD order must be in [1,8]");
    mha::mesh_structured(dim, order, ncell, lo, hi, verts, cell2vert, lids, offsets, boundary_dof);
  });
}

int mha_mesh_multi_sizes(int dim, const int *ncell, int nvars, const int *types, const int *orders, int *nverts,
                         int *nelem, int *n_tot, int64_t *ndof) {
  return guarded([&] {
    MHA_REQUIRE(ncell && types && orders && nverts && nelem && n_tot && ndof, MHA_ERR_INVALID, "null argument");
    mha::mesh_multi_sizes(dim, ncell, nvars, types, orders, nverts, nelem, n_tot, ndof);
  });
}

int mha_mesh_structured_multi(int dim, const int *ncell, const double *lo, const double *hi, int nvars, const int *types,
                              const int *orders, double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                              int8_t *orient, uint8_t *side_mask, int32_t *dof_var) {
  return guarded([&] {
    MHA_REQUIRE(ncell && lo && hi && types && orders && verts && cell2vert && lids && offsets, MHA_ERR_INVALID, "null argument");
    mha::mesh_structured_multi(dim, ncell, lo, hi, nvars, types, orders, verts, cell2vert, lids, offsets, orient, side_mask, dof_var);
  });
}

int mha_row_partition_build(int dim, int num_elems, int n, int num_rows, const double *nodes, const int32_t *lids,
                            const int32_t *rowptr, const int *caps, mha_row_partition **out) {
  return guarded([&] {
    MHA_REQUIRE(nodes && lids && rowptr && out, MHA_ERR_INVALID, "null argument");
    MHA_REQUIRE((dim == 2 || dim == 3) && num_elems > 0 && n > 0 && num_rows > 0, MHA_ERR_INVALID, "bad sizes");
    mha::RowBlockCaps c = mha::default_caps(dim, n);
    if (caps) {
      c.chunk_elems = caps[0];
      c.max_acc = caps[1];
      c.max_rows = caps[2];
      c.max_elems = caps[3];
      c.max_pairs = c.max_rows * 8;
    }
    *out = nullptr;
    auto *p = new mha_row_partition();
    try {
      p->rb = mha::build_row_blocks(dim, 1 << dim, num_elems, n, num_rows, nodes, lids, rowptr, c);
    } catch (...) {
      delete p;
      throw;
    }
    *out = p;
  });
}

struct mha_scatter_plan {
  std::unique_ptr<mha::ScatterPlan> plan;
};

// ---- Sparse3DView, database, matrix-free mass apply -------------------------------------------------------------
struct mha_sparse3d {
  int64_t num_elems = 0;
  int n = 0, maxent = 0;
  mha::DeviceBuffer<double> values;
  mha::DeviceBuffer<int32_t> columns, nnz_row;
};

int mha_sparse3d_create(int64_t num_elems, int n, const double *dense_dev, double tol, void *hip_stream, mha_sparse3d **out) {
  return guarded([&] {
    MHA_REQUIRE(out && dense_dev && num_elems > 0 && n > 0, MHA_ERR_INVALID, "bad Sparse3DView input");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    std::unique_ptr<mha_sparse3d> s(new mha_sparse3d());
    s->num_elems = num_elems;
    s->n = n;
    const size_t rows = static_cast<size_t>(num_elems) * n;
    mha::DeviceBuffer<unsigned long long> maxbits(1);
    mha::DeviceBuffer<int> maxent(1);
    MHA_HIP(hipMemsetAsync(maxbits.data(), 0, sizeof(unsigned long long), st));
    MHA_HIP(hipMemsetAsync(maxent.data(), 0, sizeof(int), st));
    s->nnz_row.resize(rows);
    mha::launch_sparse3d_max(dense_dev, rows * n, maxbits.data(), st);
    mha::launch_sparse3d_count(dense_dev, rows, n, tol, maxbits.data(), s->nnz_row.data(), maxent.data(), st);
    MHA_HIP(hipStreamSynchronize(st));
    maxent.download(&s->maxent);
    const size_t cnt = std::max<size_t>(1, rows * static_cast<size_t>(s->maxent));
    s->values.resize(cnt);
    s->columns.resize(cnt);
    MHA_HIP(hipMemsetAsync(s->values.data(), 0, cnt * sizeof(double), st));
    MHA_HIP(hipMemsetAsync(s->columns.data(), 0, cnt * sizeof(int32_t), st));
    if (s->maxent > 0) mha::launch_sparse3d_fill(dense_dev, rows, n, tol, maxbits.data(), s->maxent, s->values.data(), s->columns.data(), st);
    MHA_HIP(hipStreamSynchronize(st));
    *out = s.release();
  });
}

int mha_sparse3d_views(mha_sparse3d *s, int *maxent, double **values_dev, int32_t **columns_dev, int32_t **nnz_row_dev) {
  return guarded([&] {
    MHA_REQUIRE(s != nullptr, MHA_ERR_INVALID, "null Sparse3DView");
    if (maxent) *maxent = s->maxent;
    if (values_dev) *values_dev = s->values.data();
    if (columns_dev) *columns_dev = s->columns.data();
    if (nnz_row_dev) *nnz_row_dev = s->nnz_row.data();
  });
}

int mha_sparse3d_size(mha_sparse3d *s, int64_t *total_nnz) {
  return guarded([&] {
    MHA_REQUIRE(s && total_nnz, MHA_ERR_INVALID, "null argument");
    std::vector<int32_t> h(s->nnz_row.size());
    s->nnz_row.download(h.data());
    int64_t t = 0;
    for (int32_t v : h) t += v;
    *total_nnz = t;
  });
}

void mha_sparse3d_destroy(mha_sparse3d *s) { delete s; }

int mha_database_build(mha_context *ctx, int *num_unique) {
  return guarded([&] {
    const int nu = mgr(ctx).databaseBuild();
    if (num_unique) *num_unique = nu;
  });
}

int mha_database_get(mha_context *ctx, int32_t *index_host, int32_t *first_users_host) {
  return guarded([&] { mgr(ctx).databaseGet(index_host, first_users_host); });
}

int mha_apply_mass_matrix_free(mha_context *ctx, int mode, const double *masswts_host, const double *mass_dev,
                               mha_sparse3d *sparse, const double *x_dev, double *y_dev) {
  return guarded([&] {
    mha::AssemblyManager &m = mgr(ctx);
    if (mode == MHA_MASS_DATABASE_SPARSE) {
      MHA_REQUIRE(sparse != nullptr, MHA_ERR_INVALID, "sparse mass format needs a Sparse3DView");
      m.applyMassMatrixFree(mode, masswts_host, nullptr, sparse->maxent, sparse->nnz_row.data(), sparse->values.data(),
                            sparse->columns.data(), x_dev, y_dev);
    } else {
      m.applyMassMatrixFree(mode, masswts_host, mass_dev, 0, nullptr, nullptr, nullptr, x_dev, y_dev);
    }
  });
}

int mha_scatter_plan_create(int n, int64_t num_elems, int64_t num_rows, const int32_t *lids_host,
                            const int32_t *rowptr_host, const int32_t *colind_host, const uint8_t *fixed_host,
                            mha_scatter_plan **out) {
  return guarded([&] {
    MHA_REQUIRE(out && lids_host, MHA_ERR_INVALID, "null argument");
    MHA_REQUIRE((rowptr_host != nullptr) == (colind_host != nullptr), MHA_ERR_INVALID, "rowptr and colind go together");
    MHA_REQUIRE(num_elems > 0 && num_elems < (int64_t(1) << 31) && num_rows > 0 && num_rows < (int64_t(1) << 31),
                MHA_ERR_INVALID, "bad sizes");
    *out = nullptr;
    auto p = std::make_unique<mha_scatter_plan>();
    p->plan = std::make_unique<mha::ScatterPlan>(n, static_cast<int>(num_elems), static_cast<int>(num_rows), lids_host,
                                                  rowptr_host, colind_host, fixed_host);
    *out = p.release();
  });
}

int mha_scatter_plan_nnz(const mha_scatter_plan *p, int64_t *nnz) {
  return guarded([&] {
    MHA_REQUIRE(p && nnz, MHA_ERR_INVALID, "null argument");
    *nnz = p->plan->nnz();
  });
}

int mha_scatter_plan_graph(const mha_scatter_plan *p, int32_t *rowptr_host, int32_t *colind_host) {
  return guarded([&] {
    MHA_REQUIRE(p && rowptr_host && colind_host, MHA_ERR_INVALID, "null argument");
    p->plan->graph(rowptr_host, colind_host);
  });
}

int mha_scatter_plan_apply(const mha_scatter_plan *p, const double *blocks_dev, const double *vec_dev, double *res_dev,
                           double *vals_dev, int overwrite, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(p, MHA_ERR_INVALID, "null plan");
    p->plan->apply(blocks_dev, vec_dev, res_dev, vals_dev, overwrite != 0, static_cast<hipStream_t>(hip_stream));
  });
}

void mha_scatter_plan_destroy(mha_scatter_plan *p) { delete p; }

int mha_test_block_patterns_host_apply(int dim, int num_rows, int num_elems, int nnodes, int n, int nsym,
                                       const double *nodes, const int32_t *lids, const int32_t *rowptr,
                                       const int32_t *colind, const uint8_t *fixed, const double *khat,
                                       const double *factors, double scale_u, double scale_t, int chunk_elems,
                                       int num_cus, int max_patterns, double *vals, int *counts) {
  return guarded([&] {
    MHA_REQUIRE(nodes && lids && rowptr && colind && khat && factors && vals && counts, MHA_ERR_INVALID, "null argument");
    MHA_REQUIRE(num_rows > 0 && num_elems > 0 && n > 0 && n <= 255 && nsym > 0, MHA_ERR_INVALID, "bad sizes");
    // element-major slot map by column search (the device builds the same map in build_elem_slot_map_kernel)
    int max_row = 0;
    for (int r = 0; r < num_rows; ++r) max_row = std::max(max_row, rowptr[r + 1] - rowptr[r]);
    const int sb = max_row <= 256 ? 1 : 2;
    std::vector<uint8_t> slot(static_cast<size_t>(num_elems) * n * n * sb);
    for (int e = 0; e < num_elems; ++e)
      for (int si = 0; si < n; ++si) {
        const int row = lids[static_cast<size_t>(e) * n + si];
        const int32_t *lo = colind + rowptr[row], *hi = colind + rowptr[row + 1];
        for (int sj = 0; sj < n; ++sj) {
          const int32_t *it = std::lower_bound(lo, hi, lids[static_cast<size_t>(e) * n + sj]);
          MHA_REQUIRE(it != hi && *it == lids[static_cast<size_t>(e) * n + sj], MHA_ERR_INVALID, "graph misses an element coupling");
          const size_t idx = (static_cast<size_t>(e) * n + si) * n + sj;
          if (sb == 1) slot[idx] = static_cast<uint8_t>(it - lo);
          else reinterpret_cast<uint16_t *>(slot.data())[idx] = static_cast<uint16_t>(it - lo);
        }
      }
    mha::RowBlockCaps caps = mha::default_caps(dim, n);
    caps.chunk_elems = chunk_elems > 0 ? chunk_elems : 16;
    caps.max_rows = 4096;
    caps.max_elems = 255;
    caps.max_pairs = 1 << 20;
    caps.max_acc = 1 << 30;
    const mha::RowBlocks rb = mha::build_row_blocks(dim, nnodes, num_elems, n, num_rows, nodes, lids, rowptr, caps, fixed, 1);
    const mha::BlockPatternPlan pl = mha::build_block_patterns(rb, n, nsym, rowptr, fixed, slot.data(), sb, khat,
                                                               num_cus > 0 ? num_cus : 8, size_t(150) * 1024,
                                                               max_patterns > 0 ? max_patterns : 256);
    MHA_REQUIRE(pl.usable, MHA_ERR_INVALID, "row blocks do not group: " << pl.why);
    counts[0] = pl.num_patterns;
    counts[1] = pl.num_roles;
    counts[2] = pl.num_wgs;
    counts[3] = pl.num_parts;
    mha::block_patterns_host_apply(pl, factors, scale_u, scale_t, true, vals);
  });
}

int mha_export_plan_create(int num_neighbors, const int32_t *neighbor_ranks, const int64_t *send_val_ptr,
                           const int32_t *send_val_index, const int64_t *send_row_ptr, const int32_t *send_row_index,
                           const int64_t *recv_val_ptr, const int32_t *recv_val_target, const int64_t *recv_row_ptr,
                           const int32_t *recv_row_target, int64_t nnz, int64_t nrows, mha_export_plan **out) {
  return guarded([&] {
    MHA_REQUIRE(out != nullptr, MHA_ERR_INVALID, "null argument");
    *out = nullptr;
    auto p = std::make_unique<mha_export_plan>();
    p->plan = std::make_unique<mha::ExportPlan>(num_neighbors, neighbor_ranks, send_val_ptr, send_val_index, send_row_ptr,
                                                send_row_index, recv_val_ptr, recv_val_target, recv_row_ptr, recv_row_target,
                                                nnz, nrows);
    *out = p.release();
  });
}

void mha_export_plan_destroy(mha_export_plan *p) { delete p; }

int mha_export_pack(const mha_export_plan *p, const double *vals_dev, const double *res_dev, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(p, MHA_ERR_INVALID, "null plan");
    p->plan->pack(vals_dev, res_dev, static_cast<hipStream_t>(hip_stream));
  });
}

int mha_export_unpack_add(const mha_export_plan *p, double *vals_dev, double *res_dev, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(p, MHA_ERR_INVALID, "null plan");
    p->plan->unpackAdd(vals_dev, res_dev, static_cast<hipStream_t>(hip_stream));
  });
}

int mha_export_buffers(const mha_export_plan *p, int k, double **send_dev, int64_t *send_count, double **recv_dev,
                       int64_t *recv_count) {
  return guarded([&] {
    MHA_REQUIRE(p && send_dev && send_count && recv_dev && recv_count, MHA_ERR_INVALID, "null argument");
    *send_dev = p->plan->sendBuffer(k, send_count);
    *recv_dev = p->plan->recvBuffer(k, recv_count);
  });
}

int mha_export_bytes_on_wire(const mha_export_plan *p, int64_t *bytes) {
  return guarded([&] {
    MHA_REQUIRE(p && bytes, MHA_ERR_INVALID, "null argument");
    *bytes = p->plan->bytesOnWire();
  });
}

int mha_comm_unique_id(char id_out[128]) {
  return guarded([&] {
    MHA_REQUIRE(id_out != nullptr, MHA_ERR_INVALID, "null argument");
    mha::Comm::uniqueId(id_out);
  });
}

int mha_comm_create(int num_ranks, int rank, const char id[128], mha_comm **out) {
  return guarded([&] {
    MHA_REQUIRE(out != nullptr, MHA_ERR_INVALID, "null argument");
    *out = nullptr;
    auto c = std::make_unique<mha_comm>();
    c->comm = std::make_unique<mha::Comm>(num_ranks, rank, id);
    *out = c.release();
  });
}

void mha_comm_destroy(mha_comm *c) { delete c; }

int mha_export_add(const mha_export_plan *p, mha_comm *c, double *vals_dev, double *res_dev, void *hip_stream) {
  return guarded([&] {
    MHA_REQUIRE(p && c, MHA_ERR_INVALID, "null plan or communicator");
    p->plan->exportAdd(*c->comm, vals_dev, res_dev, static_cast<hipStream_t>(hip_stream));
  });
}

int mha_row_partition_sizes(const mha_row_partition *p, int *num_blocks, int64_t *num_rows, int64_t *num_elems,
                            int *max_rows, int *max_elems, int *max_entries) {
  return guarded([&] {
    MHA_REQUIRE(p && num_blocks && num_rows && num_elems && max_rows && max_elems && max_entries, MHA_ERR_INVALID,
                "null argument");
    *num_blocks = p->rb.num_blocks;
    *num_rows = static_cast<int64_t>(p->rb.rows.size());
    *num_elems = static_cast<int64_t>(p->rb.elems.size());
    *max_rows = p->rb.max_rows;
    *max_elems = p->rb.max_elems;
    *max_entries = p->rb.max_acc;
  });
}

int mha_row_partition_get(const mha_row_partition *p, int32_t *row_ptr, int32_t *rows, int32_t *elem_ptr,
                          int32_t *elems) {
  return guarded([&] {
    MHA_REQUIRE(p && row_ptr && rows && elem_ptr && elems, MHA_ERR_INVALID, "null argument");
    std::memcpy(row_ptr, p->rb.row_ptr.data(), p->rb.row_ptr.size() * sizeof(int32_t));
    std::memcpy(rows, p->rb.rows.data(), p->rb.rows.size() * sizeof(int32_t));
    std::memcpy(elem_ptr, p->rb.elem_ptr.data(), p->rb.elem_ptr.size() * sizeof(int32_t));
    std::memcpy(elems, p->rb.elems.data(), p->rb.elems.size() * sizeof(int32_t));
  });
}

void mha_row_partition_destroy(mha_row_partition *p) { delete p; }

int mha_get_info(mha_context *ctx, const char *key, int64_t *value) {
  return guarded([&] {
    MHA_REQUIRE(key && value, MHA_ERR_INVALID, "null argument");
    *value = mgr(ctx).info(key);
  });
}

int mha_set_timing(mha_context *ctx, int enable) {
  return guarded([&] { mgr(ctx).setTiming(enable != 0); });
}

int mha_get_last_kernel_ms(mha_context *ctx, double *ms) {
  return guarded([&] {
    MHA_REQUIRE(ms != nullptr, MHA_ERR_INVALID, "null argument");
    *ms = mgr(ctx).lastKernelMs();
  });
}

}  // extern "C"
