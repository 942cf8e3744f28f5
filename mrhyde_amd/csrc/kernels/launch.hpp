// launch.hpp -- host-callable launchers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../common.hpp"
#include "device_types.hpp"

namespace mha {

// thermal_element.hip
bool thermal_element_supported(int dim, int order, int nq1);
void launch_thermal_element(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                            const ElemOut &out, hipStream_t stream);

// scatter.hip
void launch_scatter_local(const BlockDev &b, const double *local_J, const double *local_res, double *res,
                          double *crs_vals, int local_base, hipStream_t stream);
void launch_dbc_diag(const BlockDev &b, double *crs_vals, hipStream_t stream);
void launch_gather(const BlockDev &b, const double *vec, double *elem_vals, hipStream_t stream);

// workset_views.hip: physical basis / integration data of elements [e0, e0+ne)
struct WorksetViewsDev {
  double *basis = nullptr;       // [ne][n][nq]
  double *basis_grad = nullptr;  // [ne][n][nq][dim]
  double *wts = nullptr;         // [ne][nq]
  double *xyz[3] = {nullptr, nullptr, nullptr};  // [ne][nq] each
};
void launch_workset_views(const BlockDev &b, int e0, int ne, const WorksetViewsDev &v, hipStream_t stream);

// thermal_general_row_owner.hip: residual + Jacobian of general thermal elements in row-owner form (one launch)
bool thermal_general_row_owner_supported(int dim, int order, int nq1);
size_t thermal_general_row_owner_lds(int dim, int order, int nq1, const RowBlocksDev &rb);
void launch_thermal_general_row_owner(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                                      const RowBlocksDev &rb, const uint8_t *slot8, const int32_t *blk_rows,
                                      const double *gp1d, const int32_t *blk_hdr, long long *timing, const RowOut &out,
                                      int num_cus, hipStream_t stream);

// mass_apply.hip: applyMassMatrixFree (on the fly / stored dense / database / Sparse3DView) and Sparse3DView's constructor
void launch_mass_apply_free(const BlockDev &b, const VarLayoutDev &vl, const double *masswts, const double *x, double *y,
                            hipStream_t stream);
void launch_mass_apply_stored(const BlockDev &b, const VarLayoutDev &vl, const int32_t *index, const double *mass, int maxent,
                              const int32_t *nnz_row, const double *values, const int32_t *columns, const int32_t *pos_var,
                              const double *x, double *y, hipStream_t stream);
void launch_sparse3d_max(const double *dense, size_t total, unsigned long long *maxbits, hipStream_t stream);
void launch_sparse3d_count(const double *dense, size_t rows, int n, double tol, const unsigned long long *maxbits,
                           int32_t *nnz_row, int *maxent, hipStream_t stream);
void launch_sparse3d_fill(const double *dense, size_t rows, int n, double tol, const unsigned long long *maxbits, int maxent,
                          double *values, int32_t *columns, hipStream_t stream);

// var_views.hip: per-variable basis views (volume range e0.. when elem == null, else boundary entries), solution
// fields of the current workset, PhysicsInterface::fluxConditions, and a[i] = -a[i]
void launch_var_views(const BlockDev &b, const VarPointsDev &t, const int32_t *elem, const int32_t *side, int e0, int num,
                      const VarViewsDev &out, hipStream_t stream);
void launch_var_fields(const BlockDev &b, const TimeDev &tm, int e0, int num, int card, int var_off, int npts, int ncomp,
                       const double *basis, const double *grad, const double *div, const VarFieldsDev &out,
                       hipStream_t stream);
void launch_flux_condition(const BlockDev &b, const FuncDesc &flux, const int32_t *elem, int num, int card, int var_off,
                           int nqs, int ncomp, const double *wts, const double *const xyz[3], const double *const nrm[3],
                           const double *basis, double *res, hipStream_t stream);
void launch_negate(double *a, size_t n, hipStream_t stream);

// projection.hip
void launch_project_rhs(const BlockDev &b, const ProjectDev &p, const FuncDesc f[3], double *rhs, hipStream_t stream);
void launch_project_mass(const BlockDev &b, const ProjectDev &p, int lump, double *vals, hipStream_t stream);
void launch_fix_zero_rows(const BlockDev &b, double *vals, hipStream_t stream);
void launch_free_row_identity(const BlockDev &b, double *vals, hipStream_t stream);
void launch_interpolate_nodes(const BlockDev &b, const FuncDesc &f, int var_off, const int *vert_of_dof, double *initial,
                              hipStream_t stream);

// thermal_row_owner.hip
void launch_classify_affine(const BlockDev &b, uint8_t *flags, double tol, hipStream_t stream);
void launch_build_block_slots(const BlockDev &b, const RowBlocksDev &rb, void *bslot, int slot_bytes,
                              hipStream_t stream);
bool thermal_row_owner_supported(int dim, int order, int nq1);
size_t row_owner_jacobian_lds(const RowBlocksDev &rb, int n, int slot_bytes);
void launch_affine_geometry(const BlockDev &b, double *geo, hipStream_t stream);
// block_pattern.hip: block-major element records from the geometry cache; the matrix-core form of the row-owner Jacobian
void launch_build_erec2(int64_t total_records, int nsym, const int32_t *erec_elem, const double *geo, double *erec2,
                        hipStream_t stream);
void launch_block_pattern_jacobian(const BlockPatternDev &d, const RowOut &out, double su, double st, hipStream_t stream);
// geometry-database mode of the Jacobian: copies runs of CRS entries inside vals, 1 KB chunk by chunk (chunks: [n][4]
// ints = destination / 16 bytes, source entry of lane 0, first and one-past-last destination entry of the run)
void launch_replicate_runs(const int32_t *chunks, int nchunks, double *vals, hipStream_t stream);
void launch_build_erec(int dim, const RowBlocksDev &rb, const double *geo, double *erec, int total,
                       hipStream_t stream);
// K1: element-wise residual (-> res with atomics)
void launch_thermal_affine_element(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                                   const AffineDev &af, double *res, hipStream_t stream);
// thermal_affine_residual.hip: the same residual, one thread per element (sum-factorised through the point values)
bool thermal_affine_residual_supported(int dim, int order, int nq1);
// plan != nullptr: the workgroup-merged form (256 consecutive elements per workgroup meet in LDS: K1PlanDev)
void launch_thermal_affine_residual(int dim, int order, const BlockDev &b, const ThermalDev &ph, const double *geo,
                                    const AffineTables1D &tab, const K1PlanDev *plan, double *res, const double *max_abs_coord,
                                    hipStream_t stream);
// K2: row-owner Jacobian; scale_u = alpha_u*kappa, scale_t = alpha_t*rho*cp
void launch_row_owner_jacobian(int dim, int n, const RowBlocksDev &rb, const AffineDev &af, const RowOut &out,
                               double scale_u, double scale_t, hipStream_t stream);

// thermal_general.hip: general elements (non-affine geometry / per-ip coefficients)
void launch_build_elem_slot_map(const BlockDev &b, void *slot, int slot_bytes, hipStream_t stream);
void launch_thermal_general(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph, const AffineDev &af,
                            const void *slot, int slot_bytes, const ElemOut &out, hipStream_t stream);

// thermal_boundary.hip: boundary groups
bool thermal_boundary_supported(int n, int nqs);
void launch_boundary_views(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const BoundaryViewsDev &v,
                           hipStream_t stream);
void launch_thermal_boundary(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const TimeDev &tm,
                             const ElemOut &out, hipStream_t stream);

// porous_boundary.hip: porousMixed::boundaryResidual (weak Dirichlet on p)
void launch_porous_flux(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const VarLayoutDev &vl,
                        const TimeDev &tm, hipStream_t stream);
void launch_porous_boundary(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const VarLayoutDev &vl,
                            const ElemOut &out, hipStream_t stream);

// porous_element.hip: porousMixed volume terms, one thread per element, dense element arrays out
void launch_porous_element(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                           const ElemOut &out, hipStream_t stream);
// database mode on a uniform block: residual parts of every element from the common element matrix `uniform[0 .. n*n)`
// (dof order, alpha_u included; written by the dense kernel on element 0 just before); fills the point tables behind it
void launch_porous_uniform_points(const BlockDev &b, double *uniform, hipStream_t stream);
void launch_porous_uniform_residual(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                                    const ElemOut &out, double *uniform, hipStream_t stream);
// finishing pass of the direct form: one thread per row sums the (residual, diagonal) parts its incident elements left
// in the row's record part[nrows][2][2]; diagpos[row] = CRS position of the diagonal of a face row, -1 otherwise; fixed
// rows are zeroed when overwriting
void launch_porous_direct_finish(const BlockDev &b, const int32_t *inc_ptr, const int32_t *diagpos, const double *part,
                                 double *res, double *vals, int overwrite, hipStream_t stream);

// row_gather.hip: CRS rows summed from dense element matrices, no global atomics
void launch_row_gather(const BlockDev &b, const RowGatherDev &g, const double *local_J, const double *local_res,
                       double *res, double *vals, int overwrite, hipStream_t stream);

// swhdg_side.hip: shallowwaterHybridized side terms + derivatives, one thread per side point
void launch_swhdg_side(const SwhSideArgs &a, hipStream_t stream);

// swhdg_boundary.hip: shallowwaterHybridized::boundaryResidual, trace state given at the side points
void launch_swhdg_boundary(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const SwhBoundaryDev &sw,
                           const TimeDev &tm, const ElemOut &out, hipStream_t stream);

// swhdg_element.hip: HDG element blocks of shallowwaterHybridized (interior + trace unknowns), side part
void launch_swhdg_element(const BlockDev &b, const SideTablesDev &st, const SwhElementDev &a, const TimeDev &tm,
                          hipStream_t stream);

// condense.hip: batched static condensation of element blocks (one wavefront per element)
// newton.hip: vector kernels of the Newton driver (newton.hpp)
void launch_norm_inf(int64_t n, const double *v, unsigned long long *out_bits, hipStream_t stream);
void launch_axpy(int64_t n, double alpha, const double *x, double *y, hipStream_t stream);
void launch_dirichlet_lift(int64_t n, const uint8_t *fixed, const double *vals, double scalar, double *u, hipStream_t stream);
// swhdg_fused.hip: side + volume assembly + static condensation of the HDG element in one kernel
void launch_swhdg_fused(const BlockDev &b, const SideTablesDev &st, const SwhElementDev &a, const TimeDev &tm,
                        const PhysParamsDev &pp, const SwhFusedOut &o, hipStream_t stream);
void launch_condense(int n_int, int n_trace, int64_t nelem, const double *blocks, const double *res, double *schur,
                     double *gvec, double *du, int *singular, hipStream_t stream);

// subgrid.hip: loop state of the subgrid sub-iteration driver (SubGridDtN_Solver::nonlinearSolver)
void launch_subgrid_combine(int64_t nelem, int ni, int n, const int32_t *offsets, const double *local_J, const double *local_res,
                            double *blocks, double *res, int pass, double tol, double *rn0, double *scaled, int32_t *iters,
                            int32_t *active, hipStream_t stream);
void launch_subgrid_update(int64_t nelem, int ni, const int32_t *lids, const int32_t *offsets, const double *du,
                           const int32_t *active, double *u, hipStream_t stream);

// export.hip: pack / unpack-add of the shared-row Export(ADD) (export_plan.hpp)
void launch_export_pack(const double *src, const int32_t *idx, int64_t n, double *dst, hipStream_t stream);
void launch_export_unpack_add(const double *src, const int32_t *tgt, int64_t n, double *dst, hipStream_t stream);

// point_engine.hip: multi-variable blocks, any physics module stated as a point function
// slot: element-major CRS slot map (launch_build_elem_slot_map) or null for the column search
void launch_point_engine(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                         const ElemOut &out, const void *slot, int slot_bytes, hipStream_t stream);

}  // namespace mha
