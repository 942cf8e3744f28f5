/*
 * mrhyde_amd.h -- C ABI of the MI355X-native element-local assembly path.
 *
 * Drop-in boundary for MrHyDE's AssemblyManager / Workset / PhysicsBase hot
 * path.  Plain pointers and sizes only; every entry
 * point returns an int status (0 = MHA_OK) and sets mha_last_error(); nothing
 * throws across the boundary and nothing calls back into the caller.
 *
 * The reference has no FFI: its boundary is the C++ interface contract
 *   AssemblyManager::assembleJacRes / assembleRes   src/managers/assemblyManager.hpp:173-239
 *   Workset<EvalT> field views                       src/tools/workset.hpp:193-316
 *   PhysicsBase<EvalT>::volumeResidual() etc.        src/physics/physicsBase.hpp:59-85
 * Each entry point below cites the reference routine it replaces (file:line
 * under the reference tree).  INTEGRATION.md shows the adapter a MrHyDE
 * maintainer would add on the reference side.
 *
 * Conventions:
 *  - "host" pointers are read during the call and may be freed afterwards;
 *    "dev" pointers are HIP device pointers owned by the caller (the assembler
 *    never owns res / J, as in the reference).
 *  - LIDs are int32 (LO=int, src/preferences.hpp:40-47) and are consumed
 *    bit-exact; scalars are double (ScalarT).
 *  - the global residual receives -res.val(), the Jacobian +d res(row)/d u(col)
 *    (assemblyManager.cpp:4094,4128); rows with isFixedDOF are skipped (:4075,:4120).
 *  - one HIP stream per context; calls on one context are not thread-safe
 *    (the reference processes worksets sequentially, assemblyManager.cpp:2355).
 */
#ifndef MRHYDE_AMD_H
#define MRHYDE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHA_OK 0
#define MHA_ERR_INVALID 1      /* bad argument / unsupported configuration      */
#define MHA_ERR_STATE 2        /* call order violated (e.g. assemble before mesh) */
#define MHA_ERR_DEVICE 3       /* HIP runtime error (no GPU, launch failure, ...)  */
#define MHA_ERR_UNKNOWN_FIELD 4

typedef struct mha_context mha_context; /* one element block on one GPU        */

const char *mha_last_error(void);
const char *mha_version(void);
/* number of visible HIP devices, or 0 (never fails) */
int mha_device_count(void);

/* ---- block descriptor ---------------------------------------------------
 * replaces: GroupMetaData + Workset construction for one block
 *   src/tools/groupMetaData.hpp:23-129, src/managers/assemblyManager.cpp:272-712
 *   DiscretizationInterface::getBasis/getQuadrature  discretizationInterface.cpp:346-478 */
#define MHA_TOPO_QUAD4 4
#define MHA_TOPO_HEX8 8
#define MHA_BASIS_HGRAD 0   /* Basis_HGRAD_{QUAD,HEX}_Cn_FEM, equispaced   (discretizationInterface.cpp:354-371) */
#define MHA_BASIS_HVOL 1    /* Basis_HVOL_C0_FEM: order 0                   (:372-374)                             */
#define MHA_BASIS_HDIV 2    /* Basis_HDIV_{QUAD,HEX}_In_FEM: order 1        (:375-393)                             */
#define MHA_MAX_VARS 8

typedef struct mha_block_desc {
  int dimension;                 /* 2 or 3                                      */
  int topology;                  /* MHA_TOPO_QUAD4 / MHA_TOPO_HEX8              */
  int num_vars;                  /* thermal: 1 ("e"); porousMixed: 2 ("p","u");
                                    navierstokes: dim+1 ("ux","pr","uy"[,"uz"])  */
  int basis_type[MHA_MAX_VARS];  /* MHA_BASIS_*, in the module's myvars order    */
  int basis_order[MHA_MAX_VARS]; /* Discretization: order                       */
  int quadrature_degree;         /* Discretization: quadrature (0 => 2*max order,
                                    discretizationInterface.cpp:166)             */
  int workset_size;              /* Solver: workset size (100; <=0 => all,
                                    assemblyManager.cpp:326-332)                 */
  int device;                    /* HIP device ordinal                          */
} mha_block_desc;

int mha_block_create(const mha_block_desc *desc, mha_context **out);
void mha_block_destroy(mha_context *ctx);
/* hipStream_t passed as void*; NULL = the null stream */
int mha_set_stream(mha_context *ctx, void *hip_stream);

/* ---- mesh, maps, graph ---------------------------------------------------
 * replaces: createGroups' LID/node copies (assemblyManager.cpp:656-688),
 * wkset offsets (solverManager.cpp:856-982), createFixedDOFs (assemblyManager.cpp:185-265).
 * nodes[E][nnodes][dim] f64 in the cell topology's (shards) vertex order;
 * lids[E][n] int32; offsets[var][dof] flattened in variable order, values are
 * positions in the element's LID list; is_fixed[nrows] u8 (may be NULL).        */
int mha_set_mesh(mha_context *ctx, int num_elems, const double *nodes_host, const int32_t *lids_host,
                 const int32_t *offsets_host, int num_rows, const uint8_t *is_fixed_host);
/* replaces: OrientTools::modifyBasisByOrientation for lowest-order HDIV face dofs
 * (discretizationInterface.cpp:1021-1027,1055-1061): signs[E][n] int8 (+1/-1) per element and flattened
 * (variable, dof); the caller computes them from Intrepid2::Orientation (:2467).  NULL resets to +1.
 * Call after mha_set_mesh.                                                           */
int mha_set_orientation(mha_context *ctx, const int8_t *signs_host);
/* overlapped CRS graph of the caller (Tpetra local graph: rowptr[nrows+1], colind sorted
 * ascending per row).  Pass NULL/NULL to build it by the reference rule
 * (linearAlgebraInterface.cpp:218-229).                                          */
int mha_set_graph(mha_context *ctx, const int32_t *rowptr_host, const int32_t *colind_host);
int mha_get_graph_sizes(mha_context *ctx, int *num_rows, int64_t *nnz);
int mha_get_graph(mha_context *ctx, int32_t *rowptr_host, int32_t *colind_host);

/* ---- physics module -------------------------------------------------------
 * replaces: PhysicsImporter::import + thermal::defineFunctions
 *   src/physics/physicsImporter.cpp:48-204, src/physics/thermal.cpp:24-66
 * Coefficients are what FunctionManager::evaluate(name,"ip") would return
 * (functionManager.cpp:543-760): a constant, or a per-(elem,ip) array.            */
#define MHA_PHYSICS_THERMAL 1        /* src/physics/thermal.cpp: e (HGRAD)                                   */
#define MHA_PHYSICS_POROUS_MIXED 2   /* src/physics/porousMixed.cpp: p (HVOL 0), u (HDIV 1)                   */
#define MHA_PHYSICS_NAVIERSTOKES 3   /* src/physics/navierstokes.cpp: ux, pr, uy[, uz] (HGRAD)                */
#define MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED 4 /* src/physics/shallowwaterHybridized.cpp: H, Hux, Huy (HGRAD, 2-D):
                                                volumeResidual only; the side terms are mha_swhdg_side_terms */
int mha_physics_select(mha_context *ctx, int physics_id);
#define MHA_FUNC_CONSTANT 0
#define MHA_FUNC_IP_ARRAY 1     /* dev pointer to [E][numip] f64                    */
#define MHA_FUNC_SINPROD 2      /* amp * prod_d sin(freq[d]*x_d), evaluated at ip   */
#define MHA_FUNC_EXPRESSION 3   /* a deck string, see mha_set_function_expression    */
/* thermal: "thermal source","thermal diffusion","specific heat","density" (thermal.cpp:52-63);
 * porousMixed: "source","Kinv_xx","Kinv_yy","Kinv_zz","total_mobility" (porousMixed.cpp:141-151);
 * navierstokes: "source ux","source pr","source uy","source uz","density","viscosity"
 * (navierstokes.cpp:68-74); shallowwaterHybridized: "source H","source Hux","source Huy"
 * (shallowwaterHybridized.cpp:84-88); or a boundary
 * data function "Neumann e <sidename>" / "Dirichlet e <sidename>" (see boundary groups)    */
int mha_set_function(mha_context *ctx, const char *name, int kind, double amp, const double *freq3,
                     const double *ip_array_dev);

/* A function given as the string of the reference's input deck ("Functions:" / boundary-condition entries), e.g.
 * "8*(pi*pi)*sin(2*pi*x)*sin(2*pi*y)": compiled on the host, interpreted at the integration points on the device.
 * replaces: FunctionManager::addFunction + decomposeFunctions + evaluate for position/time expressions
 *   src/managers/functionManager.cpp:60-93, 95-540, 543-860, src/tools/interpreter.cpp.
 * Grammar: numbers; x y z t nx ny nz h pi (functionManager.cpp:21); + - * / ^, unary minus, < > <= >=, parentheses;
 * sin cos tan exp log abs sqrt sinh cosh (:22).  Solution fields, other named functions and the view reductions
 * max/min/mean/emax/emin/emean are rejected (MHA_ERR_INVALID).  `t` is the time given to mha_set_time.              */
int mha_set_function_expression(mha_context *ctx, const char *name, const char *expression);
/* syntax / vocabulary check of a deck string without a context (host only, no GPU needed): MHA_OK or MHA_ERR_INVALID */
int mha_check_expression(const char *expression);
/* replaces: Workset::setTime (src/tools/workset.hpp:127): the value of `t` in expressions                         */
int mha_set_time(mha_context *ctx, double time);

/* ---- time integration state -----------------------------------------------
 * replaces: Workset::setDeltat/setStage + butcher/BDF tables consumed by
 * computeSolnTransientSeeded  src/tools/workset.cpp:559-623.  Steady = never call.  */
int mha_set_time_integration(mha_context *ctx, int transient, int num_steps, int num_stages, int stage,
                             double deltat, const double *butcher_A_host, const double *butcher_b_host,
                             const double *bdf_wts_host);

/* ---- assembly --------------------------------------------------------------
 * replaces: AssemblyManager::assembleJacRes<EvalT>  assemblyManager.cpp:2150-2665
 *   (volume loop: performGather :3598, updateWorkset :6512, volumeResidual,
 *    scatter :4031) and assembleRes :2946-3151 (compute_jacobian = 0).
 * u_dev[nrows]; u_prev_dev[nrows][num_steps], u_stage_dev[nrows][num_stages] (transient only).
 * flags: MHA_ASSEMBLE_JACOBIAN (seedwhat = 1: residual and Jacobian; without it the ScalarT
 *   residual-only pass), MHA_ASSEMBLE_OVERWRITE (see below).
 * res_dev[nrows] and crs_vals_dev[nnz] are ACCUMULATED into: the caller zeroes them first, as
 * SolverManager does (solverManager.cpp:1528-1533).  With MHA_ASSEMBLE_OVERWRITE the call stores
 * instead, i.e. it fuses that zeroing (res_over->putScalar(0), J_over->setAllToScalar(0)) into the
 * assembly; use it when this call is the first contribution of the Newton step.
 * crs_vals_dev may be NULL iff the Jacobian is not requested.
 * path: MHA_PATH_AUTO picks the fastest valid kernel; the others force one.          */
#define MHA_ASSEMBLE_JACOBIAN 1
#define MHA_ASSEMBLE_OVERWRITE 2
/* scatter options of AssemblyManager::scatter (assemblyManager.cpp:4124-4133), reproduced as the reference has them:
 * ADJOINT (isAdjoint_): every column of a row receives res(elem,row).fastAccessDx(row); LUMP_MASS (lump_mass_): every
 * column's value is added to the row's diagonal entry (cols[col] = rowIndex).  Both go through the element matrices +
 * row gather (MHA_PATH_AUTO picks it; the fused fast paths do not carry the options).                             */
#define MHA_ASSEMBLE_ADJOINT 4
#define MHA_ASSEMBLE_LUMP_MASS 8
/* DETERMINISTIC: results do not depend on scheduling -- two calls with the same inputs give bit-identical res / crs_vals.
 * (The reference sums element contributions in element order on one thread, or with atomics in any order on a device
 * build, assemblyManager.cpp:4058-4061; the default fast paths here add a row's contributions in an order that depends
 * on wavefront timing, which moves the last bits.)  Available on the affine row-owner path (MHA_PATH_AUTO /
 * MHA_PATH_ROW_OWNER on affine elements with constant coefficients): the Jacobian rows are register sums in a fixed order
 * (kernels/block_pattern.hip) and the residual entries are summed pair by pair in pair order by the row's owner
 * (kernels/thermal_general_row_owner.hip, residual-only, ordered) instead of the thread-per-element kernel's atomics;
 * MHA_ERR_INVALID where that combination does not exist.  Slower: the residual costs ~1 ms more at config 2.          */
#define MHA_ASSEMBLE_DETERMINISTIC 16
#define MHA_PATH_AUTO 0
#define MHA_PATH_ELEMENT_ATOMIC 1 /* per-element kernel, atomic scatter (reference's
                                     fused "assembly insert Jac" with useAtomics)     */
#define MHA_PATH_ROW_OWNER 2      /* fused row-owner kernel, no global atomics           */
#define MHA_PATH_LOCAL_THEN_SCATTER 3 /* updateJac/updateRes then scatterJac/scatterRes   */
#define MHA_PATH_POINT_ENGINE 4   /* multi-variable kernel: point-level forward AD + B^T C B
                                     contraction, atomic scatter; the path of porousMixed and
                                     navierstokes, available to thermal as a cross-check    */
#define MHA_PATH_ROW_GATHER 5     /* element kernel -> dense element matrices (device scratch,
                                     E*n*n doubles) -> every CRS row summed by one wavefront from
                                     its incident elements: no global atomics; AUTO for
                                     porousMixed / navierstokes                                */
int mha_assemble_jacres(mha_context *ctx, int flags, int path, const double *u_dev,
                        const double *u_prev_dev, const double *u_stage_dev, double *res_dev,
                        double *crs_vals_dev);
/* replaces: updateJac / updateRes  assemblyManager.cpp:7412-7455, 7115-7152
 * local_J_dev[E][n][n] += res(e,r).dx(c), local_res_dev[E][n] -= res(e,r).val()      */
int mha_compute_local_jacres(mha_context *ctx, int compute_jacobian, const double *u_dev,
                             const double *u_prev_dev, const double *u_stage_dev, double *local_J_dev,
                             double *local_res_dev);
/* replaces: getMass / getWeightedMass  assemblyManager.cpp:7776-7840, 7847-7925: dense element mass matrices
 * local_mass_dev[E][n][n] += sum_q basis_v(i,q) . basis_v(j,q) wts(q) * masswts[v] at (offsets(v,i), offsets(v,j)),
 * every variable of the block (HGRAD / HVOL values, HDIV vector values); masswts_host[num_vars] or NULL (= 1).
 * mha_scatter_local turns them into CRS values.                                                          */
int mha_get_mass(mha_context *ctx, const double *masswts_host, double *local_mass_dev);
/* ---- mass storage and the matrix-free mass apply ----------------------------------------------------------------
 * Sparse3DView (src/tools/sparse3DView.hpp:21-161): compressed storage of dense element matrices dense_dev[E][n][n]:
 * the entries with |a| / max|a| > tol, row by row in column order.  Device-resident; mha_sparse3d_views returns
 * values_dev[E][n][maxent] (f64), columns_dev[E][n][maxent] (int32, local column = position in the element's LID list)
 * and nnz_row_dev[E][n] (int32) -- getValues / getColumns / getNNZPerRow; mha_sparse3d_size = size().  Stateless.   */
typedef struct mha_sparse3d mha_sparse3d;
int mha_sparse3d_create(int64_t num_elems, int n, const double *dense_dev, double tol, void *hip_stream, mha_sparse3d **out);
int mha_sparse3d_views(mha_sparse3d *s, int *maxent, double **values_dev, int32_t **columns_dev, int32_t **nnz_row_dev);
int mha_sparse3d_size(mha_sparse3d *s, int64_t *total_nnz);
void mha_sparse3d_destroy(mha_sparse3d *s);
/* The basis database (AssemblyManager::identifyVolumetricDatabase, assemblyManager.cpp:4314-4467): elements that share
 * their geometry share one set of stored data (basis, mass).  The reference scans earlier representatives and accepts
 * relative differences below "database TOL" (1e-10) in measure and Jacobians; here the match is EXACT -- bit-identical
 * vertex offsets from the element's first vertex and identical orientation signs (found by hashing: O(E), not O(E U)),
 * which is the subset of the reference's matches that substitutes nothing: results are unchanged to rounding.
 * Representatives are numbered in order of first appearance, as in the reference.  index_host[E] (basis_index),
 * first_users_host[num_unique] (element id of every representative); either may be NULL.                          */
int mha_database_build(mha_context *ctx, int *num_unique);
int mha_database_get(mha_context *ctx, int32_t *index_host, int32_t *first_users_host);
/* AssemblyManager::applyMassMatrixFree (assemblyManager.cpp:1582-1778): y_dev += M x_dev with M block diagonal by
 * variable, never assembled.  masswts_host[num_vars] or NULL (= 1; physics->mass_wts).  mode:
 *   MHA_MASS_ON_THE_FLY   basis and weights recomputed per element (the !storeMass branch :1607-1672); mass_dev, sparse NULL
 *   MHA_MASS_LOCAL        mass_dev[E][n][n], e.g. from mha_get_mass (:1760-1772)
 *   MHA_MASS_DATABASE     mass_dev[num_unique][n][n] of the database representatives + basis_index (:1730-1755)
 *   MHA_MASS_DATABASE_SPARSE  the same in Sparse3DView storage (:1690-1726; "sparse mass format")
 * masswts applies to MHA_MASS_ON_THE_FLY only (stored masses carry their weights, as in the reference).            */
#define MHA_MASS_ON_THE_FLY 0
#define MHA_MASS_LOCAL 1
#define MHA_MASS_DATABASE 2
#define MHA_MASS_DATABASE_SPARSE 3
int mha_apply_mass_matrix_free(mha_context *ctx, int mode, const double *masswts_host, const double *mass_dev,
                               mha_sparse3d *sparse, const double *x_dev, double *y_dev);
/* replaces: scatterJac / scatterRes  assemblyManager.cpp:3882-3935, 3943-3978          */
int mha_scatter_local(mha_context *ctx, const double *local_J_dev, const double *local_res_dev,
                      double *res_dev, double *crs_vals_dev);
/* replaces: dofConstraints -> updateJacDBC  assemblyManager.cpp:3158-3187, 1166-1179   */
int mha_apply_dbc_diag(mha_context *ctx, double *crs_vals_dev);
/* replaces: performGather  assemblyManager.cpp:3598-3643: out[E][n] (dof order = basis order) */
int mha_gather(mha_context *ctx, const double *vec_dev, double *elem_vals_dev);

/* ---- the nonlinear-solve protocol around the assembler --------------------------------------------------------------
 * replaces: the loop of SolverManager::nonlinearSolver (src/managers/solverManager.cpp:1465-1709) around assembleRes /
 * assembleJacRes, and the strong-Dirichlet lifting of SolverManager::setDirichlet (:1876-1957).  The linear solver, the
 * Export / Import between ranks and the all-reduce of the norm stay with the caller (out of scope: SURVEY.md section 2);
 * the driver is a state machine stepped with plain calls, no callbacks:
 *   mha_newton_residual : zero res_dev; assembleRes (AUTOTUNE: the residual-only ScalarT path first, :1538-1552) incl. the
 *                         boundary groups -> res_dev = -res.val()                    [caller: Export(ADD) of res_dev]
 *   mha_newton_norm     : |res_dev|_inf of this rank's rows (:1571)                  [caller: all-reduce max]
 *   mha_newton_decide   : iteration 0 fixes resnorm_first; resnorm_scaled = resnorm / resnorm_first (:1575-1581);
 *                         backtracking when allowed and the scaled norm exceeds 1.1: alpha halved, u -= alpha du (the last
 *                         update), no solve (:1591-1616); relative tolerance (scaled < nl_tol, or resnorm < 1e-100) or
 *                         absolute tolerance ends the loop (:1617-1632); *action = MHA_NEWTON_SOLVE / _BACKTRACKED / _DONE
 *   mha_newton_jacobian : only when a solve is needed (:1638-1641): zero res_dev, assembleJacRes (+ boundary groups) with
 *                         the CRS values overwritten, unit diagonal on the fixed rows  [caller: Export(ADD) of the matrix,
 *                         solve J du = res, Import du]
 *   mha_newton_update   : u += alpha du with alpha = 1 (:1656-1672); closes the iteration (NLiter++, :1681-1685)
 *   mha_newton_step     : residual + norm + decide (+ jacobian) in one call for a single-rank caller.
 * autotune = 0: assembleJacRes at once, as the reference does for adjoint / multiscale runs (:1540-1547).
 * mha_newton_state: iteration count, norms, alpha, status (1: the iteration limit was reached without convergence).
 * mha_dirichlet_lift: u[row] = fixed_soln_dev[row] (or scalar_value when fixed_soln_dev is NULL) on the rows the mesh
 * marks fixed -- the lifted solution carries the strong boundary condition (:1905-1935).                              */
#define MHA_NEWTON_SOLVE 1
#define MHA_NEWTON_BACKTRACKED 2
#define MHA_NEWTON_DONE 3
typedef struct mha_newton mha_newton;
int mha_newton_create(mha_context *ctx, int max_iter, double nl_tol, double nl_abs_tol, int use_relative, int use_absolute,
                      int allow_backtracking, int autotune, mha_newton **out);
void mha_newton_destroy(mha_newton *nw);
int mha_newton_reset(mha_newton *nw);
int mha_newton_residual(mha_newton *nw, const double *u_dev, const double *u_prev_dev, const double *u_stage_dev,
                        double *res_dev);
int mha_newton_norm(mha_newton *nw, const double *res_dev, double *resnorm);
int mha_newton_decide(mha_newton *nw, double resnorm, double *u_dev, int *action);
int mha_newton_jacobian(mha_newton *nw, const double *u_dev, const double *u_prev_dev, const double *u_stage_dev,
                        double *res_dev, double *crs_vals_dev);
int mha_newton_update(mha_newton *nw, double *u_dev, const double *du_dev);
int mha_newton_step(mha_newton *nw, double *u_dev, const double *u_prev_dev, const double *u_stage_dev, double *res_dev,
                    double *crs_vals_dev, int *action);
int mha_newton_state(const mha_newton *nw, int *iteration, double *resnorm, double *resnorm_scaled, double *resnorm_first,
                     double *alpha, int *status);
int mha_dirichlet_lift(mha_context *ctx, double *u_dev, const double *fixed_soln_dev, double scalar_value);

/* ---- workset views -----------------------------------------------------------
 * replaces: updateWorkset aliasing the group's stored views (assemblyManager.cpp:6512-6596)
 * and Group::computeBasis -> getPhysicalVolumetricBasis / getPhysicalIntegrationData
 *   src/tools/group.cpp:134-243, discretizationInterface.cpp:732-776, 898-1127.
 * mha_workset_update evaluates the views for workset `index` (elements
 * [index*ws, min((index+1)*ws, E))) on the device; mha_workset_view returns a
 * device pointer + extents (LayoutRight) valid until the next update.
 * names: "basis" (numElem,n,numip,1)  "basis_grad" (numElem,n,numip,dim)
 *        "wts" (numElem,numip)  "x","y","z" (numElem,numip)  "LIDs" (numElem,n) int32
 *        "offsets" (numvars, maxdof) int32
 * per variable (any block; <var> is the module's variable name, "var<k>" without a module):
 *   "basis <var>" (numElem,card,numip,ncomp)   getBasis(var)      workset.hpp:241   ncomp = dim for HDIV (Piola
 *                                              value J phi / detJ, orientation sign applied), 1 for HGRAD / HVOL
 *   "basis_grad <var>" (numElem,card,numip,dim) getBasisGrad(var) workset.hpp:249   HGRAD
 *   "basis_div <var>" (numElem,card,numip)      getBasisDiv(var)  workset.hpp:270   HDIV
 *   (discretizationInterface.cpp:898-1127; "basis"/"basis_grad" without a name: single-variable HGRAD blocks)
 * mha_workset_compute_solution: Workset::computeSoln on the current workset (workset.cpp:1017-1190) from
 *   u / u_prev / u_stage with the transient seeding of :589-623; afterwards the solution fields are views
 *   (numElem,numip) under the reference's names -- getSolutionField (workset.hpp:229): "<var>", "<var>_t",
 *   "grad(<var>)[x|y|z]" (HGRAD), "<var>[x|y|z]", "<var>_t[x|y|z]", "div(<var>)" (HDIV).  The views hold the
 *   fields' values; their derivative arrays never leave the chip (d field / d u_j = alpha_u * basis).
 * mha_workset_compute_residual: resetResidual + volumeResidual on the current workset (what assembleJacRes does
 *   between updateWorkset and scatter, assemblyManager.cpp:2426-2440); afterwards "res" (numElem,n) = res(e,i).val()
 *   and, with compute_jacobian, "res.dx" (numElem,n,n) = res(e,i).fastAccessDx(j) -- getResidual (workset.hpp:193),
 *   i, j = positions in the element's LID list.
 * unknown names: MHA_ERR_UNKNOWN_FIELD (the reference prints and continues,
 * workset.cpp:1576-1577).                                                             */
int mha_num_worksets(mha_context *ctx);
int mha_workset_update(mha_context *ctx, int index);
int mha_workset_view(mha_context *ctx, const char *name, void **dev_ptr, int64_t extents[4], int *rank);
int mha_workset_compute_solution(mha_context *ctx, const double *u_dev, const double *u_prev_dev,
                                 const double *u_stage_dev);
int mha_workset_compute_residual(mha_context *ctx, int compute_jacobian, const double *u_dev,
                                 const double *u_prev_dev, const double *u_stage_dev);

/* ---- boundary groups -----------------------------------------------------------
 * replaces: BoundaryGroup (src/tools/boundaryGroup.hpp:23-186, boundaryGroup.cpp:25-178), the
 * boundary-group loop of assembleJacRes (assemblyManager.cpp:2518-2638: updateWorksetBoundary :5646-5710,
 * performBoundaryGather :3650-3700, boundaryResidual, scatter), side integration data
 * (discretizationInterface.cpp:1608-1790, 1810-1955) and thermal::boundaryResidual
 * (src/physics/thermal.cpp:172-281).
 * A group = the (element, local side) entries of one side set that carry one boundary-condition
 * type for the variable; local side ids are the cell topology's (shards): quad edges
 * {0,1},{1,2},{2,3},{3,0}; hex faces {0,1,5,4},{1,2,6,5},{2,3,7,6},{0,4,7,3},{0,3,2,1},{4,5,6,7}.
 * The boundary data is the function "Neumann e <sidename>" / "Dirichlet e <sidename>"
 * (thermal.cpp:217,238) registered with mha_set_function; an MHA_FUNC_IP_ARRAY for it is a
 * dev array [num_sides][side numip].  Strong Dirichlet needs no group (is_fixed rows).
 * mha_assemble_boundary ACCUMULATES all groups into res / crs_vals (call after the volume
 * assembly of the same Newton step; MHA_ASSEMBLE_OVERWRITE is rejected).
 * mha_boundary_update / mha_boundary_view expose the side data the reference keeps in the
 * workset: "wts side" (num,numip) "x","y","z" (num,numip) "n[x]","n[y]","n[z]" (num,numip)
 * "basis side" (num,n,numip,1) "basis_grad side" (num,n,numip,dim); per variable on any block
 * (getBasisSide(var) / getBasisGradSide(var), workset.hpp:281-293; getPhysicalBoundaryBasis,
 * discretizationInterface.cpp:1840-1950): "basis side <var>" (num,card,numip,ncomp), "basis_grad side <var>".
 * mha_add_flux_group: the generic "Flux" condition of PhysicsInterface::fluxConditions
 * (src/interfaces/physicsInterface.cpp:1702-1762) for one variable of any block on one side set:
 * res(elem, off(dof)) += -flux * wts side * basis side(elem,dof,pt,0) with flux = the function
 * "Flux <var> <sidename>" at the side points (constant, closed form, expression in x,y,z,t,nx,ny,nz, or a dev
 * array [num_sides][side numip]); no Jacobian contribution.  mha_assemble_boundary runs it with the other groups. */
#define MHA_BC_NEUMANN 1
#define MHA_BC_WEAK_DIRICHLET 2
#define MHA_BC_FLUX 3
#define MHA_BC_INTERFACE 5 /* thermal: the weak-Dirichlet branch with the trace, the function "aux e <sidename>", as data (thermal.cpp:227-243) */
#define MHA_BC_DIRICHLET 4 /* strong condition (mha_add_dirichlet_group): nothing in mha_assemble_boundary, see mha_set_dirichlet */
/* shallowwaterHybridized side types (bcs(H_num, side): "interface", "Far-field", "Slip",
 * shallowwaterHybridized.cpp:286-300, 612-620): the group's entries are element sides (all four sides of every
 * element for the HDG interior problem); the trace state comes from the functions "aux H <sidename>",
 * "aux Hux <sidename>", "aux Huy <sidename>" (constants or dev arrays [num_sides][side numip]), the far-field state
 * from "Far-field H <sidename>", ... (:648-653).  mha_assemble_boundary adds boundaryResidual (:190-263) and its
 * derivative with respect to the interior unknowns; settings "Roe-like stabilization" (default 1; 0 = max EV), "g". */
#define MHA_BC_SWH_INTERFACE 10
#define MHA_BC_SWH_FARFIELD 11
#define MHA_BC_SWH_SLIP 12
int mha_add_boundary_group(mha_context *ctx, const char *sidename, int bc_type, int num_sides,
                           const int32_t *elem_ids_host, const int32_t *local_side_ids_host, int *group_id);
int mha_add_flux_group(mha_context *ctx, const char *sidename, const char *varname, int num_sides,
                       const int32_t *elem_ids_host, const int32_t *local_side_ids_host, int *group_id);
int mha_clear_boundary_groups(mha_context *ctx);

/* ---- L2-projection systems of initial and strong-Dirichlet data -------------------------------------------------
 * The caller solves mass * x = rhs (the reference hands both to its linear solver, solverManager.cpp:1876-1957);
 * building them is the assembly manager's part.  mass_vals_dev / rhs_dev are ACCUMULATED into (zero them first).
 *
 * mha_set_initial replaces AssemblyManager::setInitial(set, rhs, mass, useadjoint, lumpmass, scale)
 *   (src/managers/assemblyManager.cpp:1185-1305) with getInitial(project = true) (:7632-7682), getMass (:7776-7840) and
 *   PhysicsInterface::getInitial (src/interfaces/physicsInterface.cpp:898-958):
 *     rhs[LIDs(e, off_v(i))] += sum_q init_v(e,q) . basis_v(e,i,q) wts(e,q)            (no sign change, fixed rows too)
 *     mass(LIDs(e, off_v(i)), LIDs(e, off_v(j))) += sum_q basis_v(e,i,q) . basis_v(e,j,q) wts(e,q); with lump_mass
 *     every entry goes to the diagonal; afterwards rows with sum |a| < 1e-14 get a one on the diagonal (:1284-1302).
 *   Data = the functions "initial <var>" (HGRAD / HVOL) or "initial <var>[x]", "[y]", "[z]" (HDIV) registered with
 *   mha_set_function / mha_set_function_expression (MHA_FUNC_IP_ARRAY: [E][numip]).
 * mha_set_initial_nodal replaces AssemblyManager::setInitial(set, initial, useadjoint) (:1830-1850) with
 *   getInitial(project = false) (:7683-7727): initial[LIDs(e, off_v(k))] = "initial <var>" at vertex k of element e
 *   (replaceLocalValue).  As in the reference this is for HGRAD variables of order 1 only: MHA_ERR_INVALID otherwise.
 *   (MHA_FUNC_IP_ARRAY here: [E][vertices per element].)
 * mha_add_dirichlet_group + mha_set_dirichlet replace AssemblyManager::setDirichlet (:1855-1943) with
 *   getDirichletBoundary (:6288-6350), getMassBoundary (:6360-6425) and PhysicsInterface::getDirichlet
 *   (physicsInterface.cpp:1065-1083): for every entry of the groups whose variable carries "Dirichlet" on the side set,
 *   and only in rows with is_fixed,
 *     rhs[row] += sum_pt D(k,pt) basis side(k,i,pt,0) wts side(k,pt)        HGRAD / HVOL
 *              += sum_pt D(k,pt) sum_c basis side(k,i,pt,c) n_c(k,pt) wts   HDIV
 *     mass(row, LIDs(e, off_v(j))) += sum_pt basis side_i basis side_j wts  (HDIV: sum_c b_i,c n_c b_j,c n_c wts, :6400-6408)
 *   with D = the function "Dirichlet <var> <sidename>" at the side points; every row that is NOT fixed and is touched
 *   by an element gets a one on its diagonal (replaceValues, :1920-1938).  With lump_mass the reference adds the row total
 *   to the column of the LAST entry of the element's LID list, not to the diagonal (the `cols[0]` its summing loop leaves
 *   behind, :1888-1897); reproduced as it is.                                                                        */
int mha_add_dirichlet_group(mha_context *ctx, const char *sidename, const char *varname, int num_sides,
                            const int32_t *elem_ids_host, const int32_t *local_side_ids_host, int *group_id);
int mha_set_initial(mha_context *ctx, int lump_mass, double *rhs_dev, double *mass_vals_dev);
int mha_set_initial_nodal(mha_context *ctx, double *initial_dev);
int mha_set_dirichlet(mha_context *ctx, int lump_mass, double *rhs_dev, double *mass_vals_dev);

int mha_num_boundary_groups(mha_context *ctx);
int mha_assemble_boundary(mha_context *ctx, int flags, const double *u_dev, const double *u_prev_dev,
                          const double *u_stage_dev, double *res_dev, double *crs_vals_dev);
/* computeFlux of the block's module on one boundary group: the workset's flux view (elem, auxvar, pt).
 * replaces: AssemblyManager::computeFlux -> PhysicsInterface::computeFlux -> <module>::computeFlux
 *   (src/managers/assemblyManager.hpp:573-727; called by SubGridDtN_Solver::updateFlux, subgridDtN_solver.cpp:1579)
 *   thermal (src/physics/thermal.cpp:288-347):       flux = (10 / h) kappa (lambda - T) + kappa grad T . n, lambda = the
 *                                                    function "aux e <sidename>" at the side points, h = side element size
 *   porousMixed (src/physics/porousMixed.cpp:440-500): flux = u . n (HDIV side basis, Piola, orientation sign)
 *   navierstokes (src/physics/navierstokes.cpp:1016-1018): empty in the reference: zeros
 *   shallowwaterHybridized: mha_swhdg_side_terms / mha_swhdg_element_blocks (trace rows).
 * flux_dev[num][nqs]; optional derivative arrays (what the AD type of the reference's view carries):
 * dflux_du_dev[num][nqs][n] with respect to the element's unknowns in flattened (variable, dof) order (time-integration
 * factor alpha_u included), dflux_daux_dev[num][nqs] with respect to the aux value at the point.                     */
int mha_compute_flux(mha_context *ctx, int group_id, const double *u_dev, const double *u_prev_dev, const double *u_stage_dev,
                     double *flux_dev, double *dflux_du_dev, double *dflux_daux_dev);
int mha_boundary_update(mha_context *ctx, int group_id);
int mha_boundary_view(mha_context *ctx, int group_id, const char *name, void **dev_ptr, int64_t extents[4],
                      int *rank);
/* scalar settings of the physics module; thermal: "form_param" (thermal.cpp:35, default 1:
 * symmetric Nitsche; -1 the non-symmetric variant), "include advection" (thermal.cpp:39, default 0; 1 adds
 * (b . grad e, v) with the functions "bx","by","bz", thermal.cpp:59-61, 150-160: the block then assembles on the
 * point engine, MHA_PATH_ROW_OWNER is refused); navierstokes: "useSUPG", "usePSPG"
 * (navierstokes.cpp:45-46) and "fix_uz_offsets": the reference scatters the 3-D uz momentum
 * block through uy's offsets (navierstokes.cpp:688); 0 (default) reproduces that, 1 uses uz's;
 * shallowwaterHybridized: "g" (shallowwaterHybridized.cpp:72, default 9.81).                     */
int mha_set_physics_parameter(mha_context *ctx, const char *name, double value);

/* ---- shallowwaterHybridized side terms ---------------------------------------------------
 * replaces: computeFluxVector(on_side) :409-480, eigendecompFluxJacobian :765-823,
 * computeStabilizationTerm :487-588, computeBoundaryTerm :595-758 and computeFlux :270-368 of
 * src/physics/shallowwaterHybridized.cpp, evaluated at npts side integration points (2-D; state
 * order H, Hux, Huy).  S = interior state ("H","Hux","Huy" side fields), Shat = trace state
 * ("aux H", ...), normals[npts][2], Sinf[npts][3] = "Far-field <var> <side>" (Far-field only).
 * Outputs (any may be NULL): fluxvec[npts][3][2] = fluxes_side; term[npts][3] = stab_bound_side
 * (stabilisation term for MHA_SWH_INTERFACE, boundary term otherwise); iflux[npts][3] = what
 * computeFlux stores in wkset->flux; d_iflux_dS / d_iflux_dShat [npts][3][3] = its derivatives
 * (the SFad part of the reference's result), by forward AD on the device.  All pointers device.
 * Stateless: no context; hip_stream may be NULL.                                              */
#define MHA_SWH_INTERFACE 0
#define MHA_SWH_FARFIELD 1
#define MHA_SWH_SLIP 2
int mha_swhdg_side_terms(int side_type, int roe_stabilization, double g, int64_t npts, const double *S_dev,
                         const double *Shat_dev, const double *normals_dev, const double *Sinf_dev,
                         double *fluxvec_dev, double *term_dev, double *iflux_dev, double *d_iflux_dS_dev,
                         double *d_iflux_dShat_dev, void *hip_stream);
/* The HDG element, side part: residual and derivative blocks of the 12 interior unknowns (H, Hux, Huy, HGRAD
 * order 1, flattened (variable, dof)) and the 24 trace unknowns of every element of a shallowwaterHybridized block.
 * replaces: boundaryResidual :190-263 on the four sides (interior rows), computeFlux :270-368 integrated against the
 * trace basis as SubGridDtN_Solver::updateFlux does (src/subgrid/subgridDtN_solver.cpp:1583-1601, trace rows), with
 * Basis_HFACE_QUAD_In_FEM of degree 1 (src/tools/Intrepid2_HFACE_QUAD_In_FEMdef.hpp:84-196: per edge the two linear
 * Lagrange functions of the edge's reference coordinate; edges left x=-1, bottom y=-1, right x=+1, top y=+1).
 * lambda_dev[E][24]: variable, edge, function.  side_types_dev[E][4] u8 in shards side order (bottom, right, top,
 * left): MHA_SWH_*; NULL = all interface.  farfield_host[3] (Far-field sides).  Outputs, element-major, rows/columns
 * = 12 interior then 24 traces: res_dev[E][36] = -res.val(), blocks_dev[E][36][36] = res(r).dx(c) (stored); either may
 * be NULL.  The volume part of the interior block comes from mha_compute_local_jacres.                           */
int mha_swhdg_element_blocks(mha_context *ctx, const double *u_dev, const double *u_prev_dev, const double *u_stage_dev,
                             const double *lambda_dev, const uint8_t *side_types_dev, const double *farfield_host,
                             double *res_dev, double *blocks_dev);
/* The subgrid sub-iteration loop: SubGridDtN_Solver::nonlinearSolver (src/subgrid/subgridDtN_solver.cpp:909-1041) with
 * its assembleJacobianResidual (:681-903) and element-local direct solve, for a shallowwaterHybridized block whose interior
 * unknowns are element-local (one HDG element per subgrid: every LID belongs to one element -- checked).  Per element,
 * with the trace lambda held fixed: pass 0 records resnorm_initial = |res_u|_inf and sets the scaled norm to 1 (0 if the
 * residual vanishes); while scaled > tol (sub_NLtol) and passes < max_iter (sub_maxNLiter): du = A_uu^-1 r_u,
 * u += du, reassemble, scaled = |res_u|_inf / resnorm_initial.  u_dev is updated in place.  Outputs: iters_dev[E] =
 * assemblies the element's loop performed (the reference's `iter`), resnorm_scaled_dev[E], and -- from one closing
 * assembly at the final state -- schur_dev[E][24][24], gvec_dev[E][24] as mha_batched_condense defines them (either may
 * be NULL: no closing pass), *num_singular_dev = singular interior blocks met, CUMULATIVE over all max_iter + 1
 * condensation passes (an element that stays singular is counted once per pass, also after it has left its loop):
 * zero means no pass met one; it is not a count of elements.  Everything is enqueued on the context's
 * stream: NO host synchronisation and NO allocation inside (workspace_dev of mha_swhdg_subgrid_workspace_bytes bytes is
 * the caller's; side tables are built on the first call).  Elements that have converged are still swept by the remaining
 * passes and ignored: the uniform schedule is what keeps the host out of the loop.                                   */
int mha_swhdg_subgrid_workspace_bytes(mha_context *ctx, int64_t *bytes);
int mha_swhdg_subgrid_solve(mha_context *ctx, double *u_dev, const double *u_prev_dev, const double *u_stage_dev,
                            const double *lambda_dev, const uint8_t *side_types_dev, const double *farfield_host,
                            int max_iter, double tol, void *workspace_dev, int64_t workspace_bytes, double *schur_dev,
                            double *gvec_dev, int32_t *iters_dev, double *resnorm_scaled_dev, int32_t *num_singular_dev);
/* The element step of the HDG subgrid in ONE kernel: side terms (mha_swhdg_element_blocks) + volume terms
 * (shallowwaterHybridized::volumeResidual, src/physics/shallowwaterHybridized.cpp:113-184) + static condensation
 * (mha_batched_condense), one wavefront per element -- the [36 x 36] block of SubGridDtN_Solver::assembleJacobianResidual
 * (src/subgrid/subgridDtN_solver.cpp:681-903) and updateFlux (:1542-1616) never leaves the chip.  Outputs as
 * mha_batched_condense defines them: schur_dev[E][24][24], gvec_dev[E][24], du_dev[E][12] (flattened (variable, dof));
 * any may be NULL, not all.  *num_singular_dev += singular interior blocks (device counter, may be NULL).  Enqueued on
 * the context's stream: no synchronisation, no allocation (side tables are built on the first call).  Sources given as
 * deck strings are refused (MHA_ERR_INVALID): the unfused entry points take those.  mha_swhdg_subgrid_solve runs this
 * kernel once per pass with the loop bookkeeping and sol += du inside (MHA_SUBGRID_UNFUSED=1 keeps the four-kernel
 * pipeline, the independent implementation the tests compare with).                                                 */
int mha_swhdg_condensed_element(mha_context *ctx, const double *u_dev, const double *u_prev_dev, const double *u_stage_dev,
                                const double *lambda_dev, const uint8_t *side_types_dev, const double *farfield_host,
                                double *schur_dev, double *gvec_dev, double *du_dev, int32_t *num_singular_dev);
/* Batched static condensation of element blocks: eliminates the n_int interior unknowns of every element.
 * replaces: the element-local direct solve of the subgrid solver and its forward sensitivities d u / d lambda
 * (SubGridDtN_Solver, src/subgrid/subgridDtN_solver.cpp:681-903, 1542-1616) in Schur-complement form.
 * blocks_dev[E][n][n], res_dev[E][n] with n = n_int + n_trace, interior unknowns first (the layout of
 * mha_swhdg_element_blocks; add the volume block of mha_compute_local_jacres to the interior part first);
 * outputs: schur_dev[E][n_trace][n_trace] = A_ll - A_lu A_uu^-1 A_ul, gvec_dev[E][n_trace] = r_l - A_lu A_uu^-1 r_u,
 * du_dev[E][n_int] = A_uu^-1 r_u (any may be NULL).  Gauss-Jordan with partial pivoting, one wavefront per
 * element; n_int <= 32, n_int + n_trace + 1 <= 64.  Singular interior blocks: counted into *num_singular_host
 * if given, MHA_ERR_INVALID otherwise.  Stateless; synchronises the stream.                                  */
int mha_batched_condense(int n_int, int n_trace, int64_t num_elems, const double *blocks_dev, const double *res_dev,
                         double *schur_dev, double *gvec_dev, double *du_dev, int *num_singular_host, void *hip_stream);
/* Scatter of dense element blocks through an arbitrary LID map: the flux -> trace scatter of the HDG caller (condensed
 * blocks [E][24][24] and flux vectors [E][24] of mha_batched_condense into the macro trace system).
 * replaces: SubGridDtN_Solver::updateFlux's scatter and the macro assembly's sumIntoValues for those blocks
 * (src/subgrid/subgridDtN_solver.cpp:1542-1616, src/managers/assemblyManager.cpp:4031-4145).
 * create: lids_host[E][n] = global row of unknown t of element e; the CRS graph of the target (colind ascending inside a
 * row), or NULL/NULL to have the every-dof-of-an-element-couples graph built (read it back with _graph); fixed_host
 * [num_rows] or NULL.  apply: vals[rowptr[r] + slot] (+)= sum blocks[e][t][s], res[r] (+)= sum vec[e][t], one wavefront
 * per CRS row, no global atomics, results independent of scheduling; overwrite = 1 stores (fixed rows: zeros),
 * 0 accumulates (fixed rows untouched).  blocks+vals or vec+res may be NULL.                                    */
typedef struct mha_scatter_plan mha_scatter_plan;
int mha_scatter_plan_create(int n, int64_t num_elems, int64_t num_rows, const int32_t *lids_host,
                            const int32_t *rowptr_host, const int32_t *colind_host, const uint8_t *fixed_host,
                            mha_scatter_plan **out);
int mha_scatter_plan_nnz(const mha_scatter_plan *p, int64_t *nnz);
int mha_scatter_plan_graph(const mha_scatter_plan *p, int32_t *rowptr_host, int32_t *colind_host);
int mha_scatter_plan_apply(const mha_scatter_plan *p, const double *blocks_dev, const double *vec_dev, double *res_dev,
                           double *vals_dev, int overwrite, void *hip_stream);
void mha_scatter_plan_destroy(mha_scatter_plan *p);
/* L[npts][3][3], lam[npts][3], R[npts][3][3] (row-major) of the normal flux Jacobian at Shat */
int mha_swhdg_eigendecomp(double g, int64_t npts, const double *Shat_dev, const double *normals_dev,
                          double *L_dev, double *lam_dev, double *R_dev, void *hip_stream);

/* ---- structured mesh helper (input generation, not part of the hot path) ------
 * 2-D order 1 = SimpleMeshManager_Rectangle (src/tools/simplemeshmanager.hpp:639-675)
 * with offsets {0,1,3,2} (discretizationInterface.cpp:302).                            */
int mha_mesh_sizes(int dim, int order, const int *ncell, int *nverts, int *nelem, int64_t *ndof);
int mha_mesh_structured(int dim, int order, const int *ncell, const double *lo, const double *hi,
                        double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                        uint8_t *boundary_dof);

/* The same for a block of several variables (types MHA_BASIS_*; HGRAD orders dividing the largest, HVOL order 0, HDIV
 * order 1): vertices, cell -> vertex map, the subcell-major dof map (nodes of the finest HGRAD lattice, then cells, then
 * faces by direction), offsets [n_tot] (position of every variable's dofs in an element's LID list, variables
 * concatenated), orientation signs [E][n_tot] (lowest-order HDIV: -+1, others 1; what Intrepid2's
 * modifyBasisByOrientation applies, discretizationInterface.cpp:957-1055), side_mask [ndof] (bit 2d / 2d+1: the dof
 * lies on the low / high boundary in direction d) and dof_var [ndof].  orient / side_mask / dof_var may be NULL.   */
int mha_mesh_multi_sizes(int dim, const int *ncell, int nvars, const int *types, const int *orders, int *nverts,
                         int *nelem, int *n_tot, int64_t *ndof);
int mha_mesh_structured_multi(int dim, const int *ncell, const double *lo, const double *hi, int nvars, const int *types,
                              const int *orders, double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                              int8_t *orient, uint8_t *side_mask, int32_t *dof_var);

/* ---- row partition (host only; no GPU needed) -------------------------------------
 * The fused kernel is row-owner: each CRS row is produced by exactly one workgroup that
 * visits all elements incident to its rows -- the scatter of the reference
 * (assemblyManager.cpp:4063-4144) without atomics.  These entry points expose the partition
 * for inspection: blocks of rows (ascending) and the elements each block touches.
 * caps[4] = {elements per Morton chunk, max CRS entries, max rows, max elements} or NULL.   */
typedef struct mha_row_partition mha_row_partition;
int mha_row_partition_build(int dim, int num_elems, int dofs_per_elem, int num_rows, const double *nodes_host,
                            const int32_t *lids_host, const int32_t *rowptr_host, const int *caps,
                            mha_row_partition **out);
int mha_row_partition_sizes(const mha_row_partition *p, int *num_blocks, int64_t *num_rows, int64_t *num_elems,
                            int *max_rows, int *max_elems, int *max_entries);
int mha_row_partition_get(const mha_row_partition *p, int32_t *row_ptr, int32_t *rows, int32_t *elem_ptr,
                          int32_t *elems);
void mha_row_partition_destroy(mha_row_partition *p);

/* ---- Export(ADD) of shared-DOF rows between element-block shards (one process per GPU) -----------------------------
 * Replaces LinearAlgebraInterface::exportMatrixFromOverlapped / exportVectorFromOverlapped
 * (src/interfaces/linearAlgebraInterface.hpp:296-337: Tpetra::Export, combine mode ADD): every rank assembles its
 * overlapped CRS values / residual; the ranks that do not own a shared row send its partial values to the owner, which
 * adds them.  The shared rows are explicit lists agreed at setup (any partition: HGRAD planes between slabs, HDIV face
 * dofs, HDG trace rows, unequal slabs), CSR-style per neighbour (all *_ptr have num_neighbors + 1 entries, host arrays):
 *   send_val_index: entries of my value array that go to the neighbour;  send_row_index: entries of my residual;
 *   recv_val_target: for every value arriving from it, the entry of MY value array it is added to, or -1 (an off-rank
 *                    column of one of my rows: stays in the receive buffer);  recv_row_target: likewise for the residual.
 * No indices travel: both sides list a shared row's entries in the same (global id) order.
 * mha_export_pack / mha_export_unpack_add are the device halves (the caller moves the buffers, e.g. with
 * torch.distributed P2P, whose "nccl" backend is RCCL); mha_export_add does pack + ncclSend / ncclRecv with every
 * neighbour in one group + unpack on this library's own RCCL communicator (mha_comm_*; librccl.so is loaded on first
 * use).  Neighbours are unpacked in the order given: the sum is reproducible.
 * nnz / nrows are the sizes of the value array and the residual the lists index: every index is range-checked at plan
 * creation and the receive targets of one neighbour must be distinct (MHA_ERR_INVALID otherwise).  vals_dev == NULL in
 * pack / unpack_add / export_add means a residual-only exchange (assembleRes, solverManager.cpp:1552-1556): the value
 * segments are skipped, on the wire too; res_dev == NULL likewise.                                                     */
typedef struct mha_export_plan mha_export_plan;
typedef struct mha_comm mha_comm;
int mha_export_plan_create(int num_neighbors, const int32_t *neighbor_ranks, const int64_t *send_val_ptr,
                           const int32_t *send_val_index, const int64_t *send_row_ptr, const int32_t *send_row_index,
                           const int64_t *recv_val_ptr, const int32_t *recv_val_target, const int64_t *recv_row_ptr,
                           const int32_t *recv_row_target, int64_t nnz, int64_t nrows, mha_export_plan **out);
void mha_export_plan_destroy(mha_export_plan *p);
int mha_export_pack(const mha_export_plan *p, const double *vals_dev, const double *res_dev, void *hip_stream);
int mha_export_unpack_add(const mha_export_plan *p, double *vals_dev, double *res_dev, void *hip_stream);
/* buffers of neighbour k (device pointers; layout [values | residual entries]) and the bytes one exchange sends */
int mha_export_buffers(const mha_export_plan *p, int k, double **send_dev, int64_t *send_count, double **recv_dev,
                       int64_t *recv_count);
int mha_export_bytes_on_wire(const mha_export_plan *p, int64_t *bytes);
int mha_comm_unique_id(char id_out[128]);                       /* rank 0; broadcast the 128 bytes to the others */
int mha_comm_create(int num_ranks, int rank, const char id[128], mha_comm **out); /* collective; current HIP device */
void mha_comm_destroy(mha_comm *c);
int mha_export_add(const mha_export_plan *p, mha_comm *c, double *vals_dev, double *res_dev, void *hip_stream);

/* ---- introspection for bench / tests ---------------------------------------------
 * keys: "num_elems","num_rows","nnz","dofs_per_elem","num_ip","workset_size","last_path",
 *       "row_blocks","num_affine_elems","row_block_max_rows","row_block_max_elems",
 *       "row_block_max_acc","row_owner_lds_bytes","block_patterns","block_pattern_roles",
 *       "block_pattern_blocks","block_pattern_mfma"
 * (the block_pattern keys are 0 when the matrix-core form of the row-owner Jacobian is not in use: MHA_K2=blocks,
 * or a mesh whose row blocks share too few assembly patterns)
 * round 3: "row_owner_kind" (1 affine kernels, 2 general-element kernel, of the last assembly), "general_row_blocks",
 *   "affine_shapes"          distinct geometry records (shape part, bit for bit) of the block's affine elements -- the
 *                            geometry database behind the affine path (reference: identifyVolumetricDatabase,
 *                            assemblyManager.cpp:4314-4467, with exact matching)
 *   "jacobian_database_mode" 1: the last affine row-owner Jacobian was computed for one row block per assembly pattern
 *                            and replicated (one shape, overwriting assembly); bit-identical to the full kernel
 *   "porous_direct"          last porousMixed assembly behind MHA_PATH_ROW_GATHER: 0 dense element arrays + row gather,
 *                            1 element threads stored into the CRS, 2 database mode (representative rows + replication) */
int mha_get_info(mha_context *ctx, const char *key, int64_t *value);
/* average device time (ms) of the last assembly's kernels, measured with HIP events on the
 * context's stream; valid after mha_set_timing(ctx,1).                                  */
int mha_set_timing(mha_context *ctx, int enable);
int mha_get_last_kernel_ms(mha_context *ctx, double *ms);

#ifdef __cplusplus
}
#endif
#endif
