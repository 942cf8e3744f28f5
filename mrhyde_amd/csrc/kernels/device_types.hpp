// device_types.hpp -- plain structs handed to the HIP kernels by value (host + device).
#pragma once
#include <cstdint>

namespace mha {

constexpr int kMaxStages = 8;
constexpr int kMaxSteps = 8;
// cached geometry record of an affine element: detJ*J^{-1}J^{-T} (NSYM, padded to 6), detJ, J (9), centroid (3)
constexpr int kGeoRec = 20, kGeoDet = 6, kGeoJ = 7, kGeoXc = 16;

// What FunctionManager::evaluate(name,"ip") yields for one named function
// (reference: src/managers/functionManager.cpp:543-760): constant, per-ip data, or a closed form.
// postfix program of MHA_FUNC_EXPRESSION (expression.hpp compiles, device_math.hpp interprets)
enum ExprOp : int32_t {
  EXPR_END = 0, EXPR_CONST,
  EXPR_X, EXPR_Y, EXPR_Z, EXPR_T, EXPR_NX, EXPR_NY, EXPR_NZ, EXPR_H, EXPR_PI,  // operands
  EXPR_ADD, EXPR_SUB, EXPR_MUL, EXPR_DIV, EXPR_POW, EXPR_LT, EXPR_GT, EXPR_LE, EXPR_GE,  // binary
  EXPR_NEG, EXPR_SIN, EXPR_COS, EXPR_TAN, EXPR_EXP, EXPR_LOG, EXPR_ABS, EXPR_SQRT, EXPR_SINH, EXPR_COSH,  // unary
  // solution fields at the point (an index follows, as after EXPR_CONST): slot of the point engine's field array
  // (physics_points.hpp: per variable value, d/dx, d/dy(, d/dz) ...) and of the time-derivative array ("e_t")
  EXPR_FIELD, EXPR_FIELD_T
};
constexpr int kExprStack = 12;

struct FuncDesc {
  int kind = 0;            // MHA_FUNC_*
  double amp = 0.0;
  double freq[3] = {0, 0, 0};
  const double *ip = nullptr;  // [E][numip] device
  const int32_t *code = nullptr;    // MHA_FUNC_EXPRESSION: postfix program (device)
  const double *consts = nullptr;   //                      its constants (device)
  double t = 0.0;                   //                      current time (Workset::setTime)
  int uses_fields = 0;              //                      the program reads solution fields (EXPR_FIELD*): Dual evaluation
};

// Device view of one element block.
struct BlockDev {
  int dim = 0, nelem = 0, nrows = 0, n = 0, nq = 0, nnodes = 0;
  int e_begin = 0, e_count = 0;      // element range a launch works on (a workset, or all)
  const double *nodes = nullptr;     // [E][nnodes][dim]
  const int32_t *lids = nullptr;     // [E][n]
  const int32_t *offsets = nullptr;  // [n]  (variable 0)
  const uint8_t *fixed = nullptr;    // [nrows] or null
  const int32_t *rowptr = nullptr;   // [nrows+1]
  const int32_t *colind = nullptr;   // [nnz]
  // reference tables (device copies)
  const double *ref_basis = nullptr;   // [n][nq]
  const double *ref_grad = nullptr;    // [n][nq][dim]
  const double *ref_wts = nullptr;     // [nq]
  const double *nodeval = nullptr;     // [nnodes][nq]
  const double *nodegrad = nullptr;    // [nnodes][nq][dim]
};

// Time-integration coefficients of Workset::computeSolnTransientSeeded, seedwhat == 1
// (reference: src/tools/workset.cpp:589-623).
struct TimeDev {
  int transient = 0, nsteps = 0, nstages = 0, stage = 0;
  double alpha_u = 1.0, alpha_t = 0.0, timewt = 0.0, dt = 1.0;
  double stage_ratio[kMaxStages] = {0};  // A(stage,s)/b(s), s < stage
  double bdf[kMaxSteps + 1] = {0};
  const double *u = nullptr;        // [nrows] current (stage) solution
  const double *u_prev = nullptr;   // [nrows][nsteps]
  const double *u_stage = nullptr;  // [nrows][nstages]
};

struct ThermalDev {
  FuncDesc source, diff, cp, rho;
  TimeDev time;
};

// Where an element kernel puts its results.
struct ElemOut {
  int compute_jacobian = 1;
  int local_store = 0;          // 1: local_J / local_res are stored (scratch of the row-gather path), 0: accumulated
  int local_base = 0;           // element id stored at local_J[0] / local_res[0]
  int local_dof_order = 0;      // row-gather scratch only (porousMixed element kernel): rows and columns of local_J and the
                                // entries of local_res in (variable, dof) order instead of LID-position order
  double *local_J = nullptr;    // [E][n][n]   (updateJac convention, +=)
  double *local_res = nullptr;  // [E][n]      (updateRes convention, -=)
  double *res = nullptr;        // [nrows]     atomic scatter of -res.val()
  double *crs_vals = nullptr;   // [nnz]       atomic scatter of +res.dx()
  // porousMixed direct form (kernels/porous_element.hip): the element thread stores its matrix entries straight into
  // the CRS through the element-major slot map -- on a conforming lowest-order mixed mesh two elements share exactly one
  // dof, so every entry but the diagonal of a face row has ONE contributor -- and leaves its residual entry and its
  // diagonal part of every row in that row's record direct_part[nrows][2][2] (slot direct_side[e][dof] = 0 or 1: which
  // of the row's incident elements this one is) for the finishing pass over the rows
  double *direct_part = nullptr;
  const uint8_t *direct_side = nullptr; // [E][n] (dof order)
  const int32_t *direct_elist = nullptr;    // direct form on a LIST of elements (e_count entries) instead of a range
  double *direct_uniform = nullptr;         // database mode: [n*n] element matrix of the uniform block (dof order, alpha_u included),
                                            // then [nq][dim] point offsets from an element's first vertex, then [nq] weights w_q
  int direct_axis_aligned = 0;              // database mode: every element is the same axis-aligned box (checked by the host):
                                            // a closed-form source amp prod sin(freq_d x_d) needs 2 sines per direction, not 8 x dim
  int direct_res_only = 0;                  // 1: the lean build: residual parts only (rows' records), no matrix arithmetic
  const uint8_t *direct_jacflag = nullptr;  // [E] or null: 0 = the element's matrix entries are not stored (its rows are
                                            // replicated from representative rows: database mode), residual parts always are
  double *direct_vals = nullptr;        // CRS values (null: residual only)
  const uint8_t *direct_slot = nullptr; // [E][n][n] position of column LIDs[e][j] inside row LIDs[e][i] (LID-position order)
  int direct_overwrite = 0;
};

// Row-block partition on the device (row_blocks.hpp).
struct RowBlocksDev {
  int num_blocks = 0;
  const int32_t *row_ptr = nullptr, *rows = nullptr, *row_off = nullptr, *acc_size = nullptr;
  const int32_t *elem_ptr = nullptr, *elems = nullptr;
  const int32_t *pair_ptr = nullptr;
  const uint32_t *pairs = nullptr;       // local_row << 16 | local_elem << 8 | LID slot
  const int32_t *pair_off = nullptr, *row_base = nullptr, *row_len = nullptr, *emask = nullptr, *epbase = nullptr;
  const int64_t *slot_ptr = nullptr;     // [nb+1] byte offset of the block's slot table (16-byte aligned)
  const int32_t *seg_ptr = nullptr, *seg_acc = nullptr, *seg_base = nullptr, *seg_len = nullptr;
  const int32_t *block_list = nullptr;  // blocks this launch handles (null = all)
  int list_len = 0;
  int lds_rows = 0, lds_elems = 0, lds_acc = 0, lds_pairs = 0, lds_segs = 0;  // LDS carve sizes (maxima over the partition)
};

// Data of the affine fast path: reference stiffness / mass tables in LID-slot space and the
// element -> CRS slot map.
struct AffineDev {
  const double *khat = nullptr;    // [NSYM+1][n*n]: sum_q w (d_a N_i d_b N_j + sym), last = sum_q w N_i N_j
  const double *phi1d = nullptr;   // [order+1][nq1]
  const double *dphi1d = nullptr;  // [order+1][nq1]
  const double *gw1d = nullptr;    // [nq1]
  const double *gp1d = nullptr;    // [nq1]
  const void *slot = nullptr;      // block-major [pair][n] (blocks padded to 16 B): position of column LIDs[e][j] inside the pair's CRS row
  const double *geo = nullptr;     // [E][kGeoRec] cached element geometry (affine elements)
  const double *erec = nullptr;    // block-major [touched element][8]: geometric factors + ownership data
  const uint16_t *pair_off16 = nullptr;  // block-major [pair]: accumulator offset of the pair's row
  const int *slot_pair = nullptr;        // [2*ceil(n/2)]: LID slots paired by co-ownership (-1 = none), K2's lane layout
  int slot_bytes = 1;              // 1 (uint8) or 2 (uint16)
};

// 1-D tables of the thread-per-element residual kernel (kernels/thermal_affine_residual.hip), passed by value: as many
// integration points per direction as dofs (m = order + 1 <= 5).
struct AffineTables1D {
  double phi[25] = {0};   // [m][m]  phi_i(xi_q)
  double dcol[25] = {0};  // [m][m]  collocation derivative: f'(xi_q) = sum_q' dcol[q][q'] f(xi_q')
  double gw[5] = {0}, gp[5] = {0};
};

// Plan of the workgroup-merged residual kernel (kernels/thermal_affine_residual.hip), made by the host at setup:
// workgroup g takes the elements wg_elems[256 g .. 256 g + 256) (the last group repeats the last element),
// wg_rows[wg_row_ptr[g] .. wg_row_ptr[g+1]) are the distinct rows their dofs touch, ascending, and
// loc[(g * n + ib) * 256 + t] is the position in that list of dof ib (BASIS order) of the group's element t.
#ifndef MHA_K1_THREADS
#define MHA_K1_THREADS 256
#endif
constexpr int kK1PlanThreads = MHA_K1_THREADS;  // elements (= threads) of a workgroup
struct K1PlanDev {
  const int32_t *wg_row_ptr = nullptr;
  const int32_t *wg_rows = nullptr;
  const int32_t *wg_elems = nullptr;
  const uint16_t *loc = nullptr;
  int max_rows = 0, num_elems = 0;
  int axis_aligned = 0;  // every element's J is diagonal (checked on the cached geometry records)
  // geometry database of the affine path (the reference's identifyVolumetricDatabase idea, exact matching): the shape
  // part of the geometry record -- detJ J^-1 J^-T, detJ, J: 16 doubles -- once per DISTINCT shape, an index per element
  // (a uniform mesh has one shape: the kernel reads 4 + 32 bytes per element instead of 160); the centroid stays per element
  const double *shape = nullptr;       // [num_shapes][16]
  const int32_t *shape_idx = nullptr;  // [E]
  int num_shapes = 0;
};

// Side reference tables on the device (ref_tables.hpp: SideTables).
struct SideTablesDev {
  int nsides = 0, nqs = 0;
  const double *wts = nullptr, *tanU = nullptr, *tanV = nullptr;
  const double *ip = nullptr;  // [ns][nqs][dim] side points in cell reference coordinates
  const double *basis = nullptr, *grad = nullptr, *nodeval = nullptr, *nodegrad = nullptr;
};

// One boundary group: (element, local side) entries that share a side name and a boundary-condition type
// (reference: src/tools/boundaryGroup.hpp; wkset->sidename, bcs(var,side), thermal.cpp:188-216).
struct BoundaryDev {
  int num = 0;
  const int32_t *elem = nullptr, *side = nullptr;
  int bc_type = 0;       // MHA_BC_*
  FuncDesc data;         // "Neumann e <side>" or "Dirichlet e <side>" at the side ip: ip array is [num][nqs]
  FuncDesc diff;         // "thermal diffusion" at the side ip (constant or closed form)
  double form_param = 1.0;
  // computeFlux (src/physics/thermal.cpp:288-347, porousMixed.cpp:440-500): wkset->flux(elem, auxvar, pt) of the group's
  // entries and its derivative arrays; null = the launch is a boundaryResidual
  double *flux = nullptr;        // [num][nqs]
  double *dflux_du = nullptr;    // [num][nqs][n]: d flux / d u_j, j in flattened (variable, dof) order (may be null)
  double *dflux_daux = nullptr;  // [num][nqs]:    d flux / d (aux value at the point) (may be null)
};

// Side views of one boundary group (getPhysicalBoundaryIntegrationData / getPhysicalBoundaryBasis).
struct BoundaryViewsDev {
  double *wts = nullptr;                          // [num][nqs]
  double *xyz[3] = {nullptr, nullptr, nullptr};   // [num][nqs]
  double *nrm[3] = {nullptr, nullptr, nullptr};   // [num][nqs] unit outward normals
  double *basis = nullptr, *basis_grad = nullptr; // [num][n][nqs], [num][n][nqs][dim]
};

// ---- multi-variable blocks (kernels/point_engine.hip) -------------------------------------------------------------
constexpr int kMaxVars = 8, kMaxSlots = 24, kMaxFuncs = 8;

// Variables of a block and their "slots": the quantities of a basis function that enter a weak form --
// HGRAD: value, d/dx, d/dy(, d/dz); HVOL: value; HDIV: the vector components, then the divergence.
struct VarLayoutDev {
  int nvars = 0, n_tot = 0, ns_tot = 0, nq = 0;
  int type[kMaxVars] = {0}, card[kMaxVars] = {0}, nslot[kMaxVars] = {0};
  int cardpad[kMaxVars] = {0};          // card rounded up to a multiple of 4 (table row length)
  int varptr[kMaxVars + 1] = {0}, slotptr[kMaxVars + 1] = {0};
  int table_off[kMaxVars] = {0};        // offset (doubles) of the variable's slot table inside `tables`
  int tables_size = 0;                  // doubles; variables with the same (type, order) share one table
  const double *tables = nullptr;       // per distinct basis: [nq][nslot][cardpad] reference slot values, dof fastest
  const int8_t *orient = nullptr;       // [E][n_tot] basis signs (modifyBasisByOrientation, lowest order) or null
};

// Reference values of ONE variable's basis at sets of points (kernels/var_views.hip): the volume integration points
// (one set) or the side integration points (one set per local side).
struct VarPointsDev {
  int type = 0, card = 0, npts = 0;
  const double *val = nullptr;       // [set][card][npts][ncomp], ncomp = dim for HDIV, 1 otherwise
  const double *grad = nullptr;      // [set][card][npts][dim]   (HGRAD)
  const double *div = nullptr;       // [set][card][npts]        (HDIV)
  const double *nodegrad = nullptr;  // [set][nnodes][npts][dim] geometry basis gradients at the points
  const int8_t *orient = nullptr;    // [E][n_tot] basis signs or null
  int n_tot = 0, var_off = 0;        // dofs per element of the block; first dof of the variable
};
struct VarViewsDev {
  double *basis = nullptr;  // [num][card][npts][ncomp]
  double *grad = nullptr;   // [num][card][npts][dim]
  double *div = nullptr;    // [num][card][npts]
};
// solution fields of one variable at the points ([num][npts] each): value components, time derivative, gradient, divergence
struct VarFieldsDev {
  double *val[3] = {nullptr, nullptr, nullptr}, *dot[3] = {nullptr, nullptr, nullptr};
  double *grad[3] = {nullptr, nullptr, nullptr}, *div = nullptr;
};

// One variable's stored views on a list of elements or (element, side) entries, for the L2-projection systems of
// initial and Dirichlet data (kernels/projection.hip)
struct ProjectDev {
  int num = 0, card = 0, var_off = 0, np = 0, ncomp = 1;
  int e0 = 0;                       // first element of a contiguous range (elem == null)
  const int32_t *elem = nullptr;    // [num] element of each boundary entry
  int fixed_only = 0;               // setDirichlet: only rows with isFixedDOF
  int normal_trace = 0;             // HDIV on a side: data * (basis . n), mass of the normal components
  const double *wts = nullptr;      // [num][np]
  const double *xyz[3] = {nullptr, nullptr, nullptr};  // [num][np]
  const double *nrm[3] = {nullptr, nullptr, nullptr};  // [num][np] (sides)
  const double *basis = nullptr;    // [num][card][np][ncomp]
};

// What a physics module's point function reads besides the fields: its named functions and scalar settings.
struct PhysParamsDev {
  int physics = 0;
  FuncDesc f[kMaxFuncs];
  double p[8] = {0};
};

// Row -> (element, LID position) incidences + the element-major slot map (kernels/row_gather.hip).
struct RowGatherDev {
  const int32_t *inc_ptr = nullptr, *inc_elem = nullptr, *inc_pos = nullptr;
  const void *slot = nullptr;
  int slot_bytes = 1, max_row = 0;
  // scatter options of the reference (assemblyManager.cpp:4124-4133): isAdjoint_ takes res(row).dx(row) for every column
  // of the row; lump_mass_ sends every column's value to the diagonal entry (cols[col] = rowIndex)
  int adjoint = 0, lump_mass = 0;
  const int32_t *inc_dof = nullptr;  // non-null: the element arrays are in (variable, dof) order (ElemOut::local_dof_order);
                                     // inc_dof[k] = flattened dof index of incidence k (inc_pos[k] is its LID position),
                                     // BlockDev::offsets maps a column's dof index to its position
};

// shallowwaterHybridized side terms at npts side integration points (kernels/swhdg_side.hip); state order H, Hux, Huy
struct SwhSideArgs {
  int64_t npts = 0;
  int side_type = 0, roe = 1;  // MHA_SWH_*; Roe-like (1) or max-eigenvalue (0) stabilisation
  double g = 9.81;
  const double *S = nullptr, *Shat = nullptr, *normals = nullptr, *Sinf = nullptr;  // [npts][3], [npts][3], [npts][2], [npts][3]
  double *fluxvec = nullptr;   // [npts][3][2]  F(Shat)
  double *term = nullptr;      // [npts][3]     stabilisation term (interface) or boundary term
  double *iflux = nullptr;     // [npts][3]     wkset->flux of computeFlux
  double *d_dS = nullptr, *d_dShat = nullptr;  // [npts][3][3] derivatives of iflux
  double *L = nullptr, *lam = nullptr, *R = nullptr;  // eigendecomposition at Shat, [npts][3][3], [npts][3], [npts][3][3]
};

// shallowwaterHybridized::boundaryResidual on a boundary group (kernels/swhdg_boundary.hip)
struct SwhBoundaryDev {
  int side_type = 0, roe = 1;   // MHA_SWH_*; Roe-like (1) or max-eigenvalue (0) stabilisation
  double g = 9.81;
  FuncDesc aux[3];              // trace state "aux H|Hux|Huy <side>" at the side points ([num][nqs] arrays or constants)
  FuncDesc farfield[3];         // "Far-field H|Hux|Huy <side>"
};

// HDG element of shallowwaterHybridized, side part (kernels/swhdg_element.hip)
struct SwhElementDev {
  const double *lambda = nullptr;        // [E][24] trace unknowns: variable, HFACE edge (left, bottom, right, top), function
  const uint8_t *side_types = nullptr;   // [E][4] in shards side order: MHA_SWH_*; null = all interface
  double farfield[3] = {0, 0, 0};
  double g = 9.81;
  int roe = 1;
  double *res = nullptr;                 // [E][36]     -res.val()
  double *blocks = nullptr;              // [E][36][36] res(r).dx(c), stored
};

// Outputs and loop state of the fused HDG element step (kernels/swhdg_fused.hip); every pointer may be null.
struct SwhFusedOut {
  double *schur = nullptr;    // [E][24][24] S = A_ll - A_lu A_uu^-1 A_ul
  double *gvec = nullptr;     // [E][24]     g = r_l - A_lu A_uu^-1 r_u
  double *du = nullptr;       // [E][12]     A_uu^-1 r_u, flattened (variable, dof)
  int *singular = nullptr;    // += 1 per element whose interior block is singular
  double *update_u = nullptr; // [nrows]: sol += du for the elements still in their loop (active, or all when active is null)
  int pass = -1;              // >= 0: bookkeeping of nonlinearSolver's loop for this pass (subgrid.hip)
  double tol = 0.0;
  double *rn0 = nullptr, *scaled = nullptr;
  int32_t *iters = nullptr, *active = nullptr;
};

// Row blocks keyed by assembly pattern (block_pattern.hpp) for the matrix-core row-owner Jacobian.
struct BlockPatternDev {
  int num_wgs = 0, max_w_doubles = 0;
  int max_rec_doubles = 0;             // doubles of the largest block's element records ((T + 1) * 8): LDS image size
  int dbg = 0;                         // profiling / cross-check aid (env MHA_BP_DBG): 1 plain-load form of every part, 2 no stores, 4 no products
  const double *erec2 = nullptr;       // role-major, block-major element records [T + 1][8]
  const int32_t *rowbase = nullptr;    // role-major, block-major CRS offsets of the owned rows [R]
  const double *w = nullptr;           // LDS images of the roles
  const int32_t *role = nullptr, *seg = nullptr, *wg_seg_ptr = nullptr, *part_ptr = nullptr, *part_hdr = nullptr, *part_lane = nullptr;
  const int32_t *wg_seg_ptr_img = nullptr;  // segments of the image roles' kernel
  bool has_direct = true, has_image = false;
  const int32_t *chunk_tab = nullptr;  // image roles: 64-entry chunks of a block's LDS image (block_pattern.hpp)
  long long nnz = 0;                   // CRS entries (the kernel addresses them with 32-bit byte offsets: nnz < 2^28)
  long long *timing = nullptr;         // profiling aid (env MHA_BP_TIMING): [num_wgs][16 waves][8] wall-clock stamps (10 ns)
};

// Destination of the row-owner kernels.
struct RowOut {
  double *res = nullptr;
  double *vals = nullptr;
  int overwrite = 0;         // 1: store (fuses the caller's zeroing), 0: accumulate
  int compute_jacobian = 1;
  int ordered = 0;           // general row-owner kernel, residual-only: pair sums in pair order (deterministic mode)
};

}  // namespace mha
