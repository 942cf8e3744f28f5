// assembly_manager.hpp -- host orchestration of one element block on one GPU.
//
// Plays the role of AssemblyManager for the volume terms of a block (reference:
// src/managers/assemblyManager.hpp:173-239, src/managers/assemblyManager.cpp:2150-2665):
// owns the block's mesh/map copies on the device, the reference tables, the physics module, the
// function table and the workset; assembleJacRes drives gather -> residual/Jacobian -> scatter.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "physics.hpp"
#include "ref_tables.hpp"
#include "row_blocks.hpp"
#include "block_pattern.hpp"
#include "workset.hpp"

namespace mha {

class AssemblyManager {
 public:
  explicit AssemblyManager(const mha_block_desc &desc);
  ~AssemblyManager();

  void setStream(hipStream_t s) { stream_ = s; wkset_.stream = s; }
  hipStream_t stream() const { return stream_; }
  // strong-Dirichlet lifting: u[row] = vals[row] (or `scalar` when vals is null) on the fixed rows
  // (SolverManager::setDirichlet, solverManager.cpp:1876-1957)
  void dirichletLift(double *u, const double *vals, double scalar);
  // modules whose volume term only exists as a point function (multi-variable blocks; thermal with advection)
  bool engineOnly() const { return physics_id_ != MHA_PHYSICS_THERMAL || (physics_ && physics_->pointEngineOnly()); }
  void setMesh(int nelem, const double *nodes, const int32_t *lids, const int32_t *offsets, int nrows,
               const uint8_t *fixed);
  void setOrientation(const int8_t *signs);
  void setGraph(const int32_t *rowptr, const int32_t *colind);
  void selectPhysics(int physics_id);
  void setFunction(const std::string &name, int kind, double amp, const double *freq3, const double *ip_dev);
  void setFunctionExpression(const std::string &name, const std::string &text) { functions_.addExpression(name, text); }
  void setTime(double t) { functions_.setTime(t); wkset_.setTime(t); }
  void setTimeIntegration(int transient, int nsteps, int nstages, int stage, double dt, const double *A,
                          const double *b, const double *bdf);

  void assembleJacRes(int flags, int path, const double *u, const double *u_prev,
                      const double *u_stage, double *res, double *crs_vals);
  void computeLocalJacRes(int compute_jacobian, const double *u, const double *u_prev, const double *u_stage,
                          double *local_J, double *local_res);
  void getMass(const double *masswts, double *local_mass);
  // basis database with exact-geometry matching, matrix-free mass apply (assemblyManager.cpp:4314-4467, 1582-1778)
  int databaseBuild();
  void databaseGet(int32_t *index, int32_t *first_users) const;
  void applyMassMatrixFree(int mode, const double *masswts, const double *mass, int maxent, const int32_t *nnz_row,
                           const double *values, const int32_t *columns, const double *x, double *y);
  void swhdgElementBlocks(const double *u, const double *u_prev, const double *u_stage, const double *lambda,
                          const uint8_t *side_types, const double *farfield, double *res, double *blocks);
  // SubGridDtN_Solver::nonlinearSolver for HDG elements with element-local interior unknowns (subgridDtN_solver.cpp:909-1041)
  size_t subgridWorkspaceBytes() const;
  // fused element step of the HDG subgrid (side + volume assembly + static condensation in one kernel)
  bool swhdgFusedUsable() const;
  void swhdgCondensedElement(const double *u, const double *u_prev, const double *u_stage, const double *lambda,
                             const uint8_t *side_types, const double *farfield, SwhFusedOut out);
  void subgridSolve(double *u, const double *u_prev, const double *u_stage, const double *lambda, const uint8_t *side_types,
                    const double *farfield, int max_iter, double tol, void *workspace, size_t workspace_bytes, double *schur,
                    double *gvec, int32_t *iters, double *resnorm_scaled, int32_t *num_singular);
  void scatterLocal(const double *local_J, const double *local_res, double *res, double *crs_vals);
  // boundary groups (reference: src/tools/boundaryGroup.hpp, assemblyManager.cpp:2518-2638)
  int addBoundaryGroup(const std::string &sidename, int bc_type, int num, const int32_t *elem_ids,
                       const int32_t *side_ids);
  // "Flux" condition of one variable on a side set (PhysicsInterface::fluxConditions, physicsInterface.cpp:1702-1762)
  int addFluxGroup(const std::string &sidename, const std::string &varname, int num, const int32_t *elem_ids,
                   const int32_t *side_ids);
  int addDirichletGroup(const std::string &sidename, const std::string &varname, int num, const int32_t *elem_ids,
                        const int32_t *side_ids);
  void setInitial(int lump_mass, double *rhs, double *mass_vals);
  void setInitialNodal(double *initial);
  void setDirichlet(int lump_mass, double *rhs, double *mass_vals);
  void clearBoundaryGroups() { boundary_groups_.clear(); }
  int numBoundaryGroups() const { return static_cast<int>(boundary_groups_.size()); }
  void assembleBoundary(int flags, const double *u, const double *u_prev, const double *u_stage, double *res,
                        double *crs_vals);
  void computeFlux(int group, const double *u, const double *u_prev, const double *u_stage, double *flux, double *dflux_du,
                   double *dflux_daux);
  void boundaryUpdate(int group);
  View boundaryView(int group, const std::string &name) const;
  void setPhysicsParameter(const std::string &name, double value);
  void applyDbcDiag(double *crs_vals);
  void gather(const double *vec, double *elem_vals);

  int numWorksets() const;
  void worksetUpdate(int index);
  View worksetView(const std::string &name) const;
  // Workset::computeSoln for the current workset (workset.cpp:1017-1190): fills the solution-field views
  void worksetComputeSolution(const double *u, const double *u_prev, const double *u_stage);
  // resetResidual + volumeResidual on the current workset: fills the "res" / "res.dx" views (Workset::getResidual)
  void worksetComputeResidual(int compute_jacobian, const double *u, const double *u_prev, const double *u_stage);

  int64_t info(const std::string &key) const;
  void setTiming(bool on) { timing_ = on; }
  double lastKernelMs() const { return last_ms_; }

  const std::vector<int32_t> &rowptr() const { return h_rowptr_; }
  const std::vector<int32_t> &colind() const { return h_colind_; }
  int numRows() const { return nrows_; }
  int device() const { return device_; }

 private:
  void requireReady(bool need_graph) const;
  void prepareRowOwner();
  bool rowOwnerUsable(std::string *why) const;
  void launchRowOwner(bool compute_jacobian, bool overwrite, double *res, double *crs_vals, bool deterministic = false);
  RowBlocksDev rowBlocksDev() const;
  BlockDev blockDev() const;
  void bindState(const double *u, const double *u_prev, const double *u_stage);
  void timedBegin();
  void timedEnd();

  int dim_ = 0, order_ = 0, qdeg_ = 0, n_ = 0, nq_ = 0, nnodes_ = 0;  // n_ = dofs per element over all variables
  // variables of the block (GroupMetaData::basis_types / basis orders; wkset->usebasis) and the point engine's tables
  struct VarInfo { int type, order, card; };
  std::vector<VarInfo> vars_;
  bool single_hgrad_ = true;  // the thermal kernels of thermal_*.hip need one HGRAD variable
  int physics_id_ = 0;
  VarLayoutDev layout_;
  DeviceBuffer<double> d_slot_tables_;
  DeviceBuffer<int8_t> d_orient_;
  bool has_orient_ = false;
  void buildVarLayout();
  // row-gather path: row -> (element, position) incidences on the device, full-block dense scratch
  DeviceBuffer<int32_t> d_inc_ptr_, d_inc_elem_, d_inc_pos_, d_inc_dof_;
  DeviceBuffer<double> d_gather_J_, d_gather_res_;
  // porousMixed direct form (ElemOut::direct_*): -1 not decided, 0 no (why in porous_direct_why_), 1 yes
  int porous_direct_ = -1, last_porous_direct_ = 0;
  int last_db_mode_ = 0;
  std::string porous_direct_why_;
  DeviceBuffer<double> d_direct_part_;   // [nrows][2][2]
  DeviceBuffer<uint8_t> d_direct_side_;  // [E][n] (dof order): which incidence of its row the element is
  DeviceBuffer<int32_t> d_direct_diag_;  // [nrows] CRS position of the diagonal of a face row, -1 otherwise
  bool porousDirectUsable();
  // database mode of the direct form (uniform mesh, constant coefficients: every element matrix is the same): rows of
  // the same CLASS -- same incident local dofs, same column slots -- are equal, so the element threads store the entries
  // of a few representative rows per class only and replicate_runs_kernel fills the rest
  struct PorousDatabase {
    int state = -1;  // -1 not tried, 0 not usable (why), 1 built
    std::string why;
    DeviceBuffer<uint8_t> jacflag;     // [E]
    DeviceBuffer<int32_t> elist;       // the elements with jacflag set
    DeviceBuffer<double> uniform;      // element matrix of the uniform block + point tables (ElemOut::direct_uniform)
    double uniform_key[5] = {0, 0, 0, 0, 0};  // (Kinv_xx, _yy, _zz, mobility, alpha_u) the tables were made for
    bool uniform_valid = false;
    int num_listed = 0;
    DeviceBuffer<int32_t> diag, chunks;  // finishing pass: diagonal positions of the COMPUTED face rows; copy chunks
    int num_chunks = 0, num_classes = 0;
    bool axis_aligned = false;  // the common element shape is an axis-aligned box
    int64_t computed_rows = 0;
  } porous_db_;
  bool porousDatabaseUsable();
  bool has_incidence_ = false;
  int max_row_ = 0;
  void prepareRowGather(bool need_jacobian, bool dense = true);
  void launchPointEngine(int compute_jacobian, const ElemOut &out, int e_begin, int e_count);
  int nelem_ = 0, nrows_ = 0, workset_size_ = 0;
  bool has_mesh_ = false, has_graph_ = false;
  int last_path_ = 0;
  int last_row_owner_kind_ = 0;
  hipStream_t stream_ = nullptr;
  int device_ = 0;  // the context's device (mha_block_desc.device): every C-ABI entry runs under a DeviceGuard for it

  RefTables ref_;
  DeviceBuffer<double> d_ref_basis_, d_ref_grad_, d_ref_wts_, d_nodeval_, d_nodegrad_;
  DeviceBuffer<double> d_phi1d_, d_dphi1d_, d_gw1d_, d_gp1d_;
  DeviceBuffer<uint8_t> d_elem_slot_;  // element-major CRS slot map of the general-element kernel (lazy)
  int elem_slot_bytes_ = 1;
  bool has_elem_slot_ = false;
  void prepareElemSlots();
  void useGeneralKernel(bool need_slots);
  DeviceBuffer<double> d_nodes_;
  DeviceBuffer<int32_t> d_lids_, d_offsets_, d_rowptr_, d_colind_;
  DeviceBuffer<uint8_t> d_fixed_;
  std::vector<int32_t> h_lids_, h_rowptr_, h_colind_;
  std::vector<uint8_t> h_fixed_;
  bool has_fixed_ = false;

  // row-owner path (row_blocks.hpp, kernels/thermal_row_owner.hip), built lazily
  struct RowOwnerData {
    bool ready = false;
    bool failed = false;  // prepareRowOwner threw: AUTO stops trying
    RowBlocks rb;
    DeviceBuffer<int32_t> row_ptr, rows, row_off, acc_size, elem_ptr, elems, pair_ptr, affine_list, general_list;
    DeviceBuffer<int32_t> pair_off, row_base, row_len, emask, epbase, seg_ptr, seg_acc, seg_base, seg_len;
    DeviceBuffer<int64_t> slot_ptr;
    DeviceBuffer<double> geo;  // [E][kGeoRec] cached element geometry
    DeviceBuffer<double> erec;  // block-major element records of K2
    DeviceBuffer<uint16_t> pair_off16;
    DeviceBuffer<int> slot_pair;  // LID slots paired by co-ownership (K2 lane layout)
    DeviceBuffer<uint32_t> pairs;
    DeviceBuffer<uint8_t> slot, flags;
    DeviceBuffer<double> khat, phi, dphi, gw, gp;
    AffineTables1D tab1d;  // thread-per-element K1
    bool k1_thread = false;
    // workgroup-merged K1 (K1PlanDev): distinct rows of every 256 consecutive elements + 16-bit positions
    DeviceBuffer<int32_t> k1_row_ptr, k1_rows, k1_elems, k1_shape_idx;
    DeviceBuffer<double> k1_shape;
    bool k1_wg = false;
    DeviceBuffer<uint16_t> k1_loc;
    K1PlanDev k1_plan;
    double max_abs_coord[3] = {0, 0, 0};  // of the block's vertices (bounds the arguments of a closed-form source)
    int slot_bytes = 1;
    int num_affine_elems = 0, num_affine_blocks = 0, num_general_blocks = 0;
    int num_shapes = 0;  // distinct geometry records (shape part, bit for bit) of the block's affine elements
    bool all_rows_covered = false;
  } ro_;
  // general-element row-owner kernel (kernels/thermal_general_row_owner.hip): its own row blocks (2x2x2 / 4x4 chunks,
  // caps sized to that kernel's LDS) and block-major slot table
  struct GeneralRowOwnerData {
    bool tried = false, usable = false;
    std::string why;
    RowBlocks rb;
    DeviceBuffer<int32_t> row_ptr, rows, elem_ptr, elems, pair_ptr, pair_off, row_len, seg_ptr, seg_acc, seg_base, seg_len;
    DeviceBuffer<int32_t> blk_rows;  // [touched element of every block][n]: global row of dof j
    DeviceBuffer<int32_t> blk_hdr;   // [block][12] counts and offsets of the block's tables
    DeviceBuffer<long long> timing;  // profiling aid (MHA_GRO_TIMING)
    DeviceBuffer<int64_t> slot_ptr;
    DeviceBuffer<uint32_t> pairs;
    DeviceBuffer<uint8_t> slot;
    bool all_rows_covered = false;
    size_t lds_bytes = 0;
  } gro_;
  void prepareGeneralRowOwner();
  RowBlocksDev generalRowBlocksDev() const;
  void launchGeneralRowOwner(bool compute_jacobian, bool overwrite, double *res, double *crs_vals, bool ordered = false);
  // row blocks keyed by assembly pattern: the matrix-core form of K2 (block_pattern.hpp); !usable -> the row-block kernel
  struct BlockPatternData {
    bool tried = false, usable = false;
    std::string why;
    int num_patterns = 0, num_roles = 0, num_blocks = 0;
    int64_t mfma_per_assembly = 0;
    BlockPatternDev dev;
    DeviceBuffer<int32_t> role, seg, wg_seg_ptr, part_ptr, part_hdr, part_lane, rowbase, erec_elem, chunk_tab, wg_seg_ptr_img;
    DeviceBuffer<double> w, erec2;
    DeviceBuffer<long long> timing;
    // geometry-database mode (one shape in the block): the kernel on one representative block per role + replication
    bool db_mode = false;
    BlockPatternDev dev_rep;  // dev with the segment tables of the representatives
    DeviceBuffer<int32_t> rep_seg, rep_wg_seg_ptr, copy_chunks;  // [chunks][4]: see launch_replicate_runs
    int copy_runs = 0;                                           // chunks
  } bpat_;
  void prepareBlockPattern();

  std::vector<int32_t> db_index_, db_first_users_;  // basis database: representative of every element, their element ids
  DeviceBuffer<int32_t> d_db_index_, d_pos_var_;
  std::vector<int8_t> h_orient_;
  bool subgrid_checked_ = false;
  // per-variable views (multi-variable blocks): reference tables at the volume / side points, physical basis arrays
  struct VarTables { DeviceBuffer<double> val, grad, div; bool ready = false; };
  struct VarViews { DeviceBuffer<double> basis, grad, div; };
  std::vector<VarTables> var_vol_tables_, var_side_tables_;
  std::vector<VarViews> ws_var_views_;
  std::map<std::string, DeviceBuffer<double>> ws_fields_;  // solution fields of the current workset, by reference name
  int ws_fields_first_ = -1;
  DeviceBuffer<double> ws_res_, ws_res_dx_;
  int ws_res_first_ = -1, ws_res_num_ = 0;
  bool ws_res_has_dx_ = false;
  int varIndex(const std::string &name) const;
  std::string varName(int v) const;
  int varComps(int v) const { return vars_[v].type == MHA_BASIS_HDIV ? dim_ : 1; }
  VarPointsDev varPoints(int v, bool side);
  int addVarGroup(int bc_type, const std::string &sidename, const std::string &varname, int num, const int32_t *elem_ids,
                  const int32_t *side_ids);
  void initialFunctions(int v, FuncDesc f[3]) const;
  // basis arrays of variable v on the current workset (aliases the single-variable views of the workset)
  void worksetVarArrays(int v, const double **basis, const double **grad, const double **div) const;

  // host mirror of BoundaryGroup: entries + the side views evaluated on request
  struct BoundaryGroupData {
    std::string sidename;
    int bc_type = 0, num = 0;
    int var = -1;  // MHA_BC_FLUX / MHA_BC_DIRICHLET: the variable the condition is for
    DeviceBuffer<int32_t> elem, side;
    DeviceBuffer<double> wts, xyz[3], nrm[3], basis, basis_grad;
    std::vector<VarViews> var_views;  // multi-variable blocks: "basis side <var>", "basis_grad side <var>"
    bool has_views = false;
  };
  std::vector<std::unique_ptr<BoundaryGroupData>> boundary_groups_;
  SideTables side_ref_;
  DeviceBuffer<double> d_side_ip_, d_side_wts_, d_side_tanU_, d_side_tanV_, d_side_basis_, d_side_grad_, d_side_nodeval_,
      d_side_nodegrad_;
  bool has_side_tables_ = false;
  void prepareSideTables();
  SideTablesDev sideTablesDev() const;
  BoundaryDev boundaryDev(const BoundaryGroupData &g) const;

  // scratch of the two-step (updateJac/scatterJac) path, one workset wide
  DeviceBuffer<double> d_local_J_, d_local_res_;

  std::unique_ptr<PhysicsBase> physics_;
  FunctionManager functions_;
  Workset wkset_;
  TimeDev time_;

  bool timing_ = false;
  double last_ms_ = 0.0;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  hipStream_t side_stream_ = nullptr;  // MHA_K1K2_OVERLAP
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
};

}  // namespace mha
