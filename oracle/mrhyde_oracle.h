/*
 * mrhyde_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the MrHyDE element-local assembly hot path
 * (reference: dannys4/MrHyDE @ 2024_08_07).  It exists to CHECK the HIP
 * product path (mrhyde_amd/csrc) and to serve as the timed CPU baseline
 * ("port") in bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does.
 *
 * Parity status: pinned against the reference's own golden data --
 *   regression/discretization/HGRAD/mrhyde.gold   (basis / basis_grad, dof &
 *       quadrature-point ordering, Q1 quad + hex)
 *   regression/thermal/{2D_verification,3D_verification,
 *       2D_verification_highorder}/mrhyde.gold   (end-to-end L2 errors)
 * Q2-hex (config 2) has no reference test: "parity unpinned" at the
 * reference-test level for that order; it is bracketed by the Q1-hex and
 * Q4-quad golds running the same code, plus analytic invariants.
 *
 * The reference path cannot be compiled here (needs Trilinos: Kokkos, Sacado,
 * Intrepid2, Panzer, Tpetra -- none in the image), so there is no oracle/_ref.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef MRHYDE_ORACLE_H
#define MRHYDE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference element tables ------------------------------------------
 * src/interfaces/discretizationInterface.cpp:346-478 (getBasis/getQuadrature)
 * HGRAD = Basis_HGRAD_{QUAD,HEX}_Cn_FEM on POINTTYPE_EQUISPACED: tensor
 * Lagrange, dof index x-fastest, dof 0 at the (-1,-1[,-1]) corner.
 * Cubature = tensor Gauss-Legendre, ceil((degree+1)/2) points per direction,
 * point index x-fastest, each direction DESCENDING (+g first): pinned by
 * regression/discretization/HGRAD/mrhyde.gold:37-52,89-104.                */
int orc_gauss_npts(int degree);
void orc_gauss_line(int n, double *pts, double *wts);
void orc_lagrange_1d(int p, double x, double *val, double *der);
int orc_ref_sizes(int dim, int order, int qdeg, int *nbasis, int *nq, int *nnodes);
/* ip[nq][dim], wts[nq], basis[nbasis][nq], grad[nbasis][nq][dim],
 * nodeval[nnodes][nq], nodegrad[nnodes][nq][dim] (geometry = C1 nodal basis
 * in shards vertex order, as CellTools uses the cell topology's basis).     */
int orc_ref_tables(int dim, int order, int qdeg, double *ip, double *wts,
                   double *basis, double *grad, double *nodeval, double *nodegrad);

/* ---- structured mesh + DOF map -----------------------------------------
 * 2-D order 1 reproduces SimpleMeshManager_Rectangle bit for bit
 * (src/tools/simplemeshmanager.hpp:639-675) with offsets {0,1,3,2}
 * (src/interfaces/discretizationInterface.cpp:302).  Other cases follow the
 * same rule: element LID list = vertices in shards order first, then the
 * remaining tensor dofs in tensor order; global ids lexicographic on the
 * (order*N+1)^dim dof grid; offsets[tensor dof] = position in the LID list.  */
int orc_mesh_sizes(int dim, int order, const int *ncell, int *nverts, int *nelem, long long *ndof);
int orc_mesh_structured(int dim, int order, const int *ncell, const double *lo, const double *hi,
                        double *verts, int *cell2vert, int *lids, int *offsets,
                        unsigned char *boundary_dof);

/* ---- physical basis / integration data ---------------------------------
 * src/interfaces/discretizationInterface.cpp:732-776, 898-981.
 * nodes[E][nnodes][dim]; basis[E][n][q]; basis_grad[E][n][q][dim];
 * wts[E][q]; ip[E][q][dim].                                                 */
int orc_physical_basis(int dim, int order, int qdeg, int nelem, const double *nodes,
                       double *basis, double *basis_grad, double *wts, double *ip);

/* ---- CRS graph ----------------------------------------------------------
 * src/interfaces/linearAlgebraInterface.cpp:218-229: every dof of an element
 * couples to every dof of that element; columns sorted ascending per row.
 * Call with colind == NULL to get rowptr (and nnz = rowptr[nrows]).          */
int orc_build_graph(int nrows, int nelem, int n, const int *lids, int *rowptr, int *colind);

/* ---- thermal assembly ---------------------------------------------------*/
typedef struct orc_thermal_args {
  int dim, order, qdeg;
  int nelem, nrows, workset_size;
  const double *nodes;          /* [E][nnodes][dim]                          */
  const int *lids;              /* [E][n]                                    */
  const int *offsets;           /* [n] (one variable "e")                     */
  const unsigned char *fixed;   /* [nrows] isFixedDOF, may be NULL           */
  const double *u;              /* [nrows] current (stage) solution          */
  /* stored physical basis (Group::computeBasis, src/tools/group.cpp:134-243) */
  const double *basis, *basis_grad, *wts, *ip;
  /* transient data (src/tools/workset.cpp:559-623); ignored if !transient   */
  int transient, nsteps, nstages, stage;
  const double *u_prev;         /* [nrows][nsteps]                           */
  const double *u_stage;        /* [nrows][nstages]                          */
  const double *butcher_A;      /* [nstages][nstages]                        */
  const double *butcher_b;      /* [nstages]                                 */
  const double *bdf;            /* [nsteps+1]                                */
  double dt;
  /* coefficients ("thermal diffusion", "density", "specific heat",
   * src/physics/thermal.cpp:52-63); *_ip != NULL overrides the constant      */
  double diff, rho, cp;
  const double *diff_ip;        /* [E][q] or NULL                            */
  /* "thermal source": kind 0 = constant source_amp, 1 = per-ip array,
   * 2 = source_amp * prod_d sin(source_freq[d] * x_d)                        */
  int source_kind;
  double source_amp, source_freq[3];
  const double *source_ip;      /* [E][q]                                    */
  int compute_jacobian;
  int num_threads;              /* <=1: serial, deterministic                */
  /* outputs, accumulated into (caller zeroes, solverManager.cpp:1528-1533)  */
  const int *rowptr, *colind;
  double *crs_vals;             /* may be NULL                               */
  double *res;                  /* [nrows], receives -res.val(); may be NULL */
  double *local_J;              /* [E][n][n] updateJac convention, or NULL   */
  double *local_res;            /* [E][n]   updateRes convention, or NULL    */
  /* settings "include advection" (thermal.cpp:39): (b . grad e, v) with the
   * functions "bx","by","bz" (thermal.cpp:59-61, 150-160); adv_ip != NULL
   * overrides the constants                                                  */
  int have_advection;
  double adv[3];
  const double *adv_ip;         /* [E][q][dim] or NULL                       */
} orc_thermal_args;

int orc_ad_width(int n);
int orc_assemble_thermal(const orc_thermal_args *a);
/* strong-Dirichlet rows: diagonal := 1 (assemblyManager.cpp:1166-1179)      */
int orc_apply_dbc_diag(int nrows, const unsigned char *fixed, const int *rowptr,
                       const int *colind, double *crs_vals);
/* L2 error  sqrt(sum_e sum_pt (u_h - u_true)^2 wts)  with u_true =
 * prod_d sin(freq[d] x_d)  (src/managers/postprocessManager.cpp:1255-1268)   */
double orc_l2_error_sinprod(int dim, int order, int qdeg, int nelem, const int *lids,
                            const int *offsets, const double *basis, const double *wts,
                            const double *ip, const double *u, const double *freq);

/* ---- boundary (side) terms ----------------------------------------------------------------
 * Reference side data (setReferenceData, discretizationInterface.cpp:523-548): side cubature mapped to
 * the cell by CellTools::mapToReferenceSubcell, reference edge tangent (2-D) / face tangents (3-D), basis
 * values and gradients at the side points.  Sides in shards order (quad edges {0,1},{1,2},{2,3},{3,0};
 * hex faces {0,1,5,4},{1,2,6,5},{2,3,7,6},{0,4,7,3},{0,3,2,1},{4,5,6,7}).
 * sip[ns][nqs][dim], swts[nqs], tanU[ns][dim], tanV[ns][dim], sbasis[ns][n][nqs], sgrad[ns][n][nqs][dim],
 * snodeval[ns][nn][nqs], snodegrad[ns][nn][nqs][dim].                                              */
int orc_side_sizes(int dim, int qdeg, int *nsides, int *nqs);
int orc_side_tables(int dim, int order, int qdeg, double *sip, double *swts, double *tanU, double *tanV,
                    double *sbasis, double *sgrad, double *snodeval, double *snodegrad);
/* physical side data of boundary entries (getPhysicalBoundaryIntegrationData/Basis,
 * discretizationInterface.cpp:1608-1790, 1810-1955): wts[nb][nqs], normals[nb][nqs][dim] (unit),
 * ip[nb][nqs][dim], basis[nb][n][nqs], basis_grad[nb][n][nqs][dim]                                   */
int orc_physical_side_basis(int dim, int order, int qdeg, int nb, const double *nodes, const int *belem,
                            const int *bside, double *wts, double *normals, double *ip, double *basis,
                            double *basis_grad);

#define ORC_BC_NEUMANN 1
#define ORC_BC_WEAK_DIRICHLET 2
typedef struct orc_thermal_bnd_args {
  int dim, order, qdeg, nrows;
  const double *nodes;          /* [E][nnodes][dim] (all elements of the block)   */
  const int *lids, *offsets;
  const unsigned char *fixed;
  const double *u;
  int transient, nsteps, nstages, stage;
  const double *u_prev, *u_stage, *butcher_A, *butcher_b, *bdf;
  double dt;
  int nb;                       /* boundary entries (element, local side)          */
  const int *belem, *bside;
  int bc_type;                  /* ORC_BC_*                                         */
  int data_kind;                /* "Neumann e <side>" / "Dirichlet e <side>": 0 const, 1 array [nb][nqs], 2 sinprod */
  double data_amp, data_freq[3];
  const double *data_ip;
  double diff;                  /* "thermal diffusion" at side ip (constant)        */
  double form_param;            /* thermal.cpp:35, default 1                        */
  int compute_jacobian;
  const int *rowptr, *colind;
  double *crs_vals, *res;
} orc_thermal_bnd_args;
/* thermal::boundaryResidual (src/physics/thermal.cpp:172-281) inside the boundary-group loop of
 * assembleJacRes (assemblyManager.cpp:2518-2638): gather, seed, side fields, residual, scatter.      */
int orc_assemble_thermal_boundary(const orc_thermal_bnd_args *a);

/* ================================================================================================
 * Multi-variable blocks (porousMixed: p HVOL + u HDIV; navierstokes: ux, pr, uy[, uz] HGRAD; thermal)
 * ================================================================================================
 * Bases (getBasis, discretizationInterface.cpp:346-462): HGRAD = tensor Lagrange (as above); HVOL =
 * Basis_HVOL_C0_FEM, the constant 1 (:372-374), transformed like an HGRAD value (no 1/detJ, :1003-1012);
 * HDIV = Basis_HDIV_{QUAD,HEX}_In_FEM of degree 1 (:381-393): 2*dim functions, dof 2c = (1-x_c)/2 e_c,
 * dof 2c+1 = (1+x_c)/2 e_c (tensor line(x_c) x bubble(others), component-major), value J phi/detJ, div
 * div_hat/detJ (:1014-1065).  modifyBasisByOrientation for the lowest-order face dofs is a sign: phi.n_out of the
 * reference face (-1 for dof 2c, +1 for dof 2c+1) times -1 if the face's global vertex ids are a flipped
 * permutation (edges: id0 > id1; quads: after rotating the smallest id first, next > previous).
 * Trilinos is not vendored: the In_FEM ordering/normalisation and the orientation rule restate Intrepid2's
 * published definitions; they are pinned only through the porous L2 golds (consistency), not entry by entry. */
#define ORC_BASIS_HGRAD 0
#define ORC_BASIS_HVOL 1
#define ORC_BASIS_HDIV 2
#define ORC_PHYS_THERMAL 1
#define ORC_PHYS_POROUS_MIXED 2
#define ORC_PHYS_NAVIERSTOKES 3
#define ORC_PHYS_SHALLOWWATER_HYBRIDIZED 4
#define ORC_MAX_VARS 8
#define ORC_MAX_FUNCS 8
int orc_basis_card(int dim, int type, int order);
/* reference basis of one variable at npts reference points x[npts][dim]:
 * val[n][npts][ncomp] (ncomp = dim for HDIV, else 1), grad[n][npts][dim] (HGRAD, else untouched), div[n][npts] (HDIV) */
int orc_ref_basis_var(int dim, int type, int order, int npts, const double *x, double *val, double *grad, double *div);

/* Structured mesh + multi-variable dof map.  HGRAD orders must divide the largest HGRAD order.  Element LID
 * list: vertex sites in shards order, remaining tensor sites of the finest HGRAD lattice in tensor order, the
 * cell site (HVOL), the face sites in HDIV dof order; at every site the variables that live there in variable
 * order (the subcell-major layout panzer::DOFManager produces).  Global ids follow the same site-major rule on
 * the global lattices.  offsets: ragged [var][dof] flattened (var v starts at sum of cards before it), values =
 * positions in the LID list.  orient[E][n_tot]: sign per flattened (var,dof).  side_mask[ndof]: bit s set if
 * the dof lies on side s (0 left x-, 1 right x+, 2 bottom y-, 3 top y+, 4 back z-, 5 front z+).           */
int orc_mesh_multi_sizes(int dim, const int *ncell, int nvars, const int *types, const int *orders, int *nverts,
                         int *nelem, int *n_tot, long long *ndof);
int orc_mesh_multi(int dim, const int *ncell, const double *lo, const double *hi, int nvars, const int *types,
                   const int *orders, double *verts, int *cell2vert, int *lids, int *offsets, signed char *orient,
                   unsigned char *side_mask, int *dof_var);
/* physical basis of one variable at the volume integration points: basis[E][n][q][ncomp], grad[E][n][q][dim]
 * (HGRAD), div[E][n][q] (HDIV), wts[E][q], ip[E][q][dim]; orient_var[E][n] or NULL; any output may be NULL  */
int orc_physical_basis_var(int dim, int type, int order, int qdeg, int nelem, const double *nodes,
                           const signed char *orient, int orient_stride, int orient_off, double *basis, double *grad,
                           double *div, double *wts, double *ip);

typedef struct orc_func {
  int kind;              /* 0 constant amp, 1 per-ip array [E][q] (boundary: [nb][nqs]), 2 amp*prod sin(freq_d x_d),
                            3 deck string `expr` evaluated with orc_eval_expression at time `t`                      */
  double amp, freq[3];
  const double *ip;
  const char *expr;
  double t;
} orc_func;
/* Value of a deck function string at a point: numbers, x y z t nx ny nz h pi, + - * / ^ (right-assoc, above unary
 * minus), < > <= >=, parentheses, sin cos tan exp log abs sqrt sinh cosh (functionManager.cpp:21-22; the reference
 * builds a DAG with Interpreter::split, src/tools/interpreter.cpp).  Recursive descent, evaluated directly -- an
 * implementation independent of the product's postfix compiler.  Returns 0 and sets *err on a syntax error.        */
double orc_eval_expression(const char *expr, const double *xyz, double t, const double *nrm, double h, int *err);

/* function slots: thermal {source, diffusion, specific heat, density}; porousMixed {source, Kinv_xx, Kinv_yy,
 * Kinv_zz, total_mobility}; navierstokes {source ux, source pr, source uy, source uz, density, viscosity}.
 * params: navierstokes {useSUPG, usePSPG, fix_uz_offsets}: the reference writes the 3-D uz momentum residual through
 * uy's offsets (navierstokes.cpp:688); params[2] = 0 reproduces that, 1 uses uz's offsets.                          */
typedef struct orc_block_args {
  int dim, qdeg, nvars;
  int types[ORC_MAX_VARS], orders[ORC_MAX_VARS];
  int physics;
  int nelem, nrows;
  const double *nodes;          /* [E][nnodes][dim] */
  const int *lids;              /* [E][n_tot]       */
  const int *offsets;           /* ragged [var][dof] */
  const signed char *orient;    /* [E][n_tot] or NULL */
  const unsigned char *fixed;
  const double *u;
  int transient, nsteps, nstages, stage;
  const double *u_prev, *u_stage, *butcher_A, *butcher_b, *bdf;
  double dt;
  orc_func funcs[ORC_MAX_FUNCS];
  double params[8];
  int compute_jacobian;
  const int *rowptr, *colind;
  double *crs_vals, *res;
  double *local_J, *local_res;  /* [E][n_tot][n_tot] / [E][n_tot] in LID-position order, or NULL */
  /* boundary entries (orc_assemble_block_boundary only) */
  int nb;
  const int *belem, *bside;
  int bc_type;                  /* porousMixed: 1 = "Dirichlet" on p (weak, porousMixed.cpp:400-418);
                                   shallowwaterHybridized: 10 interface, 11 Far-field, 12 Slip                */
  orc_func bdata;               /* "Dirichlet p <side>" at the side ip */
  /* shallowwaterHybridized: trace ("aux") state and far-field state at the side ip, [nb][nqs][3] (H, Hux, Huy);
   * params = {g, Roe-like stabilisation (1) or max-EV (0)}                                                     */
  const double *aux_ip, *farfield_ip;
} orc_block_args;
/* volume terms: gather -> seed -> fields -> <physics>::volumeResidual -> scatter, AD arrays of width n_tot
 * (porousMixed.cpp:158-338, navierstokes.cpp:82-849, thermal.cpp:71-165)                                  */
int orc_assemble_block(const orc_block_args *a);
/* AssemblyManager::getWeightedMass (assemblyManager.cpp:7847-7925; getMass :7776-7840 = all weights 1): dense element
 * mass matrices mass[E][n_tot][n_tot] += sum_q basis_v(i,q) . basis_v(j,q) wts(q) masswts[v] at (off_v(i), off_v(j)),
 * HGRAD/HVOL: value; HDIV: dot product of the vector values.  Uses dim, qdeg, nvars, types, orders, nelem, nodes,
 * offsets, orient of the block description.                                                                  */
int orc_get_mass(const orc_block_args *a, const double *masswts, double *mass);
/* boundary terms (porousMixed::boundaryResidual, porousMixed.cpp:345-432)                                  */
int orc_assemble_block_boundary(const orc_block_args *a);
/* HDIV side basis of boundary entries: basis[nb][n][nqs][dim] = J phi/detJ at the side points (with orientation) */
int orc_physical_side_basis_hdiv(int dim, int qdeg, int nb, const double *nodes, const int *belem, const int *bside,
                                 const signed char *orient, int orient_stride, int orient_off, double *basis);

/* ---- shallowwaterHybridized, point level (mrhyde_oracle_swhdg.c) ------------------------------------------------
 * State order H, Hux[, Huy]; matrices row-major; dim = 1 or 2.  Each function cites the reference lines in its
 * definition; all are pinned by unit_tests/physics/shallowwaterHybridized.cpp.  The volume residual of the module
 * ((v, dS/dt) - (grad v, F(S)) - (v, source), :113-184) runs through orc_assemble_block with
 * ORC_PHYS_SHALLOWWATER_HYBRIDIZED: variables H, Hux, Huy (HGRAD), functions {source H, source Hux, source Huy},
 * params {g}.                                                                                                   */
void orc_swh_matvec(int n, const double *A, const double *x, double *y);
void orc_swh_eigendecomp(int dim, const double *Shat, const double *nrm, double g, double *L, double *lam, double *R);
void orc_swh_flux_vector(int dim, const double *S, double g, double *F);
void orc_swh_stab_term(int dim, const double *S, const double *Shat, const double *nrm, double g, int roe, double *out);
void orc_swh_boundary_term(int dim, int type, const double *S, const double *Shat, const double *Sinf,
                           const double *nrm, double g, double *out);
void orc_swh_interface_flux(int dim, int side_type, int roe, const double *S, const double *Shat, const double *Sinf,
                            const double *nrm, double g, double *out);

/* ---- shallowwaterHybridized, the HDG element (side part) ------------------------------------------------------
 * Per element: 12 interior unknowns (H, Hux, Huy; HGRAD order 1, flattened (variable, dof)) and 24 trace unknowns
 * (3 variables x Basis_HFACE_QUAD_In_FEM of degree 1, vendored in the reference: src/tools/Intrepid2_HFACE_QUAD_In_FEMdef.hpp
 * :84-196 -- per edge the 2 linear Lagrange functions of the edge's reference coordinate, edges in the order left x=-1,
 * bottom y=-1, right x=+1, top y=+1, zero off their edge).  With derivative arrays of width 36 (interior slots 0..11,
 * trace slots 12..35) the reference's operators give
 *   interior rows: boundaryResidual (shallowwaterHybridized.cpp:190-263) on all four sides,
 *                  res_(i,a) += flux_i wts N_a(side point)
 *   trace rows:    computeFlux (:270-368) integrated against the trace basis (SubGridDtN_Solver::updateFlux,
 *                  src/subgrid/subgridDtN_solver.cpp:1583-1601): res_(i,(edge,k)) += mu_k flux_i wts
 * with flux = F(Shat).n + Stab (S - Shat) on interface sides, the boundary term on Far-field / Slip sides.
 * lambda[E][24] (variable-major, then edge, then function), side_types[E][4] in shards side order (0 interface,
 * 1 Far-field, 2 Slip), farfield[3]; outputs res[E][36] = -res.val(), blocks[E][36][36] = res(r).dx(c) (stored).
 * Uses dim (2), qdeg, orders, nelem, nodes, lids, offsets, u, the time-integration data and params {g, Roe} of `a`. */
int orc_swh_hdg_element(const orc_block_args *a, const double *lambda, const unsigned char *side_types,
                        const double *farfield, double *blocks, double *res);
/* AssemblyManager::applyMassMatrixFree (assemblyManager.cpp:1582-1778): y += M x, M block diagonal by variable.
 * _free: basis recomputed per element (the !storeMass branch); _stored: dense element mass, optionally of the
 * element's database representative (index[e]); _sparse: the same through Sparse3DView storage.                     */
int orc_apply_mass_matrix_free(const orc_block_args *a, const double *masswts, const double *x, double *y);
int orc_apply_mass_stored(int nelem, int n_tot, int nvars, const int *varptr, const int *offsets, const int *lids,
                          const int *index, const double *mass, const double *x, double *y);
/* Sparse3DView(denseview, tol) (src/tools/sparse3DView.hpp:32-92); call with values == NULL to size (maxent)        */
int orc_sparse3d(int nelem, int n, const double *dense, double tol, int *maxent, int *nnz_row, double *values, int *columns);
int orc_apply_mass_sparse(int nelem, int n_tot, int nvars, const int *varptr, const int *offsets, const int *lids,
                          const int *index, int maxent, const int *nnz_row, const double *values, const int *columns,
                          const double *x, double *y);
/* identifyVolumetricDatabase (assemblyManager.cpp:4314-4467): first-match scan with orientation / measure / Jacobian
 * checks at tolerance tol; returns the number of representatives                                                    */
int orc_identify_database(const orc_block_args *a, double tol, int *index, int *first_users);

/* L2-projection systems of initial and Dirichlet data (AssemblyManager::setInitial assemblyManager.cpp:1185-1305,
 * :1830-1850, getInitial :7632-7728; setDirichlet :1855-1943, getDirichletBoundary :6288-6350, getMassBoundary
 * :6360-6425), one variable / one boundary group at a time; see the definitions for the array shapes               */
int orc_project_rhs(int nelem, int card, int nq, int ncomp, const int *lids, int n_tot, const int *off, const double *data,
                    const double *basis, const double *wts, double *rhs);
int orc_set_initial_mass(int nelem, int n_tot, const int *lids, const double *mass, int lump, int nrows, const int *rowptr,
                         const int *colind, double *vals);
int orc_set_initial_nodal(int nelem, int nnodes, const int *lids, int n_tot, const int *off, const int *vert_of_dof,
                          const double *vals, double *initial);
int orc_dirichlet_boundary(int nb, int card, int nqs, int ncomp, int n_tot, const int *off, int hdiv, const double *dip,
                           const double *basis, const double *wts, const double *normals, double *dvals, double *mass);
int orc_set_dirichlet_group(int nb, int n_tot, const int *belem, const int *lids, const unsigned char *fixed,
                            const double *dvals, const double *mass, int lump, const int *rowptr, const int *colind,
                            double *vals, double *rhs);
int orc_set_dirichlet_identity(int nelem, int n_tot, const int *lids, const unsigned char *fixed, const int *rowptr,
                               const int *colind, double *vals);

/* PhysicsInterface::fluxConditions (physicsInterface.cpp:1702-1762) + scatterRes for one variable of a boundary group:
 * res[LIDs(elem, off(dof))] -= sum_pt -flux(k,pt) wts(k,pt) basis(k,dof,pt,0); flux[nb][nqs], wts[nb][nqs],
 * basis[nb][card][nqs][ncomp], off[card], fixed rows skipped                                                  */
int orc_flux_condition(int nb, int card, int nqs, int ncomp, const int *belem, const int *lids, int n_tot,
                       const int *off, const unsigned char *fixed, const double *flux, const double *wts,
                       const double *basis, double *res);

#ifdef __cplusplus
}
#endif
#endif
