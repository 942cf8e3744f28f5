// mesh.hpp -- structured quad/hex mesh + HGRAD dof map generator, and the CRS graph rule.
// Input generation for tests/bench (the reference's mesh stack is out of scope, SURVEY.md
// section 2 row 11).  2-D order 1 reproduces SimpleMeshManager_Rectangle
// (reference: src/tools/simplemeshmanager.hpp:639-675) with offsets {0,1,3,2}
// (src/interfaces/discretizationInterface.cpp:302).
#pragma once
#include <cstdint>
#include <vector>

namespace mha {

void mesh_sizes(int dim, int order, const int *ncell, int *nverts, int *nelem, int64_t *ndof);
void mesh_structured(int dim, int order, const int *ncell, const double *lo, const double *hi,
                     double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                     uint8_t *boundary_dof);

// Blocks with several variables (HGRAD of orders dividing the largest, HVOL order 0, HDIV order 1): sizes, then the mesh
// + subcell-major dof map + lowest-order HDIV orientation signs (mesh.cpp).
void mesh_multi_sizes(int dim, const int *ncell, int nvars, const int *types, const int *orders, int *nverts, int *nelem,
                      int *n_tot, int64_t *ndof);
void mesh_structured_multi(int dim, const int *ncell, const double *lo, const double *hi, int nvars, const int *types,
                           const int *orders, double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                           int8_t *orient, uint8_t *side_mask, int32_t *dof_var);

// Overlapped CRS graph: every dof of an element couples to every dof of that element;
// columns ascending (reference: src/interfaces/linearAlgebraInterface.cpp:218-229).
void build_crs_graph(int nrows, int nelem, int n, const int32_t *lids, std::vector<int32_t> &rowptr,
                     std::vector<int32_t> &colind);

// Checks a caller-supplied graph (throws MHA_ERR_INVALID): well-formed CRS, and every (LID_i, LID_j) of every element present.
void validate_crs_graph(int nrows, int nelem, int n, const int32_t *lids, const int32_t *rowptr, const int32_t *colind);

// row -> incident (element, local position) pairs, CSR layout, element-ascending per row.
void build_row_incidence(int nrows, int nelem, int n, const int32_t *lids, std::vector<int32_t> &ptr,
                         std::vector<int32_t> &elem, std::vector<int32_t> &lpos);

}  // namespace mha
