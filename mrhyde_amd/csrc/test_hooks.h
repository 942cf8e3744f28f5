/* test_hooks.h -- exported by libmrhyde_amd.so for the CPU test-suite only; NOT part of the drop-in boundary
 * (include/mrhyde_amd.h).  Host-only, no GPU needed. */
#ifndef MRHYDE_AMD_TEST_HOOKS_H
#define MRHYDE_AMD_TEST_HOOKS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Builds the row blocks (Morton chunks of chunk_elems elements) and the block-pattern plan of the matrix-core
 * row-owner Jacobian (csrc/block_pattern.hpp) for the block (nodes [num_elems][nnodes][dim], lids [num_elems][n],
 * CRS graph, fixed [num_rows] or NULL), then walks workgroups / wavefronts / parts / blocks / MFMA panels exactly as
 * kernels/block_pattern.hip does and evaluates on the host
 *   vals[rowptr[r] + slot] = sum_(e incident to r) sum_m scale(m) * factors[e][m] * khat[m][si(e,r)][sj -> slot]
 * (khat [nsym+1][n*n] in LID-slot space, factors [num_elems][nsym+1], scale = scale_u for m < nsym, scale_t for the
 * mass component; fixed rows give zeros).  counts[4] = {patterns, roles, workgroups, parts}.
 * Returns MHA_ERR_INVALID with a reason when the blocks do not group (too many patterns). */
int mha_test_block_patterns_host_apply(int dim, int num_rows, int num_elems, int nnodes, int n, int nsym,
                                       const double *nodes, const int32_t *lids, const int32_t *rowptr,
                                       const int32_t *colind, const uint8_t *fixed, const double *khat,
                                       const double *factors, double scale_u, double scale_t, int chunk_elems,
                                       int num_cus, int max_patterns, double *vals, int *counts);

#ifdef __cplusplus
}
#endif
#endif
