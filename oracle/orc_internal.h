/* orc_internal.h -- helpers shared by the oracle's translation units (TEST INFRASTRUCTURE ONLY). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#define ORC_MAXP 8
int orc__ipow(int b, int e);
extern const double orc__quad_node[4][2];
extern const double orc__hex_node[8][3];
extern const int orc__quad_side[4][2];
extern const int orc__hex_side[6][4];
void orc__jac_inv_det(int dim, const double *J, double *Ji, double *det);
#define QUAD_NODE orc__quad_node
#define HEX_NODE orc__hex_node
#define QUAD_SIDE orc__quad_side
#define HEX_SIDE orc__hex_side
#define jac_inv_det orc__jac_inv_det
#endif
