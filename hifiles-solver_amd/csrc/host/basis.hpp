// basis.hpp -- 1-D building blocks of the FR operators (setup time only).
// Definitions follow /root/reference/src/funcs.cpp:316-509,1631-1675 and the quadrature
// tables the reference reads from data/JacobiGQ.bin / JacobiGL.bin (src/cubature_1d.cpp:48-84);
// the nodes are computed here (Newton on Legendre polynomials) instead of read from a file.
#pragma once
#include <vector>

#include "hf_array.hpp"

// n_pts Gauss-Legendre (rule 0) or Gauss-Lobatto (rule 1) nodes on [-1,1], ascending
void cubature_1d_nodes(int rule, int n_pts, hf_array<double> &locs, hf_array<double> &weights);

double eval_lagrange(double r, int mode, const hf_array<double> &loc_pts);
double eval_d_lagrange(double r, int mode, const hf_array<double> &loc_pts);
double eval_legendre(double r, int mode);
double eval_d_legendre(double r, int mode);
// derivative of the left (mode 0) / right (mode 1) VCJH correction function
double eval_d_vcjh_1d(double r, int mode, int order, double eta);
// eta of the named schemes: 1 DG, 2 SD, 3 Huynh, 4 c+ ; returns false for an invalid scheme
bool compute_eta(int vcjh_scheme, int order, double &eta);
