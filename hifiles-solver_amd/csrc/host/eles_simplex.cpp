// eles_simplex.cpp -- host-side mirror of the reference's tetrahedron and prism classes as PRODUCERS of the dense FR
// operators and the metrics: /root/reference/src/eles_tets.cpp (set_loc_upts :224, set_tloc_fpts
// :238, set_tnorm_fpts :540, eval_nodal_basis :977, eval_d_nodal_basis :1015, shape functions :1034-1143, fill_opp_3
// :1144-1304) and src/eles_pris.cpp (point sets :210-306, set_tnorm_fpts :551, nodal basis :973-1094, shape functions
// :1095-1237, fill_opp_3 :1323-1412, face0_map :1504).
//
// What is taken from the reference is the DEFINITION of every matrix, not its code path:
//  * the nodal (Lagrange) basis of a point set does not depend on the modal basis it is computed through, so the
//    orthonormal simplex bases here are built from the three-term Jacobi recurrence (Hesthaven & Warburton's
//    Simplex2DP / Simplex3DP and their gradients) instead of the reference's hand-expanded Dubiner formulas
//    (src/funcs.cpp:1245-1617);
//  * the DG lifting coefficients are surface integrals of polynomials: they are integrated exactly with Gauss rules on
//    the collapsed square instead of the reference's fixed-order tables (cubature_tri(0,7), cubature_1d(0,10));
//  * the point sets ARE data of the reference (data/tri_inter.bin, data/tet_inter.bin): read from
//    data/simplex_points.txt (tools/export_point_tables.py).
// Correction functions: the DG member of the VCJH family on triangles and tetrahedra (vcjh_scheme_tri / _tet 1, the one the
// shipped cases use) is the lifting itself; the other members (0: c given, 2: SD-like, 3: Huynh-like, 4: c+) multiply it by
// the filter matrix Filt = (I + M^-1 K)^-1, K = c sum_alpha binom(alpha) (D^alpha)^T D^alpha over the derivatives of order p
// (src/eles_tets.cpp:1305-1503, src/funcs.cpp:717-880) -- built from the NODAL differentiation matrices and the inverse
// nodal mass matrix V V^T, neither of which depends on which orthonormal modal basis they are computed through.
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>

#include "basis.hpp"
#include "eles.hpp"

namespace
{

// ---- point tables ---------------------------------------------------------------------------------------------
struct PointTables
{
  std::map<std::pair<std::string, int>, std::vector<std::vector<double>>> tab;
  bool loaded = false;
  std::string err;
};

std::string data_dir()
{
  if (const char *e = getenv("HFX_DATA_DIR")) return e;
  Dl_info info;
  if (dladdr((void *)&data_dir, &info) && info.dli_fname)
  {
    std::string p(info.dli_fname);
    const size_t s = p.find_last_of('/');
    return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/data";
  }
  return "data";
}

PointTables &tables()
{
  static PointTables T;
  if (T.loaded) return T;
  T.loaded = true;
  const std::string path = data_dir() + "/simplex_points.txt";
  std::ifstream f(path);
  if (!f) { T.err = "Unable to open cubature file " + path; return T; } /* src/cubature_tri.cpp:88 */
  std::string line;
  while (std::getline(f, line))
  {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream h(line);
    std::string name;
    int order = 0, n = 0;
    h >> name >> order >> n;
    const int nc = (name == "tet_inter") ? 4 : 3;
    std::vector<std::vector<double>> rows(n, std::vector<double>(nc));
    for (int i = 0; i < n; i++)
    {
      if (!std::getline(f, line)) { T.err = "truncated point table " + path; return T; }
      std::istringstream r(line);
      for (int c = 0; c < nc; c++) r >> rows[i][c];
    }
    T.tab[{name, order}] = rows;
  }
  return T;
}

const std::vector<std::vector<double>> *point_table(const char *name, int order, std::string &err)
{
  PointTables &T = tables();
  if (!T.err.empty()) { err = T.err; return nullptr; }
  auto it = T.tab.find({name, order});
  if (it == T.tab.end()) { err = "cubature order not implemented."; return nullptr; } /* src/cubature_tri.cpp:108 */
  return &it->second;
}

// ---- orthonormal Jacobi polynomials (weight (1-x)^alpha (1+x)^beta) by the three-term recurrence --------------------
double jacobi_p(double x, double alpha, double beta, int N)
{
  const double g0 = std::pow(2.0, alpha + beta + 1) / (alpha + beta + 1) * std::tgamma(alpha + 1) * std::tgamma(beta + 1) /
                    std::tgamma(alpha + beta + 1);
  double p0 = 1.0 / std::sqrt(g0);
  if (N == 0) return p0;
  const double g1 = (alpha + 1) * (beta + 1) / (alpha + beta + 3) * g0;
  double p1 = ((alpha + beta + 2) * x / 2 + (alpha - beta) / 2) / std::sqrt(g1);
  if (N == 1) return p1;
  double aold = 2 / (2 + alpha + beta) * std::sqrt((alpha + 1) * (beta + 1) / (alpha + beta + 3));
  for (int i = 1; i <= N - 1; i++)
  {
    const double h1 = 2 * i + alpha + beta;
    const double anew = 2 / (h1 + 2) * std::sqrt((i + 1) * (i + 1 + alpha + beta) * (i + 1 + alpha) * (i + 1 + beta) / (h1 + 1) / (h1 + 3));
    const double bnew = -(alpha * alpha - beta * beta) / h1 / (h1 + 2);
    const double p2 = 1 / anew * (-aold * p0 + (x - bnew) * p1);
    p0 = p1; p1 = p2; aold = anew;
  }
  return p1;
}

double grad_jacobi_p(double x, double alpha, double beta, int N)
{
  return N == 0 ? 0.0 : std::sqrt(N * (N + alpha + beta + 1.0)) * jacobi_p(x, alpha + 1, beta + 1, N - 1);
}

// ---- triangle {r, s >= -1, r + s <= 0}: orthonormal basis psi_ij, i + j <= p, and its gradient ------------------------
void rs_to_ab(double r, double s, double &a, double &b)
{
  a = (s != 1.0) ? 2 * (1 + r) / (1 - s) - 1 : -1.0;
  b = s;
}

double simplex2d(double r, double s, int i, int j)
{
  double a, b;
  rs_to_ab(r, s, a, b);
  return std::sqrt(2.0) * jacobi_p(a, 0, 0, i) * jacobi_p(b, 2 * i + 1, 0, j) * std::pow(1 - b, i);
}

void grad_simplex2d(double r, double s, int id, int jd, double &dr, double &ds)
{
  double a, b;
  rs_to_ab(r, s, a, b);
  const double fa = jacobi_p(a, 0, 0, id), dfa = grad_jacobi_p(a, 0, 0, id);
  const double gb = jacobi_p(b, 2 * id + 1, 0, jd), dgb = grad_jacobi_p(b, 2 * id + 1, 0, jd);
  dr = dfa * gb;
  if (id > 0) dr *= std::pow(0.5 * (1 - b), id - 1);
  ds = dfa * (gb * (0.5 * (1 + a)));
  if (id > 0) ds *= std::pow(0.5 * (1 - b), id - 1);
  double tmp = dgb * std::pow(0.5 * (1 - b), id);
  if (id > 0) tmp -= 0.5 * id * gb * std::pow(0.5 * (1 - b), id - 1);
  ds += fa * tmp;
  const double sc = std::pow(2.0, id + 0.5);
  dr *= sc;
  ds *= sc;
}

// ---- tetrahedron {r, s, t >= -1, r + s + t <= -1}: orthonormal basis psi_ijk, i + j + k <= p, and its gradient ----------
void rst_to_abc(double r, double s, double t, double &a, double &b, double &c)
{
  a = (s + t != 0.0) ? 2 * (1 + r) / (-s - t) - 1 : -1.0;
  b = (t != 1.0) ? 2 * (1 + s) / (1 - t) - 1 : -1.0;
  c = t;
}

double simplex3d(double r, double s, double t, int i, int j, int k)
{
  double a, b, c;
  rst_to_abc(r, s, t, a, b, c);
  return 2 * std::sqrt(2.0) * jacobi_p(a, 0, 0, i) * jacobi_p(b, 2 * i + 1, 0, j) * std::pow(1 - b, i) *
         jacobi_p(c, 2 * (i + j) + 2, 0, k) * std::pow(1 - c, i + j);
}

void grad_simplex3d(double r, double s, double t, int id, int jd, int kd, double (&g)[3])
{
  double a, b, c;
  rst_to_abc(r, s, t, a, b, c);
  const double fa = jacobi_p(a, 0, 0, id), dfa = grad_jacobi_p(a, 0, 0, id);
  const double gb = jacobi_p(b, 2 * id + 1, 0, jd), dgb = grad_jacobi_p(b, 2 * id + 1, 0, jd);
  const double hc = jacobi_p(c, 2 * (id + jd) + 2, 0, kd), dhc = grad_jacobi_p(c, 2 * (id + jd) + 2, 0, kd);
  double vr = dfa * (gb * hc);
  if (id > 0) vr *= std::pow(0.5 * (1 - b), id - 1);
  if (id + jd > 0) vr *= std::pow(0.5 * (1 - c), id + jd - 1);
  double vs = 0.5 * (1 + a) * vr;
  double tmp = dgb * std::pow(0.5 * (1 - b), id);
  if (id > 0) tmp += (-0.5 * id) * (gb * std::pow(0.5 * (1 - b), id - 1));
  if (id + jd > 0) tmp *= std::pow(0.5 * (1 - c), id + jd - 1);
  tmp = fa * (tmp * hc);
  vs += tmp;
  double vt = 0.5 * (1 + a) * vr + 0.5 * (1 + b) * tmp;
  tmp = dhc * std::pow(0.5 * (1 - c), id + jd);
  if (id + jd > 0) tmp -= 0.5 * (id + jd) * (hc * std::pow(0.5 * (1 - c), id + jd - 1));
  tmp = fa * (gb * tmp);
  tmp *= std::pow(0.5 * (1 - b), id);
  vt += tmp;
  const double sc = std::pow(2.0, 2 * id + jd + 1.5);
  g[0] = vr * sc; g[1] = vs * sc; g[2] = vt * sc;
}

// mode numbering: any fixed order (the nodal basis does not depend on it)
void modes2d(int p, std::vector<int> &mi, std::vector<int> &mj)
{
  for (int i = 0; i <= p; i++)
    for (int j = 0; j <= p - i; j++) { mi.push_back(i); mj.push_back(j); }
}
void modes3d(int p, std::vector<int> &mi, std::vector<int> &mj, std::vector<int> &mk)
{
  for (int i = 0; i <= p; i++)
    for (int j = 0; j <= p - i; j++)
      for (int k = 0; k <= p - i - j; k++) { mi.push_back(i); mj.push_back(j); mk.push_back(k); }
}

// dense inverse by Gauss-Jordan elimination with partial pivoting (n x n, row-major in / out)
bool invert(std::vector<double> &A, int n)
{
  std::vector<double> I((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; c++)
  {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
    if (A[(size_t)piv * n + c] == 0.0) return false;
    if (piv != c)
      for (int k = 0; k < n; k++) { std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]); std::swap(I[(size_t)c * n + k], I[(size_t)piv * n + k]); }
    const double d = 1.0 / A[(size_t)c * n + c];
    for (int k = 0; k < n; k++) { A[(size_t)c * n + k] *= d; I[(size_t)c * n + k] *= d; }
    for (int r = 0; r < n; r++)
      if (r != c)
      {
        const double f = A[(size_t)r * n + c];
        if (f != 0.0)
          for (int k = 0; k < n; k++) { A[(size_t)r * n + k] -= f * A[(size_t)c * n + k]; I[(size_t)r * n + k] -= f * I[(size_t)c * n + k]; }
      }
  }
  A = I;
  return true;
}

// ---- the VCJH filter matrix of a simplex point set ---------------------------------------------------------------------
double factorial(int n)
{
  double f = 1.0;
  for (int i = 2; i <= n; i++) f *= i;
  return f;
}

// c of the scheme (src/eles_tets.cpp:1336-1390, src/funcs.cpp:744-795): 0 the user's value, 1 DG (0), 2 SD-like, 3 Huynh-like,
// 4 c+; the c+ values are tabulated for orders 2..5 (tetrahedra) / 2..4 (triangles)
bool vcjh_c_simplex(int nd, int scheme, int order, double c_user, double &c, std::string &err)
{
  const double ap = 1. / std::pow(2.0, order) * factorial(2 * order) / (factorial(order) * factorial(order));
  const double c_sd_1d = (2 * order) / ((2 * order + 1) * (order + 1) * (factorial(order) * ap) * (factorial(order) * ap));
  const double c_hu_1d = (2 * (order + 1)) / ((2 * order + 1) * order * (factorial(order) * ap) * (factorial(order) * ap));
  double c_plus = 0.0, c_plus_1d = 1.0;
  if (scheme > 1)
  {
    static const double cp1d[6] = {0, 0, 0.206, 3.80e-3, 4.67e-5, 4.28e-7};
    static const double cp3d[6] = {0, 0, 3.07e-2, 5.44e-4, 9.92e-6, 1.10e-7};
    static const double cp2d[6] = {0, 0, 3.13e-2, 4.67e-4, 6.55e-6, 0};
    if (order < 2 || order > (nd == 3 ? 5 : 4)) { err = "C_plus scheme not implemented for this order"; return false; }
    c_plus_1d = cp1d[order];
    c_plus = nd == 3 ? cp3d[order] : cp2d[order];
  }
  if (scheme == 0) c = c_user;
  else if (scheme == 1) c = 0.0;
  else if (scheme == 2) c = (c_sd_1d / c_plus_1d) * c_plus;
  else if (scheme == 3) c = (c_hu_1d / c_plus_1d) * c_plus;
  else if (scheme == 4) c = c_plus;
  else { err = nd == 3 ? "VCJH tetrahedral scheme not recognized" : "VCJH triangular scheme not recognized"; return false; }
  return true;
}

typedef std::vector<double> Mat; // n x n, row-major
static Mat matmul(const Mat &A, const Mat &B, int n)
{
  Mat C((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < n; k++)
    {
      const double a = A[(size_t)i * n + k];
      if (a != 0.0)
        for (int j = 0; j < n; j++) C[(size_t)i * n + j] += a * B[(size_t)k * n + j];
    }
  return C;
}

// Filt (n x n, row-major) from the nodal differentiation matrices D[d](i, j) = d l_j / d xi_d at node i and V(i, m) = psi_m(node i)
bool vcjh_filter(int nd, int order, int n, const std::vector<Mat> &D, const Mat &V, double c, Mat &Filt)
{
  Mat K((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0;
  auto add = [&](const Mat &Dh, double coeff) {
    // K += c coeff Dh^T Dh
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
      {
        double t = 0.0;
        for (int k = 0; k < n; k++) t += Dh[(size_t)k * n + i] * Dh[(size_t)k * n + j];
        K[(size_t)i * n + j] += c * coeff * t;
      }
  };
  if (nd == 2)
    for (int k = 0; k <= order; k++)
    {
      Mat Dh = I;
      for (int q = 0; q < k; q++) Dh = matmul(Dh, D[1], n);
      for (int q = 0; q < order - k; q++) Dh = matmul(Dh, D[0], n);
      add(Dh, (1. / n) * (factorial(order) / (factorial(k) * factorial(order - k))));
    }
  else
    for (int v = 1; v <= order + 1; v++)
      for (int w = 1; w <= v; w++)
      {
        Mat Dh = I;
        for (int q = 1; q <= order - v + 1; q++) Dh = matmul(Dh, D[0], n);
        for (int q = 1; q <= v - w; q++) Dh = matmul(Dh, D[1], n);
        for (int q = 1; q <= w - 1; q++) Dh = matmul(Dh, D[2], n);
        add(Dh, (1. / n) * (factorial(order) / (factorial(v - 1) * factorial(order - (v - 1)))) *
                    (factorial(v - 1) / (factorial(w - 1) * factorial((v - 1) - (w - 1)))));
      }
  // inverse nodal mass matrix V V^T, then (I + M^-1 K)^-1
  Mat Minv((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
    {
      double t = 0.0;
      for (int m = 0; m < n; m++) t += V[(size_t)i * n + m] * V[(size_t)j * n + m];
      Minv[(size_t)i * n + j] = t;
    }
  Filt = matmul(Minv, K, n);
  for (int i = 0; i < n; i++) Filt[(size_t)i * n + i] += 1.0;
  return invert(Filt, n);
}

// Lagrange basis of a triangle point set: coefficients C[m][i] with l_i(r,s) = sum_m C[m][i] psi_m(r,s) (= V^-T)
struct TriNodal
{
  int p = 0, n = 0;
  std::vector<int> mi, mj;
  std::vector<double> vinv; // (n x n) row-major: vinv[m*n + i] = (V^-1)(m, i), V(i, m) = psi_m(point i)
  bool build(int order, const std::vector<double> &r, const std::vector<double> &s)
  {
    p = order;
    mi.clear(); mj.clear();
    modes2d(p, mi, mj);
    n = (int)mi.size();
    std::vector<double> V((size_t)n * n);
    for (int i = 0; i < n; i++)
      for (int m = 0; m < n; m++) V[(size_t)i * n + m] = simplex2d(r[i], s[i], mi[m], mj[m]);
    vinv = V;
    return invert(vinv, n);
  }
  double value(int idx, double r, double s) const
  {
    double v = 0.0;
    for (int m = 0; m < n; m++) v += vinv[(size_t)m * n + idx] * simplex2d(r, s, mi[m], mj[m]);
    return v;
  }
  void gradient(int idx, double r, double s, double &dr, double &ds) const
  {
    dr = ds = 0.0;
    for (int m = 0; m < n; m++)
    {
      double a, b;
      grad_simplex2d(r, s, mi[m], mj[m], a, b);
      dr += vinv[(size_t)m * n + idx] * a;
      ds += vinv[(size_t)m * n + idx] * b;
    }
  }
};

// n-point Gauss-Legendre rule on [-1,1]
void gauss(int n, std::vector<double> &x, std::vector<double> &w)
{
  hf_array<double> lx, lw;
  cubature_1d_nodes(0, n, lx, lw);
  x.resize(n); w.resize(n);
  for (int i = 0; i < n; i++) { x[i] = lx(i); w[i] = lw(i); }
}

// DG lifting of the triangle's edges: out(j, edge*(p+1) + fp) = divergence of the correction function of edge flux
// point fp at triangle point j (src/funcs.cpp:644-665,962-1046): sum_m sigma_m psi_m(x_j), sigma_m = the edge integral
// of the 1-D Lagrange polynomial of fp times psi_m.  Edge parametrisation as the reference's: edge 0: (xi, -1),
// edge 1: (-xi, xi), edge 2: (-1, -xi), xi in [-1,1] running with the edge's 1-D flux points.
void tri_dg_lifting(int p, const TriNodal &T, const std::vector<double> &r, const std::vector<double> &s, const hf_array<double> &loc_1d,
                    std::vector<double> &out)
{
  const int n = T.n, ne = p + 1;
  out.assign((size_t)n * 3 * ne, 0.0);
  std::vector<double> gx, gw;
  gauss(p + 3, gx, gw);
  const double len[3] = {2.0, 2.0 * std::sqrt(2.0), 2.0};
  for (int edge = 0; edge < 3; edge++)
    for (int fp = 0; fp < ne; fp++)
    {
      std::vector<double> sigma(n, 0.0);
      for (size_t q = 0; q < gx.size(); q++)
      {
        const double xi = gx[q];
        const double rr = (edge == 0) ? xi : (edge == 1) ? -xi : -1.0;
        const double ss = (edge == 0) ? -1.0 : (edge == 1) ? xi : -xi;
        const double l = eval_lagrange(xi, fp, loc_1d);
        for (int m = 0; m < n; m++) sigma[m] += gw[q] * simplex2d(rr, ss, T.mi[m], T.mj[m]) * l;
      }
      for (int m = 0; m < n; m++) sigma[m] *= len[edge] / 2;
      for (int j = 0; j < n; j++)
      {
        double v = 0.0;
        for (int m = 0; m < n; m++) v += sigma[m] * simplex2d(r[j], s[j], T.mi[m], T.mj[m]);
        out[(size_t)j + (size_t)n * (edge * ne + fp)] = v;
      }
    }
}

} // namespace

// =====================================================================================================================
// tetrahedra
// =====================================================================================================================
int eles_tets::setup_ele_type_specific()
{
  ele_type = 2;
  n_dims = 3;
  if (run_input->equation != 0) { fail("Equation not supported"); return 1; }
  if (run_input->over_int) { fail("eles_tets: over-integration is built for the tensor-product classes only (no cubature tables of tetrahedra here)"); return 1; }
  if (run_input->vcjh_scheme_tet < 0 || run_input->vcjh_scheme_tet > 4) { fail("VCJH tetrahedral scheme not recognized"); return 1; } /* src/eles_tets.cpp:1389 */
  if (run_input->upts_type_tet != 0 || run_input->fpts_type_tet != 0) { fail("eles_tets: point rule not implemented (rule 0, internal, is)"); return 1; }
  n_fields = 5;
  n_inters_per_ele = 4;
  const int p = order;
  n_upts_per_ele = (p + 1) * (p + 2) * (p + 3) / 6;
  upts_type = run_input->upts_type_tet;
  std::string terr;
  const auto *tet = point_table("tet_inter", p, terr);
  const auto *tri = point_table("tri_inter", p, terr);
  if (!tet || !tri) { fail(terr); return 1; }
  // src/eles_tets.cpp:224-236
  loc_upts.setup(n_dims, n_upts_per_ele);
  for (int i = 0; i < n_upts_per_ele; i++)
    for (int d = 0; d < 3; d++) loc_upts(d, i) = (*tet)[i][d];
  const int nft = (p + 1) * (p + 2) / 2;
  n_fpts_per_inter.setup(4);
  for (int f = 0; f < 4; f++) n_fpts_per_inter(f) = nft;
  n_fpts_per_ele = 4 * nft;
  // flux points: the triangle's points mapped to the four faces (src/eles_tets.cpp:238-282)
  tloc_fpts.setup(n_dims, n_fpts_per_ele);
  for (int j = 0; j < p + 1; j++)
    for (int i = 0; i < p + 1 - j; i++)
    {
      const int it = j * (p + 1) - (j - 1) * j / 2 + i;
      const int ia0 = j * (p + 1) - (j - 1) * j / 2 + (p - j - i);
      const double a0 = (*tri)[ia0][0], r0 = (*tri)[it][0], r1 = (*tri)[it][1];
      tloc_fpts(0, it) = a0; tloc_fpts(1, it) = r0; tloc_fpts(2, it) = r1;
      tloc_fpts(0, nft + it) = -1; tloc_fpts(1, nft + it) = r1; tloc_fpts(2, nft + it) = r0;
      tloc_fpts(0, 2 * nft + it) = r0; tloc_fpts(1, 2 * nft + it) = -1; tloc_fpts(2, 2 * nft + it) = r1;
      tloc_fpts(0, 3 * nft + it) = r1; tloc_fpts(1, 3 * nft + it) = r0; tloc_fpts(2, 3 * nft + it) = -1;
    }
  // reference normals (src/eles_tets.cpp:540-573)
  tnorm_fpts.setup(n_dims, n_fpts_per_ele);
  tnorm_fpts.initialize_to_zero();
  for (int f = 0; f < 4; f++)
    for (int j = 0; j < nft; j++)
    {
      const int fpt = nft * f + j;
      if (f == 0)
        for (int d = 0; d < 3; d++) tnorm_fpts(d, fpt) = 1. / std::sqrt(3.);
      else
        tnorm_fpts(f - 1, fpt) = -1.0;
    }
  // nodal basis through the orthonormal modal one: l = V^-T psi (src/eles_tets.cpp:705-716,977-994)
  mode_i.clear(); mode_j.clear(); mode_k.clear();
  modes3d(p, mode_i, mode_j, mode_k);
  const int n = n_upts_per_ele;
  std::vector<double> V((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int m = 0; m < n; m++) V[(size_t)i * n + m] = simplex3d(loc_upts(0, i), loc_upts(1, i), loc_upts(2, i), mode_i[m], mode_j[m], mode_k[m]);
  vinv = V;
  if (!invert(vinv, n)) { fail("eles_tets: singular Vandermonde matrix"); return 1; }
  set_opp_0();
  set_opp_1();
  set_opp_2();
  set_opp_3();
  if (viscous)
  {
    set_opp_4();
    set_opp_5();
    set_opp_6();
  }
  return failed() ? 1 : 0;
}

// The LES filter of the closures that filter the solution (SGS_model 2, 3, 4) on tetrahedra, src/eles_tets.cpp:576-690:
//   filter_type 2: modal, V diag(exp(-(2 m / N)^2 / 48)) V^-1 with the modes numbered as the reference's Vandermonde matrix
//     numbers them (src/funcs.cpp:1461-1497: by total degree, m the mode's position) -- the scaling of a mode cancels;
//   filter_type 0, 1: the reference stops ("not implemented for tris");  otherwise the element average 1/N;
//   then the reference's symmetrisation and row normalisation passes, statement by statement (they work in place).
int eles_tets::compute_filter_upts()
{
  const int N = n_upts_per_ele, p = order;
  const int type = run_input->filter_type;
  if (type == 0) { fail("Vasilyev filters not implemented for tris. Exiting."); return 1; } /* src/eles_tets.cpp:610 */
  if (type == 1) { fail("Gaussian filter not implemented for tris. Exiting."); return 1; }  /* src/eles_tets.cpp:615 */
  filter_upts.setup(N, N);
  filter_upts.initialize_to_zero();
  if (type == 2)
  {
    // position of every mode in the reference's numbering
    std::vector<int> ri, rj, rk;
    for (int m = 0; m <= p; m++)
      for (int n = 0; n <= m; n++)
        for (int k = 0; k <= n; k++) { const int j = n - k; ri.push_back(m - j - k); rj.push_back(j); rk.push_back(k); }
    std::vector<double> V((size_t)N * N), Vi;
    for (int i = 0; i < N; i++)
      for (int m = 0; m < N; m++) V[(size_t)i * N + m] = simplex3d(loc_upts(0, i), loc_upts(1, i), loc_upts(2, i), ri[m], rj[m], rk[m]);
    Vi = V;
    if (!invert(Vi, N)) { fail("eles_tets: singular Vandermonde matrix"); return 1; }
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++)
      {
        double t = 0.0;
        for (int m = 0; m < N; m++)
        {
          const double eta = m / double(N);
          t += V[(size_t)i * N + m] * std::exp(-std::pow(2.0 * eta, 2.0) / 48.0) * Vi[(size_t)m * N + j];
        }
        filter_upts(i, j) = t;
      }
  }
  else
    filter_upts.initialize_to_value(1.0 / N);
  int N2 = N / 2;
  if (N % 2 != 0) N2 += 1;
  for (int i = 0; i < N2; i++)
    for (int j = 0; j < N; j++)
    {
      filter_upts(i, j) = 0.5 * filter_upts(i, j) + filter_upts(N - i - 1, N - j - 1);
      filter_upts(N - i - 1, N - j - 1) = filter_upts(i, j);
    }
  for (int i = 0; i < N2; i++)
  {
    double norm = 0.0;
    for (int j = 0; j < N; j++) norm += filter_upts(i, j);
    for (int j = 0; j < N; j++) filter_upts(i, j) /= norm;
    for (int j = 0; j < N; j++) filter_upts(N - i - 1, N - j - 1) = filter_upts(i, j);
  }
  return 0;
}

// Shock capturing on a simplex class: from the modes (in the reference's numbering), their values at the solution points, the
// filter strength and the norm of every mode and the set of highest modes -> vandermonde, inv_vandermonde,
// exp_filter = V diag(sigma) V^-1, norm_basis_persson, persson_high_modes
static int modal_shock_operators(eles *E, int n, const std::vector<double> &V, const std::vector<double> &sigma,
                                 const std::vector<double> &norm, const std::vector<int> &high, std::string &err)
{
  std::vector<double> Vi = V;
  if (!invert(Vi, n)) { err = "singular Vandermonde matrix"; return 1; }
  E->vandermonde.setup(n, n);
  E->inv_vandermonde.setup(n, n);
  E->exp_filter.setup(n, n);
  E->norm_basis_persson.setup(n);
  E->persson_high_modes.setup(n);
  for (int i = 0; i < n; i++)
  {
    E->norm_basis_persson(i) = norm[i];
    E->persson_high_modes(i) = high[i];
    for (int j = 0; j < n; j++)
    {
      E->vandermonde(i, j) = V[(size_t)i * n + j];
      E->inv_vandermonde(i, j) = Vi[(size_t)i * n + j];
      double t = 0.0;
      for (int m = 0; m < n; m++) t += V[(size_t)i * n + m] * sigma[m] * Vi[(size_t)m * n + j];
      E->exp_filter(i, j) = t;
    }
  }
  return 0;
}

// the exponential filter's strength at the normalised degree eta (src/eles_tets.cpp:733-737, src/eles_pris.cpp:647-653)
static double expf_sigma(const input *in, double eta, double eta_c)
{
  if (eta <= eta_c) return 1.0;
  return std::exp(-in->expf_fac * std::pow((eta - eta_c) / (1. - eta_c), in->expf_order));
}

// src/eles_tets.cpp:705-797.  The modes in the order of the reference's Vandermonde matrix (by total degree, src/funcs.cpp:
// 1461-1497); the basis is orthonormal, so every norm is one and the sensor's highest modes are those past the P(order-1) space.
int eles_tets::set_shock_capture_operators()
{
  if (run_input->shock_cap != 1) { fail("Shock capturing method not implemented."); return 1; } /* src/eles_tets.cpp:79 */
  const int n = n_upts_per_ele, p = order;
  std::vector<double> V((size_t)n * n), sigma(n), norm(n, 1.0);
  std::vector<int> high(n);
  const double eta_c = (double)run_input->expf_cutoff / (double)p;
  int m = 0;
  for (int tot = 0; tot <= p; tot++)
    for (int nn = 0; nn <= tot; nn++)
      for (int k = 0; k <= nn; k++, m++)
      {
        const int j = nn - k, i = tot - j - k;
        for (int u = 0; u < n; u++) V[(size_t)u * n + m] = simplex3d(loc_upts(0, u), loc_upts(1, u), loc_upts(2, u), i, j, k);
        sigma[m] = expf_sigma(run_input, (double)tot / (double)p, eta_c);
        high[m] = (m >= p * (p + 1) * (p + 2) / 6) ? 1 : 0;
      }
  std::string e;
  if (modal_shock_operators(this, n, V, sigma, norm, high, e)) { fail("eles_tets: " + e); return 1; }
  return 0;
}

// the insphere diameter 6 V / (sum of the face areas) of the tetrahedron on the first four shape nodes (src/eles_tets.cpp:1599-1633)
double eles_tets::calc_h_ref_specific(int in_ele)
{
  double a[3], b[3], c[3], d[3], e[3];
  for (int i = 0; i < 3; i++)
  {
    a[i] = shape(i, 1, in_ele) - shape(i, 0, in_ele);
    b[i] = shape(i, 2, in_ele) - shape(i, 0, in_ele);
    c[i] = shape(i, 3, in_ele) - shape(i, 0, in_ele);
    d[i] = shape(i, 2, in_ele) - shape(i, 1, in_ele);
    e[i] = shape(i, 3, in_ele) - shape(i, 1, in_ele);
  }
  auto area = [](const double *p, const double *q) {
    return 0.5 * std::sqrt(std::pow(p[1] * q[2] - p[2] * q[1], 2) + std::pow(p[0] * q[2] - p[2] * q[0], 2) + std::pow(p[0] * q[1] - p[1] * q[0], 2));
  };
  // triple product a . (b x c) (src/funcs.cpp trip_prod)
  const double trip = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
  const double vol = 1. / 6. * trip;
  return 6. * vol / (area(a, b) + area(a, c) + area(b, c) + area(d, e));
}

double eles_tets::eval_nodal_basis(int in_index, const hf_array<double> &loc)
{
  const int n = n_upts_per_ele;
  double v = 0.0;
  for (int m = 0; m < n; m++) v += vinv[(size_t)m * n + in_index] * simplex3d(loc(0), loc(1), loc(2), mode_i[m], mode_j[m], mode_k[m]);
  return v;
}

double eles_tets::eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &loc)
{
  const int n = n_upts_per_ele;
  double v = 0.0;
  for (int m = 0; m < n; m++)
  {
    double g[3];
    grad_simplex3d(loc(0), loc(1), loc(2), mode_i[m], mode_j[m], mode_k[m], g);
    v += vinv[(size_t)m * n + in_index] * g[in_cpnt];
  }
  return v;
}

// the DG lifting (src/eles_tets.cpp:1168-1303): opp_3(j, i) = sum_m sigma_m psi_m(upt j), sigma_m = face_jac * the
// integral over the face of (2-D Lagrange polynomial of face flux point i) * psi_m
void eles_tets::fill_opp_3(hf_array<double> &o3)
{
  const int p = order, n = n_upts_per_ele, nft = (p + 1) * (p + 2) / 2;
  std::vector<double> gx, gw;
  gauss(p + 4, gx, gw);
  for (int face = 0; face < 4; face++)
  {
    // the face's own coordinates of its flux points (src/eles_tets.cpp:1222-1237)
    std::vector<double> fr(nft), fs(nft);
    for (int i = 0; i < nft; i++)
    {
      const double r = tloc_fpts(0, face * nft + i), s = tloc_fpts(1, face * nft + i), t = tloc_fpts(2, face * nft + i);
      if (face == 0) { fr[i] = r; fs[i] = t; }
      else if (face == 1) { fr[i] = t; fs[i] = s; }
      else if (face == 2) { fr[i] = r; fs[i] = t; }
      else { fr[i] = s; fs[i] = r; }
    }
    TriNodal F;
    if (!F.build(p, fr, fs)) { fail("eles_tets: singular face Vandermonde matrix"); return; }
    const double face_jac = (face == 0) ? std::sqrt(3.) : 1.0;
    for (int fp = 0; fp < nft; fp++)
    {
      std::vector<double> sigma(n, 0.0);
      // exact integration over the reference triangle through the collapsed square: r = (1+a)(1-b)/2 - 1, s = b
      for (size_t qa = 0; qa < gx.size(); qa++)
        for (size_t qb = 0; qb < gx.size(); qb++)
        {
          const double a = gx[qa], b = gx[qb];
          const double rf = 0.5 * (1 + a) * (1 - b) - 1, sf = b, w = gw[qa] * gw[qb] * 0.5 * (1 - b);
          double r, s, t; // src/eles_tets.cpp:1258-1283
          if (face == 0) { r = rf; t = sf; s = -1. - t - r; }
          else if (face == 1) { r = -1.0; s = sf; t = rf; }
          else if (face == 2) { r = rf; s = -1.0; t = sf; }
          else { r = sf; s = rf; t = -1.0; }
          const double l = F.value(fp, rf, sf);
          for (int m = 0; m < n; m++) sigma[m] += w * simplex3d(r, s, t, mode_i[m], mode_j[m], mode_k[m]) * l;
        }
      for (int j = 0; j < n; j++)
      {
        double v = 0.0;
        for (int m = 0; m < n; m++) v += sigma[m] * face_jac * simplex3d(loc_upts(0, j), loc_upts(1, j), loc_upts(2, j), mode_i[m], mode_j[m], mode_k[m]);
        o3(j, face * nft + fp) = v;
      }
    }
  }
  // the other members of the VCJH family: opp_3 = Filt . opp_3_dg (src/eles_tets.cpp:1144-1166)
  double c = 0.0;
  std::string err;
  if (!vcjh_c_simplex(3, run_input->vcjh_scheme_tet, p, run_input->c_tet, c, err)) { fail(err); return; }
  if (c == 0.0) return;
  std::vector<Mat> D(3, Mat((size_t)n * n));
  Mat V((size_t)n * n), Filt;
  hf_array<double> loc(3);
  for (int i = 0; i < n; i++)
  {
    for (int d = 0; d < 3; d++) loc(d) = loc_upts(d, i);
    for (int j = 0; j < n; j++)
    {
      for (int d = 0; d < 3; d++) D[d][(size_t)i * n + j] = eval_d_nodal_basis(j, d, loc);
      V[(size_t)i * n + j] = simplex3d(loc(0), loc(1), loc(2), mode_i[j], mode_j[j], mode_k[j]);
    }
  }
  if (!vcjh_filter(3, p, n, D, V, c, Filt)) { fail("eles_tets: singular VCJH filter matrix"); return; }
  const int nfp = n_fpts_per_ele;
  std::vector<double> dg((size_t)n * nfp);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < nfp; i++) dg[(size_t)j * nfp + i] = o3(j, i);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < nfp; i++)
    {
      double t = 0.0;
      for (int k = 0; k < n; k++) t += Filt[(size_t)j * n + k] * dg[(size_t)k * nfp + i];
      o3(j, i) = t;
    }
}

// Shape functions of the straight-sided (4 nodes) and the quadratic (10 nodes) tetrahedron, src/eles_tets.cpp:1034-1141, through the
// barycentric coordinates l_0 = -(r+s+t+1)/2, l_1 = (r+1)/2, l_2 = (s+1)/2, l_3 = (t+1)/2: vertex v -> l_v (2 l_v - 1), the node
// on edge (a, b) -> 4 l_a l_b; the reference numbers the edge nodes 4 (0,1), 5 (0,2), 6 (0,3), 7 (1,2), 8 (2,3), 9 (3,1).
static const int tet_edge[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {2, 3}, {3, 1}};
static inline void tet_bary(const hf_array<double> &loc, double l[4])
{
  l[0] = -0.5 * (loc(0) + loc(1) + loc(2) + 1.);
  for (int v = 1; v < 4; v++) l[v] = 0.5 * (loc(v - 1) + 1.);
}
static inline double tet_dbary(int v, int k) { return v == 0 ? -0.5 : (v - 1 == k ? 0.5 : 0.0); }

double eles_tets::eval_nodal_s_basis(int in_index, const hf_array<double> &loc, int in_n_spts)
{
  double l[4];
  tet_bary(loc, l);
  if (in_n_spts == 4) return l[in_index];
  if (in_n_spts != 10) { fail("Shape order not implemented yet, exiting"); return 0.0; } /* src/eles_tets.cpp:1075 */
  if (in_index < 4) return l[in_index] * (2. * l[in_index] - 1.);
  const int *e = tet_edge[in_index - 4];
  return 4. * l[e[0]] * l[e[1]];
}

void eles_tets::eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &loc, int in_n_spts)
{
  if (in_n_spts != 4 && in_n_spts != 10) { fail("Shape order not implemented yet, exiting"); return; }
  double l[4];
  tet_bary(loc, l);
  for (int k = 0; k < 3; k++)
  {
    if (in_n_spts == 4)
    {
      for (int v = 0; v < 4; v++) d(v, k) = tet_dbary(v, k);
      continue;
    }
    for (int v = 0; v < 4; v++) d(v, k) = (4. * l[v] - 1.) * tet_dbary(v, k);
    for (int q = 0; q < 6; q++)
    {
      const int a = tet_edge[q][0], b = tet_edge[q][1];
      d(4 + q, k) = 4. * (l[a] * tet_dbary(b, k) + l[b] * tet_dbary(a, k));
    }
  }
}

// =====================================================================================================================
// triangular prisms = triangle (x) line
// =====================================================================================================================
struct eles_pris::Impl
{
  TriNodal tri;
  std::vector<double> tr, ts;
};

eles_pris::eles_pris() : impl(new Impl) {}
eles_pris::~eles_pris() { delete impl; }

int eles_pris::setup_ele_type_specific()
{
  ele_type = 3;
  n_dims = 3;
  if (run_input->equation != 0) { fail("Equation not supported"); return 1; }
  if (run_input->over_int) { fail("eles_pris: over-integration is built for the tensor-product classes only (no cubature tables of prisms here)"); return 1; }
  // the reference's prism class builds no LES filter (src/eles_pris.cpp:134, commented out): the closures that filter the
  // solution would multiply by an empty matrix there
  if (run_input->LES && run_input->SGS_model >= 2) { fail("eles_pris: the reference builds no LES filter for prisms (SGS_model 0 or 1 only)"); return 1; }
  if (run_input->vcjh_scheme_tri < 0 || run_input->vcjh_scheme_tri > 4) { fail("VCJH triangular scheme not recognized"); return 1; } /* src/funcs.cpp:794 */
  if (run_input->upts_type_pri_tri != 0) { fail("eles_pris: point rule not implemented (rule 0, internal, is)"); return 1; }
  if (run_input->upts_type_pri_tri != run_input->fpts_type_tet) { fail("upts_type_pri_tri != fpts_type_tet"); return 1; }   /* src/eles_pris.cpp:116 */
  if (run_input->upts_type_pri_1d != run_input->upts_type_hexa) { fail("upts_type_pri_1d != upts_type_hexa"); return 1; } /* :120 */
  n_fields = 5;
  n_inters_per_ele = 5;
  const int p = order, N = p + 1;
  n_upts_tri = (p + 1) * (p + 2) / 2;
  n_upts_per_ele = n_upts_tri * N;
  std::string terr;
  const auto *tri = point_table("tri_inter", p, terr);
  if (!tri) { fail(terr); return 1; }
  hf_array<double> w;
  cubature_1d_nodes(run_input->upts_type_pri_1d, N, loc_1d_upts, w);
  if (run_input->loc_1d_upts_override.get_dim(0) == N) loc_1d_upts = run_input->loc_1d_upts_override;
  impl->tr.resize(n_upts_tri); impl->ts.resize(n_upts_tri);
  for (int i = 0; i < n_upts_tri; i++) { impl->tr[i] = (*tri)[i][0]; impl->ts[i] = (*tri)[i][1]; }
  // src/eles_pris.cpp:210-240
  loc_upts.setup(n_dims, n_upts_per_ele);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < n_upts_tri; j++)
    {
      loc_upts(0, n_upts_tri * i + j) = impl->tr[j];
      loc_upts(1, n_upts_tri * i + j) = impl->ts[j];
      loc_upts(2, n_upts_tri * i + j) = loc_1d_upts(i);
    }
  n_fpts_per_inter.setup(5);
  n_fpts_per_inter(0) = n_fpts_per_inter(1) = n_upts_tri;
  n_fpts_per_inter(2) = n_fpts_per_inter(3) = n_fpts_per_inter(4) = N * N;
  n_fpts_per_ele = 3 * N * N + 2 * n_upts_tri;
  // src/eles_pris.cpp:245-306
  tloc_fpts.setup(n_dims, n_fpts_per_ele);
  for (int i = 0; i < n_upts_tri; i++)
  {
    tloc_fpts(0, i) = impl->ts[i]; tloc_fpts(1, i) = impl->tr[i]; tloc_fpts(2, i) = -1.;
    tloc_fpts(0, n_upts_tri + i) = impl->tr[i]; tloc_fpts(1, n_upts_tri + i) = impl->ts[i]; tloc_fpts(2, n_upts_tri + i) = 1.;
  }
  const int off = 2 * n_upts_tri;
  for (int face = 0; face < 3; face++)
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++)
      {
        const int f = off + face * N * N + i * N + j;
        if (face == 0) { tloc_fpts(0, f) = loc_1d_upts(j); tloc_fpts(1, f) = -1; }
        else if (face == 1) { tloc_fpts(0, f) = loc_1d_upts(p - j); tloc_fpts(1, f) = loc_1d_upts(j); }
        else { tloc_fpts(0, f) = -1.; tloc_fpts(1, f) = loc_1d_upts(p - j); }
        tloc_fpts(2, f) = loc_1d_upts(i);
      }
  // src/eles_pris.cpp:551-590
  tnorm_fpts.setup(n_dims, n_fpts_per_ele);
  tnorm_fpts.initialize_to_zero();
  int fpt = -1;
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < n_fpts_per_inter(i); j++)
    {
      fpt++;
      if (i == 0) tnorm_fpts(2, fpt) = -1.;
      else if (i == 1) tnorm_fpts(2, fpt) = 1.;
      else if (i == 2) tnorm_fpts(1, fpt) = -1.;
      else if (i == 3) { tnorm_fpts(0, fpt) = 1. / std::sqrt(2.); tnorm_fpts(1, fpt) = 1. / std::sqrt(2.); }
      else tnorm_fpts(0, fpt) = -1.;
    }
  if (!impl->tri.build(p, impl->tr, impl->ts)) { fail("eles_pris: singular Vandermonde matrix"); return 1; }
  set_opp_0();
  set_opp_1();
  set_opp_2();
  set_opp_3();
  if (viscous)
  {
    set_opp_4();
    set_opp_5();
    set_opp_6();
  }
  return failed() ? 1 : 0;
}

// src/eles_pris.cpp:973-998: triangle nodal basis x 1-D Lagrange polynomial
double eles_pris::eval_nodal_basis(int in_index, const hf_array<double> &loc)
{
  const int it = in_index % n_upts_tri, i1 = in_index / n_upts_tri;
  return impl->tri.value(it, loc(0), loc(1)) * eval_lagrange(loc(2), i1, loc_1d_upts);
}

double eles_pris::eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &loc)
{
  const int it = in_index % n_upts_tri, i1 = in_index / n_upts_tri;
  if (in_cpnt == 2) return impl->tri.value(it, loc(0), loc(1)) * eval_d_lagrange(loc(2), i1, loc_1d_upts);
  double dr, ds;
  impl->tri.gradient(it, loc(0), loc(1), dr, ds);
  return (in_cpnt == 0 ? dr : ds) * eval_lagrange(loc(2), i1, loc_1d_upts);
}

// src/eles_pris.cpp:1504-1522
static int face0_map(int index, int p)
{
  for (int j = 0; j < p + 1; j++)
    for (int i = 0; i < p + 1 - j; i++)
      if (j * (p + 1) - (j - 1) * j / 2 + i == index) return i * (p + 1) - (i - 1) * i / 2 + j;
  return -1;
}

// src/eles_pris.cpp:1323-1410: the 1-D correction on the two triangular faces, the triangle's edge lifting on the three
// quadrilateral faces
void eles_pris::fill_opp_3(hf_array<double> &o3)
{
  const int p = order, N = p + 1, nt = n_upts_tri;
  std::vector<double> o3tri;
  tri_dg_lifting(p, impl->tri, impl->tr, impl->ts, loc_1d_upts, o3tri);
  {
    // the triangle's other VCJH members: opp_3_tri = Filt . opp_3_dg (get_opp_3_tri, src/funcs.cpp:629-641)
    double c = 0.0;
    std::string err;
    if (!vcjh_c_simplex(2, run_input->vcjh_scheme_tri, p, run_input->c_tri, c, err)) { fail(err); return; }
    if (c != 0.0)
    {
      std::vector<Mat> D(2, Mat((size_t)nt * nt));
      Mat V((size_t)nt * nt), Filt;
      for (int i = 0; i < nt; i++)
        for (int j = 0; j < nt; j++)
        {
          impl->tri.gradient(j, impl->tr[i], impl->ts[i], D[0][(size_t)i * nt + j], D[1][(size_t)i * nt + j]);
          V[(size_t)i * nt + j] = simplex2d(impl->tr[i], impl->ts[i], impl->tri.mi[j], impl->tri.mj[j]);
        }
      if (!vcjh_filter(2, p, nt, D, V, c, Filt)) { fail("eles_pris: singular VCJH filter matrix"); return; }
      const int ne = 3 * N; // edge flux points of the triangle; o3tri(ut, i) = o3tri[ut + nt * i]
      std::vector<double> dg = o3tri;
      for (int j = 0; j < nt; j++)
        for (int i = 0; i < ne; i++)
        {
          double t = 0.0;
          for (int k = 0; k < nt; k++) t += Filt[(size_t)j * nt + k] * dg[(size_t)k + (size_t)nt * i];
          o3tri[(size_t)j + (size_t)nt * i] = t;
        }
    }
  }
  double eta = run_input->eta_pri;
  if (run_input->vcjh_scheme_pri_1d != 0 && !compute_eta(run_input->vcjh_scheme_pri_1d, p, eta)) { fail("Invalid VCJH scheme"); return; }
  for (int upt = 0; upt < n_upts_per_ele; upt++)
  {
    const int u1 = upt / nt, ut = upt % nt;
    const double z = loc_upts(2, upt);
    for (int idx = 0; idx < n_fpts_per_ele; idx++)
    {
      double v = 0.0;
      if (idx < nt)
        v = (face0_map(idx, p) == ut) ? -eval_d_vcjh_1d(z, 0, p, eta) : 0.0;
      else if (idx < 2 * nt)
        v = (idx - nt == ut) ? eval_d_vcjh_1d(z, 1, p, eta) : 0.0;
      else
      {
        const int edge = (idx - 2 * nt) / (N * N), ff = (idx - 2 * nt) % (N * N);
        v = (ff / N == u1) ? o3tri[(size_t)ut + (size_t)nt * (edge * N + ff % N)] : 0.0;
      }
      o3(upt, idx) = v;
    }
  }
}

// The 15-node prism of src/eles_pris.cpp:1115-1146 as products of triangle and line functions: with the triangle's barycentric
// coordinates m_0 = -(r+s)/2, m_1 = (r+1)/2, m_2 = (s+1)/2 and the quadratic line functions q_-(t) = t (t-1)/2, q_0 = 1 - t^2,
// q_+ = t (t+1)/2:  nodes 0-2 / 3-5 (vertices below / above): m_v (2 m_v - 1) q_-/+;  6-8 / 12-14 (edges (0,1), (1,2), (0,2) of
// the lower / upper triangle): 4 m_a m_b q_-/+;  9-11 (mid-height of the vertical edges): m_v q_0.
static const int pri_edge[3][2] = {{0, 1}, {1, 2}, {0, 2}};
static inline void pri15(const hf_array<double> &loc, int idx, double &tri, double dtri[2], double &lin, double &dlin)
{
  const double r = loc(0), s = loc(1), t = loc(2);
  const double m[3] = {-0.5 * (r + s), 0.5 * (r + 1.), 0.5 * (s + 1.)};
  const double dm[3][2] = {{-0.5, -0.5}, {0.5, 0.0}, {0.0, 0.5}};
  const double q[3] = {0.5 * t * (t - 1.), 1. - t * t, 0.5 * t * (t + 1.)}, dq[3] = {t - 0.5, -2. * t, t + 0.5};
  int level, kind, which; // level: 0 below, 1 mid-height, 2 above; kind 0 vertex, 1 edge
  if (idx < 3) { level = 0; kind = 0; which = idx; }
  else if (idx < 6) { level = 2; kind = 0; which = idx - 3; }
  else if (idx < 9) { level = 0; kind = 1; which = idx - 6; }
  else if (idx < 12) { level = 1; kind = 0; which = idx - 9; }
  else { level = 2; kind = 1; which = idx - 12; }
  lin = q[level]; dlin = dq[level];
  if (level == 1)
  {
    tri = m[which];
    for (int k = 0; k < 2; k++) dtri[k] = dm[which][k];
  }
  else if (kind == 0)
  {
    tri = m[which] * (2. * m[which] - 1.);
    for (int k = 0; k < 2; k++) dtri[k] = (4. * m[which] - 1.) * dm[which][k];
  }
  else
  {
    const int a = pri_edge[which][0], b = pri_edge[which][1];
    tri = 4. * m[a] * m[b];
    for (int k = 0; k < 2; k++) dtri[k] = 4. * (m[a] * dm[b][k] + m[b] * dm[a][k]);
  }
}

// src/eles_pris.cpp:609-730.  The hierarchical prism basis: orthonormal triangle mode (i, j) times the Legendre polynomial P_k
// along the line, numbered by i + j + k, then k, then j (src/eles_pris.cpp:1238-1283); norm 2 / (2k + 1); the filter damps the
// triangle's and the line's degree separately; the sensor's highest modes have i + j == order or k == order.
int eles_pris::set_shock_capture_operators()
{
  if (run_input->shock_cap != 1) { fail("Shock capturing method not implemented."); return 1; }
  const int n = n_upts_per_ele, p = order;
  std::vector<double> V((size_t)n * n), sigma(n), norm(n);
  std::vector<int> high(n);
  const double eta_c = (double)run_input->expf_cutoff / (double)p;
  int m = 0;
  for (int l = 0; l <= 2 * p; l++)
    for (int k = 0; k <= l; k++)
      for (int j = 0; j <= l - k; j++)
      {
        const int i = l - k - j;
        if (k > p || i + j > p) continue;
        for (int u = 0; u < n; u++)
          V[(size_t)u * n + m] = simplex2d(loc_upts(0, u), loc_upts(1, u), i, j) * eval_legendre(loc_upts(2, u), k);
        sigma[m] = expf_sigma(run_input, (double)(i + j) / (double)p, eta_c) * expf_sigma(run_input, (double)k / (double)p, eta_c);
        norm[m] = 2.0 / (2.0 * k + 1.0);
        high[m] = (i + j == p || k == p) ? 1 : 0;
        m++;
      }
  if (m != n) { fail("eles_pris: mode count"); return 1; }
  std::string e;
  if (modal_shock_operators(this, n, V, sigma, norm, high, e)) { fail("eles_pris: " + e); return 1; }
  return 0;
}

// the smallest of the three vertical edges and of the incircle diameters of the lower and the upper triangle (src/eles_pris.cpp:1535-1557)
double eles_pris::calc_h_ref_specific(int in_ele)
{
  auto len = [&](int p, int q) {
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += std::pow(shape(k, p, in_ele) - shape(k, q, in_ele), 2.0);
    return std::sqrt(s);
  };
  double length[5];
  for (int i = 0; i < 3; i++) length[i] = len(i, i + 3);
  for (int i = 3; i < 5; i++)
  {
    const int d = (i - 3) * 3;
    const double a = len(d, d + 1), b = len(d + 1, d + 2), c = len(d + 2, d), s = 0.5 * (a + b + c);
    length[i] = 2 * std::sqrt(((s - a) * (s - b) * (s - c)) / s);
  }
  double m = length[0];
  for (int i = 1; i < 5; i++) m = std::fmin(m, length[i]);
  return m;
}

double eles_pris::eval_nodal_s_basis(int in_index, const hf_array<double> &loc, int in_n_spts)
{
  if (in_n_spts == 15)
  {
    double tri, dtri[2], lin, dlin;
    pri15(loc, in_index, tri, dtri, lin, dlin);
    return tri * lin;
  }
  if (in_n_spts != 6) { fail("Shape order not implemented yet, exiting"); return 0.0; }
  const double r = loc(0), s = loc(1), t = loc(2);
  switch (in_index) /* src/eles_pris.cpp:1100-1112 */
  {
  case 0: return 1. / 4. * (r + s) * (t - 1.);
  case 1: return -1. / 4. * (r + 1.) * (t - 1.);
  case 2: return -1. / 4. * (s + 1.) * (t - 1.);
  case 3: return -1. / 4. * (r + s) * (t + 1.);
  case 4: return 1. / 4. * (r + 1.) * (t + 1.);
  default: return 1. / 4. * (s + 1.) * (t + 1.);
  }
}

void eles_pris::eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &loc, int in_n_spts)
{
  if (in_n_spts == 15)
  {
    for (int i = 0; i < 15; i++)
    {
      double tri, dtri[2], lin, dlin;
      pri15(loc, i, tri, dtri, lin, dlin);
      d(i, 0) = dtri[0] * lin; d(i, 1) = dtri[1] * lin; d(i, 2) = tri * dlin;
    }
    return;
  }
  if (in_n_spts != 6) { fail("Shape order not implemented yet, exiting"); return; }
  const double r = loc(0), s = loc(1), t = loc(2);
  // derivatives of the six functions above
  d(0, 0) = 1. / 4. * (t - 1.);  d(0, 1) = 1. / 4. * (t - 1.);  d(0, 2) = 1. / 4. * (r + s);
  d(1, 0) = -1. / 4. * (t - 1.); d(1, 1) = 0.;                  d(1, 2) = -1. / 4. * (r + 1.);
  d(2, 0) = 0.;                  d(2, 1) = -1. / 4. * (t - 1.); d(2, 2) = -1. / 4. * (s + 1.);
  d(3, 0) = -1. / 4. * (t + 1.); d(3, 1) = -1. / 4. * (t + 1.); d(3, 2) = -1. / 4. * (r + s);
  d(4, 0) = 1. / 4. * (t + 1.);  d(4, 1) = 0.;                  d(4, 2) = 1. / 4. * (r + 1.);
  d(5, 0) = 0.;                  d(5, 1) = 1. / 4. * (t + 1.);  d(5, 2) = 1. / 4. * (s + 1.);
}
