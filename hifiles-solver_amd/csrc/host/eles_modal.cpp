// eles_modal.cpp -- the modal-basis operators of the tensor-product element classes (hexes, quads): what the
// reference builds in eles_hexas / eles_quads ::set_vandermonde1D/3D (src/eles_hexas.cpp:926-951), set_exp_filter
// (:953-989), calc_norm_basis (:991-1004), set_over_int (:1096-1129) and their eles_quads twins
// (src/eles_quads.cpp:759-959), i.e. the matrices behind shock capturing and over-integration
// (row a5).
//
// The reference assembles the n_upts x n_upts Vandermonde matrix of the hierarchical tensor Legendre basis, inverts it
// by full-pivot Gaussian elimination (src/funcs.cpp:2751) and multiplies dense matrices.  Here every operator is
// assembled from its 1-D factor: the basis is a tensor product, so
//     vandermonde      = P (V1 x V1 x V1),        inv_vandermonde = (W1 x W1 x W1) P^T,      W1 = V1^-1,
//     exp_filter       = E1 x E1 x E1,            E1 = V1 diag(sigma) W1,
//     over_int_filter  = F1 x F1 x F1,            F1 = V1 diag((2m+1)/2) [P_m(c) w_c],
//     opp_over_int_cubpts = I1 x I1 x I1,         I1 = l_i(c)
// (P: the reference's "hierarchical" mode order, modes sorted by the sum of their 1-D degrees).  Same matrices to
// rounding (tests/test_host_setup_vs_golden.py holds them to 1e-12 of the reference's), O(N^2) work per factor and
// a 1-D inverse of an (order+1)^2 matrix instead of a 125 x 125 one.
#include <cmath>
#include <vector>

#include "basis.hpp"
#include "eles.hpp"

namespace
{
// inverse of a small dense matrix (n x n, column-major a(i,j) = a[i + n*j]), Gauss-Jordan with row pivoting
bool invert_small(int n, std::vector<double> a, std::vector<double> &inv)
{
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) inv[i + (size_t)n * i] = 1.0;
  for (int c = 0; c < n; c++)
  {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (std::fabs(a[r + (size_t)n * c]) > std::fabs(a[piv + (size_t)n * c])) piv = r;
    if (a[piv + (size_t)n * c] == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < n; j++)
      {
        std::swap(a[c + (size_t)n * j], a[piv + (size_t)n * j]);
        std::swap(inv[c + (size_t)n * j], inv[piv + (size_t)n * j]);
      }
    const double d = 1.0 / a[c + (size_t)n * c];
    for (int j = 0; j < n; j++)
    {
      a[c + (size_t)n * j] *= d;
      inv[c + (size_t)n * j] *= d;
    }
    for (int r = 0; r < n; r++)
    {
      if (r == c) continue;
      const double f = a[r + (size_t)n * c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++)
      {
        a[r + (size_t)n * j] -= f * a[c + (size_t)n * j];
        inv[r + (size_t)n * j] -= f * inv[c + (size_t)n * j];
      }
    }
  }
  return true;
}
} // namespace

// degrees (i, j[, k]) of every mode in the reference's hierarchical order: sorted by the sum of the degrees, the
// last direction's degree ascending inside a sum, then the second's (eval_legendre_basis_3D_hierarchical,
// src/eles_hexas.cpp:1364-1403; 2-D: src/eles_quads.cpp:1116-1150)
void eles::tensor_modes(hf_array<int> &deg) const
{
  const int p = order;
  deg.setup(n_dims, n_upts_per_ele);
  int mode = 0;
  if (n_dims == 3)
  {
    for (int l = 0; l < 3 * p + 1; l++)
      for (int k = 0; k < l + 1; k++)
        for (int j = 0; j < l - k + 1; j++)
        {
          const int i = l - k - j;
          if (i <= p && j <= p && k <= p)
          {
            deg(0, mode) = i;
            deg(1, mode) = j;
            deg(2, mode) = k;
            mode++;
          }
        }
  }
  else
  {
    for (int k = 0; k < 2 * p + 1; k++)
      for (int j = 0; j < k + 1; j++)
      {
        const int i = k - j;
        if (i <= p && j <= p)
        {
          deg(0, mode) = i;
          deg(1, mode) = j;
          mode++;
        }
      }
  }
}

// out(row, col) = prod_d f(r_d, c_d): rows / columns are tensor indices with the first direction fastest
// (solution points upt = k + N j + N^2 i <-> (x_k, y_j, z_i), cubature points likewise), f is nr1 x nc1 column-major
static void kron_fill(int nd, int nr1, int nc1, const std::vector<double> &f, hf_array<double> &out)
{
  int nr = 1, nc = 1;
  for (int d = 0; d < nd; d++)
  {
    nr *= nr1;
    nc *= nc1;
  }
  out.setup(nr, nc);
  for (int c = 0; c < nc; c++)
    for (int r = 0; r < nr; r++)
    {
      double v = 1.0;
      int rr = r, cc = c;
      for (int d = 0; d < nd; d++)
      {
        v *= f[(rr % nr1) + (size_t)nr1 * (cc % nc1)];
        rr /= nr1;
        cc /= nc1;
      }
      out(r, c) = v;
    }
}

// The LES filter of the similarity-type closures (SGS_model 2, 3, 4): a 1-D filter over the solution points of one
// direction, F1(target, source), and its tensor product (src/eles_hexas.cpp:583-790, src/eles_quads.cpp:428-622).
//   filter_type 0, order >= 2: the high-order-commuting filter of Vasilyev -- for every target point i the weights
//     w_j = F1(j, i) (NB: stored by column, as the reference does) solve the moment conditions
//        sum_j w_j                                   = 1
//        sum_j w_j cos(pi k_c b_ji)                  = exp(-pi^2/24)            (Gaussian transfer function at k_c)
//        sum_j w_j (-b_ji pi sin(pi k_c b_ji))       = -exp(-pi^2/24) pi^2 / (12 k_c)   (its slope)
//        sum_j w_j b_ji^(m+1)                        = 0,   m = 3 .. N-1               (vanishing moments)
//     with b_ji = (x_j - x_i)/dlt, dlt = 2/order, k_c = 1/filter_ratio; at the centre point of an odd N the third
//     condition is replaced by sum_j w_j b_ji^3 = 0.
//   filter_type 1: discrete Gaussian, F1(i, j) = w_j exp(-6 (k_c b_ij)^2) normalised over j (Gauss weights w_j).
//   filter_type 2: modal, V1 diag(exp(-(2 m / N)^2 / 48)) V1^-1.
//   otherwise (and type 0 below order 2): the element average 1/N.
int eles::compute_filter_upts()
{
  const int N = order + 1;
  const double pi = 3.141592653589793238462643383279502884;
  const double k_c = 1.0 / run_input->filter_ratio, dlt = 2.0 / order;
  std::vector<double> F1((size_t)N * N, 0.0);
  auto beta = [&](int j, int i) { return (loc_1d_upts(j) - loc_1d_upts(i)) / dlt; };
  const int type = run_input->filter_type;
  if (type == 0 && N >= 3)
  {
    const int centre = (N % 2 == 1) ? (N + 1) / 2 - 1 : -1;
    for (int i = 0; i < N; i++)
    {
      // M(k, j): condition k, weight j
      std::vector<double> M((size_t)N * N), Minv, rhs(N, 0.0);
      rhs[0] = 1.0;
      rhs[1] = std::exp(-pi * pi / 24.0);
      rhs[2] = (i == centre) ? 0.0 : -rhs[1] * pi * pi / k_c / 12.0;
      for (int j = 0; j < N; j++)
      {
        const double b = beta(j, i);
        M[0 + (size_t)N * j] = 1.0;
        M[1 + (size_t)N * j] = std::cos(pi * k_c * b);
        M[2 + (size_t)N * j] = (i == centre) ? b * b * b : -b * pi * std::sin(pi * k_c * b);
        for (int k = 3; k < N; k++) M[k + (size_t)N * j] = std::pow(b, k + 1);
      }
      if (!invert_small(N, M, Minv))
      {
        fail("LES filter: singular moment system");
        return 1;
      }
      for (int j = 0; j < N; j++)
      {
        double w = 0.0;
        for (int k = 0; k < N; k++) w += Minv[j + (size_t)N * k] * rhs[k];
        F1[j + (size_t)N * i] = w;
      }
    }
  }
  else if (type == 1)
  {
    hf_array<double> xg, wg;
    cubature_1d_nodes(0, N, xg, wg);
    for (int i = 0; i < N; i++)
    {
      double norm = 0.0;
      for (int j = 0; j < N; j++)
      {
        const double a = k_c * beta(i, j);
        F1[i + (size_t)N * j] = wg(j) * std::exp(-6.0 * a * a);
        norm += F1[i + (size_t)N * j];
      }
      for (int j = 0; j < N; j++) F1[i + (size_t)N * j] /= norm;
    }
  }
  else if (type == 2)
  {
    std::vector<double> V1((size_t)N * N), W1;
    for (int i = 0; i < N; i++)
      for (int m = 0; m < N; m++) V1[i + (size_t)N * m] = eval_legendre(loc_1d_upts(i), m);
    if (!invert_small(N, V1, W1))
    {
      fail("singular 1-D Vandermonde matrix");
      return 1;
    }
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++)
      {
        double v = 0.0;
        for (int m = 0; m < N; m++)
        {
          const double eta = m / double(N);
          v += V1[i + (size_t)N * m] * std::exp(-(2.0 * eta) * (2.0 * eta) / 48.0) * W1[m + (size_t)N * j];
        }
        F1[i + (size_t)N * j] = v;
      }
  }
  else
    for (auto &v : F1) v = 1.0 / N;
  filter_upts_1D.setup(N, N);
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) filter_upts_1D(i, j) = F1[i + (size_t)N * j];
  kron_fill(n_dims, N, N, F1, filter_upts);
  return 0;
}

int eles::set_shock_capture_operators()
{
  const int N = order + 1, nu = n_upts_per_ele;
  // 1-D Vandermonde V1(point, degree) = P_degree(x_point) and its inverse (set_vandermonde1D)
  std::vector<double> V1((size_t)N * N), W1;
  for (int i = 0; i < N; i++)
    for (int m = 0; m < N; m++) V1[i + (size_t)N * m] = eval_legendre(loc_1d_upts(i), m);
  if (!invert_small(N, V1, W1))
  {
    fail("singular 1-D Vandermonde matrix");
    return 1;
  }
  hf_array<int> deg;
  tensor_modes(deg);
  // set_vandermonde3D / 2D and its inverse
  vandermonde.setup(nu, nu);
  inv_vandermonde.setup(nu, nu);
  for (int pt = 0; pt < nu; pt++)
    for (int mode = 0; mode < nu; mode++)
    {
      double v = 1.0, w = 1.0;
      int r = pt;
      for (int d = 0; d < n_dims; d++)
      {
        v *= V1[(r % N) + (size_t)N * deg(d, mode)];
        w *= W1[deg(d, mode) + (size_t)N * (r % N)];
        r /= N;
      }
      vandermonde(pt, mode) = v;
      inv_vandermonde(mode, pt) = w;
    }
  // calc_norm_basis and the highest modes of the Persson sensor (shock_det_persson: any degree == order)
  norm_basis_persson.setup(nu);
  persson_high_modes.setup(nu);
  for (int mode = 0; mode < nu; mode++)
  {
    double nrm = 1.0;
    int high = 0;
    for (int d = 0; d < n_dims; d++)
    {
      nrm *= 2.0 / (2.0 * deg(d, mode) + 1.0);
      if (deg(d, mode) == order) high = 1;
    }
    norm_basis_persson(mode) = nrm;
    persson_high_modes(mode) = high;
  }
  // set_exp_filter: sigma(degree) per direction, E1 = V1 diag(sigma) W1
  if (run_input->shock_cap == 1)
  {
    const double eta_c = (double)run_input->expf_cutoff / (double)order;
    std::vector<double> E1((size_t)N * N, 0.0);
    for (int m = 0; m < N; m++)
    {
      const double eta = (double)m / (double)order;
      double sigma = 1.0;
      if (eta > eta_c) sigma = std::exp(-run_input->expf_fac * std::pow((eta - eta_c) / (1. - eta_c), run_input->expf_order));
      for (int a = 0; a < N; a++)
        for (int b = 0; b < N; b++) E1[a + (size_t)N * b] += V1[a + (size_t)N * m] * sigma * W1[m + (size_t)N * b];
    }
    kron_fill(n_dims, N, N, E1, exp_filter);
  }
  return 0;
}

int eles::set_over_int()
{
  const int N = order + 1, Nc = run_input->over_int_order + 1;
  if (Nc < 1 || Nc > 16)
  {
    fail("cubature order not implemented."); // src/cubature_1d.cpp:84
    return 1;
  }
  // set_volume_cubpts: tensor Gauss points, first direction fastest (src/cubature_hexa.cpp:49-74)
  hf_array<double> c1, w1;
  cubature_1d_nodes(0, Nc, c1, w1);
  int n_cub = 1;
  for (int d = 0; d < n_dims; d++) n_cub *= Nc;
  loc_over_int_cubpts.setup(n_dims, n_cub);
  weight_over_int_cubpts.setup(n_cub);
  for (int q = 0; q < n_cub; q++)
  {
    int r = q;
    double w = 1.0;
    for (int d = 0; d < n_dims; d++)
    {
      loc_over_int_cubpts(d, q) = c1(r % Nc);
      w *= w1(r % Nc);
      r /= Nc;
    }
    weight_over_int_cubpts(q) = w;
  }
  // set_opp_volume_cubpts: nodal basis at the cubature points, I1(c, i) = l_i(x_c)
  std::vector<double> I1((size_t)Nc * N), F1((size_t)N * Nc, 0.0);
  for (int c = 0; c < Nc; c++)
    for (int i = 0; i < N; i++) I1[c + (size_t)Nc * i] = eval_lagrange(c1(c), i, loc_1d_upts);
  kron_fill(n_dims, Nc, N, I1, opp_over_int_cubpts);
  // L2 projection on the modal basis, then back to the solution points: F1(a, c) = sum_m P_m(x_a) (2m+1)/2 P_m(x_c) w_c
  for (int a = 0; a < N; a++)
    for (int c = 0; c < Nc; c++)
    {
      double s = 0.0;
      for (int m = 0; m < N; m++) s += eval_legendre(loc_1d_upts(a), m) * (eval_legendre(c1(c), m) / (2.0 / (2.0 * m + 1.0)) * w1(c));
      F1[a + (size_t)N * c] = s;
    }
  kron_fill(n_dims, N, Nc, F1, over_int_filter);
  return 0;
}

// ---- plot points: p_res equispaced points per edge, first direction fastest (src/eles_hexas.cpp:498-521,
// src/eles_quads.cpp:367-385), and the nodal basis there (eles::set_opp_p, src/eles.cpp:3600-3621)
void eles::set_loc_ppts()
{
  p_res = run_input->p_res;
  n_ppts_per_ele = 1;
  for (int d = 0; d < n_dims; d++) n_ppts_per_ele *= p_res;
  loc_ppts.setup(n_dims, n_ppts_per_ele);
  for (int ppt = 0; ppt < n_ppts_per_ele; ppt++)
  {
    int r = ppt;
    for (int d = 0; d < n_dims; d++)
    {
      loc_ppts(d, ppt) = -1.0 + ((2.0 * (r % p_res)) / (1.0 * (p_res - 1)));
      r /= p_res;
    }
  }
}

void eles::set_opp_p()
{
  hf_array<double> loc(n_dims);
  opp_p.setup(n_ppts_per_ele, n_upts_per_ele);
  for (int i = 0; i < n_upts_per_ele; i++)
    for (int j = 0; j < n_ppts_per_ele; j++)
    {
      for (int k = 0; k < n_dims; k++) loc(k) = loc_ppts(k, j);
      opp_p(j, i) = eval_nodal_basis(i, loc);
    }
}

int eles::calc_disu_ppts_all()
{
  if (!dev) { fail("element block is not on the device"); return 1; }
  disu_ppts.setup(n_ppts_per_ele, n_eles, n_fields);
  if (hfx_eles_calc_disu_ppts(dev, disu_ppts.get_ptr_cpu())) { fail(hfx_last_error()); return 1; }
  return 0;
}

void eles::calc_disu_ppts(int in_ele, hf_array<double> &out_disu_ppts)
{
  // the reference interpolates one element per call; here the whole block is interpolated once and read per element
  if (disu_ppts.get_dim(0) != n_ppts_per_ele && calc_disu_ppts_all()) return;
  for (int k = 0; k < n_fields; k++)
    for (int j = 0; j < n_ppts_per_ele; j++) out_disu_ppts(j, k) = disu_ppts(j, in_ele, k);
}
