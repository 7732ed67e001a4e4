/* oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See oracle.h.
 *
 * CPU restatement of the reference's hot path.  Every function cites the
 * reference file:line it follows; loop nests and summation orders are kept so
 * that results agree with the genuine reference to rounding (the fixtures in
 * tests/golden pin that).  OpenMP only distributes independent columns /
 * points / faces over threads; it never changes a summation order.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void orc_set_threads(int n)
{
  g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}

#define MAXF 6 /* n_fields <= 5 here; room for an SA field */
#define MAXD 3

/* ------------------------------------------------------------------------ */
/* src/funcs.cpp:49-123 : C := alpha*A*B + beta*C, column-major, no transposes.
 * For each column j, for l ascending: C(:,j) += (alpha*B(l,j)) * A(:,l).    */
void orc_dgemm(int Arows, int Bcols, int Acols, double alpha, double beta,
               const double *a, const double *b, double *c)
{
  if (Arows == 0 || Bcols == 0 || ((alpha == 0. || Acols == 0) && beta == 1.)) return;
  if (alpha == 0.)
  {
    for (long j = 0; j < Bcols; j++)
      for (int i = 0; i < Arows; i++)
        c[i + j * Arows] = (beta == 0.) ? 0. : beta * c[i + j * Arows];
    return;
  }
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (long j = 0; j < Bcols; j++)
  {
    double *cj = c + j * (long)Arows;
    const double *bj = b + j * (long)Acols;
    if (beta == 0.)
      for (int i = 0; i < Arows; i++) cj[i] = 0.;
    else if (beta != 1.)
      for (int i = 0; i < Arows; i++) cj[i] = beta * cj[i];
    for (int l = 0; l < Acols; l++)
    {
      double temp = alpha * bj[l];
      const double *al = a + (long)l * Arows;
      for (int i = 0; i < Arows; i++) cj[i] += temp * al[i];
    }
  }
}

/* ------------------------------------------------------------------------ */
/* src/flux.cpp:33-72 (2-D), :74-125 (3-D); f(k,m) = f[k + n_fields*m]      */
void orc_calc_invf(int n_dims, double gamma, const double *u, double *f)
{
  if (n_dims == 2)
  {
    const int nf = 4;
    double vx = u[1] / u[0];
    double vy = u[2] / u[0];
    double p = (gamma - 1.0) * (u[3] - (0.5 * u[0] * ((vx * vx) + (vy * vy))));
    f[0 + nf * 0] = u[1];
    f[1 + nf * 0] = p + (u[1] * vx);
    f[2 + nf * 0] = u[2] * vx;
    f[3 + nf * 0] = vx * (u[3] + p);
    f[0 + nf * 1] = u[2];
    f[1 + nf * 1] = u[1] * vy;
    f[2 + nf * 1] = p + (u[2] * vy);
    f[3 + nf * 1] = vy * (u[3] + p);
  }
  else
  {
    const int nf = 5;
    double vx = u[1] / u[0];
    double vy = u[2] / u[0];
    double vz = u[3] / u[0];
    double p = (gamma - 1.0) * (u[4] - (0.5 * u[0] * ((vx * vx) + (vy * vy) + (vz * vz))));
    f[0 + nf * 0] = u[1];
    f[1 + nf * 0] = p + (u[1] * vx);
    f[2 + nf * 0] = u[2] * vx;
    f[3 + nf * 0] = u[3] * vx;
    f[4 + nf * 0] = vx * (u[4] + p);
    f[0 + nf * 1] = u[2];
    f[1 + nf * 1] = u[1] * vy;
    f[2 + nf * 1] = p + (u[2] * vy);
    f[3 + nf * 1] = u[3] * vy;
    f[4 + nf * 1] = vy * (u[4] + p);
    f[0 + nf * 2] = u[3];
    f[1 + nf * 2] = u[1] * vz;
    f[2 + nf * 2] = u[2] * vz;
    f[3 + nf * 2] = p + (u[3] * vz);
    f[4 + nf * 2] = vz * (u[4] + p);
  }
}

/* src/flux.cpp:129-254 (2-D), :257-422 (3-D); RANS off (mu_t = 0).
 * grad_u(k,m) = g[k + n_fields*m]                                           */
void orc_calc_visf(int n_dims, const orc_params *P, const double *u, const double *g, double *f)
{
  const double gamma = P->gamma;
  if (n_dims == 2)
  {
    const int nf = 4;
    double rho = u[0], mom_x = u[1], mom_y = u[2], ene = u[3];
    double rho_dx = g[0], mom_x_dx = g[1], mom_y_dx = g[2], ene_dx = g[3];
    double rho_dy = g[0 + nf], mom_x_dy = g[1 + nf], mom_y_dy = g[2 + nf], ene_dy = g[3 + nf];
    double uu = mom_x / rho, v = mom_y / rho;
    double inte = ene / rho - 0.5 * (uu * uu + v * v);
    double rt_ratio = (gamma - 1.0) * inte / (P->rt_inf);
    double mu = (P->mu_inf) * pow(rt_ratio, 1.5) * (1. + (P->c_sth)) / (rt_ratio + (P->c_sth));
    mu = mu + P->fix_vis * (P->mu_inf - mu);
    double mu_t = 0.0;
    double du_dx = (mom_x_dx - rho_dx * uu) / rho;
    double du_dy = (mom_x_dy - rho_dy * uu) / rho;
    double dv_dx = (mom_y_dx - rho_dx * v) / rho;
    double dv_dy = (mom_y_dy - rho_dy * v) / rho;
    double dke_dx = 0.5 * (uu * uu + v * v) * rho_dx + rho * (uu * du_dx + v * dv_dx);
    double dke_dy = 0.5 * (uu * uu + v * v) * rho_dy + rho * (uu * du_dy + v * dv_dy);
    double de_dx = (ene_dx - dke_dx - rho_dx * inte) / rho;
    double de_dy = (ene_dy - dke_dy - rho_dy * inte) / rho;
    double diag = (du_dx + dv_dy) / 3.0;
    double tauxx = 2.0 * (mu + mu_t) * (du_dx - diag);
    double tauxy = (mu + mu_t) * (du_dy + dv_dx);
    double tauyy = 2.0 * (mu + mu_t) * (dv_dy - diag);
    /* mu_t/prandtl_t term is exactly 0 with RANS off */
    f[0] = 0.0;
    f[1] = -tauxx;
    f[2] = -tauxy;
    f[3] = -(uu * tauxx + v * tauxy + (mu / P->prandtl + 0.0) * (gamma)*de_dx);
    f[0 + nf] = 0.0;
    f[1 + nf] = -tauxy;
    f[2 + nf] = -tauyy;
    f[3 + nf] = -(uu * tauxy + v * tauyy + (mu / P->prandtl + 0.0) * (gamma)*de_dy);
  }
  else
  {
    const int nf = 5;
    double rho = u[0], mom_x = u[1], mom_y = u[2], mom_z = u[3], ene = u[4];
    double rho_dx = g[0], mom_x_dx = g[1], mom_y_dx = g[2], mom_z_dx = g[3], ene_dx = g[4];
    double rho_dy = g[0 + nf], mom_x_dy = g[1 + nf], mom_y_dy = g[2 + nf], mom_z_dy = g[3 + nf], ene_dy = g[4 + nf];
    double rho_dz = g[0 + 2 * nf], mom_x_dz = g[1 + 2 * nf], mom_y_dz = g[2 + 2 * nf], mom_z_dz = g[3 + 2 * nf],
           ene_dz = g[4 + 2 * nf];
    double uu = mom_x / rho, v = mom_y / rho, w = mom_z / rho;
    double inte = ene / rho - 0.5 * (uu * uu + v * v + w * w);
    double rt_ratio = (gamma - 1.0) * inte / (P->rt_inf);
    double mu = (P->mu_inf) * pow(rt_ratio, 1.5) * (1 + (P->c_sth)) / (rt_ratio + (P->c_sth));
    mu = mu + P->fix_vis * (P->mu_inf - mu);
    double mu_t = 0.0;
    double du_dx = (mom_x_dx - rho_dx * uu) / rho;
    double du_dy = (mom_x_dy - rho_dy * uu) / rho;
    double du_dz = (mom_x_dz - rho_dz * uu) / rho;
    double dv_dx = (mom_y_dx - rho_dx * v) / rho;
    double dv_dy = (mom_y_dy - rho_dy * v) / rho;
    double dv_dz = (mom_y_dz - rho_dz * v) / rho;
    double dw_dx = (mom_z_dx - rho_dx * w) / rho;
    double dw_dy = (mom_z_dy - rho_dy * w) / rho;
    double dw_dz = (mom_z_dz - rho_dz * w) / rho;
    double dke_dx = 0.5 * (uu * uu + v * v + w * w) * rho_dx + rho * (uu * du_dx + v * dv_dx + w * dw_dx);
    double dke_dy = 0.5 * (uu * uu + v * v + w * w) * rho_dy + rho * (uu * du_dy + v * dv_dy + w * dw_dy);
    double dke_dz = 0.5 * (uu * uu + v * v + w * w) * rho_dz + rho * (uu * du_dz + v * dv_dz + w * dw_dz);
    double de_dx = (ene_dx - dke_dx - rho_dx * inte) / rho;
    double de_dy = (ene_dy - dke_dy - rho_dy * inte) / rho;
    double de_dz = (ene_dz - dke_dz - rho_dz * inte) / rho;
    double diag = (du_dx + dv_dy + dw_dz) / 3.0;
    double tauxx = 2.0 * (mu + mu_t) * (du_dx - diag);
    double tauyy = 2.0 * (mu + mu_t) * (dv_dy - diag);
    double tauzz = 2.0 * (mu + mu_t) * (dw_dz - diag);
    double tauxy = (mu + mu_t) * (du_dy + dv_dx);
    double tauxz = (mu + mu_t) * (du_dz + dw_dx);
    double tauyz = (mu + mu_t) * (dv_dz + dw_dy);
    f[0] = 0.0;
    f[1] = -tauxx;
    f[2] = -tauxy;
    f[3] = -tauxz;
    f[4] = -(uu * tauxx + v * tauxy + w * tauxz + (mu / P->prandtl + 0.0) * (gamma)*de_dx);
    f[0 + nf] = 0.0;
    f[1 + nf] = -tauxy;
    f[2 + nf] = -tauyy;
    f[3 + nf] = -tauyz;
    f[4 + nf] = -(uu * tauxy + v * tauyy + w * tauyz + (mu / P->prandtl + 0.0) * (gamma)*de_dy);
    f[0 + 2 * nf] = 0.0;
    f[1 + 2 * nf] = -tauxz;
    f[2 + 2 * nf] = -tauyz;
    f[3 + 2 * nf] = -tauzz;
    f[4 + 2 * nf] = -(uu * tauxz + v * tauyz + w * tauzz + (mu / P->prandtl + 0.0) * (gamma)*de_dz);
  }
}

/* normal flux fn_x(k) = sum_l f_x(k,l)*norm(l), starting from 0 (the BLAS=NO
 * branches of src/inters.cpp:288-297, :463-473)                              */
static void normal_fluxes(int nd, int nf, const double *fl, const double *fr, const double *norm, double *fn_l,
                          double *fn_r)
{
  for (int k = 0; k < nf; k++)
  {
    fn_l[k] = 0.;
    fn_r[k] = 0.;
    for (int l = 0; l < nd; l++)
    {
      fn_l[k] += fl[k + nf * l] * norm[l];
      fn_r[k] += fr[k + nf * l] * norm[l];
    }
  }
}

/* src/inters.cpp:277-324 */
void orc_rusanov_flux(int nd, double gamma, const double *ul, const double *ur, const double *fl, const double *fr,
                      const double *norm, double *fn)
{
  const int nf = nd + 2;
  double fn_l[MAXF], fn_r[MAXF], v_l[MAXD], v_r[MAXD];
  normal_fluxes(nd, nf, fl, fr, norm, fn_l, fn_r);
  double vn_l = 0, vn_r = 0, vsq_l = 0, vsq_r = 0;
  for (int i = 0; i < nd; i++)
  {
    v_l[i] = ul[i + 1] / ul[0];
    v_r[i] = ur[i + 1] / ur[0];
    vn_l += v_l[i] * norm[i];
    vn_r += v_r[i] * norm[i];
    vsq_l += pow(v_l[i], 2.);
    vsq_r += pow(v_r[i], 2.);
  }
  double p_l = (gamma - 1.0) * (ul[nd + 1] - 0.5 * ul[0] * vsq_l);
  double p_r = (gamma - 1.0) * (ur[nd + 1] - 0.5 * ur[0] * vsq_r);
  double eig = sqrt(gamma * (p_l + p_r) / (ul[0] + ur[0])) + 0.5 * fabs(vn_l + vn_r);
  for (int k = 0; k < nf; k++) fn[k] = 0.5 * ((fn_l[k] + fn_r[k]) - eig * (ur[k] - ul[k]));
}

/* src/inters.cpp:327-437 */
void orc_roeM_flux(int nd, double gamma, const double *ul, const double *ur, const double *fl, const double *fr,
                   const double *norm, double *fn)
{
  const int nf = nd + 2;
  double v_l[MAXD], v_r[MAXD], va[MAXD], dv[MAXD];
  double du[MAXF], bdq[MAXF], fn_l[MAXF], fn_r[MAXF];
  double vn_l = 0., vsq_l = 0., vn_r = 0., vsq_r = 0.;
  for (int i = 0; i < nd; i++)
  {
    v_l[i] = ul[i + 1] / ul[0];
    v_r[i] = ur[i + 1] / ur[0];
    vn_l += v_l[i] * norm[i];
    vn_r += v_r[i] * norm[i];
    vsq_l += v_l[i] * v_l[i];
    vsq_r += v_r[i] * v_r[i];
    dv[i] = v_r[i] - v_l[i];
  }
  double p_l = (gamma - 1.0) * (ul[nd + 1] - 0.5 * ul[0] * vsq_l);
  double p_r = (gamma - 1.0) * (ur[nd + 1] - 0.5 * ur[0] * vsq_r);
  double h_l = (ul[nd + 1] + p_l) / ul[0];
  double h_r = (ur[nd + 1] + p_r) / ur[0];
  double drho = ur[0] - ul[0];
  double dp = p_r - p_l;
  double dh = h_r - h_l;
  double dvn = vn_r - vn_l;
  double sq_rho = sqrt(ur[0] / ul[0]);
  double rrho = 1.0 / (1.0 + sq_rho);
  double ratr = sq_rho * rrho;
  double ra = sq_rho * ul[0];
  double ha = h_l * rrho + h_r * ratr;
  double qq = 0., va_n = 0.;
  for (int i = 0; i < nd; i++)
  {
    va[i] = v_l[i] * rrho + v_r[i] * ratr;
    qq += va[i] * va[i];
    va_n += norm[i] * va[i];
  }
  double aa = sqrt((gamma - 1) * (ha - 0.5 * qq));
  double rcp_aa = 1.0 / aa;
  double abs_ma = fabs(va_n * rcp_aa);
  double b1 = fmax(0.0, fmax(va_n + aa, vn_r + aa));
  double b2 = fmin(0.0, fmin(va_n - aa, vn_l - aa));
  double b1b2 = b1 * b2;
  double rcp_b1_b2 = 1.0 / (b1 - b2);
  b1 = b1 * rcp_b1_b2;
  b2 = b2 * rcp_b1_b2;
  b1b2 = b1b2 * rcp_b1_b2;
  double h = 1.0 - ((p_l < p_r) ? (p_l / p_r) : (p_r / p_l));
  double f = ((abs_ma != 0) ? pow(abs_ma, h) : 1.);
  double g = f / (1.0 + abs_ma);
  for (int i = 0; i < nf - 1; i++) du[i] = ur[i] - ul[i];
  du[nd + 1] = ur[0] * h_r - ul[0] * h_l;
  bdq[0] = drho - f * dp * rcp_aa * rcp_aa;
  bdq[nd + 1] = bdq[0] * ha + ra * dh;
  for (int i = 0; i < nd; i++) bdq[i + 1] = bdq[0] * va[i] + ra * (dv[i] - norm[i] * dvn);
  normal_fluxes(nd, nf, fl, fr, norm, fn_l, fn_r);
  for (int i = 0; i < nf; i++) fn[i] = (b1 * fn_l[i] - b2 * fn_r[i]) + b1b2 * (du[i] - g * bdq[i]);
}

/* src/inters.cpp:439-532 */
void orc_hllc_flux(int nd, double gamma, const double *ul, const double *ur, const double *fl, const double *fr,
                   const double *norm, double *fn)
{
  const int nf = nd + 2;
  double fn_l[MAXF], fn_r[MAXF], v_l[MAXD], v_r[MAXD];
  double vn_l = 0., vsq_l = 0., vn_r = 0., vsq_r = 0.;
  for (int i = 0; i < nd; i++)
  {
    v_l[i] = ul[i + 1] / ul[0];
    v_r[i] = ur[i + 1] / ur[0];
    vn_l += v_l[i] * norm[i];
    vn_r += v_r[i] * norm[i];
    vsq_l += v_l[i] * v_l[i];
    vsq_r += v_r[i] * v_r[i];
  }
  double p_l = (gamma - 1.0) * (ul[nd + 1] - 0.5 * ul[0] * vsq_l);
  double p_r = (gamma - 1.0) * (ur[nd + 1] - 0.5 * ur[0] * vsq_r);
  double h_l = (ul[nd + 1] + p_l) / ul[0];
  double h_r = (ur[nd + 1] + p_r) / ur[0];
  normal_fluxes(nd, nf, fl, fr, norm, fn_l, fn_r);
  double sq_rho = sqrt(ur[0] / ul[0]);
  double rrho = 1. / (sq_rho + 1.);
  double vn_m = rrho * (vn_l + sq_rho * vn_r);
  double h_m = rrho * (h_l + sq_rho * h_r);
  double a_m = sqrt((gamma - 1.) * (h_m - 0.5 * vn_m * vn_m));
  double S_R = vn_m + a_m;
  double S_L = vn_m - a_m;
  double S_star = (p_r - p_l + ul[0] * vn_l * (S_L - vn_l) - ur[0] * vn_r * (S_R - vn_r)) /
                  (ul[0] * (S_L - vn_l) - ur[0] * (S_R - vn_r));
  if (S_L >= 0)
    for (int k = 0; k < nf; k++) fn[k] = fn_l[k];
  else
  {
    if (S_star >= 0)
    {
      double rcp_star = S_L - S_star;
      fn[0] = S_star * (S_L * ul[0] - fn_l[0]) / rcp_star;
      for (int i = 0; i < nd; i++)
        fn[i + 1] = (S_star * (S_L * ul[i + 1] - fn_l[i + 1]) +
                     S_L * (p_l + ul[0] * (S_L - vn_l) * (S_star - vn_l)) * norm[i]) /
                    rcp_star;
      fn[nd + 1] = (S_star * (S_L * ul[nd + 1] - fn_l[nd + 1]) +
                    S_L * (p_l + ul[0] * (S_L - vn_l) * (S_star - vn_l)) * S_star) /
                   rcp_star;
    }
    else
    {
      if (S_R >= 0)
      {
        double rcp_star = S_R - S_star;
        fn[0] = S_star * (S_R * ur[0] - fn_r[0]) / rcp_star;
        for (int i = 0; i < nd; i++)
          fn[i + 1] = (S_star * (S_R * ur[i + 1] - fn_r[i + 1]) +
                       S_R * (p_r + ur[0] * (S_R - vn_r) * (S_star - vn_r)) * norm[i]) /
                      rcp_star;
        fn[nd + 1] = (S_star * (S_R * ur[nd + 1] - fn_r[nd + 1]) +
                      S_R * (p_r + ur[0] * (S_R - vn_r) * (S_star - vn_r)) * S_star) /
                     rcp_star;
      }
      else
        for (int k = 0; k < nf; k++) fn[k] = fn_r[k];
    }
  }
}

/* the "consistent switch" of src/inters.cpp:568-581 and :620-633.
 * NOTE the reference reads norm(2) even in 2-D when n_x = n_y = 0, which a
 * unit normal can never be, so the third test is only reached in 3-D.       */
static double ldg_switch(int nd, double ldg_beta, const double *norm)
{
  if (ldg_beta != 0.)
  {
    if (norm[0] < 0.)
      ldg_beta = -ldg_beta;
    else if (norm[0] == 0.)
    {
      if ((norm[0] + norm[1]) < 0.)
        ldg_beta = -ldg_beta;
      else if ((norm[0] + norm[1]) == 0)
      {
        if (nd > 2 && (norm[0] + norm[2]) < 0.) ldg_beta = -ldg_beta;
      }
    }
  }
  return ldg_beta;
}

/* src/inters.cpp:561-611 */
void orc_ldg_flux(int flux_spec, int nd, const double *ul, const double *ur, const double *fl, const double *fr,
                  const double *norm, double *fn, double ldg_tau, double ldg_beta)
{
  const int nf = nd + 2;
  double f_c[MAXF * MAXD];
  if (flux_spec == 0)
  {
    ldg_beta = ldg_switch(nd, ldg_beta, norm);
    for (int k = 0; k < nf; k++)
      for (int i = 0; i < nd; i++)
        f_c[k + nf * i] = (0.5 + ldg_beta) * fl[k + nf * i] + (0.5 - ldg_beta) * fr[k + nf * i];
  }
  else
  {
    for (int k = 0; k < nf; k++)
      for (int i = 0; i < nd; i++) f_c[k + nf * i] = fr[k + nf * i];
  }
  for (int k = 0; k < nf; k++)
  {
    fn[k] = 0.;
    for (int l = 0; l < nd; l++) fn[k] += f_c[k + nf * l] * norm[l];
  }
  for (int k = 0; k < nf; k++) fn[k] -= ldg_tau * (ur[k] - ul[k]);
}

/* src/inters.cpp:615-646 */
void orc_ldg_solution(int flux_spec, int nd, const double *ul, const double *ur, double *uc, double ldg_beta,
                      const double *norm)
{
  const int nf = nd + 2;
  if (flux_spec == 0)
  {
    ldg_beta = ldg_switch(nd, ldg_beta, norm);
    for (int k = 0; k < nf; k++) uc[k] = 0.5 * (ul[k] + ur[k]) - ldg_beta * (ul[k] - ur[k]);
  }
  else
    for (int k = 0; k < nf; k++) uc[k] = ur[k];
}

/* ------------------------------------------------------------------------ */
/* element methods                                                           */

/* src/eles.cpp:1360-1411 : disu_fpts = opp_0 * disu_upts(0) */
void orc_extrapolate_solution(orc_eles *e)
{
  if (e->n_eles == 0) return;
  orc_dgemm(e->n_fpts, e->n_fields * e->n_eles, e->n_upts, 1.0, 0.0, e->opp_0, e->disu_upts[0], e->disu_fpts);
}

/* src/eles.cpp:1823-1886 : grad_disu_upts(:,:,:,d) = opp_4[d] * disu_upts(0) */
void orc_calculate_gradient(orc_eles *e)
{
  if (e->n_eles == 0) return;
  const long slab = (long)e->n_upts * e->n_eles * e->n_fields;
  for (int d = 0; d < e->n_dims; d++)
    orc_dgemm(e->n_upts, e->n_fields * e->n_eles, e->n_upts, 1.0, 0.0, e->opp_4[d], e->disu_upts[0],
              e->grad_disu_upts + d * slab);
}

/* src/eles.cpp:1415-1478 : pointwise Euler flux, then
 * tdisf(j,i,k,l) = sum_m JGinv(l,m,j,i) * f(k,m)                             */
void orc_evaluate_invFlux(orc_eles *e, const orc_params *P)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < ne; i++)
    for (int j = 0; j < nu; j++)
    {
      double u[MAXF], f[MAXF * MAXD];
      for (int k = 0; k < nf; k++) u[k] = e->disu_upts[0][j + (long)nu * (i + (long)ne * k)];
      orc_calc_invf(nd, P->gamma, u, f);
      const double *JG = e->JGinv_upts + (long)nd * nd * (j + (long)nu * i);
      for (int k = 0; k < nf; k++)
        for (int l = 0; l < nd; l++)
        {
          double *t = &e->tdisf_upts[j + (long)nu * (i + (long)ne * (k + (long)nf * l))];
          *t = 0.;
          for (int m = 0; m < nd; m++) *t += JG[l + nd * m] * f[k + nf * m];
        }
    }
}

/* physical gradient at one point, the BLAS=NO branch of src/eles.cpp:1975-1979:
 * dgemm(n_dims,n_fields,n_dims, inv_detjac, 0, JGinv^T, g_ref, g_phys)        */
static void grad_to_physical(int nd, int nf, double inv_detjac, const double *JG, double *base, long stride_field,
                             long stride_dim)
{
  double tg[MAXD * MAXF], cg[MAXD * MAXF];
  for (int k = 0; k < nf; k++)
    for (int d = 0; d < nd; d++) tg[d + nd * k] = base[k * stride_field + d * stride_dim];
  for (int k = 0; k < nf; k++)
  {
    for (int d = 0; d < nd; d++) cg[d + nd * k] = 0.;
    for (int l = 0; l < nd; l++)
    {
      double temp = inv_detjac * tg[l + nd * k];
      /* A(i,l) = temp_JGinv(i,l) = JGinv(l,i) */
      for (int d = 0; d < nd; d++) cg[d + nd * k] += temp * JG[l + nd * d];
    }
  }
  for (int k = 0; k < nf; k++)
    for (int d = 0; d < nd; d++) base[k * stride_field + d * stride_dim] = cg[d + nd * k];
}

/* src/eles.cpp:1890-2052 */
void orc_correct_gradient(orc_eles *e)
{
  if (e->n_eles == 0) return;
  const int nu = e->n_upts, nfp = e->n_fpts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
  const long slab_u = (long)nu * ne * nf, slab_f = (long)nfp * ne * nf;
  for (int d = 0; d < nd; d++)
    orc_dgemm(nu, nf * ne, nfp, 1.0, 1.0, e->opp_5[d], e->delta_disu_fpts, e->grad_disu_upts + d * slab_u);
  for (int d = 0; d < nd; d++)
    orc_dgemm(nfp, nf * ne, nu, 1.0, 0.0, e->opp_6, e->grad_disu_upts + d * slab_u, e->grad_disu_fpts + d * slab_f);
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < ne; i++)
  {
    for (int j = 0; j < nu; j++)
    {
      double inv_detjac = 1.0 / e->detjac_upts[j + (long)nu * i];
      grad_to_physical(nd, nf, inv_detjac, e->JGinv_upts + (long)nd * nd * (j + (long)nu * i),
                       e->grad_disu_upts + j + (long)nu * i, (long)nu * ne, slab_u);
    }
    for (int j = 0; j < nfp; j++)
    {
      double inv_detjac = 1.0 / e->detjac_fpts[j + (long)nfp * i];
      grad_to_physical(nd, nf, inv_detjac, e->JGinv_fpts + (long)nd * nd * (j + (long)nfp * i),
                       e->grad_disu_fpts + j + (long)nfp * i, (long)nfp * ne, slab_f);
    }
  }
}

/* src/eles.cpp:2285-2392 (LES off): tdisf(j,i,k,l) += JGinv(l,m,j,i)*f_v(k,m), m ascending */
void orc_evaluate_viscFlux(orc_eles *e, const orc_params *P)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < ne; i++)
    for (int j = 0; j < nu; j++)
    {
      double u[MAXF], g[MAXF * MAXD], f[MAXF * MAXD];
      for (int k = 0; k < nf; k++)
      {
        u[k] = e->disu_upts[0][j + (long)nu * (i + (long)ne * k)];
        for (int m = 0; m < nd; m++) g[k + nf * m] = e->grad_disu_upts[j + (long)nu * (i + (long)ne * (k + (long)nf * m))];
      }
      orc_calc_visf(nd, P, u, g, f);
      const double *JG = e->JGinv_upts + (long)nd * nd * (j + (long)nu * i);
      if (e->sgs_model >= 0 && e->sgsf_upts)
      {
        /* src/eles.cpp:2322-2348 */
        double sg[MAXF * MAXD];
        orc_calc_sgsf_upts(e, P, u, g, e->detjac_upts[j + (long)nu * i], i, j, sg);
        for (int q = 0; q < nf * nd; q++) f[q] += 1.0 * sg[q]; /* daxpy */
        for (int k = 0; k < nf; k++)
          for (int l = 0; l < nd; l++)
          {
            double t = 0.0;
            for (int m = 0; m < nd; m++) t += JG[l + nd * m] * sg[k + nf * m];
            e->sgsf_upts[j + (long)nu * (i + (long)ne * (k + (long)nf * l))] = t;
          }
      }
      for (int k = 0; k < nf; k++)
        for (int l = 0; l < nd; l++)
        {
          double *t = &e->tdisf_upts[j + (long)nu * (i + (long)ne * (k + (long)nf * l))];
          for (int m = 0; m < nd; m++) *t += JG[l + nd * m] * f[k + nf * m];
        }
    }
}

/* eles::calc_sgsf_upts, src/eles.cpp:2395-2650: sgs_model 0 (Smagorinsky, wall-damped) and 1 (WALE); f(k,m) = f[k + nf*m] */
void orc_calc_sgsf_upts(const orc_eles *e, const orc_params *P, const double *temp_u, const double *temp_grad_u, double detjac,
                        int ele, int upt, double *temp_sgsf)
{
  const int nd = e->n_dims, nf = e->n_fields, nu = e->n_upts, ne = e->n_eles;
  double u[MAXD], drho[MAXD], dene[MAXD], dke[MAXD], de[MAXD], dmom[MAXD][MAXD], du[MAXD][MAXD], S[MAXD][MAXD];
  double y = 0.0, mu_t = 0.0, diag;
  const double rho = temp_u[0];
  double ke = 0.;
  for (int i = 0; i < nd; i++)
  {
    u[i] = temp_u[i + 1] / rho;
    ke += 0.5 * pow(u[i], 2);
  }
  const double inte = temp_u[nf - 1] / rho - ke;
  if (e->sgs_model == 0)
  {
    for (int i = 0; i < nd; i++)
    {
      const double w = e->wall_distance[upt + (long)nu * (ele + (long)ne * i)];
      y += w * w;
    }
    y = sqrt(y);
  }
  for (int q = 0; q < nf * nd; q++) temp_sgsf[q] = 0.0;
  const double vol = detjac * e->les_vol_factor; /* calc_ele_vol of the class (src/eles_hexas.cpp:1542, eles_pris.cpp:1525, eles_tets.cpp:1589) */
  const double delta = e->filter_ratio * pow(vol, 1. / nd) / (e->order + 1.);
  for (int i = 0; i < nd; i++)
  {
    drho[i] = temp_grad_u[0 + nf * i];
    dene[i] = temp_grad_u[(nf - 1) + nf * i];
    for (int j = 1; j < nf - 1; j++) dmom[i][j - 1] = temp_grad_u[j + nf * i];
  }
  for (int i = 0; i < nd; i++)
  {
    dke[i] = ke * drho[i];
    for (int j = 0; j < nd; j++)
    {
      du[i][j] = (dmom[i][j] - u[j] * drho[i]) / rho;
      dke[i] += rho * u[j] * du[i][j];
    }
    de[i] = (dene[i] - dke[i] - drho[i] * inte) / rho;
  }
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < nd; j++) S[i][j] = (du[i][j] + du[j][i]) / 2.0;
  /* 0 Smagorinsky, 1 WALE, 2 WALE + similarity, 3 SVV (no SGS flux), 4 similarity (src/eles.cpp:2436-2465) */
  const int eddy = e->sgs_model <= 2, sim = e->sgs_model == 2 || e->sgs_model == 4;
  if (!eddy)
    ;
  else if (e->sgs_model == 0)
  {
    double Smod = 0.0;
    for (int i = 0; i < nd; i++)
      for (int j = 0; j < nd; j++) Smod += 2.0 * S[i][j] * S[i][j];
    Smod = sqrt(Smod);
    mu_t = rho * fmin(y * y * e->Kappa * e->Kappa, e->C_s * e->C_s * delta * delta) * Smod;
  }
  else
  {
    double num = 0.0, denom = 0.0;
    const double eps = 1.e-12;
    double Sq[MAXD][MAXD], gT[MAXD][MAXD]; /* gT = du*du (column-major dgemm of the (i,j)-stored arrays), g = gT^T */
    for (int i = 0; i < nd; i++)
      for (int j = 0; j < nd; j++)
      {
        /* hf_array du(i,j) is column-major: dgemm(A=du,B=du) gives C(i,j) = sum_l du(i,l) du(l,j) */
        double s = 0.;
        for (int l = 0; l < nd; l++) s += du[i][l] * du[l][j];
        gT[i][j] = s;
      }
    for (int i = 0; i < nd; i++)
      for (int j = 0; j < nd; j++)
      {
        Sq[i][j] = 0.;
        Sq[i][j] += 0.5 * gT[j][i]; /* g_bar = transpose */
        Sq[i][j] += 0.5 * gT[i][j];
      }
    diag = 0.0;
    for (int i = 0; i < nd; i++) diag += gT[i][i] / 3.0;
    for (int i = 0; i < nd; i++) Sq[i][i] -= diag;
    for (int i = 0; i < nd; i++)
      for (int j = 0; j < nd; j++)
      {
        num += Sq[i][j] * Sq[i][j];
        denom += S[i][j] * S[i][j];
      }
    denom = pow(denom, 2.5) + pow(num, 1.25);
    num = pow(num, 1.5);
    mu_t = rho * e->C_s * e->C_s * delta * delta * num / (denom + eps);
  }
  if (eddy)
  {
    diag = 0.;
    for (int i = 0; i < nd; i++) diag += S[i][i] / 3.0;
    for (int i = 0; i < nd; i++) S[i][i] -= diag;
    for (int j = 0; j < nd; j++)
    {
      temp_sgsf[0 + nf * j] = 0.0;
      temp_sgsf[(nf - 1) + nf * j] = -1.0 * P->gamma * mu_t / e->prandtl_t * de[j];
      for (int k = 0; k < nd; k++) temp_sgsf[(nf - 1) + nf * j] -= u[k] * 2.0 * mu_t * S[k][j];
      for (int i = 1; i < nf - 1; i++) temp_sgsf[i + nf * j] = -2.0 * mu_t * S[i - 1][j];
    }
  }
  if (sim)
  {
    /* src/eles.cpp:2602-2634: the Leonard terms of this step's calc_sgs_terms; note that the off-diagonal momentum
     * entries are filled from their transposes AFTER those received their own term (and the eddy part) */
    const long pl = (long)nu * ne, o = upt + (long)nu * ele;
#define SG(i, j) temp_sgsf[(i) + nf * (j)]
    for (int j = 0; j < nd; j++)
    {
      SG(0, j) += 0.0;
      SG(nf - 1, j) += P->gamma * rho * e->Le[o + j * pl];
    }
    if (nd == 2)
    {
      SG(1, 0) += rho * e->Lu[o + 0 * pl];
      SG(1, 1) += rho * e->Lu[o + 2 * pl];
      SG(2, 0) += SG(1, 1);
      SG(2, 1) += rho * e->Lu[o + 1 * pl];
    }
    else
    {
      SG(1, 0) += rho * e->Lu[o + 0 * pl];
      SG(1, 1) += rho * e->Lu[o + 3 * pl];
      SG(1, 2) += rho * e->Lu[o + 4 * pl];
      SG(2, 0) += SG(1, 1);
      SG(2, 1) += rho * e->Lu[o + 1 * pl];
      SG(2, 2) += rho * e->Lu[o + 5 * pl];
      SG(3, 0) += SG(1, 2);
      SG(3, 1) += SG(2, 2);
      SG(3, 2) += rho * e->Lu[o + 2 * pl];
    }
#undef SG
  }
}

/* eles::calc_sgs_terms (src/eles.cpp:2058-2283), called at the first RK stage of a step for the closures 2, 3, 4
 * (src/solver.cpp:55-62): filtered solution; 3 (SVV): it replaces the solution; 2 / 4: products, their filtered values and
 * the Leonard tensors Lu (n_upts,n_eles,3|6), Le (n_upts,n_eles,n_dims).  Returns -1 or the index of a NaN in disuf_upts. */
long orc_calc_sgs_terms(orc_eles *e)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
  const long pl = (long)nu * ne;
  orc_dgemm(nu, nf * ne, nu, 1.0, 0.0, e->filter_upts, e->disu_upts[0], e->disuf_upts);
  for (long q = 0; q < pl * nf; q++)
    if (isnan(e->disuf_upts[q])) return q;
  if (e->sgs_model == 3)
  {
    for (long q = 0; q < pl * nf; q++) e->disu_upts[0][q] = e->disuf_upts[q];
    return -1;
  }
  if (e->sgs_model != 2 && e->sgs_model != 4) return -1;
  const int dim3 = (nd == 2) ? 3 : 6;
  for (int i = 0; i < nu; i++)
    for (int j = 0; j < ne; j++)
    {
      const long o = i + (long)nu * j;
      double ut[MAXF];
      for (int k = 0; k < nf; k++) ut[k] = e->disu_upts[0][o + k * pl];
      const double rsq = ut[0] * ut[0];
      if (nd == 2)
      {
        e->uu[o + 0 * pl] = ut[1] * ut[1] / rsq;
        e->uu[o + 1 * pl] = ut[2] * ut[2] / rsq;
        e->uu[o + 2 * pl] = ut[1] * ut[2] / rsq;
        ut[3] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2]) / ut[0];
        e->ue[o + 0 * pl] = ut[1] * ut[3] / rsq;
        e->ue[o + 1 * pl] = ut[2] * ut[3] / rsq;
      }
      else
      {
        e->uu[o + 0 * pl] = ut[1] * ut[1] / rsq;
        e->uu[o + 1 * pl] = ut[2] * ut[2] / rsq;
        e->uu[o + 2 * pl] = ut[3] * ut[3] / rsq;
        e->uu[o + 3 * pl] = ut[1] * ut[2] / rsq;
        e->uu[o + 4 * pl] = ut[1] * ut[3] / rsq;
        e->uu[o + 5 * pl] = ut[2] * ut[3] / rsq;
        ut[4] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2] + ut[3] * ut[3]) / ut[0];
        e->ue[o + 0 * pl] = ut[1] * ut[4] / rsq;
        e->ue[o + 1 * pl] = ut[2] * ut[4] / rsq;
        e->ue[o + 2 * pl] = ut[3] * ut[4] / rsq;
      }
    }
  orc_dgemm(nu, dim3 * ne, nu, 1.0, 0.0, e->filter_upts, e->uu, e->Lu);
  /* as the reference has it: the second product is filtered over dim3 * n_eles columns too, although ue has n_dims
   * components -- the surplus columns lie outside Le and are not computed here */
  orc_dgemm(nu, nd * ne, nu, 1.0, 0.0, e->filter_upts, e->ue, e->Le);
  for (int i = 0; i < nu; i++)
    for (int j = 0; j < ne; j++)
    {
      const long o = i + (long)nu * j;
      double ut[MAXF], diag;
      for (int k = 0; k < nf; k++) ut[k] = e->disuf_upts[o + k * pl];
      const double rsq = ut[0] * ut[0];
      if (nd == 2)
      {
        e->Lu[o + 0 * pl] -= (ut[1] * ut[1]) / rsq;
        e->Lu[o + 1 * pl] -= (ut[2] * ut[2]) / rsq;
        e->Lu[o + 2 * pl] -= (ut[1] * ut[2]) / rsq;
        diag = (e->Lu[o + 0 * pl] + e->Lu[o + 1 * pl]) / 3.0;
        ut[3] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2]) / ut[0];
        e->Le[o + 0 * pl] = (e->Le[o + 0 * pl] - ut[1] * ut[3]) / rsq;
        e->Le[o + 1 * pl] = (e->Le[o + 1 * pl] - ut[2] * ut[3]) / rsq;
      }
      else
      {
        e->Lu[o + 0 * pl] -= (ut[1] * ut[1]) / rsq;
        e->Lu[o + 1 * pl] -= (ut[2] * ut[2]) / rsq;
        e->Lu[o + 2 * pl] -= (ut[3] * ut[3]) / rsq;
        e->Lu[o + 3 * pl] -= (ut[1] * ut[2]) / rsq;
        e->Lu[o + 4 * pl] -= (ut[1] * ut[3]) / rsq;
        e->Lu[o + 5 * pl] -= (ut[2] * ut[3]) / rsq;
        diag = (e->Lu[o + 0 * pl] + e->Lu[o + 1 * pl] + e->Lu[o + 2 * pl]) / 3.0;
        ut[4] -= 0.5 * (ut[1] * ut[1] + ut[2] * ut[2] + ut[3] * ut[3]) / ut[0];
        e->Le[o + 0 * pl] = (e->Le[o + 0 * pl] - ut[1] * ut[4]) / rsq;
        e->Le[o + 1 * pl] = (e->Le[o + 1 * pl] - ut[2] * ut[4]) / rsq;
        e->Le[o + 2 * pl] = (e->Le[o + 2 * pl] - ut[3] * ut[4]) / rsq;
      }
      for (int k = 0; k < nd; ++k) e->Lu[o + k * pl] -= diag;
    }
  return -1;
}

/* src/eles.cpp:2817-2910 : sgsf_fpts(:,:,:,d) = opp_0 * sgsf_upts(:,:,:,d) */
void orc_extrapolate_sgsFlux(orc_eles *e)
{
  if (e->n_eles == 0) return;
  const long su = (long)e->n_upts * e->n_eles * e->n_fields, sf = (long)e->n_fpts * e->n_eles * e->n_fields;
  for (int d = 0; d < e->n_dims; d++)
    orc_dgemm(e->n_fpts, e->n_fields * e->n_eles, e->n_upts, 1.0, 0.0, e->opp_0, e->sgsf_upts + d * su, e->sgsf_fpts + d * sf);
  /* transform back to the physical domain: f = |J|^-1 * J * F (:2862-2893) */
  const int nd = e->n_dims, nf = e->n_fields, nfp = e->n_fpts, ne = e->n_eles;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < ne; i++)
    for (int j = 0; j < nfp; j++)
    {
      const double inv_detjac = 1.0 / e->detjac_fpts[j + (long)nfp * i];
      double ts[MAXD * MAXF], ps[MAXD * MAXF];
      for (int k = 0; k < nf; k++)
        for (int d = 0; d < nd; d++) ts[d + nd * k] = e->sgsf_fpts[j + (long)nfp * (i + (long)ne * (k + (long)nf * d))];
      orc_dgemm(nd, nf, nd, inv_detjac, 0.0, e->Jacobian_fpts + (long)nd * nd * (j + (long)nfp * i), ts, ps);
      for (int k = 0; k < nf; k++)
        for (int d = 0; d < nd; d++) e->sgsf_fpts[j + (long)nfp * (i + (long)ne * (k + (long)nf * d))] = ps[d + nd * k];
    }
}

/* src/eles.cpp:1549-1620 : norm_tdisf_fpts = sum_d opp_1[d]*tdisf_upts(:,:,:,d) (beta 0 then 1) */
void orc_extrapolate_totalFlux(orc_eles *e)
{
  if (e->n_eles == 0) return;
  const long slab = (long)e->n_upts * e->n_eles * e->n_fields;
  orc_dgemm(e->n_fpts, e->n_fields * e->n_eles, e->n_upts, 1.0, 0.0, e->opp_1[0], e->tdisf_upts, e->norm_tdisf_fpts);
  for (int d = 1; d < e->n_dims; d++)
    orc_dgemm(e->n_fpts, e->n_fields * e->n_eles, e->n_upts, 1.0, 1.0, e->opp_1[d], e->tdisf_upts + d * slab,
              e->norm_tdisf_fpts);
}

/* src/eles.cpp:1651-1725 : div_tconf_upts(0) = sum_d opp_2[d]*tdisf_upts(:,:,:,d) */
void orc_calculate_divergence(orc_eles *e)
{
  if (e->n_eles == 0) return;
  const long slab = (long)e->n_upts * e->n_eles * e->n_fields;
  orc_dgemm(e->n_upts, e->n_fields * e->n_eles, e->n_upts, 1.0, 0.0, e->opp_2[0], e->tdisf_upts, e->div_tconf_upts);
  for (int d = 1; d < e->n_dims; d++)
    orc_dgemm(e->n_upts, e->n_fields * e->n_eles, e->n_upts, 1.0, 1.0, e->opp_2[d], e->tdisf_upts + d * slab,
              e->div_tconf_upts);
}

/* src/eles.cpp:1738-1817 : norm_tconf -= norm_tdisf (daxpy, overwrites norm_tconf);
 * div_tconf += opp_3*norm_tconf ; NaN scan                                   */
long orc_calculate_corrected_divergence(orc_eles *e)
{
  if (e->n_eles == 0) return -1;
  const long nfl = (long)e->n_eles * e->n_fields * e->n_fpts;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (long i = 0; i < nfl; i++) e->norm_tconf_fpts[i] += -1.0 * e->norm_tdisf_fpts[i];
  orc_dgemm(e->n_upts, e->n_fields * e->n_eles, e->n_fpts, 1.0, 1.0, e->opp_3, e->norm_tconf_fpts, e->div_tconf_upts);
  const long nul = (long)e->n_eles * e->n_upts * e->n_fields;
  for (long ct = 0; ct < nul; ct++)
    if (isnan(e->div_tconf_upts[ct])) return ct;
  return -1;
}

/* src/eles.cpp:1080-1265 */
void orc_AdvanceSolution(orc_eles *e, const orc_params *P, int in_step)
{
  if (e->n_eles == 0) return;
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields;
  const long n = (long)nu * ne * nf;
  double *u0 = e->disu_upts[0], *u1 = e->disu_upts[1];
  const double *div = e->div_tconf_upts, *dj = e->detjac_upts, *src = e->src_upts;
  const int adv = P->adv_type;
  if ((adv == 1 || adv == 2) && in_step == 0) memcpy(u1, u0, sizeof(double) * n); /* disu_upts(1) = disu_upts(0) */
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < nf; i++)
    for (int ic = 0; ic < ne; ic++)
      for (int inp = 0; inp < nu; inp++)
      {
        const long q = inp + (long)nu * (ic + (long)ne * i);
        const double s = src ? src[q] : 0.0;
        const double dt = (P->dt_type == 2) ? e->dt_local[ic] : P->dt;
        const double dd = div[q] / dj[inp + (long)nu * ic];
        if (adv == 0)
          u0[q] -= dt * (dd - s);
        else if (adv == 1)
        {
          if (in_step < 3)
            u0[q] -= dt / 3.0 * (dd - s);
          else
          {
            double rhs = -dd + s;
            u0[q] = 3.0 / 4.0 * u0[q] + 1.0 / 4.0 * u1[q] + dt / 4.0 * rhs;
          }
        }
        else if (adv == 2)
        {
          if (in_step < 2 || in_step == 3)
            u0[q] -= dt / 2.0 * (dd - s);
          else if (in_step == 2)
          {
            double rhs = -dd + s;
            u0[q] = 1.0 / 3.0 * u0[q] + 2.0 / 3.0 * u1[q] + dt / 6.0 * rhs;
          }
        }
        else
        {
          double rhs = -dd + s;
          u1[q] = P->RK_a[in_step] * u1[q] + dt * rhs;
          u0[q] += P->RK_b[in_step] * u1[q];
        }
      }
}

/* src/eles.cpp:1267-1356 */
double orc_calc_dt_local(const orc_eles *e, const orc_params *P, int ele, double h_ref, double CFL, int order)
{
  const int nu = e->n_upts, ne = e->n_eles, nd = e->n_dims;
  const double *U = e->disu_upts[0];
  double lam_inv = 0, lam_visc = 0;
  for (int i = 0; i < nu; i++)
  {
    double rho = U[i + (long)nu * (ele + (long)ne * 0)];
    double vsq = 0;
    for (int d = 0; d < nd; d++)
    {
      double v = U[i + (long)nu * (ele + (long)ne * (d + 1))] / rho;
      vsq += v * v;
    }
    double p = (P->gamma - 1.0) * (U[i + (long)nu * (ele + (long)ne * (nd + 1))] - 0.5 * rho * vsq);
    double c = sqrt(P->gamma * p / rho);
    double inte = p / ((P->gamma - 1.0) * rho);
    double rt_ratio = (P->gamma - 1.0) * inte / (P->rt_inf);
    double mu = (P->mu_inf) * pow(rt_ratio, 1.5) * (1. + (P->c_sth)) / (rt_ratio + (P->c_sth));
    mu = mu + P->fix_vis * (P->mu_inf - mu);
    double lam_inv_new = sqrt(vsq) + c;
    double lam_visc_new = fmax(4.0 / 3.0, P->gamma / P->prandtl) * mu / rho;
    if (lam_inv < lam_inv_new) lam_inv = lam_inv_new;
    if (lam_visc < lam_visc_new) lam_visc = lam_visc_new;
  }
  double dt_visc, dt_inv;
  if (P->viscous)
  {
    dt_visc = (CFL * 0.25 * h_ref * h_ref) / (lam_visc)*1.0 / (2.0 * order + 1.0);
    dt_inv = CFL * h_ref / lam_inv * 1.0 / (2.0 * order + 1.0);
  }
  else
  {
    dt_visc = 1e16;
    dt_inv = CFL * h_ref / lam_inv * 1.0 / (2.0 * order + 1.0);
  }
  return fmin(dt_visc, dt_inv);
}

/* eles::calc_disu_ppts (src/eles.cpp:3757-3778) for every element: out(j, ele, k) = sum_l opp_p(j,l) disu_upts(0)(l, ele, k),
 * l ascending as in the reference's dgemm (src/funcs.cpp:49-123); opp_p (n_ppts, n_upts) from eles::set_opp_p (:3600) */
void orc_calc_disu_ppts(const orc_eles *e, int n_ppts, const double *opp_p, double *out)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields;
  for (int k = 0; k < nf; k++)
    for (int i = 0; i < ne; i++)
      for (int j = 0; j < n_ppts; j++)
      {
        double s = 0.0;
        for (int l = 0; l < nu; l++) s += opp_p[j + (long)n_ppts * l] * e->disu_upts[0][l + (long)nu * (i + (long)ne * k)];
        out[j + (long)n_ppts * (i + (long)ne * k)] = s;
      }
}

/* src/eles.cpp:5045-5074 */
double orc_compute_res_upts(const orc_eles *e, int norm_type, int field)
{
  const int nu = e->n_upts, ne = e->n_eles;
  double sum = 0.;
  for (int i = 0; i < ne; i++)
    for (int j = 0; j < nu; j++)
    {
      const long q = j + (long)nu * (i + (long)ne * field);
      double s = e->src_upts ? e->src_upts[q] : 0.0;
      double r = e->div_tconf_upts[q] / e->detjac_upts[j + (long)nu * i] - s;
      if (norm_type == 0)
        sum = fmax(sum, fabs(r));
      else if (norm_type == 1)
        sum += fabs(r);
      else
        sum += r * r;
    }
  return sum;
}

/* ------------------------------------------------------------------------ */
/* face methods                                                              */

/* src/int_inters.cpp:160-249 */
void orc_int_calculate_common_invFlux_lr(const orc_int_inters *F, orc_eles *el, orc_eles *er, const orc_params *P)
{
  /* the left and right side of a face block may belong to different element classes (a mixed mesh: the reference wires
   * raw pointers per (ctype(ic_l), ctype(ic_r)), src/geometry.cpp:637-706, src/int_inters.cpp:67-121) */
  const int nd = el->n_dims, nf = el->n_fields;
  const long pl = (long)el->n_fpts * el->n_eles, pr = (long)er->n_fpts * er->n_eles;
  const int nfi = F->n_fpts_per_inter;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < F->n_inters; i++)
    for (int j = 0; j < nfi; j++)
    {
      const long il = F->L[j + (long)nfi * i], ir = F->R[j + (long)nfi * i];
      double ul[MAXF], ur[MAXF], fl[MAXF * MAXD], fr[MAXF * MAXD], norm[MAXD], fn[MAXF], uc[MAXF];
      for (int k = 0; k < nf; k++)
      {
        ul[k] = el->disu_fpts[il + k * pl];
        ur[k] = er->disu_fpts[ir + k * pr];
      }
      for (int m = 0; m < nd; m++) norm[m] = el->norm_fpts[il + m * pl];
      orc_calc_invf(nd, P->gamma, ul, fl);
      orc_calc_invf(nd, P->gamma, ur, fr);
      if (P->riemann_solve_type == 0)
        orc_rusanov_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      else if (P->riemann_solve_type == 2)
        orc_roeM_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      else
        orc_hllc_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      for (int k = 0; k < nf; k++)
      {
        el->norm_tconf_fpts[il + k * pl] = fn[k] * el->tdA_fpts[il];
        er->norm_tconf_fpts[ir + k * pr] = -fn[k] * er->tdA_fpts[ir];
      }
      if (P->viscous)
      {
        orc_ldg_solution(0, nd, ul, ur, uc, P->ldg_beta, norm);
        for (int k = 0; k < nf; k++)
        {
          el->delta_disu_fpts[il + k * pl] = (uc[k] - ul[k]);
          er->delta_disu_fpts[ir + k * pr] = (uc[k] - ur[k]);
        }
      }
    }
}

void orc_int_calculate_common_invFlux(const orc_int_inters *F, orc_eles *e, const orc_params *P)
{
  orc_int_calculate_common_invFlux_lr(F, e, e, P);
}

/* src/int_inters.cpp:254-343 */
void orc_int_calculate_common_viscFlux_lr(const orc_int_inters *F, orc_eles *el, orc_eles *er, const orc_params *P)
{
  const int nd = el->n_dims, nf = el->n_fields;
  const long pl = (long)el->n_fpts * el->n_eles, pr = (long)er->n_fpts * er->n_eles;
  const int nfi = F->n_fpts_per_inter;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < F->n_inters; i++)
    for (int j = 0; j < nfi; j++)
    {
      const long il = F->L[j + (long)nfi * i], ir = F->R[j + (long)nfi * i];
      double ul[MAXF], ur[MAXF], gl[MAXF * MAXD], gr[MAXF * MAXD], fl[MAXF * MAXD], fr[MAXF * MAXD], norm[MAXD],
          fn[MAXF];
      for (int k = 0; k < nf; k++)
      {
        ul[k] = el->disu_fpts[il + k * pl];
        ur[k] = er->disu_fpts[ir + k * pr];
      }
      for (int k = 0; k < nd; k++)
        for (int l = 0; l < nf; l++)
        {
          gl[l + nf * k] = el->grad_disu_fpts[il + (l + (long)nf * k) * pl];
          gr[l + nf * k] = er->grad_disu_fpts[ir + (l + (long)nf * k) * pr];
        }
      orc_calc_visf(nd, P, ul, gl, fl);
      orc_calc_visf(nd, P, ur, gr, fr);
      if (el->sgs_model >= 0 && el->sgsf_fpts && er->sgsf_fpts) /* src/int_inters.cpp:302-318 */
        for (int k = 0; k < nd; k++)
          for (int l = 0; l < nf; l++)
          {
            fl[l + nf * k] += el->sgsf_fpts[il + (l + (long)nf * k) * pl];
            fr[l + nf * k] += er->sgsf_fpts[ir + (l + (long)nf * k) * pr];
          }
      for (int m = 0; m < nd; m++) norm[m] = el->norm_fpts[il + m * pl];
      orc_ldg_flux(0, nd, ul, ur, fl, fr, norm, fn, P->ldg_tau, P->ldg_beta);
      for (int k = 0; k < nf; k++)
      {
        el->norm_tconf_fpts[il + k * pl] += fn[k] * el->tdA_fpts[il];
        er->norm_tconf_fpts[ir + k * pr] += -fn[k] * er->tdA_fpts[ir];
      }
    }
}

void orc_int_calculate_common_viscFlux(const orc_int_inters *F, orc_eles *e, const orc_params *P)
{
  orc_int_calculate_common_viscFlux_lr(F, e, e, P);
}

/* ------------------------------------------------------------------------ */
/* integral diagnostics                                                      */

/* src/eles.cpp:5485-5627 ; vol_detjac (n_cub,n_eles) */
void orc_CalcIntegralQuantities(const orc_eles *e, const orc_params *P, int nc, const double *opp, const double *wgt,
                                const double *vdj, int nq, const int *ids, double *out)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
  const long slab = (long)nu * ne;
  const double *U = e->disu_upts[0], *G = e->grad_disu_upts;
  for (int i = 0; i < ne; i++)
    for (int j = 0; j < nc; j++)
    {
      const double detjac = vdj[j + (long)nc * i];
      double u[MAXF], g[MAXF * MAXD];
      for (int m = 0; m < nf; m++)
      {
        u[m] = 0.;
        for (int k = 0; k < nu; k++) u[m] += opp[j + (long)nc * k] * U[k + (long)nu * i + m * slab];
      }
      for (int m = 0; m < nf; m++)
        for (int n = 0; n < nd; n++)
        {
          double s = 0.;
          for (int k = 0; k < nu; k++) s += opp[j + (long)nc * k] * G[k + (long)nu * i + (m + (long)nf * n) * slab];
          g[m + nf * n] = s;
        }
#define GD(m, n) g[(m) + nf * (n)]
      const double irho = 1. / u[0];
      double dudx, dudy, dudz = 0, dvdx, dvdy, dvdz = 0, dwdx = 0, dwdy = 0, dwdz = 0;
      dudx = irho * (GD(1, 0) - u[1] * irho * GD(0, 0));
      dudy = irho * (GD(1, 1) - u[1] * irho * GD(0, 1));
      dvdx = irho * (GD(2, 0) - u[2] * irho * GD(0, 0));
      dvdy = irho * (GD(2, 1) - u[2] * irho * GD(0, 1));
      if (nd == 3)
      {
        dudz = irho * (GD(1, 2) - u[1] * irho * GD(0, 2));
        dvdz = irho * (GD(2, 2) - u[2] * irho * GD(0, 2));
        dwdx = irho * (GD(3, 0) - u[3] * irho * GD(0, 0));
        dwdy = irho * (GD(3, 1) - u[3] * irho * GD(0, 1));
        dwdz = irho * (GD(3, 2) - u[3] * irho * GD(0, 2));
      }
#undef GD
      for (int m = 0; m < nq; ++m)
      {
        double diagnostic = 0.0;
        if (ids[m] == 0)
        {
          double tke = 0.0;
          for (int n = 1; n < nd + 1; n++) tke += 0.5 * u[n] * u[n];
          diagnostic = irho * tke;
        }
        else if (ids[m] == 1)
        {
          const double wz = dvdx - dudy;
          diagnostic = wz * wz;
          if (nd == 3)
          {
            const double wx = dwdy - dvdz, wy = dudz - dwdx;
            diagnostic += wx * wx + wy * wy;
          }
          diagnostic *= 0.5 / irho;
        }
        else if (ids[m] == 2)
        {
          double tke = 0.0;
          for (int n = 1; n < nd + 1; n++) tke += 0.5 * u[n] * u[n];
          const double pressure = (P->gamma - 1.0) * (u[nd + 1] - irho * tke);
          diagnostic = (nd == 2) ? pressure * (dudx + dvdy) : pressure * (dudx + dvdy + dwdz);
        }
        else
        {
          double S[3][3] = {{0}};
          S[0][0] = dudx;
          S[0][1] = (dudy + dvdx) / 2.0;
          S[1][0] = S[0][1];
          S[1][1] = dvdy;
          double diag = (S[0][0] + S[1][1]) / 3.0;
          if (nd == 3)
          {
            S[0][2] = (dudz + dwdx) / 2.0;
            S[1][2] = (dvdz + dwdy) / 2.0;
            S[2][0] = S[0][2];
            S[2][1] = S[1][2];
            S[2][2] = dwdz;
            diag += S[2][2] / 3.0;
          }
          if (ids[m] == 4)
            for (int a = 0; a < nd; a++) S[a][a] -= diag;
          for (int a = 0; a < nd; a++)
            for (int b = 0; b < nd; b++) diagnostic += S[a][b] * S[a][b];
        }
        out[m] += diagnostic * wgt[j] * detjac;
      }
    }
}

/* ------------------------------------------------------------------------ */
/* over-integration                                                          */

/* src/eles.cpp:1480-1545 (BLAS=NO branch) */
void orc_evaluate_invFlux_over_int(orc_eles *e, const orc_params *P, int nc, const double *opp, const double *filter,
                                   const double *JGinv_cub)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
  const long slab = (long)nu * ne;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < ne; i++)
  {
    double *ucub = (double *)malloc(sizeof(double) * nc * nf);
    double *tcub = (double *)malloc(sizeof(double) * nc * nf * nd);
    for (int k = 0; k < nf; k++) orc_dgemm(nc, 1, nu, 1.0, 0.0, opp, e->disu_upts[0] + (long)nu * i + k * slab, ucub + (long)nc * k);
    for (int j = 0; j < nc; j++)
    {
      double u[MAXF], f[MAXF * MAXD];
      for (int k = 0; k < nf; k++) u[k] = ucub[j + (long)nc * k];
      orc_calc_invf(nd, P->gamma, u, f);
      for (int k = 0; k < nf; k++)
        for (int l = 0; l < nd; l++)
        {
          double t = 0.;
          for (int m = 0; m < nd; m++) t += JGinv_cub[l + nd * (m + nd * (j + (long)nc * i))] * f[k + nf * m];
          tcub[j + (long)nc * (k + nf * l)] = t;
        }
    }
    for (int j = 0; j < nu; j++)
      for (int k = 0; k < nf; k++)
        for (int l = 0; l < nd; l++)
        {
          double t = 0.;
          for (int m = 0; m < nc; m++) t += filter[j + (long)nu * m] * tcub[m + (long)nc * (k + nf * l)];
          e->tdisf_upts[j + (long)nu * i + (k + (long)nf * l) * slab] = t;
        }
    free(ucub);
    free(tcub);
  }
}

/* ------------------------------------------------------------------------ */
/* shock capturing                                                           */

/* eles::shock_capture (src/eles.cpp:2918-2959) with shock_det_persson (src/eles_hexas.cpp:1007-1059) */
void orc_shock_capture(orc_eles *e, const orc_shock *S)
{
  const int nu = e->n_upts, ne = e->n_eles, nf = e->n_fields, nd = e->n_dims;
  double *u = e->disu_upts[0];
  const int fld = (S->shock_det_field == 0) ? 0 : nd + 1;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int ic = 0; ic < ne; ic++)
  {
    double modal[512];
    /* step 1: modal = inv_vandermonde * u(:, ic, fld) */
    orc_dgemm(nu, 1, nu, 1.0, 0.0, S->inv_vandermonde, u + (long)nu * ic + (long)fld * nu * ne, modal);
    /* step 2 */
    for (int j = 0; j < nu; j++) modal[j] = modal[j] * modal[j];
    /* step 3 */
    double s = 0;
    for (int j = 0; j < nu; j++)
      if (S->high_modes[j]) s += modal[j] * S->norm_basis_persson[j];
    /* step 4: inner_product(norm, norm + n, modal, 0.) */
    double den = 0.;
    for (int j = 0; j < nu; j++) den = den + S->norm_basis_persson[j] * modal[j];
    S->sensor[ic] = s / den;
  }
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < ne; i++)
    if (S->sensor[i] >= S->s0)
    {
      double temp_sol[512 * MAXF], filt_sol[512 * MAXF];
      for (int j = 0; j < nu; j++)
        for (int k = 0; k < nf; k++) temp_sol[j + nu * k] = u[j + (long)nu * i + (long)k * nu * ne];
      orc_dgemm(nu, nf, nu, 1.0, 0.0, S->exp_filter, temp_sol, filt_sol);
      for (int j = 0; j < nu; j++)
        for (int k = 0; k < nf; k++) u[j + (long)nu * i + (long)k * nu * ne] = filt_sol[j + nu * k];
    }
}

/* ------------------------------------------------------------------------ */
/* boundary faces                                                            */

/* src/bdy_inters.cpp:340-1019, equation 0 (Navier-Stokes / Euler), RANS off, wall model off */
void orc_set_boundary_conditions(int sol_spec, const orc_bc *bc, int nd, int viscous, const double *u_l, double *u_r,
                                 const double *norm, double gamma, double R_ref, int ramp_counter)
{
  double rho_l, rho_r = 0., v_l[MAXD], v_r[MAXD] = {0., 0., 0.}, e_l, e_r = 0., p_l, p_r, T_l, T_r, vn_l, v_sq, machn_l;
  const int bc_flag = bc->flag;
  (void)viscous;
  rho_l = u_l[0];
  for (int i = 0; i < nd; i++) v_l[i] = u_l[i + 1] / u_l[0];
  e_l = u_l[nd + 1];
  v_sq = 0.;
  for (int i = 0; i < nd; i++) v_sq += (v_l[i] * v_l[i]);
  p_l = (gamma - 1.0) * (e_l - 0.5 * rho_l * v_sq);
  T_l = p_l / (rho_l * R_ref);

  if (bc_flag == ORC_SUB_IN_SIMP)
  {
    rho_r = bc->rho;
    for (int i = 0; i < nd; i++) v_r[i] = bc->velocity[i];
    v_sq = 0.;
    for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = p_l / (gamma - 1.0) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_SUB_OUT_SIMP)
  {
    vn_l = 0.;
    for (int i = 0; i < nd; i++) vn_l += v_l[i] * norm[i];
    machn_l = fabs(vn_l) / sqrt(gamma * p_l / rho_l);
    if (vn_l < 0)
    {
      for (int i = 0; i < nd; i++) v_r[i] = vn_l * norm[i];
      v_sq = 0.;
      for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
      T_r = bc->T_total - 0.5 * v_sq * (gamma - 1.0) / (R_ref * gamma);
      p_r = bc->p_static * pow((1.0 + 0.5 * (gamma - 1.0) * (v_sq / (gamma * R_ref * T_r))), -gamma / (gamma - 1.0));
      rho_r = p_r / (R_ref * T_r);
      e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
    }
    else if (vn_l >= 0 && machn_l >= 1)
    {
      rho_r = rho_l;
      for (int i = 0; i < nd; i++) v_r[i] = v_l[i];
      e_r = e_l;
    }
    else
    {
      for (int i = 0; i < nd; i++) v_r[i] = v_l[i];
      rho_r = rho_l;
      p_r = bc->p_static;
      v_sq = 0.;
      for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
      e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
    }
  }
  else if (bc_flag == ORC_SUB_IN_CHAR)
  {
    double V_r, c_l, c_r_sq, c_total_sq, R_plus, aa, bb, cc, dd, Mach_sq, alpha, p_total_temp, T_total_temp;
    if (bc->pressure_ramp)
    {
      if (bc->p_ramp_coeff)
      {
        p_total_temp = bc->p_total_old + (bc->p_total - bc->p_total_old) * bc->p_ramp_coeff * ramp_counter;
        if (p_total_temp >= bc->p_total) p_total_temp = bc->p_total;
      }
      else
        p_total_temp = bc->p_total;
      if (bc->T_ramp_coeff > 0)
      {
        T_total_temp = bc->T_total_old + (bc->T_total - bc->T_total_old) * bc->T_ramp_coeff * ramp_counter;
        if (T_total_temp >= bc->T_total) T_total_temp = bc->T_total;
      }
      else if (bc->T_ramp_coeff < 0)
        T_total_temp = T_l * pow(p_total_temp / p_l, (gamma - 1.0) / gamma);
      else
        T_total_temp = bc->T_total;
    }
    else
    {
      p_total_temp = bc->p_total;
      T_total_temp = bc->T_total;
    }
    const double n_free_stream[3] = {bc->nx, bc->ny, bc->nz};
    vn_l = 0.;
    for (int i = 0; i < nd; i++) vn_l += v_l[i] * norm[i];
    c_l = sqrt(gamma * p_l / rho_l);
    R_plus = vn_l + 2.0 * c_l / (gamma - 1.0);
    c_total_sq = gamma * R_ref * T_total_temp;
    alpha = 0.;
    for (int i = 0; i < nd; i++) alpha += norm[i] * n_free_stream[i];
    aa = 1.0 + 0.5 * (gamma - 1.0) * alpha * alpha;
    bb = -(gamma - 1.0) * alpha * R_plus;
    cc = 0.5 * (gamma - 1.0) * R_plus * R_plus - 2.0 * c_total_sq / (gamma - 1.0);
    dd = bb * bb - 4.0 * aa * cc;
    dd = sqrt(dd > 0.0 ? dd : 0.0); /* std::max(dd, 0.0) */
    V_r = (-bb + dd) / (2.0 * aa);
    V_r = V_r > 0.0 ? V_r : 0.0;
    v_sq = V_r * V_r;
    c_r_sq = c_total_sq - 0.5 * (gamma - 1.0) * v_sq;
    Mach_sq = v_sq / (c_r_sq);
    Mach_sq = Mach_sq < 1.0 ? Mach_sq : 1.0;
    v_sq = Mach_sq * c_r_sq;
    V_r = sqrt(v_sq);
    c_r_sq = c_total_sq - 0.5 * (gamma - 1.0) * v_sq;
    for (int i = 0; i < nd; i++) v_r[i] = V_r * n_free_stream[i];
    T_r = c_r_sq / (gamma * R_ref);
    p_r = p_total_temp * pow(T_r / T_total_temp, gamma / (gamma - 1.0));
    rho_r = p_r / (R_ref * T_r);
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_SUB_OUT_CHAR)
  {
    double c_l, c_r, R_plus, s, vn_r;
    vn_l = 0.;
    for (int i = 0; i < nd; i++) vn_l += v_l[i] * norm[i];
    c_l = sqrt(gamma * p_l / rho_l);
    R_plus = vn_l + 2.0 * c_l / (gamma - 1.0);
    s = p_l / pow(rho_l, gamma);
    p_r = bc->p_static;
    rho_r = pow(p_r / s, 1.0 / gamma);
    c_r = sqrt(gamma * p_r / rho_r);
    vn_r = R_plus - 2.0 * c_r / (gamma - 1.0);
    v_sq = 0.;
    for (int i = 0; i < nd; i++)
    {
      v_r[i] = v_l[i] + (vn_r - vn_l) * norm[i];
      v_sq += (v_r[i] * v_r[i]);
    }
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_SUP_IN)
  {
    rho_r = bc->rho;
    for (int i = 0; i < nd; i++) v_r[i] = bc->velocity[i];
    p_r = bc->p_static;
    v_sq = 0.;
    for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_SUP_OUT)
  {
    rho_r = rho_l;
    for (int i = 0; i < nd; i++) v_r[i] = v_l[i];
    e_r = e_l;
  }
  else if (bc_flag == ORC_SLIP_WALL)
  {
    rho_r = rho_l;
    vn_l = 0.;
    for (int i = 0; i < nd; i++) vn_l += v_l[i] * norm[i];
    if (sol_spec == 0)
      for (int i = 0; i < nd; i++) v_r[i] = v_l[i] - 2 * vn_l * norm[i];
    else
      for (int i = 0; i < nd; i++) v_r[i] = v_l[i] - vn_l * norm[i];
    v_sq = 0.;
    for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = p_l / (gamma - 1.0) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_ISOTHERM_WALL)
  {
    T_r = bc->T_static;
    rho_r = rho_l;
    if (sol_spec == 0)
      for (int i = 0; i < nd; i++) v_r[i] = 2 * bc->velocity[i] - v_l[i];
    else
      for (int i = 0; i < nd; i++) v_r[i] = bc->velocity[i];
    v_sq = 0.;
    for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = rho_r * (R_ref / (gamma - 1.0) * T_r) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_ADIABAT_WALL)
  {
    rho_r = rho_l;
    if (sol_spec == 0)
      for (int i = 0; i < nd; i++) v_r[i] = 2 * bc->velocity[i] - v_l[i];
    else
      for (int i = 0; i < nd; i++) v_r[i] = bc->velocity[i];
    v_sq = 0.;
    for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
    e_r = p_l / (gamma - 1.0) + 0.5 * rho_r * v_sq;
  }
  else if (bc_flag == ORC_CHAR)
  {
    double c_star, vn_star, vn_r, r_plus, r_minus, c_l, c_r, one_over_s, mach;
    vn_l = 0.;
    for (int i = 0; i < nd; i++) vn_l += v_l[i] * norm[i];
    vn_r = 0;
    for (int i = 0; i < nd; i++) vn_r += bc->velocity[i] * norm[i];
    c_l = sqrt(gamma * p_l / rho_l);
    c_r = sqrt(gamma * bc->p_static / bc->rho);
    mach = fabs(vn_l) / c_l;
    if (vn_l < 0)
    {
      if (mach >= 1)
      {
        r_minus = vn_r - 2. / (gamma - 1.) * c_r;
        r_plus = vn_r + 2. / (gamma - 1.) * c_r;
      }
      else
      {
        r_plus = vn_l + 2. / (gamma - 1.) * c_l;
        r_minus = vn_r - 2. / (gamma - 1.) * c_r;
      }
      c_star = 0.25 * (gamma - 1.) * (r_plus - r_minus);
      vn_star = 0.5 * (r_plus + r_minus);
      one_over_s = pow(bc->rho, gamma) / bc->p_static;
      rho_r = pow(1. / gamma * (one_over_s * c_star * c_star), 1. / (gamma - 1.));
      for (int i = 0; i < nd; i++) v_r[i] = vn_star * norm[i] + (bc->velocity[i] - vn_r * norm[i]);
      v_sq = 0.;
      for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
      p_r = rho_r / gamma * c_star * c_star;
      e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
    }
    else
    {
      if (mach >= 1)
      {
        r_minus = vn_l - 2. / (gamma - 1.) * c_l;
        r_plus = vn_l + 2. / (gamma - 1.) * c_l;
      }
      else
      {
        r_plus = vn_l + 2. / (gamma - 1.) * c_l;
        r_minus = vn_r - 2. / (gamma - 1.) * c_r;
      }
      c_star = 0.25 * (gamma - 1.) * (r_plus - r_minus);
      vn_star = 0.5 * (r_plus + r_minus);
      one_over_s = pow(rho_l, gamma) / p_l;
      rho_r = pow(1. / gamma * (one_over_s * c_star * c_star), 1. / (gamma - 1.));
      for (int i = 0; i < nd; i++) v_r[i] = vn_star * norm[i] + (v_l[i] - vn_l * norm[i]);
      v_sq = 0.;
      for (int i = 0; i < nd; i++) v_sq += (v_r[i] * v_r[i]);
      p_r = rho_r / gamma * c_star * c_star;
      e_r = (p_r / (gamma - 1.0)) + 0.5 * rho_r * v_sq;
    }
  }
  else if (bc_flag == ORC_SLIP_WALL_DUAL)
  {
    rho_r = rho_l;
    vn_l = 0.;
    for (int i = 0; i < nd; i++) vn_l += v_l[i] * norm[i];
    for (int i = 0; i < nd; i++) v_r[i] = v_l[i] - 2 * vn_l * norm[i];
    e_r = e_l;
  }
  u_r[0] = rho_r;
  for (int i = 0; i < nd; i++) u_r[i + 1] = rho_r * v_r[i];
  u_r[nd + 1] = e_r;
}

/* src/bdy_inters.cpp:1138-1189 ; gradients stored (field, dim) */
void orc_set_boundary_gradients(const orc_bc *bc, int nd, const double *u_r, const double *grad_ul, double *grad_ur,
                                const double *norm)
{
  const int nf = nd + 2, bc_flag = bc->flag;
  if (bc_flag == ORC_CHAR || bc_flag == ORC_SUP_IN || bc_flag == ORC_SUB_IN_SIMP || bc_flag == ORC_SUB_OUT_SIMP)
    for (int q = 0; q < nf * nd; q++) grad_ur[q] = 0.;
  else
    for (int q = 0; q < nf * nd; q++) grad_ur[q] = grad_ul[q];
  if (bc_flag == ORC_ADIABAT_WALL)
  {
    double v_sq = 0., inte, grad_vel[MAXD * MAXD], grad_inte[MAXD];
#define GU(l, k) grad_ur[(l) + nf * (k)]
    for (int i = 0; i < nd; i++) v_sq += (u_r[i + 1] * u_r[i + 1]);
    inte = (u_r[nd + 1] - 0.5 * v_sq / u_r[0]) / u_r[0];
    for (int j = 0; j < nd; j++)
      for (int i = 0; i < nd; i++) grad_vel[i + nd * j] = (GU(i + 1, j) - GU(0, j) * u_r[i + 1] / u_r[0]) / u_r[0];
    if (nd == 2)
    {
      for (int i = 0; i < nd; i++)
        grad_inte[i] = GU(3, i) - (inte * GU(0, i) + 0.5 * v_sq / (u_r[0] * u_r[0]) * GU(0, i) + u_r[1] * grad_vel[0 + nd * i] +
                                   u_r[2] * grad_vel[1 + nd * i]);
      for (int i = 0; i < nd; i++) GU(3, i) -= (grad_inte[0] * norm[0] + grad_inte[1] * norm[1]) * norm[i];
    }
    else
    {
      for (int i = 0; i < nd; i++)
        grad_inte[i] = GU(4, i) - (inte * GU(0, i) + 0.5 * v_sq / (u_r[0] * u_r[0]) * GU(0, i) + u_r[1] * grad_vel[0 + nd * i] +
                                   u_r[2] * grad_vel[1 + nd * i] + u_r[3] * grad_vel[2 + nd * i]);
      for (int i = 0; i < nd; i++)
        GU(4, i) -= (grad_inte[0] * norm[0] + grad_inte[1] * norm[1] + grad_inte[2] * norm[2]) * norm[i];
    }
#undef GU
  }
}

static int orc_is_wall(int f)
{
  return f == ORC_SLIP_WALL || f == ORC_ISOTHERM_WALL || f == ORC_ADIABAT_WALL || f == ORC_SLIP_WALL_DUAL;
}

/* src/bdy_inters.cpp:213-338 */
void orc_bdy_evaluate_boundaryConditions_invFlux(const orc_bdy_inters *F, orc_eles *e, const orc_params *P)
{
  const int nd = e->n_dims, nf = e->n_fields;
  const long plane = (long)e->n_fpts * e->n_eles;
  const int nfi = F->n_fpts_per_inter;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < F->n_inters; i++)
  {
    const orc_bc *bc = &F->bcs[F->boundary_id[i]];
    for (int j = 0; j < nfi; j++)
    {
      const long il = F->L[j + (long)nfi * i];
      double ul[MAXF], ur[MAXF], fl[MAXF * MAXD], fr[MAXF * MAXD], norm[MAXD], fn[MAXF], uc[MAXF];
      for (int m = 0; m < nd; m++) norm[m] = e->norm_fpts[il + m * plane];
      for (int k = 0; k < nf; k++) ul[k] = e->disu_fpts[il + k * plane];
      orc_set_boundary_conditions(0, bc, nd, P->viscous, ul, ur, norm, P->gamma, F->R_ref, F->ramp_counter);
      orc_calc_invf(nd, P->gamma, ul, fl);
      orc_calc_invf(nd, P->gamma, ur, fr);
      if (bc->flag == ORC_SLIP_WALL_DUAL)
      {
        for (int k = 0; k < nf; k++)
        {
          fn[k] = 0.;
          for (int l = 0; l < nd; l++) fn[k] += fl[k + nf * l] * norm[l];
        }
      }
      else if (P->riemann_solve_type == 0)
        orc_rusanov_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      else if (P->riemann_solve_type == 2)
        orc_roeM_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      else
        orc_hllc_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      for (int k = 0; k < nf; k++) e->norm_tconf_fpts[il + k * plane] = fn[k] * e->tdA_fpts[il];
      if (P->viscous)
      {
        if (orc_is_wall(bc->flag))
          orc_set_boundary_conditions(1, bc, nd, P->viscous, ul, ur, norm, P->gamma, F->R_ref, F->ramp_counter);
        orc_ldg_solution(1, nd, ul, ur, uc, P->ldg_beta, norm);
        for (int k = 0; k < nf; k++) e->delta_disu_fpts[il + k * plane] = (uc[k] - ul[k]);
      }
    }
  }
}

/* src/bdy_inters.cpp:1024-1136, faces without wall model */
void orc_bdy_evaluate_boundaryConditions_viscFlux(const orc_bdy_inters *F, orc_eles *e, const orc_params *P)
{
  const int nd = e->n_dims, nf = e->n_fields;
  const long plane = (long)e->n_fpts * e->n_eles;
  const int nfi = F->n_fpts_per_inter;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < F->n_inters; i++)
  {
    const orc_bc *bc = &F->bcs[F->boundary_id[i]];
    if (bc->flag == ORC_SLIP_WALL) continue;
    for (int j = 0; j < nfi; j++)
    {
      const long il = F->L[j + (long)nfi * i];
      double ul[MAXF], ur[MAXF], gl[MAXF * MAXD], gr[MAXF * MAXD], fl[MAXF * MAXD], fr[MAXF * MAXD], norm[MAXD], fn[MAXF];
      for (int k = 0; k < nf; k++) ul[k] = e->disu_fpts[il + k * plane];
      for (int m = 0; m < nd; m++) norm[m] = e->norm_fpts[il + m * plane];
      for (int k = 0; k < nd; k++)
        for (int l = 0; l < nf; l++) gl[l + nf * k] = e->grad_disu_fpts[il + (l + (long)nf * k) * plane];
      orc_set_boundary_conditions(1, bc, nd, P->viscous, ul, ur, norm, P->gamma, F->R_ref, F->ramp_counter);
      orc_set_boundary_gradients(bc, nd, ur, gl, gr, norm);
      orc_calc_visf(nd, P, ur, gr, fr);
      for (int q = 0; q < nf * nd; q++) fl[q] = 0.; /* unused by flux_spec 1 */
      orc_ldg_flux(1, nd, ul, ur, fl, fr, norm, fn, P->ldg_tau, P->ldg_beta);
      for (int k = 0; k < nf; k++) e->norm_tconf_fpts[il + k * plane] += fn[k] * e->tdA_fpts[il];
    }
  }
}

/* ------------------------------------------------------------------------ */
/* partition faces                                                           */

/* src/mpi_inters.cpp:225-229 : counter order inter -> field -> fpt */
void orc_mpi_pack_solution(const orc_mpi_inters *F, const orc_eles *e)
{
  const int nf = e->n_fields, nfi = F->n_fpts_per_inter;
  const long plane = (long)e->n_fpts * e->n_eles;
  long counter = 0;
  for (int i = 0; i < F->n_inters; i++)
    for (int k = 0; k < nf; k++)
      for (int j = 0; j < nfi; j++) F->out_disu[counter++] = e->disu_fpts[F->L[j + (long)nfi * i] + k * plane];
}

/* src/mpi_inters.cpp:284-289 : inter -> dim -> field -> fpt */
void orc_mpi_pack_corrected_gradient(const orc_mpi_inters *F, const orc_eles *e)
{
  const int nf = e->n_fields, nd = e->n_dims, nfi = F->n_fpts_per_inter;
  const long plane = (long)e->n_fpts * e->n_eles;
  long counter = 0;
  for (int i = 0; i < F->n_inters; i++)
    for (int m = 0; m < nd; m++)
      for (int k = 0; k < nf; k++)
        for (int j = 0; j < nfi; j++)
          F->out_grad[counter++] = e->grad_disu_fpts[F->L[j + (long)nfi * i] + (k + (long)nf * m) * plane];
}

/* mpi_inters::send_sgsf_fpts, the packing half (src/mpi_inters.cpp:339-351): out_buffer_sgsf(fpt, field, dim, inter) */
void orc_mpi_pack_sgsf(const orc_mpi_inters *F, const orc_eles *e)
{
  const int nf = e->n_fields, nd = e->n_dims, nfi = F->n_fpts_per_inter;
  const long plane = (long)e->n_fpts * e->n_eles;
  long counter = 0;
  for (int i = 0; i < F->n_inters; i++)
    for (int m = 0; m < nd; m++)
      for (int k = 0; k < nf; k++)
        for (int j = 0; j < nfi; j++)
          F->out_sgsf[counter++] = e->sgsf_fpts[F->L[j + (long)nfi * i] + (k + (long)nf * m) * plane];
}

/* src/mpi_inters.cpp:400-483 */
void orc_mpi_calculate_common_invFlux(const orc_mpi_inters *F, orc_eles *e, const orc_params *P)
{
  const int nd = e->n_dims, nf = e->n_fields, nfi = F->n_fpts_per_inter;
  const long plane = (long)e->n_fpts * e->n_eles;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < F->n_inters; i++)
    for (int j = 0; j < nfi; j++)
    {
      const long il = F->L[j + (long)nfi * i];
      const int jr = F->Rlut[j + (long)nfi * i];
      double ul[MAXF], ur[MAXF], fl[MAXF * MAXD], fr[MAXF * MAXD], norm[MAXD], fn[MAXF], uc[MAXF];
      for (int k = 0; k < nf; k++)
      {
        ul[k] = e->disu_fpts[il + k * plane];
        ur[k] = F->in_disu[jr + (long)nfi * (k + (long)nf * i)];
      }
      for (int m = 0; m < nd; m++) norm[m] = e->norm_fpts[il + m * plane];
      orc_calc_invf(nd, P->gamma, ul, fl);
      orc_calc_invf(nd, P->gamma, ur, fr);
      if (P->riemann_solve_type == 0)
        orc_rusanov_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      else if (P->riemann_solve_type == 2)
        orc_roeM_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      else
        orc_hllc_flux(nd, P->gamma, ul, ur, fl, fr, norm, fn);
      for (int k = 0; k < nf; k++) e->norm_tconf_fpts[il + k * plane] = fn[k] * e->tdA_fpts[il];
      if (P->viscous)
      {
        orc_ldg_solution(0, nd, ul, ur, uc, P->ldg_beta, norm);
        for (int k = 0; k < nf; k++) e->delta_disu_fpts[il + k * plane] = (uc[k] - ul[k]);
      }
    }
}

/* src/mpi_inters.cpp:485-576 (LES off) */
void orc_mpi_calculate_common_viscFlux(const orc_mpi_inters *F, orc_eles *e, const orc_params *P)
{
  const int nd = e->n_dims, nf = e->n_fields, nfi = F->n_fpts_per_inter;
  const long plane = (long)e->n_fpts * e->n_eles;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < F->n_inters; i++)
    for (int j = 0; j < nfi; j++)
    {
      const long il = F->L[j + (long)nfi * i];
      const int jr = F->Rlut[j + (long)nfi * i];
      double ul[MAXF], ur[MAXF], gl[MAXF * MAXD], gr[MAXF * MAXD], fl[MAXF * MAXD], fr[MAXF * MAXD], norm[MAXD], fn[MAXF];
      for (int k = 0; k < nf; k++)
      {
        ul[k] = e->disu_fpts[il + k * plane];
        ur[k] = F->in_disu[jr + (long)nfi * (k + (long)nf * i)];
      }
      for (int m = 0; m < nd; m++) norm[m] = e->norm_fpts[il + m * plane];
      for (int k = 0; k < nd; k++)
        for (int l = 0; l < nf; l++)
        {
          gl[l + nf * k] = e->grad_disu_fpts[il + (l + (long)nf * k) * plane];
          gr[l + nf * k] = F->in_grad[jr + (long)nfi * (l + (long)nf * (k + (long)nd * i))];
        }
      orc_calc_visf(nd, P, ul, gl, fl);
      orc_calc_visf(nd, P, ur, gr, fr);
      if (e->sgs_model >= 0 && e->sgsf_fpts && F->in_sgsf) /* src/mpi_inters.cpp:536-551: physical SGS flux of both sides */
        for (int k = 0; k < nd; k++)
          for (int l = 0; l < nf; l++)
          {
            fl[l + nf * k] += e->sgsf_fpts[il + (l + (long)nf * k) * plane];
            fr[l + nf * k] += F->in_sgsf[jr + (long)nfi * (l + (long)nf * (k + (long)nd * i))];
          }
      orc_ldg_flux(0, nd, ul, ur, fl, fr, norm, fn, P->ldg_tau, P->ldg_beta);
      for (int k = 0; k < nf; k++) e->norm_tconf_fpts[il + k * plane] += fn[k] * e->tdA_fpts[il];
    }
}

/* ------------------------------------------------------------------------ */
/* src/solver.cpp:50-223, single rank, LES / RANS / forcing / over_int off   */
long orc_CalcResidual_bdy(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_bdy_inters *bdy,
                          int n_bdy_blocks, const orc_params *P)
{
  orc_extrapolate_solution(e);
  if (P->viscous) orc_calculate_gradient(e);
  if (e->n_cub > 0) /* src/solver.cpp:82-91 */
    orc_evaluate_invFlux_over_int(e, P, e->n_cub, e->opp_over_int_cubpts, e->over_int_filter, e->JGinv_over_int_cubpts);
  else
    orc_evaluate_invFlux(e, P);
  for (int b = 0; b < n_face_blocks; b++) orc_int_calculate_common_invFlux(&faces[b], e, P);
  for (int b = 0; b < n_bdy_blocks; b++) orc_bdy_evaluate_boundaryConditions_invFlux(&bdy[b], e, P);
  if (P->viscous)
  {
    orc_correct_gradient(e);
    orc_evaluate_viscFlux(e, P);
    if (e->sgs_model >= 0 && e->sgsf_upts) orc_extrapolate_sgsFlux(e); /* src/solver.cpp:162-167 */
  }
  orc_extrapolate_totalFlux(e);
  orc_calculate_divergence(e);
  if (P->viscous)
  {
    for (int b = 0; b < n_face_blocks; b++) orc_int_calculate_common_viscFlux(&faces[b], e, P);
    for (int b = 0; b < n_bdy_blocks; b++) orc_bdy_evaluate_boundaryConditions_viscFlux(&bdy[b], e, P);
  }
  return orc_calculate_corrected_divergence(e);
}

long orc_CalcResidual(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_params *P)
{
  return orc_CalcResidual_bdy(e, faces, n_face_blocks, 0, 0, P);
}

/* one time step = the RK-stage loop of src/HiFiLES.cpp:201-217 */
long orc_rk_step_bdy(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_bdy_inters *bdy,
                     int n_bdy_blocks, const orc_params *P)
{
  int RKSteps = 1;
  if (P->adv_type == 1 || P->adv_type == 2)
    RKSteps = 4;
  else if (P->adv_type == 3)
    RKSteps = 5;
  else if (P->adv_type == 4)
    RKSteps = 14;
  for (int s = 0; s < RKSteps; s++)
  {
    long bad = -1;
    if (s == 0 && e->sgs_model >= 2) /* src/solver.cpp:55-62 */
      bad = orc_calc_sgs_terms(e);
    if (bad >= 0) return bad;
    bad = orc_CalcResidual_bdy(e, faces, n_face_blocks, bdy, n_bdy_blocks, P);
    if (bad >= 0) return bad;
    orc_AdvanceSolution(e, P, s);
  }
  return -1;
}

long orc_rk_step(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_params *P)
{
  return orc_rk_step_bdy(e, faces, n_face_blocks, 0, 0, P);
}
