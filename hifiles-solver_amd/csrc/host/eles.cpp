// eles.cpp -- host-side mirror of eles / eles_hexas / eles_quads (see eles.hpp).
//
// Setup follows the definitions of the reference:
//   operators   opp_0(j,i) = l_i(fpt_j); opp_1[d](k,j) = l_j(fpt_k)*tnorm(d,k);
//               opp_2[d] = opp_4[d] (k,j) = d_d l_j(upt_k); opp_3 = div of the VCJH correction;
//               opp_5[d](k,j) = opp_3(k,j)*tnorm(d,j); opp_6 = opp_0
//               (/root/reference/src/eles.cpp:3074-3596)
//   metrics     JGinv = det(J) J^-1 by cofactors, tdA, norm (/root/reference/src/eles.cpp:4035-4393)
//   TGV state   ic_form 7 (/root/reference/src/eles.cpp:348-371)
#include "eles.hpp"

#include <cmath>

#include "basis.hpp"

eles::eles() {}

eles::~eles() { free_device(); }

void eles::free_device()
{
  if (dev) hfx_eles_destroy(dev);
  dev = nullptr;
}

void eles::fail(const std::string &msg)
{
  if (err.empty()) err = msg;
}

int eles::setup(int in_n_eles, int in_max_n_spts_per_ele, input *in_run_input)
{
  run_input = in_run_input;
  n_eles = in_n_eles;
  max_n_spts_per_ele = in_max_n_spts_per_ele;
  order = run_input->order;
  viscous = run_input->viscous;
  if (setup_ele_type_specific()) return 1;
  // src/eles_hexas.cpp:74-92, src/eles_quads.cpp:77-102
  if (run_input->shock_cap && set_shock_capture_operators()) return 1;
  if (run_input->over_int && set_over_int()) return 1;
  if (run_input->LES && run_input->SGS_model >= 2 && compute_filter_upts()) return 1; // LES_filter, src/eles.cpp:74-77
  if (run_input->p_res >= 2) /* src/eles_hexas.cpp:97-100 */
  {
    set_loc_ppts();
    set_opp_p();
  }
  shape.setup(n_dims, max_n_spts_per_ele, n_eles);
  n_spts_per_ele.setup(n_eles);
  n_spts_per_ele.initialize_to_value(max_n_spts_per_ele);
  // two time levels for every scheme but forward Euler, zero-initialised (src/eles.cpp:88-110)
  disu_upts.setup(2);
  div_tconf_upts.setup(1);
  for (int l = 0; l < 2; l++) disu_upts(l).setup(n_upts_per_ele, n_eles, n_fields);
  div_tconf_upts(0).setup(n_upts_per_ele, n_eles, n_fields);
  h_ref.setup(n_eles);
  dt_local.setup(n_eles);
  return 0;
}

void eles::set_shape_node(int in_spt, int in_ele, const hf_array<double> &in_pos)
{
  for (int d = 0; d < n_dims; d++) shape(d, in_spt, in_ele) = in_pos(d);
}

int eles::get_fpt_offset(int in_ele, int in_inter, int in_fpt) const
{
  int fpt = in_fpt;
  for (int i = 0; i < in_inter; i++) fpt += n_fpts_per_inter(i); // src/eles.cpp:4655-4660
  return fpt + n_fpts_per_ele * in_ele;
}

// ---- operators --------------------------------------------------------------------
void eles::set_opp_0()
{
  hf_array<double> loc(n_dims);
  opp_0.setup(n_fpts_per_ele, n_upts_per_ele);
  for (int i = 0; i < n_upts_per_ele; i++)
    for (int j = 0; j < n_fpts_per_ele; j++)
    {
      for (int k = 0; k < n_dims; k++) loc(k) = tloc_fpts(k, j);
      opp_0(j, i) = eval_nodal_basis(i, loc);
    }
}

void eles::set_opp_1()
{
  hf_array<double> loc(n_dims);
  opp_1.setup(n_dims);
  for (int d = 0; d < n_dims; d++)
  {
    opp_1(d).setup(n_fpts_per_ele, n_upts_per_ele);
    for (int j = 0; j < n_upts_per_ele; j++)
      for (int k = 0; k < n_fpts_per_ele; k++)
      {
        for (int l = 0; l < n_dims; l++) loc(l) = tloc_fpts(l, k);
        opp_1(d)(k, j) = eval_nodal_basis(j, loc) * tnorm_fpts(d, k);
      }
  }
}

void eles::set_opp_2()
{
  hf_array<double> loc(n_dims);
  opp_2.setup(n_dims);
  for (int d = 0; d < n_dims; d++)
  {
    opp_2(d).setup(n_upts_per_ele, n_upts_per_ele);
    for (int j = 0; j < n_upts_per_ele; j++)
      for (int k = 0; k < n_upts_per_ele; k++)
      {
        for (int l = 0; l < n_dims; l++) loc(l) = loc_upts(l, k);
        opp_2(d)(k, j) = eval_d_nodal_basis(j, d, loc);
      }
  }
}

void eles::set_opp_3()
{
  opp_3.setup(n_upts_per_ele, n_fpts_per_ele);
  fill_opp_3(opp_3);
}

void eles::set_opp_4()
{
  hf_array<double> loc(n_dims);
  opp_4.setup(n_dims);
  for (int d = 0; d < n_dims; d++)
  {
    opp_4(d).setup(n_upts_per_ele, n_upts_per_ele);
    for (int j = 0; j < n_upts_per_ele; j++)
      for (int k = 0; k < n_upts_per_ele; k++)
      {
        for (int l = 0; l < n_dims; l++) loc(l) = loc_upts(l, k);
        opp_4(d)(k, j) = eval_d_nodal_basis(j, d, loc);
      }
  }
}

void eles::set_opp_5()
{
  opp_5.setup(n_dims);
  for (int d = 0; d < n_dims; d++)
  {
    opp_5(d).setup(n_upts_per_ele, n_fpts_per_ele);
    for (int j = 0; j < n_fpts_per_ele; j++)
      for (int k = 0; k < n_upts_per_ele; k++) opp_5(d)(k, j) = opp_3(k, j) * tnorm_fpts(d, j);
  }
}

void eles::set_opp_6()
{
  hf_array<double> loc(n_dims);
  opp_6.setup(n_fpts_per_ele, n_upts_per_ele);
  for (int j = 0; j < n_upts_per_ele; j++)
    for (int l = 0; l < n_fpts_per_ele; l++)
    {
      for (int m = 0; m < n_dims; m++) loc(m) = tloc_fpts(m, l);
      opp_6(l, j) = eval_nodal_basis(j, loc);
    }
}

// ---- metrics ------------------------------------------------------------------------
void eles::calc_pos(const hf_array<double> &in_loc, int in_ele, hf_array<double> &out_pos)
{
  for (int i = 0; i < n_dims; i++)
  {
    out_pos(i) = 0.0;
    for (int j = 0; j < n_spts_per_ele(in_ele); j++)
      out_pos(i) += eval_nodal_s_basis(j, in_loc, n_spts_per_ele(in_ele)) * shape(i, j, in_ele);
  }
}

void eles::calc_d_pos(const hf_array<double> &in_loc, int in_ele, hf_array<double> &out_d_pos)
{
  hf_array<double> d_nodal_s_basis(max_n_spts_per_ele, n_dims);
  eval_d_nodal_s_basis(d_nodal_s_basis, in_loc, n_spts_per_ele(in_ele));
  for (int j = 0; j < n_dims; j++)
    for (int k = 0; k < n_dims; k++)
    {
      out_d_pos(j, k) = 0.0;
      for (int i = 0; i < n_spts_per_ele(in_ele); i++) out_d_pos(j, k) += d_nodal_s_basis(i, k) * shape(j, i, in_ele);
    }
}

void eles::calc_wall_distance(const std::vector<double> &loc_noslip_bdy)
{
  const size_t np = loc_noslip_bdy.size() / (size_t)n_dims;
  wall_distance.setup(n_upts_per_ele, n_eles, n_dims);
  for (int i = 0; i < n_eles; i++)
    for (int j = 0; j < n_upts_per_ele; j++)
    {
      double distmin = 1e20, vecmin[3] = {1e20, 1e20, 1e20};
      for (size_t q = 0; q < np; q++)
      {
        double vec[3] = {0, 0, 0}, dist = 0.0;
        for (int n = 0; n < n_dims; n++)
        {
          vec[n] = pos_upts(j, i, n) - loc_noslip_bdy[n + (size_t)n_dims * q];
          dist += vec[n] * vec[n];
        }
        dist = std::sqrt(dist);
        if (dist < distmin) // (the first of equally near points stays, as in the reference)
        {
          for (int n = 0; n < n_dims; n++) vecmin[n] = vec[n];
          distmin = dist;
        }
      }
      for (int n = 0; n < n_dims; n++) wall_distance(j, i, n) = vecmin[n];
    }
}

int eles::set_transforms_pts(int which)
{
  // which 2: the over-integration cubature points (src/eles.cpp:4155-4210): JGinv only
  const bool at_fpts = which == 1, at_cub = which == 2;
  const int npts = at_cub ? loc_over_int_cubpts.get_dim(1) : (at_fpts ? n_fpts_per_ele : n_upts_per_ele);
  hf_array<double> detjac_cub, pos_cub;
  hf_array<double> &detjac = at_cub ? detjac_cub : (at_fpts ? detjac_fpts : detjac_upts);
  hf_array<double> &JGinv = at_cub ? JGinv_over_int_cubpts : (at_fpts ? JGinv_fpts : JGinv_upts);
  hf_array<double> &posa = at_cub ? pos_cub : (at_fpts ? pos_fpts : pos_upts);
  const hf_array<double> &locs = at_cub ? loc_over_int_cubpts : (at_fpts ? tloc_fpts : loc_upts);
  detjac.setup(npts, n_eles);
  JGinv.setup(n_dims, n_dims, npts, n_eles);
  posa.setup(npts, n_eles, n_dims);
  if (at_fpts)
  {
    tdA_fpts.setup(npts, n_eles);
    norm_fpts.setup(npts, n_eles, n_dims);
    if (run_input->LES) Jacobian_fpts.setup(n_dims, n_dims, npts, n_eles);
  }
  hf_array<double> loc(n_dims), pos(n_dims), d_pos(n_dims, n_dims), v(n_dims);

  // the shape-function values depend on the point only, not on the element: tabulate once
  // (the reference re-evaluates them per element; same values, src/eles.cpp:4956-4990)
  const int ns = max_n_spts_per_ele;
  std::vector<double> sb((size_t)npts * ns), dsb((size_t)npts * ns * n_dims);
  {
    hf_array<double> d_nodal_s_basis(ns, n_dims);
    for (int j = 0; j < npts; j++)
    {
      for (int k = 0; k < n_dims; k++) loc(k) = locs(k, j);
      for (int s = 0; s < ns; s++) sb[(size_t)j * ns + s] = eval_nodal_s_basis(s, loc, ns);
      eval_d_nodal_s_basis(d_nodal_s_basis, loc, ns);
      for (int k = 0; k < n_dims; k++)
        for (int s = 0; s < ns; s++) dsb[((size_t)j * n_dims + k) * ns + s] = d_nodal_s_basis(s, k);
    }
  }

  for (int i = 0; i < n_eles; i++)
    for (int j = 0; j < npts; j++)
    {
      for (int d = 0; d < n_dims; d++)
      {
        double p = 0.0;
        for (int s = 0; s < ns; s++) p += sb[(size_t)j * ns + s] * shape(d, s, i);
        posa(j, i, d) = p;
        for (int k = 0; k < n_dims; k++)
        {
          double q = 0.0;
          for (int s = 0; s < ns; s++) q += dsb[((size_t)j * n_dims + k) * ns + s] * shape(d, s, i);
          d_pos(d, k) = q;
        }
      }
      if (at_fpts && run_input->LES)
        for (int d = 0; d < n_dims; d++)
          for (int k = 0; k < n_dims; k++) Jacobian_fpts(d, k, j, i) = d_pos(d, k);
      if (n_dims == 2)
      {
        const double xr = d_pos(0, 0), xs = d_pos(0, 1), yr = d_pos(1, 0), ys = d_pos(1, 1);
        detjac(j, i) = xr * ys - xs * yr;
        if (detjac(j, i) < 0 && !at_cub)
        {
          fail(at_fpts ? "Negative Jacobian at flux points" : "Negative Jacobian at solution points");
          return 1;
        }
        JGinv(0, 0, j, i) = ys;
        JGinv(0, 1, j, i) = -xs;
        JGinv(1, 0, j, i) = -yr;
        JGinv(1, 1, j, i) = xr;
        if (at_fpts)
        {
          v(0) = (tnorm_fpts(0, j) * d_pos(1, 1)) - (tnorm_fpts(1, j) * d_pos(1, 0));
          v(1) = -(tnorm_fpts(0, j) * d_pos(0, 1)) + (tnorm_fpts(1, j) * d_pos(0, 0));
          tdA_fpts(j, i) = std::sqrt(v(0) * v(0) + v(1) * v(1));
          norm_fpts(j, i, 0) = v(0) / tdA_fpts(j, i);
          norm_fpts(j, i, 1) = v(1) / tdA_fpts(j, i);
        }
      }
      else
      {
        const double xr = d_pos(0, 0), xs = d_pos(0, 1), xt = d_pos(0, 2);
        const double yr = d_pos(1, 0), ys = d_pos(1, 1), yt = d_pos(1, 2);
        const double zr = d_pos(2, 0), zs = d_pos(2, 1), zt = d_pos(2, 2);
        detjac(j, i) = xr * (ys * zt - yt * zs) - xs * (yr * zt - yt * zr) + xt * (yr * zs - ys * zr);
        JGinv(0, 0, j, i) = ys * zt - yt * zs;
        JGinv(0, 1, j, i) = xt * zs - xs * zt;
        JGinv(0, 2, j, i) = xs * yt - xt * ys;
        JGinv(1, 0, j, i) = yt * zr - yr * zt;
        JGinv(1, 1, j, i) = xr * zt - xt * zr;
        JGinv(1, 2, j, i) = xt * yr - xr * yt;
        JGinv(2, 0, j, i) = yr * zs - ys * zr;
        JGinv(2, 1, j, i) = xs * zr - xr * zs;
        JGinv(2, 2, j, i) = xr * ys - xs * yr;
        if (at_fpts)
        {
          const double t0 = tnorm_fpts(0, j), t1 = tnorm_fpts(1, j), t2 = tnorm_fpts(2, j);
          v(0) = ((t0 * (ys * zt - yt * zs)) + (t1 * (yt * zr - yr * zt)) + (t2 * (yr * zs - ys * zr)));
          v(1) = ((t0 * (xt * zs - xs * zt)) + (t1 * (xr * zt - xt * zr)) + (t2 * (xs * zr - xr * zs)));
          v(2) = ((t0 * (xs * yt - xt * ys)) + (t1 * (xt * yr - xr * yt)) + (t2 * (xr * ys - xs * yr)));
          tdA_fpts(j, i) = std::sqrt(v(0) * v(0) + v(1) * v(1) + v(2) * v(2));
          norm_fpts(j, i, 0) = v(0) / tdA_fpts(j, i);
          norm_fpts(j, i, 1) = v(1) / tdA_fpts(j, i);
          norm_fpts(j, i, 2) = v(2) / tdA_fpts(j, i);
        }
      }
    }
  return 0;
}

int eles::set_transforms()
{
  if (n_eles == 0) return 0;
  if (set_transforms_pts(0)) return 1;
  if (set_transforms_pts(1)) return 1;
  if (run_input->over_int && set_transforms_pts(2)) return 1;
  return 0;
}

// ---- initial condition ----------------------------------------------------------------
int eles::set_ics(double &time)
{
  const double gamma = run_input->gamma;
  time = 0.;
  hf_array<double> pos(n_dims), ics(n_fields);
  for (int i = 0; i < n_eles; i++)
    for (int j = 0; j < n_upts_per_ele; j++)
    {
      for (int k = 0; k < n_dims; k++) pos(k) = pos_upts(j, i, k);
      if (run_input->ic_form == 0) // isentropic vortex (BASELINE.json configs[0]), src/eles.cpp:261-280 + src/funcs.cpp:1724-1739
      {
        const double pi = 3.141592653589793, ev_eps_ic = 5.0;
        const double x = pos(0) - time, y = pos(1) - time;
        const double f = 1.0 - (x * x + y * y);
        const double rho = std::pow(1.0 - ev_eps_ic * ev_eps_ic * (gamma - 1.0) / (8.0 * gamma * pi * pi) * std::exp(f), 1.0 / (gamma - 1.0));
        const double vx = 1. - ev_eps_ic * y / (2.0 * pi) * std::exp(f / 2.0);
        const double vy = 1. + ev_eps_ic * x / (2.0 * pi) * std::exp(f / 2.0);
        const double vz = 0., p = std::pow(rho, gamma);
        ics(0) = rho;
        ics(1) = rho * vx;
        ics(2) = rho * vy;
        if (n_dims == 2)
          ics(3) = (p / (gamma - 1.0)) + (0.5 * rho * ((vx * vx) + (vy * vy)));
        else
        {
          ics(3) = rho * vz;
          ics(4) = (p / (gamma - 1.0)) + (0.5 * rho * ((vx * vx) + (vy * vy) + (vz * vz)));
        }
      }
      else if (run_input->ic_form == 1) // uniform flow, src/eles.cpp:284-316
      {
        const double rho = run_input->rho_c_ic, vx = run_input->u_c_ic, vy = run_input->v_c_ic, vz = run_input->w_c_ic;
        const double p = run_input->p_c_ic;
        ics(0) = rho;
        ics(1) = rho * vx;
        ics(2) = rho * vy;
        if (n_dims == 2)
          ics(3) = (p / (gamma - 1.0)) + (0.5 * rho * ((vx * vx) + (vy * vy)));
        else
        {
          ics(3) = rho * vz;
          ics(4) = (p / (gamma - 1.0)) + (0.5 * rho * ((vx * vx) + (vy * vy) + (vz * vz)));
        }
      }
      else if (run_input->ic_form == 7) // Taylor-Green vortex, src/eles.cpp:348-371
      {
        const double V_0 = run_input->uvw_c_ic / run_input->uvw_ref;
        if (n_dims == 2)
        {
          const double p = run_input->p_c_ic +
                           run_input->rho_c_ic * std::pow(V_0, 2) / 4.0 * (std::cos(2.0 * pos(0)) + std::cos(2.0 * pos(1)));
          ics(0) = p / (run_input->R_ref * run_input->T_c_ic);
          ics(1) = ics(0) * V_0 * std::sin(pos(0)) * std::cos(pos(1));
          ics(2) = -ics(0) * V_0 * std::cos(pos(0)) * std::sin(pos(1));
          ics(3) = p / (gamma - 1.0) + 0.5 * (ics(1) * ics(1) + ics(2) * ics(2)) / ics(0);
        }
        else
        {
          const double p = run_input->p_c_ic + run_input->rho_c_ic * std::pow(V_0, 2) / 16.0 *
                                                   (std::cos(2.0 * pos(0)) + std::cos(2.0 * pos(1))) *
                                                   (std::cos(2.0 * pos(2)) + 2.0);
          ics(0) = p / (run_input->R_ref * run_input->T_c_ic);
          ics(1) = ics(0) * V_0 * std::sin(pos(0)) * std::cos(pos(1)) * std::cos(pos(2));
          ics(2) = -ics(0) * V_0 * std::cos(pos(0)) * std::sin(pos(1)) * std::cos(pos(2));
          ics(3) = 0.0;
          ics(4) = p / (gamma - 1.0) + 0.5 * (ics(1) * ics(1) + ics(2) * ics(2) + ics(3) * ics(3)) / ics(0);
        }
      }
      else
      {
        fail("ERROR: Invalid form of initial condition ... (this build: ic_form 0, 1 and 7)");
        return 1;
      }
      for (int k = 0; k < n_fields; k++) disu_upts(0)(j, i, k) = ics(k);
    }
  return 0;
}

// ---- device residency -----------------------------------------------------------------
int eles::mv_all_cpu_gpu(hfx_ctx *ctx)
{
  hfx_eles_desc d{};
  d.n_eles = n_eles; d.n_upts = n_upts_per_ele; d.n_fpts = n_fpts_per_ele; d.n_fields = n_fields; d.n_dims = n_dims;
  d.ele_type = ele_type; d.order = order;
  d.opp_0 = opp_0.get_ptr_cpu(); d.opp_3 = opp_3.get_ptr_cpu();
  for (int i = 0; i < n_dims; i++)
  {
    d.opp_1[i] = opp_1(i).get_ptr_cpu();
    d.opp_2[i] = opp_2(i).get_ptr_cpu();
    if (viscous)
    {
      d.opp_4[i] = opp_4(i).get_ptr_cpu();
      d.opp_5[i] = opp_5(i).get_ptr_cpu();
    }
  }
  if (viscous) d.opp_6 = opp_6.get_ptr_cpu();
  d.detjac_upts = detjac_upts.get_ptr_cpu(); d.JGinv_upts = JGinv_upts.get_ptr_cpu();
  d.detjac_fpts = detjac_fpts.get_ptr_cpu(); d.JGinv_fpts = JGinv_fpts.get_ptr_cpu();
  d.tdA_fpts = tdA_fpts.get_ptr_cpu(); d.norm_fpts = norm_fpts.get_ptr_cpu();
  if (hfx_eles_create(ctx, &d, &dev)) { fail(hfx_last_error()); return 1; }
  for (int l = 0; l < 2; l++)
    if (hfx_eles_upload(dev, l == 0 ? HFX_DISU_UPTS0 : HFX_DISU_UPTS1, disu_upts(l).get_ptr_cpu()))
    {
      fail(hfx_last_error());
      return 1;
    }
  if (hfx_eles_set_h_ref(dev, h_ref.get_ptr_cpu())) { fail(hfx_last_error()); return 1; }
  if (run_input->LES)
  {
    hfx_les l{};
    l.sgs_model = run_input->SGS_model;
    l.C_s = run_input->C_s; l.filter_ratio = run_input->filter_ratio; l.Kappa = run_input->Kappa; l.prandtl_t = run_input->prandtl_t;
    if (run_input->SGS_model == 0 && wall_distance.get_dim(0) == 0)
    {
      fail("Smagorinsky closure: no wall distance (the mesh has no no-slip wall on this rank, or its walls lie on other ranks)");
      return 1;
    }
    if (hfx_eles_set_les(dev, &l, run_input->SGS_model == 0 ? wall_distance.get_ptr_cpu() : nullptr, Jacobian_fpts.get_ptr_cpu()))
    {
      fail(hfx_last_error());
      return 1;
    }
    if (run_input->SGS_model >= 2 && hfx_eles_set_les_filter(dev, filter_upts.get_ptr_cpu())) { fail(hfx_last_error()); return 1; }
  }
  if (n_ppts_per_ele > 0 && hfx_eles_set_opp_p(dev, n_ppts_per_ele, opp_p.get_ptr_cpu()))
  {
    fail(hfx_last_error());
    return 1;
  }
  if (run_input->over_int &&
      hfx_eles_set_over_int(dev, loc_over_int_cubpts.get_dim(1), opp_over_int_cubpts.get_ptr_cpu(), over_int_filter.get_ptr_cpu(),
                            JGinv_over_int_cubpts.get_ptr_cpu()))
  {
    fail(hfx_last_error());
    return 1;
  }
  if (run_input->shock_cap &&
      hfx_eles_set_shock_capture(dev, inv_vandermonde.get_ptr_cpu(), exp_filter.get_ptr_cpu(), norm_basis_persson.get_ptr_cpu(),
                                 persson_high_modes.get_ptr_cpu(), run_input->s0, run_input->shock_det_field))
  {
    fail(hfx_last_error());
    return 1;
  }
  return 0;
}

int eles::cp_array_gpu_cpu(int id, hf_array<double> &dst)
{
  if (!dev) { fail("element block is not on the device"); return 1; }
  if (hfx_eles_download(dev, id, dst.get_ptr_cpu())) { fail(hfx_last_error()); return 1; }
  return 0;
}

int eles::cp_disu_upts_gpu_cpu() { return cp_array_gpu_cpu(HFX_DISU_UPTS0, disu_upts(0)); }
int eles::cp_div_tconf_upts_gpu_cpu() { return cp_array_gpu_cpu(HFX_DIV_TCONF_UPTS, div_tconf_upts(0)); }
int eles::cp_grad_disu_upts_gpu_cpu()
{
  grad_disu_upts.setup(n_upts_per_ele, n_eles, n_fields, n_dims);
  return cp_array_gpu_cpu(HFX_GRAD_DISU_UPTS, grad_disu_upts);
}
int eles::cp_disu_upts_cpu_gpu()
{
  if (!dev) { fail("element block is not on the device"); return 1; }
  if (hfx_eles_upload(dev, HFX_DISU_UPTS0, disu_upts(0).get_ptr_cpu())) { fail(hfx_last_error()); return 1; }
  return 0;
}

// ---- per-stage methods: thin calls into libhfx ---------------------------------------
#define HFX_CALL(expr)                 \
  do                                   \
  {                                    \
    if (n_eles != 0 && (expr) != 0) fail(hfx_last_error()); \
  } while (0)

void eles::extrapolate_solution() { HFX_CALL(hfx_eles_extrapolate_solution(dev)); }
void eles::calculate_gradient() { HFX_CALL(hfx_eles_calculate_gradient(dev)); }
void eles::evaluate_invFlux() { HFX_CALL(hfx_eles_evaluate_invFlux(dev)); }
void eles::evaluate_invFlux_over_int() { HFX_CALL(hfx_eles_evaluate_invFlux_over_int(dev)); }
void eles::shock_capture() { HFX_CALL(hfx_eles_shock_capture(dev)); }
void eles::correct_gradient() { HFX_CALL(hfx_eles_correct_gradient(dev)); }
void eles::evaluate_viscFlux() { HFX_CALL(hfx_eles_evaluate_viscFlux(dev)); }
void eles::extrapolate_sgsFlux() { HFX_CALL(hfx_eles_extrapolate_sgsFlux(dev)); }
void eles::calc_sgs_terms() { HFX_CALL(hfx_eles_calc_sgs_terms(dev)); }
void eles::extrapolate_totalFlux() { HFX_CALL(hfx_eles_extrapolate_totalFlux(dev)); }
void eles::calculate_divergence() { HFX_CALL(hfx_eles_calculate_divergence(dev)); }
void eles::calculate_corrected_divergence() { HFX_CALL(hfx_eles_calculate_corrected_divergence(dev)); }
void eles::AdvanceSolution(int in_step, int adv_type) { HFX_CALL(hfx_eles_AdvanceSolution(dev, in_step, adv_type)); }

double eles::compute_res_upts(int in_norm_type, int in_field)
{
  double v = 0.0;
  HFX_CALL(hfx_eles_compute_res_upts(dev, in_norm_type, in_field, &v));
  return v;
}

double eles::calc_dt_local(int in_ele)
{
  // host copy of disu_upts(0) must be current (cp_disu_upts_gpu_cpu)
  const input &R = *run_input;
  double lam_inv = 0, lam_visc = 0;
  for (int i = 0; i < n_upts_per_ele; i++)
  {
    const double rho = disu_upts(0)(i, in_ele, 0);
    double vsq = 0;
    for (int d = 0; d < n_dims; d++)
    {
      const double v = disu_upts(0)(i, in_ele, d + 1) / rho;
      vsq += v * v;
    }
    const double p = (R.gamma - 1.0) * (disu_upts(0)(i, in_ele, n_dims + 1) - 0.5 * rho * vsq);
    const double c = std::sqrt(R.gamma * p / rho);
    const double inte = p / ((R.gamma - 1.0) * rho);
    const double rt_ratio = (R.gamma - 1.0) * inte / (R.rt_inf);
    double mu = (R.mu_inf) * std::pow(rt_ratio, 1.5) * (1. + (R.c_sth)) / (rt_ratio + (R.c_sth));
    mu = mu + R.fix_vis * (R.mu_inf - mu);
    lam_inv = std::max(lam_inv, std::sqrt(vsq) + c);
    lam_visc = std::max(lam_visc, std::max(4.0 / 3.0, R.gamma / R.prandtl) * mu / rho);
  }
  const double dt_inv = R.CFL * h_ref(in_ele) / lam_inv * 1.0 / (2.0 * R.order + 1.0);
  const double dt_visc = viscous ? (R.CFL * 0.25 * h_ref(in_ele) * h_ref(in_ele)) / (lam_visc)*1.0 / (2.0 * R.order + 1.0) : 1e16;
  return std::min(dt_visc, dt_inv);
}

// =========================================================================================
// hexahedra (/root/reference/src/eles_hexas.cpp)
// =========================================================================================
int eles_hexas::setup_ele_type_specific()
{
  ele_type = 4;
  n_dims = 3;
  if (run_input->equation != 0) { fail("Equation not supported"); return 1; }
  n_fields = 5;
  n_inters_per_ele = 6;
  const int N = order + 1;
  n_upts_per_ele = N * N * N;
  upts_type = run_input->upts_type_hexa;
  hf_array<double> w;
  cubature_1d_nodes(upts_type, N, loc_1d_upts, w);
  if (run_input->loc_1d_upts_override.get_dim(0) == N) loc_1d_upts = run_input->loc_1d_upts_override;
  // solution points: upt = k + N*j + N*N*i <-> (x1d[k], x1d[j], x1d[i])  (eles_hexas.cpp:198-218)
  loc_upts.setup(n_dims, n_upts_per_ele);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++)
      for (int k = 0; k < N; k++)
      {
        const int upt = k + N * j + N * N * i;
        loc_upts(0, upt) = loc_1d_upts(k);
        loc_upts(1, upt) = loc_1d_upts(j);
        loc_upts(2, upt) = loc_1d_upts(i);
      }
  n_fpts_per_inter.setup(6);
  for (int f = 0; f < 6; f++) n_fpts_per_inter(f) = N * N;
  n_fpts_per_ele = 6 * N * N;
  // flux points and reference normals per face (eles_hexas.cpp:222-282, :525-580)
  tloc_fpts.setup(n_dims, n_fpts_per_ele);
  tnorm_fpts.setup(n_dims, n_fpts_per_ele);
  tnorm_fpts.initialize_to_zero();
  static const int fdir[6] = {2, 1, 0, 1, 0, 2};
  static const double fsgn[6] = {-1.0, -1.0, 1.0, 1.0, -1.0, 1.0};
  for (int f = 0; f < 6; f++)
    for (int j = 0; j < N; j++)
      for (int k = 0; k < N; k++)
      {
        const int fpt = k + N * j + N * N * f;
        const double a = loc_1d_upts(k), ar = loc_1d_upts(order - k), b = loc_1d_upts(j);
        switch (f)
        {
        case 0: tloc_fpts(0, fpt) = ar; tloc_fpts(1, fpt) = b; tloc_fpts(2, fpt) = -1.0; break;
        case 1: tloc_fpts(0, fpt) = a; tloc_fpts(1, fpt) = -1.0; tloc_fpts(2, fpt) = b; break;
        case 2: tloc_fpts(0, fpt) = 1.0; tloc_fpts(1, fpt) = a; tloc_fpts(2, fpt) = b; break;
        case 3: tloc_fpts(0, fpt) = ar; tloc_fpts(1, fpt) = 1.0; tloc_fpts(2, fpt) = b; break;
        case 4: tloc_fpts(0, fpt) = -1.0; tloc_fpts(1, fpt) = ar; tloc_fpts(2, fpt) = b; break;
        default: tloc_fpts(0, fpt) = a; tloc_fpts(1, fpt) = b; tloc_fpts(2, fpt) = 1.0; break;
        }
        tnorm_fpts(fdir[f], fpt) = fsgn[f];
      }
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

double eles_hexas::eval_nodal_basis(int in_index, const hf_array<double> &loc)
{
  const int N = order + 1;
  const int i = in_index / (N * N), j = (in_index - N * N * i) / N, k = in_index - N * j - N * N * i;
  return eval_lagrange(loc(0), k, loc_1d_upts) * eval_lagrange(loc(1), j, loc_1d_upts) * eval_lagrange(loc(2), i, loc_1d_upts);
}

double eles_hexas::eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &loc)
{
  const int N = order + 1;
  const int i = in_index / (N * N), j = (in_index - N * N * i) / N, k = in_index - N * j - N * N * i;
  if (in_cpnt == 0)
    return eval_d_lagrange(loc(0), k, loc_1d_upts) * eval_lagrange(loc(1), j, loc_1d_upts) * eval_lagrange(loc(2), i, loc_1d_upts);
  if (in_cpnt == 1)
    return eval_lagrange(loc(0), k, loc_1d_upts) * eval_d_lagrange(loc(1), j, loc_1d_upts) * eval_lagrange(loc(2), i, loc_1d_upts);
  return eval_lagrange(loc(0), k, loc_1d_upts) * eval_lagrange(loc(1), j, loc_1d_upts) * eval_d_lagrange(loc(2), i, loc_1d_upts);
}

void eles_hexas::fill_opp_3(hf_array<double> &o3)
{
  // divergence of the VCJH correction basis (eles_hexas.cpp:1444-1495)
  const int scheme = run_input->vcjh_scheme_hexa;
  double eta = run_input->eta_hexa;
  if (scheme != 0 && !compute_eta(scheme, order, eta))
  {
    fail("ERROR: Invalid VCJH scheme ... (this build: vcjh_scheme_hexa 0-4)");
    return;
  }
  const int N = order + 1;
  hf_array<double> loc(n_dims);
  for (int fp = 0; fp < n_fpts_per_ele; fp++)
  {
    const int f = fp / (N * N), j = (fp - N * N * f) / N, k = fp - N * N * f - N * j;
    for (int u = 0; u < n_upts_per_ele; u++)
    {
      for (int d = 0; d < n_dims; d++) loc(d) = loc_upts(d, u);
      double v;
      switch (f)
      {
      case 0: v = -eval_lagrange(loc(0), order - k, loc_1d_upts) * eval_lagrange(loc(1), j, loc_1d_upts) * eval_d_vcjh_1d(loc(2), 0, order, eta); break;
      case 1: v = -eval_lagrange(loc(0), k, loc_1d_upts) * eval_lagrange(loc(2), j, loc_1d_upts) * eval_d_vcjh_1d(loc(1), 0, order, eta); break;
      case 2: v = eval_lagrange(loc(1), k, loc_1d_upts) * eval_lagrange(loc(2), j, loc_1d_upts) * eval_d_vcjh_1d(loc(0), 1, order, eta); break;
      case 3: v = eval_lagrange(loc(0), order - k, loc_1d_upts) * eval_lagrange(loc(2), j, loc_1d_upts) * eval_d_vcjh_1d(loc(1), 1, order, eta); break;
      case 4: v = -eval_lagrange(loc(1), order - k, loc_1d_upts) * eval_lagrange(loc(2), j, loc_1d_upts) * eval_d_vcjh_1d(loc(0), 0, order, eta); break;
      default: v = eval_lagrange(loc(0), k, loc_1d_upts) * eval_lagrange(loc(1), j, loc_1d_upts) * eval_d_vcjh_1d(loc(2), 1, order, eta); break;
      }
      o3(u, fp) = v;
    }
  }
}

static void equispaced(hf_array<double> &x, int n)
{
  x.setup(n);
  for (int i = 0; i < n; i++) x(i) = -1.0 + ((2.0 * i) / (1.0 * (n - 1)));
}

double eles_hexas::eval_nodal_s_basis(int in_index, const hf_array<double> &loc, int in_n_spts)
{
  // tensor Lagrange on equispaced shape nodes (eles_hexas.cpp:1198-1214); 8-node bricks here
  const int n1 = (int)std::lround(std::cbrt((double)in_n_spts));
  hf_array<double> x;
  equispaced(x, n1);
  const int i = in_index / (n1 * n1), j = (in_index - n1 * n1 * i) / n1, k = in_index - n1 * j - n1 * n1 * i;
  return eval_lagrange(loc(0), k, x) * eval_lagrange(loc(1), j, x) * eval_lagrange(loc(2), i, x);
}

void eles_hexas::eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &loc, int in_n_spts)
{
  const int n1 = (int)std::lround(std::cbrt((double)in_n_spts));
  hf_array<double> x;
  equispaced(x, n1);
  for (int m = 0; m < in_n_spts; m++)
  {
    const int i = m / (n1 * n1), j = (m - n1 * n1 * i) / n1, k = m - n1 * j - n1 * n1 * i;
    d(m, 0) = eval_d_lagrange(loc(0), k, x) * eval_lagrange(loc(1), j, x) * eval_lagrange(loc(2), i, x);
    d(m, 1) = eval_lagrange(loc(0), k, x) * eval_d_lagrange(loc(1), j, x) * eval_lagrange(loc(2), i, x);
    d(m, 2) = eval_lagrange(loc(0), k, x) * eval_lagrange(loc(1), j, x) * eval_d_lagrange(loc(2), i, x);
  }
}

// =========================================================================================
// quadrilaterals (/root/reference/src/eles_quads.cpp)
// =========================================================================================
int eles_quads::setup_ele_type_specific()
{
  ele_type = 1;
  n_dims = 2;
  if (run_input->equation != 0) { fail("Equation not supported"); return 1; }
  n_fields = 4;
  n_inters_per_ele = 4;
  const int N = order + 1;
  n_upts_per_ele = N * N;
  upts_type = run_input->upts_type_quad;
  hf_array<double> w;
  cubature_1d_nodes(upts_type, N, loc_1d_upts, w);
  if (run_input->loc_1d_upts_override.get_dim(0) == N) loc_1d_upts = run_input->loc_1d_upts_override;
  loc_upts.setup(n_dims, n_upts_per_ele);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++)
    {
      loc_upts(0, j + N * i) = loc_1d_upts(j);
      loc_upts(1, j + N * i) = loc_1d_upts(i);
    }
  n_fpts_per_inter.setup(4);
  for (int f = 0; f < 4; f++) n_fpts_per_inter(f) = N;
  n_fpts_per_ele = 4 * N;
  // faces 0: eta=-1, 1: xi=+1, 2: eta=+1 (reversed), 3: xi=-1 (reversed)  (eles_quads.cpp:209-248)
  tloc_fpts.setup(n_dims, n_fpts_per_ele);
  tnorm_fpts.setup(n_dims, n_fpts_per_ele);
  tnorm_fpts.initialize_to_zero();
  for (int f = 0; f < 4; f++)
    for (int j = 0; j < N; j++)
    {
      const int fpt = j + N * f;
      switch (f)
      {
      case 0: tloc_fpts(0, fpt) = loc_1d_upts(j); tloc_fpts(1, fpt) = -1.0; tnorm_fpts(1, fpt) = -1.0; break;
      case 1: tloc_fpts(0, fpt) = 1.0; tloc_fpts(1, fpt) = loc_1d_upts(j); tnorm_fpts(0, fpt) = 1.0; break;
      case 2: tloc_fpts(0, fpt) = loc_1d_upts(order - j); tloc_fpts(1, fpt) = 1.0; tnorm_fpts(1, fpt) = 1.0; break;
      default: tloc_fpts(0, fpt) = -1.0; tloc_fpts(1, fpt) = loc_1d_upts(order - j); tnorm_fpts(0, fpt) = -1.0; break;
      }
    }
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

double eles_quads::eval_nodal_basis(int in_index, const hf_array<double> &loc)
{
  const int N = order + 1;
  const int i = in_index / N, j = in_index - N * i;
  return eval_lagrange(loc(0), j, loc_1d_upts) * eval_lagrange(loc(1), i, loc_1d_upts);
}

double eles_quads::eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &loc)
{
  const int N = order + 1;
  const int i = in_index / N, j = in_index - N * i;
  if (in_cpnt == 0) return eval_d_lagrange(loc(0), j, loc_1d_upts) * eval_lagrange(loc(1), i, loc_1d_upts);
  return eval_lagrange(loc(0), j, loc_1d_upts) * eval_d_lagrange(loc(1), i, loc_1d_upts);
}

void eles_quads::fill_opp_3(hf_array<double> &o3)
{
  // eles_quads.cpp:1192-1270
  const int scheme = run_input->vcjh_scheme_quad;
  double eta = run_input->eta_quad;
  if (scheme != 0 && !compute_eta(scheme, order, eta))
  {
    fail("ERROR: Invalid VCJH scheme ... (this build: vcjh_scheme_quad 0-4)");
    return;
  }
  const int N = order + 1;
  hf_array<double> loc(n_dims);
  for (int fp = 0; fp < n_fpts_per_ele; fp++)
  {
    const int f = fp / N, j = fp - N * f;
    for (int u = 0; u < n_upts_per_ele; u++)
    {
      for (int d = 0; d < n_dims; d++) loc(d) = loc_upts(d, u);
      double v;
      switch (f)
      {
      case 0: v = -eval_lagrange(loc(0), j, loc_1d_upts) * eval_d_vcjh_1d(loc(1), 0, order, eta); break;
      case 1: v = eval_lagrange(loc(1), j, loc_1d_upts) * eval_d_vcjh_1d(loc(0), 1, order, eta); break;
      case 2: v = eval_lagrange(loc(0), order - j, loc_1d_upts) * eval_d_vcjh_1d(loc(1), 1, order, eta); break;
      default: v = -eval_lagrange(loc(1), order - j, loc_1d_upts) * eval_d_vcjh_1d(loc(0), 0, order, eta); break;
      }
      o3(u, fp) = v;
    }
  }
}

double eles_quads::eval_nodal_s_basis(int in_index, const hf_array<double> &loc, int in_n_spts)
{
  const int n1 = (int)std::lround(std::sqrt((double)in_n_spts));
  hf_array<double> x;
  equispaced(x, n1);
  const int j = in_index / n1, i = in_index - n1 * j;
  return eval_lagrange(loc(0), i, x) * eval_lagrange(loc(1), j, x);
}

void eles_quads::eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &loc, int in_n_spts)
{
  const int n1 = (int)std::lround(std::sqrt((double)in_n_spts));
  hf_array<double> x;
  equispaced(x, n1);
  for (int m = 0; m < in_n_spts; m++)
  {
    const int j = m / n1, i = m - n1 * j;
    d(m, 0) = eval_d_lagrange(loc(0), i, x) * eval_lagrange(loc(1), j, x);
    d(m, 1) = eval_lagrange(loc(0), i, x) * eval_d_lagrange(loc(1), j, x);
  }
}

// =========================================================================================
// ASCII restart files: the on-disk state format of the reference (output::write_restart_ascii
// src/output.cpp:1753-1818, read_restart_ascii src/solver.cpp:377-434)
// =========================================================================================
static const char *restart_ele_name(int ele_type) { return ele_type == 1 ? "QUADS" : (ele_type == 4 ? "HEXAS" : "?"); }

void eles::write_restart_info_ascii(std::ostream &f)
{
  // src/eles_hexas.cpp:872-890, src/eles_quads.cpp write_restart_info_ascii
  f << restart_ele_name(ele_type) << std::endl;
  f << "Order" << std::endl;
  f << order << std::endl;
  f << (ele_type == 4 ? "Number of solution points per hexahedral element" : "Number of solution points per quadrilateral element")
    << std::endl;
  f << n_upts_per_ele << std::endl;
  f << "Location of solution points in 1D" << std::endl;
  for (int i = 0; i < order + 1; ++i) f << loc_1d_upts(i) << " ";
  f << std::endl;
}

void eles::write_restart_data_ascii(std::ostream &f)
{
  // src/eles.cpp:845-870
  f << "n_eles" << std::endl;
  f << n_eles << std::endl;
  f << "ele2global_ele hf_array" << std::endl;
  for (int i = 0; i < n_eles; i++) f << ele2global_ele(i) << " ";
  f << std::endl;
  f << "data" << std::endl;
  for (int i = 0; i < n_eles; i++)
  {
    f << ele2global_ele(i) << std::endl;
    for (int j = 0; j < n_upts_per_ele; j++)
    {
      for (int k = 0; k < n_fields; k++) f << disu_upts(0)(j, i, k) << " ";
      f << std::endl;
    }
  }
  f << std::endl;
}

int eles::read_restart_info_ascii(std::istream &f)
{
  // src/eles_hexas.cpp:799-827
  std::string str;
  const std::string name = restart_ele_name(ele_type);
  while (1)
  {
    std::getline(f, str);
    if (str == name) break;
    if (f.eof()) return 0;
  }
  std::getline(f, str);
  f >> order_rest;
  std::getline(f, str);
  std::getline(f, str);
  f >> n_upts_per_ele_rest;
  std::getline(f, str);
  std::getline(f, str);
  loc_1d_upts_rest.setup(order_rest + 1);
  for (int i = 0; i < order_rest + 1; ++i) f >> loc_1d_upts_rest(i);
  set_opp_r();
  return 1;
}

// opp_r(j,k) = restart basis k at solution point j (eval_nodal_basis_restart, src/eles_hexas.cpp:1149-1162)
void eles::set_opp_r()
{
  const int n1 = order_rest + 1;
  opp_r.setup(n_upts_per_ele, n_upts_per_ele_rest);
  for (int j = 0; j < n_upts_per_ele; j++)
    for (int idx = 0; idx < n_upts_per_ele_rest; idx++)
    {
      double v;
      if (n_dims == 3)
      {
        const int i = idx / (n1 * n1), jj = (idx - n1 * n1 * i) / n1, k = idx - n1 * jj - n1 * n1 * i;
        v = eval_lagrange(loc_upts(0, j), k, loc_1d_upts_rest) * eval_lagrange(loc_upts(1, j), jj, loc_1d_upts_rest) *
            eval_lagrange(loc_upts(2, j), i, loc_1d_upts_rest);
      }
      else
      {
        const int i = idx / n1, jj = idx - n1 * i;
        v = eval_lagrange(loc_upts(0, j), jj, loc_1d_upts_rest) * eval_lagrange(loc_upts(1, j), i, loc_1d_upts_rest);
      }
      opp_r(j, idx) = v;
    }
}

int eles::read_restart_data_ascii(std::istream &f)
{
  // src/eles.cpp:655-728
  std::string str;
  const std::string name = restart_ele_name(ele_type);
  f.clear();
  f.seekg(0, f.beg);
  while (1)
  {
    std::getline(f, str);
    if (str == name) break;
    if (f.eof()) return 0; // Restart file doesn't contain my elements
  }
  while (1)
  {
    std::getline(f, str);
    if (str == "n_eles") break;
    if (f.eof()) { fail("restart file: no n_eles block"); return 1; }
  }
  int num_eles_to_read = 0;
  f >> num_eles_to_read;
  std::getline(f, str);
  std::getline(f, str);
  std::getline(f, str);
  std::getline(f, str);
  hf_array<double> rest(n_upts_per_ele_rest, n_fields);
  for (int i = 0; i < num_eles_to_read; i++)
  {
    int ele = -1;
    f >> ele;
    int index = -1;
    for (int q = 0; q < n_eles; q++)
      if (ele2global_ele(q) == ele) { index = q; break; }
    if (index != -1)
    {
      for (int j = 0; j < n_upts_per_ele_rest; j++)
        for (int k = 0; k < n_fields; k++) f >> rest(j, k);
      for (int m = 0; m < n_fields; m++)
        for (int j = 0; j < n_upts_per_ele; j++)
        {
          double value = 0.;
          for (int k = 0; k < n_upts_per_ele_rest; k++) value += opp_r(j, k) * rest(k, m);
          disu_upts(0)(j, index, m) = value;
        }
    }
    else
    {
      std::getline(f, str);
      for (int j = 0; j < n_upts_per_ele_rest; j++) std::getline(f, str);
    }
  }
  if (!f) { fail("restart file: truncated data block"); return 1; }
  return 0;
}
