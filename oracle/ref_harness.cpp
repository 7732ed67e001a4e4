// ref_harness.cpp -- TEST INFRASTRUCTURE, container-only.
//
// Drives the GENUINE reference (compiled from /root/reference/src where the
// sources lie, see oracle/Makefile) through the same sequence as
// /root/reference/src/HiFiLES.cpp:113-217 (setup, GeoPreprocess, InitSolution,
// RK loop of CalcResidual + AdvanceSolution) and dumps every array the hot
// path consumes or produces into one binary file, which
// oracle/capture_golden.py converts into the fixtures under tests/golden/.
//
// Nothing here is product code and nothing here travels to the GPU box
// except the built binary under oracle/_ref/ (which cannot run there anyway
// because it needs /root/reference/data).  No reference source is copied:
// the reference headers are #included from /root/reference/include.
//
// usage: HIFILES_HOME=/root/reference ref_harness <input_file> <out.bin> <n_steps> <level>
//   level 0: state after every stage only
//   level 1: + operators, params, face tables, metrics, div_tconf of the first residual
//   level 2: + every intermediate array of the first CalcResidual

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <iostream>
#include <fstream>
#include <sstream>
#include <iomanip>
#include <algorithm>
#include <map>
#include <cstdint>
#include <unistd.h>

// The hot-path arrays are protected members of eles / inters
// (include/eles.h:470-935, include/inters.h:86-126): open them for dumping.
#define protected public
#define private public
#include "global.h"
#include "geometry.h"
#include "solver.h"
#include "output.h"
#include "solution.h"
#include "mesh.h"
#undef protected
#undef private

using namespace std;

static FILE *g_out = NULL;

static void put_header(const char *name, char dtype, const vector<int64_t> &dims)
{
  int32_t nl = (int32_t)strlen(name);
  fwrite(&nl, 4, 1, g_out);
  fwrite(name, 1, nl, g_out);
  fwrite(&dtype, 1, 1, g_out);
  int32_t nd = (int32_t)dims.size();
  fwrite(&nd, 4, 1, g_out);
  fwrite(dims.data(), 8, nd, g_out);
}

static void put_d(const string &name, const double *p, vector<int64_t> dims)
{
  int64_t n = 1;
  for (auto d : dims) n *= d;
  put_header(name.c_str(), 'd', dims);
  fwrite(p, 8, n, g_out);
}

static void put_i(const string &name, const int32_t *p, vector<int64_t> dims)
{
  int64_t n = 1;
  for (auto d : dims) n *= d;
  put_header(name.c_str(), 'i', dims);
  fwrite(p, 4, n, g_out);
}

static void put_arr(const string &name, hf_array<double> &a)
{
  vector<int64_t> dims;
  // trailing unit dimensions are dropped, but keep at least one
  int d[4] = {a.get_dim(0), a.get_dim(1), a.get_dim(2), a.get_dim(3)};
  int nd = 4;
  while (nd > 1 && d[nd - 1] == 1) nd--;
  for (int i = 0; i < nd; i++) dims.push_back(d[i]);
  put_d(name, a.get_ptr_cpu(), dims);
}

static void put_scalar(const string &name, double v) { put_d(name, &v, {1}); }

// pointer table -> index table relative to the owning array
static void put_ptr_table(const string &name, hf_array<double *> &t, double *base, int n0, int n1)
{
  vector<int32_t> idx((size_t)n0 * n1);
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n0; j++)
      idx[j + (size_t)n0 * i] = (int32_t)(t(j, i) - base);
  put_i(name, idx.data(), {n0, n1});
}

int main(int argc, char *argv[])
{
  if (argc < 5)
  {
    fprintf(stderr, "usage: ref_harness <input> <out.bin> <n_steps> <level>\n");
    return 2;
  }
  int n_steps = atoi(argv[3]);
  int level = atoi(argv[4]);

  struct solution FlowSol;
  mesh *mesh_data = new mesh();

  run_input.setup(argv[1], 0);
  SetInput(&FlowSol);
  GeoPreprocess(&FlowSol, *mesh_data);
  delete mesh_data;
  InitSolution(&FlowSol);

  int RKSteps = 1;
  if (run_input.adv_type == 1 || run_input.adv_type == 2) RKSteps = 4;
  else if (run_input.adv_type == 3) RKSteps = 5;
  else if (run_input.adv_type == 4) RKSteps = 14;

  g_out = fopen(argv[2], "wb");
  if (!g_out) { perror("open out"); return 1; }

  // the element classes that carry the mesh.  One class: the arrays keep their plain names (the single-type
  // fixtures).  Several (a MIXED mesh, e.g. prisms on the walls and tetrahedra in the core): class c's arrays are
  // prefixed "c<ele_type>_", and the face tables carry the class of either side of every face.
  vector<eles *> Es;
  vector<int> etypes;
  for (int i = 0; i < FlowSol.n_ele_types; i++)
    if (FlowSol.mesh_eles(i)->get_n_eles() != 0)
    {
      Es.push_back(FlowSol.mesh_eles(i));
      etypes.push_back(i);
    }
  const bool mixed = Es.size() > 1;
  auto pre = [&](size_t c) { return mixed ? "c" + to_string(etypes[c]) + "_" : string(); };
  eles *E = Es[0];
  int etype = etypes[0];
  int n_eles = E->n_eles, n_upts = E->n_upts_per_ele, n_fpts = E->n_fpts_per_ele;
  int n_fields = E->n_fields, n_dims = E->n_dims;
  if (mixed && (run_input.LES || run_input.n_integral_quantities != 0 || run_input.over_int || run_input.shock_cap ||
                run_input.dt_type != 0 || getenv("HFX_DUMP_PPTS") || getenv("HFX_DUMP_RESTART")))
  {
    fprintf(stderr, "harness: mixed meshes are dumped for the plain Navier-Stokes / Euler path only\n");
    return 1;
  }
  for (size_t c = 0; c < Es.size(); c++)
  {
    eles *X = Es[c];
    int32_t sizes[8] = {X->n_eles, X->n_upts_per_ele, X->n_fpts_per_ele, X->n_fields, X->n_dims, X->order, etypes[c], RKSteps};
    put_i(pre(c) + "sizes", sizes, {8});
  }
  if (mixed)
  {
    vector<int32_t> cl(etypes.begin(), etypes.end());
    put_i("classes", cl.data(), {(int64_t)cl.size()});
  }
  // which class owns a pointer into a flux-point array, and the offset inside the class's (fpt, ele) plane x fields
  auto locate = [&](double *ptr, int which, int &cls) -> int64_t {
    for (size_t c = 0; c < Es.size(); c++)
    {
      hf_array<double> &a = (which == 0) ? Es[c]->disu_fpts : (which == 1) ? Es[c]->norm_tconf_fpts : (which == 2) ? Es[c]->tdA_fpts : Es[c]->norm_fpts;
      double *b = a.get_ptr_cpu();
      const int64_t n = (int64_t)a.get_dim(0) * a.get_dim(1) * a.get_dim(2) * a.get_dim(3);
      if (ptr >= b && ptr < b + n) { cls = (int)c; return ptr - b; }
    }
    cls = -1;
    return -1;
  };

  if (level >= 1)
  {
    // frozen scalars the path reads from run_input
    put_scalar("gamma", run_input.gamma);
    put_scalar("prandtl", run_input.prandtl);
    put_scalar("rt_inf", run_input.rt_inf);
    put_scalar("mu_inf", run_input.mu_inf);
    put_scalar("c_sth", run_input.c_sth);
    put_scalar("fix_vis", run_input.fix_vis);
    put_scalar("dt", run_input.dt);
    put_scalar("ldg_beta", run_input.ldg_beta);
    put_scalar("ldg_tau", run_input.ldg_tau);
    put_scalar("viscous", run_input.viscous);
    put_scalar("riemann_solve_type", run_input.riemann_solve_type);
    put_scalar("vis_riemann_solve_type", run_input.vis_riemann_solve_type);
    put_scalar("adv_type", run_input.adv_type);
    put_scalar("dt_type", run_input.dt_type);
    put_scalar("R_ref", run_input.R_ref);
    put_scalar("p_c_ic", run_input.p_c_ic);
    put_scalar("rho_c_ic", run_input.rho_c_ic);
    put_scalar("T_c_ic", run_input.T_c_ic);
    put_scalar("uvw_c_ic", run_input.uvw_c_ic);
    put_scalar("uvw_ref", run_input.uvw_ref);
    put_arr("RK_a", run_input.RK_a);
    put_arr("RK_b", run_input.RK_b);

    for (size_t c = 0; c < Es.size(); c++)
    {
      eles *X = Es[c];
      const string q = pre(c);
      put_arr(q + "opp_0", X->opp_0);
      put_arr(q + "opp_3", X->opp_3);
      for (int d = 0; d < n_dims; d++)
      {
        string s = to_string(d);
        put_arr(q + "opp_1_" + s, X->opp_1(d));
        put_arr(q + "opp_2_" + s, X->opp_2(d));
        if (run_input.viscous)
        {
          put_arr(q + "opp_4_" + s, X->opp_4(d));
          put_arr(q + "opp_5_" + s, X->opp_5(d));
        }
      }
      if (run_input.viscous) put_arr(q + "opp_6", X->opp_6);
      put_arr(q + "loc_upts", X->loc_upts);
      put_arr(q + "tloc_fpts", X->tloc_fpts);
      put_arr(q + "tnorm_fpts", X->tnorm_fpts);
      put_arr(q + "shape", X->shape);
    }

    // interior face tables as offsets into the owning element arrays
    for (int t = 0; t < FlowSol.n_int_inter_types; t++)
    {
      int_inters &I = FlowSol.mesh_int_inters(t);
      if (I.n_inters == 0) continue;
      string s = "int" + to_string(t) + "_";
      int nf = I.n_fpts_per_inter, ni = I.n_inters;
      // field-0 plane; consistency of the other fields / arrays is asserted below
      vector<int32_t> L((size_t)nf * ni), R((size_t)nf * ni), cl(ni), cr(ni);
      for (int i = 0; i < ni; i++)
        for (int j = 0; j < nf; j++)
        {
          int kl, kr, k2;
          const int64_t ol = locate(I.disu_fpts_l(j, i, 0), 0, kl), orr = locate(I.disu_fpts_r(j, i, 0), 0, kr);
          if (kl < 0 || kr < 0) { fprintf(stderr, "harness: face pointer outside every element class\n"); return 1; }
          if (j == 0) { cl[i] = etypes[kl]; cr[i] = etypes[kr]; }
          if (cl[i] != etypes[kl] || cr[i] != etypes[kr]) { fprintf(stderr, "harness: a face changes class between flux points\n"); return 1; }
          L[j + (size_t)nf * i] = (int32_t)ol;
          R[j + (size_t)nf * i] = (int32_t)orr;
          const ptrdiff_t pl = (ptrdiff_t)Es[kl]->n_fpts_per_ele * Es[kl]->n_eles, pr = (ptrdiff_t)Es[kr]->n_fpts_per_ele * Es[kr]->n_eles;
          for (int k = 0; k < n_fields; k++)
          {
            if (locate(I.disu_fpts_l(j, i, k), 0, k2) != ol + k * pl || k2 != kl || locate(I.disu_fpts_r(j, i, k), 0, k2) != orr + k * pr || k2 != kr ||
                locate(I.norm_tconf_fpts_l(j, i, k), 1, k2) != ol + k * pl || k2 != kl ||
                locate(I.norm_tconf_fpts_r(j, i, k), 1, k2) != orr + k * pr || k2 != kr)
            {
              fprintf(stderr, "harness: face table layout assumption violated\n");
              return 1;
            }
          }
          if (locate(I.tdA_fpts_l(j, i), 2, k2) != ol || k2 != kl || locate(I.tdA_fpts_r(j, i), 2, k2) != orr || k2 != kr ||
              locate(I.norm_fpts(j, i, 0), 3, k2) != ol || k2 != kl)
          {
            fprintf(stderr, "harness: face metric table layout assumption violated\n");
            return 1;
          }
        }
      if (mixed)
      {
        put_i(s + "cl", cl.data(), {ni});
        put_i(s + "cr", cr.data(), {ni});
      }
      put_i(s + "L", L.data(), {nf, ni});
      put_i(s + "R", R.data(), {nf, ni});
    }
  }
  // boundary faces: left offsets, the bc_list index of every face, and bc_list itself after
  // input::read_boundary_param's non-dimensionalisation
  {
    // bdy_inters::inlet.nbs is read by evaluate_boundaryConditions_* (src/bdy_inters.cpp:249) but only
    // ever written by add_les_inlet (LES runs): give it the value an LES-off run means
    for (int t = 0; t < FlowSol.n_bdy_inter_types; t++) FlowSol.mesh_bdy_inters(t).inlet.nbs = 0;
    // likewise turbinlet::type, never initialised but read at the first stage of every step when LES is on
    // (src/solver.cpp:110-117): 0 = no synthetic-turbulence inlet
    for (int t = 0; t < FlowSol.n_bdy_inter_types; t++) FlowSol.mesh_bdy_inters(t).inlet.type = 0;
    bool any = false;
    for (int t = 0; t < FlowSol.n_bdy_inter_types; t++)
    {
      bdy_inters &B = FlowSol.mesh_bdy_inters(t);
      if (B.n_inters == 0) continue;
      any = true;
      string s = "bdy" + to_string(t) + "_";
      int nf = B.n_fpts_per_inter, ni = B.n_inters;
      vector<int32_t> L((size_t)nf * ni), id(ni), cl(ni);
      for (int i = 0; i < ni; i++)
      {
        id[i] = B.boundary_id(i);
        for (int j = 0; j < nf; j++)
        {
          int kl, k2;
          const int64_t ol = locate(B.disu_fpts_l(j, i, 0), 0, kl);
          if (kl < 0) { fprintf(stderr, "harness: boundary face pointer outside every element class\n"); return 1; }
          if (j == 0) cl[i] = etypes[kl];
          L[j + (size_t)nf * i] = (int32_t)ol;
          if (cl[i] != etypes[kl] || locate(B.norm_tconf_fpts_l(j, i, 0), 1, k2) != ol || k2 != kl || locate(B.tdA_fpts_l(j, i), 2, k2) != ol || k2 != kl ||
              locate(B.norm_fpts(j, i, 0), 3, k2) != ol || k2 != kl)
          {
            fprintf(stderr, "harness: boundary face table layout assumption violated\n");
            return 1;
          }
        }
      }
      put_i(s + "L", L.data(), {nf, ni});
      put_i(s + "id", id.data(), {ni});
      if (mixed) put_i(s + "cl", cl.data(), {ni});
    }
    if (any)
    {
      const int nb = run_input.bc_list.get_dim(0);
      vector<int32_t> flags(nb * 3);
      vector<double> par(nb * 15, 0.0);
      for (int b = 0; b < nb; b++)
      {
        bc &c = run_input.bc_list(b);
        const int fl = c.get_bc_flag();
        flags[b * 3 + 0] = fl;
        flags[b * 3 + 1] = (fl == SUB_IN_CHAR) ? c.pressure_ramp : 0;
        flags[b * 3 + 2] = (fl == ISOTHERM_WALL || fl == ADIABAT_WALL) ? (run_input.wall_model ? c.use_wm : 0) : 0;
        double *q = &par[b * 15];
        const bool has_vel = (fl == SUB_IN_SIMP || fl == SUP_IN || fl == ISOTHERM_WALL || fl == CHAR || fl == ADIABAT_WALL);
        if (fl == SUB_IN_SIMP || fl == SUP_IN || fl == CHAR) q[0] = c.rho;
        if (has_vel) for (int d = 0; d < 3; d++) q[1 + d] = c.velocity(d);
        if (fl == SUB_OUT_SIMP || fl == SUB_OUT_CHAR || fl == SUP_IN || fl == CHAR) q[4] = c.p_static;
        if (fl == SUP_IN || fl == ISOTHERM_WALL || fl == CHAR) q[5] = c.T_static;
        if (fl == SUB_IN_CHAR) q[6] = c.p_total;
        if (fl == SUB_IN_CHAR || fl == SUB_OUT_SIMP || fl == SUB_OUT_CHAR) q[7] = c.T_total;
        if (fl == SUB_IN_CHAR || fl == SUP_IN || fl == CHAR) { q[8] = c.nx; q[9] = c.ny; q[10] = c.nz; }
        if (fl == SUB_IN_CHAR && c.pressure_ramp)
        {
          q[11] = c.p_ramp_coeff; q[12] = c.T_ramp_coeff; q[13] = c.p_total_old; q[14] = c.T_total_old;
        }
      }
      put_i("bc_flags", flags.data(), {3, nb});
      put_d("bc_params", par.data(), {15, nb});
      put_scalar("bc_R_ref", run_input.viscous ? run_input.R_ref : run_input.R_gas);
      put_scalar("ramp_counter", run_input.ramp_counter);
    }
  }
  // LES closure (src/eles.cpp:2395-2650)
  if (run_input.LES)
  {
    put_scalar("LES", run_input.LES);
    put_scalar("SGS_model", run_input.SGS_model);
    put_scalar("C_s", run_input.C_s);
    put_scalar("filter_ratio", run_input.filter_ratio);
    put_scalar("Kappa", run_input.Kappa);
    put_scalar("prandtl_t", run_input.prandtl_t);
    if (run_input.SGS_model == 0) put_arr("wall_distance", E->wall_distance);
    put_arr("Jacobian_fpts", E->Jacobian_fpts); // used by extrapolate_sgsFlux to take the SGS flux back to physical space
    // similarity (2: WALE + similarity, 4: similarity) and SVV (3) closures: the filter matrix of calc_sgs_terms
    if (run_input.SGS_model >= 2) put_arr("filter_upts", E->filter_upts);
  }
  // integral diagnostics (src/eles.cpp:5485-5627): volume cubature interpolation, weights, Jacobians
  if (run_input.n_integral_quantities != 0)
  {
    put_arr("opp_volume_cubpts", E->opp_volume_cubpts);
    put_arr("weight_volume_cubpts", E->weight_volume_cubpts);
    const int nc = E->weight_volume_cubpts.get_dim(0);
    vector<double> dj((size_t)nc * n_eles);
    for (int j = 0; j < nc; j++)
      for (int i = 0; i < n_eles; i++) dj[j + (size_t)nc * i] = E->vol_detjac_vol_cubpts(j)(i);
    put_d("vol_detjac_vol_cubpts", dj.data(), {nc, n_eles});
    vector<int32_t> ids(run_input.n_integral_quantities);
    for (int m = 0; m < run_input.n_integral_quantities; m++)
    {
      const string &q = run_input.integral_quantities(m);
      ids[m] = q == "kineticenergy" ? 0 : q == "enstropy" ? 1 : q == "pressuredilatation" ? 2 : q == "straincolonproduct" ? 3 : 4;
    }
    put_i("integral_quantity_ids", ids.data(), {run_input.n_integral_quantities});
  }
  // over-integration (src/eles_hexas.cpp:1096-1129, src/eles.cpp:1480-1545)
  if (run_input.over_int)
  {
    put_arr("opp_over_int_cubpts", E->opp_over_int_cubpts);
    put_arr("over_int_filter", E->over_int_filter);
    put_arr("JGinv_over_int_cubpts", E->JGinv_over_int_cubpts);
    put_scalar("over_int", run_input.over_int);
  }
  // shock capturing (src/eles.cpp:2918, src/eles_hexas.cpp:953-1059): the registered matrices and the set of
  // highest modes of the Persson sensor
  if (run_input.shock_cap)
  {
    put_arr("inv_vandermonde", E->inv_vandermonde);
    put_arr("exp_filter", E->exp_filter);
    // the class's own mode norms and highest modes (hexes / quads: a mode of degree `order` in any direction, src/eles_hexas.cpp:
    // 1007-1059; prisms: degree `order` on the triangle or along the line, src/eles_pris.cpp:716-722; tetrahedra: the modes past
    // the P(order-1) space of an orthonormal basis -- all norms one, src/eles_tets.cpp:748-797)
    eles_hexas *EH = (etype == 4) ? &FlowSol.mesh_eles_hexas : NULL;
    eles_quads *EQ = (etype == 1) ? &FlowSol.mesh_eles_quads : NULL;
    eles_pris *EP = (etype == 3) ? &FlowSol.mesh_eles_pris : NULL;
    if (EH) put_arr("norm_basis_persson", EH->norm_basis_persson);
    if (EQ) put_arr("norm_basis_persson", EQ->norm_basis_persson);
    if (EP) put_arr("norm_basis_persson", EP->norm_basis_persson);
    if (etype == 2)
    {
      vector<double> ones(n_upts, 1.0);
      put_d("norm_basis_persson", ones.data(), {n_upts});
    }
    if (etype == 0) { fprintf(stderr, "harness: shock capturing on triangles is not dumped\n"); return 1; }
    vector<int32_t> hi(n_upts);
    for (int j = 0; j < n_upts; j++)
    {
      int x = 0, y = 0, z = 0;
      const int p = run_input.order;
      if (EH)
      {
        EH->get_legendre_basis_3D_index(j, p, x, y, z);
        hi[j] = (x == p || y == p || z == p) ? 1 : 0;
      }
      else if (EQ)
      {
        EQ->get_legendre_basis_2D_index(j, p, x, y);
        hi[j] = (x == p || y == p) ? 1 : 0;
      }
      else if (EP)
      {
        EP->get_pris_basis_index(j, p, x, y, z);
        hi[j] = (x + y == p || z == p) ? 1 : 0;
      }
      else
        hi[j] = (j >= p * (p + 1) * (p + 2) / 6) ? 1 : 0;
    }
    put_i("persson_high_modes", hi.data(), {n_upts});
    put_scalar("s0", run_input.s0);
    put_scalar("shock_det_field", run_input.shock_det_field);
    put_scalar("shock_cap", run_input.shock_cap);
  }
  if (level >= 1)
    for (size_t c = 0; c < Es.size(); c++)
    {
      eles *X = Es[c];
      const string q = pre(c);
      put_arr(q + "detjac_upts", X->detjac_upts);
      put_arr(q + "JGinv_upts", X->JGinv_upts);
      put_arr(q + "detjac_fpts", X->detjac_fpts);
      put_arr(q + "JGinv_fpts", X->JGinv_fpts);
      put_arr(q + "tdA_fpts", X->tdA_fpts);
      put_arr(q + "norm_fpts", X->norm_fpts);
      put_arr(q + "pos_upts", X->pos_upts);
      put_arr(q + "pos_fpts", X->pos_fpts);
    }

  for (size_t c = 0; c < Es.size(); c++) put_arr(pre(c) + "u_init", Es[c]->disu_upts(0));
  // one dump of a member array per class: put_all("s0_disu_fpts", [](eles *X) -> hf_array<double> & { return X->disu_fpts; })
  auto put_all = [&](const string &name, hf_array<double> &(*get)(eles *)) {
    for (size_t c = 0; c < Es.size(); c++) put_arr(pre(c) + name, get(Es[c]));
  };
  if (getenv("HFX_DUMP_PPTS"))
  {
    // plot-point interpolation (eles::set_opp_p src/eles.cpp:3600, calc_disu_ppts :3757): the operator, the plot
    // points and the interpolated initial state of every element, (n_ppts, n_eles, n_fields)
    const int npp = E->opp_p.get_dim(0), nf = E->disu_upts(0).get_dim(2);
    put_arr("opp_p", E->opp_p);
    put_arr("loc_ppts", E->loc_ppts);
    vector<double> pp((size_t)npp * n_eles * nf);
    hf_array<double> one(npp, nf);
    for (int i = 0; i < n_eles; i++)
    {
      E->calc_disu_ppts(i, one);
      for (int k = 0; k < nf; k++)
        for (int j = 0; j < npp; j++) pp[j + (size_t)npp * (i + (size_t)n_eles * k)] = one(j, k);
    }
    put_d("disu_ppts", pp.data(), {npp, n_eles, nf});
  }

  // wall clock of the RK loop alone (setup excluded): what bench.py's cpu_baseline leg reports for the genuine reference
  const auto t_loop0 = std::chrono::steady_clock::now();
  for (int step = 0; step < n_steps; step++)
  {
    calc_time_step(&FlowSol);
    if (run_input.dt_type != 0)
    {
      // CFL time stepping (src/solver.cpp:484-549, src/eles.cpp:1267-1356)
      char nm[64];
      snprintf(nm, sizeof nm, "dt_step%d", step);
      put_scalar(nm, run_input.dt);
      if (step == 0)
      {
        put_arr("h_ref", E->h_ref);
        put_scalar("CFL", run_input.CFL);
      }
      if (run_input.dt_type == 2)
      {
        snprintf(nm, sizeof nm, "dt_local_step%d", step);
        put_arr(nm, E->dt_local);
      }
    }
    for (int rk = 0; rk < RKSteps; rk++)
    {
      if (step == 0 && rk == 0 && level >= 1)
      {
        // the sequence of CalcResidual (src/solver.cpp:50-223) for a
        // single-rank, LES-off, RANS-off, forcing-off run, with dumps between calls
        int i;
        if (run_input.LES == 1 && (run_input.SGS_model == 2 || run_input.SGS_model == 3 || run_input.SGS_model == 4))
        {
          // first stage of a step: filtered solution, Leonard tensors (src/solver.cpp:55-62)
          for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->calc_sgs_terms();
          put_arr("s0_disuf_upts", E->disuf_upts);
          put_arr("s0_u_after_sgs_terms", E->disu_upts(0));
          if (run_input.SGS_model != 3)
          {
            put_arr("s0_Lu", E->Lu);
            put_arr("s0_Le", E->Le);
          }
        }
        for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->extrapolate_solution();
        if (level >= 2) put_all("s0_disu_fpts", [](eles *X) -> hf_array<double> & { return X->disu_fpts; });
        if (run_input.viscous)
        {
          for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->calculate_gradient();
          if (level >= 2) put_all("s0_grad_disu_upts_ref", [](eles *X) -> hf_array<double> & { return X->grad_disu_upts; });
        }
        if (run_input.over_int)
          for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->evaluate_invFlux_over_int();
        else
          for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->evaluate_invFlux();
        if (level >= 2) put_all("s0_tdisf_upts_inv", [](eles *X) -> hf_array<double> & { return X->tdisf_upts; });
        for (i = 0; i < FlowSol.n_int_inter_types; i++) FlowSol.mesh_int_inters(i).calculate_common_invFlux();
        for (i = 0; i < FlowSol.n_bdy_inter_types; i++)
          FlowSol.mesh_bdy_inters(i).evaluate_boundaryConditions_invFlux(&FlowSol, FlowSol.time);
        if (level >= 2)
        {
          put_all("s0_norm_tconf_fpts_inv", [](eles *X) -> hf_array<double> & { return X->norm_tconf_fpts; });
          if (run_input.viscous) put_all("s0_delta_disu_fpts", [](eles *X) -> hf_array<double> & { return X->delta_disu_fpts; });
        }
        if (run_input.viscous)
        {
          for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->correct_gradient();
          if (level >= 2)
          {
            put_all("s0_grad_disu_upts", [](eles *X) -> hf_array<double> & { return X->grad_disu_upts; });
            put_all("s0_grad_disu_fpts", [](eles *X) -> hf_array<double> & { return X->grad_disu_fpts; });
          }
          for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->evaluate_viscFlux();
          if (level >= 2) put_all("s0_tdisf_upts", [](eles *X) -> hf_array<double> & { return X->tdisf_upts; });
          if (run_input.LES)
          {
            // src/solver.cpp:162-167
            for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->extrapolate_sgsFlux();
            if (level >= 2)
            {
              put_arr("s0_sgsf_upts", E->sgsf_upts);
              put_arr("s0_sgsf_fpts", E->sgsf_fpts);
            }
          }
        }
        for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->extrapolate_totalFlux();
        if (level >= 2) put_all("s0_norm_tdisf_fpts", [](eles *X) -> hf_array<double> & { return X->norm_tdisf_fpts; });
        for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->calculate_divergence();
        if (level >= 2) put_all("s0_div_tconf_upts_disc", [](eles *X) -> hf_array<double> & { return X->div_tconf_upts(0); });
        if (run_input.viscous)
        {
          for (i = 0; i < FlowSol.n_int_inter_types; i++) FlowSol.mesh_int_inters(i).calculate_common_viscFlux();
          for (i = 0; i < FlowSol.n_bdy_inter_types; i++)
            FlowSol.mesh_bdy_inters(i).evaluate_boundaryConditions_viscFlux(FlowSol.time);
          if (level >= 2) put_all("s0_norm_tconf_fpts", [](eles *X) -> hf_array<double> & { return X->norm_tconf_fpts; });
        }
        for (i = 0; i < FlowSol.n_ele_types; i++) FlowSol.mesh_eles(i)->calculate_corrected_divergence();
        put_all("s0_div_tconf_upts", [](eles *X) -> hf_array<double> & { return X->div_tconf_upts(0); });
        // residual norms as the reference's monitor computes them (eles.cpp:5045-5074):
        // L1 / L2 sums over all upts of |div_tconf/detjac - src|
        for (size_t c = 0; c < Es.size(); c++)
        {
          vector<double> res(2 * n_fields);
          for (int f = 0; f < n_fields; f++)
          {
            res[f] = Es[c]->compute_res_upts(1, f);
            res[n_fields + f] = Es[c]->compute_res_upts(2, f);
          }
          put_d(pre(c) + "s0_res_sums", res.data(), {n_fields, 2});
        }
        if (run_input.n_integral_quantities != 0)
        {
          // state u_init, corrected gradients of this residual (output::CalcIntegralQuantities, src/output.cpp:2017)
          hf_array<double> iq(run_input.n_integral_quantities);
          iq.initialize_to_zero();
          E->CalcIntegralQuantities(run_input.n_integral_quantities, iq);
          put_arr("s0_integral_quantities", iq);
        }
      }
      else
        CalcResidual(FlowSol.ini_iter + step, rk, &FlowSol);

      for (int j = 0; j < FlowSol.n_ele_types; j++)
        FlowSol.mesh_eles(j)->AdvanceSolution(rk, run_input.adv_type);
      if (run_input.shock_cap)
      {
        if (step == 0 && rk == 0) put_arr("s0_u_before_shock_capture", E->disu_upts(0));
        for (int j = 0; j < FlowSol.n_ele_types; j++) FlowSol.mesh_eles(j)->shock_capture();
        if (step == 0 && rk == 0) put_arr("s0_sensor", E->sensor);
      }

      if (level >= 1 || rk == RKSteps - 1)
      {
        char nm[64];
        snprintf(nm, sizeof nm, "u_step%d_stage%d", step, rk);
        for (size_t c = 0; c < Es.size(); c++) put_arr(pre(c) + nm, Es[c]->disu_upts(0));
      }
      if (rk == RKSteps - 1 && getenv("HFX_DUMP_DIV"))
      {
        // full-size parity fixtures (oracle/capture_fullsize.py): the residual of the step's last stage
        char nm[64];
        snprintf(nm, sizeof nm, "div_step%d", step);
        for (size_t c = 0; c < Es.size(); c++) put_arr(pre(c) + nm, Es[c]->div_tconf_upts(0));
      }
    }
    FlowSol.time += run_input.dt;
    run_input.time = FlowSol.time;
    if (run_input.pressure_ramp) run_input.ramp_counter++; /* src/HiFiLES.cpp:224-225 */
  }
  {
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
    fprintf(stderr, "ref_harness: RK loop %d steps %d stages %.6f s\n", n_steps, RKSteps, secs);
  }
  if (getenv("HFX_DUMP_RESTART"))
  {
    // the ASCII restart of the final state, exactly as output::write_restart_ascii composes it
    // (src/output.cpp:1753-1818): time, then per element class the info block and the data block
    {
      ofstream rf("restart_dump.dat");
      rf.precision(15);
      rf << FlowSol.time << endl;
      for (int i = 0; i < FlowSol.n_ele_types; i++)
        if (FlowSol.mesh_eles(i)->get_n_eles() != 0)
        {
          FlowSol.mesh_eles(i)->write_restart_info_ascii(rf);
          FlowSol.mesh_eles(i)->write_restart_data_ascii(rf);
        }
    }
    ifstream in("restart_dump.dat", ios::binary);
    string txt((istreambuf_iterator<char>(in)), istreambuf_iterator<char>());
    vector<int32_t> bytes(txt.size());
    for (size_t i = 0; i < txt.size(); i++) bytes[i] = (unsigned char)txt[i];
    put_i("restart_ascii", bytes.data(), {(int64_t)bytes.size()});
    put_scalar("restart_time", FlowSol.time);
  }
  fclose(g_out);
  // the dump is complete; leave without running the reference's destructors (with the similarity closures one of them
  // frees an invalid pointer at exit -- the reference's own main never gets that far with these arrays alive)
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}
