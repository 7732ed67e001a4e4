// capi.cpp -- small extern "C" facade over the host-side mirror so that the Python test and
// bench plumbing (ctypes) can build a case, read its host arrays and drive it.  Declared in
// include/hfx_host.h.
#include "../../../include/hfx_host.h"

#include <cstring>
#include <map>
#include <string>

#include "solver.hpp"

struct hfxh_case
{
  solution S;
  box_mesh mesh;
  std::vector<hfx_inters *> faces;
  std::string err;
};

static thread_local std::string g_err;
extern "C" const char *hfxh_last_error(void) { return g_err.c_str(); }

static eles *the_eles(hfxh_case *c) { return c->S.n_dims == 3 ? (eles *)&c->S.mesh_eles_hexas : (eles *)&c->S.mesh_eles_quads; }
static int_inters *the_faces(hfxh_case *c) { return &c->S.mesh_int_inters(c->S.n_dims == 3 ? 2 : 0); }

extern "C" int hfxh_case_create(const hfxh_case_desc *d, hfxh_case **out)
{
  if (!d || !out) { g_err = "hfxh_case_create: NULL argument"; return 1; }
  hfxh_case *c = new hfxh_case();
  input &in = c->S.run_input;
  in.equation = 0;
  in.viscous = d->viscous; in.order = d->order;
  in.riemann_solve_type = d->riemann_solve_type; in.vis_riemann_solve_type = 0;
  in.ic_form = d->ic_form; in.adv_type = d->adv_type; in.dt_type = d->dt_type; in.dt = d->dt; in.CFL = d->CFL;
  in.ldg_tau = d->ldg_tau; in.ldg_beta = d->ldg_beta;
  in.upts_type_hexa = in.upts_type_quad = d->upts_type;
  in.vcjh_scheme_hexa = in.vcjh_scheme_quad = d->vcjh_scheme;
  in.eta_hexa = in.eta_quad = d->eta;
  in.gamma = d->gamma; in.prandtl = d->prandtl; in.S_gas = d->S_gas; in.T_gas = d->T_gas; in.R_gas = d->R_gas;
  in.mu_gas = d->mu_gas; in.fix_vis = d->fix_vis;
  in.Mach_free_stream = d->Mach_free_stream; in.L_free_stream = d->L_free_stream;
  in.T_free_stream = d->T_free_stream; in.rho_free_stream = d->rho_free_stream;
  in.Mach_c_ic = d->Mach_c_ic; in.T_c_ic = d->T_c_ic; in.rho_c_ic = d->rho_c_ic;
  in.u_c_ic = d->u_c_ic; in.v_c_ic = d->v_c_ic; in.w_c_ic = d->w_c_ic; in.p_c_ic = d->p_c_ic;
  in.dx_cyclic = in.dy_cyclic = in.dz_cyclic = d->length;
  if (d->p_res >= 2) in.p_res = d->p_res;
  in.LES = d->LES;
  if (d->LES)
  {
    in.SGS_model = d->SGS_model; in.C_s = d->C_s; in.filter_ratio = d->filter_ratio; in.filter_type = d->filter_type;
    if (d->prandtl_t > 0) in.prandtl_t = d->prandtl_t;
  }
  in.over_int = d->over_int; in.over_int_order = d->over_int_order;
  in.shock_cap = d->shock_cap; in.shock_det_field = d->shock_det_field; in.s0 = d->s0;
  if (d->shock_cap)
  {
    in.expf_fac = d->expf_fac; in.expf_order = d->expf_order; in.expf_cutoff = d->expf_cutoff;
  }
  if (d->loc_1d_upts)
  {
    in.loc_1d_upts_override.setup(d->order + 1);
    for (int i = 0; i <= d->order; i++) in.loc_1d_upts_override(i) = d->loc_1d_upts[i];
  }
  if (in.setup_params(g_err)) { delete c; return 1; }
  for (int i = 0; i < d->n_bcs; i++)
  {
    const hfxh_bc_desc &b = d->bcs[i];
    bc_spec s;
    s.flag = b.flag; s.pressure_ramp = b.pressure_ramp;
    s.rho = b.rho; s.u = b.u; s.v = b.v; s.w = b.w; s.p_static = b.p_static; s.T_static = b.T_static;
    s.p_total = b.p_total; s.T_total = b.T_total; s.T_total_given = b.T_total >= 0;
    s.nx = b.nx; s.ny = b.ny; s.nz = b.nz; s.mach = b.mach;
    s.p_ramp_coeff = b.p_ramp_coeff; s.T_ramp_coeff = b.T_ramp_coeff; s.p_total_old = b.p_total_old;
    s.T_total_old = b.T_total_old; s.T_total_old_given = b.T_total_old >= 0;
    in.bc_specs.push_back(s);
  }
  if (in.read_boundary_param(g_err)) { delete c; return 1; }
  for (int i = 0; i < 6; i++) c->mesh.side_bc[i] = (d->n_bcs > 0) ? d->side_bc[i] : -1;

  c->mesh.dims = d->dims;
  c->S.nproc = d->nproc > 0 ? d->nproc : 1;
  c->S.rank = d->nproc > 0 ? d->rank : 0;
  {
    int r = c->S.rank;
    for (int i = 0; i < 3; i++)
    {
      c->mesh.pgrid[i] = (d->nproc > 0 && d->pgrid[i] > 0 && i < d->dims) ? d->pgrid[i] : 1;
      c->mesh.pcoord[i] = r % c->mesh.pgrid[i];
      r /= c->mesh.pgrid[i];
    }
  }
  for (int i = 0; i < 3; i++) c->mesh.n[i] = d->n[i];
  for (int i = 0; i < 3; i++) c->mesh.self_partition[i] = (i < d->dims) ? d->self_partition[i] : 0;
  c->mesh.length = d->length;
  c->mesh.amp = d->amp;
  if (d->xv)
    c->mesh.xv.assign(d->xv, d->xv + (size_t)c->mesh.nv() * d->dims);
  else
    c->mesh.generate();
  if (GeoPreprocess_box(&c->S, c->mesh)) { g_err = c->S.err; delete c; return 1; }
  if (InitSolution(&c->S)) { g_err = c->S.err; delete c; return 1; }
  *out = c;
  return 0;
}

extern "C" int hfxh_case_destroy(hfxh_case *c)
{
  delete c;
  return 0;
}

extern "C" int hfxh_case_sizes(hfxh_case *c, int sizes[8])
{
  eles *E = the_eles(c);
  sizes[0] = E->n_eles; sizes[1] = E->n_upts_per_ele; sizes[2] = E->n_fpts_per_ele; sizes[3] = E->n_fields;
  sizes[4] = E->n_dims; sizes[5] = E->order; sizes[6] = E->ele_type; sizes[7] = c->S.run_input.n_rk_stages();
  return 0;
}

extern "C" int hfxh_case_params(hfxh_case *c, hfx_params *p)
{
  c->S.run_input.fill(*p);
  return 0;
}

extern "C" int hfxh_case_get_array(hfxh_case *c, const char *name, const double **ptr, int dims[4])
{
  eles *E = the_eles(c);
  std::string n(name);
  hf_array<double> *a = nullptr;
  if (n == "opp_0") a = &E->opp_0;
  else if (n == "opp_3") a = &E->opp_3;
  else if (n == "opp_6") a = &E->opp_6;
  else if (n.rfind("opp_", 0) == 0 && n.size() == 7)
  {
    const int which = n[4] - '0', d = n[6] - '0';
    if (d < 0 || d >= E->n_dims) { g_err = "bad operator dimension"; return 1; }
    if (which == 1) a = &E->opp_1(d);
    else if (which == 2) a = &E->opp_2(d);
    else if (which == 4 && E->viscous) a = &E->opp_4(d);
    else if (which == 5 && E->viscous) a = &E->opp_5(d);
  }
  else if (n == "Jacobian_fpts") a = &E->Jacobian_fpts;
  else if (n == "opp_p") a = &E->opp_p;
  else if (n == "loc_ppts") a = &E->loc_ppts;
  else if (n == "inv_vandermonde") a = &E->inv_vandermonde;
  else if (n == "vandermonde") a = &E->vandermonde;
  else if (n == "exp_filter") a = &E->exp_filter;
  else if (n == "norm_basis_persson") a = &E->norm_basis_persson;
  else if (n == "opp_over_int_cubpts") a = &E->opp_over_int_cubpts;
  else if (n == "over_int_filter") a = &E->over_int_filter;
  else if (n == "filter_upts") a = &E->filter_upts;
  else if (n == "wall_distance" && E->wall_distance.get_dim(0) > 0) a = &E->wall_distance;
  else if (n == "filter_upts_1D") a = &E->filter_upts_1D;
  else if (n == "JGinv_over_int_cubpts") a = &E->JGinv_over_int_cubpts;
  else if (n == "loc_over_int_cubpts") a = &E->loc_over_int_cubpts;
  else if (n == "detjac_upts") a = &E->detjac_upts;
  else if (n == "JGinv_upts") a = &E->JGinv_upts;
  else if (n == "detjac_fpts") a = &E->detjac_fpts;
  else if (n == "JGinv_fpts") a = &E->JGinv_fpts;
  else if (n == "tdA_fpts") a = &E->tdA_fpts;
  else if (n == "norm_fpts") a = &E->norm_fpts;
  else if (n == "pos_upts") a = &E->pos_upts;
  else if (n == "pos_fpts") a = &E->pos_fpts;
  else if (n == "shape") a = &E->shape;
  else if (n == "loc_upts") a = &E->loc_upts;
  else if (n == "tloc_fpts") a = &E->tloc_fpts;
  else if (n == "tnorm_fpts") a = &E->tnorm_fpts;
  else if (n == "loc_1d_upts") a = &E->loc_1d_upts;
  else if (n == "disu_upts0") a = &E->disu_upts(0);
  else if (n == "disu_upts1") a = &E->disu_upts(1);
  else if (n == "div_tconf_upts") a = &E->div_tconf_upts(0);
  else if (n == "grad_disu_upts") a = &E->grad_disu_upts;
  else if (n == "h_ref") a = &E->h_ref;
  if (!a) { g_err = "hfxh_case_get_array: unknown array " + n; return 1; }
  *ptr = a->get_ptr_cpu();
  for (int i = 0; i < 4; i++) dims[i] = a->get_dim(i);
  return 0;
}

extern "C" int hfxh_case_get_faces(hfxh_case *c, const int **L, const int **R, int *n_fpts_per_inter, int *n_inters)
{
  int_inters *I = the_faces(c);
  *L = I->disu_fpts_l.get_ptr_cpu();
  *R = I->disu_fpts_r.get_ptr_cpu();
  *n_fpts_per_inter = I->n_fpts_per_inter;
  *n_inters = I->n_inters;
  return 0;
}

static bdy_inters *the_bdy_faces(hfxh_case *c) { return &c->S.mesh_bdy_inters(c->S.n_dims == 3 ? 2 : 0); }

extern "C" int hfxh_case_get_bdy_faces(hfxh_case *c, const int **L, const int **boundary_id, int *n_fpts_per_inter, int *n_inters)
{
  bdy_inters *B = the_bdy_faces(c);
  *L = B->disu_fpts_l.get_ptr_cpu();
  *boundary_id = B->boundary_id.get_ptr_cpu();
  *n_fpts_per_inter = B->n_fpts_per_inter;
  *n_inters = B->n_inters;
  return 0;
}

extern "C" int hfxh_case_get_bcs(hfxh_case *c, const hfx_bc **bcs, int *n_bcs, double *R_ref, int *ramp_counter)
{
  input &in = c->S.run_input;
  *bcs = in.bc_list.data();
  *n_bcs = (int)in.bc_list.size();
  *R_ref = in.bc_R_ref();
  *ramp_counter = in.ramp_counter;
  return 0;
}

static mpi_inters *the_mpi_faces(hfxh_case *c) { return &c->S.mesh_mpi_inters(c->S.n_dims == 3 ? 2 : 0); }

extern "C" int hfxh_case_get_mpi_faces(hfxh_case *c, const int **L, const int **Rlut, int *n_fpts_per_inter, int *n_inters,
                                       const int **nout_proc)
{
  mpi_inters *M = the_mpi_faces(c);
  *L = M->disu_fpts_l.get_ptr_cpu();
  *Rlut = M->disu_fpts_r.get_ptr_cpu();
  *n_fpts_per_inter = M->n_fpts_per_inter;
  *n_inters = M->n_inters;
  *nout_proc = M->Nout_proc.get_ptr_cpu();
  return 0;
}

extern "C" int hfxh_case_get_mpi_segments(hfxh_case *c, const int **peer, const int **send_first, const int **recv_first,
                                          const int **count, int *n_seg)
{
  mpi_inters *M = the_mpi_faces(c);
  *peer = M->seg_peer.data(); *send_first = M->seg_send.data(); *recv_first = M->seg_recv.data(); *count = M->seg_count.data();
  *n_seg = (int)M->seg_peer.size();
  return 0;
}

extern "C" int hfxh_case_set_reduce_min(hfxh_case *c, hfxh_reduce_min_cb fn, void *user)
{
  SetReduceMin(&c->S, fn, user);
  return 0;
}

extern "C" int hfxh_case_set_comm(hfxh_case *c, const char *unique_id)
{
  if (SetComm(&c->S, unique_id)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_time_partitioned(hfxh_case *c, int reps, double ms[8])
{
  if (TimePartitioned(&c->S, reps, ms)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_set_exchange(hfxh_case *c, hfxh_exchange_cb fn, void *user)
{
  SetExchange(&c->S, fn, user);
  return 0;
}

extern "C" int hfxh_case_mpi_handle(hfxh_case *c, hfx_inters **f)
{
  *f = the_mpi_faces(c)->device();
  return 0;
}

extern "C" int hfxh_case_run_partitioned(hfxh_case *c, int n_steps)
{
  if (RunStepsPartitionedFused(&c->S, n_steps)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_comm_info(hfxh_case *c, int *nranks, int *rank, int *device, char pci_bus_id[32])
{
  if (CommInfo(&c->S, nranks, rank, device, pci_bus_id)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_set_deferred(hfxh_case *c, int on)
{
  if (SetDeferred(&c->S, on != 0)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_to_device(hfxh_case *c, int device)
{
  if (MoveToDevice(&c->S, device)) { g_err = c->S.err; return 1; }
  c->faces.clear();
  for (int i = 0; i < c->S.n_int_inter_types; i++)
    if (c->S.mesh_int_inters(i).device()) c->faces.push_back(c->S.mesh_int_inters(i).device());
  for (int i = 0; i < c->S.n_bdy_inter_types; i++)
    if (c->S.mesh_bdy_inters(i).device()) c->faces.push_back(c->S.mesh_bdy_inters(i).device());
  return 0;
}

extern "C" int hfxh_case_handles(hfxh_case *c, hfx_ctx **ctx, hfx_eles **e, hfx_inters ***faces, int *n_face_blocks)
{
  if (!c->S.ctx) { g_err = "case is not on the device"; return 1; }
  *ctx = c->S.ctx;
  *e = the_eles(c)->device();
  *faces = c->faces.data();
  *n_face_blocks = (int)c->faces.size();
  return 0;
}

extern "C" int hfxh_case_CalcResidual(hfxh_case *c)
{
  CalcResidual(0, 0, &c->S);
  eles *E = the_eles(c);
  if (E->failed()) { g_err = E->last_error(); return 1; }
  if (the_faces(c)->failed()) { g_err = the_faces(c)->last_error(); return 1; }
  if (the_mpi_faces(c)->failed()) { g_err = the_mpi_faces(c)->last_error(); return 1; }
  if (the_bdy_faces(c)->failed()) { g_err = the_bdy_faces(c)->last_error(); return 1; }
  return 0;
}

extern "C" int hfxh_case_run(hfxh_case *c, int n_steps)
{
  if (RunSteps(&c->S, n_steps)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_write_restart(hfxh_case *c, const char *dir, int file_num)
{
  eles *E = the_eles(c);
  if (c->S.ctx && E->cp_disu_upts_gpu_cpu()) { g_err = E->last_error(); return 1; }
  if (write_restart_ascii(&c->S, dir ? dir : "", file_num)) { g_err = c->S.err; return 1; }
  return 0;
}

extern "C" int hfxh_case_read_restart(hfxh_case *c, const char *dir, int file_num, int n_files)
{
  if (read_restart_ascii(&c->S, dir ? dir : "", file_num, n_files)) { g_err = c->S.err; return 1; }
  eles *E = the_eles(c);
  if (c->S.ctx && hfx_eles_upload(E->device(), HFX_DISU_UPTS0, E->disu_upts(0).get_ptr_cpu())) { g_err = hfx_last_error(); return 1; }
  return 0;
}

extern "C" int hfxh_case_calc_time_step(hfxh_case *c, double *dt)
{
  if (calc_time_step(&c->S)) { g_err = c->S.err; return 1; }
  *dt = c->S.run_input.dt;
  return 0;
}

extern "C" int hfxh_case_sync_host(hfxh_case *c)
{
  eles *E = the_eles(c);
  // the gradient first: with deferred execution a pending stage then runs call by call and leaves every array of the
  // reference behind (asked for the state first it would run fused and keep the gradient on chip)
  // (... unless an earlier request has already made the last stage run fused: the gradient then stays what it was, as
  // hfx_run_steps(..., 3) leaves it)
  int grad_current = 1;
  if (E->viscous && E->device() && hfx_eles_is_current(E->device(), HFX_GRAD_DISU_UPTS, &grad_current)) { g_err = hfx_last_error(); return 1; }
  if (E->viscous && grad_current && E->cp_grad_disu_upts_gpu_cpu()) { g_err = E->last_error(); return 1; }
  if (E->cp_disu_upts_gpu_cpu() || E->cp_div_tconf_upts_gpu_cpu() || E->cp_array_gpu_cpu(HFX_DISU_UPTS1, E->disu_upts(1)))
  {
    g_err = E->last_error();
    return 1;
  }
  return 0;
}

extern "C" int hfxh_case_calc_disu_ppts(hfxh_case *c, const double **out, int dims[3])
{
  eles *E = the_eles(c);
  if (E->calc_disu_ppts_all()) { g_err = E->last_error(); return 1; }
  *out = E->disu_ppts.get_ptr_cpu();
  dims[0] = E->n_ppts_per_ele; dims[1] = E->n_eles; dims[2] = E->n_fields;
  return 0;
}


// ---- tetrahedra / prisms as producers of operators and metrics -------------------------------------------------------
struct hfxh_simplex
{
  input in;
  eles *E = nullptr;
  hf_array<double> high_modes_d;
  ~hfxh_simplex() { delete E; }
};

extern "C" int hfxh_simplex_create(int ele_type, int order, int viscous, int n_eles, const double *shape, const double *loc_1d_upts,
                                   hfxh_simplex **out)
{
  return hfxh_simplex_create_vcjh(ele_type, order, viscous, n_eles, shape, loc_1d_upts, 1, 0.0, out);
}

extern "C" int hfxh_simplex_create_vcjh(int ele_type, int order, int viscous, int n_eles, const double *shape, const double *loc_1d_upts,
                                        int vcjh_scheme, double c, hfxh_simplex **out)
{
  hfxh_simplex_keys k{};
  k.vcjh_scheme = vcjh_scheme; k.c = c; k.SGS_model = -1;
  return hfxh_simplex_create_keys(ele_type, order, viscous, n_eles, 0, shape, loc_1d_upts, &k, out);
}

extern "C" int hfxh_simplex_create_keys(int ele_type, int order, int viscous, int n_eles, int n_spts, const double *shape,
                                        const double *loc_1d_upts, const hfxh_simplex_keys *keys, hfxh_simplex **out)
{
  if (!keys) { g_err = "hfxh_simplex_create_keys: NULL keys"; return 1; }
  const int vcjh_scheme = keys->vcjh_scheme, SGS_model = keys->SGS_model, filter_type = keys->filter_type;
  const double c = keys->c, filter_ratio = keys->filter_ratio;
  if (!out || !shape || n_eles <= 0) { g_err = "hfxh_simplex_create: bad argument"; return 1; }
  if (ele_type != 2 && ele_type != 3) { g_err = "hfxh_simplex_create: ele_type must be 2 (tetrahedra) or 3 (prisms)"; return 1; }
  hfxh_simplex *s = new hfxh_simplex();
  s->in.equation = 0;
  s->in.order = order;
  s->in.viscous = viscous;
  s->in.vcjh_scheme_tet = s->in.vcjh_scheme_tri = vcjh_scheme;
  s->in.c_tet = s->in.c_tri = c;
  if (SGS_model >= 0)
  {
    s->in.LES = 1;
    s->in.SGS_model = SGS_model;
    s->in.filter_type = filter_type;
    s->in.filter_ratio = filter_ratio;
  }
  if (keys->shock_cap)
  {
    s->in.shock_cap = keys->shock_cap;
    s->in.expf_fac = keys->expf_fac; s->in.expf_order = keys->expf_order; s->in.expf_cutoff = keys->expf_cutoff;
  }
  if (loc_1d_upts)
  {
    s->in.loc_1d_upts_override.setup(order + 1);
    for (int i = 0; i <= order; i++) s->in.loc_1d_upts_override(i) = loc_1d_upts[i];
  }
  s->E = (ele_type == 2) ? (eles *)new eles_tets() : (eles *)new eles_pris();
  const int ns = n_spts > 0 ? n_spts : ((ele_type == 2) ? 4 : 6);
  if (ele_type == 2 ? (ns != 4 && ns != 10) : (ns != 6 && ns != 15))
  {
    g_err = "hfxh_simplex_create: shape nodes per element must be 4 or 10 (tetrahedra), 6 or 15 (prisms)";
    delete s;
    return 1;
  }
  if (s->E->setup(n_eles, ns, &s->in)) { g_err = s->E->last_error(); delete s; return 1; }
  hf_array<double> pos(3);
  for (int e = 0; e < n_eles; e++)
    for (int k = 0; k < ns; k++)
    {
      for (int d = 0; d < 3; d++) pos(d) = shape[d + 3 * (k + (size_t)ns * e)];
      s->E->set_shape_node(k, e, pos);
    }
  if (s->E->set_transforms()) { g_err = s->E->last_error(); delete s; return 1; }
  *out = s;
  return 0;
}

extern "C" int hfxh_simplex_get_array(hfxh_simplex *s, const char *name, const double **ptr, int dims[4])
{
  eles *E = s->E;
  std::string n(name);
  hf_array<double> *a = nullptr;
  if (n == "opp_0") a = &E->opp_0;
  else if (n == "opp_3") a = &E->opp_3;
  else if (n == "opp_6" && E->viscous) a = &E->opp_6;
  else if (n.rfind("opp_", 0) == 0 && n.size() == 7)
  {
    const int which = n[4] - '0', d = n[6] - '0';
    if (d < 0 || d >= E->n_dims) { g_err = "bad operator dimension"; return 1; }
    if (which == 1) a = &E->opp_1(d);
    else if (which == 2) a = &E->opp_2(d);
    else if (which == 4 && E->viscous) a = &E->opp_4(d);
    else if (which == 5 && E->viscous) a = &E->opp_5(d);
  }
  else if (n == "loc_upts") a = &E->loc_upts;
  else if (n == "tloc_fpts") a = &E->tloc_fpts;
  else if (n == "tnorm_fpts") a = &E->tnorm_fpts;
  else if (n == "detjac_upts") a = &E->detjac_upts;
  else if (n == "JGinv_upts") a = &E->JGinv_upts;
  else if (n == "detjac_fpts") a = &E->detjac_fpts;
  else if (n == "JGinv_fpts") a = &E->JGinv_fpts;
  else if (n == "tdA_fpts") a = &E->tdA_fpts;
  else if (n == "norm_fpts") a = &E->norm_fpts;
  else if (n == "pos_upts") a = &E->pos_upts;
  else if (n == "pos_fpts") a = &E->pos_fpts;
  else if (n == "filter_upts" && E->filter_upts.get_dim(0) > 0) a = &E->filter_upts;
  else if (n == "h_ref")
  {
    for (int i = 0; i < E->n_eles; i++) E->h_ref(i) = E->calc_h_ref_specific(i);
    a = &E->h_ref;
  }
  else if (n == "Jacobian_fpts" && E->Jacobian_fpts.get_dim(0) > 0) a = &E->Jacobian_fpts;
  else if (n == "inv_vandermonde" && E->inv_vandermonde.get_dim(0) > 0) a = &E->inv_vandermonde;
  else if (n == "exp_filter" && E->exp_filter.get_dim(0) > 0) a = &E->exp_filter;
  else if (n == "norm_basis_persson" && E->norm_basis_persson.get_dim(0) > 0) a = &E->norm_basis_persson;
  else if (n == "persson_high_modes" && E->persson_high_modes.get_dim(0) > 0)
  {
    // (an int array: handed out as doubles through this double-typed getter)
    hf_array<int> &hm = E->persson_high_modes;
    s->high_modes_d.setup(hm.get_dim(0));
    for (int i = 0; i < hm.get_dim(0); i++) s->high_modes_d(i) = hm(i);
    a = &s->high_modes_d;
  }
  if (!a) { g_err = "hfxh_simplex_get_array: unknown array " + n; return 1; }
  *ptr = a->get_ptr_cpu();
  for (int i = 0; i < 4; i++) dims[i] = a->get_dim(i);
  return 0;
}

extern "C" int hfxh_simplex_destroy(hfxh_simplex *s)
{
  delete s;
  return 0;
}
