#include "solver.hpp"

#include <cmath>

solution::~solution()
{
  // face blocks reference element blocks: release them first
  mesh_int_inters.setup(0);
  mesh_eles_quads.free_device();
  mesh_eles_hexas.free_device();
  if (ctx) hfx_ctx_destroy(ctx);
}

// ---- box mesh ----------------------------------------------------------------------------
void box_mesh::generate()
{
  const int nx = n[0], ny = n[1], nz = (dims == 3) ? n[2] : 0;
  const int NV = nv();
  xv.assign((size_t)NV * dims, 0.0);
  const double two_pi_over_l = 2.0 * 3.14159265358979323846 / length;
  auto coord = [&](int i, int nn) { return (i == nn) ? length : length * ((double)i / nn); };
  // periodic images must get bit-identical displacements: evaluate them at the wrapped coordinate
  auto wrap = [&](int i, int nn) { return (i == nn) ? 0.0 : length * ((double)i / nn); };
  for (int k = 0; k <= nz; k++)
    for (int j = 0; j <= ny; j++)
      for (int i = 0; i <= nx; i++)
      {
        const int v = i + (nx + 1) * (j + (ny + 1) * k);
        double x = coord(i, nx), y = coord(j, ny), z = (dims == 3) ? coord(k, nz) : 0.0;
        if (amp != 0.0)
        {
          const double xw = wrap(i, nx), yw = wrap(j, ny), zw = (dims == 3) ? wrap(k, nz) : 0.0;
          const double kk = two_pi_over_l;
          if (dims == 3)
          {
            x += amp * std::sin(kk * yw + 0.3) * std::cos(kk * zw + 0.5);
            y += amp * std::cos(kk * xw + 0.7) * std::sin(kk * zw + 0.2);
            z += amp * std::sin(kk * xw + 0.1) * std::sin(kk * yw + 0.9);
          }
          else
          {
            x += amp * std::sin(kk * yw + 0.3);
            y += amp * std::cos(kk * xw + 0.7);
          }
        }
        xv[v] = x;
        xv[v + (size_t)NV] = y;
        if (dims == 3) xv[v + 2 * (size_t)NV] = z;
      }
}

// rotation tag of a face pair by coincidence of flux-point positions modulo the period
static int find_rot_tag(int_inters &I, eles *el, eles *er, int ele_l, int ele_r, int face_l, int face_r, int dims,
                        double length, double tol)
{
  const int n_rot = (I.inters_type == 0) ? 1 : 4;
  for (int rot = 0; rot < n_rot; rot++)
  {
    I.get_lut(rot);
    bool ok = true;
    for (int j = 0; j < I.n_fpts_per_inter && ok; j++)
    {
      const int ol = el->get_fpt_offset(ele_l, face_l, j) - el->n_fpts_per_ele * ele_l;
      const int orr = er->get_fpt_offset(ele_r, face_r, I.lut(j)) - er->n_fpts_per_ele * ele_r;
      for (int d = 0; d < dims; d++)
      {
        double diff = el->pos_fpts(ol, ele_l, d) - er->pos_fpts(orr, ele_r, d);
        diff -= length * std::round(diff / length);
        if (std::fabs(diff) > tol) { ok = false; break; }
      }
    }
    if (ok) return rot;
  }
  return -1;
}

int GeoPreprocess_box(solution *S, const box_mesh &mesh)
{
  input &in = S->run_input;
  S->n_dims = mesh.dims;
  S->num_cells_global = mesh.ne();
  S->mesh_eles.setup(S->n_ele_types);
  S->mesh_eles.initialize_to_value(nullptr);
  S->mesh_eles(1) = &S->mesh_eles_quads;
  S->mesh_eles(4) = &S->mesh_eles_hexas;
  const int dims = mesh.dims, nx = mesh.n[0], ny = mesh.n[1], nz = (dims == 3) ? mesh.n[2] : 1;
  const int NV = mesh.nv();
  if ((int)mesh.xv.size() != NV * dims) { S->err = "box mesh: vertex array has the wrong size"; return 1; }
  if (nx < 3 || ny < 3 || (dims == 3 && nz < 3)) { S->err = "box mesh: need >= 3 cells per direction (periodic matching)"; return 1; }

  eles *E = (dims == 3) ? (eles *)&S->mesh_eles_hexas : (eles *)&S->mesh_eles_quads;
  const int etype = (dims == 3) ? 4 : 1;
  const int nspt = (dims == 3) ? 8 : 4;
  if (E->setup(mesh.ne(), nspt, &in)) { S->err = E->last_error(); return 1; }
  // the classes without elements still answer get_n_eles() == 0
  eles *other = (dims == 3) ? (eles *)&S->mesh_eles_quads : (eles *)&S->mesh_eles_hexas;
  other->run_input = &in;

  // shape nodes: slot = r + 2 s + 4 t (eles_hexas.cpp:1198-1214, eles_quads eval_nodal_s_basis)
  hf_array<double> pos(dims);
  auto vid = [&](int i, int j, int k) { return i + (nx + 1) * (j + (ny + 1) * k); };
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
      {
        const int e = i + nx * (j + ny * k);
        for (int t = 0; t < (dims == 3 ? 2 : 1); t++)
          for (int s = 0; s < 2; s++)
            for (int r = 0; r < 2; r++)
            {
              const int v = vid(i + r, j + s, k + t);
              for (int d = 0; d < dims; d++) pos(d) = mesh.xv[v + (size_t)NV * d];
              E->set_shape_node(r + 2 * s + 4 * t, e, pos);
            }
      }
  if (E->set_transforms()) { S->err = E->last_error(); return 1; }
  // reference length for CFL time stepping: shortest edge through the element centre is not
  // needed by the fixed-dt configurations; use the cube root / square root of the volume
  for (int e = 0; e < E->n_eles; e++)
  {
    double vol = 0.0;
    for (int j = 0; j < E->n_upts_per_ele; j++) vol += E->detjac_upts(j, e);
    vol *= std::pow(2.0, dims) / E->n_upts_per_ele;
    E->h_ref(e) = (dims == 3) ? std::cbrt(vol) : std::sqrt(vol);
  }

  // interior faces.  Same numbering and left/right orientation as the reference derives from the
  // mesh (src/mesh.cpp set_face_connectivity + src/geometry.cpp:351-415): cells ascending, local
  // faces ascending, a face is created by the first (= lower-numbered) cell that meets it, which
  // becomes its LEFT side.  The orientation matters: the LDG switch reads the left normal only
  // (src/inters.cpp:568-581), and on meshes whose face normals carry rounding noise in n_x the
  // outcome depends on which side is left.
  S->mesh_int_inters.setup(S->n_int_inter_types);
  const int ftype = (dims == 3) ? 2 : 0;
  const int n_faces = dims * mesh.ne();
  for (int t = 0; t < 3; t++) S->mesh_int_inters(t).setup(t == ftype ? n_faces : 0, t, &in);
  int_inters &I = S->mesh_int_inters(ftype);
  // local face -> (axis, direction, neighbour's local face)
  static const int hex_face[6][3] = {{2, -1, 5}, {1, -1, 3}, {0, 1, 4}, {1, 1, 1}, {0, -1, 2}, {2, 1, 0}};
  static const int quad_face[4][3] = {{1, -1, 2}, {0, 1, 3}, {1, 1, 0}, {0, -1, 1}};
  const int nfaces_loc = (dims == 3) ? 6 : 4;
  const double tol = 1e-8 * mesh.length;
  int fi = 0;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
      {
        const int e = i + nx * (j + ny * k);
        for (int f = 0; f < nfaces_loc; f++)
        {
          const int *fd = (dims == 3) ? hex_face[f] : quad_face[f];
          int c[3] = {i, j, k};
          const int nn[3] = {nx, ny, nz};
          c[fd[0]] = (c[fd[0]] + fd[1] + nn[fd[0]]) % nn[fd[0]];
          const int er = c[0] + nx * (c[1] + ny * c[2]);
          if (er < e) continue; // created from the other side already
          const int rot = find_rot_tag(I, E, E, e, er, f, fd[2], dims, mesh.length, tol);
          if (rot < 0) { S->err = "Can't find coupled cyclic interface"; return 1; } /* src/geometry.cpp:442 */
          I.set_interior(fi++, etype, etype, e, er, f, fd[2], rot, S);
        }
      }
  if (fi != n_faces) { S->err = "box mesh: face count mismatch"; return 1; }
  if (I.failed()) { S->err = I.last_error(); return 1; }
  return 0;
}

int InitSolution(solution *S)
{
  S->ini_iter = 0;
  for (int i = 0; i < S->n_ele_types; i++)
    if (S->mesh_eles(i) && S->mesh_eles(i)->get_n_eles() != 0)
      if (S->mesh_eles(i)->set_ics(S->time)) { S->err = S->mesh_eles(i)->last_error(); return 1; }
  return 0;
}

int MoveToDevice(solution *S, int device)
{
  if (hfx_ctx_create(device, &S->ctx)) { S->err = hfx_last_error(); return 1; }
  hfx_params p;
  S->run_input.fill(p);
  if (hfx_ctx_set_params(S->ctx, &p)) { S->err = hfx_last_error(); return 1; }
  for (int i = 0; i < S->n_ele_types; i++)
    if (S->mesh_eles(i) && S->mesh_eles(i)->get_n_eles() != 0)
      if (S->mesh_eles(i)->mv_all_cpu_gpu(S->ctx)) { S->err = S->mesh_eles(i)->last_error(); return 1; }
  for (int i = 0; i < S->n_int_inter_types; i++)
    if (S->mesh_int_inters(i).mv_all_cpu_gpu(S->ctx, S)) { S->err = S->mesh_int_inters(i).last_error(); return 1; }
  return 0;
}

// ---- the stage scheduler: same call order as the reference --------------------------------
void CalcResidual(int /*in_file_num*/, int /*in_rk_stage*/, solution *FlowSol)
{
  int i;
  auto each_ele = [&](void (eles::*m)()) {
    for (i = 0; i < FlowSol->n_ele_types; i++)
      if (FlowSol->mesh_eles(i)) (FlowSol->mesh_eles(i)->*m)();
  };
  /*! Extrapolate the solution to the flux points. */
  each_ele(&eles::extrapolate_solution);
  if (FlowSol->run_input.viscous)
    /*! Compute the uncorrected transformed gradient of the solution at the solution points. */
    each_ele(&eles::calculate_gradient);
  /*! Compute the transformed inviscid flux at the solution points. */
  each_ele(&eles::evaluate_invFlux);
  /*! Compute the transformed normal inviscid numerical fluxes, common solution and corrections. */
  for (i = 0; i < FlowSol->n_int_inter_types; i++) FlowSol->mesh_int_inters(i).calculate_common_invFlux();
  if (FlowSol->run_input.viscous)
  {
    /*! Compute physical corrected gradient of the solution at the solution and flux points. */
    each_ele(&eles::correct_gradient);
    /*! Compute discontinuous transformed viscous flux at upts and add to total transformed flux. */
    each_ele(&eles::evaluate_viscFlux);
  }
  /*! Compute the transformed normal discontinuous total flux at flux points. */
  each_ele(&eles::extrapolate_totalFlux);
  /*! Compute the transformed divergence of total flux at solution points. */
  each_ele(&eles::calculate_divergence);
  if (FlowSol->run_input.viscous)
    /*! Compute transformed normal interface viscous flux and add to transformed normal inviscid flux. */
    for (i = 0; i < FlowSol->n_int_inter_types; i++) FlowSol->mesh_int_inters(i).calculate_common_viscFlux();
  /*! Compute the transformed divergence of the continuous flux. */
  each_ele(&eles::calculate_corrected_divergence);
}

int RunSteps(solution *FlowSol, int n_steps)
{
  const int RKSteps = FlowSol->run_input.n_rk_stages();
  for (int i_steps = 0; i_steps < n_steps; i_steps++)
  {
    for (int i = 0; i < RKSteps; i++)
    {
      CalcResidual(FlowSol->ini_iter + i_steps, i, FlowSol);
      for (int j = 0; j < FlowSol->n_ele_types; j++)
        if (FlowSol->mesh_eles(j)) FlowSol->mesh_eles(j)->AdvanceSolution(i, FlowSol->run_input.adv_type);
    }
    FlowSol->time += FlowSol->run_input.dt;
    FlowSol->run_input.time = FlowSol->time;
  }
  for (int j = 0; j < FlowSol->n_ele_types; j++)
    if (FlowSol->mesh_eles(j) && FlowSol->mesh_eles(j)->failed())
    {
      FlowSol->err = FlowSol->mesh_eles(j)->last_error();
      return 1;
    }
  for (int j = 0; j < FlowSol->n_int_inter_types; j++)
    if (FlowSol->mesh_int_inters(j).failed())
    {
      FlowSol->err = FlowSol->mesh_int_inters(j).last_error();
      return 1;
    }
  return 0;
}
