#include "solver.hpp"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <cmath>

solution::~solution()
{
  if (comm) hfx_comm_destroy(comm);
  // face blocks reference element blocks: release them first
  mesh_int_inters.setup(0);
  mesh_bdy_inters.setup(0);
  mesh_mpi_inters.setup(0);
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
  // vertex coordinates are functions of the GLOBAL index, so that a block of a partitioned box gets
  // bit-identical vertices to the same cells of the unpartitioned one
  auto coord = [&](int i, int d) {
    const int gi = i + pcoord[d] * n[d], gn = n[d] * pgrid[d];
    return (gi == gn) ? length : length * ((double)gi / gn);
  };
  // periodic images must get bit-identical displacements: evaluate them at the wrapped coordinate
  auto wrap = [&](int i, int d) {
    const int gi = i + pcoord[d] * n[d], gn = n[d] * pgrid[d];
    return (gi == gn) ? 0.0 : length * ((double)gi / gn);
  };
  for (int k = 0; k <= nz; k++)
    for (int j = 0; j <= ny; j++)
      for (int i = 0; i <= nx; i++)
      {
        const int v = i + (nx + 1) * (j + (ny + 1) * k);
        double x = coord(i, 0), y = coord(j, 1), z = (dims == 3) ? coord(k, 2) : 0.0;
        if (amp != 0.0)
        {
          const double xw = wrap(i, 0), yw = wrap(j, 1), zw = (dims == 3) ? wrap(k, 2) : 0.0;
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
  S->mesh_eles.setup(S->n_ele_types);
  S->mesh_eles.initialize_to_value(nullptr);
  S->mesh_eles(1) = &S->mesh_eles_quads;
  S->mesh_eles(4) = &S->mesh_eles_hexas;
  const int dims = mesh.dims, nx = mesh.n[0], ny = mesh.n[1], nz = (dims == 3) ? mesh.n[2] : 1;
  const int NV = mesh.nv();
  if ((int)mesh.xv.size() != NV * dims) { S->err = "box mesh: vertex array has the wrong size"; return 1; }
  // local face -> (axis, direction, neighbour's local face)
  static const int hex_face[6][3] = {{2, -1, 5}, {1, -1, 3}, {0, 1, 4}, {1, 1, 1}, {0, -1, 2}, {2, 1, 0}};
  static const int quad_face[4][3] = {{1, -1, 2}, {0, 1, 3}, {1, 1, 0}, {0, -1, 1}};
  const int nfaces_loc = (dims == 3) ? 6 : 4;
  // a side is periodic when it has no group or a cyclic one; a direction when both of its sides are
  auto side_group = [&](int f) {
    const int g = mesh.side_bc[f];
    if (g < 0) return -1;
    if (g >= (int)in.bc_specs.size()) return -2;
    return in.bc_specs[g].flag == HFX_BC_CYCLIC ? -1 : g;
  };
  bool periodic[3] = {true, true, true};
  for (int f = 0; f < nfaces_loc; f++)
  {
    const int *fd = (dims == 3) ? hex_face[f] : quad_face[f];
    const int g = side_group(f);
    if (g == -2) { S->err = "box mesh: side refers to a boundary group that does not exist"; return 1; }
    if (g >= 0) periodic[fd[0]] = false;
  }
  for (int f = 0; f < nfaces_loc; f++)
  {
    const int *fd = (dims == 3) ? hex_face[f] : quad_face[f];
    if (!periodic[fd[0]] && side_group(f) < 0) { S->err = "box mesh: one side of a direction is cyclic, the other is not"; return 1; }
  }
  {
    const int nn[3] = {nx, ny, nz};
    int np = 1;
    for (int d = 0; d < dims; d++)
    {
      np *= mesh.pgrid[d];
      if (mesh.pgrid[d] < 1 || mesh.pcoord[d] < 0 || mesh.pcoord[d] >= mesh.pgrid[d]) { S->err = "box mesh: bad process grid"; return 1; }
      if (mesh.pgrid[d] == 1 && periodic[d] && !mesh.self_partition[d] && nn[d] < 3) { S->err = "box mesh: need >= 3 cells per direction (periodic matching)"; return 1; }
      if (mesh.self_partition[d] && (mesh.pgrid[d] != 1 || !periodic[d])) { S->err = "box mesh: self_partition needs a periodic direction that is not split"; return 1; }
      if ((mesh.pgrid[d] > 1 || mesh.self_partition[d]) && nn[d] < 2) { S->err = "box mesh: need >= 2 cells per partitioned direction"; return 1; }
    }
    if (np != S->nproc) { S->err = "box mesh: process grid does not match nproc"; return 1; }
    if (mesh.rank_of(mesh.pcoord[0], mesh.pcoord[1], dims == 3 ? mesh.pcoord[2] : 0) != S->rank) { S->err = "box mesh: process coordinates do not match rank"; return 1; }
    S->num_cells_global = mesh.ne() * np;
  }

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
  // global element numbers (src/geometry.cpp: ele2global_ele): x-fastest over the GLOBAL box
  E->ele2global_ele.setup(E->n_eles);
  {
    const int G[3] = {nx * mesh.pgrid[0], ny * mesh.pgrid[1], nz * (dims == 3 ? mesh.pgrid[2] : 1)};
    for (int k = 0; k < nz; k++)
      for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++)
          E->ele2global_ele(i + nx * (j + ny * k)) = (i + mesh.pcoord[0] * nx) + G[0] * ((j + mesh.pcoord[1] * ny) + G[1] * (k + (dims == 3 ? mesh.pcoord[2] : 0) * nz));
  }
  // reference length for CFL time stepping = the shortest edge of the (linear) element
  // (eles_hexas::calc_h_ref_specific src/eles_hexas.cpp:1551-1571, eles_quads:: src/eles_quads.cpp:1287-1301)
  {
    static const int hex_edge[12][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}, {4, 5}, {5, 7}, {7, 6}, {6, 4}, {1, 5}, {3, 7}, {0, 4}, {2, 6}};
    const int n_edges = (dims == 3) ? 12 : 4;
    for (int e = 0; e < E->n_eles; e++)
    {
      double hmin = 0.0;
      for (int q = 0; q < n_edges; q++)
      {
        double l2 = 0.0;
        for (int d = 0; d < dims; d++) l2 += std::pow(E->shape(d, hex_edge[q][0], e) - E->shape(d, hex_edge[q][1], e), 2.0);
        const double l = std::sqrt(l2);
        if (q == 0 || l < hmin) hmin = l;
      }
      E->h_ref(e) = hmin;
    }
  }

  // interior faces.  Same numbering and left/right orientation as the reference derives from the
  // mesh (src/mesh.cpp set_face_connectivity + src/geometry.cpp:351-415): cells ascending, local
  // faces ascending, a face is created by the first (= lower-numbered) cell that meets it, which
  // becomes its LEFT side.  The orientation matters: the LDG switch reads the left normal only
  // (src/inters.cpp:568-581), and on meshes whose face normals carry rounding noise in n_x the
  // outcome depends on which side is left.
  S->mesh_int_inters.setup(S->n_int_inter_types);
  S->mesh_mpi_inters.setup(S->n_mpi_inter_types);
  const int ftype = (dims == 3) ? 2 : 0;
  const int nn[3] = {nx, ny, nz};
  // is the neighbour through face f of local cell (c) beyond a non-periodic end of the GLOBAL box?
  auto beyond_domain = [&](const int *fd, const int *c) {
    const int d = fd[0];
    if (periodic[d]) return false;
    const int gc = c[d] + mesh.pcoord[d] * nn[d] + fd[1];
    return gc < 0 || gc >= nn[d] * mesh.pgrid[d];
  };
  const double tol = 1e-8 * mesh.length;

  // pass 1: count.  A face whose neighbour cell lives on another rank is a partition face
  // (src/mesh.cpp match_mpifaces / src/geometry.cpp:566-663 build the same lists from ParMETIS output).
  struct mpi_face { int nbr, key_e, key_f, e, f; };
  auto split_dir = [&](int d) { return mesh.pgrid[d] > 1 || mesh.self_partition[d]; };
  struct bdy_face { int e, f, g; };
  std::vector<mpi_face> mf;
  std::vector<bdy_face> bf;
  int n_int = 0;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
      {
        const int e = i + nx * (j + ny * k);
        for (int f = 0; f < nfaces_loc; f++)
        {
          const int *fd = (dims == 3) ? hex_face[f] : quad_face[f];
          const int d = fd[0];
          int c[3] = {i, j, k};
          if (beyond_domain(fd, c))
          {
            // boundary faces in the order the cells meet them (src/geometry.cpp:617-663)
            bdy_face b = {e, f, side_group(f)};
            bf.push_back(b);
            continue;
          }
          c[d] += fd[1];
          const bool outside = c[d] < 0 || c[d] >= nn[d];
          c[d] = (c[d] + nn[d]) % nn[d];
          const int er = c[0] + nx * (c[1] + ny * c[2]);
          if (outside && split_dir(d))
          {
            int pc[3] = {mesh.pcoord[0], mesh.pcoord[1], dims == 3 ? mesh.pcoord[2] : 0};
            pc[d] = (pc[d] + fd[1] + mesh.pgrid[d]) % mesh.pgrid[d];
            const int nbr = mesh.rank_of(pc[0], pc[1], pc[2]);
            // both sides list the faces they share in the order of the lower rank's (cell, local face); the faces a rank
            // shares with itself (self_partition) are grouped by local face and ordered by cell, so that the group of a
            // face and the group of its mate list the pairs in the same order
            mpi_face m = {nbr, (S->rank < nbr) ? e : er, (S->rank < nbr) ? f : fd[2], e, f};
            if (nbr == S->rank) { m.key_e = f; m.key_f = e; }
            mf.push_back(m);
          }
          else if (er >= e)
            n_int++;
        }
      }
  std::sort(mf.begin(), mf.end(), [](const mpi_face &a, const mpi_face &b) {
    if (a.nbr != b.nbr) return a.nbr < b.nbr;
    if (a.key_e != b.key_e) return a.key_e < b.key_e;
    return a.key_f < b.key_f;
  });
  S->n_mpi_inters = (int)mf.size();

  for (int t = 0; t < 3; t++) S->mesh_int_inters(t).setup(t == ftype ? n_int : 0, t, &in);
  S->mesh_bdy_inters.setup(S->n_bdy_inter_types);
  for (int t = 0; t < 3; t++) S->mesh_bdy_inters(t).setup(t == ftype ? (int)bf.size() : 0, t, &in);
  for (size_t m = 0; m < bf.size(); m++) S->mesh_bdy_inters(ftype).set_boundary((int)m, bf[m].g, etype, bf[m].e, bf[m].f, S);
  if (S->mesh_bdy_inters(ftype).failed()) { S->err = S->mesh_bdy_inters(ftype).last_error(); return 1; }
  for (int t = 0; t < 3; t++)
  {
    S->mesh_mpi_inters(t).setup(t == ftype ? S->n_mpi_inters : 0, t, &in);
    S->mesh_mpi_inters(t).set_nproc(S->nproc, S->rank);
  }
  int_inters &I = S->mesh_int_inters(ftype);
  // interior faces.  Same numbering and left/right orientation as the reference derives from the
  // mesh (src/mesh.cpp set_face_connectivity + src/geometry.cpp:351-415): cells ascending, local
  // faces ascending, a face is created by the first (= lower-numbered) cell that meets it, which
  // becomes its LEFT side.  The orientation matters: the LDG switch reads the left normal only
  // (src/inters.cpp:568-581), and on meshes whose face normals carry rounding noise in n_x the
  // outcome depends on which side is left.
  int fi = 0;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
      {
        const int e = i + nx * (j + ny * k);
        for (int f = 0; f < nfaces_loc; f++)
        {
          const int *fd = (dims == 3) ? hex_face[f] : quad_face[f];
          const int d = fd[0];
          int c[3] = {i, j, k};
          if (beyond_domain(fd, c)) continue; // boundary face
          c[d] += fd[1];
          if ((c[d] < 0 || c[d] >= nn[d]) && split_dir(d)) continue; // partition face
          c[d] = (c[d] + nn[d]) % nn[d];
          const int er = c[0] + nx * (c[1] + ny * c[2]);
          if (er < e) continue; // created from the other side already
          const int rot = find_rot_tag(I, E, E, e, er, f, fd[2], dims, mesh.length, tol);
          if (rot < 0) { S->err = "Can't find coupled cyclic interface"; return 1; } /* src/geometry.cpp:442 */
          I.set_interior(fi++, etype, etype, e, er, f, fd[2], rot, S);
        }
      }
  if (fi != n_int) { S->err = "box mesh: face count mismatch"; return 1; }
  if (I.failed()) { S->err = I.last_error(); return 1; }

  // partition faces: the rotation tag of a (local face, remote face) pair is topological on the
  // structured box; read it off an adjacent pair of local cells with the same two local faces
  if (S->n_mpi_inters)
  {
    mpi_inters &M = S->mesh_mpi_inters(ftype);
    int rot_of_face[6];
    for (int f = 0; f < nfaces_loc; f++)
    {
      const int *fd = (dims == 3) ? hex_face[f] : quad_face[f];
      const int d = fd[0];
      int c[3] = {0, 0, 0};
      c[d] = (fd[1] > 0) ? 0 : nn[d] - 1;
      const int ea = c[0] + nx * (c[1] + ny * c[2]);
      c[d] += fd[1];
      const int eb = c[0] + nx * (c[1] + ny * c[2]);
      rot_of_face[f] = find_rot_tag(I, E, E, ea, eb, f, fd[2], dims, mesh.length, tol);
      if (rot_of_face[f] < 0) { S->err = "Can't find coupled cyclic interface"; return 1; }
    }
    std::vector<int> nout(S->nproc, 0);
    for (size_t m = 0; m < mf.size(); m++)
    {
      M.set_mpi((int)m, etype, mf[m].e, mf[m].f, rot_of_face[mf[m].f], S);
      nout[mf[m].nbr]++;
    }
    for (int p = 0; p < S->nproc; p++) M.set_nout_proc(nout[p], p);
    // neighbour segments: one per neighbour rank; the faces shared with this rank itself are one segment per local face,
    // received where the mate's group sits
    for (size_t m = 0; m < mf.size();)
    {
      size_t q = m;
      if (mf[m].nbr != S->rank)
      {
        while (q < mf.size() && mf[q].nbr == mf[m].nbr) q++;
        M.add_segment(mf[m].nbr, (int)m, (int)m, (int)(q - m));
      }
      else
      {
        while (q < mf.size() && mf[q].nbr == S->rank && mf[q].f == mf[m].f) q++;
        const int mate = ((dims == 3) ? hex_face[mf[m].f] : quad_face[mf[m].f])[2];
        size_t r = 0;
        while (r < mf.size() && !(mf[r].nbr == S->rank && mf[r].f == mate)) r++;
        if (r == mf.size()) { S->err = "box mesh: a self-partition face group has no mate"; return 1; }
        M.add_segment(S->rank, (int)m, (int)r, (int)(q - m));
      }
      m = q;
    }
    if (M.failed()) { S->err = M.last_error(); return 1; }
  }
  // wall distance of the damped Smagorinsky closure (src/geometry.cpp:735-892): the flux points of every no-slip wall face, in the
  // mesh's face order, then eles::calc_wall_distance.  The reference gathers the wall points of all ranks; this box mesh does not
  if (in.LES && in.SGS_model == 0)
  {
    bdy_inters &B = S->mesh_bdy_inters(ftype);
    std::vector<double> wall;
    for (int i = 0; i < B.get_n_inters(); i++)
    {
      const int flag = in.bc_list[B.boundary_id(i)].flag;
      if (flag != HFX_BC_ISOTHERM_WALL && flag != HFX_BC_ADIABAT_WALL) continue;
      for (int j = 0; j < B.n_fpts_per_inter; j++)
      {
        const int off = B.disu_fpts_l(j, i), fpt = off % E->n_fpts_per_ele, ele = off / E->n_fpts_per_ele;
        for (int n = 0; n < dims; n++) wall.push_back(E->pos_fpts(fpt, ele, n));
      }
    }
    if (S->nproc > 1 && mesh.pgrid[0] * mesh.pgrid[1] * (dims == 3 ? mesh.pgrid[2] : 1) > 1)
    {
      S->err = "Smagorinsky closure on a partitioned box: the wall points of the other ranks are not gathered (use WALE, SGS_model 1)";
      return 1;
    }
    if (!wall.empty()) E->calc_wall_distance(wall);
  }
  return 0;
}

// output::write_restart_ascii (src/output.cpp:1753-1818): "Rest_%09d_p%04d.dat" in `dir`
int write_restart_ascii(solution *S, const std::string &dir, int in_file_num)
{
  char name[64];
  snprintf(name, sizeof name, "Rest_%.09d_p%.04d.dat", in_file_num, S->rank);
  std::ofstream f((dir.empty() ? std::string(".") : dir) + "/" + name);
  if (!f) { S->err = "Unable to open restart file for writing"; return 1; }
  f.precision(15);
  f << S->time << std::endl;
  for (int i = 0; i < S->n_ele_types; i++)
    if (S->mesh_eles(i) && S->mesh_eles(i)->get_n_eles() != 0)
    {
      S->mesh_eles(i)->write_restart_info_ascii(f);
      S->mesh_eles(i)->write_restart_data_ascii(f);
    }
  return f ? 0 : 1;
}

// read_restart_ascii (src/solver.cpp:377-434): time + every element class's data from the rank files
int read_restart_ascii(solution *S, const std::string &dir, int in_file_num, int in_n_files)
{
  for (int i = 0; i < S->n_ele_types; i++)
    if (S->mesh_eles(i) && S->mesh_eles(i)->get_n_eles() != 0)
    {
      bool info = false;
      for (int j = 0; j < in_n_files && !info; j++)
      {
        char name[64];
        snprintf(name, sizeof name, "Rest_%.09d_p%.04d.dat", in_file_num, j);
        std::ifstream f((dir.empty() ? std::string(".") : dir) + "/" + name);
        if (!f) { S->err = "Could not open restart file "; S->err += name; return 1; } /* src/solver.cpp:393 */
        f >> S->time;
        info = S->mesh_eles(i)->read_restart_info_ascii(f) != 0;
      }
      if (!info) { S->err = "Could not find restart info in the restart files"; return 1; }
      for (int j = 0; j < in_n_files; j++)
      {
        char name[64];
        snprintf(name, sizeof name, "Rest_%.09d_p%.04d.dat", in_file_num, j);
        std::ifstream f((dir.empty() ? std::string(".") : dir) + "/" + name);
        if (!f) { S->err = "Could not open restart file "; S->err += name; return 1; }
        if (S->mesh_eles(i)->read_restart_data_ascii(f)) { S->err = S->mesh_eles(i)->last_error(); return 1; }
      }
    }
  S->run_input.time = S->time;
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
  if (S->run_input.dt_type != 0 && hfx_ctx_set_CFL(S->ctx, S->run_input.CFL)) { S->err = hfx_last_error(); return 1; }
  if (hfx_ctx_set_option(S->ctx, "deferred", S->deferred ? 1 : 0)) { S->err = hfx_last_error(); return 1; }
  for (int i = 0; i < S->n_ele_types; i++)
    if (S->mesh_eles(i) && S->mesh_eles(i)->get_n_eles() != 0)
      if (S->mesh_eles(i)->mv_all_cpu_gpu(S->ctx)) { S->err = S->mesh_eles(i)->last_error(); return 1; }
  for (int i = 0; i < S->n_int_inter_types; i++)
    if (S->mesh_int_inters(i).mv_all_cpu_gpu(S->ctx, S)) { S->err = S->mesh_int_inters(i).last_error(); return 1; }
  for (int i = 0; i < S->n_bdy_inter_types; i++)
    if (S->mesh_bdy_inters(i).mv_all_cpu_gpu(S->ctx, S)) { S->err = S->mesh_bdy_inters(i).last_error(); return 1; }
  for (int i = 0; i < S->n_mpi_inter_types; i++)
    if (S->mesh_mpi_inters(i).mv_all_cpu_gpu(S->ctx, S)) { S->err = S->mesh_mpi_inters(i).last_error(); return 1; }
  return 0;
}

int SetDeferred(solution *S, bool on)
{
  S->deferred = on;
  if (S->ctx && hfx_ctx_set_option(S->ctx, "deferred", on ? 1 : 0)) { S->err = hfx_last_error(); return 1; }
  return 0;
}

// ---- the stage scheduler: same call order as the reference --------------------------------
void CalcResidual(int /*in_file_num*/, int in_rk_stage, solution *FlowSol)
{
  int i;
  // 0: closures that filter the solution do so once per time step, at its first stage (src/solver.cpp:55-62)
  if (FlowSol->run_input.LES && FlowSol->run_input.SGS_model >= 2 && in_rk_stage == 0)
    for (i = 0; i < FlowSol->n_ele_types; i++)
      if (FlowSol->mesh_eles(i)) FlowSol->mesh_eles(i)->calc_sgs_terms();
  // the reference asks `nproc > 1`; a rank that is its own neighbour (self_partition) has partition faces on one rank
  const bool mpi = FlowSol->nproc > 1 || FlowSol->n_mpi_inters > 0;
  auto each_ele = [&](void (eles::*m)()) {
    for (i = 0; i < FlowSol->n_ele_types; i++)
      if (FlowSol->mesh_eles(i)) (FlowSol->mesh_eles(i)->*m)();
  };
  // 1: flux-point values of the state (every class)
  each_ele(&eles::extrapolate_solution);
  // 2: the packed flux-point solution leaves for the neighbour ranks while the element-local work below runs
  if (mpi)
    for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).send_solution();
  if (FlowSol->run_input.viscous)
    // 3: reference-space gradient before the interface correction
    each_ele(&eles::calculate_gradient);
  // 4: inviscid part of the transformed flux (de-aliased through the cubature points when over_int is set)
  if (FlowSol->run_input.over_int) /* src/solver.cpp:82-91 */
    each_ele(&eles::evaluate_invFlux_over_int);
  else
    each_ele(&eles::evaluate_invFlux);
  // 5-7: Riemann flux and LDG common solution on interior, boundary and -- once their data has arrived -- partition faces
  for (i = 0; i < FlowSol->n_int_inter_types; i++) FlowSol->mesh_int_inters(i).calculate_common_invFlux();
  for (i = 0; i < FlowSol->n_bdy_inter_types; i++)
    FlowSol->mesh_bdy_inters(i).evaluate_boundaryConditions_invFlux(FlowSol, FlowSol->time);
  if (mpi)
  {
    for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).receive_solution();
    for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).calculate_common_invFlux();
  }
  if (FlowSol->run_input.viscous)
  {
    // 8: interface correction of the gradient, extrapolation to the flux points, transform to physical space
    each_ele(&eles::correct_gradient);
    // 9: the corrected gradients leave for the neighbour ranks
    if (mpi)
      for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).send_corrected_gradient();
    // 10: viscous part added to the transformed flux
    each_ele(&eles::evaluate_viscFlux);
    // LES: the SGS flux at the flux points, back in physical space (src/solver.cpp:162-167)
    if (FlowSol->run_input.LES)
    {
      each_ele(&eles::extrapolate_sgsFlux);
      // the physical SGS flux leaves for the neighbour ranks (src/solver.cpp:168-178)
      if (mpi)
        for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).send_sgsf_fpts();
    }
  }
  // 11: normal component of the discontinuous flux at the flux points
  each_ele(&eles::extrapolate_totalFlux);
  // 12: divergence of the discontinuous flux
  each_ele(&eles::calculate_divergence);
  if (FlowSol->run_input.viscous)
  {
    // 13-15: LDG viscous common flux added on interior, boundary and partition faces
    for (i = 0; i < FlowSol->n_int_inter_types; i++) FlowSol->mesh_int_inters(i).calculate_common_viscFlux();
    for (i = 0; i < FlowSol->n_bdy_inter_types; i++) FlowSol->mesh_bdy_inters(i).evaluate_boundaryConditions_viscFlux(FlowSol->time);
    if (mpi)
    {
      for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).receive_corrected_gradient();
      if (FlowSol->run_input.LES) /* src/solver.cpp:203-206 */
        for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).receive_sgsf_fpts();
      for (i = 0; i < FlowSol->n_mpi_inter_types; i++) FlowSol->mesh_mpi_inters(i).calculate_common_viscFlux();
    }
  }
  // 16: correction by the flux jump: the divergence of the continuous flux
  each_ele(&eles::calculate_corrected_divergence);
}

// calc_time_step (src/solver.cpp:484-549): per-element CFL steps on the device, minimum over blocks and ranks
int calc_time_step(solution *FlowSol)
{
  input &in = FlowSol->run_input;
  if (in.dt_type != 1 && in.dt_type != 2) return 0;
  double dt_min = 1e12;
  for (int j = 0; j < FlowSol->n_ele_types; j++)
    if (FlowSol->mesh_eles(j) && FlowSol->mesh_eles(j)->get_n_eles() != 0)
    {
      double v = 0.0;
      if (hfx_eles_calc_dt_local(FlowSol->mesh_eles(j)->device(), in.CFL, &v)) { FlowSol->err = hfx_last_error(); return 1; }
      if (v < dt_min) dt_min = v;
    }
  if (FlowSol->nproc > 1)
  {
    // MPI_Allreduce(MIN) over the ranks for dt_type 1 (src/solver.cpp:511) AND 2 (src/solver.cpp:540-547: run_input.dt, which
    // advances `time`, is the global minimum of the local steps): without it every rank would go on with its own minimum
    if (FlowSol->comm)
    {
      if (hfx_comm_allreduce(FlowSol->comm, &dt_min, 1, 0)) { FlowSol->err = hfx_last_error(); return 1; }
    }
    else if (FlowSol->reduce_min)
      dt_min = FlowSol->reduce_min(FlowSol->reduce_user, dt_min);
    else
    {
      FlowSol->err = "calc_time_step: dt_type 1 / 2 on more than one rank needs a MIN reduction over the ranks (SetComm or SetReduceMin)";
      return 1;
    }
  }
  in.dt = dt_min;
  hfx_params p;
  in.fill(p);
  if (hfx_ctx_set_params(FlowSol->ctx, &p)) { FlowSol->err = hfx_last_error(); return 1; }
  return 0;
}

// `if (run_input.pressure_ramp) run_input.ramp_counter++` after every time step (src/HiFiLES.cpp:224-225), passed on to
// the boundary blocks on the device
static int advance_ramp(solution *FlowSol)
{
  input &in = FlowSol->run_input;
  if (!in.pressure_ramp) return 0;
  in.ramp_counter++;
  for (int j = 0; j < FlowSol->n_bdy_inter_types; j++)
    if (FlowSol->mesh_bdy_inters(j).get_n_inters() && FlowSol->mesh_bdy_inters(j).device())
      if (hfx_bdy_inters_set_ramp_counter(FlowSol->mesh_bdy_inters(j).device(), in.ramp_counter)) { FlowSol->err = hfx_last_error(); return 1; }
  return 0;
}

int RunSteps(solution *FlowSol, int n_steps)
{
  const int RKSteps = FlowSol->run_input.n_rk_stages();
  for (int i_steps = 0; i_steps < n_steps; i_steps++)
  {
    if (calc_time_step(FlowSol)) return 1; /* src/HiFiLES.cpp:198 */
    for (int i = 0; i < RKSteps; i++)
    {
      CalcResidual(FlowSol->ini_iter + i_steps, i, FlowSol);
      for (int j = 0; j < FlowSol->n_ele_types; j++)
        if (FlowSol->mesh_eles(j)) FlowSol->mesh_eles(j)->AdvanceSolution(i, FlowSol->run_input.adv_type);
      if (FlowSol->run_input.shock_cap) /* src/HiFiLES.cpp:214-216 */
        for (int j = 0; j < FlowSol->n_ele_types; j++)
          if (FlowSol->mesh_eles(j)) FlowSol->mesh_eles(j)->shock_capture();
    }
    FlowSol->time += FlowSol->run_input.dt;
    FlowSol->run_input.time = FlowSol->time;
    if (advance_ramp(FlowSol)) return 1;
  }
  // (deferred execution: the last stage's record is still pending here, on purpose -- what the caller asks for next decides how
  // it runs: the state alone -> the fused stage; the gradient arrays, as the reference's CopyGPUCPU does before its
  // diagnostics -> call by call.  An error in it is reported by that next call.)
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
  for (int j = 0; j < FlowSol->n_mpi_inter_types; j++)
    if (FlowSol->mesh_mpi_inters(j).failed())
    {
      FlowSol->err = FlowSol->mesh_mpi_inters(j).last_error();
      return 1;
    }
  for (int j = 0; j < FlowSol->n_bdy_inter_types; j++)
    if (FlowSol->mesh_bdy_inters(j).failed())
    {
      FlowSol->err = FlowSol->mesh_bdy_inters(j).last_error();
      return 1;
    }
  return 0;
}

void SetExchange(solution *FlowSol, hfxh_exchange_fn fn, void *user)
{
  FlowSol->exchange = fn;
  FlowSol->exchange_user = user;
  for (int j = 0; j < FlowSol->n_mpi_inter_types; j++) FlowSol->mesh_mpi_inters(j).set_exchange(fn, user);
}

void SetReduceMin(solution *FlowSol, double (*fn)(void *user, double v), void *user)
{
  FlowSol->reduce_min = fn;
  FlowSol->reduce_user = user;
}

int SetComm(solution *FlowSol, const char *unique_id)
{
  if (!FlowSol->ctx) { FlowSol->err = "SetComm: the case is not on the device"; return 1; }
  if (FlowSol->comm) { hfx_comm_destroy(FlowSol->comm); FlowSol->comm = nullptr; }
  if (!unique_id) // back to the caller's hooks
  {
    for (int j = 0; j < FlowSol->n_mpi_inter_types; j++) FlowSol->mesh_mpi_inters(j).set_comm(nullptr);
    return 0;
  }
  if (hfx_comm_create(FlowSol->ctx, unique_id, FlowSol->nproc, FlowSol->rank, &FlowSol->comm)) { FlowSol->err = hfx_last_error(); return 1; }
  for (int j = 0; j < FlowSol->n_mpi_inter_types; j++)
    if (FlowSol->mesh_mpi_inters(j).set_comm(FlowSol->comm)) { FlowSol->err = FlowSol->mesh_mpi_inters(j).last_error(); return 1; }
  return 0;
}

// the blocks of a partitioned run through the split fused kernels: one tensor-product class, its interior and boundary
// blocks (boundary blocks ride with the interior ones), its partition-face blocks
static int partitioned_blocks(solution *FlowSol, eles *&E, std::vector<hfx_inters *> &fi, std::vector<hfx_inters *> &fm)
{
  E = nullptr;
  for (int j = 0; j < FlowSol->n_ele_types; j++)
    if (FlowSol->mesh_eles(j) && FlowSol->mesh_eles(j)->get_n_eles() != 0)
    {
      if (E) { FlowSol->err = "RunStepsPartitionedFused: one element class only"; return 1; }
      E = FlowSol->mesh_eles(j);
    }
  if (!E) { FlowSol->err = "RunStepsPartitionedFused: no elements"; return 1; }
  for (int j = 0; j < FlowSol->n_int_inter_types; j++)
    if (FlowSol->mesh_int_inters(j).get_n_inters()) fi.push_back(FlowSol->mesh_int_inters(j).device());
  for (int j = 0; j < FlowSol->n_bdy_inter_types; j++)
    if (FlowSol->mesh_bdy_inters(j).get_n_inters()) fi.push_back(FlowSol->mesh_bdy_inters(j).device());
  for (int j = 0; j < FlowSol->n_mpi_inter_types; j++)
    if (FlowSol->mesh_mpi_inters(j).get_n_inters()) fm.push_back(FlowSol->mesh_mpi_inters(j).device());
  return 0;
}

int TimePartitioned(solution *FlowSol, int reps, double ms[8])
{
  eles *E;
  std::vector<hfx_inters *> fi, fm;
  if (partitioned_blocks(FlowSol, E, fi, fm)) return 1;
  if (!FlowSol->comm) { FlowSol->err = "TimePartitioned: needs the library's communicator (SetComm)"; return 1; }
  if (hfx_time_partitioned(E->device(), fi.data(), (int)fi.size(), fm.data(), (int)fm.size(), FlowSol->comm, reps, ms))
  {
    FlowSol->err = hfx_last_error();
    return 1;
  }
  return 0;
}

int CommInfo(solution *FlowSol, int *nranks, int *rank, int *device, char pci_bus_id[32])
{
  if (!FlowSol->comm) { FlowSol->err = "CommInfo: needs the library's communicator (SetComm)"; return 1; }
  if (hfx_comm_info(FlowSol->comm, nranks, rank, device, pci_bus_id)) { FlowSol->err = hfx_last_error(); return 1; }
  return 0;
}

int RunStepsPartitionedFused(solution *FlowSol, int n_steps)
{
  eles *E;
  std::vector<hfx_inters *> fi, fm;
  if (partitioned_blocks(FlowSol, E, fi, fm)) return 1;
  input &in = FlowSol->run_input;
  if (FlowSol->comm)
  {
    // the whole loop inside the library: phases, RCCL exchanges on its communication stream, calc_time_step with the
    // all-reduce, ramp counters.  With CFL steps the host follows step by step to keep `time` (the library recomputes dt).
    const int chunk = (in.dt_type == 0) ? n_steps : 1;
    for (int done = 0; done < n_steps; done += chunk)
    {
      if (hfx_run_steps_partitioned(E->device(), fi.data(), (int)fi.size(), fm.data(), (int)fm.size(), FlowSol->comm, chunk))
      {
        FlowSol->err = hfx_last_error();
        return 1;
      }
      if (in.dt_type != 0 && hfx_ctx_get_dt(FlowSol->ctx, &in.dt)) { FlowSol->err = hfx_last_error(); return 1; }
      FlowSol->time += chunk * in.dt;
      if (in.pressure_ramp) in.ramp_counter += chunk; // the library advanced its boundary blocks' counters
    }
    in.time = FlowSol->time;
    return 0;
  }
  const int RKSteps = in.n_rk_stages();
  const bool ex = !fm.empty() && FlowSol->exchange;
  auto phase = [&](int ph, int stage, int first) {
    return hfx_stage_partitioned(E->device(), fi.data(), (int)fi.size(), fm.data(), (int)fm.size(), ph, stage, first);
  };
  auto xch = [&](int kind, int ph) {
    if (ex) FlowSol->exchange(FlowSol->exchange_user, kind, ph);
  };
  const bool visc = in.viscous != 0;
  bool first = true;
  for (int i_steps = 0; i_steps < n_steps; i_steps++)
  {
    if (calc_time_step(FlowSol)) return 1; /* src/HiFiLES.cpp:198 */
    for (int i = 0; i < RKSteps; i++)
    {
      if (first)
      {
        if (phase(0, i, 1)) { FlowSol->err = hfx_last_error(); return 1; }
        xch(0, 0); // later stages: started after phase 4 of the previous one
        first = false;
      }
      if (phase(1, i, 0)) { FlowSol->err = hfx_last_error(); return 1; }
      xch(0, 1);
      if (phase(2, i, 0)) { FlowSol->err = hfx_last_error(); return 1; }
      if (visc) xch(1, 0);
      if (visc && in.LES) xch(2, 0); // third message: the SGS flux
      if (phase(3, i, 0)) { FlowSol->err = hfx_last_error(); return 1; }
      if (visc) xch(1, 1);
      if (visc && in.LES) xch(2, 1);
      if (phase(4, i, 0)) { FlowSol->err = hfx_last_error(); return 1; }
      xch(0, 0);
    }
    FlowSol->time += in.dt;
    in.time = FlowSol->time;
    if (advance_ramp(FlowSol)) return 1;
  }
  // the exchange started after the last stage belongs to a stage that is not run: complete it so that
  // no request is left in flight (a following call starts over with `first`)
  xch(0, 1);
  return 0;
}
