#include "inters.hpp"

#include "solver.hpp"

int_inters::~int_inters()
{
  if (dev) hfx_inters_destroy(dev);
}

void int_inters::setup(int in_n_inters, int in_inter_type, input *in)
{
  n_inters = in_n_inters;
  inters_type = in_inter_type;
  order = in->order;
  viscous = in->viscous;
  if (inters_type == 0)
  {
    n_fpts_per_inter = order + 1;
    n_dims = 2;
  }
  else if (inters_type == 1)
  {
    n_fpts_per_inter = (order + 2) * (order + 1) / 2;
    n_dims = 3;
  }
  else
  {
    n_fpts_per_inter = (order + 1) * (order + 1);
    n_dims = 3;
  }
  n_fields = n_dims + 2;
  disu_fpts_l.setup(n_fpts_per_inter, n_inters);
  disu_fpts_r.setup(n_fpts_per_inter, n_inters);
  lut.setup(n_fpts_per_inter);
}

void int_inters::get_lut(int rot)
{
  const int N = order + 1;
  if (inters_type == 0)
  {
    for (int i = 0; i < n_fpts_per_inter; i++) lut(i) = n_fpts_per_inter - i - 1;
  }
  else if (inters_type == 2)
  {
    // i = slow, j = fast index of the left face's (k,j) frame
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++)
      {
        int v;
        if (rot == 0) v = (N - 1 - j) + N * i;
        else if (rot == 1) v = n_fpts_per_inter - (N - 1 - j) - N * i - 1;
        else if (rot == 2) v = N * j + i;
        else v = n_fpts_per_inter - N * j - i - 1;
        lut(i * N + j) = v;
      }
  }
  else
  {
    if (err.empty()) err = "ERROR: Invalid interface type ... (triangular faces are not built yet)";
  }
}

void int_inters::set_interior(int in_inter, int in_ele_type_l, int in_ele_type_r, int in_ele_l, int in_ele_r,
                              int in_local_inter_l, int in_local_inter_r, int rot_tag, struct solution *FlowSol)
{
  if (ele_type_l < 0)
  {
    ele_type_l = in_ele_type_l;
    ele_type_r = in_ele_type_r;
  }
  if (ele_type_l != in_ele_type_l || ele_type_r != in_ele_type_r)
  {
    if (err.empty()) err = "int_inters: one face block must connect one pair of element classes";
    return;
  }
  get_lut(rot_tag);
  eles *el = FlowSol->mesh_eles(in_ele_type_l), *er = FlowSol->mesh_eles(in_ele_type_r);
  for (int j = 0; j < n_fpts_per_inter; j++)
  {
    const int j_rhs = lut(j);
    disu_fpts_l(j, in_inter) = el->get_fpt_offset(in_ele_l, in_local_inter_l, j);
    disu_fpts_r(j, in_inter) = er->get_fpt_offset(in_ele_r, in_local_inter_r, j_rhs);
  }
}

int int_inters::mv_all_cpu_gpu(hfx_ctx *ctx, struct solution *FlowSol)
{
  if (n_inters == 0) return 0;
  eles *el = FlowSol->mesh_eles(ele_type_l), *er = FlowSol->mesh_eles(ele_type_r);
  if (hfx_int_inters_create(ctx, el->device(), er->device(), n_inters, n_fpts_per_inter, disu_fpts_l.get_ptr_cpu(),
                            disu_fpts_r.get_ptr_cpu(), &dev))
  {
    err = hfx_last_error();
    return 1;
  }
  return 0;
}

void int_inters::calculate_common_invFlux()
{
  if (n_inters != 0 && hfx_int_inters_calculate_common_invFlux(dev) && err.empty()) err = hfx_last_error();
}

void int_inters::calculate_common_viscFlux()
{
  if (n_inters != 0 && hfx_int_inters_calculate_common_viscFlux(dev) && err.empty()) err = hfx_last_error();
}

// ---- mpi_inters -----------------------------------------------------------------------------
mpi_inters::~mpi_inters()
{
  if (dev) hfx_inters_destroy(dev);
}

void mpi_inters::setup(int in_n_inters, int in_inter_type, input *in)
{
  n_inters = in_n_inters;
  inters_type = in_inter_type;
  order = in->order;
  viscous = in->viscous;
  if (inters_type == 0) { n_fpts_per_inter = order + 1; n_dims = 2; }
  else if (inters_type == 1) { n_fpts_per_inter = (order + 2) * (order + 1) / 2; n_dims = 3; }
  else { n_fpts_per_inter = (order + 1) * (order + 1); n_dims = 3; }
  n_fields = n_dims + 2;
  disu_fpts_l.setup(n_fpts_per_inter, n_inters);
  disu_fpts_r.setup(n_fpts_per_inter, n_inters);
  lut.setup(n_fpts_per_inter);
}

void mpi_inters::set_nproc(int in_nproc, int in_rank)
{
  nproc = in_nproc;
  rank = in_rank;
  Nout_proc.setup(nproc);
  Nout_proc.initialize_to_zero();
}

void mpi_inters::set_nout_proc(int in_nout, int in_p) { Nout_proc(in_p) = in_nout; }

void mpi_inters::set_mpi(int in_inter, int in_ele_type_l, int in_ele_l, int in_local_inter_l, int rot_tag,
                         struct solution *FlowSol)
{
  if (ele_type_l < 0) ele_type_l = in_ele_type_l;
  if (ele_type_l != in_ele_type_l)
  {
    if (err.empty()) err = "mpi_inters: one partition-face block must belong to one element class";
    return;
  }
  // the same look-up table as interior faces (src/inters.cpp:153-262)
  int_inters tmp;
  tmp.inters_type = inters_type;
  tmp.order = order;
  tmp.n_fpts_per_inter = n_fpts_per_inter;
  tmp.lut.setup(n_fpts_per_inter);
  tmp.get_lut(rot_tag);
  eles *el = FlowSol->mesh_eles(in_ele_type_l);
  for (int j = 0; j < n_fpts_per_inter; j++)
  {
    disu_fpts_l(j, in_inter) = el->get_fpt_offset(in_ele_l, in_local_inter_l, j);
    disu_fpts_r(j, in_inter) = tmp.lut(j); // in_buffer_disu.get_ptr(j_rhs, field, in_inter)
  }
}

int mpi_inters::mv_all_cpu_gpu(hfx_ctx *ctx, struct solution *FlowSol)
{
  if (n_inters == 0) return 0;
  eles *el = FlowSol->mesh_eles(ele_type_l);
  if (hfx_mpi_inters_create(ctx, el->device(), n_inters, n_fpts_per_inter, disu_fpts_l.get_ptr_cpu(),
                            disu_fpts_r.get_ptr_cpu(), &dev))
  {
    err = hfx_last_error();
    return 1;
  }
  return 0;
}

#define HFX_MPI_CALL(expr)                                            \
  do                                                                  \
  {                                                                   \
    if (n_inters != 0 && (expr) != 0 && err.empty()) err = hfx_last_error(); \
  } while (0)

void mpi_inters::add_segment(int peer, int send_first, int recv_first, int count)
{
  seg_peer.push_back(peer); seg_send.push_back(send_first); seg_recv.push_back(recv_first); seg_count.push_back(count);
}

// the reference's layout: the faces shared with rank p are contiguous, ranks ascending (src/mpi_inters.cpp:244-256)
void mpi_inters::set_segments_from_nout()
{
  seg_peer.clear(); seg_send.clear(); seg_recv.clear(); seg_count.clear();
  int start = 0;
  for (int p = 0; p < nproc; p++)
    if (Nout_proc(p))
    {
      add_segment(p, start, start, Nout_proc(p));
      start += Nout_proc(p);
    }
}

int mpi_inters::set_comm(hfx_comm *c)
{
  comm = c;
  if (n_inters == 0 || !c) return 0;
  if (hfx_mpi_inters_set_neighbours(dev, (int)seg_peer.size(), seg_peer.data(), seg_send.data(), seg_recv.data(), seg_count.data()))
  {
    err = hfx_last_error();
    return 1;
  }
  return 0;
}

// with the library's transport, send_* = pack + grouped ncclSend / ncclRecv on the communication stream and receive_* = the
// compute stream's wait for them; with the hook the caller moves the packed buffers
void mpi_inters::send_solution()
{
  if (comm) { HFX_MPI_CALL(hfx_mpi_inters_send_solution(dev, comm)); return; }
  HFX_MPI_CALL(hfx_mpi_inters_pack_solution(dev));
  if (n_inters != 0 && exchange) exchange(exchange_user, 0, 0);
}
void mpi_inters::receive_solution()
{
  if (comm) { HFX_MPI_CALL(hfx_mpi_inters_receive_solution(dev, comm)); return; }
  if (n_inters != 0 && exchange) exchange(exchange_user, 0, 1);
}
void mpi_inters::send_corrected_gradient()
{
  if (comm) { HFX_MPI_CALL(hfx_mpi_inters_send_corrected_gradient(dev, comm)); return; }
  HFX_MPI_CALL(hfx_mpi_inters_pack_corrected_gradient(dev));
  if (n_inters != 0 && exchange) exchange(exchange_user, 1, 0);
}
void mpi_inters::receive_corrected_gradient()
{
  if (comm) { HFX_MPI_CALL(hfx_mpi_inters_receive_corrected_gradient(dev, comm)); return; }
  if (n_inters != 0 && exchange) exchange(exchange_user, 1, 1);
}
void mpi_inters::send_sgsf_fpts()
{
  if (comm) { HFX_MPI_CALL(hfx_mpi_inters_send_sgsf_fpts(dev, comm)); return; }
  HFX_MPI_CALL(hfx_mpi_inters_pack_sgsf(dev));
  if (n_inters != 0 && exchange) exchange(exchange_user, 2, 0);
}
void mpi_inters::receive_sgsf_fpts()
{
  if (comm) { HFX_MPI_CALL(hfx_mpi_inters_receive_sgsf_fpts(dev, comm)); return; }
  if (n_inters != 0 && exchange) exchange(exchange_user, 2, 1);
}
void mpi_inters::calculate_common_invFlux() { HFX_MPI_CALL(hfx_mpi_inters_calculate_common_invFlux(dev)); }
void mpi_inters::calculate_common_viscFlux() { HFX_MPI_CALL(hfx_mpi_inters_calculate_common_viscFlux(dev)); }

// ---- bdy_inters -----------------------------------------------------------------------------
bdy_inters::~bdy_inters()
{
  if (dev) hfx_inters_destroy(dev);
}

void bdy_inters::setup(int in_n_inters, int in_inter_type, input *in)
{
  n_inters = in_n_inters;
  inters_type = in_inter_type;
  run_input = in;
  order = in->order;
  viscous = in->viscous;
  if (inters_type == 0) { n_fpts_per_inter = order + 1; n_dims = 2; }
  else if (inters_type == 1) { n_fpts_per_inter = (order + 2) * (order + 1) / 2; n_dims = 3; }
  else { n_fpts_per_inter = (order + 1) * (order + 1); n_dims = 3; }
  n_fields = n_dims + 2;
  boundary_id.setup(n_inters);
  disu_fpts_l.setup(n_fpts_per_inter, n_inters);
}

void bdy_inters::set_boundary(int in_inter, int bc_id, int in_ele_type_l, int in_ele_l, int in_local_inter_l,
                              struct solution *FlowSol)
{
  boundary_id(in_inter) = bc_id;
  if (in_ele_type < 0) in_ele_type = in_ele_type_l;
  if (in_ele_type != in_ele_type_l)
  {
    if (err.empty()) err = "bdy_inters: one boundary-face block must belong to one element class";
    return;
  }
  eles *el = FlowSol->mesh_eles(in_ele_type_l);
  for (int j = 0; j < n_fpts_per_inter; j++) disu_fpts_l(j, in_inter) = el->get_fpt_offset(in_ele_l, in_local_inter_l, j);
}

int bdy_inters::mv_all_cpu_gpu(hfx_ctx *ctx, struct solution *FlowSol)
{
  if (n_inters == 0) return 0;
  eles *el = FlowSol->mesh_eles(in_ele_type);
  if (hfx_bdy_inters_create(ctx, el->device(), n_inters, n_fpts_per_inter, disu_fpts_l.get_ptr_cpu(),
                            boundary_id.get_ptr_cpu(), run_input->bc_list.data(), (int)run_input->bc_list.size(),
                            run_input->bc_R_ref(), &dev))
  {
    err = hfx_last_error();
    return 1;
  }
  if (hfx_bdy_inters_set_ramp_counter(dev, run_input->ramp_counter))
  {
    err = hfx_last_error();
    return 1;
  }
  return 0;
}

void bdy_inters::evaluate_boundaryConditions_invFlux(struct solution * /*FlowSol*/, double time_bound)
{
  if (n_inters != 0 && hfx_bdy_inters_evaluate_boundaryConditions_invFlux(dev, time_bound) != 0 && err.empty())
    err = hfx_last_error();
}

void bdy_inters::evaluate_boundaryConditions_viscFlux(double time_bound)
{
  if (n_inters != 0 && hfx_bdy_inters_evaluate_boundaryConditions_viscFlux(dev, time_bound) != 0 && err.empty())
    err = hfx_last_error();
}
