// inters.hpp -- host-side mirror of the reference's interior-face class.
//
// /root/reference/include/inters.h:40-126 + int_inters.h:40-75: the reference keeps
// hf_array<double*> tables of host pointers into the element arrays; a device library
// cannot take host pointers, so the same tables are kept as OFFSETS `fpt + n_fpts*ele`
// into the (fpt,ele) plane of the owning element block -- the registration form of
// hfx_int_inters_create (include/hfx.h).
#pragma once
#include <string>
#include <vector>

#include "eles.hpp"

struct solution;

class int_inters
{
public:
  ~int_inters();
  // inters::setup_inters + int_inters::setup (src/inters.cpp:60-150, src/int_inters.cpp:48-65)
  // in_inter_type: 0 segment, 1 triangle, 2 quadrilateral
  void setup(int in_n_inters, int in_inter_type, input *in_run_input);
  // int_inters::set_interior (src/int_inters.cpp:67-121): same arguments
  void set_interior(int in_inter, int in_ele_type_l, int in_ele_type_r, int in_ele_l, int in_ele_r, int in_local_inter_l,
                    int in_local_inter_r, int rot_tag, struct solution *FlowSol);
  // look-up table of the flux-point permutation for a rotation tag (src/inters.cpp:153-262)
  void get_lut(int in_rot_tag);
  int mv_all_cpu_gpu(hfx_ctx *ctx, struct solution *FlowSol);
  void calculate_common_invFlux();  // src/int_inters.cpp:160
  void calculate_common_viscFlux(); // src/int_inters.cpp:254

  int get_n_inters() const { return n_inters; }
  hfx_inters *device() { return dev; }
  const std::string &last_error() const { return err; }
  bool failed() const { return !err.empty(); }

  int inters_type = 0, order = 0, viscous = 0, n_inters = 0, n_fpts_per_inter = 0, n_fields = 0, n_dims = 0;
  int ele_type_l = -1, ele_type_r = -1; // one face block connects one pair of element classes
  hf_array<int> disu_fpts_l, disu_fpts_r; // (n_fpts_per_inter, n_inters) offsets; the other tables share them
  hf_array<int> lut;

private:
  hfx_inters *dev = nullptr;
  std::string err;
};

// exchange hook: the reference calls MPI_Isend/Irecv + MPI_Waitall inside send_* / receive_*
// (src/mpi_inters.cpp:244-270); here the transport is pluggable (RCCL through torch.distributed in
// the bench, gloo in the CPU tests).  kind: 0 solution, 1 corrected gradient, 2 SGS flux (LES); phase: 0 start, 1 wait.
typedef void (*hfxh_exchange_fn)(void *user, int kind, int phase);

// Host-side mirror of the reference's partition-face class (include/mpi_inters.h:40-105).
class mpi_inters
{
public:
  ~mpi_inters();
  void setup(int in_n_inters, int in_inter_type, input *in_run_input); // src/mpi_inters.cpp:50
  void set_nproc(int in_nproc, int in_rank);                           // :91
  void set_nout_proc(int in_nout, int in_p);                           // :101
  // mpi_inters::set_mpi (src/mpi_inters.cpp:148-216): same arguments
  void set_mpi(int in_inter, int in_ele_type_l, int in_ele_l, int in_local_inter_l, int rot_tag, struct solution *FlowSol);
  int mv_all_cpu_gpu(hfx_ctx *ctx, struct solution *FlowSol);
  void send_solution();              // :218 pack + start of the exchange
  void receive_solution();           // :261 wait
  void send_corrected_gradient();    // :278
  void receive_corrected_gradient(); // :323
  void send_sgsf_fpts();             // :339 (LES)
  void receive_sgsf_fpts();          // :384
  void calculate_common_invFlux();   // :400
  void calculate_common_viscFlux();  // :485
  void set_exchange(hfxh_exchange_fn fn, void *user) { exchange = fn; exchange_user = user; }
  // the library's own transport (RCCL send / recv on its communication stream, include/hfx.h): takes the place of the hook
  int set_comm(hfx_comm *c);
  // neighbour segments (hfx_mpi_inters_set_neighbours): derived from Nout_proc by set_segments_from_nout, or given
  void add_segment(int peer, int send_first, int recv_first, int count);
  void set_segments_from_nout();

  int get_n_inters() const { return n_inters; }
  hfx_inters *device() { return dev; }
  const std::string &last_error() const { return err; }
  bool failed() const { return !err.empty(); }

  int inters_type = 0, order = 0, viscous = 0, n_inters = 0, n_fpts_per_inter = 0, n_fields = 0, n_dims = 0;
  int nproc = 1, rank = 0, ele_type_l = -1;
  hf_array<int> Nout_proc;                // faces exchanged with each rank; a rank's faces are contiguous
  std::vector<int> seg_peer, seg_send, seg_recv, seg_count;
  hf_array<int> disu_fpts_l, disu_fpts_r; // left offsets ; received-record slot lut(j)
  hf_array<int> lut;

private:
  hfx_inters *dev = nullptr;
  hfx_comm *comm = nullptr;
  hfxh_exchange_fn exchange = nullptr;
  void *exchange_user = nullptr;
  std::string err;
};

// Host-side mirror of the reference's boundary-face class (include/bdy_inters.h:31-105).
class bdy_inters
{
public:
  ~bdy_inters();
  void setup(int in_n_inters, int in_inter_type, input *in_run_input); // src/bdy_inters.cpp:62
  // bdy_inters::set_boundary (src/bdy_inters.cpp:75-135): same arguments
  void set_boundary(int in_inter, int bc_id, int in_ele_type_l, int in_ele_l, int in_local_inter_l, struct solution *FlowSol);
  int mv_all_cpu_gpu(hfx_ctx *ctx, struct solution *FlowSol);
  void evaluate_boundaryConditions_invFlux(struct solution *FlowSol, double time_bound); // :213
  void evaluate_boundaryConditions_viscFlux(double time_bound);                            // :1024

  int get_n_inters() const { return n_inters; }
  hfx_inters *device() { return dev; }
  const std::string &last_error() const { return err; }
  bool failed() const { return !err.empty(); }

  int inters_type = 0, order = 0, viscous = 0, n_inters = 0, n_fpts_per_inter = 0, n_fields = 0, n_dims = 0;
  int in_ele_type = -1;
  hf_array<int> boundary_id;  // index into run_input.bc_list
  hf_array<int> disu_fpts_l;  // left offsets
  input *run_input = nullptr;

private:
  hfx_inters *dev = nullptr;
  std::string err;
};
