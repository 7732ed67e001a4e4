// solver.hpp -- `struct solution`, the stage scheduler CalcResidual and the box-mesh setup.
//
// struct solution mirrors /root/reference/include/solution.h:44-96; CalcResidual has the
// call order of /root/reference/src/solver.cpp:50-223 (single rank, LES / RANS / forcing off)
// and calls only the public eles / int_inters methods, like the reference's.
#pragma once
#include <string>
#include <vector>

#include "eles.hpp"
#include "inters.hpp"

struct solution
{
  int rank = 0, nproc = 1;
  double time = 0.0;
  int n_ele_types = 5; // 0 tri, 1 quad, 2 tet, 3 pri, 4 hex (src/geometry.cpp:135)
  int n_dims = 0;
  int num_cells_global = 0, ini_iter = 0;
  hf_array<eles *> mesh_eles; // entries of classes this build does not carry are NULL
  eles_quads mesh_eles_quads;
  eles_hexas mesh_eles_hexas;
  int n_int_inter_types = 3;
  hf_array<int_inters> mesh_int_inters;
  int n_bdy_inter_types = 3;
  hf_array<bdy_inters> mesh_bdy_inters; // include/solution.h:84-86
  // partition faces (include/solution.h:88-94, _MPI only in the reference)
  int n_mpi_inter_types = 3;
  hf_array<mpi_inters> mesh_mpi_inters;
  int n_mpi_inters = 0;
  input run_input;
  hfx_ctx *ctx = nullptr;
  hfxh_exchange_fn exchange = nullptr;
  void *exchange_user = nullptr;
  double (*reduce_min)(void *user, double v) = nullptr; // MPI_Allreduce(MIN) of calc_time_step, supplied by the caller
  void *reduce_user = nullptr;
  hfx_comm *comm = nullptr; // the library's RCCL transport (SetComm); takes the place of `exchange` and `reduce_min`
  // deferred execution (include/hfx.h): the method calls of CalcResidual + AdvanceSolution below are recorded by libhfx and
  // a whole stage runs as one fused stage -- the call sequence itself is the reference's, unchanged.  On by default.
  bool deferred = true;
  std::string err;
  ~solution();
};

// periodic box [0,length]^dims of n[0] x n[1] (x n[2]) quads / hexes, optional smooth periodic
// deformation (same formula as oracle/gen_neu_mesh.py) or caller-supplied vertices
// xv[v + nv*d], v = ix + (nx+1)*(iy + (ny+1)*iz).
struct box_mesh
{
  int dims = 3, n[3] = {1, 1, 1}; // cells of THIS rank's block
  double length = 0.0, amp = 0.0; // edge of the GLOBAL periodic box
  // block decomposition of the structured box: pgrid ranks per direction, this rank's coordinates.
  // The reference partitions with ParMETIS (src/mesh.cpp:72-314), which is not available; the
  // structured block split is the documented stand-in.
  int pgrid[3] = {1, 1, 1}, pcoord[3] = {0, 0, 0};
  int rank_of(int px, int py, int pz) const { return px + pgrid[0] * (py + pgrid[1] * pz); }
  // a periodic direction that is NOT split (pgrid 1) whose wrap-around faces are nevertheless made partition faces, with
  // this rank as its own neighbour: the whole partition-face path (pack, exchange, one-sided kernels) then runs on one
  // rank -- how the RCCL transport is exercised on a one-GPU box
  int self_partition[3] = {0, 0, 0};
  // boundary group of each side of the GLOBAL box, indexed by the element-local face number (hex: z- y- x+ y+ x-
  // z+, quad: y- x+ y+ x-): index into run_input.bc_specs; -1 or a cyclic group = periodic.  A direction is
  // periodic when both of its sides are.
  int side_bc[6] = {-1, -1, -1, -1, -1, -1};
  std::vector<double> xv; // (nv, dims) column-major
  int nv() const { return (n[0] + 1) * (n[1] + 1) * (dims == 3 ? n[2] + 1 : 1); }
  int ne() const { return n[0] * n[1] * (dims == 3 ? n[2] : 1); }
  void generate();
};

// the part of GeoPreprocess + InitSolution the hot path needs (src/geometry.cpp:106-915,
// src/solver.cpp:321): elements, shapes, transforms, interior (periodic) faces, initial state
int GeoPreprocess_box(solution *FlowSol, const box_mesh &mesh);
int InitSolution(solution *FlowSol);
// move everything to the device (the reference's mv_all_cpu_gpu calls, src/geometry.cpp:310-320,552-557)
int MoveToDevice(solution *FlowSol, int device);
// switch deferred execution on / off (before or after MoveToDevice)
int SetDeferred(solution *FlowSol, bool on);

void CalcResidual(int in_file_num, int in_rk_stage, solution *FlowSol);
// RK loop of src/HiFiLES.cpp:194-221 through the mirrored class methods
int calc_time_step(solution *FlowSol); // src/solver.cpp:484-549
// ASCII restart files "Rest_%09d_p%04d.dat" (host arrays; sync with the device before / after)
int write_restart_ascii(solution *FlowSol, const std::string &dir, int in_file_num);             // src/output.cpp:1753
int read_restart_ascii(solution *FlowSol, const std::string &dir, int in_file_num, int n_files); // src/solver.cpp:377
int RunSteps(solution *FlowSol, int n_steps);
// the same RK loop through the split fused kernels on a partitioned block, exchanging between the
// phases of hfx_stage_partitioned
int RunStepsPartitionedFused(solution *FlowSol, int n_steps);
void SetExchange(solution *FlowSol, hfxh_exchange_fn fn, void *user);
void SetReduceMin(solution *FlowSol, double (*fn)(void *user, double v), void *user);
// collective: the library's communicator for this rank (id from hfx_comm_get_unique_id on rank 0)
int SetComm(solution *FlowSol, const char *unique_id);
// per-phase / per-exchange times of the partitioned fused stage (hfx_time_partitioned)
int TimePartitioned(solution *FlowSol, int reps, double ms[8]);
// what RCCL reports for the communicator (hfx_comm_info)
int CommInfo(solution *FlowSol, int *nranks, int *rank, int *device, char pci_bus_id[32]);
