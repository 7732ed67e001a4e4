/* hfx_host.h -- extern "C" facade of libhfx_host: the host-side mirror of the reference's
 * eles / int_inters / solver interface (C++ classes under hifiles-solver_amd/csrc/host, same
 * method names as /root/reference/include/eles.h:55-125, int_inters.h:48-60, solver.h:34),
 * built on the C ABI of libhfx (hfx.h).  The facade exists for test / bench plumbing written
 * in Python; a C++ caller uses the classes directly.
 *
 * A "case" is a periodic box of quads (dims 2) or hexes (dims 3) -- the synthetic meshes of
 * BASELINE.json's configurations -- with its elements, metrics, interior faces and initial state.
 */
#ifndef HFX_HOST_H
#define HFX_HOST_H

#include "hfx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfxh_case hfxh_case;

/* one boundary group: `bc_<name>_type` (as hfx_bc_flag) and its parameters as the reference's input file
 * gives them, DIMENSIONAL (/root/reference/src/input.cpp:328-437); T_total / T_total_old < 0: not given
 * (default T_free_stream) */
typedef struct hfxh_bc_desc
{
  int flag, pressure_ramp;
  double rho, u, v, w, p_static, T_static, p_total, T_total, nx, ny, nz, mach;
  double p_ramp_coeff, T_ramp_coeff, p_total_old, T_total_old;
} hfxh_bc_desc;

/* inputs are DIMENSIONAL, exactly the keys of the reference's input file
 * (/root/reference/src/input.cpp:62-327); they are non-dimensionalised as input::setup_params does */
typedef struct hfxh_case_desc
{
  int dims;         /* 2 quads, 3 hexes */
  int n[3];         /* cells per direction (>= 3) */
  int order;        /* `order` */
  double length;    /* box edge = dx_cyclic = dy_cyclic = dz_cyclic */
  double amp;       /* amplitude of the smooth periodic vertex deformation (0: uniform) */
  const double *xv; /* optional vertices (nv,dims) column-major, v = ix + (nx+1)*(iy + (ny+1)*iz); NULL: generate */
  const double *loc_1d_upts; /* optional order+1 solution-point abscissae (e.g. the reference's data/JacobiGQ.bin
                              * row, which is not bit-symmetric); NULL: computed Gauss / Lobatto nodes */
  int viscous, riemann_solve_type, adv_type, ic_form;
  int upts_type;   /* upts_type_hexa / upts_type_quad: 0 Gauss, 1 Gauss-Lobatto */
  int vcjh_scheme; /* vcjh_scheme_hexa / _quad: 0 eta given, 1 DG, 2 SD, 3 Huynh, 4 c+ */
  double eta;
  int fix_vis;
  double dt, ldg_beta, ldg_tau;
  double gamma, prandtl, S_gas, T_gas, R_gas, mu_gas;
  double Mach_free_stream, rho_free_stream, L_free_stream, T_free_stream;
  double rho_c_ic, Mach_c_ic, T_c_ic;     /* viscous initial condition */
  double u_c_ic, v_c_ic, w_c_ic, p_c_ic;  /* inviscid initial condition */
  /* block decomposition (all 0: one rank).  n[] is THIS rank's block, the global box has n[d]*pgrid[d]
   * cells per direction and edge `length`; rank = px + pgrid[0]*(py + pgrid[1]*pz).  Stands in for the
   * reference's ParMETIS partition (src/mesh.cpp:72-314). */
  int rank, nproc, pgrid[3];
  /* boundary groups (n_bcs 0: fully periodic).  side_bc[f], f = element-local face number of the side of the
   * GLOBAL box (hexes: z- y- x+ y+ x- z+; quads: y- x+ y+ x-), is an index into bcs; a group of type
   * HFX_BC_CYCLIC (or index -1) makes the side periodic.  Mirrors the mesh file's boundary groups + the input
   * file's bc_<name>_* keys. */
  int n_bcs;
  const hfxh_bc_desc *bcs;
  int side_bc[6];
  /* time step: dt_type 0 fixed `dt`, 1 global CFL minimum, 2 local CFL steps (src/input.cpp:141-158) */
  int dt_type;
  double CFL;
  /* polynomial de-aliasing and shock capturing, the reference's keys (src/input.cpp:248-263): over_int,
   * over_int_order; shock_cap (1: exponential modal filter), shock_det_field (0 density, 1 total energy), s0,
   * expf_fac, expf_order, expf_cutoff.  The matrices (opp_over_int_cubpts, over_int_filter, JGinv_over_int_cubpts,
   * inv_vandermonde, exp_filter, norm_basis_persson) are built by the host mirror (csrc/host/eles_modal.cpp). */
  int over_int, over_int_order;
  int shock_cap, shock_det_field;
  double s0, expf_fac;
  int expf_order, expf_cutoff;
  /* LES closure (src/input.cpp:167-182): LES 1, SGS_model 1 WALE / 2 WALE + similarity / 3 SVV / 4 similarity, C_s,
   * filter_ratio, prandtl_t (0: 0.9); filter_type at the end of this struct */
  int LES, SGS_model;
  double C_s, filter_ratio, prandtl_t;
  int p_res; /* plot points per edge (`p_res`, src/input.cpp:110); 0: the reference's default 2 */
  /* self_partition[d] != 0: the wrap-around faces of the periodic direction d (which must not be split, pgrid[d] 1)
   * become partition faces whose neighbour is this rank itself.  The complete partition-face path -- pack, exchange,
   * one-sided kernels -- then runs on ONE rank: how the RCCL transport is exercised on a one-GPU box, and how
   * bench.py prices the partitioned stage on one GPU (--self-partition). */
  int self_partition[3];
  /* LES closures 2 (WALE + similarity), 3 (spectral vanishing viscosity), 4 (similarity) filter the solution:
   * filter_type 0 high-order-commuting Vasilyev, 1 discrete Gaussian, 2 modal, other: average (src/input.cpp:173,
   * src/eles_hexas.cpp:583-790, src/eles_quads.cpp:428-622; the mirror builds filter_upts in csrc/host/eles_modal.cpp) */
  int filter_type;
} hfxh_case_desc;

const char *hfxh_last_error(void);

/* host only: builds operators, metrics, faces, initial state (no GPU needed) */
int hfxh_case_create(const hfxh_case_desc *d, hfxh_case **out);
int hfxh_case_destroy(hfxh_case *c);
/* {n_eles, n_upts, n_fpts, n_fields, n_dims, order, ele_type, n_rk_stages} */
int hfxh_case_sizes(hfxh_case *c, int sizes[8]);
int hfxh_case_params(hfxh_case *c, hfx_params *p);
/* host array by the reference's member name (opp_0, opp_1_0, ..., JGinv_upts, disu_upts0, ...) */
int hfxh_case_get_array(hfxh_case *c, const char *name, const double **ptr, int dims[4]);
int hfxh_case_get_faces(hfxh_case *c, const int **L, const int **R, int *n_fpts_per_inter, int *n_inters);

/* boundary faces (bdy_inters): left offsets L(j,i), boundary_id(i), and the non-dimensional bc_list
 * (what input::read_boundary_param leaves in run_input.bc_list) with R_ref and ramp_counter */
int hfxh_case_get_bdy_faces(hfxh_case *c, const int **L, const int **boundary_id, int *n_fpts_per_inter, int *n_inters);
int hfxh_case_get_bcs(hfxh_case *c, const hfx_bc **bcs, int *n_bcs, double *R_ref, int *ramp_counter);
/* partition faces (mpi_inters): left offsets L(j,i), received-record slots Rlut(j,i), and Nout_proc[nproc]
 * = faces shared with each rank; a rank's faces are contiguous and ordered by rank */
int hfxh_case_get_mpi_faces(hfxh_case *c, const int **L, const int **Rlut, int *n_fpts_per_inter, int *n_inters,
                            const int **nout_proc);
/* exchange hook called by mpi_inters::send_* (phase 0, after packing) and receive_* (phase 1); kind 0
 * solution, 1 corrected gradient.  The transport (RCCL / gloo through torch.distributed) belongs to the caller. */
typedef void (*hfxh_exchange_cb)(void *user, int kind, int phase);
int hfxh_case_set_exchange(hfxh_case *c, hfxh_exchange_cb fn, void *user);

/* neighbour segments of the partition-face block (hfx_mpi_inters_set_neighbours): seg[s] = {peer, send_first, recv_first,
 * count}; for a block without self-partition faces: one segment per neighbour rank, send_first == recv_first */
int hfxh_case_get_mpi_segments(hfxh_case *c, const int **peer, const int **send_first, const int **recv_first, const int **count,
                               int *n_seg);
/* MPI_Allreduce(MIN) hook of calc_time_step (dt_type 1 on more than one rank) when the transport is the caller's */
typedef double (*hfxh_reduce_min_cb)(void *user, double v);
int hfxh_case_set_reduce_min(hfxh_case *c, hfxh_reduce_min_cb fn, void *user);
/* collective over the case's nproc ranks, after hfxh_case_to_device: the library's own transport (RCCL, hfx_comm_*) with the
 * 128-byte id of hfx_comm_get_unique_id (made on rank 0, distributed by the launcher).  From then on send_* / receive_*,
 * hfxh_case_run_partitioned and calc_time_step's MIN reduction go through it instead of the hooks. */
int hfxh_case_set_comm(hfxh_case *c, const char *unique_id);
/* hfx_comm_info of the case's communicator (after hfxh_case_set_comm) */
int hfxh_case_comm_info(hfxh_case *c, int *nranks, int *rank, int *device, char pci_bus_id[32]);
/* hfx_time_partitioned on this case's blocks: ms[0..3] phases 1-4, ms[4] solution exchange, ms[5] flux exchange, ms[6] stage */
int hfxh_case_time_partitioned(hfxh_case *c, int reps, double ms[8]);

/* ---- tetrahedra / prisms as producers of operators and metrics (row a17) ----------------------------------------------
 * eles_tets / eles_pris of the host mirror (csrc/host/eles_simplex.cpp) set up for `order` and the given straight-sided
 * elements: shape (3, n_spts, n_eles) column-major, n_spts 4 (tetrahedra, ele_type 2) or 6 (prisms, ele_type 3), the
 * reference's eles::shape.  Builds loc_upts, tloc_fpts, tnorm_fpts, opp_0 .. opp_6 and the metrics (set_transforms);
 * read them with hfxh_simplex_get_array (names as hfxh_case_get_array).  loc_1d_upts: optional 1-D abscissae of the
 * prism's line direction (NULL: computed Gauss nodes). */
typedef struct hfxh_simplex hfxh_simplex;
int hfxh_simplex_create(int ele_type, int order, int viscous, int n_eles, const double *shape, const double *loc_1d_upts,
                        hfxh_simplex **out);
/* the same with another member of the VCJH family on the tetrahedron / on the prism's triangle: vcjh_scheme_tet / vcjh_scheme_tri
 * 0 (the filter's c given: c_tet / c_tri), 1 DG, 2 SD-like, 3 Huynh-like, 4 c+ (/root/reference/src/eles_tets.cpp:1305-1390,
 * src/funcs.cpp:717-795) */
int hfxh_simplex_create_vcjh(int ele_type, int order, int viscous, int n_eles, const double *shape, const double *loc_1d_upts,
                             int vcjh_scheme, double c, hfxh_simplex **out);
/* the general form: the input keys the simplex classes read beyond order / viscous.
 *  - vcjh_scheme, c: as above;
 *  - SGS_model >= 0: a run with an LES closure (run_input.LES = 1; < 0: none): set_transforms also fills Jacobian_fpts (what
 *    hfx_eles_set_les takes), and for the closures that filter the solution (SGS_model 2, 3, 4) eles_tets builds filter_upts
 *    (/root/reference/src/eles_tets.cpp:576-690: filter_type 2 modal, 3 element average; 0 and 1 stop as the reference does).  The
 *    reference's prism class builds no filter (src/eles_pris.cpp:134): SGS_model >= 2 is refused there;
 *  - shock_cap 1: the operators of shock capturing (what hfx_eles_set_shock_capture takes): inv_vandermonde, exp_filter,
 *    norm_basis_persson, persson_high_modes (read as doubles) -- src/eles_tets.cpp:705-797, src/eles_pris.cpp:609-730.
 * n_spts: shape nodes per element -- 0 for the straight-sided shapes, else 4 or 10 (the quadratic tetrahedron of
 * src/eles_tets.cpp:1047-1069, nodes 4-9 on the edges (0,1) (0,2) (0,3) (1,2) (2,3) (3,1)), 6 or 15 (the quadratic prism of
 * src/eles_pris.cpp:1115-1146); shape is then (3, n_spts, n_eles). */
typedef struct hfxh_simplex_keys
{
  int vcjh_scheme; /* vcjh_scheme_tet / vcjh_scheme_tri */
  double c;        /* c_tet / c_tri (scheme 0) */
  int SGS_model;   /* < 0: no LES */
  int filter_type;
  double filter_ratio;
  int shock_cap;   /* 0 off, 1 exponential filter */
  double expf_fac;
  int expf_order, expf_cutoff;
} hfxh_simplex_keys;
int hfxh_simplex_create_keys(int ele_type, int order, int viscous, int n_eles, int n_spts, const double *shape,
                             const double *loc_1d_upts, const hfxh_simplex_keys *keys, hfxh_simplex **out);
int hfxh_simplex_get_array(hfxh_simplex *s, const char *name, const double **ptr, int dims[4]);
int hfxh_simplex_destroy(hfxh_simplex *s);

/* device */
int hfxh_case_to_device(hfxh_case *c, int device);
/* deferred execution of the mirrored method calls (hfx.h, option "deferred"): ON by default -- hfxh_case_CalcResidual / hfxh_case_run
 * make the reference's calls one by one, libhfx records them and runs every whole stage as one fused stage.  0: every call
 * launches its own kernels (the per-method path).  May be called before or after hfxh_case_to_device. */
int hfxh_case_set_deferred(hfxh_case *c, int on);
int hfxh_case_handles(hfxh_case *c, hfx_ctx **ctx, hfx_eles **e, hfx_inters ***faces, int *n_face_blocks);
int hfxh_case_CalcResidual(hfxh_case *c); /* the mirrored CalcResidual (src/solver.cpp:50-223) */
int hfxh_case_run(hfxh_case *c, int n_steps); /* the mirrored RK loop (src/HiFiLES.cpp:194-221) */
/* device handle of the partition-face block (NULL on one rank) */
int hfxh_case_mpi_handle(hfxh_case *c, hfx_inters **f);
/* the RK loop through hfx_stage_partitioned (split fused kernels), exchanging between its phases */
int hfxh_case_run_partitioned(hfxh_case *c, int n_steps);
/* ASCII restart files of the reference, "Rest_%09d_p%04d.dat" in `dir` (output::write_restart_ascii
 * src/output.cpp:1753-1818, read_restart_ascii src/solver.cpp:377-434): host state in / out; a case that is on
 * the device downloads before writing and uploads after reading */
int hfxh_case_write_restart(hfxh_case *c, const char *dir, int file_num);
int hfxh_case_read_restart(hfxh_case *c, const char *dir, int file_num, int n_files);
/* calc_time_step (src/solver.cpp:484-549) on the device; hfxh_case_run calls it before every step */
int hfxh_case_calc_time_step(hfxh_case *c, double *dt);
/* eles::calc_disu_ppts for every element (device contraction + download): out (n_ppts, n_eles, n_fields) */
int hfxh_case_calc_disu_ppts(hfxh_case *c, const double **out, int dims[3]);
int hfxh_case_sync_host(hfxh_case *c);        /* cp_*_gpu_cpu of state, divergence, gradient */

#ifdef __cplusplus
}
#endif
#endif
