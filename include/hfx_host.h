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

/* device */
int hfxh_case_to_device(hfxh_case *c, int device);
int hfxh_case_handles(hfxh_case *c, hfx_ctx **ctx, hfx_eles **e, hfx_inters ***faces, int *n_face_blocks);
int hfxh_case_CalcResidual(hfxh_case *c); /* the mirrored CalcResidual (src/solver.cpp:50-223) */
int hfxh_case_run(hfxh_case *c, int n_steps); /* the mirrored RK loop (src/HiFiLES.cpp:194-221) */
int hfxh_case_sync_host(hfxh_case *c);        /* cp_*_gpu_cpu of state, divergence, gradient */

#ifdef __cplusplus
}
#endif
#endif
