/* oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's per-RK-stage hot path
 * (SURVEY.md section 8a rows a2-a13), same data layout (hf_array column-major,
 * include/hf_array.h:303-325) and same operation order as the reference's
 * BLAS=NO CPU branch.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libhfx) never does.
 *
 * Parity status: PINNED -- checked against fixtures captured from the genuine
 * reference compiled in the build container (oracle/_ref, tests/golden,
 * tests/test_oracle_vs_golden.py).
 */
#ifndef HFX_ORACLE_H
#define HFX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* frozen scalars the path reads from the reference's global `run_input`
 * (include/input.h; set in src/input.cpp:138-187,596-614) */
typedef struct orc_params
{
  double gamma, prandtl, rt_inf, mu_inf, c_sth, fix_vis;
  double ldg_beta, ldg_tau;
  double dt;
  int viscous;
  int riemann_solve_type;     /* 0 Rusanov, 2 RoeM, 3 HLLC (src/int_inters.cpp:185-205) */
  int vis_riemann_solve_type; /* 0 LDG */
  int adv_type;               /* 0 Euler, 1 RK24, 2 RK34, 3 RK45, 4 RK414 (src/eles.cpp:1080) */
  int dt_type;                /* 0 fixed, 2 local (dt_local array) */
  int n_rk;                   /* length of RK_a / RK_b */
  double RK_a[16], RK_b[16];  /* data/RK_coeff.dat */
} orc_params;

/* one element class = the arrays of reference class `eles` (include/eles.h:659-899) */
typedef struct orc_eles
{
  int n_eles, n_upts, n_fpts, n_fields, n_dims;
  /* operators, dense column-major */
  const double *opp_0;    /* (n_fpts,n_upts) */
  const double *opp_1[3]; /* (n_fpts,n_upts) */
  const double *opp_2[3]; /* (n_upts,n_upts) */
  const double *opp_3;    /* (n_upts,n_fpts) */
  const double *opp_4[3]; /* (n_upts,n_upts) */
  const double *opp_5[3]; /* (n_upts,n_fpts) */
  const double *opp_6;    /* (n_fpts,n_upts) */
  /* metrics */
  const double *detjac_upts; /* (n_upts,n_eles) */
  const double *JGinv_upts;  /* (n_dims,n_dims,n_upts,n_eles) */
  const double *detjac_fpts; /* (n_fpts,n_eles) */
  const double *JGinv_fpts;  /* (n_dims,n_dims,n_fpts,n_eles) */
  const double *tdA_fpts;    /* (n_fpts,n_eles) */
  const double *norm_fpts;   /* (n_fpts,n_eles,n_dims) */
  /* state and work arrays */
  double *disu_upts[2];    /* (n_upts,n_eles,n_fields) */
  double *disu_fpts;       /* (n_fpts,n_eles,n_fields) */
  double *tdisf_upts;      /* (n_upts,n_eles,n_fields,n_dims) */
  double *norm_tdisf_fpts; /* (n_fpts,n_eles,n_fields) */
  double *norm_tconf_fpts; /* (n_fpts,n_eles,n_fields) */
  double *div_tconf_upts;  /* (n_upts,n_eles,n_fields) */
  double *delta_disu_fpts; /* (n_fpts,n_eles,n_fields) */
  double *grad_disu_upts;  /* (n_upts,n_eles,n_fields,n_dims) */
  double *grad_disu_fpts;  /* (n_fpts,n_eles,n_fields,n_dims) */
  const double *src_upts;  /* (n_upts,n_eles,n_fields) or NULL (= 0) */
  const double *dt_local;  /* (n_eles) or NULL */
  /* LES eddy-viscosity closure (run_input.LES): sgs_model < 0 = off; 0 Smagorinsky (wall_distance), 1 WALE */
  int sgs_model;
  double C_s, filter_ratio, Kappa, prandtl_t;
  int order;                   /* run_input.order (filter width) */
  double les_vol_factor;       /* calc_ele_vol / detjac of the class: hexes 8, quads 4, prisms 4, tetrahedra 8/6, triangles 2 (src/eles_*.cpp calc_ele_vol) */
  const double *wall_distance; /* (n_upts,n_eles,n_dims), model 0 */
  double *sgsf_upts;           /* (n_upts,n_eles,n_fields,n_dims) */
  double *sgsf_fpts;           /* (n_fpts,n_eles,n_fields,n_dims) */
  const double *Jacobian_fpts; /* (n_dims,n_dims,n_fpts,n_eles): extrapolate_sgsFlux takes the flux back to physical space */
  /* over-integration (run_input.over_int): n_cub 0 = off */
  int n_cub;
  const double *opp_over_int_cubpts;   /* (n_cub,n_upts) */
  const double *over_int_filter;       /* (n_upts,n_cub) */
  const double *JGinv_over_int_cubpts; /* (n_dims,n_dims,n_cub,n_eles) */
  /* similarity-type closures (sgs_model 2 WALE + similarity, 3 SVV, 4 similarity; src/eles.cpp:138-165) */
  const double *filter_upts; /* (n_upts,n_upts) */
  double *disuf_upts;        /* (n_upts,n_eles,n_fields) filtered solution */
  double *uu, *Lu;           /* (n_upts,n_eles,3|6) velocity products, Leonard tensor */
  double *ue, *Le;           /* (n_upts,n_eles,n_dims) velocity-energy products, Leonard vector */
} orc_eles;

/* interior faces (reference class int_inters): the hf_array<double*> tables of
 * include/inters.h:86-116 as offsets into the field-0 (fpt,ele) plane */
typedef struct orc_int_inters
{
  int n_inters, n_fpts_per_inter;
  const int *L; /* (n_fpts_per_inter,n_inters): fpt + n_fpts*ele of the left side  */
  const int *R; /* same for the right side, permutation `lut` already applied */
} orc_int_inters;

/* partition faces (reference class mpi_inters, src/mpi_inters.cpp): left side local, right side in the
 * received buffer at slot Rlut(j) = lut(j) of record (fpt, field[, dim], inter) (set_mpi :165-172).
 * PARITY NOTE: the reference's MPI build cannot be compiled in the build container (no ParMETIS), so
 * these four functions are pinned through partition invariance (N-rank run == 1-rank run of the pinned
 * single-rank oracle, tests/test_partition_gloo.py), not against reference output. */
typedef struct orc_mpi_inters
{
  int n_inters, n_fpts_per_inter;
  const int *L;    /* (n_fpts_per_inter,n_inters) left offsets */
  const int *Rlut; /* (n_fpts_per_inter,n_inters) flux-point slot in the received face record */
  double *out_disu, *in_disu; /* (n_fpts_per_inter,n_fields,n_inters) */
  double *out_grad, *in_grad; /* (n_fpts_per_inter,n_fields,n_dims,n_inters) */
  double *out_sgsf, *in_sgsf; /* LES: the physical SGS flux at the flux points, same layout (src/mpi_inters.cpp:65-66); NULL: off */
} orc_mpi_inters;

/* boundary faces (reference class bdy_inters, src/bdy_inters.cpp).  One record per entry of
 * run_input.bc_list (include/bc.h:48-62), values AFTER input::read_boundary_param's
 * non-dimensionalisation (src/input.cpp:440-525). */
enum
{
  ORC_SUB_IN_SIMP = 0, ORC_SUB_OUT_SIMP = 1, ORC_SUB_IN_CHAR = 2, ORC_SUB_OUT_CHAR = 3, ORC_SUP_IN = 4,
  ORC_SUP_OUT = 5, ORC_SLIP_WALL = 6, ORC_CYCLIC = 7, ORC_ISOTHERM_WALL = 8, ORC_ADIABAT_WALL = 9, ORC_CHAR = 10,
  ORC_SLIP_WALL_DUAL = 11 /* src/bc.cpp:36-48 */
};
typedef struct orc_bc
{
  int flag, pressure_ramp, use_wm, pad;
  double rho, velocity[3], p_static, T_static, p_total, T_total, nx, ny, nz;
  double p_ramp_coeff, T_ramp_coeff, p_total_old, T_total_old;
} orc_bc;
typedef struct orc_bdy_inters
{
  int n_inters, n_fpts_per_inter;
  const int *L;           /* (n_fpts_per_inter,n_inters) left offsets */
  const int *boundary_id; /* (n_inters) index into bcs */
  const orc_bc *bcs;
  int n_bcs;
  double R_ref;     /* run_input.R_ref (viscous) / run_input.R_gas (inviscid), src/bdy_inters.cpp:368-369 */
  int ramp_counter; /* run_input.ramp_counter */
} orc_bdy_inters;

/* shock capturing (src/eles.cpp:2918-2959, src/eles_hexas.cpp:1007-1059): matrices built by the element class
 * (set_vandermonde / set_exp_filter / calc_norm_basis) */
typedef struct orc_shock
{
  const double *inv_vandermonde;    /* (n_upts,n_upts) */
  const double *exp_filter;         /* (n_upts,n_upts) */
  const double *norm_basis_persson; /* (n_upts) */
  const int *high_modes;            /* (n_upts) 1 where a mode index equals the order */
  double s0;
  int shock_det_field; /* 0 density, 1 total energy */
  double *sensor;      /* (n_eles) out */
} orc_shock;

void orc_set_threads(int n);

/* src/funcs.cpp:49-123 */
void orc_dgemm(int Arows, int Bcols, int Acols, double alpha, double beta,
               const double *a, const double *b, double *c);

/* point physics: src/flux.cpp:33,74,129,257 ; src/inters.cpp:277,327,439,561,615 */
void orc_calc_invf(int n_dims, double gamma, const double *u, double *f);
void orc_calc_visf(int n_dims, const orc_params *p, const double *u, const double *grad_u, double *f);
void orc_rusanov_flux(int n_dims, double gamma, const double *ul, const double *ur, const double *fl,
                      const double *fr, const double *norm, double *fn);
void orc_roeM_flux(int n_dims, double gamma, const double *ul, const double *ur, const double *fl,
                   const double *fr, const double *norm, double *fn);
void orc_hllc_flux(int n_dims, double gamma, const double *ul, const double *ur, const double *fl,
                   const double *fr, const double *norm, double *fn);
void orc_ldg_flux(int flux_spec, int n_dims, const double *ul, const double *ur, const double *fl,
                  const double *fr, const double *norm, double *fn, double ldg_tau, double ldg_beta);
void orc_ldg_solution(int flux_spec, int n_dims, const double *ul, const double *ur, double *uc,
                      double ldg_beta, const double *norm);

/* element methods: src/eles.cpp */
void orc_extrapolate_solution(orc_eles *e);                            /* :1360 */
void orc_calculate_gradient(orc_eles *e);                              /* :1823 */
void orc_evaluate_invFlux(orc_eles *e, const orc_params *p);           /* :1415 */
void orc_correct_gradient(orc_eles *e);                                /* :1890 */
void orc_evaluate_viscFlux(orc_eles *e, const orc_params *p);          /* :2285 */
void orc_extrapolate_totalFlux(orc_eles *e);                           /* :1549 */
void orc_calculate_divergence(orc_eles *e);                            /* :1651 */
long orc_calculate_corrected_divergence(orc_eles *e);                  /* :1738 ; returns index of first NaN or -1 */
void orc_AdvanceSolution(orc_eles *e, const orc_params *p, int in_step); /* :1080 */
double orc_calc_dt_local(const orc_eles *e, const orc_params *p, int ele, double h_ref, double CFL, int order); /* :1267 */
/* src/eles.cpp:5045 ; norm_type 0 max, 1 L1 sum, 2 L2 sum */
double orc_compute_res_upts(const orc_eles *e, int norm_type, int field);
/* eles::calc_disu_ppts for every element (src/eles.cpp:3757): out (n_ppts, n_eles, n_fields) */
void orc_calc_disu_ppts(const orc_eles *e, int n_ppts, const double *opp_p, double *out);

/* face methods: src/int_inters.cpp */
void orc_int_calculate_common_invFlux(const orc_int_inters *f, orc_eles *e, const orc_params *p);  /* :160 */
void orc_int_calculate_common_viscFlux(const orc_int_inters *f, orc_eles *e, const orc_params *p); /* :254 */
/* the same with the two sides of the block in different element classes (mixed meshes, src/geometry.cpp:637-706) */
void orc_int_calculate_common_invFlux_lr(const orc_int_inters *f, orc_eles *el, orc_eles *er, const orc_params *p);
void orc_int_calculate_common_viscFlux_lr(const orc_int_inters *f, orc_eles *el, orc_eles *er, const orc_params *p);

/* partition faces: src/mpi_inters.cpp */
/* eles::evaluate_invFlux_over_int (src/eles.cpp:1480-1545): opp (n_cub,n_upts), filter (n_upts,n_cub), JGinv (nd,nd,n_cub,n_eles) */
void orc_evaluate_invFlux_over_int(orc_eles *e, const orc_params *p, int n_cub, const double *opp_over_int_cubpts,
                                   const double *over_int_filter, const double *JGinv_over_int_cubpts);
/* eles::CalcIntegralQuantities (src/eles.cpp:5485-5627); ids: 0 kineticenergy 1 enstropy 2 pressuredilatation
 * 3 straincolonproduct 4 devstraincolonproduct; out[m] += contribution of this block; uses grad_disu_upts */
void orc_CalcIntegralQuantities(const orc_eles *e, const orc_params *p, int n_cub, const double *opp_volume_cubpts,
                                const double *weight_volume_cubpts, const double *vol_detjac_vol_cubpts, int n_q,
                                const int *ids, double *out);
/* LES: eles::calc_sgsf_upts (src/eles.cpp:2395-2650, eddy-viscosity models 0 and 1) and extrapolate_sgsFlux (:2817) */
void orc_calc_sgsf_upts(const orc_eles *e, const orc_params *p, const double *u, const double *grad_u, double detjac, int ele,
                        int upt, double *sgsf);
void orc_extrapolate_sgsFlux(orc_eles *e);
long orc_calc_sgs_terms(orc_eles *e); /* eles::calc_sgs_terms, src/eles.cpp:2058 (first RK stage of a step, models 2-4) */
void orc_shock_capture(orc_eles *e, const orc_shock *s); /* eles::shock_capture, shock_cap 1 + shock_det 0 */

/* boundary faces, src/bdy_inters.cpp (wall model, LES inlet, RANS off) */
void orc_set_boundary_conditions(int sol_spec, const orc_bc *bc, int n_dims, int viscous, const double *u_l, double *u_r,
                                 const double *norm, double gamma, double R_ref, int ramp_counter);            /* :340-1019 */
void orc_set_boundary_gradients(const orc_bc *bc, int n_dims, const double *u_r, const double *grad_ul, double *grad_ur,
                                const double *norm);                                                           /* :1138-1189 */
void orc_bdy_evaluate_boundaryConditions_invFlux(const orc_bdy_inters *f, orc_eles *e, const orc_params *p);  /* :213-338 */
void orc_bdy_evaluate_boundaryConditions_viscFlux(const orc_bdy_inters *f, orc_eles *e, const orc_params *p); /* :1024-1136 */

void orc_mpi_pack_solution(const orc_mpi_inters *f, const orc_eles *e);           /* :218-229 */
void orc_mpi_pack_corrected_gradient(const orc_mpi_inters *f, const orc_eles *e); /* :278-289 */
void orc_mpi_pack_sgsf(const orc_mpi_inters *f, const orc_eles *e);               /* :339-351 (LES) */
void orc_mpi_calculate_common_invFlux(const orc_mpi_inters *f, orc_eles *e, const orc_params *p);  /* :400 */
void orc_mpi_calculate_common_viscFlux(const orc_mpi_inters *f, orc_eles *e, const orc_params *p); /* :485 */

/* the caller contract: src/solver.cpp:50-223 (single rank, LES/RANS/forcing off) */
long orc_CalcResidual(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_params *p);
/* the same with boundary-face blocks, in the reference's order (src/solver.cpp:124-129,190-195) */
long orc_CalcResidual_bdy(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_bdy_inters *bdy,
                          int n_bdy_blocks, const orc_params *p);
long orc_rk_step_bdy(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_bdy_inters *bdy,
                     int n_bdy_blocks, const orc_params *p);
/* RK loop body of src/HiFiLES.cpp:201-217 for n_stages consecutive stages starting at stage 0 */
long orc_rk_step(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_params *p);

#ifdef __cplusplus
}
#endif
#endif
