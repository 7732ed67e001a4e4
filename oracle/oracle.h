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
} orc_mpi_inters;

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

/* face methods: src/int_inters.cpp */
void orc_int_calculate_common_invFlux(const orc_int_inters *f, orc_eles *e, const orc_params *p);  /* :160 */
void orc_int_calculate_common_viscFlux(const orc_int_inters *f, orc_eles *e, const orc_params *p); /* :254 */

/* partition faces: src/mpi_inters.cpp */
void orc_mpi_pack_solution(const orc_mpi_inters *f, const orc_eles *e);           /* :218-229 */
void orc_mpi_pack_corrected_gradient(const orc_mpi_inters *f, const orc_eles *e); /* :278-289 */
void orc_mpi_calculate_common_invFlux(const orc_mpi_inters *f, orc_eles *e, const orc_params *p);  /* :400 */
void orc_mpi_calculate_common_viscFlux(const orc_mpi_inters *f, orc_eles *e, const orc_params *p); /* :485 */

/* the caller contract: src/solver.cpp:50-223 (single rank, LES/RANS/forcing off) */
long orc_CalcResidual(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_params *p);
/* RK loop body of src/HiFiLES.cpp:201-217 for n_stages consecutive stages starting at stage 0 */
long orc_rk_step(orc_eles *e, const orc_int_inters *faces, int n_face_blocks, const orc_params *p);

#ifdef __cplusplus
}
#endif
#endif
