/* hfx.h -- C ABI of libhfx: the MI355X (gfx950) implementation of the HiFiLES
 * per-RK-stage hot path.
 *
 * The reference has no FFI: the seam this library sits behind is the C++ class
 * API of `eles`, `int_inters` (`bdy_inters`, `mpi_inters`) that
 * CalcResidual (/root/reference/src/solver.cpp:50-223) and the RK loop
 * (/root/reference/src/HiFiLES.cpp:201-217) call, plus the legacy `_GPU` seam
 * (include/cuda_kernels.h:31-120, hf_array::cp_cpu_gpu / cp_gpu_cpu
 * include/hf_array.h:541-580).  Every entry point below names the reference
 * method it replaces.  All arrays are `double`, column-major in the hf_array
 * layout (include/hf_array.h:303-325): a(i,j,k,l) = data[i + d0*(j + d1*(k + d2*l))].
 *
 * Conventions
 *  - plain pointers and sizes only; opaque handles own all device state
 *  - every function returns 0 on success, non-zero on error; the message is
 *    available from hfx_last_error() (the reference's FatalError = print + exit,
 *    include/error.h:33-43; the adapter maps non-zero to FatalError)
 *  - not re-entrant per context; one context <-> one device (the reference is
 *    single-threaded per rank, one rank <-> one device, src/geometry.cpp:97)
 *  - the authoritative copy of the state lives on the device between explicit
 *    hfx_eles_download() calls (placed where the legacy code had cp_*_gpu_cpu,
 *    src/eles.cpp:998-1062)
 */
#ifndef HFX_H
#define HFX_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfx_ctx hfx_ctx;
typedef struct hfx_eles hfx_eles;
typedef struct hfx_inters hfx_inters;

/* Frozen scalars the path reads from the reference's global `run_input`
 * (include/input.h; src/input.cpp:138-187,596-614; data/RK_coeff.dat). */
typedef struct hfx_params
{
  double gamma, prandtl, rt_inf, mu_inf, c_sth, fix_vis;
  double ldg_beta, ldg_tau;
  double dt;
  int viscous;
  int riemann_solve_type;     /* 0 Rusanov, 2 RoeM, 3 HLLC (src/int_inters.cpp:185-205) */
  int vis_riemann_solve_type; /* 0 LDG (the only one the reference implements) */
  int adv_type;               /* 0 Euler, 1 RK24, 2 RK34, 3 RK45, 4 RK414 (src/eles.cpp:1080) */
  int dt_type;                /* 0/1 one dt for all elements, 2 per-element dt_local */
  int n_rk;
  double RK_a[16], RK_b[16];
} hfx_params;

/* Registration data of one element class = what a constructed reference
 * `eles` object holds (include/eles.h:659-899).  Host pointers; copied once. */
typedef struct hfx_eles_desc
{
  int n_eles, n_upts, n_fpts, n_fields, n_dims;
  int ele_type; /* 0 tri, 1 quad, 2 tet, 3 pri, 4 hex (include/eles.h "ele_type") */
  int order;
  const double *opp_0;    /* (n_fpts,n_upts)  src/eles.cpp:3074 */
  const double *opp_1[3]; /* (n_fpts,n_upts)  :3145 */
  const double *opp_2[3]; /* (n_upts,n_upts)  :3230 */
  const double *opp_3;    /* (n_upts,n_fpts)  :3320 */
  const double *opp_4[3]; /* (n_upts,n_upts)  :3375 ; NULL when inviscid */
  const double *opp_5[3]; /* (n_upts,n_fpts)  :3455 ; NULL when inviscid */
  const double *opp_6;    /* (n_fpts,n_upts)  :3540 ; NULL when inviscid */
  const double *detjac_upts; /* (n_upts,n_eles)               src/eles.cpp:4048 */
  const double *JGinv_upts;  /* (n_dims,n_dims,n_upts,n_eles)  :4050 */
  const double *detjac_fpts; /* (n_fpts,n_eles)               :4230 */
  const double *JGinv_fpts;  /* (n_dims,n_dims,n_fpts,n_eles)  :4233 */
  const double *tdA_fpts;    /* (n_fpts,n_eles)               :4234 */
  const double *norm_fpts;   /* (n_fpts,n_eles,n_dims)        :4235 */
} hfx_eles_desc;

/* arrays of an element block addressable by upload/download */
enum hfx_array_id
{
  HFX_DISU_UPTS0 = 0,   /* disu_upts(0)      (n_upts,n_eles,n_fields) */
  HFX_DISU_UPTS1 = 1,   /* disu_upts(1) */
  HFX_DISU_FPTS = 2,    /* (n_fpts,n_eles,n_fields) */
  HFX_TDISF_UPTS = 3,   /* (n_upts,n_eles,n_fields,n_dims) */
  HFX_NORM_TDISF_FPTS = 4,
  HFX_NORM_TCONF_FPTS = 5,
  HFX_DIV_TCONF_UPTS = 6, /* div_tconf_upts(0) */
  HFX_DELTA_DISU_FPTS = 7,
  HFX_GRAD_DISU_UPTS = 8, /* (n_upts,n_eles,n_fields,n_dims) */
  HFX_GRAD_DISU_FPTS = 9, /* (n_fpts,n_eles,n_fields,n_dims) */
  HFX_SRC_UPTS = 10,      /* (n_upts,n_eles,n_fields); zero unless uploaded */
  HFX_DT_LOCAL = 11,      /* (n_eles) */
  HFX_SENSOR = 12,        /* (n_eles) eles::sensor, written by hfx_eles_shock_capture */
  HFX_SGSF_UPTS = 13,     /* (n_upts,n_eles,n_fields,n_dims) LES: transformed SGS flux (allocated by hfx_eles_set_les) */
  HFX_SGSF_FPTS = 14,     /* (n_fpts,n_eles,n_fields,n_dims) LES: physical SGS flux at the flux points */
  HFX_DISUF_UPTS = 15,    /* (n_upts,n_eles,n_fields) LES closures 2-4: filtered solution (hfx_eles_set_les_filter allocates 15-17) */
  HFX_LU = 16,            /* (n_upts,n_eles,3|6) Leonard tensor of the similarity term */
  HFX_LE = 17,            /* (n_upts,n_eles,n_dims) Leonard vector */
  HFX_N_ARRAYS = 18
};

/* which implementation the operator contractions use */
enum hfx_contract_mode
{
  HFX_CONTRACT_AUTO = 0,   /* sparse when the registered operator is sparse enough, else dense */
  HFX_CONTRACT_DENSE = 1,  /* dense FP64 MFMA GEMM, as the reference's sparse_* = 0 */
  HFX_CONTRACT_SPARSE = 2  /* exact-zero skipping, as the reference's sparse_* = 1 (src/eles.cpp:3114) */
};

const char *hfx_last_error(void);
int hfx_version(void);

/* ---- context ---------------------------------------------------------- */
int hfx_ctx_create(int device, hfx_ctx **out);
int hfx_ctx_destroy(hfx_ctx *ctx);
int hfx_ctx_set_params(hfx_ctx *ctx, const hfx_params *p);
int hfx_ctx_set_contract_mode(hfx_ctx *ctx, int mode);
/* which variant of the split fused stage (pairwise face kernels + element kernels, three or four launches per stage) the
 * measurement entry points describe and hfx_stage_partitioned / hfx_run_steps_partitioned run: 2 keeps the reference's
 * gradient arrays in HBM, 3 (default) evaluates the fluxes in the gradient kernel */
int hfx_ctx_set_fused_mode(hfx_ctx *ctx, int mode);
/* run_input.CFL for dt_type 1 / 2 (src/input.cpp:141-158): hfx_run_steps and hfx_run_steps_partitioned then start every
 * time step with calc_time_step (src/HiFiLES.cpp:198, src/solver.cpp:484-549) */
int hfx_ctx_set_CFL(hfx_ctx *ctx, double CFL);
/* Measurement knobs: kernel variants with the same results (A/B runs; defaults are the product path).  name:
 * "split_grid_per_cu" (0 = the workgroups resident at once | n), "xcd_order" (1), "dictionary_rows" (0), "flux_waves" (2 | 3), "buffer_addressing" (1),
 * "loader_wave" (1), "fold_general" (1: the general fused stage applies opp_2 - opp_3 opp_1 and never forms norm_tdisf), "gather_delta" (1: the loader-wave flux kernel forms the interior LDG corrections itself, no pairwise
 * LDG launch), "simd_roles" (1: the flux kernel deals its waves' parts by SIMD), "comm_stream_faces" (1:
 * hfx_run_steps_partitioned launches the one-sided partition-face kernels on the communication stream), "flux_stamps" (0; n >= 1: cycle
 * stamps of iteration max(n, 2) of one workgroup of the flux kernels, printed by the hfx_time_* entry points), "tensor_ops" (1),
 * "split_flux" (1: hfx_run_steps_partitioned runs the flux kernel in three launches -- half of the elements without partition-face points,
 * those with, the other half -- so that both exchanges run beside element work), "split_update" (1: hfx_run_steps_partitioned updates the elements with partition-face points in a first launch, so that their exchange
 * runs beside the update of the others), "light_wave_short" (1: a flux-kernel wave without solution points runs the flux-point physics alone), "les_flux_kernel" (1: the LES
 * closure inside the flux kernel of variant 3), "over_int_fold" (1: the sum-factorised over-integration kernel hands the loader-wave flux kernel its
 * contribution to the divergence, n_fields values per solution point, instead of tdisf_upts), "bdy_beside" (0; 1: the fused stages' viscous boundary-face kernels on a side stream
 * beside the interior-face kernel), "general_waves" (0 = by LDS image | 3 | 4 | 8), "general_update_waves" (0 = by the staging registers | 4 | 8), "dense_waves" (0 = by the operator's rows | 4 | 8) and "dense_split" (0 | 1 | 2 | 4:
 * shape of the dense MFMA contraction's workgroup) -- see hfx_ctx::Options in csrc/hfx_internal.hpp; and "deferred" (0), which is
 * not a measurement knob: see below;
 * tests/test_gpu_fused.py::test_split3_variant_knobs_agree holds the variants to each other. */
int hfx_ctx_set_option(hfx_ctx *ctx, const char *name, int value);
/* ---- deferred execution: the reference's UNCHANGED call sequence on the fused stages --------------------------------
 * hfx_ctx_set_option(ctx, "deferred", 1).  From then on the per-method entry points below -- exactly the calls
 * CalcResidual (src/solver.cpp:59-221) and the RK loop (src/HiFiLES.cpp:201-217) make: hfx_eles_extrapolate_solution ...
 * hfx_eles_calculate_corrected_divergence, hfx_eles_calc_sgs_terms, hfx_eles_extrapolate_sgsFlux,
 * hfx_int_inters_calculate_common_*, hfx_bdy_inters_evaluate_boundaryConditions_*, hfx_mpi_inters_send_* / receive_* /
 * calculate_common_*, hfx_eles_AdvanceSolution, hfx_eles_shock_capture -- RECORD the call and return.  The record runs
 * when the next stage begins (the first call behind an AdvanceSolution / shock_capture) or when any other entry point needs
 * the device state (upload, download, monitors, calc_dt_local, new parameters, synchronize, hfx_ctx_flush ...):
 *   - a record that is one whole stage in CalcResidual's order (every method over every block it belongs to, phases in
 *     the reference's order, one stage number) runs as ONE fused stage: the split stage for a tensor-product block
 *     (= hfx_run_steps(..., 3), or 2 in fused mode 2 / with an LES closure the flux kernel cannot carry), the partitioned split
 *     stage when the block has partition faces whose send_* / receive_* calls name one hfx_comm (= hfx_run_steps_partitioned),
 *     the general stage for tetrahedra / prisms / mixed meshes (= hfx_run_steps_blocks(..., 4)), with partition faces
 *     = hfx_run_steps_partitioned_blocks;
 *   - anything else -- a partial stage, another order, a block the fused stages refuse, the packing halves
 *     hfx_mpi_inters_pack_* of a caller-side transport -- is replayed call by call: the per-method path, unchanged.
 * Results are those of hfx_run_steps* (1e-11 of the per-method path).  The fused stages leave disu_upts(0),
 * disu_upts(1), disu_fpts of the new state, and div_tconf_upts at the last stage of a step (where the monitors read it,
 * src/output.cpp:2166) or when a pending stage is flushed by a request for it.  A download of any OTHER array that arrives
 * while a whole stage is pending makes that stage run call by call, so that the caller sees the reference's values (a
 * monitor that reads grad_disu_upts after a step, src/eles.cpp:5485, costs one per-method stage); once the next stage has
 * begun such an array is stale and its download FAILS rather than return older values.
 * An error in a recorded call surfaces at the entry point that makes the record run. */
int hfx_ctx_flush(hfx_ctx *ctx); /* run what has been recorded (asynchronously, on the context's stream); no-op otherwise */
/* stages run fused / records replayed call by call since the context was created, and why the last replay was one */
int hfx_ctx_deferred_stats(hfx_ctx *ctx, long *n_fused, long *n_replayed, const char **why_last_replay);
/* run_input.dt as the last calc_time_step left it (dt_type 1), or as set (dt_type 0) */
int hfx_ctx_get_dt(hfx_ctx *ctx, double *dt);
int hfx_ctx_synchronize(hfx_ctx *ctx);
/* the HIP stream (hipStream_t) all kernels of this context are launched on */
void *hfx_ctx_stream(hfx_ctx *ctx);

/* ---- element blocks (reference class eles) ----------------------------- */
int hfx_eles_create(hfx_ctx *ctx, const hfx_eles_desc *desc, hfx_eles **out);
int hfx_eles_destroy(hfx_eles *e);
/* replaces hf_array::cp_cpu_gpu / cp_gpu_cpu (include/hf_array.h:541-580) */
int hfx_eles_upload(hfx_eles *e, int array_id, const double *host);
int hfx_eles_download(hfx_eles *e, int array_id, double *host);
/* deferred execution: *current = 1 when the array holds the last stage's values; 2 when it does not yet but a whole stage is
 * pending that a download of this array would run call by call (and so refresh it); 0 when the fused stage that ran last did
 * not refresh it and nothing pending will (its download would fail) */
int hfx_eles_is_current(hfx_eles *e, int array_id, int *current);
/* raw device pointer of an array (for zero-copy interop with a resident caller) */
int hfx_eles_device_ptr(hfx_eles *e, int array_id, double **dev);

int hfx_eles_extrapolate_solution(hfx_eles *e);           /* eles::extrapolate_solution           src/eles.cpp:1360 */
int hfx_eles_calculate_gradient(hfx_eles *e);             /* eles::calculate_gradient             :1823 */
int hfx_eles_evaluate_invFlux(hfx_eles *e);               /* eles::evaluate_invFlux               :1415 */
int hfx_eles_correct_gradient(hfx_eles *e);               /* eles::correct_gradient               :1890 */
int hfx_eles_evaluate_viscFlux(hfx_eles *e);              /* eles::evaluate_viscFlux              :2285 */
int hfx_eles_extrapolate_totalFlux(hfx_eles *e);          /* eles::extrapolate_totalFlux          :1549 */
int hfx_eles_calculate_divergence(hfx_eles *e);           /* eles::calculate_divergence           :1651 */
int hfx_eles_calculate_corrected_divergence(hfx_eles *e); /* eles::calculate_corrected_divergence :1738 */
int hfx_eles_AdvanceSolution(hfx_eles *e, int in_step, int adv_type); /* eles::AdvanceSolution    :1080 */
/* the NaN scan of src/eles.cpp:1781-1795 as a device flag: *first_nan = flat index
 * into div_tconf_upts(0) of a NaN seen since the last call, or -1.  Synchronizes. */
int hfx_eles_check_nan(hfx_eles *e, long *first_nan);
/* eles::compute_res_upts (src/eles.cpp:5045): norm_type 0 max, 1 L1 sum, 2 L2 sum */
int hfx_eles_compute_res_upts(hfx_eles *e, int norm_type, int field, double *out);

/* ---- interior faces (reference class int_inters) ----------------------- */
/* L, R: (n_fpts_per_inter, n_inters) offsets `fpt + n_fpts*ele` into the
 * (fpt,ele) plane of the left / right element block: the hf_array<double*>
 * tables of include/inters.h:86-116 minus the owning array base, with the
 * flux-point permutation `lut` (src/inters.cpp:153-262) already applied to R,
 * exactly as int_inters::set_interior (src/int_inters.cpp:67-121) wires them. */
int hfx_int_inters_create(hfx_ctx *ctx, hfx_eles *left, hfx_eles *right, int n_inters, int n_fpts_per_inter,
                          const int *L, const int *R, hfx_inters **out);
int hfx_inters_destroy(hfx_inters *f);
int hfx_int_inters_calculate_common_invFlux(hfx_inters *f);  /* int_inters::calculate_common_invFlux  src/int_inters.cpp:160 */
int hfx_int_inters_calculate_common_viscFlux(hfx_inters *f); /* int_inters::calculate_common_viscFlux src/int_inters.cpp:254 */

/* ---- partition faces (reference class mpi_inters) ------------------------ */
/* A partition face has its LEFT side on this rank; the right state arrives through an exchange.
 * L: (n_fpts_per_inter, n_inters) left offsets as for interior faces.
 * Rlut: (n_fpts_per_inter, n_inters) the flux-point slot `lut(j)` of the RECEIVED face record that
 *   meets left flux point j (mpi_inters::set_mpi wires disu_fpts_r(j,i,k) to
 *   in_buffer_disu(lut(j), k, i), src/mpi_inters.cpp:165-172).
 * Buffers (device, owned by the block): out/in_buffer_disu (fpt, field, inter) and
 *   out/in_buffer_grad_disu (fpt, field, dim, inter), filled in loop order inter -> [dim ->] field -> fpt
 *   (src/mpi_inters.cpp:56-66,225-229,284-289).  Faces of one neighbour rank are contiguous, in the
 *   order both ranks agreed on (src/geometry.cpp:1184-1239), so a message is one contiguous slice.
 * The exchange moves out_buffer slices to the neighbours' in_buffer slices between `pack` and `calculate_common_*`:
 *   either inside the library (hfx_comm_*, hfx_mpi_inters_send_* / receive_* below: grouped ncclSend / ncclRecv on the
 *   library's communication stream) or by the caller (the hfx_mpi_inters_pack_* halves + its own transport). */
int hfx_mpi_inters_create(hfx_ctx *ctx, hfx_eles *left, int n_inters, int n_fpts_per_inter, const int *L,
                          const int *Rlut, hfx_inters **out);
/* the packing half of mpi_inters::send_solution / send_corrected_gradient (src/mpi_inters.cpp:218-229,278-289) */
int hfx_mpi_inters_pack_solution(hfx_inters *f);
int hfx_mpi_inters_pack_corrected_gradient(hfx_inters *f);
/* LES, the packing half of mpi_inters::send_sgsf_fpts (src/mpi_inters.cpp:339-351): out_buffer_sgsf = the left block's
 * physical SGS flux at the partition faces' flux points (after hfx_eles_extrapolate_sgsFlux) */
int hfx_mpi_inters_pack_sgsf(hfx_inters *f);
/* device pointers and lengths (doubles) of the buffers: which = 0 out_disu, 1 in_disu, 2 out_grad, 3 in_grad;
 * 4 / 5 = the leading n_fpts_per_inter*n_fields*n_inters doubles of out_grad / in_grad, which is what
 * hfx_stage_partitioned sends in fused mode 3 (the projected viscous flux instead of the gradient);
 * 6 / 7 = out / in_buffer_sgsf (LES only; (fpt, field, dim, inter) like the gradient) */
int hfx_mpi_inters_buffer(hfx_inters *f, int which, double **dev, long *n);
int hfx_mpi_inters_calculate_common_invFlux(hfx_inters *f);  /* mpi_inters::calculate_common_invFlux  src/mpi_inters.cpp:400 */
int hfx_mpi_inters_calculate_common_viscFlux(hfx_inters *f); /* mpi_inters::calculate_common_viscFlux src/mpi_inters.cpp:485 */

/* ---- LES sub-grid closure (8f-4) ----------------------------------------- */
/* run_input.LES == 1 with an eddy-viscosity model (src/eles.cpp:2395-2650): sgs_model 0 Smagorinsky with near-wall
 * damping (wall_distance (n_upts,n_eles,n_dims), eles::calc_wall_distance), 1 WALE.  Jacobian_fpts
 * (n_dims,n_dims,n_fpts,n_eles) is what extrapolate_sgsFlux uses to take the flux back to physical space
 * (src/eles.cpp:2862-2893).  Once set, evaluate_viscFlux adds the SGS flux, hfx_CalcResidual calls
 * extrapolate_sgsFlux (src/solver.cpp:162-167) and interior faces add sgsf_fpts to both sides' viscous flux
 * (src/int_inters.cpp:302-318), partition faces after the third exchange (hfx_mpi_inters_send / receive_sgsf_fpts).
 * The filter width uses the element class's calc_ele_vol (|J| times the reference element's volume: hexes 8, quads 4,
 * prisms 4, tetrahedra 8/6, triangles 2; the block's ele_type decides).
 * The fused stages evaluate the closure inside their flux kernels -- split variant 3 on hexes / quads (option "les_flux_kernel"),
 * the general stage on tetrahedra / prisms of orders 1..3 -- and add F_sgs . n to the projected viscous flux of every flux point,
 * so a partition face's second message already carries it and the third one is not sent there; hfx_run_steps(..., fused=2) keeps
 * the reference's arrays (gradients, sgsf_upts, sgsf_fpts) and the third message (hfx_stage_partitioned). */
typedef struct hfx_les
{
  int sgs_model, pad; /* 0 Smagorinsky, 1 WALE, 2 WALE + similarity, 3 spectral vanishing viscosity, 4 similarity */
  double C_s, filter_ratio, Kappa, prandtl_t;
} hfx_les;
int hfx_eles_set_les(hfx_eles *e, const hfx_les *les, const double *wall_distance, const double *Jacobian_fpts);
/* Closures 2, 3, 4 filter the solution: filter_upts (n_upts,n_upts) is what the element class builds in
 * compute_filter_upts (src/eles_hexas.cpp:583, src/eles.cpp:138) -- registered like the operators.  Required before the
 * first stage with such a model. */
int hfx_eles_set_les_filter(hfx_eles *e, const double *filter_upts);
/* eles::calc_sgs_terms (src/eles.cpp:2058-2283), which CalcResidual calls at the FIRST RK stage of a time step for the
 * closures 2, 3, 4 (src/solver.cpp:55-62): filtered solution -> HFX_DISUF_UPTS; model 3: it replaces disu_upts(0);
 * models 2 / 4: products, their filtered values and the Leonard terms -> HFX_LU, HFX_LE, which calc_sgsf_upts reads for
 * the rest of the step.  A NaN in the filtered solution is reported through hfx_eles_check_nan.  The hfx_run_steps*
 * loops call it themselves. */
int hfx_eles_calc_sgs_terms(hfx_eles *e);
int hfx_eles_extrapolate_sgsFlux(hfx_eles *e); /* eles::extrapolate_sgsFlux, src/eles.cpp:2817 */

/* ---- integral diagnostics (8f-1: the TGV monitors) ------------------------ */
/* opp_volume_cubpts (n_cubpts,n_upts), weight_volume_cubpts (n_cubpts), vol_detjac_vol_cubpts (n_cubpts,n_eles):
 * what eles::set_opp_volume_cubpts / set_transforms_vol_cubpts build when n_integral_quantities != 0
 * (src/eles.cpp:3667,4027) */
int hfx_eles_set_volume_cubpts(hfx_eles *e, int n_cubpts, const double *opp_volume_cubpts, const double *weight_volume_cubpts,
                               const double *vol_detjac_vol_cubpts);
/* eles::CalcIntegralQuantities (src/eles.cpp:5485-5627): integral_quantities[m] += this block's integral of
 * quantity quantity_ids[m] (0 kineticenergy, 1 enstropy, 2 pressuredilatation, 3 straincolonproduct,
 * 4 devstraincolonproduct) from disu_upts(0) and grad_disu_upts (the corrected gradient of the last
 * per-method / fused-mode-2 stage).  The caller adds over blocks and ranks (src/output.cpp:2017-2050). */
int hfx_eles_CalcIntegralQuantities(hfx_eles *e, int n_quantities, const int *quantity_ids, double *integral_quantities);

/* ---- plot-point interpolation (8f-3: the VTU writer's input) -------------- */
/* opp_p (n_ppts,n_upts): the nodal basis at the plot points, eles::set_opp_p (src/eles.cpp:3600-3621) */
int hfx_eles_set_opp_p(hfx_eles *e, int n_ppts, const double *opp_p);
/* eles::calc_disu_ppts (src/eles.cpp:3757-3778) for every element at once instead of per element:
 * disu_ppts (n_ppts,n_eles,n_fields) = opp_p . disu_upts(0), written to the HOST array the plot writer reads
 * (output::write_vtu loops the elements, src/output.cpp) */
int hfx_eles_calc_disu_ppts(hfx_eles *e, double *disu_ppts_host);

/* ---- CFL time stepping (calc_time_step, src/solver.cpp:484-549) ---------- */
int hfx_eles_set_h_ref(hfx_eles *e, const double *h_ref); /* eles::h_ref (n_eles), src/eles.cpp:3985 */
/* dt_local(ic) = eles::calc_dt_local(ic) (src/eles.cpp:1267-1356) for every element -> HFX_DT_LOCAL, and the
 * block's minimum.  dt_type 1: the caller takes the minimum over blocks and ranks (all-reduce MIN) and sets
 * hfx_params.dt; dt_type 2: AdvanceSolution reads HFX_DT_LOCAL. */
int hfx_eles_calc_dt_local(hfx_eles *e, double CFL, double *dt_min);

/* ---- over-integration (row a5) ------------------------------------------- */
/* Registers what eles_hexas::set_over_int (src/eles_hexas.cpp:1096-1129) and set_transforms build when
 * run_input.over_int == 1: opp_over_int_cubpts (n_cubpts,n_upts), over_int_filter (n_upts,n_cubpts),
 * JGinv_over_int_cubpts (n_dims,n_dims,n_cubpts,n_eles).  Once set, hfx_CalcResidual evaluates the inviscid
 * flux through evaluate_invFlux_over_int as the reference does (src/solver.cpp:82-91). */
int hfx_eles_set_over_int(hfx_eles *e, int n_cubpts, const double *opp_over_int_cubpts, const double *over_int_filter,
                          const double *JGinv_over_int_cubpts);
int hfx_eles_evaluate_invFlux_over_int(hfx_eles *e); /* eles::evaluate_invFlux_over_int, src/eles.cpp:1480-1545 */

/* ---- shock capturing (row a16) ------------------------------------------ */
/* Registers what eles_hexas / eles_quads build when run_input.shock_cap == 1 (src/eles_hexas.cpp:74-85):
 * inv_vandermonde (n_upts,n_upts), exp_filter (n_upts,n_upts), norm_basis_persson (n_upts), and
 * high_modes(j) = 1 where a 1-D index of Legendre mode j equals the order (the set summed in
 * shock_det_persson, src/eles_hexas.cpp:1042-1047); s0 and shock_det_field are run_input's. */
int hfx_eles_set_shock_capture(hfx_eles *e, const double *inv_vandermonde, const double *exp_filter,
                               const double *norm_basis_persson, const int *high_modes, double s0, int shock_det_field);
/* eles::shock_capture (src/eles.cpp:2918-2959), shock_det 0 (Persson) + shock_cap 1 (exponential modal
 * filter): sensor(ele) -> HFX_SENSOR; disu_upts(0) of every element with sensor >= s0 is filtered.
 * Called by the caller after AdvanceSolution (src/HiFiLES.cpp:214-216); invalidates disu_fpts. */
int hfx_eles_shock_capture(hfx_eles *e);

/* ---- boundary faces (reference class bdy_inters) ------------------------ */
/* bc_flag values of the reference (src/bc.cpp:36-48) */
enum hfx_bc_flag
{
  HFX_BC_SUB_IN_SIMP = 0, HFX_BC_SUB_OUT_SIMP = 1, HFX_BC_SUB_IN_CHAR = 2, HFX_BC_SUB_OUT_CHAR = 3, HFX_BC_SUP_IN = 4,
  HFX_BC_SUP_OUT = 5, HFX_BC_SLIP_WALL = 6, HFX_BC_CYCLIC = 7, HFX_BC_ISOTHERM_WALL = 8, HFX_BC_ADIABAT_WALL = 9,
  HFX_BC_CHAR = 10, HFX_BC_SLIP_WALL_DUAL = 11
};
/* one entry of run_input.bc_list (include/bc.h:48-62), values AFTER input::read_boundary_param's
 * non-dimensionalisation (src/input.cpp:440-525); fields a type does not use are ignored */
typedef struct hfx_bc
{
  int flag, pressure_ramp, use_wm, pad;
  double rho, velocity[3], p_static, T_static, p_total, T_total, nx, ny, nz;
  double p_ramp_coeff, T_ramp_coeff, p_total_old, T_total_old;
} hfx_bc;
/* L(j,i): offset of the face's flux point in the left block's (fpt,ele) plane (what bdy_inters::set_boundary,
 * src/bdy_inters.cpp:75-135, stores as pointers); boundary_id(i): index into bcs (bdy_inters::boundary_id);
 * R_ref: run_input.R_ref for viscous runs, run_input.R_gas for inviscid ones (src/bdy_inters.cpp:368-369).
 * Wall-model groups (use_wm) and the LES inlet are not part of this path and are refused. */
int hfx_bdy_inters_create(hfx_ctx *ctx, hfx_eles *left, int n_inters, int n_fpts_per_inter, const int *L,
                          const int *boundary_id, const hfx_bc *bcs, int n_bcs, double R_ref, hfx_inters **out);
int hfx_bdy_inters_set_ramp_counter(hfx_inters *f, int ramp_counter); /* run_input.ramp_counter (pressure ramps) */
int hfx_bdy_inters_evaluate_boundaryConditions_invFlux(hfx_inters *f, double time_bound);  /* src/bdy_inters.cpp:213 */
int hfx_bdy_inters_evaluate_boundaryConditions_viscFlux(hfx_inters *f, double time_bound); /* src/bdy_inters.cpp:1024 */

/* ---- the caller contract ---------------------------------------------- */
/* CalcResidual (src/solver.cpp:50-223) for one element block and its interior and boundary
 * face blocks (any mix, in `faces`), LES / RANS / forcing off; same call order as the reference. */
int hfx_CalcResidual(hfx_eles *e, hfx_inters *const *faces, int n_face_blocks);
/* The same for a MIXED mesh: several element blocks (the reference's mesh_eles(i), one per element class) and face blocks
 * whose left and right sides may belong to different element blocks (int_inters::set_interior is called with
 * ctype(ic_l), ctype(ic_r), src/geometry.cpp:637-706).  Every method runs for all element blocks before the next one,
 * exactly as src/solver.cpp:59-221 loops `for (i = 0; i < n_ele_types; i++) mesh_eles(i)->method()`. */
int hfx_CalcResidual_blocks(hfx_eles *const *eles, int n_ele_blocks, hfx_inters *const *faces, int n_face_blocks);
/* n_steps time steps = the RK-stage loop of src/HiFiLES.cpp:194-217:
 * for each stage CalcResidual + AdvanceSolution.  `fused`: 0 the per-method path, 2 the split fused
 * kernels, 3 the split kernels with the fluxes evaluated in the gradient kernel (same results to
 * rounding).  All leave disu_upts(0), disu_upts(1), disu_fpts of the new state and, for the step's last
 * stage, div_tconf_upts in the public arrays; mode 2 also leaves grad_disu_upts / grad_disu_fpts of the
 * last stage, mode 3 keeps them in registers and folds opp_3 . norm_tdisf_fpts into the divergence it stores, so
 * norm_tdisf_fpts is not refreshed either; nor is delta_disu_fpts at interior points, whose LDG corrections the flux kernel
 * forms itself from the partner's flux-point solution (run one per-method stage when a monitor needs them).
 * With dt_type 1 / 2 every step starts with calc_time_step (hfx_ctx_set_CFL, hfx_eles_set_h_ref); boundary
 * blocks whose groups ramp get run_input.ramp_counter advanced after every step (src/HiFiLES.cpp:224-225). */
int hfx_run_steps(hfx_eles *e, hfx_inters *const *faces, int n_face_blocks, int n_steps, int fused);
/* The RK loop over several element blocks (mixed meshes).  fused 0: the per-method path; fused 4: the fused stage for
 * general (non-tensor-product) element classes, three-dimensional Navier-Stokes / Euler blocks with interior and
 * boundary faces (csrc/general.hip): three or four launches per element block and stage, the dense operator contractions on the
 * FP64 matrix cores over batches of 16 elements; like fused 3 it keeps the corrected gradients on chip (only boundary
 * points get grad_disu_fpts) and leaves disu_upts(0), disu_upts(1), disu_fpts of the new state and div_tconf_upts;
 * norm_tdisf_fpts (folded into the divergence operator) and delta_disu_fpts of pairs inside a block are not refreshed.
 * What the blocks registered takes part: an LES closure is evaluated inside the flux kernel (tetrahedra / prisms of orders 1..3;
 * calc_sgs_terms before the first stage of a step), over-integration forms tdisf_upts by the dense contractions and the flux kernel
 * takes it, shock capturing follows every stage (the filter, then disu_fpts of the filtered state); a closure TOGETHER with
 * over-integration is refused (run fused 0).
 * A block with one element class may of course be passed alone (hfx_run_steps(e, ..., 4) is the same call). */
int hfx_run_steps_blocks(hfx_eles *const *eles, int n_ele_blocks, hfx_inters *const *faces, int n_face_blocks, int n_steps,
                         int fused);

/* One RK stage of the split fused path on a PARTITIONED block, cut so that the caller can move the
 * partition-face buffers while interior faces are being worked on (the order of CalcResidual's
 * send / receive calls, src/solver.cpp:70-72,134-138,150-154,200-209):
 *   phase 0: (first stage only) extrapolate, pack out_buffer_disu          -> caller STARTS solution exchange
 *   phase 1: LDG common solution on interior faces                         -> caller WAITS for the solution
 *   phase 2: the same on partition faces, corrected gradients, pack out_buffer_grad_disu
 *                                                                          -> caller STARTS gradient exchange
 *   phase 3: common fluxes on interior faces, inviscid part on partition faces
 *                                                                          -> caller WAITS for the gradient
 *   phase 4: viscous common flux on partition faces, residual, RK update (in_step), the new state's
 *            flux-point solution packed into out_buffer_disu               -> caller STARTS solution exchange
 * Inviscid runs have nothing to send after phase 2.  `first` != 0 on the first stage after the caller
 * changed disu_upts(0).  With the context's fused mode 2 the kernels are those of hfx_run_steps(fused=2)
 * and the second exchange carries the corrected gradient (buffers 2/3, as the reference does); otherwise
 * (default) those of fused=3, and the second exchange carries each side's viscous flux projected on
 * its own normal (buffers 4/5: n_fields instead of n_fields*n_dims doubles per flux point) and the
 * partition faces' common flux is evaluated in phase 4. */
int hfx_stage_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces,
                          int n_mpi, int phase, int in_step, int first);

/* ---- partition-face transport inside the library: RCCL point-to-point over xGMI ------------------------
 * Replaces the MPI_Isend / MPI_Irecv / MPI_Waitall of mpi_inters::send_* / receive_* (src/mpi_inters.cpp:244-270,
 * 304-332) by grouped ncclSend / ncclRecv on a communication stream the library owns, ordered against the
 * context's compute stream with events -- no host synchronisation inside a stage.  librccl is resolved at run time
 * (dlopen "librccl.so.1"; the copy already mapped into the process is reused), so callers that never create a
 * communicator do not need it.  Launch contract: one process per GPU; rank 0 calls hfx_comm_get_unique_id and the
 * launcher distributes the 128 bytes (MPI_Bcast in the reference's world, a file / store / gloo broadcast under
 * Python); then every rank calls hfx_comm_create collectively. */
typedef struct hfx_comm hfx_comm;
#define HFX_COMM_ID_BYTES 128
int hfx_comm_get_unique_id(char id[HFX_COMM_ID_BYTES]);
int hfx_comm_create(hfx_ctx *ctx, const char id[HFX_COMM_ID_BYTES], int nranks, int rank, hfx_comm **out);
int hfx_comm_destroy(hfx_comm *c);
/* what RCCL reports for this communicator: ncclCommCount, ncclCommUserRank, ncclCommCuDevice, and that device's PCI bus id
 * (hipDeviceGetPCIBusId) -- evidence in a benchmark line that N ranks on N different devices took part */
int hfx_comm_info(hfx_comm *c, int *nranks, int *rank, int *device, char pci_bus_id[32]);
/* MPI_Allreduce(MIN / MAX / SUM) of a few doubles (calc_time_step's dt, src/solver.cpp:511,543; the monitors'
 * reductions); op 0 min, 1 max, 2 sum.  Synchronises the host with the communication stream. */
int hfx_comm_allreduce(hfx_comm *c, double *values, int n, int op);
/* Neighbour table of a partition-face block = mpi_inters::Nout_proc generalised to segments: the faces
 * [send_first[s], send_first[s]+count[s]) of out_buffer_* go to rank peer[s], and what peer[s] sends in return lands
 * in the faces [recv_first[s], recv_first[s]+count[s]) of in_buffer_*.  The reference's layout (faces of one rank
 * contiguous, ordered by rank, src/mpi_inters.cpp:244-256) is send_first == recv_first == running sum of Nout_proc.
 * peer[s] == own rank is allowed (a periodic direction that is not split). */
int hfx_mpi_inters_set_neighbours(hfx_inters *f, int n_seg, const int *peer, const int *send_first, const int *recv_first,
                                  const int *count);
/* mpi_inters::send_solution / receive_solution / send_corrected_gradient / receive_corrected_gradient
 * (src/mpi_inters.cpp:218-336): send_* packs on the compute stream and starts the grouped exchange on the
 * communication stream; receive_* makes the compute stream wait for it (the MPI_Waitall). */
int hfx_mpi_inters_send_solution(hfx_inters *f, hfx_comm *c);
int hfx_mpi_inters_receive_solution(hfx_inters *f, hfx_comm *c);
int hfx_mpi_inters_send_corrected_gradient(hfx_inters *f, hfx_comm *c);
int hfx_mpi_inters_receive_corrected_gradient(hfx_inters *f, hfx_comm *c);
/* LES: mpi_inters::send_sgsf_fpts / receive_sgsf_fpts (src/mpi_inters.cpp:339-397), called by CalcResidual after
 * extrapolate_sgsFlux and before the partition faces' viscous flux (src/solver.cpp:168-178,203-206) */
int hfx_mpi_inters_send_sgsf_fpts(hfx_inters *f, hfx_comm *c);
int hfx_mpi_inters_receive_sgsf_fpts(hfx_inters *f, hfx_comm *c);
/* n_steps time steps of the split fused path on a partitioned block: the five phases of hfx_stage_partitioned with
 * the two exchanges of every stage started and awaited from inside the library in CalcResidual's order
 * (src/solver.cpp:68-72,131-139,148-155,197-210): solution exchange in flight during the interior LDG sweep, flux /
 * gradient exchange in flight during the interior common-flux sweep.  With dt_type 1 / 2 the step starts with
 * calc_time_step (src/HiFiLES.cpp:198): per-element CFL steps, block minimum, all-reduce MIN over the ranks.
 * h_ref (hfx_eles_set_h_ref) and run_input.CFL (hfx_ctx_set_CFL) must have been set then.  Boundary blocks in
 * int_faces get run_input.ramp_counter advanced after every step when one of their groups ramps
 * (src/HiFiLES.cpp:224-225). */
int hfx_run_steps_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                              hfx_comm *comm, int n_steps);
/* The same for a mesh of SEVERAL element blocks and / or non-tensor classes (tetrahedra, prisms, mixed meshes): the general fused
 * stage (hfx_run_steps_blocks(..., 4)) on partitioned blocks.  A partition-face block belongs to one element block
 * (mpi_inters::set_mpi takes any in_ele_type_l, src/mpi_inters.cpp:154); int_faces holds the interior blocks (whose two sides may
 * lie in different element blocks) and the boundary blocks.  Each stage: LDG corrections and common fluxes of the partition faces
 * by the one-sided kernels, the flux-point solution and each side's projected viscous flux Fn as the two messages.  Fixed time
 * step (dt_type 0) only; three-dimensional blocks as hfx_run_steps_blocks(..., 4) takes them -- an LES closure is evaluated in the
 * flux kernel (its F_sgs . n travels inside Fn: no third message; the SVV closure is refused here), over-integration feeds it,
 * shock capturing follows the update before the next solution message is packed.
 * With deferred execution the mirrored CalcResidual of such a mesh runs this stage when its send_* / receive_* name one hfx_comm. */
int hfx_run_steps_partitioned_blocks(hfx_eles *const *eles, int n_ele_blocks, hfx_inters *const *int_faces, int n_int,
                                     hfx_inters *const *mpi_faces, int n_mpi, hfx_comm *comm, int n_steps);
/* Average durations (ms, HIP events) over `reps` stages of the same loop: ms[0..3] phases 1-4 on the compute stream,
 * ms[4] solution exchange and ms[5] flux / gradient exchange on the communication stream (from the moment their
 * data is packed to the last byte received), ms[6] the whole stage, ms[7] the element kernel of phase 2 alone (the split flux
 * kernel; 0 for inviscid runs and fused mode 2).  This is the SERIALISED schedule: every kernel on the compute stream so that
 * the events bracket the phases -- hfx_run_steps_partitioned itself puts the one-sided partition-face kernels on the
 * communication stream (option "comm_stream_faces") and is timed as a whole by the caller.  The state advances by reps stages. */
int hfx_time_partitioned(hfx_eles *e, hfx_inters *const *int_faces, int n_int, hfx_inters *const *mpi_faces, int n_mpi,
                         hfx_comm *comm, int reps, double ms[8]);

/* ---- measurement ------------------------------------------------------- */
/* Average duration (ms, HIP events on the context's stream) of every per-method entry
 * point over `reps` repetitions of one RK stage, in CalcResidual order:
 *  0 extrapolate_solution  1 calculate_gradient  2 evaluate_invFlux  3 common_invFlux (all face blocks)
 *  4 correct_gradient      5 evaluate_viscFlux   6 extrapolate_totalFlux  7 calculate_divergence
 *  8 common_viscFlux       9 calculate_corrected_divergence  10 AdvanceSolution
 * The state advances by reps stages (stage index cycles through the scheme). */
#define HFX_N_TIMED_METHODS 11
int hfx_time_methods(hfx_eles *e, hfx_inters *const *faces, int n_face_blocks, int reps, double ms[HFX_N_TIMED_METHODS]);
/* The same for the kernels of the fused path: ms[8] (unused entries 0), names = comma-separated
 * kernel names (buffer of 256 chars); the state advances by reps stages. */
int hfx_time_fused_kernels(hfx_eles *e, hfx_inters *const *faces, int n_face_blocks, int reps, double ms[8], char names[256]);
/* ALGORITHMIC HBM bytes per launch of each fused kernel (same order), see DESIGN.md */
int hfx_fused_kernel_bytes(hfx_eles *e, double bytes[8]);
/* The same for the general fused stage (hfx_run_steps_blocks(..., fused = 4)): ms[0..3] = pairwise LDG kernels, flux
 * kernels of all element blocks, pairwise common-flux kernels, update kernels of all blocks */
int hfx_time_general_kernels(hfx_eles *const *eles, int n_ele_blocks, hfx_inters *const *faces, int n_face_blocks, int reps,
                             double ms[8], char names[256]);
int hfx_general_kernel_bytes(hfx_eles *const *eles, int n_ele_blocks, double bytes[8]);

#ifdef __cplusplus
}
#endif
#endif
