// eles.hpp -- host-side mirror of the reference's element classes for the hot path.
//
// Same public method names, argument meaning and array layouts as
// /root/reference/include/eles.h:55-125 (eles), eles_hexas.h, eles_quads.h, so that a
// CalcResidual written against the reference's interface (src/solver.cpp:50-223) drives
// this implementation unchanged.  The per-stage methods do no arithmetic on the host:
// they call the C ABI of libhfx (include/hfx.h), which owns the device copy of every array.
// The setup methods (operators, metrics, initial condition) are host code, as in the reference.
#pragma once
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/hfx.h"
#include "hf_array.hpp"
#include "input.hpp"

class eles
{
public:
  eles();
  virtual ~eles();

  // ---- setup (reference: eles::setup src/eles.cpp:66-235) -------------------------
  // returns 0 or non-zero with last_error() set (the reference calls FatalError)
  int setup(int in_n_eles, int in_max_n_spts_per_ele, input *in_run_input);
  void set_shape_node(int in_spt, int in_ele, const hf_array<double> &in_pos); // src/eles.cpp set_shape_node
  int set_transforms();                                                          // src/eles.cpp:4015
  int set_ics(double &time);                                                     // src/eles.cpp:237

  // ---- device residency (replaces mv_all_cpu_gpu / cp_*_gpu_cpu, src/eles.cpp:930-1062)
  int mv_all_cpu_gpu(hfx_ctx *ctx);
  void free_device();
  int cp_disu_upts_gpu_cpu();
  int cp_disu_upts_cpu_gpu();
  int cp_div_tconf_upts_gpu_cpu();
  int cp_grad_disu_upts_gpu_cpu();
  int cp_array_gpu_cpu(int hfx_id, hf_array<double> &dst);

  // ---- the per-stage methods CalcResidual calls (src/solver.cpp:65-216) -------------
  void extrapolate_solution();
  void calculate_gradient();
  void evaluate_invFlux();
  void evaluate_invFlux_over_int(); // src/eles.cpp:1480 (run_input.over_int)
  void shock_capture();             // src/eles.cpp:2918 (run_input.shock_cap), after AdvanceSolution
  void correct_gradient();
  void evaluate_viscFlux();
  void extrapolate_sgsFlux(); // src/eles.cpp:2817 (run_input.LES)
  void calc_sgs_terms();      // src/eles.cpp:2058 (SGS_model 2, 3, 4: at the first RK stage of a step)
  void extrapolate_totalFlux();
  void calculate_divergence();
  void calculate_corrected_divergence();
  void AdvanceSolution(int in_step, int adv_type);
  double calc_dt_local(int in_ele); // host evaluation on the host copy of disu_upts(0), src/eles.cpp:1267
  double compute_res_upts(int in_norm_type, int in_field);

  // ---- getters used by face wiring (src/eles.cpp:4638-4949 return pointers; here: offsets)
  int get_n_eles() const { return n_eles; }
  int get_n_dims() const { return n_dims; }
  int get_n_fields() const { return n_fields; }
  int get_n_fpts_per_inter(int f) const { return n_fpts_per_inter(f); }
  // offset `fpt + n_fpts*ele` of local flux point `in_fpt` of local face `in_inter` of element `in_ele`
  int get_fpt_offset(int in_ele, int in_inter, int in_fpt) const;
  hfx_eles *device() { return dev; }
  const std::string &last_error() const { return err; }
  bool failed() const { return !err.empty(); }

  // ---- data (public here; protected in the reference: include/eles.h:470-935) --------
  input *run_input = nullptr;
  int n_eles = 0, n_dims = 0, n_fields = 0, order = 0, ele_type = -1, viscous = 0;
  int n_upts_per_ele = 0, n_fpts_per_ele = 0, n_inters_per_ele = 0, upts_type = 0;
  hf_array<int> n_fpts_per_inter, n_spts_per_ele;
  hf_array<double> loc_1d_upts, loc_upts, tloc_fpts, tnorm_fpts;
  hf_array<double> shape; // (n_dims, max_n_spts, n_eles)
  hf_array<double> opp_0, opp_3, opp_6;
  hf_array<hf_array<double>> opp_1, opp_2, opp_4, opp_5;
  hf_array<double> detjac_upts, JGinv_upts, detjac_fpts, JGinv_fpts, tdA_fpts, norm_fpts, pos_upts, pos_fpts;
  hf_array<hf_array<double>> disu_upts;      // host copies, valid after cp_*_gpu_cpu
  hf_array<hf_array<double>> div_tconf_upts;
  hf_array<double> grad_disu_upts;
  hf_array<double> h_ref, dt_local;
  hf_array<double> Jacobian_fpts; // (n_dims,n_dims,n_fpts,n_eles) d(pos)/d(loc) at the flux points, LES only (src/eles.cpp:4231)
  // eles::calc_wall_distance (src/eles.cpp:2701-2810): for every solution point the vector from the nearest flux point of a no-slip
  // wall face (brute force; loc_noslip_bdy: (n_dims, n_points) column-major, faces in the mesh's order); the damped Smagorinsky closure
  // uses its length
  void calc_wall_distance(const std::vector<double> &loc_noslip_bdy);
  hf_array<double> wall_distance; // (n_upts,n_eles,n_dims)
  // ---- modal operators of the tensor-product classes (eles_modal.cpp): shock capturing and over-integration
  // (include/eles.h:926-935; eles_hexas.h / eles_quads.h: vandermonde, inv_vandermonde, norm_basis_persson)
  hf_array<double> vandermonde, inv_vandermonde, exp_filter, norm_basis_persson;
  hf_array<int> persson_high_modes; // 1: a mode with a degree == order in some direction
  hf_array<double> loc_over_int_cubpts, weight_over_int_cubpts, opp_over_int_cubpts, over_int_filter, JGinv_over_int_cubpts;
  // ---- plot points (src/eles_hexas.cpp:498-521 set_loc_ppts, src/eles.cpp:3600 set_opp_p, :3757 calc_disu_ppts)
  int p_res = 0, n_ppts_per_ele = 0;
  hf_array<double> loc_ppts, opp_p, disu_ppts; // disu_ppts (n_ppts, n_eles, n_fields): filled by calc_disu_ppts_all
  void set_loc_ppts();
  void set_opp_p();
  int calc_disu_ppts_all();                                         // every element at once, on the device
  void calc_disu_ppts(int in_ele, hf_array<double> &out_disu_ppts); // the reference's per-element accessor
  virtual int set_shock_capture_operators(); // set_vandermonde1D/3D, calc_norm_basis, set_exp_filter; eles_tets / eles_pris override it
  virtual int compute_filter_upts(); // src/eles_hexas.cpp:583, src/eles_quads.cpp:428 (LES_filter); eles_tets overrides it
  // reference length of an element for the CFL time step (calc_h_ref_specific of the class); the box meshes of hexes / quads set
  // h_ref themselves (solver.cpp), the simplex classes here
  virtual double calc_h_ref_specific(int) { return 0.0; }
  hf_array<double> filter_upts_1D, filter_upts;
  int set_over_int();                // set_over_int (cubature points, interpolation, L2-projection filter)
  void tensor_modes(hf_array<int> &deg) const;
  // ---- ASCII restart (src/eles.cpp:655-760,845-870; info blocks src/eles_hexas.cpp:799-890, eles_quads.cpp)
  hf_array<int> ele2global_ele;
  int order_rest = 0, n_upts_per_ele_rest = 0;
  hf_array<double> loc_1d_upts_rest, opp_r;
  void write_restart_info_ascii(std::ostream &restart_file);
  void write_restart_data_ascii(std::ostream &restart_file);
  int read_restart_info_ascii(std::istream &restart_file); // 1 found, 0 not in the file
  int read_restart_data_ascii(std::istream &restart_file); // fills disu_upts(0); returns 0 or fails
  void set_opp_r();

protected:
  virtual int setup_ele_type_specific() = 0;
  virtual double eval_nodal_basis(int in_index, const hf_array<double> &in_loc) = 0;
  virtual double eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &in_loc) = 0;
  virtual void fill_opp_3(hf_array<double> &opp_3) = 0;
  virtual double eval_nodal_s_basis(int in_index, const hf_array<double> &in_loc, int in_n_spts) = 0;
  virtual void eval_d_nodal_s_basis(hf_array<double> &d_nodal_s_basis, const hf_array<double> &in_loc, int in_n_spts) = 0;

  void set_opp_0();
  void set_opp_1();
  void set_opp_2();
  void set_opp_3();
  void set_opp_4();
  void set_opp_5();
  void set_opp_6();
  void calc_pos(const hf_array<double> &in_loc, int in_ele, hf_array<double> &out_pos);
  void calc_d_pos(const hf_array<double> &in_loc, int in_ele, hf_array<double> &out_d_pos);
  int set_transforms_pts(int which); // 0 solution points, 1 flux points, 2 over-integration cubature points
  void fail(const std::string &msg);

  hfx_eles *dev = nullptr;
  std::string err;
  int max_n_spts_per_ele = 0;
};

class eles_hexas : public eles
{
protected:
  int setup_ele_type_specific() override;
  double eval_nodal_basis(int in_index, const hf_array<double> &in_loc) override;
  double eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &in_loc) override;
  void fill_opp_3(hf_array<double> &opp_3) override;
  double eval_nodal_s_basis(int in_index, const hf_array<double> &in_loc, int in_n_spts) override;
  void eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &in_loc, int in_n_spts) override;
};

class eles_quads : public eles
{
protected:
  int setup_ele_type_specific() override;
  double eval_nodal_basis(int in_index, const hf_array<double> &in_loc) override;
  double eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &in_loc) override;
  void fill_opp_3(hf_array<double> &opp_3) override;
  double eval_nodal_s_basis(int in_index, const hf_array<double> &in_loc, int in_n_spts) override;
  void eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &in_loc, int in_n_spts) override;
};

// Tetrahedra and triangular prisms (src/eles_tets.cpp, src/eles_pris.cpp) as producers of the dense operators and the
// metrics: csrc/host/eles_simplex.cpp.  4-node / 6-node (straight-sided) shapes.
class eles_tets : public eles
{
protected:
  int setup_ele_type_specific() override;
  double eval_nodal_basis(int in_index, const hf_array<double> &in_loc) override;
  double eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &in_loc) override;
  void fill_opp_3(hf_array<double> &opp_3) override;
  double eval_nodal_s_basis(int in_index, const hf_array<double> &in_loc, int in_n_spts) override;
  void eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &in_loc, int in_n_spts) override;
  int compute_filter_upts() override; // src/eles_tets.cpp:576-690 (modal filter, element average)
  double calc_h_ref_specific(int in_ele) override; // src/eles_tets.cpp:1599-1633: the insphere diameter
  int set_shock_capture_operators() override; // src/eles_tets.cpp:705-797 (set_vandermonde, set_exp_filter, shock_det_persson's mode set)
  std::vector<int> mode_i, mode_j, mode_k; // the orthonormal modal basis the nodal one is computed through
  std::vector<double> vinv;                // inverse Vandermonde matrix, row-major
};

class eles_pris : public eles
{
public:
  eles_pris();
  ~eles_pris() override;
  int n_upts_tri = 0;

protected:
  int setup_ele_type_specific() override;
  double eval_nodal_basis(int in_index, const hf_array<double> &in_loc) override;
  double eval_d_nodal_basis(int in_index, int in_cpnt, const hf_array<double> &in_loc) override;
  void fill_opp_3(hf_array<double> &opp_3) override;
  double eval_nodal_s_basis(int in_index, const hf_array<double> &in_loc, int in_n_spts) override;
  void eval_d_nodal_s_basis(hf_array<double> &d, const hf_array<double> &in_loc, int in_n_spts) override;
  int set_shock_capture_operators() override; // src/eles_pris.cpp:609-730 (set_vandermonde3D, set_exp_filter, calc_norm_basis)
  double calc_h_ref_specific(int in_ele) override; // src/eles_pris.cpp:1535-1557: shortest vertical edge or triangle incircle diameter
  struct Impl;
  Impl *impl;
};
