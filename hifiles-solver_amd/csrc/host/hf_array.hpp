// hf_array.hpp -- owning <=4-D column-major array with the reference's layout contract
// (/root/reference/include/hf_array.h:303-325): a(i,j,k,l) = data[i + d0*(j + d1*(k + d2*l))].
// Own implementation (std::vector storage, value semantics); the device mirror of an array
// lives in libhfx, reached through hfx_eles_upload / hfx_eles_download instead of the
// reference's cp_cpu_gpu / cp_gpu_cpu members.
#pragma once
#include <cstddef>
#include <vector>

template <typename T>
class hf_array
{
public:
  hf_array() { setup(0); }
  explicit hf_array(int d0, int d1 = 1, int d2 = 1, int d3 = 1) { setup(d0, d1, d2, d3); }

  void setup(int d0, int d1 = 1, int d2 = 1, int d3 = 1)
  {
    dim_[0] = d0; dim_[1] = d1; dim_[2] = d2; dim_[3] = d3;
    data_.assign((size_t)d0 * d1 * d2 * d3, T());
  }
  T &operator()(int i) { return data_[i]; }
  T &operator()(int i, int j) { return data_[i + (size_t)dim_[0] * j]; }
  T &operator()(int i, int j, int k) { return data_[i + (size_t)dim_[0] * (j + (size_t)dim_[1] * k)]; }
  T &operator()(int i, int j, int k, int l)
  {
    return data_[i + (size_t)dim_[0] * (j + (size_t)dim_[1] * (k + (size_t)dim_[2] * l))];
  }
  const T &operator()(int i) const { return data_[i]; }
  const T &operator()(int i, int j) const { return data_[i + (size_t)dim_[0] * j]; }
  const T &operator()(int i, int j, int k) const { return data_[i + (size_t)dim_[0] * (j + (size_t)dim_[1] * k)]; }
  const T &operator()(int i, int j, int k, int l) const
  {
    return data_[i + (size_t)dim_[0] * (j + (size_t)dim_[1] * (k + (size_t)dim_[2] * l))];
  }
  T &operator[](size_t q) { return data_[q]; }
  T *get_ptr_cpu() { return data_.data(); }
  const T *get_ptr_cpu() const { return data_.data(); }
  T *get_ptr_cpu(int i, int j = 0, int k = 0, int l = 0) { return &(*this)(i, j, k, l); }
  int get_dim(int d) const { return dim_[d]; }
  size_t size() const { return data_.size(); }
  void initialize_to_zero() { data_.assign(data_.size(), T()); }
  void initialize_to_value(const T v) { data_.assign(data_.size(), v); }

private:
  int dim_[4];
  std::vector<T> data_;
};
