// DenseLinearEquationSystemParametersEstimator.h -- drop-in for the reference's header of the same
// name: AugmentedRow<T,n> (one equation a^T x = b, laid out aValues[n], bValue) and the estimator of
// x for Ax = b with outliers.  T must be double (the reference only instantiates for double as well,
// DenseLinear...Estimator.hxx:38).  n = 1..64.  getAugmentedRows takes plain arrays instead of
// vnl_matrix / vnl_vector (VNL is not a dependency of the drop-in).
#ifndef _DENSE_LINEAR_EQUATION_SYSTEM_PARAMETERS_ESTIMATOR_H_
#define _DENSE_LINEAR_EQUATION_SYSTEM_PARAMETERS_ESTIMATOR_H_

#include <cstring>
#include <exception>
#include <ostream>

#include "LsqrDevice.h"
#include "ParametersEstimator.h"

namespace lsqrRecipes {

template <class T, unsigned int n>
class AugmentedRow {
 public:
  enum { dimension = n };
  AugmentedRow() {
    std::memset(aValues, 0, n * sizeof(T));
    bValue = static_cast<T>(0.0);
  }
  AugmentedRow(T *fillData) { set(fillData); }
  AugmentedRow(T *fillData, T bData) { set(fillData, bData); }
  AugmentedRow(const AugmentedRow<T, n> &o) {
    std::memcpy(aValues, o.aValues, n * sizeof(T));
    bValue = o.bValue;
  }
  AugmentedRow<T, n> &operator=(const AugmentedRow<T, n> &o) {
    std::memcpy(aValues, o.aValues, n * sizeof(T));
    bValue = o.bValue;
    return *this;
  }
  T &operator[](unsigned int i) { return i == n ? bValue : aValues[i]; }
  const T &operator[](unsigned int i) const { return i == n ? bValue : aValues[i]; }
  void set(T *fillData) {
    std::memcpy(aValues, fillData, n * sizeof(T));
    bValue = fillData[n];
  }
  void set(T *fillData, T bData) {
    std::memcpy(aValues, fillData, n * sizeof(T));
    bValue = bData;
  }
  void get(T *a, T &b) {
    std::memcpy(a, aValues, n * sizeof(T));
    b = bValue;
  }
  void get(T *a) {
    std::memcpy(a, aValues, n * sizeof(T));
    a[n] = bValue;
  }
  unsigned int size() { return n + 1; }
  friend std::ostream &operator<<(std::ostream &out, const AugmentedRow &r) {
    out << "[ ";
    for (unsigned int i = 0; i < n; i++) out << r.aValues[i] << ", ";
    return out << r.bValue << " ]";
  }

 private:
  T aValues[n];
  T bValue;
};

template <class T, unsigned int n>
class DenseLinearEquationSystemParametersEstimator
    : public ParametersEstimator<AugmentedRow<T, n>, T> {
  typedef AugmentedRow<T, n> RowT;
  static_assert(sizeof(T) == sizeof(double), "the device path is fp64 (as the reference's solver)");
  static_assert(n >= 1 && n <= 64, "the device model supports n = 1..64");

 public:
  DenseLinearEquationSystemParametersEstimator(T delta)
      : ParametersEstimator<RowT, T>(n), delta(delta) {}

  virtual void estimate(std::vector<RowT *> &data, std::vector<T> &parameters) {
    std::vector<RowT> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<RowT> &data, std::vector<T> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<RowT *> &data, std::vector<T> &parameters) {
    std::vector<RowT> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<RowT> &data, std::vector<T> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::lsFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<T> &parameters, RowT &data) {
    return detail::agreeOne(cfg(), parameters, data);
  }
  void setDelta(T d) { this->delta = d; }

  // A: rowNum x n row-major, b: rowNum
  static void getAugmentedRows(const T *A, const T *b, unsigned int rowNum, std::vector<RowT> &rows) {
    if (!A || !b) throw std::exception();
    rows.resize(rowNum);
    for (unsigned int i = 0; i < rowNum; i++) rows[i].set(const_cast<T *>(A + (size_t)i * n), b[i]);
  }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg();
    return true;
  }

 private:
  lsqr_model_cfg cfg() const {
    lsqr_model_cfg c = {LSQR_MODEL_DENSE, (int32_t)n, (double)delta, 0, 0};
    return c;
  }
  double delta;
};

}  // namespace lsqrRecipes
#endif
