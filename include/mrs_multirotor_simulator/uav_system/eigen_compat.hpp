// eigen_compat.hpp — the reference's header API speaks Eigen (Vector3d, Matrix3d, VectorXd, MatrixXd).
// When Eigen3 is installed it is used as is; otherwise a minimal value-type subset with the same names and the
// members the UavSystem API surface needs (element access, Zero/Identity, size, data) is provided, so that the
// facade compiles on a bare ROCm box.  These types only carry values across the C ABI; no arithmetic happens here.
#ifndef MRS_EIGEN_COMPAT_HPP
#define MRS_EIGEN_COMPAT_HPP

#if defined(MRS_USE_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Dense>) && !defined(MRS_NO_EIGEN))
#include <Eigen/Dense>
#define MRS_HAVE_EIGEN 1
#else
#define MRS_HAVE_EIGEN 0
#include <cstddef>
#include <vector>

namespace Eigen
{

class Vector3d {
public:
  Vector3d() : v_{0, 0, 0} {}
  Vector3d(double x, double y, double z) : v_{x, y, z} {}
  static Vector3d Zero() { return Vector3d(); }
  static Vector3d Identity() { return Vector3d(1, 0, 0); }
  double&       operator()(int i) { return v_[i]; }
  const double& operator()(int i) const { return v_[i]; }
  double&       operator[](int i) { return v_[i]; }
  const double& operator[](int i) const { return v_[i]; }
  double        x() const { return v_[0]; }
  double        y() const { return v_[1]; }
  double        z() const { return v_[2]; }
  double*       data() { return v_; }
  const double* data() const { return v_; }
  int           size() const { return 3; }
  void          setZero() { v_[0] = v_[1] = v_[2] = 0; }
  Vector3d&     operator+=(const Vector3d& o) { v_[0] += o.v_[0]; v_[1] += o.v_[1]; v_[2] += o.v_[2]; return *this; }

private:
  double v_[3];
};

// column-major like Eigen's default, so that data() matches
class Matrix3d {
public:
  Matrix3d() : m_{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
  static Matrix3d Zero() { return Matrix3d(); }
  static Matrix3d Identity() {
    Matrix3d r;
    r(0, 0) = r(1, 1) = r(2, 2) = 1.0;
    return r;
  }
  double&       operator()(int r, int c) { return m_[c * 3 + r]; }
  const double& operator()(int r, int c) const { return m_[c * 3 + r]; }
  double*       data() { return m_; }
  const double* data() const { return m_; }
  Vector3d      col(int c) const { return Vector3d(m_[c * 3], m_[c * 3 + 1], m_[c * 3 + 2]); }
  int           rows() const { return 3; }
  int           cols() const { return 3; }

private:
  double m_[9];
};

class VectorXd {
public:
  VectorXd() {}
  explicit VectorXd(int n) : v_((size_t)n, 0.0) {}
  static VectorXd Zero(int n) { return VectorXd(n); }
  double&       operator()(int i) { return v_[(size_t)i]; }
  const double& operator()(int i) const { return v_[(size_t)i]; }
  double&       operator[](int i) { return v_[(size_t)i]; }
  const double& operator[](int i) const { return v_[(size_t)i]; }
  int           size() const { return (int)v_.size(); }
  void          resize(int n) { v_.assign((size_t)n, 0.0); }
  double*       data() { return v_.data(); }
  const double* data() const { return v_.data(); }

private:
  std::vector<double> v_;
};

// column-major dynamic matrix
class MatrixXd {
public:
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(int r, int c) : r_(r), c_(c), m_((size_t)r * c, 0.0) {}
  static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
  double&       operator()(int r, int c) { return m_[(size_t)c * r_ + r]; }
  const double& operator()(int r, int c) const { return m_[(size_t)c * r_ + r]; }
  int           rows() const { return r_; }
  int           cols() const { return c_; }
  double*       data() { return m_.data(); }
  const double* data() const { return m_.data(); }

private:
  int                 r_, c_;
  std::vector<double> m_;
};

}  // namespace Eigen
#endif  // Eigen present?
#endif  // MRS_EIGEN_COMPAT_HPP
