// eigen_shim.h -- NOT Eigen.  A syntax-only stand-in for the slice of the Eigen 3 API that the reference's interface headers and
// our adapters touch, so that tests/test_adapter_syntax_cpu.py can put ekf_slam_adapter.hpp / detect_adapter.hpp through a
// compiler (-fsyntax-only) against the REAL reference headers on an image that has no Eigen.  Nothing here computes anything
// meaningful and nothing links against it: it checks the boundary's spelling (names, signatures, overrides), never numbers.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>
namespace Eigen {
const int Dynamic = -1;
template <class T, int R, int C> struct ShimStorage {
    T d_[R * C];
    ShimStorage() : d_() {}
    void resize(int, int) {}
    int rows() const { return R; }
    int cols() const { return C; }
    T *ptr() { return d_; }
    const T *ptr() const { return d_; }
};
template <class T> struct ShimDyn {
    std::vector<T> v_;
    int r_ = 0, c_ = 0;
    void resize(int r, int c) { r_ = r; c_ = c; v_.resize((size_t)r * c); }
    int rows() const { return r_; }
    int cols() const { return c_; }
    T *ptr() { return v_.data(); }
    const T *ptr() const { return v_.data(); }
};
template <class T, int C> struct ShimStorage<T, Dynamic, C> : ShimDyn<T> {};
template <class T> struct ShimStorage<T, Dynamic, Dynamic> : ShimDyn<T> {};
template <class T, int R> struct ShimStorage<T, R, Dynamic> : ShimDyn<T> {};

template <class M> struct Map;
template <class M> struct CommaInit {
    M &m_;
    int k_;
    template <class S> CommaInit &operator,(const S &s) { m_.data()[k_++] = (typename M::Scalar)s; return *this; }
};

template <class T, int R, int C> class Matrix : public ShimStorage<T, R, C> {
public:
    typedef T Scalar;
    Matrix() {}
    template <class A, class B> Matrix(const A &a, const B &b) { this->resize(2, 1); data()[0] = (T)a; data()[1] = (T)b; }
    template <class A, class B, class D> Matrix(const A &a, const B &b, const D &c) { this->resize(3, 1); data()[0] = (T)a; data()[1] = (T)b; data()[2] = (T)c; }
    Matrix(const Map<Matrix> &m);
    Matrix &operator=(const Map<Matrix> &m);
    T *data() { return this->ptr(); }
    const T *data() const { return this->ptr(); }
    int size() const { return this->rows() * this->cols(); }
    T &x() { return data()[0]; }
    T &y() { return data()[1]; }
    T &z() { return data()[2]; }
    T &w() { return data()[3]; }
    const T &x() const { return data()[0]; }
    const T &y() const { return data()[1]; }
    const T &z() const { return data()[2]; }
    const T &w() const { return data()[3]; }
    T &operator()(int i) { return data()[i]; }
    const T &operator()(int i) const { return data()[i]; }
    T &operator()(int i, int j) { return data()[i + j * this->rows()]; }
    const T &operator()(int i, int j) const { return data()[i + j * this->rows()]; }
    T &operator[](int i) { return data()[i]; }
    const T &operator[](int i) const { return data()[i]; }
    template <class U> Matrix<U, R, C> cast() const { return Matrix<U, R, C>(); }
    template <int N> Matrix<T, N, 1> head() const { return Matrix<T, N, 1>(); }
    template <int N> Matrix<T, N, 1> tail() const { return Matrix<T, N, 1>(); }
    T norm() const { return T(); }
    T squaredNorm() const { return T(); }
    T dot(const Matrix &) const { return T(); }
    Matrix normalized() const { return *this; }
    Matrix<T, C, R> transpose() const { return Matrix<T, C, R>(); }
    Matrix inverse() const { return *this; }
    static Matrix Identity() { return Matrix(); }
    static Matrix Identity(int, int) { return Matrix(); }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int) { return Matrix(); }
    static Matrix Zero(int, int) { return Matrix(); }
    static Matrix UnitX() { return Matrix(); }
    static Matrix UnitY() { return Matrix(); }
    static Matrix UnitZ() { return Matrix(); }
    Matrix operator+(const Matrix &) const { return *this; }
    Matrix operator-(const Matrix &) const { return *this; }
    Matrix operator-() const { return *this; }
    Matrix &operator+=(const Matrix &) { return *this; }
    Matrix &operator-=(const Matrix &) { return *this; }
    Matrix operator*(const T &) const { return *this; }
    Matrix operator/(const T &) const { return *this; }
    template <int C2> Matrix<T, R, C2> operator*(const Matrix<T, C, C2> &) const { return Matrix<T, R, C2>(); }
    template <class S> CommaInit<Matrix> operator<<(const S &s) { data()[0] = (T)s; return CommaInit<Matrix>{*this, 1}; }
};
template <class T, int R, int C> Matrix<T, R, C> operator*(const T &, const Matrix<T, R, C> &m) { return m; }

template <class M> struct Map {
    typedef typename M::Scalar Scalar;
    const Scalar *p_;
    int r_, c_;
    Map(const Scalar *p) : p_(p), r_(0), c_(0) {}
    Map(const Scalar *p, int n) : p_(p), r_(n), c_(1) {}
    Map(const Scalar *p, int r, int c) : p_(p), r_(r), c_(c) {}
};
template <class T, int R, int C> Matrix<T, R, C>::Matrix(const Map<Matrix> &) {}
template <class T, int R, int C> Matrix<T, R, C> &Matrix<T, R, C>::operator=(const Map<Matrix> &) { return *this; }

template <class T> class AngleAxis {
public:
    AngleAxis() {}
    AngleAxis(const T &, const Matrix<T, 3, 1> &) {}
    T angle() const { return T(); }
    Matrix<T, 3, 1> axis() const { return Matrix<T, 3, 1>(); }
};
template <class T> class Quaternion {
    T q_[4];
public:
    Quaternion() : q_() {}
    Quaternion(const T &w, const T &x, const T &y, const T &z) { q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w; }
    Quaternion(const AngleAxis<T> &) : q_() {}
    T &x() { return q_[0]; }
    T &y() { return q_[1]; }
    T &z() { return q_[2]; }
    T &w() { return q_[3]; }
    const T &x() const { return q_[0]; }
    const T &y() const { return q_[1]; }
    const T &z() const { return q_[2]; }
    const T &w() const { return q_[3]; }
    Matrix<T, 3, 1> vec() const { return Matrix<T, 3, 1>(); }
    Quaternion normalized() const { return *this; }
    Quaternion conjugate() const { return *this; }
    Quaternion inverse() const { return *this; }
    T norm() const { return T(); }
    T angularDistance(const Quaternion &) const { return T(); }
    Quaternion slerp(const T &, const Quaternion &) const { return *this; }
    Matrix<T, 3, 3> toRotationMatrix() const { return Matrix<T, 3, 3>(); }
    template <class U> Quaternion<U> cast() const { return Quaternion<U>(); }
    static Quaternion Identity() { return Quaternion(); }
    Quaternion operator*(const Quaternion &) const { return *this; }
    Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1> &v) const { return v; }
};
template <class T> class Rotation2D {
    T a_;
public:
    Rotation2D() : a_() {}
    Rotation2D(const T &a) : a_(a) {}
    T angle() const { return a_; }
    T smallestAngle() const { return a_; }
    Rotation2D inverse() const { return *this; }
    template <class U> Rotation2D<U> cast() const { return Rotation2D<U>(); }
    static Rotation2D Identity() { return Rotation2D(); }
    Rotation2D operator*(const Rotation2D &) const { return *this; }
    Matrix<T, 2, 1> operator*(const Matrix<T, 2, 1> &v) const { return v; }
    Matrix<T, 2, 2> toRotationMatrix() const { return Matrix<T, 2, 2>(); }
};
template <class T, int R, int C> using Array = Matrix<T, R, C>;
typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<int, 2, 1> Array2i;
typedef Matrix<int, 3, 1> Array3i;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<float, 2, 2> Matrix2f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;
typedef AngleAxis<double> AngleAxisd;
typedef AngleAxis<float> AngleAxisf;
typedef Rotation2D<double> Rotation2Dd;
typedef Rotation2D<float> Rotation2Df;
}  // namespace Eigen
