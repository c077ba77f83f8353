// se3.h -- SE3 / SO3 device math for the BA and geometry kernels (gfx950).
//
// Conventions follow the reference kernels (src/droid_kernels.cu:66-188, 994-1012):
// pose = [tx,ty,tz, qx,qy,qz,qw] mapping world -> camera, quaternion [x,y,z,w], homogeneous
// points [X,Y,Z,h] with h = inverse depth.
#pragma once
#include <hip/hip_runtime.h>

#define NS_MIN_DEPTH 0.25f  // droid_kernels.cu:26

namespace se3 {

template <typename T>
__host__ __device__ __forceinline__ void act_so3(const T* q, const T* X, T* Y) {
  const T uvx = T(2) * (q[1] * X[2] - q[2] * X[1]);
  const T uvy = T(2) * (q[2] * X[0] - q[0] * X[2]);
  const T uvz = T(2) * (q[0] * X[1] - q[1] * X[0]);
  const T y0 = X[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  const T y1 = X[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  const T y2 = X[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
  Y[0] = y0;
  Y[1] = y1;
  Y[2] = y2;
}

// Y = [R X + h t, h]
template <typename T>
__host__ __device__ __forceinline__ void act_se3(const T* t, const T* q, const T* X, T* Y) {
  const T h = X[3];
  act_so3(q, X, Y);
  Y[3] = h;
  Y[0] += h * t[0];
  Y[1] += h * t[1];
  Y[2] += h * t[2];
}

// Row-vector times adjoint, exactly as adjSE3 of the reference (droid_kernels.cu:88-105),
// INCLUDING its behaviour when X and Y alias: the cross product then sees the already rotated
// X[0..2].  `aliased` selects that behaviour explicitly (the reference calls it in place for the
// camera-to-body adjoint, :380-381).
template <typename T>
__host__ __device__ __forceinline__ void adj_se3(const T* t, const T* q, const T* X, T* Y, bool aliased) {
  const T qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  T a[3], b[3];
  act_so3(qinv, &X[0], a);
  act_so3(qinv, &X[3], b);
  const T* Xc = aliased ? a : X;
  T u[3], v[3];
  u[0] = t[2] * Xc[1] - t[1] * Xc[2];
  u[1] = t[0] * Xc[2] - t[2] * Xc[0];
  u[2] = t[1] * Xc[0] - t[0] * Xc[1];
  act_so3(qinv, u, v);
  Y[0] = a[0];
  Y[1] = a[1];
  Y[2] = a[2];
  Y[3] = b[0] + v[0];
  Y[4] = b[1] + v[1];
  Y[5] = b[2] + v[2];
}

// G_ij = G_j * G_i^-1   (droid_kernels.cu:107-120)
template <typename T>
__host__ __device__ __forceinline__ void rel_se3(const T* ti, const T* qi, const T* tj, const T* qj, T* tij,
                                                 T* qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  T r[3];
  act_so3(qij, ti, r);
  tij[0] = tj[0] - r[0];
  tij[1] = tj[1] - r[1];
  tij[2] = tj[2] - r[2];
}

// Hamilton product a*b
template <typename T>
__host__ __device__ __forceinline__ void qmul(const T* a, const T* b, T* o) {
  const T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const T y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  const T z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  const T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x;
  o[1] = y;
  o[2] = z;
  o[3] = w;
}

// (ta,qa)*(tb,qb): x -> Ra (Rb x + tb) + ta
template <typename T>
__host__ __device__ __forceinline__ void mul(const T* a, const T* b, T* o) {
  T t[3], q[4];
  act_so3(a + 3, b, t);
  qmul(a + 3, b + 3, q);
  o[0] = t[0] + a[0];
  o[1] = t[1] + a[1];
  o[2] = t[2] + a[2];
  o[3] = q[0];
  o[4] = q[1];
  o[5] = q[2];
  o[6] = q[3];
}

template <typename T>
__host__ __device__ __forceinline__ void inv(const T* a, T* o) {
  const T qi[4] = {-a[3], -a[4], -a[5], a[6]};
  T t[3];
  act_so3(qi, a, t);
  o[0] = -t[0];
  o[1] = -t[1];
  o[2] = -t[2];
  o[3] = qi[0];
  o[4] = qi[1];
  o[5] = qi[2];
  o[6] = qi[3];
}

// SE3 exponential, xi = [omega(3), v(3)] (GTSAM Pose3 order). out = [t, q].
__device__ __forceinline__ void exp_wv(const double* xi, double* out) {
  const double w0 = xi[0], w1 = xi[1], w2 = xi[2];
  const double v[3] = {xi[3], xi[4], xi[5]};
  const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
  const double th = sqrt(th2);
  double a, b, imag, real;
  if (th < 1e-10) {
    imag = 0.5;
    real = 1.0;
    a = 0.5;
    b = 1.0 / 6.0;
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
    a = (1.0 - cos(th)) / th2;
    b = (th - sin(th)) / (th * th2);
  }
  double q[4] = {imag * w0, imag * w1, imag * w2, real};
  const double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double wv[3] = {w1 * v[2] - w2 * v[1], w2 * v[0] - w0 * v[2], w0 * v[1] - w1 * v[0]};
  const double wwv[3] = {w1 * wv[2] - w2 * wv[1], w2 * wv[0] - w0 * wv[2], w0 * wv[1] - w1 * wv[0]};
  out[0] = v[0] + a * wv[0] + b * wwv[0];
  out[1] = v[1] + a * wv[1] + b * wwv[1];
  out[2] = v[2] + a * wv[2] + b * wwv[2];
  out[3] = q[0] * qn;
  out[4] = q[1] * qn;
  out[5] = q[2] * qn;
  out[6] = q[3] * qn;
}

// inverse of exp_wv: pose [t,q] -> [omega, v]
__device__ __forceinline__ void log_wv(const double* p, double* xi) {
  double q[4] = {p[3], p[4], p[5], p[6]};
  const double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double s = (q[3] < 0 ? -qn : qn);
  q[0] *= s;
  q[1] *= s;
  q[2] *= s;
  q[3] *= s;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const double th = 2.0 * atan2(n, q[3]);
  const double k = (n < 1e-12) ? 2.0 : th / n;
  const double w[3] = {q[0] * k, q[1] * k, q[2] * k};
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t1 = sqrt(t2);
  // V^-1 = I - 1/2 W + c W^2
  const double c = (t1 < 1e-10) ? (1.0 / 12.0) : (1.0 / t2 - (1.0 + cos(t1)) / (2.0 * t1 * sin(t1)));
  const double* t = p;
  const double wt[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
  const double wwt[3] = {w[1] * wt[2] - w[2] * wt[1], w[2] * wt[0] - w[0] * wt[2], w[0] * wt[1] - w[1] * wt[0]};
  xi[0] = w[0];
  xi[1] = w[1];
  xi[2] = w[2];
  xi[3] = t[0] - 0.5 * wt[0] + c * wwt[0];
  xi[4] = t[1] - 0.5 * wt[1] + c * wwt[1];
  xi[5] = t[2] - 0.5 * wt[2] + c * wwt[2];
}

}  // namespace se3
