// Frame.h -- layout-compatible subset of the reference's common/Frame.h: the rigid transform
// record that the ultrasound-calibration data types embed (rotation[3][3], translation[3], int
// outputFormat: 104 bytes, common/Frame.h:30-31,41).  Only what the hot path reads is provided
// (set/get of the matrix and translation, ZYX Euler composition, common/Frame.cxx:87-113); the
// interpolation utilities are out of scope.  The unit-quaternion
// constructor / setter / getter and apply() are provided for the AbsoluteOrientation and
// PivotCalibration estimators (common/Frame.cxx:174-198,208-247,750-771,952-991).
#ifndef _FRAME_H_
#define _FRAME_H_

#include <cmath>

#include "Point3D.h"

namespace lsqrRecipes {

class Frame {
 private:
  double rotation[3][3];
  double translation[3];

 public:
  enum { MATRIX = 0, EULER_ANGLES, AXIS_ANGLE, QUATERNION };
  int outputFormat;

  Frame() : outputFormat(MATRIX) { setIdentity(); }
  // translation + unit quaternion [s, qx, qy, qz] (common/Frame.h:127)
  Frame(double x, double y, double z, double s, double qx, double qy, double qz,
        bool normalizeQuaternion = false)
      : outputFormat(MATRIX) {
    setTranslation(x, y, z);
    setRotationQuaternion(s, qx, qy, qz, normalizeQuaternion);
  }
  void setRotationQuaternion(double s, double qx, double qy, double qz,
                             bool normalizeQuaternion = false) {
    if (normalizeQuaternion) {
      const double norm = std::sqrt(s * s + qx * qx + qy * qy + qz * qz);
      s /= norm; qx /= norm; qy /= norm; qz /= norm;
    }
    rotation[0][0] = 1 - 2 * (qy * qy + qz * qz);
    rotation[0][1] = 2 * (qx * qy - s * qz);
    rotation[0][2] = 2 * (qx * qz + s * qy);
    rotation[1][0] = 2 * (qx * qy + s * qz);
    rotation[1][1] = 1 - 2 * (qx * qx + qz * qz);
    rotation[1][2] = 2 * (qy * qz - s * qx);
    rotation[2][0] = 2 * (qx * qz - s * qy);
    rotation[2][1] = 2 * (qy * qz + s * qx);
    rotation[2][2] = 1 - 2 * (qx * qx + qy * qy);
  }
  // [s, qx, qy, qz]; the vector part is stabilised when the half angle is within 0.5 degrees of 90
  void getRotationQuaternion(double q[4]) const {
    const double small = 0.008726535498373935, halfPi = 3.14159265358979323846 / 2.0;
    q[0] = 0.5 * std::sqrt(rotation[0][0] + rotation[1][1] + rotation[2][2] + 1);
    const double halfTheta = std::acos(q[0]);
    if (!(halfTheta > halfPi - small && halfTheta < halfPi + small)) {
      const double denom = 4 * q[0];
      q[1] = (rotation[2][1] - rotation[1][2]) / denom;
      q[2] = (rotation[0][2] - rotation[2][0]) / denom;
      q[3] = (rotation[1][0] - rotation[0][1]) / denom;
    } else {
      int i = 0;
      if (rotation[1][1] > rotation[i][i]) i = 1;
      if (rotation[2][2] > rotation[i][i]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      const double w = std::sqrt(rotation[i][i] - rotation[j][j] - rotation[k][k] + 1);
      q[i + 1] = w / 2.0;
      q[j + 1] = (rotation[i][j] + rotation[j][i]) / (2 * w);
      q[k + 1] = (rotation[i][k] + rotation[k][i]) / (2 * w);
    }
  }
  void apply(const Point3D &p, Point3D &out) const {
    double in[3] = {p[0], p[1], p[2]}, o[3];
    apply(in, o);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
  }
  void apply(Point3D &p) const { apply(p, p); }
  void setIdentity() {
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) rotation[i][j] = (i == j) ? 1.0 : 0.0;
      translation[i] = 0.0;
    }
  }
  void setRotationMatrix(const double R[3][3]) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) rotation[i][j] = R[i][j];
  }
  void setRotationMatrix(double m00, double m01, double m02, double m10, double m11, double m12,
                         double m20, double m21, double m22) {
    rotation[0][0] = m00; rotation[0][1] = m01; rotation[0][2] = m02;
    rotation[1][0] = m10; rotation[1][1] = m11; rotation[1][2] = m12;
    rotation[2][0] = m20; rotation[2][1] = m21; rotation[2][2] = m22;
  }
  // R = Rz(az) * Ry(ay) * Rx(ax)
  void setRotationEulerAngles(double ax, double ay, double az) {
    double cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay),
           cz = std::cos(az), sz = std::sin(az);
    rotation[0][0] = cz * cy; rotation[0][1] = cz * sy * sx - sz * cx; rotation[0][2] = cz * sy * cx + sz * sx;
    rotation[1][0] = sz * cy; rotation[1][1] = sz * sy * sx + cz * cx; rotation[1][2] = sz * sy * cx - cz * sx;
    rotation[2][0] = -sy;     rotation[2][1] = cy * sx;                rotation[2][2] = cy * cx;
  }
  void setTranslation(const double t[3]) {
    translation[0] = t[0]; translation[1] = t[1]; translation[2] = t[2];
  }
  void setTranslation(double x, double y, double z) {
    translation[0] = x; translation[1] = y; translation[2] = z;
  }
  void getRotationMatrix(double R[3][3]) const {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = rotation[i][j];
  }
  void getTranslation(double t[3]) const {
    t[0] = translation[0]; t[1] = translation[1]; t[2] = translation[2];
  }
  void getTranslation(double &x, double &y, double &z) const {
    x = translation[0]; y = translation[1]; z = translation[2];
  }
  void apply(const double in[3], double out[3]) const {
    for (int i = 0; i < 3; i++)
      out[i] = rotation[i][0] * in[0] + rotation[i][1] * in[1] + rotation[i][2] * in[2] + translation[i];
  }
};

}  // namespace lsqrRecipes
#endif
