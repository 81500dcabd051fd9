// Frame.h -- layout-compatible subset of the reference's common/Frame.h: the rigid transform
// record that the ultrasound-calibration data types embed (rotation[3][3], translation[3], int
// outputFormat: 104 bytes, common/Frame.h:30-31,41).  Only what the hot path reads is provided
// (set/get of the matrix and translation, ZYX Euler composition, common/Frame.cxx:87-113); the
// quaternion / interpolation utilities are out of scope.
#ifndef _FRAME_H_
#define _FRAME_H_

#include <cmath>

namespace lsqrRecipes {

class Frame {
 private:
  double rotation[3][3];
  double translation[3];

 public:
  enum { MATRIX = 0, EULER_ANGLES, AXIS_ANGLE, QUATERNION };
  int outputFormat;

  Frame() : outputFormat(MATRIX) { setIdentity(); }
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
