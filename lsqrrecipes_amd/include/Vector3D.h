// Vector3D.h -- minimal VNL-free counterpart of the reference's common/Vector3D.h / Vector.h: three
// doubles with element access, normalisation and the products the ray example needs.  Layout: 24 bytes.
#ifndef _VECTOR3D_H_
#define _VECTOR3D_H_

#include <cmath>
#include <ostream>

namespace lsqrRecipes {

class Vector3D {
 public:
  enum { dimension = 3 };
  Vector3D() { data[0] = data[1] = data[2] = 0.0; }
  Vector3D(double x, double y, double z) { data[0] = x; data[1] = y; data[2] = z; }
  double &operator[](int i) { return data[i]; }
  const double &operator[](int i) const { return data[i]; }
  double l2Norm() const { return std::sqrt(data[0] * data[0] + data[1] * data[1] + data[2] * data[2]); }
  void normalize() {
    const double n = l2Norm();
    if (n != 0) { data[0] /= n; data[1] /= n; data[2] /= n; }
  }
  friend std::ostream &operator<<(std::ostream &o, const Vector3D &v) {
    return o << "[" << v.data[0] << "," << v.data[1] << "," << v.data[2] << "]";
  }

 private:
  double data[3];
};

}  // namespace lsqrRecipes
#endif
