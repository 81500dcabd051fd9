// Ray3D.h -- layout-compatible counterpart of the reference's common/Ray3D.h:20-24: origin p and
// direction n (assumed unit length by the estimator), r(t) = p + t n, t >= 0.  48 bytes.
#ifndef _RAY3D_H_
#define _RAY3D_H_

#include <cmath>
#include <ostream>

#include "Point3D.h"
#include "Vector3D.h"

namespace lsqrRecipes {

class Ray3D {
 public:
  Point3D p;
  Vector3D n;
  Ray3D() {}
  // distance between a point and the line of the ray (common/Ray3D.h:36-60)
  double distance(const Point3D &pnt) const {
    double t = 0, d2 = 0;
    for (int i = 0; i < 3; i++) t += n[i] * (pnt[i] - p[i]);
    for (int i = 0; i < 3; i++) {
      const double e = pnt[i] - p[i] - t * n[i];
      d2 += e * e;
    }
    return std::sqrt(d2);
  }
  friend std::ostream &operator<<(std::ostream &o, const Ray3D &r) {
    return o << "p: " << r.p << " n: " << r.n;
  }
};

}  // namespace lsqrRecipes
#endif
