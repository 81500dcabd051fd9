#ifndef _POINT3D_H_
#define _POINT3D_H_
#include "Point.h"
namespace lsqrRecipes {
typedef Point<double, 3> Point3D;
}
#endif
