#ifndef _POINT2D_H_
#define _POINT2D_H_
#include "Point.h"
namespace lsqrRecipes {
typedef Point<double, 2> Point2D;
}
#endif
