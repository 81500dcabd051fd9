#ifndef _EPSILON_H_
#define _EPSILON_H_
namespace lsqrRecipes {
// value of the reference's common/Epsilon.h:19 (DBL_EPSILON)
const double EPS = 2.220446049250313e-016;
}  // namespace lsqrRecipes
#endif
