// Point.h -- drop-in for the reference's common/Point.h: fixed-size POD point, T data[n], so that
// std::vector<Point<double,n>> has exactly the reference's memory layout (8*n bytes per record,
// common/Point.h:127) and can be handed to the device without repacking.  VNL-free.
#ifndef _POINT_H_
#define _POINT_H_

#include <cstring>
#include <ostream>

namespace lsqrRecipes {

template <class T, unsigned int n>
class Point {
 public:
  enum { dimension = n };

  Point() { std::memset(data, 0, n * sizeof(T)); }
  Point(T *fillData) { std::memcpy(data, fillData, n * sizeof(T)); }
  Point(const Point<T, n> &other) { std::memcpy(data, other.data, n * sizeof(T)); }
  Point<T, n> &operator=(const Point<T, n> &other) {
    std::memcpy(data, other.data, n * sizeof(T));
    return *this;
  }
  T &operator[](int index) { return data[index]; }
  const T &operator[](int index) const { return data[index]; }
  void set(T *fillData) { std::memcpy(data, fillData, n * sizeof(T)); }
  unsigned int size() { return n; }
  double distanceSquared(const Point<T, n> &other) const {
    double s = 0;
    for (unsigned int i = 0; i < n; i++) s += (data[i] - other.data[i]) * (data[i] - other.data[i]);
    return s;
  }
  friend std::ostream &operator<<(std::ostream &output, const Point &p) {
    output << "[ " << p.data[0];
    for (unsigned int i = 1; i < n; i++) output << ", " << p.data[i];
    return output << " ]";
  }

 private:
  T data[n];
};

}  // namespace lsqrRecipes
#endif
