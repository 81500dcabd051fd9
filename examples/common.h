// Shared helpers of the example programs (seeded data synthesis; nothing here is on the hot path).
#ifndef LSQR_EXAMPLES_COMMON_H
#define LSQR_EXAMPLES_COMMON_H
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

struct Rng {
  std::mt19937_64 g;
  explicit Rng(unsigned long long seed) : g(seed) {}
  double uniform(double a = 0.0, double b = 1.0) {
    return std::uniform_real_distribution<double>(a, b)(g);
  }
  double normal(double sigma) { return std::normal_distribution<double>(0.0, sigma)(g); }
};

inline void printVec(const char *label, const std::vector<double> &v) {
  std::printf("%s\n\t [ ", label);
  for (size_t i = 0; i < v.size(); i++) std::printf("%.10g%s", v[i], i + 1 < v.size() ? ", " : "");
  std::printf(" ]\n\n");
}
#endif
