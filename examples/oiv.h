// oiv.h -- Open Inventor 2.1 ASCII scene files for the 3-D examples: every observation a small ball
// (green inside the consensus set, red outside) plus the estimated model (translucent quad, ball or
// line segment), what the reference's example mains save next to their console output
// (examples/planeEstimation.cxx:205-330, sphereEstimation.cxx:190-255, lineEstimation.cxx:225-390).
// Visualisation only; nothing here is on the hot path.
#ifndef LSQR_EXAMPLES_OIV_H
#define LSQR_EXAMPLES_OIV_H
#include <cmath>
#include <fstream>
#include <string>
#include <vector>

class OivScene {
 public:
  explicit OivScene(const std::string &fileName) : out(fileName.c_str()) {
    if (out.is_open()) out << "#Inventor V2.1 ascii\n\n";
    for (int j = 0; j < 3; j++) lo[j] = hi[j] = 0.0;
    count = 0;
  }
  bool good() const { return out.good(); }

  // one ball per observation; P is indexable with [0..2]
  template <class P>
  void observations(const std::vector<P> &pts, const std::vector<bool> &inConsensus, double radius) {
    for (size_t i = 0; i < pts.size(); i++) {
      const bool in = i < inConsensus.size() && inConsensus[i];
      begin(in ? "0.0 1.0 0.0" : "1.0 0.0 0.0", in ? "0.0 0.27 0.15" : "0.27 0.15 0.0",
            in ? "0.0 1.0 0.0" : "1.0 0.0 0.0", 0.0);
      ball(pts[i][0], pts[i][1], pts[i][2], radius);
      end();
      for (int j = 0; j < 3; j++) {
        const double v = pts[i][j];
        if (count == 0 || v < lo[j]) lo[j] = v;
        if (count == 0 || v > hi[j]) hi[j] = v;
      }
      count++;
    }
  }
  // largest coordinate range of the observations written so far
  double extent() const {
    double e = 0;
    for (int j = 0; j < 3; j++) e = std::fmax(e, hi[j] - lo[j]);
    return e;
  }

  // plane [n, a]: a square through a, spanning the data's bounding box diagonal
  void plane(const std::vector<double> &p) {
    const double half = std::sqrt(3.0) * extent() / 2.0;
    const double *n = &p[0], *a = &p[3];
    // any direction not parallel to n, made orthogonal to it
    int k = std::fabs(n[0]) <= std::fabs(n[1]) ? (std::fabs(n[0]) <= std::fabs(n[2]) ? 0 : 2)
                                               : (std::fabs(n[1]) <= std::fabs(n[2]) ? 1 : 2);
    double e1[3] = {0, 0, 0}, e2[3], d = n[k], len = 0;
    e1[k] = 1.0;
    for (int j = 0; j < 3; j++) {
      e1[j] -= d * n[j];
      len += e1[j] * e1[j];
    }
    for (int j = 0; j < 3; j++) e1[j] /= std::sqrt(len);
    e2[0] = n[1] * e1[2] - n[2] * e1[1];
    e2[1] = n[2] * e1[0] - n[0] * e1[2];
    e2[2] = n[0] * e1[1] - n[1] * e1[0];
    const double sgn[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
    begin("0.5 0.5 0.5", "0.2 0.2 0.2", "0.8 0.8 0.8", 0.4);
    out << "\tIndexedFaceSet {\n\t\tvertexProperty VertexProperty {\n\t\t\tvertex [ ";
    for (int c = 0; c < 4; c++) {
      for (int j = 0; j < 3; j++) out << a[j] + half * (sgn[c][0] * e1[j] + sgn[c][1] * e2[j]) << (j < 2 ? " " : "");
      out << (c < 3 ? ",\n\t\t\t         " : " ]\n");
    }
    out << "\t\t}\n\t\tcoordIndex [ 0, 1, 2, 3, -1 ]\n\t}\n";
    end();
  }
  // sphere [c, r]
  void sphere(const std::vector<double> &p) {
    begin("0.5 0.5 0.5", "0.2 0.2 0.2", "0.8 0.8 0.8", 0.4);
    ball(p[0], p[1], p[2], p[3]);
    end();
  }
  // line [direction, a]: the segment of the line inside the data's bounding range
  void line(const std::vector<double> &p) {
    const double half = std::sqrt(3.0) * extent() / 2.0;
    begin("0.5 0.5 0.5", "0.2 0.2 0.2", "0.8 0.8 0.8", 0.0);
    out << "\tCoordinate3 {\n\t\tpoint [ ";
    for (int s = -1; s <= 1; s += 2) {
      for (int j = 0; j < 3; j++) out << p[3 + j] + s * half * p[j] << (j < 2 ? " " : "");
      out << (s < 0 ? ", " : " ]\n");
    }
    out << "\t}\n\tLineSet {\n\t\tnumVertices [ 2 ]\n\t}\n";
    end();
  }

 private:
  void begin(const char *ambient, const char *diffuse, const char *specular, double transparency) {
    out << "Separator {\n\tMaterial {\n\t\tambientColor " << ambient << "\n\t\tdiffuseColor " << diffuse
        << "\n\t\tspecularColor " << specular << "\n";
    if (transparency > 0) out << "\t\ttransparency " << transparency << "\n";
    out << "\t}\n";
  }
  void ball(double x, double y, double z, double radius) {
    out << "\tTransform {\n\t\ttranslation " << x << " " << y << " " << z << "\n\t}\n";
    out << "\tSphere {\n\t\tradius " << radius << "\n\t}\n";
  }
  void end() { out << "}\n"; }

  std::ofstream out;
  double lo[3], hi[3];
  size_t count;
};

// consensus set of a model the way the reference's writers classify: estimator.agree() per observation
template <class Estimator, class P>
inline std::vector<bool> classify(Estimator &est, std::vector<double> &params, std::vector<P> &pts) {
  std::vector<bool> in(pts.size(), false);
  if (params.empty()) return in;
  for (size_t i = 0; i < pts.size(); i++) in[i] = est.agree(params, pts[i]);
  return in;
}
#endif
