// fx_pcd.hpp — minimal .pcd reader/writer (ASCII and binary, float32 fields) for the CLI:
// BASELINE config 1 feeds one VLP-16 sweep from a .pcd file.
#ifndef FX_PCD_HPP_
#define FX_PCD_HPP_
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "fx_node.hpp"

namespace fx {

inline PointCloud read_pcd(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t points = 0;
  std::string data_kind, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "FIELDS") {
      for (std::string s; ss >> s;) fields.push_back(s);
    } else if (key == "SIZE") {
      for (int v; ss >> v;) sizes.push_back(v);
    } else if (key == "TYPE") {
      for (char c; ss >> c;) types.push_back(c);
    } else if (key == "COUNT") {
      for (int v; ss >> v;) counts.push_back(v);
    } else if (key == "POINTS") {
      ss >> points;
    } else if (key == "DATA") {
      ss >> data_kind;
      break;
    }
  }
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (fields.size() != sizes.size() || fields.size() != types.size() || fields.size() != counts.size())
    throw std::runtime_error("malformed .pcd header in " + path);
  int off[4] = {-1, -1, -1, -1}, col[4] = {-1, -1, -1, -1};
  int stride = 0, ncol = 0;
  const char *want[4] = {"x", "y", "z", "intensity"};
  for (size_t i = 0; i < fields.size(); ++i) {
    for (int w = 0; w < 4; ++w)
      if (fields[i] == want[w]) {
        if (sizes[i] != 4 || types[i] != 'F') throw std::runtime_error("field " + fields[i] + " must be float32");
        off[w] = stride;
        col[w] = ncol;
      }
    stride += sizes[i] * counts[i];
    ncol += counts[i];
  }
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) throw std::runtime_error(".pcd needs x y z fields");
  PointCloud cloud(points);
  if (data_kind == "ascii") {
    std::vector<double> row(ncol);
    for (size_t p = 0; p < points; ++p) {
      for (int c = 0; c < ncol; ++c)
        if (!(f >> row[c])) throw std::runtime_error("truncated ascii .pcd");
      cloud[p] = Point{(float)row[col[0]], (float)row[col[1]], (float)row[col[2]], col[3] >= 0 ? (float)row[col[3]] : 0.f};
    }
  } else if (data_kind == "binary") {
    std::vector<char> rec(stride);
    for (size_t p = 0; p < points; ++p) {
      if (!f.read(rec.data(), stride)) throw std::runtime_error("truncated binary .pcd");
      Point q{0, 0, 0, 0};
      std::memcpy(&q.x, rec.data() + off[0], 4);
      std::memcpy(&q.y, rec.data() + off[1], 4);
      std::memcpy(&q.z, rec.data() + off[2], 4);
      if (off[3] >= 0) std::memcpy(&q.intensity, rec.data() + off[3], 4);
      cloud[p] = q;
    }
  } else {
    throw std::runtime_error("unsupported .pcd DATA kind: " + data_kind);
  }
  return cloud;
}

inline void write_pcd(const std::string &path, const PointCloud &cloud, bool binary = true) {
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot write " + path);
  f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
    << "COUNT 1 1 1 1\nWIDTH " << cloud.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << cloud.size()
    << "\nDATA " << (binary ? "binary" : "ascii") << "\n";
  if (binary) {
    f.write(reinterpret_cast<const char *>(cloud.data()), (std::streamsize)(cloud.size() * sizeof(Point)));
  } else {
    char buf[128];
    for (const Point &p : cloud) {
      std::snprintf(buf, sizeof(buf), "%.9g %.9g %.9g %.9g\n", p.x, p.y, p.z, p.intensity);
      f << buf;
    }
  }
}

}  // namespace fx
#endif
