// fx_pcd_selftest — CPU check of the .pcd reader / writer (fx_pcd.hpp), built with the sanitizers by
// tests/test_sanitizers.py: binary and ASCII round trips, a file with extra fields, and malformed files that must be
// refused with an exception (never read out of bounds).   fx_pcd_selftest TMP_DIR
#include <cmath>
#include <cstdio>
#include <fstream>

#include "fx_pcd.hpp"

static int fails = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++fails;                                                       \
    }                                                                \
  } while (0)

template <typename F>
static bool throws(F &&f) {
  try {
    f();
  } catch (const std::exception &) {
    return true;
  }
  return false;
}

int main(int argc, char **argv) {
  if (argc != 2) return 2;
  const std::string dir = argv[1];
  fx::PointCloud cloud;
  for (int i = 0; i < 1000; ++i) cloud.push_back(fx::Point{0.001f * i, -3.5f + 0.25f * i, std::ldexp(1.0f, i % 20 - 10), (float)(i % 7)});
  for (bool binary : {true, false}) {
    const std::string path = dir + (binary ? "/b.pcd" : "/a.pcd");
    fx::write_pcd(path, cloud, binary);
    const fx::PointCloud back = fx::read_pcd(path);
    CHECK(back.size() == cloud.size());
    for (size_t i = 0; i < back.size() && i < cloud.size(); ++i)
      CHECK(back[i].x == cloud[i].x && back[i].y == cloud[i].y && back[i].z == cloud[i].z && back[i].intensity == cloud[i].intensity);
  }
  {  // extra fields before and after x y z (a driver's ring / time), no intensity
    std::ofstream f(dir + "/extra.pcd", std::ios::binary);
    f << "VERSION 0.7\nFIELDS t x y z ring\nSIZE 4 4 4 4 2\nTYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA binary\n";
    for (int i = 0; i < 3; ++i) {
      const float rec[4] = {9.0f, 1.0f + i, 2.0f + i, 3.0f + i};
      const unsigned short ring = (unsigned short)i;
      f.write((const char *)rec, 16);
      f.write((const char *)&ring, 2);
    }
  }
  {
    const fx::PointCloud c = fx::read_pcd(dir + "/extra.pcd");
    CHECK(c.size() == 3 && c[2].x == 3.0f && c[2].y == 4.0f && c[2].z == 5.0f && c[2].intensity == 0.0f);
  }
  auto write = [&](const char *name, const std::string &text) {
    std::ofstream f(dir + "/" + name, std::ios::binary);
    f << text;
  };
  write("trunc_b.pcd", "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 100\nHEIGHT 1\nPOINTS 100\nDATA binary\nabcdefgh");
  write("trunc_a.pcd", "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n1 2 3\n4 5\n");
  write("noxyz.pcd", "FIELDS a b\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2\n");
  write("kind.pcd", "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n");
  write("ragged.pcd", "FIELDS x y z\nSIZE 4 4\nTYPE F\nCOUNT 1 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2 3\n");
  for (const char *bad : {"trunc_b.pcd", "trunc_a.pcd", "noxyz.pcd", "kind.pcd", "missing.pcd"})
    CHECK(throws([&] { fx::read_pcd(dir + "/" + bad); }));
  (void)throws([&] { fx::read_pcd(dir + "/ragged.pcd"); });  // (refused or read: either way without touching memory it does not own)
  std::printf("fx_pcd_selftest: %s\n", fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}
