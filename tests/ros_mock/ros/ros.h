// MOCK of ros/ros.h (test infrastructure, see tests/ros_mock/README.md).  NodeHandle::param reads overrides from the
// environment (FX_ROS_PARAM_<name>), subscribe() registers the callback, ros::spin() replays the scenario file
// FX_ROS_MOCK_SCENARIO through the callbacks, and every publish() is written to FX_ROS_MOCK_OUT/<n>_<topic>.{txt,bin}.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../sensor_msgs/Imu.h"
#include "../sensor_msgs/PointCloud2.h"

#define ROS_ERROR(...) (std::fprintf(stderr, "[ros mock ERROR] "), std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
#define ROS_WARN(...) (std::fprintf(stderr, "[ros mock WARN] "), std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
#define ROS_ERROR_THROTTLE(period, ...) ROS_ERROR(__VA_ARGS__)
#define ROS_WARN_THROTTLE(period, ...) ROS_WARN(__VA_ARGS__)

namespace ros {
namespace mock {
struct State {
  std::map<std::string, std::function<void(const sensor_msgs::PointCloud2ConstPtr &)>> cloud_subs;
  std::map<std::string, std::function<void(const sensor_msgs::ImuConstPtr &)>> imu_subs;
  std::vector<std::string> advertised;
  int published = 0;
  std::string node_name;
};
inline State &state() {
  static State s;
  return s;
}
inline const char *env(const char *name, const char *dflt) {
  const char *v = std::getenv(name);
  return v ? v : dflt;
}
inline void dump(const std::string &topic, const sensor_msgs::PointCloud2 &m) {
  State &s = state();
  const std::string base = std::string(env("FX_ROS_MOCK_OUT", ".")) + "/" + std::to_string(s.published++) + "_" + topic;
  std::ofstream t(base + ".txt");
  t << "topic " << topic << "\nframe_id " << m.header.frame_id << "\nstamp " << m.header.stamp.sec << " " << m.header.stamp.nsec
    << "\nheight " << m.height << "\nwidth " << m.width << "\npoint_step " << m.point_step << "\nrow_step " << m.row_step
    << "\nis_bigendian " << (m.is_bigendian ? 1 : 0) << "\nis_dense " << (m.is_dense ? 1 : 0) << "\n";
  for (const auto &f : m.fields) t << "field " << f.name << " " << f.offset << " " << (int)f.datatype << " " << f.count << "\n";
  std::ofstream b(base + ".bin", std::ios::binary);
  b.write(reinterpret_cast<const char *>(m.data.data()), (std::streamsize)m.data.size());
}
}  // namespace mock

inline void init(int &, char **, const std::string &name) { mock::state().node_name = name; }

class Publisher {
 public:
  Publisher() = default;
  explicit Publisher(std::string topic) : topic_(std::move(topic)) {}
  void publish(const sensor_msgs::PointCloud2 &m) const { mock::dump(topic_, m); }

 private:
  std::string topic_;
};
class Subscriber {};

class NodeHandle {
 public:
  explicit NodeHandle(const std::string &ns = "") : ns_(ns) {}
  // private parameter `name`: the environment variable FX_ROS_PARAM_<name> overrides the default
  template <typename T>
  bool param(const std::string &name, T &value, const T &dflt) const {
    const char *v = std::getenv(("FX_ROS_PARAM_" + name).c_str());
    if (!v) {
      value = dflt;
      return false;
    }
    std::istringstream is(v);
    is >> value;
    return true;
  }
  template <typename M>
  Publisher advertise(const std::string &topic, uint32_t /*queue*/) {
    mock::state().advertised.push_back(topic);
    return Publisher(topic);
  }
  template <typename T>
  Subscriber subscribe(const std::string &topic, uint32_t /*queue*/, void (T::*fp)(const sensor_msgs::PointCloud2ConstPtr &), T *obj) {
    mock::state().cloud_subs[topic] = [obj, fp](const sensor_msgs::PointCloud2ConstPtr &m) { (obj->*fp)(m); };
    return Subscriber();
  }
  template <typename T>
  Subscriber subscribe(const std::string &topic, uint32_t /*queue*/, void (T::*fp)(const sensor_msgs::ImuConstPtr &), T *obj) {
    mock::state().imu_subs[topic] = [obj, fp](const sensor_msgs::ImuConstPtr &m) { (obj->*fp)(m); };
    return Subscriber();
  }

 private:
  std::string ns_;
};

// Replays FX_ROS_MOCK_SCENARIO, one message per line, through the subscribed callbacks:
//   imu <topic> qx qy qz qw
//   cloud <topic> <data file> frame_id sec nsec height width point_step row_step is_bigendian n_fields {name offset datatype count}...
inline void spin() {
  mock::State &s = mock::state();
  {
    std::ofstream t(std::string(mock::env("FX_ROS_MOCK_OUT", ".")) + "/node.txt");
    t << "node " << s.node_name << "\n";
    for (const auto &a : s.advertised) t << "advertise " << a << "\n";
    for (const auto &c : s.cloud_subs) t << "subscribe " << c.first << "\n";
    for (const auto &c : s.imu_subs) t << "subscribe " << c.first << "\n";
  }
  std::ifstream in(mock::env("FX_ROS_MOCK_SCENARIO", "scenario.txt"));
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream is(line);
    std::string kind, topic;
    is >> kind >> topic;
    if (kind == "imu") {
      auto m = std::make_shared<sensor_msgs::Imu>();
      is >> m->orientation.x >> m->orientation.y >> m->orientation.z >> m->orientation.w;
      auto it = s.imu_subs.find(topic);
      if (it != s.imu_subs.end()) it->second(m);
    } else if (kind == "cloud") {
      auto m = std::make_shared<sensor_msgs::PointCloud2>();
      std::string file;
      int big = 0, nf = 0;
      is >> file >> m->header.frame_id >> m->header.stamp.sec >> m->header.stamp.nsec >> m->height >> m->width >> m->point_step >>
          m->row_step >> big >> nf;
      m->is_bigendian = big != 0;
      for (int i = 0; i < nf; ++i) {
        sensor_msgs::PointField f;
        int dt = 0;
        is >> f.name >> f.offset >> dt >> f.count;
        f.datatype = (uint8_t)dt;
        m->fields.push_back(f);
      }
      std::ifstream b(file, std::ios::binary);
      m->data.assign(std::istreambuf_iterator<char>(b), std::istreambuf_iterator<char>());
      m->is_dense = true;
      auto it = s.cloud_subs.find(topic);
      if (it != s.cloud_subs.end()) it->second(m);
    }
  }
}
}  // namespace ros
