// tests/mocks/ros1: see README.md (syntax check of the ROS-1 adapter; declarations only)
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#define ROS_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_FATAL(...) std::fprintf(stderr, __VA_ARGS__)
namespace ros {
struct Time { uint32_t sec = 0, nsec = 0; static Time now(); };
class Publisher {
public:
    template <class M> void publish(const M& msg) const;
};
class Subscriber {};
class NodeHandle {
public:
    NodeHandle(const std::string& ns = std::string());
    template <class T> bool param(const std::string& name, T& val, const T& default_val) const;
    template <class M> Publisher advertise(const std::string& topic, uint32_t queue_size, bool latch = false);
    template <class M> Subscriber subscribe(const std::string& topic, uint32_t queue_size, void (*fp)(const std::shared_ptr<M const>&));
};
void init(int& argc, char** argv, const std::string& name, uint32_t options = 0);
void spin();
}  // namespace ros
