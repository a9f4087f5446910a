// stands in for open3d_slam/include/open3d_slam/time.hpp:40-49
#pragma once
#include <chrono>
#include <cstdint>
#include <ratio>
namespace o3d_slam {
struct UniversalTimeScaleClock {
  using rep = int64_t;
  using period = std::ratio<1, 10000000>;
  using duration = std::chrono::duration<rep, period>;
  using time_point = std::chrono::time_point<UniversalTimeScaleClock>;
  static constexpr bool is_steady = true;
};
using Time = UniversalTimeScaleClock::time_point;
}  // namespace o3d_slam
