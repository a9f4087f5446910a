// NOT boost: the one macro / concept open3d_slam's Parameters.hpp names (see ../Eigen/eigen_shim.hpp for why this exists)
#pragma once
namespace boost {
template <typename A, typename B>
struct Convertible {};
}  // namespace boost
#define BOOST_CONCEPT_ASSERT(x) static_assert(true, "")
