// NOT Boost: open3d_slam/Parameters.hpp asserts one pointer convertibility with BOOST_CONCEPT_ASSERT; the same check with the standard library
#pragma once
#include <type_traits>
namespace boost {
template <typename From, typename To>
struct Convertible {
  static_assert(std::is_convertible<From, To>::value, "boost::Convertible");
};
template <typename F>
struct concept_arg_;
template <typename M>
struct concept_arg_<void (*)(M)> {
  enum { size = sizeof(M) };
};
}  // namespace boost
#define BOOST_CONCEPT_ASSERT(ModelInParens) static_assert(::boost::concept_arg_<void(*) ModelInParens>::size > 0, "concept")
