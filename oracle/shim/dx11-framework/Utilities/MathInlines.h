// oracle/shim/dx11-framework/Utilities/MathInlines.h — stand-in for the un-vendored
// stoyannk/dx11-framework header (no pinned version in the reference tree).  Only min/max/clamp
// are used by the reference: TransVoxelImpl.cpp:1519-1521, VoxelGrid.cpp:39,199,378-384,
// 444-450,480-485,538,544,576-581.  Semantics restated here: plain comparisons, two-type
// overloads return the common type.  Test infrastructure only.
#pragma once
#include <type_traits>

namespace StMath
{
template <typename A, typename B>
inline typename std::common_type<A, B>::type min_value(const A& a, const B& b)
{
	typedef typename std::common_type<A, B>::type R;
	return (R(a) < R(b)) ? R(a) : R(b);
}

template <typename A, typename B>
inline typename std::common_type<A, B>::type max_value(const A& a, const B& b)
{
	typedef typename std::common_type<A, B>::type R;
	return (R(a) > R(b)) ? R(a) : R(b);
}

template <typename T>
inline T clamp_value(const T& v, const T& lo, const T& hi)
{
	return (v < lo) ? lo : ((v > hi) ? hi : v);
}
}
