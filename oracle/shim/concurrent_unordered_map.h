// oracle/shim/concurrent_unordered_map.h — stand-in for MSVC PPL's
// concurrency::concurrent_unordered_map (TransVoxelImpl.cpp:30, used at :369-375, :509-512,
// :2145).  A mutex around std::unordered_map exposing only what the reference touches.
// Iterators stay valid across inserts of other keys (node-based container), which is all the
// reference relies on.  Test infrastructure only.
#pragma once
#include <unordered_map>
#include <mutex>
#include <utility>

namespace concurrency
{
template <typename K, typename V>
class concurrent_unordered_map
{
public:
	typedef std::unordered_map<K, V> map_type;
	typedef typename map_type::value_type value_type;
	typedef typename map_type::iterator iterator;
	typedef typename map_type::const_iterator const_iterator;

	iterator find(const K& k)
	{
		std::lock_guard<std::mutex> g(m_Lock);
		return m_Map.find(k);
	}
	iterator end() { return m_Map.end(); }
	const_iterator cbegin() const { return m_Map.cbegin(); }
	const_iterator cend() const { return m_Map.cend(); }
	template <typename P>
	std::pair<iterator, bool> insert(P&& v)
	{
		std::lock_guard<std::mutex> g(m_Lock);
		return m_Map.insert(std::forward<P>(v));
	}
	void clear()
	{
		std::lock_guard<std::mutex> g(m_Lock);
		m_Map.clear();
	}

private:
	map_type m_Map;
	std::mutex m_Lock;
};
}
