// vx_grid_host.cpp — see vx_grid_host.h.  Host-only grid maintenance (creation, quantisation, edits, file format);
// none of this is on the polygonization hot path.  Reference rules restated with citations into
// /root/reference/src/VoxelGrid.cpp.
#include "vx_grid_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>

#include "../../include/Voxels.h"

namespace Voxels
{
namespace
{

// :610-672 — run-length form [count, value]*, 255 max per run; falls back to raw bytes when longer than the block.
// `empty` = every run boundary sees the same strict sign as the first voxel (and the coded form is kept).
template <typename T>
bool Encode(const T* data, std::vector<T>& out, bool* empty)
{
	out.clear();
	out.push_back(0);
	size_t ctr = 0;
	unsigned counter = 0;
	bool effective = true;
	if (empty) *empty = true;
	const int first = data[0];
	T last = data[0];
	for (unsigned i = 0; i < VoxelGrid::BLOCK_VOXELS; ++i) {
		const T cur = data[i];
		if (last == cur && counter < 0xFF) { ++counter; continue; }
		out[ctr] = (T)(unsigned char)counter;
		out.push_back(last);
		out.push_back(0);
		ctr = out.size() - 1;
		counter = 1;
		last = cur;
		if (first * (int)last <= 0 && empty) *empty = false;
		if (out.size() > VoxelGrid::BLOCK_VOXELS) { effective = false; break; }
	}
	if (!effective) {
		if (empty) *empty = false;
		out.assign(data, data + VoxelGrid::BLOCK_VOXELS);
		return false;
	}
	out[ctr] = (T)(unsigned char)counter;
	out.push_back(last);
	return true;
}

// :674-694
template <typename T>
void Decode(const T* data, size_t size, bool raw, T* out)
{
	if (raw) { memcpy(out, data, size); return; }
	for (size_t i = 0; i + 1 < size; i += 2) {
		const unsigned len = (unsigned char)data[i];
		for (unsigned k = 0; k < len; ++k) *out++ = data[i + 1];
	}
}

// :37-40 then :42-50
inline int8_t RoundDistance(float v)
{
	float a = std::ceil(std::fabs(v));
	float b = a * (float)(v > 0 ? 1 : -1);
	if (b > 127.f) b = 127.f;
	return (int8_t)(int)b;
}
inline int8_t ClampDistance(int8_t v) { return v > 4 ? 4 : (v < -4 ? -4 : v); }
inline float Clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

} // namespace

namespace { std::atomic<uint64_t> g_NextGridUid(1); }

VoxelGrid::VoxelGrid(uint32_t n) : m_N(n), m_Nb(n / BLOCK), m_Uid(g_NextGridUid.fetch_add(1)), m_Generation(1)
{
	const size_t tot = size_t(n) * n * n;
	m_Dist.assign(tot, 0);
	m_Mat.assign(tot, 0);
	m_Blend.assign(tot, 0);
	m_Meta.assign(size_t(m_Nb) * m_Nb * m_Nb, BlockMeta{ 0, 0, 0, 0 });
	m_BlockGeneration.assign(size_t(m_Nb) * m_Nb * m_Nb, 0);
}

void VoxelGrid::Gather(const uint8_t* src, uint32_t bx, uint32_t by, uint32_t bz, uint8_t* out) const
{
	for (uint32_t z = 0; z < BLOCK; ++z)
	for (uint32_t y = 0; y < BLOCK; ++y)
		memcpy(out + z * 256 + y * 16, src + Index(bx * BLOCK, by * BLOCK + y, bz * BLOCK + z), 16);
}

void VoxelGrid::Scatter(uint8_t* dst, uint32_t bx, uint32_t by, uint32_t bz, const uint8_t* in)
{
	for (uint32_t z = 0; z < BLOCK; ++z)
	for (uint32_t y = 0; y < BLOCK; ++y)
		memcpy(dst + Index(bx * BLOCK, by * BLOCK + y, bz * BLOCK + z), in + z * 256 + y * 16, 16);
}

void VoxelGrid::Touch(uint32_t blockId)
{
	m_BlockGeneration[blockId] = ++m_Generation;
}

void VoxelGrid::DirtySince(uint64_t generation, std::vector<uint32_t>& out) const
{
	out.clear();
	for (size_t id = 0; id < m_BlockGeneration.size(); ++id) if (m_BlockGeneration[id] > generation) out.push_back((uint32_t)id);
}

// what PushBlock / Modify*Data derive from the codec (:52-77, :696-741)
void VoxelGrid::Refresh(uint32_t bx, uint32_t by, uint32_t bz, bool distance, bool material)
{
	BlockMeta& m = m_Meta[BlockId(bx, by, bz)];
	uint8_t tmp[BLOCK_VOXELS];
	if (distance) {
		Gather((const uint8_t*)m_Dist.data(), bx, by, bz, tmp);
		std::vector<char> enc;
		bool empty = false;
		const bool ok = Encode<char>((const char*)tmp, enc, &empty);
		m.flags = ok ? (m.flags & ~BF_DistanceUncompressed) : (m.flags | BF_DistanceUncompressed);
		m.flags = empty ? (m.flags | BF_Empty) : (m.flags & ~BF_Empty);
		m.sizeDist = (uint32_t)enc.size();
	}
	if (material) {
		std::vector<uint8_t> enc;
		Gather(m_Mat.data(), bx, by, bz, tmp);
		bool ok = Encode<uint8_t>(tmp, enc, nullptr);
		m.flags = ok ? (m.flags & ~BF_MaterialUncompressed) : (m.flags | BF_MaterialUncompressed);
		m.sizeMat = (uint32_t)enc.size();
		Gather(m_Blend.data(), bx, by, bz, tmp);
		ok = Encode<uint8_t>(tmp, enc, nullptr);
		m.flags = ok ? (m.flags & ~BF_BlendUncompressed) : (m.flags | BF_BlendUncompressed);
		m.sizeBlend = (uint32_t)enc.size();
	}
}

// :79-132 — one GetSurface call per block, values quantised with round() then clamped to +-4
VoxelGrid* VoxelGrid::FromSurface(uint32_t n, float sx, float sy, float sz, float step, VoxelSurface* surface)
{
	VoxelGrid* g = new VoxelGrid(n);
	std::vector<float> values(BLOCK_VOXELS);
	uint8_t mat[BLOCK_VOXELS], blend[BLOCK_VOXELS], dist[BLOCK_VOXELS];
	for (uint32_t bz = 0; bz < g->m_Nb; ++bz)
	for (uint32_t by = 0; by < g->m_Nb; ++by)
	for (uint32_t bx = 0; bx < g->m_Nb; ++bx) {
		memset(mat, 0, sizeof(mat));
		memset(blend, 0, sizeof(blend));
		const float x0 = sx + bx * BLOCK * step, y0 = sy + by * BLOCK * step, z0 = sz + bz * BLOCK * step;
		surface->GetSurface(x0, x0 + BLOCK * step, step, y0, y0 + BLOCK * step, step, z0, z0 + BLOCK * step, step,
		                    values.data(), mat, blend);
		for (unsigned i = 0; i < BLOCK_VOXELS; ++i) dist[i] = (uint8_t)ClampDistance(RoundDistance(values[i]));
		g->Scatter((uint8_t*)g->m_Dist.data(), bx, by, bz, dist);
		g->Scatter(g->m_Mat.data(), bx, by, bz, mat);
		g->Scatter(g->m_Blend.data(), bx, by, bz, blend);
		g->Refresh(bx, by, bz, true, true);
	}
	return g;
}

// :159-213
VoxelGrid* VoxelGrid::FromHeightmap(uint32_t n, const char* heightmap)
{
	VoxelGrid* g = new VoxelGrid(n);
	for (uint32_t z = 0; z < n; ++z)
	for (uint32_t y = 0; y < n; ++y)
	for (uint32_t x = 0; x < n; ++x) {
		const int slice = (int)z - 127;
		int h = slice - (int)heightmap[y * n + x];
		h = h < -127 ? -127 : (h > 127 ? 127 : h);
		g->m_Dist[g->Index(x, y, z)] = ClampDistance((int8_t)h);
	}
	for (uint32_t bz = 0; bz < g->m_Nb; ++bz) for (uint32_t by = 0; by < g->m_Nb; ++by) for (uint32_t bx = 0; bx < g->m_Nb; ++bx) g->Refresh(bx, by, bz, true, true);
	return g;
}

// :215-267 (file format v1)
VoxelGrid* VoxelGrid::Load(const char* blob, size_t size)
{
	const char* p = blob;
	auto get32 = [&p]() { uint32_t v; memcpy(&v, p, 4); p += 4; return v; };
	if (get32() != 1) return nullptr;
	const uint32_t w = get32();
	get32(); get32();
	VoxelGrid* g = new VoxelGrid(w);
	for (BlockMeta& m : g->m_Meta) { m.sizeDist = get32(); m.sizeMat = get32(); m.sizeBlend = get32(); }
	uint8_t tmp[BLOCK_VOXELS];
	for (uint32_t bz = 0; bz < g->m_Nb; ++bz) for (uint32_t by = 0; by < g->m_Nb; ++by) for (uint32_t bx = 0; bx < g->m_Nb; ++bx) {
		BlockMeta& m = g->m_Meta[g->BlockId(bx, by, bz)];
		m.flags = get32();
		Decode<char>(p, m.sizeDist, (m.flags & BF_DistanceUncompressed) != 0, (char*)tmp); p += m.sizeDist;
		g->Scatter((uint8_t*)g->m_Dist.data(), bx, by, bz, tmp);
		Decode<uint8_t>((const uint8_t*)p, m.sizeMat, (m.flags & BF_MaterialUncompressed) != 0, tmp); p += m.sizeMat;
		g->Scatter(g->m_Mat.data(), bx, by, bz, tmp);
		Decode<uint8_t>((const uint8_t*)p, m.sizeBlend, (m.flags & BF_BlendUncompressed) != 0, tmp); p += m.sizeBlend;
		g->Scatter(g->m_Blend.data(), bx, by, bz, tmp);
	}
	if (size >= (size_t)(p - blob)) { // keep the file: the polygonizer's first upload sends it instead of the dense fields
		g->m_File.assign(blob, p);
		g->m_FileGeneration = g->m_Generation;
	}
	return g;
}

// :269-315
void VoxelGrid::Pack(std::vector<char>& out) const
{
	out.clear();
	auto put32 = [&out](uint32_t v) { const char* p = (const char*)&v; out.insert(out.end(), p, p + 4); };
	put32(1); put32(m_N); put32(m_N); put32(m_N);
	for (const BlockMeta& m : m_Meta) { put32(m.sizeDist); put32(m.sizeMat); put32(m.sizeBlend); }
	uint8_t tmp[BLOCK_VOXELS];
	for (uint32_t bz = 0; bz < m_Nb; ++bz) for (uint32_t by = 0; by < m_Nb; ++by) for (uint32_t bx = 0; bx < m_Nb; ++bx) {
		put32(m_Meta[BlockId(bx, by, bz)].flags);
		std::vector<char> ed;
		Gather((const uint8_t*)m_Dist.data(), bx, by, bz, tmp);
		Encode<char>((const char*)tmp, ed, nullptr);
		out.insert(out.end(), ed.begin(), ed.end());
		std::vector<uint8_t> eu;
		Gather(m_Mat.data(), bx, by, bz, tmp);
		Encode<uint8_t>(tmp, eu, nullptr);
		out.insert(out.end(), (const char*)eu.data(), (const char*)eu.data() + eu.size());
		Gather(m_Blend.data(), bx, by, bz, tmp);
		Encode<uint8_t>(tmp, eu, nullptr);
		out.insert(out.end(), (const char*)eu.data(), (const char*)eu.data() + eu.size());
	}
}

void VoxelGrid::EmptyFlags(std::vector<uint8_t>& out) const
{
	out.resize(m_Meta.size());
	for (size_t i = 0; i < m_Meta.size(); ++i) out[i] = (m_Meta[i].flags & BF_Empty) ? 1 : 0;
}

size_t VoxelGrid::MemoryForBlocks() const
{
	size_t t = 0;
	for (const BlockMeta& m : m_Meta) t += m.sizeDist + m.sizeMat + m.sizeBlend;
	return t;
}

void VoxelGrid::GetBlock(uint32_t bx, uint32_t by, uint32_t bz, int8_t* dist, uint8_t* mat, uint8_t* blend) const
{
	if (dist) Gather((const uint8_t*)m_Dist.data(), bx, by, bz, (uint8_t*)dist);
	if (mat) Gather(m_Mat.data(), bx, by, bz, mat);
	if (blend) Gather(m_Blend.data(), bx, by, bz, blend);
}

void VoxelGrid::SetBlockDistances(uint32_t bx, uint32_t by, uint32_t bz, const int8_t* dist)
{
	Scatter((uint8_t*)m_Dist.data(), bx, by, bz, (const uint8_t*)dist);
	Refresh(bx, by, bz, true, false);
	Touch(BlockId(bx, by, bz));
}

void VoxelGrid::SetBlockMaterials(uint32_t bx, uint32_t by, uint32_t bz, const uint8_t* mat, const uint8_t* blend)
{
	Scatter(m_Mat.data(), bx, by, bz, mat);
	Scatter(m_Blend.data(), bx, by, bz, blend);
	Refresh(bx, by, bz, false, true);
	Touch(BlockId(bx, by, bz));
}

// :331-366 — closed-box test of pos +- ext against every block
void VoxelGrid::TouchedBlocks(const float pos[3], const float ext[3], std::vector<uint32_t>& out) const
{
	for (uint32_t z = 0; z < m_Nb; ++z) for (uint32_t y = 0; y < m_Nb; ++y) for (uint32_t x = 0; x < m_Nb; ++x) {
		const float bmin[3] = { (float)(x * 16), (float)(y * 16), (float)(z * 16) };
		bool hit = true;
		for (int k = 0; k < 3; ++k) {
			const float bmax = (bmin[k] + 8.f) + 8.f;
			if (pos[k] - ext[k] > bmax || bmin[k] > pos[k] + ext[k]) hit = false;
		}
		if (hit) out.push_back(BlockId(x, y, z));
	}
}

// :477-487 — returned in output (Y-up) order
void VoxelGrid::ModifiedBox(const float pos[3], const float ext[3], float outMin[3], float outMax[3]) const
{
	const float p[3] = { pos[0] - ext[0] / 2.0f, pos[1] - ext[1] / 2.0f, pos[2] - ext[2] / 2.0f };
	outMin[0] = std::max(0.f, p[0]); outMin[1] = std::max(0.f, p[2]); outMin[2] = std::max(0.f, p[1]);
	outMax[0] = std::min((float)m_N, outMin[0] + ext[0]);
	outMax[1] = std::min((float)m_N, outMin[1] + ext[2]);
	outMax[2] = std::min((float)m_N, outMin[2] + ext[1]);
}

// :388-488 — the brush is sampled in coordinates relative to `pos`, step 1, over the touched section of each block
void VoxelGrid::InjectSurface(const float pos[3], const float ext[3], VoxelSurface* surface, int type, float outMin[3], float outMax[3])
{
	std::vector<uint32_t> touched;
	TouchedBlocks(pos, ext, touched);
	std::vector<float> vals;
	for (uint32_t id : touched) {
		const uint32_t bx = id % m_Nb, by = (id / m_Nb) % m_Nb, bz = id / (m_Nb * m_Nb);
		const float bmin[3] = { (float)(bx * 16), (float)(by * 16), (float)(bz * 16) };
		float bs[3], be[3];
		for (int k = 0; k < 3; ++k) { // :368-386
			const float p = pos[k] - ext[k] / 2;
			bs[k] = Clampf(p, bmin[k], bmin[k] + 16.f) - bmin[k];
			be[k] = Clampf(p + ext[k], bmin[k], bmin[k] + 16.f) - bmin[k];
		}
		const float s0[3] = { bmin[0] + bs[0] - pos[0], bmin[1] + bs[1] - pos[1], bmin[2] + bs[2] - pos[2] };
		const float s1[3] = { bmin[0] + be[0] - pos[0], bmin[1] + be[1] - pos[1], bmin[2] + be[2] - pos[2] };
		const size_t count = (size_t)std::ceil(s1[0] - s0[0]) * (size_t)std::ceil(s1[1] - s0[1]) * (size_t)std::ceil(s1[2] - s0[2]);
		vals.assign(count + 1, 0.f);
		surface->GetSurface(s0[0], s1[0], 1.f, s0[1], s1[1], 1.f, s0[2], s1[2], 1.f, vals.data(), nullptr, nullptr);
		size_t vi = 0;
		for (float z = bs[2]; z < be[2]; ++z)
		for (float y = bs[1]; y < be[1]; ++y)
		for (float x = bs[0]; x < be[0]; ++x) {
			const size_t i = Index(bx * 16 + (unsigned)x, by * 16 + (unsigned)y, bz * 16 + (unsigned)z);
			const float value = (float)m_Dist[i];
			const float sv = vals[vi++];
			float r;
			if (type == IT_Add) r = std::min(value, sv);
			else if (type == IT_SubtractAddInner) r = std::max(value, sv);
			else r = std::max(-sv, value);
			m_Dist[i] = RoundDistance(r);
		}
		Refresh(bx, by, bz, true, false);
		Touch(id);
	}
	ModifiedBox(pos, ext, outMin, outMax);
}

// :490-584
void VoxelGrid::InjectMaterial(const float pos[3], const float ext[3], uint8_t material, bool add, float outMin[3], float outMax[3])
{
	std::vector<uint32_t> touched;
	TouchedBlocks(pos, ext, touched);
	const float coeff = (ext[0] / 2.0f) * 0.75f;
	for (uint32_t id : touched) {
		const uint32_t bx = id % m_Nb, by = (id / m_Nb) % m_Nb, bz = id / (m_Nb * m_Nb);
		const float bmin[3] = { (float)(bx * 16), (float)(by * 16), (float)(bz * 16) };
		float bs[3], be[3];
		for (int k = 0; k < 3; ++k) {
			const float p = pos[k] - ext[k] / 2;
			bs[k] = Clampf(p, bmin[k], bmin[k] + 16.f) - bmin[k];
			be[k] = Clampf(p + ext[k], bmin[k], bmin[k] + 16.f) - bmin[k];
		}
		for (float z = bs[2]; z < be[2]; ++z)
		for (float y = bs[1]; y < be[1]; ++y)
		for (float x = bs[0]; x < be[0]; ++x) {
			const float cx = x + bmin[0] - pos[0], cy = y + bmin[1] - pos[1], cz = z + bmin[2] - pos[2];
			const float dist = std::sqrt((cx * cx + cy * cy) + cz * cz) / coeff;
			const uint8_t outBlend = (uint8_t)(std::min(1.f, std::max(0.f, (1 - dist))) * 255.f);
			const size_t i = Index(bx * 16 + (unsigned)x, by * 16 + (unsigned)y, bz * 16 + (unsigned)z);
			if (m_Mat[i] == material) {
				m_Blend[i] = (uint8_t)std::max(0, std::min(255, (add ? 1 : -1) * (int)outBlend + (int)m_Blend[i]));
			} else {
				m_Mat[i] = material;
				m_Blend[i] = outBlend;
			}
		}
		Refresh(bx, by, bz, false, true);
		Touch(id);
	}
	ModifiedBox(pos, ext, outMin, outMax);
}

} // namespace Voxels
