// vx_grid_host.h — host-side voxel grid behind Voxels::Grid (include/Voxels.h).
//
// Replaces the reference's VoxelGrid (src/VoxelGrid.{h,cpp}) on the host.  The polygonizer only needs dense bytes in
// HBM, so the primary store is dense (x fastest, Z up); what the reference derives from its per-block run-length
// codec is kept as per-block metadata, computed by the same rules: BF_Empty / *Uncompressed flags and encoded sizes
// (VoxelGrid.cpp:52-77, :610-672) for IsBlockEmpty, GetGridBlocksMemorySize and the v1 file format (:215-315).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace Voxels
{
class VoxelSurface;

class VoxelGrid
{
public:
	enum { BLOCK = 16, BLOCK_VOXELS = 4096 };
	enum { BF_Empty = 1, BF_DistanceUncompressed = 2, BF_MaterialUncompressed = 4, BF_BlendUncompressed = 8 };

	struct BlockMeta { uint32_t flags, sizeDist, sizeMat, sizeBlend; };

	explicit VoxelGrid(uint32_t n);

	static VoxelGrid* FromSurface(uint32_t n, float sx, float sy, float sz, float step, VoxelSurface* surface);
	static VoxelGrid* FromHeightmap(uint32_t n, const char* heightmap);
	static VoxelGrid* Load(const char* blob, size_t size = 0);
	void Pack(std::vector<char>& out) const;

	uint32_t Size() const { return m_N; }
	uint32_t BlocksPerAxis() const { return m_Nb; }
	size_t Index(uint32_t x, uint32_t y, uint32_t z) const { return (size_t(z) * m_N + y) * m_N + x; }
	uint32_t BlockId(uint32_t bx, uint32_t by, uint32_t bz) const { return bx + by * m_Nb + bz * m_Nb * m_Nb; }

	const int8_t* Distances() const { return m_Dist.data(); }
	const uint8_t* Materials() const { return m_Mat.data(); }
	const uint8_t* Blends() const { return m_Blend.data(); }
	void EmptyFlags(std::vector<uint8_t>& out) const;
	size_t MemoryForBlocks() const;

	void GetBlock(uint32_t bx, uint32_t by, uint32_t bz, int8_t* dist, uint8_t* mat, uint8_t* blend) const;
	void SetBlockDistances(uint32_t bx, uint32_t by, uint32_t bz, const int8_t* dist);
	void SetBlockMaterials(uint32_t bx, uint32_t by, uint32_t bz, const uint8_t* mat, const uint8_t* blend);

	void InjectSurface(const float pos[3], const float ext[3], VoxelSurface* surface, int type, float outMin[3], float outMax[3]);
	void InjectMaterial(const float pos[3], const float ext[3], uint8_t material, bool add, float outMin[3], float outMax[3]);

	// Change tracking for device mirrors.  Every grid has a process-unique id (a new grid may reuse the address of a
	// destroyed one) and a generation that grows with every edited block; every block remembers the generation of its
	// last edit, so any number of mirrors can ask "what changed since the generation I hold" without sharing state.
	uint64_t Uid() const { return m_Uid; }
	uint64_t Generation() const { return m_Generation; }
	void DirtySince(uint64_t generation, std::vector<uint32_t>& out) const;
	// the file a grid was loaded from, kept until the first upload so that the device can expand it itself
	// (valid only while nothing was edited since the load)
	const std::vector<char>* PristineFile() const { return (m_FileGeneration == m_Generation && !m_File.empty()) ? &m_File : nullptr; }
	void DropFile() { std::vector<char>().swap(m_File); }

private:
	void Gather(const uint8_t* src, uint32_t bx, uint32_t by, uint32_t bz, uint8_t* out) const;
	void Scatter(uint8_t* dst, uint32_t bx, uint32_t by, uint32_t bz, const uint8_t* in);
	void Refresh(uint32_t bx, uint32_t by, uint32_t bz, bool distance, bool material);
	void Touch(uint32_t blockId);
	void TouchedBlocks(const float pos[3], const float ext[3], std::vector<uint32_t>& out) const;
	void ModifiedBox(const float pos[3], const float ext[3], float outMin[3], float outMax[3]) const;

	uint32_t m_N, m_Nb;
	std::vector<int8_t> m_Dist;
	std::vector<uint8_t> m_Mat, m_Blend;
	std::vector<BlockMeta> m_Meta;
	uint64_t m_Uid;
	uint64_t m_Generation;
	std::vector<uint64_t> m_BlockGeneration; // per block: generation of its last edit (0 = as created)
	std::vector<char> m_File;
	uint64_t m_FileGeneration = 0;
};

} // namespace Voxels
