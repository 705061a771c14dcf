// Voxels.h — public C++ API of libVoxels.so, the MI355X-backed drop-in for the stoyannk/voxels library.
//
// The declarations below keep the API *surface* of the reference's public headers (class names, member
// signatures, virtual-function order, struct layouts), so application code written against the reference compiles
// and links unchanged:
//     reference include/Declarations.h, Structs.h, Version.h, Library.h, VoxelSurface.h, MaterialMap.h,
//     Grid.h:29-161, Polygonizer.h:14-239.
// The per-file headers of the same names next to this file simply include it.  The implementation is new:
// the Grid lives on the host (voxels_amd/csrc/vx_grid_host.cpp) and Polygonizer::Execute runs the HIP kernels of
// libvoxels_hip.so through the C ABI in voxels_hip.h (voxels_amd/csrc/vx_api_cpp.cpp).
#pragma once
#ifndef VOXELS_MI355X_API_H
#define VOXELS_MI355X_API_H

#include <stddef.h>

#if defined(_WIN32)
#error "this build of the Voxels API targets Linux + ROCm"
#endif
#ifndef VOXELS_API
#define VOXELS_API __attribute__((visibility("default")))
#endif
#ifndef VOXELS_CDECL
#define VOXELS_CDECL
#endif

#define VOXELS_VERSION 0x00050001 // 0.5.0.1 — same value the reference's headers carry

namespace Voxels
{

// ---------------------------------------------------------------- plain structs
struct VOXELS_API float3
{
	float x, y, z;
	float3() {}
	float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct VOXELS_API float4
{
	float x, y, z, w;
	float4() {}
	float4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};

struct VOXELS_API float3pair
{
	float3 first;
	float3 second;
};

// ---------------------------------------------------------------- library init
enum LogSeverity { LS_Trace = 0, LS_Debug, LS_Info, LS_Warning, LS_Error, LS_CriticalError };

typedef void (*LogMessage)(LogSeverity severity, const char* message);
typedef void* (*VoxelsAllocate_f)(size_t size);
typedef void (*VoxelsDeallocate_f)(void* ptr);
typedef void* (*VoxelsAllocateAligned_f)(size_t size, size_t alignment);
typedef void (*VoxelsDeallocateAligned_f)(void* ptr);

struct VoxelsAllocators
{
	VoxelsAllocate_f VoxelsAllocate;
	VoxelsDeallocate_f VoxelsDeallocate;
	VoxelsAllocateAligned_f VoxelsAllocateAligned;
	VoxelsDeallocateAligned_f VoxelsDeallocateAligned;
};

enum InitError { IE_Ok = 0, IE_VersionMismatch };

// ---------------------------------------------------------------- application callbacks
// Distance (and optional material/blend) provider used while a grid is created or edited. Values are written
// x fastest, then y, then z, for the half-open box [start, end) sampled with the given steps.
class VOXELS_API VoxelSurface
{
public:
	virtual ~VoxelSurface() {}
	virtual void GetSurface(float xStart, float xEnd, float xStep,
	                        float yStart, float yEnd, float yStep,
	                        float zStart, float zEnd, float zStep,
	                        float* output, unsigned char* materialid, unsigned char* blend) = 0;
};

// Material id -> two triplets of texture ids (triplanar: z, xy, -z)
class VOXELS_API MaterialMap
{
public:
	struct Material
	{
		unsigned char DiffuseIds0[3];
		unsigned char DiffuseIds1[3];
	};
	virtual ~MaterialMap() {}
	virtual Material* GetMaterial(unsigned char id) const = 0;
};

// ---------------------------------------------------------------- grid
class VoxelGrid; // host representation (opaque to applications)

enum InjectionType { IT_Add, IT_SubtractAddInner, IT_Subtract };

typedef unsigned char MaterialId;
typedef unsigned char BlendFactor;

// Signed-distance voxel grid, Z up, 16^3 blocks. Owned by the host; the polygonizer mirrors it into HBM.
class VOXELS_API Grid
{
public:
	struct PackedGrid
	{
		virtual void Destroy() = 0;
		virtual unsigned GetSize() const = 0;
		virtual const char* GetData() const = 0;
	};

	static Grid* Create(unsigned w, unsigned d, unsigned h,
	                    float startX, float startY, float startZ, float step, VoxelSurface* surface);
	static Grid* Create(unsigned w, unsigned d, unsigned h);
	static Grid* Create(unsigned w, const char* heightmap);
	static Grid* Load(const char* blob, unsigned size);
	void Destroy();
	PackedGrid* PackForSave() const;

	unsigned GetWidth() const;
	unsigned GetDepth() const;
	unsigned GetHeight() const;

	float3pair InjectSurface(const float3& position, const float3& extents, VoxelSurface* surface, InjectionType type);
	float3pair InjectMaterial(const float3& position, const float3& extents, MaterialId material, bool addSubtractBlend);

	unsigned GetBlockExtent() const;
	bool GetBlockDistanceData(const float3& coords, char* output) const;
	void ModifyBlockDistanceData(const float3& coords, const char* distances);
	bool GetBlockMaterialData(const float3& coords, MaterialId* materials, BlendFactor* blends) const;
	void ModifyBlockMaterialData(const float3& coords, const MaterialId* materials, const BlendFactor* blends);

	unsigned GetGridBlocksMemorySize();
	VoxelGrid* GetInternalRepresentation() const;

private:
	~Grid();
	Grid(VoxelGrid*);
	Grid(const Grid&);
	Grid& operator=(const Grid&);

	VoxelGrid* m_InternalGrid;
};

// ---------------------------------------------------------------- polygonization output
// 48 bytes; SecondaryPosition.w carries the transition-face adjacency mask as raw integer bits.
struct VOXELS_API PolygonVertex
{
	float3 Position;
	float4 SecondaryPosition;
	float3 Normal;
	union {
		struct {
			unsigned char Reserved;
			unsigned char Blend;
			unsigned char Uxz;
			unsigned char Txz;
			unsigned char Uny;
			unsigned char Upy;
			unsigned char Tny;
			unsigned char Tpy;
		} TextureIndices;
		unsigned TI[2];
	} Textures;
};

class BlockPolygons
{
public:
	enum TransitionFaceId { YNeg, ZNeg, XNeg, YPos, ZPos, XPos, Face_Count };

	virtual unsigned GetId() const = 0;
	virtual const PolygonVertex* GetVertices(unsigned* count) const = 0;
	virtual const unsigned* GetIndices(unsigned* count) const = 0;
	virtual const PolygonVertex* GetTransitionVertices(TransitionFaceId face, unsigned* count) const = 0;
	virtual const unsigned* GetTransitionIndices(TransitionFaceId face, unsigned* count) const = 0;
	virtual float3 GetMinimalCorner() const = 0;
	virtual float3 GetMaximalCorner() const = 0;
};

struct VOXELS_API PolygonizationStatistics
{
	unsigned BlocksCalculated;
	unsigned TrivialCells;
	unsigned NonTrivialCells;
	unsigned DegenerateTrianglesRemoved;
	static const unsigned CASES_COUNT = 16;
	unsigned PerCaseCellsCount[CASES_COUNT];
};

class PolygonSurface
{
public:
	virtual float3 GetExtents() const = 0;
	virtual unsigned GetLevelsCount() const = 0;
	virtual unsigned GetBlocksForLevelCount(unsigned level) const = 0;
	virtual const BlockPolygons* GetBlockForLevel(unsigned level, unsigned id) const = 0;
	virtual const PolygonizationStatistics* GetStatistics() const = 0;
	virtual unsigned GetCacheSizeBytes() const = 0;
	virtual unsigned GetPolygonDataSizeBytes() const = 0;
	virtual void Destroy() = 0;

	VOXELS_API static const unsigned INVALID_ID;
};

// Dirty region for incremental re-polygonization; corners as Grid::InjectSurface returns them.
struct VOXELS_API Modification
{
	static Modification* Create();

	PolygonSurface* Map;
	float3 MinCornerModified;
	float3 MaxCornerModified;

	virtual const unsigned* GetModifiedBlocks(unsigned* count) const = 0;
	virtual void Destroy() = 0;
	virtual ~Modification();
};

class VOXELS_API Polygonizer
{
public:
	Polygonizer();
	~Polygonizer();

	PolygonSurface* Execute(const Grid& grid, const MaterialMap* materials, Modification* modification = nullptr);

private:
	Polygonizer(const Polygonizer&);
	Polygonizer& operator=(const Polygonizer&);

	class TransVoxelImpl* m_Impl;
};

} // namespace Voxels

extern "C" VOXELS_API Voxels::InitError VOXELS_CDECL InitializeVoxels(int version, Voxels::LogMessage logger, Voxels::VoxelsAllocators* allocators);
extern "C" VOXELS_API void VOXELS_CDECL DeinitializeVoxels();
extern "C" VOXELS_API unsigned VOXELS_CDECL GetBuildVersion();

#endif
