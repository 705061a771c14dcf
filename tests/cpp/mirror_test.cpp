// tests/cpp/mirror_test.cpp — the device mirror of a Grid behind Polygonizer::Execute (libVoxels.so):
//   * a Polygonizer reused on a NEW grid that happens to live at the address of a destroyed one must upload it (the
//     reference's Polygonizer keeps no grid state, so this usage is legitimate);
//   * two Polygonizers mirroring ONE grid must both see an edit (change tracking is per consumer).
// Prints OK / FAIL; run by tests/test_dropin_cpp.py on a GPU box.
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <Voxels.h>
using namespace Voxels;
struct Ball : VoxelSurface { float r, c; void GetSurface(float xs, float xe, float xst, float ys, float ye, float yst, float zs, float ze, float zst, float* o, unsigned char* m, unsigned char* b) override {
  size_t k = 0; for (float z = zs; z < ze; z += zst) for (float y = ys; y < ye; y += yst) for (float x = xs; x < xe; x += xst) { o[k] = sqrtf((x-c)*(x-c)+(y-c)*(y-c)+(z-c)*(z-c)) - r; if (m) m[k] = 0; if (b) b[k] = 0; ++k; } } };
static void Quiet(LogSeverity, const char*) {}
static unsigned long long Count(PolygonSurface* s) { unsigned long long v = 0; for (unsigned l = 0; l < s->GetLevelsCount(); ++l) for (unsigned b = 0; b < s->GetBlocksForLevelCount(l); ++b) { unsigned c = 0; s->GetBlockForLevel(l, b)->GetVertices(&c); v += c; } return v; }
int main() {
  InitializeVoxels(VOXELS_VERSION, &Quiet, nullptr);
  Polygonizer p, q;
  Ball a; a.r = 20; a.c = 32; Ball bb; bb.r = 10; bb.c = 32;
  Grid* A = Grid::Create(64, 64, 64, 0, 0, 0, 1, &a);
  PolygonSurface* sA = p.Execute(*A, nullptr); const unsigned long long vA = Count(sA); sA->Destroy();
  void* addrA = (void*)A->GetInternalRepresentation(); A->Destroy();
  Grid* B = Grid::Create(64, 64, 64, 0, 0, 0, 1, &bb);
  PolygonSurface* sB = p.Execute(*B, nullptr); const unsigned long long vB = Count(sB); sB->Destroy();
  PolygonSurface* sBq = q.Execute(*B, nullptr); const unsigned long long vBq = Count(sBq); sBq->Destroy();
  printf("same address: %d  vertsA %llu vertsB %llu (fresh polygonizer: %llu)\n", addrA == (void*)B->GetInternalRepresentation(), vA, vB, vBq);
  // two mirrors of one grid: an edit must reach both
  float3 pos(32, 32, 32), ext(30, 30, 30); Ball carve; carve.r = 14; carve.c = 0;
  auto box = B->InjectSurface(pos, ext, &carve, IT_Add); (void)box;
  PolygonSurface* s1 = p.Execute(*B, nullptr); PolygonSurface* s2 = q.Execute(*B, nullptr);
  printf("after edit: %llu vs %llu\n", Count(s1), Count(s2));
  const bool ok = vB == vBq && vA != vB && Count(s1) == Count(s2) && Count(s1) != vB;
  s1->Destroy(); s2->Destroy(); B->Destroy();
  printf("%s\n", ok ? "OK" : "FAIL"); return ok ? 0 : 1;
}
