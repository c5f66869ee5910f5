// Driver for the reference ObjLoader built from /root/reference (see oracle/Makefile).
// TEST INFRASTRUCTURE: dumps what XUSG::ObjLoader::Import(file, true, true) produces, i.e. what
// Content/Voxelizer.cpp:46-57 consumes, as one binary blob:
//   u32 V, u32 nIdx, u32 stride, f32 aabb[6], V*stride bytes VB, nIdx*4 bytes IB.
#include "XUSGObjLoader.h"

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s in.obj out.bin\n", argv[0]); return 2; }
	XUSG::ObjLoader loader;
	if (!loader.Import(argv[1], true, true)) { fprintf(stderr, "Import failed\n"); return 1; }
	const uint32_t hdr[3] = { loader.GetNumVertices(), loader.GetNumIndices(), loader.GetVertexStride() };
	const auto& aabb = loader.GetAABB();
	const float box[6] = { aabb.Min.x, aabb.Min.y, aabb.Min.z, aabb.Max.x, aabb.Max.y, aabb.Max.z };
	FILE* f = fopen(argv[2], "wb");
	if (!f) return 3;
	fwrite(hdr, sizeof(hdr), 1, f);
	fwrite(box, sizeof(box), 1, f);
	fwrite(loader.GetVertices(), 1, size_t(hdr[0]) * hdr[2], f);
	fwrite(loader.GetIndices(), sizeof(uint32_t), hdr[1], f);
	fclose(f);
	return 0;
}
