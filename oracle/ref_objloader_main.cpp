// Driver for the reference ObjLoader built from /root/reference (see oracle/Makefile).
// TEST INFRASTRUCTURE: dumps what XUSG::ObjLoader::Import(file, true, true) produces, i.e. what
// Content/Voxelizer.cpp:46-57 consumes, as one binary blob:
//   u32 V, u32 nIdx, u32 stride, f32 aabb[6], V*stride bytes VB, nIdx*4 bytes IB.
// With a third argument N: Import runs N times on fresh loaders and the call's wall time (steady clock around Import alone: no
// process start, no dump) is printed as one JSON line per run -- the one same-box timing of reference code this project can have
// (tools/obj_ingest_vs_reference.py).
#include "XUSGObjLoader.h"
#include <chrono>

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s in.obj out.bin [timed runs]\n", argv[0]); return 2; }
	for (int run = 0; argc > 3 && run < atoi(argv[3]); ++run) {
		XUSG::ObjLoader timed;
		const auto t0 = std::chrono::steady_clock::now();
		if (!timed.Import(argv[1], true, true)) { fprintf(stderr, "Import failed\n"); return 1; }
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		printf("{\"what\": \"XUSG::ObjLoader::Import\", \"run\": %d, \"ms\": %.3f, \"verts\": %u, \"indices\": %u}\n", run, ms, timed.GetNumVertices(), timed.GetNumIndices());
	}
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
