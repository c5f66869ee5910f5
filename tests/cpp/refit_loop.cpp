// A mesh animated on fixed topology, voxelized every frame from C++ (the reference API's PERFORM_UPDATE use,
// XUSG/RayTracing/XUSGRayTracing.h:13-22, through include/dxv_voxelizer.hpp): the loop that hides the vertex upload --
//   VoxelizeAsync(frame i);  UploadVertices(frame i + 1)  (does not wait: launches read the scene, not the vertex buffer);
//   Refit()  (waits for the launch, then refits);  ...
// usage: refit_loop verts_a.bin verts_b.bin indices.bin gridDim frames out.bin
//   verts_*.bin: numVerts x 6 floats (two poses of the same mesh), indices.bin: 3 x numTris uint32.
// Alternates the two poses; writes the grid of the LAST frame and prints its solid count and pose (a / b).
#include "../../include/dxv_voxelizer.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

template <class T>
static bool readAll(const char* path, std::vector<T>& out)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	const long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	out.resize(static_cast<size_t>(bytes) / sizeof(T));
	const bool ok = fread(out.data(), sizeof(T), out.size(), f) == out.size();
	fclose(f);
	return ok;
}

int main(int argc, char** argv)
{
	if (argc < 7) { fprintf(stderr, "usage: %s verts_a.bin verts_b.bin indices.bin gridDim frames out.bin\n", argv[0]); return 2; }
	std::vector<float> pose[2];
	std::vector<uint32_t> ib;
	if (!readAll(argv[1], pose[0]) || !readAll(argv[2], pose[1]) || !readAll(argv[3], ib) || pose[0].size() != pose[1].size()) {
		fprintf(stderr, "cannot read the mesh files\n");
		return 2;
	}
	const uint32_t numVerts = static_cast<uint32_t>(pose[0].size() / 6), numTris = static_cast<uint32_t>(ib.size() / 3);
	const uint32_t gridDim = static_cast<uint32_t>(atoi(argv[4]));
	const int frames = atoi(argv[5]);
	const float posScale[4] = { 0.0f, 0.0f, 0.0f, 1.0f };
	Voxelizer voxelizer;
	if (!voxelizer.InitFromArrays(pose[0].data(), numVerts, ib.data(), numTris, posScale)) { fprintf(stderr, "Init failed: %s\n", voxelizer.LastError()); return 1; }
	int cur = 0;
	for (int i = 0; i + 1 < frames; ++i) {
		if (!voxelizer.VoxelizeAsync(0, gridDim)) { fprintf(stderr, "VoxelizeAsync failed: %s\n", voxelizer.LastError()); return 1; }
		cur ^= 1;
		if (!voxelizer.UploadVertices(pose[cur].data(), numVerts)) { fprintf(stderr, "UploadVertices failed: %s\n", voxelizer.LastError()); return 1; }
		if (!voxelizer.Refit()) { fprintf(stderr, "Refit failed: %s\n", voxelizer.LastError()); return 1; }
	}
	if (!voxelizer.Voxelize(gridDim)) { fprintf(stderr, "Voxelize failed: %s\n", voxelizer.LastError()); return 1; }
	std::vector<uint8_t> grid;
	uint64_t solid = 0;
	if (!voxelizer.Download(grid) || !voxelizer.CountSolid(solid)) { fprintf(stderr, "%s\n", voxelizer.LastError()); return 1; }
	FILE* f = fopen(argv[6], "wb");
	if (!f) return 1;
	fwrite(grid.data(), 1, grid.size(), f);
	fclose(f);
	printf("%llu %c\n", static_cast<unsigned long long>(solid), cur ? 'b' : 'a');
	return 0;
}
