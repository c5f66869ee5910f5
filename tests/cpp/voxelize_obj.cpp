// Reads like the reference's call site (DXRVoxelizer.cpp:186-193 -> Voxelizer::Init, :466 ->
// Voxelizer::Render -> voxelize): Init(fileName, posScale) then Voxelize(gridDim), through the
// host-side C++ mirror include/dxv_voxelizer.hpp.  Writes the grid to a file for the test.
#include "../../include/dxv_voxelizer.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s mesh.obj gridDim out.bin [parity]\n", argv[0]); return 2; }
	const float posScale[4] = { 0.0f, 0.0f, 0.0f, 1.0f };	// DXRVoxelizer.cpp:37 default
	Voxelizer voxelizer;
	if (!voxelizer.Init(argv[1], posScale)) { fprintf(stderr, "Init failed: %s\n", voxelizer.LastError()); return 1; }
	const auto mode = argc > 4 ? Voxelizer::PARITY : Voxelizer::REFERENCE;
	if (!voxelizer.Voxelize(static_cast<uint32_t>(atoi(argv[2])), mode)) { fprintf(stderr, "Voxelize failed: %s\n", voxelizer.LastError()); return 1; }
	std::vector<uint8_t> grid;
	uint64_t solid = 0;
	if (!voxelizer.Download(grid) || !voxelizer.CountSolid(solid)) { fprintf(stderr, "%s\n", voxelizer.LastError()); return 1; }
	// error convention: bool returns, never exceptions (XUSG/Core/XUSG.h:12-15)
	if (voxelizer.Voxelize(63)) { fprintf(stderr, "odd grid accepted\n"); return 1; }
	FILE* f = fopen(argv[3], "wb");
	if (!f) return 1;
	fwrite(grid.data(), 1, grid.size(), f);
	fclose(f);
	printf("%llu\n", static_cast<unsigned long long>(solid));
	return 0;
}
