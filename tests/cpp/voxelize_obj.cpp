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
	const uint32_t gridDim = static_cast<uint32_t>(atoi(argv[2]));		// the reference's GRID_SIZE: a constant its Init knows (Content/Voxelizer.cpp:8)
	if (!voxelizer.Init(argv[1], posScale, false, gridDim)) { fprintf(stderr, "Init failed: %s\n", voxelizer.LastError()); return 1; }
	const auto mode = argc > 4 ? Voxelizer::PARITY : Voxelizer::REFERENCE;
	if (!voxelizer.Voxelize(gridDim, mode)) { fprintf(stderr, "Voxelize failed: %s\n", voxelizer.LastError()); return 1; }
	// Init was told the grid: a launch of the reference rule through the lists is the prepared one (queue from Init, one dispatch)
	dxv_stats st;
	if (!voxelizer.GetStats(st)) return 1;
	if (mode == Voxelizer::REFERENCE && st.list_entries && !st.plan_prepared) { fprintf(stderr, "the launch did not use the queue Init prepared\n"); return 1; }
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
