// The multi-GPU host in C++ (include/dxv_multi.hpp): one process, one context per device, rank-0 build, the scene blob
// broadcast through the RCCL C API, block-cyclic and slab partitions, no Python.  Same code path for 1 and for N devices.
//   multi_gpu mesh.bin gridDim out.bin [numDevices]
// mesh.bin: uint32 V, uint32 T, V x 6 floats (pos.xyz, nrm.xyz), 3T uint32 indices -- the layout ObjLoader produces
// (XUSG/Optional/XUSGObjLoader.cpp:25-26), written by the test from the committed loader fixture.
#include "../../include/dxv_multi.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s mesh.bin gridDim out.bin [numDevices]\n", argv[0]); return 2; }
	FILE* f = fopen(argv[1], "rb");
	if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
	uint32_t hdr[2];
	if (fread(hdr, 4, 2, f) != 2) return 2;
	std::vector<float> vb(static_cast<size_t>(hdr[0]) * 6);
	std::vector<uint32_t> ib(static_cast<size_t>(hdr[1]) * 3);
	if (fread(vb.data(), 4, vb.size(), f) != vb.size() || fread(ib.data(), 4, ib.size(), f) != ib.size()) return 2;
	fclose(f);
	const uint32_t N = static_cast<uint32_t>(atoi(argv[2]));

	int visible = 0;
	if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { fprintf(stderr, "no HIP device\n"); return 1; }
	int want = argc > 4 ? atoi(argv[4]) : visible;
	if (want > visible) want = visible;
	std::vector<int> devices;
	for (int d = 0; d < want; ++d) devices.push_back(d);

	MultiVoxelizer vox(devices);
	vox.SetGridHint(N, MultiVoxelizer::BLOCK_CYCLIC, 8);			// Init knows the grid (the reference's GRID_SIZE): every device prepares its share's work queue after the import
	if (!vox.InitFromArrays(vb.data(), hdr[0], ib.data(), hdr[1])) { fprintf(stderr, "Init failed: %s\n", vox.LastError()); return 1; }
	std::vector<uint8_t> cyclic, slabs;
	uint64_t solidCyclic = 0, solidSlabs = 0;
	if (!vox.Voxelize(N, MultiVoxelizer::REFERENCE, MultiVoxelizer::BLOCK_CYCLIC, 8) || !vox.Download(cyclic) || !vox.CountSolid(solidCyclic)) {
		fprintf(stderr, "block-cyclic: %s\n", vox.LastError()); return 1;
	}
	for (size_t d = 0; d < vox.DeviceCount(); ++d) {				// the prepared partition's launch is the one dispatch; the slabs below were not prepared
		dxv_stats st;
		if (!vox.GetStats(d, st)) return 1;
		if (st.list_entries && N % (8u * static_cast<uint32_t>(vox.DeviceCount())) == 0 && !st.plan_prepared) { fprintf(stderr, "device %zu: the block-cyclic launch did not use the prepared queue\n", d); return 1; }
	}
	if (!vox.Voxelize(N, MultiVoxelizer::REFERENCE, MultiVoxelizer::SLABS) || !vox.Download(slabs) || !vox.CountSolid(solidSlabs)) {
		fprintf(stderr, "slabs: %s\n", vox.LastError()); return 1;
	}
	if (cyclic != slabs || solidCyclic != solidSlabs) { fprintf(stderr, "the two partitions give different grids\n"); return 1; }
	if (vox.Voxelize(63)) { fprintf(stderr, "odd grid accepted\n"); return 1; }				// bool returns, never exceptions
	FILE* o = fopen(argv[3], "wb");
	if (!o) return 1;
	fwrite(cyclic.data(), 1, cyclic.size(), o);
	fclose(o);
	// the partition arithmetic of EIGHT shares, played by the first context alone (a box with one GPU cannot hold eight contexts'
	// worth of devices, but the index arithmetic is the shares'): blocks of 4 slices dealt round-robin, what bench.py --gpus 8 runs
	if (argc > 5) {
		struct Probe : MultiVoxelizer { using MultiVoxelizer::MultiVoxelizer; dxv_ctx* first() { return m_ctx[0]; } } ;
		const uint32_t G = 8, zb = 4;
		if (N % (G * zb) == 0) {
			dxv_ctx* ctx = static_cast<Probe&>(vox).first();
			std::vector<uint8_t> whole(static_cast<size_t>(N) * N * N), part(static_cast<size_t>(N) * N * (N / G));
			const size_t plane = static_cast<size_t>(N) * N;
			for (uint32_t g = 0; g < G; ++g) {
				if (dxv_voxelize_interleaved(ctx, N, DXV_MODE_REFERENCE, g, G, zb) || dxv_grid_download(ctx, part.data(), part.size())) { fprintf(stderr, "share %u: %s\n", g, dxv_last_error(ctx)); return 1; }
				for (uint32_t lz = 0; lz < N / G; ++lz) memcpy(whole.data() + plane * MultiVoxelizer::GlobalSlice(lz, g, G, zb), part.data() + plane * lz, plane);
			}
			if (whole != cyclic) { fprintf(stderr, "eight shares of 4-slice blocks do not reassemble to the grid\n"); return 1; }
		}
	}
	printf("%llu %zu %zu %.3f %016llx\n", static_cast<unsigned long long>(solidCyclic), vox.DeviceCount(), vox.SceneBytes(), vox.BroadcastMs(),
		static_cast<unsigned long long>(vox.SceneChecksum()));
	return 0;
}
