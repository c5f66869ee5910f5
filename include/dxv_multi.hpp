// dxv_multi.hpp -- the multi-GPU host of the voxelizer in C++, no Python: one context per device of ONE process, the scene
// built once and broadcast over RCCL (xGMI), the grid's Z axis partitioned across the devices, no per-frame collective.
// Header only; compile with the HIP and RCCL headers (-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include) and link with libdxv.so,
// libamdhip64 and librccl.
//
// What it stands in for: the reference has one Voxelizer per application and one GPU (Content/Voxelizer.h:10-24,
// DXRVoxelizer.cpp:186-193); north_star spreads that component over the 8 GPUs of a node -- "the voxel grid is
// slab-partitioned along Z across the 8 GPUs ... with the BVH broadcast once via RCCL over xGMI and no per-frame
// collectives".  The shape to match is the reference's: ONE build in Init (Content/Voxelizer.cpp:73, :264-326), then
// independent per-voxel work per frame (:351-369, DXRVoxelizer.hlsl:58-85).  Same surface as include/dxv_voxelizer.hpp:
// Init / InitFromArrays / Voxelize, bool returns, never throws (XUSG/Core/XUSG.h:12-15).
//
// One host thread drives all devices: every dxv_* call selects its context's device itself, launches are asynchronous, so
// the N launches of a Voxelize overlap; Voxelize returns when all of them have finished.  (The Python host in
// dxrvoxelizer_amd/slabs.py + bench.py does the same with one PROCESS per GPU and torch.distributed; both paths move the
// same blob through the same dxv_scene_export / dxv_scene_import.)
#pragma once
#include "dxv.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

class MultiVoxelizer
{
public:
	enum Mode : int { REFERENCE = DXV_MODE_REFERENCE, PARITY = DXV_MODE_PARITY };
	// SLABS: device g of G owns the contiguous slices [g N / G, (g + 1) N / G) (north_star's partition; the GPUs that own
	// empty space idle).  BLOCK_CYCLIC: blocks of `zblock` slices dealt round-robin (SURVEY 8(e)'s fallback, what bench.py
	// measures; needs gridDim % (zblock * G) == 0, else SLABS is used).
	enum Partition : int { SLABS = 0, BLOCK_CYCLIC = 1 };

	explicit MultiVoxelizer(const std::vector<int>& devices) : m_devices(devices) {}
	~MultiVoxelizer() { release(); }
	MultiVoxelizer(const MultiVoxelizer&) = delete;
	MultiVoxelizer& operator=(const MultiVoxelizer&) = delete;

	// Load the OBJ on the host, then as InitFromArrays (Content/Voxelizer.cpp:30-79).
	bool Init(const char* fileName, const float posScale[4] = nullptr)
	{
		float* vb = nullptr; uint32_t* ib = nullptr; uint32_t numVerts = 0, numIndices = 0; float aabb[6];
		if (dxv_obj_load(fileName, &vb, &numVerts, &ib, &numIndices, aabb)) return setError("cannot load OBJ file");
		const bool ok = InitFromArrays(vb, numVerts, ib, numIndices / 3, posScale);
		dxv_free(vb); dxv_free(ib);
		return ok;
	}

	// Device 0 of the set uploads the mesh and builds the LBVH and the candidate lists; the scene blob is broadcast to the
	// other devices once (ncclBroadcast, one group call over all communicators of this process) and imported there.
	bool InitFromArrays(const float* vb, uint32_t numVerts, const uint32_t* ib, uint32_t numTris, const float posScale[4] = nullptr)
	{
		(void)posScale;						// display only in the reference (Content/Voxelizer.cpp:84-87)
		if (m_devices.empty()) return setError("empty device set");
		if (!m_ready && !create()) return false;
		dxv_ctx* root = m_ctx[0];
		if (dxv_set_mesh(root, vb, numVerts, ib, numTris) || dxv_build(root) ||
			(m_gridHint ? dxv_build_lists_for_grid(root, m_gridHint) : dxv_build_lists(root))) return ctxError(0);
		const size_t bytes = dxv_scene_bytes(root);
		if (!bytes) return setError("no scene to broadcast");
		// one blob buffer per device; the root exports into its own
		for (size_t i = 0; i < m_devices.size(); ++i) {
			if (m_blobBytes[i] >= bytes) continue;
			if (!hipOk(hipSetDevice(m_devices[i]), "hipSetDevice")) return false;
			if (m_blob[i]) (void)hipFree(m_blob[i]);
			m_blob[i] = nullptr; m_blobBytes[i] = 0;
			if (!hipOk(hipMalloc(&m_blob[i], bytes), "hipMalloc(blob)")) return false;
			m_blobBytes[i] = bytes;
		}
		if (dxv_scene_export(root, m_blob[0], bytes)) return ctxError(0);
		// the only collective of the whole path, once per mesh (timed: host clock from the group call to the last stream's end)
		const auto t0 = std::chrono::steady_clock::now();
		if (!ncclOk(ncclGroupStart(), "ncclGroupStart")) return false;
		for (size_t i = 0; i < m_devices.size(); ++i)
			if (!ncclOk(ncclBroadcast(m_blob[0], m_blob[i], bytes, ncclUint8, 0, m_comm[i], m_stream[i]), "ncclBroadcast")) { (void)ncclGroupEnd(); return false; }
		if (!ncclOk(ncclGroupEnd(), "ncclGroupEnd")) return false;
		for (size_t i = 0; i < m_devices.size(); ++i)
			if (!hipOk(hipSetDevice(m_devices[i]), "hipSetDevice") || !hipOk(hipStreamSynchronize(m_stream[i]), "hipStreamSynchronize")) return false;
		m_broadcastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		// what arrived is what was sent: every device sums its copy, nobody imports a blob that differs from the root's
		m_checksums.assign(m_devices.size(), 0);
		for (size_t i = 0; i < m_devices.size(); ++i)
			if (dxv_scene_checksum(m_ctx[i], m_blob[i], bytes, &m_checksums[i])) return ctxError(i);
		for (size_t i = 1; i < m_devices.size(); ++i)
			if (m_checksums[i] != m_checksums[0]) return setError(("device " + std::to_string(m_devices[i]) + ": the broadcast blob's checksum differs from the root's").c_str());
		for (size_t i = 1; i < m_devices.size(); ++i)
			if (dxv_scene_import(m_ctx[i], m_blob[i], bytes)) return ctxError(i);
		m_sceneBytes = bytes;
		// with a grid hint every device's share of that grid is prepared now (dxv_prepare_launch*): Init-time structure, like the lists
		if (m_gridHint && !prepare(m_gridHint, m_partitionHint, m_zblockHint)) return false;
		return true;
	}

	// The hot call (Content/Voxelizer.cpp:351-369) on every device's share of the grid; no collective.
	// zblock = 0: 8 slices, 4 from eight devices on (one layer of 4^3-voxel bricks: the finer deal balances eight shares better,
	// slowest share 0.119 instead of 0.127 ms at 512^3 -- what bench.py does)
	bool Voxelize(uint32_t gridDim, Mode mode = REFERENCE, Partition partition = BLOCK_CYCLIC, uint32_t zblock = 0)
	{
		if (!zblock) zblock = m_devices.size() >= 8 ? 4u : 8u;
		if (!m_sceneBytes) return setError("Voxelize before Init");
		const uint32_t G = static_cast<uint32_t>(m_devices.size());
		const bool cyclic = partition == BLOCK_CYCLIC && zblock && !(zblock & (zblock - 1u)) && gridDim % (zblock * G) == 0;
		m_gridDim = gridDim; m_cyclic = cyclic; m_zblock = zblock;
		for (uint32_t g = 0; g < G; ++g) {
			int rc;
			if (cyclic) rc = dxv_voxelize_interleaved_async(m_ctx[g], gridDim, mode, g, G, zblock);
			else {
				uint32_t z0, nz;
				slab(gridDim, g, G, z0, nz);
				m_z0[g] = z0; m_nz[g] = nz;
				rc = nz ? dxv_voxelize_async(m_ctx[g], gridDim, mode, z0, nz) : 0;
			}
			if (rc) return ctxError(g);
		}
		for (uint32_t g = 0; g < G; ++g)
			if ((cyclic || m_nz[g]) && dxv_sync(m_ctx[g])) return ctxError(g);
		return true;
	}

	// The whole grid on the host (x fastest, then y top to bottom, then z: hlsl:64-67), reassembled from the devices' shares.
	bool Download(std::vector<uint8_t>& grid)
	{
		if (!m_gridDim) return setError("Download before Voxelize");
		const uint32_t N = m_gridDim, G = static_cast<uint32_t>(m_devices.size());
		const size_t plane = static_cast<size_t>(N) * N;
		grid.assign(plane * N, 0);
		std::vector<uint8_t> part;
		for (uint32_t g = 0; g < G; ++g) {
			if (!m_cyclic && !m_nz[g]) continue;
			part.resize(dxv_grid_bytes(m_ctx[g]));
			if (dxv_grid_download(m_ctx[g], part.data(), part.size())) return ctxError(g);
			if (!m_cyclic) { memcpy(grid.data() + plane * m_z0[g], part.data(), part.size()); continue; }
			const uint32_t nzLocal = N / G;
			for (uint32_t lz = 0; lz < nzLocal; ++lz)			// local slice -> global slice (dxv.h, dxv_voxelize_interleaved)
				memcpy(grid.data() + plane * GlobalSlice(lz, g, G, m_zblock), part.data() + plane * lz, plane);
		}
		return true;
	}

	bool CountSolid(uint64_t& solid)
	{
		solid = 0;
		for (size_t g = 0; g < m_devices.size(); ++g) {
			if (!m_cyclic && !m_nz[g]) continue;
			uint64_t s = 0;
			if (dxv_grid_count(m_ctx[g], &s)) return ctxError(g);
			solid += s;
		}
		return true;
	}

	bool SetOption(const char* key, int64_t value)
	{
		if (!m_ready && !create()) return false;
		for (size_t g = 0; g < m_devices.size(); ++g) if (dxv_set_option(m_ctx[g], key, value)) return ctxError(g);
		return true;
	}
	bool GetStats(size_t device, dxv_stats& s) const { return device < m_ctx.size() && dxv_get_stats(m_ctx[device], &s) == 0; }
	// the grid size the scene will be launched at (before Init): the root builds the lists' map those launches want (dxv.h,
	// dxv_build_lists_for_grid), so that the other devices do not each rebuild it
	// ... and every device prepares the work queue of its share of that grid after the import (partition / zblock as Voxelize's)
	void SetGridHint(uint32_t gridDim, Partition partition = BLOCK_CYCLIC, uint32_t zblock = 0) { m_gridHint = gridDim; m_partitionHint = partition; m_zblockHint = zblock; }
	double BroadcastMs() const { return m_broadcastMs; }				// the scene broadcast of the last Init, host clock
	uint64_t SceneChecksum() const { return m_checksums.empty() ? 0 : m_checksums[0]; }	// ... and the blob's checksum (equal on every device, or Init failed)
	// global slice of local slice lz of share g in the block-cyclic partition (blocks of zblock slices dealt round-robin over G shares)
	static uint32_t GlobalSlice(uint32_t lz, uint32_t g, uint32_t G, uint32_t zblock) { return (lz / zblock * G + g) * zblock + lz % zblock; }
	size_t DeviceCount() const { return m_devices.size(); }
	size_t SceneBytes() const { return m_sceneBytes; }
	const char* LastError() const { return m_err.c_str(); }

	static void slab(uint32_t N, uint32_t g, uint32_t G, uint32_t& z0, uint32_t& nz)		// contiguous, near-equal split (dxrvoxelizer_amd/slabs.py)
	{
		const uint32_t base = N / G, rem = N % G;
		nz = base + (g < rem ? 1u : 0u);
		z0 = g * base + (g < rem ? g : rem);
	}

protected:
	bool prepare(uint32_t gridDim, Partition partition, uint32_t zblock)
	{
		if (!zblock) zblock = m_devices.size() >= 8 ? 4u : 8u;
		const uint32_t G = static_cast<uint32_t>(m_devices.size());
		const bool cyclic = partition == BLOCK_CYCLIC && !(zblock & (zblock - 1u)) && gridDim % (zblock * G) == 0;
		for (uint32_t g = 0; g < G; ++g) {
			int rc = 0;
			if (cyclic) rc = dxv_prepare_launch_interleaved(m_ctx[g], gridDim, g, G, zblock);
			else {
				uint32_t z0, nz;
				slab(gridDim, g, G, z0, nz);
				if (nz) rc = dxv_prepare_launch(m_ctx[g], gridDim, z0, nz);
			}
			if (rc) return ctxError(g);
		}
		return true;
	}
	bool create()
	{
		const size_t G = m_devices.size();
		m_ctx.assign(G, nullptr); m_stream.assign(G, nullptr); m_comm.assign(G, nullptr); m_blob.assign(G, nullptr);
		m_blobBytes.assign(G, 0); m_z0.assign(G, 0); m_nz.assign(G, 0);
		for (size_t i = 0; i < G; ++i) {
			if (dxv_create(&m_ctx[i], m_devices[i])) return setError(dxv_last_error(nullptr));
			if (!hipOk(hipSetDevice(m_devices[i]), "hipSetDevice") || !hipOk(hipStreamCreateWithFlags(&m_stream[i], hipStreamNonBlocking), "hipStreamCreate")) return false;
			if (dxv_set_stream(m_ctx[i], m_stream[i])) return ctxError(i);		// the context's work and the broadcast share one stream per device
		}
		// one communicator per device, all in this process (RCCL over xGMI between the GPUs of the node)
		if (!ncclOk(ncclCommInitAll(m_comm.data(), static_cast<int>(G), m_devices.data()), "ncclCommInitAll")) return false;
		m_ready = true;
		return true;
	}
	void release()
	{
		for (size_t i = 0; i < m_ctx.size(); ++i) {
			if (m_ctx[i]) dxv_destroy(m_ctx[i]);
			if (i < m_devices.size()) (void)hipSetDevice(m_devices[i]);
			if (m_blob[i]) (void)hipFree(m_blob[i]);
			if (m_comm[i]) (void)ncclCommDestroy(m_comm[i]);
			if (m_stream[i]) (void)hipStreamDestroy(m_stream[i]);
		}
		m_ctx.clear(); m_blob.clear(); m_comm.clear(); m_stream.clear();
		m_ready = false; m_sceneBytes = 0;
	}
	bool setError(const char* msg) { m_err = msg ? msg : ""; return false; }
	bool ctxError(size_t i) { m_err = "device " + std::to_string(m_devices[i]) + ": " + dxv_last_error(m_ctx[i]); return false; }
	bool hipOk(hipError_t e, const char* what) { if (e == hipSuccess) return true; m_err = std::string(what) + ": " + hipGetErrorString(e); return false; }
	bool ncclOk(ncclResult_t r, const char* what) { if (r == ncclSuccess) return true; m_err = std::string(what) + ": " + ncclGetErrorString(r); return false; }

	std::vector<int>			m_devices;
	std::vector<dxv_ctx*>		m_ctx;
	std::vector<hipStream_t>	m_stream;
	std::vector<ncclComm_t>		m_comm;
	std::vector<void*>			m_blob;
	std::vector<size_t>			m_blobBytes;
	std::vector<uint32_t>		m_z0, m_nz;
	std::vector<uint64_t>		m_checksums;
	double						m_broadcastMs = 0.0;
	uint32_t					m_gridHint = 0, m_zblockHint = 0;
	Partition					m_partitionHint = BLOCK_CYCLIC;
	size_t						m_sceneBytes = 0;
	uint32_t					m_gridDim = 0, m_zblock = 8;
	bool						m_cyclic = false, m_ready = false;
	std::string					m_err;
};
