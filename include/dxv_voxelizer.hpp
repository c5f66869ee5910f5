// dxv_voxelizer.hpp -- host-side C++ mirror of the reference's Voxelizer component over the
// C-ABI (include/dxv.h).  Header only; link with libdxv.so.
//
// Reference surface (Content/Voxelizer.h:10-24):
//   bool Init(pCommandList, descriptorTableLib, width, height, rtFormat, dsFormat, uploaders,
//             pGeometry, fileName, posScale);
//   void UpdateFrame(frameIndex, eyePt, viewProj);  void Render(pCommandList, frameIndex, rtv, dsv);
//   protected: void voxelize(pCommandList, frameIndex);   // the hot call, GRID_SIZE = 64 macro
//
// Here: the D3D12-only parameters are gone; `voxelize` is public as Voxelize(gridDim) with the
// grid size promoted from the GRID_SIZE macro (Content/Voxelizer.cpp:8) to a parameter; every
// fallible call returns bool like the reference (XUSG/Core/XUSG.h:12-15) and never throws.
// posScale is accepted and, exactly as in the reference, does not affect voxelisation
// (Content/Voxelizer.cpp:84-87 uses it for display matrices only).
#pragma once
#include "dxv.h"

#include <cstdint>
#include <string>
#include <vector>

class Voxelizer
{
public:
	enum Mode : int { REFERENCE = DXV_MODE_REFERENCE, PARITY = DXV_MODE_PARITY };

	explicit Voxelizer(int device = 0) : m_device(device) {}
	virtual ~Voxelizer() { dxv_destroy(m_ctx); }
	Voxelizer(const Voxelizer&) = delete;
	Voxelizer& operator=(const Voxelizer&) = delete;

	// Load the OBJ, upload VB/IB, extract the bound, build the acceleration structure
	// (Content/Voxelizer.cpp:30-79).  Like the reference's Init (:73: buildAccelerationStructures), it leaves EVERYTHING the
	// launches trace through finished -- the LBVH and the candidate lists of the reference rule on the map a static scene is
	// launched with: the first Voxelize costs what every later one costs.
	// gridDim: the grid the scene will be voxelized at -- the reference's GRID_SIZE, which is a compile-time constant there
	// (Content/Voxelizer.cpp:8) and so is known to its Init as well.  The scene's work queue for that grid is then built here too
	// (dxv_prepare_launch) and every Voxelize(gridDim) is one dispatch behind a clear (Content/Voxelizer.cpp:351-369); 0: launches build
	// their queue themselves.
	bool Init(const char* fileName, const float posScale[4] = nullptr, bool dynamicMesh = false, uint32_t gridDim = 0)
	{
		float* vb = nullptr; uint32_t* ib = nullptr; uint32_t numVerts = 0, numIndices = 0; float aabb[6];
		if (dxv_obj_load(fileName, &vb, &numVerts, &ib, &numIndices, aabb)) return setError("cannot load OBJ file");
		const bool ok = InitFromArrays(vb, numVerts, ib, numIndices / 3, posScale, dynamicMesh, gridDim);
		dxv_free(vb); dxv_free(ib);
		return ok;
	}

	// Same from memory: vb = numVerts x {pos.xyz, nrm.xyz}, ib = 3*numTris indices, both in the
	// layout ObjLoader produces (createVB/createIB, Content/Voxelizer.cpp:115-138).
	// dynamicMesh: the vertices will be replaced and the hierarchy refitted every frame (UpdateVertices*): only the LBVH is
	// built here, and every frame's lists are built for that frame on the coarser map (include/dxv.h, option lists).
	bool InitFromArrays(const float* vb, uint32_t numVerts, const uint32_t* ib, uint32_t numTris,
		const float posScale[4] = nullptr, bool dynamicMesh = false, uint32_t gridDim = 0)
	{
		for (int i = 0; i < 4; ++i) m_posScale[i] = posScale ? posScale[i] : (i == 3 ? 1.0f : 0.0f);
		if (!m_ctx && dxv_create(&m_ctx, m_device)) return setError(dxv_last_error(nullptr));
		if (dxv_set_mesh(m_ctx, vb, numVerts, ib, numTris)) return false;
		if (dxv_build(m_ctx)) return false;
		return dynamicMesh || dxv_build_lists_for_grid(m_ctx, gridDim) == 0;
	}
	// the work queue of another grid size or of a slab, built now (dxv_prepare_launch)
	bool PrepareLaunch(uint32_t gridDim) { return PrepareLaunch(gridDim, 0, gridDim); }
	bool PrepareLaunch(uint32_t gridDim, uint32_t z0, uint32_t nz)
	{
		if (!m_ctx) return setError("PrepareLaunch before Init");
		return dxv_prepare_launch(m_ctx, gridDim, z0, nz) == 0;
	}
	bool InitDynamic(const float* vb, uint32_t numVerts, const uint32_t* ib, uint32_t numTris, const float posScale[4] = nullptr)
	{
		return InitFromArrays(vb, numVerts, ib, numTris, posScale, true);
	}

	// The hot call (Content/Voxelizer.cpp:351-369): whole grid, or slices [z0, z0+nz).
	bool Voxelize(uint32_t gridDim, Mode mode = REFERENCE) { return Voxelize(gridDim, mode, 0, gridDim); }
	bool Voxelize(uint32_t gridDim, Mode mode, uint32_t z0, uint32_t nz)
	{
		if (!m_ctx) return setError("Voxelize before Init");
		return dxv_voxelize(m_ctx, gridDim, mode, z0, nz) == 0;
	}

	// Frames in flight, as the reference's per-frame calls take a frameIndex (Content/Voxelizer.h:20-22) and the
	// component owns FrameCount grids (:24, :110): VoxelizeAsync(frameIndex, ...) launches into that frame's grid on
	// that frame's stream and returns; WaitFrame(frameIndex) waits for it and reports a deferred kernel error.
	// Download / DownloadBits / CountSolid / DeviceGrid / Render then refer to the frame last selected.
	bool SetFrame(uint8_t frameIndex) { return m_ctx ? dxv_set_frame(m_ctx, frameIndex) == 0 : setError("SetFrame before Init"); }
	bool VoxelizeAsync(uint8_t frameIndex, uint32_t gridDim, Mode mode = REFERENCE)
	{
		return SetFrame(frameIndex) && dxv_voxelize_async(m_ctx, gridDim, mode, 0, gridDim) == 0;
	}
	bool WaitFrame(uint8_t frameIndex) { return SetFrame(frameIndex) && dxv_sync(m_ctx) == 0; }
	bool WaitAll() { return m_ctx && dxv_sync_all(m_ctx) == 0; }

	// Dynamic meshes: new vertex data on the same topology, refit of the existing hierarchy.
	bool UpdateVertices(const float* vb, uint32_t numVerts)
	{
		if (!m_ctx) return setError("UpdateVertices before Init");
		return dxv_update_vertices(m_ctx, vb, numVerts) == 0 && dxv_refit(m_ctx) == 0;
	}

	// The two halves on their own, for a loop that hides the upload: VoxelizeAsync(frame i); UploadVertices(frame i + 1) --
	// dxv_update_vertices does not wait for launches in flight, they read the scene and not the vertex buffer --;
	// Refit() (the frame's one host round trip: its kernels and the lists' counting pass queue up behind the launch);
	// VoxelizeAsync(frame i + 1) (queued behind the list build); ...  From a pose already on the GPU: UpdateVerticesDevice.
	bool UploadVertices(const float* vb, uint32_t numVerts)
	{
		if (!m_ctx) return setError("UploadVertices before Init");
		return dxv_update_vertices(m_ctx, vb, numVerts) == 0;
	}
	bool Refit() { return m_ctx ? dxv_refit(m_ctx) == 0 : setError("Refit before Init"); }

	// ... from a device buffer (a mesh animated on the GPU)
	bool UpdateVerticesDevice(const void* deviceVb, uint32_t numVerts)
	{
		if (!m_ctx) return setError("UpdateVerticesDevice before Init");
		return dxv_update_vertices_device(m_ctx, deviceVb, numVerts) == 0 && dxv_refit(m_ctx) == 0;
	}

	// Content/Voxelizer.h:20-22: UpdateFrame(frameIndex, eyePt, viewProj) stores the camera, Render
	// runs voxelize + the ray-cast display pass.  eyePt[3]; viewProj row-major, row vectors
	// (XMFLOAT4X4 of view * proj, DXRVoxelizer.cpp:249-254).
	void UpdateFrame(const float eyePt[3], const float viewProj[16])
	{
		for (int i = 0; i < 3; ++i) m_eyePt[i] = eyePt[i];
		for (int i = 0; i < 16; ++i) m_viewProj[i] = viewProj[i];
	}
	bool Render(uint32_t gridDim, uint32_t width, uint32_t height, std::vector<uint8_t>& rgba)
	{
		if (!Voxelize(gridDim)) return false;
		rgba.resize(static_cast<size_t>(width) * height * 4);
		return dxv_render(m_ctx, m_eyePt, m_viewProj, m_posScale, width, height, rgba.data()) == 0;
	}

	// Result: uint8 occupancy, x fastest, then y (top to bottom), then z.
	bool Download(std::vector<uint8_t>& grid)
	{
		if (!m_ctx) return setError("Download before Init");
		grid.resize(dxv_grid_bytes(m_ctx));
		return dxv_grid_download(m_ctx, grid.data(), grid.size()) == 0;
	}
	// One bit per voxel (voxel 8j+i in bit i of byte j), packed on the device: 8x less PCIe traffic.
	bool DownloadBits(std::vector<uint8_t>& bits)
	{
		if (!m_ctx) return setError("DownloadBits before Init");
		bits.resize(dxv_grid_packed_bytes(m_ctx));
		return dxv_grid_download_packed(m_ctx, bits.data(), bits.size()) == 0;
	}
	const void* DeviceGrid() const { return m_ctx ? dxv_grid_device_ptr_ro(m_ctx) : nullptr; }
	bool CountSolid(uint64_t& solid) { return m_ctx && dxv_grid_count(m_ctx, &solid) == 0; }

	bool GetStats(dxv_stats& s) const { return m_ctx && dxv_get_stats(m_ctx, &s) == 0; }
	const char* LastError() const { return m_ctx && *dxv_last_error(m_ctx) ? dxv_last_error(m_ctx) : m_err.c_str(); }
	dxv_ctx* Context() { return m_ctx; }

	static const uint8_t FrameCount = DXV_FRAME_COUNT; // Content/Voxelizer.h:24: grids (frames) the component owns

protected:
	bool setError(const char* msg) { m_err = msg ? msg : ""; return false; }

	dxv_ctx*	m_ctx = nullptr;
	int			m_device;
	float		m_posScale[4] = { 0.0f, 0.0f, 0.0f, 1.0f };
	float		m_eyePt[3] = { 8.0f, 12.0f, -14.0f };	// DXRVoxelizer.cpp:230
	float		m_viewProj[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
	std::string	m_err;
};
