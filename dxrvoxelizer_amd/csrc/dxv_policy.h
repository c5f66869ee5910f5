// dxv_policy.h -- the decisions of the host side as pure functions of plain state: WHEN the candidate lists of the reference rule
// are built and on WHICH cube map, and whether a launch builds its work queue, keeps it, or has the hardware deal it out.
// dxv_lists.hip / dxv_frames.hip carry the decisions out; tests/hostcheck compiles this header for the CPU and
// tests/test_policy.py walks every transition (static scene, dynamic scene, the C-ABI's own rules, options) as a table.
// No HIP, no context: nothing here touches a device.
#pragma once
#include <stdint.h>

namespace dxv {

// ---------------------------------------------------------------------------------------------
// lists: state of a scene's direction-space lists as the host sees it in front of a reference-rule launch
// ---------------------------------------------------------------------------------------------
struct ListsState {
    int optLists;            // option lists: 0 tree walk, 1 by the rules below, 2 from the first launch
    int optListRes;          // option listres: 0 automatic, else the map asked for
    int listOpt;             // the listres option the current lists (or the decision against them) were made with
    int listState;           // 0 not built for this scene, 1 built, -1 over the caps (tree walk)
    uint32_t listRes;        // map of the lists there are
    uint64_t listEntries;
    uint32_t numTris;
    uint32_t launchesOfScene; // reference-rule launches since the scene last changed (build / refit / import)
    bool refitted;           // dxv_refit has run since dxv_build: the mesh is being animated
    bool floorTried;         // the move to the fine map has been made (or tried) for this scene
};
constexpr uint32_t kListsFineMap = 512u;          // the map of a static scene of kListsFineFrom triangles or more
constexpr uint32_t kListsFineFrom = 20000u;
constexpr uint64_t kListsPayFromVoxels = 1ull << 26;   // a FIRST launch of this size may build its lists at once (when the build's estimate agrees)

// base map by triangle count (5 - 10 entries per texel: coarser maps have long lists, finer ones stop fitting the caches)
inline uint32_t lists_base_map(uint32_t numTris, int optListRes)
{
    if (optListRes) return (uint32_t)optListRes;
    return numTris < kListsFineFrom ? 128u : numTris < 3000000u ? 256u : 512u;
}
// deep scenes (over 32 entries per texel: soups) keep coarse maps -- what matters there is the size of the structure
inline bool lists_deep(uint64_t entries, uint32_t R) { return (double)entries > 32.0 * 6.0 * (double)R * (double)R; }

// What to do to the lists in front of a launch of `voxels` voxels (relaunch: the same launch again with a deeper column or after
// withdrawn lists -- it builds nothing).  At most one build per call; the caller applies it, updates the state and asks
// lists_used.
enum class ListsStep {
    none,
    build_if_it_pays,        // a scene's FIRST launch under lists = 1, large enough: count, estimate, build on the base map or decline
    move_to_fine_map,        // lists on a coarser map and the scene is launched AGAIN without a refit in between: a static scene -- once, the 512 map
    build                    // lists wanted and not there (or made for another listres option)
};
inline ListsStep lists_step(const ListsState& s, uint64_t voxels, bool relaunch)
{
    if (relaunch || !s.optLists) return ListsStep::none;
    if (s.optLists == 1 && s.launchesOfScene == 0 && s.listState == 0 && voxels >= kListsPayFromVoxels) return ListsStep::build_if_it_pays;
    if (!s.optListRes && s.listState == 1 && s.listRes < kListsFineMap && !s.refitted && !s.floorTried && s.launchesOfScene > 0 &&
        s.numTris >= kListsFineFrom && !lists_deep(s.listEntries, s.listRes))
        return ListsStep::move_to_fine_map;
    const bool want = s.optLists == 2 || s.launchesOfScene > 0 || s.listState != 0;
    if (want && (s.listState == 0 || s.listOpt != s.optListRes)) return ListsStep::build;
    return ListsStep::none;
}
// ... and whether the launch then goes through the lists (state after the step has been applied)
inline bool lists_used(const ListsState& s, bool relaunch)
{
    if (!s.optLists || s.listState != 1) return false;
    if (relaunch) return s.listOpt == s.optListRes;
    return s.optLists == 2 || s.launchesOfScene > 0 || s.listState != 0;
}
// dxv_build_lists_for_grid (what Init of the host mirrors calls for a static scene): straight to the fine map?
inline bool lists_static_scene_takes_fine_map(const ListsState& s, uint32_t floorNow)
{
    return !s.optListRes && s.numTris >= kListsFineFrom && !s.refitted && !s.floorTried && floorNow < kListsFineMap;
}

// The map a build settles on once the counting pass has said how many entries the map it counted on would hold (0: keep it).
//   oneLaunch: the build must pay for itself on one launch (a first-launch build, a mesh that is being refitted): base map
//   coarser: this count was already a step down from a finer map (a deep scene): only further down from here
inline uint32_t lists_recount_on(uint32_t R, uint64_t entries, bool oneLaunch, int optListRes, bool coarser)
{
    if (optListRes) return 0u;
    if (!coarser && !oneLaunch && R == 256u && !lists_deep(entries, R)) return kListsFineMap;   // a static scene: the fine map beats the 256 map at every grid size
    if (coarser ? R == 256u : lists_deep(entries, R)) {
        if (R != 256u) return 256u;
        if (entries > (320ull << 20)) return 128u;
    }
    return 0u;
}
// a build of this many triangles counts every kListsSampleStride-th of them first and lets the estimate pick the map
constexpr uint32_t kListsSampleFrom = 3000000u, kListsSampleStride = 16u;
inline bool lists_sample_first(uint32_t numTris, int optListRes) { return !optListRes && numTris >= kListsSampleFrom; }
// scenes whose lists would exceed 256 entries per triangle + 64 M, or 2^31, keep the tree walk
inline bool lists_over_the_caps(uint64_t entries, uint32_t numTris) { return entries > 256ull * numTris + (64ull << 20) || entries > 0x7fffffffull; }
// a first-launch build goes on only when what the lists save on THIS launch exceeds what the rest of the build costs
inline bool lists_pay_on_first_launch(uint64_t voxels, uint64_t entries, uint32_t R)
{
    const double perTexel = (double)entries / (6.0 * R * R), depth = perTexel > 10.0 ? perTexel / 10.0 : 1.0;
    return (double)voxels * 1e-8 * depth >= 0.1 + 0.15e-6 * (double)entries;
}

// ---------------------------------------------------------------------------------------------
// far-radius map of a scene WITHOUT lists (the brick test of its tree walks, dirmap_far): made when the scene is launched over the brick
// box a SECOND time without having changed -- a mesh refitted every frame never pays 0.13 ms (1 M triangles) for a map that one
// launch would read and that saves that launch 1 - 5 %
// ---------------------------------------------------------------------------------------------
inline bool far_map_build_now(bool haveForScene, uint32_t boxLaunchesOfScene) { return !haveForScene && boxLaunchesOfScene >= 1u; }

// ---------------------------------------------------------------------------------------------
// work queue: what a launch through the lists does about its queue
// ---------------------------------------------------------------------------------------------
struct QueueState {
    int optPlan;             // 2 every launch builds (default), 1 kept while the launch is the same, (0: no queue -- not asked here)
    int optDispatch;         // a kept queue of known size: 1 dealt out by the hardware, 2 only for partitions of up to 2^25 voxels, 0 never
    bool ptrExposed;         // the caller holds a writable pointer to the frame's grid
    uint64_t keptSig;        // signature of the launch whose queue and zeros the frame still carries (0: none)
    uint64_t lensSig;        // signature of the queue whose lengths a dxv_sync has read (0: none)
    uint32_t queuedBricks;   // ... their sum
    int optPrepared;         // 1: launches of a prepared partition use its queue
    bool prepared;           // the context holds a queue PREPARED for this launch's (lists, grid, partition, queue options)
};
// Persistent waves of a launch through the queue, in sevenths of what the device holds at once (7 waves per SIMD).  A brick of a
// coarse grid looks into a large patch of the map (5.6 R / N texels across), and on a mesh of many small triangles the rays of
// one brick meet dozens of different triangles: seven such waves per SIMD get in each other's way in the vector caches, five or four
// finish the launch sooner (256^3 on the 512 map, nothing carried: torus-1M 0.190 -> 0.166 ms, bunny x16 0.217 -> 0.183, dragon x9
// 0.145 -> 0.138; 128^3: 0.069 -> 0.064 with four).  Meshes of few triangles lose (bunny 256^3: +5 % with five), and so does every
// mesh on a grid as fine as the map (512^3: +12 %): all waves there.  (profiles/r05/ab_waves_by_grid.jsonl; option queuewaves
// overrides.)
constexpr uint32_t kQueueFineMeshFrom = 500000u;
inline uint32_t queue_waves_sevenths(uint32_t numTris, uint32_t R, uint32_t N)
{
    if (numTris < kQueueFineMeshFrom) return 7u;
    if (4ull * N <= R) return 4u;
    if (2ull * N <= R) return 5u;
    return 7u;
}
// prepared_hardware: the queue came from Init (dxv_prepare_launch); the launch clears its grid and the hardware deals the bricks out --
// nothing of the OUTPUT is carried, and what is read (the queue) is structure of the static scene like the lists.  It wins over a
// kept queue (plan = 1) too: same kernel, and no dependence on the frame's last launch.
enum class QueueLaunch { build_and_persistent, kept_persistent, kept_hardware, prepared_hardware };
inline QueueLaunch queue_policy(const QueueState& q, uint64_t sig, uint64_t voxels)
{
    if (q.prepared && q.optPrepared && q.optPlan != 0) return QueueLaunch::prepared_hardware;
    if (q.optPlan == 2 || q.ptrExposed || q.keptSig != sig) return QueueLaunch::build_and_persistent;
    const bool sizeKnown = q.lensSig == sig && q.queuedBricks != 0u;
    if (sizeKnown && (q.optDispatch == 1 || (q.optDispatch == 2 && voxels <= (1ull << 25)))) return QueueLaunch::kept_hardware;
    return QueueLaunch::kept_persistent;
}

} // namespace dxv
