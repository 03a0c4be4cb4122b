// lsm2d_kernels.h -- HIP kernels of the scan-matching hot path for gfx950 (CDNA4, wave64).
//
//   k_align            the product kernel: ONE workgroup owns ONE alignment for all max_iterations
//                      (MultiAligner2D::compute restated): per iteration and slice it
//                      streams the moving cloud through an LDS polar z-buffer (64-bit ds_min keys),
//                      walks the two canvases bin by bin (registration/correspondence_finder_projective_2d.cpp:55-74),
//                      accumulates the plane-to-plane factor (octave/solver/nicp_post.m:4-26,69-90) in
//                      registers, reduces with wave shuffles + one LDS hop, and lane 0 solves the 3x3
//                      system and right-updates the pose -- no host round trip, no global atomics.
//   k_find_projective  CorrespondenceFinderProjective2f::compute for one (fixed, moving, pose): ordered pairs.
//   k_project_canvas   PointNormal2fProjectorPolar::compute: source index / depth / transformed point per column.
//   k_linearize_*      SE2Plane2PlaneErrorFactor over a correspondence vector (two-stage deterministic reduce).
//   k_repack_cloud     AoS float4 (x,y,nx,ny) -> split xy / normal arrays with even-aligned cloud starts.
#pragma once
#include "lsm2d_device.h"

#include <type_traits>

namespace lsm2d {

static constexpr int kMaxSlices = 4;
// Measured-and-rejected experiments (DESIGN App. A) are compiled only into a -DLSM2D_EXPERIMENTS build of the library (round 5): the second launch form of
// a culled batch (k_first_iteration / k_balance_only), the row-major culled stream ("cull" 2), the round-3 stream's block-length knob ("cull_block") and the
// A/B option keys of lsm2d_capi.hip.  The shipped library carries none of them; their bit-identity tests run against the experiments build only.
#ifdef LSM2D_EXPERIMENTS
static constexpr bool kExperiments = true;
#else
static constexpr bool kExperiments = false;
#endif
// exact culling of a projective slice's moving cloud (k_align): a thread's chunk of T steps is cut into at most kCullBlocks blocks of B steps
static constexpr int kCullBlocks = 7;
// Round 5 (experiments build only; measured, no gain: DESIGN App. A): a MAP-SIZED chunk (T >= kCullBigT steps: a cloud of half a million points and more) cut
// into 14 blocks instead of 7 -- the same point visits on configs[4] to four digits, 14 KB more LDS.  Which count a SET uses (CloudDev::block_stride, the
// stride of its block_bounds) is decided once, by its largest cloud; every cloud of the set is cut into that many.  The shipped library: 7 everywhere.
static constexpr int kCullBlocksMax = 14, kCullBigT = 512;
LSM2D_HD int cull_blocks_for(int maxT) { return (kExperiments && maxT >= kCullBigT) ? kCullBlocksMax : kCullBlocks; }
LSM2D_HD int cull_block_steps(int T, int nbs = kCullBlocks) { return 2 * ((T + 2 * nbs - 1) / (2 * nbs)); }      // even; ceil(T / B) <= nbs for every T >= 1
// Wave priority by progress.  The SIMD arbitrates by priority, then AGE: with equal priorities the oldest two waves of a SIMD run
// at full single-wave speed and the younger workgroups of a CU wait (tools/occupancy_probe.py: lifetimes 0.97 .. 2.19 ms in one
// launch), so the last workgroup of a CU ends up alone, with nobody to issue under its barriers, bin walks and solves.  A
// workgroup that lowers its priority as it advances lets the ones behind it catch up: all of a CU's workgroups finish together.
// 0 off (1.86 ms on configs[1]), 1 quarters of the iterations (1.69), 2 halving intervals -- 1/2, 3/4, 7/8 (1.65).
#ifndef LSM2D_SEQ_WALK_PRIO
#define LSM2D_SEQ_WALK_PRIO 1      // k_align_seq: the walking wave at top priority while it walks (configs[1], "sum_order" 1: 0.954 -> 0.937 ms; profiles/r06/sum_order_walker_quads_ab_r06.txt)
#endif
#ifndef LSM2D_PRIO_BY_PROGRESS
#define LSM2D_PRIO_BY_PROGRESS 2
#endif
#ifndef LSM2D_ALIGN_BLOCK
#define LSM2D_ALIGN_BLOCK 512
#endif
static constexpr int kAlignBlock = LSM2D_ALIGN_BLOCK;
static constexpr int kFindBlock = 1024;

#include "lsm2d_k_search.h"
#include "lsm2d_k_structures.h"
#include "lsm2d_k_kdbuild.h"
#include "lsm2d_k_align_args.h"
#include "lsm2d_k_align.h"
#include "lsm2d_k_placement.h"
#include "lsm2d_k_align_pair.h"
#include "lsm2d_k_split_finder.h"
#include "lsm2d_k_mapping.h"
#include "lsm2d_k_layout.h"

}  // namespace lsm2d
