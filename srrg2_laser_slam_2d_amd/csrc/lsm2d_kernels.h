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
#ifndef LSM2D_PRIO_BY_PROGRESS
#define LSM2D_PRIO_BY_PROGRESS 2
#endif
#ifndef LSM2D_ALIGN_BLOCK
#define LSM2D_ALIGN_BLOCK 512
#endif
static constexpr int kAlignBlock = LSM2D_ALIGN_BLOCK;
static constexpr int kFindBlock = 1024;

// Uniform search grid over every cloud of a set (NN finder): cells of side h >= max_distance, points
// counting-sorted by cell.  Replaces the KDTree the reference rebuilds in reset()
// (registration/correspondence_finder_kd_tree_2d.cpp:31-38).
struct GridMeta { float minx, miny, inv_h, h; int32_t gw, gh, cell_base, pad; };
struct GridDev {
  const GridMeta* meta;        // [n_clouds]
  const int32_t*  cell_start;  // per cloud: gw*gh+1 entries from meta.cell_base (positions relative to the cloud)
  const int32_t*  sorted_idx;  // [padded total] original point index, cloud-relative, grouped by cell
  const float2*   sorted_xy;   // [padded total] coordinates in the same order
  const float2*   sorted_nrm;  // [padded total] normals in the same order: the fused aligner takes a match's normal from where the search found the point
};
// (Round 4, measured and dropped: the cell table COMPRESSED to its occupied cells -- a map's points lie on walls, 2 % of the 3.6 M cells of a 100k-point
// map's grid hold one; per block of 64 cells a 16-byte record {occupancy mask, start, index} + one word per occupied cell, 1.3 MB instead of 14 MB, L2-resident --
// took configs[1] role B from 1.72 to 2.12 ms: the record and the word behind it are two dependent requests to L2 where the dense table needs one line, and
// this search is bound by the L2's request rate, not by the misses of the dense table: DESIGN App. A.)

// Distance map over every cloud of a set (CorrespondenceFinderNN2D, registration/correspondence_finder_nn_2d.cpp):
// parent[r*cols + c] = the nearest fixed point's pixel within max_distance as (squared pixel distance << gbits | lowest point index in
// that pixel) -- the form the scatter build (k_distmap_stamp) takes its minimum over -- or -1.
struct DistMeta { float lx, ly, inv_res, half_pad; int32_t rows, cols; long long base; int32_t gbits, gmask; };
struct DistDev { const DistMeta* meta; const int32_t* parent; };

LSM2D_DEV int distmap_lookup(const DistMeta& d, const int32_t* __restrict__ parent, float qx, float qy) {
  const float gx = (qx - d.lx) * d.inv_res + d.half_pad, gy = (qy - d.ly) * d.inv_res + d.half_pad;
  if (!(gx >= 0.0f && gy >= 0.0f && gx < (float) d.rows && gy < (float) d.cols)) return -1;
  const int v = parent[d.base + (long long) (int) gx * d.cols + (int) gy];
  return v < 0 ? -1 : (v & d.gmask);
}

// The reference's own search structure (LSM2D_FINDER_KDTREE): KDTree2D(coordinates, max_leaf_range, min_leaf_points) built in
// CorrespondenceFinderKDTree2D::reset() (registration/correspondence_finder_kd_tree_2d.cpp:31-38) and searched by findNeighbor (.cpp:18-19),
// restated as SURVEY.md App. A.4 believes upstream implements them (the CPU restatement's kd_build_node / kd_find mirror it).  One tree per
// cloud of the set.  Node k of cloud c lives at meta[c].node_base + k (node 0 = root; the two children of a node are adjacent; nodes of a
// level come before the nodes of the next), ONE 32-byte record -- a descent touches one cache line per level: (mx, my, nx, ny) = mean and
// unit normal of the splitting plane; link_x >= 0: the LEFT child's id (the right one is link_x + 1); link_x < 0: a leaf holding the points
// [-1 - link_x, link_y) of the cloud's leaf arrays.  leaf_xy / leaf_idx: the cloud's coordinates / original indices permuted into leaf
// order (ascending original index inside a leaf, as the reference's stable partition leaves them) at the cloud's own offset start[c].
struct KdMeta { int32_t node_base, n_nodes, pad0, pad1; };
struct __attribute__((aligned(32))) KdNode { float mx, my, nx, ny; int32_t link_x, link_y, pad0, pad1; };
struct KdDev {
  const KdMeta*  meta;       // [n_clouds]
  const KdNode*  nodes;      // [total nodes]
  const float2*  leaf_xy;    // [padded total]
  const int32_t* leaf_idx;   // [padded total]
  const float2*  leaf_nrm;   // [padded total] the normals in the same order: the fused aligner takes a match's point and normal from where the
                             // leaf scan found it -- no detour through the original index (two requests to L2 fewer per query)
};

// findNeighbor as the reference calls it (correspondence_finder_kd_tree_2d.cpp:18-19): descend to the ONE leaf on the query's side of
// every splitting plane (no backtracking), scan it for the nearest point with squared distance < md2; first point wins ties, i.e. the
// lowest original index; none -> -1 (.cpp:21).  The operation sequence is the CPU restatement's kd_find: the plane test is two products
// and a sum, NOT fused (the library is built with -ffp-contract=off); the distance is the fused form every finder of this library uses.
// lds_nodes > 0: the first lds_nodes nodes of the tree (its top levels) are staged in LDS (l_plane / l_link) -- a descent pays one LDS
// round trip per level up there instead of one trip to L2.  The leaf is read two points per 16-byte load, two loads in flight (leaf
// arrays start 16-byte aligned at even positions); the winner's original index is fetched once, at the end.
// XyT / IdxT: float2 / int32_t for the arrays in global memory; a scan-sized cloud's leaf arrays staged in LDS use uint16_t indices.
// kd_query_pos: the winner's POSITION in the leaf arrays (-1: none) and its coordinates; kd_query: its original index.
// kAllLds: the WHOLE tree is staged (scan-sized clouds): no node ever comes from global memory, no range checks in the descent.
template <bool kAllLds = false>
LSM2D_DEV int kd_query_pos(const KdNode* __restrict__ nodes, const float2* __restrict__ lxy, float qx, float qy, float md2, float2& best_xy,
                           const float4* l_plane = nullptr, const int2* l_link = nullptr, int lds_nodes = 0) {
  int k = 0;
  int2 L;
  if (kAllLds || lds_nodes > 0) L = l_link[0]; else { const int4 w = reinterpret_cast<const int4*>(nodes)[1]; L = make_int2(w.x, w.y); }
  while (L.x >= 0) {
    float4 P;
    if (kAllLds || k < lds_nodes) P = l_plane[k]; else P = reinterpret_cast<const float4*>(nodes)[2 * k];
    const float t = (qx - P.x) * P.z + (qy - P.y) * P.w;
    k = L.x + (t < 0.0f ? 0 : 1);
    if (kAllLds || k < lds_nodes) L = l_link[k]; else { const int4 w = reinterpret_cast<const int4*>(nodes)[2 * k + 1]; L = make_int2(w.x, w.y); }
  }
  const int b = -1 - L.x, e = L.y;
  int bestpos = -1; float bd = md2;
  auto consider = [&](int j, float px, float py) {
    const float dx = px - qx, dy = py - qy;
    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
    if (d2 < bd) { bd = d2; bestpos = j; best_xy = make_float2(px, py); }
  };
  int j = b;
  if (j < e && (j & 1)) { const float2 p = lxy[j]; consider(j, p.x, p.y); ++j; }      // up to an even position: pairs are 16-byte aligned from here
  for (; j + 3 < e; j += 4) {
    const float4 v0 = *reinterpret_cast<const float4*>(lxy + j), v1 = *reinterpret_cast<const float4*>(lxy + j + 2);
    consider(j, v0.x, v0.y); consider(j + 1, v0.z, v0.w); consider(j + 2, v1.x, v1.y); consider(j + 3, v1.z, v1.w);
  }
  if (j + 1 < e) { const float4 v = *reinterpret_cast<const float4*>(lxy + j); consider(j, v.x, v.y); consider(j + 1, v.z, v.w); j += 2; }
  if (j < e) { const float2 p = lxy[j]; consider(j, p.x, p.y); }
  return bestpos;
}
template <typename IdxT = int32_t>
LSM2D_DEV int kd_query(const KdNode* __restrict__ nodes, const float2* __restrict__ lxy, const IdxT* __restrict__ lidx, float qx, float qy, float md2) {
  float2 bxy;
  const int pos = kd_query_pos(nodes, lxy, qx, qy, md2, bxy);
  return pos >= 0 ? (int) lidx[pos] : -1;
}

struct CloudDev {            // device view of a cloud set
  const float2* xy;          // [padded total] coordinates
  const float2* nrm;         // [padded total] normals
  const int32_t* start;      // [n_clouds] first (even) padded index of each cloud
  const int32_t* count;      // [n_clouds] points per cloud
  const int32_t* index;      // [n_alignments] cloud chosen per alignment, or nullptr
  int32_t n_clouds;
  const float4* lane_xy;     // lane-chunked copy of xy for the k_align streaming pass (see project_cloud_lanes), or nullptr
  const long long* lane_start; // [n_clouds] first float4 slot of each cloud in lane_xy
  const int32_t* lane_T;     // [n_clouds] steps per thread
  const float4* lane_bounds; // [n_clouds][kAlignBlock] bounding circle (cx, cy, rho; rho < 0: no points) of the chunk each thread owns, or nullptr
  const float4* block_bounds; // [n_clouds][block_stride][kAlignBlock] the same per BLOCK of a chunk (block b of chunk g = its steps [b B, (b + 1) B), B = cull_block_steps(T, block_stride)), or nullptr
  int32_t block_stride;       // blocks per chunk of this set: kCullBlocks, or kCullBlocksMax when it holds a map-sized cloud (cull_blocks_for)
  const float4* aos;         // [padded total] (x, y, nx, ny) rows next to xy / nrm -- one 16-byte gather per z-buffer winner in k_align's bin walk -- or nullptr
  const float4* tile_bounds; // bounding circle of every TILE of 64 consecutive points of every cloud (k_tile_bounds), or nullptr: what the point-query
  const int32_t* tile_start; //   finders' culling tests; [n_clouds] first tile of each cloud
  GridDev grid;              // valid only when the slice uses the NN finder on this (fixed) cloud
  DistDev dist;              // valid only when the slice uses the distance-map finder on this (fixed) cloud
  KdDev kd;                  // valid only when the slice uses the KD-tree finder on this (fixed) cloud
};

// exact nearest neighbour of q among the cloud's points within sqrt(md2); ties -> lowest index
// (SURVEY.md App. D.2).  Cells are SMALLER than max_distance (h >= max_distance/64, see ensure_grid): the search visits the
// (2k+1)^2 block around q for k = 1, 2, 4, ... and stops as soon as the best distance is strictly inside the
// block (every point outside it is farther than k*h, so it can neither win nor tie) or the block covers
// max_distance.  Converged ICP queries finish in the first 3x3 block.
#ifndef LSM2D_NN_GROUP
#define LSM2D_NN_GROUP 4
#endif
static constexpr int kNNGroup = LSM2D_NN_GROUP;      // lanes that cooperate on one query on a dense fixed cloud: they read 4 consecutive candidates (32 bytes);
                                        // measured on configs[1] role B with ~13 points per cell: 2 lanes 2.81 ms, 4 lanes 2.48, 8 lanes 2.76, 16 lanes 3.92;
                                        // again with two candidates per trip: 2.53 / 2.36 / 2.73 (the oracle's device-order mode encodes 4)

// CellT / IdxT: int32_t for the tables in global memory, uint16_t for a scan-sized cloud's tables staged in LDS (k_align)
template <int group, typename CellT = int32_t, typename IdxT = int32_t>
LSM2D_DEV int nn_query(const GridMeta& g, const CellT* __restrict__ cell_start, const IdxT* __restrict__ sidx,
                       const float2* __restrict__ sxy, float qx, float qy, float md, float md2, int sub) {
  // group == kNNGroup: every lane of a group of kNNGroup consecutive lanes calls this with the SAME query and its own
  // `sub` in [0, kNNGroup) (dense fixed clouds: a scan point has ~100 map points in its 3x3 block); group == 1: one lane per
  // query (sparse fixed clouds, where most queries find an empty block).  Candidates are strided over the group (coalesced loads instead of 64 private streams per wave),
  // the group's (d2, index) minimum is combined with three xor-shuffles per block level, so control flow is uniform
  // inside a group and every lane returns the same answer.
  const float fx = __builtin_floorf((qx - g.minx) * g.inv_h), fy = __builtin_floorf((qy - g.miny) * g.inv_h);
  const float reach = __builtin_ceilf(md * g.inv_h * 1.002f);                 // cells max_distance can span (0.2 % fp slack)
  if (!(fx >= -reach && fx <= (float) g.gw + reach && fy >= -reach && fy <= (float) g.gh + reach)) return -1;
  const int cx = (int) fx, cy = (int) fy, kmax = (int) reach;
  int best = -1; float bd = 3.402823466e+38f;
  for (int k = 1;; k *= 2) {
    if (k > kmax) k = kmax;
    const int x0 = cx - k < 0 ? 0 : cx - k, x1 = cx + k > g.gw - 1 ? g.gw - 1 : cx + k;
    const int y0 = cy - k < 0 ? 0 : cy - k, y1 = cy + k > g.gh - 1 ? g.gh - 1 : cy + k;
    if (x0 <= x1 && y0 <= y1) {
      // a candidate's original index is only needed when it improves on or ties with the best so far (ties -> lowest index)
      auto consider = [&](int t, float2 p) {
        const float dx = p.x - qx, dy = p.y - qy;
        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
        if (d2 <= md2 && d2 <= bd) {
          const int i = (int) sidx[t];
          if (d2 < bd || i < best || best < 0) { bd = d2; best = i; }
        }
      };
      // two candidates per trip, both loads in flight (the (d2, index) minimum does not depend on the order of the candidates).
      // Measured on configs[1] (A/B on one box, tools/variant_bench.sh): role B (4 lanes per query, tables in global memory) 2.63 ms
      // one per trip, 2.34 two per trip in the array form below, 2.51 in the straight form; role A (one lane per query, tables in
      // LDS) 9.62 / 9.40 / 9.18; three or four per trip lose on both (registers).
      auto scan_row = [&](int s, int e) {
        if (group > 1) {
          for (int t = s + sub; t < e; t += 2 * group) {
            float2 p[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int tu = t + u * group; p[u] = sxy[tu < e ? tu : t]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int tu = t + u * group; if (tu < e) consider(tu, p[u]); }
          }
        } else {
          for (int t = s + sub; t < e; t += 2 * group) {
            const int t1 = t + group; const bool h1 = t1 < e;
            const float2 p0 = sxy[t], p1 = sxy[h1 ? t1 : t];
            consider(t, p0);
            if (h1) consider(t1, p1);
          }
        }
      };
      for (int yy = y0; yy <= y1; ++yy) scan_row((int) cell_start[yy * g.gw + x0], (int) cell_start[yy * g.gw + x1 + 1]);
    }
    if (group > 1)
#pragma unroll
    for (int o = 1; o < kNNGroup; o <<= 1) {                     // lexicographic (d2, index) minimum over the group
      const float od = __shfl_xor(bd, o, 64); const int oi = __shfl_xor(best, o, 64);
      if (oi >= 0 && (best < 0 || od < bd || (od == bd && oi < best))) { bd = od; best = oi; }
    }
    if (k >= kmax) break;
    // q sits in cell (cx,cy): anything outside the block is at least k*h away (0.998: fp slack of the cell assignment)
    const float inside = (float) k * g.h * 0.998f;
    if (best >= 0 && bd < inside * inside) break;
  }
  return best;
}

// The same search keeping the winner's POSITION in the sorted arrays (-1: none), for tables in global memory (k_align<..., kNNGlobal>): a
// candidate's original index is read only to break an exact tie of distances (ties -> lowest index) -- the one index read per improving
// candidate of nn_query is gone -- and whoever needs the winner's coordinates or normal reads sorted_xy / sorted_nrm there once.  The group's
// minimum goes through DPP moves inside the quad (no LDS crossbar).
template <int o> LSM2D_DEV int quad_xor(int v) {      // lane ^ 1 or lane ^ 2 inside a quad
  static_assert(o == 1 || o == 2, "inside a quad");
  return __builtin_amdgcn_update_dpp(0, v, o == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true);
}
// qc: this query's cache row in LDS (8 words: cell x, y and the candidate ranges of the three rows of its 3 x 3 block), or nullptr.  Between two
// iterations of an alignment a query moves by less than a cell more often than not: its block's ranges are then read from LDS instead of
// six entries of the cell table (the one structure of this search that misses the L2s: 14 MB for a 100k-point map).
template <int group>
LSM2D_DEV int nn_query_pos(const GridMeta& g, const int32_t* __restrict__ cell_start, const int32_t* __restrict__ sidx,
                           const float2* __restrict__ sxy, float qx, float qy, float md, float md2, int sub, int* qc = nullptr) {
  static_assert(group == 1 || group == 2 || group == 4, "a group is (part of) a quad");
  const float fx = __builtin_floorf((qx - g.minx) * g.inv_h), fy = __builtin_floorf((qy - g.miny) * g.inv_h);
  const float reach = __builtin_ceilf(md * g.inv_h * 1.002f);
  if (!(fx >= -reach && fx <= (float) g.gw + reach && fy >= -reach && fy <= (float) g.gh + reach)) return -1;
  const int cx = (int) fx, cy = (int) fy, kmax = (int) reach;
  int best = -1; float bd = 3.402823466e+38f;
  for (int k = 1;; k *= 2) {
    if (k > kmax) k = kmax;
    const int x0 = cx - k < 0 ? 0 : cx - k, x1 = cx + k > g.gw - 1 ? g.gw - 1 : cx + k;
    const int y0 = cy - k < 0 ? 0 : cy - k, y1 = cy + k > g.gh - 1 ? g.gh - 1 : cy + k;
    if (x0 <= x1 && y0 <= y1) {
      auto consider = [&](int t, float2 p) {
        const float dx = p.x - qx, dy = p.y - qy;
        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
        if (d2 <= md2 && d2 <= bd) {
          bool take = d2 < bd || best < 0;
          if (!take && t != best) take = sidx[t] < sidx[best];      // an exact tie: the lower original index wins
          if (take) { bd = d2; best = t; }
        }
      };
      // (measured and dropped: two candidates per 16-byte load with the odd head and tail on one lane each, 1.91-1.93 ms against 1.81 for this form)
      auto scan_row = [&](int s, int e) {
        for (int t = s + sub; t < e; t += 2 * group) {
          float2 p[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) { const int tu = t + u * group; p[u] = sxy[tu < e ? tu : t]; }
#pragma unroll
          for (int u = 0; u < 2; ++u) { const int tu = t + u * group; if (tu < e) consider(tu, p[u]); }
        }
      };
      if (qc && k == 1) {
        const int4 c0 = *reinterpret_cast<const int4*>(qc), c1 = *reinterpret_cast<const int4*>(qc + 4);
        int r[6] = {c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        if (!(c0.x == cx && c0.y == cy)) {                       // a new cell: read the table, remember what it said (every lane of the group reads the same entries)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int yy = y0 + i;
            r[2 * i] = yy <= y1 ? cell_start[yy * g.gw + x0] : 0; r[2 * i + 1] = yy <= y1 ? cell_start[yy * g.gw + x1 + 1] : 0;
          }
          if (sub == 0) { *reinterpret_cast<int4*>(qc) = make_int4(cx, cy, r[0], r[1]); *reinterpret_cast<int4*>(qc + 4) = make_int4(r[2], r[3], r[4], r[5]); }
        }
        scan_row(r[0], r[1]); scan_row(r[2], r[3]); scan_row(r[4], r[5]);
      }
      else for (int yy = y0; yy <= y1; ++yy) scan_row(cell_start[yy * g.gw + x0], cell_start[yy * g.gw + x1 + 1]);
    }
    if (group > 1) {                                             // lexicographic (d2, index) minimum over the group
      auto merge = [&](float od, int oi) {
        bool take = oi >= 0 && (best < 0 || od < bd);
        if (!take && oi >= 0 && best >= 0 && od == bd && oi != best) take = sidx[oi] < sidx[best];
        if (take) { bd = od; best = oi; }
      };
      if (group >= 2) merge(__int_as_float(quad_xor<1>(__float_as_int(bd))), quad_xor<1>(best));
      if (group >= 4) merge(__int_as_float(quad_xor<2>(__float_as_int(bd))), quad_xor<2>(best));
    }
    if (k >= kmax) break;
    const float inside = (float) k * g.h * 0.998f;
    if (best >= 0 && bd < inside * inside) break;
  }
  return best;
}

// bounding box per cloud as CorrespondenceFinderNN2D::_adjustSize computes it (correspondence_finder_nn_2d.cpp:28-43):
// upper bounds start at the smallest positive float (the reference's numeric_limits<float>::min()).
__global__ __launch_bounds__(256) void k_cloud_bbox(const float2* __restrict__ xy, const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ count, float4* __restrict__ out) {
  const int c = blockIdx.x, tid = threadIdx.x, n = count[c];
  const float2* p = xy + start[c];
  __shared__ float s[4][4];
  float lx = 3.402823466e+38f, ly = lx, ux = 1.175494351e-38f, uy = ux;
  for (int i = tid; i < n; i += 256) { const float2 v = p[i]; lx = fminf(lx, v.x); ly = fminf(ly, v.y); ux = fmaxf(ux, v.x); uy = fmaxf(uy, v.y); }
  for (int o = 32; o > 0; o >>= 1) {
    lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64));
    ux = fmaxf(ux, __shfl_xor(ux, o, 64)); uy = fmaxf(uy, __shfl_xor(uy, o, 64));
  }
  if ((tid & 63) == 0) { s[0][tid >> 6] = lx; s[1][tid >> 6] = ly; s[2][tid >> 6] = ux; s[3][tid >> 6] = uy; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) { lx = fminf(lx, s[0][w]); ly = fminf(ly, s[1][w]); ux = fmaxf(ux, s[2][w]); uy = fmaxf(uy, s[3][w]); }
    if (n == 0) { lx = 0.0f; ly = 0.0f; }
    out[c] = make_float4(lx, ly, ux, uy);
  }
}

// lowest goal index per pixel (goals sharing a pixel are equidistant from every pixel, so only the lowest can win)
__global__ void k_distmap_goals(const float2* __restrict__ xy, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                                const DistMeta* __restrict__ meta, int32_t* __restrict__ cellgoal, int cloud0) {
  const int c = cloud0 + blockIdx.y; const DistMeta d = meta[c];
  const float2* p = xy + start[c];
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < count[c]; f += gridDim.x * blockDim.x) {
    const float gx = (p[f].x - d.lx) * d.inv_res + d.half_pad, gy = (p[f].y - d.ly) * d.inv_res + d.half_pad;
    if (!(gx >= 0.0f && gy >= 0.0f && gx < (float) d.rows && gy < (float) d.cols)) continue;
    atomicMin(&cellgoal[d.base + (long long) (int) gx * d.cols + (int) gy], f);
  }
}

// every pixel: nearest goal pixel within mds_px (squared integer pixel distance), ties -> lowest goal index
__global__ __launch_bounds__(256) void k_distmap_fill(const DistMeta* __restrict__ meta, const int32_t* __restrict__ cellgoal,
                                                      int32_t* __restrict__ parent, float mds_px, int R, int cloud0) {
  const int c = cloud0 + blockIdx.y; const DistMeta d = meta[c];
  const long long npx = (long long) d.rows * d.cols;
  for (long long k = blockIdx.x * 256ll + threadIdx.x; k < npx; k += (long long) gridDim.x * 256) {
    const int r = (int) (k / d.cols), cc = (int) (k % d.cols);
    int best = -1, bd = 0x7fffffff;
    for (int dr = -R; dr <= R; ++dr) {
      const int rr = r + dr; if (rr < 0 || rr >= d.rows) continue;
      for (int dc = -R; dc <= R; ++dc) {
        const int c2 = cc + dc; if (c2 < 0 || c2 >= d.cols) continue;
        const int d2 = dr * dr + dc * dc;
        if ((float) d2 > mds_px) continue;
        const int g = cellgoal[d.base + (long long) rr * d.cols + c2];
        if (g != 0x7f7f7f7f && (d2 < bd || (d2 == bd && g < best))) { bd = d2; best = g; }
      }
    }
    parent[d.base + k] = best;
  }
}

// The same map built from the points' side: every point stamps the disc of pixels it can be the nearest goal of with an unsigned
// minimum over (d2 << gbits | index) -- the lexicographic (d2, index) minimum k_distmap_fill gathers, so the two builds agree bit for
// bit -- (2R+1)^2 atomics per POINT instead of (2R+1)^2 reads per PIXEL: a scan's map has ~600 pixels per point (the reference pads
// every side by 75 pixels, correspondence_finder_nn_2d.cpp:28-43).  One wave per point; a point that finds a lower index already in
// its own pixel stops there (that point stamps the same disc).  The map starts as all ones (= -1: nobody within reach).
__global__ __launch_bounds__(256) void k_distmap_stamp(const float2* __restrict__ xy, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                                                       const DistMeta* __restrict__ meta, uint32_t* __restrict__ parent, float mds_px, int R, int cloud0) {
  const int c = cloud0 + blockIdx.y; const DistMeta d = meta[c];
  const int lane = threadIdx.x & 63, n = count[c];
  const float2* p = xy + start[c];
  uint32_t* map = parent + d.base;
  const int side = 2 * R + 1, area = side * side;
  const float inv_side = 1.0f / (float) side;
  for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < n; f += gridDim.x * 4) {
    const float2 v = p[f];
    const float gx = (v.x - d.lx) * d.inv_res + d.half_pad, gy = (v.y - d.ly) * d.inv_res + d.half_pad;
    if (!(gx >= 0.0f && gy >= 0.0f && gx < (float) d.rows && gy < (float) d.cols)) continue;
    const int r = (int) gx, cc = (int) gy;
    uint32_t old = 0;
    if (lane == 0) old = atomicMin(&map[(long long) r * d.cols + cc], (uint32_t) f);      // d2 = 0
    old = (uint32_t) __shfl((int) old, 0, 64);
    if (old < (uint32_t) f) continue;
    for (int k = lane; k < area; k += 64) {
      const int kr = (int) (((float) k + 0.5f) * inv_side);      // k / side, exact for k < 2^20
      const int dr = kr - R, dc = k - kr * side - R;
      const int d2 = dr * dr + dc * dc;
      const int rr = r + dr, c2 = cc + dc;
      if (d2 == 0 || (float) d2 > mds_px || rr < 0 || rr >= d.rows || c2 < 0 || c2 >= d.cols) continue;
      atomicMin(&map[(long long) rr * d.cols + c2], ((uint32_t) d2 << d.gbits) | (uint32_t) f);
    }
  }
}

// One workgroup builds the grid of one cloud: bounding box -> cell size -> counting sort by cell.
struct GridBuildArgs {
  const float2* xy; const int32_t* start; const int32_t* count; int32_t n_clouds;
  float h_min;                  // max_distance / 64 (cells smaller than the gate; the query widens its block as needed)
  const int32_t* cell_base;     // [n_clouds] host-computed: room for gcap^2 + 1 entries per cloud
  const int32_t* gcap;          // [n_clouds] max grid dimension per cloud
  GridMeta* meta; int32_t* cell_start; int32_t* cursor; int32_t* sorted_idx; float2* sorted_xy;
  const float2* nrm; float2* sorted_nrm;      // the normals travel with the points
  int32_t big_threshold;        // clouds of at least this many points only get their bounding box and meta here; the chip-wide
                                // kernels below (k_grid_big_*) do the rest -- one workgroup scanning 3.6 M cells took 5 ms for a 100k-point map
};

LSM2D_DEV float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }      // a value every lane holds alike, moved to an SGPR

LSM2D_DEV int grid_cell_of(const GridMeta& g, float2 p) {
  int cx = (int) floorf((p.x - g.minx) * g.inv_h), cy = (int) floorf((p.y - g.miny) * g.inv_h);
  cx = cx < 0 ? 0 : (cx > g.gw - 1 ? g.gw - 1 : cx); cy = cy < 0 ? 0 : (cy > g.gh - 1 ? g.gh - 1 : cy);
  return cy * g.gw + cx;
}

__global__ __launch_bounds__(1024) void k_grid_build(const GridBuildArgs A) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const int n = A.count[c], base = A.start[c];
  const float2* xy = A.xy + base;
  __shared__ float s_min[2][16], s_max[2][16];
  __shared__ GridMeta s_g;
  __shared__ int s_carry, s_wtot[16];
  // ---- bounding box
  float mnx = 3.402823466e+38f, mny = mnx, mxx = -mnx, mxy = -mnx;
  for (int i = tid; i < n; i += 1024) { const float2 p = xy[i]; mnx = fminf(mnx, p.x); mxx = fmaxf(mxx, p.x); mny = fminf(mny, p.y); mxy = fmaxf(mxy, p.y); }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
    mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
  }
  if ((tid & 63) == 0) { s_min[0][tid >> 6] = mnx; s_min[1][tid >> 6] = mny; s_max[0][tid >> 6] = mxx; s_max[1][tid >> 6] = mxy; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) { mnx = fminf(mnx, s_min[0][w]); mny = fminf(mny, s_min[1][w]); mxx = fmaxf(mxx, s_max[0][w]); mxy = fmaxf(mxy, s_max[1][w]); }
    if (n == 0) { mnx = mny = 0.0f; mxx = mxy = 0.0f; }
    const float cap = (float) A.gcap[c];
    float h = fmaxf(A.h_min, fmaxf(mxx - mnx, mxy - mny) / cap * 1.001f);
    if (!(h > 0.0f)) h = 1.0f;
    GridMeta g; g.minx = mnx; g.miny = mny; g.h = h; g.inv_h = 1.0f / h;
    int gw = (int) floorf((mxx - mnx) * g.inv_h) + 1, gh = (int) floorf((mxy - mny) * g.inv_h) + 1;
    g.gw = gw < 1 ? 1 : (gw > A.gcap[c] ? A.gcap[c] : gw); g.gh = gh < 1 ? 1 : (gh > A.gcap[c] ? A.gcap[c] : gh);
    g.cell_base = A.cell_base[c]; g.pad = 0;
    s_g = g; A.meta[c] = g; s_carry = 0;
  }
  __syncthreads();
  if (n >= A.big_threshold) return;
  const GridMeta g = s_g;
  const int ncell = g.gw * g.gh;
  int32_t* cstart = A.cell_start + g.cell_base; int32_t* cur = A.cursor + g.cell_base;
  for (int i = tid; i <= ncell; i += 1024) cur[i] = 0;
  __syncthreads();
  // ---- histogram
  auto cell_of = [&](float2 p) { return grid_cell_of(g, p); };
  for (int i = tid; i < n; i += 1024) atomicAdd(&cur[cell_of(xy[i])], 1);
  __syncthreads();
  // ---- exclusive scan of the counts, 1024 cells per round, carried in LDS
  for (int c0 = 0; c0 <= ncell; c0 += 1024) {
    const int i = c0 + tid;
    const int v = i < ncell ? cur[i] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < (tid >> 6); ++w) before += s_wtot[w];
    if (i <= ncell) cstart[i] = before + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = before + incl;
    __syncthreads();
  }
  for (int i = tid; i < ncell; i += 1024) cur[i] = cstart[i];
  __syncthreads();
  // ---- scatter (order inside a cell is arbitrary; the query breaks ties by index)
  for (int i = tid; i < n; i += 1024) {
    const float2 p = xy[i];
    const int pos = atomicAdd(&cur[cell_of(p)], 1);
    A.sorted_idx[base + pos] = i; A.sorted_xy[base + pos] = p; A.sorted_nrm[base + pos] = A.nrm[base + i];
  }
}

// ---- the same counting sort for ONE map-sized cloud, over the whole chip: histogram (global atomics on a zeroed cursor table), exclusive
// scan of the cell counts in tiles of kGridTile cells (tile totals -> their scan by one workgroup -> tiles again), scatter.  The order
// of the points inside a cell differs from launch to launch; the query's (d2, index) minimum does not depend on it.
static constexpr int kGridTile = 4096;      // cells per workgroup of the scan: 4 per thread
struct GridBigArgs {
  const float2* xy; const int32_t* start; const int32_t* count; int32_t cloud;
  const GridMeta* meta; int32_t* cell_start; int32_t* cursor; int32_t* tile_sums; int32_t* sorted_idx; float2* sorted_xy;
  const float2* nrm; float2* sorted_nrm;
};

__global__ __launch_bounds__(256) void k_grid_big_hist(const GridBigArgs A) {
  const GridMeta g = A.meta[A.cloud];
  const int n = A.count[A.cloud];
  const float2* xy = A.xy + A.start[A.cloud];
  int32_t* cur = A.cursor + g.cell_base;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&cur[grid_cell_of(g, xy[i])], 1);
}

// exclusive scan of this workgroup's tile of counts; kPhase 0: only the tile's total goes out; kPhase 1: cell_start = scanned tile total
// + position in the tile (entry ncell, the end of the last cell, included) and the cursor restarts from it
template <int kPhase>
__global__ __launch_bounds__(1024) void k_grid_big_scan(const GridBigArgs A) {
  const GridMeta g = A.meta[A.cloud];
  const int ncell = g.gw * g.gh, tid = threadIdx.x;
  const int i0 = blockIdx.x * kGridTile + tid * 4;
  if (blockIdx.x * kGridTile > ncell) { if (kPhase == 0 && tid == 0) A.tile_sums[blockIdx.x] = 0; return; }      // launched for the largest grid the cloud may get
  int32_t* cur = A.cursor + g.cell_base; int32_t* cstart = A.cell_start + g.cell_base;
  __shared__ int s_wtot[16];
  int v[4], sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { v[u] = i0 + u < ncell ? cur[i0 + u] : 0; sum += v[u]; }
  int incl = sum;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
  if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < (tid >> 6); ++w) before += s_wtot[w];
  if (kPhase == 0) {
    if (tid == 1023) A.tile_sums[blockIdx.x] = before + incl;
  } else {
    int run = A.tile_sums[blockIdx.x] + before + incl - sum;
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (i0 + u <= ncell) { cstart[i0 + u] = run; if (i0 + u < ncell) cur[i0 + u] = run; } run += v[u]; }
  }
}

// tile totals -> exclusive scan in place (at most 2048 tiles: grids are capped at 2048 x 2048 cells)
__global__ __launch_bounds__(1024) void k_grid_big_scan_tiles(int32_t* __restrict__ tile_sums, int n_tiles) {
  const int tid = threadIdx.x;
  __shared__ int s_wtot[16];
  const int a = 2 * tid < n_tiles ? tile_sums[2 * tid] : 0, b = 2 * tid + 1 < n_tiles ? tile_sums[2 * tid + 1] : 0;
  int incl = a + b;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
  if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < (tid >> 6); ++w) before += s_wtot[w];
  const int ex = before + incl - (a + b);
  if (2 * tid < n_tiles) tile_sums[2 * tid] = ex;
  if (2 * tid + 1 < n_tiles) tile_sums[2 * tid + 1] = ex + a;
}

__global__ __launch_bounds__(256) void k_grid_big_scatter(const GridBigArgs A) {
  const GridMeta g = A.meta[A.cloud];
  const int n = A.count[A.cloud], base = A.start[A.cloud];
  const float2* xy = A.xy + base;
  int32_t* cur = A.cursor + g.cell_base;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float2 p = xy[i];
    const int pos = atomicAdd(&cur[grid_cell_of(g, p)], 1);
    A.sorted_idx[base + pos] = i; A.sorted_xy[base + pos] = p; A.sorted_nrm[base + pos] = A.nrm[base + i];
  }
}

// ---- KD-tree build (CorrespondenceFinderKDTree2D::reset, registration/correspondence_finder_kd_tree_2d.cpp:31-38) -----------------------------
// The oracle's kd_build_node, level by level over every cloud of the set at once: ONE WAVE owns one node of the current level.
//   mean, covariance   sums over the node's points IN THEIR ORDER (ascending original index -- the partitions are stable), each a plain
//                      SEQUENTIAL fp32 sum as the reference's loop forms it: ((0 + x0) + x1) + ...  A parallel reduction would round
//                      differently, move a splitting plane by an ulp and send a point next to it into the other leaf, so the chain is kept
//                      and made cheap instead: the wave holds 64 consecutive values, one per lane, and runs the chain as a systolic pass --
//                      acc <- rotate_right(acc) + v, 64 times: lane 63 ends with ((carry + v0) + v1) + ... + v63, one DPP add per value and
//                      chain, carry in lane 63 for the next 64 values (kChain 1); kChain 0 is the plain form of the same chain (a v_readlane
//                      and an add per value), kept as the reference the systolic form is tested against on the card.
//   principal axis     closed form of the 2x2 symmetric eigenproblem, IEEE sqrt and divide, every lane alike
//   extents, split     projections on the two axes: minima / maxima do not depend on the order; (p - mean).v < 0 goes left
//   partition          stable, by ballot ranks, chunk after chunk; a child too small to be split again is written straight into the leaf
//                      arrays, the others into the next level's input and queue
// No fused multiply-add anywhere in here: the CPU restatement's build has none, the library is built with -ffp-contract=off.
struct KdBuildArgs {
  const int32_t* start;                              // [n_clouds] first point of each cloud
  const KdMeta*  meta;                               // [n_clouds] node_base
  const float2*  xy_in; const int32_t* idx_in;       // this level's input, cloud-relative positions (idx_in == nullptr: the identity, level 0)
  float2* xy_out; int32_t* idx_out;                  // ranges of the children the next level will process
  KdNode* nodes; int32_t* n_nodes;                   // n_nodes[c]: nodes handed out so far in cloud c's region
  float2* leaf_xy; int32_t* leaf_idx;
  const int4* q_in; int4* q_out; int32_t* q_out_count; int32_t n_items;      // work items: (cloud, node, begin, end)
  const int32_t* n_items_ptr;                        // k_kd_level: the number of items, on the device (nullptr: n_items)
  int32_t io_base, io_node_base;                     // local_io: the cloud's start[c] and node_base, read once by the kernel (not once per node)
  int32_t local_io;                                  // 1: xy_in / idx_in / xy_out / idx_out and n_nodes are ONE cloud's own (k_kd_build_scan keeps them in LDS): no start[c] / [c] offset
  float max_leaf_range; int32_t min_leaf_points;
};

// kLocal (the compact forms, k_kd_build_scan): the level's input and output ranges are in LDS -- say so to the compiler.  Through a plain pointer these were
// FLAT accesses, and a flat load waits for every global store issued before it (one counter for both): each pass of each level stood behind the leaf and node
// records on their way to memory, ~15 k cycles per level whatever the nodes' sizes (clock stamps, 8 levels of a 1081-point scan).
typedef float kd_v2f __attribute__((ext_vector_type(2)));
// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global store in flight (vmcnt(0)) -- the leaf points and node
// records of a level, which nobody reads before the build's last pass -- a trip to memory per barrier, six barriers per level
template <bool kLocal> LSM2D_DEV void kd_barrier() {
  if (kLocal) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  } else __syncthreads();
}
template <bool kLocal> LSM2D_DEV float2 kd_ld2(const float2* p, int i) {
  if (kLocal) { const kd_v2f t = ((const __attribute__((address_space(3))) kd_v2f*) p)[i]; return make_float2(t.x, t.y); }
  return p[i];
}
template <bool kLocal> LSM2D_DEV int kd_ldi(const int32_t* p, int i) {
  if (kLocal) return ((const __attribute__((address_space(3))) int32_t*) p)[i];
  return p[i];
}
template <bool kLocal> LSM2D_DEV void kd_st2(float2* p, int i, const float2& v) {
  if (kLocal) { kd_v2f t; t.x = v.x; t.y = v.y; ((__attribute__((address_space(3))) kd_v2f*) p)[i] = t; }
  else p[i] = v;
}
template <bool kLocal> LSM2D_DEV void kd_sti(int32_t* p, int i, int v) {
  if (kLocal) ((__attribute__((address_space(3))) int32_t*) p)[i] = v;
  else p[i] = v;
}
template <int kChain>
struct SeqSum {      // one sequential fp32 sum over values that arrive 64 at a time, one per lane, in lane order
  float acc = 0.0f;  // kChain 1: lane 63 carries the running sum between chunks; kChain 0: every lane holds it
  LSM2D_DEV void step(float v, int i) {
    if (kChain == 1) acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x13C /* wave_ror:1 */, 0xF, 0xF, false)) + v;
    else acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
  }
  LSM2D_DEV float total() const { return kChain == 1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63)) : acc; }
  // after a LAST chunk of only `cnt` values (cnt wave-uniform, 1 .. 64) that was stepped cnt times: the travelling sum sits in lane cnt - 1
  LSM2D_DEV float total_after(int cnt) const { return kChain == 1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), cnt - 1)) : acc; }
};

// one node of the build, by one wave: `it` = (cloud, node, begin, end).  push(n_next, item_left, item_right): lane 0 hands the children that must be split
// again to the next level's queue (n_next of them: the left one first when both go)
// kCompact: the same passes with one chunk per trip and the chains' steps in a loop of four -- a tenth of the code.  For a kernel that runs ONCE per call on an
// otherwise idle chip (k_kd_build_scan) the instruction fetch of 40 KB of unrolled chains was most of its time (measured: 470 k cycles of wave lifetime
// for 108 k wave-instructions).
template <int kChain, bool kCompact = false, typename Push>
LSM2D_DEV void kd_node(const KdBuildArgs& A, const int4 it, const int lane, Push push) {
  constexpr int G = kCompact ? 1 : 4, kSteps = kCompact ? 4 : 64;
  const int c = __builtin_amdgcn_readfirstlane(it.x), node = __builtin_amdgcn_readfirstlane(it.y);
  const int begin = __builtin_amdgcn_readfirstlane(it.z), end = __builtin_amdgcn_readfirstlane(it.w);
  const int n = end - begin, base = A.local_io ? A.io_base : A.start[c], nbase = A.local_io ? A.io_node_base : A.meta[c].node_base;
  const int iob = A.local_io ? 0 : base;
  const float2* xin = A.xy_in + iob + begin;
  const int32_t* iin = A.idx_in ? A.idx_in + iob + begin : nullptr;
  const u64 lt_mask = (1ull << lane) - 1ull;
  bool split = false; int nl = 0;
  float mx = 0.0f, my = 0.0f, vx = 0.0f, vy = 0.0f;
  if (n >= A.min_leaf_points && n >= 2) {
    // (round 4: the wave works through GROUPS of four chunks of 64 points, the next group's four loads in flight while the chains of the current one
    // run -- one chunk of look-ahead left the top levels waiting on memory: a node of 100 000 points is 1 563 chunks, and its four passes took 2.5 ms.
    // A chunk beyond the node's end is SKIPPED, not added as zeros: the sums see exactly the values they saw before, in the same order.)
    const float2 zero2 = make_float2(0.0f, 0.0f);
    // ---- mean: two sequential chains
    { SeqSum<kChain> sx, sy;
      float2 nx[G];
#pragma unroll
      for (int j = 0; j < G; ++j) { const int k = 64 * j + lane; nx[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
      for (int k0 = 0; k0 < n; k0 += 64 * G) {
        float2 cur[G];
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = nx[j];                          // adding +0 is exact: the tail of the last chunk
#pragma unroll
        for (int j = 0; j < G; ++j) { const int k = k0 + 64 * G + 64 * j + lane; nx[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
#pragma unroll
        for (int j = 0; j < G; ++j) if (k0 + 64 * j < n) {
          if (k0 + 64 * j + 64 <= n) {
#pragma unroll kSteps
            for (int i = 0; i < 64; ++i) { sx.step(cur[j].x, i); sy.step(cur[j].y, i); }
          } else {      // the node's last, partial chunk: as many steps as it has points (most nodes of a tree are such chunks alone: 20 .. 60 points)
            const int cnt = n - (k0 + 64 * j);
#pragma nounroll
            for (int i = 0; i < cnt; ++i) { sx.step(cur[j].x, i); sy.step(cur[j].y, i); }
          }
        }
      }
      const float fn = (float) n;
      const int tail = n & 63;
      mx = (tail ? sx.total_after(tail) : sx.total()) / fn; my = (tail ? sy.total_after(tail) : sy.total()) / fn; }
    // ---- covariance: three sequential chains of unfused products
    float sxx, sxy, syy;
    { SeqSum<kChain> cxx, cxy, cyy;
      float2 nx[G];
#pragma unroll
      for (int j = 0; j < G; ++j) { const int k = 64 * j + lane; nx[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
      for (int k0 = 0; k0 < n; k0 += 64 * G) {
        float2 cur[G];
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = nx[j];
#pragma unroll
        for (int j = 0; j < G; ++j) { const int k = k0 + 64 * G + 64 * j + lane; nx[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
#pragma unroll
        for (int j = 0; j < G; ++j) if (k0 + 64 * j < n) {
          float pxx = 0.0f, pxy = 0.0f, pyy = 0.0f;
          if (k0 + 64 * j + lane < n) { const float dx = cur[j].x - mx, dy = cur[j].y - my; pxx = dx * dx; pxy = dx * dy; pyy = dy * dy; }
          if (k0 + 64 * j + 64 <= n) {
#pragma unroll kSteps
            for (int i = 0; i < 64; ++i) { cxx.step(pxx, i); cxy.step(pxy, i); cyy.step(pyy, i); }
          } else {
            const int cnt = n - (k0 + 64 * j);
#pragma nounroll
            for (int i = 0; i < cnt; ++i) { cxx.step(pxx, i); cxy.step(pxy, i); cyy.step(pyy, i); }
          }
        }
      }
      const int tail = n & 63;
      sxx = tail ? cxx.total_after(tail) : cxx.total(); sxy = tail ? cxy.total_after(tail) : cxy.total(); syy = tail ? cyy.total_after(tail) : cyy.total(); }
    // ---- principal eigenvector of [[sxx, sxy], [sxy, syy]] (closed form, as the oracle writes it)
    const float tr = sxx + syy, df = sxx - syy;
    const float disc = __builtin_sqrtf(df * df + 4.0f * sxy * sxy);
    const float l1 = (tr + disc) / 2.0f;
    if (__builtin_fabsf(sxy) > 0.0f) { vx = l1 - syy; vy = sxy; } else if (sxx >= syy) { vx = 1.0f; vy = 0.0f; } else { vx = 0.0f; vy = 1.0f; }
    const float vn = __builtin_sqrtf(vx * vx + vy * vy);
    if (vn > 0.0f) {
      vx = vx / vn; vy = vy / vn;
      // ---- extents along the two axes and the size of the left part
      float lo1 = 3.402823466e+38f, hi1 = -3.402823466e+38f, lo2 = lo1, hi2 = hi1;
      for (int k0 = 0; k0 < n; k0 += 64 * G) {      // four chunks' loads in flight (minima, maxima and the count do not depend on the order)
        float2 p[G];
#pragma unroll
        for (int j = 0; j < G; ++j) { const int k = k0 + 64 * j + lane; p[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          bool left = false;
          if (k0 + 64 * j + lane < n) {
            const float dx = p[j].x - mx, dy = p[j].y - my;
            const float a = dx * vx + dy * vy, b = -dx * vy + dy * vx;
            lo1 = a < lo1 ? a : lo1; hi1 = a > hi1 ? a : hi1; lo2 = b < lo2 ? b : lo2; hi2 = b > hi2 ? b : hi2;
            left = a < 0.0f;
          }
          nl += __popcll(__ballot(left));
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        lo1 = fminf(lo1, __shfl_xor(lo1, o, 64)); hi1 = fmaxf(hi1, __shfl_xor(hi1, o, 64));
        lo2 = fminf(lo2, __shfl_xor(lo2, o, 64)); hi2 = fmaxf(hi2, __shfl_xor(hi2, o, 64));
      }
      const float e1 = (hi1 - lo1) / 2.0f, e2 = (hi2 - lo2) / 2.0f;
      split = (e1 > e2 ? e1 : e2) >= A.max_leaf_range && nl > 0 && nl < n;
    }
  }
  if (!split) {      // a leaf: its points, in their order, go to their final place
    if (lane == 0) { KdNode nd = {0.0f, 0.0f, 0.0f, 0.0f, -1 - begin, end, 0, 0}; A.nodes[nbase + node] = nd; }
    float2* lxy = A.leaf_xy + base + begin; int32_t* lix = A.leaf_idx + base + begin;
    for (int k = lane; k < n; k += 64) { lxy[k] = kd_ld2<kCompact>(xin, k); lix[k] = iin ? kd_ldi<kCompact>(iin, k) : begin + k; }
    return;
  }
  int left_id = 0;
  if (lane == 0) left_id = kCompact ? (int) __hip_atomic_fetch_add((__attribute__((address_space(3))) int32_t*) A.n_nodes, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : atomicAdd(A.local_io ? A.n_nodes : A.n_nodes + c, 2);
  left_id = __builtin_amdgcn_readfirstlane(left_id);
  const int nr = n - nl;
  // a child that cannot be split again (kd_build_node's first test) is a leaf already: straight into the leaf arrays
  const bool leaf_l = !(nl >= A.min_leaf_points && nl >= 2), leaf_r = !(nr >= A.min_leaf_points && nr >= 2);
  if (lane == 0) {
    { KdNode nd = {mx, my, vx, vy, left_id, 0, 0, 0}; A.nodes[nbase + node] = nd; }
    if (leaf_l) { KdNode nd = {0.0f, 0.0f, 0.0f, 0.0f, -1 - begin, begin + nl, 0, 0}; A.nodes[nbase + left_id] = nd; }
    if (leaf_r) { KdNode nd = {0.0f, 0.0f, 0.0f, 0.0f, -1 - (begin + nl), end, 0, 0}; A.nodes[nbase + left_id + 1] = nd; }
    const int n_next = (leaf_l ? 0 : 1) + (leaf_r ? 0 : 1);
    if (n_next) {
      const int4 il = make_int4(c, left_id, begin, begin + nl), ir = make_int4(c, left_id + 1, begin + nl, end);
      push(n_next, leaf_l ? ir : il, ir);
    }
  }
  float2* oxy_l = leaf_l ? A.leaf_xy + base + begin : A.xy_out + iob + begin;  int32_t* oix_l = leaf_l ? A.leaf_idx + base + begin : A.idx_out + iob + begin;
  float2* oxy_r = leaf_r ? A.leaf_xy + base + begin + nl : A.xy_out + iob + begin + nl;  int32_t* oix_r = leaf_r ? A.leaf_idx + base + begin + nl : A.idx_out + iob + begin + nl;
  int cl = 0, cr = 0;
  for (int k0 = 0; k0 < n; k0 += 64 * G) {      // stable partition: ranks by ballot, chunk after chunk (four chunks' loads in flight)
    float2 p[G]; int src[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = k0 + 64 * j + lane; const bool valid = k < n;
      p[j] = valid ? kd_ld2<kCompact>(xin, k) : make_float2(0.0f, 0.0f);
      src[j] = valid ? (iin ? kd_ldi<kCompact>(iin, k) : begin + k) : 0;
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const bool valid = k0 + 64 * j + lane < n;
      bool left = false;
      if (valid) { const float dx = p[j].x - mx, dy = p[j].y - my; left = dx * vx + dy * vy < 0.0f; }
      const u64 bl = __ballot(valid && left), br = __ballot(valid && !left);
      if (valid) {
        if (left) { const int d = cl + __popcll(bl & lt_mask); if (leaf_l) { oxy_l[d] = p[j]; oix_l[d] = src[j]; } else { kd_st2<kCompact>(oxy_l, d, p[j]); kd_sti<kCompact>(oix_l, d, src[j]); } }
        else { const int d = cr + __popcll(br & lt_mask); if (leaf_r) { oxy_r[d] = p[j]; oix_r[d] = src[j]; } else { kd_st2<kCompact>(oxy_r, d, p[j]); kd_sti<kCompact>(oix_r, d, src[j]); } }
      }
      cl += __popcll(bl); cr += __popcll(br);
    }
  }
}

// Round 4: the same node by a WORKGROUP of four waves, for the top levels of a map-sized cloud (a handful of nodes of 10^4 .. 10^6 points each, one wave per
// node = one wave on the whole chip).  Nothing about the sums changes -- every chain is still ONE sequential fp32 sum over the node's points in their order --
// but every chain gets a wave, hence a SIMD, of its own (two chains interleaved in one wave issue at 4 cycles per instruction plus DPP wait states: 12 cycles
// per point; a chain alone runs at its dependent latency), and the two passes that do not depend on the order (extents + left count, stable partition) are cut
// into four contiguous stretches, one per wave, the partition's ranks offset by the left counts of the stretches before.  Bit-identical to kd_node by
// construction (test_kdtree_finder_bit_exact_both_roles runs maps through both).
template <int kChain, bool kCompact = false, typename F>
LSM2D_DEV float kd_seq_chain(const float2* __restrict__ xin, int n, int lane, F value) {
  constexpr int G = kCompact ? 1 : 4, kSteps = kCompact ? 4 : 64;      // one sequential sum of value(point, in range) over the node, by one wave
  SeqSum<kChain> acc;
  const float2 zero2 = make_float2(0.0f, 0.0f);
  float2 nx[G];
#pragma unroll
  for (int j = 0; j < G; ++j) { const int k = 64 * j + lane; nx[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
  for (int k0 = 0; k0 < n; k0 += 64 * G) {
    float2 cur[G];
#pragma unroll
    for (int j = 0; j < G; ++j) cur[j] = nx[j];
#pragma unroll
    for (int j = 0; j < G; ++j) { const int k = k0 + 64 * G + 64 * j + lane; nx[j] = k < n ? kd_ld2<kCompact>(xin, k) : zero2; }
#pragma unroll
    for (int j = 0; j < G; ++j) if (k0 + 64 * j < n) {
      const float v = value(cur[j], k0 + 64 * j + lane < n);
#pragma unroll kSteps
      for (int i = 0; i < 64; ++i) acc.step(v, i);
    }
  }
  return acc.total();
}
// The same chain with the running sum UNIFORM over the wave: the 64 values of a chunk go through 256 bytes of LDS, every lane reads them back 16 bytes at
// a time (one address for the whole wave: a broadcast) and every lane adds them, in their order, to its own copy of the sum -- plain v_add_f32 on a register
// the previous add wrote, ~6 cycles a step for a wave alone on its SIMD, where the systolic form's add reads its operand through DPP and waits ~12.6 (measured:
// level 0 of a 100k-point map 1.26 ms = 1 770 cycles per 64 points and two passes).  Same values, same order, same roundings: the same sum.
template <typename F>
LSM2D_DEV float kd_seq_chain_lds(const float2* __restrict__ xin, int n, int lane, F value, float* stage /* this wave's 256 floats of LDS, 16-byte aligned */) {
  float acc = 0.0f;
  const float2 zero2 = make_float2(0.0f, 0.0f);
  float2 nx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = 64 * j + lane; nx[j] = k < n ? xin[k] : zero2; }
  // a chunk's sixteen reads are all in flight while the chunk before it is added up (two reads of cover left the chain waiting on the LDS: 1 277 us for
  // level 0 of a 100k-point map, no better than the systolic form)
  auto read16 = [&](int j, float4 (&r)[16]) {
    const float4* b4 = reinterpret_cast<const float4*>(stage + 64 * j);
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = b4[i];
  };
  auto add16 = [&](const float4 (&r)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc = acc + r[i].x; acc = acc + r[i].y; acc = acc + r[i].z; acc = acc + r[i].w; }
  };
  for (int k0 = 0; k0 < n; k0 += 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j) stage[64 * j + lane] = value(nx[j], k0 + 64 * j + lane < n);      // the group's four chunks (those beyond the node's end are never read)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int k = k0 + 256 + 64 * j + lane; nx[j] = k < n ? xin[k] : zero2; }
    __builtin_amdgcn_wave_barrier();
    float4 ra[16], rb[16];
    read16(0, ra);
    if (k0 + 64 < n) read16(1, rb);
    add16(ra);
    if (k0 + 64 < n) {
      if (k0 + 128 < n) read16(2, ra);
      add16(rb);
      if (k0 + 128 < n) {
        if (k0 + 192 < n) read16(3, rb);
        add16(ra);
        if (k0 + 192 < n) add16(rb);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  return acc;
}
// ... and its compact form (see kd_node): a chunk per trip, its sixteen reads issued together, then its 64 adds
template <typename F>
LSM2D_DEV float kd_seq_chain_lds_compact(const float2* __restrict__ xin, int n, int lane, F value, float* stage) {
  float acc = 0.0f;
  float2 nx = lane < n ? kd_ld2<true>(xin, lane) : make_float2(0.0f, 0.0f);
#pragma nounroll
  for (int k0 = 0; k0 < n; k0 += 64) {
    stage[lane] = value(nx, k0 + lane < n);
    const int kn = k0 + 64 + lane;
    nx = kn < n ? kd_ld2<true>(xin, kn) : make_float2(0.0f, 0.0f);
    __builtin_amdgcn_wave_barrier();
    const float4* b4 = reinterpret_cast<const float4*>(stage);
    float4 ra[4], rb[4];      // (a quarter's four reads are in flight while the quarter before it is added: 32 registers -- a 1024-thread workgroup has 128 per lane)
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = b4[i];
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = b4[4 * (q + 1) + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc = acc + ra[i].x; acc = acc + ra[i].y; acc = acc + ra[i].z; acc = acc + ra[i].w; }
      if (q + 2 < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = b4[4 * (q + 2) + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc = acc + rb[i].x; acc = acc + rb[i].y; acc = acc + rb[i].z; acc = acc + rb[i].w; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  return acc;
}
template <int kChain, bool kCompact = false, typename F>
LSM2D_DEV float kd_wide_chain(const float2* __restrict__ xin, int n, int lane, F value, float* stage) {      // kChain 1: through LDS; 0: the plain v_readlane form (the reference on the card)
  if (kChain == 1) return kCompact ? kd_seq_chain_lds_compact(xin, n, lane, value, stage) : kd_seq_chain_lds(xin, n, lane, value, stage);
  return kd_seq_chain<0, kCompact>(xin, n, lane, value);
}
// `tid` is the thread's index inside its GROUP of four waves (0 .. 255); `active` false: a group without a node in this round -- it only keeps the workgroup's
// barriers company.  Every group of a workgroup runs through the same FOUR barriers whatever its node does (a leaf, a node too small to split, no node).
template <int kChain, bool kCompact = false, typename Push>
LSM2D_DEV void kd_node_wide(const KdBuildArgs& A, const int4 it, const int tid, const bool active, Push push, float* sh /* [32] */, int* shi /* [8] */, float* stage_all /* [4][256] */) {
  constexpr int G = kCompact ? 1 : 4;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = __builtin_amdgcn_readfirstlane(it.x), node = __builtin_amdgcn_readfirstlane(it.y);
  const int begin = __builtin_amdgcn_readfirstlane(it.z), end = __builtin_amdgcn_readfirstlane(it.w);
  const int n = active ? end - begin : 0, base = A.local_io ? A.io_base : (active ? A.start[c] : 0), nbase = A.local_io ? A.io_node_base : (active ? A.meta[c].node_base : 0);
  const int iob = A.local_io ? 0 : base;      // (local_io: the level's input and output ranges are the cloud's own -- LDS of k_kd_build_scan -- and start at 0)
  const float2* xin = A.xy_in + iob + begin;
  const int32_t* iin = A.idx_in ? A.idx_in + iob + begin : nullptr;
  const u64 lt_mask = (1ull << lane) - 1ull;
  float* stage = stage_all + 256 * w;
  // the order-free passes: wave w owns the chunks [w Cq, (w + 1) Cq) of the node's ceil(n / 64)
  const int n_chunks = (n + 63) >> 6, Cq = (n_chunks + 3) >> 2;
  const int k_lo = w * Cq * 64, k_hi = (w + 1) * Cq * 64 < n ? (w + 1) * Cq * 64 : n;
  bool split = false; int nl = 0;
  float mx = 0.0f, my = 0.0f, vx = 0.0f, vy = 0.0f;
  const bool big = active && n >= A.min_leaf_points && n >= 2;
  if (big) {
    if (w == 0) { const float t = kd_wide_chain<kChain, kCompact>(xin, n, lane, [](const float2& p, bool) { return p.x; }, stage); if (lane == 0) sh[0] = t; }
    if (w == 1) { const float t = kd_wide_chain<kChain, kCompact>(xin, n, lane, [](const float2& p, bool) { return p.y; }, stage); if (lane == 0) sh[1] = t; }
  }
  kd_barrier<kCompact>();
  if (big) {
    const float fn = (float) n;
    mx = sh[0] / fn; my = sh[1] / fn;
    if (w == 0) { const float t = kd_wide_chain<kChain, kCompact>(xin, n, lane, [&](const float2& p, bool in) { const float dx = p.x - mx; return in ? dx * dx : 0.0f; }, stage); if (lane == 0) sh[2] = t; }
    if (w == 1) { const float t = kd_wide_chain<kChain, kCompact>(xin, n, lane, [&](const float2& p, bool in) { const float dx = p.x - mx, dy = p.y - my; return in ? dx * dy : 0.0f; }, stage); if (lane == 0) sh[3] = t; }
    if (w == 2) { const float t = kd_wide_chain<kChain, kCompact>(xin, n, lane, [&](const float2& p, bool in) { const float dy = p.y - my; return in ? dy * dy : 0.0f; }, stage); if (lane == 0) sh[4] = t; }
  }
  kd_barrier<kCompact>();
  bool axis = false;
  float lo1 = 3.402823466e+38f, hi1 = -3.402823466e+38f, lo2 = lo1, hi2 = hi1;
  if (big) {
    const float sxx = sh[2], sxy = sh[3], syy = sh[4];
    const float tr = sxx + syy, df = sxx - syy;
    const float disc = __builtin_sqrtf(df * df + 4.0f * sxy * sxy);
    const float l1 = (tr + disc) / 2.0f;
    if (__builtin_fabsf(sxy) > 0.0f) { vx = l1 - syy; vy = sxy; } else if (sxx >= syy) { vx = 1.0f; vy = 0.0f; } else { vx = 0.0f; vy = 1.0f; }
    const float vn = __builtin_sqrtf(vx * vx + vy * vy);
    axis = vn > 0.0f;      // (the same value in every thread of the group)
    if (axis) {
      vx = vx / vn; vy = vy / vn;
      int nl_w = 0;
      for (int k0 = k_lo; k0 < k_hi; k0 += 64 * G) {
        float2 p[G];
#pragma unroll
        for (int j = 0; j < G; ++j) { const int k = k0 + 64 * j + lane; p[j] = k < k_hi ? kd_ld2<kCompact>(xin, k) : make_float2(0.0f, 0.0f); }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          bool left = false;
          if (k0 + 64 * j + lane < k_hi) {
            const float dx = p[j].x - mx, dy = p[j].y - my;
            const float a = dx * vx + dy * vy, b = -dx * vy + dy * vx;
            lo1 = a < lo1 ? a : lo1; hi1 = a > hi1 ? a : hi1; lo2 = b < lo2 ? b : lo2; hi2 = b > hi2 ? b : hi2;
            left = a < 0.0f;
          }
          nl_w += __popcll(__ballot(left));
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        lo1 = fminf(lo1, __shfl_xor(lo1, o, 64)); hi1 = fmaxf(hi1, __shfl_xor(hi1, o, 64));
        lo2 = fminf(lo2, __shfl_xor(lo2, o, 64)); hi2 = fmaxf(hi2, __shfl_xor(hi2, o, 64));
      }
      if (lane == 0) { sh[8 + 4 * w] = lo1; sh[9 + 4 * w] = hi1; sh[10 + 4 * w] = lo2; sh[11 + 4 * w] = hi2; shi[w] = nl_w; }
    }
  }
  kd_barrier<kCompact>();
  if (axis) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lo1 = fminf(lo1, sh[8 + 4 * u]); hi1 = fmaxf(hi1, sh[9 + 4 * u]); lo2 = fminf(lo2, sh[10 + 4 * u]); hi2 = fmaxf(hi2, sh[11 + 4 * u]);
      nl += shi[u];
    }
    const float e1 = (hi1 - lo1) / 2.0f, e2 = (hi2 - lo2) / 2.0f;
    split = (e1 > e2 ? e1 : e2) >= A.max_leaf_range && nl > 0 && nl < n;
  }
  int32_t* n_nodes = A.local_io ? A.n_nodes : A.n_nodes + c;
  if (active && !split) {      // a leaf: its points, in their order, go to their final place
    if (tid == 0) { KdNode nd = {0.0f, 0.0f, 0.0f, 0.0f, -1 - begin, end, 0, 0}; A.nodes[nbase + node] = nd; }
    float2* lxy = A.leaf_xy + base + begin; int32_t* lix = A.leaf_idx + base + begin;
    for (int k = tid; k < n; k += 256) { lxy[k] = kd_ld2<kCompact>(xin, k); lix[k] = iin ? kd_ldi<kCompact>(iin, k) : begin + k; }
  }
  if (split && tid == 0) shi[4] = kCompact ? (int) __hip_atomic_fetch_add((__attribute__((address_space(3))) int32_t*) A.n_nodes, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : atomicAdd(n_nodes, 2);
  kd_barrier<kCompact>();
  if (!split) return;
  const int left_id = __builtin_amdgcn_readfirstlane(shi[4]);
  const int nr = n - nl;
  const bool leaf_l = !(nl >= A.min_leaf_points && nl >= 2), leaf_r = !(nr >= A.min_leaf_points && nr >= 2);
  if (tid == 0) {
    { KdNode nd = {mx, my, vx, vy, left_id, 0, 0, 0}; A.nodes[nbase + node] = nd; }
    if (leaf_l) { KdNode nd = {0.0f, 0.0f, 0.0f, 0.0f, -1 - begin, begin + nl, 0, 0}; A.nodes[nbase + left_id] = nd; }
    if (leaf_r) { KdNode nd = {0.0f, 0.0f, 0.0f, 0.0f, -1 - (begin + nl), end, 0, 0}; A.nodes[nbase + left_id + 1] = nd; }
    const int n_next = (leaf_l ? 0 : 1) + (leaf_r ? 0 : 1);
    if (n_next) {
      const int4 il = make_int4(c, left_id, begin, begin + nl), ir = make_int4(c, left_id + 1, begin + nl, end);
      push(n_next, leaf_l ? ir : il, ir);
    }
  }
  float2* oxy_l = leaf_l ? A.leaf_xy + base + begin : A.xy_out + iob + begin;  int32_t* oix_l = leaf_l ? A.leaf_idx + base + begin : A.idx_out + iob + begin;
  float2* oxy_r = leaf_r ? A.leaf_xy + base + begin + nl : A.xy_out + iob + begin + nl;  int32_t* oix_r = leaf_r ? A.leaf_idx + base + begin + nl : A.idx_out + iob + begin + nl;
  // this wave's stretch of the stable partition: the lefts before it are the earlier stretches' counts, the rights the rest of the points before it
  int cl = 0;
  for (int u = 0; u < w; ++u) cl += shi[u];
  int cr = (k_lo < n ? k_lo : n) - cl;
  for (int k0 = k_lo; k0 < k_hi; k0 += 64 * G) {
    float2 p[G]; int src[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = k0 + 64 * j + lane; const bool valid = k < k_hi;
      p[j] = valid ? kd_ld2<kCompact>(xin, k) : make_float2(0.0f, 0.0f);
      src[j] = valid ? (iin ? kd_ldi<kCompact>(iin, k) : begin + k) : 0;
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const bool valid = k0 + 64 * j + lane < k_hi;
      bool left = false;
      if (valid) { const float dx = p[j].x - mx, dy = p[j].y - my; left = dx * vx + dy * vy < 0.0f; }
      const u64 bl = __ballot(valid && left), br = __ballot(valid && !left);
      if (valid) {
        if (left) { const int d = cl + __popcll(bl & lt_mask); if (leaf_l) { oxy_l[d] = p[j]; oix_l[d] = src[j]; } else { kd_st2<kCompact>(oxy_l, d, p[j]); kd_sti<kCompact>(oix_l, d, src[j]); } }
        else { const int d = cr + __popcll(br & lt_mask); if (leaf_r) { oxy_r[d] = p[j]; oix_r[d] = src[j]; } else { kd_st2<kCompact>(oxy_r, d, p[j]); kd_sti<kCompact>(oix_r, d, src[j]); } }
      }
      cl += __popcll(bl); cr += __popcll(br);
    }
  }
}
template <int kChain>
__global__ __launch_bounds__(256) void k_kd_level_wide(const KdBuildArgs A) {      // one WORKGROUP per node
  __shared__ float sh[32]; __shared__ int shi[8]; __shared__ __align__(16) float s_stage[4 * 256];
  const int n_items = A.n_items_ptr ? __builtin_amdgcn_readfirstlane(*A.n_items_ptr) : A.n_items;
  if ((int) blockIdx.x >= n_items) return;
  kd_node_wide<kChain>(A, A.q_in[blockIdx.x], threadIdx.x, true, [&](int n_next, const int4& first, const int4& second) {
    int q = atomicAdd(A.q_out_count, n_next);
    A.q_out[q] = first;
    if (n_next == 2) A.q_out[q + 1] = second;
  }, sh, shi, s_stage);
}

template <int kChain>
__global__ __launch_bounds__(256) void k_kd_level(const KdBuildArgs A) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  // (round 4: the level's item count is read where the previous level left it -- the host launches a level with an upper bound of its own and learns the
  // counts once, after the last one: a host round trip per level was 25 us x 15 levels of a 100k-point map's build)
  const int n_items = A.n_items_ptr ? __builtin_amdgcn_readfirstlane(*A.n_items_ptr) : A.n_items;
  if (item >= n_items) return;                       // whole waves leave: no workgroup barrier below
  kd_node<kChain>(A, A.q_in[item], lane, [&](int n_next, const int4& first, const int4& second) {
    int q = atomicAdd(A.q_out_count, n_next);
    A.q_out[q] = first;
    if (n_next == 2) A.q_out[q + 1] = second;
  });
}

// Round 4: the WHOLE build of a scan-sized cloud in ONE launch -- a workgroup per cloud walks its tree's levels itself (two barriers per level), its waves
// take the nodes of a level in turn, the queue of the next level sits in the cloud's own stretch of the queue buffers, and the leaf-order normals are written
// at the end.  CorrespondenceFinderKDTree2D::reset() runs whenever the fixed cloud changes (correspondence_finder_kd_tree_2d.cpp:6-8,31-38): for the live
// tracker that is once per scan, and the level-by-level build of round 3 paid a host round trip per level (~7 for a 1081-point scan).  Same kd_node, same
// order of every sequential sum, hence the same tree bit for bit (tests).  Clouds above max_points are left to the level loop (k_kd_level).
struct KdBuildWgArgs {
  KdBuildArgs B;                       // xy_in / idx_in / xy_out / idx_out / q_in / q_out are set per level by the kernel itself
  const int32_t* count; const float2* xy0; const float2* nrm0;
  float2* xy_buf[2]; int32_t* idx_buf[2]; int4* q_buf[2];
  float2* leaf_nrm; KdMeta* meta_rw;
  int32_t max_points;
};
template <int kChain>
__global__ __launch_bounds__(256) void k_kd_build_wg(const KdBuildWgArgs W) {
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = W.count[c], base = W.B.start[c];
  if (n > W.max_points) return;
  __shared__ int s_cnt[2], s_levels;
  const int qbase = (base >> 1) + c;                 // this cloud's stretch of the queue buffers: a level never holds more than n / 2 (+ 1) nodes
  if (tid == 0) { W.q_buf[0][qbase] = make_int4(c, 0, 0, n); s_cnt[0] = 1; s_cnt[1] = 0; s_levels = 0; W.B.n_nodes[c] = 1; }
  __syncthreads();
  KdBuildArgs A = W.B;
  for (int level = 0;; ++level) {
    const int cur = level & 1, nxt = cur ^ 1;
    const int items = s_cnt[cur];
    if (items == 0) break;                           // (workgroup-uniform)
    A.xy_in = level == 0 ? W.xy0 : W.xy_buf[cur]; A.idx_in = level == 0 ? nullptr : W.idx_buf[cur];
    A.xy_out = W.xy_buf[nxt]; A.idx_out = W.idx_buf[nxt];
    const int4* qin = W.q_buf[cur] + qbase; int4* qout = W.q_buf[nxt] + qbase;
    for (int item = wave; item < items; item += 4)
      kd_node<kChain>(A, qin[item], lane, [&](int n_next, const int4& first, const int4& second) {
        const int q = atomicAdd(&s_cnt[nxt], n_next);
        qout[q] = first;
        if (n_next == 2) qout[q + 1] = second;
      });
    __syncthreads();                                 // the level's writes (children's points, queue, counter) are complete and visible to the workgroup
    if (tid == 0) { s_cnt[cur] = 0; s_levels = level + 1; }
    __syncthreads();
  }
  // the normals in leaf order (k_kd_permute_normals), and the tree's size
  for (int i = tid; i < n; i += 256) W.leaf_nrm[base + i] = W.nrm0[base + W.B.leaf_idx[base + i]];
  if (tid == 0) { W.meta_rw[c].n_nodes = W.B.n_nodes[c]; W.meta_rw[c].pad0 = s_levels; }
}

// Round 4, the LATENCY form of the single-launch build: ONE scan (or a handful), the live tracker's reset() per scan.  k_kd_build_wg above walks the levels with
// its points, queue and node counter in global memory -- per level and node half a dozen dependent trips to the L2 and one returning atomic, 130 .. 170 us for a
// 1081-point scan on a chip that is otherwise idle.  Here the cloud's working set lives in LDS for the whole build (two copies of points and indices, both queues,
// the node counter; kd_node / kd_node_wide reach them through flat addresses: KdBuildArgs::local_io), the workgroup has sixteen waves, and the levels with at
// most four nodes run them as GROUPS of four waves (kd_node_wide: a chain per wave), the others a wave per node.  Leaves and node records go straight to
// their final places in global memory (stores nobody waits for: the barriers order LDS traffic only).  Same sums, same order, same tree.
// Measured (clock stamps inside the launch, 1081-point scan, 8 levels, 153 nodes): 82 us = 197 k cycles at 2.38 GHz against 130 .. 170 us; the first version
// of this kernel, with the chains unrolled as in the throughput kernels (42 KB of code, 400 bytes of scratch under the 128 registers of sixteen waves), took
// 184 us -- a kernel that runs once on an idle chip pays for every instruction it FETCHES.  What is left is the algorithm's own chain of dependent
// instructions: a level lasts as long as its LARGEST node (the splits of a scan are far from even), and a node is ~7 k cycles of one wave's dependent work
// beside its sums (two IEEE square roots, seven divisions, the extents' shuffles, ballots and ranks) -- 12 k cycles for a level of four 25-point nodes.
struct KdBuildScanArgs {
  KdBuildArgs B;
  const int32_t* count; const float2* xy0; const float2* nrm0;
  float2* leaf_nrm; KdMeta* meta_rw;
  int32_t cap;                         // points the LDS layout is sized for (the host launches this kernel only for clouds that fit)
  int32_t n_clouds;                    // clouds of THIS set (a launch over several sets is as wide as the largest)
  int32_t node_base[8];                // first node of every cloud's region (the kernel writes the set's KdMeta itself: nothing is uploaded ahead of it)
};
struct KdBuildScanMulti { KdBuildScanArgs w[kMaxSlices]; };      // the fixed sets of an aligner call's KD-tree slices, built side by side by ONE launch (grid.y = set)
static constexpr int kKdScanThreads = 1024, kKdScanGroups = kKdScanThreads / 256;
LSM2D_HD size_t kd_scan_lds_bytes(int cap) {      // (cap <= 32767: a queue entry packs begin and end into 16 bits each)
  const size_t q = (size_t) (cap / 2 + 2);
  return 2 * sizeof(float2) * (size_t) cap + 2 * sizeof(int32_t) * (size_t) cap + 2 * sizeof(int2) * q
       + kKdScanGroups * (32 * sizeof(float) + 8 * sizeof(int32_t)) + (kKdScanThreads / 64) * 256 * sizeof(float) + 64;
}
template <int kChain>
LSM2D_DEV void kd_build_scan_body(const KdBuildScanArgs& W, const int c) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, group = tid >> 8, gtid = tid & 255;
  if (c >= W.n_clouds) return;         // (a launch over several sets: this one has fewer clouds)
  const int n = W.count[c], base = W.B.start[c], cap = W.cap;
  if (n > cap) return;                 // (never: the host checked every cloud of the launch)
  const int qcap = cap / 2 + 2;
  unsigned char* at = smem;
  float* stage_all = reinterpret_cast<float*>(at); at += (kKdScanThreads / 64) * 256 * sizeof(float);      // (16-byte rows first)
  int2* qb[2]; qb[0] = reinterpret_cast<int2*>(at); at += sizeof(int2) * (size_t) qcap; qb[1] = reinterpret_cast<int2*>(at); at += sizeof(int2) * (size_t) qcap;      // (node, begin | end << 16)
  float2* xyb[2]; xyb[0] = reinterpret_cast<float2*>(at); at += sizeof(float2) * (size_t) cap; xyb[1] = reinterpret_cast<float2*>(at); at += sizeof(float2) * (size_t) cap;
  int32_t* ixb[2]; ixb[0] = reinterpret_cast<int32_t*>(at); at += sizeof(int32_t) * (size_t) cap; ixb[1] = reinterpret_cast<int32_t*>(at); at += sizeof(int32_t) * (size_t) cap;
  float* sh_all = reinterpret_cast<float*>(at); at += kKdScanGroups * 32 * sizeof(float);
  int32_t* shi_all = reinterpret_cast<int32_t*>(at); at += kKdScanGroups * 8 * sizeof(int32_t);
  int32_t* s_ctl = reinterpret_cast<int32_t*>(at);      // [0], [1]: items of the two queues; [2]: nodes handed out; [3]: levels walked
  for (int i = tid; i < n; i += kKdScanThreads) xyb[0][i] = W.xy0[base + i];
  if (tid == 0) { qb[0][0] = make_int2(0, n << 16); s_ctl[0] = 1; s_ctl[1] = 0; s_ctl[2] = 1; s_ctl[3] = 0; }
  __syncthreads();
  KdBuildArgs A = W.B;
  A.local_io = 1; A.n_nodes = s_ctl + 2; A.io_base = base; A.io_node_base = W.node_base[c];
  for (int level = 0;; ++level) {
    const int cur = level & 1, nxt = cur ^ 1;
    const int items = s_ctl[cur];
    if (items == 0) break;                           // (workgroup-uniform)
    A.xy_in = xyb[cur]; A.idx_in = level == 0 ? nullptr : ixb[cur];
    A.xy_out = xyb[nxt]; A.idx_out = ixb[nxt];
    const int2* qin = qb[cur]; int2* qout = qb[nxt];
    // (the queues and the counters are LDS: said explicitly -- through the pointer arrays above they were flat accesses, see kd_ld2)
    typedef int kd_v2i __attribute__((ext_vector_type(2)));
    const __attribute__((address_space(3))) kd_v2i* qin3 = (const __attribute__((address_space(3))) kd_v2i*) qin;
    __attribute__((address_space(3))) kd_v2i* qout3 = (__attribute__((address_space(3))) kd_v2i*) qout;
    __attribute__((address_space(3))) int32_t* ctl3 = (__attribute__((address_space(3))) int32_t*) s_ctl;
    auto item_of = [&](int i) { const kd_v2i e = qin3[i]; return make_int4(c, e.x, e.y & 0xFFFF, (int) ((unsigned) e.y >> 16)); };
    auto push = [&](int n_next, const int4& first, const int4& second) {
      const int q = (int) __hip_atomic_fetch_add(ctl3 + nxt, n_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      kd_v2i e0; e0.x = first.y; e0.y = first.z | (first.w << 16); qout3[q] = e0;
      if (n_next == 2) { kd_v2i e1; e1.x = second.y; e1.y = second.z | (second.w << 16); qout3[q + 1] = e1; }
    };
    // few, large nodes: a group of four waves each, ONE round (every group passes the same barriers, node or not).  (Eight nodes in two rounds lose to a
    // wave per node: a node costs ~10 k cycles of dependent instructions beside its chains -- two IEEE square roots, three divisions, the extents' shuffles,
    // the partition -- whoever runs it; clock stamps, level 3 of a 1081-point scan.)
    if (items <= kKdScanGroups) {
      for (int r = 0; r < items; r += kKdScanGroups) {
        const bool act = r + group < items;
        const int4 it = act ? item_of(r + group) : make_int4(c, 0, 0, 0);
        kd_node_wide<kChain, true>(A, it, gtid, act, push, sh_all + 32 * group, shi_all + 8 * group, stage_all + 4 * 256 * group);
      }
    } else {
      for (int item = wave; item < items; item += kKdScanThreads / 64) kd_node<kChain, true>(A, item_of(item), lane, push);
    }
    kd_barrier<true>();                              // the level's LDS writes (children's points, queue, counter) are complete and visible to the workgroup
    if (tid == 0) { s_ctl[cur] = 0; s_ctl[3] = level + 1; }
    kd_barrier<true>();
  }
  __syncthreads();                                   // ... and the leaf arrays in global memory, for the pass below
  // the normals in leaf order (k_kd_permute_normals), and the tree's size
  for (int i = tid; i < n; i += kKdScanThreads) W.leaf_nrm[base + i] = W.nrm0[base + W.B.leaf_idx[base + i]];
  if (tid == 0) { KdMeta km; km.node_base = W.node_base[c]; km.n_nodes = s_ctl[2]; km.pad0 = s_ctl[3]; km.pad1 = 0; W.meta_rw[c] = km; }
}
template <int kChain>
__global__ __launch_bounds__(kKdScanThreads) void k_kd_build_scan(const KdBuildScanArgs W) { kd_build_scan_body<kChain>(W, (int) blockIdx.x); }
template <int kChain>
__global__ __launch_bounds__(kKdScanThreads) void k_kd_build_scan_multi(const KdBuildScanMulti M) { kd_build_scan_body<kChain>(M.w[blockIdx.y], (int) blockIdx.x); }

// roots of every cloud's tree: work item (c, 0, 0, count[c]); one node handed out per cloud
__global__ void k_kd_init(const int32_t* __restrict__ count, int n_clouds, int4* __restrict__ q, int32_t* __restrict__ n_nodes) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_clouds) { q[c] = make_int4(c, 0, 0, count[c]); n_nodes[c] = 1; }
}
__global__ void k_kd_finish(const int32_t* __restrict__ n_nodes, int n_clouds, KdMeta* __restrict__ meta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_clouds) meta[c].n_nodes = n_nodes[c];
}

// the normals in leaf order, next to leaf_xy (every cloud of the set at once)
__global__ void k_kd_permute_normals(const float2* __restrict__ nrm, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                                     const int32_t* __restrict__ leaf_idx, float2* __restrict__ leaf_nrm, int cloud0) {
  const int c = cloud0 + blockIdx.y, n = count[c], base = start[c];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) leaf_nrm[base + i] = nrm[base + leaf_idx[base + i]];
}

struct SliceDev {
  CloudDev fixed, moving;
  int32_t finder;
  ProjK   proj;
  float   point_distance, normal_cos, max_distance;
  int32_t nn_group;          // host hint (largest fixed vs largest moving cloud of the sets): 1 = scan-sized fixed clouds, worth staging their tables in LDS
  int32_t cauchy;
  float   tau;
  int32_t min_corr;
  int32_t has_sensor;        // X_eff = S^-1 * X
  float   Sinv[3], cSinv, sSinv;
  int32_t fcan_offset;       // start of this slice's fixed canvas, in cells
  // single-alignment calls whose fixed set still sits in its pinned upload buffer (lsm2d_cloudset_upload defers the unpacking):
  // the kernel's prologue reads the host's AoS points over the bus itself, writes the set's arrays and count (later consumers
  // find them there) and goes on -- no separate k_upload_unpack launch in front of the alignment
  const float4* unpack_src;  // device view of the pinned AoS points, or nullptr
  int32_t unpack_n;
};

struct PriorDev { float z_inv[3], cz, sz, omega[9]; };   // Z^-1 and cos/sin of its angle, host-computed
struct StatsDev { int32_t n_corr, n_in, n_out; float chi_in, chi_out; uint32_t dig_lo, dig_hi; };

// lsm2d_pair_hash (include/lsm2d.h) on the device: the per-pair term of lsm2d_iteration_stats.pair_digest.  slice_salt = slice * 0x632BE5AB
// (wave-uniform).  Integer arithmetic only; tests hold the two definitions against each other through the oracle's digest.
LSM2D_DEV u64 pair_hash_dev(uint32_t slice_salt, uint32_t f, uint32_t m) {
  const uint32_t a = f * 0x9E3779B1u, b = (m ^ slice_salt) * 0x85EBCA77u;
  uint32_t lo = a ^ __builtin_rotateleft32(b, 13), hi = b ^ __builtin_rotateleft32(a, 19);
  lo += __builtin_rotateleft32(lo, 17) ^ b;
  hi += __builtin_rotateleft32(hi, 11) ^ a;
  return ((u64) hi << 32) | (u64) lo;
}
// one pair into the iteration's digest: a fire-and-forget 64-bit LDS add (order-independent: the sum wraps mod 2^64), no register held across the loops
LSM2D_DEV void digest_add(u64* s_dig, uint32_t slice_salt, int f, int m) { atomicAdd(reinterpret_cast<unsigned long long*>(s_dig), (unsigned long long) pair_hash_dev(slice_salt, (uint32_t) f, (uint32_t) m)); }

struct ResumeDev {      // what an alignment carries from one iteration to the next (thread 0's serial state in k_align)
  float pose[3]; float H[9]; float prev_chi;
  int32_t phase, phase_start, phase_end, last_n_in, status, done, it;
};
struct AlignArgs {
  int32_t n_align, n_slices, max_it, min_inliers;
  float   damping;
  float   term_eps;                         // lsm2d_aligner_params.termination_chi_epsilon (0 = run all iterations)
  int32_t inlier_runs;                      // lsm2d_aligner_params.enable_inlier_only_runs: a second loop of up to max_it iterations over inliers only (lsm2d.h)
  int32_t stats_stride;                     // iterations an alignment may run = row length of out_stats: max_it * (1 + inlier_runs), at least 1
  float*  out_last_pose;                    // [n][3] or nullptr: the pose the LAST started iteration began at (lsm2d_align_batch_pairs re-derives that iteration's pairs from it)
  int32_t cols_max, fcan_total;
  int32_t nn_lds_points, nn_lds_cells;      // > 0: single NN slice over scan-sized fixed clouds -- their search tables are staged in LDS (room for this many)
  int32_t nn_qcache;                        // > 0: single NN slice with its tables in global memory (kNNGlobal): room in LDS for this many queries' cached cell ranges (32 bytes each)
  int32_t kd_lds_nodes;                     // > 0: single KD-tree slice -- room in LDS for this many nodes of the fixed cloud's tree (its top levels)
  int32_t kd_lds_points;                    // > 0: ... and, for scan-sized fixed clouds, for this many leaf points (whole trees on chip)
  const int32_t* order;                     // alignment handled by workgroup b (nullptr: b itself) -- the balanced placement of k_balance_order
  float cull_est_mt, cull_est_mth;          // margins of k_cull_estimate's chunk test (metres, radians)
  // two launches for one batch (k_first_iteration, then k_align: see k_first_iteration): stage 0 the whole alignment in this launch; 1 the iterations before
  // stage_split, then the next iteration's unit lists for their LENGTH only, the state to `resume`, the length to `stage_work`; 2 the rest, from `resume`
  int32_t stage, stage_split;
  struct ResumeDev* resume;                 // [n]
  int32_t* stage_work;                      // [n] 0 .. 512: the units the alignment will stream per iteration (0: it finished in the first stage)
  float cull_mt2;                           // cull_mt squared (host)
  int32_t* wg_place;                        // [grid] or nullptr: every workgroup notes the CU it ran on (place_key) for the next call's placement
  int32_t cull_block;                       // steps per unit of the culled stream (0: automatic; tuning knob)
  int32_t cull;                             // 1: projective slices drop the chunks of the moving cloud that cannot yield a pair (chunk_may_matter), results unchanged
  // kProjCulled (round 4): every slice keeps a LIST of the (block, chunk) units that survive the test at block level, in dynamic LDS at units_off
  // (kCullBlocks * kAlignBlock 16-bit entries per slice), built with margins (cull_mt metres, cull_mth radians) and kept while the slice's transform
  // stays within them of the one it was built at (cull_keep 0: rebuilt every iteration, zero margins -- A/B knob)
  int32_t units_off, cull_keep;
  int32_t units_stride;                     // entries per slice of the unit lists: the largest block_stride of the batch's moving sets x kAlignBlock
  float   cull_mt, cull_mth;
  // Round 5, big maps (k_align<1,0,0,0,6>): the workgroups of one XCD walk the map IN STEP, pass by pass.  A 1M-point map's lane copy (8 MB) does not fit an
  // XCD's 4 MiB L2, and 125 workgroups streaming different parts of it at the same time missed on 44 % of their requests (41 GB of fabric reads per
  // 1000-alignment launch for 50 MB of data, the chip at 1.83 GHz under that load: profiles/r05/size_sweep_r05a.txt).  All workgroups of a one-round launch start
  // together and walk their unit lists in the same block-major order -- what pulls them apart is only that their lists differ in length, a quarter of a pass per
  // iteration.  So every workgroup counts itself into done[g] when ITS pass g = (iteration, slice) is over, and its thread 0 -- at the end of the serial solve,
  // while the other threads stand at the iteration's closing barrier anyway -- waits until every workgroup registered on ITS XCD has finished the pass that
  // lies xcd_window passes back (0: the one just finished) or has gone.  One atomic add and a handful of scalar looks per workgroup and iteration; the counters
  // of an XCD are touched by that XCD only (plain L2 atomics, no fabric traffic, no fence).  Nothing but the ORDER IN TIME of the z-buffer updates changes:
  // every result keeps its bits.  xcd_sync == nullptr: free-running (the host offers the lockstep only to launches of one dispatch round, whose workgroups
  // are all resident from the start).  What was tried before this form -- per wave and per BLOCK of the map -- and what it cost: DESIGN App. A.
  uint32_t* xcd_sync;                       // [16 XCC ids][xcd_stride]: word 0 workgroups registered, word 1 workgroups gone, words 2..6 the watchdog's notes, word 16 + g: workgroups that have finished pass g
  int32_t xcd_stride, xcd_window, xcd_positions;
  int32_t pq_cull_off;                      // > 0: byte offset in dynamic LDS of the point-query finders' culling state (occupancy bitmap of the fixed cloud, then
  int32_t pq_keep_words;                    //   pq_keep_words 64-bit words of per-tile keep bits); single-slice NN / KD-tree batches with scan-sized fixed clouds
  int32_t pair_mov_cap;                     // latency kernel: moving points per slice it may keep in LDS (kPairMovCap, or 0: no room)
  int32_t pair_fix_cap;                     // latency kernel: fixed points per slice it may keep in LDS (0: no room)
  const float* init_pose;
  const PriorDev* prior;
  int32_t  host_polls;                      // results go to pinned host memory and the host polls the status words: release them to the system
  int32_t  inline_n1;                       // 1: a single alignment whose start pose / prior travel in the kernel arguments (pose1, prior1)
  float    pose1[3];
  PriorDev prior1;
  float* out_pose; float* out_H; int32_t* out_status; int32_t* out_its; StatsDev* out_stats;
  // kernel timing on (lsm2d_set_option "kernel_timing"): thread 0 of every clock_stride-th workgroup stamps s_memtime (shader
  // cycles) and s_memrealtime (100 MHz) once at its start and once at its end -- [a / clock_stride][4] = {cycles, 10 ns ticks, start tick, hardware id} of
  // the workgroup's lifetime, from which the host reads the clock the chip held under THIS load (MI355X_MICROARCH.md, DVFS note 6)
  unsigned long long* clock_out; int32_t clock_stride;
  // "sum_order" 1 (the k_align_seq / k_split_finish<true> instantiations): byte offset in dynamic LDS of the trip's pair records, kAlignBlock x kSeqFields floats
  // (lsm2d_device.h, "sum_order"); sits in what was padding, so the other fields keep their offsets
  int32_t seq_off;
  SliceDev s[kMaxSlices];
};

LSM2D_DEV int pick_cloud(const CloudDev& c, int a) {
  return c.index ? c.index[a] : (c.n_clouds == 1 ? 0 : a);
}

// bin walk of one column: gates of correspondence_finder_projective_2d.cpp:61-69
LSM2D_DEV bool match_bin(u64 fk, u64 mk, const SliceDev& S, const Iso& T, const float2* fn, const float2* mn,
                         int& fi, int& mi, float2& nf, float2& nm) {
  if (mk == kEmptyCell || fk == kEmptyCell) return false;
  const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
  if (__builtin_fabsf(fd - md) > S.point_distance) return false;
  fi = (int) (uint32_t) fk; mi = (int) (uint32_t) mk;
  nf = fn[fi]; nm = mn[mi];
  float nqx, nqy;
  xf_normal(T, nm.x, nm.y, nqx, nqy);
  const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
  return !(dot < S.normal_cos);
}

LSM2D_DEV Iso slice_iso_of(int has_sensor, float cSinv, float sSinv, const float Sinv[3], const float pose[3]) {
  float Xe[3] = {pose[0], pose[1], pose[2]};
  if (has_sensor) compose(cSinv, sSinv, Sinv, pose, Xe);
  Iso T; sincos_fixed(Xe[2], T.s, T.c); T.tx = Xe[0]; T.ty = Xe[1];
  return T;
}
LSM2D_DEV Iso slice_iso(const SliceDev& S, const float pose[3]) {      // X_eff = S^-1 X (AlignerSliceProcessorLaser2DWithSensor) as rotation + translation
  return slice_iso_of(S.has_sensor, S.cSinv, S.sSinv, S.Sinv, pose);
}
// prologue of the single-alignment kernels: unpack the slices' freshly uploaded fixed sets (see SliceDev::unpack_src)
LSM2D_DEV void unpack_fixed_set(const SliceDev& S, int tid, int nthreads) {
  float2* xy = const_cast<float2*>(S.fixed.xy); float2* nrm = const_cast<float2*>(S.fixed.nrm);
  for (int i = tid; i < S.unpack_n; i += nthreads) {
    const float4 v = S.unpack_src[i];
    xy[i] = make_float2(v.x, v.y); nrm[i] = make_float2(v.z, v.w);
  }
  if (tid == 0) *const_cast<int32_t*>(S.fixed.count) = S.unpack_n;
}
#ifndef LSM2D_ALIGN_MIN_WAVES
#define LSM2D_ALIGN_MIN_WAVES 8      // waves per SIMD the register allocator must leave room for
#endif
#ifndef LSM2D_QUERY_MIN_WAVES
#define LSM2D_QUERY_MIN_WAVES 8      // the same for the instantiations without a projective slice (point-query finders); 6 and 4 measured 15-50 % slower
#endif
// SE2 odometry prior (AlignerSliceOdom2DPrior, MULTI.json:402-422): e = t2v(Z^-1 X), J = blkdiag(R_e, 1) for the right perturbation;
// adds J^T Omega J to H and J^T Omega e to b.  One definition for k_align and the split path: the same operation order in both.
// prior_terms: the nine and three values that go into H and b -- they do not depend on H or b, so whoever has the pose can have them ready
// (the latency kernel's thread 0 computes them while it would otherwise wait at the barrier for the slowest wave)
// (kAdd: the terms are added to H and b as they come -- the form k_align's out-of-line call takes: twelve registers fewer)
template <bool kAdd>
LSM2D_DEV void prior_apply(const PriorDev& Pz, const float pose[3], float Hp[9], float bp[3]) {
  float E[3]; compose(Pz.cz, Pz.sz, Pz.z_inv, pose, E);
  float c, s_; sincos_fixed(E[2], s_, c);
  const float Jp[9] = {c, -s_, 0.0f, s_, c, 0.0f, 0.0f, 0.0f, 1.0f};
  float OJ[9], Oe[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Oe[r] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) Oe[r] += Pz.omega[3 * r + k] * E[k];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      OJ[3 * r + cc] = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) OJ[3 * r + cc] += Pz.omega[3 * r + k] * Jp[3 * k + cc];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      float v = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) v += Jp[3 * k + r] * OJ[3 * k + cc];
      if (kAdd) Hp[3 * r + cc] += v; else Hp[3 * r + cc] = v;
    }
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) v += Jp[3 * k + r] * Oe[k];
    if (kAdd) bp[r] += v; else bp[r] = v;
  }
}
LSM2D_DEV void prior_terms(const PriorDev& Pz, const float pose[3], float Hp[9], float bp[3]) { prior_apply<false>(Pz, pose, Hp, bp); }
LSM2D_DEV void add_prior_inline(const PriorDev& Pz, const float pose[3], float H[9], float b[3]) { prior_apply<true>(Pz, pose, H, b); }
// k_align (64 VGPRs) and the split path call it: rarely taken, and out of the register allocation of their loops
__device__ __noinline__ void add_prior(const PriorDev& Pz, const float pose[3], float H[9], float b[3]) { add_prior_inline(Pz, pose, H, b); }

LSM2D_DEV int block_compact_pos(bool flag, int* s_tot, int parity, int& base, int tid, int nwaves);      // (defined with the mapping kernels below)

// kHasProj / kHasNN: which finders the batch's slices use -- the unused one is compiled out so the
// projective hot loop does not carry the NN path's register pressure (and vice versa).
// kHasDist: a slice uses the distance-map finder (compiled out otherwise so the NN search keeps its registers).
// kHasKd: a slice uses the KD-tree finder (LSM2D_FINDER_KDTREE); compiled out otherwise.
// kNNGlobal: a pure grid-NN batch whose search tables stay in global memory (the map is the fixed cloud: BASELINE's wording with the exact search) --
// an instantiation of its own, so that its position-keeping search (nn_query_pos) does not share 64 registers with the LDS-table path of the
// tracker's wiring (both forms in one kernel: scratch 16 -> 80 bytes, role A 7.2 -> 10.6 ms)
// kNNMode 2: the counterpart -- a pure grid-NN batch whose tables the host has PROVED to fit the LDS staging for every alignment (nn_lds_points is the
// largest fixed cloud, nn_lds_cells the grid ensure_grid() gives that size): the search in global memory and the cooperative loop are compiled out
// where a workgroup runs: the key the balanced placement groups workgroup ids by (see k_balance_order)
static constexpr int kPlaceKeys = 4096;      // XCC (4 bits) | SE (3) | SH (1) | CU (4)
LSM2D_DEV int place_key() {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);      // HW_REG_HW_ID, HW_REG_XCC_ID
  return (int) (((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u));
}
// kSeq: "sum_order" 1 -- H, b and the chi^2 statistics are added pair after pair in the reference's order (lsm2d_device.h: pair_terms / seq_walk) instead of
// in trees; instantiations of their own (k_align_seq), so the default kernels do not carry the records' code
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd = false, int kNNMode = 0, bool kFirstStage = false, bool kSeq = false>
LSM2D_DEV void align_body(const AlignArgs& A) {
  __builtin_assume(A.n_slices >= 1 && A.n_slices <= kMaxSlices);      // (the host refuses anything else: the slice loops need no guard -- which, as a flag, was kept in a vector register and spilled)
  constexpr bool kNNGlobal = kNNMode == 1, kNNLds = kNNMode == 2;
  // the same for a pure KD-tree batch: 3 = every alignment's whole tree, leaf arrays included, is in LDS (the tracker's wiring: a tree per scan); 4 = only the
  // top of the tree is (the map is the fixed cloud): the other form of the descent and of the leaf scan is compiled out
  constexpr bool kKdAllLds = kNNMode == 3, kKdTop = kNNMode == 4;
  // ... and for a pure projective batch: 5 = every slice streams a lane-chunked moving cloud through the exact culling in units (what a batch against a map
  // does): the plain lane stream, the row-major variant and the per-pair stream of small clouds are compiled out
  // 6 = 5 with the XCD window (AlignArgs::xcd_sync: big maps, one dispatch round): an instantiation of its own, so that the headline's loop does not carry the
  // window's state (in one body: 16 bytes of scratch in the kernel that had none)
  constexpr bool kProjCulled = kNNMode == 5 || kNNMode == 6, kXcdWindow = kNNMode == 6;
  static_assert(kNNMode == 0 || ((kNNMode <= 2) && kHasNN && !kHasProj && !kHasDist && !kHasKd) || ((kNNMode == 3 || kNNMode == 4) && kHasKd && !kHasProj && !kHasDist && !kHasNN) ||
                ((kNNMode == 5 || kNNMode == 6) && kHasProj && !kHasNN && !kHasDist && !kHasKd), "kNNMode: one finder only");
  extern __shared__ __align__(16) unsigned char smem[];
  // (round 4: the fixed winners' payload no longer sits in LDS -- 16 bytes per column, 17 KB at 1081 -- the bin walk gathers it like the moving winner's,
  // one 16-byte row of the cloud's AoS copy each, both in flight together; the room holds the culled stream's unit lists)
  u64* mcan = reinterpret_cast<u64*>(smem);
  u64* fcan = mcan + A.cols_max;
  float* red = reinterpret_cast<float*>(fcan + A.fcan_total + ((A.cols_max + A.fcan_total) & 1));     // [nwaves][kAccumWords], 16-byte aligned
  // NN finder over a scan-sized fixed cloud (the tracker wiring: tree over the scan, every map point a query): the cloud's search
  // tables live in LDS for the whole alignment -- 20 iterations x N_m queries then touch global memory only for the query stream
  float2* l_sxy = reinterpret_cast<float2*>(red + (kAlignBlock / 64) * kAccumWords);
  int* l_qc = reinterpret_cast<int*>(red + (kAlignBlock / 64) * kAccumWords);      // kNNGlobal: [nn_qcache][8] cached cell ranges per query (16-byte aligned: the host pads)
  uint16_t* l_cst = reinterpret_cast<uint16_t*>(l_sxy + A.nn_lds_points);
  uint16_t* l_sidx = l_cst + ((A.nn_lds_cells + 2) & ~1);
  // KD-tree finder: the top levels of the fixed cloud's tree (its first kd_lds_nodes nodes) live in LDS for the whole alignment -- every
  // descent starts there (the host offers this only to pure KD-tree batches, where the region behind `red` is 16-byte aligned and free)
  float4* l_kpl = reinterpret_cast<float4*>(red + (kAlignBlock / 64) * kAccumWords);
  int2* l_klk = reinterpret_cast<int2*>(l_kpl + A.kd_lds_nodes);
  float2* l_kxy = reinterpret_cast<float2*>(l_klk + A.kd_lds_nodes + (A.kd_lds_nodes & 1));      // 16-byte aligned: pairs of points are read as one
  float2* l_knr = l_kxy + A.kd_lds_points + (A.kd_lds_points & 1);
  __shared__ float s_pose[3];
  __shared__ Iso   s_iso[kMaxSlices];
  // s_H: information matrix (H of the last solved iteration, built and solved IN LDS: thread 0's serial code has 64 registers like
  // everybody else, and what it kept in private arrays went to scratch -- eleven dependent round trips to memory per iteration);
  // s_sum: this iteration's sums in the order of Accum (h00 h01 h02 h11 h12 h22 b0 b1 b2 chi_in chi_out | n_in n_out as integers),
  // each added by the lane of wave 0 that gathered it
  __shared__ float s_H[9], s_rhs[3], s_sum[kAccumWords + 2];
  __shared__ int   s_n_corr, s_active, s_done, s_status, s_last_n_in;
  __shared__ float s_prev_chi;      // total chi^2 of the previous iteration (termination_chi_epsilon)
  __shared__ u64 s_dig;             // this iteration's pair digest (lsm2d_iteration_stats.pair_digest): every matched pair adds its hash; only when statistics go out
  __shared__ int s_it0;            // the iteration this launch starts at (0, or where the first of two launches stopped)
  __shared__ int s_phase, s_phase_start, s_phase_end;      // 0: the regular loop, 1: the inlier-only runs (enable_inlier_only_runs); iterations [start, end) belong to the phase
  __shared__ uint16_t s_surv[kAlignBlock];      // culling: the chunks of the moving cloud that survived this iteration's test, compacted in thread order
  __shared__ int s_wcnt[2 * (kAlignBlock / 64)];
  __shared__ int s_nunits[kMaxSlices], s_rebuild[kMaxSlices];      // kProjCulled: entries in a slice's unit list; the list must be rebuilt before it is streamed again
  __shared__ Iso s_list_iso[kMaxSlices];                             // ... and the transform it was built at
  uint16_t* l_units = reinterpret_cast<uint16_t*>(smem + A.units_off);      // [n_slices][kCullBlocks * kAlignBlock]
  float* l_rec = reinterpret_cast<float*>(smem + (kSeq ? A.seq_off : 0));   // kSeq: [kSeqHalf][kSeqFields]: half a trip's pair records
  __shared__ PriorDev s_prior;      // read once: with zero-copy arguments A.prior is host memory, a PCIe round trip per access

  // (the alignment's index is wave-uniform: said so, or everything indexed by it would live in vector registers)
  const int a = A.order ? __builtin_amdgcn_readfirstlane(A.order[blockIdx.x]) : (int) blockIdx.x, tid = threadIdx.x;
  constexpr int nwaves = kAlignBlock / 64;
  constexpr int kPriorWords = (int) (sizeof(PriorDev) / sizeof(float));
  __shared__ unsigned long long s_clk[2];      // start stamps wait in LDS: no register is held across the kernel for them
#ifdef LSM2D_PHASE_PROBE      // diagnostics build: thread 0 sums the cycles it spends in the query / projection phase, at the barrier + reduction, and in the solve
  __shared__ unsigned long long s_ph[4];
  if (tid == 0) { s_ph[0] = s_ph[1] = s_ph[2] = 0; s_ph[3] = __builtin_amdgcn_s_memtime(); }
#define LSM2D_PH(k) do { if (tid == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); s_ph[k] += now__ - s_ph[3]; s_ph[3] = now__; } } while (0)
#else
#define LSM2D_PH(k) do { } while (0)
#endif
  if (A.wg_place && tid == 0) A.wg_place[blockIdx.x] = place_key();
  // the XCD lockstep (AlignArgs::xcd_sync): this workgroup's counters are its XCD's; it counts itself in before anything else
  uint32_t* xsync = nullptr;      // (wave-uniform: SGPRs)
  if (kXcdWindow && !kFirstStage && A.xcd_sync) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 15u;      // HW_REG_XCC_ID
    xsync = A.xcd_sync + (size_t) xcc * A.xcd_stride;
    if (tid == 0) __hip_atomic_fetch_add(&xsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const bool stamp = A.clock_out && tid == 0 && a % A.clock_stride == 0;
  if (stamp) { s_clk[0] = __builtin_amdgcn_s_memtime(); s_clk[1] = __builtin_amdgcn_s_memrealtime(); }
  if (A.prior && tid >= 64 && tid < 64 + kPriorWords)
    ((float*) &s_prior)[tid - 64] = A.inline_n1 ? ((const float*) &A.prior1)[tid - 64] : ((const float*) (A.prior + a))[tid - 64];

  // ---- point-query finders, tracker wiring (a scan-sized fixed cloud, every point of a big moving cloud a query): EXACT culling of the queries.
  // A pair needs a fixed point within max_distance of the transformed moving point (the grid search's gate d2 <= md2, the tree's d2 < md2, the
  // distance map's parent pixel within max_distance of the query's pixel), so a TILE of 64 consecutive moving points -- what one wave handles in
  // one trip of the query loop -- can be skipped when no fixed point lies within rho + max_distance of its bounding circle's centre
  // (k_tile_bounds: centre, rho).  The fixed cloud is rasterised ONCE per alignment into a 128 x 128 occupancy bitmap (cell side g: its extent
  // / 125, at least a sixth of the reach); the test looks at the (2k + 1)^2 cells around the centre's, k = floor(reach / g) + 1: if they are
  // clear, every fixed point is more than `reach` away.  The queries keep their threads and a skipped query could not have paired: every sum
  // keeps its bits.  (Measured on configs[1], role A: 49 % of the tiles survive where an exact distance test would keep 32 %.)
  constexpr int kPqRowWords = 5, kPqOccWords = 128 * kPqRowWords;      // rows of 128 bits and one word that stays zero: a row's window is read as two words
  __shared__ unsigned s_pqbb[4];      // the fixed cloud's bounding box as order-preserving integers: min x, min y, max x, max y
  __shared__ float s_pq[4];           // bitmap origin x, y, 1 / g, the reach beyond a tile's own radius (< 0: culling off for this alignment)
  const bool pq_on = (kHasNN || kHasKd || kHasDist) && A.pq_cull_off > 0;
  uint32_t* l_occ = reinterpret_cast<uint32_t*>(smem + (pq_on ? A.pq_cull_off : 0));
  u64* l_keep = reinterpret_cast<u64*>(l_occ + kPqOccWords);
  auto ordered = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
  auto unordered = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); };
  if (pq_on) {
    for (int i = tid; i < kPqOccWords; i += kAlignBlock) l_occ[i] = 0u;
    if (tid < 4) s_pqbb[tid] = tid < 2 ? 0xFFFFFFFFu : 0u;
  }

  if (kHasProj && !kHasNN && !kHasDist && !kHasKd && A.inline_n1)
    for (int s = 0; s < A.n_slices; ++s) if (A.s[s].unpack_src) unpack_fixed_set(A.s[s], tid, kAlignBlock);      // visible after the barrier below
  // ---- prologue: fixed canvases, camera at identity (correspondence_finder_projective_2d.cpp:37-44)
  for (int i = tid; i < A.fcan_total; i += kAlignBlock) fcan[i] = kEmptyCell;
  for (int i = tid; i < A.cols_max; i += kAlignBlock) mcan[i] = kEmptyCell;      // afterwards the bin walk resets what it reads
  // per-iteration set-up by lane 0: X_eff = S^-1 X per slice (AlignerSliceProcessorLaser2DWithSensor), cos/sin once per
  // slice, zeroed sums.  Done here for iteration 0 and at the end of every solve for the next one (no extra barrier).
  auto begin_iteration = [&]() {
    for (int s = 0; s < A.n_slices; ++s) {
      float Xe[3] = {s_pose[0], s_pose[1], s_pose[2]};
      if (A.s[s].has_sensor) compose(A.s[s].cSinv, A.s[s].sSinv, A.s[s].Sinv, s_pose, Xe);
      sincos_fixed(Xe[2], s_iso[s].s, s_iso[s].c); s_iso[s].tx = Xe[0]; s_iso[s].ty = Xe[1];
    }
    // (the zero is made HERE, every time: a constant the compiler may hoist becomes a zero quad that every thread keeps -- and spills --
    // across the whole kernel for thread 0's sake)
    float zf = 0.0f; int zi = 0;
    asm volatile("" : "+v"(zf), "+v"(zi));
    for (int k = 0; k < 11; ++k) s_sum[k] = zf;
    s_sum[11] = s_sum[12] = __int_as_float(zi);
    s_n_corr = s_active = zi;
    s_dig = (u64) (unsigned) zi;
    if (kProjCulled) for (int s = 0; s < A.n_slices; ++s) {
      // has the slice's transform left the neighbourhood its unit list serves?  Seen from the sensor the change is a rotation by dth about the origin
      // and a translation d = t - R(dth) t0 (chunk_may_matter): the list holds while |d| <= cull_mt and |dth| <= cull_mth
      const Iso N = s_iso[s], L = s_list_iso[s];
      const float cd = N.c * L.c + N.s * L.s, sd = N.s * L.c - N.c * L.s;
      const float dx = N.tx - (cd * L.tx - sd * L.ty), dy = N.ty - (sd * L.tx + cd * L.ty);
      s_rebuild[s] = (A.cull_keep && cd > 0.5f && __builtin_fabsf(sd) <= A.cull_mth && dx * dx + dy * dy <= A.cull_mt2) ? zi : 1;      // (the square comes with the arguments: formed here it was a
      // loop invariant in a vector register, kept -- and spilled -- across the whole iteration for thread 0's sake)
    }
    if (A.out_last_pose) { A.out_last_pose[3 * a + 0] = s_pose[0]; A.out_last_pose[3 * a + 1] = s_pose[1]; A.out_last_pose[3 * a + 2] = s_pose[2]; }
  };
  const bool resumed = kExperiments && kProjCulled && !kFirstStage && A.stage == 2;      // the second of two launches: the alignment goes on where k_first_iteration left it
  if (tid == 0) {
    if (resumed) {
      const ResumeDev R = A.resume[a];
      s_pose[0] = R.pose[0]; s_pose[1] = R.pose[1]; s_pose[2] = R.pose[2];
      s_done = R.done; s_status = R.status; s_last_n_in = R.last_n_in; s_prev_chi = R.prev_chi;
      s_phase = R.phase; s_phase_start = R.phase_start; s_phase_end = R.phase_end; s_it0 = R.it;
      for (int k = 0; k < 9; ++k) s_H[k] = R.H[k];
    } else {
      if (A.inline_n1) { s_pose[0] = A.pose1[0]; s_pose[1] = A.pose1[1]; s_pose[2] = A.pose1[2]; }
      else { s_pose[0] = A.init_pose[3 * a + 0]; s_pose[1] = A.init_pose[3 * a + 1]; s_pose[2] = A.init_pose[3 * a + 2]; }
      s_done = 0; s_status = LSM2D_RUNNING; s_last_n_in = 0;
      s_phase = 0; s_phase_start = 0; s_phase_end = A.max_it; s_it0 = 0;
      for (int k = 0; k < 9; ++k) s_H[k] = 0.0f;
    }
    for (int s = 0; s < kMaxSlices; ++s) { s_list_iso[s].c = 1.0f; s_list_iso[s].s = 0.0f; s_list_iso[s].tx = 0.0f; s_list_iso[s].ty = 0.0f; }
    if (!s_done) begin_iteration();      // (resumed: the transforms and the zeroed sums the first launch's last begin_iteration() made, made again from the same pose)
    for (int s = 0; s < kMaxSlices; ++s) s_rebuild[s] = 1;      // no list yet
  }
  __syncthreads();
  if (resumed && s_done) return;         // it finished in the first launch: its results are out
  if (kNNGlobal) for (int i = tid; i < A.nn_qcache; i += kAlignBlock) l_qc[8 * i] = 0x7fffffff;      // no cell cached yet (visible after the barriers below)
  bool nn_lds = false;
  if (kHasNN && !kNNGlobal && A.nn_lds_points > 0) {
    const SliceDev& S = A.s[0];
    const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc];
    const GridMeta g0 = S.fixed.grid.meta[fc];
    const int ncell = g0.gw * g0.gh;
    nn_lds = nf <= A.nn_lds_points && ncell + 1 <= A.nn_lds_cells;     // workgroup-uniform; else this alignment searches in global memory
    if (nn_lds) {
      const int32_t* cst = S.fixed.grid.cell_start + g0.cell_base;
      const int fbase = S.fixed.start[fc];
      for (int i = tid; i <= ncell; i += kAlignBlock) l_cst[i] = (uint16_t) cst[i];
      for (int i = tid; i < nf; i += kAlignBlock) { l_sxy[i] = S.fixed.grid.sorted_xy[fbase + i]; l_sidx[i] = (uint16_t) S.fixed.grid.sorted_idx[fbase + i]; }
    }
  }
  if (kNNLds && !nn_lds) {      // cannot happen (the host sized the staging for the set's largest cloud): refuse loudly rather than search tables that are not there
    if (tid == 0) {
      A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
      if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = 0.0f;      // (no iteration ran: the information matrix is the zero the regular path would hand back)
      if (A.out_its) A.out_its[a] = 0;
      if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], (int) LSM2D_CAPACITY_EXCEEDED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      else A.out_status[a] = LSM2D_CAPACITY_EXCEEDED;
    }
    return;
  }
  int kd_lds = 0;                            // nodes of this alignment's tree that were staged (workgroup-uniform)
  bool kd_leaves_lds = false;                // ... and its leaf arrays
  if (kHasKd && A.kd_lds_nodes > 0) {
    const SliceDev& S = A.s[0];
    const KdMeta km = S.fixed.kd.meta[pick_cloud(S.fixed, a)];
    kd_lds = km.n_nodes < A.kd_lds_nodes ? km.n_nodes : A.kd_lds_nodes;
    const KdNode* nd = S.fixed.kd.nodes + km.node_base;
    for (int i = tid; i < kd_lds; i += kAlignBlock) {
      l_kpl[i] = reinterpret_cast<const float4*>(nd)[2 * i];
      const int4 w = reinterpret_cast<const int4*>(nd)[2 * i + 1]; l_klk[i] = make_int2(w.x, w.y);
    }
    // a scan-sized fixed cloud (the tracker wiring: a tree per scan, every map point a query): its leaf arrays ride in LDS too -- the whole
    // tree is on chip and 20 iterations x N_m queries touch global memory for the query stream only
    if (A.kd_lds_points > 0) {
      const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc], fb = S.fixed.start[fc];
      kd_leaves_lds = !kKdTop && nf <= A.kd_lds_points && kd_lds == km.n_nodes;      // workgroup-uniform
      if (kd_leaves_lds) for (int i = tid; i < nf; i += kAlignBlock) { l_kxy[i] = S.fixed.kd.leaf_xy[fb + i]; l_knr[i] = S.fixed.kd.leaf_nrm[fb + i]; }
    }
  }
  if (pq_on) {      // bounding box of the fixed cloud (finite points only)
    const SliceDev& S = A.s[0];
    const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc];
    const float2* fp = S.fixed.xy + S.fixed.start[fc];
    for (int i = tid; i < nf; i += kAlignBlock) {
      const float2 p = fp[i];
      if (__builtin_fabsf(p.x) < 1e30f && __builtin_fabsf(p.y) < 1e30f) {
        atomicMin(&s_pqbb[0], ordered(p.x)); atomicMin(&s_pqbb[1], ordered(p.y)); atomicMax(&s_pqbb[2], ordered(p.x)); atomicMax(&s_pqbb[3], ordered(p.y));
      }
    }
  }
  if (kKdAllLds && !kd_leaves_lds) {      // cannot happen (the host sized the staging for the set's largest tree): refuse loudly, as above
    if (tid == 0) {
      A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
      if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = 0.0f;      // (no iteration ran: the information matrix is the zero the regular path would hand back)
      if (A.out_its) A.out_its[a] = 0;
      if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], (int) LSM2D_CAPACITY_EXCEEDED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      else A.out_status[a] = LSM2D_CAPACITY_EXCEEDED;
    }
    return;
  }
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  for (int s = 0; s < A.n_slices; ++s) {
    const SliceDev& S = A.s[s];
    if (!kHasProj || S.finder != LSM2D_FINDER_PROJECTIVE) continue;
    const int fc = pick_cloud(S.fixed, a);
    // (a set unpacked by this launch: its size comes with the arguments -- the scalar cache may not have seen the count written above)
    project_cloud(S.fixed.xy + S.fixed.start[fc], (A.inline_n1 && S.unpack_src) ? S.unpack_n : S.fixed.count[fc], ident, S.proj, fcan + S.fcan_offset, tid, kAlignBlock);
  }
  __syncthreads();
  if (pq_on) {      // the occupancy bitmap: every thread derives the same cell size and origin from the box, then stamps its points
    const SliceDev& S = A.s[0];
    const int fc = pick_cloud(S.fixed, a), nf = S.fixed.count[fc];
    const float2* fp = S.fixed.xy + S.fixed.start[fc];
    const unsigned k0 = s_pqbb[0], k1 = s_pqbb[1], k2 = s_pqbb[2], k3 = s_pqbb[3];
    const float minx = unordered(k0), miny = unordered(k1), maxx = unordered(k2), maxy = unordered(k3);
    const bool have = k0 <= k2 && k1 <= k3;                           // at least one finite point
    // a skipped tile's points stay farther than max_distance from every fixed point; the distance map pairs a query with the point of a PIXEL whose
    // centre is within max_distance of the query's pixel centre: two pixel diagonals more
    float reach = S.max_distance * 1.002f + 2e-3f;
    if (kHasDist && S.finder == LSM2D_FINDER_DISTMAP) reach += 3.0f / S.fixed.dist.meta[fc].inv_res;
    const float ext = __builtin_fmaxf(maxx - minx, maxy - miny);
    const float g = __builtin_fmaxf((reach + 0.1f) * (1.0f / 6.0f), ext * (1.0f / 125.0f));
    const float ox = minx - g, oy = miny - g, inv_g = 1.0f / g;
    const bool usable = have && g > 0.0f && g < 1e30f && S.max_distance >= 0.0f;
    if (tid == 0) { s_pq[0] = ox; s_pq[1] = oy; s_pq[2] = inv_g; s_pq[3] = usable ? reach : -1.0f; }
    if (usable) for (int i = tid; i < nf; i += kAlignBlock) {
      const float2 p = fp[i];
      if (!(__builtin_fabsf(p.x) < 1e30f && __builtin_fabsf(p.y) < 1e30f)) continue;
      const int cx = (int) ((p.x - ox) * inv_g), cy = (int) ((p.y - oy) * inv_g);       // 1 .. 126 by construction
      if ((unsigned) cx < 128u && (unsigned) cy < 128u) atomicOr(&l_occ[cy * kPqRowWords + (cx >> 5)], 1u << (cx & 31));
    }
  }
  __syncthreads();

  int it = __builtin_amdgcn_readfirstlane(s_it0);
  const int it_cap = __builtin_amdgcn_readfirstlane(A.inlier_runs ? 2 * A.max_it : A.max_it);      // (a scalar: as a select and a shift it lived in a vector register, spilled for the loop's back edge)
  const bool want_dig = A.out_stats != nullptr;      // the digest leaves the kernel through the statistics only
  for (; it < it_cap; ++it) {
    const bool lists_only = kFirstStage && it == A.stage_split;      // the first of two launches enters this iteration for the LENGTH of its unit lists alone
    const bool inl_only = A.inlier_runs && __builtin_amdgcn_readfirstlane(s_phase) != 0;
#if LSM2D_PRIO_BY_PROGRESS == 1
    { const int q = (4 * it) / A.max_it; if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2); else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#elif LSM2D_PRIO_BY_PROGRESS == 2
    { if (2 * it < A.max_it) __builtin_amdgcn_s_setprio(3); else if (4 * it < 3 * A.max_it) __builtin_amdgcn_s_setprio(2); else if (8 * it < 7 * A.max_it) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
    for (int s = 0; s < A.n_slices; ++s) {
      const SliceDev& S = A.s[s];
      const Iso T = s_iso[s];
      const uint32_t salt = (uint32_t) s * 0x632BE5ABu;
      Accum acc; accum_zero(acc);
      float seq_acc = 0.0f;      // kSeq: lane q < 11 of wave 0 holds the slice's running sum of quantity q (in Accum's order); the counts stay in acc
      // kSeq: a matched pair becomes a record instead of being added into the thread's partial sums
      auto seq_pair = [&](float2 pf, float2 nf, float2 pm, float2 nm, float (&t)[kSeqFields]) {
        bool inl; pair_terms(T, pf, nf, pm, nm, S.cauchy != 0, S.tau, inl_only, t, inl);
        ++acc.n_corr; acc.n_in += inl ? 1 : 0; acc.n_out += inl ? 0 : 1;
      };
      // kSeq: the end of a trip -- every thread's record (zeros: no pair) is in LDS behind the first barrier, wave 0 adds the trip's n_rec records in ascending
      // slot, and nobody overwrites them before the second
      // (in halves of kSeqHalf records: 12 KB of LDS instead of 24, three workgroups per CU)
      auto seq_trip = [&](int slot, bool writer, const float (&t)[kSeqFields], int n_rec) {
        for (int h0 = 0; h0 < n_rec; h0 += kSeqHalf) {
          if (writer && slot >= h0 && slot < h0 + kSeqHalf) seq_store(l_rec, slot - h0, t);
          __syncthreads();
          const int left = n_rec - h0;
          if (tid < 64) seq_acc = seq_walk(l_rec, left < kSeqHalf ? left : kSeqHalf, tid, seq_acc);
          __syncthreads();
        }
      };
      LSM2D_PH(2);
      if (kHasProj && ((!kHasNN && !kHasDist && !kHasKd) || S.finder == LSM2D_FINDER_PROJECTIVE)) {
        {
          // HOT: every moving point, every iteration (correspondence_finder_projective_2d.cpp:47-48)
          const int mc = pick_cloud(S.moving, a);
          // clouds of at most one pair per thread (the tracker's clipped scenes) need no lane-chunked copy
          if (kProjCulled) {
            // Round 4: the survivors of the BLOCK-level test as a list in LDS, kept across iterations (see AlignArgs::units_off).  Build, when thread 0 found the
            // slice's transform outside the list's neighbourhood: (A) every thread tests the chunk it owns -- as the per-iteration test of round 3 did, with the
            // margins -- and the surviving chunks are compacted; (B) the nb blocks of every surviving chunk are tested the same way, dealt to the threads in
            // block-major order and compacted in that order: the list.  2 + ceil(nb s / 512) + 1 barriers, a few times per alignment.
            const int lane = tid & 63, wave = tid >> 6;
            uint16_t* units = l_units + s * A.units_stride;
            const int Tm = S.moving.lane_T[mc];
            const int nbs = __builtin_amdgcn_readfirstlane(S.moving.block_stride);
            const int B = cull_block_steps(Tm, nbs), nb = (Tm + B - 1) / B;
            if (__builtin_amdgcn_readfirstlane(s_rebuild[s]) || lists_only) {      // (lists_only: the list the second launch will build first, whatever the kept one covers)
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              const float m_t = A.cull_mt, m_th = A.cull_mth;      // (0 when the lists are not kept: the host sees to that -- a select made here was a vector register held across the iteration)
              const unsigned long long bb = reinterpret_cast<unsigned long long>(S.moving.lane_bounds + (size_t) mc * kAlignBlock);
              float4* bbase = reinterpret_cast<float4*>(((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (bb >> 32)) << 32) |
                                                        (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) bb));
              const u32x4 bw = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(bbase, (short) 0, kAlignBlock * 16, 0x00020000), tid * 16, 0, 0);
              const bool keep = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(bw.x), __uint_as_float(bw.y), __uint_as_float(bw.z), 0.0f), fcan + S.fcan_offset, S.point_distance, m_t, m_th);
              const u64 bal = __ballot(keep);
              { int wv = tid >> 6; asm volatile("" : "+v"(wv)); if (lane == 0) s_wcnt[wv] = __popcll(bal); }      // (the address made here, not in front of the iteration loop and kept)
              __syncthreads();
              int before = 0, n_surv = 0;
#pragma unroll
              for (int w = 0; w < nwaves; ++w) { const int cw = s_wcnt[w]; before += w < wave ? cw : 0; n_surv += cw; }
              n_surv = __builtin_amdgcn_readfirstlane(n_surv);
              if (keep) s_surv[before + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) bal, 0u))] = (uint16_t) tid;
              __syncthreads();
              // (B) test v = (block v / n_surv, survivor v mod n_surv), v = tid, tid + 512, ...; block_compact_pos: one barrier per round, buffers alternating
              const unsigned long long kb = reinterpret_cast<unsigned long long>(S.moving.block_bounds + (size_t) mc * nbs * kAlignBlock);
              float4* kbase = reinterpret_cast<float4*>(((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (kb >> 32)) << 32) |
                                                        (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) kb));
              const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(kbase, (short) 0, nbs * kAlignBlock * 16, 0x00020000);
              int n_units = 0, parity = 1, i = tid, blk = 0;
              while (n_surv > 0 && i >= n_surv && blk < nb) { i -= n_surv; ++blk; }
              const int n_tests = nb * n_surv;
              for (int v0 = 0; v0 < n_tests; v0 += kAlignBlock, parity ^= 1) {
                bool k2 = false; int code = 0;
                if (blk < nb) {
                  const int g = (int) s_surv[i];
                  code = (blk << 9) | g;
                  const u32x4 w4 = __builtin_amdgcn_raw_buffer_load_b128(krs, code * 16, 0, 0);      // entry (blk * 512 + g) of the cloud's block circles
                  k2 = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), 0.0f), fcan + S.fcan_offset, S.point_distance, m_t, m_th);
                }
                const int pos = block_compact_pos(k2, s_wcnt, parity, n_units, tid, nwaves);
                if (k2) units[pos] = (uint16_t) code;
                i += kAlignBlock;
                while (n_surv > 0 && i >= n_surv && blk < nb) { i -= n_surv; ++blk; }
              }
              if (tid == 0) { s_nunits[s] = n_units; s_list_iso[s] = T; }
              __syncthreads();
            }
            if (lists_only) continue;
            const int n_units = __builtin_amdgcn_readfirstlane(s_nunits[s]);
            if (n_units > 0) project_cloud_list(S.moving.lane_xy + S.moving.lane_start[mc], Tm, T, S.proj, mcan, tid, kAlignBlock, units, n_units, B);
          }
          else if (S.moving.lane_xy && S.moving.lane_bounds && A.cull) {
            // exact culling against the fixed canvas (chunk_may_matter): every thread tests the chunk it would stream, the survivors are
            // compacted (two barriers: counts, then the list) and their points spread evenly over the workgroup (project_cloud_units)
            const int lane = tid & 63, wave = tid >> 6;
            // (the chunk's circle through a buffer resource: base in SGPRs, one 32-bit lane offset -- a per-thread 64-bit pointer would be
            // hoisted out of the iteration loop and spilled: 64 registers)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned long long bb = reinterpret_cast<unsigned long long>(S.moving.lane_bounds + (size_t) mc * kAlignBlock);
            float4* bbase = reinterpret_cast<float4*>(((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (bb >> 32)) << 32) |
                                                      (unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) bb));
            const u32x4 bw = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(bbase, (short) 0, kAlignBlock * 16, 0x00020000), tid * 16, 0, 0);
            const bool keep = chunk_may_matter(T, S.proj, make_float4(__uint_as_float(bw.x), __uint_as_float(bw.y), __uint_as_float(bw.z), 0.0f), fcan + S.fcan_offset, S.point_distance);
            const u64 bal = __ballot(keep);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, n_surv = 0;
#pragma unroll
            for (int w = 0; w < nwaves; ++w) { const int cw = s_wcnt[w]; before += w < wave ? cw : 0; n_surv += cw; }
            // (rank below the lane by v_mbcnt: a hoisted 64-bit lane mask would be two more registers held -- and spilled -- across the loops)
            if (keep) s_surv[before + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) bal, 0u))] = (uint16_t) tid;
            __syncthreads();
            const int Tm = S.moving.lane_T[mc];
            // blocks of an even number of steps, 7 per chunk (measured on configs[1], T = 98: blocks of 2 / 4 / 6 / 8 / 14 steps 1.098 / 1.017 /
            // 0.991 / 0.983 / 0.969 ms -- what a unit costs to set up outweighs the better balance of smaller ones; element-wise row-major
            // order, balanced to one step, 1.066: project_cloud_rows, "cull" 2)
            const int B = (kExperiments && A.cull_block > 0) ? A.cull_block : cull_block_steps(Tm), nb = (Tm + B - 1) / B;
            if (n_surv > 0) {
#ifdef LSM2D_EXPERIMENTS
              if (!kProjCulled && A.cull == 2) project_cloud_rows(S.moving.lane_xy + S.moving.lane_start[mc], Tm, T, S.proj, mcan, tid, kAlignBlock, s_surv, n_surv);
              else
#endif
              project_cloud_units(S.moving.lane_xy + S.moving.lane_start[mc], Tm, T, S.proj, mcan, tid, kAlignBlock, s_surv, n_surv, B, nb);
            }
          }
          else if (S.moving.lane_xy) project_cloud_lanes(S.moving.lane_xy + S.moving.lane_start[mc], S.moving.lane_T[mc], T, S.proj, mcan, tid, kAlignBlock);
          else project_cloud(S.moving.xy + S.moving.start[mc], S.moving.count[mc], T, S.proj, mcan, tid, kAlignBlock);
        }
        __syncthreads();
        if (kXcdWindow && xsync && tid == 0 && it * A.n_slices + s < A.xcd_positions)      // this workgroup's pass (it, s) over the map is behind all its waves
          __hip_atomic_fetch_add(&xsync[16 + it * A.n_slices + s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // bin walk (correspondence_finder_projective_2d.cpp:55-74): the fixed side comes from LDS, the two gathers of the
        // moving winner are issued together, and every cell read is reset for the next projection
        const int mbase = S.moving.start[pick_cloud(S.moving, a)], fbase = S.fixed.start[pick_cloud(S.fixed, a)];
        const float2* mn = S.moving.nrm + mbase; const float2* mp = S.moving.xy + mbase;
        const float2* fnr = S.fixed.nrm + fbase; const float2* fpp = S.fixed.xy + fbase;
        const float4* maos = S.moving.aos ? S.moving.aos + mbase : nullptr;      // (a set without its AoS copy -- sizes still pending on the device, or
        const float4* faos = S.fixed.aos ? S.fixed.aos + fbase : nullptr;        //  unpacked by this launch -- is gathered from its split arrays)
        const u64* fcs = fcan + S.fcan_offset;
#ifndef LSM2D_BINWALK_BATCHED
#define LSM2D_BINWALK_BATCHED 1
#endif
        if constexpr (kSeq) {
          // "sum_order" 1: trips of kAlignBlock consecutive columns (every thread takes part in every trip: the barriers), slot = column - first column of the trip
          for (int col0 = 0; col0 < S.proj.cols; col0 += kAlignBlock) {
            const int col = col0 + tid;
            float t[kSeqFields]; seq_zero(t);
            if (col < S.proj.cols) {
              const u64 fk = fcs[col], mk = mcan[col];
              mcan[col] = kEmptyCell;
              const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
              if (mk != kEmptyCell && fk != kEmptyCell && !(__builtin_fabsf(fd - md) > S.point_distance)) {
                const int mi = (int) (uint32_t) mk, fi = (int) (uint32_t) fk;
                float2 nm, pm; float4 f;
                if (maos) { const float4 m4 = maos[mi]; pm = make_float2(m4.x, m4.y); nm = make_float2(m4.z, m4.w); } else { nm = mn[mi]; pm = mp[mi]; }
                if (faos) f = faos[fi]; else { const float2 p2 = fpp[fi], n2 = fnr[fi]; f = make_float4(p2.x, p2.y, n2.x, n2.y); }
                float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
                if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos)) {
                  if (want_dig) digest_add(&s_dig, salt, fi, mi);
                  seq_pair(make_float2(f.x, f.y), make_float2(f.z, f.w), pm, nm, t);
                }
              }
            }
            const int left = S.proj.cols - col0;
            seq_trip(tid, true, t, left < kAlignBlock ? left : kAlignBlock);
          }
        }
        else
        if (LSM2D_BINWALK_BATCHED && kProjCulled && maos && faos) {
          // Round 5: the walk as a two-deep pipeline -- the NEXT column's cells are read and gated (LDS) and its winners' two 16-byte rows asked for BEFORE this
          // column's factor terms are formed; the terms are added in the same column order as before (the sums keep their bits).  A trip used to end in two
          // dependent gathers from L2 that nothing covered: three exposed round trips per thread, iteration and slice; now the second and third travel under
          // the arithmetic of the one before.  (All three trips asked for up front needed 24 registers more than the kernel has: 96 bytes of scratch.)
          auto gate = [&](int col, int& fi, int& mi) -> bool {
            if (col >= S.proj.cols) return false;
            const u64 fk = fcs[col], mk = mcan[col];
            mcan[col] = kEmptyCell;
            if (mk == kEmptyCell || fk == kEmptyCell) return false;
            const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
            if (__builtin_fabsf(fd - md) > S.point_distance) return false;
            mi = (int) (uint32_t) mk; fi = (int) (uint32_t) fk;
            return true;
          };
          int fi_a = 0, mi_a = 0, fi_b = 0, mi_b = 0;
          float4 m_a = make_float4(0.f, 0.f, 0.f, 0.f), f_a = m_a, m_b = m_a, f_b = m_a;
          bool ok_a = gate(tid, fi_a, mi_a);
          if (ok_a) { m_a = maos[mi_a]; f_a = faos[fi_a]; }
          for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
            const bool ok_b = gate(col + kAlignBlock, fi_b, mi_b);
            if (ok_b) { m_b = maos[mi_b]; f_b = faos[fi_b]; }
            if (ok_a) {
              const float2 pm = make_float2(m_a.x, m_a.y), nm = make_float2(m_a.z, m_a.w);
              float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
              if (!(__builtin_fmaf(nqx, f_a.z, nqy * f_a.w) < S.normal_cos)) {
                if (want_dig) digest_add(&s_dig, salt, fi_a, mi_a);
                accumulate_pair(T, make_float2(f_a.x, f_a.y), make_float2(f_a.z, f_a.w), pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
              }
            }
            ok_a = ok_b; fi_a = fi_b; mi_a = mi_b; m_a = m_b; f_a = f_b;
          }
        }
        else
        for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
          const u64 fk = fcs[col], mk = mcan[col];
          mcan[col] = kEmptyCell;
          if (mk == kEmptyCell || fk == kEmptyCell) continue;
          const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
          if (__builtin_fabsf(fd - md) > S.point_distance) continue;
          const int mi = (int) (uint32_t) mk, fi = (int) (uint32_t) fk;
          float2 nm, pm; float4 f;
          if (maos) { const float4 m4 = maos[mi]; pm = make_float2(m4.x, m4.y); nm = make_float2(m4.z, m4.w); } else { nm = mn[mi]; pm = mp[mi]; }
          if (faos) f = faos[fi]; else { const float2 p2 = fpp[fi], n2 = fnr[fi]; f = make_float4(p2.x, p2.y, n2.x, n2.y); }
          float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
          if (__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos) continue;
          if (want_dig) digest_add(&s_dig, salt, fi, mi);
          accumulate_pair(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
        }
      } else if (kHasNN || kHasDist || kHasKd) {
        // NN finder fused with the factor (correspondence_finder_kd_tree_2d.cpp:12-27): every moving point is
        // transformed, matched to its exact nearest fixed point within max_distance, normal-gated, accumulated
        const int fc = pick_cloud(S.fixed, a), mc = pick_cloud(S.moving, a);
        const int mbase = S.moving.start[mc], fbase = S.fixed.start[fc];
        const float2* fn = S.fixed.nrm + fbase; const float2* mn = S.moving.nrm + mbase;
        const float2* fp = S.fixed.xy + fbase;  const float2* mp = S.moving.xy + mbase;
        const bool use_grid = kHasNN && ((!kHasDist && !kHasKd) || S.finder == LSM2D_FINDER_NN);
        const bool use_kd = kHasKd && ((!kHasNN && !kHasDist) || S.finder == LSM2D_FINDER_KDTREE);
        GridMeta g; DistMeta dm;
        const int32_t* cst = nullptr; const int32_t* sidx = nullptr; const float2* sxy = nullptr;
        const KdNode* knd = nullptr; const float2* knr = nullptr;
        if (use_kd) {      // the reference's tree over the fixed cloud (correspondence_finder_kd_tree_2d.cpp:18-19): one descent + one leaf per query
          knd = S.fixed.kd.nodes + __builtin_amdgcn_readfirstlane(S.fixed.kd.meta[fc].node_base);
          sxy = S.fixed.kd.leaf_xy + fbase; knr = S.fixed.kd.leaf_nrm + fbase;
        } else if (use_grid) {
          g = S.fixed.grid.meta[fc];
          // the meta comes through a vector load: tell the compiler it is wave-uniform -- seven VGPRs fewer across the query loops, which
          // takes the last spills out of them (NN role B 2.27 -> 2.06 ms, role A 7.91 -> 7.82; variants_r02v_nn_scalar_meta.log)
          g.minx = uniform_f(g.minx); g.miny = uniform_f(g.miny); g.h = uniform_f(g.h); g.inv_h = uniform_f(g.inv_h);
          g.gw = __builtin_amdgcn_readfirstlane(g.gw); g.gh = __builtin_amdgcn_readfirstlane(g.gh); g.cell_base = __builtin_amdgcn_readfirstlane(g.cell_base);
          cst = S.fixed.grid.cell_start + g.cell_base;
          sidx = S.fixed.grid.sorted_idx + fbase; sxy = S.fixed.grid.sorted_xy + fbase;
          if (kNNGlobal) knr = S.fixed.grid.sorted_nrm + fbase;
        } else {
          dm = S.fixed.dist.meta[fc];       // distance-map finder: one lookup per query (correspondence_finder_nn_2d.cpp:63-80)
        }
        const float md2 = S.max_distance * S.max_distance;
        const int nm_pts = S.moving.count[mc];
        // cooperative search (kNNGroup lanes per query) when THIS alignment's fixed cloud is at least four times its moving one --
        // decided per alignment from the device-side counts, so ragged batches get the right loop for each cloud (the oracle's
        // device-order mode applies the same rule)
        const bool coop = use_grid && (long long) S.fixed.count[fc] >= 4ll * nm_pts;
        // this iteration's keep bits, one per tile of 64 moving points (see the prologue): bit t of l_keep <-> tile t
        const int n_tiles = (nm_pts + 63) >> 6;
        const bool pq_cull = pq_on && S.moving.tile_bounds && !coop && ((n_tiles + kAlignBlock - 1) / kAlignBlock) * (kAlignBlock / 64) <= A.pq_keep_words;
        if (pq_cull) {
          const float4* tb = S.moving.tile_bounds + S.moving.tile_start[mc];
          const float ox = s_pq[0], oy = s_pq[1], inv_g = s_pq[2], reach = s_pq[3];
          for (int t0 = 0; t0 < n_tiles; t0 += kAlignBlock) {
            const int t = t0 + tid; bool keep = false;
            if (t < n_tiles) {
              const float4 b = tb[t];
              keep = true;
              const float kf = (b.z * 1.002f + reach) * inv_g;      // cells the tile's reach spans (rho = +inf, a tile with a non-finite point: no claim)
              if (reach >= 0.0f && kf < 10.0f) {
                float qx, qy; xf_point(T, b.x, b.y, qx, qy);
                const float fx = __builtin_floorf((qx - ox) * inv_g), fy = __builtin_floorf((qy - oy) * inv_g);
                if (fx > -64.0f && fx < 192.0f && fy > -64.0f && fy < 192.0f) {
                  const int k = (int) kf + 1, cx = (int) fx, cy = (int) fy;
                  const int x0 = cx - k < 0 ? 0 : cx - k, x1 = cx + k > 127 ? 127 : cx + k, y0 = cy - k < 0 ? 0 : cy - k, y1 = cy + k > 127 ? 127 : cy + k;
                  keep = false;
                  if (x0 <= x1) {
                    const u64 mask = (~0ull >> (63 - (x1 - x0))) << (x0 & 31);      // at most 21 bits, from bit x0 of the two-word window
                    for (int y = y0; y <= y1; ++y) {
                      const uint32_t* row = l_occ + y * kPqRowWords + (x0 >> 5);
                      if ((((u64) row[1] << 32) | (u64) row[0]) & mask) { keep = true; break; }
                    }
                  }
                }
                else keep = !(fx == fx && fy == fy);      // far beyond the bitmap: nothing within reach; not a number: no claim
              }
            }
            const u64 bal = __ballot(keep);
            if ((tid & 63) == 0) l_keep[(t0 >> 6) + (tid >> 6)] = bal;
          }
          __syncthreads();
        }
        // the grid search is cooperative on dense fixed clouds (kNNGroup lanes per query); the distance map is one lookup
        auto query_loop = [&](auto group_tag) {
          constexpr int group = decltype(group_tag)::value;
          const int sub = tid & (group - 1);
          constexpr int per_step = kAlignBlock / group;
          for (int j0 = 0; j0 < nm_pts; j0 += per_step) {
            bool skip = false;                       // kSeq: a culled tile's wave still takes part in the trip (its records are zeros, the barriers are everybody's)
            float t[kSeqFields];                     // kSeq: this thread's record of the trip
            if constexpr (kSeq) seq_zero(t);
            if (group == 1 && pq_cull) {             // this wave's 64 queries of the trip are one tile
              const int tile = (j0 >> 6) + __builtin_amdgcn_readfirstlane(tid >> 6);
              const u64 w = l_keep[tile >> 6];
              const unsigned half = (tile & 32) ? (unsigned) (w >> 32) : (unsigned) w;
              if (!((__builtin_amdgcn_readfirstlane((int) half) >> (tile & 31)) & 1)) { if constexpr (kSeq) skip = true; else continue; }
            }
            const int j = j0 + tid / group;
            const bool live = j < nm_pts && !skip;           // whole groups are live or not: the shuffles inside stay uniform
            const float2 pm = live ? mp[j] : make_float2(0.0f, 0.0f);
            float qx, qy; xf_point(T, pm.x, pm.y, qx, qy);
            int best = -1;
            if (use_kd) {      // the match's point and normal come from the leaf arrays, where the scan found it: the original index is never needed
              if (live) {
                float2 bxy;
                const int pos = kKdAllLds ? kd_query_pos<true>(knd, l_kxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds)
                              : kKdTop ? kd_query_pos(knd, sxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds)
                              : kd_leaves_lds ? kd_query_pos<true>(knd, l_kxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds) : kd_query_pos(knd, sxy, qx, qy, md2, bxy, l_kpl, l_klk, kd_lds);
                if (pos >= 0) {
                  const float2 nm = mn[j], nf = kKdAllLds ? l_knr[pos] : (kKdTop ? knr[pos] : (kd_leaves_lds ? l_knr[pos] : knr[pos]));
                  float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
                  const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
                  if (!(dot < S.normal_cos)) {
                    if (want_dig) digest_add(&s_dig, salt, S.fixed.kd.leaf_idx[fbase + pos], j);      // the original index: only the digest asks for it
                    if constexpr (kSeq) seq_pair(bxy, nf, pm, nm, t); else
                    accumulate_pair(T, bxy, nf, pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
                  }
                }
              }
            } else
            if (kNNGlobal) {      // the match's point and normal come from where the search found it: no detour through the original index
              const int pos = live ? nn_query_pos<group>(g, cst, sidx, sxy, qx, qy, S.max_distance, md2, sub, j < A.nn_qcache ? l_qc + 8 * j : nullptr) : -1;
              if (pos >= 0 && sub == 0) {                    // one lane per query accumulates
                const float2 nm = mn[j], nf = knr[pos], pf = sxy[pos];
                float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
                const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
                if (!(dot < S.normal_cos)) {
                  if (want_dig) digest_add(&s_dig, salt, sidx[pos], j);
                  if constexpr (kSeq) seq_pair(pf, nf, pm, nm, t); else
                  accumulate_pair(T, pf, nf, pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
                }
              }
            } else
            if (kNNLds) { if (live) best = nn_query<1, uint16_t, uint16_t>(g, l_cst, l_sidx, l_sxy, qx, qy, S.max_distance, md2, 0); }
            else
            if (use_grid) {
              if (live) best = (group == 1 && nn_lds) ? nn_query<1, uint16_t, uint16_t>(g, l_cst, l_sidx, l_sxy, qx, qy, S.max_distance, md2, 0)
                                                      : nn_query<group>(g, cst, sidx, sxy, qx, qy, S.max_distance, md2, sub);
            }
            else if (live) best = distmap_lookup(dm, S.fixed.dist.parent, qx, qy);
            if (best >= 0 && sub == 0) {                     // one lane per query accumulates
              const float2 nm = mn[j], nf = fn[best];
              float nqx, nqy; xf_normal(T, nm.x, nm.y, nqx, nqy);
              const float dot = __builtin_fmaf(nqx, nf.x, nqy * nf.y);
              if (!(dot < S.normal_cos)) {
                if (want_dig) digest_add(&s_dig, salt, best, j);
                if constexpr (kSeq) seq_pair(fp[best], nf, pm, nm, t); else
                accumulate_pair(T, fp[best], nf, pm, nm, S.cauchy != 0, S.tau, acc, inl_only);
              }
            }
            if constexpr (kSeq) {      // "sum_order" 1: the trip's queries are per_step consecutive moving indices, slot = query - first query of the trip
              const int left = nm_pts - j0;
              seq_trip(tid / group, sub == 0, t, left < per_step ? left : per_step);
            }
          }
        };
        // (the cooperative search on a scan-sized fixed cloud with its tables in LDS: 2 / 4 / 8 lanes per query take 21 / 41 / 90 ms against
        // 8.1 ms with one lane per query -- the time goes with the number of wave-queries, i.e. into the fixed cost of a query, not its candidates)
        // (several queries of a thread in flight together -- all points, then all pixels, then all parents -- measured with the registers
        // for it: 4 waves per SIMD and 3-4 trips tie with this loop at 8 waves per SIMD on role B and lose 10-50 % elsewhere; at 8 waves per SIMD
        // with only the parents' indices kept live, 2 / 3 trips take 0.29 / 0.38 ms against 0.21 on role B and lose on role A too; DESIGN App. A)
        // (matched pairs queued per wave in LDS and added up 64 at a time with every lane busy, instead of ~70 instructions of accumulate_pair on
        // every trip for the quarter of the lanes that matched: slower everywhere -- distance map role A 5.21 -> 5.87 ms, NN role A 7.95 -> 8.61,
        // distance map role B 0.21 -> 0.30: the ballot, the queue and the reloads cost more than the idle lanes; DESIGN App. A)
        if (!kNNLds && coop) query_loop(std::integral_constant<int, kNNGroup>{});
        else query_loop(std::integral_constant<int, 1>{});
      }
      LSM2D_PH(0);
      // (a projective slice's thread accumulates at most ceil(cols / block) pairs: its counts are a few bits, summed by ballots)
      block_reduce_store(acc, red, tid, (kHasProj && !kHasNN && !kHasDist && !kHasKd) ? 32 - __builtin_clz(((S.proj.cols + kAlignBlock - 1) / kAlignBlock) | 1) : 0);
      __syncthreads();
      if (tid < 64) {
        // lanes 0..13 of wave 0 each add one quantity over the waves (wave order) and then into the iteration's sum themselves
        float v; int vi; block_reduce_gather_lane(red, nwaves, tid, v, vi);
        if constexpr (kSeq) v = seq_acc;      // (the partial sums' floats were never touched: the eleven quantities are the walker's)
        const int n_corr = __builtin_amdgcn_readlane(vi, 13);
        if (tid == 0) s_n_corr += n_corr;
        if (n_corr > S.min_corr) {     // slices with #pairs <= min_num_correspondences are skipped
          int ts = tid; asm volatile("" : "+v"(ts));      // (s_sum's address for this lane made here: hoisted, it was spilled across the iteration)
          if (tid < 11) s_sum[ts] += v;
          else if (tid < 13) s_sum[ts] = __int_as_float(__float_as_int(s_sum[ts]) + vi);
          if (tid == 0) ++s_active;
        }
      }
      LSM2D_PH(1);
      // pure projective kernels need no barrier here: the other waves go on to the next slice's projection (the cells it
      // writes were reset by the bin walk) and touch `red` again only after the barrier that follows it, which lane 0
      // joins once it is done with the partials.  The point-query branches write `red` without such a barrier in between.
      if (kHasNN || kHasDist || kHasKd) __syncthreads();
    }
    if (lists_only) {      // what the second launch will stream per iteration -> its placement; the state it goes on from
      if (tid == 0) {
        ResumeDev R;
        R.pose[0] = s_pose[0]; R.pose[1] = s_pose[1]; R.pose[2] = s_pose[2];
        for (int k = 0; k < 9; ++k) R.H[k] = s_H[k];
        R.prev_chi = s_prev_chi; R.phase = s_phase; R.phase_start = s_phase_start; R.phase_end = s_phase_end;
        R.last_n_in = s_last_n_in; R.status = s_status; R.done = 0; R.it = it;
        A.resume[a] = R;
        int units = 0;
        for (int s = 0; s < A.n_slices; ++s) units += s_nunits[s];
        const int w = (units + 7 * A.n_slices - 1) / (7 * A.n_slices);      // <= 512: a slice's list holds at most kCullBlocks x 512 units
        A.stage_work[a] = w < 1 ? 1 : (w > kAlignBlock ? kAlignBlock : w);
      }
      return;
    }
    if (tid == 0) {
      // (thread 0's serial state lives in LDS, not in registers every thread would carry -- and spill -- across the loops)
      StatsDev last; last.n_corr = s_n_corr; last.n_in = __float_as_int(s_sum[11]); last.n_out = __float_as_int(s_sum[12]); last.chi_in = s_sum[9]; last.chi_out = s_sum[10];
      s_last_n_in = last.n_in;
#ifdef LSM2D_DEBUG_UNITS      // diagnostics build: the culled stream's list length and whether it was rebuilt, in place of the outlier statistics
      if (kProjCulled) { last.n_out = s_nunits[0]; last.chi_out = (float) s_rebuild[0]; }
#endif
      if (A.out_stats) { const u64 dg = s_dig; last.dig_lo = (uint32_t) dg; last.dig_hi = (uint32_t) (dg >> 32); A.out_stats[(size_t) a * A.stats_stride + it] = last; }
      if (!s_active) { s_status = LSM2D_NOT_ENOUGH_CORRESPONDENCES; s_done = 1; }
      else {
        // information matrix = H of the last iteration: assembled, given its prior and solved where it lies
        s_H[0] = s_sum[0]; s_H[1] = s_sum[1]; s_H[2] = s_sum[2]; s_H[3] = s_sum[1]; s_H[4] = s_sum[3]; s_H[5] = s_sum[4];
        s_H[6] = s_sum[2]; s_H[7] = s_sum[4]; s_H[8] = s_sum[5];
        s_rhs[0] = s_sum[6]; s_rhs[1] = s_sum[7]; s_rhs[2] = s_sum[8];
        if (A.prior) add_prior(s_prior, s_pose, s_H, s_rhs);
        float dmp = A.damping;
        asm volatile("" : "+v"(dmp));      // (not a loop invariant to hoist -- as a double it was kept, and spilled, across the whole kernel)
        if (!solve_update(s_H, s_rhs, dmp, s_pose)) { s_status = LSM2D_SINGULAR_H; s_done = 1; }
        else {
          bool phase_over = it + 1 >= s_phase_end;
          if (A.term_eps > 0.0f) {      // the aligner's termination criterion: relative decay of the total chi^2 (lsm2d.h), afresh in every phase
            const float chi_now = s_sum[9] + s_sum[10];      // (= last.chi_in + last.chi_out, read again: kept in registers across the solve they were spilled)
            if (it > s_phase_start && __builtin_fabsf(s_prev_chi - chi_now) < A.term_eps * chi_now) phase_over = true;      // status stays RUNNING: decided below as after max_iterations
            s_prev_chi = chi_now;
          }
          if (phase_over) {
            // enable_inlier_only_runs (lsm2d.h): the regular loop ended without a failure and with enough inliers -> up to max_it iterations over inliers only
            if (A.inlier_runs && s_phase == 0 && last.n_in >= A.min_inliers) { s_phase = 1; s_phase_start = it + 1; s_phase_end = it + 1 + A.max_it; }
            else s_done = 1;
          }
        }
      }
      if (!s_done) begin_iteration();        // next iteration's transforms and zeroed sums, under the same barrier
      // the XCD lockstep: nobody starts its next pass before everybody on this XCD has finished the pass xcd_window back (the others stand at the barrier below anyway)
      if (kXcdWindow && xsync && !s_done) xcd_wait(xsync, (it + 1) * A.n_slices - 1 - A.xcd_window * A.n_slices, A.xcd_positions);
    }
    LSM2D_PH(2);
    __syncthreads();
    if (s_done) { ++it; break; }
  }
  if (kXcdWindow && xsync && tid == 0) __hip_atomic_fetch_add(&xsync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // gone: nobody waits for this workgroup any more
  if (tid == 0) {
    int st = s_status;
    if (st == LSM2D_RUNNING) st = (A.max_it > 0 && s_last_n_in < A.min_inliers) ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    A.out_pose[3 * a + 0] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
    if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = s_H[k];
    if (A.out_its) A.out_its[a] = it;
    if (kFirstStage) { A.resume[a].done = 1; A.stage_work[a] = 0; }      // finished before the second launch: its workgroup there leaves at once
    // the status goes last, behind a system-scope release: with results written straight to pinned host memory the host polls
    // this word instead of waiting for the stream (lsm2d_align_batch), and whoever sees it sees everything above
    if (stamp) {
      unsigned long long* co = A.clock_out + 4 * (a / A.clock_stride);
      co[0] = __builtin_amdgcn_s_memtime() - s_clk[0];
      co[1] = __builtin_amdgcn_s_memrealtime() - s_clk[1];
      co[2] = s_clk[1];                                                            // when it started (100 MHz ticks): which dispatch round it was in
#ifdef LSM2D_PHASE_PROBE
      co[1] = s_ph[0]; co[2] = s_ph[1]; co[3] = s_ph[2];      // cycles: query phase, barrier + reduction, solve + the rest (co[0] stays the lifetime)
      if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } else A.out_status[a] = st;
      return;
#endif
      co[3] = (unsigned long long) __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |      // HW_REG_HW_ID (id 4): wave / SIMD / CU / SE it ran on
              ((unsigned long long) __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);  // HW_REG_XCC_ID (id 20)
    }
    if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out_status[a], st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    else A.out_status[a] = st;
  }
}
// Registers: 8 waves per SIMD (64 VGPRs, four workgroups per CU) for every single-finder instantiation.  The MIXED instantiations (a projective slice next to a
// point-query slice in one aligner: all forms of all finders in one body) spilled 176 bytes per thread at that budget; they are given 4 waves per SIMD
// (128 VGPRs, two workgroups per CU) and have no private segment -- a configuration no BASELINE workload uses (round 5; tools/isa_dump.sh prints every kernel's frame).
#ifndef LSM2D_MIXED_MIN_WAVES
#define LSM2D_MIXED_MIN_WAVES 4
#endif
#ifndef LSM2D_NNGLOBAL_MIN_WAVES
#define LSM2D_NNGLOBAL_MIN_WAVES LSM2D_QUERY_MIN_WAVES      // k_align<0,1,0,0,1>: A/B knob (6: 80 VGPRs, no frame, three workgroups per CU)
#endif
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd, int kNNMode>
constexpr int align_min_waves() {
  return (kHasProj && (kHasNN || kHasDist || kHasKd)) ? LSM2D_MIXED_MIN_WAVES : kHasProj ? LSM2D_ALIGN_MIN_WAVES : kNNMode == 1 ? LSM2D_NNGLOBAL_MIN_WAVES : LSM2D_QUERY_MIN_WAVES;
}
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd = false, int kNNMode = 0>
__global__ __launch_bounds__(kAlignBlock, (align_min_waves<kHasProj, kHasNN, kHasDist, kHasKd, kNNMode>())) void k_align(const AlignArgs A) {
  align_body<kHasProj, kHasNN, kHasDist, kHasKd, kNNMode, false>(A);
}
// "sum_order" 1: the same kernel with the reference's order of summation (align_body<.., kSeq = true>).  14 KB of pair records per workgroup beside the canvases (four workgroups per CU still fit the headline's shape):
// the register budget stays that of 8 waves per SIMD -- 4 for the mixed instantiations, as above
template <bool kHasProj, bool kHasNN, bool kHasDist, bool kHasKd = false, int kNNMode = 0>
#ifndef LSM2D_SEQ_MIN_WAVES
#define LSM2D_SEQ_MIN_WAVES 8
#endif
__global__ __launch_bounds__(kAlignBlock, ((kHasProj && (kHasNN || kHasDist || kHasKd)) ? LSM2D_MIXED_MIN_WAVES : LSM2D_SEQ_MIN_WAVES)) void k_align_seq(const AlignArgs A) {
  align_body<kHasProj, kHasNN, kHasDist, kHasKd, kNNMode, false, true>(A);
}
// Round 4 (late): TWO launches for a culled batch of about one dispatch round.  The placement of such a batch decides its tail (the launch lasts as long as
// the CU with the largest sum of work), and what an alignment will stream is known badly at its START pose -- the estimate of k_cull_estimate left a tail of
// 10 % -- but well after ONE Gauss-Newton iteration, which takes most of the start error out.  So: this kernel runs iteration 0 of every alignment in any
// order (a twentieth of the work: its own tail does not matter), builds the unit lists of iteration 1 for their length alone, and leaves pose, phase and
// statistics state in ResumeDev; k_balance_only deals the alignments out by those lengths; k_align runs the other nineteen iterations from the saved state.
// The same arithmetic on the same values in the same order: every result keeps its bits (the second launch rebuilds its lists; a list is a superset of
// what can pair, whatever pose within its margins it was built at).  MEASURED AND NOT SHIPPED ("two_stage" 0 by default): the second launch takes 0.698 ms
// instead of 0.745 -- but a twentieth of that is the iteration it no longer runs, its tail is still 7 % (the length of a list is not the whole of an
// alignment's cost), and this kernel takes 95 us for its twentieth of the work: all thousand workgroups are in the same phase at the same time, and the
// phases that wait (prologue, list building, barriers) have no other workgroup's stream to hide under.  0.861 vs 0.836 ms per step.
#ifdef LSM2D_EXPERIMENTS
__global__ __launch_bounds__(kAlignBlock, LSM2D_ALIGN_MIN_WAVES) void k_first_iteration(const AlignArgs A) {
  align_body<true, false, false, false, 5, true>(A);
}
#endif

// ---- balanced placement for culled batches -------------------------------------------------------------------------------------------
// With the exact culling an alignment's work depends on its pose and scan (33 .. 59 % of the map's chunks survive on configs[1]), and a
// batch of about one workgroup per slot of the chip runs in ONE dispatch round: the CU that happens to get four heavy alignments ends the
// launch (workgroup lifetimes 0.64 .. 1.07 ms in one launch, tools/occupancy_probe.py).  One small launch ahead of k_align fixes that:
// k_cull_estimate counts, per alignment, the chunks that survive at the START pose under the margins A.cull_est_mt / cull_est_mth (what the
// iterations will stream while the pose moves by centimetres), and the workgroup that finishes LAST (an agent-scope counter) ranks the
// alignments by that count and deals them to workgroup ids (balance_order) so that the ids which share a CU carry about the same sum.
//
// Round 4.  WHICH workgroup ids share a CU is the dispatcher's business: the probe of round 3 saw b, b + n_cu, b + 2 n_cu, ...; with this round's
// smaller LDS footprint the groups look irregular ([0, 394, 527, 763], ...) -- but they are THE SAME from launch to launch
// (tools/mapping_stability_probe.py: 256 of 256 groups identical over six launches, although the CUs' names permute).  So every k_align workgroup
// notes where it ran (place_key, one store per workgroup), and the next call of the same shape groups the first round's workgroup ids by what the
// previous launch noted; no notes yet (first call of a shape): the round-3 assumption.  Within the groups the alignments are dealt level by level:
// the k-th member of every group takes one of the next-lighter block of alignments, and the group that carries most so far takes the lightest of the block (a
// group with fewer members -- 1000 alignments on 1024 slots leave 24 CUs with three workgroups -- carries less and so draws the heavier ones); for equal
// loads this is the boustrophedon of round 3.  Beyond the first round the heaviest go first.
// Only WHERE an alignment runs changes; every result is the same.
// Measured on configs[1] (profiles/r04/balance_ab_r04n.txt; k_align alone / whole step): no placement 0.793 / 0.842 ms; round-3 grouping, margins 3 cm and
// 0.02 rad 0.762 / 0.848; noted grouping, margins 0 and 0.04 rad (the defaults) 0.751 / 0.833.  The estimate's own launch is 35 us of the step.
static constexpr int kBalMaxFirst = 1024, kBalMaxLevels = 8, kBalMaxGroups = 512;
struct BalanceLds {                               // < 40 KB: four workgroups of k_cull_estimate per CU, the whole batch in one dispatch round
  union { int bin[kAlignBlock + 2]; int gsize[kBalMaxGroups]; };      // (the bins are done with when the groups are formed)
  int sorted[kBalMaxFirst];                       // the first round's alignments, heaviest first
  unsigned short sw[kBalMaxFirst];                // ... and their counts
  union {
    unsigned short wall[2048];                    // the counts of alignments 0 .. 2047 (one agent-scope load each; beyond: loaded twice) -- until the ranks are out
    unsigned short assign[kBalMaxGroups * kBalMaxLevels];      // rank (in `sorted`) of the alignment in (group, slot) -- afterwards
  };
  unsigned int cnt[kPlaceKeys / 4];               // members per place key (8 bits each; more than 8 on a key: fallback)
  unsigned short gid[kPlaceKeys];                 // the key's dense group id
  unsigned short wg_key[kBalMaxFirst]; unsigned char wg_slot[kBalMaxFirst];
  int gload[kBalMaxGroups];
  __attribute__((aligned(16))) int gproj[kBalMaxGroups];
  int grank[kBalMaxGroups];
  int lvl[kBalMaxLevels + 1];
  int ngroups, bad;
};
static_assert(sizeof(BalanceLds) <= 39 * 1024, "k_cull_estimate: four workgroups per CU");
LSM2D_DEV void balance_order(BalanceLds& L, const int32_t* work, int n, int n_cu, int per_cu, int32_t* __restrict__ order, const int32_t* place, int tid, int nt) {
  for (int i = tid; i < kAlignBlock + 2; i += nt) L.bin[i] = 0;
  for (int i = tid; i < kPlaceKeys / 4; i += nt) L.cnt[i] = 0;
  for (int i = tid; i < kBalMaxGroups; i += nt) L.gproj[i] = INT_MIN;
  if (tid <= kBalMaxLevels) L.lvl[tid] = 0;
  if (tid == 0) { L.ngroups = 0; L.bad = 0; }
  __syncthreads();
  // (the counts were written by other workgroups, on other XCDs: agent-scope loads)
  for (int a = tid; a < n; a += nt) {
    const int w = __hip_atomic_load(&work[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a < 2048) L.wall[a] = (unsigned short) w;
    atomicAdd(&L.bin[w], 1);
  }
  __syncthreads();
  if (tid < 64) {      // exclusive prefix over the bins, heaviest first: one wave, 9 bins per lane
    constexpr int kPer = (kAlignBlock + 2 + 63) / 64;
    int c[kPer], sum = 0;
    #pragma unroll
    for (int j = 0; j < kPer; ++j) { const int w = kAlignBlock + 1 - (tid * kPer + j); c[j] = w >= 0 ? L.bin[w] : 0; sum += c[j]; }
    int incl = sum;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
    int pos = incl - sum;
    #pragma unroll
    for (int j = 0; j < kPer; ++j) { const int w = kAlignBlock + 1 - (tid * kPer + j); if (w >= 0) L.bin[w] = pos; pos += c[j]; }
  }
  __syncthreads();
  int first = n < n_cu * per_cu ? n : n_cu * per_cu;      // the ranks that go out in the first dispatch round
  if (first > kBalMaxFirst) first = kBalMaxFirst;
  for (int a = tid; a < n; a += nt) {
    const int w = a < 2048 ? (int) L.wall[a] : __hip_atomic_load(&work[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int r = atomicAdd(&L.bin[w], 1);                      // rank among all alignments (ties in any order: placement only)
    if (r < first) { L.sorted[r] = a; L.sw[r] = (unsigned short) w; } else order[r] = a;      // beyond the first round: the heaviest go first
  }
  // the groups of the first round's workgroup ids: by the previous launch's notes, or by the round-3 assumption
  for (int b = tid; b < first; b += nt) {
    const int key = place ? (place[b] & (kPlaceKeys - 1)) : (b % n_cu);
    const unsigned int old = atomicAdd(&L.cnt[key >> 2], 1u << (8 * (key & 3)));
    const int slot = (int) ((old >> (8 * (key & 3))) & 0xFFu);
    if (slot >= kBalMaxLevels) L.bad = 1;      // (checked before anything reads a byte that overflowed into its neighbour)
    L.wg_key[b] = (unsigned short) key; L.wg_slot[b] = (unsigned char) slot;
  }
  __syncthreads();
  const bool bad = L.bad != 0;                      // (uniform: read after the barrier)
  if (!bad) for (int kk = tid; kk < kPlaceKeys / 4; kk += nt) {
    const unsigned int four = L.cnt[kk];
    if (four) {
      #pragma nounroll
      for (int q = 0; q < 4; ++q) {
        const int c = (int) ((four >> (q * 8)) & 0xFFu);
        if (c > 0) {
          const int g = atomicAdd(&L.ngroups, 1);
          if (g < kBalMaxGroups) {
            L.gsize[g] = c; L.gload[g] = 0; L.gid[4 * kk + q] = (unsigned short) g;
            #pragma nounroll
            for (int k = 0; k < c; ++k) atomicAdd(&L.lvl[k + 1], 1);
          }
        }
      }
    }
  }
  __syncthreads();
  const int G = L.ngroups;
  if (bad || G > kBalMaxGroups) {                // notes this cannot use (more than 8 workgroups on a CU, more than 512 CUs): the plain heaviest-first order
    for (int r = tid; r < first; r += nt) order[r] = L.sorted[r];
    return;
  }
  if (tid == 0) { for (int k = 0; k < kBalMaxLevels; ++k) L.lvl[k + 1] += L.lvl[k]; }      // lvl[k+1] held the groups with a k-th member: now level offsets
  __syncthreads();
  // (this runs once per launch in ONE workgroup while the chip waits: measured with clock stamps, a rank loop of 256 tie-breaking compares per thread was
  // 5.5 us per level; unique keys and the split between threads below: one compare per key.  The level loop stays a loop: unrolled it was 8200 instructions)
  // every group's rank by its load (descending; ties by id): the keys are unique, a rank is a count of larger keys; 128-bit LDS reads, a wave on one address
  const int G4 = (G + 3) >> 2;
  auto rank_groups = [&](int k_members) {      // groups with more than k_members members take part (-1: all)
    if (tid < kBalMaxGroups) { L.gproj[tid] = (tid < G && L.gsize[tid] > k_members) ? L.gload[tid] * kBalMaxGroups + (kBalMaxGroups - 1 - tid) : INT_MIN; L.grank[tid] = 0; }
    __syncthreads();
    const int parts = G * 2 <= nt ? 2 : 1, per = nt / parts, g = tid % per, part = tid / per;
    if (g < G && L.gsize[g] > k_members) {
      const int mine = L.gproj[g];
      const int4* gp = reinterpret_cast<const int4*>(L.gproj);
      const int q0 = part * G4 / parts, q1 = (part + 1) * G4 / parts;
      int r = 0;
#pragma unroll 2
      for (int q = q0; q < q1; ++q) { const int4 v = gp[q]; r += (v.x > mine ? 1 : 0) + (v.y > mine ? 1 : 0) + (v.z > mine ? 1 : 0) + (v.w > mine ? 1 : 0); }
      if (parts == 1) L.grank[g] = r; else atomicAdd(&L.grank[g], r);
    }
    __syncthreads();
  };
  // (1) the deal, level by level: the k-th member of every group takes one of the next-lighter block of alignments, the group that carries most the lightest
#pragma nounroll
  for (int k = 0; k < kBalMaxLevels; ++k) {
    const int base = L.lvl[k], m = L.lvl[k + 1] - base;
    if (m == 0) break;
    rank_groups(k);
    if (tid < G && L.gsize[tid] > k) {
      const int r = base + (m - 1 - L.grank[tid]);
      L.assign[tid * kBalMaxLevels + k] = (unsigned short) r; L.gload[tid] += L.sw[r];
    }
    __syncthreads();
  }
  // (A refinement of the deal was built and measured -- rank the groups by sum, pair the i-th heaviest with the i-th lightest, let each pair make the one
  // exchange that brings its sums closest, three or six rounds: the sums of configs[1]'s CUs with four workgroups go from 899 .. 1015 to 967 .. 1003, the
  // launch gains 1 %, and the rounds cost 17 .. 23 us of the step's 830: dropped.  What decides a CU's end is the sum it carries -- end = const + slope x sum,
  // the constant the same for CUs with three and with four workgroups (tools/balance_probe.py) -- and at equal ESTIMATED sums the sums of the units really
  // streamed still differ by 2.4 .. 3.3 % rms: the tail that is left, ~5 % over 256 CUs, is the estimate's, made at the start pose, not the deal's.)
  for (int b2 = tid; b2 < first; b2 += nt) order[b2] = L.sorted[L.assign[(int) L.gid[L.wg_key[b2]] * kBalMaxLevels + L.wg_slot[b2]]];
}

__global__ __launch_bounds__(kAlignBlock) void k_cull_estimate(const AlignArgs A, int slice, int32_t* __restrict__ work,
                                                               int32_t* __restrict__ order /* or nullptr: counts only */, const int32_t* __restrict__ place, int n_cu, unsigned int* __restrict__ done_counter) {
  extern __shared__ __align__(16) unsigned char smem[];      // the fixed canvas; the last workgroup's BalanceLds afterwards (the host sizes it for both)
  u64* fcan = reinterpret_cast<u64*>(smem);
  __shared__ Iso s_T;
  __shared__ int s_last;
  const int a = blockIdx.x, tid = threadIdx.x;
  const SliceDev& S = A.s[slice];
  for (int i = tid; i < S.proj.cols; i += kAlignBlock) fcan[i] = kEmptyCell;
  if (tid == 0) { const float p[3] = {A.init_pose[3 * a], A.init_pose[3 * a + 1], A.init_pose[3 * a + 2]}; s_T = slice_iso(S, p); }
  __syncthreads();
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  const int fc = pick_cloud(S.fixed, a), mc = pick_cloud(S.moving, a);
  project_cloud(S.fixed.xy + S.fixed.start[fc], S.fixed.count[fc], ident, S.proj, fcan, tid, kAlignBlock);
  __syncthreads();
  const bool keep = chunk_may_matter(s_T, S.proj, S.moving.lane_bounds[(size_t) mc * kAlignBlock + tid], fcan, S.point_distance, A.cull_est_mt, A.cull_est_mth);
  const int n_keep = __syncthreads_count(keep);
  if (!order) { if (tid == 0) work[a] = n_keep; return; }
  if (tid == 0) {
    // no fences (an agent-scope release writes the XCD's L2 back, a thousand times over): the count goes out as a RETURNING agent-scope exchange -- performed
    // where all XCDs meet once its value is back -- and the ticket's increment depends on that value, so the ticket cannot be taken before the count is there
    // (round 5: the dependence is on the exchange's ARRIVAL, never on what it returned -- work[] is scratch nobody clears, and an increment computed from its stale
    // contents (round 4: `1 + (was == INT_MIN)`, -0.0f of an earlier call's pose is exactly that pattern) could jump the ticket past a workgroup that had not published yet)
    const int was = __hip_atomic_exchange(&work[a], n_keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int inc = 1u;
    asm volatile("" : "+v"(inc) : "v"(was));      // `inc` cannot be formed before `was` is in its register: the ticket waits for the exchange, whatever value came back
    const unsigned int before = __hip_atomic_fetch_add(done_counter, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = before + 1u == gridDim.x;
  }
  __syncthreads();
  if (!s_last) return;
  // every other workgroup has published its count: this one deals the alignments out
  balance_order(*reinterpret_cast<BalanceLds*>(smem), work, (int) gridDim.x, n_cu, 4, order, place, tid, kAlignBlock);
  if (tid == 0) __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next call
}

#ifdef LSM2D_EXPERIMENTS
__global__ __launch_bounds__(kAlignBlock) void k_balance_only(const int32_t* __restrict__ work, int n, int n_cu, int32_t* __restrict__ order, const int32_t* __restrict__ place) {
  extern __shared__ __align__(16) unsigned char smem[];      // BalanceLds
  balance_order(*reinterpret_cast<BalanceLds*>(smem), work, n, n_cu, 4, order, place, threadIdx.x, kAlignBlock);
}
#endif

// ---- the latency kernel: one alignment per workgroup, tuned for calls that cannot fill the chip ------------------
// (one or two projective slices; with two, side by side)
// The live tracker's aligner has two laser slices (front and rear scanner, MULTI.json:396-401) and runs one alignment at a
// time: with one workgroup per alignment the chip is empty and the call is a chain of latencies on ONE compute unit, where sixteen
// waves share four SIMDs -- so the kernel is built around (i) the number of instructions all waves issue per iteration and (ii) the
// length of the stretch only one wave can run (sums -> 3x3 solve -> next transforms).  1024 threads: waves 0-7 own slice 0, waves
// 8-15 slice 1; every thread keeps the COLUMNS it has in k_align (thread = tid mod 512 of its slice: col, col + 512, ...), the
// per-wave sums are the same 64-leaf trees, gathered in the same wave order, the slice totals added in slice order, the prior's
// terms and the solve are the same IEEE operations -- the sums, hence the poses, have k_align's bits (tests: fused == latency kernel).
//   * a moving cloud of <= 1024 points (the tracker's clipped scene: one point per column) lives in LDS, coordinates and normal
//     in one 16-byte row: the bin walk's gather of the moving winner is one LDS read instead of two dependent global loads; its
//     coordinates also sit in registers, ONE point per thread (the few beyond 512 go to the highest lanes), so an iteration's
//     projection is one point's chain per thread with no load in front;
//   * wave totals: the eleven sums go through the DPP tree level by level (independent instructions back to back: no wait
//     states between a VALU write and the DPP read of it), the three counts through ballots and scalar popcounts; counts travel
//     as exact floats so that the gather is one add per word;
//   * the serial stretch runs on wave 0 as a VECTOR: lane q owns quantity q (6 of H, 3 of b, 2 chi, 3 counts; lanes 32-40 the
//     nine entries of the information matrix handed back), gathers it over waves and slices, adds ITS term of the prior -- which
//     every lane computed for itself before the barrier, while the other waves were still summing -- and only the nine inputs of
//     the 3x3 solve are broadcast (v_readlane).  The solve has no early exits (a failed pivot is a flag), the pose's sine and
//     cosine are ready before the barrier, and lanes 0 / 1 turn the new pose into the slices' transforms side by side.
static constexpr int kPairBlock = 2 * kAlignBlock;
static constexpr int kPairMovCap = 2 * kAlignBlock;     // moving points per slice kept on chip
static constexpr int kPairRedStride = 16;               // words per (slice, wave) record: 11 sums, n_in, n_out, n_corr (exact floats), 2 spare = one 64-byte row

// all threads of a slice call; afterwards red[wave][0..13] holds the wave's totals.  count_bits: bits a thread's counts can occupy
// (a thread accumulates at most ceil(cols / 512) pairs).  A slice without robustifier has chi_out == +0 and n_in == n_corr in
// every lane: nothing to add up.
LSM2D_DEV void pair_wave_sums(const Accum& A, float* red, int tid, bool cauchy, int count_bits) {
  const int lane = tid & 63, wave = tid >> 6;
  float f[11] = {A.h00, A.h01, A.h02, A.h11, A.h12, A.h22, A.b0, A.b1, A.b2, A.chi_in, A.chi_out};
  int nc = 0, ni = 0;
  if (cauchy) {
    wave_tree63<11>(f);
    for (int b = 0; b < count_bits; ++b) {
      nc += __builtin_popcountll(__ballot((A.n_corr >> b) & 1)) << b;
      ni += __builtin_popcountll(__ballot((A.n_in >> b) & 1)) << b;
    }
  } else {
    wave_tree63<10>(f);
    for (int b = 0; b < count_bits; ++b) nc += __builtin_popcountll(__ballot((A.n_corr >> b) & 1)) << b;
    ni = nc; f[10] = 0.0f;
  }
  if (lane == 63) {
    float4* r = reinterpret_cast<float4*>(red + wave * kPairRedStride);
    r[0] = make_float4(f[0], f[1], f[2], f[3]); r[1] = make_float4(f[4], f[5], f[6], f[7]);
    r[2] = make_float4(f[8], f[9], f[10], (float) ni);
    *reinterpret_cast<float2*>(r + 3) = make_float2((float) (nc - ni), (float) nc);
  }
}

// ONE entry of the odometry prior's J^T Omega [J | e] (prior_apply's operations for that entry, in its order): row r in 0..2, column c in
// 0..2 of the H term, c == 3 the b term.  J = [[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]].
LSM2D_DEV float prior_term_lane(const PriorDev& Pz, const float pose[3], int r, int c) {
  float E[3]; compose(Pz.cz, Pz.sz, Pz.z_inv, pose, E);
  float cs, sn; sincos_fixed(E[2], sn, cs);
  const float msn = -sn;
  const float x0 = c == 0 ? cs : (c == 1 ? msn : (c == 2 ? 0.0f : E[0]));
  const float x1 = c == 0 ? sn : (c == 1 ? cs : (c == 2 ? 0.0f : E[1]));
  const float x2 = c == 2 ? 1.0f : (c == 3 ? E[2] : 0.0f);
  const float j0 = r == 0 ? cs : (r == 1 ? msn : 0.0f);
  const float j1 = r == 0 ? sn : (r == 1 ? cs : 0.0f);
  const float j2 = r == 2 ? 1.0f : 0.0f;
  float oj[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { float v = 0.0f; v += Pz.omega[3 * k + 0] * x0; v += Pz.omega[3 * k + 1] * x1; v += Pz.omega[3 * k + 2] * x2; oj[k] = v; }
  float v = 0.0f; v += j0 * oj[0]; v += j1 * oj[1]; v += j2 * oj[2];
  return v;
}

// solve_update's LDL^T without early exits: the same operations on the same values whenever it succeeds; a failed pivot or a
// non-finite step is reported at the end (what was computed behind it is discarded by the caller, as solve_update's return does)
LSM2D_DEV bool solve_flat(float h00, float h01, float h02, float h11, float h12, float h22, float b0, float b1, float b2, float damping,
                          float& dx, float& dy, float& dth) {
  const double a00 = (double) h00 + (double) damping, a01 = h01, a02 = h02;
  const double a11 = (double) h11 + (double) damping, a12 = h12, a22 = (double) h22 + (double) damping;
  const double r0 = -(double) b0, r1 = -(double) b1, r2 = -(double) b2;
  const double d0 = a00;
  const double l10 = a01 / d0, l20 = a02 / d0;
  const double d1 = a11 - l10 * a01;
  const double l21 = (a12 - l20 * a01) / d1;
  const double d2 = a22 - l20 * a02 - l21 * l21 * d1;
  const double y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
  const double z2 = y2 / d2;
  const double z1 = y1 / d1 - l21 * z2;
  const double z0 = y0 / d0 - l10 * z1 - l20 * z2;
  dx = (float) z0; dy = (float) z1; dth = (float) z2;
  return (d0 > 0) & (d1 > 0) & (d2 > 0) & (bool) __builtin_isfinite(d0) & (bool) __builtin_isfinite(d1) & (bool) __builtin_isfinite(d2) &
         (bool) __builtin_isfinite(z0) & (bool) __builtin_isfinite(z1) & (bool) __builtin_isfinite(z2);
}

#ifndef LSM2D_PAIR_READ_FIRST
#define LSM2D_PAIR_READ_FIRST false      // z-buffer updates of the on-chip moving cloud: fire-and-forget (neighbouring lanes hold neighbouring columns' points)
#endif

__global__ __launch_bounds__(kPairBlock) void k_align_pair(const AlignArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int nwaves = kAlignBlock / 64;
  float4* fwin = reinterpret_cast<float4*>(smem);                                 // 16-byte rows first (alignment)
  float4* mwin2 = fwin + A.fcan_total;                                             // [n_slices][pair_mov_cap]: the moving clouds, (x, y, nx, ny)
  float4* fall2 = mwin2 + A.n_slices * A.pair_mov_cap;                             // [n_slices][pair_fix_cap]: the fixed clouds
  float* red2 = reinterpret_cast<float*>(fall2 + A.n_slices * A.pair_fix_cap);     // [n_slices][nwaves][kPairRedStride]
  u64* mcan2 = reinterpret_cast<u64*>(red2 + A.n_slices * nwaves * kPairRedStride);      // [n_slices][cols_max]: one moving canvas per slice
  u64* fcan = mcan2 + A.n_slices * A.cols_max;
  __shared__ Iso   s_iso[2];
  __shared__ int   s_done, s_inl;      // s_inl: the iteration about to run belongs to the inlier-only runs (enable_inlier_only_runs)
  __shared__ u64   s_dig;              // the iteration's pair digest, as in k_align
  __shared__ PriorDev s_prior;

  const int a = blockIdx.x, gtid = threadIdx.x, nthr = kAlignBlock * A.n_slices;      // launched with 512 threads per slice (one or two slices)
#ifdef LSM2D_PHASE_CLOCKS      // debug build: where one alignment's time goes (10 ns ticks), printed by thread 0
  unsigned long long pc_t = __builtin_amdgcn_s_memrealtime(), pc_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define LSM2D_PC(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); pc_acc[k] += n_ - pc_t; pc_t = n_; } while (0)
#else
#define LSM2D_PC(k) do { } while (0)
#endif
  const int half = __builtin_amdgcn_readfirstlane(gtid >> 9);      // wave-uniform: the slice this wave works for
  const bool w0 = __builtin_amdgcn_readfirstlane(gtid >> 6) == 0;  // wave 0: the serial stretch
  const int tid = gtid & (kAlignBlock - 1), lane = gtid & 63;
  u64* mcan = mcan2 + half * A.cols_max;
  float* red = red2 + half * nwaves * kPairRedStride;
  // ---- everything that comes from memory is asked for first: the clouds' places and sizes, then this thread's rows of both clouds (the
  //      fixed one possibly still in the host's pinned upload buffer: SliceDev::unpack_src) -- in flight while the canvases are cleared
  const SliceDev& S = A.s[half];
  const int mc = pick_cloud(S.moving, a), fc = pick_cloud(S.fixed, a);
  const bool unpack = A.inline_n1 && S.unpack_src;
  const int mbase = S.moving.start[mc], fbase = S.fixed.start[fc];
  const int m_count = S.moving.count[mc], f_count = unpack ? S.unpack_n : S.fixed.count[fc];
  const float2* mn = S.moving.nrm + mbase; const float2* mp = S.moving.xy + mbase;
  // clouds on chip (see the head comment): the moving one at most two points per thread (coordinates stay in registers), the fixed one as many rows as LDS has
  const bool m_on_chip = !S.moving.lane_xy && m_count <= A.pair_mov_cap;      // pair_mov_cap: kPairMovCap, or 0 when LDS has no room
  // (workgroup-uniform: the branch below holds a barrier.  pair_fix_cap > 0 means the host sized the rows for the LARGEST fixed cloud of every slice, so
  // both halves take the same side; with pair_fix_cap == 0 an empty fixed cloud must not count as "on chip" while the other slice's is not)
  const bool f_on_chip = A.pair_fix_cap > 0 && f_count <= A.pair_fix_cap;
  float4* mwin = mwin2 + half * A.pair_mov_cap;
  float4* fall = fall2 + half * A.pair_fix_cap;
  const int j1 = kPairMovCap - 1 - tid;                   // this thread's second moving point, if the cloud has more than 512
  float2 p0 = make_float2(0.0f, 0.0f), p1 = p0, n0 = p0, n1 = p0;
  if (m_on_chip) {
    if (tid < m_count) { p0 = mp[tid]; n0 = mn[tid]; }
    if (j1 < m_count) { p1 = mp[j1]; n1 = mn[j1]; }
  }
  auto fixed_row = [&](int i) {
    if (unpack) return S.unpack_src[i];
    const float2 p = S.fixed.xy[fbase + i], n = S.fixed.nrm[fbase + i];
    return make_float4(p.x, p.y, n.x, n.y);
  };
  float4 frow = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (f_on_chip && tid < f_count) frow = fixed_row(tid);
  constexpr int kPriorWords = (int) (sizeof(PriorDev) / sizeof(float));
  if (A.prior && gtid >= 64 && gtid < 64 + kPriorWords)
    ((float*) &s_prior)[gtid - 64] = A.inline_n1 ? ((const float*) &A.prior1)[gtid - 64] : ((const float*) (A.prior + a))[gtid - 64];
  if (unpack && !f_on_chip) unpack_fixed_set(S, tid, kAlignBlock);      // visible after the barrier below
  for (int i = gtid; i < A.fcan_total; i += nthr) fcan[i] = kEmptyCell;
  for (int i = gtid; i < A.n_slices * A.cols_max; i += nthr) mcan2[i] = kEmptyCell;

  // wave 0's state: the estimate (the same value in every lane), each lane's quantity and its entry of the prior, lanes 0 / 1 their slice's sensor offset
  float pose[3] = {0.0f, 0.0f, 0.0f}, hl = 0.0f;
  int status = LSM2D_RUNNING, last_n_in = 0;
  float prev_chi = 0.0f;            // total chi^2 of the previous iteration (termination_chi_epsilon)
  int phase = 0, phase_start = 0, phase_end = A.max_it;      // as in k_align
  int q = 15, pr = 0, pc = 0; bool has_pterm = false;
  float kS[3] = {0.0f, 0.0f, 0.0f}, kc = 1.0f, ks = 0.0f; int khs = 0;
  if (w0) {
    if (A.inline_n1) { pose[0] = A.pose1[0]; pose[1] = A.pose1[1]; pose[2] = A.pose1[2]; }
    else { pose[0] = A.init_pose[3 * a + 0]; pose[1] = A.init_pose[3 * a + 1]; pose[2] = A.init_pose[3 * a + 2]; }
    if (lane < 16) {
      q = lane;
      if (lane < 9) { has_pterm = true; pr = (int) ((0x210211000ull >> (4 * lane)) & 15); pc = (int) ((0x333221210ull >> (4 * lane)) & 15); }
    } else if (lane >= 32 && lane < 41) {
      const int k = lane - 32;
      q = (int) ((0x542431210ull >> (4 * k)) & 15); has_pterm = true; pr = k / 3; pc = k - 3 * pr;
    }
    const SliceDev& Sa = A.s[0]; const SliceDev& Sb = A.s[A.n_slices - 1];
    const bool second = lane == 1;
    kS[0] = second ? Sb.Sinv[0] : Sa.Sinv[0]; kS[1] = second ? Sb.Sinv[1] : Sa.Sinv[1]; kS[2] = second ? Sb.Sinv[2] : Sa.Sinv[2];
    kc = second ? Sb.cSinv : Sa.cSinv; ks = second ? Sb.sSinv : Sa.sSinv; khs = second ? Sb.has_sensor : Sa.has_sensor;
    if (lane < A.n_slices) s_iso[lane] = slice_iso_of(khs, kc, ks, kS, pose);
    if (lane == 0) { s_done = 0; s_inl = 0; s_dig = 0ull; }
    if (A.out_last_pose && lane < 3) A.out_last_pose[3 * a + lane] = lane == 0 ? pose[0] : (lane == 1 ? pose[1] : pose[2]);
  }
  __syncthreads();                  // canvases cleared, prior and first transforms in LDS
  PriorDev pz;                      // wave 0's copy of the prior, in registers
  if (w0 && A.prior) pz = s_prior;
  if (m_on_chip) {
    if (tid < m_count) mwin[tid] = make_float4(p0.x, p0.y, n0.x, n0.y);
    if (j1 < m_count) mwin[j1] = make_float4(p1.x, p1.y, n1.x, n1.y);
  }
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  if (f_on_chip) {
    // the fixed cloud: every row into LDS (the bin walk reads the winner's row there: no table of winners, no pass to fill it), a set that
    // was still in the upload buffer also into its arrays (later consumers find them there), and into the z-buffer -- project_cloud's
    // operations per point (project_point with the identity)
    float2* oxy = const_cast<float2*>(S.fixed.xy) + fbase; float2* onr = const_cast<float2*>(S.fixed.nrm) + fbase;
    for (int i = tid; i < f_count; i += kAlignBlock) {
      if (i != tid) frow = fixed_row(i);
      fall[i] = frow;
      if (unpack) { oxy[i] = make_float2(frow.x, frow.y); onr[i] = make_float2(frow.z, frow.w); }
      project_point(ident, S.proj, frow.x, frow.y, i, fcan + S.fcan_offset);
    }
    if (unpack && tid == 0) *const_cast<int32_t*>(S.fixed.count) = S.unpack_n;
  } else {
    project_cloud(S.fixed.xy + fbase, f_count, ident, S.proj, fcan + S.fcan_offset, tid, kAlignBlock);
    __syncthreads();
    for (int col = tid; col < S.proj.cols; col += kAlignBlock) {
      const u64 k = fcan[S.fcan_offset + col];
      if (k != kEmptyCell) {
        const int fi = (int) (uint32_t) k;
        const float2 p = S.fixed.xy[fbase + fi], n = S.fixed.nrm[fbase + fi];
        fwin[S.fcan_offset + col] = make_float4(p.x, p.y, n.x, n.y);
      }
    }
  }
  // what the serial stretch reads from the kernel arguments, fetched once (an s_load and its wait per use otherwise)
  int min_corr0 = A.s[0].min_corr, min_corr1 = A.s[1].min_corr, n_slices = A.n_slices;
  unsigned long long prior_ptr = reinterpret_cast<unsigned long long>(A.prior);
  int term_eps_b = __float_as_int(A.term_eps), damping_b = __float_as_int(A.damping);
  StatsDev* out_stats = A.out_stats ? A.out_stats + (size_t) a * A.stats_stride : nullptr;
  asm volatile("" : "+s"(min_corr0), "+s"(min_corr1), "+s"(n_slices), "+s"(prior_ptr), "+s"(term_eps_b), "+s"(damping_b));
  asm volatile("" : "+v"(out_stats));
  const bool two_slices = n_slices == 2, has_prior = prior_ptr != 0;
  const float term_eps = __int_as_float(term_eps_b), damping = __int_as_float(damping_b);
  // ... and what every wave's projection and walk read, per slice
  const bool on_chip = m_on_chip && f_on_chip;
  ProjK Pk = S.proj;
  int k00_b = __float_as_int(Pk.K00), k01_b = __float_as_int(Pk.K01), r2lo_b = __float_as_int(Pk.r2lo), r2hi_b = __float_as_int(Pk.r2hi), kcols = Pk.cols;
  int pd_b = __float_as_int(S.point_distance), ncos_b = __float_as_int(S.normal_cos), tau_b = __float_as_int(S.tau);
  asm volatile("" : "+s"(k00_b), "+s"(k01_b), "+s"(r2lo_b), "+s"(r2hi_b), "+s"(kcols), "+s"(pd_b), "+s"(ncos_b), "+s"(tau_b));
  Pk.K00 = __int_as_float(k00_b); Pk.K01 = __int_as_float(k01_b); Pk.r2lo = __int_as_float(r2lo_b); Pk.r2hi = __int_as_float(r2hi_b); Pk.cols = kcols;
  const float k_pd = __int_as_float(pd_b), k_ncos = __int_as_float(ncos_b), k_tau = __int_as_float(tau_b);
  const int per_thread = (S.proj.cols + kAlignBlock - 1) / kAlignBlock;      // pairs a thread can accumulate
  const int count_bits = 32 - __builtin_clz(per_thread | 1);
  const bool cauchy = S.cauchy != 0;
  const bool want_dig = A.out_stats != nullptr;
  const uint32_t salt = (uint32_t) half * 0x632BE5ABu;
  const int it_cap = A.inlier_runs ? 2 * A.max_it : A.max_it;
  __syncthreads();
  LSM2D_PC(0);

  const u64* fcs = fcan + S.fcan_offset; const float4* fws = fwin + S.fcan_offset;
  int it = 0;
  for (; it < it_cap; ++it) {
    const Iso T = s_iso[half];
    const bool inl_only = A.inlier_runs && __builtin_amdgcn_readfirstlane(s_inl) != 0;
    Accum acc; accum_zero(acc);
    if (on_chip) {
      // both clouds in LDS (the tracker's case): one point's z-buffer update per thread, then the walk one column at a time -- every gather is
      // an LDS row, so there is no latency worth a second column in flight, and a wave without a second column does not walk through one
      if (tid < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, Pk, p0.x, p0.y, tid, mcan);
      if (j1 < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, Pk, p1.x, p1.y, j1, mcan);
      __syncthreads();
      LSM2D_PC(1);
      for (int col = tid; col < Pk.cols; col += kAlignBlock) {
        const u64 fk = fcs[col], mk = mcan[col];
        mcan[col] = kEmptyCell;
        const uint32_t fdb = (uint32_t) (fk >> 32), mdb = (uint32_t) (mk >> 32);       // an empty cell's depth bits are all ones, no depth's are
        if (fdb != 0xFFFFFFFFu && mdb != 0xFFFFFFFFu && !(__builtin_fabsf(__uint_as_float(fdb) - __uint_as_float(mdb)) > k_pd)) {
          const float4 m = mwin[(uint32_t) mk], f = fall[(uint32_t) fk];
          float nqx, nqy; xf_normal(T, m.z, m.w, nqx, nqy);
          if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < k_ncos)) {
            if (want_dig) digest_add(&s_dig, salt, (int) (uint32_t) fk, (int) (uint32_t) mk);
            accumulate_pair<true>(T, make_float2(f.x, f.y), make_float2(f.z, f.w), make_float2(m.x, m.y), make_float2(m.z, m.w), cauchy, k_tau, acc, inl_only);
          }
        }
      }
    } else {
    if (S.moving.lane_xy) project_cloud_lanes(S.moving.lane_xy + S.moving.lane_start[mc], S.moving.lane_T[mc], T, S.proj, mcan, tid, kAlignBlock);
    else if (m_on_chip) {         // the same points every iteration: no load, no wait
      if (tid < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, S.proj, p0.x, p0.y, tid, mcan);
      if (j1 < m_count) project_point<LSM2D_PAIR_READ_FIRST>(T, S.proj, p1.x, p1.y, j1, mcan);
    }
    else project_cloud(mp, m_count, T, S.proj, mcan, tid, kAlignBlock);
    __syncthreads();
    LSM2D_PC(1);
    // k_align's bin walk, same thread <-> column mapping and order (col, then col + 512, ...), two columns per trip: both
    // columns' gathers of the moving winner are in flight together
    for (int col = tid; col < S.proj.cols; col += 2 * kAlignBlock) {
      const int col1 = col + kAlignBlock;
      const bool in1 = col1 < S.proj.cols;
      const u64 fk0 = fcs[col], mk0 = mcan[col];
      const u64 fk1 = in1 ? fcs[col1] : kEmptyCell, mk1 = in1 ? mcan[col1] : kEmptyCell;
      mcan[col] = kEmptyCell;
      if (in1) mcan[col1] = kEmptyCell;
      auto depth_gate = [&](u64 fk, u64 mk) {
        if (mk == kEmptyCell || fk == kEmptyCell) return false;
        const float fd = __uint_as_float((uint32_t) (fk >> 32)), md = __uint_as_float((uint32_t) (mk >> 32));
        return !(__builtin_fabsf(fd - md) > S.point_distance);
      };
      const bool g0 = depth_gate(fk0, mk0), g1 = depth_gate(fk1, mk1);
      const int mi0 = g0 ? (int) (uint32_t) mk0 : 0, mi1 = g1 ? (int) (uint32_t) mk1 : 0;
      float2 nm0, pm0, nm1, pm1;
      if (m_on_chip) {
        if (g0) { const float4 m = mwin[mi0]; pm0 = make_float2(m.x, m.y); nm0 = make_float2(m.z, m.w); }
        if (g1) { const float4 m = mwin[mi1]; pm1 = make_float2(m.x, m.y); nm1 = make_float2(m.z, m.w); }
      } else {
        if (g0) { nm0 = mn[mi0]; pm0 = mp[mi0]; }
        if (g1) { nm1 = mn[mi1]; pm1 = mp[mi1]; }
      }
      if (g0) {
        const float4 f = f_on_chip ? fall[(uint32_t) fk0] : fws[col];
        float nqx, nqy; xf_normal(T, nm0.x, nm0.y, nqx, nqy);
        if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos)) {
          if (want_dig) digest_add(&s_dig, salt, (int) (uint32_t) fk0, mi0);
          accumulate_pair<true>(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm0, nm0, cauchy, S.tau, acc, inl_only);
        }
      }
      if (g1) {
        const float4 f = f_on_chip ? fall[(uint32_t) fk1] : fws[col1];
        float nqx, nqy; xf_normal(T, nm1.x, nm1.y, nqx, nqy);
        if (!(__builtin_fmaf(nqx, f.z, nqy * f.w) < S.normal_cos)) {
          if (want_dig) digest_add(&s_dig, salt, (int) (uint32_t) fk1, mi1);
          accumulate_pair<true>(T, make_float2(f.x, f.y), make_float2(f.z, f.w), pm1, nm1, cauchy, S.tau, acc, inl_only);
        }
      }
    }
    }
    LSM2D_PC(6);                 // thread 0's wave: bin walk
    pair_wave_sums(acc, red, tid, cauchy, count_bits);
    LSM2D_PC(7);                 // its wave sums
    // what depends on the pose alone, computed by wave 0 HERE, where it would otherwise wait for the slowest of the sixteen: each
    // lane's entry of the prior's terms, and the rotation of the update X <- X * v2t(dx)
    float P = 0.0f, sp = 0.0f, cp = 1.0f;
    if (w0) {
      sincos_fixed(pose[2], sp, cp);
      if (has_prior) P = prior_term_lane(pz, pose, pr, pc);
    }
    LSM2D_PC(9);                 // prior
    __syncthreads();
    LSM2D_PC(2);                 // waiting for the other waves
    if (w0) {
      // gather: this lane's quantity over the waves (wave order, from +0 -- block_reduce_gather's sums), both slices
      float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
      for (int w = 0; w < nwaves; ++w) v0 += red2[w * kPairRedStride + q];
      const bool two = two_slices;
      if (two) {
#pragma unroll
        for (int w = 0; w < nwaves; ++w) v1 += red2[(nwaves + w) * kPairRedStride + q];
      }
      LSM2D_PC(3);
      // k_align's per-slice accumulation (zeroed sums, then slice 0, then slice 1; the pair count of every slice, the rest of active ones)
      const int nc0 = (int) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), 13));
      const int nc1 = (int) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), 13));
      const bool act0 = nc0 > min_corr0, act1 = two && nc1 > min_corr1;
      const bool always = lane == 13;
      float tot = 0.0f;
      tot += (act0 || always) ? v0 : 0.0f;        // (adding +0 to a sum that started from +0 changes nothing)
      tot += (act1 || always) ? v1 : 0.0f;
      last_n_in = (int) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 11));
      if (out_stats && lane >= 9 && lane < 14) {        // StatsDev {n_corr, n_in, n_out, chi_in, chi_out} <- lanes 13, 11, 12, 9, 10
        const int word = lane == 13 ? 0 : (lane == 11 ? 1 : (lane == 12 ? 2 : lane - 6));
        reinterpret_cast<int32_t*>(out_stats + it)[word] = lane < 11 ? __float_as_int(tot) : (int) tot;
      }
      if (out_stats && lane == 0) {      // the iteration's pair digest (every pair's add landed before the barrier above); zeroed for the next iteration
        const u64 dg = s_dig; s_dig = 0ull;
        reinterpret_cast<uint32_t*>(out_stats + it)[5] = (uint32_t) dg; reinterpret_cast<uint32_t*>(out_stats + it)[6] = (uint32_t) (dg >> 32);
      }
      LSM2D_PC(8);               // sums of the slices, statistics
      bool done_now = false;
      if (!(act0 || act1)) { status = LSM2D_NOT_ENOUGH_CORRESPONDENCES; done_now = true; }
      else {
        const float Hq = (has_prior && has_pterm) ? tot + P : tot;
        hl = Hq;
#define LSM2D_RL_F(k) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Hq), k))
        float dx, dy, dth;
        const bool ok = solve_flat(LSM2D_RL_F(0), LSM2D_RL_F(1), LSM2D_RL_F(2), LSM2D_RL_F(3), LSM2D_RL_F(4), LSM2D_RL_F(5),
                                   LSM2D_RL_F(6), LSM2D_RL_F(7), LSM2D_RL_F(8), damping, dx, dy, dth);
#undef LSM2D_RL_F
        if (!ok) { status = LSM2D_SINGULAR_H; done_now = true; }
        else {
          const float nx = __builtin_fmaf(cp, dx, __builtin_fmaf(-sp, dy, pose[0]));
          const float ny = __builtin_fmaf(sp, dx, __builtin_fmaf(cp, dy, pose[1]));
          pose[0] = nx; pose[1] = ny; pose[2] = wrap_angle(pose[2] + dth);
          bool phase_over = it + 1 >= phase_end;
          if (term_eps > 0.0f) {         // as in k_align
            const float chi_now = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 9)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 10));
            if (it > phase_start && __builtin_fabsf(prev_chi - chi_now) < term_eps * chi_now) phase_over = true;
            prev_chi = chi_now;
          }
          if (phase_over) {              // as in k_align: the inlier-only runs follow a regular loop that ended well
            if (A.inlier_runs && phase == 0 && last_n_in >= A.min_inliers) { phase = 1; phase_start = it + 1; phase_end = it + 1 + A.max_it; if (lane == 0) s_inl = 1; }
            else done_now = true;
          }
        }
      }
      LSM2D_PC(10);              // 3x3 solve and pose update
      if (done_now) { if (lane == 0) s_done = 1; }
      else {
        if (lane < A.n_slices) s_iso[lane] = slice_iso_of(khs, kc, ks, kS, pose);      // the next iteration's transforms: one slice per lane
        if (A.out_last_pose && lane < 3) A.out_last_pose[3 * a + lane] = lane == 0 ? pose[0] : (lane == 1 ? pose[1] : pose[2]);
      }
      LSM2D_PC(4);
    }
    __syncthreads();
    LSM2D_PC(5);
    if (s_done) { ++it; break; }
  }
#ifdef LSM2D_PHASE_CLOCKS
  if (gtid == 0 && a == 0) printf("k_align_pair ticks(10ns): prologue %llu project %llu walk %llu wave-sums %llu wait %llu gather %llu solve %llu barrier %llu its %d\n",
                                  pc_acc[0], pc_acc[1], pc_acc[6], pc_acc[7], pc_acc[2], pc_acc[3], pc_acc[4] + pc_acc[8] + pc_acc[9] + pc_acc[10], pc_acc[5], it);
  if (gtid == 0 && a == 0) printf("  solve = sums %llu + prior %llu + ldlt/update %llu + next transforms %llu\n", pc_acc[8], pc_acc[9], pc_acc[10], pc_acc[4]);
#endif
#undef LSM2D_PC
  if (w0) {
    int st = status;
    if (st == LSM2D_RUNNING) st = (A.max_it > 0 && last_n_in < A.min_inliers) ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    if (lane < 3) A.out_pose[3 * a + lane] = lane == 0 ? pose[0] : (lane == 1 ? pose[1] : pose[2]);
    if (A.out_H && lane >= 32 && lane < 41) A.out_H[9 * a + lane - 32] = hl;
    if (A.out_its && lane == 0) A.out_its[a] = it;
    if (A.host_polls) {      // status last (see k_align); the fence covers every lane's stores above
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&A.out_status[a], st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    else if (lane == 0) A.out_status[a] = st;
  }
}

// ---- split path: the same alignment spread over many workgroups -------------------------------------------------
// For a handful of alignments against a big cloud one workgroup per alignment leaves the chip empty, so each iteration
// becomes two launches: k_split_project z-buffers slices of the cloud in LDS and folds them into a global canvas
// (atomicMin u64 is order independent), k_split_finish does the bin walk, the reduction (same thread <-> column mapping,
// same order as k_align, hence bit-identical sums), the 3x3 solve and the pose update.  Projective slices only.
struct SplitArgs {
  AlignArgs A;
  u64* gcan;             // [n_align][2 * fcan_total]: fixed canvases then moving canvases, pre-filled with kEmptyCell
  float* pose;           // [n_align][3] current estimate
  int32_t* done;         // [n_align] 0 = running
  float* H_last;         // [n_align][9]
  StatsDev* last;        // [n_align]
  int32_t* phase;        // [n_align][3]: phase (0 regular, 1 inlier-only runs), its first iteration, its end -- zero-filled means (0, 0, max_it)
  int32_t it;            // iteration this launch belongs to
};


template <bool kFixed>
__global__ __launch_bounds__(512) void k_split_project(const SplitArgs S) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  __shared__ Iso s_T;
  const int a = blockIdx.y, sl = blockIdx.z, tid = threadIdx.x;
  if (S.done[a]) return;
  const SliceDev& SL = S.A.s[sl];
  const CloudDev& C = kFixed ? SL.fixed : SL.moving;
  const int ci = pick_cloud(C, a), n = C.count[ci];
  const int npairs = (n + 1) >> 1;
  const int per = (npairs + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = lo + per < npairs ? lo + per : npairs;
  if (lo >= hi) return;
  if (tid == 0) {
    if (kFixed) { s_T.c = 1.0f; s_T.s = 0.0f; s_T.tx = 0.0f; s_T.ty = 0.0f; }
    else { const float p[3] = {S.pose[3 * a], S.pose[3 * a + 1], S.pose[3 * a + 2]}; s_T = slice_iso(SL, p); }
  }
  const ProjK P = SL.proj;
  for (int i = tid; i < P.cols; i += 512) can[i] = kEmptyCell;
  __syncthreads();
  const Iso T = s_T;
  const float4* xy4 = reinterpret_cast<const float4*>(C.xy + C.start[ci]);
  for (int j = lo + tid; j < hi; j += 512) {
    const float4 v = xy4[j];
    project_point(T, P, v.x, v.y, 2 * j, can);
    if (2 * j + 1 < n) project_point(T, P, v.z, v.w, 2 * j + 1, can);
  }
  __syncthreads();
  u64* g = S.gcan + (size_t) a * 2 * S.A.fcan_total + (kFixed ? 0 : S.A.fcan_total) + SL.fcan_offset;
  for (int i = tid; i < P.cols; i += 512) { const u64 k = can[i]; if (k != kEmptyCell) atomicMin(&g[i], k); }
}

// kSeq: "sum_order" 1 -- the sums pair after pair in ascending column (lsm2d_device.h: pair_terms / seq_walk), as k_align_seq forms them
template <bool kSeq>
__global__ __launch_bounds__(kAlignBlock) void k_split_finish(const SplitArgs S) {
  const AlignArgs& A = S.A;
  __shared__ float red[(kAlignBlock / 64) * kAccumWords];
  __shared__ __attribute__((aligned(16))) float s_rec[kSeq ? kSeqHalf * kSeqFields : 4];
  __shared__ Iso s_iso[kMaxSlices];
  // as in k_align: the iteration's sums are added in LDS by the lanes that gathered them, the matrix is assembled, given its prior and
  // solved where it lies (no private arrays, no scratch on the serial stretch)
  __shared__ float s_H[9], s_rhs[3], s_sum[kAccumWords + 2], s_pose[3];
  __shared__ int s_n_corr, s_active;
  __shared__ u64 s_dig;
  const int a = blockIdx.x, tid = threadIdx.x;
  constexpr int nwaves = kAlignBlock / 64;
  if (S.done[a]) return;
  const bool want_dig = A.out_stats != nullptr;
  const bool inl_only = A.inlier_runs && S.phase[3 * a] != 0;
  if (tid == 0) {
    s_dig = 0ull;
    s_pose[0] = S.pose[3 * a]; s_pose[1] = S.pose[3 * a + 1]; s_pose[2] = S.pose[3 * a + 2];
    if (A.out_last_pose) { A.out_last_pose[3 * a] = s_pose[0]; A.out_last_pose[3 * a + 1] = s_pose[1]; A.out_last_pose[3 * a + 2] = s_pose[2]; }
    for (int s = 0; s < A.n_slices; ++s) s_iso[s] = slice_iso(A.s[s], s_pose);
    for (int k = 0; k < 11; ++k) s_sum[k] = 0.0f;
    s_sum[11] = s_sum[12] = __int_as_float(0);
    s_n_corr = s_active = 0;
  }
  __syncthreads();
  u64* gF = S.gcan + (size_t) a * 2 * A.fcan_total; u64* gM = gF + A.fcan_total;
  for (int s = 0; s < A.n_slices; ++s) {
    const SliceDev& SL = A.s[s];
    const Iso T = s_iso[s];
    const int fc = pick_cloud(SL.fixed, a), mc = pick_cloud(SL.moving, a);
    const int mbase = SL.moving.start[mc], fbase = SL.fixed.start[fc];
    const float2* fn = SL.fixed.nrm + fbase; const float2* mn = SL.moving.nrm + mbase;
    const float2* fp = SL.fixed.xy + fbase;  const float2* mp = SL.moving.xy + mbase;
    Accum acc; accum_zero(acc);
    float seq_acc = 0.0f;
    if constexpr (kSeq) {
      for (int col0 = 0; col0 < SL.proj.cols; col0 += kAlignBlock) {      // trips of kAlignBlock consecutive columns, every thread in every trip (barriers)
        const int col = col0 + tid;
        float t[kSeqFields]; seq_zero(t);
        if (col < SL.proj.cols) {
          const u64 mk = gM[SL.fcan_offset + col];
          gM[SL.fcan_offset + col] = kEmptyCell;
          int fi, mi; float2 nf, nm;
          if (match_bin(gF[SL.fcan_offset + col], mk, SL, T, fn, mn, fi, mi, nf, nm)) {
            if (want_dig) digest_add(&s_dig, (uint32_t) s * 0x632BE5ABu, fi, mi);
            bool inl; pair_terms(T, fp[fi], nf, mp[mi], nm, SL.cauchy != 0, SL.tau, inl_only, t, inl);
            ++acc.n_corr; acc.n_in += inl ? 1 : 0; acc.n_out += inl ? 0 : 1;
          }
        }
        const int n_rec = SL.proj.cols - col0 < kAlignBlock ? SL.proj.cols - col0 : kAlignBlock;
        for (int h0 = 0; h0 < n_rec; h0 += kSeqHalf) {      // the trip's records in two halves
          if (tid >= h0 && tid < h0 + kSeqHalf) seq_store(s_rec, tid - h0, t);
          __syncthreads();
          const int left = n_rec - h0;
          if (tid < 64) seq_acc = seq_walk(s_rec, left < kSeqHalf ? left : kSeqHalf, tid, seq_acc);
          __syncthreads();
        }
      }
    } else
    for (int col = tid; col < SL.proj.cols; col += kAlignBlock) {
      const u64 mk = gM[SL.fcan_offset + col];
      gM[SL.fcan_offset + col] = kEmptyCell;                  // ready for the next iteration's projection
      int fi, mi; float2 nf, nm;
      if (match_bin(gF[SL.fcan_offset + col], mk, SL, T, fn, mn, fi, mi, nf, nm)) {
        if (want_dig) digest_add(&s_dig, (uint32_t) s * 0x632BE5ABu, fi, mi);
        accumulate_pair(T, fp[fi], nf, mp[mi], nm, SL.cauchy != 0, SL.tau, acc, inl_only);
      }
    }
    block_reduce_store(acc, red, tid);
    __syncthreads();
    if (tid < 64) {
      float v; int vi; block_reduce_gather_lane(red, nwaves, tid, v, vi);
      if constexpr (kSeq) v = seq_acc;
      const int n_corr = __builtin_amdgcn_readlane(vi, 13);
      if (tid == 0) s_n_corr += n_corr;
      if (n_corr > SL.min_corr) {
        if (tid < 11) s_sum[tid] += v;
        else if (tid < 13) s_sum[tid] = __int_as_float(__float_as_int(s_sum[tid]) + vi);
        if (tid == 0) ++s_active;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    StatsDev last; last.n_corr = s_n_corr; last.n_in = __float_as_int(s_sum[11]); last.n_out = __float_as_int(s_sum[12]); last.chi_in = s_sum[9]; last.chi_out = s_sum[10];
    if (A.out_stats) { const u64 dg = s_dig; last.dig_lo = (uint32_t) dg; last.dig_hi = (uint32_t) (dg >> 32); A.out_stats[(size_t) a * A.stats_stride + S.it] = last; }
    int status = LSM2D_RUNNING;
    bool stop_now = false;
    const int ph = A.inlier_runs ? S.phase[3 * a] : 0, ph_start = ph ? S.phase[3 * a + 1] : 0, ph_end = ph ? S.phase[3 * a + 2] : A.max_it;
    if (!s_active) {
      status = LSM2D_NOT_ENOUGH_CORRESPONDENCES;
      for (int k = 0; k < 9; ++k) s_H[k] = S.it == 0 ? 0.0f : S.H_last[9 * a + k];      // the information matrix stays the last solved iteration's
    } else {
      s_H[0] = s_sum[0]; s_H[1] = s_sum[1]; s_H[2] = s_sum[2]; s_H[3] = s_sum[1]; s_H[4] = s_sum[3]; s_H[5] = s_sum[4];
      s_H[6] = s_sum[2]; s_H[7] = s_sum[4]; s_H[8] = s_sum[5];
      s_rhs[0] = s_sum[6]; s_rhs[1] = s_sum[7]; s_rhs[2] = s_sum[8];
      if (A.prior) add_prior(A.prior[a], s_pose, s_H, s_rhs);
      for (int k = 0; k < 9; ++k) S.H_last[9 * a + k] = s_H[k];
      if (!solve_update(s_H, s_rhs, A.damping, s_pose)) status = LSM2D_SINGULAR_H;
      else {
        S.pose[3 * a] = s_pose[0]; S.pose[3 * a + 1] = s_pose[1]; S.pose[3 * a + 2] = s_pose[2];
        if (A.term_eps > 0.0f) {       // as in k_align; the previous iteration's statistics wait in S.last
          const float chi_now = last.chi_in + last.chi_out;
          if (S.it > ph_start) { const StatsDev pv = S.last[a]; stop_now = __builtin_fabsf((pv.chi_in + pv.chi_out) - chi_now) < A.term_eps * chi_now; }
          S.last[a] = last;
        }
      }
    }
    bool last_it = S.it + 1 >= ph_end || stop_now;
    if (status == LSM2D_RUNNING && last_it && A.inlier_runs && ph == 0 && last.n_in >= A.min_inliers) {      // as in k_align: on to the inlier-only runs
      S.phase[3 * a] = 1; S.phase[3 * a + 1] = S.it + 1; S.phase[3 * a + 2] = S.it + 1 + A.max_it; last_it = false;
    }
    if (status == LSM2D_RUNNING && last_it) status = last.n_in < A.min_inliers ? LSM2D_NOT_ENOUGH_INLIERS : LSM2D_SUCCESS;
    if (status != LSM2D_RUNNING) {
      S.done[a] = 1;
      A.out_status[a] = status;
      A.out_pose[3 * a] = s_pose[0]; A.out_pose[3 * a + 1] = s_pose[1]; A.out_pose[3 * a + 2] = s_pose[2];
      if (A.out_H) for (int k = 0; k < 9; ++k) A.out_H[9 * a + k] = S.it == 0 && !s_active ? 0.0f : S.H_last[9 * a + k];
      if (A.out_its) A.out_its[a] = S.it + 1;
    }
  }
}

// ---- finder-level: one (fixed, moving, pose) -> pairs in ascending column ------------------------
struct FindArgs {
  CloudDev fixed, moving; int32_t fc, mc;
  ProjK proj; float point_distance, normal_cos;
  Iso T;
  int32_t* out_pairs;  // [cols][2]
  int32_t* out_count;
  const u64* fcan_global; const u64* mcan_global;      // a map-sized cloud's canvas, projected over many workgroups beforehand (k_project_split), or nullptr
  float inl_tau;         // > 0: only pairs whose factor is an inlier under a Cauchy robustifier of this threshold (chi^2 < tau) are emitted -- the aligner's
                         // keep_only_inlier_correspondences (lsm2d_align_batch_pairs); 0: every pair
};

__global__ __launch_bounds__(kFindBlock) void k_find_projective(const FindArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* mcan = reinterpret_cast<u64*>(smem);
  u64* fcan = mcan + A.proj.cols;
  __shared__ int s_wave_tot[kFindBlock / 64];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) { mcan[i] = kEmptyCell; fcan[i] = kEmptyCell; }
  if (tid == 0) s_base = 0;
  __syncthreads();
  const Iso ident = {1.0f, 0.0f, 0.0f, 0.0f};
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  if (A.fcan_global) { for (int i = tid; i < A.proj.cols; i += kFindBlock) fcan[i] = A.fcan_global[i]; }
  else project_cloud(A.fixed.xy + fbase, A.fixed.count[A.fc], ident, A.proj, fcan, tid, kFindBlock);
  if (A.mcan_global) { for (int i = tid; i < A.proj.cols; i += kFindBlock) mcan[i] = A.mcan_global[i]; }
  else project_cloud(A.moving.xy + mbase, A.moving.count[A.mc], A.T, A.proj, mcan, tid, kFindBlock);
  __syncthreads();
  SliceDev S; S.point_distance = A.point_distance; S.normal_cos = A.normal_cos;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < A.proj.cols; c0 += kFindBlock) {
    const int col = c0 + tid;
    int fi = -1, mi = -1; float2 nf, nm; bool ok = false;
    if (col < A.proj.cols) ok = match_bin(fcan[col], mcan[col], S, A.T, A.fixed.nrm + fbase, A.moving.nrm + mbase, fi, mi, nf, nm);
    if (ok && A.inl_tau > 0.0f) ok = pair_chi(A.T, A.fixed.xy[fbase + fi], nf, A.moving.xy[mbase + mi], nm) < A.inl_tau;
    // order-preserving compaction: ballot prefix inside the wave, wave totals through LDS
    const u64 bal = __ballot(ok);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base, total = 0;
    for (int w = 0; w < kFindBlock / 64; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
    if (ok) { A.out_pairs[2 * (before + prefix)] = fi; A.out_pairs[2 * (before + prefix) + 1] = mi; }
    __syncthreads();
    if (tid == 0) s_base += total;
    __syncthreads();
  }
  if (tid == 0) *A.out_count = s_base;
}

// ---- finder-level NN: pairs in ascending moving index (correspondence_finder_kd_tree_2d.cpp:12-27) ------
struct FindNNArgs {
  CloudDev fixed, moving; int32_t fc, mc; int32_t use_distmap; int32_t use_kd;      // at most one of the two set; neither: the exact grid search
  float max_distance, normal_cos; Iso T; int32_t nn_group;
  int32_t* out_pairs; int32_t* out_count;
  int32_t* match; int32_t* block_count;      // k_find_nn_multi: per query the matched fixed index or -1; pairs per workgroup
  float inl_tau;                              // as FindArgs::inl_tau
};

__global__ __launch_bounds__(kFindBlock) void k_find_nn(const FindNNArgs A) {
  __shared__ int s_wave_tot[kFindBlock / 64];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc], n = A.moving.count[A.mc];
  GridMeta g; DistMeta dm;
  const int32_t* cst = nullptr; const int32_t* sidx = nullptr; const float2* sxy = nullptr;
  const KdNode* knd = nullptr;
  if (A.use_distmap) dm = A.fixed.dist.meta[A.fc];
  else if (A.use_kd) { knd = A.fixed.kd.nodes + A.fixed.kd.meta[A.fc].node_base; sxy = A.fixed.kd.leaf_xy + fbase; sidx = A.fixed.kd.leaf_idx + fbase; }
  else {
    g = A.fixed.grid.meta[A.fc]; cst = A.fixed.grid.cell_start + g.cell_base;
    sidx = A.fixed.grid.sorted_idx + fbase; sxy = A.fixed.grid.sorted_xy + fbase;
  }
  const float md2 = A.max_distance * A.max_distance;
  const int group = (A.use_distmap || A.use_kd) ? 1 : A.nn_group, sub = tid & (group - 1);
  const int per_step = kFindBlock / group;
  auto query = [&](float qx, float qy) {
    if (A.use_kd) return kd_query(knd, sxy, sidx, qx, qy, md2);
    return group == kNNGroup ? nn_query<kNNGroup>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub)
                             : nn_query<1>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub);
  };
  for (int j0 = 0; j0 < n; j0 += per_step) {
    const int j = j0 + tid / group;
    int best = -1; bool ok = false;
    if (j < n) {
      const float2 pm = A.moving.xy[mbase + j];
      float qx, qy; xf_point(A.T, pm.x, pm.y, qx, qy);
      best = A.use_distmap ? distmap_lookup(dm, A.fixed.dist.parent, qx, qy) : query(qx, qy);
      if (best >= 0 && sub == 0) {
        const float2 nm = A.moving.nrm[mbase + j], nf = A.fixed.nrm[fbase + best];
        float nqx, nqy; xf_normal(A.T, nm.x, nm.y, nqx, nqy);
        ok = !(__builtin_fmaf(nqx, nf.x, nqy * nf.y) < A.normal_cos);
        if (ok && A.inl_tau > 0.0f) ok = pair_chi(A.T, A.fixed.xy[fbase + best], nf, pm, nm) < A.inl_tau;
      }
    }
    // lanes are in ascending query order (tid / group), so the ballot compaction keeps ascending moving index
    const u64 bal = __ballot(ok);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base, total = 0;
    for (int w = 0; w < kFindBlock / 64; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
    if (ok) { A.out_pairs[2 * (before + prefix)] = best; A.out_pairs[2 * (before + prefix) + 1] = j; }
    __syncthreads();
    if (tid == 0) s_base += total;
    __syncthreads();
  }
  if (tid == 0) *A.out_count = s_base;
}

// The same finder over many workgroups (more queries than one workgroup takes in one trip: a map-sized moving cloud against a scan's
// structure is 98 trips of one workgroup otherwise).  Workgroup b owns the queries [b * per_step, (b + 1) * per_step), ascending.
// Phase 0: search, normal gate, match[j] = fixed index or -1, pairs per workgroup.  Phase 1 (a second launch of the same shape): every
// workgroup adds up the counts in front of it, ranks its own pairs by ballot and writes them -- ascending moving index, as the
// reference emits them (correspondence_finder_kd_tree_2d.cpp:12-27, correspondence_finder_nn_2d.cpp:63-80).
template <int kPhase>
__global__ __launch_bounds__(kFindBlock) void k_find_nn_multi(const FindNNArgs A) {
  __shared__ int s_wave_tot[kFindBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = A.moving.count[A.mc];
  const int group = (A.use_distmap || A.use_kd) ? 1 : A.nn_group, sub = tid & (group - 1);
  const int per_step = kFindBlock / group;
  const int j = blockIdx.x * per_step + tid / group;
  if (kPhase == 0) {
    const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
    int best = -1; bool ok = false;
    if (j < n) {
      const float2 pm = A.moving.xy[mbase + j];
      float qx, qy; xf_point(A.T, pm.x, pm.y, qx, qy);
      if (A.use_distmap) best = distmap_lookup(A.fixed.dist.meta[A.fc], A.fixed.dist.parent, qx, qy);
      else if (A.use_kd) {
        best = kd_query(A.fixed.kd.nodes + A.fixed.kd.meta[A.fc].node_base, A.fixed.kd.leaf_xy + fbase, A.fixed.kd.leaf_idx + fbase, qx, qy, A.max_distance * A.max_distance);
      } else {
        const GridMeta g = A.fixed.grid.meta[A.fc];
        const int32_t* cst = A.fixed.grid.cell_start + g.cell_base; const int32_t* sidx = A.fixed.grid.sorted_idx + fbase; const float2* sxy = A.fixed.grid.sorted_xy + fbase;
        const float md2 = A.max_distance * A.max_distance;
        best = group == kNNGroup ? nn_query<kNNGroup>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub) : nn_query<1>(g, cst, sidx, sxy, qx, qy, A.max_distance, md2, sub);
      }
      if (best >= 0 && sub == 0) {
        const float2 nm = A.moving.nrm[mbase + j], nf = A.fixed.nrm[fbase + best];
        float nqx, nqy; xf_normal(A.T, nm.x, nm.y, nqx, nqy);
        ok = !(__builtin_fmaf(nqx, nf.x, nqy * nf.y) < A.normal_cos);
        if (ok && A.inl_tau > 0.0f) ok = pair_chi(A.T, A.fixed.xy[fbase + best], nf, pm, nm) < A.inl_tau;
      }
      if (sub == 0) A.match[j] = ok ? best : -1;
    }
    const u64 bal = __ballot(ok);
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < kFindBlock / 64; ++w) t += s_wave_tot[w]; A.block_count[blockIdx.x] = t; }
  } else {
    __shared__ int s_before;
    if (tid == 0) s_before = 0;
    __syncthreads();
    int mine = 0;
    for (int b = tid; b < (int) blockIdx.x; b += kFindBlock) mine += A.block_count[b];
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0 && mine) atomicAdd(&s_before, mine);
    const int best = (j < n && sub == 0) ? A.match[j] : -1;
    const bool ok = best >= 0;
    const u64 bal = __ballot(ok);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = s_before, total = 0;
    for (int w = 0; w < kFindBlock / 64; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
    if (ok) { A.out_pairs[2 * (before + prefix)] = best; A.out_pairs[2 * (before + prefix) + 1] = j; }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *A.out_count = s_before + total;
  }
}

// ---- projector-level: canvas of one cloud --------------------------------------------------------
struct ProjectArgs {
  CloudDev cloud; int32_t ci; ProjK proj; Iso T;
  int32_t* out_src; float* out_depth; float4* out_xynn;
};

__global__ __launch_bounds__(kFindBlock) void k_project_canvas(const ProjectArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) can[i] = kEmptyCell;
  __syncthreads();
  const int base = A.cloud.start[A.ci];
  project_cloud(A.cloud.xy + base, A.cloud.count[A.ci], A.T, A.proj, can, tid, kFindBlock);
  __syncthreads();
  for (int col = tid; col < A.proj.cols; col += kFindBlock) {
    const u64 k = can[col];
    int src = -1; float depth = 3.402823466e+38f; float4 t = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k != kEmptyCell) {
      src = (int) (uint32_t) k; depth = __uint_as_float((uint32_t) (k >> 32));
      const float2 p = A.cloud.xy[base + src], n = A.cloud.nrm[base + src];
      xf_point(A.T, p.x, p.y, t.x, t.y);
      xf_normal(A.T, n.x, n.y, t.z, t.w);
    }
    if (A.out_src) A.out_src[col] = src;
    if (A.out_depth) A.out_depth[col] = depth;
    if (A.out_xynn) A.out_xynn[col] = t;
  }
}

// ---- factor-level: H, b, stats for a given correspondence vector ----------------------------------
struct LinArgs {
  CloudDev fixed, moving; int32_t fc, mc;
  const int32_t* pairs; int32_t n_pairs;
  Iso T; int32_t cauchy; float tau;
  float* partial;     // [n_blocks][kAccumWords]
  float* out;         // [kAccumWords]
  unsigned long long* dig;      // the pairs' digest (lsm2d_iteration_stats.pair_digest, slice 0), zeroed by the host: every workgroup adds its share
};

__global__ __launch_bounds__(256) void k_linearize_partial(const LinArgs A) {
  __shared__ float red[4 * kAccumWords];
  __shared__ u64 s_dig;
  const int tid = threadIdx.x;
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  Accum acc; accum_zero(acc);
  if (tid == 0) s_dig = 0ull;
  __syncthreads();
  u64 dg = 0ull;
  for (int k = blockIdx.x * 256 + tid; k < A.n_pairs; k += gridDim.x * 256) {
    const int fi = A.pairs[2 * k], mi = A.pairs[2 * k + 1];
    dg += pair_hash_dev(0u, (uint32_t) fi, (uint32_t) mi);
    accumulate_pair(A.T, A.fixed.xy[fbase + fi], A.fixed.nrm[fbase + fi], A.moving.xy[mbase + mi], A.moving.nrm[mbase + mi],
                    A.cauchy != 0, A.tau, acc);
  }
  if (dg) atomicAdd(reinterpret_cast<unsigned long long*>(&s_dig), (unsigned long long) dg);
  block_reduce_store(acc, red, tid);
  __syncthreads();
  if (tid == 0) {
    if (A.dig && s_dig) atomicAdd(A.dig, (unsigned long long) s_dig);
    Accum t; block_reduce_gather(red, 4, t);
    float* p = A.partial + (size_t) blockIdx.x * kAccumWords;
    p[0] = t.h00; p[1] = t.h01; p[2] = t.h02; p[3] = t.h11; p[4] = t.h12; p[5] = t.h22; p[6] = t.b0; p[7] = t.b1; p[8] = t.b2;
    p[9] = t.chi_in; p[10] = t.chi_out; p[11] = __int_as_float(t.n_in); p[12] = __int_as_float(t.n_out); p[13] = __int_as_float(t.n_corr);
  }
}

// "sum_order" 1: the same factor with the sums formed pair after pair in the order of the correspondence vector (the reference's loop): ONE workgroup,
// trips of kAlignBlock consecutive pairs, their terms as records in LDS, eleven lanes of wave 0 adding them in ascending position (lsm2d_device.h)
__global__ __launch_bounds__(kAlignBlock) void k_linearize_seq(const LinArgs A) {
  __shared__ __attribute__((aligned(16))) float s_rec[kSeqHalf * kSeqFields];
  __shared__ float red[(kAlignBlock / 64) * kAccumWords];
  __shared__ u64 s_dig;
  const int tid = threadIdx.x;
  const int fbase = A.fixed.start[A.fc], mbase = A.moving.start[A.mc];
  Accum acc; accum_zero(acc);
  float seq_acc = 0.0f;
  if (tid == 0) s_dig = 0ull;
  __syncthreads();
  u64 dg = 0ull;
  for (int k0 = 0; k0 < A.n_pairs; k0 += kAlignBlock) {
    const int k = k0 + tid;
    float t[kSeqFields]; seq_zero(t);
    if (k < A.n_pairs) {
      const int fi = A.pairs[2 * k], mi = A.pairs[2 * k + 1];
      dg += pair_hash_dev(0u, (uint32_t) fi, (uint32_t) mi);
      bool inl; pair_terms(A.T, A.fixed.xy[fbase + fi], A.fixed.nrm[fbase + fi], A.moving.xy[mbase + mi], A.moving.nrm[mbase + mi], A.cauchy != 0, A.tau, false, t, inl);
      ++acc.n_corr; acc.n_in += inl ? 1 : 0; acc.n_out += inl ? 0 : 1;
    }
    const int n_rec = A.n_pairs - k0 < kAlignBlock ? A.n_pairs - k0 : kAlignBlock;
    for (int h0 = 0; h0 < n_rec; h0 += kSeqHalf) {
      if (tid >= h0 && tid < h0 + kSeqHalf) seq_store(s_rec, tid - h0, t);
      __syncthreads();
      const int left = n_rec - h0;
      if (tid < 64) seq_acc = seq_walk(s_rec, left < kSeqHalf ? left : kSeqHalf, tid, seq_acc);
      __syncthreads();
    }
  }
  if (dg) atomicAdd(reinterpret_cast<unsigned long long*>(&s_dig), (unsigned long long) dg);
  block_reduce_store(acc, red, tid);
  __syncthreads();
  if (tid < 64) {
    float v; int vi; block_reduce_gather_lane(red, kAlignBlock / 64, tid, v, vi);
    if (tid < 11) A.out[tid] = seq_acc;
    else if (tid < kAccumWords) A.out[tid] = __int_as_float(vi);
    if (tid == 0 && A.dig) *A.dig = (unsigned long long) s_dig;
  }
}

__global__ void k_linearize_final(const float* partial, int n_blocks, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Accum t; block_reduce_gather(partial, n_blocks, t);     // fixed block order => deterministic
  out[0] = t.h00; out[1] = t.h01; out[2] = t.h02; out[3] = t.h11; out[4] = t.h12; out[5] = t.h22; out[6] = t.b0; out[7] = t.b1; out[8] = t.b2;
  out[9] = t.chi_in; out[10] = t.chi_out; out[11] = __int_as_float(t.n_in); out[12] = __int_as_float(t.n_out); out[13] = __int_as_float(t.n_corr);
}

// ---- mapping kernels around the aligner: the same polar z-buffer, spread over many
//      workgroups for one big cloud, then an O(Bins) pass.  -----------------------------------------------
struct ProjectSplitArgs {
  const float2* xy; int32_t n; Iso T; ProjK proj;
  u64* gcanvas;            // [cols], pre-filled with kEmptyCell
};

// each workgroup z-buffers a contiguous slice of the cloud in LDS, then folds its canvas into the global one
__global__ __launch_bounds__(512) void k_project_split(const ProjectSplitArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += 512) can[i] = kEmptyCell;
  __syncthreads();
  const Iso T = A.T; const ProjK P = A.proj;
  const int npairs = (A.n + 1) >> 1;
  const int per = (npairs + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = lo + per < npairs ? lo + per : npairs;
  const float4* xy4 = reinterpret_cast<const float4*>(A.xy);
  for (int j = lo + tid; j < hi; j += 512) {
    const float4 v = xy4[j];
    project_point(T, P, v.x, v.y, 2 * j, can);
    if (2 * j + 1 < A.n) project_point(T, P, v.z, v.w, 2 * j + 1, can);
  }
  __syncthreads();
  for (int i = tid; i < P.cols; i += 512) { const u64 k = can[i]; if (k != kEmptyCell) atomicMin(&A.gcanvas[i], k); }
}

LSM2D_DEV int block_compact_offset(bool flag, int* s_wave_tot, int* s_base, int tid, int nwaves) {
  // order-preserving position of this thread's element among the flagged ones (all threads must call)
  const int lane = tid & 63, wave = tid >> 6;
  const u64 bal = __ballot(flag);
  const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int before = *s_base, total = 0;
  for (int w = 0; w < nwaves; ++w) { const int t = s_wave_tot[w]; if (w < wave) before += t; total += t; }
  __syncthreads();
  if (tid == 0) *s_base += total;
  __syncthreads();
  return before + prefix;
}

// the same with ONE barrier per call: the per-wave totals alternate between two buffers (`parity`: 0, 1, 0, ... from call to call; the
// barrier of call i + 1 separates the reads of call i from the writes of call i + 2), and every thread keeps the running base itself
// (`base`, the same value in all threads; in: flagged elements so far, out: including this call's)
LSM2D_DEV int block_compact_pos(bool flag, int* s_tot /* [2][nwaves] */, int parity, int& base, int tid, int nwaves) {
  const int lane = tid & 63, wave = tid >> 6;
  const u64 bal = __ballot(flag);
  const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
  int* t = s_tot + parity * nwaves;
  if (lane == 0) t[wave] = __popcll(bal);
  __syncthreads();
  int before = base, total = 0;
  for (int w = 0; w < nwaves; ++w) { const int v = t[w]; if (w < wave) before += v; total += v; }
  // the running base is the same in every lane: say so (a count that came out of LDS reads is a per-lane value to the compiler, and loops
  // bounded by it compile to per-lane forms -- the preprocessor's window walks ran 58 % slower over a batch before this line)
  base = __builtin_amdgcn_readfirstlane(base + total);
  return before + prefix;
}

// SceneClipperProjective2D::compute tail (mapping/scene_clipper_projective_2d.cpp:53-63): filled cells in ascending
// column -> transformed point (sensor frame), then moved to the robot frame by sensor_in_robot
struct ClipEmitArgs {
  const u64* gcanvas; int32_t cols;
  const float2* xy; const float2* nrm;       // full scene
  Iso T;                                      // sensor_in_local_map^-1
  Iso S; int32_t s_identity;                  // sensor_in_robot
  float2* out_xy; float2* out_nrm; int32_t* out_src; int32_t* out_count_dev /* count[0] of the clipped set */; int32_t* out_count;
  int32_t host_polls;                          // out_src / out_count are pinned host memory and the host polls out_count: write it last, released to the system
};

__global__ __launch_bounds__(kFindBlock) void k_clip_emit(const ClipEmitArgs A);

// small scenes (the tracker's local map between key frames): clipper and merger as ONE workgroup-resident kernel each --
// z-buffers in LDS, no global canvas, no memsets, one launch instead of three resp. six
// n_dev: when non-null the scene's size is only known on the device (its set was last written by an asynchronous clip / merge)
struct ClipSmallArgs { const float2* xy; const float2* nrm; int32_t n; const int32_t* n_dev; ProjK proj; ClipEmitArgs emit; };

LSM2D_DEV void clip_emit_body(const ClipEmitArgs& A, const u64* canvas, int* s_tot /* [2][kFindBlock / 64] */, int tid) {
  int base = 0, parity = 0;
  for (int c0 = 0; c0 < A.cols; c0 += kFindBlock, parity ^= 1) {
    const int col = c0 + tid;
    const u64 k = col < A.cols ? canvas[col] : kEmptyCell;
    const bool ok = k != kEmptyCell;
    const int pos = block_compact_pos(ok, s_tot, parity, base, tid, kFindBlock / 64);
    if (ok) {
      const int src = (int) (uint32_t) k;
      const float2 p = A.xy[src], n = A.nrm[src];
      float x, y, nx, ny;
      xf_point(A.T, p.x, p.y, x, y); xf_normal(A.T, n.x, n.y, nx, ny);
      if (!A.s_identity) {
        float tx, ty, tnx, tny;
        xf_point(A.S, x, y, tx, ty); xf_normal(A.S, nx, ny, tnx, tny);
        x = tx; y = ty; nx = tnx; ny = tny;
      }
      A.out_xy[pos] = make_float2(x, y); A.out_nrm[pos] = make_float2(nx, ny);
      if (A.out_src) A.out_src[pos] = src;
    }
  }
  if (A.host_polls) {      // the count goes last, behind every thread's system-scope release of its rows: the synchronous form's host side polls it
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { *A.out_count_dev = base; __hip_atomic_store(A.out_count, base, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  } else if (tid == 0) { *A.out_count = base; *A.out_count_dev = base; }
}

__global__ __launch_bounds__(kFindBlock) void k_clip_small(const ClipSmallArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* can = reinterpret_cast<u64*>(smem);
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) can[i] = kEmptyCell;
  __syncthreads();
  project_cloud(A.xy, A.n_dev ? *A.n_dev : A.n, A.emit.T, A.proj, can, tid, kFindBlock);
  __syncthreads();
  clip_emit_body(A.emit, can, s_tot, tid);
}

__global__ __launch_bounds__(kFindBlock) void k_clip_emit(const ClipEmitArgs A) {
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  clip_emit_body(A, A.gcanvas, s_tot, threadIdx.x);
}

// transform a cloud (measurement -> scene frame, mapping/merger_projective_2d.cpp:22-23)
__global__ void k_transform_cloud(const float2* __restrict__ xy, const float2* __restrict__ nrm, int n, const Iso T,
                                  float2* __restrict__ oxy, float2* __restrict__ onrm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float2 p = xy[i], q = nrm[i];
    float x, y, nx, ny;
    xf_point(T, p.x, p.y, x, y); xf_normal(T, q.x, q.y, nx, ny);
    oxy[i] = make_float2(x, y); onrm[i] = make_float2(nx, ny);
  }
}

// MergerProjective2D::compute column walk (mapping/merger_projective_2d.cpp:39-95)
struct MergeArgs {
  const u64* scanvas; const u64* mcanvas; int32_t cols;
  float2* sxy; float2* snrm; int32_t n_scene;        // scene, updated in place and appended to
  const float2* mxy; const float2* mnrm;             // measurement already in the scene frame
  float far_limit, merge_threshold;
  int32_t* out;                                       // [4]: new size, new, merged, replaced
  int32_t* count_dev;                                 // count[0] of the scene set
  int32_t host_polls;                                 // out is pinned host memory and the host polls out[0]: write it last, released to the system
};

// mkT: when non-null the measurement is still in its own frame and is moved by *mkT on the fly (fused small-scene kernel);
// the transform is the same operation sequence as k_transform_cloud, so both forms give the same bits
// returns the number of appended points (the same value in every thread); s_tot: [2][kFindBlock / 64]; s_cnt: [1] new, [2] merged, [3] replaced,
// zeroed by the caller, read here by thread 0 behind a barrier of its own
LSM2D_DEV int merge_apply_body(const MergeArgs& A, const u64* scanvas, const u64* mcanvas, const Iso* mkT, int* s_tot, int* s_cnt, int tid) {
  int appended = 0, parity = 0;
  for (int c0 = 0; c0 < A.cols; c0 += kFindBlock, parity ^= 1) {
    const int col = c0 + tid;
    bool append = false; float2 mp = make_float2(0.f, 0.f), mn = mp;
    if (col < A.cols) {
      const u64 mk = mcanvas[col], sk = scanvas[col];
      const float md = __uint_as_float((uint32_t) (mk >> 32));
      if (mk != kEmptyCell && !(md > A.far_limit)) {
        const int mi = (int) (uint32_t) mk;
        mp = A.mxy[mi]; mn = A.mnrm[mi];
        if (mkT) {
          float x, y, nx, ny;
          xf_point(*mkT, mp.x, mp.y, x, y); xf_normal(*mkT, mn.x, mn.y, nx, ny);
          mp = make_float2(x, y); mn = make_float2(nx, ny);
        }
        if (sk == kEmptyCell) { append = true; atomicAdd(&s_cnt[1], 1); }
        else {
          const int si = (int) (uint32_t) sk;
          const float dr = md - __uint_as_float((uint32_t) (sk >> 32));
          if (__builtin_fabsf(dr) < A.merge_threshold) {
            const float2 sp = A.sxy[si], sn = A.snrm[si];
            const float x = (sp.x + mp.x) * 0.5f, y = (sp.y + mp.y) * 0.5f;
            float nx = (sn.x + mn.x) * 0.5f, ny = (sn.y + mn.y) * 0.5f;
            const float nn = __builtin_sqrtf(__builtin_fmaf(nx, nx, ny * ny));
            if (nn > 0.0f) { nx = nx / nn; ny = ny / nn; }
            A.sxy[si] = make_float2(x, y); A.snrm[si] = make_float2(nx, ny);
            atomicAdd(&s_cnt[2], 1);
          } else if (dr > 0.0f) { A.sxy[si] = mp; A.snrm[si] = mn; atomicAdd(&s_cnt[3], 1); }
          else append = true;
        }
      }
    }
    const int pos = block_compact_pos(append, s_tot, parity, appended, tid, kFindBlock / 64);
    if (append) { A.sxy[A.n_scene + pos] = mp; A.snrm[A.n_scene + pos] = mn; }
  }
  if (A.host_polls) __threadfence_system();      // every thread's rows, ahead of the size the host polls
  __syncthreads();                                // the counters are final (and, with host_polls, every thread's rows are released)
  if (tid == 0) {      // the new size goes last; the synchronous form's host side polls it in pinned memory: released to the system then
    A.out[1] = s_cnt[1]; A.out[2] = s_cnt[2]; A.out[3] = s_cnt[3]; *A.count_dev = A.n_scene + appended;
    if (A.host_polls) { __threadfence_system(); __hip_atomic_store(&A.out[0], A.n_scene + appended, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    else A.out[0] = A.n_scene + appended;
  }
  return appended;
}

__global__ __launch_bounds__(kFindBlock) void k_merge_apply(const MergeArgs A) {
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  __shared__ int s_cnt[4];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  merge_apply_body(A, A.scanvas, A.mcanvas, nullptr, s_tot, s_cnt, threadIdx.x);
}

// small scene: transform + both projections + column walk in one workgroup (mxy / mnrm hold the measurement in ITS frame)
struct MergeSmallArgs { MergeArgs m; ProjK proj; Iso Tinv, M; int32_t n_meas; const int32_t* n_scene_dev; const int32_t* n_meas_dev; };   // *_dev: see ClipSmallArgs

__global__ __launch_bounds__(kFindBlock) void k_merge_small(const MergeSmallArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* scan = reinterpret_cast<u64*>(smem);
  u64* mcan = scan + A.proj.cols;
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < A.proj.cols; i += kFindBlock) { scan[i] = kEmptyCell; mcan[i] = kEmptyCell; }
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  MergeArgs m = A.m;
  if (A.n_scene_dev) m.n_scene = *A.n_scene_dev;
  const int n_meas = A.n_meas_dev ? *A.n_meas_dev : A.n_meas;
  project_cloud(m.sxy, m.n_scene, A.Tinv, A.proj, scan, tid, kFindBlock);
  for (int i = tid; i < n_meas; i += kFindBlock) {               // measurement -> scene frame -> camera frame
    const float2 p = m.mxy[i];
    float x, y; xf_point(A.M, p.x, p.y, x, y);
    project_point(A.Tinv, A.proj, x, y, i, mcan);
  }
  __syncthreads();
  merge_apply_body(m, scan, mcan, &A.M, s_tot, s_cnt, tid);
}

// several measurements merged into the scene one after the other by ONE launch (lsm2d_merge_scenes: the live tracker's front and
// rear scan): the same passes as k_merge_small per measurement, the scene's new size carried from one to the next in the workgroup
static constexpr int kMergeMulti = 4;
struct MergeMultiArgs { MergeSmallArgs a[kMergeMulti]; int32_t n; };
__global__ __launch_bounds__(kFindBlock) void k_merge_multi(const MergeMultiArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* scan = reinterpret_cast<u64*>(smem);
  u64* mcan = scan + A.a[0].proj.cols;                             // one projector for all of them
  __shared__ int s_tot[2 * (kFindBlock / 64)];
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x;
  // sizes only the device knows: all of them up front, the loads in flight together (not one round trip per measurement)
  int n_scene = A.a[0].n_scene_dev ? *A.a[0].n_scene_dev : A.a[0].m.n_scene;
  int n_meas_of[kMergeMulti];
#pragma unroll
  for (int k = 0; k < kMergeMulti; ++k) n_meas_of[k] = k < A.n ? (A.a[k].n_meas_dev ? *A.a[k].n_meas_dev : A.a[k].n_meas) : 0;
  for (int k = 0; k < A.n; ++k) {
    const MergeSmallArgs& S = A.a[k];
    for (int i = tid; i < S.proj.cols; i += kFindBlock) { scan[i] = kEmptyCell; mcan[i] = kEmptyCell; }
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    MergeArgs m = S.m;
    m.n_scene = n_scene;
    const int n_meas = k == 0 ? n_meas_of[0] : (k == 1 ? n_meas_of[1] : (k == 2 ? n_meas_of[2] : n_meas_of[3]));
    project_cloud(m.sxy, m.n_scene, S.Tinv, S.proj, scan, tid, kFindBlock);
    for (int i = tid; i < n_meas; i += kFindBlock) {
      const float2 p = m.mxy[i];
      float x, y; xf_point(S.M, p.x, p.y, x, y);
      project_point(S.Tinv, S.proj, x, y, i, mcan);
    }
    __syncthreads();
    n_scene = m.n_scene + merge_apply_body(m, scan, mcan, &S.M, s_tot, s_cnt, tid);      // what the next measurement is merged into
    __syncthreads();                         // thread 0 has read the counters; rows and canvases are free for the next measurement
  }
}

// split a single device cloud back into AoS (download)
__global__ void k_pack_aos(const float2* __restrict__ xy, const float2* __restrict__ nrm, int n, float4* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float2 p = xy[i], q = nrm[i];
    out[i] = make_float4(p.x, p.y, q.x, q.y);
  }
}

// ---- PointCloud::voxelize as the reference uses it (sensor_processing/raw_data_preprocessor_projective_2d.cpp:38-41 with
// coefficients (res, res, 1, 1); mapping/scene_clipper_projective_2d.cpp:44-48 with (res, res, 0.1, 0.1); assumption F2.3,
// PARITY.md section 3): k points staged in LDS (s_q coordinates, s_n normals), key = floor of (x, y) * inv_rx and of the normal
// components * inv_rn, equal keys averaged (normal re-normalised), voxels in ascending lexicographic key order.  One workgroup of
// kVoxBlock threads, k <= kVoxMax; emit(pos, x, y, nx, ny) is called once per voxel; returns the number of voxels (every thread).
static constexpr int kVoxBlock = 1024;
static constexpr int kVoxMax = 2048;
template <int kBlock = kVoxBlock, typename Emit>
LSM2D_DEV int voxelize_lds(const float2* s_q, const float2* s_n, u64* s_key, int k, float inv_rx, float inv_rn, int* s_tot /* [2][kBlock / 64] */,
                           int tid, Emit emit) {
  int np2 = 1; while (np2 < k) np2 <<= 1;
  for (int i = tid; i < np2; i += kBlock) {
    u64 key = ~0ull;
    if (i < k) {
      const float kx = __builtin_floorf(s_q[i].x * inv_rx), ky = __builtin_floorf(s_q[i].y * inv_rx);
      const float knx = __builtin_floorf(s_n[i].x * inv_rn), kny = __builtin_floorf(s_n[i].y * inv_rn);
      if (kx >= -32768.0f && kx < 32768.0f && ky >= -32768.0f && ky < 32768.0f && knx >= -16.0f && knx <= 15.0f && kny >= -16.0f && kny <= 15.0f) {
        const u64 v = ((u64) ((int) kx + 32768) << 26) | ((u64) ((int) ky + 32768) << 10) | ((u64) ((int) knx + 16) << 5) | (u64) ((int) kny + 16);
        key = (v << 16) | (u64) i;
      }
    }
    s_key[i] = key;
  }
  __syncthreads();
  // bitonic network, one compare-exchange per thread and step (np2 / 2 <= kVoxBlock).  Pair t touches elements inside the
  // aligned 128-element block of its wave whenever stride <= 64, so those steps need no workgroup barrier -- LDS operations
  // of one wave complete in order -- only the compiler must keep them in order (wavefront fence).  6 of the 55 steps of a
  // 1024-key sort cross waves.
  // (a smaller workgroup -- the batch preprocessor's 512 threads -- takes its pairs t = tid, tid + kBlock, ...: pair t of a wave still lies in ONE aligned
  // 128-element block, the same one in every step, so the wave-local ordering holds per trip)
  static_assert(kVoxMax / 2 <= kVoxBlock && kBlock % 64 == 0, "one compare-exchange per thread at the full block size");
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma nounroll
      for (int t = tid; t < (np2 >> 1); t += kBlock) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 a = s_key[lo], b = s_key[hi];
        if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
      }
      // the next step's stride is stride / 2, or `size` when this was the last step of its stage
      if (stride > 64 || (stride == 1 && size > 64)) __syncthreads();
      else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
  }
  __syncthreads();
  int nv = 0, parity = 0;
  for (int t0 = 0; t0 < np2; t0 += kBlock, parity ^= 1) {
    const int t = t0 + tid;
    bool head = false; u64 key = ~0ull;
    if (t < np2) { key = s_key[t]; head = key != ~0ull && (t == 0 || (s_key[t - 1] >> 16) != (key >> 16)); }
    const int pos = block_compact_pos(head, s_tot, parity, nv, tid, kBlock / 64);
    if (head) {
      float ax = 0.0f, ay = 0.0f, anx = 0.0f, any_ = 0.0f; int cnt = 0;
      for (int e = t; e < np2 && (s_key[e] >> 16) == (key >> 16); ++e) {
        const int i = (int) (s_key[e] & 0xFFFFull);
        ax += s_q[i].x; ay += s_q[i].y; anx += s_n[i].x; any_ += s_n[i].y; ++cnt;
      }
      const float inv = 1.0f / (float) cnt;
      ax *= inv; ay *= inv; anx *= inv; any_ *= inv;
      const float nn = __builtin_sqrtf(__builtin_fmaf(anx, anx, any_ * any_));
      if (nn > 0.0f) { anx = anx / nn; any_ = any_ / nn; }
      emit(pos, ax, ay, anx, any_);
    }
  }
  return nv;
}

// the clipper's voxelize_resolution > 0 branch (mapping/scene_clipper_projective_2d.cpp:36-48,60-62): the clipped cloud -- written
// by the clip kernels in the SENSOR frame, ascending column -- is voxelised with coefficients (res, res, 0.1, 0.1) and only then
// moved to the robot frame by sensor_in_robot.  In place: everything is staged in LDS first and a voxelised cloud never grows.
struct VoxArgs {
  float2* xy; float2* nrm; int32_t* count_dev;      // the clipped set (one cloud)
  float inv_rx, inv_rn; Iso S; int32_t s_identity;
  int32_t* out_count; int32_t host_polls;            // the synchronous form's count, in pinned memory, written last
};
__global__ __launch_bounds__(kVoxBlock) void k_voxelize_clipped(const VoxArgs A) {
  __shared__ float2 s_q[kVoxMax];
  __shared__ float2 s_n[kVoxMax];
  __shared__ u64 s_key[kVoxMax];
  __shared__ int s_tot[2 * (kVoxBlock / 64)];
  const int tid = threadIdx.x;
  int k = *A.count_dev; if (k > kVoxMax) k = kVoxMax;      // the host refuses canvases beyond kVoxMax columns
  for (int i = tid; i < k; i += kVoxBlock) { s_q[i] = A.xy[i]; s_n[i] = A.nrm[i]; }
  __syncthreads();
  const int nv = voxelize_lds(s_q, s_n, s_key, k, A.inv_rx, A.inv_rn, s_tot, tid, [&](int pos, float x, float y, float nx, float ny) {
    if (!A.s_identity) {
      float tx, ty, tnx, tny;
      xf_point(A.S, x, y, tx, ty); xf_normal(A.S, nx, ny, tnx, tny);
      x = tx; y = ty; nx = tnx; ny = tny;
    }
    A.xy[pos] = make_float2(x, y); A.nrm[pos] = make_float2(nx, ny);
  });
  if (A.host_polls) {
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { *A.count_dev = nv; __hip_atomic_store(A.out_count, nv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  } else if (tid == 0) { *A.count_dev = nv; if (A.out_count) *A.out_count = nv; }
}

// ---- RawDataPreprocessorProjective2D (row f2): one workgroup per scan, everything in LDS -----------------------
static constexpr int kPrepBlock = kVoxBlock;     // one beam per thread for the window walks; one compare-exchange per thread in the sort
static constexpr int kPrepMaxBeams = kVoxMax;
struct PrepArgs {
  const float* ranges; const float2* beam_dir;      // [n_scans][n_beams]; (cos, sin) per beam, host-computed
  int32_t n_beams, stride;                           // stride: points reserved per output cloud (even)
  float rmin, rmax, d2max; int32_t min_points; float inv_res;   // inv_res <= 0: no voxelisation
  float2* out_xy; float2* out_nrm; int32_t* out_count;
  float4* out_aos = nullptr;                         // the set's (x, y, nx, ny) rows, rewritten in place by a refill (lsm2d_preprocess_scans_refill), or nullptr
};

// kBlock threads, room for kCap beams.  (1024, 2048): one beam per thread, what a scan alone on the chip wants (the live tracker).  (512, 1152), round 5: a
// BATCH of scans preprocessed beside a k_align launch in flight -- a workgroup of 512 threads and 37 KB is exactly what one retiring k_align workgroup leaves
// free, where the 1024-thread, 64 KB form had to wait for two slots of one CU to come free together; four of them per CU when the chip is theirs.
template <int kBlock, int kCap>
LSM2D_DEV void preprocess_scan_body(const PrepArgs& A, const int scan) {
  constexpr int kPrepBlock = kBlock;         // (shadows the full-size constant: every loop below strides by the workgroup's own size)
  // (the sort pads to a power of two: its keys need 2048 entries as soon as more than 1024 points carry a normal.  The small form cannot afford them beside the
  // three point arrays -- so its keys LIVE where the unprojected points were: those are dead once the normals are out, a barrier before the first key is written)
  constexpr int kKeyCap = kCap <= 1024 ? 1024 : 2048;
  constexpr bool kKeysOverPoints = kCap < kPrepMaxBeams;
  __shared__ u64 s_key[kKeyCap];             // (voxel key << 16) | index, bitonic-sorted
  __shared__ float2 s_p_own[kKeysOverPoints ? 1 : kCap];
  float2* const s_p = kKeysOverPoints ? reinterpret_cast<float2*>(s_key) : s_p_own;      // unprojected points, beam order
  __shared__ float2 s_q[kCap];               // points that got a normal
  __shared__ float2 s_n[kCap];               // their normals
  __shared__ int s_tot[2 * (kPrepBlock / 64)];
  const int tid = threadIdx.x, nb = A.n_beams;
#ifdef LSM2D_PHASE_CLOCKS
  unsigned long long pc_t = __builtin_amdgcn_s_memrealtime(), pc_acc[6] = {0, 0, 0, 0, 0, 0};
#define LSM2D_PC(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); pc_acc[k] += n_ - pc_t; pc_t = n_; } while (0)
#else
#define LSM2D_PC(k) do { } while (0)
#endif
  const float* rg = A.ranges + (size_t) scan * nb;
  float2* oxy = A.out_xy + (size_t) scan * A.stride; float2* onr = A.out_nrm + (size_t) scan * A.stride;
  float4* oaos = A.out_aos ? A.out_aos + (size_t) scan * A.stride : nullptr;
  // ---- F2.1 unprojection, valid beams compacted in beam order
  int m = 0, parity = 0;
  for (int c0 = 0; c0 < nb; c0 += kPrepBlock, parity ^= 1) {
    const int c = c0 + tid;
    float r = 0.0f; bool ok = false; float2 d = make_float2(0.0f, 0.0f);
    if (c < nb) { r = rg[c]; d = A.beam_dir[c]; ok = r >= A.rmin && r <= A.rmax; }
    const int pos = block_compact_pos(ok, s_tot, parity, m, tid, kPrepBlock / 64);
    if (ok) s_p[pos] = make_float2(r * d.x, r * d.y);
  }
  __syncthreads();
  LSM2D_PC(0);
  // ---- F2.2 sliding-window normals
  int k = 0;
  for (int i0 = 0; i0 < m; i0 += kPrepBlock, parity ^= 1) {
    const int i = i0 + tid;
    bool ok = false; float vx = 0.0f, vy = 0.0f; float2 pi = make_float2(0.0f, 0.0f);
    if (i < m) {
      pi = s_p[i];
      // Round 5: the window walks and the two sums were one LDS round trip per step (load -> compare -> branch; load -> add), 13-33 us of a scan's ~35 -- the
      // longest window of the workgroup (hundreds of points at close range) sets the pace.  Four neighbours per trip now: the loads of a trip are independent, the
      // tests and the additions keep the reference's order (first neighbour that fails ends the walk; sums run j = lo .. hi one after the other): same bits.
      const auto within = [&](int j) { const float2 q = s_p[j]; const float dx = q.x - pi.x, dy = q.y - pi.y; return __builtin_fmaf(dx, dx, dy * dy) <= A.d2max; };
      int lo = i, hi = i;
      while (lo > 0) {
        const int j1 = lo - 1, j2 = lo >= 2 ? lo - 2 : 0, j3 = lo >= 3 ? lo - 3 : 0, j4 = lo >= 4 ? lo - 4 : 0;      // (clamped: the extra loads are of valid cells and never counted)
        const bool f1 = within(j1), f2 = within(j2), f3 = within(j3), f4 = within(j4);
        const int room = lo < 4 ? lo : 4;
        const int adv = !f1 ? 0 : (room < 2 || !f2) ? 1 : (room < 3 || !f3) ? 2 : (room < 4 || !f4) ? 3 : 4;
        lo -= adv;
        if (adv < 4) break;
      }
      while (hi < m - 1) {
        const int last = m - 1, j1 = hi + 1, j2 = hi + 2 <= last ? hi + 2 : last, j3 = hi + 3 <= last ? hi + 3 : last, j4 = hi + 4 <= last ? hi + 4 : last;
        const bool f1 = within(j1), f2 = within(j2), f3 = within(j3), f4 = within(j4);
        const int room = last - hi < 4 ? last - hi : 4;
        const int adv = !f1 ? 0 : (room < 2 || !f2) ? 1 : (room < 3 || !f3) ? 2 : (room < 4 || !f4) ? 3 : 4;
        hi += adv;
        if (adv < 4) break;
      }
      const int cnt = hi - lo + 1;
      if (cnt >= A.min_points) {
        float sx = 0.0f, sy = 0.0f;
        int j = lo;
        for (; j + 3 <= hi; j += 4) {
          const float2 a = s_p[j], b = s_p[j + 1], c = s_p[j + 2], d = s_p[j + 3];
          sx += a.x; sy += a.y; sx += b.x; sy += b.y; sx += c.x; sy += c.y; sx += d.x; sy += d.y;
        }
        for (; j <= hi; ++j) { sx += s_p[j].x; sy += s_p[j].y; }
        const float inv = 1.0f / (float) cnt, mx = sx * inv, my = sy * inv;
        float sxx = 0.0f, sxy = 0.0f, syy = 0.0f;
        const auto cov = [&](const float2 q) {
          const float dx = q.x - mx, dy = q.y - my;
          sxx = __builtin_fmaf(dx, dx, sxx); sxy = __builtin_fmaf(dx, dy, sxy); syy = __builtin_fmaf(dy, dy, syy);
        };
        for (j = lo; j + 3 <= hi; j += 4) {
          const float2 a = s_p[j], b = s_p[j + 1], c = s_p[j + 2], d = s_p[j + 3];
          cov(a); cov(b); cov(c); cov(d);
        }
        for (; j <= hi; ++j) cov(s_p[j]);
        const float tr = sxx + syy, df = sxx - syy;
        const float disc = __builtin_sqrtf(__builtin_fmaf(df, df, 4.0f * (sxy * sxy)));
        const float lmin = 0.5f * (tr - disc);
        const float v1x = sxy, v1y = lmin - sxx, v2x = lmin - syy, v2y = sxy;
        const float n1 = __builtin_fmaf(v1x, v1x, v1y * v1y), n2 = __builtin_fmaf(v2x, v2x, v2y * v2y);
        float nn = n1; vx = v1x; vy = v1y;
        if (n2 > n1) { vx = v2x; vy = v2y; nn = n2; }
        if (nn > 0.0f) {
          const float s = __builtin_sqrtf(nn);
          vx = vx / s; vy = vy / s;
          if (__builtin_fmaf(vx, pi.x, vy * pi.y) > 0.0f) { vx = -vx; vy = -vy; }
          ok = true;
        }
      }
    }
    LSM2D_PC(1);
    const int pos = block_compact_pos(ok, s_tot, parity, k, tid, kPrepBlock / 64);
    if (ok) { s_q[pos] = pi; s_n[pos] = make_float2(vx, vy); }
  }
  __syncthreads();
  LSM2D_PC(2);
  if (!(A.inv_res > 0.0f)) {                       // no voxelisation: every valid point, beam order
    for (int i = tid; i < k; i += kPrepBlock) { oxy[i] = s_q[i]; onr[i] = s_n[i]; if (oaos) oaos[i] = make_float4(s_q[i].x, s_q[i].y, s_n[i].x, s_n[i].y); }
    if (tid == 0) A.out_count[scan] = k;
    return;
  }
  // ---- F2.3 voxelisation: sort (key, index), average equal-key runs, ascending key order
  const int nv = voxelize_lds<kPrepBlock>(s_q, s_n, s_key, k, A.inv_res, 1.0f, s_tot, tid,
                                          [&](int pos, float x, float y, float nx, float ny) { oxy[pos] = make_float2(x, y); onr[pos] = make_float2(nx, ny); if (oaos) oaos[pos] = make_float4(x, y, nx, ny); });
  if (tid == 0) A.out_count[scan] = nv;
  LSM2D_PC(3);
#ifdef LSM2D_PHASE_CLOCKS
  if (tid == 0) printf("preprocess ticks(10ns): unproject %llu normals %llu compaction %llu voxelise %llu (beams %d valid %d normals %d voxels %d)\n", pc_acc[0], pc_acc[1], pc_acc[2], pc_acc[3], nb, m, k, nv);
#endif
#undef LSM2D_PC
}
__global__ __launch_bounds__(kPrepBlock) void k_preprocess_scans(const PrepArgs A) { preprocess_scan_body<kPrepBlock, kPrepMaxBeams>(A, blockIdx.x); }
static constexpr int kPrepSmallBlock = 512, kPrepSmallBeams = 1152;      // 4 x 8 B x 1152 = 36 KB + the wave totals: beside three k_align workgroups of a CU
__global__ __launch_bounds__(kPrepSmallBlock) void k_preprocess_scans_small(const PrepArgs A) { preprocess_scan_body<kPrepSmallBlock, kPrepSmallBeams>(A, blockIdx.x); }
// several scans, each with its own sensor geometry and its own output set, side by side (the live tracker's front and rear scanner:
// lsm2d_preprocess_scan_into defers its launch, the aligner call that reads both sets queues them together)
static constexpr int kPrepMulti = 4;
struct PrepMultiArgs { PrepArgs a[kPrepMulti]; };
__global__ __launch_bounds__(kPrepBlock) void k_preprocess_multi(const PrepMultiArgs M) { preprocess_scan_body<kPrepBlock, kPrepMaxBeams>(M.a[blockIdx.x], 0); }

// ---- lane-chunked copy of every cloud of a set for k_align's streaming pass (project_cloud_lanes) -------------
// slot t*nthreads + g of cloud c  <-  pair g*T_c + t of the cloud (two points), +inf where the cloud has ended
__global__ void k_lane_layout(const float2* __restrict__ xy, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                              const long long* __restrict__ lane_start, const int32_t* __restrict__ lane_T, int nthreads,
                              float4* __restrict__ out, int cloud0) {
  const int c = cloud0 + blockIdx.y, n = count[c], T = lane_T[c];
  const float2* p = xy + start[c];
  float4* o = out + lane_start[c];
  const long long slots = (long long) T * nthreads;
  const float inf = __builtin_huge_valf();
  for (long long m = blockIdx.x * (long long) blockDim.x + threadIdx.x; m < slots; m += (long long) gridDim.x * blockDim.x) {
    const int t = (int) (m / nthreads), g = (int) (m % nthreads);
    const long long pair = (long long) g * T + t;
    float4 v = make_float4(inf, inf, inf, inf);
    if (2 * pair < n) { const float2 a = p[2 * pair]; v.x = a.x; v.y = a.y; }
    if (2 * pair + 1 < n) { const float2 b = p[2 * pair + 1]; v.z = b.x; v.w = b.y; }
    o[m] = v;
  }
}

// bounding circle of every thread's chunk of the lane-chunked copy (chunk g of cloud c = the points [2 g T, 2 (g + 1) T) of the cloud):
// centre = centre of the chunk's bounding box, radius = the largest distance to it, rounded up; what chunk_may_matter() tests.
// One wave per chunk.
__global__ __launch_bounds__(256) void k_lane_bounds(const float2* __restrict__ xy, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                                                     const int32_t* __restrict__ lane_T, int nthreads, float4* __restrict__ out, int cloud0) {
  const int c = cloud0 + blockIdx.y, g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= nthreads) return;
  const int n = count[c], T = lane_T[c];
  const long long lo = 2ll * g * T, hi = lo + 2ll * T < n ? lo + 2ll * T : n;
  const float2* p = xy + start[c];
  float mnx = 3.402823466e+38f, mny = mnx, mxx = -mnx, mxy = -mnx;
  for (long long i = lo + lane; i < hi; i += 64) { const float2 v = p[i]; mnx = fminf(mnx, v.x); mxx = fmaxf(mxx, v.x); mny = fminf(mny, v.y); mxy = fmaxf(mxy, v.y); }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
    mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
  }
  float4 r = make_float4(0.0f, 0.0f, -1.0f, 0.0f);
  if (hi > lo) {
    const float cx = 0.5f * (mnx + mxx), cy = 0.5f * (mny + mxy);
    float d2 = 0.0f;
    for (long long i = lo + lane; i < hi; i += 64) { const float2 v = p[i]; const float dx = v.x - cx, dy = v.y - cy; d2 = fmaxf(d2, dx * dx + dy * dy); }
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 64));
    // non-finite points (they fail the range gate anyway) must not poison the circle: a chunk holding one keeps its points (rho = +inf never culls)
    const float rho = (d2 == d2) ? __builtin_sqrtf(d2) * 1.00001f + 1e-6f : __builtin_huge_valf();
    r = make_float4(cx, cy, (cx == cx && cy == cy) ? rho : __builtin_huge_valf(), 0.0f);
  }
  if (lane == 0) out[(size_t) c * nthreads + g] = r;
}

// the same per BLOCK of a chunk (block b of chunk g = the points [2 (g T + b B), 2 (g T + min((b + 1) B, T))) of the cloud, B = cull_block_steps(T)): entry
// (c * nbs + b) * nthreads + g (nbs = the set's block_stride); blocks beyond the chunk's last (or beyond the cloud's end) get rho < 0 = "no points".  One wave per block.
__global__ __launch_bounds__(256) void k_block_bounds(const float2* __restrict__ xy, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                                                      const int32_t* __restrict__ lane_T, int nthreads, float4* __restrict__ out, int cloud0, int nbs) {
  const int c = cloud0 + blockIdx.y, w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= nthreads * nbs) return;
  const int b = w / nthreads, g = w - b * nthreads;
  const int n = count[c], T = lane_T[c], B = cull_block_steps(T, nbs);
  const int t0 = b * B, t1 = t0 + B < T ? t0 + B : T;
  long long lo = 2ll * ((long long) g * T + t0), hi = 2ll * ((long long) g * T + t1);
  if (hi > n) hi = n;
  const float2* p = xy + start[c];
  float4 r = make_float4(0.0f, 0.0f, -1.0f, 0.0f);
  if (t0 < T && hi > lo) {
    float mnx = 3.402823466e+38f, mny = mnx, mxx = -mnx, mxy = -mnx;
    for (long long i = lo + lane; i < hi; i += 64) { const float2 v = p[i]; mnx = fminf(mnx, v.x); mxx = fmaxf(mxx, v.x); mny = fminf(mny, v.y); mxy = fmaxf(mxy, v.y); }
    for (int o = 32; o > 0; o >>= 1) {
      mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
      mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
    }
    const float cx = 0.5f * (mnx + mxx), cy = 0.5f * (mny + mxy);
    float d2 = 0.0f;
    for (long long i = lo + lane; i < hi; i += 64) { const float2 v = p[i]; const float dx = v.x - cx, dy = v.y - cy; d2 = fmaxf(d2, dx * dx + dy * dy); }
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 64));
    const float rho = (d2 == d2) ? __builtin_sqrtf(d2) * 1.00001f + 1e-6f : __builtin_huge_valf();      // (a non-finite point: never culled, as in k_lane_bounds)
    r = make_float4(cx, cy, (cx == cx && cy == cy) ? rho : __builtin_huge_valf(), 0.0f);
  }
  if (lane == 0) out[((size_t) c * nbs + b) * nthreads + g] = r;
}

// (x, y, nx, ny) rows of a whole set next to its split arrays (CloudDev::aos)
__global__ void k_aos_rows(const float2* __restrict__ xy, const float2* __restrict__ nrm, long long n, float4* __restrict__ out) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
    const float2 p = xy[i], q = nrm[i];
    out[i] = make_float4(p.x, p.y, q.x, q.y);
  }
}

// bounding circle of every tile of 64 consecutive points (the point-query finders' culling, k_align): one wave per tile
__global__ __launch_bounds__(256) void k_tile_bounds(const float2* __restrict__ xy, const int32_t* __restrict__ start, const int32_t* __restrict__ count,
                                                     const int32_t* __restrict__ tile_start, float4* __restrict__ out, int cloud0) {
  const int c = cloud0 + blockIdx.y, lane = threadIdx.x & 63;
  const int n = count[c], n_tiles = (n + 63) >> 6;
  const float2* p = xy + start[c];
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += gridDim.x * 4) {
    const int i = t * 64 + lane; const bool in = i < n;
    const float2 v = in ? p[i] : make_float2(0.0f, 0.0f);
    float mnx = in ? v.x : 3.402823466e+38f, mny = in ? v.y : 3.402823466e+38f, mxx = in ? v.x : -3.402823466e+38f, mxy = in ? v.y : -3.402823466e+38f;
    for (int o = 32; o > 0; o >>= 1) {
      mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
      mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
    }
    const float cx = 0.5f * (mnx + mxx), cy = 0.5f * (mny + mxy);
    const float dx = v.x - cx, dy = v.y - cy;
    float d2 = in ? dx * dx + dy * dy : 0.0f;
    bool bad = in && !(d2 == d2);                 // a non-finite point must not poison the circle: its tile is never skipped (rho = +inf)
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 64));
    bad = __ballot(bad) != 0ull || !(cx == cx && cy == cy);
    if (lane == 0) out[(size_t) tile_start[c] + t] = make_float4(cx, cy, bad ? __builtin_huge_valf() : __builtin_sqrtf(d2) * 1.00001f + 1e-6f, 0.0f);
  }
}

// ---- refill of a small single-cloud set straight from its pinned staging buffer (lsm2d_cloudset_upload): the kernel reads the
//      host's AoS points over the bus and writes the split arrays and the count -- one launch instead of three copies ----
__global__ __launch_bounds__(256) void k_upload_unpack(const float4* __restrict__ host_aos, int n, float2* __restrict__ xy, float2* __restrict__ nrm,
                                                       int32_t* __restrict__ count) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 v = host_aos[i];
    xy[i] = make_float2(v.x, v.y); nrm[i] = make_float2(v.z, v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = n;
}

// ---- cloud repack: AoS float4 -> xy / normal arrays, cloud c starting at padded index pstart[c] ----
__global__ void k_repack_cloud(const float4* __restrict__ src, const int32_t* __restrict__ offsets, const int32_t* __restrict__ pstart,
                               int n_clouds, long long total, float2* __restrict__ xy, float2* __restrict__ nrm) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x) {
    int lo = 0, hi = n_clouds - 1;               // last cloud with offsets[c] <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((long long) offsets[mid] <= i) lo = mid; else hi = mid - 1; }
    const long long d = (long long) pstart[lo] + (i - offsets[lo]);
    const float4 v = src[i];
    xy[d] = make_float2(v.x, v.y); nrm[d] = make_float2(v.z, v.w);
  }
}

}  // namespace lsm2d
