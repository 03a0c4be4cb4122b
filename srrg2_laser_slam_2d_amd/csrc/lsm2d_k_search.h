// lsm2d_k_search.h -- the search structures' device views (uniform grid, distance map, KD-tree), the cloud view, and the point-query search functions the aligner and the finder-level kernels share (CorrespondenceFinderKDTree2D / NN2D: registration/correspondence_finder_kd_tree_2d.cpp:12-27, correspondence_finder_nn_2d.cpp:63-80).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
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
#ifndef LSM2D_KD_NODE_TOGETHER
#define LSM2D_KD_NODE_TOGETHER 1
#endif
  int k = 0;
  int2 L;
#if LSM2D_KD_NODE_TOGETHER
  // Round 6: a node's two halves -- its plane and its link, one 32-byte record -- are asked for TOGETHER as soon as the node is known.  (Before: the link first,
  // and the plane only inside the loop, behind the test of the link: below the levels staged in LDS every level of the descent cost two dependent trips to L2
  // instead of one.  The leaf's plane is fetched for nothing: the same line.)  The same tests on the same values: the same leaf.
  float4 P;
  auto node = [&](int kk) {
    if (kAllLds || kk < lds_nodes) { P = l_plane[kk]; L = l_link[kk]; }
    else { P = reinterpret_cast<const float4*>(nodes)[2 * kk]; const int4 w = reinterpret_cast<const int4*>(nodes)[2 * kk + 1]; L = make_int2(w.x, w.y); }
  };
  node(0);
  while (L.x >= 0) {
    const float t = (qx - P.x) * P.z + (qy - P.y) * P.w;
    k = L.x + (t < 0.0f ? 0 : 1);
    node(k);
  }
#else
  if (kAllLds || lds_nodes > 0) L = l_link[0]; else { const int4 w = reinterpret_cast<const int4*>(nodes)[1]; L = make_int2(w.x, w.y); }
  while (L.x >= 0) {
    float4 P;
    if (kAllLds || k < lds_nodes) P = l_plane[k]; else P = reinterpret_cast<const float4*>(nodes)[2 * k];
    const float t = (qx - P.x) * P.z + (qy - P.y) * P.w;
    k = L.x + (t < 0.0f ? 0 : 1);
    if (kAllLds || k < lds_nodes) L = l_link[k]; else { const int4 w = reinterpret_cast<const int4*>(nodes)[2 * k + 1]; L = make_int2(w.x, w.y); }
  }
#endif
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
