// lsm2d_k_structures.h -- building the distance map and the uniform search grid (CorrespondenceFinderNN2D: registration/correspondence_finder_nn_2d.cpp:20-52; the grid stands where the reference rebuilds its tree in reset(), correspondence_finder_kd_tree_2d.cpp:31-38).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
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
