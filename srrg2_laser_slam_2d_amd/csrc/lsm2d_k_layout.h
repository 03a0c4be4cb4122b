// lsm2d_k_layout.h -- derived layouts of a cloud set: the lane-chunked copy k_align streams, chunk / block / tile bounding circles, AoS rows; upload unpacking and repacking.
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
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
