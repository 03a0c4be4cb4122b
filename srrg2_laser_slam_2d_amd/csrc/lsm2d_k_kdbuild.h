// lsm2d_k_kdbuild.h -- the reference's KD-tree, built on the device (CorrespondenceFinderKDTree2D::reset, registration/correspondence_finder_kd_tree_2d.cpp:31-38).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
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
