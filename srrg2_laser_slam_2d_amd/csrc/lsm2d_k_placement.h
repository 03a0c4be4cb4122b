// lsm2d_k_placement.h -- balanced placement of culled batches: the work estimate and the deal of alignments to workgroup ids (k_cull_estimate).
// Part of lsm2d_kernels.h (included there, inside namespace lsm2d, in this order); not a translation unit of its own.
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
// Round 6 (late), PACKED batches: order2 != nullptr and first-round slots < n <= 2 x slots -- the batch runs in ONE dispatch round of `slots` workgroups (k_align_two), its
// lightest 2 E alignments (E = n - slots) two to a workgroup, one after the other: rank slots - E + j (heaviest-of-the-light) with rank n - 1 - j (lightest).  The
// deal below then places `slots` ITEMS: item j < E = pair j, item E + r = the single alignment of rank r; an alignment counts as its chunks + kPackConst (the phases
// outside its stream: with them a pair of light alignments is the heavy item it is -- measured: 1025 alignments 0.835 us each without the constant, 0.734 with).  The
// pairs' sums are roughly equal and above the singles': the item order is close enough to descending for the level-by-level deal.  order[b] / order2[b]: workgroup
// b's first and second alignment (-1: none); order + kPackSecondsAt is scratch for the seconds (alignment | chunks << 16).  The same for a batch that fills two or
// three rounds and a little more (n_bins = 2 or 3 x slots workgroups, dispatched round after round): the pairs are the heaviest items and all lie in the first round's deal.
static constexpr int kPackConst = 150, kPackMaxRounds = 3;
static constexpr int kOrderInts = 8192, kPackOrder2At = 4096, kPackSecondsAt = 4096 + kPackMaxRounds * 1024;      // lsm2d_context::d_order: [4096] order | [3072] order2 | [1024] seconds
LSM2D_DEV void balance_order(BalanceLds& L, const int32_t* work, int n, int n_cu, int per_cu, int32_t* __restrict__ order, const int32_t* place, int tid, int nt, int32_t* __restrict__ order2 = nullptr) {
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
  // (the host asks for it only then: n_bins = the whole rounds the batch fills, at most kPackMaxRounds of them -- the seconds' alignment numbers have 12 bits)
  const int n_bins = first > 0 ? (n / first) * first : 0;
  const bool packed = order2 != nullptr && first == n_cu * per_cu && n > n_bins && n_bins >= first && n_bins <= kPackMaxRounds * first && n - n_bins <= first && n < 4096;
  const int n_pairs = packed ? n - n_bins : 0, n_singles = n_bins - n_pairs;
  int32_t* seconds = order + kPackSecondsAt;
  for (int a = tid; a < n; a += nt) {
    const int w = a < 2048 ? (int) L.wall[a] : __hip_atomic_load(&work[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int r = atomicAdd(&L.bin[w], 1);                      // rank among all alignments (ties in any order: placement only)
    if (packed) {
      if (r < n_singles) {      // item n_pairs + r: in the first round's deal, or -- beyond it -- in its place of the heaviest-first order
        const int item = n_pairs + r;
        if (item < first) { L.sorted[item] = a; L.sw[item] = (unsigned short) (w + kPackConst); } else order[item] = a;
      }
      else if (r < n_bins) { L.sorted[r - n_singles] = a; L.sw[r - n_singles] = (unsigned short) (w + kPackConst); }      // a pair's first (n_pairs <= first: every pair is in the deal)
      else __hip_atomic_store(&seconds[n - 1 - r], a | (w << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // ... and its second
    }
    else
    if (r < first) { L.sorted[r] = a; L.sw[r] = (unsigned short) w; } else order[r] = a;      // beyond the first round: the heaviest go first
  }
  if (packed) {
    __syncthreads();
    for (int j = tid; j < n_pairs; j += nt) L.sw[j] = (unsigned short) ((int) L.sw[j] + (__hip_atomic_load(&seconds[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) + kPackConst);
    for (int b2 = first + tid; b2 < n_bins; b2 += nt) order2[b2] = -1;      // (beyond the first round: singles)
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
    for (int r = tid; r < first; r += nt) { order[r] = L.sorted[r]; if (packed) order2[r] = r < n_pairs ? (__hip_atomic_load(&seconds[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFF) : -1; }
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
  for (int b2 = tid; b2 < first; b2 += nt) {
    const int item = (int) L.assign[(int) L.gid[L.wg_key[b2]] * kBalMaxLevels + L.wg_slot[b2]];
    order[b2] = L.sorted[item];
    if (packed) order2[b2] = item < n_pairs ? (__hip_atomic_load(&seconds[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFF) : -1;
  }
}

__global__ __launch_bounds__(kAlignBlock) void k_cull_estimate(const AlignArgs A, int slice, int32_t* __restrict__ work,
                                                               int32_t* __restrict__ order /* or nullptr: counts only */, const int32_t* __restrict__ place, int n_cu, unsigned int* __restrict__ done_counter,
                                                               int32_t* __restrict__ order2 = nullptr /* packed batch: see balance_order */) {
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
  balance_order(*reinterpret_cast<BalanceLds*>(smem), work, (int) gridDim.x, n_cu, 4, order, place, tid, kAlignBlock, order2);
  if (tid == 0) __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next call
}

#ifdef LSM2D_EXPERIMENTS
__global__ __launch_bounds__(kAlignBlock) void k_balance_only(const int32_t* __restrict__ work, int n, int n_cu, int32_t* __restrict__ order, const int32_t* __restrict__ place) {
  extern __shared__ __align__(16) unsigned char smem[];      // BalanceLds
  balance_order(*reinterpret_cast<BalanceLds*>(smem), work, n, n_cu, 4, order, place, threadIdx.x, kAlignBlock);
}
#endif
