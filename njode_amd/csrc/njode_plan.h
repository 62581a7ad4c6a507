// njode_plan.h -- the segment plan of a batch in ONE launch (round 5).
//
// The plan (row -> time slice, the rows of every path linked in time order, the STABLE order of the
// rows by segment length, the trajectory layout and the split points of the mixed ODE kernels) used
// to be seven to ten small dispatches: beside the step on a second queue they cost the step 32 us at
// 20 000 paths and 17 us at B = 100 -- half of it the hand-over between the two queues (event record,
// event wait, the first launch behind them), half the kernels of the step getting slower
// (profiles/r05_plan_cost.txt).  gfx950 has no way to let a kernel overtake its predecessor on one
// queue (hipExtAnyOrderLaunch is not honoured: tools/ubench/anyorder_check.hip,
// profiles/r05_anyorder_check.txt), so the plan of the NEXT batch is built by the FIRST `P` BLOCKS OF
// THE ODE FORWARD'S LAUNCH of the current step (k_ode_fwd_mixed, njode_mfma_split.h): same queue, no
// events, no dispatches of its own, beside a kernel that is long enough to cover it.  The stages are
// separated by grid barriers among the P plan blocks (all of them resident from the start: they hold the
// lowest block indices; P = rows / 4096 + 1 <= 64 when hosted, at most 256 = one per CU: njode_api.hip);
// P = 1 needs workgroup barriers only.  The same body is also a kernel of its own (k_plan_grid,
// njode_api.hip) for calls that build their plan in line -- whole for small plans, from its third stage
// on (first_stage = 2) behind k_row_time for large ones.
//
// Same arrays, bit for bit, as the multi-launch plan: the order is a stable counting sort whatever
// the row-block size, the layout's sums are integers (tests/test_hip_switches.py).
#pragma once
#include <hip/hip_runtime.h>

namespace njode {

constexpr int PLAN_KEYS = 512;            // K + 1 must fit
constexpr int PLAN_TIMES = 1500;          // n_times + 1 must fit (beside the histogram)
constexpr int PLAN_THREADS = 256;
constexpr int PLAN_LDS_INTS = 4 * PLAN_KEYS + 96;
constexpr int PLAN_SCAN_CHUNKS = 32;      // row blocks per key held in registers: 64 x 32 = 2 048
constexpr int SPLIT_KMAX = 4095;
struct SplitCfg { int on, ns[2], nw[2]; float r[2]; int queue[2]; };   // queue[v]: kernel v pops its tiles (tile queue)

struct PlanJob {
  int P;                        // plan blocks in front of the launch (0: none)
  int first_stage;              // 0: everything; 2: the dense matrix, the row times and the cleared histogram
                                // are already in place (k_row_time: a call that plans in line hands its
                                // encoder rows to a helper stream as soon as the row times exist)
  int last_stage;               // 5: everything; 2: stop behind the links (the consumer runs one wave per
                                // item, njode_chain_seg.h: it needs neither the order by length nor the layout)
  int n, B, K, n_times;
  int cs_shift, cs_nwb;         // row blocks of the counting sort: 2^cs_shift rows each
  const int* sched_src;         // pinned host copy of the schedule (device-visible), or null: already in place
  int* sched_dst;               // ... its place in the plan buffer
  int sched_ints;
  const int* time_ptr;          // (inside sched_dst)
  const int* k_jump;
  const int* obs_idx;
  int* t_of_row;
  int* dense;
  int* first_row;
  int* last_row;
  int* item_prev;
  int* item_next;
  int* item_kbeg;
  int* item_len;
  unsigned* sort_key;
  int* len_hist;
  int* cs_tab;
  int* order;
  long long* base_s;
  long long* base16_s;
  unsigned* sync;               // [8] grid-barrier counters OF THIS PLAN (in its own buffer, zeroed on the launch's
  unsigned sync_base;           // stream right before it: no two launches ever share a counter); base = 0
  unsigned* fail;               // device word of the library: set when a barrier gave up waiting (plan_sync)
  unsigned long long* stamps;   // maintainer aid (NJODE_PLAN_STAMPS=1): [block][8] wall clock at the stage ends
  SplitCfg sc;
};

// What the plan blocks hand to each other between two stages goes through pld / pst: agent-scope
// relaxed atomics, i.e. loads and stores with the sc1 bit -- coherent across the eight XCDs' L2s access
// by access.  The textbook alternative (plain accesses + __threadfence() around the barrier) makes
// every thread write back and invalidate its XCD's whole L2 at every barrier: with the ODE forward
// streaming its step records beside the plan that took the plan 370 us at 20 000 paths (137 alone) and the
// kernels beside it up to twice their time (profiles/r05_plan_in_forward.txt).  COH = false (one
// plan block): plain accesses.
template <bool COH, class T> __device__ __forceinline__ T pld(const T* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH, class T> __device__ __forceinline__ void pst(T* p, T v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// Barrier among the P plan blocks: every wave waits for its own stores (write-through: acknowledged at
// the coherence point), the workgroup meets, thread 0 counts the block in and waits for the others.
// No cache-wide operation.
//
// Ordering: the handed-over data is written with sc1 stores, and what orders them in front
// of the arrival is the s_waitcnt below -- inline assembly with a memory clobber, so the compiler can
// move none of those stores (nor the histogram's global atomics) behind it; the same on the reader's
// side behind the spin.  Visibility itself is MI355X_MICROARCH.md's measured sc1 / drained-counter form.
//
// The spin is BOUNDED: a barrier that is not complete after ~2^22 polls (seconds; a
// healthy one takes microseconds) sets *fail and lets the launch run to its end -- on garbage, which the
// host reports (njode_plan_barrier_failures(), checked by the tests) instead of a hung device.
constexpr int PLAN_SPIN_MAX = 1 << 22;
__device__ inline void plan_sync(unsigned* ctr, unsigned base, int P, unsigned* fail) {
  if (P > 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base) < P) {
        if (++spins > PLAN_SPIN_MAX) {
          if (fail) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  } else {
    __syncthreads();
  }
}

// Trajectory layout + split points from the length histogram (one workgroup of NT threads).
// cnt_s = #{len > s}; base_s = exclusive prefix sum of cnt; base16_s the same with every count rounded
// up to a whole tile; base_s[K+1], [K+2] = split points T of the mixed ODE backward / forward.
// lds: SPLIT_KMAX + 1 + 64 ints when K <= SPLIT_KMAX is to take the workgroup scans (lds_cnt), else unused.
template <int NT, bool COH = false>
__device__ inline void traj_layout_body(const int* len_hist, int n_obs, int K, long long* base_s,
                                        long long* base16_s, const SplitCfg& sc, int* lds, bool lds_cnt) {
  constexpr int NWV = NT / 64;
  int* cnt = lds;
  long long* wsum = (long long*)(lds + ((K + 2) & ~1));            // [NWV]
  float* best_c = (float*)(wsum + NWV);                            // [2][NWV]
  int* best_t = (int*)(best_c + 2 * NWV);                          // [2][NWV]
  const int tid = threadIdx.x;
  // exclusive scan of one value per thread over the workgroup (ascending thread order);
  // *total receives the workgroup sum
  auto block_excl = [&](long long v, long long* total) -> long long {
    long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long up = __shfl_up(incl, o);
      if ((tid & 63) >= o) incl += up;
    }
    __syncthreads();
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    long long before = 0, all = 0;
    for (int i = 0; i < NWV; ++i) {
      if (i < (tid >> 6)) before += wsum[i];
      all += wsum[i];
    }
    if (total) *total = all;
    return before + incl - v;
  };
  // Both as workgroup scans: thread t owns the chunk [t * ch, (t + 1) * ch).
  if (lds_cnt) {
    const int ch = (K + 1 + NT - 1) / NT;
    for (int s = tid; s <= K; s += NT) cnt[s] = pld<COH>(len_hist + s);
    __syncthreads();
    {   // suffix: scan the reversed array (thread t owns the reversed chunk)
      long long loc = 0;
      for (int q = 0; q < ch; ++q) {
        const int s = K - (tid * ch + q);
        if (s >= 0) loc += cnt[s];
      }
      long long run = block_excl(loc, nullptr);   // histogram mass strictly above this chunk
      for (int q = 0; q < ch; ++q) {
        const int s = K - (tid * ch + q);
        if (s >= 0) {
          const int h = cnt[s];
          cnt[s] = (int)run;
          run += h;
        }
      }
    }
    __syncthreads();
    {
      long long loc = 0;
      for (int q = 0; q < ch; ++q) {
        const int s = tid * ch + q;
        if (s <= K) loc += cnt[s];
      }
      long long run = block_excl(loc, nullptr);
      for (int q = 0; q < ch; ++q) {
        const int s = tid * ch + q;
        if (s <= K) {
          base_s[s] = run;
          run += cnt[s];
        }
      }
    }
    __syncthreads();
    {   // the same with every count rounded up to a whole tile (stored activations)
      long long loc = 0;
      for (int q = 0; q < ch; ++q) {
        const int s = tid * ch + q;
        if (s <= K) loc += (cnt[s] + 15) & ~15;
      }
      long long run = block_excl(loc, nullptr);
      for (int q = 0; q < ch; ++q) {
        const int s = tid * ch + q;
        if (s <= K) {
          base16_s[s] = run;
          run += (cnt[s] + 15) & ~15;
        }
      }
    }
    __syncthreads();
  } else if (tid == 0) {       // very long schedules: two serial passes over global memory
    long long above = 0;
    for (int s = K; s >= 0; --s) {
      const long long h = len_hist[s];
      base_s[s] = above;
      above += h;
    }
    long long run = 0, run16 = 0;
    for (int s = 0; s <= K; ++s) {
      const long long c = base_s[s];
      base_s[s] = run;
      base16_s[s] = run16;
      run += c;
      run16 += (c + 15) & ~15ll;
    }
  }
  if (!sc.on || !lds_cnt) return;
  // ---- split point: thread t owns the candidate tiles [lo, hi); pass 1 sums their lengths
  // (workgroup scan -> H(lo) and S), pass 2 evaluates the cost of every candidate
  constexpr float INF = 3.0e38f;
  const int n_tiles = (n_obs + 15) / 16;
  const int chunk = (n_tiles + NT - 1) / NT;
  const int lo = tid * chunk < n_tiles ? tid * chunk : n_tiles;
  const int hi = lo + chunk < n_tiles ? lo + chunk : n_tiles;
  auto len = [&](int t) {           // first s with cnt[s] <= 16 t  (cnt is non-increasing)
    int a = 0, b = K + 1;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (cnt[m] > 16 * t) a = m + 1; else b = m;
    }
    return a;
  };
  const int l_lo = lo < n_tiles ? len(lo) : 0;
  long long mine = 0;
  {
    int lcur = l_lo;                           // lengths only shrink along the chunk
    for (int t = lo; t < hi; ++t) {
      while (lcur > 0 && cnt[lcur - 1] <= 16 * t) --lcur;
      mine += lcur;
    }
  }
  long long all = 0;
  const float head = (float)block_excl(mine, &all);
  const float total = (float)all;
  const float l0 = n_tiles > 0 ? (float)len(0) : 0.0f;
  float bc[2] = {INF, INF};
  int bt[2] = {0, 0};
  float h = head;
  int lcur = l_lo;
  for (int t = lo; t <= hi; ++t) {            // candidate T = t (t == hi only at the very end)
    if (t == hi && hi != n_tiles) break;
    while (lcur > 0 && cnt[lcur - 1] <= 16 * t) --lcur;
    const float lt = t < n_tiles ? (float)lcur : 0.0f;
    for (int v = 0; v < 2; ++v) {
      float cs = 0.0f, cw = 0.0f;
      if (t > 0) cs = sc.ns[v] > 0 ? fmaxf(l0, h / sc.ns[v]) / sc.r[v] : INF;
      if (t < n_tiles) cw = sc.nw[v] > 0 ? fmaxf(lt, (total - h) / sc.nw[v]) : INF;
      float cost = fmaxf(cs, cw);
      if (sc.queue[v] && sc.ns[v] > 0 && sc.nw[v] > 0) {
        // Tile queue (njode_ode2.h): both roles pop until the tiles are gone -- four-wave blocks go
        // on with the bulk's when theirs are done -- so the launch takes ~ mk = S / (nw + R ns)
        // whatever T is; what T must guarantee is that no single tile outlasts that: the longest
        // bulk tile (it starts first: the queue is longest-first) within 0.9 mk, the four-wave tiles
        // within mk.  A tile costs a four-wave block 4 / R = ~2x the SIMD time it costs a bulk wave
        // (profiles/r05_bwd_fixed_costs.txt: 2.3 us on four SIMDs against 4.6 us on one), so the
        // SMALLEST such T is the cheapest.  (No T qualifies: the balance formula above.)
        const float mk = total / ((float)sc.nw[v] + sc.r[v] * (float)sc.ns[v]);
        const bool ok = lt <= 0.9f * mk && (t == 0 || (l0 / sc.r[v] <= mk && h / (sc.ns[v] * sc.r[v]) <= mk));
        cost = ok ? (float)t * 1.0e-3f : 1.0e6f + cost;
      }
      if (cost < bc[v]) { bc[v] = cost; bt[v] = t; }
    }
    h += lt;
  }
  // argmin over the workgroup (ties: smaller T): wave shuffles, then the wave results
  for (int v = 0; v < 2; ++v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float oc = __shfl_xor(bc[v], o);
      const int ot = __shfl_xor(bt[v], o);
      if (oc < bc[v] || (oc == bc[v] && ot < bt[v])) { bc[v] = oc; bt[v] = ot; }
    }
  }
  if ((tid & 63) == 0)
    for (int v = 0; v < 2; ++v) { best_c[v * NWV + (tid >> 6)] = bc[v]; best_t[v * NWV + (tid >> 6)] = bt[v]; }
  __syncthreads();
  if (tid < 2) {
    float c = INF;
    int t = 0;
    for (int i = 0; i < NWV; ++i)
      if (best_c[tid * NWV + i] < c || (best_c[tid * NWV + i] == c && best_t[tid * NWV + i] < t)) {
        c = best_c[tid * NWV + i];
        t = best_t[tid * NWV + i];
      }
    base_s[K + 1 + tid] = t;
  }
}

// One plan block (PLAN_THREADS threads) of P.  lds: PLAN_LDS_INTS ints.
template <bool COH>
__device__ inline void plan_grid_stages(const PlanJob& j, int pb, int* lds) {
  constexpr int NT = PLAN_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int P = j.P, n = j.n, B = j.B, K = j.K, nt = j.n_times;
  const int gtid = pb * NT + tid, gthreads = P * NT;
  const int gw = pb * 4 + w, gwaves = P * 4;
  const int nkeys = K + 1;
  auto stamp = [&](int slot) {
    if (j.stamps && tid == 0) j.stamps[pb * 8 + slot] = wall_clock64();
  };
  stamp(6);
  // (hosted by the ODE forward: its waves keep the SIMD's matrix / vector pipe ~90 % busy; the plan's
  // short dependent instruction chains go first whenever they are ready)
  __builtin_amdgcn_s_setprio(3);
  if (j.first_stage >= 2) {
    // (the two barriers this launch does not need still get their P arrivals: the host hands out a
    // counter SET per launch and every counter of it moves on by P)
    if (P > 1 && tid == 0) {
      __hip_atomic_fetch_add(j.sync + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(j.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
  // ---- stage 0: schedule into place (block 0, from the pinned host copy), dense = -1, histogram = 0
  if (pb == 0) {
    if (j.sched_src)
      for (int i = tid; i < j.sched_ints; i += NT) pst<COH>(j.sched_dst + i, j.sched_src[i]);
    for (int s = tid; s < nkeys; s += NT) pst<COH>(j.len_hist + s, 0);
  }
  {
    const size_t cells = (size_t)nt * B;
    if constexpr (COH) {
      long long* d2 = (long long*)j.dense;     // (256-byte aligned region)
      const size_t c2 = cells >> 1;
      for (size_t i = gtid; i < c2; i += gthreads) pst<COH>(d2 + i, -1ll);
      if ((cells & 1) && gtid == 0) pst<COH>(j.dense + cells - 1, -1);
    } else {
      int4* d4 = (int4*)j.dense;
      const size_t c4 = cells >> 2;
      for (size_t i = gtid; i < c4; i += gthreads) d4[i] = make_int4(-1, -1, -1, -1);
      for (size_t i = (c4 << 2) + gtid; i < cells; i += gthreads) j.dense[i] = -1;
    }
  }
  plan_sync(j.sync + 0, j.sync_base, P, j.fail);
  stamp(0);
  // ---- stage 1: time slice of every row (binary search over the CSR offsets, held in LDS), scatter
  // into the dense [time slice][path] matrix
  for (int i = tid; i <= nt; i += NT) lds[i] = pld<COH>(j.time_ptr + i);
  __syncthreads();
  // (a wave takes 256 consecutive rows at a time, lane l the rows base + 64 u + l: the rows are sorted
  // by time, so the slice of u + 1 is found from the slice of u by stepping, not by searching again)
  for (int base = gw * 256; base < n; base += gwaves * 256) {
    int path[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = base + u * 64 + lane;
      path[u] = r < n ? j.obs_idx[r] : 0;
    }
    int lo = 0;
    {
      const int r = min(base + lane, n - 1);
      int hi = nt;                             // time_ptr[lo] <= r < time_ptr[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (lds[mid] <= r) lo = mid; else hi = mid;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = base + u * 64 + lane;
      if (r < n) {
        while (lo + 1 < nt && lds[lo + 1] <= r) ++lo;
        j.t_of_row[r] = lo;
        pst<COH>(j.dense + (size_t)lo * B + path[u], r);
      }
    }
  }
  plan_sync(j.sync + 1, j.sync_base, P, j.fail);
  }
  stamp(1);
  // ---- stage 2: every path walks its column in time order: links, item lengths, sort keys, the
  // length histogram (per block in LDS, integer atomics: order-independent).  k_jump from LDS.
  int* hist = lds;
  int* kj = lds + PLAN_KEYS;
  for (int s = tid; s < nkeys; s += NT) hist[s] = 0;
  for (int i = tid; i < nt; i += NT) kj[i] = pld<COH>(j.k_jump + i);
  __syncthreads();
  for (int b = gtid; b < B; b += gthreads) {
    int prev = -1, kprev = 0;
    // 32 independent loads in flight.  (NOT more: the hosting kernel's register count is the maximum over
    // the plan's branch and the forward's -- with 64 here k_ode_fwd_mixed_plan took 256 VGPRs instead of
    // 154, the FORWARD's blocks fell to one wave per SIMD and the step went from 0.86 to 1.07 ms.  After
    // touching this file: tools/isa_dump.sh 0 and look at .vgpr_count of k_ode_fwd_mixed_plan, <= 168.)
    constexpr int CH = 32;
    for (int i0 = 0; i0 < nt; i0 += CH) {
      int rr[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) rr[q] = i0 + q < nt ? pld<COH>(j.dense + (size_t)(i0 + q) * B + b) : -1;
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int r = rr[q];
        if (r < 0) continue;
        const int kend = kj[i0 + q];
        const int len = kend - kprev;
        j.item_prev[r] = prev;
        if (prev >= 0) j.item_next[prev] = r; else j.first_row[b] = r;
        j.item_kbeg[r] = kprev;
        j.item_len[r] = len;
        pst<COH>(j.sort_key + r, (unsigned)(K - len));   // ascending key == descending length
        if (len >= 0 && len <= K) atomicAdd(&hist[len], 1);
        prev = r;
        kprev = kend;
      }
    }
    if (prev >= 0) { j.item_next[prev] = -1; j.last_row[b] = prev; }
    else { j.first_row[b] = -1; j.last_row[b] = -1; }
  }
  __syncthreads();
  for (int s = tid; s < nkeys; s += NT)
    if (hist[s]) atomicAdd(&j.len_hist[s], hist[s]);
  plan_sync(j.sync + 2, j.sync_base, P, j.fail);
  stamp(2);
  if (j.last_stage <= 2) {
    // (the two barriers this launch does not reach still get their P arrivals: see first_stage above)
    if (P > 1 && tid == 0) {
      __hip_atomic_fetch_add(j.sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(j.sync + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(3);
    stamp(4);
    stamp(5);
    return;
  }
  // ---- stage 3: counting sort, count: one wave per row block counts its rows per key in counters of
  // its own (integer LDS atomics) -> table[key][row block].  The LAST block derives the trajectory
  // layout instead (P == 1: the only block, after its share): it only needs the histogram.
  const int nwb = j.cs_nwb, rounds = 1 << (j.cs_shift - 6);
  const bool layout_block = pb == P - 1;
  const int wwaves = P == 1 ? gwaves : gwaves - 4, wthreads = wwaves * 64;   // the blocks that sort
  if (!layout_block || P == 1) {
    int* cnt = lds + w * PLAN_KEYS;
    for (int wb = gw; wb < nwb; wb += wwaves) {
      for (int s = lane; s < nkeys; s += 64) cnt[s] = 0;
      const int r0 = wb << j.cs_shift;
      for (int it = 0; it < rounds; ++it) {    // (integer LDS atomics on the wave's own counters)
        const int r = r0 + it * 64 + lane;
        if (r < n) atomicAdd(&cnt[min((int)pld<COH>(j.sort_key + r), nkeys - 1)], 1);
      }
      for (int s = lane; s < nkeys; s += 64) pst<COH>(j.cs_tab + (size_t)s * nwb + wb, cnt[s]);
    }
  }
  if (layout_block) {
    if (P > 1) {
      // the layout only needs the histogram and nobody in this launch reads it: this block counts
      // itself in at the two barriers the others still need and works through their stages 3 - 5
      if (tid == 0) {
        __hip_atomic_fetch_add(j.sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(j.sync + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();                           // (P == 1: the counters above live in the same LDS)
    traj_layout_body<NT, COH>(j.len_hist, n, K, j.base_s, j.base16_s, j.sc, lds, true);
    if (P > 1) {
      stamp(3);
      stamp(4);
      stamp(5);
      return;
    }
  }
  plan_sync(j.sync + 3, j.sync_base, P, j.fail);
  stamp(3);
  // ---- stage 4: scan: base of every key (exclusive scan of the histogram by descending length, in
  // LDS) + exclusive scan of the key's row over the row blocks -- a thread per key while a row is short
  // (its loads independent, the scan in registers), a wave per key otherwise
  {
    int* kb = lds;                             // kb[key] = rows with a smaller key
    int* ws = lds + PLAN_KEYS;                 // [4] wave sums
    int carry = 0;
    for (int k0 = 0; k0 < nkeys; k0 += NT) {
      const int key = k0 + tid;
      const int v = key < nkeys ? pld<COH>(j.len_hist + K - key) : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      __syncthreads();
      if (lane == 63) ws[w] = incl;
      __syncthreads();
      int before = carry;
      for (int i = 0; i < w; ++i) before += ws[i];
      if (key < nkeys) kb[key] = before + incl - v;
      carry += ws[0] + ws[1] + ws[2] + ws[3];
    }
    __syncthreads();
    if (nwb <= PLAN_SCAN_CHUNKS) {
      for (int key = gtid; key < nkeys; key += wthreads) {
        int* row = j.cs_tab + (size_t)key * nwb;
        int v[PLAN_SCAN_CHUNKS];
#pragma unroll
        for (int q = 0; q < PLAN_SCAN_CHUNKS; ++q) v[q] = q < nwb ? pld<COH>(row + q) : 0;
        int run = kb[key];
#pragma unroll
        for (int q = 0; q < PLAN_SCAN_CHUNKS; ++q) {
          if (q < nwb) pst<COH>(row + q, run);
          run += v[q];
        }
      }
    } else {
      for (int key = gw; key < nkeys; key += wwaves) {
        int* row = j.cs_tab + (size_t)key * nwb;
        int v[PLAN_SCAN_CHUNKS];                 // all loads in flight before the first scan
#pragma unroll
        for (int q = 0; q < PLAN_SCAN_CHUNKS; ++q) {
          const int i = q * 64 + lane;
          v[q] = i < nwb ? pld<COH>(row + i) : 0;
        }
        int run = kb[key];
#pragma unroll
        for (int q = 0; q < PLAN_SCAN_CHUNKS; ++q) {
          if (q * 64 < nwb) {
            const int i = q * 64 + lane;
            int incl = v[q];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int up = __shfl_up(incl, o);
              if (lane >= o) incl += up;
            }
            if (i < nwb) pst<COH>(row + i, run + incl - v[q]);
            run += __shfl(incl, 63);
          }
        }
      }
    }
  }
  plan_sync(j.sync + 4, j.sync_base, P, j.fail);
  stamp(4);
  // ---- stage 5: scatter: one wave per row block ranks its rows in row order (equal keys keep their
  // row order: nothing depends on timing)
  {
    int* cnt = lds + w * PLAN_KEYS;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int key_bits = 1;
    while ((1 << key_bits) < nkeys) ++key_bits;
    for (int wb = gw; wb < nwb; wb += wwaves) {
      for (int s = lane; s < nkeys; s += 64) cnt[s] = pld<COH>(j.cs_tab + (size_t)s * nwb + wb);
      const int r0 = wb << j.cs_shift;
      for (int it = 0; it < rounds; ++it) {
        const int r = r0 + it * 64 + lane;
        const bool ok = r < n;
        const int k = ok ? min((int)pld<COH>(j.sort_key + r), nkeys - 1) : -1;
        // lanes with the same key: one ballot per key bit instead of one round per distinct key
        unsigned long long same = __ballot(ok);
        for (int bit = 0; bit < key_bits; ++bit) {
          const unsigned long long m = __ballot((k >> bit) & 1);
          same &= ((k >> bit) & 1) ? m : ~m;
        }
        int pos = 0;
        if (ok) pos = cnt[k] + __popcll(same & lt);
        // (one wave: its LDS accesses execute in order -- every lane has read before the leaders write)
        if (ok && (same & lt) == 0) cnt[k] += __popcll(same);
        if (ok) j.order[pos] = r;
      }
    }
  }
  stamp(5);
}
__device__ inline void plan_grid_body(const PlanJob& j, int pb, int* lds) {
  if (j.P > 1) plan_grid_stages<true>(j, pb, lds);
  else plan_grid_stages<false>(j, pb, lds);
}

}  // namespace njode
