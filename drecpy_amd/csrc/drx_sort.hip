// Stable LSD radix sort of (uint32 key, uint32 val) pairs on the low `bits` bits of the key — the inverted-index builder of the
// sparse steps (touched row -> samples, in sample order, so that every row's gradient is summed in a fixed order).
//
// Hand-written for gfx950 in the "onesweep" form (one launch per digit instead of histogram + scan + scatter launches):
//   k_digit_histograms   one read of the keys -> the global histogram of EVERY digit position (LDS atomics, then one integer
//                        atomic per non-empty bin and workgroup)
//   k_onesweep_pass<R>   one launch per digit.  A tile of 4096 pairs per workgroup; tiles take their number from an atomic ticket
//                        (so every predecessor of a running tile is itself running or done).  Per tile:
//                          1. every wave ranks its 64-item rows in order: lanes holding the same digit find each other with R
//                             ballots, the lowest of them bumps the wave's private counter of that digit in LDS and hands the old
//                             value to the others (no LDS atomics, stable by construction);
//                          2. the per-wave counters are scanned over the waves -> tile histogram -> published as one 32-bit word
//                             per (tile, digit): 2 status bits | 30 count bits, a relaxed agent-scope store (one word: nothing to
//                             order against; the XCDs' L2s are not coherent, so these words go around them: sc1);
//                          3. decoupled look-back: the thread of a digit sums its predecessors' words, eight loads in flight, until
//                             it meets one that already holds an inclusive prefix, then publishes its own inclusive prefix;
//                          4. the tile is laid out sorted in LDS and leaves in runs of equal digits (coalesced stores).
// Digit width: 8 bits (256 bins) by default; key widths a 10- or 11-bit digit covers in fewer launches use those (a 20-bit key — item
// rows of a 1M-item catalogue — is 2 launches of 10 bits instead of 3 of 8).
// All scratch (ping-pong buffers, histograms, tile words, tickets) comes from the caller's `temp`; nothing is allocated here.
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include "drx_common.hpp"

namespace drx {

#ifndef DRX_SORT_THREADS
#define DRX_SORT_THREADS 512
#endif
#ifndef DRX_SORT_IPT
#define DRX_SORT_IPT 16          // r06: tiles of 8192 pairs (8 through r05).  Alone the sort takes the same 78 - 79 us for 1.18 M pairs; beside the
#endif                           // training kernels the step ran 0.3333 - 0.3372 ms against 0.3385 - 0.3391 (tiles of 2048: 0.350 - 0.354)
                                 // [profiles/r06_sort_tile_variants_beside_training.log]: half as many workgroups asking for room on the CUs
constexpr int kSortThreads = DRX_SORT_THREADS;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortIPT = DRX_SORT_IPT;            // items per thread
constexpr int kSortTile = kSortThreads * kSortIPT;
constexpr int kMaxPasses = 4;
constexpr uint32_t kStAgg = 1u << 30, kStIncl = 2u << 30, kCountMask = (1u << 30) - 1;
constexpr int kDropNone = 1, kFirstPass = 2, kLastPass = 4;

struct SortPlan {
  int passes;
  int rbits[kMaxPasses];       // digit width of each pass
  int shift[kMaxPasses];
};

// fewest launches with digits of at most 11 bits; equal-width digits
inline SortPlan sort_plan(int bits) {
  SortPlan p{};
  int passes = (bits + 10) / 11;
  if (bits <= 8) passes = 1;
  int r = (bits + passes - 1) / passes;
  if (r < 8 && bits >= 8) r = 8;
  if (r <= 8) r = 8; else if (r <= 10) r = 10; else r = 11;
  p.passes = (bits + r - 1) / r;
  for (int i = 0; i < p.passes; ++i) { p.rbits[i] = r; p.shift[i] = i * r; }
  return p;
}

struct SortLayout {
  uint32_t *tmp_k, *tmp_v;     // ping-pong partner of the output buffers (passes >= 2)
  uint32_t *hist;              // [passes][2048] global digit histograms
  uint32_t *ticket;            // [passes]
  uint32_t *desc;              // [passes][tiles][radix]
  size_t zero_begin, zero_bytes;
  size_t total;
  int tiles;
};

inline SortLayout sort_layout(void *temp, size_t n, const SortPlan &P) {
  SortLayout L{};
  Carver cv(temp, (size_t)-1);
  L.tiles = (int)((n + kSortTile - 1) / kSortTile);
  if (P.passes >= 2) { L.tmp_k = cv.take<uint32_t>(n); L.tmp_v = cv.take<uint32_t>(n); }
  cv.off = align_up(cv.off, 256);
  L.zero_begin = cv.off;
  L.hist = cv.take<uint32_t>((size_t)kMaxPasses * 2048);
  L.ticket = cv.take<uint32_t>(64);
  size_t d = 0;
  for (int i = 0; i < P.passes; ++i) d += (size_t)L.tiles << P.rbits[i];
  L.desc = cv.take<uint32_t>(d);
  L.zero_bytes = cv.off - L.zero_begin;
  L.total = align_up(cv.off, 256);
  return L;
}

// Every digit histogram of the sort in one pass over the keys, LDS counters, then one global add per non-empty bin.  The touch
// lists this sorts are skewed — the W keys' top digit takes a dozen values — and one LDS atomic per key on so few addresses serialises
// lane by lane; each digit's 2048 words hold REP = 2048 >> rbits REPLICAS of its histogram, a lane adds to replica lane % REP (8-bit
// digits: 8 replicas, an eighth of the conflicts; r03: 45 - 60 us of the preparation's stream).
__global__ __launch_bounds__(512) void k_digit_histograms(const uint32_t *__restrict__ keys, size_t n, SortPlan P, uint32_t *__restrict__ hist,
                                                          int drop_none, SortRider rider, int rider_blocks) {
  __shared__ uint32_t h[kMaxPasses * 2048];
  if ((int)blockIdx.x >= (int)gridDim.x - rider_blocks) {          // (the caller's extra work: see SortRider)
    degree_counts_body<512>(rider.keep_off, rider.B, rider.counts, (int)blockIdx.x - ((int)gridDim.x - rider_blocks), h);
    return;
  }
  const size_t hist_threads = (size_t)((int)gridDim.x - rider_blocks) * blockDim.x;
  // (P.passes here = the digit positions to count: only the first when every pass counts its successor's digits itself)
  for (int i = threadIdx.x; i < P.passes * 2048; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += hist_threads) {
    const uint32_t k = keys[i];
    if (drop_none && k == DRX_KEY_NONE) continue;
    for (int p = 0; p < P.passes; ++p) {
      const uint32_t rep_bits = 11u - (uint32_t)P.rbits[p];
      const uint32_t d = (k >> P.shift[p]) & ((1u << P.rbits[p]) - 1);
      atomicAdd(&h[p * 2048 + (d << rep_bits) + (lane & ((1u << rep_bits) - 1u))], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < P.passes; ++p) {
    const uint32_t rep_bits = 11u - (uint32_t)P.rbits[p], reps = 1u << rep_bits;
    for (uint32_t d = threadIdx.x; d < (1u << P.rbits[p]); d += blockDim.x) {
      uint32_t c = 0;
      for (uint32_t r = 0; r < reps; ++r) c += h[p * 2048 + (d << rep_bits) + r];
      if (c) atomicAdd(&hist[p * 2048 + d], c);
    }
  }
}

template <int R>
__global__ __launch_bounds__(kSortThreads) void k_onesweep_pass(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                               uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                               const uint32_t *__restrict__ ghist, uint32_t *ticket, uint32_t *desc, uint32_t n_tiles,
                                                               int flags, uint32_t *next_hist, int next_shift) {
  constexpr int RADIX = 1 << R;
  constexpr int BPT = (RADIX + kSortThreads - 1) / kSortThreads;      // bins per thread
  extern __shared__ __align__(16) uint32_t sort_lds[];                // onesweep_lds_bytes(R): 42 / 72 / 112 KB for R = 8 / 10 / 11
  uint32_t (*whist)[RADIX] = reinterpret_cast<uint32_t (*)[RADIX]>(sort_lds);   // [waves][RADIX] per-wave digit counters, then offsets
  uint32_t *gbase = sort_lds + kSortWaves * RADIX;  // where LDS slot i of digit d goes: gbase[d] + i
  uint32_t *lstart = gbase + RADIX;                 // start of digit d's run inside the tile
  uint32_t *skey = lstart + RADIX, *sval = skey + kSortTile;
  // next_hist (or nullptr): this pass also counts the NEXT pass's digits of the pairs it moves — 2048 LDS words per workgroup, REP =
  // 2048 >> R replicas of the histogram (a lane adds to replica lane % REP: skewed digits would otherwise serialise), folded into the
  // global histogram once, when the workgroup has run out of tiles.  r03: the one kernel that counted every digit position up front
  // took 52 us of the preparation's stream beside the training kernels; it now counts the first position only.
  constexpr int REPB = 11 - R, REP = 1 << REPB;
  uint32_t *nh = sval + kSortTile;
  __shared__ uint32_t s_tile, s_tile_valid;
  __shared__ uint32_t wl[kSortWaves], wg[kSortWaves];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // kDropNone: pairs whose key is DRX_KEY_NONE (padding of a touch list) are left out by the first pass — the later passes then
  // move only what is left — and the last pass writes the padding back behind the sorted pairs (keys only).
  const bool drop = flags & kDropNone, first = flags & kFirstPass, last = flags & kLastPass;
  size_t n_valid = n;
  if (drop) {                                       // what the histograms counted = the pairs that are not padding
    uint32_t part = 0;
    for (int d = tid; d < RADIX; d += kSortThreads) part += ghist[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) wl[w] = part;
    __syncthreads();
    uint32_t tot = 0;
    for (int ww = 0; ww < kSortWaves; ++ww) tot += wl[ww];
    n_valid = tot;
    __syncthreads();
  }
  const size_t n_in = (drop && !first) ? n_valid : n;
  n_tiles = (uint32_t)((n_in + kSortTile - 1) / kSortTile);
  if (next_hist) {
    for (int i = tid; i < 2048; i += kSortThreads) nh[i] = 0;      // (the first tile's barriers order this before the first add)
  }
  // a workgroup takes tiles by ticket until none is left: a grid much smaller than the number of tiles keeps the sort's footprint on
  // the chip small (it runs beside the training kernels), and a tile's predecessors are then mostly finished when it looks back
  for (;;) {
  __syncthreads();                                  // (the previous tile's readers of LDS are done)
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);
  for (int i = tid; i < kSortWaves * RADIX; i += kSortThreads) sort_lds[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  if (tile >= n_tiles) break;
  const size_t base = (size_t)tile * kSortTile;
  const int tile_n = (int)((n_in - base) < (size_t)kSortTile ? (n_in - base) : (size_t)kSortTile);

  // ---- 1. load + rank (wave w owns items [w*512, w*512+512) of the tile, row `it` = 64 consecutive items) ----
  uint32_t key[kSortIPT], val[kSortIPT], pre[kSortIPT];
#pragma unroll
  for (int it = 0; it < kSortIPT; ++it) {
    const int idx = w * (64 * kSortIPT) + it * 64 + lane;
    key[it] = idx < tile_n ? kin[base + idx] : 0xFFFFFFFFu;
    val[it] = idx < tile_n ? vin[base + idx] : 0u;
  }
#pragma unroll
  for (int it = 0; it < kSortIPT; ++it) {
    const int idx = w * (64 * kSortIPT) + it * 64 + lane;
    const bool valid = idx < tile_n && !(drop && key[it] == DRX_KEY_NONE);
    const uint32_t d = (key[it] >> shift) & (RADIX - 1);
    if (next_hist && valid) atomicAdd(&nh[(((key[it] >> next_shift) & (RADIX - 1)) << REPB) + (lane & (REP - 1))], 1u);
    uint64_t mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < R; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bl = __ballot(bit);
      mask &= bit ? bl : ~bl;
    }
    const uint32_t below = __popcll(mask & ((1ull << lane) - 1ull));
    uint32_t old = 0;
    if (valid && below == 0) { old = whist[w][d]; whist[w][d] = old + (uint32_t)__popcll(mask); }
    wave_lds_sync();      // the leader's counter update of row `it` is read by (another lane as) the leader of row `it + 1`
    const int leader = valid ? (__ffsll((long long)mask) - 1) : lane;
    old = __shfl(old, leader);
    pre[it] = old + below;
  }
  __syncthreads();

  // ---- 2. per-wave counters -> per-wave offsets, tile histogram, publish ----
  uint32_t cnt[BPT], excl[BPT];
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int d = tid + q * kSortThreads;
    cnt[q] = 0;
    if (d < RADIX) {
      uint32_t run = 0;
#pragma unroll
      for (int ww = 0; ww < kSortWaves; ++ww) { const uint32_t c = whist[ww][d]; whist[ww][d] = run; run += c; }
      cnt[q] = run;
      uint32_t *slot = desc + ((size_t)tile << R) + d;
      __hip_atomic_store(slot, (tile == 0 ? kStIncl : kStAgg) | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // exclusive scans over the digits: of the tile histogram (lstart) and of the global histogram (gstart); RADIX <= 2048, 512 threads
  {
    uint32_t local[BPT], glob[BPT];
    uint32_t sl = 0, sg = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {                         // thread t owns digits t*BPT .. t*BPT+BPT-1 for the scans
      const int d = tid * BPT + q;
      local[q] = 0; glob[q] = 0;
      if (d < RADIX) glob[q] = ghist[d];
      sg += glob[q];
    }
    // the tile counts live in registers of the thread that owns digit (tid + q*threads): pass them through LDS (lstart as staging)
#pragma unroll
    for (int q = 0; q < BPT; ++q) { const int d = tid + q * kSortThreads; if (d < RADIX) lstart[d] = cnt[q]; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BPT; ++q) { const int d = tid * BPT + q; if (d < RADIX) local[q] = lstart[d]; sl += local[q]; }
    __syncthreads();
    // block-wide exclusive scan of (sl, sg): wave scan + scan of the wave totals
    uint32_t il = sl, ig = sg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t tl = __shfl_up(il, o), tg = __shfl_up(ig, o);
      if (lane >= o) { il += tl; ig += tg; }
    }
    if (lane == 63) { wl[w] = il; wg[w] = ig; }
    __syncthreads();
    uint32_t ol = il - sl, og = ig - sg;
    for (int ww = 0; ww < w; ++ww) { ol += wl[ww]; og += wg[ww]; }
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const int d = tid * BPT + q;
      if (d < RADIX) { lstart[d] = ol; gbase[d] = og; }
      ol += local[q]; og += glob[q];
      if (d == RADIX - 1) s_tile_valid = ol;              // pairs of this tile that take part
    }
  }
  __syncthreads();

  // ---- 3. decoupled look-back over the predecessors' words (eight in flight) ----
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int d = tid + q * kSortThreads;
    excl[q] = 0;
    if (d < RADIX && tile > 0) {
      uint32_t sum = 0;
      int p = (int)tile - 1;
      bool done = false;
      while (!done) {
        uint32_t wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          wv[u] = (p - u >= 0) ? __hip_atomic_load(desc + ((size_t)(p - u) << R) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStIncl;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (done) break;
          uint32_t x = wv[u];
          while ((x >> 30) == 0) {                   // predecessor has not published yet (it holds an earlier ticket: it is running)
            __builtin_amdgcn_s_sleep(1);
            x = __hip_atomic_load(desc + ((size_t)(p - u) << R) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          sum += x & kCountMask;
          if ((x >> 30) == 2u) done = true;
        }
        p -= 8;
      }
      excl[q] = sum;
      __hip_atomic_store(desc + ((size_t)tile << R) + d, kStIncl | (sum + cnt[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // gbase[d] += tile's exclusive prefix - start of the digit's run in the tile
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int d = tid + q * kSortThreads;
    if (d < RADIX) gbase[d] = gbase[d] + excl[q] - lstart[d];
  }
  // ---- 4. sorted tile in LDS, then out in runs ----
#pragma unroll
  for (int it = 0; it < kSortIPT; ++it) {
    const int idx = w * (64 * kSortIPT) + it * 64 + lane;
    if (idx < tile_n && !(drop && key[it] == DRX_KEY_NONE)) {
      const uint32_t d = (key[it] >> shift) & (RADIX - 1);
      const uint32_t pos = lstart[d] + whist[w][d] + pre[it];
      skey[pos] = key[it];
      sval[pos] = val[it];
    }
  }
  __syncthreads();
  const int tile_out = (int)s_tile_valid;
  for (int i = tid; i < tile_out; i += kSortThreads) {
    const uint32_t k = skey[i];
    const uint32_t d = (k >> shift) & (RADIX - 1);
    const size_t o = (size_t)gbase[d] + (size_t)i;
    kout[o] = k;
    vout[o] = sval[i];
  }
  }   // tiles of this workgroup
  if (next_hist) {                                    // (the loop's exit came right after a barrier: every add of this workgroup is visible)
    for (int d = tid; d < RADIX; d += kSortThreads) {
      uint32_t c = 0;
#pragma unroll
      for (int r = 0; r < REP; ++r) c += nh[(d << REPB) + r];
      if (c) atomicAdd(&next_hist[d], c);
    }
  }
  if (drop && last)
    for (size_t i = n_valid + (size_t)blockIdx.x * kSortThreads + tid; i < n; i += (size_t)gridDim.x * kSortThreads) kout[i] = DRX_KEY_NONE;
}

// workgroups per launch (each loops over tiles; -DDRX_SORT_GRID=n builds a variant).  r02 kept the grid at 128 so that the sort
// stayed out of the training kernels' way; with the r03 training kernels the PREPARATION is what bounds the pipeline and the
// measurement turned around (r03m, step time at 64 / 128 / 192 / 256 / 352 workgroups: 0.424 / 0.394 / 0.383 / 0.380 / 0.375 ms;
// 352 = one workgroup per tile of the 1.44 M-pair list): 512, i.e. a workgroup per tile up to 2 M pairs.
#ifndef DRX_SORT_GRID
#define DRX_SORT_GRID 512
#endif
inline int sort_grid() { return DRX_SORT_GRID; }

inline size_t onesweep_lds_bytes(int r) { return ((size_t)(kSortWaves + 2) * ((size_t)1 << r) + 2 * (size_t)kSortTile + 2048) * 4; }

template <int R>
static int launch_pass(const uint32_t *sk, const uint32_t *sv, uint32_t *dk, uint32_t *dv, size_t n, int shift, const uint32_t *gh, uint32_t *ticket,
                       uint32_t *desc, int tiles, int flags, uint32_t *next_hist, int next_shift, hipStream_t stream) {
  const size_t lds = onesweep_lds_bytes(R);
  if (lds > 48 * 1024) DRX_HIP(hipFuncSetAttribute((const void *)k_onesweep_pass<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_onesweep_pass<R>, dim3(tiles < sort_grid() ? tiles : sort_grid()), dim3(kSortThreads), lds, stream, sk, sv, dk, dv, n, shift, gh,
                     ticket, desc, (uint32_t)tiles, flags, next_hist, next_shift);
  return 0;
}

size_t sort_pairs_temp_bytes(size_t n, int end_bit) {
  const SortPlan P = sort_plan(end_bit);
  return sort_layout(nullptr, n ? n : 1, P).total + 256;
}

void sort_pairs_zero_region(void *temp, size_t n, int end_bit, uint32_t **words, size_t *n_words) {
  const SortPlan P = sort_plan(end_bit);
  char *t = (char *)align_up((size_t)temp, 256);
  const SortLayout L = sort_layout(t, n ? n : 1, P);
  *words = (uint32_t *)(t + L.zero_begin);
  *n_words = L.zero_bytes / 4;
}

int sort_pairs_ex(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int end_bit,
                  bool drop_none, hipStream_t stream, bool pre_zeroed, SortRider rider) {
  if (n == 0) return 0;
  if (n >= (1u << 30)) return DRX_EINVAL;                      // tile words carry 30-bit counts
  const SortPlan P = sort_plan(end_bit);
  char *t = (char *)align_up((size_t)temp, 256);
  const SortLayout L = sort_layout(t, n, P);
  if ((size_t)(t - (char *)temp) + L.total > temp_bytes) return DRX_ESCRATCH;
  if (!pre_zeroed) DRX_HIP(hipMemsetAsync(t + L.zero_begin, 0, L.zero_bytes, stream));
  int hgrid = (int)((n + 8 * 512 - 1) / (8 * 512));
  if (hgrid > sort_grid()) hgrid = sort_grid();
  SortPlan P1 = P;
  P1.passes = 1;                                               // every pass counts its successor's digits while it moves the pairs
  const int rider_blocks = rider.keep_off ? (rider.B + 511) / 512 : 0;
  hipLaunchKernelGGL(k_digit_histograms, dim3(hgrid + rider_blocks), dim3(512), 0, stream, kin, n, P1, L.hist, drop_none ? 1 : 0, rider,
                     rider_blocks);
  const uint32_t *src_k = kin, *src_v = vin;
  uint32_t *desc = L.desc;
  for (int p = 0; p < P.passes; ++p) {
    const bool to_out = ((P.passes - 1 - p) & 1) == 0;         // the last pass lands in the caller's output buffers
    uint32_t *dk = to_out ? kout : L.tmp_k, *dv = to_out ? vout : L.tmp_v;
    const uint32_t *gh = L.hist + (size_t)p * 2048;
    const int flags = (drop_none ? kDropNone : 0) | (p == 0 ? kFirstPass : 0) | (p == P.passes - 1 ? kLastPass : 0);
    int rc;
    uint32_t *nh = p + 1 < P.passes ? L.hist + (size_t)(p + 1) * 2048 : nullptr;
    const int ns = p + 1 < P.passes ? P.shift[p + 1] : 0;
    if (P.rbits[p] == 8) rc = launch_pass<8>(src_k, src_v, dk, dv, n, P.shift[p], gh, L.ticket + p, desc, L.tiles, flags, nh, ns, stream);
    else if (P.rbits[p] == 10) rc = launch_pass<10>(src_k, src_v, dk, dv, n, P.shift[p], gh, L.ticket + p, desc, L.tiles, flags, nh, ns, stream);
    else rc = launch_pass<11>(src_k, src_v, dk, dv, n, P.shift[p], gh, L.ticket + p, desc, L.tiles, flags, nh, ns, stream);
    if (rc) return rc;
    desc += (size_t)L.tiles << P.rbits[p];
    src_k = dk; src_v = dv;
  }
  DRX_LAUNCH_CHECK();
  return 0;
}

int sort_pairs(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
               uint32_t *vout, size_t n, int end_bit, hipStream_t stream) {
  return sort_pairs_ex(temp, temp_bytes, kin, kout, vin, vout, n, end_bit, false, stream);
}

}  // namespace drx

// C ABI (include/drx.h): the sort as a utility of its own
extern "C" {
size_t drx_sort_pairs_temp_bytes(int64_t n, int32_t key_bits) {
  if (n < 0 || key_bits < 1 || key_bits > 32) return 0;
  return drx::sort_pairs_temp_bytes((size_t)n, key_bits) + 256;
}
int drx_sort_pairs(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int32_t key_bits,
                   void *temp, size_t temp_bytes, void *stream) {
  if (n < 0 || key_bits < 1 || key_bits > 32 || (n > 0 && (!keys_in || !keys_out || !vals_in || !vals_out || !temp))) return DRX_EINVAL;
  return drx::sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, key_bits, (hipStream_t)stream);
}
}
