// Host-side (CPU) pieces of libdrx.so: CPython-exact MT19937 streams, the PointSampler algorithm and the CDAE
// corruption stream.  These replace the O(nnz)-per-draw pandas scans of DRecPy/Dataset/mem_dataset.py:111-163
// and the N-long Python list comprehensions of DRecPy/Recommender/cdae.py:61-63 while producing bit-identical
// streams (checked against stdlib `random.Random` and the golden vectors generated from the reference).
#include <sched.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <vector>
#include "drx.h"

namespace {

// MT19937 exactly as CPython's Modules/_randommodule.c drives it (algorithm of Matsumoto & Nishimura).

// One MT19937 state transition (624 words in place).  The corruption stream of reference mode is N uniform draws per batch row
// (cdae.py:63), almost all of them skipped over: this loop IS that cost, so it is compiled for the widest vectors the host has
// (runtime dispatch: the library is built on one machine and runs on another).
__attribute__((target_clones("avx512f", "avx2", "default"))) void mt_twist(uint32_t *__restrict__ s) {
  // word kk needs words kk + 1 (old) and kk + 397 mod 624 (old for kk < 227, else written 227 iterations earlier): in the three
  // ranges below no iteration depends on another of the same range
  auto range = [&](int lo, int hi, int off) {
#pragma GCC ivdep
    for (int kk = lo; kk < hi; ++kk) {
      const uint32_t y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7FFFFFFFu);
      s[kk] = s[kk + off] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908B0DFu);
    }
  };
  range(0, 227, 397);
  range(227, 454, -227);
  range(454, 623, -227);
  const uint32_t y = (s[623] & 0x80000000u) | (s[0] & 0x7FFFFFFFu);
  s[623] = s[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908B0DFu);
}

struct MT {
  uint32_t s[624];
  int idx;
  void init_genrand(uint32_t seed) {
    s[0] = seed;
    for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void init_by_array(const uint32_t *key, size_t len) {
    init_genrand(19650218u);
    size_t i = 1, j = 0;
    size_t k = 624 > len ? 624 : len;
    for (; k; --k) {
      s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      ++i; ++j;
      if (i >= 624) { s[0] = s[623]; i = 1; }
      if (j >= len) j = 0;
    }
    for (k = 623; k; --k) {
      s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      ++i;
      if (i >= 624) { s[0] = s[623]; i = 1; }
    }
    s[0] = 0x80000000u;
  }
  // random.Random(seed) with an int seed: key = 32-bit little-endian digits of abs(seed) (>= 1 digit)
  void seed_int(int64_t seed) {
    uint64_t a = seed < 0 ? (uint64_t)(-(seed + 1)) + 1u : (uint64_t)seed;
    uint32_t key[2] = {(uint32_t)(a & 0xFFFFFFFFu), (uint32_t)(a >> 32)};
    init_by_array(key, key[1] ? 2 : 1);
  }
  // discards k outputs: the state is regenerated block by block as usual, but nothing is tempered
  void skip(uint64_t k) {
    while (k) {
      if (idx >= 624) regenerate();
      const uint64_t step = k < (uint64_t)(624 - idx) ? k : (uint64_t)(624 - idx);
      idx += (int)step;
      k -= step;
    }
  }
  void regenerate() {
    mt_twist(s);
    idx = 0;
  }
  uint32_t next() {
    if (idx >= 624) regenerate();
    uint32_t y = s[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
  }
  double random() {   // 53-bit construction of random_random()
    uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  uint64_t getrandbits(int k) {   // k in 1..64; words are produced least-significant first
    if (k <= 32) return next() >> (32 - k);
    uint64_t lo = next();
    uint64_t hi = next() >> (64 - k);
    return (hi << 32) | lo;
  }
  uint64_t randbelow(uint64_t n) {   // Random._randbelow_with_getrandbits
    if (n == 0) return 0;
    int k = 64 - __builtin_clzll(n);
    uint64_t r = getrandbits(k);
    while (r >= n) r = getrandbits(k);
    return r;
  }
  int64_t randint(int64_t a, int64_t b) { return a + (int64_t)randbelow((uint64_t)(b - a + 1)); }
  double uniform(double a, double b) { return a + (b - a) * random(); }
};

}  // namespace

struct DrxRng {
  MT mt;
};

struct DrxSampler {
  MT r_sel, r_neg, r_pos;                 // point_sampler.py:32, mem_dataset.py:136, mem_dataset.py:115
  int32_t neg_ratio;
  int32_t max_uid, max_iid;
  // which (uid, iid) pairs occur in ANY dataframe row (sample_negative rejects them, mem_dataset.py:154-163): the items of every
  // user, ascending (a draw probes ~8 entries of one short row instead of 20 of a sorted list of all pairs: 226 -> 70 ns per
  // negative at ml-1m), and — when users x items is at most 2^28 — one bit per pair (a single probe)
  std::vector<int64_t> pair_ptr;
  std::vector<int32_t> pair_items;
  std::vector<uint64_t> pair_bits;
  bool has_pair(int64_t u, int64_t i) const {
    if (!pair_bits.empty()) {
      const uint64_t k = (uint64_t)u * (uint64_t)(max_iid + 1) + (uint64_t)i;
      return (pair_bits[k >> 6] >> (k & 63)) & 1u;
    }
    return std::binary_search(pair_items.begin() + pair_ptr[(size_t)u], pair_items.begin() + pair_ptr[(size_t)u + 1], (int32_t)i);
  }
  std::vector<int64_t> pos_ptr;           // per-user CSR over rows eligible as positives (dataframe order)
  std::vector<int64_t> pos_rows;
  std::vector<int32_t> uid, iid;
  std::vector<double> val;
};

// ---- ListSampler (DRecPy/Sampler/list_sampler.py:76-151) -------------------------------------------------------------------
// One draw: rng.choice(unique_groups) -> the group's positive rows (already threshold-filtered and ordered) -> optional
// window start rng.randint -> inputs / targets -> eligible negative ids = tuple(unique_negative_ids - ids of the rows), in
// CPython's set iteration order -> rng.sample.  Ids are dense small ints whose hash is the id itself, so the order of the
// eligible tuple is reproduced by following Objects/setobject.c (3.7 .. 3.12): see eligible_order().
namespace {

constexpr int kLinearProbes = 9, kPerturbShift = 5;

struct PySetOfInts {            // a CPython set holding non-negative ints (hash(i) == i): open addressing, linear probes, perturb
  std::vector<int64_t> tab;    // -1 = unused slot
  size_t mask, fill;
  PySetOfInts() : tab(8, -1), mask(7), fill(0) {}
  static void insert_clean(std::vector<int64_t> &t, size_t mask, int64_t key) {      // set_insert_clean
    size_t perturb = (size_t)key, i = (size_t)key & mask;
    for (;;) {
      if (t[i] < 0) { t[i] = key; return; }
      if (i + kLinearProbes <= mask)
        for (int j = 1; j <= kLinearProbes; ++j)
          if (t[i + j] < 0) { t[i + j] = key; return; }
      perturb >>= kPerturbShift;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }
  void resize(size_t minused) {                                                      // set_table_resize
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    std::vector<int64_t> nt(newsize, -1);
    for (int64_t k : tab)
      if (k >= 0) insert_clean(nt, newsize - 1, k);
    tab.swap(nt);
    mask = newsize - 1;
  }
  void add(int64_t key) {                                                            // set_add_entry (key known to be absent)
    size_t perturb = (size_t)key, i = (size_t)key & mask;
    for (;;) {
      const int probes = (i + kLinearProbes <= mask) ? kLinearProbes : 0;
      bool placed = false;
      for (int j = 0; j <= probes; ++j)
        if (tab[i + j] < 0) { tab[i + j] = key; placed = true; break; }
      if (placed) break;
      perturb >>= kPerturbShift;
      i = (i * 5 + 1 + perturb) & mask;
    }
    ++fill;
    if (fill * 5 >= mask * 3) resize(fill > 50000 ? fill * 2 : fill * 4);
  }
};

struct ListGroup {
  int64_t begin, end;                 // rows [begin, end) in the flattened row arrays
  std::vector<int32_t> held;          // ascending distinct negative-ids among the group's rows
  std::vector<int32_t> gap;           // held[i] - i: how many eligible ids lie below held id i (complement_at searches it)
  std::vector<int32_t> order;         // eligible ids in set order, only when that order is not simply ascending
  bool order_built = false;
};

}  // namespace

// A few persistent helper threads for the second phase of drx_list_sampler_sample (creating threads per call cost more than they
// saved: 0.1 - 0.2 ms on a 256-core host).  run(n_parts, fn): fn(part) for part = 0 .. n_parts - 1, part 0 on the calling thread.
struct PartPool {
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  const std::function<void(int)> *job = nullptr;
  uint64_t generation = 0;
  int pending = 0;
  bool stop = false;
  explicit PartPool(int n_workers) {
    for (int w = 0; w < n_workers; ++w)
      workers.emplace_back([this, w] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)> *j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            j = job;
          }
          (*j)(w + 1);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_one();
          }
        }
      });
  }
  void run(const std::function<void(int)> &fn) {          // parts 0 .. workers.size()
    {
      std::lock_guard<std::mutex> lk(mu);
      job = &fn;
      pending = (int)workers.size();
      ++generation;
    }
    cv_go.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  ~PartPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_go.notify_all();
    for (auto &t : workers) t.join();
  }
};

struct DrxListSampler {
  MT rng;
  int32_t n_groups, n_ids, neg_ratio, n_targets, min_pos, max_pos;     // n_targets / max_pos < 0: None
  std::vector<int64_t> rows;          // dataset row numbers, group by group
  std::vector<int32_t> ids;           // negative_ids_col value of each of those rows
  std::vector<ListGroup> groups;      // in unique_groups order
  int32_t last_hint = 0;
  std::unique_ptr<PartPool> pool;     // made by the first large batch
  long pool_pid = 0;                  // the process that made it: a forked child has the object without the threads

  // j-th (0-based) id of the ascending complement of `held` in [0, n_ids): the smallest v with v - #(held <= v) == j, i.e.
  // v = j + #(held ids h with h - rank(h) <= j).  `gap` = h - rank(h) per held id (non-decreasing), built with the group; the count is
  // a branch-free binary search (nine of them per window at examples/caser.py's 9 negatives: the sampler's stream is sequential, its
  // cost per draw is what bounds Caser.fit() on the reference-exact stream).
  static int32_t complement_at(const std::vector<int32_t> &gap, int64_t j) {
    const int32_t *base = gap.data();
    size_t n = gap.size();
    if (n == 0) return (int32_t)j;
    while (n > 1) {
      const size_t half = n >> 1;
      base = ((int64_t)base[half - 1] <= j) ? base + half : base;
      n -= half;
    }
    const size_t lo = (size_t)(base - gap.data()) + ((int64_t)base[0] <= j ? 1 : 0);
    return (int32_t)(j + (int64_t)lo);
  }
  int64_t eligible_count(const ListGroup &g) const { return (int64_t)n_ids - (int64_t)g.held.size(); }
  // tuple(unique_negative_ids.difference(set(held))) [j]
  int32_t eligible_at(ListGroup &g, int64_t j) {
    // set_difference (setobject.c): when len(so) >> 2 > len(other) the result is a COPY of so with the held ids discarded;
    // so's table (and the copy's, sized for 2 * len) is larger than every id, each id sits in slot == id: ascending order.
    if (((int64_t)n_ids >> 2) > (int64_t)g.held.size()) return complement_at(g.gap, j);
    if (!g.order_built) {                   // otherwise a NEW set receives the surviving ids one by one (ascending), growing
      PySetOfInts so;                       // as it fills: ids beyond the table size collide and the order is the table's
      const int64_t n = eligible_count(g);
      for (int64_t t = 0; t < n; ++t) so.add(complement_at(g.gap, t));
      g.order.reserve((size_t)n);
      for (int64_t k : so.tab)
        if (k >= 0) g.order.push_back((int32_t)k);
      g.order_built = true;
    }
    return g.order[(size_t)j];
  }
};

extern "C" {

DrxListSampler *drx_list_sampler_create(const int64_t *grp_indptr, const int64_t *grp_rows, const int32_t *grp_ids, int32_t n_groups,
                                        int32_t n_ids, int32_t neg_ratio, int32_t n_targets, int32_t min_positive,
                                        int32_t max_positive, int64_t seed) {
  if (!grp_indptr || n_groups < 1 || n_ids < 1 || neg_ratio < 0 || min_positive < 0) return nullptr;
  const int64_t nnz = grp_indptr[n_groups];
  if (nnz > 0 && (!grp_rows || !grp_ids)) return nullptr;
  DrxListSampler *s = new (std::nothrow) DrxListSampler;
  if (!s) return nullptr;
  s->rng.seed_int(seed);
  s->n_groups = n_groups; s->n_ids = n_ids; s->neg_ratio = neg_ratio; s->n_targets = n_targets;
  s->min_pos = min_positive; s->max_pos = max_positive;
  s->rows.assign(grp_rows, grp_rows + nnz);
  s->ids.assign(grp_ids, grp_ids + nnz);
  s->groups.resize((size_t)n_groups);
  for (int32_t g = 0; g < n_groups; ++g) {
    ListGroup &G = s->groups[(size_t)g];
    G.begin = grp_indptr[g]; G.end = grp_indptr[g + 1];
    G.held.assign(s->ids.begin() + G.begin, s->ids.begin() + G.end);
    std::sort(G.held.begin(), G.held.end());
    G.held.erase(std::unique(G.held.begin(), G.held.end()), G.held.end());
    if (!G.held.empty() && (G.held.front() < 0 || G.held.back() >= n_ids)) { delete s; return nullptr; }
    G.gap.resize(G.held.size());
    for (size_t i = 0; i < G.held.size(); ++i) G.gap[i] = G.held[i] - (int32_t)i;
  }
  return s;
}

void drx_list_sampler_destroy(DrxListSampler *s) {
  if (!s) return;
  // a forked child holds the parent's pool object without its threads: joining them would never return
  if (s->pool && s->pool_pid != (long)getpid()) (void)s->pool.release();
  delete s;
}

// Two phases.  (1) The draws in order, on the calling thread: everything that consumes random numbers — the group, the window's start, the
// k indices rng.sample picks in the eligible tuple (with its own rejections) — and the offsets; the picked INDICES are parked in neg_ids.
// (2) What consumes none, per draw and independent of the other draws: the indices turned into ids (a search in the group's gap
// array each) and the window's rows copied — on up to four threads when the batch is large.  The stream of random numbers is the
// reference's, one MT19937 consumed in order; only the work behind it leaves the sequential path (it was two thirds of a draw's cost,
// and the sampler's pace is what bounds Caser.fit() on the reference-exact stream).
int drx_list_sampler_sample(DrxListSampler *s, int32_t n, int32_t *group_out, int64_t *in_off, int64_t *in_rows, int64_t in_cap,
                            int64_t *tg_off, int64_t *tg_rows, int64_t tg_cap, int64_t *ng_off, int32_t *neg_ids, int64_t ng_cap) {
  if (!s || n < 0 || !group_out || !in_off || !in_rows || !tg_off || !ng_off) return DRX_EINVAL;
  const bool has_targets = s->n_targets >= 0;
  const int64_t T = has_targets ? s->n_targets : 0;
  in_off[0] = tg_off[0] = ng_off[0] = 0;
  std::vector<int32_t> pool;
  std::vector<int64_t> picked;
  struct Window { int64_t i0, t0; bool map_negatives; };
  std::vector<Window> win((size_t)n);
  int64_t setsize_k = -1, setsize = 21;
  for (int32_t d = 0; d < n; ++d) {
    int failures = 0;
    for (;;) {
      // one attempt (list_sampler.py:92-148); a failed attempt has consumed the draws it made
      const int32_t gi = (int32_t)s->rng.randbelow((uint64_t)s->n_groups);               // rng.choice(unique_groups)
      ListGroup &G = s->groups[(size_t)gi];
      const int64_t n_rows = G.end - G.begin;
      bool ok = !(n_rows < s->min_pos || n_rows < (int64_t)s->min_pos + T);
      if (!ok) s->last_hint = 1;
      int64_t start = -1;
      if (ok && s->max_pos >= 0 && n_rows > s->max_pos)
        start = s->rng.randint(0, n_rows - s->max_pos - T);                              // rng.randint(0, ...)
      int64_t i0 = 0, i1 = 0, t0 = 0, t1 = 0;
      if (ok) {
        if (!has_targets) {
          if (start < 0) { i0 = 0; i1 = n_rows; } else { i0 = start; i1 = start + s->max_pos; }
        } else if (start < 0) {               // reference quirk: without a window, inputs = first T rows, targets = the rest
          i0 = 0; i1 = std::min<int64_t>(T, n_rows); t0 = i1; t1 = n_rows;
        } else {
          i0 = start; i1 = start + s->max_pos; t0 = i1; t1 = std::min<int64_t>(i1 + T, n_rows);
        }
      }
      int64_t n_neg = 0;
      if (ok && has_targets) {
        n_neg = (int64_t)s->neg_ratio * (t1 - t0);
        if (s->eligible_count(G) < n_neg) { ok = false; s->last_hint = 2; }
      }
      if (!ok) {
        if (++failures > 20) return DRX_ERETRY;
        continue;
      }
      if (in_off[d] + (i1 - i0) > in_cap || (has_targets && (!tg_rows || tg_off[d] + (t1 - t0) > tg_cap)) ||
          (n_neg > 0 && (!neg_ids || ng_off[d] + n_neg > ng_cap)))
        return DRX_ESCRATCH;
      group_out[d] = gi;
      in_off[d + 1] = in_off[d] + (i1 - i0);
      tg_off[d + 1] = tg_off[d] + (t1 - t0);
      win[(size_t)d] = Window{i0, t0, false};
      if (n_neg > 0) {                                                                     // rng.sample(eligible, n_neg)
        const int64_t n_pop = s->eligible_count(G), k = n_neg;
        if (k != setsize_k) {                 // (random.sample's threshold, the same for every draw of a sampler: computed once)
          setsize_k = k;
          setsize = 21;
          if (k > 5) setsize += (int64_t)std::pow(4.0, std::ceil(std::log((double)(k * 3)) / std::log(4.0)));
        }
        int32_t *out = neg_ids + ng_off[d];
        // ids looked up in a set ORDER (dense groups: eligible_at builds and caches it) stay on this thread
        const bool pure = ((int64_t)s->n_ids >> 2) > (int64_t)G.held.size();
        if (n_pop <= setsize) {
          pool.resize((size_t)n_pop);
          for (int64_t j = 0; j < n_pop; ++j) pool[(size_t)j] = s->eligible_at(G, j);
          for (int64_t i = 0; i < k; ++i) {
            const int64_t j = (int64_t)s->rng.randbelow((uint64_t)(n_pop - i));
            out[i] = pool[(size_t)j];
            pool[(size_t)j] = pool[(size_t)(n_pop - i - 1)];
          }
        } else {
          picked.clear();
          for (int64_t i = 0; i < k; ++i) {
            int64_t j = (int64_t)s->rng.randbelow((uint64_t)n_pop);
            while (std::find(picked.begin(), picked.end(), j) != picked.end()) j = (int64_t)s->rng.randbelow((uint64_t)n_pop);
            picked.push_back(j);
            out[i] = pure ? (int32_t)j : s->eligible_at(G, j);
          }
          win[(size_t)d].map_negatives = pure;
        }
      }
      ng_off[d + 1] = ng_off[d] + n_neg;
      break;
    }
  }
  auto finish = [&](int32_t d0, int32_t d1) {
    for (int32_t d = d0; d < d1; ++d) {
      const ListGroup &G = s->groups[(size_t)group_out[d]];
      const Window &w = win[(size_t)d];
      if (w.map_negatives) {
        int32_t *out = neg_ids + ng_off[d];
        for (int64_t i = 0, k = ng_off[d + 1] - ng_off[d]; i < k; ++i) out[i] = DrxListSampler::complement_at(G.gap, out[i]);
      }
      const int64_t *src = s->rows.data() + G.begin;
      for (int64_t r = 0, m = in_off[d + 1] - in_off[d]; r < m; ++r) in_rows[in_off[d] + r] = src[w.i0 + r];
      for (int64_t r = 0, m = tg_off[d + 1] - tg_off[d]; r < m; ++r) tg_rows[tg_off[d] + r] = src[w.t0 + r];
    }
  };
  const unsigned hw = std::thread::hardware_concurrency();
  const int n_parts = n >= 2048 ? (int)std::min<unsigned>(4u, hw > 1 ? hw : 1u) : 1;
  if (n_parts <= 1) {
    finish(0, n);
  } else {
    if (s->pool && s->pool_pid != (long)getpid()) (void)s->pool.release();     // (inherited through fork(): its threads do not exist here; not joined)
    if (!s->pool) { s->pool.reset(new PartPool(n_parts - 1)); s->pool_pid = (long)getpid(); }
    const int np = (int)s->pool->workers.size() + 1;
    const std::function<void(int)> part = [&](int p) { finish((int32_t)((int64_t)n * p / np), (int32_t)((int64_t)n * (p + 1) / np)); };
    s->pool->run(part);
  }
  return DRX_OK;
}

int32_t drx_list_sampler_last_hint(const DrxListSampler *s) { return s ? s->last_hint : 0; }

DrxRng *drx_rng_create(int64_t seed) {
  DrxRng *r = new (std::nothrow) DrxRng;
  if (r) r->mt.seed_int(seed);
  return r;
}
void drx_rng_destroy(DrxRng *r) { delete r; }
double drx_rng_random(DrxRng *r) { return r->mt.random(); }
int64_t drx_rng_randint(DrxRng *r, int64_t a, int64_t b) { return r->mt.randint(a, b); }
void drx_rng_discard(DrxRng *r, uint64_t n_words) { if (r) r->mt.skip(n_words); }

int drx_rng_corruption_keep(DrxRng *r, const int64_t *h_indptr, const int32_t *h_indices, int32_t n_items,
                            const int32_t *h_uid, int32_t B, double q, int32_t *h_keep_off, uint8_t *h_keep,
                            int64_t keep_capacity) {
  if (!r || !h_indptr || !h_indices || !h_uid || !h_keep_off || !h_keep || B < 0) return DRX_EINVAL;
  int64_t off = 0;
  for (int32_t b = 0; b < B; ++b) {
    h_keep_off[b] = (int32_t)off;
    const int64_t s = h_indptr[h_uid[b]], e = h_indptr[h_uid[b] + 1];
    if (off + (e - s) > keep_capacity) return DRX_ESCRATCH;
    // cdae.py:63 draws one uniform(0,1) for EVERY item n = 0..N-1 of the row, positives or not; only the draws that fall on a
    // positive are looked at, the others just advance the stream (two 32-bit outputs per uniform)
    int32_t n = 0;
    for (int64_t j = s; j < e; ++j) {
      const int32_t col = h_indices[j];
      if (col < n || col >= n_items) return DRX_EINVAL;      // columns must ascend within a row
      r->mt.skip(2ull * (uint64_t)(col - n));
      const double x = r->mt.random();     // uniform(0, 1) == 0 + (1 - 0) * random()
      h_keep[off + (j - s)] = (x < q) ? 0 : 1;
      n = col + 1;
    }
    r->mt.skip(2ull * (uint64_t)(n_items - n));
    off += e - s;
  }
  h_keep_off[B] = (int32_t)off;
  return DRX_OK;
}

DrxSampler *drx_sampler_create(const int32_t *h_uid, const int32_t *h_iid, const double *h_val, int64_t n_rows,
                               int32_t neg_ratio, int32_t has_threshold, double threshold, int64_t seed) {
  if (!h_uid || !h_iid || !h_val || n_rows <= 0) return nullptr;
  DrxSampler *s = new (std::nothrow) DrxSampler;
  if (!s) return nullptr;
  s->r_sel.seed_int(seed); s->r_neg.seed_int(seed); s->r_pos.seed_int(seed);
  s->neg_ratio = neg_ratio;
  s->uid.assign(h_uid, h_uid + n_rows);
  s->iid.assign(h_iid, h_iid + n_rows);
  s->val.assign(h_val, h_val + n_rows);
  s->max_uid = *std::max_element(s->uid.begin(), s->uid.end());   // max over the WHOLE frame (mem_dataset.py:117,148)
  s->max_iid = *std::max_element(s->iid.begin(), s->iid.end());
  if (*std::min_element(s->uid.begin(), s->uid.end()) < 0 || *std::min_element(s->iid.begin(), s->iid.end()) < 0) { delete s; return nullptr; }
  const uint64_t cells = (uint64_t)(s->max_uid + 1) * (uint64_t)(s->max_iid + 1);
  if (cells <= (1ull << 28)) {
    s->pair_bits.assign((size_t)((cells + 63) >> 6), 0);
    for (int64_t r = 0; r < n_rows; ++r) {
      const uint64_t k = (uint64_t)h_uid[r] * (uint64_t)(s->max_iid + 1) + (uint64_t)h_iid[r];
      s->pair_bits[k >> 6] |= 1ull << (k & 63);
    }
  } else {
    s->pair_ptr.assign((size_t)s->max_uid + 2, 0);
    for (int64_t r = 0; r < n_rows; ++r) s->pair_ptr[(size_t)h_uid[r] + 1]++;
    for (size_t u = 0; u + 1 < s->pair_ptr.size(); ++u) s->pair_ptr[u + 1] += s->pair_ptr[u];
    s->pair_items.resize((size_t)n_rows);
    std::vector<int64_t> at(s->pair_ptr.begin(), s->pair_ptr.end() - 1);
    for (int64_t r = 0; r < n_rows; ++r) s->pair_items[(size_t)at[(size_t)h_uid[r]]++] = h_iid[r];
    for (size_t u = 0; u + 1 < s->pair_ptr.size(); ++u)
      std::sort(s->pair_items.begin() + s->pair_ptr[u], s->pair_items.begin() + s->pair_ptr[u + 1]);
  }
  s->pos_ptr.assign((size_t)s->max_uid + 2, 0);
  for (int64_t r = 0; r < n_rows; ++r)
    if (!has_threshold || h_val[r] >= threshold) s->pos_ptr[(size_t)h_uid[r] + 1]++;
  for (size_t u = 0; u + 1 < s->pos_ptr.size(); ++u) s->pos_ptr[u + 1] += s->pos_ptr[u];
  s->pos_rows.resize((size_t)s->pos_ptr.back());
  std::vector<int64_t> cur(s->pos_ptr.begin(), s->pos_ptr.end() - 1);
  for (int64_t r = 0; r < n_rows; ++r)
    if (!has_threshold || h_val[r] >= threshold) s->pos_rows[(size_t)cur[(size_t)h_uid[r]]++] = r;
  return s;
}

int drx_sampler_draw(DrxSampler *s, int32_t kind, int32_t n, int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out,
                     uint8_t *h_neg_out) {
  if (!s || n < 0 || kind < 0 || kind > 2 || !h_uid_out || !h_iid_out || !h_val_out) return DRX_EINVAL;
  for (int32_t k = 0; k < n; ++k) {
    bool null_pair;
    if (kind == DRX_DRAW_MIXED) null_pair = s->r_sel.uniform(0.0, (double)(s->neg_ratio + 1)) > 1.0;   // point_sampler.py:58
    else null_pair = (kind == DRX_DRAW_NEGATIVE);
    if (h_neg_out) h_neg_out[k] = null_pair ? 1 : 0;
    if (null_pair) {                                                                   // mem_dataset.py:154-163
      for (;;) {
        const int64_t u = s->r_neg.randint(0, s->max_uid);
        const int64_t i = s->r_neg.randint(0, s->max_iid);
        if (!s->has_pair(u, i)) {
          h_uid_out[k] = (int32_t)u; h_iid_out[k] = (int32_t)i; h_val_out[k] = 0.0;
          break;
        }
      }
    } else {                                                                           // mem_dataset.py:119-129
      if (s->pos_rows.empty()) return DRX_EINVAL;    // no eligible positive at all: the reference would spin forever
      for (;;) {
        const int64_t u = s->r_pos.randint(0, s->max_uid);
        const int64_t lo = s->pos_ptr[(size_t)u], hi = s->pos_ptr[(size_t)u + 1];
        if (hi == lo) continue;
        const int64_t r = s->pos_rows[(size_t)(lo + s->r_pos.randint(0, hi - lo - 1))];
        h_uid_out[k] = s->uid[(size_t)r]; h_iid_out[k] = s->iid[(size_t)r]; h_val_out[k] = s->val[(size_t)r];
        break;
      }
    }
  }
  return DRX_OK;
}

// One reference-mode batch in ONE call (so that a worker thread of fit() takes and drops the interpreter lock once): waits until the
// shared counter *turn equals `ticket` (sampler draws of concurrent workers happen in submission order), draws the B triples,
// passes the turn on, advances this worker's corruption generator by discard_words (the batches other workers draw) and
// produces the keep flags of the rows.
int drx_cdae_reference_draw(DrxSampler *smp, DrxRng *rng, int64_t *turn, int64_t ticket, uint64_t discard_words,
                            const int64_t *h_indptr, const int32_t *h_indices, int32_t n_items, int32_t B, double q,
                            int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out, uint8_t *h_neg_out, int32_t *h_keep_off,
                            uint8_t *h_keep, int64_t keep_capacity) {
  if (!smp || !rng || !turn) return DRX_EINVAL;
  for (unsigned spins = 0; __atomic_load_n(turn, __ATOMIC_ACQUIRE) != ticket; ++spins)
    if (spins > 64) sched_yield();
  const int rc = drx_sampler_draw(smp, DRX_DRAW_MIXED, B, h_uid_out, h_iid_out, h_val_out, h_neg_out);
  __atomic_store_n(turn, ticket + 1, __ATOMIC_RELEASE);
  if (rc) return rc;
  rng->mt.skip(discard_words);
  return drx_rng_corruption_keep(rng, h_indptr, h_indices, n_items, h_uid_out, B, q, h_keep_off, h_keep, keep_capacity);
}

// ---- draw-ahead workers of reference-mode fit() ------------------------------------------------------------------------------
// Two native threads, one per corruption generator, each fed through a small ring of jobs.  They poll for work for a few hundred
// microseconds before going to sleep and the consumer polls for results the same way: during a fit a batch is needed every
// 40-70 us, and handing work to a sleeping thread (futex wake-up on an idle core) costs about as much as the batch itself —
// with Python futures the same loop ran anywhere between 65 and 180 us per step depending on where the threads happened to sit.
struct DrawJob {
  int64_t ticket;
  uint64_t discard;
  int32_t B;
  double q;
  int32_t *uid, *iid;
  double *val;
  uint8_t *neg;
  int32_t *keep_off;
  uint8_t *keep;
  int64_t keep_cap;
  int rc;
};

struct DrxDrawAhead {
  static constexpr int kRing = 8;
  static constexpr int kPollUs = 400;
  DrxSampler *smp;
  DrxRng *rng[2];
  const int64_t *indptr;
  const int32_t *indices;
  int32_t n_items;
  int64_t turn = 0;                      // whose sampler draw is next (tickets are handed out in submission order)
  std::atomic<bool> stop{false};
  struct Worker {
    std::thread th;
    DrawJob ring[kRing];
    std::atomic<int64_t> submitted{0}, done{0};
    std::atomic<bool> sleeping{false};
    std::mutex mu;
    std::condition_variable cv;
  } w[2];

  void run(int g) {
    Worker &me = w[g];
    for (int64_t next = 0;; ++next) {
      auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0; me.submitted.load() <= next; ++spins) {
        if (stop.load()) return;
        __builtin_ia32_pause();
        if ((spins & 255) == 255 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kPollUs) {
          std::unique_lock<std::mutex> lk(me.mu);
          me.sleeping.store(true);
          // wait_until on the system clock = pthread_cond_timedwait; wait_for would use pthread_cond_clockwait, which GCC 11's
          // ThreadSanitizer does not intercept (it then reports the re-acquired mutex as a "double lock")
          me.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(50),
                           [&] { return me.submitted.load() > next || stop.load(); });
          me.sleeping.store(false);
          t0 = std::chrono::steady_clock::now();
        }
      }
      DrawJob &j = me.ring[next % kRing];
      for (unsigned spins = 0; __atomic_load_n(&turn, __ATOMIC_ACQUIRE) != j.ticket; ++spins) {   // (so that destroy never hangs)
        if (stop.load()) return;
        if (spins > 64) sched_yield();
      }
      j.rc = drx_cdae_reference_draw(smp, rng[g], &turn, j.ticket, j.discard, indptr, indices, n_items, j.B, j.q, j.uid, j.iid,
                                     j.val, j.neg, j.keep_off, j.keep, j.keep_cap);
      me.done.store(next + 1);
    }
  }
};

DrxDrawAhead *drx_drawahead_create(DrxSampler *smp, DrxRng *rng0, DrxRng *rng1, const int64_t *h_indptr, const int32_t *h_indices,
                                   int32_t n_items) {
  if (!smp || !rng0 || !rng1 || !h_indptr || !h_indices) return nullptr;
  DrxDrawAhead *d = new (std::nothrow) DrxDrawAhead;
  if (!d) return nullptr;
  d->smp = smp; d->rng[0] = rng0; d->rng[1] = rng1;
  d->indptr = h_indptr; d->indices = h_indices; d->n_items = n_items;
  for (int g = 0; g < 2; ++g) d->w[g].th = std::thread([d, g] { d->run(g); });
  return d;
}

int64_t drx_drawahead_submit(DrxDrawAhead *d, int32_t gen, int64_t ticket, uint64_t discard_words, int32_t B, double q,
                             int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out, uint8_t *h_neg_out, int32_t *h_keep_off,
                             uint8_t *h_keep, int64_t keep_capacity) {
  if (!d || gen < 0 || gen > 1 || B < 1 || !h_uid_out || !h_iid_out || !h_val_out || !h_keep_off || !h_keep) return DRX_EINVAL;
  DrxDrawAhead::Worker &wk = d->w[gen];
  const int64_t job = wk.submitted.load();
  if (job - wk.done.load() >= DrxDrawAhead::kRing) return DRX_ERETRY;          // ring full: wait for an earlier job first
  wk.ring[job % DrxDrawAhead::kRing] = DrawJob{ticket, discard_words, B, q, h_uid_out, h_iid_out, h_val_out, h_neg_out, h_keep_off,
                                                h_keep, keep_capacity, 0};
  wk.submitted.store(job + 1);
  if (wk.sleeping.load()) {
    std::lock_guard<std::mutex> lk(wk.mu);
    wk.cv.notify_one();
  }
  return job;
}

int drx_drawahead_wait(DrxDrawAhead *d, int32_t gen, int64_t job) {
  if (!d || gen < 0 || gen > 1 || job < 0 || job >= d->w[gen].submitted.load()) return DRX_EINVAL;
  DrxDrawAhead::Worker &wk = d->w[gen];
  auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; wk.done.load() <= job; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 255) == 255 &&
        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > DrxDrawAhead::kPollUs)
      std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return wk.ring[job % DrxDrawAhead::kRing].rc;
}

void drx_drawahead_destroy(DrxDrawAhead *d) {
  if (!d) return;
  d->stop.store(true);
  for (int g = 0; g < 2; ++g) {
    { std::lock_guard<std::mutex> lk(d->w[g].mu); d->w[g].cv.notify_all(); }
    if (d->w[g].th.joinable()) d->w[g].th.join();
  }
  delete d;
}

// Polls *p until it is at least `at_least` or `poll_us` microseconds have passed (1 / 0): the hand-over between fit()'s issuing thread
// and its sampling worker (drecpy_amd/_spinpool.py) — called through ctypes, i.e. WITHOUT the interpreter lock.  A batch is needed
// every 90 - 250 us; putting either thread to sleep between two of them costs a futex wake-up on an idle core each way, and the same
// loop ran at 0.09 or at 0.23 ms per step depending on where the scheduler had put the threads (r06, profiles/r06_host_handover.log).
int drx_spin_until(const int64_t *p, int64_t at_least, int32_t poll_us) {
  if (!p) return DRX_EINVAL;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) >= at_least) return 1;
    __builtin_ia32_pause();
    if ((spins & 255) == 255 &&
        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= poll_us)
      return 0;
  }
}

// first[c] = position of the first occurrence of code c in codes[0..n) (or -1): Dataset.unique() on columns of small dense integer codes
// without a sort.  Returns the number of distinct codes, or DRX_EINVAL for a code outside [0, n_codes).
int64_t drx_first_occurrence(const int64_t *codes, int64_t n, int64_t n_codes, int64_t *first) {
  if (!codes || !first || n < 0 || n_codes < 1) return DRX_EINVAL;
  for (int64_t c = 0; c < n_codes; ++c) first[c] = -1;
  int64_t distinct = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t c = codes[i];
    if (c < 0 || c >= n_codes) return DRX_EINVAL;
    if (first[c] < 0) { first[c] = i; ++distinct; }
  }
  return distinct;
}

// Lookups grouped by the row they name: row_ptr [n_rows + 1], order [T] = the lookups (positions in `keys`) of row 0, then of row 1, ...,
// each row's in ascending position — a stable counting sort on the host for drx_rows_csr_adam.
int drx_batch_csr(const int32_t *keys, int32_t T, int32_t n_rows, int32_t *row_ptr, int32_t *order) {
  if (!keys || !row_ptr || !order || T < 0 || n_rows < 1) return DRX_EINVAL;
  for (int32_t r = 0; r <= n_rows; ++r) row_ptr[r] = 0;
  for (int32_t t = 0; t < T; ++t) {
    if (keys[t] < 0 || keys[t] >= n_rows) return DRX_EINVAL;
    row_ptr[keys[t] + 1]++;
  }
  for (int32_t r = 0; r < n_rows; ++r) row_ptr[r + 1] += row_ptr[r];
  for (int32_t t = 0; t < T; ++t) order[row_ptr[keys[t]]++] = t;
  for (int32_t r = n_rows; r > 0; --r) row_ptr[r] = row_ptr[r - 1];
  row_ptr[0] = 0;
  return DRX_OK;
}

// Distinct ids of a batch (ascending) and what the DMF kernels index them with — counting through the caller's id -> rank scratch
// instead of a sort (ids are bounded by the table size).  ~20 us for 4096 ids where the numpy version took 130.
int32_t drx_batch_distinct(const int32_t *ids, int32_t B, int32_t n_rows, const int64_t *indptr, int32_t *scratch, int32_t *distinct,
                           int32_t *inv, int32_t *gptr, int32_t *grows, int32_t *off) {
  if (!ids || !scratch || !distinct || !inv || !gptr || !grows || B < 1 || n_rows < 1) return DRX_EINVAL;
  int32_t nd = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int32_t id = ids[b];
    if (id < 0 || id >= n_rows) {                    // leave the scratch as it was found
      for (int32_t j = 0; j < nd; ++j) scratch[distinct[j]] = -1;
      return DRX_EINVAL;
    }
    if (scratch[id] < 0) { scratch[id] = 0; distinct[nd++] = id; }
  }
  if ((int64_t)n_rows <= 8 * (int64_t)nd) {          // few rows: collecting the marks in id order beats sorting the list
    int32_t j = 0;
    for (int32_t id = 0; id < n_rows && j < nd; ++id)
      if (scratch[id] == 0) distinct[j++] = id;
  } else {
    std::sort(distinct, distinct + nd);
  }
  for (int32_t j = 0; j <= nd; ++j) gptr[j] = 0;
  for (int32_t j = 0; j < nd; ++j) scratch[distinct[j]] = j;
  for (int32_t b = 0; b < B; ++b) { inv[b] = scratch[ids[b]]; gptr[inv[b] + 1]++; }
  for (int32_t j = 0; j < nd; ++j) gptr[j + 1] += gptr[j];
  for (int32_t b = 0; b < B; ++b) grows[gptr[inv[b]]++] = b;       // samples of a distinct id in ascending order
  for (int32_t j = nd; j > 0; --j) gptr[j] = gptr[j - 1];
  gptr[0] = 0;
  bool overflow = false;
  if (off) {
    int64_t t = 0;
    off[0] = 0;
    for (int32_t j = 0; j < nd; ++j) {
      t += indptr ? indptr[distinct[j] + 1] - indptr[distinct[j]] : 0;
      if (t > INT32_MAX) { overflow = true; t = INT32_MAX; }      // (int32 touch offsets: a saturated prefix would make touches overlap)
      off[j + 1] = (int32_t)t;
    }
  }
  for (int32_t j = 0; j < nd; ++j) scratch[distinct[j]] = -1;
  return overflow ? DRX_EINVAL : nd;
}

// Work items of the DMF first-layer gather (DrxDmfArgs::work_order), longest first, LONG rows / columns CUT INTO SEGMENTS: entry =
// work index | segment << 24, work index i < n_u = distinct user i (off_u[i + 1] - off_u[i] non-zeros), n_u + j = distinct item j.  A
// counting sort over the bit length of the degree (descending), stable inside a class, in O(n): the few popular items whose columns
// hold thousands of entries come out in front, each as ceil(deg / seg_len) entries of at most seg_len non-zeros — segment 0 writes the
// id's first-layer sum, segment g > 0 partial row zseg[i] >> 8 + g - 1 (zseg[i] = first partial << 8 | number of partials; 0: none).
// seg_len = 0: no cutting.  Returns the number of entries (<= order_cap), or DRX_ESCRATCH when they do not fit; *n_part = partial rows.
int32_t drx_dmf_work_order(const int32_t *off_u, int32_t n_u, const int32_t *off_i, int32_t n_i, int32_t seg_len, int32_t *order,
                           int32_t order_cap, int32_t *zseg, int32_t *n_part) {
  if (!off_u || !off_i || !order || n_u < 0 || n_i < 0 || seg_len < 0 || (int64_t)n_u + n_i >= (1 << 24)) return DRX_EINVAL;
  if (seg_len > 0 && !zseg) return DRX_EINVAL;
  int32_t count[33] = {0}, start[33];
  auto cls = [](int32_t d) { return d <= 0 ? 0 : 32 - __builtin_clz((uint32_t)d); };      // 0 .. 32
  auto segs = [seg_len](int32_t d) { return seg_len > 0 && d > seg_len ? (d + seg_len - 1) / seg_len : 1; };
  int64_t entries = 0;
  for (int32_t i = 0; i < n_u + n_i; ++i) {
    const int32_t d = i < n_u ? off_u[i + 1] - off_u[i] : off_i[i - n_u + 1] - off_i[i - n_u];
    const int32_t ns = segs(d);
    if (ns > 255) return DRX_EINVAL;                  // (the caller picks seg_len >= max degree / 255)
    count[cls(d)] += ns;
    entries += ns;
  }
  if (entries > order_cap) return DRX_ESCRATCH;
  int32_t run = 0;
  for (int c = 32; c >= 0; --c) { start[c] = run; run += count[c]; }
  int32_t parts = 0;
  for (int32_t i = 0; i < n_u + n_i; ++i) {
    const int32_t d = i < n_u ? off_u[i + 1] - off_u[i] : off_i[i - n_u + 1] - off_i[i - n_u];
    const int32_t ns = segs(d);
    int32_t &at = start[cls(d)];
    for (int32_t g = 0; g < ns; ++g) order[at++] = i | (g << 24);
    if (zseg) zseg[i] = ns > 1 ? ((parts << 8) | (ns - 1)) : 0;
    parts += ns - 1;
  }
  if (n_part) *n_part = parts;
  return (int32_t)entries;
}

int drx_sampler_sample(DrxSampler *s, int32_t n, int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out) {
  return drx_sampler_draw(s, DRX_DRAW_MIXED, n, h_uid_out, h_iid_out, h_val_out, nullptr);
}

void drx_sampler_destroy(DrxSampler *s) { delete s; }

}  // extern "C"
