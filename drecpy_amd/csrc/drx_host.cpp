// Host-side (CPU) pieces of libdrx.so: CPython-exact MT19937 streams, the PointSampler algorithm and the CDAE
// corruption stream.  These replace the O(nnz)-per-draw pandas scans of DRecPy/Dataset/mem_dataset.py:111-163
// and the N-long Python list comprehensions of DRecPy/Recommender/cdae.py:61-63 while producing bit-identical
// streams (checked against stdlib `random.Random` and the golden vectors generated from the reference).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <vector>
#include "drx.h"

namespace {

// MT19937 exactly as CPython's Modules/_randommodule.c drives it (algorithm of Matsumoto & Nishimura).
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
  uint32_t next() {
    if (idx >= 624) {
      int kk;
      uint32_t y;
      for (kk = 0; kk < 624 - 397; ++kk) {
        y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7FFFFFFFu);
        s[kk] = s[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
      }
      for (; kk < 623; ++kk) {
        y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7FFFFFFFu);
        s[kk] = s[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
      }
      y = (s[623] & 0x80000000u) | (s[0] & 0x7FFFFFFFu);
      s[623] = s[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
      idx = 0;
    }
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
  std::vector<uint64_t> pairs;            // sorted (uid << 32 | iid) of EVERY dataframe row
  std::vector<int64_t> pos_ptr;           // per-user CSR over rows eligible as positives (dataframe order)
  std::vector<int64_t> pos_rows;
  std::vector<int32_t> uid, iid;
  std::vector<double> val;
};

extern "C" {

DrxRng *drx_rng_create(int64_t seed) {
  DrxRng *r = new (std::nothrow) DrxRng;
  if (r) r->mt.seed_int(seed);
  return r;
}
void drx_rng_destroy(DrxRng *r) { delete r; }
double drx_rng_random(DrxRng *r) { return r->mt.random(); }
int64_t drx_rng_randint(DrxRng *r, int64_t a, int64_t b) { return r->mt.randint(a, b); }

int drx_rng_corruption_keep(DrxRng *r, const int64_t *h_indptr, const int32_t *h_indices, int32_t n_items,
                            const int32_t *h_uid, int32_t B, double q, int32_t *h_keep_off, uint8_t *h_keep,
                            int64_t keep_capacity) {
  if (!r || !h_indptr || !h_indices || !h_uid || !h_keep_off || !h_keep || B < 0) return DRX_EINVAL;
  int64_t off = 0;
  for (int32_t b = 0; b < B; ++b) {
    h_keep_off[b] = (int32_t)off;
    const int64_t s = h_indptr[h_uid[b]], e = h_indptr[h_uid[b] + 1];
    if (off + (e - s) > keep_capacity) return DRX_ESCRATCH;
    int64_t j = s;
    // cdae.py:63 draws one uniform(0,1) for EVERY item n = 0..N-1 of the row, positives or not
    for (int32_t n = 0; n < n_items; ++n) {
      const double x = r->mt.random();     // uniform(0, 1) == 0 + (1 - 0) * random()
      if (j < e && h_indices[j] == n) {
        h_keep[off + (j - s)] = (x < q) ? 0 : 1;
        ++j;
      }
    }
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
  s->pairs.resize(n_rows);
  for (int64_t r = 0; r < n_rows; ++r) s->pairs[r] = ((uint64_t)(uint32_t)h_uid[r] << 32) | (uint32_t)h_iid[r];
  std::sort(s->pairs.begin(), s->pairs.end());
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
        const uint64_t key = ((uint64_t)u << 32) | (uint64_t)i;
        if (!std::binary_search(s->pairs.begin(), s->pairs.end(), key)) {
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

int drx_sampler_sample(DrxSampler *s, int32_t n, int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out) {
  return drx_sampler_draw(s, DRX_DRAW_MIXED, n, h_uid_out, h_iid_out, h_val_out, nullptr);
}

void drx_sampler_destroy(DrxSampler *s) { delete s; }

}  // extern "C"
