// Row-sharded (multi-GPU) CDAE sampled step: the per-rank device work around the two RCCL all-to-all exchanges (round-4 rewrite on
// the single-GPU step's machinery: prepared touch list, span plan, sole-toucher marks, launch order — drx_prep.hpp / drx_segreduce.hpp).
//
// Layout (BASELINE.json configuration 4): users — V rows, their histories and the triples sampled for them — are sharded
// by contiguous uid range and never move.  Item-side rows (W, W2T, b2 and their optimizer slots) are sharded by contiguous item range
// of ipr = ceil(N / world) rows.  A step requests each DISTINCT item row it touches from its owner (all-to-all of keys, then of rows),
// runs forward / backward against that row cache, reduces its contributions to one gradient row per distinct row, and returns those to
// the owners (all-to-all), which sum what the ranks sent in rank order and apply the sparse optimizer once per row.
// Everything here is per-rank and collective-free; drecpy_amd/dist.py drives the exchanges.
//
// Two key spaces:
//   * the touch list is sorted in the single-GPU step's keys ([0,N) W rows, [N,2N) W2T rows of GLOBAL item ids, 2N + local user):
//     preparation, plan, marks and reduction are the shared code;
//   * rows travel under WIRE keys (WireGeo, drx_prep.hpp), UNIT-major: an owner's local keys l = 2 * local item + (W2T ? 1 : 0) are
//     cut into `chunks` (a power of two) equal key ranges of 1 << cshift keys — whole 8192-key tiles of the presence map — and unit
//     v = chunk * world + owner holds the keys of one (chunk, owner) pair.  r06: the chunks are what the exchanges are pipelined over —
//     chunk c of an exchange is ONE all-to-all over the contiguous units c * world .. c * world + world - 1, the owner applies chunk
//     c's gradient rows while chunk c + 1 travels and gathers chunk c's rows for the NEXT step right behind (dist.py).  chunks == 1
//     is the r04 / r05 format (one all-to-all per direction).
//
// Exchange buffers (both directions): unit v's piece is n_v rows of ld floats followed by n_v scalars padded to 32 floats (rows stay
// 128-byte aligned for ld % 32 == 0).  n_v = the distinct rows asked of owner v % world in chunk v / world + 1: the piece's last row
// is a SENTINEL (key DRX_KEY_NONE) — unused on the way out; on the way back the sentinels carry this rank's gradient of the replicated
// hidden bias (scalar: the rank's loss sum), and the owner sums those of the LAST chunk's segments in rank order — every rank the same
// `world` rows: no all-reduce, no extra collective.  The scalar of a W2T row is b2 on the way out and its gradient on the way back.
#include "drx_common.hpp"
#include "drx_rows.hpp"
#include "drx_segreduce.hpp"
#include "drx_scan.hpp"
#include "drx_prep.hpp"
#include "drx_segstream.hpp"

namespace drx {

constexpr int kTileKeys = 8192;            // wire keys per workgroup of the presence-map kernels (256 words of 32 keys)
constexpr int kTileWords = kTileKeys / 32;

struct ShardGeo {
  int world, rank, ipr, shift, ld, bypass;      // bypass: this rank's OWN rows never pass through a collective (DRX_SHARD_SELF_BYPASS)
  int chunks, cshift, units;                    // exchange chunks per owner, log2 of a unit's key span, world * chunks
  uint32_t n_words, n_tiles, tiles_per_unit;
  __host__ __device__ WireGeo wg() const { return WireGeo{ipr, cshift, world}; }
  __host__ __device__ bool own_unit(int v) const { return bypass && v % world == rank; }
};

// chunks of a shard description: a power of two, at most what leaves every unit whole 8192-key tiles (smaller shards: fewer chunks)
static int chunks_of(const DrxShard &sh, int shift) {
  int c = sh.chunks < 1 ? 1 : sh.chunks;
  while (c > 1 && (shift - __builtin_ctz((unsigned)c)) < 13) c >>= 1;
  return c;
}

static ShardGeo geo_of(const DrxShard &sh, int ld) {
  ShardGeo g{};
  g.world = sh.world; g.rank = sh.rank; g.ipr = sh.items_per_rank; g.ld = ld;
  g.bypass = (sh.flags & DRX_SHARD_SELF_BYPASS) ? 1 : 0;
  g.shift = 13;
  while ((1ll << g.shift) < 2ll * sh.items_per_rank) ++g.shift;
  g.chunks = chunks_of(sh, g.shift);
  g.cshift = g.shift - __builtin_ctz((unsigned)g.chunks);
  g.units = g.world * g.chunks;
  g.tiles_per_unit = (uint32_t)((1ull << g.cshift) / kTileKeys);
  g.n_tiles = g.tiles_per_unit * (uint32_t)g.units;
  g.n_words = g.n_tiles * kTileWords;
  return g;
}

__host__ __device__ inline uint32_t pad32(uint32_t n) { return (n + 31u) & ~31u; }

// Where the rows of a prepared batch sit in its exchange buffers.
struct ShardXfer {
  const uint2 *ptab;          // [n_words]: x = the word's 32 presence bits, y = position of its first present key
  const uint32_t *first;      // [units + 1]: position of unit v's first row (sentinels included); first[units] = all rows
  const uint32_t *foff;       // [units + 1]: float offset of unit v's piece in this rank's exchange buffers — the pieces in unit order;
                              //   with the self-bypass the rank's own units come LAST (they do not travel); foff[units] = all floats
  int shift, ld;              // shift: log2 of a UNIT's key span (ShardGeo::cshift)
  __device__ __forceinline__ uint32_t pos_of(uint32_t w) const {
    const uint2 e = ptab[w >> 5];
    return e.y + (uint32_t)__popc(e.x & ((1u << (w & 31u)) - 1u));
  }
  __device__ __forceinline__ size_t row_off(uint32_t w, uint32_t pos) const {
    const uint32_t o = w >> shift;
    return (size_t)foff[o] + (size_t)(pos - first[o]) * ld;
  }
  __device__ __forceinline__ size_t scal_off(uint32_t w, uint32_t pos) const {
    const uint32_t o = w >> shift;
    return (size_t)foff[o] + (size_t)(first[o + 1] - first[o]) * ld + (pos - first[o]);
  }
  __device__ __forceinline__ size_t sentinel_row(int v) const { return (size_t)foff[v] + (size_t)(first[v + 1] - 1 - first[v]) * ld; }
  __device__ __forceinline__ size_t sentinel_scal(int v) const { return (size_t)foff[v] + (size_t)(first[v + 1] - first[v]) * (ld + 1) - 1; }
};

// ---- 1. presence map -> distinct rows, positions, per-owner counts (two small launches; independent of the sort) -----------------
static __global__ __launch_bounds__(kTileWords) void k_shard_pack(uint8_t *__restrict__ present, uint2 *__restrict__ ptab,
                                                                  int *__restrict__ tile_sum) {
  __shared__ int lds[kScanThreads / 64];
  const uint32_t w = blockIdx.x * kTileWords + threadIdx.x;
  uint4 *src = reinterpret_cast<uint4 *>(present + (size_t)w * 32);
  const uint4 a = src[0], c = src[1];
  const uint32_t x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
  uint32_t bits = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    bits |= (((x[i] & 0xFFu) ? 1u : 0u) | ((x[i] & 0xFF00u) ? 2u : 0u) | ((x[i] & 0xFF0000u) ? 4u : 0u) | ((x[i] & 0xFF000000u) ? 8u : 0u)) << (4 * i);
  if (bits) { src[0] = make_uint4(0, 0, 0, 0); src[1] = make_uint4(0, 0, 0, 0); }       // the map is all zero again for the next batch
  ptab[w].x = bits;
  int total;
  (void)block_scan_incl((int)__popc(bits), lds, total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// Positions: every workgroup sums the tile sums in front of its tile itself (n_tiles is a few hundred: cheaper than a launch of ONE
// workgroup between the two — a launch on the preparation's stream waits 10 - 40 us for room whatever it computes); one sentinel
// position behind every owner's keys.  Workgroup 0 also writes first / foff / counts and the sentinels.
static __global__ __launch_bounds__(kTileWords) void k_shard_emit(uint2 *__restrict__ ptab, const int *__restrict__ tile_sum, ShardGeo g,
                                                                  uint32_t *__restrict__ first, uint32_t *__restrict__ foff,
                                                                  long long *__restrict__ counts, uint32_t *__restrict__ uniq) {
  __shared__ int lds[kScanThreads / 64];
  __shared__ uint32_t sfirst[DRX_MAX_WORLD + 1];
  const int tpo = (int)g.tiles_per_unit, blk = (int)blockIdx.x;
  int before = 0;
  for (int i = threadIdx.x; i < blk; i += kTileWords) before += tile_sum[i];
  int tile_off;
  (void)block_scan_incl(before, lds, tile_off);
  tile_off += blk / tpo;                                   // the sentinels of the units that end in front of this tile
  const uint32_t w = blockIdx.x * kTileWords + threadIdx.x;
  uint32_t bits = ptab[w].x;
  const int c = (int)__popc(bits);
  int total;
  const int incl = block_scan_incl(c, lds, total);
  uint32_t pos = (uint32_t)(tile_off + incl - c);
  ptab[w].y = pos;
  while (bits) {
    const int i = __builtin_ctz(bits);
    bits &= bits - 1;
    uniq[pos++] = w * 32 + (uint32_t)i;
  }
  if (blk != 0) return;
  if ((int)threadIdx.x <= g.units) {                       // thread v: where unit v's rows start
    const int v = threadIdx.x;
    int sum = v;
    for (int i = 0; i < v * tpo; ++i) sum += tile_sum[i];
    sfirst[v] = (uint32_t)sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int v = 0; v <= g.units; ++v) {
      first[v] = sfirst[v];
      if (v == g.units) break;
      const uint32_t n = sfirst[v + 1] - sfirst[v];
      counts[(v % g.world) * g.chunks + v / g.world] = (long long)n;      // OWNER-major (what the count exchange sends: `chunks` per peer)
      uniq[sfirst[v + 1] - 1] = DRX_KEY_NONE;
      if (g.own_unit(v)) continue;                      // the own units: behind all the others
      foff[v] = run;
      run += n * (uint32_t)g.ld + pad32(n);
    }
    if (g.bypass)
      for (int c = 0; c < g.chunks; ++c) {
        const int v = c * g.world + g.rank;
        const uint32_t n = sfirst[v + 1] - sfirst[v];
        foff[v] = run;
        run += n * (uint32_t)g.ld + pad32(n);
      }
    foff[g.units] = run;
  }
}

// ---- 2. owner side ------------------------------------------------------------------------------------------------------------------
// The keys a rank receives are n_seg segments — micro-batch-major, then source rank — of ascending distinct wire keys closed by a
// sentinel.  koff: key index where a segment starts; foff: where its chunk (m rows, then m scalars padded to 32) starts in the
// exchange buffer.
struct SegOff {
  int koff[DRX_MAX_WORLD * DRX_MAX_MICRO + 1];
  uint32_t foff[DRX_MAX_WORLD * DRX_MAX_MICRO + 1];     // float offset of the segment's chunk in the exchange buffer
  int n_seg;
  // self-bypass: the segments this rank sent to itself (one per micro-batch, s % world == rank) are not in the exchange buffer —
  // their chunk lies in the rank's own gradient buffer of that micro-batch, at own_off floats
  int self_rank, world;                                  // self_rank < 0: no bypass
  const float *own_buf[DRX_MAX_MICRO];
  uint32_t own_off[DRX_MAX_MICRO];
  __device__ __forceinline__ bool is_own(int s) const { return self_rank >= 0 && s % world == self_rank; }
  __device__ __forceinline__ const float *chunk(const float *xbuf, int s) const {
    return is_own(s) ? own_buf[s / world] + own_off[s / world] : xbuf + foff[s];
  }
};

__device__ __forceinline__ int source_of(const SegOff &so, int j) {
  int lo = 0, hi = so.n_seg;                      // last segment whose start is <= j
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (so.koff[mid] <= j) lo = mid; else hi = mid;
  }
  return lo;
}

static int seg_off(const ShardGeo &g, const int32_t *recv_counts, int n_segments, int n, SegOff &so) {
  so.n_seg = n_segments;
  so.world = g.world;
  so.self_rank = g.bypass ? g.rank : -1;
  so.koff[0] = 0; so.foff[0] = 0;
  unsigned long long f = 0;
  for (int s = 0; s < n_segments; ++s) {
    if (recv_counts[s] < 1) return DRX_EINVAL;                 // every segment holds at least its sentinel
    so.koff[s + 1] = so.koff[s] + recv_counts[s];
    if (!(g.bypass && s % g.world == g.rank)) f += (unsigned long long)recv_counts[s] * g.ld + pad32((uint32_t)recv_counts[s]);
    if (f >= 0xFFFFFFFFull) return DRX_EINVAL;
    so.foff[s + 1] = (uint32_t)f;
  }
  for (int m = 0; m < DRX_MAX_MICRO; ++m) { so.own_buf[m] = nullptr; so.own_off[m] = 0; }
  return so.koff[n_segments] == n ? DRX_OK : DRX_EINVAL;
}

// Every received key is entered in a direct-address table tab[local key][segment] = key index (no two writers per entry): the owner
// apply then lets the row of the LOWEST segment holding a key sum all holders in segment order — one pass, fixed order, no sort, no
// atomics.  Parameter-independent: built when the keys arrive, ahead of the step.
static __global__ void k_owner_scatter(ShardGeo g, SegOff so, const uint32_t *__restrict__ recv_keys, int n, uint32_t *tab) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t k = recv_keys[j];
    if (k == DRX_KEY_NONE) continue;
    const uint32_t lk = g.wg().local(k);
    tab[(size_t)lk * so.n_seg + source_of(so, j)] = (uint32_t)j;
  }
}

// rows[j] = the W / W2T row named by recv_keys[j], its scalar = the output bias (W2T rows); sentinels are skipped
template <int G, int J>
static __global__ __launch_bounds__(kBlock) void k_shard_gather_rows(DrxCdaeParams P, ShardGeo g, SegOff so,
                                                                     const uint32_t *__restrict__ recv_keys, int n, float *__restrict__ out) {
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  constexpr int NF = J == 1 ? 4 : 2;
  for (int j0 = (blockIdx.x * gpb + threadIdx.x / G) * NF; j0 < n; j0 += gridDim.x * gpb * NF) {
    float4 v[NF][J];
    size_t dst[NF], sdst[NF];
    float bv[NF];
    bool live[NF], outr[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int j = j0 + q;
      live[q] = false; outr[q] = false; bv[q] = 0.f; dst[q] = 0; sdst[q] = 0;
      const uint32_t k = j < n ? recv_keys[j] : DRX_KEY_NONE;
      if (k == DRX_KEY_NONE) continue;
      const uint32_t t = g.wg().local(k);                              // local key in [0, 2 * ipr): 2 * local item + (W2T row)
      const bool is_out = t & 1u;
      const size_t row = t >> 1;
      const int s = source_of(so, j);
      if (so.is_own(s)) continue;                                      // the requester reads its own rows from the tables
      dst[q] = (size_t)so.foff[s] + (size_t)(j - so.koff[s]) * P.ld;
      sdst[q] = (size_t)so.foff[s] + (size_t)(so.koff[s + 1] - so.koff[s]) * P.ld + (size_t)(j - so.koff[s]);
      live[q] = true; outr[q] = is_out;
      float *const tw = P.W, *const to = P.W2T;
      load_row<G, J>(is_out ? to : tw, row, P.ld, lane, v[q]);
      if (is_out && lane == 0) bv[q] = P.b2[row];
    }
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (live[q]) {
        store_row<G, J>(out + dst[q], 0, P.ld, lane, v[q]);
        if (outr[q] && lane == 0) out[sdst[q]] = bv[q];
      }
  }
}

// ---- 3. forward / backward of the local triples against the row cache -------------------------------------------------------------
struct ShardStep {
  float *dz1, *g2, *dz2, *lossb;              // [B, ld] x 2, [B] x 2
  const uint8_t *solo_v, *solo_o;             // sole-toucher marks of the prepared list ([B] each) or nullptr (rows of <= 16 floats)
  const int32_t *order;                       // launch order (longest histories first)
  const float *cache;                         // the rows this rank asked for, as they arrived
  float *gsend;                               // gradient rows on their way back: same geometry
  ShardXfer X;
  WireGeo G;
  int ipr, b_norm, n_items;                   // n_items: GLOBAL
  int self;                                   // this rank when its own rows bypass the exchange (read from the tables), else -1
};

template <int G, int J, int KIND>
static __global__ __launch_bounds__(kBlock) void k_shard_fwd_bwd(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                                 uint32_t qthr, int loss_kind, ShardStep S) {
  const int lane = threadIdx.x % G;
  const int slot = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (slot >= bt.B) return;
  const int b = S.order ? S.order[slot] : slot;
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  // a group fetches CH history entries per round; every lane turns ITS entries into cache offsets (index -> wire key -> position: two
  // small dependent loads, in parallel over the lanes), then the rows are fetched NF at a time
  constexpr int IPL = G >= 16 ? 1 : 16 / G;
  constexpr int CH = G * IPL;
  constexpr int NF = J == 1 ? 8 : 4;
  for (int64_t c = s; c < e; c += CH) {
    uint32_t off[IPL];              // in floats (the host checked that the buffer stays below 2^32 floats)
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int64_t j = c + r * G + lane;
      off[r] = DRX_KEY_NONE;
      if (j < e) {
        const int item = H.indices[j];
        const uint32_t jj = (uint32_t)(j - s);
        const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, jj) >= qthr);
        if (kf) {
          const int o = item / S.ipr;
          if (o == S.self) off[r] = 0x80000000u | (uint32_t)((item - o * S.ipr) * P.ld);        // an own row: straight from the table
          else {
            const uint32_t w = S.G.wire(item, 0);
            off[r] = (uint32_t)S.X.row_off(w, S.X.pos_of(w));
          }
        }
      }
    }
    const int n_here = (int)((e - c) < (int64_t)CH ? (e - c) : (int64_t)CH);
    for (int t = 0; t < n_here; t += NF) {
      float4 r[NF][J];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int tt = t + q;
        uint32_t so_ = off[0];
#pragma unroll
        for (int rr = 1; rr < IPL; ++rr) so_ = (tt / G == rr) ? off[rr] : so_;
        const uint32_t oq = tt < n_here ? (uint32_t)__shfl((int)so_, tt % G, G) : DRX_KEY_NONE;
#pragma unroll
        for (int jx = 0; jx < J; ++jx) r[q][jx] = f4_zero();
        if (oq != DRX_KEY_NONE) {
          const float *const cb = S.cache, *const wb = P.W;             // (pointers read as scalars, the VALUE selected)
          load_row<G, J>(((oq & 0x80000000u) ? wb : cb) + (size_t)(oq & 0x7FFFFFFFu), 0, P.ld, lane, r[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < NF; ++q)
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_add(acc[jx], r[q][jx]);
    }
  }
  float4 h[J], w2[J];
  hidden_act<G, J>(P, u, scale, lane, acc, h);
  const int i = bt.iid[b];
  const int oo = i / S.ipr;
  const uint32_t wo = S.G.wire(i, 1);
  const uint32_t po = S.X.pos_of(wo);
  const size_t ro = S.X.row_off(wo, po), sco = S.X.scal_off(wo, po);
  const bool own_out = oo == S.self;
  {
    const float *const cb = S.cache, *const tb = P.W2T;
    load_row<G, J>(own_out ? tb + (size_t)(i - oo * S.ipr) * P.ld : cb + ro, 0, P.ld, lane, w2);
  }
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) d += f4_dot(w2[j], h[j]);
  d = group_sum<G>(d);
  const float y = bt.y[b];
  const float p = sigmoidf_(d + (own_out ? P.b2[i - oo * S.ipr] : S.cache[sco]));
  const float invB = 1.0f / (float)S.b_norm;
  float lval, dp;
  if (loss_kind == DRX_LOSS_BCE) { lval = bce_elem(y, p); dp = bce_grad(y, p) * invB; }
  else { lval = (p - y) * (p - y); dp = 2.0f * (p - y) * invB; }
  const float dz2 = dp * p * (1.0f - p);
  float4 dz1[J], g2[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    dz1[j].x = dz2 * w2[j].x * h[j].x * (1.0f - h[j].x); dz1[j].y = dz2 * w2[j].y * h[j].y * (1.0f - h[j].y);
    dz1[j].z = dz2 * w2[j].z * h[j].z * (1.0f - h[j].z); dz1[j].w = dz2 * w2[j].w * h[j].w * (1.0f - h[j].w);
    g2[j].x = dz2 * h[j].x; g2[j].y = dz2 * h[j].y; g2[j].z = dz2 * h[j].z; g2[j].w = dz2 * h[j].w;
  }
  store_row<G, J>(S.dz1, (size_t)b, P.ld, lane, dz1);
  if (lane == 0) S.lossb[b] = lval;
  const bool solo_v = S.solo_v && S.solo_v[b], solo_o = S.solo_o && S.solo_o[b];
  if (solo_o) {          // this sample alone touches W2T[i] on this rank: its gradient row goes straight into the exchange buffer
    store_row<G, J>(S.gsend + ro, 0, P.ld, lane, g2);
    if (lane == 0) S.gsend[sco] = dz2;
  } else {
    store_row<G, J>(S.g2, (size_t)b, P.ld, lane, g2);
    if (lane == 0) S.dz2[b] = dz2;
  }
  if (solo_v) {          // V rows are local: a row only this sample touches is updated here (same arithmetic as the segment path)
    OptScalars o = opt_for(opt, 0, S.b_norm);
    o.inv_k = 1.0f / (float)P.k;
    float4 w[J];
    load_row<G, J>(P.V, (size_t)u, P.ld, lane, w);
    row_update<G, J, KIND>(o, P.V, opt.s1[2], opt.s2[2], (size_t)u, P.ld, lane, w, dz1);
  }
}

// ---- 4. local reduction: one gradient row per distinct item row into the exchange buffer, V rows updated in place -----------------
template <int KIND>
struct LocalPolicyT {
  DrxCdaeParams P;               // local tables (V used here)
  DrxOptim opt;
  int b_norm, n_items, ipr;      // n_items: GLOBAL
  float scale;
  const float *dz1;              // g2 = dz1 + g2_off (value select, see DirectPolicyT in drx_cdae.hip)
  long long g2_off;
  const float *dz2;
  float *gsend;
  ShardXfer X;
  WireGeo WG;
  template <int G, int J>
  __device__ __forceinline__ void load(uint32_t key, uint32_t b, int lane, float4 (&row)[J], float &sc, float &coef) const {
    const uint32_t N = (uint32_t)n_items;
    const bool is_out = key >= N && key < 2 * N;
    load_row<G, J>(dz1 + (is_out ? g2_off : 0ll), (size_t)b, P.ld, lane, row);
    if (is_out) sc = dz2[b];
    coef = key < N ? scale : 1.0f;
  }
  template <int G, int J>
  __device__ __forceinline__ void finish(uint32_t key, int, int lane, const float4 (&g)[J], float gs) const {
    const uint32_t N = (uint32_t)n_items;
    if (key >= 2 * N) {
      const size_t row = key - 2 * N;
      OptScalars o = opt_for(opt, 0, b_norm);
      o.inv_k = 1.0f / (float)P.k;
      float4 w[J];
      load_row<G, J>(P.V, row, P.ld, lane, w);
      row_update<G, J, KIND>(o, P.V, opt.s1[2], opt.s2[2], row, P.ld, lane, w, g);
    } else {
      const bool is_out = key >= N;
      const int item = (int)(is_out ? key - N : key);
      const uint32_t w = WG.wire(item, is_out ? 1 : 0);
      const uint32_t pos = X.pos_of(w);
      store_row<G, J>(gsend + X.row_off(w, pos), 0, P.ld, lane, g);
      if (is_out && lane == 0) gsend[X.scal_off(w, pos)] = gs;
    }
  }
  // for the streamed reduction (drx_segstream.hpp): three arrays — W and W2T rows (GLOBAL item ids) whose sums are PARKED in the
  // gradient exchange buffer at the row's place there, V rows (local) applied in place
  static constexpr bool kStreamParks = true;
  __device__ __forceinline__ StreamArrays stream_arrays() const {
    const uint32_t N = (uint32_t)n_items;
    return StreamArrays{{0u, N, 2u * N}, {dz1, dz1 + g2_off, dz1}, {nullptr, nullptr, P.V}, {nullptr, nullptr, opt.s1[2]},
                        {scale, 1.0f, 1.0f}, {nullptr, dz2, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  }
  __device__ __forceinline__ bool stream_parked(uint32_t av) const { return av < 2u; }
  __device__ __forceinline__ void stream_park_at(uint32_t av, uint32_t item, uint32_t &roff, uint32_t &soff) const {
    const uint32_t w = WG.wire((int)item, av == 1u ? 1 : 0);
    const uint32_t pos = X.pos_of(w);
    roff = (uint32_t)X.row_off(w, pos);
    soff = (uint32_t)X.scal_off(w, pos);
  }
  __device__ __forceinline__ float *stream_park_base() const { return gsend; }
  __device__ __forceinline__ float stream_decay() const { return opt.reg_rate / (float)b_norm; }
  __device__ __forceinline__ void stream_update(float g, float &p, float &a) const {
    static_assert(KIND == DRX_OPT_ADAGRAD || KIND < 0, "one slot per element");
    OptScalars o = opt_for(opt, 0, b_norm);
    float unused = 0.f;
    opt_update1<DRX_OPT_ADAGRAD>(o, g, p, a, unused);
  }
  __device__ __forceinline__ void stream_update_scalar(float g, float &p, float &a) const { stream_update(g, p, a); }
};

// Last extra workgroup of the span launch: the rank's gradient of the hidden bias (sum of the column-sum partials the reduction's
// extra workgroups left) into the sentinel row of EVERY destination, the rank's loss sum into that row's scalar.
template <int G, int J>
struct ShardBiasExtra {
  int ld, units;
  BiasArgs A;
  float *gsend;
  ShardXfer X;
  __device__ __forceinline__ void operator()(float *lds) const {
    __shared__ float red[kFixBlock / 64];
    constexpr int R = kFixBlock / G;
    const int lane = threadIdx.x % G, r = threadIdx.x / G;
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
    constexpr int NB = J == 1 ? 8 : 2;
    for (int i = r; i < A.n_part; i += NB * R) {
      float4 v[NB][J];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
#pragma unroll
        for (int j = 0; j < J; ++j) v[q][j] = f4_zero();
        if (i + q * R < A.n_part) load_row<G, J>(A.part, (size_t)(i + q * R), ld, lane, v[q]);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(acc[j], v[q][j]);
    }
    store_row<G, J>(lds, (size_t)r, ld, lane, acc);
    __syncthreads();
    const float *lp = A.part + (size_t)A.n_part * ld;
    float a = 0.f;
    for (int i = threadIdx.x; i < A.n_part; i += kFixBlock) a += lp[i];
    const float tl = block_sum(a, red);                        // (thread 0 holds it)
    if (r == 0) {
      float4 g[J];
#pragma unroll
      for (int j = 0; j < J; ++j) g[j] = f4_zero();
#pragma unroll 8
      for (int rr = 0; rr < R; ++rr) {
        float4 v[J];
        load_row<G, J>(lds, (size_t)rr, ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(g[j], v[j]);
      }
      for (int v = 0; v < units; ++v) {                        // the sentinel closes unit v's piece (the owner reads the last chunk's)
        store_row<G, J>(gsend + X.sentinel_row(v), 0, ld, lane, g);
        if (threadIdx.x == 0) gsend[X.sentinel_scal(v)] = tl;
      }
    }
  }
};

// ---- 5. owner side: sum the gradient rows received for each owned row in segment order, apply the optimizer once per row ----------
template <int G, int J, int KIND>
static __global__ __launch_bounds__(kBlock) void k_shard_apply(DrxCdaeParams P, DrxOptim opt, ShardGeo g, int b_norm, SegOff so,
                                                               const uint32_t *__restrict__ recv_keys, int n,
                                                               const uint32_t *__restrict__ tab, const float *__restrict__ grecv,
                                                               int row_blocks, float *loss_out, int do_bias) {
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  const int W = so.n_seg;
  if ((int)blockIdx.x >= row_blocks) {
    // the hidden bias: the sentinel rows of all segments in order (every rank sums the same rows: b stays identical everywhere);
    // only with the step's LAST exchange chunk
    if (do_bias && threadIdx.x < G) {
      float4 gb[J], w[J];
#pragma unroll
      for (int j = 0; j < J; ++j) gb[j] = f4_zero();
      float ls = 0.f;
      for (int s = 0; s < W; ++s) {
        const int m = so.koff[s + 1] - so.koff[s], i = m - 1;
        const float *ch = so.chunk(grecv, s);
        float4 v[J];
        load_row<G, J>(ch + (size_t)i * P.ld, 0, P.ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(gb[j], v[j]);
        ls += ch[(size_t)m * P.ld + i];
      }
      load_row<G, J>(P.b, 0, P.ld, lane, w);
      OptScalars o = opt_for(opt, 0, b_norm);
      if (o.kind == DRX_OPT_ROWWISE_ADAGRAD) o.kind = DRX_OPT_ADAGRAD;      // the bias vectors keep one accumulator per element
      o.rb = 0.f;
      row_update<G, J>(o, P.b, opt.s1[3], opt.s2[3], 0, P.ld, lane, w, gb);
      if (threadIdx.x == 0 && loss_out) { loss_out[0] = ls / (float)b_norm; loss_out[1] = 0.f; }
    }
    return;
  }
  // NR consecutive keys per group and round: their holder lists, gradient rows, parameter and slot rows are requested side by side
  // (one key at a time left the kernel a chain of four dependent round trips per row: 141 us for 260 k rows; the keys of a segment
  // are ascending, so consecutive keys walk the tables and the gradient buffer in order)
  constexpr int NR = J == 1 ? 4 : 2;
  for (int j0 = (blockIdx.x * gpb + threadIdx.x / G) * NR; j0 < n; j0 += row_blocks * gpb * NR) {
    uint32_t lk[NR];
    bool live[NR];
    float4 gsum[NR][J];
    float gs[NR];
    if (W == 1) {                                                  // one segment: every key is its own (only) holder
      const float *ch = so.chunk(grecv, 0);
      const int m = so.koff[1];
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int j = j0 + q;
        const uint32_t key = j < n ? recv_keys[j] : DRX_KEY_NONE;
        live[q] = key != DRX_KEY_NONE;
        lk[q] = g.wg().local(key);
        gs[q] = 0.f;
#pragma unroll
        for (int jx = 0; jx < J; ++jx) gsum[q][jx] = f4_zero();
        if (live[q]) {
          load_row<G, J>(ch + (size_t)j * P.ld, 0, P.ld, lane, gsum[q]);
          gs[q] = ch[(size_t)m * P.ld + j];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int j = j0 + q;
        const uint32_t key = j < n ? recv_keys[j] : DRX_KEY_NONE;
        live[q] = key != DRX_KEY_NONE;
        lk[q] = g.wg().local(key);
        gs[q] = 0.f;
#pragma unroll
        for (int jx = 0; jx < J; ++jx) gsum[q][jx] = f4_zero();
        if (!live[q]) continue;
        const uint32_t *row_tab = tab + (size_t)lk[q] * W;
        bool leader = true, first = true;
        for (int s0 = 0; s0 < W && leader; s0 += G) {               // G segments at a time, one table entry per lane
          const uint32_t ent = (s0 + lane < W) ? row_tab[s0 + lane] : DRX_KEY_NONE;
          const int n_here = min(G, W - s0);
          for (int t = 0; t < n_here; t += 4) {
            uint32_t r[4];
            float4 v[4][J];
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              r[u] = (t + u < n_here) ? (uint32_t)__shfl((int)ent, t + u, G) : DRX_KEY_NONE;
              sc[u] = 0.f;
#pragma unroll
              for (int jx = 0; jx < J; ++jx) v[u][jx] = f4_zero();
            }
            if (first) {                                           // the first holder must be this very row
              uint32_t r0 = DRX_KEY_NONE;
#pragma unroll
              for (int u = 3; u >= 0; --u) if (r[u] != DRX_KEY_NONE) r0 = r[u];
              if (r0 != DRX_KEY_NONE) { first = false; if (r0 != (uint32_t)j) { leader = false; break; } }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (r[u] != DRX_KEY_NONE) {
                const int sg = s0 + t + u;                         // the table's column IS the segment
                const int i = (int)r[u] - so.koff[sg];
                const float *ch = so.chunk(grecv, sg);
                load_row<G, J>(ch + (size_t)i * P.ld, 0, P.ld, lane, v[u]);
                sc[u] = ch[(size_t)(so.koff[sg + 1] - so.koff[sg]) * P.ld + i];
              }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
              for (int jx = 0; jx < J; ++jx) f4_add(gsum[q][jx], v[u][jx]);
              gs[q] += sc[u];
            }
          }
        }
        if (!leader || first) live[q] = false;
      }
    }
    // the NR rows' parameter rows first (independent loads), then the updates
    float4 w[NR][J];
    float *const wt = P.W, *const w2 = P.W2T, *const a0 = opt.s1[0], *const a1 = opt.s1[1], *const c0 = opt.s2[0], *const c1 = opt.s2[1];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
#pragma unroll
      for (int jx = 0; jx < J; ++jx) w[q][jx] = f4_zero();
      if (live[q]) {
        const bool is_out = lk[q] & 1u;
        load_row<G, J>(is_out ? w2 : wt, lk[q] >> 1, P.ld, lane, w[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      if (!live[q]) continue;
      const bool is_out = lk[q] & 1u;
      const size_t row = lk[q] >> 1;
      OptScalars o = opt_for(opt, 0, b_norm);
      o.inv_k = 1.0f / (float)P.k;
      row_update<G, J, KIND>(o, is_out ? w2 : wt, is_out ? a1 : a0, is_out ? c1 : c0, row, P.ld, lane, w[q], gsum[q]);
      if (is_out && lane == 0) {
        const int kind = KIND >= 0 ? KIND : o.kind;
        float pb = P.b2[row], m = opt.s1[4][row], v = kind == DRX_OPT_ADAM ? opt.s2[4][row] : 0.f;
        o.rb = 0.f;
        opt_update1<KIND>(o, gs[q], pb, m, v);
        P.b2[row] = pb; opt.s1[4][row] = m;
        if (kind == DRX_OPT_ADAM) opt.s2[4][row] = v;
      }
    }
  }
}

// ---- layouts ------------------------------------------------------------------------------------------------------------------------
struct ShardPrepBufs {
  PrepBufs R;
  uint2 *ptab;
  uint32_t *uniq;
  uint32_t *first, *foff;
  long long *counts;
  size_t off_uniq, off_counts, uniq_cap, result_bytes;
};

static DrxCdaeParams key_params(const DrxCdaeParams &p, const DrxShard &sh) {
  DrxCdaeParams g = p;
  g.n_items = sh.n_items;              // the touch list's key space: GLOBAL item ids, local users
  g.n_users = sh.n_users_local;
  return g;
}

static ShardPrepBufs shard_prep_layout(Carver &cv, const DrxCdaeParams &p, const DrxShard &sh, int B, int n_touch_slots) {
  ShardPrepBufs L{};
  const ShardGeo g = geo_of(sh, p.ld);
  L.R = prep_layout(cv, key_params(p, sh), B, n_touch_slots);
  L.ptab = cv.take<uint2>(g.n_words);
  L.uniq_cap = (size_t)n_touch_slots + (size_t)B + (size_t)g.units;
  L.off_uniq = align_up(cv.off, 256);
  L.uniq = cv.take<uint32_t>(L.uniq_cap);
  L.first = cv.take<uint32_t>(DRX_MAX_WORLD + 1);
  L.foff = cv.take<uint32_t>(DRX_MAX_WORLD + 1);
  L.off_counts = align_up(cv.off, 256);
  L.counts = cv.take<long long>(DRX_MAX_WORLD);
  L.result_bytes = align_up(cv.off, 256);
  return L;
}

struct ShardStepBufs {
  float *dz1, *g2, *dz2, *lossb, *phead, *ptail, *phs, *pts, *pblock, *pbs, *bpart;
  int T, n_chunks, n_bpart;
};

static ShardStepBufs shard_step_layout(Carver &cv, int chunk /* of the list: SpanPlan::chunk */, int ld, int B, int n_touch_slots) {
  ShardStepBufs S{};
  S.T = n_touch_slots + 2 * B;
  S.n_chunks = (S.T + chunk - 1) / chunk;
  S.n_bpart = 1024;
  S.dz1 = cv.take<float>((size_t)B * ld);
  S.g2 = cv.take<float>((size_t)B * ld);
  S.dz2 = cv.take<float>(B);
  S.lossb = cv.take<float>(B);
  S.phead = cv.take<float>((size_t)S.n_chunks * ld);
  S.ptail = cv.take<float>((size_t)S.n_chunks * ld);
  S.phs = cv.take<float>(S.n_chunks);
  S.pts = cv.take<float>(S.n_chunks);
  const int cpb = kSegBlock / pick_geom(ld).G;
  const int n_blocks = (S.n_chunks + cpb - 1) / cpb;
  S.pblock = cv.take<float>((size_t)n_blocks * ld);
  S.pbs = cv.take<float>(n_blocks);
  S.bpart = cv.take<float>((size_t)S.n_bpart * (ld + 1));
  return S;
}

static int check_shard(const DrxShard *sh);
static int check_bypass(const DrxShard *sh, int ld) {       // own rows are addressed by a 31-bit float offset into the local table
  return ((sh->flags & DRX_SHARD_SELF_BYPASS) && (unsigned long long)sh->items_per_rank * ld >= 0x80000000ull) ? DRX_EINVAL : DRX_OK;
}
static int check_shard(const DrxShard *sh) {
  if (!sh || sh->world < 1 || sh->world > DRX_MAX_WORLD || sh->rank < 0 || sh->rank >= sh->world || sh->items_per_rank < 1 || sh->n_items < 1 ||
      sh->n_users_local < 1)
    return DRX_EINVAL;
  if ((long long)sh->items_per_rank * sh->world < (long long)sh->n_items) return DRX_EINVAL;
  if ((uint64_t)2 * sh->n_items + (uint64_t)sh->n_users_local + 1 >= 0xFFFFFFFFull) return DRX_EINVAL;
  if (sh->chunks < 0 || sh->chunks > DRX_MAX_CHUNKS || (sh->chunks & (sh->chunks - 1))) return DRX_EINVAL;       // 0 / 1: unchunked
  const ShardGeo g = geo_of(*sh, 4);
  if (g.shift > 26 || ((uint64_t)sh->world << g.shift) >= 0x80000000ull) return DRX_EINVAL;
  if (g.units > DRX_MAX_WORLD) return DRX_EINVAL;                  // (first / foff / counts hold DRX_MAX_WORLD + 1 entries)
  return DRX_OK;
}

static int check_local_params(const DrxCdaeParams *p, const DrxShard *sh) {
  if (!p || !p->W || !p->W2T || !p->V || !p->b || !p->b2) return DRX_EINVAL;
  if (p->k < 1 || p->k > DRX_MAX_K || p->ld < p->k || (p->ld & 3) || p->ld > DRX_MAX_K) return DRX_EINVAL;
  if (p->n_items != sh->items_per_rank || p->n_users != sh->n_users_local) return DRX_EINVAL;       // the LOCAL tables
  return DRX_OK;
}

}  // namespace drx

using namespace drx;

extern "C" {

int32_t drx_shard_chunks(const DrxShard *sh) { return check_shard(sh) ? 0 : geo_of(*sh, 4).chunks; }
int32_t drx_shard_unit_shift(const DrxShard *sh) { return check_shard(sh) ? 0 : geo_of(*sh, 4).cshift; }

size_t drx_shard_prep_bytes(const DrxCdaeParams *p, const DrxShard *sh, int32_t B, int32_t n_touch_slots) {
  if (!p || check_shard(sh) || B < 1 || n_touch_slots < 0) return 0;
  Carver cv(nullptr, 0);
  (void)shard_prep_layout(cv, *p, *sh, B, n_touch_slots);
  return align_up(cv.off, 256) + 256;
}

size_t drx_shard_work_bytes(const DrxShard *sh) {
  if (check_shard(sh)) return 0;
  const ShardGeo g = geo_of(*sh, 4);
  return align_up((size_t)g.n_words * 32, 256) + align_up((size_t)g.n_tiles * sizeof(int), 256) + 256;
}

int drx_shard_prep_layout(const DrxCdaeParams *p, const DrxShard *sh, int32_t B, int32_t n_touch_slots, size_t *out4) {
  if (!p || check_shard(sh) || B < 1 || n_touch_slots < 0 || !out4) return DRX_EINVAL;
  Carver cv(nullptr, 0);
  const ShardPrepBufs L = shard_prep_layout(cv, *p, *sh, B, n_touch_slots);
  out4[0] = L.off_uniq; out4[1] = L.off_counts; out4[2] = L.uniq_cap; out4[3] = L.result_bytes;
  return DRX_OK;
}

int drx_shard_prepare(const DrxCdaeParams *p, const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt, void *prepared,
                      size_t prepared_bytes, void *work, size_t work_bytes, void *stream) {
  if (check_shard(sh)) return DRX_EINVAL;
  int rc = check_local_params(p, sh);
  if (rc) return rc;
  if (!hist || !hist->indptr || !hist->indices || !bt || !bt->uid || !bt->iid || !bt->keep_off || bt->B < 1 || !prepared || !work ||
      bt->q < 0.f || bt->q >= 1.f)
    return DRX_EINVAL;
  if (((uintptr_t)prepared | (uintptr_t)work) & 255) return DRX_EINVAL;
  if (work_bytes < drx_shard_work_bytes(sh)) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  Carver cp(prepared, prepared_bytes);
  const ShardPrepBufs L = shard_prep_layout(cp, *p, *sh, bt->B, bt->n_touch_slots);
  if (!cp.ok()) return DRX_ESCRATCH;
  const ShardGeo g = geo_of(*sh, p->ld);
  uint8_t *present = (uint8_t *)work;
  int *tile_sum = (int *)((char *)work + align_up((size_t)g.n_words * 32, 256));
  const DrxCdaeParams pk = key_params(*p, *sh);
  rc = prepare_impl(&pk, hist, bt, L.R, st, true, TouchPresence{present, g.wg()});
  if (rc) return rc;
  if (p->ld <= 16) order_by_degree(bt, L.R, st, true);
  hipLaunchKernelGGL(k_shard_pack, dim3(g.n_tiles), dim3(kTileWords), 0, st, present, L.ptab, tile_sum);
  hipLaunchKernelGGL(k_shard_emit, dim3(g.n_tiles), dim3(kTileWords), 0, st, L.ptab, tile_sum, g, L.first, L.foff, L.counts, L.uniq);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

size_t drx_shard_owner_table_bytes(const DrxShard *sh, int32_t n_segments) {
  if (check_shard(sh) || n_segments < 1 || n_segments > DRX_MAX_WORLD * DRX_MAX_MICRO) return 0;
  return align_up((size_t)2 * sh->items_per_rank * n_segments * sizeof(uint32_t), 256);
}

int drx_shard_owner_index(const DrxShard *sh, const uint32_t *recv_keys, int32_t n, const int32_t *recv_counts, int32_t n_segments,
                          int32_t chunk, void *table, size_t table_bytes, void *stream) {
  if (check_shard(sh) || !recv_keys || !recv_counts || !table || n < 1) return DRX_EINVAL;
  if (n_segments < sh->world || n_segments % sh->world || n_segments > sh->world * DRX_MAX_MICRO) return DRX_EINVAL;
  if (table_bytes < drx_shard_owner_table_bytes(sh, n_segments)) return DRX_ESCRATCH;
  SegOff so{};
  const ShardGeo g = geo_of(*sh, 4);
  const int rc = seg_off(g, recv_counts, n_segments, n, so);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  // the table rows of THIS chunk's key range only (the chunks of a step share one table: their key ranges are disjoint)
  if (chunk < 0 || chunk >= g.chunks) return DRX_EINVAL;
  const size_t lk_lo = (size_t)chunk << g.cshift, lk_all = (size_t)2 * sh->items_per_rank;
  const size_t lk_hi = std::min(lk_all, ((size_t)chunk + 1) << g.cshift);
  if (lk_hi > lk_lo)
    DRX_HIP(hipMemsetAsync((uint32_t *)table + lk_lo * n_segments, 0xFF, (lk_hi - lk_lo) * n_segments * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_owner_scatter, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), dim3(256), 0, st, g, so, recv_keys, n,
                     (uint32_t *)table);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_gather_rows(const DrxCdaeParams *p, const DrxShard *sh, const uint32_t *recv_keys, int32_t n, const int32_t *recv_counts,
                          int32_t n_segments, float *out, void *stream) {
  if (check_shard(sh)) return DRX_EINVAL;
  int rc = check_local_params(p, sh);
  if (rc) return rc;
  if (!recv_keys || !recv_counts || !out || n < 1 || n_segments < 1 || n_segments > DRX_MAX_WORLD * DRX_MAX_MICRO) return DRX_EINVAL;
  SegOff so{};
  const ShardGeo g = geo_of(*sh, p->ld);
  rc = seg_off(g, recv_counts, n_segments, n, so);
  if (rc) return rc;
  if (g.bypass && sh->world == 1) return DRX_OK;               // every request is the rank's own: nothing to gather
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, J)                                                                                                   \
  {                                                                                                                  \
    const int per = (kBlock / G) * (J == 1 ? 4 : 2);                                                                 \
    int blocks = (n + per - 1) / per;                                                                                \
    if (blocks > 16384) blocks = 16384;                                                                              \
    hipLaunchKernelGGL((k_shard_gather_rows<G, J>), dim3(blocks), dim3(kBlock), 0, st, *p, g, so, recv_keys, n, out); \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

size_t drx_shard_step_scratch_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots) {
  if (!p || B < 1 || n_touch_slots < 0) return 0;
  Carver cv(nullptr, 0);
  (void)shard_step_layout(cv, kChunk, p->ld, B, n_touch_slots);          // (the shorter chunks: the larger of the two layouts)
  return align_up(cv.off, 256) + 256;
}

int drx_shard_step_local(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt,
                         const void *prepared, size_t prepared_bytes, const float *rows_cache, float *grad_send, int32_t b_norm,
                         int32_t loss_kind, void *scratch, size_t scratch_bytes, void *const *events, void *stream) {
  if (check_shard(sh)) return DRX_EINVAL;
  int rc = check_local_params(p, sh);
  if (rc) return rc;
  if (!opt || !hist || !hist->indptr || !hist->indices || !bt || !bt->uid || !bt->iid || !bt->y || !bt->keep_off || bt->B < 1 ||
      !prepared || !rows_cache || !grad_send || !scratch || b_norm < 1)
    return DRX_EINVAL;
  if (opt->kind != DRX_OPT_ADAM && opt->kind != DRX_OPT_ADAGRAD && opt->kind != DRX_OPT_ROWWISE_ADAGRAD) return DRX_EINVAL;
  for (int i = 0; i < 5; ++i)
    if (!opt->s1[i] || (opt->kind == DRX_OPT_ADAM && !opt->s2[i])) return DRX_EINVAL;
  if (check_bypass(sh, p->ld)) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cp(const_cast<void *>(prepared), prepared_bytes);
  const ShardPrepBufs L = shard_prep_layout(cp, *p, *sh, bt->B, bt->n_touch_slots);
  if (!cp.ok()) return DRX_ESCRATCH;
  Carver cv(scratch, scratch_bytes);
  const ShardStepBufs B = shard_step_layout(cv, L.R.plan.chunk, p->ld, bt->B, bt->n_touch_slots);
  if (!cv.ok()) return DRX_ESCRATCH;
  const ShardGeo g = geo_of(*sh, p->ld);
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  const bool marks = p->ld > 16;                               // (prepare_impl: rows of <= 16 floats carry no sole-toucher marks)
  const ShardXfer X{L.ptab, L.first, L.foff, g.cshift, p->ld};
  ShardStep S{B.dz1, B.g2, B.dz2, B.lossb, marks ? L.R.solo_v : nullptr, marks ? L.R.solo_o : nullptr, L.R.order, rows_cache, grad_send,
              X, g.wg(), g.ipr, b_norm, sh->n_items, g.bypass ? g.rank : -1};
  SegBufs SB{L.R.keys_s, L.R.vals_s, B.phead, B.ptail, B.phs, B.pts, nullptr, nullptr, nullptr, nullptr, B.T, B.n_chunks, p->ld, nullptr};
  PlanBufs PB{B.pblock, B.pbs};
  const int rows_per_block = (bt->B + B.n_bpart - 1) / B.n_bpart;
  const int n_bpart = (bt->B + rows_per_block - 1) / rows_per_block;
  BiasArgs BA{B.dz1, B.bpart, B.lossb, B.lossb /* non-null: the loss partials are always taken */, bt->B, n_bpart, rows_per_block};
  const bool long_segments = drx::long_segments(B.T, key_params(*p, *sh));
#define EV(i) do { if (events) DRX_HIP(hipEventRecord((hipEvent_t)events[i], st)); } while (0)
#define REDUCE_AND_SPANS(G, J, KIND)                                                                                   \
  {                                                                                                                    \
    using POLT = LocalPolicyT<KIND>;                                                                                   \
    POLT polk{*p, *opt, b_norm, sh->n_items, g.ipr, scale, B.dz1, (long long)(B.g2 - B.dz1), B.dz2, grad_send, X, g.wg()}; \
    BiasPartialExtra<G, J> bpx{p->ld, BA};                                                                             \
    ShardBiasExtra<G, J> bfx{p->ld, g.units, BA, grad_send, X};                                                        \
    const int cpb = kSegBlock / G;                                                                                     \
    const dim3 rgrid(n_bpart + (B.n_chunks + cpb - 1) / cpb);                                                          \
    const size_t lds_r = seg_reduce_lds_bytes(cpb, p->ld, long_segments);                                              \
    const size_t lds_b = ((size_t)(kFixBlock / G) * (p->ld + 1)) * 4;                                                  \
    bool streamed = false;                                                                                             \
    if constexpr (kStreamDepth > 0 && J == 1 && G >= 16 && KIND == DRX_OPT_ADAGRAD) {                                  \
      /* lists of short segments over rows of exactly 64 / 128 / 256 floats: the streamed form (drx_segstream.hpp) */  \
      if (!long_segments && p->ld == 4 * G && bt->B < (1 << kStreamIndexBits) && sh->n_items < (1 << kStreamIndexBits) && \
          p->n_users < (1 << kStreamIndexBits)) {           /* (places in the exchange buffer are 32-bit float offsets: ShardXfer::foff) */ \
        BiasPartialExtra<G, J, cpb * 64> bpxs{p->ld, BA};                                                              \
        hipLaunchKernelGGL((k_seg_reduce_stream<4 * G, kStreamDepth, POLT, BiasPartialExtra<G, J, cpb * 64>>), rgrid, dim3(cpb * 64), \
                           seg_stream_lds_bytes(p->ld, kStreamDepth), st, SB, PB, L.R.plan, polk, n_bpart, bpxs);      \
        streamed = true;                                                                                               \
      }                                                                                                                \
    }                                                                                                                  \
    if (streamed) { }                                                                                                  \
    else if (long_segments)                                                                                            \
      hipLaunchKernelGGL((k_seg_reduce_planned<G, J, POLT, true, BiasPartialExtra<G, J>>), rgrid, dim3(kSegBlock), lds_r, st, SB, PB, \
                         L.R.plan, polk, n_bpart, bpx);                                                                \
    else                                                                                                               \
      hipLaunchKernelGGL((k_seg_reduce_planned<G, J, POLT, false, BiasPartialExtra<G, J>>), rgrid, dim3(kSegBlock), lds_r, st, SB, PB, \
                         L.R.plan, polk, n_bpart, bpx);                                                                \
    EV(2);                                                                                                             \
    if (lds_b > 48 * 1024)                                                                                             \
      DRX_HIP(hipFuncSetAttribute((const void *)k_span_planned<G, J, POLT, ShardBiasExtra<G, J>>,                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));                            \
    hipLaunchKernelGGL((k_span_planned<G, J, POLT, ShardBiasExtra<G, J>>), dim3(kLongBlocks + kShortBlocks + 1), dim3(kFixBlock),   \
                       lds_b, st, SB, PB, L.R.plan, polk, kLongBlocks, kShortBlocks, bfx);                             \
    EV(3);                                                                                                             \
  }
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = kBlock / G;                                                                                        \
    EV(0);                                                                                                             \
    if (opt->kind == DRX_OPT_ADAGRAD) {                                                                                \
      hipLaunchKernelGGL((k_shard_fwd_bwd<G, J, DRX_OPT_ADAGRAD>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *opt, *hist, \
                         *bt, scale, qthr, loss_kind, S);                                                              \
      EV(1);                                                                                                           \
      REDUCE_AND_SPANS(G, J, DRX_OPT_ADAGRAD);                                                                         \
    } else {                                                                                                           \
      hipLaunchKernelGGL((k_shard_fwd_bwd<G, J, -1>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *opt, *hist, *bt, scale, \
                         qthr, loss_kind, S);                                                                          \
      EV(1);                                                                                                           \
      REDUCE_AND_SPANS(G, J, -1);                                                                                      \
    }                                                                                                                  \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
#undef REDUCE_AND_SPANS
#undef EV
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_apply(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, int32_t b_norm, const uint32_t *recv_keys,
                    const float *grad_recv, int32_t n, const int32_t *recv_counts, int32_t n_segments, int32_t chunk, const void *table,
                    const float *const *own_grad, const int64_t *own_off, float *loss_out, void *stream) {
  if (check_shard(sh)) return DRX_EINVAL;
  int rc = check_local_params(p, sh);
  if (rc) return rc;
  if (!opt || n < 1 || b_norm < 1 || !recv_counts || !recv_keys || !grad_recv || !table) return DRX_EINVAL;
  if (n_segments < sh->world || n_segments % sh->world || n_segments > sh->world * DRX_MAX_MICRO) return DRX_EINVAL;
  SegOff so{};
  const ShardGeo g = geo_of(*sh, p->ld);
  rc = seg_off(g, recv_counts, n_segments, n, so);
  if (rc) return rc;
  if (chunk < 0 || chunk >= g.chunks) return DRX_EINVAL;
  const int do_bias = chunk == g.chunks - 1;       // b and the loss: once per step, with the last chunk (its sentinels carry them)
  if (g.bypass) {                                  // the pieces this rank "sent" to itself: still in its own gradient buffers
    if (!own_grad || !own_off) return DRX_EINVAL;
    for (int m = 0; m < n_segments / sh->world; ++m) {
      if (!own_grad[m] || own_off[m] < 0 || own_off[m] >= 0xFFFFFFFFll) return DRX_EINVAL;
      so.own_buf[m] = own_grad[m];
      so.own_off[m] = (uint32_t)own_off[m];
    }
  }
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = (kBlock / G) * (J == 1 ? 4 : 2);                                                                   \
    int blocks = (n + gpb - 1) / gpb;                                                                                  \
    if (blocks > 16384) blocks = 16384;                                                                                \
    if (opt->kind == DRX_OPT_ADAGRAD)                                                                                  \
      hipLaunchKernelGGL((k_shard_apply<G, J, DRX_OPT_ADAGRAD>), dim3(blocks + 1), dim3(kBlock), 0, st, *p, *opt, g, b_norm, so,    \
                         recv_keys, n, (const uint32_t *)table, grad_recv, blocks, loss_out, do_bias);                 \
    else                                                                                                               \
      hipLaunchKernelGGL((k_shard_apply<G, J, -1>), dim3(blocks + 1), dim3(kBlock), 0, st, *p, *opt, g, b_norm, so, recv_keys, n, \
                         (const uint32_t *)table, grad_recv, blocks, loss_out, do_bias);                               \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

}  // extern "C"
