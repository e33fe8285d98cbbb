// The parameter-independent PREPARATION of a sampled-mode batch, shared by the single-GPU sparse step (drx_cdae.hip) and the row-sharded
// step (drx_shard.hip): the (row key, sample) touch list of a batch, its radix sort, the span plan of the segmented reduction
// (drx_segreduce.hpp), the sole-toucher marks and the forward kernel's launch order — everything that depends on the batch alone and is
// built on a side stream while earlier batches train.  Key space: [0, N) W rows, [N, 2N) W2T rows, [2N, 2N + U) V rows.
// (Kernels are `static`: each translation unit that includes this carries its own copy.)
#pragma once
#include <algorithm>
#include "drx_common.hpp"
#include "drx_rows.hpp"
#include "drx_segreduce.hpp"
#include "drx_scan.hpp"

namespace drx {

// Touch list of one batch (row key, sample): depends only on the batch, never on the parameters, so it can be built
// and sorted for batch t+1 while batch t trains (drx_cdae_sparse_prepare on a second stream).
// Also clears the sole-toucher marks of the batch (solo: [2B] bytes, or nullptr) and pads the slots beyond
// the last sample's up to T with DRX_KEY_NONE (n_touch_slots may be an upper bound) — memsets the preparation would otherwise launch.
// `present` (row-sharded step, or nullptr): one byte per WIRE key of an item row — owner-major, owner o = item / ipr at
// WireGeo::wire(item, W2T row?) below — set to 1 for every item row this batch touches (plain byte stores: every
// writer writes 1); drx_shard.hip turns the map into the batch's distinct rows, their positions in the exchange buffers and the
// per-owner counts without waiting for the sort.
// WIRE key of an item row (include/drx.h "row-sharded multi-GPU step"): owner o = item / ipr, local key l = 2 * (item - o * ipr) + (W2T
// row ? 1 : 0), exchange chunk c = l >> cshift, UNIT v = c * world + o; key = (v << cshift) | (l & ((1 << cshift) - 1)) — unit-major, so a
// rank's distinct keys are contiguous per (chunk, owner) and chunk c of the exchange is one contiguous run of `world` units.
struct WireGeo {
  int ipr, cshift, world;
  __host__ __device__ __forceinline__ uint32_t wire(int item, int is_out) const {
    const int o = item / ipr;
    const uint32_t l = 2u * (uint32_t)(item - o * ipr) + (uint32_t)(is_out ? 1 : 0);
    return (((l >> cshift) * (uint32_t)world + (uint32_t)o) << cshift) | (l & ((1u << cshift) - 1u));
  }
  // the owner's local key of a wire key (chunk bits back in front of the in-chunk bits): bit 0 = W2T row, the rest = local item
  __host__ __device__ __forceinline__ uint32_t local(uint32_t w) const {
    return (((w >> cshift) / (uint32_t)world) << cshift) | (w & ((1u << cshift) - 1u));
  }
  __host__ __device__ __forceinline__ uint32_t unit(uint32_t w) const { return w >> cshift; }
};

struct TouchPresence {
  uint8_t *present;
  WireGeo G;
  __device__ __forceinline__ uint32_t wire(int item, int is_out) const { return G.wire(item, is_out); }
};

static __global__ __launch_bounds__(kBlock) void k_sparse_touches(int n_items, DrxHistory H, DrxBatch bt, uint32_t qthr, uint32_t *keys,
                                                           uint32_t *vals, int T, uint8_t *solo, uint32_t *zero_a,
                                                           int n_zero_a, uint32_t *zero_b, int n_zero_b, uint32_t *zero_c, int n_zero_c,
                                                           TouchPresence pres) {
  constexpr int G = 16;
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  // (two more ranges of words the preparation wants zeroed: the span plan's counters + window bytes, the degree-order work area)
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < n_zero_a; w += gridDim.x * kBlock) zero_a[w] = 0u;
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < n_zero_b; w += gridDim.x * kBlock) zero_b[w] = 0u;
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < n_zero_c; w += gridDim.x * kBlock) zero_c[w] = 0u;      // (the sort's counters and tile words)
  if (b >= bt.B) return;
  if (solo && lane == 0) { solo[b] = 0; solo[bt.B + b] = 0; }
  if (b == bt.B - 1)
    for (int j = bt.keep_off[bt.B] + 2 * bt.B + lane; j < T; j += G) { keys[j] = DRX_KEY_NONE; vals[j] = 0; }
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
  const int base = bt.keep_off[b] + 2 * b;
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  for (int64_t j = s + lane; j < e; j += G) {
    const uint32_t jj = (uint32_t)(j - s);
    const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, jj) >= qthr);
    const int item = H.indices[j];
    keys[base + jj] = kf ? (uint32_t)item : DRX_KEY_NONE;
    vals[base + jj] = (uint32_t)b;
    if (pres.present && kf) pres.present[pres.wire(item, 0)] = 1;
  }
  if (lane == 0) {
    const int deg = (int)(e - s);
    keys[base + deg] = (uint32_t)(n_items + bt.iid[b]);       vals[base + deg] = (uint32_t)b;
    if (pres.present) pres.present[pres.wire(bt.iid[b], 1)] = 1;
    keys[base + deg + 1] = (uint32_t)(2 * n_items) + (uint32_t)u;       vals[base + deg + 1] = (uint32_t)b;
  }
}

// Launch order of the forward kernel's triples: longest histories first, triples of similar length side by side (r03 phase stamps:
// a triple lives 19 us on average but 25 us at the 90th percentile and far longer for the few users with hundreds of items; a
// workgroup waits for its slowest triple and the launch for its last workgroups — 47 % of the chip's group slots were occupied on
// average).  A counting sort by history length in 256 buckets of 4 items, two small launches on the preparation's stream (counts;
// scatter), each workgroup over 1024 triples; `work` = 512 zeroed ints (k_sparse_touches clears them).  A first version did it all
// in ONE workgroup: 152 us of side-stream time per step, which made the preparation — not the training — the pipeline's bound.
// The order inside a bucket comes from atomics and differs from run to run: it decides only WHERE a triple is computed, never a result.
// (degree_bucket / same_bucket_lanes / degree_counts_body: drx_common.hpp — the counts can ride in the sort's first launch)
static __global__ __launch_bounds__(1024) void k_degree_counts(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work) {
  __shared__ unsigned int cnt[256];
  degree_counts_body<1024>(keep_off, B, work, (int)blockIdx.x, cnt);
}

template <int NT>
__device__ __forceinline__ void degree_scatter_body(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work,
                                                    int32_t *__restrict__ order, int block_id, unsigned int *lds /* [768] */) {
  unsigned int *start = lds, *cnt = lds + 256, *base = lds + 512;
  for (int i = threadIdx.x; i < 256; i += NT) cnt[i] = 0;
  if (threadIdx.x < 64) {                            // exclusive scan of the 256 global counts by one wave: 4 bins per lane
    unsigned int c[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c[q] = work[threadIdx.x * 4 + q]; sum += c[q]; }
    unsigned int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
    unsigned int run = inc - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) { start[threadIdx.x * 4 + q] = run; run += c[q]; }
  }
  __syncthreads();
  const int b = block_id * NT + (int)threadIdx.x, lane = threadIdx.x & 63;
  const bool valid = b < B;
  const int d = valid ? degree_bucket(keep_off, b) : 0;
  const unsigned long long m = same_bucket_lanes(valid, d);
  const int leader = valid ? __ffsll((long long)m) - 1 : lane;
  unsigned int at = 0;
  if (valid && lane == leader) at = atomicAdd(&cnt[d], (unsigned int)__popcll(m));        // place inside this workgroup's share
  at = __shfl(at, leader);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += NT) base[i] = cnt[i] ? atomicAdd(&work[256 + i], cnt[i]) : 0u;     // the share's place in the bucket
  __syncthreads();
  if (valid) order[start[d] + base[d] + at + __popcll(m & ((1ull << lane) - 1ull))] = b;
}

static __global__ __launch_bounds__(1024) void k_degree_scatter(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work,
                                                         int32_t *__restrict__ order) {
  __shared__ unsigned int lds[768];
  degree_scatter_body<1024>(keep_off, B, work, order, (int)blockIdx.x, lds);
}

// V and W2T rows are mostly touched by ONE sample of the batch (a user is drawn once, output items are uniform).
// When the touch list is prepared ahead of the step, such rows are marked here: the forward/backward kernel, which holds the
// sample's gradient rows in registers, then applies their update itself (no g2 row written, no re-read of the gradient
// and of the parameter row later), and the touch is blanked (DRX_KEY_NONE) so that the segmented reduction passes over
// it.  A sole toucher cannot race: no other sample of the batch reads or writes that row.  The marks are a byte per sample.
// (W rows with one touch — 2/3 of a 10M x 1M batch's distinct W rows — stay with the reduction: updating them from the forward kernel
// was measured twice in r03 and cost it more than the reduction saved; HISTORY.md.)
static __global__ void k_mark_solo(uint32_t *keys_s, const uint32_t *__restrict__ vals_s, int T, uint32_t n_items, int B, uint8_t *solo_v,
                            uint8_t *solo_o) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    const uint32_t k = keys_s[j];
    if (k == DRX_KEY_NONE || k < n_items) continue;
    const uint32_t prev = j > 0 ? keys_s[j - 1] : DRX_KEY_NONE, next = j + 1 < T ? keys_s[j + 1] : DRX_KEY_NONE;
    if (k == prev || k == next) continue;            // (a neighbour blanked concurrently was a different key anyway)
    const uint32_t b = vals_s[j];
    if (k < 2 * n_items) solo_o[b] = 1;
    else solo_v[b] = 1;
    keys_s[j] = DRX_KEY_NONE;
  }
}

// The span plan and the sole-toucher marks in ONE launch over the freshly sorted list (two launches cost the preparation — the
// pipeline's bound once the training kernels got faster — a launch gap and 15 us): the first n_chunks threads plan their chunk, then
// every thread marks its share of the touches.  The two do not disturb each other: a key that is blanked has one touch, a key whose
// run the plan measures crosses a chunk border (>= 2 touches), and the plan's searches only test keys for equality with such a key.
// order_blocks workgroups behind the plan's: the scatter half of the launch order (k_degree_scatter's work, 256 triples each; the
// counts were taken before the sort) — one launch less for the preparation's stream to wait for.
static __global__ __launch_bounds__(256) void k_plan_and_mark(uint32_t *keys_s, const uint32_t *__restrict__ vals_s, int T, int n_chunks, int cpb,
                                                       SpanPlan P, uint32_t n_items, int B, uint8_t *solo_v, uint8_t *solo_o,
                                                       int order_blocks, const int32_t *keep_off, unsigned int *order_work,
                                                       int32_t *order) {
  const int plan_blocks = (int)gridDim.x - order_blocks;
  if ((int)blockIdx.x >= plan_blocks) {
    __shared__ unsigned int lds[768];
    degree_scatter_body<256>(keep_off, B, order_work, order, (int)blockIdx.x - plan_blocks, lds);
    return;
  }
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  // (a list laid down compact by the transposed preparation says how long it is: chunks behind its cnt[20] touches hold padding only)
  // (only lists with placed blocks — long segments — can be compact ones: the others never read the word)
  const int real = P.xrank ? (int)P.cnt[20] : 0;
  if (tid < n_chunks && (real == 0 || (long long)tid * P.chunk < (long long)real)) plan_chunk(keys_s, T, n_chunks, cpb, P, tid);
  const int nb = (n_chunks + cpb - 1) / cpb;
  if (P.xrank && (tid & ~63) < nb) place_block(keys_s, vals_s, T, cpb, B, P, tid, tid < nb);
  // (a list laid down compact by the transposed preparation says where its W2T / V parts are: the last 2B of its cnt[20] touches)
  const int j_lo = real != 0 ? max(0, real - 2 * B) : 0, j_hi = real != 0 ? min(T, real) : T;
  for (int j = j_lo + tid; j < j_hi; j += plan_blocks * blockDim.x) {
    const uint32_t k = keys_s[j];
    if (k == DRX_KEY_NONE || k < n_items) continue;
    const uint32_t prev = j > 0 ? keys_s[j - 1] : DRX_KEY_NONE, next = j + 1 < T ? keys_s[j + 1] : DRX_KEY_NONE;
    if (k == prev || k == next) continue;
    const uint32_t b = vals_s[j];
    if (k < 2 * n_items) solo_o[b] = 1;
    else solo_v[b] = 1;
    keys_s[j] = DRX_KEY_NONE;
  }
}

// hidden bias b: column sums of dz1 in two deterministic stages, then a dense optimizer update.  Both stages are ROLES of
// the two tail launches of the sparse step (k_sparse_tail_a / _b below): they share a launch with the span fix-ups, which
// they do not depend on.
struct BiasArgs {
  const float *dz1;       // [B, ld]
  float *part;            // [n_part, ld] column-sum partials, then [n_part] loss partials
  const float *lossb;     // [B]
  float *loss_out;        // nullptr: no loss wanted
  int B, n_part, rows_per_block;
};

template <int G, int J, int NT>
__device__ __forceinline__ void bias_partial_body(int ld, const BiasArgs &A, int block_id, float *lds /* [NT/G, ld] */, float *red) {
  constexpr int R = NT / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b0 = block_id * A.rows_per_block, b1 = min(A.B, b0 + A.rows_per_block);
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  constexpr int NB = J == 1 ? 8 : 2;            // rows in flight per group (one at a time made these sums chains of dependent loads)
  for (int b = b0 + r; b < b1; b += NB * R) {
    float4 v[NB][J];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
#pragma unroll
      for (int j = 0; j < J; ++j) v[q][j] = f4_zero();
      if (b + q * R < b1) load_row<G, J>(A.dz1, (size_t)(b + q * R), ld, lane, v[q]);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(acc[j], v[q][j]);
  }
  store_row<G, J>(lds, (size_t)r, ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 t[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
#pragma unroll 8
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
    }
    store_row<G, J>(A.part, (size_t)block_id, ld, lane, t);
  }
  if (A.loss_out) {                                // this block's slice of the per-sample losses
    float a = 0.f;
    for (int b = b0 + (int)threadIdx.x; b < b1; b += NT) a += A.lossb[b];
    const float tl = block_sum(a, red);
    if (threadIdx.x == 0) A.part[(size_t)A.n_part * ld + block_id] = tl;
  }
}

// NT: threads of the launch it rides in (k_seg_reduce_planned: kSegBlock; the streamed form: one wave per chunk)
template <int G, int J, int NT = kSegBlock>
struct BiasPartialExtra {
  int ld;
  BiasArgs A;
  __device__ __forceinline__ void operator()(float *lds) const {
    __shared__ float red[NT / 64 > 0 ? NT / 64 : 1];
    bias_partial_body<G, J, NT>(ld, A, (int)blockIdx.x, lds, red);
  }
};

constexpr int kLongBlocks = 256, kShortBlocks = 1024;      // k_span_planned: workgroups striding over the long / the short spans

// a work item of DRX_BATCH_SHARE_USERS: n <= kShareTriples consecutive sorted positions p0 .. of ONE user's samples, and where that user's
// history lies (the forward workgroup reads ONE record instead of walking usamp -> uid -> indptr)
struct __attribute__((aligned(32))) WorkItem { int32_t p0, n, user, deg; long long hist_start, pad_; };

struct PrepBufs {
  uint32_t *keys_s, *vals_s, *keys, *vals;
  void *sort_temp;
  size_t sort_bytes;
  uint8_t *solo_v, *solo_o;     // [B] each (see k_mark_solo)
  SpanPlan plan;                // chunk-crossing segments of the list (k_plan_spans)
  int32_t *order;               // [B] launch order of the forward kernel (k_degree_counts / k_degree_scatter)
  // DRX_BATCH_SHARE_USERS (lists of long segments only): the batch's WORK ITEMS — the samples of one user, in pieces of at most
  // share_item_triples(ld) (the row groups of a forward workgroup), in the order of the sorted (user, sample) pairs
  int32_t *usamp;               // [B] sorted position -> sample (the samples grouped by user, ascending inside a user)
  int32_t *pitem;               // [B] sorted position -> its work item
  WorkItem *witem;              // [B] work item -> its record
  int32_t *worder;              // [B] launch slot -> work item, longest histories first (buckets of log2(history length))
  int32_t *n_du;                // [128]: [0] work items (k_tp_item_*)
  unsigned int *order_work;     // [512] bucket counts | running places (zeroed by k_sparse_touches)
  int n_chunks;
  size_t result_bytes;
  int T, bits;
};

// more than 8 touches per table row on average: rows collect long runs of touches (MovieLens shapes)
static inline bool long_segments(int T, const DrxCdaeParams &P) { return (int64_t)T > 8 * ((int64_t)2 * P.n_items + P.n_users); }

static PrepBufs prep_layout(Carver &cv, const DrxCdaeParams &P, int B, int n_touch_slots) {
  PrepBufs R{};
  R.T = n_touch_slots + 2 * B;
  R.bits = bits_for((uint64_t)2 * P.n_items + P.n_users + 1);
  // the RESULT first and contiguous (drx_cdae_prep_result_bytes: what a step reads, and all that has to travel when one rank
  // prepares a list for the others), then what only the preparation itself needs
  R.keys_s = cv.take<uint32_t>(R.T);
  R.vals_s = cv.take<uint32_t>(R.T);
  R.solo_v = cv.take<uint8_t>((size_t)2 * B);
  R.solo_o = R.solo_v ? R.solo_v + B : nullptr;
  R.plan.chunk = seg_chunk(long_segments(R.T, P));
  R.n_chunks = (R.T + R.plan.chunk - 1) / R.plan.chunk;
  R.plan.desc = cv.take<uint2>(R.n_chunks);
  R.plan.cnt = cv.take<uint32_t>(128);        // [0] short spans, [1] long spans, [16], [17], [kXBase ..): XCD placement (drx_segreduce.hpp)
  R.plan.ext = cv.take<uint8_t>(R.n_chunks);
  {
    // XCD placement of the reduction's workgroups (SpanPlan::xlist): lists of LONG segments only — more than 8 touches per table row
    // (the lists' STRIDE is that of the smallest block — two chunks — so that the layout of a prepared list does not depend on the
    // row width: a column-sharded job hands lists between ranks of different widths)
    R.plan.xstride = (R.n_chunks + 1) / 2;
    const bool placed = long_segments(R.T, P);
    // (DRX_STREAM_PLACED: lists of short segments placing the blocks inside their hot rows too — a measured loss, off: drx_segreduce.hpp)
    const bool placed_any = placed || DRX_STREAM_PLACED != 0;
    R.plan.xrank = placed_any ? cv.take<uint32_t>((size_t)R.plan.xstride) : nullptr;
    R.plan.xperm = placed_any ? cv.take<uint32_t>((size_t)R.plan.xstride) : nullptr;
    R.usamp = placed ? cv.take<int32_t>(B) : nullptr;
    R.pitem = placed ? cv.take<int32_t>(B) : nullptr;
    R.witem = placed ? cv.take<WorkItem>((size_t)B) : nullptr;
    R.worder = placed ? cv.take<int32_t>(B) : nullptr;
    R.n_du = placed ? cv.take<int32_t>(128) : nullptr;
  }
  R.order = cv.take<int32_t>(B);
  R.result_bytes = align_up(cv.off, 256);
  R.keys = cv.take<uint32_t>(R.T);
  R.vals = cv.take<uint32_t>(R.T);
  R.sort_bytes = sort_pairs_temp_bytes(R.T, R.bits);
  R.sort_temp = cv.take<char>(R.sort_bytes);
  R.order_work = cv.take<unsigned int>(512);
  return R;
}

// plan.cnt (64 words) and plan.ext (n_chunks bytes) are consecutive 256-byte-aligned allocations: one range of words to zero
static int plan_zero_words(const PrepBufs &R) { return (int)(((const char *)R.plan.ext + R.n_chunks - (const char *)R.plan.cnt + 3) / 4); }

// The chunk-crossing segments of a sorted list, short ones and long ones (drx_segreduce.hpp, planned variant).  On the pristine list:
// BEFORE the sole-toucher marks blank any key.
static int plan_spans(const DrxCdaeParams *p, const DrxBatch *bt, const PrepBufs &R0, hipStream_t st, bool cleared) {
  const PrepBufs &R = R0;
  if (!cleared) DRX_HIP(hipMemsetAsync(R.plan.cnt, 0, (size_t)plan_zero_words(R) * 4, st));     // (prepare_impl's touch kernel clears them)
  hipLaunchKernelGGL(k_plan_spans<0>, dim3((R.n_chunks + 255) / 256), dim3(256), 0, st, R.keys_s, R.vals_s, R.T, R.n_chunks,
                     kSegBlock / pick_geom(p->ld).G, bt->B, R.plan);
  if (R.plan.xrank) hipLaunchKernelGGL(k_place_blocks, dim3(256), dim3(256), 0, st, R.plan, R.n_chunks);
  return DRX_OK;
}

// (see k_degree_counts)
static void order_by_degree(const DrxBatch *bt, const PrepBufs &R, hipStream_t st, bool cleared) {
  if (!cleared) (void)hipMemsetAsync(R.order_work, 0, 512 * sizeof(unsigned int), st);
  const int blocks = (bt->B + 1023) / 1024;
  hipLaunchKernelGGL(k_degree_counts, dim3(blocks), dim3(1024), 0, st, bt->keep_off, bt->B, R.order_work);
  hipLaunchKernelGGL(k_degree_scatter, dim3(blocks), dim3(1024), 0, st, bt->keep_off, bt->B, R.order_work, R.order);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Preparation through the history's TRANSPOSE (r04; DrxHistory::t_rank): lists of LONG segments — MovieLens shapes, where a batch of
// 65 536 triples over 6 040 users and 3 706 items has 10 M touches and the sort of those pairs (two passes of 320 us + a histogram) had
// made the PREPARATION, not the training kernels, the bound of a step.  The sorted list is the static item -> users order of the
// training set expanded by the batch's samples of every user: only the 2B (user | item, sample) pairs of the batch are sorted (one
// stable sort: samples ascending per id), then every history entry (user u, position j, item n) of a user of the batch counts the
// samples of u whose corruption mask keeps position j — the count stored at the entry's ITEM-MAJOR RANK — a scan places them, and a
// second walk writes (n, sample): the W part of the list, item-major, compact; the W2T part (N + item) and the V part (2N + user) ARE
// the sorted batch pairs.  Inside a segment the touches come user by user, samples ascending: another FIXED order than the sort's
// (samples ascending) — a function of the batch alone, as bit-reproducible as before.  Work areas are carved out of the regions the
// big sort would have used (keys / vals / sort_temp); returns kTpFallback when they do not fit (the caller then sorts).
constexpr int kTpFallback = 0x7fff0001;

static __global__ __launch_bounds__(256) void k_tp_begin(DrxBatch bt, int n_users, uint32_t *__restrict__ k2, uint32_t *__restrict__ v2,
                                                         int32_t *__restrict__ se, int n_se, uint8_t *solo, uint32_t *zero_a, int n_zero_a,
                                                         uint32_t *zero_b, int n_zero_b) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int j = tid; j < 2 * bt.B; j += nt) {
    k2[j] = j < bt.B ? (uint32_t)bt.uid[j] : (uint32_t)n_users + (uint32_t)bt.iid[j - bt.B];
    v2[j] = (uint32_t)j;
  }
  for (int j = tid; j < n_se; j += nt) se[j] = 0;                      // start / end of every id's run in the sorted pairs
  for (int j = tid; j < 2 * bt.B; j += nt) solo[j] = 0;
  for (int j = tid; j < n_zero_a; j += nt) zero_a[j] = 0u;
  for (int j = tid; j < n_zero_b; j += nt) zero_b[j] = 0u;
}

static __global__ __launch_bounds__(256) void k_tp_runs(const uint32_t *__restrict__ ks, int n, int32_t *__restrict__ start, int32_t *__restrict__ end) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t k = ks[p];
    if (p == 0 || ks[p - 1] != k) start[k] = p;
    if (p == n - 1 || ks[p + 1] != k) end[k] = p + 1;
  }
}

// triples of one work item (DRX_BATCH_SHARE_USERS; k_items_fwd_bwd keeps one bag per triple in registers); its LDS: the partial bags
// of its waves + the item's sample ids.  Rows of 33 .. 512 floats only (narrower rows: the plain kernels).
#ifndef DRX_ITEM_THREADS
#define DRX_ITEM_THREADS 256
#endif
constexpr int kShareTriples = 16;
static inline int share_item_triples(int) { return kShareTriples; }
constexpr int kItemThreads = DRX_ITEM_THREADS;          // threads of a forward workgroup of the shared form (k_items_fwd_bwd)
static inline size_t share_item_lds_bytes(int ld) { return ((size_t)(kItemThreads / 64) * kShareTriples * ld + kShareTriples) * 4; }
static inline bool share_geometry_ok(int ld) {
  const Geom g = pick_geom(ld);
  return g.G >= 16 && g.J <= 2 && share_item_lds_bytes(ld) <= 64 * 1024;
}

// The batch's work items from the sorted (user, sample) pairs — the first B of the 2B sorted pairs; start[]: the first sorted position
// of every user's run (k_tp_runs).  A work item begins at every rt-th position of a run: flags -> inclusive scan (drx_scan.hpp) ->
// items, and their launch order by history length (a counting sort over 32 buckets of log2(length)).  (r04, first form: ONE workgroup
// numbering all B positions — 230 us, 395 us with the launch order — was the longest link of the preparation's chain.)
// n_du words: [0] items, [32 + b] items of bucket b, [64 + b] the scatter's cursor of bucket b.
static __global__ __launch_bounds__(256) void k_tp_item_flags(const uint32_t *__restrict__ ks, int B, int rt, const int32_t *__restrict__ start,
                                                              int32_t *__restrict__ pitem, int32_t *n_du) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 128) n_du[threadIdx.x] = 0;
  if (p < B) pitem[p] = ((p - start[ks[p]]) % rt == 0) ? 1 : 0;
}
// bucket 0: the longest histories (2^30 ..), bucket 31: one row or none
__device__ __forceinline__ int tp_bucket_of(const int64_t *__restrict__ indptr, uint32_t u) {
  const long long d = (long long)(indptr[u + 1] - indptr[u]);
  return __clz((int)min(d, 0x7FFFFFFFll) | 1);
}
// pitem holds the inclusive scan of the flags.  (The bucket counts are summed per workgroup in LDS first: 6 000 atomics on half a dozen
// words took 40 us in each of the two kernels.)
static __global__ __launch_bounds__(256) void k_tp_item_finish(const uint32_t *__restrict__ ks, const uint32_t *__restrict__ vs, int B, int rt,
                                                               const int32_t *__restrict__ start, const int32_t *__restrict__ end,
                                                               const int64_t *__restrict__ indptr, int32_t *__restrict__ usamp,
                                                               int32_t *__restrict__ pitem, WorkItem *__restrict__ witem, int32_t *n_du,
                                                               uint32_t *plan_cnt) {
  __shared__ int lcnt[32];
  if (threadIdx.x < 32) lcnt[threadIdx.x] = 0;
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < B) {
    const int item = pitem[p] - 1;
    pitem[p] = item;
    usamp[p] = (int32_t)vs[p];
    const uint32_t u = ks[p];
    if ((p - start[u]) % rt == 0) {
      const long long hs = indptr[u], he = indptr[u + 1];
      witem[item] = WorkItem{p, min(rt, end[u] - p), (int32_t)u, (int32_t)min(he - hs, 0x7FFFFFFFll), hs, 0};
      atomicAdd(&lcnt[tp_bucket_of(indptr, u)], 1);
    }
    if (p == B - 1) { n_du[0] = item + 1; plan_cnt[21] = (uint32_t)(item + 1); }     // (SpanPlan::cnt[21]: place_block)
  }
  __syncthreads();
  if (threadIdx.x < 32 && lcnt[threadIdx.x]) atomicAdd(&n_du[32 + threadIdx.x], lcnt[threadIdx.x]);
}
// the launch order of the forward workgroups: which workgroup takes which item never changes a result (inside a bucket: any order)
static __global__ __launch_bounds__(256) void k_tp_item_order(const uint32_t *__restrict__ ks, int B, int rt, const int32_t *__restrict__ start,
                                                              const int64_t *__restrict__ indptr, const int32_t *__restrict__ pitem,
                                                              int32_t *__restrict__ worder, int32_t *n_du) {
  __shared__ int lcnt[32], lbase[32];
  if (threadIdx.x < 32) lcnt[threadIdx.x] = 0;
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool head = p < B && (p - start[ks[p]]) % rt == 0;
  int b = 0, local = 0;
  if (head) { b = tp_bucket_of(indptr, ks[p]); local = atomicAdd(&lcnt[b], 1); }
  __syncthreads();
  if (threadIdx.x < 32) {                                  // this workgroup's range inside bucket t: behind the buckets before it
    const int t = threadIdx.x;
    int base = 0;
    for (int x = 0; x < t; ++x) base += n_du[32 + x];
    lbase[t] = lcnt[t] ? base + atomicAdd(&n_du[64 + t], lcnt[t]) : 0;
  }
  __syncthreads();
  if (head) worder[lbase[b] + local] = pitem[p];
}

// The W part of the list in two passes over the history of the batch's users and a scan between them.
// SHARE (DRX_BATCH_SHARE_USERS): per work item of the entry's user (rt samples of the user's run), ONE touch of the item's summed
// gradient row (sample field B + the work item) and one touch per sample of the item that DROPPED the entry (sample field | 0x80000000:
// the reduction subtracts those) — where that is the shorter form; otherwise, as without the flag, one touch per sample that kept it.
//
// COUNT, user-major: one WORKGROUP per user of the batch, its waves take the user's history 64 positions at a time, lane = position:
// every lane walks the SAME samples (uniform trip counts; r04, first form: one thread per transpose entry — neighbouring lanes held
// users of 1 .. 79 samples, a wave took as long as its longest).  Beside the count it leaves the keep bits of the user's first 64
// samples at this position (km): the write pass evaluates no mask for them.  Both land at the entry's item-major rank (t_rank).
template <bool SHARE>
static __global__ __launch_bounds__(256) void k_tp_count(DrxHistory H, DrxBatch bt, uint32_t qthr, int n_users, const int32_t *__restrict__ start,
                                                         const int32_t *__restrict__ end, const uint32_t *__restrict__ vs,
                                                         int *__restrict__ cnt, unsigned long long *__restrict__ km, int rt) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  for (int u = blockIdx.x; u < n_users; u += gridDim.x) {
    const int s0 = start[u], c = end[u] - s0;
    const int64_t hs = H.indptr[u];
    const int deg = (int)(H.indptr[u + 1] - hs);
    if (c <= 0) {                                          // a user without a sample in this batch: its entries hold no touch
      for (int j = wv * 64 + lane; j < deg; j += nwv * 64) cnt[H.t_rank[hs + j]] = 0;
      continue;
    }
    for (int j = wv * 64 + lane; j < deg; j += nwv * 64) {
      const int e = H.t_rank[hs + j];
      unsigned long long bits = 0ull;
      int kept = 0;
      for (int q0 = 0; q0 < c; q0 += rt) {                 // pieces of rt <= 32 samples (SHARE: the work items; else any cut will do)
        const int nq = min(c - q0, rt);
        uint32_t m = 0;
        for (int q = 0; q < nq; ++q) {
          const uint32_t b = vs[s0 + q0 + q];
          m |= ((bt.keep ? (bt.keep[bt.keep_off[b] + j] != 0) : (hash_u32(bt.mask_seed, b, (uint32_t)j) >= qthr)) ? 1u : 0u) << q;
        }
        if (q0 < 64) bits |= (unsigned long long)m << q0;
        const int k = __popc(m);
        kept += (SHARE && nq > 1 && 1 + (nq - k) < k) ? 1 + (nq - k) : k;      // the shorter form of the two
      }
      cnt[e] = kept;
      km[e] = bits;
    }
  }
}

// WRITE, item-major: one thread per entry of the item-major order (its user, position and item: DrxHistory::t_users / t_pos / t_items),
// cnt holds the exclusive scan: neighbouring threads write neighbouring runs of the list.  (r04, second form: the user-major walk
// wrote too — a lane's touches go to ITS item's segment, 5.8 M scattered 4-byte stores: 79 us alone against 20 for the count pass.)
template <bool SHARE>
static __global__ __launch_bounds__(256) void k_tp_write(DrxHistory H, DrxBatch bt, uint32_t qthr, const int32_t *__restrict__ start,
                                                         const int32_t *__restrict__ end, const uint32_t *__restrict__ vs,
                                                         const int *__restrict__ cnt, const unsigned long long *__restrict__ km,
                                                         uint32_t *__restrict__ keys_s, uint32_t *__restrict__ vals_s,
                                                         const int32_t *__restrict__ pitem, int rt, int32_t *__restrict__ row_end) {
  const int64_t nnz = H.t_nnz;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int u = H.t_users[e];
    const int s0 = start[u], c = end[u] - s0;
    const uint32_t n = (uint32_t)H.t_items[e];
    // where this item's segment ends (SpanPlan::row_end): the offset of the next item's first entry
    const bool last_of_item = e + 1 == nnz || (uint32_t)H.t_items[e + 1] != n;
    if (last_of_item && e + 1 < nnz) row_end[n] = cnt[e + 1];
    if (c <= 0) {
      if (last_of_item && e + 1 == nnz) row_end[n] = cnt[e];
      continue;
    }
    const unsigned long long bits = km[e];
    int at = cnt[e];
    for (int q0 = 0; q0 < c; q0 += rt) {
      const int nq = min(c - q0, rt);
      uint32_t m;
      if (q0 + nq <= 64) m = (uint32_t)(bits >> q0) & (nq >= 32 ? 0xFFFFFFFFu : ((1u << nq) - 1u));
      else {                                               // (a user with more than 64 samples in the batch: the rest is evaluated here)
        const int j = H.t_pos[e];
        m = 0;
        for (int q = 0; q < nq; ++q) {
          const uint32_t b = vs[s0 + q0 + q];
          m |= ((bt.keep ? (bt.keep[bt.keep_off[b] + j] != 0) : (hash_u32(bt.mask_seed, b, (uint32_t)j) >= qthr)) ? 1u : 0u) << q;
        }
      }
      const int k = __popc(m);
      const bool shared = SHARE && nq > 1 && 1 + (nq - k) < k;
      if (shared) { keys_s[at] = n; vals_s[at] = (uint32_t)bt.B + (uint32_t)pitem[s0 + q0]; ++at; }
      uint32_t em = shared ? (~m & (nq >= 32 ? 0xFFFFFFFFu : ((1u << nq) - 1u))) : m;      // the samples that emit a touch
      while (em) {
        const int q = __ffs((int)em) - 1;
        em &= em - 1u;
        const uint32_t b = vs[s0 + q0 + q];
        keys_s[at] = n; vals_s[at] = shared ? (b | 0x80000000u) : b; ++at;      // top bit: subtracted
      }
    }
    if (last_of_item && e + 1 == nnz) row_end[n] = at;
  }
}

// the W2T and V parts (the sorted batch pairs themselves) behind the W part, and DRX_KEY_NONE up to T
static __global__ __launch_bounds__(256) void k_tp_tail(const uint32_t *__restrict__ ks, const uint32_t *__restrict__ vs, int B, int n_users,
                                                        int n_items, const int *__restrict__ off_last, const int *__restrict__ cnt_last, int T,
                                                        uint32_t *__restrict__ keys_s, uint32_t *__restrict__ vals_s, uint32_t *plan_cnt) {
  const int Tw = off_last[0] + cnt_last[0];          // (exclusive scan: the last entry's offset + its count)
  if (blockIdx.x == 0 && threadIdx.x == 0) plan_cnt[20] = (uint32_t)(Tw + 2 * B);      // the list's real length (SpanPlan::cnt[20])
  // (the list says how long it is: behind its 2B pairs only the chunks a reader of its last block may look into are blanked)
  // (the reduction reads no slot behind cnt[20] — k_seg_reduce_planned's Tn; the planning kernels look one key past it — the tail is
  // sized for the widest workgroup all the same: kSegBlock / 4 chunks at 4 lanes per row)
  const int n_tail = min(T - Tw, 2 * B + (kSegBlock / 4 + 1) * kChunkLong);
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_tail; p += gridDim.x * blockDim.x) {
    uint32_t k = DRX_KEY_NONE, v = 0;
    if (p < B) { k = (uint32_t)n_items + (ks[B + p] - (uint32_t)n_users); v = vs[B + p] - (uint32_t)B; }        // items: sorted positions B .. 2B-1
    else if (p < 2 * B) { k = 2u * (uint32_t)n_items + ks[p - B]; v = vs[p - B]; }                              // users: positions 0 .. B-1
    keys_s[Tw + p] = k; vals_s[Tw + p] = v;
  }
}

static int plan_zero_words(const PrepBufs &R);
static void order_by_degree(const DrxBatch *bt, const PrepBufs &R, hipStream_t st, bool cleared);

// does the transposed preparation apply to this batch (and fit its work areas)?  The step asks the same question (share_users below).
static bool transposed_applies(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const PrepBufs &R) {
  if (!hist->t_rank || !hist->t_users || !hist->t_pos || !hist->t_items || !long_segments(R.T, *p)) return false;
  const int B = bt->B, U = p->n_users, N = p->n_items;
  const int64_t nnz = hist->t_nnz;
  if (nnz < 1 || nnz > (int64_t)R.T || nnz >= (1ll << 30) || (int64_t)U + N >= 0x7FFFFFFFll) return false;
  Carver cw(nullptr, R.sort_bytes);
  for (int i = 0; i < 4; ++i) (void)cw.take<uint32_t>((size_t)2 * B);
  (void)cw.take<int32_t>((size_t)U + N); (void)cw.take<int32_t>((size_t)U + N);
  (void)cw.take<char>(sort_pairs_temp_bytes((size_t)2 * B, bits_for((uint64_t)U + (uint64_t)N + 1)));
  (void)cw.take<char>(scan_i32_temp_bytes((size_t)nnz));
  (void)cw.take<char>(scan_i32_temp_bytes((size_t)B));
  (void)cw.take<unsigned long long>((size_t)nnz);
  (void)cw.take<int32_t>((size_t)N);
  return cw.ok();
}
// DRX_BATCH_SHARE_USERS takes effect: the list is in the shared form, the step forms S_u / D_u
static bool share_users(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const PrepBufs &R) {
  return (bt->flags & DRX_BATCH_SHARE_USERS) != 0 && R.pitem != nullptr && share_geometry_ok(p->ld) &&
         transposed_applies(p, hist, bt, R);
}

// allow_share: the list is prepared AHEAD of its step (drx_cdae_sparse_prepare) — a step that builds its list inline runs the plain
// forward kernel and must get the plain list
static int prepare_transposed(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const PrepBufs &R, hipStream_t st,
                              bool allow_share, const int32_t **row_end_out) {
  const int B = bt->B, U = p->n_users, N = p->n_items;
  if (!transposed_applies(p, hist, bt, R)) return kTpFallback;
  const bool share = allow_share && share_users(p, hist, bt, R);
  const int64_t nnz = hist->t_nnz;
  if (nnz < 1 || nnz > (int64_t)R.T || nnz >= (1ll << 30) || (int64_t)U + N >= 0x7FFFFFFFll) return kTpFallback;
  // work areas: cnt in the big sort's key buffer, the rest in its temp
  int *cnt = (int *)R.keys;
  Carver cw(R.sort_temp, R.sort_bytes);
  const int bits2 = bits_for((uint64_t)U + (uint64_t)N + 1);
  uint32_t *k2 = cw.take<uint32_t>((size_t)2 * B), *v2 = cw.take<uint32_t>((size_t)2 * B);
  uint32_t *ks = cw.take<uint32_t>((size_t)2 * B), *vs = cw.take<uint32_t>((size_t)2 * B);
  int32_t *start = cw.take<int32_t>((size_t)U + N), *end = start + ((size_t)U + N);
  (void)cw.take<int32_t>((size_t)U + N);
  const size_t sb = sort_pairs_temp_bytes((size_t)2 * B, bits2);
  void *stemp = cw.take<char>(sb);
  const size_t scb = scan_i32_temp_bytes((size_t)nnz);
  void *sctemp = cw.take<char>(scb);
  const size_t scb2 = scan_i32_temp_bytes((size_t)B);
  void *sctemp2 = cw.take<char>(scb2);
  unsigned long long *km = cw.take<unsigned long long>((size_t)nnz);
  int32_t *row_end = cw.take<int32_t>((size_t)N);
  if (!cw.ok()) return kTpFallback;
  *row_end_out = row_end;
  const uint32_t qthr = q_threshold(bt->q);
  hipLaunchKernelGGL(k_tp_begin, dim3(512), dim3(256), 0, st, *bt, U, k2, v2, start, 2 * (U + N), R.solo_v, R.plan.cnt, plan_zero_words(R),
                     R.order_work, 512);
  int rc = sort_pairs(stemp, sb, k2, ks, v2, vs, (size_t)2 * B, bits2, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tp_runs, dim3((2 * B + 255) / 256 < 1024 ? (2 * B + 255) / 256 : 1024), dim3(256), 0, st, ks, 2 * B, start, end);
  const int egrid = U < (1 << 20) ? U : (1 << 20);      // count pass: one workgroup per user
  const int wgrid = (int)((nnz + 255) / 256 < 16384 ? (nnz + 255) / 256 : 16384);      // write pass: one thread per entry
  const int rt = share_item_triples(p->ld);
  if (share) {
    const int ig = (B + 255) / 256;
    hipLaunchKernelGGL(k_tp_item_flags, dim3(ig), dim3(256), 0, st, ks, B, rt, start, R.pitem, R.n_du);
    rc = scan_i32(sctemp2, scb2, R.pitem, R.pitem, (size_t)B, true, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tp_item_finish, dim3(ig), dim3(256), 0, st, ks, vs, B, rt, start, end, hist->indptr, R.usamp, R.pitem, R.witem,
                       R.n_du, R.plan.cnt);
    hipLaunchKernelGGL(k_tp_item_order, dim3(ig), dim3(256), 0, st, ks, B, rt, start, hist->indptr, R.pitem, R.worder, R.n_du);
    hipLaunchKernelGGL((k_tp_count<true>), dim3(egrid), dim3(256), 0, st, *hist, *bt, qthr, U, start, end, vs, cnt, km, rt);
  } else
    hipLaunchKernelGGL((k_tp_count<false>), dim3(egrid), dim3(256), 0, st, *hist, *bt, qthr, U, start, end, vs, cnt, km, rt);
  // (the counts of the LAST entry are needed after the scan overwrote them: a copy)
  int *cnt_last = (int *)R.vals;
  DRX_HIP(hipMemcpyAsync(cnt_last, cnt + (nnz - 1), sizeof(int), hipMemcpyDeviceToDevice, st));
  rc = scan_i32(sctemp, scb, cnt, cnt, (size_t)nnz, false, st);
  if (rc) return rc;
  if (share)
    hipLaunchKernelGGL((k_tp_write<true>), dim3(wgrid), dim3(256), 0, st, *hist, *bt, qthr, start, end, vs, cnt, km, R.keys_s, R.vals_s, R.pitem, rt, row_end);
  else
    hipLaunchKernelGGL((k_tp_write<false>), dim3(wgrid), dim3(256), 0, st, *hist, *bt, qthr, start, end, vs, cnt, km, R.keys_s, R.vals_s, R.pitem, rt, row_end);
  hipLaunchKernelGGL(k_tp_tail, dim3(2048), dim3(256), 0, st, ks, vs, B, U, N, cnt + (nnz - 1), cnt_last, R.T, R.keys_s, R.vals_s, R.plan.cnt);
  return DRX_OK;
}

static int prepare_impl(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const PrepBufs &R, hipStream_t st,
                        bool with_marks = false, TouchPresence pres = TouchPresence{nullptr, WireGeo{1, 0, 1}}) {
  const int gpb = kBlock / 16;
  if (hist->t_rank && hist->t_users && hist->t_pos && hist->t_items && !pres.present && long_segments(R.T, *p)) {
    const int32_t *row_end = nullptr;
    const int rc = prepare_transposed(p, hist, bt, R, st, with_marks, &row_end);
    if (rc == DRX_OK) {                                        // the list stands, sorted: what is left is what follows the sort below
      PrepBufs Rb = R;
      Rb.plan.row_end = row_end;                               // (where every input row's segment ends: plan_chunk)
      Rb.plan.row_end_keys = (uint32_t)p->n_items;
      if (with_marks && p->ld > 16) {
        // (the launch order by history length is the plain forward kernel's: the shared form has its own, by work item)
        if (!share_users(p, hist, bt, R)) order_by_degree(bt, R, st, true);
        const int blocks = std::max(2048, (R.n_chunks + 255) / 256);
        hipLaunchKernelGGL(k_plan_and_mark, dim3(blocks), dim3(256), 0, st, R.keys_s, R.vals_s, R.T, R.n_chunks,
                           kSegBlock / pick_geom(p->ld).G, Rb.plan, (uint32_t)p->n_items, bt->B, R.solo_v, R.solo_o, 0, bt->keep_off,
                           R.order_work, R.order);
        if (R.plan.xrank) hipLaunchKernelGGL(k_place_blocks, dim3(256), dim3(256), 0, st, Rb.plan, R.n_chunks);
        return DRX_OK;
      }
      return plan_spans(p, bt, Rb, st, true);
    }
    if (rc != kTpFallback) return rc;
  }
  uint32_t *sort_zero = nullptr;
  size_t sort_zero_words = 0;
  sort_pairs_zero_region(R.sort_temp, (size_t)R.T, R.bits, &sort_zero, &sort_zero_words);
  hipLaunchKernelGGL(k_sparse_touches, dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, p->n_items, *hist, *bt,
                     q_threshold(bt->q), R.keys, R.vals, R.T, R.solo_v, R.plan.cnt, plan_zero_words(R), R.order_work, 512,
                     sort_zero, (int)sort_zero_words, pres);
  // the launch order (see k_degree_counts): its counts ride in the sort's first launch, its scatter in the plan + marks launch below
  const bool fused_order = with_marks && p->ld > 16;
  const SortRider rider{fused_order ? bt->keep_off : nullptr, bt->B, R.order_work};
  // dropped inputs (DRX_KEY_NONE) take no part in the sort: its last pass writes them back behind the sorted touches
  const int rc = sort_pairs_ex(R.sort_temp, R.sort_bytes, R.keys, R.keys_s, R.vals, R.vals_s, (size_t)R.T, R.bits, true, st, true, rider);
  if (rc) return rc;
  // rows of <= 16 floats (K = 128 sharded over 8 GPUs): a 64-byte random read-modify-write in the forward kernel costs more than
  // the segmented reduction saves (measured 1.018 vs 0.995 ms per step); no marks = no fusion
  if (with_marks && p->ld > 16) {
    const int blocks = std::max(2048, (R.n_chunks + 255) / 256);
    const int order_blocks = (bt->B + 255) / 256;
    hipLaunchKernelGGL(k_plan_and_mark, dim3(blocks + order_blocks), dim3(256), 0, st, R.keys_s, R.vals_s, R.T, R.n_chunks,
                       kSegBlock / pick_geom(p->ld).G, R.plan, (uint32_t)p->n_items, bt->B, R.solo_v, R.solo_o,
                       order_blocks, bt->keep_off, R.order_work, R.order);
    if (R.plan.xrank) hipLaunchKernelGGL(k_place_blocks, dim3(256), dim3(256), 0, st, R.plan, R.n_chunks);
    return DRX_OK;
  }
  return plan_spans(p, bt, R, st, true);
}

// Only for touch lists prepared AHEAD of the step (the forward kernel must see the marks): see k_mark_solo.
static int mark_solo(const DrxCdaeParams *p, const DrxBatch *bt, const PrepBufs &R, hipStream_t st, bool cleared) {
  if (!cleared) {                                                                  // (prepare_impl's touch kernel clears them)
    DRX_HIP(hipMemsetAsync(R.solo_v, 0, (size_t)bt->B * 2, st));
  }
  // rows of <= 16 floats (K = 128 sharded over 8 GPUs): a 64-byte random read-modify-write in the forward kernel costs more than
  // the segmented reduction saves (measured 1.018 vs 0.995 ms per step); no marks = no fusion
  if (p->ld <= 16) return DRX_OK;       // solo_v and solo_o are adjacent
  hipLaunchKernelGGL(k_mark_solo, dim3(2048), dim3(256), 0, st, R.keys_s, R.vals_s, R.T, (uint32_t)p->n_items, bt->B, R.solo_v,
                     R.solo_o);
  return DRX_OK;
}


}  // namespace drx
