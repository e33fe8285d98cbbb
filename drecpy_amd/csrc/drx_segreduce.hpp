// Segmented reduction over a key-sorted touch list, shared by the single-GPU sparse step (drx_cdae.hip) and the
// row-sharded multi-GPU step (drx_shard.hip).  A POLICY supplies where a touch's contribution row comes from and
// what happens to a finished segment (apply the optimizer to a table row, or park the gradient row for an exchange):
//
//   struct Policy {
//     // contribution of touch (key, val): row (float4[J] of this lane), scalar side-value, multiplier
//     template <int G, int J> __device__ void load(uint32_t key, uint32_t val, int lane, float4 (&row)[J], float &sc, float &coef) const;
//     // segment `key` is complete: g = sum coef*row, gs = sum sc; `pos` is the sorted position of one of its touches
//     template <int G, int J> __device__ void finish(uint32_t key, int pos, int lane, const float4 (&g)[J], float gs) const;
//   };
//
// Work split: fixed chunks of kChunk touches per group (load-balanced under Zipf skew); segments inside one chunk are
// finished in place, chunk-crossing segments leave partial rows that the two fix-up tiers combine in chunk order.
// Every sum has a fixed order => bit-reproducible results, no float atomics.
#pragma once
#include "drx_rows.hpp"

namespace drx {

#ifndef DRX_CHUNK
#define DRX_CHUNK 32
#endif
constexpr int kChunk = DRX_CHUNK;      // touches per group in the segmented reduction
// ... and for lists of LONG segments (planned variant only; SpanPlan::chunk): rows collect thousands of touches, longer chunks halve
// the partial rows and the workgroups (ml-1m shape: 0.685 -> 0.630 ms per step; the headline's short segments: 0.346 -> 0.375 with 64)
constexpr int kChunkLong = 64;
static inline int seg_chunk(bool long_segments) { return long_segments ? kChunkLong : kChunk; }
#ifdef DRX_SEG_WAVE_PER_CHUNK
#define SEG_GPB(G) (drx::kBlock / 64)
#else
#define SEG_GPB(G) (drx::kBlock / (G))
#endif
constexpr int kShortSpan = 64;     // chunk borders a segment may cross and still be combined by one group
#ifndef DRX_SEG_BLOCK
#define DRX_SEG_BLOCK 128
#endif
constexpr int kSegBlock = DRX_SEG_BLOCK;   // threads per workgroup of k_seg_reduce_planned (a workgroup keeps its LDS and its slot until its
                                   //   slowest chunk is done: smaller workgroups pack uneven chunks better, larger ones fold longer runs)
constexpr int kPlanShort = 16;     // planned variant: spans of up to this many partial rows are combined by ONE group (two rounds of 8 loads in
                                   //   flight), longer ones by a workgroup; 64 left the second-hottest rows to a single group each: 8 rounds
constexpr int kFixBlock = 512;     // (1024: the long-span fix-up kernel hit the 128-VGPR cap of a 16-wave workgroup and spilled)

struct SegBufs {
  const uint32_t *keys_s, *vals_s;            // [T] sorted touches (padding keys DRX_KEY_NONE sort last)
  float *phead, *ptail;                       // [n_chunks, ld]
  float *phs, *pts;                           // [n_chunks] scalar partials
  uint32_t *span_list, *long_list;            // [n_chunks] each
  uint32_t *n_span;                           // [0] crossing segments, [1] long ones (zeroed by the caller)
  uint8_t *cflag;                             // [n_chunks] 0: chunk's first segment starts here; 2: the whole chunk is the
                                              //   middle of one crossing segment; 1: it starts with the END of one
  int T, n_chunks, ld;
  unsigned long long *stamps;                 // diagnostic builds (DRX_STAMPS) only; nullptr otherwise
};

// Segmented reduction over the sorted touch list in fixed chunks of kChunk touches per group.
// Segments that lie inside one chunk are updated here; segments crossing chunk borders leave
// partial rows that k_span_fixup combines in chunk order (deterministic).
// The chunk's (key, sample) pairs are fetched with one coalesced load per lane and broadcast by shuffles; the
// contribution rows are then loaded LB at a time (independent loads in flight) before they are folded in order.
// LB1 = contribution rows a group of one-float4-per-lane rows (J == 1) keeps in flight.  2 where segments are short (the 10M x 1M
// set: 0.09 touches per table row; 92 instead of 127 VGPRs, 5 waves per SIMD instead of 4: the reduction 0.215 -> 0.192 ms, the step
// 150 -> 160 M triples/s); 8 where a row collects hundreds of touches (the MovieLens shapes: +5 % there).
template <int G, int J, class Policy, int LB1 = 2>
__global__ __launch_bounds__(kBlock) void k_seg_reduce(SegBufs S, Policy pol) {
#ifdef DRX_SEG_WAVE_PER_CHUNK
  // experiment: one chunk per WAVE (lanes >= G idle) so that a group's flush never stalls a sibling group
  const int lane = threadIdx.x % 64;
  if (lane >= G) return;
  const int g = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
#else
  const int lane = threadIdx.x % G;
  const int g = blockIdx.x * (kBlock / G) + threadIdx.x / G;
#endif
  if (g >= S.n_chunks) return;
  const int start = g * kChunk, end = min(S.T, start + kChunk);
  const int n = end - start;
  const uint32_t prev_key = start > 0 ? S.keys_s[start - 1] : DRX_KEY_NONE;
  const uint32_t next_key = end < S.T ? S.keys_s[end] : DRX_KEY_NONE;
  constexpr int KPL = (kChunk + G - 1) / G;          // (key, val) registers per lane
  constexpr int LB = J == 1 ? LB1 : (J == 2 ? 4 : 2);  // rows in flight per group
  uint32_t kreg[KPL], vreg[KPL];
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    const int t = r * G + lane;
    const bool ok = t < n && t < kChunk;
    kreg[r] = ok ? S.keys_s[start + t] : DRX_KEY_NONE;
    vreg[r] = ok ? S.vals_s[start + t] : 0u;
  }
  auto bcast = [&](const uint32_t (&reg)[KPL], int t) -> uint32_t {
    uint32_t sel = reg[0];
#pragma unroll
    for (int r = 1; r < KPL; ++r) sel = (t / G == r) ? reg[r] : sel;
    return (uint32_t)__shfl((int)sel, t % G, G);
  };
  float4 acc[J];
#pragma unroll
  for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
  float accs = 0.f;
  uint32_t cur = DRX_KEY_NONE;
  int cur_pos = 0;
  bool cur_from_start = false;
  auto flush = [&](bool at_end) {
    if (cur == DRX_KEY_NONE) return;
    const bool cont_left = cur_from_start && prev_key == cur;
    const bool cont_right = at_end && next_key == cur;
    if (!cont_left && !cont_right) {
      pol.template finish<G, J>(cur, cur_pos, lane, acc, accs);
    } else if (cont_left) {
      store_row<G, J>(S.phead, (size_t)g, S.ld, lane, acc);
      if (lane == 0) S.phs[g] = accs;
    } else {
      store_row<G, J>(S.ptail, (size_t)g, S.ld, lane, acc);
      if (lane == 0) {
        S.pts[g] = accs;
        const uint32_t slot = atomicAdd(S.n_span, 1u);
        S.span_list[slot] = (uint32_t)g;
      }
    }
  };
  for (int t0 = 0; t0 < n; t0 += LB) {
    uint32_t k8[LB];
    float s8[LB], c8[LB];
    float4 rows[LB][J];
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int t = t0 + u;
      k8[u] = t < n ? bcast(kreg, t) : DRX_KEY_NONE;    // padding (dropped inputs) sorts last
      const uint32_t b = bcast(vreg, t < n ? t : 0);
      s8[u] = 0.f;
      c8[u] = 1.f;
#pragma unroll
      for (int jx = 0; jx < J; ++jx) rows[u][jx] = f4_zero();
      if (k8[u] != DRX_KEY_NONE) pol.template load<G, J>(k8[u], b, lane, rows[u], s8[u], c8[u]);
    }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const uint32_t key = k8[u];
      if (key != DRX_KEY_NONE) {
        if (key != cur) {
          flush(false);
          cur = key;
          cur_from_start = (t0 + u == 0);
#pragma unroll
          for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
          accs = 0.f;
        }
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_fma(acc[jx], c8[u], rows[u][jx]);
        accs += s8[u];
        cur_pos = start + t0 + u;
      }
    }
  }
  // the last segment ends at the chunk border iff the final touch of the chunk is a real key
  const bool ran_to_end = n > 0 && bcast(kreg, n - 1) != DRX_KEY_NONE;
  flush(ran_to_end);
  if (lane == 0) {
    const uint32_t first = n > 0 ? S.keys_s[start] : DRX_KEY_NONE, last = n > 0 ? S.keys_s[end - 1] : DRX_KEY_NONE;
    const bool cont = first != DRX_KEY_NONE && first == prev_key;
    const bool middle = cont && last == first && next_key == first;
    S.cflag[g] = middle ? 2 : (cont ? 1 : 0);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// PLANNED variant (the CDAE sparse step, r03): everything about the list's STRUCTURE — which segments cross chunk borders, how far,
// which of them are long — depends on the sorted keys alone, so it is worked out when the list is prepared (k_plan_spans, on the
// preparation's stream, ahead of the step) instead of between the training kernels:
//   * k_seg_reduce_planned writes no chunk flags and appends to no list; a workgroup whose chunks ALL lie inside one segment that
//     began before it ("all-inner": the middle of a hot row) folds its kBlock/G chunk sums in LDS, in chunk order, into ONE block
//     partial — a row with 50 000 touches leaves 200 partial rows instead of 1 600;
//   * a segment that starts in one chunk and ends inside the NEXT one is simply finished by the chunk it starts in (SpanPlan::ext: that
//     chunk's window reaches ext touches into its neighbour's, whose window starts behind them) — r02 tried this with the lengths found
//     at run time and lost to the extra dependent loads; here they are one byte per chunk, known beforehand;
//   * k_span_planned combines short and long spans in ONE launch (their lengths are known: no flag probing, no second tier queued by
//     the first) — the step's tail is one launch instead of two.
// Sum order of a crossing segment: tail partial of its first chunk, then its partials in chunk order, a block partial being the
// in-order sum of its chunks.  Fixed by the list alone: bit-reproducible.
struct SpanPlan {
  uint2 *desc;          // [n_chunks]: short spans from the front, long spans from the back; x = first chunk g0, y = m_in | has_end << 31
  uint32_t *cnt;        // [0] short spans, [1] long spans
  uint8_t *ext;         // [n_chunks] (zeroed before k_plan_spans): chunk g also takes the first ext[g] touches of chunk g + 1 — the end of
                        //   a segment that starts in g and ends inside g + 1 (the commonest crossing by far: no partial rows, no span)
  // XCD PLACEMENT (r04; lists of LONG segments only — MovieLens shapes: a few thousand rows, thousands of touches each).  The touches
  // of a segment are sorted by sample, so a workgroup's chunks read a narrow band of the batch's gradient rows (dz1 / g2: 33 MB at
  // B = 65 536, eight times an XCD's L2).  The workgroup-sized blocks of chunks are binned by the 64th of the batch their first touch's
  // sample lies in (cnt[kXBase + t] blocks in bin t; xrank[blk] = bin << 24-bit rank inside it) and xperm lists them bin by bin.  The
  // hardware deals workgroups to the 8 XCDs round robin: a workgroup of residue x takes the blocks of bins 8x .. 8x + 7 in that order,
  // so the workgroups resident on an XCD at a time re-read ONE 64th of the gradient rows (0.5 MB: its L2) instead of all of them.
  // Which workgroup sums which block never changes a result.  nullptr: blocks in list order.
  uint32_t *xrank;      // [xstride] block -> bin << 24 | rank      (k_plan_spans / k_plan_and_mark)
  uint32_t *xperm;      // [xstride] slot -> block                  (k_place_blocks)
  int xstride;          // entries (the number of blocks of the smallest block size: a layout independent of the row width)
  int chunk;            // touches per chunk of THIS list: seg_chunk(long segments?)
  // lists laid down by the transposed preparation know where the segment of every INPUT row ends (plan_chunk then looks the end of a
  // crossing segment up instead of searching for it: 22 dependent probes were the planning kernel's longest chain); nullptr: search
  const int32_t *row_end;     // [row_end_keys] list position behind the last touch of key k
  uint32_t row_end_keys;
};
// r06 experiment, OFF: lists of SHORT segments (the headline) placing the blocks that lie inside a hot row by the XCD whose band of the
// gradient rows their samples fall in.  scripts/mb/mb_l2band.hip had promised 157 -> ~145 us for the streamed reduction; built
// (-DDRX_STREAM_PLACED=1) it measured 170 - 172 us against 160 - 161 in list order, and 163 with the placement's lookups paid but the
// blocks left in list order (-DDRX_STREAM_PLACED=2) [profiles/r06_stream_placement_ab.log]: a 4.2 MB band does not stay in a 4 MB L2
// that also streams the parameter / slot rows and the other 55 % of the gathers, and the binned order costs balance.
#ifndef DRX_STREAM_PLACED
#define DRX_STREAM_PLACED 0
#endif
#ifndef DRX_XBINS
#define DRX_XBINS 8      // (64 — bins of 0.5 MB of gradient rows, taken in order — measured SLOWER at the ml-1m shape: 0.60 against 0.54 ms)
#endif
constexpr int kXBase = 32, kXBins = DRX_XBINS;     // SpanPlan::cnt[kXBase ..]: the bins' counts; cnt[16] != 0: placed; cnt[17]: the block size

// block `blk` (chunks blk * cpb ...) joins the bin of the 64th of the batch its first touch's sample lies in.  Called by whole waves
// (`on`: this lane has a block): the lanes of a wave that chose the same bin take their ranks with ONE atomic (single atomics on a
// handful of counters took the planning kernel from 30 to 480 us and slowed everything beside it).
__device__ __forceinline__ void place_block(const uint32_t *__restrict__ keys_s, const uint32_t *__restrict__ vals_s, int T, int cpb, int B,
                                            const SpanPlan &P, int blk, bool on) {
  int x = blk & (kXBins - 1);
  // (a list that says how long it is — cnt[20], the transposed preparation's — places only the blocks that hold touches: the others
  // have no workgroup to wait for, and every rank taken is an atomic on one of kXBins counters)
  if (on && P.cnt[20] != 0u && (long long)blk * cpb * P.chunk >= (long long)P.cnt[20]) on = false;
  // Lists of SHORT segments (r06; the 10M x 1M headline: 4 touches per distinct row on average, but the few hundred HOT rows collect
  // 45 % of the touches): only the blocks that lie wholly INSIDE one segment — the middle of a hot row, whose touches are sample-
  // ascending — read a narrow band of the gradient rows and are worth a bin; every other block mixes samples from everywhere and keeps
  // the bin of its index (scripts/mb/mb_l2band.hip: gathers out of an XCD-sized window run at 14 - 19 TB/s instead of 7).
  bool banded = true;
  if (on && P.chunk == kChunk) {
    const long long first = (long long)blk * cpb * P.chunk, last = first + (long long)cpb * P.chunk - 1;
    banded = first > 0 && last < T && keys_s[first] != DRX_KEY_NONE && keys_s[first - 1] == keys_s[first] && keys_s[last] == keys_s[first];
  }
  if (on && banded) {
    const int at = blk * cpb * P.chunk;
    if (at < T && keys_s[at] != DRX_KEY_NONE) {                 // (a blanked or dropped touch carries no sample: any bin will do)
      uint32_t b = vals_s[at];
      // (DRX_BATCH_SHARE_USERS: a subtracted touch carries its sample under the top bit; a touch of a user's summed row carries B + the
      // user's slot — slots ascend with the samples of a by-user batch: cnt[21] distinct users)
      if (b & 0x80000000u) b &= 0x7FFFFFFFu;
      else if (b >= (uint32_t)B) b = (uint32_t)(((unsigned long long)(b - (uint32_t)B) * (unsigned long long)B) / (unsigned long long)max(1u, P.cnt[21]));
      x = (int)min((uint32_t)(kXBins - 1), (uint32_t)(((unsigned long long)b * (unsigned long long)kXBins) / (unsigned long long)(B > 0 ? B : 1)));
    }
  }
  // the wave's blocks bin by bin: ONE atomic per bin present, all of them in flight together (lane r takes the ranks of bin r) — one
  // round trip instead of one per bin, one after the other
  const int lane = threadIdx.x & 63;
  unsigned long long mine = 0ull, of_lane = 0ull;
#pragma unroll
  for (int r = 0; r < kXBins; ++r) {
    const unsigned long long m = __ballot(on && x == r);
    if (x == r) mine = m;
    if (lane == r) of_lane = m;
  }
  uint32_t base = 0;
  if (lane < kXBins && of_lane) base = atomicAdd(&P.cnt[kXBase + lane], (uint32_t)__popcll(of_lane));
  base = (uint32_t)__shfl((int)base, x);
  if (on) P.xrank[blk] = ((uint32_t)x << 24) | (base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull)));
  if (on && blk == 0) { P.cnt[16] = 1u; P.cnt[17] = (uint32_t)cpb; }
}

// after the plan: slot -> block, bin by bin (one more small launch on the preparation's stream, lists of long segments only)
static __global__ __launch_bounds__(256) void k_place_blocks(SpanPlan P, int n_chunks) {
  __shared__ uint32_t pre[kXBins];
  if (threadIdx.x < kXBins) {
    uint32_t run = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) run += P.cnt[kXBase + t];
    pre[threadIdx.x] = run;
  }
  __syncthreads();
  const int cpb = (int)P.cnt[17];
  const int nb = cpb > 0 ? (n_chunks + cpb - 1) / cpb : 0;
  for (int blk = blockIdx.x * blockDim.x + threadIdx.x; blk < nb; blk += gridDim.x * blockDim.x) {
    if (P.cnt[20] != 0u && (long long)blk * cpb * P.chunk >= (long long)P.cnt[20]) continue;      // (not placed: see place_block)
    const uint32_t e = P.xrank[blk];
    P.xperm[pre[e >> 24] + (e & 0xFFFFFFu)] = (uint32_t)blk;
  }
}

// the block a reduction workgroup takes: launch index j (its XCD = j % 8), `extra` workgroups of another role in front, n_wg of this one
__device__ __forceinline__ int placed_block(const SpanPlan &P, int j, int extra, int n_wg, int cpb) {
  // (cnt[17]: the block size the bins were made for — a list prepared by a rank of another row width is taken in list order)
  if (!P.xperm || P.cnt[16] == 0u || P.cnt[17] != (uint32_t)cpb) return j - extra;
  const int x = j & 7;
  int own[8], q[8], have[8], start[8];
  int run = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int c = 0;
#pragma unroll
    for (int t = 0; t < kXBins / 8; ++t) c += (int)P.cnt[kXBase + r * (kXBins / 8) + t];
    have[r] = c; start[r] = run; run += c;
    const int first = extra + ((r - extra) & 7);                 // smallest launch index >= extra with residue r
    q[r] = first < extra + n_wg ? (extra + n_wg - 1 - first) / 8 + 1 : 0;
    own[r] = min(c, q[r]);
  }
  const int i = (j - (extra + ((x - extra) & 7))) / 8;
  if (i < own[x]) return (int)P.xperm[start[x] + i];
  int f = i - own[x];                                            // this workgroup's number among the FREE ones ...
#pragma unroll
  for (int r = 0; r < 8; ++r) f += r < x ? q[r] - own[r] : 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {                                  // ... takes the f-th block its XCD's workgroups could not
    const int over = have[r] - own[r];
    if (f < over) return (int)P.xperm[start[r] + own[r] + f];
    f -= over;
  }
  return -1;                       // more workgroups than placed blocks (a compact list's padding): nothing to do
}

struct PlanBufs {
  float *pblock;        // [n_blocks, ld] block partials (all-inner workgroups)
  float *pbs;           // [n_blocks]
};

// partials of a span after its first chunk: lead chunks up to the next workgroup boundary, whole all-inner workgroups, trailing chunks
struct SpanShape {
  int first, n_lead, n_blk, n_trail;
  __host__ __device__ SpanShape(uint2 d, int cpb) {
    const int m_in = (int)(d.y & 0x7FFFFFFFu), has_end = (int)(d.y >> 31);
    first = (int)d.x + 1;
    const int to_boundary = (cpb - first % cpb) % cpb;
    n_lead = m_in < to_boundary ? m_in : to_boundary;
    n_blk = (m_in - n_lead) / cpb;
    n_trail = m_in - n_lead - n_blk * cpb + has_end;
  }
  __host__ __device__ int total() const { return n_lead + n_blk + n_trail; }
};

__device__ __forceinline__ void plan_chunk(const uint32_t *__restrict__ keys_s, int T, int n_chunks, int cpb, const SpanPlan &P, int g) {
  const int CH = P.chunk;
  // (a compact list — SpanPlan::cnt[20], lists with placed blocks only — holds nothing behind its real length but a few chunks of padding)
  if (P.xrank && P.cnt[20] != 0u) T = min(T, (int)P.cnt[20]);
  const int start = g * CH, end = min(T, start + CH);
  if (start >= T) return;
  const uint32_t first = keys_s[start], last = keys_s[end - 1];
  const uint32_t prev = start > 0 ? keys_s[start - 1] : DRX_KEY_NONE, next = end < T ? keys_s[end] : DRX_KEY_NONE;
  const bool head_cont = first != DRX_KEY_NONE && prev == first;        // the run at the chunk's start began before it
  const bool tail_cont = last != DRX_KEY_NONE && next == last;          // the run at its end goes on behind it
  if (!tail_cont || (head_cont && first == last)) return;              // nothing STARTS to cross here
  int lo = end, hi = T;              // first position in [end, T) with another key
  if (P.row_end && last < P.row_end_keys) lo = hi = P.row_end[last];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys_s[mid] == last) lo = mid + 1; else hi = mid;
  }
  const int seg_end = lo;
  const int full_end = seg_end / CH;                           // chunks [g + 1, full_end) lie wholly inside the segment
  const uint32_t m_in = (uint32_t)(full_end - (g + 1)), has_end = (seg_end % CH) ? 1u : 0u;
  if (m_in == 0) {                                               // ends inside the next chunk: this chunk's window takes those touches
    P.ext[g] = (uint8_t)(seg_end - end);
    return;
  }
  const uint2 d = make_uint2((uint32_t)g, m_in | (has_end << 31));
  if (SpanShape(d, cpb).total() <= kPlanShort) P.desc[atomicAdd(&P.cnt[0], 1u)] = d;
  else P.desc[n_chunks - 1 - (int)atomicAdd(&P.cnt[1], 1u)] = d;
}

// One thread per chunk of the sorted list: a chunk whose last key continues into the next chunk and that is not itself the inside of
// that segment starts a span; the segment's end is found by binary search.
template <int DUMMY = 0>
__global__ void k_plan_spans(const uint32_t *__restrict__ keys_s, const uint32_t *__restrict__ vals_s, int T, int n_chunks, int cpb, int B,
                             SpanPlan P) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_chunks) plan_chunk(keys_s, T, n_chunks, cpb, P, g);
  const int nb = (n_chunks + cpb - 1) / cpb;
  if (P.xrank && (g & ~63) < nb) place_block(keys_s, vals_s, T, cpb, B, P, g, g < nb);
}

// The fold of ONE chunk window of a list of SHORT segments by a row group of G lanes (the planned reduction's plain form; the streamed
// kernel of drx_segstream.hpp runs it for the few windows it does not stream): the window's (key, sample) pairs are fetched with one
// coalesced load per lane and broadcast by shuffles; contribution rows go out LB at a time and are folded in list order.  Segments inside
// the window are finished in place; the run that began before the window / goes on behind it leaves a head / tail partial (no head
// partial in an all-inner workgroup: the caller folds its chunks' sums into one block partial).  acc / accs: the last run's sum.
template <int G, int J, int CH, int LB1, class Policy>
__device__ __forceinline__ void seg_fold_short(const SegBufs &S, const Policy &pol, int g, int lane, int start, int n, uint32_t prev_key,
                                               uint32_t next_key, bool all_inner, float4 (&acc)[J], float &accs) {
  constexpr int KPL = (2 * CH + G - 1) / G;
  constexpr int LB = J == 1 ? LB1 : (J == 2 ? 4 : 2);
  uint32_t kreg[KPL], vreg[KPL];
#pragma unroll
  for (int q = 0; q < KPL; ++q) {
    const int t = q * G + lane;
    const bool ok = t < n;
    kreg[q] = ok ? S.keys_s[start + t] : DRX_KEY_NONE;
    vreg[q] = ok ? S.vals_s[start + t] : 0u;
  }
  auto bcast = [&](const uint32_t (&reg)[KPL], int t) -> uint32_t {
    uint32_t sel = reg[0];
#pragma unroll
    for (int q = 1; q < KPL; ++q) sel = (t / G == q) ? reg[q] : sel;
    return (uint32_t)__shfl((int)sel, t % G, G);
  };
#pragma unroll
  for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
  accs = 0.f;
  uint32_t cur = DRX_KEY_NONE;
  int cur_pos = 0;
  bool cur_from_start = false;
  auto flush = [&](bool at_end) {
    if (cur == DRX_KEY_NONE) return;
    const bool cont_left = cur_from_start && prev_key == cur;
    const bool cont_right = at_end && next_key == cur;
    if (!cont_left && !cont_right) {
      pol.template finish<G, J>(cur, cur_pos, lane, acc, accs);
    } else if (cont_left) {
      if (!all_inner) {
        store_row<G, J>(S.phead, (size_t)g, S.ld, lane, acc);
        if (lane == 0) S.phs[g] = accs;
      }
    } else {
      store_row<G, J>(S.ptail, (size_t)g, S.ld, lane, acc);
      if (lane == 0) S.pts[g] = accs;
    }
  };
  for (int t0 = 0; t0 < n; t0 += LB) {
    uint32_t k8[LB];
    float s8[LB], c8[LB];
    float4 rows[LB][J];
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int t = t0 + u;
      k8[u] = t < n ? bcast(kreg, t) : DRX_KEY_NONE;
      const uint32_t b = bcast(vreg, t < n ? t : 0);
      s8[u] = 0.f;
      c8[u] = 1.f;
#pragma unroll
      for (int jx = 0; jx < J; ++jx) rows[u][jx] = f4_zero();
      if (k8[u] != DRX_KEY_NONE) pol.template load<G, J>(k8[u], b, lane, rows[u], s8[u], c8[u]);
    }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const uint32_t key = k8[u];
      if (key != DRX_KEY_NONE) {
        if (key != cur) {
          flush(false);
          cur = key;
          cur_from_start = (t0 + u == 0);
#pragma unroll
          for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
          accs = 0.f;
        }
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_fma(acc[jx], c8[u], rows[u][jx]);
        accs += s8[u];
        cur_pos = start + t0 + u;
      }
    }
  }
  flush(n > 0 && bcast(kreg, n - 1) != DRX_KEY_NONE);
}

// LDS: [kSegBlock/G, ld] floats + [kSegBlock/G] floats (+, LONG: 4 * CH words per chunk — seg_reduce_lds_bytes).  extra_blocks workgroups in front of the chunk workgroups run `extra(block)` (the
// CDAE step's bias column sums: independent work that fills the launch's ramp).
#ifdef DRX_SEGP_W8
#define DRX_SEGP_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define DRX_SEGP_ATTR
#endif
static inline size_t seg_reduce_lds_bytes(int cpb, int ld, bool long_segments) {
  return (size_t)((cpb * (ld + 1) + 3) & ~3) * 4 + (long_segments ? (size_t)cpb * 4 * kChunkLong * 4 : 0);
}
template <int G, int J, class Policy, bool LONG, class Extra>
__global__ __launch_bounds__(kSegBlock) DRX_SEGP_ATTR void k_seg_reduce_planned(SegBufs S, PlanBufs PB, SpanPlan SP, Policy pol,
                                                                            int extra_blocks, Extra extra) {
  extern __shared__ __align__(16) float seg_lds[];
  constexpr int CPB = kSegBlock / G;
  constexpr int CH = LONG ? kChunkLong : kChunk;          // (= SP.chunk: the list was laid out for it)
  constexpr int LB1 = LONG ? 8 : 2;
  if ((int)blockIdx.x < extra_blocks) { extra(seg_lds); return; }
  const uint8_t *__restrict__ ext = SP.ext;
  const int blk = placed_block(SP, (int)blockIdx.x, extra_blocks, (int)gridDim.x - extra_blocks, CPB);
  // (lists laid down compact — the transposed preparation — say how long they are: SpanPlan::cnt[20] real touches, the rest of the
  // T slots is padding; a workgroup whose chunks all lie in the padding has nothing to do)
  if (blk < 0 || (LONG && SP.cnt[20] != 0u && (long long)blk * CPB * CH >= (long long)SP.cnt[20] + CH)) return;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int g = blk * CPB + r;
  // (the slots behind a compact list's real length are NOT the list: only a few chunks of them are blanked, the rest is whatever the
  // buffer held before — every chunk reads the list up to its real end and no further)
  const int Tn = (LONG && SP.cnt[20] != 0u) ? min(S.T, (int)SP.cnt[20]) : S.T;
  bool inner = false;          // this chunk is one whole run of a segment that began before it
  if (g < S.n_chunks && (g + 1) * CH <= Tn) {
    const uint32_t f = S.keys_s[g * CH], l = S.keys_s[(g + 1) * CH - 1], pk = g > 0 ? S.keys_s[g * CH - 1] : DRX_KEY_NONE;
    inner = f != DRX_KEY_NONE && f == pk && l == f;
  }
  const bool all_inner = __syncthreads_and(inner ? 1 : 0) != 0;
  if (g >= S.n_chunks) return;          // (never in an all-inner workgroup: its second barrier below sees every thread)
  // this chunk's window: behind the touches its left neighbour finishes for it, and into the right neighbour's for the segment it
  // finishes itself (SpanPlan::ext); up to 2 * CH - 1 touches
  const int start = min(Tn, g * CH + (g > 0 ? (int)ext[g - 1] : 0)), end = min(Tn, (g + 1) * CH + (int)ext[g]);
  const int n = max(0, end - start);
  const uint32_t prev_key = start > 0 ? S.keys_s[start - 1] : DRX_KEY_NONE;
  const uint32_t next_key = end < Tn ? S.keys_s[end] : DRX_KEY_NONE;
  float4 acc[J];
#pragma unroll
  for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
  float accs = 0.f;
  if constexpr (LONG) {
    constexpr int KPL = (2 * CH + G - 1) / G;
    constexpr int LB = J == 1 ? LB1 : (J == 2 ? 4 : 2);
    uint32_t kreg[KPL], vreg[KPL];
  #pragma unroll
    for (int q = 0; q < KPL; ++q) {
      const int t = q * G + lane;
      const bool ok = t < n;
      kreg[q] = ok ? S.keys_s[start + t] : DRX_KEY_NONE;
      vreg[q] = ok ? S.vals_s[start + t] : 0u;
    }
    auto bcast = [&](const uint32_t (&reg)[KPL], int t) -> uint32_t {
      uint32_t sel = reg[0];
  #pragma unroll
      for (int q = 1; q < KPL; ++q) sel = (t / G == q) ? reg[q] : sel;
      return (uint32_t)__shfl((int)sel, t % G, G);
    };
    uint32_t cur = DRX_KEY_NONE;
    int cur_pos = 0;
    bool cur_from_start = false;
    auto flush = [&](bool at_end) {
      if (cur == DRX_KEY_NONE) return;
      const bool cont_left = cur_from_start && prev_key == cur;
      const bool cont_right = at_end && next_key == cur;
      if (!cont_left && !cont_right) {
        pol.template finish<G, J>(cur, cur_pos, lane, acc, accs);
      } else if (cont_left) {
        if (!all_inner) {
          store_row<G, J>(S.phead, (size_t)g, S.ld, lane, acc);
          if (lane == 0) S.phs[g] = accs;
        }
      } else {
        store_row<G, J>(S.ptail, (size_t)g, S.ld, lane, acc);
        if (lane == 0) S.pts[g] = accs;
      }
    };
    // Lists of LONG segments: a round of LB touches nearly always continues the running segment.  The window's keys / samples are staged
    // in LDS (group-private: 2 CH words each) and a round reads its LB of each with four 16-byte broadcast reads — the register form
    // below spends 4 selects + a shuffle per key and per sample — then, when the round's first and last key ARE the running key (sorted:
    // so is everything between), its LB rows go out together and are added without a compare.  Any other round walks its touches one
    // by one (boundaries of the short W2T / V segments, blanked touches, the first round of a window).
    static_assert(LB == 8 || LB == 4 || LB == 2, "a round is one or two 16-byte reads of each kind");
    uint32_t *const lk = reinterpret_cast<uint32_t *>(seg_lds + ((CPB * (S.ld + 1) + 3) & ~3)) + (size_t)r * 4 * CH;
    uint32_t *const lv = lk + 2 * CH;
#pragma unroll
    for (int q = 0; q < KPL; ++q) {
      if (q * G + lane < 2 * CH) { lk[q * G + lane] = kreg[q]; lv[q * G + lane] = vreg[q]; }
    }
    wave_lds_sync();
    for (int t0 = 0; t0 < n; t0 += LB) {
      uint32_t k8[LB], b8[LB];
      if constexpr (LB >= 4) {
#pragma unroll
        for (int u = 0; u < LB; u += 4) {
          const uint4 kk = *reinterpret_cast<const uint4 *>(lk + t0 + u), bb = *reinterpret_cast<const uint4 *>(lv + t0 + u);
          k8[u] = kk.x; k8[u + 1] = kk.y; k8[u + 2] = kk.z; k8[u + 3] = kk.w;
          b8[u] = bb.x; b8[u + 1] = bb.y; b8[u + 2] = bb.z; b8[u + 3] = bb.w;
        }
      } else {
        k8[0] = lk[t0]; k8[1] = lk[t0 + 1]; b8[0] = lv[t0]; b8[1] = lv[t0 + 1];
      }
      if (cur != DRX_KEY_NONE && k8[0] == cur && k8[LB - 1] == cur) {
        float s8[LB], c8[LB];
        float4 rows[LB][J];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          s8[u] = 0.f; c8[u] = 1.f;
          pol.template load<G, J>(cur, b8[u], lane, rows[u], s8[u], c8[u]);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
#pragma unroll
          for (int jx = 0; jx < J; ++jx) f4_fma(acc[jx], c8[u], rows[u][jx]);
          accs += s8[u];
        }
        cur_pos = start + t0 + LB - 1;
      } else {
        for (int u = 0; u < LB; ++u) {
          const uint32_t key = lk[t0 + u];
          if (key == DRX_KEY_NONE) continue;
          float sv = 0.f, cv = 1.f;
          float4 row[J];
          pol.template load<G, J>(key, lv[t0 + u], lane, row, sv, cv);
          if (key != cur) {
            flush(false);
            cur = key;
            cur_from_start = (t0 + u == 0);
#pragma unroll
            for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
            accs = 0.f;
          }
#pragma unroll
          for (int jx = 0; jx < J; ++jx) f4_fma(acc[jx], cv, row[jx]);
          accs += sv;
          cur_pos = start + t0 + u;
        }
      }
    }
    flush(n > 0 && bcast(kreg, n - 1) != DRX_KEY_NONE);
  } else {
    seg_fold_short<G, J, CH, LB1>(S, pol, g, lane, start, n, prev_key, next_key, all_inner, acc, accs);
  }
  if (all_inner) {                 // every chunk of this workgroup is one whole run of the same segment: one partial for all of them
    float *sc = seg_lds + (size_t)CPB * S.ld;
    store_row<G, J>(seg_lds, (size_t)r, S.ld, lane, acc);
    if (lane == 0) sc[r] = accs;
    __syncthreads();
    if (r == 0) {
      float4 t[J];
#pragma unroll
      for (int jx = 0; jx < J; ++jx) t[jx] = f4_zero();
      float ts = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < CPB; ++rr) {
        float4 v[J];
        load_row<G, J>(seg_lds, (size_t)rr, S.ld, lane, v);
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_add(t[jx], v[jx]);
        ts += sc[rr];
      }
      store_row<G, J>(PB.pblock, (size_t)blk, S.ld, lane, t);
      if (lane == 0) PB.pbs[blk] = ts;
    }
  }
}

// partial i of a span (after the tail partial of its first chunk)
template <int G, int J>
__device__ __forceinline__ void span_partial(const SegBufs &S, const PlanBufs &PB, const SpanShape &sh, int cpb, int i, int lane,
                                             float4 (&v)[J], float &sv) {
  if (i < sh.n_lead) {
    const int c = sh.first + i;
    load_row<G, J>(S.phead, (size_t)c, S.ld, lane, v); sv = S.phs[c];
  } else if (i < sh.n_lead + sh.n_blk) {
    const int bb = (sh.first + sh.n_lead) / cpb + (i - sh.n_lead);
    load_row<G, J>(PB.pblock, (size_t)bb, S.ld, lane, v); sv = PB.pbs[bb];
  } else {
    const int c = sh.first + sh.n_lead + sh.n_blk * cpb + (i - sh.n_lead - sh.n_blk);
    load_row<G, J>(S.phead, (size_t)c, S.ld, lane, v); sv = S.phs[c];
  }
}

// One launch, kFixBlock threads: workgroups [0, n_long_blocks) take the long spans (one workgroup per span, its groups stride over the
// partials, group sums combined in LDS in group order), workgroups [n_long_blocks, n_long_blocks + n_short_blocks) the short ones (one
// group per span), the rest run `extra` (the CDAE step: one workgroup finishing the hidden bias).  LDS: [R, ld] + [R] floats.
template <int G, int J, class Policy, class Extra>
__global__ __launch_bounds__(kFixBlock) void k_span_planned(SegBufs S, PlanBufs PB, SpanPlan SP, Policy pol, int n_long_blocks,
                                                            int n_short_blocks, Extra extra) {
  extern __shared__ __align__(16) float span_lds[];
  constexpr int R = kFixBlock / G, CPB = kSegBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  constexpr int UL = J == 1 ? 8 : 2;                       // partial rows in flight per group
  if ((int)blockIdx.x >= n_long_blocks + n_short_blocks) { extra(span_lds); return; }
  if ((int)blockIdx.x >= n_long_blocks) {
    const uint32_t n_short = SP.cnt[0];
    for (uint32_t si = ((int)blockIdx.x - n_long_blocks) * R + r; si < n_short; si += (uint32_t)n_short_blocks * R) {
      const uint2 d = SP.desc[si];
      const SpanShape sh(d, CPB);
      const int g0 = (int)d.x, tot = sh.total();
      const int kpos = min(S.T, (g0 + 1) * SP.chunk) - 1;
      float4 t[J];
      load_row<G, J>(S.ptail, (size_t)g0, S.ld, lane, t);
      float ts = S.pts[g0];
      for (int i0 = 0; i0 < tot; i0 += UL) {
        float4 v[UL][J];
        float sv[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          sv[u] = 0.f;
#pragma unroll
          for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
          if (i0 + u < tot) span_partial<G, J>(S, PB, sh, CPB, i0 + u, lane, v[u], sv[u]);
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
#pragma unroll
          for (int j = 0; j < J; ++j) f4_add(t[j], v[u][j]);
          ts += sv[u];
        }
      }
      pol.template finish<G, J>(S.keys_s[kpos], kpos, lane, t, ts);
    }
    return;
  }
  float *sc = span_lds + (size_t)R * S.ld;
  const uint32_t n_long = SP.cnt[1];
  for (uint32_t si = blockIdx.x; si < n_long; si += (uint32_t)n_long_blocks) {
    const uint2 d = SP.desc[S.n_chunks - 1 - (int)si];
    const SpanShape sh(d, CPB);
    const int g0 = (int)d.x, tot = sh.total();
    const int kpos = min(S.T, (g0 + 1) * SP.chunk) - 1;
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
    float accs = 0.f;
    // group r takes the contiguous slice [r * per, (r + 1) * per) of the partials, so that the combine below is in partial order
    const int per = (tot + R - 1) / R;
    const int i_lo = r * per, i_hi = min(tot, i_lo + per);
    for (int i0 = i_lo; i0 < i_hi; i0 += UL) {
      float4 v[UL][J];
      float sv[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        sv[u] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
        if (i0 + u < i_hi) span_partial<G, J>(S, PB, sh, CPB, i0 + u, lane, v[u], sv[u]);
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(acc[j], v[u][j]);
        accs += sv[u];
      }
    }
    __syncthreads();                                         // (the previous span's readers are done with the LDS rows)
    store_row<G, J>(span_lds, (size_t)r, S.ld, lane, acc);
    if (lane == 0) sc[r] = accs;
    __syncthreads();
    if (r == 0) {
      float4 t[J];
      load_row<G, J>(S.ptail, (size_t)g0, S.ld, lane, t);
      float ts = S.pts[g0];
#pragma unroll 8
      for (int rr = 0; rr < R; ++rr) {
        float4 v[J];
        load_row<G, J>(span_lds, (size_t)rr, S.ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
        ts += sc[rr];
      }
      pol.template finish<G, J>(S.keys_s[kpos], kpos, lane, t, ts);
    }
  }
}

// Fix-up of chunk-crossing segments, two tiers.
//   k_span_short : one GROUP per crossing segment: tail partial of its first chunk + head partials of the next chunks
//                  whose first key equals the segment key, in chunk order.  Segments that cross more than
//                  kShortSpan chunk borders (hot items) are queued for
//   k_span_long  : one 1024-thread workgroup per such segment; its R = 1024/G groups stride over the chunks and the
//                  R partial sums are combined in a fixed order.  Both tiers are deterministic.

template <int G, int J, class Policy>
__device__ __forceinline__ void span_short_body(const SegBufs &S, const Policy &pol, int block_id, int n_blocks) {
  const int lane = threadIdx.x % G;
  const int gshift = (threadIdx.x & 63) / G * G;          // position of this group's lanes in the wave's ballot
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const uint32_t n_span = S.n_span[0];
  const int gpb = kBlock / G;
  constexpr int UL = J == 1 ? 8 : 2;                       // partial rows in flight
  for (uint32_t si = block_id * gpb + threadIdx.x / G; si < n_span; si += n_blocks * gpb) {
    const int g0 = (int)S.span_list[si];
    const uint32_t key = S.keys_s[min(S.T, (g0 + 1) * kChunk) - 1];
    // m = number of following chunks that continue this segment: a run of "middle" chunks (flag 2), closed by an
    // optional "end" chunk (flag 1); the lanes probe G one-byte chunk flags at a time
    int m = 0;
    bool is_long = false;
    for (int base = g0 + 1;; base += G) {
      const int c = base + lane;
      const int fl = c < S.n_chunks ? (int)S.cflag[c] : 0;
      const unsigned long long mid = (__ballot(fl == 2) >> gshift) & gmask;
      const int run = mid == gmask ? G : __builtin_ctzll(~mid);       // leading run of middle chunks
      m += run;
      if (run < G) {
        m += (__shfl(fl, run, G) == 1) ? 1 : 0;
        break;
      }
      if (m >= kShortSpan) { is_long = true; break; }
    }
    if (is_long || m > kShortSpan) {
      if (lane == 0) S.long_list[atomicAdd(&S.n_span[1], 1u)] = (uint32_t)g0;
      continue;
    }
    float4 t[J];
    load_row<G, J>(S.ptail, (size_t)g0, S.ld, lane, t);
    float ts = S.pts[g0];
    for (int c0 = g0 + 1; c0 <= g0 + m; c0 += UL) {
      float4 v[UL][J];
      float sv[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        sv[u] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
        if (c0 + u <= g0 + m) { load_row<G, J>(S.phead, (size_t)(c0 + u), S.ld, lane, v[u]); sv[u] = S.phs[c0 + u]; }
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {           // chunk order
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(t[j], v[u][j]);
        ts += sv[u];
      }
    }
    pol.template finish<G, J>(key, min(S.T, (g0 + 1) * kChunk) - 1, lane, t, ts);
  }
}

template <int G, int J, class Policy>
__global__ __launch_bounds__(kBlock) void k_span_short(SegBufs S, Policy pol) {
  span_short_body<G, J, Policy>(S, pol, (int)blockIdx.x, (int)gridDim.x);
}

// lds: [R, ld] + [R] floats, R = kFixBlock / G
template <int G, int J, class Policy>
__device__ __forceinline__ void span_long_body(const SegBufs &S, const Policy &pol, int block_id, int n_blocks, float *lds) {
  constexpr int R = kFixBlock / G;
  float *sc = lds + (size_t)R * S.ld;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const uint32_t n_long = S.n_span[1];
  for (uint32_t si = block_id; si < n_long; si += n_blocks) {
    const int g0 = (int)S.long_list[si];
    const uint32_t key = S.keys_s[min(S.T, (g0 + 1) * kChunk) - 1];
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
    float accs = 0.f;
    // span length: all threads probe one-byte chunk flags, kFixBlock at a time (a run of "middle" chunks, closed by
    // an optional "end" chunk); positions beyond the last chunk read as 0, so the loop always terminates
    __shared__ int s_stop, s_len;
    int len = 0;
    for (int base = g0 + 1;; base += kFixBlock) {
      if (threadIdx.x == 0) s_stop = kFixBlock;
      __syncthreads();
      const int c = base + (int)threadIdx.x;
      const int fl = c < S.n_chunks ? (int)S.cflag[c] : 0;
      if (fl != 2) atomicMin(&s_stop, (int)threadIdx.x);
      __syncthreads();
      const int stop = s_stop;
      if (stop < kFixBlock) {
        if ((int)threadIdx.x == stop) s_len = len + stop + (fl == 1 ? 1 : 0);
        __syncthreads();
        break;
      }
      len += kFixBlock;
      __syncthreads();
    }
    const int m = s_len;
    constexpr int UL = J == 1 ? 8 : 2;        // partial rows in flight per group
    for (int c = g0 + 1 + r; c <= g0 + m; c += UL * R) {
      float4 v[UL][J];
      float sv[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int cu = c + u * R;
        sv[u] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
        if (cu <= g0 + m) { load_row<G, J>(S.phead, (size_t)cu, S.ld, lane, v[u]); sv[u] = S.phs[cu]; }
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(acc[j], v[u][j]);
        accs += sv[u];
      }
    }
    __syncthreads();
    store_row<G, J>(lds, (size_t)r, S.ld, lane, acc);
    if (lane == 0) sc[r] = accs;
    __syncthreads();
    if (r == 0) {
      float4 t[J];
      load_row<G, J>(S.ptail, (size_t)g0, S.ld, lane, t);
      float ts = S.pts[g0];
#pragma unroll 8
      for (int rr = 0; rr < R; ++rr) {
        float4 v[J];
        load_row<G, J>(lds, (size_t)rr, S.ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
        ts += sc[rr];
      }
      pol.template finish<G, J>(key, min(S.T, (g0 + 1) * kChunk) - 1, lane, t, ts);
    }
  }
}

template <int G, int J, class Policy>
__global__ __launch_bounds__(kFixBlock) void k_span_long(SegBufs S, Policy pol) {
  extern __shared__ __align__(16) float lds[];   // [R, ld] + [R]
  span_long_body<G, J, Policy>(S, pol, (int)blockIdx.x, (int)gridDim.x, lds);
}

}  // namespace drx
