// STREAMED form of the planned segmented reduction (drx_segreduce.hpp) for lists of SHORT segments whose rows are exactly one, two or
// four wave-wide pieces (ld = 256 / 128 / 64 floats) — the headline shape of the CDAE sparse step (K = 128).
//
// Why.  The planned kernel keeps every row it is waiting for in registers: 2 contribution rows per group in flight, and each finished
// segment pays a dependent read of its parameter + slot row before the next round may start (40 KB in flight per CU: by Little's law
// the 4.3 TB/s that kernel fetches at).  Here a wave turns its chunk window into ONE ordered stream of 16-byte-aligned ROW FETCHES —
// the contribution row of every touch, and behind the last touch of every segment that ends inside the window the segment's parameter
// row and its slot row — and moves that stream through a ring in LDS by LDS-DMA (global_load_lds_dwordx4: per-lane source address, no
// destination registers), D instructions (D KiB) ahead of the fold.  Nothing the fold needs is ever requested when it is needed.
//
//   * one chunk per WAVE: the 64 lanes hold one row (ld / 64 floats each); one LDS-DMA instruction brings 1 KiB = 256 / ld rows, all
//     64 lanes busy whatever the window looks like; every branch of the fold is wave-uniform (a scalar branch on the item's code);
//   * the stream's items live in LDS as 4-byte words: the row's index in its array << 4 | its kind;
//   * a wave loads all it must know about a chunk — 64 list positions, two ext bytes, the block's border keys — as independent loads, ONE
//     round instead of the planned kernel's three dependent ones (ext -> window -> keys).  One block of the plan per workgroup, dealt out
//     by the hardware as slots free up: a persistent launch striding over the blocks (6 per workgroup at the headline shape, 4 or 5 of
//     them real work: a static split is unbalanced) measured 30 % slower, two or four consecutive blocks per workgroup with the next
//     block's round prefetched no faster;
//   * vmcnt is counted by hand (the DMA is inline asm: hipcc neither counts it nor drains it): `s_waitcnt vmcnt(D - 1)` after issuing
//     instruction j + D - 1 guarantees instruction j has landed — loads, stores and LDS-DMA retire in issue order, so the stores of a
//     finished segment issued in between only make the wait stricter, never weaker;
//   * sums keep the planned kernel's order (a segment's touches in list order, fma(coef, row, acc) from zero; block partials of
//     all-inner workgroups in chunk order): the result is bit-identical to k_seg_reduce_planned's.
// Every key of the CDAE step's list streams: the key space is three row arrays side by side (W, W2T, V: SURVEY App. A) and an item's
// kind says which; touches of W2T rows also carry a scalar (the output bias's gradient).  Scalars do not ride in the ring: every lane
// loads its touch's scalar, and the bias + slot of the row its touch finishes, BEFORE the stream starts — a compiler-counted load
// waited for inside the loop would drain the ring at every use (hipcc waits vmcnt(0) for it: a window of 32 output-row touches then
// costs 32 full round trips and ends the launch 100 us late).  Blanked touches (DRX_KEY_NONE) are not items.
//
// The POLICY describes the arrays (DirectPolicyT, drx_cdae.hip):
//   StreamArrays stream_arrays()      per array v = 0, 1, 2: first key, contribution rows (touch (key, val): grad[v] + val * ld) and their
//                                     coefficient, the table and its one slot array (an element-wise optimizer with one slot)
//   float stream_decay()              weight of a row's own value in its gradient (reg / B)
//   void stream_update(g, p, a)       one element: gradient g (decay applied), parameter p, slot a
//   void stream_update_scalar(g, p, a)   the same for an array's scalar side (the output biases: no decay)
//   static constexpr bool kStreamParks   true: finished rows of some arrays are not applied to a table but PARKED — the sum stored at a
//                                     place the policy computes (the row-sharded step's gradient exchange buffer, drx_shard.hip):
//   bool stream_parked(v)             ... rows of array v
//   void stream_park_at(v, row, roff, soff)   float offsets of the row's sum / of its scalar sum in stream_park_base() (may load: runs
//                                     before the stream starts, on the lane whose touch finishes the row)
#pragma once
#include "drx_segreduce.hpp"

namespace drx {

// one LDS-DMA instruction: 16 bytes per lane from gsrc (per lane) to LDS bytes [lds_dst, lds_dst + 1024) in lane order (lds_dst
// wave-uniform).  M0 carries the destination and belongs to the compiler: saved and restored around the instruction.
// The ring slot it overwrites was read (ds_read) by the SAME wave an iteration earlier, and the values read were used (an fma: hipcc
// waits lgkmcnt for them in front of it) before this statement in program order — no wait is needed here.
__device__ __forceinline__ void lds_dma16(const void *gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// (native vectors: assignable through address-space-qualified pointers, which HIP's float2 / float4 structs are not)
template <int VL> struct LaneVec;
template <> struct LaneVec<1> { using T = float; };
template <> struct LaneVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct LaneVec<4> { typedef float T __attribute__((ext_vector_type(4))); };

// ring depth of the streamed reduction in LDS-DMA instructions (KiB) per wave; 0: the planned kernel everywhere (a variant is a build:
// scripts/build_variant.sh <name> "-DDRX_STREAM_DEPTH=..")
#ifndef DRX_STREAM_DEPTH
#define DRX_STREAM_DEPTH 4
#endif
constexpr int kStreamDepth = DRX_STREAM_DEPTH;
// The streamed kernels take their arguments (five structs of pointers) in ~106 scalar registers; a CU admits 256-thread workgroups up to
// 800 / (sgprs rounded up to 16, + 16): 6 at 106, 8 at 80 (MI355X_MICROARCH: Residency).  Capped, the compiler parks the surplus in
// lanes of a vector register (v_writelane / v_readlane): 8 workgroups = all 32 wave slots of a CU.
#ifndef DRX_STREAM_SGPRS
#define DRX_STREAM_SGPRS 80
#endif
#define DRX_STREAM_SGPR_CAP __attribute__((amdgpu_num_sgpr(DRX_STREAM_SGPRS)))
constexpr int kStreamItems = 3 * (2 * kChunk - 1) + 3;        // a window's items at most (every touch a segment of its own), rounded up to a multiple of 4
constexpr int kStreamIndexBits = 28;                          // an item word: row index << 4 | kind
static inline size_t seg_stream_lds_bytes(int ld, int depth) {
  const int cpb = kSegBlock / (ld / 4);
  return (size_t)cpb * ((size_t)depth * 1024 + (size_t)kStreamItems * 4 + 128) + (size_t)cpb * (ld + 4) * 4;
}
struct StreamArrays {
  uint32_t first_key[3];       // keys [first_key[v], first_key[v + 1]) are rows of array v (ascending; the last array is open-ended)
  const float *grad[3];
  float *table[3], *slot[3];
  float coef[3];
  // scalar side of an array (nullptr: none): per touch (key, val) the value sgrad[v][val]; per finished row one parameter + slot scalar
  const float *sgrad[3];
  float *sparam[3], *sslot[3];
};

// item kinds (low 4 bits of an item word): 5 * array + ...
enum : uint32_t { kItGrad = 0, kItGradHead = 1, kItGradTail = 2, kItParam = 3, kItSlot = 4 };

template <int LD, int D, class Policy, class Extra>
__global__ __launch_bounds__((kSegBlock / (LD / 4)) * 64) DRX_STREAM_SGPR_CAP void k_seg_reduce_stream(SegBufs S, PlanBufs PB, SpanPlan SP, Policy pol,
                                                                               int extra_blocks, Extra extra) {
  extern __shared__ __align__(16) float seg_lds[];
  constexpr int G = LD / 4;                  // the row group the list's plan was made for (pick_geom(LD): J == 1)
  constexpr int CPB = kSegBlock / G;         // chunks per workgroup: the plan's block size (SpanShape, PlanBufs::pblock)
  constexpr int VL = LD / 64;                // floats of a row per lane
  constexpr int IPI = 256 / LD;              // rows per LDS-DMA instruction
  constexpr int LPI = 64 / IPI;              // lanes per row in that instruction (16 bytes each)
  constexpr int CH = kChunk;
  using V = typename LaneVec<VL>::T;
  using GV = V __attribute__((address_space(1)));
  static_assert(LD == 64 || LD == 128 || LD == 256, "one, two or four rows per KiB");
  static_assert(D >= 2 && D <= 32, "ring depth");
  static_assert(2 * CH <= 64 && 3 * CPB <= 64, "a window is one touch per lane; a block's border keys one per lane");
  if ((int)blockIdx.x < extra_blocks) { extra(seg_lds); return; }
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t *__restrict__ const ks = S.keys_s, *__restrict__ const vs = S.vals_s;
  const uint8_t *__restrict__ const ext = SP.ext;
  // LDS: the waves' rings, their item tables, their copies of the arrays' offsets, the rows of an all-inner workgroup's chunk sums
  char *const lds_b = reinterpret_cast<char *>(seg_lds);
  float *const ring = reinterpret_cast<float *>(lds_b + (size_t)wv * D * 1024);
  uint32_t *const tab = reinterpret_cast<uint32_t *>(lds_b + (size_t)CPB * D * 1024 + (size_t)wv * kStreamItems * 4);
  long long *const offs = reinterpret_cast<long long *>(lds_b + (size_t)CPB * (D * 1024 + kStreamItems * 4) + (size_t)wv * 128);
  float *const comb = reinterpret_cast<float *>(lds_b + (size_t)CPB * (D * 1024 + kStreamItems * 4 + 128));      // [CPB, LD] then [CPB]
  const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)ring);
  const StreamArrays A = pol.stream_arrays();
  const char *const origin = reinterpret_cast<const char *>(A.grad[0]);
  // an item's row array as a byte OFFSET from `origin`, by kind (a select between pointers is lowered to a table in scratch memory)
  if (lane < 15) {
    const int v = lane / 5, k = lane % 5;
    const uintptr_t g = (uintptr_t)(v == 0 ? A.grad[0] : (v == 1 ? A.grad[1] : A.grad[2]));
    const uintptr_t t = (uintptr_t)(v == 0 ? A.table[0] : (v == 1 ? A.table[1] : A.table[2]));      // (nullptr: a parked array — never an item)
    const uintptr_t sl = (uintptr_t)(v == 0 ? A.slot[0] : (v == 1 ? A.slot[1] : A.slot[2]));
    offs[lane] = (long long)((k <= (int)kItGradTail ? g : (k == (int)kItParam ? t : sl)) - (uintptr_t)origin);
  }
  const float decay = pol.stream_decay();
  const int sub = lane / LPI, piece = lane % LPI;     // this lane's row of an LDS-DMA instruction, its 16 bytes of that row

  // Everything a wave must know about a chunk before it can stream it, in ONE round of independent loads (the planned kernel walks
  // ext -> window -> keys: three dependent rounds, 2 - 3 us each beside the streams of the other waves): the 64 list positions behind
  // the chunk's start (its window lies in them: it begins ext[g - 1] <= 31 positions in and ends before position 32 + ext[g] <= 63),
  // the key in front of them, the two ext bytes, and — lanes 0 .. 3 CPB - 1 — the first / last / preceding key of every chunk of the
  // block (is the block all-inner?).
  struct Meta { uint32_t key, val, bk, prevk; int e0, e1; } m;
  // (the block of this workgroup: by the list's XCD placement — blocks inside a hot row go to the XCD whose band of the gradient rows
  // their samples lie in, drx_segreduce.hpp place_block — or in list order)
#if DRX_STREAM_PLACED == 2          // (diagnostic: the placement's lookups paid for, the blocks in list order all the same)
  const int pb_ = placed_block(SP, (int)blockIdx.x, extra_blocks, (int)gridDim.x - extra_blocks, CPB);
  const int blk = pb_ < -5 ? pb_ : (int)blockIdx.x - extra_blocks;
#else
  const int blk = DRX_STREAM_PLACED ? placed_block(SP, (int)blockIdx.x, extra_blocks, (int)gridDim.x - extra_blocks, CPB)
                                    : (int)blockIdx.x - extra_blocks;
#endif
  if (blk < 0) return;
  {
    const int g = blk * CPB + wv, base = g * CH, pos = base + lane;
    const bool have = g < S.n_chunks;
    m.key = have && pos < S.T ? ks[pos] : DRX_KEY_NONE;
    m.val = have && pos < S.T ? vs[pos] : 0u;
    m.prevk = have && g > 0 ? ks[base - 1] : DRX_KEY_NONE;
    m.e0 = have && g > 0 ? (int)ext[g - 1] : 0;
    m.e1 = have ? (int)ext[g] : 0;
    const int c = blk * CPB + lane % CPB, role = lane / CPB;
    const int p = role == 0 ? c * CH : (role == 1 ? (c + 1) * CH - 1 : c * CH - 1);
    m.bk = lane < 3 * CPB && c < S.n_chunks && (c + 1) * CH <= S.T && p >= 0 ? ks[p] : DRX_KEY_NONE;
  }
  wave_lds_sync();                                      // (offs)
  {
    // a chunk is INNER when it is one whole run of a segment that began before it; a block of inner chunks leaves one partial
    const uint32_t bl = (uint32_t)__shfl((int)m.bk, lane + CPB, 64), bp = (uint32_t)__shfl((int)m.bk, lane + 2 * CPB, 64);
    const bool inner = lane < CPB && m.bk != DRX_KEY_NONE && m.bk == bp && bl == m.bk;
    const bool all_inner = (__ballot(inner) & ((1ull << CPB) - 1ull)) == ((1ull << CPB) - 1ull);
    const int g = blk * CPB + wv;
    V acc;                                                 // this lane's floats of the running segment's sum
    float accs = 0.f;                                      // ... and its scalar side-sum
    float *const accf = reinterpret_cast<float *>(&acc);
#pragma unroll
    for (int v = 0; v < VL; ++v) accf[v] = 0.f;
    if (g < S.n_chunks) {
      // the chunk's window: behind the touches its left neighbour finishes for it, and into the right neighbour's for the segment it
      // finishes itself (SpanPlan::ext)
      const int base = g * CH;
      const int e0 = __builtin_amdgcn_readfirstlane(m.e0), e1 = __builtin_amdgcn_readfirstlane(m.e1);
      const int end_off = min(CH + e1, S.T - base);                // <= 63
      const int n = max(0, end_off - e0);
      const uint32_t key = m.key, val = m.val;
      const bool valid = lane >= e0 && lane < end_off && key != DRX_KEY_NONE;      // (blanked and dropped touches are no items)
      const uint32_t prev_key = e0 > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)key, e0 - 1) : (uint32_t)__builtin_amdgcn_readfirstlane((int)m.prevk);
      const uint32_t next_key = n > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)key, end_off) : DRX_KEY_NONE;      // (a position behind T reads as DRX_KEY_NONE)
      // ---- the window's items -------------------------------------------------------------------------------------------------
      const uint32_t key_right = (uint32_t)__shfl_down((int)key, 1, 64);
      const uint32_t key0 = (uint32_t)__builtin_amdgcn_readlane((int)key, n > 0 ? e0 : 0);
      const bool seg_end = valid && (lane == end_off - 1 || key_right != key);
      const bool cont_left = key == key0 && prev_key == key;            // the window's first run began before it
      const bool cont_right = lane == end_off - 1 && next_key == key;   // its last run goes on behind it
      const bool apply = seg_end && !cont_left && !cont_right;
      const uint32_t av = key < A.first_key[1] ? 0u : (key < A.first_key[2] ? 1u : 2u);        // the touch's array, its row there
      const uint32_t row = key - (av == 0 ? A.first_key[0] : (av == 1 ? A.first_key[1] : A.first_key[2]));
      bool parked = false;                                              // the finished row's sum is stored, not applied
      if constexpr (Policy::kStreamParks) parked = apply && pol.stream_parked(av);
      const int w = valid ? (apply && !parked ? 3 : 1) : 0;
      int incl = w;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      const int pos = incl - w;
      const int M = __builtin_amdgcn_readlane(incl, 63);
      float sc = 0.f, sp = 0.f, ss = 0.f;                  // this touch's scalar; bias + slot of the row it finishes
      uint32_t park_row = 0xFFFFFFFFu, park_scal = 0u;     // where the sum of the row this touch finishes is parked (float offsets)
      if (valid) {
        const uint32_t kind = 5u * av + (!seg_end || apply ? kItGrad : (cont_left ? kItGradHead : kItGradTail));
        tab[pos] = (val << 4) | kind;
        if (apply && !parked) { tab[pos + 1] = (row << 4) | (5u * av + kItParam); tab[pos + 2] = (row << 4) | (5u * av + kItSlot); }
        const float *const sg = av == 0 ? A.sgrad[0] : (av == 1 ? A.sgrad[1] : A.sgrad[2]);
        if (sg) {
          sc = sg[val];
          if (apply && !parked) {
            sp = (av == 0 ? A.sparam[0] : (av == 1 ? A.sparam[1] : A.sparam[2]))[row];
            ss = (av == 0 ? A.sslot[0] : (av == 1 ? A.sslot[1] : A.sslot[2]))[row];
          }
        }
        if constexpr (Policy::kStreamParks) {
          if (parked) pol.stream_park_at(av, row, park_row, park_scal);
        }
      }
      // (the scalars and places have LANDED before the first LDS-DMA goes out: used here, never waited for inside the loop)
      asm volatile("" : "+v"(sc), "+v"(sp), "+v"(ss), "+v"(park_row), "+v"(park_scal));
      unsigned long long todo = __ballot(valid);            // the window's touches not yet folded (bit = lane)
      int at_lane = 0;                                      // lane of the touch folded last
      wave_lds_sync();
      // ---- the stream ---------------------------------------------------------------------------------------------------------
      const int ns = (M + IPI - 1) / IPI;                 // LDS-DMA instructions
      auto issue = [&](int j, int slot) __attribute__((always_inline)) {
        const int i = j * IPI + sub;
        const uint32_t it = tab[i < kStreamItems ? i : kStreamItems - 1];
        const long long off = offs[it & 15u] + (long long)(it >> 4) * (LD * 4);
        const char *const src = origin + (i < M ? off : 0ll);                       // (lanes behind the stream's end re-read a hot line)
        lds_dma16(src + piece * 16, ring_lds + (uint32_t)slot * 1024u);
      };
      V prow;                                             // the parameter row of the segment being finished
      float *const pf = reinterpret_cast<float *>(&prow);
#pragma unroll
      for (int v = 0; v < VL; ++v) pf[v] = 0.f;
      int issued = 0;
      for (; issued < D - 1 && issued < ns; ++issued) issue(issued, issued);
      bool drained = false;
      int slot = 0, islot = issued % D;
      for (int j = 0; j < ns; ++j) {
        if (issued < ns) {
          issue(issued, islot);
          ++issued;
          islot = islot + 1 == D ? 0 : islot + 1;
          wait_vmcnt<D - 1>();
        } else if (!drained) {
          wait_vmcnt<0>();
          drained = true;
        }
        const float *const rows = ring + (size_t)slot * 256;
        slot = slot + 1 == D ? 0 : slot + 1;
#pragma unroll
        for (int s = 0; s < IPI; ++s) {
          const int i = j * IPI + s;
          if (i < M) {
            const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[i]);
            const uint32_t av = (it & 15u) / 5u, kind = (it & 15u) % 5u, idx = it >> 4;
            const V row = *reinterpret_cast<const V *>(rows + s * LD + lane * VL);
            const float *const rf = reinterpret_cast<const float *>(&row);
            if (kind <= kItGradTail) {
              const float coef = av == 0 ? A.coef[0] : (av == 1 ? A.coef[1] : A.coef[2]);
#pragma unroll
              for (int v = 0; v < VL; ++v) accf[v] = fmaf(coef, rf[v], accf[v]);
              at_lane = __builtin_ctzll(todo);
              todo &= todo - 1ull;
              accs += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), at_lane));
              if constexpr (Policy::kStreamParks) {
                const uint32_t ro = (uint32_t)__builtin_amdgcn_readlane((int)park_row, at_lane);
                if (ro != 0xFFFFFFFFu) {                      // this touch finished a row whose sum is parked
                  float *const pb = pol.stream_park_base();
                  ((GV *)(uintptr_t)(pb + ro))[lane] = acc;
                  if (av == 0 ? A.sgrad[0] != nullptr : (av == 1 ? A.sgrad[1] != nullptr : A.sgrad[2] != nullptr)) {
                    if (lane == 0) pb[(uint32_t)__builtin_amdgcn_readlane((int)park_scal, at_lane)] = accs;
                  }
#pragma unroll
                  for (int v = 0; v < VL; ++v) accf[v] = 0.f;
                  accs = 0.f;
                }
              }
              if (kind == kItGradHead) {
                if (!all_inner) {
                  *reinterpret_cast<V *>(S.phead + (size_t)g * LD + lane * VL) = acc;
                  if (lane == 0) S.phs[g] = accs;
#pragma unroll
                  for (int v = 0; v < VL; ++v) accf[v] = 0.f;
                  accs = 0.f;
                }
              } else if (kind == kItGradTail) {
                *reinterpret_cast<V *>(S.ptail + (size_t)g * LD + lane * VL) = acc;
                if (lane == 0) S.pts[g] = accs;
#pragma unroll
                for (int v = 0; v < VL; ++v) accf[v] = 0.f;
                accs = 0.f;
              }
            } else if (kind == kItParam) {
              prow = row;
            } else {
              V slotv = row;
              float *const sf = reinterpret_cast<float *>(&slotv);
#pragma unroll
              for (int v = 0; v < VL; ++v) pol.stream_update(fmaf(decay, pf[v], accf[v]), pf[v], sf[v]);
              // (addresses rebuilt from integers: said to be GLOBAL ones, or the stores become flat_store — counted in lgkmcnt too and
              // retired out of order)
              const long long roff = (long long)idx * (LD * 4);
              ((GV *)(uintptr_t)(origin + offs[5u * av + kItParam] + roff))[lane] = prow;
              ((GV *)(uintptr_t)(origin + offs[5u * av + kItSlot] + roff))[lane] = slotv;
              float *const spar = av == 0 ? A.sparam[0] : (av == 1 ? A.sparam[1] : A.sparam[2]);
              if (spar) {                                     // the row's scalar side: bias and slot came with the touch that finished it
                float pb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sp), at_lane));
                float ps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ss), at_lane));
                pol.stream_update_scalar(accs, pb, ps);
                if (lane == 0) { spar[idx] = pb; (av == 0 ? A.sslot[0] : (av == 1 ? A.sslot[1] : A.sslot[2]))[idx] = ps; }
              }
#pragma unroll
              for (int v = 0; v < VL; ++v) accf[v] = 0.f;
              accs = 0.f;
            }
          }
        }
      }
      if (all_inner) *reinterpret_cast<V *>(comb + (size_t)wv * LD + lane * VL) = acc;
    }
    if (all_inner) {               // every chunk of this block is one whole run of the same segment: one partial for all of them
      float *const sc = comb + (size_t)CPB * LD;
      if (lane == 0) sc[wv] = accs;
      __syncthreads();
      if (wv == 0) {
        V t;
        float *const tf = reinterpret_cast<float *>(&t);
#pragma unroll
        for (int v = 0; v < VL; ++v) tf[v] = 0.f;
        float ts = 0.f;
#pragma unroll
        for (int rr = 0; rr < CPB; ++rr) {
          const V x = *reinterpret_cast<const V *>(comb + (size_t)rr * LD + lane * VL);
          const float *const xf = reinterpret_cast<const float *>(&x);
#pragma unroll
          for (int v = 0; v < VL; ++v) tf[v] += xf[v];
          ts += sc[rr];
        }
        *reinterpret_cast<V *>(PB.pblock + (size_t)blk * LD + lane * VL) = t;
        if (lane == 0) PB.pbs[blk] = ts;
      }
    }
  }
}

}  // namespace drx
