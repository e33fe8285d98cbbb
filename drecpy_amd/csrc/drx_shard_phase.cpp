// The exchange phases of a row-sharded step, issued from C (include/drx.h "drx_shard_phase_*"; no reference equivalent: DRecPy is
// single-process, recommender_abc.py:16 is its only device line).  Host code only: every launch and every exchange goes through the
// library's own C ABI (drx_shard_*, drx_comm_*) — this file is the arithmetic of the unit-major exchange geometry and the order of the
// calls, which drecpy_amd/dist.py states in Python (and keeps for micro-batches and for torch.distributed as the transport).
#include <cstdint>
#include <cstring>
#include "drx.h"

namespace {

inline int64_t pad32(int64_t n) { return (n + 31) & ~int64_t(31); }

struct Geo {
  int world, rank, chunks, ld, bypass;
  // one chunk of one direction: per peer the float offset / float count of its piece inside the chunk's run, keys likewise
  struct Chunk {
    int64_t key_off[DRX_MAX_WORLD], keys[DRX_MAX_WORLD];          // relative to the chunk's first key
    int64_t flt_off[DRX_MAX_WORLD], flts[DRX_MAX_WORLD];          // relative to the chunk's first float; 0 floats for the rank itself (bypass)
    int64_t key0, flt0, n_keys, n_flts;                            // the chunk's place in the whole buffer
    int32_t counts32[DRX_MAX_WORLD];
  };
};

// counts[p * chunks + c] -> the layout of chunk c (pieces in peer order; chunks follow each other)
int chunk_layout(const Geo &g, const int64_t *counts, int c, Geo::Chunk &out) {
  int64_t key0 = 0, flt0 = 0;
  for (int cc = 0; cc <= c; ++cc) {
    int64_t k = 0, f = 0;
    for (int p = 0; p < g.world; ++p) {
      const int64_t n = counts[p * g.chunks + cc];
      if (n < 1 || n > 0x7FFFFFFF) return DRX_EINVAL;             // every unit holds at least its sentinel
      const int64_t fl = (g.bypass && p == g.rank) ? 0 : n * g.ld + pad32(n);
      if (cc == c) { out.key_off[p] = k; out.keys[p] = n; out.flt_off[p] = f; out.flts[p] = fl; out.counts32[p] = (int32_t)n; }
      k += n; f += fl;
    }
    if (cc == c) { out.key0 = key0; out.flt0 = flt0; out.n_keys = k; out.n_flts = f; }
    key0 += k; flt0 += f;
  }
  return DRX_OK;
}

// float offset of the rank's OWN piece of chunk c in a requester buffer (behind all the pieces that travel, in chunk order)
int64_t own_offset(const Geo &g, const int64_t *send_counts, int c) {
  int64_t f = 0;
  for (int cc = 0; cc < g.chunks; ++cc)
    for (int p = 0; p < g.world; ++p)
      if (p != g.rank) { const int64_t n = send_counts[p * g.chunks + cc]; f += n * g.ld + pad32(n); }
  for (int cc = 0; cc < c; ++cc) { const int64_t n = send_counts[g.rank * g.chunks + cc]; f += n * g.ld + pad32(n); }
  return f;
}

int geo_of(const DrxShard *sh, int ld, Geo &g) {
  if (!sh) return DRX_EINVAL;
  const int chunks = drx_shard_chunks(sh);
  if (chunks < 1 || chunks > DRX_MAX_CHUNKS || sh->world < 1 || sh->world > DRX_MAX_WORLD) return DRX_EINVAL;
  g.world = sh->world; g.rank = sh->rank; g.chunks = chunks; g.ld = ld;
  g.bypass = (sh->flags & DRX_SHARD_SELF_BYPASS) ? 1 : 0;
  return DRX_OK;
}

int check_x(const DrxShardExchange *x) {
  return (x && x->send_counts && x->recv_counts && x->uniq && x->req && x->table) ? DRX_OK : DRX_EINVAL;
}

// one all-to-all of a chunk: piece p of `send` (layout s) to peer p, piece p of `recv` (layout r) from it
int64_t exchange(DrxComm *comm, const Geo &g, const void *send, const Geo::Chunk &s, void *recv, const Geo::Chunk &r, bool keys,
                 void *after_stream) {
  int64_t so[DRX_MAX_WORLD], sb[DRX_MAX_WORLD], ro[DRX_MAX_WORLD], rb[DRX_MAX_WORLD];
  for (int p = 0; p < g.world; ++p) {
    so[p] = 4 * (keys ? s.key0 + s.key_off[p] : s.flt0 + s.flt_off[p]);
    sb[p] = 4 * (keys ? s.keys[p] : s.flts[p]);
    ro[p] = 4 * (keys ? r.key0 + r.key_off[p] : r.flt0 + r.flt_off[p]);
    rb[p] = 4 * (keys ? r.keys[p] : r.flts[p]);
  }
  return drx_comm_alltoallv(comm, send, so, sb, recv, ro, rb, after_stream);
}

}  // namespace

extern "C" {

int drx_shard_exchange_sizes(const DrxCdaeParams *p, const DrxShard *sh, const int64_t *send_counts, const int64_t *recv_counts,
                             int64_t *sizes4) {
  Geo g;
  if (!p || !send_counts || !recv_counts || !sizes4 || geo_of(sh, p->ld, g)) return DRX_EINVAL;
  int64_t req_f = 0, own_f = 0, keys_in = 0, keys_out = 0;
  for (int c = 0; c < g.chunks; ++c)
    for (int q = 0; q < g.world; ++q) {
      const int64_t ns = send_counts[q * g.chunks + c], nr = recv_counts[q * g.chunks + c];
      if (ns < 1 || nr < 1 || ns > 0x7FFFFFFF || nr > 0x7FFFFFFF) return DRX_EINVAL;
      req_f += ns * g.ld + pad32(ns);                               // (the rank's own units included: they sit at the buffer's end)
      if (!(g.bypass && q == g.rank)) own_f += nr * g.ld + pad32(nr);
      keys_in += nr; keys_out += ns;
    }
  sizes4[0] = req_f < 32 ? 32 : req_f;
  sizes4[1] = own_f < 32 ? 32 : own_f;
  sizes4[2] = keys_in < 32 ? 32 : keys_in;
  sizes4[3] = keys_out;
  return DRX_OK;
}

// The geometry the phases use, laid open for tests (no device call): for exchange chunk `chunk`, per peer p the BYTE offset and size of
//   out[0 .. 4 world)   the key exchange:      send offset, send bytes, recv offset, recv bytes   (uniq -> req)
//   out[4 .. 8 world)   the row exchange:      send (rows_send, the owner's geometry) and recv (rows_cache, the requester's)
//   out[8 .. 12 world)  the gradient exchange: send (grad_send, requester's) and recv (grad_recv, owner's)
// each block as four runs of `world` values; then out[12 world] = key offset of the chunk in req, out[12 world + 1] = float offset of the
// chunk in the owner's buffers, out[12 world + 2] = float offset of the rank's OWN piece of the chunk in grad_send (0 without the bypass).
int drx_shard_phase_layout(const DrxCdaeParams *p, const DrxShard *sh, const int64_t *send_counts, const int64_t *recv_counts, int32_t chunk,
                           int64_t *out) {
  Geo g;
  if (!p || !send_counts || !recv_counts || !out || geo_of(sh, p->ld, g) || chunk < 0 || chunk >= g.chunks) return DRX_EINVAL;
  Geo::Chunk s, r;
  int rc = chunk_layout(g, send_counts, chunk, s);
  if (!rc) rc = chunk_layout(g, recv_counts, chunk, r);
  if (rc) return rc;
  const int W = g.world;
  for (int q = 0; q < W; ++q) {
    out[0 * W + q] = 4 * (s.key0 + s.key_off[q]); out[1 * W + q] = 4 * s.keys[q];
    out[2 * W + q] = 4 * (r.key0 + r.key_off[q]); out[3 * W + q] = 4 * r.keys[q];
    out[4 * W + q] = 4 * (r.flt0 + r.flt_off[q]); out[5 * W + q] = 4 * r.flts[q];
    out[6 * W + q] = 4 * (s.flt0 + s.flt_off[q]); out[7 * W + q] = 4 * s.flts[q];
    out[8 * W + q] = 4 * (s.flt0 + s.flt_off[q]); out[9 * W + q] = 4 * s.flts[q];
    out[10 * W + q] = 4 * (r.flt0 + r.flt_off[q]); out[11 * W + q] = 4 * r.flts[q];
  }
  out[12 * W] = r.key0;
  out[12 * W + 1] = r.flt0;
  out[12 * W + 2] = g.bypass ? own_offset(g, send_counts, chunk) : 0;
  return DRX_OK;
}

int drx_shard_phase_keys(const DrxShard *sh, DrxComm *comm, DrxShardExchange *x, void *stream) {
  Geo g;
  if (!comm || check_x(x) || geo_of(sh, 4, g)) return DRX_EINVAL;
  for (int c = 0; c < g.chunks; ++c) {
    Geo::Chunk s, r;
    int rc = chunk_layout(g, x->send_counts, c, s);
    if (!rc) rc = chunk_layout(g, x->recv_counts, c, r);
    if (rc) return rc;
    const int64_t t = exchange(comm, g, x->uniq, s, x->req, r, true, stream);
    if (t < 0) return (int)t;
    rc = drx_comm_wait(comm, t, stream);
    if (!rc)
      rc = drx_shard_owner_index(sh, x->req + r.key0, (int32_t)r.n_keys, r.counts32, g.world, c, x->table, x->table_bytes, stream);
    if (rc) return rc;
  }
  return DRX_OK;
}

int drx_shard_phase_rows(const DrxCdaeParams *p, const DrxShard *sh, DrxComm *comm, DrxShardExchange *x, int32_t chunk, void *stream) {
  Geo g;
  if (!p || !comm || check_x(x) || !x->rows_cache || !x->rows_send || geo_of(sh, p->ld, g) || chunk < 0 || chunk >= g.chunks)
    return DRX_EINVAL;
  Geo::Chunk s, r;                                       // s: what this rank asks for (requester side), r: what it is asked for
  int rc = chunk_layout(g, x->send_counts, chunk, s);
  if (!rc) rc = chunk_layout(g, x->recv_counts, chunk, r);
  if (rc) return rc;
  rc = drx_shard_gather_rows(p, sh, x->req + r.key0, (int32_t)r.n_keys, r.counts32, g.world, x->rows_send + r.flt0, stream);
  if (rc) return rc;
  const int64_t t = exchange(comm, g, x->rows_send, r, x->rows_cache, s, false, stream);
  if (t < 0) return (int)t;
  x->rows_ticket[chunk] = t;
  return DRX_OK;
}

int drx_shard_phase_local(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt,
                          DrxComm *comm, DrxShardExchange *x, const void *prepared, size_t prepared_bytes, int32_t b_norm,
                          int32_t loss_kind, void *scratch, size_t scratch_bytes, void *const *events, void *stream) {
  Geo g;
  if (!p || !comm || check_x(x) || !x->rows_cache || !x->grad_send || geo_of(sh, p->ld, g)) return DRX_EINVAL;
  for (int c = 0; c < g.chunks; ++c) {
    const int rc = drx_comm_wait(comm, x->rows_ticket[c], stream);
    if (rc) return rc;
  }
  return drx_shard_step_local(p, opt, sh, hist, bt, prepared, prepared_bytes, x->rows_cache, x->grad_send, b_norm, loss_kind, scratch,
                              scratch_bytes, events, stream);
}

int drx_shard_phase_tail(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, DrxComm *comm, DrxShardExchange *x,
                         DrxShardExchange *next, int32_t b_norm, float *loss_out, void *stream) {
  Geo g;
  if (!p || !opt || !comm || check_x(x) || !x->grad_send || !x->grad_recv || geo_of(sh, p->ld, g)) return DRX_EINVAL;
  if (next && (check_x(next) || !next->rows_cache || !next->rows_send)) return DRX_EINVAL;
  Geo::Chunk s[DRX_MAX_CHUNKS], r[DRX_MAX_CHUNKS];
  for (int c = 0; c < g.chunks; ++c) {
    int rc = chunk_layout(g, x->send_counts, c, s[c]);
    if (!rc) rc = chunk_layout(g, x->recv_counts, c, r[c]);
    if (rc) return rc;
  }
  // the gradient rows leave chunk by chunk (the buffer has the row cache's geometry: chunk c's units at the same place) ...
  for (int c = 0; c < g.chunks; ++c) {
    const int64_t t = exchange(comm, g, x->grad_send, s[c], x->grad_recv, r[c], false, stream);
    if (t < 0) return (int)t;
    x->grad_ticket[c] = t;
  }
  // ... and the owner works through them in the same order: apply what has arrived, answer the next step's requests for that key range
  for (int c = 0; c < g.chunks; ++c) {
    int rc = drx_comm_wait(comm, x->grad_ticket[c], stream);
    if (rc) return rc;
    const float *own_grad = x->grad_send;
    const int64_t own_off = g.bypass ? own_offset(g, x->send_counts, c) : 0;
    rc = drx_shard_apply(p, opt, sh, b_norm, x->req + r[c].key0, x->grad_recv + r[c].flt0, (int32_t)r[c].n_keys, r[c].counts32, g.world, c,
                         x->table, g.bypass ? &own_grad : nullptr, g.bypass ? &own_off : nullptr, c == g.chunks - 1 ? loss_out : nullptr,
                         stream);
    if (rc) return rc;
    if (next) {
      rc = drx_shard_phase_rows(p, sh, comm, next, c, stream);
      if (rc) return rc;
    }
  }
  return DRX_OK;
}

}  // extern "C"
