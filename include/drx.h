/* drx.h — C ABI of libdrx.so, the MI355X (gfx950) engine behind the DRecPy-compatible Python surface.
 *
 * The reference (fabioiuri/DRecPy) is pure Python and has no FFI of its own: its device boundary is
 * "whatever TensorFlow eager op the hook calls" (DRecPy/Recommender/recommender_abc.py:191-205).  This
 * header is therefore the boundary a maintainer would bind with ctypes; each entry point names the
 * reference lines whose work it replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - plain C types only; every pointer whose name is not prefixed `h_` is DEVICE memory owned by the
 *     caller (torch-allocated in the Python host); the library never allocates or frees device memory,
 *     never synchronises the device, and keeps no global state.  Work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).
 *   - return value: 0 on success, <0 = DRX_E* below, >0 = a hipError_t.  Never throws.
 *   - parameter tables are fp32, row-major, row stride `ld` floats (ld % 4 == 0, ld >= k, the
 *     padding columns are zero and stay zero).
 *   - W2T is the reference's W_ [K,N] stored transposed [N,ld] so that an output unit is one row.
 */
#ifndef DRX_H_
#define DRX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRX_VERSION 100

enum {
  DRX_OK = 0,
  DRX_EINVAL = -1,     /* bad argument (null pointer, k > DRX_MAX_K, ld % 4, ...) */
  DRX_ESCRATCH = -2,   /* scratch buffer too small */
  DRX_ENOTIMPL = -3,
  DRX_ERETRY = -4,     /* a sampler gave up after its maximum number of consecutive failed attempts */
  DRX_ECOMM = -5       /* the RCCL transport failed (librccl missing, or an nccl* call: drx_comm_last_error() has the text) */
};

#define DRX_MAX_K 1024
#define DRX_KEY_NONE 0xFFFFFFFFu

enum { DRX_LOSS_BCE = 0, DRX_LOSS_MSE = 1 };           /* cdae.py:30-31 */
enum { DRX_TARGETS_REFERENCE = 0, DRX_TARGETS_PER_ROW = 1 };   /* (B,B,N) broadcast == batch-mean target, cdae.py:78-79 */
/* or-ed into targets_kind of drx_cdae_step_dense: the batch-membership arrays inside `scratch` are known to be zero (zero-initialised
 * scratch, or scratch last used by a COMPLETED dense step of the same batch size: consumers clear what they read) -> no memset */
#define DRX_DENSE_AUX_CLEAN 0x100
enum { DRX_OPT_ADAM = 0, DRX_OPT_ADAGRAD = 1,
       DRX_OPT_ROWWISE_ADAGRAD = 2   /* sampled mode only: ONE accumulator per table row, kept in the first float of the row's slot
                                        (s1[var][row * ld]): acc += mean_k(g^2), p -= lr * g / (sqrt(acc) + eps); b and b2 per element */ };

/* CDAE parameters (cdae.py:34-41) and optimizer slots of the same shapes.
 * Adam uses s1 = m, s2 = v; Adagrad uses s1 = accumulator (s2 unused, may be NULL). */
typedef struct DrxCdaeParams {
  int32_t n_users, n_items, k, ld;
  float *W;    /* [n_items, ld]  cdae.py:36 */
  float *W2T;  /* [n_items, ld]  cdae.py:37 (transposed) */
  float *V;    /* [n_users, ld]  cdae.py:38 */
  float *b;    /* [ld]           cdae.py:40 */
  float *b2;   /* [n_items]      cdae.py:41 */
} DrxCdaeParams;

/* The training set as the hot loop needs it: CSR of each user's POSITIVE items
 * (interaction >= threshold, duplicates merged, columns ascending) — the non-zeros of
 * `select_user_interaction_vec(uid)` after the binarisation of cdae.py:61. */
typedef struct DrxHistory {
  const int64_t *indptr;   /* [n_users + 1] */
  const int32_t *indices;  /* [indptr[n_users]] */
  /* Optional ITEM-MAJOR RANK of every entry (NULL / 0: none): t_rank[p] = the place of entry p = indptr[u] + j (item indices[p] in the
   * row of user u) when all entries are ordered by (item, user) — the permutation that transposes the matrix, inverted.  With it the
   * preparation of a sampled batch whose rows collect long runs of touches (MovieLens shapes: more than 8 touches per table row) lays
   * the batch's touches down item by item through this static order (a count per entry, a scan, a write entry by entry) instead of sorting millions
   * of (row, sample) pairs per step (csrc/drx_prep.hpp). */
  const int32_t *t_rank;   /* [t_nnz] */
  int64_t t_nnz;           /* = indptr[n_users] < 2^30 */
  /* ... and the entries IN that order (the transpose itself): entry e = item t_items[e] in the row of user t_users[e], at position
   * t_pos[e] of that row — t_rank[indptr[t_users[e]] + t_pos[e]] == e.  All four or none. */
  const int32_t *t_users;  /* [t_nnz] */
  const int32_t *t_pos;    /* [t_nnz] */
  const int32_t *t_items;  /* [t_nnz] */
} DrxHistory;

/* One mini-batch (one fit() "epoch", recommender_abc.py:189-205).
 * uid  : the sampled users (PointSampler triples' first field, cdae.py:52).
 * iid,y: sampled output unit and its {0,1} target — used by the sampled-output mode only.
 * Corruption (cdae.py:63): entry j of user uid[b]'s history survives iff
 *     keep != NULL ?  keep[keep_off[b] + j] != 0            (host MT19937 stream, parity mode)
 *                  :  drx_hash_u32(mask_seed, b, j) >= q * 2^32  (counter-based, throughput mode)
 * keep_off[b] = sum_{b' < b} deg(uid[b'])  (exclusive prefix sum, keep_off[B] = total). */
typedef struct DrxBatch {
  int32_t B;
  const int32_t *uid;       /* [B] */
  const int32_t *iid;       /* [B] or NULL */
  const float   *y;         /* [B] or NULL */
  const int32_t *keep_off;  /* [B + 1] */
  const uint8_t *keep;      /* [keep_off[B]] or NULL */
  uint64_t mask_seed;
  float q;                  /* corruption level; survivors are scaled by 1/(1-q) */
  int32_t n_touch_slots;    /* host-known upper bound of keep_off[B] (sizes the sort) */
  uint32_t flags;           /* DRX_BATCH_* (0: none) */
} DrxBatch;
/* Sampled mode, batches in which users REPEAT (B several times the number of users: MovieLens shapes), prepared through the history's
 * transpose (DrxHistory::t_rank; lists of long segments): the triples of one user share their gather and their gradient.  The samples
 * of a user, ascending, are cut into WORK ITEMS of up to 16 triples (csrc/drx_prep.hpp k_tp_item_*):
 *   forward : one workgroup per work item loads every row of the user's history ONCE and adds it into the bags of the item's triples
 *             under their keep bits — a masked matrix product [16 triples x history] x [history x K] (csrc/drx_cdae.hip
 *             k_items_fwd_bwd, v_mfma_f32_16x16x4_f32: fp32, products with 0 / 1) — instead of one gather per triple;
 *   backward: an item row of the user's history receives, per work item,  D_w - sum of dz1[b] over the triples b of the item that
 *             DROPPED it,  D_w = sum of dz1 over the item's triples (a row of its own behind the samples' gradient rows) — where
 *             that is the shorter form (1 + droppers < keepers), else one touch per keeper as without the flag.
 * With corruption level q the touch list shrinks towards q + 1/16 of its plain length (ml-1m shape: 8.0 M -> 2.9 M touches), the rows
 * the forward kernel loads to 1 / (triples per work item) (8.0 M -> 1.0 M).  Another association of the same sums: same oracle, same
 * tolerance; bit-reproducible (every order is a function of the batch).  Prepare and step must see the same flag; it is ignored
 * (plain lists) where the transposed preparation does not apply, rows are narrower than 33 / wider than 255 floats, or the step
 * builds its list inline (drx_cdae_step_sparse: only lists prepared ahead, drx_cdae_sparse_prepare, take the shared form).
 * Single-GPU step only. */
#define DRX_BATCH_SHARE_USERS 1u

typedef struct DrxOptim {
  int32_t kind;             /* DRX_OPT_* */
  float lr, reg_rate;       /* reg is applied as reg_rate / B (cdae.py:82) */
  float beta1, beta2, eps;  /* Adam: .9 .999 1e-7 (Keras); Adagrad: eps 1e-7 */
  /* Keras-Adam lr_t = lr*sqrt(1-b2^t)/(1-b1^t) per variable in registration order W, W_, V, b, b_
   * (recommender_abc.py:328-334 advances the counter once per variable: t = 5*step + j + 1). */
  float alpha[5];
  float *s1[5];             /* slots in the order W, W2T, V, b, b2 */
  float *s2[5];
} DrxOptim;

int drx_version(void);
const char *drx_strerror(int code);

/* Stream-ordering events for the host's run-ahead pipelines (sampler / touch-list preparation on a side stream of the same device):
 * created with hipEventDisableTiming | hipEventDisableSystemFence, i.e. a record is an agent-scope release, not an L2 write-back
 * for host visibility.  NOT for host-side reads of device results other than through drx_event_synchronize + pinned memory. */
void *drx_event_create(void);
void drx_event_destroy(void *ev);
int drx_event_record(void *ev, void *stream);
int drx_stream_wait_event(void *stream, void *ev);
int drx_event_synchronize(void *ev);

/* A stream confined to a slice of the chip, for the run-ahead work that shares it with the training kernels (sampler, touch-list
 * preparation): hipExtStreamCreateWithCUMask.  cus_per_xcd CUs of EVERY XCD (1 .. 32 on gfx950; mask bit i names CU i / 8 of XCD i % 8 —
 * scripts/mb/mb_cumask.hip prints the census), taken from the top of each XCD's CUs.  The preparation's small launches then queue for
 * THEIR CUs instead of displacing the training kernels' waves everywhere.  Returns NULL when the runtime refuses. */
void *drx_stream_create_cu_slice(int32_t cus_per_xcd);
void drx_stream_destroy(void *stream);

/* 64-bit mix used for the counter-based corruption mask; exported so hosts/tests can reproduce it. */
uint32_t drx_hash_u32(uint64_t seed, uint32_t a, uint32_t b);

/* ---- inference ------------------------------------------------------------------------------
 * h[B,ld] = sigmoid(scale * sum_{kept} W[n] + V[uid] + b);  p[B,n_items] = sigmoid(h W_ + b_).
 * Replaces CDAE._reconstruct (cdae.py:73-76) as used by _predict/_rank (cdae.py:67-71,84-103:
 * keep == NULL && q == 0 -> uncorrupted, unscaled) and by _predict_batch (cdae.py:50-65).
 * p may be NULL (hidden layer only). */
int drx_cdae_forward(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt,
                     float *h, float *pred, void *stream);

/* ---- scratch sizing ---------------------------------------------------------------------- */
/* dense_mode != 0: drx_cdae_step_dense (per-row batch bitmasks: grows with B * (n_users + n_items) / 8 bytes);
 * dense_mode == 0: drx_cdae_step_sparse* (grows with the number of touches). */
size_t drx_cdae_scratch_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t dense_mode);

/* ---- one reference-mode training step ------------------------------------------------------
 * forward + Keras BCE/MSE against the batch-mean (or per-row) target over ALL output units +
 * L2/B on W, W_, V + backward + dense Adam on every parameter: replaces the body of the fit() loop
 * recommender_abc.py:190-204 for CDAE (cdae.py:50-82).  loss_out (device, 2 floats: prediction
 * loss, regularisation loss) may be NULL. */
int drx_cdae_step_dense(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist,
                        const DrxBatch *bt, int32_t loss_kind, int32_t targets_kind,
                        void *scratch, size_t scratch_bytes, float *loss_out, void *stream);

/* ---- one sampled-output training step (engine mode) ----------------------------
 * per triple (uid, iid, y): one output unit, loss mean over B, L2/B on touched rows, sparse
 * Adagrad / lazy Adam on touched rows of W, W2T, V, b2 and dense update of b.
 * alpha[0] is used as the Adam lr_t for every table. */
int drx_cdae_step_sparse(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist,
                         const DrxBatch *bt, int32_t loss_kind,
                         void *scratch, size_t scratch_bytes, float *loss_out, void *stream);

/* Same step with per-phase timing: `events` holds DRX_SPARSE_PHASES + 1 caller-created hipEvent_t; event i is recorded
 * on `stream` before phase i, the last one after the final phase.  Phases: 0 gather+forward+backward (k_sampled_fwd_bwd),
 * 1 touch sort, 2 segmented reduce + row update (k_seg_reduce), 3 tail launch A (short chunk-crossing segments, and beside
 * them the column-sum partials of the hidden-bias gradient), 4 tail launch B (long segments, and the hidden-bias update). */
#define DRX_SPARSE_PHASES 5
int drx_cdae_step_sparse_timed(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist,
                               const DrxBatch *bt, int32_t loss_kind, void *scratch, size_t scratch_bytes,
                               float *loss_out, void *const *events, void *stream);

/* The touch list of a batch (row keys sorted, with the contributing sample) depends only on the batch, not on the
 * parameters: it can be prepared for batch t+1 on another stream while batch t trains.  `prepared` is an opaque
 * device buffer of drx_cdae_prep_bytes() bytes; events may be NULL (else as in drx_cdae_step_sparse_timed; phase 1 is
 * then empty). */
size_t drx_cdae_prep_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots);
int drx_cdae_sparse_prepare(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, void *prepared,
                            size_t prepared_bytes, void *stream);
int drx_cdae_step_sparse_prepared(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                                  int32_t loss_kind, const void *prepared, size_t prepared_bytes, void *scratch,
                                  size_t scratch_bytes, float *loss_out, void *const *events, void *stream);

/* off[0] = 0, off[b + 1] = sum_{b' <= b} (indptr[ids[b'] + 1] - indptr[ids[b']]) for device ids: the keep_off of a batch of users (DrxBatch)
 * or the touch offsets of a batch of CSR rows.  scratch: drx_point_sample_scratch_bytes(B). */
int drx_batch_offsets(const int64_t *indptr, const int32_t *ids, int32_t B, int32_t *off, void *scratch, size_t scratch_bytes, void *stream);

/* ---- device-side point sampler (throughput mode; distribution of point_sampler.py:44-61) ------------
 * Draws B triples with a counter-based generator keyed by (seed, b): negatives with probability
 * neg_ratio/(neg_ratio+1) = uniform (u,i) outside u's positives, positives = uniform user then uniform positive.
 * Also writes keep_off[B+1] (exclusive prefix sum of deg(uid[b])).
 * host_mailbox (optional): 8 bytes of PINNED host memory; the last kernel stores ((uint64)tag << 32) | keep_off[B] there with one
 * system-scope store, so a host that runs ahead learns the batch's touch count by polling for its tag — no copy, no event. */
size_t drx_point_sample_scratch_bytes(int32_t B);
int drx_point_sample(const DrxHistory *hist, int32_t n_users, int32_t n_items, int32_t B, int32_t neg_ratio,
                     uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *keep_off,
                     void *scratch, size_t scratch_bytes, uint64_t *host_mailbox, uint32_t tag, void *stream);
/* recorded: CSR (columns ascending) of EVERY pair the training set records, whatever its value — the reference draws its negatives among
 * the pairs absent from the frame (point_sampler.py:56, mem_dataset.py:131-163), not among the non-positives; NULL = hist (frames whose
 * recorded pairs all are positives).  drx_point_sample = this with NULL. */
int drx_point_sample_recorded(const DrxHistory *hist, const DrxHistory *recorded, int32_t n_users, int32_t n_items, int32_t B,
                              int32_t neg_ratio, uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *keep_off,
                              void *scratch, size_t scratch_bytes, uint64_t *host_mailbox, uint32_t tag, void *stream);

/* The same draws (the same triples for the same seed), handed out SORTED BY USER, a user's triples in the order they were drawn —
 * for batches whose touch lists are prepared through the history's transpose (DrxHistory::t_rank): with the batch in user order the
 * expanded list has every row's touches sample-ascending, i.e. the reduction streams through the gradient rows instead of hopping,
 * and the triples of one user sit side by side in the forward kernel.  The order of a batch changes no sum's terms. */
size_t drx_point_sample_by_user_scratch_bytes(int32_t B, int32_t n_users);
int drx_point_sample_by_user(const DrxHistory *hist, const DrxHistory *recorded, int32_t n_users, int32_t n_items, int32_t B,
                             int32_t neg_ratio, uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *keep_off, void *scratch,
                             size_t scratch_bytes, uint64_t *host_mailbox, uint32_t tag, void *stream);

/* The same draws carrying VALUES (DMF.fit(device_sampler=True), dmf.py:64-73): y[b] = 0 for a negative, else the drawn pair's
 * interaction value pos_values[position in hist] — standardised (v - vmin) / vrange when vrange > 0 (use_nce:
 * recommender_abc.py:463-465), raw otherwise.  No keep_off / mailbox: DMF needs neither. */
int drx_point_sample_valued(const DrxHistory *hist, const DrxHistory *recorded, const float *pos_values, float vmin, float vrange,
                            int32_t n_users, int32_t n_items, int32_t B, int32_t neg_ratio, uint64_t seed, int32_t *uid, int32_t *iid,
                            float *y, void *stream);

/* ---- device-side list sampler (throughput mode; distribution of list_sampler.py:74-151 as caser.py:72-75 configures it) -----
 * The reference's ListSampler is ONE MT19937 stream: draw d depends on every draw before it, so a reference-exact fit() is bound by
 * one host thread (about 0.5 us per window).  This entry point draws B windows independently with the counter-based generator of
 * the other throughput modes, draw d from drx_hash_u32(seed, d, k):
 *   k = 0      group   = eligible[(h * n_eligible) >> 32]                 (rng.choice over the groups, ineligible ones re-drawn)
 *   k = 1      start   = (h * (rows - n_inputs - n_targets + 1)) >> 32    (rng.randint(0, rows - max_positive_records - n_targets))
 *   k = 2+16i+a  negative i, attempt a: the ((h * n_pop) >> 32)-th id of the ascending complement of the group's ids in
 *              [0, n_ids); an id already drawn for this window is drawn again (a = 1 .. 15), then the next free one is taken
 *              (rng.sample(eligible, n): uniform without replacement)
 * before[d] = the ids of rows start .. start + n_inputs - 1 of the group (rows in the sampler's sort order), after[d] = the ids of
 * the next n_targets rows followed by the n_targets * neg_ratio negatives.  Restated on the CPU by oracle/data_oracle.py::
 * list_sample_counter (the test compares bit for bit). */
typedef struct DrxListGroups {
  const int64_t *indptr;       /* [n_groups + 1] rows of every group */
  const int32_t *seq_ids;      /* [indptr[n_groups]] negative_ids_col value of every row, a group's rows in the sampler's order */
  const int64_t *held_indptr;  /* [n_groups + 1] */
  const int32_t *held;         /* ascending distinct ids among a group's rows */
  const int32_t *group_value;  /* [n_groups] the value of the group column (Caser: uid) */
  const int32_t *eligible;     /* [n_eligible] groups with >= n_inputs + n_targets rows and >= n_targets * neg_ratio ids outside */
  int32_t n_groups, n_eligible, n_ids;
} DrxListGroups;
int drx_list_sample_device(const DrxListGroups *g, int32_t B, int32_t n_inputs, int32_t n_targets, int32_t neg_ratio, uint64_t seed,
                           int32_t *group_out, int32_t *before, int32_t *after, void *stream);

/* ---- column-sharded ("K-sharded") multi-GPU step (no reference equivalent) ------------------------------------------------
 * Every rank holds ALL rows but only its own columns of W, W2T, V and b (DrxCdaeParams describes that slice: k = local columns;
 * b2 is replicated) and trains on the SAME global batch.  The one exchange of a step is the sum over ranks of the per-triple
 * partial dot products: forward -> all-reduce(dot_partial) -> step.  Everything else (touch list, segmented reduction, updates)
 * is the single-GPU step on K/N columns.  drx_cdae_kshard_step is drx_cdae_step_sparse(_prepared) with the forward half
 * replaced by (h, dot_total); `prepared` may be NULL (touch list built inline), `events` as in drx_cdae_step_sparse_timed. */
int drx_cdae_kshard_forward(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, float *h_out /* [B, ld] */,
                            float *dot_partial /* [B] */, void *stream);
/* the same with the batch's prepared list at hand: the triples are launched in the list's order of history lengths (longest first,
 * similar lengths side by side), like the single-GPU forward kernel; prepared == NULL: the batch's own order */
int drx_cdae_kshard_forward_prepared(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const void *prepared,
                                     size_t prepared_bytes, float *h_out /* [B, ld] */, float *dot_partial /* [B] */, void *stream);
int drx_cdae_kshard_step(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt, int32_t loss_kind,
                         const float *h, const float *dot_total, const void *prepared, size_t prepared_bytes, void *scratch,
                         size_t scratch_bytes, float *loss_out, void *const *events, void *stream);

/* The leading drx_cdae_prep_result_bytes of a `prepared` buffer are all that a step reads of it (the sorted list and the
 * sole-toucher marks): a buffer whose leading bytes were copied from another device's drx_cdae_sparse_prepare of the SAME batch
 * (same B, n_touch_slots, n_users, n_items) is a valid `prepared` argument — how one rank of a column-sharded job prepares the
 * list of a step for all the others. */
size_t drx_cdae_prep_result_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots);

/* Touch list prepared in PARTS (column-sharded layout: all ranks need the same list of the same batch, and sorting it on every
 * rank is the one cost that does not shrink with N).  Part r of `parts` = the touches of the rows with id % parts == r (item id for W
 * and W2T keys, user id for V keys), taken in sample order and sorted by key; the list of the batch is the
 * concatenation of the parts in order (equal keys adjacent, the touches of a key in sample order: all the segmented reduction needs).
 *   drx_cdae_sparse_prepare_part     -> part_out [drx_cdae_prep_part_out_bytes]: 4 int32 (touches, distinct keys, overflow, 0), then per
 *                                       distinct key (key << 32 | first position), then the samples of the touches — 4 bytes a touch
 *   (the host gathers the parts of all ranks, in rank order, into all_parts [parts * part_out_bytes])
 *   drx_cdae_sparse_prepare_assemble -> a `prepared` buffer like drx_cdae_sparse_prepare's; overflow_out[0] (device) = 1 when a part
 *                                       did not fit its fixed capacity (1.25 x the even share): that list is incomplete and the
 *                                       caller must prepare the batch with drx_cdae_sparse_prepare instead. */
size_t drx_cdae_prep_part_out_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t parts);
/* out4 = byte offset of the runs, byte offset of the samples, capacity in runs, capacity in touches (of one part) */
int drx_cdae_prep_part_layout(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t parts, size_t *out4);
size_t drx_cdae_prep_part_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t parts);
int drx_cdae_sparse_prepare_part(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, int32_t part, int32_t parts,
                                 void *part_out, size_t part_out_bytes, void *scratch, size_t scratch_bytes, void *stream);
int drx_cdae_sparse_prepare_assemble(const DrxCdaeParams *p, const DrxBatch *bt, const void *all_parts, int32_t parts, void *prepared,
                                     size_t prepared_bytes, int32_t *overflow_out, void *stream);

/* ---- row-sharded multi-GPU step (BASELINE.json configuration 4; no reference equivalent — DRecPy is single-process:
 * recommender_abc.py:16 is its only device line) -------------------------------------------------------------------------------------
 * Per-rank, collective-free pieces of the sampled step; the host (drecpy_amd/dist.py) runs the RCCL all-to-all exchanges between
 * them.  Users (V, histories, samples) are sharded by uid range; item rows (W, W2T, b2) by item range of `items_per_rank` rows; in
 * this mode DrxCdaeParams describes the LOCAL tables (n_users = local users, n_items = items_per_rank).
 *
 * WIRE keys of item rows are UNIT-major (r06).  Item n of owner o = n / ipr has the local key l = 2 * (n - o * ipr) + (W2T row ? 1 : 0);
 * with 1 << shift the smallest power of two >= max(2 * ipr, 8192) and C = `chunks` (a power of two; the library lowers it until
 * cshift = shift - log2 C >= 13), the local keys are cut into C EXCHANGE CHUNKS of 1 << cshift keys, unit v = (l >> cshift) * world + o,
 * and the wire key is (v << cshift) | (l & ((1 << cshift) - 1)).  A rank's distinct keys ("uniq"), ascending, are contiguous per unit,
 * each unit's run closed by one SENTINEL key DRX_KEY_NONE: n_v = distinct rows asked of owner v % world in chunk v / world, + 1.
 * Chunk c of every exchange (keys, rows, gradient rows) is ONE all-to-all over the units c * world .. c * world + world - 1, which are
 * contiguous in `uniq` and in the float buffers: the host pipelines the chunks (gradient rows of chunk c + 1 travel while the owner
 * applies chunk c and gathers chunk c's rows for the next step).  chunks <= 1: one all-to-all per exchange (the r04 / r05 format).
 *
 * EXCHANGE BUFFER (float32, the same geometry in both directions): for every unit, in unit order, a piece of n rows of ld floats
 * followed by n scalars padded to a multiple of 32 floats — n * ld + roundup(n, 32) floats (n = n_v on the requester's side, the
 * received counts on the owner's: one SEGMENT per source rank and chunk).  Row i of a piece answers key i of the unit's run.
 * Owner -> requester: the parameter row, scalar = b2 of a W2T row; the sentinel row is unused.  Requester -> owner: the rank's summed
 * gradient row, scalar = gradient of b2; the SENTINEL rows carry the rank's gradient of the replicated hidden bias b and, as their
 * scalar, the rank's loss sum — the owner adds those of the LAST chunk's `world` segments in rank order, the same rows on every rank,
 * so b needs no all-reduce. */
#define DRX_MAX_WORLD 64                /* world x chunks of one sharded job (per-unit offsets: DRX_MAX_WORLD + 1 entries) */
#define DRX_MAX_MICRO 4                 /* micro-batches of one step (their exchanges overlap each other's compute) */
#define DRX_MAX_CHUNKS 16               /* exchange chunks per owner */
typedef struct DrxShard {
  int32_t world, rank;
  int32_t n_items;          /* global number of items */
  int32_t items_per_rank;   /* ceil(n_items / world) */
  int32_t n_users_local;
  uint32_t flags;           /* DRX_SHARD_* */
  int32_t chunks;           /* exchange chunks per owner: 0 / 1 = one all-to-all per exchange, else a power of two <= DRX_MAX_CHUNKS */
} DrxShard;
/* the chunk count the library actually uses for a description (chunks lowered until every unit spans whole 8192-key tiles), and log2
 * of a unit's key span; 0 for an invalid description */
int32_t drx_shard_chunks(const DrxShard *sh);
int32_t drx_shard_unit_shift(const DrxShard *sh);
/* The rank's OWN item rows never pass through a collective: the forward kernel reads them from the local tables, the gather skips
 * the requests the rank sent to itself, and its own gradient pieces are read by drx_shard_apply where drx_shard_step_local left them.
 * The float exchange buffers then hold the units of the OTHER peers in unit order (split size 0 for the rank itself), and on the
 * requester's side the own units (one per chunk, in chunk order) follow them all (rows_cache never reads them, grad_send keeps
 * them).  The KEY exchange is unchanged. */
#define DRX_SHARD_SELF_BYPASS 1u

/* Parameter-independent preparation of one local batch, built ahead on a side stream: the sorted touch list with its span plan,
 * sole-toucher marks and launch order (as drx_cdae_sparse_prepare), plus the batch's distinct item rows: wire keys with sentinels, their
 * positions in the exchange buffers, per-owner counts.  `prepared` (256-byte aligned, drx_shard_prep_bytes) is opaque except for the
 * two arrays drx_shard_prep_layout names: out4 = { byte offset of uniq (uint32[capacity]), byte offset of counts (int64[world x chunks],
 * OWNER-major: counts[o * chunks + c] = n of unit c * world + o — what the count exchange sends, `chunks` entries per peer),
 * capacity of uniq in keys, bytes a step reads }.  `work` (drx_shard_work_bytes) must be ZERO when first handed in; the library
 * leaves it zero after every call (it may be shared by all preparations issued on one stream). */
size_t drx_shard_prep_bytes(const DrxCdaeParams *p, const DrxShard *sh, int32_t B, int32_t n_touch_slots);
size_t drx_shard_work_bytes(const DrxShard *sh);
int drx_shard_prep_layout(const DrxCdaeParams *p, const DrxShard *sh, int32_t B, int32_t n_touch_slots, size_t *out4);
int drx_shard_prepare(const DrxCdaeParams *p, const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt, void *prepared,
                      size_t prepared_bytes, void *work, size_t work_bytes, void *stream);
/* Owner side, ONE EXCHANGE CHUNK per call.  The n keys a rank received for chunk `chunk` are `n_segments` = world x (micro-batches of
 * the step, <= DRX_MAX_MICRO) segments, micro-batch-major then source rank, recv_counts[s] (HOST array, sentinel included) keys in
 * segment s.  drx_shard_owner_index: parameter-independent, run when the keys arrive: table[local key][segment] = index of the key in
 * the chunk's key array (drx_shard_owner_table_bytes; ONE table for all chunks of a step — a call clears and fills the rows of its
 * chunk's key range only).  drx_shard_gather_rows: the requested rows + scalars into an exchange buffer of the geometry above
 * (segment after segment). */
size_t drx_shard_owner_table_bytes(const DrxShard *sh, int32_t n_segments);
int drx_shard_owner_index(const DrxShard *sh, const uint32_t *recv_keys, int32_t n, const int32_t *recv_counts, int32_t n_segments,
                          int32_t chunk, void *table, size_t table_bytes, void *stream);
int drx_shard_gather_rows(const DrxCdaeParams *p, const DrxShard *sh, const uint32_t *recv_keys, int32_t n, const int32_t *recv_counts,
                          int32_t n_segments, float *out, void *stream);
/* Requester side, three launches: forward + backward of the local triples against the received rows (`rows_cache`; V rows and W2T
 * gradient rows only one triple touches are finished there), the planned segmented reduction and its span launch: one summed
 * gradient row per distinct item row into `grad_send` (same geometry as rows_cache), shared V rows updated in place, the bias
 * gradient + loss sum into the sentinel rows.  Losses and L2 are normalised by b_norm, the GLOBAL batch size.
 * events: NULL or 4 hipEvent_t recorded before / between / after the launches. */
size_t drx_shard_step_scratch_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots);
int drx_shard_step_local(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt,
                         const void *prepared, size_t prepared_bytes, const float *rows_cache, float *grad_send, int32_t b_norm,
                         int32_t loss_kind, void *scratch, size_t scratch_bytes, void *const *events, void *stream);
/* Owner side, one exchange chunk per call: sum the gradient rows received for each owned row of the chunk in segment order and apply
 * the optimizer ONCE per row; with the LAST chunk (chunk == chunks - 1) one extra workgroup sums the sentinel rows in segment order,
 * updates b and writes loss_out[0] = global loss sum / b_norm (loss_out may be NULL).  `table` = what drx_shard_owner_index built from
 * the same keys.  With DRX_SHARD_SELF_BYPASS: own_grad[m] = the grad_send buffer of the step's micro-batch m, own_off[m] = float offset
 * of the rank's own piece OF THIS CHUNK in it; otherwise both NULL. */
int drx_shard_apply(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, int32_t b_norm, const uint32_t *recv_keys,
                    const float *grad_recv, int32_t n, const int32_t *recv_counts, int32_t n_segments, int32_t chunk, const void *table,
                    const float *const *own_grad, const int64_t *own_off, float *loss_out, void *stream);

/* ---- RCCL transport of the row-sharded step (no reference equivalent; SURVEY §8b allows this one piece of state) ------------------------
 * One communicator per rank (process) with a HIP stream of its own.  drx_comm_unique_id on one rank, the 128 bytes handed to the others
 * by the host (torch.distributed's store, a file, ...), drx_comm_create on every rank with its device current.  librccl is opened at run
 * time: without it these calls return DRX_ECOMM and everything else in the library works.
 * drx_comm_alltoallv: peer p is sent send[send_off[p] .. + send_bytes[p]) and this rank receives recv_bytes[p] bytes from it at
 *   recv + recv_off[p] (HOST arrays of `world` entries; zero sizes allowed, the rank itself included) — one group of ncclSend / ncclRecv
 *   pairs on the communicator's stream, behind everything `after_stream` has queued so far.  Returns a ticket (>= 0) or a negative code.
 * drx_comm_wait: `stream` waits for the exchange with that ticket (and, the communicator's stream being in order, all before it).
 * The buffers must stay allocated until a stream that waited for the ticket has passed that point.
 * DRX_COMM_THREAD: the nccl* calls are made by a thread of the communicator's own (drx_comm_alltoallv only posts the request; the buffers'
 * addresses and sizes are copied); drx_comm_wait then first waits (on the host, microseconds) until that thread has enqueued the
 * exchange.  A failure of the issuing thread is returned by the next drx_comm_* call. */
#define DRX_COMM_ID_BYTES 128
#define DRX_COMM_THREAD 1u
typedef struct DrxComm DrxComm;
int drx_comm_unique_id(void *id128);
int drx_comm_create(const void *id128, int32_t world, int32_t rank, uint32_t flags, DrxComm **out);
int drx_comm_destroy(DrxComm *c);
void *drx_comm_stream(DrxComm *c);
int64_t drx_comm_alltoallv(DrxComm *c, const void *send, const int64_t *send_off, const int64_t *send_bytes, void *recv,
                           const int64_t *recv_off, const int64_t *recv_bytes, void *after_stream);
int drx_comm_wait(DrxComm *c, int64_t ticket, void *stream);
const char *drx_comm_last_error(void);

/* ---- the exchange PHASES of a row-sharded step, issued from C (r06; no reference equivalent) -------------------------------------------
 * What drecpy_amd/dist.py does between the drx_shard_* launches — offsets of every unit's piece, one drx_comm_alltoallv per chunk, the
 * waits — as four calls per step with the split sizes READ FROM THE COUNT EXCHANGE'S PINNED MAILBOX: the chunked schedule of a step is
 * 2 + 3 C exchanges, 2 C waits and 3 C launches, and issued from Python the host was the next wall behind the links.
 * One micro-batch per step; one DrxShardExchange per prepared batch, filled by the caller:
 *   send_counts / recv_counts  HOST arrays [world x chunks] as the count exchange delivers them: send_counts[o * chunks + c] = rows
 *                              (sentinel included) this rank asks of owner o in chunk c, recv_counts[s * chunks + c] = rows rank s asks
 *                              of this rank in chunk c
 *   uniq                       device: the rank's distinct wire keys (drx_shard_prepare)
 *   req                        device: the keys this rank receives, chunk after chunk, per chunk source after source (sizes[2] keys)
 *   table                      device: drx_shard_owner_table_bytes(sh, world)
 *   rows_cache, grad_send      device: the requester's exchange buffers (sizes[0] floats each; geometry: "EXCHANGE BUFFER" above)
 *   rows_send, grad_recv       device: the owner's (sizes[1] floats each): chunk after chunk, per chunk one segment per source rank
 *   rows_ticket / grad_ticket  written by the phases: the communicator tickets of the chunk's row / gradient exchange
 * drx_shard_exchange_sizes: sizes4 = { requester floats, owner floats, keys received, keys sent } (every buffer: at least 32 elements).
 * drx_shard_phase_keys  (parameter-independent, on the run-ahead stream): per chunk the key all-to-all, then drx_shard_owner_index.
 * drx_shard_phase_rows  one chunk: drx_shard_gather_rows of the rows asked of this rank as they are NOW, then their all-to-all
 *                       (-> rows_ticket[chunk]).  Head of a step for all chunks, or — pipelined — by drx_shard_phase_tail.
 * drx_shard_phase_local `stream` waits for the row tickets, then drx_shard_step_local (events as there).
 * drx_shard_phase_tail  posts the gradient all-to-all of every chunk; then chunk by chunk: wait, drx_shard_apply, and — `next` not NULL —
 *                       drx_shard_phase_rows(next, chunk): the rows of a chunk leave for the NEXT step right behind their update. */
typedef struct DrxShardExchange {
  const int64_t *send_counts;
  const int64_t *recv_counts;
  const uint32_t *uniq;
  uint32_t *req;
  void *table;
  size_t table_bytes;
  float *rows_cache;
  float *rows_send;
  float *grad_send;
  float *grad_recv;
  int64_t rows_ticket[DRX_MAX_CHUNKS];
  int64_t grad_ticket[DRX_MAX_CHUNKS];
} DrxShardExchange;
int drx_shard_exchange_sizes(const DrxCdaeParams *p, const DrxShard *sh, const int64_t *send_counts, const int64_t *recv_counts,
                             int64_t *sizes4);
/* the phases' geometry for tests (host arithmetic only): out[12 world + 3], see csrc/drx_shard_phase.cpp */
int drx_shard_phase_layout(const DrxCdaeParams *p, const DrxShard *sh, const int64_t *send_counts, const int64_t *recv_counts, int32_t chunk,
                           int64_t *out);
int drx_shard_phase_keys(const DrxShard *sh, DrxComm *comm, DrxShardExchange *x, void *stream);
int drx_shard_phase_rows(const DrxCdaeParams *p, const DrxShard *sh, DrxComm *comm, DrxShardExchange *x, int32_t chunk, void *stream);
int drx_shard_phase_local(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt,
                          DrxComm *comm, DrxShardExchange *x, const void *prepared, size_t prepared_bytes, int32_t b_norm,
                          int32_t loss_kind, void *scratch, size_t scratch_bytes, void *const *events, void *stream);
int drx_shard_phase_tail(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, DrxComm *comm, DrxShardExchange *x,
                         DrxShardExchange *next, int32_t b_norm, float *loss_out, void *stream);

/* drx_copy_f4: dst[0 .. n_bytes) = src[0 .. n_bytes) on the device (16-B aligned, n_bytes a multiple of 16, no overlap): a streaming
 *   float4 copy kernel — table snapshots (recommender_abc.py:336-352 keeps a copy of every weight per epoch), and the rate bench.py
 *   reports as `hbm_copy_achievable` (the achievable HBM rate measured on the box next to the nominal peak). */
int drx_copy_f4(void *dst, const void *src, size_t n_bytes, void *stream);
/* the same with the kernel's form chosen by the caller (0 .. 4: csrc/drx_generic.hip; scripts/copy_bench.py times them side by side) */
int drx_copy_f4_variant(void *dst, const void *src, size_t n_bytes, int32_t variant, void *stream);

/* ---- model-independent pieces of the dense (Keras-Adam) steps of DMF / Caser ------------------------------------
 * drx_adam_dense: p, m, v [n] (16-B aligned): g_total = g + l2_coef * p (g may be NULL); TF ApplyAdam update with the
 *   given lr_t `alpha`.  One call = one optimizer.apply_gradients on one variable (recommender_abc.py:328-334). */
int drx_adam_dense(float *p, float *m, float *v, const float *g, int64_t n, float alpha, float l2_coef, float beta1, float beta2,
                   float eps, void *stream);
/* drx_scatter_rows: out[key] = sum over touches t with keys[t] == key of coef[t] * src[src_index[t]]  (and
 *   out_s[key] = sum coef[t] * src_s[src_index[t]]), summed in touch order (deterministic).  Rows of `out` that no
 *   touch names are left untouched (zero them first).  keys may hold DRX_KEY_NONE (ignored).  This is the gradient
 *   of an embedding lookup (tf.nn.embedding_lookup / Keras Embedding, caser.py:99-100,117-118; dmf.py:89-90 first layer). */
size_t drx_scatter_scratch_bytes(int32_t ld, int32_t n_touches, int32_t n_rows);
/* The same gradient for a table whose optimizer is DENSE (Keras Adam on an Embedding / Dense kernel: every row's moments decay every
 * step), fused with that optimizer's update — replaces drx_scatter_rows + drx_adam_dense (tape.gradient through the lookup and
 * apply_gradients, recommender_abc.py:203-204, 328-334; caser.py:47-50, 66-69) without touch keys on the device, a sort or a zeroed
 * gradient table.  row_ptr [n_rows + 1] / order [T]: the lookups grouped by the row they name, a row's lookups in batch order
 * (drx_batch_csr on the host; device copies here).  Row r: g = sum of src[order[q]] for q in [row_ptr[r], row_ptr[r + 1]), ascending;
 * p, m, v <- ApplyAdam(g + l2_coef * p) with lr_t = alpha.  src_s / p_s / m_s / v_s (all or none): one scalar per lookup / row
 * updated the same way with alpha_s and no l2 (caser.py:69's dense_1_b next to dense_1_W). */
int drx_rows_csr_adam(const int32_t *row_ptr, const int32_t *order, const float *src, const float *src_s, int32_t ld, int32_t n_rows,
                      float *p, float *m, float *v, float *p_s, float *m_s, float *v_s, float alpha, float alpha_s, float l2_coef,
                      float beta1, float beta2, float eps, void *stream);
/* The same where the gradient row of lookup o is an OUTER PRODUCT the producer did not write out: scale[o] * src[o / group]
 * (src [T / group, ld]; `group` consecutive lookups share a row of src), and scale[o] itself is the lookup's scalar gradient (p_s, m_s,
 * v_s required).  Caser's dense_1: the gradient of dense_1_W[n] through target j of sample b is dscore[b, j] * [dense_0 output | user
 * row](b) (caser.py:115-120) — drx_caser_fwd_bwd with dW1 == NULL writes the B hidden rows (cat_out) and the B * Tp score gradients
 * (db1) instead of B * Tp rows: 1.6 MB instead of 19.7 MB at examples/caser.py's batch of 4096. */
int drx_rows_csr_adam_outer(const int32_t *row_ptr, const int32_t *order, const float *scale, const float *src, int32_t group, int32_t ld,
                            int32_t n_rows, float *p, float *m, float *v, float *p_s, float *m_s, float *v_s, float alpha, float alpha_s,
                            float l2_coef, float beta1, float beta2, float eps, void *stream);
/* Several tables in ONE launch (ld <= 256 each): the arguments of drx_rows_csr_adam per table; group > 0 selects the outer-product form
 * of drx_rows_csr_adam_outer (scale required), group == 0 the plain one (scale = the optional per-lookup scalars src_s). */
#define DRX_MAX_CSR_TABLES 4
typedef struct DrxCsrAdamTable {
  const int32_t *row_ptr, *order;
  const float *src, *scale;
  int32_t group, ld, n_rows;
  float *p, *m, *v, *p_s, *m_s, *v_s;
  float alpha, alpha_s, l2_coef;
} DrxCsrAdamTable;
int drx_rows_csr_adam_multi(const DrxCsrAdamTable *tables, int32_t n_tables, float beta1, float beta2, float eps, void *stream);
/* host: first[c] = first position of code c in codes[0..n), -1 when absent; returns the number of distinct codes (Dataset.unique on
 * dense integer code columns, mem_dataset.py's drop_duplicates, without a sort) */
int64_t drx_first_occurrence(const int64_t *codes, int64_t n, int64_t n_codes, int64_t *first);
/* host: 1 as soon as *p >= at_least, 0 after poll_us microseconds of polling (the hand-over between fit()'s issuing thread and its
 * sampling worker thread: drecpy_amd/_spinpool.py; recommender_abc.py:186-188 draws the batch inline) */
int drx_spin_until(const int64_t *p, int64_t at_least, int32_t poll_us);
/* host: keys [T] in [0, n_rows) -> row_ptr [n_rows + 1], order [T] (stable counting sort) */
int drx_batch_csr(const int32_t *keys, int32_t T, int32_t n_rows, int32_t *row_ptr, int32_t *order);
/* The same on the DEVICE, for up to DRX_MAX_CSR_TABLES key lists in one go (batches drawn on the device: Caser.fit(device_sampler=True)
 * then takes the fused drx_rows_csr_adam_multi update too, instead of scatter + dense Adam per table): ONE stable sort of all lists'
 * (row, lookup) pairs, row_ptr by a binary search per row.  keys must lie in [0, n_rows) (not checked on the device).
 * scratch: drx_batch_csr_device_bytes (0 = invalid description). */
typedef struct DrxCsrList {
  const int32_t *keys;       /* device [T] */
  int32_t T, n_rows;
  int32_t *row_ptr, *order;  /* device [n_rows + 1], [T] */
} DrxCsrList;
size_t drx_batch_csr_device_bytes(const DrxCsrList *lists, int32_t n_lists);
int drx_batch_csr_device(const DrxCsrList *lists, int32_t n_lists, void *scratch, size_t scratch_bytes, void *stream);
int drx_scatter_rows(const uint32_t *keys, int32_t T, const float *src, const uint32_t *src_index, const float *coef,
                     const float *src_s, int32_t ld, int32_t n_rows, float *out, float *out_s, void *scratch,
                     size_t scratch_bytes, void *stream);
/* drx_sumsq: out[0] (+)= sum_i x[i]^2, accumulated in double in a fixed order; out = device double [1 + 1024] (out[1..] is scratch).
 * The VALUE of an L2 term for the loss log (cdae.py:82 tf.nn.l2_loss, the Keras l2 regularizers of dmf.py / caser.py) — the training
 * path never needs it (the update kernels apply reg * p themselves). */
int drx_sumsq(const float *x, int64_t n, double *out, int32_t accumulate, void *stream);
/* drx_rows_dot: out[b, n] = x[b, :] . table[n, :] + bias[n]   (all-item scoring, caser.py:137) */
int drx_rows_dot(const float *x, int32_t B, const float *table, int32_t n_rows, int32_t ld, const float *bias, float *out,
                 void *stream);

/* Adam over up to DRX_MAX_SEGMENTS slices of one flat array, each with its own Keras lr_t and L2 coefficient: the
 * small conv/dense weights of a model, one slice per registered layer kernel / bias (recommender_abc.py:328-334). */
#define DRX_MAX_SEGMENTS 24   /* (Caser with L = 8: 2 L + 4 = 20 registered kernels and biases) */
typedef struct DrxAdamSegments {
  int32_t n;
  int32_t start[DRX_MAX_SEGMENTS], len[DRX_MAX_SEGMENTS];
  float alpha[DRX_MAX_SEGMENTS], l2_coef[DRX_MAX_SEGMENTS];
} DrxAdamSegments;
int drx_adam_segments(float *p, float *m, float *v, const float *g, const DrxAdamSegments *sg, float beta1, float beta2, float eps,
                      void *stream);

/* ---- Caser (DRecPy/Recommender/caser.py) ------------------------------------------------------------------------
 * Tables: item_emb [N, ld], user_emb [U, ld] (caser.py:47-50), W1 [N, ld2] = dense_1_W rows (first d columns meet
 * dense_0's output, the next d the user embedding, caser.py:66,115-120), b1 [N] (caser.py:69).  Small weights `sw`
 * (one flat array, channel-fastest): conv_v kernel [L][n_v][ld] at off_kv + bias [n_v] at off_bv (caser.py:53);
 * convs_h[i] kernel [i+1][n_h][ld] at off_kh[i] + bias [n_h] at off_bh[i] (caser.py:55-58); dense_0 kernel
 * [n_v + L*n_h][ld] at off_wd + bias [ld] at off_bd (caser.py:63). */
enum { DRX_ACT_RELU = 0, DRX_ACT_TANH = 1, DRX_ACT_SIGMOID = 2, DRX_ACT_LINEAR = 3 };
typedef struct DrxCaserDims {
  int32_t L, T, Tp, d, ld, ld2, n_v, n_h, n_small;   /* Tp = T + T*neg_ratio targets per sample */
  int32_t off_kv, off_bv, off_kh[8], off_bh[8], off_wd, off_bd;
  int32_t act_h, act_mlp;   /* DRX_ACT_*: activation of the horizontal convolutions and of dense_0 (caser.py:29-30; default relu) */
} DrxCaserDims;
typedef struct DrxCaserArgs {
  const float *item_emb, *user_emb, *W1, *b1, *sw;
  const int32_t *uid;      /* [B] */
  const int32_t *before;   /* [B, L]  last L items (caser.py:79-83) */
  const int32_t *after;    /* [B, Tp] T targets then T*neg negatives */
  const uint8_t *keep;     /* [B, n_v + L*n_h] dropout keep mask (TF's RNG cannot be reproduced) or NULL = no dropout */
  float rate;
  int32_t B;
  float *dE;               /* [B*L, ld]   gradient row of every item lookup */
  float *dW1;              /* [B*Tp, ld2] gradient row of every dense_1_W lookup = db1[row] * cat_out[row / Tp]; NULL: not written
                            * (cat_out required then: drx_rows_csr_adam_outer forms the rows where it sums them) */
  float *db1;              /* [B*Tp] */
  float *dPu;              /* [B, ld]     gradient row of every user lookup */
  float *gsw_part;         /* [drx_caser_grid(), n_small] */
  float *loss_part;        /* [drx_caser_grid()] */
  float *cat_out;          /* [B, ld2] [dense_0 output | user row] of every sample: drx_caser_hidden; drx_caser_fwd_bwd when non-NULL */
  uint64_t mask_seed;      /* keep == NULL and rate > 0: dropout keeps element (b, j) iff drx_hash_u32(mask_seed, b, j) >= rate * 2^32 —
                            * a counter-based mask evaluated in the kernel (TF's dropout stream cannot be reproduced either way) */
} DrxCaserArgs;
int drx_caser_grid(const DrxCaserDims *D, int32_t B);
/* forward + Keras BCE + backward (caser.py:86-120 under the tape of recommender_abc.py:191-203): fills the lookup
 * gradient rows above and gsw_out[0..n_small) = gradient of the small weights, gsw_out[n_small] = prediction loss. */
int drx_caser_fwd_bwd(const DrxCaserDims *D, const DrxCaserArgs *A, float *gsw_out, void *stream);
/* The same followed by the small weights' update in the launch that sums the workgroups' partial gradients: gsw_out as above, then
 * sw, sw_m, sw_v <- l2 + Keras Adam per segment (drx_adam_segments' arithmetic; sw must be A->sw: the step's own weights, updated in
 * place once the training kernel has read them). */
int drx_caser_step_small(const DrxCaserDims *D, const DrxCaserArgs *A, float *gsw_out, float *sw, float *sw_m, float *sw_v,
                         const DrxAdamSegments *sg, float beta1, float beta2, float eps, void *stream);
/* inference hidden state cat_out[b] = [dense_0 output | user embedding] (caser.py:97-115 with training=False) */
int drx_caser_hidden(const DrxCaserDims *D, const DrxCaserArgs *A, void *stream);

/* ---- DMF (DRecPy/Recommender/dmf.py) ---------------------------------------------------------------------------
 * Tower 0 = user_nn (input: the user's interaction ROW over items), tower 1 = item_nn (input: the item's COLUMN over
 * users), both l2-normalised when l2_norm_vectors (dmf.py:75-86).  First-layer kernels K0u [n_items, ld0[0]] and
 * K0i [n_users, ld0[1]] are tables; every other weight lives in one flat array `sw`: layer l >= 1 kernel [f[l-1]][f[l]]
 * at off_k[tw][l], layer l >= 0 bias [f[l]] at off_b[tw][l].  Layer widths <= 64, <= 4 layers per tower.
 * gsw_out of drx_dmf_fwd_bwd holds the gradient of `sw` slot for slot (the scalar at off_scale included). */
typedef struct DrxDmfDims {
  int32_t n_layers[2];
  int32_t f[2][4];
  int32_t ld0[2];
  int32_t off_k[2][4], off_b[2][4];
  int32_t n_small;
  int32_t l2_norm_vectors;
  int32_t off_scale;         /* offset in `sw` of a registered scalar that multiplies every prediction, or -1: the extra
                              * tf.Variable of examples/extending_recommender_dmf.py:9-18 (ModifiedDMF, BASELINE config 3) */
} DrxDmfDims;
typedef struct DrxDmfArgs {
  const float *K0u, *K0i, *sw;
  const int64_t *u_indptr; const int32_t *u_indices; const float *u_values;   /* CSR [U, N], raw interaction values */
  const int64_t *i_indptr; const int32_t *i_indices; const float *i_values;   /* CSC (= CSR of the transpose) [N, U] */
  const int32_t *uid, *iid;  /* [B] */
  const float *y;            /* [B] targets (standardised when use_nce, dmf.py:69) */
  int32_t target_mode;       /* 0: BCE(y[b], pred[b]) (dmf.py:98-99).  1: BCE(y_mean, pred[b]) = the mean of Keras' (B,B) broadcast of
                              * (B,) targets against (B,1) predictions (ModifiedDMF's list of (1,)-tensors) */
  float y_mean;              /* mean of y over the batch (target_mode 1) */
  const int32_t *off_u, *off_i;   /* [B+1] prefix sums of the row / column lengths of the batch */
  int32_t B;
  float *dz0u, *dz0i;        /* [B, ld0] gradient wrt the first-layer pre-activation */
  uint32_t *tkeys_u, *tsrc_u; float *tcoef_u;   /* [off_u[B]] first-layer touches: kernel row, sample, input value */
  uint32_t *tkeys_i, *tsrc_i; float *tcoef_i;   /* [off_i[B]] */
  float *gsw_part, *loss_part;                  /* [drx_dmf_grid(B), n_small], [drx_dmf_grid(B)] */
  float *pred_out;           /* [B]      (drx_dmf_predict) max(1e-6, cosine) */
  float *rep_u_out, *rep_i_out;   /* [B, 64] (drx_dmf_predict, optional) l2-normalised tower outputs */
  float *work;               /* drx_dmf_work_bytes(B) bytes (drx_dmf_fwd_bwd): per-id first-layer sums, per-sample activations and
                              * pre-activation gradients handed from the gather kernel to the dense one and to the weight-gradient one */
  /* drx_dmf_fwd_bwd works on the DISTINCT users / items of the batch (a tower's first layer depends on the id alone): there
   * uid [n_du] / iid [n_di] hold the distinct ids, off_u [n_du+1] / off_i [n_di+1] their touch offsets, dz0u [n_du, ld0] / dz0i
   * [n_di, ld0] and the touches' sample field refer to distinct ids, and
   *   inv_u / inv_i [B]        : index of sample b's user / item among the distinct ones
   *   gptr_* [n_d+1], grows_* [B] : CSR of the samples per distinct id, samples ascending (order of the gradient sums)
   * (drx_dmf_predict takes uid / iid per pair as before and ignores these) */
  const int32_t *inv_u, *inv_i, *gptr_u, *gptr_i, *grows_u, *grows_i;
  int32_t n_du, n_di;
  /* Alternative to the touches (tkeys_* / tsrc_* / tcoef_* / off_* may then be NULL): drx_dmf_fwd_bwd records, for every distinct id
   * of the batch, map[id] = (stamp << 32) | its distinct index, for drx_dmf_k0_update.  map_u [n_users] / map_i [n_items]:
   * zero-initialised ONCE by the caller, never cleared; stamp: non-zero and different in every step. */
  unsigned long long *map_u, *map_i;
  /* l2 normalisers of every user row [n_users] / item column [n_items] of the interaction matrix (dmf.py:75-86), computed once per
   * dataset by drx_dmf_norms: required with the maps, optional otherwise (NULL: drx_dmf_fwd_bwd forms them per batch id) */
  const float *rho_u, *rho_i;
  uint32_t stamp;
  /* A batch prepared ON THE DEVICE (drx_dmf_batch_distinct_device): the numbers of distinct ids and the batch mean of y are only known
   * there.  nd_dev [2] = {n_du, n_di} (n_du / n_di above then are upper bounds that size the launches), y_mean_dev [1]; or NULL. */
  const int32_t *nd_dev;
  const float *y_mean_dev;
  /* r06, optional (NULL: one work item per distinct id, in their own order; ignored with nd_dev): the first-layer gather's n_work work
   * items, entry = distinct index (users 0 .. n_du - 1, items n_du ..) | segment << 24.  LONGEST rows / columns FIRST — the gather of a
   * popular item's column (thousands of non-zeros) is the launch's critical path when it happens to start late — and long ones CUT
   * into segments of seg_len non-zeros (0: uncut): segment 0 leaves the id's first-layer sum, segment g > 0 the partial row
   * zpart[(zseg[i] >> 8) + g - 1] (rows of 64 floats per unit slot), which the dense kernel adds in segment order (zseg[i] & 255 of
   * them).  drx_dmf_work_order builds order and zseg on the host. */
  const int32_t *work_order;
  int32_t n_work, seg_len;
  const int32_t *zseg;
  float *zpart;
  /* the list built on the DEVICE (drx_dmf_work_order_device, for batches prepared there: nd_dev set): n_work_dev[0] = its entries —
   * n_work above is then the list's CAPACITY and sizes the launch; NULL: a host-built list (ignored with nd_dev), or none */
  const int32_t *n_work_dev;
} DrxDmfArgs;
/* Host helper for DrxDmfArgs::work_order / zseg: the gather's work items (distinct users 0 .. n_u - 1 with off_u[i + 1] - off_u[i]
 * non-zeros, then distinct items), ordered by the bit length of their degree, descending, stable inside a class — O(n) — and cut into
 * segments of at most seg_len non-zeros (0: uncut): entry = work index | segment << 24; zseg[i] = first partial row << 8 | number of
 * partial rows of work index i (0: none).  Returns the number of entries, DRX_ESCRATCH when they exceed order_cap (use seg_len = 0
 * then), DRX_EINVAL when a row would need more than 255 segments; *n_part = partial rows. */
int32_t drx_dmf_work_order(const int32_t *off_u, int32_t n_u, const int32_t *off_i, int32_t n_i, int32_t seg_len, int32_t *order,
                           int32_t order_cap, int32_t *zseg, int32_t *n_part);
/* The same list for a batch whose distinct ids live on the device (drx_dmf_batch_distinct_device: du / di, nd_dev = {n_du, n_di}): degrees
 * from the interaction matrix's two index pointers, one workgroup, classes by bit length of the degree (descending; inside a class in no
 * particular order: a work item writes its own row, the result does not depend on the order).  out2 = {entries, partial rows}.  The
 * caller guarantees that the entries fit: order_cap >= 2 B + 2 nnz / seg_len (every distinct id brings its own row or column once). */
int drx_dmf_work_order_device(const int64_t *u_indptr, const int64_t *i_indptr, const int32_t *du, const int32_t *di, const int32_t *nd_dev,
                              int32_t seg_len, int32_t *order, int32_t order_cap, int32_t *zseg, int32_t *out2, void *stream);
/* Host helper for the arrays above: the distinct ids of a batch, ascending.  distinct [<= B], inv [B], gptr [<= B+1], grows [B],
 * off [<= B+1] (prefix sums of indptr row lengths of the distinct ids; NULL to skip) are host arrays;
 * scratch = int32 [n_rows], all -1 on entry and again on return (the caller keeps it between steps).  Returns the number of distinct
 * ids, or a negative DRX_E* code (an id outside [0, n_rows); DRX_EINVAL when the prefix sums do not fit int32 — touch offsets would overlap). */
int32_t drx_batch_distinct(const int32_t *ids, int32_t B, int32_t n_rows, const int64_t *indptr, int32_t *scratch, int32_t *distinct,
                           int32_t *inv, int32_t *gptr, int32_t *grows, int32_t *off);
/* The same on the device, both towers at once, for batches drawn there (DMF.fit(device_sampler=True)): uid / iid / y [B] device
 * arrays in; du, di [B], inv_u, inv_i [B], gptr_u, gptr_i [B + 1], grows_u, grows_i [B], nd [2] = {distinct users, distinct items},
 * y_mean [1] out — what DrxDmfArgs wants (uid = du, iid = di, nd_dev = nd, y_mean_dev = y_mean).  One stable sort of the 2B
 * (id, sample) pairs + one workgroup that numbers the runs.  n_users + n_items < 2^31, B <= 2^20. */
size_t drx_dmf_distinct_scratch_bytes(int32_t B, int32_t n_users, int32_t n_items);
int drx_dmf_batch_distinct_device(const int32_t *uid, const int32_t *iid, const float *y, int32_t B, int32_t n_users, int32_t n_items,
                                  int32_t *du, int32_t *di, int32_t *inv_u, int32_t *inv_i, int32_t *gptr_u, int32_t *gptr_i,
                                  int32_t *grows_u, int32_t *grows_i, int32_t *nd, float *y_mean, void *scratch, size_t scratch_bytes,
                                  void *stream);
/* out[id] = 1 / max(|row id|_2, 1e-6) (1 when l2_norm_vectors == 0) for the n rows of a CSR, with the summation order of the towers */
int drx_dmf_norms(const DrxDmfDims *D, const int64_t *indptr, const float *values, int32_t n, float *out, void *stream);
int drx_dmf_grid(int32_t B);      /* rows of gsw_part / entries of loss_part */
size_t drx_dmf_work_bytes(int32_t B);
/* forward + Keras BCE + backward (dmf.py:88-99 under the tape): dz0 rows + touches for drx_scatter_rows,
 * gsw_out[0..n_small) small-weight gradients, gsw_out[n_small] = prediction loss */
int drx_dmf_fwd_bwd(const DrxDmfDims *D, const DrxDmfArgs *A, float *gsw_out, void *stream);
/* The same with the small weights' Keras Adam (what drx_adam_segments does with gsw_out afterwards) in the launch that sums the chunks'
 * partial gradients: one launch less per step (r06; drx_dmf_k0_update reads none of `sw` and may follow). */
int drx_dmf_step_small(const DrxDmfDims *D, const DrxDmfArgs *A, float *gsw_out, float *sw, float *sw_m, float *sw_v,
                       const DrxAdamSegments *sg, float beta1, float beta2, float eps, void *stream);
int drx_dmf_predict(const DrxDmfDims *D, const DrxDmfArgs *A, void *stream);
/* The update of the first-layer kernels (tf.keras Dense kernels [N, f0] of user_nn / [U, f0] of item_nn, dmf.py:44-60) after
 * drx_dmf_fwd_bwd ran with the id maps: gradient (+ l2_coef * K0, dmf.py:101-103) and dense Keras Adam in one pass over both tables —
 * replaces recommender_abc.py:328-334 for these two variables without a touch list, sort or gradient arena.  Row n of K0u collects
 * from column n of the interaction matrix (i_* of A) the entries whose user carries this step's stamp in map_u; K0i likewise from
 * u_* and map_i.  Cost per step: one walk over all non-zeros (8 bytes each) — pays while nnz is within ~64x the batch's touches;
 * beyond that use the touches + drx_scatter_rows + drx_adam_dense. */
typedef struct DrxDmfK0Update {
  float *K0u, *m_u, *v_u;    /* [n_items, ld0[0]] kernel, Adam moments */
  float *K0i, *m_i, *v_i;    /* [n_users, ld0[1]] */
  int32_t n_items, n_users;
  float alpha_u, alpha_i;    /* Keras-Adam lr_t of the apply_gradients call each kernel belongs to */
  float l2_coef, beta1, beta2, eps;
  /* r06, optional (NULL: rows in table order, a workgroup each): a permutation of 0 .. n_items + n_users - 1 (K0u rows, then K0i rows),
   * the rows with the longest columns / rows of the interaction matrix first (static per dataset).  Its first n_long rows take a
   * workgroup each — every row whose walk is longer than 1024 entries must be among them —, the others a WAVE each, four per workgroup. */
  const int32_t *row_order;
  int32_t n_long;
} DrxDmfK0Update;
int drx_dmf_k0_update(const DrxDmfDims *D, const DrxDmfArgs *A, const DrxDmfK0Update *up, void *stream);
/* all-pairs cosine scores on the matrix cores: out[u, n] = max(1e-6, ru[u,:kdim] . ri[n,:kdim]) with bf16 operands /
 * fp32 accumulation (v_mfma_f32_32x32x16_bf16); ru, ri = l2-normalised tower outputs, kdim % 16 == 0 (dmf.py:92-95
 * evaluated for a block of users against all items instead of one _predict per pair, recommender_abc.py:460). */
int drx_score_pairs_bf16(const float *ru, int32_t n_u, const float *ri, int32_t n_i, int32_t ld, int32_t kdim,
                         const float *scale /* device scalar multiplying every score, or NULL */,
                         float *out, int32_t out_ld /* floats between rows of out (>= n_i; a multiple of 32 keeps every 128-byte
                                                       line inside one 64 x 64 tile, i.e. one workgroup / one XCD) */, void *stream);

/* ---- ranking (cdae.py:90-103, recommender_abc.py:454-461) --------------------------------
 * For each of R rows of `scores` [R, n] select the top `k` entries among those with
 * cand_mask == NULL || bit (r*n + i) set; order = descending score, ties by larger index
 * (heapq.nlargest over (score, iid) tuples).  out_idx/out_val [R,k]; missing = -1 / -inf. */
size_t drx_topk_scratch_bytes_k(int32_t R, int32_t n, int32_t k);   /* 0 for n <= 16384 (LDS bitonic path, scratch may be NULL); longer rows:
                                                                       * the k keys a radix select picks per row (+ sort space when k > 16384) */
size_t drx_topk_scratch_bytes(int32_t R, int32_t n);                 /* the same for k = n (an upper bound for every k) */
int drx_topk(const float *scores, const uint32_t *cand_mask, int32_t R, int32_t n, int32_t k,
             int32_t *out_idx, float *out_val, void *scratch, size_t scratch_bytes, void *stream);

/* ---- stable device radix sort of (key, val) pairs (the inverted-index builder of the sparse steps; ties keep their input order).
 * Every key must be < 2^key_bits: the sort runs ceil(key_bits / digit) passes of 8-, 10- or 11-bit digits, i.e. it orders on
 * passes * digit >= key_bits bits — bits above key_bits are NOT ignored (DRX_EINVAL is not raised for them: the result is then ordered
 * on those bits too).  keys_out / vals_out must not alias the inputs.  temp: >= drx_sort_pairs_temp_bytes(n, key_bits). */
size_t drx_sort_pairs_temp_bytes(int64_t n, int32_t key_bits);
int drx_sort_pairs(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int32_t key_bits,
                   void *temp, size_t temp_bytes, void *stream);

/* ---- raw -> internal id map (mem_dataset.py:309-330) --------------------------------------
 * codes[r] = rank of first appearance of raw[r] (int64 raw ids), bit-exact.
 * n_unique (device int32) receives the number of categories; uniques[c] the raw id of code c.
 * scratch: >= drx_idmap_scratch_bytes(n). */
size_t drx_idmap_scratch_bytes(int64_t n);
int drx_idmap_build(const int64_t *raw, int64_t n, int32_t *codes, int64_t *uniques,
                    int32_t *n_unique, void *scratch, size_t scratch_bytes, void *stream);

/* ---- host-side sampler (point_sampler.py:44-61; mem_dataset.py:111-163) ---------------------
 * CPython-exact MT19937 streams (random.Random(seed): init_by_array, 53-bit random(),
 * getrandbits-rejection randint).  All pointers here are HOST memory. */
typedef struct DrxSampler DrxSampler;
DrxSampler *drx_sampler_create(const int32_t *h_uid, const int32_t *h_iid, const double *h_val, int64_t n_rows,
                               int32_t neg_ratio, int32_t has_threshold, double threshold, int64_t seed);
int drx_sampler_sample(DrxSampler *s, int32_t n, int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out);
/* kind: DRX_DRAW_MIXED = PointSampler.sample, DRX_DRAW_NEGATIVE = sample_negative, DRX_DRAW_POSITIVE = sample_positive
 * (point_sampler.py:74-96); each advances only the stream(s) the reference method advances.
 * h_neg_out (optional): 1 where the draw is a negative (a pair absent from the frame, value 0). */
enum { DRX_DRAW_MIXED = 0, DRX_DRAW_NEGATIVE = 1, DRX_DRAW_POSITIVE = 2 };
int drx_sampler_draw(DrxSampler *s, int32_t kind, int32_t n, int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out,
                     uint8_t *h_neg_out);
void drx_sampler_destroy(DrxSampler *s);

/* ---- host-side ListSampler (list_sampler.py:76-151) ---------------------------------------------------------------------
 * The draw loop of ListSampler.sample_group_records with the stream of random.Random(seed): choice of a group, window start,
 * inputs / targets, rng.sample over the eligible negative ids in CPython's set iteration order (the ids are dense ints in
 * [0, n_ids), all present in the dataset).  Groups come in unique_groups order; group g owns the dataset rows
 * grp_rows[grp_indptr[g] .. grp_indptr[g+1]) — already filtered by interaction_threshold and ordered by sort_column — whose
 * negative_ids_col values are grp_ids[...].  n_targets < 0 / max_positive < 0 mean None.  All pointers are HOST memory. */
typedef struct DrxListSampler DrxListSampler;
DrxListSampler *drx_list_sampler_create(const int64_t *grp_indptr, const int64_t *grp_rows, const int32_t *grp_ids, int32_t n_groups,
                                        int32_t n_ids, int32_t neg_ratio, int32_t n_targets, int32_t min_positive,
                                        int32_t max_positive, int64_t seed);
/* n draws: group_out[n]; in_off / tg_off / ng_off [n+1] prefix offsets into in_rows / tg_rows (dataset rows of the inputs /
 * targets) and neg_ids.  DRX_ESCRATCH: a capacity is too small; DRX_ERETRY: more than 20 consecutive failed attempts
 * (drx_list_sampler_last_hint: 1 = too few positive records, 2 = too few eligible negatives). */
int drx_list_sampler_sample(DrxListSampler *s, int32_t n, int32_t *group_out, int64_t *in_off, int64_t *in_rows, int64_t in_cap,
                            int64_t *tg_off, int64_t *tg_rows, int64_t tg_cap, int64_t *ng_off, int32_t *neg_ids, int64_t ng_cap);
int32_t drx_list_sampler_last_hint(const DrxListSampler *s);
void drx_list_sampler_destroy(DrxListSampler *s);

/* CDAE corruption stream (cdae.py:63, RecommenderABC._rng of recommender_abc.py:74): draws
 * n_items uniforms per row, in batch order, and writes keep flags for the positives of each row. */
typedef struct DrxRng DrxRng;
DrxRng *drx_rng_create(int64_t seed);
void drx_rng_destroy(DrxRng *r);
double drx_rng_random(DrxRng *r);
int64_t drx_rng_randint(DrxRng *r, int64_t a, int64_t b);
/* advances the stream by n_words 32-bit outputs without producing them (a uniform(0,1) draw consumes two) */
void drx_rng_discard(DrxRng *r, uint64_t n_words);
int drx_rng_corruption_keep(DrxRng *r, const int64_t *h_indptr, const int32_t *h_indices, int32_t n_items,
                            const int32_t *h_uid, int32_t B, double q,
                            int32_t *h_keep_off, uint8_t *h_keep, int64_t keep_capacity);

/* One reference-mode CDAE batch in one call, for worker threads drawing ahead of the training step (cdae.py:47-63: B sampler
 * triples, then N uniform draws per batch row): waits until *turn == ticket, draws (DRX_DRAW_MIXED), stores ticket + 1 into
 * *turn, advances `rng` by discard_words outputs (batches drawn by other workers from their own generator of the same seed)
 * and fills keep_off / keep like drx_rng_corruption_keep. */
int drx_cdae_reference_draw(DrxSampler *smp, DrxRng *rng, int64_t *turn, int64_t ticket, uint64_t discard_words,
                            const int64_t *h_indptr, const int32_t *h_indices, int32_t n_items, int32_t B, double q,
                            int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out, uint8_t *h_neg_out, int32_t *h_keep_off,
                            uint8_t *h_keep, int64_t keep_capacity);

/* Draw-ahead workers for the loop above: two native threads (one per corruption generator `gen`), each running the jobs submitted
 * to it in order through drx_cdae_reference_draw with a turn counter of their own (tickets 0, 1, 2, ... in submission order
 * across both).  They and drx_drawahead_wait poll for a few hundred microseconds before sleeping, so a fit() that needs a batch
 * every 40-70 us never pays a thread wake-up.  The buffers of a job belong to the caller and must stay valid until its wait
 * returns; at most 8 jobs per generator may be outstanding (DRX_ERETRY otherwise).  submit returns the job's index (>= 0). */
typedef struct DrxDrawAhead DrxDrawAhead;
DrxDrawAhead *drx_drawahead_create(DrxSampler *smp, DrxRng *rng0, DrxRng *rng1, const int64_t *h_indptr, const int32_t *h_indices,
                                   int32_t n_items);
int64_t drx_drawahead_submit(DrxDrawAhead *d, int32_t gen, int64_t ticket, uint64_t discard_words, int32_t B, double q,
                             int32_t *h_uid_out, int32_t *h_iid_out, double *h_val_out, uint8_t *h_neg_out, int32_t *h_keep_off,
                             uint8_t *h_keep, int64_t keep_capacity);
int drx_drawahead_wait(DrxDrawAhead *d, int32_t gen, int64_t job);
void drx_drawahead_destroy(DrxDrawAhead *d);

/* ---- the quiet fit() loop of reference-mode CDAE in one call ------------------------------------------------------------------
 * n_steps iterations of recommender_abc.py:189-205 for the case in which nothing observes single steps (verbose off, no early-
 * stopping rule: no per-step loss, callback or log line): per step, wait for the drawn batch (draw-ahead workers above), queue
 * drx_cdae_step_dense on `stream`, submit the draws of later steps — at most 4 in flight and never beyond the last step, so the
 * sampler and corruption streams end exactly where the reference's do.  The Python loop does the same through _sample_batch /
 * _do_batch at ~40 us of interpreter time per step, which exceeds the device time of the ml-100k-shaped step.
 *   cursor[4] (in/out)  next sampler ticket, word of the corruption stream where the next batch begins, words consumed by
 *                       generator 0 and 1 of `draws` (what the Python loop keeps in _draw_ticket / _mask_pos / _mask_at)
 *   h_alphas            n_steps x 5 Keras-Adam lr_t, host memory (opt->alpha is ignored)
 *   h_slots             n_slots (8..64) staging slots of slot_bytes >= drx_cdae_fit_slot_bytes(B, keep_capacity) each, pinned host
 *                       memory addressable from the device (the step's kernels read the batch in place); keep_capacity >= the
 *                       largest possible sum of the B history lengths
 *   d_stage             device memory, d_stage_bytes >= 2 * slot_bytes, or NULL: the step of batch s copies what step s+1 reads of
 *                       its slot there from the last workgroup of its parameter sweep, so that only the first batch of a call is
 *                       read over PCIe by the gather kernel itself (NULL: every batch is)
 *   scratch             dense-step scratch (drx_cdae_scratch_bytes(p, B, 0, 1)) whose batch-membership arrays are zero
 *                       (DRX_DENSE_AUX_CLEAN is implied)
 * Returns after the queued steps have completed (the slots are the caller's again).  On error the draws in flight are waited for
 * and the cursor says what was consumed. */
size_t drx_cdae_fit_slot_bytes(int32_t B, int64_t keep_capacity);
int drx_cdae_fit_dense(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, DrxDrawAhead *draws, int64_t *cursor,
                       int32_t B, float q, int64_t keep_capacity, int32_t loss_kind, int32_t targets_kind, int64_t n_steps,
                       const float *h_alphas, void *h_slots, size_t slot_bytes, int32_t n_slots, void *d_stage,
                       size_t d_stage_bytes, void *scratch, size_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DRX_H_ */
