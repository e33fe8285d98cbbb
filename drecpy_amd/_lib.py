"""ctypes binding of libdrx.so (the C ABI declared in include/drx.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
PyTorch tensors are only the device-memory container — their `data_ptr()`s are what crosses the ABI.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdrx.so')

LOSS_BCE, LOSS_MSE = 0, 1
TARGETS_REFERENCE, TARGETS_PER_ROW = 0, 1
OPT_ADAM, OPT_ADAGRAD, OPT_ROWWISE_ADAGRAD = 0, 1, 2
DENSE_AUX_CLEAN = 0x100          # include/drx.h: DRX_DENSE_AUX_CLEAN
KEY_NONE = 0xFFFFFFFF
SHARD_SELF_BYPASS = 1            # include/drx.h: DRX_SHARD_SELF_BYPASS
BATCH_SHARE_USERS = 1            # include/drx.h: DRX_BATCH_SHARE_USERS


class DrxError(RuntimeError):
    pass


class CdaeParams(C.Structure):
    _fields_ = [('n_users', C.c_int32), ('n_items', C.c_int32), ('k', C.c_int32), ('ld', C.c_int32),
                ('W', C.c_void_p), ('W2T', C.c_void_p), ('V', C.c_void_p), ('b', C.c_void_p), ('b2', C.c_void_p)]


class History(C.Structure):
    _fields_ = [('indptr', C.c_void_p), ('indices', C.c_void_p), ('t_rank', C.c_void_p), ('t_nnz', C.c_int64), ('t_users', C.c_void_p),
                ('t_pos', C.c_void_p), ('t_items', C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [('B', C.c_int32), ('uid', C.c_void_p), ('iid', C.c_void_p), ('y', C.c_void_p),
                ('keep_off', C.c_void_p), ('keep', C.c_void_p), ('mask_seed', C.c_uint64), ('q', C.c_float),
                ('n_touch_slots', C.c_int32), ('flags', C.c_uint32)]


class ListGroups(C.Structure):
    _fields_ = [('indptr', C.c_void_p), ('seq_ids', C.c_void_p), ('held_indptr', C.c_void_p), ('held', C.c_void_p),
                ('group_value', C.c_void_p), ('eligible', C.c_void_p), ('n_groups', C.c_int32), ('n_eligible', C.c_int32),
                ('n_ids', C.c_int32)]


class Shard(C.Structure):
    _fields_ = [('world', C.c_int32), ('rank', C.c_int32), ('n_items', C.c_int32), ('items_per_rank', C.c_int32),
                ('n_users_local', C.c_int32), ('flags', C.c_uint32), ('chunks', C.c_int32)]


MAX_CHUNKS = 16                         # include/drx.h DRX_MAX_CHUNKS


class ShardExchange(C.Structure):
    """include/drx.h DrxShardExchange: one prepared batch's exchange state for the drx_shard_phase_* calls"""
    _fields_ = [('send_counts', C.c_void_p), ('recv_counts', C.c_void_p), ('uniq', C.c_void_p), ('req', C.c_void_p), ('table', C.c_void_p),
                ('table_bytes', C.c_size_t), ('rows_cache', C.c_void_p), ('rows_send', C.c_void_p), ('grad_send', C.c_void_p),
                ('grad_recv', C.c_void_p), ('rows_ticket', C.c_int64 * MAX_CHUNKS), ('grad_ticket', C.c_int64 * MAX_CHUNKS)]


MAX_SEGMENTS = 24                       # include/drx.h DRX_MAX_SEGMENTS


class AdamSegments(C.Structure):
    _fields_ = [('n', C.c_int32), ('start', C.c_int32 * MAX_SEGMENTS), ('len', C.c_int32 * MAX_SEGMENTS), ('alpha', C.c_float * MAX_SEGMENTS),
                ('l2_coef', C.c_float * MAX_SEGMENTS)]


class CaserDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('L', 'T', 'Tp', 'd', 'ld', 'ld2', 'n_v', 'n_h', 'n_small', 'off_kv', 'off_bv')] + \
               [('off_kh', C.c_int32 * 8), ('off_bh', C.c_int32 * 8), ('off_wd', C.c_int32), ('off_bd', C.c_int32), ('act_h', C.c_int32),
                ('act_mlp', C.c_int32)]


class CaserArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('item_emb', 'user_emb', 'W1', 'b1', 'sw', 'uid', 'before', 'after', 'keep')] + \
               [('rate', C.c_float), ('B', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('dE', 'dW1', 'db1', 'dPu', 'gsw_part', 'loss_part', 'cat_out')] + [('mask_seed', C.c_uint64)]


class CsrAdamTable(C.Structure):                        # include/drx.h DrxCsrAdamTable
    _fields_ = [(n, C.c_void_p) for n in ('row_ptr', 'order', 'src', 'scale')] + [(n, C.c_int32) for n in ('group', 'ld', 'n_rows')] + \
               [(n, C.c_void_p) for n in ('p', 'm', 'v', 'p_s', 'm_s', 'v_s')] + [(n, C.c_float) for n in ('alpha', 'alpha_s', 'l2_coef')]


class CsrList(C.Structure):                             # include/drx.h DrxCsrList
    _fields_ = [('keys', C.c_void_p), ('T', C.c_int32), ('n_rows', C.c_int32), ('row_ptr', C.c_void_p), ('order', C.c_void_p)]


class DmfDims(C.Structure):
    _fields_ = [('n_layers', C.c_int32 * 2), ('f', (C.c_int32 * 4) * 2), ('ld0', C.c_int32 * 2),
                ('off_k', (C.c_int32 * 4) * 2), ('off_b', (C.c_int32 * 4) * 2), ('n_small', C.c_int32),
                ('l2_norm_vectors', C.c_int32), ('off_scale', C.c_int32)]


class DmfArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('K0u', 'K0i', 'sw', 'u_indptr', 'u_indices', 'u_values', 'i_indptr', 'i_indices',
                                          'i_values', 'uid', 'iid', 'y')] + [('target_mode', C.c_int32), ('y_mean', C.c_float)] + \
               [(n, C.c_void_p) for n in ('off_u', 'off_i')] + [('B', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('dz0u', 'dz0i', 'tkeys_u', 'tsrc_u', 'tcoef_u', 'tkeys_i', 'tsrc_i', 'tcoef_i',
                                          'gsw_part', 'loss_part', 'pred_out', 'rep_u_out', 'rep_i_out', 'work', 'inv_u', 'inv_i',
                                          'gptr_u', 'gptr_i', 'grows_u', 'grows_i')] + [('n_du', C.c_int32), ('n_di', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('map_u', 'map_i', 'rho_u', 'rho_i')] + [('stamp', C.c_uint32)] + \
               [(n, C.c_void_p) for n in ('nd_dev', 'y_mean_dev', 'work_order')] + [('n_work', C.c_int32), ('seg_len', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('zseg', 'zpart', 'n_work_dev')]


class DmfK0Update(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('K0u', 'm_u', 'v_u', 'K0i', 'm_i', 'v_i')] + [('n_items', C.c_int32), ('n_users', C.c_int32)] + \
               [(n, C.c_float) for n in ('alpha_u', 'alpha_i', 'l2_coef', 'beta1', 'beta2', 'eps')] + [('row_order', C.c_void_p), ('n_long', C.c_int32)]


class Optim(C.Structure):
    _fields_ = [('kind', C.c_int32), ('lr', C.c_float), ('reg_rate', C.c_float), ('beta1', C.c_float),
                ('beta2', C.c_float), ('eps', C.c_float), ('alpha', C.c_float * 5),
                ('s1', C.c_void_p * 5), ('s2', C.c_void_p * 5)]


_lib = None

# name -> (restype, argtypes); every symbol declared in include/drx.h
SIGNATURES = {
    'drx_version': (C.c_int, []),
    'drx_strerror': (C.c_char_p, [C.c_int]),
    'drx_hash_u32': (C.c_uint32, [C.c_uint64, C.c_uint32, C.c_uint32]),
    'drx_event_create': (C.c_void_p, []),
    'drx_event_destroy': (None, [C.c_void_p]),
    'drx_event_record': (C.c_int, [C.c_void_p, C.c_void_p]),
    'drx_stream_wait_event': (C.c_int, [C.c_void_p, C.c_void_p]),
    'drx_event_synchronize': (C.c_int, [C.c_void_p]),
    'drx_stream_create_cu_slice': (C.c_void_p, [C.c_int32]),
    'drx_stream_destroy': (None, [C.c_void_p]),
    'drx_cdae_forward': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(History), C.POINTER(Batch), C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    'drx_cdae_scratch_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.c_int32, C.c_int32, C.c_int32]),
    'drx_cdae_step_dense': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(History), C.POINTER(Batch),
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    'drx_cdae_step_sparse': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(History), C.POINTER(Batch),
                                       C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    'drx_cdae_step_sparse_timed': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(History),
                                             C.POINTER(Batch), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.POINTER(C.c_void_p), C.c_void_p]),
    'drx_cdae_prep_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.c_int32, C.c_int32]),
    'drx_cdae_sparse_prepare': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(History), C.POINTER(Batch), C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    'drx_cdae_step_sparse_prepared': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(History),
                                                C.POINTER(Batch), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                                C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    'drx_point_sample_scratch_bytes': (C.c_size_t, [C.c_int32]),
    'drx_list_sample_device': (C.c_int, [C.POINTER(ListGroups), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'drx_point_sample': (C.c_int, [C.POINTER(History), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                                   C.c_void_p]),
    'drx_point_sample_recorded': (C.c_int, [C.POINTER(History), C.POINTER(History), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                                   C.c_void_p]),
    'drx_point_sample_by_user_scratch_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'drx_point_sample_by_user': (C.c_int, [C.POINTER(History), C.POINTER(History), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                                           C.c_void_p]),
    'drx_point_sample_valued': (C.c_int, [C.POINTER(History), C.POINTER(History), C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'drx_dmf_distinct_scratch_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'drx_dmf_batch_distinct_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 10 +
                                      [C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_shard_prep_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.POINTER(Shard), C.c_int32, C.c_int32]),
    'drx_shard_work_bytes': (C.c_size_t, [C.POINTER(Shard)]),
    'drx_shard_prep_layout': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Shard), C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    'drx_shard_prepare': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Shard), C.POINTER(History), C.POINTER(Batch), C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_shard_owner_table_bytes': (C.c_size_t, [C.POINTER(Shard), C.c_int32]),
    'drx_shard_chunks': (C.c_int32, [C.POINTER(Shard)]),
    'drx_shard_unit_shift': (C.c_int32, [C.POINTER(Shard)]),
    'drx_shard_owner_index': (C.c_int, [C.POINTER(Shard), C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_size_t, C.c_void_p]),
    'drx_shard_gather_rows': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Shard), C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                        C.c_int32, C.c_void_p, C.c_void_p]),
    'drx_shard_step_scratch_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.c_int32, C.c_int32]),
    'drx_shard_step_local': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(Shard), C.POINTER(History), C.POINTER(Batch),
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_void_p), C.c_void_p]),
    'drx_shard_apply': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(Shard), C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                  C.c_void_p, C.c_void_p]),
    'drx_shard_exchange_sizes': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Shard), C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    'drx_shard_phase_layout': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Shard), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    'drx_shard_phase_keys': (C.c_int, [C.POINTER(Shard), C.c_void_p, C.POINTER(ShardExchange), C.c_void_p]),
    'drx_shard_phase_rows': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Shard), C.c_void_p, C.POINTER(ShardExchange), C.c_int32, C.c_void_p]),
    'drx_shard_phase_local': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(Shard), C.POINTER(History), C.POINTER(Batch),
                                        C.c_void_p, C.POINTER(ShardExchange), C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]),
    'drx_shard_phase_tail': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(Shard), C.c_void_p, C.POINTER(ShardExchange),
                                       C.POINTER(ShardExchange), C.c_int32, C.c_void_p, C.c_void_p]),
    'drx_copy_f4': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_copy_f4_variant': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    'drx_comm_unique_id': (C.c_int, [C.c_void_p]),
    'drx_comm_create': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]),
    'drx_comm_destroy': (C.c_int, [C.c_void_p]),
    'drx_comm_stream': (C.c_void_p, [C.c_void_p]),
    'drx_comm_alltoallv': (C.c_int64, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), C.c_void_p]),
    'drx_comm_wait': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'drx_comm_last_error': (C.c_char_p, []),
    'drx_adam_dense': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_void_p]),
    'drx_scatter_scratch_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'drx_scatter_rows': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_batch_offsets': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_sumsq': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    'drx_rows_dot': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    'drx_adam_segments': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamSegments), C.c_float,
                                    C.c_float, C.c_float, C.c_void_p]),
    'drx_caser_grid': (C.c_int, [C.POINTER(CaserDims), C.c_int32]),
    'drx_caser_fwd_bwd': (C.c_int, [C.POINTER(CaserDims), C.POINTER(CaserArgs), C.c_void_p, C.c_void_p]),
    'drx_caser_step_small': (C.c_int, [C.POINTER(CaserDims), C.POINTER(CaserArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(AdamSegments), C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'drx_caser_hidden': (C.c_int, [C.POINTER(CaserDims), C.POINTER(CaserArgs), C.c_void_p]),
    'drx_rows_csr_adam_multi': (C.c_int, [C.POINTER(CsrAdamTable), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'drx_dmf_grid': (C.c_int, [C.c_int32]),
    'drx_dmf_work_bytes': (C.c_size_t, [C.c_int32]),
    'drx_dmf_fwd_bwd': (C.c_int, [C.POINTER(DmfDims), C.POINTER(DmfArgs), C.c_void_p, C.c_void_p]),
    'drx_dmf_step_small': (C.c_int, [C.POINTER(DmfDims), C.POINTER(DmfArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(AdamSegments), C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'drx_dmf_predict': (C.c_int, [C.POINTER(DmfDims), C.POINTER(DmfArgs), C.c_void_p]),
    'drx_first_occurrence': (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'drx_spin_until': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32]),
    'drx_batch_csr': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'drx_batch_csr_device_bytes': (C.c_size_t, [C.POINTER(CsrList), C.c_int32]),
    'drx_batch_csr_device': (C.c_int, [C.POINTER(CsrList), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_rows_csr_adam': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_void_p]),
    'drx_rows_csr_adam_outer': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'drx_dmf_work_order_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    'drx_dmf_work_order': (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.POINTER(C.c_int32)]),
    'drx_batch_distinct': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    'drx_dmf_norms': (C.c_int, [C.POINTER(DmfDims), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'drx_dmf_k0_update': (C.c_int, [C.POINTER(DmfDims), C.POINTER(DmfArgs), C.POINTER(DmfK0Update), C.c_void_p]),
    'drx_score_pairs_bf16': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_void_p]),
    'drx_topk_scratch_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'drx_topk_scratch_bytes_k': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'drx_topk': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_idmap_scratch_bytes': (C.c_size_t, [C.c_int64]),
    'drx_idmap_build': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    'drx_sampler_create': (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_double, C.c_int64]),
    'drx_sampler_sample': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'drx_cdae_reference_draw': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int64]),
    'drx_drawahead_create': (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    'drx_drawahead_submit': (C.c_int64, [C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    'drx_drawahead_wait': (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    'drx_drawahead_destroy': (None, [C.c_void_p]),
    'drx_cdae_fit_slot_bytes': (C.c_size_t, [C.c_int32, C.c_int64]),
    'drx_cdae_fit_dense': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(History), C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_float, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_sampler_draw': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'drx_sampler_destroy': (None, [C.c_void_p]),
    'drx_cdae_kshard_forward': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(History), C.POINTER(Batch), C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    'drx_cdae_kshard_forward_prepared': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(History), C.POINTER(Batch), C.c_void_p, C.c_size_t, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]),
    'drx_cdae_kshard_step': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Optim), C.POINTER(History), C.POINTER(Batch), C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    'drx_cdae_prep_result_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.c_int32, C.c_int32]),
    'drx_cdae_prep_part_out_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.c_int32, C.c_int32, C.c_int32]),
    'drx_cdae_prep_part_layout': (C.c_int, [C.POINTER(CdaeParams), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    'drx_cdae_prep_part_bytes': (C.c_size_t, [C.POINTER(CdaeParams), C.c_int32, C.c_int32, C.c_int32]),
    'drx_cdae_sparse_prepare_part': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(History), C.POINTER(Batch), C.c_int32, C.c_int32,
                                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_cdae_sparse_prepare_assemble': (C.c_int, [C.POINTER(CdaeParams), C.POINTER(Batch), C.c_void_p, C.c_int32, C.c_void_p,
                                                   C.c_size_t, C.c_void_p, C.c_void_p]),
    'drx_sort_pairs_temp_bytes': (C.c_size_t, [C.c_int64, C.c_int32]),
    'drx_sort_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    'drx_list_sampler_create': (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int64]),
    'drx_list_sampler_sample': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]),
    'drx_list_sampler_last_hint': (C.c_int32, [C.c_void_p]),
    'drx_list_sampler_destroy': (None, [C.c_void_p]),
    'drx_rng_create': (C.c_void_p, [C.c_int64]),
    'drx_rng_destroy': (None, [C.c_void_p]),
    'drx_rng_random': (C.c_double, [C.c_void_p]),
    'drx_rng_randint': (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64]),
    'drx_rng_discard': (None, [C.c_void_p, C.c_uint64]),
    'drx_rng_corruption_keep': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                          C.c_double, C.c_void_p, C.c_void_p, C.c_int64]),
}


def batch_offsets(indptr, ids):
    """int32 [B + 1] exclusive prefix of the CSR row lengths of the device ids (drx_batch_offsets; no torch arithmetic)."""
    import torch
    B = int(ids.numel())
    off = torch.empty(B + 1, dtype=torch.int32, device=ids.device)
    need = int(lib().drx_point_sample_scratch_bytes(B))
    sc = torch.empty(need, dtype=torch.uint8, device=ids.device)
    check(lib().drx_batch_offsets(ptr(indptr), ptr(ids), B, ptr(off), ptr(sc), need, stream_ptr(ids.device)), 'drx_batch_offsets')
    return off


def sumsq(tensors, device=None):
    """sum of squares of the given fp32 device tensors (contiguous), as a Python float — drx_sumsq, no torch arithmetic."""
    import torch
    ts = [t for t in tensors if t is not None and t.numel()]
    if not ts:
        return 0.0
    dev = ts[0].device
    out = torch.zeros(1 + 1024, dtype=torch.float64, device=dev)
    for i, t in enumerate(ts):
        t = t if t.is_contiguous() else t.contiguous()
        check(lib().drx_sumsq(ptr(t), t.numel(), ptr(out), 1 if i else 0, stream_ptr(dev)), 'drx_sumsq')
    return float(out[:1].cpu().numpy()[0])


def lib():
    """Loads libdrx.so once.  Raises DrxError when it has not been built (run `python -m drecpy_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DrxError(f'{LIB_PATH} not found: the HIP engine is not built (python -m drecpy_amd.build). '
                           'drecpy_amd has no CPU fallback.')
        # torch ships its own libamdhip64 (SONAME libamdhip64.so.7); it must be in the process BEFORE
        # libdrx.so so that both share ONE HIP runtime (otherwise torch's stream handles and device
        # pointers would belong to a different runtime instance than the one launching our kernels).
        host_only = os.environ.get('DRX_HOST_SANITIZER_LIB')
        if host_only:
            # CPU sanitizer runs (scripts/sanitize_host.sh): the host half alone, built by `python -m drecpy_amd.build
            # --sanitize=...`; device entry points are absent from it and stay unbound (calling one raises AttributeError)
            L = C.CDLL(host_only)
        else:
            import torch  # noqa: F401
            L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if host_only and not hasattr(L, name):
                continue
            fn = getattr(L, name)            # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.drx_version() != 100:
            raise DrxError('libdrx.so version mismatch')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().drx_strerror(rc).decode()
        raise DrxError(f'{what} failed: {msg} (code {rc})')


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream (0 = default stream).  torch.cuda.current_stream() builds a Stream object
    and resolves the device through several Python layers (6 us a call, ~18 calls per row-sharded step): the raw getter is the
    same lookup without them."""
    import torch
    try:
        idx = device.index if (device is not None and getattr(device, 'index', None) is not None) else torch._C._cuda_getDevice()
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))
    except AttributeError:
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
