/*
 * mi355x_rec.h — C ABI of libmi355x_rec.so, the MI355X (gfx950 / CDNA4) DeepFM / Wide&Deep
 * training hot path.
 *
 * The reference (leotimus/recommender-tensorflow) has no FFI of its own: every FLOP on the path
 * is executed by TensorFlow 1.12 ops that trainers/deep_fm.py:36-125 strings together.  Each entry
 * point below therefore names the reference *call site* whose arithmetic it replaces (file:line
 * relative to the reference root) and, where TensorFlow supplies the semantics implicitly, the
 * SURVEY.md appendix paragraph that restates them.
 *
 * Conventions
 *   - plain C, no exceptions; every entry returns 0 on success or a negative mi_status; the text
 *     of the last failure on the calling thread is mi_last_error().
 *   - the caller owns every buffer.  Device entries take device pointers (allocated by whoever
 *     owns the HIP context — PyTorch in the shipped host) and a hipStream_t passed as void*;
 *     they only enqueue work on that stream and never synchronise, allocate or free.
 *   - scratch space is an explicit caller-sized workspace: ask mi_*_workspace_bytes() first.
 *   - shapes are validated on the host before any launch (a bad shape returns
 *     MI_ERR_INVALID instead of faulting the GPU).
 *   - variables, activations, gradients and results are IEEE fp32, and so is every accumulation; index work is
 *     int32/int64, bit exact.  ONE family of entries feeds the matrix pipe something other than fp32 operands: the
 *     *_planes entries (and the other GEMM entries in gemm mode 1) split each fp32 operand into fp16 high + low parts
 *     (~22 significant bits per row-scaled operand; three exact 16-bit products per fp32 product, the lo*lo term
 *     dropped, fp32 accumulate) — fp32-LEVEL results (row-relative error < 1e-5, enforced in tests/test_hip_planes.py),
 *     not IEEE fp32 products.  mi_set_gemm_mode(0) runs the same GEMMs on the fp32-input MFMA (exact products).
 *   - PROCESS-WIDE STATE.  Two switches are per process, not per call or per model, because the deployment is one
 *     process per GPU running one model: mi_set_gemm_mode (the matrix-pipe path every GEMM entry takes) and
 *     mi_set_step_state (while set, the entries listed there read the step / lr_t / dropout seed term from a device
 *     record instead of from their arguments: hipGraph capture).  A host that drives TWO models from one process must
 *     set the GEMM mode before each model's calls (the shipped host does: engine._forward sets it every step) and
 *     must not capture one model's step while another thread launches the other's — the registered step state would
 *     be read by both.  Everything else (tables, slots, workspaces, streams) is per call.
 */
#ifndef MI355X_REC_H
#define MI355X_REC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mi_stream_t; /* hipStream_t */

enum mi_status {
  MI_OK = 0,
  MI_ERR_INVALID = -1,     /* bad argument / shape / alignment */
  MI_ERR_UNSUPPORTED = -2, /* legal but not built (e.g. embedding size not a multiple of 4 > 256) */
  MI_ERR_LAUNCH = -3,      /* HIP reported an error at launch */
  MI_ERR_WORKSPACE = -4    /* workspace too small */
};

/* ---- library ---------------------------------------------------------------------------- */
int32_t mi_abi_version(void);       /* bumps when a signature changes (currently 21) */
const char* mi_last_error(void);    /* thread-local, never NULL */
const char* mi_build_info(void);    /* "gfx950 hipcc <ver>" */

/* ---- replayable steps (hipGraph) -------------------------------------------------------------------------
 * A captured train step must not carry per-step values in its launch arguments.  While a device-resident step
 * state is registered (process-wide, like the GEMM mode: one process per GPU), the entries below read them from
 * it AT RUN TIME instead of from their scalar arguments:
 *   mi_sparse_apply[_fused]              step <- state->step, hp->lr_t <- state->lr_t
 *   mi_dense_apply                       hp->lr_t <- state->lr_t
 *   mi_sparse_catchup, mi_catchup_gap_keys, mi_catchup_rows_by_gap   step_to <- state->step - 1
 *   mi_dense_fwd[_gathered|_planes], mi_hidden_logits_head_fused   seed <- seed + state->seed_term   (dropout masks differ every step)
 * mi_step_advance (the first node of the captured step): step += 1, lr_t = lr_table[step], seed_term = step * 1000003.
 * Register the state only while capturing; eager calls between replays keep using their arguments. */
typedef struct mi_step_state {
  int32_t step;        /* the step being executed (1-based, as the global step after it) */
  float lr_t;          /* Adam: lr * sqrt(1 - beta2^step) / (1 - beta1^step) */
  uint64_t seed_term;
} mi_step_state_t;
int32_t mi_set_step_state(const mi_step_state_t* device_state);   /* NULL: off */
int32_t mi_step_advance(mi_step_state_t* device_state, const float* lr_table, mi_stream_t stream);

/* ---- (a1) categorical id transforms, host side --------------------------------------------
 * replaces the column constructors of trainers/ml_100k.py:19-35.
 * All integer work; results are bit exact against oracle/fingerprint.py. */

/* FarmHash Fingerprint64 (farmhashna::Hash64) of len bytes — what
 * tf.feature_column.categorical_column_with_hash_bucket applies (ml_100k.py:19,20,29,30). */
uint64_t mi_fingerprint64(const void* data, size_t len);

/* CRC-32C (Castagnoli) of len bytes, continuing from `crc` (0 to start; unmasked) — the checksum TensorFlow's tensor-bundle
 * checkpoints carry per table block and per tensor (the files tf.estimator writes under conf_utils.py:6-10's RunConfig;
 * read and written by mi355x_rec/tf_bundle.py).  Host function. */
uint32_t mi_crc32c(const void* data, size_t len, uint32_t crc);

/* id = Fingerprint64(decimal ASCII of v) mod num_buckets — hash_bucket column with an integer
 * dtype (user_id, item_id: ml_100k.py:19-20). */
int32_t mi_hash_bucket_i64(const int64_t* values, int64_t n, int64_t num_buckets, int32_t* out_ids);

/* id = Fingerprint64(bytes[offsets[i]:offsets[i+1]]) mod num_buckets — string hash_bucket column
 * (occupation, zipcode: ml_100k.py:29-30).  offsets has n+1 entries. */
int32_t mi_hash_bucket_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n,
                             int64_t num_buckets, int32_t* out_ids);

/* id = number of boundaries <= x (upper_bound) — bucketized_column (ml_100k.py:23-24,33-34). */
int32_t mi_bucketize_f32(const float* values, int64_t n, const float* boundaries,
                         int32_t num_boundaries, int32_t* out_ids);

/* ---- (a2,a3,a5) embedding gather + wide linear reduce + FM second order, forward ------------
 * replaces tf.feature_column.linear_model (deep_fm.py:39), embedding_column + input_layer
 * (deep_fm.py:52-54) and the mf block (deep_fm.py:79-87) with ONE kernel.
 *
 *   row(b,f)      = field_off[f] + ids[b*F+f]                 (fused row-major table, all fields)
 *   concat[b,f,:] = table[row(b,f), :]                        (deep_fm.py:54; mean of one id = row)
 *   sumv[b,:]     = sum_f concat[b,f,:]                       (saved for the backward)
 *   fm[b]         = 0.5 * sum_e( sumv[b,e]^2 - sum_f concat[b,f,e]^2 )   (deep_fm.py:81-87)
 *   lin[b]        = sum_f lin_w[row(b,f)]                     (deep_fm.py:39; bias is added by the head)
 *
 * table [R,E] f32, lin_w [R] f32 (may be NULL: lin not produced), field_off [F] int64 (device),
 * ids [B,F] int32 (device).  Outputs: concat [B,F*E] with leading dimension ld_concat (floats,
 * >= F*E: numeric-embedding columns may follow, deep_fm.py:73), sumv [B,E], fm [B], lin [B]
 * (any of them may be NULL; amax_rows, if not NULL, is an abs-max vector — see mi_gemm_amax_t — that
 * receives max |table[row(b,f), :]| over the rows read, for the MLP's layer-1 GEMMs; with concat == NULL the kernel only READS rows — the form the
 * single-GPU path uses, where layer 1 gathers its operand itself, and the one bench.py prices
 * against the HBM read roofline; with table == NULL only the wide part lin is produced).  E must be a multiple of 4 and <= 256.  Fields must already be
 * in the reference's sorted-by-column-name order.
 * lin_stride: element stride of the wide part's per-row state, here and in mi_gather_rows / mi_sparse_apply* /
 * mi_sparse_catchup / mi_catchup_gap_keys — 1: lin_w, l_slot0, l_slot1 and last_step are four separate arrays;
 * 4: one 16-byte record per row {weight, slot0, slot1, stamp} (lin_w = rec, l_slot0 = rec + 1, l_slot1 = rec + 2,
 * last_step = (int32_t*)rec + 3): a row's wide-part state then costs one memory sector instead of four
 * (round 1's catchup_lin_k fetched 569 MB for 20 MB of state).
 * table_stride (round 4), here and in every entry that takes the embedding table (mi_embed_fm_planes_fwd, mi_gather_rows,
 * mi_dense_fwd_gathered, mi_dense_bwd_weight_gathered, mi_sparse_apply[_fused], mi_sparse_catchup): floats between
 * consecutive rows of table (and of its slot arrays) — 0 or E: [R, E] arrays; 3 E with t_slot0 = table + E, t_slot1 = table +
 * 2 E: ONE record [w | slot0 | slot1] per row (what the shipped host allocates: a row's weights and optimizer state are one
 * contiguous 12 E-byte run — one DRAM page visit per row and direction in the sparse apply and the catch-up instead of
 * three).  A multiple of 4, >= E.
 * The PACKED EXCHANGE of the row-sharded step (round 4): a request's row and wide weight travel as ONE record of E + 4 floats
 * [row | weight | pad x 3] (and its gradients likewise), so a chunk costs one collective per direction instead of two —
 * mi_gather_rows(out_stride = E + 4, out_lin = out_rows + E) writes such records, the forward entries read them with
 * table_stride = lin_stride = E + 4, mi_entry_grads_segsum(rows_stride, out_stride) reads and writes them,
 * mi_sparse_apply(grad_stride = E + 4, d_lin = d_rows + E) applies them.  0 everywhere = separate arrays, as before. */
int32_t mi_embed_fm_linear_fwd(const float* table, const float* lin_w, const int64_t* field_off,
                               const int32_t* ids, int64_t B, int32_t F, int32_t E,
                               float* concat, int64_t ld_concat, float* sumv, float* fm, float* lin,
                               float* amax_rows, int32_t lin_stride, int64_t table_stride, mi_stream_t stream);

/* The gather with the concat written as fp16 high/low planes (struct mi_planes, below): the operand of
 * the layer-1 GEMMs mi_dense_fwd_planes / mi_dense_bwd_weight_planes.  One exponent per example, from the
 * example's abs-max over all F*E values (the lane group that gathers an example holds all its rows in
 * registers).  Also sumv / fm as above and amax_rows.  E a multiple of 16 and >= 32, F <= 48; the wide part
 * is mi_embed_fm_linear_fwd(table = NULL).
 * x_num [B, n_numeric], tail_cols (0: none): the canned estimators' raw numeric columns (linear_deep.py:32-39 with
 * numeric_column()s in dnn_feature_columns) — the values themselves follow the embedding columns as columns
 * F*E .. F*E + tail_cols of the concat (n_numeric values, then zeros; tail_cols a multiple of 16, at most 4 E), under the
 * example's one exponent: the fp32 concat, its abs-max pass and its split into planes are never made. */
struct mi_planes;
int32_t mi_embed_fm_planes_fwd(const float* table, const int64_t* field_off, const int32_t* ids, int64_t B, int32_t F,
                               int32_t E, float* sumv, float* fm, const struct mi_planes* concat, float* amax_rows,
                               const float* x_num, int32_t n_numeric, int32_t tail_cols, int64_t table_stride, mi_stream_t stream);

/* Owner-side half of the row-sharded path (multi-GPU): out_rows[i,:] = table[rows[i],:],
 * out_lin[i] = lin_w[rows[i] * lin_stride].  rows [n] int32 local row ids; table (with out_rows) or lin_w
 * (with out_lin) may be NULL. */
int32_t mi_gather_rows(const float* table, const float* lin_w, const int32_t* rows, int64_t n,
                       int32_t E, float* out_rows, float* out_lin, int32_t lin_stride, int64_t table_stride, int64_t out_stride, mi_stream_t stream);

/* (a4) numeric embedding, deep_fm.py:62-70: out[b, j*E+e] = x[b,j] * V[j,e], written at
 * concat[b, col0 + j*E + e]; also accumulates into sumv / fm so the FM term sees the numeric
 * fields (deep_fm.py:79 reshapes the concatenated layer), and lin[b] += sum_j x[b,j]*w_num[j]. */
int32_t mi_numeric_embed_fwd(const float* x, const float* V, const float* w_num, int64_t B,
                             int32_t n_d, int32_t E, float* concat, int64_t ld_concat, int64_t col0,
                             float* sumv, float* fm, float* lin, mi_stream_t stream);

/* Numeric columns of the canned estimators (tf.feature_column.numeric_column handed to
 * DNNLinearCombinedClassifier, trainers/linear_deep.py:32-39): input_layer takes
 * the VALUE as a column of the concat, linear_model multiplies it by a [1,1] weight.
 *   concat[b, col0 + j] = x[b,j] for j < n_d, 0 for n_d <= j < n_cols   (zero pad: whole k-tiles for layer 1)
 *   lin[b] += sum_j x[b,j] * w_num[j]        (ascending j, after the categorical sum; bias added by the head)
 * concat or lin may be NULL.  Backward: dw_num[j] = sum_b d_logit_lin[b] * x[b,j] (two-stage, fixed order). */
int32_t mi_numeric_raw_fwd(const float* x, const float* w_num, int64_t B, int32_t n_d, float* concat,
                           int64_t ld_concat, int64_t col0, int32_t n_cols, float* lin, mi_stream_t stream);
size_t mi_numeric_raw_bwd_workspace_bytes(int64_t B, int32_t n_d);
int32_t mi_numeric_raw_bwd(const float* x, const float* d_logit_lin, int64_t B, int32_t n_d, float* dw_num,
                           void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* ---- (a2,a3,a5) backward: per-entry row gradients -------------------------------------------
 * d_rows[p(b,f), :] = d_concat[b,f,:] + d_logit_fm[b] * (sumv[b,:] - concat[b,f,:])
 * d_lin [p(b,f)]    = d_logit_lin[b]
 * (gradient of deep_fm.py:54,81-87,39 w.r.t. the gathered rows; TF: IndexedSlices values).
 * pos [B*F] int32 gives the destination slot p(b,f) (NULL = identity b*F+f); the sharded path
 * uses it to write straight into all-to-all send order.  d_concat may be NULL (no DNN),
 * d_logit_fm may be NULL (no FM), d_lin/d_logit_lin may be NULL (no linear part).  Instead of concat
 * the gathered rows may be given in slot order (rows [n,E], row of (b,f) at slot pos[b*F+f]: the
 * receive buffer of the row exchange, when the concat was never materialised). */
int32_t mi_embed_fm_linear_bwd(const float* d_concat, int64_t ld_dconcat, const float* concat,
                               int64_t ld_concat, const float* rows, const float* sumv, const float* d_logit_fm,
                               const float* d_logit_lin, const int32_t* pos, int64_t B, int32_t F,
                               int32_t E, float* d_rows, float* d_lin, mi_stream_t stream);

/* Requester-side half of the sparse "reduce-scatter" (multi-GPU): the entries of a chunk that asked for the
 * same row are summed BEFORE the gradient all-to-all, so one gradient row per distinct request travels.
 * For distinct requests u in [u_begin, u_begin + u_count) (segments of mi_sort_unique_rows over the request
 * keys): out_rows[u,:] = sum over the segment's entries e = (b, f), in ascending entry order, of
 *   d_concat[b - b0, f*E:(f+1)*E] + d_logit_fm[b - b0] * (sumv[b - b0,:] - rows[u,:])
 * and out_lin[u] = sum of d_logit_lin[b - b0]; rows [*,E] = the exchange's receive buffer (slot u), the
 * per-example arrays belong to examples b0.. (one chunk).  Segments longer than 48 entries are summed by a
 * workgroup in fixed slices, like mi_sparse_apply's.
 * out_row0 (0 <= out_row0 <= u_begin): the request that row 0 of out_rows / out_lin holds — request u is written at
 * row u - out_row0.  0: the out buffers are the whole send buffer; u_begin: they start at this range (a rank's
 * requests to ITSELF are summed straight into the buffer its own sparse apply reads, no exchange in between). */
int32_t mi_entry_grads_segsum(const float* rows, const int32_t* seg_start, const int32_t* sorted_entry, int64_t u_begin,
                              int64_t u_count, const float* d_concat, int64_t ld_dconcat, const float* sumv,
                              const float* d_logit_fm, const float* d_logit_lin, int64_t b0, int32_t F, int32_t E,
                              float* out_rows, float* out_lin, int64_t out_row0, int64_t rows_stride, int64_t out_stride, mi_stream_t stream);

/* (a4) backward of the numeric embedding: dV[j,e] = sum_b x[b,j]*g[b,j,e] with
 * g = d_concat + d_logit_fm*(sumv - concat);  dw_num[j] = sum_b d_logit_lin[b]*x[b,j].
 * Deterministic two-stage reduction; workspace from mi_numeric_embed_bwd_workspace_bytes. */
size_t mi_numeric_embed_bwd_workspace_bytes(int64_t B, int32_t n_d, int32_t E);
int32_t mi_numeric_embed_bwd(const float* x, const float* d_concat, int64_t ld_dconcat,
                             const float* concat, int64_t ld_concat, int64_t col0, const float* sumv,
                             const float* d_logit_fm, const float* d_logit_lin, int64_t B,
                             int32_t n_d, int32_t E, float* dV, float* dw_num, void* workspace,
                             size_t workspace_bytes, mi_stream_t stream);

/* ---- row bookkeeping for sparse updates -----------------------------------------------------
 * Sorts the n requested rows (stable LSD radix sort, key = row, payload = entry index) and
 * compacts duplicates — the device analogue of TF's unique + unsorted_segment_sum that
 * precedes every sparse optimizer apply.  Outputs (all device):
 *   sorted_entry [n]   entry index (into rows / d_rows) in ascending (row, entry) order
 *   uniq_rows    [n]   first U slots hold the distinct rows in ascending order
 *   seg_start    [n+1] segment u covers sorted_entry[seg_start[u] .. seg_start[u+1])
 *   num_uniq     [1]   U (int32) — stays on the device; later kernels read it there
 * num_rows_total bounds the key range (picks the number of radix passes).
 * With uniq_rows == NULL only the stable sort permutation is produced (sorted_entry); seg_start and
 * num_uniq are then ignored. */
size_t mi_sort_unique_workspace_bytes(int64_t n);
int32_t mi_sort_unique_rows(const int32_t* rows, int64_t n, int64_t num_rows_total,
                            int32_t* sorted_entry, int32_t* uniq_rows, int32_t* seg_start,
                            int32_t* num_uniq, void* workspace, size_t workspace_bytes,
                            mi_stream_t stream);

/* mi_sort_unique_rows + mi_segment_slots in one entry (round 5): additionally slot_of_entry [n] = the segment u of every entry
 * (the row-sharded step's "slot of my row in the receive buffer"), written by the compaction as it goes — the separate
 * binary-search entry costs 112 us at 1.7 M entries. */
int32_t mi_sort_unique_rows_slots(const int32_t* rows, int64_t n, int64_t num_rows_total,
                                  int32_t* sorted_entry, int32_t* uniq_rows, int32_t* seg_start,
                                  int32_t* num_uniq, int32_t* slot_of_entry, void* workspace, size_t workspace_bytes,
                                  mi_stream_t stream);

/* mi_global_rows + mi_sort_unique_rows in one entry for the single-GPU step: ids [B][F] (local ids, field f's rows are
 * field_off[f] + id: disjoint ranges in ascending field order), max_vocab >= every field's id range.  Same outputs,
 * bit for bit (equal rows keep ascending example order): sorted_entry [B*F] (entry = b * F + f), uniq_rows (global
 * rows), seg_start, num_uniq.  Every field's ids are sorted on their own, so the radix passes cover the bits of
 * max_vocab, not of the whole table.  B a multiple of 4096, F <= 64 (anything else: the two-entry form).
 * workspace: mi_sort_unique_fields_workspace_bytes(B, F), 256-byte aligned.
 * beside (round 5, ABI 21): how the work is cut into launches — same results either way.  0: the stream has the GPU to itself
 * (a step that was not told its next batch sorts at its head): ONE launch per radix pass and one for the compaction, the
 * tiles of a field exchanging their digit counts inside the launch (5 launches, 130 us at config 3 against 145).  1: the
 * call runs on a side stream BESIDE the step's bandwidth-bound kernels (the next batch's sort, beside this batch's Adam
 * catch-up): the round-1 form, 14 short launches that hold no resources while they wait — beside the catch-up the fused
 * form's resident, waiting workgroups cost the step 0.03 ms more than they save (profiles/r05_sort_and_side_streams.md). */
size_t mi_sort_unique_fields_workspace_bytes(int64_t B, int32_t F);
int32_t mi_sort_unique_fields(const int32_t* ids, const int64_t* field_off, int64_t B, int32_t F, int64_t max_vocab,
                              int32_t* sorted_entry, int32_t* uniq_rows, int32_t* seg_start, int32_t* num_uniq,
                              void* workspace, size_t workspace_bytes, int32_t beside, mi_stream_t stream);

/* Routing helpers of the row-sharded multi-GPU path (row r lives on rank r % world as local row
 * r / world; the reference's own multi-worker mode is TF's parameter-server placement of whole
 * variables, distributed.md:58-82 — see DESIGN.md "Multi-GPU").
 *   mi_shard_keys     : keys[i] = ((i / entries_per_chunk) * world + pos(rows[i] % world)) * rows_per_rank + rows[i] / world
 *                       (chunk-major request key of a pipelined step; entries_per_chunk == 0: one chunk).
 *                       pos: the owner's place in the send order — the rank itself (self_rank = -1), or, with
 *                       self_rank >= 0, the other ranks in rank order and the asking rank LAST (o -> o below
 *                       self_rank, o - 1 above, world - 1 for self_rank): a rank's requests to itself then end every
 *                       chunk's run, outside the part an all-to-all with a zero self split ships.
 *                       mi_sort_unique_rows of the keys orders the entries by (chunk, owner, row); its unique
 *                       keys are the DISTINCT rows each chunk needs from each owner — what travels.
 *   mi_route_requests : per distinct request u < *num_uniq: send_rows[u] = owner-local row;
 *                       counts[chunk * world + owner] = number of distinct requests of that group (device;
 *                       zeroed here) — the all-to-all split sizes
 *   mi_segment_slots  : slot_of_entry[e] = u, the distinct request entry e belongs to (the received row's
 *                       slot in the exchange buffer)
 *   mi_gather_u32     : out[i] = src[idx[i]] for 4-byte elements (int32 ids or f32 values) */
int32_t mi_shard_keys(const int32_t* rows, int64_t n, int32_t world, int64_t entries_per_chunk, int64_t rows_per_rank,
                      int32_t self_rank, int32_t* keys, mi_stream_t stream);
int32_t mi_route_requests(const int32_t* uniq_keys, const int32_t* num_uniq, int64_t n_max, int64_t rows_per_rank,
                          int32_t n_groups, int32_t* send_rows, int32_t* counts, mi_stream_t stream);
int32_t mi_segment_slots(const int32_t* seg_start, const int32_t* sorted_entry, const int32_t* num_uniq, int64_t n,
                         int32_t* slot_of_entry, mi_stream_t stream);
int32_t mi_gather_u32(const void* src, const int32_t* idx, int64_t n, void* out, mi_stream_t stream);

/* rows[b*F+f] = field_off[f] + ids[b*F+f]  (int32 global row per entry) */
int32_t mi_global_rows(const int32_t* ids, const int64_t* field_off, int64_t B, int32_t F,
                       int32_t* rows, mi_stream_t stream);

/* ---- (a9) optimizers ------------------------------------------------------------------------
 * get_optimizer (model_utils.py:57-66) -> tf.train.{Adam,Adagrad,Ftrl,RMSProp,GradientDescent}
 * Optimizer(learning_rate), applied by head.create_estimator_spec (deep_fm.py:119-125).
 * SURVEY Appendix A.6/A.7 restate the TF-1.12 update rules the kernels follow op for op
 * (compiled with -ffp-contract=off so the fp32 sequence equals the oracle's). */

enum mi_optimizer {
  MI_OPT_ADAM = 0,     /* tf.train.AdamOptimizer: beta1 .9 beta2 .999 eps 1e-8 */
  MI_OPT_ADAGRAD = 1,  /* tf.train.AdagradOptimizer: initial_accumulator_value .1 */
  MI_OPT_FTRL = 2,     /* tf.train.FtrlOptimizer: lr_power -.5, init accum .1, l1=l2=0 */
  MI_OPT_RMSPROP = 3,  /* tf.train.RMSPropOptimizer: decay .9 momentum 0 eps 1e-10 */
  MI_OPT_SGD = 4       /* tf.train.GradientDescentOptimizer */
};

typedef struct mi_opt_hparams {
  int32_t kind;      /* enum mi_optimizer */
  float lr;          /* learning_rate as given to the TF constructor */
  float beta1, beta2, epsilon;          /* Adam */
  float lr_t;        /* Adam: lr*sqrt(1-beta2^t)/(1-beta1^t) for THIS step, computed by the host in fp32 */
  float decay, momentum;                /* RMSProp */
  float lr_power, l1, l2;               /* Ftrl */
} mi_opt_hparams;

/* Dense apply over a flat parameter buffer (all MLP kernels/biases, numeric embeddings, linear
 * bias live in one buffer).  slot0/slot1: Adam m,v | Adagrad accum,- | Ftrl accum,linear |
 * RMSProp ms,mom | SGD -,-.  Fused ApplyAdam / ApplyAdagrad / ApplyFtrl / ApplyRMSProp /
 * ApplyGradientDescent semantics. */
/* y[i] += alpha * x[i]: dense-gradient accumulation over the chunks of a pipelined multi-GPU step */
int32_t mi_axpy(float* y, const float* x, int64_t n, float alpha, mi_stream_t stream);

int32_t mi_dense_apply(float* param, float* slot0, float* slot1, const float* grad, int64_t n,
                       const mi_opt_hparams* hp, mi_stream_t stream);

/* Sparse apply on the U distinct rows found by mi_sort_unique_rows: segment-sums the entry
 * gradients in ascending entry order (== TF's unsorted_segment_sum order on CPU), then applies
 * the optimizer's sparse rule to table row r (width E) and, when lin_w != NULL, to lin_w[r]
 * (width 1) with d_lin.  For Adam the rule is TF's dense-equivalent one: see mi_sparse_catchup.
 * last_step[r] is set to step.  slot arrays mirror the parameter arrays' shapes. */
int32_t mi_sparse_apply(float* table, float* t_slot0, float* t_slot1, float* lin_w, float* l_slot0,
                        float* l_slot1, int32_t* last_step, const int32_t* uniq_rows,
                        const int32_t* seg_start, const int32_t* sorted_entry,
                        const int32_t* num_uniq, int64_t n_max, const float* d_rows,
                        const float* d_lin, int32_t E, int32_t step, const mi_opt_hparams* hp,
                        int32_t lin_stride, int64_t table_stride, int64_t grad_stride, mi_stream_t stream);

/* Single-GPU form of mi_sparse_apply with mi_embed_fm_linear_bwd folded in: the gradient of entry
 * e = (b, f) = (e / F, e % F) is rebuilt inside the kernel as
 *   d_concat[b, f*E:(f+1)*E] + d_logit_fm[b] * (sumv[b,:] - w)     (w = the row itself, not yet updated;
 *   equals the concat slice the forward used)  and  d_logit_lin[b] for the linear weight,
 * so the [B*F, E] per-entry gradient matrix is never written.  Same summation order, same bits.
 * (sumv == NULL with d_logit_fm set: d_concat already carries d_logit_fm * sumv; the kernel subtracts d_logit_fm * row only.) */
int32_t mi_sparse_apply_fused(float* table, float* t_slot0, float* t_slot1, float* lin_w, float* l_slot0,
                              float* l_slot1, int32_t* last_step, const int32_t* uniq_rows,
                              const int32_t* seg_start, const int32_t* sorted_entry, const int32_t* num_uniq,
                              int64_t n_max, const float* d_concat, int64_t ld_dconcat, const float* sumv,
                              const float* d_logit_fm, const float* d_logit_lin, int32_t F, int32_t E,
                              int32_t step, const mi_opt_hparams* hp, int32_t lin_stride, int64_t table_stride, mi_stream_t stream);

/* TF-1.12 AdamOptimizer._apply_sparse decays m and v of EVERY row and moves EVERY row each step
 * (SURVEY Appendix A.6).  Instead of sweeping the table, rows carry last_step[r] and are brought
 * up to date lazily: for s in (last_step[r], step_to]: m*=b1; v*=b2; w -= lr_t[s]*m/(sqrt(v)+eps)
 * — the same fp32 op sequence as the sweep, hence the same bits.  lr_table[s] (device, f32) holds
 * lr_t of step s.  Runs on the U distinct rows about to be gathered (step_to = step-1), or on all
 * rows (uniq_rows == NULL, n_max = R) before evaluation / checkpoint.
 * flags: any of MI_CATCHUP_DEFER_SLOTS | MI_CATCHUP_BOUNDED | MI_CATCHUP_KEEP_STAMPS (0 = none).
 * MI_CATCHUP_DEFER_SLOTS (with uniq_rows): only w is written; m, v and last_step keep their old values and
 * the mi_sparse_apply[_fused] of the same step — which reads and writes m, v anyway and MUST then be
 * given last_step — decays them from the old stamp (same multiply chain, same bits).  Saves a third
 * of this kernel's HBM traffic; every row passed here must be applied in the same step.
 * MI_CATCHUP_BOUNDED: the replay with a stated error bound instead of TF's bits (the parity bar is 1e-5 on the
 * logits, not bit equality): m_j and lr_t[s] * m_j are still the reference's chain, bit for bit; sqrt(v_j) is taken as
 * sqrtf(v_0) * beta2^(j/2) (a per-row scalar chain with a two-float multiplier) and the division as a reciprocal good to
 * 1 ulp (v_rcp_f32 for the wide part; for the rows one carried from step to step and corrected against each denominator),
 * w is rounded once per step like the reference's.  Every replayed update is within a few 2^-24 (relative) of the
 * reference's, i.e. ~1e-10 |w|.  Enforced against the literal sweep after 150-200 replayed steps, for EVERY variable:
 * |w - w_sweep| <= 3 ulp(w) + 2e-6 * sum_j |t_j| (t_j: the replayed updates) — and in distribution >= 95 % of the variables
 * bit-identical, >= 98 % within 1e-7 relative (measured 96.7 % / 98.7 %; a 1-ulp difference alone is up to 1.19e-7, so
 * "1e-7 for every variable" is NOT claimed): tests/test_hip_kernels.py::test_bounded_catchup_stays_within_its_bound_of_the_sweep.  m, v and the
 * stamps are written exactly as in the exact mode.  7 packed VALU operations per element and step (no transcendental)
 * instead of 16 + 2, and no range conditions.  Needs epsilon >= 1e-30 (otherwise the exact form runs).
 * MI_CATCHUP_KEEP_STAMPS: the rows' stamps are left as they are (m, v ARE written) — for a model whose tables and wide part
 * follow two different Adam optimizers (two lr_t tables: two calls; the first must not move the stamps the second reads). */
enum mi_catchup_flags { MI_CATCHUP_DEFER_SLOTS = 1, MI_CATCHUP_BOUNDED = 2, MI_CATCHUP_KEEP_STAMPS = 4 };
/* keys[u] = how many steps row uniq_rows[u] will be replayed over by mi_sparse_catchup(step_to) (0..62,
 * clamped), 63 for the slots u >= *num_uniq.  Sorting the rows by it (mi_sort_unique_rows with
 * key_range 64, then mi_gather_u32 of uniq_rows through the permutation) groups rows of equal
 * staleness, which halves the catch-up's divergence; the valid rows stay in front. */
int32_t mi_catchup_gap_keys(const int32_t* uniq_rows, const int32_t* num_uniq, const int32_t* last_step, int64_t n_max,
                            int32_t step_to, int32_t* keys, int32_t lin_stride, mi_stream_t stream);

/* The three steps above in one entry: rows_out[0 .. *num_uniq) = uniq_rows stably sorted by that key (slots past
 * *num_uniq: unspecified), with the key computed inside the sort's histogram kernel and the row ids carried as the
 * sort's values — two launches fewer than mi_catchup_gap_keys + mi_sort_unique_rows + mi_gather_u32.
 * workspace: mi_sort_unique_workspace_bytes(n_max), 256-byte aligned. */
int32_t mi_catchup_rows_by_gap(const int32_t* uniq_rows, const int32_t* num_uniq, const int32_t* last_step, int64_t n_max,
                               int32_t step_to, int32_t lin_stride, int32_t* rows_out, void* workspace,
                               size_t workspace_bytes, mi_stream_t stream);


/* Self-test of the catch-up's fast square root (no reference analogue; the exactness claim above rests on it): for the
 * `count` fp32 bit patterns from first_bits on, mismatches[0] += how many give different bits in the replay loop's
 * in-range sqrt (v_rsq_f32 + coupled Newton step + residual correction) than in hipcc's correctly rounded sqrtf,
 * mismatches[1] += the same for the v_sqrt_f32 + one-ulp-test form.  mismatches: 2 x uint64 on the device, zeroed by the
 * caller.  The fast loop is only entered with v in [2^-93, 2^20]; the test sweeps [2^-100, 2^24] and demands 0. */
int32_t mi_selftest_sqrt(uint32_t first_bits, int64_t count, uint64_t* mismatches, mi_stream_t stream);

/* Self-test of the GEMM epilogues' division by a host-known constant (tf.nn.dropout divides by keep_prob: deep_fm.py:102-103;
 * no reference analogue: the claim that the 3-instruction form q = x r; e = fma(-d, q, x); fma(e, r, q) with r = RN(1 / d)
 * returns the bits of x / d rests on it): for the `count` fp32 bit patterns x from first_bits on, out[0] += how many with
 * 2^-100 <= |x| <= 2^100 give different bits than '/', out[1] += how many of all of them do (below 2^-100 the residual of a
 * quotient is not exactly representable any more, above 2^100 x r can overflow where x / d does not, and an infinite x gives
 * inf - inf).  out: 2 x uint64 on the device, zeroed by the caller.  The test sweeps all 2^32 patterns for a set of dropout
 * rates and demands out[0] == 0. */
int32_t mi_selftest_div(float d, uint32_t first_bits, int64_t count, uint64_t* out, mi_stream_t stream);

int32_t mi_sparse_catchup(float* table, float* t_m, float* t_v, float* lin_w, float* l_m, float* l_v,
                          int32_t* last_step, const int32_t* uniq_rows, const int32_t* num_uniq,
                          int64_t n_max, int32_t E, int32_t step_to, const float* lr_table,
                          float beta1, float beta2, float epsilon, int32_t flags, int32_t lin_stride,
                          int64_t table_stride, mi_stream_t stream);

/* ---- (a6) the [hidden_units] MLP: fp32 GEMMs on the matrix cores with fused epilogues -----------
 * replaces tf.layers.dense / tf.layers.dropout (deep_fm.py:98-108).  Row-major everywhere.
 * fp32 in, fp32 accumulate, fp32 out; error at the level of an fp32 GEMM (the 1e-5 logit bar). */

/* Matrix-pipe path of the GEMMs below (process-wide switch):
 *   1 (default) 16-bit operand split.  On CDNA4 the fp32-input MFMA runs at 1/16 of the f16/bf16
 *      rate; a product of two 16-bit floats is exact in fp32 and the MFMA accumulates in fp32.
 *      - "f16x2", mi_dense_bwd_weight[_gathered] only, when the call carries both operands' abs-max
 *        (mi_gemm_amax_t): each operand is scaled by ONE power of two (largest magnitude -> [2^14, 2^15))
 *        and split into fp16 high + low parts; three products per k-step (dropped terms <= 2^-21 |ab|).
 *        A matrix-wide scale loses the low bits of rows far below the matrix abs-max — harmless where the
 *        reduction runs over those rows (the weight gradient), not for a forward pass or a data gradient,
 *        whose per-row form is mi_dense_fwd_planes / mi_dense_bwd_data_planes;
 *      - "bf16x3" otherwise: three bf16 parts, six products (dropped <= 3*2^-24 |ab|), any input, no scale.
 *      Both measure the same max error against fp64 as the fp32-input MFMA on the layer shapes.
 *   0 "fp32": v_mfma_f32_32x32x2_f32 (exact fp32 products).
 * Operands that cannot be read as float4 (e.g. the N = 1 logits layer) always take path 0. */
int32_t mi_set_gemm_mode(int32_t mode);
int32_t mi_get_gemm_mode(void);

/* Abs-max vectors.  An abs-max "scalar" is MI_AMAX_SLOTS floats in device memory whose largest entry
 * is the value (producers spread their atomic max over the slots); zero it (hipMemsetAsync) before
 * the producer runs.  a / b: abs-max of the first / second matrix operand of the call, or an upper
 * bound of it (an under-estimate overflows fp16).  out: receives max |result| — every GEMM epilogue
 * can emit it, so the next layer's operand arrives with its abs-max already known.  A NULL struct, or
 * a NULL a or b, selects the bf16x3 split; out may always be NULL. */
#define MI_AMAX_SLOTS 64
typedef struct mi_gemm_amax {
  const float* a;
  const float* b;
  float* out;
} mi_gemm_amax_t;

/* amax_out[slot] = max(amax_out[slot], max_i |x[i]|) for buffers no kernel here produced (weights
 * after the optimizer step, a concat received from an all-to-all). */
int32_t mi_absmax(const float* x, int64_t n, float* amax_out, mi_stream_t stream);

/* Y[M,N] = act( X[M,K] * W[K,N] + bias[N] ); `relu` selects the activation of tf.layers.dense (deep_fm.py:22,100:
 * params["activation"], default tf.nn.relu): 0 none, 1 relu, 2 sigmoid, 3 tanh.  If keep_prob < 1 the
 * TRAIN-mode dropout of deep_fm.py:102-103 is applied after the activation with a counter-based
 * mask (seed, layer, element index): survivors scaled by 1/keep_prob. */
int32_t mi_dense_fwd(const float* X, int64_t ldx, const float* W, const float* bias, float* Y,
                     int64_t ldy, int64_t M, int32_t N, int32_t K, int32_t relu, float keep_prob,
                     uint64_t seed, const mi_gemm_amax_t* amax, mi_stream_t stream);

/* Layer 1 with the input_layer concat (deep_fm.py:54) read IN PLACE from the embedding table:
 * X[b, f*E + e] = table[field_off[f] + ids[b*F+f], e], K = F*E.  The GEMM's A operand is gathered
 * row by row (ids are fetched one k-tile ahead of the rows), so the [M, F*E] concat is never
 * written to or re-read from HBM.  Otherwise identical to mi_dense_fwd / mi_dense_bwd_weight. */
int32_t mi_dense_fwd_gathered(const float* table, const int64_t* field_off, const int32_t* ids, int32_t F,
                              int32_t E, const float* W, const float* bias, float* Y, int64_t ldy, int64_t M,
                              int32_t N, int32_t relu, float keep_prob, uint64_t seed,
                              const mi_gemm_amax_t* amax, int64_t table_stride, mi_stream_t stream);
int32_t mi_dense_bwd_weight_gathered(const float* table, const int64_t* field_off, const int32_t* ids, int32_t F,
                                     int32_t E, const float* dY, int64_t lddy, float* dW, float* db, int64_t M,
                                     int32_t N, void* workspace, size_t workspace_bytes,
                                     const mi_gemm_amax_t* amax, int64_t table_stride, mi_stream_t stream);

/* dX[M,K] = (dY[M,N] * W[K,N]^T) .* mask.  When Xact != NULL (the previous layer's stored
 * post-relu, post-dropout output) mask = (Xact > 0) / keep_prob — a unit with Xact > 0 was both
 * relu-active and kept, every other unit gets gradient 0 — so dX is the gradient w.r.t. the
 * previous layer's PRE-activation, ready to be that layer's dY.
 * activation (as mi_dense_fwd's): for sigmoid / tanh / none the factor is act'(pre) expressed through the stored
 * output (y (1 - y), 1 - y^2, 1) divided by keep_prob; with dropout a stored output that is exactly 0 counts
 * as dropped. */
int32_t mi_dense_bwd_data(const float* dY, int64_t lddy, const float* W, const float* Xact,
                          int64_t ldxa, float* dX, int64_t lddx, int64_t M, int32_t N, int32_t K,
                          float keep_prob, int32_t activation, const mi_gemm_amax_t* amax, mi_stream_t stream);

/* dW[K,N] = X[M,K]^T * dY[M,N], db[N] = column sums of dY.  Split-K over M with a
 * fixed-order slab reduction (bitwise reproducible). */
size_t mi_dense_bwd_weight_workspace_bytes(int64_t M, int32_t N, int32_t K);
int32_t mi_dense_bwd_weight(const float* X, int64_t ldx, const float* dY, int64_t lddy, float* dW,
                            float* db, int64_t M, int32_t N, int32_t K, void* workspace,
                            size_t workspace_bytes, const mi_gemm_amax_t* amax, mi_stream_t stream);

/* ---- (a6) the same layers with operands held as pre-split 16-bit planes ------------------------------
 * A "planes" matrix [rows][K] stores x * 2^s_r = hi + lo (fp16 high and low parts, RNE) with one power-
 * of-two exponent s_r per row, chosen from the row's abs-max (-> [2^14, 2^15)): ~22 significant bits
 * relative to the ROW's largest element, whatever the spread between rows.  Memory layout, k-block major:
 * for every block of 16 columns, all rows back to back, 64 bytes per row (16 x hi then 16 x lo); block j
 * starts at data + j * blk_stride (blk_stride >= 64 * rows, a multiple of 64; data 16-byte aligned; columns
 * K..ceil16(K) are zero).  The 16-k tile of a range of rows is one contiguous run of full cache lines.
 * Producers: mi_split_rows (fp32 -> planes; the weights once per step, W itself for the data gradient and
 * W transposed for the forward pass), the epilogues of the two GEMMs below (the next layer's operand never
 * exists in fp32) and mi_embed_fm_planes_fwd (the input_layer concat).  The GEMMs run three f16 MFMA
 * products per k-step (hi*hi, hi*lo, lo*hi; fp32 accumulate) fed by LDS-DMA with no arithmetic in the loop;
 * row exponents are undone exactly in the epilogue.  Shapes: N and K multiples of 16; a planes result needs
 * its width <= 512 (one workgroup then owns whole rows and knows their abs-max) — otherwise take the fp32
 * result and mi_split_rows, or the any-shape entries above. */
typedef struct mi_planes {
  void* data;          /* [ceil16(K)/16][blk_stride bytes]: row r of block j at data + j*blk_stride + 64*r */
  int32_t* row_exp;    /* [rows] s_r */
  int64_t blk_stride;  /* bytes between consecutive 16-column blocks */
} mi_planes_t;

size_t mi_planes_bytes(int64_t rows, int32_t K);   /* size of data with blk_stride = 64 * rows */

/* out row r = X[r, 0:K] (transpose == 0) or X[0:K, r] (transpose != 0: X is [K][rows], ldx >= rows).
 * amax_out (may be NULL): abs-max vector (MI_AMAX_SLOTS floats) receiving max |X|. */
int32_t mi_split_rows(const float* X, int64_t ldx, int64_t rows, int32_t K, int32_t transpose,
                      const mi_planes_t* out, float* amax_out, mi_stream_t stream);
/* Every hidden layer's kernel as planes, both orientations, in one launch (once per training step).
 * Job q: W = dense + offset, [K][N] row-major; w (rows = K, the data gradient's operand) and / or wt (rows
 * = N, W transposed: the forward pass's) — a NULL data pointer skips one.  All rows get ONE exponent, from
 * amax (abs-max vector of the parameter block, mi_absmax): weights more than 2^-17 below the block's
 * largest lose low bits. */
#define MI_MAX_WEIGHT_JOBS 8
typedef struct mi_weight_job {
  int64_t offset;
  int32_t K, N;
  mi_planes_t w, wt;
} mi_weight_job_t;
int32_t mi_split_weights(const float* dense, const mi_weight_job_t* jobs, int32_t n_jobs, const float* amax,
                         mi_stream_t stream);
/* X[r, k] = (hi + lo) * 2^-s_r (tests, summaries) */
int32_t mi_merge_rows(const mi_planes_t* in, int64_t rows, int32_t K, float* X, int64_t ldx, mi_stream_t stream);

/* Y[M][N] = act(X[M][K] * W[K][N] + bias) with X as planes and Wt = planes of W TRANSPOSED (rows = N);
 * epilogue as mi_dense_fwd.  Y (fp32) and / or Yp (planes, N <= 512; a positive value keeps a positive high
 * part, so "hi > 0" is the relu/dropout mask of the data gradient) receive the result; amax_out (may be
 * NULL) its abs-max.
 * mask_bits_out (may be NULL; round 4): the same mask as ONE BIT per output — [M][mask_ld] 32-bit words, mask_ld >=
 * ceil(N / 32), bit (n & 31) of word (n >> 5) of row m = "Y[m][n] > 0" (active and kept).  The data gradients below take
 * it instead of the stored activation: 1/32 of the bytes (2 MB instead of the 134 MB high-plane read at 65536 x 512). */
int32_t mi_dense_fwd_planes(const mi_planes_t* X, const mi_planes_t* Wt, const float* bias, float* Y, int64_t ldy,
                            const mi_planes_t* Yp, int64_t M, int32_t N, int32_t K, int32_t relu, float keep_prob,
                            uint64_t seed, float* amax_out, uint32_t* mask_bits_out, int64_t mask_ld, mi_stream_t stream);

/* dX[M][K] = (dY[M][N] * W[K][N]^T) .* mask with dY as planes and W = planes of W as stored (rows = K);
 * mask from Xact (planes of the previous layer's stored output: hi > 0 <=> active and kept; NULL: none),
 * survivors divided by keep_prob.  dX (fp32) and / or dXp (planes, K <= 512).
 * mask_bits / mask_ld (may be NULL): the mask as mi_dense_fwd_planes' mask_bits_out wrote it ([M][mask_ld] words, one bit per
 * element of dX's K columns); taken instead of Xact when both are given — the same decisions, hence the same bits. */
int32_t mi_dense_bwd_data_planes(const mi_planes_t* dY, const mi_planes_t* W, const mi_planes_t* Xact, float* dX,
                                 int64_t lddx, const mi_planes_t* dXp, int64_t M, int32_t N, int32_t K, float keep_prob,
                                 float* amax_out, const uint32_t* mask_bits, int64_t mask_ld, mi_stream_t stream);

/* The logits layer AND the head of a TRAIN step in one pass over the last hidden layer's output X [M][K] (round 4): replaces
 * deep_fm.py:108 (tf.layers.dense(net, 1)), :111 (logits += dnn_logits), :118-125 (the sigmoid cross-entropy head) and
 * their gradients — five launches of the entries above and below (mi_dense_fwd's N = 1 form, mi_sigmoid_ce_head,
 * mi_dense_bwd_weight's N = 1 form, mi_dense_bwd_data_vec_planes) — with two:
 *   dnn[m] = X[m,:] . w + b;  logits[m] = lin[m] + lin_bias + fm[m] + dnn[m]  (lin / fm may be NULL);  loss_out = sum of the
 *   per-example sigmoid cross-entropy * loss_scale;  d_logit[m] = (sigmoid(logits) - label) * loss_scale;  d_logit_sum (may be
 *   NULL) and db = sum d_logit;  dW[k] = sum_m X[m][k] d_logit[m];  dXp (planes, + dX in fp32 if not NULL) = d_logit[m] w[k],
 *   kept where mask_bits says so and divided by keep_prob (mask_bits NULL: no mask) — per example the bits of the unfused
 *   sequence; the sums over examples associate differently (fixed order, reproducible).  K in {64, 128, 256}; X 16-byte
 *   aligned, ldx a multiple of 4. */
size_t mi_logits_head_fused_workspace_bytes(int64_t M, int32_t K);
int32_t mi_logits_head_fused(const float* X, int64_t ldx, const float* w, const float* b, const float* lin, const float* lin_bias,
                             const float* fm, const uint8_t* labels, int64_t M, int32_t K, float loss_scale,
                             const uint32_t* mask_bits, int64_t mask_ld, float keep_prob, float* dnn, float* logits,
                             float* loss_out, float* d_logit, float* d_logit_sum, float* dW, float* db, const mi_planes_t* dXp,
                             float* dX, int64_t lddx, float* amax_out, void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* ... and the LAST HIDDEN layer with them (round 4): Y = dropout(relu(X W + bias)) [M][N] from planes as in
 * mi_dense_fwd_planes, then — in the GEMM's epilogue, Y never leaves the chip — everything mi_logits_head_fused does with it:
 * dnn, logits, loss, d_logit, the bias gradient(s), the logits layer's weight gradient dW [N] and its data gradient
 * dXp [M][N] (planes: d_logit[m] w[n], kept where Y[m][n] > 0, divided by keep_prob).  Replaces deep_fm.py:98-125 from the
 * last hidden layer up, forward and backward, with two launches (the GEMM and a fold of its workgroups' N + 2 partial sums);
 * saves the layer's output written and read back (2 x 4 M N bytes) and its mask bits.  N = 128 (one column tile holds the whole
 * layer); K a multiple of 16.  Per example the arithmetic of the two entries it fuses; the sum h . w of the logits layer and the
 * sums over examples associate in its own fixed order (tests: 2e-6 of sum |h w|, everything downstream on its own dnn). */
size_t mi_hidden_logits_head_fused_workspace_bytes(int64_t M, int32_t N);
int32_t mi_hidden_logits_head_fused(const mi_planes_t* X, const mi_planes_t* Wt, const float* bias, int64_t M, int32_t N, int32_t K,
                                    int32_t relu, float keep_prob, uint64_t seed, const float* w, const float* b, const float* lin,
                                    const float* lin_bias, const float* fm, const uint8_t* labels, float loss_scale, float* dnn,
                                    float* logits, float* loss_out, float* d_logit, float* d_logit_sum, float* dW, float* db,
                                    const mi_planes_t* dXp, float* amax_out, void* workspace, size_t workspace_bytes,
                                    mi_stream_t stream);

/* The data gradient of the N = 1 logits layer (deep_fm.py:108 backward) with the result as planes:
 * dX[m][k] = dY[m] * W[k], kept where Xact[m][k] > 0 and divided by keep_prob (Xact == NULL: no mask) — the
 * arithmetic of mi_dense_bwd_data's N = 1 form with ReLU, bit for bit — into dXp (required) and, if not NULL, dX.
 * amax_out as everywhere.  K a multiple of 16.  mask_bits / mask_ld: the one-bit mask (see mi_dense_fwd_planes), taken
 * instead of Xact when given. */
int32_t mi_dense_bwd_data_vec_planes(const float* dY, int64_t lddy, const float* W, const float* Xact, int64_t ldxa,
                                     float keep_prob, float* dX, int64_t lddx, const mi_planes_t* dXp, int64_t M, int32_t K,
                                     float* amax_out, const uint32_t* mask_bits, int64_t mask_ld, mi_stream_t stream);

/* dW[K][N] = X[M][K]^T * dY[M][N] and db[N] = column sums of dY (NULL: skipped), both operands as planes
 * (replaces the same gradients as mi_dense_bwd_weight: model_utils.py:69-72 through tf.layers.dense,
 * deep_fm.py:98-108).  The reduction runs over the examples, so every example's planes are brought to the
 * matrix-wide scales by one power of two (from its two row exponents and amax->a / amax->b, the abs-max vectors
 * of X and dY, both required); examples far below the abs-max lose low bits — invisible in a sum over
 * examples, as in mi_dense_bwd_weight's matrix-wide split.  M % 32 == 0, N % 128 == 0, K % 128 == 0 (anything
 * else: mi_dense_bwd_weight on fp32 copies).  Split-K slabs folded in a fixed order: bitwise reproducible. */
size_t mi_dense_bwd_weight_planes_workspace_bytes(int64_t M, int32_t N, int32_t K);
int32_t mi_dense_bwd_weight_planes(const mi_planes_t* X, const mi_planes_t* dY, float* dW, float* db, int64_t M,
                                   int32_t N, int32_t K, void* workspace, size_t workspace_bytes,
                                   const mi_gemm_amax_t* amax, mi_stream_t stream);

/* The weight gradients of SEVERAL layers from planes in one call (round 4): one launch makes every layer's per-example
 * factors, one launch per layer runs its split-K GEMM, one launch folds every layer's slabs — for the three hidden layers of
 * config 3 five launches instead of nine (the backward then runs all data gradients first: each needs only the layer
 * above's dY, and the factors of every layer need its finished dY planes and abs-max).  Results are those of
 * mi_dense_bwd_weight_planes per job, bit for bit.  workspace: mi_dense_bwd_weight_planes_batch_workspace_bytes, 32-byte
 * aligned; at most MI_MAX_WEIGHT_JOBS jobs. */
typedef struct mi_wgrad_job {
  mi_planes_t X, dY;        /* the layer's input and the gradient of its output, [M] rows each */
  float* dW; float* db;     /* [K][N], [N] (db may be NULL) */
  int32_t N, K;
  mi_gemm_amax_t amax;      /* a = abs-max vector of X, b = of dY */
} mi_wgrad_job_t;
size_t mi_dense_bwd_weight_planes_batch_workspace_bytes(const mi_wgrad_job_t* jobs, int32_t n_jobs, int64_t M);
int32_t mi_dense_bwd_weight_planes_batch(const mi_wgrad_job_t* jobs, int32_t n_jobs, int64_t M, void* workspace,
                                         size_t workspace_bytes, mi_stream_t stream);

/* ---- (a7,a8) logits sum + sigmoid cross-entropy head -----------------------------------------
 * replaces `logits += ...` (deep_fm.py:36,44,90,111) and
 * tf.contrib.estimator.binary_classification_head (deep_fm.py:118-125).
 *   logits[b] = ((lin[b] + lin_bias) + fm[b]) + dnn[b]        (NULL terms skipped, same order)
 *   loss_b    = max(x,0) - x*y + log1p(exp(-|x|))
 *   loss_out[0] = sum_b loss_b * loss_scale   (loss_scale = 1/B_global for the contrib head's
 *                 SUM_OVER_BATCH_SIZE, 1 for the canned estimators' SUM)
 *   d_logit[b]  = (sigmoid(x) - y) * loss_scale
 *   d_logit_sum[0] = sum_b d_logit[b]   (= gradient of the linear_model bias; fixed-order sum)
 * labels: uint8 0/1 (ml_100k.py:48 rating >= cutoff).  d_logit / d_logit_sum / loss_out may be NULL
 * (eval / predict).  workspace: mi_head_workspace_bytes(B). */
size_t mi_head_workspace_bytes(int64_t B);
int32_t mi_sigmoid_ce_head(const float* lin, const float* lin_bias, const float* fm,
                           const float* dnn, const uint8_t* labels, int64_t B, float loss_scale,
                           float* logits, float* loss_out, float* d_logit, float* d_logit_sum,
                           void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* The head's per-example outputs: get_binary_predictions (model_utils.py:9-20; the PREDICT dict of
 * binary_classification_head) and get_binary_losses' unreduced loss (model_utils.py:23-36).
 *   logistic[b] = sigmoid(x)   probabilities[b] = {1 - p, p}   class_ids[b] = p > 0.5   (int64, as TF's)
 *   unreduced_loss[b] = max(x,0) - x*y + log1p(exp(-|x|))      (needs labels)
 * Any output may be NULL. */
int32_t mi_binary_predictions(const float* logits, const uint8_t* labels, int64_t B, float* logistic,
                              float* probabilities, int64_t* class_ids, float* unreduced_loss, mi_stream_t stream);

/* column sums: out[j] = sum_b X[b,j] (used for bias-style gradients), deterministic. */
size_t mi_colsum_workspace_bytes(int64_t M, int32_t N);
int32_t mi_colsum(const float* X, int64_t ldx, int64_t M, int32_t N, float* out, void* workspace,
                  size_t workspace_bytes, mi_stream_t stream);

/* ---- (a10) layer_summary statistics (model_utils.py:4-6) ------------------------------------
 * out[0] = fraction of zeros, out[1] = min, out[2] = max, out[3] = mean. */
size_t mi_layer_stats_workspace_bytes(int64_t n);
int32_t mi_layer_stats(const float* x, int64_t n, float* out4, void* workspace,
                       size_t workspace_bytes, mi_stream_t stream);

/* The histogram of layer_summary (tf.summary.histogram("activation", value), model_utils.py:6): counts[b] += 1
 * for b = number of limits <= x[i] (std::upper_bound over the ascending fp64 bucket limits, as TensorFlow's
 * histogram::Histogram::Add; its default limits are +-1e-12 * 1.1^k up to 1e20, 0 and +-DBL_MAX: 1,551 of
 * them — mi355x_rec/metrics.py builds them), sums[0] += sum x, sums[1] += sum x^2 (fp64).  counts
 * [n_limits + 1] int64 and sums [2] are accumulated into: zero them first.  n_limits <= 2048. */
int32_t mi_layer_histogram(const float* x, int64_t n, const double* limits, int32_t n_limits, int64_t* counts,
                           double* sums, mi_stream_t stream);

/* ---- (f3) streaming eval metrics: tf.metrics.auc confusion counts --------
 * Accumulates (integer atomics, order independent) into device arrays the caller zeroed:
 *   hist   [2*201] int64: hist[y*201+k] = examples with label y whose sigmoid exceeds exactly k of
 *          tf.metrics.auc's 200 ascending thresholds; tp[j] = sum_{k>j} hist[1][k] etc. on the host
 *   counts [8] int64: n, n_pos, n_pred_pos, n_correct, tp@.5, fp@.5, fn@.5, (unused)
 *   sums   [4] f64 : sum(loss_b), sum(sigmoid), sum(label), (unused) */
int32_t mi_eval_accumulate(const float* logits, const uint8_t* labels, int64_t B,
                           int64_t* hist /*[2*201]*/, int64_t* counts /*[8]*/,
                           double* sums /*[4]*/, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_REC_H */
