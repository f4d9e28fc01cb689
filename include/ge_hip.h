/* ge_hip.h -- flat C ABI of libge_hip.so, the MI355X (gfx950) drop-in for the hot path of
 * greysun/GraphEmbeddings' holE.py (gather -> max-norm clip -> ComplEx / HolE score -> sigmoid ->
 * pairwise hinge -> gradient of the SUM -> sparse SGD scatter), plus the type-safe corruption
 * sampler and the 1-vs-K candidate scorer.
 *
 * The holE.py path has no FFI of its own (it is in-process TensorFlow); the calling convention
 * below is the one the reference uses for its only native library, init.so
 * (init.cpp:47-48,129-142,223-224 loaded by transE.py:9-10, buffers passed as raw addresses
 * transE.py:95-112): C linkage, caller-owned caller-sized flat buffers, int32 ids.  Differences
 * that make it usable from a GPU training loop: every entry point returns an int status
 * (0 = ok, <0 = -errno style argument error, >0 = hipError_t), takes the hipStream_t it must
 * enqueue on (as void*), never synchronises, allocates no device memory of its own -- with ONE exception,
 * stated where it applies: ge_rank_1vK / ge_complex_rank_1vK / ge_rank_1vK_vs_loss without a `planes` buffer and
 * ge_complex_score_1vK on large sweeps build the candidates' fp16 planes in a stream-ordered allocation
 * (hipMallocAsync / hipFreeAsync on `stream`, nothing outlives the call; pass ge_rank_planes' buffer to avoid it) --
 * and keeps no global state (the only state is inside the explicit ge_train_pipeline handle, see ge_train_steps).
 *
 * All pointers except where noted are DEVICE pointers (e.g. torch.Tensor.data_ptr()).
 * `table` is the single shared entity+relation table of holE.py:263-264: row-major fp32 [N, d],
 * first d/2 floats of a row = real parts, last d/2 = imaginary parts (holE.py:164-166).
 * Triples are int32 [B,3] in the reference's column order (head, tail, relation)
 * (holE.py:76-81, 181-185).  A triple with an id outside [0,N) yields NaN / is skipped.
 */
#ifndef GE_HIP_H
#define GE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GE_VERSION 320 /* 0.3.2: + ge_rank_1vK_vs_loss (ranks against given losses: sharded candidate lists, the is_confident gate) */

/* argument errors (negative, -errno style) */
#define GE_EINVAL (-22)  /* bad dimension / null pointer / misaligned buffer */
#define GE_ENOTSUP (-95) /* dimension outside the compiled kernel range */
#define GE_ENOMEM (-12)  /* workspace too small */

/* ge_corrupt_batch modes */
#define GE_CORRUPT_BATCH_COIN 0 /* reference: one heads/tails coin per batch (holE.py:137-140) */
#define GE_CORRUPT_ROW_COIN 1   /* one coin per row (extension) */
#define GE_CORRUPT_HEADS 2      /* holE.py:97-114 */
#define GE_CORRUPT_TAILS 3      /* holE.py:117-133 */

/* model codes (ge_hinge_loss, ge_hinge_grad, ge_train_steps) */
#define GE_MODEL_COMPLEX 0       /* holE.py:191-192 */
#define GE_MODEL_HOLE 1          /* README.md:42 on a real-valued table */
#define GE_MODEL_HOLE_SPECTRAL 2 /* the same model on a table transformed by ge_hole_to_spectral */
#define GE_MODEL_HOLE_DIRECT 3   /* ge_train_steps only: force the direct-correlation kernels */
/* ge_train_steps only, OR-ed into `model` (prepared path): rows with more than 16 gradient slots in a step -- split over
 * several work items, whose partial sums otherwise meet by float atomics in scheduler order -- are reduced in a FIXED
 * order (one more launch per step), so that every row of the table is bitwise reproducible run to run whatever the ids'
 * distribution.  Off by default: it costs a launch per step (measured: DESIGN.md section 5). */
#define GE_STEP_DETERMINISTIC 0x100

int ge_version(void);

/* Largest embedding_dim the compiled kernels accept (score path / train path). */
int ge_max_dim(void);

/* --- evaluate_triples (holE.py:179-202), ComplEx: out[i] = sigma(sum_k Re(h_k r_k conj(t_k)))
 * with rows clipped to max_norm as tf.nn.embedding_lookup(max_norm=1) does (holE.py:162).
 * apply_sigmoid=0 returns the raw score.  d must be even.  out: [B] fp32. */
int ge_complex_score(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                     float max_norm, int apply_sigmoid, float* out, void* stream);
/* The same on a table whose rows are `ld` >= d floats apart (row i at table + i * ld; the reference's table is dense,
 * ld = d).  Exists to measure padded row layouts (800-byte rows straddle 64-byte sectors; profiles/r03_padded_rows.txt). */
int ge_complex_score_strided(const float* table, int64_t N, int32_t d, int64_t ld, const int32_t* triples, int64_t B,
                             float max_norm, int apply_sigmoid, float* out, void* stream);

/* --- evaluate_triples(triple_batch, embeddings, label) in --log_loss mode (holE.py:194-196), forward only:
 * out[i] = log(1 + exp(-label * score_i)) + l2 * sum(table^2) / 2 (tf.nn.l2_loss of the WHOLE table).
 * workspace: >= 256 bytes of device scratch (the table's sum of squares). */
int ge_complex_logloss(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B, float label,
                       float l2, float max_norm, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* --- HolE score of README.md:42: sigma(sum_k r_k [h (star) t]_k), circular correlation over the
 * full d real values of the clipped rows.  Same layout as ge_complex_score. */
int ge_hole_score(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                  float max_norm, int apply_sigmoid, float* out, void* stream);

/* --- HolE in the frequency domain.  ge_hole_to_spectral replaces every row x (d reals, d even, k = d/2)
 * IN PLACE by its half spectrum X_f = sum_n x_n exp(-2 pi i f n / d), packed into the same d floats as
 * [Re X_0 .. Re X_{k-1} | Re X_k, Im X_1 .. Im X_{k-1}] (X_0 and the Nyquist bin X_k are real);
 * ge_hole_from_spectral is the inverse.  On a spectral table the HolE score of README.md:42 is
 * (1/d) sum_f w_f Re(H_f R_f conj(T_f)), w_0 = w_k = 1, else 2 -- the ComplEx-shaped trilinear form --
 * and |x|^2 = (1/d) sum_f w_f |X_f|^2, so scoring, the max-norm clip and the SGD step run without any
 * transform (model GE_MODEL_HOLE_SPECTRAL).  ge_hole_spectral_score = ge_hole_score on such a table. */
int ge_hole_to_spectral(float* table, int64_t N, int32_t d, void* stream);
int ge_hole_from_spectral(float* table, int64_t N, int32_t d, void* stream);
int ge_hole_spectral_score(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                           float max_norm, int apply_sigmoid, float* out, void* stream);

/* --- evaluate_batch forward only (holE.py:222-234): loss[i] = max(E(pos_i) - E(neg_i) + margin, 0).
 * model: GE_MODEL_COMPLEX, GE_MODEL_HOLE or GE_MODEL_HOLE_SPECTRAL.  sig_out (nullable) receives E(pos) in
 * [0,B) and E(neg) in [B,2B). */
int ge_hinge_loss(const float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                  int64_t B, float margin, float max_norm, int model, float* loss, float* sig_out,
                  void* stream);

/* --- one validation tick of the training loop (holE.py:299-304, 351-360), enqueued without any host decision:
 * a batch of B triples drawn uniformly (with replacement, Philox keyed by (seed, counter)) from the device-resident
 * validation split valid [V,3]; type-safe negatives (ge_corrupt_batch with step = counter); the hinge of
 * ge_hinge_loss; its mean -> *mean_out (device); and the reference's "pocket": if the mean is below *best (device
 * scalar, start it at 2.0 as holE.py:336 does) then *best = mean and, when pocket is not NULL, pocket <- table.
 * The host reads the means whenever it likes (e.g. once per epoch) and knows from them which tick the pocket
 * holds.  workspace: >= ge_validation_workspace_bytes(B), 16-byte aligned. */
size_t ge_validation_workspace_bytes(int64_t B);
int ge_validation_tick(const float* table, int64_t N, int32_t d, const int32_t* valid, int64_t V, int64_t B,
                       const int32_t* id_to_type, const int64_t* type_offsets, int32_t n_types, const int32_t* type_ids,
                       uint64_t seed, uint64_t counter, int32_t padded_size, int32_t mode, float margin, float max_norm,
                       int model, void* workspace, size_t workspace_bytes, float* mean_out, float* best, float* pocket,
                       void* stream);
/* The same tick for --log_loss (holE.py:194-196, 206-220 evaluated on a validation batch): the batch with label +1,
 * negative_ratio corrupted batches (ge_corrupt_batch with step = counter * negative_ratio + k) with label -1, every
 * loss = log(1 + exp(-y s)) + l2 * sum(table^2) / 2; mean over the (1 + negative_ratio) * B values, pocket as above. */
size_t ge_validation_logloss_workspace_bytes(int64_t B, int32_t negative_ratio);
int ge_validation_tick_logloss(const float* table, int64_t N, int32_t d, const int32_t* valid, int64_t V, int64_t B,
                               const int32_t* id_to_type, const int64_t* type_offsets, int32_t n_types,
                               const int32_t* type_ids, uint64_t seed, uint64_t counter, int32_t padded_size, int32_t mode,
                               int32_t negative_ratio, float l2, float max_norm, void* workspace, size_t workspace_bytes,
                               float* mean_out, float* best, float* pocket, void* stream);

/* --- one SGD step of holE.py:296 on the hinge of holE.py:231: forward of both sides, gradient of
 * SUM_i loss_i through sigmoid, score and the clip, then table[row] -= lr * grad for every
 * occurrence (ScatterSub semantics, duplicates accumulate; holE-20170724/graph.pbtxt:47850-48001).
 * Every gradient is evaluated against the table as it was before the call.  In place on `table`.
 * workspace: device scratch of at least ge_hinge_step_workspace_bytes(B, d) bytes, 256-B aligned.
 * loss: [B] fp32.  Rows that pos and neg share (relation, and the uncorrupted entity) are read
 * and updated once with the summed gradient. */
size_t ge_hinge_step_workspace_bytes(int64_t B, int32_t d);
int ge_complex_hinge_step(float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                          int64_t B, float margin, float lr, float max_norm, float* loss,
                          void* workspace, size_t workspace_bytes, void* stream);
int ge_hole_hinge_step(float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                       int64_t B, float margin, float lr, float max_norm, float* loss,
                       void* workspace, size_t workspace_bytes, void* stream);

/* --- the --log_loss branch (holE.py:194-196, 206-220 + minimize, holE.py:296), ComplEx: `triples`
 * [M,3] is the concatenation of the B positives and negative_ratio corrupted batches, `labels` [M]
 * fp32 +1 / -1 (holE.py:209, 215).  loss[i] = log(1 + exp(-y_i s_i)) + l2 * sum(table^2)/2
 * (tf.nn.l2_loss of the WHOLE table, holE.py:196).  minimize() differentiates the SUM of the loss
 * vector, so the dense term counts once per row: table <- table*(1 - lr*M*l2) - lr * sparse grads.
 * workspace >= ge_logloss_step_workspace_bytes(M, d), 256-B aligned. */
size_t ge_logloss_step_workspace_bytes(int64_t M, int32_t d);
int ge_complex_logloss_step(float* table, int64_t N, int32_t d, const int32_t* triples, const float* labels,
                            int64_t M, float lr, float l2, float max_norm, float* loss, void* workspace,
                            size_t workspace_bytes, void* stream);

/* --- the two halves of the step, exposed for the row-sharded multi-GPU path (rows are fetched
 * from / gradients routed to their owner GPU between the halves).
 * ge_hinge_grad: `rows` is any [N,d] row store (the table itself, or a staging buffer of fetched
 * rows with pos/neg re-indexed into it).  Emits IndexedSlices: grad_idx [6B] int32 (row index in
 * `rows`, or -1 for an empty slot) and grad_val [6B,d] fp32 already multiplied by -lr, slot order
 * per pair: h+, t+, r+, h-, t-, r-.  model: GE_MODEL_COMPLEX, GE_MODEL_HOLE or GE_MODEL_HOLE_SPECTRAL
 * (then `rows` holds spectral rows and the emitted gradient rows are spectral too).
 * ge_scatter_add_rows: table[idx[i]] += val[i] for idx[i] >= 0, float atomics.
 * ge_gather_rows: out[i] = table[idx[i]] (zeros for idx[i] < 0). */
int ge_hinge_grad(const float* rows, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                  int64_t B, float margin, float lr, float max_norm, int model, float* loss,
                  int32_t* grad_idx, float* grad_val, void* stream);
int ge_scatter_add_rows(float* table, int64_t N, int32_t d, const int32_t* idx, const float* val,
                        int64_t R, void* stream);
int ge_gather_rows(const float* table, int64_t N, int32_t d, const int32_t* idx, int64_t R,
                   float* out, void* stream);

/* --- corrupt_batch (holE.py:152-153 -> 136-140 -> 97-133) fused with the per-batch host
 * resample of holE.py:343-347.  id_to_type [N] int32 type code per table row (-1 = unknown, the
 * reference's '?' default -> corrupted id -1, holE.py:39); type lists as CSR type_offsets
 * [n_types+1] int64 / type_ids int32.  Row i's replaced entity is drawn as the reference does:
 * uniform slot in [0,padded_size) of a per-batch with-replacement subsample of its type's list
 * (padded_size = 0: uniform over the whole list).  Randomness is a counter-based Philox4x32-10
 * stream keyed by (seed, step, row/type) -- no state, bitwise reproducible (oracle/hole_oracle.py
 * states the stream).  neg: [B,3] int32. */
int ge_corrupt_batch(const int32_t* pos, int64_t B, const int32_t* id_to_type, int64_t N,
                     const int64_t* type_offsets, int32_t n_types, const int32_t* type_ids,
                     uint64_t seed, uint64_t step, int32_t padded_size, int32_t mode, int32_t* neg,
                     void* stream);

/* --- Bernoulli filtered corruption: GPU form of the reference's native TransX sampler init.so
 * (getBatch / corrupt_head / corrupt_tail, init.cpp:159-246; tph/hpt statistics init.cpp:107-127).
 * Known triples are given as two sorted indexes: bh_* sorted by (head, relation, tail) with
 * bh_key = head*n_rel + relation and bh_ent = tail; bt_* sorted by (tail, relation, head) likewise.
 * tail_threshold[r] = floor(2^32 * hpt_r / (hpt_r + tph_r)): a row corrupts its TAIL iff a uniform
 * 32-bit word is below it, else its head (init.cpp:226-228).  The replacement is uniform over the
 * entities [ent_lo, ent_lo+n_ent) that do not complete a known triple (filtered, init.cpp:177-188).
 * Unlike init.so (void returns, one global LCG, not thread-safe) this is stateless: per-row
 * Philox4x32-10 draws keyed by (seed, step, row); oracle/transx_oracle.py states the stream and is
 * itself pinned against the compiled init.cpp. */
int ge_bernoulli_corrupt_batch(const int32_t* pos, int64_t B, const int64_t* bh_key, const int32_t* bh_ent,
                               const int64_t* bt_key, const int32_t* bt_ent, int64_t n_known,
                               const uint32_t* tail_threshold, int32_t n_rel, int32_t ent_lo,
                               int32_t n_ent, uint64_t seed, uint64_t step, int32_t* neg, void* stream);

/* --- 1-vs-K candidate scoring (the inference loop of holE.py:564-569: fixed (head, relation)
 * against many tails; also K shared negatives per positive).  hr: [B,2] int32 (fixed entity,
 * relation); cand: [K] int32 candidate entity rows; cand_is_head = 0 scores (fixed, cand_j, rel),
 * 1 scores (cand_j, fixed, rel).  out: [B,K] fp32 row-major.  Runs as an MFMA GEMM
 * S = Q . T^T with Q = clip(fixed) o clip(rel) (complex product) and T the clipped candidates: fp32 MFMA, or --
 * d % 8 == 0 in 56 ... 224 (small sweeps) / 56 ... 288 (large ones) and max_norm <= 8 -- three f16 MFMAs on operands
 * split into fp16 high halves and remainders (22 bits), fp32 accumulation; scores within 1e-7 of fp64 either way. */
int ge_complex_score_1vK(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B,
                         const int32_t* cand, int64_t K, float max_norm, int apply_sigmoid,
                         int cand_is_head, float* out, void* stream);

/* --- link-prediction ranks straight out of the candidate sweep (holE.py:427-472 applied to the sweep of
 * holE.py:564-575): for test row i = (fixed entity, relation) = hr[i] with true candidate entity true_id[i]
 * (which must be in `cand`), over the K candidates,
 *     n_before[i]       = #{c : (E_ic, id_c) < (E_i,true, true_id[i])}   -- ascending loss, ties by entity id,
 *                         the pop order of the reference's heap; raw rank = 1 + n_before
 *     n_known_before[i] = how many of those are known-true candidates; filtered rank = raw - n_known_before.
 * The known-true candidates are given per (block of 128 test rows, tile of 128 candidates): known_off
 * [ceil(B/128) * ceil(K/128) + 1] int32 offsets into known_rc, whose entries are (row % 128) << 7 | (column % 128)
 * (both NULL: no filtering).  No [B,K] score matrix is written: the counting is the epilogue of the fp32-MFMA
 * GEMM (E = sigmoid(score) as in ge_complex_score_1vK).  true_loss (nullable) [B] receives E_i,true;
 * scores_out (nullable) [B,K] receives every loss -- for tests.  d must be a multiple of 8 and <= ge_rank_max_dim() (288; above 232 only with max_norm <= 8)
 * (the block's Q operand lives in LDS for the whole sweep), table 16-byte aligned; otherwise GE_ENOTSUP and the
 * caller falls back to ge_complex_score_1vK.
 * Arithmetic: fp32 MFMA with fp32 accumulation; for every d % 8 == 0 in 56 ... 288 and max_norm <= 8 the operands are split
 * into fp16 high halves and remainders (22 bits each, after the clip scales, so no fp16 overflow for any table) and
 * multiplied by three f16 MFMAs with fp32 accumulation -- losses within 1e-7 of the fp64 restatement either way
 * (ge_complex_score_1vK takes the same route for large sweeps at those dims). */
int ge_rank_max_dim(void);
int ge_complex_rank_1vK(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                        const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                        const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before, float* true_loss,
                        float* scores_out, void* stream);
/* The same sweep for `model` GE_MODEL_COMPLEX or GE_MODEL_HOLE_SPECTRAL (the README.md:42 HolE score on a table
 * held in the frequency domain, ge_hole_to_spectral: the ComplEx-shaped trilinear form with Hermitian weights, so
 * the candidate operand is still the row exactly as stored).  A real-valued HolE table (GE_MODEL_HOLE /
 * GE_MODEL_HOLE_DIRECT) is GE_ENOTSUP: transform a copy first. */
int ge_rank_1vK(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                const int32_t* cand, int64_t K, float max_norm, int model, int cand_is_head, const int32_t* known_off,
                const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before, float* true_loss,
                float* scores_out, void* stream);

/* The split-precision sweep (embedding_dim % 8 == 0 in 56 ... 288, max_norm <= 8) reads the candidates as pre-split fp16
 * planes (high halves and remainders of row * clip scale * 2^8, laid out per 64-candidate tile) plus an entity ->
 * candidate-position map.  ge_rank_1vK builds them on every call (one pass over the K candidate rows, in a
 * stream-ordered allocation); a caller that ranks many batches against the same (table, cand, max_norm, model) --
 * holE.py:564-575 sweeps all entities for every test triple, tails and heads -- builds them once:
 *   ge_rank_planes_bytes   bytes of the buffer (0: this embedding_dim has no such sweep -- pass planes = NULL)
 *   ge_rank_planes         fills it (256-byte aligned); valid until the table, cand or max_norm change
 *   ge_rank_1vK_planes     ge_rank_1vK with the buffer (NULL: as ge_rank_1vK).  cand_is_head does not enter the planes.
 * The contract says true_id[i] is in `cand`; where it is not, the split-precision sweep reports no rank (n_before =
 * n_known_before = 0, true_loss NaN) while the fp32 kernels rank the entity against the candidates all the same. */
int64_t ge_rank_planes_bytes(int64_t N, int32_t d, int64_t K);
int ge_rank_planes(const float* table, int64_t N, int32_t d, const int32_t* cand, int64_t K, float max_norm, int model,
                   void* planes, void* stream);
int ge_rank_1vK_planes(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                       const int32_t* cand, int64_t K, float max_norm, int model, int cand_is_head, const int32_t* known_off,
                       const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before, float* true_loss,
                       float* scores_out, const void* planes, void* stream);

/* The same sweep against GIVEN losses: n_before[i] = #{ j : E_ij < ref_loss[i], or E_ij == ref_loss[i] and cand[j] <
 * ref_id[i] }, n_known_before[i] = how many of those are listed in the known cells -- i.e. the position the reference's
 * heap (holE.py:427-472) would give a triple of loss ref_loss[i] and tail id ref_id[i] among THESE candidates, whether or
 * not it is one of them.  Two uses: (1) a candidate list sharded over several devices -- each sweeps its own candidates
 * against the true triple's loss (computed once, by whoever holds the true candidate: ge_rank_1vK_planes' true_loss) and
 * the counts ADD across shards (exactly: equal table rows give bit-equal losses on every device); (2) the reference's
 * `is_confident = min_loss < infer_threshold` gate (holE.py:436-438): min_loss < t  <=>  n_before(ref_loss = t, ref_id
 * = INT32_MIN) > 0.  ref_id need not be a table row.  planes: as ge_rank_1vK_planes (NULL: built per call).
 * embedding_dim % 8 == 0 up to ge_rank_max_dim() (above 232 only with the planes' conditions), else GE_ENOTSUP. */
int ge_rank_1vK_vs_loss(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* ref_id,
                        const float* ref_loss, const int32_t* cand, int64_t K, float max_norm, int model, int cand_is_head,
                        const int32_t* known_off, const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before,
                        const void* planes, void* stream);

/* --- the known_off / known_rc lists of ge_rank_1vK from a sorted index of the known-true triples (the filter of
 * holE.py:454-463, built by the reference as a dict of sets, holE.py:413-422).  known_key [M] ascending = fixed entity *
 * n_rows + relation of every distinct known (fixed, other, relation); known_ent [M] = the other entity; fixed / rel [B] =
 * the test rows; pos_of [n_rows] = position of an entity in the candidate list (-1: not a candidate).  Two passes:
 *   pass 0  known_off [tiles + 1] <- offsets of every (128 rows x 128 candidates) tile's cells; known_off[tiles] = total
 *   pass 1  known_rc [total] <- the cells, (row % 128) << 7 | (column % 128), in no particular order inside a tile
 * tile_scratch: int32 [tiles], the same buffer in both passes.  All ids are int64 (the evaluator's dtype). */
int ge_known_cells(int pass, const int64_t* known_key, const int64_t* known_ent, int64_t M, const int64_t* fixed,
                   const int64_t* rel, int64_t B, const int64_t* pos_of, int64_t n_rows, int64_t n_cand, int32_t* tile_scratch,
                   int32_t* known_off, uint16_t* known_rc, void* stream);

/* --- the inner training loop of holE.py:340-362 (minus validation), enqueued natively: for
 * s in [0, n_steps): batch = triples[(first_row + s*B) .. +B) (rows of a device-resident, already
 * shuffled [T,3] int32 array, wrapping to row 0 when the next batch would run past T -- the
 * reference's shuffle queue never yields a short batch, holE.py:283); negatives from
 * ge_corrupt_batch with step = global_step0 + s; lr_s = lr0 / (1 + decay_rate * (global_step0+s) /
 * decay_steps) (tf.train.inverse_time_decay, holE.py:292-294; decay_steps <= 0 keeps lr0); then one
 * ge_*_hinge_step.  No host synchronisation.
 *
 * workspace >= ge_train_workspace_bytes(B, d) enables the PREPARED path: the negatives and a
 * row-sorted index of the step's gradient slots are built ahead of time, for a chunk of steps per
 * launch (one workgroup per step and per sub-batch of 4096 pairs: sampler + stable LDS radix sort by
 * row), into two chunk buffers inside the workspace.  The update then touches every distinct row
 * once, with plain read-modify-writes at any B (float atomics only for rows with > 16 gradient slots in a
 * step; for B > 4096 the sort runs across workgroups, csrc/ge_prep_big.hip).  The workspace also holds a
 * ring of gradient-row regions, one per step in turn.  With only ge_hinge_step_workspace_bytes the
 * loop falls back to sampler + float-atomic scatter per step.
 *
 * pipeline (nullable): a handle from ge_train_pipeline_create, bound to the device that was current
 * at creation.  It owns a non-blocking side stream and four events on which the prepare launch for
 * the NEXT chunk of steps runs while `stream` executes the current one -- across calls too: when a
 * call continues exactly where the previous call with the same handle stopped (same arrays, sizes,
 * seed, mode, workspace; global_step0 and first_row advanced by the steps already run) its records
 * are already built and nothing but the two per-step kernels is on the critical path.  Any other
 * call restarts the sequence (its first prepare launch, ~20 us, is then exposed).  While a handle
 * is live the caller must not modify `triples`, the type tables or the workspace between calls
 * without ge_train_pipeline_reset; the streams are ordered by events only, the host never blocks,
 * and before returning every call makes `stream` wait for the look-ahead launch, so work enqueued
 * on `stream` afterwards (e.g. a stream-ordered free of the workspace) is ordered after it.  One
 * handle must not be used from two host threads at once.  pipeline = NULL: the prepare launches go
 * to `stream` itself and the library keeps no state whatsoever.
 *
 * neg_ws: device [B,3] int32 scratch (holds the last step's negatives on return).  loss: device
 * [n_steps*B] when keep_all_losses, else [B] (last step).  model: GE_MODEL_COMPLEX; GE_MODEL_HOLE (real
 * table; for even d the call transforms it to the frequency domain on entry, runs the spectral steps
 * and transforms back on exit -- two O(N d^2) passes per call, so prefer long calls or keep the table
 * spectral); GE_MODEL_HOLE_SPECTRAL (table already spectral, no transforms); GE_MODEL_HOLE_DIRECT
 * (direct-correlation kernels on the real table, O(d^2) per triple).
 * ev_pairs (nullable, HOST array of 2*n_steps events from ge_event_create): the events ride on the
 * dispatch of kernel `ev_kernel` of every step (1 = gather+score+hinge+grad, 2 = row update; 0 =
 * the per-step sampler of the fallback path) and report that kernel's own begin/end -- the hook
 * bench.py uses to time one kernel; NULL entries skip a step. */
size_t ge_train_workspace_bytes(int64_t B, int32_t d);
int ge_train_pipeline_create(void** pipeline);
int ge_train_pipeline_reset(void* pipeline);
int ge_train_pipeline_destroy(void* pipeline); /* waits for the side stream, then frees */
int ge_train_steps(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T,
                   int64_t first_row, int64_t B, int64_t n_steps, const int32_t* id_to_type,
                   const int64_t* type_offsets, int32_t n_types, const int32_t* type_ids,
                   uint64_t seed, uint64_t global_step0, int32_t padded_size, int32_t mode,
                   float margin, float lr0, float decay_steps, float decay_rate, float max_norm,
                   int model, float* loss, int keep_all_losses, int32_t* neg_ws, void* workspace,
                   size_t workspace_bytes, void** ev_pairs, int ev_kernel, void* pipeline, void* stream);

/* --- the same loop for the --log_loss objective (holE.py:194-196, 206-220, minimised as holE.py:296), ComplEx:
 * per step the M = (1 + negative_ratio) * B triples are the positives (label +1) followed by negative_ratio
 * corrupted batches (label -1; the k-th is ge_corrupt_batch with step = (global_step0 + s) * negative_ratio + k),
 * loss_i = log(1 + exp(-y_i s_i)) + l2 * sum(table^2) / 2, and table <- table * (1 - lr M l2) - lr * sparse gradient.
 * Negatives and the row-sorted slot index (slot = 3 * triple + {h, t, r}) are prepared ahead as in
 * ge_train_steps; the dense decay is carried as one scalar between steps (rows are read scaled, the sparse
 * update is divided by the new scale) and the table is materialised once, before the call returns -- instead of
 * two passes over the whole table per step; sum(table^2) is formed only for the steps whose loss is kept.
 * loss: [n_steps * M] with keep_all_losses, else [M] (last step), in the reference's concat order.
 * neg_ws: [negative_ratio, B, 3] int32 (last step's negatives on return).  workspace >=
 * ge_train_logloss_workspace_bytes(B, negative_ratio, d), 256-B aligned.  pipeline: as for ge_train_steps
 * (a handle used for the hinge loop restarts when it meets this one and vice versa). */
size_t ge_train_logloss_workspace_bytes(int64_t B, int32_t negative_ratio, int32_t d);
int ge_train_steps_logloss(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T, int64_t first_row,
                           int64_t B, int64_t n_steps, const int32_t* id_to_type, const int64_t* type_offsets,
                           int32_t n_types, const int32_t* type_ids, uint64_t seed, uint64_t global_step0,
                           int32_t padded_size, int32_t mode, int32_t negative_ratio, float l2, float lr0,
                           float decay_steps, float decay_rate, float max_norm, float* loss, int keep_all_losses,
                           int32_t* neg_ws, void* workspace, size_t workspace_bytes, void* pipeline, void* stream);

/* The prepare launch on its own: the records of n_steps consecutive steps, as ge_train_steps builds
 * them, into `out` (ge_train_prepare_bytes(B, n_steps) bytes: n_steps * layout[0] int32 words of records,
 * then -- for B > 4096 -- the key arrays of the multi-workgroup sort).  ge_train_prepared_layout fills out8 =
 * {words per step record, tiles per step, pairs per tile S, offset of slot_item[6B],
 * offset of tile 0, words per tile, offset of items inside a tile, offset of islots}.
 * A step record is  neg[3B] | slot_item[6B] | tiles { n_items, pad | items[4S][2] = {table row,
 * count | multi << 30} | islots[4S][16] = the IndexedSlices slots (6*pair + k) an item sums, -1 padded }.
 * The step's (row, slot) keys are sorted as one sequence of tiles * 4S positions; tile t lists the items that
 * start in positions [t*4S, (t+1)*4S).
 * direct = 1: a row with exactly one gradient slot in the step is not queued as an
 * item; its slot is tagged -2 in slot_item and the pair that produces it updates the table row. */
size_t ge_train_prepare_bytes(int64_t B, int64_t n_steps);
int ge_train_prepared_layout(int64_t B, int64_t* out8);
int ge_train_prepare_steps(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t n_steps,
                           const int32_t* id_to_type, int64_t N, const int64_t* type_offsets, int32_t n_types,
                           const int32_t* type_ids, uint64_t seed, uint64_t global_step0, int32_t padded_size,
                           int32_t mode, int direct, int32_t* out, size_t out_bytes, void* stream);

/* --- the row-sharded training step (holE.py:287-296 on a table mod-sharded over G processes: owner = id % G,
 * local row = id / G, R = ceil(N / G); the reference itself is single-device).  The host
 * (graphembeddings_amd/sharded.py) moves rows and gradient sums with torch.distributed all-to-alls; these entry
 * points are everything else.  All buffers device, caller-owned; nothing synchronises.
 *
 * ge_shard_plan -- requester side, for S steps of B pairs at once (pos, neg [S,B,3] GLOBAL ids; neg differs from
 *   pos in at most one of head / tail, as ge_corrupt_batch produces; other pairs count as invalid):
 *   records [S, ge_train_prepared_layout(B)[0]]: the step's work items in the native loop's record format, over
 *     this rank's rows (read and updated in place) and, behind them, the other owners' rows in staging order
 *     (an item row R + u = row u of the gradient-sum buffer); slot_item tags: -2 sole slot of an own row, -3 - u
 *     sole slot of staged row u (the producing pair stores the row straight into the gradient-sum buffer);
 *   pos_src [S,B,3], neg_src [S,B]: where each pair reads its rows (own row, or R + u; neg_src = source << 1 |
 *     column of the corrupted entity, -1 when it has no row of its own; pos_src -1 = invalid pair);
 *   req_row [S,4*Bt] (Bt = B rounded up to the record's tile size: layout[1] * layout[2]): entry u = staged row u's
 *     index in its owner's shard, grouped by owner; counts [S,G]: rows requested from each owner.
 *   peer_mapped = 1 (experiment, G <= 8): pos_src / neg_src name another owner's row by R * (1 + owner) + local row
 *     instead of R + u -- for ge_shard_grad with peer_shards (HOST array of the G shards' device addresses, mapped
 *     into this process; entry `rank` unused), which reads those rows in place and takes no staging buffer. */
size_t ge_shard_plan_workspace_bytes(int64_t B, int64_t S);
int ge_shard_plan(const int32_t* pos, const int32_t* neg, int64_t S, int64_t B, int64_t N, int32_t G, int32_t rank,
                  int32_t* records, int32_t* pos_src, int32_t* neg_src, int32_t* req_row, int32_t* counts,
                  void* workspace, size_t workspace_bytes, int32_t peer_mapped, void* stream);
/* One step's fused gather -> clip -> score -> sigmoid -> hinge -> gradient rows on two row stores: the shard
 * (rows < R, in place; sole-slot rows are updated here) and `staged` [n_staged,d], the rows fetched from the
 * other owners.  gsum [n_staged,d] receives the gradient rows tagged -3 - u.  model: GE_MODEL_COMPLEX or
 * GE_MODEL_HOLE_SPECTRAL.  loss [B], grad_idx [6B], grad_val [6B,d] as for ge_hinge_grad. */
int ge_shard_grad(float* shard, int64_t rows_local, int32_t d, const float* staged, int64_t n_staged, const int32_t* pos_src,
                  const int32_t* neg_src, const int32_t* record, int64_t B, int64_t N, int32_t G, float margin, float lr,
                  float max_norm, int model, float* loss, int32_t* grad_idx, float* grad_val, float* gsum,
                  const float* const* peer_shards, void* stream);
/* The step's work items: own rows get ONE read-modify-write each, staged rows their sum in gsum (which must be
 * zero beforehand: rows with more than 16 slots combine atomically). */
int ge_shard_apply(float* shard, int64_t rows_local, int32_t d, const int32_t* record, int64_t B, int64_t N, int32_t G,
                   const int32_t* grad_idx, const float* grad_val, float* gsum, void* stream);
/* Owner side: req_all = the chunk's received request lists (rows of this shard) in (step, peer) order, step s =
 * [req_start[s], req_start[s+1]) (req_start: DEVICE int64 [S+1]); cap >= the longest per-step list.  records
 * [S, ge_shard_owner_record_words(cap)]: work items that add row j of the step's receive buffer to shard row
 * req[j] -- one read-modify-write per distinct row, fixed order. */
int64_t ge_shard_owner_record_words(int64_t cap);
size_t ge_shard_owner_workspace_bytes(int64_t cap, int64_t S);
int ge_shard_owner_plan(const int32_t* req_all, const int64_t* req_start, int64_t S, int64_t cap, int64_t rows_local,
                        int32_t* records, void* workspace, size_t workspace_bytes, void* stream);
int ge_shard_owner_apply(float* shard, int64_t rows_local, int32_t d, const int32_t* record, int64_t cap, const float* recv,
                         void* stream);

/* --- thin wrappers over hipEvent_t so a ctypes host can time kernels on the launch stream. */
int ge_event_create(void** ev);
int ge_event_destroy(void* ev);
int ge_event_record(void* ev, void* stream);
int ge_event_synchronize(void* ev);
int ge_event_elapsed_ms(void* start, void* stop, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* GE_HIP_H */
