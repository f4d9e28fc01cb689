// ge_capi.hip -- the extern "C" boundary declared in include/ge_hip.h.
// Argument checks happen here, on the host, before any launch: a kernel that faults can take
// the whole node down, so shapes, alignment and workspace sizes are validated up front.
#include "ge_common.h"

namespace ge {
int complex_score_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, float, int, float*, hipStream_t, int spectral = 0, float label = 0.f, float l2 = 0.f, const float* table_sumsq = nullptr, int64_t ld = 0);
int complex_hinge_loss_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float*, float*, hipStream_t, int spectral = 0);
int complex_hinge_grad_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float, float*, int32_t*, float*, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr, const int32_t* slot_item = nullptr, float* table_rw = nullptr, int spectral = 0, const int32_t* order = nullptr);
int complex_max_dim();
int select_rows_launch(const int32_t*, int64_t, int64_t, uint64_t, uint64_t, int32_t*, hipStream_t);
int mean_pocket_launch(const float*, int64_t, float*, float*, int32_t*, hipStream_t);
int copy_if_launch(const float*, float*, int64_t, const int32_t*, hipStream_t);
int rank_max_dim();
int complex_rank_1vK_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, const int32_t*, const int32_t*, int64_t, float, int, const int32_t*, const uint16_t*, int32_t*, int32_t*, float*, float*, int, const void*, hipStream_t, int vs_loss = 0);
int64_t rank_planes_bytes(int64_t, int32_t, int64_t);
int known_cells_launch(int, const int64_t*, const int64_t*, int64_t, const int64_t*, const int64_t*, int64_t, const int64_t*, int64_t, int64_t, int32_t*, int32_t*, uint16_t*, hipStream_t);
int rank_planes_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, float, int, void*, hipStream_t);
int hole_spectral_launch(float*, int64_t, int32_t, int, hipStream_t);
int hole_score_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, float, int, float*, hipStream_t);
int hole_hinge_loss_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float*, float*, hipStream_t);
int hole_hinge_grad_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float, float*, int32_t*, float*, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr);
int hole_max_dim();
int scatter_add_rows_launch(float*, int64_t, int32_t, const int32_t*, const float*, int64_t, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr);
int gather_rows_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, float*, hipStream_t);
int corrupt_batch_launch(const int32_t*, int64_t, const int32_t*, int64_t, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, int32_t*, hipStream_t);
int complex_score_1vK_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, const int32_t*, int64_t, float, int, int, float*, hipStream_t);
int bernoulli_corrupt_launch(const int32_t*, int64_t, const int64_t*, const int32_t*, const int64_t*, const int32_t*, int64_t, const uint32_t*, int32_t, int32_t, int32_t, uint64_t, uint64_t, int32_t*, hipStream_t);
int complex_logloss_grad_launch(const float*, int64_t, int32_t, const int32_t*, const float*, int64_t, float, float, float, const float*, float*, int32_t*, float*, hipStream_t, const int32_t* negs = nullptr, int64_t B = 0, float row_scale = 1.f, float neg_lr_eff = 0.f, hipEvent_t = nullptr, hipEvent_t = nullptr);
int table_sumsq_launch(const float*, int64_t, float*, hipStream_t);
int table_scale_launch(float*, int64_t, float, hipStream_t);
size_t hinge_ws_bytes(int64_t, int32_t);
size_t train_ws_bytes(int64_t, int32_t);
int train_steps_run(float*, int64_t, int32_t, const int32_t*, int64_t, int64_t, int64_t, int64_t, const int32_t*, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, float, float, float, float, float, int, float*, int, int32_t*, void*, size_t, void**, int, void*, hipStream_t);
int train_prepare_run(const int32_t*, int64_t, int64_t, int64_t, int64_t, const int32_t*, int64_t, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, int, int32_t*, hipStream_t);
void train_prepared_layout(int64_t, int64_t*);
size_t train_prepare_bytes(int64_t, int64_t);
size_t train_logloss_ws_bytes(int64_t, int32_t, int32_t);
int train_logloss_run(float*, int64_t, int32_t, const int32_t*, int64_t, int64_t, int64_t, int64_t, const int32_t*, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, int32_t, float, float, float, float, float, float*, int, int32_t*, void*, size_t, void*, hipStream_t);
size_t shard_plan_scratch_bytes(int64_t, int64_t);
int shard_plan_launch(const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int32_t, int32_t, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, void*, int, hipStream_t);
int shard_grad_launch(float*, int32_t, const float*, const int32_t*, const int32_t*, const int32_t*, int32_t, int64_t, float, float, float, int, float*, int32_t*, float*, float*, const float* const*, int, hipStream_t, hipEvent_t, hipEvent_t);
int shard_apply_launch(float*, int32_t, const int32_t*, int64_t, const int32_t*, const float*, int32_t, float*, hipStream_t, hipEvent_t, hipEvent_t);
int64_t shard_owner_record_words(int64_t);
size_t shard_owner_scratch_bytes(int64_t, int64_t);
int shard_owner_plan_launch(const int32_t*, const int64_t*, int64_t, int64_t, int32_t, int32_t*, void*, hipStream_t);
int shard_owner_apply_launch(float*, int32_t, const int32_t*, int64_t, const float*, hipStream_t);
int pipeline_create(void**);
int pipeline_reset(void*);
int pipeline_destroy(void*);
}  // namespace ge

using namespace ge;

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline bool ok_table(const void* t, int64_t N, int32_t d) { return t != nullptr && N > 0 && d > 0; }
static inline bool max_norm_ok(float m) { return m > 0.f; }

extern "C" {

int ge_version(void) { return GE_VERSION; }

size_t ge_validation_workspace_bytes(int64_t B) {
  return B <= 0 ? 0 : ((size_t)B * (3 + 3 + 1) * 4 + 16 + 255) / 256 * 256;
}

int ge_max_dim(void) { return complex_max_dim(); }

int ge_complex_score(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                     float max_norm, int apply_sigmoid, float* out, void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (B > 0 && (!triples || !out)) return GE_EINVAL;
  return complex_score_launch(table, N, d, triples, B, max_norm, apply_sigmoid, out, (hipStream_t)stream);
}

int ge_complex_score_strided(const float* table, int64_t N, int32_t d, int64_t ld, const int32_t* triples, int64_t B,
                             float max_norm, int apply_sigmoid, float* out, void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm) || ld < d) return GE_EINVAL;
  if (B > 0 && (!triples || !out)) return GE_EINVAL;
  return complex_score_launch(table, N, d, triples, B, max_norm, apply_sigmoid, out, (hipStream_t)stream, 0, 0.f, 0.f, nullptr, ld);
}

int ge_complex_logloss(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B, float label,
                       float l2, float max_norm, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (B == 0) return 0;
  if (!triples || !out || !workspace) return GE_EINVAL;
  if (workspace_bytes < 256) return GE_ENOMEM;
  float* sumsq = reinterpret_cast<float*>(workspace);
  int rc = table_sumsq_launch(table, N * (int64_t)d, sumsq, (hipStream_t)stream);
  if (rc) return rc;
  return complex_score_launch(table, N, d, triples, B, max_norm, 2, out, (hipStream_t)stream, 0, label, l2, sumsq);
}

int ge_hole_score(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                  float max_norm, int apply_sigmoid, float* out, void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (B > 0 && (!triples || !out)) return GE_EINVAL;
  return hole_score_launch(table, N, d, triples, B, max_norm, apply_sigmoid, out, (hipStream_t)stream);
}

int ge_hole_to_spectral(float* table, int64_t N, int32_t d, void* stream) {
  if (!ok_table(table, N, d)) return GE_EINVAL;
  return hole_spectral_launch(table, N, d, 0, (hipStream_t)stream);
}

int ge_hole_from_spectral(float* table, int64_t N, int32_t d, void* stream) {
  if (!ok_table(table, N, d)) return GE_EINVAL;
  return hole_spectral_launch(table, N, d, 1, (hipStream_t)stream);
}

int ge_hole_spectral_score(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                           float max_norm, int apply_sigmoid, float* out, void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (B > 0 && (!triples || !out)) return GE_EINVAL;
  return complex_score_launch(table, N, d, triples, B, max_norm, apply_sigmoid, out, (hipStream_t)stream, 1);
}

int ge_hinge_loss(const float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                  int64_t B, float margin, float max_norm, int model, float* loss, float* sig_out,
                  void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm) || model < 0 || model > 2) return GE_EINVAL;
  if (B > 0 && (!pos || !neg || !loss)) return GE_EINVAL;
  if (model != GE_MODEL_HOLE)
    return complex_hinge_loss_launch(table, N, d, pos, neg, B, margin, max_norm, loss, sig_out, (hipStream_t)stream,
                                     model == GE_MODEL_HOLE_SPECTRAL);
  return hole_hinge_loss_launch(table, N, d, pos, neg, B, margin, max_norm, loss, sig_out, (hipStream_t)stream);
}

int ge_validation_tick(const float* table, int64_t N, int32_t d, const int32_t* valid, int64_t V, int64_t B,
                       const int32_t* id_to_type, const int64_t* type_offsets, int32_t n_types, const int32_t* type_ids,
                       uint64_t seed, uint64_t counter, int32_t padded_size, int32_t mode, float margin, float max_norm,
                       int model, void* workspace, size_t workspace_bytes, float* mean_out, float* best, float* pocket,
                       void* stream) {
  if (B <= 0 || V <= 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm) || model < 0 || model > 2) return GE_EINVAL;
  if (!valid || !id_to_type || !type_offsets || !type_ids || !workspace || !mean_out || !best) return GE_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return GE_EINVAL;
  if (workspace_bytes < ge_validation_workspace_bytes(B)) return GE_ENOMEM;
  hipStream_t st = (hipStream_t)stream;
  int32_t* pos = (int32_t*)workspace;                 // [B,3] | neg [B,3] | loss [B] | flag
  int32_t* neg = pos + 3 * B;
  float* loss = (float*)(neg + 3 * B);
  int32_t* flag = (int32_t*)(loss + B);
  int rc = select_rows_launch(valid, V, B, seed, counter, pos, st);
  if (rc) return rc;
  rc = corrupt_batch_launch(pos, B, id_to_type, N, type_offsets, n_types, type_ids, seed, counter, padded_size, mode, neg, st);
  if (rc) return rc;
  rc = ge_hinge_loss(table, N, d, pos, neg, B, margin, max_norm, model, loss, nullptr, stream);
  if (rc) return rc;
  rc = mean_pocket_launch(loss, B, mean_out, best, flag, st);
  if (rc || !pocket) return rc;
  return copy_if_launch(table, pocket, N * (int64_t)d, flag, st);
}

// the same tick for the --log_loss objective (holE.py:194-196, 206-220 on a validation batch): positives with
// label +1, K corrupted batches with label -1, every loss plus l2 * l2_loss(table)
size_t ge_validation_logloss_workspace_bytes(int64_t B, int32_t negative_ratio) {
  if (B <= 0 || negative_ratio < 1) return 0;
  return ((size_t)B * (3 + 3 + 1 + (size_t)negative_ratio) * 4 + 16 + 255) / 256 * 256 + 256;
}

int ge_validation_tick_logloss(const float* table, int64_t N, int32_t d, const int32_t* valid, int64_t V, int64_t B,
                               const int32_t* id_to_type, const int64_t* type_offsets, int32_t n_types,
                               const int32_t* type_ids, uint64_t seed, uint64_t counter, int32_t padded_size, int32_t mode,
                               int32_t negative_ratio, float l2, float max_norm, void* workspace, size_t workspace_bytes,
                               float* mean_out, float* best, float* pocket, void* stream) {
  if (B <= 0 || V <= 0 || negative_ratio < 1 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (!valid || !id_to_type || !type_offsets || !type_ids || !workspace || !mean_out || !best) return GE_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return GE_EINVAL;
  if (workspace_bytes < ge_validation_logloss_workspace_bytes(B, negative_ratio)) return GE_ENOMEM;
  hipStream_t st = (hipStream_t)stream;
  const int64_t K = negative_ratio, M = (1 + K) * B;
  float* sumsq = (float*)workspace;                        // [256 B] | pos [B,3] | neg [B,3] | loss [(1+K)B] | flag
  int32_t* pos = (int32_t*)((char*)workspace + 256);
  int32_t* neg = pos + 3 * B;
  float* loss = (float*)(neg + 3 * B);
  int32_t* flag = (int32_t*)(loss + M);
  int rc = select_rows_launch(valid, V, B, seed, counter, pos, st);
  if (rc) return rc;
  rc = table_sumsq_launch(table, N * (int64_t)d, sumsq, st);
  if (rc) return rc;
  rc = complex_score_launch(table, N, d, pos, B, max_norm, 2, loss, st, 0, 1.0f, l2, sumsq);
  if (rc) return rc;
  for (int64_t k = 0; k < K; ++k) {
    rc = corrupt_batch_launch(pos, B, id_to_type, N, type_offsets, n_types, type_ids, seed, counter * (uint64_t)K + (uint64_t)k,
                              padded_size, mode, neg, st);
    if (rc) return rc;
    rc = complex_score_launch(table, N, d, neg, B, max_norm, 2, loss + (1 + k) * B, st, 0, -1.0f, l2, sumsq);
    if (rc) return rc;
  }
  rc = mean_pocket_launch(loss, M, mean_out, best, flag, st);
  if (rc || !pocket) return rc;
  return copy_if_launch(table, pocket, N * (int64_t)d, flag, st);
}

int ge_hinge_grad(const float* rows, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                  int64_t B, float margin, float lr, float max_norm, int model, float* loss,
                  int32_t* grad_idx, float* grad_val, void* stream) {
  if (B < 0 || !ok_table(rows, N, d) || !max_norm_ok(max_norm) || model < 0 || model > 2) return GE_EINVAL;
  if (B > 0 && (!pos || !neg || !loss || !grad_idx || !grad_val)) return GE_EINVAL;
  if (model != GE_MODEL_HOLE)
    return complex_hinge_grad_launch(rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val, (hipStream_t)stream,
                                     nullptr, nullptr, nullptr, nullptr, model == GE_MODEL_HOLE_SPECTRAL);
  return hole_hinge_grad_launch(rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val, (hipStream_t)stream);
}

int ge_scatter_add_rows(float* table, int64_t N, int32_t d, const int32_t* idx, const float* val,
                        int64_t R, void* stream) {
  if (R < 0 || !ok_table(table, N, d)) return GE_EINVAL;
  if (R > 0 && (!idx || !val)) return GE_EINVAL;
  return scatter_add_rows_launch(table, N, d, idx, val, R, (hipStream_t)stream);
}

int ge_gather_rows(const float* table, int64_t N, int32_t d, const int32_t* idx, int64_t R, float* out,
                   void* stream) {
  if (R < 0 || !ok_table(table, N, d)) return GE_EINVAL;
  if (R > 0 && (!idx || !out)) return GE_EINVAL;
  return gather_rows_launch(table, N, d, idx, R, out, (hipStream_t)stream);
}

// workspace layout: [grad_idx: 6B int32, padded to 256 B][grad_val: 6B*d fp32]
size_t ge_hinge_step_workspace_bytes(int64_t B, int32_t d) {
  if (B <= 0 || d <= 0) return 0;
  return hinge_ws_bytes(B, d);
}

size_t ge_train_workspace_bytes(int64_t B, int32_t d) {
  if (B <= 0 || d <= 0) return 0;
  return train_ws_bytes(B, d);
}

static int hinge_step(float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg, int64_t B,
                      float margin, float lr, float max_norm, int model, float* loss, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (B < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (B == 0) return 0;
  if (!pos || !neg || !loss || !workspace) return GE_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return GE_EINVAL;
  if (workspace_bytes < ge_hinge_step_workspace_bytes(B, d)) return GE_ENOMEM;
  int32_t* gidx = reinterpret_cast<int32_t*>(workspace);
  float* gval = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align_up(sizeof(int32_t) * 6 * (size_t)B, 256));
  int rc = ge_hinge_grad(table, N, d, pos, neg, B, margin, lr, max_norm, model, loss, gidx, gval, stream);
  if (rc != 0) return rc;
  return scatter_add_rows_launch(table, N, d, gidx, gval, 6 * B, (hipStream_t)stream);
}

int ge_complex_hinge_step(float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                          int64_t B, float margin, float lr, float max_norm, float* loss, void* workspace,
                          size_t workspace_bytes, void* stream) {
  return hinge_step(table, N, d, pos, neg, B, margin, lr, max_norm, 0, loss, workspace, workspace_bytes, stream);
}

int ge_hole_hinge_step(float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                       int64_t B, float margin, float lr, float max_norm, float* loss, void* workspace,
                       size_t workspace_bytes, void* stream) {
  return hinge_step(table, N, d, pos, neg, B, margin, lr, max_norm, 1, loss, workspace, workspace_bytes, stream);
}

int ge_corrupt_batch(const int32_t* pos, int64_t B, const int32_t* id_to_type, int64_t N,
                     const int64_t* type_offsets, int32_t n_types, const int32_t* type_ids, uint64_t seed,
                     uint64_t step, int32_t padded_size, int32_t mode, int32_t* neg, void* stream) {
  if (B < 0 || N <= 0 || !id_to_type || !type_offsets || !type_ids) return GE_EINVAL;
  if (B > 0 && (!pos || !neg)) return GE_EINVAL;
  return corrupt_batch_launch(pos, B, id_to_type, N, type_offsets, n_types, type_ids, seed, step,
                              padded_size, mode, neg, (hipStream_t)stream);
}

int ge_bernoulli_corrupt_batch(const int32_t* pos, int64_t B, const int64_t* bh_key, const int32_t* bh_ent,
                               const int64_t* bt_key, const int32_t* bt_ent, int64_t n_known,
                               const uint32_t* tail_threshold, int32_t n_rel, int32_t ent_lo, int32_t n_ent,
                               uint64_t seed, uint64_t step, int32_t* neg, void* stream) {
  if (B < 0 || !tail_threshold) return GE_EINVAL;
  if (n_known > 0 && (!bh_key || !bh_ent || !bt_key || !bt_ent)) return GE_EINVAL;
  if (B > 0 && (!pos || !neg)) return GE_EINVAL;
  return bernoulli_corrupt_launch(pos, B, bh_key, bh_ent, bt_key, bt_ent, n_known, tail_threshold, n_rel,
                                  ent_lo, n_ent, seed, step, neg, (hipStream_t)stream);
}

int ge_complex_score_1vK(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B,
                         const int32_t* cand, int64_t K, float max_norm, int apply_sigmoid, int cand_is_head,
                         float* out, void* stream) {
  if (B < 0 || K < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm) || (d & 1)) return GE_EINVAL;
  if (B > 0 && K > 0 && (!hr || !cand || !out)) return GE_EINVAL;
  return complex_score_1vK_launch(table, N, d, hr, B, cand, K, max_norm, apply_sigmoid, cand_is_head, out,
                                  (hipStream_t)stream);
}

int ge_rank_max_dim(void) { return rank_max_dim(); }

int64_t ge_rank_planes_bytes(int64_t N, int32_t d, int64_t K) { return rank_planes_bytes(N, d, K); }

int ge_rank_planes(const float* table, int64_t N, int32_t d, const int32_t* cand, int64_t K, float max_norm, int model,
                   void* planes, void* stream) {
  if (K < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm) || !planes) return GE_EINVAL;
  if (model != GE_MODEL_COMPLEX && model != GE_MODEL_HOLE_SPECTRAL) return model == GE_MODEL_HOLE || model == GE_MODEL_HOLE_DIRECT ? GE_ENOTSUP : GE_EINVAL;
  if (K > 0 && !cand) return GE_EINVAL;
  return rank_planes_launch(table, N, d, cand, K, max_norm, model == GE_MODEL_HOLE_SPECTRAL, planes, (hipStream_t)stream);
}

int ge_rank_1vK_planes(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                       const int32_t* cand, int64_t K, float max_norm, int model, int cand_is_head, const int32_t* known_off,
                       const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before, float* true_loss,
                       float* scores_out, const void* planes, void* stream) {
  if (B < 0 || K < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (model != GE_MODEL_COMPLEX && model != GE_MODEL_HOLE_SPECTRAL) return model == GE_MODEL_HOLE || model == GE_MODEL_HOLE_DIRECT ? GE_ENOTSUP : GE_EINVAL;
  if (B > 0 && (!hr || !true_id || !n_before || !n_known_before)) return GE_EINVAL;
  if (B > 0 && K > 0 && !cand) return GE_EINVAL;
  if ((known_off == nullptr) != (known_rc == nullptr)) return GE_EINVAL;
  if (planes && (rank_planes_bytes(N, d, K) == 0 || reinterpret_cast<uintptr_t>(planes) % 256 != 0)) return GE_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(n_before, 0, sizeof(int32_t) * (size_t)B, st);
  if (e == hipSuccess) e = hipMemsetAsync(n_known_before, 0, sizeof(int32_t) * (size_t)B, st);
  if (e != hipSuccess) return (int)e;
  return complex_rank_1vK_launch(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc,
                                 n_before, n_known_before, true_loss, scores_out, model == GE_MODEL_HOLE_SPECTRAL, planes, st);
}

int ge_rank_1vK_vs_loss(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* ref_id,
                        const float* ref_loss, const int32_t* cand, int64_t K, float max_norm, int model, int cand_is_head,
                        const int32_t* known_off, const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before,
                        const void* planes, void* stream) {
  if (B < 0 || K < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (model != GE_MODEL_COMPLEX && model != GE_MODEL_HOLE_SPECTRAL) return model == GE_MODEL_HOLE || model == GE_MODEL_HOLE_DIRECT ? GE_ENOTSUP : GE_EINVAL;
  if (B > 0 && (!hr || !ref_id || !ref_loss || !n_before || !n_known_before)) return GE_EINVAL;
  if (B > 0 && K > 0 && !cand) return GE_EINVAL;
  if ((known_off == nullptr) != (known_rc == nullptr)) return GE_EINVAL;
  if (planes && (rank_planes_bytes(N, d, K) == 0 || reinterpret_cast<uintptr_t>(planes) % 256 != 0)) return GE_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(n_before, 0, sizeof(int32_t) * (size_t)B, st);
  if (e == hipSuccess) e = hipMemsetAsync(n_known_before, 0, sizeof(int32_t) * (size_t)B, st);
  if (e != hipSuccess) return (int)e;
  if (K == 0) return 0;
  // (the launchers take the losses through their true_loss argument, which this mode only reads)
  return complex_rank_1vK_launch(table, N, d, hr, B, ref_id, cand, K, max_norm, cand_is_head, known_off, known_rc, n_before,
                                 n_known_before, const_cast<float*>(ref_loss), nullptr, model == GE_MODEL_HOLE_SPECTRAL, planes, st,
                                 /*vs_loss=*/1);
}

int ge_rank_1vK(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                const int32_t* cand, int64_t K, float max_norm, int model, int cand_is_head, const int32_t* known_off,
                const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before, float* true_loss,
                float* scores_out, void* stream) {
  return ge_rank_1vK_planes(table, N, d, hr, B, true_id, cand, K, max_norm, model, cand_is_head, known_off, known_rc,
                            n_before, n_known_before, true_loss, scores_out, nullptr, stream);
}

int ge_known_cells(int pass, const int64_t* known_key, const int64_t* known_ent, int64_t M, const int64_t* fixed,
                   const int64_t* rel, int64_t B, const int64_t* pos_of, int64_t n_rows, int64_t n_cand, int32_t* tile_scratch,
                   int32_t* known_off, uint16_t* known_rc, void* stream) {
  if (pass < 0 || pass > 1 || M < 0 || B < 0 || n_rows <= 0 || n_cand <= 0 || !tile_scratch || !known_off) return GE_EINVAL;
  if (M > 0 && (!known_key || !known_ent)) return GE_EINVAL;
  if (B > 0 && (!fixed || !rel || !pos_of)) return GE_EINVAL;
  if (pass == 1 && !known_rc) return GE_EINVAL;
  return known_cells_launch(pass, known_key, known_ent, M, fixed, rel, B, pos_of, n_rows, n_cand, tile_scratch, known_off,
                            known_rc, (hipStream_t)stream);
}

int ge_complex_rank_1vK(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                        const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                        const uint16_t* known_rc, int32_t* n_before, int32_t* n_known_before, float* true_loss,
                        float* scores_out, void* stream) {
  return ge_rank_1vK(table, N, d, hr, B, true_id, cand, K, max_norm, GE_MODEL_COMPLEX, cand_is_head, known_off, known_rc,
                     n_before, n_known_before, true_loss, scores_out, stream);
}

int ge_train_steps(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T, int64_t first_row,
                   int64_t B, int64_t n_steps, const int32_t* id_to_type, const int64_t* type_offsets,
                   int32_t n_types, const int32_t* type_ids, uint64_t seed, uint64_t global_step0,
                   int32_t padded_size, int32_t mode, float margin, float lr0, float decay_steps,
                   float decay_rate, float max_norm, int model, float* loss, int keep_all_losses,
                   int32_t* neg_ws, void* workspace, size_t workspace_bytes, void** ev_pairs, int ev_kernel,
                   void* pipeline, void* stream) {
  if (B <= 0 || n_steps < 0 || T < B || first_row < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm))
    return GE_EINVAL;
  if (!triples || !id_to_type || !type_offsets || !type_ids || !loss || !neg_ws || !workspace) return GE_EINVAL;
  if ((model & ~GE_STEP_DETERMINISTIC) < 0 || (model & ~GE_STEP_DETERMINISTIC) > 3) return GE_EINVAL;
  if ((model & ~GE_STEP_DETERMINISTIC) == GE_MODEL_HOLE_SPECTRAL && (d & 1)) return GE_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return GE_EINVAL;
  if (workspace_bytes < ge_hinge_step_workspace_bytes(B, d)) return GE_ENOMEM;
  if (ev_pairs && (ev_kernel < 0 || ev_kernel > 2)) return GE_EINVAL;
  if (mode < 0 || mode > 3 || padded_size < 0 || n_types < 0) return GE_EINVAL;
  return train_steps_run(table, N, d, triples, T, first_row, B, n_steps, id_to_type, type_offsets, n_types,
                         type_ids, seed, global_step0, padded_size, mode, margin, lr0, decay_steps, decay_rate,
                         max_norm, model, loss, keep_all_losses, neg_ws, workspace, workspace_bytes, ev_pairs,
                         ev_kernel, pipeline, (hipStream_t)stream);
}

size_t ge_train_logloss_workspace_bytes(int64_t B, int32_t negative_ratio, int32_t d) {
  if (B <= 0 || negative_ratio <= 0 || d <= 0) return 0;
  return train_logloss_ws_bytes(B, negative_ratio, d);
}

int ge_train_steps_logloss(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T, int64_t first_row,
                           int64_t B, int64_t n_steps, const int32_t* id_to_type, const int64_t* type_offsets,
                           int32_t n_types, const int32_t* type_ids, uint64_t seed, uint64_t global_step0,
                           int32_t padded_size, int32_t mode, int32_t negative_ratio, float l2, float lr0,
                           float decay_steps, float decay_rate, float max_norm, float* loss, int keep_all_losses,
                           int32_t* neg_ws, void* workspace, size_t workspace_bytes, void* pipeline, void* stream) {
  if (B <= 0 || n_steps < 0 || T < B || first_row < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (negative_ratio <= 0 || negative_ratio > 1024 || (d & 1)) return GE_EINVAL;
  if (!triples || !id_to_type || !type_offsets || !type_ids || !loss || !neg_ws || !workspace) return GE_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return GE_EINVAL;
  if (mode < 0 || mode > 3 || padded_size < 0 || n_types < 0) return GE_EINVAL;
  if ((int64_t)(1 + negative_ratio) * B > ((int64_t)1 << 24)) return GE_ENOTSUP;
  return train_logloss_run(table, N, d, triples, T, first_row, B, n_steps, id_to_type, type_offsets, n_types, type_ids,
                           seed, global_step0, padded_size, mode, negative_ratio, l2, lr0, decay_steps, decay_rate,
                           max_norm, loss, keep_all_losses, neg_ws, workspace, workspace_bytes, pipeline,
                           (hipStream_t)stream);
}

int ge_train_pipeline_create(void** pipeline) { return pipeline ? pipeline_create(pipeline) : GE_EINVAL; }
int ge_train_pipeline_reset(void* pipeline) { return pipeline_reset(pipeline); }
int ge_train_pipeline_destroy(void* pipeline) { return pipeline_destroy(pipeline); }

int ge_train_prepared_layout(int64_t B, int64_t* out8) {
  if (B <= 0 || !out8) return GE_EINVAL;
  train_prepared_layout(B, out8);
  return 0;
}

size_t ge_train_prepare_bytes(int64_t B, int64_t n_steps) { return (B <= 0 || n_steps <= 0) ? 0 : train_prepare_bytes(B, n_steps); }

int ge_train_prepare_steps(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t n_steps,
                           const int32_t* id_to_type, int64_t N, const int64_t* type_offsets, int32_t n_types,
                           const int32_t* type_ids, uint64_t seed, uint64_t global_step0, int32_t padded_size,
                           int32_t mode, int direct, int32_t* out, size_t out_bytes, void* stream) {
  if (B <= 0 || n_steps < 0 || T < B || first_row < 0 || N <= 0) return GE_EINVAL;
  if (!triples || !id_to_type || !type_offsets || !type_ids || !out) return GE_EINVAL;
  if (mode < 0 || mode > 3 || padded_size < 0 || n_types < 0) return GE_EINVAL;
  if (out_bytes < train_prepare_bytes(B, n_steps)) return GE_ENOMEM;
  return train_prepare_run(triples, T, first_row, B, n_steps, id_to_type, N, type_offsets, n_types, type_ids, seed,
                           global_step0, padded_size, mode, direct, out, (hipStream_t)stream);
}

// ---------------------------------------------------------------- the row-sharded step (csrc/ge_shard.hip)
static inline bool shard_ok(int64_t N, int32_t G, int32_t rank) { return N > 0 && G >= 1 && G <= 64 && rank >= 0 && rank < G && N + 2 * ((N + G - 1) / G) < ((int64_t)1 << 31); }

size_t ge_shard_plan_workspace_bytes(int64_t B, int64_t S) { return (B <= 0 || S <= 0) ? 0 : shard_plan_scratch_bytes(B, S); }

int ge_shard_plan(const int32_t* pos, const int32_t* neg, int64_t S, int64_t B, int64_t N, int32_t G, int32_t rank,
                  int32_t* records, int32_t* pos_src, int32_t* neg_src, int32_t* req_row, int32_t* counts,
                  void* workspace, size_t workspace_bytes, int32_t peer_mapped, void* stream) {
  if (S < 0 || B <= 0 || B > ((int64_t)1 << 24) || !shard_ok(N, G, rank)) return GE_EINVAL;
  if (peer_mapped && (G > 8 || ((N + G - 1) / G) * (G + 1) >= ((int64_t)1 << 30))) return GE_EINVAL;
  if (S == 0) return 0;
  if (!pos || !neg || !records || !pos_src || !neg_src || !req_row || !counts || !workspace) return GE_EINVAL;
  if (workspace_bytes < shard_plan_scratch_bytes(B, S)) return GE_ENOMEM;
  return shard_plan_launch(pos, neg, S, B, N, G, rank, records, pos_src, neg_src, req_row, counts, workspace, peer_mapped, (hipStream_t)stream);
}

int ge_shard_grad(float* shard, int64_t rows_local, int32_t d, const float* staged, int64_t n_staged, const int32_t* pos_src,
                  const int32_t* neg_src, const int32_t* record, int64_t B, int64_t N, int32_t G, float margin, float lr,
                  float max_norm, int model, float* loss, int32_t* grad_idx, float* grad_val, float* gsum,
                  const float* const* peer_shards, void* stream) {
  if (B < 0 || !ok_table(shard, rows_local, d) || !max_norm_ok(max_norm) || !shard_ok(N, G, 0)) return GE_EINVAL;
  if (peer_shards && (staged || G > 8)) return GE_EINVAL;
  if (model != GE_MODEL_COMPLEX && model != GE_MODEL_HOLE_SPECTRAL) return GE_ENOTSUP;   // hole: keep the shard spectral
  if (B == 0) return 0;
  const int64_t R = (N + G - 1) / G;
  if (rows_local > R || n_staged < 0 || (n_staged > 0 && ((!staged && !peer_shards) || !gsum))) return GE_EINVAL;
  if (!pos_src || !neg_src || !record || !loss || !grad_idx || !grad_val) return GE_EINVAL;
  return shard_grad_launch(shard, d, staged, pos_src, neg_src, record, (int32_t)R, B, margin, lr, max_norm,
                           model == GE_MODEL_HOLE_SPECTRAL, loss, grad_idx, grad_val, gsum, peer_shards, G, (hipStream_t)stream,
                           nullptr, nullptr);
}

int ge_shard_apply(float* shard, int64_t rows_local, int32_t d, const int32_t* record, int64_t B, int64_t N, int32_t G,
                   const int32_t* grad_idx, const float* grad_val, float* gsum, void* stream) {
  if (B < 0 || !ok_table(shard, rows_local, d) || !shard_ok(N, G, 0)) return GE_EINVAL;
  if (B == 0) return 0;
  if (!record || !grad_idx || !grad_val) return GE_EINVAL;
  return shard_apply_launch(shard, d, record, B, grad_idx, grad_val, (int32_t)((N + G - 1) / G), gsum, (hipStream_t)stream,
                            nullptr, nullptr);
}

int64_t ge_shard_owner_record_words(int64_t cap) { return shard_owner_record_words(cap); }
size_t ge_shard_owner_workspace_bytes(int64_t cap, int64_t S) { return (cap <= 0 || S <= 0) ? 0 : shard_owner_scratch_bytes(cap, S); }

int ge_shard_owner_plan(const int32_t* req_all, const int64_t* req_start, int64_t S, int64_t cap, int64_t rows_local,
                        int32_t* records, void* workspace, size_t workspace_bytes, void* stream) {
  if (S < 0 || cap < 0 || rows_local <= 0 || rows_local >= ((int64_t)1 << 31) || cap >= ((int64_t)1 << 30)) return GE_EINVAL;
  if (S == 0 || cap == 0) return 0;
  if (!req_all || !req_start || !records || !workspace) return GE_EINVAL;
  if (workspace_bytes < shard_owner_scratch_bytes(cap, S)) return GE_ENOMEM;
  return shard_owner_plan_launch(req_all, req_start, S, cap, (int32_t)rows_local, records, workspace, (hipStream_t)stream);
}

int ge_shard_owner_apply(float* shard, int64_t rows_local, int32_t d, const int32_t* record, int64_t cap, const float* recv,
                         void* stream) {
  if (cap < 0 || !ok_table(shard, rows_local, d)) return GE_EINVAL;
  if (cap == 0) return 0;
  if (!record || !recv) return GE_EINVAL;
  return shard_owner_apply_launch(shard, d, record, cap, recv, (hipStream_t)stream);
}

// workspace: [sumsq: 256 B][grad_idx: 3M int32, 256-B padded][grad_val: 3M*d fp32]
size_t ge_logloss_step_workspace_bytes(int64_t M, int32_t d) {
  if (M <= 0 || d <= 0) return 0;
  return 256 + align_up(sizeof(int32_t) * 3 * (size_t)M, 256) + sizeof(float) * 3 * (size_t)M * (size_t)d;
}

int ge_complex_logloss_step(float* table, int64_t N, int32_t d, const int32_t* triples, const float* labels,
                            int64_t M, float lr, float l2, float max_norm, float* loss, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (M < 0 || !ok_table(table, N, d) || !max_norm_ok(max_norm)) return GE_EINVAL;
  if (M == 0) return 0;
  if (!triples || !labels || !loss || !workspace) return GE_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return GE_EINVAL;
  if (workspace_bytes < ge_logloss_step_workspace_bytes(M, d)) return GE_ENOMEM;
  hipStream_t st = (hipStream_t)stream;
  float* sumsq = reinterpret_cast<float*>(workspace);
  int32_t* gidx = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + 256);
  float* gval = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 256 + align_up(sizeof(int32_t) * 3 * (size_t)M, 256));
  int rc = table_sumsq_launch(table, N * (int64_t)d, sumsq, st);                 // l2_loss of the OLD table
  if (rc) return rc;
  rc = complex_logloss_grad_launch(table, N, d, triples, labels, M, lr, max_norm, l2, sumsq, loss, gidx, gval, st);
  if (rc) return rc;
  // new = old - lr * (sparse + M * l2 * old) = old * (1 - lr*M*l2) + (-lr * sparse)
  if (l2 != 0.f) {
    rc = table_scale_launch(table, N * (int64_t)d, 1.0f - lr * (float)M * l2, st);
    if (rc) return rc;
  }
  return scatter_add_rows_launch(table, N, d, gidx, gval, 3 * M, st);
}

int ge_event_create(void** ev) {
  if (!ev) return GE_EINVAL;
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess) return (int)rc;
  *ev = (void*)e;
  return 0;
}
int ge_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : GE_EINVAL; }
int ge_event_record(void* ev, void* stream) { return ev ? (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) : GE_EINVAL; }
int ge_event_synchronize(void* ev) { return ev ? (int)hipEventSynchronize((hipEvent_t)ev) : GE_EINVAL; }
int ge_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return GE_EINVAL;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}

}  // extern "C"
