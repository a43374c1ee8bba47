/*
 * sibrar_hip.h — C ABI of libsibrar_hip.so, the MI355X (gfx950) engine for the SiBraR SingleBranchNet hot path.
 *
 * Conventions:
 *   - plain C: raw DEVICE pointers (e.g. torch's tensor.data_ptr()), explicit sizes / leading dimensions in ELEMENTS,
 *     scalar hyper-parameters, and a `void* stream` that is a hipStream_t (NULL = default stream);
 *   - no ownership transfer: every buffer (parameters, gradients, activations, workspaces) is allocated and freed by
 *     the caller; kernels are stateless, asynchronous on the given stream and re-entrant per stream;
 *   - return value: 0 = ok, non-zero = error; sbr_last_error() returns the message of the calling thread's last error;
 *   - row-index arrays (`*_idx`, `rows`, `slots`) are int32 device arrays, entity ids are int64 (torch.long) as the
 *     reference's loaders produce them (data/dataloader.py:196-198); NULL index array = identity.
 *
 * The reference (Tigxy/SiBraR---Single-Branch-Recommender) has no native code and therefore no FFI: each entry point
 * below names the PyTorch call site(s) of the reference that it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 */
#ifndef SIBRAR_HIP_H
#define SIBRAR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* activation codes — modules/polylinear.py:5-10 (ACTIVATION_FN_MAP) */
#define SBR_ACT_NONE 0
#define SBR_ACT_RELU 1
#define SBR_ACT_TANH 2
#define SBR_ACT_SIGMOID 3
#define SBR_ACT_SELU 4

/* recommendation-loss kinds — train/rec_losses.py:116-119 (RecommenderSystemLossesEnum) */
#define SBR_LOSS_BCE 0
#define SBR_LOSS_BPR 1
#define SBR_LOSS_SAMPLED_SOFTMAX 2

const char* sbr_last_error(void);
int sbr_abi_version(void);

/* ---- dense products on the matrix cores (fp32 in / fp32 accumulate, v_mfma_f32_32x32x2_f32) -------------------------
 * mode 0 (NT): C[ci(m), n] = act(sum_k A[ai(m), k] * B[n, k] + bias[n])     nn.Linear forward — modules/polylinear.py:51,
 *              the modality projectors algorithms/sgd_alg.py:1342-1357 with the row gather of sgd_alg.py:1960-1974 fused
 *              in, and the all-pairs scorer einsum('be,ce->bc') algorithms/sgd_alg.py:2109 / eval/eval.py:217.
 * mode 1 (NN): C[m, n] = sum_k A[ai(m), k] * B[k, n]                        autograd of nn.Linear w.r.t. its input.
 * mode 2 (TN): C[m, n] += sum_k A[ak(k), m] * B[bk(k), n]                   autograd of nn.Linear w.r.t. its weight; split
 *              over k with float atomics, so C must be zero-initialised and accumulate_atomic must be 1.
 * a_idx / b_idx / c_idx: optional int32 row maps (NULL = identity). */
int sbr_gemm_f32(int mode, const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx,
                 const float* bias, float* C, long ldc, const int* c_idx, int M, int N, int K, int act,
                 int accumulate_atomic, void* stream);

/* NT (mode 0 of sbr_gemm_f32: nn.Linear forward with fused bias / activation / row gather / row scatter) for products with few
 * output tiles and a long K — the modality projectors at the reference's default batch of 256: K is split over workgroups, the
 * partial tiles are summed in a fixed order by a second kernel that applies bias, activation and the scatter.
 * sbr_gemm_nt_splitk_workspace() returns 0 for shapes that should go to sbr_gemm_f32 instead. */
long sbr_gemm_nt_splitk_workspace(int M, int N, int K);
int sbr_gemm_nt_splitk_f32(const float* A, long lda, const int* a_idx, const float* B, long ldb, const float* bias, float* C,
                           long ldc, const int* c_idx, int M, int N, int K, int act, void* workspace, long workspace_bytes,
                           void* stream);

/* The same fp32 products for WIDE layers on the bf16 matrix pipe (csrc/gemm_split_wide_f32.hip; new in ABI 3): N a multiple of 256,
 * K a multiple of 32 of at least 64 — modules/polylinear.py:51 with hidden widths 256 / 512, forward (mode 0, NT: W [N][K]) and input
 * gradient (mode 1, NN: W [K][N]): C[ci(m), :] = act(A[ai(m), :] x W (+ bias)). a_idx / c_idx / bias may be NULL. */
int sbr_gemm_split_wide_supported(long M, int N, int K);
int sbr_gemm_split_wide_f32(int mode, const float* A, long lda, const int* a_idx, const float* W, long ldw, const float* bias, float* C,
                            long ldc, const int* c_idx, long M, int N, int K, int act, void* stream);
/* TN with a deterministic split-K slab reducer (no atomics): C[m, n] = sum_k A[ak(k), m] * B[bk(k), n], C overwritten.
 * autograd of nn.Linear w.r.t. its weight (dW = dZ^T X[rows]). workspace: sbr_gemm_tn_f32_workspace(M, N, K) bytes. */
long sbr_gemm_tn_f32_workspace(int M, int N, int K);
/* 1 when sbr_gemm_tn_f32 / _slabs serve this shape with the bf16-split kernel (csrc/gemm_split_tn_f32.hip), 0: fp32 ring kernel */
int sbr_gemm_tn_split_supported(int M, int N, int K);
int sbr_gemm_tn_f32(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, float* C, long ldc,
                    int M, int N, int K, void* workspace, long workspace_bytes, void* stream);
/* The same product with the reduction deferred: sbr_gemm_tn_f32_slabs writes only the partial slabs (the workspace then belongs to
 * the product; *splits_out, a HOST int, receives the slab count) and sbr_splitk_reduce_multi sums the slabs of up to 8 products in
 * the same fixed order with ONE launch (a training step needs its weight gradients only at the optimizer). slabs / outs: HOST
 * arrays of device pointers; ldcs / Ms / Ns / splits: HOST arrays. */
int sbr_gemm_tn_f32_slabs(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx, int M, int N, int K,
                          void* workspace, long workspace_bytes, int* splits_out, void* stream);
int sbr_splitk_reduce_multi(int count, const void* const* slabs, const void* const* outs, const long* ldcs, const int* Ms, const int* Ns,
                            const int* splits, void* stream);
/* the same launch also finishes up to 8 pending column reductions (sbr_colred_finish: out[i] = sum of the replicas of entry i of
 * its workspace, replicas left zeroed): the two finishing launches at the end of a backward pass become one. */
int sbr_splitk_reduce_multi_fin(int count, const void* const* slabs, const void* const* outs, const long* ldcs, const int* Ms,
                                const int* Ns, const int* splits, int fin_count, const void* const* fin_workspaces,
                                const void* const* fin_outs, const int* fin_widths, void* stream);

/* Weights-resident variant for the shared MLP's own products (N = K = 128, no gathers): every wave keeps its half of the 128 x 128
 * weight in registers, only A streams through LDS (csrc/gemm_wres_f32.hip); bit-identical to sbr_gemm_f32 on the same operands.
 * mode 0 (NT): C = act(A W^T + bias) — nn.Linear forward, modules/polylinear.py:51,63-72; mode 1 (NN): C = A W — its autograd
 * w.r.t. the input. Y != NULL (mode 1): C = (A W) * act'(Y), the gradient at the pre-activation of the layer in front whose OUTPUT
 * is Y, and colsum_ws (17 * 128 doubles, contract of sbr_colsum; may be NULL) receives the pending column sums of C — that layer's
 * bias gradient, completed by sbr_colred_finish. */
int sbr_gemm_wres_supported(long M, int N, int K);
int sbr_gemm_wres_f32(int mode, const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, long M, int N,
                      int K, int act, const float* Y, long ldy, double* colsum_ws, void* stream);

/* The same products (same arguments, same epilogues) on the bf16 matrix pipe: both fp32 operands are split exactly into three bf16
 * numbers each (8 + 8 + 8 significand bits) and the six leading partial products are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 (csrc/gemm_split_f32.hip). The dropped terms are below 2^-23 of each product, i.e. the result carries
 * the error of an fp32 GEMM (summation order differs from sbr_gemm_f32, so it is not bit-identical to it); the kernel is bound by
 * HBM instead of the fp32 matrix pipe. Non-finite inputs give NaN. Replaces the same reference lines as sbr_gemm_wres_f32. */
int sbr_gemm_split_supported(long M, int N, int K);
int sbr_gemm_split_f32(int mode, const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, long M, int N,
                       int K, int act, const float* Y, long ldy, double* colsum_ws, void* stream);
/* mode 0 with the statistics epilogue and sbr_bn_finalize_stats in ONE launch (new): the last workgroup to arrive turns the pending
 * sums into save_mean / save_rstd of the M rows of C, updates running_mean / running_var (may both be NULL) and
 * num_batches_tracked (may be NULL); colsum_ws (zero on entry) and *arrive (one zeroed 64-bit word owned by the BatchNorm) are
 * left zeroed. */
int sbr_gemm_split_bnstats_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, long M, int N,
                               int K, int act, double* colsum_ws, void* arrive, float* running_mean, float* running_var,
                               long* num_batches_tracked, float* save_mean, float* save_rstd, float eps, float momentum, void* stream);
/* The dense modality projector on the same arithmetic — FeatureEmbedding's nn.Linear(F, C) over gathered feature rows
 * (algorithms/sgd_alg.py:1279-1396, forward of 1960-1974): C[ci(m), 0..127] = act(A[ai(m), 0..K-1] x W^T + bias), W [128][K],
 * N = 128, K = 128 j >= 256; a_idx (feature row of every slot), c_idx (row of the shared network's input), bias may be NULL.
 * The weight is walked in K chunks of 128 whose three bf16 planes are rebuilt in LDS (csrc/gemm_split_f32.hip). */
int sbr_gemm_split_proj_supported(long M, int N, int K);
int sbr_gemm_split_proj_f32(const float* A, long lda, const int* a_idx, const float* W, long ldw, const float* bias, float* C, long ldc,
                            const int* c_idx, long M, int N, int K, int act, void* stream);

/* HOST function (no device work): numpy's legacy `np.random.randint(0, high, size=n)` on a caller-owned MT19937 state
 * (key[624] + position from np.random.get_state(), advanced in place) — the draws of the default negative-sampling collate
 * (data/dataloader.py:154-198, np.random.choice(items_in_split, n) on the global RandomState). Bit-identical values and final
 * state for high - 1 < 2^32. */
int sbr_host_mt19937_randint(unsigned int* key, int* pos, long high, long n, long* out);

/* HOST function: out[q] = (items[q] in row users[q]) over a HOST copy of the sorted interaction CSR — the membership test of
 * the collate (data/dataloader.py:184-191) for its small redraw rounds, where a device round trip costs more than the search. */
int sbr_host_csr_contains(const long* indptr, const int* indices, const long* users, const long* items, long n,
                          unsigned char* out);
/* HOST: the layout step of the collate (data/dataloader.py:192-195): items[b, 0] = pos_items[b], items[b, 1 + j] = values[j * B + b]. */
int sbr_host_assemble_items(const long* pos_items, const long* values, long B, int n_neg, long* out_items);
/* the whole default collate (NegativeSamplingDataLoader._neg_sampling_collate_fn, data/dataloader.py:154-198) of a small batch in one
 * host call: draws every slot from the MT19937 stream (key, pos as for sbr_host_mt19937_randint), redraws the slots that hit one of
 * their user's interactions (sorted CSR) round by round in ascending slot order, writes [B, 1 + n_neg] items (column 0 = positive).
 * items_in_split NULL = identity; values, todo: scratch of B * n_neg longs. */
int sbr_host_recbole_collate(unsigned int* key, int* pos, const long* users, const long* pos_items, long B, int n_neg, long n_cand,
                             const long* items_in_split, const long* indptr, const int* indices, long* out_items, long* values,
                             long* todo);

/* Stable counting sort of the modality draw: the boolean-mask grouping of the flattened index tensor by sampled modality
 * (algorithms/sgd_alg.py:1934-1957). pos: int8 [R] modality position of every slot; segment m of slots_out
 * ([seg_offsets[m], seg_offsets[m+1]), HOST array of n_mod + 1 offsets, n_mod <= 8) receives the slots of modality m in
 * ascending order, its unused tail (capacity > count) is filled with the sentinel R. workspace: device scratch of
 * sbr_partition_slots_workspace(R) bytes (per-chunk histograms). */
long sbr_partition_slots_workspace(long R);
int sbr_partition_slots(const signed char* pos, long R, int n_mod, const int* seg_offsets, int* slots_out, void* workspace,
                        long workspace_bytes, void* stream);

/* ---- index plumbing ----------------------------------------------------------------------------------------------------
 * rows_out[j] = rowmap_seg(j)[ idx[slots[j] / k] ] for the concatenated per-modality slot lists (segment s covers
 * [seg_offsets[s], seg_offsets[s+1])): the id -> row lookup of Feature.__getitem__ (data/Feature.py:146) on the
 * repeat_interleave'd index vector of algorithms/sgd_alg.py:1944-1946. rowmaps is a HOST array of n_seg device pointers
 * (NULL entry = identity), rowmap_lens a HOST array with the number of ids each map covers (identity: the number of rows).
 * An id outside its map or mapped to -1 has no row in that feature's split: err_flag (device int, sticky) is set to 1 and
 * row 0 is used instead, so that no consumer indexes out of bounds; the host turns the flag into the reference's KeyError. */
int sbr_resolve_rows(const long* idx, int k, const int* slots, int n, int n_seg, const int* seg_offsets,
                     const int* const* rowmaps, const int* rowmap_lens, int* rows_out, int* err_flag, void* stream);

/* out[j] = 1 iff (rows[j], cols[j]) is a stored entry of the CSR matrix (sorted column indices): the `v in positives` test of
 * the negative-sampling collate, data/dataloader.py:184-191, for all slots of a round at once. */
int sbr_csr_contains(const long* indptr, const int* indices, const long* rows, const long* cols, long n, unsigned char* out,
                     void* stream);

/* dense interaction vectors of a batch of entities — InteractionRecDataset._get_interaction_vectors, data/dataset.py:306-319
 * (matrix[indices].toarray()), as consumed by DropoutNet's preference networks (algorithms/sgd_alg.py:1693-1725):
 * out[j, 0..dim) = CSR row ent[j] (data NULL: ones), all zeros for ent[j] < 0 (dropped preferences). */
int sbr_csr_rows_to_dense(const long* indptr, const int* indices, const float* data, const long* ent, long n, int dim, float* out,
                          long ldo, void* stream);

/* nn.Embedding forward — algorithms/sgd_alg.py:1331,1386: out[oi(j), :] = W[rows[j], :] */
int sbr_gather_rows(const float* W, long ldw, const int* rows, float* out, long ldo, const int* out_idx, long n, int D,
                    void* stream);
/* a plain embedding-lookup side in ONE launch (new: the fused training step): sbr_resolve_rows with one segment and k = 1
 * (data/Feature.py:146: id -> row through rowmap, or the id itself when rowmap is NULL and id < rowmap_len) + sbr_gather_rows;
 * rows_out [n] keeps the rows for sbr_scatter_add_rows. Ids without a row set *err_flag and read row 0. */
int sbr_lookup_rows_supported(const float* W, long ldw, const float* out, long ldo, int D);
int sbr_lookup_rows(const long* idx, long n, const int* rowmap, int rowmap_len, const float* W, long ldw, int* rows_out, float* out,
                    long ldo, int D, int* err_flag, void* stream);
/* its dense gradient: dW[rows[j], :] += dOut[ii(j), :] (dW zero-initialised by the caller) */
int sbr_scatter_add_rows(const float* dOut, long ldo, const int* in_idx, const int* rows, float* dW, long ldw, long n, int D,
                         void* stream);
/* the same dense gradient without float atomics, from row lists sorted by table row (new: the data-parallel exchange of
 * lookup gradients, SURVEY.md 8(e) — every rank must produce the same bits). rows_sorted[j] = table row of sorted position j,
 * perm[j] = its source row p: block p / blk (blocks are block_stride floats apart), row p % blk (rows ldo floats apart).
 * Equal rows are added in sorted order by one thread; dW[row, :] += sum. */
int sbr_scatter_add_rows_sorted(const float* dOut, long ldo, long blk, long block_stride, const long* perm,
                                const int* rows_sorted, float* dW, long ldw, long n, int D, void* stream);

/* nn.EmbeddingBag(mode='mean', padding_idx=pad) over padded tag lists — algorithms/sgd_alg.py:1336-1337, 1383-1386;
 * data/Feature.py:254-255. tags: [n_table_rows, T] int32. */
int sbr_bag_mean_fwd(const float* W, long ldw, const int* tags, int T, int pad, const int* rows, float* out, long ldo,
                     const int* out_idx, long n, int D, void* stream);
int sbr_bag_mean_bwd(const float* dOut, long ldo, const int* in_idx, const int* tags, int T, int pad, const int* rows,
                     float* dW, long ldw, long n, int D, void* stream);

/* Linear over the CSR "interactions" modality without densifying — replaces data/Feature.py:149-150 (.toarray()) +
 * algorithms/sgd_alg.py:1380 (x.float()) + modules/polylinear.py:51. Wt is the projector weight stored column-major
 * (row c of Wt = column c of the [C, n_cols] nn.Linear weight). vals may be NULL (all ones). */
int sbr_csr_project_fwd(const long* indptr, const int* indices, const float* vals, const float* Wt, long ldw,
                        const float* bias, const int* rows, float* out, long ldo, const int* out_idx, long n, int C, int act,
                        void* stream);
int sbr_csr_project_bwd(const long* indptr, const int* indices, const float* vals, const float* dZ, long ldz, const int* rows,
                        float* dWt, long ldw, long n, int C, void* stream);
/* the same gradient in gather form (new in ABI 3): the slot gradients are added up per entity into the workspace dZe
 * [n_entities, C] (overwritten; lde = C), and every feature column then sums the rows of the entities that have it — the forward
 * kernel over the TRANSPOSED feature matrix (t_indptr / t_indices / t_vals: its CSR form, [n_cols, n_entities]), no atomics on
 * dWt, fixed summation order. One float atomic per (slot, column) instead of one per (slot, nnz, column): Onion18 at batch 4096,
 * 1.25 ms -> see DESIGN.md. Needs C % 4 == 0, C <= 1024. (algorithms/sgd_alg.py:1380, the backward of that Linear.) */
int sbr_csr_project_bwd_gather(const long* t_indptr, const int* t_indices, const float* t_vals, const float* dZ, long ldz,
                               const int* dz_idx, const int* rows, long n, float* dZe, long lde, long n_entities, float* dWt,
                               long ldw, long n_cols, int C, void* stream);
/* dz_idx (may be NULL): slot j's gradient row is dZ[dz_idx[j], :]. sbr_bag_mean_bwd in gather form is this entry point with the
 * transpose of X[entity, tag] = 1 / (number of tags of the entity). */

/* dZ[j, :] = dY[ii(j), :] * act'(Y[ii(j), :]) — autograd of the activations of modules/polylinear.py:63-72 */
int sbr_act_grad_gather(const float* dY, const float* Y, long ld, const int* in_idx, float* dZ, long ldz, long n, int C,
                        int act, void* stream);
/* out[c] = sum_j X[j, c] (bias gradients). workspace: 17*C doubles (totals + 16 replicas that spread the per-block atomics);
 * it must be ZERO on first use and every call leaves it zeroed again (no memset per call). */
int sbr_colsum(const float* X, long ld, long n, int C, float* out, double* workspace, void* stream);

/* F.normalize(p=2, dim=-1, eps) — algorithms/sgd_alg.py:1873-1874 */
int sbr_l2norm_fwd(const float* X, float* Y, float* inv_norm, long n, int C, float eps, void* stream);
int sbr_l2norm_bwd(const float* dY, const float* Y, const float* inv_norm, float* dX, long n, int C, float eps, void* stream);

/* nn.Dropout(p) — algorithms/sgd_alg.py:1815, modules/polylinear.py:48; counter-based mask from (seed, element index),
 * the same call maps dY -> dX in the backward pass. */
int sbr_dropout(const float* X, float* Y, long total, float p, unsigned long long seed, void* stream);
/* the same with the seed in device memory (seed = seed_dev[0] + seed_offset, read when the kernel runs): a captured hipGraph
 * replays the launch while the host refreshes seed_dev[0] per step. sbr_dropout(seed) == sbr_dropout_dev with
 * seed_dev[0] + seed_offset == seed. */
int sbr_dropout_dev(const float* X, float* Y, long total, float p, const long* seed_dev, long seed_offset, void* stream);

/* aggregation over the k sampled modalities — algorithms/sgd_alg.py:27-31, 1861. mode 0 mean, 1 max. */
int sbr_aggregate_fwd(const float* E, float* out, unsigned char* argmax, long S, int k, int D, int mode, void* stream);
int sbr_aggregate_bwd(const float* dOut, const unsigned char* argmax, float* dE, long S, int k, int D, int mode, void* stream);

/* training scorer einsum('be,bce->bc') — algorithms/sgd_alg.py:2114 */
int sbr_score_dot_fwd(const float* U, const float* I, float* out, long B, int N, int D, void* stream);
int sbr_score_dot_bwd(const float* G, const float* U, const float* I, float* dU, float* dI, long B, int N, int D, void* stream);
/* SGDBaseline — algorithms/sgd_alg.py:110-119 */
int sbr_bias_score_fwd(const float* user_bias, const float* item_bias, const float* global_bias, const long* u, const long* i,
                       float* out, long B, int N, void* stream);
/* bias terms of SGDMatrixFactorization.combine_user_item_representations — algorithms/sgd_alg.py:186-194 (and SGDBaseline in
 * training): out[b, n] = base[b, n] + user_bias[u[b]] + item_bias[i[b, n]] + global_bias[0]; every term may be NULL; u NULL: row b,
 * i NULL: column n (all-pairs scoring against a gathered bias vector). Backward: the bias-table gradients are accumulated
 * (zero-initialise them), NULL ones skipped; d base = g. */
int sbr_bias_score_add_fwd(const float* user_bias, const float* item_bias, const float* global_bias, const long* u, const long* i,
                           const float* base, float* out, long B, int N, void* stream);
int sbr_bias_score_bwd(const float* g, const long* u, const long* i, float* d_user_bias, float* d_item_bias, float* d_global_bias,
                       long B, int N, void* stream);

/* ---- BatchNorm1d (+ fused activation) — modules/polylinear.py:61,68; algorithms/sgd_alg.py:1837 ---------------------------
 * ws: 34*D doubles of workspace (2*D totals + 16 replicas that spread the per-block atomics); ZERO on first use, every call
 * leaves the replicas zeroed again (forward and backward may share one workspace). running_mean/var/num_batches_tracked may be NULL. */
int sbr_bn_train_fwd(const float* X, float* Y, long n, int D, const float* weight, const float* bias, float* running_mean,
                     float* running_var, long* num_batches_tracked, float* save_mean, float* save_rstd, double* ws, float eps,
                     float momentum, int act, void* stream);
int sbr_bn_eval_fwd(const float* X, float* Y, long n, int D, const float* weight, const float* bias, const float* running_mean,
                    const float* running_var, float eps, int act, void* stream);
int sbr_bn_train_bwd(const float* dY, const float* Y, const float* X, float* dX, long n, int D, const float* weight,
                     const float* save_mean, const float* save_rstd, float* dWeight, float* dBias, double* ws, int act,
                     void* stream);

/* statistics half of sbr_bn_train_fwd: batch mean / rstd + running-statistics update, no normalising pass (the consumer
 * normalises on the fly: sbr_bn_score_fwd). */
int sbr_bn_train_stats(const float* X, long n, int D, float* running_mean, float* running_var, long* num_batches_tracked,
                       float* save_mean, float* save_rstd, double* ws, float eps, float momentum, void* stream);
/* the same statistics from sums that the producing GEMM left pending in `ws` (sbr_gemm_split_f32, mode 0, colsum_ws != NULL: per-column
 * sums and sums of squares of its output in the column-reduction replica layout): no pass over the BatchNorm's input at all */
int sbr_bn_finalize_stats(long n, int D, float* running_mean, float* running_var, long* num_batches_tracked, float* save_mean,
                          float* save_rstd, double* ws, float eps, float momentum, void* stream);

/* ---- trailing BatchNorm1d fused with the training scorer (one modality per slot) — algorithms/sgd_alg.py:1834-1837,
 * 1871-1877 (sb_net's trailing BatchNorm1d, no activation) followed by einsum('be,bce->bc') sgd_alg.py:2114, forward and
 * autograd: the normalised item representation [B*N, D] and its gradient are never stored (csrc/fused_tail.hip).
 *   fwd:        logits[b, n] = sum_d U[b, d] * ((Z[s, d] - mean[d]) * rstd[d] * weight[d] + bias[d]),  s = b*N + n
 *   bwd_stats:  dU[b, :] = sum_n G[b, n] * y[s, :];  ws[0..2D) = column sums of dy and dy * xhat, dy[s, :] = G[s] * U[b, :]
 *   bwd_apply:  dX = weight * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)); dWeight / dBias of the BatchNorm from ws;
 *               ws_colsum (17*D doubles, may be NULL): pending column sums of dX (bias gradient of the Linear in front of the
 *               BatchNorm), completed by sbr_colred_finish.
 * ws: the BatchNorm's 34*D-double workspace (contract of sbr_bn_train_fwd). D % 4 == 0, D <= 256, 16-byte aligned operands
 * (sbr_bn_score_supported). */
int sbr_bn_score_supported(int D);
int sbr_bn_score_fwd(const float* Z, const float* U, const float* mean, const float* rstd, const float* weight, const float* bias,
                     float* logits, long B, int N, int D, void* stream);
int sbr_bn_score_bwd_stats(const float* G, const float* U, const float* Z, float* dU, long B, int N, int D, const float* weight,
                           const float* bias, const float* save_mean, const float* save_rstd, double* ws, void* stream);
int sbr_bn_score_bwd_apply(const float* G, const float* U, const float* Z, float* dX, long B, int N, int D, const float* weight,
                           const float* save_mean, const float* save_rstd, const double* ws, float* dWeight, float* dBias,
                           double* ws_colsum, void* stream);
/* sbr_bn_score_fwd + sbr_rec_loss_fwd_bwd (train/rec_losses.py:43-113, upstream gradient 1) + sbr_bn_score_bwd_stats in ONE
 * launch (new: the fused training step): a user's N slot rows of Z stay in registers between the logits and the backward
 * statistics, logits / dlogits make no round trip between kernels. logits may be NULL; dlogits [B, N]; dU [B, D]; loss_out [1];
 * out3 (may be NULL) = (loss, loss, 0): the packed (total, rec, reg) scalars of a step without regularisation losses. ws: the
 * BatchNorm's workspace as for sbr_bn_score_bwd_stats (totals in ws[0 .. 2 D) afterwards, ready for sbr_bn_score_bwd_apply);
 * lws: sbr_bn_score_loss_workspace() bytes, zeroed ONCE by the caller and left zeroed by every call (calls that share it must not
 * overlap). kind / labels / scale / shift as for sbr_rec_loss_fwd_bwd. N <= D / 4 and N <= 16 (sbr_bn_score_loss_supported). */
int sbr_bn_score_loss_supported(int D, int N);
long sbr_bn_score_loss_workspace(void);
int sbr_bn_score_loss_fwd_bwd(const float* Z, const float* U, const float* save_mean, const float* save_rstd, const float* weight,
                              const float* bias, int kind, const double* labels, double scale, float shift, float* logits,
                              float* dlogits, float* dU, double* loss_out, double* out3, long B, int N, int D, double* ws, void* lws,
                              long lws_bytes, void* stream);

/* sbr_act_grad_gather that also accumulates the column sums of dZ (the bias gradient of its layer, modules/polylinear.py:51)
 * into a column-reduction workspace (17*C doubles, contract of sbr_colsum); sbr_colred_finish turns up to 8 pending
 * workspaces into float vectors (out[q][c] = sum over the replicas of workspaces[q], replicas re-zeroed) with one launch.
 * workspaces / outs: HOST arrays of device pointers, widths: HOST array. */
int sbr_act_grad_colsum_supported(int C);
int sbr_act_grad_gather_colsum(const float* dY, const float* Y, long ld, const int* in_idx, float* dZ, long ldz, long n, int C,
                               int act, double* ws, void* stream);
int sbr_colred_finish(int count, const void* const* workspaces, const void* const* outs, const int* widths, void* stream);

/* ---- losses ----------------------------------------------------------------------------------------------------------------
 * RecBinaryCrossEntropy / RecBayesianPersonalizedRankingLoss / RecSampledSoftmaxLoss .compute_loss —
 * train/rec_losses.py:43-58, 63-83, 88-113. labels: float64 [B, N] (unused for sampled softmax). scale = 1/count for
 * aggregator 'mean' (count = B*N bce, B*(N-1) bpr, B sampled softmax), 1 for 'sum'. shift = log(n_items / n_neg) for the
 * 'uniform' strategy of sampled softmax, else 0. loss_out: one double. */
int sbr_rec_loss_fwd(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                     double* loss_out, void* stream);
int sbr_rec_loss_bwd(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                     const void* grad_out, int grad_out_is_double, float* dlogits, void* stream);
/* both in one pass with upstream gradient 1 (the training step: total.backward() of train/trainer.py:213-221) */
int sbr_rec_loss_fwd_bwd(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                         double* loss_out, float* dlogits, void* stream);
/* the same in ONE launch (new: the fused training step; sbr_rec_loss_fwd_bwd zeroes loss_out with a launch of its own and sums the
 * block partial sums with double atomics): the partial sums go through ws — sbr_rec_loss_workspace(B) bytes, zeroed ONCE by the
 * caller, left zeroed by every call; calls sharing a workspace must not overlap — and are added in block order (the same bits on
 * every run). out3 (may be NULL): also writes (loss, loss, 0), the packed (total, rec, reg) scalars of a step without
 * regularisation losses (what sbr_pack_losses would produce). B >= 1. */
long sbr_rec_loss_workspace(long B);
int sbr_rec_loss_fwd_bwd_ws(int kind, const float* logits, const double* labels, long B, int N, double scale, float shift,
                            double* loss_out, float* dlogits, double* out3, void* ws, long ws_bytes, void* stream);

/* InfoNCE.forward — train/regularization_losses.py:14-43, called from algorithms/sgd_alg.py:1989 on e[..., 0, :] and
 * e[..., 1, :]. G groups of N rows, row stride ld; N <= sbr_infonce_max_n(). scale = 1/(G*N) for 'mean'. */
int sbr_infonce_max_n(void);
int sbr_infonce_fwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale, double* loss_out,
                    void* stream);
int sbr_infonce_bwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                    const float* grad_out, float* dA, float* dB, long ldg, void* stream);

/* out3 = (rec + reg, rec, reg), reg = w_a * reg_a + w_b * reg_b: the total loss of train/trainer.py:213-216 from the device-side
 * loss scalars (reg_a / reg_b may be NULL). */
int sbr_pack_losses(const double* rec, const double* reg_a, double w_a, const double* reg_b, double w_b, double* out3,
                    void* stream);

/* The same loss for groups of any size (in-batch contrast of B user rows, sgd_alg.py:1994-2002): the N x N logits go
 * through the fp32 MFMA GEMMs (sbr_gemm_f32 / sbr_gemm_tn_f32) instead of LDS. workspace: sbr_infonce_gemm_workspace(N, D)
 * bytes of device memory (logits, log-sum-exps, split-K slabs); groups are processed sequentially. Same argument meaning as
 * sbr_infonce_fwd / sbr_infonce_bwd; the backward recomputes the logits. */
long sbr_infonce_gemm_workspace(int N, int D);
int sbr_infonce_gemm_fwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                         double* loss_out, void* workspace, long workspace_bytes, void* stream);
int sbr_infonce_gemm_bwd(const float* A, const float* B, long ld, long G, int N, int D, float tau, double scale,
                         const float* grad_out, float* dA, float* dB, long ldg, void* workspace, long workspace_bytes,
                         void* stream);

/* ---- dense optimizers over one flat fp32 buffer — train/trainer.py:62-68, 222 ---------------------------------------------
 * kind 0 = torch.optim.AdamW, 1 = torch.optim.Adam; step is the 1-based step count. */
int sbr_adam_step(int kind, float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                  double wd, long step, void* stream);
/* optimizer.step(); optimizer.zero_grad() (train/trainer.py:221-222) in one launch: every gradient element is reset to +0 once it
 * has been consumed; elements that are +0 already are not written. */
int sbr_adam_step_zero_grad(int kind, float* p, float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                            double wd, long step, const double* copy_src, double* copy_dst, int copy_n, void* stream);
/* copy_n (0 .. 256) doubles copy_src -> copy_dst ride on the same launch: the loss scalars of a captured step, whose buffer the
 * next replay overwrites (copy_n = 0: no copy). */
/* The same dense-optimizer semantics for a [n_rows, D] lookup table, deferred row by row (train/trainer.py:62-68 updates every row
 * every step; a row without gradient can take its zero-gradient updates later, in order, bit-identically). mode 0: bring the rows
 * named by ids (int64 or int32, optionally through rowmap) up to step - 1 (before the forward pass reads them); mode 1: the same,
 * then apply `step` with their gradient rows, zero those gradient rows, record the step's scalars in sched[step]; mode 2: flush
 * every row to `step` (before any other reader). claim, last: int32 [n_rows * ceil(D / 64)] (one entry per 64-element sub-row, the
 * work of one wave), zero-initialised; sched: float2 [> step]. */
int sbr_adam_rows(int kind, int mode, float* p, float* g, float* m, float* v, long n_rows, int D, const long* ids64, const int* ids32,
                  const int* rowmap, long n, int* claim, int* last, void* sched, double lr, double b1, double b2, double eps,
                  double wd, long step, void* stream);
/* optimizer.step() + zero_grad() of a step whose flat buffers (n elements) hold ONE deferred table in [lo, hi): mode 1 of
 * sbr_adam_rows for the table's rows named by ids and sbr_adam_step_zero_grad for every other element, in one launch; rows of the
 * table that received no gradient are not touched — except the n_sweep sub-rows (64-element pieces of rows, in table order) from
 * sweep_lo on, cyclically, which are brought up to `step` beside the dense part's memory stream: a caller that sweeps 1 / W of the
 * table per step bounds every row's backlog — the length of a catch-up or flush replay — by W steps. n_sweep > 0 requires that
 * the step's catch-up (sbr_adam_rows mode 0 with the same ids and step) ran before: the sweep recognises the batch's rows by its
 * claims and leaves them to their update (new in ABI 3). */
int sbr_adam_step_rows(int kind, float* p, float* g, float* m, float* v, long n, long lo, long hi, int D, const long* ids64,
                       const int* ids32, const int* rowmap, long n_ids, int* claim, int* last, void* sched, double lr, double b1,
                       double b2, double eps, double wd, long step, long sweep_lo, long n_sweep, const double* copy_src,
                       double* copy_dst, int copy_n, void* stream);
int sbr_adagrad_step(float* p, const float* g, float* state_sum, long n, double lr, double eps, double wd, void* stream);

/* ---- full-catalogue evaluation — eval/eval.py:205-222 ------------------------------------------------------------------------
 * exclusion mask out[b, excl(u_b)] = -inf (eval/eval.py:219-220) from the CSR `exclude_data` (data/dataset.py:416-438) */
int sbr_mask_scores(float* scores, long ld, const long* u_idx, const long* excl_indptr, const int* excl_indices, long Bu,
                    void* stream);
/* the same when `scores` holds the item columns [item_offset, item_offset + n_cols) only (new: item-sharded evaluation) */
int sbr_mask_scores_shard(float* scores, long ld, const long* u_idx, const long* excl_indptr, const int* excl_indices, long Bu,
                          int item_offset, int n_cols, void* stream);
/* exact per-row top-k, sorted by (score desc, index asc) — torch.topk at eval/eval.py:320 and inside rmet.calculate */
int sbr_topk_rows(const float* scores, long ld, long Bu, int I, int k, float* out_val, int* out_idx, void* stream);
/* exact merge of W per-shard top-k lists of an item-sharded evaluation (new: the reference has no multi-GPU path):
 * vals / idxs: [W, Bu, k] (all-gathered; idx < 0 = empty slot, idx = global item index), W * k <= 256 -> [Bu, k] sorted by
 * (score desc, index asc). */
int sbr_merge_topk(const float* vals, const int* idxs, int W, long Bu, int k, float* out_val, int* out_idx, void* stream);
/* NDCG / recall / precision @ ks from top-k indices and CSR labels — eval/metrics.py:4-105. out: [3, n_ks, Bu]. */
int sbr_rank_metrics(const int* topk_idx, int kmax, const long* u_idx, const long* label_indptr, const int* label_indices,
                     long Bu, const int* ks, int n_ks, float* out, void* stream);
/* fused scorer: fp16 MFMA  U[Bu, D] x I[I_s, D]^T  with the exclusion mask and the running per-user top-k kept on chip; the
 * [Bu, I_s] score matrix is never written (BASELINE config 5). D in {64, 128, 256}, k <= 32. Output sorted by (score desc, item
 * index asc); indices are global (item_offset + column). u_idx: exclusion-CSR row of every scored row (NULL: identity).
 * excl_nnz: number of entries of `excl_indices` (an upper bound of the exclusions of the scored rows).
 * Exclusions reach the kernel as an EVENT STREAM built from the CSR rows (csrc/score_topk_f16_n.hip) in the caller-owned buffer
 * `events` (sbr_score_topk_f16_events_bytes(Bu, excl_nnz) bytes; NULL without exclusions): build_events != 0 builds it first (three
 * small launches), 0 means the buffer holds the stream an earlier call with the same (u_idx, CSR, item_offset, I, D) built — the
 * exclusion mask of an evaluation split (eval/eval.py:219: dataset.exclude_data) is the same for every evaluation of the split.
 * ABI 3: events / events_bytes / build_events are new, the workspace no longer holds the event stream. */
int sbr_score_topk_f16(const void* U_f16, const void* I_f16, int D, long Bu, int I, const long* u_idx,
                       const long* excl_indptr, const int* excl_indices, long excl_nnz, int item_offset, int k, float* out_val,
                       int* out_idx, void* workspace, long workspace_bytes, void* events, long events_bytes, int build_events,
                       void* stream);
/* bytes of `workspace` for the call above: per-user candidate buffers (4 KB per user: scratch written by the wave that owns the user
 * and read by the final-selection launch of the same call; contents need no initialisation) + their fill counts. */
long sbr_score_topk_f16_workspace(long Bu, int I, int k);
long sbr_score_topk_f16_events_bytes(long Bu, long excl_nnz);
/* ABI 4: which of the two fused scorers sbr_score_topk_f16 runs. 0 (default): the two-pass scorer (csrc/score_topk_f16_2p.hip: a pure
 * MFMA + group-maxima pass, then the ~3 % of the scores at or above each user's bound are recomputed group by group) for catalogues of
 * >= 8,192 items, the one-pass kernel (csrc/score_topk_f16_n.hip) below; 1: always one-pass; 2: two-pass or an error. Both return the same
 * lists bit for bit (same MFMA chain per score); the switch exists for tests and A/B timing. Returns the previous setting; a value
 * outside 0..2 only queries. Process-wide. */
int sbr_score_topk_f16_route(int route);
/* fp32 -> fp16 cast of an embedding matrix (row-major, contiguous) */
int sbr_cast_f32_to_f16(const float* X, void* Y_f16, long n, void* stream);

/* ---- native batch producer (csrc/producer.hip) --------------------------------------------------------------------------------
 * One C++ thread runs the host side of the training step ahead of the launch thread: the default collate of the reference
 * (data/dataloader.py:154-198: bit-exact draws from numpy's legacy MT19937 stream, `v in positives` on the resident interaction
 * CSR), the modality draw of every index slot (algorithms/sgd_alg.py:1904-1927, utilities/utils.py:60-90: PCG64 + numpy's Lemire
 * draws), the per-modality launch plan, and ONE packed host-to-device copy per batch into a ring of caller-allocated device
 * slots: [users | users[0] | items | items[0] | user draw | item draw | dropout seed], 16-byte aligned segments.
 * Generator states are handed in by sbr_producer_start and back by sbr_producer_stop. Exception to the "no ownership" rule:
 * the handle owns its pinned staging buffers, query scratch, stream and events. Batches must be consumed in order, one at a
 * time: sbr_producer_next -> sbr_producer_wait(stream) -> launches that read the slot -> sbr_producer_release(stream).
 * descriptor (32 longs): 0 slot, 1 B, 2 packed bytes, 3..8 segment offsets (users, items, user draw, item draw, seed, end),
 * 9 / 10 rows of the user / item draw, 11..18 user counts per modality (padded to the graph's bucket grid when pad != 0),
 * 19..26 item counts, 27 batch number. */
void* sbr_producer_create(int device, long B, int n_neg, long n_cand, const long* items_in_split, const long* h_indptr,
                          const int* h_indices, const long* d_indptr, const int* d_indices, long host_below, int pad, int n_slots,
                          long slot_bytes, void* const* slot_dev);
int sbr_producer_set_entity(void* handle, int which, int enabled, int n_mod, int k, int central);
int sbr_producer_start(void* handle, const long* rows_e, const long* cols_e, long n_inter, long first, long stride, long n_batches,
                       const unsigned int* mt_key, int mt_pos, const unsigned long long* pcg, long seed_base, long n_prepared);
int sbr_producer_next(void* handle, long* desc);
int sbr_producer_wait(void* handle, int slot, void* stream);
int sbr_producer_release(void* handle, int slot, void* stream);
int sbr_producer_stop(void* handle, unsigned int* mt_key, int* mt_pos, unsigned long long* pcg, long* out_counters);
int sbr_producer_destroy(void* handle);
/* the producer's generator replica and plan padding on their own (host only; pinned against numpy / the Python formulation by
 * tests/test_host_cpu.py). pcg: (state hi, state lo, inc hi, inc lo, has_uint32, uinteger) of numpy's PCG64, updated in place. */
int sbr_host_pcg64_modalities(unsigned long long* pcg, long n_slots, int n_mod, int k, int central, signed char* out, long* counts);
int sbr_host_pad_counts(long* counts, int n_mod, long R);

#ifdef __cplusplus
}
#endif
#endif
