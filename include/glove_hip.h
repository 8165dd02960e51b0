/*
 * glove_hip.h — C ABI of the MI355X (gfx950) GloVe training hot path.
 *
 * The reference (yxtay/glove-tensorflow) has no native code and no FFI: its hot path is the
 * TensorFlow graph that `model_fn` builds (reference src/models/estimator.py:13-56) and that
 * `session.run(train_op)` executes once per step.  This library replaces exactly that graph.
 * Each entry point cites the reference construct it stands in for; the binding a maintainer
 * of the reference would add (a ctypes stub called from `model_fn`'s place) is shown in
 * INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - return 0 on success, otherwise the hipError_t value (never throws, never aborts);
 *     GLOVE_E_* codes (< 0) report argument errors detected on the host before any launch;
 *   - every pointer inside the structs is a DEVICE pointer owned by the caller; the library
 *     allocates nothing, keeps no global state and is re-entrant for distinct streams;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*) and returns;
 *   - fp32 arithmetic, int32 ids (0 <= id < V), embedding size d % 4 == 0, rows 16-B aligned.
 */
#ifndef GLOVE_HIP_H
#define GLOVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLOVE_ABI_VERSION 13   /* 13: pruned — glove_plan_build_many, glove_shuffle_stream, glove_steps_rebuilt_f32 (+ glove_build_ring) are gone; glove_dense_grad_floats, glove_packed_entry_floats and glove_fused_step_bytes are what glove_dense_grad_layout returns and the macros GLOVE_PACKED_ENTRY_FLOATS / GLOVE_FUSED_STEP_BYTES; 12: glove_plan.r_chunk_hw / c_chunk_hw (the fused step forms on plans without records); 11: glove_plan.r_mark / c_mark (bitmaps of the batch's ids), the tagged form of glove_step(s)_adam_f32; 10: glove_hyper.optimizer / momentum / nesterov / rho, glove_step_sparse_f32 (SGD, RMSprop, Adamax, later Adadelta and Ftrl by their Keras names); 9: tagged step on step-tagged twinned tables (glove_tables.R_tag / C_tag, GLOVE_STEP_TAGGED); 8: epochs dealt from id-sorted master orders (glove_masters_build, glove_epoch_deal, glove_plan_build_sorted); plans whose pair fields live in their chunk records only; 7: chunk records start on 128-byte lines (capacity per record changed), glove_plan_build_many, glove_shuffle_stream; 6: glove_steps_rebuilt_f32; 5: record layout in 8-pair blocks; packing passes, loss partials */

#define GLOVE_E_BADARG   (-1)   /* null pointer / non-positive size / d % 4 != 0 */
#define GLOVE_E_WORKSPACE (-2)  /* workspace or plan storage too small */

/* Optimizer slots are generic: Adagrad uses slot1 = accumulator; Adam uses slot1 = m,
 * slot2 = v (reference src/models/train_utils.py:13-16 picks the Keras optimizer by name). */
typedef struct glove_tables {
    int32_t V;                  /* vocab size = lines of vocab.txt (reference estimator.py:31) */
    int32_t d;                  /* floats per table row: --embedding-size rounded up to a multiple of 4 (16-B rows) */
    int32_t V_row;              /* rows of R / br held by this process; 0 = V.  Smaller than V when the
                                 * row table is sharded over ranks (BASELINE config 5): row ids in the
                                 * plans are then LOCAL indices into the shard */
    int32_t d_model;            /* --embedding-size itself, 0 = d.  The activity-L2 coefficient is l2/d_model
                                 * (model_utils.py:8); columns d_model..d-1 of R, C and their slots are padding:
                                 * they must start at zero (Adagrad accumulator: any value) and every kernel
                                 * keeps them exactly zero, so dots, norms and gradients ignore them */
    float *R, *C;               /* row_embedding / col_embedding [V,d] (model_utils.py:31-34) */
    float *br, *bc;             /* row_bias / col_bias [V]         (model_utils.py:32-36) */
    float *s1_R, *s1_C, *s1_br, *s1_bc;   /* slot 1, same shapes */
    float *s2_R, *s2_C, *s2_br, *s2_bc;   /* slot 2 (Adam only; may be NULL for Adagrad) */
    /* scalars live on the device so that a captured hipGraph replays without host patching:
     *  [0] global_bias g (model_utils.py:39)  [1] slot1(g)  [2] slot2(g)  [3] which copy of twinned tables is current as a
     *  whole (the one-launch Adam step: 0 / 1)  [4], [5] Nadam's momentum cache (GLOVE_OPT_NADAM)  [6..7] reserved */
    float *scalars;             /* float[8] */
    /* global_step (estimator.py:45), int64[1].  glove_rowpass_f32 advances it by one (it is
     * the first kernel of a step and does not read it); the apply kernels read t = *step. */
    int64_t *step;
    /* Optional twin of the row table (NULL = none).  With it R has 2 x V_row rows and br 2 x V_row entries: row
     * V_row + u is a second copy of row u, and R_ver[u] (uint8[V_row]) says which copy is current (0 = row u).  The
     * fused twin step (GLOVE_STEP_FUSED_TWIN) writes a row's update into the copy that is not current and flips the
     * version once every pass of the step has read the old one, which saves the trip of the new row through a partial-row
     * slot.  EVERY other entry point expects the plain form — all versions 0, rows 0 .. V_row-1 current — which
     * glove_canonicalize_f32 restores. */
    uint8_t *R_ver;
    /* Optional step tags of BOTH tables (NULL = none; not together with R_ver).  With them R and br hold 2 x V_row rows /
     * entries and C and bc 2 x V — row V_side + u is the second copy of row u — and tag[u] (uint64) says where row u is: 0 =
     * copy 0, never written by a tagged step; otherwise (1 + global_step of the step that wrote it) << 1 | copy it wrote.
     * The tagged step (GLOVE_STEP_TAGGED) reads, during step t, the copy the tag names — unless the tag says "written in step
     * t", in which case it reads the OTHER copy, the row as it was when the step began — and writes a row's update into the
     * copy it did not read: every pair sees pre-step rows although rows are updated in the same launch, with no barrier and
     * no apply launch.  EVERY other entry point expects the plain form, which glove_canonicalize_f32 restores (current rows
     * copied home, tags cleared). */
    uint64_t *R_tag, *C_tag;
} glove_tables;

typedef struct glove_hyper {
    double beta1, beta2;        /* Adam 0.9 / 0.999.  Used rounded to fp32 everywhere, as Keras casts its
                                 * hyper-parameters to the variable dtype (1 - float(0.999) is 1.3e-5 off 0.001) */
    float l2_reg;               /* --l2-reg  (activity L2, model_utils.py:8,38) */
    float reg_mult;             /* m: times the regulariser list is counted (estimator.py:55) */
    float learning_rate;        /* --learning-rate */
    float epsilon;              /* Keras-legacy 1e-7 */
    float inv_batch;            /* 1 / (global batch size): RegressionHead SUM_OVER_BATCH_SIZE */
    /* which sides a call of glove_apply_adagrad_f32 / glove_dense_grad_f32 / glove_dense_ad*_f32 covers:
     * 0 or 3 = both, 1 = row side (R, br) only, 2 = col side (C, bc) only.  The once-per-step scalar work
     * (global bias, loss, clearing the tail) goes with the col side. */
    int32_t sides;
    /* loss head (the model, its regulariser and both optimizers are shared):
     *   GLOVE_HEAD_REGRESSION  RegressionHead(weight_column) of the GloVe estimator (estimator.py:48-56):
     *                          plan.w = glove_weight, plan.y = glove_value, loss_i = w (p - y)^2
     *   GLOVE_HEAD_LOGISTIC    MultiHead([BinaryClassHead(pos), BinaryClassHead(neg)], [1, neg_factor]) of
     *                          logistic_matrix_factorisation.py:50-54: plan.w = positive weight (`value`),
     *                          plan.y = negative weight (`neg_weight`), both heads see the same logit,
     *                          loss_i = w softplus(-p) + neg_factor y softplus(p)
     * both summed over the batch and divided by the batch size. */
    int32_t head;
    float neg_factor;           /* --neg-factor; read by the logistic head only */
    /* form of glove_step_adagrad_f32 / glove_steps_adagrad_f32 (same result bit for bit in every form):
     *   GLOVE_STEP_AUTO                the library chooses from the plan's id counts and the row width
     *   GLOVE_STEP_TWO_LAUNCH          passes (both sides) -> apply: every chunk's sums travel through a partial row
     *   GLOVE_STEP_FUSED_ONE_PASS      (tests and A/B comparisons only) both sides in one launch, every finished row through
     *                                  its partial-row slot, moved into the tables by the apply launch
     *   GLOVE_STEP_FUSED_THREE_LAUNCH  a run of chunks that holds ALL pairs of its id (most ids of a large vocabulary)
     *                                  applies Adagrad itself, accumulator in place: the row side first, its new rows
     *                                  into the chunks' partial-row slots (the col side still gathers the old ones), then
     *                                  the col side in a launch of its own, updating C and bc in place (nothing reads
     *                                  them any more), then the apply launch (moves the row side's rows into the table,
     *                                  handles the ids several lane groups share)
     *   GLOVE_STEP_FUSED_TWIN          the three-launch form on a twinned row table (glove_tables.R_ver): the row side
     *                                  writes its new rows into the other copy, the apply launch only flips versions
     *                                  (AUTO picks it whenever R_ver is set and the fused form pays)
     *                                  The fused forms read the id layout from the plan's chunk records — or, on a plan without
     *                                  records, from its run words (glove_plan.r_chunk_hw) beside the pair arrays: a lane group
     *                                  then stages the descriptors and pair fields of all its chunks in LDS in two trips up
     *                                  front (same pairs, same order: bit-identical).  A plan with neither takes two launches.
     *   GLOVE_STEP_TAGGED              for the latency-bound regime (the reference's default batch of 1,024 pairs: the two-launch
     *                                  form is two ramps, a boundary and four dependent memory round trips, nothing in them is
     *                                  bandwidth): on step-tagged twinned tables (glove_tables.R_tag) ONE launch forms the
     *                                  gradients AND applies Adagrad — the lane group that holds an id's first chunk does all
     *                                  of the id's chunks and writes the new row beside the old one — and a one-workgroup
     *                                  launch behind it does the once-per-step scalars (global bias, loss, global_step).  Ids of
     *                                  up to heavy_chunks chunks come out bit-identical to the two-launch form, the others
     *                                  within fp32 rounding of their sums' order.  Needs chunk records.  AUTO picks it for
     *                                  batches of at most 2,048 pairs when the tables carry tags. */
    int32_t step_form;
    /* which Keras optimizer glove_step_sparse_f32 applies (reference src/models/train_utils.py:13-16 resolves any Keras name
     * with `tf.keras.optimizers.get`, handing over the learning rate only: everything else keeps its Keras-legacy default):
     *   GLOVE_OPT_ADAGRAD  glove_step_adagrad_f32            GLOVE_OPT_ADAM  glove_step_adam_f32
     *   GLOVE_OPT_SGD      momentum == 0: var -= lr G on the touched rows; otherwise (slot1 = accumulator, zeros) the
     *                      deduplicated rows take accum = accum momentum - lr G; var += accum (nesterov: var += accum
     *                      momentum - lr G); untouched rows and their accumulators do not move
     *   GLOVE_OPT_RMSPROP  (slot1 = rms, zeros) the WHOLE rms slot decays by rho every step — the sparse path of the legacy
     *                      optimizer does, like its Adam —, (1 - rho) G^2 is added on the touched rows, which alone move:
     *                      var -= lr G / (sqrt(rms) + epsilon); momentum 0, not centered
     *   GLOVE_OPT_ADAMAX   (slot1 = m, slot2 = v, zeros) lazy: touched rows only: m = beta1 m + (1 - beta1) G;
     *                      v = max(beta2 v, |G|); var -= lr / (1 - beta1^t) m / (v + epsilon)
     *   GLOVE_OPT_ADADELTA (slot1 = accum_grad, slot2 = accum_var, zeros) touched rows only: a = rho a + (1 - rho) G^2;
     *                      u = sqrt(accum_var + epsilon) / sqrt(a + epsilon) G; var -= lr u; accum_var = rho accum_var +
     *                      (1 - rho) u^2   (rho: Keras default 0.95)
     *   GLOVE_OPT_FTRL     at its Keras defaults (learning_rate_power -0.5, l1 = l2 = l2_shrinkage = beta = 0; slot1 =
     *                      accumulator, 0.1, slot2 = linear, zeros) touched rows only: n = accumulator + G^2; linear += G -
     *                      (sqrt(n) - sqrt(accumulator)) / lr var; var = -linear / (sqrt(n) / lr) (0 where linear is 0);
     *                      accumulator = n
     *   GLOVE_OPT_NADAM    (slot1 = m, slot2 = v, zeros; scalars[4], scalars[5] = the momentum cache, ones, used alternately by
     *                      odd and even steps) m and v decay over the WHOLE variable and take (1 - beta) G, G^2 on the touched
     *                      rows, which alone move: with u_i = beta1 (1 - 0.5 0.96^(0.004 i)) and P_t = u_1 .. u_t,
     *                      var -= lr ((1 - u_t) G / (1 - P_t) + u_{t+1} m / (1 - P_{t+1})) / (sqrt(v / (1 - beta2^t)) + epsilon).
     *                      Needs G_flat (its bias segments carry the marks of the batch's ids, as in glove_step_adam_f32) */
    int32_t optimizer;
    float momentum;             /* SGD, Keras default 0 */
    int32_t nesterov;           /* SGD, Keras default 0 */
    float rho;                  /* RMSprop (Keras default 0.9), Adadelta (0.95) */
} glove_hyper;

#define GLOVE_OPT_ADAGRAD 0
#define GLOVE_OPT_SGD 1
#define GLOVE_OPT_RMSPROP 2
#define GLOVE_OPT_ADAMAX 3
#define GLOVE_OPT_ADAM 4
#define GLOVE_OPT_ADADELTA 5
#define GLOVE_OPT_FTRL 6
#define GLOVE_OPT_NADAM 7

#define GLOVE_HEAD_REGRESSION 0
#define GLOVE_HEAD_LOGISTIC 1

#define GLOVE_STEP_AUTO 0
#define GLOVE_STEP_TWO_LAUNCH 1
#define GLOVE_STEP_FUSED_ONE_PASS 2      /* test / comparison form: AUTO never picks it, the trainer never asks for it */
#define GLOVE_STEP_FUSED_THREE_LAUNCH 3
#define GLOVE_STEP_FUSED_TWIN 4
#define GLOVE_STEP_TAGGED 5

/*
 * The dedup index of ONE batch of co-occurrence nonzeros ("plan").  It replaces, per batch,
 * the `Unique` + `UnsortedSegmentSum` pair that Keras' OptimizerV2 runs on every sparse
 * gradient: pairs are stably sorted by row id (row side) and, independently,
 * by col id (col side); runs of equal ids are cut into chunks of at most
 * `chunk_cap` pairs; `*_uniq_slot[q]` is the first chunk of the q-th distinct id.
 * Built on the device by glove_plan_build; the arrays are plain device buffers so a caller
 * may keep one plan per batch of a static nonzero stream resident in HBM.
 */
typedef struct glove_plan {
    int64_t B;                  /* pairs in the batch */
    int32_t chunk_cap;          /* max pairs per chunk */
    int32_t cap_chunks;         /* capacity of the *_chunk_* arrays (>= chunks, <= B) */
    int32_t cap_uniq;           /* capacity of the *_uniq_slot arrays minus one */
    int32_t heavy_chunks;       /* ids with more chunks than this go to the `heavy` list (default 8) */
    int32_t cap_heavy;          /* capacity of `heavy` (>= 2 B / (heavy_chunks * chunk_cap) + 2) */
    int32_t V_row;              /* 0, or the number of rows of a row-table shard (glove_tables.V_row): glove_plan_build
                                 * then treats row ids outside [0, V_row) as id 0, like col ids outside [0, V) */
    int32_t *counts;            /* int32[8]: chunks_row, uniq_row, chunks_col, uniq_col, heavy,
                                 * ids outside [0,V) that were mapped to 0, 0, 0 */
    /* host copy of counts for plans whose build has completed (a resident plan of a static
     * stream): saves the kernels one dependent load.  -1 = unknown, read `counts` on the device.
     * [6] (host only) = the most chunks any one id of the batch has, or -1: sizes the grid that pre-sums the partial
     * rows of the heaviest ids in the fused step forms. */
    int32_t host_counts[8];
    /* row side: position k = k-th pair in (row id, original order) order */
    int32_t *r_partner;         /* [B] col id of pair k */
    float   *r_w;               /* [B] glove_weight */
    float   *r_y;               /* [B] glove_value  */
    int32_t *r_to_c;            /* [B] col-side position of pair k (inverse of c_perm).  r_to_c / c_perm link the two orders for
                                 * callers that want them (tests, diagnostics); no kernel of the library reads them: both NULL =
                                 * not computed (the per-step plans of a reshuffled epoch) */
    int32_t *r_chunk_id;        /* [cap_chunks]   row id of the chunk */
    int32_t *r_chunk_start;     /* [cap_chunks+1] first pair of the chunk; [chunks] = B */
    int32_t *r_uniq_slot;       /* [cap_uniq+1]   first chunk of the q-th distinct row id */
    int32_t *r_uniq_rec;        /* [cap_uniq][4]  {id, first chunk, chunks, pairs} of the q-th distinct
                                 * row id: everything the apply kernels need in one 16-B load */
    /* col side: position k = k-th pair in (col id, arrival order) order — sorted like the row side, independently of it */
    int32_t *c_partner;         /* [B] row id */
    int32_t *c_perm;            /* [B] row-sorted position of the pair (optional, see r_to_c) */
    float   *c_w;               /* [B] glove_weight, col-sorted order (the col side forms e_i by itself) */
    float   *c_y;               /* [B] glove_value,  col-sorted order */
    int32_t *c_chunk_id;
    int32_t *c_chunk_start;
    int32_t *c_uniq_slot;
    int32_t *c_uniq_rec;
    /* ids of the Zipf head (more than heavy_chunks chunks in this batch), both sides, in no
     * particular order: (side << 30) | q with side 0 = row, 1 = col.  The apply kernels give each of
     * them a whole workgroup that starts ahead of the per-lane-group work on the light ids. */
    int32_t *heavy;
    /* Optional per-chunk records (NULL = absent): chunk j of a side holds 4 + 3*capP dwords
     * (capP = chunk_cap rounded up to a multiple of 8: a trip reads up to 8 pair slots) laid out {id, pairs, position of the id among the side's distinct ids, (1 << 31 if it is the first chunk of its id) | chunks of the same id behind it | capP / 8 blocks of 8 pairs, each
     * partner[8] | w[8] | y[8]}, padding slots carrying weight 0 and a valid partner id; only the ceil(pairs / 8) blocks a chunk needs are written.  With them the pass kernel gets a
     * chunk's descriptor AND its pair fields in ONE memory round trip (contiguous 16-B loads) instead of two dependent
     * ones; what a chunk of n pairs needs is the prefix of 4 + 24 ceil(n / 8) dwords, so the bandwidth-bound fused forms
     * read the first block with the header and the rest only for longer chunks.  The layout is private to the library
     * (glove_plan_build / glove_plan_fill_records write it, the step kernels read it).
     * Filled by glove_plan_build when non-NULL, or later by glove_plan_fill_records for an exact-size
     * (compacted) plan.  In memory a record starts on a 128-byte line: header, block 0 and 16 bytes of padding fill the
     * first line, the other blocks follow packed; capacity: cap_chunks records of
     * rec_dwords = 32 * ceil((8 + 6 * (capP / 8 - 1)) / 8) int32 each (128 for chunk_cap 32). */
    int32_t *r_crec;
    int32_t *c_crec;
    /* Optional bitmaps of the batch's distinct ids (NULL = not computed): bit u of r_mark (ceil(V_row / 32) words; V when the row
     * table is whole) is set iff row id u occurs in the batch, c_mark (ceil(V / 32) words) likewise for col ids.  Every word is
     * written by every build.  The one-launch Keras-legacy Adam step reads them: its sweep over ALL rows (the legacy optimizer's
     * sparse path decays every row of the table every step, a11) leaves the batch's rows to the lane groups that apply them. */
    uint32_t *r_mark;
    uint32_t *c_mark;
    /* Optional per-chunk run words (NULL = not computed; [cap_chunks] each): (first chunk of its id) << 31 | chunks of the
     * same id behind this one — word 3 of the chunk's record header on its own.  With them the fused step forms run on a plan
     * WITHOUT records (its pair fields in r_partner / r_w / r_y ..., 12 B per pair, instead of a 128-byte line per chunk): the
     * form for batches indexed every step, whose records would be written once and read once.  Written by every builder. */
    uint32_t *r_chunk_hw;
    uint32_t *c_chunk_hw;
} glove_plan;

int glove_abi_version(void);

/* ---- index build: replaces tf.unique / UnsortedSegmentSum bookkeeping (a9) ------------- */
size_t glove_plan_workspace_bytes(int64_t B, int32_t V);
/* row/col/w/y: the batch as the input_fn delivers it (reference data_utils.py:4-26 after the
 * vocab lookup of estimator.py:26-28), in arbitrary order. Fills every array of `plan`. */
int glove_plan_build(const int32_t *row, const int32_t *col, const float *w, const float *y,
                     int64_t B, int32_t V, const glove_plan *plan,
                     void *ws, size_t ws_bytes, void *stream);

/* ---- epochs dealt from id-sorted master orders ----------------------------------------------------------------------------
 * The reference's input_fn reshuffles the file every epoch (data_utils.py:12-21: make_csv_dataset(shuffle=True,
 * num_epochs=None)), so Keras' Unique + UnsortedSegmentSum see new batches every step (a9).  Instead of sorting every batch
 * when it is used (glove_plan_build: two stable sorts per step), a rank sorts its nonzeros ONCE:
 *
 *   glove_masters_build   the two master orders of the rank's nonzeros — row-major = sorted by (row id, col id, stream
 *                         index), col-major = sorted by (col id, row id, stream index) — and link[q] = the row-major position
 *                         of the pair at col-major position q.  Ids outside their table count as id 0 (the reference's
 *                         unknown-token id, estimator.py:26-28): *mapped_out (device int32, optional) = how many.
 *   glove_epoch_deal      one epoch: a bijection seat() of [0, n) determined by the 128-bit key (the Feistel network of
 *                         glove_epoch.hip) gives the pair at row-major position p the seat seat(p), i.e. batch
 *                         seat(p) / B; a stable counting sort by batch number writes both orders so that batch k occupies
 *                         positions [k B, (k + 1) B) of row_side and of col_side, sorted by row id / by col id — ties in
 *                         master order.  Every full batch holds exactly B pairs; the n mod B pairs of the last, partial
 *                         batch sit behind them and wait for the next epoch's deal.  Out of place.
 *   glove_plan_build_sorted   the dedup indexes of n_batches consecutive batches of a dealt epoch in three launches, no
 *                         sort (below).
 *
 * A batch of a dealt epoch "arrives" in row-major order: its index is what glove_plan_build gives for the row side's pairs
 * (row_side.id, row_side.partner, w, y) handed over in that order — the stable sort of that order by col id is the col side's
 * (col id, row id, stream index) order. */
typedef struct glove_pairs {
    int32_t *id;                /* [n] the pair's id on this order's own side (row ids in the row-major order) */
    int32_t *partner;           /* [n] its id on the other side */
    float *w, *y;               /* [n] glove_weight, glove_value */
} glove_pairs;
size_t glove_masters_workspace_bytes(int64_t n);
int glove_masters_build(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t n, int32_t V,
                        int32_t V_row /* 0 = V */, const glove_pairs *row_major, const glove_pairs *col_major, int32_t *link,
                        int32_t *mapped_out, void *ws, size_t ws_bytes, void *stream);
size_t glove_epoch_deal_workspace_bytes(int64_t n, int64_t B);
int glove_epoch_deal(const glove_pairs *row_major, const glove_pairs *col_major, const int32_t *link, int64_t n, int64_t B,
                     uint64_t key_lo, uint64_t key_hi, const glove_pairs *row_side, const glove_pairs *col_side,
                     void *ws, size_t ws_bytes, void *stream);
/* plans[j] (a HOST array of n_batches structs; plans_dev = the same array in device memory, which the kernels read: one
 * launch covers any number of batches and no plan travels in an argument block) indexes the batch at positions
 * [first_pair + j B, first_pair + (j + 1) B) of the two orders.  All plans have the same B, chunk_cap and kind.  A plan
 * with chunk records needs no pair arrays of its own (r_partner .. c_y all NULL: the records carry the pair fields, the
 * step functions read nothing else); a plan without records gets them copied — or, when it has run words (r_chunk_hw) and
 * no pair arrays either, BORROWS them: the batch lies sorted in the epoch's arrays, nothing is copied, and the caller points
 * r_partner / r_w / r_y at row_side's partner / w / y + first_pair + j B (c_* at col_side's) in the struct it steps with;
 * the epoch's arrays must then outlive the steps.  c_perm / r_to_c must be NULL.  Capacities:
 * cap_uniq >= min(B, V) and cap_chunks >= glove_plan_chunk_bound(B, cap_uniq, chunk_cap) — an id of p pairs has at most
 * p / chunk_cap + 1 chunks. */
size_t glove_plan_sorted_workspace_bytes(int64_t B, int32_t n_batches);
int32_t glove_plan_chunk_bound(int64_t B, int32_t cap_uniq, int32_t chunk_cap);
int glove_plan_build_sorted(const glove_pairs *row_side, const glove_pairs *col_side, int64_t first_pair, int64_t B,
                            int32_t n_batches, int32_t V, const struct glove_plan *plans, const struct glove_plan *plans_dev,
                            void *ws, size_t ws_bytes, void *stream);

/* (Re)builds r_crec / c_crec of a plan whose other arrays are complete (both must be non-NULL). */
int glove_plan_fill_records(const glove_plan *plan, void *stream);

/* ---- step workspace ---------------------------------------------------------------------
 * Holds e[B] (written by glove_rowpass_f32 only), per-chunk partial gradient rows and per-block loss partials. */
size_t glove_step_workspace_bytes(int64_t B, int32_t cap_chunks, int32_t d);

/* ---- fused forward + gradient passes (model_utils.py:41-54, estimator.py:48-56, autodiff) --
 * Per pair p = r.c + br + bc + g, e = 2 w (p - y) inv_batch; then
 *   row side: loss partials and the per-chunk sums  sum_i e_i C[col_i]  /  sum_i e_i   (advances global_step)
 *   col side: the per-chunk sums  sum_i e_i R[row_i]  /  sum_i e_i
 * The sides are independent (each forms e_i itself): glove_passes_f32 runs both in ONE launch, which
 * is what the step functions use; glove_rowpass_f32 / glove_colpass_f32 run one side each.
 * glove_rowpass_f32 alone also stores e[B] (row-sorted pair order) at the start of the workspace, for
 * diagnostics and parity tests; the step path never needs it in memory.  None of them modifies the tables. */
int glove_passes_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                     void *ws, size_t ws_bytes, void *stream);
int glove_rowpass_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                      void *ws, size_t ws_bytes, void *stream);
int glove_colpass_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                      void *ws, size_t ws_bytes, void *stream);

/* ---- sparse optimizer apply: OptimizerV2 dedup + ResourceSparseApplyAdagradV2 (a9, a10) ----
 * For every distinct id: G = sum of its chunk partials + activity-L2 term, then
 * A += G^2 ; W -= lr G / (sqrt(A)+eps) on the touched rows of R, C, br, bc; dense update of the
 * global bias.  loss_out (device float[4]) = {loss, L, Reg, sum_e}. */
int glove_apply_adagrad_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                            void *ws, size_t ws_bytes, float *loss_out, void *stream);

/* ---- dense-gradient path (data-parallel all-reduce, and Keras-legacy Adam a11) -------------
 * G_flat layout (float offsets from glove_dense_grad_layout): [G_R V_row*d | G_br V_row | pad | G_C V*d |
 * G_bc V | pad | tail 8]: each side is contiguous (the col half + tail can be all-reduced alone when the row
 * table is sharded), sections 16-B aligned, tail = {sum_e, sum
 * w diff^2, sum |r|^2+|c|^2, sum br^2+bc^2, 0...}.  glove_dense_grad_f32 ADDS this batch's
 * summed gradients (incl. the activity-L2 terms) into G_flat, which the caller keeps all-zero
 * between steps (the dense apply kernels zero what they consume). */
/* offs[5] (may be NULL) = float offsets of G_R, G_br, G_C, G_bc, tail; returns the total float count of G_flat */
size_t glove_dense_grad_layout(int32_t V_row, int32_t V, int32_t d, int64_t *offs);
int glove_dense_grad_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                         void *ws, size_t ws_bytes, float *G_flat, void *stream);
int glove_dense_adagrad_f32(const glove_tables *t, const glove_hyper *h, float *G_flat,
                            float *loss_out, void *stream);
/* (Keras-legacy Adam; with glove_hyper.optimizer = GLOVE_OPT_RMSPROP the legacy RMSprop sweep instead: the whole rms slot decays,
 * entries with a non-zero summed gradient move) */
int glove_dense_adam_f32(const glove_tables *t, const glove_hyper *h, float *G_flat,
                         float *loss_out, void *stream);

/* ---- touched-rows exchange (multi-GPU forms; SURVEY.md §8e "touched-rows exchange") ----------------
 * Instead of the dense [V,d] buffer a rank hands over ONE packed list per step.  An entry is
 * GLOVE_PACKED_ENTRY_FLOATS(d) = d + 4 floats: [summed gradient row, d | bias gradient | id (int bits) |
 * side (0 row, 1 col; int bits) | 0].  Entry 0 of a list is its header [row entries, col entries (int bits), sum_e,
 * sum w diff^2, sum |r|^2+|c|^2, sum br^2+bc^2, 0...]; the row-side entries follow in plan order (ascending id), then
 * the col-side entries.  glove_pack_grad_f32 writes the list of one plan (hyper.sides selects the sides; needs
 * 1 + distinct ids entries of capacity).  It replaces glove_dense_grad_f32 in the data-parallel step whenever the
 * ranks' lists together are shorter than the dense buffer (an all-gather of lists instead of an all-reduce of
 * 2 V (d+1) floats), and it is how a rank returns its col gradients to the owners of a sharded col table. */
#define GLOVE_PACKED_ENTRY_FLOATS(d) ((size_t)(d) + 4)
int glove_pack_grad_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                        void *ws, size_t ws_bytes, float *packed, int64_t capacity_entries, void *stream);
/* The same list in two calls, for plans with chunk records and host counts (otherwise the pair falls back to
 * glove_passes_f32-of-the-selected-sides + glove_pack_grad_f32): glove_passes_packing_f32 is the pass launch of
 * hyper.sides (3 both, 1 rows, 2 cols) in the run-merged schedule of the fused step forms, in which a lane group that
 * holds every chunk of an id writes that id's entry itself, straight from registers; glove_pack_rest_f32 then packs
 * the other ids and the header.  In between the caller may run glove_rowside_step_adagrad_f32 (sides 2: the row pass's
 * loss partials reach the header that way).  Ids with several chunks are summed pair by pair in the first call and
 * chunk by chunk in glove_pack_grad_f32: equal within fp32 rounding, not bitwise. */
int glove_passes_packing_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                             void *ws, size_t ws_bytes, float *packed, int64_t capacity_entries, void *stream);
int glove_pack_rest_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                        void *ws, size_t ws_bytes, float *packed, int64_t capacity_entries, void *stream);
/* out4 (device float[4]) = {sum e, sum w diff^2, sum |r|^2+|c|^2, sum br^2+bc^2} of the plan's last row pass — what
 * a list's header carries in floats 2..5 — for callers that send the list BEFORE the row side has run (the push of
 * the col gradients then overlaps glove_rowside_step_adagrad_f32) and hand the sums over afterwards
 * (glove_apply_packed_adagrad_f32's `tail`).  Valid after glove_rowside_step_adagrad_f32 or a
 * glove_passes_packing_f32 that included the row side. */
int glove_loss_partials_f32(const glove_plan *plan, const glove_tables *t, void *ws, size_t ws_bytes,
                            float *out4, void *stream);
/* One received list.  entries: the first entry behind the header (or a bare run of entries); ids: if not NULL,
 * ids[i] replaces the id stored in entry i (an owner's local indices); header: if not NULL the entry count is read
 * from it on the device, otherwise n is the count; side: -1 = every entry names its side, 0 / 1 = all of that side. */
typedef struct glove_packed_list {
    const float *entries;
    const int32_t *ids;
    const float *header;
    int32_t n;
    int32_t side;
} glove_packed_list;
/* Optional, before the combines: counts in mark how many of the lists touch every id (at most 255 lists).  An id only
 * one list touches then skips the dense buffer altogether — glove_combine_packed_f32 leaves it alone and
 * glove_apply_packed_adagrad_f32 takes its gradient straight from the entry: two row moves less for most ids of a
 * Zipf batch.  Same results. */
int glove_count_packed_f32(const glove_packed_list *lists, int32_t n_lists, const glove_tables *t, float *G_flat,
                           int32_t *mark, int64_t capacity_entries, void *stream);
/* Adds one list into G_flat (layout of glove_dense_grad_layout): the first list to touch an id stores its row and
 * leaves its tag in mark[id] (mark: int32[V_row + V], rows first, all zero between steps), later lists add behind
 * it.  Call once per list, in rank order, on one stream: the sum over ranks then has a fixed order.  G_flat needs no
 * zeroing.  capacity_entries bounds the entry count of a list whose count lives in its header. */
int glove_combine_packed_f32(const glove_packed_list *list, int32_t tag, const glove_tables *t, float *G_flat,
                             int32_t *mark, int64_t capacity_entries, void *stream);
/* The optimizer glove_hyper.optimizer names — Adagrad, or one of the per-row Keras optimizers (GLOVE_OPT_SGD, _ADAMAX, _ADADELTA,
 * _FTRL: only touched rows move under them, so they ride the same exchange; their second slots are glove_tables.s2_*; and
 * GLOVE_OPT_NADAM, both sides in one call: the rows NO list names have their m and v decayed first, the named ones move) — on
 * every id the lists touched (lists[i] was combined with tag i): each id is applied from the list that
 * touched it first, its mark is cleared.  tail: device float[4] {sum_e, sum w diff^2, sum |r|^2+|c|^2, sum b^2}
 * already summed over the ranks, or NULL = summed here over the lists' headers in list order.  With the col side
 * selected (hyper.sides) the call also updates the global bias and writes loss_out. */
int glove_apply_packed_adagrad_f32(const glove_packed_list *lists, int32_t n_lists, const glove_tables *t,
                                   const glove_hyper *h, float *G_flat, int32_t *mark, const float *tail,
                                   float *loss_out, int64_t capacity_entries, void *stream);
/* rows[i] = W[ids[i]] (d floats), biases[i] = bias[ids[i]]: what the owner of a table shard sends to the ranks whose
 * batches touch those rows (BASELINE config 5 with the col table sharded as well). */
int glove_gather_rows_f32(const float *W, const float *bias, const int32_t *ids, int32_t n, int32_t d,
                          float *rows, float *biases, void *stream);

/* ---- whole step = session.run(train_op) (estimator.py:49-56) ------------------------------ */
int glove_step_adagrad_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                           void *ws, size_t ws_bytes, float *loss_out, void *stream);
/* The ROW side of a step done completely, for callers that exchange the col side themselves (both tables sharded,
 * trainer/stepper.py ShardedStepper): a pass over the row side in which every id one lane group holds completely is
 * applied in place (R, br, their accumulators), then the apply of the remaining row ids; the row pass's loss partials
 * are left where glove_pack_grad_f32 / glove_dense_grad_f32 look for them.  hyper.sides must be 1.
 * PRECONDITION: nothing else reads R / br later in this step — run glove_colpass_f32 (which gathers the old rows)
 * BEFORE this call.  Without chunk records it falls back to glove_rowpass_f32 + glove_apply_adagrad_f32. */
int glove_rowside_step_adagrad_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                                   void *ws, size_t ws_bytes, void *stream);
/* GLOVE_STEP_AUTO takes a fused form when (distinct row ids + distinct col ids of the plan) x d x 16 B — the rows a
 * step reads and writes — reaches this many bytes (and the plan carries chunk records or run words); callers that keep a twinned
 * table use the same number to know whether a step may have left versions flipped. */
#define GLOVE_FUSED_STEP_BYTES ((size_t)96 << 20)
/* Twinned row table (glove_tables.R_ver) back to the plain form: current rows copied into rows 0 .. V_row-1, versions
 * cleared.  A no-op without a twin.  Call before anything but glove_step(s)_adagrad_f32 reads or writes R / br. */
int glove_canonicalize_f32(const glove_tables *t, void *stream);
/* n consecutive Adagrad steps, plans[i] in order, from ONE host call (the launch loop runs in C: a Python
 * host loop costs more per step than the two kernels of a 1,024-pair step take).  loss_out, if not NULL,
 * receives the scalars of the LAST step. */
int glove_steps_adagrad_f32(const glove_plan *const *plans, int32_t n, const glove_tables *t,
                            const glove_hyper *h, void *ws, size_t ws_bytes, float *loss_out, void *stream);
/* One step under the Keras optimizer glove_hyper.optimizer names (passes + one apply launch; RMSprop: passes + the dense
 * gradient + one sweep over its slots).  G_flat: as glove_step_adam_f32 uses it — needed for GLOVE_OPT_RMSPROP and GLOVE_OPT_ADAM
 * (glove_dense_grad_layout floats, all zero on entry and on return), ignored otherwise.  GLOVE_OPT_ADAGRAD / GLOVE_OPT_ADAM go to
 * glove_step_adagrad_f32 / glove_step_adam_f32. */
int glove_step_sparse_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                          void *ws, size_t ws_bytes, float *G_flat, float *loss_out, void *stream);

/* One Keras-legacy Adam step.  G_flat (glove_dense_grad_layout floats, all zero on entry) is scratch and is all zero
 * again on return.  A batch of at most (V_row + V) / 2 pairs takes two launches: the passes also mark the batch's
 * ids (in G_flat's bias segments), then one kernel applies the marked ids and gives every other row the G = 0
 * update; larger batches run passes + glove_dense_grad_f32 + glove_dense_adam_f32.  Same result bit for bit.
 * ONE launch (glove_hyper.step_form AUTO or GLOVE_STEP_TAGGED) when both tables are twinned (glove_tables.R_tag / C_tag
 * non-NULL: R, br hold 2 x V_row rows / entries, C, bc 2 x V; the tags themselves stay zero), the plan carries chunk records
 * and id bitmaps (r_mark / c_mark) and the batch has at most 2,048 pairs and touches a minority of the rows: every row of
 * both tables moves from the current copy to the other one — the batch's rows by the lane group that holds the id's first
 * chunk (gradient, then Adam), all others by a sweep that skips the bitmaps' ids; scalars[3] says which copy is current
 * (glove_canonicalize_f32 copies the second copies home; every other entry point calls it first).  Swept rows and ids of
 * up to heavy_chunks chunks bit-identical to the two-launch form, the others within fp32 rounding of their sums' order. */
int glove_step_adam_f32(const glove_plan *plan, const glove_tables *t, const glove_hyper *h,
                        void *ws, size_t ws_bytes, float *G_flat, float *loss_out, void *stream);
/* n consecutive Keras-legacy Adam steps from one host call; G_flat is left zeroed after every step.  Consecutive steps that
 * take the one-launch form go out as a chain: one launch per step — the global bias and its moments handed on through records
 * in the workspace — and one epilogue per chain (loss, scalars, global_step += n). */
int glove_steps_adam_f32(const glove_plan *const *plans, int32_t n, const glove_tables *t,
                         const glove_hyper *h, void *ws, size_t ws_bytes, float *G_flat, float *loss_out,
                         void *stream);

/* ---- EVAL mode of model_fn: RegressionHead metrics over a batch (estimator.py:48-56) -------
 * Accumulates into sums_out (device double[4]): sum w (p-y)^2, sum w, sum w p, sum w y. */
int glove_eval_f32(const int32_t *row, const int32_t *col, const float *w, const float *y,
                   int64_t B, const glove_tables *t, double *sums_out, void *stream);
/* The same pass for GLOVE_HEAD_LOGISTIC (BinaryClassHead metrics of the two heads,
 * logistic_matrix_factorisation.py:50-54).  sums_out is device double[6]: sum pos xent(p, 1), sum pos,
 * sum neg xent(p, 0), sum neg, sum pos sigmoid(p), sum neg sigmoid(p). */
int glove_eval_logistic_f32(const int32_t *row, const int32_t *col, const float *pos, const float *neg,
                            int64_t B, const glove_tables *t, double *sums_out, void *stream);

/* ---- PREDICT mode: cosine_similarity + tf.math.top_k (model_utils.py:81-110, utils.py:12-19) --
 * For n query ids: sims/idx [n,k] sorted descending (ties: lower id first) over all V ROW embeddings; d = row
 * stride in floats (multiple of 4), 1 <= k <= min(V, 1024).  The similarity matrix is the one GEMM of the path and
 * runs on the matrix cores in exact f32 (v_mfma_f32_32x32x2_f32); the top-k is a staged selection.
 * ws: glove_topk_workspace_bytes. */
size_t glove_topk_workspace_bytes(int32_t n, int32_t V, int32_t k);
int glove_topk_cosine_f32(const float *R, int32_t V, int32_t d, const int32_t *query_ids, int32_t n,
                          int32_t k, float *sims_out, int32_t *idx_out,
                          void *ws, size_t ws_bytes, void *stream);

/* ---- data prep: windowed co-occurrence counts (reference src/data/text8.py:84-108) ----------------
 * tokens: int32[n] vocabulary ids of the corpus in order (OOV already mapped to 0, text8.py:86).
 * For every position p and offset k in 1..context with a = tok[p] != b = tok[p+k]:
 * count(a,b) += 1, value(a,b) += 1/k, and the same for (b,a) (the reference's union with the swapped
 * table).  Output sorted by (row, col): out_row/out_col int32[cap], out_count int64[cap],
 * out_value double[cap]; *out_nnz (device int64) = number of distinct (row, col) — may exceed cap, in
 * which case only the first cap entries were written. */
size_t glove_cooc_workspace_bytes(int64_t n_tokens, int32_t context);
int glove_cooccurrence_i32(const int32_t *tokens, int64_t n_tokens, int32_t V, int32_t context,
                           int32_t *out_row, int32_t *out_col, int64_t *out_count, double *out_value,
                           int64_t *out_nnz, int64_t cap, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GLOVE_HIP_H */
