/*
 * TEST INFRASTRUCTURE ONLY — scalar fp32 C restatement of the GloVe training step.
 *
 * Checker and CPU baseline ("port") for the HIP path; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg load it.  PARITY UNPINNED: the reference's step runs
 * inside TensorFlow 2.11 (absent here, no golden vectors in the reference), so this follows
 * the reference call sites and the pinned third-party semantics listed in
 * oracle/glove_ref.py's header and SURVEY.md §8a:
 *
 *   forward           src/models/model_utils.py:41-54
 *   activity L2       src/models/model_utils.py:8,18-21,32-38,52
 *   weighted MSE/B    src/models/estimator.py:48-56
 *   dedup, Adagrad    keras 2.11 OptimizerV2 (sum duplicates, then square; eps outside sqrt)
 *   Adam              keras 2.11 legacy Adam sparse path = whole-table decay every step
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; no -ffast-math so the arithmetic is plain
 * IEEE fp32 in batch order, like TF's UnsortedSegmentSum on one thread).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t V, d;
    float *R, *C, *br, *bc;          /* [V,d],[V,d],[V],[V] */
    float *S1_R, *S1_C, *S1_br, *S1_bc; /* Adagrad accumulator, or Adam m */
    float *S2_R, *S2_C, *S2_br, *S2_bc; /* Adam v (unused for Adagrad) */
    float *scal;                     /* [0]=g [1]=slot1(g) [2]=slot2(g) */
    int64_t step;
    /* scratch owned by the caller: dense gradient buffers (kept all-zero between steps)
       and first-touch lists */
    float *G_R, *G_C, *G_br, *G_bc;
    int32_t *touch_r, *touch_c;      /* [B] each */
    uint8_t *mark_r, *mark_c;        /* [V] each, zero between steps */
} glove_ref_state;

typedef struct {
    float l2_reg, reg_mult, lr, eps, beta1, beta2, inv_batch;
} glove_ref_hyper;

static void grads(glove_ref_state *s, const glove_ref_hyper *h, const int32_t *row, const int32_t *col,
                  const float *w, const float *y, int64_t B, int *nr, int *nc, float *sum_e,
                  float *loss_L, float *loss_reg)
{
    const int d = s->d;
    const float ib = h->inv_batch;
    const float kappa = 2.0f * h->reg_mult * h->l2_reg / (float)d * ib;
    const float kappa_b = 2.0f * h->reg_mult * h->l2_reg * ib;
    const float g = s->scal[0];
    double L = 0.0, sq = 0.0, sqb = 0.0; /* loss bookkeeping only */
    float se = 0.0f;
    int n_r = 0, n_c = 0;
    for (int64_t i = 0; i < B; ++i) {
        const int32_t u = row[i], v = col[i];
        const float *r = s->R + (size_t)u * d, *c = s->C + (size_t)v * d;
        float dot = 0.0f, rr = 0.0f, cc = 0.0f;
        for (int k = 0; k < d; ++k) { dot += r[k] * c[k]; rr += r[k] * r[k]; cc += c[k] * c[k]; }
        const float bru = s->br[u], bcv = s->bc[v];
        const float diff = dot + bru + bcv + g - y[i];
        const float e = 2.0f * w[i] * diff * ib;
        L += (double)(w[i] * diff * diff);
        sq += (double)(rr + cc);
        sqb += (double)(bru * bru + bcv * bcv);
        se += e;
        if (!s->mark_r[u]) { s->mark_r[u] = 1; s->touch_r[n_r++] = u; }
        if (!s->mark_c[v]) { s->mark_c[v] = 1; s->touch_c[n_c++] = v; }
        float *gr = s->G_R + (size_t)u * d, *gc = s->G_C + (size_t)v * d;
        for (int k = 0; k < d; ++k) {
            const float rk = r[k], ck = c[k];
            gr[k] += e * ck + kappa * rk;
            gc[k] += e * rk + kappa * ck;
        }
        s->G_br[u] += e + kappa_b * bru;
        s->G_bc[v] += e + kappa_b * bcv;
    }
    *nr = n_r; *nc = n_c; *sum_e = se;
    *loss_L = (float)(L * ib);
    *loss_reg = (float)(h->l2_reg / d * ib * sq + h->l2_reg * ib * sqb + h->l2_reg * g * g);
}

static inline void adagrad_row(float *W, float *A, float *G, int n, float lr, float eps)
{
    for (int k = 0; k < n; ++k) {
        const float gk = G[k];
        A[k] += gk * gk;
        W[k] -= lr * gk / (sqrtf(A[k]) + eps);
        G[k] = 0.0f;
    }
}

/* One Adagrad step over a batch. out[0]=loss, out[1]=L, out[2]=Reg. */
int glove_ref_step_adagrad_f32(glove_ref_state *s, const glove_ref_hyper *h, const int32_t *row,
                               const int32_t *col, const float *w, const float *y, int64_t B, float *out)
{
    int nr, nc; float se, L, reg;
    grads(s, h, row, col, w, y, B, &nr, &nc, &se, &L, &reg);
    const int d = s->d;
    for (int q = 0; q < nr; ++q) {
        const int32_t u = s->touch_r[q];
        adagrad_row(s->R + (size_t)u * d, s->S1_R + (size_t)u * d, s->G_R + (size_t)u * d, d, h->lr, h->eps);
        adagrad_row(s->br + u, s->S1_br + u, s->G_br + u, 1, h->lr, h->eps);
        s->mark_r[u] = 0;
    }
    for (int q = 0; q < nc; ++q) {
        const int32_t v = s->touch_c[q];
        adagrad_row(s->C + (size_t)v * d, s->S1_C + (size_t)v * d, s->G_C + (size_t)v * d, d, h->lr, h->eps);
        adagrad_row(s->bc + v, s->S1_bc + v, s->G_bc + v, 1, h->lr, h->eps);
        s->mark_c[v] = 0;
    }
    float dg = se + 2.0f * h->reg_mult * h->l2_reg * s->scal[0];
    adagrad_row(&s->scal[0], &s->scal[1], &dg, 1, h->lr, h->eps);
    s->step += 1;
    out[0] = L + h->reg_mult * reg; out[1] = L; out[2] = reg;
    return 0;
}

static inline void adam_sweep(float *W, float *M, float *Vv, float *G, size_t n, float lr_t, float b1,
                              float b2, float eps)
{
    for (size_t k = 0; k < n; ++k) {
        const float gk = G[k];
        M[k] = b1 * M[k] + (1.0f - b1) * gk;
        Vv[k] = b2 * Vv[k] + (1.0f - b2) * gk * gk;
        W[k] -= lr_t * M[k] / (sqrtf(Vv[k]) + eps);
        G[k] = 0.0f;
    }
}

/* One Keras-legacy Adam step: every row of every table moves (dense decay). */
int glove_ref_step_adam_f32(glove_ref_state *s, const glove_ref_hyper *h, const int32_t *row,
                            const int32_t *col, const float *w, const float *y, int64_t B, float *out)
{
    int nr, nc; float se, L, reg;
    grads(s, h, row, col, w, y, B, &nr, &nc, &se, &L, &reg);
    for (int q = 0; q < nr; ++q) s->mark_r[s->touch_r[q]] = 0;
    for (int q = 0; q < nc; ++q) s->mark_c[s->touch_c[q]] = 0;
    const double t = (double)(s->step + 1);
    const float lr_t = (float)((double)h->lr * sqrt(1.0 - pow((double)h->beta2, t)) / (1.0 - pow((double)h->beta1, t)));
    const size_t n = (size_t)s->V * s->d;
    adam_sweep(s->R, s->S1_R, s->S2_R, s->G_R, n, lr_t, h->beta1, h->beta2, h->eps);
    adam_sweep(s->C, s->S1_C, s->S2_C, s->G_C, n, lr_t, h->beta1, h->beta2, h->eps);
    adam_sweep(s->br, s->S1_br, s->S2_br, s->G_br, (size_t)s->V, lr_t, h->beta1, h->beta2, h->eps);
    adam_sweep(s->bc, s->S1_bc, s->S2_bc, s->G_bc, (size_t)s->V, lr_t, h->beta1, h->beta2, h->eps);
    float dg = se + 2.0f * h->reg_mult * h->l2_reg * s->scal[0];
    adam_sweep(&s->scal[0], &s->scal[1], &s->scal[2], &dg, 1, lr_t, h->beta1, h->beta2, h->eps);
    s->step += 1;
    out[0] = L + h->reg_mult * reg; out[1] = L; out[2] = reg;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * All-core form of the same step (OpenMP), the CPU baseline bench.py times beside the GPU
 * ("all threads").  Same arithmetic per pair as above; what changes is who sums what:
 * the pairs of one id are summed by ONE thread in batch order (so an id's gradient has the bits of the
 * scalar port), ids with more than `chunk` pairs are cut into chunks summed by different threads whose
 * partial rows are then added in chunk order.  The grouping of a batch's pairs by id (`glove_ref_index`)
 * is built once per resident batch by the caller (numpy), outside the timed region, exactly as the GPU
 * path keeps its dedup index of a static stream resident.
 * ------------------------------------------------------------------------------------------------ */
#include <omp.h>

typedef struct {
    int32_t n_uniq, n_chunks;
    const int32_t *order;     /* [B]  pair indices grouped by id, batch order inside a group */
    const int32_t *uid;       /* [n_uniq] the ids */
    const int32_t *first;     /* [n_uniq+1] first chunk of the q-th id */
    const int32_t *ch_lo;     /* [n_chunks+1] chunk c covers order[ch_lo[c] .. ch_lo[c+1]) */
    const int32_t *ch_q;      /* [n_chunks] which id the chunk belongs to */
} glove_ref_side_index;

typedef struct {
    glove_ref_side_index r, c;
    float *e;                 /* [B] scratch */
    float *P_r, *P_c;         /* [n_chunks, d] partial rows (used by ids with several chunks) */
    float *Pb_r, *Pb_c;       /* [n_chunks] */
} glove_ref_index;

static void side_grads_mt(const glove_ref_state *s, const glove_ref_side_index *ix, const int32_t *own_id,
                          const int32_t *other_id, const float *e, const float *W, const float *Wo, const float *bias,
                          float *G, float *Gb, float *P, float *Pb, float kappa, float kappa_b)
{
    const int d = s->d;
#pragma omp parallel for schedule(dynamic, 16)
    for (int32_t c = 0; c < ix->n_chunks; ++c) {
        const int32_t q = ix->ch_q[c];
        const int32_t u = ix->uid[q];
        const int single = ix->first[q + 1] - ix->first[q] == 1;
        float *dst = single ? G + (size_t)u * d : P + (size_t)c * d;
        float gb = 0.0f;
        const float *own = W + (size_t)u * d;
        const float bu = bias[u];
        if (!single) memset(dst, 0, (size_t)d * sizeof(float));
        for (int32_t k = ix->ch_lo[c]; k < ix->ch_lo[c + 1]; ++k) {
            const int32_t i = ix->order[k];
            (void)own_id;
            const float *oth = Wo + (size_t)other_id[i] * d;
            const float ei = e[i];
            for (int x = 0; x < d; ++x) dst[x] += ei * oth[x] + kappa * own[x];
            gb += ei + kappa_b * bu;
        }
        if (single) Gb[u] += gb; else Pb[c] = gb;
    }
}

static void side_apply_mt(const glove_ref_state *s, const glove_ref_hyper *h, const glove_ref_side_index *ix,
                          float *W, float *S1, float *bias, float *S1b, float *G, float *Gb, const float *P,
                          const float *Pb, int adagrad)
{
    const int d = s->d;
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t q = 0; q < ix->n_uniq; ++q) {
        const int32_t u = ix->uid[q];
        float *g = G + (size_t)u * d;
        if (ix->first[q + 1] - ix->first[q] > 1) {
            for (int32_t c = ix->first[q]; c < ix->first[q + 1]; ++c) {
                const float *p = P + (size_t)c * d;
                for (int x = 0; x < d; ++x) g[x] += p[x];
                Gb[u] += Pb[c];
            }
        }
        if (adagrad) {
            adagrad_row(W + (size_t)u * d, S1 + (size_t)u * d, g, d, h->lr, h->eps);
            adagrad_row(bias + u, S1b + u, Gb + u, 1, h->lr, h->eps);
        }
    }
}

static void adam_sweep_mt(float *W, float *M, float *Vv, float *G, size_t n, float lr_t, float b1, float b2, float eps)
{
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)n; ++k) {
        const float gk = G[k];
        M[k] = b1 * M[k] + (1.0f - b1) * gk;
        Vv[k] = b2 * Vv[k] + (1.0f - b2) * gk * gk;
        W[k] -= lr_t * M[k] / (sqrtf(Vv[k]) + eps);
        G[k] = 0.0f;
    }
}

/* One step on `threads` cores. adam = 0: Adagrad, 1: Keras-legacy Adam. out[0]=loss, out[1]=L, out[2]=Reg. */
int glove_ref_step_mt_f32(glove_ref_state *s, const glove_ref_hyper *h, const glove_ref_index *ix, const int32_t *row,
                          const int32_t *col, const float *w, const float *y, int64_t B, int adam, int threads, float *out)
{
    const int d = s->d;
    const float ib = h->inv_batch;
    const float kappa = 2.0f * h->reg_mult * h->l2_reg / (float)d * ib;
    const float kappa_b = 2.0f * h->reg_mult * h->l2_reg * ib;
    const float g = s->scal[0];
    double L = 0.0, sq = 0.0, sqb = 0.0, se = 0.0;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) reduction(+ : L, sq, sqb, se)
    for (int64_t i = 0; i < B; ++i) {
        const int32_t u = row[i], v = col[i];
        const float *r = s->R + (size_t)u * d, *c = s->C + (size_t)v * d;
        float dot = 0.0f, rr = 0.0f, cc = 0.0f;
        for (int k = 0; k < d; ++k) { dot += r[k] * c[k]; rr += r[k] * r[k]; cc += c[k] * c[k]; }
        const float bru = s->br[u], bcv = s->bc[v];
        const float diff = dot + bru + bcv + g - y[i];
        const float e = 2.0f * w[i] * diff * ib;
        ix->e[i] = e;
        L += (double)(w[i] * diff * diff);
        sq += (double)(rr + cc);
        sqb += (double)(bru * bru + bcv * bcv);
        se += (double)e;
    }
    side_grads_mt(s, &ix->r, row, col, ix->e, s->R, s->C, s->br, s->G_R, s->G_br, ix->P_r, ix->Pb_r, kappa, kappa_b);
    side_grads_mt(s, &ix->c, col, row, ix->e, s->C, s->R, s->bc, s->G_C, s->G_bc, ix->P_c, ix->Pb_c, kappa, kappa_b);
    float dg = (float)se + 2.0f * h->reg_mult * h->l2_reg * g;
    if (!adam) {
        side_apply_mt(s, h, &ix->r, s->R, s->S1_R, s->br, s->S1_br, s->G_R, s->G_br, ix->P_r, ix->Pb_r, 1);
        side_apply_mt(s, h, &ix->c, s->C, s->S1_C, s->bc, s->S1_bc, s->G_C, s->G_bc, ix->P_c, ix->Pb_c, 1);
        adagrad_row(&s->scal[0], &s->scal[1], &dg, 1, h->lr, h->eps);
    } else {
        side_apply_mt(s, h, &ix->r, s->R, s->S1_R, s->br, s->S1_br, s->G_R, s->G_br, ix->P_r, ix->Pb_r, 0);
        side_apply_mt(s, h, &ix->c, s->C, s->S1_C, s->bc, s->S1_bc, s->G_C, s->G_bc, ix->P_c, ix->Pb_c, 0);
        const double t = (double)(s->step + 1);
        const float lr_t = (float)((double)h->lr * sqrt(1.0 - pow((double)h->beta2, t)) / (1.0 - pow((double)h->beta1, t)));
        const size_t n = (size_t)s->V * d;
        adam_sweep_mt(s->R, s->S1_R, s->S2_R, s->G_R, n, lr_t, h->beta1, h->beta2, h->eps);
        adam_sweep_mt(s->C, s->S1_C, s->S2_C, s->G_C, n, lr_t, h->beta1, h->beta2, h->eps);
        adam_sweep_mt(s->br, s->S1_br, s->S2_br, s->G_br, (size_t)s->V, lr_t, h->beta1, h->beta2, h->eps);
        adam_sweep_mt(s->bc, s->S1_bc, s->S2_bc, s->G_bc, (size_t)s->V, lr_t, h->beta1, h->beta2, h->eps);
        adam_sweep(&s->scal[0], &s->scal[1], &s->scal[2], &dg, 1, lr_t, h->beta1, h->beta2, h->eps);
    }
    s->step += 1;
    const float Lf = (float)(L * ib);
    const float reg = (float)(h->l2_reg / d * ib * sq + h->l2_reg * ib * sqb + h->l2_reg * g * g);
    out[0] = Lf + h->reg_mult * reg; out[1] = Lf; out[2] = reg;
    return 0;
}

int glove_ref_max_threads(void) { return omp_get_max_threads(); }
