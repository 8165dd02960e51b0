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
