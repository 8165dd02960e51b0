// EVAL and PREDICT modes of the reference model_fn on gfx950.
//
//   glove_eval_f32         RegressionHead eval metrics (reference src/models/estimator.py:48-56:
//                          average_loss / prediction/mean / label/mean are weighted sums over the
//                          eval pass) — forward only, tables untouched.
//   glove_topk_cosine_f32  get_predictions (src/models/model_utils.py:81-110): cosine_similarity
//                          (src/models/utils.py:12-19) of query ROW embeddings against all V row
//                          embeddings + tf.math.top_k (descending, ties -> lower index).
#include "glove_common.h"

namespace glove {

template <int LPR, int NV>
__device__ inline void load_row_p(f4 (&dst)[NV], const float *table, int32_t id, int d4, int lg)
{
    const f4 *p = reinterpret_cast<const f4 *>(table) + (size_t)id * d4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = lg + k * LPR;
        dst[k] = (i4 < d4) ? p[i4] : f4{0.f, 0.f, 0.f, 0.f};
    }
}

__device__ inline double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void eval_kernel(const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                      const float *__restrict__ w, const float *__restrict__ y,
                                                      int64_t B, const float *__restrict__ R,
                                                      const float *__restrict__ C, const float *__restrict__ br,
                                                      const float *__restrict__ bc, const float *__restrict__ scalars,
                                                      int d4, double *__restrict__ sums, int head)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const float g = scalars[0];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
    for (int64_t i = (int64_t)blockIdx.x * GPB + grp; i < B; i += (int64_t)gridDim.x * GPB) {
        const int32_t u = row[i], v = col[i];
        f4 r[NV], c[NV];
        load_row_p<LPR, NV>(r, R, u, d4, lg);
        load_row_p<LPR, NV>(c, C, v, d4, lg);
        float dp = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) dp += dot4(r[k], c[k]);
        const float p = group_sum<LPR>(dp) + br[u] + bc[v] + g;
        if (lg == 0 && head == GLOVE_HEAD_REGRESSION) {
            const double wi = w[i], yi = y[i], diff = (double)p - yi;
            a0 += wi * diff * diff; a1 += wi; a2 += wi * (double)p; a3 += wi * yi;
        } else if (lg == 0) {
            // w = positive weight (label 1), y = negative weight (label 0); sigmoid cross-entropy of one logit
            const double pos = w[i], neg = y[i], x = p;
            const double lse = log1p(exp(-fabs(x))), sg = 1.0 / (1.0 + exp(-x));
            a0 += pos * (fmax(-x, 0.0) + lse); a1 += pos; a2 += neg * (fmax(x, 0.0) + lse); a3 += neg;
            a4 += pos * sg; a5 += neg * sg;
        }
    }
    a0 = wave_sum_f64(a0); a1 = wave_sum_f64(a1); a2 = wave_sum_f64(a2); a3 = wave_sum_f64(a3);
    if (head != GLOVE_HEAD_REGRESSION) { a4 = wave_sum_f64(a4); a5 = wave_sum_f64(a5); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], a0); atomicAdd(&sums[1], a1); atomicAdd(&sums[2], a2); atomicAdd(&sums[3], a3);
        if (head != GLOVE_HEAD_REGRESSION) { atomicAdd(&sums[4], a4); atomicAdd(&sums[5], a5); }
    }
}

// inv_norm[v] = 1/sqrt(max(|R_v|^2, 1e-12))  (tf.math.l2_normalize epsilon)
template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void inv_norm_kernel(const float *__restrict__ R, int32_t V, int d4,
                                                          float *__restrict__ inv_norm)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int v = blockIdx.x * GPB + grp; v < V; v += gridDim.x * GPB) {
        f4 r[NV];
        load_row_p<LPR, NV>(r, R, v, d4, lg);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += dot4(r[k], r[k]);
        s = group_sum<LPR>(s);
        if (lg == 0) inv_norm[v] = 1.0f / sqrtf(fmaxf(s, 1e-12f));
    }
}

constexpr int kQT = 4;   // queries per workgroup row in the similarity kernel

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void cosine_kernel(const float *__restrict__ R, int32_t V, int d4,
                                                        const int32_t *__restrict__ qid, int32_t n,
                                                        const float *__restrict__ inv_norm,
                                                        float *__restrict__ sims /* [n,V] */)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int q0 = blockIdx.y * kQT;
    f4 q[kQT][NV];
    float qi[kQT];
#pragma unroll
    for (int a = 0; a < kQT; ++a) {
        const int32_t id = qid[(q0 + a < n) ? q0 + a : q0];
        load_row_p<LPR, NV>(q[a], R, id, d4, lg);
        qi[a] = inv_norm[id];
    }
    for (int v = blockIdx.x * GPB + grp; v < V; v += gridDim.x * GPB) {
        f4 r[NV];
        load_row_p<LPR, NV>(r, R, v, d4, lg);
        const float iv = inv_norm[v];
#pragma unroll
        for (int a = 0; a < kQT; ++a) {
            float dp = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) dp += dot4(q[a][k], r[k]);
            dp = group_sum<LPR>(dp);
            if (lg == 0 && q0 + a < n) sims[(size_t)(q0 + a) * V + v] = dp * qi[a] * iv;
        }
    }
}

// order of tf.math.top_k: larger similarity first, ties -> lower index first
__device__ inline bool topk_before(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

// One pass over the V similarities of a query (one workgroup per query): every thread keeps the best K of
// its strided share in registers (sorted insertion, static indices), the 256 lists meet in LDS and k rounds
// of workgroup arg-max pick the result.  O(V) reads per query instead of k * V.
template <int K>
__global__ __launch_bounds__(kBlock) void topk_kernel(const float *__restrict__ sims, int32_t V, int32_t k,
                                                      float *__restrict__ out_sim, int32_t *__restrict__ out_idx)
{
    __shared__ float c_val[kBlock * K];
    __shared__ int c_idx[kBlock * K];
    __shared__ float s_val[kBlock / 64];
    __shared__ int s_idx[kBlock / 64];
    __shared__ float prev_val;
    __shared__ int prev_idx;
    const float *row = sims + (size_t)blockIdx.x * V;
    float val[K];
    int idx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { val[j] = -INFINITY; idx[j] = 0x7fffffff; }
    for (int v = threadIdx.x; v < V; v += kBlock) {
        const float s = row[v];
        if (topk_before(s, v, val[K - 1], idx[K - 1])) {
            val[K - 1] = s; idx[K - 1] = v;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                if (topk_before(val[j], idx[j], val[j - 1], idx[j - 1])) {
                    const float tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
                    const int ti = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = ti;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) { c_val[threadIdx.x * K + j] = val[j]; c_idx[threadIdx.x * K + j] = idx[j]; }
    if (threadIdx.x == 0) { prev_val = INFINITY; prev_idx = -1; }
    __syncthreads();
    for (int t = 0; t < k; ++t) {
        const float pv = prev_val;
        const int pi = prev_idx;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        // each thread's list is sorted: its first entry behind the previous pick is its best remaining one
        for (int j = 0; j < K; ++j) {
            const float s = c_val[threadIdx.x * K + j];
            const int v = c_idx[threadIdx.x * K + j];
            if ((s < pv || (s == pv && v > pi)) && topk_before(s, v, best, bi)) { best = s; bi = v; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (topk_before(ob, oi, best, bi)) { best = ob; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float bv = s_val[0]; int i = s_idx[0];
            for (int wv = 1; wv < kBlock / 64; ++wv)
                if (topk_before(s_val[wv], s_idx[wv], bv, i)) { bv = s_val[wv]; i = s_idx[wv]; }
            if (i == 0x7fffffff) { bv = -INFINITY; i = -1; }   // fewer than k candidates
            out_sim[(size_t)blockIdx.x * k + t] = bv;
            out_idx[(size_t)blockIdx.x * k + t] = i;
            prev_val = bv; prev_idx = (i < 0) ? V : i;
        }
        __syncthreads();
    }
}

}  // namespace glove

using namespace glove;

extern "C" {

static int launch_eval(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B,
                       const glove_tables *t, double *sums_out, void *stream, int head);

int glove_eval_f32(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B,
                   const glove_tables *t, double *sums_out, void *stream)
{
    return launch_eval(row, col, w, y, B, t, sums_out, stream, GLOVE_HEAD_REGRESSION);
}

int glove_eval_logistic_f32(const int32_t *row, const int32_t *col, const float *pos, const float *neg, int64_t B,
                            const glove_tables *t, double *sums_out, void *stream)
{
    return launch_eval(row, col, pos, neg, B, t, sums_out, stream, GLOVE_HEAD_LOGISTIC);
}

static int launch_eval(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B,
                       const glove_tables *t, double *sums_out, void *stream, int head)
{
    if (!t || !sums_out || B < 0 || t->V <= 0 || t->d <= 0 || (t->d % 4) != 0) return GLOVE_E_BADARG;
    if (!t->R || !t->C || !t->br || !t->bc || !t->scalars) return GLOVE_E_BADARG;
    if (B == 0) return 0;
    if (!row || !col || !w || !y) return GLOVE_E_BADARG;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    const int nb = blocks_for(B, kBlock / (shape.lpr ? shape.lpr : 64));
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV)                                                                                           \
    hipLaunchKernelGGL((eval_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, row, col, w, y, B, t->R, t->C, \
                       t->br, t->bc, t->scalars, d4, sums_out, head)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

size_t glove_topk_workspace_bytes(int32_t n, int32_t V, int32_t k)
{
    (void)k;
    if (n < 0 || V <= 0) return 0;
    return align_up((size_t)V * sizeof(float), 256) + align_up((size_t)n * V * sizeof(float), 256);
}

int glove_topk_cosine_f32(const float *R, int32_t V, int32_t d, const int32_t *query_ids, int32_t n, int32_t k,
                          float *sims_out, int32_t *idx_out, void *ws, size_t ws_bytes, void *stream)
{
    if (!R || V <= 0 || d <= 0 || (d % 4) != 0 || n < 0 || n > 65535 * kQT || k <= 0 || k > V || k > 64) return GLOVE_E_BADARG;
    if (n == 0) return 0;
    if (!query_ids || !sims_out || !idx_out || !ws) return GLOVE_E_BADARG;
    if (glove_topk_workspace_bytes(n, V, k) > ws_bytes) return GLOVE_E_WORKSPACE;
    float *inv_norm = (float *)ws;
    float *sims = (float *)((char *)ws + align_up((size_t)V * sizeof(float), 256));
    const int d4 = d / 4;
    const RowShape shape = pick_row_shape(d4);
    if (shape.lpr == 0) return GLOVE_E_BADARG;
    const int gpb = kBlock / shape.lpr;
    const int nbv = blocks_for(V, gpb);
    const int nbx = nbv > 256 ? 256 : nbv;
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV)                                                                                             \
    hipLaunchKernelGGL((inv_norm_kernel<LPR, NV>), dim3(nbv), dim3(kBlock), 0, st, R, V, d4, inv_norm);           \
    hipLaunchKernelGGL((cosine_kernel<LPR, NV>), dim3(nbx, (n + kQT - 1) / kQT), dim3(kBlock), 0, st, R, V, d4, \
                       query_ids, n, inv_norm, sims)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    if (k <= 8) hipLaunchKernelGGL(topk_kernel<8>, dim3(n), dim3(kBlock), 0, st, sims, V, k, sims_out, idx_out);
    else if (k <= 20) hipLaunchKernelGGL(topk_kernel<20>, dim3(n), dim3(kBlock), 0, st, sims, V, k, sims_out, idx_out);
    else hipLaunchKernelGGL(topk_kernel<64>, dim3(n), dim3(kBlock), 0, st, sims, V, k, sims_out, idx_out);
    return (int)hipGetLastError();
}

}  // extern "C"
