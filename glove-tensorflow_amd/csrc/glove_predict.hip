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

// sims[q, v] = (R[qid[q]] . R[v]) inv_norm[qid[q]] inv_norm[v]: the one GEMM-shaped piece of the path
// (tf.matmul of the l2-normalised query rows with all rows, utils.py:12-19), on the matrix cores in exact f32:
// v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain, so the numerics are those of a scalar loop.
// Workgroup = 128 queries x 128 vocabulary rows, four waves of 2 x 2 MFMA tiles (64 accumulator VGPRs); the
// operands go through LDS in slabs of 32 columns (row stride 33 floats: a tile column is read conflict-free).
// One call reads R once per 128 queries instead of once per 4.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kSimTile = 128, kSimK = 32;

__global__ __launch_bounds__(kBlock) void cosine_mfma_kernel(const float *__restrict__ R, int32_t V, int32_t d,
                                                             const int32_t *__restrict__ qid, int32_t n,
                                                             const float *__restrict__ inv_norm,
                                                             float *__restrict__ sims /* [n,V] */)
{
    __shared__ float Qs[kSimTile][kSimK + 1];
    __shared__ float Rs[kSimTile][kSimK + 1];
    __shared__ float q_inv[kSimTile];                      // inverse norms of this tile's queries, for the epilogue
    const int v0 = blockIdx.x * kSimTile, q0 = blockIdx.y * kSimTile;
    if (threadIdx.x < kSimTile) q_inv[threadIdx.x] = q0 + threadIdx.x < n ? inv_norm[qid[q0 + threadIdx.x]] : 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;               // this wave's 64 x 64 quadrant of the tile
    const int r32 = lane & 31, kh = lane >> 5;             // operand maps: A[i = lane & 31][k = lane >> 5], B likewise
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // a slab = 128 rows x 32 columns of each operand: 8 threads x 16 B per row, zero beyond n / V / d.  The next
    // slab's global loads are issued before the MFMAs of the current one and land in LDS after them.
    constexpr int kPer = kSimTile * (kSimK / 4) / kBlock;  // float4 per thread, operand and slab
    f4 qn[kPer], rn[kPer];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int x = 0; x < kPer; ++x) {
            const int i = threadIdx.x + x * kBlock;
            const int row = i / (kSimK / 4), c = (i % (kSimK / 4)) * 4;
            const bool kin = k0 + c < d;                   // d is a multiple of 4
            qn[x] = rn[x] = f4{0.f, 0.f, 0.f, 0.f};
            if (kin && q0 + row < n) qn[x] = *reinterpret_cast<const f4 *>(R + (size_t)qid[q0 + row] * d + k0 + c);
            if (kin && v0 + row < V) rn[x] = *reinterpret_cast<const f4 *>(R + (size_t)(v0 + row) * d + k0 + c);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < d; k0 += kSimK) {
#pragma unroll
        for (int x = 0; x < kPer; ++x) {
            const int i = threadIdx.x + x * kBlock;
            const int row = i / (kSimK / 4), c = (i % (kSimK / 4)) * 4;
            Qs[row][c] = qn[x].x; Qs[row][c + 1] = qn[x].y; Qs[row][c + 2] = qn[x].z; Qs[row][c + 3] = qn[x].w;
            Rs[row][c] = rn[x].x; Rs[row][c + 1] = rn[x].y; Rs[row][c + 2] = rn[x].z; Rs[row][c + 3] = rn[x].w;
        }
        __syncthreads();
        if (k0 + kSimK < d) fetch(k0 + kSimK);
#pragma unroll 4
        for (int kk = 0; kk < kSimK; kk += 2) {
            const float a0 = Qs[wm * 64 + r32][kk + kh], a1 = Qs[wm * 64 + 32 + r32][kk + kh];
            const float b0 = Rs[wn * 64 + r32][kk + kh], b1 = Rs[wn * 64 + 32 + r32][kk + kh];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map: column (vocabulary row) = lane & 31, row (query) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int v = v0 + wn * 64 + b * 32 + r32;
        const float iv = v < V ? inv_norm[v] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ql = wm * 64 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh, q = q0 + ql;
                if (q < n && v < V) sims[(size_t)q * V + v] = acc[a][b][reg] * q_inv[ql] * iv;
            }
        }
    }
}

// order of tf.math.top_k: larger similarity first, ties -> lower index first
__device__ inline bool topk_before(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

// Top-k of a query's candidates by selection.  Workgroup = (query, segment of kTopkSeg candidates): every thread
// holds kTopkPer of them in registers and the workgroup runs k rounds of arg-max over the threads' best remaining
// candidates in the order of tf.math.top_k — one barrier per round, nothing sorted and no memory traffic after the
// first load.  A vocabulary larger than one segment is reduced in stages: each launch turns
// `len` candidates per query into ceil(len / kTopkSeg) * k winners (values + vocabulary ids) until one segment is
// left.  (A per-thread sorted list of the best k, the first form of this kernel, cost ~2,300 instructions per
// inserted element and took 9 ms per 256 queries at V = 400 k; this takes 0.1 ms.)
constexpr int kTopkPer = 16;
constexpr int kTopkSeg = kBlock * kTopkPer;

__global__ __launch_bounds__(kBlock) void topk_select_kernel(const float *__restrict__ vals,
                                                             const int32_t *__restrict__ ids, int64_t row_stride,
                                                             int32_t len, int32_t k, float *__restrict__ out_val,
                                                             int32_t *__restrict__ out_idx)
{
    __shared__ float s_val[2][kBlock / 64];
    __shared__ int s_idx[2][kBlock / 64];
    const float *row = vals + (size_t)blockIdx.x * row_stride;
    const int32_t *row_ids = ids ? ids + (size_t)blockIdx.x * row_stride : nullptr;
    const size_t out_row = ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * k;
    float sv[kTopkPer];
    int id[kTopkPer];
#pragma unroll
    for (int e = 0; e < kTopkPer; ++e) {
        const int p = blockIdx.y * kTopkSeg + e * kBlock + threadIdx.x;
        const bool in = p < len;
        const int v = in ? (row_ids ? row_ids[p] : p) : -1;
        sv[e] = (in && v >= 0) ? row[p] : -INFINITY;       // v < 0: an empty slot of a short earlier segment
        id[e] = v >= 0 ? v : 0x7fffffff;
    }
    // every thread keeps its best remaining candidate; a round is one workgroup arg-max over those 256, and only the
    // thread that owned the winner rescans its 16 registers for its next best (the rest of its wave idles through it):
    // ~260 wave-instructions per round instead of ~1,000 when every thread rescanned every round
    float mine = -INFINITY;
    int mine_id = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < kTopkPer; ++e)
        if (topk_before(sv[e], id[e], mine, mine_id)) { mine = sv[e]; mine_id = id[e]; }
    for (int t = 0; t < k; ++t) {
        float best = mine;
        int bi = mine_id;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (topk_before(ob, oi, best, bi)) { best = ob; bi = oi; }
        }
        const int buf = t & 1;                              // double-buffered: one barrier per round
        if ((threadIdx.x & 63) == 0) { s_val[buf][threadIdx.x >> 6] = best; s_idx[buf][threadIdx.x >> 6] = bi; }
        __syncthreads();
        best = s_val[buf][0];
        bi = s_idx[buf][0];
#pragma unroll
        for (int wv = 1; wv < kBlock / 64; ++wv)
            if (topk_before(s_val[buf][wv], s_idx[buf][wv], best, bi)) { best = s_val[buf][wv]; bi = s_idx[buf][wv]; }
        const bool none = bi == 0x7fffffff;                 // fewer than k candidates in this segment
        if (threadIdx.x == 0) {
            out_val[out_row + t] = none ? -INFINITY : best;
            out_idx[out_row + t] = none ? -1 : bi;
        }
        if (!none && mine_id == bi) {                       // ids are unique: exactly one thread owned the winner
            mine = -INFINITY;
            mine_id = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < kTopkPer; ++e) {
                const bool behind = sv[e] < best || (sv[e] == best && id[e] > bi);
                if (behind && topk_before(sv[e], id[e], mine, mine_id)) { mine = sv[e]; mine_id = id[e]; }
            }
        }
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

// winners the first top-k stage leaves per query (the later stages only shrink)
static size_t topk_stage_items(int32_t V, int32_t k) { return (size_t)((V + kTopkSeg - 1) / kTopkSeg) * (size_t)k; }

size_t glove_topk_workspace_bytes(int32_t n, int32_t V, int32_t k)
{
    if (n < 0 || V <= 0 || k < 0) return 0;
    const size_t cand = (size_t)n * topk_stage_items(V, k);                     // two ping-pong buffers of winners
    return align_up((size_t)V * sizeof(float), 256) + align_up((size_t)n * V * sizeof(float), 256) +
           2 * (align_up(cand * sizeof(float), 256) + align_up(cand * sizeof(int32_t), 256));
}

int glove_topk_cosine_f32(const float *R, int32_t V, int32_t d, const int32_t *query_ids, int32_t n, int32_t k,
                          float *sims_out, int32_t *idx_out, void *ws, size_t ws_bytes, void *stream)
{
    if (!R || V <= 0 || d <= 0 || (d % 4) != 0 || n < 0 || n > 65535 * kSimTile || k <= 0 || k > V || k > 1024) return GLOVE_E_BADARG;
    if (n == 0) return 0;
    if (!query_ids || !sims_out || !idx_out || !ws) return GLOVE_E_BADARG;
    if (glove_topk_workspace_bytes(n, V, k) > ws_bytes) return GLOVE_E_WORKSPACE;
    float *inv_norm = (float *)ws;
    float *sims = (float *)((char *)ws + align_up((size_t)V * sizeof(float), 256));
    const int d4 = d / 4;
    const RowShape shape = pick_row_shape(d4);
    if (shape.lpr == 0) return GLOVE_E_BADARG;
    const int nbv = blocks_for(V, kBlock / shape.lpr);
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV) hipLaunchKernelGGL((inv_norm_kernel<LPR, NV>), dim3(nbv), dim3(kBlock), 0, st, R, V, d4, inv_norm)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    hipLaunchKernelGGL(cosine_mfma_kernel, dim3((V + kSimTile - 1) / kSimTile, (n + kSimTile - 1) / kSimTile), dim3(kBlock),
                       0, st, R, V, d, query_ids, n, inv_norm, sims);
    // reduce in stages until one segment holds a query's candidates; the last stage writes the outputs
    const size_t cand = (size_t)n * topk_stage_items(V, k);
    char *pp = (char *)sims + align_up((size_t)n * V * sizeof(float), 256);
    float *cv[2];
    int32_t *ci[2];
    for (int b = 0; b < 2; ++b) {
        cv[b] = (float *)pp;
        pp += align_up(cand * sizeof(float), 256);
        ci[b] = (int32_t *)pp;
        pp += align_up(cand * sizeof(int32_t), 256);
    }
    const float *src_v = sims;
    const int32_t *src_i = nullptr;
    int64_t stride = V;
    int32_t len = V;
    for (int stage = 0;; ++stage) {
        const int nseg = (len + kTopkSeg - 1) / kTopkSeg;
        const bool last = nseg == 1;
        float *dv = last ? sims_out : cv[stage & 1];
        int32_t *di = last ? idx_out : ci[stage & 1];
        hipLaunchKernelGGL(topk_select_kernel, dim3(n, nseg), dim3(kBlock), 0, st, src_v, src_i, stride, len, k, dv, di);
        if (last) break;
        src_v = dv;
        src_i = di;
        stride = len = nseg * k;
    }
    return (int)hipGetLastError();
}

}  // extern "C"
