// Per-batch dedup index ("plan") built on the device.
//
// Replaces the bookkeeping half of OptimizerV2._resource_apply_sparse_duplicate_indices, i.e. the
// tf.unique + unsorted_segment_sum pair Keras runs on each of the four IndexedSlices gradients
// of every step (reached from reference src/models/train_utils.py:13-16).  The
// sums themselves happen in glove_step.hip; this file only orders the pairs.
//
// Integer work: stable LSD radix sorts (rocPRIM device primitives) + two scans per side.  The
// result is bit-exact against oracle/glove_ref.py:build_plan.
#include "glove_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

namespace glove {

// glove_plan_small.hip: one-workgroup build for batches of at most kSmallPlanMax pairs
constexpr int kSmallPlanMax = 4096;
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, hipStream_t st);

// positions 0..n-1, and copies of the ids with anything outside [0, V) mapped to 0 — the id the reference's
// vocabulary lookup gives an unknown token (reference src/models/estimator.py:26-28) — so that no later kernel
// can index outside the tables whatever the caller hands over; counts[5] reports how many were mapped
__global__ void prepare_ids(const int32_t *__restrict__ row, const int32_t *__restrict__ col, int64_t n, int32_t V,
                            int32_t *__restrict__ iota, int32_t *__restrict__ row_clean,
                            int32_t *__restrict__ col_clean, int32_t *__restrict__ n_mapped)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)row[i], c = (uint32_t)col[i];
        iota[i] = (int32_t)i;
        row_clean[i] = r < (uint32_t)V ? (int32_t)r : 0;
        col_clean[i] = c < (uint32_t)V ? (int32_t)c : 0;
        bad += (r >= (uint32_t)V) + (c >= (uint32_t)V);
    }
    if (bad) atomicAdd(n_mapped, bad);
}

// row side: pull col/w/y through the row-sort permutation
__global__ void gather_row_side(const int32_t *__restrict__ perm, const int32_t *__restrict__ col,
                                const float *__restrict__ w, const float *__restrict__ y, int64_t n,
                                int32_t *__restrict__ partner, float *__restrict__ ow, float *__restrict__ oy)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = perm[i];
        partner[i] = col[p];
        ow[i] = w[p];
        oy[i] = y[p];
    }
}

// col side: partner = row id of the row-sorted pair the permutation points at; r_to_c = inverse
__global__ void gather_col_side(const int32_t *__restrict__ perm, const int32_t *__restrict__ sorted_row,
                                const float *__restrict__ r_w, const float *__restrict__ r_y, int64_t n,
                                int32_t *__restrict__ partner, int32_t *__restrict__ r_to_c,
                                float *__restrict__ c_w, float *__restrict__ c_y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = perm[i];
        partner[i] = sorted_row[p];
        r_to_c[p] = (int32_t)i;
        c_w[i] = r_w[p];
        c_y[i] = r_y[p];
    }
}

// seg_start_in[k] = k where a new id starts, else 0 (max-scan turns it into "start of my run")
__global__ void mark_runs(const int32_t *__restrict__ keys, int64_t n, int32_t *__restrict__ run_start)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        run_start[i] = (i == 0 || keys[i] != keys[i - 1]) ? (int32_t)i : 0;
}

// flags packed as (is_unique << 32) | is_chunk so that one 64-bit sum-scan numbers both
__global__ void mark_chunks(const int32_t *__restrict__ keys, const int32_t *__restrict__ run_start, int64_t n,
                            int32_t chunk_cap, uint64_t *__restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool uniq = (i == 0 || keys[i] != keys[i - 1]);
        const bool chunk = uniq || ((i - run_start[i]) % chunk_cap == 0);
        flags[i] = ((uint64_t)uniq << 32) | (uint64_t)chunk;
    }
}

__global__ void emit_side(const int32_t *__restrict__ keys, const uint64_t *__restrict__ flags,
                          const uint64_t *__restrict__ scanned, int64_t n, int32_t *__restrict__ chunk_id,
                          int32_t *__restrict__ chunk_start, int32_t *__restrict__ uniq_slot,
                          int32_t *__restrict__ counts /* [0]=chunks [1]=uniq */)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t f = flags[i], sc = scanned[i];
        const int32_t ci = (int32_t)(sc & 0xffffffffu) - 1;
        const int32_t ui = (int32_t)(sc >> 32) - 1;
        if (f & 1u) { chunk_id[ci] = keys[i]; chunk_start[ci] = (int32_t)i; }
        if (f >> 32) uniq_slot[ui] = ci;
        if (i == n - 1) {
            chunk_start[ci + 1] = (int32_t)n;
            uniq_slot[ui + 1] = ci + 1;
            counts[0] = ci + 1;
            counts[1] = ui + 1;
        }
    }
}

// {id, first chunk, chunks, pairs} per distinct id, from the arrays emit_side wrote
// also appends the ids with more than heavy_chunks chunks to the plan's heavy list (any order)
__global__ void emit_uniq_rec(const int32_t *__restrict__ counts, const int32_t *__restrict__ chunk_id,
                              const int32_t *__restrict__ chunk_start, const int32_t *__restrict__ uniq_slot,
                              int32_t *__restrict__ rec, int side, int heavy_chunks, int cap_heavy,
                              int32_t *__restrict__ heavy, int32_t *__restrict__ n_heavy)
{
    const int nu = counts[1];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nu; q += gridDim.x * blockDim.x) {
        const int a = uniq_slot[q], b = uniq_slot[q + 1];
        reinterpret_cast<int4 *>(rec)[q] = make_int4(chunk_id[a], a, b - a, chunk_start[b] - chunk_start[a]);
        if (b - a > heavy_chunks) {
            const int slot = atomicAdd(n_heavy, 1);
            if (slot < cap_heavy) heavy[slot] = (side << 30) | q;
        }
    }
}

// per-chunk records: one thread per (chunk, float4 of the record)
__global__ void fill_records(const int32_t *__restrict__ counts, int count_index, const int32_t *__restrict__ chunk_id,
                             const int32_t *__restrict__ chunk_start, const int32_t *__restrict__ partner,
                             const float *__restrict__ w, const float *__restrict__ y, int capP,
                             int32_t *__restrict__ crec)
{
    const int n_chunks = counts[count_index];
    const int rq = 1 + 3 * capP / 4;                       // float4 per record
    const int64_t total = (int64_t)n_chunks * rq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / rq), f = (int)(i - (int64_t)j * rq);
        const int s = chunk_start[j], n = chunk_start[j + 1] - s;
        int4 v;
        if (f == 0) {
            v = make_int4(chunk_id[j], n, s, 0);
        } else {
            const int field = (f - 1) / (capP / 4), t0 = ((f - 1) % (capP / 4)) * 4;
            int o[4];
            for (int x = 0; x < 4; ++x) {
                const int t = t0 + x;
                const int k = s + (t < n ? t : 0);              // padding replays pair 0 with weight 0
                o[x] = field == 0 ? partner[k] : field == 1 ? __float_as_int(t < n ? w[k] : 0.f) : __float_as_int(y[k]);
            }
            v = make_int4(o[0], o[1], o[2], o[3]);
        }
        reinterpret_cast<int4 *>(crec)[i] = v;
    }
}

static int launch_fill_records(const glove_plan *plan, hipStream_t st)
{
    const int capP = (plan->chunk_cap + 7) & ~7;      // a trip of the pass kernel reads up to 8 slots from q0
    const int64_t work = (int64_t)plan->cap_chunks * (1 + 3 * capP / 4);
    const int nb = blocks_for(work, kBlock);
    hipLaunchKernelGGL(fill_records, dim3(nb), dim3(kBlock), 0, st, plan->counts, 0, plan->r_chunk_id, plan->r_chunk_start,
                       plan->r_partner, plan->r_w, plan->r_y, capP, plan->r_crec);
    hipLaunchKernelGGL(fill_records, dim3(nb), dim3(kBlock), 0, st, plan->counts, 2, plan->c_chunk_id, plan->c_chunk_start,
                       plan->c_partner, plan->c_w, plan->c_y, capP, plan->c_crec);
    return (int)hipGetLastError();
}

struct PlanWs {
    int32_t *iota, *perm, *keys_sorted, *row_sorted, *run_start, *row_clean, *col_clean;
    uint64_t *flags, *scanned;
    void *prim;          // rocPRIM temporary storage
    size_t prim_bytes;
    size_t bytes;
};

static size_t prim_budget(int64_t B) { return (size_t)(4u << 20) + (size_t)B * 24; }

static PlanWs carve_plan_ws(void *ws, int64_t B)
{
    PlanWs p;
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { void *q = base + off; off += align_up(bytes, 256); return q; };
    const size_t n = (size_t)(B > 0 ? B : 1);
    p.iota = (int32_t *)take(n * 4);
    p.perm = (int32_t *)take(n * 4);
    p.keys_sorted = (int32_t *)take(n * 4);
    p.row_sorted = (int32_t *)take(n * 4);
    p.run_start = (int32_t *)take(n * 4);
    p.row_clean = (int32_t *)take(n * 4);
    p.col_clean = (int32_t *)take(n * 4);
    p.flags = (uint64_t *)take(n * 8);
    p.scanned = (uint64_t *)take(n * 8);
    p.prim_bytes = prim_budget(B);
    p.prim = take(p.prim_bytes);
    p.bytes = off;
    return p;
}

static int ceil_log2(int32_t v)
{
    int b = 1;
    while (b < 31 && (1 << b) < v) ++b;
    return b;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

static int build_side(const int32_t *keys_sorted, int64_t B, int32_t chunk_cap, const PlanWs &w, int32_t *chunk_id,
                      int32_t *chunk_start, int32_t *uniq_slot, int32_t *uniq_rec, int32_t cap_uniq, int32_t *counts,
                      int side, const glove_plan *plan, hipStream_t st)
{
    const int nb = blocks_for(B, kBlock);
    hipLaunchKernelGGL(mark_runs, dim3(nb), dim3(kBlock), 0, st, keys_sorted, B, w.run_start);
    size_t need = 0;
    HIP_TRY(rocprim::inclusive_scan(nullptr, need, w.run_start, w.run_start, (size_t)B, rocprim::maximum<int32_t>(), st));
    if (need > w.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::inclusive_scan(w.prim, need, w.run_start, w.run_start, (size_t)B, rocprim::maximum<int32_t>(), st));
    hipLaunchKernelGGL(mark_chunks, dim3(nb), dim3(kBlock), 0, st, keys_sorted, w.run_start, B, chunk_cap, w.flags);
    HIP_TRY(rocprim::inclusive_scan(nullptr, need, w.flags, w.scanned, (size_t)B, rocprim::plus<uint64_t>(), st));
    if (need > w.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::inclusive_scan(w.prim, need, w.flags, w.scanned, (size_t)B, rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(emit_side, dim3(nb), dim3(kBlock), 0, st, keys_sorted, w.flags, w.scanned, B, chunk_id,
                       chunk_start, uniq_slot, counts);
    hipLaunchKernelGGL(emit_uniq_rec, dim3(blocks_for(cap_uniq, kBlock)), dim3(kBlock), 0, st, counts, chunk_id,
                       chunk_start, uniq_slot, uniq_rec, side, plan->heavy_chunks, plan->cap_heavy, plan->heavy,
                       plan->counts + 4);
    return (int)hipGetLastError();
}

}  // namespace glove

using namespace glove;

extern "C" {

size_t glove_plan_workspace_bytes(int64_t B, int32_t V)
{
    (void)V;
    if (B < 0) return 0;
    return carve_plan_ws(nullptr, B).bytes;
}

int glove_plan_build(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, void *ws, size_t ws_bytes, void *stream)
{
    if (!plan || !ws || B < 0 || V <= 0 || plan->chunk_cap <= 0 || plan->B != B || !plan->counts) return GLOVE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        HIP_TRY(hipMemsetAsync(plan->counts, 0, 8 * sizeof(int32_t), st));
        return 0;
    }
    if (!row || !col || !w || !y) return GLOVE_E_BADARG;
    if (!plan->r_partner || !plan->r_w || !plan->r_y || !plan->r_chunk_id || !plan->r_chunk_start || !plan->r_uniq_slot ||
        !plan->c_w || !plan->c_y || !plan->r_uniq_rec || !plan->c_uniq_rec || !plan->r_to_c || !plan->c_partner || !plan->c_perm || !plan->c_chunk_id || !plan->c_chunk_start || !plan->c_uniq_slot)
        return GLOVE_E_BADARG;
    if (!plan->heavy || plan->heavy_chunks < 1 ||
        plan->cap_heavy < 2 * B / ((int64_t)plan->heavy_chunks * plan->chunk_cap) + 2)
        return GLOVE_E_WORKSPACE;
    // the chunk / uniq arrays must be able to hold the worst case (every pair its own chunk)
    if (plan->cap_chunks < B || plan->cap_uniq < (B < V ? B : V)) return GLOVE_E_WORKSPACE;
    if ((plan->r_crec == nullptr) != (plan->c_crec == nullptr)) return GLOVE_E_BADARG;
    if (B <= kSmallPlanMax) {                                                          // launch-bound regime
        if (int rc = plan_build_small(row, col, w, y, B, V, plan, st)) return rc;
        return plan->r_crec ? launch_fill_records(plan, st) : 0;
    }
    HIP_TRY(hipMemsetAsync(plan->counts, 0, 8 * sizeof(int32_t), st));
    const PlanWs pw = carve_plan_ws(ws, B);
    if (pw.bytes > ws_bytes) return GLOVE_E_WORKSPACE;

    const int nb = blocks_for(B, kBlock);
    const int bits = ceil_log2(V);
    size_t need = 0;

    // ---- row side: stable sort (row id, position)
    hipLaunchKernelGGL(prepare_ids, dim3(nb), dim3(kBlock), 0, st, row, col, B, V, pw.iota, pw.row_clean, pw.col_clean,
                       plan->counts + 5);
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, (const int32_t *)pw.row_clean, pw.row_sorted, pw.iota, pw.perm,
                                      (size_t)B, 0, bits, st));
    if (need > pw.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::radix_sort_pairs(pw.prim, need, (const int32_t *)pw.row_clean, pw.row_sorted, pw.iota, pw.perm,
                                      (size_t)B, 0, bits, st));
    hipLaunchKernelGGL(gather_row_side, dim3(nb), dim3(kBlock), 0, st, pw.perm, pw.col_clean, w, y, B, plan->r_partner,
                       plan->r_w, plan->r_y);
    if (int rc = build_side(pw.row_sorted, B, plan->chunk_cap, pw, plan->r_chunk_id, plan->r_chunk_start,
                            plan->r_uniq_slot, plan->r_uniq_rec, plan->cap_uniq, plan->counts + 0, 0, plan, st))
        return rc;

    // ---- col side: stable sort of the row-sorted pairs by col id
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, (const int32_t *)plan->r_partner, pw.keys_sorted, pw.iota,
                                      plan->c_perm, (size_t)B, 0, bits, st));
    if (need > pw.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::radix_sort_pairs(pw.prim, need, (const int32_t *)plan->r_partner, pw.keys_sorted, pw.iota,
                                      plan->c_perm, (size_t)B, 0, bits, st));
    hipLaunchKernelGGL(gather_col_side, dim3(nb), dim3(kBlock), 0, st, plan->c_perm, pw.row_sorted, plan->r_w, plan->r_y, B,
                       plan->c_partner, plan->r_to_c, plan->c_w, plan->c_y);
    if (int rc = build_side(pw.keys_sorted, B, plan->chunk_cap, pw, plan->c_chunk_id, plan->c_chunk_start,
                            plan->c_uniq_slot, plan->c_uniq_rec, plan->cap_uniq, plan->counts + 2, 1, plan, st))
        return rc;
    if (plan->r_crec) return launch_fill_records(plan, st);
    return (int)hipGetLastError();
}

int glove_plan_fill_records(const glove_plan *plan, void *stream)
{
    if (!plan || !plan->r_crec || !plan->c_crec || !plan->counts || plan->chunk_cap <= 0) return GLOVE_E_BADARG;
    if (plan->B == 0) return 0;
    if (!plan->r_chunk_id || !plan->r_chunk_start || !plan->r_partner || !plan->r_w || !plan->r_y || !plan->c_chunk_id ||
        !plan->c_chunk_start || !plan->c_partner || !plan->c_w || !plan->c_y)
        return GLOVE_E_BADARG;
    return launch_fill_records(plan, (hipStream_t)stream);
}

}  // extern "C"
