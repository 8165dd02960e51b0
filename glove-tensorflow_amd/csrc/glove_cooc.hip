// Windowed co-occurrence counting on gfx950 — the producer of the hot path's input format.
//
// Replaces the 85 M-row pandas self-join of the reference's data prep
// (reference src/data/text8.py:84-108: right-context pairs inside a window, value = sum of 1/distance,
// self pairs dropped, then the union with the swapped table is summed).  Integer / byte work, HBM-bound:
//   1. every (position, offset k <= context) with a != b emits BOTH orientations as one 64-bit key
//      ((a V + b) context + (k-1));
//   2. one LSD radix sort (rocPRIM) over the used key bits;
//   3. run-length encode -> n_k per (a, b, k); runs of one (a, b) are adjacent and at most `context` long;
//   4. per (a, b): count = sum_k n_k, value = sum_k n_k / k (k ascending, fp64), written in (a, b) order.
// count is bit-exact against the reference; value differs from pandas' position-order fp64 sum by rounding
// only (tests: rtol 1e-12).
#include "glove_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

namespace glove {

// slot layout: position p owns 2*context key slots; unused slots hold the sentinel (sorts last)
constexpr uint64_t kNoKey = ~0ull;

__global__ void cooc_emit_keys(const int32_t *__restrict__ tok, int64_t n, int32_t V, int32_t context,
                               uint64_t *__restrict__ keys)
{
    const int64_t total = n * context;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / context;
        const int k = (int)(i - p * context) + 1;
        uint64_t fwd = kNoKey, bwd = kNoKey;
        if (p + k < n) {
            const int64_t a = tok[p], b = tok[p + k];
            if (a != b) {
                fwd = ((uint64_t)a * V + b) * context + (k - 1);
                bwd = ((uint64_t)b * V + a) * context + (k - 1);
            }
        }
        keys[2 * i] = fwd;
        keys[2 * i + 1] = bwd;
    }
}

// head[i] = 1 where a new (a, b) starts among the unique (a, b, k) keys (the sentinel run is no head)
__global__ void cooc_mark_heads(const uint64_t *__restrict__ ukeys, const int64_t *__restrict__ n_unique,
                                int32_t context, int64_t *__restrict__ head)
{
    const int64_t m = *n_unique;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = ukeys[i];
        head[i] = (key != kNoKey && (i == 0 || ukeys[i - 1] / context != key / context)) ? 1 : 0;
    }
}

__global__ void cooc_aggregate(const uint64_t *__restrict__ ukeys, const int32_t *__restrict__ ucounts,
                               const int64_t *__restrict__ n_unique, const int64_t *__restrict__ head,
                               const int64_t *__restrict__ head_scan /* exclusive */, int32_t V, int32_t context,
                               int64_t cap, int32_t *__restrict__ out_row, int32_t *__restrict__ out_col,
                               int64_t *__restrict__ out_count, double *__restrict__ out_value,
                               int64_t *__restrict__ out_nnz)
{
    const int64_t m = *n_unique;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        const uint64_t ab = ukeys[i] / context;
        int64_t cnt = 0;
        double val = 0.0;
        for (int64_t j = i; j < m && j < i + context; ++j) {
            const uint64_t key = ukeys[j];
            if (key == kNoKey || key / context != ab) break;
            const int k = (int)(key % context) + 1;
            cnt += ucounts[j];
            val += (double)ucounts[j] / (double)k;
        }
        const int64_t o = head_scan[i];
        if (o < cap) {
            out_row[o] = (int32_t)(ab / V);
            out_col[o] = (int32_t)(ab % V);
            out_count[o] = cnt;
            out_value[o] = val;
        }
    }
    // total number of (a, b) pairs = scan at the last element + its head flag
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_nnz = m > 0 ? head_scan[m - 1] + head[m - 1] : 0;
}

struct CoocWs {
    uint64_t *keys, *keys_sorted, *ukeys;
    int32_t *ucounts;
    int64_t *n_unique, *head, *head_scan;
    void *prim;
    size_t prim_bytes, bytes;
};

static CoocWs carve_cooc_ws(void *ws, int64_t n, int32_t context)
{
    CoocWs w;
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { void *q = base + off; off += align_up(bytes, 256); return q; };
    const size_t slots = (size_t)(n > 0 ? n : 1) * context * 2;
    w.keys = (uint64_t *)take(slots * 8);
    w.keys_sorted = (uint64_t *)take(slots * 8);
    w.ukeys = w.keys;                       // the unsorted keys are dead once sorted
    w.ucounts = (int32_t *)take(slots * 4);
    w.head = (int64_t *)take(slots * 8);
    w.head_scan = (int64_t *)take(slots * 8);
    w.n_unique = (int64_t *)take(256);
    w.prim_bytes = (size_t)(16u << 20) + slots * 16;
    w.prim = take(w.prim_bytes);
    w.bytes = off;
    return w;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

}  // namespace glove

using namespace glove;

extern "C" {

size_t glove_cooc_workspace_bytes(int64_t n_tokens, int32_t context)
{
    if (n_tokens < 0 || context <= 0) return 0;
    return carve_cooc_ws(nullptr, n_tokens, context).bytes;
}

int glove_cooccurrence_i32(const int32_t *tokens, int64_t n, int32_t V, int32_t context, int32_t *out_row,
                           int32_t *out_col, int64_t *out_count, double *out_value, int64_t *out_nnz, int64_t cap,
                           void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || V <= 0 || context <= 0 || context > 255 || cap < 0 || !out_nnz || !ws) return GLOVE_E_BADARG;
    if ((double)V * (double)V * (double)context >= 9.0e18) return GLOVE_E_BADARG;      // key must fit 63 bits
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        HIP_TRY(zero_words(out_nnz, 2, st));
        return 0;
    }
    if (!tokens || !out_row || !out_col || !out_count || !out_value) return GLOVE_E_BADARG;
    const CoocWs w = carve_cooc_ws(ws, n, context);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const size_t slots = (size_t)n * context * 2;
    hipLaunchKernelGGL(cooc_emit_keys, dim3(blocks_for(n * context, kBlock)), dim3(kBlock), 0, st, tokens, n, V, context,
                       w.keys);
    // all 64 bits: the sentinel (all ones) must sort behind every real key
    size_t need = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, need, w.keys, w.keys_sorted, slots, 0, 64, st));
    if (need > w.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::radix_sort_keys(w.prim, need, w.keys, w.keys_sorted, slots, 0, 64, st));
    if (slots > 0xffffffffull) return GLOVE_E_BADARG;
    HIP_TRY(rocprim::run_length_encode(nullptr, need, w.keys_sorted, (unsigned int)slots, w.ukeys, w.ucounts,
                                       w.n_unique, st));
    if (need > w.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::run_length_encode(w.prim, need, w.keys_sorted, (unsigned int)slots, w.ukeys, w.ucounts,
                                       w.n_unique, st));
    // n_unique lives on the device: size the follow-up launches by the worst case and let them read it
    const int nb = blocks_for((int64_t)slots, kBlock);
    hipLaunchKernelGGL(cooc_mark_heads, dim3(nb), dim3(kBlock), 0, st, w.ukeys, w.n_unique, context, w.head);
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, w.head, w.head_scan, (int64_t)0, slots, rocprim::plus<int64_t>(), st));
    if (need > w.prim_bytes) return GLOVE_E_WORKSPACE;
    // entries past n_unique are stale but never read: cooc_aggregate stops at n_unique
    HIP_TRY(rocprim::exclusive_scan(w.prim, need, w.head, w.head_scan, (int64_t)0, slots, rocprim::plus<int64_t>(), st));
    hipLaunchKernelGGL(cooc_aggregate, dim3(nb), dim3(kBlock), 0, st, w.ukeys, w.ucounts, w.n_unique, w.head, w.head_scan,
                       V, context, cap, out_row, out_col, out_count, out_value, out_nnz);
    return (int)hipGetLastError();
}

}  // extern "C"
