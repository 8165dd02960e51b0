// Per-batch dedup index ("plan") built on the device.
//
// Replaces the bookkeeping half of OptimizerV2._resource_apply_sparse_duplicate_indices, i.e. the
// tf.unique + unsorted_segment_sum pair Keras runs on each of the four IndexedSlices gradients
// of every step (reached from reference src/models/train_utils.py:13-16).  The
// sums themselves happen in glove_step.hip; this file only orders the pairs.
//
// Integer work: two stable sorts (rocPRIM device primitives) + two tile kernels that number the chunks and
// ids of both sides.  The result is bit-exact against oracle/glove_ref.py:build_plan.
#include "glove_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/functional.hpp>

namespace glove {

// glove_plan_small.hip: one-workgroup build for batches of at most kSmallPlanMax pairs
constexpr int kSmallPlanMax = 4096;
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, hipStream_t st);

// positions 0..n-1, and copies of the ids with anything outside [0, V) mapped to 0 — the id the reference's
// vocabulary lookup gives an unknown token (reference src/models/estimator.py:26-28) — so that no later kernel
// can index outside the tables whatever the caller hands over; counts[5] reports how many were mapped
__global__ void prepare_ids(const int32_t *__restrict__ row, const int32_t *__restrict__ col, int64_t n, int32_t Vr, int32_t V,
                            int32_t *__restrict__ iota, int32_t *__restrict__ row_clean,
                            int32_t *__restrict__ col_clean, int32_t *__restrict__ n_mapped)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)row[i], c = (uint32_t)col[i];
        iota[i] = (int32_t)i;
        row_clean[i] = r < (uint32_t)Vr ? (int32_t)r : 0;
        col_clean[i] = c < (uint32_t)V ? (int32_t)c : 0;
        bad += (r >= (uint32_t)Vr) + (c >= (uint32_t)V);
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

// ---- chunk / id numbering of BOTH sides in two launches --------------------------------------------------
// A side's structure follows from its sorted keys alone: position k opens a new id where keys[k] != keys[k-1], and
// a new chunk where it opens an id or lies a multiple of chunk_cap behind the start of its run.  Numbering those
// flags needs prefix sums over the whole batch; instead of two device-wide scans per side (eight launches with the
// marking kernels around them) the batch is cut into tiles of kTile positions, one workgroup each:
//   side_tiles  counts the flags of every tile.  The only thing a tile needs from outside is where the run that
//               crosses its left edge started: one binary search over the sorted keys by one thread;
//   side_emit   adds up the counts of the tiles to its left (a block reduction over at most B / kTile pairs),
//               recomputes its flags, numbers them with one block scan and writes chunk_id / chunk_start /
//               uniq_slot; the last tile also writes the totals and the closing entries.
// blockIdx.y selects the side.  Bit-exact against oracle/glove_ref.py:build_plan like the scans it replaces.
__device__ inline int wave_sum_int(int v)
{
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) v += __shfl_xor(v, dlt, 64);
    return v;
}

constexpr int kTileThreads = 256;
constexpr int kTilePer = 8;                               // consecutive positions per thread
constexpr int kTile = kTileThreads * kTilePer;

struct SideKeys { const int32_t *keys[2]; };
struct SideOut {
    int32_t *chunk_id[2], *chunk_start[2], *uniq_slot[2];
    int32_t *counts;                                      // plan counts: [0],[1] row side, [2],[3] col side
};

// flags of this thread's kTilePer positions: bit 0 = opens a chunk, bit 1 = opens an id; nu / nc = their counts.
// Leaves the (tile-local) inclusive thread prefix of both counts in lds_u / lds_c for the caller's use.
__device__ inline void tile_flags(const int32_t *__restrict__ keys, int64_t B, int64_t begin, int32_t chunk_cap,
                                  int64_t first_run_start, unsigned (&flag)[kTilePer], int &nu, int &nc, int64_t *lds_rs)
{
    // start of the run each of my positions belongs to: max-scan of "k where an id opens", seeded from the left
    const int64_t k0 = begin + (int64_t)threadIdx.x * kTilePer;
    int32_t key[kTilePer + 1];
    key[0] = (k0 > 0 && k0 <= B) ? keys[k0 - 1] : -1;
#pragma unroll
    for (int i = 0; i < kTilePer; ++i) key[i + 1] = (k0 + i < B) ? keys[k0 + i] : -1;
    int64_t my_last_start = -1;                            // last id opening among my positions
#pragma unroll
    for (int i = 0; i < kTilePer; ++i) {
        const int64_t k = k0 + i;
        if (k < B && (k == 0 || key[i + 1] != key[i])) my_last_start = k;
    }
    // exclusive max-scan of my_last_start over the threads of the workgroup (wave shuffles, then the 4 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t incl = my_last_start;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int64_t o = __shfl_up(incl, dlt, 64);
        if (lane >= dlt) incl = o > incl ? o : incl;
    }
    if (lane == 63) lds_rs[wave] = incl;
    __syncthreads();
    int64_t carry = first_run_start;
    for (int wv = 0; wv < wave; ++wv) carry = lds_rs[wv] > carry ? lds_rs[wv] : carry;
    int64_t prev = __shfl_up(incl, 1, 64);
    if (lane == 0) prev = -1;
    int64_t run_start = prev > carry ? prev : carry;       // run that is open when my first position begins
    __syncthreads();
    nu = nc = 0;
#pragma unroll
    for (int i = 0; i < kTilePer; ++i) {
        const int64_t k = k0 + i;
        flag[i] = 0;
        if (k >= B) continue;
        const bool uniq = (k == 0 || key[i + 1] != key[i]);
        if (uniq) run_start = k;
        const bool chunk = uniq || ((int)(k - run_start) % chunk_cap == 0);
        flag[i] = (chunk ? 1u : 0u) | (uniq ? 2u : 0u);
        nu += uniq;
        nc += chunk;
    }
}

// start of the run that position `pos` lies in: first index in [0, pos] holding keys[pos] (keys sorted ascending).
// Called by one whole wave: a 64-ary search, three dependent rounds of loads for any batch below 2^18 positions
// (a one-thread binary search cost this kernel 17 round trips, most of its run time).
__device__ inline int64_t run_start_of(const int32_t *__restrict__ keys, int64_t pos)
{
    const int lane = threadIdx.x & 63;
    const int32_t key = keys[pos];
    int64_t lo = 0, hi = pos;                              // answer in [lo, hi]; keys[hi] >= key throughout
    while (lo < hi) {
        // 64 probes spread over the undecided positions lo .. hi-1
        const int64_t n = hi - lo, step = (n + 63) / 64;
        int64_t q = lo + step * lane;
        if (q > hi - 1) q = hi - 1;
        const bool ge = keys[q] >= key;
        const unsigned long long m = __ballot(ge);
        if (m == 0) {                                      // every probe is below: the answer lies behind the last one
            lo = __shfl(q, 63, 64) + 1;
            continue;
        }
        const int first = __ffsll((long long)m) - 1;       // first probe at or above the key
        const int64_t q_first = __shfl(q, first, 64);
        const int64_t q_before = __shfl(q, first > 0 ? first - 1 : 0, 64);
        lo = first == 0 ? lo : q_before + 1;               // first == 0: the probe at lo itself holds it
        hi = first == 0 ? lo : q_first;
    }
    return lo;
}

__global__ __launch_bounds__(kTileThreads) void side_tiles(SideKeys sk, int64_t B, int32_t chunk_cap, int ntiles,
                                                           int64_t *__restrict__ tile_rs, int2 *__restrict__ tile_sums)
{
    __shared__ int64_t lds_rs[kTileThreads / 64 + 1];
    __shared__ int red[2][kTileThreads / 64];
    const int side = blockIdx.y, t = blockIdx.x;
    const int32_t *keys = sk.keys[side];
    const int64_t begin = (int64_t)t * kTile;
    if (threadIdx.x < 64) {                                // wave 0 searches together
        const int64_t r = begin > 0 ? run_start_of(keys, begin) : 0;
        if (threadIdx.x == 0) lds_rs[kTileThreads / 64] = r;
    }
    __syncthreads();
    const int64_t frs = lds_rs[kTileThreads / 64];
    unsigned flag[kTilePer];
    int nu, nc;
    tile_flags(keys, B, begin, chunk_cap, frs, flag, nu, nc, lds_rs);
    nu = wave_sum_int(nu);
    nc = wave_sum_int(nc);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nu; red[1][threadIdx.x >> 6] = nc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int su = 0, sc = 0;
        for (int wv = 0; wv < kTileThreads / 64; ++wv) { su += red[0][wv]; sc += red[1][wv]; }
        tile_sums[(size_t)side * ntiles + t] = make_int2(su, sc);
        tile_rs[(size_t)side * ntiles + t] = frs;
    }
}

__global__ __launch_bounds__(kTileThreads) void side_emit(SideKeys sk, int64_t B, int32_t chunk_cap, int ntiles,
                                                          const int64_t *__restrict__ tile_rs,
                                                          const int2 *__restrict__ tile_sums, SideOut out)
{
    __shared__ int64_t lds_rs[kTileThreads / 64 + 1];
    __shared__ int red[2][kTileThreads / 64];
    __shared__ int wave_tot[2][kTileThreads / 64];
    const int side = blockIdx.y, t = blockIdx.x;
    const int32_t *keys = sk.keys[side];
    const int64_t begin = (int64_t)t * kTile;
    // ids / chunks opened by the tiles to my left
    int pu = 0, pc = 0;
    for (int i = threadIdx.x; i < t; i += kTileThreads) {
        const int2 v = tile_sums[(size_t)side * ntiles + i];
        pu += v.x;
        pc += v.y;
    }
    pu = wave_sum_int(pu);
    pc = wave_sum_int(pc);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pu; red[1][threadIdx.x >> 6] = pc; }
    __syncthreads();
    int base_u = 0, base_c = 0;
    for (int wv = 0; wv < kTileThreads / 64; ++wv) { base_u += red[0][wv]; base_c += red[1][wv]; }
    unsigned flag[kTilePer];
    int nu, nc;
    tile_flags(keys, B, begin, chunk_cap, tile_rs[(size_t)side * ntiles + t], flag, nu, nc, lds_rs);
    // exclusive prefix of (nu, nc) over the threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int iu = nu, ic = nc;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int ou = __shfl_up(iu, dlt, 64), oc = __shfl_up(ic, dlt, 64);
        if (lane >= dlt) { iu += ou; ic += oc; }
    }
    if (lane == 63) { wave_tot[0][wave] = iu; wave_tot[1][wave] = ic; }
    __syncthreads();
    int ui = base_u + iu - nu, ci = base_c + ic - nc;      // numbers of my first id / chunk opening
    for (int wv = 0; wv < wave; ++wv) { ui += wave_tot[0][wv]; ci += wave_tot[1][wv]; }
    const int64_t k0 = begin + (int64_t)threadIdx.x * kTilePer;
    int32_t *chunk_id = out.chunk_id[side], *chunk_start = out.chunk_start[side], *uniq_slot = out.uniq_slot[side];
#pragma unroll
    for (int i = 0; i < kTilePer; ++i) {
        const int64_t k = k0 + i;
        if (flag[i] & 2u) uniq_slot[ui++] = ci;
        if (flag[i] & 1u) { chunk_id[ci] = keys[k]; chunk_start[ci] = (int32_t)k; ++ci; }   // keys[k]: L1-hot, read by tile_flags
        if (k == B - 1) {                                  // closing entries and totals
            chunk_start[ci] = (int32_t)B;
            uniq_slot[ui] = ci;
            out.counts[2 * side] = ci;
            out.counts[2 * side + 1] = ui;
        }
    }
}

// {id, first chunk, chunks, pairs} per distinct id, from the arrays side_emit wrote (blockIdx.y = side);
// also appends the ids with more than heavy_chunks chunks to the plan's heavy list (any order)
struct UniqRecArgs {
    const int32_t *chunk_id[2], *chunk_start[2], *uniq_slot[2];
    int32_t *rec[2];
};
__global__ void emit_uniq_rec(const int32_t *__restrict__ counts, UniqRecArgs a, int heavy_chunks, int cap_heavy,
                              int32_t *__restrict__ heavy, int32_t *__restrict__ n_heavy)
{
    const int side = blockIdx.y;
    const int nu = counts[2 * side + 1];
    const int32_t *chunk_id = a.chunk_id[side], *chunk_start = a.chunk_start[side], *uniq_slot = a.uniq_slot[side];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nu; q += gridDim.x * blockDim.x) {
        const int x = uniq_slot[q], y = uniq_slot[q + 1];
        reinterpret_cast<int4 *>(a.rec[side])[q] = make_int4(chunk_id[x], x, y - x, chunk_start[y] - chunk_start[x]);
        if (y - x > heavy_chunks) {
            const int slot = atomicAdd(n_heavy, 1);
            if (slot < cap_heavy) heavy[slot] = (side << 30) | q;
        }
    }
}

// per-chunk records of both sides (blockIdx.y = side): one thread per (chunk, float4 of the record)
struct RecordArgs {
    const int32_t *uniq_slot[2];
    const int32_t *chunk_id[2], *chunk_start[2], *partner[2];
    const float *w[2], *y[2];
    int32_t *crec[2];
};
__global__ void fill_records(const int32_t *__restrict__ counts, RecordArgs a, int capP)
{
    const int side = blockIdx.y;
    const int n_chunks = counts[2 * side];
    const int32_t *chunk_id = a.chunk_id[side], *chunk_start = a.chunk_start[side], *partner = a.partner[side];
    const float *w = a.w[side], *y = a.y[side];
    const int rq = 1 + 3 * capP / 4;                       // float4 per record
    const int64_t total = (int64_t)n_chunks * rq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / rq), f = (int)(i - (int64_t)j * rq);
        const int s = chunk_start[j], n = chunk_start[j + 1] - s;
        int4 v;
        if (f == 0) {
            // word 3: (first chunk of its id) << 31 | chunks of the same id behind this one.  The id's chunks are
            // [uniq_slot[q], uniq_slot[q + 1]) for the q found by bisection (the slots are ascending)
            const int32_t id = chunk_id[j];
            const int32_t *slot = a.uniq_slot[side];
            int lo = 0, hi = counts[2 * side + 1];             // slot[lo] <= j < slot[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (slot[mid] <= j) lo = mid; else hi = mid;
            }
            const uint32_t rem = (uint32_t)(slot[lo + 1] - 1 - j);
            // word 2: the id's position among the side's distinct ids (ascending id order)
            v = make_int4(id, n, lo, (int)(rem | (slot[lo] == j ? 0x80000000u : 0u)));
        } else {
            // blocks of kRecPad = 8 pairs, each {partner[8] | w[8] | y[8]}: float4 r of block b holds field r / 2, pairs 8 b + 4 (r % 2) ..
            const int b = (f - 1) / 6, r = (f - 1) % 6;
            const int field = r / 2, t0 = b * kRecPad + (r % 2) * 4;
            int o[4];
            for (int x = 0; x < 4; ++x) {
                const int t = t0 + x;
                const int k = s + (t < n ? t : 0);              // padding replays pair 0 with weight 0
                o[x] = field == 0 ? partner[k] : field == 1 ? __float_as_int(t < n ? w[k] : 0.f) : __float_as_int(y[k]);
            }
            v = make_int4(o[0], o[1], o[2], o[3]);
        }
        reinterpret_cast<int4 *>(a.crec[side])[i] = v;
    }
}

static int launch_fill_records(const glove_plan *plan, hipStream_t st)
{
    const int capP = rec_cap(plan->chunk_cap);        // glove_common.h: a trip of the pass kernel reads up to kRecPad slots from q0
    const int64_t work = (int64_t)plan->cap_chunks * (1 + 3 * capP / 4);
    const RecordArgs a = {{plan->r_uniq_slot, plan->c_uniq_slot}, {plan->r_chunk_id, plan->c_chunk_id}, {plan->r_chunk_start, plan->c_chunk_start},
                          {plan->r_partner, plan->c_partner}, {plan->r_w, plan->c_w}, {plan->r_y, plan->c_y},
                          {plan->r_crec, plan->c_crec}};
    hipLaunchKernelGGL(fill_records, dim3(blocks_for(work, kBlock), 2), dim3(kBlock), 0, st,
                       (const int32_t *)plan->counts, a, capP);
    return (int)hipGetLastError();
}

struct PlanWs {
    int32_t *iota, *perm, *keys_sorted, *row_sorted, *row_clean, *col_clean;
    int64_t *tile_rs;    // [2][ntiles] start of the run that crosses a tile's left edge
    int2 *tile_sums;     // [2][ntiles] (ids, chunks) opened inside a tile
    int ntiles;
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
    p.row_clean = (int32_t *)take(n * 4);
    p.col_clean = (int32_t *)take(n * 4);
    p.ntiles = (int)((n + kTile - 1) / kTile);
    p.tile_rs = (int64_t *)take((size_t)2 * p.ntiles * 8);
    p.tile_sums = (int2 *)take((size_t)2 * p.ntiles * 8);
    p.prim_bytes = prim_budget(B);
    p.prim = take(p.prim_bytes);
    p.bytes = off;
    return p;
}

// Below 2^20 items rocPRIM's radix_sort_pairs is a merge sort: a block sort of tiles, then one launch per doubling.
// The default tile is 512 items (nine launches for a 131 k batch, each a few microseconds of mostly latency);
// 8,192-item tiles (512 threads x 16) need four merges: 150 -> 133 us per dynamic step, 90 -> 67 with six builds in
// flight.  (Tiles of 4,096 / 16,384: 139 / 156 us; forcing the onesweep radix path instead: 180 us.)
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<256, 512, 16>>;

static int ceil_log2(int32_t v)
{
    int b = 1;
    while (b < 31 && (1 << b) < v) ++b;
    return b;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

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
    hipLaunchKernelGGL(prepare_ids, dim3(nb), dim3(kBlock), 0, st, row, col, B, plan->V_row > 0 ? plan->V_row : V, V, pw.iota, pw.row_clean, pw.col_clean,
                       plan->counts + 5);
    HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(nullptr, need, (const int32_t *)pw.row_clean, pw.row_sorted, pw.iota, pw.perm,
                                      (size_t)B, 0, bits, st));
    if (need > pw.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(pw.prim, need, (const int32_t *)pw.row_clean, pw.row_sorted, pw.iota, pw.perm,
                                      (size_t)B, 0, bits, st));
    hipLaunchKernelGGL(gather_row_side, dim3(nb), dim3(kBlock), 0, st, pw.perm, pw.col_clean, w, y, B, plan->r_partner,
                       plan->r_w, plan->r_y);

    // ---- col side: stable sort of the row-sorted pairs by col id
    HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(nullptr, need, (const int32_t *)plan->r_partner, pw.keys_sorted, pw.iota,
                                      plan->c_perm, (size_t)B, 0, bits, st));
    if (need > pw.prim_bytes) return GLOVE_E_WORKSPACE;
    HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(pw.prim, need, (const int32_t *)plan->r_partner, pw.keys_sorted, pw.iota,
                                      plan->c_perm, (size_t)B, 0, bits, st));
    hipLaunchKernelGGL(gather_col_side, dim3(nb), dim3(kBlock), 0, st, plan->c_perm, pw.row_sorted, plan->r_w, plan->r_y, B,
                       plan->c_partner, plan->r_to_c, plan->c_w, plan->c_y);

    // ---- chunks and ids of both sides: two launches over tiles of the sorted keys, then the id records
    const SideKeys sk = {{pw.row_sorted, pw.keys_sorted}};
    const SideOut so = {{plan->r_chunk_id, plan->c_chunk_id}, {plan->r_chunk_start, plan->c_chunk_start},
                        {plan->r_uniq_slot, plan->c_uniq_slot}, plan->counts};
    hipLaunchKernelGGL(side_tiles, dim3(pw.ntiles, 2), dim3(kTileThreads), 0, st, sk, B, plan->chunk_cap, pw.ntiles,
                       pw.tile_rs, pw.tile_sums);
    hipLaunchKernelGGL(side_emit, dim3(pw.ntiles, 2), dim3(kTileThreads), 0, st, sk, B, plan->chunk_cap, pw.ntiles,
                       (const int64_t *)pw.tile_rs, (const int2 *)pw.tile_sums, so);
    const UniqRecArgs ua = {{plan->r_chunk_id, plan->c_chunk_id}, {plan->r_chunk_start, plan->c_chunk_start},
                            {plan->r_uniq_slot, plan->c_uniq_slot}, {plan->r_uniq_rec, plan->c_uniq_rec}};
    hipLaunchKernelGGL(emit_uniq_rec, dim3(blocks_for(plan->cap_uniq, kBlock), 2), dim3(kBlock), 0, st,
                       (const int32_t *)plan->counts, ua, plan->heavy_chunks, plan->cap_heavy, plan->heavy,
                       plan->counts + 4);
    if (plan->r_crec) return launch_fill_records(plan, st);
    return (int)hipGetLastError();
}

int glove_plan_fill_records(const glove_plan *plan, void *stream)
{
    if (!plan || !plan->r_crec || !plan->c_crec || !plan->counts || plan->chunk_cap <= 0) return GLOVE_E_BADARG;
    if (plan->B == 0) return 0;
    if (!plan->r_uniq_slot || !plan->c_uniq_slot) return GLOVE_E_BADARG;
    if (!plan->r_chunk_id || !plan->r_chunk_start || !plan->r_partner || !plan->r_w || !plan->r_y || !plan->c_chunk_id ||
        !plan->c_chunk_start || !plan->c_partner || !plan->c_w || !plan->c_y)
        return GLOVE_E_BADARG;
    return launch_fill_records(plan, (hipStream_t)stream);
}

}  // extern "C"
