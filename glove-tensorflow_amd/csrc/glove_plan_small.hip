// Single-workgroup index build for small batches (B <= kSmallPlanMax).
//
// glove_plan_build's general path is ~20 dependent rocPRIM launches: launch-bound (~130 us) for the
// reference's default batch of 1,024 nonzeros, where one step of the kernels takes ~10 us.  A caller
// that hands over a fresh batch every step (the reference's input_fn does: data_utils.py:12-21) needs
// the index in a few microseconds, so batches up to 4,096 pairs are indexed by ONE workgroup entirely
// in LDS: two bitonic sorts of (id << 13 | position) keys — stable by construction — and three block
// scans per side.  The result is identical to the general path and to oracle/glove_ref.py:build_plan.
#include "glove_common.h"

namespace glove {

constexpr int kSmallThreads = 1024;
constexpr int kPosBits = 13;                      // position < 8192
constexpr uint64_t kPosMask = (1ull << kPosBits) - 1;

// in-place ascending bitonic sort of np (power of two) 64-bit keys in LDS by the whole workgroup
__device__ inline void bitonic_sort(uint64_t *keys, int np)
{
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < np / 2; t += kSmallThreads) {
                // t-th compare-exchange of this stage: lower index i has bit j clear
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const uint64_t a = keys[i], b = keys[l];
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }
}

// inclusive scan of n ints in LDS (in place) by the whole workgroup; op: 0 = sum, 1 = max
template <int OP>
__device__ inline void block_scan(int *v, int n, int *wave_tot /* [16] */)
{
    const int per = (n + kSmallThreads - 1) / kSmallThreads;
    const int lo = threadIdx.x * per, hi = min(lo + per, n);
    int acc = 0;                                        // identity of both ops on non-negative data
    for (int i = lo; i < hi; ++i) { acc = OP ? max(acc, v[i]) : acc + v[i]; v[i] = acc; }
    // scan the per-thread totals: inside the wave, then across the 16 waves
    int x = acc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x = OP ? max(x, y) : x + y;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    int base = 0;
    for (int wv = 0; wv < wave; ++wv) base = OP ? max(base, wave_tot[wv]) : base + wave_tot[wv];
    const int excl = OP ? max(base, __shfl_up(x, 1, 64)) : base + __shfl_up(x, 1, 64);
    const int offset = lane == 0 ? base : excl;          // total of everything before this thread
    for (int i = lo; i < hi; ++i) v[i] = OP ? max(v[i], offset) : v[i] + offset;
    __syncthreads();
}

struct SmallSideOut {
    int32_t *chunk_id, *chunk_start, *uniq_slot, *uniq_rec;
};

// keys_sorted[k] (ids in sorted order, LDS) -> chunk / uniq arrays of one side.  a, b, c: int scratch [np].
__device__ inline void small_side(const int *ids, int B, int cap, int heavy_chunks, int cap_heavy, int side,
                                  int *a, int *b, int *c, int *wave_tot, const SmallSideOut &o,
                                  int32_t *counts /* [0] chunks [1] uniq */, int32_t *heavy, int32_t *n_heavy)
{
    // a = start position of the run each element belongs to
    for (int k = threadIdx.x; k < B; k += kSmallThreads) a[k] = (k == 0 || ids[k] != ids[k - 1]) ? k : 0;
    __syncthreads();
    block_scan<1>(a, B, wave_tot);
    // b = is-chunk-start flag, c = is-new-id flag; then inclusive sums
    for (int k = threadIdx.x; k < B; k += kSmallThreads) {
        const int uniq = (k == 0 || ids[k] != ids[k - 1]) ? 1 : 0;
        c[k] = uniq;
        b[k] = (uniq || ((k - a[k]) % cap == 0)) ? 1 : 0;
    }
    __syncthreads();
    // remember the flags in `a` (bit 0 chunk, bit 1 uniq) before the scans overwrite them
    for (int k = threadIdx.x; k < B; k += kSmallThreads) a[k] = b[k] | (c[k] << 1);
    __syncthreads();
    block_scan<0>(b, B, wave_tot);
    block_scan<0>(c, B, wave_tot);
    const int n_chunks = b[B - 1], n_uniq = c[B - 1];
    for (int k = threadIdx.x; k < B; k += kSmallThreads) {
        const int ci = b[k] - 1, ui = c[k] - 1;
        if (a[k] & 1) { o.chunk_id[ci] = ids[k]; o.chunk_start[ci] = k; }
        if (a[k] & 2) o.uniq_slot[ui] = ci;
    }
    if (threadIdx.x == 0) {
        o.chunk_start[n_chunks] = B;
        o.uniq_slot[n_uniq] = n_chunks;
        counts[0] = n_chunks;
        counts[1] = n_uniq;
    }
    __syncthreads();
    // {id, first chunk, chunks, pairs} per distinct id + heavy list: positions of the id starts are the
    // elements with the uniq flag; the next id's start closes the record
    for (int k = threadIdx.x; k < B; k += kSmallThreads) {
        if (!(a[k] & 2)) continue;
        const int ui = c[k] - 1, first = b[k] - 1;
        // end of this id's run: next uniq start, found by walking chunk starts is costly; use the run-start
        // scan instead: the run of element B-1 backwards is not needed — look ahead with the chunk index
        int next_first, next_pos;
        {
            // binary search for the first position p > k with (a[p] & 2)
            int lo = k + 1, hi = B;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (c[mid] - 1 > ui) hi = mid; else lo = mid + 1;
            }
            next_pos = lo;
            next_first = next_pos < B ? b[next_pos] - 1 : n_chunks;
        }
        const int nch = next_first - first;
        reinterpret_cast<int4 *>(o.uniq_rec)[ui] = make_int4(ids[k], first, nch, next_pos - k);
        if (nch > heavy_chunks) {
            const int slot = atomicAdd(n_heavy, 1);
            if (slot < cap_heavy) heavy[slot] = (side << 30) | ui;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kSmallThreads) void plan_small_kernel(
    const int32_t *__restrict__ row, const int32_t *__restrict__ col, const float *__restrict__ w,
    const float *__restrict__ y, int B, int V, int np, glove_plan plan)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);                 // [np]
    int *srow = reinterpret_cast<int *>(keys + np);                      // [np] row ids, row-sorted
    int *scol = srow + np;                                               // [np] col ids, row-sorted; then col-sorted
    float *sw = reinterpret_cast<float *>(scol + np);                    // [np] w, row-sorted
    float *sy = sw + np;                                                 // [np]
    int *sa = reinterpret_cast<int *>(sy + np);                          // scan scratch
    int *sb = sa + np;
    int *sc = sb + np;
    __shared__ int wave_tot[16];
    if (threadIdx.x < 8) plan.counts[threadIdx.x] = 0;
    __syncthreads();

    // ---- row side: stable sort by (row id, position); ids outside [0, V) count as id 0 (see prepare_ids)
    int mapped = 0;
    for (int i = threadIdx.x; i < np; i += kSmallThreads) {
        uint64_t key = ~0ull;
        if (i < B) {
            uint32_t r = (uint32_t)row[i];
            if (r >= (uint32_t)V) { r = 0; ++mapped; }
            key = ((uint64_t)r << kPosBits) | (uint64_t)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort(keys, np);
    for (int k = threadIdx.x; k < B; k += kSmallThreads) {
        const int p = (int)(keys[k] & kPosMask);
        int c = col[p];
        if ((uint32_t)c >= (uint32_t)V) { c = 0; ++mapped; }
        const float wv = w[p], yv = y[p];
        srow[k] = (int)(keys[k] >> kPosBits);
        scol[k] = c; sw[k] = wv; sy[k] = yv;
        plan.r_partner[k] = c; plan.r_w[k] = wv; plan.r_y[k] = yv;
    }
    if (mapped) atomicAdd(plan.counts + 5, mapped);
    __syncthreads();
    small_side(srow, B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, 0, sa, sb, sc, wave_tot,
               SmallSideOut{plan.r_chunk_id, plan.r_chunk_start, plan.r_uniq_slot, plan.r_uniq_rec}, plan.counts + 0,
               plan.heavy, plan.counts + 4);

    // ---- col side: stable sort of the row-sorted pairs by (col id, row-sorted position)
    for (int i = threadIdx.x; i < np; i += kSmallThreads)
        keys[i] = i < B ? (((uint64_t)(uint32_t)scol[i] << kPosBits) | (uint64_t)i) : ~0ull;
    __syncthreads();
    bitonic_sort(keys, np);
    for (int j = threadIdx.x; j < B; j += kSmallThreads) {
        const int p = (int)(keys[j] & kPosMask);
        plan.c_perm[j] = p;
        plan.r_to_c[p] = j;
        plan.c_partner[j] = srow[p];
        plan.c_w[j] = sw[p];
        plan.c_y[j] = sy[p];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < B; j += kSmallThreads) scol[j] = (int)(keys[j] >> kPosBits);   // col ids, col-sorted
    __syncthreads();
    small_side(scol, B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, 1, sa, sb, sc, wave_tot,
               SmallSideOut{plan.c_chunk_id, plan.c_chunk_start, plan.c_uniq_slot, plan.c_uniq_rec}, plan.counts + 2,
               plan.heavy, plan.counts + 4);
}

// host side: called from glove_plan_build for B <= kSmallPlanMax
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, hipStream_t st)
{
    int np = 64;
    while (np < B) np <<= 1;
    const size_t smem = (size_t)np * (8 + 4 * 7);
    // up to 144 KiB of the CU's 160 KiB LDS: above the 64 KiB default, so the limit is raised explicitly
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(plan_small_kernel, dim3(1), dim3(kSmallThreads), smem, st, row, col, w, y, (int)B, (int)V, np, *plan);
    return (int)hipGetLastError();
}

}  // namespace glove
