// Single-workgroup index build for small batches (B <= kSmallPlanMax).
//
// glove_plan_build's general path is ~20 dependent rocPRIM launches: launch-bound (~130 us) for the
// reference's default batch of 1,024 nonzeros, where one step of the kernels takes ~10 us.  A caller
// that hands over a fresh batch every step (the reference's input_fn does: data_utils.py:12-21) needs
// the index in a few microseconds, so batches up to 4,096 pairs are indexed by ONE workgroup entirely
// in LDS: two stable block radix sorts by id (hand-written, the ballot-rank scheme of glove_plan.hip's tiled sort on one
// workgroup of 16 waves: passes of up to 8 bits, two for a 10^4-id vocabulary; earlier forms: rocPRIM's block primitive,
// and before it a bitonic sort of 64-bit (id, position) keys that spent 70 of its 110 us at B = 4,096 in its 78
// LDS-bound sub-stages) and three block scans per side.  The result is identical to the general path and to
// oracle/glove_ref.py:build_plan.
#include "glove_common.h"

namespace glove {

constexpr int kSmallThreads = 1024;
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

// ---- stable sort of up to kSmallThreads x E (id, position) pairs by id, one workgroup, in LDS -------------------------
// LSD radix, P = ceil(bits / 8) passes of ceil(bits / P)-bit digits.  A pass: every wave takes 64 E consecutive positions,
// 64 per round; the lanes of a round that hold the same digit find each other with `db` ballots, the group's first lane
// fetch-adds the wave's running count of that digit (LDS atomic with return) and hands the old value round: a key's stable
// rank inside its wave.  Thread d then turns column d of the 16 wave counters into the waves' starting offsets, a scan over
// the digits gives where each digit starts, and every pair moves to its place through LDS.  Four barriers per pass.
constexpr int kSmallWaves = kSmallThreads / 64;
constexpr int kSmallDigits = 256;

struct SmallSortLds {
    int wcnt[kSmallWaves][kSmallDigits];     // per wave: running digit counts, then the wave's offset inside the digit
    int dig[kSmallDigits];                   // where the keys of a digit start
    int red[kSmallWaves];
};

template <int E>
constexpr size_t small_sort_bytes() { return (size_t)2 * kSmallThreads * E * 4 + sizeof(SmallSortLds); }

// key / pos: blocked arrangement in and out (thread t holds positions t E .. t E + E - 1); n = valid pairs (the first n
// positions); the others keep their place behind them
template <int E>
__device__ inline void block_sort_pairs(uint32_t (&key)[E], int32_t (&pos)[E], int n, int bits, unsigned char *scratch)
{
    constexpr int np = kSmallThreads * E;
    uint32_t *kbuf = reinterpret_cast<uint32_t *>(scratch);
    int32_t *vbuf = reinterpret_cast<int32_t *>(kbuf + np);
    SmallSortLds &L = *reinterpret_cast<SmallSortLds *>(vbuf + np);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        kbuf[threadIdx.x * E + e] = key[e];
        vbuf[threadIdx.x * E + e] = pos[e];
    }
    __syncthreads();
    const int P = (bits + 7) / 8, db = (bits + P - 1) / P, nd = 1 << db;
    for (int p = 0; p < P; ++p) {
        const int shift = p * db;
        uint32_t k[E];
        int32_t v[E];
        int rank[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int i = wave * 64 * E + j * 64 + lane;
            k[j] = kbuf[i];
            v[j] = vbuf[i];
        }
        for (int i = lane; i < nd; i += 64) L.wcnt[wave][i] = 0;          // a wave's own counters: LDS ops of one wave are ordered
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool valid = wave * 64 * E + j * 64 + lane < n;
            const int digit = (int)(k[j] >> shift) & (nd - 1);
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < db; ++b) {
                const bool bit = (digit >> b) & 1;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
            int before = 0;
            if (valid && lane == leader) before = atomicAdd(&L.wcnt[wave][digit], __popcll(peers));
            before = __shfl(before, leader, 64);
            rank[j] = before + __popcll(peers & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        int total = 0;
        if ((int)threadIdx.x < nd) {
#pragma unroll
            for (int wv = 0; wv < kSmallWaves; ++wv) {
                const int c = L.wcnt[wv][threadIdx.x];
                L.wcnt[wv][threadIdx.x] = total;
                total += c;
            }
        }
        int incl = total;                                                   // exclusive scan over the digits (threads)
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int o = __shfl_up(incl, dlt, 64);
            if (lane >= dlt) incl += o;
        }
        if (lane == 63) L.red[wave] = incl;
        __syncthreads();
        int start = incl - total;
        for (int wv = 0; wv < wave; ++wv) start += L.red[wv];
        if ((int)threadIdx.x < nd) L.dig[threadIdx.x] = start;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (wave * 64 * E + j * 64 + lane >= n) continue;
            const int digit = (int)(k[j] >> shift) & (nd - 1);
            const int dest = L.dig[digit] + L.wcnt[wave][digit] + rank[j];
            kbuf[dest] = k[j];
            vbuf[dest] = v[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        key[e] = kbuf[threadIdx.x * E + e];
        pos[e] = vbuf[threadIdx.x * E + e];
    }
    __syncthreads();                                                        // the scratch area is free again
}

// dynamic LDS: four int arrays of np (row ids / col ids / w / y in sorted order) followed by a scratch area that is
// the sort's storage during the sorts and three int arrays of np (scan scratch) between them
template <int E>
constexpr size_t small_scratch_bytes()
{
    return small_sort_bytes<E>() > (size_t)3 * kSmallThreads * E * 4 ? small_sort_bytes<E>() : (size_t)3 * kSmallThreads * E * 4;
}

template <int E>
__global__ __launch_bounds__(kSmallThreads) void plan_small_kernel(
    const int32_t *__restrict__ row, const int32_t *__restrict__ col, const float *__restrict__ w,
    const float *__restrict__ y, int B, int V, int key_bits, glove_plan plan)
{
    constexpr int np = kSmallThreads * E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *srow = reinterpret_cast<int *>(smem);                           // [np] row ids, row-sorted
    int *scol = srow + np;                                               // [np] col ids, row-sorted; then col-sorted
    float *sw = reinterpret_cast<float *>(scol + np);                    // [np] w, row-sorted
    float *sy = sw + np;                                                 // [np]
    unsigned char *scratch = reinterpret_cast<unsigned char *>(sy + np);
    int *sa = reinterpret_cast<int *>(scratch);                          // scan scratch, live between the sorts
    int *sb = sa + np;
    int *sc = sb + np;
    __shared__ int wave_tot[16];
    if (threadIdx.x < 8) plan.counts[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t pad_key = 1u << (key_bits - 1);                       // positions behind the batch (never moved by the sorts)

    // ---- row side: stable sort by row id (blocked arrangement: thread t holds positions t E .. t E + E - 1);
    // ids outside [0, V) count as id 0 (see prepare_ids in glove_plan.hip)
    uint32_t key[E];
    int32_t pos[E];
    int mapped = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = threadIdx.x * E + e;
        key[e] = pad_key;
        pos[e] = i;
        if (i < B) {
            uint32_t r = (uint32_t)row[i];
            if (r >= (uint32_t)(plan.V_row > 0 ? plan.V_row : V)) { r = 0; ++mapped; }
            key[e] = r;
        }
    }
    block_sort_pairs<E>(key, pos, B, key_bits - 1, scratch);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = threadIdx.x * E + e;
        if (k >= B) continue;
        const int p = pos[e];
        int c = col[p];
        if ((uint32_t)c >= (uint32_t)V) { c = 0; ++mapped; }
        const float wv = w[p], yv = y[p];
        srow[k] = (int)key[e];
        scol[k] = c; sw[k] = wv; sy[k] = yv;
        plan.r_partner[k] = c; plan.r_w[k] = wv; plan.r_y[k] = yv;
    }
    if (mapped) atomicAdd(plan.counts + 5, mapped);
    __syncthreads();
    small_side(srow, B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, 0, sa, sb, sc, wave_tot,
               SmallSideOut{plan.r_chunk_id, plan.r_chunk_start, plan.r_uniq_slot, plan.r_uniq_rec}, plan.counts + 0,
               plan.heavy, plan.counts + 4);

    // ---- col side: stable sort of the row-sorted pairs by col id
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = threadIdx.x * E + e;
        key[e] = i < B ? (uint32_t)scol[i] : pad_key;
        pos[e] = i;
    }
    __syncthreads();                                                     // scol read, scan scratch dead: storage free
    block_sort_pairs<E>(key, pos, B, key_bits - 1, scratch);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = threadIdx.x * E + e;
        if (j >= B) continue;
        const int p = pos[e];
        plan.c_perm[j] = p;
        plan.r_to_c[p] = j;
        plan.c_partner[j] = srow[p];
        plan.c_w[j] = sw[p];
        plan.c_y[j] = sy[p];
    }
    __syncthreads();                                                     // srow / sw / sy gathers done before scol changes
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = threadIdx.x * E + e;
        if (j < B) scol[j] = (int)key[e];                                // col ids, col-sorted
    }
    __syncthreads();
    small_side(scol, B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, 1, sa, sb, sc, wave_tot,
               SmallSideOut{plan.c_chunk_id, plan.c_chunk_start, plan.c_uniq_slot, plan.c_uniq_rec}, plan.counts + 2,
               plan.heavy, plan.counts + 4);
}

template <int E>
static int launch_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                        const glove_plan *plan, hipStream_t st)
{
    const size_t smem = (size_t)4 * kSmallThreads * E * 4 + small_scratch_bytes<E>();
    int key_bits = 2;                                                    // ids < 2^(key_bits-1), padding key 2^(key_bits-1)
    while (key_bits < 32 && (1u << (key_bits - 1)) < (uint32_t)V) ++key_bits;
    // above the 64 KiB default of dynamic LDS: the limit is raised explicitly
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_kernel<E>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(plan_small_kernel<E>, dim3(1), dim3(kSmallThreads), smem, st, row, col, w, y, (int)B, (int)V,
                       key_bits, *plan);
    return (int)hipGetLastError();
}

// host side: called from glove_plan_build for B <= kSmallPlanMax
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, hipStream_t st)
{
    if (B <= kSmallThreads) return launch_small<1>(row, col, w, y, B, V, plan, st);
    if (B <= 2 * kSmallThreads) return launch_small<2>(row, col, w, y, B, V, plan, st);
    return launch_small<4>(row, col, w, y, B, V, plan, st);
}

}  // namespace glove
