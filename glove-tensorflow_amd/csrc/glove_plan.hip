// Per-batch dedup index ("plan") built on the device.
//
// Replaces the bookkeeping half of OptimizerV2._resource_apply_sparse_duplicate_indices, i.e. the
// tf.unique + unsorted_segment_sum pair Keras runs on each of the four IndexedSlices gradients
// of every step (reached from reference src/models/train_utils.py:13-16).  The
// sums themselves happen in glove_step.hip; this file only orders the pairs.
//
// Integer work: two stable sorts in the same launches (a hand-written tiled LSD radix sort, below) + two tile kernels that
// number the chunks and ids of both sides (+ one that fills the optional chunk records).  The result is bit-exact against
// oracle/glove_ref.py:build_plan.
#include "glove_common.h"

#include <cstring>

namespace glove {

// glove_plan_small.hip: one-workgroup build for batches of at most kSmallPlanMax pairs
constexpr int kSmallPlanMax = 4096;
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const PlanSet &set, int n, hipStream_t st);

// ---- stable LSD radix sort of (id, position), B > kSmallPlanMax ------------------------------------------------
// Both sides need the batch's pairs in (id, arrival order) order: the row side sorted by row id, the col side by col id.
// The two sorts are independent, so they share their launches: grid.y = 2 picks the side, every launch has twice the
// workgroups instead of the build having twice the launches (these kernels are short latency chains: V = 400 k, B = 1 M:
// one workgroup per SIMD row either way).  Keys are ids below V, i.e. ceil(log2 V) bits: P = ceil(bits / 8) passes over
// digits of db = ceil(bits / P) bits (V = 10 k: two passes of 7 bits; V = 2 M: three of 7), least significant digit first,
// every pass stable.  A pass is two launches over tiles of kSortThreads x E consecutive positions, one workgroup each:
//   radix_hist     count[side][digit][tile] = keys of the tile with that digit (every entry written: nothing relies on
//                  zeroed memory)
//   radix_scatter  position of a key = keys of smaller digits anywhere + keys of its digit in earlier tiles (both summed
//                  from the count table by the workgroup itself: thread d reads row d, 128 tiles per round trip) + keys of
//                  its digit earlier in its own tile (its stable local rank)
// Local ranks come from wave ballots: a wave takes 64 consecutive positions per round, the lanes holding the same digit
// find each other with db ballots, the group's first lane fetches-and-adds the group's size to the wave's running count of
// that digit (LDS atomic with return: rounds and peers are ordered by construction) and hands the old value round.
// The LAST pass writes the side's plan arrays itself (no gather launches): the sorted ids, the partner id / w / y of every
// pair pulled through the permutation (partner ids outside their table mapped to 0), and where the pair went — the row
// side its row-sorted position by arrival index (rpos), the col side the arrival index by col-sorted position (c_orig);
// side_tiles joins the two into c_perm / r_to_c.  The first pass maps ids outside their table to 0 on the fly — id 0 is
// what the reference's vocabulary lookup gives an unknown token (reference src/models/estimator.py:26-28) — so that no
// later kernel can index outside the tables whatever the caller hands over; counts[5] reports how many.
constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kMaxDigits = 256;
constexpr int kWalk = 16;                                 // 16-byte loads of a count-table row in flight per thread (= 128 tiles)

struct SortSide {
    const int32_t *keys;        // [n] keys of this pass, in the order the previous pass left them (first pass: the raw ids)
    const int32_t *vals;        // [n] arrival indices travelling with them; nullptr = the position itself (first pass)
    int32_t clean_below;        // first pass: anything outside [0, clean_below) counts as id 0 (0: keys are clean)
    int32_t *out_keys, *out_vals;   // not the last pass: the pair at its new position
    // last pass: the side's plan arrays
    int32_t *sorted_keys;       // [n] the sorted ids (the tile kernels below read them)
    const int32_t *other;       // the other side's raw ids by arrival index
    int32_t other_below;        // partner ids outside [0, other_below) count as id 0
    int32_t *partner;           // [n] partner id of the pair at each sorted position
    float *w_out, *y_out;       // [n]
    int32_t *where;             // row side: rpos[arrival index] = sorted position; col side: c_orig[sorted position] = arrival index
                                // (nullptr: the plan does not want c_perm / r_to_c)
};

struct SortPass {
    SortSide s[2];              // 0 row side, 1 col side (blockIdx.y)
    const float *w, *y;         // the batch's weights and values by arrival index
    int4 *pairs;                // [n] {row id, col id, w, y} by arrival index: written by the first radix_hist (row-side tiles),
                                // gathered by the last radix_scatter — one 16-byte fetch per pair and side instead of three
                                // 4-byte ones, each of which moves a whole sector (1 M pairs: 123 -> ... us for that launch)
    int64_t n;
    int shift, db;              // digit = (key >> shift) & ((1 << db) - 1)
    int ntiles;
    uint16_t *count;            // [2][kMaxDigits][tile_stride]: row d of a side = every tile's count of digit d (a tile holds < 65536 keys)
    int tile_stride;            // ntiles rounded up to a multiple of 8 (16-byte rows); the padding is never written or used
    int32_t *mapped;            // [2][ntiles]: ids this tile's workgroup mapped to 0 (first pass).  Every entry is written by
                                // every build — no counter to zero, no atomics; side_tiles adds them up into plan counts[5]
};

// *out = sum of v over the workgroup (kSortThreads threads; called by all of them)
__device__ inline void block_store_sum(int v, int32_t *out)
{
    __shared__ int part[kSortWaves];
    v = wave_sum_int(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int wv = 0; wv < kSortWaves; ++wv) s += part[wv];
        *out = s;
    }
}

template <int E>
__device__ inline void sort_load_keys(const SortSide &sd, int64_t n, int64_t base, int32_t (&key)[E], int &mapped)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = base + j * 64 + lane;
        int32_t k = 0;
        if (i < n) {
            k = sd.keys[i];
            if (sd.clean_below > 0 && (uint32_t)k >= (uint32_t)sd.clean_below) { k = 0; ++mapped; }
        }
        key[j] = k;
    }
}

// The passes of up to kPlanSetMax batches of the same size share their launches as well (blockIdx.z = batch): the
// indexes of consecutive batches of a stream in the launches of one (the set builders below; glove_plan_build_sorted).
struct SortPassSet { SortPass b[kPlanSetMax]; };
// (a single batch takes its arguments bare: a 2-KB argument block costs every launch of a lone build ~1.5 us)
__device__ inline const SortPass &pick(const SortPass &a) { return a; }
__device__ inline const SortPass &pick(const SortPassSet &a) { return a.b[blockIdx.z]; }

template <int E, class Args>
__global__ __launch_bounds__(kSortThreads) void radix_hist(Args args)
{
    const SortPass &in = pick(args);
    static_assert(kSortThreads == kMaxDigits, "one thread per digit zeroes and stores the histogram");
    __shared__ int hist[kMaxDigits];
    const int side = blockIdx.y;
    const SortSide &sd = in.s[side];
    const int nd = 1 << in.db, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = ((int64_t)blockIdx.x * kSortWaves + wave) * (64 * E);
    int32_t key[E];
    int mapped = 0;
    sort_load_keys<E>(sd, in.n, base, key, mapped);
    if (side == 0 && sd.vals == nullptr) {
        // first pass, row-side tiles: the batch as 16-byte pairs for the last pass's gathers (coalesced in and out)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t i = base + j * 64 + lane;
            if (i < in.n) in.pairs[i] = make_int4(key[j], sd.other[i], __float_as_int(in.w[i]), __float_as_int(in.y[i]));
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool valid = base + j * 64 + lane < in.n;
        const int digit = (key[j] >> in.shift) & (nd - 1);
        const unsigned long long peers = digit_peers(digit, in.db, valid);
        if (valid && lane == __ffsll((long long)peers) - 1) atomicAdd(&hist[digit], __popcll(peers));
    }
    __syncthreads();
    uint16_t *count = in.count + (size_t)side * kMaxDigits * in.tile_stride;
    if ((int)threadIdx.x < nd) count[(size_t)threadIdx.x * in.tile_stride + blockIdx.x] = (uint16_t)hist[threadIdx.x];
    // ids the cleaning mapped to 0 are reported once, by the pass that sees the raw ids
    if (sd.clean_below > 0) block_store_sum(mapped, in.mapped + (size_t)side * in.ntiles + blockIdx.x);
}

template <int E, bool LAST, class Args>
__global__ __launch_bounds__(kSortThreads) void radix_scatter(Args args)
{
    const SortPass &in = pick(args);
    __shared__ int wcnt[kSortWaves][kMaxDigits];          // a wave's running digit counts; then every wave's bases
    __shared__ int scan_red[kSortWaves];
    const int side = blockIdx.y;
    const SortSide &sd = in.s[side];
    const int nd = 1 << in.db, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // The kernel is a chain of short memory round trips on a few hundred workgroups: everything that does not depend on
    // the keys is requested up front, in the order it is needed (vector loads return in order), and consumed later.
    const int64_t base = ((int64_t)blockIdx.x * kSortWaves + wave) * (64 * E);
    int32_t key[E], rank[E], val[E];
    int mapped = 0;
    sort_load_keys<E>(sd, in.n, base, key, mapped);       // (ids mapped to 0 are counted by radix_hist)
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = base + j * 64 + lane;
        val[j] = sd.vals ? (i < in.n ? sd.vals[i] : 0) : (int32_t)i;
    }
    // row d of the side's count table (thread d; 16-bit counts, one row = the tiles' counts of digit d): the first kWalk x 8 tiles
    const int d_mine = (int)threadIdx.x < nd ? (int)threadIdx.x : 0;
    const uint4 *rowp = reinterpret_cast<const uint4 *>(in.count + ((size_t)side * kMaxDigits + d_mine) * in.tile_stride);
    const int nq = in.tile_stride / 8;                    // uint4 per row
    uint4 cw[kWalk];
#pragma unroll
    for (int x = 0; x < kWalk; ++x) cw[x] = rowp[x < nq ? x : 0];
    // what the last pass pulls through the permutation (depends on the values only)
    int32_t g_id[E];
    float g_w[E], g_y[E];
    if (LAST) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool valid = base + j * 64 + lane < in.n;
            const int32_t p = valid ? val[j] : 0;
            const int4 g = in.pairs[p];
            g_id[j] = side == 0 ? g.y : g.x;
            g_w[j] = __int_as_float(g.z);
            g_y[j] = __int_as_float(g.w);
        }
    }
    // ---- stable rank of every key among the keys of its digit in this wave's range (a wave zeroes and uses its own
    // counters: LDS operations of one wave complete in order, no barrier)
    for (int i = lane; i < kMaxDigits; i += 64) wcnt[wave][i] = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool valid = base + j * 64 + lane < in.n;
        const int digit = (key[j] >> in.shift) & (nd - 1);
        const unsigned long long peers = digit_peers(digit, in.db, valid);
        const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
        int before = 0;
        if (valid && lane == leader) before = atomicAdd(&wcnt[wave][digit], __popcll(peers));
        before = __shfl(before, leader, 64);
        rank[j] = before + __popcll(peers & ((1ull << lane) - 1ull));
    }
    // ---- thread d: keys of digit d in all tiles / in the tiles before this one
    int all = 0, earlier = 0;
    const int me = blockIdx.x;
    for (int q0 = 0; q0 < nq; q0 += kWalk) {
        if (q0 > 0) {
#pragma unroll
            for (int x = 0; x < kWalk; ++x) cw[x] = rowp[q0 + x < nq ? q0 + x : 0];
        }
#pragma unroll
        for (int x = 0; x < kWalk; ++x) {
            const uint32_t wd[4] = {cw[x].x, cw[x].y, cw[x].z, cw[x].w};
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int t = (q0 + x) * 8 + h;
                const int c = (int)((wd[h >> 1] >> (16 * (h & 1))) & 0xffffu);
                all += (q0 + x < nq && t < in.ntiles) ? c : 0;      // (the row's padding behind the last tile is never written)
                earlier += (q0 + x < nq && t < me) ? c : 0;
            }
        }
    }
    if ((int)threadIdx.x >= nd) all = earlier = 0;
    // exclusive scan of `all` over the digits (threads): wave scan, then the four wave totals
    int incl = all;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int o = __shfl_up(incl, dlt, 64);
        if (lane >= dlt) incl += o;
    }
    if (lane == 63) scan_red[wave] = incl;
    __syncthreads();                                      // also: every wave's running counts are final
    int start = incl - all + earlier;
    for (int wv = 0; wv < wave; ++wv) start += scan_red[wv];
    if ((int)threadIdx.x < nd) {
        // wcnt[w][d] becomes the position of the first key of digit d that wave w of this tile holds
        int run = start;
#pragma unroll
        for (int wv = 0; wv < kSortWaves; ++wv) {
            const int c = wcnt[wv][threadIdx.x];
            wcnt[wv][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
    // ---- every pair to its place
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = base + j * 64 + lane;
        if (i >= in.n) continue;
        const int digit = (key[j] >> in.shift) & (nd - 1);
        const int dest = wcnt[wave][digit] + rank[j];
        const int32_t p = val[j];
        if (!LAST) {
            sd.out_keys[dest] = key[j];
            sd.out_vals[dest] = p;
        } else {
            sd.sorted_keys[dest] = key[j];
            int32_t o = g_id[j];
            if ((uint32_t)o >= (uint32_t)sd.other_below) o = 0;     // (counted by the other side's first radix_hist)
            sd.partner[dest] = o;
            sd.w_out[dest] = g_w[j];
            sd.y_out[dest] = g_y[j];
            if (sd.where) { if (side == 0) sd.where[p] = dest; else sd.where[dest] = p; }
        }
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

// x[side] of a two-entry member of an argument struct, as a select: indexing by a run-time value would put a struct that was
// composed in registers (pick(SideDev)) into scratch memory (312 bytes per lane, the numbering kernels 20 -> 14 us per batch)
#define SIDE(x) (side ? x[1] : x[0])

constexpr int kTileThreads = 256;
constexpr int kTilePer = 8;                               // consecutive positions per thread
constexpr int kTile = kTileThreads * kTilePer;

struct SideKeys { const int32_t *keys[2]; };
struct SideOut {
    int32_t *chunk_id[2], *chunk_start[2], *uniq_slot[2];
    int32_t *counts;                                      // plan counts: [0],[1] row side, [2],[3] col side
    int32_t *uniq_rec[2];                                 // {id, first chunk, chunks, pairs} per distinct id
    const int64_t *tile_re;                               // [2][ntiles] end of the run crossing a tile's right edge (side_tiles)
    int32_t *heavy;                                       // ids with more than heavy_chunks chunks: (side << 30) | position, any order
    int heavy_chunks, cap_heavy;
    int2 *chunk_aux[2];                                   // per chunk (or nullptr): {position of its id among the side's ids, first chunk
                                                          // of its id << 31 | chunks of the id behind it}: words 2, 3 of its record header
    uint32_t *mark[2];                                    // per side (or nullptr): bitmap of the batch's ids — zeroed by side_tiles, set by side_emit
    int mark_words[2];
    uint32_t *chunk_hw[2];                                // per side (or nullptr): per chunk, word 3 of its record header on its own (glove_plan.r_chunk_hw)
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
    if (pos > 0) {
        // most runs of a batch are a few pairs long: the 64 positions in front of pos first, in the same round trip as the key —
        // one round for every run that starts among them instead of the three or four of the search over [0, pos]
        const int64_t q = pos - 1 - lane;
        const bool other = q >= 0 && keys[q] != key;       // (sorted: a different key in front of pos is a smaller one)
        const unsigned long long m = __ballot(other);
        if (m != 0) return pos - (__ffsll((long long)m) - 1);      // first lane whose position differs: the run starts right behind it
        if (pos < 64) return 0;                            // everything in front of pos holds the key
        hi = pos - 64;                                     // keys[pos - 64 .. pos] all hold the key
    }
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

// end of the run that position `pos` lies in: first index in (pos, B) whose key differs (is larger), or B.  One whole wave,
// the same 64-ary search forwards.
__device__ inline int64_t run_end_of(const int32_t *__restrict__ keys, int64_t B, int64_t pos)
{
    const int lane = threadIdx.x & 63;
    const int32_t key = keys[pos];
    int64_t lo = pos + 1, hi = B;                          // answer in [lo, hi]; everything below lo holds the key
    {
        // the 64 positions behind pos first (see run_start_of)
        const int64_t q = pos + 1 + lane;
        const bool other = q < B && keys[q] != key;
        const unsigned long long m = __ballot(other);
        if (m != 0) return pos + 1 + (__ffsll((long long)m) - 1);
        if (pos + 65 >= B) return B;                       // everything behind pos holds the key
        lo = pos + 65;
    }
    while (lo < hi) {
        const int64_t n = hi - lo, step = (n + 63) / 64;
        int64_t q = lo + step * lane;
        if (q > hi - 1) q = hi - 1;
        const bool gt = keys[q] > key;
        const unsigned long long m = __ballot(gt);
        if (m == 0) {                                      // every probe still holds the key: the answer lies behind the last one
            lo = __shfl(q, 63, 64) + 1;
            continue;
        }
        const int first = __ffsll((long long)m) - 1;       // first probe with a larger key
        const int64_t q_first = __shfl(q, first, 64);
        const int64_t q_before = __shfl(q, first > 0 ? first - 1 : 0, 64);
        lo = first == 0 ? lo : q_before + 1;
        hi = first == 0 ? lo : q_first;
    }
    return lo;
}

struct TileExtra {
    int64_t *tile_re;           // [2][ntiles] end of the run that crosses a tile's right edge
    int32_t *counts;            // plan counts
    const int32_t *mapped;      // per sort tile: ids mapped to 0 (row ids, then col ids)
    int n_mapped;
    // the join of the two sorts (col-side tiles): c_perm[q] = rpos[c_orig[q]], r_to_c its inverse
    const int32_t *c_orig;      // [B] arrival index of the pair at col-sorted position q
    const int32_t *rpos;        // [B] row-sorted position of the pair that arrived i-th
    int32_t *c_perm, *r_to_c;
};

// a batch that arrives sorted (glove_plan_build_sorted) and whose plan keeps pair arrays of its own: copied by side_tiles
struct SideCopy {
    const int32_t *p_src[2]; const float *w_src[2], *y_src[2];
    int32_t *p_dst[2]; float *w_dst[2], *y_dst[2];         // p_dst[side] == nullptr: nothing to copy
};

struct SideOne {                                           // the numbering kernels' arguments for one batch
    SideKeys sk;
    int64_t *tile_rs;
    int2 *tile_sums;
    TileExtra ex;
    SideOut out;
    SideCopy cp;
};
struct SideSet { SideOne b[kPlanSetMax]; };                // per batch of a set (blockIdx.z)
__device__ inline const SideOne &pick(const SideOne &a) { return a; }
__device__ inline const SideOne &pick(const SideSet &a) { return a.b[blockIdx.z]; }

// Any number of consecutive batches that arrive SORTED on both sides (an epoch dealt by glove_epoch_deal: batch z of a run
// occupies positions [z B, (z + 1) B) of the two orders): the plans are read from a device array, so one launch covers the
// whole run whatever its length (grid.z = batch) and its argument block stays small.
struct SortedSrc { const int32_t *id[2], *partner[2]; const float *w[2], *y[2]; };    // the first batch of the run, both orders
struct SortedWs { char *base; size_t per_batch, tile_rs, tile_re, tile_sums, aux0, aux1; };
struct SideDev {
    const glove_plan *plans;                               // device array, entry z = the plan of batch z of the run
    SortedSrc src;
    SortedWs ws;
    int64_t B;
    int ntiles;
    int32_t V;
};
__device__ inline SideOne pick(const SideDev &a)
{
    const glove_plan &pl = a.plans[blockIdx.z];
    const size_t at = (size_t)blockIdx.z * (size_t)a.B;
    char *w = a.ws.base + (size_t)blockIdx.z * a.ws.per_batch;
    SideOne o;
    o.sk = SideKeys{{a.src.id[0] + at, a.src.id[1] + at}};
    o.tile_rs = reinterpret_cast<int64_t *>(w + a.ws.tile_rs);
    o.tile_sums = reinterpret_cast<int2 *>(w + a.ws.tile_sums);
    int64_t *tile_re = reinterpret_cast<int64_t *>(w + a.ws.tile_re);
    int2 *aux0 = reinterpret_cast<int2 *>(w + a.ws.aux0), *aux1 = reinterpret_cast<int2 *>(w + a.ws.aux1);
    o.ex = TileExtra{tile_re, pl.counts, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    o.out = SideOut{{pl.r_chunk_id, pl.c_chunk_id}, {pl.r_chunk_start, pl.c_chunk_start}, {pl.r_uniq_slot, pl.c_uniq_slot},
                    pl.counts, {pl.r_uniq_rec, pl.c_uniq_rec}, tile_re, pl.heavy, pl.heavy_chunks, pl.cap_heavy,
                    {pl.r_crec ? aux0 : nullptr, pl.r_crec ? aux1 : nullptr},
                    {pl.r_mark, pl.c_mark}, {((pl.V_row > 0 ? pl.V_row : a.V) + 31) / 32, (a.V + 31) / 32},
                    {pl.r_chunk_hw, pl.c_chunk_hw}};
    o.cp = SideCopy{{a.src.partner[0] + at, a.src.partner[1] + at}, {a.src.w[0] + at, a.src.w[1] + at}, {a.src.y[0] + at, a.src.y[1] + at},
                    {pl.r_partner, pl.c_partner}, {pl.r_w, pl.c_w}, {pl.r_y, pl.c_y}};
    return o;
}

template <class Args>
__global__ __launch_bounds__(kTileThreads) void side_tiles(Args args, int64_t B, int32_t chunk_cap, int ntiles)
{
    const SideOne &one = pick(args);
    const SideKeys &sk = one.sk;
    int64_t *__restrict__ tile_rs = one.tile_rs;
    int2 *__restrict__ tile_sums = one.tile_sums;
    const TileExtra &ex = one.ex;
    __shared__ int64_t lds_rs[kTileThreads / 64 + 1];
    __shared__ int red[2][kTileThreads / 64];
    const int side = blockIdx.y, t = blockIdx.x;
    const int32_t *keys = SIDE(sk.keys);
    const int64_t begin = (int64_t)t * kTile;
    if (SIDE(one.out.mark)) {
        // this tile's share of the side's id bitmap starts at zero (side_emit, the next launch, sets the bits)
        uint32_t *mk = SIDE(one.out.mark);
        const int words = SIDE(one.out.mark_words);
        for (int i = t * kTileThreads + (int)threadIdx.x; i < words; i += ntiles * kTileThreads) mk[i] = 0u;
    }
    if (SIDE(one.cp.p_dst)) {
        // the batch arrived sorted and its plan keeps the pair fields itself (no chunk records): coalesced copy, a wave's
        // lanes on consecutive positions
        const SideCopy &cp = one.cp;
        for (int64_t k = begin + threadIdx.x; k < begin + kTile && k < B; k += kTileThreads) {
            SIDE(cp.p_dst)[k] = SIDE(cp.p_src)[k];
            SIDE(cp.w_dst)[k] = SIDE(cp.w_src)[k];
            SIDE(cp.y_dst)[k] = SIDE(cp.y_src)[k];
        }
    }
    if (side == 1 && ex.c_perm) {
        // the two sorts ran side by side: what links them is where the row side put each pair (two coalesced accesses, a
        // gather and a scatter per position, under the wave searches below)
        const int64_t q0 = begin + (int64_t)threadIdx.x * kTilePer;
        int32_t o[kTilePer], rp[kTilePer];
#pragma unroll
        for (int i = 0; i < kTilePer; ++i) o[i] = q0 + i < B ? ex.c_orig[q0 + i] : 0;
#pragma unroll
        for (int i = 0; i < kTilePer; ++i) rp[i] = ex.rpos[o[i]];
#pragma unroll
        for (int i = 0; i < kTilePer; ++i) {
            if (q0 + i >= B) continue;
            ex.c_perm[q0 + i] = rp[i];
            ex.r_to_c[rp[i]] = (int32_t)(q0 + i);
        }
    }
    if (threadIdx.x < 64) {                                // wave 0 searches together
        const int64_t r = begin > 0 ? run_start_of(keys, begin) : 0;
        if (threadIdx.x == 0) lds_rs[kTileThreads / 64] = r;
    } else if (threadIdx.x < 128) {                        // wave 1, meanwhile: where the tile's last run ends (side_emit's id records)
        const int64_t end = begin + kTile < B ? begin + kTile : B;
        const int64_t r = end < B ? run_end_of(keys, B, end - 1) : B;
        if (threadIdx.x == 64) ex.tile_re[(size_t)side * ntiles + t] = r;
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
    if (t == 0 && side == 0) {                             // (block-uniform)
        // The words of `counts` no later kernel of the build writes whole: the heavy-id counter side_emit (the next
        // launch) appends behind starts at zero, the spare words are zero, the ids mapped to 0 are the sum over the sort
        // tiles — no word of `counts` depends on what the buffer held before this build.
        int v = 0;
        for (int i = threadIdx.x; i < ex.n_mapped; i += kTileThreads) v += ex.mapped[i];
        v = wave_sum_int(v);
        __syncthreads();                                   // red[] of the tile sums has been read
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int wv = 0; wv < kTileThreads / 64; ++wv) total += red[0][wv];
            ex.counts[5] = total;
            ex.counts[4] = ex.counts[6] = ex.counts[7] = 0;
        }
    }
}

template <class Args>
__global__ __launch_bounds__(kTileThreads) void side_emit(Args args, int64_t B, int32_t chunk_cap, int ntiles)
{
    const SideOne &one = pick(args);
    const SideKeys &sk = one.sk;
    const int64_t *__restrict__ tile_rs = one.tile_rs;
    const int2 *__restrict__ tile_sums = one.tile_sums;
    const SideOut &out = one.out;
    __shared__ int64_t lds_rs[kTileThreads / 64 + 1];
    __shared__ int64_t lds_open[kTileThreads / 64];
    __shared__ int red[2][kTileThreads / 64];
    __shared__ int wave_tot[2][kTileThreads / 64];
    const int side = blockIdx.y, t = blockIdx.x;
    const int32_t *keys = SIDE(sk.keys);
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
    int32_t *chunk_id = SIDE(out.chunk_id), *chunk_start = SIDE(out.chunk_start), *uniq_slot = SIDE(out.uniq_slot);
    int open_ui[kTilePer], open_ci[kTilePer];              // number and first chunk of the id a position opens
    int64_t my_first_open = INT64_MAX;                     // first position of mine that opens an id
#pragma unroll
    for (int i = kTilePer - 1; i >= 0; --i)
        if (flag[i] & 2u) my_first_open = k0 + i;
#pragma unroll
    for (int i = 0; i < kTilePer; ++i) {
        const int64_t k = k0 + i;
        open_ui[i] = ui;
        open_ci[i] = ci;
        if (flag[i] & 2u) uniq_slot[ui++] = ci;
        if (flag[i] & 1u) { chunk_id[ci] = keys[k]; chunk_start[ci] = (int32_t)k; ++ci; }   // keys[k]: L1-hot, read by tile_flags
        if (k == B - 1) {                                  // closing entries and totals
            chunk_start[ci] = (int32_t)B;
            uniq_slot[ui] = ci;
            out.counts[2 * side] = ci;
            out.counts[2 * side + 1] = ui;
        }
    }
    // ---- {id, first chunk, chunks, pairs} of every id that opens in this tile.  An id's pairs end where the next id opens:
    // further on in this thread, in a thread to the right (suffix minimum of the threads' first openings), or beyond the
    // tile (tile_re: side_tiles searched the end of the run that crosses the right edge).  The chunks of an id restart
    // with it, so it has ceil(pairs / chunk_cap) of them.
    int64_t sfx = my_first_open;                           // min over lanes >= mine of this wave
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int64_t o = __shfl_down(sfx, dlt, 64);
        if (lane + dlt < 64) sfx = o < sfx ? o : sfx;
    }
    if (lane == 0) lds_open[wave] = sfx;
    __syncthreads();
    int64_t next_open = out.tile_re[(size_t)side * ntiles + t];
    for (int wv = kTileThreads / 64 - 1; wv > wave; --wv) next_open = lds_open[wv] < next_open ? lds_open[wv] : next_open;
    int64_t right = __shfl_down(sfx, 1, 64);               // min over the lanes to my right in this wave
    if (lane == 63) right = INT64_MAX;
    next_open = right < next_open ? right : next_open;
    int2 *aux = SIDE(out.chunk_aux);
#pragma unroll
    for (int i = kTilePer - 1; i >= 0; --i) {
        if (!(flag[i] & 1u)) continue;                     // (a position that opens an id opens a chunk)
        const int64_t k = k0 + i;
        const int pairs = (int)(next_open - k), chunks = (pairs + chunk_cap - 1) / chunk_cap;   // from k to the end of its id
        if (aux || SIDE(out.chunk_hw)) {
            // words 2 and 3 of the chunk's record header are at hand here (fill_records would bisect uniq_slot for them)
            const bool opens = (flag[i] & 2u) != 0;
            const uint32_t hw = (uint32_t)(chunks - 1) | (opens ? 0x80000000u : 0u);
            if (aux) aux[open_ci[i]] = make_int2(opens ? open_ui[i] : open_ui[i] - 1, (int)hw);
            if (SIDE(out.chunk_hw)) SIDE(out.chunk_hw)[open_ci[i]] = hw;
        }
        if (!(flag[i] & 2u)) continue;
        reinterpret_cast<int4 *>(SIDE(out.uniq_rec))[open_ui[i]] = make_int4(keys[k], open_ci[i], chunks, pairs);
        if (SIDE(out.mark)) atomicOr(SIDE(out.mark) + (keys[k] >> 5), 1u << (keys[k] & 31));
        if (chunks > out.heavy_chunks) {
            const int slot = atomicAdd(out.counts + 4, 1);  // zeroed by side_tiles, the launch before this one
            if (slot < out.cap_heavy) out.heavy[slot] = (side << 30) | open_ui[i];
        }
        next_open = k;
    }
}

// per-chunk records of both sides (blockIdx.y = side): one thread per (chunk, float4 of the record)
struct RecordArgs {
    const int32_t *uniq_slot[2];
    const int32_t *chunk_id[2], *chunk_start[2], *partner[2];
    const float *w[2], *y[2];
    int32_t *crec[2];
    const int2 *chunk_aux[2];     // side_emit's {id position, chunks-behind word} per chunk, or nullptr
    int header_in_record;         // no chunk_aux: 1 = the builder left words 2, 3 of every header in the record itself
                                  // (the one-workgroup builder), 0 = bisect uniq_slot (records of a finished plan)
};
struct RecordOne { RecordArgs a; const int32_t *counts; };
struct RecordSet { RecordOne b[kPlanSetMax]; };            // blockIdx.z: which plan of the set
__device__ inline const RecordOne &pick(const RecordOne &a) { return a; }
__device__ inline const RecordOne &pick(const RecordSet &a) { return a.b[blockIdx.z]; }
struct RecordDev { SideDev d; };                           // batches that arrived sorted: the pair fields are the epoch's
__device__ inline RecordOne pick(const RecordDev &r)
{
    const SideDev &a = r.d;
    const glove_plan &pl = a.plans[blockIdx.z];
    const size_t at = (size_t)blockIdx.z * (size_t)a.B;
    char *w = a.ws.base + (size_t)blockIdx.z * a.ws.per_batch;
    RecordOne o;
    o.a = RecordArgs{{pl.r_uniq_slot, pl.c_uniq_slot}, {pl.r_chunk_id, pl.c_chunk_id}, {pl.r_chunk_start, pl.c_chunk_start},
                     {a.src.partner[0] + at, a.src.partner[1] + at}, {a.src.w[0] + at, a.src.w[1] + at}, {a.src.y[0] + at, a.src.y[1] + at},
                     {pl.r_crec, pl.c_crec},
                     {reinterpret_cast<const int2 *>(w + a.ws.aux0), reinterpret_cast<const int2 *>(w + a.ws.aux1)}, 0};
    o.counts = pl.counts;
    return o;
}

typedef int v4i_t __attribute__((ext_vector_type(4)));

template <class Args>
__global__ __launch_bounds__(kBlock) void fill_records(Args args, int capP, int nt)
{
    const RecordOne &rone = pick(args);
    const RecordArgs &a = rone.a;
    const int32_t *__restrict__ counts = rone.counts;
    // A wave takes 32 consecutive chunks: their bounds, ids and header words arrive in three coalesced loads, then eight
    // lanes per chunk write line 0 of its record — header | block 0 | 16 B of padding: lane g of the octet stores float4 g,
    // the wave stores eight whole 128-byte lines per instruction — four chunks per lane, the loads of all four in flight
    // together.  (One chunk per 32-lane group, two dependent round trips each, ran 82 us for 2 x 373 k chunks: bound by the
    // turnover of 373 k waves.)  A chunk of n pairs is read up to its last block of 8: the blocks behind it are never
    // looked at (the pass kernels may copy them, they use slots below the next multiple of 8 only) and stay unwritten;
    // blocks 1 .. of the few longer chunks follow in a loop of the same eight lanes.
    const int side = blockIdx.y;
    const int n_chunks = counts[2 * side];
    const int32_t *chunk_id = SIDE(a.chunk_id), *chunk_start = SIDE(a.chunk_start), *partner = SIDE(a.partner);
    const float *w = SIDE(a.w), *y = SIDE(a.y);
    const int2 *aux = SIDE(a.chunk_aux);
    const int sq = rec_stride_q(capP);                     // float4 per record in memory
    const int lane = threadIdx.x & 63, g = lane & 7, oct = lane >> 3;
    const int j0 = (int)((blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 32);
    if (j0 >= n_chunks) return;
    // lanes 0 .. 32: chunk_start[j0 + lane]; lanes 0 .. 31: the chunk's id and header words 2, 3
    const int jl = j0 + lane < n_chunks ? j0 + lane : n_chunks;       // chunk_start[n_chunks] = B closes the last chunk
    const int cs = lane <= 32 ? chunk_start[jl] : 0;
    // The pair fields of the wave's 32 chunks are one contiguous run of positions (a few hundred for the short chunks of a big
    // batch): fetched coalesced into LDS once, the records are put together from there.  (Every lane pulling its four values of
    // a field out of global memory by itself was 16 scattered 4-byte load instructions per lane: 39 us per 1 M-pair batch of
    // V = 400 k against 33 with the staging.)  A run too long for the stage — the full chunks
    // of the Zipf head — is read directly as before.
    constexpr int kStage = 384;
    __shared__ int32_t st_p[kBlock / 64][kStage];
    __shared__ float st_w[kBlock / 64][kStage], st_y[kBlock / 64][kStage];
    const int wv = threadIdx.x >> 6;
    const int p0 = __shfl(cs, 0, 64);
    const int last = n_chunks - j0 < 32 ? n_chunks - j0 : 32;
    const int span = __shfl(cs, last, 64) - p0;
    const bool staged = span <= kStage;
    if (staged) {
        for (int i = lane; i < span; i += 64) {
            st_p[wv][i] = partner[p0 + i];
            st_w[wv][i] = w[p0 + i];
            st_y[wv][i] = y[p0 + i];
        }
    }
    int32_t cid = 0;
    int2 ax = make_int2(0, 0);
    if (lane < 32 && j0 + lane < n_chunks) {
        cid = chunk_id[j0 + lane];
        if (aux) {
            ax = aux[j0 + lane];
        } else if (a.header_in_record) {
            ax = *reinterpret_cast<const int2 *>(SIDE(a.crec) + (size_t)(j0 + lane) * sq * 4 + 2);
        } else {
            // word 3: (first chunk of its id) << 31 | chunks of the same id behind this one.  The id's chunks are
            // [uniq_slot[q], uniq_slot[q + 1]) for the q found by bisection (the slots are ascending)
            const int j = j0 + lane;
            const int32_t *slot = SIDE(a.uniq_slot);
            int lo = 0, hi = counts[2 * side + 1];             // slot[lo] <= j < slot[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (slot[mid] <= j) lo = mid; else hi = mid;
            }
            // word 2: the id's position among the side's distinct ids (ascending id order)
            ax = make_int2(lo, (int)((uint32_t)(slot[lo + 1] - 1 - j) | (slot[lo] == j ? 0x80000000u : 0u)));
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 8 + oct, j = j0 + c;
        const int s = __shfl(cs, c, 64), n = __shfl(cs, c + 1, 64) - s;
        const int32_t id = __shfl(cid, c, 64);
        const int x2 = __shfl(ax.x, c, 64), x3 = __shfl(ax.y, c, 64);
        if (j >= n_chunks) continue;
        int4 *dst = reinterpret_cast<int4 *>(SIDE(a.crec)) + (size_t)j * sq;
        const int nq = 1 + 6 * ((n + kRecPad - 1) / kRecPad);   // logical float4 the chunk needs
        // gq: float4 of the record in memory; 7 is line 0's padding; logical f = gq below 7, gq - 1 above
        for (int gq = g; gq < (nq > 7 ? nq + 1 : 8); gq += 8) {
            const int f = gq < 7 ? gq : gq - 1;
            int4 v;
            if (gq == 7) {
                v = make_int4(0, 0, 0, 0);
            } else if (f == 0) {
                v = make_int4(id, n, x2, x3);
            } else {
                // blocks of kRecPad = 8 pairs, each {partner[8] | w[8] | y[8]}: float4 r of block b holds field r / 2, pairs 8 b + 4 (r % 2) ..
                const int b = (f - 1) / 6, r = (f - 1) % 6;
                const int field = r / 2, t0 = b * kRecPad + (r % 2) * 4;
                int o[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int t = t0 + x;
                    const int k = s + (t < n ? t : 0);              // padding replays pair 0 with weight 0
                    if (staged)                                     // (same wave wrote the stage: LDS ops of one wave complete in order)
                        o[x] = field == 0 ? st_p[wv][k - p0] : field == 1 ? __float_as_int(t < n ? st_w[wv][k - p0] : 0.f) : __float_as_int(st_y[wv][k - p0]);
                    else
                        o[x] = field == 0 ? partner[k] : field == 1 ? __float_as_int(t < n ? w[k] : 0.f) : __float_as_int(y[k]);
                }
                v = make_int4(o[0], o[1], o[2], o[3]);
            }
            // (nt: records of a big batch are read a segment of steps later, long after the L2s have turned over: they go
            // out non-temporal instead of pushing table rows out)
            if (nt) __builtin_nontemporal_store(v4i_t{v.x, v.y, v.z, v.w}, reinterpret_cast<v4i_t *>(dst + gq));
            else dst[gq] = v;
        }
    }
}

static RecordArgs record_args(const glove_plan *plan, const int2 *aux_r, const int2 *aux_c, bool header_in_record)
{
    return RecordArgs{{plan->r_uniq_slot, plan->c_uniq_slot}, {plan->r_chunk_id, plan->c_chunk_id}, {plan->r_chunk_start, plan->c_chunk_start},
                      {plan->r_partner, plan->c_partner}, {plan->r_w, plan->c_w}, {plan->r_y, plan->c_y},
                      {plan->r_crec, plan->c_crec}, {aux_r, aux_c}, header_in_record ? 1 : 0};
}

// records of n plans of the same shape (same B, chunk_cap, capacities) in one launch
static int launch_fill_records_set(const glove_plan *const *plans, int n, hipStream_t st, const int2 *const *aux_r = nullptr,
                                   const int2 *const *aux_c = nullptr, bool header_in_record = false)
{
    const glove_plan *plan = plans[0];
    const int capP = rec_cap(plan->chunk_cap);        // glove_common.h: a trip of the pass kernel reads up to kRecPad slots from q0
    RecordSet set = {};
    int64_t most = 1;
    for (int j = 0; j < n; ++j) {
        set.b[j].a = record_args(plans[j], aux_r ? aux_r[j] : nullptr, aux_c ? aux_c[j] : nullptr, header_in_record);
        set.b[j].counts = plans[j]->counts;
        const int64_t nr = most_chunks(plans[j], true), nc = most_chunks(plans[j], false);
        most = nr > most ? nr : most;
        most = nc > most ? nc : most;
    }
    // 32 chunks per wave, four waves per workgroup
    const int64_t per_block = (kBlock / 64) * 32;
    const int64_t nb = (most + per_block - 1) / per_block;
    const dim3 grid((unsigned)(nb < 1 ? 1 : nb), 2, n);
    if (n == 1) hipLaunchKernelGGL(fill_records<RecordOne>, grid, dim3(kBlock), 0, st, set.b[0], capP, 0);
    else hipLaunchKernelGGL(fill_records<RecordSet>, grid, dim3(kBlock), 0, st, set, capP, 0);
    return (int)hipGetLastError();
}

static int launch_fill_records(const glove_plan *plan, hipStream_t st, bool header_in_record = false)
{
    return launch_fill_records_set(&plan, 1, st, nullptr, nullptr, header_in_record);
}

struct PlanWs {
    int32_t *keys[2][2], *vals[2][2];   // [side][ping-pong] buffers of the sort passes
    int32_t *row_sorted;             // row ids in row-side order
    int32_t *col_sorted;             // col ids in col-side order
    int32_t *rpos, *c_orig;          // [B] row-sorted position by arrival index / arrival index by col-sorted position
    int4 *pairs;                     // [B] the batch as {row, col, w, y} by arrival index
    uint16_t *count;                 // [2][digits][tile_stride] of the pass in flight
    int tile_stride;
    int32_t *mapped;                 // [2][sort tiles] ids mapped to 0 per tile (row ids, col ids)
    int64_t *tile_rs;                // [2][ntiles] start of the run that crosses a tile's left edge
    int64_t *tile_re;                // [2][ntiles] end of the run that crosses a tile's right edge
    int2 *tile_sums;                 // [2][ntiles] (ids, chunks) opened inside a tile
    int2 *chunk_aux[2];              // [B] per side: record header words 2, 3 of every chunk (plans with chunk records)
    int ntiles;                      // tiles of the numbering kernels (kTile positions)
    int sort_e, sort_tiles;          // positions per thread and tiles of the sort passes
    size_t bytes;
};

// positions per thread of a sort tile: enough tiles to spread over the chip, few enough that walking a column of the
// count table (one entry per tile) stays short
static int sort_e_for(int64_t B) { return B <= 131072 ? 4 : B <= 524288 ? 8 : 16; }

static PlanWs carve_plan_ws(void *ws, int64_t B)
{
    PlanWs p;
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { void *q = base + off; off += align_up(bytes, 256); return q; };
    const size_t n = (size_t)(B > 0 ? B : 1);
    for (int sd = 0; sd < 2; ++sd)
        for (int i = 0; i < 2; ++i) {
            p.keys[sd][i] = (int32_t *)take(n * 4);
            p.vals[sd][i] = (int32_t *)take(n * 4);
        }
    p.row_sorted = (int32_t *)take(n * 4);
    p.col_sorted = (int32_t *)take(n * 4);
    p.rpos = (int32_t *)take(n * 4);
    p.c_orig = (int32_t *)take(n * 4);
    p.pairs = (int4 *)take(n * 16);
    p.sort_e = sort_e_for(B);
    const size_t per_tile = (size_t)kSortThreads * p.sort_e;
    p.sort_tiles = (int)((n + per_tile - 1) / per_tile);
    p.tile_stride = (p.sort_tiles + 7) / 8 * 8;
    p.count = (uint16_t *)take((size_t)2 * p.tile_stride * kMaxDigits * 2);
    p.mapped = (int32_t *)take((size_t)2 * p.sort_tiles * 4);
    p.ntiles = (int)((n + kTile - 1) / kTile);
    p.tile_rs = (int64_t *)take((size_t)2 * p.ntiles * 8);
    p.tile_re = (int64_t *)take((size_t)2 * p.ntiles * 8);
    p.tile_sums = (int2 *)take((size_t)2 * p.ntiles * 8);
    for (int i = 0; i < 2; ++i) p.chunk_aux[i] = (int2 *)take(n * 8);
    p.bytes = off;
    return p;
}

static inline int32_t Vr_of(const glove_plan *plan, int32_t V) { return plan->V_row > 0 ? plan->V_row : V; }

static int ceil_log2(int32_t v)
{
    int b = 1;
    while (b < 31 && (1 << b) < v) ++b;
    return b;
}

// Both stable sorts by id of the nb batches of a set, P passes of (radix_hist, radix_scatter) with grid.y = side and
// grid.z = batch.  fin[j] carries, per side, the raw ids (first pass) and the destination arrays of the last pass; pw[j]
// is batch j's workspace.
template <int E>
static void launch_sorts(const SortPass *fin, const PlanWs *pw, int nb, int64_t B, int bits, hipStream_t st)
{
    const int P = (bits + 7) / 8, db = (bits + P - 1) / P;
    for (int p = 0; p < P; ++p) {
        SortPassSet set;
        for (int j = 0; j < nb; ++j) {
            SortPass &in = set.b[j];
            in = fin[j];
            for (int sd = 0; sd < 2; ++sd) {
                if (p > 0) {
                    in.s[sd].keys = pw[j].keys[sd][(p - 1) & 1];
                    in.s[sd].vals = pw[j].vals[sd][(p - 1) & 1];
                    in.s[sd].clean_below = 0;
                }
                in.s[sd].out_keys = pw[j].keys[sd][p & 1];
                in.s[sd].out_vals = pw[j].vals[sd][p & 1];
            }
            in.n = B;
            in.shift = p * db;
            in.db = db;
            in.ntiles = pw[j].sort_tiles;
            in.count = pw[j].count;
            in.tile_stride = pw[j].tile_stride;
            in.mapped = pw[j].mapped;
        }
        const dim3 grid(pw[0].sort_tiles, 2, nb);
        if (nb == 1) {
            hipLaunchKernelGGL((radix_hist<E, SortPass>), grid, dim3(kSortThreads), 0, st, set.b[0]);
            if (p < P - 1) hipLaunchKernelGGL((radix_scatter<E, false, SortPass>), grid, dim3(kSortThreads), 0, st, set.b[0]);
            else hipLaunchKernelGGL((radix_scatter<E, true, SortPass>), grid, dim3(kSortThreads), 0, st, set.b[0]);
        } else {
            hipLaunchKernelGGL((radix_hist<E, SortPassSet>), grid, dim3(kSortThreads), 0, st, set);
            if (p < P - 1) hipLaunchKernelGGL((radix_scatter<E, false, SortPassSet>), grid, dim3(kSortThreads), 0, st, set);
            else hipLaunchKernelGGL((radix_scatter<E, true, SortPassSet>), grid, dim3(kSortThreads), 0, st, set);
        }
    }
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// The tiled builder over nb <= kPlanSetMax batches of B pairs each (batch j: pairs [j B, (j + 1) B) of the arrays, plan
// plans[j], workspace slice j): every launch covers all of them.
static int build_tiled_set(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                           const glove_plan *const *plans, int nb, void *ws, size_t ws_bytes, hipStream_t st)
{
    const size_t per = carve_plan_ws(nullptr, B).bytes;
    if (per * (size_t)nb > ws_bytes) return GLOVE_E_WORKSPACE;
    PlanWs pw[kPlanSetMax];
    SortPass fin[kPlanSetMax];
    const int bits = ceil_log2(V);
    for (int j = 0; j < nb; ++j) {
        const glove_plan *plan = plans[j];
        pw[j] = carve_plan_ws((char *)ws + per * j, B);
        const int32_t Vr = plan->V_row > 0 ? plan->V_row : V;
        const size_t off = (size_t)j * B;
        // ---- both sides: stable sort of the batch by (id, arrival index); the last pass fills the side's pair arrays
        SortPass &f = fin[j];
        f = SortPass{};
        f.w = w + off; f.y = y + off; f.pairs = pw[j].pairs;
        SortSide &rs = f.s[0], &cs = f.s[1];
        rs.keys = row + off; rs.vals = nullptr; rs.clean_below = Vr;
        rs.sorted_keys = pw[j].row_sorted; rs.other = col + off; rs.other_below = V;
        rs.partner = plan->r_partner; rs.w_out = plan->r_w; rs.y_out = plan->r_y; rs.where = plan->c_perm ? pw[j].rpos : nullptr;
        cs.keys = col + off; cs.vals = nullptr; cs.clean_below = V;
        cs.sorted_keys = pw[j].col_sorted; cs.other = row + off; cs.other_below = Vr;
        cs.partner = plan->c_partner; cs.w_out = plan->c_w; cs.y_out = plan->c_y; cs.where = plan->c_perm ? pw[j].c_orig : nullptr;
    }
    if (pw[0].sort_e == 4) launch_sorts<4>(fin, pw, nb, B, bits, st);
    else if (pw[0].sort_e == 8) launch_sorts<8>(fin, pw, nb, B, bits, st);
    else launch_sorts<16>(fin, pw, nb, B, bits, st);

    // ---- chunks and ids of both sides: two launches over tiles of the sorted keys, then the id records
    SideSet ss;
    const int2 *aux_r[kPlanSetMax], *aux_c[kPlanSetMax];
    for (int j = 0; j < nb; ++j) {
        const glove_plan *plan = plans[j];
        ss.b[j].sk = SideKeys{{pw[j].row_sorted, pw[j].col_sorted}};
        ss.b[j].tile_rs = pw[j].tile_rs;
        ss.b[j].tile_sums = pw[j].tile_sums;
        ss.b[j].out = SideOut{{plan->r_chunk_id, plan->c_chunk_id}, {plan->r_chunk_start, plan->c_chunk_start},
                            {plan->r_uniq_slot, plan->c_uniq_slot}, plan->counts, {plan->r_uniq_rec, plan->c_uniq_rec},
                            (const int64_t *)pw[j].tile_re, plan->heavy, plan->heavy_chunks, plan->cap_heavy,
                            {plan->r_crec ? pw[j].chunk_aux[0] : nullptr, plan->r_crec ? pw[j].chunk_aux[1] : nullptr},
                            {plan->r_mark, plan->c_mark}, {(Vr_of(plan, V) + 31) / 32, (V + 31) / 32},
                            {plan->r_chunk_hw, plan->c_chunk_hw}};
        ss.b[j].cp = SideCopy{};
        ss.b[j].ex = TileExtra{pw[j].tile_re, plan->counts, (const int32_t *)pw[j].mapped, 2 * pw[j].sort_tiles,
                             (const int32_t *)pw[j].c_orig, (const int32_t *)pw[j].rpos, plan->c_perm, plan->r_to_c};
        aux_r[j] = pw[j].chunk_aux[0];
        aux_c[j] = pw[j].chunk_aux[1];
    }
    const dim3 grid(pw[0].ntiles, 2, nb);
    if (nb == 1) {
        hipLaunchKernelGGL(side_tiles<SideOne>, grid, dim3(kTileThreads), 0, st, ss.b[0], B, plans[0]->chunk_cap, pw[0].ntiles);
        hipLaunchKernelGGL(side_emit<SideOne>, grid, dim3(kTileThreads), 0, st, ss.b[0], B, plans[0]->chunk_cap, pw[0].ntiles);
    } else {
        hipLaunchKernelGGL(side_tiles<SideSet>, grid, dim3(kTileThreads), 0, st, ss, B, plans[0]->chunk_cap, pw[0].ntiles);
        hipLaunchKernelGGL(side_emit<SideSet>, grid, dim3(kTileThreads), 0, st, ss, B, plans[0]->chunk_cap, pw[0].ntiles);
    }
    if (plans[0]->r_crec) return launch_fill_records_set(plans, nb, st, aux_r, aux_c);
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

// the argument checks of an index build into `plan` (B > 0)
static int check_plan_for_build(const glove_plan *plan, int64_t B, int32_t V)
{
    if (!plan->r_partner || !plan->r_w || !plan->r_y || !plan->r_chunk_id || !plan->r_chunk_start || !plan->r_uniq_slot ||
        !plan->c_w || !plan->c_y || !plan->r_uniq_rec || !plan->c_uniq_rec || !plan->c_partner || !plan->c_chunk_id || !plan->c_chunk_start || !plan->c_uniq_slot)
        return GLOVE_E_BADARG;
    if (!plan->heavy || plan->heavy_chunks < 1 ||
        plan->cap_heavy < 2 * B / ((int64_t)plan->heavy_chunks * plan->chunk_cap) + 2)
        return GLOVE_E_WORKSPACE;
    // the chunk / uniq arrays must be able to hold the worst case (every pair its own chunk)
    if (plan->cap_chunks < B || plan->cap_uniq < (B < V ? B : V)) return GLOVE_E_WORKSPACE;
    if ((plan->r_crec == nullptr) != (plan->c_crec == nullptr)) return GLOVE_E_BADARG;
    if ((plan->r_to_c == nullptr) != (plan->c_perm == nullptr)) return GLOVE_E_BADARG;     // the links come as a pair or not at all
    if ((plan->r_mark == nullptr) != (plan->c_mark == nullptr)) return GLOVE_E_BADARG;     // so do the id bitmaps
    if ((plan->r_chunk_hw == nullptr) != (plan->c_chunk_hw == nullptr)) return GLOVE_E_BADARG; // and the run words
    return 0;
}

int glove_plan_build(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, void *ws, size_t ws_bytes, void *stream)
{
    if (!plan || !ws || B < 0 || V <= 0 || plan->chunk_cap <= 0 || plan->B != B || !plan->counts) return GLOVE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        HIP_TRY(zero_words(plan->counts, 8, st));
        return 0;
    }
    if (!row || !col || !w || !y) return GLOVE_E_BADARG;
    if (int rc = check_plan_for_build(plan, B, V)) return rc;
    if (B <= kSmallPlanMax) {                                                          // launch-bound regime
        PlanSet set;
        set.p[0] = *plan;
        if (int rc = plan_build_small(row, col, w, y, B, V, set, 1, st)) return rc;
        return plan->r_crec ? launch_fill_records(plan, st, true) : 0;
    }
    return build_tiled_set(row, col, w, y, B, V, &plan, 1, ws, ws_bytes, st);
}

// ---- the index of batches that arrive sorted on both sides (an epoch dealt by glove_epoch_deal) ---------------------------
static SortedWs carve_sorted_ws(void *ws, int64_t B, int *ntiles_out)
{
    SortedWs s;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t q = off; off += align_up(bytes, 256); return q; };
    const size_t n = (size_t)(B > 0 ? B : 1);
    const int ntiles = (int)((n + kTile - 1) / kTile);
    s.base = (char *)ws;
    s.tile_rs = take((size_t)2 * ntiles * 8);
    s.tile_re = take((size_t)2 * ntiles * 8);
    s.tile_sums = take((size_t)2 * ntiles * 8);
    s.aux0 = take(n * 8);
    s.aux1 = take(n * 8);
    s.per_batch = off;
    if (ntiles_out) *ntiles_out = ntiles;
    return s;
}

size_t glove_plan_sorted_workspace_bytes(int64_t B, int32_t n_batches)
{
    if (B < 0 || n_batches < 0) return 0;
    return carve_sorted_ws(nullptr, B, nullptr).per_batch * (size_t)n_batches;
}

int32_t glove_plan_chunk_bound(int64_t B, int32_t cap_uniq, int32_t chunk_cap)
{
    if (B < 0 || cap_uniq < 0 || chunk_cap <= 0) return 0;
    const int64_t bound = (int64_t)cap_uniq + B / chunk_cap + 1;
    return (int32_t)(bound < B ? bound : B);
}

int glove_plan_build_sorted(const glove_pairs *row_side, const glove_pairs *col_side, int64_t first_pair, int64_t B,
                            int32_t n_batches, int32_t V, const glove_plan *plans, const glove_plan *plans_dev, void *ws,
                            size_t ws_bytes, void *stream)
{
    if (!row_side || !col_side || !plans || !plans_dev || first_pair < 0 || B <= 0 || n_batches < 0 || V <= 0) return GLOVE_E_BADARG;
    if (n_batches == 0) return 0;
    if (!row_side->id || !row_side->partner || !row_side->w || !row_side->y || !col_side->id || !col_side->partner || !col_side->w ||
        !col_side->y || !ws)
        return GLOVE_E_BADARG;
    int64_t most = 1;
    for (int j = 0; j < n_batches; ++j) {
        const glove_plan *p = plans + j;
        if (p->B != B || p->chunk_cap <= 0 || p->chunk_cap != plans[0].chunk_cap || !p->counts) return GLOVE_E_BADARG;
        if ((p->r_crec == nullptr) != (plans[0].r_crec == nullptr) || (p->r_crec == nullptr) != (p->c_crec == nullptr)) return GLOVE_E_BADARG;
        if (!p->r_chunk_id || !p->r_chunk_start || !p->r_uniq_slot || !p->r_uniq_rec || !p->c_chunk_id || !p->c_chunk_start ||
            !p->c_uniq_slot || !p->c_uniq_rec)
            return GLOVE_E_BADARG;
        // a plan without chunk records keeps the pair fields itself (the step kernels read them): all six or none
        const bool own = p->r_partner || p->r_w || p->r_y || p->c_partner || p->c_w || p->c_y;
        if (own && (!p->r_partner || !p->r_w || !p->r_y || !p->c_partner || !p->c_w || !p->c_y)) return GLOVE_E_BADARG;
        // neither records nor arrays: a plan with run words that BORROWS its pair fields — the batch lies sorted in the epoch's
        // arrays already; the caller points r_partner .. c_y of the struct it steps with at them (nothing is copied)
        if (!own && !p->r_crec && !p->r_chunk_hw) return GLOVE_E_BADARG;
        if (p->c_perm || p->r_to_c) return GLOVE_E_BADARG;              // the links between the orders are not computed here
        if ((p->r_mark == nullptr) != (p->c_mark == nullptr)) return GLOVE_E_BADARG;
        if ((p->r_chunk_hw == nullptr) != (p->c_chunk_hw == nullptr)) return GLOVE_E_BADARG;
        if (!p->heavy || p->heavy_chunks < 1 || p->cap_heavy < 2 * B / ((int64_t)p->heavy_chunks * p->chunk_cap) + 2) return GLOVE_E_WORKSPACE;
        if (p->cap_uniq < (B < V ? B : V)) return GLOVE_E_WORKSPACE;
        if (p->cap_chunks < glove_plan_chunk_bound(B, p->cap_uniq, p->chunk_cap)) return GLOVE_E_WORKSPACE;
        const int64_t mr = most_chunks(p, true), mc = most_chunks(p, false);
        most = mr > most ? mr : most;
        most = mc > most ? mc : most;
    }
    SideDev sd;
    sd.plans = plans_dev;
    sd.src = SortedSrc{{row_side->id + first_pair, col_side->id + first_pair}, {row_side->partner + first_pair, col_side->partner + first_pair},
                       {row_side->w + first_pair, col_side->w + first_pair}, {row_side->y + first_pair, col_side->y + first_pair}};
    sd.ws = carve_sorted_ws(ws, B, &sd.ntiles);
    sd.B = B;
    sd.V = V;
    if (sd.ws.per_batch * (size_t)n_batches > ws_bytes) return GLOVE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int32_t cap = plans[0].chunk_cap;
    for (int z0 = 0; z0 < n_batches; z0 += 65535) {                      // (grid.z is 16 bits)
        const int nz = n_batches - z0 < 65535 ? n_batches - z0 : 65535;
        SideDev part = sd;
        part.plans = plans_dev + z0;
        part.ws.base = sd.ws.base + (size_t)z0 * sd.ws.per_batch;
        for (int x = 0; x < 2; ++x) {
            part.src.id[x] += (size_t)z0 * B; part.src.partner[x] += (size_t)z0 * B;
            part.src.w[x] += (size_t)z0 * B; part.src.y[x] += (size_t)z0 * B;
        }
        const dim3 grid(sd.ntiles, 2, nz);
        hipLaunchKernelGGL(side_tiles<SideDev>, grid, dim3(kTileThreads), 0, st, part, B, cap, sd.ntiles);
        hipLaunchKernelGGL(side_emit<SideDev>, grid, dim3(kTileThreads), 0, st, part, B, cap, sd.ntiles);
        if (plans[0].r_crec) {
            const int64_t per_block = (kBlock / 64) * 32;
            const int64_t nb = (most + per_block - 1) / per_block;
            hipLaunchKernelGGL(fill_records<RecordDev>, dim3((unsigned)(nb < 1 ? 1 : nb), 2, nz), dim3(kBlock), 0, st, RecordDev{part}, rec_cap(cap), B >= 262144 ? 1 : 0);
        }
    }
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
