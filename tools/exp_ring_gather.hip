// Timing experiment (not product code): what does a lane group sustain when every row it needs — own row, partner rows,
// accumulator row — arrives through an LDS ring filled by global_load_lds_dwordx4, D rows in flight per group at all
// times, instead of U = 4 partner rows per dependent trip into VGPRs (sidepass_kernel<32, 3, *, false, 1>)?
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp_ring_gather.hip -o gpurun_out/exp_ring_gather && gpurun_out/exp_ring_gather
//
// One side of a config-4 step: V = 400 k rows of d = 300 floats, B = 1 M pairs sorted by own id (Zipf(1.0) ids on both
// sides), chunks of <= 32 pairs, a lane group (32 lanes x 3 float4) owns `per` consecutive chunks.  Per run of an id inside
// a group the entry list is  OWN(u), PAIR(p) ..., ACC(u);  OWN and ACC read 1,200 B each, ACC writes two rows (the new
// table row into a second copy, the accumulator in place).  Arithmetic per pair: dot (butterfly over the group) and an axpy.
// A small case is checked against the host first (the DMA image, the counted vmcnt waits).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef D4
#define D4 75
#endif
constexpr int kD4 = D4, kLPR = 32, kNV = (D4 + 31) / 32, kBlock = 256;
constexpr uint32_t kOwn = 0u, kPair = 1u, kAcc = 2u;
constexpr int kMaxEntries = 192;          // per group: 12 chunks x 1..32 pairs would be 384 + 24; the generator caps a group's pairs at 128

__device__ inline float grp_sum32(float v)
{
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
    return v;
}

struct Args {
    const uint32_t *entries;      // flat, per group [off[g], off[g+1])
    const int *off;
    int ngroups;
    const float *own, *partner, *acc_in;
    float *own_out, *acc_out;
    const float *bias;
    int mode;            // bit 0: no stores; bit 1: own / accumulator rows folded into the first 1,024 rows (cache hits); bit 2: stores folded likewise
};

// ---- the ring: all rows through LDS, D in flight per group ----------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void ring_kernel(Args a)
{
    constexpr int kWaves = kBlock / 64;
    __shared__ __attribute__((aligned(16))) f4 ring[kWaves * D * kNV * 64];
    __shared__ uint32_t ent[(kBlock / kLPR) * kMaxEntries];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = threadIdx.x % kLPR, grp = threadIdx.x / kLPR;
    const int gi = blockIdx.x * (kBlock / kLPR) + grp;
    int T = 0, off = 0;
    if (gi < a.ngroups) { off = a.off[gi]; T = a.off[gi + 1] - off; }
    uint32_t *my = ent + grp * kMaxEntries;
    for (int i = lg; i < T; i += kLPR) my[i] = a.entries[off + i];
    // the wave's step count: the longer of its two groups' lists
    const int Tw = __builtin_amdgcn_readfirstlane(max(__shfl(T, 0, 64), __shfl(T, 32, 64)));
    f4 *wring = ring + wave * D * kNV * 64;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)wring;         // LDS byte address of the wave's ring (shared pointers are 32-bit offsets)
    (void)ring_lds;

    auto issue = [&](int t) {
        if (t < T) {
            const uint32_t e = my[t];
            const uint32_t type = e >> 30;
            uint32_t row = e & 0x3fffffffu;
            if ((a.mode & 2) && type != kPair) row &= 1023u;
            const float *base = type == kOwn ? a.own : type == kPair ? a.partner : a.acc_in;
            const f4 *src = reinterpret_cast<const f4 *>(base) + (size_t)row * kD4;
            f4 *dst = wring + (t % D) * kNV * 64;
#pragma unroll
            for (int k = 0; k < kNV; ++k) {
                const int i4 = lg + k * kLPR;
                if (i4 < kD4)
                    __builtin_amdgcn_global_load_lds(src + i4, (__attribute__((address_space(3))) void *)(dst + k * 64), 16, 0, 0);
            }
        }
    };

    f4 r[kNV], acc[kNV];
#pragma unroll
    for (int k = 0; k < kNV; ++k) { r[k] = f4{0, 0, 0, 0}; acc[k] = f4{0, 0, 0, 0}; }

    auto consume = [&](int t) {
        if (t < T) {
            const uint32_t e = my[t];
            const uint32_t type = e >> 30;
            uint32_t row = e & 0x3fffffffu;
            if (a.mode & 4) row &= 1023u;
            f4 c[kNV];
            const uint32_t addr = (uint32_t)(uintptr_t)(wring + (t % D) * kNV * 64) + lane * 16;
            if constexpr (kNV == 3)
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]) : "v"(addr) : "memory");
            else if constexpr (kNV == 2)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c[0]), "=&v"(c[1]) : "v"(addr) : "memory");
            else
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c[0]) : "v"(addr) : "memory");
            if (lg + (kNV - 1) * kLPR >= kD4) c[kNV - 1] = f4{0, 0, 0, 0};
            if (type == kOwn) {
#pragma unroll
                for (int k = 0; k < kNV; ++k) { r[k] = c[k]; acc[k] = f4{0, 0, 0, 0}; }
            } else if (type == kPair) {
                float dp = 0.f;
#pragma unroll
                for (int k = 0; k < kNV; ++k) dp += r[k].x * c[k].x + r[k].y * c[k].y + r[k].z * c[k].z + r[k].w * c[k].w;
                dp = grp_sum32(dp);
                const float ev = 0.001f * dp - 0.0005f;
#pragma unroll
                for (int k = 0; k < kNV; ++k) acc[k] += ev * c[k];
            } else if (!(a.mode & 1)) {
                f4 *wo = reinterpret_cast<f4 *>(a.own_out) + (size_t)row * kD4, *ao = reinterpret_cast<f4 *>(a.acc_out) + (size_t)row * kD4;
#pragma unroll
                for (int k = 0; k < kNV; ++k) {
                    const int i4 = lg + k * kLPR;
                    if (i4 < kD4) {
                        const f4 A = c[k] + acc[k] * acc[k];
                        ao[i4] = A;
                        wo[i4] = r[k] - 0.05f * acc[k] / (f4{sqrtf(A.x), sqrtf(A.y), sqrtf(A.z), sqrtf(A.w)} + 1e-7f);
                    }
                }
            }
        }
    };

    // prologue: D - 1 entries in flight
    for (int t = 0; t < D - 1 && t < Tw; ++t) issue(t);
    // steady state: entry t + D - 1 is issued, then entry t is waited for with the (D - 1) x NV younger DMAs left in flight
    int t = 0;
    for (; t + D - 1 < Tw; ++t) {
        issue(t + D - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * kNV) : "memory");
        consume(t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; t < Tw; ++t) consume(t);
}

// ---- the present structure, reduced: per trip U partner rows into registers, W waves per SIMD allowed ------------------
// mode bit 3: a 4-byte partner-bias gather per pair (lane 0 of the group), as the product does
template <int U, int W>
__global__ __launch_bounds__(kBlock, W) void trip_kernel(Args a)
{
    __shared__ uint32_t ent[(kBlock / kLPR) * kMaxEntries];
    const int lg = threadIdx.x % kLPR, grp = threadIdx.x / kLPR;
    const int gi = blockIdx.x * (kBlock / kLPR) + grp;
    int T = 0, off = 0;
    if (gi < a.ngroups) { off = a.off[gi]; T = a.off[gi + 1] - off; }
    uint32_t *my = ent + grp * kMaxEntries;
    for (int i = lg; i < T; i += kLPR) my[i] = a.entries[off + i];
    f4 r[kNV], acc[kNV], A[kNV];
    auto load = [&](f4(&dst)[kNV], const float *base, uint32_t row) {
        const f4 *src = reinterpret_cast<const f4 *>(base) + (size_t)row * kD4;
#pragma unroll
        for (int k = 0; k < kNV; ++k) {
            const int i4 = lg + k * kLPR;
            const f4 v = src[i4 < kD4 ? i4 : kD4 - 1];
            dst[k] = i4 < kD4 ? v : f4{0, 0, 0, 0};
        }
    };
    int t = 0;
    while (t < T) {
        const uint32_t e = my[t];
        const uint32_t type = e >> 30, row = e & 0x3fffffffu;
        if (type == kOwn) {
            load(r, a.own, row);
            load(A, a.acc_in, row);           // (the product parks it in LDS; here it simply travels with the own row)
#pragma unroll
            for (int k = 0; k < kNV; ++k) acc[k] = f4{0, 0, 0, 0};
            ++t;
        } else if (type == kPair) {
            // up to U consecutive pairs (the product's trips also stop at chunk ends: 32 pairs; here at the run's end)
            uint32_t rows[U];
            int n = 0;
#pragma unroll
            for (int x = 0; x < U; ++x) {
                const uint32_t ex = t + x < T ? my[t + x] : 0u;
                const bool ok = n == x && t + x < T && (ex >> 30) == kPair;
                rows[x] = ok ? (ex & 0x3fffffffu) : rows[0];
                n = ok ? x + 1 : n;
            }
            f4 c[U][kNV];
            float bv[U];
#pragma unroll
            for (int x = 0; x < U; ++x) {
                load(c[x], a.partner, rows[x]);
                bv[x] = 0.f;
                if ((a.mode & 8) && lg == 0) bv[x] = a.bias[rows[x]];
            }
#pragma unroll
            for (int x = 0; x < U; ++x) {
                float dp = bv[x];
#pragma unroll
                for (int k = 0; k < kNV; ++k) dp += r[k].x * c[x][k].x + r[k].y * c[x][k].y + r[k].z * c[x][k].z + r[k].w * c[x][k].w;
                dp = grp_sum32(dp);
                const float ev = x < n ? 0.001f * dp - 0.0005f : 0.f;
#pragma unroll
                for (int k = 0; k < kNV; ++k) acc[k] += ev * c[x][k];
            }
            t += n;
        } else {
            if (!(a.mode & 1)) {
            f4 *wo = reinterpret_cast<f4 *>(a.own_out) + (size_t)row * kD4, *ao = reinterpret_cast<f4 *>(a.acc_out) + (size_t)row * kD4;
#pragma unroll
            for (int k = 0; k < kNV; ++k) {
                const int i4 = lg + k * kLPR;
                if (i4 < kD4) {
                    const f4 An = A[k] + acc[k] * acc[k];
                    ao[i4] = An;
                    wo[i4] = r[k] - 0.05f * acc[k] / (f4{sqrtf(An.x), sqrtf(An.y), sqrtf(An.z), sqrtf(An.w)} + 1e-7f);
                }
            }
            }
            ++t;
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------
struct Side { std::vector<uint32_t> entries; std::vector<int> off; long pairs = 0, ids = 0; };

static Side make_side(int V, long B, int per, int cap, uint64_t seed, int fold)
{
    std::vector<double> cdf(V);
    double s = 0;
    for (int i = 0; i < V; ++i) { s += 1.0 / (i + 1); cdf[i] = s; }
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> u(0.0, s);
    auto zipf = [&]() { return (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u(rng)) - cdf.begin()); };
    std::vector<std::pair<uint32_t, uint32_t>> pr(B);
    for (long i = 0; i < B; ++i) { pr[i].first = zipf(); uint32_t p; do p = zipf(); while (p == pr[i].first); pr[i].second = fold ? p % fold : p; }
    std::sort(pr.begin(), pr.end(), [](auto &x, auto &y) { return x.first < y.first; });
    // chunks of <= cap pairs of one id
    std::vector<long> cstart;
    for (long i = 0; i < B;) {
        long j = i;
        while (j < B && pr[j].first == pr[i].first) ++j;
        for (long k = i; k < j; k += cap) cstart.push_back(k);
        i = j;
    }
    cstart.push_back(B);
    const long nchunks = (long)cstart.size() - 1;
    Side sd;
    sd.pairs = B;
    sd.off.push_back(0);
    for (long c0 = 0; c0 < nchunks;) {
        // a group: up to `per` consecutive chunks, at most 128 pairs... but at least one chunk
        long c1 = c0, np = 0;
        while (c1 < nchunks && c1 - c0 < per && (c1 == c0 || np + (cstart[c1 + 1] - cstart[c1]) <= 128)) { np += cstart[c1 + 1] - cstart[c1]; ++c1; }
        // a run that holds ALL chunks of its id ends in ACC (apply here); any other run stops after its pairs (the product
        // stores a partial row for the apply launch: one row written per ~12 chunks of a heavy id, not modelled)
        auto whole = [&](long ca, long cb) {      // chunks [ca, cb) of one id: the id starts at ca and ends at cb?
            const uint32_t u_ = pr[cstart[ca]].first;
            const bool starts = cstart[ca] == 0 || pr[cstart[ca] - 1].first != u_;
            const bool ends = cstart[cb] == B || pr[cstart[cb]].first != u_;
            return starts && ends;
        };
        long run0 = c0;
        for (long c = c0; c <= c1; ++c) {
            const bool brk = c == c1 || pr[cstart[c]].first != pr[cstart[run0]].first;
            if (brk && c > run0) {
                const uint32_t u_ = pr[cstart[run0]].first;
                sd.entries.push_back(kOwn << 30 | u_);
                ++sd.ids;
                for (long i = cstart[run0]; i < cstart[c]; ++i) sd.entries.push_back(kPair << 30 | pr[i].second);
                if (whole(run0, c)) sd.entries.push_back(kAcc << 30 | u_);
                run0 = c;
            }
        }
        if ((long)sd.entries.size() - sd.off.back() > kMaxEntries) { fprintf(stderr, "group too long\n"); exit(1); }
        sd.off.push_back((int)sd.entries.size());
        c0 = c1;
    }
    return sd;
}

template <class L>
static float time_us(L launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char **argv)
{
    const int V = argc > 1 ? atoi(argv[1]) : 400000;
    const long B = argc > 2 ? atol(argv[2]) : 1048576;
    const int per = argc > 3 ? atoi(argv[3]) : 12;
    const int fold = argc > 4 ? atoi(argv[4]) : 0;
    const size_t tbytes = (size_t)V * kD4 * 16;
    float *own, *partner, *acc, *own_out, *acc_out, *bias;
    CK(hipMalloc(&bias, (size_t)V * 4)); CK(hipMemset(bias, 0, (size_t)V * 4));
    CK(hipMalloc(&own, tbytes)); CK(hipMalloc(&partner, tbytes)); CK(hipMalloc(&acc, tbytes)); CK(hipMalloc(&own_out, tbytes)); CK(hipMalloc(&acc_out, tbytes));
    {
        std::vector<float> h((size_t)V * kD4 * 4);
        std::mt19937 rng(1);
        std::uniform_real_distribution<float> u(-0.05f, 0.05f);
        for (auto &x : h) x = u(rng);
        CK(hipMemcpy(own, h.data(), tbytes, hipMemcpyHostToDevice));
        for (auto &x : h) x = u(rng);
        CK(hipMemcpy(partner, h.data(), tbytes, hipMemcpyHostToDevice));
        for (auto &x : h) x = 0.1f;
        CK(hipMemcpy(acc, h.data(), tbytes, hipMemcpyHostToDevice));
    }
    // ---- check on a small case: ring<4> against the trip kernel (same arithmetic order per run: pair by pair)
    {
        Side s = make_side(2000, 20000, 12, 32, 7, 0);
        uint32_t *de; int *dof;
        CK(hipMalloc(&de, s.entries.size() * 4)); CK(hipMalloc(&dof, s.off.size() * 4));
        CK(hipMemcpy(de, s.entries.data(), s.entries.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dof, s.off.data(), s.off.size() * 4, hipMemcpyHostToDevice));
        const int ng = (int)s.off.size() - 1, nb = (ng + 7) / 8;
        const size_t sb = (size_t)2000 * kD4 * 16;
        std::vector<float> o1(sb / 4), o2(sb / 4), a1(sb / 4), a2(sb / 4);
        Args a{de, dof, ng, own, partner, acc, own_out, acc_out, bias, 0};
        CK(hipMemset(own_out, 0, sb)); CK(hipMemset(acc_out, 0, sb));
        hipLaunchKernelGGL((trip_kernel<4, 3>), dim3(nb), dim3(kBlock), 0, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o1.data(), own_out, sb, hipMemcpyDeviceToHost)); CK(hipMemcpy(a1.data(), acc_out, sb, hipMemcpyDeviceToHost));
        CK(hipMemset(own_out, 0, sb)); CK(hipMemset(acc_out, 0, sb));
        hipLaunchKernelGGL(ring_kernel<4>, dim3(nb), dim3(kBlock), 0, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o2.data(), own_out, sb, hipMemcpyDeviceToHost)); CK(hipMemcpy(a2.data(), acc_out, sb, hipMemcpyDeviceToHost));
        size_t bad = 0;
        double md = 0;
        for (size_t i = 0; i < o1.size(); ++i) {
            const double dd = std::fabs((double)o1[i] - o2[i]) + std::fabs((double)a1[i] - a2[i]);
            md = std::max(md, dd);
            if (dd > 1e-6) ++bad;
        }
        printf("check (V 2000, 20000 pairs, %d groups): %zu of %zu values differ, max |diff| %.3g\n", ng, bad, o1.size(), md);
        CK(hipFree(de)); CK(hipFree(dof));
        if (bad) return 2;
    }
    Side s = make_side(V, B, per, 32, 11, fold);
    uint32_t *de; int *dof;
    CK(hipMalloc(&de, s.entries.size() * 4)); CK(hipMalloc(&dof, s.off.size() * 4));
    CK(hipMemcpy(de, s.entries.data(), s.entries.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dof, s.off.data(), s.off.size() * 4, hipMemcpyHostToDevice));
    const int ng = (int)s.off.size() - 1, nb = (ng + 7) / 8;
    Args a{de, dof, ng, own, partner, acc, own_out, acc_out, bias, 0};
    const double rows = (double)s.entries.size();
    const double alg = s.ids * 4.0 * (kD4 * 16) + B * 16.0;           // own + acc read, two rows written, the pair stream
    printf("V %d, B %ld, per %d, partner fold %d: %d groups, %ld run entries (ids x groups), %.0f row loads, %.1f entries per group; algorithmic %.3f GB per side\n",
           V, B, per, fold, ng, s.ids, rows, rows / ng, alg / 1e9);
#define RUN(NAME, K) { const float us = time_us([&] { hipLaunchKernelGGL(K, dim3(nb), dim3(kBlock), 0, 0, a); }, 20); \
    printf("%-28s %8.1f us   %.2f row loads/ns   loads %.2f TB/s   algorithmic %.2f TB/s\n", NAME, us, rows / us / 1e3, rows * (kD4 * 16) / us / 1e6, alg / us / 1e6); fflush(stdout); }
    for (int mode : {0, 8, 1}) {
        a.mode = mode;
        printf("-- mode %d (%s%s)\n", mode, mode & 1 ? "no stores " : "", mode & 8 ? "a 4-byte partner-bias gather per pair" : "");
        RUN("trip U = 4, <= 3 waves/SIMD", (trip_kernel<4, 3>));
        RUN("trip U = 8, <= 3 waves/SIMD", (trip_kernel<8, 3>));
        RUN("trip U = 2, <= 8 waves/SIMD", (trip_kernel<2, 8>));
        RUN("trip U = 4, <= 8 waves/SIMD", (trip_kernel<4, 8>));
        RUN("trip U = 4, <= 4 waves/SIMD", (trip_kernel<4, 4>));
        RUN("trip U = 8, <= 4 waves/SIMD", (trip_kernel<8, 4>));
        RUN("trip U = 2, <= 4 waves/SIMD", (trip_kernel<2, 4>));
        RUN("trip U = 2, <= 5 waves/SIMD", (trip_kernel<2, 5>));
        RUN("trip U = 2, <= 6 waves/SIMD", (trip_kernel<2, 6>));
        RUN("trip U = 1, <= 8 waves/SIMD", (trip_kernel<1, 8>));
    }
    for (int mode : {0}) {
        a.mode = mode;
        printf("-- mode %d (%s%s%s)\n", mode, mode & 1 ? "no stores " : "", mode & 2 ? "own/acc loads folded " : "", mode & 4 ? "stores folded" : "");
        RUN("ring D = 2", ring_kernel<2>);
        RUN("ring D = 3", ring_kernel<3>);
        RUN("ring D = 5", ring_kernel<5>);
        RUN("ring D = 8", ring_kernel<8>);
#if D4 <= 32
        RUN("ring D = 12", ring_kernel<12>); RUN("ring D = 16", ring_kernel<16>);
#endif
    }
    return 0;
}
