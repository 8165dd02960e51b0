// Pure-HIP reproductions (no torch, no libglove_hip.so) of the two hipGraph problems DESIGN.md §7 records.
//
//   repro_hipgraph fork [--runners N] [--big-args 0|1] [--destroy-streams 0|1] [--destroy-events 0|1] [--keep-alive 0|1]
//       Round 3: tests/rccl_graph_case.py built four "runners" one after the other; each captured bursts whose index builds
//       ran on side streams forked off the capturing stream (events), with kernel nodes that carry ~2 KB by-value argument
//       blocks (eight plans per launch), and dropped its events right after the capture.  With the FIRST runner's graphs still
//       alive, the replay of the FOURTH runner's graph segfaulted inside hipGraphLaunch.  This mode rebuilds that sequence:
//       per runner three streams, a capture on the first with two forked branches of big-argument kernels joined back, an
//       instantiated exec that is launched a few times; the switches pick which of the suspected conditions hold.
//   repro_hipgraph memset [--replays N] [--builds N]
//       Round 2/3: a hipMemsetAsync node of 32 bytes inside a captured graph came back wrong on the second replay
//       (counts[4..7] held host-pointer-like words).  This mode captures `builds` x (memset 32 B -> kernel that appends
//       behind the zeroed counter), poisons the buffer between replays and checks every replay.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/repro_hipgraph tools/repro_hipgraph.hip      Exit code 0 = nothing reproduced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Slot { int *out; int n; int pad[60]; };          // 256 bytes
struct BigArgs { Slot s[8]; };                            // 2 KB by value, like eight plans per launch
struct SmallArgs { Slot s[1]; };

template <class A>
__global__ void branch_kernel(A a, int round)
{
    const Slot &s = a.s[blockIdx.z % (sizeof(A) / sizeof(Slot))];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < s.n) s.out[i] = s.out[i] * 3 + round + (int)blockIdx.z;
}

__global__ void join_kernel(const int *a, const int *b, int *sum, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sum[i] += a[i] ^ b[i];
}

struct Runner {
    hipStream_t st[3];
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int *buf[2], *sum;
    std::vector<hipEvent_t> events;
};

static int flag(int argc, char **argv, const char *name, int dflt)
{
    for (int i = 2; i + 1 < argc; ++i)
        if (!strcmp(argv[i], name)) return atoi(argv[i + 1]);
    return dflt;
}

static int run_fork(int argc, char **argv)
{
    const int runners = flag(argc, argv, "--runners", 4), big = flag(argc, argv, "--big-args", 1);
    const int destroy_streams = flag(argc, argv, "--destroy-streams", 1), destroy_events = flag(argc, argv, "--destroy-events", 1);
    const int keep_alive = flag(argc, argv, "--keep-alive", 1), groups = flag(argc, argv, "--groups", 3);
    const int n = 1 << 16;
    printf("fork: runners %d big-args %d destroy-streams %d destroy-events %d keep-alive %d groups %d\n", runners, big, destroy_streams,
           destroy_events, keep_alive, groups);
    fflush(stdout);
    std::vector<Runner> rs(runners);
    for (int r = 0; r < runners; ++r) {
        Runner &R = rs[r];
        for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&R.st[i], hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) { CK(hipMalloc(&R.buf[i], n * 4)); CK(hipMemset(R.buf[i], 0, n * 4)); }
        CK(hipMalloc(&R.sum, n * 4)); CK(hipMemset(R.sum, 0, n * 4));
        CK(hipDeviceSynchronize());
        auto ev = [&]() { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); R.events.push_back(e); return e; };
        CK(hipStreamBeginCapture(R.st[0], hipStreamCaptureModeThreadLocal));
        hipEvent_t start = ev();
        CK(hipEventRecord(start, R.st[0]));
        hipEvent_t stepped[2] = {start, start};
        for (int g = 0; g < groups; ++g) {
            // group g's "build" on side stream g % 2 (forked off the capturing stream), then its "steps" on the main stream
            hipStream_t side = R.st[1 + g % 2];
            CK(hipStreamWaitEvent(side, stepped[g % 2], 0));
            BigArgs a = {};
            for (int z = 0; z < 8; ++z) { a.s[z].out = R.buf[g % 2]; a.s[z].n = n; }
            for (int k = 0; k < 3; ++k) {
                if (big) hipLaunchKernelGGL(branch_kernel<BigArgs>, dim3(n / 256, 1, 8), dim3(256), 0, side, a, g * 10 + k);
                else { SmallArgs sa = {{a.s[0]}}; hipLaunchKernelGGL(branch_kernel<SmallArgs>, dim3(n / 256, 1, 8), dim3(256), 0, side, sa, g * 10 + k); }
            }
            hipEvent_t built = ev();
            CK(hipEventRecord(built, side));
            CK(hipStreamWaitEvent(R.st[0], built, 0));
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(join_kernel, dim3(n / 256), dim3(256), 0, R.st[0], R.buf[0], R.buf[1], R.sum, n);
            hipEvent_t done = ev();
            CK(hipEventRecord(done, R.st[0]));
            stepped[g % 2] = done;
        }
        // every forked stream joins back before the capture ends
        for (int i = 1; i < 3; ++i) { hipEvent_t j = ev(); CK(hipEventRecord(j, R.st[i])); CK(hipStreamWaitEvent(R.st[0], j, 0)); }
        CK(hipStreamEndCapture(R.st[0], &R.graph));
        CK(hipGraphInstantiate(&R.exec, R.graph, nullptr, nullptr, 0));
        if (destroy_events) { for (hipEvent_t e : R.events) CK(hipEventDestroy(e)); R.events.clear(); }
        for (int k = 0; k < 5; ++k) CK(hipGraphLaunch(R.exec, R.st[0]));
        CK(hipStreamSynchronize(R.st[0]));
        printf("runner %d: captured and replayed 5 times\n", r);
        fflush(stdout);
        if (destroy_streams) { CK(hipStreamDestroy(R.st[1])); CK(hipStreamDestroy(R.st[2])); }     // the exec outlives the streams it was captured on
        if (!keep_alive) { CK(hipGraphExecDestroy(R.exec)); CK(hipGraphDestroy(R.graph)); R.exec = nullptr; }
    }
    // every earlier exec is still alive: replay them all again, newest first (round 3's crash was the newest one's replay)
    for (int r = runners - 1; r >= 0; --r) {
        if (!rs[r].exec) continue;
        for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(rs[r].exec, rs[r].st[0]));
        CK(hipStreamSynchronize(rs[r].st[0]));
        printf("runner %d: replayed again with %d other execs alive\n", r, runners - 1);
        fflush(stdout);
    }
    printf("fork: nothing reproduced\n");
    return 0;
}

__global__ void append_kernel(int *counts, int *list, int cap, int tag)
{
    // like side_emit behind the memset: counts[4] is a cursor that must start at zero
    if (threadIdx.x < 8 && blockIdx.x == 0) {
        const int slot = atomicAdd(counts + 4, 1);
        if (slot < cap) list[slot] = tag * 100 + threadIdx.x;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = tag;
}

__global__ void poison_kernel(int *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 0x5a5a5a5a; }

static int run_memset(int argc, char **argv)
{
    const int replays = flag(argc, argv, "--replays", 6), builds = flag(argc, argv, "--builds", 3);
    printf("memset: replays %d builds %d\n", replays, builds);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<int *> counts(builds), lists(builds);
    for (int b = 0; b < builds; ++b) { CK(hipMalloc(&counts[b], 32)); CK(hipMalloc(&lists[b], 64 * 4)); }
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int b = 0; b < builds; ++b) {
        CK(hipMemsetAsync(counts[b], 0, 32, st));                  // 8 words, as glove_plan_build zeroed plan->counts
        hipLaunchKernelGGL(append_kernel, dim3(1), dim3(64), 0, st, counts[b], lists[b], 64, b + 1);
    }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    int bad = 0;
    for (int r = 0; r < replays; ++r) {
        for (int b = 0; b < builds; ++b) hipLaunchKernelGGL(poison_kernel, dim3(1), dim3(64), 0, st, counts[b], 8);
        CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        for (int b = 0; b < builds; ++b) {
            int h[8];
            CK(hipMemcpy(h, counts[b], 32, hipMemcpyDeviceToHost));
            const bool ok = h[0] == b + 1 && h[4] == 8 && h[1] == 0 && h[2] == 0 && h[3] == 0 && h[5] == 0 && h[6] == 0 && h[7] == 0;
            if (!ok) {
                ++bad;
                printf("replay %d build %d: counts = %d %d %d %d | %d %d %d %d\n", r, b, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            }
        }
    }
    printf(bad ? "memset: REPRODUCED (%d bad reads)\n" : "memset: nothing reproduced (%d bad reads)\n", bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: repro_hipgraph fork|memset [options]\n"); return 2; }
    if (!strcmp(argv[1], "fork")) return run_fork(argc, argv);
    if (!strcmp(argv[1], "memset")) return run_memset(argc, argv);
    return 2;
}
