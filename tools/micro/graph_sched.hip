// How does this HIP runtime schedule the branches of a stream-captured graph?  Stand-alone probe: spin kernels that stamp their start and end
// (100 MHz wall clock) captured into small two- and three-stream graphs whose shapes mirror the training step's (fork, join, a branch that hangs off
// the middle of a chain, the order in which the branches were captured); prints every node's start/end relative to the graph's first node.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/graph_sched.hip -o tools/_build/graph_sched
// run:   tools/_build/graph_sched            (LD_LIBRARY_PATH=<torch>/lib to probe the runtime bundled with torch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long *stamps, int id, int us) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamps[2 * id] = t0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) {}
    if (threadIdx.x == 0) stamps[2 * id + 1] = wall_clock64();
}

// replay-boundary probe: every replay's first node stamps its start, its last node its end (slot = replay index, counted on the device)
__global__ void first_node(unsigned long long *stamps, unsigned int *replay, int us) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamps[2 * *replay] = t0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) {}
}
__global__ void last_node(unsigned long long *stamps, unsigned int *replay, int us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) {}
    if (threadIdx.x == 0) { stamps[2 * *replay + 1] = wall_clock64(); *replay += 1; }
}

struct Probe {
    unsigned long long *stamps;
    std::vector<std::string> names;
    hipStream_t s[3];
    std::vector<hipEvent_t> events;
    Probe() {
        CK(hipMalloc(&stamps, 2 * 64 * sizeof(unsigned long long)));
        for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    }
    void k(int stream, const char *name, int us) {
        const int id = (int)names.size();
        names.push_back(name);
        spin<<<1, 64, 0, s[stream]>>>(stamps, id, us);
    }
    // stream `to` waits for what `from` holds now
    void dep(int from, int to) {
        hipEvent_t e;
        CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CK(hipEventRecord(e, s[from]));
        CK(hipStreamWaitEvent(s[to], e, 0));
        events.push_back(e);
    }
    hipEvent_t mark(int from) {
        hipEvent_t e;
        CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CK(hipEventRecord(e, s[from]));
        events.push_back(e);
        return e;
    }
    void wait(int to, hipEvent_t e) { CK(hipStreamWaitEvent(s[to], e, 0)); }
    template <class F> void run(const char *title, F body) {
        names.clear();
        CK(hipMemset(stamps, 0, 2 * 64 * sizeof(unsigned long long)));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        body();
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        std::vector<unsigned long long> h(2 * names.size());
        CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < names.size(); ++i) { if (h[2 * i] < t0) t0 = h[2 * i]; if (h[2 * i + 1] > t1) t1 = h[2 * i + 1]; }
        printf("== %s   (span %.0f us)\n", title, (t1 - t0) / 100.0);
        for (size_t i = 0; i < names.size(); ++i)
            printf("   %-6s %7.1f -> %7.1f\n", names[i].c_str(), (h[2 * i] - t0) / 100.0, (h[2 * i + 1] - t0) / 100.0);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
};

int main(int argc, char **argv) {
    // argv[1]: number of extra streams created (and used once) BEFORE the probe's own -- do the graph's internal streams share hardware queues with them?
    const int extra = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<hipStream_t> pad(extra);
    for (auto &x : pad) { CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking)); CK(hipMemsetAsync(nullptr, 0, 0, x)); }
    Probe p;
    if (argc > 2 && std::string(argv[2]) == "gap") {
        // the gap between two replays of the same graph on one stream: (a) one chain, (b) fork/join over two streams, (c) over three
        unsigned int *replay;
        CK(hipMalloc(&replay, sizeof(unsigned int)));
        for (int streams = 1; streams <= 3; ++streams) {
            CK(hipMemset(replay, 0, sizeof(unsigned int)));
            CK(hipMemset(p.stamps, 0, 2 * 64 * sizeof(unsigned long long)));
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(p.s[0], hipStreamCaptureModeThreadLocal));
            first_node<<<1, 64, 0, p.s[0]>>>(p.stamps, replay, 10);
            for (int k = 1; k < streams; ++k) p.dep(0, k);
            const int per_stream = getenv("PROBE_NODES") ? atoi(getenv("PROBE_NODES")) : 3;      // (does the gap grow with the graph's node count?)
            for (int k = 0; k < streams; ++k)
                for (int j = 0; j < per_stream; ++j) spin<<<1, 64, 0, p.s[k]>>>(p.stamps + 100, 0, 10);
            for (int k = 1; k < streams; ++k) p.dep(k, 0);
            last_node<<<1, 64, 0, p.s[0]>>>(p.stamps, replay, 10);
            CK(hipStreamEndCapture(p.s[0], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipStream_t launch = getenv("PROBE_NULL") ? nullptr : p.s[0];      // (torch replays on the stream that is current: by default the null stream)
            hipEvent_t evs[20];
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipGraphLaunch(ge, launch));
                if (getenv("PROBE_EVENT")) { CK(hipEventCreateWithFlags(&evs[rep], hipEventDisableTiming)); CK(hipEventRecord(evs[rep], launch)); }
            }
            CK(hipStreamSynchronize(launch));
            unsigned long long h[40];
            CK(hipMemcpy(h, p.stamps, sizeof(h), hipMemcpyDeviceToHost));
            printf("== replay gap, %d captured stream(s): span of a replay %.1f us; gaps between replays (us):", streams, (h[2 * 10 + 1] - h[2 * 10]) / 100.0);
            for (int rep = 5; rep < 19; ++rep) printf(" %.1f", (h[2 * (rep + 1)] - h[2 * rep + 1]) / 100.0);
            printf("\n");
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
        return 0;
    }
    if (argc > 2) {     // the step's head only: S = k_sample_rays, content chain C1..C6 captured first, block chain B1..B6 second (both children of S)
        for (int rep = 0; rep < 2; ++rep)
        p.run("H1 content chain captured first", [&] {
            p.k(0, "R", 5); p.k(0, "S", 5);
            p.dep(0, 1);
            for (const char *n : {"C1", "C2", "C3", "C4", "C5", "C6"}) p.k(1, n, 20);
            for (const char *n : {"B1", "B2", "B3", "B4", "B5", "B6"}) p.k(0, n, 20);
            p.dep(1, 0);
            p.k(0, "L", 5);
        });
        for (int rep = 0; rep < 2; ++rep)
        p.run("H2 as H1 with a device-to-device copy node between B3 and B4 (the all-gather's place)", [&] {
            p.k(0, "R", 5); p.k(0, "S", 5);
            p.dep(0, 1);
            for (const char *n : {"C1", "C2", "C3", "C4", "C5", "C6"}) p.k(1, n, 20);
            for (const char *n : {"B1", "B2", "B3"}) p.k(0, n, 20);
            CK(hipMemcpyAsync(p.stamps + 100, p.stamps + 110, 64, hipMemcpyDeviceToDevice, p.s[0]));
            for (const char *n : {"B4", "B5", "B6"}) p.k(0, n, 20);
            p.dep(1, 0);
            p.k(0, "L", 5);
        });
        for (int rep = 0; rep < 2; ++rep)
        p.run("H3 as H2, block chain captured first", [&] {
            p.k(0, "R", 5); p.k(0, "S", 5);
            p.dep(0, 1);
            for (const char *n : {"B1", "B2", "B3"}) p.k(0, n, 20);
            CK(hipMemcpyAsync(p.stamps + 100, p.stamps + 110, 64, hipMemcpyDeviceToDevice, p.s[0]));
            for (const char *n : {"B4", "B5", "B6"}) p.k(0, n, 20);
            for (const char *n : {"C1", "C2", "C3", "C4", "C5", "C6"}) p.k(1, n, 20);
            p.dep(1, 0);
            p.k(0, "L", 5);
        });
        return 0;
    }
    const int U = 20;   // us per node
    p.run("T1 fork at R: side chain captured FIRST, main chain second, join at J", [&] {
        p.k(0, "R", U);
        p.dep(0, 1);
        for (const char *n : {"S1", "S2", "S3", "S4"}) p.k(1, n, U);
        for (const char *n : {"M1", "M2", "M3", "M4"}) p.k(0, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    p.run("T2 fork at R: main chain captured first, side chain second, join at J", [&] {
        p.k(0, "R", U);
        p.dep(0, 1);
        for (const char *n : {"M1", "M2", "M3", "M4"}) p.k(0, n, U);
        for (const char *n : {"S1", "S2", "S3", "S4"}) p.k(1, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    p.run("T3 a branch hangs off the MIDDLE of the main chain (after M2), captured after the whole main chain; join at J", [&] {
        p.k(0, "R", U);
        p.k(0, "M1", U);
        p.k(0, "M2", U);
        hipEvent_t e = p.mark(0);
        for (const char *n : {"M3", "M4", "M5", "M6"}) p.k(0, n, U);
        p.wait(1, e);
        for (const char *n : {"S1", "S2"}) p.k(1, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    p.run("T4 as T3, the branch captured right at the fork (before M3..M6)", [&] {
        p.k(0, "R", U);
        p.k(0, "M1", U);
        p.k(0, "M2", U);
        p.dep(0, 1);
        for (const char *n : {"S1", "S2"}) p.k(1, n, U);
        for (const char *n : {"M3", "M4", "M5", "M6"}) p.k(0, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    p.run("T5 as T3 plus a redundant second parent for M3 (event recorded after M1, waited on again before M3)", [&] {
        p.k(0, "R", U);
        p.k(0, "M1", U);
        hipEvent_t e1 = p.mark(0);
        p.k(0, "M2", U);
        hipEvent_t e = p.mark(0);
        p.wait(0, e1);
        for (const char *n : {"M3", "M4", "M5", "M6"}) p.k(0, n, U);
        p.wait(1, e);
        for (const char *n : {"S1", "S2"}) p.k(1, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    p.run("T6 the step's shape: side = content forward (C1..C3) from R, main = block forward (B1..B3), join at L (loss), then main = decoder backward "
          "(D1..D4) + block backward (E1, E2), side = content backward (X1, X2) hanging off L and captured LAST, join at J",
          [&] {
              p.k(0, "R", U);
              p.dep(0, 1);
              for (const char *n : {"C1", "C2", "C3"}) p.k(1, n, U);
              for (const char *n : {"B1", "B2", "B3"}) p.k(0, n, U);
              p.dep(1, 0);
              p.k(0, "L", U);
              hipEvent_t e = p.mark(0);
              for (const char *n : {"D1", "D2", "D3", "D4", "E1", "E2"}) p.k(0, n, U);
              p.wait(1, e);
              for (const char *n : {"X1", "X2"}) p.k(1, n, U);
              p.dep(1, 0);
              p.k(0, "J", U);
          });
    p.run("T7 as T6, content backward on a THIRD stream", [&] {
        p.k(0, "R", U);
        p.dep(0, 1);
        for (const char *n : {"C1", "C2", "C3"}) p.k(1, n, U);
        for (const char *n : {"B1", "B2", "B3"}) p.k(0, n, U);
        p.dep(1, 0);
        p.k(0, "L", U);
        hipEvent_t e = p.mark(0);
        for (const char *n : {"D1", "D2", "D3", "D4", "E1", "E2"}) p.k(0, n, U);
        p.wait(2, e);
        for (const char *n : {"X1", "X2"}) p.k(2, n, U);
        p.dep(2, 0);
        p.k(0, "J", U);
    });
    p.run("T8 as T6, content backward captured right behind L (before D1..)", [&] {
        p.k(0, "R", U);
        p.dep(0, 1);
        for (const char *n : {"C1", "C2", "C3"}) p.k(1, n, U);
        for (const char *n : {"B1", "B2", "B3"}) p.k(0, n, U);
        p.dep(1, 0);
        p.k(0, "L", U);
        p.dep(0, 1);
        for (const char *n : {"X1", "X2"}) p.k(1, n, U);
        for (const char *n : {"D1", "D2", "D3", "D4", "E1", "E2"}) p.k(0, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    p.run("T9 as T6, with a redundant second parent for D1 (event recorded after B3, before the join, waited on again before D1)", [&] {
        p.k(0, "R", U);
        p.dep(0, 1);
        for (const char *n : {"C1", "C2", "C3"}) p.k(1, n, U);
        for (const char *n : {"B1", "B2", "B3"}) p.k(0, n, U);
        hipEvent_t eb = p.mark(0);
        p.dep(1, 0);
        p.k(0, "L", U);
        hipEvent_t e = p.mark(0);
        p.wait(0, eb);
        for (const char *n : {"D1", "D2", "D3", "D4", "E1", "E2"}) p.k(0, n, U);
        p.wait(1, e);
        for (const char *n : {"X1", "X2"}) p.k(1, n, U);
        p.dep(1, 0);
        p.k(0, "J", U);
    });
    return 0;
}
