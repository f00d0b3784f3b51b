// The host copy worker of libmfbank (pycusdr_amd/csrc/hostcopy.hpp) on its own, for ThreadSanitizer / AddressSanitizer
// (tests/test_host_sanitizers.py): bursts of copies from one submitting thread, drains between bursts and none before shutdown,
// several workers side by side.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../pycusdr_amd/csrc/hostcopy.hpp"

static int one_worker(unsigned seed, int bursts) {
    mfb_hostcopy q;
    if (!q.start()) return 1;
    std::vector<float> src(1 << 18), dst(1 << 18), want(1 << 18, 0.f);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)(i * 7 + seed);
    unsigned s = seed;
    auto rnd = [&](unsigned n) {
        s = s * 1664525u + 1013904223u;
        return (s >> 8) % n;
    };
    int bad = 0;
    for (int b = 0; b < bursts; ++b) {
        const int jobs = 1 + (int)rnd(40);
        for (int j = 0; j < jobs; ++j) {
            const size_t n = 1 + rnd(20000), at = rnd((unsigned)(dst.size() - n)), from = rnd((unsigned)(src.size() - n));
            q.submit(dst.data() + at, src.data() + from, n * sizeof(float));
            for (size_t i = 0; i < n; ++i) want[at + i] = src[from + i];      // submission order: later copies win
        }
        if (b % 3 != 2) {                      // (every third burst runs into the next one without a drain)
            q.drain();
            for (size_t i = 0; i < dst.size(); i += 97) bad += dst[i] != want[i];
        }
    }
    q.shutdown();                              // finishes what was submitted
    for (size_t i = 0; i < dst.size(); ++i) bad += dst[i] != want[i];
    return bad;
}

int main() {
    int bad = one_worker(1, 200);
    std::vector<std::thread> ts;
    std::vector<int> res(4, -1);
    for (int t = 0; t < 4; ++t) ts.emplace_back([t, &res] { res[t] = one_worker(10 + t, 60); });
    for (auto &t : ts) t.join();
    for (int r : res) bad += r;
    {   // a worker that never gets a job
        mfb_hostcopy idle;
        if (!idle.start()) return 2;
        idle.drain();
        idle.shutdown();
    }
    if (bad) {
        printf("%d wrong elements\n", bad);
        return 1;
    }
    printf("copies ok\n");
    return 0;
}
