// Development aid: the library's host pool under ThreadSanitizer (CPU only).
//   g++ -O1 -g -std=c++17 -fsanitize=thread -pthread tools/tsan_pool.cpp -o /tmp/tsan_pool && /tmp/tsan_pool
// Thousands of jobs of 1 to 300 chunks back to back (so that workers waking late for one job meet the next),
// every chunk must run exactly once and its writes must be visible to the caller on return.
#include "../slam_amd/csrc/pf_hostpool.h"
#include <cstdio>
#include <cstdlib>
int main() {
    for (int threads : {1, 3, 15}) {
        HostPool pool(threads);
        std::vector<int> hits(300);
        std::vector<double> out(300);
        long total = 0;
        for (int job = 0; job < 20000; ++job) {
            const int n = 1 + (job * 7919) % 300;
            std::fill(hits.begin(), hits.begin() + n, 0);
            pool.run_chunks(n, [&](int c) { hits[(size_t)c] += 1; out[(size_t)c] = job + 0.5 * c; });
            for (int c = 0; c < n; ++c) {
                if (hits[(size_t)c] != 1 || out[(size_t)c] != job + 0.5 * c) { std::printf("job %d chunk %d: hits %d\n", job, c, hits[(size_t)c]); return 1; }
            }
            total += n;
        }
        std::printf("%d workers: %ld chunks, each once\n", threads, total);
    }
    return 0;
}
