// Host thread pool of libslam_pf (header-only so that tools/tsan_pool.cpp can run it under ThreadSanitizer).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// A few persistent host threads for the caller-side loops that stay on the CPU by design: the
// per-particle cos/sin of pf_predict come from the host libm (S3P:860; the device's differ in the
// last place for some arguments), the same function on every element whoever calls it.
class HostPool {
public:
    explicit HostPool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // fn(chunk) for chunk = 0 .. nchunks-1, each exactly once, by the caller and by whichever workers are awake in
    // time; returns when every chunk is done. Chunks are claimed from a counter, so the call does not wait for a
    // worker that wakes late (waking fifteen threads on one mutex takes longer than the work at 10 k particles):
    // such a worker finds nothing left to claim. A job lives in a shared object, so a late worker never touches
    // the next job's counter with this job's function.
    void run_chunks(int nchunks, const std::function<void(int)>& fn) {
        auto job = std::make_shared<Job>();
        job->fn = &fn; job->nchunks = nchunks;
        { std::lock_guard<std::mutex> lk(mu_); job_ = job; ++gen_; }
        cv_.notify_all();
        work(*job);
        // (the last chunks may still be running on workers: they are short)
        for (int spin = 0; job->done.load(std::memory_order_acquire) < nchunks; ++spin) {
            if (spin < 2000) __builtin_ia32_pause(); else std::this_thread::yield();
        }
        { std::lock_guard<std::mutex> lk(mu_); if (job_ == job) job_.reset(); }
    }
private:
    struct Job {
        const std::function<void(int)>* fn = nullptr;     // valid until done == nchunks (the caller waits for that)
        int nchunks = 0;
        std::atomic<int> next{0}, done{0};
    };
    static void work(Job& j) {
        for (;;) {
            const int c = j.next.fetch_add(1, std::memory_order_relaxed);
            if (c >= j.nchunks) return;
            (*j.fn)(c);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void run() {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::shared_ptr<Job> job_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};
