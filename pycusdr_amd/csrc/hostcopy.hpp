// hostcopy.hpp -- the host copy worker behind mfb_hostcopy_* (include/mfbank.h).  Plain C++17, no GPU: also compiled on its own under
// ThreadSanitizer / AddressSanitizer (tests/csrc/hostcopy_sanitize.cpp).
#pragma once
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

// One thread, one queue: the receive loop hands it the chunk -> window copies of the next batch and runs the host stages of the
// previous batch meanwhile.  The worker spins for a few microseconds before it sleeps (chunks arrive in bursts; a futex wake costs
// more than a 128 KiB copy).
// a polite spin: the x86 pause, the aarch64 yield, a compiler barrier anywhere else
static inline void mfb_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

struct mfb_hostcopy {
    struct Job {
        void *dst;
        const void *src;
        size_t bytes;
    };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> jobs;
    std::atomic<long long> submitted{0}, done{0};
    bool stop = false;
    std::thread worker;

    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (jobs.empty() && !stop) {
                    lk.unlock();
                    for (int spin = 0; spin < 4000 && submitted.load(std::memory_order_acquire) == done.load(std::memory_order_relaxed); ++spin)
                        mfb_cpu_relax();
                    lk.lock();
                }
                cv_work.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;              // stop, nothing left
                j = jobs.front();
                jobs.pop_front();
            }
            memcpy(j.dst, j.src, j.bytes);
            {
                std::lock_guard<std::mutex> lk(mu);
                done.fetch_add(1, std::memory_order_release);
            }
            cv_done.notify_all();
        }
    }
    bool start() {
        try {
            worker = std::thread([this] { run(); });
        } catch (...) {
            return false;
        }
        return true;
    }
    bool submit(void *dst, const void *src, size_t bytes) {       // false: the queue could not grow (nothing was queued)
        try {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back({dst, src, bytes});
            submitted.fetch_add(1, std::memory_order_release);
        } catch (...) {
            return false;
        }
        cv_work.notify_one();
        return true;
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) == submitted.load(std::memory_order_acquire); });
    }
    void shutdown() {          // finishes what was submitted, then stops the thread
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        if (worker.joinable()) worker.join();
    }
};

