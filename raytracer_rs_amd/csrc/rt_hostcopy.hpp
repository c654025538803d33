// Host-side helper of the C ABI: copies out of pinned staging memory into the caller's (pageable) frame with several
// threads.  One thread moves about 10 GB/s, a PCIe 5 link delivers 50+; a 3840x2160 f64 frame is 199 MB.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

namespace rt {

struct CopySegment {
    void *dst;
    const void *src;
    size_t bytes;
};

// A small pool of worker threads, started on first use and kept for the life of the process.  run() copies every
// segment and returns when all bytes have landed; segments are cut into slices that the workers (and the calling
// thread) take from a shared counter.  One run() at a time (callers serialise through the pool's own mutex).
class HostCopyPool {
public:
    static HostCopyPool &instance() {
        static HostCopyPool pool;
        return pool;
    }

    void run(const std::vector<CopySegment> &segments) {
        size_t total = 0;
        for (const CopySegment &s : segments) total += s.bytes;
        // (a process forked after the pool was built has the pool's bookkeeping but none of its threads: it copies alone)
        if (total < (4u << 20) || workers_.empty() || getpid() != owner_) {  // not worth waking anybody
            for (const CopySegment &s : segments) std::memcpy(s.dst, s.src, s.bytes);
            return;
        }
        std::lock_guard<std::mutex> serial(run_mutex_);
        slices_.clear();
        for (const CopySegment &s : segments)
            for (size_t off = 0; off < s.bytes; off += SLICE)
                slices_.push_back(CopySegment{(char *)s.dst + off, (const char *)s.src + off, s.bytes - off < SLICE ? s.bytes - off : SLICE});
        next_.store(0);
        pending_.store((int)workers_.size());
        {
            std::lock_guard<std::mutex> lock(mutex_);
            generation_++;
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [&] { return pending_.load() == 0; });
    }

private:
    static constexpr size_t SLICE = 1u << 20;
    pid_t owner_ = 0;  // the process whose threads these are

    HostCopyPool() {
        owner_ = getpid();
        unsigned n = std::thread::hardware_concurrency();
        if (const char *e = std::getenv("RT_COPY_THREADS")) n = (unsigned)std::strtoul(e, nullptr, 10) + 1;
        n = n > 8 ? 8 : n;  // the caller's thread is one of them
        for (unsigned i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostCopyPool() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
            generation_++;
        }
        wake_.notify_all();
        for (std::thread &t : workers_) t.join();
    }

    void work() {
        for (;;) {
            const size_t k = next_.fetch_add(1);
            if (k >= slices_.size()) return;
            std::memcpy(slices_[k].dst, slices_[k].src, slices_[k].bytes);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            work();
            if (pending_.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lock(mutex_);
                done_.notify_all();
            }
        }
    }

    std::vector<std::thread> workers_;
    std::vector<CopySegment> slices_;
    std::atomic<size_t> next_{0};
    std::atomic<int> pending_{0};
    std::mutex mutex_, run_mutex_;
    std::condition_variable wake_, done_;
    unsigned long generation_ = 0;
    bool stop_ = false;
};

}  // namespace rt
