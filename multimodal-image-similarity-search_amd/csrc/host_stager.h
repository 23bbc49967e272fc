// host_stager.h — host buffers to HBM through a ring of pinned blocks.
//
// The reference hands the boundary HOST data (the processor's float32 pixel_values stay on the CPU, backend/app/utils.py:76;
// uploads are decoded on the host, backend/app/main.py:140-143). hipMemcpyAsync from pageable memory stages through the
// runtime's own bounce buffer one piece at a time on the calling thread: measured 26 GB/s of the ~55 the link gives
// (VERDICT r4 weak #8). Here the bounce buffer is ours: a ring of NSLOT pinned blocks (hipHostMalloc); a block is filled by a
// few host threads copying in parallel (one memcpy stream does not reach the link rate either) while the copy engine moves
// the previous blocks, and an event per block says when its bytes have left. The caller's thread is busy for the duration
// of the host copies — as it was inside hipMemcpyAsync — the GPU is not.
#pragma once
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "common.h"

struct HostStager {
    static constexpr int NSLOT = 4;
    size_t slot_bytes = 0;
    char* pin[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    bool used[NSLOT] = {false, false, false, false};
    int next_slot = 0;

    // worker pool: a generation counter announces a job, every worker copies its slice, the last one to finish wakes the caller
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    char* job_dst = nullptr;
    const char* job_src = nullptr;
    size_t job_bytes = 0;
    int job_gen = 0, job_pending = 0, nthreads = 1;
    bool stop = false;

    static void copy_slice(char* dst, const char* src, size_t n, int part, int parts) {
        // slices on 4 KB boundaries
        const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        const size_t lo = (size_t)part * per;
        if (lo >= n) return;
        const size_t len = (lo + per <= n) ? per : n - lo;
        memcpy(dst + lo, src + lo, len);
    }

    void worker(int id, int seen) {   // seen = the generation at the time the pool was started (a re-initialised ring must not replay the last job)
        for (;;) {
            char* d; const char* s; size_t n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || job_gen != seen; });
                if (stop) return;
                seen = job_gen;
                d = job_dst; s = job_src; n = job_bytes;
            }
            copy_slice(d, s, n, id, nthreads);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--job_pending == 0) cv_done.notify_one();
            }
        }
    }

    int init_blocks(size_t block_bytes) {
        for (int i = 0; i < NSLOT; ++i) {
            MM_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin[i]), block_bytes, hipHostMallocDefault));
            MM_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        return MMISS_OK;
    }
    // block_bytes in [1 MB, 256 MB], threads in [1, 64] (the caller clamps the options to that; anything else is refused: a
    // zero block would never advance h2d's loop). A ring that fails part-way is torn down again, nothing is left pinned.
    int init(size_t block_bytes, int threads) {
        if (slot_bytes) return MMISS_OK;
        if (block_bytes < ((size_t)1 << 20) || block_bytes > ((size_t)256 << 20) || threads < 1 || threads > 64)
            MM_FAIL(MMISS_ERR_ARG, "pinned staging ring: block of %zu bytes / %d threads outside 1..256 MB / 1..64", block_bytes, threads);
        const int rc = init_blocks(block_bytes);
        if (rc != MMISS_OK) { shutdown(); return rc; }
        slot_bytes = block_bytes;
        stop = false;
        nthreads = threads;
        for (int t = 1; t < nthreads; ++t) workers.emplace_back([this, t, g0 = job_gen] { worker(t, g0); });   // (the caller is slice 0)
        return MMISS_OK;
    }

    void parallel_copy(char* dst, const char* src, size_t n) {
        if (nthreads <= 1 || n < (size_t)(1 << 20)) { memcpy(dst, src, n); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            job_dst = dst; job_src = src; job_bytes = n;
            job_pending = nthreads - 1;
            ++job_gen;
        }
        cv_work.notify_all();
        copy_slice(dst, src, n, 0, nthreads);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return job_pending == 0; });
    }

    // dst_dev[0, bytes) = src_host[0, bytes), queued on `stream`; returns once the last block's host copy is done (its bytes
    // may still be crossing the link: stream order covers every consumer on `stream` or behind an event recorded on it)
    int h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t stream) {
        if (!slot_bytes) MM_FAIL(MMISS_ERR_ARG, "pinned staging ring used before init");
        const char* src = reinterpret_cast<const char*>(src_host);
        char* dst = reinterpret_cast<char*>(dst_dev);
        for (size_t off = 0; off < bytes; off += slot_bytes) {
            const size_t n = bytes - off < slot_bytes ? bytes - off : slot_bytes;
            const int s = next_slot;
            next_slot = (next_slot + 1) % NSLOT;
            if (used[s]) MM_HIP(hipEventSynchronize(done[s]));   // the block's previous bytes have left
            parallel_copy(pin[s], src + off, n);
            MM_HIP(hipMemcpyAsync(dst + off, pin[s], n, hipMemcpyHostToDevice, stream));
            MM_HIP(hipEventRecord(done[s], stream));
            used[s] = true;
        }
        return MMISS_OK;
    }

    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
        workers.clear();
        for (int i = 0; i < NSLOT; ++i) {
            if (done[i]) { (void)hipEventSynchronize(done[i]); (void)hipEventDestroy(done[i]); done[i] = nullptr; }
            if (pin[i]) { (void)hipHostFree(pin[i]); pin[i] = nullptr; }
            used[i] = false;
        }
        next_slot = 0;
        slot_bytes = 0;
    }
    ~HostStager() { shutdown(); }
};
