// pm_hostpool.h -- the CPU side of the host <-> HBM leg, free of HIP: the pool of copy threads (a queue of
// jobs cut into parts), the ring of pinned staging buffers with its retire thread, and the routines that
// move device results into pageable caller memory through them. What the device side has to offer is the
// small CopyBackend below; pm_hostpipe.hip implements it with hipMemcpyAsync / hipMemcpy2DAsync + events,
// tests/hostpool/harness.cpp with a thread that plays the DMA engine - so this unit runs, unchanged, under
// ThreadSanitizer and AddressSanitizer on a machine without a GPU (tests/test_hostpool_sanitizers.py).
//
// Reference context: the reference hands numpy arrays around (body_xy.py:3166 makes the planes,
// observation.py:876-905 maps a host cube); this is the machinery that gets the engine's results into such
// arrays at the rate of the link. No compute happens here: the threads move bytes (and write the constant
// NaN where the kernels' own pre-mask says nothing else can be).
#pragma once

#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// how far ahead of the copy the pool's block gather requests cache lines, and into which level
#ifndef PM_GATHER_AHEAD_BYTES
#define PM_GATHER_AHEAD_BYTES 4096
#endif
#ifndef PM_GATHER_HINT
#define PM_GATHER_HINT _MM_HINT_T0
#endif

namespace pmh {

// What the pool needs from the device side. `stream` is the backend's own notion of an in-order queue
// (a hipStream_t in the library). Every call returns 0 or a backend error code (kept, reported by finish()).
struct CopyBackend {
    virtual ~CopyBackend() {}
    virtual void thread_init() {}  // called once on the retire thread (hipSetDevice)
    // enqueue dst_host <- src_dev (dst is PINNED memory of alloc_pinned, or the caller's own pinned array)
    virtual int copy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream) = 0;
    // the same for `height` rows of `width` bytes, `dpitch` / `spitch` bytes apart
    virtual int copy_d2h_2d(void *dst_host, size_t dpitch, const void *src_dev, size_t spitch, size_t width, size_t height, void *stream) = 0;
    // mark / wait for the completion of everything enqueued on `stream` so far, one marker per staging slot
    virtual int record(int slot, void *stream) = 0;
    virtual int wait(int slot) = 0;
    virtual void *alloc_pinned(size_t bytes) = 0;
    virtual void free_pinned(void *p) = 0;
};

struct HostPool {
    CopyBackend *be = nullptr;
    // ---- pinned staging ring of the D2H leg
    static constexpr int kSlots = 4;
    char *stage[kSlots] = {};
    size_t stage_bytes = 0;
    // rows of a frame plane of which only the columns [xa, xb) hold anything but NaN (issue_disc)
    struct Rows {
        size_t nx = 0, n = 0;  // row length (pixels), rows
        size_t xa = 0, xb = 0;
    };
    struct Piece {
        int slot;
        char *dst;
        size_t bytes;
        Rows rows;  // rows.n != 0: the staged piece is the rectangle rows.n x (xb - xa), dst the first row
    };
    // pieces whose DMA has been enqueued, oldest first; the retire thread waits for each DMA and has
    // the pool copy the piece out, so the calling thread stays free to feed the next chunk
    std::deque<Piece> inflight;
    int next_slot = 0;
    std::thread retirer;
    std::mutex rmu;
    std::condition_variable cv_piece, cv_slot;
    int pieces_out = 0;           // issued and not yet copied out (<= kSlots + jobs without a slot)
    bool slot_busy[kSlots] = {};  // DMA in flight or being copied out (copy-outs may finish out of order)
    bool rstop = false;
    int rerror = 0;  // first backend error met by the retire thread

    ~HostPool()
    {
        stop_retirer();
        stop_workers();
        free_stage();
    }

    // (slot < 0: an asynchronous job that holds no staging slot, counted in pieces_out all the same)
    void release_slot(int slot, int err)
    {
        {
            std::lock_guard<std::mutex> lk(rmu);
            if (err != 0 && rerror == 0) rerror = err;
            if (slot >= 0) slot_busy[slot] = false;
            pieces_out--;
        }
        cv_slot.notify_all();
    }
    // Waits for each DMA in turn and hands the piece to the pool WITHOUT waiting for the copy-out: while
    // the pool copies piece k the thread is already waiting for the DMA of piece k + 1 (the wake-up
    // latencies of an event wait and of a pool job, ~0.1 ms each, used to sit between any two pieces).
    void retire_main()
    {
        be->thread_init();
        for (;;) {
            Piece pc;
            {
                std::unique_lock<std::mutex> lk(rmu);
                cv_piece.wait(lk, [&] { return rstop || !inflight.empty(); });
                if (inflight.empty()) return;  // rstop
                pc = inflight.front();
                inflight.pop_front();
            }
            const int e = be->wait(pc.slot);
            if (e != 0)
                release_slot(pc.slot, e);
            else if (pc.rows.n)
                rows_async(pc.dst, stage[pc.slot], pc.rows, pc.slot);
            else
                copy_async(pc.dst, stage[pc.slot], pc.bytes, pc.slot);
        }
    }
    void start_retirer()
    {
        if (retirer.joinable()) return;
        rstop = false;
        retirer = std::thread([this] { retire_main(); });
    }
    void stop_retirer()
    {
        if (!retirer.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(rmu);
            rstop = true;
        }
        cv_piece.notify_all();
        retirer.join();
    }
    // ---- copy pool: a queue of jobs, each cut into parts that any pool thread takes; threads move on
    // to the next job as soon as the parts of the front one are handed out
    struct Job {
        char *dst = nullptr;
        const char *src = nullptr;
        size_t total = 0, part = 0;  // units: bytes (copy) or blocks (gather)
        // gather: block i of dst is block list[i % nlist] of source plane i / nlist
        const int *list = nullptr;
        size_t nlist = 0, plane_bytes = 0;
        int shift = 0;  // log2 of the block size (>= 4)
        size_t nparts = 0;
        std::atomic<size_t> next{0}, finished{0};
        int slot = -1;         // copy-out of a staged piece: the staging slot it frees
        bool counted = false;  // an asynchronous job without a slot: completion decrements pieces_out
        bool done = false;     // (under mu)
        // rows job (units: rows): row r of dst (rows.nx doubles) <- NaN | row r of the staged rectangle | NaN;
        // src == nullptr: the columns [xa, xb) are left alone (a DMA writes them) or do not exist
        Rows rows;
    };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<Job>> queue;  // jobs that may still have parts to hand out
    bool stop = false;
    // A call feeds the pool a job every few hundred microseconds (a chunk's collection, a staged piece's copy-out): a
    // thread that went to sleep on the condition variable between two of them comes back 30-60 us late, sixteen of them
    // one after the other. So a thread with nothing to do keeps looking for `spin_us` before it sleeps.
    std::atomic<unsigned> posted{0};  // bumped by every post()
    int spin_us = 150;
    template <typename Pred>
    void spin_until(Pred &&ready) const
    {
        if (spin_us <= 0) return;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
        for (int i = 0; !ready(); i++) {
            _mm_pause();
            if ((i & 63) == 63 && std::chrono::steady_clock::now() >= deadline) return;
        }
    }

    void run_part(const Job &j, size_t i)
    {
        const size_t a = i * j.part;
        const size_t n = std::min(j.part, j.total - a);
        if (j.rows.n) {
            const Rows &rw = j.rows;
            const uint64_t nan_bits = 0x7ff8000000000000ull;  // the quiet NaN the kernels store
            const size_t w = rw.xb - rw.xa;
            for (size_t r = a; r < a + n; r++) {
                uint64_t *row = (uint64_t *)(j.dst + r * rw.nx * sizeof(double));
                for (size_t x = 0; x < rw.xa; x++) row[x] = nan_bits;
                if (j.src && w) std::memcpy(row + rw.xa, j.src + r * w * sizeof(double), w * sizeof(double));
                for (size_t x = rw.xb; x < rw.nx; x++) row[x] = nan_bits;
            }
            return;
        }
        if (!j.list) {
            std::memcpy(j.dst + a, j.src + a, n);
            return;
        }
        // Scattered small reads: the hardware prefetchers see no stream, so the cache line of the
        // block `ahead` rows on is requested by hand (once per line); the table itself is written
        // around the caches - it is read next by the DMA engine, not by this core.
        const int sh = j.shift;
        const size_t kB = (size_t)1 << sh;
        const size_t ahead = PM_GATHER_AHEAD_BYTES >> sh;
        size_t plane = a / j.nlist, row = a % j.nlist;
        const char *src = j.src + plane * j.plane_bytes;
        char *dst = j.dst + a * kB;
        size_t last_line = ~(size_t)0;
        for (size_t q = 0; q < n; q++, dst += kB) {
            if (row + ahead < j.nlist) {
                const size_t o = (size_t)j.list[row + ahead] << sh;
                for (size_t l = o; l < o + kB; l += 64)
                    if ((l >> 6) != last_line) {
                        _mm_prefetch(src + l, PM_GATHER_HINT);
                        last_line = l >> 6;
                    }
            }
            const char *from = src + ((size_t)j.list[row] << sh);
            for (size_t l = 0; l < kB; l += 16)
                _mm_stream_si128((__m128i *)(dst + l), _mm_loadu_si128((const __m128i *)(from + l)));
            if (++row == j.nlist) {
                row = 0;
                src += j.plane_bytes;
            }
        }
        _mm_sfence();
    }
    void finish_part(const std::shared_ptr<Job> &j)
    {
        if (j->finished.fetch_add(1, std::memory_order_acq_rel) + 1 != j->nparts) return;
        if (j->slot >= 0 || j->counted) release_slot(j->slot, 0);
        {
            std::lock_guard<std::mutex> lk(mu);
            j->done = true;
        }
        cv_done.notify_all();
    }
    // parts of `j` until none is left to hand out
    void take_parts(const std::shared_ptr<Job> &j)
    {
        for (;;) {
            const size_t i = j->next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j->nparts) return;
            run_part(*j, i);
            finish_part(j);
        }
    }
    void worker_main()
    {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                bool spun = false;
                for (;;) {
                    while (!queue.empty() && queue.front()->next.load(std::memory_order_relaxed) >= queue.front()->nparts)
                        queue.pop_front();
                    if (!queue.empty()) {
                        j = queue.front();
                        break;
                    }
                    // (only with nothing left to hand out: a queued copy-out owns a staging slot that
                    //  nobody else would ever release)
                    if (stop) return;
                    if (!spun) {
                        // look a little longer before sleeping (the lock is not held meanwhile)
                        const unsigned seen = posted.load(std::memory_order_acquire);
                        lk.unlock();
                        spin_until([&] { return posted.load(std::memory_order_acquire) != seen; });
                        lk.lock();
                        spun = true;
                        continue;
                    }
                    cv_work.wait(lk);
                    spun = false;
                }
            }
            take_parts(j);
        }
    }
    void start_workers(int threads)
    {
        const int want = std::max(0, threads - 1);  // the calling thread works too
        if ((int)workers.size() == want) return;
        stop_workers();
        stop = false;
        for (int i = 0; i < want; i++) workers.emplace_back([this] { worker_main(); });
    }
    void stop_workers()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
        workers.clear();
    }
    // (a pool of one has nobody to take a posted job: its poster does all the parts itself)
    void post(const std::shared_ptr<Job> &j)
    {
        if (workers.empty()) {
            take_parts(j);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(j);
            posted.fetch_add(1, std::memory_order_release);
        }
        cv_work.notify_all();
    }
    // post, work on it, return when it is complete
    void run(const std::shared_ptr<Job> &j)
    {
        post(j);
        take_parts(j);
        // (the last parts are in other threads' hands for a few microseconds more)
        spin_until([&] { return j->finished.load(std::memory_order_acquire) == j->nparts; });
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return j->done; });
    }
    std::shared_ptr<Job> copy_job(char *dst, const char *src, size_t bytes)
    {
        auto j = std::make_shared<Job>();
        j->dst = dst;
        j->src = src;
        j->total = bytes;
        const size_t t = workers.size() + 1;
        size_t part = (bytes + 2 * t - 1) / (2 * t);
        part = std::max<size_t>(part, (size_t)1 << 20);
        j->part = (part + 4095) & ~(size_t)4095;
        j->nparts = (bytes + j->part - 1) / j->part;
        return j;
    }
    // dst <- src with every pool thread (and the caller) taking 1 MiB+ parts
    void copy(char *dst, const char *src, size_t bytes)
    {
        if (bytes < ((size_t)1 << 20) || workers.empty()) {
            std::memcpy(dst, src, bytes);
            return;
        }
        run(copy_job(dst, src, bytes));
    }
    // the same for a staged piece, without waiting: completion frees staging slot `slot`
    void copy_async(char *dst, const char *src, size_t bytes, int slot)
    {
        if (bytes < ((size_t)1 << 20) || workers.empty()) {
            std::memcpy(dst, src, bytes);
            release_slot(slot, 0);
            return;
        }
        auto j = copy_job(dst, src, bytes);
        j->slot = slot;
        post(j);
    }
    // rows of a frame plane, asynchronously: NaN outside [xa, xb), the staged rectangle (or nothing) inside.
    // slot >= 0: the staging slot to free on completion; otherwise the job is counted in pieces_out here.
    void rows_async(char *dst, const char *src, const Rows &rw, int slot)
    {
        auto j = std::make_shared<Job>();
        j->dst = dst;
        j->src = src;
        j->rows = rw;
        j->total = rw.n;
        const size_t t = workers.size() + 1;
        const size_t row_bytes = std::max<size_t>(rw.nx * sizeof(double), 1);
        j->part = std::max<size_t>(std::max<size_t>((rw.n + 2 * t - 1) / (2 * t), ((size_t)512 << 10) / row_bytes), 1);
        j->nparts = (rw.n + j->part - 1) / j->part;
        j->slot = slot;
        if (slot < 0) {
            j->counted = true;
            std::lock_guard<std::mutex> lk(rmu);
            pieces_out++;
        }
        post(j);
    }
    // dst[plane][row] <- the (1 << shift)-byte block list[row] of source plane `plane`, for n_planes planes
    // plane_bytes apart: the rows of a block table (pm::BlockTable), collected by the pool
    void gather(char *dst, const char *src, size_t plane_bytes, size_t n_planes, const int *list, size_t n_list, int shift)
    {
        auto j = std::make_shared<Job>();
        j->dst = dst;
        j->src = src;
        j->total = n_planes * n_list;
        j->list = list;
        j->nlist = n_list;
        j->plane_bytes = plane_bytes;
        j->shift = shift;
        const size_t t = workers.size() + 1;
        j->part = std::max<size_t>((j->total + 4 * t - 1) / (4 * t), ((size_t)256 << 10) >> shift);
        j->nparts = (j->total + j->part - 1) / j->part;
        if (j->nparts == 0) return;
        run(j);
    }

    // ---- the staging ring
    void free_stage()
    {
        for (int i = 0; i < kSlots; i++) {
            if (stage[i] && be) be->free_pinned(stage[i]);
            stage[i] = nullptr;
        }
        stage_bytes = 0;
    }
    // kSlots pinned buffers of `want` bytes (resized between calls only: waits until nothing is out)
    bool ensure_stage(size_t want)
    {
        if (stage[0] && stage_bytes == want) return true;
        drain();
        free_stage();
        for (int i = 0; i < kSlots; i++) {
            stage[i] = (char *)be->alloc_pinned(want);
            if (!stage[i]) {
                free_stage();
                return false;
            }
        }
        stage_bytes = want;
        return true;
    }
    // the next slot of the ring, once the piece that used it last has been copied out
    int claim_slot(int *err)
    {
        const int slot = next_slot;
        next_slot = (slot + 1) % kSlots;
        std::unique_lock<std::mutex> lk(rmu);
        cv_slot.wait(lk, [&] { return !slot_busy[slot]; });
        if (rerror != 0) {
            *err = rerror;
            return -1;
        }
        slot_busy[slot] = true;
        pieces_out++;
        return slot;
    }
    void enqueue_piece(const Piece &pc)
    {
        {
            std::lock_guard<std::mutex> lk(rmu);
            inflight.push_back(pc);
        }
        cv_piece.notify_one();
    }
    // dst_host (pageable) <- src_dev: pieces through the staging ring; at most kSlots pieces are out at a
    // time (DMA in flight or being copied out by the retire thread and the pool). finish() completes them.
    int issue(void *dst_host, const void *src_dev, size_t bytes, void *stream)
    {
        for (size_t off = 0; off < bytes; off += stage_bytes) {
            const size_t n = std::min(stage_bytes, bytes - off);
            int err = 0;
            const int slot = claim_slot(&err);
            if (slot < 0) return err;
            int e = be->copy_d2h(stage[slot], (const char *)src_dev + off, n, stream);
            if (e == 0) e = be->record(slot, stream);
            if (e != 0) {
                release_slot(slot, 0);  // nothing will ever retire it
                return e;
            }
            enqueue_piece({slot, (char *)dst_host + off, n, Rows{}});
        }
        return 0;
    }
    // A frame plane that is NaN outside the circle (x - x0)^2 + (y - y0)^2 <= r2 (the image kernels' radius
    // pre-mask, y counted from `y_first` for the plane's first row): only bands of rows around that circle
    // cross the link, as rectangles; the copy threads write the NaN of everything else. The spans are
    // taken a pixel wider than the circle, so no rounding of this arithmetic decides a pixel.
    // `pinned`: dst_host is page-locked - the rectangles are copied in place, the pool writes the NaN around.
    int issue_disc(double *dst_host, const double *src_dev, size_t nx, size_t n_rows, double y_first, double x0, double y0, double r2,
                   void *stream, bool pinned)
    {
        const size_t band = std::max<size_t>(1, std::min<size_t>(256, stage_bytes ? stage_bytes / (nx * sizeof(double)) : 256));
        const double rr = std::sqrt(std::fmax(r2, 0.0));
        for (size_t r0 = 0; r0 < n_rows; r0 += band) {
            const size_t nr = std::min(band, n_rows - r0);
            // widest span of the circle over the rows of the band (dy closest to 0), one pixel of margin
            const double ya = y_first + (double)r0 - y0, yb = y_first + (double)(r0 + nr - 1) - y0;
            const double dy = (ya <= 0.0 && yb >= 0.0) ? 0.0 : std::fmin(std::fabs(ya), std::fabs(yb));
            Rows rw;
            rw.nx = nx;
            rw.n = nr;
            char *dst = (char *)(dst_host + r0 * nx);
            const double reach = dy - 0.5 <= rr ? std::sqrt(std::fmax(rr * rr - std::fmax(dy - 0.5, 0.0) * std::fmax(dy - 0.5, 0.0), 0.0)) : -1.0;
            const double fa = std::floor(x0 - reach) - 1.0, fb = std::ceil(x0 + reach) + 2.0;
            if (reach < 0.0 || fb <= 0.0 || fa >= (double)nx) {
                rw.xa = rw.xb = 0;  // nothing of the circle in these rows: all NaN
                rows_async(dst, nullptr, rw, -1);
                continue;
            }
            rw.xa = (size_t)std::fmax(fa, 0.0);
            rw.xb = (size_t)std::fmin(fb, (double)nx);
            const size_t w = rw.xb - rw.xa;
            const double *src = src_dev + r0 * nx + rw.xa;
            if (pinned) {
                // the DMA writes the rectangle in place, the pool the NaN around it
                const int e = be->copy_d2h_2d(dst_host + r0 * nx + rw.xa, nx * sizeof(double), src, nx * sizeof(double), w * sizeof(double), nr, stream);
                if (e != 0) return e;
                rows_async(dst, nullptr, rw, -1);
                continue;
            }
            int err = 0;
            const int slot = claim_slot(&err);
            if (slot < 0) return err;
            int e = be->copy_d2h_2d(stage[slot], w * sizeof(double), src, nx * sizeof(double), w * sizeof(double), nr, stream);
            if (e == 0) e = be->record(slot, stream);
            if (e != 0) {
                release_slot(slot, 0);
                return e;
            }
            enqueue_piece({slot, dst, w * nr * sizeof(double), rw});
        }
        return 0;
    }
    // every issued piece copied out; returns (and clears) the first backend error the retire thread met
    int finish()
    {
        std::unique_lock<std::mutex> lk(rmu);
        cv_slot.wait(lk, [&] { return pieces_out == 0; });
        const int e = rerror;
        rerror = 0;
        return e;
    }
    // the same after an error in the middle of a call: whatever was issued is waited for, errors are dropped
    void drain() { (void)finish(); }
};

}  // namespace pmh
