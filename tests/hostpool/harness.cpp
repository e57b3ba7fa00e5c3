// harness.cpp -- planetmapper_amd/csrc/pm_hostpool.h (the copy-thread pool, the pinned staging ring and its retire
// thread: the CPU side of the host <-> HBM leg of libplanetmapper_hip.so) driven WITHOUT a GPU, for the sanitizers.
//
// TEST INFRASTRUCTURE. A thread plays the DMA engine behind pmh::CopyBackend: "device" buffers are ordinary memory,
// copies are executed in stream order after random delays, completion markers are counters. The scenarios replay
// what tests/soak_hostpath.py does on the GPU box - 1 ... 16 copy threads, staging buffers of 0.25 ... 16 MiB,
// plane copies of every size, disc-plane span copies into pageable and "pinned" destinations, block-table gathers,
// backend failures in the middle of a call, thread-count and staging-size changes between calls - and check every
// destination byte. Built with -fsanitize=thread and with -fsanitize=address,undefined by
// tests/test_hostpool_sanitizers.py.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "pm_hostpool.h"

namespace {

struct MockBackend : pmh::CopyBackend {
    struct Op {
        int kind;  // 0: copy, 1: 2-D copy, 2: marker
        void *dst;
        const void *src;
        size_t a, b, c, d;  // bytes | dpitch, spitch, width, height
        int slot;
    };
    std::thread dma;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Op> q;
    bool stop = false;
    std::atomic<long> recorded[pmh::HostPool::kSlots], completed[pmh::HostPool::kSlots];
    std::mutex em;
    std::condition_variable ecv;
    std::atomic<long> copies{0};
    long fail_at = -1;  // the copy with this ordinal fails (once)
    std::atomic<int> thread_inits{0};
    const unsigned delay_seed;

    explicit MockBackend(unsigned seed) : delay_seed(seed)
    {
        for (auto &r : recorded) r = 0;
        for (auto &c : completed) c = 0;
        dma = std::thread([this] { run(); });
    }
    ~MockBackend() override
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        dma.join();
    }
    void run()
    {
        std::minstd_rand rng(delay_seed);
        for (;;) {
            Op op;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                op = q.front();
                q.pop_front();
            }
            if (op.kind != 2 && rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
            if (op.kind == 0) {
                std::memcpy(op.dst, op.src, op.a);
            } else if (op.kind == 1) {
                for (size_t r = 0; r < op.d; r++) std::memcpy((char *)op.dst + r * op.a, (const char *)op.src + r * op.b, op.c);
            } else {
                {
                    std::lock_guard<std::mutex> lk(em);
                    completed[op.slot]++;
                }
                ecv.notify_all();
            }
        }
    }
    int push(const Op &op)
    {
        if (op.kind != 2 && copies.fetch_add(1) == fail_at) return 719;  // (an error code of the backend's own)
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(op);
        }
        cv.notify_one();
        return 0;
    }
    void thread_init() override { thread_inits++; }
    int copy_d2h(void *dst, const void *src, size_t bytes, void *) override { return push({0, dst, src, bytes, 0, 0, 0, -1}); }
    int copy_d2h_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, void *) override
    {
        return push({1, dst, src, dpitch, spitch, width, height, -1});
    }
    int record(int slot, void *) override
    {
        recorded[slot]++;
        return push({2, nullptr, nullptr, 0, 0, 0, 0, slot});
    }
    int wait(int slot) override
    {
        std::unique_lock<std::mutex> lk(em);
        ecv.wait(lk, [&] { return completed[slot].load() >= recorded[slot].load(); });
        return 0;
    }
    // (everything enqueued so far has been executed: what hipStreamSynchronize is to the library)
    void sync()
    {
        record(0, nullptr);
        wait(0);
    }
    void *alloc_pinned(size_t bytes) override { return std::aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095); }
    void free_pinned(void *p) override { std::free(p); }
};

long checks = 0;
#define REQUIRE(cond)                                                                  \
    do {                                                                               \
        checks++;                                                                      \
        if (!(cond)) {                                                                 \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
            std::exit(3);                                                              \
        }                                                                              \
    } while (0)

std::vector<double> random_plane(std::mt19937_64 &rng, size_t n)
{
    std::vector<double> v(n);
    for (auto &x : v) x = (double)(rng() >> 11) * 0x1.0p-53 - 0.5;
    return v;
}

void scenario(std::mt19937_64 &rng, int threads, size_t stage_bytes, bool quick)
{
    MockBackend be((unsigned)rng());
    {
        pmh::HostPool pool;
        pool.be = &be;
        pool.start_workers(threads);
        pool.start_retirer();
        REQUIRE(pool.ensure_stage(stage_bytes));
        auto pick = [&](size_t lo, size_t hi) { return lo + (size_t)(rng() % (hi - lo + 1)); };
        // ---- (a) plane copies of every size into pageable destinations, several in flight, one finish
        for (int round = 0; round < (quick ? 2 : 4); round++) {
            const int n = (int)pick(1, 5);
            std::vector<std::vector<double>> src, dst;
            for (int i = 0; i < n; i++) {
                const size_t elems = pick(40000, quick ? 600000 : 3000000);
                src.push_back(random_plane(rng, elems));
                dst.emplace_back(elems, -1.0);
            }
            for (int i = 0; i < n; i++) REQUIRE(pool.issue(dst[i].data(), src[i].data(), src[i].size() * 8, nullptr) == 0);
            REQUIRE(pool.finish() == 0);
            for (int i = 0; i < n; i++) REQUIRE(std::memcmp(dst[i].data(), src[i].data(), src[i].size() * 8) == 0);
        }
        // ---- (b) planes that are NaN outside a circle: only spans around it travel, the pool writes the NaN
        for (int round = 0; round < (quick ? 3 : 8); round++) {
            const size_t nx = pick(17, quick ? 700 : 2300), ny = pick(5, quick ? 500 : 1500);
            const double x0 = (double)nx * ((double)(rng() % 1400) / 1000.0 - 0.2), y0 = (double)ny * ((double)(rng() % 1400) / 1000.0 - 0.2);
            const double r = (double)std::min(nx, ny) * (double)(rng() % 900 + 1) / 1000.0, r2 = r * r;
            const size_t row_begin = rng() % 3 == 0 ? pick(0, ny / 2) : 0, n_rows = ny - row_begin;
            std::vector<double> src(nx * n_rows), dst(nx * n_rows, 42.0);
            const double nan = std::nan("");
            for (size_t yy = 0; yy < n_rows; yy++)
                for (size_t xx = 0; xx < nx; xx++) {
                    const double dx = (double)xx - x0, dy = (double)(yy + row_begin) - y0;
                    src[yy * nx + xx] = (dx * dx + dy * dy) > r2 ? nan : (double)(rng() >> 12);
                }
            const bool pinned = rng() % 2;
            REQUIRE(pool.issue_disc(dst.data(), src.data(), nx, n_rows, (double)row_begin, x0, y0, r2, nullptr, pinned) == 0);
            REQUIRE(pool.finish() == 0);
            be.sync();  // (a pinned destination's rectangles: complete with the stream, like the library's d2h_finish)
            REQUIRE(std::memcmp(dst.data(), src.data(), src.size() * 8) == 0);
        }
        // ---- (c) the pool's own parallel copy and the block-table gather
        {
            const size_t elems = pick(200000, quick ? 800000 : 4000000);
            auto src = random_plane(rng, elems);
            std::vector<double> dst(elems);
            pool.copy((char *)dst.data(), (const char *)src.data(), elems * 8);
            REQUIRE(dst == src);
            for (int shift : {4, 8}) {
                const size_t plane_bytes = ((size_t)1 << shift) * pick(2000, quick ? 20000 : 60000), n_planes = pick(1, 7);
                const size_t n_blk = plane_bytes >> shift;
                std::vector<char> cube(plane_bytes * n_planes);
                for (auto &c : cube) c = (char)rng();
                std::vector<int> list;
                for (size_t b = 0; b < n_blk; b++)
                    if (rng() % 7 == 0) list.push_back((int)b);
                if (list.empty()) list.push_back(0);
                std::vector<char> table((list.size() * n_planes) << shift, 0);
                pool.gather(table.data(), cube.data(), plane_bytes, n_planes, list.data(), list.size(), shift);
                for (size_t p = 0; p < n_planes; p++)
                    for (size_t k = 0; k < list.size(); k += 1 + list.size() / 97)
                        REQUIRE(std::memcmp(&table[(p * list.size() + k) << shift], &cube[p * plane_bytes + ((size_t)list[k] << shift)],
                                            (size_t)1 << shift) == 0);
            }
        }
        // ---- (d) the backend fails in the middle of a call: the error comes back, the pool drains, the next call works
        {
            const size_t elems = stage_bytes / 8 * 5 + 1234;
            auto src = random_plane(rng, elems);
            std::vector<double> dst(elems, 0.0);
            be.fail_at = be.copies.load() + 2;
            REQUIRE(pool.issue(dst.data(), src.data(), elems * 8, nullptr) == 719);
            pool.drain();
            be.fail_at = -1;
            REQUIRE(pool.issue(dst.data(), src.data(), elems * 8, nullptr) == 0);
            REQUIRE(pool.finish() == 0);
            REQUIRE(dst == src);
        }
        // ---- (e) another thread count and staging size between calls, then again
        pool.start_workers(threads == 1 ? 4 : 1);
        REQUIRE(pool.ensure_stage(stage_bytes / 2 + 4096));
        {
            auto src = random_plane(rng, 300000);
            std::vector<double> dst(src.size());
            REQUIRE(pool.issue(dst.data(), src.data(), src.size() * 8, nullptr) == 0);
            REQUIRE(pool.finish() == 0);
            REQUIRE(dst == src);
        }
        REQUIRE(be.thread_inits.load() == 1);
        pool.stop_retirer();
        pool.stop_workers();
        pool.free_stage();
        pool.be = nullptr;
    }
}

}  // namespace

int main(int argc, char **argv)
{
    const unsigned long seed = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1;
    const bool quick = argc > 2 && std::atoi(argv[2]) != 0;
    std::mt19937_64 rng(seed);
    const int threads[] = {1, 2, 3, 8, 16};
    const size_t stages[] = {(size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20};
    int n = 0;
    for (int t : threads)
        for (size_t st : stages) {
            if (quick && (n++ % 3) != 0) continue;  // (a third of the grid per quick run; the seed moves which)
            scenario(rng, t, st, quick);
        }
    std::printf("hostpool harness: seed %lu, %ld checks passed\n", seed, checks);
    return 0;
}
