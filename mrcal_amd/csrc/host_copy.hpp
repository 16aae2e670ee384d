// Device -> pageable host memory at PCIe speed (round 6).
//
// optimizer_callback() hands the caller 466 MB of CSR Jacobian at the metric's size, into arrays the caller has just
// allocated: untouched pages. hipMemcpyAsync() to pageable memory goes through the runtime's own staging buffer, one
// thread copying out of it and faulting the destination's pages in as it goes: 10-13 GB/s, 35-50 ms.
// Here: a ring of pinned chunks (allocated once per process), the DMA of chunk i + 1 .. i + 3 in flight while a small
// pool of threads copies chunk i into the destination - each thread its own slice, so the page faults run in parallel
// too. What bounds it is the link (~55 GB/s) or the host's memory, whichever is slower
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>

namespace mrcal_amd {

class HostCopyPool
{
public:
    static HostCopyPool& get() { static HostCopyPool pool; return pool; }

    // dst (pageable) <- src (device), on `stream` (synchronized on return). false: fall back to hipMemcpy
    bool copy(void* dst, const void* src, size_t bytes, hipStream_t stream)
    {
        std::lock_guard<std::mutex> one_at_a_time(api_mutex);
        if(!ready && !init()) return false;
        char* d = (char*)dst; const char* s = (const char*)src;
        const size_t nchunks = (bytes + CHUNK - 1)/CHUNK;
        for(size_t i = 0; i < nchunks + NSLOT - 1; i++)
        {
            if(i < nchunks)
            {
                const size_t off = i*CHUNK, n = (bytes - off < CHUNK) ? bytes - off : CHUNK;
                if(hipMemcpyAsync(slot[i % NSLOT], s + off, n, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
                if(hipEventRecord(ev[i % NSLOT], stream) != hipSuccess) return false;
            }
            if(i + 1 >= NSLOT)
            {
                // chunk j has had NSLOT - 1 chunks' worth of time on the link: take it out (its slot is the next to be refilled)
                const size_t j = i + 1 - NSLOT, off = j*CHUNK, n = (bytes - off < CHUNK) ? bytes - off : CHUNK;
                if(hipEventSynchronize(ev[j % NSLOT]) != hipSuccess) return false;
                run(d + off, slot[j % NSLOT], n);
            }
        }
        return hipStreamSynchronize(stream) == hipSuccess;
    }

private:
    static constexpr size_t CHUNK = (size_t)8 << 20;
    static constexpr int    NSLOT = 4;
    char*      slot[NSLOT] = {NULL, NULL, NULL, NULL};
    hipEvent_t ev[NSLOT];
    bool       ready = false, failed = false;
    std::mutex api_mutex;

    // the pool: a job is (dst, src, n) cut into nthreads slices; generation counts the jobs
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    char* job_dst = NULL; const char* job_src = NULL; size_t job_n = 0;
    unsigned long generation = 0;
    int remaining = 0;
    bool quit = false;

    bool init()
    {
        if(failed) return false;
        for(int i = 0; i < NSLOT; i++)
        {
            if(hipHostMalloc((void**)&slot[i], CHUNK, hipHostMallocDefault) != hipSuccess ||
               hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { failed = true; return false; }
        }
        unsigned hw = std::thread::hardware_concurrency();
        int nt = (int)(hw >= 32 ? 16 : (hw >= 8 ? hw/2 : 2));
        for(int t = 0; t < nt; t++) threads.emplace_back([this, t, nt] { worker(t, nt); });
        ready = true;
        return true;
    }
    void worker(int t, int nt)
    {
        unsigned long seen = 0;
        for(;;)
        {
            char* d; const char* s; size_t n;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&] { return quit || generation != seen; });
                if(quit) return;
                seen = generation; d = job_dst; s = job_src; n = job_n;
            }
            // slices on 4 KB boundaries: a page is faulted in by one thread
            const size_t per = (((n + nt - 1)/nt) + 4095) & ~(size_t)4095;
            const size_t lo = (size_t)t*per, hi = lo + per < n ? lo + per : n;
            if(lo < n) memcpy(d + lo, s + lo, hi - lo);
            {
                std::lock_guard<std::mutex> lk(m);
                if(--remaining == 0) cv_done.notify_one();
            }
        }
    }
    void run(char* d, const char* s, size_t n)
    {
        std::unique_lock<std::mutex> lk(m);
        job_dst = d; job_src = s; job_n = n; remaining = (int)threads.size(); generation++;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return remaining == 0; });
    }
    HostCopyPool() {}
    ~HostCopyPool()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for(auto& th : threads) if(th.joinable()) th.join();
        // (the pinned chunks and events go with the process: the HIP runtime may be gone by now)
    }
};

// bytes from the device into pageable host memory: the pipeline above for what is big, hipMemcpyAsync otherwise
inline bool device_to_host(void* dst, const void* src, size_t bytes, hipStream_t stream)
{
    if(bytes >= ((size_t)8 << 20) && HostCopyPool::get().copy(dst, src, bytes, stream)) return true;
    if(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
    return hipStreamSynchronize(stream) == hipSuccess;
}

} // namespace mrcal_amd
