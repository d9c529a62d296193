// Device-memory pool of the batch plans.  A caller of a ragged workload meets a new molecule-size vector on every call
// (the reference's evaluation protocol generates for 1 000 different references, research_scripts/evaluation.py:98-103), i.e.
// a new plan per call and - with a bounded plan cache - a destroyed one per call.  hipMalloc / hipFree cost 0.1-1 ms each
// and hipFree synchronises the device; a plan used to make ~20 of each.  Plans now take TWO blocks (tables, workspace) from
// this pool: freed blocks are kept per device in size classes (1/8-octave rounding; a request takes the smallest cached block within 1.5x of it) and handed out again,
// so after the first few calls a new plan allocates nothing and device memory stays flat.  288 GB of HBM per GPU: the cap on
// cached bytes (4 GiB per device) is about what 20 plans of the 256-molecule ragged workload hold.
#include "mcg_egnn_internal.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace {

constexpr size_t kCacheCap = 4ull << 30;      // cached (free) bytes per device before blocks go back to the driver

struct DevPool {
    std::multimap<size_t, void*> free_blocks;             // size class -> block
    std::unordered_map<void*, size_t> live;               // block -> size class
    size_t cached = 0, in_use = 0;
    int64_t n_driver_allocs = 0, n_pool_hits = 0;
};

std::mutex g_mu;
DevPool g_pools[64];

size_t size_class(size_t bytes) {
    if (bytes < 4096) return 4096;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);            // 8 classes per octave
    return (bytes + step - 1) / step * step;
}

int cur_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev >= 0 && dev < 64 ? dev : 0;
}

}  // namespace

int mcg_dev_alloc(size_t bytes, void** out) {
    *out = nullptr;
    const size_t cls = size_class(bytes ? bytes : 1);
    const int dev = cur_dev();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DevPool& P = g_pools[dev];
        // best fit: the smallest cached block that holds the request, as long as it is not more than 1.5x too large
        // (ragged size vectors of one workload differ by a few per cent: their blocks serve one another)
        auto it = P.free_blocks.lower_bound(cls);
        if (it != P.free_blocks.end() && it->first <= cls + cls / 2) {
            const size_t got = it->first;
            *out = it->second;
            P.free_blocks.erase(it);
            P.cached -= got; P.in_use += got; ++P.n_pool_hits;
            P.live[*out] = got;
            return MCG_OK;
        }
    }
    void* p = nullptr;
    if (hipMalloc(&p, cls) != hipSuccess) {
        (void)hipGetLastError();
        mcg_dev_trim();                                    // give the cached blocks back and try once more
        if (hipMalloc(&p, cls) != hipSuccess) {
            (void)hipGetLastError();
            mcg_set_error("out of device memory (%zu bytes)", cls);
            return MCG_ERR_HIP;
        }
    }
    std::lock_guard<std::mutex> lk(g_mu);
    DevPool& P = g_pools[dev];
    P.live[p] = cls; P.in_use += cls; ++P.n_driver_allocs;
    *out = p;
    return MCG_OK;
}

// The caller guarantees that no kernel still uses the block (mcg_plan_destroy waits for the plan's completion event on the
// plan's own device first - what the hipFree it replaces did implicitly).
void mcg_dev_free(void* p) {
    if (!p) return;
    const int dev = cur_dev();
    bool release = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DevPool* P = &g_pools[dev];
        bool found = false;
        auto it = P->live.find(p);
        if (it != P->live.end()) found = true;
        else                                               // allocated while another device was current: look it up
            for (DevPool& Q : g_pools) {
                auto jt = Q.live.find(p);
                if (jt != Q.live.end()) { P = &Q; it = jt; found = true; break; }      // (iterators of ONE map only)
            }
        if (!found) release = true;                        // not one of ours: back to the driver
        else {
            const size_t cls = it->second;
            P->live.erase(it);
            P->in_use -= cls;
            if (P->cached + cls <= kCacheCap) { P->free_blocks.emplace(cls, p); P->cached += cls; }
            else release = true;
        }
    }
    if (release) (void)hipFree(p);
}

void mcg_dev_trim() {
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (DevPool& P : g_pools) {
            for (auto& kv : P.free_blocks) blocks.push_back(kv.second);
            P.free_blocks.clear();
            P.cached = 0;
        }
    }
    for (void* b : blocks) (void)hipFree(b);
}

extern "C" int mcg_pool_stats(int64_t* stats_host /*[4]*/, int trim) {
    if (trim) mcg_dev_trim();
    if (!stats_host) return MCG_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    const DevPool& P = g_pools[cur_dev()];
    stats_host[0] = (int64_t)P.in_use; stats_host[1] = (int64_t)P.cached;
    stats_host[2] = P.n_driver_allocs; stats_host[3] = P.n_pool_hits;
    return MCG_OK;
}
