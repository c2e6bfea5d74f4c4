/*
 * dpx_capi.cpp -- implementation of include/dpx_align.h on the HIP runtime (host side of libdpxalign.so).
 *
 * Owns device memory, streams and launch geometry; the arithmetic is in dpx_kernels.hip.  There is no CPU
 * fallback anywhere in this file: without a gfx950 device every entry point fails with DPX_ERR_NO_DEVICE.
 */
#include "../../include/dpx_align.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

#include "dpx_kernels.h"
#include "dpx_layout.h"

namespace {

std::mutex g_mu;
int g_device = -1;             /* default device: the one dpx_init() bound (dpx_batch_create, dpx_device_info, dpx_prim_eval) */
thread_local int t_device = -1; /* device of the call this thread is in (a batch's own device, or the default) */
thread_local std::string t_err;

int hip_fail(hipError_t e, const char *what) {
    t_err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? DPX_ERR_NOMEM : DPX_ERR_HIP;
}
#define HIP_TRY(call)                                     \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return hip_fail(e__, #call); \
    } while (0)

/* Make `device` (-1: the default device, initialising device 0 if dpx_init() was never called) current for this thread.
 * Every buffer / stream cache entry carries its device, so one process can drive several GPUs: a batch lives on the
 * device it was created on and every call on it binds that device first. */
int bind_device(int device = -1) {
    if (device < 0) {
        if (g_device < 0) {
            int rc = dpx_init(0);
            if (rc != DPX_OK) return rc;
        }
        device = g_device;
    }
    HIP_TRY(hipSetDevice(device));
    t_device = device;
    return DPX_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kSmallResultBytes = 12288; /* score / end row / end column of a small batch, as they lie in the arena (dpx_batch_output_begin) */

/* Development and test knobs.  The environment is read in ONE place, when dpx_init() binds a device and again at the start of every
 * dpx_batch_create*() (the tests flip a knob between two batches of one process); everything else reads this snapshot.  -1 / 0 = not
 * set.  None of them changes a result: they force or forbid a kernel family, change the launch shape or trace the runtime. */
struct Knobs {
    int rowsPerLane = 0;   /* DPX_R=2|4|8|16 */
    int packed = -1;       /* DPX_PACKED=0|1: the two-pairs-per-wave int16 kernels */
    int lanes = -1;        /* DPX_LANES=0|1: the lane-packed (several pairs per wave) kernels */
    int lanesPk = -1;      /* DPX_LANES_PK=0: keep the lane-packed batches on the int32 kernels */
    int split = -1;        /* DPX_SPLIT=0|1: one wave per stripe */
    int rowTags = -1;      /* DPX_ROW_TAGS=0: per-row start-cell keys in the packed SW kernel */
    int rampLines = -1;    /* DPX_RAMP_LINES=0|1: skew ramps store lines / whole chunks */
    int wavesPerBlock = 0; /* DPX_WPB=1|4 */
    int group = 0;         /* DPX_GROUP: pairs per interleaved block of the matrix layout */
    int tbWalk = -1;       /* DPX_TB_WALK=0|1|2 */
    int poolProbe = -1;    /* DPX_POOL_PROBE=0|1|2: nothing / time the pool / time + shop, whatever DPX_TUNE_PLACEMENT says */
    bool poolMalloc = false;  /* DPX_POOL=malloc: one hipMalloc instead of the chunked virtual range */
    size_t poolChunkBytes = 0; /* DPX_POOL_CHUNK_MB */
    int poolGuard = 0;     /* DPX_POOL_GUARD: 1 = pattern band behind the matrices, 2 = "selftest" (the band is born damaged) */
    bool trace = false, traceVmm = false; /* DPX_TRACE / DPX_TRACE_VMM: phase timings / pool mapping calls on stderr */
};
std::mutex g_knobsMu;
Knobs g_knobs;
void refresh_knobs() {
    auto num = [](const char *name, int unset) { const char *e = getenv(name); return e ? atoi(e) : unset; };
    Knobs k;
    k.rowsPerLane = num("DPX_R", 0);
    k.packed = num("DPX_PACKED", -1);
    k.lanes = num("DPX_LANES", -1);
    k.lanesPk = num("DPX_LANES_PK", -1);
    k.split = num("DPX_SPLIT", -1);
    k.rowTags = num("DPX_ROW_TAGS", -1);
    k.rampLines = num("DPX_RAMP_LINES", -1);
    k.wavesPerBlock = num("DPX_WPB", 0);
    k.group = num("DPX_GROUP", 0);
    k.tbWalk = num("DPX_TB_WALK", -1);
    k.poolProbe = num("DPX_POOL_PROBE", -1);
    { const char *e = getenv("DPX_POOL"); k.poolMalloc = e && !strcmp(e, "malloc"); }
    { const long v = num("DPX_POOL_CHUNK_MB", 0); k.poolChunkBytes = v > 0 ? (size_t)v << 20 : 0; }
    { const char *e = getenv("DPX_POOL_GUARD"); k.poolGuard = !e ? 0 : !strcmp(e, "selftest") ? 2 : 1; }
    k.trace = getenv("DPX_TRACE") != nullptr;
    k.traceVmm = getenv("DPX_TRACE_VMM") != nullptr;
    std::lock_guard<std::mutex> lk(g_knobsMu);
    g_knobs = k;
}
Knobs knobs() { std::lock_guard<std::mutex> lk(g_knobsMu); return g_knobs; }

/* LDS request that caps residency at 16 fill waves per CU (9 KiB per wave of the workgroup) */
constexpr size_t kLdsFloor = 36u * 1024u * (DPX_FILL_THREADS / 64) / 4;

/* Parked buffers.  The matrix pool is by far the largest allocation (22 GB for the headline batch) and hipMalloc/hipFree
 * of that size cost 50-1000 ms -- the "memory management" slice that dominated the reference's V12 profile (187 of 440 ms)
 * and that it attacked by pooling (V9) and sizing once (V14); pinning ~70 MB of host memory costs ~10 ms; and even the
 * small calls add up when the reference's main.cpp aligns one pair per call from 20 threads.  So nothing a batch
 * allocates is freed when the batch goes away: buffers are parked per kind and handed to the next batch that fits
 * (batched drivers create same-sized batches back to back; the class-per-pair drivers create thousands of tiny ones). */
/* The matrix pool is not one hipMalloc.  Round 3 (tools/poolstudy.hip, profiles/r03/poolstudy.txt): one hipMalloc of 22 GB is
 * written by hipMemset at 6.3 TB/s or at 6.05 TB/s and by a streaming-store kernel at 5.5 or 4.8 TB/s depending on the
 * allocation -- although every single GiB of a "slow" allocation is written as fast as every GiB of a "fast" one, so the
 * mode is a property of how the runtime backed and mapped the whole range, not of where it lies.  The same range built from
 * the virtual-memory API -- one hipMemAddressReserve, physical chunks from hipMemCreate, hipMemMap -- was written at 6.5 - 7.2 TB/s
 * (memset) in every trial, and the headline fill follows: 3.62 ms on one hipMalloc, 3.2 - 3.35 ms on 256-MiB chunks (eight fresh
 * pools each, alternating in one process, profiles/r03/pool_ab_variants_*.txt; 1-GiB and 2-MiB chunks: wider spread, 3.28 - 3.61).
 * So every matrix pool of every caller is a chunked virtual range of 256-MiB chunks (DPX_POOL=malloc restores hipMalloc for
 * A/B runs; DPX_POOL_CHUNK_MB sets the chunk size).  Small pools (< 64 MiB) stay on hipMalloc: the class-per-pair drivers create
 * thousands of them. */
struct VmmRange {
    size_t bytes; int device; size_t chunkBytes;
    void *reserved; size_t reservedBytes; /* what hipMemAddressReserve returned (the pool starts at the first chunk-aligned address inside) */
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks;
};
std::mutex g_vmmMu;
std::map<void *, VmmRange> g_vmmRanges;
void forget_pool_record(void *pool);

/* Allocation granularity of device memory behind the virtual-memory API, queried once per device (round 4; rounds 1-3 assumed 2 MiB):
 * every chunk size, every mapping offset and the reserved range are multiples of the RECOMMENDED granularity, and the range is reserved
 * with the chunk size as its alignment, so a chunk never straddles a boundary of its own size. */
size_t vmm_granularity(int device) {
    static std::mutex mu;
    static std::map<int, size_t> known;
    std::lock_guard<std::mutex> lk(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gmin = 0, grec = 0;
    if (hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) { (void)hipGetLastError(); gmin = 0; }
    if (hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { (void)hipGetLastError(); grec = 0; }
    size_t g = std::max<size_t>({gmin, grec, (size_t)2 << 20});
    while (g & (g - 1)) g += g & (~g + 1); /* up to a power of two (it is one on every runtime seen) */
    if (knobs().traceVmm) { fprintf(stderr, "[vmm] device %d: allocation granularity minimum %zu recommended %zu -> %zu\n", device, gmin, grec, g); fflush(stderr); }
    known[device] = g;
    return g;
}

/* Tearing a range down.  ROCm 7.2 crashed inside hipMemAddressFree (SIGSEGV in libamdhip64, profiles/r03/vmm_address_free_crash.txt) in
 * about half the runs of the pool tests.  The crashing version (commit 46f7574) already undid every mapping with exactly the (address,
 * size) it was made with, released the handles and then freed the address range -- in this order; what the fix (c07d102) added, and
 * what the crash therefore depended on, are the two waits for the device: one before the first unmap (work of ANY stream of this
 * process, e.g. a hipMemsetAsync of the previous candidate or a fill on a side stream, may still touch the range) and one between the
 * releases and hipMemAddressFree (unmap / release are queued inside the runtime).  The two were added together and never separated;
 * both stay.  One hipMemUnmap over the whole range instead of one per chunk is a different bug: the chunks behind the first stayed
 * allocated (tools/pool_leak.py).  Every status is checked: on a failure the range stays in g_vmmRanges (nothing is freed twice, the
 * leak is visible) and the error is left in dpx_last_error(). */
bool vmm_release(void *va, VmmRange &r, size_t mappedBytes) {
    const bool dbg = knobs().traceVmm;
    if (dbg) { fprintf(stderr, "[vmm] release [%p, %p) mapped %zu chunks %zu\n", va, (void *)((char *)va + r.bytes), mappedBytes, r.chunks.size()); fflush(stderr); }
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (r.device >= 0 && r.device != cur) (void)hipSetDevice(r.device); /* (one process may drive several devices: wait for the range's own) */
    hipError_t bad = hipDeviceSynchronize();
    const char *where = "hipDeviceSynchronize";
    size_t off = 0;
    for (auto &c : r.chunks) {
        if (bad == hipSuccess && off + c.second <= mappedBytes) { bad = hipMemUnmap((char *)va + off, c.second); where = "hipMemUnmap"; }
        off += c.second;
    }
    for (auto &c : r.chunks) {
        if (bad != hipSuccess) break;
        bad = hipMemRelease(c.first);
        where = "hipMemRelease";
    }
    if (bad == hipSuccess) { bad = hipDeviceSynchronize(); where = "hipDeviceSynchronize"; }
    if (bad == hipSuccess) { bad = hipMemAddressFree(r.reserved, r.reservedBytes); where = "hipMemAddressFree"; }
    if (bad != hipSuccess) {
        t_err = std::string("matrix pool teardown: ") + where + ": " + hipGetErrorString(bad);
        if (dbg) { fprintf(stderr, "[vmm] %s\n", t_err.c_str()); fflush(stderr); }
        (void)hipGetLastError();
    } else if (dbg) { fprintf(stderr, "[vmm] done\n"); fflush(stderr); }
    if (r.device >= 0 && r.device != cur && cur >= 0) (void)hipSetDevice(cur);
    return bad == hipSuccess;
}

hipError_t pool_alloc(void **out, size_t bytes) {
    const Knobs kn = knobs();
    const bool useMalloc = kn.poolMalloc;
    const size_t chunkEnv = kn.poolChunkBytes;
    if (useMalloc || bytes < ((size_t)64 << 20) || t_device < 0) return hipMalloc(out, bytes);
    const size_t gran = vmm_granularity(t_device);
    size_t chunk = std::max(chunkEnv ? chunkEnv : (size_t)256 << 20, gran);
    while (chunk & (chunk - 1)) chunk += chunk & (~chunk + 1); /* a power of two >= the granularity: it is the alignment of the reserved range */
    VmmRange r;
    r.bytes = align_up(bytes, gran);
    r.device = t_device;
    r.chunkBytes = chunk;
    /* hipMemAddressReserve of ROCm 7.2 ignores its alignment argument (ranges come back 2-MiB aligned whatever is asked for:
     * profiles/r04/vmm_granularity_and_alignment.txt), so the range is reserved one chunk longer and the pool starts at the first
     * chunk-aligned address inside it: every chunk is mapped at a multiple of its own size */
    const size_t align = chunk; /* (2 MiB, 256 MiB or 1 GiB: the fills do not care, profiles/r04/vmm_granularity_and_alignment.txt) */
    r.reservedBytes = r.bytes + align;
    r.reserved = nullptr;
    hipError_t e = hipMemAddressReserve(&r.reserved, r.reservedBytes, align, nullptr, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); return hipMalloc(out, bytes); }
    void *va = (void *)align_up((size_t)(uintptr_t)r.reserved, align);
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = t_device;
    size_t mapped = 0;
    for (size_t off = 0; off < r.bytes && e == hipSuccess; off += chunk) {
        const size_t sz = std::min(chunk, r.bytes - off); /* (the last one: a multiple of the granularity, like r.bytes) */
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, sz, &prop, 0);
        if (e != hipSuccess) break;
        r.chunks.emplace_back(h, sz);
        e = hipMemMap((char *)va + off, sz, 0, h, 0);
        if (e == hipSuccess) mapped += sz;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = t_device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, r.bytes, &acc, 1);
    }
    if (e != hipSuccess) {
        if (kn.traceVmm) { fprintf(stderr, "[vmm] building %p failed after %zu of %zu bytes mapped: %s\n", va, mapped, r.bytes, hipGetErrorString(e)); fflush(stderr); }
        (void)vmm_release(va, r, mapped);
        if (e == hipErrorOutOfMemory) return e;
        /* a runtime without (working) virtual-memory management: one hipMalloc, as before round 3 -- slower to write, never wrong */
        (void)hipGetLastError();
        return hipMalloc(out, bytes);
    }
    if (kn.traceVmm) { fprintf(stderr, "[vmm] alloc [%p, %p) chunks %zu of %zu MiB\n", va, (void *)((char *)va + r.bytes), r.chunks.size(), chunk >> 20); fflush(stderr); }
    { std::lock_guard<std::mutex> lk(g_vmmMu); g_vmmRanges.emplace(va, std::move(r)); }
    *out = va;
    return hipSuccess;
}

/* how the pool at `p` was built: chunk size of a virtual range, 0 for one hipMalloc */
size_t pool_chunk_bytes(void *p) {
    std::lock_guard<std::mutex> lk(g_vmmMu);
    auto it = g_vmmRanges.find(p);
    return it == g_vmmRanges.end() ? 0 : it->second.chunkBytes;
}

void pool_free(void *p) {
    if (!p) return;
    forget_pool_record(p); /* (a later pool at the same address must not inherit this one's timings) */
    VmmRange r;
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_vmmMu);
        auto it = g_vmmRanges.find(p);
        if (it != g_vmmRanges.end()) { r = std::move(it->second); g_vmmRanges.erase(it); found = true; }
    }
    if (!found) { (void)hipFree(p); return; }
    if (!vmm_release(p, r, r.bytes)) { std::lock_guard<std::mutex> lk(g_vmmMu); g_vmmRanges.emplace(p, std::move(r)); } /* kept: see vmm_release */
}

class BufCache {
public:
    enum Kind { Device, PinnedHost, DevicePool };
    BufCache(Kind kind, size_t maxEntries, size_t maxBytes) : kind_(kind), maxEntries_(maxEntries), maxBytes_(maxBytes) {}

    /* a parked buffer of [need, 1.5 * need + 1 MiB], else a fresh allocation (after an out-of-memory: drop everything
     * that is parked anywhere and retry once) */
    hipError_t take(void **out, size_t need, size_t *actual, bool *fresh = nullptr);
    void discard(void *ptr) { raw_free(ptr); } /* a buffer that will not be parked */
    void park(void *ptr, size_t bytes);
    void drain();

private:
    struct Entry { void *ptr; size_t bytes; int device; };
    hipError_t raw_alloc(void **out, size_t bytes) const {
        return kind_ == Device ? hipMalloc(out, bytes) : kind_ == DevicePool ? pool_alloc(out, bytes) : hipHostMalloc(out, bytes, hipHostMallocDefault);
    }
    void raw_free(void *p) const {
        if (!p) return;
        if (kind_ == DevicePool) pool_free(p);
        else (void)(kind_ == Device ? hipFree(p) : hipHostFree(p));
    }
    const Kind kind_;
    const size_t maxEntries_, maxBytes_;
    std::mutex mu_;
    std::vector<Entry> parked_; /* oldest first */
    size_t total_ = 0;
};

void trim_all_caches();

hipError_t BufCache::take(void **out, size_t need, size_t *actual, bool *fresh) {
    if (fresh) *fresh = false;
    {
        std::lock_guard<std::mutex> lk(mu_);
        size_t best = parked_.size();
        for (size_t i = 0; i < parked_.size(); i++) {
            const Entry &e = parked_[i];
            /* (a parked matrix pool of up to 13 GiB serves any smaller batch: a driver that cuts a file into batches of one pool
             * budget ends with a short batch, and a pool reserved ahead of time -- dpx_pool_reserve -- is sized by the budget) */
            /* (a parked pinned buffer of up to 32 MiB -- dpx_text_reserve -- serves any smaller text) */
            /* (... but not a tiny one: the class-per-pair drivers create thousands of batches of a few MiB, which stay on hipMalloc) */
            const size_t roof = (kind_ == DevicePool && need >= ((size_t)64 << 20)) ? std::max(need + need / 2 + (1u << 20), (size_t)13 << 30)
                                : (kind_ == PinnedHost && need >= ((size_t)2 << 20)) ? std::max(need + need / 2 + (1u << 20), (size_t)32 << 20) : need + need / 2 + (1u << 20);
            if (e.device != t_device || e.bytes < need || e.bytes > roof) continue;
            if (best == parked_.size() || e.bytes < parked_[best].bytes) best = i;
        }
        if (best != parked_.size()) {
            *out = parked_[best].ptr;
            *actual = parked_[best].bytes;
            total_ -= parked_[best].bytes;
            parked_.erase(parked_.begin() + (long)best);
            return hipSuccess;
        }
    }
    /* a fresh buffer gets 1/8 of headroom (and a 64 KiB granule) so that the next batch of about the same size can take
     * it over: batches of one file differ by a few per cent, and pinning 8 MB costs ~1 ms every time it misses */
    const size_t want = need < ((size_t)1 << 30) ? align_up(need + need / 8, (size_t)64 << 10) : need;
    *actual = want;
    if (fresh) *fresh = true;
    hipError_t e = raw_alloc(out, want);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        trim_all_caches();
        *actual = need;
        e = raw_alloc(out, need);
    }
    return e;
}

void BufCache::park(void *ptr, size_t bytes) {
    if (!ptr) return;
    std::vector<void *> evicted;
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (t_device < 0 || bytes > maxBytes_) {
            evicted.push_back(ptr);
        } else {
            /* per device: at most one parked buffer of 16 GiB or more, at most eight of a GiB or more and 48 GiB of them (a pipelined driver
             * keeps up to eight batches in flight on pools of a few GiB each -- dpx_main -inflight; tens of GB twice over is not worth holding) */
            const size_t big = (size_t)1 << 30, huge = (size_t)16 << 30;
            size_t bigOnes = 0, bigBytes = bytes;
            for (size_t i = parked_.size(); i-- > 0;) { /* newest first: the oldest go */
                if (parked_[i].device != t_device || parked_[i].bytes < big || bytes < big) continue;
                bigBytes += parked_[i].bytes;
                const bool drop = bytes >= huge || parked_[i].bytes >= huge || ++bigOnes >= 8 || bigBytes > ((size_t)48 << 30);
                if (drop) { evicted.push_back(parked_[i].ptr); total_ -= parked_[i].bytes; parked_.erase(parked_.begin() + (long)i); }
            }
            while (!parked_.empty() && (parked_.size() >= maxEntries_ || total_ + bytes > maxBytes_)) {
                evicted.push_back(parked_.front().ptr);
                total_ -= parked_.front().bytes;
                parked_.erase(parked_.begin());
            }
            parked_.push_back(Entry{ptr, bytes, t_device});
            total_ += bytes;
        }
    }
    for (void *p : evicted) raw_free(p);
}

void BufCache::drain() {
    std::vector<Entry> gone;
    {
        std::lock_guard<std::mutex> lk(mu_);
        gone.swap(parked_);
        total_ = 0;
    }
    for (const Entry &e : gone) raw_free(e.ptr);
}

/* what a batch allocates: the int16 matrices, the arena of small arrays, the traceback line buffers (device + pinned) */
BufCache g_matCache(BufCache::DevicePool, 64, (size_t)96 << 30), g_arenaCache(BufCache::Device, 64, (size_t)2 << 30),
    g_tbDevCache(BufCache::Device, 64, (size_t)8 << 30), g_tbHostCache(BufCache::PinnedHost, 64, (size_t)2 << 30),
    g_stageCache(BufCache::PinnedHost, 64, (size_t)64 << 20); /* upload images of small batches (dpx_batch_create: one H2D instead of five) */

/* hipStreamCreate / hipStreamDestroy cost ~2 ms each on this stack: a finished batch parks its (idle) stream */
struct StreamCache {
    std::mutex mu;
    std::vector<std::pair<hipStream_t, int>> parked; /* (stream, device) */
} g_streams;

hipError_t stream_take(hipStream_t *out) {
    {
        std::lock_guard<std::mutex> lk(g_streams.mu);
        /* the most recently parked one first: a pipeline of two batches in flight then keeps using the same two streams.  The runtime
         * maps streams onto four hardware queues; cycling through six parked streams put the fill of batch k+1 on the queue of batch
         * k's traceback every few batches, where it waited for it (profiles/r04/e2e_long_timeline_*.txt) */
        for (size_t i = g_streams.parked.size(); i-- > 0;)
            if (g_streams.parked[i].second == t_device) {
                *out = g_streams.parked[i].first;
                g_streams.parked.erase(g_streams.parked.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_park(hipStream_t s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(g_streams.mu);
        if (g_streams.parked.size() < 256 && t_device >= 0) { g_streams.parked.emplace_back(s, t_device); return; }
    }
    (void)hipStreamDestroy(s);
}

void trim_all_caches() {
    std::vector<std::pair<hipStream_t, int>> st;
    { std::lock_guard<std::mutex> lk(g_streams.mu); st.swap(g_streams.parked); }
    for (auto &e : st) (void)hipStreamDestroy(e.first);
    g_matCache.drain();
    g_arenaCache.drain();
    g_tbDevCache.drain();
    g_tbHostCache.drain();
    g_stageCache.drain();
}

} // namespace

namespace {
/* DPX_TRACE=1: phase timings of dpx_batch_create / destroy on stderr (development aid) */
struct PhaseTrace {
    bool on = knobs().trace;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[dpx] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};
} // namespace

/* The record of a matrix pool: how it was built and how fast hipMemset writes it (dpx_batch_describe -> bench.py's roofline.pool),
 * so that a slow run can be attributed to the pool or to something else.  Kept per pool address for the life of the pool (a
 * parked pool keeps its record for the next batch). */
struct PoolRecord {
    std::string mode = "malloc";
    size_t bytes = 0, chunkBytes = 0;
    std::vector<float> candidatesMs; /* hipMemset time of every candidate allocation (empty: never timed) */
    std::vector<float> fillMs;       /* time of one fill of the batch on every candidate (empty: never shopped) */
    std::vector<std::string> kinds;  /* how every candidate was built: "vmm256", "malloc" ... (empty: never shopped) */
    std::vector<std::string> ranges; /* "address+bytes" of every candidate (a GPU fault address can be placed against them) */
    int kept = 0;
};
static std::mutex g_poolRecMu;
static std::map<void *, PoolRecord> g_poolRecords;
namespace { void forget_pool_record(void *pool) { std::lock_guard<std::mutex> lk(g_poolRecMu); g_poolRecords.erase(pool); } }
/* the record of a pool nobody has timed yet: how it was built (looked up by address: the pool may have been built by another thread) */
static PoolRecord fresh_pool_record(void *pool, size_t bytes) {
    PoolRecord rec;
    const size_t chunk = pool_chunk_bytes(pool);
    rec.mode = chunk ? "vmm" : "malloc";
    rec.chunkBytes = chunk;
    rec.bytes = bytes;
    return rec;
}

struct dpx_batch {
    int device = -1; /* the device the batch lives on */
    dpx_params prm{};
    unsigned flags = 0;
    size_t numPairs = 0;
    int R = 8;          /* rows per lane (full-matrix kernels) or cells per lane (band kernel) */
    int kernelAlgo = 0; /* algorithm the kernel runs (BSW with a covering band runs as LSW) */
    int planes = 1;
    bool store = true;
    bool filled = false;
    uint64_t cells = 0, matElems = 0, algBytes = 0, bandCells = 0;
    size_t matPoolBytes = 0; /* bytes of the block behind dMat (may exceed matElems*2 when a parked pool was reused) */
    size_t guardBytes = 0;   /* DPX_POOL_GUARD: pattern bytes behind the matrices, checked by dpx_batch_sync() */
    bool guardSelfTest = false;
    int maxN = 0, maxM = 0;
    std::vector<dpx_pair_dev> pairs; /* host mirror of the device pair table */
    char *dSeq = nullptr;
    dpx_pair_dev *dPairs = nullptr;
    int32_t *dOrder = nullptr;
    char *arena = nullptr;   /* one device allocation behind dSeq, dPairs, dScore/dEndRow/dEndCol, dOrder, dCouples, dTbOff, dTbLen
                                (parked and reused across batches like the matrix pool: 9 hipMalloc/hipFree pairs per batch otherwise) */
    size_t arenaCap = 0;
    int16_t *dMat = nullptr;
    int32_t *dScore = nullptr, *dEndRow = nullptr, *dEndCol = nullptr;
    hipStream_t stream = nullptr;     /* the batch's own stream */
    hipStream_t sideStream = nullptr; /* secondary kernels of a fill run here, concurrently with the main one (launch_all) */
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipEvent_t evT0 = nullptr, evT1 = nullptr; /* DPX_TIME_FILLS: recorded around every dpx_batch_fill() */
    hipEvent_t evOrder = nullptr;              /* orders the output path behind a fill that ran on a caller's stream */
    hipEvent_t evOut0 = nullptr, evOut1 = nullptr; /* DPX_TIME_FILLS: around the traceback + text kernels of dpx_batch_output_begin() */
    bool outTimed = false;
    bool fillTimed = false;
    hipStream_t lastStream = nullptr; /* stream of the most recent fill (caller's or own) */
    dpx_fill_args args{};
    size_t ldsBytes = 0;
    /* packed two-pairs-per-wave path (LNW/LSW with matrices): couples of equal-shaped pairs + leftover singles */
    bool packed = false;
    bool split = false;    /* small batch: one workgroup per pair, one wave per stripe (k_linear_split) */
    bool packed2 = false;  /* the sequences arrived as 2-bit codes (dpx_batch_create_packed2) */
    bool lanesPk = false;  /* lane-packed batch on k_linear_lanes_pk (two row blocks of a pair in the halves of every register) */
    int splitWaves = 0;
    size_t splitLds = 0;
    bool lanePacked = false; /* short queries: several pairs per wave (k_linear_lanes / k_affine_lanes, 8 x 8 tile layout); wave
                              descriptors in dCouples, launch arguments in pkArgs */
    int32_t *dCouples = nullptr;
    dpx_fill_args pkArgs{};
    size_t pkLdsBytes = 0;
    /* result text (lazy): device line buffers of k_traceback, the packed blocks built from them, pinned host mirrors */
    uint64_t *dTbOff = nullptr;
    char *dTb = nullptr;
    int32_t *dTbLen = nullptr;
    uint64_t *dOutOff = nullptr, *dOutScratch = nullptr; /* per-pair byte offsets of the blocks (+ total), scan scratch */
    char *dOut = nullptr;
    char *hStage = nullptr; /* pinned image of the arena's uploaded front (small batches), alive until the batch is destroyed */
    size_t hStageCap = 0;
    bool resultsCopied = false; /* ... and of score / end row / end column, into the tail of hMeta */
    bool textCopied = false;    /* dpx_batch_output_begin() has queued the text's D2H itself (small batches) */
    bool uploadPending = false; /* ... and its one asynchronous H2D on b->stream has not been waited for by anybody yet */
    std::vector<uint64_t> tbOff;
    char *hMeta = nullptr; /* pinned: uint64 offsets[numPairs + 1], then int32 alignment lengths[numPairs] */
    char *hOut = nullptr;  /* pinned: the packed text */
    size_t hOutBytes = 0;
    size_t dTbCap = 0, dOutCap = 0, hMetaCap = 0, hOutCap = 0; /* capacities of the (possibly recycled) buffers */
    bool tbLinesValid = false; /* k_traceback has run since the last fill */
    int outState = 0;          /* 0 none, 1 dpx_batch_output_begin() in flight, 2 text on the host */
    uint64_t outFirst = 0;     /* pair number of the batch's first pair in the text */
    size_t nSingles = 0, nCouples = 0, nLanePairs = 0, nWaves = 0; /* launch-list sizes (dpx_batch_describe) */
    PoolRecord poolRec;    /* how the matrix pool behind dMat was built / timed */
    bool tunePool = false, tuneShop = false; /* DPX_TUNE_PLACEMENT: the pool is timed / shopped for at the end of dpx_batch_create */
};

extern "C" {

int dpx_abi_version(void) { return DPX_ABI_VERSION; }

const char *dpx_strerror(int status) {
    switch (status) {
    case DPX_OK: return "ok";
    case DPX_ERR_INVALID: return "invalid argument";
    case DPX_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case DPX_ERR_HIP: return "HIP runtime error";
    case DPX_ERR_RANGE: return "scores of this batch do not fit the int16 matrix cells";
    case DPX_ERR_NOMEM: return "out of memory";
    case DPX_ERR_NOT_FILLED: return "batch has not been filled yet";
    case DPX_ERR_NO_MATRIX: return "batch was created with DPX_SCORE_ONLY";
    case DPX_ERR_UNSUPPORTED: return "unsupported parameter combination";
    default: return "unknown status";
    }
}

const char *dpx_last_error(void) { return t_err.c_str(); }

int dpx_device_count(int *count) {
    if (!count) return DPX_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); *count = 0; return DPX_ERR_NO_DEVICE; }
    *count = n;
    return DPX_OK;
}

int dpx_init(int device) {
    refresh_knobs();
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        t_err = "hipGetDeviceCount found no device";
        return DPX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return DPX_ERR_INVALID;
    e = hipSetDevice(device);
    if (e != hipSuccess) { hip_fail(e, "hipSetDevice"); return DPX_ERR_NO_DEVICE; }
    const bool first = g_device != device;
    g_device = device;
    t_device = device;
    if (first) {
        /* One-time costs of the HIP runtime belong to device initialisation (the reference's mains likewise create their
         * context and query the device before they start their timer, cuda/LNW/LinearNeedlemanWunschV19.cu:357-409):
         * the first hipStreamCreate costs ~9 ms, the first pageable H2D / D2H another ~9 ms (runtime staging buffers).
         * Two streams go to the stream cache, and a small copy runs in each direction.  The first ASYNCHRONOUS copy into pinned host
         * memory on a stream costs another ~18 ms (round 3, DPX_TRACE of dpx_main: "output: D2H text 18.7 ms" for the first batch's
         * 5 MB, 0.3 ms for every later one): one such copy of 4 MiB runs here, in both directions, on both parked streams. */
        hipStream_t s0 = nullptr, s1 = nullptr;
        const bool have0 = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking) == hipSuccess;
        const bool have1 = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess;
        void *d = nullptr;
        std::vector<char> h((size_t)1 << 20, 0);
        const size_t warm = (size_t)4 << 20;
        if (hipMalloc(&d, warm) == hipSuccess) {
            (void)hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
            (void)hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost);
            void *pinned = nullptr;
            if (hipHostMalloc(&pinned, warm, hipHostMallocDefault) == hipSuccess) {
                for (hipStream_t s : {have0 ? s0 : nullptr, have1 ? s1 : nullptr}) {
                    if (!s) continue;
                    (void)hipMemcpyAsync(pinned, d, warm, hipMemcpyDeviceToHost, s);
                    (void)hipMemcpyAsync(d, pinned, warm, hipMemcpyHostToDevice, s);
                    (void)hipStreamSynchronize(s);
                }
                (void)hipHostFree(pinned);
            }
            (void)hipFree(d);
        }
        if (have0) stream_park(s0);
        if (have1) stream_park(s1);
        (void)hipGetLastError();
    }
    return DPX_OK;
}

int dpx_device_info(char *name, size_t nameCap, int *computeUnits, size_t *hbmBytes) {
    int rc = bind_device();
    if (rc != DPX_OK) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, t_device));
    if (name && nameCap) { snprintf(name, nameCap, "%s (%s)", prop.name, prop.gcnArchName); }
    if (computeUnits) *computeUnits = prop.multiProcessorCount;
    if (hbmBytes) *hbmBytes = prop.totalGlobalMem;
    return DPX_OK;
}

/* Allocate `count` matrix pools of `bytes` on the default device and park them for the batches to come.  A driver calls this
 * on a helper thread while it is still parsing its input: building a pool costs tens of ms per GiB (the reference's V9 / V14
 * lesson: size the buffers once, outside the loop -- cuda/LNW/LinearNeedlemanWunschV9.cu:26-46, V14.cu:144-213). */
int dpx_pool_reserve(size_t bytes, int count) {
    if (count < 1 || count > 8 || bytes == 0) return DPX_ERR_INVALID;
    if (bytes >= ((size_t)16 << 30)) count = 1; /* (only one pool of 16 GiB or more stays parked per device: a second one would be built and dropped at once) */
    int rc = bind_device();
    if (rc != DPX_OK) return rc;
    void *p[8] = {nullptr};
    size_t got[8] = {0};
    for (int k = 0; k < count; k++) {
        bool fresh = false;
        hipError_t e = g_matCache.take(&p[k], bytes, &got[k], &fresh);
        if (e != hipSuccess) { for (int j = 0; j < k; j++) g_matCache.park(p[j], got[j]); return hip_fail(e, "dpx_pool_reserve"); }
    }
    for (int k = 0; k < count; k++) {
        { /* the pool's record, for the batch (of whatever thread) that takes it */
            std::lock_guard<std::mutex> lk(g_poolRecMu);
            if (!g_poolRecords.count(p[k])) g_poolRecords[p[k]] = fresh_pool_record(p[k], got[k]);
        }
        g_matCache.park(p[k], got[k]);
    }
    /* ... and the streams of the batches that will take the pools: a batch runs on its own stream plus a side stream for the pairs its
     * main kernel leaves over (an odd pair beside the couples of the packed kernel), and creating a stream costs ~10 ms each time the
     * cache is empty (DPX_TRACE of the batched driver: "create: stream 9.6 ms" twice for two batches in flight) */
    {
        hipStream_t extra[10] = {nullptr};
        int made = 0;
        for (int k = 0; k < count + 2; k++) /* (one per batch in flight + two for side kernels) */
            if (hipStreamCreateWithFlags(&extra[made], hipStreamNonBlocking) == hipSuccess) made++;
            else (void)hipGetLastError();
        for (int k = 0; k < made; k++) stream_park(extra[k]);
    }
    return DPX_OK;
}

/* The same for the pinned host buffers that the result text of a batch is copied into (dpx_batch_output_end / _take): pinning
 * 8 MB costs 1-2 ms, and a pipelined driver holds up to three of them (one being printed, two batches in flight). */
int dpx_text_reserve(size_t bytes, int count) {
    if (count < 1 || count > 9 || bytes == 0 || bytes > ((size_t)1 << 30)) return DPX_ERR_INVALID;
    int rc = bind_device();
    if (rc != DPX_OK) return rc;
    void *p[9] = {nullptr};
    size_t got[9] = {0};
    for (int k = 0; k < count; k++) {
        hipError_t e = g_tbHostCache.take(&p[k], bytes, &got[k]);
        if (e != hipSuccess) { for (int j = 0; j < k; j++) g_tbHostCache.park(p[j], got[j]); return hip_fail(e, "dpx_text_reserve"); }
    }
    for (int k = 0; k < count; k++) g_tbHostCache.park(p[k], got[k]);
    return DPX_OK;
}

int dpx_shutdown(void) {
    trim_all_caches();
    std::lock_guard<std::mutex> lk(g_mu);
    g_device = -1;
    return DPX_OK;
}

/* ------------------------------------------------------------------------------------------ batch */

static int validate_params(const dpx_params *p) {
    if (!p) return DPX_ERR_INVALID;
    if (p->algo < DPX_ALGO_LNW || p->algo > DPX_ALGO_BSW) return DPX_ERR_INVALID;
    if (p->algo == DPX_ALGO_BSW && p->band < 1) return DPX_ERR_INVALID;
    /* the int32 kernels add a weight to a cell value (|H| <= 32767 after fits_int16) and to the affine kernels' virtual
     * -2^29 borders: weights beyond +-2^20 could wrap those sums (and no int16 matrix could hold what they produce) */
    const long long lim = 1ll << 20;
    for (long long w : {(long long)p->match, (long long)p->mismatch, (long long)p->gapOpen,
                        p->algo == DPX_ALGO_ANW ? (long long)p->gapExtend : 0ll})
        if (w > lim || w < -lim) return DPX_ERR_RANGE;
    return DPX_OK;
}

/* Can every value the algorithm stores for an (m x n) pair be held in an int16 cell?  Rigorous bounds: every H cell is
 * the maximum over alignment paths, so it is >= the score of the all-gap path and <= the sum of the positive
 * contributions any path can collect; the affine gap matrices are one open/extension away from an H cell. */
static bool fits_int16(const dpx_params &p, long long m, long long n) {
    auto pos = [](long long v) { return v > 0 ? v : 0; };
    auto neg = [](long long v) { return v < 0 ? v : 0; };
    const long long lim = 32767, diag = pos(std::max<long long>(p.match, p.mismatch)) * std::min(m, n);
    if (p.algo == DPX_ALGO_LSW || p.algo == DPX_ALGO_BSW) {
        /* 0 <= H <= best diagonal run (+ positive gaps); the kernels also pack a column / step index into 16 bits */
        const long long top = diag + pos(p.gapOpen) * (m + n);
        return top <= lim && n <= 65000 && (m + n) <= 65000;
    }
    if (p.algo == DPX_ALGO_LNW) {
        const long long lo = neg(p.gapOpen) * (m + n), hi = diag + pos(p.gapOpen) * (m + n);
        return lo >= -lim && hi <= lim;
    }
    const long long o = p.gapOpen, e = p.gapExtend;
    const long long loH = 2 * neg(o) + neg(e) * (m + n), hiH = diag + (pos(o) + pos(e)) * (m + n);
    const long long lo = loH + neg(o + e), hi = hiH + pos(o) + pos(e) * std::max(m, n);
    return lo >= -lim && hi <= lim;
}

/* The "+Opt" packed kernel (k_linear_fill_pk) does EVERY add in wrapping 16-bit halves (v_pk_add_i16 / v_pk_mad_i16), so
 * not only the stored H values but the weights themselves and the intermediates `diag + s` and `max(up, left) + gap`
 * must stay inside int16: H lies in [lo, hi] (fits_int16's bounds), an intermediate is one weight away from an H value
 * or a border value.  Batches that fail this run on the int32 kernels (same results, whatever the batch size). */
static bool packed_safe(const dpx_params &p, long long m, long long n) {
    if (p.algo != DPX_ALGO_LNW && p.algo != DPX_ALGO_LSW && p.algo != DPX_ALGO_BSW) return false;
    auto pos = [](long long v) { return v > 0 ? v : 0; };
    auto neg = [](long long v) { return v < 0 ? v : 0; };
    const long long wmin = std::min<long long>({p.match, p.mismatch, p.gapOpen, 0}), wmax = std::max<long long>({p.match, p.mismatch, p.gapOpen, 0});
    if (wmin < -32768 || wmax > 32767) return false;
    const long long diag = pos(std::max<long long>(p.match, p.mismatch)) * std::min(m, n);
    const long long hi = diag + pos(p.gapOpen) * (m + n);
    const long long lo = p.algo == DPX_ALGO_LNW ? neg(p.gapOpen) * (m + n) : 0;
    return lo + wmin >= -32768 && hi + wmax <= 32767 && n <= 65535;
}

/* rows per lane of the lane-packed kernels: a pair of m rows takes ceil(m / 8) lanes (m <= 512); 16 rows per lane (two row
 * blocks) for the linear-gap kernels up to 1024 rows */
static int lanes_rows(int maxM, int algo) { return (maxM <= 512 || algo == DPX_ALGO_ANW) ? 8 : 16; }

/* Pack pairs (`idx`, sorted by reference length, longest first) into waves of 64 lanes for the lane-packed kernels:
 * a pair takes ceil(m / R) consecutive lanes, a wave up to DPX_WAVE_SLOTS pairs whose staged references fit the wave's
 * LDS reference area.  Best fit over a window of open waves, so that the pairs of a wave have nearly the same reference
 * length (the wave runs max(n + lanes) steps) and the lanes fill up: 100k short reads (queries 80-130) reach 94 % lane
 * occupancy.  Returns the reference area in bytes; `idx` comes back in slot order (= matrix placement order). */
static size_t pack_waves(const std::vector<dpx_pair_dev> &pairs, std::vector<int32_t> &idx, int R, int maxN, size_t refFloor, int maxSlots,
                         std::vector<dpx_wave_desc> &waves) {
    struct Bin { dpx_wave_desc d; int lanes = 0, slots = 0; size_t ref = 0; bool closed = false; };
    const size_t refCap = std::max<size_t>(refFloor, align_up((size_t)maxN + 31, 16)); /* bytes of staged references per wave */
    constexpr size_t kWindow = 256; /* open waves: pairs arrive sorted by reference length, so a window keeps a wave's pairs alike */
    std::vector<Bin> bins;          /* in opening order = emission order */
    std::vector<int32_t> byFree[65]; /* open bins by free lanes (entries go stale when a bin moves on: checked on use) */
    uint64_t nonEmpty = 0;           /* bit f-1: byFree[f] holds entries (the search skips the empty buckets with one ctz) */
    size_t oldestOpen = 0, numOpen = 0;
    bins.reserve(idx.size() / 3 + 8);
    for (int32_t i : idx) {
        const dpx_pair_dev &pd = pairs[i];
        const int L = (pd.m + R - 1) / R;
        const size_t need = align_up((size_t)pd.n + 31, 16);
        int best = -1;
        for (uint64_t cand = L <= 64 ? nonEmpty & (~0ull << (L - 1)) : 0; cand && best < 0; cand &= cand - 1) { /* tightest fit first */
            const int f = __builtin_ctzll(cand) + 1;
            std::vector<int32_t> &lst = byFree[f];
            while (!lst.empty()) {
                const int32_t k = lst.back();
                const Bin &bn = bins[k];
                if (bn.closed || 64 - bn.lanes != f) { lst.pop_back(); continue; } /* stale entry */
                if (bn.slots >= maxSlots || bn.ref + need > refCap) break;   /* this bucket's newest bin is full in another way */
                best = k;
                lst.pop_back();
                break;
            }
            if (lst.empty()) nonEmpty &= ~(1ull << (f - 1));
        }
        if (best < 0) {
            if (numOpen >= kWindow) { /* retire the oldest open wave */
                while (bins[oldestOpen].closed) oldestOpen++;
                bins[oldestOpen].closed = true;
                numOpen--;
            }
            bins.emplace_back();
            memset(&bins.back().d, 0, sizeof(dpx_wave_desc));
            best = (int)bins.size() - 1;
            numOpen++;
        }
        Bin &bn = bins[best];
        bn.d.pair[bn.slots] = i;
        bn.d.first[bn.slots] = (uint8_t)bn.lanes;
        bn.d.num[bn.slots] = (uint8_t)L;
        bn.d.refOff[bn.slots] = (uint16_t)(bn.ref >> 4);
        bn.lanes += L;
        bn.slots++;
        bn.ref += need;
        if (bn.lanes < 64 && bn.slots < maxSlots) { byFree[64 - bn.lanes].push_back(best); nonEmpty |= 1ull << (64 - bn.lanes - 1); }
        else { bn.closed = true; numOpen--; }
    }
    std::vector<int32_t> order;
    order.reserve(idx.size());
    size_t area = 0;
    waves.reserve(bins.size());
    for (const Bin &bn : bins) {
        waves.push_back(bn.d);
        for (int k = 0; k < bn.slots; k++) order.push_back(bn.d.pair[k]);
        area = std::max(area, bn.ref);
    }
    idx.swap(order);
    return area;
}

/* cells (i, j) with 1 <= i <= m, 1 <= j <= n, |i - j| <= B - 1: sum over the rows of min(n, i+B-1) - max(1, i-B+1) + 1 */
static uint64_t band_cells(long long m, long long n, long long B) {
    if (m <= 0 || n <= 0 || B <= 0) return 0;
    const long long M = std::min(m, n + B - 1);                 /* rows that still reach the band */
    const long long a = std::max(0ll, std::min(n - B + 1, M));  /* rows whose right end is i + B - 1 (not clipped at n) */
    const long long s1 = a * (a + 1) / 2 + a * (B - 1) + (M - a) * n;
    const long long b = std::min(B, M);                         /* rows whose left end is clipped at column 1 */
    const long long s2 = b + (M > B ? (M - B + 1) * (M - B + 2) / 2 - 1 : 0);
    return (uint64_t)(s1 - s2 + M);
}

/* hipMemset time of a pool (the record of a pool that a resident-batch caller is going to fill many times; 2 x 3.5 ms for 22 GB) */
static float time_memset(void *p, size_t bytes, hipStream_t s) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) { (void)hipGetLastError(); return -1.f; }
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e0); return -1.f; }
    float best = 1e30f;
    for (int k = 0; k < 2; k++) {
        float ms = 1e30f;
        if (hipEventRecord(e0, s) != hipSuccess || hipMemsetAsync(p, 0, bytes, s) != hipSuccess || hipEventRecord(e1, s) != hipSuccess ||
            hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); best = -1.f; break; }
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

static hipError_t launch_all(dpx_batch *b, hipStream_t s);

/* Waves per workgroup of the one-wave-per-pair / -couple kernels.  Their waves share nothing, so the workgroup is only the unit in which
 * the dispatcher hands waves to CUs: with four-wave workgroups 1250 waves are 313 workgroups on 256 CUs -- 57 CUs get eight waves, the
 * others four, and the launch takes as long as eight (tools/pairs_sweep.py: 2500 pairs of 1024^2 on the packed kernel 2322 GCUPS, 1900
 * pairs = 238 workgroups 3265).  Small launches therefore use one-wave workgroups (the CUs differ by at most one wave); big ones keep
 * four (a few per cent faster there: fewer workgroups to dispatch, profiles/r03/ab_nontemporal_stores_and_wg64.txt).  DPX_WPB=1|4 forces one. */
static uint32_t fill_waves_per_block(size_t waves) {
    { const int v = knobs().wavesPerBlock; if (v == 1 || v == 4) return (uint32_t)v; }
    return waves <= 4096 ? 1u : 4u;
}

/* DPX_TUNE_PLACEMENT (callers that fill a resident batch many times: bench.py, iterative drivers).  The same fill runs up to
 * 27 % apart on two pools of the same construction (ANW 1000 x 1024^2: 1.05 vs 1.20 ms, alternating from one allocation to the
 * next while hipMemset sees no difference; LSW 10k x 1024^2: +-2 %; tools/mode_watch.py, profiles/r03/): the mode belongs to the
 * allocation -- where its physical chunks lie -- and only the fill itself shows it.  So the batch shops with its own fill: up to five
 * candidate pools of the SAME construction, one warm-up + three timed fills each, the fastest is kept (and parked for later batches),
 * the losers are freed together when the last candidate has been timed.  Every candidate's address range and times go into the pool
 * record (dpx_batch_describe -> bench.py roofline.pool; fillMs[0] is the pool a caller without the flag would have got).
 * Round 3 also compared CONSTRUCTIONS here (candidates of 512-MiB, 1-GiB and 2-GiB chunks): twice in ~60 runs a process died with a GPU
 * memory access fault while such a candidate was being filled (profiles/r03/gpu_fault_while_shopping.txt), never in hundreds of runs on
 * 256-MiB chunks.  Those ranges were reserved with alignment 0 and their chunks mapped at offsets that were multiples of 2 MiB only, not
 * of the chunk size or of the queried granularity (pool_alloc now reserves with the chunk size as alignment); whether that was the
 * cause cannot be shown from one clean run, so the branch is gone (round 4) rather than shipped on trust. */
static void shop_pool_by_fill(dpx_batch *b, PoolRecord &rec, PhaseTrace &trace) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e0); return; }
    auto set_pool = [&](void *p) { b->dMat = (int16_t *)p; b->args.mat = b->dMat; b->pkArgs.mat = b->dMat; };
    auto time_fill = [&]() -> float {
        float ms = -1.f;
        hipError_t e = launch_all(b, b->stream);
        if (e == hipSuccess) e = hipEventRecord(e0, b->stream);
        for (int i = 0; i < 3 && e == hipSuccess; i++) e = launch_all(b, b->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, b->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) { (void)hipGetLastError(); return -1.f; }
        return ms / 3.f;
    };
    auto kind_of = [](void *pool) { const size_t c = pool_chunk_bytes(pool); return c ? "vmm" + std::to_string(c >> 20) : std::string("malloc"); };
    auto range_of = [&](void *pool) { char t[64]; snprintf(t, sizeof t, "%p+%zu", pool, b->matPoolBytes); return std::string(t); };
    const size_t bytes = b->matPoolBytes;
    void *best = b->dMat;
    float bestMs = time_fill();
    rec.fillMs.assign(1, bestMs);
    rec.kept = 0;
    rec.kinds.assign(1, kind_of(best));
    rec.ranges.assign(1, range_of(best));
    /* NO pool is unmapped before the last candidate has been filled: the losers are freed together at the end (memory permitting -- the
     * loop stops when the next candidate would not leave 8 GiB of THIS device's memory free; every rank of a multi-GPU job shops on its
     * own device only).  Five candidates in all, no early stop: a rehearsal that stopped after [3.73, 3.59] -- "both modes seen" -- ran
     * at 2907 GCUPS, the next one found 3.40 with its fourth candidate after [3.82, 3.80, 3.76]. */
    std::vector<void *> losers;
    for (int k = 1; k < 5 && bestMs > 0.f; k++) {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || freeB < bytes + ((size_t)8 << 30)) { (void)hipGetLastError(); break; }
        void *cand = nullptr;
        if (pool_alloc(&cand, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        rec.kinds.push_back(kind_of(cand));
        rec.ranges.push_back(range_of(cand));
        if (trace.on) { fprintf(stderr, "[dpx] pool candidate %d: %s %s\n", k, rec.kinds.back().c_str(), rec.ranges.back().c_str()); fflush(stderr); }
        set_pool(cand);
        const float ms = time_fill();
        rec.fillMs.push_back(ms);
        rec.candidatesMs.push_back(time_memset(cand, bytes, b->stream));
        if (ms > 0.f && ms < bestMs) { losers.push_back(best); best = cand; bestMs = ms; rec.kept = k; rec.mode = pool_chunk_bytes(cand) ? "vmm" : "malloc"; rec.chunkBytes = pool_chunk_bytes(cand); }
        else losers.push_back(cand);
    }
    set_pool(best);
    (void)hipStreamSynchronize(b->stream);
    for (void *p : losers) pool_free(p);
    if (trace.on) for (size_t k = 0; k < rec.fillMs.size(); k++) fprintf(stderr, "[dpx] pool candidate %zu (%s): fill %.3f ms%s\n", k, rec.kinds[k].c_str(), rec.fillMs[k], (int)k == rec.kept ? "  <- kept" : "");
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

static int check_guard(dpx_batch *b);

int dpx_batch_destroy(dpx_batch *b) {
    if (!b) return DPX_OK;
    PhaseTrace trace;
    if (b->device >= 0) { (void)hipSetDevice(b->device); t_device = b->device; }
    /* buffers are parked for the next batch, not freed: nothing of this batch may still be running on them */
    if (b->lastStream && b->lastStream != b->stream) (void)hipStreamSynchronize(b->lastStream);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); stream_park(b->stream); }
    if (b->sideStream) { (void)hipStreamSynchronize(b->sideStream); stream_park(b->sideStream); }
    if (b->guardBytes && !b->guardSelfTest && check_guard(b) != DPX_OK) { fprintf(stderr, "[dpx] %s\n", t_err.c_str()); abort(); }
    if (b->evT0) (void)hipEventDestroy(b->evT0);
    if (b->evT1) (void)hipEventDestroy(b->evT1);
    if (b->evFork) (void)hipEventDestroy(b->evFork);
    if (b->evJoin) (void)hipEventDestroy(b->evJoin);
    if (b->evOrder) (void)hipEventDestroy(b->evOrder);
    if (b->evOut0) (void)hipEventDestroy(b->evOut0);
    if (b->evOut1) (void)hipEventDestroy(b->evOut1);
    g_arenaCache.park(b->arena, b->arenaCap);
    g_matCache.park(b->dMat, b->matPoolBytes);
    g_tbDevCache.park(b->dTb, b->dTbCap);
    g_tbDevCache.park(b->dOut, b->dOutCap);
    g_tbHostCache.park(b->hMeta, b->hMetaCap);
    g_tbHostCache.park(b->hOut, b->hOutCap);
    g_stageCache.park(b->hStage, b->hStageCap);
    delete b;
    trace.mark("destroy");
    return DPX_OK;
}

int dpx_batch_create(const dpx_params *params, const char *sequences, size_t numBytes, const dpx_seq_pair *pairs,
                     size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out) {
    return dpx_batch_create_on(-1, params, sequences, numBytes, pairs, firstPair, numPairs, flags, out);
}

/* `alphabet` == nullptr: `sequences` are numBytes plain bytes.  Otherwise `sequences` holds numBytes BASES, four per byte (base k in
 * bits 2*(k%4) of byte k/4), alphabet[code] is the byte a code stands for, and the pairs' indices count bases. */
static int create_impl(int device, const dpx_params *params, const char *sequences, size_t numBytes, const uint8_t *alphabet,
                       const dpx_seq_pair *pairs, size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out);

int dpx_batch_create_on(int device, const dpx_params *params, const char *sequences, size_t numBytes, const dpx_seq_pair *pairs,
                        size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out) {
    return create_impl(device, params, sequences, numBytes, nullptr, pairs, firstPair, numPairs, flags, out);
}

int dpx_batch_create_packed2(int device, const dpx_params *params, const uint8_t *packed, size_t numBases, const uint8_t alphabet[4],
                             const dpx_seq_pair *pairs, size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out) {
    if (!alphabet) return DPX_ERR_INVALID;
    return create_impl(device, params, reinterpret_cast<const char *>(packed), numBases, alphabet, pairs, firstPair, numPairs, flags, out);
}

/* Host side of the 2-bit input: the distinct byte values inside the pairs' ranges become the alphabet (in order of first appearance;
 * more than four: DPX_ERR_UNSUPPORTED, the caller keeps its bytes), every byte of `sequences` its 2-bit code (bytes outside
 * every pair -- parseInput's separators -- and bytes of other values: code 0, never read).  packed: (numBytes + 3) / 4 bytes. */
int dpx_pack2(const char *sequences, size_t numBytes, const dpx_seq_pair *pairs, size_t numPairs, uint8_t alphabet[4], uint8_t *packed) {
    if ((numBytes && !sequences) || (numPairs && !pairs) || !alphabet || (numBytes && !packed)) return DPX_ERR_INVALID;
    int code[256];
    for (int &c : code) c = -1;
    int used = 0;
    const unsigned char *sq = reinterpret_cast<const unsigned char *>(sequences);
    for (size_t i = 0; i < numPairs; i++) {
        const dpx_seq_pair &sp = pairs[i];
        if (sp.referenceSize < 0 || sp.querySize < 0 || sp.referenceIdx < 0 || sp.queryIdx < 0 ||
            (size_t)sp.referenceIdx + (size_t)sp.referenceSize > numBytes || (size_t)sp.queryIdx + (size_t)sp.querySize > numBytes)
            return DPX_ERR_INVALID;
        for (int side = 0; side < 2; side++) {
            const unsigned char *p = sq + (side ? sp.queryIdx : sp.referenceIdx);
            const size_t len = (size_t)(side ? sp.querySize : sp.referenceSize);
            for (size_t k = 0; k < len; k++) {
                if (code[p[k]] >= 0) continue;
                if (used == 4) return DPX_ERR_UNSUPPORTED;
                alphabet[used] = p[k];
                code[p[k]] = used++;
            }
        }
    }
    for (int k = used; k < 4; k++) alphabet[k] = used ? alphabet[0] : (uint8_t)'0';
    uint8_t lut[256];
    for (int v = 0; v < 256; v++) lut[v] = (uint8_t)(code[v] < 0 ? 0 : code[v]);
    const size_t whole = numBytes / 4;
    for (size_t q = 0; q < whole; q++)
        packed[q] = (uint8_t)(lut[sq[4 * q]] | (lut[sq[4 * q + 1]] << 2) | (lut[sq[4 * q + 2]] << 4) | (lut[sq[4 * q + 3]] << 6));
    if (numBytes & 3) {
        uint8_t v = 0;
        for (size_t k = 4 * whole; k < numBytes; k++) v = (uint8_t)(v | (lut[sq[k]] << (2 * (k & 3))));
        packed[whole] = v;
    }
    return DPX_OK;
}

static int create_impl(int device, const dpx_params *params, const char *sequences, size_t numBytes, const uint8_t *alphabet,
                       const dpx_seq_pair *pairs, size_t firstPair, size_t numPairs, unsigned flags, dpx_batch **out) {
    if (!out) return DPX_ERR_INVALID;
    *out = nullptr;
    int rc = validate_params(params);
    if (rc != DPX_OK) return rc;
    if ((numPairs && (!pairs || !sequences)) || numPairs > 0x7fffffffu) return DPX_ERR_INVALID;
    if (device >= 0) { /* an explicit device: must exist (-1 = the default device of dpx_init) */
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return DPX_ERR_NO_DEVICE; }
        if (device >= n) return DPX_ERR_INVALID;
    } else if (device != -1) return DPX_ERR_INVALID;
    rc = bind_device(device);
    if (rc != DPX_OK) return rc;

    refresh_knobs();
    const Knobs kn = knobs();
    PhaseTrace trace;
    dpx_batch *b = new (std::nothrow) dpx_batch();
    if (!b) return DPX_ERR_NOMEM;
    b->device = t_device;
    b->prm = *params;
    b->flags = flags;
    b->numPairs = numPairs;
    b->store = !(flags & DPX_SCORE_ONLY);
    b->planes = params->algo == DPX_ALGO_ANW ? 3 : 1;
    b->pairs.resize(numPairs);

    /* geometry, validation */
    bool ragged = false;
    for (size_t i = 0; i < numPairs; i++) {
        const dpx_seq_pair &sp = pairs[firstPair + i];
        if (sp.referenceSize < 0 || sp.querySize < 0 || sp.referenceIdx < 0 || sp.queryIdx < 0 ||
            (size_t)sp.referenceIdx + (size_t)sp.referenceSize > numBytes ||
            (size_t)sp.queryIdx + (size_t)sp.querySize > numBytes) {
            delete b;
            return DPX_ERR_INVALID;
        }
        if (!fits_int16(*params, sp.querySize, sp.referenceSize)) { delete b; return DPX_ERR_RANGE; }
        dpx_pair_dev &pd = b->pairs[i];
        pd.refIdx = sp.referenceIdx; pd.n = sp.referenceSize;
        pd.qryIdx = sp.queryIdx;     pd.m = sp.querySize;
        b->maxN = std::max(b->maxN, pd.n);
        b->maxM = std::max(b->maxM, pd.m);
        b->cells += (uint64_t)pd.n * (uint64_t)pd.m;
        if (pd.n != b->pairs[0].n || pd.m != b->pairs[0].m) ragged = true;
    }

    trace.mark("create: validation");
    /* rows per lane: smallest tile that keeps short queries in one stripe, 8 (or DPX_R) otherwise */
    /* linear gaps: 16 rows per lane once a query is longer than 512 (one stripe up to 1024 rows, two 1-KiB sub-tiles per
     * step: measured 4 % faster than 8 rows x 2 rolling stripes); the affine kernel carries three chains and stays at 8 */
    int R = b->maxM <= 128 ? 2 : b->maxM <= 256 ? 4 : (b->maxM <= 512 || params->algo == DPX_ALGO_ANW) ? 8 : 16;
    if (kn.rowsPerLane) {
        const int v = kn.rowsPerLane;
        if (v == 2 || v == 4 || v == 8 || (v == 16 && params->algo != DPX_ALGO_ANW)) R = v;
    }
    b->R = R;
    int kernelAlgo = params->algo;
    if (params->algo == DPX_ALGO_BSW) {
        if (params->band >= std::max(b->maxM, b->maxN)) {
            kernelAlgo = DPX_ALGO_LSW; /* the band covers every cell: identical to the unbanded recurrence */
        } else if (params->band > 512) {
            delete b;
            return DPX_ERR_UNSUPPORTED; /* band kernel holds <= 8 cells per lane (band <= 512) */
        } else {
            b->R = dpx_band_cpl(params->band);
        }
    }
    b->kernelAlgo = kernelAlgo;
    const bool banded = kernelAlgo == DPX_ALGO_BSW;

    /* matrix placement + algorithmic bytes: int16 cells incl. borders, sequences, 16 B pair record, 12 B result */
    for (size_t i = 0; i < numPairs; i++) {
        dpx_pair_dev &pd = b->pairs[i];
        pd.matOff = 0;
        pd.chunkStride = 0;
        pd.lanes = 64;
        pd.rows = 0;
        b->algBytes += (uint64_t)pd.m + (uint64_t)pd.n + 16u + 12u;
        if (b->store) {
            if (banded) { /* 2 B per in-band cell */
                const uint64_t inband = band_cells(pd.m, pd.n, params->band);
                b->bandCells += inband;
                b->algBytes += 2ull * inband;
            } else {
                b->algBytes += 2ull * (uint64_t)b->planes * (uint64_t)(pd.m + 1) * (uint64_t)(pd.n + 1);
            }
        }
    }

    /* LDS per wave: edge row(s) of int16 [n+2] + staged reference [n+128] */
    const size_t edgeBytes = align_up((size_t)(b->maxN + 2) * 2, 16);
    const size_t nEdges = params->algo == DPX_ALGO_ANW ? 2 : 1;
    /* (+16 everywhere: a staged string starts up to 15 bytes into its buffer, at its own address mod 16 -- stage_bytes) */
    const size_t refBytes = align_up((size_t)b->maxN + 128 + 16, 16);
    const size_t qBytes = align_up((size_t)b->maxM + 64 + 32, 16); /* banded kernel: staged query + 64 B of index slack */
    const size_t rollQ = align_up((size_t)b->maxM + 64 * 16 + 32, 16); /* staged query of the rolling multi-stripe schedule */
    const size_t perWave = banded ? qBytes + refBytes : edgeBytes * nEdges + refBytes + rollQ;
    b->ldsBytes = perWave * (DPX_FILL_THREADS / 64);
    /* Store-bound fills run ~2 % faster with 3-4 waves per SIMD than with 6-7 (fewer write streams in flight,
     * profiles/README.md): cap residency at 4 workgroups per CU through the LDS request. */
    if (b->store && !banded && b->ldsBytes < kLdsFloor) b->ldsBytes = kLdsFloor;
    if (b->ldsBytes > 160u * 1024u) { delete b; return DPX_ERR_UNSUPPORTED; }

#define CREATE_TRY(call)                                                      \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) { int r__ = hip_fail(e__, #call); dpx_batch_destroy(b); return r__; } \
    } while (0)

    /* launch lists.  Packed path: couple pairs of identical (m, n); everything else runs one pair per wave, longest first. */
    std::vector<int32_t> singles, couples;
    /* "+Opt" packed path (two equal-shaped pairs per wave on the v_pk_*_i16 pipe).  The fill is bound by store
     * instructions per CU-cycle, so halving the VALU work buys no cycles -- it buys clock: the chip holds ~2.3 GHz
     * instead of ~2.1 GHz under the lighter instruction stream (profiles/README.md), 4-7 % wall time.  Used when every
     * query fits one stripe (the packed kernel has no rolling schedule); DPX_PACKED=0/1 overrides. */
    /* Lane-packed path (short reads, the reference's own dataset shape): queries of <= 512 rows in a batch large enough to
     * fill the chip with a fraction of the waves run several pairs per wave, ceil(m/8) lanes each (k_linear_lanes /
     * k_affine_lanes, tile layout); DPX_LANES=0/1 overrides. */
    const bool linearAlgo = kernelAlgo == DPX_ALGO_LNW || kernelAlgo == DPX_ALGO_LSW;
    const bool lanesAlgo = linearAlgo || kernelAlgo == DPX_ALGO_ANW;
    /* (the staged references of a wave's pairs share its LDS: keep the path to references that leave the request small) */
    /* Packed lane kernel (round 3, k_linear_lanes_pk): 16 rows per lane as two 8-row blocks of the SAME pair in the two halves of every
     * register.  Needs the 16-bit wrapping adds to be safe (packed_safe), room below the smallest border for its "minus infinity"
     * (the low half's column 0), and for SW: score * 8 + 7 in 16 bits (row tags) and gap <= 0, mismatch <= 0 (rows past the query's
     * end are not masked: with such weights they never exceed the real rows above them).  DPX_LANES_PK=0 keeps the int32 kernels. */
    bool lanesPk = false;
    if (linearAlgo && b->store && b->maxM > 0 && b->maxM <= 1024) {
        dpx_params kp = *params;
        kp.algo = kernelAlgo;
        auto pos = [](long long v) { return v > 0 ? v : 0; };
        const long long wmin = std::min<long long>({params->match, params->mismatch, params->gapOpen, 0});
        const long long wmax = std::max<long long>({params->match, params->mismatch, params->gapOpen, 0});
        const long long lowest = kernelAlgo == DPX_ALGO_LNW ? std::min<long long>(params->gapOpen, 0) * ((long long)b->maxM + b->maxN) : 0;
        const long long top = pos(std::max<long long>(params->match, params->mismatch)) * std::min<long long>(b->maxM, b->maxN) +
                              pos(params->gapOpen) * ((long long)b->maxM + b->maxN);
        lanesPk = packed_safe(kp, b->maxM, b->maxN) && (-32768 - wmin + wmax <= lowest) &&
                  (kernelAlgo != DPX_ALGO_LSW || (top * 8 + 7 <= 65535 && params->gapOpen <= 0 && params->mismatch <= 0));
        if (kn.lanesPk >= 0) lanesPk = lanesPk && kn.lanesPk != 0;
    }
    const int kLanesR = lanesPk ? 16 : lanes_rows(b->maxM, kernelAlgo);
    const bool lanesShape = lanesAlgo && b->maxM <= 64 * kLanesR && b->maxM > 0 && b->maxN <= 4096;
    /* measured (tools/lanes_threshold.py): short reads x2048 56 vs 66 us, x4096 56 vs 114, x16384 121 vs 222 (below 2048 pairs the
     * split / one-wave-per-pair kernels win); 20000 x 250x300 (32 lanes per pair, two per wave) 608 vs 645 us, but 300x300 (38 lanes, one
     * per wave) 1074 vs 818 and 180x200 (23 lanes, two per wave) 395 vs 363: the path pays when the packing fills the lanes */
    bool useLanes = lanesShape && b->maxM <= 256 && numPairs >= 2048;
    bool lanesForced = false;
    if (kn.lanes >= 0) { useLanes = kn.lanes != 0 && lanesShape; lanesForced = true; }
    std::vector<dpx_wave_desc> waves;
    size_t lanesRefArea = 0, lanesPairs = 0;
    if (useLanes) {
        b->R = kLanesR; /* empty pairs, if any, run on the one-pair-per-wave kernel at this tile height (they have no cells) */
        for (size_t i = 0; i < numPairs; i++) {
            dpx_pair_dev &pd = b->pairs[i];
            if (pd.m > 0 && pd.n > 0) { couples.push_back((int32_t)i); pd.lanes = 16; pd.rows = kLanesR; }
            else singles.push_back((int32_t)i);
        }
        /* a wave runs max(n + lanes) steps: neighbours of similar reference length, longest first (then longest query first).
         * Both keys are small integers: two stable counting passes instead of a comparison sort (12 ms per 100k pairs) */
        {
            std::vector<int32_t> tmp(couples.size());
            auto pass = [&](const std::vector<int32_t> &in, std::vector<int32_t> &out, int maxKey, auto key) {
                std::vector<uint32_t> cnt((size_t)maxKey + 2, 0);
                for (int32_t c : in) cnt[(size_t)(maxKey - key(c)) + 1]++; /* descending */
                for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
                for (int32_t c : in) out[cnt[(size_t)(maxKey - key(c))]++] = c;
            };
            pass(couples, tmp, b->maxM, [&](int32_t c) { return b->pairs[c].m; });
            pass(tmp, couples, b->maxN, [&](int32_t c) { return b->pairs[c].n; });
        }
        trace.mark("create: lanes sort");
        lanesPairs = couples.size();
        b->lanePacked = !couples.empty();
        if (b->lanePacked) {
            /* (reference area: 1 KiB keeps four workgroups of the 8-rows-per-lane kernels on a CU, up to 8 pairs per wave as in round 2; the
             * packed kernel's pairs take half the lanes, so its waves hold up to 12 pairs in 2 KiB: 2 x (18 + 2) KiB x 4 waves = 160 KiB) */
            lanesRefArea = pack_waves(b->pairs, couples, kLanesR, b->maxN, lanesPk ? 2048 : 1024, lanesPk ? DPX_WAVE_SLOTS : 8, waves); /* `couples` comes back in slot order */
            trace.mark("create: pack_waves");
            size_t lanesUsed = 0;
            for (const dpx_wave_desc &wd : waves) for (int k = 0; k < DPX_WAVE_SLOTS; k++) lanesUsed += wd.num[k];
            if (!lanesForced && lanesUsed * 100 < waves.size() * 64 * 85) b->lanePacked = false; /* under 85 % of the lanes own rows: not worth it */
        }
        b->lanesPk = b->lanePacked && lanesPk;
        if (!b->lanePacked) { /* back to the other kernels */
            b->R = R;
            singles.clear();
            couples.clear();
            waves.clear();
            lanesPairs = 0;
            for (size_t i = 0; i < numPairs; i++) { b->pairs[i].lanes = 64; b->pairs[i].rows = 0; }
        }
    }
    /* small batches need every wave they can get: one pair per wave there.  Measured with one-wave workgroups (round 3,
     * tools/pairs_sweep.py, profiles/r03/kernel_choice_by_batch_size.txt; GCUPS packed / one wave per pair / split): 1024 x 1024 at 1000
     * pairs 2014 / 2813 / 2615, 1500: 2971 / 2622 / 2487, 1900: 3052 / 2770 / 2650, 2500: 2749 / 2731 / 2683, 4000: 3226 / 2993 / 2650 --
     * the 16-rows-per-lane packed kernel wins from ~700 couples on; 512 x 512 (8 rows per lane) at 1500: 2235 / 2060 / 1987, 2000: 2432 /
     * 2495 / 2211, 3000: 2334 / 2744 / 2446, 4000: 2877 / 2811 / 2478 -- there only from ~2000 couples on */
    const size_t pkMinPairs = (linearAlgo && b->R == 16) ? 1400 : 4096;
    bool usePacked = !b->lanePacked && b->store && ((linearAlgo && dpx_tiled_stripes(b->maxM, b->R) == 1) || banded) &&
                     numPairs >= pkMinPairs;
    if (b->lanePacked) usePacked = false;
    else
    if (kn.packed >= 0) usePacked = kn.packed != 0 && b->store && (linearAlgo || banded);
    /* 16-bit wrapping arithmetic: only when weights and every intermediate provably fit (also under DPX_PACKED=1) */
    if (usePacked) {
        dpx_params kp = *params;
        kp.algo = kernelAlgo;
        usePacked = packed_safe(kp, b->maxM, b->maxN) && (!banded || params->gapOpen <= 0); /* (the packed band kernel's gap term saturates at 0) */
        /* 4-byte edge entries + 2-byte reference entries per wave: very long references do not fit the LDS twice over
         * (band kernel: 2-byte query and reference entries) */
        const size_t pkNeed = (banded ? align_up(((size_t)b->maxM + 96) * 2, 16) + align_up(((size_t)b->maxN + 32) * 2, 16)
                                      : align_up((size_t)(b->maxN + 2) * 4, 16) + align_up(((size_t)b->maxN + 128) * 2, 16)) * (DPX_FILL_THREADS / 64);
        if (pkNeed > 160u * 1024u) usePacked = false;
    }
    if (usePacked) {
        std::vector<int32_t> idx;
        idx.reserve(numPairs);
        for (size_t i = 0; i < numPairs; i++) {
            if (b->pairs[i].m > 0 && b->pairs[i].n > 0) idx.push_back((int32_t)i);
            else singles.push_back((int32_t)i);
        }
        if (ragged)
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) {
                const dpx_pair_dev &X = b->pairs[x], &Y = b->pairs[y];
                const uint64_t cx = (uint64_t)X.m * X.n, cy = (uint64_t)Y.m * Y.n;
                if (cx != cy) return cx > cy; /* longest first */
                if (X.m != Y.m) return X.m > Y.m;
                return X.n > Y.n;
            });
        for (size_t i = 0; i < idx.size();) {
            if (i + 1 < idx.size() && b->pairs[idx[i]].m == b->pairs[idx[i + 1]].m && b->pairs[idx[i]].n == b->pairs[idx[i + 1]].n) {
                couples.push_back(idx[i]);
                couples.push_back(idx[i + 1]);
                i += 2;
            } else {
                singles.push_back(idx[i]);
                i += 1;
            }
        }
        b->packed = !couples.empty();
    }
    /* Split path (small batches: fewer pairs than the chip has wave slots worth filling): one workgroup per pair, one wave per
     * stripe of 64 * R rows with R = 4 (2 for queries up to 256 rows), stripes concurrent (k_linear_split).  1000 pairs of
     * 512 x 512 become 2000 waves with half the dependent chain per step.  DPX_SPLIT=0/1 overrides. */
    {
        bool anyEmpty = false;
        for (size_t i = 0; i < numPairs && !anyEmpty; i++) anyEmpty = b->pairs[i].m <= 0 || b->pairs[i].n <= 0;
        /* (queries over 4096 rows would need more than 16 stripes: not split.  Twice the stripes at 2 rows per lane -- four waves per 512-row pair --
         * lose: 1000 x 512^2 0.146 vs 0.120 ms, profiles/r04/configs1_split_variants.txt: the kernel is bound by its instruction stream, not by waves) */
        const int sR = b->maxM > 256 ? 4 : 2;
        const int sW = dpx_tiled_stripes(b->maxM, sR);
        const size_t edgeStride = align_up((size_t)b->maxN + 2, 8); /* int16 elements */
        const size_t lds = 512 + align_up((size_t)b->maxN + 128 + 16, 16) + (size_t)std::max(sW - 1, 0) * edgeStride * 2;
        const bool shape = linearAlgo && b->store && !b->lanePacked && !b->packed && !anyEmpty && numPairs > 0 && sW >= 2 && sW <= 16 && lds <= 160u * 1024u;
        /* measured (bench.py --pairs N, DPX_SPLIT=0/1, HIP events around every fill; GCUPS split vs one wave per pair):
         *   1024 x 1024 (4 stripes of 4 rows per lane against ONE 16-rows-per-lane wave): 600 pairs 1864 vs 1768, 1200: 2229 vs 1765,
         *   2500: 2483 vs 2315, 3500: 2592 vs 2393, 4096: 2603 vs 2671 (and the packed kernel takes over);
         *   512 x 512 (2 stripes against one 8-rows-per-lane wave): 500 pairs 1021 vs 1003, 1000: 2084 vs 2003, 1500: 1841 vs 1935.
         * So: any shape up to ~1100 pairs, shapes of four or more stripes up to the packed kernel's threshold.
         * Round 3, with one-wave workgroups for the one-wave-per-pair kernels (tools/pairs_sweep.py; split vs one wave per pair): 1024 x 1024
         * 300 pairs 1373 vs 855, 600: 2111 vs 1699, 1000: 2615 vs 2805, 1500: 2487 vs 2622, 3000: 2636 vs 2779-2962; 512 x 512 600 pairs 1323 vs
         * 1173, 1000: 1943 vs 1932, 1500: 1987 vs 2060, 3000: 2446 vs 2744 -- queries that fit one stripe of the other kernels (<= 1024
         * rows) split only up to ~900 pairs (up to 512 rows: ~1100, a tie from there on); longer ones (the other kernels roll over several stripes) as before. */
        bool useSplit = shape && (b->maxM > 1024 ? (numPairs <= 1100 || (sW >= 4 && numPairs < 4096)) : numPairs <= (b->maxM > 512 ? 900u : 1100u));
        if (kn.split >= 0) useSplit = kn.split != 0 && shape;
        if (useSplit) {
            b->split = true;
            b->R = sR;
            b->splitWaves = sW;
            b->splitLds = lds;
            for (size_t i = 0; i < numPairs; i++) { b->pairs[i].lanes = 32; b->pairs[i].rows = (uint16_t)sR; }
            /* (round 3's packed variant of this kernel -- couples of equal-shaped pairs, two per workgroup on the VOP3P pipe -- halved the waves as
             * well as the instructions and lost wherever the split kernel is used: 1000 x 512^2 0.158 vs 0.119 ms, 2000 x 512^2 0.209 vs 0.206;
             * deleted in round 4, profiles/r04/configs1_split_variants.txt) */
        }
    }
    trace.mark("create: validate+geometry+launch lists");
    CREATE_TRY(stream_take(&b->stream));
    trace.mark("create: stream");
    int32_t *arenaOrder = nullptr, *arenaCouples = nullptr;
    /* only the bytes this batch's pairs touch go to the device (a driver that cuts one big file into batches hands the
     * whole file to every dpx_batch_create): upload [seqLo, seqHi) and rebase the device-side indices */
    size_t seqLo = numBytes, seqHi = 0;
    for (size_t i = 0; i < numPairs; i++) {
        const dpx_pair_dev &pd = b->pairs[i];
        seqLo = std::min(seqLo, (size_t)std::min(pd.refIdx, pd.qryIdx));
        seqHi = std::max(seqHi, std::max((size_t)pd.refIdx + (size_t)pd.n, (size_t)pd.qryIdx + (size_t)pd.m));
    }
    if (seqHi <= seqLo) seqLo = seqHi = 0;
    if (alphabet) seqLo &= ~(size_t)15; /* 2-bit input: the device expands whole dwords of 16 bases */
    for (size_t i = 0; i < numPairs; i++) { b->pairs[i].refIdx -= (int32_t)seqLo; b->pairs[i].qryIdx -= (int32_t)seqLo; }
    const size_t packedTotal = alphabet ? (numBytes + 3) / 4 : 0; /* bytes of the caller's packed buffer */
    sequences += alphabet ? seqLo / 4 : seqLo;
    numBytes = seqHi - seqLo;
    const size_t packedDwords = alphabet ? (numBytes + 15) / 16 : 0;
    const size_t packedCopy = alphabet ? std::min(packedDwords * 4, packedTotal - std::min(packedTotal, seqLo / 4)) : 0;
    char *dPacked = nullptr;
    size_t stageBytes = 0;
    {
        const size_t np1 = std::max<size_t>(numPairs, 1);
        const size_t szSeq = align_up(std::max<size_t>(std::max(numBytes, packedDwords * 16), 16), 256), szPairs = align_up(np1 * sizeof(dpx_pair_dev), 256);
        const size_t szPacked = alphabet ? align_up(std::max<size_t>(packedDwords * 4, 16), 256) : 0;
        const size_t szI32 = align_up(np1 * sizeof(int32_t), 256), szOff = align_up((np1 + 1) * sizeof(uint64_t), 256);
        const size_t szCouples = std::max(szI32, align_up(waves.size() * sizeof(dpx_wave_desc), 256)); /* couples, or the wave descriptors */
        const size_t szScan = align_up((dpx_out_scan_tiles(np1) + 1) * sizeof(uint64_t), 256);
        const size_t need = szSeq + szPairs + 5 * szI32 + szCouples + 2 * szOff + szScan + szPacked; /* score, endRow, endCol, order, couples, tbLen; tbOff, outOff, scan; 2-bit staging */
        CREATE_TRY(g_arenaCache.take((void **)&b->arena, need, &b->arenaCap));
        char *q = b->arena;
        b->dSeq = q;                  q += szSeq;
        b->dPairs = (dpx_pair_dev *)q; q += szPairs;
        b->dScore = (int32_t *)q;     q += szI32;
        b->dEndRow = (int32_t *)q;    q += szI32;
        b->dEndCol = (int32_t *)q;    q += szI32;
        b->dOrder = nullptr;          arenaOrder = (int32_t *)q; q += szI32;
        b->dCouples = nullptr;        arenaCouples = (int32_t *)q; q += szCouples;
        b->dTbLen = (int32_t *)q;     q += szI32;
        b->dTbOff = (uint64_t *)q;    q += szOff;
        b->dOutOff = (uint64_t *)q;   q += szOff;
        b->dOutScratch = (uint64_t *)q; q += szScan;
        dPacked = alphabet ? q : nullptr;
        /* Small batches (the class-per-pair drivers send 20 pairs per round trip, most tests a handful): everything the host uploads --
         * sequences, pair table, launch lists, line offsets -- lies in the arena's front, so it is assembled in ONE pinned image and sent
         * with ONE asynchronous copy on the batch's stream instead of five synchronous ones (~15 us each whatever the size: 70 -> 25 us
         * per create).  The fill is ordered behind it by the stream (a fill on a caller's stream waits for it, dpx_batch_fill). */
        const size_t front = (size_t)((char *)b->dTbOff - b->arena) + szOff;
        if (!alphabet && front <= ((size_t)256 << 10)) {
            CREATE_TRY(g_stageCache.take((void **)&b->hStage, align_up(front, (size_t)64 << 10), &b->hStageCap));
            stageBytes = front;
        }
    }
    /* host -> arena: into the image when there is one */
    auto upload = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        if (!b->hStage) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
        memcpy(b->hStage + ((char *)dst - b->arena), src, bytes);
        return hipSuccess;
    };
    trace.mark("create: arena");
    if (alphabet) { /* a quarter of the bytes over PCIe, expanded by k_unpack2 into the byte buffer the kernels read */
        if (packedCopy) CREATE_TRY(hipMemcpy(dPacked, sequences, packedCopy, hipMemcpyHostToDevice));
        const uint32_t alpha = (uint32_t)alphabet[0] | ((uint32_t)alphabet[1] << 8) | ((uint32_t)alphabet[2] << 16) | ((uint32_t)alphabet[3] << 24);
        CREATE_TRY(dpx_launch_unpack2(reinterpret_cast<const uint32_t *>(dPacked), alpha, b->dSeq, packedDwords, b->stream));
        CREATE_TRY(hipStreamSynchronize(b->stream)); /* (a fill may run on a caller's stream) */
        b->packed2 = true;
    } else if (numBytes) CREATE_TRY(upload(b->dSeq, sequences, numBytes));
    trace.mark("create: H2D sequences");
    if (b->lanePacked) {
        b->dCouples = arenaCouples;
        CREATE_TRY(upload(b->dCouples, waves.data(), waves.size() * sizeof(dpx_wave_desc)));
    } else if (b->packed) {
        b->dCouples = arenaCouples;
        CREATE_TRY(upload(b->dCouples, couples.data(), couples.size() * sizeof(int32_t)));
    } else if (ragged) {
        singles.resize(numPairs);
        std::iota(singles.begin(), singles.end(), 0);
    }
    if (!singles.empty()) {
        std::stable_sort(singles.begin(), singles.end(), [&](int32_t x, int32_t y) {
            return (uint64_t)b->pairs[x].m * b->pairs[x].n > (uint64_t)b->pairs[y].m * b->pairs[y].n;
        });
        b->dOrder = arenaOrder;
        CREATE_TRY(upload(b->dOrder, singles.data(), singles.size() * sizeof(int32_t)));
    }
    const size_t numSingles = (b->packed || b->lanePacked) ? singles.size() : numPairs;
    const size_t numCouples = couples.size() / 2;

    /* matrix placement (dpx_layout.h): pairs that are launched next to each other are interleaved chunk by chunk in
     * groups of `group` waves, so a group writes one compact moving window instead of `group` far-apart streams */
    if (b->store) {
        int group = 64;
        if (kn.group >= 1 && kn.group <= 1000000) group = kn.group;
        const uint32_t chunkElems = (banded || b->split) ? 512u : dpx_tiled_chunk_elems(b->R, b->planes);
        auto chunksOf = [&](const dpx_pair_dev &pd) -> uint64_t {
            if (pd.lanes == 32) return dpx_split_chunks(pd.m, pd.n, b->R);
            return banded ? dpx_band_chunks(pd.m, pd.n, params->band) : dpx_tiled_chunks(pd.m, pd.n, b->R);
        };
        uint64_t off = 0;
        auto place = [&](const std::vector<int32_t> &slots, size_t slotsPerGroup, uint32_t chunkElems) { /* slots in launch order */
            for (size_t s0 = 0; s0 < slots.size(); s0 += slotsPerGroup) {
                const size_t cnt = std::min(slotsPerGroup, slots.size() - s0);
                uint64_t maxChunks = 0;
                for (size_t g = 0; g < cnt; g++) maxChunks = std::max(maxChunks, chunksOf(b->pairs[slots[s0 + g]]));
                for (size_t g = 0; g < cnt; g++) {
                    dpx_pair_dev &pd = b->pairs[slots[s0 + g]];
                    pd.matOff = off + (uint64_t)g * chunkElems;
                    pd.chunkStride = (uint32_t)(cnt * chunkElems);
                }
                off += maxChunks * (uint64_t)cnt * chunkElems;
            }
        };
        if (b->packed) place(couples, (size_t)group * 2, chunkElems); /* one wave (workgroup) = two adjacent slots */
        else if (b->lanePacked) { /* tile layout (dpx_layout.h): every wave a contiguous stream of chunks, one per step; its pairs share the base */
            const uint32_t stepElems = dpx_wtile_step_elems(b->R / 8, b->planes);
            for (const dpx_wave_desc &wd : waves) {
                uint64_t steps = 0;
                for (int k = 0; k < DPX_WAVE_SLOTS; k++) {
                    if (!wd.num[k]) continue;
                    dpx_pair_dev &pd = b->pairs[wd.pair[k]];
                    pd.matOff = off;
                    pd.chunkStride = wd.first[k];
                    steps = std::max(steps, dpx_wtile_steps(pd.m, pd.n, b->R, wd.first[k]));
                }
                off += steps * (uint64_t)stepElems;
            }
        }
        if (b->packed || b->lanePacked || !singles.empty()) {
            place(singles, (size_t)group, chunkElems);
        } else { /* launch order == pair order */
            std::vector<int32_t> ident(numPairs);
            std::iota(ident.begin(), ident.end(), 0);
            place(ident, (size_t)group, chunkElems);
        }
        b->matElems = off;
    }
    trace.mark("create: launch lists+placement");
    if (numPairs) CREATE_TRY(upload(b->dPairs, b->pairs.data(), numPairs * sizeof(dpx_pair_dev)));
    if (b->store) { /* where k_traceback puts a pair's three lines (each with a dword-aligned capacity of m + n + 1): needed by the
                       output path, uploaded here so that dpx_batch_output_begin() never has to wait for the host */
        b->tbOff.resize(numPairs + 1);
        uint64_t off = 0;
        for (size_t i = 0; i < numPairs; i++) { b->tbOff[i] = off; off += 3ull * (uint64_t)((b->pairs[i].m + b->pairs[i].n + 1 + 3) & ~3); }
        b->tbOff[numPairs] = off;
        CREATE_TRY(upload(b->dTbOff, b->tbOff.data(), (numPairs + 1) * sizeof(uint64_t)));
    }
    if (b->hStage) { /* the one copy; the image stays untouched until dpx_batch_destroy() has waited for the stream */
        CREATE_TRY(hipMemcpyAsync(b->arena, b->hStage, stageBytes, hipMemcpyHostToDevice, b->stream));
        b->uploadPending = true;
    }
    if (b->store && b->matElems) {
        void *pool = nullptr;
        bool fresh = false;
        /* DPX_POOL_GUARD=1 (tests): 4 MiB behind the matrices are filled with a pattern here and checked by dpx_batch_sync() -- a kernel that
         * writes past the end of the batch's matrices fails the test instead of hitting whatever is mapped behind the pool */
        const bool guardOn = kn.poolGuard != 0;
        b->guardBytes = guardOn ? (size_t)4 << 20 : 0;
        CREATE_TRY(g_matCache.take(&pool, b->matElems * sizeof(int16_t) + b->guardBytes, &b->matPoolBytes, &fresh));
        if (b->guardBytes) CREATE_TRY(hipMemset((char *)pool + b->matElems * sizeof(int16_t), 0xA5, b->guardBytes));
        if (b->guardBytes && kn.poolGuard == 2) { /* (the checker's own test: one byte of the band is already wrong) */
            CREATE_TRY(hipMemset((char *)pool + b->matElems * sizeof(int16_t) + 12345, 0, 1));
            b->guardSelfTest = true;
        }
        /* DPX_TUNE_PLACEMENT (callers that fill the batch many times): the pool is timed with hipMemset and, if it is a fresh one,
         * shopped for with the batch's own fill at the end of this function (shop_pool_by_fill: four more allocations of the
         * pool's size).  DPX_POOL_PROBE=0 / 1 / 2 forces nothing / timing only / timing + shopping, whatever the flag says */
        bool tune = (flags & DPX_TUNE_PLACEMENT) != 0;
        int probeEnv = -1;
        if (kn.poolProbe >= 0) { probeEnv = kn.poolProbe; tune = probeEnv != 0; }
        PoolRecord rec;
        bool known = false;
        if (!fresh) {
            std::lock_guard<std::mutex> lk(g_poolRecMu);
            auto it = g_poolRecords.find(pool);
            if (it != g_poolRecords.end()) { rec = it->second; known = true; }
        }
        if (!known) rec = fresh_pool_record(pool, b->matPoolBytes);
        b->tunePool = tune && rec.candidatesMs.empty() && b->matPoolBytes >= ((size_t)1 << 30) && !b->guardBytes; /* (the memset probe would wipe the band) */
        /* (shopping for a pool nobody has shopped for yet: a fresh one, or one that dpx_pool_reserve built ahead of time) */
        b->tuneShop = b->tunePool && (fresh || rec.fillMs.empty()) && probeEnv != 1; /* DPX_POOL_PROBE=1: time only, no shopping */
        if (b->tunePool) rec.candidatesMs.assign(1, time_memset(pool, b->matPoolBytes, b->stream));
        b->dMat = (int16_t *)pool;
        b->poolRec = rec;
        { std::lock_guard<std::mutex> lk(g_poolRecMu); g_poolRecords[pool] = rec; }

    }
    trace.mark("create: H2D pairs+matrix pool");
#undef CREATE_TRY

    dpx_fill_args &a = b->args;
    a.seq = b->dSeq;
    a.pairs = b->dPairs;
    a.order = b->dOrder;
    a.numPairs = (int32_t)numSingles;
    a.wavesPerBlock = fill_waves_per_block(numSingles);
    a.match = params->match; a.mismatch = params->mismatch;
    a.gapOpen = params->gapOpen; a.gapExtend = params->gapExtend; a.band = params->band;
    a.mat = b->dMat;
    a.score = b->dScore; a.endRow = b->dEndRow; a.endCol = b->dEndCol;
    a.ldsPerWave = (uint32_t)perWave;
    a.ldsEdge2Off = (uint32_t)edgeBytes;
    a.ldsRefOff = banded ? (uint32_t)qBytes : (uint32_t)(edgeBytes * nEdges);
    a.ldsQryOff = banded ? 0u : (uint32_t)(edgeBytes * nEdges + refBytes);
    a.ldsBufStride = 0;
    a.rowTags = 0;
    /* big batches are bound by the bytes they write: their ramp steps store only the lines that hold cells (6 % fewer bytes at
     * 1024 x 1024: +3 % LSW, +5 % LNW); small ones are bound by step latency and keep the cheaper whole-chunk stores */
    a.rampLines = numPairs >= 2048 ? 1 : 0;
    if (kn.rampLines >= 0) a.rampLines = kn.rampLines != 0;
    if (b->split) { /* [control 512 B][staged reference][edge rows, one per stripe boundary] */
        a.ldsRefOff = 512u;
        a.ldsQryOff = (uint32_t)(512 + align_up((size_t)b->maxN + 128 + 16, 16));
        a.ldsBufStride = (uint32_t)align_up((size_t)b->maxN + 2, 8);
    }
    if (b->packed && banded) { /* packed band kernel: 2-byte query and reference entries (two chars each) */
        dpx_fill_args &k = b->pkArgs;
        k = a;
        k.order = b->dCouples;
        k.numPairs = (int32_t)numCouples;
        k.wavesPerBlock = fill_waves_per_block(numCouples);
        const size_t q2 = align_up(((size_t)b->maxM + 96) * 2, 16), r2 = align_up(((size_t)b->maxN + 32) * 2, 16);
        k.ldsPerWave = (uint32_t)(q2 + r2);
        k.ldsRefOff = (uint32_t)q2;
        b->pkLdsBytes = (q2 + r2) * (DPX_FILL_THREADS / 64);
    } else if (b->packed) { /* packed kernel: 4-byte edge entries (two int16), 2-byte reference entries (two chars) */
        dpx_fill_args &k = b->pkArgs;
        k = a;
        k.order = b->dCouples;
        k.numPairs = (int32_t)numCouples;
        k.wavesPerBlock = fill_waves_per_block(numCouples);
        const size_t pkEdge = align_up((size_t)(b->maxN + 2) * 4, 16);
        const size_t pkRef = align_up(((size_t)b->maxN + 128) * 2, 16);
        k.ldsPerWave = (uint32_t)(pkEdge + pkRef);
        k.ldsRefOff = (uint32_t)pkEdge;
        b->pkLdsBytes = (pkEdge + pkRef) * (DPX_FILL_THREADS / 64);
        if (b->pkLdsBytes > 160u * 1024u) { dpx_batch_destroy(b); return DPX_ERR_UNSUPPORTED; }
        /* SW start cell: one (score, row-in-lane, column) key per pair and lane where score * R + R-1 provably fits 16 bits
         * (1024 x 1024 at match 3: 3072 * 16 + 15), else one (score, column) key per pair and row (32 more registers at R = 16) */
        {
            auto pos = [](long long v) { return v > 0 ? v : 0; };
            const long long top = pos(std::max<long long>(params->match, params->mismatch)) * std::min<long long>(b->maxM, b->maxN) +
                                  pos(params->gapOpen) * ((long long)b->maxM + b->maxN); /* fits_int16()'s bound on H */
            k.rowTags = (kernelAlgo == DPX_ALGO_LSW && top * b->R + b->R - 1 <= 65535 && params->gapOpen <= 0) ? 1 : 0; /* (gap <= 0: the saturating gap term) */
            if (kn.rowTags >= 0) k.rowTags = (kn.rowTags != 0 && k.rowTags) ? 1 : 0; /* (tests: 0 = the per-row keys) */
        }
    }
    if (b->lanePacked) { /* per wave: the line stage of the writeback (dpx_kernels.hip: LineStage) + the staged references of its pairs */
        dpx_fill_args &k = b->pkArgs;
        k = a;
        k.order = nullptr;
        k.waves = reinterpret_cast<const dpx_wave_desc *>(b->dCouples);
        k.numPairs = (int32_t)waves.size();
        k.ldsPerWave = (uint32_t)(dpx_lanes_stage_bytes(kernelAlgo, kLanesR, b->store) + lanesRefArea);
        /* (the lane-packed kernels keep their four-wave workgroups at every size: 20 000 short reads 131-148 us against 150-153 with one-wave
         * workgroups, 100 000 the same; DPX_WPB=1 forces the latter) */
        k.wavesPerBlock = kernelAlgo == DPX_ALGO_ANW ? 1u : (kn.wavesPerBlock == 1 ? 1u : (uint32_t)dpx_lanes_waves_per_block(kernelAlgo));
        b->pkLdsBytes = (size_t)k.ldsPerWave * (size_t)k.wavesPerBlock;
        if (b->pkLdsBytes > 160u * 1024u) { dpx_batch_destroy(b); return DPX_ERR_UNSUPPORTED; }
    }
    if ((b->packed || b->lanePacked) && numSingles > 0) {
        /* more than one kernel per fill: a side stream + fork/join events (failure here only costs the overlap) */
        if (stream_take(&b->sideStream) != hipSuccess) { b->sideStream = nullptr; (void)hipGetLastError(); }
        if (b->sideStream && (hipEventCreateWithFlags(&b->evFork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&b->evJoin, hipEventDisableTiming) != hipSuccess)) (void)hipGetLastError();
    }
    b->nSingles = numSingles;
    b->nCouples = b->packed ? numCouples : 0;
    b->nLanePairs = b->lanePacked ? lanesPairs : 0;
    b->nWaves = b->lanePacked ? waves.size() : 0;
    if (b->tuneShop && b->dMat && !b->guardBytes) {
        shop_pool_by_fill(b, b->poolRec, trace);
        std::lock_guard<std::mutex> lk(g_poolRecMu);
        g_poolRecords[b->dMat] = b->poolRec;
        trace.mark("create: pool shopped by fill");
    }
    *out = b;
    return DPX_OK;
}

/* One fill = up to two kernels over disjoint pairs (couples or lane-packed waves + leftover singles).  They do not depend
 * on each other, but launches on one stream run one after the other -- and a small one (a few hundred leftover waves on
 * 1024 SIMDs) then costs a latency-bound tail of its own.
 * The secondary kernels therefore go to the batch's side stream between a fork and a join event; on `s` the fill still
 * looks like one operation (events recorded on `s` around it time all of it). */
static hipError_t launch_all(dpx_batch *b, hipStream_t s) {
    const bool hasMain = b->args.numPairs > 0;
    int kernels = 0;
    if (b->packed) kernels++;
    if (b->lanePacked) kernels++;
    if (hasMain) kernels++;
    hipStream_t side = s;
    bool forked = false;
    if (kernels > 1 && b->sideStream && b->evFork && b->evJoin) {
        hipError_t e = hipEventRecord(b->evFork, s);
        if (e == hipSuccess) e = hipStreamWaitEvent(b->sideStream, b->evFork, 0);
        if (e != hipSuccess) return e;
        side = b->sideStream;
        forked = true;
    }
    hipError_t e = hipSuccess;
    /* secondary kernel first (it is the short one; the main kernel then fills the chip around it) */
    if (b->lanePacked) {
        if (e == hipSuccess && hasMain) e = dpx_launch_fill(b->args, b->kernelAlgo, b->R, b->store, b->ldsBytes, side); /* empty pairs */
        if (e == hipSuccess) e = b->lanesPk ? dpx_launch_fill_lanes_packed(b->pkArgs, b->kernelAlgo, b->pkLdsBytes, s)
                                            : dpx_launch_fill_lanes(b->pkArgs, b->kernelAlgo, b->R, b->store, b->pkLdsBytes, s);
    } else if (b->packed) {
        if (e == hipSuccess && hasMain) e = dpx_launch_fill(b->args, b->kernelAlgo, b->R, b->store, b->ldsBytes, side);
        if (e == hipSuccess) e = b->kernelAlgo == DPX_ALGO_BSW ? dpx_launch_banded_packed(b->pkArgs, b->R, b->pkLdsBytes, s)
                                                               : dpx_launch_fill_packed(b->pkArgs, b->kernelAlgo, b->R, b->pkLdsBytes, s);
    } else if (b->split) {
        e = dpx_launch_fill_split(b->args, b->kernelAlgo, b->R, b->splitWaves, b->splitLds, s);
    } else {
        e = dpx_launch_fill(b->args, b->kernelAlgo, b->R, b->store, b->ldsBytes, s);
    }
    if (forked) {
        hipError_t j = hipEventRecord(b->evJoin, side);
        if (j == hipSuccess) j = hipStreamWaitEvent(s, b->evJoin, 0);
        if (e == hipSuccess) e = j;
    }
    return e;
}

int dpx_batch_fill(dpx_batch *b, void *stream) {
    if (!b) return DPX_ERR_INVALID;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    if (b->uploadPending) { /* the batch's inputs travel on b->stream (dpx_batch_create, small batches): a caller's stream waits for them once */
        if (s != b->stream) HIP_TRY(hipStreamSynchronize(b->stream));
        b->uploadPending = false;
    }
    const bool timed = (b->flags & DPX_TIME_FILLS) != 0;
    if (timed) {
        if (!b->evT0) HIP_TRY(hipEventCreate(&b->evT0));
        if (!b->evT1) HIP_TRY(hipEventCreate(&b->evT1));
        HIP_TRY(hipEventRecord(b->evT0, s));
    }
    HIP_TRY(launch_all(b, s));
    if (timed) { HIP_TRY(hipEventRecord(b->evT1, s)); b->fillTimed = true; }
    b->lastStream = s;
    b->filled = true;
    b->tbLinesValid = false;
    b->outState = 0;
    return DPX_OK;
}

int dpx_batch_last_fill_usec(dpx_batch *b, double *usec) {
    if (!b || !usec) return DPX_ERR_INVALID;
    if (!b->fillTimed) return DPX_ERR_NOT_FILLED; /* no fill yet, or the batch was created without DPX_TIME_FILLS */
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    HIP_TRY(hipEventSynchronize(b->evT1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b->evT0, b->evT1));
    *usec = (double)ms * 1000.0;
    return DPX_OK;
}

int dpx_batch_last_output_usec(dpx_batch *b, double *usec) {
    if (!b || !usec) return DPX_ERR_INVALID;
    if (!b->outTimed) return DPX_ERR_NOT_FILLED; /* no dpx_batch_output_begin() yet, or the batch was created without DPX_TIME_FILLS */
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    HIP_TRY(hipEventSynchronize(b->evOut1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b->evOut0, b->evOut1));
    *usec = (double)ms * 1000.0;
    return DPX_OK;
}

int dpx_batch_fill_timed(dpx_batch *b, int repeats, double *usecPerFill) {
    if (!b || repeats < 1 || !usecPerFill) return DPX_ERR_INVALID;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, b->stream);
    for (int i = 0; i < repeats && e == hipSuccess; i++) e = launch_all(b, b->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, b->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0); /* on every path */
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return hip_fail(e, "dpx_batch_fill_timed");
    *usecPerFill = (double)ms * 1000.0 / repeats;
    b->lastStream = b->stream;
    b->filled = true;
    b->tbLinesValid = false;
    b->outState = 0;
    return DPX_OK;
}

/* DPX_POOL_GUARD: nothing may have written behind the matrices (checked by dpx_batch_sync, dpx_batch_results and -- fatally, so that a
 * whole test run under the knob cannot miss it -- by dpx_batch_destroy) */
static int check_guard(dpx_batch *b) {
    if (!b->guardBytes || !b->dMat) return DPX_OK;
    std::vector<unsigned char> h(b->guardBytes);
    HIP_TRY(hipMemcpy(h.data(), (const char *)b->dMat + b->matElems * sizeof(int16_t), b->guardBytes, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < h.size(); k++)
        if (h[k] != 0xA5) { t_err = "DPX_POOL_GUARD: byte " + std::to_string(k) + " behind the matrices was overwritten"; return DPX_ERR_HIP; }
    return DPX_OK;
}

int dpx_batch_sync(dpx_batch *b) {
    if (!b) return DPX_ERR_INVALID;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    if (b->lastStream && b->lastStream != b->stream) HIP_TRY(hipStreamSynchronize(b->lastStream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return check_guard(b);
}

int dpx_batch_device_results(dpx_batch *b, void **dScores, void **dEndRow, void **dEndCol) {
    if (!b) return DPX_ERR_INVALID;
    if (dScores) *dScores = b->dScore;
    if (dEndRow) *dEndRow = b->dEndRow;
    if (dEndCol) *dEndCol = b->dEndCol;
    return DPX_OK;
}

int dpx_batch_results(dpx_batch *b, int32_t *scores, int32_t *endRow, int32_t *endCol) {
    if (!b) return DPX_ERR_INVALID;
    if (!b->filled) return DPX_ERR_NOT_FILLED;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    if (b->lastStream && b->lastStream != b->stream) HIP_TRY(hipStreamSynchronize(b->lastStream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    const size_t bytes = b->numPairs * sizeof(int32_t);
    const size_t stride = (size_t)((const char *)b->dEndRow - (const char *)b->dScore); /* the three arrays follow each other in the arena */
    if (bytes && b->resultsCopied && b->outState != 0 && b->hMeta) { /* dpx_batch_output_begin() brought them along (small batches) */
        const char *tail = b->hMeta + (b->numPairs + 1) * sizeof(uint64_t) + b->numPairs * sizeof(int32_t) + 16;
        if (scores) memcpy(scores, tail, bytes);
        if (endRow) memcpy(endRow, tail + stride, bytes);
        if (endCol) memcpy(endCol, tail + 2 * stride, bytes);
        return check_guard(b);
    }
    if (bytes && scores && endRow && endCol && (const char *)b->dEndCol - (const char *)b->dEndRow == (ptrdiff_t)stride && 3 * stride <= 12288) {
        /* small batches (the class-per-pair drivers: 20 pairs per round trip): one copy instead of three -- a synchronous copy costs
         * ~15 us whatever its size */
        char tmp[12288];
        HIP_TRY(hipMemcpy(tmp, b->dScore, 2 * stride + bytes, hipMemcpyDeviceToHost));
        memcpy(scores, tmp, bytes);
        memcpy(endRow, tmp + stride, bytes);
        memcpy(endCol, tmp + 2 * stride, bytes);
    } else if (bytes) {
        if (scores) HIP_TRY(hipMemcpy(scores, b->dScore, bytes, hipMemcpyDeviceToHost));
        if (endRow) HIP_TRY(hipMemcpy(endRow, b->dEndRow, bytes, hipMemcpyDeviceToHost));
        if (endCol) HIP_TRY(hipMemcpy(endCol, b->dEndCol, bytes, hipMemcpyDeviceToHost));
    }
    return check_guard(b);
}

int dpx_batch_matrix(dpx_batch *b, size_t pair, int which, int16_t *out) {
    if (!b || !out || pair >= b->numPairs || which < 0 || which >= b->planes) return DPX_ERR_INVALID;
    if (!b->store) return DPX_ERR_NO_MATRIX;
    if (!b->filled) return DPX_ERR_NOT_FILLED;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    const dpx_pair_dev &pd = b->pairs[pair];
    const size_t total = (size_t)(pd.m + 1) * (size_t)(pd.n + 1);
    int16_t *dOut = nullptr;
    size_t dOutCap = 0;
    if (b->lastStream && b->lastStream != b->stream) HIP_TRY(hipStreamSynchronize(b->lastStream));
    HIP_TRY(g_tbDevCache.take((void **)&dOut, total * sizeof(int16_t), &dOutCap)); /* row-major scratch */
    hipError_t e = dpx_launch_export(b->dMat, pd, b->kernelAlgo, b->R, b->planes, which, b->prm.gapOpen, b->prm.gapExtend,
                                     b->prm.band, dOut, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dOut, total * sizeof(int16_t), hipMemcpyDeviceToHost);
    g_tbDevCache.park(dOut, dOutCap);
    if (e != hipSuccess) return hip_fail(e, "dpx_batch_matrix");
    return DPX_OK;
}

/* ---- result text: device traceback -> per-pair block lengths -> exclusive scan -> packed blocks -> D2H of the real bytes ----
 * (the reference's V15 packed variable-length strings, cuda/LNW/LinearNeedlemanWunschV15.cu:168-172,372-425; the blocks are
 * already formatted as c++/main.cpp prints them, so a driver writes a batch with one fwrite) */

/* stage 1, asynchronous on the batch's stream: kernels + D2H of the offsets / lengths */

static int output_begin(dpx_batch *b, uint64_t firstNumber) {
    PhaseTrace trace;
    const size_t np = b->numPairs;
    if (b->outState == 2 && b->outFirst == firstNumber) return DPX_OK; /* already on the host */
    if (b->outState == 1 && b->outFirst == firstNumber) return DPX_OK; /* already in flight */
    if (b->outState == 1) HIP_TRY(hipStreamSynchronize(b->stream));    /* a run with another numbering is in flight: let it finish */
    if (!b->dTb || !b->dOut || !b->hMeta) { /* buffers of the output path, on first use (a failed attempt is simply repeated) */
        const uint64_t lines = b->tbOff[np];
        if (!b->dTb) HIP_TRY(g_tbDevCache.take((void **)&b->dTb, (size_t)std::max<uint64_t>(lines, 16), &b->dTbCap));
        /* packed text, worst case: every alignment m + n long, 20 digits of pair number, 11 of score */
        if (!b->dOut) HIP_TRY(g_tbDevCache.take((void **)&b->dOut, (size_t)(lines + 40ull * np + 16), &b->dOutCap));
        if (!b->hMeta) HIP_TRY(g_tbHostCache.take((void **)&b->hMeta, (np + 1) * sizeof(uint64_t) + np * sizeof(int32_t) + 16 + kSmallResultBytes, &b->hMetaCap));
        trace.mark("output: buffers");
    }
    uint64_t *hOff = reinterpret_cast<uint64_t *>(b->hMeta);
    int32_t *hLen = reinterpret_cast<int32_t *>(hOff + np + 1);
    if (b->lastStream && b->lastStream != b->stream) { /* the fill ran on a caller's stream: order behind it without blocking the host */
        if (!b->evOrder) HIP_TRY(hipEventCreateWithFlags(&b->evOrder, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(b->evOrder, b->lastStream));
        HIP_TRY(hipStreamWaitEvent(b->stream, b->evOrder, 0));
    }
    const bool timeOut = (b->flags & DPX_TIME_FILLS) != 0;
    if (timeOut) {
        if (!b->evOut0) HIP_TRY(hipEventCreate(&b->evOut0));
        if (!b->evOut1) HIP_TRY(hipEventCreate(&b->evOut1));
        HIP_TRY(hipEventRecord(b->evOut0, b->stream));
    }
    if (!b->tbLinesValid) {
        /* How to walk (round 4, tools/tb_kernel.sh, kernel time alone; profiles/r04/traceback_walks.txt).  Walk 2 = one WAVE per pair
         * (k_traceback_wave: runs of path steps decided by all lanes at once from an LDS window): LSW / LNW always -- 1000 x 512^2 0.13 vs
         * 0.60 ms for one lane per pair, 16 000 x 512^2 0.50 vs 0.76, 20 000 x 300^2 0.36 vs 0.54, 100 000 short reads 0.33 vs 0.57 (LSW) /
         * 0.71 vs 0.69 (LNW); ANW (three planes per window, 48-row banded windows) up to 20 000 pairs -- 1000 x 512^2 0.17 vs 1.16,
         * 5000 x 1024^2 0.86 vs 2.51, 20 000 x 300^2 0.92 vs 0.98, but 100 000 short reads 1.48 vs 0.98.  Walks 0 / 1 = one lane
         * per pair, cell by cell / through register-cached column vectors (the latter from 64k pairs on: enough lanes in flight to thrash
         * L1 / L2 between two steps of a lane).  Banded matrices: walk 2 as well (its band-layout window loads; numbers in the same file).
         * DPX_TB_WALK=0/1/2 forces one (tests). */
        int walk = b->numPairs >= 65536 ? 1 : 0;
        if (b->kernelAlgo == DPX_ALGO_LSW || b->kernelAlgo == DPX_ALGO_LNW || b->kernelAlgo == DPX_ALGO_BSW) walk = 2;
        else if (b->kernelAlgo == DPX_ALGO_ANW && b->numPairs <= 20000) walk = 2;
        { const int w = knobs().tbWalk; if (w >= 0) walk = std::min(2, w); }
        HIP_TRY(dpx_launch_traceback(b->args, (int)np, b->kernelAlgo, b->R, b->planes, walk, b->dTbOff, b->dTb, b->dTbLen, b->stream));
        b->tbLinesValid = true;
    }
    HIP_TRY(dpx_launch_output(b->dPairs, b->dScore, b->dTbLen, b->dTbOff, b->dTb, (int)np, (unsigned long long)firstNumber,
                              reinterpret_cast<unsigned long long *>(b->dOutScratch), reinterpret_cast<unsigned long long *>(b->dOutOff), b->dOut,
                              false, false, b->stream));
    if (timeOut) { HIP_TRY(hipEventRecord(b->evOut1, b->stream)); b->outTimed = true; }
    b->textCopied = false;
    b->resultsCopied = false;
    if (np) {
        HIP_TRY(hipMemcpyAsync(hOff, b->dOutOff, (np + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemcpyAsync(hLen, b->dTbLen, np * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
        /* small batches: the text follows at once, at its worst-case size -- dpx_batch_output_end() then waits once instead of waiting,
         * reading the real size and copying (a second round trip of ~15 us; the class-per-pair drivers pay it 200 times per 4000 pairs) */
        const size_t worst = (size_t)(b->tbOff[np] + 40ull * np + 16);
        if (worst <= ((size_t)256 << 10)) {
            if (!b->hOut || b->hOutCap < worst + 1) {
                g_tbHostCache.park(b->hOut, b->hOutCap);
                b->hOut = nullptr;
                HIP_TRY(g_tbHostCache.take((void **)&b->hOut, worst + 1, &b->hOutCap));
            }
            HIP_TRY(hipMemcpyAsync(b->hOut, b->dOut, worst, hipMemcpyDeviceToHost, b->stream));
            b->textCopied = true;
            /* ... and so do score / end row / end column (they lie side by side in the arena): dpx_batch_results() after this call is a
             * wait and three memcpy instead of a wait and a synchronous copy of its own */
            const size_t stride = (size_t)((const char *)b->dEndRow - (const char *)b->dScore), bytes = np * sizeof(int32_t);
            if ((const char *)b->dEndCol - (const char *)b->dEndRow == (ptrdiff_t)stride && 2 * stride + bytes <= kSmallResultBytes &&
                (!b->lastStream || b->lastStream == b->stream)) {
                char *tail = b->hMeta + (np + 1) * sizeof(uint64_t) + np * sizeof(int32_t) + 16;
                HIP_TRY(hipMemcpyAsync(tail, b->dScore, 2 * stride + bytes, hipMemcpyDeviceToHost, b->stream));
                b->resultsCopied = true;
            }
        }
    } else {
        hOff[0] = 0;
    }
    b->outState = 1;
    b->outFirst = firstNumber;
    trace.mark("output: launches");
    return DPX_OK;
}

/* stage 2: wait, then copy exactly the bytes the blocks occupy */
static int output_end(dpx_batch *b) {
    if (b->outState == 2) return DPX_OK;
    if (b->outState != 1) return DPX_ERR_INVALID;
    PhaseTrace trace;
    HIP_TRY(hipStreamSynchronize(b->stream));
    trace.mark("output: wait for device");
    const uint64_t total = reinterpret_cast<const uint64_t *>(b->hMeta)[b->numPairs];
    if (b->textCopied) { /* (the text came with the offsets) */
        b->hOut[total] = 0;
        b->hOutBytes = (size_t)total;
        b->outState = 2;
        trace.mark("output: D2H text");
        return DPX_OK;
    }
    if (!b->hOut || b->hOutCap < total + 1) {
        g_tbHostCache.park(b->hOut, b->hOutCap);
        b->hOut = nullptr;
        HIP_TRY(g_tbHostCache.take((void **)&b->hOut, (size_t)total + 1, &b->hOutCap));
        trace.mark("output: pinned text buffer");
    }
    if (total) {
        HIP_TRY(hipMemcpyAsync(b->hOut, b->dOut, (size_t)total, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    b->hOut[total] = 0;
    b->hOutBytes = (size_t)total;
    b->outState = 2;
    trace.mark("output: D2H text");
    return DPX_OK;
}

int dpx_batch_output_begin(dpx_batch *b, uint64_t firstPairNumber) {
    if (!b) return DPX_ERR_INVALID;
    if (!b->store) return DPX_ERR_NO_MATRIX;
    if (!b->filled) return DPX_ERR_NOT_FILLED;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    return output_begin(b, firstPairNumber);
}

int dpx_batch_output_end(dpx_batch *b, const char **text, size_t *bytes, const uint64_t **offsets) {
    if (!b) return DPX_ERR_INVALID;
    if (b->outState == 0) return DPX_ERR_NOT_FILLED; /* no dpx_batch_output_begin() since the last fill */
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    rc = output_end(b);
    if (rc != DPX_OK) return rc;
    if (text) *text = b->hOut;
    if (bytes) *bytes = b->hOutBytes;
    if (offsets) *offsets = reinterpret_cast<const uint64_t *>(b->hMeta);
    return DPX_OK;
}

/* pinned text buffers handed to the caller by dpx_batch_output_take(): pointer -> capacity, for dpx_text_free() */
namespace { struct Taken { void *ptr; size_t cap; int device; }; std::mutex g_takenMu; std::vector<Taken> g_taken; }

int dpx_batch_output_take(dpx_batch *b, char **text, size_t *bytes) {
    if (!b || !text) return DPX_ERR_INVALID;
    if (b->outState == 0) return DPX_ERR_NOT_FILLED;
    int rc = bind_device(b->device);
    if (rc != DPX_OK) return rc;
    rc = output_end(b);
    if (rc != DPX_OK) return rc;
    { std::lock_guard<std::mutex> lk(g_takenMu); g_taken.push_back(Taken{(void *)b->hOut, b->hOutCap, b->device}); }
    *text = b->hOut;
    if (bytes) *bytes = b->hOutBytes;
    b->hOut = nullptr; /* the batch no longer owns it; a later dpx_batch_traceback() / _output_end() rebuilds the text */
    b->hOutCap = 0;
    b->outState = 0;
    return DPX_OK;
}

int dpx_text_free(char *text) {
    if (!text) return DPX_OK;
    size_t cap = 0;
    int dev = -1;
    {
        std::lock_guard<std::mutex> lk(g_takenMu);
        for (size_t i = 0; i < g_taken.size(); i++)
            if (g_taken[i].ptr == (void *)text) { cap = g_taken[i].cap; dev = g_taken[i].device; g_taken.erase(g_taken.begin() + (long)i); break; }
    }
    if (!cap) return DPX_ERR_INVALID; /* not a buffer dpx_batch_output_take() handed out */
    if (dev >= 0) { (void)hipSetDevice(dev); t_device = dev; }
    g_tbHostCache.park(text, cap);
    return DPX_OK;
}

int dpx_batch_traceback(dpx_batch *b, size_t pair, char *refLine, char *relLine, char *qryLine, int32_t *len) {
    if (!b || pair >= b->numPairs) return DPX_ERR_INVALID;
    if (!b->store) return DPX_ERR_NO_MATRIX;
    if (!b->filled) return DPX_ERR_NOT_FILLED;
    if (b->outState != 2) { /* first call after a fill walks every pair of the batch on the device; later calls only copy lines */
        int rc = bind_device(b->device);
        if (rc != DPX_OK) return rc;
        rc = output_begin(b, b->outState == 1 ? b->outFirst : 0);
        if (rc == DPX_OK) rc = output_end(b);
        if (rc != DPX_OK) return rc;
    }
    const uint64_t *hOff = reinterpret_cast<const uint64_t *>(b->hMeta);
    const int32_t *hLen = reinterpret_cast<const int32_t *>(hOff + b->numPairs + 1);
    const int k = hLen[pair];
    /* the pair's block ends with its three lines, each k characters + '\n' */
    const char *lines = b->hOut + hOff[pair + 1] - 3 * (size_t)(k + 1);
    char *dst[3] = {refLine, relLine, qryLine};
    for (int l = 0; l < 3; l++)
        if (dst[l]) { memcpy(dst[l], lines + (size_t)l * (size_t)(k + 1), (size_t)k); dst[l][k] = 0; }
    if (len) *len = k;
    return DPX_OK;
}

int dpx_batch_describe(dpx_batch *b, char *buf, size_t cap) {
    if (!b || !buf || !cap) return DPX_ERR_INVALID;
    static const char *names[] = {"LNW", "LSW", "ANW", "BSW"};
    const char *kernel = b->kernelAlgo == DPX_ALGO_BSW ? (b->packed ? "k_banded_fill_pk" : "k_banded_fill") : b->kernelAlgo == DPX_ALGO_ANW ? (b->lanePacked ? "k_affine_lanes" : "k_affine_fill")
                         : b->packed ? "k_linear_fill_pk" : b->lanesPk ? "k_linear_lanes_pk" : b->lanePacked ? "k_linear_lanes" : b->split ? "k_linear_split" : "k_linear_fill";
    /* dtype = the arithmetic type of the kernel that fills (most of) the batch */
    int len = snprintf(buf, cap, "algo=%s kernel_algo=%s kernel=%s dtype=%s rows_per_lane=%d store=%d couples=%zu lane_pairs=%zu waves=%zu singles=%zu row_tags=%d seq_input=%s waves_per_workgroup=%u",
                       names[b->prm.algo], names[b->kernelAlgo], kernel, (b->packed || b->lanesPk) ? "int16" : "int32", b->R, b->store ? 1 : 0, b->nCouples, b->nLanePairs,
                       b->nWaves, b->nSingles, (int)b->pkArgs.rowTags, b->packed2 ? "packed2" : "bytes",
                       (b->packed || b->lanePacked) ? b->pkArgs.wavesPerBlock : b->split ? (unsigned)b->splitWaves : b->args.wavesPerBlock);
    if (b->dMat && len > 0 && (size_t)len < cap) { /* the matrix pool: how it was built, and the memset time of every candidate that was timed */
        const PoolRecord &r = b->poolRec;
        len += snprintf(buf + len, cap - (size_t)len, " pool=%s pool_bytes=%zu pool_chunk_mb=%zu pool_kept=%d pool_memset_ms=", r.mode.c_str(), b->matPoolBytes,
                        r.chunkBytes >> 20, r.kept);
        for (size_t k = 0; k < r.candidatesMs.size() && (size_t)len < cap; k++)
            len += snprintf(buf + len, cap - (size_t)len, "%s%.3f", k ? "," : "", r.candidatesMs[k]);
        if (r.candidatesMs.empty() && (size_t)len < cap) len += snprintf(buf + len, cap - (size_t)len, "untimed");
        if ((size_t)len < cap) len += snprintf(buf + len, cap - (size_t)len, " pool_fill_ms=");
        for (size_t k = 0; k < r.fillMs.size() && (size_t)len < cap; k++)
            len += snprintf(buf + len, cap - (size_t)len, "%s%.3f", k ? "," : "", r.fillMs[k]);
        if (r.fillMs.empty() && (size_t)len < cap) len += snprintf(buf + len, cap - (size_t)len, "unshopped");
        if (!r.kinds.empty() && (size_t)len < cap) {
            len += snprintf(buf + len, cap - (size_t)len, " pool_kinds=");
            for (size_t k = 0; k < r.kinds.size() && (size_t)len < cap; k++) len += snprintf(buf + len, cap - (size_t)len, "%s%s", k ? "," : "", r.kinds[k].c_str());
        }
        if (!r.ranges.empty() && (size_t)len < cap) {
            len += snprintf(buf + len, cap - (size_t)len, " pool_ranges=");
            for (size_t k = 0; k < r.ranges.size() && (size_t)len < cap; k++) len += snprintf(buf + len, cap - (size_t)len, "%s%s", k ? "," : "", r.ranges[k].c_str());
        }
    }
    return DPX_OK;
}

int dpx_batch_info(dpx_batch *b, size_t *numPairs, uint64_t *cells, uint64_t *matrixBytes, uint64_t *algorithmicBytes) {
    if (!b) return DPX_ERR_INVALID;
    if (numPairs) *numPairs = b->numPairs;
    if (cells) *cells = b->cells;
    if (matrixBytes) *matrixBytes = b->matElems * sizeof(int16_t);
    if (algorithmicBytes) *algorithmicBytes = b->algBytes;
    return DPX_OK;
}

/* ------------------------------------------------------------------------------------------ one-shot */

int dpx_align_batch(const dpx_params *params, const char *sequences, size_t numBytes, const dpx_seq_pair *pairs,
                    size_t numPairs, int32_t *scores, int32_t *endRow, int32_t *endCol, int16_t **H, int16_t **I,
                    int16_t **D) {
    const bool wantMat = H || I || D;
    dpx_batch *b = nullptr;
    int rc = dpx_batch_create(params, sequences, numBytes, pairs, 0, numPairs, wantMat ? DPX_KEEP_MATRICES : DPX_SCORE_ONLY, &b);
    if (rc != DPX_OK) return rc;
    rc = dpx_batch_fill(b, nullptr);
    if (rc == DPX_OK) rc = dpx_batch_results(b, scores, endRow, endCol);
    for (size_t p = 0; rc == DPX_OK && wantMat && p < numPairs; p++) {
        if (H && H[p]) rc = dpx_batch_matrix(b, p, DPX_MAT_H, H[p]);
        if (rc == DPX_OK && I && I[p] && b->planes == 3) rc = dpx_batch_matrix(b, p, DPX_MAT_I, I[p]);
        if (rc == DPX_OK && D && D[p] && b->planes == 3) rc = dpx_batch_matrix(b, p, DPX_MAT_D, D[p]);
    }
    dpx_batch_destroy(b);
    return rc;
}

/* ------------------------------------------------------------------------------------------ primitives */

int dpx_prim_eval(const int32_t *op, const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t count,
                  uint32_t *result, uint32_t *pred) {
    if (count && (!op || !a || !b || !c || !result || !pred)) return DPX_ERR_INVALID;
    int rc = bind_device();
    if (rc != DPX_OK) return rc;
    if (!count) return DPX_OK;
    void *d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t bytes = count * 4;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipMalloc(&d[i], bytes);
    if (e == hipSuccess) e = hipMemcpy(d[0], op, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d[1], a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d[2], b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d[3], c, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = dpx_launch_prim_eval((const int32_t *)d[0], (const uint32_t *)d[1], (const uint32_t *)d[2], (const uint32_t *)d[3],
                                 count, (uint32_t *)d[4], (uint32_t *)d[5], nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(result, d[4], bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(pred, d[5], bytes, hipMemcpyDeviceToHost);
    for (int i = 0; i < 6; i++) (void)hipFree(d[i]);
    if (e != hipSuccess) return hip_fail(e, "dpx_prim_eval");
    return DPX_OK;
}

} /* extern "C" */
