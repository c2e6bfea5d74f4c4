// tools/poolstudy.hip -- round 3: WHY is the same 22 GB matrix pool written at 6.3 TB/s by one allocation and at 6.05 TB/s by the
// next (profiles/r02/pool_placement_probe.txt)?  One process, one box; every variant is timed with hipMemset and with a plain
// streaming-store kernel, as a whole and GiB by GiB (is the slow mode spread over the pool or local to some of it?):
//   malloc    one hipMalloc of the pool, several times over (freed in between / held side by side)
//   vmm S     one virtual range (hipMemAddressReserve) backed by physical chunks of S bytes (hipMemCreate + hipMemMap)
//   chunks    separate hipMallocs of 1 GiB
// development aid: build with  hipcc --offload-arch=gfx950 -O2 tools/poolstudy.hip -o tools/bin/poolstudy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_store(u4 *p, size_t n16) {
    u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static hipEvent_t ea, eb;
template <class F> static float timeit(F f, int reps, int tries = 3) {
    float best = 1e30f;
    for (int k = 0; k < tries; k++) {
        CK(hipEventRecord(ea));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        best = std::min(best, ms / reps);
    }
    return best;
}

static const size_t GiB = (size_t)1 << 30;

// whole-pool memset / store rates + the per-GiB memset profile of [p, p + bytes)
static void measure(const char *tag, char *p, size_t bytes) {
    CK(hipMemset(p, 0, bytes)); CK(hipDeviceSynchronize());
    const float msSet = timeit([&] { CK(hipMemsetAsync(p, 0, bytes, 0)); }, 2);
    const float msSt = timeit([&] { hipLaunchKernelGGL(k_store, dim3(4096), dim3(256), 0, 0, (u4 *)p, bytes / 16); }, 2);
    std::vector<float> sl;
    for (size_t off = 0; off + GiB <= bytes; off += GiB)
        sl.push_back(timeit([&] { CK(hipMemsetAsync(p + off, 0, GiB, 0)); }, 4, 2));
    std::vector<float> s2 = sl; std::sort(s2.begin(), s2.end());
    const float med = s2[s2.size() / 2];
    printf("%-18s %p  memset %.3f ms %.2f TB/s | store %.3f ms %.2f TB/s | GiB slices: min %.1f med %.1f max %.1f us  [", tag, (void *)p,
           msSet, bytes / msSet / 1e9, msSt, bytes / msSt / 1e9, s2.front() * 1e3, med * 1e3, s2.back() * 1e3);
    for (float v : sl) printf("%c", v > med * 1.06f ? 'S' : v > med * 1.03f ? 's' : v < med * 0.97f ? 'f' : '.');
    printf("]\n");
    fflush(stdout);
}

struct Vmm { char *va = nullptr; size_t bytes = 0, chunk = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static bool vmm_make(Vmm &v, size_t bytes, size_t chunk, int dev) {
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("granularity query failed\n"); return false; }
    chunk = (chunk + gran - 1) / gran * gran;
    v.chunk = chunk;
    v.bytes = (bytes + chunk - 1) / chunk * chunk;
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, v.bytes, 0, nullptr, 0);
    if (e != hipSuccess) { printf("hipMemAddressReserve: %s\n", hipGetErrorString(e)); return false; }
    v.va = (char *)va;
    for (size_t off = 0; off < v.bytes; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) { printf("hipMemCreate: %s\n", hipGetErrorString(e)); return false; }
        e = hipMemMap(v.va + off, chunk, 0, h, 0);
        if (e != hipSuccess) { printf("hipMemMap: %s\n", hipGetErrorString(e)); return false; }
        v.h.push_back(h);
    }
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(v.va, v.bytes, &acc, 1);
    if (e != hipSuccess) { printf("hipMemSetAccess: %s\n", hipGetErrorString(e)); return false; }
    return true;
}
static void vmm_free(Vmm &v) {
    if (!v.va) return;
    (void)hipMemUnmap(v.va, v.bytes);
    for (auto h : v.h) (void)hipMemRelease(h);
    (void)hipMemAddressFree(v.va, v.bytes);
    v = Vmm();
}

int main(int argc, char **argv) {
    const size_t want = argc > 1 ? (size_t)atof(argv[1]) : (size_t)22261760000ull;
    const size_t bytes = (want + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("pool %.3f GB; device memory free %.1f / %.1f GB; VMM granularity min %zu recommended %zu\n", bytes / 1e9, fr / 1e9, tot / 1e9, gmin, grec);
    // warm the clocks
    { char *w; CK(hipMalloc(&w, GiB)); for (int i = 0; i < 200; i++) CK(hipMemsetAsync(w, 0, GiB, 0)); CK(hipDeviceSynchronize()); CK(hipFree(w)); }

    printf("-- malloc, freed in between --\n");
    for (int i = 0; i < 6; i++) { char *p; CK(hipMalloc(&p, bytes)); char tag[32]; snprintf(tag, sizeof tag, "malloc seq %d", i); measure(tag, p, bytes); CK(hipFree(p)); }
    printf("-- malloc, behind a dummy of varying size --\n");
    for (int i = 0; i < 4; i++) {
        char *d; const size_t ds = (size_t)(i + 1) * 1234567 * 512; CK(hipMalloc(&d, ds));
        char *p; CK(hipMalloc(&p, bytes)); char tag[32]; snprintf(tag, sizeof tag, "malloc dummy %d", i); measure(tag, p, bytes); CK(hipFree(p)); CK(hipFree(d));
    }
    printf("-- malloc, four held side by side --\n");
    { char *p[4]; for (int i = 0; i < 4; i++) CK(hipMalloc(&p[i], bytes));
      for (int i = 0; i < 4; i++) { char tag[32]; snprintf(tag, sizeof tag, "malloc held %d", i); measure(tag, p[i], bytes); }
      for (int i = 0; i < 4; i++) { char tag[32]; snprintf(tag, sizeof tag, "malloc held %d again", i); measure(tag, p[i], bytes); }
      for (int i = 0; i < 4; i++) CK(hipFree(p[i])); }
    printf("-- VMM: one virtual range, physical chunks of S --\n");
    for (size_t chunk : {(size_t)4 * GiB, GiB, GiB / 4, GiB / 16, GiB / 64}) {
        for (int rep = 0; rep < 2; rep++) {
            Vmm v;
            if (vmm_make(v, bytes, chunk, 0)) { char tag[32]; snprintf(tag, sizeof tag, "vmm %zu MiB #%d", v.chunk >> 20, rep); measure(tag, v.va, bytes); }
            vmm_free(v);
        }
    }
    printf("-- 1 GiB hipMallocs (not contiguous): per-chunk memset --\n");
    for (int rep = 0; rep < 2; rep++) {
        std::vector<char *> c(bytes / GiB);
        for (auto &p : c) CK(hipMalloc(&p, GiB));
        std::vector<float> t;
        for (auto p : c) t.push_back(timeit([&] { CK(hipMemsetAsync(p, 0, GiB, 0)); }, 4, 2));
        const float all = timeit([&] { for (auto p : c) CK(hipMemsetAsync(p, 0, GiB, 0)); }, 1);
        std::vector<float> s2 = t; std::sort(s2.begin(), s2.end());
        printf("chunks #%d: all %zu GiB %.3f ms %.2f TB/s | per chunk min %.1f med %.1f max %.1f us\n", rep, c.size(), all, c.size() * (double)GiB / all / 1e9,
               s2.front() * 1e3, s2[s2.size() / 2] * 1e3, s2.back() * 1e3);
        for (auto p : c) CK(hipFree(p));
    }
    printf("-- malloc again at the end --\n");
    for (int i = 0; i < 3; i++) { char *p; CK(hipMalloc(&p, bytes)); char tag[32]; snprintf(tag, sizeof tag, "malloc end %d", i); measure(tag, p, bytes); CK(hipFree(p)); }
    return 0;
}
