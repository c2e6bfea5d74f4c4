// dpx_main.cpp -- batched GPU driver of the MI355X engine, shaped like the reference's CUDA mains
// (cuda/LNW/LinearNeedlemanWunschV19.cu:357-680, cuda/LinearSmithWaterman.cu:172-435): parse the pairs file,
// move sequences + seqPair[] to the device once, fill batch after batch, trace back on the device, and print
// "<pair> | <score>" + three lines per pair in input order.  The host prints batch k-1 while the GPU works on
// batch k (the software pipeline of V19.cu:546-579).  stdout keeps the reference's lines so logs stay diff-able.
//
//   dpx_main -pairs <file> [-match 3] [-mismatch -1] [-open -2 | -gap -2] [-extend -1]
//            [-algo LSW|LNW|ANW|BSW] [-band 128] [-batch 10000] [-device 0] [-noprint] [-rank r -world w]
//
// Multi-GPU: pairs are independent, so every GPU gets one process with its own contiguous shard of the file:
// `-rank r -world w` aligns only pairs [ceil(N/w)*r, ceil(N/w)*(r+1)) (the same split as shard.py / bench.py) and
// prints them with their global pair numbers; tools/run_multi_gpu.sh starts one rank per device and concatenates
// the outputs in rank order, which is input order.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dpx_align.h"
#include "parseInput.h"
#include "timing.h"

namespace {

struct BatchOut { // one text arena per batch instead of three std::strings per pair (300k small allocations per 100k pairs)
    size_t first = 0;
    std::vector<int32_t> score, len;
    std::vector<size_t> off; // lines of pair k: text[off[k] + l * (len[k] + 1)], l = 0 (reference), 1 (relation), 2 (query)
    std::unique_ptr<char[]> text; // not a vector: no zero-fill of tens of MB that are overwritten anyway
    size_t textCap = 0;
};

[[noreturn]] void die(const char *what, int rc) {
    printf("%s\nDPX ERROR: %s (%s)\n", what, dpx_strerror(rc), dpx_last_error());
    exit(1);
}

void print_batch(const BatchOut *o, bool local) {
    for (size_t k = 0; k < o->score.size(); k++) {
        if (local && o->score[k] == 0) printf("%zu | 0\n\n\n\n", o->first + k);
        else {
            const char *t = o->text.get() + o->off[k];
            const size_t stride = (size_t)o->len[k] + 1;
            printf("%zu | %d\n%s\n%s\n%s\n", o->first + k, o->score[k], t, t + stride, t + 2 * stride);
        }
    }
}

} // namespace

int main(int argc, char *argv[]) {
    if (argc < 3) {
        fprintf(stderr, "usage: dpx_main -pairs <InSeqFile> -match <matchWeight> -mismatch <mismatchWeight> -open <gapWeight> "
                        "[-extend <gapExtend>] [-algo LSW|LNW|ANW|BSW] [-band <B>] [-batch <N>] [-device <D>] [-noprint]\n");
        exit(EXIT_FAILURE);
    }
    const char *pairFileName = nullptr;
    int match = 3, mismatch = -1, gapOpen = -2, gapExtend = -1, band = 128, device = 0, rank = 0, world = 1;
    size_t batchSize = 10000; // BATCH_SIZE of the reference's final version (V19.cu:9)
    bool print = true;
    std::string algoName = "LSW";
    for (int i = 1; i < argc; i++) {
        auto next = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", flag); exit(EXIT_FAILURE); }
            return argv[++i];
        };
        if (!strcmp(argv[i], "-pairs")) pairFileName = next("-pairs");
        else if (!strcmp(argv[i], "-match")) match = atoi(next("-match"));
        else if (!strcmp(argv[i], "-mismatch")) mismatch = atoi(next("-mismatch"));
        else if (!strcmp(argv[i], "-open") || !strcmp(argv[i], "-gap")) gapOpen = atoi(next("-open"));
        else if (!strcmp(argv[i], "-extend")) gapExtend = atoi(next("-extend"));
        else if (!strcmp(argv[i], "-algo")) algoName = next("-algo");
        else if (!strcmp(argv[i], "-band")) band = atoi(next("-band"));
        else if (!strcmp(argv[i], "-batch")) batchSize = (size_t)atoll(next("-batch"));
        else if (!strcmp(argv[i], "-device")) device = atoi(next("-device"));
        else if (!strcmp(argv[i], "-noprint")) print = false;
        else if (!strcmp(argv[i], "-rank")) rank = atoi(next("-rank"));
        else if (!strcmp(argv[i], "-world")) world = atoi(next("-world"));
        else { fprintf(stderr, "unknown argument: %s\n", argv[i]); exit(EXIT_FAILURE); }
    }
    if (!pairFileName || batchSize == 0) { fprintf(stderr, "need -pairs <file>\n"); exit(EXIT_FAILURE); }
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad -rank/-world\n"); exit(EXIT_FAILURE); }
    const int algo = algoName == "LNW" ? DPX_ALGO_LNW : algoName == "LSW" ? DPX_ALGO_LSW : algoName == "ANW" ? DPX_ALGO_ANW
                     : algoName == "BSW" ? DPX_ALGO_BSW : -1;
    if (algo < 0) { fprintf(stderr, "unknown -algo %s\n", algoName.c_str()); exit(EXIT_FAILURE); }
    const bool local = algo == DPX_ALGO_LSW || algo == DPX_ALGO_BSW;

    printf("[Device Details]\n");
    int deviceCount = 0;
    int rc = dpx_device_count(&deviceCount);
    if (rc != DPX_OK || deviceCount == 0) die("FAILED TO GET DEVICE COUNT", rc != DPX_OK ? rc : DPX_ERR_NO_DEVICE);
    printf("Device count: %d\n", deviceCount);
    if ((rc = dpx_init(device)) != DPX_OK) die("FAILED TO BIND DEVICE", rc);
    char name[256];
    int cus = 0;
    size_t hbm = 0;
    if ((rc = dpx_device_info(name, sizeof name, &cus, &hbm)) != DPX_OK) die("FAILED TO GET DEVICE PROPERTIES", rc);
    printf("Device %d is%s with %d compute units, %.0f GB.\n\n", device, name, cus, (double)hbm / 1e9);

    printf("Parsing input file: %s\n", pairFileName);
    seqPair *sequenceIdxs;
    char *sequences;
    // one process per GPU: every rank maps the file and materialises only its own contiguous ceil(N/world)-sized shard
    size_t shardFirst = 0, totalPairs = 0;
    inputInfo fileInfo = world > 1 ? parseInputShard(pairFileName, rank, world, sequenceIdxs, sequences, shardFirst, totalPairs)
                                   : parseInput(pairFileName, sequenceIdxs, sequences);
    if (world == 1) totalPairs = fileInfo.numPairs;
    printf("Num Pairs: %zu\n\n", totalPairs);
    const size_t shardLo = 0, shardHi = fileInfo.numPairs; // indices into this rank's own records
    if (world > 1) printf("Rank %d of %d: pairs [%zu, %zu)\n\n", rank, world, shardFirst, shardFirst + fileInfo.numPairs);

    start_timer();
    uint64_t kernel_time = 0, memalloc_time = 0, backtracking_time = 0, printing_time = 0; // usec, as V19.cu:411-415
    printf("Pair # | Score\n");
    const dpx_params prm{algo, match, mismatch, gapOpen, gapExtend, band};
    static_assert(sizeof(seqPair) == sizeof(dpx_seq_pair), "seqPair must stay layout-compatible with the C ABI");

    std::thread printer;
    BatchOut outs[2]; // batch k is printed from one while batch k+1 is assembled in the other; buffers are reused
    size_t batchNo = 0;
    size_t shardCells = 0;
    for (size_t i = shardLo; i < shardHi; i++) shardCells += (size_t)sequenceIdxs[i].referenceSize * (size_t)sequenceIdxs[i].querySize;
    for (size_t first = shardLo; first < shardHi; first += batchSize) {
        const size_t count = std::min(batchSize, shardHi - first);
        uint64_t t0 = get_time();
        dpx_batch *b = nullptr;
        rc = dpx_batch_create(&prm, sequences, fileInfo.numBytes, reinterpret_cast<const dpx_seq_pair *>(sequenceIdxs), first, count,
                              DPX_KEEP_MATRICES, &b);
        if (rc != DPX_OK) die("FAILED TO CREATE DEVICE BATCH", rc);
        memalloc_time += get_time() - t0;

        double usec = 0;
        if ((rc = dpx_batch_fill_timed(b, 1, &usec)) != DPX_OK) die("KERNEL LAUNCH FAILED", rc);
        kernel_time += (uint64_t)usec;

        t0 = get_time();
        BatchOut *out = &outs[batchNo++ & 1];
        out->first = shardFirst + first; // global pair number of the batch's first pair
        out->score.resize(count);
        if ((rc = dpx_batch_results(b, out->score.data(), nullptr, nullptr)) != DPX_OK) die("FAILED TO COPY SCORES", rc);
        if (print) {
            out->len.resize(count);
            out->off.resize(count);
            size_t total = 0; // worst case: an alignment is at most m + n columns long
            for (size_t k = 0; k < count; k++)
                total += 3 * ((size_t)sequenceIdxs[first + k].referenceSize + (size_t)sequenceIdxs[first + k].querySize + 2);
            if (out->textCap < total) { out->text.reset(new char[total]); out->textCap = total; }
            size_t at = 0;
            for (size_t k = 0; k < count; k++) {
                const size_t cap = (size_t)sequenceIdxs[first + k].referenceSize + (size_t)sequenceIdxs[first + k].querySize + 2;
                char *t = out->text.get() + at; // the three lines land back to back once the length is known
                int32_t len = 0;
                if ((rc = dpx_batch_traceback(b, k, t, t + cap, t + 2 * cap, &len)) != DPX_OK) die("TRACEBACK FAILED", rc);
                const size_t stride = (size_t)len + 1;
                if (stride != cap) { memmove(t + stride, t + cap, stride); memmove(t + 2 * stride, t + 2 * cap, stride); }
                out->off[k] = at;
                out->len[k] = len;
                at += 3 * stride;
            }
        }
        backtracking_time += get_time() - t0;
        t0 = get_time();
        dpx_batch_destroy(b);
        memalloc_time += get_time() - t0;

        // hand the finished batch to the printer; it prints while the next batch is created and filled
        if (printer.joinable()) { t0 = get_time(); printer.join(); printing_time += get_time() - t0; }
        if (print) printer = std::thread(print_batch, out, local);
    }
    if (printer.joinable()) { uint64_t t0 = get_time(); printer.join(); printing_time += get_time() - t0; }

    const uint64_t elapsed_time = get_elapsed_time();
    printf("Elapsed time (usec): %llu\n", (unsigned long long)elapsed_time);
    printf("Num Pairs: %zu\n", totalPairs);
    printf("Num Cells: %zu\n", fileInfo.numCells);
    printf("Reference length min/avg/max: %zu / %.1f / %zu\n", fileInfo.minReferenceLength, fileInfo.avgReferenceLength, fileInfo.maxReferenceLength);
    printf("Query length min/avg/max: %zu / %.1f / %zu\n", fileInfo.minQueryLength, fileInfo.avgQueryLength, fileInfo.maxQueryLength);
    printf("Kernel time (usec): %llu\n", (unsigned long long)kernel_time);
    printf("Memory management time (usec): %llu\n", (unsigned long long)memalloc_time);
    printf("Backtracking time (usec): %llu\n", (unsigned long long)backtracking_time);
    printf("Printing wait time (usec): %llu\n", (unsigned long long)printing_time);
    // GCUPS exactly as the reference computes it (V12.cu:487-491): numCells / kernel seconds / 1e9
    printf("GCUPS: %f\n", kernel_time ? (double)shardCells / ((double)kernel_time * 1e-6) / 1e9 : 0.0);

    printf("Cleaning up\n");
    cleanupParsedFile(sequenceIdxs, sequences);
    return 0;
}
