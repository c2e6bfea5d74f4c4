// dpx_main.cpp -- batched GPU driver of the MI355X engine, shaped like the reference's CUDA mains
// (cuda/LNW/LinearNeedlemanWunschV19.cu:357-680, cuda/LinearSmithWaterman.cu:172-435): parse the pairs file,
// fill batch after batch, trace back on the device, and print "<pair> | <score>" + three lines per pair in input order.
// Two batches are in flight on two streams (the copy/compute overlap of cuda/LNW/LinearNeedlemanWunschV13.cu:414-495):
// while the device fills batch k+1, batch k's traceback, packing and D2H run on the other stream and batch k-1 is
// being written to stdout by the printer thread (the software pipeline of V19.cu:546-579).  The device formats every
// pair's block itself (packed variable-length result strings, V15.cu:168-172,372-425), so printing a batch is one fwrite.
// stdout keeps the reference's lines so logs stay diff-able.
//
//   dpx_main -pairs <file> [-match 3] [-mismatch -1] [-open -2 | -gap -2] [-extend -1]
//            [-algo LSW|LNW|ANW|BSW] [-band 128] [-batch N | -pool-gb 4] [-inflight K] [-tune 0|1] [-device 0] [-noprint] [-pack2] [-producer P] [-rank r -world w]
//
// Batch size: by default from a matrix-pool BUDGET (-pool-gb, 4 GiB): as many pairs as fit the budget, at most 20000 (the
// reference sizes its buffers once for BATCH_SIZE = 10000 reads of 150 bases, cuda/LNW/LinearNeedlemanWunschV9.cu:26-46,
// V14.cu:144-213 -- 10000 pairs of 1024 x 1024 would be a 22 GB pool whose allocation costs 50 times the fill).  -inflight such
// pools (default 3, at most 8) are built on a helper thread while the file is being parsed and then recycled by every batch: that many
// batches are on the device at a time.  Round 4 (profiles/r04/e2e_*.txt): a batch of 1700 pairs of 1024 x 1024 is a chain fill -> traceback
// -> text -> copy of ~1.5 ms; with two in flight the device idles while this thread creates the next batch (7.1 - 7.9 ms for 10 000 pairs),
// with three the next fill is already queued (6.6 - 7.4 ms); deeper pipelines only add fills that share the same HBM (4 ... 8: slower again).
//
// Multi-GPU: pairs are independent, so every GPU gets one process with its own contiguous shard of the file:
// `-rank r -world w` aligns only pairs [ceil(N/w)*r, ceil(N/w)*(r+1)) (the same split as shard.py / bench.py) and
// prints them with their global pair numbers; tools/run_multi_gpu.sh starts one rank per device and concatenates
// the outputs in rank order, which is input order.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dpx_align.h"
#include "parseInput.h"
#include "timing.h"

namespace {

[[noreturn]] void die(const char *what, int rc) {
    printf("%s\nDPX ERROR: %s (%s)\n", what, dpx_strerror(rc), dpx_last_error());
    exit(1);
}

struct InFlight { // one batch between dpx_batch_create and dpx_batch_destroy
    dpx_batch *b = nullptr;
    size_t first = 0, count = 0;
};

} // namespace

int main(int argc, char *argv[]) {
    if (argc < 3) {
        fprintf(stderr, "usage: dpx_main -pairs <InSeqFile> -match <matchWeight> -mismatch <mismatchWeight> -open <gapWeight> "
                        "[-extend <gapExtend>] [-algo LSW|LNW|ANW|BSW] [-band <B>] [-batch <N>] [-device <D>] [-noprint]\n");
        exit(EXIT_FAILURE);
    }
    const char *pairFileName = nullptr;
    int match = 3, mismatch = -1, gapOpen = -2, gapExtend = -1, band = 128, device = 0, rank = 0, world = 1;
    size_t batchSize = 0;     // 0: from the pool budget (the reference's BATCH_SIZE, V19.cu:9, assumes short reads)
    double poolGb = 4.0;
    bool print = true, pack2 = false;
    int producerFlag = -1; // producer threads; -1: by batch size (2 for batches of many short pairs, none for few long ones)
    int inflight = 3;      // batches on the device at a time (= matrix pools reserved)
    int tuneFlag = -1;     // -1: by the length of the job
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
        else if (!strcmp(argv[i], "-pool-gb")) poolGb = atof(next("-pool-gb"));
        else if (!strcmp(argv[i], "-device")) device = atoi(next("-device"));
        else if (!strcmp(argv[i], "-noprint")) print = false;
        else if (!strcmp(argv[i], "-pack2")) pack2 = true;
        else if (!strcmp(argv[i], "-producer")) producerFlag = atoi(next("-producer"));
        else if (!strcmp(argv[i], "-inflight")) inflight = atoi(next("-inflight"));
        else if (!strcmp(argv[i], "-tune")) tuneFlag = atoi(next("-tune"));
        else if (!strcmp(argv[i], "-rank")) rank = atoi(next("-rank"));
        else if (!strcmp(argv[i], "-world")) world = atoi(next("-world"));
        else { fprintf(stderr, "unknown argument: %s\n", argv[i]); exit(EXIT_FAILURE); }
    }
    if (!pairFileName) { fprintf(stderr, "need -pairs <file>\n"); exit(EXIT_FAILURE); }
    if (poolGb < 0.0625 || poolGb > 200) { fprintf(stderr, "bad -pool-gb\n"); exit(EXIT_FAILURE); }
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad -rank/-world\n"); exit(EXIT_FAILURE); }
    if (inflight < 1 || inflight > 8) { fprintf(stderr, "bad -inflight (1..8)\n"); exit(EXIT_FAILURE); }
    const int algo = algoName == "LNW" ? DPX_ALGO_LNW : algoName == "LSW" ? DPX_ALGO_LSW : algoName == "ANW" ? DPX_ALGO_ANW
                     : algoName == "BSW" ? DPX_ALGO_BSW : -1;
    if (algo < 0) { fprintf(stderr, "unknown -algo %s\n", algoName.c_str()); exit(EXIT_FAILURE); }

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

    // the two matrix pools of the pipeline are built while the file is parsed (nothing else needs the device yet)
    // (the budget is per score plane: the three planes of the affine algorithm get three times the bytes, so that a batch holds
    // as many pairs -- and fills the chip as well -- as a linear-gap batch of the same shapes)
    const size_t poolBudget = (size_t)(poolGb * (double)(1ull << 30)) * (algo == DPX_ALGO_ANW ? 3 : 1);
    std::thread reserve;
    // (and pinned text buffers: one being printed, one per batch in flight, two spare)
    const bool budgetedBatches = batchSize == 0;
    if (budgetedBatches) reserve = std::thread([poolBudget, print, inflight]() { (void)dpx_pool_reserve(poolBudget, inflight); if (print) (void)dpx_text_reserve((size_t)16 << 20, inflight + 3); });

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

    if (batchSize == 0) { // pairs per batch from the pool budget: 2 bytes per cell and plane, rows / columns padded as the layouts pad them
        const double cols = (algo == DPX_ALGO_BSW && 2.0 * band < (double)fileInfo.maxReferenceLength) ? 2.0 * band + 8 : (double)fileInfo.maxReferenceLength + 128;
        const double perPair = 2.0 * (algo == DPX_ALGO_ANW ? 3 : 1) * ((double)fileInfo.maxQueryLength + 64) * cols;
        const double fit = (double)poolBudget / (perPair > 0 ? perPair : 1);
        batchSize = (size_t)std::min(20000.0, std::max(64.0, fit));
        batchSize &= ~(size_t)1; // even: the packed kernels fill COUPLES of equal-shaped pairs, an odd batch leaves one pair to a second kernel
                                 // of one wave (0.4 ms of latency on 1024 x 1024, and on a shared hardware queue the couples' kernel waits for it)
    }
    // -pack2: the parse step emits four bases per byte (alphabets of up to four symbols); the batches then move a quarter of the
    // sequence bytes to the device, which expands them (dpx_batch_create_packed2).  Like parsing, outside the timer.
    std::vector<uint8_t> packed;
    uint8_t alphabet[4] = {0, 0, 0, 0};
    if (pack2) {
        packed.resize((fileInfo.numBytes + 3) / 4 + 1);
        rc = dpx_pack2(sequences, fileInfo.numBytes, reinterpret_cast<const dpx_seq_pair *>(sequenceIdxs), fileInfo.numPairs, alphabet, packed.data());
        if (rc == DPX_ERR_UNSUPPORTED) { fprintf(stderr, "-pack2: more than four distinct symbols, keeping bytes\n"); pack2 = false; }
        else if (rc != DPX_OK) die("FAILED TO PACK THE SEQUENCES", rc);
    }
    if (reserve.joinable()) reserve.join();
    // Producer threads (below) keep one batch more alive than -inflight says; without a pool of its own that batch would build one inside
    // the timed region (a 4-GiB pool: 100 - 450 ms, seen with -producer 2 on long pairs; a short-read pool: a few ms of the job's six).
    const int producers = producerFlag >= 0 ? std::min(producerFlag, 8) : (batchSize >= 8192 ? 2 : 0);
    if (budgetedBatches && producers > 0 && producers + 1 > inflight) {
        (void)dpx_pool_reserve(poolBudget, std::min(8, producers + 1));
        if (print) (void)dpx_text_reserve((size_t)16 << 20, std::min(9, producers + 1 + 3));
    }
    start_timer();
    uint64_t kernel_time = 0, memalloc_time = 0, backtracking_time = 0, printing_time = 0; // usec, as V19.cu:411-415
    uint64_t traceback_kernel_time = 0; // device time of the traceback + text kernels (backtracking_time is the host's wait for them)
    printf("Pair # | Score\n");
    const dpx_params prm{algo, match, mismatch, gapOpen, gapExtend, band};
    static_assert(sizeof(seqPair) == sizeof(dpx_seq_pair), "seqPair must stay layout-compatible with the C ABI");

    // pipeline of three host threads: the producer issues [create + fill + output_begin] of batch k+1 while this thread waits
    // for batch k and takes its text; the printer thread writes batch k-1 meanwhile from the text buffer it took over, so batch
    // k-1 itself is destroyed (its matrix pool parked for batch k+1) as soon as its text is on the host.
    std::thread printer;
    char *printingText = nullptr;
    InFlight filling; // issued to the device, not yet waited for
    auto retire_printed = [&]() {
        if (printer.joinable()) { const uint64_t t0 = get_time(); printer.join(); printing_time += get_time() - t0; }
        if (printingText) { dpx_text_free(printingText); printingText = nullptr; }
    };
    auto finish = [&](InFlight &f) { // wait for batch f, account its kernel time, hand its text to the printer
        uint64_t t0 = get_time();
        char *text = nullptr;
        size_t bytes = 0;
        if (print) {
            if ((rc = dpx_batch_output_take(f.b, &text, &bytes)) != DPX_OK) die("TRACEBACK FAILED", rc);
        } else if ((rc = dpx_batch_sync(f.b)) != DPX_OK) die("KERNEL FAILED", rc);
        double usec = 0;
        if ((rc = dpx_batch_last_fill_usec(f.b, &usec)) != DPX_OK) die("KERNEL TIMING FAILED", rc);
        kernel_time += (uint64_t)usec;
        if (print && dpx_batch_last_output_usec(f.b, &usec) == DPX_OK) traceback_kernel_time += (uint64_t)usec;
        backtracking_time += get_time() - t0;
        t0 = get_time();
        dpx_batch_destroy(f.b);
        f = InFlight{};
        memalloc_time += get_time() - t0;
        retire_printed();
        printingText = text;
        if (print) printer = std::thread([text, bytes]() { fwrite(text, 1, bytes, stdout); });
    };
    size_t shardCells = 0;
    for (size_t i = shardLo; i < shardHi; i++) shardCells += (size_t)sequenceIdxs[i].referenceSize * (size_t)sequenceIdxs[i].querySize;
    fflush(stdout); // the printer thread writes with fwrite from here on
    // One batch: create, fill, start the output.  Two batches alive need two matrix pools.  Allocating tens of GB costs hundreds of
    // ms (more than the overlap of one batch's traceback with the next batch's fill can ever win back), so batches with pools of
    // 16 GiB or more (an explicit -batch) run one after the other and share ONE parked pool; the printer thread still overlaps.
    size_t maxAlive = (size_t)inflight;
    // Pool placement: the same fill runs up to 27 % apart on two allocations of one pool (ANW 1000 x 1024^2: 0.94 vs 1.20 ms;
    // no address pattern of the fill moves it, profiles/r04/anw_group_sweep_on_fixed_allocations.txt).  A job of 256 batches or more lets the engine shop
    // for its pools with the first batch that uses each of them (DPX_TUNE_PLACEMENT: five candidate allocations, four fills each -- ~35 ms per pool,
    // once); shorter jobs would not earn that back.  -tune 0|1 overrides.
    const size_t jobBatches = (shardHi - shardLo + batchSize - 1) / batchSize;
    const bool tunePools = tuneFlag >= 0 ? tuneFlag != 0 : jobBatches >= 256;
    std::atomic<int> tuned{0}; // batches created with the flag so far (one per pool in flight)
    std::atomic<uint64_t> create_time{0}; // summed over the threads that produce
    auto produce = [&](size_t first) -> InFlight {
        InFlight next;
        next.first = first;
        next.count = std::min(batchSize, shardHi - first);
        const uint64_t t0 = get_time();
        const unsigned flags = DPX_KEEP_MATRICES | DPX_TIME_FILLS | ((tunePools && tuned.fetch_add(1) < inflight) ? DPX_TUNE_PLACEMENT : 0u);
        int prc = pack2 ? dpx_batch_create_packed2(-1, &prm, packed.data(), fileInfo.numBytes, alphabet, reinterpret_cast<const dpx_seq_pair *>(sequenceIdxs),
                                                   first, next.count, flags, &next.b)
                        : dpx_batch_create(&prm, sequences, fileInfo.numBytes, reinterpret_cast<const dpx_seq_pair *>(sequenceIdxs), first, next.count,
                                           flags, &next.b);
        if (prc != DPX_OK) die("FAILED TO CREATE DEVICE BATCH", prc);
        create_time += get_time() - t0;
        if ((prc = dpx_batch_fill(next.b, nullptr)) != DPX_OK) die("KERNEL LAUNCH FAILED", prc);
        // global pair numbers: shardFirst + index inside the shard
        if (print && (prc = dpx_batch_output_begin(next.b, shardFirst + first)) != DPX_OK) die("TRACEBACK LAUNCH FAILED", prc);
        return next;
    };
    auto pool_is_huge = [](const InFlight &f) { uint64_t mb = 0; dpx_batch_info(f.b, nullptr, nullptr, &mb, nullptr); return mb >= (16ull << 30); };
    // Batches of many short pairs are bound by the HOST (dpx_batch_create: 1.1-1.4 ms per 20000 pairs for validation, two counting
    // sorts, the wave packing and the H2D copies, against 0.15 + 0.3 ms of kernels).  Round 3 moved that onto ONE producer thread
    // (100k short reads: 12 -> 7.5 ms, of which 5.9 ms were still that thread creating five batches one after the other); batches are
    // independent, so round 4 creates them on several threads (-producer P, default 2: 100k short reads 6.3 -> 5.9 ms LSW, 9.2 -> 8.1 ANW; three or four threads contend and lose again, profiles/r04/e2e_producers.txt): every thread takes the next batch index, creates
    // and fills the batch and starts its output, this thread finishes the batches in input order.  At most -inflight batches exist
    // between the one being finished and the newest one being created.  Batches of few long pairs are bound by the device, and there one
    // issuing thread is as fast or faster (10000 x 1024^2: 8.1-8.9 vs 8.4-9.3 ms).  -producer 0 turns the threads off.
    if (producers > 0) {
        const size_t numBatches = (shardHi - shardLo + batchSize - 1) / batchSize;
        if ((size_t)producers + 1 > maxAlive) maxAlive = std::min<size_t>(8, (size_t)producers + 1); // (every producer needs a slot of its own)
        std::mutex qm;
        std::condition_variable qcv;
        std::map<size_t, InFlight> ready; // created, filling, not yet finished; by batch index
        std::atomic<size_t> nextBatch{0};
        size_t consumed = 0; // batches finished (and destroyed) so far
        std::vector<std::thread> pool;
        for (int t = 0; t < producers; t++)
            pool.emplace_back([&]() {
                (void)dpx_init(device); // (binds the device for this thread)
                for (;;) {
                    const size_t k = nextBatch.fetch_add(1);
                    if (k >= numBatches) return;
                    {
                        std::unique_lock<std::mutex> lk(qm);
                        qcv.wait(lk, [&]() { return k < consumed + maxAlive; });
                    }
                    const InFlight next = produce(shardLo + k * batchSize);
                    std::lock_guard<std::mutex> lk(qm);
                    if (pool_is_huge(next)) maxAlive = 1;
                    ready[k] = next;
                    qcv.notify_all();
                }
            });
        for (size_t k = 0; k < numBatches; k++) {
            {
                std::unique_lock<std::mutex> lk(qm);
                qcv.wait(lk, [&]() { return ready.count(k) != 0; });
                filling = ready[k];
                ready.erase(k);
            }
            finish(filling); // (destroys the batch: its pool is parked for the batches still to be created)
            std::lock_guard<std::mutex> lk(qm);
            consumed = k + 1;
            qcv.notify_all();
        }
        for (std::thread &t : pool) t.join();
    } else {
        std::deque<InFlight> alive; // issued to the device, oldest first
        for (size_t first = shardLo; first < shardHi; first += batchSize) {
            if (alive.size() >= maxAlive) { finish(alive.front()); alive.pop_front(); } // (its pool is parked for the batch produced next)
            alive.push_back(produce(first));
            if (pool_is_huge(alive.back())) maxAlive = 1;
        }
        while (!alive.empty()) { finish(alive.front()); alive.pop_front(); }
    }
    memalloc_time += create_time.load();
    retire_printed();
    fflush(stdout);

    const uint64_t elapsed_time = get_elapsed_time();
    printf("Elapsed time (usec): %llu\n", (unsigned long long)elapsed_time);
    printf("Num Pairs: %zu\n", totalPairs);
    printf("Num Cells: %zu\n", fileInfo.numCells);
    printf("Reference length min/avg/max: %zu / %.1f / %zu\n", fileInfo.minReferenceLength, fileInfo.avgReferenceLength, fileInfo.maxReferenceLength);
    printf("Query length min/avg/max: %zu / %.1f / %zu\n", fileInfo.minQueryLength, fileInfo.avgQueryLength, fileInfo.maxQueryLength);
    printf("Kernel time (usec): %llu\n", (unsigned long long)kernel_time);
    printf("Memory management time (usec): %llu\n", (unsigned long long)memalloc_time);
    printf("Backtracking time (usec): %llu\n", (unsigned long long)backtracking_time);
    printf("Traceback kernel time (usec): %llu\n", (unsigned long long)traceback_kernel_time);
    printf("Printing wait time (usec): %llu\n", (unsigned long long)printing_time);
    // GCUPS exactly as the reference computes it (V12.cu:487-491): numCells / kernel seconds / 1e9
    printf("GCUPS: %f\n", kernel_time ? (double)shardCells / ((double)kernel_time * 1e-6) / 1e9 : 0.0);

    printf("Cleaning up\n");
    cleanupParsedFile(sequenceIdxs, sequences);
    return 0;
}
