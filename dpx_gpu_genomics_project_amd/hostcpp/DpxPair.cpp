#include "DpxPair.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <iomanip>
#include <iostream>

#include "../../include/dpx_align.h"

namespace {
[[noreturn]] void fail(const char *what, int rc) {
    fprintf(stderr, "DPX ENGINE ERROR: %s: %s (%s)\n", what, dpx_strerror(rc), dpx_last_error());
    exit(1);
}
} // namespace

namespace {

// One alignment request of one caller thread.
struct Request {
    int algo, match, mismatch, gapOpen, gapExtend, band;
    const std::string *reference, *query;
    DpxPairResult *out;
    bool done = false;
    bool sameParams(const Request &o) const {
        return algo == o.algo && match == o.match && mismatch == o.mismatch && gapOpen == o.gapOpen && gapExtend == o.gapExtend &&
               band == o.band;
    }
};

// Run a group of requests (same algorithm and weights) as ONE device batch: one create / fill / results / traceback
// round trip instead of one per pair.
void runGroup(const std::vector<Request *> &grp, bool wantMatrices, int device) {
    std::string flat; // parseInput layout: reference '\0' query '\0' per pair
    std::vector<dpx_seq_pair> pairs(grp.size());
    size_t maxCap = 0;
    for (size_t k = 0; k < grp.size(); k++) {
        const std::string &r = *grp[k]->reference, &q = *grp[k]->query;
        pairs[k].referenceIdx = (int)flat.size();
        pairs[k].referenceSize = (int)r.size();
        flat += r;
        flat.push_back('\0');
        pairs[k].queryIdx = (int)flat.size();
        pairs[k].querySize = (int)q.size();
        flat += q;
        flat.push_back('\0');
        maxCap = std::max(maxCap, r.size() + q.size() + 2);
    }
    const Request &first = *grp[0];
    dpx_params prm{first.algo, first.match, first.mismatch, first.gapOpen, first.gapExtend, first.band};
    dpx_batch *b = nullptr;
    int rc = dpx_batch_create_on(device, &prm, flat.data(), flat.size(), pairs.data(), 0, grp.size(), DPX_KEEP_MATRICES, &b);
    if (rc != DPX_OK) fail("dpx_batch_create_on", rc);
    if ((rc = dpx_batch_fill(b, nullptr)) != DPX_OK) fail("dpx_batch_fill", rc);
    // (the traceback and text kernels queue up behind the fill before the host waits for anything: one wait instead of two per round trip)
    if ((rc = dpx_batch_output_begin(b, 0)) != DPX_OK) fail("dpx_batch_output_begin", rc);
    std::vector<int32_t> score(grp.size()), er(grp.size()), ec(grp.size());
    if ((rc = dpx_batch_results(b, score.data(), er.data(), ec.data())) != DPX_OK) fail("dpx_batch_results", rc);
    std::vector<char> l0(maxCap), l1(maxCap), l2(maxCap);
    for (size_t k = 0; k < grp.size(); k++) {
        DpxPairResult &out = *grp[k]->out;
        out.score = score[k];
        out.endRow = er[k];
        out.endCol = ec[k];
        int32_t len = 0;
        if ((rc = dpx_batch_traceback(b, k, l0.data(), l1.data(), l2.data(), &len)) != DPX_OK) fail("dpx_batch_traceback", rc);
        out.refLine.assign(l0.data(), (size_t)len);
        out.relLine.assign(l1.data(), (size_t)len);
        out.qryLine.assign(l2.data(), (size_t)len);
        if (wantMatrices) {
            const size_t cells = (grp[k]->reference->size() + 1) * (grp[k]->query->size() + 1);
            out.H.resize(cells);
            if ((rc = dpx_batch_matrix(b, k, DPX_MAT_H, out.H.data())) != DPX_OK) fail("dpx_batch_matrix(H)", rc);
            if (first.algo == DPX_ALGO_ANW) {
                out.I.resize(cells);
                out.D.resize(cells);
                if ((rc = dpx_batch_matrix(b, k, DPX_MAT_I, out.I.data())) != DPX_OK) fail("dpx_batch_matrix(I)", rc);
                if ((rc = dpx_batch_matrix(b, k, DPX_MAT_D, out.D.data())) != DPX_OK) fail("dpx_batch_matrix(D)", rc);
            }
        }
    }
    dpx_batch_destroy(b);
}

// The reference's driver aligns one pair per call from 20 threads (c++/main.cpp:18-19,203).  A GPU round trip costs
// ~0.2 ms whatever the number of pairs in it, so concurrent callers are combined: a thread that finds a free device
// becomes a leader, lets the others queue up for a moment, runs everybody's pairs as one device batch and hands the
// results back.  Same results, same stdout; the per-pair cost drops by about the number of threads.
// Several devices: up to one leader per visible GPU at a time, devices dealt round-robin (DPX_DEVICES=n limits the devices
// used; DPX_CLASS_LEADERS=n sets the number of concurrent leaders independently, e.g. to rehearse the multi-device hand-over
// on a one-GPU box).  While a leader is gathering nobody else is elected (the arrivals are what it is waiting for); when its
// window closes it takes only ITS SHARE of the queue -- queue / (free leader slots + 1) requests -- and wakes the others, who
// lead the rest at once, without a window of their own: 20 callers on 8 devices become groups of 3,3,3,3,2,2,2,2, on one
// device one group of 20.  (Rehearsed with several leaders on ONE GPU only -- no multi-GPU box was available to the builder.)
std::mutex g_mu;
std::condition_variable g_cv;
std::vector<Request *> g_queue;
int g_activeLeaders = 0, g_maxLeaders = 0, g_numDevices = 0;
bool g_gathering = false; // a leader is inside its gather window
bool g_splitRest = false; // the queue holds what a leader left to the other devices when its window closed
unsigned g_nextSlot = 0;

void initDevices() { // under g_mu
    if (g_maxLeaders) return;
    int n = 0;
    if (dpx_device_count(&n) != DPX_OK || n <= 0) fail("dpx_device_count", DPX_ERR_NO_DEVICE);
    if (const char *env = getenv("DPX_DEVICES")) { const int v = atoi(env); if (v >= 1 && v < n) n = v; }
    g_numDevices = n;
    g_maxLeaders = n;
    if (const char *env = getenv("DPX_CLASS_LEADERS")) { const int v = atoi(env); if (v >= 1 && v <= 64) g_maxLeaders = v; }
}

size_t g_crowd = 0; // callers the last gather window ended with

void gatherWindow(std::unique_lock<std::mutex> &lk) {
    // Wait while callers keep arriving: stop after 40 us without a new request, 400 us in total at most -- or after 8 us without one when
    // as many have arrived as the last window ended with (the reference's driver runs a fixed number of threads in lockstep: the crowd
    // is complete, but a late-comer inside those 8 us still grows it; when threads retire the 40-us rule shrinks it again).  Round 4:
    // without the 8 us of grace the crowd only ever shrank (417 device round trips instead of 241 for 4000 pairs).
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    auto lastArrival = t0;
    size_t seen = g_queue.size();
    for (;;) {
        lk.unlock();
        std::this_thread::yield();
        lk.lock();
        const auto now = clock::now();
        if (g_queue.size() != seen) { seen = g_queue.size(); lastArrival = now; }
        const auto quiet = now - lastArrival;
        if (quiet > std::chrono::microseconds(40) || now - t0 > std::chrono::microseconds(400)) break;
        if (g_crowd > 1 && seen >= g_crowd && quiet > std::chrono::microseconds(8)) break;
    }
    g_crowd = g_queue.size();
}

} // namespace

void dpxAlignPair(int algo, const std::string &reference, const std::string &query, int match, int mismatch, int gapOpen,
                  int gapExtend, int band, bool wantMatrices, DpxPairResult &out) {
    Request rq{algo, match, mismatch, gapOpen, gapExtend, band, &reference, &query, &out};
    if (wantMatrices) { // matrix dumps (PRINT_MATRIX builds, print_matrix()): one pair, its own batch, default device
        runGroup({&rq}, true, -1);
        return;
    }
    std::unique_lock<std::mutex> lk(g_mu);
    initDevices();
    g_queue.push_back(&rq);
    for (;;) {
        if (rq.done) return;                                          // a leader has served this request
        if (g_activeLeaders >= g_maxLeaders || g_queue.empty() || g_gathering) { // no free device, nothing queued (this request is in
            g_cv.wait(lk);                                                         // a leader's hands), or a leader is still gathering
            continue;
        }
        g_activeLeaders++; // become a leader on the next device
        const int device = (int)(g_nextSlot++ % (unsigned)g_numDevices);
        // Give the other callers a moment to arrive -- also when some are queued already: they came while the device was busy with
        // another group, and served at once the two groups stay out of phase for good (round 4: the reference driver's 20 threads then
        // alternate as 10 + 10: 280 - 400 device round trips instead of 205 - 220 for 4000 short pairs, 93 - 120 ms instead of 86 - 106 on
        // the same box).  Only the rest that a leader left to the other devices after ITS window is served without one.
        if (!g_splitRest) {
            g_gathering = true;
            gatherWindow(lk);
            g_gathering = false;
        }
        // this leader's share of the requests that carry the parameters of the oldest one; the rest stays queued for the next leader
        std::vector<Request *> grp, rest;
        if (!g_queue.empty()) {
            const Request &head = *g_queue.front();
            size_t same = 0;
            for (Request *r : g_queue) same += r->sameParams(head) ? 1 : 0;
            const size_t couldRun = (size_t)std::max(1, g_maxLeaders - g_activeLeaders + 1); // this leader + the free slots
            const size_t share = (same + couldRun - 1) / couldRun;
            for (Request *r : g_queue) ((r->sameParams(head) && grp.size() < share) ? grp : rest).push_back(r);
            g_queue.swap(rest);
            g_splitRest = !g_queue.empty() && couldRun > 1;
        }
        if (!g_queue.empty()) g_cv.notify_all(); // somebody else leads the rest, now
        lk.unlock();
        if (!grp.empty()) runGroup(grp, false, device);
        lk.lock();
        for (Request *r : grp) r->done = true;
        g_activeLeaders--;
        g_cv.notify_all(); // served followers return; one of the others (if any) becomes the next leader
    }
}

void dpxPrintScoreMatrix(const std::string &reference, const std::string &query, const std::vector<short> &M) {
    using std::cout;
    const size_t rows = query.size() + 1, cols = reference.size() + 1;
    const int w = 2;
    cout << "Reference: " << reference << " Size: " << reference.size() << "\n";
    cout << "Query: " << query << " Size: " << query.size() << "\n";
    cout << "Matrix Dim: [ " << rows << " x " << cols << " ]\n";
    cout << "    " << std::setw(w) << " " << " ";
    for (char c : reference) cout << "  " << std::setw(w) << c << " ";
    cout << "\n";
    for (size_t i = 0; i < rows; i++) {
        if (i == 0) cout << "  ";
        else cout << query[i - 1] << " ";
        cout << "[";
        for (size_t j = 0; j < cols; j++) {
            cout << " " << std::setw(w) << M[i * cols + j];
            if (j + 1 != cols) cout << ", ";
        }
        cout << "],\n";
    }
    cout << std::endl;
}
