// printLock.h -- one process-wide stdout mutex, the interface of the reference's c++/printLock.h:1-8 (bodies:
// printLock.cpp).  The reference's main.cpp keys its pthread driver on USE_THREADS being defined by this header.
#ifndef DPX_HOSTCPP_PRINTLOCK_H
#define DPX_HOSTCPP_PRINTLOCK_H
#include <pthread.h>

#ifndef USE_THREADS
#define USE_THREADS 1
#endif

void printLock();   // blocks until this thread owns stdout
void printUnlock(); // releases it

// scope guard for new code: { PrintGuard g; printf(...); } -- flushes before it lets go
struct PrintGuard {
    PrintGuard() { printLock(); }
    ~PrintGuard();
    PrintGuard(const PrintGuard &) = delete;
    PrintGuard &operator=(const PrintGuard &) = delete;
};
#endif
