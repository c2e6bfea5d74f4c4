#include "printLock.h"

#include <cstdio>

namespace {
pthread_mutex_t g_stdoutMutex = PTHREAD_MUTEX_INITIALIZER;
}

void printLock() { pthread_mutex_lock(&g_stdoutMutex); }
void printUnlock() { pthread_mutex_unlock(&g_stdoutMutex); }

PrintGuard::~PrintGuard() { printUnlock(); }
