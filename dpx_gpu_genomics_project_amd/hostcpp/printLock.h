// printLock.h -- one process-wide stdout mutex (reference: c++/printLock.h:1-8, printLock.cpp:3-10).
// USE_THREADS is defined here exactly as the reference does, because its main.cpp keys its pthread driver on it.
#pragma once
#include "pthread.h"

#define USE_THREADS

void printLock();
void printUnlock();
