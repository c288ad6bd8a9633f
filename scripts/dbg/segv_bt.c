/* Developer aid: LD_PRELOAD=scripts/dbg/segv_bt.so prints a backtrace on SIGSEGV / SIGABRT (also during process exit, where Python's faulthandler is gone). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void handler(int sig) {
    void *frames[64];
    int n = backtrace(frames, 64);
    dprintf(2, "[segv_bt] signal %d, %d frames\n", sig, n);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}
__attribute__((constructor)) static void install(void) {
    signal(SIGSEGV, handler);
    signal(SIGABRT, handler);
    signal(SIGBUS, handler);
}
