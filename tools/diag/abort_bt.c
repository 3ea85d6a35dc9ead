// LD_PRELOAD shim for diagnosing an abort() inside a test process (round-5 investigation of the round-4 SIGABRT in the
// GPU suite): on SIGABRT -- and on a direct call of abort() through the PLT -- the ABORTING thread writes its native
// backtrace, its name, and whatever the process wrote to a captured stderr (pytest's fd-capture points fd 2 at an unlinked
// temporary file: messages of the GPU runtime written just before the abort are otherwise lost) to $SIGMA_ABORT_BT.
//   gcc -shared -fPIC -O1 -o tools/diag/libabort_bt.so tools/diag/abort_bt.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

static volatile int g_done;

static void dump(const char* why) {
    if (__sync_lock_test_and_set(&g_done, 1)) return;
    const char* path = getenv("SIGMA_ABORT_BT");
    int fd = open(path ? path : "/tmp/sigma_abort_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd < 0) return;
    char name[32] = {0};
    prctl(PR_GET_NAME, name, 0, 0, 0);
    dprintf(fd, "==== %s: pid %d tid %ld thread name '%s'\n", why, getpid(), (long)syscall(SYS_gettid), name);
    void* bt[128];
    int n = backtrace(bt, 128);
    backtrace_symbols_fd(bt, n, fd);
    for (int cap = 1; cap <= 2; ++cap) {                       // a captured stdout / stderr: regular file -> copy its tail
        struct stat st;
        if (fstat(cap, &st) == 0 && S_ISREG(st.st_mode)) {
            char link[64], buf[4096];
            snprintf(link, sizeof link, "/proc/self/fd/%d", cap);
            int r = open(link, O_RDONLY);
            if (r >= 0) {
                off_t sz = lseek(r, 0, SEEK_END);
                lseek(r, sz > 16384 ? sz - 16384 : 0, SEEK_SET);
                dprintf(fd, "---- tail of captured fd %d (%ld bytes)\n", cap, (long)sz);
                ssize_t k;
                while ((k = read(r, buf, sizeof buf)) > 0) (void)!write(fd, buf, (size_t)k);
                close(r);
                dprintf(fd, "\n---- end of fd %d\n", cap);
            }
        }
    }
    int m = open("/proc/self/maps", O_RDONLY);                  // load addresses, to resolve lib+offset frames afterwards
    if (m >= 0) {
        char buf[4096]; ssize_t k;
        dprintf(fd, "---- maps (executable segments)\n");
        FILE* f = fdopen(m, "r");
        while (f && fgets(buf, sizeof buf, f)) if (strstr(buf, " r-xp ")) (void)!write(fd, buf, strlen(buf));
        if (f) fclose(f);
    }
    close(fd);
}

static void on_abort(int sig) {
    dump("SIGABRT handler");
    signal(sig, SIG_DFL);
    raise(sig);
}

void abort(void) {
    dump("abort() called");
    void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
    signal(SIGABRT, SIG_DFL);
    if (real) real();
    _exit(134);
}

__attribute__((constructor)) static void init(void) {
    void* warm[4];
    backtrace(warm, 4);                                         // loads libgcc now, not inside the handler
    signal(SIGABRT, on_abort);
}
