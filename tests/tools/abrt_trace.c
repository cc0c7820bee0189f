/* abrt_trace.c — debugging aid (LD_PRELOAD): a native backtrace of the thread that raises a fatal signal.
 *
 * A process that dies of abort() inside a runtime thread leaves nothing to read when a test runner holds
 * stderr (pytest's fd capture swallows what native code printed right before the abort) and a debugger
 * perturbs the timing.  This shim costs nothing until the signal arrives: its constructor duplicates the
 * ORIGINAL stderr and installs handlers for SIGABRT / SIGSEGV / SIGBUS; the handler writes the faulting
 * thread's frames (module + offset, resolvable with addr2line against the same image) to that descriptor
 * and re-raises with the default action, so the exit status and a core dump are what they would have been.
 * Python's faulthandler, installed later, chains to the previous handler — this one — on the same thread.
 *
 * And because a test runner's fd-level capture swallows what native code printed just before it died (pytest points
 * descriptor 2 at an unlinked temporary file), the handler also copies the TAIL of whatever descriptor 2 is now to the
 * original stderr: the runtime's own last words ("Memory access fault by GPU node ...", a queue error, a glibc heap
 * check) survive the crash.
 *
 *   gcc -O1 -g -fPIC -shared tests/tools/abrt_trace.c -o tests/tools/libabrt_trace.so
 *   loaded by tests/conftest.py (ctypes) before pytest enables faulthandler, or LD_PRELOAD=... for any other process
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int out_fd = 2;      /* the stderr of load time (under a runner that captures early this is ALREADY its capture file) */
static int file_fd = -1;    /* $HODOR_ABORT_TRACE_DIR/abort_trace.<pid>.log: nobody redirects that; opened when the signal arrives */
static char file_path[512];

static void put_n(const char *s, size_t n)
{
    (void)!write(out_fd, s, n);
    if (file_fd >= 0) (void)!write(file_fd, s, n);
}
static void put(const char *s) { put_n(s, strlen(s)); }

static void handler(int sig, siginfo_t *si, void *uc)
{
    (void)uc;
    char line[160];
    void *bt[96];
    if (file_path[0]) file_fd = open(file_path, O_WRONLY | O_CREAT | O_APPEND | O_CLOEXEC, 0644);   /* async-signal-safe */
    int n = backtrace(bt, 96);
    snprintf(line, sizeof line, "\n==== abrt_trace: signal %d (si_code %d, addr %p) in tid %ld of pid %d ====\n", sig,
             si ? si->si_code : 0, si ? si->si_addr : (void *)0, (long)syscall(SYS_gettid), (int)getpid());
    put(line);
    backtrace_symbols_fd(bt, n, out_fd);
    if (file_fd >= 0) backtrace_symbols_fd(bt, n, file_fd);
    /* the thread's name says which runtime it belongs to */
    snprintf(line, sizeof line, "/proc/self/task/%ld/comm", (long)syscall(SYS_gettid));
    int fd = open(line, O_RDONLY);
    if (fd >= 0) {
        char name[64];
        ssize_t k = read(fd, name, sizeof name - 1);
        close(fd);
        if (k > 0) {
            name[k] = 0;
            put("thread name: ");
            put(name);
        }
    }
    /* the last words of this process on its CURRENT stderr, if that is a file we can read back (a capture file) */
    {
        off_t end = lseek(2, 0, SEEK_CUR);
        if (end > 0) {
            static char tail[8192];
            off_t from = end > (off_t)sizeof tail ? end - (off_t)sizeof tail : 0;
            ssize_t k = pread(2, tail, (size_t)(end - from), from);
            if (k > 0) {
                put("---- tail of the captured stderr ----\n");
                put_n(tail, (size_t)k);
                put("\n---- end of captured stderr ----\n");
            }
        }
    }
    put("==== abrt_trace: end ====\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void abrt_trace_init(void)
{
    int d = dup(2);
    if (d >= 0) {
        out_fd = d;
        (void)fcntl(out_fd, F_SETFD, FD_CLOEXEC);
    }
    const char *dir = getenv("HODOR_ABORT_TRACE_DIR");
    if (dir && *dir) snprintf(file_path, sizeof file_path, "%s/abort_trace.%d.log", dir, (int)getpid());
    void *warm[4];
    (void)backtrace(warm, 4); /* loads libgcc's unwinder now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_RESETHAND;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
