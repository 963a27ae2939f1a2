/* LD_PRELOAD shim for the hunt of the test process's intermittent abort (profiles/r05/README.md): a SIGABRT handler that is
 * installed before the interpreter starts (faulthandler chains to it) and leaves, in the file $SRLA_ABORT_LOG, the native call
 * stack of the thread that aborted and the tail of whatever file descriptor 2 points to at that moment -- pytest's capture
 * replaces fd 2 by a temporary file, which is where the runtime's "Memory access fault by GPU ..." or glibc's "free(): invalid
 * pointer" went and got lost with the process.
 *     gcc -O1 -g -fPIC -shared -o tools/r06/libabort_shim.so tools/r06/abort_shim.c -ldl
 *     SRLA_ABORT_LOG=gpurun_out/abort.log LD_PRELOAD=$PWD/tools/r06/libabort_shim.so python -m pytest ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static char log_path[512];

static void put(int fd, const char *s) { (void)!write(fd, s, strlen(s)); }

static void on_abort(int sig)
{
    static volatile int once;
    if (!once) {
        once = 1;
        int fd = open(log_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) {
            void *frames[96];
            put(fd, "==== SIGABRT: native call stack of the aborting thread\n");
            int n = backtrace(frames, 96);
            backtrace_symbols_fd(frames, n, fd);
            put(fd, "==== tail of what file descriptor 2 points to\n");
            int r = open("/proc/self/fd/2", O_RDONLY);
            if (r >= 0) {
                static char buf[16384];
                off_t sz = lseek(r, 0, SEEK_END);
                if (sz > 0) {
                    lseek(r, sz > (off_t)sizeof(buf) ? sz - (off_t)sizeof(buf) : 0, SEEK_SET);
                    ssize_t got = read(r, buf, sizeof(buf));
                    if (got > 0) (void)!write(fd, buf, (size_t)got);
                } else put(fd, "(not a seekable file)\n");
                close(r);
            }
            put(fd, "\n==== /proc/self/maps entries that contain 'heap' or 'kfd' or 'dri'\n");
            FILE *m = fopen("/proc/self/maps", "r");
            if (m) {
                char line[512];
                while (fgets(line, sizeof(line), m))
                    if (strstr(line, "heap") || strstr(line, "kfd") || strstr(line, "dri")) put(fd, line);
                fclose(m);
            }
            close(fd);
        }
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    const char *p = getenv("SRLA_ABORT_LOG");
    snprintf(log_path, sizeof(log_path), "%s", p ? p : "/tmp/srla_abort.log");
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abort;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, NULL);
}
