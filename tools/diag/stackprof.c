/* stackprof.c - two small diagnostics for a process that drives the library (no perf / strace in the image):
 *   stackprof_start(hz) / stackprof_stop() / stackprof_dump(path): SIGPROF sampling of the PROCESS's CPU time (ITIMER_PROF: the
 *     signal lands on a thread that is burning CPU), native stacks via backtrace(); the dump is one line per sample,
 *     "tid frame0;frame1;..." leaf first, frames as module!symbol+off or module+0xoff (tools/diag/r06_worker_cpu.py folds them).
 *   stackprof_install_crash_handler(): SIGSEGV / SIGBUS / SIGABRT print the faulting thread's native stack, the fault address and
 *     the mappings it and the top frames lie in to stderr, then hand over to the previous handler (or the default action).
 * Built on the box by the scripts that use it:  gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
 * Test infrastructure: never linked into the library. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <unistd.h>

#define DEPTH 28
#define MAX_SAMPLES 400000
struct sample { int tid, n; void *pc[DEPTH]; };
static struct sample *g_samples;
static volatile int g_count;
static volatile int g_running;

static void on_prof(int sig, siginfo_t *si, void *uc) {
    (void)sig; (void)si; (void)uc;
    if (!g_running) return;
    const int i = __sync_fetch_and_add(&g_count, 1);
    if (i >= MAX_SAMPLES) return;
    g_samples[i].tid = (int)syscall(SYS_gettid);
    g_samples[i].n = backtrace(g_samples[i].pc, DEPTH);
}

int stackprof_start(int hz) {
    if (!g_samples) {
        g_samples = mmap(NULL, sizeof(struct sample) * MAX_SAMPLES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (g_samples == MAP_FAILED) { g_samples = NULL; return -1; }
    }
    void *warm[4];
    (void)backtrace(warm, 4);                       /* loads libgcc's unwinder outside the handler */
    g_count = 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGPROF, &sa, NULL) != 0) return -2;
    g_running = 1;
    struct itimerval it;
    it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000000 / (hz > 0 ? hz : 1000);
    it.it_value = it.it_interval;
    return setitimer(ITIMER_PROF, &it, NULL);
}

int stackprof_stop(void) {
    struct itimerval it;
    memset(&it, 0, sizeof it);
    g_running = 0;
    setitimer(ITIMER_PROF, &it, NULL);
    return g_count < MAX_SAMPLES ? g_count : MAX_SAMPLES;
}

static void frame_name(void *pc, char *out, size_t cap) {
    Dl_info di;
    if (dladdr(pc, &di) && di.dli_fname) {
        const char *base = strrchr(di.dli_fname, '/');
        base = base ? base + 1 : di.dli_fname;
        if (di.dli_sname) snprintf(out, cap, "%s!%s+%#lx", base, di.dli_sname, (unsigned long)((uintptr_t)pc - (uintptr_t)di.dli_saddr));
        else snprintf(out, cap, "%s+%#lx", base, (unsigned long)((uintptr_t)pc - (uintptr_t)di.dli_fbase));
    } else {
        snprintf(out, cap, "?+%p", pc);
    }
}

int stackprof_dump(const char *path) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    const int n = g_count < MAX_SAMPLES ? g_count : MAX_SAMPLES;
    char name[768];
    for (int i = 0; i < n; i++) {
        fprintf(f, "%d ", g_samples[i].tid);
        /* frames 0 and 1 are the handler and the signal trampoline */
        for (int k = 2; k < g_samples[i].n; k++) {
            frame_name(g_samples[i].pc[k], name, sizeof name);
            fprintf(f, "%s%s", k > 2 ? ";" : "", name);
        }
        fputc('\n', f);
    }
    fclose(f);
    return n;
}

/* ---- crash handler ------------------------------------------------------------------------------------------------------------ */
static struct sigaction g_prev[65];

static void print_mapping_of(const void *addr, const char *what) {
    FILE *m = fopen("/proc/self/maps", "r");
    if (!m) return;
    char line[1024];
    while (fgets(line, sizeof line, m)) {
        unsigned long a = 0, b = 0;
        if (sscanf(line, "%lx-%lx", &a, &b) == 2 && (unsigned long)(uintptr_t)addr >= a && (unsigned long)(uintptr_t)addr < b) {
            fprintf(stderr, "[stackprof]   %s %p lies in: %s", what, addr, line);
            fclose(m);
            return;
        }
    }
    fprintf(stderr, "[stackprof]   %s %p lies in NO mapping\n", what, addr);
    fclose(m);
}

static void on_crash(int sig, siginfo_t *si, void *uc) {
    (void)uc;
    void *pcs[64];
    const int n = backtrace(pcs, 64);
    fprintf(stderr, "\n[stackprof] signal %d (%s) in thread %d, fault address %p, si_code %d; native stack of the faulting thread:\n", sig,
            sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGABRT ? "SIGABRT" : "?", (int)syscall(SYS_gettid), si ? si->si_addr : NULL, si ? si->si_code : 0);
    char name[768];
    for (int k = 0; k < n; k++) {
        frame_name(pcs[k], name, sizeof name);
        fprintf(stderr, "[stackprof]   #%02d %p %s\n", k, pcs[k], name);
    }
    if (si && (sig == SIGSEGV || sig == SIGBUS)) print_mapping_of(si->si_addr, "fault address");
    fflush(stderr);
    /* hand over: the previous handler if there was one, else the default action (core / exit code 139) */
    struct sigaction *p = &g_prev[sig];
    if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) { p->sa_sigaction(sig, si, uc); return; }
    if (!(p->sa_flags & SA_SIGINFO) && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN && p->sa_handler) { p->sa_handler(sig); return; }
    signal(sig, SIG_DFL);
    raise(sig);
}

int stackprof_install_crash_handler(void) {
    void *warm[4];
    (void)backtrace(warm, 4);
    const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT};
    for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; i++) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_crash;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) != 0) return -1;
    }
    return 0;
}
