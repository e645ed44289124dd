/* Debugging aid: a SIGSEGV / SIGBUS / SIGABRT handler that prints the native call stack of the faulting thread
 * (module + offset per frame, so that frames inside stripped libraries can be located) and then hands over to the
 * handler that was installed before it (Python's faulthandler, which prints the Python frames of every thread).
 *
 *   gcc -O1 -g -fPIC -shared -o tools/libsegvtrace.so tools/segv_trace.c
 *   ctypes.CDLL("tools/libsegvtrace.so").segv_trace_install()      (after faulthandler.enable())
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <ucontext.h>

static struct sigaction old_segv, old_bus, old_abrt;
static char alt_stack[1 << 16];

static void put(const char* s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

static void put_hex(unsigned long v) {
    char b[19] = "0x0000000000000000";
    for (int i = 0; i < 16; ++i) b[17 - i] = "0123456789abcdef"[(v >> (4 * i)) & 15];
    put(b);
}

static void handler(int sig, siginfo_t* info, void* uctx) {
    put("\n=== segv_trace: signal ");
    put(sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : "SIGABRT");
    put(" fault address ");
    put_hex((unsigned long)info->si_addr);
    ucontext_t* uc = (ucontext_t*)uctx;
    put(" rip ");
    put_hex((unsigned long)uc->uc_mcontext.gregs[REG_RIP]);
    put(" ===\n");
    void* frames[64];
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    put("=== end of native stack ===\n");
    struct sigaction* old = sig == SIGSEGV ? &old_segv : sig == SIGBUS ? &old_bus : &old_abrt;
    if ((old->sa_flags & SA_SIGINFO) && old->sa_sigaction) {
        old->sa_sigaction(sig, info, uctx);
    } else if (old->sa_handler != SIG_DFL && old->sa_handler != SIG_IGN && old->sa_handler) {
        old->sa_handler(sig);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

void segv_trace_install(void) {
    void* warm[4];
    backtrace(warm, 4);     /* loads libgcc now: not safe to do that inside the handler */
    stack_t ss = {.ss_sp = alt_stack, .ss_size = sizeof alt_stack, .ss_flags = 0};
    sigaltstack(&ss, NULL);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    sigaction(SIGABRT, &sa, &old_abrt);
}
