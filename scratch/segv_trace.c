/* LD_PRELOAD helper for the exit-fault forensics (round 4): prints the native call stack of a SIGSEGV / SIGBUS / SIGABRT
 * to stderr (glibc backtrace_symbols_fd: async-signal-safe enough for a dying process), then dies with the default action.
 * build: gcc -O1 -g -shared -fPIC scratch/segv_trace.c -o gpurun_out/libsegv_trace.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  static const char head[] = "\n[segv_trace] fatal signal; faulting address and native stack follow\n";
  write(2, head, sizeof(head) - 1);
  char buf[64]; int n = 0; unsigned long a = (unsigned long)si->si_addr;
  buf[n++] = 's'; buf[n++] = 'i'; buf[n++] = 'g'; buf[n++] = '='; buf[n++] = '0' + sig / 10; buf[n++] = '0' + sig % 10;
  buf[n++] = ' '; buf[n++] = 'a'; buf[n++] = 'd'; buf[n++] = 'd'; buf[n++] = 'r'; buf[n++] = '='; buf[n++] = '0'; buf[n++] = 'x';
  for (int s = 60; s >= 0; s -= 4) buf[n++] = "0123456789abcdef"[(a >> s) & 15];
  buf[n++] = '\n';
  write(2, buf, n);
  void* frames[64];
  int k = backtrace(frames, 64);
  backtrace_symbols_fd(frames, k, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  static char stack[1 << 16];
  stack_t ss; ss.ss_sp = stack; ss.ss_size = sizeof(stack); ss.ss_flags = 0;
  sigaltstack(&ss, NULL);
  struct sigaction sa; memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, NULL); sigaction(SIGBUS, &sa, NULL); sigaction(SIGABRT, &sa, NULL);
  void* warm[2]; backtrace(warm, 2);   /* loads libgcc now, not inside the handler */
}
