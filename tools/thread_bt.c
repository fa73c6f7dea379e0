// Which runtime component is a spinning thread in?  A SIGUSR2 handler that prints the interrupted thread's backtrace (library +
// offset per frame, dladdr) -- loaded into the Python process with ctypes by tools/host_spin_probe.py, which sends the signal to
// the hottest non-main thread (tgkill).    gcc -O1 -g -shared -fPIC -o /tmp/thread_bt.so tools/thread_bt.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* uc) {
  void* frames[48];
  const int n = backtrace(frames, 48);
  dprintf(2, "== backtrace of tid %ld (%d frames)\n", (long)syscall(SYS_gettid), n);
  for (int i = 0; i < n; ++i) {
    Dl_info info;
    if (dladdr(frames[i], &info) && info.dli_fname) {
      const char* base = strrchr(info.dli_fname, '/');
      dprintf(2, "  #%02d %s +0x%lx %s\n", i, base ? base + 1 : info.dli_fname, (unsigned long)((char*)frames[i] - (char*)info.dli_fbase),
              info.dli_sname ? info.dli_sname : "");
    } else {
      dprintf(2, "  #%02d %p\n", i, frames[i]);
    }
  }
}

void ca_bt_install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGUSR2, &sa, NULL);
}

int ca_bt_signal(int tid) { return (int)syscall(SYS_tgkill, getpid(), tid, SIGUSR2); }
