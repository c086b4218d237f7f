// sigprof.c — a sampling profiler for where a call's CPU time goes when the box has no perf (round 6): ITIMER_PROF delivers SIGPROF to whichever thread is
// burning CPU; the handler keeps the top frames; sp_report() names them with dladdr.   gcc -O2 -shared -fPIC -o libsigprof.so sigprof.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#define DEPTH 12
#define CAP (1 << 18)
static void* g_frames[CAP][DEPTH];
static int g_n[CAP];
static volatile int g_count = 0, g_on = 0;
static void handler(int sig, siginfo_t* si, void* uc) {
  (void)sig; (void)si; (void)uc;
  if (!g_on) return;
  int i = __sync_fetch_and_add(&g_count, 1);
  if (i >= CAP) return;
  g_n[i] = backtrace(g_frames[i], DEPTH);
}
void sp_start(int hz) {
  void* warm[4]; backtrace(warm, 4);   // (loads the unwinder outside the handler)
  struct sigaction sa; memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigemptyset(&sa.sa_mask);
  sigaction(SIGPROF, &sa, NULL);
  g_count = 0; g_on = 1;
  struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000000 / hz; it.it_value = it.it_interval;
  setitimer(ITIMER_PROF, &it, NULL);
}
int sp_stop(void) {
  struct itimerval it; memset(&it, 0, sizeof it); setitimer(ITIMER_PROF, &it, NULL);
  g_on = 0;
  return g_count < CAP ? g_count : CAP;
}
// writes "module\tsymbol" of frame `depth` (0 = innermost after the handler's own two frames) of every sample, one line per sample, frames joined by " < "
void sp_dump(const char* path, int frames) {
  FILE* f = fopen(path, "w");
  if (!f) return;
  int n = g_count < CAP ? g_count : CAP;
  for (int i = 0; i < n; ++i) {
    int printed = 0;
    for (int d = 2; d < g_n[i] && printed < frames; ++d) {   // (0: handler, 1: the signal trampoline)
      Dl_info info;
      const char* mod = "?"; const char* sym = "?"; uintptr_t off = 0;
      if (dladdr(g_frames[i][d], &info)) {
        if (info.dli_fname) { const char* s = strrchr(info.dli_fname, '/'); mod = s ? s + 1 : info.dli_fname; }
        if (info.dli_sname) sym = info.dli_sname;
        off = (uintptr_t)g_frames[i][d] - (uintptr_t)info.dli_fbase;
      }
      fprintf(f, "%s%s!%s+0x%lx", printed ? " < " : "", mod, sym, (unsigned long)off);
      ++printed;
    }
    fprintf(f, "\n");
  }
  fclose(f);
}
