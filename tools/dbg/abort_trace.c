// LD_PRELOAD helper: print the native stack of the thread that raises SIGABRT (teardown crashes under torch.distributed.run).
//   gcc -shared -fPIC -o abort_trace.so abort_trace.c ; LD_PRELOAD=$PWD/abort_trace.so python ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
static void on_abort(int sig) {
  void* frames[64];
  const char msg[] = "---- abort_trace: native stack of the aborting thread ----\n";
  write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}
__attribute__((constructor)) static void install(void) {
  void* warm[4];
  backtrace(warm, 4);   // loads libgcc now, not inside the handler
  signal(SIGABRT, on_abort);
}
