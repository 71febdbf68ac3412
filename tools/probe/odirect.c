// Is O_DIRECT worth it for the config-5 close (27 GB of row blocks, 6.5 GB/s through the page cache)?
// T threads write `total` bytes of one file in 64 MiB pieces at disjoint offsets, buffered or direct.
//   gcc -O2 -pthread tools/probe/odirect.c -o /tmp/odirect && /tmp/odirect /tmp/odirect.bin 16 16
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static const size_t PIECE = 64ull << 20;
static int g_fd; static size_t g_total; static int g_threads; static char* g_buf;
static void* worker(void* a) {
  size_t id = (size_t)a;
  for (size_t off = id * PIECE; off < g_total; off += (size_t)g_threads * PIECE) {
    size_t done = 0;
    while (done < PIECE) {
      ssize_t w = pwrite(g_fd, g_buf + (id % 4) * PIECE + done, PIECE - done, (off_t)(off + done));
      if (w <= 0) { perror("pwrite"); exit(1); }
      done += (size_t)w;
    }
  }
  return NULL;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char** argv) {
  const char* path = argv[1]; size_t gib = (size_t)atoi(argv[2]); g_threads = atoi(argv[3]);
  g_total = gib << 30;
  if (posix_memalign((void**)&g_buf, 4096, 4 * PIECE)) return 1;
  memset(g_buf, 0x5a, 4 * PIECE);
  for (int direct = 0; direct < 2; direct++) {
    for (int rep = 0; rep < 2; rep++) {
      unlink(path);
      g_fd = open(path, O_CREAT | O_RDWR | (direct ? O_DIRECT : 0), 0644);
      if (g_fd < 0) { perror(direct ? "open O_DIRECT" : "open"); break; }
      double t0 = now();
      pthread_t th[256];
      for (size_t i = 0; i < (size_t)g_threads; i++) pthread_create(&th[i], NULL, worker, (void*)i);
      for (int i = 0; i < g_threads; i++) pthread_join(th[i], NULL);
      double t1 = now();
      close(g_fd);
      double t2 = now();
      printf("%s rep %d: %zu GiB by %d threads: %.2f s = %.2f GB/s (close +%.2f s)\n", direct ? "O_DIRECT" : "buffered", rep, gib, g_threads,
             t1 - t0, g_total / 1e9 / (t1 - t0), t2 - t1);
      fflush(stdout);
    }
  }
  unlink(path);
  return 0;
}
