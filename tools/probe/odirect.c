// How fast can ONE file take 16-27 GB of fresh row blocks (the config-5 close)?  T threads write `total` bytes of one file
// in 64 MiB pieces at disjoint offsets, in six ways:
//   buffered      pwrite on one descriptor                      (what smx_file.inc does: 8.7-9.4 GB/s, round 3)
//   O_DIRECT      the same, bypassing the page cache            (7.1-7.4 GB/s, round 3: DESIGN rejected #32)
//   multi-fd      every thread opens the file itself            (round 5: is the plateau the per-descriptor or the per-inode lock?)
//   fallocate     fallocate() the whole range first, then pwrite (no block allocation inside the writes)
//   mmap          ftruncate + mmap(MAP_SHARED), threads memcpy   (no write() path at all: page faults instead of the inode lock)
//   mmap+falloc   fallocate + mmap(MAP_SHARED | MAP_POPULATE off), threads memcpy
//   gcc -O2 -pthread tools/probe/odirect.c -o /tmp/odirect && /tmp/odirect /tmp/odirect.bin 16 16
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static const size_t PIECE = 64ull << 20;
static int g_fd; static size_t g_total; static int g_threads; static char* g_buf; static const char* g_path; static int g_mode; static char* g_map;
enum { M_BUFFERED, M_DIRECT, M_MULTIFD, M_FALLOC, M_MMAP, M_MMAP_FALLOC, M_COUNT };
static const char* NAME[] = {"buffered", "O_DIRECT", "multi-fd", "fallocate", "mmap", "mmap+falloc"};
static void* worker(void* a) {
  size_t id = (size_t)a;
  int fd = g_fd;
  if (g_mode == M_MULTIFD) { fd = open(g_path, O_RDWR); if (fd < 0) { perror("open"); exit(1); } }
  for (size_t off = id * PIECE; off < g_total; off += (size_t)g_threads * PIECE) {
    const char* src = g_buf + (id % 4) * PIECE;
    if (g_mode == M_MMAP || g_mode == M_MMAP_FALLOC) { memcpy(g_map + off, src, PIECE); continue; }
    size_t done = 0;
    while (done < PIECE) {
      ssize_t w = pwrite(fd, src + done, PIECE - done, (off_t)(off + done));
      if (w <= 0) { perror("pwrite"); exit(1); }
      done += (size_t)w;
    }
  }
  if (g_mode == M_MULTIFD) close(fd);
  return NULL;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: odirect <file> <GiB> <threads>\n"); return 2; }
  g_path = argv[1]; size_t gib = (size_t)atoi(argv[2]); g_threads = atoi(argv[3]);
  g_total = gib << 30;
  if (posix_memalign((void**)&g_buf, 4096, 4 * PIECE)) return 1;
  memset(g_buf, 0x5a, 4 * PIECE);
  for (g_mode = 0; g_mode < M_COUNT; g_mode++) {
    for (int rep = 0; rep < 2; rep++) {
      unlink(g_path);
      g_fd = open(g_path, O_CREAT | O_RDWR | (g_mode == M_DIRECT ? O_DIRECT : 0), 0644);
      if (g_fd < 0) { perror(NAME[g_mode]); break; }
      double t0 = now();
      if (g_mode == M_FALLOC || g_mode == M_MMAP_FALLOC) { if (fallocate(g_fd, 0, 0, (off_t)g_total) != 0) { perror("fallocate"); close(g_fd); break; } }
      if (g_mode == M_MMAP && ftruncate(g_fd, (off_t)g_total) != 0) { perror("ftruncate"); close(g_fd); break; }
      if (g_mode == M_MMAP || g_mode == M_MMAP_FALLOC) {
        g_map = mmap(NULL, g_total, PROT_READ | PROT_WRITE, MAP_SHARED, g_fd, 0);
        if (g_map == MAP_FAILED) { perror("mmap"); close(g_fd); break; }
      }
      double t_prep = now();
      pthread_t th[256];
      for (size_t i = 0; i < (size_t)g_threads; i++) pthread_create(&th[i], NULL, worker, (void*)i);
      for (int i = 0; i < g_threads; i++) pthread_join(th[i], NULL);
      double t1 = now();
      if (g_mode == M_MMAP || g_mode == M_MMAP_FALLOC) munmap(g_map, g_total);
      close(g_fd);
      double t2 = now();
      printf("%-11s rep %d: %zu GiB by %d threads: %.2f s = %.2f GB/s (set-up %.2f s, unmap/close +%.2f s)\n", NAME[g_mode], rep, gib, g_threads,
             t1 - t0, g_total / 1e9 / (t1 - t0), t_prep - t0, t2 - t1);
      fflush(stdout);
    }
  }
  unlink(g_path);
  return 0;
}
