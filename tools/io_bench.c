/* io_bench.c - what the host side of a file-to-file encode can move: T threads writing a buffer into a tmpfs file with
 * pwrite, reading it back with pread, copying it out of an mmap and into an mmap of a fresh file (the ways the drivers touch big files).
 *   io_bench <file> <GiB> <threads>...                                        (development aid, not part of the product) */
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

typedef struct { int fd, mode; uint8_t *buf; uint8_t *map; size_t n; off_t off; } part_t;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void *run(void *a) {
    part_t *p = (part_t *)a;
    const size_t slab = 48u << 20;
    for (size_t at = 0; at < p->n; at += slab) {
        const size_t m = p->n - at < slab ? p->n - at : slab;
        if (p->mode == 0) { if (pwrite(p->fd, p->buf, m, p->off + (off_t)at) != (ssize_t)m) perror("pwrite"); }
        else if (p->mode == 1) { if (pread(p->fd, p->buf, m, p->off + (off_t)at) != (ssize_t)m) perror("pread"); }
        else if (p->mode == 2) memcpy(p->buf, p->map + p->off + at, m);
        else memcpy(p->map + p->off + at, p->buf, m);
    }
    return NULL;
}
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const size_t total = (size_t)atol(argv[2]) << 30;
    for (int a = 3; a < argc; ++a) {
        const int T = atoi(argv[a]);
        for (int mode = 0; mode < 4; ++mode) {
            int fd = open(argv[1], mode == 0 || mode == 3 ? O_RDWR | O_CREAT | O_TRUNC : O_RDONLY, 0644);
            if (mode == 3 && ftruncate(fd, (off_t)total)) perror("ftruncate");
            uint8_t *map = mode >= 2 ? (uint8_t *)mmap(NULL, total, mode == 3 ? PROT_READ | PROT_WRITE : PROT_READ, MAP_SHARED, fd, 0) : NULL;
            pthread_t th[64];
            part_t part[64];
            for (int i = 0; i < T; ++i) {
                part[i].fd = fd; part[i].mode = mode; part[i].map = map; part[i].n = total / T; part[i].off = (off_t)(total / T) * i;
                part[i].buf = (uint8_t *)aligned_alloc(4096, 48u << 20);
                memset(part[i].buf, i + 1, 48u << 20);
            }
            const double t0 = now();
            for (int i = 0; i < T; ++i) pthread_create(&th[i], NULL, run, &part[i]);
            for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
            const double dt = now() - t0;
            printf("threads %2d %-6s %6.2f GB/s\n", T, mode == 0 ? "pwrite" : mode == 1 ? "pread" : mode == 2 ? "mmap" : "mmapwr", total / dt / 1e9);
            fflush(stdout);
            for (int i = 0; i < T; ++i) free(part[i].buf);
            if (map) munmap(map, total);
            close(fd);
        }
    }
    unlink(argv[1]);
    return 0;
}
