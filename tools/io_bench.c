/* io_bench.c - what the host side of a file-to-file encode can move: T threads writing a buffer into a tmpfs file with
 * pwrite, reading it back with pread, copying it out of an mmap and into an mmap of a fresh file (the ways the drivers touch big files),
 * making the pages of a fresh file with fallocate and writing into pages that exist.
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
        else if (p->mode == 3) memcpy(p->map + p->off + at, p->buf, m);
        else if (p->mode == 4) { if (fallocate(p->fd, 0, p->off + (off_t)at, (off_t)m)) perror("fallocate"); }   /* pages made, nothing copied */
        else if (p->mode == 5) { if (pwrite(p->fd, p->buf, m, p->off + (off_t)at) != (ssize_t)m) perror("pwrite"); }  /* into the pages mode 4 made */
        else memcpy(p->map + p->off + at, p->buf, m);                                                              /* modes 6, 7: through an mmap */
    }
    return NULL;
}
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const size_t total = (size_t)atol(argv[2]) << 30;
    for (int a = 3; a < argc; ++a) {
        const int T = atoi(argv[a]);
        for (int mode = 0; mode < 8; ++mode) {
            /* 6: mmap writes into the pages that exist (after 4/5); 7: a fresh file, ONE thread fallocates from the front while T threads write through the mmap */
            int fd = open(argv[1], mode == 0 || mode == 3 || mode == 4 || mode == 7 ? O_RDWR | O_CREAT | O_TRUNC : mode == 5 || mode == 6 ? O_RDWR : O_RDONLY, 0644);
            if ((mode == 3 || mode == 7) && ftruncate(fd, (off_t)total)) perror("ftruncate");
            const int wr = mode == 3 || mode == 6 || mode == 7;
            uint8_t *map = mode == 2 || wr ? (uint8_t *)mmap(NULL, total, wr ? PROT_READ | PROT_WRITE : PROT_READ, MAP_SHARED, fd, 0) : NULL;
            pthread_t fa; part_t fpart;
            if (mode == 7) { fpart.fd = fd; fpart.mode = 4; fpart.n = total; fpart.off = 0; fpart.buf = NULL; fpart.map = NULL; }
            pthread_t th[64];
            part_t part[64];
            for (int i = 0; i < T; ++i) {
                part[i].fd = fd; part[i].mode = mode; part[i].map = map; part[i].n = total / T; part[i].off = (off_t)(total / T) * i;
                part[i].buf = (uint8_t *)aligned_alloc(4096, 48u << 20);
                memset(part[i].buf, i + 1, 48u << 20);
            }
            const double t0 = now();
            if (mode == 7) pthread_create(&fa, NULL, run, &fpart);
            for (int i = 0; i < T; ++i) pthread_create(&th[i], NULL, run, &part[i]);
            for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
            if (mode == 7) pthread_join(fa, NULL);
            const double dt = now() - t0;
            printf("threads %2d %-6s %6.2f GB/s\n", T, mode == 0 ? "pwrite" : mode == 1 ? "pread" : mode == 2 ? "mmap" : mode == 3 ? "mmapwr" : mode == 4 ? "falloc" : mode == 5 ? "rewrite" : mode == 6 ? "mmap-rewrite" : "falloc+mmapwr", total / dt / 1e9);
            fflush(stdout);
            for (int i = 0; i < T; ++i) free(part[i].buf);
            if (map) munmap(map, total);
            close(fd);
        }
    }
    unlink(argv[1]);
    return 0;
}
