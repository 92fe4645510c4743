// What does reading one small WAV file cost on this box, per syscall and per strategy, at 1 .. 32 threads?
//   gcc -O2 -pthread tools/probe/file_ingest_probe.c -o /tmp/file_ingest_probe && /tmp/file_ingest_probe [dir] [n_files]
// (Round 4: the from-files leg of bench.py sits at ~150 k files/s on the GPU box = ~100 us of thread time per 44 KB file,
//  against ~14 us in the build container: where does it go?)  Strategies, each over the same n files in one directory, page cache warm:
//   A  open(path) + read(4 KB) + pread(rest) + close      (the reader of rounds 1 - 3)
//   B  openat(dirfd, name) + preadv(header 44 B | payload straight into the row) + close
//   C  openat + close only                                  (the price of the two syscalls every strategy needs)
//   D  fstatat only
//   E  B without close (descriptors closed afterwards in one go: close() itself)
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

static int n_files = 8000, n_samples = 22050, dirfd_g = -1;
static char dir_g[512];
static char (*names)[32];
static int16_t* rows;
static atomic_int next_i;
static int strategy;
static int* fds;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void* work(void* arg) {
    (void)arg;
    unsigned char hdr[4096];
    char path[640];
    for (;;) {
        const int i = atomic_fetch_add(&next_i, 1);
        if (i >= n_files) break;
        int16_t* out = rows + (size_t)i * n_samples;
        if (strategy == 'A') {
            snprintf(path, sizeof path, "%s/%s", dir_g, names[i]);
            const int fd = open(path, O_RDONLY | O_CLOEXEC);
            if (fd < 0) continue;
            const ssize_t have = read(fd, hdr, sizeof hdr);
            if (have > 44) memcpy(out, hdr + 44, have - 44);
            if (pread(fd, (char*)out + (have - 44), (size_t)n_samples * 2 - (have - 44), have) < 0) {}
            close(fd);
        } else if (strategy == 'B' || strategy == 'E') {
            const int fd = openat(dirfd_g, names[i], O_RDONLY | O_CLOEXEC);
            if (fd < 0) continue;
            struct iovec iov[2] = {{hdr, 44}, {out, (size_t)n_samples * 2}};
            if (preadv(fd, iov, 2, 0) < 0) {}
            if (strategy == 'B') close(fd); else fds[i] = fd;
        } else if (strategy == 'C') {
            const int fd = openat(dirfd_g, names[i], O_RDONLY | O_CLOEXEC);
            if (fd >= 0) close(fd);
        } else if (strategy == 'D') {
            struct stat st;
            if (fstatat(dirfd_g, names[i], &st, 0) < 0) {}
        }
    }
    return NULL;
}

int main(int argc, char** argv) {
    snprintf(dir_g, sizeof dir_g, "%s", argc > 1 ? argv[1] : "/tmp/file_ingest_probe_dir");
    if (argc > 2) n_files = atoi(argv[2]);
    mkdir(dir_g, 0755);
    names = malloc((size_t)n_files * 32);
    rows = malloc((size_t)n_files * n_samples * 2);
    fds = malloc((size_t)n_files * sizeof(int));
    unsigned char* file = calloc(1, 44 + (size_t)n_samples * 2);
    memcpy(file, "RIFF", 4); memcpy(file + 8, "WAVEfmt ", 8); memcpy(file + 36, "data", 4);
    for (int i = 0; i < n_files; ++i) {
        snprintf(names[i], 32, "chunk_%06d.wav", i);
        char path[640];
        snprintf(path, sizeof path, "%s/%s", dir_g, names[i]);
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror(path); return 1; }
        if (write(fd, file, 44 + (size_t)n_samples * 2) < 0) {}
        close(fd);
    }
    dirfd_g = open(dir_g, O_RDONLY | O_DIRECTORY);
    memset(rows, 1, (size_t)n_files * n_samples * 2);      // pages touched
    const char strategies[] = "ABCDE";
    const int threads[] = {1, 4, 8, 16, 32};
    printf("%d files of %d bytes in %s; files/s (us of thread time per file)\n", n_files, 44 + n_samples * 2, dir_g);
    for (int s = 0; strategies[s]; ++s) {
        strategy = strategies[s];
        printf("  %c:", strategy);
        for (int k = 0; k < 5; ++k) {
            const int T = threads[k];
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                atomic_store(&next_i, 0);
                pthread_t th[32];
                const double t0 = now();
                for (int t = 1; t < T; ++t) pthread_create(&th[t], NULL, work, NULL);
                work(NULL);
                for (int t = 1; t < T; ++t) pthread_join(th[t], NULL);
                const double dt = now() - t0;
                if (dt < best) best = dt;
                if (strategy == 'E') { const double c0 = now(); for (int i = 0; i < n_files; ++i) close(fds[i]); if (rep == 2 && k == 0) printf(" [close alone: %.1f us each]", (now() - c0) / n_files * 1e6); }
            }
            printf("  T=%-2d %8.0f (%5.1f)", T, n_files / best, best * T / n_files * 1e6);
        }
        printf("\n");
    }
    return 0;
}
