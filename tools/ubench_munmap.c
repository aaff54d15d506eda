/* tools/ubench_munmap.c — what does releasing KMC's stage-2 arena cost on this host? mmap 14 GB anonymous, touch 2.3 GB of it in 3.4 MB pieces (512 bin images +
 * their outputs), munmap; with and without MADV_HUGEPAGE, with 1 and 8 touching threads. gcc -O2 -pthread tools/ubench_munmap.c -o tools/ubench_munmap.bin */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static char *base; static size_t piece = 3400000, stride = 26000000; static int n_pieces = 512, n_thr = 1;
static size_t g_total; static int g_zap_thr;
static void *zapper(void *arg) { long t = (long)arg; size_t per = (g_total / g_zap_thr) & ~((size_t)(2 << 20) - 1); madvise(base + t * per, t == g_zap_thr - 1 ? g_total - t * per : per, MADV_DONTNEED); return 0; }
static void *toucher(void *arg) { long t = (long)arg; for (int i = (int)t; i < n_pieces; i += n_thr) memset(base + (size_t)i * stride, 1, piece + 1100000); return 0; }
int main(void)
{
	const size_t total = (size_t)14 << 30;
	for (int huge = 0; huge < 2; ++huge)
		for (n_thr = 1; n_thr <= 8; n_thr *= 8) {
			base = mmap(0, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
			if (base == MAP_FAILED) { perror("mmap"); return 1; }
			if (huge) madvise(base, total, MADV_HUGEPAGE);
			double t0 = now();
			pthread_t th[8];
			for (long t = 0; t < n_thr; ++t) pthread_create(&th[t], 0, toucher, (void *)t);
			for (int t = 0; t < n_thr; ++t) pthread_join(th[t], 0);
			double t1 = now();
			for (g_zap_thr = 0; g_zap_thr <= 16; g_zap_thr = g_zap_thr ? g_zap_thr * 4 : 1) { /* 0: no zapping; 1, 4, 16 threads of MADV_DONTNEED before the munmap */
				if (g_zap_thr) { /* touch again for this variant */
					for (long t = 0; t < n_thr; ++t) pthread_create(&th[t], 0, toucher, (void *)t);
					for (int t = 0; t < n_thr; ++t) pthread_join(th[t], 0);
				}
				double z0 = now();
				pthread_t zt[16];
				g_total = total;
				for (long t = 0; t < g_zap_thr; ++t) pthread_create(&zt[t], 0, zapper, (void *)t);
				for (int t = 0; t < g_zap_thr; ++t) pthread_join(zt[t], 0);
				double z1 = now();
				if (g_zap_thr == 16 || g_zap_thr == 0) {
					if (g_zap_thr == 0) { munmap(base, total); double t2 = now(); printf("hugepage advice %d, %d toucher threads: touch 2.3 GB %.3f s, munmap %.3f s\n", huge, n_thr, t1 - t0, t2 - z1);
						base = mmap(0, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if (huge) madvise(base, total, MADV_HUGEPAGE); }
					else { munmap(base, total); double t2 = now(); printf("   zap with 16 threads %.3f s, then munmap %.3f s\n", z1 - z0, t2 - z1); }
				} else
					printf("   zap with %d threads %.3f s\n", g_zap_thr, z1 - z0);
			}
		}
	return 0;
}
