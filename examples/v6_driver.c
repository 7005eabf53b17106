/* A plain-C driver on include/cholmi.h: the call sequence of the reference's single-node
 * Chameleon driver (Cholesky_chameleon_VM/cho/docker_installation_and_bench_files/v6_test.c)
 * with every CHAMELEON_* call replaced by its libcholmi export, as INTEGRATION.md section 2
 * describes.  Same 16 positional arguments and the same output lines, so that the reference's
 * benchmark.c (which parses "Performance:" and the "||A - LL^T||" line) can launch it unchanged.
 * Built by examples/Makefile with a C compiler (no HIP, no C++): the ABI is C.
 *
 *   v6_driver ncpu ngpu N NB mb nb bsiz lm ln ioff joff m n p q seed
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "cholmi.h"

#define CK(call)                                                              \
  do {                                                                        \
    int rc_ = (call);                                                         \
    if (rc_ < 0) {                                                            \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, chol_last_error());       \
      return 2;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 17) {
    fprintf(stderr,
            "Usage: %s <num_cpu> <num_gpu> <matrix_size_N> <tile_size_NB> <mb> <nb> <bsiz> <lm> <ln> "
            "<ioff> <joff> <m> <n> <p> <q> <seed>\n",
            argv[0]);
    return 1;
  }
  int a[16];
  for (int i = 0; i < 16; ++i) a[i] = atoi(argv[i + 1]);
  const int ncpu = a[0], ngpu = a[1], N = a[2], NB = a[3], mb = a[4], nb = a[5], bsiz = a[6], lm = a[7],
            ln = a[8], ioff = a[9], joff = a[10], m = a[11], n = a[12], p = a[13], q = a[14], seed = a[15];
  printf("[setup] ncpu=%d ngpu=%d N=%d NB=%d backend=%s\n", ncpu, ngpu, N, NB, chol_version());

  chol_desc_t *descA = NULL, *descAorig = NULL, *descR = NULL;
  CK(chol_init(ncpu, ngpu));
  CK(chol_desc_create(&descA, NULL, CHOL_REAL_DOUBLE, mb, nb, bsiz, lm, ln, ioff, joff, m, n, p, q));
  CK(chol_plgsy_tile((double)N, CHOL_LOWER, descA, (unsigned long long)seed));
  CK(chol_desc_create(&descAorig, NULL, CHOL_REAL_DOUBLE, mb, nb, bsiz, lm, ln, ioff, joff, m, n, p, q));
  CK(chol_lacpy_tile(CHOL_UPPER_LOWER, descA, descAorig));

  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int info = chol_potrf_tile(CHOL_LOWER, descA);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (info < 0) {
    fprintf(stderr, "chol_potrf_tile -> %d: %s\n", info, chol_last_error());
    return 2;
  }
  const double secs = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_nsec - t0.tv_nsec) / 1e9;
  printf("N = %d, NB = %d\n", N, NB);
  printf("Time: %.3f s\n", secs);
  printf("Performance: %.2f Gflop/s\n", (1.0 / 3.0) * (double)N * (double)N * (double)N / (secs * 1e9));
  if (info != 0) fprintf(stderr, "Erreur dans chol_potrf_tile: %d\n", info);

  /* the validation block as the reference wrote it (it forms L^T L: not a residual) */
  double normA = 0.0, residual = 0.0;
  CK(chol_lange_tile(CHOL_INF_NORM, descAorig, &normA));
  CK(chol_desc_create(&descR, NULL, CHOL_REAL_DOUBLE, mb, nb, bsiz, lm, ln, ioff, joff, m, n, p, q));
  CK(chol_lacpy_tile(CHOL_LOWER, descA, descR));
  CK(chol_lauum_tile(CHOL_LOWER, descR));
  CK(chol_geadd_tile(CHOL_NOTRANS, -1.0, descR, 1.0, descAorig));
  CK(chol_lange_tile(CHOL_INF_NORM, descAorig, &residual));
  const double rel = residual / (normA > 0 ? normA : 1.0);
  printf("||A - LL^T||_inf / ||A||_inf = %.2e\n", rel);
  printf("Validation num\xc3\xa9rique : %s\n", rel < 1e-10 ? "PASS" : "FAIL");

  /* the check that block was meant to be, A regenerated on the device */
  double fro = -1.0, inf = -1.0;
  if (info == 0) {
    CK(chol_residual_plgsy(descA, (double)N, (unsigned long long)seed, &fro));
    CK(chol_residual_plgsy_inf(descA, (double)N, (unsigned long long)seed, &inf));
  }
  printf("[cholmi] ||A - L L^T||_F / ||A||_F = %.2e   ||A - L L^T||_inf / ||A||_inf = %.2e\n", fro, inf);

  CK(chol_desc_destroy(&descA));
  CK(chol_desc_destroy(&descAorig));
  CK(chol_desc_destroy(&descR));
  CK(chol_finalize());
  return info != 0;
}
