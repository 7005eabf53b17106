/* The reference's long-option driver
 * (Cholesky_chameleon_VM/cho/Cholesky_Chameleon_sauv/code_c/v3_script_cholesky_x_arg_gpt.c) in plain C
 * on include/cholmi.h: the same 20 required options, the same type / uplo letters, the same three
 * output lines -- every CHAMELEON_* call replaced by its libcholmi export.  C99 + getopt_long only.
 *
 *   v3_driver --N 3000 --NB 256 --ncpu 4 --ngpu 1 --mat none --dtyp d --mb 256 --nb 256 --bsiz 65536 \
 *             --lm 3000 --ln 3000 --i 0 --j 0 --m 3000 --n 3000 --p 1 --q 1 --bump 3000 --uplo L --seed 51
 */
#define _POSIX_C_SOURCE 200809L
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cholmi.h"

static const char *NAMES[] = {"N", "NB", "ncpu", "ngpu", "mat", "dtyp", "mb", "nb", "bsiz", "lm", "ln", "i", "j",
                              "m", "n", "p", "q", "bump", "uplo", "seed"};
enum { NOPT = 20 };

static void usage(const char *prog) {
  fprintf(stderr,
          "Usage: %s --N INT --NB INT --ncpu INT --ngpu INT --mat none|user --dtyp d|s|z|c \\\n"
          "          --mb INT --nb INT --bsiz INT --lm INT --ln INT --i INT --j INT \\\n"
          "          --m INT --n INT --p INT --q INT --bump DOUBLE --uplo L|U|B --seed ULL\n\n"
          "ALL options are required. No defaults.\n",
          prog);
}

static int one_of(const char *s, const char *a, const char *b, const char *c) {
  return !strcmp(s, a) || !strcmp(s, b) || !strcmp(s, c);
}

int main(int argc, char **argv) {
  const char *val[NOPT] = {0};
  struct option lo[NOPT + 2];
  for (int k = 0; k < NOPT; ++k) {
    lo[k].name = NAMES[k];
    lo[k].has_arg = required_argument;
    lo[k].flag = 0;
    lo[k].val = 0;
  }
  lo[NOPT].name = "help", lo[NOPT].has_arg = no_argument, lo[NOPT].flag = 0, lo[NOPT].val = 'h';
  memset(&lo[NOPT + 1], 0, sizeof lo[0]);
  int opt, idx;
  while ((opt = getopt_long(argc, argv, "h", lo, &idx)) != -1) {
    if (opt == 'h') return usage(argv[0]), 0;
    if (opt != 0) return usage(argv[0]), 1;
    val[idx] = optarg;
  }
  for (int k = 0; k < NOPT; ++k)
    if (!val[k]) {
      fprintf(stderr, "Error: all options are required. Missing at least one.\n");
      return usage(argv[0]), 1;
    }
  const int N = atoi(val[0]), NB = atoi(val[1]), ncpu = atoi(val[2]), ngpu = atoi(val[3]), mb = atoi(val[6]),
            nb = atoi(val[7]), lm = atoi(val[9]), ln = atoi(val[10]), ioff = atoi(val[11]), joff = atoi(val[12]),
            m = atoi(val[13]), n = atoi(val[14]), p = atoi(val[15]), q = atoi(val[16]);
  const long bsiz = strtol(val[8], NULL, 10);
  const double bump = strtod(val[17], NULL);
  const unsigned long long seed = strtoull(val[19], NULL, 10);
  int dtyp, uplo;
  if (one_of(val[5], "d", "D", "0")) dtyp = CHOL_REAL_DOUBLE;
  else if (one_of(val[5], "s", "S", "1")) dtyp = CHOL_REAL_FLOAT;
  else if (one_of(val[5], "z", "Z", "2") || one_of(val[5], "c", "C", "3"))
    return fprintf(stderr, "Error: --dtyp z|c (complex) is not supported by this library\n"), 1;
  else return fprintf(stderr, "Error: invalid --dtyp %s\n", val[5]), 1;
  if (one_of(val[18], "L", "l", "0")) uplo = CHOL_LOWER;
  else if (one_of(val[18], "U", "u", "1")) uplo = CHOL_UPPER;
  else if (one_of(val[18], "B", "b", "2")) uplo = CHOL_UPPER_LOWER;
  else return fprintf(stderr, "Error: invalid --uplo %s\n", val[18]), 1;
  if (N <= 0 || NB <= 0 || mb <= 0 || nb <= 0 || lm <= 0 || ln <= 0 || m <= 0 || n <= 0 || p <= 0 || q <= 0)
    return fprintf(stderr, "Error: dimension arguments must be >0.\n"), 1;
  if (bsiz < (long)mb * nb) return fprintf(stderr, "Error: --bsiz < mb*nb (bsiz=%ld mb=%d nb=%d).\n", bsiz, mb, nb), 1;
  if (ioff < 0 || joff < 0 || ioff >= lm || joff >= ln)
    return fprintf(stderr, "Error: invalid offsets i=%d j=%d (lm=%d ln=%d).\n", ioff, joff, lm, ln), 1;
  if (ioff + m > lm || joff + n > ln)
    return fprintf(stderr, "Error: submatrix (i=%d,m=%d) outside lm=%d OR (j=%d,n=%d) outside ln=%d.\n", ioff, m, lm, joff, n, ln), 1;
  if (bump == 0.0) fprintf(stderr, "Warning: bump==0 -> matrix may not be SPD.\n");
  if (strcmp(val[4], "none") && strcmp(val[4], "NULL") && strcmp(val[4], "0"))
    return fprintf(stderr, "Error: --mat user: this C example keeps the matrix in HBM (use --mat none)\n"), 1;

  chol_desc_t *descA = NULL;
  int rc = chol_init(ncpu, ngpu);
  if (rc == 0) rc = chol_desc_create(&descA, NULL, dtyp, mb, nb, mb * nb, lm, ln, ioff, joff, m, n, p, q);
  if (rc == 0) rc = chol_plgsy_tile(bump, uplo, descA, seed);
  if (rc < 0) return fprintf(stderr, "Error: %s\n", chol_last_error()), 1;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  const int info = chol_potrf_tile(uplo, descA);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (info < 0) return fprintf(stderr, "Error: %s\n", chol_last_error()), 1;
  const double secs = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_nsec - t0.tv_nsec) / 1e9;
  const double dim = (double)(m < n ? m : n);
  printf("N=%d NB=%d ncpu=%d ngpu=%d p=%d q=%d bump=%g uplo=%d seed=%llu\n", N, NB, ncpu, ngpu, p, q, bump, uplo, seed);
  printf("Time: %.6f s\n", secs);
  printf("Performance: %.2f Gflop/s\n", (1.0 / 3.0) * dim * dim * dim / (secs * 1e9));
  if (info != 0) fprintf(stderr, "Erreur dans CHAMELEON_dpotrf_Tile: %d\n", info);
  chol_desc_destroy(&descA);
  chol_finalize();
  return info != 0;
}
