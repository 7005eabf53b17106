#!/bin/sh
# Builds oracle/_ref/libref_client.so from the reference's OWN source text.
#
# The reference client (client_distrib.cpp) cannot be compiled as a whole: it
# includes the ArmoniK SDK, gRPC, abseil and rapidjson, none of which exist in
# this image.  Its input-construction and parameter functions are
# dependency-free (C++ standard library only), so exactly those line ranges are
# streamed from the file where it lies under /root/reference straight into g++
# on stdin, between a standard-header preamble and extern "C" shims.  No reference
# source is written anywhere; the only output is the .so under oracle/_ref/
# (git-ignored).  Used to validate chol_oracle.c and to generate tests/golden/.
#
#   C2:41-93    struct Params, parse_int_str, load_params
#   C2:224-264  make_spd_like_chameleon, enforce_strict_diag_dominance
#   C2:280-321  extract_block_from_spd_matrix_colmajor, block_id_from_ij
set -e
REF=${REFERENCE_ROOT:-/root/reference}
SRC="$REF/cholesky_armonik/w_c_cons_v2/client_construction2/client/src/client_distrib.cpp"
HERE=$(cd "$(dirname "$0")" && pwd)
[ -f "$SRC" ] || { echo "build_ref: $SRC not found, skipping"; exit 0; }
mkdir -p "$HERE/_ref"
{
cat <<'PRE'
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <optional>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
PRE
sed -n '41,93p;224,264p;280,321p' "$SRC"
cat <<'POST'
extern "C" {
void ref_make_spd_like_chameleon(double* A, int N, int LDA, double bump, char uplo, std::uint64_t seed)
{ make_spd_like_chameleon(A, N, LDA, bump, uplo, seed); }
void ref_enforce_strict_diag_dominance(double* A, int N, int LDA)
{ enforce_strict_diag_dominance(A, N, LDA); }
void ref_enforce_strict_diag_dominance_eps(double* A, int N, int LDA, double eps)
{ enforce_strict_diag_dominance(A, N, LDA, eps); }
void ref_extract_block(const double* A, int N, int LDA, int B, int bi, int bj, double* out)
{ std::vector<double> blk; extract_block_from_spd_matrix_colmajor(A, N, LDA, B, bi, bj, blk);
  std::memcpy(out, blk.data(), blk.size() * sizeof(double)); }
int ref_block_id_from_ij(int i, int j, char* out, int cap)
{ std::string s = block_id_from_ij(i, j); if ((int)s.size() + 1 > cap) return -1;
  std::memcpy(out, s.c_str(), s.size() + 1); return (int)s.size(); }
int ref_parse_int_str(const char* s, int fallback)
{ return parse_int_str(std::string(s), fallback, "ref"); }
/* argv[0] is the program name, as in main() */
int ref_load_params(int argc, char** argv, int* N, int* B)
{ try { Params p = load_params(argc, argv); *N = p.N; *B = p.B; return 0; } catch (...) { return 1; } }
}
POST
} | g++ -std=gnu++17 -O2 -fPIC -shared -Wno-unused-function -x c++ - -o "$HERE/_ref/libref_client.so"
echo "build_ref: built $HERE/_ref/libref_client.so"
