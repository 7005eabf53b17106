#!/bin/bash
# A/B of two builds of the library on the same box: scripts/exp/ab_lib.sh cfg...   (libcholmi_old.so / libcholmi_new.so beside libcholmi.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  for v in old new old new; do
    cp dense_linear_app_amd/libcholmi_$v.so dense_linear_app_amd/libcholmi.so
    echo -n "$v "; bash scripts/ab.sh "CHOLMI_X=1" $cfg | sed 's/CHOLMI_X=1 //'
  done
done
cp dense_linear_app_amd/libcholmi_new.so dense_linear_app_amd/libcholmi.so
