#!/bin/bash
# wall time of the reference's own test program (test/test.c, unchanged, 876 VIO rows with a pose each + 1006 ranges):
# compiled reference on one CPU core vs the same program linked against libparticlefilter.so
# (oracle/_ref/ is built by oracle/Makefile from /root/reference)
D=$GRAFT_REPO_ROOT/tests/golden/test1/
t() { local s=$(date +%s%N); "${@:2}" > /tmp/td.log 2>&1; local rc=$?; local e=$(date +%s%N); echo "$1 rc=$rc $(( (e - s) / 1000000 )) ms: $(tail -1 /tmp/td.log)"; }
t ref_cpu_n10k oracle/_ref/ref_test_glibc --nofail $D /tmp/out_cpu.csv
t hip_n10k timeout -k 10 100 oracle/_ref/ref_test_hip --nofail $D /tmp/out_hip.csv
export PF_HIP_N_TAG_LOC=1000000
t hip_n1M timeout -k 10 100 oracle/_ref/ref_test_hip --nofail $D /tmp/out_hip2.csv
export PF_HIP_N_TAG_LOC=16000000
t hip_n16M timeout -k 10 100 oracle/_ref/ref_test_hip --nofail $D /tmp/out_hip3.csv
