#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
for c in 256 192 320 384 512 768 1024 2048; do
HEXL_KS_CHUNK=$c python tools/ks_rate.py 8192 7 51 6 > $O/rate_chunk$c.txt 2>&1; echo "chunk=$c: $(tail -1 $O/rate_chunk$c.txt)"
done
for c in 256 512 1024; do
HEXL_KS_ONE_LANE=1 HEXL_KS_CHUNK=$c python tools/ks_rate.py 8192 7 51 6 > $O/rate_chunk${c}_onelane.txt 2>&1; echo "chunk=$c one lane: $(tail -1 $O/rate_chunk${c}_onelane.txt)"
done
for c in 256 512; do
HEXL_KS_CHUNK=$c python tools/ks_rate.py 1024 7 51 20 > $O/rate_chunk${c}_b1024.txt 2>&1; echo "chunk=$c batch 1024: $(tail -1 $O/rate_chunk${c}_b1024.txt)"
done
