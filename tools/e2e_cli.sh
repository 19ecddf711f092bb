#!/bin/bash
# tools/e2e_cli.sh — end-to-end wall clock of the drop-in CLI vs the real reference binary on the GPU box's host
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TIMEFORMAT="%R s wall, %U s user, %S s sys"
W=/tmp/e2e; mkdir -p $W; cd $W
python $ROOT/tools/make_fastq.py $W/s 8 ${E2E_READS:-2000000}
ls -la $W/*.fq | head -3
nproc
for t in 1 8; do
  echo "== reference ntcard -t $t"; time $ROOT/oracle/_ref/ntcard_ref -t $t -k 32 -p ref$t $W/s_*.fq
  echo "== MI355X ntcard -t $t"; time $ROOT/ntcard_amd/bin/ntcard -t $t -k 32 -p gpu$t $W/s_*.fq
  cmp ref${t}_k32.hist gpu${t}_k32.hist && echo IDENTICAL
done
echo "== multi-k 16,24,32,48 -t 8"
time $ROOT/oracle/_ref/ntcard_ref -t 8 -k 16,24,32,48 -p refm $W/s_*.fq
time $ROOT/ntcard_amd/bin/ntcard -t 8 -k 16,24,32,48 -p gpum $W/s_*.fq
for k in 16 24 32 48; do cmp refm_k$k.hist gpum_k$k.hist && echo IDENTICAL k$k; done
