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
echo "== ONE file holding all the reads (round 6: the threads without a file of their own read and split its blocks), -t 8 and -t 1"
cat $W/s_*.fq > $W/all.fastq; ls -la $W/all.fastq
echo "-- reference -t 8"; time $ROOT/oracle/_ref/ntcard_ref -t 8 -k 32 -p ref_one $W/all.fastq
echo "-- MI355X -t 8"; time $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32 -p gpu_one $W/all.fastq
cmp ref_one_k32.hist gpu_one_k32.hist && echo IDENTICAL
echo "-- MI355X -t 1"; time $ROOT/ntcard_amd/bin/ntcard -t 1 -k 32 -p gpu_one1 $W/all.fastq
cmp ref_one_k32.hist gpu_one1_k32.hist && echo IDENTICAL
rm -f $W/all.fastq
echo "== multi-k 16,24,32,48 -t 8"
time $ROOT/oracle/_ref/ntcard_ref -t 8 -k 16,24,32,48 -p refm $W/s_*.fq
time $ROOT/ntcard_amd/bin/ntcard -t 8 -k 16,24,32,48 -p gpum $W/s_*.fq
for k in 16 24 32 48; do cmp refm_k$k.hist gpum_k$k.hist && echo IDENTICAL k$k; done
echo "== FASTA (one-line records) and SAM renderings of the same reads, -t 8 (block splitters since round 5)"
for f in $W/s_*.fq; do
  awk 'NR%4==1{print ">" substr($0,2)} NR%4==2{print}' $f > ${f%.fq}.fa
  awk 'BEGIN{OFS="\t"; print "@HD","VN:1.6"} NR%4==1{n=substr($0,2)} NR%4==2{s=$0} NR%4==0{print n,4,"*",0,0,"*","*",0,0,s,$0}' $f > ${f%.fq}.sam
done
for ext in fa sam; do
  echo "-- reference, .$ext"; time $ROOT/oracle/_ref/ntcard_ref -t 8 -k 32 -p ref_$ext $W/s_*.$ext
  echo "-- MI355X, .$ext"; time $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32 -p gpu_$ext $W/s_*.$ext
  cmp ref_${ext}_k32.hist gpu_${ext}_k32.hist && echo IDENTICAL
done
echo "== adapter-trimmed rendering: every read cut to 100 .. 150 bases (ragged tiles, all length bins in one launch: round 5), -t 8"
for f in $W/s_*.fq; do
  awk 'NR%4==1{print} NR%4==2{l=100+int((NR*7919)%51); print substr($0,1,l)} NR%4==3{print} NR%4==0{print substr($0,1,l)}' $f > ${f%.fq}.trim.fastq
done
echo "-- reference"; time $ROOT/oracle/_ref/ntcard_ref -t 8 -k 32 -p ref_trim $W/s_*.trim.fastq
echo "-- MI355X"; time $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32 -p gpu_trim $W/s_*.trim.fastq
cmp ref_trim_k32.hist gpu_trim_k32.hist && echo IDENTICAL
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/e2e_prof_trim
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_prof_trim -o t -- $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32 -p $W/proft $W/s_*.trim.fastq > /tmp/e2e_prof_trim.log 2>&1
f=$(find /tmp/e2e_prof_trim -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -6 "$f" | cut -c1-150
cd $W
echo "== the trimmed reads under a mixed k list, -k 32,64 -t 8 (round 6: the length bins stay on tiles — K1h + K1f for k = 32, K1 staging the same tiles for k = 64)"
echo "-- reference"; time $ROOT/oracle/_ref/ntcard_ref -t 8 -k 32,64 -p ref_trim2 $W/s_*.trim.fastq
echo "-- MI355X"; time $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32,64 -p gpu_trim2 $W/s_*.trim.fastq
for k in 32 64; do cmp ref_trim2_k$k.hist gpu_trim2_k$k.hist && echo IDENTICAL k$k; done
cd /tmp && rm -rf /tmp/e2e_prof_trim2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_prof_trim2 -o t -- $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32,64 -p $W/proft2 $W/s_*.trim.fastq > /tmp/e2e_prof_trim2.log 2>&1
f=$(find /tmp/e2e_prof_trim2 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -7 "$f" | cut -c1-150
cd $W
echo "== kernels of one CLI run (rocprofv3 --kernel-trace --stats: which hash kernel the parsed reads reach)"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/e2e_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_prof -o t -- $ROOT/ntcard_amd/bin/ntcard -t 8 -k 32 -p $W/prof $W/s_*.fq > /tmp/e2e_prof.log 2>&1
f=$(find /tmp/e2e_prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %5s  total %9.3f ms" % (row["Name"][:70], row["Calls"], float(row["TotalDurationNs"]) / 1e6))
PY
