# tools/dbg/repro_cli_trim.sh — CLI vs reference binary on trimmed FASTQ (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/rt; mkdir -p $W; cd $W
python $ROOT/tools/make_fastq.py $W/s 2 ${1:-500000} > /dev/null
for f in $W/s_*.fq; do
  awk 'NR%4==1{print} NR%4==2{l=100+int((NR*7919)%51); print substr($0,1,l)} NR%4==3{print} NR%4==0{print substr($0,1,l)}' $f > ${f%.fq}.trim.fastq
done
for t in 1 2; do
  $ROOT/oracle/_ref/ntcard_ref -t $t -k 32 -p ref$t $W/s_*.trim.fastq > /dev/null
  $ROOT/ntcard_amd/bin/ntcard -t $t -k 32 -p gpu$t $W/s_*.trim.fastq > /dev/null
  cmp ref${t}_k32.hist gpu${t}_k32.hist && echo "t=$t IDENTICAL"
  head -3 ref${t}_k32.hist gpu${t}_k32.hist
done
NTC_BIN_MIN=100000000 $ROOT/ntcard_amd/bin/ntcard -t 1 -k 32 -p gpur $W/s_*.trim.fastq > /dev/null; cmp ref1_k32.hist gpur_k32.hist && echo "rows-only IDENTICAL"
