ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for n in ${AB_LIBS:-cur}; do
  if [ "$n" = cur ]; then L=""; else L="--lib $ROOT/tools/lib_$n.so"; fi
  python $ROOT/bench.py --no-cpu-baseline --no-live-pmc --steps 20 --warmup 5 --repeats 3 $L | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); n=j['roofline_nodefer']; print('%-8s nodefer step %.4f ms  hash %.4f fix %.4f apply %.4f frac %.3f | defer step %.4f fix %.4f' % ('$n',n['ms_per_step'],n['hash_ms'],n['fixup_ms'],n['apply_ms'],n['frac'],j['ms_per_step'],j['roofline']['fixup_ms']))"
done
