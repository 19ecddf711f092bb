# tools/r06_misc.sh — the bench lines DESIGN §5 (round 6) quotes that have no profile directory of their own (GPU box) -> gpurun_out/r06_misc_runs.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_misc_runs.txt; : > $O
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; n=j.get('roofline_nodefer')
print('%-44s %.3f T  %.4f ms/step (min %.4f max %.4f)  hash %.4f fixup %.4f apply %.4f  frac %.3f' % ('$1', j['value']/1e12, j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max'], r['hash_ms'], r.get('fixup_ms') or 0, r['apply_ms'], r['frac']))
if n: print('%-44s %.3f T  %.4f ms/step  hash %.4f fixup %.4f apply %.4f  frac %.3f' % ('   roofline_nodefer (no NTC_FLAG_DEFER_REDO)', n['value']/1e12, n['ms_per_step'], n['hash_ms'], n['fixup_ms'], n['apply_ms'], n['frac']))
"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc"
$B 2>/dev/null | tail -1 | line "driver command (config 2, dist g)" >> $O
$B --no-nodefer --k 32 --gap 8 2>/dev/null | tail -1 | line "k = 32, gap 8 (config 5, second form)" >> $O
$B --no-nodefer --s-bits 11 2>/dev/null | tail -1 | line "sBits = 11, 20 steps" >> $O
$B --no-nodefer --klist 21,25,31 2>/dev/null | tail -1 | line "k list 21,25,31" >> $O
echo "-- ragged 100..150 bp, 10 M reads per batch, binned by ceil(len/16) into tiles on the device (tools/ragged_tiled_time.py)" >> $O
python tools/ragged_tiled_time.py 2>&1 | tail -4 >> $O
$B --no-nodefer --klist 16,24,32,48 2>/dev/null | tail -1 | line "k list 16,24,32,48 (K1h + K1 from the same tiles)" >> $O
$B --reads-per-step 1000000 2>/dev/null | tail -1 | line "1 M reads per step (+ its roofline_nodefer)" >> $O
cat $O
