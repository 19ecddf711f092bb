# tools/prof_r06.sh — round 6's profile set at HEAD (GPU box): the unprofiled bench lines of every tag first (a bench run right behind PMC passes
# has measured slow), then the rocprofv3 kernel trace + PMC passes.  tools/collect_prof.py <tag> r06 copies the summaries into profiles/.
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; PROF_DEFAULT=1 PROF_PHASE=$PH bash tools/prof.sh $tag "$@" > gpurun_out/prof_${tag}_$PH.log 2>&1; }
for PH in bench counters; do
  run driver --steps 20 --warmup 5
  run dist_u --steps 20 --warmup 5 --dist u
  run cfg3 --config 3
  run cfg4 --config 4
  run cfg5 --config 5
  run default
done
for t in driver dist_u cfg3 cfg4 cfg5 default; do cut -c1-300 gpurun_out/prof_$t/bench.json; echo; done
