cd $GRAFT_REPO_ROOT
PROF_DEFAULT=1 bash tools/prof.sh default > gpurun_out/prof_default.log 2>&1
PROF_DEFAULT=1 bash tools/prof.sh dist_u --dist u > gpurun_out/prof_dist_u.log 2>&1
PROF_DEFAULT=1 bash tools/prof.sh cfg4 --config 4 > gpurun_out/prof_cfg4.log 2>&1
PROF_DEFAULT=1 bash tools/prof.sh cfg5 --config 5 > gpurun_out/prof_cfg5.log 2>&1
PROF_DEFAULT=1 bash tools/prof.sh lane_u --dist u --lane-kernel > gpurun_out/prof_lane_u.log 2>&1
PROF_DEFAULT=1 bash tools/prof.sh lane_g --lane-kernel > gpurun_out/prof_lane_g.log 2>&1
for t in default dist_u cfg4 cfg5 lane_u lane_g; do cat gpurun_out/prof_$t/bench.json | cut -c1-400; done
