# GPU-box script (round 4, call 2): evidence set (bench line, kernel stats, PMC folds) + ordered timeline + queue breakdown
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r04a > gpurun_out/refresh_r04a.log 2>&1
tail -20 gpurun_out/refresh_r04a.log | cut -c1-600
bash tools/runtrace_update.sh
bash tools/runtrace.sh > gpurun_out/runtrace.log 2>&1
