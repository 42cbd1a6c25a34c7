# GPU-box diagnostic: socket power and clocks while the bench runs (is the update power-limited?)
cd $GRAFT_REPO_ROOT
python bench.py --steps 2500 --warmup 5 --no-cpu-baseline --no-f32 > gpurun_out/pw_bench.json 2> gpurun_out/pw_bench.err &
BP=$!
sleep 22
for i in 1 2 3 4 5 6 7 8 9 10; do
  rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)|Max Graphics" | sed 's/^/  /'
  echo ---
  sleep 2
done
wait $BP
tail -1 gpurun_out/pw_bench.json | cut -c60-140
