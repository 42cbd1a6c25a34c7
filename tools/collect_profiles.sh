# build-container script: copy the evidence set of `gpurun ... bash tools/refresh_all.sh <tag>` (gpurun_out/all_<tag>/) into profiles/ under
# the names profiles/README.md lists.   usage: bash tools/collect_profiles.sh r04
TAG=${1:-r04}; S=gpurun_out/all_$TAG; D=profiles
grep -v "amdgpu.ids" $S/bench_driver_flags.json | tail -1 > $D/${TAG}_bench_bf16_driver_flags.json
grep -v "amdgpu.ids" $S/bench_default.json | tail -1 > $D/${TAG}_bench_bf16.json
grep -v "amdgpu.ids" $S/bench_sustained500.json | tail -1 > $D/${TAG}_bench_bf16_sustained500.json
grep -v "amdgpu.ids" $S/bench_dp1_rccl_one_rank.json | tail -1 > $D/${TAG}_bench_dp1_rccl_one_rank.json
cp $S/kernel_stats.csv $D/${TAG}_bench_bf16_T64N8_kernel_stats.csv
cp $S/hbm_traffic.json $D/${TAG}_bench_bf16_hbm_traffic.json
cp $S/mfma_busy.json $D/${TAG}_bench_bf16_mfma_busy.json
cp $S/update_timeline.txt $D/${TAG}_update_timeline.txt
cp $S/main_queue_breakdown.txt $D/${TAG}_bench_bf16_main_queue_breakdown.txt
for f in conv_by_layer update_sections feeder bev attn_fp8 act; do grep -v "amdgpu.ids" $S/$f.txt > $D/${TAG}_$f.txt; done
mv $D/${TAG}_conv_by_layer.txt $D/${TAG}_conv_bf16_by_layer.txt
tail -3 $S/pytest_gpu.log > $D/${TAG}_pytest_gpu_tail.txt
ls -la $D | grep ${TAG}_ | wc -l
