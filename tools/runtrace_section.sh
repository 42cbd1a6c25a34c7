cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 tools/gpu_only_time.py > gpurun_out/tr.log 2>&1; echo rc=$?
f=$(ls gpurun_out/tr/*/*kernel_trace.csv | head -1)
python3 tools/section_kernels.py $f 7 gru_fwd 1 gru_bwd 0 > gpurun_out/sec_loss.txt
python3 tools/section_kernels.py $f 7 gru_bwd 0 gru_bwd 1 > gpurun_out/sec_attn_bwd.txt
python3 tools/section_kernels.py $f 7 gru_bwd 1 conv_igemm_bf16_kernel 20 > gpurun_out/sec_pre_map_bwd.txt
python3 tools/section_kernels.py $f 7 gru_fwd 0 gru_fwd 1 > gpurun_out/sec_attn_fwd.txt
head -3 gpurun_out/sec_*.txt
rm -rf gpurun_out/tr
