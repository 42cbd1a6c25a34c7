cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tile in 21 42; do
export WSMG_WGRAD_TILE=$tile
timeout 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pw_a$tile -- python3 tools/bench_conv.py --reps 1 --only cated --dtype bf16 > gpurun_out/pw_a$tile.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pw_b$tile -- python3 tools/bench_conv.py --reps 1 --only cated --dtype bf16 > gpurun_out/pw_b$tile.log 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum --output-format csv -d gpurun_out/pw_c$tile -- python3 tools/bench_conv.py --reps 1 --only cated --dtype bf16 > gpurun_out/pw_c$tile.log 2>&1
done
ls gpurun_out/pw_*
