cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=${1:-enc0}
timeout 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pe_a -- python3 tools/bench_conv.py --reps 1 --only $L --dtype bf16 > gpurun_out/pe_a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pe_b -- python3 tools/bench_conv.py --reps 1 --only $L --dtype bf16 > gpurun_out/pe_b.log 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d gpurun_out/pe_c -- python3 tools/bench_conv.py --reps 1 --only $L --dtype bf16 > gpurun_out/pe_c.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES --output-format csv -d gpurun_out/pe_d -- python3 tools/bench_conv.py --reps 1 --only $L --dtype bf16 > gpurun_out/pe_d.log 2>&1
ls gpurun_out/pe_*/runc/ | head -20
