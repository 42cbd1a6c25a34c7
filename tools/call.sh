#!/bin/bash
# the round's GPU calls: tools/call.sh <n>  (each block is one gpurun command line; kept so that profiles/ can name what produced them)
case "$1" in
1) python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/c1_tests.txt
   tools/ab.sh c1_early_ego 3 30 "X=1" "WSMG_EARLY_EGO=0" > gpurun_out/c1_early_ego.txt 2>&1 ;;
2) python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -30 > gpurun_out/c2_r6.txt
   python -m pytest tests/test_gpu_policy.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q 2>&1 | tail -15 > gpurun_out/c2_tests.txt
   tools/ab.sh c2_bn 3 30 "X=1" "WSMG_BN_PRODUCER_SUMS=0" "WSMG_BN_PRODUCER_SUMS=0 WSMG_BN_BWD_SLABS=0" > gpurun_out/c2_bn.txt 2>&1 ;;
3) python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -30 > gpurun_out/c3_r6.txt
   python -m pytest tests/test_gpu_policy.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q 2>&1 | tail -15 > gpurun_out/c3_tests.txt
   tools/ab.sh c3_bn 3 30 "X=1" "WSMG_BN_PRODUCER_SUMS=0" "WSMG_BN_PRODUCER_SUMS=0 WSMG_RELU_PRODUCER_MASK=0 WSMG_CONV_INTO_CAT=0" > gpurun_out/c3_bn.txt 2>&1 ;;
4) python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -40 > gpurun_out/c4_r6.txt
   python -m pytest tests/test_gpu_kernels.py tests/test_gpu_policy.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q 2>&1 | tail -15 > gpurun_out/c4_tests.txt
   tools/ab.sh c4_bn 3 30 "X=1" "WSMG_BN_PRODUCER_SUMS=0" "WSMG_BN_PRODUCER_SUMS=0 WSMG_RELU_PRODUCER_MASK=0 WSMG_CONV_INTO_CAT=0" > gpurun_out/c4_bn.txt 2>&1
   python - > gpurun_out/c4_legs.txt 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, "ws-mgmap_amd")
import bench_legs
dev = torch.device("cuda")
for name, fn in (("cfg4", bench_legs.cfg4_bev_mapenc), ("cfg5", bench_legs.cfg5_attn_fp8)):
    r = fn(dev)
    print(name, json.dumps(r)[:3000])
PY
   python tools/stock_ops.py 3 > gpurun_out/c4_stock_ops.txt 2>&1 ;;
5) python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -12 > gpurun_out/c5_r6.txt
   python tools/breg_ab.py 30 > gpurun_out/c5_breg.txt 2>&1
   for arm in 1 0; do
     WSMG_BEV_COMPACT=$arm python - > gpurun_out/c5_cfg4_compact$arm.txt 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, "ws-mgmap_amd")
import bench_legs
r = bench_legs.cfg4_bev_mapenc(torch.device("cuda"))
print(r["us"], {k: v["us"] for k, v in r["stages"].items()})
PY
   done
   cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   for arm in new old; do
     if [ $arm = old ]; then export WSMG_BN_PRODUCER_SUMS=0; else unset WSMG_BN_PRODUCER_SUMS; fi
     rm -rf gpurun_out/st_$arm
     timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$arm -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c5_st_$arm.log 2>&1
     cp $(ls gpurun_out/st_$arm/*/*kernel_stats.csv | head -1) gpurun_out/c5_kernel_stats_$arm.csv
     rm -rf gpurun_out/st_$arm
   done
   unset WSMG_BN_PRODUCER_SUMS ;;
6) python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -12 > gpurun_out/c6_r6.txt
   python tools/igemm_tile_ab.py 30 > gpurun_out/c6_igemm.txt 2>&1
   tools/ab.sh c6_bm 3 30 "X=1" "WSMG_IGEMM_BM256=0" > gpurun_out/c6_bm.txt 2>&1 ;;
7) python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/c7_tests.txt
   python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c7_bench.json 2> gpurun_out/c7_bench.err
   tail -c 600 gpurun_out/c7_bench.err > gpurun_out/c7_bench_err_tail.txt ;;
8) python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -x -q 2>&1 | tail -5 > gpurun_out/c8_tests.txt
   WSMG_BENCH_DP_ONE_RANK=1 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-other-configs > gpurun_out/c8_dp1.json 2> gpurun_out/c8_dp1.err
   timeout 300 python3 tools/section_times.py bf16 8 > gpurun_out/c8_sections.txt 2>&1
   bash tools/runtrace_update.sh > gpurun_out/c8_runtrace.log 2>&1; cp gpurun_out/update_timeline.txt gpurun_out/c8_update_timeline.txt ;;
9) python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -6 > gpurun_out/c9_tests.txt
   timeout 900 python3 tools/bf16_vs_f32_long.py 200 8 gpurun_out/c9_long.json > gpurun_out/c9_long.txt 2>&1
   tools/ab.sh c9_modes 2 30 "X=1" > /dev/null 2>&1
   for r in 1 2; do for m in bf16 bf16+f32grad; do python3 bench.py --dtype $m --steps 30 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$m', d['ms_per_step'])"; done; done > gpurun_out/c9_modes.txt 2>&1 ;;
10) timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > gpurun_out/c10_conv.txt 2>&1
   cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   rm -rf gpurun_out/st10
   timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st10 -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c10_st.log 2>&1
   cp $(ls gpurun_out/st10/*/*kernel_stats.csv | head -1) gpurun_out/c10_kernel_stats.csv
   rm -rf gpurun_out/st10 ;;
11) python -m pytest tests/test_gpu_round6.py -x -q -k "k32 or colsum or concat" 2>&1 | tail -15 > gpurun_out/c11_tests.txt
   python -m pytest tests/test_gpu_kernels.py tests/test_gpu_policy.py -x -q 2>&1 | tail -5 >> gpurun_out/c11_tests.txt
   timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only cls_k3 > gpurun_out/c11_conv.txt 2>&1
   timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only classified >> gpurun_out/c11_conv.txt 2>&1
   WSMG_CONV_K32=0 timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only cls_k3 >> gpurun_out/c11_conv.txt 2>&1
   WSMG_CONV_K32=0 timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only classified >> gpurun_out/c11_conv.txt 2>&1
   for w in 2 4; do WSMG_CONV_K32_WGS=$w timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only cls_k3 | tail -2 >> gpurun_out/c11_conv.txt 2>&1; done
   tools/ab.sh c11_k32 3 30 "WSMG_CONV_K32=0" "WSMG_CONV_K32=1" > gpurun_out/c11_ab.txt 2>&1 ;;
12) python -m pytest tests/test_gpu_round6.py -x -q -k "k32 or colsum or concat" 2>&1 | tail -8 > gpurun_out/c12_tests.txt
   python -m pytest tests/test_gpu_kernels.py -x -q -k "lstm or bilstm" 2>&1 | tail -3 >> gpurun_out/c12_tests.txt
   for k in 1 0; do for l in cls_k3 classified; do WSMG_CONV_K32=$k timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only $l | tail -2 | head -1; done; done > gpurun_out/c12_conv.txt 2>&1
   python3 -c "import torch; print(torch.cuda.Stream.priority_range())" > gpurun_out/c12_prio.txt 2>&1
   tools/ab.sh c12_prio 3 30 "X=1" "WSMG_EXP_PRIO_INSTRUCTION=1" "WSMG_EXP_PRIO_INSTRUCTION=1 WSMG_EXP_PRIO_DECODER=1" "WSMG_EXP_PRIO_INSTRUCTION=-1 WSMG_EXP_PRIO_DECODER=-1" > gpurun_out/c12_ab.txt 2>&1 ;;
13) python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_kernels.py tests/test_gpu_policy.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q 2>&1 | tail -8 > gpurun_out/c13_tests.txt
   timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > gpurun_out/c13_conv_new.txt 2>&1
   WSMG_LIB=$GRAFT_REPO_ROOT/ws-mgmap_amd/wsmgmap/lib/libwsmgmap_prev.so timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > gpurun_out/c13_conv_prev.txt 2>&1
   tools/ab.sh c13_lib 4 30 "WSMG_LIB=$GRAFT_REPO_ROOT/ws-mgmap_amd/wsmgmap/lib/libwsmgmap_prev.so" "X=1" > gpurun_out/c13_ab.txt 2>&1 ;;
16) python -m pytest tests/test_gpu_round6.py tests/test_gpu_policy.py tests/test_gpu_round3.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/c16_tests.txt
   for db in 1 0; do WSMG_CONV_K32_DB=$db timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only cls_k3 | tail -2 | head -1; done > gpurun_out/c16_conv.txt 2>&1
   for w in 2 3 6 8; do WSMG_CONV_K32_WGS2=$w timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only cls_k3 | tail -2 | head -1; done >> gpurun_out/c16_conv.txt 2>&1
   tools/ab.sh c16_db 3 30 "WSMG_CONV_K32_DB=0" "X=1" "WSMG_CONV_K32=0" > gpurun_out/c16_ab.txt 2>&1 ;;
17) cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   for arm in dflt nodec; do
     if [ $arm = nodec ]; then export WSMG_DECODER_STREAMS=0; else unset WSMG_DECODER_STREAMS; fi
     rm -rf gpurun_out/st17
     timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st17 -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c17_$arm.log 2>&1
     cp $(ls gpurun_out/st17/*/*kernel_stats.csv | head -1) gpurun_out/c17_kernel_stats_$arm.csv
     rm -rf gpurun_out/st17
   done
   unset WSMG_DECODER_STREAMS ;;
18) python -m pytest tests/test_gpu_round6.py tests/test_gpu_kernels.py -x -q -k "feeder_layout or bev or map_ or rollout" 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/c18_tests.txt
   python -m pytest tests/test_gpu_policy.py -x -q -k "rollout or act" 2>&1 | grep -E "passed|failed|Error" | tail -3 >> gpurun_out/c18_tests.txt
   timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c18_bench.json 2> gpurun_out/c18_bench.err ;;
19) python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_policy.py -x -q -k "cls_tail or update_path or g3 or gradient" 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/c19_tests.txt
   cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   rm -rf gpurun_out/st19
   timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st19 -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c19_st.log 2>&1
   cp $(ls gpurun_out/st19/*/*kernel_stats.csv | head -1) gpurun_out/c19_kernel_stats.csv
   rm -rf gpurun_out/st19
   tools/ab.sh c19_ab 2 30 "X=1" > gpurun_out/c19_ab.txt 2>&1 ;;
20) python -m pytest tests/test_gpu_round6.py -x -q -k "convtranspose" 2>&1 | tail -12 > gpurun_out/c20_tests.txt
   python -m pytest tests/test_gpu_policy.py tests/test_gpu_round3.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3 >> gpurun_out/c20_tests.txt
   for k in 1 0; do WSMG_CONVT_K4S2=$k timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only convT | tail -2 | head -1; done > gpurun_out/c20_conv.txt 2>&1
   for w in 2 4 12; do WSMG_CONVT_WGS=$w timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only convT | tail -2 | head -1; done >> gpurun_out/c20_conv.txt 2>&1
   tools/ab.sh c20_ab 3 30 "WSMG_CONVT_K4S2=0" "X=1" > gpurun_out/c20_ab.txt 2>&1 ;;
22) python -m pytest tests/test_gpu_kernels.py tests/test_gpu_policy.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/c22_tests.txt
   tools/ab.sh c22_ab 4 30 "WSMG_BN_RED8=0" "X=1" > gpurun_out/c22_ab.txt 2>&1
   cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   rm -rf gpurun_out/st22
   timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st22 -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c22_st.log 2>&1
   cp $(ls gpurun_out/st22/*/*kernel_stats.csv | head -1) gpurun_out/c22_kernel_stats.csv
   rm -rf gpurun_out/st22 ;;
23) python -m pytest tests/test_gpu_round6.py tests/test_gpu_kernels.py tests/test_gpu_round2.py -x -q -k "upsample or small_ops or strided or slices" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 > gpurun_out/c23_tests.txt
   python -m pytest tests/test_gpu_policy.py tests/test_gpu_round3.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3 >> gpurun_out/c23_tests.txt
   cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   rm -rf gpurun_out/st23
   timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st23 -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c23_st.log 2>&1
   cp $(ls gpurun_out/st23/*/*kernel_stats.csv | head -1) gpurun_out/c23_kernel_stats.csv
   rm -rf gpurun_out/st23 ;;
25) python -m pytest tests/test_gpu_round6.py -x -q -k "concat" 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/c25_tests.txt
   cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
   rm -rf gpurun_out/st25
   timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st25 -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/c25_st.log 2>&1
   cp $(ls gpurun_out/st25/*/*kernel_stats.csv | head -1) gpurun_out/c25_kernel_stats.csv
   rm -rf gpurun_out/st25
   tools/ab.sh c25_ab 3 30 "X=1" > gpurun_out/c25_ab.txt 2>&1 ;;
esac
