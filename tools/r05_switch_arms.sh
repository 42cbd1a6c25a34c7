#!/bin/bash
# defaults re-checked at the end of round 5, one box, interleaved
O=gpurun_out/r05_switch_arms; mkdir -p $O
B="bench.py --gpus 1 --steps 30 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
ARMS=("X=1" "WSMG_DECODER_STREAMS=0" "WSMG_ENC_PROJ_SIDE=0" "WSMG_EARLY_DEDUP=0" "WSMG_RECURRENT_CHUNKS=8" "WSMG_CONV_WIN3_MIXED=0" "WSMG_WIN3W_STAGES=3")
for rep in a b c; do
  i=0
  for arm in "${ARMS[@]}"; do
    env $arm python3 $B > $O/arm${i}_$rep.json 2> $O/arm${i}_$rep.err; i=$((i+1))
  done
done
python3 - <<PY
import json
arms = ["default", "DECODER_STREAMS=0", "ENC_PROJ_SIDE=0", "EARLY_DEDUP=0", "RECURRENT_CHUNKS=8", "CONV_WIN3_MIXED=0", "WIN3W_STAGES=3"]
for i, t in enumerate(arms):
    v = []
    for rep in "abc":
        try:
            d=json.loads(open("$O/arm%d_%s.json" % (i, rep)).read().strip().splitlines()[-1]); v.append(d["ms_per_step"])
        except Exception as e:
            v.append(None)
    print("%-20s" % t, v)
PY
