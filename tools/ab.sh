#!/bin/bash
# The one A/B driver: bench.py arms, interleaved, on ONE box (boxes differ by ~3 %, so only interleaved arms on one box compare).
#   tools/ab.sh <out-name> <rounds> <steps> "<arm 0 env>" "<arm 1 env>" ...       (an arm is a space-separated list of VAR=value; "X=1" = defaults)
# writes gpurun_out/<out-name>/arm<i>_<round>.json and prints ms_per_step per arm and round.
NAME=$1; ROUNDS=$2; STEPS=$3; shift 3
O=gpurun_out/$NAME; mkdir -p $O
B="bench.py --gpus 1 --steps $STEPS --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
for rep in $(seq 1 $ROUNDS); do
  i=0
  for arm in "$@"; do
    env $arm python3 $B > $O/arm${i}_$rep.json 2> $O/arm${i}_$rep.err; i=$((i+1))
  done
done
python3 - "$O" "$ROUNDS" "$@" <<'PY'
import json, sys
O, rounds, arms = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for i, t in enumerate(arms):
    v = []
    for rep in range(1, rounds + 1):
        try:
            d = json.loads(open("%s/arm%d_%d.json" % (O, i, rep)).read().strip().splitlines()[-1]); v.append(d["ms_per_step"])
        except Exception:
            v.append(None)
    print("%-60s %s" % (t, v))
PY
