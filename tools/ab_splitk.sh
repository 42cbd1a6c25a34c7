# rollout step under split-K plan variants (one box): bash tools/ab_splitk.sh
for cfg in "256 4 128" "512 4 128" "512 2 128" "256 2 128" "512 4 256" "1024 2 256" "256 4 128"; do
  set -- $cfg
  echo "target=$1 minsteps=$2 maxtiles=$3: $(WSMG_SPLITK_TARGET=$1 WSMG_SPLITK_MINSTEPS=$2 WSMG_SPLITK_MAXTILES=$3 python tools/bench_act.py 2>/dev/null | grep 'bf16: act() as one' | sed 's/(wsmgmap.graph.GraphedAct)//' | cut -c1-60 | tr '\n' '|')"
done
