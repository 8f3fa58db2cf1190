# round 6, GPU call 11: why is the chain pass's tail ~15 % faster with the evaluation pass beside it?  Clocks sampled during a cbox + medium render in reference-order streams,
# overlap on / off; and the launch policy of the evaluation pass (smaller, more frequent launches) against the chain pass's time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g11; mkdir -p $O
sample() { while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --csv 2>/dev/null | tr '\n' ' ' | cut -c1-400)"; sleep 0.15; done; }
for ov in 0 1; do
  if [ $ov = 1 ]; then export RL_NO_OVERLAP=1; else unset RL_NO_OVERLAP; fi
  sample > $O/clocks_noov$ov.txt & SP=$!
  echo "-- RL_NO_OVERLAP=$ov $(date +%s.%N)" >> $O/log.txt
  REPS=3 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175 >> $O/log.txt
  echo "   end $(date +%s.%N)" >> $O/log.txt
  kill $SP
done
unset RL_NO_OVERLAP
rocm-smi --showclocks 2>&1 | head -30 > $O/rocm_smi_idle.txt
{
echo "== launch policy of the evaluation pass beside the chain pass (cbox + medium)"
for rep in 1 2; do
for kv in "RL_EVAL_MIN=64 RL_EVAL_DIV=5" "RL_EVAL_MIN=16 RL_EVAL_DIV=40" "RL_EVAL_MIN=8 RL_EVAL_DIV=200" "RL_EVAL_MIN=256 RL_EVAL_DIV=2" "RL_EVAL_SPLIT=1" "RL_EVAL_SPLIT=16"; do
  echo "-- $kv"; env $kv REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175
done; done
} >> $O/log.txt 2>&1
cat $O/log.txt
python - <<PY
import re
for ov in (0, 1):
    vals = []
    for l in open("$O/clocks_noov%d.txt" % ov):
        m = re.findall(r"\((\d+)Mhz\)", l)
        if m: vals.append(tuple(int(x) for x in m[:6]))
    print("noov", ov, "samples", len(vals), "distinct clock tuples (first 6 fields):", sorted(set(vals))[:12])
PY
head -3 $O/clocks_noov0.txt | cut -c1-500
