#!/bin/bash
# Round 6: the driver's exact bench command, N times on one box, each under a time limit, stderr (stage lines) kept.
# A poller mimics the driver's rocm-smi sampling beside the run.
out=gpurun_out/${1:-r06_repro_loop}; n=${2:-6}; lim=${3:-150}
mkdir -p $out
( while true; do rocm-smi --showuse --json > $out/smi.json 2>/dev/null; sleep 5; done ) &
poll=$!
for i in $(seq 1 $n); do
  s=$(date +%s.%N)
  timeout -s KILL $lim python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/out_$i.json 2> $out/err_$i.txt
  rc=$?
  e=$(date +%s.%N)
  echo "run $i rc=$rc wall=$(python3 -c "print(round($e - $s, 1))") last_stage=$(grep '^\[bench\]' $out/err_$i.txt | tail -1)" | tee -a $out/summary.txt
done
kill $poll
