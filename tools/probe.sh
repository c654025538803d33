for cb in 60 50 40 30; do
  echo "== RT_COST_BOX=$cb"
  RT_COST_BOX=$cb timeout 200 python tests/devtools/perf.py --configs c4 --reps 10 --no-check
  RT_COST_BOX=$cb timeout 200 python tests/devtools/perf.py --configs c5 --size 7680x4320 --reps 3 --no-check
done
