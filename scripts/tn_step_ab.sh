F="--no-hifigan --no-fp32 --no-roofline --no-forward-only --no-cpu-baseline --no-inference --steps 40 --warmup 5"
for i in 1 2 3; do
  for t in 64129 0; do
    echo -n "KANTTS_TN_TILE=$t  "
    KANTTS_TN_TILE=$t python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
