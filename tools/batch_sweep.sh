for b in 8 16 32 64 128; do
  python bench.py --no-cpu-baseline --batch $b --steps 10 2>/dev/null > gpurun_out/sweep_$b.json
  python -c "import json; d=json.load(open('gpurun_out/sweep_$b.json')); print('B', $b, d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
