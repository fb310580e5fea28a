for r in 1 2; do for lr in 2e-4 1e-3; do
  python bench.py --lr $lr --steps 20 --warmup 5 --no-cpu-baseline --no-infer 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lr $lr', '|', d['value'], 'tiles/s', d['ms_per_step'], 'ms | final_loss', d['final_loss'], '| halo', d['roofline']['achieved'], d.get('power'))"
done; done
