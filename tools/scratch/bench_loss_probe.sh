for args in "--steps 20 --warmup 5" "--precision fp16 --classes 4 --batch 128 --steps 20 --warmup 5" "--precision fp16x3 --steps 10 --warmup 3" "--precision bf16x3 --steps 10 --warmup 3"; do
  python bench.py $args --no-cpu-baseline --no-power --no-infer 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$args', '|', d['value'], 'tiles/s', d['ms_per_step'], 'ms | final_loss', d['final_loss'], d.get('loss_scale'))"
done
