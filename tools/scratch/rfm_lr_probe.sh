for lr in 1e-3 1e-4 1e-5; do
python bench.py --workload rfm --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --lr $lr 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('lr $lr', d['value'], d['ms_per_step'], d['final_losses'])"
done
