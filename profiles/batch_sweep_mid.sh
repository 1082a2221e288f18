for B in 33 48 64 96 128 129; do
  python3 bench.py --batch $B --steps 10 --warmup 2 --no-cpu-baseline --skip-single 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['config']['batch_per_gpu']
print('batch %4d  %7.1f proofs/s  %8.2f ms/step  %6.2f ms/proof' % (b, d['value'], d['ms_per_step'], d['ms_per_step']/b))"
done
