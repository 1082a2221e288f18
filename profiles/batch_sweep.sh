# Resident throughput against batch size (run on the GPU box from the repository root): bash profiles/batch_sweep.sh > gpurun_out/batch_sweep.txt
echo "# resident throughput against batch size (acct-d8 machine proofs, format v15, one MI355X): python bench.py --batch B --steps S --no-cpu-baseline --skip-single"
for B in 1 2 4 8 9 16 32 64 128 192 224; do
  S=5; if [ $B -le 16 ]; then S=20; fi
  python3 bench.py --batch $B --steps $S --warmup 2 --no-cpu-baseline --skip-single 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['config']['batch_per_gpu']
print('batch %4d  %7.1f proofs/s  %8.2f ms/step  %6.2f ms/proof' % (b, d['value'], d['ms_per_step'], d['ms_per_step']/b))"
done
