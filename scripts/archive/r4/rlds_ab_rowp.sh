for i in 1 2 3; do
  for r in 1 2 0; do
    python3 bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline --sharded-steps 0 --no-power-probe --residual-lds $r 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
pk=d.get('per_kernel') or d['roofline'].get('per_kernel')
print('residual_lds=$r  frame %.3f ms | ' % d['ms_per_step'] + '  '.join('%s %.3f' % (k.replace('conv3x3_pc',''), v['ms_total']) for k, v in pk.items()))
"
  done
done
