for s in 32 64 128 256; do for pipe in 0 1; do
  TREPAMD_NEWTON_PIPELINE=$pipe timeout 600 python bench_discopt.py --seeds $s --horizon 1000 --quasi 1 --newton 1 > /tmp/d.json 2>/tmp/d.err
  python -c "
import json; d=json.load(open('/tmp/d.json'))
print('seeds %4d pipeline %s  %.1f it/s  quasi %.4f s  newton %.4f s  failures %d' % ($s, '$pipe', d['iters_per_s'], d['s_per_batched_quasi_step'], d['s_per_batched_newton_step'], d['armijo_failures']))" || tail -3 /tmp/d.err
done; done
for c in 4 12 16; do TREPAMD_NEWTON_CHUNKS=$c TREPAMD_NEWTON_PIPELINE=1 timeout 600 python bench_discopt.py --seeds 32 --horizon 1000 --quasi 1 --newton 1 > /tmp/d.json 2>/tmp/d.err; python -c "
import json; d=json.load(open('/tmp/d.json'))
print('seeds 32 chunks $c  %.1f it/s  newton %.4f s' % (d['iters_per_s'], d['s_per_batched_newton_step']))"; done
