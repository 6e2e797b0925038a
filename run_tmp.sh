python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --rk4-steps 0 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step %.3f rhs_frac %.3f' % (d['ms_per_step'], r['rhs']['frac']), {k.split('(')[0]:round(v,2) for k,v in r['kernels_ms'].items()})"; }
run A=1
run OMEGA_FUSE_FINAL=0
