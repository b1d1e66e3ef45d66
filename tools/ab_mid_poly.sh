cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
 for v in default nomidpoly; do
  if [ $v = default ]; then L=""; else L="NPHM_AMD_LIB=$PWD/gpurun_tmp/libnomidpoly.so"; fi
  env $L python bench.py --workload identity --no-cpu-baseline --no-sub 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
n=d['config'].get('numerics',{})
print('$v', round(d['value'],1), round(d['roofline']['kernel_ms'],3), 'flops/pt', d['roofline'].get('executed_flops_per_point'), 'knobs', {k:n.get(k) for k in ('prune_tol','light_tol','mid_tol','sample_err','verified_err','max_err')})
"
 done
done
