python -m pytest tests/test_kernels_gpu.py tests/test_attn_fp8_gpu.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -2
for v in 0 1 0 1; do
  export VDS_ATTN_TAIL_LAST=$v
  python bench.py --batch 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms']; print('tail_last=$v B=2', round(d['value'],3), round(d['ms_per_step'],1), round(d['mfma_util_step'],4), 'fwd', k.get('attn_fwd'), 'dkv', k.get('attn_bwd_dkv'), 'dq', k.get('attn_bwd_dq'))"
  python tools/bench_sampler.py 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tail_last=$v sampler', round(d['value'],3), round(d['mfma_util'],4))"
done
for v in 0 1; do VDS_ATTN_TAIL_LAST=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tail_last=$v B=12', round(d['value'],3), round(d['ms_per_step'],1))"; done
