timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/b.json || exit 1
python -c "
import json
j=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('%.4g sims/s'%j['value'], 'ms/step %.4f'%j['ms_per_step'], 'kernel_ms %.4f'%j['roofline']['kernel_ms'])"
MZS_LIB=$PWD/muzero_amd/libmzsearch_diag.so timeout -k 10 200 python profiles/diag_phases.py
