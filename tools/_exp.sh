python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "late_reconstruction or full_size or mixed_configuration" 2>&1 | tail -5
for m in lane wave; do
LC3GPU_RECON=$m python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r03_d.json 2>gpurun_out/r03_d.err
python -c "
import json
j=json.load(open('gpurun_out/r03_d.json')); k=j['kernel_ms']; print('$m', j['value'], k['lc3_parse_kernel'], k['lc3_recon_kernel'], k['lc3_tns_kernel'], k['lc3_decode_kernel'], j['parity'])"
done
