for lib in liblc3gpu.so liblc3gpu_eko1.so liblc3gpu_eko16.so liblc3gpu_eko24.so liblc3gpu_ko8.so liblc3gpu_ko16.so; do
  for g in 0 1; do
  LC3GPU_GENERIC=$g LC3GPU_LIB=$lib python3 bench.py --arrangement single --no-parity --no-cpu-baseline --no-overlap-probe --sustain-seconds 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib generic=$g', {k[4:-7]: round(v, 4) for k, v in j['kernel_ms'].items() if v > 0})"
  done
done
