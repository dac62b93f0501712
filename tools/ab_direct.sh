for i in 1 2 3; do
for flag in "" "--direct"; do
python3 bench.py --no-front --cpu-sample 0 $flag 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$flag]', round(r['ms_per_step'], 3), 'ode', round(r['config']['ode_ms'], 3), 'pde', round(r['config']['pde_ms'], 3), 'sum', round(r['config']['ode_ms']+r['config']['pde_ms'],3), 'gap', round(r['ms_per_step']-r['config']['ode_ms']-r['config']['pde_ms'],3), flush=True)"
done; done
