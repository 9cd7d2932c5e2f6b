# E-step lanes A/B: bench.py with VLGP_ESTEP_LANES = 1 .. 4 (same build, same box); WL=C2 etc. selects the workload
for L in 1 2 3 4; do
VLGP_ESTEP_LANES=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${WL:+--workload $WL} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes', $L, round(d['value'],2), round(d['ms_per_step'],3), 'E', round(d['ms_per_e_step'],3), 'M', round(d['ms_per_m_step'],3), 'H', round(d['ms_per_h_step'],3), d['h_step']['rounds_per_step'])"
done
