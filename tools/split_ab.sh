for wl in C3s4 C3s8 C2; do for cfg in "0 1" "1 1" "1 2"; do set -- $cfg
VLGP_ESTEP_SPLIT=$1 VLGP_ESTEP_LANES=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl split $1 lanes $2', round(d['value'],2), round(d['ms_per_step'],3), 'E', round(d['ms_per_e_step'],3), 'H', round(d['ms_per_h_step'],3))"
done; done
