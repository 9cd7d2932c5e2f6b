for LN in 1 2 3 4; do echo "streams $LN"; VLGP_ESTEP_LANEPT=1 VLGP_ESTEP_LANES=$LN OMS=5e-3,8e-3 python tools/estep_rank_classes.py 2>&1 | grep omega; done
echo "prio 0"; VLGP_LANE_PRIO=0 VLGP_ESTEP_LANEPT=1 OMS=5e-3 python tools/estep_rank_classes.py 2>&1 | grep omega
