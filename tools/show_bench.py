import json,sys
d=json.load(open(sys.argv[1])); print(round(d["value"],1), round(d["ms_per_step"],3), "E",round(d["ms_per_e_step"],3), "M",round(d["ms_per_m_step"],3), "H",round(d["ms_per_h_step"],3))
for k,v in d["kernels"].items(): print("  ",k[:58], round(v.get("avg_ms",0)*1000,1),"us x", v.get("launches"))
