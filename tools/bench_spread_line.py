import sys, json
d = json.loads(sys.stdin.read())
bt = d.get("behavior_train") or {}
fs = (bt.get("flow_stage") or {}).get("ms_per_step")
cs = (bt.get("cvae_stage") or {}).get("ms_per_step")
bh = (d.get("behavior") or {}).get("flow_reverse_ms")
print(round(d["value"], 1), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 3), fs, cs, bh)
