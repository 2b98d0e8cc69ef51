"""Developer helper: the numbers of a bench.py line at a glance (value, step, dominant kernel, roofline fraction, end to end, CPU baseline).
usage: python3 scripts/show_bench.py FILE.json"""
import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["cpu_baseline"])
for k,v in d["formats"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("kernel_ms_avg"), v.get("roofline",{}).get("frac"), v.get("end_to_end",{}).get("value"), v.get("end_to_end",{}).get("windowed",{}).get("value"), v.get("cpu_baseline",{}).get("value"))
