import json,sys
d=json.load(open(sys.argv[1]))
print("steps/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],5), "frac", round(d["roofline"]["frac"],4))
for r in d["roofline"]["launches"]: print(f'{r["launch"]:32s} {r["avg_us"]:7.2f}')
print("eager_step_us", d["roofline"]["eager_step_us"])
