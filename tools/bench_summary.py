import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "value", d["value"], "one_handle", d["one_handle"]["value"], "single", d["single_batch"]["value"])
print("  breakdown", d["breakdown_ms_per_step"])
for f in ("ped6", "mix11"):
    l = d["latency"][f]
    print("  ", f, "b1 mean/med/p95", l["plan_b1"]["mean_ms"], l["plan_b1"]["median_ms"], l["plan_b1"]["p95_ms"], "tail_us/it", l["plan_b1_phases"]["tail_us_per_iteration"], "prologue", l["plan_b1_phases"]["load_initguess_export_ms"], "b64", l["solve_batch"]["mean_ms"])
