"""Key numbers of a bench.py JSON line: python scripts/show_bench.py <file>"""
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][0]
print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 2), d.get("time_shares"))
r = d.get("roofline", {})
print("roofline", {k: r.get(k) for k in ("achieved", "frac", "traffic_per_step", "launches", "time_share")})
if "parity" in d: print("parity", d["parity"].get("active_set_hamming"), d["parity"].get("max_rel_err_vs_fp64_oracle"))
if "first_move_output" in d: print("first moves", round(d["first_move_output"]["value"]), d["first_move_output"]["active_sets_equal_to_sequence_call"])
for k, v in d.get("sweep_sx", {}).items(): print(k, round(v["value"]), round(v["ms_per_step"], 2), v["status_hist"])
c = d.get("configs", {}).get("cstrs_10k")
if c: print("cstrs_10k", round(c["value"]), c["ms_per_step"], c.get("ms_per_step_of_each_handle"))
if "chains_task" in d: print("chains_task wall", d["chains_task"]["wall_s"], "chains", d["chains"]["ms_per_step"])
print(d.get("leg_seconds"))
