import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
for rep in range(2):
    a = bench.also_cfg3(0)
    b = bench.also_cfg1(0)
    print(json.dumps({"where": os.getcwd().split("/")[-1], "cfg3_pib": round(a["process_input_buffers"]["ms_per_hop"], 4), "cfg3_sig": round(a["process_signal"]["ms_per_hop"], 4),
                      "cfg1_pib": round(b["process_input_buffers"]["ms_per_hop"], 4), "cfg1_sig": round(b["process_signal"]["ms_per_hop"], 4), "cfg1_sig_out": round(b["process_signal_out"]["ms_per_hop"], 4)}), flush=True)
