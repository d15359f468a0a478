#!/usr/bin/env python3
"""A/B of the broadband process_signal schedule switches (stream_bb.hip: APV_BB_FRONT2, APV_BB_FRONT_THREAD).  Each setting runs
in a child of its own (the switches are read once per process); the timing is bench.py's (also_cfg1 / also_reftest).
usage: bb_front_ab.py [reps]      -> one JSON line per (setting, repetition)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import bench
a = bench.also_cfg1(0)
b = bench.also_reftest(0)
pick = lambda r: {k: round(r[k]["ms_per_hop"], 4) for k in ("process_input_buffers", "process_signal", "process_signal_out")}
print(json.dumps({"cfg1": pick(a), "n800": pick(b)}))
""" % ROOT


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    settings = [("default", {}), ("front2=0", {"APV_BB_FRONT2": "0"}), ("thread=0", {"APV_BB_FRONT_THREAD": "0"}),
                ("front2=0,thread=0", {"APV_BB_FRONT2": "0", "APV_BB_FRONT_THREAD": "0"})]
    for rep in range(reps):
        for name, env in settings:
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
            print(json.dumps({"setting": name, "rep": rep, "result": json.loads(line) if line.startswith("{") else line}), flush=True)


if __name__ == "__main__":
    main()
