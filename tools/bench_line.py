import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"]["step_breakdown_ms_rank0"])
