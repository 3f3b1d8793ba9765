import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1:], round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"], 2), "ms", "launches", d["roofline"]["launches"],
      "TF", round(d["roofline"]["achieved"], 1), d["stage_ms"])
