import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line); c = r.get("complex128_kernels") or {}
        print(r["config"]["n_sites"], r["config"]["vectors_per_gpu"], r["roofline"]["kernel"], "strip", r["roofline"].get("strip_rows"),
              round(r["value"]), "steps/s", round(r["roofline"]["achieved"]), "GB/s", round(r["roofline"]["launch_ms"], 3), "ms grid", r["roofline"]["grid"],
              "| streamed:", round((r.get("streamed_blocks_kernels") or {}).get("value", 0)), round((r.get("streamed_blocks_kernels") or {}).get("achieved_GBps", 0)), "| complex:", round(c.get("value", 0)), round(c.get("achieved_GBps", 0)))
    elif "failed" in line or "Error" in line:
        print(line.strip())
