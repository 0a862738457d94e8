import json, sys
tag = sys.argv[1]
O = "gpurun_out/cons_ab/"
print(open(O + tag + ".tests.txt").read().strip())
for ln in open(O + tag + ".trace.txt"):
    if "workgroups, span" in ln or "in flight" in ln: print(ln.rstrip()[15:])
for f in ("bench2", "bench1"):
    d = json.load(open(O + "%s.%s.json" % (tag, f)))
    print(f, "ms/step %.3f" % d["ms_per_step"], "value %.3g" % d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "verified", d.get("config", {}).get("verified", d.get("verified")))
