"""One-line summary of a bench.py JSON line (tools/gpu_lease.sh)."""
import json
import sys

try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = "eval %.0f img/s %.4f ms" % (j["value"], j["ms_per_step"])
    r = j.get("roofline") or {}
    if r:
        s += " | %s %.1f us/launch frac %.3f | step frac %.3f" % (r["kernel"], r["us_per_launch"], r["frac"], r["whole_step"]["frac"])
    for k in ("train_step", "train_step_bf16"):
        if j.get(k):
            s += " | %s %.2f ms" % (k, j[k]["ms_per_step"])
    u = j.get("train_step_unpruned_net")
    if u and "error" not in u:
        s += " | unpruned " + " / ".join("%s %.2f ms" % (a, u[a]["ms_per_step"]) for a in ("fp32", "bf16") if a in u)
    if j.get("latency_b1") and "error" not in j["latency_b1"]:
        s += " | b1 %s" % {k: v for k, v in j["latency_b1"].items() if "ms" in k}
    print(s)
except Exception as e:
    print("FAILED", type(e).__name__, e)
    try:
        print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
    except Exception:
        pass
