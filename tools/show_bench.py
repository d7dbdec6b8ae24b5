"""Print the headline and the per-kernel table of a bench.py JSON line."""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s = %.1f %s   %.4f ms/step" % (j["metric"], j["value"], j["unit"], j["ms_per_step"]))
k = j.get("kernel_us_per_iter", {})
print("  ".join("%s %.1f" % (a.replace("k_", ""), b) for a, b in k.items()), " | sum %.1f" % sum(k.values()))
