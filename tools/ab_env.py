"""Across-process A/B of an environment switch of libgcmf (alternating fresh processes; 'tools/ab_tuning.py' is the in-process form).

    python tools/ab_env.py GCMF_ARENA 0 1 [configs=3,2,4] [n=6]
"""
import json, os, subprocess, sys
var, a, b = sys.argv[1:4]
cfgs = [int(c) for c in (sys.argv[4] if len(sys.argv) > 4 else "3,2,4").split(",")]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 6
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for i in range(n):
    for val in (a, b):
        for cfg in cfgs:
            env = dict(os.environ, **{var: val})
            out = subprocess.run([sys.executable, os.path.join(here, "bench.py"), "--config", str(cfg), "--no-cpu", "--no-extra"],
                                 capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
            d = json.loads(out)
            res.setdefault((cfg, val), []).append((d["value"] / 1e9, d["roofline"]["avg_launch_ms"] * 1e3))
for (cfg, val), v in sorted(res.items()):
    g = sorted(x for x, _ in v); us = sorted(y for _, y in v)
    print(f"config {cfg} {var}={val}: G mean {sum(g)/len(g):.1f} median {g[len(g)//2]:.1f} | dominant launch us " + " ".join(f"{u:.1f}" for u in us))
