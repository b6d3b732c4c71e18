import sys, time, os, numpy as np
sys.path.insert(0, "/root/repo")
from pyfastx_amd import synth
print("native", synth._native() is not None, "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
rng = np.random.default_rng(1)
raw = np.frombuffer(b"ACGTacgtN\n", dtype=np.uint8)[rng.choice(10, 600_000_000, p=[.2,.2,.2,.2,.04,.04,.04,.04,.02,.02])]
for th in (32, 64, 128, 192, 256):
    t = time.time(); x = synth.bgzf_compress_parallel(raw, procs=th); print(th, "threads: 600 MB in", round(time.time() - t, 2), "s", len(x), flush=True)
