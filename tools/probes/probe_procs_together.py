"""tools/probes/probe_concurrent_procs.py, the 'together' half only: three processes (one stream each) run the same
forward + backward 30 times at once; prints, per process, the number of DISTINCT gradient hashes (1 = deterministic).
The children inherit the environment, so `SLNLP_SPLITK_MODE=n python tools/probes/probe_procs_together.py` bisects the split-K
meeting point (csrc/gemm_planes.hip, splitk_mode)."""
import os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
child = os.path.join(HERE, "probe_concurrent_procs.py")
t = time.time() + float(sys.argv[1] if len(sys.argv) > 1 else 25)
ps = [subprocess.Popen([sys.executable, child, "child", str(s), str(t)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for s in (1, 2, 3)]
out = [p.communicate()[0].strip() for p in ps]
print(f"SLNLP_SPLITK_MODE={os.environ.get('SLNLP_SPLITK_MODE', '0')} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '-')}  distinct hashes per process:",
      [o.split()[1] if o else "?" for o in out], out)
