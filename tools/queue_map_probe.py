"""Which hardware queue does each stream of a context get?  (DESIGN.md section 5 "the queue rule"; docs/HISTORY.md section 4 "Streams and hardware queues")

Runs a child with AMD_LOG_LEVEL=4 that creates contexts one after the other and keeps the runtime's own lines about
its queue pool: "acquireQueue refCount: <hsa queue> (<users>)", "Selected queue refCount: ..." (an existing queue is
handed out again) and "Number of allocated hardware queues with low priority: .., with normal priority: .., with high
priority: .., maximum per priority is: ..".  Nothing is computed.

    python tools/queue_map_probe.py [n_contexts] > gpurun_out/queue_map.txt
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
from gauss_amd import hotpath
ctxs = []
for k in range(%(n)d):
    sys.stderr.write("##### creating context %%d\n" %% k); sys.stderr.flush()
    ctxs.append(hotpath.Context(0))
    sys.stderr.write("##### context %%d made: gauss_hip_queues says %%s\n" %% (k, ctxs[-1].queues())); sys.stderr.flush()
for k, c in enumerate(ctxs):
    sys.stderr.write("##### closing context %%d\n" %% k); sys.stderr.flush()
    c.close()
"""


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    env = dict(os.environ, AMD_LOG_LEVEL="4")
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, n=n)], env=env, capture_output=True, text=True, timeout=600)
    keep = ("#####", "acquireQueue", "Selected queue", "releaseQueue", "hardware queue", "Deleting hardware")
    for line in out.stderr.splitlines():
        if any(k in line for k in keep):
            print(line[-220:])
    print("exit code", out.returncode)


if __name__ == "__main__":
    main()
